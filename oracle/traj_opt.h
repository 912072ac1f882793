// oracle/traj_opt.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// CPU restatement of the reference's trajectory optimizer
// (reference optimizer/trajectory_optimizer.cc, "TO.cc" below; every method
// cites the lines it follows).  Plain C++17 + OpenMP, fp64, no dependencies.
// It doubles as the CPU baseline timed by bench.py (`cpu_baseline.kind="port"`):
// the OpenMP `parallel for` regions sit exactly where the reference has them
// (TO.cc:214-216, 455-457).
//
// PARITY STATUS: pinned against the reference's closed-form tests
// (optimizer/test/trajectory_optimizer_test.cc:848-998, 1058-1150, 1155-1246,
// 1251-1304, 1314-1386, 1394-1443), all of
// optimizer/test/penta_diagonal_solver_test.cc, and the spinner end-to-end
// golden (python_bindings/test/trajectory_optimizer_test.py:84-85) in
// tests/test_oracle_*.py.  The reference itself cannot be built here (needs
// Drake v1.30.0 + Eigen, neither in /root/reference nor in the image), so for
// hopper / mini_cheetah / allegro absolute values are "parity unpinned"
// (SURVEY.md §8c, Appendix D).
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

#include "penta.h"
#include "rigid_body.h"

namespace oracle {

using Vec = std::vector<double>;

enum LinesearchMethod { kArmijo = 0, kBacktracking = 1 };
enum SolverMethod { kLinesearch = 0, kTrustRegion = 1 };
enum GradientsMethod { kForwardDifferences = 0, kCentralDifferences, kCentralDifferences4, kAutoDiff, kNoGradients };
enum ScalingMethod { kSqrt = 0, kAdaptiveSqrt, kDoubleSqrt, kAdaptiveDoubleSqrt };
enum LinearSolverType { kDenseLdlt = 0, kPentaDiagonalLu = 1 };
enum SolverFlag { kSuccess = 0, kLinesearchMaxIters, kFactorizationFailed, kMaxIterationsReached };
enum ConvergenceReason {
  kNoConvergenceCriteriaSatisfied = 0,
  kCostReductionCriterionSatisfied = 1,
  kGradientCriterionSatisfied = 2,
  kSateCriterionSatisfied = 4
};

// reference optimizer/solver_parameters.h:64-167 (defaults identical)
struct Params {
  bool check_convergence = false;
  double rel_cost_reduction = 0, abs_cost_reduction = 0, rel_gradient_along_dq = 0, abs_gradient_along_dq = 0,
         rel_state_change = 0, abs_state_change = 0;
  int method = kTrustRegion;
  int linesearch_method = kArmijo;
  int max_iterations = 100;
  int max_linesearch_iterations = 50;
  int gradients_method = kForwardDifferences;
  int linear_solver = kPentaDiagonalLu;
  bool normalize_quaternions = false;
  bool verbose = false;
  bool scaling = true;
  int scaling_method = kDoubleSqrt;
  bool equality_constraints = true;
  double Delta0 = 1e-1, Delta_max = 1e5;
  int num_threads = 1;
};

// reference optimizer/problem_definition.h:24-59 (dense weights, column-major)
struct Problem {
  int N = 0;
  double dt = 0;
  Vec q_init, v_init, Qq, Qv, Qf_q, Qf_v, R, q_nom, v_nom;
};

// reference optimizer/trajectory_optimizer_solution.h:58-139
struct Stats {
  double solve_time = 0;
  std::vector<double> iteration_times, iteration_costs, linesearch_alphas, trust_region_radii, q_norms, dq_norms,
      dqH_norms, trust_ratios, gradient_norms, dL_dqs, h_norms, merits;
  std::vector<int> linesearch_iterations;
  void push_data(double t, double cost, int ls, double alpha, double delta, double qn, double dqn, double dqHn,
                 double rho, double gn, double dL, double hn, double merit) {
    iteration_times.push_back(t); iteration_costs.push_back(cost); linesearch_iterations.push_back(ls);
    linesearch_alphas.push_back(alpha); trust_region_radii.push_back(delta); q_norms.push_back(qn);
    dq_norms.push_back(dqn); dqH_norms.push_back(dqHn); trust_ratios.push_back(rho); gradient_norms.push_back(gn);
    dL_dqs.push_back(dL); h_norms.push_back(hn); merits.push_back(merit);
  }
  bool is_empty() const { return iteration_times.empty(); }
};

// q plus the lazily evaluated cache (reference optimizer/trajectory_optimizer_state.h:38-351)
struct State {
  Vec q;  // (N+1)*nq
  // cache
  bool traj_ok = false, nplus_ok = false, tau_ok = false, cost_ok = false, deriv_ok = false, grad_ok = false,
       hess_ok = false, scale_ok = false, shess_ok = false, sgrad_ok = false, h_ok = false, J_ok = false,
       lambda_ok = false, merit_ok = false, mgrad_ok = false;
  Vec Nplus;          // (N+1) blocks nv x nq
  Vec v, a, tau;      // (N+1)*nv, N*nv, N*nv
  double cost = 0, merit = 0;
  Vec dtau_dqm, dtau_dqt, dtau_dqp;  // N blocks nv x nq
  Vec dvt_dqt, dvt_dqm;              // (N+1) blocks nv x nq
  Vec gradient, scaled_gradient, merit_gradient, scale_factors, h, J, lambda;
  PentaMatrix hessian, scaled_hessian;
  void invalidate() {  // trajectory_optimizer_state.h:333-350
    traj_ok = nplus_ok = tau_ok = cost_ok = deriv_ok = grad_ok = hess_ok = scale_ok = shess_ok = sgrad_ok = h_ok =
        J_ok = lambda_ok = merit_ok = mgrad_ok = false;
  }
};

struct WarmStartData {  // reference optimizer/warm_start.h:23-76
  State state, scratch_state;
  double Delta = 0;
  Vec dq, dqH;
};

class Optimizer {
 public:
  Dynamics dyn;
  Problem prob;
  Params params;
  std::vector<int> unactuated_dofs;
  int nq = 0, nv = 0, N = 0;

  void Init() {
    nq = dyn.model.nq; nv = dyn.model.nv; N = prob.N;
    // TO.cc:63-72: unactuated DoFs = zero rows of the actuation matrix; no
    // actuators at all => assume fully actuated.
    unactuated_dofs.clear();
    int nact = 0;
    for (int i = 0; i < nv; ++i) nact += dyn.model.actuated[i] ? 1 : 0;
    if (nact > 0)
      for (int i = 0; i < nv; ++i)
        if (!dyn.model.actuated[i]) unactuated_dofs.push_back(i);
    if ((int)prob.q_nom.size() != (N + 1) * nq || (int)prob.v_nom.size() != (N + 1) * nv)
      throw std::runtime_error("q_nom / v_nom must have num_steps+1 entries");  // TO.cc:75-82
  }
  double time_step() const { return prob.dt; }
  int num_steps() const { return N; }
  int num_vars() const { return (N + 1) * nq; }
  int num_equality_constraints() const { return (int)unactuated_dofs.size() * N; }

  State CreateState() const {
    State s;
    s.q.assign((size_t)(N + 1) * nq, 0.0);
    s.scale_factors.assign(num_vars(), 1.0);  // trajectory_optimizer_state.h: D initialised to ones
    return s;
  }
  void set_q(State* s, const Vec& q) const { s->q = q; s->invalidate(); }
  void AddToQ(State* s, const Vec& dq) const {
    for (size_t i = 0; i < s->q.size(); ++i) s->q[i] += dq[i];
    s->invalidate();
  }
  static double norm(const Vec& x) {
    double s = 0;
    for (double xi : x) s += xi * xi;
    return std::sqrt(s);
  }

  // ---- trajectory data -------------------------------------------------
  // TO.cc:1633-1647
  const Vec& EvalNplus(State& s) const {
    if (!s.nplus_ok) {
      s.Nplus.resize((size_t)(N + 1) * nv * nq);
      for (int t = 0; t <= N; ++t) dyn.Nplus(&s.q[(size_t)t * nq], &s.Nplus[(size_t)t * nv * nq]);
      s.nplus_ok = true;
    }
    return s.Nplus;
  }
  // TO.cc:178-202
  void CalcTrajectoryData(State& s) const {
    const Vec& Np = EvalNplus(s);
    s.v.assign((size_t)(N + 1) * nv, 0.0);
    s.a.assign((size_t)N * nv, 0.0);
    for (int j = 0; j < nv; ++j) s.v[j] = prob.v_init[j];
    Vec dq(nq);
    for (int t = 1; t <= N; ++t) {
      for (int i = 0; i < nq; ++i) dq[i] = s.q[(size_t)t * nq + i] - s.q[(size_t)(t - 1) * nq + i];
      const double* Nt = &Np[(size_t)t * nv * nq];
      for (int r = 0; r < nv; ++r) {
        double acc = Nt[r] * dq[0];
        for (int c = 1; c < nq; ++c) acc += Nt[(size_t)c * nv + r] * dq[c];
        s.v[(size_t)t * nv + r] = acc / prob.dt;
      }
    }
    for (int t = 0; t < N; ++t)
      for (int r = 0; r < nv; ++r)
        s.a[(size_t)t * nv + r] = (s.v[(size_t)(t + 1) * nv + r] - s.v[(size_t)t * nv + r]) / prob.dt;
    s.traj_ok = true;
  }
  const Vec& EvalV(State& s) const { if (!s.traj_ok) CalcTrajectoryData(s); return s.v; }
  const Vec& EvalA(State& s) const { if (!s.traj_ok) CalcTrajectoryData(s); return s.a; }

  // TO.cc:204-226 ; tau[t] = ID(q[t+1], v[t+1], a[t])
  const Vec& EvalTau(State& s) const {
    if (!s.tau_ok) {
      EvalV(s);
      s.tau.assign((size_t)N * nv, 0.0);
#if defined(_OPENMP)
#pragma omp parallel for num_threads(params.num_threads)
#endif
      for (int t = 0; t < N; ++t)
        dyn.InverseDynamics(&s.q[(size_t)(t + 1) * nq], &s.v[(size_t)(t + 1) * nv], &s.a[(size_t)t * nv], true,
                            &s.tau[(size_t)t * nv]);
      s.tau_ok = true;
    }
    return s.tau;
  }

  static double QuadForm(const double* e, const double* W, int n) {  // e^T W e, W column-major
    double tot = 0;
    for (int c = 0; c < n; ++c) {
      double acc = 0;
      for (int r = 0; r < n; ++r) acc += e[r] * W[(size_t)c * n + r];
      tot += acc * e[c];
    }
    return tot;
  }
  // TO.cc:147-176
  double CalcCost(const Vec& q, const Vec& v, const Vec& tau) const {
    double cost = 0;
    Vec qe(nq), ve(nv);
    for (int t = 0; t < N; ++t) {
      for (int i = 0; i < nq; ++i) qe[i] = q[(size_t)t * nq + i] - prob.q_nom[(size_t)t * nq + i];
      for (int i = 0; i < nv; ++i) ve[i] = v[(size_t)t * nv + i] - prob.v_nom[(size_t)t * nv + i];
      cost += QuadForm(qe.data(), prob.Qq.data(), nq);
      cost += QuadForm(ve.data(), prob.Qv.data(), nv);
      cost += QuadForm(&tau[(size_t)t * nv], prob.R.data(), nv);
    }
    cost *= prob.dt;
    for (int i = 0; i < nq; ++i) qe[i] = q[(size_t)N * nq + i] - prob.q_nom[(size_t)N * nq + i];
    for (int i = 0; i < nv; ++i) ve[i] = v[(size_t)N * nv + i] - prob.v_nom[(size_t)N * nv + i];
    cost += QuadForm(qe.data(), prob.Qf_q.data(), nq);
    cost += QuadForm(ve.data(), prob.Qf_v.data(), nv);
    return cost;
  }
  double EvalCost(State& s) const {
    if (!s.cost_ok) {
      EvalV(s); EvalTau(s);
      s.cost = CalcCost(s.q, s.v, s.tau);
      s.cost_ok = true;
    }
    return s.cost;
  }

  // ---- derivatives -----------------------------------------------------
  // TO.cc:426-563
  void CalcInverseDynamicsPartialsFiniteDiff(State& s) const {
    const Vec& q = s.q;
    const Vec& v = EvalV(s);
    const Vec& a = EvalA(s);
    const Vec& tau = EvalTau(s);
    const Vec& Np = EvalNplus(s);
    const size_t bsz = (size_t)nv * nq;
    const double eps = std::sqrt(std::numeric_limits<double>::epsilon());
    const double dt = prob.dt;
#if defined(_OPENMP)
#pragma omp parallel for num_threads(params.num_threads)
#endif
    for (int t = 1; t <= N; ++t) {
      Vec q_eps(q.begin() + (size_t)t * nq, q.begin() + (size_t)(t + 1) * nq);
      Vec v_eps_t(nv), v_eps_tp(nv), a_eps_tm(nv), a_eps_t(nv), tau_eps(nv);
      const double* Nt = &Np[(size_t)t * bsz];
      const double* Ntp = (t < N) ? &Np[(size_t)(t + 1) * bsz] : nullptr;
      for (int i = 0; i < nq; ++i) {
        const double qi = q[(size_t)t * nq + i];
        double dq_i = eps * std::max(1.0, std::fabs(qi));      // :504
        const double temp = qi + dq_i;                          // :507
        dq_i = temp - qi;                                       // :508
        const double dv_i = dq_i / dt, da_i = dv_i / dt;        // :510-511
        q_eps[i] = qi + dq_i;                                   // :514
        for (int j = 0; j < nv; ++j) {
          const double nti = Nt[(size_t)i * nv + j];
          v_eps_t[j] = v[(size_t)t * nv + j] + dv_i * nti;               // :516
          a_eps_tm[j] = a[(size_t)(t - 1) * nv + j] + da_i * nti;        // :517
          if (t < N) {
            const double ntpi = Ntp[(size_t)i * nv + j];
            v_eps_tp[j] = v[(size_t)(t + 1) * nv + j] - dv_i * ntpi;     // :519
            a_eps_t[j] = a[(size_t)t * nv + j] - da_i * (ntpi + nti);    // :520
          }
        }
        // tau[t-1] = ID(q[t], v[t], a[t-1])  (:526-531)
        dyn.InverseDynamics(q_eps.data(), v_eps_t.data(), a_eps_tm.data(), true, tau_eps.data());
        for (int j = 0; j < nv; ++j)
          s.dtau_dqp[(size_t)(t - 1) * bsz + (size_t)i * nv + j] = (tau_eps[j] - tau[(size_t)(t - 1) * nv + j]) / dq_i;
        // tau[t] = ID(q[t+1], v[t+1], a[t])  (:533-540)
        if (t < N) {
          dyn.InverseDynamics(&q[(size_t)(t + 1) * nq], v_eps_tp.data(), a_eps_t.data(), true, tau_eps.data());
          for (int j = 0; j < nv; ++j)
            s.dtau_dqt[(size_t)t * bsz + (size_t)i * nv + j] = (tau_eps[j] - tau[(size_t)t * nv + j]) / dq_i;
        }
        q_eps[i] = qi;  // :543
      }
      // dtau[t+1]/dq[t] = M(q[t+2]) N+[t+1] / dt^2  (:556-561)
      if (t < N - 1) {
        Vec M((size_t)nv * nv);
        dyn.MassMatrix(&q[(size_t)(t + 2) * nq], M.data());
        const double sc = 1 / dt / dt;
        for (double& x : M) x = sc * x;
        double* out = &s.dtau_dqm[(size_t)(t + 1) * bsz];
        const double* Nn = &Np[(size_t)(t + 1) * bsz];
        for (int c = 0; c < nq; ++c)
          for (int r = 0; r < nv; ++r) {
            double acc = M[r] * Nn[(size_t)c * nv];
            for (int j = 1; j < nv; ++j) acc += M[(size_t)j * nv + r] * Nn[(size_t)c * nv + j];
            out[(size_t)c * nv + r] = acc;
          }
      }
    }
  }

  // TO.cc:565-885 (2nd / 4th order central differences; serial like the reference)
  void CalcInverseDynamicsPartialsCentralDiff(State& s) const {
    const bool fourth = (params.gradients_method == kCentralDifferences4);
    const Vec& q = s.q;
    const Vec& v = EvalV(s);
    const Vec& a = EvalA(s);
    const Vec& Np = EvalNplus(s);
    const size_t bsz = (size_t)nv * nq;
    const double eps = std::sqrt(std::numeric_limits<double>::epsilon());
    const double dt = prob.dt;
    Vec qe(nq), ve(nv), ae(nv), tp(nv), tm(nv), tpp(nv), tmm(nv);
    for (int t = 1; t <= N; ++t) {
      const double* Nt = &Np[(size_t)t * bsz];
      const double* Ntp = (t < N) ? &Np[(size_t)(t + 1) * bsz] : nullptr;
      for (int i = 0; i < nq; ++i) {
        const double qi = q[(size_t)t * nq + i];
        double dq = eps * std::max(1.0, std::fabs(qi));  // :709
        const double temp = qi + dq;
        dq = temp - qi;                                   // :712-713
        const double dv = dq / dt, da = dv / dt;
        // ID at q[t] +- m*dq  -> tau[t-1]      (:763-787)
        auto eval_tm = [&](double mult, Vec& out) {
          for (int k = 0; k < nq; ++k) qe[k] = q[(size_t)t * nq + k];
          qe[i] = qi + mult * dq;
          for (int j = 0; j < nv; ++j) {
            ve[j] = v[(size_t)t * nv + j] + (mult * dv) * Nt[(size_t)i * nv + j];
            ae[j] = a[(size_t)(t - 1) * nv + j] + (mult * da) * Nt[(size_t)i * nv + j];
          }
          dyn.InverseDynamics(qe.data(), ve.data(), ae.data(), true, out.data());
        };
        eval_tm(1.0, tp); eval_tm(-1.0, tm);
        if (fourth) { eval_tm(2.0, tpp); eval_tm(-2.0, tmm); }
        for (int j = 0; j < nv; ++j) {
          double d = fourth ? 2.0 / 3.0 * (tp[j] - tm[j]) / dq - 1.0 / 12.0 * (tpp[j] - tmm[j]) / dq
                            : 0.5 * (tp[j] - tm[j]) / dq;
          s.dtau_dqp[(size_t)(t - 1) * bsz + (size_t)i * nv + j] = d;
        }
        if (t < N) {  // tau[t] = ID(q[t+1], v[t+1]-+, a[t]-+)  (:788-814)
          auto eval_t = [&](double mult, Vec& out) {
            for (int j = 0; j < nv; ++j) {
              ve[j] = v[(size_t)(t + 1) * nv + j] - (mult * dv) * Ntp[(size_t)i * nv + j];
              ae[j] = a[(size_t)t * nv + j] - (mult * da) * (Ntp[(size_t)i * nv + j] + Nt[(size_t)i * nv + j]);
            }
            dyn.InverseDynamics(&q[(size_t)(t + 1) * nq], ve.data(), ae.data(), true, out.data());
          };
          eval_t(1.0, tp); eval_t(-1.0, tm);
          if (fourth) { eval_t(2.0, tpp); eval_t(-2.0, tmm); }
          for (int j = 0; j < nv; ++j) {
            double d = fourth ? 2.0 / 3.0 * (tp[j] - tm[j]) / dq - 1.0 / 12.0 * (tpp[j] - tmm[j]) / dq
                              : 0.5 * (tp[j] - tm[j]) / dq;
            s.dtau_dqt[(size_t)t * bsz + (size_t)i * nv + j] = d;
          }
        }
        if (t < N - 1) {  // tau[t+1] = ID(q[t+2], v[t+2], a[t+1]+-)  (:815-839)
          auto eval_tp = [&](double mult, Vec& out) {
            for (int j = 0; j < nv; ++j)
              ae[j] = a[(size_t)(t + 1) * nv + j] + (mult * da) * Ntp[(size_t)i * nv + j];
            dyn.InverseDynamics(&q[(size_t)(t + 2) * nq], &v[(size_t)(t + 2) * nv], ae.data(), true, out.data());
          };
          eval_tp(1.0, tp); eval_tp(-1.0, tm);
          if (fourth) { eval_tp(2.0, tpp); eval_tp(-2.0, tmm); }
          for (int j = 0; j < nv; ++j) {
            double d = fourth ? 2.0 / 3.0 * (tp[j] - tm[j]) / dq - 1.0 / 12.0 * (tpp[j] - tmm[j]) / dq
                              : 0.5 * (tp[j] - tm[j]) / dq;
            s.dtau_dqm[(size_t)(t + 1) * bsz + (size_t)i * nv + j] = d;
          }
        }
      }
    }
  }

  // TO.cc:1587-1604, 388-424, 962-973 ; container conventions
  // inverse_dynamics_partials.h:29-43 (dtau_dqm[0] = NaN, dtau_dqt[0] = 0, dtau_dqm[1] = 0)
  void CalcCacheDerivativesData(State& s) const {
    const size_t bsz = (size_t)nv * nq;
    s.dtau_dqm.assign((size_t)N * bsz, 0.0);
    s.dtau_dqt.assign((size_t)N * bsz, 0.0);
    s.dtau_dqp.assign((size_t)N * bsz, 0.0);
    for (size_t i = 0; i < bsz; ++i) s.dtau_dqm[i] = std::numeric_limits<double>::quiet_NaN();
    switch (params.gradients_method) {
      case kForwardDifferences: CalcInverseDynamicsPartialsFiniteDiff(s); break;
      case kCentralDifferences:
      case kCentralDifferences4: CalcInverseDynamicsPartialsCentralDiff(s); break;
      default: throw std::runtime_error("gradients method not supported by the oracle (autodiff needs Drake)");
    }
    // velocity partials, TO.cc:962-973 ; velocity_partials.h:21-27 (dvt_dqm[0] = NaN)
    const Vec& Np = EvalNplus(s);
    s.dvt_dqt.assign((size_t)(N + 1) * bsz, 0.0);
    s.dvt_dqm.assign((size_t)(N + 1) * bsz, 0.0);
    for (size_t i = 0; i < bsz; ++i) s.dvt_dqm[i] = std::numeric_limits<double>::quiet_NaN();
    const double idt = 1 / prob.dt, midt = -1 / prob.dt;
    for (int t = 0; t <= N; ++t)
      for (size_t i = 0; i < bsz; ++i) {
        s.dvt_dqt[(size_t)t * bsz + i] = idt * Np[(size_t)t * bsz + i];
        if (t > 0) s.dvt_dqm[(size_t)t * bsz + i] = midt * Np[(size_t)t * bsz + i];
      }
    s.deriv_ok = true;
  }
  void EvalDerivatives(State& s) const { if (!s.deriv_ok) CalcCacheDerivativesData(s); }

  // out[j] (+)= sum_r (sum_i e[i] W[i][r]) J[r][j] ; e: n, W: n x n, J: n x nq (column-major)
  void AccVecWMat(const double* e, const double* W, const double* J, int n, double* out, bool init) const {
    Vec tmp(n);
    for (int r = 0; r < n; ++r) {
      double acc = e[0] * W[(size_t)r * n];
      for (int i = 1; i < n; ++i) acc += e[i] * W[(size_t)r * n + i];
      tmp[r] = acc;
    }
    for (int j = 0; j < nq; ++j) {
      double acc = tmp[0] * J[(size_t)j * n];
      for (int r = 1; r < n; ++r) acc += tmp[r] * J[(size_t)j * n + r];
      out[j] = init ? acc : out[j] + acc;
    }
  }
  Vec Scaled(const Vec& W, double s1, double s2) const {  // (s1 * W) * s2
    Vec out(W.size());
    for (size_t i = 0; i < W.size(); ++i) out[i] = (s1 * W[i]) * s2;
    return out;
  }

  // TO.cc:1021-1081
  void CalcGradient(State& s, Vec* g_out) const {
    const double dt = prob.dt;
    const Vec& q = s.q;
    const Vec& v = EvalV(s);
    const Vec& tau = EvalTau(s);
    EvalDerivatives(s);
    const size_t bsz = (size_t)nv * nq;
    const Vec Qq = Scaled(prob.Qq, 2, dt), Qv = Scaled(prob.Qv, 2, dt), R = Scaled(prob.R, 2, dt),
              Qfq = Scaled(prob.Qf_q, 2, 1), Qfv = Scaled(prob.Qf_v, 2, 1);
    Vec& g = *g_out;
    g.assign((size_t)(N + 1) * nq, 0.0);
    Vec qe(nq), ve(nv), vep(nv);
    for (int t = 1; t < N; ++t) {
      double* gt = &g[(size_t)t * nq];
      for (int i = 0; i < nq; ++i) qe[i] = q[(size_t)t * nq + i] - prob.q_nom[(size_t)t * nq + i];
      for (int i = 0; i < nv; ++i) {
        ve[i] = v[(size_t)t * nv + i] - prob.v_nom[(size_t)t * nv + i];
        vep[i] = v[(size_t)(t + 1) * nv + i] - prob.v_nom[(size_t)(t + 1) * nv + i];
      }
      for (int j = 0; j < nq; ++j) {  // :1050
        double acc = qe[0] * Qq[(size_t)j * nq];
        for (int i = 1; i < nq; ++i) acc += qe[i] * Qq[(size_t)j * nq + i];
        gt[j] = acc;
      }
      AccVecWMat(ve.data(), Qv.data(), &s.dvt_dqt[(size_t)t * bsz], nv, gt, false);            // :1053
      AccVecWMat(vep.data(), (t == N - 1) ? Qfv.data() : Qv.data(), &s.dvt_dqm[(size_t)(t + 1) * bsz], nv, gt,
                 false);                                                                         // :1054-1061
      AccVecWMat(&tau[(size_t)(t - 1) * nv], R.data(), &s.dtau_dqp[(size_t)(t - 1) * bsz], nv, gt, false);  // :1064
      AccVecWMat(&tau[(size_t)t * nv], R.data(), &s.dtau_dqt[(size_t)t * bsz], nv, gt, false);              // :1065
      if (t != N - 1)
        AccVecWMat(&tau[(size_t)(t + 1) * nv], R.data(), &s.dtau_dqm[(size_t)(t + 1) * bsz], nv, gt, false);  // :1068
    }
    double* gT = &g[(size_t)N * nq];  // :1074-1080
    AccVecWMat(&tau[(size_t)(N - 1) * nv], R.data(), &s.dtau_dqp[(size_t)(N - 1) * bsz], nv, gT, true);
    for (int i = 0; i < nq; ++i) qe[i] = q[(size_t)N * nq + i] - prob.q_nom[(size_t)N * nq + i];
    for (int i = 0; i < nv; ++i) ve[i] = v[(size_t)N * nv + i] - prob.v_nom[(size_t)N * nv + i];
    for (int j = 0; j < nq; ++j) {
      double acc = qe[0] * Qfq[(size_t)j * nq];
      for (int i = 1; i < nq; ++i) acc += qe[i] * Qfq[(size_t)j * nq + i];
      gT[j] += acc;
    }
    AccVecWMat(ve.data(), Qfv.data(), &s.dvt_dqt[(size_t)N * bsz], nv, gT, false);
  }
  const Vec& EvalGradient(State& s) const {
    if (!s.grad_ok) { CalcGradient(s, &s.gradient); s.grad_ok = true; }
    return s.gradient;
  }

  // out (nq x nq, col-major) (+)= A^T W B ; A, B: nv x nq, W: nv x nv.
  // (A^T W) first, then times B, like Eigen's left-to-right evaluation.
  void AccATWB(const double* A, const double* W, const double* B, double* out, bool init) const {
    Vec AtW((size_t)nq * nv);  // nq x nv, column-major
    for (int c = 0; c < nv; ++c)
      for (int r = 0; r < nq; ++r) {
        double acc = A[(size_t)r * nv] * W[(size_t)c * nv];
        for (int l = 1; l < nv; ++l) acc += A[(size_t)r * nv + l] * W[(size_t)c * nv + l];
        AtW[(size_t)c * nq + r] = acc;
      }
    for (int c = 0; c < nq; ++c)
      for (int r = 0; r < nq; ++r) {
        double acc = AtW[r] * B[(size_t)c * nv];
        for (int l = 1; l < nv; ++l) acc += AtW[(size_t)l * nq + r] * B[(size_t)c * nv + l];
        out[(size_t)c * nq + r] = init ? acc : out[(size_t)c * nq + r] + acc;
      }
  }

  // TO.cc:1093-1165
  void CalcHessian(State& s, PentaMatrix* H) const {
    const double dt = prob.dt;
    EvalDerivatives(s);
    const size_t bsz = (size_t)nv * nq, qq = (size_t)nq * nq;
    const Vec Qq = Scaled(prob.Qq, 2, dt), Qv = Scaled(prob.Qv, 2, dt), R = Scaled(prob.R, 2, dt),
              Qfq = Scaled(prob.Qf_q, 2, 1), Qfv = Scaled(prob.Qf_v, 2, 1);
    H->Resize(N + 1, nq);
    const Vec &P = s.dtau_dqp, &T = s.dtau_dqt, &Mm = s.dtau_dqm, &V = s.dvt_dqt, &Wm = s.dvt_dqm;
    double* C0 = H->blk(H->C, 0);
    for (int i = 0; i < nq; ++i) C0[(size_t)i * nq + i] = 1.0;  // :1124
    for (int t = 1; t < N; ++t) {
      double* Ct = H->blk(H->C, t);
      for (size_t i = 0; i < qq; ++i) Ct[i] = Qq[i];                                         // :1128
      AccATWB(&V[(size_t)t * bsz], Qv.data(), &V[(size_t)t * bsz], Ct, false);               // :1129
      AccATWB(&P[(size_t)(t - 1) * bsz], R.data(), &P[(size_t)(t - 1) * bsz], Ct, false);    // :1130
      AccATWB(&T[(size_t)t * bsz], R.data(), &T[(size_t)t * bsz], Ct, false);                // :1131
      if (t < N - 1) {
        AccATWB(&Mm[(size_t)(t + 1) * bsz], R.data(), &Mm[(size_t)(t + 1) * bsz], Ct, false);  // :1133
        AccATWB(&Wm[(size_t)(t + 1) * bsz], Qv.data(), &Wm[(size_t)(t + 1) * bsz], Ct, false); // :1134
      } else {
        AccATWB(&Wm[(size_t)(t + 1) * bsz], Qfv.data(), &Wm[(size_t)(t + 1) * bsz], Ct, false);  // :1136
      }
      double* Bt = H->blk(H->B, t + 1);
      AccATWB(&P[(size_t)t * bsz], R.data(), &T[(size_t)t * bsz], Bt, true);                 // :1141
      if (t < N - 1) {
        AccATWB(&T[(size_t)(t + 1) * bsz], R.data(), &Mm[(size_t)(t + 1) * bsz], Bt, false);   // :1143
        AccATWB(&V[(size_t)(t + 1) * bsz], Qv.data(), &Wm[(size_t)(t + 1) * bsz], Bt, false);  // :1144
      } else {
        AccATWB(&V[(size_t)(t + 1) * bsz], Qfv.data(), &Wm[(size_t)(t + 1) * bsz], Bt, false); // :1146
      }
      if (t < N - 1)
        AccATWB(&P[(size_t)(t + 1) * bsz], R.data(), &Mm[(size_t)(t + 1) * bsz], H->blk(H->A, t + 2), true);  // :1152
    }
    double* CN = H->blk(H->C, N);  // :1157-1161
    for (size_t i = 0; i < qq; ++i) CN[i] = Qfq[i];
    AccATWB(&V[(size_t)N * bsz], Qfv.data(), &V[(size_t)N * bsz], CN, false);
    AccATWB(&P[(size_t)(N - 1) * bsz], R.data(), &P[(size_t)(N - 1) * bsz], CN, false);
    H->MakeSymmetric();  // :1164
  }
  const PentaMatrix& EvalHessian(State& s) const {
    if (!s.hess_ok) { CalcHessian(s, &s.hessian); s.hess_ok = true; }
    return s.hessian;
  }

  // ---- scaling & equality constraints ------------------------------------
  // TO.cc:1225-1255
  const Vec& EvalScaleFactors(State& s) const {
    if (!s.scale_ok) {
      const PentaMatrix& H = EvalHessian(s);
      Vec d(num_vars());
      H.ExtractDiagonal(d.data());
      Vec& D = s.scale_factors;
      if ((int)D.size() != num_vars()) D.assign(num_vars(), 1.0);
      for (int i = 0; i < num_vars(); ++i) {
        switch (params.scaling_method) {
          case kSqrt: D[i] = std::min(1.0, 1 / std::sqrt(d[i])); break;
          case kAdaptiveSqrt: D[i] = std::min(D[i], 1 / std::sqrt(d[i])); break;
          case kDoubleSqrt: D[i] = std::min(1.0, 1 / std::sqrt(std::sqrt(d[i]))); break;
          case kAdaptiveDoubleSqrt: D[i] = std::min(D[i], 1 / std::sqrt(std::sqrt(d[i]))); break;
        }
      }
      s.scale_ok = true;
    }
    return s.scale_factors;
  }
  // TO.cc:1181-1202
  const PentaMatrix& EvalScaledHessian(State& s) const {
    if (!params.scaling) return EvalHessian(s);
    if (!s.shess_ok) {
      s.scaled_hessian = EvalHessian(s);
      s.scaled_hessian.ScaleByDiagonal(EvalScaleFactors(s).data());
      s.shess_ok = true;
    }
    return s.scaled_hessian;
  }
  // TO.cc:1204-1223
  const Vec& EvalScaledGradient(State& s) const {
    if (!params.scaling) return EvalGradient(s);
    if (!s.sgrad_ok) {
      const Vec& g = EvalGradient(s);
      const Vec& D = EvalScaleFactors(s);
      s.scaled_gradient.resize(g.size());
      for (size_t i = 0; i < g.size(); ++i) s.scaled_gradient[i] = D[i] * g[i];
      s.sgrad_ok = true;
    }
    return s.scaled_gradient;
  }
  // TO.cc:1267-1290
  const Vec& EvalEqualityConstraintViolations(State& s) const {
    if (!s.h_ok) {
      const Vec& tau = EvalTau(s);
      const int nu = (int)unactuated_dofs.size();
      s.h.assign((size_t)nu * N, 0.0);
      for (int t = 0; t < N; ++t)
        for (int j = 0; j < nu; ++j) s.h[(size_t)t * nu + j] = tau[(size_t)t * nv + unactuated_dofs[j]];
      s.h_ok = true;
    }
    return s.h;
  }
  // TO.cc:1292-1345 ; J is num_eq x num_vars, column-major
  const Vec& EvalEqualityConstraintJacobian(State& s) const {
    if (!s.J_ok) {
      EvalDerivatives(s);
      const int nu = (int)unactuated_dofs.size();
      const int neq = nu * N, nvars = num_vars();
      const size_t bsz = (size_t)nv * nq;
      s.J.assign((size_t)neq * nvars, 0.0);
      for (int t = 0; t < N; ++t)
        for (int i = 0; i < nu; ++i) {
          const int row = t * nu + i, dof = unactuated_dofs[i];
          for (int c = 0; c < nq; ++c) {
            s.J[(size_t)((t + 1) * nq + c) * neq + row] = s.dtau_dqp[(size_t)t * bsz + (size_t)c * nv + dof];
            if (t > 0) s.J[(size_t)(t * nq + c) * neq + row] = s.dtau_dqt[(size_t)t * bsz + (size_t)c * nv + dof];
            if (t > 1)
              s.J[(size_t)((t - 1) * nq + c) * neq + row] = s.dtau_dqm[(size_t)t * bsz + (size_t)c * nv + dof];
          }
        }
      if (params.scaling) {  // J~ = J D (:1330-1333)
        const Vec& D = EvalScaleFactors(s);
        for (int c = 0; c < nvars; ++c)
          for (int r = 0; r < neq; ++r) s.J[(size_t)c * neq + r] *= D[c];
      }
      s.J_ok = true;
    }
    return s.J;
  }
  // TO.cc:1371-1396 ; lambda = (J H^-1 J^T)^-1 (h - J H^-1 g)
  const Vec& EvalLagrangeMultipliers(State& s) const {
    if (!s.lambda_ok) {
      const PentaMatrix& H = EvalScaledHessian(s);
      const Vec& g = EvalScaledGradient(s);
      const Vec& h = EvalEqualityConstraintViolations(s);
      const Vec& J = EvalEqualityConstraintJacobian(s);
      const int neq = num_equality_constraints(), nvars = num_vars();
      Vec HinvJT((size_t)nvars * neq);  // nvars x neq, column-major
      for (int c = 0; c < neq; ++c)
        for (int r = 0; r < nvars; ++r) HinvJT[(size_t)c * nvars + r] = J[(size_t)r * neq + c];
      PentaFactorization Hlu(H);
      for (int c = 0; c < neq; ++c) Hlu.SolveInPlace(&HinvJT[(size_t)c * nvars]);
      Vec S((size_t)neq * neq, 0.0), rhs(neq);
      for (int c = 0; c < neq; ++c)
        for (int r = 0; r < neq; ++r) {
          double acc = 0;
          for (int k = 0; k < nvars; ++k) acc += J[(size_t)k * neq + r] * HinvJT[(size_t)c * nvars + k];
          S[(size_t)c * neq + r] = acc;
        }
      for (int r = 0; r < neq; ++r) {
        double acc = 0;
        for (int k = 0; k < nvars; ++k) acc += HinvJT[(size_t)r * nvars + k] * g[k];
        rhs[r] = h[r] - acc;
      }
      DenseLdltSolve(S, neq, rhs.data());
      s.lambda = rhs;
      s.lambda_ok = true;
    }
    return s.lambda;
  }
  // TO.cc:1411-1433
  double EvalMeritFunction(State& s) const {
    if (!params.equality_constraints) return EvalCost(s);
    if (!s.merit_ok) {
      const double L = EvalCost(s);
      const Vec& h = EvalEqualityConstraintViolations(s);
      const Vec& lam = EvalLagrangeMultipliers(s);
      double d = 0;
      for (size_t i = 0; i < h.size(); ++i) d += h[i] * lam[i];
      s.merit = L + d;
      s.merit_ok = true;
    }
    return s.merit;
  }
  // TO.cc:1435-1456
  const Vec& EvalMeritFunctionGradient(State& s) const {
    if (!params.equality_constraints) return EvalScaledGradient(s);
    if (!s.mgrad_ok) {
      const Vec& g = EvalScaledGradient(s);
      const Vec& lam = EvalLagrangeMultipliers(s);
      const Vec& J = EvalEqualityConstraintJacobian(s);
      const int neq = num_equality_constraints(), nvars = num_vars();
      s.merit_gradient.resize(nvars);
      for (int c = 0; c < nvars; ++c) {
        double acc = 0;
        for (int r = 0; r < neq; ++r) acc += J[(size_t)c * neq + r] * lam[r];
        s.merit_gradient[c] = g[c] + acc;
      }
      s.mgrad_ok = true;
    }
    return s.merit_gradient;
  }

  // TO.cc:2691-2707
  void NormalizeQuaternions(State* s) const {
    for (int b = 0; b < dyn.model.nb; ++b)
      if (dyn.model.jtype[b] == IDTO_JOINT_FLOATING)
        for (int t = 0; t <= N; ++t) {
          double* qq = &s->q[(size_t)t * nq + dyn.model.qstart[b]];
          const double n = std::sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
          for (int k = 0; k < 4; ++k) qq[k] /= n;
        }
    s->invalidate();
  }

  // ---- linear solve, dogleg, trust ratio -----------------------------------
  // TO.cc:2077-2096
  void SolveLinearSystemInPlace(const PentaMatrix& H, Vec* b) const {
    if (params.linear_solver == kPentaDiagonalLu) {
      PentaFactorization Hlu(H);
      Hlu.SolveInPlace(b->data());
    } else {
      std::vector<double> Hd = H.MakeDense();
      DenseLdltSolve(Hd, H.size(), b->data());
    }
  }
  // TO.cc:2037-2066
  double SolveDoglegQuadratic(double a, double b, double c) const {
    if (!(a > 0)) throw std::runtime_error("dogleg: a <= 0");
    double s;
    if (a < std::numeric_limits<double>::epsilon()) {
      s = -c / b;
    } else {
      const double bt = b / a, ct = c / a;
      const double det = bt * bt - 4 * ct;
      if (!(det > 0)) throw std::runtime_error("dogleg: determinant <= 0");
      s = (-bt + std::sqrt(det)) / 2;
    }
    if (!(0 < s && s < 1)) throw std::runtime_error("dogleg: s not in (0,1)");
    return s;
  }
  static double Dot(const Vec& a, const Vec& b) {
    double s = 0;
    for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
    return s;
  }
  // TO.cc:2108-2202
  bool CalcDoglegPoint(State& s, double Delta, Vec* dq, Vec* dqH) const {
    const PentaMatrix& H = EvalScaledHessian(s);
    const Vec& g = EvalMeritFunctionGradient(s);
    const int n = num_vars();
    Vec Hg(n);
    H.MultiplyBy(g.data(), Hg.data());
    const double gHg = Dot(g, Hg);
    Vec pH(n);
    for (int i = 0; i < n; ++i) pH[i] = -g[i] / Delta;  // :2139
    SolveLinearSystemInPlace(H, &pH);
    dqH->resize(n);
    for (int i = 0; i < n; ++i) (*dqH)[i] = pH[i] * Delta;  // :2152
    Vec pU(n);
    const double coef = -(Dot(g, g) / gHg);
    for (int i = 0; i < n; ++i) pU[i] = coef * g[i] / Delta;  // :2157
    dq->resize(n);
    auto apply_scaling = [&]() {
      if (params.scaling) {
        const Vec& D = EvalScaleFactors(s);
        for (int i = 0; i < n; ++i) (*dq)[i] = D[i] * (*dq)[i];
      }
    };
    const double pUn = norm(pU);
    if (1.0 <= pUn) {  // :2160-2168
      for (int i = 0; i < n; ++i) (*dq)[i] = (Delta / pUn) * pU[i];
      apply_scaling();
      return true;
    }
    if (1.0 >= norm(pH)) {  // :2171-2178
      for (int i = 0; i < n; ++i) (*dq)[i] = pH[i] * Delta;
      apply_scaling();
      return false;
    }
    Vec d(n);
    for (int i = 0; i < n; ++i) d[i] = pH[i] - pU[i];
    const double a = Dot(d, d), b = 2 * Dot(pU, d), c = Dot(pU, pU) - 1.0;  // :2192-2194
    const double sq = SolveDoglegQuadratic(a, b, c);
    for (int i = 0; i < n; ++i) (*dq)[i] = (pU[i] + sq * d[i]) * Delta;  // :2197
    apply_scaling();
    return true;
  }
  // TO.cc:1979-2035
  double CalcTrustRatio(State& s, const Vec& dq, State* scratch) const {
    const double merit_k = EvalMeritFunction(s);
    const Vec& g_tilde_k = EvalMeritFunctionGradient(s);
    const PentaMatrix& H_k = EvalScaledHessian(s);
    set_q(scratch, s.q);
    AddToQ(scratch, dq);
    if (params.normalize_quaternions) NormalizeQuaternions(scratch);
    double merit_kp = EvalCost(*scratch);
    if (params.equality_constraints) {
      const Vec& lambda_k = EvalLagrangeMultipliers(s);
      const Vec& h_kp = EvalEqualityConstraintViolations(*scratch);
      merit_kp += Dot(h_kp, lambda_k);
    }
    const int n = num_vars();
    Vec dqs(n), Hdq(n);
    if (params.scaling) {
      const Vec& D = EvalScaleFactors(s);
      for (int i = 0; i < n; ++i) dqs[i] = (1.0 / D[i]) * dq[i];
    } else {
      dqs = dq;
    }
    H_k.MultiplyBy(dqs.data(), Hdq.data());
    const double hessian_term = 0.5 * Dot(dqs, Hdq);
    const double gradient_term = Dot(g_tilde_k, dqs);
    const double predicted = -gradient_term - hessian_term;
    const double actual = merit_k - merit_kp;
    const double eps = 10 * std::numeric_limits<double>::epsilon() / prob.dt / prob.dt;
    if (predicted < eps && actual < eps) return 0.5;
    return actual / predicted;
  }
  // TO.cc:2653-2689
  int VerifyConvergenceCriteria(State& s, double previous_cost, const Vec& dq) const {
    int reason = kNoConvergenceCriteriaSatisfied;
    const double cost = EvalCost(s);
    if (std::fabs(previous_cost - cost) < params.abs_cost_reduction + params.rel_cost_reduction * cost)
      reason |= kCostReductionCriterionSatisfied;
    const Vec& g = EvalMeritFunctionGradient(s);
    if (std::fabs(Dot(g, dq)) < params.abs_gradient_along_dq + params.rel_gradient_along_dq * cost)
      reason |= kGradientCriterionSatisfied;
    if (norm(dq) < params.abs_state_change + params.rel_state_change * norm(s.q)) reason |= kSateCriterionSatisfied;
    return reason;
  }

  WarmStartData CreateWarmStart(const Vec& q_guess) const {  // TO.cc:1353-1361
    WarmStartData w;
    w.state = CreateState();
    w.scratch_state = CreateState();
    set_q(&w.state, q_guess);
    w.Delta = params.Delta0;
    return w;
  }

  // TO.cc:2449-2651
  int SolveFromWarmStart(WarmStartData* ws, Vec* sol_q, Vec* sol_v, Vec* sol_tau, Stats* stats,
                         int* reason_out) const {
    using clock = std::chrono::high_resolution_clock;
    const auto start_time = clock::now();
    auto iter_start = clock::now();
    if (params.method != kTrustRegion) throw std::runtime_error("warm start requires the trust-region method");
    State& state = ws->state;
    State& scratch = ws->scratch_state;
    Vec& dq = ws->dq;
    Vec& dqH = ws->dqH;
    const double eta = 0.0;
    int k = 0;
    double& Delta = ws->Delta;
    double rho;
    bool active;
    double previous_cost = EvalCost(state);
    while (k < params.max_iterations) {
      active = CalcDoglegPoint(state, Delta, &dq, &dqH);  // :2497
      const Vec& g = EvalMeritFunctionGradient(state);
      const Vec& h = EvalEqualityConstraintViolations(state);
      const double cost = EvalCost(state);
      const double merit = EvalMeritFunction(state);
      const double q_norm = norm(state.q);
      double dL_dq;
      if (params.scaling) {
        const Vec& D = EvalScaleFactors(state);
        double acc = 0;
        for (size_t i = 0; i < dq.size(); ++i) acc += g[i] * ((1.0 / D[i]) * dq[i]);
        dL_dq = acc / cost;
      } else {
        dL_dq = Dot(g, dq) / cost;
      }
      rho = CalcTrustRatio(state, dq, &scratch);  // :2526
      if (!(dL_dq < std::numeric_limits<double>::epsilon()))
        throw std::runtime_error("step is not a descent direction (TO.cc:2531)");
      const double g_norm = norm(g), h_norm = norm(h), dq_norm = norm(dq), dqH_norm = norm(dqH);
      if (rho > eta) {  // :2550-2553
        AddToQ(&state, dq);
        if (params.normalize_quaternions) NormalizeQuaternions(&state);
      }
      const double iter_time = std::chrono::duration<double>(clock::now() - iter_start).count();
      iter_start = clock::now();
      if (params.verbose)
        std::printf("| %6d | %8.3g | %7.2g | %7.3g | %10.5g | %10.5g | %10.4g | %10.4g |\n", k, cost, Delta, rho,
                    iter_time, g_norm / cost, dL_dq, h_norm);
      stats->push_data(iter_time, cost, 0, std::numeric_limits<double>::quiet_NaN(), Delta, q_norm, dq_norm,
                       dqH_norm, rho, g_norm, dL_dq, h_norm, merit);  // :2586-2598
      int reason = kNoConvergenceCriteriaSatisfied;
      if (params.check_convergence && rho > eta) {
        reason = VerifyConvergenceCriteria(state, previous_cost, dq);
        previous_cost = EvalCost(state);
        if (reason_out) *reason_out = reason;
      }
      if (reason != kNoConvergenceCriteriaSatisfied) break;
      if (rho < 0.25) Delta *= 0.25;                                   // :2614-2617
      else if (rho > 0.75 && active) Delta = std::min(2 * Delta, params.Delta_max);  // :2618-2622
      ++k;
    }
    stats->solve_time = std::chrono::duration<double>(clock::now() - start_time).count();
    *sol_q = state.q;
    *sol_v = EvalV(state);
    *sol_tau = EvalTau(state);
    if (k == params.max_iterations) return kMaxIterationsReached;
    return kSuccess;
  }

  // TO.cc:1931-1977
  std::pair<double, int> ArmijoLinesearch(State& s, const Vec& dq, State* scratch) const {
    const double L = EvalCost(s);
    const Vec& g = EvalGradient(s);
    const double c = 1e-4, rho = 0.8;
    double alpha = 1.0 / rho;
    const double L_prime = Dot(g, dq);
    if (!(L_prime <= 0)) throw std::runtime_error("linesearch: not a descent direction");
    const double thr = 10 * std::numeric_limits<double>::epsilon() / prob.dt / prob.dt;
    if (std::fabs(L_prime) / std::fabs(L) <= thr) return {1.0, 0};
    int i = 0;
    double L_new;
    Vec step(dq.size());
    do {
      alpha *= rho;
      set_q(scratch, s.q);
      for (size_t j = 0; j < dq.size(); ++j) step[j] = alpha * dq[j];
      AddToQ(scratch, step);
      if (params.normalize_quaternions) NormalizeQuaternions(scratch);
      L_new = EvalCost(*scratch);
      ++i;
    } while ((L_new > L + c * alpha * L_prime) && (i < params.max_linesearch_iterations));
    return {alpha, i};
  }
  // TO.cc:1852-1929
  std::pair<double, int> BacktrackingLinesearch(State& s, const Vec& dq, State* scratch) const {
    double mu = 0.0;
    if (params.equality_constraints) mu = 1e3;
    auto l1 = [](const Vec& h) { double t = 0; for (double x : h) t += std::fabs(x); return t; };
    const Vec& h = EvalEqualityConstraintViolations(s);
    const double L = EvalCost(s) + mu * l1(h);
    const Vec& g = EvalGradient(s);
    const double c = 1e-4, rho = 0.8;
    double alpha = 1.0;
    const double L_prime = Dot(g, dq) - mu * l1(h);
    if (!(L_prime <= 0)) throw std::runtime_error("linesearch: not a descent direction");
    if (std::fabs(L_prime) / std::fabs(L) <= std::sqrt(std::numeric_limits<double>::epsilon())) return {1.0, 0};
    Vec step(dq.size());
    auto eval_at = [&](double al) {
      set_q(scratch, s.q);
      for (size_t j = 0; j < dq.size(); ++j) step[j] = al * dq[j];
      AddToQ(scratch, step);
      if (params.normalize_quaternions) NormalizeQuaternions(scratch);
      return EvalCost(*scratch) + mu * l1(EvalEqualityConstraintViolations(*scratch));
    };
    double L_old = eval_at(alpha), L_new = L_old;
    int i = 0;
    bool armijo_met = false;
    while (!(armijo_met && (L_new > L_old))) {
      L_old = L_new;
      alpha *= rho;
      L_new = eval_at(alpha);
      if (L_new <= L + c * alpha * L_prime) armijo_met = true;
      ++i;
    }
    return {alpha / rho, i};
  }
  // TO.cc:2244-2407
  int SolveWithLinesearch(const Vec& q_guess, Vec* sol_q, Vec* sol_v, Vec* sol_tau, Stats* stats) const {
    using clock = std::chrono::high_resolution_clock;
    const auto start_time = clock::now();
    State state = CreateState();
    set_q(&state, q_guess);
    State scratch = CreateState();
    Vec dq(num_vars());
    int k = 0;
    bool failed = false;
    do {
      const auto it0 = clock::now();
      const double cost = EvalCost(state);
      const Vec& h = EvalEqualityConstraintViolations(state);
      const Vec& g = EvalMeritFunctionGradient(state);
      const PentaMatrix& H = EvalHessian(state);
      for (size_t i = 0; i < dq.size(); ++i) dq[i] = -g[i];
      SolveLinearSystemInPlace(H, &dq);
      auto [alpha, ls_iters] = (params.linesearch_method == kArmijo) ? ArmijoLinesearch(state, dq, &scratch)
                                                                      : BacktrackingLinesearch(state, dq, &scratch);
      if (ls_iters >= params.max_linesearch_iterations) failed = true;
      Vec step(dq.size());
      for (size_t i = 0; i < dq.size(); ++i) step[i] = alpha * dq[i];
      const double trust_ratio = CalcTrustRatio(state, step, &scratch);
      const double g_norm = norm(g), h_norm = norm(h), dq_norm = norm(dq), dL_dq = Dot(g, dq) / cost;
      AddToQ(&state, step);
      if (params.normalize_quaternions) NormalizeQuaternions(&state);
      const double iter_time = std::chrono::duration<double>(clock::now() - it0).count();
      stats->push_data(iter_time, cost, ls_iters, alpha, std::numeric_limits<double>::quiet_NaN(), norm(state.q),
                       dq_norm, dq_norm, trust_ratio, g_norm, dL_dq, h_norm, cost);
      ++k;
    } while (k < params.max_iterations && !failed);
    stats->solve_time = std::chrono::duration<double>(clock::now() - start_time).count();
    *sol_q = state.q;
    *sol_v = EvalV(state);
    *sol_tau = EvalTau(state);
    return failed ? kLinesearchMaxIters : kSuccess;
  }

  // TO.cc:2213-2234
  int Solve(const Vec& q_guess, Vec* sol_q, Vec* sol_v, Vec* sol_tau, Stats* stats, int* reason_out) const {
    for (int i = 0; i < nq; ++i)
      if (q_guess[i] != prob.q_init[i]) throw std::runtime_error("q_guess[0] != q_init (TO.cc:2221)");
    if ((int)q_guess.size() != (N + 1) * nq) throw std::runtime_error("q_guess has the wrong size");
    if (!stats->is_empty()) throw std::runtime_error("stats must be empty (TO.cc:2225)");
    if (params.method == kLinesearch) return SolveWithLinesearch(q_guess, sol_q, sol_v, sol_tau, stats);
    WarmStartData ws = CreateWarmStart(q_guess);
    return SolveFromWarmStart(&ws, sol_q, sol_v, sol_tau, stats, reason_out);
  }
};

}  // namespace oracle
