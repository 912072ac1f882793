// oracle/idto_oracle.cc — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// C interface (for ctypes) over the CPU restatement in rigid_body.h / penta.h /
// traj_opt.h.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product (idto_amd/, include/)
// never does.  Build: `make -C oracle` (g++ -O3 -fopenmp -ffp-contract=off).
//
// Every function returns 0 on success, -1 on error (message via orc_last_error).
#include <cstring>
#include <string>

#include "traj_opt.h"

using namespace oracle;

namespace {
thread_local std::string g_err;

struct Handle {
  Optimizer opt;
};
struct WsHandle {
  WarmStartData ws;
};

void CopyParams(const idto_solver_params_t& p, Params* o) {
  o->check_convergence = p.check_convergence != 0;
  o->rel_cost_reduction = p.rel_cost_reduction; o->abs_cost_reduction = p.abs_cost_reduction;
  o->rel_gradient_along_dq = p.rel_gradient_along_dq; o->abs_gradient_along_dq = p.abs_gradient_along_dq;
  o->rel_state_change = p.rel_state_change; o->abs_state_change = p.abs_state_change;
  o->method = p.method; o->linesearch_method = p.linesearch_method;
  o->max_iterations = p.max_iterations; o->max_linesearch_iterations = p.max_linesearch_iterations;
  o->gradients_method = p.gradients_method; o->linear_solver = p.linear_solver;
  o->normalize_quaternions = p.normalize_quaternions != 0; o->verbose = p.verbose != 0;
  o->scaling = p.scaling != 0; o->scaling_method = p.scaling_method;
  o->equality_constraints = p.equality_constraints != 0;
  o->Delta0 = p.Delta0; o->Delta_max = p.Delta_max; o->num_threads = p.num_threads;
}

void CopyStats(const Stats& s, idto_stats_t* o) {
  if (!o) return;
  o->solve_time = s.solve_time;
  const int n = std::min<int>((int)s.iteration_times.size(), o->capacity);
  o->count = n;
  for (int i = 0; i < n; ++i) {
    if (o->iteration_times) o->iteration_times[i] = s.iteration_times[i];
    if (o->iteration_costs) o->iteration_costs[i] = s.iteration_costs[i];
    if (o->linesearch_iterations) o->linesearch_iterations[i] = s.linesearch_iterations[i];
    if (o->linesearch_alphas) o->linesearch_alphas[i] = s.linesearch_alphas[i];
    if (o->trust_region_radii) o->trust_region_radii[i] = s.trust_region_radii[i];
    if (o->q_norms) o->q_norms[i] = s.q_norms[i];
    if (o->dq_norms) o->dq_norms[i] = s.dq_norms[i];
    if (o->dqH_norms) o->dqH_norms[i] = s.dqH_norms[i];
    if (o->trust_ratios) o->trust_ratios[i] = s.trust_ratios[i];
    if (o->gradient_norms) o->gradient_norms[i] = s.gradient_norms[i];
    if (o->dL_dqs) o->dL_dqs[i] = s.dL_dqs[i];
    if (o->h_norms) o->h_norms[i] = s.h_norms[i];
    if (o->merits) o->merits[i] = s.merits[i];
  }
}

template <class F>
int Guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_create(const idto_model_t* model, const idto_problem_t* prob, const idto_contact_params_t* cp,
                 const idto_solver_params_t* sp) {
  Handle* h = new Handle();
  try {
    Optimizer& o = h->opt;
    o.dyn.model.FromC(*model);
    const int nq = model->nq, nv = model->nv, N = prob->num_steps;
    o.dyn.contact.k = cp->contact_stiffness; o.dyn.contact.vd = cp->dissipation_velocity;
    o.dyn.contact.vs = cp->stiction_velocity; o.dyn.contact.mu = cp->friction_coefficient;
    o.dyn.contact.sigma = cp->smoothing_factor;
    o.dyn.contact.Finalize();
    o.prob.N = N; o.prob.dt = prob->time_step;
    o.prob.q_init.assign(prob->q_init, prob->q_init + nq);
    o.prob.v_init.assign(prob->v_init, prob->v_init + nv);
    o.prob.Qq.assign(prob->Qq, prob->Qq + nq * nq);
    o.prob.Qv.assign(prob->Qv, prob->Qv + nv * nv);
    o.prob.Qf_q.assign(prob->Qf_q, prob->Qf_q + nq * nq);
    o.prob.Qf_v.assign(prob->Qf_v, prob->Qf_v + nv * nv);
    o.prob.R.assign(prob->R, prob->R + nv * nv);
    o.prob.q_nom.assign(prob->q_nom, prob->q_nom + (size_t)(N + 1) * nq);
    o.prob.v_nom.assign(prob->v_nom, prob->v_nom + (size_t)(N + 1) * nv);
    CopyParams(*sp, &o.params);
    o.Init();
  } catch (const std::exception& e) {
    g_err = e.what();
    delete h;
    return nullptr;
  }
  return h;
}
void orc_destroy(void* hv) { delete static_cast<Handle*>(hv); }

double orc_contact_threshold(void* hv) { return static_cast<Handle*>(hv)->opt.dyn.contact.threshold; }
int orc_num_equality_constraints(void* hv) { return static_cast<Handle*>(hv)->opt.num_equality_constraints(); }
int orc_num_unactuated(void* hv, int* dofs) {
  const auto& u = static_cast<Handle*>(hv)->opt.unactuated_dofs;
  if (dofs) for (size_t i = 0; i < u.size(); ++i) dofs[i] = u[i];
  return (int)u.size();
}

// reference TO.cc:2700ff `ResetInitialConditions` / `UpdateNominalTrajectory` (TO.h:429-470)
int orc_reset_initial_conditions(void* hv, const double* q_init, const double* v_init) {
  Optimizer& o = static_cast<Handle*>(hv)->opt;
  o.prob.q_init.assign(q_init, q_init + o.nq);
  o.prob.v_init.assign(v_init, v_init + o.nv);
  return 0;
}
int orc_update_nominal_trajectory(void* hv, const double* q_nom, const double* v_nom) {
  Optimizer& o = static_cast<Handle*>(hv)->opt;
  o.prob.q_nom.assign(q_nom, q_nom + (size_t)(o.N + 1) * o.nq);
  o.prob.v_nom.assign(v_nom, v_nom + (size_t)(o.N + 1) * o.nv);
  return 0;
}

// ---- single-configuration physics -----------------------------------------
int orc_inverse_dynamics(void* hv, const double* q, const double* v, const double* a, int full, double* tau) {
  return Guard([&] { static_cast<Handle*>(hv)->opt.dyn.InverseDynamics(q, v, a, full != 0, tau); });
}
int orc_mass_matrix(void* hv, const double* q, double* M) {
  return Guard([&] { static_cast<Handle*>(hv)->opt.dyn.MassMatrix(q, M); });
}
int orc_nplus(void* hv, const double* q, double* N) {
  return Guard([&] { static_cast<Handle*>(hv)->opt.dyn.Nplus(q, N); });
}
// body poses: X[nb*12] (R row-major, p)
int orc_body_poses(void* hv, const double* q, double* X) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    std::vector<double> z(o.nv, 0.0);
    std::vector<BodyKin> kin;
    o.dyn.Kinematics(q, z.data(), z.data(), &kin);
    for (int i = 0; i < o.dyn.model.nb; ++i) {
      std::memcpy(X + 12 * i, kin[i].R.m, 9 * sizeof(double));
      X[12 * i + 9] = kin[i].p.x; X[12 * i + 10] = kin[i].p.y; X[12 * i + 11] = kin[i].p.z;
    }
  });
}
// signed distance of every candidate pair at q: phi[npairs], n[3*npairs], Ca, Cb
int orc_signed_distances(void* hv, const double* q, double* phi, double* nrm, double* Ca, double* Cb) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    const Model& m = o.dyn.model;
    std::vector<double> z(o.nv, 0.0);
    std::vector<BodyKin> kin;
    o.dyn.Kinematics(q, z.data(), z.data(), &kin);
    for (int pi = 0; pi < m.npairs; ++pi) {
      const int ga = m.pair_a[pi], gb = m.pair_b[pi];
      const int ba = m.geom_body[ga], bb = m.geom_body[gb];
      const M3 RbA = ba < 0 ? Identity3() : kin[ba].R, RbB = bb < 0 ? Identity3() : kin[bb].R;
      const V3 pbA = ba < 0 ? V3{0, 0, 0} : kin[ba].p, pbB = bb < 0 ? V3{0, 0, 0} : kin[bb].p;
      const SignedDistanceResult sd =
          SignedDistance(m.geom_type[ga], RbA * m.geom_R[ga], pbA + RbA * m.geom_p[ga], m.geom_size[ga],
                         m.geom_type[gb], RbB * m.geom_R[gb], pbB + RbB * m.geom_p[gb], m.geom_size[gb]);
      phi[pi] = sd.phi;
      nrm[3 * pi] = sd.n.x; nrm[3 * pi + 1] = sd.n.y; nrm[3 * pi + 2] = sd.n.z;
      Ca[3 * pi] = sd.Ca.x; Ca[3 * pi + 1] = sd.Ca.y; Ca[3 * pi + 2] = sd.Ca.z;
      Cb[3 * pi] = sd.Cb.x; Cb[3 * pi + 1] = sd.Cb.y; Cb[3 * pi + 2] = sd.Cb.z;
    }
  });
}

// ---- whole-trajectory evaluations (fresh state each call) -------------------
static State MakeState(Optimizer& o, const double* q) {
  State s = o.CreateState();
  o.set_q(&s, Vec(q, q + (size_t)(o.N + 1) * o.nq));
  return s;
}
static void CopyOut(const Vec& v, double* out) {
  if (out) std::memcpy(out, v.data(), v.size() * sizeof(double));
}

int orc_eval_traj(void* hv, const double* q, double* v, double* a, double* tau, double* cost) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    State s = MakeState(o, q);
    CopyOut(o.EvalV(s), v); CopyOut(o.EvalA(s), a); CopyOut(o.EvalTau(s), tau);
    if (cost) *cost = o.EvalCost(s);
  });
}
int orc_calc_cost(void* hv, const double* q, const double* v, const double* tau, double* cost) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    *cost = o.CalcCost(Vec(q, q + (size_t)(o.N + 1) * o.nq), Vec(v, v + (size_t)(o.N + 1) * o.nv),
                       Vec(tau, tau + (size_t)o.N * o.nv));
  });
}
int orc_eval_partials(void* hv, const double* q, double* dtau_dqm, double* dtau_dqt, double* dtau_dqp,
                      double* dvt_dqt, double* dvt_dqm) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    State s = MakeState(o, q);
    o.EvalDerivatives(s);
    CopyOut(s.dtau_dqm, dtau_dqm); CopyOut(s.dtau_dqt, dtau_dqt); CopyOut(s.dtau_dqp, dtau_dqp);
    CopyOut(s.dvt_dqt, dvt_dqt); CopyOut(s.dvt_dqm, dvt_dqm);
  });
}
// g[(N+1)nq]; A, B, C, D, E [(N+1) nq nq]
int orc_grad_hess(void* hv, const double* q, double* g, double* A, double* B, double* C, double* D, double* E) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    State s = MakeState(o, q);
    CopyOut(o.EvalGradient(s), g);
    const PentaMatrix& H = o.EvalHessian(s);
    CopyOut(H.A, A); CopyOut(H.B, B); CopyOut(H.C, C); CopyOut(H.D, D); CopyOut(H.E, E);
  });
}
// One Gauss-Newton step "grad + Hessian + solve" as defined in SURVEY.md §8(d):
// tau -> partials -> g, H -> p = H^{-1} (-g)   (scaling/equality constraints off).
int orc_gn_step(void* hv, const double* q, double* g_out, double* p_out) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    State s = MakeState(o, q);
    const Vec& g = o.EvalGradient(s);
    const PentaMatrix& H = o.EvalHessian(s);
    Vec p(g.size());
    for (size_t i = 0; i < g.size(); ++i) p[i] = -g[i];
    PentaFactorization Hlu(H);
    Hlu.SolveInPlace(p.data());
    CopyOut(g, g_out); CopyOut(p, p_out);
  });
}
// Everything the trust-region iteration derives from q (for the invariants of
// TO_test.cc:1637-1751): any output pointer may be NULL.
int orc_eval_all(void* hv, const double* q, double* D, double* g_scaled, double* h, double* J, double* lambda,
                 double* merit, double* merit_grad, double* As, double* Bs, double* Cs) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    State s = MakeState(o, q);
    if (D) CopyOut(o.EvalScaleFactors(s), D);
    if (g_scaled) CopyOut(o.EvalScaledGradient(s), g_scaled);
    if (h) CopyOut(o.EvalEqualityConstraintViolations(s), h);
    if (J) CopyOut(o.EvalEqualityConstraintJacobian(s), J);
    if (lambda) CopyOut(o.EvalLagrangeMultipliers(s), lambda);
    if (merit) *merit = o.EvalMeritFunction(s);
    if (merit_grad) CopyOut(o.EvalMeritFunctionGradient(s), merit_grad);
    if (As || Bs || Cs) {
      const PentaMatrix& H = o.EvalScaledHessian(s);
      CopyOut(H.A, As); CopyOut(H.B, Bs); CopyOut(H.C, Cs);
    }
  });
}
int orc_dogleg(void* hv, const double* q, double Delta, double* dq, double* dqH, int* active) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    State s = MakeState(o, q);
    Vec d, dH;
    *active = o.CalcDoglegPoint(s, Delta, &d, &dH) ? 1 : 0;
    CopyOut(d, dq); CopyOut(dH, dqH);
  });
}
int orc_trust_ratio(void* hv, const double* q, const double* dq, double* rho) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    State s = MakeState(o, q);
    State scratch = o.CreateState();
    *rho = o.CalcTrustRatio(s, Vec(dq, dq + o.num_vars()), &scratch);
  });
}

// ---- solves -----------------------------------------------------------------
int orc_solve(void* hv, const double* q_guess, double* sol_q, double* sol_v, double* sol_tau, idto_stats_t* stats,
              int* flag, int* reason) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    Vec q, v, tau;
    Stats st;
    int rsn = 0;
    *flag = o.Solve(Vec(q_guess, q_guess + (size_t)(o.N + 1) * o.nq), &q, &v, &tau, &st, &rsn);
    if (reason) *reason = rsn;
    CopyOut(q, sol_q); CopyOut(v, sol_v); CopyOut(tau, sol_tau);
    CopyStats(st, stats);
  });
}
void* orc_ws_create(void* hv, const double* q_guess) {
  Optimizer& o = static_cast<Handle*>(hv)->opt;
  WsHandle* w = new WsHandle();
  w->ws = o.CreateWarmStart(Vec(q_guess, q_guess + (size_t)(o.N + 1) * o.nq));
  return w;
}
void orc_ws_destroy(void* wv) { delete static_cast<WsHandle*>(wv); }
int orc_ws_set_q(void* hv, void* wv, const double* q) {
  Optimizer& o = static_cast<Handle*>(hv)->opt;
  o.set_q(&static_cast<WsHandle*>(wv)->ws.state, Vec(q, q + (size_t)(o.N + 1) * o.nq));
  return 0;
}
int orc_ws_get(void* hv, void* wv, double* q, double* Delta) {
  WsHandle* w = static_cast<WsHandle*>(wv);
  (void)hv;
  CopyOut(w->ws.state.q, q);
  if (Delta) *Delta = w->ws.Delta;
  return 0;
}
int orc_ws_solve(void* hv, void* wv, double* sol_q, double* sol_v, double* sol_tau, idto_stats_t* stats, int* flag,
                 int* reason) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    WsHandle* w = static_cast<WsHandle*>(wv);
    Vec q, v, tau;
    Stats st;
    int rsn = 0;
    *flag = o.SolveFromWarmStart(&w->ws, &q, &v, &tau, &st, &rsn);
    if (reason) *reason = rsn;
    CopyOut(q, sol_q); CopyOut(v, sol_v); CopyOut(tau, sol_tau);
    CopyStats(st, stats);
  });
}

// Times `iters` Gauss-Newton steps (SURVEY.md §8d unit of work) on q with the
// handle's num_threads; returns seconds per step.  Used by bench.py's
// cpu_baseline leg only.
double orc_time_gn_steps(void* hv, const double* q, int iters) {
  Optimizer& o = static_cast<Handle*>(hv)->opt;
  std::vector<double> g(o.num_vars()), p(o.num_vars());
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) orc_gn_step(hv, q, g.data(), p.data());
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / iters;
}

// Where one Gauss-Newton step's time goes (bench.py's cpu_baseline only): seconds per step spent in
// [0] tau (OpenMP loop over t, TO.cc:209), [1] derivatives (the finite-difference loop, OpenMP over t,
// TO.cc:476, plus the serial N+ / velocity partials), [2] gradient + Hessian assembly (serial),
// [3] factor + solve (serial).  The reference parallelises [0] and [1] only: the rest is its Amdahl floor.
int orc_time_gn_parts(void* hv, const double* q, int iters, double* parts) {
  return Guard([&] {
    Optimizer& o = static_cast<Handle*>(hv)->opt;
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    for (int k = 0; k < 4; ++k) parts[k] = 0.0;
    for (int i = 0; i < iters; ++i) {
      State s = MakeState(o, q);
      const auto t0 = clk::now();
      o.EvalTau(s);
      const auto t1 = clk::now();
      o.EvalDerivatives(s);
      const auto t2 = clk::now();
      const Vec& g = o.EvalGradient(s);
      const PentaMatrix& H = o.EvalHessian(s);
      const auto t3 = clk::now();
      Vec p(g.size());
      for (size_t j = 0; j < g.size(); ++j) p[j] = -g[j];
      PentaFactorization Hlu(H);
      Hlu.SolveInPlace(p.data());
      const auto t4 = clk::now();
      parts[0] += secs(t0, t1); parts[1] += secs(t1, t2); parts[2] += secs(t2, t3); parts[3] += secs(t3, t4);
    }
    for (int k = 0; k < 4; ++k) parts[k] /= iters;
  });
}

// ---- block penta-diagonal algebra (penta_diagonal_solver_test.cc) ------------
static PentaMatrix MakePenta(int n, int bs, const double* A, const double* B, const double* C, const double* D,
                             const double* E) {
  PentaMatrix M(n, bs);
  const size_t sz = (size_t)n * bs * bs;
  M.A.assign(A, A + sz); M.B.assign(B, B + sz); M.C.assign(C, C + sz);
  if (D && E) { M.D.assign(D, D + sz); M.E.assign(E, E + sz); }
  else M.MakeSymmetric();
  return M;
}
// if D/E are NULL the matrix is made symmetric from (A, B, C) (constructor of
// penta_diagonal_matrix.cc:44-61); D_out/E_out return the bands actually used.
int orc_penta_make_symmetric(int n, int bs, const double* A, const double* B, double* C, double* D, double* E) {
  PentaMatrix M = MakePenta(n, bs, A, B, C, nullptr, nullptr);
  CopyOut(M.C, C); CopyOut(M.D, D); CopyOut(M.E, E);
  return 0;
}
int orc_penta_solve(int n, int bs, const double* A, const double* B, const double* C, const double* D,
                    const double* E, double* rhs, int nrhs) {
  return Guard([&] {
    PentaMatrix M = MakePenta(n, bs, A, B, C, D, E);
    PentaFactorization F(M);
    for (int c = 0; c < nrhs; ++c) F.SolveInPlace(rhs + (size_t)c * n * bs);
  });
}
int orc_penta_multiply(int n, int bs, const double* A, const double* B, const double* C, const double* D,
                       const double* E, const double* v, double* out) {
  PentaMatrix M = MakePenta(n, bs, A, B, C, D, E);
  M.MultiplyBy(v, out);
  return 0;
}
int orc_penta_make_dense(int n, int bs, const double* A, const double* B, const double* C, const double* D,
                         const double* E, double* dense) {
  PentaMatrix M = MakePenta(n, bs, A, B, C, D, E);
  CopyOut(M.MakeDense(), dense);
  return 0;
}
int orc_penta_extract_diagonal(int n, int bs, const double* C, double* d) {
  for (int i = 0; i < n; ++i)
    for (int r = 0; r < bs; ++r) d[(size_t)i * bs + r] = C[(size_t)i * bs * bs + (size_t)r * bs + r];
  return 0;
}
int orc_penta_scale_by_diagonal(int n, int bs, double* A, double* B, double* C, double* D, double* E,
                                const double* s) {
  PentaMatrix M = MakePenta(n, bs, A, B, C, D, E);
  M.ScaleByDiagonal(s);
  CopyOut(M.A, A); CopyOut(M.B, B); CopyOut(M.C, C); CopyOut(M.D, D); CopyOut(M.E, E);
  return 0;
}
int orc_dense_ldlt_solve(int n, double* M, double* b) {
  std::vector<double> Mv(M, M + (size_t)n * n);
  return DenseLdltSolve(Mv, n, b) ? 0 : -1;
}

// ---- deterministic math (tests/test_detmath.py) --------------------------------
void orc_det_sincos(const double* x, double* s, double* c, int n) {
  for (int i = 0; i < n; ++i) idto::detmath::sincos(x[i], &s[i], &c[i]);
}
void orc_det_exp(const double* x, double* y, int n) {
  for (int i = 0; i < n; ++i) y[i] = idto::detmath::exp(x[i]);
}
void orc_det_log(const double* x, double* y, int n) {
  for (int i = 0; i < n; ++i) y[i] = idto::detmath::log(x[i]);
}
int orc_uses_libm() {
#ifdef IDTO_ORACLE_LIBM
  return 1;
#else
  return 0;
#endif
}

}  // extern "C"
