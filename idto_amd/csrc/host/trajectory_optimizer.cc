// trajectory_optimizer.cc — host side of idto::optimizer::TrajectoryOptimizer<double>
// (include/idto/optimizer/trajectory_optimizer.h) written directly on the C-ABI of
// libidto_hip.so.  What runs where:
//   device (include/idto_hip.h): N+, v, a, tau, cost, dtau/dq, gradient, Hessian bands, every
//                                linear solve with H, the equality-constraint Schur complement
//                                J H^-1 J^T and H^-1 (g + J^T lambda);
//   host (this file):            the O(num_vars) bookkeeping of one trust-region / linesearch
//                                iteration and the small dense solve for the multipliers,
//                                following reference optimizer/trajectory_optimizer.cc
//                                ("TO.cc") function by function (cited at each one).
// Nothing here evaluates dynamics or factorises H: without the HIP library/device the
// constructor throws.
#include "idto/optimizer/trajectory_optimizer.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <string>

namespace idto {
namespace optimizer {

namespace {

using Vec = std::vector<double>;

// Optional wall-clock profile of the host loop (environment variable IDTO_OPT_PROFILE=<s>): time
// per named phase from the (s+1)-th SolveFromWarmStart of the process on (the first solves pay
// one-off warm-up), printed to stderr when the process exits.  tools/host_profile.py uses it.
struct Profile {
  static constexpr int kMax = 16;
  const char* name[kMax] = {};
  double sec[kMax] = {};
  long calls[kMax] = {};
  bool on = false;
  int skip = std::getenv("IDTO_OPT_PROFILE") ? std::atoi(std::getenv("IDTO_OPT_PROFILE")) : -1;
  void NewSolve() {
    if (skip >= 0 && skip-- == 0) on = true;
  }
  int Slot(const char* n) {
    for (int i = 0; i < kMax; ++i) {
      if (name[i] == n) return i;
      if (!name[i]) { name[i] = n; return i; }
    }
    return kMax - 1;
  }
  ~Profile() {
    if (!on) return;
    for (int i = 0; i < kMax && name[i]; ++i)
      std::fprintf(stderr, "[idto_opt profile] %-28s %10.3f ms in %6ld calls\n", name[i], 1e3 * sec[i], calls[i]);
  }
};
Profile g_profile;
struct Scope {
  int slot = -1;
  std::chrono::steady_clock::time_point t0;
  explicit Scope(const char* n) {
    if (g_profile.on) { slot = g_profile.Slot(n); t0 = std::chrono::steady_clock::now(); }
  }
  ~Scope() {
    if (slot >= 0) {
      g_profile.sec[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      ++g_profile.calls[slot];
    }
  }
};

double Dot(const Vec& a, const Vec& b) {
  double s = 0;
  for (std::size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}
double Norm(const Vec& a) { return std::sqrt(Dot(a, a)); }

Vec Flatten(const std::vector<Vec>& x) {
  Vec out;
  for (const Vec& xi : x) out.insert(out.end(), xi.begin(), xi.end());
  return out;
}
std::vector<Vec> Unflatten(const Vec& flat, int count, int width) {
  std::vector<Vec> out((std::size_t)count, Vec((std::size_t)width));
  for (int t = 0; t < count; ++t)
    std::copy(flat.begin() + (std::size_t)t * width, flat.begin() + (std::size_t)(t + 1) * width, out[t].begin());
  return out;
}
// the same into `out`, whose rows keep their storage when the sizes repeat (the solution object of an MPC re-plan)
void UnflattenInto(const Vec& flat, int count, int width, std::vector<Vec>* out) {
  out->resize((std::size_t)count);
  for (int t = 0; t < count; ++t)
    (*out)[t].assign(flat.begin() + (std::size_t)t * width, flat.begin() + (std::size_t)(t + 1) * width);
}
std::vector<MatrixXd> UnflattenBlocks(const Vec& flat, int count, int rows, int cols) {
  std::vector<MatrixXd> out((std::size_t)count, MatrixXd(rows, cols));
  for (int t = 0; t < count; ++t)
    std::copy(flat.begin() + (std::size_t)t * rows * cols, flat.begin() + (std::size_t)(t + 1) * rows * cols,
              out[t].data());
  return out;
}

}  // namespace

namespace internal {
// Solves the symmetric positive (semi-)definite n x n system S x = b in place with an LDL^T
// factorisation with diagonal pivoting (what Eigen's ldlt() does for the reference at
// TO.cc:1395).  S is column-major; only its lower triangle is read and it is overwritten by the
// factors.  Right-looking in panels of kPanel pivots: inside a panel only the pivot column and the
// running diagonal are brought up to date, the rank-kPanel update of the trailing matrix is
// applied once per panel (one pass over the matrix per panel instead of one per pivot: the
// unblocked version is bound by cache bandwidth at n = 240).
// (compiled for AVX2+FMA as well; the loader picks the variant the host CPU supports)
__attribute__((target_clones("arch=haswell", "default")))
void DenseLdltSolve(std::vector<double>* S_io, int n, double* b) {
  constexpr int kPanel = 16;
  std::vector<double>& S = *S_io;
  std::vector<int> perm((std::size_t)n);
  for (int i = 0; i < n; ++i) perm[i] = i;
  auto at = [&](int r, int c) -> double& { return S[(std::size_t)c * n + r]; };  // r >= c
  std::vector<double> diag((std::size_t)n), W((std::size_t)n * kPanel), t((std::size_t)n);
  for (int i = 0; i < n; ++i) diag[i] = at(i, i);
  for (int j0 = 0; j0 < n; j0 += kPanel) {
    const int j1 = std::min(n, j0 + kPanel);
    for (int j = j0; j < j1; ++j) {
      int p = j;
      for (int i = j + 1; i < n; ++i)
        if (std::fabs(diag[i]) > std::fabs(diag[p])) p = i;
      if (p != j) {  // symmetric interchange j <-> p: lower triangle, panel workspace, diagonal
        for (int i = 0; i < j; ++i) std::swap(at(j, i), at(p, i));
        for (int i = j + 1; i < p; ++i) std::swap(at(i, j), at(p, i));
        for (int i = p + 1; i < n; ++i) std::swap(at(i, j), at(i, p));
        std::swap(at(j, j), at(p, p));
        for (int q = 0; q < j - j0; ++q) std::swap(W[(std::size_t)q * n + j], W[(std::size_t)q * n + p]);
        std::swap(diag[j], diag[p]);
        std::swap(perm[j], perm[p]);
      }
      // column j of the matrix as updated by the panel's earlier pivots
      double* __restrict tj = t.data();
      const double* cj0 = &at(0, j);
      for (int i = j + 1; i < n; ++i) tj[i] = cj0[i];
      for (int q = 0; q < j - j0; ++q) {
        const double f = at(j, j0 + q);  // L[j][pivot q]
        if (f == 0.0) continue;
        const double* __restrict wq = &W[(std::size_t)q * n];
        for (int i = j + 1; i < n; ++i) tj[i] -= wq[i] * f;
      }
      const double d = diag[j];
      at(j, j) = d;
      double* __restrict cj = &at(0, j);
      double* __restrict wj = &W[(std::size_t)(j - j0) * n];
      double* __restrict dg = diag.data();
      if (d == 0.0) {  // exactly singular direction: drop it (Eigen does the same)
        for (int i = j + 1; i < n; ++i) { cj[i] = 0.0; wj[i] = 0.0; }
        continue;
      }
      for (int i = j + 1; i < n; ++i) {
        const double l = tj[i] / d;
        wj[i] = tj[i];  // = d * l
        cj[i] = l;
        dg[i] -= tj[i] * l;
      }
    }
    // rank-(j1 - j0) update of the trailing lower triangle
    const int np = j1 - j0;
    for (int c = j1; c < n; ++c) {
      double* __restrict cc = &at(0, c);
      int q = 0;
      for (; q + 3 < np; q += 4) {  // four pivots per pass over the column
        const double f0 = at(c, j0 + q), f1 = at(c, j0 + q + 1), f2 = at(c, j0 + q + 2), f3 = at(c, j0 + q + 3);
        const double* __restrict w0 = &W[(std::size_t)q * n];
        const double* __restrict w1 = w0 + n;
        const double* __restrict w2 = w1 + n;
        const double* __restrict w3 = w2 + n;
        for (int r = c; r < n; ++r) cc[r] -= (w0[r] * f0 + w1[r] * f1) + (w2[r] * f2 + w3[r] * f3);
      }
      for (; q < np; ++q) {
        const double f = at(c, j0 + q);
        const double* __restrict wq = &W[(std::size_t)q * n];
        for (int r = c; r < n; ++r) cc[r] -= wq[r] * f;
      }
    }
  }
  std::vector<double> y((std::size_t)n);
  for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
  for (int j = 0; j < n; ++j) {
    const double yj = y[j];
    const double* cj = &at(0, j);
    for (int i = j + 1; i < n; ++i) y[i] -= cj[i] * yj;
  }
  for (int j = 0; j < n; ++j) y[j] = (at(j, j) != 0.0) ? y[j] / at(j, j) : 0.0;
  for (int j = n - 1; j >= 0; --j) {
    const double* cj = &at(0, j);
    double acc0 = 0, acc1 = 0;
    int i = j + 1;
    for (; i + 1 < n; i += 2) { acc0 += cj[i] * y[i]; acc1 += cj[i + 1] * y[i + 1]; }
    if (i < n) acc0 += cj[i] * y[i];
    y[j] -= acc0 + acc1;
  }
  for (int i = 0; i < n; ++i) b[perm[i]] = y[i];
}

}  // namespace internal

namespace {
using internal::DenseLdltSolve;

// sum_k a[k] * b[k] with four independent partial sums (lets the compiler keep four FMA chains
// in flight; plain left-to-right summation is latency-bound)
inline double Dot4(const double* a, const double* b, int len) {
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int k = 0;
  for (; k + 3 < len; k += 4) {
    s0 += a[k] * b[k];
    s1 += a[k + 1] * b[k + 1];
    s2 += a[k + 2] * b[k + 2];
    s3 += a[k + 3] * b[k + 3];
  }
  for (; k < len; ++k) s0 += a[k] * b[k];
  return (s0 + s1) + (s2 + s3);
}

}  // namespace

using TO = TrajectoryOptimizer<double>;

namespace {
// H was not numerically positive definite (IDTO_HIP_FACTORIZATION_FAILED).  The reference aborts
// here (DRAKE_DEMAND(Hlu.status() == kSuccess), TO.cc:2084 / :2091); Solve() turns it into
// SolverFlag::kFactorizationFailed (trajectory_optimizer_solution.h:19), direct callers of the
// Eval* methods see the exception.
struct FactorizationFailedError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// IDTO_HIP_SOLVER_TIMEOUT: a multi-workgroup solver launch did not find its partner workgroups resident (a
// device shared with other kernels); the context has stepped down to a safer variant and the solve is repeated.
struct SolverTimeoutError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
}  // namespace

void TO::Check(int rc) const {
  if (rc == IDTO_HIP_FACTORIZATION_FAILED) throw FactorizationFailedError(std::string("idto_hip: ") + idto_hip_last_error());
  if (rc == IDTO_HIP_SOLVER_TIMEOUT) throw SolverTimeoutError(std::string("idto_hip: ") + idto_hip_last_error());
  if (rc != 0) throw std::runtime_error(std::string("idto_hip: ") + idto_hip_last_error());
}

TO::TrajectoryOptimizer(const idto_model_t& model, double time_step, const ProblemDefinition& prob,
                        const SolverParameters& params, int device)
    : time_step_(time_step), prob_(prob), params_(params) {
  // same run-time checks as the reference constructor / CalcInverseDynamicsPartials (TO.cc:37-74, 400-424)
  if (params_.gradients_method != kForwardDifferences && params_.gradients_method != kCentralDifferences &&
      params_.gradients_method != kCentralDifferences4)
    throw std::runtime_error("TrajectoryOptimizer (HIP): gradients_method must be a finite-difference method (kAutoDiff needs Drake)");
  if (params_.exact_hessian) throw std::runtime_error("TrajectoryOptimizer (HIP): exact_hessian needs autodiff");
  if (params_.save_contour_data || params_.save_lineplot_data || params_.linesearch_plot_every_iteration)
    throw std::runtime_error("TrajectoryOptimizer (HIP): the plotting dumps (save_contour_data, save_lineplot_data, "
                             "linesearch_plot_every_iteration: TO.cc:1650-1830) are not produced by this build");
  for (int b = 0; b < model.nbodies; ++b) {
    const int jt = model.jtype[b];
    if (jt == IDTO_JOINT_FLOATING) quaternion_starts_.push_back(nq_);
    nq_ += (jt == IDTO_JOINT_FLOATING) ? 7 : (jt == IDTO_JOINT_PLANAR ? 3 : 1);
    nv_ += (jt == IDTO_JOINT_FLOATING) ? 6 : (jt == IDTO_JOINT_PLANAR ? 3 : 1);
  }
  if ((int)prob_.q_init.size() != nq_ || (int)prob_.v_init.size() != nv_ || (int)prob_.q_nom.size() != num_steps() + 1 ||
      (int)prob_.v_nom.size() != num_steps() + 1)
    throw std::runtime_error("TrajectoryOptimizer: problem definition has the wrong sizes");
  // unactuated dofs: rows of B that are zero; a model without any actuator counts as fully
  // actuated (TO.cc:63-72)
  bool any = false;
  for (int j = 0; j < nv_; ++j) any |= model.actuated[j] != 0;
  if (any)
    for (int j = 0; j < nv_; ++j)
      if (!model.actuated[j]) unactuated_dofs_.push_back(j);

  const Vec qn = Flatten(prob_.q_nom), vn = Flatten(prob_.v_nom);
  idto_problem_t p = {};
  p.num_steps = num_steps(); p.time_step = time_step_;
  p.q_init = prob_.q_init.data(); p.v_init = prob_.v_init.data();
  p.Qq = prob_.Qq.data(); p.Qv = prob_.Qv.data(); p.Qf_q = prob_.Qf_q.data(); p.Qf_v = prob_.Qf_v.data();
  p.R = prob_.R.data(); p.q_nom = qn.data(); p.v_nom = vn.data();
  idto_contact_params_t c = {params_.contact_stiffness, params_.dissipation_velocity, params_.stiction_velocity,
                             params_.friction_coefficient, params_.smoothing_factor};
  Check(idto_hip_create(&model, &p, &c, device, &hip_));
  Check(idto_hip_set_option(hip_, "gradients_method", static_cast<int>(params_.gradients_method)));
  // (this class talks to the context through its stream-ordered entry points only: q and the problem's arrays go up from
  // pinned staging without a wait of their own - two of the five waits of an MPC re-plan)
  if (!std::getenv("IDTO_OPT_BLOCKING_UPLOADS")) Check(idto_hip_set_option(hip_, "async_uploads", 1));
}

TO::TrajectoryOptimizer(const idto_model_t& model, double time_step, const ProblemDefinition& prob,
                        const SolverParameters& params, const std::vector<int>& devices)
    : TrajectoryOptimizer(model, time_step, prob, params, devices.empty() ? 0 : devices[0]) {
  if (devices.empty()) return;
  const Vec qn = Flatten(prob_.q_nom), vn = Flatten(prob_.v_nom);
  idto_problem_t p = {};
  p.num_steps = num_steps(); p.time_step = time_step_;
  p.q_init = prob_.q_init.data(); p.v_init = prob_.v_init.data();
  p.Qq = prob_.Qq.data(); p.Qv = prob_.Qv.data(); p.Qf_q = prob_.Qf_q.data(); p.Qf_v = prob_.Qf_v.data();
  p.R = prob_.R.data(); p.q_nom = qn.data(); p.v_nom = vn.data();
  idto_contact_params_t c = {params_.contact_stiffness, params_.dissipation_velocity, params_.stiction_velocity,
                             params_.friction_coefficient, params_.smoothing_factor};
  shard_ctx_.push_back(hip_);
  for (std::size_t i = 1; i < devices.size(); ++i) {
    idto_hip_ctx* h = nullptr;
    Check(idto_hip_create(&model, &p, &c, devices[i], &h));
    shard_ctx_.push_back(h);
    Check(idto_hip_set_option(h, "gradients_method", static_cast<int>(params_.gradients_method)));
  }
  Check(idto_hip_comm_init_all(shard_ctx_.data(), (int)shard_ctx_.size()));
}

TO::~TrajectoryOptimizer() {
  for (std::size_t i = 1; i < shard_ctx_.size(); ++i) idto_hip_destroy(shard_ctx_[i]);
  idto_hip_destroy(hip_);
}

void TO::UploadProblem() const {
  problem_dirty_ = false;
  const Vec qn = Flatten(prob_.q_nom), vn = Flatten(prob_.v_nom);
  idto_problem_t p = {};
  p.num_steps = num_steps(); p.time_step = time_step_;
  p.q_init = prob_.q_init.data(); p.v_init = prob_.v_init.data();
  p.Qq = prob_.Qq.data(); p.Qv = prob_.Qv.data(); p.Qf_q = prob_.Qf_q.data(); p.Qf_v = prob_.Qf_v.data();
  p.R = prob_.R.data(); p.q_nom = qn.data(); p.v_nom = vn.data();
  Check(idto_hip_set_problem(hip_, &p));
  for (std::size_t i = 1; i < shard_ctx_.size(); ++i) Check(idto_hip_set_problem(shard_ctx_[i], &p));
  resident_ = nullptr;  // every cached device result depends on the problem data
  device_level_ = 0;
}

// TO.h:429-450
void TO::ResetInitialConditions(const VectorXd& q_init, const VectorXd& v_init) {
  if ((int)q_init.size() != nq_ || (int)v_init.size() != nv_) throw std::runtime_error("ResetInitialConditions: wrong size");
  prob_.q_init = q_init;
  prob_.v_init = v_init;
  problem_dirty_ = true;   // (uploaded with the next device call: dev())
  resident_ = nullptr;     // every cached device result depends on the problem data
  device_level_ = 0;
}
// TO.h:452-470
void TO::UpdateNominalTrajectory(const std::vector<VectorXd>& q_nom, const std::vector<VectorXd>& v_nom) {
  if ((int)q_nom.size() != num_steps() + 1 || (int)v_nom.size() != num_steps() + 1)
    throw std::runtime_error("UpdateNominalTrajectory: wrong size");
  prob_.q_nom = q_nom;
  prob_.v_nom = v_nom;
  problem_dirty_ = true;
  resident_ = nullptr;
  device_level_ = 0;
}

// ---- device residency: level 0 = q, 1 = + tau/cost, 2 = + partials, 3 = + gradient/Hessian
void TO::EnsureDevice(const TrajectoryOptimizerState<T>& state, int level) const {
  if (resident_ != &state || !state.cache_.uploaded) {
    const Vec q = Flatten(state.q());
    Check(idto_hip_set_q(dev(), q.data()));
    resident_ = &state;
    state.cache_.uploaded = true;
    device_level_ = 0;
  }
  if (level >= 1 && device_level_ < 1) { Check(idto_hip_eval_tau(dev())); device_level_ = 1; }
  if (level >= 2 && device_level_ < 2) {
    if (shard_ctx_.empty()) {
      Check(idto_hip_eval_partials(dev()));
    } else {
      // sharded over the devices: every device gets q, evaluates its k-range, one all-gather
      const Vec q = Flatten(state.q());
      for (std::size_t i = 1; i < shard_ctx_.size(); ++i) Check(idto_hip_set_q(shard_ctx_[i], q.data()));
      Check(idto_hip_eval_partials_multi(const_cast<idto_hip_ctx**>(shard_ctx_.data()), (int)shard_ctx_.size()));
    }
    device_level_ = 2;
  }
  if (level >= 3 && device_level_ < 3) { Check(idto_hip_grad_hess(dev())); device_level_ = 3; }
}

Vec TO::Fetch(int what) const {
  Vec out((std::size_t)idto_hip_array_size(dev(), what));
  Check(idto_hip_get(dev(), what, out.data()));
  return out;
}

// TO.cc:1463-1480 (v, a, tau, N+) + CalcCost :147-176.  Every idto_hip_get is a synchronising
// device-to-host copy (~20 us), so the cache is filled in few, large pieces: tau + cost here (all a
// trial point needs), v / a / N+ only when somebody asks, the whole slab (dtau/dq + tau) and the
// three Hessian bands in one copy each.
void TO::CalcTrajectoryData(const TrajectoryOptimizerState<T>& state) const {
  auto& c = state.cache_;
  if (c.traj) return;
  Scope prof_("tau + cost (device + fetch)");
  if (resident_ != &state || !c.uploaded) {
    // a new point: upload, evaluate and read back in one call / one synchronisation
    const Vec q = Flatten(state.q());
    Vec tau((std::size_t)num_steps() * nv_);
    Check(idto_hip_trial_cost(dev(), q.data(), tau.data(), &c.cost));
    c.tau = Unflatten(tau, num_steps(), nv_);
    resident_ = &state;
    c.uploaded = true;
    device_level_ = 1;
  } else {
    EnsureDevice(state, 1);
    c.tau = Unflatten(Fetch(IDTO_ARR_TAU), num_steps(), nv_);
    c.cost = Fetch(IDTO_ARR_COST)[0];
  }
  c.traj = true;
}
void TO::CalcKinematics(const TrajectoryOptimizerState<T>& state) const {
  auto& c = state.cache_;
  if (c.kin) return;
  CalcTrajectoryData(state);
  EnsureDevice(state, 1);
  const int N = num_steps();
  c.v = Unflatten(Fetch(IDTO_ARR_V), N + 1, nv_);
  c.a = Unflatten(Fetch(IDTO_ARR_A), N, nv_);
  c.nplus = UnflattenBlocks(Fetch(IDTO_ARR_NPLUS), N + 1, nv_, nq_);
  c.kin = true;
}

// TO.cc:400-563 and :962-973
void TO::CalcDerivatives(const TrajectoryOptimizerState<T>& state) const {
  auto& c = state.cache_;
  if (c.deriv) return;
  CalcTrajectoryData(state);
  Scope prof_("partials (device + fetch)");
  EnsureDevice(state, 2);
  const int N = num_steps();
  const Vec slab = Fetch(IDTO_ARR_SLAB);  // per k: [dtau_dqm | dtau_dqt | dtau_dqp | tau_k]
  const int stride = idto_hip_slab_stride(dev()), bsz = nv_ * nq_;
  c.id_partials.dtau_dqm.assign((std::size_t)N, MatrixXd(nv_, nq_));
  c.id_partials.dtau_dqt.assign((std::size_t)N, MatrixXd(nv_, nq_));
  c.id_partials.dtau_dqp.assign((std::size_t)N, MatrixXd(nv_, nq_));
  for (int k = 0; k < N; ++k) {
    const double* rec = slab.data() + (std::size_t)k * stride;
    std::copy(rec, rec + bsz, c.id_partials.dtau_dqm[k].data());
    std::copy(rec + bsz, rec + 2 * bsz, c.id_partials.dtau_dqt[k].data());
    std::copy(rec + 2 * bsz, rec + 3 * bsz, c.id_partials.dtau_dqp[k].data());
  }
  c.deriv = true;
}
void TO::CalcVelocityPartials(const TrajectoryOptimizerState<T>& state) const {
  auto& c = state.cache_;
  if (c.vpart) return;
  CalcKinematics(state);
  const int N = num_steps();
  c.v_partials.dvt_dqt.assign((std::size_t)N + 1, MatrixXd(nv_, nq_));
  c.v_partials.dvt_dqm.assign((std::size_t)N + 1, MatrixXd(nv_, nq_));
  for (int t = 0; t <= N; ++t)
    for (int col = 0; col < nq_; ++col)
      for (int r = 0; r < nv_; ++r) {
        c.v_partials.dvt_dqt[t](r, col) = c.nplus[t](r, col) / time_step_;
        if (t > 0) c.v_partials.dvt_dqm[t](r, col) = -c.nplus[t](r, col) / time_step_;
      }
  c.vpart = true;
}

// TO.cc:1021-1081 and :1093-1165
void TO::CalcGradHess(const TrajectoryOptimizerState<T>& state) const {
  auto& c = state.cache_;
  if (c.grad && c.hess) return;
  CalcTrajectoryData(state);  // (the partials stay on the device unless somebody asks for them)
  Scope prof_("grad + hess (device + fetch)");
  EnsureDevice(state, 3);
  // g and the bands are copied on a side stream while the solver - which every iteration needs
  // next (CalcDoglegPoint :2139-2149 / CalcLagrangeMultipliers :1371-1396) - already runs
  Check(idto_hip_prefetch(dev(), IDTO_ARR_GRADIENT));
  Check(idto_hip_prefetch(dev(), IDTO_ARR_HBANDS));
  if (params_.equality_constraints && num_equality_constraints() > 0) {
    Scope prof2_("  launch constraint solve");
    Check(idto_hip_constraint_schur_begin(dev(), unactuated_dofs_.data(), (int)unactuated_dofs_.size()));
  } else if (!DenseLinearSolver()) {
    Check(idto_hip_factor_solve(dev(), nullptr, 1, nullptr));  // IDTO_ARR_STEP = -H^-1 g
    c.step_on_device = true;
  }
  {
    Scope prof2_("  fetch g");
    c.gradient = Fetch(IDTO_ARR_GRADIENT);
  }
  Vec bands;
  {
    Scope prof2_("  fetch H bands");
    bands = Fetch(IDTO_ARR_HBANDS);
  }  // [A | B | C], (N+6) blocks each
  const std::size_t used = (std::size_t)(num_steps() + 1) * nq_ * nq_, span = (std::size_t)(num_steps() + 6) * nq_ * nq_;
  c.hessian = PentaDiagonalMatrix<T>(num_steps() + 1, nq_);
  c.hessian.mutable_A().assign(bands.begin(), bands.begin() + used);
  c.hessian.mutable_B().assign(bands.begin() + span, bands.begin() + span + used);
  c.hessian.mutable_C().assign(bands.begin() + 2 * span, bands.begin() + 2 * span + used);
  c.hessian.MakeSymmetric();   // TO.cc:1165 (the device's C already carries both triangles; D, E are the mirrors)
  c.grad = c.hess = true;
}

const std::vector<VectorXd>& TO::EvalV(const TrajectoryOptimizerState<T>& s) const { CalcKinematics(s); return s.cache_.v; }
const std::vector<VectorXd>& TO::EvalA(const TrajectoryOptimizerState<T>& s) const { CalcKinematics(s); return s.cache_.a; }
const std::vector<VectorXd>& TO::EvalTau(const TrajectoryOptimizerState<T>& s) const { CalcTrajectoryData(s); return s.cache_.tau; }
const std::vector<MatrixXd>& TO::EvalNplus(const TrajectoryOptimizerState<T>& s) const { CalcKinematics(s); return s.cache_.nplus; }
double TO::EvalCost(const TrajectoryOptimizerState<T>& s) const { CalcTrajectoryData(s); return s.cache_.cost; }
const VelocityPartials<double>& TO::EvalVelocityPartials(const TrajectoryOptimizerState<T>& s) const {
  CalcVelocityPartials(s);
  return s.cache_.v_partials;
}
const InverseDynamicsPartials<double>& TO::EvalInverseDynamicsPartials(const TrajectoryOptimizerState<T>& s) const {
  CalcDerivatives(s);
  return s.cache_.id_partials;
}
const VectorXd& TO::EvalGradient(const TrajectoryOptimizerState<T>& s) const { CalcGradHess(s); return s.cache_.gradient; }
const PentaDiagonalMatrix<double>& TO::EvalHessian(const TrajectoryOptimizerState<T>& s) const {
  CalcGradHess(s);
  return s.cache_.hessian;
}

// TO.cc:1225-1255
const VectorXd& TO::EvalScaleFactors(const TrajectoryOptimizerState<T>& s) const {
  auto& c = s.cache_;
  if (!c.scale) {
    Scope prof_("scale factors");
    Vec d;
    EvalHessian(s).ExtractDiagonal(&d);
    Vec& D = c.scale_factors;
    if ((int)D.size() != num_vars()) D.assign((std::size_t)num_vars(), 1.0);
    for (int i = 0; i < num_vars(); ++i) {
      switch (params_.scaling_method) {
        case kSqrt: D[i] = std::min(1.0, 1 / std::sqrt(d[i])); break;
        case kAdaptiveSqrt: D[i] = std::min(D[i], 1 / std::sqrt(d[i])); break;
        case kDoubleSqrt: D[i] = std::min(1.0, 1 / std::sqrt(std::sqrt(d[i]))); break;
        case kAdaptiveDoubleSqrt: D[i] = std::min(D[i], 1 / std::sqrt(std::sqrt(d[i]))); break;
      }
    }
    c.scale = true;
  }
  return c.scale_factors;
}
// TO.cc:1181-1202
const PentaDiagonalMatrix<double>& TO::EvalScaledHessian(const TrajectoryOptimizerState<T>& s) const {
  if (!params_.scaling) return EvalHessian(s);
  auto& c = s.cache_;
  if (!c.shess) {
    Scope prof_("scaled hessian");
    c.scaled_hessian = EvalHessian(s);
    c.scaled_hessian.ScaleByDiagonal(EvalScaleFactors(s));
    c.shess = true;
  }
  return c.scaled_hessian;
}
// TO.cc:1204-1223
const VectorXd& TO::EvalScaledGradient(const TrajectoryOptimizerState<T>& s) const {
  if (!params_.scaling) return EvalGradient(s);
  auto& c = s.cache_;
  if (!c.sgrad) {
    const Vec& g = EvalGradient(s);
    const Vec& D = EvalScaleFactors(s);
    c.scaled_gradient.resize(g.size());
    for (std::size_t i = 0; i < g.size(); ++i) c.scaled_gradient[i] = D[i] * g[i];
    c.sgrad = true;
  }
  return c.scaled_gradient;
}
// TO.cc:1267-1290
const VectorXd& TO::EvalEqualityConstraintViolations(const TrajectoryOptimizerState<T>& s) const {
  auto& c = s.cache_;
  if (!c.h) {
    const auto& tau = EvalTau(s);
    const int nu = (int)unactuated_dofs_.size();
    c.h_viol.assign((std::size_t)nu * num_steps(), 0.0);
    for (int t = 0; t < num_steps(); ++t)
      for (int j = 0; j < nu; ++j) c.h_viol[(std::size_t)t * nu + j] = tau[t][unactuated_dofs_[j]];
    c.h = true;
  }
  return c.h_viol;
}
// TO.cc:1292-1345
const MatrixXd& TO::EvalEqualityConstraintJacobian(const TrajectoryOptimizerState<T>& s) const {
  auto& c = s.cache_;
  if (!c.J) {
    const auto& P = EvalInverseDynamicsPartials(s);
    Scope prof_("constraint jacobian");
    const int nu = (int)unactuated_dofs_.size(), N = num_steps();
    c.J_unscaled = MatrixXd(nu * N, num_vars());
    for (int t = 0; t < N; ++t)
      for (int i = 0; i < nu; ++i) {
        const int row = t * nu + i, dof = unactuated_dofs_[i];
        for (int col = 0; col < nq_; ++col) {
          c.J_unscaled(row, (t + 1) * nq_ + col) = P.dtau_dqp[t](dof, col);
          if (t > 0) c.J_unscaled(row, t * nq_ + col) = P.dtau_dqt[t](dof, col);
          if (t > 1) c.J_unscaled(row, (t - 1) * nq_ + col) = P.dtau_dqm[t](dof, col);
        }
      }
    c.J_scaled = c.J_unscaled;
    if (params_.scaling) {  // J~ = J D (:1330-1333)
      const Vec& D = EvalScaleFactors(s);
      for (int col = 0; col < num_vars(); ++col)
        for (int r = 0; r < nu * N; ++r) c.J_scaled(r, col) *= D[col];
    }
    c.J = true;
  }
  return c.J_scaled;
}

// TO.cc:2077-2096: the block Thomas algorithm (default) or a dense LDL^T of MakeDense(), both on the device, for
// the Hessian of `s` (unscaled: see EvalHinvMeritGradient).
bool TO::DenseLinearSolver() const { return params_.linear_solver == SolverParameters::LinearSolverType::kDenseLdlt; }
void TO::SolveLinearSystem(const TrajectoryOptimizerState<T>& s, const VectorXd& b, VectorXd* x, int solver) const {
  EvalGradient(s);
  EnsureDevice(s, 3);
  x->resize(b.size());
  const bool dense = solver < 0 ? DenseLinearSolver() : solver == 0;
  if (dense) Check(idto_hip_solve_dense_ldlt(dev(), b.data(), x->data()));
  else Check(idto_hip_solve_host(dev(), b.data(), 1, x->data()));
}

// print_debug_data's "condition number" (TO.cc:2351, :2503-2504): 1 / H.MakeDense().ldlt().rcond().  Eigen's rcond()
// is ||H||_1 times Hager's estimate of ||H^-1||_1 with Higham's alternating-sign safeguard (Eigen/src/Core/
// ConditionEstimator.h, rcond_invmatrix_L1_norm_estimate: at most 2 + 2 * 4 solves); restated here with the solves on the
// device (dense LDL^T, as the reference's).  scaled: H~ = D H D, whose inverse is applied as D^-1 H^-1 D^-1.
double TO::DebugConditionNumber(const TrajectoryOptimizerState<T>& s, bool scaled) const {
  const int n = num_vars();
  const PentaDiagonalMatrix<T>& H = scaled ? EvalScaledHessian(s) : EvalHessian(s);
  const Vec* D = (scaled && params_.scaling) ? &EvalScaleFactors(s) : nullptr;
  // the matrix 1-norm, the largest absolute column sum (= row sum: H is symmetric)
  double norm1 = 0;
  {
    const int nb = H.block_rows(), bs = H.block_size();
    for (int i = 0; i < nb; ++i)
      for (int r = 0; r < bs; ++r) {
        double sum = 0;
        auto add = [&](const double* M) { for (int cc = 0; cc < bs; ++cc) sum += std::fabs(M[(std::size_t)cc * bs + r]); };
        if (i >= 2) add(H.block(H.A(), i));
        if (i >= 1) add(H.block(H.B(), i));
        add(H.block(H.C(), i));
        if (i + 1 < nb) add(H.block(H.D(), i));
        if (i + 2 < nb) add(H.block(H.E(), i));
        norm1 = std::max(norm1, sum);
      }
  }
  if (norm1 == 0) return std::numeric_limits<double>::infinity();
  auto solve = [&](Vec v) {
    if (D) for (int i = 0; i < n; ++i) v[i] /= (*D)[i];
    Vec x;
    SolveLinearSystem(s, v, &x, /*dense*/ 0);
    if (D) for (int i = 0; i < n; ++i) x[i] /= (*D)[i];
    return x;
  };
  auto l1 = [](const Vec& v) { double a = 0; for (double x : v) a += std::fabs(x); return a; };
  Vec v = solve(Vec((std::size_t)n, 1.0 / n));
  double lower = l1(v);
  if (n > 1) {
    Vec sign((std::size_t)n), old_sign;
    int imax = -1, old_imax = -1;
    for (int k = 0; k < 4; ++k) {
      for (int i = 0; i < n; ++i) sign[i] = v[i] >= 0 ? 1.0 : -1.0;
      if (k > 0 && sign == old_sign) break;
      v = solve(sign);   // (H is symmetric: the adjoint solve is the solve)
      imax = 0;
      for (int i = 1; i < n; ++i) if (std::fabs(v[i]) > std::fabs(v[imax])) imax = i;
      if (imax == old_imax) break;
      Vec e((std::size_t)n, 0.0);
      e[imax] = 1.0;
      v = solve(e);
      const double old_lower = lower;
      lower = l1(v);
      if (lower <= old_lower) break;
      old_sign = sign;
      old_imax = imax;
    }
    double alt = 1.0;
    for (int i = 0; i < n; ++i) { v[i] = alt * (1.0 + double(i) / double(n - 1)); alt = -alt; }
    v = solve(v);
    lower = std::max(lower, 2 * l1(v) / (3.0 * n));
  }
  return norm1 * lower;
}

// H^-1 g_merit with the unscaled H, where g_merit = g + J^T lambda (g when there are no equality
// constraints).  Everything the iteration needs from a factorisation of the scaled H~ = D H D
// follows from it:  H~^-1 g~_merit = D^-1 H^-1 g_merit.
const VectorXd& TO::EvalHinvMeritGradient(const TrajectoryOptimizerState<T>& s) const {
  auto& c = s.cache_;
  if (params_.equality_constraints && num_equality_constraints() > 0) {
    EvalLagrangeMultipliers(s);
    if (DenseLinearSolver() && !c.hinv) {
      // the multipliers come from the block Thomas factorisation whatever `linear_solver` says (TO.cc:1385, "add options
      // for other linear systems solvers" is a TODO there); the dogleg's Newton step takes the selected solver (:2140)
      const Vec& g = EvalGradient(s);
      Vec gm(g.size());
      for (std::size_t i = 0; i < g.size(); ++i) gm[i] = g[i] + c.JT_lambda[i];
      SolveLinearSystem(s, gm, &c.Hinv_gm);
      c.hinv = true;
    }
    return c.Hinv_gm;
  }
  if (!c.hinv) {
    const Vec& g = EvalGradient(s);
    Scope prof_("device solve H^-1 g");
    if (DenseLinearSolver()) {
      SolveLinearSystem(s, g, &c.Hinv_gm);
    } else if (c.step_on_device && resident_ == &s && device_level_ == 3) {
      // launched right behind the assembly (CalcGradHess): -H^-1 g is waiting in device memory
      c.Hinv_gm = Fetch(IDTO_ARR_STEP);
      for (double& x : c.Hinv_gm) x = -x;
    } else {
      EnsureDevice(s, 3);
      c.Hinv_gm.resize(g.size());
      Check(idto_hip_solve_host(dev(), g.data(), 1, c.Hinv_gm.data()));
    }
    c.hinv = true;
  }
  return c.Hinv_gm;
}

// TO.cc:1371-1396 : lambda = (J~ H~^-1 J~^T)^-1 (h - J~ H~^-1 g~) = (J H^-1 J^T)^-1 (h - J H^-1 g).
// The n_eq + 1 solves with H, the Schur complement S = J H^-1 J^T and J H^-1 g are formed on the
// device from the dtau/dq blocks already there (idto_hip_constraint_schur); the host factorises
// the small dense S; the device then combines H^-1 (g + J^T lambda) and J^T lambda
// (idto_hip_constraint_step), which the dogleg step and the merit gradient use.
const VectorXd& TO::EvalLagrangeMultipliers(const TrajectoryOptimizerState<T>& s) const {
  auto& c = s.cache_;
  if (!c.lambda && num_equality_constraints() == 0) {  // fully actuated model: nothing to enforce
    c.lambda_v.clear();
    c.JT_lambda.assign((std::size_t)num_vars(), 0.0);
    c.lambda = true;
  }
  if (!c.lambda) {
    const Vec& h = EvalEqualityConstraintViolations(s);
    EvalGradient(s);
    EnsureDevice(s, 3);
    const int neq = num_equality_constraints(), nu = (int)unactuated_dofs_.size();
    c.lambda_v.resize((std::size_t)neq);
    c.Hinv_gm.resize((std::size_t)num_vars());
    c.JT_lambda.resize((std::size_t)num_vars());
    // Where S is factorised: on the device when it is large (the 25 small launches of the blocked
    // LDL^T cost ~0.3 ms at n_eq = 150 and 0.6 ms at 360; one host core needs 0.08 ms and 1.0 ms),
    // otherwise - and whenever the device finds S numerically singular - on the host.
    constexpr int kDeviceLdltFrom = 256;
    int rc = 1;
    Check(idto_hip_constraint_schur_begin(dev(), unactuated_dofs_.data(), nu));  // (no-op if CalcGradHess launched it)
    if (neq >= kDeviceLdltFrom) {
      Scope prof_("device: constraint solve");
      rc = idto_hip_constraint_solve(dev(), h.data(), c.lambda_v.data(), c.Hinv_gm.data(), c.JT_lambda.data());
      if (rc != 0 && rc != 1) Check(rc);  // (incl. IDTO_HIP_FACTORIZATION_FAILED)
    }
    if (rc == 1) {
      // pivoted LDL^T on the host, as Eigen's ldlt() of the reference (tolerates semi-definite S)
      Scope prof_("device schur + host LDLT + device step");
      std::vector<double> S((std::size_t)neq * neq);
      Vec rhs((std::size_t)neq);
      Check(idto_hip_constraint_schur(dev(), unactuated_dofs_.data(), nu, S.data(), rhs.data()));
      for (int r = 0; r < neq; ++r) rhs[r] = h[r] - rhs[r];
      {
        Scope prof_("dense LDLT (host)");
        DenseLdltSolve(&S, neq, rhs.data());
      }
      c.lambda_v = rhs;
      Check(idto_hip_constraint_step(dev(), c.lambda_v.data(), c.Hinv_gm.data(), c.JT_lambda.data()));
    }
    c.lambda = true;
  }
  return c.lambda_v;
}
// TO.cc:1411-1433
double TO::EvalMeritFunction(const TrajectoryOptimizerState<T>& s) const {
  if (!params_.equality_constraints) return EvalCost(s);
  auto& c = s.cache_;
  if (!c.merit) {
    c.merit_v = EvalCost(s) + Dot(EvalEqualityConstraintViolations(s), EvalLagrangeMultipliers(s));
    c.merit = true;
  }
  return c.merit_v;
}
// TO.cc:1435-1456
const VectorXd& TO::EvalMeritFunctionGradient(const TrajectoryOptimizerState<T>& s) const {
  if (!params_.equality_constraints) return EvalScaledGradient(s);
  auto& c = s.cache_;
  if (!c.mgrad) {
    const Vec& g = EvalScaledGradient(s);
    EvalLagrangeMultipliers(s);  // also leaves J^T lambda (unscaled J); J~^T lambda = D J^T lambda
    const Vec* D = params_.scaling ? &EvalScaleFactors(s) : nullptr;
    c.merit_gradient.resize((std::size_t)num_vars());
    for (int col = 0; col < num_vars(); ++col)
      c.merit_gradient[col] = g[col] + (D ? (*D)[col] * c.JT_lambda[col] : c.JT_lambda[col]);
    c.mgrad = true;
  }
  return c.merit_gradient;
}

// TO.cc:2691-2707
void TO::NormalizeQuaternions(TrajectoryOptimizerState<T>* state) const {
  std::vector<Vec> q = state->q();
  for (int qs : quaternion_starts_)
    for (Vec& qt : q) {
      const double n = std::sqrt(qt[qs] * qt[qs] + qt[qs + 1] * qt[qs + 1] + qt[qs + 2] * qt[qs + 2] + qt[qs + 3] * qt[qs + 3]);
      for (int k = 0; k < 4; ++k) qt[qs + k] /= n;
    }
  state->set_q(q);
}

namespace {
// TO.cc:2037-2066
double SolveDoglegQuadratic(double a, double b, double c) {
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
}  // namespace

// TO.cc:2108-2202
bool TO::CalcDoglegPoint(const TrajectoryOptimizerState<T>& s, double Delta, VectorXd* dq, VectorXd* dqH) const {
  const PentaDiagonalMatrix<T>& H = EvalScaledHessian(s);
  const Vec& g = EvalMeritFunctionGradient(s);
  const int n = num_vars();
  Vec Hg;
  H.MultiplyBy(g, &Hg);
  const double gHg = Dot(g, Hg);
  // pH = H~^-1 (-g~_merit / Delta) from the cached device solve (:2139-2149)
  const Vec& y = EvalHinvMeritGradient(s);
  Vec pH((std::size_t)n);
  {
    const Vec* D = params_.scaling ? &EvalScaleFactors(s) : nullptr;
    for (int i = 0; i < n; ++i) pH[i] = -(D ? y[i] / (*D)[i] : y[i]) / Delta;
  }
  if (params_.debug_compare_against_dense) {  // :2142-2150 (the reference's reference solution: dense LDL^T)
    Vec gm_unscaled = EvalGradient(s), yd;   // g + J^T lambda, unscaled (see EvalHinvMeritGradient)
    if (params_.equality_constraints && num_equality_constraints() > 0) {
      EvalLagrangeMultipliers(s);
      for (int i = 0; i < n; ++i) gm_unscaled[i] += s.cache_.JT_lambda[i];
    }
    SolveLinearSystem(s, gm_unscaled, &yd, /*dense*/ 0);
    const Vec* D = params_.scaling ? &EvalScaleFactors(s) : nullptr;
    double num = 0, den = 0;
    for (int i = 0; i < n; ++i) {
      const double pd = -(D ? yd[i] / (*D)[i] : yd[i]) / Delta;
      num += (pH[i] - pd) * (pH[i] - pd);
      den += pd * pd;
    }
    last_sparse_vs_dense_ = std::sqrt(num) / std::sqrt(den);
    std::printf("Sparse vs. Dense error: %g\n", last_sparse_vs_dense_);
    std::fflush(stdout);
  }
  dqH->resize((std::size_t)n);
  for (int i = 0; i < n; ++i) (*dqH)[i] = pH[i] * Delta;  // :2152
  Vec pU((std::size_t)n);
  const double coef = -(Dot(g, g) / gHg);
  for (int i = 0; i < n; ++i) pU[i] = coef * g[i] / Delta;  // :2157
  dq->resize((std::size_t)n);
  auto apply_scaling = [&]() {
    if (params_.scaling) {
      const Vec& D = EvalScaleFactors(s);
      for (int i = 0; i < n; ++i) (*dq)[i] = D[i] * (*dq)[i];
    }
  };
  const double pUn = Norm(pU);
  if (1.0 <= pUn) {  // :2160-2168
    for (int i = 0; i < n; ++i) (*dq)[i] = (Delta / pUn) * pU[i];
    apply_scaling();
    return true;
  }
  if (1.0 >= Norm(pH)) {  // :2171-2178
    for (int i = 0; i < n; ++i) (*dq)[i] = pH[i] * Delta;
    apply_scaling();
    return false;
  }
  Vec d((std::size_t)n);
  for (int i = 0; i < n; ++i) d[i] = pH[i] - pU[i];
  const double a = Dot(d, d), b = 2 * Dot(pU, d), c = Dot(pU, pU) - 1.0;  // :2192-2194
  const double sq = SolveDoglegQuadratic(a, b, c);
  for (int i = 0; i < n; ++i) (*dq)[i] = (pU[i] + sq * d[i]) * Delta;  // :2197
  apply_scaling();
  return true;
}

// The accepted iterate q_k + dq is the trial point CalcTrustRatio just evaluated in `scratch`
// (same additions, same bits): its tau / cost / h and its residency on the device carry over
// instead of being recomputed (the reference re-evaluates them through its cache, TO.cc:2550-2553).
void TO::AdoptTrialPoint(const TrajectoryOptimizerState<T>& scratch, TrajectoryOptimizerState<T>* state) const {
  if (!scratch.cache_.traj || scratch.q() != state->q()) return;
  auto& c = state->cache_;
  c.tau = scratch.cache_.tau;
  c.cost = scratch.cache_.cost;
  c.traj = true;
  if (scratch.cache_.h) { c.h_viol = scratch.cache_.h_viol; c.h = true; }
  if (resident_ == &scratch && scratch.cache_.uploaded) {
    resident_ = state;
    c.uploaded = true;
  }
}

// TO.cc:1979-2035
double TO::CalcTrustRatio(const TrajectoryOptimizerState<T>& s, const VectorXd& dq,
                          TrajectoryOptimizerState<T>* scratch) const {
  const double merit_k = EvalMeritFunction(s);
  const Vec& g_tilde_k = EvalMeritFunctionGradient(s);
  const PentaDiagonalMatrix<T>& H_k = EvalScaledHessian(s);
  if (params_.equality_constraints) EvalLagrangeMultipliers(s);
  scratch->set_q(s.q());
  scratch->AddToQ(dq);
  if (params_.normalize_quaternions) NormalizeQuaternions(scratch);
  double merit_kp = EvalCost(*scratch);
  if (params_.equality_constraints)
    merit_kp += Dot(EvalEqualityConstraintViolations(*scratch), EvalLagrangeMultipliers(s));
  const int n = num_vars();
  Vec dqs((std::size_t)n), Hdq;
  if (params_.scaling) {
    const Vec& D = EvalScaleFactors(s);
    for (int i = 0; i < n; ++i) dqs[i] = (1.0 / D[i]) * dq[i];
  } else {
    dqs = dq;
  }
  H_k.MultiplyBy(dqs, &Hdq);
  const double hessian_term = 0.5 * Dot(dqs, Hdq);
  const double gradient_term = Dot(g_tilde_k, dqs);
  const double predicted = -gradient_term - hessian_term;
  const double actual = merit_k - merit_kp;
  const double eps = 10 * std::numeric_limits<double>::epsilon() / time_step_ / time_step_;
  if (predicted < eps && actual < eps) return 0.5;
  return actual / predicted;
}

// TO.cc:2653-2689
ConvergenceReason TO::VerifyConvergenceCriteria(const TrajectoryOptimizerState<T>& s, double previous_cost,
                                                const VectorXd& dq) const {
  const auto& tol = params_.convergence_tolerances;
  int reason = kNoConvergenceCriteriaSatisfied;
  const double cost = EvalCost(s);
  if (std::fabs(previous_cost - cost) < tol.abs_cost_reduction + tol.rel_cost_reduction * cost)
    reason |= kCostReductionCriterionSatisfied;
  const Vec& g = EvalMeritFunctionGradient(s);
  if (std::fabs(Dot(g, dq)) < tol.abs_gradient_along_dq + tol.rel_gradient_along_dq * cost)
    reason |= kGradientCriterionSatisfied;
  if (Norm(dq) < tol.abs_state_change + tol.rel_state_change * s.norm()) reason |= kSateCriterionSatisfied;
  return static_cast<ConvergenceReason>(reason);
}

// TO.cc:2213-2234
SolverFlag TO::Solve(const std::vector<VectorXd>& q_guess, TrajectoryOptimizerSolution<T>* solution,
                     TrajectoryOptimizerStats<T>* stats, ConvergenceReason* reason) const {
  if (!stats->is_empty()) throw std::runtime_error("Solve: stats must be empty (TO.cc:2225)");
  if ((int)q_guess.size() != num_steps() + 1) throw std::runtime_error("Solve: q_guess has the wrong length");
  try {
    if (params_.method == kLinesearch) return SolveWithLinesearch(q_guess, solution, stats);
    std::unique_ptr<WarmStart> ws = CreateWarmStart(q_guess);
    return SolveFromWarmStart(ws.get(), solution, stats, reason);
  } catch (const FactorizationFailedError& e) {
    if (params_.verbose) std::printf("FACTORIZATION FAILED\n%s\n", e.what());
    return SolverFlag::kFactorizationFailed;
  }
}

// TO.cc:1931-1977
std::pair<double, int> TO::ArmijoLinesearch(const TrajectoryOptimizerState<T>& s, const VectorXd& dq,
                                            TrajectoryOptimizerState<T>* scratch) const {
  const double L = EvalCost(s);
  const Vec& g = EvalGradient(s);
  const double c = 1e-4, rho = 0.8;
  double alpha = 1.0 / rho;
  const double L_prime = Dot(g, dq);
  if (!(L_prime <= 0)) throw std::runtime_error("linesearch: not a descent direction (TO.cc:1951)");
  const double thr = 10 * std::numeric_limits<double>::epsilon() / time_step_ / time_step_;
  if (std::fabs(L_prime) / std::fabs(L) <= thr) return {1.0, 0};
  int i = 0;
  double L_new;
  Vec step(dq.size());
  do {
    alpha *= rho;
    scratch->set_q(s.q());
    for (std::size_t j = 0; j < dq.size(); ++j) step[j] = alpha * dq[j];
    scratch->AddToQ(step);
    if (params_.normalize_quaternions) NormalizeQuaternions(scratch);
    L_new = EvalCost(*scratch);  // one device trial point per linesearch iteration
    ++i;
  } while ((L_new > L + c * alpha * L_prime) && (i < params_.max_linesearch_iterations));
  return {alpha, i};
}

// TO.cc:1852-1929
std::pair<double, int> TO::BacktrackingLinesearch(const TrajectoryOptimizerState<T>& s, const VectorXd& dq,
                                                  TrajectoryOptimizerState<T>* scratch) const {
  const double mu = params_.equality_constraints ? 1e3 : 0.0;  // l1 penalty on the constraint violations
  auto l1 = [](const Vec& h) { double t = 0; for (double x : h) t += std::fabs(x); return t; };
  const Vec& h = EvalEqualityConstraintViolations(s);
  const double L = EvalCost(s) + mu * l1(h);
  const Vec& g = EvalGradient(s);
  const double c = 1e-4, rho = 0.8;
  double alpha = 1.0;
  const double L_prime = Dot(g, dq) - mu * l1(h);
  if (!(L_prime <= 0)) throw std::runtime_error("linesearch: not a descent direction (TO.cc:1888)");
  if (std::fabs(L_prime) / std::fabs(L) <= std::sqrt(std::numeric_limits<double>::epsilon())) return {1.0, 0};
  Vec step(dq.size());
  auto eval_at = [&](double al) {
    scratch->set_q(s.q());
    for (std::size_t j = 0; j < dq.size(); ++j) step[j] = al * dq[j];
    scratch->AddToQ(step);
    if (params_.normalize_quaternions) NormalizeQuaternions(scratch);
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
SolverFlag TO::SolveWithLinesearch(const std::vector<VectorXd>& q_guess, TrajectoryOptimizerSolution<T>* solution,
                                   TrajectoryOptimizerStats<T>* stats) const {
  using clock = std::chrono::high_resolution_clock;
  const auto start_time = clock::now();
  TrajectoryOptimizerState<T> state = CreateState();
  state.set_q(q_guess);
  TrajectoryOptimizerState<T> scratch = CreateState();
  Vec dq((std::size_t)num_vars()), rhs((std::size_t)num_vars()), step((std::size_t)num_vars());
  if (params_.verbose) {
    std::printf("-----------------------------------------------------------------------------------\n");
    std::printf("|  iter  |   cost   |  alpha  |  LS_iters  |  time (s)  |  |g|/cost  |    |h|     |\n");
    std::printf("-----------------------------------------------------------------------------------\n");
  }
  int k = 0;
  bool linesearch_failed = false;
  do {
    const auto iter_start = clock::now();
    const double cost = EvalCost(state);
    const double h_norm = Norm(EvalEqualityConstraintViolations(state));
    const Vec& g = EvalMeritFunctionGradient(state);
    // dq = -H^-1 g with the unscaled Hessian (:2299-2300): one device factorisation + solve
    EvalGradient(state);
    EnsureDevice(state, 3);
    for (std::size_t i = 0; i < rhs.size(); ++i) rhs[i] = -g[i];
    SolveLinearSystem(state, rhs, &dq);  // :2302
    double debug_cond = 0, debug_diag_norm = 0;
    if (params_.print_debug_data) {
      debug_cond = DebugConditionNumber(state, false);
      std::vector<double> diag;
      EvalHessian(state).ExtractDiagonal(&diag);
      debug_diag_norm = Norm(diag);
    }
    const auto [alpha, ls_iters] = (params_.linesearch_method == kArmijo) ? ArmijoLinesearch(state, dq, &scratch)
                                                                         : BacktrackingLinesearch(state, dq, &scratch);
    if (ls_iters >= params_.max_linesearch_iterations) {
      linesearch_failed = true;
      if (params_.verbose)
        std::printf("LINESEARCH FAILED\nReached maximum linesearch iterations (%d).\n", params_.max_linesearch_iterations);
    }
    for (std::size_t i = 0; i < dq.size(); ++i) step[i] = alpha * dq[i];
    const double trust_ratio = CalcTrustRatio(state, step, &scratch);
    const double g_norm = Norm(g), dq_norm = Norm(dq), dL_dq = Dot(g, dq) / cost;  // (g dies with the cache below)
    state.AddToQ(step);
    if (params_.normalize_quaternions) NormalizeQuaternions(&state);
    AdoptTrialPoint(scratch, &state);
    const double iter_time = std::chrono::duration<double>(clock::now() - iter_start).count();
    if (params_.verbose)
      std::printf("| %6d | %8.3f | %7.4f | %6d     | %8.8f | %10.3e | %10.3e |\n", k, cost, alpha, ls_iters, iter_time,
                  g_norm / cost, h_norm);
    if (params_.print_debug_data) {  // :2349-2365 (the quantities of the iterate the step was computed at)
      std::printf("Condition #: %g\n|| dq ||   : %g\n||  g ||   : %g\nL'         : %g\nL          : %g\nL' / L     : %g\n"
                  "||diag(H)||: %g\n", debug_cond, dq_norm, g_norm, dL_dq * cost, cost, dL_dq, debug_diag_norm);
      if (k > 0) std::printf("L[k] - L[k-1]: %g\n", cost - stats->iteration_costs[(std::size_t)k - 1]);
      std::fflush(stdout);
    }
    stats->push_data(iter_time, cost, ls_iters, alpha, std::numeric_limits<double>::quiet_NaN(), state.norm(), dq_norm,
                     dq_norm, trust_ratio, g_norm, dL_dq, h_norm, cost);  // :2373-2385
    ++k;
  } while (k < params_.max_iterations && !linesearch_failed);
  if (params_.verbose)
    std::printf("-----------------------------------------------------------------------------------\n");
  stats->solve_time = std::chrono::duration<double>(clock::now() - start_time).count();
  solution->q = state.q();
  solution->v = EvalV(state);
  solution->tau = EvalTau(state);
  return linesearch_failed ? SolverFlag::kLinesearchMaxIters : SolverFlag::kSuccess;
}

// TO.cc:2449-2651
SolverFlag TO::SolveFromWarmStart(WarmStart* ws, TrajectoryOptimizerSolution<T>* solution,
                                  TrajectoryOptimizerStats<T>* stats, ConvergenceReason* reason_out) const {
  for (int attempt = 0;; ++attempt) {
    try {
      return SolveFromWarmStartImpl(ws, solution, stats, reason_out);
    } catch (const FactorizationFailedError& e) {
      if (params_.verbose) std::printf("FACTORIZATION FAILED\n%s\n", e.what());
      return SolverFlag::kFactorizationFailed;
    } catch (const SolverTimeoutError& e) {
      // (the warm start's state is only advanced by completed iterations: run again from it, on the variant
      // the context has stepped down to - at most once per variant that can time out)
      if (attempt >= 4) throw;
      if (params_.verbose) std::printf("%s\n", e.what());
      *stats = TrajectoryOptimizerStats<T>();
      resident_ = nullptr;
      device_level_ = 0;
    }
  }
}

// The trust-region loop with every O(num_vars) quantity left on the device: per
// iteration the host launches idto_hip_gn_step (partials, g, H, -H^-1 g in one launch),
// idto_hip_tr_prepare (scale factors, g~, H~ g~, H~ w, nine inner products) and idto_hip_tr_trial
// (dq = D (a g~ + b w), q + dq, tau and cost there), and reads back 9 + 4 scalars.  The dogleg
// (TO.cc:2139-2201) and the trust ratio (TO.cc:1979-2035) are functions of those inner products:
// with pU = cU g~ and pH = -w / Delta every step is dqs = a g~ + b w, so
//   g~.dqs = a g~.g~ + b g~.w,   dqs.H~ dqs = a^2 g~.H~g~ + 2 a b g~.H~w + b^2 w.H~w.
// Used when no equality constraints are enforced and convergence checks are off (the configuration
// of the mini_cheetah example); otherwise SolveFromWarmStartImpl below keeps g / H on the host.
bool TO::DeviceLoopEligible() const {
  if (std::getenv("IDTO_OPT_HOST_LOOP") || force_host_loop_) return false;
  if (!shard_ctx_.empty()) return false;
  // the dense solver and the two debugging switches live in the stepwise loop (SolveLinearSystem, CalcDoglegPoint)
  if (DenseLinearSolver() || params_.debug_compare_against_dense || params_.print_debug_data) return false;
  const bool constrained = params_.equality_constraints && num_equality_constraints() > 0;
  if (!constrained && !params_.check_convergence) return true;
  // enforced constraints and the convergence criteria: only the resident loop (idto_hip_tr_solve) has them on the
  // device (multipliers by a single-workgroup LDL^T of S for n_eq <= 128, by the blocked one above; the criteria
  // by the iteration kernel that follows an accepted step)
  return ResidentLoopEligible();
}

// every iteration enqueued at once, decisions on the device, one wait (idto_hip_tr_solve)
bool TO::ResidentLoopEligible() const {
  if (params_.max_iterations <= 0 || std::getenv("IDTO_OPT_STEPWISE")) return false;
  int resident_ok = 1;   // (0 once the iteration kernel's workgroups timed out waiting for each other on this context)
  Check(idto_hip_get_option(dev(), "tr_resident_ok", &resident_ok));
  if (!resident_ok) return false;
  const int scal = params_.scaling ? static_cast<int>(params_.scaling_method) : -1;
  const bool adaptive = scal == static_cast<int>(kAdaptiveSqrt) || scal == static_cast<int>(kAdaptiveDoubleSqrt);
  if (!adaptive) return true;
  int weights_diagonal = 0;   // (the adaptive scalings ride on the gated assembly, which serves diagonal cost weights)
  Check(idto_hip_get_option(dev(), "weights_diagonal", &weights_diagonal));
  return weights_diagonal != 0;
}

SolverFlag TO::SolveOnDevice(WarmStart* ws, TrajectoryOptimizerSolution<T>* solution,
                             TrajectoryOptimizerStats<T>* stats, ConvergenceReason* reason_out) const {
  using clock = std::chrono::high_resolution_clock;
  const auto start_time = clock::now();
  auto iter_start = clock::now();
  TrajectoryOptimizerState<T>& state = ws->state;
  double& Delta = ws->Delta;
  const double eta = 0.0;
  {
    const Vec q0 = Flatten(state.q());
    Check(idto_hip_set_q(dev(), q0.data()));
  }
  Check(idto_hip_set_unactuated_dofs(dev(), unactuated_dofs_.data(), (int)unactuated_dofs_.size()));
  idto_hip_trace_mark("opt: q and the unactuated dofs set");
  // (the resident loop starts with the partials of this q: one finite-difference launch for both)
  Check(ResidentLoopEligible() ? idto_hip_eval_tau_partials(dev()) : idto_hip_eval_tau(dev()));
  // (the resident loop keeps the cost on the device and reports it in its rows: no fetch, no synchronisation here)
  double cost = ResidentLoopEligible() ? 0.0 : Fetch(IDTO_ARR_COST)[0];
  const int scal = params_.scaling ? static_cast<int>(params_.scaling_method) : -1;
  // the adaptive scalings' memory of D belongs to the state (TO.cc:1241-1255 reads the cached scale
  // factors): ones for a fresh state, the last solve's for a warm start - never the context's leftovers
  const bool adaptive = scal == static_cast<int>(kAdaptiveSqrt) || scal == static_cast<int>(kAdaptiveDoubleSqrt);
  if (adaptive) {
    const Vec& Dmem = state.cache_.scale_factors;
    Check(idto_hip_tr_set_scale_memory(dev(), (int)Dmem.size() == num_vars() ? Dmem.data() : nullptr));
  }
  if (params_.verbose) {
    std::printf("-------------------------------------------------------------------------------------\n");
    std::printf("|  iter  |   cost   |    Δ    |    ρ    |  time (s)  |  |g|/cost  |    dL_dq   |    |h|     |\n");
    std::printf("-------------------------------------------------------------------------------------\n");
  }
  bool have = false, last_accepted = true;
  double S[9] = {0};
  int k = 0;
  Vec fq, fv, ft, fdq, fw;   // what idto_hip_tr_solve_fetch brought along
  bool fetched = false;
  // every iteration enqueued at once, decisions on the device, one wait (idto_hip_tr_solve); the
  // stepwise loop below serves IDTO_OPT_STEPWISE=1 (and the adaptive scalings with dense cost weights)
  const bool resident_loop = ResidentLoopEligible();
  bool converged = false;
  if (resident_loop) {
    if (params_.check_convergence) {   // VerifyConvergenceCriteria (TO.cc:2654-2689) inside the device loop
      const auto& t = params_.convergence_tolerances;
      const double tol[6] = {t.rel_cost_reduction, t.abs_cost_reduction, t.rel_gradient_along_dq, t.abs_gradient_along_dq,
                             t.rel_state_change, t.abs_state_change};
      Check(idto_hip_tr_set_convergence(dev(), tol));
    } else {
      Check(idto_hip_tr_set_convergence(dev(), nullptr));
    }
    const int iters = params_.max_iterations;
    std::vector<double> rows((std::size_t)iters * IDTO_TR_ROW);
    double Delta_end = Delta;
    const bool constrained = params_.equality_constraints && num_equality_constraints() > 0;
    // (the solution comes back under the loop's own wait: idto_hip_tr_solve_fetch)
    fq.resize((std::size_t)num_vars()); fv.resize((std::size_t)(num_steps() + 1) * nv_); ft.resize((std::size_t)num_steps() * nv_);
    fdq.resize((std::size_t)num_vars()); fw.resize((std::size_t)num_vars());
    Check(idto_hip_tr_solve_fetch(dev(), iters, scal, params_.scaling ? 1 : 0, params_.normalize_quaternions ? 1 : 0, Delta,
                                  params_.Delta_max, eta, constrained ? unactuated_dofs_.data() : nullptr,
                                  constrained ? (int)unactuated_dofs_.size() : 0, rows.data(), &Delta_end, fq.data(), fv.data(),
                                  ft.data(), fdq.data(), fw.data()));
    fetched = true;
    idto_hip_trace_mark("opt: idto_hip_tr_solve_fetch returned");
    const double total = std::chrono::duration<double>(clock::now() - start_time).count();
    // (with the convergence criteria on the device loop stops within eight iterations of the one that met them: the
    // rows behind it are zeros)
    int ran = iters;
    while (ran > 1 && rows[(std::size_t)(ran - 1) * IDTO_TR_ROW + 10] == 0.0) --ran;
    double timed = 0.0;
    for (int i = 1; i < ran; ++i) timed += (rows[(std::size_t)i * IDTO_TR_ROW + 10] - rows[(std::size_t)(i - 1) * IDTO_TR_ROW + 10]) * 1e-8;
    for (; k < ran; ++k) {
      const double* R = rows.data() + (std::size_t)k * IDTO_TR_ROW;
      const int flags = (int)R[14];
      if (flags & 8) {
        // numerically singular S = J H^-1 J^T (redundant constraints) at this iterate: the host loop's pivoted
        // LDL^T (Eigen's, in the reference) handles it; q on the device is still this iteration's iterate
        state.set_q(Unflatten(Fetch(IDTO_ARR_Q), num_steps() + 1, nq_));
        resident_ = nullptr;
        device_level_ = 0;
        Delta = R[1];
        force_host_loop_ = true;
        resume_k_ = k;
        struct Reset { const TO* t; ~Reset() { t->force_host_loop_ = false; t->resume_k_ = 0; } } reset{this};
        const double device_part = std::chrono::duration<double>(clock::now() - start_time).count();
        const SolverFlag flag = SolveFromWarmStartImpl(ws, solution, stats, nullptr);
        stats->solve_time += device_part;
        return flag;
      }
      if (flags & 32)   // (in the iteration it happened in, not only when it is the last one: idto_hip_tr_solve rows [14])
        throw FactorizationFailedError("idto_hip: factorisation failed in iteration " + std::to_string(k));
      if (flags & 3) throw FactorizationFailedError("idto_hip: the dogleg step is not finite");
      if (flags & 4) throw std::runtime_error("step is not a descent direction (TO.cc:2531)");
      const double iter_time = (k == 0) ? std::max(0.0, total - timed) : (R[10] - R[10 - IDTO_TR_ROW]) * 1e-8;
      if (params_.verbose)
        std::printf("| %6d | %8.3g | %7.2g | %7.3g | %10.5g | %10.5g | %10.4g | %10.4g |\n", k, R[0], R[1], R[2], iter_time,
                    R[6] / R[0], R[7], R[8]);
      stats->push_data(iter_time, R[0], 0, std::numeric_limits<double>::quiet_NaN(), R[1], R[3], R[4], R[5], R[2], R[6],
                       R[7], R[8], R[15]);   // :2586-2598 (merit = cost without constraints)
      last_accepted = R[9] != 0.0;
      if (params_.check_convergence && last_accepted) {   // :2600-2612
        const ConvergenceReason reason = static_cast<ConvergenceReason>((int)R[16]);
        stats->convergence_reason = reason;
        if (reason_out) *reason_out = reason;
        if (reason != kNoConvergenceCriteriaSatisfied) {   // (the reference leaves the loop before the radius update)
          converged = true;
          Delta = R[1];
          break;
        }
      }
    }
    if (!converged) Delta = Delta_end;
    // (the resident loop carries the cost on the device and `cost` is still the placeholder.  Every device
    // flag either throws or yields a converged row, so the stepwise loop below is not reached from here in normal
    // operation - but if it ever is, after the chunked early exit's truncation say, its trust ratio must not be
    // formed against 0.  The iterate's cost is in the last row: L(q_k + dq) if that step was accepted, L(q_k) if not -
    // NOT IDTO_ARR_COST, which every trial point overwrites, a rejected one included.)
    if (k < params_.max_iterations && !converged && ran > 0) {
      const double* R = rows.data() + (std::size_t)(ran - 1) * IDTO_TR_ROW;
      cost = (R[9] != 0.0) ? R[13] : R[0];
    }
  }
  while (k < params_.max_iterations && !converged) {
    fetched = false;   // (the device moves on: what the resident loop brought along is stale)
    if (!have) {
      Check(idto_hip_gn_step(dev()));
      Check(idto_hip_tr_prepare(dev(), scal, 0, S));   // (reports a failed factorisation)
      have = true;
    }
    const double gg = S[0], gHg = S[1], ww = S[2], gw = S[3], gHw = S[4], wHw = S[5], qq = S[6], hh = S[7];
    // CalcDoglegPoint, normalised by Delta: pU = cU g~ (:2157), pH = -w / Delta (:2139-2149)
    const double cU = -(gg / gHg) / Delta;
    const double pUn = std::fabs(cU) * std::sqrt(gg), pHn = std::sqrt(ww) / Delta;
    double a, b;
    bool active;
    if (1.0 <= pUn) {          // :2160-2168
      a = (Delta / pUn) * cU; b = 0.0; active = true;
    } else if (1.0 >= pHn) {   // :2171-2178
      a = 0.0; b = -1.0; active = false;
    } else {                   // :2180-2199
      const double pUpU = cU * cU * gg, pHpH = ww / (Delta * Delta), pUpH = -cU * gw / Delta;
      const double sq = SolveDoglegQuadratic(pHpH - 2 * pUpH + pUpU, 2 * (pUpH - pUpU), pUpU - 1.0);
      a = Delta * (1.0 - sq) * cU; b = -sq; active = true;
    }
    if (!std::isfinite(a) || !std::isfinite(b)) throw FactorizationFailedError("idto_hip: the dogleg step is not finite");
    double Tr[4];
    // speculate on acceptance (the next iteration is enqueued behind the trial point) unless the last
    // step was rejected: a rejection costs a recomputation of g and H at the old q
    const bool speculate = last_accepted && !std::getenv("IDTO_OPT_NO_SPECULATION");
    Check(idto_hip_tr_trial(dev(), a, b, params_.scaling ? 1 : 0, params_.normalize_quaternions ? 1 : 0, 0,
                            speculate ? scal : -2, Tr));
    const double dq_norm = std::sqrt(Tr[0]), gdqs = Tr[1], cost_trial = Tr[2];
    if (!std::isfinite(Tr[0])) throw FactorizationFailedError("idto_hip: the dogleg step is not finite");
    const double dL_dq = gdqs / cost;   // :2517-2524
    // CalcTrustRatio (:2004-2034)
    const double gradient_term = a * gg + b * gw;
    const double hessian_term = 0.5 * (a * a * gHg + 2 * a * b * gHw + b * b * wHw);
    const double predicted = -gradient_term - hessian_term, actual = cost - cost_trial;
    const double eps = 10 * std::numeric_limits<double>::epsilon() / time_step_ / time_step_;
    const double rho = (predicted < eps && actual < eps) ? 0.5 : actual / predicted;
    if (!(dL_dq < std::numeric_limits<double>::epsilon()))
      throw std::runtime_error("step is not a descent direction (TO.cc:2531)");
    const double g_norm = std::sqrt(gg), h_norm = std::sqrt(hh), q_norm = std::sqrt(qq), dqH_norm = std::sqrt(ww);
    last_accepted = rho > eta;
    const double cost_k = cost;
    if (last_accepted) {   // :2550-2553
      Check(idto_hip_tr_accept(dev()));
      cost = cost_trial;
      have = false;
    } else {
      Check(idto_hip_tr_reject(dev()));
      if (speculate) have = false;   // the speculative launch overwrote g, H and the Newton step of q
    }
    const double iter_time = std::chrono::duration<double>(clock::now() - iter_start).count();
    iter_start = clock::now();
    if (params_.verbose)
      std::printf("| %6d | %8.3g | %7.2g | %7.3g | %10.5g | %10.5g | %10.4g | %10.4g |\n", k, cost_k, Delta, rho, iter_time,
                  g_norm / cost_k, dL_dq, h_norm);
    stats->push_data(iter_time, cost_k, 0, std::numeric_limits<double>::quiet_NaN(), Delta, q_norm, dq_norm, dqH_norm, rho,
                     g_norm, dL_dq, h_norm, cost_k);   // :2586-2598 (merit = cost without constraints)
    if (rho < 0.25) Delta *= 0.25;                                                   // :2614-2617
    else if (rho > 0.75 && active) Delta = std::min(2 * Delta, params_.Delta_max);   // :2618-2622
    ++k;
  }
  idto_hip_trace_mark("opt: statistics rows taken in");
  // the solution: q from the device; v, tau belong to it unless the last trial point was rejected
  int two_sets = 0;   // (diagonal cost weights: the trial point was evaluated into the set the iterate does not occupy)
  if (fetched) Check(idto_hip_get_option(dev(), "weights_diagonal", &two_sets));
  fetched = fetched && (two_sets != 0 || last_accepted);
  if (!fetched && !last_accepted) Check(idto_hip_eval_tau(dev()));
  // (the solution's arrays: with the resident loop's own wait - idto_hip_tr_solve_fetch - or with one synchronisation,
  // idto_hip_get_many)
  {
    Vec qf, vf, tf, dq, dqh;
    if (fetched) {
      qf.swap(fq); vf.swap(fv); tf.swap(ft); dq.swap(fdq); dqh.swap(fw);
    } else {
      qf.resize((std::size_t)num_vars()); vf.resize((std::size_t)(num_steps() + 1) * nv_); tf.resize((std::size_t)num_steps() * nv_);
      dq.resize((std::size_t)num_vars()); dqh.resize((std::size_t)num_vars());
      const int what[5] = {IDTO_ARR_Q, IDTO_ARR_V, IDTO_ARR_TAU, IDTO_ARR_TR_DQ, IDTO_ARR_TR_W};
      double* const dst[5] = {qf.data(), vf.data(), tf.data(), dq.data(), dqh.data()};
      Check(idto_hip_get_many(dev(), k > 0 ? 5 : 3, what, dst));
    }
    UnflattenInto(qf, num_steps() + 1, nq_, &solution->q);
    state.set_q(solution->q);
    UnflattenInto(vf, num_steps() + 1, nv_, &solution->v);
    UnflattenInto(tf, num_steps(), nv_, &solution->tau);
    if (k > 0) {
      ws->dq = dq;
      ws->dqH = dqh;
      for (double& x : ws->dqH) x = -x;   // dqH = Delta pH = -w (:2152)
    }
  }
  idto_hip_trace_mark("opt: solution unpacked");
  if (adaptive && k > 0) state.cache_.scale_factors = Fetch(IDTO_ARR_TR_SCALE);   // (kept across set_q: invalidate_cache)
  resident_ = nullptr;   // the device arrays were advanced without the host-side cache
  device_level_ = 0;
  stats->solve_time = std::chrono::duration<double>(clock::now() - start_time).count();
  if (k == params_.max_iterations) return SolverFlag::kMaxIterationsReached;
  return SolverFlag::kSuccess;
}

SolverFlag TO::SolveFromWarmStartImpl(WarmStart* ws, TrajectoryOptimizerSolution<T>* solution,
                                      TrajectoryOptimizerStats<T>* stats, ConvergenceReason* reason_out) const {
  using clock = std::chrono::high_resolution_clock;
  if (params_.method == kTrustRegion && DeviceLoopEligible()) return SolveOnDevice(ws, solution, stats, reason_out);
  if (params_.method != kTrustRegion) throw std::runtime_error("warm start requires the trust-region method");
  const auto start_time = clock::now();
  auto iter_start = clock::now();
  g_profile.NewSolve();
  Scope prof_("SolveFromWarmStart total");
  TrajectoryOptimizerState<T>& state = ws->state;
  TrajectoryOptimizerState<T>& scratch = ws->scratch_state;
  Vec& dq = ws->dq;
  Vec& dqH = ws->dqH;
  const double eta = 0.0;  // trust ratio threshold (:2466)
  int k = resume_k_;       // (> 0: taking over from the device-resident loop)
  double& Delta = ws->Delta;
  double previous_cost = EvalCost(state);
  if (params_.verbose) {
    std::printf("-------------------------------------------------------------------------------------\n");
    std::printf("|  iter  |   cost   |    Δ    |    ρ    |  time (s)  |  |g|/cost  |    dL_dq   |    |h|     |\n");
    std::printf("-------------------------------------------------------------------------------------\n");
  }
  while (k < params_.max_iterations) {
    const bool active = CalcDoglegPoint(state, Delta, &dq, &dqH);  // :2497
    if (params_.print_debug_data) {  // :2499-2507
      std::printf("condition_number = %g\n", DebugConditionNumber(state, false));
      std::printf("condition_number_scaled = %g\n", DebugConditionNumber(state, true));
      std::fflush(stdout);
    }
    // a step that is not finite can only come from a Hessian the factorisation could not handle
    if (!std::isfinite(Dot(dq, dq))) throw FactorizationFailedError("idto_hip: the dogleg step is not finite");
    const Vec& g = EvalMeritFunctionGradient(state);
    const Vec& h = EvalEqualityConstraintViolations(state);
    const double cost = EvalCost(state);
    const double merit = EvalMeritFunction(state);
    const double q_norm = state.norm();
    double dL_dq;
    if (params_.scaling) {
      const Vec& D = EvalScaleFactors(state);
      double acc = 0;
      for (std::size_t i = 0; i < dq.size(); ++i) acc += g[i] * ((1.0 / D[i]) * dq[i]);
      dL_dq = acc / cost;
    } else {
      dL_dq = Dot(g, dq) / cost;
    }
    const double rho = CalcTrustRatio(state, dq, &scratch);  // :2526
    if (!(dL_dq < std::numeric_limits<double>::epsilon()))
      throw std::runtime_error("step is not a descent direction (TO.cc:2531)");
    const double g_norm = Norm(g), h_norm = Norm(h), dq_norm = Norm(dq), dqH_norm = Norm(dqH);
    if (rho > eta) {  // :2550-2553
      state.AddToQ(dq);
      if (params_.normalize_quaternions) NormalizeQuaternions(&state);
      AdoptTrialPoint(scratch, &state);
    }
    const double iter_time = std::chrono::duration<double>(clock::now() - iter_start).count();
    iter_start = clock::now();
    if (params_.verbose)
      std::printf("| %6d | %8.3g | %7.2g | %7.3g | %10.5g | %10.5g | %10.4g | %10.4g |\n", k, cost, Delta, rho, iter_time,
                  g_norm / cost, dL_dq, h_norm);
    stats->push_data(iter_time, cost, 0, std::numeric_limits<double>::quiet_NaN(), Delta, q_norm, dq_norm, dqH_norm, rho,
                     g_norm, dL_dq, h_norm, merit);  // :2586-2598
    ConvergenceReason reason = kNoConvergenceCriteriaSatisfied;
    if (params_.check_convergence && rho > eta) {
      reason = VerifyConvergenceCriteria(state, previous_cost, dq);
      previous_cost = EvalCost(state);
      stats->convergence_reason = reason;
      if (reason_out) *reason_out = reason;
    }
    if (reason != kNoConvergenceCriteriaSatisfied) break;
    if (rho < 0.25) Delta *= 0.25;                                                   // :2614-2617
    else if (rho > 0.75 && active) Delta = std::min(2 * Delta, params_.Delta_max);   // :2618-2622
    ++k;
  }
  stats->solve_time = std::chrono::duration<double>(clock::now() - start_time).count();
  solution->q = state.q();
  solution->v = EvalV(state);
  solution->tau = EvalTau(state);
  if (k == params_.max_iterations) return SolverFlag::kMaxIterationsReached;
  return SolverFlag::kSuccess;
}

}  // namespace optimizer
}  // namespace idto
