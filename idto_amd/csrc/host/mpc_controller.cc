// mpc_controller.cc — include/idto/examples/mpc_controller.h (reference examples/mpc_controller.cc:13-178).
#include "idto/examples/mpc_controller.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace idto {
namespace examples {
namespace mpc {

// ---- PiecewiseCubic: first derivatives m_i at the knots from the linear system of C2 continuity
//   h_i m_{i-1} + 2 (h_{i-1} + h_i) m_i + h_{i-1} m_{i+1} = 3 (h_i d_{i-1} + h_{i-1} d_i),  d_i = (y_{i+1} - y_i) / h_i
// closed by the not-a-knot conditions at both ends; one dense LU with partial pivoting (n <= a few hundred knots)
// serves every component.
PiecewiseCubic::PiecewiseCubic(const std::vector<double>& breaks, const std::vector<VectorXd>& knots) {
  Assign(breaks, knots, (int)knots.size());
}
// (an object that is assigned again keeps its storage: three splines per MPC re-plan, no allocation in the steady state)
void PiecewiseCubic::Assign(const std::vector<double>& breaks, const std::vector<VectorXd>& knots, int count) {
  const int n = (int)breaks.size();
  if (n < 2 || count != n || (int)knots.size() < n) throw std::invalid_argument("PiecewiseCubic: at least two knots, one value per break");
  dim_ = (int)knots[0].size();
  y_.resize((size_t)n * dim_);
  for (int i = 0; i < n; ++i) {
    if ((int)knots[i].size() != dim_) throw std::invalid_argument("PiecewiseCubic: knots of different sizes");
    for (int c = 0; c < dim_; ++c) y_[(size_t)i * dim_ + c] = knots[i][c];
  }
  Fit(breaks);
}
void PiecewiseCubic::AssignFlat(const std::vector<double>& breaks, const double* knots, int dim) {
  const int n = (int)breaks.size();
  if (n < 2 || dim < 0) throw std::invalid_argument("PiecewiseCubic: at least two knots, one value per break");
  dim_ = dim;
  y_.assign(knots, knots + (size_t)n * dim_);
  Fit(breaks);
}
void PiecewiseCubic::Fit(const std::vector<double>& breaks) {
  const int n = (int)breaks.size();
  for (int i = 0; i + 1 < n; ++i)
    if (!(breaks[i + 1] > breaks[i])) throw std::invalid_argument("PiecewiseCubic: breaks must increase");
  t_ = breaks;
  m_.assign((size_t)n * dim_, 0.0);
  std::vector<double>& h = h_;
  h.resize(n - 1);
  for (int i = 0; i + 1 < n; ++i) h[i] = breaks[i + 1] - breaks[i];
  auto slope = [&](int i, int c) { return (y_[(size_t)(i + 1) * dim_ + c] - y_[(size_t)i * dim_ + c]) / h[i]; };
  if (n == 2) {   // the line
    for (int c = 0; c < dim_; ++c) m_[c] = m_[dim_ + c] = slope(0, c);
    return;
  }
  if (n == 3) {   // the parabola through the three points
    for (int c = 0; c < dim_; ++c) {
      const double d0 = slope(0, c), d1 = slope(1, c), a2 = (d1 - d0) / (h[0] + h[1]);
      m_[c] = d0 - a2 * h[0];
      m_[dim_ + c] = d0 + a2 * h[0];
      m_[2 * dim_ + c] = d1 + a2 * h[1];
    }
    return;
  }
  // The not-a-knot system for the knot derivatives is TRIDIAGONAL as it stands - row 0 = [h1, h0 + h1], rows i =
  // [h_i, 2 (h_{i-1} + h_i), h_{i-1}], row n - 1 = [h_{n-2} + h_{n-3}, h_{n-3}] - and one elimination without pivoting
  // is stable on it: after row 0 (multiplier 1) row 1's diagonal is h0 + h1 > h0 and the interior rows are diagonally
  // dominant.  O(n dim) instead of the dense LU's O(n^3) that ran three times per re-plan.
  std::vector<double>&lo = lo_, &di = di_, &up = up_, &B = B_;
  lo.assign(n, 0.0); di.assign(n, 0.0); up.assign(n, 0.0); B.assign((size_t)n * dim_, 0.0);
  for (int i = 1; i + 1 < n; ++i) {
    lo[i] = h[i]; di[i] = 2 * (h[i - 1] + h[i]); up[i] = h[i - 1];
    for (int c = 0; c < dim_; ++c) B[(size_t)i * dim_ + c] = 3 * (h[i] * slope(i - 1, c) + h[i - 1] * slope(i, c));
  }
  {
    const double d = h[0] + h[1];
    di[0] = h[1]; up[0] = d;
    for (int c = 0; c < dim_; ++c) B[c] = ((h[0] + 2 * d) * h[1] * slope(0, c) + h[0] * h[0] * slope(1, c)) / d;
    const double e = h[n - 2] + h[n - 3];
    di[n - 1] = h[n - 3]; lo[n - 1] = e;
    for (int c = 0; c < dim_; ++c)
      B[(size_t)(n - 1) * dim_ + c] = (h[n - 2] * h[n - 2] * slope(n - 3, c) + (2 * e + h[n - 2]) * h[n - 3] * slope(n - 2, c)) / e;
  }
  for (int i = 1; i < n; ++i) {
    if (di[i - 1] == 0.0) throw std::runtime_error("PiecewiseCubic: singular system");
    const double f = lo[i] / di[i - 1];
    di[i] -= f * up[i - 1];
    for (int c = 0; c < dim_; ++c) B[(size_t)i * dim_ + c] -= f * B[(size_t)(i - 1) * dim_ + c];
  }
  if (di[n - 1] == 0.0) throw std::runtime_error("PiecewiseCubic: singular system");
  for (int i = n - 1; i >= 0; --i)
    for (int c = 0; c < dim_; ++c) {
      const double s = B[(size_t)i * dim_ + c] - (i + 1 < n ? up[i] * m_[(size_t)(i + 1) * dim_ + c] : 0.0);
      m_[(size_t)i * dim_ + c] = s / di[i];
    }
}

VectorXd PiecewiseCubic::value(double t) const {
  VectorXd out((size_t)dim_);
  value(t, &out);
  return out;
}
void PiecewiseCubic::value(double t, VectorXd* out_ptr) const {
  if (t_.empty()) throw std::runtime_error("PiecewiseCubic: empty trajectory");
  const int n = (int)t_.size();
  t = std::min(std::max(t, t_.front()), t_.back());
  int i = (int)(std::upper_bound(t_.begin(), t_.end(), t) - t_.begin()) - 1;
  i = std::min(std::max(i, 0), n - 2);
  const double h = t_[i + 1] - t_[i], s = t - t_[i];
  VectorXd& out = *out_ptr;
  out.resize((size_t)dim_);
  for (int c = 0; c < dim_; ++c) {
    const double y0 = y_[(size_t)i * dim_ + c], y1 = y_[(size_t)(i + 1) * dim_ + c];
    const double m0 = m_[(size_t)i * dim_ + c], m1 = m_[(size_t)(i + 1) * dim_ + c];
    const double d = (y1 - y0) / h;
    const double c2 = (3 * d - 2 * m0 - m1) / h, c3 = (m0 + m1 - 2 * d) / (h * h);
    out[c] = y0 + s * (m0 + s * (c2 + s * c3));
  }
}

// ---- ModelPredictiveController (mpc_controller.cc:13-41)
ModelPredictiveController::ModelPredictiveController(TrajectoryOptimizer<double>* optimizer,
                                                     const TrajectoryOptimizerSolution<double>& warm_start_solution,
                                                     const std::vector<int>& actuated, double replan_period,
                                                     const std::vector<bool>& q_nom_relative_to_q_init)
    : time_step_(optimizer->time_step()),
      num_steps_(optimizer->num_steps() + 1),
      nq_(optimizer->num_positions()),
      nv_(optimizer->num_velocities()),
      optimizer_(optimizer),
      warm_start_(optimizer->CreateWarmStart(warm_start_solution.q)),
      replan_period_(replan_period),
      selector_override_(q_nom_relative_to_q_init) {
  for (int j = 0; j < (int)actuated.size() && j < nv_; ++j)
    if (actuated[j]) actuated_dofs_.push_back(j);
  if (actuated_dofs_.empty())
    for (int j = 0; j < nv_; ++j) actuated_dofs_.push_back(j);
  nu_ = (int)actuated_dofs_.size();
  StoreOptimizerSolution(warm_start_solution, 0.0, &stored_);
  solution_ = warm_start_solution;
}

// UpdateAbstractState (:43-85)
const StoredTrajectory& ModelPredictiveController::UpdateAbstractState(double time, const VectorXd& x0) {
  const SolverParameters& params = optimizer_->params();
  const std::vector<bool>& selector = selector_override_.empty() ? params.q_nom_relative_to_q_init : selector_override_;
  if ((int)selector.size() != nq_)
    throw std::invalid_argument("q_nom_relative_to_q_init must have one entry per position (mpc_controller.cc:45)");
  if ((int)x0.size() != nq_ + nv_) throw std::invalid_argument("state estimate must be [q0; v0]");
  idto_hip_trace_mark("mpc: UpdateAbstractState begins");
  const VectorXd q0(x0.begin(), x0.begin() + nq_), v0(x0.begin() + nq_, x0.end());
  // the initial guess from the stored solution, consistent with the initial condition (:55-58)
  // (the three trajectories below live in the controller: 120 small vectors allocated and freed per re-plan were ~10 us
  // of a 0.2 ms call)
  std::vector<VectorXd>& q_guess = guess_scratch_;
  if ((int)q_guess.size() != num_steps_) q_guess.assign((size_t)num_steps_, VectorXd((size_t)nq_));
  UpdateInitialGuess(stored_, time, &q_guess);
  q_guess[0] = q0;
  warm_start_->set_q(q_guess);
  last_guess_ = q_guess;
  idto_hip_trace_mark("mpc: initial guess from the stored splines");
  // shift the nominal trajectory for some DoFs, if requested (:60-69)
  const ProblemDefinition& prob = optimizer_->prob();
  const VectorXd q0_nom_old = prob.q_nom[0];
  std::vector<VectorXd>& q_nom_new = q_nom_scratch_;
  q_nom_new = prob.q_nom;
  for (VectorXd& qt_nom : q_nom_new)
    for (int i = 0; i < nq_; ++i) qt_nom[i] += (selector[i] ? 1.0 : 0.0) * (q0[i] - q0_nom_old[i]);
  std::vector<VectorXd>& v_nom = v_nom_scratch_;
  v_nom = prob.v_nom;
  optimizer_->UpdateNominalTrajectory(q_nom_new, v_nom);
  idto_hip_trace_mark("mpc: nominal trajectory shifted");
  // solve from the new initial condition (:71-75)
  optimizer_->ResetInitialConditions(q0, v0);
  idto_hip_trace_mark("mpc: initial conditions reset");
  stats_ = TrajectoryOptimizerStats<double>();
  TrajectoryOptimizerSolution<double>& solution = scratch_solution_;   // (keeps the rows' storage of the re-plan before last)
  last_flag_ = optimizer_->SolveFromWarmStart(warm_start_.get(), &solution, &stats_);
  idto_hip_trace_mark("mpc: SolveFromWarmStart returned");
  // (a failed factorisation returns an empty solution - the previous plan stays in force and the caller is
  // told through last_flag(); the reference, a Drake LeafSystem, has nowhere to report it and would throw from
  // StoreOptimizerSolution with the controller's state half updated)
  if (last_flag_ == optimizer::SolverFlag::kFactorizationFailed || (int)solution.q.size() < num_steps_) return stored_;
  std::swap(solution_.q, solution.q); std::swap(solution_.v, solution.v); std::swap(solution_.tau, solution.tau);
  StoreOptimizerSolution(solution_, time, &stored_);
  idto_hip_trace_mark("mpc: solution stored as splines");
  return stored_;
}

// UpdateInitialGuess (:87-97)
void ModelPredictiveController::UpdateInitialGuess(const StoredTrajectory& stored_trajectory, double current_time,
                                                   std::vector<VectorXd>* q_guess) const {
  if ((int)q_guess->size() != num_steps_) throw std::invalid_argument("UpdateInitialGuess: q_guess must have num_steps + 1 entries");
  const double start_time = current_time - stored_trajectory.start_time;
  for (int i = 0; i < num_steps_; ++i) stored_trajectory.q.value(start_time + i * time_step_, &(*q_guess)[i]);   // (in place: no temporary per step)
}

// StoreOptimizerSolution (:99-138)
void ModelPredictiveController::StoreOptimizerSolution(const TrajectoryOptimizerSolution<double>& solution, double start_time,
                                                       StoredTrajectory* stored_trajectory) const {
  if ((int)solution.q.size() < num_steps_ || (int)solution.v.size() < num_steps_ || (int)solution.tau.size() < num_steps_ - 1)
    throw std::invalid_argument("StoreOptimizerSolution: solution shorter than the horizon");
  std::vector<double>& time_steps = times_;
  time_steps.resize((size_t)num_steps_);
  for (int i = 0; i < num_steps_; ++i) time_steps[i] = i * time_step_;
  // control inputs, which are undefined at the last time step (:122-126): u = B^T tau
  u_flat_.resize((size_t)num_steps_ * nu_);
  for (int i = 0; i < num_steps_; ++i) {
    const VectorXd& tau = solution.tau[i == num_steps_ - 1 ? i - 1 : i];
    for (int j = 0; j < nu_; ++j) u_flat_[(size_t)i * nu_ + j] = tau[actuated_dofs_[j]];
  }
  stored_trajectory->start_time = start_time;
  stored_trajectory->q.Assign(time_steps, solution.q, num_steps_);
  stored_trajectory->v.Assign(time_steps, solution.v, num_steps_);
  stored_trajectory->u.AssignFlat(time_steps, u_flat_.data(), nu_);
}

// Interpolator::SendState / SendControl (:163-178)
VectorXd Interpolator::State(const StoredTrajectory& traj, double time) {
  VectorXd x = traj.q.value(time - traj.start_time);
  const VectorXd v = traj.v.value(time - traj.start_time);
  x.insert(x.end(), v.begin(), v.end());
  return x;
}
VectorXd Interpolator::Control(const StoredTrajectory& traj, double time) { return traj.u.value(time - traj.start_time); }

}  // namespace mpc
}  // namespace examples
}  // namespace idto
