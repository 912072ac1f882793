// mpc_controller.h — the model-predictive-control shell around TrajectoryOptimizer<double>: the caller of the hot
// path in the reference's closed-loop examples (reference examples/mpc_controller.h:43-214, mpc_controller.cc:13-178).
//
// Same classes, names and behaviour: StoredTrajectory (q, v, u as cubic splines with continuous second
// derivatives), ModelPredictiveController::UpdateAbstractState (time-shifted initial guess from the stored solution,
// q_nom shifted for the DoFs of q_nom_relative_to_q_init, ResetInitialConditions, SolveFromWarmStart, store),
// Interpolator (x(t), u(t)).  What Drake provides is replaced: the LeafSystem ports / abstract state by plain method
// calls (the caller owns the clock), PiecewisePolynomial<double>::CubicWithContinuousSecondDerivatives by
// PiecewiseCubic below (not-a-knot end conditions - Drake's default `periodic_end_condition = false` - and, like
// PiecewisePolynomial::value, evaluation clamped to the knots' time range), plant->MakeActuationMatrix() by the
// model's actuated mask (B^T tau = the actuated components of tau).
// Every solve runs on the device through the optimizer; this shell is O(num_steps) host bookkeeping.
#pragma once

#include <memory>
#include <vector>

#include "idto/optimizer/trajectory_optimizer.h"

namespace idto {
namespace examples {
namespace mpc {

using optimizer::ProblemDefinition;
using optimizer::SolverParameters;
using optimizer::TrajectoryOptimizer;
using optimizer::TrajectoryOptimizerSolution;
using optimizer::TrajectoryOptimizerStats;
using optimizer::VectorXd;
using optimizer::WarmStart;

// A vector-valued cubic spline through (t_i, y_i) with continuous first and second derivatives and not-a-knot end
// conditions (the third derivative is continuous at the second and the second-to-last knot).  Three knots: the
// parabola through them; two: the line.
class PiecewiseCubic {
 public:
  PiecewiseCubic() = default;
  PiecewiseCubic(const std::vector<double>& breaks, const std::vector<VectorXd>& knots);
  // the same through the first `count` rows of `knots` / through row-major values [break][dim], into an existing object
  // (its storage is kept: no allocation when the sizes repeat)
  void Assign(const std::vector<double>& breaks, const std::vector<VectorXd>& knots, int count);
  void AssignFlat(const std::vector<double>& breaks, const double* knots, int dim);
  bool empty() const { return t_.empty(); }
  double start_time() const { return t_.front(); }
  double end_time() const { return t_.back(); }
  int rows() const { return dim_; }
  // the value at t clamped to [start_time, end_time] (drake::trajectories::PiecewisePolynomial::value)
  VectorXd value(double t) const;
  void value(double t, VectorXd* out) const;   // (the same into existing storage)

 private:
  int dim_ = 0;
  std::vector<double> t_;
  std::vector<double> y_, m_;   // [knot][dim]: values and first derivatives at the knots
  std::vector<double> h_, lo_, di_, up_, B_;   // work space of Fit
  void Fit(const std::vector<double>& breaks);
};

// reference examples/mpc_controller.h:43-55
struct StoredTrajectory {
  double start_time{-1.0};   // time (in seconds) at which this trajectory was generated
  PiecewiseCubic q;          // generalized positions
  PiecewiseCubic v;          // generalized velocities
  PiecewiseCubic u;          // control torques
};

// reference examples/mpc_controller.h:59-150
class ModelPredictiveController {
 public:
  // `optimizer` (not owned) solves the problem `prob` with `params` (params.max_iterations = the example's mpc_iters);
  // `actuated[j] != 0` marks the actuated velocities (all of them if empty or all zero: B = I).
  // `q_nom_relative_to_q_init`: if not empty, used instead of the optimizer's params().q_nom_relative_to_q_init (bindings
  // whose parameter struct has no room for a vector: include/idto_opt.h).
  ModelPredictiveController(TrajectoryOptimizer<double>* optimizer, const TrajectoryOptimizerSolution<double>& warm_start_solution,
                            const std::vector<int>& actuated, double replan_period,
                            const std::vector<bool>& q_nom_relative_to_q_init = {});

  double replan_period() const { return replan_period_; }
  int num_actuators() const { return nu_; }
  const StoredTrajectory& stored_trajectory() const { return stored_; }
  const TrajectoryOptimizerStats<double>& last_stats() const { return stats_; }
  const TrajectoryOptimizerSolution<double>& last_solution() const { return solution_; }
  // what the last UpdateAbstractState's SolveFromWarmStart returned; kFactorizationFailed: the stored trajectory and
  // last_solution() are still the previous re-plan's
  optimizer::SolverFlag last_flag() const { return last_flag_; }
  // the initial guess the last UpdateAbstractState used (the stored trajectory shifted to its time, row 0 = q0)
  const std::vector<VectorXd>& last_guess() const { return last_guess_; }

  // UpdateAbstractState (mpc_controller.cc:43-85) at time `time` with the state estimate x0 = [q0; v0]
  const StoredTrajectory& UpdateAbstractState(double time, const VectorXd& x0);
  // StoreOptimizerSolution (:99-138)
  void StoreOptimizerSolution(const TrajectoryOptimizerSolution<double>& solution, double start_time,
                              StoredTrajectory* stored_trajectory) const;
  // UpdateInitialGuess (:87-97)
  void UpdateInitialGuess(const StoredTrajectory& stored_trajectory, double current_time, std::vector<VectorXd>* q_guess) const;

 private:
  const double time_step_;
  const int num_steps_;   // knots: prob.num_steps + 1 (:16)
  const int nq_, nv_;
  int nu_;
  std::vector<int> actuated_dofs_;
  TrajectoryOptimizer<double>* optimizer_;
  std::unique_ptr<WarmStart> warm_start_;
  StoredTrajectory stored_;
  TrajectoryOptimizerStats<double> stats_;
  TrajectoryOptimizerSolution<double> solution_, scratch_solution_;
  mutable std::vector<double> times_, u_flat_;   // work space of StoreOptimizerSolution
  optimizer::SolverFlag last_flag_{optimizer::SolverFlag::kSuccess};
  std::vector<VectorXd> last_guess_;
  std::vector<VectorXd> guess_scratch_, q_nom_scratch_, v_nom_scratch_;   // (UpdateAbstractState's work trajectories: no allocation per re-plan)
  double replan_period_;
  std::vector<bool> selector_override_;
};

// reference examples/mpc_controller.h:155-214: x(t) = [q(t); v(t)] and u(t) of a stored trajectory
struct Interpolator {
  static VectorXd State(const StoredTrajectory& traj, double time);
  static VectorXd Control(const StoredTrajectory& traj, double time);
};

}  // namespace mpc
}  // namespace examples
}  // namespace idto
