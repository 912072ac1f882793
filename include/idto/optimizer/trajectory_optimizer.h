// trajectory_optimizer.h — idto::optimizer::TrajectoryOptimizer<double> on MI355X.
//
// Mirrors the public API of reference optimizer/trajectory_optimizer.h:41-483 (same method
// names, argument meaning, return values and error behaviour) with the Drake handles
// replaced: the constructor takes the model tables (include/idto_model.h) and the time step
// instead of (Diagram*, MultibodyPlant*).  Every Calc* on the hot path runs in
// libidto_hip.so through the C-ABI of include/idto_hip.h:
//
//   CalcCacheTrajectoryData (TO.cc:1463-1480)            -> idto_hip_eval_tau
//   CalcInverseDynamicsPartials (:400-563)               -> idto_hip_eval_partials
//   CalcGradient / CalcHessian (:1021-1165)              -> idto_hip_grad_hess
//   SolveLinearSystemInPlace (:2077-2096), H^-1 J^T of
//   CalcLagrangeMultipliers (:1371-1396)                 -> idto_hip_solve_host
//
// and only the O(num_vars) outer-loop numerics (scaling :1181-1255, equality constraints
// :1267-1456, dogleg :2108-2202, trust ratio :1979-2035, convergence :2653-2689, the
// trust-region loop :2449-2651) run on the host.  There is no CPU implementation of the hot
// path: construction throws std::runtime_error if no HIP device is available.
//
// Differences from the reference, reported through exceptions in the constructor: gradients_method must
// be one of the finite-difference methods (forward, central, central4; kAutoDiff needs Drake's AutoDiffXd
// plant, reference TO.cc:410-423 has the same kind of runtime check), exact_hessian is not supported
// (it needs autodiff as well).  Every other field of SolverParameters is honoured: linear_solver =
// kDenseLdlt routes SolveLinearSystemInPlace (:2088-2093) to a dense LDL^T of MakeDense() on the device
// (idto_hip_solve_dense_ldlt), debug_compare_against_dense prints the reference's "Sparse vs. Dense error"
// per dogleg point (:2142-2150), print_debug_data its condition numbers (:2349-2365, :2499-2507); these
// three take the stepwise loop (one synchronisation per quantity), not the device-resident one.
// The plotting dumps (save_contour_data, save_lineplot_data, linesearch_plot_every_iteration: CSV files for
// the 2-DoF toy examples' figures, TO.cc:1650-1830) are not produced: the constructor throws if one is set.
#pragma once

#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

#include "idto/optimizer/penta_diagonal_matrix.h"
#include "idto/optimizer/problem_definition.h"
#include "idto/optimizer/solver_parameters.h"
#include "idto/optimizer/trajectory_optimizer_solution.h"
#include "idto/optimizer/trajectory_optimizer_state.h"
#include "idto/optimizer/warm_start.h"
#include "idto_hip.h"

namespace idto {
namespace optimizer {

namespace internal {
// S x = b for the symmetric positive (semi-)definite Schur complement of the multipliers
// (column-major, lower triangle read, overwritten): LDL^T with diagonal pivoting (TO.cc:1395).
void DenseLdltSolve(std::vector<double>* S, int n, double* b);
}  // namespace internal

template <typename T>
class TrajectoryOptimizer;

template <>
class TrajectoryOptimizer<double> {
 public:
  using T = double;
  // `model` is copied into the device context; it need not outlive the optimizer.
  TrajectoryOptimizer(const idto_model_t& model, double time_step, const ProblemDefinition& prob,
                      const SolverParameters& params = SolverParameters{}, int device = 0);
  // Several devices of one node (the counterpart of the reference's `num_threads`, which
  // parallelises CalcInverseDynamicsPartialsFiniteDiff over timesteps with OpenMP,
  // optimizer/trajectory_optimizer.cc:455-457, :476): devices[0] hosts the optimizer as above, the
  // others hold a copy of the problem and evaluate their k-range of the finite-difference grid;
  // one RCCL all-gather per evaluation of the partials completes dtau/dq on every device
  // (idto_hip_comm_init_all / idto_hip_eval_partials_multi of include/idto_hip.h; no torch, no MPI).
  TrajectoryOptimizer(const idto_model_t& model, double time_step, const ProblemDefinition& prob,
                      const SolverParameters& params, const std::vector<int>& devices);
  ~TrajectoryOptimizer();
  TrajectoryOptimizer(const TrajectoryOptimizer&) = delete;
  TrajectoryOptimizer& operator=(const TrajectoryOptimizer&) = delete;

  double time_step() const { return time_step_; }
  int num_steps() const { return prob_.num_steps; }
  int num_positions() const { return nq_; }
  int num_velocities() const { return nv_; }
  const std::vector<int>& unactuated_dofs() const { return unactuated_dofs_; }
  int num_equality_constraints() const { return (int)unactuated_dofs_.size() * num_steps(); }
  const SolverParameters& params() const { return params_; }
  const ProblemDefinition& prob() const { return prob_; }

  TrajectoryOptimizerState<T> CreateState() const { return TrajectoryOptimizerState<T>(num_steps(), nq_); }
  std::unique_ptr<WarmStart> CreateWarmStart(const std::vector<VectorXd>& q_guess) const {
    return std::make_unique<WarmStart>(num_steps(), nq_, q_guess, params_.Delta0);
  }

  void CalcGradient(const TrajectoryOptimizerState<T>& state, VectorXd* g) const { *g = EvalGradient(state); }
  void CalcHessian(const TrajectoryOptimizerState<T>& state, PentaDiagonalMatrix<T>* H) const { *H = EvalHessian(state); }

  SolverFlag Solve(const std::vector<VectorXd>& q_guess, TrajectoryOptimizerSolution<T>* solution,
                   TrajectoryOptimizerStats<T>* stats, ConvergenceReason* reason = nullptr) const;
  SolverFlag SolveFromWarmStart(WarmStart* warm_start, TrajectoryOptimizerSolution<T>* solution,
                                TrajectoryOptimizerStats<T>* stats, ConvergenceReason* reason = nullptr) const;

  const std::vector<VectorXd>& EvalV(const TrajectoryOptimizerState<T>& state) const;
  const std::vector<VectorXd>& EvalA(const TrajectoryOptimizerState<T>& state) const;
  const std::vector<VectorXd>& EvalTau(const TrajectoryOptimizerState<T>& state) const;
  const std::vector<MatrixXd>& EvalNplus(const TrajectoryOptimizerState<T>& state) const;
  const VelocityPartials<T>& EvalVelocityPartials(const TrajectoryOptimizerState<T>& state) const;
  const InverseDynamicsPartials<T>& EvalInverseDynamicsPartials(const TrajectoryOptimizerState<T>& state) const;
  T EvalCost(const TrajectoryOptimizerState<T>& state) const;
  const VectorXd& EvalGradient(const TrajectoryOptimizerState<T>& state) const;
  const PentaDiagonalMatrix<T>& EvalHessian(const TrajectoryOptimizerState<T>& state) const;
  const VectorXd& EvalScaleFactors(const TrajectoryOptimizerState<T>& state) const;
  const PentaDiagonalMatrix<T>& EvalScaledHessian(const TrajectoryOptimizerState<T>& state) const;
  const VectorXd& EvalScaledGradient(const TrajectoryOptimizerState<T>& state) const;
  const VectorXd& EvalEqualityConstraintViolations(const TrajectoryOptimizerState<T>& state) const;
  const MatrixXd& EvalEqualityConstraintJacobian(const TrajectoryOptimizerState<T>& state) const;
  const VectorXd& EvalLagrangeMultipliers(const TrajectoryOptimizerState<T>& state) const;
  T EvalMeritFunction(const TrajectoryOptimizerState<T>& state) const;
  const VectorXd& EvalMeritFunctionGradient(const TrajectoryOptimizerState<T>& state) const;

  void ResetInitialConditions(const VectorXd& q_init, const VectorXd& v_init);
  void UpdateNominalTrajectory(const std::vector<VectorXd>& q_nom, const std::vector<VectorXd>& v_nom);

  // pieces of the iteration the reference's tests reach through TrajectoryOptimizerTester
  // (optimizer/test/trajectory_optimizer_test.cc:31-89)
  bool CalcDoglegPoint(const TrajectoryOptimizerState<T>& state, double Delta, VectorXd* dq, VectorXd* dqH) const;
  T CalcTrustRatio(const TrajectoryOptimizerState<T>& state, const VectorXd& dq,
                   TrajectoryOptimizerState<T>* scratch_state) const;

  idto_hip_ctx* device_context() const { return dev(); }

 private:
  int num_vars() const { return (num_steps() + 1) * nq_; }
  void UploadProblem() const;
  // the context, with the problem data on the device up to date (ResetInitialConditions / UpdateNominalTrajectory only
  // mark them: an MPC re-plan calls both, examples/mpc_controller.cc:60-75, and pays for one upload)
  idto_hip_ctx* dev() const { if (problem_dirty_) UploadProblem(); return hip_; }
  // makes the device hold `state`: level 0 q, 1 + tau/cost, 2 + dtau/dq, 3 + gradient/Hessian
  void EnsureDevice(const TrajectoryOptimizerState<T>& state, int level) const;
  std::vector<double> Fetch(int what) const;
  void CalcTrajectoryData(const TrajectoryOptimizerState<T>& state) const;   // tau, cost
  void CalcKinematics(const TrajectoryOptimizerState<T>& state) const;       // v, a, N+
  void CalcVelocityPartials(const TrajectoryOptimizerState<T>& state) const;
  void CalcDerivatives(const TrajectoryOptimizerState<T>& state) const;
  void CalcGradHess(const TrajectoryOptimizerState<T>& state) const;
  const VectorXd& EvalHinvMeritGradient(const TrajectoryOptimizerState<T>& state) const;
  // SolveLinearSystemInPlace (TO.cc:2077-2096) for the Hessian of `state`: params().linear_solver (solver < 0), the
  // dense LDL^T (0) or the block Thomas algorithm (1)
  bool DenseLinearSolver() const;
  void SolveLinearSystem(const TrajectoryOptimizerState<T>& state, const VectorXd& b, VectorXd* x, int solver = -1) const;
  double DebugConditionNumber(const TrajectoryOptimizerState<T>& state, bool scaled) const;
  void NormalizeQuaternions(TrajectoryOptimizerState<T>* state) const;
  std::pair<double, int> ArmijoLinesearch(const TrajectoryOptimizerState<T>& state, const VectorXd& dq,
                                          TrajectoryOptimizerState<T>* scratch) const;
  std::pair<double, int> BacktrackingLinesearch(const TrajectoryOptimizerState<T>& state, const VectorXd& dq,
                                                TrajectoryOptimizerState<T>* scratch) const;
  SolverFlag SolveWithLinesearch(const std::vector<VectorXd>& q_guess, TrajectoryOptimizerSolution<T>* solution,
                                 TrajectoryOptimizerStats<T>* stats) const;
  SolverFlag SolveFromWarmStartImpl(WarmStart* warm_start, TrajectoryOptimizerSolution<T>* solution,
                                    TrajectoryOptimizerStats<T>* stats, ConvergenceReason* reason) const;
  bool DeviceLoopEligible() const;
  bool ResidentLoopEligible() const;
  SolverFlag SolveOnDevice(WarmStart* warm_start, TrajectoryOptimizerSolution<T>* solution,
                           TrajectoryOptimizerStats<T>* stats, ConvergenceReason* reason) const;
  void AdoptTrialPoint(const TrajectoryOptimizerState<T>& scratch, TrajectoryOptimizerState<T>* state) const;
  ConvergenceReason VerifyConvergenceCriteria(const TrajectoryOptimizerState<T>& state, T previous_cost,
                                              const VectorXd& dq) const;
  void Check(int rc) const;

  double time_step_;
  int nq_ = 0, nv_ = 0;
  ProblemDefinition prob_;
  const SolverParameters params_;
  std::vector<int> unactuated_dofs_;
  std::vector<int> quaternion_starts_;
  idto_hip_ctx* hip_ = nullptr;
  std::vector<idto_hip_ctx*> shard_ctx_;    // [hip_, contexts on the other devices] when sharded over devices
  mutable bool problem_dirty_ = false;      // prob_ changed since the last upload
  mutable const void* resident_ = nullptr;  // state whose q is on the device
  mutable int device_level_ = 0;            // what has been evaluated for it there
  // the device-resident loop met a singular constraint Schur complement at iteration resume_k_: the
  // host loop (pivoted LDL^T) finishes the solve from that iterate
  mutable bool force_host_loop_ = false;
  mutable int resume_k_ = 0;
  mutable double last_sparse_vs_dense_ = -1.0;   // debug_compare_against_dense: the figure CalcDoglegPoint printed last
};

}  // namespace optimizer
}  // namespace idto
