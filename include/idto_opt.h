/* idto_opt.h — C-ABI of libidto_opt.so: the host-side TrajectoryOptimizer
 * (include/idto/optimizer/trajectory_optimizer.h, C++) for bindings that cannot consume
 * C++ (ctypes, cgo, JNI ...).  One entry point per public method of the reference class
 * (optimizer/trajectory_optimizer.h:41-483) that the reference's own pybind11 module exports
 * (python_bindings/trajectory_optimizer_py.cc:34-59: constructor, time_step, num_steps, Solve,
 * SolveFromWarmStart, CreateWarmStart, ResetInitialConditions, UpdateNominalTrajectory, params,
 * prob) plus the Eval* accessors the reference's tests use.  All heavy work happens in
 * libidto_hip.so (include/idto_hip.h); return value 0 = ok, negative = failure described by
 * idto_opt_last_error(); C++ exceptions never cross this boundary. */
#ifndef IDTO_OPT_H_
#define IDTO_OPT_H_

#include "idto_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct idto_opt idto_opt;
typedef struct idto_opt_warm_start idto_opt_warm_start;

const char* idto_opt_last_error(void);

/* The small dense solve of CalcLagrangeMultipliers (optimizer/trajectory_optimizer.cc:1395,
 * Eigen ldlt()): S x = b, S symmetric positive (semi-)definite n x n column-major (lower triangle
 * read; overwritten by the factors), b overwritten by x.  Host only; exported for tests. */
int idto_opt_dense_ldlt_solve(double* S, int n, double* b);

/* TrajectoryOptimizer(diagram, plant, prob, params) (trajectory_optimizer.h:56-72) with the
 * plant replaced by model tables + time step. */
int idto_opt_create(const idto_model_t* model, const idto_problem_t* problem, const idto_contact_params_t* contact,
                    const idto_solver_params_t* params, int device, idto_opt** out);
/* The same optimizer sharded over `ndev` >= 1 devices of one node (devices[0] hosts it): the
 * finite-difference grid is split into k-ranges, one RCCL all-gather per evaluation of the
 * partials (idto_hip_comm_init_all / idto_hip_eval_partials_multi). */
int idto_opt_create_multi(const idto_model_t* model, const idto_problem_t* problem, const idto_contact_params_t* contact,
                          const idto_solver_params_t* params, const int* devices, int ndev, idto_opt** out);
void idto_opt_destroy(idto_opt* opt);

int idto_opt_num_steps(const idto_opt* opt);
double idto_opt_time_step(const idto_opt* opt);
int idto_opt_num_equality_constraints(const idto_opt* opt);

/* Solve (trajectory_optimizer.h:176-194).  q_guess: (N+1)*nq; outputs sized (N+1)*nq, (N+1)*nv,
 * N*nv.  *flag receives the SolverFlag, *reason the ConvergenceReason bitmask. */
int idto_opt_solve(idto_opt* opt, const double* q_guess, double* sol_q, double* sol_v, double* sol_tau,
                   idto_stats_t* stats, int* flag, int* reason);

/* CreateWarmStart / SolveFromWarmStart (trajectory_optimizer.h:156-211, warm_start.h:23-76) */
int idto_opt_ws_create(idto_opt* opt, const double* q_guess, idto_opt_warm_start** out);
void idto_opt_ws_destroy(idto_opt_warm_start* ws);
int idto_opt_ws_set_q(idto_opt* opt, idto_opt_warm_start* ws, const double* q);
int idto_opt_ws_get(idto_opt* opt, idto_opt_warm_start* ws, double* q, double* Delta);
int idto_opt_ws_solve(idto_opt* opt, idto_opt_warm_start* ws, double* sol_q, double* sol_v, double* sol_tau,
                      idto_stats_t* stats, int* flag, int* reason);

/* ResetInitialConditions / UpdateNominalTrajectory (trajectory_optimizer.h:429-470) */
int idto_opt_reset_initial_conditions(idto_opt* opt, const double* q_init, const double* v_init);
int idto_opt_update_nominal_trajectory(idto_opt* opt, const double* q_nom, const double* v_nom);

/* Eval* at a given q (a fresh state is created per call): cost, gradient ((N+1)*nq),
 * lagrange multipliers (num_equality_constraints), merit; NULL outputs are skipped. */
int idto_opt_eval(idto_opt* opt, const double* q, double* cost, double* gradient, double* scaled_gradient,
                  double* scale_factors, double* lambda, double* merit, double* merit_gradient);
/* CalcDoglegPoint / CalcTrustRatio (TO.cc:2108-2202, 1979-2035) for the property tests */
int idto_opt_dogleg(idto_opt* opt, const double* q, double Delta, double* dq, double* dqH, int* active);
int idto_opt_trust_ratio(idto_opt* opt, const double* q, const double* dq, double* rho);

/* ---- the model-predictive-control shell (include/idto/examples/mpc_controller.h; reference examples/mpc_controller.h:59-214,
 * mpc_controller.cc:13-178).  The optimizer handed in is the controller's (params.max_iterations = the example's mpc_iters) and
 * must outlive it. */
typedef struct idto_mpc idto_mpc;
/* ModelPredictiveController(diagram, plant, prob, warm_start_solution, params, replan_period) (mpc_controller.cc:13-41).
 * warm_q / warm_v: (N+1)*nq, (N+1)*nv; warm_tau: N*nv; actuated: nv flags (NULL: every DoF); q_nom_relative_to_q_init: nq flags
 * (SolverParameters::q_nom_relative_to_q_init, solver_parameters.h:166; NULL: none). */
int idto_mpc_create(idto_opt* opt, const double* warm_q, const double* warm_v, const double* warm_tau, const int* actuated,
                    const int* q_nom_relative_to_q_init, double replan_period, idto_mpc** out);
void idto_mpc_destroy(idto_mpc* mpc);
int idto_mpc_num_actuators(const idto_mpc* mpc);
/* UpdateAbstractState (mpc_controller.cc:43-85): replan at `time` from the state estimate x0 = [q0; v0]; outputs (any may be
 * NULL): the initial guess that was used ((N+1)*nq), the solution, the cost of the first iteration, the solver flag. */
int idto_mpc_update(idto_mpc* mpc, double time, const double* x0, double* q_guess, double* sol_q, double* sol_v, double* sol_tau,
                    double* first_cost, int* flag);
/* Interpolator::SendState / SendControl (mpc_controller.cc:163-178) on the stored trajectory: x = [q(t); v(t)], u(t) */
int idto_mpc_state(const idto_mpc* mpc, double time, double* x);
int idto_mpc_control(const idto_mpc* mpc, double time, double* u);
double idto_mpc_start_time(const idto_mpc* mpc);
/* PiecewisePolynomial::CubicWithContinuousSecondDerivatives(breaks, knots).value(t) (host only; exported for tests): n knots of
 * `dim` components at `breaks`, evaluated at nt times (clamped to the breaks' range); out: nt*dim. */
int idto_mpc_spline_eval(const double* breaks, const double* knots, int n, int dim, const double* times, int nt, double* out);

#ifdef __cplusplus
}
#endif
#endif /* IDTO_OPT_H_ */
