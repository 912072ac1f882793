/* idto_hip.h — C-ABI of libidto_hip.so, the MI355X (gfx950) implementation of
 * IDTO's per-iteration gradient/Hessian path.
 *
 * The reference (ToyotaResearchInstitute/idto) has no FFI layer: this path lives
 * in private C++ methods of `TrajectoryOptimizer<double>` that call Drake.  The
 * entry points below are what a binding for that path replaces, one per
 * reference routine (file:line under /root/reference):
 *
 *   idto_hip_eval_tau        CalcNplus                optimizer/trajectory_optimizer.cc:1633-1647
 *                            CalcVelocities           :178-191
 *                            CalcAccelerations        :193-202
 *                            CalcInverseDynamics      :204-226  (+ :228-245, contact :247-386)
 *                            CalcCost                 :147-176
 *   idto_hip_eval_partials   CalcInverseDynamicsPartialsFiniteDiff :426-563
 *                            CalcVelocityPartials     :962-973
 *   idto_hip_grad_hess       CalcGradient             :1021-1081
 *                            CalcHessian              :1093-1165
 *   idto_hip_factor_solve    PentaDiagonalFactorization::Factorize   optimizer/penta_diagonal_solver.h:124-197
 *                            PentaDiagonalFactorization::SolveInPlace :199-248
 *                            SolveLinearSystemInPlace optimizer/trajectory_optimizer.cc:2077-2096
 *   idto_hip_gn_step         the chain above, i.e. what the first CalcDoglegPoint after an
 *                            accepted step computes (:2108-2140) with scaling and equality
 *                            constraints off — the unit of BASELINE.json's metric.
 *
 * Data contract: fp64 everywhere; trajectories are t-major flat arrays
 * (q: (N+1)*nq, v: (N+1)*nv, a/tau: N*nv); per-timestep blocks are column-major
 * (Eigen's default), stored t-major and contiguous.  Everything stays resident
 * in HBM inside the context between calls; `idto_hip_get` copies out.
 * All functions return 0 on success and a negative code on failure
 * (`idto_hip_last_error()` describes it); nothing throws across the boundary.
 * Calls on one context must come from one thread at a time (the reference's
 * optimizer is likewise single-caller: SURVEY.md §8b "Threading").
 */
#ifndef IDTO_HIP_H_
#define IDTO_HIP_H_

#include "idto_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct idto_hip_ctx idto_hip_ctx;

enum idto_hip_array {
  IDTO_ARR_Q = 0,        /* (N+1)*nq */
  IDTO_ARR_V = 1,        /* (N+1)*nv */
  IDTO_ARR_A = 2,        /* N*nv */
  IDTO_ARR_TAU = 3,      /* N*nv */
  IDTO_ARR_NPLUS = 4,    /* (N+1) blocks nv x nq */
  IDTO_ARR_DTAU_DQM = 5, /* N blocks nv x nq ; block 0 is NaN, block 1 is 0 (inverse_dynamics_partials.h:35-42) */
  IDTO_ARR_DTAU_DQT = 6, /* N blocks ; block 0 is 0 */
  IDTO_ARR_DTAU_DQP = 7, /* N blocks */
  IDTO_ARR_GRADIENT = 8, /* (N+1)*nq */
  IDTO_ARR_H_A = 9,      /* (N+1) blocks nq x nq: two below the diagonal */
  IDTO_ARR_H_B = 10,     /* one below the diagonal */
  IDTO_ARR_H_C = 11,     /* diagonal (symmetric, both triangles filled) */
  IDTO_ARR_STEP = 12,    /* (N+1)*nq : solution of the last factor_solve / gn_step */
  IDTO_ARR_COST = 13,    /* 1 */
  IDTO_ARR_SLAB = 14,    /* N * slab_stride: per tau-index k [dtau_dqm | dtau_dqt | dtau_dqp | tau_k] — the
                            buffer a multi-GPU run all-gathers (contiguous in k) */
  IDTO_ARR_DEBUG = 15,   /* solver cycle stamps (option "solver_debug") */
  IDTO_ARR_HBANDS = 16,  /* the three Hessian bands in one copy: [A | B | C], each (N+6) blocks nq x nq
                            (N+1 used, 5 trailing zero blocks: the solver's prefetch margin) */
  IDTO_ARR_TR_DQ = 17,   /* (N+1)*nq : the step dq of the last idto_hip_tr_trial */
  IDTO_ARR_TR_W = 18,    /* (N+1)*nq : w = D^-1 H^-1 (g + J^T lambda) of the last idto_hip_tr_prepare */
  IDTO_ARR_TR_SCALE = 19, /* (N+1)*nq : scale factors D (CalcScaleFactors) */
  IDTO_ARR_CON_S = 21,      /* n_eq*n_eq + n_eq : S = J H^-1 J^T (column-major) then J H^-1 g, of the last constraint step
                               (overwritten by its factors when idto_hip_constraint_solve factorised it in place) */
  IDTO_ARR_CON_LAMBDA = 22, /* n_eq : the multipliers of the last constraint step */
  IDTO_ARR_ASM_TERMS = 20 /* N*(6 nq^2 + 3 nq (+1)) : per-record assembly products (kernels.h asm_terms_stride) */
};

const char* idto_hip_last_error(void);

/* Positive status of the entry points that synchronise after a factorisation of H
 * (idto_hip_get(IDTO_ARR_STEP), idto_hip_solve_host, idto_hip_constraint_schur / _solve / _step):
 * a pivot of the block factorisation was non-positive, non-finite or cancelled completely, i.e. H
 * is not numerically positive definite.  The reference reports this as
 * PentaDiagonalFactorizationStatus::kFailure (optimizer/penta_diagonal_solver.h:181-185) and the
 * optimizer demands success (optimizer/trajectory_optimizer.cc:2084, :2091); the host-side
 * TrajectoryOptimizer maps it to SolverFlag::kFactorizationFailed. */
#define IDTO_HIP_FACTORIZATION_FAILED 2
/* The solver variants that spread one factorisation over several workgroups (two-sided, nested dissection,
 * pipelined chains, the fused Gauss-Newton launch) need those workgroups resident at the same time.  On a
 * device shared with other kernels that is not guaranteed: every wait between workgroups is bounded (50 ms),
 * a launch whose wait ran out ends normally with this status at the next synchronisation, and the context
 * steps down to a variant with fewer co-resident workgroups (ultimately one, which cannot wait for anybody).
 * idto_hip_get(IDTO_ARR_STEP) repeats the solve itself; from the other entry points the caller repeats the
 * call.  idto_hip_get_option("solver_timeouts") counts the occurrences. */
#define IDTO_HIP_SOLVER_TIMEOUT 3

/* Creates a context on HIP device `device` for the given model, problem and
 * contact parameters (copied). */
int idto_hip_create(const idto_model_t* model, const idto_problem_t* problem,
                    const idto_contact_params_t* contact, int device, idto_hip_ctx** out);
void idto_hip_destroy(idto_hip_ctx* ctx);

/* Batch of `batch` problems of the same model, horizon and time step (different initial
 * conditions, weights, nominal and decision trajectories) resident on one device and advanced by
 * ONE launch per kernel: what a controller does that runs one Gauss-Newton iteration per control
 * tick on several warm-started problems (examples/mpc_controller.cc:43-85, BASELINE config 5), or a
 * sampling-based planner.  Every per-problem array gets a leading batch dimension; idto_hip_set_q
 * / idto_hip_get / idto_hip_set_problem address problem 0, the *_batch forms any problem;
 * idto_hip_eval_tau / eval_partials / grad_hess / factor_solve(NULL) / gn_step work on all problems
 * at once (grid.y = problem).  Entry points with explicit right-hand sides, the trial-point call
 * and the equality-constraint step serve single-problem contexts only (the host-side
 * TrajectoryOptimizer uses those).  idto_hip_create == idto_hip_create_batch(..., 1, ...). */
int idto_hip_create_batch(const idto_model_t* model, const idto_problem_t* problems /* [batch] */,
                          const idto_contact_params_t* contact, int device, int batch, idto_hip_ctx** out);
int idto_hip_batch_size(idto_hip_ctx* ctx);
int idto_hip_set_problem_batch(idto_hip_ctx* ctx, int problem, const idto_problem_t* p);
int idto_hip_set_q_batch(idto_hip_ctx* ctx, const double* q_host /* [batch][(N+1)*nq] */);
int idto_hip_gn_step_batch(idto_hip_ctx* ctx); /* = idto_hip_gn_step: every problem of the batch */
int idto_hip_get_batch(idto_hip_ctx* ctx, int what, int problem, double* host_out);
/* Several arrays with ONE synchronisation: the copies are enqueued side by side into pinned staging memory (what a
 * caller that wants the solution of a solve - q, v, tau, the step - pays for is one wait, not one per array; reference:
 * the fields TrajectoryOptimizer<T>::SolveFromWarmStart hands back, trajectory_optimizer.cc:2627-2650).  Not for
 * IDTO_ARR_STEP (whose idto_hip_get also reports the factorisation's status). */
int idto_hip_get_many(idto_hip_ctx* ctx, int n, const int* what, double* const* host_out);
int idto_hip_solver_status_batch(idto_hip_ctx* ctx, int* failed /* [batch] */);

/* Replaces q_init/v_init/weights/q_nom/v_nom (ResetInitialConditions /
 * UpdateNominalTrajectory, reference trajectory_optimizer.h:429-470). num_steps and
 * time_step must not change. */
int idto_hip_set_problem(idto_hip_ctx* ctx, const idto_problem_t* problem);

/* All work of the context is enqueued on this hipStream_t (default: a private stream). */
int idto_hip_set_stream(idto_hip_ctx* ctx, void* hip_stream);
void* idto_hip_get_stream(idto_hip_ctx* ctx);

/* Restricts eval_partials to tau-indices k in [k_begin, k_end) (multi-GPU t-range
 * sharding); the caller all-gathers IDTO_ARR_SLAB before grad_hess. */
int idto_hip_set_shard(idto_hip_ctx* ctx, int k_begin, int k_end);

/* Multi-GPU iteration (the reference parallelises the same loop over timesteps with
 * OpenMP, optimizer/trajectory_optimizer.cc:455-457, :476): every GPU holds the whole problem, its
 * finite-difference kernel covers a contiguous k-range of the (k, column) perturbation grid, ONE
 * RCCL all-gather (in place: a rank's records already sit at their final offset of the slab)
 * completes dtau/dq on every GPU, and every GPU assembles and solves redundantly - identical
 * bits on all ranks, no second collective.  Two ways to form the communicator:
 *   one process per GPU:  rank 0 calls idto_hip_comm_unique_id and ships the 128 bytes to the
 *     other ranks by any means (MPI, a file, torch.distributed's store); every rank then calls
 *     idto_hip_comm_init(ctx, id, rank, world) - collective, like ncclCommInitRank;
 *   one process, several devices:  idto_hip_comm_init_all(ctxs, n) with one context per device,
 *     then idto_hip_gn_step_multi(ctxs, n).
 * Both set the context's shard to its rank's k-range.  idto_hip_gn_step_sharded =
 * eval_partials (own range) + idto_hip_allgather_slab + grad_hess + factor_solve, all enqueued on
 * the context's stream. */
int idto_hip_comm_unique_id(char* id_out, int bytes /* >= 128 */);
/* The RCCL the process resolved (a host that loaded torch first carries torch's bundled librccl under the
 * same soname): file path and ncclGetVersion() code.  Forming a communicator fails with an error naming both
 * versions when the loaded library's major version is not the one this library was compiled against. */
int idto_hip_rccl_info(char* path_out, int path_cap, int* version_out);
int idto_hip_comm_init(idto_hip_ctx* ctx, const char* unique_id, int rank, int world);
int idto_hip_comm_init_all(idto_hip_ctx** ctxs, int n);
int idto_hip_comm_destroy(idto_hip_ctx* ctx);
int idto_hip_allgather_slab(idto_hip_ctx* ctx);
int idto_hip_gn_step_sharded(idto_hip_ctx* ctx);
int idto_hip_eval_partials_multi(idto_hip_ctx** ctxs, int n); /* fd shards + grouped all-gather */
int idto_hip_gn_step_multi(idto_hip_ctx** ctxs, int n);

int idto_hip_set_q(idto_hip_ctx* ctx, const double* q_host);         /* H2D copy */
int idto_hip_set_q_device(idto_hip_ctx* ctx, const double* q_device); /* D2D copy */

int idto_hip_eval_tau(idto_hip_ctx* ctx);
/* The same for a q whose partials come next (idto_hip_tr_solve, idto_hip_gn_step): v, a, N+, tau, the cost AND the
 * partials from one finite-difference launch (CalcInverseDynamicsPartialsFiniteDiff, optimizer/trajectory_optimizer.cc:426-563,
 * evaluates the nominal point as well); the next idto_hip_eval_partials of this q returns at once. */
int idto_hip_eval_tau_partials(idto_hip_ctx* ctx);

/* One trial point of the trust-region loop in one call and one synchronisation (what
 * CalcTrustRatio, optimizer/trajectory_optimizer.cc:1979-2035, needs at q + dq): uploads q
 * ((N+1)*nq, host), evaluates v, a, tau and the cost L(q) (:147-176), returns tau (N*nv, may be
 * NULL) and the cost to host memory.  Equivalent to set_q + eval_tau + get(TAU) + get(COST). */
int idto_hip_trial_cost(idto_hip_ctx* ctx, const double* q_host, double* tau_host, double* cost_host);
int idto_hip_eval_partials(idto_hip_ctx* ctx);
int idto_hip_grad_hess(idto_hip_ctx* ctx);
/* Factorises the resident Hessian and solves H x = rhs for `nrhs` right-hand
 * sides given as DEVICE pointers (column-major (N+1)*nq each); rhs == NULL means
 * rhs = -gradient, result in IDTO_ARR_STEP. */
int idto_hip_factor_solve(idto_hip_ctx* ctx, const double* rhs_device, int nrhs, double* x_device);
int idto_hip_gn_step(idto_hip_ctx* ctx);
/* Same solve with HOST right-hand sides / solutions ((N+1)*nq doubles each, nrhs of them,
 * contiguous), staged through buffers the context owns; synchronises.  This is how the
 * host-side trust-region loop obtains H^-1 [J^T | g] for CalcLagrangeMultipliers
 * (optimizer/trajectory_optimizer.cc:1371-1396) and CalcDoglegPoint (:2108-2202). */
int idto_hip_solve_host(idto_hip_ctx* ctx, const double* rhs_host, int nrhs, double* x_host);
/* SolveLinearSystemInPlace with SolverParameters::linear_solver = kDenseLdlt (reference
 * optimizer/trajectory_optimizer.cc:2088-2093: H.MakeDense().ldlt().solve(b)): the resident Hessian
 * as a dense (N+1)*nq square matrix, a blocked LDL^T of it on the device (csrc/dense_ldl.h) and the
 * solution of H x = rhs for ONE host right-hand side; synchronises.  The reference's cross-check of
 * the block Thomas solver (debug_compare_against_dense, :2142-2150) is this call next to
 * idto_hip_solve_host.  IDTO_HIP_FACTORIZATION_FAILED when a pivot is not positive and finite. */
int idto_hip_solve_dense_ldlt(idto_hip_ctx* ctx, const double* rhs_host, double* x_host);
long idto_hip_dense_solve_count(void);   /* calls of the above in this process (the tests' proof of which branch ran) */

/* Equality-constraint step of the trust-region iteration, kept on the device (reference
 * CalcEqualityConstraintJacobian / CalcLagrangeMultipliers, optimizer/trajectory_optimizer.cc:
 * 1292-1396, and the H^-1 (g + J^T lambda) of CalcDoglegPoint :2139-2149).  The constraint is
 * h(q) = [tau_t[dof] : t < N, dof in `dofs`] (the unactuated dofs, :1267-1290); its Jacobian J
 * consists of rows of the dtau/dq blocks already in device memory.
 *   idto_hip_constraint_schur: after idto_hip_grad_hess, solves H Y = [g | J^T] (n_eq + 1
 *     columns, n_eq = nu * N) and returns S = J H^-1 J^T (n_eq x n_eq, column-major) and
 *     J H^-1 g (n_eq) to host memory; Y stays on the device.
 *   idto_hip_constraint_step: given the multipliers lambda (n_eq, host) returns
 *     H^-1 (g + J^T lambda) and J^T lambda (both (N+1)*nq) to host memory. */
int idto_hip_constraint_schur(idto_hip_ctx* ctx, const int* dofs, int nu, double* S_host, double* Jy_host);
/* Only enqueues the work of idto_hip_constraint_schur (no synchronisation; a no-op if it is
 * already enqueued for the current Hessian); a following idto_hip_constraint_schur with the same
 * dofs waits for it and returns the results. */
int idto_hip_constraint_schur_begin(idto_hip_ctx* ctx, const int* dofs, int nu);
int idto_hip_constraint_step(idto_hip_ctx* ctx, const double* lambda_host, double* step_host, double* jtl_host);
/* The whole multiplier computation on the device, after idto_hip_constraint_schur_begin: uploads
 * the constraint violations h (n_eq), factorises S = J H^-1 J^T there (blocked LDL^T without
 * pivoting, csrc/dense_ldl.h), solves S lambda = h - J H^-1 g and returns lambda (n_eq),
 * H^-1 (g + J^T lambda) and J^T lambda (both (N+1)*nq) - one synchronisation, S never leaves
 * the device.  Returns 1 (outputs untouched) when S is numerically singular (smallest pivot
 * <= 1e-13 x largest): the caller then uses idto_hip_constraint_schur + a pivoted factorisation
 * on the host + idto_hip_constraint_step, as Eigen's ldlt() tolerates semi-definite S. */
int idto_hip_constraint_solve(idto_hip_ctx* ctx, const double* h_host, double* lambda_host, double* step_host,
                              double* jtl_host);

/* Trust-region bookkeeping on the resident arrays: what CalcScaleFactors, the
 * scaled gradient / Hessian products, CalcDoglegPoint and CalcTrustRatio
 * (optimizer/trajectory_optimizer.cc:1181-1255, 2108-2202, 1979-2035) compute from g and H, without g
 * or H leaving the device.
 *   idto_hip_tr_prepare: after idto_hip_gn_step (or, with_lambda = 1, after the constraint step).
 *     scaling_method -1 = no scaling, else SolverParameters::scaling_method.  With D the scale
 *     factors, g~ = D (g + J^T lambda), H~ = D H D and w = D^-1 H^-1 (g + J^T lambda) it returns
 *     out[9] = { g~.g~, g~.H~g~, w.w, g~.w, g~.H~w, w.H~w, q.q, h.h, h.lambda } (synchronises).
 *     The Newton point of the dogleg is pH = -w / Delta, the Cauchy point pU = -(g~.g~ / g~.H~g~) g~ / Delta.
 *   idto_hip_tr_trial: every dogleg step is dq = D (a g~ + b w); forms q + dq (quaternions
 *     normalised on request), evaluates tau and the cost there and returns
 *     out[4] = { dq.dq, g~.(a g~ + b w), L(q + dq), h(q + dq).lambda } (synchronises).
 *     speculate_scaling_method >= -1 (else -2): right behind the trial point's evaluation the NEXT
 *     iteration on q + dq is enqueued too - idto_hip_gn_step and idto_hip_tr_prepare with that scaling
 *     method - while the host is still waiting for the trial point's cost: when the step is accepted
 *     (idto_hip_tr_accept) the following idto_hip_gn_step / idto_hip_tr_prepare return what is already
 *     there; when it is rejected (idto_hip_tr_reject) g, H and the Newton step of q have been
 *     overwritten and must be recomputed.  (Not for the adaptive scaling methods, which update D in
 *     place; then the call behaves as without speculation.)
 *   idto_hip_tr_accept: q + dq becomes the resident q (its v, a, tau, cost are already resident).
 *   idto_hip_tr_reject: q stays; v, a, tau in device memory belong to the dropped trial point. */
/* The unactuated dofs (rows of the actuation matrix that are zero, TO.cc:63-72): h.h of
 * idto_hip_tr_prepare is reported for them even when the constraints are not enforced (the
 * reference logs |h| in every iteration, TO.cc:2509, :2586-2598). */
int idto_hip_set_unactuated_dofs(idto_hip_ctx* ctx, const int* dofs, int nu);
int idto_hip_tr_prepare(idto_hip_ctx* ctx, int scaling_method, int with_lambda, double* out_host /* [9] */);
int idto_hip_tr_trial(idto_hip_ctx* ctx, double a, double b, int scaling, int normalize_quaternions, int with_lambda,
                      int speculate_scaling_method, double* out_host /* [4] */);
int idto_hip_tr_accept(idto_hip_ctx* ctx);
int idto_hip_tr_reject(idto_hip_ctx* ctx);
/* The adaptive scalings (kAdaptiveSqrt / kAdaptiveDoubleSqrt, reference optimizer/trajectory_optimizer.cc:1241-1255)
 * take the minimum of the new scale factors and the previous ones, which the reference keeps in the state's
 * cache: ones in a fresh state (`Solve`), the last solve's in a WarmStart (`SolveFromWarmStart`).  This sets
 * that memory: D_prev_host[(N+1)*nq], or NULL for ones.  After a solve IDTO_ARR_TR_SCALE holds the last D. */
int idto_hip_tr_set_scale_memory(idto_hip_ctx* ctx, const double* D_prev_host);

/* The whole trust-region loop without the host (reference optimizer/trajectory_optimizer.cc:2495-2625
 * for the case the stepwise calls above serve: no enforced constraints, no convergence checks).
 * `iterations` iterations are enqueued back to back - per iteration idto_hip_gn_step, one launch
 * for scale factors / inner products / dogleg point (CalcDoglegPoint :2108-2202) / trial point, the
 * inverse dynamics and the cost there, whose kernel also forms the trust ratio (:1979-2035), accepts
 * (q <- q + dq on the device) or rejects, and updates the radius (:2614-2622) - and the host waits
 * once.  The iterate q must be resident with its cost evaluated (idto_hip_set_q + idto_hip_eval_tau).
 * rows_host[iterations][IDTO_TR_ROW]: per iteration
 *   [0] L(q_k) [1] Delta_k [2] rho [3] |q| [4] |dq| [5] |dqH| [6] |g~| [7] dL/dq [8] |h| [9] accepted
 *   [10] device clock at the decision (100 MHz ticks) [11] a [12] b (dq = D (a g~ + b w)) [13] L(q_k + dq)
 *   [14] flags: 1 dogleg quadratic has no root in (0,1), 2 step not finite, 4 step is not a descent
 *   direction (where the reference throws), 16 a convergence criterion held (idto_hip_tr_set_convergence), 32 the
 *   factorisation behind this iteration's step met a bad pivot; once a flag is set the remaining iterations are idle.
 *   [15] the merit function L(q_k) + h(q_k).lambda_k.  [16] see idto_hip_tr_set_convergence.  Flag 8: the constraints' Schur complement is
 *   numerically singular (redundant constraints; the host's pivoted LDL^T copes, the single-workgroup
 *   solve does not): continue from the iterate with the stepwise calls.
 * constrained_dofs / nu: the unactuated degrees of freedom whose tau is constrained to zero
 * (CalcEqualityConstraintViolations, TO.cc:1257-1290), nu = 0: none enforced.  With nu > 0 every iteration
 * also runs H^-1 [g | J^T], S = J H^-1 J^T, lambda = S^-1 (h - J H^-1 g) (TO.cc:1371-1396; one workgroup for
 * nu * num_steps <= 128, the blocked factorisation above) and the step H^-1 (g + J^T lambda) on the device, and
 * the ratio uses the merit function.
 * scaling_method: -1 none, else ScalingMethod (solver_parameters.h:52-62): 0 kSqrt, 1 kAdaptiveSqrt, 2 kDoubleSqrt,
 * 3 kAdaptiveDoubleSqrt; the adaptive methods' memory of D is idto_hip_tr_set_scale_memory's (a rejected step does
 * not advance it: g and H stay, min(D, f(diag H)) = D).
 * On return q, v, a, tau, N+ in device memory are those of the final iterate when the last step was
 * accepted; after a rejected last step v, a, tau belong to the dropped trial point.
 * Returns IDTO_HIP_FACTORIZATION_FAILED when any iteration's factorisation failed. */
#define IDTO_TR_ROW 17
/* VerifyConvergenceCriteria (reference trajectory_optimizer.cc:2654-2689) inside idto_hip_tr_solve.
 * tolerances: {rel_cost_reduction, abs_cost_reduction, rel_gradient_along_dq, abs_gradient_along_dq, rel_state_change,
 * abs_state_change} (solver_parameters.h:17-31), or NULL: no checks (the default).  With tolerances set, row k of
 * idto_hip_tr_solve carries in [16] the ConvergenceReason bitmask of the step accepted in iteration k (1 cost reduction,
 * 2 gradient along dq, 4 state change; 0: none, or the step was rejected); the first row with a non-zero mask is the
 * last iteration that ran - the ones behind it are idle, flag 16 - and q on the device is its iterate. */
int idto_hip_tr_set_convergence(idto_hip_ctx* ctx, const double* tolerances);
int idto_hip_tr_solve(idto_hip_ctx* ctx, int iterations, int scaling_method, int scaling, int normalize_quaternions,
                      double Delta0, double Delta_max, double eta, const int* constrained_dofs, int nu,
                      double* rows_host, double* Delta_out);

/* idto_hip_tr_solve that also returns what the caller fetches next anyway - the iterate q ((N+1) nq), its v ((N+1) nv) and
 * tau (N nv), the last step dq and w = H^-1 (g + J^T lambda) ((N+1) nq each; any pointer may be NULL) - gathered on the
 * device into one buffer and copied under the solve's own wait (the solution fetch of
 * TrajectoryOptimizer::SolveFromWarmStart, optimizer/trajectory_optimizer.cc:2627-2640, without a second round trip).
 * v and tau are the iterate's when the cost weights are diagonal (two output sets) or the last trial point was
 * accepted (rows[last][9] != 0); with dense weights after a rejected last step they are the trial point's: evaluate
 * tau again then, as after idto_hip_tr_solve. */
int idto_hip_tr_solve_fetch(idto_hip_ctx* ctx, int iterations, int scaling_method, int scaling, int normalize_quaternions,
                            double Delta0, double Delta_max, double eta, const int* constrained_dofs, int nu,
                            double* rows_host, double* Delta_out, double* q_out, double* v_out, double* tau_out,
                            double* dq_out, double* w_out);


/* The same loop for every problem of a batch context at once (idto_hip_create_batch): one launch set per iteration
 * with grid.y = problem, one host thread, one wait - the call pattern of an MPC server that advances several
 * warm-started problems per tick (reference examples/mpc_controller.cc:43-85, BASELINE config 5).  Every problem keeps
 * its own radius, accepts or rejects on its own, and idles on its own flags; no equality constraints (nu = 0; with constraints:
 * idto_hip_tr_solve_batch_constrained).
 * Delta0[batch], Delta_out[batch] (may be NULL); rows_host[batch][iterations][IDTO_TR_ROW]: problem b's rows are
 * what idto_hip_tr_solve returns for the same problem in a context of its own - bit for bit when both contexts run the
 * same linear solver, to the solver's round-off otherwise: from 16 block rows on (option "nd_min_rows") a
 * single-problem context takes the scalar band factorisation for blocks up to 5, while a batch of more than two
 * problems keeps the block kernels; option "solver_band" = 0 on both contexts puts them on the same kernel
 * (tests/test_gpu_batch.py does that for the spinner).  Every problem's q must be resident with its cost evaluated
 * (idto_hip_set_q_batch + idto_hip_eval_tau). */
int idto_hip_tr_solve_batch(idto_hip_ctx* ctx, int iterations, int scaling_method, int scaling, int normalize_quaternions,
                            const double* Delta0, double Delta_max, double eta, double* rows_host, double* Delta_out);
/* ... with ENFORCED equality constraints on the `nu` degrees of freedom `constrained_dofs` (the unactuated ones:
 * h = tau[unactuated] = 0, reference TO.cc:1267-1396; BASELINE config 5's own YAML enforces them,
 * examples/allegro_hand/allegro_hand.yaml:95).  With the banded KKT step (nq + nu <= 30: every example) the constrained
 * iteration is one launch set for the whole batch, grid.y = problem, like idto_hip_tr_solve_batch; otherwise (option
 * con_kkt = 0) the multiplier chain is single-problem launches and every problem is advanced in a single-problem context
 * of its own (created on first use inside this context) on its own stream and host thread.  Either way rows and iterates
 * are those of idto_hip_tr_solve on the same problem alone (bit for bit under the same linear solver: see
 * idto_hip_tr_solve_batch).  nu = 0 forwards to idto_hip_tr_solve_batch.
 * Same residency requirement: every problem's q set (idto_hip_set_q_batch). */
int idto_hip_tr_solve_batch_constrained(idto_hip_ctx* ctx, int iterations, int scaling_method, int scaling,
                                        int normalize_quaternions, const double* Delta0, double Delta_max, double eta,
                                        const int* constrained_dofs, int nu, double* rows_host, double* Delta_out);

/* Options: "gradients_method" = 0 forward differences (default), 1 / 2 central differences of
 * 2nd / 4th order (SolverParameters::gradients_method, reference solver_parameters.h:26-50,
 * trajectory_optimizer.cc:565-885); 3 (autodiff) is refused.  "reference_solver" = 1 selects the bit-exact restatement of the reference's
 * pivoted-LU block Thomas (slow) instead of the banded block LDL^T solver (default 0; the
 * environment variable IDTO_SOLVER_REFERENCE=1 sets it at creation); "two_sided" = 0 keeps the
 * LDL^T solver on one workgroup (default 1: two workgroups eliminate from both ends of the
 * horizon); "fused" = 0 makes idto_hip_gn_step three launches
 * (fd, assemble, solve) instead of one persistent launch whose workgroups take the three roles and
 * synchronise through device-memory counters (default 1; single-problem contexts with diagonal
 * weights and the reference's example models; csrc/fused.h); "solver_nd" = 0 keeps the single-right-hand-side solve on the
 * two-workgroup kernel instead of the nested-dissection form (csrc/penta_nd.h: four chain workgroups,
 * two spike workgroups and a separator; default 1, used for block sizes 2 / 3 / 5 / 19 and at least 24
 * block rows; explicit multi-right-hand-side solves always use the two-workgroup factors);
 * "solver_pipe" = 0 keeps the nested dissection on the seven-workgroup kernel instead of the pipelined chains
 * (csrc/penta_pipe.h: five workgroups, block sizes up to 20; default 1); "solver_band": the small models' scalar band
 * factorisation in one workgroup, which takes the pipelined chains' place (csrc/penta_band.h; 0 off, 1 = blocks up to 4 - in a
 * batch context up to 2 -, the default, 2 = up to 5, batches included; IDTO_SOLVER_BAND overrides it at creation); "asm_in_solver" = 0 gives the assembly of
 * idto_hip_gn_step / of the trust-region loop a launch of its own instead of workgroups of the pipelined solver's
 * launch (default 1); a launch whose workgroups were not co-resident steps these down by itself (IDTO_HIP_SOLVER_TIMEOUT);
 * "gn_small" = 0 keeps the small all-revolute models (acrobot, spinner) on fd_kernel + the band solver's launch instead of the
 * one-workgroup step (csrc/gn_small.h; default 1); inside idto_hip_tr_solve: "tr_small" = 0 keeps fd_kernel, cost_kernel and
 * the solver's launch per iteration for them, "tr_fold" = 0 keeps tr_iter_kernel a launch of its own in front of the
 * one-workgroup launch (defaults 1: a whole trust-region iteration of a small model, its enforced constraint included, is
 * ONE launch), "kkt_fold" = 0 takes the banded KKT step's solution apart in a launch of its own (kkt_extract_kernel) instead of
 * inside tr_iter_kernel, "kkt_in_asm" = 0 builds the KKT system in a launch of its own (kkt_build_kernel) instead of
 * having the gated assembly write it along, "decide_in_solver" = 0 keeps cost_kernel a launch of its own in front of the
 * pipelined solver's instead of one more workgroup of that launch; all of these leave every result bit for bit as it is (tests/test_gpu_small.py,
 * tests/test_gpu_trust_region.py);
 * "solver_debug" = 1 records per-phase cycle stamps (IDTO_ARR 15, tools/solver_phases.py);
 * "asm_stop" truncates the assembly kernel after a phase (tools/asm_phases.py). */
int idto_hip_set_option(idto_hip_ctx* ctx, const char* name, int value);
/* Reads an option back; additionally "last_solver": which factorisation the last solve used
 * (1 two-workgroup block LDL^T, 2 nested dissection over seven workgroups, 3 reference-order LU, 4 nested dissection
 * with pipelined chains, 5 inside the fused launch, 6 the scalar band factorisation). */
int idto_hip_get_option(idto_hip_ctx* ctx, const char* name, int* value);

/* Device-side timing of the last `idto_hip_gn_step`-shaped launches: average
 * milliseconds per launch of kernel `which` (0 fd, 1 assemble, 2 factor_solve, 3 the fused
 * single-launch iteration of idto_hip_gn_step)
 * over the launches recorded since idto_hip_timing_reset, measured with HIP
 * events on the context's stream.  enable = 1 times every launch, enable = s > 1 every
 * s-th launch (an event pair costs ~4 us of stream time: sampling keeps the timed
 * region representative), 0 switches timing off. */
int idto_hip_timing_enable(idto_hip_ctx* ctx, int enable);
int idto_hip_timing_reset(idto_hip_ctx* ctx);
int idto_hip_timing_get(idto_hip_ctx* ctx, int which, double* avg_ms, int* launches);

int idto_hip_sync(idto_hip_ctx* ctx);
/* Synchronises and reports the status of the most recent factorisation launched on this context
 * (idto_hip_factor_solve / idto_hip_gn_step / the constraint step): *failed = 1 if it met a bad
 * pivot (see IDTO_HIP_FACTORIZATION_FAILED), else 0; *failed_rows_total (may be NULL) counts the
 * failing block rows since the context was created. */
int idto_hip_solver_status(idto_hip_ctx* ctx, int* failed, int* failed_rows_total);
/* Enqueues, behind the work submitted so far, an asynchronous copy of array `what` (a
 * contiguous one: not TAU / DTAU_*) to pinned staging memory on a side stream.  The next
 * idto_hip_get of the same array waits for that copy only - not for kernels launched after
 * the prefetch (how the host loop reads g and the Hessian bands while the solver runs).  A
 * pending prefetch is dropped if the array is recomputed before it is read. */
int idto_hip_prefetch(idto_hip_ctx* ctx, int what);
/* Synchronises and copies array `what` to host memory (sizes above). */
int idto_hip_get(idto_hip_ctx* ctx, int what, double* host_out);
/* Raw device pointer / element count of a resident array (for zero-copy interop).  Handing out a
 * Hessian band or the slab tells the context that the caller may write it (the solver then takes H
 * as a general matrix, the assembly reads the slab instead of fd_kernel's products).  The pointers of
 * IDTO_ARR_V / A / NPLUS / SLAB / ASM_TERMS are stable between calls EXCEPT across idto_hip_tr_solve,
 * which keeps two sets of them and may leave the iterate's in the other one: ask again afterwards. */
void* idto_hip_device_ptr(idto_hip_ctx* ctx, int what);
long idto_hip_array_size(idto_hip_ctx* ctx, int what);
int idto_hip_slab_stride(idto_hip_ctx* ctx);

/* Self-test of device arithmetic the bit-exactness argument relies on: evaluates
 * sqrt, division and idto::detmath on `n` inputs on the device (outputs to host arrays). */
int idto_hip_math_probe(int device, const double* x, int n, double* sqrt_out, double* recip_out,
                        double* sin_out, double* cos_out, double* exp_out, double* log_out);

/* Host-side timeline: wall-clock marks of what the calling thread does inside and between the entry points above
 * (tools/host_profile.py --mpc: one MPC re-plan, reference examples/mpc_controller.cc:43-85, accounted for step by step).
 * enable(1) clears the marks and starts the clock; mark(label) appends "<us since enable> label"; dump writes the lines
 * into out (NUL-terminated, truncated to cap) and returns the bytes the whole text needs.  Not thread-safe: a
 * measurement aid for one host thread. */
void idto_hip_trace_enable(int on);
void idto_hip_trace_mark(const char* label);
int idto_hip_trace_dump(char* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* IDTO_HIP_H_ */
