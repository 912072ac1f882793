// idto_opt_c.cc — extern "C" wrapper of idto::optimizer::TrajectoryOptimizer<double>
// (include/idto_opt.h).  Converts flat arrays <-> the C++ containers and exceptions -> codes.
#include "idto_opt.h"

#include <cstring>
#include <memory>
#include <string>

#include "idto/examples/mpc_controller.h"
#include "idto/optimizer/trajectory_optimizer.h"

using namespace idto::optimizer;

struct idto_opt {
  std::unique_ptr<TrajectoryOptimizer<double>> to;
  int nq = 0, nv = 0, N = 0;
};
struct idto_opt_warm_start {
  std::unique_ptr<WarmStart> ws;
};
struct idto_mpc {
  std::unique_ptr<idto::examples::mpc::ModelPredictiveController> mpc;
  idto_opt* opt = nullptr;
};

namespace {
thread_local std::string g_err;

std::vector<VectorXd> Rows(const double* flat, int count, int width) {
  std::vector<VectorXd> out((size_t)count, VectorXd((size_t)width));
  for (int t = 0; t < count; ++t) std::memcpy(out[t].data(), flat + (size_t)t * width, sizeof(double) * width);
  return out;
}
void Flat(const std::vector<VectorXd>& rows, double* out) {
  if (!out) return;
  size_t o = 0;
  for (const auto& r : rows) { std::memcpy(out + o, r.data(), sizeof(double) * r.size()); o += r.size(); }
}
void Flat(const VectorXd& v, double* out) {
  if (out) std::memcpy(out, v.data(), sizeof(double) * v.size());
}
MatrixXd Mat(const double* data, int n) {
  MatrixXd m(n, n);
  std::memcpy(m.data(), data, sizeof(double) * n * n);
  return m;
}
void FillStats(const TrajectoryOptimizerStats<double>& s, idto_stats_t* out) {
  if (!out) return;
  out->solve_time = s.solve_time;
  const int n = std::min<int>(out->capacity, (int)s.iteration_times.size());
  out->count = n;
  out->total = (int)s.iteration_times.size();
  for (int i = 0; i < n; ++i) {
    out->iteration_times[i] = s.iteration_times[i];
    out->iteration_costs[i] = s.iteration_costs[i];
    out->linesearch_iterations[i] = s.linesearch_iterations[i];
    out->linesearch_alphas[i] = s.linesearch_alphas[i];
    out->trust_region_radii[i] = s.trust_region_radii[i];
    out->q_norms[i] = s.q_norms[i];
    out->dq_norms[i] = s.dq_norms[i];
    out->dqH_norms[i] = s.dqH_norms[i];
    out->trust_ratios[i] = s.trust_ratios[i];
    out->gradient_norms[i] = s.gradient_norms[i];
    out->dL_dqs[i] = s.dL_dqs[i];
    out->h_norms[i] = s.h_norms[i];
    out->merits[i] = s.merits[i];
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
  } catch (...) {
    g_err = "unknown exception";
    return -1;
  }
}
}  // namespace

extern "C" {

const char* idto_opt_last_error(void) { return g_err.c_str(); }

int idto_opt_dense_ldlt_solve(double* S, int n, double* b) {
  if (!S || !b || n < 1) { g_err = "dense_ldlt_solve: bad arguments"; return -1; }
  std::vector<double> M(S, S + (std::size_t)n * n);
  idto::optimizer::internal::DenseLdltSolve(&M, n, b);
  std::copy(M.begin(), M.end(), S);
  return 0;
}

int idto_opt_create(const idto_model_t* model, const idto_problem_t* p, const idto_contact_params_t* c,
                    const idto_solver_params_t* sp, int device, idto_opt** out) {
  return idto_opt_create_multi(model, p, c, sp, &device, 0, out);
}

int idto_opt_create_multi(const idto_model_t* model, const idto_problem_t* p, const idto_contact_params_t* c,
                          const idto_solver_params_t* sp, const int* devices, int ndev, idto_opt** out) {
  return Guard([&] {
    int nq = 0, nv = 0;
    for (int b = 0; b < model->nbodies; ++b) {
      const int jt = model->jtype[b];
      nq += (jt == IDTO_JOINT_FLOATING) ? 7 : (jt == IDTO_JOINT_PLANAR ? 3 : 1);
      nv += (jt == IDTO_JOINT_FLOATING) ? 6 : (jt == IDTO_JOINT_PLANAR ? 3 : 1);
    }
    ProblemDefinition prob;
    prob.num_steps = p->num_steps;
    prob.q_init.assign(p->q_init, p->q_init + nq);
    prob.v_init.assign(p->v_init, p->v_init + nv);
    prob.Qq = Mat(p->Qq, nq); prob.Qv = Mat(p->Qv, nv); prob.Qf_q = Mat(p->Qf_q, nq); prob.Qf_v = Mat(p->Qf_v, nv);
    prob.R = Mat(p->R, nv);
    prob.q_nom = Rows(p->q_nom, p->num_steps + 1, nq);
    prob.v_nom = Rows(p->v_nom, p->num_steps + 1, nv);
    SolverParameters params;
    params.check_convergence = sp->check_convergence != 0;
    params.convergence_tolerances = {sp->rel_cost_reduction, sp->abs_cost_reduction, sp->rel_gradient_along_dq,
                                     sp->abs_gradient_along_dq, sp->rel_state_change, sp->abs_state_change};
    params.method = static_cast<SolverMethod>(sp->method);
    params.linesearch_method = static_cast<LinesearchMethod>(sp->linesearch_method);
    params.max_iterations = sp->max_iterations;
    params.max_linesearch_iterations = sp->max_linesearch_iterations;
    params.gradients_method = static_cast<GradientsMethod>(sp->gradients_method);
    params.linear_solver = static_cast<SolverParameters::LinearSolverType>(sp->linear_solver);
    params.normalize_quaternions = sp->normalize_quaternions != 0;
    params.verbose = sp->verbose != 0;
    params.scaling = sp->scaling != 0;
    params.scaling_method = static_cast<ScalingMethod>(sp->scaling_method);
    params.equality_constraints = sp->equality_constraints != 0;
    params.Delta0 = sp->Delta0;
    params.Delta_max = sp->Delta_max;
    params.num_threads = sp->num_threads;
    params.print_debug_data = sp->print_debug_data != 0;
    params.debug_compare_against_dense = sp->debug_compare_against_dense != 0;
    params.exact_hessian = sp->exact_hessian != 0;
    params.save_contour_data = sp->plot_dumps != 0;
    params.contact_stiffness = c->contact_stiffness;
    params.dissipation_velocity = c->dissipation_velocity;
    params.stiction_velocity = c->stiction_velocity;
    params.friction_coefficient = c->friction_coefficient;
    params.smoothing_factor = c->smoothing_factor;
    auto h = std::make_unique<idto_opt>();
    if (ndev <= 0)   // (idto_opt_create: one device, no communicator)
      h->to = std::make_unique<TrajectoryOptimizer<double>>(*model, p->time_step, prob, params, devices[0]);
    else
      h->to = std::make_unique<TrajectoryOptimizer<double>>(*model, p->time_step, prob, params,
                                                            std::vector<int>(devices, devices + ndev));
    h->nq = nq; h->nv = nv; h->N = p->num_steps;
    *out = h.release();
  });
}
void idto_opt_destroy(idto_opt* o) { delete o; }

int idto_opt_num_steps(const idto_opt* o) { return o->to->num_steps(); }
double idto_opt_time_step(const idto_opt* o) { return o->to->time_step(); }
int idto_opt_num_equality_constraints(const idto_opt* o) { return o->to->num_equality_constraints(); }

int idto_opt_solve(idto_opt* o, const double* q_guess, double* sol_q, double* sol_v, double* sol_tau,
                   idto_stats_t* stats, int* flag, int* reason) {
  return Guard([&] {
    TrajectoryOptimizerSolution<double> sol;
    TrajectoryOptimizerStats<double> st;
    ConvergenceReason r = kNoConvergenceCriteriaSatisfied;
    const SolverFlag f = o->to->Solve(Rows(q_guess, o->N + 1, o->nq), &sol, &st, &r);
    Flat(sol.q, sol_q); Flat(sol.v, sol_v); Flat(sol.tau, sol_tau);
    FillStats(st, stats);
    if (flag) *flag = (int)f;
    if (reason) *reason = (int)r;
  });
}

int idto_opt_ws_create(idto_opt* o, const double* q_guess, idto_opt_warm_start** out) {
  return Guard([&] {
    auto w = std::make_unique<idto_opt_warm_start>();
    w->ws = o->to->CreateWarmStart(Rows(q_guess, o->N + 1, o->nq));
    *out = w.release();
  });
}
void idto_opt_ws_destroy(idto_opt_warm_start* w) { delete w; }
int idto_opt_ws_set_q(idto_opt* o, idto_opt_warm_start* w, const double* q) {
  return Guard([&] { w->ws->set_q(Rows(q, o->N + 1, o->nq)); });
}
int idto_opt_ws_get(idto_opt*, idto_opt_warm_start* w, double* q, double* Delta) {
  return Guard([&] {
    Flat(w->ws->get_q(), q);
    if (Delta) *Delta = w->ws->Delta;
  });
}
int idto_opt_ws_solve(idto_opt* o, idto_opt_warm_start* w, double* sol_q, double* sol_v, double* sol_tau,
                      idto_stats_t* stats, int* flag, int* reason) {
  return Guard([&] {
    TrajectoryOptimizerSolution<double> sol;
    TrajectoryOptimizerStats<double> st;
    ConvergenceReason r = kNoConvergenceCriteriaSatisfied;
    const SolverFlag f = o->to->SolveFromWarmStart(w->ws.get(), &sol, &st, &r);
    Flat(sol.q, sol_q); Flat(sol.v, sol_v); Flat(sol.tau, sol_tau);
    FillStats(st, stats);
    if (flag) *flag = (int)f;
    if (reason) *reason = (int)r;
  });
}

int idto_opt_reset_initial_conditions(idto_opt* o, const double* q_init, const double* v_init) {
  return Guard([&] { o->to->ResetInitialConditions(VectorXd(q_init, q_init + o->nq), VectorXd(v_init, v_init + o->nv)); });
}
int idto_opt_update_nominal_trajectory(idto_opt* o, const double* q_nom, const double* v_nom) {
  return Guard([&] { o->to->UpdateNominalTrajectory(Rows(q_nom, o->N + 1, o->nq), Rows(v_nom, o->N + 1, o->nv)); });
}

int idto_opt_eval(idto_opt* o, const double* q, double* cost, double* gradient, double* scaled_gradient,
                  double* scale_factors, double* lambda, double* merit, double* merit_gradient) {
  return Guard([&] {
    TrajectoryOptimizerState<double> s = o->to->CreateState();
    s.set_q(Rows(q, o->N + 1, o->nq));
    if (cost) *cost = o->to->EvalCost(s);
    if (gradient) Flat(o->to->EvalGradient(s), gradient);
    if (scaled_gradient) Flat(o->to->EvalScaledGradient(s), scaled_gradient);
    if (scale_factors) Flat(o->to->EvalScaleFactors(s), scale_factors);
    if (lambda) Flat(o->to->EvalLagrangeMultipliers(s), lambda);
    if (merit) *merit = o->to->EvalMeritFunction(s);
    if (merit_gradient) Flat(o->to->EvalMeritFunctionGradient(s), merit_gradient);
  });
}
int idto_opt_dogleg(idto_opt* o, const double* q, double Delta, double* dq, double* dqH, int* active) {
  return Guard([&] {
    TrajectoryOptimizerState<double> s = o->to->CreateState();
    s.set_q(Rows(q, o->N + 1, o->nq));
    VectorXd a, b;
    const bool act = o->to->CalcDoglegPoint(s, Delta, &a, &b);
    Flat(a, dq); Flat(b, dqH);
    if (active) *active = act ? 1 : 0;
  });
}
int idto_opt_trust_ratio(idto_opt* o, const double* q, const double* dq, double* rho) {
  return Guard([&] {
    TrajectoryOptimizerState<double> s = o->to->CreateState(), scratch = o->to->CreateState();
    s.set_q(Rows(q, o->N + 1, o->nq));
    *rho = o->to->CalcTrustRatio(s, VectorXd(dq, dq + (size_t)(o->N + 1) * o->nq), &scratch);
  });
}


// ---- model-predictive-control shell
int idto_mpc_create(idto_opt* opt, const double* warm_q, const double* warm_v, const double* warm_tau, const int* actuated,
                    const int* q_nom_relative_to_q_init, double replan_period, idto_mpc** out) {
  return Guard([&] {
    TrajectoryOptimizerSolution<double> warm;
    warm.q = Rows(warm_q, opt->N + 1, opt->nq);
    warm.v = Rows(warm_v, opt->N + 1, opt->nv);
    warm.tau = Rows(warm_tau, opt->N, opt->nv);
    std::vector<int> act;
    if (actuated) act.assign(actuated, actuated + opt->nv);
    std::vector<bool> sel((size_t)opt->nq, false);
    if (q_nom_relative_to_q_init)
      for (int i = 0; i < opt->nq; ++i) sel[i] = q_nom_relative_to_q_init[i] != 0;
    auto m = std::make_unique<idto_mpc>();
    m->opt = opt;
    m->mpc = std::make_unique<idto::examples::mpc::ModelPredictiveController>(opt->to.get(), warm, act, replan_period, sel);
    *out = m.release();
  });
}
void idto_mpc_destroy(idto_mpc* mpc) { delete mpc; }
int idto_mpc_num_actuators(const idto_mpc* mpc) { return mpc->mpc->num_actuators(); }
int idto_mpc_update(idto_mpc* mpc, double time, const double* x0, double* q_guess, double* sol_q, double* sol_v, double* sol_tau,
                    double* first_cost, int* flag) {
  return Guard([&] {
    const idto_opt* o = mpc->opt;
    mpc->mpc->UpdateAbstractState(time, VectorXd(x0, x0 + o->nq + o->nv));
    if (q_guess) Flat(mpc->mpc->last_guess(), q_guess);   // (what it used: the stored trajectory shifted to `time`, row 0 = q0)
    const auto& sol = mpc->mpc->last_solution();
    Flat(sol.q, sol_q); Flat(sol.v, sol_v); Flat(sol.tau, sol_tau);
    const auto& st = mpc->mpc->last_stats();
    if (first_cost) *first_cost = st.iteration_costs.empty() ? 0.0 : st.iteration_costs[0];
    if (flag) *flag = (int)mpc->mpc->last_flag();   // (kFactorizationFailed = 2: the outputs are the previous re-plan's)
  });
}
int idto_mpc_state(const idto_mpc* mpc, double time, double* x) {
  return Guard([&] { Flat(idto::examples::mpc::Interpolator::State(mpc->mpc->stored_trajectory(), time), x); });
}
int idto_mpc_control(const idto_mpc* mpc, double time, double* u) {
  return Guard([&] { Flat(idto::examples::mpc::Interpolator::Control(mpc->mpc->stored_trajectory(), time), u); });
}
double idto_mpc_start_time(const idto_mpc* mpc) { return mpc->mpc->stored_trajectory().start_time; }
int idto_mpc_spline_eval(const double* breaks, const double* knots, int n, int dim, const double* times, int nt, double* out) {
  return Guard([&] {
    const idto::examples::mpc::PiecewiseCubic sp(std::vector<double>(breaks, breaks + n), Rows(knots, n, dim));
    for (int i = 0; i < nt; ++i) Flat(sp.value(times[i]), out + (size_t)i * dim);
  });
}
}  // extern "C"
