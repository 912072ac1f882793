// trajectory_optimizer_solution.h — mirrors reference
// optimizer/trajectory_optimizer_solution.h:16-185 (flags, solution and statistics containers,
// CSV column order of SaveToCsv).
#pragma once

#include <cstdio>
#include <string>
#include <vector>

#include "idto/optimizer/types.h"

namespace idto {
namespace optimizer {

enum SolverFlag { kSuccess, kLinesearchMaxIters, kFactorizationFailed, kMaxIterationsReached };

enum ConvergenceReason : int {
  kNoConvergenceCriteriaSatisfied = 0b000,
  kCostReductionCriterionSatisfied = 0b001,
  kGradientCriterionSatisfied = 0b010,
  kSateCriterionSatisfied = 0b100  // (sic) the reference's spelling
};

inline std::string DecodeConvergenceReasons(ConvergenceReason reason) {
  if (reason == kNoConvergenceCriteriaSatisfied) return "no convergence criterion satisfied";
  std::string s;
  auto add = [&s](const char* t) { s += (s.empty() ? "" : ", "); s += t; };
  if (reason & kCostReductionCriterionSatisfied) add("cost reduction");
  if (reason & kGradientCriterionSatisfied) add("gradient");
  if (reason & kSateCriterionSatisfied) add("state change");
  return s;
}

template <typename T>
struct TrajectoryOptimizerSolution {
  std::vector<std::vector<T>> q;    // N+1 positions
  std::vector<std::vector<T>> v;    // N+1 velocities
  std::vector<std::vector<T>> tau;  // N generalized forces
};

template <typename T>
struct TrajectoryOptimizerStats {
  ConvergenceReason convergence_reason{ConvergenceReason::kNoConvergenceCriteriaSatisfied};
  double solve_time = 0;
  std::vector<double> iteration_times;
  std::vector<T> iteration_costs;
  std::vector<int> linesearch_iterations;
  std::vector<double> linesearch_alphas;
  std::vector<T> trust_region_radii;
  std::vector<T> gradient_norms;
  std::vector<T> q_norms;
  std::vector<T> dq_norms;
  std::vector<T> dqH_norms;
  std::vector<T> trust_ratios;
  std::vector<T> dL_dqs;
  std::vector<T> h_norms;
  std::vector<T> merits;

  void push_data(double iter_time, T iter_cost, int linesearch_iters, double alpha, double delta, T q_norm, T dq_norm,
                 T dqH_norm, T trust_ratio, T grad_norm, T dL_dq, T h_norm, T merit) {
    iteration_times.push_back(iter_time);
    iteration_costs.push_back(iter_cost);
    linesearch_iterations.push_back(linesearch_iters);
    linesearch_alphas.push_back(alpha);
    trust_region_radii.push_back(delta);
    q_norms.push_back(q_norm);
    dq_norms.push_back(dq_norm);
    dqH_norms.push_back(dqH_norm);
    trust_ratios.push_back(trust_ratio);
    gradient_norms.push_back(grad_norm);
    dL_dqs.push_back(dL_dq);
    h_norms.push_back(h_norm);
    merits.push_back(merit);
  }

  bool is_empty() const {
    return iteration_times.empty() && iteration_costs.empty() && linesearch_iterations.empty() &&
           linesearch_alphas.empty() && trust_region_radii.empty() && q_norms.empty() && dq_norms.empty() &&
           dqH_norms.empty() && trust_ratios.empty() && gradient_norms.empty() && dL_dqs.empty() && h_norms.empty() &&
           merits.empty();
  }

  void SaveToCsv(const std::string& fname) const {
    std::FILE* f = std::fopen(fname.c_str(), "w");
    if (!f) return;
    std::fprintf(f, "iter, time, cost, ls_iters, alpha, delta, q_norm, dq_norm, dqH_norm, trust_ratio, grad_norm, "
                    "dL_dq, h_norm, merit\n");
    for (std::size_t i = 0; i < iteration_times.size(); ++i)
      std::fprintf(f, "%zu, %.17g, %.17g, %d, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g\n", i,
                   iteration_times[i], (double)iteration_costs[i], linesearch_iterations[i], linesearch_alphas[i],
                   (double)trust_region_radii[i], (double)q_norms[i], (double)dq_norms[i], (double)dqH_norms[i],
                   (double)trust_ratios[i], (double)gradient_norms[i], (double)dL_dqs[i], (double)h_norms[i],
                   (double)merits[i]);
    std::fclose(f);
  }
};

}  // namespace optimizer
}  // namespace idto
