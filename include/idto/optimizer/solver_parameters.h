// solver_parameters.h — mirrors reference optimizer/solver_parameters.h:14-166 and
// optimizer/convergence_criteria_tolerances.h:13-64: same enumerators, member names and
// defaults.  Members that only drive the reference's plotting/debug output
// (save_contour_data, lineplot_*, print_debug_data, ...) are kept so that configuration
// code compiles unchanged; the device path ignores them.
#pragma once

#include <vector>

namespace idto {
namespace optimizer {

enum LinesearchMethod { kArmijo, kBacktracking };
enum SolverMethod { kLinesearch, kTrustRegion };
enum GradientsMethod { kForwardDifferences, kCentralDifferences, kCentralDifferences4, kAutoDiff, kNoGradients };
enum ScalingMethod { kSqrt, kAdaptiveSqrt, kDoubleSqrt, kAdaptiveDoubleSqrt };

struct ConvergenceCriteriaTolerances {
  double rel_cost_reduction{0.0};
  double abs_cost_reduction{0.0};
  double rel_gradient_along_dq{0.0};
  double abs_gradient_along_dq{0.0};
  double rel_state_change{0.0};
  double abs_state_change{0.0};
};

struct SolverParameters {
  enum LinearSolverType { kDenseLdlt, kPentaDiagonalLu };

  bool check_convergence = false;
  ConvergenceCriteriaTolerances convergence_tolerances;
  SolverMethod method{SolverMethod::kTrustRegion};
  LinesearchMethod linesearch_method{LinesearchMethod::kArmijo};
  int max_iterations{100};
  int max_linesearch_iterations{50};
  GradientsMethod gradients_method{kForwardDifferences};
  LinearSolverType linear_solver{LinearSolverType::kPentaDiagonalLu};
  bool normalize_quaternions{false};
  bool verbose{true};
  bool print_debug_data{false};
  bool debug_compare_against_dense{false};
  bool linesearch_plot_every_iteration{false};
  double contact_stiffness{100};     // N/m
  double dissipation_velocity{0.1};  // m/s
  double stiction_velocity{0.05};    // m/s
  double friction_coefficient{0.5};
  double smoothing_factor{0.1};
  bool save_contour_data{false};
  double contour_q1_min{0.0}, contour_q1_max{1.0}, contour_q2_min{0.0}, contour_q2_max{1.0};
  bool save_lineplot_data{false};
  double lineplot_q_min{0.0}, lineplot_q_max{1.0};
  bool exact_hessian{false};
  bool scaling{true};
  ScalingMethod scaling_method{ScalingMethod::kDoubleSqrt};
  bool equality_constraints{true};
  double Delta0{1e-1};
  double Delta_max{1e5};
  int num_threads{1};  // CPU threads of the reference; the device path does not use it
  std::vector<bool> q_nom_relative_to_q_init;
};

}  // namespace optimizer
}  // namespace idto
