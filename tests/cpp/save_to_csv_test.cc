// TrajectoryOptimizerStats::SaveToCsv (include/idto/optimizer/trajectory_optimizer_solution.h) writes the file the
// reference's plotting scripts read: header and column order of optimizer/trajectory_optimizer_solution.h:161-184
// ("iter, time, cost, ls_iters, alpha, delta, q_norm, dq_norm, dqH_norm, trust_ratio, grad_norm, dL_dq, h_norm, merit"),
// one line per iteration, the iteration index first.  Every column gets a value that names it (push_data's argument
// order is the reference's, :124-150): the test that runs this program checks where each value lands.
#include <cstdio>
#include <string>

#include "idto/optimizer/trajectory_optimizer_solution.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  idto::optimizer::TrajectoryOptimizerStats<double> stats;
  if (!stats.is_empty()) return 3;
  for (int i = 0; i < 3; ++i) {
    const double b = 100.0 * i;
    // iter_time, iter_cost, linesearch_iters, alpha, delta, q_norm, dq_norm, dqH_norm, trust_ratio, grad_norm, dL_dq, h_norm, merit
    stats.push_data(b + 1.5, b + 2.5, 3 + i, b + 4.5, b + 5.5, b + 6.5, b + 7.5, b + 8.5, b + 9.5, b + 10.5, b + 11.5, b + 12.5, b + 13.5);
  }
  if (stats.is_empty()) return 4;
  stats.SaveToCsv(argv[1]);
  std::printf("written\n");
  return 0;
}
