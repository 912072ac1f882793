// warm_start.h — mirrors reference optimizer/warm_start.h:23-76: everything that is re-used
// between MPC solves (states, trust-region radius, the last steps).
#pragma once

#include <vector>

#include "idto/optimizer/trajectory_optimizer_state.h"

namespace idto {
namespace optimizer {

class WarmStart {
 public:
  WarmStart(int num_steps, int nq, const std::vector<VectorXd>& q_guess, double Delta0)
      : state(num_steps, nq), scratch_state(num_steps, nq), Delta(Delta0) {
    state.set_q(q_guess);
    dq.assign((std::size_t)(num_steps + 1) * nq, 0.0);
    dqH.assign((std::size_t)(num_steps + 1) * nq, 0.0);
  }
  void set_q(const std::vector<VectorXd>& q_guess) { state.set_q(q_guess); }
  const std::vector<VectorXd>& get_q() const { return state.q(); }

  TrajectoryOptimizerState<double> state;
  TrajectoryOptimizerState<double> scratch_state;
  double Delta;
  VectorXd dq;
  VectorXd dqH;
};

}  // namespace optimizer
}  // namespace idto
