// trajectory_optimizer_state.h — decision variables plus the lazily evaluated cache,
// the role of reference optimizer/trajectory_optimizer_state.h:38-351: any write to q
// invalidates everything (:333-350), every Eval* of the optimizer fills what is missing.
// The cache holds HOST copies of device results; the heavy quantities are produced by
// libidto_hip.so (include/idto_hip.h).
#pragma once

#include <cmath>
#include <vector>

#include "idto/optimizer/penta_diagonal_matrix.h"
#include "idto/optimizer/types.h"

namespace idto {
namespace optimizer {

// reference optimizer/inverse_dynamics_partials.h:21-85 (blocks nv x nq, column-major;
// dtau_dqm[0] is NaN, dtau_dqm[1] and dtau_dqt[0] are zero)
template <typename T>
struct InverseDynamicsPartials {
  std::vector<MatrixXd> dtau_dqm, dtau_dqt, dtau_dqp;  // N blocks each
};
// reference optimizer/velocity_partials.h:20-40
template <typename T>
struct VelocityPartials {
  std::vector<MatrixXd> dvt_dqt, dvt_dqm;  // N+1 blocks each; dvt_dqm[0] unused
};

template <typename T>
class TrajectoryOptimizer;

template <typename T>
class TrajectoryOptimizerState {
 public:
  TrajectoryOptimizerState(int num_steps, int nq) : q_((std::size_t)num_steps + 1, std::vector<T>(nq, T(0))) {}
  TrajectoryOptimizerState(TrajectoryOptimizerState&&) = default;  // move-only like the reference (:207-214)
  TrajectoryOptimizerState& operator=(TrajectoryOptimizerState&&) = default;
  TrajectoryOptimizerState(const TrajectoryOptimizerState&) = delete;
  TrajectoryOptimizerState& operator=(const TrajectoryOptimizerState&) = delete;

  const std::vector<std::vector<T>>& q() const { return q_; }
  void set_q(const std::vector<std::vector<T>>& q) { q_ = q; invalidate_cache(); }
  // q += dq (dq is the stacked vector of all decision variables)
  void AddToQ(const std::vector<T>& dq) {
    const std::size_t nq = q_[0].size();
    for (std::size_t t = 0; t < q_.size(); ++t)
      for (std::size_t i = 0; i < nq; ++i) q_[t][i] += dq[t * nq + i];
    invalidate_cache();
  }
  T norm() const {
    T s = 0;
    for (const auto& qt : q_)
      for (T x : qt) s += x * x;
    return std::sqrt(s);
  }

 private:
  friend class TrajectoryOptimizer<T>;
  struct Cache {
    bool traj = false, kin = false, vpart = false, deriv = false, grad = false, hess = false, scale = false, shess = false, sgrad = false,
         h = false, J = false, lambda = false, merit = false, mgrad = false, hinv = false, uploaded = false,
         step_on_device = false;  // -H^-1 g of this state was launched with the assembly
    std::vector<std::vector<T>> v, a, tau;
    std::vector<MatrixXd> nplus;
    T cost = 0;
    InverseDynamicsPartials<T> id_partials;
    VelocityPartials<T> v_partials;
    std::vector<T> gradient, scale_factors, scaled_gradient, h_viol, lambda_v, merit_gradient;
    PentaDiagonalMatrix<T> hessian, scaled_hessian;
    MatrixXd J_unscaled, J_scaled;  // num_eq x num_vars; J~ = J D when scaling is on
    std::vector<T> Hinv_gm;         // H^-1 (g + J^T lambda) with the unscaled H and J (H^-1 g without constraints)
    std::vector<T> JT_lambda;       // J^T lambda (unscaled J)
    T merit_v = 0;
  };
  void invalidate_cache() {
    const std::vector<T> keep = std::move(cache_.scale_factors);  // adaptive scaling methods reuse the last D
    cache_ = Cache();
    cache_.scale_factors = keep;
  }
  std::vector<std::vector<T>> q_;
  mutable Cache cache_;
};

}  // namespace optimizer
}  // namespace idto
