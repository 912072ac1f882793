// problem_definition.h — mirrors reference optimizer/problem_definition.h:24-59
// (same member names and meaning; Eigen types replaced by types.h).
#pragma once

#include <vector>

#include "idto/optimizer/types.h"

namespace idto {
namespace optimizer {

struct ProblemDefinition {
  int num_steps = 0;        // N: the trajectory has N+1 positions q_0..q_N
  VectorXd q_init;          // initial generalized positions
  VectorXd v_init;          // initial generalized velocities
  MatrixXd Qq;              // running cost on positions   (nq x nq)
  MatrixXd Qv;              // running cost on velocities  (nv x nv)
  MatrixXd Qf_q;            // terminal cost on positions
  MatrixXd Qf_v;            // terminal cost on velocities
  MatrixXd R;               // cost on generalized forces  (nv x nv)
  std::vector<VectorXd> q_nom;  // target positions at each of the N+1 steps
  std::vector<VectorXd> v_nom;  // target velocities at each of the N+1 steps
};

}  // namespace optimizer
}  // namespace idto
