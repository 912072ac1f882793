// A consumer of the drop-in boundary with NO Python and NO torch in the process (VERDICT r4 "weak" #8, "next" #6): what
// reference examples/acrobot/acrobot.cc:43-50 + examples/example_base.cc:189-334 (SolveTrajectoryOptimization) do with
// Drake - build the plant, fill a ProblemDefinition from the YAML's values, construct TrajectoryOptimizer<double>,
// Solve, report the cost per iteration - written against include/idto/optimizer/*.h and linked with libidto_opt.so /
// libidto_hip.so only.  tests/test_gpu_cpp_consumer.py runs it on the GPU box and holds the printed series to the CPU
// oracle's; the program itself reports which shared objects the process has mapped (no libtorch, no libpython, and no
// librccl either: RCCL is resolved only when a communicator is asked for).
//
// usage: solve_acrobot <path to acrobot.model> [max_iterations]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "idto/model_file.h"
#include "idto/optimizer/trajectory_optimizer.h"

using idto::optimizer::MatrixXd;
using idto::optimizer::ProblemDefinition;
using idto::optimizer::SolverFlag;
using idto::optimizer::SolverParameters;
using idto::optimizer::TrajectoryOptimizer;
using idto::optimizer::TrajectoryOptimizerSolution;
using idto::optimizer::TrajectoryOptimizerStats;
using idto::optimizer::VectorXd;

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: solve_acrobot <acrobot.model> [max_iterations]\n"); return 2; }
  try {
    const idto::ModelFile mf = idto::ModelFile::Load(argv[1]);
    // reference examples/acrobot/acrobot.yaml (= idto_amd/configs/acrobot.yaml), as SetProblemDefinition reads it
    // (examples/example_base.cc:377-426): diagonal weights, the nominal trajectory interpolated from q_nom_start to q_nom_end
    const int N = 40, nq = mf.nq, nv = mf.nv;
    const double dt = 0.05;
    ProblemDefinition prob;
    prob.num_steps = N;
    prob.q_init = {0.0, 0.0};
    prob.v_init = {0.0, 0.0};
    prob.Qq = MatrixXd::Diagonal({1.0, 1.0});
    prob.Qv = MatrixXd::Diagonal({1.0, 1.0});
    prob.R = MatrixXd::Diagonal({1e3, 0.1});
    prob.Qf_q = MatrixXd::Diagonal({100.0, 100.0});
    prob.Qf_v = MatrixXd::Diagonal({1.0, 1.0});
    const VectorXd q_nom_start = {3.1415, 0.0}, q_nom_end = {3.1415, 0.0};
    for (int t = 0; t <= N; ++t) {   // examples/example_base.h:195-205 MakeLinearInterpolation, v_nom by differences (:414-420)
      VectorXd q(nq);
      const double lam = t / ((N + 1) - 1.0);
      for (int i = 0; i < nq; ++i) q[i] = (1 - lam) * q_nom_start[i] + lam * q_nom_end[i];
      prob.q_nom.push_back(q);
    }
    prob.v_nom.push_back(prob.v_init);
    for (int t = 1; t <= N; ++t) {
      VectorXd v(nv);
      for (int i = 0; i < nv; ++i) v[i] = (prob.q_nom[t][i] - prob.q_nom[t - 1][i]) / dt;
      prob.v_nom.push_back(v);
    }
    SolverParameters params;
    params.max_iterations = argc > 2 ? std::atoi(argv[2]) : 30;
    params.method = idto::optimizer::SolverMethod::kTrustRegion;
    params.linesearch_method = idto::optimizer::LinesearchMethod::kBacktracking;
    params.max_linesearch_iterations = 60;   // example_base.cc:474
    params.scaling = false;
    params.equality_constraints = true;
    params.Delta0 = 1e3;
    params.verbose = false;
    params.gradients_method = idto::optimizer::GradientsMethod::kForwardDifferences;

    TrajectoryOptimizer<double> opt(mf.c_model(), dt, prob, params);
    std::vector<VectorXd> q_guess;   // the YAML's q_guess, interpolated from q_init (example_base.cc:208-213)
    const VectorXd q_guess_end = {0.0, 0.0};
    for (int t = 0; t <= N; ++t) {
      VectorXd q(nq);
      const double lam = t / ((N + 1) - 1.0);
      for (int i = 0; i < nq; ++i) q[i] = (1 - lam) * prob.q_init[i] + lam * q_guess_end[i];
      q_guess.push_back(q);
    }
    TrajectoryOptimizerSolution<double> solution;
    TrajectoryOptimizerStats<double> stats;
    const SolverFlag flag = opt.Solve(q_guess, &solution, &stats);
    std::printf("flag %d\n", (int)flag);
    std::printf("num_equality_constraints %d\n", opt.num_equality_constraints());
    for (std::size_t i = 0; i < stats.iteration_costs.size(); ++i)
      std::printf("iter %zu cost %.17g delta %.17g h_norm %.17g\n", i, stats.iteration_costs[i], stats.trust_region_radii[i], stats.h_norms[i]);
    std::printf("qN");
    for (double x : solution.q.back()) std::printf(" %.17g", x);
    std::printf("\n");
    if (argc > 3) stats.SaveToCsv(argv[3]);
    // what the process has mapped
    std::ifstream maps("/proc/self/maps");
    int torch = 0, python = 0, rccl = 0, idto = 0;
    for (std::string line; std::getline(maps, line);) {
      torch += line.find("libtorch") != std::string::npos || line.find("libc10") != std::string::npos;
      python += line.find("libpython") != std::string::npos;
      rccl += line.find("librccl") != std::string::npos;
      idto += line.find("libidto_hip.so") != std::string::npos;
    }
    std::printf("mapped torch %d python %d rccl %d idto_hip %d\n", torch > 0, python > 0, rccl > 0, idto > 0);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "solve_acrobot: %s\n", e.what());
    return 1;
  }
}
