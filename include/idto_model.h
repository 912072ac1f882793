/* idto_model.h — plain-C description of a multibody model, a trajectory
 * optimisation problem and the contact parameters, as handed across the C-ABI.
 *
 * The reference gets all of this from a Drake `MultibodyPlant` + `SceneGraph`
 * (reference optimizer/trajectory_optimizer.cc:43-72) and from
 * `ProblemDefinition` / `SolverParameters`
 * (reference optimizer/problem_definition.h:24-59,
 * optimizer/solver_parameters.h:64-167).  Drake is not part of this build, so
 * the plant is replaced by the flat tables below (produced offline by
 * tools/convert_models.py from the reference's URDF/SDF files, or filled in by
 * hand).  Only plain pointers and sizes: no C++ types, no torch types.
 *
 * Conventions
 *  - bodies are the *moving* bodies (welded links are merged into their
 *    parent), numbered so that parent[i] < i; parent -1 is the world;
 *  - 3x3 rotations are row-major, X = [R(9) | p(3)] is 12 doubles;
 *  - per-timestep blocks (partials, Hessian bands) are column-major like
 *    Eigen's default, blocks are stored t-major and contiguous.
 */
#ifndef IDTO_MODEL_H_
#define IDTO_MODEL_H_

#ifdef __cplusplus
extern "C" {
#endif

#define IDTO_MAX_PATHS 8   /* lanes cooperating on one inverse-dynamics evaluation */
#define IDTO_MAX_CHAIN 8   /* bodies in one path's own chain */

enum idto_joint_type {
  IDTO_JOINT_REVOLUTE = 0,  /* 1 q, 1 v: rotation about `axis` (unit, in F) */
  IDTO_JOINT_PRISMATIC = 1, /* 1 q, 1 v: translation along `axis` */
  IDTO_JOINT_PLANAR = 2,    /* 3 q, 3 v: [x, y, theta] in F's x-y plane, about F's z */
  IDTO_JOINT_FLOATING = 3   /* 7 q [qw qx qy qz x y z], 6 v [w_W(3) v_W(3)]; parent must be world */
};

enum idto_geom_type {
  IDTO_GEOM_SPHERE = 0, /* size[0] = radius */
  IDTO_GEOM_BOX = 1     /* size = half extents */
};

typedef struct idto_model {
  int nbodies, nq, nv;
  const int* parent;        /* [nbodies] */
  const int* jtype;         /* [nbodies] idto_joint_type */
  const int* qstart;        /* [nbodies] */
  const int* vstart;        /* [nbodies] */
  const double* X_PF;       /* [nbodies*12] joint frame F in the parent body frame */
  const double* axis;       /* [nbodies*3] */
  const double* mass;       /* [nbodies] */
  const double* com;        /* [nbodies*3] centre of mass in the body frame */
  const double* inertia;    /* [nbodies*6] about the COM, body axes: xx yy zz xy xz yz */
  const double* damping;    /* [nv] joint viscous damping */
  const int* actuated;      /* [nv] 1 = actuated, 0 = unactuated DoF */
  double gravity[3];

  int ngeoms;
  const int* geom_body;     /* [ngeoms] body index, -1 = world */
  const int* geom_type;     /* [ngeoms] */
  const double* geom_X;     /* [ngeoms*12] geometry frame in its body frame */
  const double* geom_size;  /* [ngeoms*3] */

  int npairs;               /* candidate signed-distance pairs (after collision filters) */
  const int* pair_a;        /* [npairs] geometry A (lower registration index) */
  const int* pair_b;        /* [npairs] geometry B */
  /* Supported pairs: sphere-sphere, sphere-box (either order, any poses), and box-box ONLY as
   * (A = box on a moving body, B = world-fixed box with identity rotation): B's top face is
   * taken as the half-space z <= top (its x/y extent is not tested) and A's lowest vertex is the
   * witness point - the body-box / foot-box vs ground-box pairs of the reference's examples
   * (examples/mini_cheetah/mini_cheetah.cc:50-55).  idto_hip_create refuses any other box-box
   * pair; the reference gets general closest points from Drake/FCL (TO.cc:271-279). */

  /* Evaluation/summation-order specification ("star" decomposition): one
   * optional common root body (computed by every path) plus npaths disjoint
   * chains hanging off the world or off the common body.  It fixes the
   * association order of the floating-point sums over children / contact
   * pairs so that a serial CPU evaluation and the
   * lane-parallel HIP evaluation produce identical bits.
   * HARD LIMIT: this is the only topology the device evaluates - a tree whose branching happens
   * at the world and at ONE body (no closed loops, no second branching body further down a
   * chain), chains of at most IDTO_MAX_CHAIN bodies.  It covers the reference's five example
   * models; the reference itself takes any MultibodyPlant (TO.cc:271-279).  idto_hip_create
   * rejects a model whose tables do not describe such a tree. */
  int npaths;               /* power of two, <= IDTO_MAX_PATHS */
  int common_body;          /* body index or -1 */
  const int* body_path;     /* [nbodies] path of each body, -1 for the common body */
  const int* pair_path;     /* [npairs] path that evaluates the pair */
} idto_model_t;

typedef struct idto_contact_params {
  double contact_stiffness;     /* k   [N/m]  (solver_parameters.h:122) */
  double dissipation_velocity;  /* v_d [m/s]  (:123) */
  double stiction_velocity;     /* v_s [m/s]  (:124) */
  double friction_coefficient;  /* mu         (:125) */
  double smoothing_factor;      /* sigma      (:126) */
} idto_contact_params_t;

typedef struct idto_problem {
  int num_steps;                /* N */
  double time_step;             /* dt (the reference reads plant.time_step()) */
  const double* q_init;         /* [nq] */
  const double* v_init;         /* [nv] */
  const double* Qq;             /* [nq*nq] column-major, per unit time */
  const double* Qv;             /* [nv*nv] */
  const double* Qf_q;           /* [nq*nq] */
  const double* Qf_v;           /* [nv*nv] */
  const double* R;              /* [nv*nv] */
  const double* q_nom;          /* [(N+1)*nq] */
  const double* v_nom;          /* [(N+1)*nv] */
} idto_problem_t;

/* Mirror of the reference's SolverParameters (optimizer/solver_parameters.h:64-167)
 * and ConvergenceCriteriaTolerances (optimizer/convergence_criteria_tolerances.h:8-39);
 * enum values follow the reference's declaration order (solver_parameters.h:14-62). */
typedef struct idto_solver_params {
  int check_convergence;            /* default 0 */
  double rel_cost_reduction, abs_cost_reduction;
  double rel_gradient_along_dq, abs_gradient_along_dq;
  double rel_state_change, abs_state_change;
  int method;                       /* 0 kLinesearch, 1 kTrustRegion (default) */
  int linesearch_method;            /* 0 kArmijo (default), 1 kBacktracking */
  int max_iterations;               /* 100 */
  int max_linesearch_iterations;    /* 50 */
  int gradients_method;             /* 0 fwd (default), 1 central, 2 central4, 3 autodiff, 4 none */
  int linear_solver;                /* 0 kDenseLdlt, 1 kPentaDiagonalLu (default) */
  int normalize_quaternions;        /* 0 */
  int verbose;                      /* reference default 1 */
  int scaling;                      /* 1 */
  int scaling_method;               /* 0 sqrt, 1 adaptive sqrt, 2 double sqrt (default), 3 adaptive double sqrt */
  int equality_constraints;         /* 1 */
  double Delta0, Delta_max;         /* 1e-1, 1e5 */
  int num_threads;                  /* 1 */
  int print_debug_data;             /* 0: condition numbers etc. per iteration (solver_parameters.h:101-103) */
  int debug_compare_against_dense;  /* 0: "Sparse vs. Dense error" per dogleg point (:105-110) */
  int exact_hessian;                /* 0; 1 is rejected (needs autodiff) */
  int plot_dumps;                   /* 0; save_contour_data | save_lineplot_data | linesearch_plot_every_iteration: rejected */
} idto_solver_params_t;

/* Per-iteration statistics, the 13 series of TrajectoryOptimizerStats
 * (optimizer/trajectory_optimizer_solution.h:58-139); caller provides arrays of
 * capacity `capacity`, the callee fills `count` entries. */
typedef struct idto_stats {
  int capacity, count;
  double solve_time;
  double* iteration_times; double* iteration_costs; int* linesearch_iterations;
  double* linesearch_alphas; double* trust_region_radii; double* q_norms; double* dq_norms;
  double* dqH_norms; double* trust_ratios; double* gradient_norms; double* dL_dqs;
  double* h_norms; double* merits;
  int total;  /* iterations the solver ran; count = min(capacity, total): total > count means the arrays were too short */
} idto_stats_t;

#ifdef __cplusplus
}
#endif
#endif  /* IDTO_MODEL_H_ */
