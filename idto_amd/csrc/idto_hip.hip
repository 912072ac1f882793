// idto_hip.hip — C-ABI (include/idto_hip.h) over the gfx950 kernels in kernels.h.
// Owns the HBM-resident state of one trajectory-optimisation problem:
//   q, N+, v, a | slab (dtau/dq + tau, t-major) | g, H bands | factors | step
// Nothing in here falls back to the CPU: if no HIP device is present every
// entry point fails with a negative status.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <memory>
#include <mutex>
#include <type_traits>
#include <utility>
#include <thread>
#include <string>
#include <vector>

#include "idto_hip.h"
#include "kernels.h"
#include "fd_launch.h"
#include "penta_ldl.h"
#include "fused.h"
#include "penta_nd.h"
#include "penta_pipe.h"
#include "penta_band.h"
#include "gn_small.h"
#include "penta_apply.h"
#include "constraints.h"
#include "dense_ldl.h"
#include "trust_region.h"
#include "kkt.h"

using namespace idto_dev;

namespace {
thread_local std::string g_err;

// ---- host-side timeline (idto_hip_trace_*): wall-clock marks of what the host thread does between the C-ABI's entry
// points, for tools/host_profile.py --mpc (one MPC re-plan accounted for in steps of 10 us).  Off: one
// predictable branch per mark.
struct HostTrace {
  bool on = false;
  std::chrono::steady_clock::time_point t0;
  std::vector<std::pair<std::string, double>> ev;
};
static HostTrace g_trace;
static std::mutex g_trace_mu;   // (idto_hip_tr_solve_batch_constrained marks from several host threads)
extern "C" void idto_hip_trace_enable(int on) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace.on = on != 0;
  g_trace.ev.clear();
  g_trace.ev.reserve(256);
  g_trace.t0 = std::chrono::steady_clock::now();
}
extern "C" void idto_hip_trace_mark(const char* label) {
  if (!g_trace.on) return;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace.ev.emplace_back(label, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g_trace.t0).count());
}
// "<microseconds since enable> <label>\n" per mark; returns the number of bytes the whole text needs
extern "C" int idto_hip_trace_dump(char* out, int cap) {
  std::string t;
  char buf[64];
  std::lock_guard<std::mutex> lk(g_trace_mu);
  for (const auto& e : g_trace.ev) { std::snprintf(buf, sizeof buf, "%.1f ", e.second); t += buf; t += e.first; t += "\n"; }
  if (out && cap > 0) std::snprintf(out, (size_t)cap, "%s", t.c_str());
  return (int)t.size() + 1;
}
#define TRACE(label) idto_hip_trace_mark(label)

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                                  \
      return -2;                                                                                  \
    }                                                                                             \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};
}  // namespace

// Host copies of what a context was created from: a batch context builds single-problem child contexts from them for
// the trust-region loop with ENFORCED CONSTRAINTS (idto_hip_tr_solve_batch_constrained)
struct HostModelCopy {
  idto_model_t m;
  std::vector<int> parent, jtype, qstart, vstart, actuated, geom_body, geom_type, pair_a, pair_b, body_path, pair_path;
  std::vector<double> X_PF, axis, mass, com, inertia, damping, geom_X, geom_size;
  void Set(const idto_model_t& s) {
    m = s;
    const int nb = s.nbodies, ng = s.ngeoms, np = s.npairs;
    auto ci = [](const int* p, size_t n) { return p ? std::vector<int>(p, p + n) : std::vector<int>(); };
    auto cd = [](const double* p, size_t n) { return p ? std::vector<double>(p, p + n) : std::vector<double>(); };
    parent = ci(s.parent, nb); jtype = ci(s.jtype, nb); qstart = ci(s.qstart, nb); vstart = ci(s.vstart, nb);
    actuated = ci(s.actuated, s.nv); geom_body = ci(s.geom_body, ng); geom_type = ci(s.geom_type, ng);
    pair_a = ci(s.pair_a, np); pair_b = ci(s.pair_b, np); body_path = ci(s.body_path, nb); pair_path = ci(s.pair_path, np);
    X_PF = cd(s.X_PF, (size_t)12 * nb); axis = cd(s.axis, (size_t)3 * nb); mass = cd(s.mass, nb); com = cd(s.com, (size_t)3 * nb);
    inertia = cd(s.inertia, (size_t)6 * nb); damping = cd(s.damping, s.nv); geom_X = cd(s.geom_X, (size_t)12 * ng);
    geom_size = cd(s.geom_size, (size_t)3 * ng);
    m.parent = parent.data(); m.jtype = jtype.data(); m.qstart = qstart.data(); m.vstart = vstart.data();
    m.actuated = actuated.empty() ? nullptr : actuated.data(); m.geom_body = geom_body.data(); m.geom_type = geom_type.data();
    m.pair_a = pair_a.data(); m.pair_b = pair_b.data(); m.body_path = body_path.data(); m.pair_path = pair_path.data();
    m.X_PF = X_PF.data(); m.axis = axis.data(); m.mass = mass.data(); m.com = com.data(); m.inertia = inertia.data();
    m.damping = damping.data(); m.geom_X = geom_X.data(); m.geom_size = geom_size.data();
  }
};
struct HostProblemCopy {
  idto_problem_t p;
  std::vector<double> q_init, v_init, Qq, Qv, Qf_q, Qf_v, R, q_nom, v_nom;
  void Set(const idto_problem_t& s, int nq, int nv) {
    p = s;
    const size_t N1 = (size_t)s.num_steps + 1;
    auto cd = [](const double* x, size_t n) { return x ? std::vector<double>(x, x + n) : std::vector<double>(n, 0.0); };
    q_init = cd(s.q_init, nq); v_init = cd(s.v_init, nv); Qq = cd(s.Qq, (size_t)nq * nq); Qv = cd(s.Qv, (size_t)nv * nv);
    Qf_q = cd(s.Qf_q, (size_t)nq * nq); Qf_v = cd(s.Qf_v, (size_t)nv * nv); R = cd(s.R, (size_t)nv * nv);
    q_nom = cd(s.q_nom, N1 * nq); v_nom = cd(s.v_nom, N1 * nv);
    p.q_init = q_init.data(); p.v_init = v_init.data(); p.Qq = Qq.data(); p.Qv = Qv.data(); p.Qf_q = Qf_q.data();
    p.Qf_v = Qf_v.data(); p.R = R.data(); p.q_nom = q_nom.data(); p.v_nom = v_nom.data();
  }
};

// RCCL is resolved at run time, when the first communicator call needs it: libidto_hip.so has no NEEDED entry for it
// (VERDICT r4 "weak" #8: the link-time dependency carried a RUNPATH of the build image's /opt/rocm-7.2.0/lib, and on a
// box with another ROCm the library loaded only where torch had mapped its own librccl first - a plain C++ consumer has
// no such luck, and one that never shards needs no RCCL at all).  Order: a librccl the process has mapped already (a
// host that imported torch: both must talk to the SAME library), $IDTO_RCCL_LIB, the loader's search path, /opt/rocm/lib.
struct RcclApi {
  void* handle = nullptr;
  std::string path, tried;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};
static RcclApi& RcclState() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<std::pair<std::string, int>> cand = {{"librccl.so.1", RTLD_NOW | RTLD_NOLOAD}, {"librccl.so", RTLD_NOW | RTLD_NOLOAD}};
    if (const char* e = std::getenv("IDTO_RCCL_LIB")) cand.push_back({e, RTLD_NOW});
    for (const char* n : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) cand.push_back({n, RTLD_NOW});
    for (const auto& c : cand) {
      void* h = dlopen(c.first.c_str(), c.second | RTLD_GLOBAL);
      if (!h) { if (!(c.second & RTLD_NOLOAD)) api.tried += " " + c.first; continue; }
      api.handle = h;
      break;
    }
    if (!api.handle) return;
    bool all = true;
    auto sym = [&](const char* name, auto& fn) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, name));
      if (!fn) { all = false; api.tried += std::string(" (no ") + name + ")"; }
    };
    sym("ncclGetVersion", api.GetVersion); sym("ncclGetErrorString", api.GetErrorString); sym("ncclGetUniqueId", api.GetUniqueId);
    sym("ncclCommInitRank", api.CommInitRank); sym("ncclCommInitAll", api.CommInitAll); sym("ncclCommDestroy", api.CommDestroy);
    sym("ncclAllGather", api.AllGather); sym("ncclGroupStart", api.GroupStart); sym("ncclGroupEnd", api.GroupEnd);
    if (!all) { api.handle = nullptr; return; }
    Dl_info info;
    api.path = (dladdr(reinterpret_cast<void*>(api.AllGather), &info) && info.dli_fname) ? info.dli_fname : "?";
  });
  return api;
}
#define RCCL_OR_FAIL(R)                                                                                        \
  RcclApi* R = RcclState().handle ? &RcclState() : nullptr;                                                    \
  if (!R) { g_err = "librccl could not be loaded (set IDTO_RCCL_LIB; tried:" + RcclState().tried + ")"; return -4; }


struct idto_hip_ctx {
  int device = 0;
  // (batch contexts) what the context was created from, and one single-problem context per problem, made on first use
  std::unique_ptr<HostModelCopy> host_model;
  std::vector<std::unique_ptr<HostProblemCopy>> host_problems;
  idto_contact_params_t host_contact{};
  std::vector<idto_hip_ctx*> children;
  // batch: `batch` problems of the same model / horizon, one arena each (identical layout,
  // `pstride` bytes apart); the pointers below address problem 0
  int batch = 1;
  size_t pstride = 0;
  char* arena = nullptr;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int nb = 0, nq = 0, nv = 0, N = 0, npaths = 1, maxc = 1;
  double dt = 0;
  std::vector<void*> allocs;
  DevModel M;
  DevContact cp;
  DevProblem P;
  // problem arrays (device)
  double *d_vinit = nullptr, *d_qnom = nullptr, *d_vnom = nullptr;
  double* d_w[10] = {nullptr};
  // state
  double *q = nullptr, *v = nullptr, *a = nullptr, *nplus = nullptr, *slab = nullptr;
  double *g = nullptr, *HA = nullptr, *HB = nullptr, *HC = nullptr, *step = nullptr, *cost = nullptr;
  double *g2 = nullptr, *HA2 = nullptr, *HB2 = nullptr, *HC2 = nullptr;   // the trial point's g and H while a deciding launch runs (penta_pipe.h PipeAsm::g2)
  double *Kst = nullptr, *LUst = nullptr, *Yst = nullptr, *Zst = nullptr;
  int* pivst = nullptr;
  double *Ust = nullptr, *Hst = nullptr, *Est = nullptr, *Dst = nullptr;  // LDL^T factors (padded K x K blocks)
  double* dbg = nullptr;  // cycle stamps of the solver (8 per block row)
  bool solver_debug = false;
  int slab_stride = 0;
  int k_begin = 0, k_end = 0;
  bool weights_diagonal = false;
  std::vector<char> prob_diag;            // per problem of the batch
  bool reference_solver = false;  // bit-exact pivoted-LU block Thomas (kernels.h penta_kernel)
  int asm_diag_lds = 0;
  int asm_terms_lds = 0;
  int asm_terms_threads = 256;              // a thread per item of the largest part (allegro: 299 items - a second pass of 43 cost 2 us)
  double* terms = nullptr;                // per-record assembly products written by fd_kernel (asm_terms_stride)
  bool asm_fold = true;                   // option "asm_fold": fd_kernel forms them, assemble_terms_kernel combines
  bool terms_valid = false;               // ... and they belong to the resident slab, for every k
  int last_assembly = 0;                  // 1 assemble_terms_kernel, 2 assemble_diag_kernel, 3 assemble_kernel
  // launch geometry
  int fd_threads = 256, fd_lds = 0, asm_lds = 0, penta_lds = 0, solve_lds = 0, cost_lds = 0;
  // timing
  bool timing = false;
  int timing_stride = 1;     // record events on every timing_stride-th launch of each kernel
  unsigned timing_tick[4] = {0, 0, 0, 0};
  bool timing_open = false;  // TimeBegin recorded an event that TimeEnd must close
  struct Timed { hipEvent_t a, b; int which; };
  std::vector<Timed> pending;
  bool fd_full = false;                   // v / N+ in HBM belong to the resident q for every t
  bool partials_ahead = false;            // idto_hip_eval_tau_partials has left the partials of the resident q: the next idto_hip_eval_partials is done
  int gradients_method = 0;               // 0 forward, 1 central, 2 central 4th order (solver_parameters.h:26-50)
  int fd_stop = 0;                        // same for the finite-difference kernel
  int nd_min_rows = 16;                   // option "nd_min_rows": systems of at least this many block rows take the five-workgroup / band kernels
  bool fd_fast = true;                    // option "fd_fast": id_fast.h's evaluation when the model has an instantiated shape
  int asm_stop = 0;                       // profiling aid: truncate the assembly kernel after a phase
  bool two_sided = true;                  // solver: two workgroups eliminating from both ends
  bool fused = true;                      // gn_step: one persistent launch (fused.h) when eligible
  bool solver_nd = true;                  // solver: nested dissection over 7 workgroups (penta_nd.h) when eligible
  bool solver_pipe = true;                // ... with pipelined chains (penta_pipe.h: 5 workgroups) when the block size allows
  BandStageItem* small_stage = nullptr; int small_stage_n = 0, small_stage_N = -1;   // gn_small.h: the solver's staging table for horizon small_stage_N
  int gn_small = 1;                       // option "gn_small": the whole step of a small model in one workgroup (gn_small.h; see SmallEligible)
  int solver_band = 1;                    // small blocks: the scalar band factorisation in one workgroup (penta_band.h; see BandEligible)
  unsigned long long* nd_rowcnt = nullptr; // its per-row release counters, buffers and launch count
  unsigned* asm_ready = nullptr;               // penta_pipe.h PipeAsm: [N + 1][4] epoch words of the assembly inside the solver's launch
  bool tr_conv_on = false;                 // idto_hip_tr_set_convergence
  double tr_conv_tol[6] = {0, 0, 0, 0, 0, 0};
  bool asm_in_solver = true;                  // option "asm_in_solver": idto_hip_gn_step assembles g and H inside the pipelined solver's launch
  const double* fuse_gate = nullptr;          // ... gated: problems whose word is 0 keep their g and H (idto_hip_tr_solve)
  bool fuse_asm_next = false;                 // (set by idto_hip_gn_step for the FactorSolve that follows)
  unsigned long long* pipe_rowcnt = nullptr;   // the same for the pipelined variant (its own launch count: the two
  unsigned long long pipe_launches = 0;        // variants release a row with different increments)
  int solver_timeouts = 0;                // launches whose waits between workgroups ran out (FactorStatus)
  unsigned timeouts_handled = 0;          // ... the device's count of them (status word) that FactorStatus has acted on
  int last_step_kind = 0;                 // what produced IDTO_ARR_STEP last: 1 factor_solve of -g, 2 the fused launch, 0 other
  int debug_skip_role = -1;               // test aid: a role of the nested-dissection kernels that returns at once
  int debug_pipe_tail = 0;                // measurement aid: pipelined solver with the row-by-row back substitution
  double* nd_buf = nullptr;
  double* nd_wst = nullptr;   // the seven-workgroup kernel's W rows (penta_pipe.h chain_recursion_tail), blocks of 21 .. 32 only
  int nd_recursion = 1;       // option "nd_recursion": its back substitution in recursion form where the matrices fit the LDS
  unsigned long long nd_launches = 0;
  int last_solver = 0;                     // 0 none yet, 1 two-workgroup LDL^T, 2 nested dissection, 3 reference LU
  bool kkt_debug = false;                 // option "solver_debug" 2
  bool fused_debug = false;               // ... with per-workgroup time stamps in IDTO_ARR_DEBUG
  double* xch = nullptr;                  // their exchange buffer / flags
  unsigned* flags = nullptr;
  size_t xch_count = 0, flag_count = 0;
  unsigned epoch = 0;
  unsigned long long* sync_cnt = nullptr;            // fused launch: [fd blocks done, assembly blocks done] (monotonic)
  unsigned long long sync_steps = 0;                 // fused launches so far
  double* Tst = nullptr;                             // column-major copies of the factor blocks (penta_apply.h)
  double *stage_rhs = nullptr, *stage_x = nullptr;  // idto_hip_solve_host
  double* pack = nullptr;                            // [tau | cost] of idto_hip_trial_cost (device)
  double* pin = nullptr;                             // pinned host staging: q in, [tau | cost] out
  char* prob_pin = nullptr; size_t prob_pin_bytes = 0;   // ... of the problem arrays (UploadProblemArrays)
  double* many_pin = nullptr; size_t many_cap = 0;        // ... of idto_hip_get_many
  double* rows_pin = nullptr; size_t rows_cap = 0;        // ... of idto_hip_tr_solve's statistics rows
  // equality-constraint step (constraints.h)
  int* con_dofs = nullptr; int con_nu = 0, con_neq = 0;
  std::vector<int> con_dofs_host;
  bool con_schur_valid = false;            // con_S, con_d, ... are allocated for con_dofs_host
  int con_dofs_cap = 0;                    // entries con_dofs has room for
  double *con_S = nullptr, *con_lambda = nullptr, *con_out = nullptr;  // device: [S | J y_g], lambda, [step | J^T lambda]
  double *con_d = nullptr, *con_h = nullptr;                            // dense LDL^T: pivots, [min, max | h]
  double* con_rv = nullptr;                                            // ... and L^-1 (h - J y_g), carried along by the factorisation
  double* con_L = nullptr;                                             // ... and the factor L (dense_ldl_step_kernel only reads S's panels)
  bool con_S_factored = false;                                         // con_S holds the LDL^T factors, not S
  // SolverParameters::linear_solver = kDenseLdlt (idto_hip_solve_dense_ldlt): [H dense | L], pivots, [min, max], [r | y]
  double *dn_S = nullptr, *dn_d = nullptr, *dn_stat = nullptr, *dn_rv = nullptr; int dn_n = 0;
  double* fetch_dev = nullptr; double* fetch_pin = nullptr; size_t fetch_cap = 0;   // idto_hip_tr_solve_fetch's staging
  // option "async_uploads": idto_hip_set_q / idto_hip_set_problem copy from pinned staging of the context - two buffers
  // taken in turn, an event each - and return without waiting (everything else the context does is ordered behind them on
  // its stream); a caller that reads device memory on ANOTHER stream keeps the default, the blocking copies
  bool async_uploads = false;
  char* up_pin[2] = {nullptr, nullptr}; size_t up_cap[2] = {0, 0}; hipEvent_t up_ev[2] = {nullptr, nullptr}; int up_next = 0;
  double* con_pin = nullptr; size_t con_pin_count = 0;                 // pinned host staging for the above
  bool h_assembled = false;                                            // H comes from idto_hip_grad_hess: block row 0 is the identity
  bool con_begun = false;                                              // constraint_schur_begin enqueued for the current H
  // idto_hip_prefetch: copies enqueued on a side stream behind an event of the main stream
  struct Prefetch { int what = -1; size_t off = 0, count = 0; bool pending = false; hipEvent_t ev = nullptr; };
  Prefetch pre[4];
  hipStream_t side = nullptr;
  double* pre_pin = nullptr; size_t pre_cap = 0;
  bool con_ready = false;                                              // stage_x holds H^-1 [g | J^T] of the current H
  size_t stage_count = 0;
  std::vector<hipEvent_t> event_pool;  // recycled: creating events in the timed loop costs host time
  double tsum[4] = {0, 0, 0, 0};
  int tcnt[4] = {0, 0, 0, 0};
  // factorisation status, written by the solver kernels into pinned host memory only when a pivot
  // fails: [0] = id of the last failing factorisation, [1] = failing block rows since creation
  unsigned* status_pin = nullptr;
  unsigned* status_dev = nullptr;   // the same memory as the device addresses it
  unsigned fact_id = 0;             // id of the most recent factorisation launched
  // multi-GPU: this context is rank `comm_rank` of `comm_world` in an RCCL communicator;
  // its finite-difference kernel covers the k-range [k_begin, k_end) and one in-place all-gather
  // completes the slab (records are contiguous in k, every rank's range has `comm_per` records,
  // the last ones padded: the slab is allocated with IDTO_SLAB_PAD spare records)
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1, comm_per = 0;
  // device-side trust-region bookkeeping (trust_region.h): scale factors, scaled merit gradient,
  // w = D^-1 H^-1 g_merit, the last step, the trial trajectory; [16] scalars on the device + pinned
  double *tr_Dprev = nullptr, *tr_part = nullptr;   // the adaptive scalings' previous D; per-block-row partial sums
  // idto_hip_tr_solve: the iteration state and the arrival counter of tr_iter_kernel (device), the
  // per-iteration statistics rows (device, grown on demand), a pinned staging area for both
  double* tr_state = nullptr;
  // second set of fd_kernel's outputs (v, a, N+, slab, products), alt_off bytes behind the first, and
  // the selectors the launches pass (batch.h AltSel): inactive (state == nullptr) outside idto_hip_tr_solve
  long long alt_off = 0;
  AltSel alt_r{nullptr, 0, 0}, alt_w{nullptr, 0, 1};
  unsigned long long* tr_cnt = nullptr;
  unsigned long long tr_target = 0;
  double *tr_part_ll = nullptr, *tr_part2 = nullptr;   // tr_iter_kernel: the block rows' sums with the launch's epoch in every word; dq.dq, g~.dqs per row
  unsigned tr_epoch = 0;
  double* tr_rows = nullptr;
  int tr_rows_cap = 0;
  double *tr_D = nullptr, *tr_gt = nullptr, *tr_w = nullptr, *tr_dq = nullptr, *q_trial = nullptr, *tr_out = nullptr;
  double* tr_pin = nullptr;
  int* tr_quat = nullptr; int tr_nquat = 0;
  bool trial_resident = false;         // v / a / tau / cost in device memory belong to q_trial
  // speculation: idto_hip_tr_trial enqueued the next iteration's gn_step + tr_prepare on the trial
  // point before the host knew whether it accepts it (spec_scaling: the arguments it used)
  bool spec_pending = false, spec_ready = false;
  int spec_scaling = -2;
  hipEvent_t spec_ev = nullptr;
  const double* con_lambda_at = nullptr;  // where the current multipliers live (con_lambda or con_lambda + 2)
  int* una_dofs = nullptr; int una_nu = 0; // unactuated dofs for |h| of the statistics (idto_hip_set_unactuated_dofs)
  std::vector<int> una_dofs_host;
  // the equality-constraint step of the resident loop as one banded solve (kkt.h): a solver-only context of block size
  // nq + nu on this context's stream, made on first use
  idto_hip_ctx* kkt = nullptr; int kkt_nu = 0;
  bool con_kkt = true;                     // option "con_kkt" (0: the Schur-complement chain of constraints.h)
  bool kkt_fold = true;                    // option "kkt_fold" (0: kkt_extract_kernel in a launch of its own in front of tr_iter_kernel)
  bool tr_small = true;                    // option "tr_small" (0: fd_kernel, cost_kernel and the solver's launch per iteration also for the small models)
  bool tr_fold = true;                     // option "tr_fold" (0: tr_iter_kernel stays a launch of its own in front of the small models' launch)
  bool kkt_in_asm = true;                  // option "kkt_in_asm" (0: kkt_build_kernel stays a launch of its own behind the assembly)
  bool decide_in_solver = true;            // option "decide_in_solver" (0: cost_kernel stays a launch of its own in front of the pipelined solver's)
  const TrDecideArgs* fuse_decide = nullptr;   // (during idto_hip_tr_solve's call of the solver) the decision the launch is to make
  double* decide_word = nullptr; unsigned decide_epoch = 0;   // ... and the word it publishes it in (per problem, epoch in every word)
  bool tr_resident_ok = true;              // false once tr_iter_kernel's wait between its workgroups ran out (FactorStatus): idto_hip_tr_solve then refuses
  int ldl_npos = 0;                        // (a KKT context) the solver expects the pivots [ldl_npos, nq) of a block negative
};
enum { IDTO_SLAB_PAD = 64 };

namespace {

template <class T>
int Upload(idto_hip_ctx* c, const T* host, size_t count, T** dev) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
  c->allocs.push_back(p);
  if (count) HIP_OK(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
  *dev = static_cast<T*>(p);
  return 0;
}
template <class T>
int Alloc(idto_hip_ctx* c, size_t count, T** dev) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
  // hipMemset on device memory runs on the NULL stream and does not block the host; the context's
  // stream is non-blocking, i.e. NOT ordered after the NULL stream: without the synchronisation the
  // zero fill can land after work the caller enqueues next on the context's stream (seen as a
  // one-in-fifty wrong first solve on the smallest model)
  HIP_OK(hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T)));
  HIP_OK(hipDeviceSynchronize());
  c->allocs.push_back(p);
  *dev = static_cast<T*>(p);
  return 0;
}
// a buffer of Alloc's that has been superseded: returned to the device now, not at idto_hip_destroy
template <class T>
void Release(idto_hip_ctx* c, T** dev) {
  if (!*dev) return;
  auto it = std::find(c->allocs.begin(), c->allocs.end(), static_cast<void*>(*dev));
  if (it != c->allocs.end()) { (void)hipFree(*it); c->allocs.erase(it); }
  *dev = nullptr;
}

// pinned staging for an upload that must not wait (option "async_uploads"): the buffer that was used the time before
// last - its copy has long run, the event only makes that certain
static int UploadStage(idto_hip_ctx* c, size_t bytes, char** buf, int* slot) {
  const int s = c->up_next;
  c->up_next ^= 1;
  if (c->up_ev[s]) HIP_OK(hipEventSynchronize(c->up_ev[s]));
  else HIP_OK(hipEventCreateWithFlags(&c->up_ev[s], hipEventDisableTiming));
  if (c->up_cap[s] < bytes) {
    if (c->up_pin[s]) (void)hipHostFree(c->up_pin[s]);
    c->up_pin[s] = nullptr; c->up_cap[s] = 0;
    HIP_OK(hipHostMalloc((void**)&c->up_pin[s], std::max<size_t>(bytes, 4096), hipHostMallocDefault));
    c->up_cap[s] = std::max<size_t>(bytes, 4096);
  }
  *buf = c->up_pin[s]; *slot = s;
  return 0;
}

// (problem `pb` of the batch: destination = problem 0's arrays shifted by pb * pstride)
int UploadProblemArrays(idto_hip_ctx* c, const idto_problem_t* p, int pb = 0) {
  TRACE("hip: problem upload begins");
  const int nq = c->nq, nv = c->nv, N = c->N;
  const size_t po = (size_t)pb * c->pstride;
  const double dt = c->dt;
  auto scaled = [](const double* W, size_t n, double s1, double s2) {
    std::vector<double> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = (s1 * W[i]) * s2;  // reference TO.cc:1103-1107
    return out;
  };
  const std::vector<double> w[10] = {
      scaled(p->Qq, (size_t)nq * nq, 2, dt), scaled(p->Qv, (size_t)nv * nv, 2, dt), scaled(p->R, (size_t)nv * nv, 2, dt),
      scaled(p->Qf_q, (size_t)nq * nq, 2, 1), scaled(p->Qf_v, (size_t)nv * nv, 2, 1),
      std::vector<double>(p->Qq, p->Qq + (size_t)nq * nq), std::vector<double>(p->Qv, p->Qv + (size_t)nv * nv),
      std::vector<double>(p->R, p->R + (size_t)nv * nv), std::vector<double>(p->Qf_q, p->Qf_q + (size_t)nq * nq),
      std::vector<double>(p->Qf_v, p->Qf_v + (size_t)nv * nv)};
  {
    // ONE copy: v_init, q_nom, v_nom and the ten weight matrices sit next to each other in the problem's arena (64-byte
    // aligned), a pinned host buffer mirrors that stretch (13 copies from pageable memory were 0.12 ms of an MPC re-plan)
    char* lo = reinterpret_cast<char*>(c->d_vinit);
    char* hi = reinterpret_cast<char*>(c->d_w[9]) + w[9].size() * sizeof(double);
    const size_t bytes = (size_t)(hi - lo);
    char* stage = nullptr; int slot = -1;
    if (c->async_uploads) { if (int rc = UploadStage(c, bytes, &stage, &slot)) return rc; }
    if (!stage && c->prob_pin_bytes < bytes) {
      if (c->prob_pin) (void)hipHostFree(c->prob_pin);
      c->prob_pin = nullptr; c->prob_pin_bytes = 0;
      HIP_OK(hipHostMalloc((void**)&c->prob_pin, bytes, hipHostMallocDefault));
      c->prob_pin_bytes = bytes;
      std::memset(c->prob_pin, 0, bytes);
    }
    if (!stage) stage = c->prob_pin;
    else std::memset(stage, 0, bytes);   // (the alignment gaps between the arrays)
    auto put = [&](const double* dev, const double* src, size_t count) {
      std::memcpy(stage + (reinterpret_cast<const char*>(dev) - lo), src, count * sizeof(double));
    };
    put(c->d_vinit, p->v_init, nv);
    put(c->d_qnom, p->q_nom, (size_t)(N + 1) * nq);
    put(c->d_vnom, p->v_nom, (size_t)(N + 1) * nv);
    for (int i = 0; i < 10; ++i) put(c->d_w[i], w[i].data(), w[i].size());
    HIP_OK(hipMemcpyAsync(lo + po, stage, bytes, hipMemcpyHostToDevice, c->stream));
    if (slot >= 0) {
      HIP_OK(hipEventRecord(c->up_ev[slot], c->stream));
      TRACE("hip: problem upload done (one copy enqueued)");
    } else {
      HIP_OK(hipStreamSynchronize(c->stream));  // (the pinned buffer is reused by the next upload)
      TRACE("hip: problem upload done (one copy + wait)");
    }
  }
  auto is_diag = [](const double* W, int n) {
    for (int c2 = 0; c2 < n; ++c2)
      for (int r = 0; r < n; ++r)
        if (r != c2 && W[(size_t)c2 * n + r] != 0.0) return false;
    return true;
  };
  const bool diag = is_diag(p->Qq, nq) && is_diag(p->Qv, nv) && is_diag(p->R, nv) && is_diag(p->Qf_q, nq) &&
                    is_diag(p->Qf_v, nv);
  // (one assembly kernel serves the whole batch: the diagonal fast path needs every problem's
  // weights diagonal; a batch re-evaluates the flag when problem 0 is replaced)
  if ((int)c->prob_diag.size() != c->batch) c->prob_diag.assign((size_t)c->batch, 0);
  c->prob_diag[(size_t)pb] = diag ? 1 : 0;
  c->weights_diagonal = true;
  for (char d : c->prob_diag) c->weights_diagonal = c->weights_diagonal && d;
  DevProblem& P = c->P;
  P.N = N; P.dt = dt; P.v_init = c->d_vinit; P.q_nom = c->d_qnom; P.v_nom = c->d_vnom;
  P.Qq = c->d_w[0]; P.Qv = c->d_w[1]; P.R = c->d_w[2]; P.Qfq = c->d_w[3]; P.Qfv = c->d_w[4];
  P.Qq0 = c->d_w[5]; P.Qv0 = c->d_w[6]; P.R0 = c->d_w[7]; P.Qfq0 = c->d_w[8]; P.Qfv0 = c->d_w[9];
  return 0;
}

int BuildModel(idto_hip_ctx* c, const idto_model_t* m) {
  const int nb = m->nbodies, K = m->npaths;
  if (K < 1 || K > IDTO_MAX_PATHS || (K & (K - 1))) { g_err = "npaths must be a power of two <= 8"; return -1; }
  // star decomposition tables
  std::vector<int> chain((size_t)K * IDTO_MAX_CHAIN, -1), nchain(K, 0), pkind((size_t)K * IDTO_MAX_CHAIN, 0);
  std::vector<int> slot_of(nb, -3);
  for (int i = 0; i < nb; ++i)
    if ((m->jtype[i] == IDTO_JOINT_PLANAR || m->jtype[i] == IDTO_JOINT_FLOATING) && m->parent[i] >= 0) {
      g_err = "planar and floating joints must be attached to the world";
      return -1;
    }
  if (m->common_body >= 0 && m->parent[m->common_body] >= 0) { g_err = "the common body must be attached to the world"; return -1; }
  for (int i = 0; i < nb; ++i) {
    if (i == m->common_body) { slot_of[i] = -1; continue; }
    const int p = m->body_path[i];
    if (p < 0 || p >= K) { g_err = "body without a valid path"; return -1; }
    const int s = nchain[p]++;
    if (s >= IDTO_MAX_CHAIN) { g_err = "chain longer than IDTO_MAX_CHAIN"; return -1; }
    chain[(size_t)p * IDTO_MAX_CHAIN + s] = i;
    slot_of[i] = s;
    const int par = m->parent[i];
    int kind;
    if (par < 0) kind = PK_WORLD;
    else if (par == m->common_body) kind = PK_COMMON;
    else if (s > 0 && chain[(size_t)p * IDTO_MAX_CHAIN + s - 1] == par) kind = PK_PREV;
    else { g_err = "model is not a star decomposition (body parent is neither world, common nor previous in path)"; return -1; }
    pkind[(size_t)p * IDTO_MAX_CHAIN + s] = kind;
  }
  int maxc = 1;
  for (int p = 0; p < K; ++p) maxc = std::max(maxc, nchain[p]);
  c->maxc = maxc;
  std::vector<int> path_npairs(K, 0);
  for (int i = 0; i < m->npairs; ++i) path_npairs[m->pair_path[i]]++;
  int maxpp = 1;
  for (int p = 0; p < K; ++p) maxpp = std::max(maxpp, path_npairs[p]);
  std::vector<int> path_pairs((size_t)K * maxpp, 0), fill(K, 0), sa(m->npairs), sb(m->npairs);
  for (int i = 0; i < m->npairs; ++i) {
    const int p = m->pair_path[i];
    path_pairs[(size_t)p * maxpp + fill[p]++] = i;
    const int ba = m->geom_body[m->pair_a[i]], bb = m->geom_body[m->pair_b[i]];
    sa[i] = ba < 0 ? -2 : slot_of[ba];
    sb[i] = bb < 0 ? -2 : slot_of[bb];
    for (int b : {ba, bb})
      if (b >= 0 && b != m->common_body && m->body_path[b] != p) { g_err = "pair touches a body outside its path"; return -1; }
    // box-box is implemented for ONE configuration only (id_eval.h signed_distance): A = a box on a
    // moving body, B = a world-fixed, axis-aligned box whose top face acts as the half-space
    // z <= top (the ground boxes of the reference's examples).  Anything else would silently get
    // wrong witness points, so it is refused here.
    if (m->geom_type[m->pair_a[i]] == IDTO_GEOM_BOX && m->geom_type[m->pair_b[i]] == IDTO_GEOM_BOX) {
      const double* XB = m->geom_X + (size_t)12 * m->pair_b[i];
      bool ident = true;
      for (int e = 0; e < 9; ++e) ident &= (XB[e] == ((e % 4 == 0) ? 1.0 : 0.0));
      if (ba < 0 || bb >= 0 || !ident) {
        g_err = "box-box contact pairs must be (box on a moving body, world-fixed axis-aligned box), in this order";
        return -1;
      }
    }
  }
  DevModel& M = c->M;
  M.nb = nb; M.nq = m->nq; M.nv = m->nv; M.npaths = K; M.common_body = m->common_body;
  M.ngeoms = m->ngeoms; M.npairs = m->npairs; M.maxpp = maxpp;
  for (int i = 0; i < 3; ++i) M.gravity[i] = m->gravity[i];
  // one blob: double tables, then int tables (two per double slot)
  std::vector<double> dbl;
  std::vector<int> ints;
  auto addd = [&](const double* src, size_t n) { const size_t o = dbl.size(); dbl.insert(dbl.end(), src, src + n); return o; };
  auto addi = [&](const int* src, size_t n) { const size_t o = ints.size(); ints.insert(ints.end(), src, src + n); return o; };
  const size_t o_XPF = addd(m->X_PF, (size_t)12 * nb), o_axis = addd(m->axis, (size_t)3 * nb), o_mass = addd(m->mass, nb),
               o_com = addd(m->com, (size_t)3 * nb), o_in = addd(m->inertia, (size_t)6 * nb),
               o_damp = addd(m->damping, m->nv), o_gX = addd(m->geom_X, (size_t)12 * m->ngeoms),
               o_gs = addd(m->geom_size, (size_t)3 * m->ngeoms);
  // N+ (TO.cc:1633-1647): its constant entries, and the non-zero range of every column and row
  const int nqm = m->nq, nvm = m->nv;
  std::vector<double> npc((size_t)nvm * nqm, 0.0);
  std::vector<int> colinfo(nqm, 0), rowinfo(nvm, 0);
  int nfloat = 0, float_qs[4] = {0, 0, 0, 0}, float_vs[4] = {0, 0, 0, 0};
  for (int b = 0; b < nb; ++b) {
    const int qs = m->qstart[b], vs = m->vstart[b], jt = m->jtype[b];
    if (jt == IDTO_JOINT_REVOLUTE || jt == IDTO_JOINT_PRISMATIC) {
      npc[(size_t)qs * nvm + vs] = 1.0; colinfo[qs] = vs | 1 << 16; rowinfo[vs] = qs | 1 << 16;
    } else if (jt == IDTO_JOINT_PLANAR) {
      for (int kq = 0; kq < 3; ++kq) {
        npc[(size_t)(qs + kq) * nvm + vs + kq] = 1.0; colinfo[qs + kq] = (vs + kq) | 1 << 16; rowinfo[vs + kq] = (qs + kq) | 1 << 16;
      }
    } else {
      for (int r = 0; r < 3; ++r)
        for (int cq = 0; cq < 4; ++cq) npc[(size_t)(qs + cq) * nvm + vs + r] = std::numeric_limits<double>::quiet_NaN();
      for (int kq = 0; kq < 4; ++kq) colinfo[qs + kq] = vs | 3 << 16;
      for (int r = 0; r < 3; ++r) rowinfo[vs + r] = qs | 4 << 16;
      for (int kq = 0; kq < 3; ++kq) {
        npc[(size_t)(qs + 4 + kq) * nvm + vs + 3 + kq] = 1.0;
        colinfo[qs + 4 + kq] = (vs + 3 + kq) | 1 << 16; rowinfo[vs + 3 + kq] = (qs + 4 + kq) | 1 << 16;
      }
      if (nfloat >= 0 && nfloat < 4) { float_qs[nfloat] = qs; float_vs[nfloat] = vs; ++nfloat; }
      else nfloat = -1;
    }
  }
  const size_t o_npc = addd(npc.data(), npc.size());
  // ---- id_fast.h: does the model have one of the instantiated tree shapes?  If so, gather one record of
  // constants per (path, slot) and per contact pair in the order id_eval_fast walks them.
  int fast_shape = 0, f_maxpp = 1;
  size_t o_fbody = 0, o_fcbody = 0, o_fpairs = 0, fast_lo = 0;
  std::vector<int> fseg(1, 0);
  {
    const int cbody = m->common_body;
    bool ok = true;
    for (int p = 0; p < K; ++p) ok = ok && nchain[p] == maxc;
    const int cj = cbody >= 0 ? m->jtype[cbody] : -1;
    if (cbody >= 0 && cj != IDTO_JOINT_FLOATING) ok = false;
    int j0 = -1, k0 = -1, w2 = -1;   // w2: a later slot of the (single) path that hangs off the world again (the spinner)
    for (int p = 0; p < K && ok; ++p)
      for (int s = 0; s < maxc; ++s) {
        const int b = chain[(size_t)p * IDTO_MAX_CHAIN + s], jt = m->jtype[b], kd = pkind[(size_t)p * IDTO_MAX_CHAIN + s];
        if (s == 0) {
          if (p == 0) { j0 = jt; k0 = kd; }
          if (jt != j0 || kd != k0) ok = false;
        } else if (jt == IDTO_JOINT_REVOLUTE && kd == PK_WORLD && K == 1 && w2 < 0) {
          w2 = s;
        } else if (jt != IDTO_JOINT_REVOLUTE || kd != PK_PREV) {
          ok = false;
        }
      }
    if (ok) {
      if (w2 >= 0) { if (w2 == 2 && maxc == 3 && K == 1 && cj == -1 && j0 == IDTO_JOINT_REVOLUTE && k0 == PK_WORLD) fast_shape = 5; }   // spinner
      else if (maxc == 2 && K == 1 && cj == -1 && j0 == IDTO_JOINT_REVOLUTE && k0 == PK_WORLD) fast_shape = 1;   // acrobot
      else if (maxc == 3 && K == 1 && cj == -1 && j0 == IDTO_JOINT_PLANAR && k0 == PK_WORLD) fast_shape = 2;     // hopper
      else if (maxc == 3 && K == 4 && cj == IDTO_JOINT_FLOATING && j0 == IDTO_JOINT_REVOLUTE && k0 == PK_COMMON) fast_shape = 3;   // mini_cheetah
      else if (maxc == 4 && K == 4 && cj == IDTO_JOINT_FLOATING && j0 == IDTO_JOINT_REVOLUTE && k0 == PK_WORLD) fast_shape = 4;    // allegro_hand + ball
    }
    // processing order of a path's pairs: [pairs without a chain body that come first | slot 0 | ... | slot maxc-1 |
    // the other pairs without a chain body].  The sums that have an order are those onto one chain body (its pairs stay
    // in list order) and the path's partial sum onto the common body: its pairs must keep their list order too.
    std::vector<std::vector<int>> order(K);
    std::vector<int>& segw = fseg;
    segw.assign((size_t)K * (maxc + 2), 0);
    for (int p = 0; p < K && fast_shape; ++p) {
      std::vector<int> mine;
      for (int j = 0; j < path_npairs[p]; ++j) mine.push_back(path_pairs[(size_t)p * maxpp + j]);
      int lo_chain_common = 1 << 30, hi_chain_common = -1, last_slot = -1;
      for (int pi : mine) {
        const int nchainb = (sa[pi] >= 0) + (sb[pi] >= 0);
        if (nchainb > 1) {
          // two chain bodies: slots (w2 - 1, w2) of the spinner's shape only, and slot w2 - 1 has no other pair (the
          // force on it is taken out of its wrench in one subtraction, as the generic sum fin - (0 + f) is)
          bool fine = fast_shape == 5 && std::min(sa[pi], sb[pi]) == w2 - 1 && std::max(sa[pi], sb[pi]) == w2;
          for (int pj : mine) fine = fine && (pj == pi || (sa[pj] != w2 - 1 && sb[pj] != w2 - 1));
          if (!fine) { fast_shape = 0; break; }
          continue;
        }
        if (nchainb == 1 && (sa[pi] == -1 || sb[pi] == -1)) {
          const int sl = std::max(sa[pi], sb[pi]);
          if (sl < last_slot) { fast_shape = 0; break; }   // (slot, index) order != index order on the common body's sum
          last_slot = sl;
          lo_chain_common = std::min(lo_chain_common, pi);
          hi_chain_common = std::max(hi_chain_common, pi);
        }
      }
      if (!fast_shape) break;
      std::vector<std::vector<int>> groups(maxc + 2);
      for (int pi : mine) {
        const int sl = std::max(sa[pi], sb[pi]);
        if (sl >= 0) { groups[1 + sl].push_back(pi); continue; }
        const bool touches_common = sa[pi] == -1 || sb[pi] == -1;
        if (!touches_common || (sa[pi] == -1 && sb[pi] == -1)) { fast_shape = 0; break; }   // (world, world) / (common, common)
        if (pi < lo_chain_common) groups[0].push_back(pi);
        else if (pi > hi_chain_common) groups[maxc + 1].push_back(pi);
        else { fast_shape = 0; break; }
      }
      if (!fast_shape) break;
      for (int gi = 0; gi < maxc + 2; ++gi) {
        segw[(size_t)p * (maxc + 2) + gi] = (int)order[p].size() | ((int)groups[gi].size() << 16);
        for (int pi : groups[gi]) order[p].push_back(pi);
      }
    }
    if (fast_shape) {
      auto ident_mul = [](const double* X, double* out) {   // [I * R | I * p] with the fused forms of dev_math.h
        static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int r = 0; r < 3; ++r) {
          for (int cc = 0; cc < 3; ++cc)
            out[3 * r + cc] = std::fma(I[3 * r + 2], X[6 + cc], std::fma(I[3 * r + 1], X[3 + cc], I[3 * r] * X[cc]));
          out[9 + r] = std::fma(I[3 * r + 2], X[11], std::fma(I[3 * r + 1], X[10], I[3 * r] * X[9]));
        }
      };
      auto body_record = [&](int b, bool world, double* rec) {
        if (world) ident_mul(m->X_PF + (size_t)12 * b, rec + FB_XPF);
        else std::memcpy(rec + FB_XPF, m->X_PF + (size_t)12 * b, 12 * sizeof(double));
        std::memcpy(rec + FB_AXIS, m->axis + (size_t)3 * b, 3 * sizeof(double));
        rec[FB_MASS] = m->mass[b];
        std::memcpy(rec + FB_COM, m->com + (size_t)3 * b, 3 * sizeof(double));
        std::memcpy(rec + FB_INERTIA, m->inertia + (size_t)6 * b, 6 * sizeof(double));
        const int ndof = m->jtype[b] == IDTO_JOINT_FLOATING ? 6 : (m->jtype[b] == IDTO_JOINT_PLANAR ? 3 : 1);
        for (int d = 0; d < ndof; ++d) rec[FB_DAMP + d] = m->damping[m->vstart[b] + d];
        const int ix[2] = {m->qstart[b], m->vstart[b]};
        std::memcpy(rec + FB_IDX, ix, sizeof(ix));
      };
      std::vector<double> fb((size_t)K * maxc * FB_STRIDE, 0.0), fc(FB_STRIDE, 0.0);
      for (int p = 0; p < K; ++p)
        for (int s = 0; s < maxc; ++s)
          body_record(chain[(size_t)p * IDTO_MAX_CHAIN + s], pkind[(size_t)p * IDTO_MAX_CHAIN + s] == PK_WORLD,
                      fb.data() + ((size_t)p * maxc + s) * FB_STRIDE);
      if (cbody >= 0) body_record(cbody, true, fc.data());
      for (int p = 0; p < K; ++p) f_maxpp = std::max(f_maxpp, (int)order[p].size());
      std::vector<double> fp((size_t)K * f_maxpp * FP_STRIDE, 0.0);
      for (int p = 0; p < K; ++p)
        for (size_t j = 0; j < order[p].size(); ++j) {
          const int pi = order[p][j], ga = m->pair_a[pi], gb = m->pair_b[pi];
          double* rec = fp.data() + ((size_t)p * f_maxpp + j) * FP_STRIDE;
          // C: the chain body of the pair's group, or the common body for a pair without one; the other body is
          // the common one or the world
          const bool a_chain = sa[pi] >= 0, b_chain = sb[pi] >= 0;
          // (two chain bodies - the spinner's shape: C is the body of the later slot, the other one is handed to
          // pair_eval where the common body goes)
          const bool cia = (a_chain && b_chain) ? sa[pi] > sb[pi] : (a_chain || (!b_chain && sa[pi] == -1));
          const int gc = cia ? ga : gb, go = cia ? gb : ga, so = cia ? sb[pi] : sa[pi];
          const int info[4] = {m->geom_type[gc], m->geom_type[go], cia ? 1 : 0, (so == -1 || so >= 0) ? 1 : 0};
          std::memcpy(rec + FP_INFO, info, sizeof(info));
          std::memcpy(rec + FP_XC, m->geom_X + (size_t)12 * gc, 12 * sizeof(double));
          std::memcpy(rec + FP_SC, m->geom_size + (size_t)3 * gc, 3 * sizeof(double));
          if (so == -2) {   // world: [I R | 0 + I p], the expressions id_eval.h evaluates for a world-fixed geometry
            ident_mul(m->geom_X + (size_t)12 * go, rec + FP_XO);
            for (int e = 0; e < 3; ++e) rec[FP_XO + 9 + e] = 0.0 + rec[FP_XO + 9 + e];
          } else {
            std::memcpy(rec + FP_XO, m->geom_X + (size_t)12 * go, 12 * sizeof(double));
          }
          std::memcpy(rec + FP_SO, m->geom_size + (size_t)3 * go, 3 * sizeof(double));
        }
      o_fbody = addd(fb.data(), fb.size());
      o_fcbody = addd(fc.data(), fc.size());
      o_fpairs = addd(fp.data(), fp.size());
    }
    // the int tables, those fd_kernel needs beside the gathered records first: with a fast shape it stages only
    // [fast_lo, fast_lo + fast_n) of the blob in LDS
    fast_lo = fast_shape ? o_fbody : 0;
  }
  const size_t i_jt = addi(m->jtype, nb), i_qs = addi(m->qstart, nb), i_vs = addi(m->vstart, nb);
  const size_t i_colinfo = addi(colinfo.data(), colinfo.size()), i_rowinfo = addi(rowinfo.data(), rowinfo.size());
  const size_t i_fseg = addi(fseg.data(), fseg.size());
  const size_t i_fast_end = ints.size();
  const size_t i_par = addi(m->parent, nb), i_gt = addi(m->geom_type, m->ngeoms), i_ch = addi(chain.data(), chain.size()),
               i_nch = addi(nchain.data(), nchain.size()), i_pk = addi(pkind.data(), pkind.size()),
               i_pnp = addi(path_npairs.data(), path_npairs.size()), i_pp = addi(path_pairs.data(), path_pairs.size()),
               i_ga = addi(m->pair_a, m->npairs), i_gb = addi(m->pair_b, m->npairs), i_sa = addi(sa.data(), sa.size()),
               i_sb = addi(sb.data(), sb.size());
  const size_t nd = dbl.size(), ni = ints.size();
  std::vector<double> blob(nd + (ni + 1) / 2 + 1, 0.0);
  std::memcpy(blob.data(), dbl.data(), nd * sizeof(double));
  std::memcpy(blob.data() + nd, ints.data(), ni * sizeof(int));
  double* bd = nullptr;
  if (Upload(c, blob.data(), blob.size(), &bd)) return -2;
  const int* bi = reinterpret_cast<const int*>(bd + nd);
  M.blob = bd; M.blob_n = (int)blob.size();
  M.X_PF = bd + o_XPF; M.axis = bd + o_axis; M.mass = bd + o_mass; M.com = bd + o_com; M.inertia = bd + o_in;
  M.damping = bd + o_damp; M.geom_X = bd + o_gX; M.geom_size = bd + o_gs;
  M.parent = bi + i_par; M.jtype = bi + i_jt; M.qstart = bi + i_qs; M.vstart = bi + i_vs; M.geom_type = bi + i_gt;
  M.chain = bi + i_ch; M.nchain = bi + i_nch; M.pkind = bi + i_pk; M.path_npairs = bi + i_pnp; M.path_pairs = bi + i_pp;
  M.pair_ga = bi + i_ga; M.pair_gb = bi + i_gb; M.pair_sa = bi + i_sa; M.pair_sb = bi + i_sb;
  M.nfloat = nfloat;
  for (int i = 0; i < 4; ++i) { M.float_qs[i] = float_qs[i]; M.float_vs[i] = float_vs[i]; }
  M.nplus_const = bd + o_npc; M.colinfo = bi + i_colinfo; M.rowinfo = bi + i_rowinfo;
  M.fast_shape = fast_shape; M.f_maxpp = f_maxpp;
  M.fast_lo = (int)fast_lo; M.fast_n = (int)(nd + (i_fast_end + 1) / 2 - fast_lo);
  M.f_body = bd + o_fbody; M.f_cbody = bd + o_fcbody; M.f_pairs = bd + o_fpairs; M.f_seg = bi + i_fseg;
  return 0;
}


int FdEvals(const idto_hip_ctx* c, int mode) {
  return (mode == 0) ? 1 : ((mode == 1) ? 1 + 2 * c->nq + c->nv : 1 + ((mode == 2) ? 2 : 4) * 3 * c->nq);
}

// dynamic LDS of fd_kernel when it builds the inputs of `ec` evaluations per pass
// (fast: fd_kernel<MAXC, SHAPE != 0>, which stages only the shape's records of the model; the fused launch runs the
// generic evaluation and stages the whole blob)
int FdLds(const idto_hip_ctx* c, int mode, int ec, bool with_terms = false, bool fast = true) {
  const int nq = c->nq, nv = c->nv, E = FdEvals(c, mode), nvp = (nv + 1) & ~1;
  const int rec = with_terms ? 6 * nvp * nq + nvp + 1 : 0;   // the record, its weighted copy, diag R' (+1: 16-byte alignment)
  const int blob_n = (fast && c->fd_fast && c->M.fast_shape) ? c->M.fast_n : c->M.blob_n;   // what fd_body stages of the model
  return (int)sizeof(double) * (3 * nq + 2 * nv * nq + 3 * nv + 3 * E + E * nv + ec * (nq + 2 * nv) + nv + blob_n + 2 + nq / 2 + 2 + rec);
}

// fd_kernel also forms the single-record products of the Gauss-Newton assembly (diagonal weights,
// derivatives requested): grad_hess then only combines them
static bool FoldTerms(const idto_hip_ctx* c, int mode) { return c->asm_fold && c->weights_diagonal && mode >= 1; }

int LaunchFd(idto_hip_ctx* c, int mode, int kb, int ke, AltSel alt = AltSel{nullptr, 0, 0}) {
  c->partials_ahead = false;   // (whatever idto_hip_eval_tau_partials left is about to be overwritten; it sets the flag after its own launch)
  if (ke <= kb) return 0;
  if (mode >= 1) mode = 1 + c->gradients_method;  // 1 forward, 2 central, 3 central (4th order)
  dim3 grid(ke - kb, c->batch), block(mode >= 1 ? c->fd_threads : 64);
  // evaluations per pass: all of them if they fit in LDS, otherwise the largest multiple of
  // the number of evaluations the block runs concurrently
  const int E = FdEvals(c, mode), groups = (int)block.x / c->npaths;
  int ec = E;
  bool fold = FoldTerms(c, mode);
  if (fold && FdLds(c, mode, std::min(ec, groups), true) > 160 * 1024) fold = false;
  while (ec > groups && FdLds(c, mode, ec, fold) > 160 * 1024) ec = ((ec - 1) / groups) * groups;
  const int lds = FdLds(c, mode, ec, fold);
  if (lds > 160 * 1024) { g_err = "finite-difference evaluation set does not fit in LDS"; return -1; }
  double* terms = fold ? c->terms : nullptr;
  if (mode >= 1) c->terms_valid = fold && kb == 0 && ke == c->N;
  FdLaunch fl;
  fl.grid = grid; fl.block = block; fl.lds = lds; fl.stream = c->stream; fl.M = c->M; fl.cp = c->cp; fl.P = c->P;
  fl.q = c->q; fl.slab = c->slab; fl.slab_stride = c->slab_stride; fl.v = c->v; fl.a = c->a; fl.nplus = c->nplus;
  fl.k_begin = kb; fl.mode = mode; fl.stop_after = c->fd_stop; fl.echunk = ec; fl.pstride = c->pstride; fl.terms = terms;
  fl.alt = alt;
  fl.shape = c->fd_fast ? c->M.fast_shape : 0;   // id_fast.h: the straight-line evaluation of the model's tree shape
  fl.maxc = c->maxc;
  fd_launch(fl);
  HIP_OK(hipGetLastError());
  return 0;
}

int TimeBegin(idto_hip_ctx* c, int which) {
  c->timing_open = false;
  if (!c->timing) return 0;
  if ((c->timing_tick[which]++ % (unsigned)c->timing_stride) != 0) return 0;
  c->timing_open = true;
  idto_hip_ctx::Timed t;
  t.which = which;
  hipEvent_t* ev[2] = {&t.a, &t.b};
  for (hipEvent_t* e : ev) {
    if (c->event_pool.empty()) {
      HIP_OK(hipEventCreate(e));
    } else {
      *e = c->event_pool.back();
      c->event_pool.pop_back();
    }
  }
  HIP_OK(hipEventRecord(t.a, c->stream));
  c->pending.push_back(t);
  return 0;
}
int TimeEnd(idto_hip_ctx* c) {
  if (!c->timing || !c->timing_open) return 0;
  c->timing_open = false;
  HIP_OK(hipEventRecord(c->pending.back().b, c->stream));
  return 0;
}
int TimeDrain(idto_hip_ctx* c) {
  for (auto& t : c->pending) {
    HIP_OK(hipEventSynchronize(t.b));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, t.a, t.b));
    c->tsum[t.which] += ms;
    c->tcnt[t.which] += 1;
    c->event_pool.push_back(t.a);
    c->event_pool.push_back(t.b);
  }
  c->pending.clear();
  return 0;
}

int EnsureStage(idto_hip_ctx* c, size_t count) {
  if (count > c->stage_count) {  // staging buffers grow on demand and are reused
    double *a = nullptr, *b = nullptr;
    if (Alloc(c, count, &a) || Alloc(c, count, &b)) return -2;
    c->stage_rhs = a; c->stage_x = b; c->stage_count = count;
  }
  return 0;
}
// a pending idto_hip_prefetch of an array is dropped when that array is about to be recomputed
// (idto_hip_get then reads the new contents instead of the stale staged copy)
// after a stream synchronisation: did the most recent factorisation report a failed pivot?
// (problem `pb` of the batch, or any of them for pb < 0)
int FactorStatus(idto_hip_ctx* c, int pb = -1) {
  const volatile unsigned* st = c->status_pin;
  // (the COUNT of launches whose waits ran out, not the id of the last one: idto_hip_tr_solve enqueues a factorisation
  // per iteration and waits once - a timeout in any of them shows here; and a second status query about a launch
  // that has been handled does not step the context down once more)
  const unsigned timeouts_now = st[2 * c->batch + 1];
  if (timeouts_now != c->timeouts_handled) {
    const bool iteration_kernel = ((timeouts_now ^ c->timeouts_handled) >> 16) != 0u;   // (trust_region.h tr_iter_kernel's own wait)
    c->timeouts_handled = timeouts_now;
    if (iteration_kernel) {
      // the workgroups of tr_iter_kernel did not find each other resident within their bound: the resident loop is off for
      // the rest of the context's life (option "tr_resident_ok" reads 0), the caller's loop returns to the host twice an
      // iteration - kernels without a wait between workgroups
      c->tr_resident_ok = false;
      ++c->solver_timeouts;
      g_err = "the trust-region iteration's workgroups timed out waiting for each other (device shared with other kernels); "
              "the resident loop is off for this context: repeat the call";
      return IDTO_HIP_SOLVER_TIMEOUT;
    }
    // A wait between the workgroups of a multi-workgroup solver ran out: its partners were not resident at the
    // same time (other contexts' kernels on the device).  The result of that launch is garbage.  Step down to a
    // variant with fewer co-resident workgroups for the rest of the context's life; the caller repeats the solve
    // (idto_hip_get(STEP) and idto_hip_tr_prepare do it themselves).
    if (c->solver_band > 0 && c->last_solver == 6) c->solver_band = 0;   // (its wait for the launch's own assembly)
    else if (c->solver_pipe && c->last_solver == 4) c->solver_pipe = false;
    else if (c->solver_nd && (c->last_solver == 2 || c->last_solver == 4)) c->solver_nd = false;
    else if (c->fused && c->last_solver == 5) c->fused = false;
    else if (c->two_sided) { c->solver_nd = false; c->fused = false; c->two_sided = false; }
    ++c->solver_timeouts;
    g_err = "a solver launch timed out waiting for a partner workgroup (device shared with other kernels); the "
            "context has stepped down to a variant with fewer co-resident workgroups: repeat the call";
    return IDTO_HIP_SOLVER_TIMEOUT;
  }
  bool any = false;
  for (int b = (pb < 0 ? 0 : pb); b < (pb < 0 ? c->batch : pb + 1); ++b) any = any || st[2 * b] == c->fact_id;
  if (c->fact_id != 0 && any) {
    g_err = "factorisation failed: the Hessian is not numerically positive definite (a pivot was non-positive, "
            "non-finite or fully cancelled)";
    return IDTO_HIP_FACTORIZATION_FAILED;
  }
  return 0;
}
void* DevPtr(idto_hip_ctx* c, int what);
void DropPrefetch(idto_hip_ctx* c, std::initializer_list<int> arrays) {
  for (auto& pf : c->pre)
    for (int a : arrays)
      if (pf.what == a) pf.pending = false;
}
}  // namespace

extern "C" {

const char* idto_hip_last_error(void) { return g_err.c_str(); }

int idto_hip_create(const idto_model_t* model, const idto_problem_t* problem, const idto_contact_params_t* contact,
                    int device, idto_hip_ctx** out) {
  return idto_hip_create_batch(model, problem, contact, device, 1, out);
}

int idto_hip_create_batch(const idto_model_t* model, const idto_problem_t* problems, const idto_contact_params_t* contact,
                          int device, int batch, idto_hip_ctx** out) {
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device available (the product path has no CPU fallback)";
    return -3;
  }
  if (batch < 1 || batch > 65535) { g_err = "batch must be in [1, 65535]"; return -1; }
  for (int b = 1; b < batch; ++b)
    if (problems[b].num_steps != problems[0].num_steps || problems[b].time_step != problems[0].time_step) {
      g_err = "all problems of a batch share num_steps and time_step";
      return -1;
    }
  const idto_problem_t* problem = problems;
  HIP_OK(hipSetDevice(device));
  idto_hip_ctx* c = new idto_hip_ctx();
  c->device = device;
  c->batch = batch;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_err = "hipStreamCreate failed"; delete c; return -2; }
  c->own_stream = true;
  c->nb = model->nbodies; c->nq = model->nq; c->nv = model->nv; c->N = problem->num_steps;
  c->dt = problem->time_step; c->npaths = model->npaths;
  const int nq = c->nq, nv = c->nv, N = c->N;
  int rc = BuildModel(c, model);
  if (rc) { idto_hip_destroy(c); return rc; }
  if (batch > 1) {
    c->host_model = std::make_unique<HostModelCopy>();
    c->host_model->Set(*model);
    for (int b = 0; b < batch; ++b) {
      c->host_problems.push_back(std::make_unique<HostProblemCopy>());
      c->host_problems.back()->Set(problems[b], model->nq, model->nv);
    }
    c->host_contact = *contact;
  }
  c->cp.k = contact->contact_stiffness; c->cp.vd = contact->dissipation_velocity;
  c->cp.vs = contact->stiction_velocity; c->cp.mu = contact->friction_coefficient;
  c->cp.sigma = contact->smoothing_factor;
  {  // reference TO.cc:266-269, evaluated with the same deterministic exp/log as the device code
    const double eps = std::sqrt(2.220446049250313e-16);
    c->cp.threshold = -c->cp.sigma * idto::detmath::log(idto::detmath::exp(eps / (c->cp.sigma * c->cp.k)) - 1.0);
  }
  const size_t bsz = (size_t)nv * nq, qq = (size_t)nq * nq;
  c->slab_stride = (int)(3 * bsz + nv);
  // ---- one arena per problem (identical layout): carve offsets first, allocate batch * pstride
  // bytes at once (zero-filled), then point the problem-0 pointers into the first arena
  size_t top = 0;
  auto carve = [&](size_t count, size_t elem) {
    const size_t o = (top + 63) & ~(size_t)63;
    top = o + std::max<size_t>(count, 1) * elem;
    return o;
  };
  const size_t D = sizeof(double);
  const size_t o_vinit = carve(nv, D), o_qnom = carve((size_t)(N + 1) * nq, D), o_vnom = carve((size_t)(N + 1) * nv, D);
  size_t o_w[10];
  for (int i = 0; i < 10; ++i) o_w[i] = carve((i == 0 || i == 3 || i == 5 || i == 8) ? qq : (size_t)nv * nv, D);
  const size_t o_q = carve((size_t)(N + 1) * nq, D);
  // what fd_kernel writes, twice (batch.h AltSel): two identical carve sequences = identical relative offsets
  size_t o_v = 0, o_a = 0, o_np = 0, o_slab = 0, o_terms = 0, o_setB = 0;
  for (int set = 0; set < 2; ++set) {
    const size_t v0 = carve((size_t)(N + 1) * nv, D), a0 = carve((size_t)N * nv, D), n0 = carve((size_t)(N + 1) * bsz, D);
    const size_t s0 = carve((size_t)(N + IDTO_SLAB_PAD) * c->slab_stride, D), t0 = carve((size_t)N * asm_terms_stride(nq), D);
    if (set == 0) { o_v = v0; o_a = a0; o_np = n0; o_slab = s0; o_terms = t0; }
    else o_setB = v0;
  }
  c->alt_off = (long long)(o_setB - o_v);
  const size_t o_g = carve((size_t)(N + 1) * nq, D);
  // two extra zero blocks (five are reserved): the solver prefetches rows i+1, i+2 without bounds
  // checks (one allocation: the solver addresses all three bands from HA with 32-bit offsets)
  const size_t o_H = carve((size_t)3 * (N + 6) * qq, D);
  // (a second g and H: the trial point's, formed by the solver's launch while it decides on the point - penta_pipe.h PipeAsm::g2)
  const size_t o_H2 = carve((size_t)3 * (N + 6) * qq, D), o_g2 = carve((size_t)(N + 1) * nq, D);
  const size_t o_step = carve((size_t)(N + 1) * nq, D), o_cost = carve(1, D);
  const size_t o_K = carve((size_t)(N + 1) * qq, D), o_LU = carve((size_t)(N + 1) * qq, D);
  const size_t o_Y = carve((size_t)(N + 1) * qq, D), o_Z = carve((size_t)(N + 1) * qq, D);
  const size_t o_piv = carve((size_t)(N + 1) * nq, sizeof(int));
  const size_t o_dbg = carve((size_t)(N + 4) * 8 * 32, D);
  const size_t o_U = carve((size_t)(N + 1) * 32 * 36, D), o_Hs = carve((size_t)(N + 1) * 32 * 36, D);
  const size_t o_E = carve((size_t)(N + 1) * 32 * 36, D), o_Ds = carve((size_t)(N + 1) * 32, D);
  // exchange buffer of the two-sided solver (one right-hand side): 2 augmented blocks + [2][K]
  c->xch_count = 2 * (size_t)(3 * 32 + 1) * ldl_ks(32) + 2 * 32;
  const size_t o_xch = carve(2 * c->xch_count, D);   // (two producer / joiner pairs in the nested-dissection kernel)
  const size_t o_ndcnt = carve(4 * ND_MAXROWS, sizeof(unsigned long long)), o_ndbuf = carve((size_t)nd_layout(32).end, D);
  const size_t o_pipecnt = carve(4 * ND_MAXROWS, sizeof(unsigned long long));
  const bool nd_wst_on = nq > 20 && nq <= 32;
  const size_t o_ndwst = carve(nd_wst_on ? 2 * (size_t)ND_MAXROWS * nd_layout(nq).frow : 1, D);
  const size_t o_asmready = carve(4 * (size_t)(N + 1), sizeof(unsigned));
  c->flag_count = 16;
  const size_t o_flags = carve(c->flag_count, sizeof(unsigned)), o_sync = carve(2, sizeof(unsigned long long));
  const size_t nvars = (size_t)(N + 1) * nq;
  const size_t o_trD = carve(nvars, D), o_trg = carve(nvars, D), o_trw = carve(nvars, D), o_trdq = carve(nvars, D),
               o_qt = carve(nvars, D), o_trout = carve(16, D), o_trDp = carve(nvars, D), o_trpart = carve((size_t)9 * (N + 1), D),
               o_trpll = carve((size_t)2 * (TR_NSUM * (N + 1) + 3), D), o_trp2 = carve((size_t)2 * (N + 1), D),
               o_decide = carve(4, D);
  const size_t o_trstate = carve(TRS_COUNT, D), o_trcnt = carve(1, sizeof(unsigned long long));
  // the equality-constraint step's outputs (per problem, so that the batched loop finds them at the arena stride):
  // [H^-1 (g + J^T lambda) | J^T lambda] and the multipliers (nu <= nv; + 2: the blocked dense LDL^T's [min, max | ...])
  const size_t o_conout = carve(2 * nvars, D), o_conlam = carve((size_t)N * nv + 4, D);
  c->pstride = (top + 255) & ~(size_t)255;
  {
    void* p = nullptr;
    if (hipMalloc(&p, c->pstride * (size_t)batch) != hipSuccess || hipMemset(p, 0, c->pstride * (size_t)batch) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {   // (see Alloc: the fill must not trail the uploads on the context's stream)
      g_err = "hipMalloc of the problem arenas failed";
      idto_hip_destroy(c);
      return -2;
    }
    c->allocs.push_back(p);
    c->arena = static_cast<char*>(p);
  }
  auto dp = [&](size_t o) { return reinterpret_cast<double*>(c->arena + o); };
  c->d_vinit = dp(o_vinit); c->d_qnom = dp(o_qnom); c->d_vnom = dp(o_vnom);
  for (int i = 0; i < 10; ++i) c->d_w[i] = dp(o_w[i]);
  c->q = dp(o_q); c->v = dp(o_v); c->a = dp(o_a); c->nplus = dp(o_np); c->slab = dp(o_slab); c->g = dp(o_g);
  c->HA = dp(o_H);
  c->HB = c->HA + (size_t)(N + 6) * qq;
  c->HC = c->HB + (size_t)(N + 6) * qq;
  c->HA2 = dp(o_H2); c->HB2 = c->HA2 + (size_t)(N + 6) * qq; c->HC2 = c->HB2 + (size_t)(N + 6) * qq; c->g2 = dp(o_g2);
  c->step = dp(o_step); c->cost = dp(o_cost);
  c->Kst = dp(o_K); c->LUst = dp(o_LU); c->Yst = dp(o_Y); c->Zst = dp(o_Z);
  c->pivst = reinterpret_cast<int*>(c->arena + o_piv);
  c->dbg = dp(o_dbg);
  c->Ust = dp(o_U); c->Hst = dp(o_Hs); c->Est = dp(o_E); c->Dst = dp(o_Ds);
  c->xch = dp(o_xch);
  c->flags = reinterpret_cast<unsigned*>(c->arena + o_flags);
  c->sync_cnt = reinterpret_cast<unsigned long long*>(c->arena + o_sync);
  c->nd_rowcnt = reinterpret_cast<unsigned long long*>(c->arena + o_ndcnt);
  c->pipe_rowcnt = reinterpret_cast<unsigned long long*>(c->arena + o_pipecnt);
  c->asm_ready = reinterpret_cast<unsigned*>(c->arena + o_asmready);
  c->nd_buf = dp(o_ndbuf);
  c->nd_wst = nd_wst_on ? dp(o_ndwst) : nullptr;
  c->tr_D = dp(o_trD); c->tr_gt = dp(o_trg); c->tr_w = dp(o_trw); c->tr_dq = dp(o_trdq); c->q_trial = dp(o_qt);
  c->tr_out = dp(o_trout); c->tr_Dprev = dp(o_trDp); c->tr_part = dp(o_trpart);
  c->tr_part_ll = dp(o_trpll); c->tr_part2 = dp(o_trp2); c->decide_word = dp(o_decide);
  c->terms = dp(o_terms);
  c->tr_state = dp(o_trstate);
  c->con_out = dp(o_conout); c->con_lambda = dp(o_conlam);
  c->tr_cnt = reinterpret_cast<unsigned long long*>(c->arena + o_trcnt);
  {  // adaptive scaling methods start from D = 1 (TO.cc:1233-1236: scale_factors initialised to ones)
    std::vector<double> ones(nvars, 1.0);
    for (int b = 0; b < batch; ++b)
      if (hipMemcpy(at_problem(c->tr_D, (size_t)b * c->pstride), ones.data(), nvars * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(at_problem(c->tr_Dprev, (size_t)b * c->pstride), ones.data(), nvars * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        g_err = "hipMemcpy (scale factors) failed"; idto_hip_destroy(c); return -2;
      }
    std::vector<int> qs;
    for (int i = 0; i < model->nbodies; ++i)
      if (model->jtype[i] == IDTO_JOINT_FLOATING) qs.push_back(model->qstart[i]);
    c->tr_nquat = (int)qs.size();
    if (!qs.empty() && Upload(c, qs.data(), qs.size(), &c->tr_quat)) { idto_hip_destroy(c); return -2; }
    if (hipHostMalloc((void**)&c->tr_pin, 16 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
      g_err = "hipHostMalloc (trust-region scalars) failed";
      idto_hip_destroy(c);
      return -2;
    }
  }
  for (int b = 0; b < batch; ++b) {
    rc = UploadProblemArrays(c, problems + b, b);
    if (rc) { idto_hip_destroy(c); return rc; }
  }
  // per problem [id of the last failed factorisation, count]; behind them [id of the last launch in which a
  // wait between workgroups ran out, count] (penta_ldl.h spin_wait)
  if (hipHostMalloc((void**)&c->status_pin, (2 * (size_t)batch + 2) * sizeof(unsigned), hipHostMallocDefault) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->status_dev, c->status_pin, 0) != hipSuccess) {
    g_err = "hipHostMalloc (solver status) failed";
    idto_hip_destroy(c);
    return -2;
  }
  for (int i = 0; i < 2 * batch + 2; ++i) c->status_pin[i] = 0;
  c->k_begin = 0; c->k_end = N;

  // launch geometry
  const int K = c->npaths;
  const int E = 1 + 2 * nq + nv;
  int threads = ((E * K + 63) / 64) * 64;
  if (threads > 256) threads = 256;  // one wave per SIMD: the evaluation keeps its bodies in up to 512 VGPRs
  c->fd_threads = threads;
  // smallest LDS carve-up of the finite-difference kernel: the inputs of one round of concurrent
  // evaluations per pass (LaunchFd uses more when they fit)
  c->fd_lds = FdLds(c, 1, threads / K);
  c->asm_lds = (int)sizeof(double) * (3 * nq + 5 * (int)bsz + 4 * nv + nq + std::max(nq, nv) + (int)bsz + 3 * (int)qq + nq);
  c->asm_diag_lds = (int)sizeof(double) * (14 * ((nv + 1) & ~1) * nq + 2 * (int)bsz + 10 * nv + 6 * nq + 2);
  c->asm_terms_lds = (int)sizeof(double) * (5 * ((nv + 1) & ~1) * nq + 4 * nv + nq + 2);
  c->asm_terms_threads = std::min(512, std::max(256, (std::max(nq * (nq + 1) / 2 + nq, ((nq + 1) / 2) * nq) + 63) / 64 * 64));
  const int n = N + 1;
  c->penta_lds = (int)sizeof(double) * (10 * (int)qq + nq * (3 * nq + 1) + (n + 2) * nq + nq) + (int)sizeof(int) * nq + 16;
  c->solve_lds = (int)sizeof(double) * ((n + 2) * nq + nq);
  c->cost_lds = (int)sizeof(double) * ((3 * N + 2) * (1 + std::max(nq, nv)) + 2 * (N + 1) + 2);
  const int max_lds = 160 * 1024;
  // (cost_kernel is a single block holding one column of every cost term; penta_apply_kernel keeps
  // the right-hand side of each of its four wavefronts: both bound the horizon as well)
  const int Kpad = (nq == 2 || nq == 3 || nq == 5 || nq == 19 || nq == 23) ? nq : (nq <= 8 ? 8 : nq <= 16 ? 16 : nq <= 24 ? 24 : 32);
  const int apply_lds = 4 * (n * Kpad + 4 * 64 + 2) * (int)sizeof(double);
  if (c->fd_lds > max_lds || c->asm_lds > max_lds || c->penta_lds > max_lds || c->cost_lds > max_lds ||
      apply_lds > max_lds) {
    g_err = "problem too large for the 160 KiB LDS carve-up of the v1 kernels (fd / assemble / solver / cost / "
            "multi-rhs substitution)";
    idto_hip_destroy(c);
    return -1;
  }
  // kernels that need more than the default 64 KiB of dynamic LDS must opt in
  fd_set_max_lds(max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&assemble_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
#define BAND_ATTR(WM) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&penta_band_kernel<WM>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  BAND_ATTR(6) BAND_ATTR(9) BAND_ATTR(12) BAND_ATTR(15)
#undef BAND_ATTR
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_small_kernel<1, 6, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_small_kernel<5, 9, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_small_kernel<1, 6, 256, 6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_small_kernel<5, 9, 256, 9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_small_kernel<1, 6, 256, 9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_small_kernel<5, 9, 256, 12, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&penta_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&assemble_diag_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cost_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&assemble_terms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&constraint_lambda_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
#define APPLY_ATTR(KM) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&penta_apply_kernel<KM, (KM <= 8)>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  APPLY_ATTR(2) APPLY_ATTR(3) APPLY_ATTR(5) APPLY_ATTR(8) APPLY_ATTR(16) APPLY_ATTR(19) APPLY_ATTR(23) APPLY_ATTR(24) APPLY_ATTR(32)
#undef APPLY_ATTR
#define LDL_ATTR(KM, PD, GW)                                                                   \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&penta_ldl_kernel<KM, 256, PD, GW>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  LDL_ATTR(2, false, 1) LDL_ATTR(3, false, 1) LDL_ATTR(5, false, 1) LDL_ATTR(19, false, 1) LDL_ATTR(23, false, 2)
  LDL_ATTR(8, true, 1) LDL_ATTR(16, true, 1) LDL_ATTR(24, true, 2) LDL_ATTR(30, true, 2) LDL_ATTR(29, false, 2) LDL_ATTR(4, false, 1) LDL_ATTR(32, true, 3)
#undef LDL_ATTR
#define ND_ATTR(KM, PD) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&penta_nd_kernel<KM, PD>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  ND_ATTR(2, false) ND_ATTR(3, false) ND_ATTR(5, false) ND_ATTR(19, false) ND_ATTR(23, false) ND_ATTR(29, false) ND_ATTR(8, false)
#undef ND_ATTR
#define PIPE_ATTR(KM) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&penta_pipe_kernel<KM>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&penta_pipe_kernel<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&penta_pipe_kernel<19, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  PIPE_ATTR(2) PIPE_ATTR(3) PIPE_ATTR(5) PIPE_ATTR(19)
#undef PIPE_ATTR
#define FUSED_ATTR(MC, KM, PD, GW)                                                                 \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_fused_kernel<MC, KM, PD, GW>),       \
                            hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
  FUSED_ATTR(2, 2, false, 1) FUSED_ATTR(3, 3, false, 1) FUSED_ATTR(3, 5, false, 1) FUSED_ATTR(3, 19, false, 1)
  FUSED_ATTR(4, 23, false, 2)
#undef FUSED_ATTR
  if (const char* e = getenv("IDTO_SOLVER_REFERENCE")) c->reference_solver = (e[0] == '1');
  if (const char* e = getenv("IDTO_TWO_SIDED")) c->two_sided = (e[0] == '1');   // (debugging aids: option defaults)
  if (const char* e = getenv("IDTO_FUSED")) c->fused = (e[0] == '1');
  if (const char* e = getenv("IDTO_ND_MIN_ROWS")) c->nd_min_rows = std::max(16, std::atoi(e));
  if (const char* e = getenv("IDTO_ND_RECURSION")) c->nd_recursion = std::atoi(e) != 0;
  if (const char* e = getenv("IDTO_SOLVER_ND")) c->solver_nd = (e[0] == '1');
  if (const char* e = getenv("IDTO_SOLVER_PIPE")) c->solver_pipe = (e[0] == '1');
  if (const char* e = getenv("IDTO_ASM_FOLD")) c->asm_fold = (e[0] == '1');
  if (const char* e = getenv("IDTO_CON_KKT")) c->con_kkt = (e[0] == '1');
  if (const char* e = getenv("IDTO_KKT_FOLD")) c->kkt_fold = (e[0] == '1');
  if (const char* e = getenv("IDTO_DEBUG_SKIP_ROLE")) c->debug_skip_role = std::atoi(e);   // test aid (tests/test_gpu_timeout.py)
  if (const char* e = getenv("IDTO_TR_SMALL")) c->tr_small = (e[0] == '1');
  if (const char* e = getenv("IDTO_TR_FOLD")) c->tr_fold = (e[0] == '1');
  if (const char* e = getenv("IDTO_KKT_IN_ASM")) c->kkt_in_asm = (e[0] == '1');
  if (const char* e = getenv("IDTO_DECIDE_IN_SOLVER")) c->decide_in_solver = (e[0] == '1');
  if (const char* e = getenv("IDTO_SOLVER_BAND")) c->solver_band = std::atoi(e);   // (measurement aid: penta_band.h off / on / on for blocks of 5 too)
  (void)hipGetLastError();
  *out = c;
  return 0;
}

void idto_hip_destroy(idto_hip_ctx* c) {
  if (!c) return;
  for (idto_hip_ctx* ch : c->children) idto_hip_destroy(ch);
  c->children.clear();
  if (c->kkt) { idto_hip_destroy(c->kkt); c->kkt = nullptr; }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) { if (RcclState().handle) (void)RcclState().CommDestroy(c->comm); c->comm = nullptr; }
  (void)TimeDrain(c);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->prob_pin) (void)hipHostFree(c->prob_pin);
  if (c->many_pin) (void)hipHostFree(c->many_pin);
  if (c->rows_pin) (void)hipHostFree(c->rows_pin);
  if (c->status_pin) (void)hipHostFree(c->status_pin);
  if (c->tr_pin) (void)hipHostFree(c->tr_pin);
  if (c->spec_ev) (void)hipEventDestroy(c->spec_ev);
  if (c->con_pin) (void)hipHostFree(c->con_pin);
  if (c->fetch_pin) (void)hipHostFree(c->fetch_pin);
  for (int i = 0; i < 2; ++i) {
    if (c->up_pin[i]) (void)hipHostFree(c->up_pin[i]);
    if (c->up_ev[i]) (void)hipEventDestroy(c->up_ev[i]);
  }
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  for (auto& pf : c->pre) if (pf.ev) (void)hipEventDestroy(pf.ev);
  if (c->pre_pin) (void)hipHostFree(c->pre_pin);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int idto_hip_set_problem(idto_hip_ctx* c, const idto_problem_t* p) { return idto_hip_set_problem_batch(c, 0, p); }

int idto_hip_set_problem_batch(idto_hip_ctx* c, int b, const idto_problem_t* p) {
  if (b < 0 || b >= c->batch) { g_err = "problem index outside the batch"; return -1; }
  if (p->num_steps != c->N || p->time_step != c->dt) { g_err = "num_steps / time_step cannot change"; return -1; }
  HIP_OK(hipSetDevice(c->device));
  c->fd_full = false; c->partials_ahead = false;  // v_0 = v_init
  c->con_ready = false; c->con_begun = false;
  if (b < (int)c->host_problems.size()) {
    c->host_problems[b]->Set(*p, c->nq, c->nv);
    if (b < (int)c->children.size() && c->children[b]) {
      const int rc = idto_hip_set_problem_batch(c->children[b], 0, &c->host_problems[b]->p);
      if (rc) return rc;
    }
  }
  return UploadProblemArrays(c, p, b);
}

int idto_hip_batch_size(idto_hip_ctx* c) { return c->batch; }

int idto_hip_set_stream(idto_hip_ctx* c, void* s) {
  if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  c->stream = static_cast<hipStream_t>(s);
  c->own_stream = false;
  return 0;
}
void* idto_hip_get_stream(idto_hip_ctx* c) { return c->stream; }

int idto_hip_set_shard(idto_hip_ctx* c, int kb, int ke) {
  if (kb < 0 || ke > c->N || kb > ke) { g_err = "bad shard range"; return -1; }
  c->k_begin = kb; c->k_end = ke;
  return 0;
}

int idto_hip_set_q(idto_hip_ctx* c, const double* q_host) {
  TRACE("hip: set_q begins");
  HIP_OK(hipSetDevice(c->device));
  c->trial_resident = false; c->spec_pending = false; c->spec_ready = false;
  DropPrefetch(c, {IDTO_ARR_Q});
  c->fd_full = false; c->partials_ahead = false;
  c->con_ready = false; c->con_begun = false;
  const size_t qbytes = (size_t)(c->N + 1) * c->nq * sizeof(double);
  if (c->async_uploads) {
    char* buf = nullptr; int slot = 0;
    if (int rc = UploadStage(c, qbytes, &buf, &slot)) return rc;
    std::memcpy(buf, q_host, qbytes);
    HIP_OK(hipMemcpyAsync(c->q, buf, qbytes, hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipEventRecord(c->up_ev[slot], c->stream));
    TRACE("hip: set_q done (copy enqueued)");
    return 0;
  }
  HIP_OK(hipMemcpyAsync(c->q, q_host, qbytes, hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));  // the host buffer may be reused by the caller
  TRACE("hip: set_q done (copy + wait)");
  return 0;
}
int idto_hip_set_q_batch(idto_hip_ctx* c, const double* q_host) {
  HIP_OK(hipSetDevice(c->device));
  c->trial_resident = false; c->spec_pending = false; c->spec_ready = false;
  DropPrefetch(c, {IDTO_ARR_Q});
  c->fd_full = false; c->partials_ahead = false;
  c->con_ready = false; c->con_begun = false;
  const size_t row = (size_t)(c->N + 1) * c->nq * sizeof(double);  // one trajectory per arena
  HIP_OK(hipMemcpy2DAsync(c->q, c->pstride, q_host, row, row, (size_t)c->batch, hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));
  return 0;
}
int idto_hip_set_q_device(idto_hip_ctx* c, const double* q_dev) {
  HIP_OK(hipSetDevice(c->device));
  c->trial_resident = false; c->spec_pending = false; c->spec_ready = false;
  DropPrefetch(c, {IDTO_ARR_Q});
  c->fd_full = false; c->partials_ahead = false;
  c->con_ready = false; c->con_begun = false;
  HIP_OK(hipMemcpyAsync(c->q, q_dev, (size_t)(c->N + 1) * c->nq * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int idto_hip_eval_tau(idto_hip_ctx* c) {
  HIP_OK(hipSetDevice(c->device));
  c->partials_ahead = false;
  c->trial_resident = false; c->spec_pending = false; c->spec_ready = false;
  DropPrefetch(c, {IDTO_ARR_V, IDTO_ARR_A, IDTO_ARR_NPLUS, IDTO_ARR_SLAB, IDTO_ARR_COST});
  int rc = LaunchFd(c, 0, 0, c->N);
  if (rc) return rc;
  hipLaunchKernelGGL(cost_kernel, dim3(1, c->batch), dim3(1024), c->cost_lds, c->stream, c->M, c->P, c->q, c->v, c->slab,
                     c->slab_stride, c->cost, c->weights_diagonal ? 1 : 0, (double*)nullptr, c->pstride, (double*)nullptr,
                     TrDecideArgs{}, AltSel{nullptr, 0, 0});
  HIP_OK(hipGetLastError());
  return 0;
}

// idto_hip_eval_tau for a q whose partials are wanted next (the start of idto_hip_tr_solve / idto_hip_gn_step): ONE
// finite-difference launch leaves v, a, N+, tau AND the partials (fd_kernel's derivative modes evaluate the nominal
// point too: same tau, bit for bit), the cost follows; the idto_hip_eval_partials that comes next returns at once.
int idto_hip_eval_tau_partials(idto_hip_ctx* c) {
  HIP_OK(hipSetDevice(c->device));
  if (!(c->k_begin == 0 && c->k_end == c->N)) return idto_hip_eval_tau(c);   // (a shard of the horizon: tau of every step is needed)
  c->trial_resident = false; c->spec_pending = false; c->spec_ready = false;
  DropPrefetch(c, {IDTO_ARR_V, IDTO_ARR_A, IDTO_ARR_NPLUS, IDTO_ARR_SLAB, IDTO_ARR_COST});
  c->con_ready = false; c->con_begun = false;
  int rc = LaunchFd(c, 1, 0, c->N);
  if (rc) return rc;
  c->fd_full = true;
  c->partials_ahead = true;
  TRACE("hip: eval_tau_partials: fd_kernel enqueued");
  hipLaunchKernelGGL(cost_kernel, dim3(1, c->batch), dim3(1024), c->cost_lds, c->stream, c->M, c->P, c->q, c->v, c->slab,
                     c->slab_stride, c->cost, c->weights_diagonal ? 1 : 0, (double*)nullptr, c->pstride, (double*)nullptr,
                     TrDecideArgs{}, AltSel{nullptr, 0, 0});
  HIP_OK(hipGetLastError());
  return 0;
}

int idto_hip_trial_cost(idto_hip_ctx* c, const double* q_host, double* tau_host, double* cost_host) {
  HIP_OK(hipSetDevice(c->device));
  if (c->batch != 1) { g_err = "trial_cost serves single-problem contexts"; return -1; }
  if (!q_host || !cost_host) { g_err = "trial_cost: bad arguments"; return -1; }
  DropPrefetch(c, {IDTO_ARR_Q, IDTO_ARR_V, IDTO_ARR_A, IDTO_ARR_NPLUS, IDTO_ARR_SLAB, IDTO_ARR_COST});
  c->trial_resident = false; c->spec_pending = false; c->spec_ready = false;
  const size_t nq_all = (size_t)(c->N + 1) * c->nq, ntau = (size_t)c->N * c->nv;
  if (!c->pin) {
    if (Alloc(c, ntau + 1, &c->pack)) return -2;
    HIP_OK(hipHostMalloc((void**)&c->pin, (nq_all + ntau + 1) * sizeof(double), hipHostMallocDefault));
  }
  // one stream synchronisation for the whole trial point: q through pinned memory, N
  // inverse-dynamics evaluations, the cost, and [tau | cost] back in one copy
  std::memcpy(c->pin, q_host, nq_all * sizeof(double));
  c->fd_full = false; c->partials_ahead = false;
  c->con_ready = false; c->con_begun = false;
  HIP_OK(hipMemcpyAsync(c->q, c->pin, nq_all * sizeof(double), hipMemcpyHostToDevice, c->stream));
  int rc = LaunchFd(c, 0, 0, c->N);
  if (rc) return rc;
  hipLaunchKernelGGL(cost_kernel, dim3(1), dim3(1024), c->cost_lds, c->stream, c->M, c->P, c->q, c->v, c->slab,
                     c->slab_stride, c->cost, c->weights_diagonal ? 1 : 0, c->pack, (size_t)0, (double*)nullptr, TrDecideArgs{}, AltSel{nullptr, 0, 0});
  HIP_OK(hipGetLastError());
  double* out = c->pin + nq_all;
  HIP_OK(hipMemcpyAsync(out, c->pack, (ntau + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));
  if (tau_host) std::memcpy(tau_host, out, ntau * sizeof(double));
  *cost_host = out[ntau];
  return 0;
}

int idto_hip_eval_partials(idto_hip_ctx* c) {
  HIP_OK(hipSetDevice(c->device));
  if (c->partials_ahead && c->k_begin == 0 && c->k_end == c->N) {   // (idto_hip_eval_tau_partials: they are there)
    c->partials_ahead = false;
    return 0;
  }
  c->partials_ahead = false;
  DropPrefetch(c, {IDTO_ARR_V, IDTO_ARR_A, IDTO_ARR_NPLUS, IDTO_ARR_SLAB});
  c->con_ready = false; c->con_begun = false;
  if (TimeBegin(c, 0)) return -2;
  int rc = LaunchFd(c, 1, c->k_begin, c->k_end);
  if (rc) return rc;
  c->fd_full = (c->k_begin == 0 && c->k_end == c->N);
  return TimeEnd(c);
}

// g and the bands of H from the resident slab / products (gate: see assemble_terms_kernel)
// sink: (the constrained loop) the banded KKT system's bands and right-hand side are written along (kernels.h KktSink) -
// by the kernel that combines the records' products only; *sink_used says whether that was the one
static int LaunchAssemble(idto_hip_ctx* c, const double* gate, const KktSink* sink = nullptr, size_t kstride = 0, bool* sink_used = nullptr) {
  if (!c->h_assembled) {  // x_0 = -g_0 = 0 is no longer written by the solver (SolverFirstRow)
    HIP_OK(hipMemset2DAsync(c->step, c->pstride, 0, (size_t)c->nq * sizeof(double), (size_t)c->batch, c->stream));
    c->h_assembled = true;
  }
  c->con_ready = false; c->con_begun = false;
  const bool combine = c->weights_diagonal && c->terms_valid && c->fd_full && c->asm_stop == 0;
  c->last_assembly = combine ? 1 : (c->weights_diagonal ? 2 : 3);
  if (sink_used) *sink_used = combine && sink;
  if (combine) {
    hipLaunchKernelGGL(assemble_terms_kernel, dim3(c->N + 1, 4, c->batch), dim3(c->asm_terms_threads), c->asm_terms_lds, c->stream, c->M, c->P,
                       c->q, c->terms, c->v, c->nplus, c->g, c->HA, c->HB, c->HC, c->pstride, gate, c->alt_r,
                       sink ? *sink : KktSink{}, kstride);
  } else if (c->weights_diagonal) {
    hipLaunchKernelGGL(assemble_diag_kernel, dim3(c->N + 1, 4, c->batch), dim3(256), c->asm_diag_lds, c->stream, c->M,
                       c->P, c->q, c->slab, c->slab_stride, c->g, c->HA, c->HB, c->HC, c->asm_stop,
                       c->fd_full ? c->v : nullptr, c->fd_full ? c->nplus : nullptr, c->pstride, gate, c->alt_r);
  } else {
    if (gate) { g_err = "gated assembly needs diagonal cost weights"; return -1; }
    hipLaunchKernelGGL(assemble_kernel, dim3(c->N + 1, c->batch), dim3(256), c->asm_lds, c->stream, c->M, c->P, c->q,
                       c->slab, c->slab_stride, c->g, c->HA, c->HB, c->HC, c->pstride);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int idto_hip_grad_hess(idto_hip_ctx* c) {
  HIP_OK(hipSetDevice(c->device));
  DropPrefetch(c, {IDTO_ARR_GRADIENT, IDTO_ARR_H_A, IDTO_ARR_H_B, IDTO_ARR_H_C, IDTO_ARR_HBANDS});
  if (TimeBegin(c, 1)) return -2;
  if (int rc = LaunchAssemble(c, nullptr)) return rc;
  return TimeEnd(c);
}

// Rows the fast solver works on.  The assembled Gauss-Newton Hessian has C_0 = I, B_1 = A_2 = 0
// and g_0 = 0 (q_0 is not a decision variable, TO.cc:1093-1165): block row 0 is decoupled, so the
// factorisation starts at row 1 (one block row less on the serial chain of the top workgroup)
// and x_0 = rhs_0.  Bands written into the context behind the API's back get the full system.
static int SolverFirstRow(const idto_hip_ctx* c) { return (c->h_assembled && c->N >= 2) ? 1 : 0; }

// Geometry of one launch of the banded block LDL^T solver (one right-hand side in the kernel; more
// go through penta_apply_kernel).
struct LdlPlan {
  int r0, n, k, K, gj_waves, lds, m_split;
  int lds_full;   // the chain code's carve-up with every row of the system (nested dissection, fused launch: their own row counts)
  size_t qq0;
};
static int SolverBlockSize(int k, bool single_rhs_only = false) {
  // block sizes of the reference's example models are instantiated exactly, others are padded
  // (30: the factorisation alone - no penta_apply_kernel of that size -, for the KKT systems of kkt.h: allegro's 23 + 6.
  // The 32 x 32 instantiation needs three elimination wavefronts and spills 378 registers.)
  if (single_rhs_only && (k == 29 || k == 4)) return k;   // (exact instantiations for the KKT systems of allegro and spinner)
  if (single_rhs_only && k > 24 && k <= 30) return 30;
  return (k == 2 || k == 3 || k == 5 || k == 19 || k == 23) ? k : (k <= 8 ? 8 : k <= 16 ? 16 : k <= 24 ? 24 : 32);
}
static int PlanLdl(idto_hip_ctx* c, bool one_sided, LdlPlan* p) {
  p->r0 = SolverFirstRow(c);
  p->n = c->N + 1 - p->r0;
  p->k = c->nq;
  p->qq0 = (size_t)p->r0 * p->k * p->k;
  if (p->k > 32) { g_err = "fast solver supports nq <= 32"; return -1; }
  p->K = SolverBlockSize(p->k, c->ldl_npos > 0);
  const int per_wave = 64 - p->K, ncr = 2 * p->K + 1;
  p->gj_waves = (ncr + per_wave - 1) / per_wave;
  // two-sided elimination (two workgroups meeting at block rows m, m+1) once the horizon is long
  // enough to pay for the hand-over
  p->m_split = (c->two_sided && !one_sided && p->n >= 10) ? (p->n - 1) / 2 : 0;
  // (a workgroup of the two-sided elimination keeps the right-hand side and rt / x of its own rows only)
  p->lds = penta_ldl_layout(p->n, p->K, 1, ldl_two_sided_rows(p->n, p->m_split, 1)).end * (int)sizeof(double);
  p->lds_full = penta_ldl_layout(p->n, p->K, 1).end * (int)sizeof(double);
  if (p->lds > 160 * 1024) { g_err = "right-hand sides do not fit the LDS carve-up"; return -1; }
  // the two workgroups must not share a CU (each is one wavefront per SIMD, issue-bound): ask for
  // more than half of the 160 KB LDS so that the dispatcher cannot co-locate them
  if (p->m_split > 0) p->lds = std::max(p->lds, 84 * 1024);
  if (p->lds > 160 * 1024) { g_err = "LDS carve-up too large"; return -1; }
  return 0;
}

// Nested dissection (penta_nd.h): seven workgroups - two producer / joiner pairs, two spike
// workgroups, the separator.  Its factors are not what penta_apply_kernel walks, so it serves the
// single-right-hand-side solves only (the Gauss-Newton step).
// separator in the middle; in each half the joiner chain (next to the separator) gets the extra row
struct NdSplit { int s, j1, j2; };
// Pipelined chains (penta_pipe.h): a joiner's block row costs ~1.35x a producer's (its spike wavefronts share the
// SIMDs) and its two join rows come after the producer's hand-over, so the producers take ~57% of the rows that are
// not join rows: both sides then reach the join together (measured at K = 19: 2.56 / 3.45 us per row).
static NdSplit nd_split(int n, bool pipe, int K) {
  NdSplit sp;
  sp.s = (n - 2) / 2;
  const int htop = sp.s, hbot = n - sp.s - 2;
  auto producer_rows = [&](int half) {
    if (!pipe) {
      // (seven workgroups.  Measured: producer done at 0.6 + (np + 2) t, joiner at the join at 0.6 + d + (half - np - 2) t'
      // with t = 4.2, t' = 4.53, d = 6.3 us at K = 23 (the joiner publishes every row and starts later) and t = 6.2,
      // t' = 6.3, d = 8.5 at K = 29: both sides meet at np = (half - 2) / 2 + 0.27 resp. - 0.24 rows.  An odd
      // half - 2 (every even n) therefore rounds UP at K = 23 - allegro N = 60: 67.7 / 66.6 us instead of 63.6 / 70.3 -
      // and down at K = 29 - N = 40: 62 / 66 instead of 68 / 59.5.)
      static const int extra = [] { const char* e = std::getenv("IDTO_ND_PRODUCER_EXTRA"); return e ? std::atoi(e) : -1; }();   // (measurement aid)
      const int np = extra >= 0 ? (half - 2) / 2 + extra : (half - 2 + (K > 20 && K <= 24 ? 1 : 0)) / 2;
      return std::max(1, std::min(np, half - 3));
    }
    static const double share = [] { const char* e = std::getenv("IDTO_PIPE_SPLIT"); return e ? std::atof(e) : 0.52; }();   // (measurement aid)
    int np = (int)(share * (half - 2) + 0.6);
    return std::max(1, std::min(np, half - 3));
  };
  sp.j1 = producer_rows(htop);                     // producer P0: rows 0 .. j1-1
  sp.j2 = n - producer_rows(hbot) - 2;             // producer P3: rows j2+2 .. n-1
  return sp;
}
static int NdLds(const idto_hip_ctx* c, const LdlPlan& p, int nloc_max);
static bool NdEligible(const idto_hip_ctx* c, const LdlPlan& p) {
  const bool inst = (p.K == 2 || p.K == 3 || p.K == 5 || p.K == 19 || p.K == 23 || p.K == 29 || (p.K == 8 && c->ldl_npos > 0)) && p.K == p.k;
  // (seven workgroups per problem, one per CU: a batch that would not fit the 256 CUs at once is
  // better served by the two-workgroup form - same work per problem on fewer CUs)
  // (horizons from nd_min_rows block rows on - option "nd_min_rows", 16: the MPC examples plan over 20 steps, and the
  // one-launch iteration that serves shorter systems takes 99 us for the cheetah there against 21 + ~40 of fd_kernel and
  // the pipelined solver.  The seven-workgroup kernel keeps 24: its chains of 4 - 5 rows buy nothing below that.)
  const bool pipe_kernel = c->solver_pipe && p.K <= 20;
  if (!(c->solver_nd && c->two_sided && inst && p.n >= (pipe_kernel ? c->nd_min_rows : std::max(24, c->nd_min_rows)) && 7 * c->batch <= 256)) return false;
  // the joiner chains' per-row tables hold ND_MAXROWS local rows: longer horizons (n >= 127) take the
  // two-workgroup factorisation
  NdSplit sp = nd_split(p.n, c->solver_pipe && p.K <= 20, p.K);
  const int nloc_max = std::max(std::max(sp.s - sp.j1, sp.j2 - sp.s), std::max(sp.j1, p.n - sp.j2 - 2));
  return nloc_max <= ND_MAXROWS && NdLds(c, p, nloc_max) <= 160 * 1024;
}
// (the chains keep the right-hand side and rt / x of their own rows: joiner nloc, producer nloc + 2 pseudo-rows, + 2 spare)
static int NdChainRows(int nloc_max) { return nloc_max + 4; }
static int NdLds(const idto_hip_ctx* c, const LdlPlan& p, int nloc_max) {
  const int ks = ldl_ks(p.K), NF = 2 * p.K, KP = 4 * ((p.K + 3) / 4);
  const int chain = penta_ldl_layout(p.n, p.K, 1, NdChainRows(nloc_max)).end * (int)sizeof(double);
  const int spike = (3 * NF * ks + 3 * KP * ks + 3 * p.K * p.K + 2 * ks + 4) * (int)sizeof(double);
  const int sep = nd_sep_lds_doubles(p.K) * (int)sizeof(double);
  return std::max(std::max(spike, sep), chain);
}
// idto_hip_gn_step / idto_hip_tr_solve asked the solver's launch to assemble g and the bands itself (AsmInSolver): the
// 4 (N + 1) workgroups behind the solver's own run assemble_terms_kernel's rows (penta_pipe.h PipeAsm)
static PipeAsm TakeAsm(idto_hip_ctx* c, const LdlPlan& p) {
  PipeAsm F{};
  F.on = c->fuse_asm_next ? 1 : 0;
  c->fuse_asm_next = false;
  if (F.on) {
    F.nq = c->nq; F.nv = c->nv; F.rows = c->N + 1; F.first = p.r0;
    F.P = c->P; F.q = c->q; F.terms = c->terms; F.v_res = c->v; F.nplus = c->nplus;
    F.g = c->g; F.HA = c->HA; F.HB = c->HB; F.HC = c->HC; F.alt = c->alt_r; F.ready = c->asm_ready; F.gate = c->fuse_gate;
    c->last_assembly = 4;
  }
  return F;
}
// ... and (the pipelined chains' launch inside idto_hip_tr_solve) to decide on the trial point first
static void TakeDecide(idto_hip_ctx* c, PipeAsm* F) {
  if (!(F->on && c->fuse_decide)) return;
  F->decide = 1;
  F->dq = c->q_trial; F->dv = c->v; F->dslab = c->slab; F->dslab_stride = (int)c->slab_stride; F->dcost = c->cost;
  F->ddiag = c->weights_diagonal ? 1 : 0;
  F->dT = *c->fuse_decide; F->dalt = c->alt_w;
  F->dword = c->decide_word;
  F->g2 = c->g2; F->HA2 = c->HA2; F->HB2 = c->HB2; F->HC2 = c->HC2;
  F->curpre = c->decide_word + 2;
  if (++c->decide_epoch == 0u) c->decide_epoch = 1u;
  F->depoch = c->decide_epoch;
  c->fuse_decide = nullptr;
}

// The scalar band factorisation (penta_band.h): blocks up to 5 (half width 3 k - 1 <= 14: a lane per diagonal in a row
// of 16), one workgroup per problem, single right-hand side.
static bool BandEligible(const idto_hip_ctx* c, const LdlPlan& p, bool whole_step = false) {
  // (option solver_band: 0 off, 1 blocks up to 4 - at 5 the pipelined kernel is faster, 41 against 47 us for hopper -, 2 up to 5)
  if (!(c->solver_band > 0 && c->two_sided && p.K == p.k && p.k >= 2 && p.k <= (c->solver_band > 1 ? 5 : 4))) return false;
  const int M = p.n * p.k, W = 3 * p.k;
  // (a batch: two wavefronts per problem shorten ONE problem's solve; with many in flight the five workgroups' work per
  // problem is what counts - 64 spinner problems 554k against 567k it/s, 256: 747k / 782k; acrobot 649k / 619k)
  // (gn_small.h - `whole_step` - is one workgroup per problem for EVERYTHING: there the batch argument points the other
  // way, 64 acrobot problems 650k -> 2.89M it/s)
  if (c->batch > 1 && c->solver_band < 2 && p.k > 2 && !whole_step) return false;
  // (horizons the pipelined kernel would take: shorter ones keep the fused launch / the two-workgroup factorisation)
  return p.n >= c->nd_min_rows && M >= 4 * W && band_layout(M, W).end * (int)sizeof(double) <= 160 * 1024;
}
static int LaunchBand(idto_hip_ctx* c, const LdlPlan& p, const double* b, double sign, double* xo) {
  BandArgs A;
  A.n = p.n; A.k = p.k;
  A.HA = c->HA + p.qq0; A.HB = c->HB + p.qq0; A.HC = c->HC + p.qq0;
  A.b = b + (size_t)p.r0 * p.k; A.rhs_sign = sign; A.x = xo + (size_t)p.r0 * p.k; A.Dst = c->Dst;
  ++c->epoch;
  if (++c->fact_id == 0) c->fact_id = 1;
  A.status = c->status_dev; A.fact_id = c->fact_id; A.epoch = c->epoch; A.pstride = c->pstride;
  A.npos = c->ldl_npos;
  A.ts = c->solver_debug ? c->dbg : nullptr;
  c->last_solver = 6;
  const PipeAsm F = TakeAsm(c, p);
  const int W = 3 * p.k;
  int lds = band_layout(p.n * p.k, W).end * (int)sizeof(double);
  if (F.on) lds = std::max(lds, c->asm_terms_lds);
  const dim3 grid(1 + (F.on ? 4 * (c->N + 1) : 0), c->batch);
#define BAND_LAUNCH(WM) hipLaunchKernelGGL((penta_band_kernel<WM>), grid, dim3(256), lds, c->stream, A, F)
  switch (W) {
    case 6: BAND_LAUNCH(6); break;
    case 9: BAND_LAUNCH(9); break;
    case 12: BAND_LAUNCH(12); break;
    default: BAND_LAUNCH(15); break;
  }
#undef BAND_LAUNCH
  HIP_OK(hipGetLastError());
  return 0;
}
static int LaunchNd(idto_hip_ctx* c, const LdlPlan& p, const double* b, double sign, double* xo) {
  NdArgs A;
  A.debug_skip_role = c->debug_skip_role;
  A.debug_pipe_tail = c->debug_pipe_tail;
  A.asm_ready = nullptr; A.asm_first = 0; A.wt_rows = 0;
  A.rec_tail = 0; A.lds_doubles = 0; A.wst = nullptr;
  A.spin = SpinCtl{nullptr, 0};   // (set by the kernel: behind the per-problem status words)
  A.n = p.n; A.k = p.k;
  A.HA = c->HA + p.qq0; A.HB = c->HB + p.qq0; A.HC = c->HC + p.qq0;
  A.b = b + (size_t)p.r0 * p.k; A.rhs_sign = sign; A.x = xo + (size_t)p.r0 * p.k;
  A.Ust = c->Ust; A.Hst = c->Hst; A.Est = c->Est; A.Dst = c->Dst;
  { const NdSplit sp = nd_split(p.n, c->solver_pipe && p.K <= 20, p.K); A.s = sp.s; A.j1 = sp.j1; A.j2 = sp.j2; }
  const int nloc_max = std::max(std::max(A.s - A.j1, A.j2 - A.s), std::max(A.j1, A.n - A.j2 - 2));
  if (nloc_max > ND_MAXROWS) { g_err = "horizon too long for the nested-dissection solver's row tables"; return -1; }
  const int lds = NdLds(c, p, nloc_max);
  if (lds > 160 * 1024) { g_err = "nested-dissection solver: LDS carve-up too large"; return -1; }
  A.npos = c->ldl_npos; A.lds_rows = NdChainRows(nloc_max);
  A.xch = c->xch; A.xch_pair = (int)c->xch_count; A.flags = c->flags;
  if (c->solver_pipe && p.K <= 20) {
    // pipelined chains (penta_pipe.h): five workgroups of eight wavefronts, the joiners carry their spike columns
    int plds = 0;
    switch (p.K) {
      case 2: plds = pipe_layout<2>(p.n, true).end; break;
      case 3: plds = pipe_layout<3>(p.n, true).end; break;
      case 5: plds = pipe_layout<5>(p.n, true).end; break;
      default: plds = pipe_layout<19>(p.n, true).end; break;
    }
    const int sep = nd_sep_lds_doubles(p.K) * (int)sizeof(double);
    plds = std::max(plds * (int)sizeof(double), sep);
    if (plds > 160 * 1024) { g_err = "pipelined solver: LDS carve-up too large"; return -1; }
    A.rowcnt = c->pipe_rowcnt; A.ndbuf = c->nd_buf;
    ++c->pipe_launches;
    c->last_solver = 4;
    A.rowunit = 1ull; A.rowtarget = c->pipe_launches;
    ++c->epoch;
    if (++c->fact_id == 0) c->fact_id = 1;
    A.epoch = c->epoch; A.status = c->status_dev; A.fact_id = c->fact_id; A.pstride = c->pstride;
    A.ts = c->solver_debug ? c->dbg : nullptr;
    // (idto_hip_gn_step: g and the bands are assembled by 4 (N + 1) more workgroups of this launch, penta_pipe.h PipeAsm)
    PipeAsm F = TakeAsm(c, p);
    if (p.K == 5 || p.K == 19) TakeDecide(c, &F);
    if (F.on) plds = std::max(plds, c->asm_terms_lds);
    if (F.decide) plds = std::max(plds, c->cost_lds);
    A.asm_ready = nullptr; A.asm_first = 0;
    const dim3 pgrid(5 + (F.on ? 4 * (c->N + 1) : 0) + (F.decide ? 1 : 0), c->batch);
#define PIPE_LAUNCH(KM) hipLaunchKernelGGL((penta_pipe_kernel<KM>), pgrid, dim3(512), plds, c->stream, A, F)
#define PIPE_LAUNCH_DEC(KM) hipLaunchKernelGGL((penta_pipe_kernel<KM, true>), pgrid, dim3(512), plds, c->stream, A, F)
    if (F.decide) {
      if (p.K == 5) PIPE_LAUNCH_DEC(5); else PIPE_LAUNCH_DEC(19);
    } else
    switch (p.K) {
      case 2: PIPE_LAUNCH(2); break;
      case 3: PIPE_LAUNCH(3); break;
      case 5: PIPE_LAUNCH(5); break;
      default: PIPE_LAUNCH(19); break;
    }
#undef PIPE_LAUNCH
#undef PIPE_LAUNCH_DEC
    HIP_OK(hipGetLastError());
    return 0;
  }
  A.rowcnt = c->nd_rowcnt; A.ndbuf = c->nd_buf;
  { static const int wt = [] { const char* e = std::getenv("IDTO_ND_WT"); return e ? std::atoi(e) : 0; }(); A.wt_rows = wt; }   // (measurement aid)
  ++c->nd_launches;
  c->last_solver = 2;
  {  // I/O wavefronts of a chain workgroup (penta_ldl_body: NW = 4, one elimination wavefront, three helpers)
    const int nhelp = 4 - p.gj_waves, io_first = (nhelp > 2) ? p.gj_waves + 1 : p.gj_waves;
    A.rowunit = (unsigned long long)(4 - io_first);
    A.rowtarget = c->nd_launches * A.rowunit;
  }
  ++c->epoch;
  if (++c->fact_id == 0) c->fact_id = 1;
  A.epoch = c->epoch; A.status = c->status_dev; A.fact_id = c->fact_id; A.pstride = c->pstride;
  A.ts = c->solver_debug ? c->dbg : nullptr;
  // back substitution in recursion form (penta_pipe.h chain_recursion_tail) where every row's [Y | Z | c] fits the
  // 160 KB: allegro's 23 x 23 blocks up to N = 60, its 29 x 29 KKT blocks up to N = 40
  int nd_lds = lds;
  if (c->nd_recursion && c->nd_wst && p.K == p.k && (p.K == 23 || p.K == 29)) {
    const int nj = std::max(A.s - A.j1, A.j2 - A.s), np = std::max(A.j1, A.n - A.j2 - 2), all = 160 * 1024 / (int)sizeof(double);
    if (const int ww = p.K == 23 ? pipe_recursion_tail_fits<23>(all, nj, np) : pipe_recursion_tail_fits<29>(all, nj, np)) {
      A.rec_tail = ww; A.lds_doubles = all; A.wst = c->nd_wst;
      nd_lds = 160 * 1024;
    }
  }
  const dim3 grid(7, c->batch);
#define ND_LAUNCH(KM, PD) hipLaunchKernelGGL((penta_nd_kernel<KM, PD>), grid, dim3(256), nd_lds, c->stream, A)
  switch (p.K) {
    case 2: ND_LAUNCH(2, false); break;
    case 3: ND_LAUNCH(3, false); break;
    case 5: ND_LAUNCH(5, false); break;
    case 23: ND_LAUNCH(23, false); break;
    case 29: ND_LAUNCH(29, false); break;
    case 8: ND_LAUNCH(8, false); break;
    default: ND_LAUNCH(19, false); break;
  }
#undef ND_LAUNCH
  HIP_OK(hipGetLastError());
  return 0;
}

static int LaunchLdl(idto_hip_ctx* c, const double* b, double sign, double* xo, bool one_sided = false, bool allow_nd = false,
                     bool factor_only = false) {
  LdlPlan p;
  if (int rc = PlanLdl(c, one_sided, &p)) return rc;
  if (allow_nd && !one_sided && BandEligible(c, p)) return LaunchBand(c, p, b, sign, xo);
  if (allow_nd && !one_sided && NdEligible(c, p)) return LaunchNd(c, p, b, sign, xo);
  const int n = p.n, k = p.k, m_split = p.m_split, lds = p.lds, nrhs = 1;
  const size_t qq0 = p.qq0;
  b += (size_t)p.r0 * k;
  xo += (size_t)p.r0 * k;
  double* dbg = c->solver_debug ? c->dbg : nullptr;
  const dim3 grid(m_split > 0 ? 2 : 1, c->batch);
  c->last_solver = 1;
  if (m_split > 0) ++c->epoch;  // (exchange buffer and flags live in the problem's arena)
  if (++c->fact_id == 0) c->fact_id = 1;  // (0 is the initial value of the status word)
#define LDL_ARGS n, k, c->HA + qq0, c->HB + qq0, c->HC + qq0, b, sign, nrhs, xo, c->Ust, c->Hst, c->Est, c->Dst, dbg, \
                 m_split, c->xch, c->flags, c->epoch, c->status_dev, c->fact_id, c->pstride, factor_only ? 1 : 0, c->ldl_npos
#define LDL_LAUNCH(KM, PD, GW) \
  hipLaunchKernelGGL((penta_ldl_kernel<KM, 256, PD, GW>), grid, dim3(256), lds, c->stream, LDL_ARGS)
  switch (p.K) {
    case 2: LDL_LAUNCH(2, false, 1); break;
    case 3: LDL_LAUNCH(3, false, 1); break;
    case 5: LDL_LAUNCH(5, false, 1); break;
    case 8: LDL_LAUNCH(8, true, 1); break;
    case 16: LDL_LAUNCH(16, true, 1); break;
    case 19: LDL_LAUNCH(19, false, 1); break;
    case 23: LDL_LAUNCH(23, false, 2); break;
    case 24: LDL_LAUNCH(24, true, 2); break;
    case 30: LDL_LAUNCH(30, true, 2); break;
    case 29: LDL_LAUNCH(29, false, 2); break;
    case 4: LDL_LAUNCH(4, false, 1); break;
    default: LDL_LAUNCH(32, true, 3); break;
  }
#undef LDL_LAUNCH
#undef LDL_ARGS
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- one persistent launch for the whole Gauss-Newton iteration (fused.h)
static int FusedVariant(const idto_hip_ctx* c) {  // instantiated (MAXC, K) combinations: the reference's example models
  const int mc = c->maxc <= 2 ? 2 : (c->maxc <= 3 ? 3 : (c->maxc <= 4 ? 4 : 8));
  if (mc == 2 && c->nq == 2) return 1;
  if (mc == 3 && c->nq == 3) return 2;
  if (mc == 3 && c->nq == 5) return 3;
  if (mc == 3 && c->nq == 19) return 4;
  if (mc == 4 && c->nq == 23) return 5;
  return 0;
}
static bool FusedEligible(const idto_hip_ctx* c) {
  {  // the nested-dissection solver is its own launch (seven workgroups): the three-launch path takes it
    LdlPlan p;
    idto_hip_ctx* cc = const_cast<idto_hip_ctx*>(c);
    if (PlanLdl(cc, false, &p) == 0 && (NdEligible(c, p) || BandEligible(c, p))) return false;   // (likewise the scalar band factorisation's)
  }
  return c->fused && c->batch == 1 && c->weights_diagonal && !c->reference_solver && !c->solver_debug && c->fd_stop == 0 &&
         c->asm_stop == 0 && c->k_begin == 0 && c->k_end == c->N && c->N >= 2 && FusedVariant(c) != 0;
}
static int LaunchFused(idto_hip_ctx* c) {
  DropPrefetch(c, {IDTO_ARR_V, IDTO_ARR_A, IDTO_ARR_NPLUS, IDTO_ARR_SLAB, IDTO_ARR_GRADIENT, IDTO_ARR_H_A, IDTO_ARR_H_B,
                   IDTO_ARR_H_C, IDTO_ARR_HBANDS, IDTO_ARR_STEP});
  c->con_ready = false; c->con_begun = false;
  if (!c->h_assembled) {  // x_0 = -g_0 = 0 is not written by the solver (SolverFirstRow)
    HIP_OK(hipMemsetAsync(c->step, 0, (size_t)c->nq * sizeof(double), c->stream));
    c->h_assembled = true;
  }
  LdlPlan p;
  if (int rc = PlanLdl(c, false, &p)) return rc;
  const int mode = 1 + c->gradients_method;
  const int E = FdEvals(c, mode), groups = 256 / c->npaths;
  int ec = E;
  while (ec > groups && FdLds(c, mode, ec, false, false) > 160 * 1024) ec = ((ec - 1) / groups) * groups;
  const int fd_lds = FdLds(c, mode, ec, false, false);
  if (fd_lds > 160 * 1024) { g_err = "finite-difference evaluation set does not fit in LDS"; return -1; }
  const int lds = std::max(std::max(fd_lds, c->asm_diag_lds), std::max(p.lds_full, 84 * 1024));
  FusedArgs A;
  A.M = c->M; A.cp = c->cp; A.P = c->P;
  A.q = c->q; A.slab = c->slab; A.slab_stride = c->slab_stride; A.v = c->v; A.a = c->a; A.nplus = c->nplus;
  A.fd_mode = mode; A.fd_echunk = ec; A.nfd = c->N;
  A.g = c->g; A.HA = c->HA; A.HB = c->HB; A.HC = c->HC; A.nrows = c->N + 1;
  A.n = p.n; A.k = p.k;
  A.sHA = c->HA + p.qq0; A.sHB = c->HB + p.qq0; A.sHC = c->HC + p.qq0;
  A.b = c->g + (size_t)p.r0 * p.k; A.rhs_sign = -1.0; A.x = c->step + (size_t)p.r0 * p.k;
  A.Ust = c->Ust; A.Hst = c->Hst; A.Est = c->Est; A.Dst = c->Dst;
  if (p.m_split > 0) ++c->epoch;
  if (++c->fact_id == 0) c->fact_id = 1;
  A.m_split = p.m_split; A.xch = c->xch; A.flags = c->flags; A.epoch = c->epoch; A.status = c->status_dev;
  A.fact_id = c->fact_id;
  ++c->sync_steps;
  A.sync = c->sync_cnt;
  A.dbg = c->fused_debug ? c->dbg : nullptr;
  A.fd_target = c->sync_steps * (unsigned long long)A.nfd;
  A.asm_target = c->sync_steps * (unsigned long long)(4 * A.nrows);
  const dim3 grid(A.nfd + 4 * A.nrows + (p.m_split > 0 ? 2 : 1));
  c->last_solver = 5;
  c->last_step_kind = 2;
  if (TimeBegin(c, 3)) return -2;
#define FUSED_LAUNCH(MC, KM, PD, GW) \
  hipLaunchKernelGGL((gn_fused_kernel<MC, KM, PD, GW>), grid, dim3(256), lds, c->stream, A)
  switch (FusedVariant(c)) {
    case 1: FUSED_LAUNCH(2, 2, false, 1); break;
    case 2: FUSED_LAUNCH(3, 3, false, 1); break;
    case 3: FUSED_LAUNCH(3, 5, false, 1); break;
    case 4: FUSED_LAUNCH(3, 19, false, 1); break;
    default: FUSED_LAUNCH(4, 23, false, 2); break;
  }
#undef FUSED_LAUNCH
  HIP_OK(hipGetLastError());
  c->fd_full = true; c->partials_ahead = false;
  c->terms_valid = false;   // (the fused kernel assembles from the slab itself)
  return TimeEnd(c);
}

// gn_small.h: fd + assembly + band solve of a small all-revolute model in ONE workgroup per problem.  What it stands in
// for must be what the two launches would have run: forward differences from the straight-line evaluation, diagonal
// weights, the whole horizon, the scalar band factorisation (BandEligible), nothing switched to a measurement mode.
// (p: the plan of the system the launch solves - H's, or the KKT context's with blocks of nq + nu)
static int SmallLds(const idto_hip_ctx* c, const LdlPlan& p, int* lds_small) {
  int band = band_layout(p.n * p.k, 3 * p.k).end;
  band += band & 1;
  if (lds_small) *lds_small = band;
  return (band + gn_small_doubles(c->N, c->nq, c->M.fast_n, p.k)) * (int)sizeof(double);
}
static bool SmallEligible(const idto_hip_ctx* c) {
  if (!c->gn_small || !c->fd_fast || c->gradients_method != 0 || !c->weights_diagonal || c->reference_solver) return false;
  if (!(c->M.fast_shape == 1 || c->M.fast_shape == 5) || c->M.nfloat != 0 || c->nq != c->nv || c->npaths != 1) return false;
  if (!((c->M.fast_shape == 1 && c->nq == 2) || (c->M.fast_shape == 5 && c->nq == 3))) return false;   // (the instantiations)
  if (c->k_begin != 0 || c->k_end != c->N || c->fd_stop || c->asm_stop || c->solver_debug || c->ldl_npos > 0) return false;
  LdlPlan p;
  idto_hip_ctx* cc = const_cast<idto_hip_ctx*>(c);
  const bool assembled = c->h_assembled;
  cc->h_assembled = true;   // (the kernel assembles H itself: block row 0 is the identity, the solver starts at row 1)
  const bool ok = c->N >= 2 && PlanLdl(cc, false, &p) == 0 && BandEligible(c, p, true) && p.r0 == 1 && SmallLds(c, p, nullptr) <= 160 * 1024;
  cc->h_assembled = assembled;
  return ok;
}
// tr: inside idto_hip_tr_solve - the launch evaluates the trial point c->q_trial into the output set `alt`, decides, and
// goes on to g, H and the step only for an accepted step (gn_small.h SmallArgs::T); last: tau, cost and decision only
// kkt: with enforced constraints - the step is the banded KKT solve, z into c->kkt's step (gn_small.h SmallArgs::kkt_r0)
// iter: ... and tr_iter_kernel's part for the iterate in front (SmallArgs::I; a second workgroup reads the status words)
static int LaunchSmall(idto_hip_ctx* c, const TrDecideArgs* tr = nullptr, bool last = false, AltSel alt = AltSel{nullptr, 0, 0},
                       bool kkt = false, const TrIterArgs* iter = nullptr) {
  DropPrefetch(c, {IDTO_ARR_V, IDTO_ARR_A, IDTO_ARR_NPLUS, IDTO_ARR_SLAB, IDTO_ARR_GRADIENT, IDTO_ARR_H_A, IDTO_ARR_H_B,
                   IDTO_ARR_H_C, IDTO_ARR_HBANDS, IDTO_ARR_STEP, IDTO_ARR_COST});
  c->con_ready = false; c->con_begun = false;
  if (!c->h_assembled) {  // x_0 = -g_0 = 0 is not written by the solver (SolverFirstRow)
    HIP_OK(hipMemset2DAsync(c->step, c->pstride, 0, (size_t)c->nq * sizeof(double), (size_t)c->batch, c->stream));
    c->h_assembled = true;
  }
  LdlPlan p;
  idto_hip_ctx* sc = kkt ? c->kkt : c;   // whose system the launch solves
  if (int rc = PlanLdl(sc, false, &p)) return rc;
  SmallArgs A;
  A.kkt_r0 = p.r0; A.kstride = sc->pstride;
  A.M = c->M; A.cp = c->cp; A.P = c->P; A.q = c->q; A.slab = c->slab; A.slab_stride = c->slab_stride;
  A.v = c->v; A.a = c->a; A.nplus = c->nplus; A.g = c->g; A.HA = c->HA; A.HB = c->HB; A.HC = c->HC;
  A.pstride = c->pstride; A.alt = alt;
  A.T = tr ? *tr : TrDecideArgs{}; A.tau_only = last ? 1 : 0; A.cost_out = c->cost;
  if (iter) A.I = *iter; else { A.I = TrIterArgs{}; A.I.state = nullptr; }
  const unsigned gx = iter ? 2u : 1u;
  if (tr) A.q = c->q_trial;
  BandArgs& B = A.B;
  B.n = p.n; B.k = p.k;
  B.HA = sc->HA + p.qq0; B.HB = sc->HB + p.qq0; B.HC = sc->HC + p.qq0;
  B.b = sc->g + (size_t)p.r0 * p.k; B.rhs_sign = -1.0; B.x = sc->step + (size_t)p.r0 * p.k; B.Dst = sc->Dst;
  ++sc->epoch;
  if (++sc->fact_id == 0) sc->fact_id = 1;
  B.status = sc->status_dev; B.fact_id = sc->fact_id; B.epoch = sc->epoch; B.pstride = sc->pstride;
  B.npos = sc->ldl_npos; B.ts = nullptr;
  A.ts = std::getenv("IDTO_SMALL_STAMPS") ? c->dbg : nullptr;
  const int lds = SmallLds(c, p, &A.lds_small);
  if (sc->small_stage_N != c->N) {   // (once per horizon; the KKT system's table lives in its own context)
    const int qq = p.k * p.k, nb = c->N + 1;
    std::vector<BandStageItem> tab((size_t)band_stage_table(p.n, p.k, 0, 0, 0, 0, nullptr));
    band_stage_table(p.n, p.k, p.r0 * qq, nb * qq + p.r0 * qq, 2 * nb * qq + p.r0 * qq, 3 * nb * qq + p.r0 * p.k, tab.data());
    Release(sc, &sc->small_stage);
    if (Alloc(sc, tab.size(), &sc->small_stage)) return -2;
    HIP_OK(hipMemcpy(sc->small_stage, tab.data(), tab.size() * sizeof(BandStageItem), hipMemcpyHostToDevice));
    sc->small_stage_n = (int)tab.size(); sc->small_stage_N = c->N;
  }
  A.stage = sc->small_stage; A.nstage = sc->small_stage_n;
  c->last_solver = 7;
  c->last_step_kind = 2;
  c->last_assembly = 5;
  if (!tr && TimeBegin(c, 3)) return -2;
  // (256 threads, one wavefront per SIMD: the evaluation needs more than the 256 registers a lane has at two per SIMD -
  // 512 threads spilled 19 / 67 registers to scratch inside it and the step was slower than the two launches)
  if (kkt && c->nq == 2) hipLaunchKernelGGL((gn_small_kernel<1, 6, 256, 9, true>), dim3(gx, c->batch), dim3(256), lds, c->stream, A);
  else if (kkt) hipLaunchKernelGGL((gn_small_kernel<5, 9, 256, 12, true>), dim3(gx, c->batch), dim3(256), lds, c->stream, A);
  else if (tr && c->nq == 2) hipLaunchKernelGGL((gn_small_kernel<1, 6, 256, 6, true>), dim3(gx, c->batch), dim3(256), lds, c->stream, A);
  else if (tr) hipLaunchKernelGGL((gn_small_kernel<5, 9, 256, 9, true>), dim3(gx, c->batch), dim3(256), lds, c->stream, A);
  else if (c->nq == 2) hipLaunchKernelGGL((gn_small_kernel<1, 6, 256>), dim3(gx, c->batch), dim3(256), lds, c->stream, A);
  else hipLaunchKernelGGL((gn_small_kernel<5, 9, 256>), dim3(gx, c->batch), dim3(256), lds, c->stream, A);
  HIP_OK(hipGetLastError());
  c->fd_full = !last; c->partials_ahead = false;
  c->terms_valid = false;   // (no single-record products: a later idto_hip_grad_hess assembles from the slab)
  return tr ? 0 : TimeEnd(c);
}

static int FactorSolve(idto_hip_ctx* c, const double* rhs, int nrhs, double* x, const RhsSource* src);
int idto_hip_factor_solve(idto_hip_ctx* c, const double* rhs, int nrhs, double* x) {
  return FactorSolve(c, rhs, nrhs, x, nullptr);
}
// src: the right-hand sides are [g | J^T] read in place (penta_apply.h RhsSource; nrhs > 1), `rhs` is then any valid
// pointer to (N + 1) nq doubles for the factorisation kernel's unused first column
static int FactorSolve(idto_hip_ctx* c, const double* rhs, int nrhs, double* x, const RhsSource* src) {
  const bool x0_written = src != nullptr;   // (the substitution kernel zeroes x_0 itself)
  HIP_OK(hipSetDevice(c->device));
  if (!rhs) DropPrefetch(c, {IDTO_ARR_STEP});
  const int n = c->N + 1, k = c->nq;
  const double* b = rhs ? rhs : c->g;
  double* xo = rhs ? x : c->step;
  if (!rhs) nrhs = 1;
  if (nrhs < 1) { g_err = "nrhs < 1"; return -1; }
  if (rhs && c->batch != 1) { g_err = "explicit right-hand sides serve single-problem contexts"; return -1; }
  if (TimeBegin(c, 2)) return -2;
  c->last_step_kind = rhs ? 0 : 1;
  if (c->reference_solver) {
    c->last_solver = 3;
    if (++c->fact_id == 0) c->fact_id = 1;
    hipLaunchKernelGGL(penta_kernel, dim3(1, c->batch), dim3(256), c->penta_lds, c->stream, n, k, c->HA, c->HB, c->HC, b,
                       rhs ? 1.0 : -1.0, xo, c->Kst, c->LUst, c->pivst, c->Yst, c->Zst, c->status_dev, c->fact_id,
                       c->pstride);
    HIP_OK(hipGetLastError());
    if (TimeEnd(c)) return -2;
    if (nrhs > 1) {
      hipLaunchKernelGGL(penta_solve_kernel, dim3(nrhs - 1), dim3(64), c->solve_lds, c->stream, n, k, c->HA, c->Kst,
                         c->LUst, c->pivst, c->Yst, c->Zst, rhs + (size_t)n * k, x + (size_t)n * k);
      HIP_OK(hipGetLastError());
    }
    return 0;
  }
  // block LDL^T: factorise once with the first right-hand side (two-sided when the horizon is
  // long enough; the substitution kernel walks both chains of factors) ...
  const int K = (k == 2 || k == 3 || k == 5 || k == 19 || k == 23) ? k : (k <= 8 ? 8 : k <= 16 ? 16 : k <= 24 ? 24 : 32);
  // (several right-hand sides: the factorisation stops after its forward pass, every column incl. the first is
  // substituted by penta_apply_kernel - the chains' own back substitution would only delay the others)
  int rc = LaunchLdl(c, b, rhs ? 1.0 : -1.0, xo, false, /*allow_nd=*/nrhs == 1, /*factor_only=*/nrhs > 1);
  const int r0 = SolverFirstRow(c), ns = n - r0;                      // the sub-system LaunchLdl factorised
  const int m_split = (c->two_sided && ns >= 10) ? (ns - 1) / 2 : 0;  // as LaunchLdl chose
  if (r0 && rhs && !x0_written)  // x_0 = rhs_0 for every column (row 0 of H is the identity); the default rhs has g_0 = 0 = x_0
    HIP_OK(hipMemcpy2DAsync(x, (size_t)n * k * sizeof(double), rhs, (size_t)n * k * sizeof(double), (size_t)k * sizeof(double),
                            (size_t)nrhs, hipMemcpyDeviceToDevice, c->stream));
  if (rc) return rc;
  if (nrhs > 1) {
    // ... then substitute the other right-hand sides in parallel: one wavefront each (two, one per chain, when the
    // factorisation was two-sided)
    const int waves = 4, cols = m_split > 0 ? waves / 2 : waves, blocks = (nrhs + cols - 1) / cols;
    const int lds = waves * (ns * K + 4 * 64 + 2) * (int)sizeof(double);   // (per column: rt of every row, the chains' exchange)
    const double* b1 = b + (size_t)r0 * k;
    double* x1 = xo + (size_t)r0 * k;
    if (!c->Tst && Alloc(c, (size_t)3 * (c->N + 1) * 32 * 36, &c->Tst)) return -2;
    RhsSource RS{};
    if (src) { RS = *src; RS.r0 = r0; }
#define APPLY_LAUNCH(KM)                                                                                          \
    if (KM <= 8) {   /* small blocks: the forward pass reads the row-major factors directly, no transposed copies */ \
      hipLaunchKernelGGL((penta_apply_kernel<KM, true>), dim3(blocks), dim3(64 * waves), lds, c->stream, ns, k, c->Ust, \
                         c->Hst, c->Est, c->Dst, c->Tst, b1, rhs ? 1.0 : -1.0, nrhs, x1, m_split, (size_t)n * k, RS);  \
    } else {                                                                                                        \
      hipLaunchKernelGGL(penta_factor_transpose_kernel<KM>, dim3(ns, 3), dim3(256), 0, c->stream, c->Ust, c->Hst,  \
                         c->Est, c->Tst);                                                                          \
      hipLaunchKernelGGL((penta_apply_kernel<KM, false>), dim3(blocks), dim3(64 * waves), lds, c->stream, ns, k, c->Ust, \
                         c->Hst, c->Est, c->Dst, c->Tst, b1, rhs ? 1.0 : -1.0, nrhs, x1, m_split, (size_t)n * k, RS); \
    }
    switch (K) {
      case 2: APPLY_LAUNCH(2); break;
      case 3: APPLY_LAUNCH(3); break;
      case 5: APPLY_LAUNCH(5); break;
      case 8: APPLY_LAUNCH(8); break;
      case 16: APPLY_LAUNCH(16); break;
      case 19: APPLY_LAUNCH(19); break;
      case 23: APPLY_LAUNCH(23); break;
      case 24: APPLY_LAUNCH(24); break;
      default: APPLY_LAUNCH(32); break;
    }
#undef APPLY_LAUNCH
    HIP_OK(hipGetLastError());
  }
  return TimeEnd(c);
}

int idto_hip_solve_host(idto_hip_ctx* c, const double* rhs_host, int nrhs, double* x_host) {
  HIP_OK(hipSetDevice(c->device));
  if (!rhs_host || !x_host || nrhs < 1) { g_err = "solve_host: bad arguments"; return -1; }
  if (c->batch != 1) { g_err = "solve_host serves single-problem contexts"; return -1; }
  const size_t count = (size_t)nrhs * (c->N + 1) * c->nq;
  if (EnsureStage(c, count)) return -2;
  c->con_ready = false; c->con_begun = false;  // the staging buffers are shared with the constraint step
  HIP_OK(hipMemcpyAsync(c->stage_rhs, rhs_host, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  int rc = idto_hip_factor_solve(c, c->stage_rhs, nrhs, c->stage_x);
  if (rc) return rc;
  HIP_OK(hipMemcpyAsync(x_host, c->stage_x, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));
  return FactorStatus(c);
}

static void LaunchDenseLdl(idto_hip_ctx* c, double* S, double* L, double* d, double* stat, double* rv, int neq, const double* b,
                           double b_sign, const double* b2);
static std::atomic<long> g_dense_solves{0};
long idto_hip_dense_solve_count() { return g_dense_solves.load(); }   // (tests: which branch of SolveLinearSystemInPlace ran)
#ifdef IDTO_TR_STAMPS
// (measurement build, tools/tr_stamps.py) the trust-region kernels' phase stamps of the last iteration, 100 MHz ticks
extern "C" int idto_hip_debug_tr_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(idto_dev::g_tr_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

// SolveLinearSystemInPlace's kDenseLdlt branch (reference optimizer/trajectory_optimizer.cc:2088-2093): H.MakeDense(),
// LDL^T, solve.  The debugging / cross-checking solver of the reference (its default is the block Thomas algorithm): the
// resident bands are spread into an n x n matrix (dense_from_bands_kernel), factorised by the blocked LDL^T of
// dense_ldl.h - one launch per panel of 32 columns, the right-hand side's forward substitution carried along - and
// substituted backwards by one workgroup.  Without pivoting (Eigen's ldlt() pivots on the diagonal): H is positive
// definite whenever the block Thomas branch succeeds; a pivot that is not positive and finite is reported as
// IDTO_HIP_FACTORIZATION_FAILED, where the reference's DRAKE_DEMAND(info() == Success) aborts.
int idto_hip_solve_dense_ldlt(idto_hip_ctx* c, const double* rhs_host, double* x_host) {
  HIP_OK(hipSetDevice(c->device));
  if (!rhs_host || !x_host) { g_err = "solve_dense_ldlt: bad arguments"; return -1; }
  if (c->batch != 1) { g_err = "solve_dense_ldlt serves single-problem contexts"; return -1; }
  const int nblk = c->N + 1, bs = c->nq, n = nblk * bs;
  ++g_dense_solves;
  if (c->dn_n != n) {
    Release(c, &c->dn_S); Release(c, &c->dn_d); Release(c, &c->dn_stat); Release(c, &c->dn_rv);
    c->dn_n = 0;
    if (Alloc(c, 2 * (size_t)n * n, &c->dn_S) || Alloc(c, (size_t)n, &c->dn_d) || Alloc(c, (size_t)2, &c->dn_stat) ||
        Alloc(c, 3 * (size_t)n, &c->dn_rv))   // [r with the finished panels eliminated | y = L^-1 r | x]
      return -2;
    c->dn_n = n;
  }
  if (EnsureStage(c, (size_t)n)) return -2;
  c->con_ready = false; c->con_begun = false;  // the staging buffers are shared with the constraint step
  const double stat0[2] = {std::numeric_limits<double>::infinity(), 0.0};
  HIP_OK(hipMemcpyAsync(c->stage_rhs, rhs_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipMemcpyAsync(c->dn_stat, stat0, sizeof(stat0), hipMemcpyHostToDevice, c->stream));
  double *S = c->dn_S, *L = c->dn_S + (size_t)n * n, *x = c->dn_rv + 2 * (size_t)n;
  hipLaunchKernelGGL(dense_from_bands_kernel, dim3(nblk), dim3(256), 0, c->stream, c->HA, c->HB, c->HC, nblk, bs, S);
  LaunchDenseLdl(c, S, L, c->dn_d, c->dn_stat, c->dn_rv, n, c->stage_rhs, 1.0, nullptr);
  hipLaunchKernelGGL(dense_ldl_solve_kernel, dim3(1), dim3(512), (n + 512 + DENSE_NB * (DENSE_NB + 1)) * sizeof(double), c->stream,
                     L, n, c->dn_d, c->dn_rv + n, 1.0, (const double*)nullptr, x, 1);
  HIP_OK(hipGetLastError());
  double stat[2];
  HIP_OK(hipMemcpyAsync(x_host, x, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(hipMemcpyAsync(stat, c->dn_stat, sizeof(stat), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));
  if (!(stat[0] > 0.0) || !std::isfinite(stat[1])) {
    g_err = "solve_dense_ldlt: a pivot of the dense LDL^T is not positive and finite";
    return IDTO_HIP_FACTORIZATION_FAILED;
  }
  return 0;
}

// The constrained degrees of freedom on the device: ONE copy, shared by the banded KKT step (which needs nothing else;
// batch contexts included) and by the Schur-complement route.  (Round 4 kept a list per route and each invalidated the
// other's: a resident KKT solve followed by the host loop - the TRF_SINGULAR_S fallback, or SolveFromWarmStart followed
// by EvalLagrangeMultipliers - allocated con_dofs, con_S, con_L, ... again on every alternation, and Alloc never frees:
// an MPC server on that path leaked ~2 MB per re-plan at allegro's size and paid a device-wide synchronisation each time.)
static int DeviceDofs(idto_hip_ctx* c, const int* dofs, int nu, const char* who, bool* changed) {
  if (!dofs || nu < 1 || nu > c->nv) { g_err = std::string(who) + ": bad arguments"; return -1; }
  for (int j = 0; j < nu; ++j)
    if (dofs[j] < 0 || dofs[j] >= c->nv) { g_err = std::string(who) + ": dof index out of range"; return -1; }
  *changed = !(c->con_dofs && c->con_dofs_host.size() == (size_t)nu && std::equal(dofs, dofs + nu, c->con_dofs_host.begin()));
  if (!*changed) return 0;
  if (c->con_dofs_cap < nu) {
    Release(c, &c->con_dofs);
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, (size_t)nu * sizeof(int)));
    c->allocs.push_back(p);
    c->con_dofs = static_cast<int*>(p);
    c->con_dofs_cap = nu;
  }
  HIP_OK(hipStreamSynchronize(c->stream));   // (work that reads the previous set may still be enqueued)
  HIP_OK(hipMemcpy(c->con_dofs, dofs, (size_t)nu * sizeof(int), hipMemcpyHostToDevice));
  c->con_dofs_host.assign(dofs, dofs + nu);
  c->con_nu = nu; c->con_neq = nu * c->N;
  c->con_schur_valid = false;   // (the Schur route's buffers, if any, were sized for another set)
  c->con_begun = false;
  return 0;
}
static int ConstraintDofs(idto_hip_ctx* c, const int* dofs, int nu) {
  bool changed = false;
  return DeviceDofs(c, dofs, nu, "equality constraints", &changed);
}
// the device arrays of the Schur-complement route for this set of degrees of freedom (made again - the superseded ones
// freed - only when the set really changes)
static int ConstraintBuffers(idto_hip_ctx* c, const int* dofs, int nu) {
  if (c->batch != 1) { g_err = "the equality-constraint step serves single-problem contexts"; return -1; }
  bool changed = false;
  if (int rc = DeviceDofs(c, dofs, nu, "constraint_schur", &changed)) return rc;
  const int N = c->N, n = (N + 1) * c->nq, neq = nu * N;
  if (!c->con_schur_valid) {
    Release(c, &c->con_S); Release(c, &c->con_d); Release(c, &c->con_h); Release(c, &c->con_L); Release(c, &c->con_rv);
    if (Alloc(c, (size_t)neq * neq + neq, &c->con_S) ||
        Alloc(c, (size_t)neq, &c->con_d) || Alloc(c, (size_t)neq + 2, &c->con_h) || Alloc(c, (size_t)neq * neq, &c->con_L) ||
        Alloc(c, 2 * (size_t)neq, &c->con_rv))   // [r with the finished panels eliminated | y = L^-1 r]
      return -2;
    const size_t need = (size_t)neq * neq + neq + 2 * (size_t)n + 2 * (size_t)neq + 4;
    if (c->con_pin_count < need) {
      if (c->con_pin) (void)hipHostFree(c->con_pin);
      c->con_pin = nullptr; c->con_pin_count = 0;
      HIP_OK(hipHostMalloc((void**)&c->con_pin, need * sizeof(double), hipHostMallocDefault));
      c->con_pin_count = need;
    }
    c->con_schur_valid = true;
    c->con_begun = false;
  }
  return 0;
}

int idto_hip_constraint_schur_begin(idto_hip_ctx* c, const int* dofs, int nu) {
  HIP_OK(hipSetDevice(c->device));
  if (!dofs || nu < 1 || nu > c->nv) { g_err = "constraint_schur: bad arguments"; return -1; }
  if (c->batch != 1) { g_err = "the equality-constraint step serves single-problem contexts"; return -1; }
  const int N = c->N, n = (N + 1) * c->nq, neq = nu * N;
  if (c->con_begun && c->con_nu == nu && std::equal(dofs, dofs + nu, c->con_dofs_host.begin()))
    return 0;  // already enqueued for the current Hessian
  if (int rc = ConstraintBuffers(c, dofs, nu)) return rc;
  c->con_ready = false; c->con_begun = false;
  if (EnsureStage(c, (size_t)(neq + 1) * n)) return -2;
  // Y = H^-1 [g | J^T]: the right-hand sides are read where they are (g; rows of the slab records), nothing is staged
  RhsSource src{};
  src.slab = c->slab; src.slab_stride = c->slab_stride; src.nu = nu; src.nv = c->nv; src.r0 = 0;
  src.dofs = c->con_dofs; src.g = c->g; src.alt = c->alt_r;
  int rc = FactorSolve(c, c->g, neq + 1, c->stage_x, &src);
  if (rc) return rc;
  hipLaunchKernelGGL(constraint_schur_kernel, dim3(neq), dim3(256), 3 * c->nq * sizeof(double), c->stream, c->slab,
                     c->slab_stride, c->con_dofs, nu, N, c->nq, c->nv, c->stage_x, neq, c->con_S,
                     c->con_S + (size_t)neq * neq, c->alt_r);
  HIP_OK(hipGetLastError());
  c->con_S_factored = false;
  c->con_begun = true;
  return 0;
}

int idto_hip_constraint_schur(idto_hip_ctx* c, const int* dofs, int nu, double* S_host, double* Jy_host) {
  HIP_OK(hipSetDevice(c->device));
  if (!S_host || !Jy_host || !dofs) { g_err = "constraint_schur: bad arguments"; return -1; }
  const bool same = c->con_begun && nu == c->con_nu && std::equal(dofs, dofs + nu, c->con_dofs_host.begin());
  if (!same) {
    int rc = idto_hip_constraint_schur_begin(c, dofs, nu);
    if (rc) return rc;
  }
  const int neq = c->con_neq;
  if (c->con_S_factored) {  // the device factorisation overwrote S: form it again from Y
    hipLaunchKernelGGL(constraint_schur_kernel, dim3(neq), dim3(256), 3 * c->nq * sizeof(double), c->stream, c->slab,
                       c->slab_stride, c->con_dofs, c->con_nu, c->N, c->nq, c->nv, c->stage_x, neq, c->con_S,
                       c->con_S + (size_t)neq * neq, c->alt_r);
    HIP_OK(hipGetLastError());
    c->con_S_factored = false;
  }
  HIP_OK(hipMemcpyAsync(c->con_pin, c->con_S, ((size_t)neq * neq + neq) * sizeof(double), hipMemcpyDeviceToHost,
                        c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));
  std::memcpy(S_host, c->con_pin, (size_t)neq * neq * sizeof(double));
  std::memcpy(Jy_host, c->con_pin + (size_t)neq * neq, (size_t)neq * sizeof(double));
  c->con_ready = true;
  return FactorStatus(c);
}

// S = L D L^T without pivoting, one launch per panel of 32 columns (dense_ldl.h dense_ldl_step_kernel); L -> con_L
// together with the forward substitution of the right-hand side r = b2 + b_sign b (-> con_rv = L^-1 r)
static void LaunchDenseLdl(idto_hip_ctx* c, double* S, double* L, double* d, double* stat, double* rv, int neq, const double* b,
                           double b_sign, const double* b2) {
  for (int j0 = 0; j0 < neq; j0 += DENSE_NB) {
    const int j1 = j0 + DENSE_NB, below = neq > j1 ? neq - j1 : 0;
    const int tiles = below > 0 ? (below + 31) / 32 : 1;
    hipLaunchKernelGGL(dense_ldl_step_kernel, dim3(tiles, tiles), dim3(256), 0, c->stream, S, L, neq, j0, d, stat,
                       b, b_sign, b2, rv, rv + neq);
  }
}
static void LaunchDenseLdl(idto_hip_ctx* c, double* S, int neq, const double* b, double b_sign, const double* b2) {
  LaunchDenseLdl(c, S, c->con_L, c->con_d, c->con_h, c->con_rv, neq, b, b_sign, b2);
}

int idto_hip_constraint_solve(idto_hip_ctx* c, const double* h_host, double* lambda_host, double* step_host,
                              double* jtl_host) {
  HIP_OK(hipSetDevice(c->device));
  if (!c->con_begun) { g_err = "constraint_solve: call idto_hip_constraint_schur_begin for the current Hessian first"; return -1; }
  if (!h_host || !lambda_host || !step_host || !jtl_host) { g_err = "constraint_solve: bad arguments"; return -1; }
  if (c->con_S_factored) { g_err = "constraint_solve: already called for this Hessian"; return -1; }
  const int N = c->N, n = (N + 1) * c->nq, neq = c->con_neq;
  // pinned layout behind [S | Jy]: [min, max | h] up, [min, max | lambda | step | J^T lambda] down
  double* pin = c->con_pin + (size_t)neq * neq + neq;
  pin[0] = std::numeric_limits<double>::infinity(); pin[1] = 0.0;
  std::memcpy(pin + 2, h_host, (size_t)neq * sizeof(double));
  HIP_OK(hipMemcpyAsync(c->con_h, pin, (size_t)(neq + 2) * sizeof(double), hipMemcpyHostToDevice, c->stream));
  double* S = c->con_S;
  LaunchDenseLdl(c, S, neq, S + (size_t)neq * neq, -1.0, c->con_h + 2);
  c->con_S_factored = true;
  c->con_lambda_at = c->con_lambda + 2;
  // lambda = S^-1 (h - J y_g): the forward substitution went with the factorisation
  hipLaunchKernelGGL(dense_ldl_solve_kernel, dim3(1), dim3(512), (neq + 512 + DENSE_NB * (DENSE_NB + 1)) * sizeof(double), c->stream, c->con_L, neq, c->con_d,
                     c->con_rv + neq, 1.0, (const double*)nullptr, c->con_lambda + 2, 1);
  hipLaunchKernelGGL(constraint_step_kernel, dim3((n + 63) / 64), dim3(64 * STEP_WAVES), (neq + 64 * STEP_WAVES) * sizeof(double), c->stream, c->slab,
                     c->slab_stride, c->con_dofs, c->con_nu, N, c->nq, c->nv, c->stage_x, neq, c->con_lambda + 2,
                     c->con_out, c->con_out + n, c->alt_r);
  HIP_OK(hipGetLastError());
  double* down = pin + neq + 2;
  HIP_OK(hipMemcpyAsync(down, c->con_h, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(hipMemcpyAsync(down + 2, c->con_lambda + 2, (size_t)neq * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(hipMemcpyAsync(down + 2 + neq, c->con_out, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));
  c->con_ready = true;  // Y is in place for idto_hip_constraint_step (fallback path)
  if (int fs = FactorStatus(c)) return fs;  // H itself was not positive definite
  const double dmin = down[0], dmax = down[1];
  if (!(dmin > 1e-13 * dmax) || !std::isfinite(dmax)) return 1;  // (semi-)singular S: the caller pivots on the host
  std::memcpy(lambda_host, down + 2, (size_t)neq * sizeof(double));
  std::memcpy(step_host, down + 2 + neq, (size_t)n * sizeof(double));
  std::memcpy(jtl_host, down + 2 + neq + n, (size_t)n * sizeof(double));
  return 0;
}

int idto_hip_constraint_step(idto_hip_ctx* c, const double* lambda_host, double* step_host, double* jtl_host) {
  HIP_OK(hipSetDevice(c->device));
  if (!c->con_ready) { g_err = "constraint_step: call idto_hip_constraint_schur for the current Hessian first"; return -1; }
  if (!lambda_host || !step_host || !jtl_host) { g_err = "constraint_step: bad arguments"; return -1; }
  const int N = c->N, n = (N + 1) * c->nq, neq = c->con_neq;
  double* pl = c->con_pin + (size_t)neq * neq + neq;  // [lambda | step | J^T lambda]
  std::memcpy(pl, lambda_host, (size_t)neq * sizeof(double));
  HIP_OK(hipMemcpyAsync(c->con_lambda, pl, (size_t)neq * sizeof(double), hipMemcpyHostToDevice, c->stream));
  c->con_lambda_at = c->con_lambda;
  hipLaunchKernelGGL(constraint_step_kernel, dim3((n + 63) / 64), dim3(64 * STEP_WAVES), (neq + 64 * STEP_WAVES) * sizeof(double), c->stream, c->slab,
                     c->slab_stride, c->con_dofs, c->con_nu, N, c->nq, c->nv, c->stage_x, neq, c->con_lambda, c->con_out,
                     c->con_out + n, c->alt_r);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(pl + neq, c->con_out, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));
  std::memcpy(step_host, pl + neq, (size_t)n * sizeof(double));
  std::memcpy(jtl_host, pl + neq + n, (size_t)n * sizeof(double));
  return FactorStatus(c);
}

int idto_hip_set_unactuated_dofs(idto_hip_ctx* c, const int* dofs, int nu) {
  HIP_OK(hipSetDevice(c->device));
  if (nu < 0 || nu > c->nv || (nu > 0 && !dofs)) { g_err = "set_unactuated_dofs: bad arguments"; return -1; }
  for (int j = 0; j < nu; ++j)
    if (dofs[j] < 0 || dofs[j] >= c->nv) { g_err = "set_unactuated_dofs: dof index out of range"; return -1; }
  // (every SolveFromWarmStart sets them: a device allocation and a copy only when the set changes - the allocation alone
  // was 0.1 ms of an MPC re-plan, and stayed allocated until the context went)
  if (nu == c->una_nu && (nu == 0 || (c->una_dofs && std::equal(dofs, dofs + nu, c->una_dofs_host.begin())))) return 0;
  c->una_nu = nu;
  c->una_dofs_host.assign(dofs, dofs + nu);
  if (nu > 0 && Upload(c, dofs, (size_t)nu, &c->una_dofs)) return -2;
  return 0;
}

// ---- device-side trust-region bookkeeping (trust_region.h)
static TrRowsArgs PrepareArgs(idto_hip_ctx* c, int scaling_method, int with_lambda) {
  const int n = (c->N + 1) * c->nq;
  TrRowsArgs A;
  A.nblk = c->N + 1; A.K = c->nq;
  A.HA = c->HA; A.HB = c->HB; A.HC = c->HC; A.g = c->g;
  A.jtl = with_lambda ? c->con_out + n : nullptr;
  A.yin = with_lambda ? c->con_out : c->step;
  A.ysign = with_lambda ? 1.0 : -1.0;
  A.q = c->q; A.scaling_method = scaling_method;
  A.Dprev = c->tr_Dprev; A.D = c->tr_D; A.gt = c->tr_gt; A.w = c->tr_w;
  A.slab = c->slab; A.slab_stride = c->slab_stride; A.tau_off = 3 * c->nv * c->nq;
  A.dofs = with_lambda ? c->con_dofs : c->una_dofs;
  A.nu = with_lambda ? c->con_nu : c->una_nu;
  A.N = c->N;
  A.lambda = with_lambda ? c->con_lambda_at : nullptr;
  A.partial = c->tr_part;
  A.freeze = nullptr;   // (idto_hip_tr_solve points it at the loop's sticky flags)
  A.part_ll = nullptr; A.epoch = 0u; A.dq_old = nullptr;   // (... and these at the hand-over between tr_iter_kernel's workgroups)
  A.kx = TrKkt{};
  return A;
}

static int EnqueuePrepare(idto_hip_ctx* c, int scaling_method, int with_lambda) {
  const int n = (c->N + 1) * c->nq;
  const int lds = tr_rows_lds(c->nq) * (int)sizeof(double);
  hipLaunchKernelGGL(tr_prepare_rows_kernel, dim3(c->N + 1), dim3(256), lds, c->stream,
                     PrepareArgs(c, scaling_method, with_lambda));
  hipLaunchKernelGGL(tr_prepare_sum_kernel, dim3(1), dim3(256), 0, c->stream, c->N + 1, c->tr_part, c->tr_out, c->tr_D,
                     c->tr_Dprev, n);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(c->tr_pin, c->tr_out, 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return 0;
}

int idto_hip_tr_set_scale_memory(idto_hip_ctx* c, const double* D_prev_host) {
  HIP_OK(hipSetDevice(c->device));
  if (c->batch != 1) { g_err = "tr_set_scale_memory serves single-problem contexts"; return -1; }
  const size_t n = (size_t)(c->N + 1) * c->nq;
  std::vector<double> ones;
  if (!D_prev_host) { ones.assign(n, 1.0); D_prev_host = ones.data(); }
  c->spec_ready = false;   // (a speculative tr_prepare used the previous memory)
  HIP_OK(hipMemcpyAsync(c->tr_Dprev, D_prev_host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_OK(hipStreamSynchronize(c->stream));   // (the host buffer may be a temporary)
  return 0;
}

int idto_hip_tr_set_convergence(idto_hip_ctx* c, const double* tolerances) {
  c->tr_conv_on = tolerances != nullptr;
  for (int i = 0; i < 6; ++i) c->tr_conv_tol[i] = tolerances ? tolerances[i] : 0.0;
  return 0;
}

int idto_hip_tr_prepare(idto_hip_ctx* c, int scaling_method, int with_lambda, double* out_host) {
  HIP_OK(hipSetDevice(c->device));
  if (c->batch != 1) { g_err = "tr_prepare serves single-problem contexts"; return -1; }
  if (scaling_method < -1 || scaling_method > 3) { g_err = "tr_prepare: bad scaling method"; return -1; }
  if (with_lambda && (!c->con_ready || !c->con_lambda_at)) {
    g_err = "tr_prepare: multipliers requested but no constraint step is resident";
    return -1;
  }
  if (c->spec_ready && !with_lambda && c->spec_scaling == scaling_method) {
    // the accepted trial point's iteration was enqueued speculatively by idto_hip_tr_trial
    c->spec_ready = false;
  } else {
    c->spec_ready = false;
    if (int rc = EnqueuePrepare(c, scaling_method, with_lambda)) return rc;
  }
  HIP_OK(hipStreamSynchronize(c->stream));
  std::memcpy(out_host, c->tr_pin, 9 * sizeof(double));
  return FactorStatus(c);
}

int idto_hip_tr_trial(idto_hip_ctx* c, double a, double b, int scaling, int normalize_quaternions, int with_lambda,
                      int speculate_scaling_method, double* out_host) {
  HIP_OK(hipSetDevice(c->device));
  if (c->batch != 1) { g_err = "tr_trial serves single-problem contexts"; return -1; }
  c->spec_pending = false; c->spec_ready = false;
  const int n = (c->N + 1) * c->nq;
  DropPrefetch(c, {IDTO_ARR_V, IDTO_ARR_A, IDTO_ARR_NPLUS, IDTO_ARR_SLAB, IDTO_ARR_COST});
  hipLaunchKernelGGL(tr_trial_kernel, dim3(1), dim3(1024), (size_t)2 * (n + c->N + 1) * sizeof(double), c->stream, n, c->nq, c->tr_D, c->tr_gt,
                     c->tr_w, a, b, scaling, c->q, c->q_trial, c->tr_dq, c->tr_quat,
                     normalize_quaternions ? c->tr_nquat : 0, c->tr_out + 9);
  HIP_OK(hipGetLastError());
  // tau and the cost at the trial point: the same kernels as idto_hip_eval_tau, reading q_trial
  std::swap(c->q, c->q_trial);
  int rc = LaunchFd(c, 0, 0, c->N);
  if (!rc) {
    hipLaunchKernelGGL(cost_kernel, dim3(1), dim3(1024), c->cost_lds, c->stream, c->M, c->P, c->q, c->v, c->slab,
                       c->slab_stride, c->cost, c->weights_diagonal ? 1 : 0, (double*)nullptr, (size_t)0, c->tr_out + 11, TrDecideArgs{}, AltSel{nullptr, 0, 0});
    if (with_lambda)
      hipLaunchKernelGGL(tr_hlambda_kernel, dim3(1), dim3(256), 16 * sizeof(double), c->stream, c->slab, c->slab_stride,
                         3 * c->nv * c->nq, c->con_dofs, c->con_nu, c->N, c->con_lambda_at, c->tr_out + 12);
  }
  std::swap(c->q, c->q_trial);
  if (rc) return rc;
  HIP_OK(hipGetLastError());
  c->fd_full = false; c->partials_ahead = false;
  c->trial_resident = true;
  HIP_OK(hipMemcpyAsync(c->tr_pin + 9, c->tr_out + 9, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  const bool speculate = speculate_scaling_method >= -1 && !with_lambda && FusedEligible(c) &&
                         speculate_scaling_method != 1 && speculate_scaling_method != 3;  // (adaptive scalings update D in place)
  if (speculate) {
    // the next iteration on the trial point, before the host has seen its cost: accepted steps (the
    // common case) find g, H, the Newton step and the inner products ready; a rejected step costs
    // one more idto_hip_gn_step on the old q
    if (!c->spec_ev) HIP_OK(hipEventCreateWithFlags(&c->spec_ev, hipEventDisableTiming));
    HIP_OK(hipEventRecord(c->spec_ev, c->stream));
    std::swap(c->q, c->q_trial);
    int rs = LaunchFused(c);
    if (!rs) rs = EnqueuePrepare(c, speculate_scaling_method, 0);
    std::swap(c->q, c->q_trial);
    if (rs) return rs;
    c->spec_pending = true;
    c->spec_scaling = speculate_scaling_method;
    HIP_OK(hipEventSynchronize(c->spec_ev));   // the trial point's scalars only
  } else {
    HIP_OK(hipStreamSynchronize(c->stream));
  }
  out_host[0] = c->tr_pin[9];    // dq . dq
  out_host[1] = c->tr_pin[10];   // g~ . D^-1 dq
  out_host[2] = c->tr_pin[11];   // cost at q + dq
  out_host[3] = with_lambda ? c->tr_pin[12] : 0.0;  // h(q + dq) . lambda
  return 0;
}

int idto_hip_tr_accept(idto_hip_ctx* c) {
  if (!c->trial_resident) { g_err = "tr_accept: no trial point is resident"; return -1; }
  std::swap(c->q, c->q_trial);   // v, a, tau, N+ and the cost in device memory already belong to it
  c->trial_resident = false;
  c->spec_ready = c->spec_pending;   // ... and so does the speculative iteration, if one was enqueued
  c->spec_pending = false;
  c->con_ready = false; c->con_begun = false;
  DropPrefetch(c, {IDTO_ARR_Q});
  return 0;
}

int idto_hip_tr_reject(idto_hip_ctx* c) {
  // the trial point is dropped: v, a, tau (and, after a speculative launch, the partials, g, H and the
  // Newton step) in device memory do not belong to the resident q any more
  c->trial_resident = false; c->spec_pending = false; c->spec_ready = false;
  c->fd_full = false; c->partials_ahead = false;
  c->con_ready = false; c->con_begun = false;
  return 0;
}

// The solver-only context of the KKT system [H J^T; J 0] (kkt.h): block size nq + nu, the parent's horizon, stream and
// solver options; only what LaunchLdl / FactorStatus touch is allocated.
static int MakeKkt(idto_hip_ctx* c, int nu) {
  if (c->kkt && c->kkt_nu == nu) return 0;
  if (c->kkt) { idto_hip_destroy(c->kkt); c->kkt = nullptr; }
  const int K = c->nq + nu, N = c->N, B = c->batch;
  if (K > 32) { g_err = "tr_solve: nq + nu <= 32 for the banded equality-constraint step"; return -1; }
  std::unique_ptr<idto_hip_ctx> k(new idto_hip_ctx);
  k->device = c->device; k->stream = c->stream; k->own_stream = false;
  k->batch = B; k->nq = K; k->nv = 0; k->N = N; k->dt = c->dt;
  // (seven workgroups for allegro's 29 x 29 blocks: 0.46 -> 0.29 ms per iteration; the small systems stay on two - the
  // nested-dissection order buys them 3 us and costs acrobot's multipliers a digit: 2e-8 against 3e-9)
  k->two_sided = c->two_sided; k->solver_nd = c->solver_nd && (K == 29 || K == 8); k->solver_pipe = false; k->fused = false; k->asm_in_solver = false;
  k->solver_band = c->solver_band; k->nd_min_rows = c->nd_min_rows; k->nd_recursion = c->nd_recursion; k->solver_debug = c->kkt_debug; k->debug_skip_role = c->debug_skip_role;
  k->h_assembled = true;      // block row 0 is decoupled (q_0 is no variable, mu_0 a dummy): the chains start at row 1
  k->ldl_npos = c->nq;
  const size_t kk = (size_t)K * K;
  idto_hip_ctx* kc = k.get();
  auto fail = [&](const char* what) { g_err = what; idto_hip_destroy(k.release()); return -2; };
  // one arena per problem, as in idto_hip_create_batch: the solver kernels take the problem from blockIdx.y
  size_t top = 0;
  auto carve = [&](size_t count, size_t elem) {
    const size_t o = (top + 63) & ~(size_t)63;
    top = o + std::max<size_t>(count, 1) * elem;
    return o;
  };
  const size_t D = sizeof(double);
  kc->xch_count = 2 * (size_t)(3 * 32 + 1) * ldl_ks(32) + 2 * 32;
  kc->flag_count = 16;
  const size_t o_H = carve((size_t)3 * (N + 6) * kk, D), o_g = carve((size_t)(N + 1) * K, D), o_step = carve((size_t)(N + 1) * K, D);
  const size_t o_U = carve((size_t)(N + 1) * 32 * 36, D), o_Hs = carve((size_t)(N + 1) * 32 * 36, D), o_E = carve((size_t)(N + 1) * 32 * 36, D);
  const size_t o_Ds = carve((size_t)(N + 1) * 32, D), o_dbg = carve((size_t)(N + 4) * 8 * 32, D);
  const size_t o_xch = carve(2 * kc->xch_count, D), o_flags = carve(kc->flag_count, sizeof(unsigned));
  const size_t o_ndcnt = carve(4 * ND_MAXROWS, sizeof(unsigned long long)), o_pipecnt = carve(4 * ND_MAXROWS, sizeof(unsigned long long));
  const size_t o_ndbuf = carve((size_t)nd_layout(32).end, D);
  const bool nd_wst_on = K > 20 && K <= 32;
  const size_t o_ndwst = carve(nd_wst_on ? 2 * (size_t)ND_MAXROWS * nd_layout(K).frow : 1, D);
  kc->pstride = (top + 255) & ~(size_t)255;
  {
    void* p = nullptr;
    if (hipMalloc(&p, kc->pstride * (size_t)B) != hipSuccess || hipMemset(p, 0, kc->pstride * (size_t)B) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess)
      return fail("hipMalloc (KKT context) failed");
    kc->allocs.push_back(p);
    kc->arena = static_cast<char*>(p);
    if (getenv("IDTO_DEBUG_PTRS"))   // (investigation aid: where a faulting address lies)
      fprintf(stderr, "kkt arena %p .. %p (pstride %zu x %d): H %zu g %zu step %zu U %zu Hs %zu E %zu Ds %zu dbg %zu xch %zu flags %zu ndcnt %zu pipecnt %zu ndbuf %zu ndwst %zu\n",
              p, (void*)((char*)p + kc->pstride * (size_t)B), kc->pstride, B, o_H, o_g, o_step, o_U, o_Hs, o_E, o_Ds, o_dbg, o_xch, o_flags, o_ndcnt, o_pipecnt, o_ndbuf, o_ndwst);
  }
  auto dp = [&](size_t o) { return reinterpret_cast<double*>(kc->arena + o); };
  kc->HA = dp(o_H); kc->HB = kc->HA + (size_t)(N + 6) * kk; kc->HC = kc->HB + (size_t)(N + 6) * kk;
  kc->g = dp(o_g); kc->step = dp(o_step); kc->Ust = dp(o_U); kc->Hst = dp(o_Hs); kc->Est = dp(o_E); kc->Dst = dp(o_Ds); kc->dbg = dp(o_dbg);
  kc->xch = dp(o_xch); kc->flags = reinterpret_cast<unsigned*>(kc->arena + o_flags);
  kc->nd_rowcnt = reinterpret_cast<unsigned long long*>(kc->arena + o_ndcnt);
  kc->pipe_rowcnt = reinterpret_cast<unsigned long long*>(kc->arena + o_pipecnt);
  kc->nd_buf = dp(o_ndbuf);
  kc->nd_wst = nd_wst_on ? dp(o_ndwst) : nullptr;
  if (hipHostMalloc((void**)&kc->status_pin, (2 * (size_t)B + 2) * sizeof(unsigned), hipHostMallocDefault) != hipSuccess ||
      hipHostGetDevicePointer((void**)&kc->status_dev, kc->status_pin, 0) != hipSuccess)
    return fail("hipHostMalloc (KKT solver status) failed");
  for (int i = 0; i < 2 * B + 2; ++i) kc->status_pin[i] = 0;
  c->kkt = k.release(); c->kkt_nu = nu;
  return 0;
}

static bool AsmInSolver(idto_hip_ctx* c);
static bool DecideInSolver(idto_hip_ctx* c);
// Delta0s / Delta_out: one radius per problem of the context; rows_host: [batch][iterations][TRR_COUNT]
static int TrSolve(idto_hip_ctx* c, int iterations, int scaling_method, int scaling, int normalize_quaternions,
                   const double* Delta0s, double Delta_max, double eta, const int* constrained_dofs, int nu,
                   double* rows_host, double* Delta_out, double* const* fetch = nullptr) {
  HIP_OK(hipSetDevice(c->device));
  const int B = c->batch;
  TRACE("hip: tr_solve begins");
  if (iterations <= 0) { g_err = "tr_solve: iterations must be positive"; return -1; }
  if (!c->tr_resident_ok) {
    g_err = "tr_solve: the iteration kernel's workgroups timed out on this context before (option tr_resident_ok): use "
            "idto_hip_tr_prepare / trial / accept / reject";
    return -1;
  }
  if (nu < 0 || (nu > 0 && !constrained_dofs)) { g_err = "tr_solve: bad constraint arguments"; return -1; }
  const int neq = nu * c->N;
  if (nu > 0) {
    if (!c->weights_diagonal) { g_err = "tr_solve: enforced constraints need diagonal cost weights"; return -1; }
  }
  // (the adaptive methods, D = min(D_prev, f(diag H)): tr_iter_kernel keeps D_prev.  A rejected step leaves g and H
  // where they are - the gated assembly - so the next iteration forms min(D, f(diag H)) = D again: the memory does
  // not advance, as in the reference, whose cached scale factors belong to the state that did not change.)
  if (scaling_method < -1 || scaling_method > 3) { g_err = "tr_solve: bad scaling method"; return -1; }
  const bool adaptive = scaling_method == 1 || scaling_method == 3;
  if (iterations > c->tr_rows_cap) {
    double* p = nullptr;
    if (Alloc(c, (size_t)B * iterations * TRR_COUNT, &p)) return -2;   // (the previous, smaller one stays in the context's pool)
    c->tr_rows = p;
    c->tr_rows_cap = iterations;
  }
  const size_t rows_stride = (size_t)iterations * TRR_COUNT;
  c->spec_pending = false; c->spec_ready = false; c->trial_resident = false;
  const int n = (c->N + 1) * c->nq, nblk = c->N + 1;
  // tr_iter_kernel finds its last workgroup by counter == target: both restart with every solve, so that a
  // launch that failed in an earlier solve cannot leave them out of step
  c->tr_target = 0;
  // state of every problem: [Delta, L(q) (resident: the caller evaluated the cost of q), ...]
  if (B == 1) {
    // (one problem: one launch sets the state words, the cost among them, and the counter - no wait here, the loop is
    // enqueued while the evaluation of the initial guess still runs)
    static_assert(TRS_COUNT <= 16, "tr_pin holds 16 doubles");
    hipLaunchKernelGGL(tr_begin_kernel, dim3(1), dim3(64), 0, c->stream, c->tr_state, (int)TRS_COUNT, (int)TRS_DELTA,
                       (int)TRS_ACCEPTED, (int)TRS_COST, Delta0s[0], c->cost, c->tr_cnt);
    HIP_OK(hipGetLastError());
  } else {
    HIP_OK(hipMemset2DAsync(c->tr_cnt, c->pstride, 0, sizeof(unsigned long long), (size_t)B, c->stream));
    std::vector<double> st((size_t)B * TRS_COUNT, 0.0);
    for (int b = 0; b < B; ++b) { st[(size_t)b * TRS_COUNT + TRS_DELTA] = Delta0s[b]; st[(size_t)b * TRS_COUNT + TRS_ACCEPTED] = 1.0; }
    HIP_OK(hipMemcpy2DAsync(c->tr_state, c->pstride, st.data(), TRS_COUNT * sizeof(double), TRS_COUNT * sizeof(double), (size_t)B,
                            hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));   // (the host buffer is a temporary)
  }
  if (B != 1)
    HIP_OK(hipMemcpy2DAsync(c->tr_state + TRS_COST, c->pstride, c->cost, c->pstride, sizeof(double), (size_t)B,
                            hipMemcpyDeviceToDevice, c->stream));
  const double eps = 10 * std::numeric_limits<double>::epsilon() / c->P.dt / c->P.dt;   // TO.cc:2024
  const int lds_iter = (int)sizeof(double) * (tr_rows_lds(c->nq) + TR_NSUM * nblk + 3 + TR_NSUM + 3 * c->nq);
  // g, H and the Newton step of the first iterate (with constraints the step comes out of the multiplier chain)
  int rc = 0;
  if (nu > 0) {
    rc = idto_hip_eval_partials(c);
    if (!rc) rc = idto_hip_grad_hess(c);
  } else {
    rc = idto_hip_gn_step(c);
  }
  if (rc) return rc;
  // (blocks of nq + nu up to the 30 x 30 instantiation of the two-workgroup factorisation: allegro's 23 + 6, N = 60,
  // 0.523 against 0.665 ms per iteration with the Schur-complement chain; the 32 x 32 one needs a third elimination
  // wavefront and spills: 442 against 245 us at N = 20)
  const bool use_kkt = nu > 0 && c->con_kkt && SolverBlockSize(c->nq + nu, true) <= 30;
  if (B != 1 && nu > 0 && !use_kkt) { g_err = "tr_solve: enforced constraints on a batch need the banded KKT step (nq + nu <= 30, option con_kkt)"; return -1; }
  if (use_kkt) {
    if ((rc = ConstraintDofs(c, constrained_dofs, nu)) != 0) return rc;
    if ((rc = MakeKkt(c, nu)) != 0) return rc;
  }
  const bool lookahead = c->weights_diagonal && c->asm_stop == 0 && c->fd_stop == 0;
  if (B != 1 && !lookahead) { g_err = "tr_solve on a batch context needs the two-set evaluation (diagonal cost weights)"; return -1; }
  if (adaptive && !lookahead) { g_err = "tr_solve: the adaptive scalings need the gated assembly (diagonal cost weights)"; return -1; }
  if (nu > 0 && !lookahead) { g_err = "tr_solve: enforced constraints need the two-set evaluation"; return -1; }
  // from here on the trial point's v, a, N+, tau, partials go to the set the iterate does not occupy
  struct AltGuard {
    idto_hip_ctx* c;
    ~AltGuard() { c->alt_r.state = nullptr; c->alt_w.state = nullptr; }
  } alt_guard{c};
  if (lookahead) {
    c->alt_r = AltSel{c->tr_state, c->alt_off, 0};
    c->alt_w = AltSel{c->tr_state, c->alt_off, 1};
  }
  // (the small models' one-workgroup launch, gn_small.h: unconstrained, or with ONE enforced constraint through the banded
  // KKT step - the instantiations: acrobot 2 + 1, spinner 3 + 1 - whose context the scalar band solver takes)
  bool tr_small = c->tr_small && lookahead && SmallEligible(c) && c->h_assembled;
  if (tr_small && nu > 0) {
    LdlPlan kp;
    tr_small = use_kkt && nu == 1 && c->kkt->batch == c->batch && PlanLdl(c->kkt, false, &kp) == 0 && BandEligible(c->kkt, kp, true) &&
               kp.k == c->nq + 1 && SmallLds(c, kp, nullptr) <= 160 * 1024;
  }
  bool kkt_built = false;
  bool tr_fold_fits = false;
  if (tr_small) {   // (the folded iteration's arrays live in the band solver's carve-up)
    LdlPlan sp;
    int band = 0;
    if (PlanLdl(nu > 0 ? c->kkt : c, false, &sp) == 0) { (void)SmallLds(c, sp, &band); tr_fold_fits = gn_small_fold_doubles(c->N, c->nq) <= band; }
  }
  TrConvergence conv{};
  conv.on = c->tr_conv_on ? 1 : 0;
  conv.rel_cost = c->tr_conv_tol[0]; conv.abs_cost = c->tr_conv_tol[1]; conv.rel_grad = c->tr_conv_tol[2];
  conv.abs_grad = c->tr_conv_tol[3]; conv.rel_state = c->tr_conv_tol[4]; conv.abs_state = c->tr_conv_tol[5];
  conv.rows = c->tr_rows;
  // (with convergence checks the last iteration, too, is followed by g at its iterate: one more pass of the loop
  // body up to tr_iter_kernel, which then only evaluates the criteria)
  const int passes = iterations + (conv.on ? 1 : 0);
  // With the convergence criteria on, the loop is enqueued in chunks of TR_CHUNK iterations and the sticky flags
  // are looked at in between: once a criterion holds (or an iteration failed) the iterations that are left would run
  // every kernel in full - none of fd, cost and the solver reads the flags - only to decide nothing, and a solve that
  // converges after 20 of max_iterations = 500 would take 25 times as long as the loop it replaces (TO.cc:2600-2612
  // leaves the loop).  Rows of iterations that never ran are zeros.
  constexpr int TR_CHUNK = 8;
  if (conv.on) HIP_OK(hipMemsetAsync(c->tr_rows, 0, (size_t)B * rows_stride * sizeof(double), c->stream));
  TRACE("hip: tr_solve: loop state enqueued, constraint buffers ready");
  for (int k = 0; k < passes; ++k) {
    if (conv.on && B == 1 && k > 0 && k % TR_CHUNK == 0) {
      HIP_OK(hipMemcpyAsync(c->tr_pin, c->tr_state, TRS_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIP_OK(hipStreamSynchronize(c->stream));
      if (c->tr_pin[TRS_FLAGS] != 0.0) break;
    }
    conv.check_only = (k == iterations) ? 1 : 0;
    bool kkt_fold = false;
    KktExtractArgs kkt_ex{};
    if (nu > 0 && use_kkt) {
      // multipliers of the iterate and H^-1 (g + J^T lambda) (TO.cc:1371-1396, :2139-2149) from ONE banded solve of the
      // KKT system (kkt.h): build its bands from H and the slab's rows of J, factorise + solve, take the result apart
      // (tr_small: from the second iteration on the launch that evaluated, decided and assembled has solved it too)
      idto_hip_ctx* kc = c->kkt;
      const bool solved = tr_small && k > 0;
      const bool built = kkt_built;   // (the assembly of the previous iteration wrote the KKT system along: option "kkt_in_asm")
      kkt_built = false;
      KktBuildArgs Kb;
      Kb.N = c->N; Kb.nq = c->nq; Kb.nv = c->nv; Kb.nu = nu;
      Kb.HA = c->HA; Kb.HB = c->HB; Kb.HC = c->HC; Kb.g = c->g;
      Kb.slab = c->slab; Kb.slab_stride = c->slab_stride; Kb.dofs = c->con_dofs;
      Kb.KA = kc->HA; Kb.KB = kc->HB; Kb.KC = kc->HC; Kb.rhs = kc->g; Kb.alt = c->alt_r;
      Kb.pstride = c->pstride; Kb.kstride = kc->pstride;
      if (!solved) {
        if (!built) hipLaunchKernelGGL(kkt_build_kernel, dim3(c->N + 1, B), dim3(256), 0, c->stream, Kb);
        HIP_OK(hipGetLastError());
        rc = idto_hip_factor_solve(kc, nullptr, 1, nullptr);
        if (rc) return rc;
      }
      // (taking z apart - w, J^T lambda, lambda, the multiplier pivots' range - is folded into tr_iter_kernel below; option
      // "kkt_fold" 0 keeps kkt_extract_kernel's launch: tests hold the two against each other)
      KktExtractArgs Ke;
      Ke.N = c->N; Ke.nq = c->nq; Ke.nv = c->nv; Ke.nu = nu;
      Ke.z = kc->step; Ke.slab = c->slab; Ke.slab_stride = c->slab_stride; Ke.dofs = c->con_dofs;
      Ke.w = c->con_out; Ke.jtl = c->con_out + n; Ke.lambda = c->con_lambda;
      Ke.Dinv = kc->Dst; Ke.dstride = SolverBlockSize(kc->nq, true); Ke.first_row = SolverFirstRow(kc);
      Ke.state = c->tr_state; Ke.alt = c->alt_r;
      Ke.pstride = c->pstride; Ke.kstride = kc->pstride;
      kkt_fold = c->kkt_fold;
      if (kkt_fold) kkt_ex = Ke;
      else hipLaunchKernelGGL(kkt_extract_kernel, dim3(c->N + 1, B), dim3(64), 0, c->stream, Ke);
      HIP_OK(hipGetLastError());
      c->con_lambda_at = c->con_lambda;
      c->con_ready = false; c->con_begun = false;
    } else if (nu > 0) {
      // multipliers of the iterate (TO.cc:1371-1396), all on the device: Y = H^-1 [g | J^T], S = J Y_J, J y_g
      // (idto_hip_constraint_schur_begin), lambda = S^-1 (h - J y_g) in one workgroup, then H^-1 (g + J^T lambda)
      // and J^T lambda (constraint_step_kernel)
      c->con_begun = false;
      rc = idto_hip_constraint_schur_begin(c, constrained_dofs, nu);
      if (rc) return rc;
      if (neq <= CON_LAMBDA_MAX) {   // S in the LDS of one workgroup
        hipLaunchKernelGGL(constraint_lambda_kernel, dim3(1), dim3(CON_LAMBDA_THREADS),
                           (size_t)con_lambda_lds_doubles(neq) * sizeof(double), c->stream,
                           c->con_S, neq, c->slab, c->slab_stride, 3 * c->nv * c->nq, c->con_dofs, nu, c->con_lambda, c->tr_state,
                           c->alt_r, c->solver_debug ? c->dbg : (double*)nullptr);
        c->con_lambda_at = c->con_lambda;
      } else {                       // the blocked factorisation of dense_ldl.h, as idto_hip_constraint_solve runs it
        hipLaunchKernelGGL(constraint_h_kernel, dim3((neq + 255) / 256), dim3(256), 0, c->stream, c->slab, c->slab_stride,
                           3 * c->nv * c->nq, c->con_dofs, nu, neq, c->con_h, c->alt_r);
        double* S = c->con_S;
        LaunchDenseLdl(c, S, neq, S + (size_t)neq * neq, -1.0, c->con_h + 2);
        c->con_S_factored = true;
        hipLaunchKernelGGL(dense_ldl_solve_kernel, dim3(1), dim3(512), (neq + 512 + DENSE_NB * (DENSE_NB + 1)) * sizeof(double),
                           c->stream, c->con_L, neq, c->con_d, c->con_rv + neq, 1.0, (const double*)nullptr, c->con_lambda + 2, 1);
        hipLaunchKernelGGL(constraint_flag_kernel, dim3(1), dim3(1), 0, c->stream, c->con_h, c->tr_state);
        c->con_lambda_at = c->con_lambda + 2;
      }
      hipLaunchKernelGGL(constraint_step_kernel, dim3((n + 63) / 64), dim3(64 * STEP_WAVES),
                         (neq + 64 * STEP_WAVES) * sizeof(double), c->stream, c->slab, c->slab_stride, c->con_dofs, nu, c->N,
                         c->nq, c->nv, c->stage_x, neq, c->con_lambda_at, c->con_out, c->con_out + n, c->alt_r);
      HIP_OK(hipGetLastError());
      c->con_ready = true;
    }
    TrIterArgs T;
    T.rows = PrepareArgs(c, scaling_method, nu > 0 ? 1 : 0);
    T.alt = c->alt_r;
    T.rows.part_ll = c->tr_part_ll;
    T.rows.epoch = ++c->tr_epoch;
    if (T.rows.epoch == 0u) T.rows.epoch = ++c->tr_epoch;   // (0 is what an arena that was never written holds)
    T.part2 = c->tr_part2;
    T.out = c->tr_out; T.state = c->tr_state;
    T.n = n; T.nq = c->nq; T.scaling = scaling;
    T.nquat = normalize_quaternions ? c->tr_nquat : 0;
    T.quat = c->tr_quat; T.q_trial = c->q_trial; T.dq = c->tr_dq;
    T.conv = conv;
    T.fact_status = c->status_dev; T.fact_id = c->fact_id;   // (the most recent factorisation: this iteration's step)
    T.timeout_status = c->status_dev + 2 * c->batch;
    if (nu > 0 && use_kkt) { T.fact_status = c->kkt->status_dev; T.fact_id = c->kkt->fact_id; T.timeout_status = c->kkt->status_dev + 2 * B; }
    T.rows.freeze = c->tr_state + TRS_FLAGS;
    T.pstride = c->pstride; T.rows_stride = rows_stride;
    T.kdinv = nullptr; T.kdstride = 0; T.kfirst_row = 0; T.kstride = 0;
    T.debug_skip_row = c->debug_skip_role >= 100 ? c->debug_skip_role - 100 : -1;
    T.curpre = c->decide_word + 2;
    if (kkt_fold) {
      T.rows.kx.z = kkt_ex.z; T.rows.kx.KK = c->nq + nu; T.rows.kx.nv = c->nv;
      T.rows.kx.w_out = kkt_ex.w; T.rows.kx.jtl_out = kkt_ex.jtl; T.rows.kx.lambda_out = kkt_ex.lambda;
      T.kdinv = kkt_ex.Dinv; T.kdstride = kkt_ex.dstride; T.kfirst_row = kkt_ex.first_row; T.kstride = kkt_ex.kstride;
    }
    // (the small models' launch below takes this kernel's part too - option "tr_fold" -: the whole iteration is one launch)
    const bool fold = tr_small && tr_fold_fits && c->tr_fold && k < iterations && T.nquat == 0 && T.rows.nu <= 4;
    if (!fold) hipLaunchKernelGGL(tr_iter_kernel, dim3(nblk + 1, B), dim3(256), lds_iter, c->stream, T);
    HIP_OK(hipGetLastError());
    if (k == iterations) break;   // (the check-only pass)
    // tau (with its partials: the trial point is the next iterate unless rejected) and the cost at the
    // trial point, then the decision (cost_kernel's epilogue)
    TrDecideArgs Dc;
    Dc.state = c->tr_state; Dc.out = c->tr_out; Dc.rows = c->tr_rows; Dc.q = c->q; Dc.q_trial = c->q_trial; Dc.n = n;
    Dc.rows_stride = rows_stride;
    Dc.eta = eta; Dc.Delta_max = Delta_max; Dc.eps = eps;
    Dc.lambda = nu > 0 ? c->con_lambda_at : nullptr; Dc.dofs = nu > 0 ? c->con_dofs : nullptr;
    Dc.nu = nu; Dc.N = c->N; Dc.slab_stride = c->slab_stride; Dc.tau_off = 3 * c->nv * c->nq;
    Dc.part2 = c->tr_part2; Dc.nblk = nblk;
    const bool more = k + 1 < passes;
    if (tr_small) {
      // a small all-revolute model: the trial point's evaluation, its cost, the decision and - accepted - g, H and the next
      // step in ONE workgroup of ONE launch (gn_small.h), the bits of the three launches below
      rc = LaunchSmall(c, &Dc, !more, c->alt_w, nu > 0, fold ? &T : nullptr);
      if (rc) return rc;
      if (!more) break;
      continue;
    }
    std::swap(c->q, c->q_trial);
    rc = LaunchFd(c, (lookahead && more) ? 1 : 0, 0, c->N, c->alt_w);
    c->fd_full = lookahead && more;   // (v, N+ of the trial point = of the iterate the gated assembly runs for)
    // the cost of the trial point and the decision: a launch of their own (cost_kernel), or one more workgroup of the
    // pipelined solver's launch that follows (penta_pipe.h PipeAsm::decide, option "decide_in_solver")
    const bool decide_in_solver = !rc && more && lookahead && nu == 0 && c->decide_in_solver && AsmInSolver(c) && DecideInSolver(c);
    if (!rc && !decide_in_solver)
      hipLaunchKernelGGL(cost_kernel, dim3(1, B), dim3(1024), c->cost_lds, c->stream, c->M, c->P, c->q, c->v, c->slab,
                         c->slab_stride, c->cost, c->weights_diagonal ? 1 : 0, (double*)nullptr, c->pstride,
                         (double*)nullptr, Dc, c->alt_w);
    std::swap(c->q, c->q_trial);
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    if (!more) break;
    if (lookahead && nu == 0 && AsmInSolver(c)) {
      // the gated assembly inside the pipelined solver's launch (penta_pipe.h PipeAsm): a problem whose step was
      // rejected keeps g and H and the solver reads them where they are -> the same step
      DropPrefetch(c, {IDTO_ARR_GRADIENT, IDTO_ARR_H_A, IDTO_ARR_H_B, IDTO_ARR_H_C, IDTO_ARR_HBANDS});
      c->con_ready = false; c->con_begun = false;
      c->fuse_asm_next = true;
      c->fuse_gate = c->tr_state + TRS_ACCEPTED;
      c->fuse_decide = decide_in_solver ? &Dc : nullptr;
      rc = idto_hip_factor_solve(c, nullptr, 1, nullptr);
      c->fuse_gate = nullptr;
      if (c->fuse_decide) {
        c->fuse_decide = nullptr;
        if (!rc) { g_err = "tr_solve: the solver's launch that was to decide on the trial point did not run"; rc = -1; }
      }
      if (c->fuse_asm_next) {
        c->fuse_asm_next = false;
        if (!rc) { g_err = "tr_solve: the solver that was to assemble g and H did not run"; rc = -1; }
      }
    } else if (lookahead) {
      if (nu > 0 && use_kkt && c->kkt_in_asm) {
        // (the next iteration's KKT system written by this assembly: a rejected step keeps g, H - and the system, which the
        // solvers only read)
        idto_hip_ctx* kc = c->kkt;
        KktSink S{};
        S.KA = kc->HA; S.KB = kc->HB; S.KC = kc->HC; S.rhs = kc->g; S.K = c->nq + nu; S.nu = nu;
        S.slab = c->slab; S.slab_stride = (int)c->slab_stride; S.dofs = c->con_dofs;
        rc = LaunchAssemble(c, c->tr_state + TRS_ACCEPTED, &S, kc->pstride, &kkt_built);
      } else {
        rc = LaunchAssemble(c, c->tr_state + TRS_ACCEPTED);
      }
      if (!rc && nu == 0) rc = idto_hip_factor_solve(c, nullptr, 1, nullptr);   // (after a rejection: the same H, g -> the same step)
    } else {
      rc = idto_hip_gn_step(c);   // (dense cost weights: the partials again, at the iterate)
    }
    if (rc) return rc;
  }
  c->fd_full = false; c->partials_ahead = false;
  TRACE("hip: tr_solve: every iteration enqueued");
  if (B != 1) {
    // every problem has its own current set: the ones whose iterate ended up in the other set get it copied over
    // (a single-problem context swaps its pointers instead, below)
    if (c->alt_off <= 0) { g_err = "tr_solve: batch contexts keep their primary output set"; return -1; }
    hipLaunchKernelGGL(tr_fold_sets_kernel, dim3(64, B), dim3(256), 0, c->stream, c->v, (size_t)c->alt_off / sizeof(double),
                       c->tr_state, c->pstride);
    HIP_OK(hipGetLastError());
    std::vector<double> st((size_t)B * TRS_COUNT);
    HIP_OK(hipMemcpy2DAsync(st.data(), TRS_COUNT * sizeof(double), c->tr_state, c->pstride, TRS_COUNT * sizeof(double), (size_t)B,
                            hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(rows_host, c->tr_rows, (size_t)B * rows_stride * sizeof(double), hipMemcpyDeviceToHost));
    if (Delta_out) for (int b = 0; b < B; ++b) Delta_out[b] = st[(size_t)b * TRS_COUNT + TRS_DELTA];
    for (int b = 0; b < B; ++b) {
      if (use_kkt) {   // (see the single-problem exit below)
        const int fs = FactorStatus(c->kkt, b);
        const bool singular = ((int)st[(size_t)b * TRS_COUNT + TRS_FLAGS] & TRF_SINGULAR_S) != 0;
        if (fs == IDTO_HIP_SOLVER_TIMEOUT || (fs && !singular)) return fs;
      }
      if (int fs = FactorStatus(c, b)) return fs;
    }
    return 0;
  }
  if (fetch) {
    // ... and the solution with them (idto_hip_tr_solve_fetch): one gathering launch, one copy, the same wait
    const size_t nrow = (size_t)iterations * TRR_COUNT;
    const size_t nqa = (size_t)(c->N + 1) * c->nq, nva = (size_t)(c->N + 1) * c->nv, nta = (size_t)c->N * c->nv;
    const size_t total = TRS_COUNT + nrow + nqa + nva + nta + 2 * nqa;
    if (c->fetch_cap < total) {
      Release(c, &c->fetch_dev);
      if (c->fetch_pin) (void)hipHostFree(c->fetch_pin);
      c->fetch_pin = nullptr; c->fetch_cap = 0;
      if (Alloc(c, total, &c->fetch_dev)) return -2;
      HIP_OK(hipHostMalloc((void**)&c->fetch_pin, total * sizeof(double), hipHostMallocDefault));
      c->fetch_cap = total;
    }
    TrGatherArgs G;
    G.state = c->tr_state; G.rows = c->tr_rows; G.nstate = TRS_COUNT; G.nrows = (int)nrow;
    G.q = c->q; G.v = c->v; G.slab = c->slab; G.dq = c->tr_dq; G.w = c->tr_w;
    G.N = c->N; G.nq = c->nq; G.nv = c->nv; G.slab_stride = (int)c->slab_stride;
    G.alt_off = lookahead ? (long long)c->alt_off : 0;
    G.out = c->fetch_dev;
    hipLaunchKernelGGL(tr_gather_kernel, dim3(16), dim3(256), 0, c->stream, G);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(c->fetch_pin, c->fetch_dev, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRACE("hip: tr_solve: gather + one copy enqueued");
    HIP_OK(hipStreamSynchronize(c->stream));
    TRACE("hip: tr_solve: waited for the device (state words, rows and the solution back)");
    const double* p = c->fetch_pin;
    std::memcpy(c->tr_pin, p, TRS_COUNT * sizeof(double)); p += TRS_COUNT;
    std::memcpy(rows_host, p, nrow * sizeof(double)); p += nrow;
    const size_t cnt[5] = {nqa, nva, nta, nqa, nqa};
    for (int i = 0; i < 5; ++i) { if (fetch[i]) std::memcpy(fetch[i], p, cnt[i] * sizeof(double)); p += cnt[i]; }
  } else {   // the state words and the statistics rows with ONE wait (the rows through pinned staging of the context)
    const size_t nrow = (size_t)iterations * TRR_COUNT;
    if (c->rows_cap < nrow) {
      if (c->rows_pin) (void)hipHostFree(c->rows_pin);
      c->rows_pin = nullptr; c->rows_cap = 0;
      HIP_OK(hipHostMalloc((void**)&c->rows_pin, std::max<size_t>(nrow, 64 * TRR_COUNT) * sizeof(double), hipHostMallocDefault));
      c->rows_cap = std::max<size_t>(nrow, 64 * TRR_COUNT);
    }
    HIP_OK(hipMemcpyAsync(c->tr_pin, c->tr_state, TRS_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipMemcpyAsync(c->rows_pin, c->tr_rows, nrow * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    TRACE("hip: tr_solve: waited for the device (state words + rows back)");
    std::memcpy(rows_host, c->rows_pin, nrow * sizeof(double));
  }
  if (Delta_out) *Delta_out = c->tr_pin[TRS_DELTA];
  if (c->tr_pin[TRS_CUR] != 0.0) {   // the iterate's v, a, N+, slab, products ended up in the other set: it is "the" set now
    c->v = at_problem(c->v, (size_t)c->alt_off); c->a = at_problem(c->a, (size_t)c->alt_off);
    c->nplus = at_problem(c->nplus, (size_t)c->alt_off); c->slab = at_problem(c->slab, (size_t)c->alt_off);
    c->terms = at_problem(c->terms, (size_t)c->alt_off);
    c->alt_off = -c->alt_off;
  }
  if (use_kkt) {
    // (a multiplier pivot that vanished - redundant constraints, TRF_SINGULAR_S in the rows: the caller's pivoted
    // factorisation takes over - leaves garbage in the rows of H behind it, whose pivot test then fails as well)
    const int fs = FactorStatus(c->kkt);
    const bool singular = ((int)c->tr_pin[TRS_FLAGS] & TRF_SINGULAR_S) != 0;
    if (fs == IDTO_HIP_SOLVER_TIMEOUT || (fs && !singular)) return fs;
  }
  return FactorStatus(c);
}

int idto_hip_tr_solve(idto_hip_ctx* c, int iterations, int scaling_method, int scaling, int normalize_quaternions,
                      double Delta0, double Delta_max, double eta, const int* constrained_dofs, int nu,
                      double* rows_host, double* Delta_out) {
  if (c->batch != 1) { g_err = "tr_solve serves single-problem contexts (batches: idto_hip_tr_solve_batch)"; return -1; }
  return TrSolve(c, iterations, scaling_method, scaling, normalize_quaternions, &Delta0, Delta_max, eta, constrained_dofs, nu,
                 rows_host, Delta_out);
}

int idto_hip_tr_solve_fetch(idto_hip_ctx* c, int iterations, int scaling_method, int scaling, int normalize_quaternions,
                            double Delta0, double Delta_max, double eta, const int* constrained_dofs, int nu,
                            double* rows_host, double* Delta_out, double* q_out, double* v_out, double* tau_out, double* dq_out,
                            double* w_out) {
  if (c->batch != 1) { g_err = "tr_solve_fetch serves single-problem contexts"; return -1; }
  double* const fetch[5] = {q_out, v_out, tau_out, dq_out, w_out};
  return TrSolve(c, iterations, scaling_method, scaling, normalize_quaternions, &Delta0, Delta_max, eta, constrained_dofs, nu,
                 rows_host, Delta_out, fetch);
}

int idto_hip_tr_solve_batch(idto_hip_ctx* c, int iterations, int scaling_method, int scaling, int normalize_quaternions,
                            const double* Delta0, double Delta_max, double eta, double* rows_host, double* Delta_out) {
  if (!Delta0 || !rows_host) { g_err = "tr_solve_batch: Delta0[batch] and rows_host[batch][iterations][IDTO_TR_ROW] are required"; return -1; }
  return TrSolve(c, iterations, scaling_method, scaling, normalize_quaternions, Delta0, Delta_max, eta, nullptr, 0, rows_host,
                 Delta_out);
}

// The trust-region loop with ENFORCED equality constraints for every problem of a batch context (BASELINE config 5:
// examples/allegro_hand/allegro_hand.yaml:95 `equality_constraints : true`; reference TO.cc:1267-1396, :2495-2625).
// The multiplier chain of an iteration - H^-1 [g | J^T] for n_eq + 1 right-hand sides, S = J H^-1 J^T, its dense
// factorisation, the step - is a sequence of single-problem launches with buffers of its own (constraints.h, dense_ldl.h,
// penta_apply.h), so the batch runs as one single-problem context per problem (created on first use from the host copies
// of the model and the problems), each advanced by idto_hip_tr_solve on a stream and a host thread of its own: the
// problems' launches overlap on the device, every problem's rows are bit for bit what idto_hip_tr_solve gives it alone,
// and the iterates are written back into the batch's arenas.
int idto_hip_tr_solve_batch_constrained(idto_hip_ctx* c, int iterations, int scaling_method, int scaling, int normalize_quaternions,
                                        const double* Delta0, double Delta_max, double eta, const int* constrained_dofs, int nu,
                                        double* rows_host, double* Delta_out) {
  if (!Delta0 || !rows_host) { g_err = "tr_solve_batch_constrained: Delta0[batch] and rows_host[batch][iterations][IDTO_TR_ROW] are required"; return -1; }
  if (nu <= 0 || !constrained_dofs)
    return idto_hip_tr_solve_batch(c, iterations, scaling_method, scaling, normalize_quaternions, Delta0, Delta_max, eta, rows_host, Delta_out);
  const int B = c->batch;
  if (B == 1)
    return TrSolve(c, iterations, scaling_method, scaling, normalize_quaternions, Delta0, Delta_max, eta, constrained_dofs, nu, rows_host, Delta_out);
  // the banded KKT step (kkt.h) is a sequence of launches with grid.y = problem like everything else of the loop: one
  // launch set per iteration for the whole batch.  The Schur-complement route (option con_kkt = 0, or nq + nu > 30) is
  // single-problem launches: a child context, stream and host thread per problem.
  if (c->con_kkt && SolverBlockSize(c->nq + nu, true) <= 30 && c->weights_diagonal) {
    if (int rc = idto_hip_eval_tau(c)) return rc;   // (the loop starts from the cost of the resident q; the other route evaluates it per problem)
    return TrSolve(c, iterations, scaling_method, scaling, normalize_quaternions, Delta0, Delta_max, eta, constrained_dofs, nu, rows_host, Delta_out);
  }
  if (!c->host_model) { g_err = "tr_solve_batch_constrained: the context keeps no host copy of its model"; return -1; }
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipStreamSynchronize(c->stream));
  c->children.resize((size_t)B, nullptr);
  for (int b = 0; b < B; ++b) {
    idto_hip_ctx* ch = c->children[b];
    if (!ch) {
      const int rc = idto_hip_create(&c->host_model->m, &c->host_problems[b]->p, &c->host_contact, c->device, &ch);
      if (rc) return rc;
      c->children[b] = ch;
    }
    // (at EVERY call, not only when the child is created - options set on the batch context afterwards, the
    // |h| column's degrees of freedom, a solver stepped down after a time-out must reach the children, or "rows bit for
    // bit those of idto_hip_tr_solve" breaks silently)
    ch->gradients_method = c->gradients_method; ch->fd_fast = c->fd_fast; ch->asm_fold = c->asm_fold;
    ch->solver_pipe = c->solver_pipe; ch->solver_nd = c->solver_nd; ch->two_sided = c->two_sided; ch->fused = c->fused;
    ch->reference_solver = c->reference_solver; ch->solver_band = c->solver_band; ch->asm_in_solver = c->asm_in_solver;
    ch->con_kkt = c->con_kkt;
    if (int rc = idto_hip_set_unactuated_dofs(ch, c->una_dofs_host.data(), c->una_nu)) return rc;
  }
  const size_t qbytes = (size_t)(c->N + 1) * c->nq * sizeof(double), row = (size_t)iterations * TRR_COUNT;
  std::vector<int> rcs((size_t)B, 0);
  std::vector<std::string> errs((size_t)B);
  auto work = [&](int b) {
    idto_hip_ctx* ch = c->children[b];
    auto fail = [&](int rc, const char* what) { rcs[b] = rc; errs[b] = g_err.empty() ? what : g_err; };
    if (hipSetDevice(c->device) != hipSuccess) return fail(-2, "hipSetDevice failed");
    ch->tr_conv_on = false;
    if (hipMemcpyAsync(ch->q, at_problem(c->q, (size_t)b * c->pstride), qbytes, hipMemcpyDeviceToDevice, ch->stream) != hipSuccess)
      return fail(-2, "copy of the problem's q failed");
    ch->fd_full = false; ch->partials_ahead = false; ch->con_ready = false; ch->con_begun = false; ch->trial_resident = false;
    int rc = idto_hip_eval_tau(ch);
    if (rc) return fail(rc, "eval_tau failed");
    rc = TrSolve(ch, iterations, scaling_method, scaling, normalize_quaternions, Delta0 + b, Delta_max, eta, constrained_dofs, nu,
                 rows_host + (size_t)b * row, Delta_out ? Delta_out + b : nullptr);
    if (rc) return fail(rc, "tr_solve failed");
    if (hipMemcpyAsync(at_problem(c->q, (size_t)b * c->pstride), ch->q, qbytes, hipMemcpyDeviceToDevice, ch->stream) != hipSuccess ||
        hipStreamSynchronize(ch->stream) != hipSuccess)
      return fail(-2, "copy of the iterate failed");
  };
  std::vector<std::thread> threads;
  for (int b = 1; b < B; ++b) threads.emplace_back(work, b);
  work(0);
  for (auto& t : threads) t.join();
  c->fd_full = false; c->partials_ahead = false; c->trial_resident = false; c->con_ready = false; c->con_begun = false;   // q of every problem moved
  for (int b = 0; b < B; ++b)
    if (rcs[b]) { g_err = "problem " + std::to_string(b) + ": " + errs[b]; return rcs[b]; }
  return 0;
}

#define NCCL_OK(expr)                                                                 \
  do {                                                                                \
    ncclResult_t r_ = (expr);                                                         \
    if (r_ != ncclSuccess) {                                                          \
      g_err = std::string(#expr) + ": " + R->GetErrorString(r_);                      \
      return -4;                                                                      \
    }                                                                                 \
  } while (0)

// the RCCL that got loaded must have the major version of the headers this library was compiled with
static int CommVersionOk() {
  RCCL_OR_FAIL(R);
  int v = 0;
  NCCL_OK(R->GetVersion(&v));
  const int major_rt = v / 10000, major_ct = NCCL_VERSION_CODE / 10000;
  if (major_rt != major_ct) {
    g_err = "librccl version mismatch: compiled against " + std::to_string(NCCL_VERSION_CODE) + ", loaded " + std::to_string(v);
    return -1;
  }
  return 0;
}
static int CommAttach(idto_hip_ctx* c, ncclComm_t comm, int rank, int world) {
  if (c->batch != 1) { g_err = "the sharded iteration serves single-problem contexts"; return -1; }
  if (world < 1 || world > IDTO_SLAB_PAD || rank < 0 || rank >= world) { g_err = "bad rank / world size"; return -1; }
  c->comm = comm; c->comm_rank = rank; c->comm_world = world;
  c->comm_per = (c->N + world - 1) / world;
  const int lo = std::min(c->N, rank * c->comm_per);
  return idto_hip_set_shard(c, lo, std::min(c->N, lo + c->comm_per));
}

int idto_hip_comm_unique_id(char* id_out, int bytes) {
  if (int rc = CommVersionOk()) return rc;
  if (!id_out || bytes < (int)sizeof(ncclUniqueId)) { g_err = "comm_unique_id: buffer of at least 128 bytes required"; return -1; }
  RCCL_OR_FAIL(R);
  ncclUniqueId id;
  NCCL_OK(R->GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return 0;
}

int idto_hip_comm_init(idto_hip_ctx* c, const char* unique_id, int rank, int world) {
  HIP_OK(hipSetDevice(c->device));
  if (c->comm) { g_err = "comm_init: the context already belongs to a communicator"; return -1; }
  if (int rc = CommVersionOk()) return rc;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  ncclComm_t comm = nullptr;
  RCCL_OR_FAIL(R);
  NCCL_OK(R->CommInitRank(&comm, world, id, rank));
  return CommAttach(c, comm, rank, world);
}

int idto_hip_comm_init_all(idto_hip_ctx** ctxs, int n) {
  if (!ctxs || n < 1 || n > IDTO_SLAB_PAD) { g_err = "comm_init_all: bad arguments"; return -1; }
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (ctxs[i]->comm) { g_err = "comm_init_all: a context already belongs to a communicator"; return -1; }
    devs[i] = ctxs[i]->device;
  }
  if (int rc = CommVersionOk()) return rc;
  std::vector<ncclComm_t> comms(n, nullptr);
  RCCL_OR_FAIL(R);
  NCCL_OK(R->CommInitAll(comms.data(), n, devs.data()));
  for (int i = 0; i < n; ++i)
    if (int rc = CommAttach(ctxs[i], comms[i], i, n)) return rc;
  return 0;
}

int idto_hip_comm_destroy(idto_hip_ctx* c) {
  if (c->comm) {
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    RCCL_OR_FAIL(R);
    NCCL_OK(R->CommDestroy(c->comm));
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
    return idto_hip_set_shard(c, 0, c->N);
  }
  return 0;
}

// in place: rank r's records start at r * per * stride of the receive buffer, which is where its
// fd_kernel wrote them
static int AllGatherSlab(idto_hip_ctx* c) {
  const size_t count = (size_t)c->comm_per * c->slab_stride;
  RCCL_OR_FAIL(R);
  NCCL_OK(R->AllGather(c->slab + (size_t)c->comm_rank * count, c->slab, count, ncclDouble, c->comm, c->stream));
  return 0;
}

int idto_hip_allgather_slab(idto_hip_ctx* c) {
  HIP_OK(hipSetDevice(c->device));
  if (!c->comm) { g_err = "allgather_slab: call idto_hip_comm_init first"; return -1; }
  DropPrefetch(c, {IDTO_ARR_SLAB});
  return AllGatherSlab(c);
}

int idto_hip_gn_step_sharded(idto_hip_ctx* c) {
  if (!c->comm) { g_err = "gn_step_sharded: call idto_hip_comm_init first"; return -1; }
  int rc = idto_hip_eval_partials(c);   // this rank's k-range
  if (rc) return rc;
  rc = idto_hip_allgather_slab(c);      // RCCL, on the context's stream
  if (rc) return rc;
  rc = idto_hip_grad_hess(c);           // every rank assembles and solves redundantly: identical bits, no second collective
  if (rc) return rc;
  return idto_hip_factor_solve(c, nullptr, 1, nullptr);
}

int idto_hip_eval_partials_multi(idto_hip_ctx** ctxs, int n) {
  // one process driving n devices (communicator from idto_hip_comm_init_all): every device's
  // finite-difference kernel covers its k-range, the all-gathers of all ranks go into one group
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]->comm) { g_err = "eval_partials_multi: call idto_hip_comm_init_all first"; return -1; }
    if (int rc = idto_hip_eval_partials(ctxs[i])) return rc;
  }
  RCCL_OR_FAIL(R);
  NCCL_OK(R->GroupStart());
  for (int i = 0; i < n; ++i) {
    HIP_OK(hipSetDevice(ctxs[i]->device));
    DropPrefetch(ctxs[i], {IDTO_ARR_SLAB});
    if (int rc = AllGatherSlab(ctxs[i])) { (void)R->GroupEnd(); return rc; }
  }
  NCCL_OK(R->GroupEnd());
  return 0;
}

int idto_hip_gn_step_multi(idto_hip_ctx** ctxs, int n) {
  if (int rc = idto_hip_eval_partials_multi(ctxs, n)) return rc;
  for (int i = 0; i < n; ++i) {
    if (int rc = idto_hip_grad_hess(ctxs[i])) return rc;
    if (int rc = idto_hip_factor_solve(ctxs[i], nullptr, 1, nullptr)) return rc;
  }
  return 0;
}

// Which RCCL the process resolved (RcclState: a copy the process had mapped already - torch's bundled one when torch was
// imported first - else the loader's).  Any is fine as long as the major version is the one of the headers this library
// was compiled against: checked where a communicator is created (CommVersionOk), reported here for the bench line.
int idto_hip_rccl_info(char* path_out, int path_cap, int* version_out) {
  RCCL_OR_FAIL(R);
  int v = 0;
  NCCL_OK(R->GetVersion(&v));
  if (version_out) *version_out = v;
  if (path_out && path_cap > 0) std::snprintf(path_out, (size_t)path_cap, "%s", R->path.c_str());
  return 0;
}

int idto_hip_get_option(idto_hip_ctx* c, const char* name, int* value) {
  if (std::strcmp(name, "rccl_version") == 0) { return idto_hip_rccl_info(nullptr, 0, value); }
  if (std::strcmp(name, "last_solver") == 0) { *value = c->last_solver; return 0; }
  if (std::strcmp(name, "weights_diagonal") == 0) { *value = c->weights_diagonal ? 1 : 0; return 0; }
  if (std::strcmp(name, "solver_nd") == 0) { *value = c->solver_nd; return 0; }
  if (std::strcmp(name, "solver_pipe") == 0) { *value = c->solver_pipe; return 0; }
  if (std::strcmp(name, "solver_band") == 0) { *value = c->solver_band; return 0; }
  if (std::strcmp(name, "gn_small") == 0) { *value = c->gn_small; return 0; }
  if (std::strcmp(name, "asm_in_solver") == 0) { *value = c->asm_in_solver; return 0; }
  if (std::strcmp(name, "solver_timeouts") == 0) { *value = c->solver_timeouts; return 0; }
  if (std::strcmp(name, "asm_fold") == 0) { *value = c->asm_fold; return 0; }
  if (std::strcmp(name, "con_kkt") == 0) { *value = c->con_kkt; return 0; }
  if (std::strcmp(name, "kkt_fold") == 0) { *value = c->kkt_fold; return 0; }
  if (std::strcmp(name, "tr_small") == 0) { *value = c->tr_small; return 0; }
  if (std::strcmp(name, "tr_fold") == 0) { *value = c->tr_fold; return 0; }
  if (std::strcmp(name, "kkt_in_asm") == 0) { *value = c->kkt_in_asm; return 0; }
  if (std::strcmp(name, "decide_in_solver") == 0) { *value = c->decide_in_solver; return 0; }
  if (std::strcmp(name, "tr_resident_ok") == 0) { *value = c->tr_resident_ok; return 0; }
  if (std::strcmp(name, "kkt_last_solver") == 0) { *value = c->kkt ? c->kkt->last_solver : 0; return 0; }
  if (std::strcmp(name, "last_assembly") == 0) { *value = c->last_assembly; return 0; }
  if (std::strcmp(name, "fused") == 0) { *value = c->fused; return 0; }
  if (std::strcmp(name, "nd_min_rows") == 0) { *value = c->nd_min_rows; return 0; }
  if (std::strcmp(name, "nd_recursion") == 0) { *value = c->nd_recursion; return 0; }
  if (std::strcmp(name, "async_uploads") == 0) { *value = c->async_uploads ? 1 : 0; return 0; }
  if (std::strcmp(name, "two_sided") == 0) { *value = c->two_sided; return 0; }
  if (std::strcmp(name, "reference_solver") == 0) { *value = c->reference_solver; return 0; }
  if (std::strcmp(name, "gradients_method") == 0) { *value = c->gradients_method; return 0; }
  if (std::strcmp(name, "fast_shape") == 0) { *value = c->M.fast_shape; return 0; }   // id_fast.h: 0 none, 1 acrobot, 2 hopper, 3 mini_cheetah, 4 allegro_hand, 5 spinner
  if (std::strcmp(name, "fd_fast") == 0) { *value = c->fd_fast ? 1 : 0; return 0; }
  g_err = std::string("unknown option ") + name;
  return -1;
}

int idto_hip_set_option(idto_hip_ctx* c, const char* name, int value) {
  if (std::strcmp(name, "reference_solver") == 0) { c->reference_solver = value != 0; return 0; }
  // (2: the stamps of the KKT context's solver - the constrained step's banded solve - are what IDTO_ARR_DEBUG returns)
  if (std::strcmp(name, "solver_debug") == 0) { c->solver_debug = value == 1; c->kkt_debug = value == 2; if (c->kkt) c->kkt->solver_debug = c->kkt_debug; return 0; }
  if (std::strcmp(name, "two_sided") == 0) { c->two_sided = value != 0; return 0; }
  if (std::strcmp(name, "fused") == 0) { c->fused = value != 0; return 0; }
  if (std::strcmp(name, "async_uploads") == 0) { c->async_uploads = value != 0; return 0; }
  if (std::strcmp(name, "nd_min_rows") == 0) { c->nd_min_rows = std::max(16, value); /* (below 16 block rows nd_split leaves a joiner chains of three rows with one row of their own: untested, not offered) */ if (c->kkt) c->kkt->nd_min_rows = c->nd_min_rows; return 0; }
  if (std::strcmp(name, "nd_recursion") == 0) { c->nd_recursion = value != 0; if (c->kkt) c->kkt->nd_recursion = c->nd_recursion; return 0; }
  if (std::strcmp(name, "solver_nd") == 0) { c->solver_nd = value != 0; return 0; }
  if (std::strcmp(name, "solver_pipe") == 0) { c->solver_pipe = value != 0; return 0; }
  if (std::strcmp(name, "solver_band") == 0) { c->solver_band = value; if (c->kkt) c->kkt->solver_band = value; return 0; }
  if (std::strcmp(name, "gn_small") == 0) { c->gn_small = value != 0; return 0; }
  if (std::strcmp(name, "asm_in_solver") == 0) { c->asm_in_solver = value != 0; return 0; }
  if (std::strcmp(name, "debug_skip_role") == 0) { c->debug_skip_role = value; if (c->kkt) c->kkt->debug_skip_role = value; return 0; }   // test aid
  if (std::strcmp(name, "debug_pipe_tail") == 0) { c->debug_pipe_tail = value; return 0; }   // measurement aid
  if (std::strcmp(name, "asm_fold") == 0) { c->asm_fold = value != 0; c->terms_valid = false; return 0; }
  if (std::strcmp(name, "con_kkt") == 0) { c->con_kkt = value != 0; return 0; }
  if (std::strcmp(name, "kkt_fold") == 0) { c->kkt_fold = value != 0; return 0; }
  if (std::strcmp(name, "tr_small") == 0) { c->tr_small = value != 0; return 0; }
  if (std::strcmp(name, "tr_fold") == 0) { c->tr_fold = value != 0; return 0; }
  if (std::strcmp(name, "kkt_in_asm") == 0) { c->kkt_in_asm = value != 0; return 0; }
  if (std::strcmp(name, "decide_in_solver") == 0) { c->decide_in_solver = value != 0; return 0; }
  if (std::strcmp(name, "fused_debug") == 0) { c->fused_debug = value != 0; return 0; }
  if (std::strcmp(name, "asm_stop") == 0) { c->asm_stop = value; return 0; }  // profiling aid
  if (std::strcmp(name, "fd_stop") == 0) { c->fd_stop = value; return 0; }    // profiling aid
  if (std::strcmp(name, "fd_fast") == 0) { c->fd_fast = value != 0; return 0; }
  if (std::strcmp(name, "gradients_method") == 0) {
    if (value < 0 || value > 2) { g_err = "gradients_method: 0 forward, 1 central, 2 central4 (autodiff needs Drake)"; return -1; }
    c->gradients_method = value;
    return 0;
  }
  g_err = std::string("unknown option ") + name;
  return -1;
}

// Does idto_hip_gn_step's solve take the pipelined kernel with the assembly inside (the conditions of LaunchAssemble's
// products path and of LaunchNd's pipelined branch, evaluated as they will be once the bands count as assembled)?
static bool AsmInSolver(idto_hip_ctx* c) {
  if (!(c->asm_in_solver && c->solver_pipe && !c->reference_solver)) return false;
  if (!(c->weights_diagonal && c->terms_valid && c->fd_full && c->asm_stop == 0)) return false;
  const bool was = c->h_assembled;
  c->h_assembled = true;   // (the plan's first row depends on it)
  LdlPlan p;
  const bool ok = PlanLdl(c, false, &p) == 0 && (BandEligible(c, p) || (p.K <= 20 && NdEligible(c, p) && 5 * c->batch <= 64));
  c->h_assembled = was;
  return ok;
}

// (AsmInSolver holds) is the launch the pipelined chains' - the kernel that can also decide on the trial point - and one
// of its DEC instantiations?
static bool DecideInSolver(idto_hip_ctx* c) {
  const bool was = c->h_assembled;
  c->h_assembled = true;
  LdlPlan p;
  const bool ok = PlanLdl(c, false, &p) == 0 && !BandEligible(c, p) && NdEligible(c, p) && c->solver_pipe && (p.K == 5 || p.K == 19) &&
                  c->cost_lds <= 160 * 1024;
  c->h_assembled = was;
  return ok;
}

int idto_hip_gn_step(idto_hip_ctx* c) {
  HIP_OK(hipSetDevice(c->device));
  if (c->spec_ready) return 0;   // already enqueued for this q by idto_hip_tr_trial (speculation)
  c->spec_pending = false;
  if (FusedEligible(c)) return LaunchFused(c);
  if (SmallEligible(c)) return LaunchSmall(c);
  int rc = idto_hip_eval_partials(c);
  if (rc) return rc;
  if (AsmInSolver(c)) {
    // two launches: the pipelined solver's grid carries the assembly (LaunchNd).  LaunchAssemble's bookkeeping:
    DropPrefetch(c, {IDTO_ARR_GRADIENT, IDTO_ARR_H_A, IDTO_ARR_H_B, IDTO_ARR_H_C, IDTO_ARR_HBANDS});
    if (!c->h_assembled) {
      HIP_OK(hipMemset2DAsync(c->step, c->pstride, 0, (size_t)c->nq * sizeof(double), (size_t)c->batch, c->stream));
      c->h_assembled = true;
    }
    c->con_ready = false; c->con_begun = false;
    c->fuse_asm_next = true;
    rc = idto_hip_factor_solve(c, nullptr, 1, nullptr);
    if (c->fuse_asm_next) {   // (nobody took it: the solve went another way than AsmInSolver foresaw)
      c->fuse_asm_next = false;
      if (!rc) { g_err = "gn_step: the solver that was to assemble g and H did not run"; rc = -1; }
    }
    return rc;
  }
  rc = idto_hip_grad_hess(c);
  if (rc) return rc;
  return idto_hip_factor_solve(c, nullptr, 1, nullptr);
}

int idto_hip_timing_enable(idto_hip_ctx* c, int enable) {
  c->timing = enable != 0;
  c->timing_stride = enable > 1 ? enable : 1;  // enable = s > 1: sample every s-th launch
  for (unsigned& t : c->timing_tick) t = 0;
  return 0;
}
int idto_hip_timing_reset(idto_hip_ctx* c) {
  if (TimeDrain(c)) return -2;
  for (int i = 0; i < 4; ++i) { c->tsum[i] = 0; c->tcnt[i] = 0; }
  return 0;
}
int idto_hip_timing_get(idto_hip_ctx* c, int which, double* avg_ms, int* launches) {
  if (which < 0 || which > 3) { g_err = "bad kernel index"; return -1; }
  if (TimeDrain(c)) return -2;
  *launches = c->tcnt[which];
  *avg_ms = c->tcnt[which] ? c->tsum[which] / c->tcnt[which] : 0.0;
  return 0;
}

int idto_hip_sync(idto_hip_ctx* c) {
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipStreamSynchronize(c->stream));
  return 0;
}

long idto_hip_array_size(idto_hip_ctx* c, int what) {
  const long nq = c->nq, nv = c->nv, N = c->N, bsz = nv * nq, qq = nq * nq;
  switch (what) {
    case IDTO_ARR_Q: return (N + 1) * nq;
    case IDTO_ARR_V: return (N + 1) * nv;
    case IDTO_ARR_A: case IDTO_ARR_TAU: return N * nv;
    case IDTO_ARR_NPLUS: return (N + 1) * bsz;
    case IDTO_ARR_DTAU_DQM: case IDTO_ARR_DTAU_DQT: case IDTO_ARR_DTAU_DQP: return N * bsz;
    case IDTO_ARR_GRADIENT: case IDTO_ARR_STEP: return (N + 1) * nq;
    case IDTO_ARR_H_A: case IDTO_ARR_H_B: case IDTO_ARR_H_C: return (N + 1) * qq;
    case IDTO_ARR_COST: return 1;
    case IDTO_ARR_SLAB: return N * (long)c->slab_stride;
    case IDTO_ARR_HBANDS: return 3 * (N + 6) * qq;
    case 15: return (N + 4) * 8 * 32;
    case IDTO_ARR_TR_DQ: case IDTO_ARR_TR_W: case IDTO_ARR_TR_SCALE: return (N + 1) * nq;
    case IDTO_ARR_ASM_TERMS: return N * (long)asm_terms_stride((int)nq);
    case IDTO_ARR_CON_S: return c->con_S ? (long)c->con_neq * c->con_neq + c->con_neq : -1;
    case IDTO_ARR_CON_LAMBDA: return (c->con_lambda_at && c->con_neq > 0) ? (long)c->con_neq : -1;
    default: return -1;
  }
}
int idto_hip_slab_stride(idto_hip_ctx* c) { return c->slab_stride; }

}  // extern "C"
namespace {
void* DevPtr(idto_hip_ctx* c, int what) {
  switch (what) {
    case IDTO_ARR_Q: return c->q;
    case IDTO_ARR_V: return c->v;
    case IDTO_ARR_A: return c->a;
    case IDTO_ARR_NPLUS: return c->nplus;
    case IDTO_ARR_GRADIENT: return c->g;
    case IDTO_ARR_H_A: return c->HA;
    case IDTO_ARR_H_B: return c->HB;
    case IDTO_ARR_H_C: return c->HC;
    case IDTO_ARR_STEP: return c->step;
    case IDTO_ARR_COST: return c->cost;
    case IDTO_ARR_SLAB: return c->slab;
    case IDTO_ARR_HBANDS: return c->HA;
    case 15: return (c->kkt_debug && c->kkt) ? c->kkt->dbg : c->dbg;
    case IDTO_ARR_TR_DQ: return c->tr_dq;
    case IDTO_ARR_TR_W: return c->tr_w;
    case IDTO_ARR_TR_SCALE: return c->tr_D;
    case IDTO_ARR_ASM_TERMS: return c->terms;
    case IDTO_ARR_CON_S: return c->con_S;
    case IDTO_ARR_CON_LAMBDA: return const_cast<double*>(c->con_lambda_at);
    default: return nullptr;  // tau and the three partials live strided inside the slab
  }
}
}  // namespace
extern "C" {

void* idto_hip_device_ptr(idto_hip_ctx* c, int what) {
  // a caller holding a pointer to the Hessian bands may overwrite them: from here on the solver
  // treats H as a general symmetric block penta-diagonal matrix (no identity block row 0 assumed)
  // until idto_hip_grad_hess assembles it again
  if (what == IDTO_ARR_H_A || what == IDTO_ARR_H_B || what == IDTO_ARR_H_C || what == IDTO_ARR_HBANDS)
    c->h_assembled = false;
  if (what == IDTO_ARR_SLAB) c->terms_valid = false;   // (records written from outside: assemble from the slab)
  return DevPtr(c, what);
}

int idto_hip_prefetch(idto_hip_ctx* c, int what) {
  HIP_OK(hipSetDevice(c->device));
  const long count = idto_hip_array_size(c, what);
  void* p = DevPtr(c, what);
  if (count < 0 || !p) { g_err = "prefetch: array is not contiguous in device memory"; return -1; }
  if (!c->side) HIP_OK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  // slot: the one already holding `what`, else a free one
  idto_hip_ctx::Prefetch* slot = nullptr;
  for (auto& pf : c->pre) if (pf.what == what) slot = &pf;
  if (!slot) for (auto& pf : c->pre) if (pf.what < 0 && !slot) slot = &pf;
  if (!slot) { g_err = "prefetch: at most 4 different arrays"; return -1; }
  if (slot->what != what) {  // lay the staging area out again (rare: first use of an array)
    HIP_OK(hipStreamSynchronize(c->side));
    slot->what = what; slot->count = (size_t)count;
    size_t total = 0;
    for (auto& pf : c->pre) if (pf.what >= 0) { pf.off = total; total += pf.count; pf.pending = false; }
    if (total > c->pre_cap) {
      if (c->pre_pin) (void)hipHostFree(c->pre_pin);
      c->pre_pin = nullptr;
      HIP_OK(hipHostMalloc((void**)&c->pre_pin, total * sizeof(double), hipHostMallocDefault));
      c->pre_cap = total;
    }
    if (!slot->ev) HIP_OK(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
  }
  HIP_OK(hipEventRecord(slot->ev, c->stream));
  HIP_OK(hipStreamWaitEvent(c->side, slot->ev, 0));
  HIP_OK(hipMemcpyAsync(c->pre_pin + slot->off, p, slot->count * sizeof(double), hipMemcpyDeviceToHost, c->side));
  slot->pending = true;
  return 0;
}

int idto_hip_get(idto_hip_ctx* c, int what, double* out) {
  HIP_OK(hipSetDevice(c->device));
  for (auto& pf : c->pre)
    if (pf.what == what && pf.pending) {  // the copy was enqueued earlier: wait for it alone
      HIP_OK(hipStreamSynchronize(c->side));
      std::memcpy(out, c->pre_pin + pf.off, pf.count * sizeof(double));
      pf.pending = false;
      return (what == IDTO_ARR_STEP) ? FactorStatus(c) : 0;
    }
  return idto_hip_get_batch(c, what, 0, out);
}

int idto_hip_get_batch(idto_hip_ctx* c, int what, int pb, double* out) {
  HIP_OK(hipSetDevice(c->device));
  if (pb < 0 || pb >= c->batch) { g_err = "problem index outside the batch"; return -1; }
  HIP_OK(hipStreamSynchronize(c->stream));
  const long count = idto_hip_array_size(c, what);
  if (count < 0) { g_err = "unknown array id"; return -1; }
  const size_t bsz = (size_t)c->nv * c->nq, po = (size_t)pb * c->pstride;
  if (what == IDTO_ARR_TAU || (what >= IDTO_ARR_DTAU_DQM && what <= IDTO_ARR_DTAU_DQP)) {
    const size_t off = what == IDTO_ARR_TAU ? 3 * bsz : (size_t)(what - IDTO_ARR_DTAU_DQM) * bsz;
    const size_t width = what == IDTO_ARR_TAU ? (size_t)c->nv : bsz;
    HIP_OK(hipMemcpy2D(out, width * sizeof(double), at_problem(c->slab, po) + off, (size_t)c->slab_stride * sizeof(double),
                       width * sizeof(double), c->N, hipMemcpyDeviceToHost));
    return 0;
  }
  void* p = DevPtr(c, what);
  HIP_OK(hipMemcpy(out, at_problem(static_cast<char*>(p), po), (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
  if (what != IDTO_ARR_STEP) return 0;
  int fs = FactorStatus(c, pb);
  // a launch whose waits between workgroups ran out: FactorStatus has stepped the context down, the Gauss-Newton
  // step is computed again (at most once per variant that can time out)
  for (int attempt = 0; fs == IDTO_HIP_SOLVER_TIMEOUT && attempt < 4 && c->last_step_kind != 0; ++attempt) {
    const int rc = (c->last_step_kind == 2) ? idto_hip_gn_step(c) : idto_hip_factor_solve(c, nullptr, 1, nullptr);
    if (rc) return rc;
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(out, at_problem(static_cast<char*>(p), po), (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    fs = FactorStatus(c, pb);
  }
  return fs;
}

int idto_hip_get_many(idto_hip_ctx* c, int n, const int* what, double* const* out) {
  HIP_OK(hipSetDevice(c->device));
  if (n < 0 || (n > 0 && (!what || !out))) { g_err = "get_many: bad arguments"; return -1; }
  if (c->batch != 1) { g_err = "get_many serves single-problem contexts"; return -1; }
  std::vector<size_t> off((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) {
    const long count = idto_hip_array_size(c, what[i]);
    if (count < 0 || what[i] == IDTO_ARR_STEP) { g_err = "get_many: unknown array id (or IDTO_ARR_STEP: use idto_hip_get)"; return -1; }
    off[i + 1] = off[i] + (size_t)count;
  }
  if (off[n] > c->many_cap) {
    if (c->many_pin) (void)hipHostFree(c->many_pin);
    c->many_pin = nullptr; c->many_cap = 0;
    HIP_OK(hipHostMalloc((void**)&c->many_pin, std::max<size_t>(off[n], 1) * sizeof(double), hipHostMallocDefault));
    c->many_cap = off[n];
  }
  const size_t bsz = (size_t)c->nv * c->nq;
  for (int i = 0; i < n; ++i) {
    const int w = what[i];
    for (auto& pf : c->pre) if (pf.what == w) pf.pending = false;   // (a staged copy of it is superseded)
    double* dst = c->many_pin + off[i];
    if (w == IDTO_ARR_TAU || (w >= IDTO_ARR_DTAU_DQM && w <= IDTO_ARR_DTAU_DQP)) {   // rows of the slab's records
      const size_t o = w == IDTO_ARR_TAU ? 3 * bsz : (size_t)(w - IDTO_ARR_DTAU_DQM) * bsz;
      const size_t width = w == IDTO_ARR_TAU ? (size_t)c->nv : bsz;
      HIP_OK(hipMemcpy2DAsync(dst, width * sizeof(double), c->slab + o, (size_t)c->slab_stride * sizeof(double), width * sizeof(double),
                              c->N, hipMemcpyDeviceToHost, c->stream));
    } else {
      HIP_OK(hipMemcpyAsync(dst, DevPtr(c, w), (off[i + 1] - off[i]) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
  }
  TRACE("hip: get_many: copies enqueued");
  HIP_OK(hipStreamSynchronize(c->stream));
  TRACE("hip: get_many: waited");
  for (int i = 0; i < n; ++i) std::memcpy(out[i], c->many_pin + off[i], (off[i + 1] - off[i]) * sizeof(double));
  return 0;
}

int idto_hip_solver_status(idto_hip_ctx* c, int* failed, int* failed_rows_total) {
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipStreamSynchronize(c->stream));
  const volatile unsigned* st = c->status_pin;
  int any = 0, rows = 0;
  for (int b = 0; b < c->batch; ++b) {
    any |= (c->fact_id != 0 && st[2 * b] == c->fact_id) ? 1 : 0;
    rows += (int)st[2 * b + 1];
  }
  if (failed) *failed = any;
  if (failed_rows_total) *failed_rows_total = rows;
  return 0;
}

int idto_hip_solver_status_batch(idto_hip_ctx* c, int* failed) {
  HIP_OK(hipSetDevice(c->device));
  HIP_OK(hipStreamSynchronize(c->stream));
  const volatile unsigned* st = c->status_pin;
  for (int b = 0; b < c->batch; ++b) failed[b] = (c->fact_id != 0 && st[2 * b] == c->fact_id) ? 1 : 0;
  return 0;
}

int idto_hip_gn_step_batch(idto_hip_ctx* c) { return idto_hip_gn_step(c); }

int idto_hip_math_probe(int device, const double* x, int n, double* sq, double* rc, double* sn, double* cs, double* ex,
                        double* lg) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_err = "no HIP device available"; return -3; }
  HIP_OK(hipSetDevice(device));
  double* d[7];
  for (int i = 0; i < 7; ++i) HIP_OK(hipMalloc((void**)&d[i], (size_t)n * sizeof(double)));
  HIP_OK(hipMemcpy(d[0], x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], n, d[1], d[2], d[3], d[4], d[5], d[6]);
  HIP_OK(hipGetLastError());
  HIP_OK(hipDeviceSynchronize());
  double* outs[6] = {sq, rc, sn, cs, ex, lg};
  for (int i = 0; i < 6; ++i) HIP_OK(hipMemcpy(outs[i], d[i + 1], (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  for (int i = 0; i < 7; ++i) (void)hipFree(d[i]);
  return 0;
}

}  // extern "C"
