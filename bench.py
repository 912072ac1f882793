#!/usr/bin/env python3
"""bench.py — Gauss-Newton iterations/second (grad + Hessian + solve), mini_cheetah N=40.

One "step" = one Gauss-Newton iteration of IDTO's hot path on a resident trajectory q:
N+, v, a, tau -> finite-difference dtau/dq -> gradient + Hessian bands -> block-Thomas
factor + solve H p = -g   (reference optimizer/trajectory_optimizer.cc:
426-563, 962-973, 1021-1165, optimizer/penta_diagonal_solver.h:124-248), i.e.
`idto_hip_gn_step` of the C-ABI.  fp64, synthetic trajectory,
inputs resident in HBM before the timed region.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode shard|replicas]
For N > 1 launch with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`:
  shard (default) ONE problem (BASELINE config 4): the (k, column) perturbation grid is split into
           contiguous k-ranges, one per rank; one all-gather (RCCL) of the dtau/dq slab, then
           every rank assembles and solves redundantly; value = that problem's iterations/s
           ("scaling": "strong").  One MI355X already runs all N=40 fd blocks concurrently (40
           of 256 CUs), so this mode cannot beat one GPU at this size: DESIGN.md §7.  The
           replicas rate of the same ranks is reported as the extra "replicas_mode".
  replicas every rank iterates its own problem (same model and horizon, trajectory seed = rank;
           BASELINE config 5 style: `--mode replicas --config allegro_hand --num-steps 60`) with
           no data-path collective; value = total iterations/s over all ranks ("weak").
The timed region of `value` contains no events and no host synchronisation; per-kernel
durations (HIP events) and the per-step latency distribution (one synchronisation per step) are
measured in separate passes afterwards.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from idto_amd import hip  # noqa: E402
from idto_amd.model import load_model  # noqa: E402
from idto_amd.problem import load_config, make_problem, synthetic_trajectory  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
KERNELS = ["fd_kernel", "assemble_diag_kernel", "penta_ldl_kernel", "gn_fused_kernel"]


def algorithmic_bytes(N, nq, nv):
    """every intermediate written once and read once, split per kernel."""
    part, hband, fact = 3 * N * nv * nq, 3 * (N + 1) * nq * nq, 5 * (N + 1) * nq * nq
    fd = 8 * (part + (N + 1) * nq + (N + 1) * nv + 2 * N * nv)
    asm = 8 * (part + hband + (N + 1) * nq + (N + 1) * nv + N * nv + (N + 1) * nq)
    penta = 8 * (hband + 2 * fact + 2 * (N + 1) * nq)
    return [fd, asm, penta]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate
    FETCH_SIZE and WRITE_SIZE runs of this same command, tools/gpu.sh check ->
    tools/pmc_summarize.py): FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md, WRITE_SIZE raw.  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    e = json.load(open(files[-1])).get(kernel)
    if not e or "fetch_bytes_per_launch_x2" not in e or "write_bytes_per_launch_raw" not in e:
        return None, None
    return e["fetch_bytes_per_launch_x2"] + e["write_bytes_per_launch_raw"], os.path.relpath(files[-1], ROOT)


def latency_model(kern_ms, names):
    """The bounds that apply to these kernels (they are latency / issue bound: the HBM fraction is ~0.3 %): the model of
    the newest profiles/r*_latency_model.json (tools/latency_model.py, from that round's committed PMC passes and in-kernel
    timelines) against the HIP-event kernel times of THIS run (which include ~2 us of event overhead per launch)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_latency_model.json")))
    if not files:
        return None
    m = json.load(open(files[-1]))
    out = {"source": os.path.relpath(files[-1], ROOT), "script": "tools/latency_model.py"}
    live = {names[i]: 1e3 * kern_ms[i][0] for i in range(len(names)) if i < len(kern_ms) and kern_ms[i][1] > 0}
    fd = m.get("fd_kernel")
    if fd and "fd_kernel" in live:
        out["fd_kernel"] = {"bound": fd["bound"], "valu_instructions_per_wavefront": fd["valu_instructions_per_wavefront"],
                            "issue_floor_us": fd["issue_floor_us"], "this_run_us": live["fd_kernel"],
                            "achieved_frac": fd["issue_floor_us"] / live["fd_kernel"],
                            "profiled_round": {k: fd[k] for k in ("rocprof_avg_us", "achieved_frac_of_issue_floor",
                                                                  "wait_frac_of_wave_cycles", "valu_active_frac_of_wave_cycles")}}
    so = m.get("penta_pipe_kernel")
    if so and "penta_pipe_kernel" in live:
        out["penta_pipe_kernel"] = {"bound": so["bound"], "row_model_us": so["row_model_us"], "pivot_chain_floor_us": so["pivot_chain_floor_us"],
                                    "this_run_us": live["penta_pipe_kernel"],
                                    "achieved_frac_of_pivot_chain_floor": so["pivot_chain_floor_us"] / live["penta_pipe_kernel"],
                                    "terms_us": {k: so[k] for k in ("rows_of_the_longest_chain", "row_to_row_us", "k_pivots_us",
                                                                    "hand_over_to_separator_us", "separator_us", "back_substitution_us")},
                                    "note": "this_run_us includes the assembly workgroups when the launch assembles g and H itself; "
                                            "the row model is of the solver alone"}
    return out


def host_cpu_limit():
    """CPUs this process may use: logical cores, affinity mask, CFS quota of the container (cgroup v2 cpu.max
    or v1 cpu.cfs_quota_us / cpu.cfs_period_us; None = unlimited)."""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return {"logical_cores": os.cpu_count() or 1, "affinity": len(os.sched_getaffinity(0)), "cfs_quota_cpus": quota}


# timed Gauss-Newton iterations per repeat of a cpu_baseline leg at N = 40 (scaled by 40 / N): FIXED, the same for every leg
# and every run - a shared time budget gave the 8-thread leg 291 iterations and the others 1,064 - 2,000, and the best-leg
# denominator of the headline ratio moved +-5 % between runs.  ~1.5 s of one thread per repeat.
CPU_BASELINE_ITERS = {"mini_cheetah": 300, "allegro_hand": 120, "hopper": 2500, "spinner": 6000, "acrobot": 8000}


def cpu_baseline(config, N, repeats=3):
    """The CPU oracle (a port of the reference algorithm, OpenMP where the reference has it: TO.cc:209,
    :476) timed on this box's host cores on a bounded sample of the same workload.  Every leg runs in a
    process of its own (tools/cpu_baseline.py) so that its OpenMP runtime starts with pinned threads
    (OMP_PROC_BIND=close, OMP_PLACES=cores) that spin between the parallel regions (OMP_WAIT_POLICY=
    active): this process already carries torch's OpenMP runtime and an environment that has been read.
    Every leg times the SAME fixed number of iterations `repeats` times; a leg's rate is the median, its spread
    (max - min) / median is reported, and the baseline is the best leg's median."""
    import subprocess
    cores = os.cpu_count() or 1
    limit = host_cpu_limit()
    usable = max(1, min(cores, limit["affinity"], int(limit["cfs_quota_cpus"]) if limit["cfs_quota_cpus"] else cores))
    # num_threads in {1, 4 (the reference's YAML default), 8, 16, all that the OpenMP loops over t can
    # use}: the reference parallelises over the N timesteps only, so "all" = min(N, cores); the best leg
    # is the baseline
    # ("cores" = what this process may actually use: its affinity mask and the container's CFS quota - spinning
    # threads beyond the quota are throttled and run slower than one thread)
    legs = sorted({1, min(4, usable), min(8, usable), min(16, usable), min(N, usable)})
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_WAIT_POLICY="active", OMP_DYNAMIC="false")
    env.pop("OMP_NUM_THREADS", None)
    rates, spreads, notes, parts1, best = {}, {}, [], None, None
    iters = max(20, int(CPU_BASELINE_ITERS.get(config, 300) * 40 / max(N, 1)))
    for nt in legs:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--config", config, "--num-steps", str(N),
               "--threads", str(nt), "--iters", str(iters), "--repeats", str(repeats)]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
            leg = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:   # a leg that cannot run is reported, never silently dropped
            notes.append(f"num_threads={nt}: failed ({type(e).__name__})")
            continue
        rates[str(nt)] = leg["iters_per_s"]
        spreads[str(nt)] = leg.get("spread")
        notes.append(f"num_threads={nt}: {leg['iters_per_s']:.1f} it/s (spread {100 * leg.get('spread', 0):.1f} %)")
        if nt == 1:
            parts1 = leg.get("parts_s")
        if best is None or leg["iters_per_s"] > best[0]:
            best = (leg["iters_per_s"], nt)
    if best is None:
        raise RuntimeError("cpu_baseline: no leg ran: " + "; ".join(notes))
    out = {"value": best[0], "unit": "GN iters/s", "cores": best[1], "kind": "port",
           "iters_per_s_by_num_threads": rates, "spread_by_num_threads": spreads,
           "iterations_per_repeat": iters, "repeats": repeats, "statistic": "median of the repeats; spread = (max - min) / median",
           "omp": "one process per leg, OMP_PROC_BIND=close OMP_PLACES=cores OMP_WAIT_POLICY=active",
           "host_cpu_limit": limit,
           "sample": f"same trajectory as the GPU run, {iters} Gauss-Newton iterations x {repeats} repeats per leg; " + "; ".join(notes) +
                     f" (host has {cores} logical cores, {usable} usable by this process; the OpenMP loops run over t, so at most N threads work)"}
    if parts1:
        # the reference's two parallel loops against what it runs serially: the ceiling of any thread count
        par = parts1["tau"] + parts1["derivatives"]
        ser = parts1["assembly"] + parts1["solve"]
        out["serial_fraction_at_1_thread"] = ser / (par + ser)
        out["amdahl_ceiling_iters_per_s"] = 1.0 / ser
        out["seconds_per_iter_at_1_thread"] = parts1
    return out


def full_iteration(cfg, model, N, device, with_cpu, iters=20):
    """Secondary number: one iteration of TrajectoryOptimizer::Solve with the example
    YAML's solver settings (scaling, equality constraints / multipliers, dogleg, trust ratio) through
    libidto_opt.so, next to the CPU oracle's iteration on the same problem."""
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
    sp.max_iterations, sp.verbose, sp.num_threads = iters, False, 1
    opt = TrajectoryOptimizer(model, prob, sp, device=device)
    for _ in range(2):  # the second solve is the warmed-up one
        sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
        opt.Solve(q_guess, sol, st)
    out = {"ms_per_iteration": 1e3 * st.solve_time / iters, "iterations": iters,
           "num_equality_constraints": opt.num_equality_constraints(),
           "what": "TrajectoryOptimizer::Solve, trust region, solver settings of the example YAML"}
    opt.close()
    if with_cpu:
        from oracle_lib import Oracle
        t0 = time.perf_counter()
        Oracle(model, prob, sp).solve(q_guess)
        out["cpu_port_ms_per_iteration"] = 1e3 * (time.perf_counter() - t0) / iters
    return out


def mpc_replan(cfg, model, device, replans=50):
    """Secondary number ("the MPC-relevant latency"): wall clock of one re-plan of the example's MPC loop
    (reference examples/mpc_controller.cc:43-85 UpdateAbstractState: shift the stored solution, SolveFromWarmStart with the
    YAML's mpc_iters, store the splines) through the C++ shell of include/idto/examples/mpc_controller.h - the example's own
    horizon (its YAML's num_steps), not the bench workload's.  tools/mpc_timeline.py accounts for it mark by mark."""
    from idto_amd.mpc import DeviceModelPredictiveController
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    from idto_amd.problem import SolverParameters
    prob, sp, q_guess = make_problem(cfg, model)
    sp.verbose = False
    sp.max_iterations = min(int(sp.max_iterations), 30)
    opt = TrajectoryOptimizer(model, prob, sp, device=device)
    sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
    opt.Solve(q_guess, sol, st)
    iters = int(cfg.get("mpc_iters", 1))
    period = 1.0 / float(cfg.get("controller_frequency", 200.0))
    opt1 = TrajectoryOptimizer(model, prob, SolverParameters(**{**sp.__dict__, "max_iterations": iters}), device=device)
    mpc = DeviceModelPredictiveController(opt1, sol, actuated=model.actuated, replan_period=period)
    x = np.concatenate([np.asarray(sol.q[0]), np.asarray(sol.v[0])])
    times = []
    for i in range(replans + 10):
        t = i * period
        if i:
            x = mpc.state(t)
        t0 = time.perf_counter()
        mpc.update(t, x[:model.nq], x[model.nq:], copy=False)   # (the controller's own output buffers, as a C++ caller holds them)
        times.append(time.perf_counter() - t0)
    ts = np.sort(np.array(times[10:])) * 1e3
    mpc.close(); opt1.close(); opt.close()
    return {"ms_per_replan_median": float(np.median(ts)), "p10": float(ts[len(ts) // 10]), "p90": float(ts[9 * len(ts) // 10]),
            "num_steps": int(prob.num_steps), "mpc_iters": iters, "controller_period_ms": 1e3 * period, "replans": replans,
            "what": "idto_mpc_update (C++ ModelPredictiveController on the device), the example YAML's horizon and mpc_iters"}


SHARD_NOTE = ("sharding ONE N=40 problem cannot beat one GPU: its 40 finite-difference workgroups already run side by side "
              "on 40 of the 256 CUs, the all-gather adds a hand-over, and the block solve (a dependent chain) does not "
              "shard - DESIGN.md §7; the aggregate of independent problems is reported as replicas_mode / config5_workload")


def self_launch_command(argv, gpus, visible_gpus, port=None):
    """`bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): the command and the extra
    environment that start it again as N ranks under torch.distributed.run on this node.  With fewer visible
    devices than ranks the ranks share devices (local_rank % visible) and measure independent replicas - RCCL does
    not form a communicator of two ranks on one device - and the JSON line says so."""
    import socket
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)
    env = {"MASTER_ADDR": "127.0.0.1", "IDTO_BENCH_SELF_LAUNCHED": "1"}
    if visible_gpus < gpus:
        env["IDTO_BENCH_VISIBLE_GPUS"] = str(max(1, visible_gpus))
    return cmd, env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["shard", "replicas"], default="shard")
    ap.add_argument("--config", default="mini_cheetah")
    ap.add_argument("--num-steps", type=int, default=40, help="horizon N")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-full", action="store_true", help="skip the informational full-iteration measurement")
    ap.add_argument("--batch", type=int, default=16,
                    help="also report the aggregate rate of this many independent problems on one GPU (0/1: skip)")
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=INT",
                    help="a context option of the C-ABI (idto_hip_set_option), e.g. asm_in_solver=0: profiling aid")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks here (one process per GPU over RCCL, the contract of the driver's own command)
        import subprocess
        cmd, extra = self_launch_command(sys.argv[1:], args.gpus, torch.cuda.device_count())
        raise SystemExit(subprocess.run(cmd, env=dict(os.environ, **extra)).returncode)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         "or without a launcher")
    visible = int(os.environ.get("IDTO_BENCH_VISIBLE_GPUS", "0"))   # set by the self-launch when ranks must share devices
    if not visible and world > 1 and 0 < torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        visible = torch.cuda.device_count()   # (a launcher started more ranks on this node than it has devices: same fallback)
    if visible:
        local_rank = local_rank % visible
        args.mode = "replicas"
    dist = None
    # Data path of the sharded mode: the RCCL communicator inside libidto_hip.so (ncclAllGather on
    # the context's stream, include/idto_hip.h idto_hip_comm_*); torch.distributed is the control
    # plane only (unique id, barrier, max over ranks) and runs on gloo.  IDTO_BENCH_EXCHANGE=torch
    # selects the round-1 path instead (torch.distributed "nccl" all_gather_into_tensor on a
    # zero-copy view of the slab).  IDTO_BENCH_SAME_GPU (never set by the driver) puts all ranks
    # on GPU 0 for single-GPU debugging.
    exchange = os.environ.get("IDTO_BENCH_EXCHANGE", "rccl")
    backend = os.environ.get("IDTO_BENCH_BACKEND", "gloo" if exchange == "rccl" else "nccl")
    if os.environ.get("IDTO_BENCH_SAME_GPU"):
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], device=("cuda" if backend == "nccl" else "cpu"), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    cfg = load_config(args.config)
    model = load_model(args.config)
    N = args.num_steps
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    sharded = world > 1 and args.mode == "shard"
    # replicas: every rank its own trajectory; shard / single GPU: the BASELINE trajectory (seed 0)
    q = synthetic_trajectory(cfg, model, N, seed=(0 if sharded else rank), lower=0.01)
    nq, nv = model.nq, model.nv

    dev = hip.HipPath(model, prob, sp, device=local_rank)
    stream = torch.cuda.current_stream()
    dev.set_stream(stream.cuda_stream)
    for kv in args.set:
        name, _, val = kv.partition("=")
        dev.set_option(name, int(val))
    dev.set_q(q)

    exch = None
    exchange_note = None
    if world > 1:
        from idto_amd.multi_gpu import RcclShard, SlabExchange, device_slab_view
        try:
            if visible:
                raise RuntimeError(f"{world} ranks on {visible} visible device(s): no RCCL communicator between ranks of one device")
            if exchange == "rccl":
                exch = RcclShard(dist, dev, rank, world)
                dev.set_shard(0, N)   # (the shard is switched on below, per mode)
            else:
                exch = SlabExchange(dist, device_slab_view(dev, N), N, dev.slab_stride, rank, world)
        except Exception as e:  # replicas need no exchange: never let the extra measurement cost the metric
            exch = None
            print(f"[bench] rank {rank}: slab exchange unavailable ({e})", file=sys.stderr)
        # every rank must take the same path from here on (collectives inside): agree over the control plane
        ok = torch.tensor([1.0 if exch is not None else 0.0], device=("cuda" if backend == "nccl" else "cpu"),
                          dtype=torch.float64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            if exch is not None and exchange == "rccl":
                try:
                    exch.close()
                except Exception:
                    pass
            exch = None
            if sharded:   # the sharded problem cannot be measured here: report independent replicas and say so
                sharded = False
                exchange_note = "the slab exchange could not be set up on every rank: value is replicas, not the sharded problem"
                q = synthetic_trajectory(cfg, model, N, seed=rank, lower=0.01)
                dev.set_shard(0, N)
                dev.set_q(q)

    def step_sharded():
        if exchange == "rccl":
            dev.gn_step_sharded()   # eval_partials (own k-range) + ncclAllGather + grad_hess + factor_solve
            return
        dev.eval_partials()
        exch.gather()
        dev.grad_hess()
        dev.factor_solve()

    def step():
        if sharded:
            step_sharded()
        else:
            dev.gn_step()

    if sharded:
        dev.set_shard(exch.lo, exch.hi)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # ---- the timed region: exactly `steps` steps, nothing else on the stream, no host sync
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # ---- separate pass 1: per-kernel durations, HIP events around every launch (on the context's
    # stream); never part of `value`, so `value` does not depend on --steps
    dev.timing_enable(1)
    dev.timing_reset()
    for _ in range(max(20, min(100, args.steps))):
        step()
    barrier()
    dev.timing_enable(False)
    kern = []
    for w in range(4):
        ms, n = dev.timing_get(w)
        kern.append((ms, n))
    fused_run = kern[3][1] > 0   # the iteration ran as one persistent launch (csrc/fused.h)
    # ---- separate pass 2: latency of ONE step (launch -> results complete), synchronised per step
    lat = []
    for _ in range(max(50, min(200, args.steps))):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t1))
    lat = np.sort(np.array(lat))
    latency = {"median": float(np.median(lat)), "p10": float(lat[int(0.1 * (len(lat) - 1))]),
               "p90": float(lat[int(0.9 * (len(lat) - 1))]), "samples": len(lat),
               "what": "one step with a host synchronisation before and after (ms); `value` is the "
                       "back-to-back rate of the timed region"}
    p = dev.get("step")
    g = dev.get("gradient")
    assert np.all(np.isfinite(p)) and np.all(np.isfinite(g))

    # ---- the other multi-rank mode on the same ranks, outside the timed region of `value`
    # (every rank takes the same branch: collectives inside)
    other_extra, other_key = None, None
    if world > 1 and exch is not None:
        other_key = "replicas_mode" if sharded else "shard_mode"
        try:
            if sharded:
                # N independent problems, one per rank, no collective on the data path
                dev.set_shard(0, N)
                dev.set_q(synthetic_trajectory(cfg, model, N, seed=rank, lower=0.01))
                fn, ns = dev.gn_step, max(10, args.steps // 4)
            else:
                dev.set_q(synthetic_trajectory(cfg, model, N, seed=0, lower=0.01))
                dev.gn_step()
                p_ref = dev.get("step")
                dev.set_shard(exch.lo, exch.hi)
                fn, ns = step_sharded, max(10, args.steps // 4)
            for _ in range(5):
                fn()
            barrier()
            t1 = time.perf_counter()
            for _ in range(ns):
                fn()
            barrier()
            el = max_over_ranks(time.perf_counter() - t1)
            if sharded:
                other_extra = {"value": world * ns / el, "unit": "GN iters/s (aggregate, one problem per rank)",
                               "steps": ns, "ms_per_step": 1e3 * el / ns, "scaling": "weak"}
            else:
                same = bool(np.array_equal(dev.get("step"), p_ref))   # sharded == single-GPU result, bit for bit
                other_extra = {"value": ns / el, "unit": "GN iters/s (one problem)", "steps": ns,
                               "ms_per_step": 1e3 * el / ns, "scaling": "strong",
                               "bit_identical_to_unsharded": same,
                               "exchange": f"{exchange} all-gather of the {N * dev.slab_stride * 8} B slab over {world} ranks"}
        except Exception as e:  # informational only: never lose the metric over it
            other_extra = {"error": str(e)[:200]}

    # ---- the sharded result against the same problem on one device (every rank: same bits expected everywhere)
    bit_identical = None
    if world > 1 and sharded and exch is not None:
        try:
            dev.set_shard(exch.lo, exch.hi)
            dev.set_q(q)
            step_sharded()
            p_sh = dev.get("step")
            dev.set_shard(0, N)
            dev.set_q(q)
            dev.gn_step()
            same = 1.0 if np.array_equal(dev.get("step"), p_sh) else 0.0
            tt = torch.tensor([same], device=("cuda" if backend == "nccl" else "cpu"), dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            bit_identical = bool(tt.item() == 1.0)
        except Exception as e:  # informational only
            bit_identical = f"not checked: {str(e)[:160]}"

    # ---- BASELINE config 5 on the same ranks (informational, outside the timed region of `value`): allegro_hand
    # + sphere, 60 steps, one warm-started problem per GPU, no collective on the data path
    config5 = None
    if world > 1:
        try:
            c5, m5 = load_config("allegro_hand"), load_model("allegro_hand")
            N5 = 60
            p5, s5, _ = make_problem(c5, m5, num_steps=N5)
            s5.scaling = False
            s5.equality_constraints = False
            d5 = hip.HipPath(m5, p5, s5, device=local_rank)
            d5.set_stream(stream.cuda_stream)
            q5 = synthetic_trajectory(c5, m5, N5, seed=rank, lower=0.01)
            d5.set_q(q5)
            d5.gn_step()
            q5 = q5 + 0.5 * d5.get("step").reshape(q5.shape)   # warm start: the guess after one damped Gauss-Newton step
            d5.set_q(q5)
            for _ in range(5):
                d5.gn_step()
            barrier()
            n5 = max(10, args.steps // 4)
            t1 = time.perf_counter()
            for _ in range(n5):
                d5.gn_step()
            barrier()
            el = max_over_ranks(time.perf_counter() - t1)
            ok5 = bool(np.all(np.isfinite(d5.get("step")))) and d5.solver_status() == (False, 0)
            config5 = {"workload": f"allegro_hand + sphere (nq={m5.nq}, nv={m5.nv}, {m5.npairs} contact pairs), horizon N={N5}, "
                                   f"one warm-started problem per GPU ({world} problems), no data-path collective",
                       "value": world * n5 / el, "unit": "GN iters/s (aggregate)", "steps": n5, "ms_per_step": 1e3 * el / n5,
                       "scaling": "weak", "finite_and_factorised_on_rank0": ok5}
            d5.close()
        except Exception as e:  # informational only: never lose the metric over it
            config5 = {"error": str(e)[:200]}

    batch_extra = None
    if world == 1 and args.batch > 1:
        # informational (never `value`): B independent problems resident on this ONE GPU and advanced by
        # ONE host thread with one launch per kernel (idto_hip_create_batch / idto_hip_gn_step_batch,
        # grid.y = problem) - one problem leaves most of the 256 CUs idle, a batched / sampling MPC
        # server does not
        batch_extra = []
        for B in sorted({args.batch, 4 * args.batch}):
            probs = []
            for b in range(B):
                pb, _, _ = make_problem(cfg, model, num_steps=N)
                probs.append(pb)
            bd = hip.HipPath(model, probs, sp, device=local_rank)
            bd.set_stream(stream.cuda_stream)
            bd.set_q_batch(np.array([synthetic_trajectory(cfg, model, N, seed=b, lower=0.01) for b in range(B)]))
            for _ in range(5):
                bd.gn_step()
            torch.cuda.synchronize()
            nb = max(20, args.steps // 4)
            t1 = time.perf_counter()
            for _ in range(nb):
                bd.gn_step()
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            assert bd.solver_status_batch() == [False] * B
            batch_extra.append({"problems": B, "value": B * nb / el, "unit": "GN iters/s (aggregate)",
                                "ms_per_round": 1e3 * el / nb,
                                "note": "one context, one host thread, one launch per kernel for the whole batch"})
            bd.close()

    # informational (never `value`): BASELINE config 5 on ONE GPU - 8 warm-started allegro_hand N=60 problems advanced
    # through whole trust-region iterations (scaling, dogleg, trial point, ratio, accept / reject on the device) by
    # idto_hip_tr_solve_batch: one launch set per iteration for all of them, one host thread, one wait
    batch_tr = None
    if world == 1 and args.batch > 1 and not args.no_full:
        try:
            c5, m5 = load_config("allegro_hand"), load_model("allegro_hand")
            N5, B5, it5 = 60, 8, 10
            probs5, qs5 = [], []
            for b in range(B5):
                p5, s5, _ = make_problem(c5, m5, num_steps=N5)
                probs5.append(p5)
                qs5.append(synthetic_trajectory(c5, m5, N5, seed=b, lower=0.01))
            # the example's YAML (examples/allegro_hand/allegro_hand.yaml:95): equality constraints on the unactuated ball
            # ENFORCED - idto_hip_tr_solve_batch_constrained; the unconstrained batch loop next to it
            s5.scaling, s5.equality_constraints = True, True
            dofs5 = list(m5.unactuated_dofs)
            d5 = hip.HipPath(m5, probs5, s5, device=local_rank)
            d5.set_stream(stream.cuda_stream)
            times, times_u = [], []
            for rep in range(4):   # (the first constrained call creates the per-problem contexts: not timed)
                d5.set_q_batch(np.array(qs5))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rows5, _ = d5.tr_solve_batch_constrained(it5, 2, True, False, 1e-1, 1e5, dofs5)
                if rep:
                    times.append(time.perf_counter() - t1)
            for rep in range(3):
                d5.set_q_batch(np.array(qs5))
                d5.eval_tau()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rows5u, _ = d5.tr_solve_batch(it5, 2, True, False, 1e-1, 1e5)
                times_u.append(time.perf_counter() - t1)
            d5.close()
            d1 = hip.HipPath(m5, probs5[0], s5, device=local_rank)
            d1.set_stream(stream.cuda_stream)
            t_single = []
            for rep in range(3):
                d1.set_q(qs5[0])
                d1.eval_tau()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                d1.tr_solve(it5, 2, True, False, 1e-1, 1e5, constrained_dofs=dofs5)
                t_single.append(time.perf_counter() - t1)
            d1.close()
            batch_tr = {"workload": f"allegro_hand + sphere N={N5}, {B5} problems in one batch context, {it5} trust-region "
                                    f"iterations each (double_sqrt scaling, equality constraints ENFORCED on the {len(dofs5)} "
                                    "unactuated degrees of freedom, as in the example's YAML)",
                        "ms_per_iteration_of_the_batch": 1e3 * min(times) / it5,
                        "value": B5 * it5 / min(times), "unit": "trust-region iterations/s (aggregate)",
                        "ms_per_iteration_one_problem_alone": 1e3 * min(t_single) / it5,
                        "accepted_steps": int(rows5[:, :, 9].sum()), "flags_clean": bool((rows5[:, :, 14] == 0).all()),
                        "without_enforced_constraints": {"ms_per_iteration_of_the_batch": 1e3 * min(times_u) / it5,
                                                         "value": B5 * it5 / min(times_u),
                                                         "flags_clean": bool((rows5u[:, :, 14] == 0).all())}}
        except Exception as e:  # informational only
            batch_tr = {"error": str(e)[:200]}

    units = args.steps * (world if (world > 1 and not sharded) else 1)
    value = units / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    if rank == 0:
        algb = algorithmic_bytes(N, nq, nv)
        # the fused launch: SURVEY.md §8d's B_alg of the whole iteration (every intermediate written
        # once and read once): 2,605,184 B for mini_cheetah N=40
        algb.append(8 * (2 * 3 * N * nv * nq + 2 * 3 * (N + 1) * nq * nq + 2 * 5 * (N + 1) * nq * nq
                         + 4 * (N + 1) * nq + 2 * (N + 1) * nv + 3 * N * nv))
        dom = 3 if fused_run else int(np.argmax([k[0] for k in kern[:3]]))
        names = list(KERNELS)
        if dev.get_option("last_solver") == 2:   # nested dissection over seven workgroups (csrc/penta_nd.h)
            names[2] = "penta_nd_kernel"
        elif dev.get_option("last_solver") == 4:   # ... with pipelined chains, five workgroups (csrc/penta_pipe.h)
            names[2] = "penta_pipe_kernel"
        elif dev.get_option("last_solver") == 6:   # the small models: scalar band factorisation, one workgroup (csrc/penta_band.h)
            names[2] = "penta_band_kernel"
        elif dev.get_option("last_solver") == 7:   # ... and their whole step in one workgroup of one launch (csrc/gn_small.h)
            names[3] = "gn_small_kernel"
        if dev.get_option("last_assembly") == 1:   # products formed by fd_kernel, combined here (kernels.h)
            names[1] = "assemble_terms_kernel"
        asm_inside = dev.get_option("last_assembly") == 4   # the solver's launch assembled g and H itself (penta_pipe.h PipeAsm)
        if asm_inside:
            names[1] = "(inside %s)" % names[2]
            algb[2] += algb[1]   # ... so its algorithmic bytes are the assembly's plus the solver's (2.6 MB)
        dur_s = kern[dom][0] * 1e-3
        achieved = algb[dom] / dur_s / 1e9 if dur_s > 0 else 0.0
        traffic, traffic_src = pmc_traffic(names[dom])
        out = {
            "metric": f"Gauss-Newton iters/sec (grad+Hessian+solve), {args.config} N={N}",
            "value": value, "unit": "GN iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config} (nq={nq}, nv={nv}, {model.npairs} contact pairs), horizon N={N}, "
                                   f"dt={prob.time_step}, forward differences, one Gauss-Newton iteration per step",
                       "physics": "inverse dynamics / contact as defined by the in-repo oracle (oracle/rigid_body.h); "
                                  "Drake is not available, multi-body conventions are unpinned against it (DESIGN.md §8)",
                       "parallelism": ("single GPU" if world == 1 else
                                       (f"t-range shard of the perturbation grid over {world} GPUs + RCCL all-gather "
                                        f"of the dtau/dq slabs ({'ncclAllGather inside libidto_hip.so' if exchange == 'rccl' else 'torch.distributed'}), "
                                        f"redundant assemble+solve" if sharded else
                                        f"{world} independent replicas"))},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": algb[dom], "avg_launch_ms": kern[dom][0],
                         "launches_timed": kern[dom][1],
                         "all_kernels_avg_ms": {names[i]: kern[i][0] for i in range(4) if kern[i][1] > 0},
                         "wasted_traffic_ratio": (traffic / algb[dom]) if traffic else None,
                         "algorithmic_bytes_include_the_assembly": bool(asm_inside and dom == 2),
                         "latency_model": latency_model(kern, names),
                         "note": "latency/dependency-bound path (SURVEY.md §8d): the HBM fraction is intrinsically small; "
                                 "latency_model states the bounds that apply and the achieved fraction of those"},
        }
        notes = []
        if world > 1:
            notes.append(SHARD_NOTE)
        if visible:
            notes.append(f"{world} ranks share {visible} visible device(s): value is {world} independent replicas time-sharing "
                         "them, NOT a multi-GPU measurement")
            out["gpus_visible"] = visible
        elif exchange_note:
            notes.append(exchange_note)
        if notes:
            out["config"]["note"] = "; ".join(notes)
        if other_extra is not None:
            out[other_key] = other_extra
        if bit_identical is not None:
            out["bit_identical_to_unsharded"] = bit_identical
        if config5 is not None:
            out["config5_workload"] = config5
        try:   # which RCCL this process resolved (libidto_hip.so checks the major version where a communicator is created)
            rp, rv = hip.rccl_info()
            out["rccl"] = {"path": rp, "version_code": rv}
        except Exception as e:
            out["rccl"] = {"error": str(e)[:120]}
        out["step_latency_ms"] = latency
        if batch_extra is not None:
            out["batch_mode"] = batch_extra
        if batch_tr is not None:
            out["batch_trust_region"] = batch_tr
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, N)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if world == 1 and not args.no_full:
            out["full_iteration"] = full_iteration(cfg, model, N, local_rank, not args.no_cpu)
            try:
                out["mpc_replan"] = mpc_replan(cfg, model, local_rank)
            except Exception as e:   # (informational: never costs the bench line)
                out["mpc_replan"] = {"error": str(e)[:200]}
        print(json.dumps(out))
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
