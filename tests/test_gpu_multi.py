"""The multi-GPU iteration through the library's own RCCL communicator (include/idto_hip.h
idto_hip_comm_*) on the ONE GPU of the test box:
  * world = 1: comm_init / comm_init_all + gn_step_sharded / gn_step_multi execute ncclCommInitRank,
    ncclAllGather (in place on the slab) and ncclCommDestroy on hardware and give the unsharded step;
  * world = 2 with both ranks on the same device, one process each: RCCL may refuse a duplicate
    GPU - then the test is skipped with RCCL's message; where it is allowed, both ranks must hold
    the complete slab and the step of the unsharded run, bit for bit.
The world_size-2 host logic (shard bounds, padding, idempotence) is covered on CPU with gloo in
tests/test_multi_rank.py."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from idto_amd import hip
from idto_amd.model import load_model
from idto_amd.problem import load_config, make_problem, synthetic_trajectory

pytestmark = pytest.mark.gpu


def _setup(name, N, seed=3):
    cfg, model = load_config(name), load_model(name)
    prob, sp, _ = make_problem(cfg, model, num_steps=N)
    sp.scaling = False
    sp.equality_constraints = False
    return model, prob, sp, synthetic_trajectory(cfg, model, N, seed=seed, lower=0.01)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("name,N", [("mini_cheetah", 40), ("hopper", 7)])
def test_world1_communicator_runs_the_rccl_path(name, N):
    model, prob, sp, q = _setup(name, N)
    ref = hip.HipPath(model, prob, sp)
    ref.set_q(q)
    ref.gn_step()
    dev = hip.HipPath(model, prob, sp)
    dev.set_q(q)
    dev.comm_init(hip.comm_unique_id(), 0, 1)
    for _ in range(2):
        dev.gn_step_sharded()
    assert _same(dev.get("slab"), ref.get("slab")) and _same(dev.get("step"), ref.get("step"))
    dev.comm_destroy()
    dev.gn_step()           # back to the single-GPU path (full k-range again)
    assert _same(dev.get("step"), ref.get("step"))
    # one process, "several" devices (here one): ncclCommInitAll + grouped all-gather
    multi = [hip.HipPath(model, prob, sp)]
    multi[0].set_q(q)
    hip.comm_init_all(multi)
    hip.gn_step_multi(multi)
    assert _same(multi[0].get("step"), ref.get("step"))
    for d in (ref, dev, multi[0]):
        d.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, name, N, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only
    try:
        from idto_amd.multi_gpu import RcclShard
        model, prob, sp, q = _setup(name, N)
        ref = hip.HipPath(model, prob, sp, device=0)
        ref.set_q(q)
        ref.gn_step()
        dev = hip.HipPath(model, prob, sp, device=0)   # both ranks on the one GPU
        dev.set_q(q)
        try:
            sh = RcclShard(dist, dev, rank, world)
        except hip.HipError as e:
            out[rank] = "refused: " + str(e)[:300]
            return
        for _ in range(3):
            dev.gn_step_sharded()
        ok = _same(dev.get("slab"), ref.get("slab")) and _same(dev.get("step"), ref.get("step"))
        out[rank] = "ok" if ok else "mismatch"
        sh.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("name,N", [("mini_cheetah", 9)])   # ragged split: 5 + 4 records
def test_two_ranks_on_one_gpu(name, N):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    env = dict(os.environ)
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        mp.spawn(_rank_main, args=(world, _free_port(), name, N, out), nprocs=world, join=True)
    finally:
        os.environ.clear()
        os.environ.update(env)
    res = dict(out)
    if any(str(v).startswith("refused") for v in res.values()):
        pytest.skip(f"RCCL does not form a 2-rank communicator on one device: {res}")
    assert all(v == "ok" for v in res.values()), res


def test_cpp_optimizer_over_a_device_list(monkeypatch):
    """idto::optimizer::TrajectoryOptimizer with a device list (here the one device of the box):
    the sharded evaluation of the partials (idto_hip_comm_init_all + idto_hip_eval_partials_multi)
    inside Solve gives the iterates of the single-device optimizer, bit for bit.  (hopper's YAML enforces its equality
    constraints: the device-list optimizer runs the host loop with the Schur-complement route, so the single-device
    one is held to that route too - IDTO_CON_KKT=0; the banded KKT step agrees with it to the tolerances of
    tests/test_gpu_trust_region.py, not to the bit.)"""
    monkeypatch.setenv("IDTO_CON_KKT", "0")
    from idto_amd.optimizer import TrajectoryOptimizer, TrajectoryOptimizerSolution, TrajectoryOptimizerStats
    name, N = "hopper", 20
    cfg, model = load_config(name), load_model(name)
    prob, sp, q_guess = make_problem(cfg, model, num_steps=N)
    sp.max_iterations, sp.verbose = 6, False
    outs = []
    for kw in (dict(device=0), dict(devices=[0])):
        opt = TrajectoryOptimizer(model, prob, sp, **kw)
        sol, st = TrajectoryOptimizerSolution(), TrajectoryOptimizerStats()
        flag = opt.Solve(q_guess, sol, st)
        outs.append((flag, sol.q.copy(), np.array(st.iteration_costs)))
        opt.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="needs two visible devices (runs by itself on a multi-GPU node)")
@pytest.mark.parametrize("name,N,ndev", [("mini_cheetah", 9, 2), ("mini_cheetah", 40, 2), ("allegro_hand", 13, 2),
                                         ("mini_cheetah", 41, 4), ("mini_cheetah", 40, 8)])
def test_devices_of_one_process_share_the_horizon(name, N, ndev):
    """One process, one context per device, idto_hip_comm_init_all + idto_hip_gn_step_multi: every device evaluates
    its k-range (ragged: 9 = 5 + 4, 41 over 4, 13 over 2), the grouped RCCL all-gather completes every slab in place,
    every device assembles and solves - slab and step of EVERY device equal the unsharded run bit for bit
    (trajectory_optimizer.cc:455-457: the reference's parallel loop over time steps, here over devices)."""
    if _device_count() < ndev:
        pytest.skip(f"{ndev} devices wanted, {_device_count()} visible")
    model, prob, sp, q = _setup(name, N)
    ref = hip.HipPath(model, prob, sp, device=0)
    ref.set_q(q)
    ref.gn_step()
    want_slab, want_step = ref.get("slab"), ref.get("step")
    ref.close()
    devs = [hip.HipPath(model, prob, sp, device=d) for d in range(ndev)]
    for d in devs:
        d.set_q(q)
    hip.comm_init_all(devs)
    for _ in range(3):
        hip.gn_step_multi(devs)
    for r, d in enumerate(devs):
        assert _same(d.get("slab"), want_slab), f"slab of device {r}"
        assert _same(d.get("step"), want_step), f"step of device {r}"
    for d in devs:
        d.comm_destroy()
        d.close()
