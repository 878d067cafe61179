"""CPU tests of host logic: reference generators, sharding and the summary gather (gloo, world 2)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from linearmpchumanoid_amd import sharding, trajectories
from oracle.pyoracle import Oracle


def test_stance_zmp_matches_reference_rule():
    zx, zy = trajectories.stance_zmp(5.0, 0.01, 2)
    assert len(zx) == 550 and not zx.any() and not zy.any()
    assert trajectories.stance_zmp(1.0, 0.001, 0)[1][0] == -0.05
    assert trajectories.stance_zmp(1.0, 0.001, 1)[1][0] == 0.05


def test_foot_coeff_trajectory_against_oracle():
    o = Oracle(do_ik=False)
    rF, rn, lF, ln = o.foot_coeffs()
    co, n = trajectories.foot_coeff_trajectory([0, -0.05, 0], [0, -0.05, 0], 0.0, 5.0)
    assert list(n) == list(rn) == [6, 6, 8]
    assert np.abs(co - rF).max() < 1e-12
    co2, _ = trajectories.foot_coeff_trajectory([0, 0.05, 0], [0.04, 0.05, 0], 0.02, 0.5)
    t = np.linspace(0, 0.5, 11)
    z = sum(co2[2, i] * t ** i for i in range(8))
    assert abs(z[0]) < 1e-9 and abs(z[-1]) < 1e-9 and abs(z[5] - 0.02) < 1e-9
    x = sum(co2[0, i] * t ** i for i in range(6))
    assert abs(x[0]) < 1e-12 and abs(x[-1] - 0.04) < 1e-9


def test_shard_ranges_cover_everything():
    for total in (1, 7, 1024, 32768, 1000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                first, count = sharding.shard_range(total, world, r)
                seen += list(range(first, first + count))
            assert seen == list(range(total))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = sharding.shard_range(total, world, rank)
    g = torch.Generator().manual_seed(1234)
    full_state = torch.rand((total, 96), dtype=torch.float64, generator=g)
    full_out = torch.rand((total, 72), dtype=torch.float64, generator=g)
    full_status = torch.randint(-2 ** 31, 2 ** 31 - 1, (total, 4), dtype=torch.int64, generator=g).to(torch.int32)
    s = sharding.make_summary(full_state[first:first + count], full_out[first:first + count], full_status[first:first + count])
    gathered = sharding.gather_summaries(s, world, rank)
    if rank == 0:
        ref = sharding.make_summary(full_state, full_out, full_status)
        q.put(bool(torch.equal(gathered, ref)))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_summary_gather_world2_gloo():
    """N>1 path: shards + one gather equal the single-process summary (uneven tail included)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 37, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_summary_fields():
    st = torch.zeros((2, 96), dtype=torch.float64); out = torch.zeros((2, 72), dtype=torch.float64)
    status = torch.tensor([[5, 2, 0, 0x0F], [7, 1, 1, -1]], dtype=torch.int32)
    out[:, 24 + 5] = 20.0; out[:, 24 + 11] = 32.0; out[0, 3] = -9.0
    s = sharding.make_summary(st, out, status)
    assert s[0, 7] == 9.0 and s[0, 8] == 52.0 and s[0, 14] == 4 and s[1, 14] == 32 and s[1, 13] == 1 and s[1, 11] == 7


def test_walk_plan_is_consistent():
    """Build-defined walking references: phases, ZMP and swing polynomials agree with each other."""
    from linearmpchumanoid_amd.capi import PHASE_DOUBLE, PHASE_LEFT, PHASE_RIGHT
    dt = 1e-3
    p = trajectories.walk_plan(2.5, dt, num_steps=4, time_per_step=0.5, ds_time=0.1, step_height=0.02, settle_time=0.3)
    n = len(p["zmp_x"])
    assert n == int((2.5 + 0.5) / dt) and len(p["phase"]) == n and len(p["seg_of_sample"]) == n
    assert p["seg_of_sample"].max() < len(p["segs"]) and p["segs"].shape[1] == 52
    ph = p["phase"]
    assert set(np.unique(ph)) == {PHASE_DOUBLE, PHASE_LEFT, PHASE_RIGHT}
    assert (p["zmp_y"][ph == PHASE_RIGHT] == -0.05).all() and (p["zmp_y"][ph == PHASE_LEFT] == 0.05).all()
    assert (p["zmp_y"][ph == PHASE_DOUBLE] == 0).all()
    # swing polynomials start / end at rest on the ground and reach the step height half-way
    for k in np.where(np.diff(ph.astype(int)) != 0)[0][:2]:
        kk = k + 1
        if ph[kk] == PHASE_DOUBLE:
            continue
        g = p["segs"][p["seg_of_sample"][kk]]
        co = g[1 + 24:1 + 48].reshape(3, 8) if ph[kk] == PHASE_RIGHT else g[1:25].reshape(3, 8)
        T = (np.sum(p["seg_of_sample"] == p["seg_of_sample"][kk])) * dt
        z = lambda t: sum(co[2, i] * t ** i for i in range(8))
        assert abs(z(0)) < 1e-9 and abs(z(T)) < 1e-9 and abs(z(T / 2) - 0.02) < 1e-9
    # feet end side by side
    last = p["segs"][p["seg_of_sample"][-1]]
    assert last[1] == last[1 + 24]


def test_jump_plan_schedule():
    """Build-defined jumping schedule (BASELINE config 5): DS -> flight -> DS on the stanceZMP sample grid."""
    from linearmpchumanoid_amd.capi import PHASE_DOUBLE, PHASE_FLIGHT
    dt = 1e-3
    p = trajectories.jump_plan(1.0, dt, stance_time=0.4, flight_time=0.15)
    n = int((1.0 + 0.5) / dt)
    assert len(p["phase"]) == n == len(p["zmp_x"]) == len(p["zmp_y"])
    assert (p["phase"][:400] == PHASE_DOUBLE).all() and (p["phase"][400:550] == PHASE_FLIGHT).all() and (p["phase"][550:] == PHASE_DOUBLE).all()
    assert not p["zmp_x"].any() and not p["zmp_y"].any()
