"""CPU tests of host logic: reference generators, sharding and the summary gather (gloo, world 2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from linearmpchumanoid_amd import sharding, trajectories
from oracle.pyoracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_summary_gather_world8_gloo_at_config4_size():
    """BASELINE configs[3] / [4] as the driver's 8-GPU run shards them (no multi-GPU node was available to any
    session): 8 ranks over gloo, 32 768 + 5 summary rows so that the shards are uneven (ranks 0..4 own 4097 rows, 5..7 own 4096) --
    the gathered table equals the single-process summary row for row: order and count exact."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    total = 32768 + 5
    assert [sharding.shard_range(total, 8, r)[1] for r in range(8)] == [4097] * 5 + [4096] * 3
    procs = [ctx.Process(target=_worker, args=(r, 8, port, total, q)) for r in range(8)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok


def test_bench_refuses_a_local_rank_without_a_device():
    """bench.py under a launcher whose LOCAL_RANK has no GPU behind it (a node with fewer cards than ranks) says so in one line instead
    of failing inside torch.cuda.set_device / the process group."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="11", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "LOCAL_RANK=11" in r.stderr and "visible" in r.stderr, r.stderr[-500:]


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


# --------------------------------------------------------------------------- bench.py entry path (no GPU needed)
def _bench_mod():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lmh_bench", os.path.join(root, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m, root


def test_bench_defaults_name_the_largest_single_gpu_config():
    """python bench.py (N = 1) measures BASELINE configs[2]: 4096 robots, walking, N = 32 x mpc_dt 10 ms, one step = one whole 4 000-tick
    rollout segment, nothing restarted; --gpus N > 1 one GPU's share of configs[3]; --coupled keeps rounds 1-2's line."""
    b, _ = _bench_mod()
    a = b.parse([])
    assert (a.config, a.instances, a.horizon, a.gpus, a.ticks, a.mpc_dt, a.reset_every) == (3, 4096, 32, 1, 4000, 1e-2, 10)
    assert (a.step_time, a.ds_time) == (0.5, 0.2)                   # the reference's default timePerStep (zmpGeneration.hpp:37-38)
    assert a.horizon * a.mpc_dt >= 0.32 - 1e-12                     # a preview the LIPM loop is stable with
    assert a.steps * a.ticks * a.instances / 20e6 >= 3.0            # timed region >= 3 s even at 20 M ticks/s
    d = b.parse(["--steps", "20", "--warmup", "5"])                 # the driver's flags
    assert d.ticks == 4000 and d.reset_every == 10
    a8 = b.parse(["--gpus", "8"])
    assert (a8.config, a8.instances, a8.ticks) == (4, 4096, 4000)
    a2 = b.parse(["--config", "2"])
    assert (a2.instances, a2.horizon, a2.ticks, a2.mpc_dt, a2.push) == (1024, 16, 2000, 2e-2, 0.5)
    a5 = b.parse(["--config", "5", "--gpus", "8"])
    assert (a5.instances, a5.horizon, a5.ticks, a5.mpc_dt, a5.reset_every) == (4096, 48, 2000, 1e-2, 1)
    c = b.parse(["--coupled"])
    assert (c.ticks, c.mpc_dt, c.step_time) == (40, 1e-3, 0.2) and c.reset_every * c.ticks <= 480
    c2 = b.parse(["--coupled", "--config", "2"])
    assert (c2.ticks, c2.horizon, c2.push) == (10, 16, 1.0) and c2.reset_every * c2.ticks <= 230
    txt = b.WORKLOAD_TEXT[3].format(B=4096, dt=1e-3, N=32, md=1e-2, push=1.0, st=0.5)
    assert "4096" in txt and "walking" in txt and "N=32" in txt and "mpc_dt=0.01" in txt
    assert b.HARD_FLAGS == 15 | 32                                  # MAXITER | NONFINITE | ZMP_RANGE | NOT_SPD | UNFINISHED
    e = b.parse(["--mode", "eval"])
    assert (e.config, e.instances, e.steps, e.warmup) == (3, 4096, 200, 20) and (d.steps, d.warmup) == (20, 5) and (a2.steps, a2.warmup) == (8, 2)


def test_bench_gpus_flag_launches_ranks_and_propagates_failure():
    """`bench.py --gpus 2` with no launcher starts two ranks of itself and fails if a rank fails.  In this container there is
    no GPU, so both ranks stop at the product's `needs a GPU` assertion: the parent must exit non-zero and print no JSON line.
    A --gpus that disagrees with WORLD_SIZE is refused before anything is initialised."""
    import subprocess
    if torch.cuda.is_available():
        return                                                      # the GPU form of this test lives in test_gpu_parity.py
    _, root = _bench_mod()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-cpu-baseline", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '"metric"' not in r.stdout
    assert "needs a GPU" in r.stderr or "HIP" in r.stderr or "no CPU fallback" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "disagrees with WORLD_SIZE" in r.stderr


def test_c_abi_record_files_round_trip_and_match_the_python_format(tmp_path, hip_lib):
    """lmh_write_summary / lmh_read_summary / lmh_write_log / lmh_read_log (host-only entry points of the C ABI) against
    linearmpchumanoid_amd.wire: same 64-byte header, same payload, both directions; bad files are refused."""
    import ctypes as C
    from linearmpchumanoid_amd import wire
    from linearmpchumanoid_amd.controller import BatchedController
    from linearmpchumanoid_amd.capi import LmhError
    rng = np.random.default_rng(11)
    s = rng.normal(size=(37, 16)); lg = rng.normal(size=(5, 3, 36))
    pc, pp = tmp_path / "c.lmhsum", tmp_path / "p.lmhsum"
    BatchedController.write_summary(pc, s, dt=1e-3)
    wire.write_summary(pp, s, dt=1e-3)
    assert pc.read_bytes() == pp.read_bytes()
    back, dt = BatchedController.read_summary(pp)
    assert np.array_equal(back, s) and dt == 1e-3
    lc, lp = tmp_path / "c.lmhlog", tmp_path / "p.lmhlog"
    BatchedController.write_log(lc, lg, dt=1e-3, t0=0.25)
    wire.write_log(lp, lg, dt=1e-3, t0=0.25)
    assert lc.read_bytes() == lp.read_bytes()
    back, dt, t0 = BatchedController.read_log(lp)
    assert np.array_equal(back, lg) and (dt, t0) == (1e-3, 0.25)
    import pytest
    with pytest.raises(LmhError):
        BatchedController.read_log(pc)                              # wrong magic
    pc.write_bytes(pc.read_bytes()[:-8])
    with pytest.raises(LmhError):
        BatchedController.read_summary(pc)                          # truncated payload
    n = C.c_uint64(0)
    assert hip_lib.lmh_read_summary(str(pp).encode(), back.ctypes.data_as(C.c_void_p), 5, C.byref(n), None) != 0    # buffer too small


def test_create_rejects_values_the_kernels_would_divide_by(hip_lib):
    """lmh_create validates the literals before it touches a device: non-positive weights / mu / eps are LMH_ERR_BAD_ARG
    (a zero weight is 1/0 in the Woodbury set-up); the defaults pass validation (and then fail with NO_DEVICE here)."""
    import ctypes as C
    from linearmpchumanoid_amd.capi import LmhConfig
    bad = [("w_com_lin", 0.0), ("w_foot", 0.0), ("w_force", -1.0), ("w_joints", 0.0), ("w_base_pos", 0.0), ("w_base_ang", float("nan")),
           ("mu", 0.0), ("eps_coeff", 0.0), ("w_com_ang", -1.0), ("max_qp_iters", 0), ("precision", 7), ("dt", 0.0), ("gravity", 0.0)]
    for name, val in bad:
        cfg = LmhConfig()
        hip_lib.lmh_config_default(C.byref(cfg))
        setattr(cfg, name, val)
        h = C.c_void_p()
        assert hip_lib.lmh_create(C.byref(cfg), 2, 0, C.byref(h)) == -2, name
        assert not h.value
    cfg = LmhConfig()
    hip_lib.lmh_config_default(C.byref(cfg))
    cfg.w_com_ang = 50.0; cfg.mu = 0.4; cfg.bpp_rounds = -1
    h = C.c_void_p()
    rc = hip_lib.lmh_create(C.byref(cfg), 2, 0, C.byref(h))
    assert rc == (0 if torch.cuda.is_available() else -1)
    if h.value:
        hip_lib.lmh_destroy(h)


def _queue_model(rng, n_inst, n_chunks, grid, resident, spin_limit=None, atomic_take=True):
    """One run of the model of the rollout kernel's work queue (lmh_kernels.hip: rollout_claim, the push at the end of a chunk, the last
    workgroup's clean-up) under a random interleaving.  The take of a ring entry is TWO steps, as on the device: a load that sees a
    non-zero slot, then an exchange whose RESULT decides (atomic_take=False models the round-3 kernel, which trusted the load).
    spin_limit: polls after which a wait gives up (None = never).  Returns what the launch leaves behind."""
    head = tail = left = err = 0
    ring = [0] * n_inst
    prog = [0] * n_inst
    done = [[] for _ in range(n_inst)]
    flagged = set()
    n_units = n_inst * n_chunks
    # a workgroup: 'claim' -> ('wait', slot, polls) -> ('take', slot, seen, polls) -> ('run', robot, chunk) -> ('push', slot, robot, polls) -> 'claim' ... -> 'gone'
    wgs = ["new"] * grid
    running = set()
    steps = 0
    while any(w != "gone" for w in wgs):
        steps += 1
        assert steps < 400000, "the queue does not drain"
        # the hardware keeps at most `resident` workgroups on the chip; one that has started stays until it leaves
        cand = [i for i, w in enumerate(wgs) if w != "gone" and (i in running or len(running) < resident)]
        i = rng.choice(cand)
        running.add(i)
        w = wgs[i]

        def leave():
            nonlocal left, head, tail
            wgs[i] = "gone"; running.discard(i)
            left += 1
            if left == grid:                                   # the last workgroup to leave resets the slot
                if err:
                    for r in range(n_inst):
                        if prog[r]:
                            flagged.add(r); prog[r] = 0
                        ring[r] = 0
                head = tail = left = 0

        if w == "new" or w == "claim":
            n = head; head += 1
            if n >= n_units:
                leave()
            elif n < n_inst:
                wgs[i] = ("run", n, 0)
            else:
                wgs[i] = ("wait", (n - n_inst) % n_inst, 0)
        elif w[0] == "push":                                   # (a slow taker of the entry one lap earlier may still hold the slot)
            if ring[w[1]] == 0:
                ring[w[1]] = w[2] + 1
                wgs[i] = "claim"
            elif spin_limit is not None and w[3] + 1 >= spin_limit:
                err |= 2; wgs[i] = "claim"                     # the robot stays out of the queue, prog[robot] != 0 marks it
            else:
                wgs[i] = ("push", w[1], w[2], w[3] + 1)
        elif w[0] == "wait":                                   # the relaxed load
            if ring[w[1]]:
                wgs[i] = ("take", w[1], ring[w[1]], w[2])
            elif spin_limit is not None and w[2] + 1 >= spin_limit:
                err |= 1; leave()
            else:
                wgs[i] = ("wait", w[1], w[2] + 1)
        elif w[0] == "take":                                   # the exchange
            v = ring[w[1]]; ring[w[1]] = 0
            if not atomic_take:
                v = w[2]                                       # round 3: the value the load saw
            if v:
                wgs[i] = ("run", v - 1, prog[v - 1])
            else:
                wgs[i] = ("wait", w[1], w[3] + 1)              # somebody else got it: keep waiting for the next push to this slot
        else:
            _, r, c = w
            done[r].append(c)
            if c + 1 < n_chunks:
                prog[r] = c + 1
                wgs[i] = ("push", tail % n_inst, r, 0); tail += 1   # the position is reserved; the entry goes in once the slot is empty
            else:
                prog[r] = 0
                wgs[i] = "claim"
    return dict(done=done, ring=ring, prog=prog, head=head, tail=tail, left=left, err=err, flagged=flagged)


QUEUE_CASES = [(7, 3, 4, 2), (16, 5, 8, 8), (5, 1, 5, 3), (9, 4, 12, 1), (32, 6, 8, 5), (3, 9, 6, 4), (9, 4, 9, 9), (4, 8, 4, 4), (6, 3, 6, 2)]


def test_rollout_work_queue_protocol_drains_under_any_schedule():
    """Model of the rollout kernel's work queue (_queue_model), run under random interleavings with FEWER runners than workgroups of
    the grid resident at a time (and with grid == robots, where laps of the ring meet): every (robot, chunk) unit is executed exactly
    once and in order per robot, no runner waits for ever, and the counters / ring are back at zero.  The invariants the kernel relies
    on: a claim beyond the robots' first chunks waits for a ring entry that only a RUNNING workgroup can push, and an entry is taken by
    the exchange's result."""
    import random
    for seed, case in enumerate(QUEUE_CASES * 12):
        n_inst, n_chunks, grid, resident = case
        r = _queue_model(random.Random(seed), *case)
        assert all(d == list(range(n_chunks)) for d in r["done"]), case
        assert not any(r["ring"]) and not any(r["prog"]) and r["err"] == 0 and not r["flagged"]
        assert r["head"] == 0 and r["tail"] == 0 and r["left"] == 0


def test_rollout_work_queue_model_sees_the_non_atomic_take():
    """The same model with the round-3 take (load, then an unconditional exchange whose result is ignored) runs robots' chunks twice
    under some schedules: the model is able to see the defect the advisor reported, i.e. the green test above means
    something."""
    import random
    broken = 0
    for seed in range(400):
        r = _queue_model(random.Random(seed), 9, 4, 9, 9, atomic_take=False, spin_limit=50)
        if any(d != list(range(4)) for d in r["done"]):
            broken += 1
    assert broken > 0


def test_rollout_work_queue_timeouts_are_loud_and_leave_a_clean_slot():
    """Bounded waits (LMH_SPIN_LIMIT): when a poll budget runs out the launch's error word is set, every robot is either complete --
    all its chunks exactly once, in order -- or flagged by the last workgroup to leave (a flagged robot ran a prefix of its chunks),
    nothing runs twice, and ring / progress / counters are back at zero for the next launch on the slot."""
    import random
    seen_err = seen_partial = 0
    for seed in range(300):
        case = QUEUE_CASES[seed % len(QUEUE_CASES)]
        n_inst, n_chunks, grid, resident = case
        r = _queue_model(random.Random(1000 + seed), *case, spin_limit=(seed % 4) + 1)
        for rb, d in enumerate(r["done"]):
            assert d == list(range(len(d))), (case, seed)                       # in order, never twice
            assert len(d) == n_chunks or rb in r["flagged"], (case, seed)       # complete or flagged
            assert not (len(d) == n_chunks and rb in r["flagged"])
        assert bool(r["flagged"]) <= bool(r["err"])
        assert not any(r["ring"]) and not any(r["prog"]) and r["head"] == 0 and r["tail"] == 0 and r["left"] == 0
        seen_err += bool(r["err"]); seen_partial += bool(r["flagged"])
    assert seen_err > 20 and seen_partial > 20
