"""GPU tests added in round 4: an incomplete rollout is loud (bounded waits of the kernel's work queue, LMH_FLAG_UNFINISHED,
LMH_ERR_UNFINISHED, clean launch slot), BASELINE config 2 at its full push amplitude against the oracle on robots that fall, the
evaluation-API lines of bench.py.  Same rules as the other GPU files: HIP path through the C ABI, the CPU oracle is the checker.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import WEIGHT, close, perturbed_velocities, vec_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DT = 1e-3


def _run_probe(code, variant, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("LMH_VARIANT", "LMH_DIAG")}
    if variant:
        env["LMH_VARIANT"] = variant
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


# ------------------------------------------------------------------------------- an incomplete rollout is loud
_QFAULT_PROBE = r"""
import hashlib, json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from linearmpchumanoid_amd import capi
from linearmpchumanoid_amd.controller import BatchedController, default_config
from helpers import perturbed_velocities
ik = json.load(open("tests/golden/ik_posture.json"))
B, NT = 96, 620                                    # 620 ticks = chunks of 250 + 250 + 120: every robot passes through the ring queue twice
capi.lib().lmh_debug_build_flags.restype = int
ctl = BatchedController(B, default_config(dt=1e-3, time_horizon=0.32 + 1e-9, z_com=ik["z_com"], mpc_dt=1e-2, warm_start=1))
ctl.gen_walk(2.0, num_steps=3, time_per_step=0.3, ds_time=0.1, step_height=0.02, settle_time=0.05)
ctl.set_xscale(np.linspace(0.02, 0.05, B))
v0 = perturbed_velocities(B) * 0.3
res = {"build_flags": capi.lib().lmh_debug_build_flags()}
st = ctl.new_state(np.array(ik["q"]), v0, t=0.0)
out, status, _ = ctl.rollout(st, NT)
codes = []
for _ in range(2):                                  # the error is reported once
    try:
        ctl.synchronize(); codes.append(0)
    except capi.LmhError as e:
        codes.append(e.code)
s = status.cpu().numpy(); t = st[:, 90].cpu().numpy()
res.update(sync_codes=codes, flags=s[:, 2].tolist(), ticks=[int(round(x / 1e-3)) for x in t])
h1 = hashlib.sha256()
for a in (out, status, st):
    h1.update(a.cpu().numpy().tobytes())
res["first_sha"] = h1.hexdigest()
# the following launches on the SAME handle: single chunks (no ring traffic, so the fault injection has nothing to act on), enough of them to
# come back to the launch slot the incomplete launch used (8 slots per handle)
h = hashlib.sha256()
st2 = ctl.new_state(np.array(ik["q"]), v0, t=0.0)
for i in range(9):
    out2, status2, log2 = ctl.rollout(st2, 60, log=True)
    ctl.synchronize()
    for a in (out2, status2, log2, st2):
        h.update(a.cpu().numpy().tobytes())
res["next_sha"] = h.hexdigest()
res["next_flags"] = int((status2.cpu().numpy()[:, 2] != 0).sum())
print(json.dumps(res))
"""


def test_incomplete_rollout_is_loud_and_leaves_a_clean_slot():
    """The waits of the rollout kernel's work queue are bounded; when one runs out nothing may fail silently (the reference prints and
    aborts, src/controller.cpp:448-476).  Checker build `qfault` (-DLMH_SPIN_LIMIT=64 -DLMH_TEST_LOSE_PUSH=7, build.CHECKER_VARIANTS):
    the ring entry of every robot with index = 3 mod 7 is never pushed, and a claim gives up after 64 polls.  On a three-chunk launch of
    96 robots: (i) every robot is either complete (620 ticks, no UNFINISHED flag) or carries LMH_FLAG_UNFINISHED with a whole number of
    chunks behind it -- the robots whose push was lost among them; (ii) lmh_synchronize reports LMH_ERR_UNFINISHED, once; (iii) nine
    following launches on the same handle (they come back to the slot of the incomplete one) are flag-free and BIT-IDENTICAL to the same
    launches of the shipped library, whose first launch is complete and reports nothing."""
    from linearmpchumanoid_amd import build as hipbuild
    from linearmpchumanoid_amd import capi
    hipbuild.build_variant("qfault")
    good = _run_probe(_QFAULT_PROBE, "")
    bad = _run_probe(_QFAULT_PROBE, "qfault")
    assert good["build_flags"] == 0 and bad["build_flags"] == 4 | 32, (good["build_flags"], bad["build_flags"])
    assert good["sync_codes"] == [0, 0] and all(f == 0 for f in good["flags"]) and all(t == 620 for t in good["ticks"])
    assert bad["sync_codes"] == [capi.ERR_UNFINISHED, 0], bad["sync_codes"]
    lost = [i for i in range(96) if i % 7 == 3]
    n_done = 0
    for i, (f, t) in enumerate(zip(bad["flags"], bad["ticks"])):
        if f & capi.FLAG_UNFINISHED:
            assert t in (250, 500), (i, t)                       # part-way: the record of the last chunk it completed
        else:
            assert t == 620 and f == 0, (i, f, t)
            n_done += 1
    assert all(bad["flags"][i] & capi.FLAG_UNFINISHED for i in lost)
    assert n_done >= 48, n_done                                    # the queue kept working for the others
    assert bad["next_flags"] == 0 and good["next_flags"] == 0
    assert bad["next_sha"] == good["next_sha"]


def test_poison_build_really_is_the_poison_build():
    """Positive control of test_results_do_not_depend_on_uninitialised_lds: the library LMH_VARIANT=poison loads was compiled
    with -DLMH_POISON (lmh_debug_build_flags bit 0), the shipped one was not."""
    code = ("import json, os, sys; sys.path.insert(0, os.getcwd())\n"
            "from linearmpchumanoid_amd import capi\n"
            "L = capi.lib(); L.lmh_debug_build_flags.restype = int\n"
            "print(json.dumps({'flags': L.lmh_debug_build_flags(), 'so': capi.SO_PATH}))\n")
    from linearmpchumanoid_amd import build as hipbuild
    hipbuild.build_variant("poison")
    a, b = _run_probe(code, ""), _run_probe(code, "poison")
    assert a["flags"] == 0 and a["so"].endswith("liblmh_hip.so")
    assert b["flags"] == 1 and b["so"].endswith("liblmh_hip_var_poison.so")


# ------------------------------------------------------------------------------- config 2 at BASELINE's amplitude: parity of the failure
def test_config2_full_amplitude_fallers_match_the_oracle_until_either_gives_up():
    """BASELINE configs[1] as SURVEY 8d states it: pushes U(-0.3, 0.3) m/s (seed 20260001 + i).  A backward push beyond ~0.18 m/s puts the
    capture point behind the heel: the robot falls whatever the torques do (test_config2_balancers_complete_2000_ticks_without_a_flag).
    The first two such robots of the draw (v_x = -0.294, -0.298 m/s), tick by tick against the oracle from the same state
    (profiles/r04_config2_fallers.txt is the full table): tau and f agree to 1e-6 -- measured <= 3e-10 -- over the first 800 ticks, through
    joint velocities of 100 rad/s; from there the fall is a chaotic tumble (|v| > 200 rad/s, |tau| > 1e3 N m) in which round-off of either
    side doubles every few ticks, and BOTH give up within ten ticks of each other: the launch in which the GPU raises
    LMH_FLAG_NONFINITE / NOT_SPD / QP_MAXITER is the one that covers the tick at which the oracle's log stops being finite."""
    from linearmpchumanoid_amd.controller import BatchedController, default_config, ik_start_posture
    from oracle.pyoracle import Oracle
    N, md, nt, chunk = 16, 2e-2, 1200, 10
    th = N * md + 1e-9
    q0, zcom = ik_start_posture(0)
    v_full = perturbed_velocities(1024)
    idx = [i for i in range(1024) if v_full[i, 0] < -0.22][:2]
    assert idx == [16, 32]
    ctl = BatchedController(2, default_config(dt=DT, time_horizon=th, z_com=zcom, mpc_dt=md, warm_start=1))
    ctl.set_refs_stance(nt * DT + 1.0, 2)
    st = ctl.new_state(q0, v_full[idx], t=0.0)
    out, status = ctl.new_out(), ctl.new_status()
    log = torch.zeros((chunk, 2, 36), dtype=torch.float64, device=ctl.device)
    glog = np.zeros((nt, 2, 36)); gflag = np.zeros((nt // chunk, 2), dtype=np.int64)
    for c in range(nt // chunk):
        ctl.rollout(st, chunk, out, status, log)
        torch.cuda.synchronize()
        glog[c * chunk:(c + 1) * chunk] = log.cpu().numpy()
        gflag[c] = status.cpu().numpy()[:, 2]
    for j, i in enumerate(idx):
        o = Oracle(sim_time=nt * DT + 1.0, dt=md, horizon_time=th, do_ik=True)
        ref = o.rollout(np.concatenate([q0, v_full[i]]), 0.0, nt, dt=DT, log=True)["log"]
        fin = np.isfinite(ref).all(axis=1)
        assert not fin.all(), "the oracle's robot did not fall"
        t_orc = int(np.argmin(fin))
        fl = np.nonzero(gflag[:, j])[0]
        assert len(fl), "the GPU's robot did not raise a flag"
        t_gpu = int(fl[0]) * chunk
        assert t_gpu - chunk <= t_orc < t_gpu + 2 * chunk, (i, t_gpu, t_orc)
        assert t_orc > 900
        for tk in range(0, 800):
            assert close(glog[tk, j, :24], ref[tk, :24]) and close(glog[tk, j, 24:], ref[tk, 24:], scale=WEIGHT), (i, tk)
        assert max(vec_err(glog[tk, j, :24], ref[tk, :24]) for tk in range(0, 800, 7)) < 1e-8


# ------------------------------------------------------------------------------- bench.py: the evaluation-API lines
def test_bench_eval_mode_reports_both_lines():
    """`bench.py --mode eval` (secondary lines): (a) lmh_eval on every robot of the walking workload per launch, at a
    double-support and a single-support state, with the HBM roofline on SURVEY 8d's 1 008 B per evaluation; (b) B = 1 through lmh_eval_host,
    four calls per tick as the shim's Controller::standStep is driven by apps/offline/main.cpp:66-89, beside the C oracle on one core and the
    shim-built apps/offline_stand run end to end (its CoM-x trace ends where SURVEY's anchor says: -1.34e-4)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "eval", "--instances", "512", "--steps", "30", "--warmup", "5"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["unit"] == "evaluations/s" and d["secondary_line"] is True and d["config"]["mode"] == "eval"
    ph = d["by_support_phase"]
    assert ph["double_support"]["support_phase"] == 0 and ph["single_support"]["support_phase"] in (1, 2)
    assert all(v["instances_flagged"] == 0 and v["evaluations_per_s"] > 1e5 for v in ph.values())
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "lmh_eval_kernel" and rf["algorithmic_bytes_per_launch"] == 512 * 1008
    assert abs(rf["achieved"] - 512 * 1008 / (rf["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    b1 = d["b1_host_path"]
    assert 0.0 < b1["ms_per_tick"] < 50.0 and abs(b1["ms_per_evaluation"] * 4 - b1["ms_per_tick"]) < 1e-9
    assert b1["cpu_oracle_single_core"]["cores"] == 1 and b1["cpu_oracle_single_core"]["ms_per_tick"] > 0.0
    app = b1["offline_stand_app"]
    assert app["returncode"] == 0 and abs(app["last_com_x"] - (-1.34198e-4)) < 1e-8


# ------------------------------------------------------------------------------- edge-contact push-through (cone_pushthrough, round 4)
_EDGE_PROBE = r"""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from linearmpchumanoid_amd.controller import BatchedController, default_config
ik = json.load(open("tests/golden/ik_posture.json"))
B, N, NT, CH = 8, 32, 900, 50
ctl = BatchedController(B, default_config(dt=1e-3, time_horizon=N * 1e-2 + 1e-9, z_com=ik["z_com"], mpc_dt=1e-2, warm_start=1))
ctl.gen_walk(2.0, num_steps=3, time_per_step=0.5, ds_time=0.2, step_height=0.02, settle_time=0.1)
ctl.set_xscale(np.linspace(0.02, 0.05, B))
st = ctl.new_state(np.array(ik["q"]), np.zeros(30), t=0.0)
logs, masks, flags = [], [], 0
for c in range(NT // CH):
    out, status, log = ctl.rollout(st, CH, log=True)
    torch.cuda.synchronize()
    s = status.cpu().numpy()
    logs.append(log.cpu().numpy()); flags |= int(np.bitwise_or.reduce(s[:, 2]))
    masks.append([int((~int(x)) & 0xFFFFFFFF) for x in s[:, 3]])
path = sys.argv[1]
np.save(path, np.concatenate(logs, axis=0))
print(json.dumps({"flags": flags, "masks": masks}))
"""


def test_edge_contact_pushthrough_against_the_other_routes_and_the_oracle(tmp_path):
    """A foot that presses on one side of its sole (free coefficients on two vertices of one edge: K_f has rank 5) is solved by the
    push-through system in its five free wrench coordinates (cone_pushthrough) -- the set of every double support right
    after a touch-down of the walking gait.  The checker build `noedge` (-DLMH_NO_EDGE) sends the same sets down the register / general
    route: the two libraries must meet such sets (the free-set masks say so), differ in rounding only, and the shipped one must agree with
    the C oracle through the touch-downs like everywhere else."""
    from linearmpchumanoid_amd import build as hipbuild
    from oracle.pyoracle import Oracle
    hipbuild.build_variant("noedge")
    res, logs = {}, {}
    for variant in ("", "noedge"):
        path = str(tmp_path / f"edge_{variant or 'shipped'}.npy")
        code = _EDGE_PROBE.replace("sys.argv[1]", repr(path))
        res[variant] = _run_probe(code, variant)
        logs[variant] = np.load(path)
        assert res[variant]["flags"] == 0
    # the gait met edge sets: some foot's free mask lies on one side of the sole (vertices 0, 2 | 1, 3 | 0, 1 | 2, 3) with at least six members
    def edge_foot(m):
        return m != 0 and bin(m).count("1") >= 6 and any((m & ~side) == 0 for side in (0x0F0F, 0xF0F0, 0x00FF, 0xFF00))
    n_edge = sum(1 for launch in res[""]["masks"] for f in launch if edge_foot(f & 0xFFFF) or edge_foot(f >> 16))
    assert n_edge >= 8, n_edge
    a, b = logs[""], logs["noedge"]
    assert a.shape == b.shape == (900, 8, 36)
    assert not np.array_equal(a, b)                                # the routes differ in rounding, so the edge form really ran
    worst = 0.0
    for tk in range(a.shape[0]):
        for i in range(a.shape[1]):
            worst = max(worst, vec_err(a[tk, i, :24], b[tk, i, :24]), np.abs(a[tk, i, 24:] - b[tk, i, 24:]).max() / WEIGHT)
    assert worst < 1e-7, worst
    # the shipped library against the oracle on two robots, every third tick (the touch-downs of this gait fall at ticks 300 and 800)
    ik = json.load(open(os.path.join(ROOT, "tests", "golden", "ik_posture.json")))
    from linearmpchumanoid_amd.controller import BatchedController, default_config
    ctl = BatchedController(8, default_config(dt=DT, time_horizon=32 * 1e-2 + 1e-9, z_com=ik["z_com"], mpc_dt=1e-2, warm_start=1))
    ctl.gen_walk(2.0, num_steps=3, time_per_step=0.5, ds_time=0.2, step_height=0.02, settle_time=0.1)
    plan = ctl.get_refs()
    xs = np.linspace(0.02, 0.05, 8)
    for i in (0, 7):
        o = Oracle(sim_time=2.0, dt=1e-2, horizon_time=32 * 1e-2 + 1e-9, do_ik=True)
        o.set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        o.set_segments(plan["segs"], plan["seg_of_sample"], xscale=float(xs[i]))
        r = o.rollout(np.concatenate([np.array(ik["q"]), np.zeros(30)]), 0.0, 900, dt=DT, log=True)
        for tk in range(0, 900, 3):
            ref = r["log"][tk]
            assert close(a[tk, i, :24], ref[:24], 1e-6), (i, tk, vec_err(a[tk, i, :24], ref[:24]))
            assert close(a[tk, i, 24:], ref[24:], 1e-6, scale=WEIGHT), (i, tk, vec_err(a[tk, i, 24:], ref[24:]))


_EDGE_SIDES_PROBE = r"""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from linearmpchumanoid_amd.controller import BatchedController, default_config
ik = json.load(open("tests/golden/ik_posture.json"))
v = np.load(sys.argv[2])
B = v.shape[0]
ctl = BatchedController(B, default_config(dt=1e-3, time_horizon=0.32 + 1e-9, z_com=ik["z_com"], mpc_dt=1e-2, warm_start=0))
ctl.set_refs_stance(1.0, 2)
st = ctl.new_state(np.array(ik["q"]), v, t=0.0)
out, status = ctl.stand_step(st)
torch.cuda.synchronize()
s = status.cpu().numpy()
np.save(sys.argv[1], out.cpu().numpy())
print(json.dumps({"flags": s[:, 2].tolist(), "rounds": s[:, 1].tolist(), "masks": [int((~int(x)) & 0xFFFFFFFF) for x in s[:, 3]]}))
"""


def _push_velocities():
    """base velocity pushes of a standing robot in eight directions: the contact solve ends on the heel side, on either lateral side, on
    corners -- and, pushed hard backwards, on the toe side"""
    dirs = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (1, -1), (-1, 1), (-1, -1)]
    mags = [0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.4, 0.5]
    rows = [(d[0] * m, d[1] * m * 0.5) for d in dirs for m in mags] + [(-m, 0.0) for m in (0.6, 0.7, 0.8, 0.9, 1.0, 1.2, 1.5, 2.0)]
    v = np.zeros((len(rows), 30))
    v[:, 0] = [r[0] for r in rows]; v[:, 1] = [r[1] for r in rows]
    return v


def test_edge_contact_on_every_side_of_the_sole_against_the_general_route_and_the_oracle(tmp_path):
    """A standing robot pushed in eight directions (cold start, the iteration run to its end): the accepted free sets sit on the heel side
    (tau_y = -p_x f_z bound), on either lateral side (tau_x = p_y f_z bound; the walking gait only ever meets one of them), on corners and on
    mixed patterns.  The shipped library (edge-contact push-through) against the checker build `noedge` (register / general route on the same
    sets): same rounds, same final sets, results equal to rounding; the robots that end on an edge also against the C oracle."""
    from linearmpchumanoid_amd import build as hipbuild
    from oracle.pyoracle import Oracle
    hipbuild.build_variant("noedge")
    v = _push_velocities()
    vpath = str(tmp_path / "v.npy")
    np.save(vpath, v)
    outs, res = {}, {}
    for variant in ("", "noedge"):
        path = str(tmp_path / f"sides_{variant or 'shipped'}.npy")
        res[variant] = _run_probe(_EDGE_SIDES_PROBE.replace("sys.argv[1]", repr(path)).replace("sys.argv[2]", repr(vpath)), variant)
        outs[variant] = np.load(path)
    a, b = outs[""], outs["noedge"]
    assert all(f == 0 for f in res[""]["flags"]) and all(f == 0 for f in res["noedge"]["flags"])
    assert res[""]["masks"] == res["noedge"]["masks"] and res[""]["rounds"] == res["noedge"]["rounds"]
    sides = {0x0F0F: "y+", 0xF0F0: "y-", 0x00FF: "x+", 0xFF00: "x-"}
    def side_of(m):
        return next((n for sd, n in sides.items() if m != 0 and bin(m).count("1") >= 6 and (m & ~sd) == 0), None)
    seen, edge_robots = set(), []
    for i, f in enumerate(res[""]["masks"]):
        sr, sl = side_of(f & 0xFFFF), side_of(f >> 16)
        if sr or sl:
            seen.update(x for x in (sr, sl) if x); edge_robots.append(i)
    assert {"y+", "y-", "x-"} <= seen, seen                        # both lateral sides and the heel side (tau_y bound) are met
    worst = 0.0
    for i in range(a.shape[0]):
        worst = max(worst, vec_err(a[i, :24], b[i, :24]), np.abs(a[i, 24:36] - b[i, 24:36]).max() / WEIGHT, vec_err(a[i, 36:66], b[i, 36:66]))
    assert worst < 1e-7, worst
    assert any(not np.array_equal(a[i], b[i]) for i in edge_robots)
    ik = json.load(open(os.path.join(ROOT, "tests", "golden", "ik_posture.json")))
    o = Oracle(sim_time=1.0, dt=1e-2, horizon_time=0.32 + 1e-9, do_ik=True)
    for i in edge_robots[::2]:
        e = o.eval(np.array(ik["q"]), v[i], 0.0)
        assert close(a[i, :24], e["tau"]) and close(a[i, 24:36], e["f"], scale=WEIGHT) and close(a[i, 36:66], e["qpp"]), (i, hex(res[""]["masks"][i]))
