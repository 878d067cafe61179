#!/usr/bin/env python3
"""Headline benchmark: control ticks/s/node of the batched NAO WBC+MPC closed loop.

One "step" = ONE launch of the fused rollout kernel = one whole rollout segment of `--ticks` RK4 control ticks (4 controller
evaluations each) for every robot instance of this rank: 4 000 ticks (config 3 / 4: SURVEY 8d's length) or 2 000 (config 2 / 5).
The control step is dt = 1 ms; the LIPM preview is sampled at mpc_dt = 10 ms (config 2: 20 ms), so N = 32 samples preview 0.32 s and
the closed loop is stable over any number of ticks: consecutive steps CONTINUE the same rollouts (config 3: every step walks eight
more 0.5 s steps), nothing is restarted, no launch boundary sits inside a step.

Workloads (BASELINE.json configs):
  --config 3 (default at --gpus 1): configs[2], the largest single-GPU configuration: 4096 NAO instances, walking
      (footRefTrajectory swing polynomials + piecewise ZMP, contact switching DS / SS-R / SS-L, timePerStep 0.5 s with 0.2 s of double
      support), per-instance step length U(0.02, 0.05) m (seed 20260003 + i), dt = 1 ms, LIPM-MPC horizon N = 32 x 10 ms, warm-started
      WBC QP, tau | f log on.
  --config 4 (default at --gpus N > 1): configs[3], one GPU's share of it per rank: as config 3 plus per-link mass
      x U(0.9, 1.1) and CoM +- 5 mm (seed 20260004 + i), start posture per instance from the IK KERNEL, per-instance
      LIPM height; the end-of-run RCCL gather of 128-B summaries is inside the timed region.
  --config 2: configs[1]: 1024 instances, balance task with velocity pushes (SURVEY's seeds at half amplitude: inside the capture
      region of the support polygon), N = 16 x 20 ms.
  --config 5: configs[4], one GPU's share: 4096 instances, jump schedule (stance 0.4 s, flight 0.15 s, double support), N = 48 x 10 ms;
      one step = one jump from the initial state (the schedule is not periodic: restarted per step, reported).
  --coupled: round 1-2's line for continuity: mpc_dt = dt (the reference app's own choice of one value for Clock, ZMP and Mpc3dLip),
      40 (config 2: 10) ticks per step, rollouts restarted inside their ~0.5 s validity range.

`--gpus N` with no RANK in the environment starts N child processes of this script (one per GPU, before anything
touches the GPU) and relays rank 0's JSON line; under torchrun (RANK set) it is one rank.  Ranks shard instances
(weak scaling), no data-path collective.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_TICK = 1248        # state in (60 f64) + state out (60) + tau|f log (36)
ALG_FLOP_PER_TICK = 8.0e5        # nominal 2.0e5 flop per controller evaluation x 4
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s HBM3E
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X fp64 vector peak = 256 CU x 64 lanes x 2 flop x 2.4 GHz
FP64_MFMA_PEAK_TFLOPS = 78.6     # v_mfma_f64_16x16x4_f64: fp64 matrix rate = fp64 vector rate on gfx950
MFMA_MOP_FLOP = 512              # SQ_INSTS_VALU_MFMA_MOPS_F64 unit (one 16x16x4 f64 MFMA = 2048 flop = 4 MOPS)
PROFILE_TAGS = {3: "r04_c3", 2: "r04_c2"}   # profiles/<tag>_rollout_summary.json: PMC passes of the committed kernel on the default command
HARD_FLAGS = 1 | 2 | 4 | 8 | 32  # LMH_FLAG_QP_MAXITER | NONFINITE | ZMP_RANGE | NOT_SPD | UNFINISHED; LMH_FLAG_QP_FP64_ROUTE (16) is informational

# One step = `ticks` ticks in one launch.  mpc_dt: MPC sample time / reference sample period (lmh_config.mpc_dt); the preview spans
# horizon x mpc_dt = 0.32 s (0.48 s for the jump) -- with the preview tied to the 1 ms control step (16..48 ms, far below the LIPM's time
# constant sqrt(z/g) = 0.16 s) every loop diverges after ~0.5 s, which is what --coupled reproduces.  push: amplitude factor on SURVEY's
# U(-0.3, 0.3) m/s pushes (beyond 0.18 m/s backwards the capture point leaves the heel: such robots fall whatever the controller does).
# reset_every: consecutive steps CONTINUE the same rollouts (no restart inside BASELINE's 4000 / 2000-tick rollout, none in the default 2 + 8
# steps); after 10 steps = 40 s of walking the robots go back to their initial states.  The walking loop is flag-free for 90 s, but the
# robots walk away from the origin the reference's spatial velocities refer to (xdot adds omega x p_base): joint velocities grow slowly
# from 35 s on and the loop leaves its range after ~92 s / 9 m -- in the CPU oracle exactly as on the GPU (profiles/r03_long_walk.txt) --
# so a run of many steps (the driver's 5 + 20) restarts at the validated 40 s instead of running into that.
DEFAULTS = {2: dict(instances=1024, ticks=2000, horizon=16, mpc_dt=2e-2, push=0.5, reset_every=10),
            3: dict(instances=4096, ticks=4000, horizon=32, mpc_dt=1e-2, reset_every=10),
            4: dict(instances=4096, ticks=4000, horizon=32, mpc_dt=1e-2, reset_every=10),
            5: dict(instances=4096, ticks=2000, horizon=48, mpc_dt=1e-2, reset_every=1)}
WALK = dict(step_time=0.5, ds_time=0.2, settle_time=0.3)
# --coupled (rounds 1-2): mpc_dt = dt; max_ticks = the range in which that loop stays finite
COUPLED = {2: dict(instances=1024, ticks=10, horizon=16, max_ticks=230, push=1.0),
           3: dict(instances=4096, ticks=40, horizon=32, max_ticks=480),
           4: dict(instances=4096, ticks=40, horizon=32, max_ticks=480)}
COUPLED_WALK = dict(step_time=0.2, ds_time=0.05, settle_time=0.1)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)      # default 8 launches x ~1.2 s on the default workload (--mode eval: 200 launches of ~0.1 ms)
    ap.add_argument("--warmup", type=int, default=None)     # default 2 (--mode eval: 20)
    ap.add_argument("--config", type=int, default=None, choices=(2, 3, 4, 5),
                    help="BASELINE config number (1-based); default 3 at --gpus 1, 4 (randomised models + IK kernel in set-up) otherwise")
    ap.add_argument("--coupled", action="store_true", help="rounds 1-2's line: mpc_dt = dt, short steps, rollouts restarted inside their validity range")
    ap.add_argument("--instances", type=int, default=None, help="robot instances per GPU")
    ap.add_argument("--ticks", type=int, default=None, help="RK4 ticks per step (per launch)")
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--mpc-dt", type=float, default=None, help="MPC sample time (lmh_config.mpc_dt); default per config, dt with --coupled")
    ap.add_argument("--cold", action="store_true", help="cold-start the QP active set every evaluation")
    ap.add_argument("--no-log", action="store_true", help="do not write the per-tick tau|f log")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg (all of its lines together)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path)")
    ap.add_argument("--reset-every", type=int, default=None,
                    help="restart the rollouts from the initial states after this many steps (0 = never: consecutive steps continue the same "
                         "rollouts); default: 0 for configs 2-4, 1 for the jump of config 5, the validity range with --coupled")
    ap.add_argument("--summary-out", default=None, help="write the gathered end-of-run summary (lmh_write_summary format) to this path")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch from rocprofv3 --pmc; default: the committed profiles/ summary when the workload matches it")
    ap.add_argument("--step-time", type=float, default=None, help="walking: time per step (double + single support) [s]")
    ap.add_argument("--ds-time", type=float, default=None, help="walking: double-support share of a step [s]")
    ap.add_argument("--settle-time", type=float, default=None, help="walking: stance before the first step [s]")
    ap.add_argument("--push", type=float, default=None, help="config 2: amplitude factor on SURVEY's U(-0.3, 0.3) m/s pushes")
    ap.add_argument("--precision", type=int, default=0, choices=(0, 1, 2), help="lmh_config.precision (0 fp64, 1 mixed, 2 fp32)")
    ap.add_argument("--max-qp-iters", type=int, default=None, help="diagnostic: lmh_config.max_qp_iters (a low cap raises LMH_FLAG_QP_MAXITER: exercises the flag accounting / exit code 3)")
    ap.add_argument("--mode", default="rollout", choices=("rollout", "eval"),
                    help="rollout (default, the headline): fused closed loop; eval: the SECONDARY lines of the evaluation API -- lmh_eval "
                         "(= Controller::standStep + WBC, src/controller.cpp:48-154) for all robots per control step, and B = 1 through "
                         "lmh_eval_host as the shim's Controller drives it from apps/offline/main.cpp:66-89")
    ap.add_argument("--host-io", action="store_true",
                    help="informational: every step also moves the state host->device and out | status | log device->host through pinned "
                         "buffers (what a caller holding HOST buffers pays over PCIe); never the default line")
    args = ap.parse_args(argv)
    if args.config is None:
        args.config = 3 if args.gpus == 1 else 4
    if args.steps is None:
        args.steps = 200 if args.mode == "eval" else 8
    if args.warmup is None:
        args.warmup = 20 if args.mode == "eval" else 2
    if args.coupled and args.config == 5:
        ap.error("--coupled has no config 5 line")
    d = dict((COUPLED if args.coupled else DEFAULTS)[args.config])
    for k in ("instances", "ticks", "horizon"):
        if getattr(args, k) is None:
            setattr(args, k, d[k])
    if args.mpc_dt is None:
        args.mpc_dt = args.dt if args.coupled else d["mpc_dt"]
    if args.push is None:
        args.push = d.get("push", 1.0)
    w = COUPLED_WALK if args.coupled else WALK
    for k in ("step_time", "ds_time", "settle_time"):
        if getattr(args, k) is None:
            setattr(args, k, w[k])
    if args.reset_every is None:
        args.reset_every = max(1, d["max_ticks"] // args.ticks) if args.coupled else d["reset_every"]
    return args


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """--gpus N without a launcher: start N fresh processes of this script (this parent has made no GPU / HIP call
    and makes none), wait for them, exit non-zero if any rank failed.  Rank 0's JSON line goes to the inherited stdout."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            r = p.poll()
            if r is None:
                continue
            pending.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in pending:                                  # a failed rank leaves the others in a collective: stop exactly those PIDs
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc if rc >= 0 else 1)


# ------------------------------------------------------------------------------------------------ workload
def perturbed_velocities(first, count, seed=20260001):
    import numpy as np
    v = np.zeros((count, 30))
    for i in range(count):
        rng = np.random.default_rng(seed + first + i)
        v[i, 0:2] = rng.uniform(-0.3, 0.3, 2)
        v[i, 6:] = rng.normal(0.0, 0.05, 24)
    return v


def step_lengths(first, count, seed=20260003):
    import numpy as np
    return np.array([np.random.default_rng(seed + first + i).uniform(0.02, 0.05) for i in range(count)])


def randomised_links(first, count, seed=20260004):
    """per-link mass x U(0.9,1.1), CoM + U(-5,5) mm per axis, on createNaoParameters() output."""
    import numpy as np
    from linearmpchumanoid_amd.controller import nominal_links
    raw = np.tile(nominal_links(), (count, 1, 1))
    for i in range(count):
        rng = np.random.default_rng(seed + first + i)
        raw[i, :, 0] *= rng.uniform(0.9, 1.1, 28)
        raw[i, :, 1:4] += rng.uniform(-5e-3, 5e-3, (28, 3)) * (raw[i, :, 0:1] > 0)
    return raw


def build_workload(args, ctl, first, count, total_ticks):
    """Uploads models / references for this rank's instances; returns (state0 tensor, dict of host-side inputs the
    CPU baseline replays).  Reference arrays are sampled at the MPC sample time (args.mpc_dt)."""
    import numpy as np
    import torch
    from linearmpchumanoid_amd import trajectories
    from linearmpchumanoid_amd.controller import ik_start_posture, initial_configuration
    dev = ctl.device_index
    host = {}
    sim_time = total_ticks * args.dt + 1.0                         # the preview window [k, k + N] of the last tick stays inside the arrays
    if args.config in (2, 5):
        q0, zcom = ik_start_posture(dev)
        ctl.set_zcom(np.array([zcom]))
        if args.config == 2:
            ctl.set_refs_stance(sim_time, 2)
            v = args.push * perturbed_velocities(first, count)
        else:                                                      # DS 0.4 s -> flight 0.15 s -> DS, small pushes
            ctl.gen_jump(sim_time, 0.4, 0.15)
            v = 0.2 * perturbed_velocities(first, count, seed=20260005)
        plan = ctl.get_refs()
        state = ctl.new_state(q0, v, t=0.0)
        host.update(q0=np.tile(q0, (count, 1)), v=v, zcom=np.array([zcom]), zmp_x=plan["zmp_x"], zmp_y=plan["zmp_y"],
                    phase=plan["phase"] if args.config == 5 else None, segs=None, sos=None, xscale=None, raw=None)
        return state, host
    n_steps = max(2, int((sim_time - args.settle_time) / args.step_time))
    # the walking plan (ZMP samples, support phase, swing-foot polynomial segments) is generated ON THE DEVICE (lmh_gen_walk); it is read
    # back only so that the CPU baseline leg replays exactly the same references
    ctl.gen_walk(sim_time, num_steps=n_steps, time_per_step=args.step_time, ds_time=args.ds_time, step_height=0.02, settle_time=args.settle_time)
    plan = ctl.get_refs()
    xs = step_lengths(first, count)
    raw = None
    if args.config == 4:
        raw = randomised_links(first, count)
        ctl.set_model(raw)
        q = torch.as_tensor(np.tile(initial_configuration(), (count, 1))).to(ctl.device)
        q, iters = ctl.ik(q)                                       # Kinematics::compute per instance on its own model
        com = ctl.robot_com(q)
        torch.cuda.synchronize(ctl.device)
        q0s, zc = q.cpu().numpy(), com.cpu().numpy()[:, 2].copy()
        if int(iters.max().item()) > 12:
            raise RuntimeError("IK kernel did not converge on a randomised model")
        ctl.set_zcom(zc)
    else:
        q0, zcom = ik_start_posture(dev)
        q0s, zc = np.tile(q0, (count, 1)), np.array([zcom])
        ctl.set_zcom(zc)
    ctl.set_xscale(xs)
    state = ctl.new_state(q0s, np.zeros(30), t=0.0)
    host.update(q0=q0s, v=np.zeros((count, 30)), zcom=zc, zmp_x=plan["zmp_x"], zmp_y=plan["zmp_y"], phase=plan["phase"],
                segs=plan["segs"], sos=plan["seg_of_sample"], xscale=xs, raw=raw)
    return state, host


_LOOP = "dt={dt} (control), LIPM-MPC horizon N={N} x mpc_dt={md}, WBC QP per evaluation, RK4 closed loop"
WORKLOAD_TEXT = {
    2: "{B} NAO instances/GPU, balance task (IK posture + velocity pushes {push} x U(-0.3,0.3) m/s), " + _LOOP,
    3: "{B} NAO instances/GPU, walking (footRefTrajectory swing polynomials + piecewise ZMP, contact switching DS/SS-R/SS-L, timePerStep {st} s, per-instance step length U(0.02,0.05) m), " + _LOOP,
    4: "{B} NAO instances/GPU, walking with contact switching (timePerStep {st} s), domain-randomised link mass/CoM, per-instance IK start posture (IK kernel in set-up) and LIPM height, " + _LOOP + ", end-of-run summary gather in the timed region",
    5: "{B} NAO instances/GPU, jump schedule (double support 0.4 s, flight 0.15 s, double support), " + _LOOP,
}


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    usable = logical
    try:
        usable = min(usable, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:                                                           # cgroup v2 CPU quota of the box's share
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()
            if a != "max":
                quota = float(a) / float(b)
    except (OSError, ValueError):
        pass
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    return model, logical, usable, quota


def _native_oracle():
    """Second CPU line of BASELINE.md section 2: the same oracle sources built -O3 -march=native on THIS host."""
    import ctypes as C
    import tempfile
    odir = os.path.join(ROOT, "oracle")
    srcs = [os.path.join(odir, f) for f in ("orc_robot.c", "orc_dynamics.c", "orc_mpc.c", "orc_qp.c", "orc_controller.c", "orc_api.c")]
    out = os.path.join(tempfile.mkdtemp(prefix="lmh_orc_native_"), "liblmh_oracle_native.so")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-shared", "-o", out] + srcs + ["-lm", "-lpthread"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return C.CDLL(out)


def cpu_baseline(args, host, step_ticks, gpu_replay=None):
    """Reference CPU path = the C oracle (restatement of the reference, dense cold-start active-set QP per evaluation), timed on
    this host's cores on a bounded sample of the SAME workload: the first n robots of rank 0 over ticks 0..T of the first step
    (T = a whole step when the budget allows, otherwise its first T ticks), static partition over threads.  Lines: one core; all
    usable cores (the reported value); all cores with the reference's literal duplicate WBC call + per-call MPC Hessian rebuild
    (apps/offline/main.cpp:103-105, mpcLinearPendulum.cpp:89-90); all cores with -O3 -march=native.
    gpu_replay(n_inst, nticks) -> [n_inst, 36]: the GPU's k4-stage tau | f of tick nticks - 1 for the same robots from the same
    initial states (untimed), compared with the CPU leg's."""
    import numpy as np
    from oracle import pyoracle
    model, logical, usable, quota = host_cpu_info()
    th = args.horizon * args.mpc_dt + 1e-9
    B = host["q0"].shape[0]

    def run(idx, nticks, nthreads, wbc_calls=1, lib=None):
        st = np.concatenate([host["q0"][idx], host["v"][idx]], axis=1)
        zc = host["zcom"] if len(host["zcom"]) == 1 else host["zcom"][idx]
        return pyoracle.batch_rollout_ex(st, 0.0, args.dt, nticks, th, host["zmp_x"], host["zmp_y"], host["phase"], host["segs"], host["sos"],
                                         None if host["xscale"] is None else host["xscale"][idx], zc,
                                         None if host["raw"] is None else host["raw"][idx], nthreads=nthreads, wbc_calls=wbc_calls, lib_override=lib,
                                         mpc_dt=args.mpc_dt)

    sec, _, _ = run(np.arange(1), 20, 1)                         # calibrate: 1 robot x 20 ticks on one core
    per_tick = sec / 20
    budget = args.cpu_seconds / 4.0                                # four lines
    per_robot = per_tick * step_ticks
    if per_robot <= budget:                                        # a whole step per robot, as many robots as the budget allows
        r = max(1, int(budget / per_robot))
        n_inst, nticks = min(B, usable * r), step_ticks
    else:                                                          # step too long for the budget: one robot per thread, its first ticks
        n_inst, nticks = min(B, usable), max(20, int(budget / per_tick))
    idx = np.arange(n_inst)
    n1 = max(20, min(nticks, int(budget / per_tick)))
    sec1, _, _ = run(np.arange(1), n1, 1)
    secN, _, outN = run(idx, nticks, usable)
    allc = n_inst * nticks / secN
    res = {"value": allc, "unit": "control ticks/s", "cores": usable, "kind": "port",
           "single_core_value": n1 / sec1, "cpu_model": model, "logical_cpus": logical, "cgroup_cpu_quota": quota,
           "compiler_flags": "-O2 (reference's own CMake: -O1 -g + ASan/UBSan)",
           "seconds": secN,
           "sample": f"C oracle, first {n_inst} robots of the same workload x ticks 0..{nticks} of the {step_ticks}-tick step the GPU runs, {usable} threads (static partition), "
                     f"1 WBC solve per evaluation, cold-start dense active-set QP; single core: 1 robot x {n1} ticks"}
    if gpu_replay is not None:                                     # same robots, same tick count: the k4-stage tau|f of the last tick must agree
        g = gpu_replay(n_inst, nticks)
        ref = outN
        err = float(np.max(np.abs(g[:n_inst, :36] - ref) / np.maximum(1e-9 * np.abs(ref).max(), np.abs(ref).max(axis=1, keepdims=True))))
        res["parity_vs_gpu_last_tick_max_rel"] = err
    try:
        sec2, _, _ = run(idx, nticks, usable, wbc_calls=2)
        res["literal_2wbc_value"] = n_inst * nticks / sec2
        res["literal_2wbc_note"] = "duplicate WBC(t) per evaluation + MPC Hessian rebuilt per call, as apps/offline/main.cpp:103-105 / mpcLinearPendulum.cpp:89-90 do"
    except Exception as e:
        res["literal_2wbc_value"] = None
        res["literal_2wbc_note"] = f"failed: {e}"
    try:
        nat = _native_oracle()
        sec3, _, _ = run(idx, nticks, usable, lib=nat)
        res["native_O3_value"] = n_inst * nticks / sec3
    except Exception as e:
        res["native_O3_value"] = None
        res["native_O3_note"] = f"failed: {e}"
    return res


def profiled_pmc(args):
    """PMC figures from the committed rocprofv3 passes (profiles/<tag>_rollout_summary.json, made by scripts/profile_rollout.sh +
    scripts/summarise_profile.py on this same default command); only valid for the workload they were collected on.
    Returns dict(traffic = HBM bytes per launch, mops = MFMA f64 MOPS per evaluation, flop_eval = counted fp64 flop per evaluation
    (VALU FMA/ADD/MUL/TRANS classes x 64 lanes + MFMA), tag) or None."""
    tag = PROFILE_TAGS.get(args.config)
    d = DEFAULTS[args.config]
    if (tag is None or args.coupled or (args.instances, args.ticks, args.horizon, args.mpc_dt) != (d["instances"], d["ticks"], d["horizon"], d["mpc_dt"])
            or args.cold or args.precision or args.no_log):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", tag + "_rollout_summary.json")) as f:
            dd = json.load(f)["derived"]
        return dict(traffic=dd["hbm_write_bytes_per_launch"] + dd["hbm_fetch_bytes_per_launch_x2_gfx950_correction"],
                    mops=dd["mfma_f64_mops_per_eval"], flop_eval=dd.get("counted_fp64_flop_per_eval"),
                    flop_note=dd.get("counted_fp64_flop_note"), tag=tag)
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------ evaluation API (secondary lines)
ALG_BYTES_PER_EVAL = 1008        # SURVEY 8d, eval-API mode: 60 f64 in + 66 f64 out per instance per evaluation


def main_eval(args, ctl, state, host, count):
    """bench.py --mode eval (N = 1 only).  (a) lmh_eval on all `count` robots of the configured workload: one step = one launch = one
    controller evaluation per robot, at a double-support and at a single-support state of the walk (reached by the rollout kernel,
    untimed); evaluations/s, HBM roofline on SURVEY 8d's 1 008 B per evaluation, the per-launch time.  (b) B = 1 through lmh_eval_host
    -- host buffers in, H2D, kernel, D2H, synchronous -- four calls per tick exactly as the shim's Controller::standStep is driven by
    rk4Step in apps/offline/main.cpp:66-89 at the reference's literals (dt = 0.01, N = 50), beside the C oracle on one core."""
    import ctypes as C
    import numpy as np
    import torch
    from linearmpchumanoid_amd import capi
    from linearmpchumanoid_amd.controller import BatchedController, default_config, ik_start_posture
    out, status = ctl.new_out(), ctl.new_status()
    lines = {}
    pos = 0
    for name, upto in (("double_support", 400), ("single_support", 700)):       # WALK: settle 0.3 s, DS 0.2 s, SS 0.3 s
        ctl.rollout(state, upto - pos, out, status)
        pos = upto
        st = state.clone()
        for _ in range(args.warmup):
            ctl.stand_step(st, out, status)
        torch.cuda.synchronize()
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            ctl.stand_step(st, out, status)                        # the same instant again: v_prev and the warm start carry over, like rk4's stages 2 | 3
        ev1.record()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        kms = ev0.elapsed_time(ev1) / args.steps
        flagged = int(((status[:, 2] & HARD_FLAGS) != 0).sum().item())
        lines[name] = dict(evaluations_per_s=count * args.steps / el, launch_ms=kms, wall_ms_per_launch=el / args.steps * 1e3,
                           support_phase=int(host["phase"][int(status[0, 0].item())]) if host.get("phase") is not None else 0, instances_flagged=flagged)
    ss = lines["single_support"]
    hbm_g = count * ALG_BYTES_PER_EVAL / (ss["launch_ms"] * 1e-3) / 1e9
    fl = count * ALG_FLOP_PER_TICK / 4 / (ss["launch_ms"] * 1e-3) / 1e12
    # (b) B = 1, host buffers, reference literals
    q0, zcom = ik_start_posture(ctl.device_index)
    c1 = BatchedController(1, default_config(dt=0.01, time_horizon=0.5, z_com=zcom), device=ctl.device_index)
    c1.set_refs_stance(5.0, 2)
    L = capi.lib()
    q = np.ascontiguousarray(q0); dq = np.zeros(30); tau = np.zeros(24); f = np.zeros(12); qdd = np.zeros(30); stt = np.zeros(4, np.int32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    n_ticks = 200
    def tick(t):
        for ts in (t, t + 0.005, t + 0.005, t + 0.01):             # rk4.hpp:12-15: four evaluations per tick (the state advance itself is ~100 flops on the host)
            capi.check(L.lmh_eval_host(c1._h, ptr(q), ptr(dq), C.c_double(ts), ptr(tau), ptr(f), ptr(qdd), ptr(stt)))
    for i in range(20):
        tick(0.01 * i)
    t0 = time.perf_counter()
    for i in range(n_ticks):
        tick(0.01 * i)
    b1 = (time.perf_counter() - t0) / n_ticks
    res_b1 = {"ms_per_tick": b1 * 1e3, "ms_per_evaluation": b1 * 250.0, "ticks_per_s": 1.0 / b1,
              "path": "lmh_eval_host, B = 1: memcpy of q | dq into the staged record, hipMemcpy H2D, lmh_eval_kernel (one workgroup), three hipMemcpy D2H (state, out, status), "
                      "all synchronous -- what the shim's Controller::standStep costs per call (apps/offline/main.cpp:66-89 through csrc/shim)",
              "config": "reference literals: dt = 0.01, N = 50, stance, IK posture"}
    app = os.path.join(ROOT, "apps", "offline_stand")
    if os.path.exists(app):                                        # the whole app, process start / lmh_create / IK included: 5 s of stand = 500 ticks
        t0 = time.perf_counter()
        r = subprocess.run([app, "5", "0.01", "0.5"], capture_output=True, text=True)
        res_b1["offline_stand_app"] = {"wall_s": time.perf_counter() - t0, "ticks": len(r.stdout.split()), "returncode": r.returncode,
                                       "last_com_x": float(r.stdout.split()[-1]) if r.stdout.split() else None}
    if not args.no_cpu_baseline:
        from oracle.pyoracle import Oracle
        o = Oracle(sim_time=5.0, dt=0.01, horizon_time=0.5, do_ik=True)
        x0 = np.concatenate([o.robot()["q"], np.zeros(30)])
        o.rollout(x0, 0.0, 20)
        t0 = time.perf_counter()
        ro = o.rollout(x0, 0.0, n_ticks)
        ob = (time.perf_counter() - t0) / n_ticks
        model, logical, usable, quota = host_cpu_info()
        res_b1["cpu_oracle_single_core"] = {"ms_per_tick": ob * 1e3, "ticks_per_s": 1.0 / ob, "cores": 1, "kind": "port", "cpu_model": model,
                                            "sample": f"C oracle, 1 robot x {n_ticks} ticks of the same stand at the reference literals (one WBC solve per evaluation)",
                                            "last_com_x": float(ro["comx"][-1])}
    res = {
        "metric": "controller evaluations/s (lmh_eval: batched Controller::standStep + WBC)",
        "value": ss["evaluations_per_s"], "unit": "evaluations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ss["wall_ms_per_launch"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "secondary_line": True,
        "config": {"workload": WORKLOAD_TEXT[args.config].format(B=count, dt=args.dt, N=args.horizon, md=args.mpc_dt, push=args.push, st=args.step_time)
                               + "; one step = ONE lmh_eval launch (one controller evaluation per robot) at the single-support state reached after 700 ticks",
                   "baseline_config": args.config, "instances_per_gpu": count, "mode": "eval"},
        "by_support_phase": lines,
        "roofline": {"bound": "hbm", "achieved": hbm_g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_g / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "lmh_eval_kernel", "kernel_ms": ss["launch_ms"], "units_per_launch": count, "algorithmic_bytes_per_launch": count * ALG_BYTES_PER_EVAL,
                     "note": "SURVEY 8d's eval-API figure (1 008 B per evaluation) on the HBM axis as asked; the kernel is bound by fp64 issue + LDS latency like the rollout, "
                             "and every launch also re-reads the robot's model / tables (3.1 KB + 1.8 KB, L2-resident) and rebuilds the LDS image the rollout keeps across 250 ticks",
                     "fp64_valu": {"achieved": fl, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / FP64_VALU_PEAK_TFLOPS, "nominal_flop_per_evaluation": ALG_FLOP_PER_TICK / 4}},
        "b1_host_path": res_b1,
    }
    print(json.dumps(res), flush=True)
    sys.exit(3 if any(v["instances_flagged"] for v in lines.values()) else 0)


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and "RANK" not in os.environ:
        if args.gpus > 1:
            launch_ranks(args.gpus)                                # never returns
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(env_world or "1")
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist
    from linearmpchumanoid_amd.controller import BatchedController, default_config
    from linearmpchumanoid_amd import sharding

    if os.environ.get("LMH_BENCH_DEVICE") is not None:          # rehearsal of N>1 on a one-GPU box (gloo): all ranks share a card
        local_rank = int(os.environ["LMH_BENCH_DEVICE"])
    n_dev = torch.cuda.device_count()                             # (counting devices does not initialise the GPU)
    if local_rank >= n_dev:
        sys.exit(f"bench.py: LOCAL_RANK={local_rank} but only {n_dev} GPU(s) are visible to this process: one rank per GPU of ONE node "
                 f"(--gpus N needs N visible devices; there is no CPU fallback for the product path)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)                             # before the process group: RCCL binds its communicator to this device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", local_rank)} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)

    B = args.instances
    first, count = sharding.shard_range(B * world, world, rank)
    th = args.horizon * args.mpc_dt + 1e-9                         # int(th / mpc_dt) == N whatever the rounding of the quotient
    cfg = default_config(dt=args.dt, time_horizon=th, z_com=0.26, mpc_dt=0.0 if args.coupled else args.mpc_dt,
                         warm_start=0 if args.cold else 1, precision=args.precision)
    if args.max_qp_iters is not None:
        cfg.max_qp_iters = args.max_qp_iters
    ctl = BatchedController(count, cfg, device=local_rank)
    assert ctl.N == args.horizon, (ctl.N, args.horizon)
    n_launch = (args.warmup + args.steps) if args.mode == "rollout" else 1
    reset_every = max(0, args.reset_every)                         # 0: consecutive steps continue the same rollouts
    total_ticks = (min(n_launch, reset_every) if reset_every else n_launch) * args.ticks
    state, host = build_workload(args, ctl, first, count, total_ticks)
    if args.mode == "eval":
        if world != 1 or args.config not in (3, 4):
            sys.exit("bench.py --mode eval: one GPU, a walking configuration (--config 3 or 4)")
        main_eval(args, ctl, state, host, count)                   # never returns
    state0 = state.clone()
    out, status = ctl.new_out(), ctl.new_status()
    log = None if args.no_log else torch.zeros((args.ticks, count, 36), dtype=torch.float64, device=ctl.device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctl.synchronize()                                          # lmh_synchronize: an incomplete launch (LMH_ERR_UNFINISHED) is an error here, not a number

    done = 0
    flags_acc = torch.zeros((count,), dtype=torch.int32, device=ctl.device)   # status flags OR-ed over EVERY launch (the kernel overwrites status[:, 2] per launch)

    pinned = None
    if args.host_io:
        pinned = dict(state=state.cpu().pin_memory(), out=torch.empty(out.shape, dtype=out.dtype).pin_memory(),
                    status=torch.empty(status.shape, dtype=status.dtype).pin_memory(),
                    log=None if log is None else torch.empty(log.shape, dtype=log.dtype).pin_memory())

    def step():
        nonlocal done
        if reset_every and done and done % reset_every == 0:
            state.copy_(state0)                                   # device-to-device, inside the timed region when it happens
            status.zero_()
            if pinned is not None:
                pinned["state"].copy_(state0, non_blocking=True)
        if pinned is not None:
            state.copy_(pinned["state"], non_blocking=True)         # the caller's state arrives over PCIe ...
        ctl.rollout(state, args.ticks, out, status, log)
        flags_acc.bitwise_or_(status[:, 2])                       # same stream, no sync
        if pinned is not None:                                       # ... and everything the rollout produced goes back
            pinned["state"].copy_(state, non_blocking=True); pinned["out"].copy_(out, non_blocking=True)
            pinned["status"].copy_(status, non_blocking=True)
            if log is not None:
                pinned["log"].copy_(log, non_blocking=True)
        done += 1

    for _ in range(args.warmup):
        step()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    # end-of-run summary: 16 f64 per instance, gathered over RCCL (the only collective) -- inside the timed region
    summary = sharding.make_summary(state, out, status, ctl)
    if world > 1 and args.backend != "nccl":
        summary = summary.cpu()
    gathered = sharding.gather_summaries(summary, world, rank)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps              # HIP events on the launch stream (torch's current stream = the stream lmh_rollout is given)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctl.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    hard = int(((flags_acc & HARD_FLAGS) != 0).sum().item())
    routed = int(((flags_acc & 16) != 0).sum().item())
    hard_last = int(((status[:, 2] & HARD_FLAGS) != 0).sum().item())
    if world > 1:
        ft = torch.tensor([hard, routed, hard_last], dtype=torch.int64, device=ctl.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        hard, routed, hard_last = int(ft[0].item()), int(ft[1].item()), int(ft[2].item())
    if rank == 0 and args.summary_out and gathered is not None:
        ctl.write_summary(args.summary_out, gathered.cpu().numpy(), dt=args.dt)

    rc = 0
    if rank == 0:
        total_instances = B * world
        ticks_total = total_instances * args.ticks * args.steps
        value = ticks_total / elapsed
        launch_s = kernel_ms * 1e-3
        units = count * args.ticks                                 # robot-ticks one launch processes
        alg_bytes = units * (ALG_BYTES_PER_TICK if log is not None else 960)
        alg_flop = units * ALG_FLOP_PER_TICK
        pmc = profiled_pmc(args)
        traffic = args.traffic if args.traffic is not None else (pmc["traffic"] if pmc else None)
        fp64_t = alg_flop / launch_s / 1e12
        hbm_g = alg_bytes / launch_s / 1e9
        mfma = counted = None
        if pmc is not None:
            mt = units * 4 * pmc["mops"] * MFMA_MOP_FLOP / launch_s / 1e12
            mfma = {"achieved_tflops": mt, "peak_tflops": FP64_MFMA_PEAK_TFLOPS, "frac": mt / FP64_MFMA_PEAK_TFLOPS,
                    "mfma_f64_mops_per_eval": pmc["mops"], "source": f"SQ_INSTS_VALU_MFMA_MOPS_F64 of the committed profile profiles/{pmc['tag']}_rollout_summary.json x {MFMA_MOP_FLOP} flop"}
            if pmc.get("flop_eval"):
                ct = units * 4 * pmc["flop_eval"] / launch_s / 1e12
                counted = {"counted_flop_per_tick": 4 * pmc["flop_eval"], "achieved": ct, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ct / FP64_VALU_PEAK_TFLOPS, "note": pmc.get("flop_note"),
                           "source": f"profiles/{pmc['tag']}_rollout_summary.json (rocprofv3 --pmc passes of this command)"}
        restarts = ((n_launch - 1) // reset_every) if reset_every else 0
        res = {
            "metric": "control ticks/s/node (batched NAO WBC+MPC @1kHz)",
            "value": value, "unit": "control ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {0: "f64", 1: "f64 (QP) / f32 (model terms)", 2: "f32 (model terms + QP), f64 state / references"}[args.precision],
            "data": "synthetic",
            "config": {"workload": WORKLOAD_TEXT[args.config].format(B=B, dt=args.dt, N=args.horizon, md=args.mpc_dt, push=args.push, st=args.step_time),
                       "baseline_config": args.config, "coupled_mpc_dt": bool(args.coupled),
                       "instances_per_gpu": B, "ticks_per_step": args.ticks, "evaluations_per_tick": 4, "mpc_dt": args.mpc_dt, "preview_s": args.horizon * args.mpc_dt,
                       "tick_range": [args.warmup * args.ticks, n_launch * args.ticks] if not restarts
                       else f"ticks 0..{reset_every * args.ticks} of every rollout, restarted from the initial states every {reset_every} step(s)",
                       "qp_start": "cold" if args.cold else "warm", "log": log is not None, "host_io_over_pcie": bool(args.host_io), "parallelism": f"instances sharded x{world}",
                       "rollout_restarts": restarts, "summary_gather_in_timed_region": True},
            "evaluations_per_s": value * 4,
            "timed_region_s": elapsed,
            "roofline": {"bound": "fp64-valu", "achieved": fp64_t, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp64_t / FP64_VALU_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": None if traffic is None else ("--traffic" if args.traffic is not None else f"committed profile profiles/{pmc['tag']}_rollout_summary.json (WRITE_SIZE + 2 x FETCH_SIZE per launch)"),
                         "kernel": "lmh_rollout_kernel", "kernel_ms": kernel_ms, "units_per_launch": units,
                         "algorithmic_flop_per_launch": alg_flop, "algorithmic_bytes_per_launch": alg_bytes,
                         "nominal_flop_per_tick": ALG_FLOP_PER_TICK,
                         "counted_flop_per_tick": None if counted is None else counted["counted_flop_per_tick"],
                         "frac_counted": None if counted is None else counted["frac"],
                         "counted": counted,
                         "note": "binding resource: fp64 vector issue + LDS latency (SURVEY 8d); `frac` prices SURVEY's nominal 8.0e5 flop per tick, `frac_counted` the fp64 operations "
                                 "the kernel actually issues (PMC instruction classes of the committed profile)",
                         "nominal_flop_skipped_by_design": "of SURVEY's nominal 2.0e5 flop per evaluation the kernel does not perform: tau = M a + C - J'w at three of the four RK4 stages "
                                 "(the integrator never reads it; k4 only: ~2.6e3 flop per skipped evaluation), the angular-momentum rows of AG / AGpqp / h (weight 0 in the reference: ~3e3 flop), "
                                 "the MPC Hessian / spatial-inertia / friction-block rebuilds of quirk A8 (constants), and SURVEY's nominal 15 working-set changes of the contact QP "
                                 "(~7e4 flop; a warm-started exact solve takes 1 round) -- which is why frac (nominal) and frac_counted (issued) are reported side by side",
                         "hbm": {"bound": "hbm", "achieved": hbm_g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_g / HBM_PEAK_GBS},
                         "mfma": mfma},
            "instances_flagged": hard,
            "flags_accumulated_over": f"all {n_launch} launches (warm-up included)", "instances_flagged_in_last_launch": hard_last,
            "instances_fp64_route": routed,
            "summary_rows_gathered": int(gathered.shape[0]) if gathered is not None else 0,
        }
        if not args.no_cpu_baseline and world == 1:              # the CPU baseline is timed on rank 0 at N = 1 only
            try:
                def gpu_replay(n_inst, nticks):                   # parity sample for the CPU leg: the same robots from the initial states (untimed)
                    state.copy_(state0); status.zero_()
                    ctl.rollout(state, nticks, out, status, log)
                    torch.cuda.synchronize()
                    return out.cpu().numpy()
                res["cpu_baseline"] = cpu_baseline(args, host, args.ticks, gpu_replay)
                if res["cpu_baseline"].get("value"):
                    res["vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is reported, never required for the GPU number
                res["cpu_baseline"] = {"value": None, "unit": "control ticks/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(res), flush=True)
        if hard:
            print(f"bench.py: {hard} instances raised a status flag in some launch of the benchmarked range", file=sys.stderr)
            rc = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
