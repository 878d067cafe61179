#!/usr/bin/env python3
"""BASELINE config-5 tolerance sweep on SURVEY 8d's schedule: 0.4 s double support -> 0.15 s flight -> double support (N = 48, dt = 1 ms,
through the landing), run in fp64, mixed (fp32 model terms, fp64 QP) and fp32 (model terms + QP in fp32) arithmetic.

Two views per reduced-precision mode, both against the fp64 run (which matches the CPU oracle to < 1e-6, tests/test_gpu_round2.py):
  * "evaluation": every SAMPLE-th tick the fp64 run's state is evaluated once in the mode (lmh_eval on the same state): the arithmetic
    error of one controller evaluation, no drift;
  * "closed_loop": the mode's own rollout, its logged tau / f against the fp64 log tick by tick (arithmetic error + closed-loop drift).
Relative error of a vector = max |a - b| / max |b|.  Pass rates at 1e-6 .. 1e-2.  Usage (GPU box): python scripts/precision_sweep.py [B] [ticks] [out.json]"""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from linearmpchumanoid_amd.controller import BatchedController, default_config, ik_start_posture
from linearmpchumanoid_amd import trajectories, capi
from helpers import perturbed_velocities

TOLS = (1e-6, 1e-5, 1e-4, 1e-3, 1e-2)
MODES = (("fp64", capi.PRECISION_FP64), ("mixed", capi.PRECISION_MIXED), ("fp32", capi.PRECISION_FP32))


def rel(a, b):        # per row: max-abs error relative to the reference vector's max-abs entry
    return np.abs(a - b).max(axis=-1) / np.maximum(np.abs(b).max(axis=-1), 1e-300)


def dist(e):
    e = np.asarray(e).ravel()
    return {"max": float(e.max()), "p50": float(np.percentile(e, 50)), "p99": float(np.percentile(e, 99)),
            "pass_rate": {f"{tol:g}": float((e <= tol).mean()) for tol in TOLS}}


def sweep(B=1024, nt=600, chunk=10, stance_time=0.4, flight_time=0.15, seed=20260005, vscale=0.1):
    dt, N = 1e-3, 48
    th = N * dt
    q0, zcom = ik_start_posture(0)
    plan = trajectories.jump_plan(nt * dt + 0.5, dt, stance_time=stance_time, flight_time=flight_time)
    v = perturbed_velocities(B, seed=seed) * vscale
    ctl, st, status = {}, {}, {}
    for name, prec in MODES:
        ctl[name] = BatchedController(B, default_config(dt=dt, time_horizon=th, z_com=zcom, warm_start=1, precision=prec))
        ctl[name].set_refs(plan["zmp_x"], plan["zmp_y"], plan["phase"])
        st[name] = ctl[name].new_state(q0, v, t=0.0)
        status[name] = ctl[name].new_status()
    out = {n: ctl[n].new_out() for n, _ in MODES}
    ev = {n: {"tau": [], "f": []} for n in ("mixed", "fp32")}
    cl = {n: {"tau": [], "f": []} for n in ("mixed", "fp32")}
    flags = {n: np.zeros(B, dtype=np.int64) for n, _ in MODES}
    fp64_route = []                       # per chunk: share of fp32-mode instances whose contact solve went the fp64 general route
    k_same, flight_zero = True, True
    for c0 in range(0, nt, chunk):
        # evaluation view: the fp64 state at the chunk boundary through each mode's single evaluation (warm start from the mode's own set)
        ref_eval = None
        for name, _ in MODES:
            s_copy = st["fp64"].clone()
            stat_copy = status[name].clone()
            o = ctl[name].new_out()
            ctl[name].stand_step(s_copy, out=o, status=stat_copy)
            torch.cuda.synchronize()
            o = o.cpu().numpy()
            if name == "fp64":
                ref_eval = o
            else:
                ev[name]["tau"].append(rel(o[:, :24], ref_eval[:, :24]))
                nz = np.abs(ref_eval[:, 24:36]).max(axis=1) > 0
                ev[name]["f"].append(np.where(nz, rel(o[:, 24:36], ref_eval[:, 24:36]), np.abs(o[:, 24:36]).max(axis=1)))
        logs = {}
        for name, _ in MODES:
            if name == "fp32":            # the informational flag is sticky: clear it per chunk to count chunks, keep the others
                sh = status[name].cpu().numpy(); sh[:, 2] &= ~capi.FLAG_QP_FP64_ROUTE
                status[name].copy_(torch.from_numpy(sh))
            _, _, lg = ctl[name].rollout(st[name], chunk, out[name], status[name], log=True)
            torch.cuda.synchronize()
            logs[name] = lg.cpu().numpy()
            sh = status[name].cpu().numpy()
            flags[name] |= sh[:, 2]
        fp64_route.append(float(((status["fp32"].cpu().numpy()[:, 2] & capi.FLAG_QP_FP64_ROUTE) != 0).mean()))
        k_same &= all(np.array_equal(status["fp64"].cpu().numpy()[:, 0], status[n].cpu().numpy()[:, 0]) for n in ("mixed", "fp32"))
        ref = logs["fp64"]
        for name in ("mixed", "fp32"):
            cl[name]["tau"].append(rel(logs[name][:, :, :24], ref[:, :, :24]))
            nz = np.abs(ref[:, :, 24:]).max(axis=2) > 0
            cl[name]["f"].append(np.where(nz, rel(logs[name][:, :, 24:], ref[:, :, 24:]), np.abs(logs[name][:, :, 24:]).max(axis=2)))
            for tk in range(chunk):       # tick tk logs the stage-4 evaluation at t + dt
                k = int(round((c0 + tk + 1) * dt / dt))
                if plan["phase"][min(k, len(plan["phase"]) - 1)] == 3:
                    flight_zero &= bool(np.abs(logs[name][tk, :, 24:]).max() == 0.0)
    hard = capi.FLAG_QP_MAXITER | capi.FLAG_NONFINITE | capi.FLAG_ZMP_RANGE | capi.FLAG_NOT_SPD
    res = {"workload": f"{B} instances x {nt} ticks, jump schedule {stance_time} s stance / {flight_time} s flight / stance, N=48, dt=1e-3, "
                       f"velocity pushes x{vscale}; evaluation view sampled every {chunk} ticks",
           "k_bit_identical": bool(k_same), "flight_forces_exactly_zero": bool(flight_zero),
           "instances_flagged": {n: int(((flags[n] & hard) != 0).sum()) for n, _ in MODES},
           "fp32_chunks_with_fp64_route": {"mean_share_of_instances": float(np.mean(fp64_route)), "max_share": float(np.max(fp64_route)),
                                           "per_chunk": [round(x, 4) for x in fp64_route]},
           "stance_ticks": int(round(stance_time / dt)), "flight_ticks": int(round(flight_time / dt))}
    n_st = int(round(stance_time / dt))
    n_land = n_st + int(round(flight_time / dt))
    c_st, c_land = n_st // chunk, -(-n_land // chunk)           # evaluation samples: chunk starts before the take-off / after the touch-down
    for name in ("mixed", "fp32"):
        res[name] = {"evaluation": {q: dist(np.concatenate(ev[name][q])) for q in ("tau", "f")},
                     "evaluation_stance_only": {q: dist(np.concatenate(ev[name][q][:c_st])) for q in ("tau", "f")},
                     "closed_loop": {q: dist(np.concatenate(cl[name][q])) for q in ("tau", "f")},
                     "closed_loop_stance_only": {q: dist(np.concatenate(cl[name][q])[:n_st - 1]) for q in ("tau", "f")}}
        if c_land < len(ev[name]["tau"]):
            res[name]["evaluation_landing_only"] = {q: dist(np.concatenate(ev[name][q][c_land:])) for q in ("tau", "f")}
            res[name]["closed_loop_landing_only"] = {q: dist(np.concatenate(cl[name][q])[n_land:]) for q in ("tau", "f")}
    return res


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    outp = sys.argv[3] if len(sys.argv) > 3 else None
    res = sweep(B, nt)
    print(json.dumps(res, indent=1))
    if outp:
        json.dump(res, open(outp, "w"), indent=1)
