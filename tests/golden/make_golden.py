#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (oracle/).

The reference holds no golden vectors, tests or published numbers and cannot be built here, so
these fixtures pin the ORACLE (parity of the oracle itself is "unpinned", see oracle/lmh_oracle.h).
They make oracle regressions visible and give the GPU tests fixed inputs/outputs that do not
depend on re-running the oracle.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Oracle  # noqa: E402
from helpers import perturbed_velocities  # noqa: E402


def main():
    # (i) IK posture of apps/offline/main.cpp:24-35 (also bench.py's initial state)
    o = Oracle(sim_time=5.0, dt=0.01, horizon_time=0.5, do_ik=True)
    r = o.robot()
    with open(os.path.join(HERE, "ik_posture.json"), "w") as f:
        json.dump({"q": [float(x) for x in r["q"]], "z_com": float(o.zcom), "com": [float(x) for x in r["CoM"]],
                   "mass": float(o.mass), "note": "oracle IK to feet (0,-/+0.05,0), CoM (-0.02,0,0.26); invKinematics.cpp:27-52"}, f, indent=1)
    q0 = r["q"].copy()

    # (ii) single-evaluation vectors: reference literals (dt=0.01, N=50) at the IK state + 8 perturbed states
    B = 9
    v = perturbed_velocities(B, seed=424200); v[0] = 0
    vprev = perturbed_velocities(B, seed=515100); vprev[0] = 0
    qs = np.tile(q0, (B, 1))
    rng = np.random.default_rng(99)
    qs[1:, 6:] += rng.normal(0, 0.02, (B - 1, 24))
    qs[1:, 0:3] += rng.normal(0, 0.003, (B - 1, 3))
    qs[1:, 3:6] += rng.normal(0, 0.01, (B - 1, 3))
    rec = {k: [] for k in ("tau", "f", "qpp", "k", "C", "Cg6", "M", "AG", "AGpqp", "Jpqp", "J", "u0", "x", "CoM", "active_mask", "qp_iters")}
    for i in range(B):
        oi = Oracle(sim_time=5.0, dt=0.01, horizon_time=0.5, do_ik=True)
        oi.set_prev_velocity(vprev[i])
        e = oi.eval(qs[i], v[i], 0.37)
        t = oi.terms(); qp = oi.qp()
        rec["tau"].append(e["tau"]); rec["f"].append(e["f"]); rec["qpp"].append(e["qpp"]); rec["k"].append(e["k"])
        rec["C"].append(t["C"]); rec["Cg6"].append(t["Cg"][:6]); rec["M"].append(t["M"]); rec["AG"].append(t["AG"])
        rec["AGpqp"].append(t["AGpqp"]); rec["Jpqp"].append(t["Jpqp"]); rec["J"].append(t["J"]); rec["u0"].append(qp["u0"])
        rec["x"].append(qp["x"]); rec["CoM"].append(oi.robot()["CoM"]); rec["active_mask"].append(e["active_mask"]); rec["qp_iters"].append(e["qp_iters"])
    np.savez_compressed(os.path.join(HERE, "eval_vectors.npz"), q=qs, v=v, v_prev=vprev, t=0.37, dt=0.01, time_horizon=0.5, z_com=o.zcom,
                        **{k: np.array(val) for k, val in rec.items()})

    # (iii) apps/offline trace: 500 ticks at dt=0.01/N=50, CoM x after every tick (main.cpp:86) and k per tick
    oo = Oracle(sim_time=5.0, dt=0.01, horizon_time=0.5, do_ik=True)
    ro = oo.rollout(np.concatenate([q0, np.zeros(30)]), 0.0, 500, log=True)
    np.savez_compressed(os.path.join(HERE, "offline_trace.npz"), comx=ro["comx"], k=ro["k"], state=ro["state"], t=ro["t"],
                        tau_f_last=ro["log"][-1], tau_f_first=ro["log"][0])

    # (iv) k sequences of the 4 RK4 stage times for dt in {0.01, 0.001} (float-accumulated clock)
    ks = {}
    for dt, n in ((0.01, 500), (0.001, 5000)):
        t = 0.0; seq = []
        for _ in range(n):
            seq.append([int(t / dt), int((t + 0.5 * dt) / dt), int((t + 0.5 * dt) / dt), int((t + dt) / dt)])
            t += dt
        ks[str(dt)] = np.array(seq, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "k_sequences.npz"), **{"dt_" + k.replace(".", "p"): v_ for k, v_ in ks.items()})

    # (v) config-2 closed loop: 8 instances x 20 ticks (dt=1e-3, N=16, velocity perturbations)
    o2 = Oracle(sim_time=2.0, dt=1e-3, horizon_time=0.016, do_ik=True)
    v2 = perturbed_velocities(8)
    states, logs, kk = [], [], []
    for i in range(8):
        oi = Oracle(sim_time=2.0, dt=1e-3, horizon_time=0.016, do_ik=True)
        rr = oi.rollout(np.concatenate([q0, v2[i]]), 0.0, 20, log=True)
        states.append(rr["state"]); logs.append(rr["log"]); kk.append(rr["k"])
    np.savez_compressed(os.path.join(HERE, "rollout_config2.npz"), q0=q0, v=v2, dt=1e-3, time_horizon=0.016, z_com=o2.zcom,
                        state=np.array(states), log=np.array(logs), k=np.array(kk))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
