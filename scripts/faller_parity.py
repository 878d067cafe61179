#!/usr/bin/env python3
"""BASELINE config 2 at its stated amplitude (U(-0.3, 0.3) m/s pushes): parity of the FAILURE.  (run on the GPU box)

A backward push beyond ~0.18 m/s puts the capture point behind the heel; such a robot falls whatever the torques do.  The first robots
of the draw that fall are run tick by tick on the GPU (tau | f log of every tick) and in the C oracle from the same state: the relative
difference of tau and f along the fall, the tick at which the GPU raises a status flag and the tick at which the oracle's state stops
being finite.  python scripts/faller_parity.py [robots=2] [ticks=2000]  ->  text table (profiles/r04_config2_fallers.txt)"""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch

from helpers import perturbed_velocities
from linearmpchumanoid_amd.controller import BatchedController, default_config, ik_start_posture
from oracle.pyoracle import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
N, md, dt = 16, 2e-2, 1e-3
th = N * md + 1e-9
q0, zcom = ik_start_posture(0)
v_full = perturbed_velocities(1024)
idx = [i for i in range(1024) if v_full[i, 0] < -0.22][:n]
print("robots", idx, "v_x", np.round(v_full[idx, 0], 4).tolist())
ctl = BatchedController(len(idx), default_config(dt=dt, time_horizon=th, z_com=zcom, mpc_dt=md, warm_start=1))
ctl.set_refs_stance(nt * dt + 1.0, 2)
st = ctl.new_state(q0, v_full[idx], t=0.0)
out, status = ctl.new_out(), ctl.new_status()
chunk = 10
log = torch.zeros((chunk, len(idx), 36), dtype=torch.float64, device=ctl.device)
glog = np.zeros((nt, len(idx), 36)); gflag = np.zeros((nt // chunk, len(idx)), dtype=np.int64); gvmax = np.zeros((nt // chunk, len(idx)))
for c in range(nt // chunk):
    ctl.rollout(st, chunk, out, status, log)
    torch.cuda.synchronize()
    glog[c * chunk:(c + 1) * chunk] = log.cpu().numpy()
    gflag[c] = status.cpu().numpy()[:, 2]
    gvmax[c] = st[:, 30:60].abs().amax(dim=1).cpu().numpy()
for j, i in enumerate(idx):
    o = Oracle(sim_time=nt * dt + 1.0, dt=md, horizon_time=th, do_ik=True)
    r = o.rollout(np.concatenate([q0, v_full[i]]), 0.0, nt, dt=dt, log=True)
    ref = r["log"]
    fin = np.isfinite(ref).all(axis=1)
    t_orc = int(np.argmin(fin)) if not fin.all() else nt
    fl = np.nonzero(gflag[:, j])[0]
    t_gpu = int(fl[0]) * chunk if len(fl) else nt
    print(f"robot {i}: GPU first flagged launch covers ticks {t_gpu}..{t_gpu + chunk - 1} (flags {int(gflag[fl[0], j]) if len(fl) else 0}); oracle log non-finite from tick {t_orc}")
    print("  tick   |tau|max   |f|max   rel err tau   rel err f   |v|max (GPU, end of its 10-tick launch)")
    for tk in list(range(0, min(t_orc, nt), 50)) + list(range(max(0, min(t_orc, t_gpu) - 40), min(t_orc, nt), 4)):
        a, b = glog[tk, j], ref[tk]
        if not (np.isfinite(a).all() and np.isfinite(b).all()):
            print(f"  {tk:5d}   non-finite (GPU finite: {bool(np.isfinite(a).all())}, oracle finite: {bool(np.isfinite(b).all())})")
            continue
        et = np.abs(a[:24] - b[:24]).max() / max(np.abs(b[:24]).max(), 1e-300)
        ef = np.abs(a[24:] - b[24:]).max() / max(np.abs(b[24:]).max(), 1e-9 * 52.0)
        print(f"  {tk:5d}   {np.abs(b[:24]).max():9.3g} {np.abs(b[24:]).max():9.3g}   {et:9.2e}   {ef:9.2e}   {gvmax[tk // chunk, j]:8.3g}")
