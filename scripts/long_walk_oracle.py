#!/usr/bin/env python3
"""How long does config 3's walking loop stay in range, in the CPU oracle and on the GPU?  (run on the GPU box)

The robots of bench.py's default workload with the longest step length, ticks 0..T, in the C oracle (one thread each) and on the GPU:
|v|max along the rollout and the first tick with a non-finite state.  python scripts/long_walk_oracle.py [T=96000] [robots=8]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import bench
from linearmpchumanoid_amd.controller import BatchedController, default_config
from oracle import pyoracle

T = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
args = bench.parse(["--config", "3", "--instances", "512"])
th = args.horizon * args.mpc_dt + 1e-9
ctl = BatchedController(512, default_config(dt=args.dt, time_horizon=th, z_com=0.26, mpc_dt=args.mpc_dt, warm_start=1))
state, host = bench.build_workload(args, ctl, 0, 512, T + 10)
idx = np.argsort(-host["xscale"])[:n]
print("robots", idx.tolist(), "step lengths", np.round(host["xscale"][idx], 4).tolist())
seg = 8000
out, status = ctl.new_out(), ctl.new_status()
gpu_v = []
for c in range(T // seg):
    ctl.rollout(state, seg, out, status)
    gpu_v.append(state[torch.as_tensor(idx, device=state.device), 30:60].abs().amax(dim=1).cpu().numpy())
st = np.concatenate([host["q0"][idx], host["v"][idx]], axis=1)
cpu_v = []
t = 0.0
for c in range(T // seg):
    sec, st_out, _ = pyoracle.batch_rollout_ex(st, t, args.dt, seg, th, host["zmp_x"], host["zmp_y"], host["phase"], host["segs"], host["sos"],
                                               host["xscale"][idx], host["zcom"], None, nthreads=n, mpc_dt=args.mpc_dt)
    st = st_out
    t += seg * args.dt
    cpu_v.append(np.abs(st[:, 30:60]).max(axis=1))
    print(f"ticks ..{(c+1)*seg:6d}: |v|max  oracle {np.array2string(cpu_v[-1], precision=2)}   GPU {np.array2string(gpu_v[c], precision=2)}", flush=True)
