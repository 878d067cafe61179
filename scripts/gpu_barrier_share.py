"""Diagnostic (LMH_DIAG=1 build): share of the PRODUCTION rollout kernel's time that each wave of a robot spends parked at the
workgroup barriers, on bench.py's workload.  Usage: LMH_DIAG=1 python scripts/gpu_barrier_share.py [config=3] [ticks=40] [pre=200]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from linearmpchumanoid_amd.controller import BatchedController, default_config
cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 200
args = bench.parse(["--config", str(cfgno)])
B = args.instances
ctl = BatchedController(B, default_config(dt=args.dt, time_horizon=args.horizon * args.dt, z_com=0.26, warm_start=1))
state, host = bench.build_workload(args, ctl, 0, B, pre + ticks + 10)
out, status = ctl.new_out(), ctl.new_status()
if pre:
    ctl.rollout(state, pre, out, status)
ctl.rollout(state, ticks, out, status)
torch.cuda.synchronize()
o = out.cpu().numpy(); s = state.cpu().numpy()
tot, w0, w1 = o[:, 78], s[:, 91], s[:, 92]
ev = ticks * 4
print(f"config {cfgno}: {B} robots, {ticks} ticks after {pre}: cycles per evaluation mean {tot.mean()/ev:.0f} (min {tot.min()/ev:.0f}, max {tot.max()/ev:.0f})")
print(f"  wave 0 inside barriers: {100*(w0/tot).mean():.1f} % ({(w0/ev).mean():.0f} cycles / evaluation);  wave 1: {100*(w1/tot).mean():.1f} % ({(w1/ev).mean():.0f})")
