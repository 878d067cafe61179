"""Diagnostic: how long does the config-3 / config-4 walking closed loop stay in range?  Runs bench.py's workload for
n ticks in launches of 40 and prints, every 200 ticks, the number of flagged robots and the spread of the base state.
Usage: python scripts/gpu_walk_long.py [config=3] [ticks=4000] [instances=4096]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from linearmpchumanoid_amd.controller import BatchedController, default_config

cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
extra = sys.argv[4:]                                            # e.g. --step-time 0.2 --ds-time 0.05 --settle-time 0.1
args = bench.parse(["--config", str(cfgno), "--instances", str(B)] + extra)
print("config", cfgno, "ticks", nt, "instances", B, "step", args.step_time, "ds", args.ds_time, "settle", args.settle_time, flush=True)
ctl = BatchedController(B, default_config(dt=args.dt, time_horizon=args.horizon * args.dt, z_com=0.26, warm_start=1))
state, host = bench.build_workload(args, ctl, 0, B, nt)
out, status = ctl.new_out(), ctl.new_status()
flag_or = torch.zeros(B, dtype=torch.int32, device=ctl.device)
for done in range(0, nt, 40):
    ctl.rollout(state, 40, out, status)
    flag_or |= status[:, 2]
    if (done + 40) % 200 == 0:
        torch.cuda.synchronize()
        s = state.cpu().numpy(); f = flag_or.cpu().numpy(); o = out.cpu().numpy()
        print("tick %5d  flagged %4d (flags or %d)  base x [%.4f, %.4f]  z [%.4f, %.4f]  |v|max %.3g  qp it max %d  sum fz [%.2f, %.2f]" % (
            done + 40, int((f != 0).sum()), int(np.bitwise_or.reduce(f)), s[:, 0].min(), s[:, 0].max(), s[:, 2].min(), s[:, 2].max(),
            np.abs(s[:, 30:60]).max(), int(status[:, 1].max().item()), (o[:, 29] + o[:, 35]).min(), (o[:, 29] + o[:, 35]).max()), flush=True)
