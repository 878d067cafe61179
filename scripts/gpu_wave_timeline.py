"""Diagnostic (LMH_DIAG=1 build + LMH_DIAG_NW2=1): per-wave timeline of ONE controller evaluation on the two-wave schedule,
with the time each wave spends waiting at every workgroup barrier.  The evaluation stamped is the first stage of the tick that
follows `pre` rollout ticks of bench.py's workload (realistic warm-start sets and support phase).
Usage: LMH_DIAG=1 LMH_DIAG_NW2=1 python scripts/gpu_wave_timeline.py [config=3] [pre=250] [instances=1024]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from linearmpchumanoid_amd.controller import BatchedController, default_config

cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 250
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
args = bench.parse(["--config", str(cfgno), "--instances", str(B)])
ctl = BatchedController(B, default_config(dt=args.dt, time_horizon=args.horizon * args.dt, z_com=0.26, warm_start=1))
state, host = bench.build_workload(args, ctl, 0, B, pre + 10)
out, status = ctl.new_out(), ctl.new_status()
if pre:
    ctl.rollout(state, pre, out, status)
for rep in range(3):                                               # the third call is the one read (instruction cache warm)
    st = state.clone(); s2 = status.clone()
    o, s2, dbg = ctl.stand_step(st, out=out, status=s2, debug=True)
torch.cuda.synchronize()
d = dbg.cpu().numpy(); s = s2.cpu().numpy()
w0, w1 = d[:, 3700:3760], d[:, 3800:3860]
t0 = w0[:, 0:1]
a0, a1 = (w0 - t0).mean(axis=0), (w1 - t0).mean(axis=0)
phase = host["phase"][int(s[0, 0])] if host["phase"] is not None else 0
nF = np.array([bin((~int(x)) & 0xFFFFFFFF).count("1") for x in s[:, 3]])
print(f"config {cfgno}  B {B}  after {pre} ticks  k {int(s[0,0])}  support phase {int(phase)}  qp iters mean {s[:,1].mean():.2f} max {s[:,1].max()}  |F| mean {nF.mean():.1f}")
names = {0: "start", 1: "fk | kinv+refs_prepare done", 2: "  joined", 3: "com_x share done", 4: "  joined", 5: "tree share done (NE+Jac | CRBA)", 6: "  joined",
         7: "refs share done", 8: "  joined", 10: "qp fills done", 11: "  joined", 12: "Cm | V tile done", 13: "  joined", 14: "w0: rows of Cm, V loaded",
         15: "w0: 15x15 solve done | w1: Z, Mb bp' tiles", 16: "  joined", 17: "w1: Y tiles done", 18: "set-up done (w0: qv | w1: Y)", 19: "w0: S tile done", 20: "w0: Si done", 21: "w0: T1 done",
         22: "w0: W,h done", 23: "w0: qv done", 24: "w0: cone start", 25: "w0: cone done", 26: "recovery done | w1 waiting since", 27: "  joined", 28: "outputs done"}
print("(debug kernel: it also forms the 32 x 32 cone Hessian for its dump, ~5.7k cycles inside 'cone'; stamps cost ~70 cycles each)")
print("%-38s %10s %10s %12s" % ("stamp (cycles from the start)", "wave 0", "wave 1", "barrier wait"))
prev = None
for i in sorted(names):
    v0 = a0[i] if w0[:, i].any() else float("nan"); v1 = a1[i] if w1[:, i].any() else float("nan")
    wait = ""
    if names[i].strip() == "joined" and prev is not None:
        p0, p1 = prev
        wait = "w0 %5.0f  w1 %5.0f" % (v0 - p0, v1 - p1)
    print("%-38s %10.0f %10.0f %12s" % (names[i], v0, v1, wait))
    prev = (v0, v1)
for i, n in ((30, "cone: entry"), (31, "cone: qmax done"), (32, "cone(all free): W rows loaded"), (33, "cone(all free): 12x12 solve done"), (34, "cone(all free): c = Gpinv u done"), (35, "cone(all free): feasibility test done"),
             (40, "last LDL' (N<=16): start"), (41, "  forward + D^-1 done"), (42, "  L rows parked"), (43, "  backward done")):
    if w0[:, i].any():
        print("%-38s %10.0f" % (n, a0[i]))
tot = (w0[:, 28] - w0[:, 0])
print("wave 0 evaluation: mean %.0f  p50 %.0f  max %.0f cycles" % (tot.mean(), np.median(tot), tot.max()))
