#!/usr/bin/env python3
"""GPU diagnostics of the rollout kernel (run on the GPU box through gpurun; none of this is product code).

  python scripts/diag.py timeline [config=3] [pre=1300] [instances=1024]     needs LMH_DIAG=1 LMH_DIAG_NW2=1
      per-wave timeline of ONE controller evaluation on the two-wave schedule (two-wave debug kernel of the diagnostic build), with the
      wait of each wave at every workgroup barrier; the evaluation stamped follows `pre` rollout ticks of bench.py's workload.
  python scripts/diag.py barrier [config=3] [ticks=200] [pre=1300]            needs LMH_DIAG=1
      share of the PRODUCTION rollout kernel's cycles that each wave of a robot spends inside workgroup barriers.
  python scripts/diag.py ptimeline [config=3] [pre=1300] [instances=4096]   needs LMH_VARIANT=<a build with -DLMH_SUBSTAMPS -DLMH_DIAG_TL=stage>
      the same timeline of the PRODUCTION rollout kernel: every mark both waves pass in the evaluation of Runge-Kutta stage `stage` of the
      launch's last tick (the stamps go to the log buffer).
  python scripts/diag.py rounds [config=3] [ticks=4000] [chunk=100]
      QP round histogram (status[:, 1] = max rounds per launch) and flag counts along a rollout of bench.py's workload.
      The last line (per-robot launch cycles, max / mean) is the wave-slot occupancy proxy of the ticket scheduler: 1.0 = every robot
      slot is busy until the launch ends.

LMH_DIAG=1 selects liblmh_hip_diag.so (build it with `LMH_DIAG=1 python linearmpchumanoid_amd/build.py`)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import bench
from linearmpchumanoid_amd.controller import BatchedController, default_config


def setup(cfgno, B, total_ticks, extra=()):
    args = bench.parse(["--config", str(cfgno), "--instances", str(B)] + list(extra))
    th = args.horizon * args.mpc_dt + 1e-9
    ctl = BatchedController(B, default_config(dt=args.dt, time_horizon=th, z_com=0.26, mpc_dt=0.0 if args.coupled else args.mpc_dt, warm_start=1))
    state, host = bench.build_workload(args, ctl, 0, B, total_ticks)
    return args, ctl, state, host


def timeline(argv):
    cfgno = int(argv[0]) if len(argv) > 0 else 3
    pre = int(argv[1]) if len(argv) > 1 else 1300
    B = int(argv[2]) if len(argv) > 2 else 1024
    args, ctl, state, host = setup(cfgno, B, pre + 10)
    out, status = ctl.new_out(), ctl.new_status()
    if pre:
        ctl.rollout(state, pre, out, status)
    for rep in range(3):                                           # the third call is the one read (instruction cache warm)
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
    print("sub-stamps (cycles from the start; w0 | w1):")
    for i, n in ((44, "fk: sincos stored"), (45, "fk: local transforms built"), (46, "com_x: (w1: CoM done)"), (47, "com_x: E, p written"), (48, "com_x: B written"),
                 (49, "NE: sweeps done"), (50, "NE: body forces done"), (51, "NE: backward done"), (52, "CRBA: iterations done")):
        v0 = a0[i] if w0[:, i].any() else float("nan"); v1 = a1[i] if w1[:, i].any() else float("nan")
        print("  %-36s %10.0f %10.0f" % (n, v0, v1))
    for i, n in ((30, "cone: entry"), (31, "cone: qmax done"), (32, "cone(all free): W rows loaded"), (33, "cone(all free): 12x12 solve done"), (34, "cone(all free): c = Gpinv u done"), (35, "cone(all free): feasibility test done"),
                 (40, "last LDL' (N<=16): start"), (41, "  forward + D^-1 done"), (42, "  L rows parked"), (43, "  backward done")):
        if w0[:, i].any():
            print("%-38s %10.0f" % (n, a0[i]))
    tot = (w0[:, 28] - w0[:, 0])
    print("wave 0 evaluation: mean %.0f  p50 %.0f  max %.0f cycles" % (tot.mean(), np.median(tot), tot.max()))


def barrier(argv):
    cfgno = int(argv[0]) if len(argv) > 0 else 3
    ticks = int(argv[1]) if len(argv) > 1 else 200
    pre = int(argv[2]) if len(argv) > 2 else 1300
    args, ctl, state, host = setup(cfgno, bench.DEFAULTS[cfgno]["instances"], pre + ticks + 10)
    B = state.shape[0]
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
    print(f"  per-robot launch cycles: mean {tot.mean():.0f}  max {tot.max():.0f}  (max / mean = {tot.max()/tot.mean():.3f})")
    # where the robots ran (HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]; XCC_ID [3:0])
    h0, h1, xcc = s[:, 93].astype(np.int64), s[:, 94].astype(np.int64), s[:, 95].astype(np.int64) & 15
    simd0, simd1 = (h0 >> 4) & 3, (h1 >> 4) & 3
    cu = ((xcc << 8) | (((h0 >> 13) & 7) << 5) | (((h0 >> 12) & 1) << 4) | ((h0 >> 8) & 15))
    print("  SIMD of (wave 0, wave 1): " + "  ".join(f"({a},{b}): {int(((simd0 == a) & (simd1 == b)).sum())}" for a in range(4) for b in range(4) if ((simd0 == a) & (simd1 == b)).any()))
    # robots whose leading wave shares its SIMD with another robot's leading wave (same CU, any time: the placement is fixed per workgroup)
    key = cu * 4 + simd0
    cnt = {}
    slot = {}
    for i in range(B):
        slot.setdefault((int(cu[i]), int(h0[i] & 15), int(simd0[i])), []).append(i)
    per_cu = {}
    for (c, w, sd), robots in slot.items():
        per_cu.setdefault(c, {}).setdefault(sd, set()).add(w)
    shared = np.array([len(per_cu[int(cu[i])][int(simd0[i])]) for i in range(B)])
    for n in sorted(set(shared)):
        m = shared == n
        print(f"  leading waves on the robot's SIMD: {n}: {int(m.sum())} robots, cycles / evaluation mean {tot[m].mean()/ev:.0f}")
    print(f"  CUs seen: {len(per_cu)}")
    print("  wait per join of an evaluation, cycles (wave 0 | wave 1); the joins in order: X images, tree + references, QP fills, Cm | V, 15 x 15 solve | tiles, Y, recovery:")
    print("    w0: " + " ".join(f"{v:6.0f}" for v in o[:, 36:44].mean(axis=0) / ev))
    print("    w1: " + " ".join(f"{v:6.0f}" for v in o[:, 48:56].mean(axis=0) / ev))
    rt = np.concatenate([o[:, 44:48].sum(axis=0), o[:, 56:58].sum(axis=0)])
    if rt[4] > 0:
        print(f"  cone solve per evaluation: all-free solve tried {rt[5]/rt[4]:.3f}, accepted {rt[0]/rt[4]:.3f};  rounds of the iteration: push-through {rt[1]/rt[4]:.3f}  thin {rt[2]/rt[4]:.3f}  general {rt[3]/rt[4]:.3f}")
    pc = np.percentile(tot / ev, [1, 10, 50, 90, 99])
    print("  cycles / evaluation percentiles 1/10/50/90/99: " + " ".join(f"{v:.0f}" for v in pc))
    for q in range(0, B, 512):
        seg = tot[q:q + 512] / ev
        print(f"    robots {q:5d}..{q+511:5d}: mean {seg.mean():.0f} min {seg.min():.0f} max {seg.max():.0f}  w0 wait {(w0[q:q+512]/ev).mean():.0f}  w1 wait {(w1[q:q+512]/ev).mean():.0f}")
    dur = o[:, 62]                                                  # busy time summed over the robot's chunks
    span = o[:, 61].max() - o[:, 60].min()
    print(f"  launch makespan {span/100:.0f} us (100 MHz clock); sum of robot times / 1024 slots = {dur.sum()/1024/100:.0f} us: slot occupancy {dur.sum()/1024/span:.3f};  robot time mean {dur.mean()/100:.0f} max {dur.max()/100:.0f} us")
    top = np.argsort(-tot)[:12]
    print("  slowest robots: " + "  ".join(f"{int(i)}:{tot[i]/ev:.0f}(cu {int(cu[i])} simd {int(simd0[i])},{int(simd1[i])} rounds {int(status[int(i),1])})" for i in top))
    st_ = status.cpu().numpy()
    for r_ in sorted(set(st_[:, 1])):
        m = st_[:, 1] == r_
        print(f"    max QP rounds {int(r_)}: {int(m.sum())} robots, cycles / evaluation mean {tot[m].mean()/ev:.0f}")
    for x_ in range(8):
        m = xcc == x_
        if m.any(): print(f"    XCC {x_}: {int(m.sum())} robots, mean {tot[m].mean()/ev:.0f}")


def rounds(argv):
    cfgno = int(argv[0]) if len(argv) > 0 else 3
    ticks = int(argv[1]) if len(argv) > 1 else 4000
    chunk = int(argv[2]) if len(argv) > 2 else 100
    args, ctl, state, host = setup(cfgno, bench.DEFAULTS[cfgno]["instances"], ticks + 10)
    out, status = ctl.new_out(), ctl.new_status()
    for c in range(ticks // chunk):
        ctl.rollout(state, chunk, out, status)
        s = status.cpu().numpy()
        nF = np.array([bin((~int(x)) & 0xFFFFFFFF).count("1") for x in s[:, 3]])
        fl = s[:, 2]
        kinds = " ".join(f"{name}:{int(((fl >> b) & 1).sum())}" for b, name in enumerate(("QP_MAXITER", "NONFINITE", "ZMP_RANGE", "NOT_SPD", "FP64_ROUTE")) if ((fl >> b) & 1).any())
        print(f"ticks {c*chunk:6d}..{(c+1)*chunk:6d}: max rounds per launch mean {s[:,1].mean():5.2f} max {s[:,1].max():3d}  flagged {int((fl != 0).sum()):5d} {kinds}  |F| mean {nF.mean():5.1f}  |v|max {float(state[:, 30:60].abs().max()):.2f}  base x {float(state[:, 0].min()):.2f}..{float(state[:, 0].max()):.2f}")


PMARKS = {0: "evaluation starts", 46: "com_x: CoM done (w1)", 47: "com_x: E, p written", 48: "com_x: B written", 3: "com_x share done", 4: "  joined (X images)",
          49: "NE: sweeps done", 50: "NE: body forces done", 51: "NE: backward sweep done", 52: "CRBA: levels done (w1)", 64: "NE done, Jacobian starts (w0)",
          5: "tree share done", 66: "chain A starts (w1)", 67: "chain B starts (w0)", 68: "chain B done, prefill starts (w0)", 7: "references done", 8: "  joined (tree + references)",
          10: "QP fills done", 11: "  joined (fills)", 12: "Cm | V tile done", 13: "  joined (Cm | V)", 14: "rows of Cm, V loaded (w0)", 15: "15 x 15 solve done (w0)",
          60: "helper: RK4 stage starts", 69: "helper: RK4 stage done", 16: "  joined (solve | tiles)", 17: "Y tiles done (w1)", 63: "K_f^-1 prework starts (w1)",
          19: "S tile done (w0)", 20: "S^-1 done (w0)", 22: "[W | h] done (w0)", 23: "set-up chain done (w0)", 18: "set-up done", 24: "cone starts (w0)", 30: "cone: entry",
          32: "cone: 12 x 12 rows loaded", 33: "cone: 12 x 12 solved", 34: "cone: coefficients", 35: "cone: all-free verdict", 31: "cone: qv, qmax done", 80: "cone: thin solve starts",
          81: "cone: thin solve done", 82: "cone: general solve starts", 83: "cone: general solve done", 84: "cone: 16-row solve starts", 85: "cone: 16-row solve done",
          40: "  LDL': forward starts", 41: "  LDL': forward done", 42: "  LDL': factor parked", 43: "  LDL': backward done", 25: "cone done (w0)", 61: "look-ahead: references start (w1)",
          62: "look-ahead: kinematics start (w1)", 44: "look-ahead: sin / cos stored (w1)", 45: "look-ahead: local transforms built (w1)", 70: "look-ahead done (w1)",
          26: "recovery done (w0) | helper arrives (w1)", 27: "  joined (recovery)", 28: "outputs done", 71: "evaluation returns", 72: "integrator stage done"}


def ptimeline(argv):
    """per-wave timeline of ONE evaluation of the production rollout kernel (libraries built with -DLMH_SUBSTAMPS -DLMH_DIAG_TL=<stage>,
    loaded through LMH_VARIANT): python scripts/diag.py ptimeline [config=3] [pre=1300] [instances=4096]"""
    cfgno = int(argv[0]) if len(argv) > 0 else 3
    pre = int(argv[1]) if len(argv) > 1 else 1300
    B = int(argv[2]) if len(argv) > 2 else 4096
    T = 8                                                          # 8 ticks x 36 doubles of log per robot >= the 256 the stamps take
    args, ctl, state, host = setup(cfgno, B, pre + T + 2)
    out, status = ctl.new_out(), ctl.new_status()
    if pre:
        ctl.rollout(state, pre, out, status)
    log = torch.zeros((T, B, 36), dtype=torch.float64, device=ctl.device)
    ctl.rollout(state, T, out, status, log)
    torch.cuda.synchronize()
    d = log.cpu().numpy().reshape(-1)[:256 * B].reshape(B, 256)
    s = status.cpu().numpy()
    ok = d[:, 0] > 0
    w0, w1 = d[ok, 0:100], d[ok, 100:200]
    t0 = w0[:, 0:1]
    k = int(s[0, 0])
    phase = host["phase"][k] if host["phase"] is not None else 0
    print(f"config {cfgno}  {int(ok.sum())} robots  production rollout kernel, last tick of an {T}-tick launch after {pre} ticks, variant {os.environ.get('LMH_VARIANT', '')}  k {k}  support phase {int(phase)}")
    print("(stamps cost ~70 cycles each on the wave that takes them; cycles since wave 0 entered the evaluation, mean over the robots; a wave that did not pass a mark: -)")
    rows = []
    for i, name in PMARKS.items():
        a = [(w[:, i][w[:, i] > 0] - t0[w[:, i] > 0, 0]) for w in (w0, w1)]
        if len(a[0]) == 0 and len(a[1]) == 0:
            continue
        m = [x.mean() if len(x) else float("nan") for x in a]
        rows.append((np.nanmin(m), i, name, m, [len(x) for x in a]))
    for _, i, name, m, n in sorted(rows):
        f = lambda v, c: ("%10.0f" % v + ("" if c == int(ok.sum()) else " (%d)" % c)) if c else "         -"
        print("%-46s [%2d] %-18s %-18s" % (name, i, f(m[0], n[0]), f(m[1], n[1])))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "timeline"
    {"timeline": timeline, "ptimeline": ptimeline, "barrier": barrier, "rounds": rounds}[cmd](sys.argv[2:])
