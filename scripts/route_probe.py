#!/usr/bin/env python3
"""Which free sets does the cone solve meet along bench.py's walking workload?  (run on the GPU box)
python scripts/route_probe.py [config=3] [instances=1024] [tick ...]: after each tick count, the warm-start free set of every robot (status[:, 3]
holds its complement): size per foot, the most common masks, and for each of those the rank / smallest eigenvalue of K_f = G_F G_F' per foot
-- the push-through route needs K_f nonsingular on every foot that has a free coefficient (cone_pushthrough), sets of at most 8 take the thin
route, the rest the general |F| x |F| solve."""
import collections
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import bench
from linearmpchumanoid_amd.controller import BatchedController, default_config

cfgno = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ticks = [int(a) for a in sys.argv[3:]] or [1100, 1250, 1300, 1350, 1400, 1450, 1500, 1600, 1800]
args = bench.parse(["--config", str(cfgno), "--instances", str(B)])
th = args.horizon * args.mpc_dt + 1e-9
cfg = default_config(dt=args.dt, time_horizon=th, z_com=0.26, mpc_dt=args.mpc_dt, warm_start=1)
ctl = BatchedController(B, cfg)
state, host = bench.build_workload(args, ctl, 0, B, max(ticks) + 10)
mu = float(cfg.mu)
ray = np.array([[mu, 0, 1], [0, mu, 1], [-mu, 0, 1], [0, -mu, 1]], dtype=float)
vtx = np.array([[0.1, 0.025, 0], [0.1, -0.025, 0], [-0.05, 0.025, 0], [-0.05, -0.025, 0]], dtype=float)
G = np.array([np.concatenate([np.cross(vtx[v], ray[e]), ray[e]]) for v in range(4) for e in range(4)]).T      # 6 x 16
out, status = ctl.new_out(), ctl.new_status()
done = 0
for tk in ticks:
    ctl.rollout(state, tk - done, out, status)
    torch.cuda.synchronize()
    done = tk
    s = status.cpu().numpy()
    F = (~s[:, 3].astype(np.int64)) & 0xFFFFFFFF
    k = int(s[0, 0])
    ph = int(host["phase"][k]) if host["phase"] is not None else -1
    cnt = collections.Counter(int(f) for f in F)
    print(f"after {tk} ticks: k {k} support phase {ph}; distinct free sets {len(cnt)}; rounds of the last launch: mean {s[:,1].mean():.2f} max {s[:,1].max()}")
    for f, n in cnt.most_common(4):
        desc = []
        for ft, name in ((0, "R"), (1, "L")):
            m = (f >> (16 * ft)) & 0xFFFF
            idx = [j for j in range(16) if (m >> j) & 1]
            if not idx:
                desc.append(f"{name}: -")
                continue
            K = G[:, idx] @ G[:, idx].T
            ev = np.linalg.eigvalsh(K)
            per_v = "".join(str(sum(1 for j in idx if j // 4 == v)) for v in range(4))
            desc.append(f"{name}: {len(idx):2d} free (per vertex {per_v}) rank {np.linalg.matrix_rank(K, tol=1e-10)} min eig {ev[0]:.1e}")
        nf = bin(f).count("1")
        route = "all-free" if f == 0xFFFFFFFF else "thin" if nf <= 8 else "push-through if every K_f is regular, else general"
        print(f"   {n:5d} robots  F = {f:08x}  |F| {nf:2d}  {'; '.join(desc)}  -> {route}")
