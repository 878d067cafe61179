#!/usr/bin/env python3
"""Headline benchmark: control ticks/s/node of the batched NAO WBC+MPC closed loop.

One "step" = one launch of the fused rollout kernel = `--ticks` RK4 control ticks (4 controller
evaluations each) for every robot instance of this rank.  Workload at N=1 is BASELINE.json
configs[1]: 1024 NAO instances, balance task, dt = 1 ms, LIPM-MPC horizon N = 16, WBC QP per
evaluation; initial state = IK posture + per-instance velocity perturbation.
Ranks shard instances (weak scaling: 1024 per GPU), no data-path collective; one RCCL gather of
128-B per-instance summaries ends the run.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_TICK = 1248        # state in (60 f64) + state out (60) + tau|f log (36)
ALG_FLOP_PER_TICK = 8.0e5        # 2.0e5 flop per controller evaluation x 4
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s HBM3E
FP64_VALU_PEAK_TFLOPS = 78.6     # vendor fp64 vector peak
PROFILE_TAG = "r01h"             # profiles/<tag>_rollout_summary.json: PMC passes of the kernel as committed


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--instances", type=int, default=1024, help="robot instances per GPU")
    ap.add_argument("--ticks", type=int, default=10, help="RK4 ticks per step (per launch)")
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--cold", action="store_true", help="cold-start the QP active set every evaluation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path)")
    ap.add_argument("--reset-every", type=int, default=23,
                    help="restart the rollouts from the initial states after this many steps (the reference's closed loop, "
                         "which integrates the controller's own acceleration, leaves its valid range ~0.5 s after a 0.3 m/s push; "
                         "DESIGN.md 'Long runs'); the default equals warmup + steps of the default run, i.e. no restart there")
    ap.add_argument("--summary-out", default=None, help="write the gathered end-of-run summary (wire.write_summary format) to this path")
    ap.add_argument("--traffic", type=float, default=None,
                    help="HBM bytes per launch from rocprofv3 --pmc; default: the committed profiles/ summary when the workload matches it")
    return ap.parse_args()


def ik_posture(device=0):
    """IK start posture of apps/offline (feet (0,-/+0.05,0), CoM (-0.02,0,0.26)) and the LIPM height, computed by the
    product's own IK kernel (Kinematics::compute on the GPU), as apps/offline/main.cpp:24-39 does at start-up."""
    from linearmpchumanoid_amd.controller import ik_start_posture
    return ik_start_posture(device)


def perturbed_velocities(first, count, seed=20260001):
    v = np.zeros((count, 30))
    for i in range(count):
        rng = np.random.default_rng(seed + first + i)
        v[i, 0:2] = rng.uniform(-0.3, 0.3, 2)
        v[i, 6:] = rng.normal(0.0, 0.05, 24)
    return v


def cpu_baseline(q0, zcom, args):
    """Reference CPU path = the C oracle in reference-faithful mode, timed on this host's cores
    on a bounded sample of the same workload (same states, same tick function)."""
    from oracle import pyoracle
    ncores = os.cpu_count() or 1
    threads = max(1, min(ncores, 64))
    th = args.horizon * args.dt
    # calibrate: 1 instance x few ticks on one core
    v = perturbed_velocities(0, threads * 2)
    st = np.concatenate([np.broadcast_to(q0, (len(v), 30)), v], axis=1)
    sec, _, _ = pyoracle.batch_rollout(st[:1], None, 0.0, args.dt, 5, 2.0, th, zcom, nthreads=1)
    per_tick = sec / 5
    ticks = max(5, int(args.cpu_seconds / max(per_tick, 1e-6) / 2))
    ticks = min(ticks, 200)
    sec1, _, _ = pyoracle.batch_rollout(st[:1], None, 0.0, args.dt, ticks, 2.0, th, zcom, nthreads=1)
    one = ticks / sec1
    tk = 100                                                  # same regime as the GPU run (first 0.1 s of the rollout)
    n_inst = max(threads, int(args.cpu_seconds / 2 * one / tk) // threads * threads)
    n_inst = min(n_inst, 4096)
    v = perturbed_velocities(0, n_inst)
    st = np.concatenate([np.broadcast_to(q0, (len(v), 30)), v], axis=1)
    secN, _, _ = pyoracle.batch_rollout(st[:n_inst], None, 0.0, args.dt, tk, 2.0, th, zcom, nthreads=threads)
    allc = n_inst * tk / secN
    return {"value": allc, "unit": "control ticks/s", "cores": threads, "kind": "port",
            "single_core_value": one,
            "sample": f"C oracle (-O2), {n_inst} instances x {tk} ticks on {threads} threads; 1 instance x {ticks} ticks on 1 thread; same states/tick function as the GPU run, 1 WBC solve per evaluation"}


def profiled_traffic(args):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, profiles/<PROFILE_TAG>_rollout_summary.json, made by scripts/profile_rollout.sh +
    scripts/summarise_profile.py); only valid for the workload it was collected on."""
    if args.traffic is not None:
        return args.traffic
    if (args.instances, args.ticks, args.horizon) != (1024, 10, 16) or args.cold:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_rollout_summary.json")) as f:
            d = json.load(f)["derived"]
        return d["hbm_write_bytes_per_launch"] + d["hbm_fetch_bytes_per_launch_x2_gfx950_correction"]
    except Exception:
        return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from linearmpchumanoid_amd.controller import BatchedController, default_config
    from linearmpchumanoid_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    if os.environ.get("LMH_BENCH_DEVICE") is not None:          # rehearsal of N>1 on a one-GPU box (gloo): all ranks share a card
        local_rank = int(os.environ["LMH_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)

    B = args.instances
    first, count = sharding.shard_range(B * world, world, rank)
    q0, zcom = ik_posture(local_rank)
    th = args.horizon * args.dt
    cfg = default_config(dt=args.dt, time_horizon=th, z_com=zcom, warm_start=0 if args.cold else 1)
    ctl = BatchedController(count, cfg, device=local_rank)
    reset_every = max(1, args.reset_every)
    total_ticks = min(args.warmup + args.steps, reset_every) * args.ticks
    ctl.set_refs_stance(total_ticks * args.dt + 1.0, 2)
    v = perturbed_velocities(first, count)
    state = ctl.new_state(q0, v, t=0.0)
    state0 = state.clone()
    out, status = ctl.new_out(), ctl.new_status()
    log = torch.zeros((args.ticks, count, 36), dtype=torch.float64, device=ctl.device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    done = 0

    def step():
        nonlocal done
        if done and done % reset_every == 0:
            state.copy_(state0)                               # device-to-device, inside the timed region when it happens
        ctl.rollout(state, args.ticks, out, status, log)
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
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctl.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # end-of-run summary: 16 f64 per instance, gathered over RCCL (the only collective)
    summary = sharding.make_summary(state, out, status)
    if world > 1 and args.backend != "nccl":
        summary = summary.cpu()
    gathered = sharding.gather_summaries(summary, world, rank)
    flags = int((status[:, 2] != 0).sum().item())
    if rank == 0 and args.summary_out and gathered is not None:
        from linearmpchumanoid_amd import wire
        wire.write_summary(args.summary_out, gathered.cpu().numpy(), dt=args.dt)

    if rank == 0:
        total_instances = B * world
        ticks_total = total_instances * args.ticks * args.steps
        value = ticks_total / elapsed
        launch_s = kernel_ms * 1e-3
        alg_bytes = count * args.ticks * ALG_BYTES_PER_TICK
        achieved = alg_bytes / launch_s / 1e9
        res = {
            "metric": "control ticks/s/node (batched NAO WBC+MPC @1kHz)",
            "value": value, "unit": "control ticks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{B} NAO instances/GPU, balance task (IK posture + velocity perturbation), dt={args.dt}, LIPM-MPC horizon N={args.horizon}, WBC QP per evaluation, RK4 closed loop",
                       "instances_per_gpu": B, "ticks_per_step": args.ticks, "evaluations_per_tick": 4,
                       "qp_start": "cold" if args.cold else "warm", "parallelism": f"instances sharded x{world}",
                       "rollout_restarts": (args.warmup + args.steps - 1) // reset_every},
            "evaluations_per_s": value * 4,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": profiled_traffic(args), "kernel": "lmh_rollout_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "fp64_valu": {"achieved_tflops": count * args.ticks * ALG_FLOP_PER_TICK / launch_s / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                          "frac": count * args.ticks * ALG_FLOP_PER_TICK / launch_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                          "note": "the path is fp64-VALU/LDS-latency bound, not HBM bound (SURVEY 8d)"},
            "instances_flagged": flags,
            "summary_rows_gathered": int(gathered.shape[0]) if gathered is not None else 0,
        }
        if not args.no_cpu_baseline and world == 1:              # the CPU baseline is timed at N = 1 only
            try:
                res["cpu_baseline"] = cpu_baseline(q0, zcom, args)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                res["cpu_baseline"] = {"value": None, "unit": "control ticks/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
