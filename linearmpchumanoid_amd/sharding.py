"""Instance sharding across the GPUs of one node and the end-of-run summary gather.

Rollouts are independent (the reference has no cross-robot coupling), so rank r of P owns the
contiguous instance range [r*B/P, (r+1)*B/P) and nothing is exchanged during the rollout.  The
only collective is one gather of a fixed 16-double (128 B) summary per instance at the end of
the run (RCCL over xGMI when the backend is nccl; gloo on CPU in tests).
"""
import torch
import torch.distributed as dist

SUMMARY_WIDTH = 16


def shard_range(total, world, rank):
    """(first, count) of the contiguous block owned by `rank`; blocks differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(int(total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def make_summary(state, out, status, ctl=None):
    """[B,16] f64: final base pose(6) | final t | max|tau| | sum f_z | f_z R, L | k | qp iters | flags | active count | checksum
    (include/lmh.h lmh_make_summary).  With a controller handle the record is produced by the HIP summary kernel; the torch
    form below is the same arithmetic for host tensors (gloo rehearsals / CPU tests)."""
    if ctl is not None and state.is_cuda:
        return ctl.make_summary(state, out, status)
    B = state.shape[0]
    s = torch.zeros((B, SUMMARY_WIDTH), dtype=torch.float64, device=state.device)
    s[:, 0:6] = state[:, 0:6]
    s[:, 6] = state[:, 90]
    s[:, 7] = out[:, 0:24].abs().amax(dim=1)
    s[:, 8] = out[:, 24 + 5] + out[:, 24 + 11]
    s[:, 9] = out[:, 24 + 5]
    s[:, 10] = out[:, 24 + 11]
    st = status.to(torch.int64)
    s[:, 11] = st[:, 0].to(torch.float64)
    s[:, 12] = st[:, 1].to(torch.float64)
    s[:, 13] = st[:, 2].to(torch.float64)
    mask = st[:, 3] & 0xFFFFFFFF
    cnt = torch.zeros_like(mask)
    for b in range(32):
        cnt += (mask >> b) & 1
    s[:, 14] = cnt.to(torch.float64)
    s[:, 15] = torch.cumsum(state[:, 0:60], dim=1)[:, -1]          # index order, like the kernel
    return s


def gather_summaries(summary, world, rank, dst=0):
    """Gather per-rank [B_r,16] summaries on `dst` in instance order; returns None elsewhere."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return summary
    n_local = torch.tensor([summary.shape[0]], dtype=torch.int64, device=summary.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(x.item()) for x in sizes]
    mx = max(sizes)
    pad = torch.zeros((mx, SUMMARY_WIDTH), dtype=summary.dtype, device=summary.device)
    pad[: summary.shape[0]] = summary
    bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    if dist.get_backend() == "nccl":
        # RCCL has gather; use all_gather for broad backend support of uneven tails
        bufs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)
    else:
        dist.gather(pad, gather_list=bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
