"""Multi-GPU sharding of a batch of scan pairs.

Every scan pair is an independent IESKF problem, so the path shards with NO
data-path collective: rank r owns the contiguous range shard_range(n, r, world) of
the global batch.  The one exchange step is a flat all-gather of the fixed-size
192-byte pose records (torch.distributed: backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  The payload is ~190 KB per rank for 1024 scans
— latency-bound, so a single flat collective is the right shape.
"""
import numpy as np

from ._ctypes_defs import POSE_DTYPE

RECORD_BYTES = POSE_DTYPE.itemsize  # 192


def shard_range(n_total, rank, world):
    """Contiguous [lo, hi) of rank `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def records_from_results(results, scan_id_base):
    """Pack host-side Result objects into pose records (what the kernel writes on device)."""
    rec = np.zeros(len(results), dtype=POSE_DTYPE)
    for i, r in enumerate(results):
        rec[i]["state"] = r.state
        rec[i]["residual_norm"] = r.residual_norm
        rec[i]["iters"], rec[i]["converged"], rec[i]["diverged"] = r.iters, r.converged, r.diverged
        rec[i]["m_surf"], rec[i]["m_corner"] = r.m_surf, r.m_corner
        rec[i]["scan_id"] = scan_id_base + i
    return rec


def gather_pose_records(local_bytes, n_total, group=None):
    """All-gather the ranks' pose-record buffers (uint8 torch tensors, possibly ragged by
    one record) and return an (n_total,) POSE_DTYPE array ordered by scan id."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    assert local_bytes.numel() == (hi - lo) * RECORD_BYTES
    padded = torch.zeros(max_n * RECORD_BYTES, dtype=torch.uint8, device=local_bytes.device)
    padded[: local_bytes.numel()] = local_bytes
    out = torch.empty(world * max_n * RECORD_BYTES, dtype=torch.uint8, device=local_bytes.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    host = out.cpu().numpy()
    parts = []
    for r, (a, b) in enumerate(sizes):
        seg = host[r * max_n * RECORD_BYTES: r * max_n * RECORD_BYTES + (b - a) * RECORD_BYTES]
        parts.append(np.frombuffer(seg.tobytes(), dtype=POSE_DTYPE))
    rec = np.concatenate(parts)
    assert np.array_equal(rec["scan_id"], np.arange(n_total)), "pose records out of order"
    return rec
