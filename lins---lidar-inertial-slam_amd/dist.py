"""Multi-GPU sharding of a batch of scan pairs.

Every scan pair is an independent IESKF problem, so the path shards with NO
data-path collective: rank r owns the contiguous range shard_range(n, r, world) of
the global batch.  The one exchange step is a flat all-gather of the fixed-size
192-byte pose records: on the GPU box through the C ABI (lins_pose_allgather ->
ncclAllGather over xGMI, include/lins_ieskf.h — what bench.py runs), in the CPU
tests through torch.distributed's gloo backend (PoseGatherPipeline below: the same
sharding, padding and ordering logic — ordered_records — without a GPU).  The
payload is ~190 KB per rank for 1024 scans — latency-bound, so a single flat
collective is the right shape.
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


def ordered_records(gathered_bytes, spans):
    """Cut the padding out of an all-gather of equal pieces (one per rank, each padded to the largest shard) and return
    the records of all ranks in scan order; raises when they are not.  gathered_bytes: host uint8 array."""
    host = np.ascontiguousarray(gathered_bytes, dtype=np.uint8).reshape(-1)
    max_n = max(max(b - a for a, b in spans), 1)
    stride = max_n * RECORD_BYTES
    parts = [np.frombuffer(host[r * stride: r * stride + (b - a) * RECORD_BYTES].tobytes(), dtype=POSE_DTYPE)
             for r, (a, b) in enumerate(spans)]
    rec = np.concatenate(parts) if parts else np.zeros(0, dtype=POSE_DTYPE)
    n_total = spans[-1][1] if spans else 0
    if not np.array_equal(rec["scan_id"], np.arange(n_total)):
        raise AssertionError("pose gather out of order")
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


class PoseGatherPipeline:
    """The exchange step over torch.distributed (rounds 1-2 benchmarked this; since round 3 bench.py runs the same
    double-buffered scheme inside the library — lins_set_pipelined + lins_pose_allgather — and this class is the CPU
    / gloo stand-in the world-size-2 tests drive): a double-buffered, asynchronous all-gather of the pose records.

    Step k's update writes its records into pose buffer k & 1; their gather is enqueued right after step
    k + 1's update has been launched and travels while that update computes (RCCL's own stream on the GPU
    box).  A buffer is handed out again only after the gather that read it has completed, and drain()
    issues and completes the last one.  Shards may be ragged by one record: every rank's buffer is padded
    to the largest shard (all_gather_into_tensor wants equal pieces) and `records()` cuts the padding out.

    Device-agnostic: CUDA tensors + "nccl" in bench.py, CPU tensors + "gloo" in tests/test_dist.py.
    With world == 1 and no process group it degenerates to two local buffers and no collective."""

    def __init__(self, n_total, rank, world, device="cpu", group=None, enabled=None):
        import torch

        self.torch = torch
        self.n_total, self.rank, self.world, self.group = n_total, rank, world, group
        self.spans = [shard_range(n_total, r, world) for r in range(world)]
        self.lo, self.hi = self.spans[rank]
        self.max_n = max(b - a for a, b in self.spans)
        self.enabled = (world > 1) if enabled is None else enabled
        self.cuda = torch.device(device).type == "cuda"
        nbytes = max(self.max_n, 1) * RECORD_BYTES
        self.poses = [torch.zeros(nbytes, dtype=torch.uint8, device=device) for _ in range(2)]
        self.gathered = [torch.zeros(world * nbytes, dtype=torch.uint8, device=device) for _ in range(2)] if self.enabled else None
        self.pending = [None, None]
        self.ungathered = None  # buffer of the newest finished step, not gathered yet
        self.steps = 0

    def _finish(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait()  # orders the collective before the current stream ...
            if self.cuda:
                self.torch.cuda.current_stream().synchronize()  # ... which the host then drains: buffers free
            self.pending[b] = None

    def begin_step(self):
        """-> (buffer index, pose tensor the update of this step must fill)."""
        b = self.steps & 1
        self.steps += 1
        if self.enabled:
            self._finish(b)
        return b, self.poses[b]

    def gather_newest(self):
        """Enqueue the gather of the newest finished step (call after launching the next update)."""
        import torch.distributed as dist

        b = self.ungathered
        if self.enabled and b is not None:
            self.pending[b] = dist.all_gather_into_tensor(self.gathered[b], self.poses[b], group=self.group, async_op=True)
            self.ungathered = None

    def end_step(self, b):
        """The update that fills buffer b has completed."""
        self.ungathered = b

    def drain(self):
        self.gather_newest()
        if self.enabled:
            self._finish(0), self._finish(1)

    def records(self):
        """The last step's records of ALL ranks in scan order (after drain()); checks order."""
        b = (self.steps - 1) & 1
        if not self.enabled:
            host = self.poses[b].cpu().numpy()
            return np.frombuffer(host[: (self.hi - self.lo) * RECORD_BYTES].tobytes(), dtype=POSE_DTYPE)
        return ordered_records(self.gathered[b].cpu().numpy(), self.spans)
