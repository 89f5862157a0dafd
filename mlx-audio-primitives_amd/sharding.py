"""Clip-sharded multi-GPU execution: every op on the hot path is per
clip, so a batch is cut into contiguous per-rank slices with NO data-path collective.
The reference has no distributed layer at all; this is the MI355X-side driver for
BASELINE config 5 (4096 clips over 8 GPUs, one process per GPU).

Only two things ever cross ranks, both optional and outside the timed feature
extraction:
  * ``gather_clips``  — final all-gather of the per-rank outputs (RCCL over xGMI when the
    backend is "nccl"; ~492 MB per peer link for config 5, reported separately);
  * ``global_max``    — the one true cross-shard dependency: ``power_to_db(top_db=...)``
    clips against the max of the WHOLE batch (convert.py:58), so a sharded mfcc needs a
    4-byte MAX all-reduce between the log and the clip.
Works with any torch.distributed backend (tests run it on gloo/CPU with world_size 2).
"""

from __future__ import annotations

import torch


def shard_range(n_clips: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of the batch owned by `rank`; sizes differ by at most 1
    and concatenating the slices in rank order restores the batch order."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size: {rank}/{world_size}")
    if n_clips < 0:
        raise ValueError(f"n_clips must be non-negative, got {n_clips}")
    base, extra = divmod(n_clips, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_clips(batch: torch.Tensor, rank: int | None = None, world_size: int | None = None):
    """This rank's clips of a (B, ...) batch (a view, no copy)."""
    rank, world_size = _rank_world(rank, world_size)
    lo, hi = shard_range(batch.shape[0], rank, world_size)
    return batch[lo:hi]


def gather_clips(local: torch.Tensor, n_clips: int, group=None) -> torch.Tensor:
    """All-gather per-rank outputs (b_r, ...) back into the (n_clips, ...) batch order."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_range(n_clips, r, world)[1] - shard_range(n_clips, r, world)[0] for r in range(world)]
    biggest = max(sizes)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def global_max(local: torch.Tensor, group=None) -> torch.Tensor:
    """max over every rank's tensor (0-dim tensor on local.device)."""
    import torch.distributed as dist

    m = local.max().reshape(1).clone() if local.numel() else \
        torch.full((1,), float("-inf"), dtype=local.dtype, device=local.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return m[0]


def global_max_key(key: torch.Tensor, group=None) -> torch.Tensor:
    """In-place MAX all-reduce of the 4-byte order-preserving key the mel / reduce kernels raise
    for max(S) (csrc: ap_fkey — keys compare as UNSIGNED 32-bit integers, the tensor holding one
    is int32).  This is the single cross-shard dependency of the path: a clip-sharded
    ``mfcc`` / ``power_to_db(top_db=...)`` clips against the max of the WHOLE batch
    (convert.py:58).  ``mfcc_sharded`` calls it between the mel kernel and the dB + DCT
    kernel; world_size 1 / no process group is a no-op."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return key
    wide = key.to(torch.int64) & 0xFFFFFFFF                  # unsigned order in a signed dtype
    dist.all_reduce(wide, op=dist.ReduceOp.MAX, group=group)
    key.copy_(wide.to(torch.int32))                          # wraps back to the same 32 bits
    return key


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX all-reduce of a host scalar (bench.py: slowest rank's time)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0])


def _rank_world(rank, world_size):
    import torch.distributed as dist

    if rank is None or world_size is None:
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1
    return rank, world_size


def mfcc_sharded(y, group=None, *, _max_reduce=None, **kw) -> torch.Tensor:
    """`mfcc` of this rank's clips when the batch is sharded by clip over the ranks of `group` (None = the
    default process group): the max(S) the dB stage clips against (global over the whole batch, convert.py:58)
    is MAX-all-reduced - 4 bytes - between the mel kernel and the dB + DCT kernel, so every rank produces
    exactly the rows the unsharded call would.  Keyword arguments are `mfcc`'s; world size 1 / no process group
    is plain `mfcc`.  ``_max_reduce`` (tests): a callable on the 1-element int32 key tensor instead of the
    all-reduce."""
    import importlib
    import inspect

    _m = importlib.import_module(__package__ + ".mfcc")      # (the package exports the function under the same name)

    names = [p for p in inspect.signature(_m.mfcc).parameters][1:]
    defaults = {n: p.default for n, p in inspect.signature(_m.mfcc).parameters.items()}
    unknown = set(kw) - set(names)
    if unknown:
        raise TypeError(f"mfcc_sharded() got unexpected keyword arguments {sorted(unknown)}")
    args = [kw.get(n, defaults[n]) for n in names]
    reduce = _max_reduce if _max_reduce is not None else (lambda key: global_max_key(key, group))
    return _m._mfcc_impl(y, *args, reduce)
