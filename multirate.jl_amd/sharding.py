"""Channel sharding across GPUs: one process per GPU, channels split contiguously, no exchange
during compute, one gather of the outputs at the end (BASELINE.json config 5).

Channels are fully independent in the reference (each FIRFilter owns its history; only the
read-only taps are shared), so rank r simply owns channels [start, start+count) and builds an
ordinary ``FIRFilter`` for them.  The only collective is the final output gather, done with
``torch.distributed`` (backend "nccl" == RCCL over xGMI on the GPU node; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, Optional


def shard_channels(nchannels: int, world_size: int, rank: int):
    """Contiguous split; the first ``nchannels % world_size`` ranks get one extra channel.
    Returns (start, count)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(nchannels, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


class ChannelShardedFilter:
    """A FIRFilter over ``nchannels`` global channels of which this rank filters its shard.

    ``filter_factory()`` must return an object with the FIRFilter interface (``filt(x)`` on a
    ``(local_channels, n)`` array/tensor).  The default builds the HIP-backed ``FIRFilter`` on this
    rank's device; tests inject their own factory to exercise the sharding and gather logic on CPU.
    """

    def __init__(self, h, ratio, nchannels: int, *, Nphi: int = 32, rank: Optional[int] = None,
                 world_size: Optional[int] = None, device: Optional[int] = None,
                 filter_factory: Optional[Callable] = None, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world_size, self.nchannels = rank, world_size, nchannels
        self.start, self.count = shard_channels(nchannels, world_size, rank)
        if filter_factory is None:
            from .host import FIRFilter

            def filter_factory():
                import torch
                dev = device if device is not None else torch.cuda.current_device()
                return FIRFilter(h, ratio, Nphi, device=dev)
        self.filter = filter_factory() if self.count > 0 else None

    def local_slice(self, x_global):
        """Rows of a (nchannels, n) global array that belong to this rank."""
        return x_global[self.start:self.start + self.count]

    def filt(self, x_local):
        """Filter this rank's channels: x_local is (count, n).  No communication."""
        if self.filter is None:
            return x_local[:, :0]
        return self.filter.filt(x_local)

    def gather(self, y_local, dst: int = 0):
        """Gather the per-rank outputs (torch tensors, (count_r, n_out)) on ``dst``; returns the
        (nchannels, n_out) tensor there and None elsewhere.  Ranks may own different channel counts,
        so shards are padded to the largest count for the collective and trimmed afterwards."""
        import torch
        dist = self._dist
        if self.world_size == 1 or not dist.is_initialized():
            return y_local
        counts = [shard_channels(self.nchannels, self.world_size, r)[1] for r in range(self.world_size)]
        cmax = max(counts)
        n_out = torch.tensor([y_local.shape[1]], dtype=torch.int64, device=y_local.device)
        dist.all_reduce(n_out, op=dist.ReduceOp.MAX, group=self.group)
        n_out = int(n_out.item())
        pad = torch.zeros((cmax, n_out), dtype=y_local.dtype, device=y_local.device)
        pad[: y_local.shape[0], : y_local.shape[1]] = y_local
        if self.rank == dst:
            parts = [torch.empty_like(pad) for _ in range(self.world_size)]
            dist.gather(pad, parts, dst=dst, group=self.group)
            return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
        dist.gather(pad, None, dst=dst, group=self.group)
        return None

    def all_gather(self, y_local):
        """Every rank receives the full (nchannels, n_out) output (ring all-gather over xGMI)."""
        import torch
        dist = self._dist
        if self.world_size == 1 or not dist.is_initialized():
            return y_local
        counts = [shard_channels(self.nchannels, self.world_size, r)[1] for r in range(self.world_size)]
        cmax = max(counts)
        pad = torch.zeros((cmax, y_local.shape[1]), dtype=y_local.dtype, device=y_local.device)
        pad[: y_local.shape[0]] = y_local
        out = torch.empty((self.world_size * cmax, y_local.shape[1]), dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        out = out.view(self.world_size, cmax, -1)
        return torch.cat([out[r, :c] for r, c in enumerate(counts)], dim=0)
