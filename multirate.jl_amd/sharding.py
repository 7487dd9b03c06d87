"""Sharding across GPUs: one process per GPU.

``ChannelShardedFilter`` (BASELINE.json config 5): channels split contiguously, no exchange
during compute, one gather of the outputs at the end.

``TimeShardedFilter``: ONE long stream (every channel of it) split along TIME.  A later ``filt!`` call sees of the earlier
ones only the stream state and the last ``tapsPerPhi - 1`` samples (``shiftin!``, src/support.jl:61-80), and the state
machine is data independent -- so rank r can enter the stream at its first sample: advance the state there without data
(``mrhip_advance_state``), take the ``tapsPerPhi - 1`` samples in front of its slice from rank r-1 (the only exchange of the
path: one point-to-point halo), and filter its slice.  The result is what a caller's chunk loop with those boundaries
produces, bit for bit; outputs are then gathered along time.

Channels are fully independent in the reference (each FIRFilter owns its history; only the
read-only taps are shared), so rank r simply owns channels [start, start+count) and builds an
ordinary ``FIRFilter`` for them.  The only collective is the final output gather, done with
``torch.distributed`` (backend "nccl" == RCCL over xGMI on the GPU node; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


def shard_channels(nchannels: int, world_size: int, rank: int):
    """Contiguous split; the first ``nchannels % world_size`` ranks get one extra channel.
    Returns (start, count)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(nchannels, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def _promoted_dtype(h, x_dtype):
    """promote_type(Th, Tx) (every filt wrapper of the reference, e.g. src/Filters.jl:581) as a torch dtype"""
    import torch
    h_f64 = np.asarray(h).dtype != np.float32
    cplx = x_dtype in (torch.complex64, torch.complex128)
    f64 = h_f64 or x_dtype in (torch.float64, torch.complex128)
    return (torch.complex128 if f64 else torch.complex64) if cplx else (torch.float64 if f64 else torch.float32)


class ChannelShardedFilter:
    """A FIRFilter over ``nchannels`` global channels of which this rank filters its shard.

    ``filter_factory()`` must return an object with the FIRFilter interface (``filt(x)`` on a
    ``(local_channels, n)`` array/tensor).  The default builds the HIP-backed ``FIRFilter`` on this
    rank's device (all constructor forms: ratio, rate + N𝜙, rate + N𝜙 + polyorder); tests inject
    their own factory to exercise the sharding and gather logic on CPU.
    """

    def __init__(self, h, ratio, nchannels: int, *, Nphi: int = 32, polyorder=None, numerics: Optional[int] = None,
                 rank: Optional[int] = None, world_size: Optional[int] = None, device: Optional[int] = None,
                 filter_factory: Optional[Callable] = None, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self._h = h
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world_size, self.nchannels = rank, world_size, nchannels
        self.start, self.count = shard_channels(nchannels, world_size, rank)
        self.counts = [shard_channels(nchannels, world_size, r)[1] for r in range(world_size)]
        if filter_factory is None:
            from .host import FIRFilter, NUMERICS_STRICT

            def filter_factory():
                import torch
                dev = device if device is not None else torch.cuda.current_device()
                return FIRFilter(h, ratio, Nphi, polyorder, device=dev,
                                 numerics=NUMERICS_STRICT if numerics is None else numerics)
        self.filter = filter_factory() if self.count > 0 else None

    def local_slice(self, x_global):
        """Rows of a (nchannels, n) global array that belong to this rank."""
        return x_global[self.start:self.start + self.count]

    def filt(self, x_local):
        """Filter this rank's channels: x_local is (count, n).  No communication."""
        if self.filter is None:    # a rank without channels: an empty shard of the PROMOTED output type
            import torch
            return torch.empty((0, 0), dtype=_promoted_dtype(self._h, x_local.dtype), device=x_local.device)
        return self.filter.filt(x_local)

    def _common_n_out(self, y_local):
        """per-channel output count, agreed over the ranks (a rank without channels has none of its own)"""
        import torch
        n_out = torch.tensor([y_local.shape[1]], dtype=torch.int64, device=y_local.device)
        self._dist.all_reduce(n_out, op=self._dist.ReduceOp.MAX, group=self.group)
        return int(n_out.item())

    @staticmethod
    def _real_view(y):
        """complex (c, n) -> real (c, 2n) view of the same memory: the collectives move plain floats"""
        import torch
        y = y.contiguous()
        return torch.view_as_real(y).reshape(y.shape[0], 2 * y.shape[1]) if y.is_complex() else y

    @staticmethod
    def _complex_view(r):
        import torch
        return torch.view_as_complex(r.reshape(r.shape[0], r.shape[1] // 2, 2))

    def gather(self, y_local, dst: int = 0, n_out: Optional[int] = None):
        """Gather the per-rank outputs (torch tensors, (count_r, n_out)) on ``dst``; returns the
        (nchannels, n_out) tensor there and None elsewhere.  When every rank owns the same number of channels the
        shards are received straight into row blocks of the result (no staging copies); otherwise shards are
        padded to the largest count for the collective and trimmed afterwards."""
        if self.world_size == 1 or not self._dist.is_initialized():
            return y_local
        if y_local.is_complex():
            r = self._gather_real(self._real_view(y_local), dst, None if n_out is None else 2 * n_out)
            return None if r is None else self._complex_view(r)
        return self._gather_real(y_local, dst, n_out)

    def _gather_real(self, y_local, dst, n_out):
        import torch
        dist = self._dist
        counts, cmax = self.counts, max(self.counts)
        if n_out is None:     # callers that know the per-channel output count (it is closed-form) save this all-reduce
            n_out = self._common_n_out(y_local)
        if min(counts) == cmax:
            y_local = y_local.contiguous()
            if self.rank == dst:
                out = torch.empty((self.nchannels, n_out), dtype=y_local.dtype, device=y_local.device)
                dist.gather(y_local, list(out.split(cmax, dim=0)), dst=dst, group=self.group)
                return out
            dist.gather(y_local, None, dst=dst, group=self.group)
            return None
        pad = torch.zeros((cmax, n_out), dtype=y_local.dtype, device=y_local.device)
        pad[: y_local.shape[0], : y_local.shape[1]] = y_local
        if self.rank == dst:
            parts = [torch.empty_like(pad) for _ in range(self.world_size)]
            dist.gather(pad, parts, dst=dst, group=self.group)
            return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
        dist.gather(pad, None, dst=dst, group=self.group)
        return None

    def all_gather(self, y_local, n_out: Optional[int] = None):
        """Every rank receives the full (nchannels, n_out) output (ring all-gather over xGMI)."""
        if self.world_size == 1 or not self._dist.is_initialized():
            return y_local
        if y_local.is_complex():
            return self._complex_view(self._all_gather_real(self._real_view(y_local), None if n_out is None else 2 * n_out))
        return self._all_gather_real(y_local, n_out)

    def _all_gather_real(self, y_local, n_out):
        import torch
        dist = self._dist
        counts, cmax = self.counts, max(self.counts)
        if n_out is None:
            n_out = self._common_n_out(y_local)
        if min(counts) == cmax:        # equal shards: gather straight into the result
            out = torch.empty((self.nchannels, n_out), dtype=y_local.dtype, device=y_local.device)
            dist.all_gather_into_tensor(out, y_local.contiguous(), group=self.group)
            return out
        pad = torch.zeros((cmax, n_out), dtype=y_local.dtype, device=y_local.device)
        pad[: y_local.shape[0], : y_local.shape[1]] = y_local
        out = torch.empty((self.world_size * cmax, n_out), dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        out = out.view(self.world_size, cmax, -1)
        return torch.cat([out[r, :c] for r, c in enumerate(counts)], dim=0)


def shard_time(n: int, world_size: int, rank: int, multiple: int = 1):
    """Contiguous split of ``n`` samples along time; slice boundaries are multiples of ``multiple`` (e.g. the decimation,
    so that every slice but the last yields whole output blocks).  Returns (start, count)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    multiple = max(1, int(multiple))
    blocks = -(-n // multiple)
    base, extra = divmod(blocks, world_size)
    b0 = rank * base + min(rank, extra)
    b1 = b0 + base + (1 if rank < extra else 0)
    start, end = min(b0 * multiple, n), min(b1 * multiple, n)
    return start, end - start


class TimeShardedFilter:
    """One stream cut into blocks of ``n_total`` samples per channel; of every block this rank filters the samples
    ``[start, start + count)``.  Successive ``filt`` calls continue the stream block after block (``reset()`` starts over).

    Per call the state enters this rank's slice without data (``advance_state`` over the samples the other ranks own) and
    the ``historyLen`` samples in front of the slice arrive from the rank before it -- for rank 0 of a later block: from
    the LAST rank's slice of the block before -- one point-to-point message per neighbour pair, a ring.  Over RCCL the halo
    stays in device memory end to end (``set_history_device``); over gloo it travels through host memory.

    ``filter_factory()`` must return an object with ``filt(x)``, ``reset()``, ``advance_state(n)``, ``set_history(h)``
    and ``historyLen`` (the HIP-backed ``FIRFilter`` by default; tests inject a CPU model).  Every slice that has a
    non-empty successor must be at least ``historyLen`` samples long (its successor's history comes from it alone).  In the
    FIRST block two owners are exempt: the first (the stream's own zero history pads its tail) and the last (its tail
    matters to a second block only -- whose call then raises).  Checked identically on every rank BEFORE any message is
    posted, so that a bad split raises everywhere instead of leaving a neighbour waiting.

    ``filt`` never resets the filter: calling it again CONTINUES the stream with the next block (``reset()`` starts over).
    """

    def __init__(self, h, ratio, n_total: int, *, Nphi: int = 32, polyorder=None, numerics: Optional[int] = None,
                 rank: Optional[int] = None, world_size: Optional[int] = None, device: Optional[int] = None,
                 multiple: int = 1, filter_factory: Optional[Callable] = None, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world_size, self.n_total = rank, world_size, int(n_total)
        self.slices = [shard_time(self.n_total, world_size, r, multiple) for r in range(world_size)]
        self.start, self.count = self.slices[rank]
        if filter_factory is None:
            from .host import FIRFilter, NUMERICS_STRICT

            def filter_factory():
                import torch
                dev = device if device is not None else torch.cuda.current_device()
                return FIRFilter(h, ratio, Nphi, polyorder, device=dev,
                                 numerics=NUMERICS_STRICT if numerics is None else numerics)
        self.filter = filter_factory()
        self.blocks = 0            # blocks filtered so far (0: the next call starts the stream)
        self._tail = None          # the last rank keeps the tail of its slice for rank 0's next block

    def local_slice(self, x_global):
        """The samples of a (..., n_total) global array that belong to this rank."""
        return x_global[..., self.start:self.start + self.count]

    def reset(self):
        """Start the stream over (every rank)."""
        self.filter.reset()
        self.blocks = 0
        self._tail = None
        return self

    def _ring(self):
        """(ranks with samples, in order): the halo travels from each to the next, and from the last to the first"""
        return [r for r in range(self.world_size) if self.slices[r][1] > 0]

    def _exchange_halo(self, x_local, H):
        """The last H samples of every owner's slice go to the next owner (point to point), the last owner's to the first
        for the NEXT block; returns this rank's history (channels, H) as a tensor in the wire's memory, or None when the
        history is the stream's own (first block of the first owner; a single rank)."""
        import torch
        dist = self._dist
        owners = self._ring()
        if self.world_size == 1 or not dist.is_initialized() or H == 0 or len(owners) < 2 or self.count == 0:
            return None
        x2 = x_local.reshape(-1, x_local.shape[-1])
        # the message lives where the backend moves it: device memory for nccl (RCCL over xGMI), host memory otherwise
        wire = x2.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
        real = (lambda t: (torch.view_as_real(t.contiguous()).reshape(t.shape[0], -1) if t.is_complex() else t.contiguous()).to(wire))
        me = owners.index(self.rank)
        first, last = me == 0, me == len(owners) - 1
        tail = x2[:, -H:]
        if tail.shape[1] < H:     # only the first owner of the first block gets here (see filt): the stream's zero history pads it
            tail = torch.cat([torch.zeros((x2.shape[0], H - tail.shape[1]), dtype=x2.dtype, device=x2.device), tail], dim=1)
        tail = real(tail)
        reqs, recv = [], None
        send_now = None
        if not last:
            send_now = tail                                  # this block's slice feeds the next owner's slice of this block
        elif self.blocks > 0:
            send_now = self._tail                            # the last owner: its PREVIOUS block's tail feeds the first owner now
        if send_now is not None:
            reqs.append(dist.isend(send_now, dst=owners[(me + 1) % len(owners)], group=self.group))
        if not first or self.blocks > 0:
            recv = torch.empty((x2.shape[0], H * (2 if x2.is_complex() else 1)), dtype=x2.real.dtype if x2.is_complex() else x2.dtype, device=wire)
            reqs.append(dist.irecv(recv, src=owners[(me - 1) % len(owners)], group=self.group))
        for r in reqs:
            r.wait()
        if last:
            # a PRIVATE copy: for one channel (or count == H) every step above returns a view of the caller's buffer, and a
            # caller that refills that buffer in place for the next block would send the wrong tail
            self._tail = tail.clone()
        if recv is None:
            return None
        return torch.view_as_complex(recv.reshape(recv.shape[0], H, 2)) if x2.is_complex() else recv

    def filt(self, x_local):
        """Filter this rank's slice ``x_local`` (..., count) of the next block of the stream.  One halo message per
        neighbour pair."""
        f = self.filter
        H = int(f.historyLen)
        if self.world_size > 1 and len(self._ring()) > 1:
            owners = self._ring()
            for r in owners:
                # exempt in the FIRST block: the first owner (zero history in front of it pads its tail) and the last owner
                # (its tail is needed by a next block only; that block's call raises on every rank if it comes)
                if self.blocks == 0 and r in (owners[0], owners[-1]):
                    continue
                if self.slices[r][1] < H:
                    raise ValueError(f"time slice of rank {r} ({self.slices[r][1]} samples) is shorter than the filter history ({H}): "
                                     f"its successor's history comes from it alone")
        if self.count == 0:
            self.blocks += 1
            return f.filt(x_local)                       # nothing to do: an empty output of the right type
        if hasattr(f, "bind") and getattr(f, "_handle", True) is None:
            import numpy as np_
            nch = 1 if x_local.ndim == 1 else int(x_local.shape[0])
            f.bind(np_.dtype(str(x_local.dtype).replace("torch.", "")), nch)
        # the stream state at this slice's first sample: over the samples the ranks before it own (first block), then
        # over everybody else's samples between this rank's slices of consecutive blocks
        skip = self.start if self.blocks == 0 else self.n_total - self.count
        if skip > 0:
            f.advance_state(skip)
        hist = self._exchange_halo(x_local, H)
        if hist is not None:
            if hist.is_cuda and hasattr(f, "set_history_device"):
                f.set_history_device(hist.contiguous())   # the halo never leaves the device (RCCL)
            else:
                f.set_history(hist.cpu().numpy())
        self.blocks += 1
        return f.filt(x_local)

    def gather(self, y_local, dst: int = 0):
        """Concatenate the per-rank outputs along time on ``dst`` (variable lengths: the counts are exchanged first);
        returns the (..., n_out_total) tensor there and None elsewhere."""
        import torch
        dist = self._dist
        if self.world_size == 1 or not dist.is_initialized():
            return y_local
        y2 = y_local.reshape(-1, y_local.shape[-1]).contiguous()
        lens = torch.zeros(self.world_size, dtype=torch.int64, device=y2.device)
        lens[self.rank] = y2.shape[1]
        dist.all_reduce(lens, group=self.group)
        lens = [int(v) for v in lens.tolist()]
        lmax = max(lens)
        cplx = y2.is_complex()
        yr = torch.view_as_real(y2).reshape(y2.shape[0], -1) if cplx else y2
        w = 2 if cplx else 1
        pad = torch.zeros((yr.shape[0], lmax * w), dtype=yr.dtype, device=yr.device)
        pad[:, : yr.shape[1]] = yr
        if self.rank == dst:
            parts = [torch.empty_like(pad) for _ in range(self.world_size)]
            dist.gather(pad, parts, dst=dst, group=self.group)
            out = torch.cat([p[:, : n * w] for p, n in zip(parts, lens)], dim=1)
            out = torch.view_as_complex(out.reshape(out.shape[0], -1, 2)) if cplx else out
            return out.reshape(*y_local.shape[:-1], out.shape[-1])
        dist.gather(pad, None, dst=dst, group=self.group)
        return None
