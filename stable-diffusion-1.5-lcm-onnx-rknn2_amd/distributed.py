"""Multi-GPU sharding of independent requests: one process per GPU over ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the MI355X node; "gloo" on CPU for tests).

The reference runs exactly one CUDA worker (server/lcm_sr_server.py:190-193; multi-GPU is a roadmap
bullet, README.md:531), so this is new design, not parity: requests are independent units (one request =
one image = one seed, backends/cuda_worker.py:210-213), so the path shards with NO collective inside the
sampler.  The single exchange step: rank 0 owns the prompt encoder, and the [N,77,768] fp16 embeddings
(118 KB per prompt) are broadcast once per batch; finished RGB8 images return to rank 0 with a gather
(or stay on each rank and leave over that GPU's own PCIe link).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous balanced split of n requests: the first n % world ranks get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


_COMM_STREAMS = {}           # device index -> the collectives' stream of this process


class comm_stream:
    """``with comm_stream(device): <collectives>`` -- every collective of this module is issued on a stream that belongs to this
    module and is NEVER captured into a hipGraph, whatever stream the caller has current.

    Why: ProcessGroupNCCL's watchdog thread polls the events of recent collectives; on HIP an event query fails ("operation not
    permitted on an event last recorded in a capturing stream") once the stream those events were recorded on is being
    captured -- and a pipeline lane's stream is captured seconds after its first use.  Round 3 saw the watchdog take the
    process down that way and moved bench.py's calls to a stream of their own; now no caller can repeat it.

    Ordering: on entry the private stream waits for what the caller's stream has enqueued so far (an event recorded there),
    on exit the caller's stream waits for the collectives; tensors made inside are marked as used by the caller's stream
    (``keep``).  A CPU device makes it a no-op (gloo).  Entering it while the caller's stream is capturing is an error: a
    collective inside a graph is exactly what this exists to prevent."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self._ctx = self._caller = self._own = None

    def __enter__(self):
        if not self.cuda:
            return self
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("collectives must not be issued while the current stream is being captured into a hipGraph")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        own = _COMM_STREAMS.get(idx)
        if own is None:
            from . import ops
            own = _COMM_STREAMS[idx] = ops.acquire_stream(torch.device("cuda", idx))     # never pooled, never handed to a lane
        self._own, self._caller = own, torch.cuda.current_stream(idx)
        ev = torch.cuda.Event()
        ev.record(self._caller)
        own.wait_event(ev)
        self._ctx = torch.cuda.stream(own)
        self._ctx.__enter__()
        return self

    def keep(self, t):
        """A tensor allocated inside the block that the caller will use on ITS stream."""
        if self.cuda and t is not None and t.is_cuda:
            t.record_stream(self._caller)
        return t

    def __exit__(self, *exc):
        if not self.cuda:
            return False
        self._ctx.__exit__(*exc)
        self._caller.wait_stream(self._own)
        return False


def _broadcast_embeddings(embeds, n, seq, dim, device, src, group):
    if dist.get_rank(group) == src:
        t = torch.as_tensor(embeds).to(device=device, dtype=torch.float16).contiguous()
        assert tuple(t.shape) == (n, seq, dim)
    else:
        t = torch.empty(n, seq, dim, dtype=torch.float16, device=device)
    dist.broadcast(t, src=src, group=group)
    return t


def broadcast_embeddings(embeds, n: int, seq: int, dim: int, device, src: int = 0, group=None):
    """Rank ``src`` passes ``embeds`` [n,seq,dim]; every rank gets the full fp16 tensor on ``device``.  Issued on this module's
    own never-captured stream (``comm_stream``); the caller's current stream waits for the result."""
    with comm_stream(device) as cs:
        return cs.keep(_broadcast_embeddings(embeds, n, seq, dim, device, src, group))


def run_sharded(generate_fn, embeds, seeds, device, gather_to: int | None = 0, group=None):
    """Generate len(seeds) images across the group.

    generate_fn(embeds_shard [k,seq,dim] fp16 on ``device``, seeds_shard list[int]) -> uint8 tensor [k,H,W,3]
    on ``device``.  Every rank calls this with the same ``seeds``; only rank 0 needs ``embeds``.
    Returns the full [n,H,W,3] uint8 tensor on rank ``gather_to`` (None elsewhere), or this rank's shard
    when ``gather_to`` is None.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(seeds)
    with comm_stream(device) as cs:                   # exchange step: the module's own stream, never the caller's
        meta = torch.zeros(2, dtype=torch.int64, device=device)
        if rank == 0:
            e = torch.as_tensor(embeds)
            meta[0], meta[1] = e.shape[1], e.shape[2]
        dist.broadcast(meta, src=0, group=group)
        allpe = cs.keep(_broadcast_embeddings(embeds, n, int(meta[0]), int(meta[1]), device, 0, group))
    lo, hi = shard_bounds(n, world, rank)
    mine = generate_fn(allpe[lo:hi], list(seeds[lo:hi])) if hi > lo else None          # the caller's stream: no collective inside
    if gather_to is None:
        return mine
    with comm_stream(device) as cs:
        # image shape from any non-empty shard
        shp = torch.zeros(3, dtype=torch.int64, device=device)
        if mine is not None:
            shp[0], shp[1], shp[2] = mine.shape[1], mine.shape[2], mine.shape[3]
        dist.all_reduce(shp, op=dist.ReduceOp.MAX, group=group)
        H, W, Cc = (int(v) for v in shp)
        kmax = shard_bounds(n, world, 0)[1]
        pad = torch.zeros(kmax, H, W, Cc, dtype=torch.uint8, device=device)
        if mine is not None:
            pad[:hi - lo].copy_(mine)
        bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == gather_to else None
        dist.gather(pad, bufs, dst=gather_to, group=group)
        if rank != gather_to:
            return None
        parts = []
        for r in range(world):
            a, b = shard_bounds(n, world, r)
            parts.append(bufs[r][:b - a])
        return cs.keep(torch.cat(parts, 0))
