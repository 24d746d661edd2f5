"""Beamform sharded over GPUs: one process per GPU, channels
split contiguously over ranks, each rank runs its channels' weighted partial sum
locally (hzsdr_beamform_partial) and ONE exchange step combines them.

Three exchanges, all plain torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests):

  ordered_alltoall rank s owns slice s of the output: every rank sends slice s of
                   each of its weighted channels to rank s, which adds all K pieces in
                   channel order from +0, then the slices are collected.  Bit-identical
                   to the single-GPU / reference result, and each xGMI link carries
                   1/world of a channel (SURVEY 8e's parity path).

  reduce_fast      dist.reduce(SUM) of the per-rank partial sums onto rank 0.
                   Fast path; the summation order is RCCL's, so the result is
                   within a few ULP of the reference, not bit-identical.
  ordered_pipeline rank r receives the running sum from rank r-1, continues the
                   reference's left-to-right accumulation with its own channels
                   (stream/add.go:115-119 order) and forwards it; sliced so hops
                   overlap.  Bit-identical to the single-GPU / reference result;
                   the final sum lands on the LAST rank.

Nothing here touches sample values on the host; `partial_fn` is the only thing
that computes, and on the GPU box it is the HIP kernel.  (Only when the backend
is gloo AND the tensors live on a GPU -- a 1-GPU debugging set-up -- are the
messages staged through host memory, because gloo has no device p2p.)
"""
import time

import numpy as np


def shard_channels(n_channels, world, rank):
    """Contiguous channel range [lo, hi) owned by `rank` (order-preserving)."""
    lo = n_channels * rank // world
    hi = n_channels * (rank + 1) // world
    return lo, hi


def _staged(dist, t):
    return t.is_cuda and dist.get_backend() == "gloo"


def reduce_fast(dist, torch, partial, dst=0):
    """Sum complex64 partials elementwise onto rank `dst` (float32 lanes)."""
    view = torch.view_as_real(partial)
    if _staged(dist, view):
        h = view.cpu()
        dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM)
        view.copy_(h)
    else:
        dist.reduce(view, dst=dst, op=dist.ReduceOp.SUM)
    return partial


def ordered_pipeline(dist, rank, world, out, partial_fn, n_slices=8):
    """Fixed-order accumulation across ranks.  partial_fn(lo, hi, accumulate)
    must do out[lo:hi] = (out[lo:hi] if accumulate else 0) + sum of this rank's
    weighted channels over samples [lo, hi), left to right."""
    import torch
    n = out.shape[0]
    bounds = [n * s // n_slices for s in range(n_slices + 1)]
    reqs = []
    for s in range(n_slices):
        lo, hi = bounds[s], bounds[s + 1]
        if hi == lo:
            continue
        view = torch.view_as_real(out[lo:hi])
        staged = _staged(dist, view)
        if rank > 0:
            if staged:
                h = torch.empty(view.shape, dtype=view.dtype)
                dist.recv(h, src=rank - 1)
                view.copy_(h)
            else:
                dist.recv(view, src=rank - 1)
        partial_fn(lo, hi, rank > 0)
        if rank < world - 1:
            if staged:
                torch.cuda.current_stream().synchronize()
                dist.send(view.cpu(), dst=rank + 1)
            else:
                reqs.append(dist.isend(view, dst=rank + 1))
    for r in reqs:
        r.wait()
    return out


def slice_bounds(n, world, s):
    """Samples [lo, hi) of slice s when n samples are cut into `world` slices."""
    return n * s // world, n * (s + 1) // world


def _exchange(dist, torch, sends, recvs):
    """sends / recvs: lists of (tensor_view_as_real, peer).  One grouped batch on NCCL
    (ncclGroupStart/End, no ordering deadlock); plain non-blocking ops on gloo, staged
    through host memory when the tensors live on a GPU (1-GPU debugging only)."""
    if not sends and not recvs:
        return
    staged = any(_staged(dist, t) for t, _ in sends + recvs)
    if staged:
        torch.cuda.current_stream().synchronize()
        hs = [(t.cpu(), p) for t, p in sends]
        hr = [(torch.empty(t.shape, dtype=t.dtype), p) for t, p in recvs]
        reqs = [dist.irecv(h, src=p) for h, p in hr] + [dist.isend(h, dst=p) for h, p in hs]
        for r in reqs:
            r.wait()
        for (t, _), (h, _) in zip(recvs, hr):
            t.copy_(h)
        return
    ops = [dist.P2POp(dist.irecv, t, p) for t, p in recvs] + [dist.P2POp(dist.isend, t, p) for t, p in sends]
    for r in dist.batch_isend_irecv(ops):
        r.wait()


def ordered_alltoall(dist, torch, rank, world, k_total, weighted, first_channel, out, sum_fn, gather_dst=0):
    """The fixed-order exchange of SURVEY 8e as an all-to-all of sample-range slices: rank s
    owns slice s of the output; every rank sends slice s of each of ITS weighted channels
    (`weighted[i]` = 0 + w_c * x_c for channel c = first_channel + i, complex64, n samples)
    to rank s, which adds the k_total pieces in channel order from +0 (`sum_fn(out_slice,
    pieces)`: stream/add.go:115-119) -- bit-identical to the one-GPU / reference sum, with
    every link carrying 1/world of a channel instead of a whole running sum.  The slices
    are then collected on `gather_dst` (None: leave them distributed).  Channels must be
    sharded contiguously (shard_channels)."""
    n = out.shape[0]
    lo, hi = slice_bounds(n, world, rank)
    owner_of = []  # channel -> owning rank
    for r in range(world):
        a, b = shard_channels(k_total, world, r)
        owner_of += [r] * (b - a)
    pieces = []
    recvs, sends = [], []
    for c in range(k_total):
        r = owner_of[c]
        if r == rank:
            pieces.append(weighted[c - first_channel][lo:hi])
        else:
            buf = torch.empty(hi - lo, dtype=out.dtype, device=out.device)
            pieces.append(buf)
            if hi > lo:
                recvs.append((torch.view_as_real(buf), r))
    for s in range(world):
        if s == rank:
            continue
        a, b = slice_bounds(n, world, s)
        if b > a:
            for w in weighted:
                sends.append((torch.view_as_real(w[a:b]), s))
    _exchange(dist, torch, sends, recvs)
    if hi > lo:
        sum_fn(out[lo:hi], pieces)
    if gather_dst is not None:
        if rank == gather_dst:
            rv = []
            for s in range(world):
                a, b = slice_bounds(n, world, s)
                if s != rank and b > a:
                    rv.append((torch.view_as_real(out[a:b]), s))
            _exchange(dist, torch, [], rv)
        elif hi > lo:
            _exchange(dist, torch, [(torch.view_as_real(out[lo:hi]), gather_dst)], [])
    return lo, hi


def bench_beamform(hz, ctx, torch, dist, rank, world, n, steps, warmup, synth):
    """4-channel (or `world`-channel when world > 4) coherent c64 beamform, one
    exchange per buffer.  Returns the JSON sub-object bench.py attaches."""
    k = max(4, world)
    lo, hi = shard_channels(k, world, rank)
    chans = [torch.from_numpy(synth(5 + c, n)).cuda() for c in range(lo, hi)]
    dists = [0.1 * c for c in range(k)]
    weights = hz.beamform_angles(433e6, 30.0, dists)
    my_w = weights[lo:hi]
    out = torch.zeros(n, dtype=torch.complex64, device="cuda")
    gpu_barrier = dist.get_backend() == "nccl"

    def fast():
        if chans:
            ctx.beamform(out, chans, my_w)
        else:
            out.zero_()
        reduce_fast(dist, torch, out, dst=0)

    def ordered():
        def part(a, b, acc):
            if chans:
                ctx.beamform(out[a:b], [c[a:b] for c in chans], my_w, accumulate=acc)
            elif not acc:
                out[a:b].zero_()
        ordered_pipeline(dist, rank, world, out, part)

    weighted = [torch.zeros(n, dtype=torch.complex64, device="cuda") for _ in chans]

    def alltoall():
        for y, x, w in zip(weighted, chans, my_w):
            ctx.beamform(y, [x], [w])  # 0 + w_c * x_c, one channel
        ordered_alltoall(dist, torch, rank, world, k, weighted, lo, out,
                         lambda dst, pieces: ctx.sum(dst, pieces), gather_dst=0)

    res = {"channels": k, "samples_per_channel": n, "channels_per_gpu": hi - lo}
    # local compute alone (no exchange) first, to show the exchange cost separately
    for _ in range(warmup):
        if chans:
            ctx.beamform(out, chans, my_w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if chans:
            ctx.beamform(out, chans, my_w)
    torch.cuda.synchronize()
    res["local_partial_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 4)

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                          device="cuda" if gpu_barrier else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()) / steps * 1e3

    # a method that raises (on every rank alike: an API the backend lacks) is reported, not fatal
    for name, fn in (("rccl_reduce", fast), ("ordered_pipeline", ordered), ("ordered_alltoall", alltoall)):
        try:
            ms = timed(fn)
            res[name] = {"ms_per_buffer": round(ms, 4),
                         "Msamples_per_s": round(k * n / (ms * 1e-3) / 1e6, 1),
                         "unit_note": "input samples over all channels per second"}
        except Exception as e:  # noqa: BLE001
            res[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res
