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

A program that is ONE process with a context per GPU (cgo) does the same exchange behind
the C ABI: hzsdr_mgpu_beamform (csrc/hz_mgpu.hip; `MultiGpu` in this package), ordered
all-to-all by peer copies or ncclReduce.  This module is the one-process-per-GPU form the
benchmark contract asks for (torch.distributed ranks).

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


def reduce_fast(dist, torch, partial, dst=0, group=None):
    """Sum complex64 partials elementwise onto rank `dst` (float32 lanes).  `group`: a
    process group whose members are global ranks 0 .. g-1 (sub_groups), so group rank ==
    global rank and `dst` / peers below need no translation."""
    view = torch.view_as_real(partial)
    if _staged(dist, view):
        h = view.cpu()
        dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
        view.copy_(h)
    else:
        dist.reduce(view, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return partial


def ordered_pipeline(dist, rank, world, out, partial_fn, n_slices=8, group=None):
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
                dist.recv(h, src=rank - 1, group=group)
                view.copy_(h)
            else:
                dist.recv(view, src=rank - 1, group=group)
        partial_fn(lo, hi, rank > 0)
        if rank < world - 1:
            if staged:
                torch.cuda.current_stream().synchronize()
                dist.send(view.cpu(), dst=rank + 1, group=group)
            else:
                reqs.append(dist.isend(view, dst=rank + 1, group=group))
    for r in reqs:
        r.wait()
    return out


def slice_bounds(n, world, s):
    """Samples [lo, hi) of slice s when n samples are cut into `world` slices."""
    return n * s // world, n * (s + 1) // world


def _exchange(dist, torch, sends, recvs, group=None):
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
        reqs = [dist.irecv(h, src=p, group=group) for h, p in hr] + [dist.isend(h, dst=p, group=group) for h, p in hs]
        for r in reqs:
            r.wait()
        for (t, _), (h, _) in zip(recvs, hr):
            t.copy_(h)
        return
    ops = ([dist.P2POp(dist.irecv, t, p, group=group) for t, p in recvs] +
           [dist.P2POp(dist.isend, t, p, group=group) for t, p in sends])
    for r in dist.batch_isend_irecv(ops):
        r.wait()


def ordered_alltoall(dist, torch, rank, world, k_total, weighted, first_channel, out, sum_fn, gather_dst=0,
                     group=None):
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
    _exchange(dist, torch, sends, recvs, group)
    if hi > lo:
        sum_fn(out[lo:hi], pieces)
    if gather_dst is not None:
        if rank == gather_dst:
            rv = []
            for s in range(world):
                a, b = slice_bounds(n, world, s)
                if s != rank and b > a:
                    rv.append((torch.view_as_real(out[a:b]), s))
            _exchange(dist, torch, [], rv, group)
        elif hi > lo:
            _exchange(dist, torch, [(torch.view_as_real(out[lo:hi]), gather_dst)], [], group)
    return lo, hi


BEAMFORM_CHANNELS = 4  # the metric: "4-ch Beamform @1/2/4 GPU" (BASELINE.json)


def sub_group_sizes(world):
    """The rank counts the 4-channel Beamform is measured on inside a `world`-rank job:
    1, 2 and 4 (those that fit).  The sub-group of size g is global ranks 0 .. g-1."""
    return [g for g in (1, 2, 4) if g <= world]


def sub_groups(dist, world):
    """{g: process group of ranks 0..g-1} for every measured size; g == 1 needs no group
    (None).  Collective: EVERY rank of the job must call this, in this order
    (torch.distributed.new_group's contract)."""
    groups = {}
    for g in sub_group_sizes(world):
        groups[g] = None if g == 1 else (dist.group.WORLD if g == world else dist.new_group(ranks=list(range(g))))
    return groups


def all_ok(dist, torch, ok, group, device):
    """True only if `ok` is true on every rank of `group` (an all-reduced flag: either every
    rank goes on to time a method or none does -- ranks cannot diverge)."""
    if group is None:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def bench_beamform(hz, ctx, torch, dist, rank, world, n, steps, warmup, synth):
    """The 4-channel coherent c64 Beamform (stream/beamform.go:148-171; the ordered sum of
    stream/add.go:115-119) on sub-groups of 1, 2 and 4 ranks of this job, channels sharded
    contiguously, one exchange per buffer.  Returns the JSON sub-object bench.py attaches:
    {"1": {...}, "2": {...}, "4": {...}} with local and exchange times separately.  Every
    rank of the job must call it (sub-group creation and the world barriers between sizes
    are collective); ranks outside a sub-group wait at the barrier."""
    k = BEAMFORM_CHANNELS
    nccl = dist.is_initialized() and dist.get_backend() == "nccl"
    flag_dev = "cuda" if nccl else "cpu"
    groups = sub_groups(dist, world) if world > 1 else {1: None}
    res = {"channels": k, "samples_per_channel": n,
           "schedule": {str(g): list(range(g)) for g in groups}}
    for g, group in groups.items():
        if rank < g:
            res[str(g)] = _bench_group(hz, ctx, torch, dist, rank, g, group, k, n, steps, warmup, synth, flag_dev)
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
    return res


def _bench_group(hz, ctx, torch, dist, rank, g, group, k, n, steps, warmup, synth, flag_dev):
    lo, hi = shard_channels(k, g, rank)
    chans = [torch.from_numpy(synth(5 + c, n)).cuda() for c in range(lo, hi)]
    weights = hz.beamform_angles(433e6, 30.0, [0.1 * c for c in range(k)])
    my_w = weights[lo:hi]
    out = torch.zeros(n, dtype=torch.complex64, device="cuda")
    weighted = [torch.zeros(n, dtype=torch.complex64, device="cuda") for _ in chans] if g > 1 else []

    def local():
        ctx.beamform(out, chans, my_w)

    def fast():
        local()
        reduce_fast(dist, torch, out, dst=0, group=group)

    def ordered():
        def part(a, b, acc):
            ctx.beamform(out[a:b], [c[a:b] for c in chans], my_w, accumulate=acc)
        ordered_pipeline(dist, rank, g, out, part, group=group)

    def alltoall():
        for y, x, w in zip(weighted, chans, my_w):
            ctx.beamform(y, [x], [w])  # 0 + w_c * x_c, one channel
        ordered_alltoall(dist, torch, rank, g, k, weighted, lo, out,
                         lambda dst, pieces: ctx.sum(dst, pieces), gather_dst=0, group=group)

    def timed(fn):
        """ms per buffer: max over the sub-group's ranks of the wall time between barriers."""
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier(group=group)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier(group=group)
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if group is not None:
            t = torch.tensor([el], dtype=torch.float64, device=flag_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            el = float(t.item())
        return el / steps * 1e3

    def row(ms):
        return {"ms_per_buffer": round(ms, 4), "Msamples_per_s": round(k * n / (ms * 1e-3) / 1e6, 1)}

    res = {"ranks": g, "channels_per_gpu": hi - lo, "unit_note": "input samples over all channels per second"}
    local_ms = timed(local)
    res["local_partial"] = row(local_ms)
    if g == 1:
        res["total"] = row(local_ms)
        return res
    for name, fn in (("rccl_reduce", fast), ("ordered_pipeline", ordered), ("ordered_alltoall", alltoall)):
        # one untimed call decides, on all ranks together, whether the method is timed at all
        err = None
        try:
            fn()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
        if not all_ok(dist, torch, err is None, group, flag_dev):
            # every rank's own reason, by rank (a method that fails on rank 3 alone must say so on rank 0's line)
            reasons = [None] * g
            try:
                dist.all_gather_object(reasons, err, group=group)
            except Exception as e:  # noqa: BLE001  (the collective that would carry the reasons is itself what is broken)
                reasons = [err if r == rank else f"unknown (all_gather_object failed here: {type(e).__name__})" for r in range(g)]
            res[name] = {"error": err or "failed on another rank",
                         "errors_by_rank": {str(r): reasons[r] for r in range(g) if reasons[r]}}
            continue
        ms = timed(fn)
        res[name] = row(ms)
        res[name]["exchange_ms"] = round(max(ms - local_ms, 0.0), 4)
    return res
