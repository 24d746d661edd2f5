"""The N > 1 Beamform path on CPU: world_size 2 (and 4) over gloo.  The exchange
logic is the product's (go-sdr_amd/multigpu.py); the per-rank partial sums,
which the HIP kernel computes on the GPU box, are supplied here by the oracle
so the sharding + exchange can be proven bit-exact without a GPU."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from util import rand_c64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, k, result_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    mg = importlib.import_module("go-sdr_amd.multigpu")
    hz = importlib.import_module("go-sdr_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        weights = hz.beamform_angles(433e6, 30.0, [0.1 * c for c in range(k)])
        lo, hi = mg.shard_channels(k, world, rank)
        chans = [rand_c64(20 + c, n) for c in range(lo, hi)]
        out = torch.zeros(n, dtype=torch.complex64)
        out_np = out.numpy()

        def part(a, b, acc):
            tmp = np.zeros(b - a, np.complex64)
            # continue the ordered sum: start from the running value when accumulating
            if acc:
                # oracle beamform starts from +0: add channel by channel onto the running sum
                run = out_np[a:b].copy()
                for x, w in zip(chans, weights[lo:hi]):
                    y = x[a:b].copy()
                    if w != 1:
                        orc.rotate(y, w)
                    t = np.zeros(b - a, np.complex64)
                    orc.add(run, y, t)
                    run = t
                out_np[a:b] = run
            else:
                orc.beamform(tmp, [x[a:b].copy() for x in chans], weights[lo:hi])
                out_np[a:b] = tmp

        mg.ordered_pipeline(dist, rank, world, out, part, n_slices=5)
        if rank == world - 1:
            np.save(os.path.join(result_dir, "ordered.npy"), out_np)

        # fast path: per-rank partial from +0, then reduce(SUM) to rank 0
        fast = torch.zeros(n, dtype=torch.complex64)
        tmp = np.zeros(n, np.complex64)
        orc.beamform(tmp, chans, weights[lo:hi])
        fast.numpy()[:] = tmp
        mg.reduce_fast(dist, torch, fast, dst=0)
        if rank == 0:
            np.save(os.path.join(result_dir, "fast.npy"), fast.numpy())

        # fixed-order all-to-all of slices: every rank ships slice s of each of its weighted
        # channels (0 + w*x) to rank s, which adds all k pieces in channel order
        weighted = []
        for x, w in zip(chans, weights[lo:hi]):
            y = np.zeros(n, np.complex64)
            orc.beamform(y, [x], [w])
            weighted.append(torch.from_numpy(y))
        a2a = torch.zeros(n, dtype=torch.complex64)

        def sum_fn(dst, pieces):
            d = dst.numpy()
            tmp2 = np.zeros(d.shape[0], np.complex64)
            orc.sum_(tmp2, [np.ascontiguousarray(p.numpy()) for p in pieces])
            d[:] = tmp2

        mg.ordered_alltoall(dist, torch, rank, world, k, weighted, lo, a2a, sum_fn, gather_dst=0)
        if rank == 0:
            np.save(os.path.join(result_dir, "alltoall.npy"), a2a.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k", [(2, 4), (4, 4), (2, 5), (3, 7)])
def test_sharded_beamform_exchange(orc, tmp_path, world, k):
    n = 10_007
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, k, str(tmp_path)), nprocs=world, join=True)
    hz = importlib.import_module("go-sdr_amd")
    weights = hz.beamform_angles(433e6, 30.0, [0.1 * c for c in range(k)])
    want = np.zeros(n, np.complex64)
    orc.beamform(want, [rand_c64(20 + c, n) for c in range(k)], weights)
    ordered = np.load(tmp_path / "ordered.npy")
    assert ordered.tobytes() == want.tobytes()  # fixed order: bit-exact with the 1-GPU / reference sum
    a2a = np.load(tmp_path / "alltoall.npy")
    assert a2a.tobytes() == want.tobytes()  # slices summed in channel order: bit-exact too
    fast = np.load(tmp_path / "fast.npy")
    assert np.allclose(fast, want, rtol=0, atol=4e-6)  # reduce(SUM): order differs, value agrees


def test_shard_channels_is_an_ordered_partition():
    mg = importlib.import_module("go-sdr_amd.multigpu")
    for k in (4, 5, 8, 16):
        for world in (1, 2, 3, 4, 8):
            spans = [mg.shard_channels(k, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == k
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
