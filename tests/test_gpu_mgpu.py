"""hzsdr_mgpu_*: Beamform sharded over several contexts of ONE process (what a Go program
has).  A 1-GPU box exercises the exchange by opening K shards on device 0 (the peer copies
degenerate to device-to-device copies; the schedule, the slices and the ordered sum are the
same code).  stream/beamform.go:148-171, stream/add.go:115-119."""
import importlib

import numpy as np
import pytest

from util import bits_equal, rand_c64, rand_i16, rand_u8, zeros

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hz():
    return importlib.import_module("go-sdr_amd")


def _run(hz, orc, torch, g, k, n, fmt, mode, dst):
    gen = {"c64": rand_c64, "u8": rand_u8, "i16": rand_i16}[fmt]
    chans = [gen(30 + c, n) for c in range(k)]
    weights = hz.beamform_angles(433e6, 30.0, [0.1 * c for c in range(k)])
    conv = []
    for x in chans:  # the per-channel ConvertReader of stream/beamform.go:151
        y = zeros("c64", n)
        if fmt == "c64":
            y[:] = x
        else:
            orc.convert(y, x)
        conv.append(y)
    want = zeros("c64", n)
    orc.beamform(want, conv, weights)
    mg = hz.MultiGpu([0] * g)
    try:
        # channel c goes to its owner's context (all on cuda:0 here)
        dev = [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in chans]
        out = torch.zeros(n, dtype=torch.complex64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):  # twice: the second call reuses scratch the first one's sums read
            mg.beamform(out, dev, weights, dst_shard=dst, mode=mode)
        mg.synchronize()
        got = out.cpu().numpy()
    finally:
        mg.close()
    return got, want


@pytest.mark.parametrize("g,k", [(2, 4), (4, 4), (3, 7), (4, 2), (1, 4)])
@pytest.mark.parametrize("fmt", ["c64", "u8"])
def test_ordered_exchange_is_bit_identical(hz, orc, g, k, fmt):
    import torch
    for n, dst in ((100_003, 0), (4096, g - 1), (3, 0)):
        got, want = _run(hz, orc, torch, g, k, n, fmt, hz.MGPU_ORDERED, dst)
        assert bits_equal(got, want), (g, k, fmt, n, dst)


def test_shard_channels_matches_the_python_partition(hz):
    mg = importlib.import_module("go-sdr_amd.multigpu")
    for k in (1, 4, 5, 16):
        for g in (1, 2, 3, 4, 8):
            for s in range(g):
                assert hz.MultiGpu.shard_channels(k, g, s) == mg.shard_channels(k, g, s)


def test_rccl_path_single_rank_and_duplicate_gpus(hz, orc):
    """One shard: ncclCommInitAll over one GPU and a one-rank ncclReduce really run (librccl
    loaded at run time); several shards on ONE GPU are refused (RCCL wants distinct GPUs)."""
    import torch
    got, want = _run(hz, orc, torch, 1, 4, 50_000, "c64", hz.MGPU_RCCL, 0)
    assert np.allclose(got, want, rtol=0, atol=4e-6)
    with pytest.raises(hz.ErrInvalidArgument):
        _run(hz, orc, torch, 2, 4, 1000, "c64", hz.MGPU_RCCL, 0)


def test_distinct_gpus_ordered_and_rccl(hz, orc):
    """The real multi-device paths -- hipMemcpyPeerAsync over xGMI, cross-device event ordering,
    ncclCommInitAll / ncclReduce over distinct GPUs: runs only where the process sees two or more
    GPUs (the 1-GPU test boxes skip it; until it has run somewhere these paths are UNTESTED, as
    DESIGN.md says)."""
    import torch
    g = min(torch.cuda.device_count(), 4)
    if g < 2:
        pytest.skip("needs two or more GPUs in one process")
    k, n = 8, 200_003
    chans = [rand_c64(60 + c, n) for c in range(k)]
    weights = hz.beamform_angles(433e6, 30.0, [0.1 * c for c in range(k)])
    want = zeros("c64", n)
    orc.beamform(want, chans, weights)
    for mode in (hz.MGPU_ORDERED, hz.MGPU_RCCL):
        mg = hz.MultiGpu(list(range(g)))
        try:
            dev = []
            for c, x in enumerate(chans):  # channel c lives on its owner's GPU
                owner = next(s for s in range(g) if hz.MultiGpu.shard_channels(k, g, s)[0] <= c < hz.MultiGpu.shard_channels(k, g, s)[1])
                dev.append(torch.from_numpy(np.ascontiguousarray(x)).to("cuda:%d" % owner))
            out = torch.zeros(n, dtype=torch.complex64, device="cuda:0")
            for d in range(g):
                torch.cuda.synchronize(d)
            for _ in range(2):
                mg.beamform(out, dev, weights, dst_shard=0, mode=mode)
            mg.synchronize()
            got = out.cpu().numpy()
        finally:
            mg.close()
        if mode == hz.MGPU_ORDERED:
            assert bits_equal(got, want)
        else:
            assert np.allclose(got, want, rtol=0, atol=4e-6)


def test_two_rank_bench_on_one_gpu_runs_every_multi_rank_code_path():
    """`bench.py --gpus 2` with both ranks on cuda:0 over gloo (HZ_BENCH_SAME_DEVICE=1 HZ_BENCH_BACKEND=gloo): a fresh
    child process that starts torch.distributed.run as ITS child -- never a re-exec of a process that has touched the
    GPU.  Not a measurement: what it proves is that nothing of the N > 1 path is left to discover on the day an 8-GPU
    node runs it -- the launcher, the sub-group schedule, the three exchange methods with their exchange times, the
    `rccl` object with the all-reduce of rank + 1, the one JSON line and the exit code."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HZ_BENCH_SAME_DEVICE="1", HZ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--ramp-ms", "0",
                        "--log2n", "20", "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] and d["scaling"] == "weak" and d["parity"]["ok"]
    assert d["rccl"]["world_size"] == 2 and d["rccl"]["allreduce_sum_of_rank_plus_1"] == 3 and d["rccl"]["backend"] == "gloo"
    bf = d["beamform"]
    assert bf["schedule"] == {"1": [0], "2": [0, 1]}
    assert bf["1"]["total"]["ms_per_buffer"] > 0
    two = bf["2"]
    assert two["ranks"] == 2 and two["channels_per_gpu"] == 2 and two["local_partial"]["ms_per_buffer"] > 0
    for method in ("rccl_reduce", "ordered_pipeline", "ordered_alltoall"):
        assert "error" not in two[method], (method, two[method])
        assert two[method]["exchange_ms"] >= 0 and two[method]["ms_per_buffer"] > 0
