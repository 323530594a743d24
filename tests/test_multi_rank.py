"""N>1 path on CPU: world_size-2 gloo run of the sharded pipeline (emulator) == single-process oracle, in locus order."""
import json
import os
import subprocess
import sys

import pytest

from manta_amd.shard import shard_bounds
from oracle_lib import asm_opts
from synth import small_indel_locus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    costs = [5, 1, 1, 1, 8, 2, 2, 4]
    for world in (1, 2, 3, 8, 11):
        cuts = [shard_bounds(costs, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == len(costs)
        for a, b in zip(cuts, cuts[1:]):
            assert a[1] == b[0] and a[0] <= a[1]
    assert shard_bounds([], 4, 2) == (0, 0)
    b = [shard_bounds([1] * 100, 4, r) for r in range(4)]
    assert [e - s for s, e in b] == [25, 25, 25, 25]


def test_two_rank_gloo_pipeline_matches_oracle(emu, oracle, tmp_path):
    n_loci = 5
    out = str(tmp_path / "gathered.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "multi_rank_worker.py"), out,
                           str(n_loci)], env=env, timeout=600)
    got = json.load(open(out))
    assert got["world"] == 2 and len(got["texts"]) == n_loci
    opts, sc, cuts = asm_opts(minWordLength=17, maxWordLength=32), [2, -8, -24, -1, -1, 0], (40, 40, 200, 200)
    for s, text in enumerate(got["texts"]):
        reads, ref = small_indel_locus(100 + s, n_reads=16 + 4 * (s % 3), read_len=50, ref_len=400)
        assert text == oracle.small_sv_locus(opts, sc, -100, reads, ref, cuts), s


@pytest.mark.gpu
def test_gpu_gather_staging_buffers():
    """the page-locked / device staging of gather_bytes on a real device (the collective itself needs >= 2 GPUs: the hops around it
    are exercised here with the gather replaced by a device copy)"""
    import numpy as np
    import torch
    from manta_amd.shard import _GatherBuffers
    b = _GatherBuffers()
    rng = np.random.default_rng(5)
    for need in (1000, 7_000_000, 5000):  # grow, then reuse
        b.ensure("cuda", 2, 0, need)
        assert b.cap >= need and b.host_pay.is_pinned() and b.host_recv.is_pinned()
        local = rng.integers(0, 256, size=need, dtype=np.uint8)
        b.host_pay[:need].copy_(torch.from_numpy(local))
        b.pay[:need].copy_(b.host_pay[:need], non_blocking=True)
        b.recv[1, :need].copy_(b.pay[:need])  # stands in for dist.gather
        b.host_recv.copy_(b.recv, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        assert np.array_equal(b.host_recv[1, :need].numpy(), local)


@pytest.mark.gpu
def test_gpu_gather_bytes_over_rccl_single_rank():
    """gather_bytes' collectives (all_gather_into_tensor + gather) and staging on the real backend: a one-rank RCCL group in a
    subprocess (more ranks need more GPUs; the multi-rank logic is the gloo test above)"""
    code = (
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from manta_amd.shard import gather_bytes\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "for n in (0, 1, 12345, 7000000, 999):\n"
        "    x = np.random.default_rng(n).integers(0, 256, size=n, dtype=np.uint8)\n"
        "    got = gather_bytes(x, device='cuda', force_collectives=True)\n"
        "    assert len(got) == 1 and np.array_equal(got[0], x), n\n"
        "dist.destroy_process_group()\n"
        "print('rccl gather ok')\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl gather ok" in out.stdout, out.stderr[-2000:]
