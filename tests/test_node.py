"""One node, several GPUs, one work queue:
  * manta_node_*: one process, one context per device, one cost-ordered block queue that every device pulls from;
  * manta_batch_plan_t::shared_queue: the same queue across PROCESSES (one per GPU), the counter in shared memory.
CPU tier: the emulator stands in for the devices (two contexts / two processes on it); GPU tier: two contexts on the one GPU of
the test box.  Batches have skewed costs (a few loci many times the size of the rest) so that a static split would be uneven."""
import multiprocessing as mp
import os
from multiprocessing import shared_memory

import numpy as np
import pytest

from manta_amd._capi import BatchOutput, Lib, Node, pack_spanning, small_sv_text
from oracle_lib import asm_opts
from synth import config2_batch, unpack_locus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so")
SCORES = [2, -8, -24, -1, -1, 0]
OPTS = asm_opts(minWordLength=25, maxWordLength=45)


def skewed_batch(n, seed):
    """config-2 shaped loci of two very different sizes, interleaved: every 5th locus has 4x the reads of the others"""
    small = config2_batch(n, seed=seed, n_reads=12, read_len=70, ref_len=420)
    big = config2_batch(n, seed=seed + 1, n_reads=48, read_len=70, ref_len=420)
    pick = [(big if l % 5 == 0 else small) for l in range(n)]
    reads, refs, cuts = [], [], []
    for l, b in enumerate(pick):
        r, ref, c = unpack_locus(b, l)
        reads.append(r)
        refs.append(ref)
        cuts.append((40, 40, 200, 200))
    flat = [x for r in reads for x in r]
    read_off = np.zeros(len(flat) + 1, dtype=np.uint64)
    np.cumsum([len(x) for x in flat], out=read_off[1:])
    bases = np.frombuffer(b"".join(flat) + b"\0" * 64, dtype=np.uint8)
    begin = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum([len(r) for r in reads], out=begin[1:])
    ref_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in refs], out=ref_off[1:])
    refs_np = np.frombuffer(b"".join(refs) + b"\0" * 64, dtype=np.uint8)
    return bases, read_off, begin, refs_np, ref_off, np.ascontiguousarray(np.array(cuts, dtype=np.int32))


def check_node(path, devices, oracle, n=20, block=3):
    batch = skewed_batch(n, 77)
    node = Node(path=path, devices=devices)
    out = BatchOutput(None, "smallsv", n, 10, 1 << 20, 1 << 16, 1 << 18)
    per_dev = node.smallsv_batch(OPTS, SCORES, -100, batch, out, block_loci=block)
    assert sum(per_dev) == n and len(per_dev) == len(devices)
    assert out.stats_dict()["n_blocks"] == (n + block - 1) // block
    res = out.decode(np.diff(batch[2]))
    for l, r in enumerate(res):
        reads, ref, cuts = unpack_locus(batch, l)
        assert small_sv_text(r) == oracle.small_sv_locus(OPTS, SCORES, -100, reads, ref, cuts), l
    node.close()
    return per_dev


def test_emulated_node_two_contexts_one_queue(emu, oracle):
    per_dev = check_node(EMU, (0, 0), oracle)
    assert all(x > 0 for x in per_dev), per_dev  # both contexts pulled blocks from the one queue


def test_emulated_node_eight_contexts_one_queue(emu, oracle):
    """eight devices (emulated contexts) on one queue, a batch of skewed costs in blocks of two: every locus exactly once (check_node
    compares all of them) and the spread between the busiest and the idlest device stays below 2x in LOCI WORTH OF COST -- a
    device that draws the big loci takes fewer of them"""
    n, block = 64, 2
    per_dev = check_node(EMU, (0,) * 8, oracle, n=n, block=block)
    assert len(per_dev) == 8 and sum(per_dev) == n and all(x > 0 for x in per_dev), per_dev
    assert max(per_dev) <= 4 * (n // 8), per_dev  # (nobody ends up with half the batch)


def test_emulated_node_single_device_and_default_blocks(emu, oracle):
    batch = skewed_batch(9, 5)
    node = Node(path=EMU, devices=(0,))
    out = BatchOutput(None, "smallsv", 9, 10, 1 << 20, 1 << 16, 1 << 18)
    assert node.smallsv_batch(OPTS, SCORES, -100, batch, out) == [9]
    for l, r in enumerate(out.decode(np.diff(batch[2]))):
        reads, ref, cuts = unpack_locus(batch, l)
        assert small_sv_text(r) == oracle.small_sv_locus(OPTS, SCORES, -100, reads, ref, cuts), l


def _rank(rank, shm_name, n, block, q):
    """one process of the node: same call on the same batch, blocks through the shared counter"""
    import ctypes
    shm = shared_memory.SharedMemory(name=shm_name)
    counter = ctypes.c_uint32.from_buffer(shm.buf)
    lib = Lib(path=EMU)
    batch = skewed_batch(n, 77)
    out = BatchOutput(lib, "smallsv", n, 10, 1 << 20, 1 << 16, 1 << 18)
    lib.smallsv_batch(OPTS, SCORES, -100, batch, out, block_loci=block, shared_queue=ctypes.addressof(counter))
    res = out.decode(np.diff(batch[2]))
    q.put((rank, [(l, small_sv_text(r)) for l, r in enumerate(res) if r["status"] != -10]))
    del counter
    shm.close()


def test_emulated_two_processes_share_one_queue(emu, oracle):
    """one PROCESS per device (the bench's multi-rank mode): the block counter lives in POSIX shared memory; every locus is
    taken by exactly one process and the merged result equals the oracle"""
    n, block = 20, 3
    shm = shared_memory.SharedMemory(create=True, size=64)
    shm.buf[:64] = bytes(64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, shm.name, n, block, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    shm.close()
    shm.unlink()
    taken = {}
    for rank, items in got:
        for l, text in items:
            assert l not in taken, "locus %d taken twice" % l
            taken[l] = text
    assert sorted(taken) == list(range(n))
    batch = skewed_batch(n, 77)
    for l in range(n):
        reads, ref, cuts = unpack_locus(batch, l)
        assert taken[l] == oracle.small_sv_locus(OPTS, SCORES, -100, reads, ref, cuts), l


@pytest.mark.gpu
def test_gpu_node_two_contexts_one_queue(gpu, oracle):
    per_dev = check_node(None, (0, 0), oracle, n=200, block=16)
    assert all(x > 0 for x in per_dev), per_dev
