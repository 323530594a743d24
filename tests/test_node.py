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


# ---- spanning candidates (breakend loci, GlobalJumpAligner) through the same queues --------------------------------------------
from manta_amd._capi import assembly_text, pack_piles  # noqa: E402
from synth import breakend_locus  # noqa: E402
from test_spanning_pipeline import SC as SPAN_SC, oracle_locus  # noqa: E402

SPAN_OPTS = asm_opts(minWordLength=25, maxWordLength=45, minContigLength=40)
SPAN_CUTS = (20, 20, 20, 20)


def spanning_loci(n, seed):
    """small breakend loci of two sizes, interleaved (every 4th holds three times the reads)"""
    return [breakend_locus(seed + s, n_reads=(36 if s % 4 == 0 else 12), read_len=70, ref_len=320, tandem_frac=0.0) for s in range(n)]


def span_text(r):
    got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"]) for a in r["aligns"]]
    return assembly_text(r), got


def span_want(oracle, locus):
    return oracle_locus(oracle, SPAN_OPTS, locus[0], locus[1], locus[2], SPAN_CUTS)


def check_span_results(oracle, loci, res, only=None):
    for l in (range(len(loci)) if only is None else only):
        text, aligns = span_want(oracle, loci[l])
        got_text, got = span_text(res[l])
        assert got_text == text and got == aligns, l


def check_node_spanning(path, devices, oracle, n=16, block=3):
    loci = spanning_loci(n, 300)
    batch = pack_spanning([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [SPAN_CUTS] * n)
    node = Node(path=path, devices=devices)
    out = BatchOutput(None, "spanning", n, 10, 1 << 20, 1 << 16, 1 << 18)
    per_dev = node.spanning_batch(SPAN_OPTS, SPAN_SC, -100, batch, out, block_loci=block)
    assert sum(per_dev) == n and len(per_dev) == len(devices)
    check_span_results(oracle, loci, out.decode(np.diff(batch[2])))
    node.close()
    return per_dev


def test_emulated_node_spanning_two_and_eight_contexts(emu, oracle):
    """manta_node_spanning_batch: every locus exactly once, results equal to the oracle's call-by-call composition"""
    per_dev = check_node_spanning(EMU, (0, 0), oracle)
    assert all(x > 0 for x in per_dev), per_dev
    per_dev = check_node_spanning(EMU, (0,) * 8, oracle, n=32, block=2)
    assert len(per_dev) == 8 and sum(per_dev) == 32 and all(x > 0 for x in per_dev), per_dev


def _span_rank(rank, shm_name, n, block, q, delay):
    """one process of the node on the spanning call; `delay` seconds late (a rank that arrives after the others took their blocks)"""
    import ctypes
    import time
    shm = shared_memory.SharedMemory(name=shm_name)
    counter = ctypes.c_uint32.from_buffer(shm.buf)
    lib = Lib(path=EMU)
    loci = spanning_loci(n, 300)
    batch = pack_spanning([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [SPAN_CUTS] * n)
    out = BatchOutput(lib, "spanning", n, 10, 1 << 20, 1 << 16, 1 << 18)
    if delay:
        time.sleep(delay)
    lib.spanning_batch(SPAN_OPTS, SPAN_SC, -100, batch, out, block_loci=block, shared_queue=ctypes.addressof(counter))
    res = out.decode(np.diff(batch[2]))
    q.put((rank, [(l, span_text(r)) for l, r in enumerate(res) if r["status"] != -10]))
    del counter
    shm.close()


def _run_span_ranks(world, n, block, delays):
    shm = shared_memory.SharedMemory(create=True, size=64)
    shm.buf[:64] = bytes(64)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_span_rank, args=(r, shm.name, n, block, q, delays[r])) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    shm.close()
    shm.unlink()
    taken, per_rank = {}, [0] * world
    for rank, items in got:
        per_rank[rank] = len(items)
        for l, text in items:
            assert l not in taken, "locus %d taken twice" % l
            taken[l] = text
    assert sorted(taken) == list(range(n))
    return taken, per_rank


def test_emulated_spanning_shared_queue_two_and_eight_processes(emu, oracle):
    """manta_spanning_batch with manta_batch_plan_t::shared_queue: one process per device, every locus taken exactly once, the merged
    result equal to the oracle's"""
    for world, n, block in ((2, 12, 2), (8, 24, 1)):
        taken, per_rank = _run_span_ranks(world, n, block, [0.0] * world)
        loci = spanning_loci(n, 300)
        for l in range(n):
            text, aligns = span_want(oracle, loci[l])
            assert taken[l] == (text, aligns), l
        assert sum(per_rank) == n


def test_emulated_spanning_shared_queue_late_rank(emu, oracle):
    """a rank that reaches the call seconds after the others: coverage still holds (every locus exactly once) and the late rank simply
    takes what is left -- possibly nothing; loci_per_rank reports it"""
    n, block = 12, 1
    taken, per_rank = _run_span_ranks(2, n, block, [0.0, 3.0])
    assert sum(per_rank) == n and per_rank[0] >= per_rank[1], per_rank
    loci = spanning_loci(n, 300)
    for l in (0, n - 1):
        text, aligns = span_want(oracle, loci[l])
        assert taken[l] == (text, aligns)


def test_emulated_spanning_batch_piles_equals_the_oracle(emu, oracle):
    """manta_spanning_batch_piles: the whole-batch spanning call on packed read piles (2-bit codes + N bitmap)"""
    n = 10
    loci = spanning_loci(n, 500)
    batch = pack_spanning([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [SPAN_CUTS] * n)
    piles = pack_piles(batch[0], batch[1], batch[2])
    out = BatchOutput(emu, "spanning", n, 10, 1 << 20, 1 << 16, 1 << 18)
    emu.spanning_batch_piles(SPAN_OPTS, SPAN_SC, -100, piles, batch[3], batch[4], batch[5], batch[6], batch[7], out, block_loci=4)
    assert out.stats_dict()["n_blocks"] == 3
    check_span_results(oracle, loci, out.decode(np.diff(batch[2])))


@pytest.mark.gpu
def test_gpu_node_spanning_two_contexts_and_batch_piles(gpu, oracle):
    per_dev = check_node_spanning(None, (0, 0), oracle, n=48, block=6)
    assert all(x > 0 for x in per_dev), per_dev
    n = 32
    loci = spanning_loci(n, 500)
    batch = pack_spanning([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [SPAN_CUTS] * n)
    piles = pack_piles(batch[0], batch[1], batch[2])
    out = BatchOutput(gpu, "spanning", n, 10, 1 << 20, 1 << 16, 1 << 18)
    gpu.spanning_batch_piles(SPAN_OPTS, SPAN_SC, -100, piles, batch[3], batch[4], batch[5], batch[6], batch[7], out, block_loci=8)
    check_span_results(oracle, loci, out.decode(np.diff(batch[2])))
