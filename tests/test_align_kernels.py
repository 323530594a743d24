"""HIP aligner kernels vs the oracle: on the wave emulator (CPU tier, small) and on the GPU (-m gpu, large)."""
import random

import pytest

from manta_amd._capi import align_text
from test_oracle_vs_ref import SCORE_SETS, _rand_align_case


def _run(lib, oracle, seed, n_batches, per_batch, maxlen):
    rng = random.Random(seed)
    n = 0
    for _ in range(n_batches):
        kind = rng.choice([0, 1, 2])
        sc = list(rng.choice(SCORE_SETS))
        if kind == 2:
            sc[5] = 0
        extra = rng.choice([-100, -3, -20, -50])
        probs = [_rand_align_case(rng, kind, maxlen) for _ in range(per_batch)]
        res = lib.align_batch(kind, sc, extra, probs)
        for p, r in zip(probs, res):
            assert r["status"] == 0
            assert align_text(kind, r) == oracle.align(kind, sc, extra, *p), (kind, sc, extra, p)
            n += 1
    return n


def test_emulated_align_small(emu, oracle):
    _run(emu, oracle, 3, 40, 4, 100)


def test_emulated_align_multi_column(emu, oracle):
    _run(emu, oracle, 5, 8, 3, 420)  # queries up to ~400 bp -> several columns per lane


def _pair_cases(rng, n, maxlen):
    return [_rand_align_case(rng, 1, maxlen) for _ in range(n)]


def _run_pairs(lib, oracle, seed, per_batch, maxlens):
    """GlobalLargeIndelAligner batches large enough per E bucket for align_pair_kernel (two alignments per wave, packed 16-bit
    arithmetic, align_pair.hpp) under score sets inside and outside its margin: the results equal the oracle's either way"""
    rng = random.Random(seed)
    n = 0
    for sc, extra in (([2, -8, -24, -1, -1, 0], -100), ([2, -4, -5, -1, -1, 0], -20), ([2, -8, -100, 0, -1, 0], -100),
                      ([2, -8, -18, -1, -1, 1], -50), ([1, -4, -6, -1, -2, 0], -3)):
        for maxlen in maxlens:
            probs = _pair_cases(rng, per_batch, maxlen)
            res = lib.align_batch(1, sc, extra, probs)
            for p, r in zip(probs, res):
                assert r["status"] == 0
                assert align_text(1, r) == oracle.align(1, sc, extra, *p), (sc, extra, p)
                n += 1
    return n


def test_emulated_align_pairs(emu, oracle):
    assert _run_pairs(emu, oracle, 23, 20, (60, 180, 380)) == 5 * 3 * 20


@pytest.mark.gpu
def test_gpu_align_pairs(gpu, oracle):
    assert _run_pairs(gpu, oracle, 29, 160, (60, 130, 200, 270, 330, 380, 520)) == 5 * 7 * 160


def _run_jump_pairs(lib, oracle, seed, per_batch, maxlens, n_random_sets=3):
    """GlobalJumpAligner batches large enough per E bucket for align_jump_pair_kernel (two alignments per wave, packed 16-bit arithmetic in
    the x4 domain, align_jump_pair.hpp): Manta's spanning scores, score sets at and beyond the kernel's margin (jumpPairEligible() then sends
    the bucket to the unpacked kernel), and randomly drawn score sets -- the results equal the oracle's either way.  Pairs are formed from
    neighbours of a bucket: tasks of different reference lengths, hence different seam rows, share a wave."""
    rng = random.Random(seed)
    sets = [([2, -8, -12, -1, -1, 0], -100), ([2, -4, -5, -1, -1, 0], -20), ([1, -4, -6, -2, -1, 0], -3), ([2, -8, -100, 0, -1, 0], -100),
            ([4, -30, -60, -8, -16, 0], -400)]
    for _ in range(n_random_sets):
        sets.append(([rng.randint(1, 3), -rng.randint(1, 12), -rng.randint(0, 30), -rng.randint(0, 3), -rng.randint(0, 4), 0], -rng.randint(0, 150)))
    n = 0
    for sc, jump in sets:
        for maxlen in maxlens:
            probs = [_rand_align_case(rng, 2, maxlen) for _ in range(per_batch)]
            res = lib.align_batch(2, sc, jump, probs)
            for p, r in zip(probs, res):
                assert r["status"] == 0
                assert align_text(2, r) == oracle.align(2, sc, jump, *p), (sc, jump, p)
                n += 1
    return n


def test_emulated_align_jump_pairs(emu, oracle, capfd, monkeypatch):
    monkeypatch.setenv("MANTA_AMD_DEBUG", "1")
    assert _run_jump_pairs(emu, oracle, 31, 16, (60, 200, 420)) == 8 * 3 * 16
    assert "align_jump_pair_kernel" in capfd.readouterr().err  # (the packed kernel did run)


@pytest.mark.gpu
def test_gpu_align_jump_pairs(gpu, oracle):
    """the sweep of tools/sweeps/sweep_align.py brought into the tier for the jump aligner: 8 score sets (3 of them random) x 7 query
    ranges x 100 alignments"""
    assert _run_jump_pairs(gpu, oracle, 37, 100, (60, 130, 200, 270, 330, 400, 520)) == 8 * 7 * 100


def _long_ref_cases():
    """short queries against a reference window of more than 65 535 rows: the packed pair kernel keeps a traceback start's row in 16
    bits, so such a bucket must run on the unpacked kernel (api.cpp: alignUsesPairs) -- the best start lies beyond row 65 536 here"""
    import numpy as np
    from synth import rand_seq, mutate
    rng = np.random.default_rng(77)
    probs = []
    for i in range(4):
        ref = rand_seq(rng, 66000 + 300 * i)
        at = 65700 + 37 * i
        q = np.concatenate([ref[at:at + 25], ref[at + 45:at + 70]]) if i % 2 else ref[at:at + 48]
        probs.append((mutate(rng, q, 0.02).tobytes(), ref.tobytes()))
    return probs


def test_emulated_align_pairs_reference_beyond_16_bit_rows(emu, oracle):
    sc = [2, -8, -24, -1, -1, 0]
    probs = _long_ref_cases()
    for p, r in zip(probs, emu.align_batch(1, sc, -100, probs)):
        assert r["status"] == 0 and r["begin1"] > 65535
        assert align_text(1, r) == oracle.align(1, sc, -100, *p)


@pytest.mark.gpu
def test_gpu_align_pairs_reference_beyond_16_bit_rows(gpu, oracle):
    sc = [2, -8, -24, -1, -1, 0]
    probs = _long_ref_cases()
    for p, r in zip(probs, gpu.align_batch(1, sc, -100, probs)):
        assert r["status"] == 0 and r["begin1"] > 65535
        assert align_text(1, r) == oracle.align(1, sc, -100, *p)


def test_empty_inputs_report_status(emu):
    res = emu.align_batch(0, [2, -4, -5, -1, -4, 0], 0, [("", "ACGT"), ("ACGT", ""), ("AC", "ACGT")], strict=False)
    assert [r["status"] for r in res] == [-4, -4, 0]
    res = emu.align_batch(2, [2, -4, -5, -1, -1, 0], -3, [("AC", "ACGT", "")], strict=False)
    assert res[0]["status"] == -4


@pytest.mark.gpu
def test_gpu_align_random(gpu, oracle):
    assert _run(gpu, oracle, 17, 60, 64, 160) == 60 * 64


@pytest.mark.gpu
def test_gpu_align_long(gpu, oracle):
    _run(gpu, oracle, 19, 12, 16, 1200)  # exercises every columns-per-lane bucket up to E=24


@pytest.mark.gpu
def test_gpu_align_config2_shape(gpu, oracle):
    """contig ~270 bp vs ~1.5 kb window, production small-SV scores (SVRefinerOptions.hpp:40,44)"""
    import numpy as np
    from synth import rand_seq, mutate
    rng = np.random.default_rng(5)
    probs = []
    for i in range(64):
        ref = rand_seq(rng, 1500)
        bp = 750
        size = int(rng.integers(10, 61))
        alt = np.concatenate([ref[:bp], ref[bp + size:]]) if i % 2 else np.concatenate([ref[:bp], rand_seq(rng, size), ref[bp:]])
        s = bp - 130
        probs.append((mutate(rng, alt[s:s + 270], 0.003).tobytes(), ref.tobytes()))
    sc = [2, -8, -24, -1, -1, 0]
    res = gpu.align_batch(1, sc, -100, probs)
    for p, r in zip(probs, res):
        assert align_text(1, r) == oracle.align(1, sc, -100, *p)
