"""Seeded synthetic read piles / reference windows (SURVEY.md section 8d shapes).  Shared by tests and bench.py."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_seq(rng, n):
    return ACGT[rng.integers(0, 4, size=n)]


def mutate(rng, seq, sub_rate, n_rate=0.0):
    seq = seq.copy()
    if sub_rate > 0:
        m = rng.random(len(seq)) < sub_rate
        # substitute with a DIFFERENT base
        idx = np.nonzero(m)[0]
        for i in idx:
            c = seq[i]
            while True:
                d = ACGT[rng.integers(0, 4)]
                if d != c:
                    break
            seq[i] = d
    if n_rate > 0:
        seq[rng.random(len(seq)) < n_rate] = ord("N")
    return seq


def small_indel_locus(seed, n_reads=80, read_len=150, ref_len=1800, sub_rate=0.003, n_rate=0.0, tandem=False):
    """Config-2 shape: ref = random ACGT; alt = ref with a deletion U{10..60} or an insertion (50/50) at the middle;
    reads sampled from alt so that each overlaps the breakpoint by >= 10 bp."""
    rng = np.random.default_rng(seed)
    ref = rand_seq(rng, ref_len)
    if tandem:
        unit = rand_seq(rng, int(rng.integers(2, 12)))
        reps = int(rng.integers(6, 20))
        blk = np.tile(unit, reps)
        p = ref_len // 2 - len(blk) // 2
        ref[p:p + len(blk)] = blk
    bp = ref_len // 2
    size = int(rng.integers(10, 61))
    if rng.random() < 0.5:
        alt = np.concatenate([ref[:bp], ref[bp + size:]])
    else:
        alt = np.concatenate([ref[:bp], rand_seq(rng, size), ref[bp:]])
    reads = []
    lo = max(0, bp - read_len + 10)
    hi = min(len(alt) - read_len, bp - 10)
    for _ in range(n_reads):
        s = int(rng.integers(lo, hi + 1))
        reads.append(mutate(rng, alt[s:s + read_len], sub_rate, n_rate).tobytes())
    return reads, ref.tobytes()


def breakend_locus(seed, n_reads=200, read_len=250, ref_len=900, sub_rate=0.005, n_rate=0.01, tandem_frac=0.1):
    """Config-5 shape: fused haplotype ref1[0..450+d1) + ins(U{0..20}) + ref2[450+d2..900); reads span the junction."""
    rng = np.random.default_rng(seed)
    ref1 = rand_seq(rng, ref_len)
    ref2 = rand_seq(rng, ref_len)
    if rng.random() < tandem_frac:
        unit = rand_seq(rng, int(rng.integers(2, 10)))
        blk = np.tile(unit, int(rng.integers(8, 25)))
        p = ref_len // 2 - len(blk) - 5
        ref1[p:p + len(blk)] = blk
    d1, d2 = int(rng.integers(-20, 21)), int(rng.integers(-20, 21))
    ins = rand_seq(rng, int(rng.integers(0, 21)))
    hap = np.concatenate([ref1[:ref_len // 2 + d1], ins, ref2[ref_len // 2 + d2:]])
    j = ref_len // 2 + d1
    reads = []
    lo = max(0, j - read_len + 20)
    hi = min(len(hap) - read_len, j - 20)
    for _ in range(n_reads):
        s = int(rng.integers(lo, hi + 1))
        reads.append(mutate(rng, hap[s:s + read_len], sub_rate, n_rate).tobytes())
    return reads, ref1.tobytes(), ref2.tobytes()


def repeat_rich_pile(seed, n_reads=12, read_len=40, alphabet=b"ACGT"):
    """Small adversarial piles with many cycles in the k-mer graph (exercise repeat detection + k iteration)."""
    rng = np.random.default_rng(seed)
    al = np.frombuffer(alphabet, dtype=np.uint8)
    n_units = int(rng.integers(1, 4))
    parts = []
    for _ in range(n_units):
        unit = al[rng.integers(0, len(al), size=int(rng.integers(1, 9)))]
        parts.append(np.tile(unit, int(rng.integers(2, 10))))
        parts.append(al[rng.integers(0, len(al), size=int(rng.integers(3, 25)))])
    hap = np.concatenate(parts)
    if len(hap) < read_len + 5:
        hap = np.concatenate([hap, al[rng.integers(0, len(al), size=read_len + 5 - len(hap))]])
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(max(8, read_len // 2), read_len + 1))
        s = int(rng.integers(0, len(hap) - L + 1))
        reads.append(mutate(rng, hap[s:s + L], 0.02, 0.01).tobytes())
    return reads


def config2_batch(n_loci, seed=12345, n_reads=80, read_len=150, ref_len=1800, sub_rate=0.003):
    """Vectorised config-2 generator: returns packed arrays ready for manta_smallsv_upload:
    bases(uint8), read_off(uint64), locus_read_begin(uint32), refs(uint8), ref_off(uint64), cuts(int32 n x 4).
    Per locus: ref = random ACGT; alt = ref with a deletion of U{10..60} (or an insertion of random bases, 50/50) at the
    middle; reads sampled from alt so that each overlaps the breakpoint by >= 10 bp; substitutions at `sub_rate`."""
    rng = np.random.default_rng(seed)
    code = rng.integers(0, 4, size=(n_loci, ref_len), dtype=np.uint8)
    bp = ref_len // 2
    size = rng.integers(10, 61, size=n_loci)
    is_del = rng.random(n_loci) < 0.5
    ins = rng.integers(0, 4, size=(n_loci, 60), dtype=np.uint8)
    reads = np.empty((n_loci, n_reads, read_len), dtype=np.uint8)
    for l in range(n_loci):
        s = int(size[l])
        alt = np.concatenate([code[l, :bp], code[l, bp + s:]]) if is_del[l] else np.concatenate([code[l, :bp], ins[l, :s], code[l, bp:]])
        lo, hi = max(0, bp - read_len + 10), min(len(alt) - read_len, bp - 10)
        starts = rng.integers(lo, hi + 1, size=n_reads)
        idx = starts[:, None] + np.arange(read_len)[None, :]
        reads[l] = alt[idx]
    m = rng.random(reads.shape) < sub_rate
    reads = np.where(m, (reads + rng.integers(1, 4, size=reads.shape, dtype=np.uint8)) & 3, reads).astype(np.uint8)
    bases = ACGT[reads].reshape(-1)
    refs = ACGT[code].reshape(-1)
    read_off = (np.arange(n_loci * n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    begin = (np.arange(n_loci + 1, dtype=np.uint32) * np.uint32(n_reads))
    ref_off = (np.arange(n_loci + 1, dtype=np.uint64) * np.uint64(ref_len))
    cuts = np.tile(np.array([100, 100, 800, 800], dtype=np.int32), (n_loci, 1))
    return (np.ascontiguousarray(bases), read_off, begin, np.ascontiguousarray(refs), ref_off, np.ascontiguousarray(cuts))


def unpack_locus(batch, l):
    """(reads list, ref bytes, cuts tuple) of locus l of a packed batch"""
    bases, read_off, begin, refs, ref_off, cuts = batch
    reads = [bases[int(read_off[r]):int(read_off[r + 1])].tobytes() for r in range(int(begin[l]), int(begin[l + 1]))]
    return reads, refs[int(ref_off[l]):int(ref_off[l + 1])].tobytes(), tuple(int(x) for x in cuts[l])


C5_WORD_LENGTHS = tuple(range(25, 80, 5))  # minWordLength drawn per locus from {25,30,...,75}


def config5_locus(i, seed0=555000):
    """Config-5 shape locus i: (reads, ref1, ref2, minWordLength, maxWordLength); k from a per-locus hash of the seed"""
    reads, ref1, ref2 = breakend_locus(seed0 + i)
    k = C5_WORD_LENGTHS[int(np.random.default_rng(seed0 + i + 7919).integers(0, len(C5_WORD_LENGTHS)))]
    return reads, ref1, ref2, k, max(76, k)


def mixed_shape_batch(n_loci, seed=777, read_len=150, ref_len=1800, sub_rate=0.003, lo=3, hi=1000):
    """Loci of very different sizes in one batch: the read count of a locus is drawn log-uniformly from lo..hi (small-indel loci
    otherwise as config2_batch).  Same array layout as config2_batch."""
    rng = np.random.default_rng(seed)
    bp = ref_len // 2
    n_reads = np.exp(rng.uniform(np.log(lo), np.log(hi + 1), size=n_loci)).astype(np.int64).clip(lo, hi)
    code = rng.integers(0, 4, size=(n_loci, ref_len), dtype=np.uint8)
    size = rng.integers(10, 61, size=n_loci)
    is_del = rng.random(n_loci) < 0.5
    parts = []
    for l in range(n_loci):
        s = int(size[l])
        alt = np.concatenate([code[l, :bp], code[l, bp + s:]]) if is_del[l] else np.concatenate([code[l, :bp], rng.integers(0, 4, size=s, dtype=np.uint8), code[l, bp:]])
        lo_s, hi_s = max(0, bp - read_len + 10), min(len(alt) - read_len, bp - 10)
        starts = rng.integers(lo_s, hi_s + 1, size=int(n_reads[l]))
        r = alt[starts[:, None] + np.arange(read_len)[None, :]]
        m = rng.random(r.shape) < sub_rate
        parts.append(np.where(m, (r + rng.integers(1, 4, size=r.shape, dtype=np.uint8)) & 3, r).astype(np.uint8).reshape(-1))
    bases = ACGT[np.concatenate(parts)]
    refs = ACGT[code].reshape(-1)
    begin = np.zeros(n_loci + 1, dtype=np.uint32)
    np.cumsum(n_reads, out=begin[1:])
    read_off = (np.arange(int(begin[-1]) + 1, dtype=np.uint64) * np.uint64(read_len))
    ref_off = (np.arange(n_loci + 1, dtype=np.uint64) * np.uint64(ref_len))
    cuts = np.tile(np.array([100, 100, 800, 800], dtype=np.int32), (n_loci, 1))
    return (np.ascontiguousarray(bases), read_off, begin, np.ascontiguousarray(refs), ref_off, np.ascontiguousarray(cuts))
