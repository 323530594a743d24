"""Synthetic refiner calls (candidate + chromosomes + read pile) for tests/test_refiner.py, and the ctypes view of the POD
input shared by oracle/ref_refiner_driver.cpp (reference refiner, in memory) and tests/cpp/host_refiner_full_capi.cpp."""
import ctypes
import random

UNKNOWN, RIGHT_OPEN, LEFT_OPEN, COMPLEX = 0, 1, 2, 3
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(_COMP[c] for c in reversed(s))


class RefineInput(ctypes.Structure):
    _fields_ = [("n_chrom", ctypes.c_int32), ("chrom_seq", ctypes.POINTER(ctypes.c_char_p)), ("bp_state", ctypes.c_int32 * 2),
                ("bp_tid", ctypes.c_int32 * 2), ("bp_begin", ctypes.c_int32 * 2), ("bp_end", ctypes.c_int32 * 2),
                ("is_find_large_insertions", ctypes.c_int32), ("n_reads", ctypes.c_int32), ("reads", ctypes.POINTER(ctypes.c_char_p)),
                ("small_word", ctypes.c_int32 * 3), ("spanning_word", ctypes.c_int32 * 3), ("n_calls", ctypes.c_int32)]


def fill(inp, case, keep):
    chroms, reads = case["chroms"], case["reads"]
    cs = (ctypes.c_char_p * len(chroms))(*[c.encode() for c in chroms])
    rs = (ctypes.c_char_p * max(1, len(reads)))(*([r.encode() for r in reads] or [b""]))
    keep.extend([cs, rs])
    inp.n_chrom, inp.chrom_seq = len(chroms), cs
    inp.bp_state[:] = case["state"]
    inp.bp_tid[:] = case["tid"]
    inp.bp_begin[:] = case["begin"]
    inp.bp_end[:] = case["end"]
    inp.is_find_large_insertions = case.get("large", 0)
    inp.n_reads, inp.reads = len(reads), rs
    inp.small_word[:] = case.get("small_word", [0, 0, 0])
    inp.spanning_word[:] = case.get("spanning_word", [0, 0, 0])
    inp.n_calls = case.get("calls", 1)


class RefinerLib:
    def __init__(self, path, prefix):
        self.lib = ctypes.CDLL(path)
        self.prefix = prefix
        self.single = getattr(self.lib, prefix + "_get_candidate_assembly_data")
        self.single.restype = ctypes.c_int
        self.multi = getattr(self.lib, prefix + "_get_candidate_assembly_data_multi", None)

    def last_stats(self):
        out = (ctypes.c_uint64 * 5)()
        self.lib.mine_last_stats(out)
        return dict(zip(("small", "spanning", "aligned", "realigned", "large_insertion"), list(out)))

    def run(self, case):
        keep, inp = [], RefineInput()
        fill(inp, case, keep)
        buf = ctypes.create_string_buffer(1 << 22)
        self.single(ctypes.byref(inp), buf, len(buf))
        return buf.value.decode()

    def vcf(self, case):
        """candidateSV.vcf records of the refined candidates of one call"""
        keep, inp = [], RefineInput()
        fill(inp, case, keep)
        buf = ctypes.create_string_buffer(1 << 22)
        getattr(self.lib, self.prefix + "_candidate_vcf_records")(ctypes.byref(inp), buf, len(buf))
        return buf.value.decode()

    def run_multi(self, cases, batched, plan_threads=None):
        """plan_threads (product adapter only): ONE batched call with per-candidate error isolation and that many host threads calling the
        input source (SVCandidateAssemblyRefiner::setPlanThreads); 1 = the sequential plan"""
        keep = []
        arr = (RefineInput * len(cases))()
        for i, c in enumerate(cases):
            fill(arr[i], c, keep)
        buf = ctypes.create_string_buffer(1 << 24)
        mode = (1 if batched else 0) if plan_threads is None else (-1 if plan_threads <= 1 else plan_threads)
        self.multi(arr, len(cases), mode, buf, len(buf))
        return buf.value.decode()


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def sample_reads(rng, hap, junction, n, read_len=150, err=0.003, min_overlap=12, n_rate=0.0):
    reads = []
    for _ in range(n):
        lo = max(0, junction - read_len + min_overlap)
        hi = max(lo, min(len(hap) - read_len, junction - min_overlap))
        s = rng.randint(lo, hi)
        r = list(hap[s:s + read_len])
        for i in range(len(r)):
            u = rng.random()
            if u < err:
                r[i] = rng.choice([b for b in "ACGT" if b != r[i]])
            elif u < err + n_rate:
                r[i] = "N"
        reads.append("".join(r))
    return reads


def complex_case(rng, kind="del", chrom_len=6000, pos=None, n_reads=40, two_haps=False, large=0, near_edge=False):
    chrom = rand_seq(rng, chrom_len)
    if pos is None:
        pos = rng.randint(300, 500) if near_edge else rng.randint(1500, chrom_len - 1500)
    if kind == "del":
        d = rng.randint(10, 80)
        hap = chrom[:pos] + chrom[pos + d:]
    elif kind == "ins":
        hap = chrom[:pos] + rand_seq(rng, rng.randint(10, 60)) + chrom[pos:]
    elif kind == "delins":
        hap = chrom[:pos] + rand_seq(rng, rng.randint(5, 30)) + chrom[pos + rng.randint(10, 60):]
    else:
        hap = chrom
    reads = sample_reads(rng, hap, pos, n_reads)
    if two_haps:
        p2 = pos + rng.randint(-40, 40)
        hap2 = chrom[:p2] + chrom[p2 + rng.randint(12, 50):]
        reads += sample_reads(rng, hap2, p2, n_reads // 2)
        rng.shuffle(reads)
    w = rng.randint(5, 40)
    return dict(chroms=[chrom], reads=reads, state=[COMPLEX, UNKNOWN], tid=[0, 0], begin=[pos - w, pos - w], end=[pos + w, pos + w],
                large=large)


def large_insertion_case(rng, chrom_len=6000, n_reads=30, ins_len=600, reach=110):
    """a long insertion whose middle is not covered: reads reach `reach` bases into the novel sequence from each side"""
    chrom = rand_seq(rng, chrom_len)
    pos = rng.randint(2000, chrom_len - 2000)
    ins = rand_seq(rng, ins_len)
    hap = chrom[:pos] + ins + chrom[pos:]
    reads = []
    for _ in range(n_reads):
        s = rng.randint(pos - 150 + 30, pos - 150 + reach)  # left edge reads
        reads.append(hap[s:s + 150])
        e = rng.randint(pos + ins_len + 150 - reach, pos + ins_len + 150 - 30)  # right edge reads
        reads.append(hap[e - 150:e])
    w = 20
    return dict(chroms=[chrom], reads=reads, state=[COMPLEX, UNKNOWN], tid=[0, 0], begin=[pos - w, pos - w], end=[pos + w, pos + w],
                large=1)


def spanning_case(rng, orient="RL", same_chrom=False, ins_len=0, n_reads=40, homology=0, near_edge=False, far=True, n_rate=0.0,
                  chrom_len=5000):
    """a breakend pair.  orient: states of (bp1, bp2), R = RIGHT_OPEN, L = LEFT_OPEN."""
    c0 = rand_seq(rng, chrom_len)
    c1 = c0 if same_chrom else rand_seq(rng, chrom_len)
    if same_chrom and far:
        chrom_len += 3000
        c0 = c1 = rand_seq(rng, chrom_len)
    p1 = rng.randint(120, 200) if near_edge else rng.randint(1000, (chrom_len - 4000) if (same_chrom and far) else (chrom_len - 1000))
    if same_chrom:
        p2 = p1 + (rng.randint(1500, 2500) if far else rng.randint(60, 300))
    else:
        p2 = rng.randint(1000, chrom_len - 1000)
    if homology:
        h = c0[p1 - homology:p1]
        c1 = c1[:p2 - homology] + h + c1[p2:]
        if same_chrom:
            c0 = c1
    ins = rand_seq(rng, ins_len)
    # sequence on the far side of each breakend, oriented away from the junction
    left = c0[:p1] if orient[0] == "R" else revcomp(c0[p1:])       # contig part that aligns to bp1's region
    right = c1[p2:] if orient[1] == "L" else revcomp(c1[:p2])      # contig part that aligns to bp2's region
    if orient == "LR":  # bp2 is aligned first: contig = bp2-side + bp1-side
        hap = c1[:p2] + ins + c0[p1:]
        junction = p2
    else:
        hap = left + ins + right
        junction = len(left)
    reads = sample_reads(rng, hap, junction + len(ins) // 2, n_reads, n_rate=n_rate)
    st = {"R": RIGHT_OPEN, "L": LEFT_OPEN}
    w1, w2 = rng.randint(5, 50), rng.randint(5, 50)
    return dict(chroms=[c0] if same_chrom else [c0, c1], reads=reads, state=[st[orient[0]], st[orient[1]]], tid=[0, 0 if same_chrom else 1],
                begin=[p1 - w1, p2 - w2], end=[p1 + w1, p2 + w2])
