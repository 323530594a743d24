"""Deterministic SmallAssembler test inputs (shared by the golden generator and the tests): the scenarios of the reference's
assembly/test/SmallAssemblerTest.cpp and seeded random read piles with alleles, tandem repeats, errors and 'N's."""
import random

UNIT_OPTS = [6, 6, 5, 2, 2, 3, 10]  # minWordLength, maxWordLength, wordStepSize, minCoverage, minConservativeCoverage, minSeedReads, maxAssemblyIterations
UNIT_CASES = {
    # SmallAssemblerTest.cpp:26-62 without its junk read "123456789123" (bytes outside {A,C,G,T,N}: DESIGN.md 6)
    "SmallAssembler1": ["ACGTGTATTACC", "GTGTATTACCTA", "ATTACCTAGTAC", "TACCTAGTACTC"],
    # :64-98
    "PoisonRead": ["ACGTGTATTACC", "GTGTATTACCTA", "ATTACCTAGTAC", "TACCTAGTACTC", "AAAAAAAAAAAAAAAAAAAA"],
    # :100-150
    "supportingReadConsistency": ["AAACGTGTATTA", "ACGTGTATTACC", "CGTGTATTACCT", "GTGTATTACCTA", "ATTACCTAGTAC", "TACCTAGTACTC",
                                  "CCCTTAGCTAAC", "CTTAGCTAACGT", "TAGCTAACGTGG", "GCTAACGTGGCC", "AACGTGGCCTAG"],
}

# (opts7, reads): empty and degenerate inputs, word length == read length, unmet thresholds, several iterations, wide read sets
EDGE_CASES = {
    "empty": ([6, 6, 1, 1, 1, 1, 10], []),
    "one_short_read": ([6, 6, 1, 1, 1, 1, 10], ["ACG"]),
    "all_N": ([6, 8, 1, 1, 1, 1, 10], ["NNNNNNNNNNNN", "NNNNNNNNNN"]),
    "no_words_minSeedReads_0": ([6, 6, 1, 1, 1, 0, 3], ["ACG", "TT"]),
    "homopolymer_only": ([4, 8, 2, 1, 1, 1, 5], ["AAAAAAAAAAAA", "AAAAAAAAAAAAAAA", "AAAAAAAA"]),
    "duplicate_reads": ([5, 5, 1, 2, 2, 2, 4], ["ACGTACGGTCA"] * 5),
    "word_is_read": ([8, 8, 1, 1, 1, 1, 4], ["ACGTTGCA", "ACGTTGCA", "CGTTGCAA"]),
    "minCoverage_unmet": ([6, 6, 1, 5, 2, 1, 4], ["ACGTGTATTACC", "GTGTATTACCTA"]),
    "two_iterations": ([6, 6, 1, 1, 1, 1, 10], ["ACGTGTATTACC", "GTGTATTACCTA", "TTTTGGGGCCCCAAAT", "TTGGGGCCCCAAATGC"]),
    "four_set_words": ([15, 25, 5, 2, 2, 3, 10], ["ACGTACGGTCAGGCTTAACGGATCCGATTACAGGCATTACGGA"[i % 10:i % 10 + 30] for i in range(200)]),
}


def random_case(seed):
    """(opts7, reads) of random case `seed`"""
    rng = random.Random(1000003 * seed + 17)
    L = rng.randint(80, 300)
    ref = "".join(rng.choice("ACGT") for _ in range(L))
    if rng.random() < 0.3:  # tandem repeat / homopolymer stretch: repeat reads, word-length escalation
        p = rng.randint(0, L - 20)
        unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 4)))
        ref = ref[:p] + unit * rng.randint(5, 15) + ref[p:]
    alt = None
    if rng.random() < 0.5:  # a second allele: branches, rejecting reads, a second iteration
        p = rng.randint(20, len(ref) - 20)
        alt = ref[:p] + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 12))) + ref[p + rng.randint(0, 8):]
    reads = []
    n = rng.randint(3, 90)
    rl = rng.randint(25, 70)
    for _ in range(n):
        src = alt if (alt and rng.random() < 0.4) else ref
        s = rng.randint(0, max(0, len(src) - rl))
        r = list(src[s:s + rl])
        for i in range(len(r)):
            x = rng.random()
            if x < 0.01:
                r[i] = rng.choice("ACGT")
            elif x < 0.013:
                r[i] = "N"
        reads.append("".join(r))
    kmin = rng.choice([6, 11, 15, 21, 25, 33])
    step = rng.choice([1, 3, 5, 7])
    kmax = kmin + step * rng.randint(0, 4)
    opts = [kmin, kmax, step, rng.choice([1, 1, 2, 3]), rng.choice([1, 2, 3]), rng.choice([1, 2, 3, 5]), rng.choice([1, 3, 10])]
    return opts, reads


def abi_opts(opts7):
    """7 reference-order values -> manta_small_asm_options_t order (min_contig_length = the reference's default 15)"""
    return [opts7[0], opts7[1], opts7[2], 15, opts7[3], opts7[4], opts7[5], opts7[6]]
