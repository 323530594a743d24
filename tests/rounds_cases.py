"""Random repeat-rich piles of 129..236 reads under random assembler options: the inputs of the big class' word-length rounds
(graph_big_kernel -> repeat_big_kernel -> contig_big_kernel per word length).  Shared by tools/sweeps/sweep_rounds.py (developer sweep,
thousands of seeds) and the GPU tier (tests/test_assemble_kernels.py: the first 500 seeds of the round-5 hardware sweep)."""
import numpy as np

import synth
from oracle_lib import asm_opts


def rounds_case(s):
    """seed -> (options, reads) or None if the pile falls outside the big class' read envelope"""
    rng = np.random.default_rng(9000 + s)
    nr = int(rng.integers(129, 237))
    if s % 3 == 0:
        reads = synth.repeat_rich_pile(s, n_reads=nr, read_len=int(rng.integers(40, 90)))
    elif s % 3 == 1:
        reads = synth.small_indel_locus(s, n_reads=nr, read_len=int(rng.integers(60, 120)), ref_len=500, sub_rate=0.01, n_rate=0.005, tandem=True)[0]
    else:
        reads = synth.breakend_locus(s, n_reads=min(nr, 200), read_len=int(rng.integers(80, 160)), ref_len=600, tandem_frac=1.0)[0]
    mac = int(rng.integers(2, 11))
    k0 = int(rng.integers(8, 33))
    o = asm_opts(minWordLength=k0, maxWordLength=k0 + int(rng.integers(0, 40)), wordStepSize=int(rng.integers(1, 8)), minCoverage=int(rng.integers(1, 4)),
                 minConservativeCoverage=int(rng.integers(1, 4)), maxAssemblyCount=mac, minContigLength=15,
                 minUnusedReads=int(rng.integers(1, 5)), minSupportReads=int(rng.integers(1, 4)))
    return (o, reads) if len(reads) + 2 * mac <= 256 else None
