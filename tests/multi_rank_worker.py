"""worker of tests/test_multi_rank.py: one rank of a world_size-2 gloo run of the sharded small-SV pipeline on the wave
emulator (CPU).  Rank 0 writes the gathered canonical texts to the output file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch.distributed as dist  # noqa: E402

from manta_amd._capi import Lib, SmallSvBatch, small_sv_text  # noqa: E402
from manta_amd.shard import shard_bounds, gather_records  # noqa: E402
from oracle_lib import asm_opts  # noqa: E402
from synth import small_indel_locus  # noqa: E402


def main():
    out_path, n_loci = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    loci = [small_indel_locus(100 + s, n_reads=16 + 4 * (s % 3), read_len=50, ref_len=400) for s in range(n_loci)]
    costs = [sum(len(r) for r in reads) for reads, _ in loci]
    b, e = shard_bounds(costs, world, rank)
    lib = Lib(path=os.path.join(ROOT, "tests", "emu", "libmanta_amd_emu.so"))
    opts, sc, cuts = asm_opts(minWordLength=17, maxWordLength=32), [2, -8, -24, -1, -1, 0], (40, 40, 200, 200)
    texts = []
    if e > b:
        pipe = SmallSvBatch(lib, opts, sc, -100)
        pipe.upload([l[0] for l in loci[b:e]], [l[1] for l in loci[b:e]], [cuts] * (e - b))
        pipe.run()
        texts = [small_sv_text(r).encode() for r in pipe.download()]
    gathered = gather_records(texts)
    if rank == 0:
        json.dump({"world": world, "bounds": [b, e], "texts": [t.decode() for t in gathered]}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
