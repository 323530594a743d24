"""tools/replay_piles.py on the demo's pile dump (tests/golden/demo_pile_dump.txt.gz, written by tests/golden/make_demo_pile_dump.py: the
read piles the reference's own getBreakendReads gathers from the bundled demo BAMs for a sweep of its covered regions, as they reach the
assembler + aligner): every candidate through the whole-batch calls on packed piles, compared with the UNMODIFIED reference
(oracle/_ref/libmanta_ref.so, live) candidate by candidate.  SURVEY section 8(d) C1: real-data replay."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
DUMP = os.path.join(ROOT, "tests", "golden", "demo_pile_dump.txt.gz")


def run(lib):
    import replay_piles
    from oracle_lib import RefLib, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libmanta_ref.so not built")
    recs = replay_piles.parse_dump(DUMP)
    assert len(recs) == 72 and sum(1 for r in recs if r["kind"] == "J") == 16
    texts, routing = replay_piles.replay(lib, recs)
    ref = RefLib()
    bad = [r["id"] for r in recs if replay_piles.checker_text(ref, r) != texts[r["id"]]]
    assert not bad, "candidates that differ from the reference: %s" % bad
    assert sum(routing.values()) == len(recs)
    assert sum(t.count("\ncontig ") + t.startswith("contig ") for t in texts.values()) > 40  # the piles do assemble
    return texts


def test_emulated_replay_of_the_demo_pile_dump_matches_the_reference(emu):
    run(emu)


@pytest.mark.gpu
def test_gpu_replay_of_the_demo_pile_dump_matches_the_reference(gpu):
    run(gpu)


def test_refiner_writes_a_dump_that_replays_to_the_reference(emu, tmp_path):
    """the refiner's own dump hook (SVCandidateAssemblyRefiner::setPileDump, manta_amd/host/pile_dump.hpp): synthetic complex and spanning
    candidates go through the product refiner with the dump switched on; the records it wrote -- reads, oriented reference windows, cuts,
    options -- replayed through the whole-batch calls equal the unmodified reference run on those same records"""
    import ctypes
    import random
    import replay_piles
    from oracle_lib import RefLib, have_ref
    from refiner_loci import complex_case, spanning_case
    from test_refiner import build_mine
    import os as _os
    if not have_ref():
        pytest.skip("oracle/_ref/libmanta_ref.so not built")
    mine = build_mine(_os.path.join(ROOT, "tests", "emu"), "manta_amd_emu", "emu")
    mine.lib.mine_set_pile_dump.restype = ctypes.c_uint64
    path = str(tmp_path / "dump.txt")
    rng = random.Random(5)
    cases = [complex_case(rng, "del"), complex_case(rng, "ins"), spanning_case(rng, "RL"), spanning_case(rng, "LL", ins_len=7), spanning_case(rng, "RR")]
    mine.lib.mine_set_pile_dump(path.encode())
    for c in cases:
        assert not mine.run(c).startswith("EXCEPTION")
    assert mine.lib.mine_set_pile_dump(None) == len(cases)
    recs = replay_piles.parse_dump(path)
    assert [r["kind"] for r in recs] == ["S", "S", "J", "J", "J"]
    texts, _ = replay_piles.replay(emu, recs)
    ref = RefLib()
    for r in recs:
        assert replay_piles.checker_text(ref, r) == texts[r["id"]], r["id"]
