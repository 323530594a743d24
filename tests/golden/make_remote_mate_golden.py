#!/usr/bin/env python3
"""tests/golden/remote_mate_cases.json.gz: the chimeric-pair candidates of tests/test_read_class.py::remote_mate_case as the GPU box
can replay them (it has neither /root/reference nor oracle/_ref/libmanta_ref_bam.so): per case the records of the candidate's
region queries, every remote region query the retrieval asks for (answered here by the reference's own BAM layer), and the pile +
RemoteReadCache of the UNMODIFIED reference (assembleComplexSVCandidate with isSearchRemoteInsertionReads).  Authoring container
only:  python tests/golden/make_remote_mate_golden.py"""
import gzip
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import read_class_util as u  # noqa: E402
import test_read_class as t  # noqa: E402
from manta_amd._capi import read_class_options  # noqa: E402


class RecordingRefBam(u.RefBam):
    def __init__(self):
        super().__init__()
        self.regions = {}

    def region_records(self, bam, fasta, tid, begin, end):
        text = self.region_text(bam, fasta, tid, begin, end)
        assert not text.startswith("EXCEPTION"), text[:300]
        self.regions["%s:%d:%d-%d" % (os.path.basename(bam), tid, begin, end)] = text.splitlines()
        return [u.parse_record(l) for l in text.splitlines()]


def main():
    gather = u.GatherLib(os.path.join(ROOT, "tests", "emu"), "manta_amd_emu", "emu")
    cases = []
    with tempfile.TemporaryDirectory(prefix="manta_rg_") as tmp:
        for seed in range(16):
            rb = RecordingRefBam()
            cand, ref, fetch = t.remote_mate_case(rb, tmp, seed)
            local = dict(rb.regions)
            keys = list(local)
            assert len(keys) == len(cand["scans"])
            remote = {}

            def recording_fetch(bam_index, tid, begin, end):
                rb.regions = {}
                recs = fetch(bam_index, tid, begin, end)
                (lines,) = rb.regions.values()
                remote["%d:%d:%d-%d" % (bam_index, tid, begin, end)] = lines
                return recs
            out, stats = gather.gather([cand], read_class_options(), recording_fetch)
            assert out[0]["pile"] == ref["reads"] and out[0]["cache"] == ref["remote"], seed
            scans = [dict({k: v for k, v in s.items() if k != "records"}, lines=local[keys[i]]) for i, s in enumerate(cand["scans"])]
            cases.append(dict(seed=seed, scans=scans, is_max_depth=cand["is_max_depth"], max_depth=cand["max_depth"], max_local=cand["max_local"],
                              remote_regions=remote, ref_pile=ref["reads"], ref_cache=ref["remote"], inserted=stats["inserted"]))
    blob = dict(source="tests/golden/make_remote_mate_golden.py (unmodified SVCandidateAssembler.cpp + htsapi + redist htslib on synthetic BAM files)",
                cases=cases)
    path = os.path.join(ROOT, "tests", "golden", "remote_mate_cases.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(blob, sort_keys=True).encode())
    print(path, os.path.getsize(path), "bytes;", sum(c["inserted"] for c in cases), "remote mates in", len(cases), "cases")


if __name__ == "__main__":
    main()
