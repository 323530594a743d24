#!/usr/bin/env python3
"""bench.py -- candidate SV loci assembled+aligned per second on MI355X (BASELINE.json metric).

A "step" is one pass of the fused hot path (assemble -> 10-mer reference trim -> large-indel align + traceback,
i.e. the arithmetic core of SVCandidateAssemblyRefiner::getSmallSVAssembly) over one resident batch of synthetic
candidate loci.  Workload at N=1: BASELINE config[1] ("Synthetic 10k small-indel loci, 80 reads/locus x150bp, k=31,
1xMI355X", generator: tests/synth.py config2_batch).  With N>1 every rank owns an independent batch of the
same shape (loci are independent units: weak scaling, no data-path collective; only the timing reduction uses RCCL).

Inputs are resident in HBM when the timed region starts; the timed region contains the three kernels of K steps plus the
small device->host fetch of the bucket counts inside every step.  Result download happens once, outside, and is checked
against the oracle on a sample.

Prints ONE JSON line (see DESIGN.md "Measurement" for the definitions of roofline / cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

ASM_K = dict(minWordLength=31, maxWordLength=76, wordStepSize=5)
SCORES = [2, -8, -24, -1, -1, 0]  # SVRefinerOptions.hpp:40
LARGE_INDEL = -100               # SVRefinerOptions.hpp:44
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(batch, results):
    """B = B_in + B_ptr + B_out summed over the batch.
    B_in  = read bases + reference bases (1 B/base as the boundary delivers them)
    B_ptr = sum over aligned contigs of P*(Q+1)*(R+1), P = 2 B (the reference's own 5x3-bit pointer cell)
    B_out = sum over contigs (Q + 2*ceil(nReads/8) + 64) + nReads*8"""
    bases, read_off, begin, refs, ref_off, cuts = batch
    b_in = int(read_off[-1]) + int(ref_off[-1])
    b_ptr = 0
    b_out = 0
    for l, r in enumerate(results):
        n_reads = int(begin[l + 1] - begin[l])
        ref_len = int(ref_off[l + 1] - ref_off[l])
        b_out += n_reads * 8
        for c, a in zip(r["contigs"], r["aligns"]):
            q = len(c["seq"])
            win = ref_len - a["lead"] - a["trail"]
            b_ptr += 2 * (q + 1) * (win + 1)
            b_out += q + 2 * ((n_reads + 7) // 8) + 64
    return b_in, b_ptr, b_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--loci", type=int, default=10000, help="loci per GPU (config 2 = 10k)")
    ap.add_argument("--workload", choices=("smallsv", "spanning"), default="smallsv",
                    help="smallsv = BASELINE config[1] (the metric's configuration, default); spanning = config[4] shape "
                         "(breakend loci, 200 reads x 250 bp, mixed k), an extra measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="loci in the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from manta_amd._capi import Lib, SmallSvBatch, small_sv_text
    from oracle_lib import asm_opts
    from synth import config2_batch, unpack_locus

    if args.workload == "spanning":
        return spanning_main(args, torch, dist, rank, local_rank, world)

    lib = Lib(device=local_rank)
    opts = asm_opts(**ASM_K)
    batch = config2_batch(args.loci, seed=12345 + 1000003 * rank)
    pipe = SmallSvBatch(lib, opts, SCORES, LARGE_INDEL)
    pipe.upload_packed(*batch)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.run()
    barrier()
    t0 = time.perf_counter()
    asm_ms = sched_ms = align_ms = 0.0
    for _ in range(args.steps):
        pipe.run()  # synchronous: returns after the last kernel of the step finished
        st = pipe.stats()
        asm_ms += st["assemble_ms"]
        sched_ms += st["schedule_ms"]
        align_ms += st["align_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    results = pipe.download()
    st = pipe.stats()
    n_contigs = sum(len(r["contigs"]) for r in results)
    n_fail = sum(1 for r in results if r["status"] != 0)

    if rank == 0:
        # ---- parity spot check against the oracle (checker only; never part of the measured path) ----
        from oracle_lib import OracleLib, RefLib, have_ref
        orc = OracleLib()
        mism = 0
        for l in range(0, args.loci, max(1, args.loci // 32)):
            reads, ref, cuts = unpack_locus(batch, l)
            if small_sv_text(results[l]) != orc.small_sv_locus(opts, SCORES, LARGE_INDEL, reads, ref, cuts):
                mism += 1
        if mism or n_fail:
            raise SystemExit("PARITY FAILURE: %d sampled loci differ from the oracle, %d loci failed" % (mism, n_fail))

        steps = args.steps
        loci_total = args.loci * world * steps
        value = loci_total / elapsed
        b_in, b_ptr, b_out = algorithmic_bytes(batch, results)
        asm_avg, align_avg = asm_ms / steps, align_ms / steps
        # dominant kernel by measured HIP-event time; its algorithmic bytes per launch:
        #   align_kernel    : contig + window bases read, pointer matrix written once, CIGAR/result out
        #   assemble_kernel : read bases in, contigs + read-support sets out
        q_bytes = sum(len(c["seq"]) for r in results for c in r["contigs"])
        win_bytes = sum(int(batch[4][l + 1] - batch[4][l]) - a["lead"] - a["trail"] for l, r in enumerate(results) for a in r["aligns"])
        align_bytes = q_bytes + win_bytes + b_ptr
        asm_bytes = int(batch[1][-1]) + b_out
        if align_avg >= asm_avg:
            dom, dom_bytes, dom_ms = "align_kernel<LARGE_INDEL>", align_bytes, align_avg
        else:
            dom, dom_bytes, dom_ms = "assemble_kernel", asm_bytes, asm_avg
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("loci") == args.loci:
                    traffic = tj.get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "candidate SV loci assembled+aligned per second (whole node)",
            "value": round(value, 1),
            "unit": "loci/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config[1]: synthetic small-indel loci, 80 reads/locus x150bp, k=31..76 step 5, "
                                   "assemble + 10-mer trim + GlobalLargeIndelAligner(2,-8,-24,-1,-1;-100) on 1800 bp windows",
                       "loci_per_gpu": args.loci, "reads_per_locus": 80, "read_len": 150, "ref_window": 1800,
                       "contigs_per_locus": round(n_contigs / args.loci, 3), "parallelism": "loci sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(dom_ms, 3)},
            "kernels_ms_per_step": {"assemble_kernel": round(asm_avg, 3), "smallsv_schedule_kernel": round(sched_ms / steps, 3),
                                    "align_kernel": round(align_avg, 3)},
            "algorithmic_bytes_per_locus": {"in": b_in / args.loci, "ptr": b_ptr / args.loci, "out": b_out / args.loci,
                                            "whole_path_GBps": round((b_in + b_ptr + b_out) * world * steps / elapsed / 1e9, 2)},
            "dp_gcups": round(st["dp_cells"] * world * steps / elapsed / 1e9, 2),
        }
        # ---- CPU baseline: the reference's own sources (oracle/_ref) on this box's host cores ----
        if world == 1 and not args.no_cpu_baseline:
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            kind, cpu = ("reference", RefLib()) if have_ref() else ("port", orc)
            # bounded sample: ~32 ms/locus/thread (BASELINE.md probe) -> about 10-20 s of CPU work in total
            n_1 = 96
            sb1 = config2_batch(n_1, seed=12345)
            secs1 = cpu.bench_small_sv(opts, SCORES, LARGE_INDEL, sb1[0], sb1[1], sb1[2], sb1[3], sb1[4], (100, 100, 800, 800), 1)
            n_s = args.cpu_sample or min(args.loci, max(128, 8 * cores))
            sb = config2_batch(n_s, seed=12345)
            secs = cpu.bench_small_sv(opts, SCORES, LARGE_INDEL, sb[0], sb[1], sb[2], sb[3], sb[4], (100, 100, 800, 800), cores)
            out["cpu_baseline"] = {"value": round(n_s / secs, 2), "unit": "loci/s", "cores": cores, "kind": kind,
                                   "sample": "%d loci of the same workload on %d host threads (one aligner per thread, as "
                                             "GenerateSVCandidates.cpp:232-266), %.1f s wall; single thread: %d loci in %.1f s"
                                             % (n_s, cores, secs, n_1, secs1),
                                   "single_thread_value": round(n_1 / secs1, 2)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def spanning_main(args, torch, dist, rank, local_rank, world):
    """BASELINE config[4] shape (SURVEY.md 8d "C5"): breakend loci, 200 reads x 250 bp, 0.5 % substitutions, 1 % N, 10 % of
    the loci with a tandem repeat, minWordLength drawn from {25,30,..,75}; GlobalJumpAligner (2,-8,-12,-1,-1;-100) on
    ref[100..800) windows with the re-align rule.  One step = one pass of the fused spanning pipeline over all 11 word
    length groups (the ABI takes one option set per batch)."""
    import re
    from manta_amd._capi import Lib, SpanningBatch
    from oracle_lib import asm_opts, OracleLib
    from synth import breakend_locus
    sys.path.insert(0, os.path.join(ROOT, "tests"))

    os.environ.setdefault("MANTA_AMD_WS_BUDGET_GB", "12")  # 11 resident batches share the HBM
    lib = Lib(device=local_rank)
    ks = list(range(25, 80, 5))
    per_group = max(1, args.loci // len(ks))
    distinct = 24
    SPAN_SC = [2, -8, -12, -1, -1, 0]
    groups = []
    for gi, k in enumerate(ks):
        base = [breakend_locus(1000003 * rank + 1000 * gi + s) for s in range(distinct)]
        loci = [base[i % distinct] for i in range(per_group)]
        o = asm_opts(minWordLength=k, maxWordLength=max(76, k), minContigLength=75)
        b = SpanningBatch(lib, o, SPAN_SC, -100)
        b.upload([l[0] for l in loci], [l[1] for l in loci], [l[2] for l in loci], [(100, 100, 100, 100)] * per_group)
        groups.append((k, o, base, b))
    n_loci = per_group * len(ks)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        for g in groups:
            g[3].run()
    barrier()
    t0 = time.perf_counter()
    asm_ms = sched_ms = align_ms = 0.0
    for _ in range(args.steps):
        for g in groups:
            g[3].run()
            st = g[3].stats()
            asm_ms += st["assemble_ms"]
            sched_ms += st["schedule_ms"]
            align_ms += st["align_ms"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        from test_spanning_pipeline import oracle_locus
        orc = OracleLib()
        b_in = b_ptr = b_out = n_contigs = cells = 0
        mism = 0
        for k, o, base, b in groups:
            res = b.download()
            st = b.stats()
            cells += st["dp_cells"]
            for i, r in enumerate(res):
                reads, ref1, ref2 = base[i % distinct]
                b_in += sum(len(x) for x in reads) + len(ref1) + len(ref2)
                b_out += len(reads) * 8
                for c, a in zip(r["contigs"], r["aligns"]):
                    q = len(c["seq"])
                    span = (len(ref1) + len(ref2)) if a["is_uncut"] else (len(ref1) + len(ref2) - 400)
                    b_ptr += (q + 1) * (span + 2)  # 1 B cells, the reference's own jump pointer matrix (GlobalJumpAligner.hpp:81-115)
                    b_out += q + 2 * ((len(reads) + 7) // 8) + 64
                n_contigs += len(r["contigs"])
            for i in (0, distinct // 2):  # parity spot check against the oracle (checker only)
                reads, ref1, ref2 = base[i]
                _, want = oracle_locus(orc, o, reads, ref1, ref2, (100, 100, 100, 100))
                got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"])
                       for a in res[i]["aligns"]]
                mism += got != want
        if mism:
            raise SystemExit("PARITY FAILURE: %d sampled spanning loci differ from the oracle" % mism)
        steps = args.steps
        value = n_loci * world * steps / elapsed
        asm_avg, align_avg = asm_ms / steps, align_ms / steps
        asm_bytes = b_in + b_out
        dom, dom_bytes, dom_ms = ("assemble_kernel", asm_bytes, asm_avg) if asm_avg >= align_avg else ("align_kernel<JUMP>", b_ptr, align_avg)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        out = {"metric": "candidate SV loci assembled+aligned per second (whole node)", "value": round(value, 1), "unit": "loci/s",
               "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
               "config": {"workload": "BASELINE config[4] shape (NOT the metric's configuration): breakend loci, 200 reads x 250 bp, 1 % N, "
                                      "10 % tandem-repeat loci, minWordLength in {25..75} (11 batches per step), assemble + "
                                      "GlobalJumpAligner(2,-8,-12,-1,-1;-100) on 700+700 bp windows + re-align rule",
                          "loci_per_gpu": n_loci, "distinct_loci": distinct * len(ks), "contigs_per_locus": round(n_contigs / n_loci, 3),
                          "parallelism": "loci sharded, %d rank(s)" % world},
               "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None, "algorithmic_bytes_per_launch": dom_bytes,
                            "avg_launch_ms": round(dom_ms, 3), "note": "summed over the 11 launches of a step"},
               "kernels_ms_per_step": {"assemble_kernel": round(asm_avg, 3), "spanning_schedule_kernel": round(sched_ms / steps, 3),
                                       "align+realign": round(align_avg, 3)},
               "algorithmic_bytes_per_locus": {"in": b_in / n_loci, "ptr": b_ptr / n_loci, "out": b_out / n_loci},
               "dp_gcups": round(cells * world * steps / elapsed / 1e9, 2),
               "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
