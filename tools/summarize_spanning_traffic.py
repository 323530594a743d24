#!/usr/bin/env python3
"""gpurun_out/<dir>/spanning_pmc_{fetch,write}/ (tools/gpu_r5_w.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
`bench.py --workload spanning --loci N --steps 1 --warmup 0`) -> profiles/traffic_spanning.json: HBM-side bytes per kernel, summed over
ALL launches of the step (the big class launches its three kernels once per word length)."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r05w")
loci = int(sys.argv[2]) if len(sys.argv) > 2 else 2048


def short(n):
    return n.replace("manta_dev::", "").replace("void ", "").split("(")[0]


tot = collections.defaultdict(float)
for which in ("fetch", "write"):
    for row in csv.DictReader(open(os.path.join(src, "spanning_pmc_%s" % which, "p_counter_collection.csv"))):
        k = short(row["Kernel_Name"])
        if "rocclr" not in k:
            tot[k] += float(row["Counter_Value"]) * 1024
st = {"loci": loci, "workload": "spanning", "source": "tools/gpu_r6_final.sh (builder-run counter passes on the final build, not the driver's run)",
      "date": __import__("datetime").date.today().isoformat(),
      "note": "HBM-side bytes of ONE step of `bench.py --workload spanning --loci <loci>` = (FETCH_SIZE + WRITE_SIZE) KiB * 1024 per kernel, rocprofv3 --pmc, "
              "separate passes, raw, summed over all launches of the step (graph_big / repeat_big / contig_big: one launch per word length).  Counter "
              "passes on 2 048 loci (the whole digest set, 186 tandem-repeat piles among them): under the profiler larger blocks run into the passes' time limit"}
jump = 0.0
for k, v in sorted(tot.items()):
    if k.startswith("align_jump_pair_kernel") or k.startswith("align_kernel<2"):
        jump += v
    else:
        st[k] = int(v)
st["align_kernel<JUMP>"] = int(jump)
st["assembler_stage"] = int(sum(v for k, v in tot.items() if k.split("<")[0] in ("graph_big_kernel", "repeat_big_kernel", "contig_big_kernel", "assemble_kernel", "graph_kernel", "contig_kernel")))
st["assembler_stage_per_locus"] = int(st["assembler_stage"] / loci)
json.dump(st, open(os.path.join(ROOT, "profiles", "traffic_spanning.json"), "w"), indent=1)
print(json.dumps(st, indent=1))
