#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) -> profiles/<tag>_*: the bench line, the
rocprofv3 kernel-stats table, one PMC row per kernel, and profiles/traffic.json (HBM-side bytes per launch that bench.py
reports as roofline.traffic)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def short(name):
    n = name.replace("manta_dev::", "").replace("void ", "")
    return n.split("(")[0]


line = open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(dst, tag + "_bench_line.json"), "w").write(line + "\n")
sp = os.path.join(src, "bench_spanning_line.json")
if os.path.exists(sp) and os.path.getsize(sp):
    open(os.path.join(dst, tag + "_bench_spanning_line.json"), "w").write(open(sp).read().strip().splitlines()[-1] + "\n")
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    open(os.path.join(dst, tag + "_bench_kernel_stats.csv"), "w").write(open(stats[0]).read())
# the spanning workload's kernel stats (round 2 forgot to copy them: the committed file was a stale one)
sstats = glob.glob(os.path.join(src, "stats_spanning", "**", "*kernel_stats.csv"), recursive=True)
if sstats:
    open(os.path.join(dst, tag + "_bench_spanning_kernel_stats.csv"), "w").write(open(sstats[0]).read())
# one kernel trace of the default step (start / end of every dispatch: the host turnarounds between kernels are the gaps)
trace = glob.glob(os.path.join(src, "trace_step", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    rows = list(csv.DictReader(open(trace[0])))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    with open(os.path.join(dst, tag + "_step_kernel_trace.csv"), "w") as out:
        out.write("# rocprofv3 --kernel-trace of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras`: every dispatch, ms since the first\n")
        out.write("kernel,start_ms,end_ms,duration_ms,stream,lds_bytes,vgpr,scratch\n")
        for r in rows:
            a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
            out.write('"%s",%.3f,%.3f,%.3f,%s,%s,%s,%s\n' % (short(r["Kernel_Name"]), a, b, b - a, r.get("Stream_Id", ""), r.get("LDS_Block_Size", ""),
                                                            r.get("VGPR_Count", ""), r.get("Scratch_Size", "")))
per, launches = {}, {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        seen = {}
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if "rocclr" in k:
                continue
            per.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            seen.setdefault(k, set()).add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
        for k, ids in seen.items():
            launches[k] = max(launches.get(k, 0), len(ids))
cols = sorted({c for v in per.values() for c in v})
if not per:  # a lines-only run (PROFILE_LINES_ONLY=1): the counter summaries of the full run stay
    print(line[:300])
    sys.exit(0)
with open(os.path.join(dst, tag + "_pmc_summary.csv"), "w") as out:
    out.write("# rocprofv3 --pmc passes over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline` (10k loci, config 2), summed per kernel\n")
    out.write("# over the launches of that one step.  FETCH_SIZE / WRITE_SIZE are KiB as rocprofv3 reports them (separate passes; see\n")
    out.write("# MI355X_MICROARCH.md HBM section: FETCH_SIZE can under-count wide coalesced reads by 2x on gfx950; these reads are narrow and\n")
    out.write("# scattered, so the value is reported raw).\n")
    out.write("kernel,launches," + ",".join(cols) + "\n")
    for k in sorted(per):
        out.write('"%s",%d,' % (k, launches.get(k, 1)) + ",".join("%.0f" % per[k].get(c, 0) for c in cols) + "\n")
traffic = {"loci": 10000, "workload": "smallsv", "source": "tools/profile_round.sh " + tag + " (builder-run counter passes, not the driver's run)",
           "date": __import__("datetime").date.today().isoformat(),
           "note": "HBM-side bytes of ONE step (one block of 10 000 loci) = (FETCH_SIZE + WRITE_SIZE) KiB * 1024 summed over the step's launches of the "
                   "kernel(s), rocprofv3 --pmc, separate passes; graph_kernel's FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads), "
                   "everything else raw; assembler_stage = graph_kernel + contig_kernel launches (+ assemble_kernel's launch for punted loci)"}
agg = {}
for k, v in per.items():
    if k.startswith("align_kernel<1") or k.startswith("align_pair"):
        name = "align_kernel<LARGE_INDEL>"  # (the E-bucket launches of a block, packed pairs or not, together)
    elif k.startswith("graph_kernel") or k.startswith("contig_kernel") or k.startswith("assemble_kernel"):
        name = "assembler_stage"  # graph_kernel + contig_kernel (+ the general kernel's launch for punted loci): what bench.py times as the assembler
    else:
        name = k
    # graph_kernel reads the pile with 16-byte-per-lane coalesced loads: FETCH_SIZE reports half of such bytes on gfx950
    # (MI355X_MICROARCH.md, HBM section) -> doubled; every other counter raw
    fetch = v.get("FETCH_SIZE", 0) * (2 if k.startswith("graph_kernel") else 1)
    agg[name] = agg.get(name, 0) + (fetch + v.get("WRITE_SIZE", 0)) * 1024
    if k.startswith("graph_kernel") or k.startswith("contig_kernel"):
        traffic[k.split("<")[0]] = int(traffic.get(k.split("<")[0], 0) + (fetch + v.get("WRITE_SIZE", 0)) * 1024)
for k, v in agg.items():
    traffic[k] = int(v)
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
# the spanning workload's passes (tools/profile_round.sh <tag> spanning)
sper, slaunch = {}, {}
for d in ("spanning_pmc_fetch", "spanning_pmc_write"):
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        seen = {}
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if "rocclr" in k:
                continue
            sper.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
            sper[k][row["Counter_Name"]] += float(row["Counter_Value"])
            seen.setdefault(k, set()).add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
        for k, ids in seen.items():
            slaunch[k] = max(slaunch.get(k, 0), len(ids))
if sper:
    span_loci = 65536
    if os.path.exists(sp) and os.path.getsize(sp):
        span_loci = json.loads(open(sp).read().strip().splitlines()[-1])["config"]["loci_per_gpu"]
    marker = os.path.join(src, "spanning_pmc_loci.txt")  # (counter passes on fewer loci than the bench line: tools/gpu_r4_q.sh)
    if os.path.exists(marker):
        span_loci = int(open(marker).read().split()[0])
    st = {"loci": span_loci, "workload": "spanning", "source": traffic["source"] + " / tools/gpu_r4_q.sh", "date": traffic["date"],
          "note": "HBM-side bytes of ONE step of `bench.py --workload spanning --loci <loci>` = (FETCH_SIZE + WRITE_SIZE) KiB * 1024 per kernel, rocprofv3 --pmc, "
                  "separate passes, raw.  `loci` is the size of the counter passes' block: under the profiler the bench line's 65 536-locus block ran into "
                  "the passes' time limit, so they ran on fewer loci (per-locus traffic is what carries over)"}
    for k, v in sper.items():
        name = "align_kernel<JUMP>" if k.startswith("align_kernel<2") else k
        n = 1 if name.startswith("align_kernel") else max(1, slaunch.get(k, 1))
        st[name] = int(st.get(name, 0) + (v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / n)
    json.dump(st, open(os.path.join(dst, "traffic_spanning.json"), "w"), indent=1)
    print(json.dumps(st, indent=1))
# read gathering (tools/profile_round.sh <tag> ... read_class): line, kernel stats, one counter row per kernel (launches of the whole
# bench_read_class.py run: 6 timed calls + the checked one)
rcl = os.path.join(src, "read_class_line.json")
if os.path.exists(rcl) and os.path.getsize(rcl):
    json.loads(open(rcl).read())
    open(os.path.join(dst, tag + "_read_class_line.json"), "w").write(open(rcl).read())
    rcs = glob.glob(os.path.join(src, "rc_stats", "**", "*kernel_stats.csv"), recursive=True)
    if rcs:
        open(os.path.join(dst, tag + "_read_class_kernel_stats.csv"), "w").write(open(rcs[0]).read())
    rper, rlaunch = {}, {}
    for d in ("rc_pmc_fetch", "rc_pmc_write", "rc_pmc_sq"):
        for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
            seen = {}
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if "rocclr" in k:
                    continue
                rper.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
                rper[k][row["Counter_Name"]] += float(row["Counter_Value"])
                seen.setdefault(k, set()).add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
            for k, ids in seen.items():
                rlaunch[k] = max(rlaunch.get(k, 0), len(ids))
    if rper:
        rcols = sorted({c for v in rper.values() for c in v})
        with open(os.path.join(dst, tag + "_read_class_pmc.csv"), "w") as out:
            out.write("# rocprofv3 --pmc passes over `python tools/bench_read_class.py 150` (315 541 records, 150 candidates), summed per kernel over the\n")
            out.write("# run's launches (column 2); FETCH_SIZE / WRITE_SIZE in KiB, separate passes\n")
            out.write("kernel,launches," + ",".join(rcols) + "\n")
            for k in sorted(rper):
                out.write('"%s",%d,' % (k, rlaunch.get(k, 1)) + ",".join("%.0f" % rper[k].get(c, 0) for c in rcols) + "\n")
print(json.dumps(traffic, indent=1))
print(line[:300])
