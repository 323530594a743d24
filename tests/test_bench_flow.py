"""bench.py's own control flow on CPU: main() is run with torch.cuda stubbed out and the library handle pointed at the wave
emulator build (test infrastructure) on a tiny batch.  This is NOT a measurement and not a CPU mode of the bench -- bench.py
itself refuses to run without a GPU; the test only makes sure the script's plumbing (argument handling, the batch call, the
checker, the JSON line as the last stdout line, the forced single-rank torch.distributed path over gloo) cannot rot unseen."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: 1
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    import manta_amd._capi as capi
    emu = os.path.join(%(root)r, "tests", "emu", "libmanta_amd_emu.so")
    capi.default_library_path = lambda: emu
    import bench
    sys.argv = ["bench.py"] + %(argv)r
    bench.main()
""")


def run_bench(argv, extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    code = HARNESS % dict(root=ROOT, argv=argv)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert lines, out.stderr[-2000:]
    return json.loads(lines[-1]), lines  # the JSON line must be the LAST line on stdout


def test_bench_main_flow_on_the_emulator(emu):
    d, lines = run_bench(["--loci", "6", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["unit"] == "loci/s" and d["higher_is_better"] is True
    assert d["config"]["loci_per_gpu"] == 6 and "0 mismatches" in d["config"]["parity"]
    assert {"roofline", "kernels_ms_per_step", "pcie"} <= set(d)
    assert d["roofline"]["bound"] == "hbm" and "kernel" in d["roofline"]  # (the emulator has no event times: which kernel dominates is moot here)


def test_bench_forced_single_rank_group_over_gloo(emu):
    """the N>1 path's torch.distributed calls (group, barriers, timing all-reduce, result gather) with one rank"""
    d, lines = run_bench(["--loci", "5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"],
                         dict(MANTA_BENCH_FORCE_DIST="1", MANTA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561"))
    assert d["n_gpus"] == 1 and "gather_MB_per_step" in d["pcie"]  # (a few KB here: rounds to 0.00 MB)


def run_bench_ranks(world, argv, port, extra_env=None):
    """`world` processes as torchrun would start them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), gloo, emulator"""
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MANTA_BENCH_BACKEND="gloo", **(extra_env or {}))
        code = HARNESS % dict(root=ROOT, argv=argv)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert lines and not any(l.lstrip().startswith("{") for so, _ in outs[1:] for l in so.splitlines()), "only rank 0 prints the line"
    return json.loads(lines[-1])


def test_bench_two_ranks_share_one_block_queue(emu):
    """N = 2 over gloo: the node's batch behind ONE queue (shared-memory counter), every locus taken once, truthful line"""
    d = run_bench_ranks(2, ["--gpus", "2", "--loci", "6", "--steps", "2", "--warmup", "1", "--block-loci", "2", "--no-cpu-baseline", "--no-extras"], 29571)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["queue"] == "node" and c["backend"] == "gloo" and c["dist_world"] == 2
    assert sum(c["loci_per_rank"]) == 12 and len(c["loci_per_rank"]) == 2
    assert "gloo" in c["timed_region"] and "RCCL" not in c["timed_region"] and "0 mismatches" in c["parity"]
    # unequal parts (every 4th part is heavier; with two ranks: none) and the per-rank form
    d = run_bench_ranks(2, ["--gpus", "2", "--loci", "4", "--steps", "1", "--warmup", "0", "--queue", "rank", "--no-cpu-baseline", "--no-extras"], 29573)
    assert d["config"]["queue"] == "rank" and d["config"]["loci_per_rank"] == [4]


def test_bench_process_queue_eight_devices(emu):
    """--queue process: ONE process drives all devices through manta_node_smallsv_batch (the shape INTEGRATION.md B prescribes),
    same JSON contract; eight emulated devices, every 4th part heavier"""
    d, lines = run_bench(["--gpus", "8", "--queue", "process", "--loci", "3", "--steps", "1", "--warmup", "1", "--block-loci", "2", "--mix",
                          "--no-cpu-baseline", "--no-extras"])
    c = d["config"]
    assert d["n_gpus"] == 8 and c["queue"] == "process" and len(c["loci_per_rank"]) == 8 and sum(c["loci_per_rank"]) == 24
    assert "manta_node_smallsv_batch" in c["parallelism"] and "gather" not in c["timed_region"] and "0 mismatches" in c["parity"]


def test_bench_eight_ranks_share_one_block_queue(emu):
    """N = 8 over gloo with unequal parts (--mix): the parts reach every rank through shared-memory files, one queue hands out the
    node's blocks, every locus is taken exactly once (bench.py fails otherwise) and no rank starves"""
    d = run_bench_ranks(8, ["--gpus", "8", "--loci", "4", "--steps", "1", "--warmup", "0", "--block-loci", "2", "--mix", "--no-cpu-baseline",
                            "--no-extras"], 29581)
    c = d["config"]
    assert d["n_gpus"] == 8 and c["queue"] == "node" and c["dist_world"] == 8 and c["mix"] is True
    assert sum(c["loci_per_rank"]) == 32 and len(c["loci_per_rank"]) == 8 and "0 mismatches" in c["parity"]


def test_bench_spanning_flow_on_the_emulator(emu):
    d, lines = run_bench(["--workload", "spanning", "--loci", "3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert d["config"]["loci_per_gpu"] == 3 and "0 mismatches" in d["config"]["parity"]


def test_bench_extra_legs_on_the_emulator(emu):
    """the kernel_only and packed_input legs of the default run"""
    d, lines = run_bench(["--loci", "5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], dict(MANTA_BENCH_MIXED_HI="150"))
    assert "kernel_only" in d and "packed_input" in d and d["packed_input"]["unit"] == "loci/s"
    mx = d["mixed_shape"]
    assert "error" not in mx and mx["loci"] == 8 and "0 mismatches" in mx["parity"]
    r = mx["routing"]
    assert r["lds_small_class"] + r["lds_big_class"] + r["outside_both_classes"] == 8


def test_bench_spanning_through_the_queues(emu):
    """--workload spanning under --queue node (two gloo ranks on the shared-memory counter) and --queue process (one process, eight
    emulated devices through manta_node_spanning_batch): every locus taken exactly once and equal to the reference digests"""
    d = run_bench_ranks(2, ["--workload", "spanning", "--gpus", "2", "--loci", "2", "--steps", "1", "--warmup", "0", "--block-loci", "1",
                            "--no-cpu-baseline", "--no-extras"], 29591)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["queue"] == "node" and sum(c["loci_per_rank"]) == 4 and "0 mismatches" in c["parity"]
    d, lines = run_bench(["--workload", "spanning", "--gpus", "8", "--queue", "process", "--loci", "1", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-extras"])
    c = d["config"]
    assert d["n_gpus"] == 8 and c["queue"] == "process" and sum(c["loci_per_rank"]) == 8 and "manta_node_spanning_batch" in c["parallelism"]
    assert "0 mismatches" in c["parity"]
