#!/usr/bin/env python3
"""bench.py -- candidate SV loci assembled+aligned per second on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic candidate loci, measured the way SURVEY.md 8(d) defines
the metric: batch submit -> all results host-visible, H2D of the read piles / reference windows and D2H of every result
INSIDE the clock (the PCIe-inclusive rate is the headline; the device-resident kernel rate is an extra key).  One step =
one manta_smallsv_batch call (include/manta_amd.h): the batch is cut into blocks, the blocks go into a cost-ordered work
queue, host workers with private pipelines (own HIP stream, pinned staging) pull them: upload -> assemble -> 10-mer trim
-> large-indel align + traceback -> download.  Inputs sit in page-locked host memory, as a feeder that reads BAM records
into a staging buffer would leave them.

Workload at N=1: BASELINE config[1] ("Synthetic 10k small-indel loci, 80 reads/locus x150bp, k=31, 1xMI355X", generator:
tests/synth.py config2_batch).  N>1 (one process per GPU): the node's batch is N such batches -- weak scaling,
loci are independent units, no data-path collective -- behind ONE work queue: every rank makes the same manta_smallsv_batch
call on the whole node batch and blocks are handed out through a counter in shared memory (manta_batch_plan_t::shared_queue),
so a rank that draws expensive blocks takes fewer; the final candidate gather (every rank's result blob to rank 0 over
torch.distributed: RCCL on GPUs) is inside the clock.  `--queue rank` gives every rank its own batch and queue instead (the
round-2 form).  `python bench.py --gpus N` launches the N ranks itself when it was not started by torchrun.

`--workload spanning` = BASELINE config[4] shape (breakend loci, 200 reads x 250 bp, mixed k): an extra measurement.

Prints ONE JSON line (DESIGN.md "Measurement" defines roofline / cpu_baseline).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

ASM_K = dict(minWordLength=31, maxWordLength=76, wordStepSize=5)
SCORES = [2, -8, -24, -1, -1, 0]      # SVRefinerOptions.hpp:40
LARGE_INDEL = -100                    # SVRefinerOptions.hpp:44
SPAN_SC = [2, -8, -12, -1, -1, 0]     # SVRefinerOptions.hpp:43
JUMP = -100                           # SVRefinerOptions.hpp:45
HBM_PEAK_GBPS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
METRIC = "candidate SV loci assembled+aligned per second (whole node)"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loci", type=int, default=0, help="loci per GPU (default: 10000 smallsv = config 2; 65536 spanning)")
    ap.add_argument("--workload", choices=("smallsv", "spanning"), default="smallsv",
                    help="smallsv = BASELINE config[1] (the metric's configuration, default); spanning = config[4] shape "
                         "(breakend loci, 200 reads x 250 bp, mixed k), an extra measurement")
    ap.add_argument("--block-loci", type=int, default=0, help="loci per block of the work queue (0 = auto)")
    ap.add_argument("--workers", type=int, default=0, help="host workers / pipelines per GPU (0 = auto)")
    ap.add_argument("--serial-kernels", action="store_true", help="one block's kernels at a time (copies of the others still overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (profiling passes): no kernel_only / packed_input legs")
    ap.add_argument("--cpu-sample", type=int, default=0, help="loci in the CPU baseline sample (0 = auto)")
    ap.add_argument("--pageable", action="store_true", help="keep inputs/outputs in pageable host memory (A/B knob)")
    ap.add_argument("--queue", choices=("node", "rank", "process"), default="node",
                    help="N > 1: node = one block queue across the ranks (shared-memory counter; default), rank = every rank its own batch and "
                         "queue, process = ONE process drives all N devices through manta_node_smallsv_batch (one queue, one host; under "
                         "torchrun rank 0 does that and the other ranks only keep the barriers)")
    ap.add_argument("--mix", action="store_true",
                    help="N > 1, --queue node: every 4th rank part holds loci with 2.5x the reads (NOT the metric's configuration) -- makes the "
                         "queue's balancing visible in loci_per_rank")
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks (one per GPU, RCCL) and relay rank 0's JSON line"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def algorithmic_bytes_smallsv(batch, results, only=None):
    """B = B_in + B_ptr + B_out summed over the batch.
    B_in  = read bases + reference bases (1 B/base as the boundary delivers them)
    B_ptr = sum over aligned contigs of P*(Q+1)*(R+1), P = 2 B (the reference's own 5x3-bit pointer cell)
    B_out = sum over contigs (Q + 2*ceil(nReads/8) + 64) + nReads*8"""
    bases, read_off, begin, refs, ref_off, cuts = batch
    only = list(range(len(results))) if only is None else only
    b_in = b_ptr = b_out = q_bytes = win_bytes = 0
    for l in only:
        r = results[l]
        b_in += int(read_off[begin[l + 1]] - read_off[begin[l]]) + int(ref_off[l + 1] - ref_off[l])
        n_reads = int(begin[l + 1] - begin[l])
        ref_len = int(ref_off[l + 1] - ref_off[l])
        b_out += n_reads * 8
        for c, a in zip(r["contigs"], r["aligns"]):
            q = len(c["seq"])
            win = ref_len - a["lead"] - a["trail"]
            b_ptr += 2 * (q + 1) * (win + 1)
            b_out += q + 2 * ((n_reads + 7) // 8) + 64
            q_bytes += q
            win_bytes += win
    return b_in, b_ptr, b_out, q_bytes, win_bytes


def cores_available():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def host_cpu_topology():
    """what the CPU baseline can use: hardware threads in the affinity mask, physical cores among them (sysfs thread siblings), and the
    cgroup's CPU quota (a container may see 256 threads and be throttled to a few CPUs' worth of time)"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        cores.add(sib)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    return {"threads": len(cpus), "physical_cores": len(cores), "cgroup_cpu_quota": quota}


def result_blob(out):
    """what the final candidate gather moves: the rank's result records and the used part of its arenas"""
    res = np.frombuffer(out.res, dtype=np.uint32).reshape(-1, 12)[:out.n_loci]  # manta_asm_locus_result_t = 12 dwords: status, n_contigs, ...
    n_contigs = int(res[:, 1][res[:, 0] == 0].sum())
    parts = [np.frombuffer(out.res, dtype=np.uint8),
             np.frombuffer(out.contigs, dtype=np.uint8)[:n_contigs * 40],
             np.frombuffer(out.aligns, dtype=np.uint8)[:n_contigs * (64 if out.kind == "smallsv" else 64)],
             out.seq[:out.used[0].value], out.bits[:out.used[1].value].view(np.uint8), out.cig[:out.used[2].value].view(np.uint8)]
    return np.concatenate(parts)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and args.queue != "process":
        self_spawn(args)

    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)  # no torch CPU compute here; an OpenMP team left spinning would only fight the library's host threads

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    # MANTA_BENCH_BACKEND=gloo: developer knob to exercise the N>1 control flow (self-spawn, barriers, the result gather) on a box
    # with fewer GPUs than ranks -- ranks then share devices and the gather goes through host memory.  The driver's runs use RCCL.
    backend = os.environ.get("MANTA_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    if world > torch.cuda.device_count():
        # ranks share a device (the gloo self-test on a one-GPU box): a streamed upload's polling kernel and another PROCESS'
        # kernels time-slice the device against each other (measured: seconds per step) -- blocking uploads there
        os.environ["MANTA_AMD_NO_STREAM_UPLOAD"] = "1"
    torch.cuda.set_device(local_rank)
    # MANTA_BENCH_FORCE_DIST=1: developer knob -- run every torch.distributed call of the N>1 path (group init, barrier, timing
    # all-reduce, result gather) with ONE rank, so that the RCCL side can be exercised on a one-GPU box
    multi = world > 1 or os.environ.get("MANTA_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        cores_before = len(os.sched_getaffinity(0))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if os.environ.get("MANTA_BENCH_DEBUG"):
            print("bench: rank %d: %d -> %d cores in the affinity mask after init_process_group" % (rank, cores_before, len(os.sched_getaffinity(0))),
                  file=sys.stderr, flush=True)

    from manta_amd._capi import BatchOutput, Lib, SmallSvBatch, pack_spanning, pinned_copy, small_sv_text, assembly_text
    from manta_amd.shard import gather_bytes
    from oracle_lib import OracleLib, RefLib, asm_opts, have_ref
    from synth import config2_batch, config5_locus, unpack_locus

    spanning = args.workload == "spanning"
    n_loci = args.loci or (65536 if spanning else 10000)
    lib = Lib(device=local_rank)
    node_queue = multi and args.queue == "node"
    # --queue process: one process, N devices, one cost-ordered block queue inside the library (manta_node_*): the deployment shape
    # INTEGRATION.md B prescribes for a GenerateSVCandidates process.  Rank 0 (or the only process) drives the devices.
    proc_queue = args.queue == "process"
    n_dev      = args.gpus if proc_queue else world
    node       = None
    if proc_queue and rank == 0:
        from manta_amd._capi import Node
        node = Node(path=lib.path, devices=tuple(range(args.gpus)))
    qshm = qcount = None
    if node_queue:
        # one 32-bit block counter per step in POSIX shared memory (a fresh one every step: no reset to race with)
        import ctypes as _ct
        from multiprocessing import shared_memory
        qname = "manta_bench_q_%s" % os.environ.get("MASTER_PORT", "0")
        n_slots = args.steps + args.warmup + 8
        if rank == 0:
            try:
                shared_memory.SharedMemory(name=qname).unlink()  # (left over from a killed run)
            except FileNotFoundError:
                pass
            qshm = shared_memory.SharedMemory(name=qname, create=True, size=4 * n_slots)
            qshm.buf[:4 * n_slots] = bytes(4 * n_slots)
        dist.barrier()
        if rank != 0:
            qshm = shared_memory.SharedMemory(name=qname)
            try:  # (rank 0 owns the segment: keep this process' resource tracker from unlinking it a second time at exit)
                from multiprocessing import resource_tracker
                resource_tracker.unregister(qshm._name, "shared_memory")
            except Exception:
                pass
        qcount = (_ct.c_uint32 * n_slots).from_buffer(qshm.buf)
    step_no = [0]
    # measured on MI355X: one block's kernels already fill the device and concurrent blocks contend for the
    # per-wave HBM slabs, so the default is ONE block per call; --workers / --block-loci select the pipelined form
    workers = args.workers or 1
    block = args.block_loci or max(1, (n_loci + workers - 1) // workers)
    if (node_queue or proc_queue) and not args.block_loci:
        # one block per part of the node's batch (N parts of n_loci loci on one queue).  Measured on one MI355X with the round-4
        # kernels: the same 10 000 loci as ONE block 971 k loci/s, as blocks of 5 000 778 k, of 4 096 633 k -- every block pays
        # its fixed host work and its kernels' tails, so smaller blocks would show up as lost scaling that is not the queue's
        # (DESIGN.md 7); --block-loci overrides, --mix makes the parts unequal
        block = n_loci

    # ---- this rank's batch (outside the clock: synthetic data generation) ----
    if spanning:
        distinct = min(n_loci, 2048)  # generating a config-5 locus costs ~10 ms of numpy: larger batches repeat the 2048 digest loci
        # (one queue across the devices: the node's batch = n_dev parts of n_loci loci, every part the digest loci again -- every locus of
        # every part is then checked against the reference digests by whoever took it)
        one_queue = node_queue or proc_queue
        base = [config5_locus(i, seed0=555000 + (0 if one_queue else 1000003 * rank)) for i in range(distinct)]
        reps = (n_loci + distinct - 1) // distinct
        n_loci = reps * distinct
        reps *= n_dev if one_queue else 1
        loci = [base[i % distinct] for i in range(reps * distinct)]
        cuts = [(100, 100, 100, 100)] * distinct
        pb = pack_spanning([l[0] for l in base], [l[1] for l in base], [l[2] for l in base], cuts)

        def tile_off(off):  # offsets of `reps` copies laid end to end
            body = off[:-1]
            return np.concatenate([body + off[-1] * np.uint64(r) for r in range(reps)] + [off[-1:] * np.uint64(reps)]).astype(off.dtype)
        nb, n1, n2 = int(pb[1][-1]), int(pb[4][-1]), int(pb[6][-1])
        begin = np.concatenate([pb[2][:-1] + pb[2][-1] * np.uint32(r) for r in range(reps)] + [pb[2][-1:] * np.uint32(reps)]).astype(np.uint32)
        batch = (np.tile(pb[0][:nb], reps), tile_off(pb[1]), begin, np.tile(pb[3][:n1], reps), tile_off(pb[4]), np.tile(pb[5][:n2], reps),
                 tile_off(pb[6]), np.tile(pb[7], (reps, 1)))
        batch = tuple(np.ascontiguousarray(a) for a in batch)
        min_wl = np.tile(np.array([l[3] for l in base], dtype=np.uint32), reps)
        max_wl = np.tile(np.array([l[4] for l in base], dtype=np.uint32), reps)
        opts = asm_opts(minWordLength=41, minContigLength=75)
        n_out = reps * distinct
        out = BatchOutput(lib, "spanning", n_out, 10, 8192 * n_out + (1 << 20), 256 * n_out + 4096, 1024 * n_out + 4096,
                          pinned=not args.pageable)
    else:
        if args.mix and multi and rank % 4 == 3:
            batch = config2_batch(n_loci, seed=12345 + 1000003 * rank, n_reads=200)
        else:
            batch = config2_batch(n_loci, seed=12345 + 1000003 * rank)
        min_wl = max_wl = None
        opts = asm_opts(**ASM_K)
        if node_queue or proc_queue:
            # the node's batch = the devices' parts laid end to end (part r at loci [r * n_loci, (r + 1) * n_loci)); every rank holds
            # all of it, as every worker thread of one GenerateSVCandidates process sees the whole edge list.  The parts travel
            # as files in shared memory (every rank writes its own: nothing is pickled through the rendezvous store -- 1.2 GB
            # at 8 x 10 000 loci); a single process generates them all.
            def part_of(r):
                heavy = args.mix and r % 4 == 3
                return config2_batch(n_loci, seed=12345 + 1000003 * r, n_reads=200) if heavy else config2_batch(n_loci, seed=12345 + 1000003 * r)
            if multi and world == n_dev:
                shm_dir = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
                stem = os.path.join(shm_dir, "manta_bench_%s" % os.environ.get("MASTER_PORT", "0"))
                for i, a in enumerate(batch):
                    np.save("%s_r%d_a%d.npy" % (stem, rank, i), np.ascontiguousarray(a))
                dist.barrier()
                parts = [batch if r == rank else tuple(np.load("%s_r%d_a%d.npy" % (stem, r, i)) for i in range(len(batch))) for r in range(world)]
                dist.barrier()
                for i in range(len(batch)):
                    os.unlink("%s_r%d_a%d.npy" % (stem, rank, i))
            else:
                parts = [batch if r == 0 else part_of(r) for r in range(n_dev)]
            nb = [int(p[1][-1]) for p in parts]
            nr = [len(p[1]) - 1 for p in parts]
            nf = [int(p[4][-1]) for p in parts]
            cat_off = lambda offs, tot: np.concatenate([o[:-1] + np.asarray(sum(tot[:i]), dtype=o.dtype) for i, o in enumerate(offs)]
                                                       + [np.asarray([sum(tot)], dtype=offs[0].dtype)])
            batch = (np.concatenate([p[0][:n] for p, n in zip(parts, nb)] + [np.zeros(64, dtype=np.uint8)]),
                     cat_off([p[1] for p in parts], nb), cat_off([p[2] for p in parts], nr).astype(np.uint32),
                     np.concatenate([p[3][:n] for p, n in zip(parts, nf)] + [np.zeros(64, dtype=np.uint8)]),
                     cat_off([p[4] for p in parts], nf), np.concatenate([p[5] for p in parts]))
            del parts
        n_out = n_loci * (n_dev if (node_queue or proc_queue) else 1)
        out = BatchOutput(lib, "smallsv", n_out, 10, 4096 * n_out + (1 << 20), 128 * n_out + 4096, 512 * n_out + 4096,
                          pinned=not args.pageable)
    dev_batch = batch if args.pageable else tuple(pinned_copy(lib, a) for a in batch)
    n_reads = np.diff(batch[2])
    # ---- a STREAM of blocks, not one block in a loop: the plain single-queue smallsv run rotates three differently seeded config-2
    # batches over the steps (batch 0 = the digest workload), so that the library's "sizes from the previous run" paths, the caches
    # and the sorts see changing data.  (The node / process queue legs keep their one node batch: N parts with different seeds.)
    rotate = (not spanning) and (not node_queue) and (not proc_queue) and not os.environ.get("MANTA_BENCH_NO_ROTATE")
    stream = [(batch, dev_batch, out)]
    if rotate:
        for j in (1, 2):
            bj = config2_batch(n_loci, seed=12345 + 1000003 * rank + 7919 * j, n_reads=200 if (args.mix and multi and rank % 4 == 3) else 80)
            oj = BatchOutput(lib, "smallsv", n_loci, 10, 4096 * n_loci + (1 << 20), 128 * n_loci + 4096, 512 * n_loci + 4096, pinned=not args.pageable)
            stream.append((bj, bj if args.pageable else tuple(pinned_copy(lib, a) for a in bj), oj))
    step_i = [0]
    ran = [False] * len(stream)  # batches of the stream that have been through the library at least once (warm-up or timed)

    per_dev = [0] * n_dev

    def step():
        if spanning and proc_queue:
            if node is not None:
                per_dev[:] = node.spanning_batch(opts, SPAN_SC, JUMP, dev_batch, out, min_wl=min_wl, max_wl=max_wl, block_loci=block, n_workers=workers)
            return None
        elif spanning and node_queue:
            import ctypes as _ct
            slot = step_no[0]
            step_no[0] += 1
            lib.spanning_batch(opts, SPAN_SC, JUMP, dev_batch, out, min_wl=min_wl, max_wl=max_wl, block_loci=block, n_workers=workers,
                               serial_kernels=args.serial_kernels, shared_queue=_ct.addressof(qcount) + 4 * slot)
        elif spanning:
            lib.spanning_batch(opts, SPAN_SC, JUMP, dev_batch, out, min_wl=min_wl, max_wl=max_wl, block_loci=block, n_workers=workers, serial_kernels=args.serial_kernels)
        elif proc_queue:
            if node is not None:
                per_dev[:] = node.smallsv_batch(opts, SCORES, LARGE_INDEL, dev_batch, out, block_loci=block, n_workers=workers)
            return None
        elif node_queue:
            import ctypes as _ct
            slot = step_no[0]
            step_no[0] += 1
            lib.smallsv_batch(opts, SCORES, LARGE_INDEL, dev_batch, out, block_loci=block, n_workers=workers, serial_kernels=args.serial_kernels,
                              shared_queue=_ct.addressof(qcount) + 4 * slot)
        else:
            _, cur_dev, cur_out = stream[step_i[0] % len(stream)]
            ran[step_i[0] % len(stream)] = True
            step_i[0] += 1
            lib.smallsv_batch(opts, SCORES, LARGE_INDEL, cur_dev, cur_out, block_loci=block, n_workers=workers, serial_kernels=args.serial_kernels)
            if multi and not os.environ.get("MANTA_BENCH_SKIP_GATHER"):
                return gather_bytes(result_blob(cur_out), device="cuda" if backend == "nccl" else "cpu", force_collectives=True)
            return None
        if multi and not os.environ.get("MANTA_BENCH_SKIP_GATHER"):  # the final candidate gather (north star: "RCCL over xGMI only for the final candidate gather")
            return gather_bytes(result_blob(out), device="cuda" if backend == "nccl" else "cpu", force_collectives=True)
        return None

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    acc = dict(assemble_ms=0.0, schedule_ms=0.0, align_ms=0.0, h2d_ms=0.0, kernel_ms=0.0, d2h_ms=0.0, h2d_bytes=0, d2h_bytes=0,
               n_blocks=0, n_align_launches=0, dp_cells=0, wall_ms=0.0)
    gathered_bytes = 0
    t0 = time.perf_counter()
    used = [0] * len(stream)  # timed steps per batch of the stream
    for _ in range(args.steps):
        cur = (step_i[0] % len(stream)) if rotate else 0
        used[cur] += 1
        g = step()
        st = stream[cur][2].stats_dict()
        for k in acc:
            acc[k] += st[k]
        if g is not None:
            gathered_bytes += sum(len(x) for x in g)
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    results = out.decode(n_reads)
    taken = [l for l, r in enumerate(results) if r["status"] != -10]  # (-10: another rank took the locus' block, node queue)
    n_contigs = sum(len(results[l]["contigs"]) for l in taken)
    n_fail = sum(1 for l in taken if results[l]["status"] != 0)
    steps = args.steps
    loci_per_rank = [len(taken)]
    node_check = None
    def span_digest_mismatches(ids):
        """loci `ids` of this process' results against the reference digests (every part of the node batch repeats the digest loci)"""
        from test_digests import c5_text
        raw = open(os.path.join(ROOT, "tests", "golden", "config5_digests.bin"), "rb").read()
        bad = n = 0
        for l in ids:
            i = l % distinct
            if 32 * i + 32 > len(raw):
                continue
            r = results[l]
            got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"]) for a in r["aligns"]]
            bad += hashlib.sha256(c5_text(assembly_text(r), got).encode("latin-1")).digest() != raw[32 * i:32 * i + 32]
            n += 1
        return bad, n

    if node_queue and spanning:
        mism, checked = span_digest_mismatches(taken)
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([mism, checked, n_fail, len(taken)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(per, torch.tensor([len(taken)], dtype=torch.int64, device=dev))
        loci_per_rank = [int(x.item()) for x in per]
        node_check = [int(x) for x in t.tolist()]
        if rank == 0 and (node_check[0] or node_check[2] or node_check[3] != n_loci * world):
            raise SystemExit("PARITY FAILURE (node queue, spanning): %d of %d checked loci differ, %d loci failed, %d of %d loci taken"
                             % (node_check[0], node_check[1], node_check[2], node_check[3], n_loci * world))
    elif node_queue:
        # every rank checks what it took: part 0 is the digest workload (config-2, seed 12345), other parts against the CPU
        # restatement on a sample; the verdicts are summed over the ranks
        orc_n = OracleLib()
        dig_path = os.path.join(ROOT, "tests", "golden", "config2_digests.bin")
        raw = open(dig_path, "rb").read() if (os.path.exists(dig_path) and n_loci == 10000 and not args.mix) else b""
        mism = checked = 0
        rest = [l for l in taken if l >= n_loci or not raw]
        for l in taken:
            if raw and l < n_loci:
                mism += hashlib.sha256(small_sv_text(results[l]).encode("latin-1")).digest() != raw[32 * l:32 * l + 32]
                checked += 1
        for l in rest[::max(1, len(rest) // 16)]:
            reads, ref, cuts = unpack_locus(batch, l)
            mism += small_sv_text(results[l]) != orc_n.small_sv_locus(opts, SCORES, LARGE_INDEL, reads, ref, cuts)
            checked += 1
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([mism, checked, n_fail, len(taken)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(per, torch.tensor([len(taken)], dtype=torch.int64, device=dev))
        loci_per_rank = [int(x.item()) for x in per]
        node_check = [int(x) for x in t.tolist()]
        if rank == 0 and (node_check[0] or node_check[2] or node_check[3] != n_loci * world):
            raise SystemExit("PARITY FAILURE (node queue): %d of %d checked loci differ, %d loci failed, %d of %d loci taken"
                             % (node_check[0], node_check[1], node_check[2], node_check[3], n_loci * world))

    if proc_queue and spanning:
        loci_per_rank = list(per_dev)
        if rank == 0:
            mism, checked = span_digest_mismatches(taken)
            node_check = [mism, checked, n_fail, len(taken)]
            if mism or n_fail or len(taken) != n_loci * n_dev or sum(per_dev) != n_loci * n_dev:
                raise SystemExit("PARITY FAILURE (process queue, spanning): %d of %d checked loci differ, %d loci failed, %d of %d loci taken (per device: %s)"
                                 % (mism, checked, n_fail, len(taken), n_loci * n_dev, per_dev))
    elif proc_queue:
        loci_per_rank = list(per_dev)
        if rank == 0:
            # one process holds every result: part 0 is the digest workload, the other parts against the CPU restatement on a sample
            orc_n = OracleLib()
            dig_path = os.path.join(ROOT, "tests", "golden", "config2_digests.bin")
            raw = open(dig_path, "rb").read() if (os.path.exists(dig_path) and n_loci == 10000 and not args.mix) else b""
            mism = checked = 0
            for l in range(n_loci if raw else 0):
                mism += hashlib.sha256(small_sv_text(results[l]).encode("latin-1")).digest() != raw[32 * l:32 * l + 32]
                checked += 1
            rest = [l for l in taken if l >= n_loci or not raw]
            for l in rest[::max(1, len(rest) // (16 * n_dev))]:
                reads, ref, cuts = unpack_locus(batch, l)
                mism += small_sv_text(results[l]) != orc_n.small_sv_locus(opts, SCORES, LARGE_INDEL, reads, ref, cuts)
                checked += 1
            node_check = [mism, checked, n_fail, len(taken)]
            if mism or n_fail or len(taken) != n_loci * n_dev or sum(per_dev) != n_loci * n_dev:
                raise SystemExit("PARITY FAILURE (process queue): %d of %d checked loci differ, %d loci failed, %d of %d loci taken (per device: %s)"
                                 % (mism, checked, n_fail, len(taken), n_loci * n_dev, per_dev))

    if rank == 0:
        orc = OracleLib()
        # ---- parity (checker only; never part of the measured path) ----
        # rank 0's batch of the default config-2 run is the digest workload: every locus against the reference's SHA-256
        mism, checked, how = 0, 0, ""
        dig_path = os.path.join(ROOT, "tests", "golden", "config5_digests.bin" if spanning else "config2_digests.bin")
        raw = open(dig_path, "rb").read() if os.path.exists(dig_path) else b""
        n_dig = min(len(raw) // 32, n_loci)
        if spanning and (node_queue or proc_queue):
            checked, how = node_check[1], "reference digests (tests/golden/config5_digests.bin; every part of the node batch repeats the digest loci)" + (", summed over the ranks" if node_queue else "")
        elif spanning:
            from test_digests import c5_text
            from test_spanning_pipeline import oracle_locus
            for i in range(n_dig):
                r = results[i]
                got = [(a["score"], a["jump_insert_size"], a["jump_range"], a["begin1"], a["cigar1"], a["begin2"], a["cigar2"], a["is_uncut"])
                       for a in r["aligns"]]
                mism += hashlib.sha256(c5_text(assembly_text(r), got).encode("latin-1")).digest() != raw[32 * i:32 * i + 32]
            checked, how = n_dig, "reference digests (tests/golden/config5_digests.bin)"
        elif node_queue or proc_queue:
            checked, how = node_check[1], "reference digests (part 0, tests/golden/config2_digests.bin) + restatement samples of the other parts" + (", summed over the ranks" if node_queue else "")
        elif n_loci == 10000:  # the digest workload (config2_batch draws every locus of a batch from one stream: other sizes differ)
            for l in range(n_dig):
                mism += hashlib.sha256(small_sv_text(results[l]).encode("latin-1")).digest() != raw[32 * l:32 * l + 32]
            checked, how = n_dig, "reference digests (tests/golden/config2_digests.bin)"
        else:
            for l in range(0, n_loci, max(1, n_loci // 32)):
                reads, ref, cuts = unpack_locus(batch, l)
                mism += small_sv_text(results[l]) != orc.small_sv_locus(opts, SCORES, LARGE_INDEL, reads, ref, cuts)
                checked += 1
            how = "oracle spot check"
        stream_res = [results]
        if rotate and len(stream) > 1:
            # the other batches of the stream: every locus must have succeeded, a sample against the CPU restatement
            extra = 0
            for j, (bj, _, oj) in enumerate(stream[1:], start=1):
                if not ran[j]:
                    stream_res.append(None)
                    continue
                rj = oj.decode(np.diff(bj[2]))
                stream_res.append(rj)
                n_fail += sum(1 for r in rj if r["status"] != 0)
                for l in range(0, n_loci, max(1, n_loci // 16)):
                    reads, ref, cuts = unpack_locus(bj, l)
                    mism += small_sv_text(rj[l]) != orc.small_sv_locus(opts, SCORES, LARGE_INDEL, reads, ref, cuts)
                    extra += 1
            how += " + %d loci of the stream's two other batches vs the CPU restatement" % extra
        if mism or n_fail:
            raise SystemExit("PARITY FAILURE: %d of %d checked loci differ (%s), %d loci failed" % (mism, checked, how, n_fail))

        loci_total = n_loci * n_dev * steps
        value = loci_total / elapsed
        asm_sum, align_sum = acc["assemble_ms"], acc["align_ms"]
        if spanning:
            b_in = sum(sum(len(x) for x in loci[i][0]) + len(loci[i][1]) + len(loci[i][2]) for i in taken)
            b_ptr = b_out = 0
            for i in taken:
                r = results[i]
                b_out += len(loci[i][0]) * 8
                for c, a in zip(r["contigs"], r["aligns"]):
                    q = len(c["seq"])
                    span = (len(loci[i][1]) + len(loci[i][2])) if a["is_uncut"] else (len(loci[i][1]) + len(loci[i][2]) - 400)
                    b_ptr += (q + 1) * (span + 2)  # 1 B cells: the reference's own jump pointer matrix (GlobalJumpAligner.hpp:81-115)
                    b_out += q + 2 * ((len(loci[i][0]) + 7) // 8) + 64
            asm_bytes, align_bytes, align_name = sum(sum(len(x) for x in loci[i][0]) for i in taken) + b_out, b_ptr, "align_kernel<JUMP>"
            workload = ("BASELINE config[4] shape (NOT the metric's configuration): breakend loci, 200 reads x 250 bp, 1 % N, 10 % "
                        "tandem-repeat loci, minWordLength per locus from {25..75} in ONE launch, assemble + "
                        "GlobalJumpAligner(2,-8,-12,-1,-1;-100) on 700+700 bp windows + re-align rule")
        else:
            # (the stream's batches weighted by the timed steps each of them ran)
            tot_w = max(1, sum(used)) if rotate else 1
            b_in = b_ptr = b_out = asm_bytes = align_bytes = 0.0
            for j, rj in enumerate(stream_res):
                w = (used[j] / tot_w) if rotate else (1.0 if j == 0 else 0.0)
                if w == 0.0 or rj is None:
                    continue
                bj = stream[j][0]
                tk = taken if j == 0 else list(range(n_loci))
                bi, bp, bo, qb, wb = algorithmic_bytes_smallsv(bj, rj, tk)
                rin = sum(int(bj[1][bj[2][l + 1]] - bj[1][bj[2][l]]) for l in tk)
                b_in, b_ptr, b_out = b_in + w * bi, b_ptr + w * bp, b_out + w * bo
                asm_bytes, align_bytes = asm_bytes + w * (rin + bo), align_bytes + w * (qb + wb + bp)
            align_name = "align_kernel<LARGE_INDEL>"
            workload = ("BASELINE config[1]: synthetic small-indel loci, 80 reads/locus x150bp, k=31..76 step 5, "
                        "assemble + 10-mer trim + GlobalLargeIndelAligner(2,-8,-24,-1,-1;-100) on 1800 bp windows")
        # dominant kernel by HIP-event time summed over the timed region.  A launch = one block's kernel; the events of
        # different blocks overlap on the device (that is the point of the block queue), each is timed on its own stream.
        n_blocks = max(1, acc["n_blocks"])
        if align_sum >= asm_sum:
            dom, dom_bytes, dom_ms, dom_launches = align_name, align_bytes * steps, align_sum, n_blocks
            dom_note = "per block: all E-bucket launches of the block (they run concurrently on side streams)"
        else:
            # the assembler stage of a block: graph_kernel -> contig_kernel (asm_lds.hpp; the general assemble_kernel for what they
            # punt and under MANTA_AMD_ASM_PATH=general), timed by the library as ONE span of HIP events on the block's stream
            dom = "assemble_kernel" if os.environ.get("MANTA_AMD_ASM_PATH") == "general" else "assembler_stage"
            dom_bytes, dom_ms, dom_launches = asm_bytes * steps, asm_sum, n_blocks
            dom_note = ("one launch per block" if dom == "assemble_kernel" else
                        "per block: graph_kernel + contig_kernel (big piles: graph_big_kernel + repeat_big_kernel + contig_big_kernel, one round per word length) + assemble_kernel on what they hand back; one HIP-event "
                        "span, rocprofv3 lists the kernels separately: their averages add up to avg_launch_ms")
        avg_launch_ms = dom_ms / dom_launches
        achieved = (dom_bytes / dom_launches) / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        # roofline.traffic is NOT measured in this run: it is the HBM-side byte count of the builder's own rocprofv3 --pmc passes
        # (tools/profile_round.sh -> profiles/traffic*.json), quoted with its source and date
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_spanning.json" if spanning else "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("loci") == n_loci and tj.get("workload", "smallsv") == args.workload:
                    traffic = tj.get(dom)
                    traffic_source = "%s: %s, %s" % (os.path.relpath(tpath, ROOT), tj.get("source", "builder-run rocprofv3 --pmc passes"), tj.get("date", "round 2"))
                elif spanning and tj.get("workload") == "spanning" and tj.get("loci") and n_loci % tj["loci"] == 0 and tj.get(dom) is not None:
                    # the spanning block TILES the distinct loci the counter passes ran on (2 048: the digest set; under the profiler larger
                    # blocks run into the passes' time limit), so a launch's traffic is the counted bytes x the number of copies
                    traffic = int(tj[dom] * (n_loci // tj["loci"]))
                    traffic_source = ("%s: %s, %s; counted on %d loci, x %d: the block repeats those loci" %
                                      (os.path.relpath(tpath, ROOT), tj.get("source", "builder-run rocprofv3 --pmc passes"), tj.get("date", ""), tj["loci"], n_loci // tj["loci"]))
            except Exception:
                traffic = None
        o = {
            "metric": METRIC, "value": round(value, 1), "unit": "loci/s", "n_gpus": n_dev, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload, "loci_per_gpu": n_loci, "reads_per_locus": int(n_reads[0]),
                       "contigs_per_locus": round(n_contigs / max(1, len(taken)), 3),
                       "timed_region": "batch submit -> all results host-visible: H2D + kernels + D2H"
                                       + ((" + gather of the result blobs to rank 0 (torch.distributed, backend %s: %s)"
                                           % (backend, "RCCL" if backend == "nccl" else "host memory, developer self-test")) if (multi and not proc_queue) else ""),
                       "host_memory": "pageable" if args.pageable else "page-locked (manta_host_alloc)",
                       "block_loci": block, "workers_per_gpu": workers,
                       "stream": ("%d differently seeded batches rotated over the steps (batch 0 = the digest workload)" % len(stream)) if rotate else "one batch",
                       "parallelism": ("one process, %d device(s), one cost-ordered block queue inside the library (%s)" % (n_dev, "manta_node_spanning_batch" if spanning else "manta_node_smallsv_batch")) if proc_queue
                                      else ("one cost-ordered block queue across %d rank(s) (shared-memory counter, manta_batch_plan_t::shared_queue)" % world)
                                      if node_queue else ("loci sharded over %d rank(s); per rank a cost-ordered block queue" % world),
                       "queue": "process" if proc_queue else ("node" if node_queue else "rank"), "backend": backend if multi else None,
                       "dist_world": dist.get_world_size() if multi else 1, "loci_per_rank": loci_per_rank,
                       "mix": bool(args.mix and node_queue),
                       "parity": "%d loci vs %s: 0 mismatches" % (checked, how)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": int(dom_bytes / dom_launches), "avg_launch_ms": round(avg_launch_ms, 3),
                         "launches": dom_launches, "note": dom_note,
                         # extra (not the contract's fraction): the counter traffic against what this chip delivers to 64-byte
                         # random accesses (tools/microbench/gather_ceiling.hip, DESIGN.md 5.1: 3.36 TB/s) -- the bound the kernel sits on
                         "sector_traffic": None if not traffic or avg_launch_ms <= 0 else
                                           {"achieved_GBps": round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1), "random_access_ceiling_GBps": 3360.0,
                                            "frac": round(traffic / (avg_launch_ms * 1e-3) / 1e9 / 3360.0, 3),
                                            "traffic_over_algorithmic": round(traffic / max(1.0, dom_bytes / dom_launches), 1)}},
            "kernels_ms_per_step": {"assembler_stage": round(asm_sum / steps, 3), "schedule_kernel": round(acc["schedule_ms"] / steps, 3),
                                    "align_kernels": round(align_sum / steps, 3),
                                    "note": "HIP events per block, summed; blocks overlap, so the sum exceeds ms_per_step"},
            "pcie": {"h2d_MB_per_step": round(acc["h2d_bytes"] / steps / 1e6, 2), "d2h_MB_per_step": round(acc["d2h_bytes"] / steps / 1e6, 2),
                     "host_ms_per_step": {"h2d": round(acc["h2d_ms"] / steps, 2), "kernels": round(acc["kernel_ms"] / steps, 2),
                                          "d2h+compact": round(acc["d2h_ms"] / steps, 2)},
                     "gather_MB_per_step": round(gathered_bytes / steps / 1e6, 2)},
            "algorithmic_bytes_per_locus": {"in": b_in / max(1, len(taken)), "ptr": b_ptr / max(1, len(taken)), "out": b_out / max(1, len(taken)),
                                            "whole_path_GBps": round((b_in + b_ptr + b_out) / max(1, len(taken)) * n_loci * n_dev * steps / elapsed / 1e9, 2)},
            "dp_gcups": round(acc["dp_cells"] * (1 if proc_queue else world) / elapsed / 1e9, 2),
        }
        # ---- what the counters and the microbenchmarks say bounds the kernels (extra keys, quoted from profiles/ like roofline.traffic) ----
        # The contract's roofline object prices the dominant stage against its ALGORITHMIC bytes; this path is not bound by them.
        # Two figures that do describe the kernels: the aligner's real HBM rate (the back-pointer stream: counter bytes / this run's aligner
        # time) and, for the two assembler kernels, their use of the measured LDS ceilings (tools/microbench/lds_ceiling.hip).
        try:
            if True:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_spanning.json" if spanning else "traffic.json")))
                at = tj.get("align_kernel<JUMP>" if spanning else "align_kernel<LARGE_INDEL>")
                if at and tj.get("loci") and n_loci % tj["loci"] == 0 and align_sum > 0:
                    at = at * (n_loci // tj["loci"])
                    gbps = at / (align_sum / n_blocks * 1e-3) / 1e9  # (align_sum: HIP-event ms summed over the timed region's blocks)
                    o["roofline"]["aligner_hbm"] = {"kernel": align_name, "traffic_bytes_per_launch": int(at), "avg_launch_ms": round(align_sum / n_blocks, 3),
                                                    "achieved_GBps": round(gbps, 1), "peak": HBM_PEAK_GBPS, "frac": round(gbps / HBM_PEAK_GBPS, 4),
                                                    "note": "counter bytes of the builder's rocprofv3 --pmc passes (profiles/) over this run's aligner time: the back-pointer stream, written once, read along the traceback"}
            cpath = os.path.join(ROOT, "profiles", "ceilings.json")
            if not spanning and os.path.exists(cpath):
                o["roofline"]["issue"] = json.load(open(cpath))
        except Exception:
            pass
        # ---- device-resident kernel rate (extra key; round 1's headline): inputs in HBM, three kernels per step ----
        if not spanning and world == 1 and not args.no_extras:
            pipe = SmallSvBatch(lib, opts, SCORES, LARGE_INDEL)
            pipe.upload_packed(*batch)
            pipe.run()
            t1 = time.perf_counter()
            ks = dict(assemble_ms=0.0, schedule_ms=0.0, align_ms=0.0)
            for _ in range(max(2, steps)):
                pipe.run()
                s2 = pipe.stats()
                for k in ks:
                    ks[k] += s2[k]
            dt = time.perf_counter() - t1
            nrun = max(2, steps)
            o["kernel_only"] = {"value": round(n_loci * nrun / dt, 1), "unit": "loci/s", "note": "inputs resident in HBM, no result download",
                                "ms_per_step": round(dt / nrun * 1e3, 3),
                                "kernels_ms": {k: round(v / nrun, 3) for k, v in ks.items()}}
            pipe.close()
        # ---- the same batch delivered as packed piles (what the read-pile builder emits from BAM records) ----
        if not spanning and world == 1 and not args.no_extras:
            from manta_amd._capi import pack_piles
            piles = pack_piles(batch[0], batch[1], batch[2])
            pin = piles if args.pageable else piles.pinned(lib)
            refs_p, roff_p, cuts_p = dev_batch[3], dev_batch[4], dev_batch[5]
            out2 = BatchOutput(lib, "smallsv", n_loci, 10, len(out.seq), len(out.bits), len(out.cig), pinned=not args.pageable)

            def step_piles():
                lib.smallsv_batch_piles(opts, SCORES, LARGE_INDEL, pin, refs_p, roff_p, cuts_p, out2, block_loci=block, n_workers=workers,
                                        serial_kernels=args.serial_kernels)
            step_piles()
            nrun = max(2, steps)
            t2 = time.perf_counter()
            h2d = 0
            for _ in range(nrun):
                step_piles()
                h2d += out2.stats_dict()["h2d_bytes"]
            dt2 = time.perf_counter() - t2
            same = [small_sv_text(r) for r in out2.decode(n_reads)] == [small_sv_text(r) for r in results]
            if not same:
                raise SystemExit("PARITY FAILURE: packed-pile input and 1-byte-per-base input disagree")
            o["packed_input"] = {"value": round(n_loci * nrun / dt2, 1), "unit": "loci/s", "ms_per_step": round(dt2 / nrun * 1e3, 3),
                                 "h2d_MB_per_step": round(h2d / nrun / 1e6, 2),
                                 "note": "read piles as 2-bit codes + N bitmap (manta_packed_piles_t, 0.375 B/base) instead of 1 B/base; "
                                         "same timed region; results identical to the default run"}
        # ---- loci of very different sizes in one batch (extra key): read counts drawn log-uniformly from 3..1000, Manta's production word
        # lengths (IterativeAssemblerOptions.hpp:33-58: 41..76 step 5) -- the rate, and how the assembler stage routed the loci (the LDS
        # pipeline's two classes, what they handed back, what lies outside both envelopes)
        if not spanning and world == 1 and not args.no_extras and not os.environ.get("MANTA_BENCH_NO_MIXED"):
            try:
                from synth import mixed_shape_batch
                n_mx = min(2048, max(8, n_loci))
                mx = mixed_shape_batch(n_mx, seed=777, hi=int(os.environ.get("MANTA_BENCH_MIXED_HI", "1000")))  # (the knob: the emulator flow test)
                mx_dev = mx if args.pageable else tuple(pinned_copy(lib, a) for a in mx)
                mx_reads = np.diff(mx[2])
                mx_opts = asm_opts(minWordLength=41, maxWordLength=76, wordStepSize=5)
                tot_b = int(mx[1][-1])
                out3 = BatchOutput(lib, "smallsv", n_mx, 10, 2 * tot_b // 8 + 4096 * n_mx + (1 << 20), 40 * int(mx_reads.sum()) // 64 + 128 * n_mx + 4096,
                                   512 * n_mx + 4096, pinned=not args.pageable)
                lib.smallsv_batch(mx_opts, SCORES, LARGE_INDEL, mx_dev, out3)
                t3 = time.perf_counter()
                for _ in range(2):
                    lib.smallsv_batch(mx_opts, SCORES, LARGE_INDEL, mx_dev, out3)
                dt3 = time.perf_counter() - t3
                st3 = out3.stats_dict()
                r3 = out3.decode(mx_reads)
                bad3 = sum(1 for r in r3 if r["status"] != 0)
                # every locus against the SHA-256 digests of the unmodified reference's output (tests/golden/mixed_digests.bin, written by
                # tests/golden/make_digests.py for exactly this batch); other sizes: a sample against the CPU restatement
                mx_dig_path = os.path.join(ROOT, "tests", "golden", "mixed_digests.bin")
                mx_raw = open(mx_dig_path, "rb").read() if (os.path.exists(mx_dig_path) and n_mx == 2048 and not os.environ.get("MANTA_BENCH_MIXED_HI")) else b""
                if len(mx_raw) == 32 * n_mx:
                    bad3 += sum(hashlib.sha256(small_sv_text(r3[l]).encode("latin-1")).digest() != mx_raw[32 * l:32 * l + 32] for l in range(n_mx))
                    mx_parity = "%d loci vs reference digests (tests/golden/mixed_digests.bin): 0 mismatches; every locus status 0" % n_mx
                else:
                    chk = sorted(set([0, n_mx - 1] + [int(i) for i in np.argsort(mx_reads)[[0, n_mx // 2, -1]]]))
                    for l in chk:
                        reads, ref, cuts = unpack_locus(mx, l)
                        bad3 += small_sv_text(r3[l]) != orc.small_sv_locus(mx_opts, SCORES, LARGE_INDEL, reads, ref, cuts)
                    mx_parity = "%d loci vs the CPU restatement (smallest, median, largest pile, first, last): 0 mismatches; every locus status 0" % len(chk)
                if bad3:
                    raise SystemExit("PARITY FAILURE (mixed_shape): %d loci failed or differ from the reference" % bad3)
                o["mixed_shape"] = {"value": round(n_mx * 2 / dt3, 1), "unit": "loci/s", "ms_per_step": round(dt3 / 2 * 1e3, 3), "loci": n_mx,
                                    "reads_per_locus": {"min": int(mx_reads.min()), "median": int(np.median(mx_reads)), "max": int(mx_reads.max()),
                                                        "mean": round(float(mx_reads.mean()), 1)},
                                    "word_lengths": "41..76 step 5",
                                    "routing": {"lds_small_class": int(st3["n_loci_lds_small"]), "lds_big_class": int(st3["n_loci_lds_big"]),
                                                "handed_back_to_general_kernel": int(st3["n_loci_handed_back"]), "outside_both_classes": int(st3["n_loci_general"]),
                                                "frac_off_the_lds_pipeline": round((st3["n_loci_handed_back"] + st3["n_loci_general"]) / max(1, n_mx), 4)},
                                    "parity": mx_parity,
                                    "note": "read counts log-uniform 3..1000 x 150 bp, one manta_smallsv_batch call per step, same timed region"}
            except SystemExit:
                raise
            except Exception as e:  # (an extra leg never takes the line down)
                o["mixed_shape"] = {"error": str(e)[-300:]}
        # ---- the config-4/5 shape (extra key, driver-visible): breakend loci, 200 reads x 250 bp, mixed word lengths, GlobalJumpAligner --
        # the same script with --workload spanning on 65 536 loci (2 048 distinct, tiled) in a process of its own (its own context and arenas), its line cut
        # down to what a reader of this line needs: rate, the assembler stage's roofline object, kernel times, the parity string
        if (not spanning and world == 1 and not args.no_extras and not os.environ.get("MANTA_BENCH_NO_SPANNING")
                and os.path.abspath(lib.path) == os.path.join(ROOT, "manta_amd", "libmanta_amd.so")):
            try:
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", "spanning", "--loci", os.environ.get("MANTA_BENCH_SPANNING_LOCI", "65536"), "--steps", "2", "--warmup", "1",
                                     "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=900)
                rows = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                if pr.returncode == 0 and rows:
                    sp = json.loads(rows[-1])
                    o["spanning"] = {"value": sp["value"], "unit": sp["unit"], "ms_per_step": sp["ms_per_step"], "steps": sp["steps"], "warmup": sp["warmup"],
                                     "loci": sp["config"]["loci_per_gpu"], "workload": sp["config"]["workload"], "parity": sp["config"]["parity"],
                                     "timed_region": sp["config"]["timed_region"], "roofline": sp["roofline"],
                                     "kernels_ms_per_step": sp["kernels_ms_per_step"], "algorithmic_bytes_per_locus": sp["algorithmic_bytes_per_locus"],
                                     "dp_gcups": sp["dp_gcups"], "note": "python bench.py --workload spanning --loci %s --steps 2 --warmup 1, run by this script" % os.environ.get("MANTA_BENCH_SPANNING_LOCI", "65536")}
                else:
                    o["spanning"] = {"error": (pr.stderr or pr.stdout)[-300:]}
            except Exception as e:  # (an extra leg never takes the line down)
                o["spanning"] = {"error": str(e)[-300:]}
        # ---- CPU baseline: the reference's own sources (oracle/_ref) on this box's host cores ----
        # ---- the candidate-level rate (extra key): SVCandidateAssemblyRefiner::getCandidateAssemblyDataBatch, the C++ host adapter a
        # GenerateSVCandidates process would call, on 10 000 config-2 shaped complex candidates: reference and read
        # callbacks, packing, the device batch, the per-candidate host glue (tools/cpp/perf_refiner.cpp, built by __graft_entry__.build())
        exe = os.path.join(ROOT, "tools", "cpp", "perf_refiner")
        if (not spanning and world == 1 and not args.no_extras and os.path.exists(exe)
                and os.path.abspath(lib.path) == os.path.join(ROOT, "manta_amd", "libmanta_amd.so")):
            try:
                pr = subprocess.run([exe, "10000", "0"], capture_output=True, text=True, timeout=600)
                rows = [l for l in pr.stdout.splitlines() if l.startswith("{")]
                if pr.returncode == 0 and rows:
                    o["refiner_batch"] = json.loads(rows[-1])
                else:
                    o["refiner_batch"] = {"error": (pr.stderr or pr.stdout)[-300:]}
            except Exception as e:  # (an extra leg never takes the line down)
                o["refiner_batch"] = {"error": str(e)[-300:]}
            # ... and the same batch through manta_amd_dropin::BatchRefiner over Manta's REAL types (::SVCandidate in, ::SVCandidateAssemblyData
            # out, conversions inside the clock): the drop-in build of oracle/_ref -- the reference's own headers -- holds the probe
            # (oracle/ref_refiner_driver.cpp: ref_perf_batch_refiner); the file is prebuilt where /root/reference existed
            dso = os.path.join(ROOT, "oracle", "_ref", "libmanta_ref_dropin_gpu.so")
            if isinstance(o.get("refiner_batch"), dict) and "error" not in o["refiner_batch"] and os.path.exists(dso):
                try:
                    import ctypes
                    dl = ctypes.CDLL(dso)
                    if hasattr(dl, "ref_perf_batch_refiner"):
                        res3 = (ctypes.c_double * 3)()
                        ht = int(o["refiner_batch"].get("host_threads", 16))
                        real = {}
                        for name, pt in (("threaded_plan", ht), ("sequential_plan", 1)):
                            if dl.ref_perf_batch_refiner(10000, ht, pt, res3) == 0:
                                real[name] = {"candidates_per_s": round(10000 / res3[0], 1), "seconds": round(res3[0], 4), "refined_svs": int(res3[1]), "contigs": int(res3[2])}
                        real["call"] = "manta_amd_dropin::BatchRefiner::getCandidateAssemblyDataBatch(std::vector<::SVCandidate>, ..., std::vector<::SVCandidateAssemblyData>&)"
                        o["refiner_batch"]["real_types"] = real
                except Exception as e:
                    o["refiner_batch"]["real_types"] = {"error": str(e)[-300:]}
        if world == 1 and not args.no_cpu_baseline:
            cores = cores_available()
            kind, cpu = ("reference", RefLib()) if have_ref() else ("port", orc)
            if spanning:
                from concurrent.futures import ThreadPoolExecutor
                from test_spanning_pipeline import oracle_locus
                n_s = args.cpu_sample or min(n_loci, max(64, cores))

                def one(i):
                    reads, ref1, ref2, k, kmax = loci[i]
                    oracle_locus(cpu, asm_opts(minWordLength=k, maxWordLength=kmax, minContigLength=75), reads, ref1, ref2, (100, 100, 100, 100))
                tc = time.perf_counter()
                one(0)
                secs1 = time.perf_counter() - tc
                tc = time.perf_counter()
                with ThreadPoolExecutor(cores) as ex:
                    list(ex.map(one, range(n_s)))
                secs = time.perf_counter() - tc
                o["cpu_baseline"] = {"value": round(n_s / secs, 2), "unit": "loci/s", "cores": min(cores, n_s), "kind": kind,
                                     "sample": "%d loci of the same workload (runIterativeAssembler + GlobalJumpAligner call by call as "
                                               "alignJumpContigs does) on %d host threads, %.1f s wall; single thread: 1 locus in %.2f s"
                                               % (n_s, min(cores, n_s), secs, secs1),
                                     "single_thread_value": round(1.0 / secs1, 2)}
            else:
                # The unmodified reference on host threads (one aligner per thread, as GenerateSVCandidates.cpp:232-266), through the thread
                # harness of oracle/bench_harness.hpp: threads started and pinned BEFORE the clock, at least 64 loci per thread (the batch's
                # loci round robin), a few seconds per thread count -- 1, 8, 32, the physical cores, every hardware thread.  (Round 5 timed
                # 8 loci per thread with thread start-up inside the clock: 716 loci/s on "256 threads" = 11 x one thread.)
                topo = host_cpu_topology()
                sb = config2_batch(min(n_loci, 2048), seed=12345)
                counts = sorted(set(t for t in (1, 8, 32, topo["physical_cores"], cores) if 1 <= t <= cores))
                budget = 24.0 / len(counts)
                table, best = [], None
                rate1 = None
                for t in counts:
                    want = 96 if t == 1 else 64 * t
                    if hasattr(cpu.lib, cpu.prefix + "bench_small_sv_timed"):
                        secs, done = cpu.bench_small_sv_timed(opts, SCORES, LARGE_INDEL, sb[0], sb[1], sb[2], sb[3], sb[4], (100, 100, 800, 800), t, want, budget)
                    else:  # (an oracle/_ref built before the harness existed)
                        nn = min(len(sb[2]) - 1, want)
                        sbn = config2_batch(nn, seed=12345)
                        secs, done = cpu.bench_small_sv(opts, SCORES, LARGE_INDEL, sbn[0], sbn[1], sbn[2], sbn[3], sbn[4], (100, 100, 800, 800), t), nn
                    rate = done / secs if secs > 0 else 0.0
                    if t == 1:
                        rate1 = rate
                    row = {"threads": t, "loci": int(done), "seconds": round(secs, 2), "loci_per_s": round(rate, 1),
                           "speedup_vs_1": round(rate / rate1, 1) if rate1 else None}
                    table.append(row)
                    if best is None or rate > best["loci_per_s"]:
                        best = row
                ideal = (rate1 or 0.0) * topo["physical_cores"]
                o["cpu_baseline"] = {"value": best["loci_per_s"], "unit": "loci/s", "cores": best["threads"], "kind": kind,
                                     "sample": "the best row of `scaling`: %d loci of the same workload on %d pinned host threads (one aligner per thread, as "
                                               "GenerateSVCandidates.cpp:232-266), %.1f s wall, threads started before the clock" % (best["loci"], best["threads"], best["seconds"]),
                                     "single_thread_value": round(rate1 or 0.0, 2),
                                     "scaling": table, "host": topo,
                                     "allocator": "glibc malloc, MALLOC_ARENA_MAX=%s" % os.environ.get("MALLOC_ARENA_MAX", "default (8 x cores)"),
                                     "single_thread_x_physical_cores": round(ideal, 1),
                                     "note": ("the best row is %.2f of single-thread rate x physical cores" % (best["loci_per_s"] / ideal if ideal else 0.0))
                                             + ("; the container's cgroup limits the process to %.1f CPUs' worth of time" % topo["cgroup_cpu_quota"] if topo["cgroup_cpu_quota"] else "")}
        line = json.dumps(o)
    # The JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio (seen after the line when stdout
    # is a pipe), so every rank flushes its C buffers, the group is torn down, and only then rank 0 prints.
    import ctypes
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    if qshm is not None:
        del qcount
        qshm.close()
    if multi:
        dist.barrier()
        if qshm is not None and rank == 0:
            qshm.unlink()
        dist.destroy_process_group()
        libc.fflush(None)
    if rank == 0:
        print(line, flush=True)


if __name__ == "__main__":
    main()
