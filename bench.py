#!/usr/bin/env python3
"""bench.py -- Gbp/s of the `match` hot path on synthetic 150 bp reads, k=31 (BASELINE.json configs[1]).

A step = one pass of the hot path over one batch: gs_match_reset + gs_match_submit (reads already resident
in HBM) + the per-taxid table merge (RCCL all-reduce / bitmap all-gather when N > 1) + gs_match_finish.
Weak scaling: every rank classifies `--reads` reads of its own slice of the global read stream against a
full replica of the store.  Rank 0 prints ONE JSON line: `value` is configs[1]; at N = 1 the same line carries
extra objects for the workloads the headline does not show:

  large_store   match against a 47 M-k-mer / 526-value store (1 GiB table: HBM resident, not Infinity-Cache resident)
  filter        the `filter` goal's kernel against the XOR index filter of the same store (~47 M keys, 27 hashes)
  end_to_end    configs[1] again with the reads in page-locked HOST memory (gs_match_submit_async: PCIe included)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--cpu-seconds S] [--legs main,large,filter,e2e,striped,dbbuild,long]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

GS_BENCH_FORCE_MERGE=1 runs the RCCL merge path even at N = 1 (rehearsal of the multi-GPU code on one GPU).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
K = 31
BYTES_PER_READ = READ_LEN + (READ_LEN - K + 1) * 64   # sequence bytes once + one 64 B line per k-mer
HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: 8 TB/s HBM3E
# committed rocprofv3 --pmc summaries per workload, newest first (r01's large-store passes ran on a smaller store)
PROFILE_ROUNDS = {"match": ("r02", "r01"), "large_store": ("r02",), "filter": ("r02",)}


def _usable_cores():
    """host cores this process may really use: affinity mask capped by the cgroup CPU quota (cpu.max)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("GS_CPU_THREADS", n))))


class _DevArray:
    """zero-copy view of a raw device pointer for torch.as_tensor"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (n,), "typestr": typestr, "version": 2}


def _pmc_traffic(name):
    """(bytes per launch through the fabric, file) from a committed rocprofv3 --pmc summary of the same command
    (FETCH_SIZE + WRITE_SIZE, KiB units; 64-byte requests are counted exactly), or (None, None)"""
    for rnd in PROFILE_ROUNDS[name]:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}_pmc_summary.csv")
        if os.path.exists(path):
            vals = dict(l.strip().split(",")[:2] for l in open(path) if l[0] not in "#c" and "," in l)
            if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
                return int((float(vals["FETCH_SIZE"]) + float(vals["WRITE_SIZE"])) * 1024), os.path.relpath(path, ROOT)
    return None, None


def _pmc_issue_bound(name, kern_ms, n_simd=1024, clock_ghz=2.4):
    """the kernel's instruction-issue floor from the same committed summary: a SIMD issues one wave64 VALU instruction
    per four cycles; SQ_INSTS_VALU per launch over the device's 1024 SIMDs at 2.4 GHz (GRBM_GUI_ACTIVE of the summary: 2.38)"""
    for rnd in PROFILE_ROUNDS[name]:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}_pmc_summary.csv")
        if os.path.exists(path):
            vals = dict(l.strip().split(",")[:2] for l in open(path) if l[0] not in "#c" and "," in l)
            if "SQ_INSTS_VALU" in vals:
                ms = float(vals["SQ_INSTS_VALU"]) * 4 / n_simd / (clock_ghz * 1e9) * 1e3
                return {"valu_instructions_per_launch": float(vals["SQ_INSTS_VALU"]), "ms": round(ms, 3),
                        "frac": round(ms / kern_ms, 4), "note": "VALU issue floor (one wave64 instruction per SIMD and 4 cycles) over the "
                        "measured kernel time; counters of %s, not measured in this run" % os.path.relpath(path, ROOT)}
    return None


def _kernel_ms(obj, launch, reps, warm=2):
    """average device time of one launch: HIP events recorded by the library on its own stream (cfg.profile)"""
    for _ in range(warm):
        launch()
    obj.sync()
    l0, ms0 = obj.kernel_time()
    t0 = time.perf_counter()
    for _ in range(reps):
        launch()
    obj.sync()
    wall = (time.perf_counter() - t0) / reps * 1e3
    l1, ms1 = obj.kernel_time()
    return (ms1 - ms0) / max(1, l1 - l0), wall


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--check-reads", type=int, default=200_000, help="reads cross-checked against the oracle")
    ap.add_argument("--mode", choices=["sharded", "striped", "partitioned"], default="sharded",
                    help="sharded: store replicated, reads sharded (configs[1]/[3], the default bench line); "
                         "striped: ONE store, its record table split over the GPUs' HBM, foreign record lines loaded over "
                         "xGMI by the same fused kernel, reads sharded (configs[4]); "
                         "partitioned: round 1's split pipeline, store split by key hash, k-mers routed by all-to-all")
    ap.add_argument("--stripes", type=int, default=8,
                    help="--mode striped with ONE rank: stripes of the record table, all in this GPU's HBM (prices the "
                         "stripe arithmetic of the kernel; with N ranks there is one stripe per rank)")
    ap.add_argument("--genera", type=int, default=0, help="size of the synthetic store: genera of 20 species (0: configs[1]'s store)")
    ap.add_argument("--legs", default="main,large,filter,e2e,striped,dbbuild,long",
                    help="comma list; large / filter / e2e / striped are the extra N = 1 objects (main always runs)")
    args = ap.parse_args()
    legs = set(args.legs.split(","))

    import torch
    import torch.distributed as dist

    import genestrip_amd as ga
    from genestrip_amd import synth
    from genestrip_amd.distributed import merge_run_state, partitioned_finish, partitioned_match_batch, striped_store

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (genestrip_amd has no CPU fallback)")
    # GS_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box: several ranks share GPU 0 and the merge runs over
    # gloo on host copies, so the multi-rank control flow (barriers, collectives, rank-0-only legs) can be exercised
    # without a multi-GPU node.  Its numbers mean nothing; the real run uses RCCL ("nccl"), one GPU per rank.
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    rehearsal = backend == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev  # where the small control tensors of the collectives live
    partitioned = args.mode == "partitioned"
    striped = args.mode == "striped"
    force_merge = os.environ.get("GS_BENCH_FORCE_MERGE", "") == "1" or partitioned
    use_dist = world > 1 or force_merge
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- inputs: store replica per GPU, this rank's slice of the read stream generated directly in HBM
    db = synth.SynthDB(k=K, genera=args.genera, species_per_genus=20) if args.genera else synth.SynthDB(k=K)
    others = []
    if striped and world > 1:  # one stripe per rank, the others attached through IPC handles
        store = striped_store(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, device=local_rank)
    elif striped:
        others = ga.DeviceKMerStore.striped(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, devices=(local_rank,) * args.stripes)
        store = others.pop(0)
    else:
        store = ga.DeviceKMerStore(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, device=local_rank,
                                   n_parts=world if partitioned else 1, part=rank if partitioned else 0, partition=partitioned)
    info = store.info
    n = args.reads
    first = rank * n
    gen = torch.from_numpy(db.genomes).to(dev)
    dseq = torch.empty(n * READ_LEN, dtype=torch.uint8, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=READ_LEN, first=first)
    torch.cuda.synchronize()

    m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
    st = m.device_state()
    nv = db.n_values
    t_sums = torch.as_tensor(_DevArray(st["sums"], nv * ga.N_SUMS, "<i8"), device=dev)
    t_max = torch.as_tensor(_DevArray(st["max_keys"], nv, "<i8"), device=dev)
    t_dsum = torch.as_tensor(_DevArray(st["dsums"], nv * ga.N_DCOLS, "<f8"), device=dev)
    t_bits = torch.as_tensor(_DevArray(st["bitmap"], st["bitmap_words"], "<i4"), device=dev)

    def step(n_reads=n):
        m.reset()
        if partitioned:
            partitioned_match_batch(m, K, dseq, doff, n_reads, first)
            return partitioned_finish(m, t_sums, t_max, t_dsum)
        m.submit(dseq, doff, first, n_reads=n_reads)
        if use_dist:
            m.device_state()  # syncs the library's stream and refreshes the compact unique bitmap (same pointers)
            if rehearsal:  # gloo: merge host copies, write them back, mark the bitmap as merged
                hs, hm, hd, hb = t_sums.cpu(), t_max.cpu(), t_dsum.cpu(), t_bits.cpu()
                merge_run_state(hs, hm, hd, hb, force=force_merge)
                t_sums.copy_(hs), t_max.copy_(hm), t_dsum.copy_(hd), t_bits.copy_(hb)
                torch.cuda.synchronize()
                m.or_bitmap(t_bits.data_ptr(), 1)
            else:
                merge_run_state(t_sums, t_max, t_dsum, t_bits, force=force_merge,
                                or_parts=lambda g, w: m.or_bitmap(g.data_ptr(), w))
        return m.finish()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    launches0, ms0 = m.kernel_time()
    barrier()
    t0 = time.perf_counter()
    table = None
    for _ in range(args.steps):
        table, _ = step()
    barrier()
    elapsed = time.perf_counter() - t0
    launches1, ms1 = m.kernel_time()
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    n_launch = max(1, launches1 - launches0)
    kern_ms = (ms1 - ms0) / n_launch
    kernel_name = "gs_match_kernel"
    if partitioned:  # the work is spread over encode / exchange / probe / reduce: price the whole step
        kern_ms = elapsed / args.steps * 1e3
        kernel_name = "encode + all-to-all + gs_probe_keys_kernel + all-to-all + reduce (whole step)"

    total_bases = float(world) * n * READ_LEN * args.steps
    gbps = total_bases / elapsed / 1e9
    achieved = n * BYTES_PER_READ / (kern_ms * 1e-3) / 1e9  # GB/s, algorithmic bytes of one launch / its duration

    # measured memory-side traffic of one launch: rocprofv3 --pmc passes of this same command (only valid for the
    # default workload the profile was taken on)
    traffic, traffic_src = (None, None)
    if n == 10_000_000 and not partitioned and not striped and not args.genera:
        traffic, traffic_src = _pmc_traffic("match")

    out = {
        "metric": "Gbp/s classified (match goal), k=31, 150bp reads; bit-exact CSV counts",
        "value": round(gbps, 3), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "match: %d synthetic 150 bp reads per GPU, k=31, %d-k-mer / %d-taxid store resident in HBM "
                               "(BASELINE.json %s)" % (n, db.n_entries, len(db.species_vi),
                                                       "configs[4]: a store spread over the GPUs' HBM" if striped else "configs[1]"),
                   "reads_per_gpu": n, "read_len": READ_LEN, "k": K, "store_kmers": int(db.n_entries),
                   "store_record_bytes": int(info.rec_bytes), "store_table_bytes": int(info.table_bytes),
                   "gate_bytes": int(info.mgate_bytes or info.gate_bytes),
                   "parallelism": ("DB-partitioned x%d, k-mers routed by all-to-all" % world) if partitioned
                   else ("reads sharded x%d, ONE store with its record table in %d stripes (%s)"
                         % (world, info.n_stripes, "one per GPU, foreign lines over xGMI" if world > 1 else "all in this GPU's HBM")) if striped
                   else ("read-sharded x%d, store replicated" % world)},
        # `frac` follows SURVEY 8(d)'s convention (one 64-byte line per k-mer position over the HBM peak).  For this
        # store it is NOT an HBM measurement: the 64 MiB table sits in the 256 MiB Infinity Cache and the minimizer gate
        # removes the lines of most misses, so the kernel is bound by the fabric's rate of random 64-byte requests
        # (~59 G/s, tools/probe_bw.hip).  `measured_frac` is what the counters saw; `large_store` below is the
        # HBM-resident case.
        "roofline": {"bound": "fabric random 64-byte line rate (table Infinity-Cache resident; HBM peak is the SURVEY 8d convention's denominator)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "measured_frac": None if traffic is None else round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic_note": None if traffic is None else
                     "bytes per launch through the fabric (rocprofv3 --pmc FETCH_SIZE+WRITE_SIZE, %s), not measured in this run" % traffic_src,
                     "kernel": kernel_name, "kernel_ms": round(kern_ms, 4),
                     "algorithmic_bytes_per_launch": n * BYTES_PER_READ,
                     "issue_bound": None if traffic is None else _pmc_issue_bound("match", kern_ms)},
    }

    # ---- parity gate.  N = 1: the first --check-reads reads against the CPU oracle (bit-exact integer table).
    # N > 1 sharded: every rank must hold the same merged table, and rank 0 re-checks its own slice against the oracle.
    # N > 1 partitioned: the first check_reads / N reads of EVERY rank's slice go through the collective pipeline once
    # more and rank 0 compares the merged table with the oracle over exactly those reads.
    nchk = min(args.check_reads, n)
    ptable = None
    if partitioned and world > 1:
        per = max(1, nchk // world)
        m.reset()
        partitioned_match_batch(m, K, dseq, doff, per, first)
        ptable, _ = partitioned_finish(m, t_sums, t_max, t_dsum)
    if use_dist and world > 1:
        digest = torch.tensor([int(np.asarray(table, dtype=np.int64).sum() % (1 << 62))], dtype=torch.int64, device=cdev)
        lo, hi = digest.clone(), digest.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out["parity"] = {"merged_table_identical_on_all_ranks": bool(lo.item() == hi.item())}
    if rank == 0:
        from oracle import gs_oracle as orc
        cores = _usable_cores()
        odb = orc.DB(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, bloom_gate=True)
        orun = orc.MatchRun(odb)
        t1 = time.perf_counter()
        if ptable is not None:
            per = max(1, nchk // world)
            for r in range(world):
                seq, off = synth.reads_host(db.genomes, per, read_len=READ_LEN, first=r * n)
                orun.submit(seq, off, first_read_no=r * n, threads=cores, per_read=False)
            nchk = per * world
        else:
            seq, off = synth.reads_host(db.genomes, nchk, read_len=READ_LEN, first=first)
            orun.submit(seq, off, first_read_no=first, threads=cores, per_read=False)
        dt_chk = time.perf_counter() - t1
        otable, _ = orun.finish()
        if ptable is not None:
            gtable = ptable
        elif world > 1:
            # rank 0 alone re-checks its own slice: no collectives here (the other ranks are already at the barrier)
            m.reset()
            m.submit(dseq, doff, first, n_reads=nchk)
            gtable, _ = m.finish()
        else:
            gtable, _ = step(nchk)
        par = out.setdefault("parity", {})
        par["reads_checked"] = nchk
        par["bit_exact"] = bool(np.array_equal(otable, gtable))
        if not par["bit_exact"] or par.get("merged_table_identical_on_all_ranks") is False:
            out["value"] = None  # a throughput without parity does not count
        if world == 1 and args.cpu_seconds > 0:
            rate = nchk / dt_chk
            ns = int(min(n, max(nchk, rate * args.cpu_seconds)))
            seq, off = synth.reads_host(db.genomes, ns, read_len=READ_LEN, first=first)
            orun = orc.MatchRun(odb)
            dt, passes = 0.0, 0
            while passes == 0 or (dt < 0.6 * args.cpu_seconds and passes < 8):  # (the estimate above comes from a cold, short run)
                t1 = time.perf_counter()
                orun.submit(seq, off, first_read_no=first, threads=cores, per_read=False)
                dt += time.perf_counter() - t1
                passes += 1
            out["cpu_baseline"] = {
                "value": round(passes * ns * READ_LEN / dt / 1e9, 5), "unit": "Gbp/s", "cores": cores, "kind": "port",
                "sample": "first %d reads of the same stream%s, C restatement of the Java path (sorted array + "
                          "Blocked-Bloom gate + binary search), %d OpenMP threads, %.1f s"
                          % (ns, " x %d passes" % passes if passes > 1 else "", cores, dt)}
            del seq, off
        odb.close()
        if world == 1 and not partitioned and not striped and not args.genera:
            if "striped" in legs:
                out["striped_store"] = leg_striped(ga, db, local_rank, dseq, doff, n, nchk, otable, kern_ms)
            if "e2e" in legs:
                out["end_to_end"] = leg_end_to_end(ga, synth, torch, db, m, min(n, 4_000_000), dseq, doff)
            if "long" in legs:
                out["long_reads"] = leg_long_reads(ga, synth, orc, torch, db, gen, m, dev, cores)
            m.close()
            store.close()
            del dseq, doff
            torch.cuda.empty_cache()
            if "large" in legs or "filter" in legs or "dbbuild" in legs:
                out.update(legs_large(ga, synth, orc, torch, dev, legs, cores))
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def leg_striped(ga, db, device, dseq, doff, n, nchk, otable, plain_ms, stripes=8):
    """BASELINE.json configs[4] (a store spread over 8 GPUs) as far as ONE GPU can show it: the same store with its record
    and overflow tables in 8 stripes (gs_db_create_striped, all stripes in this GPU's HBM), the same 10 M reads through the
    striped instantiation of the fused kernel -- stripe pointer per bucket from LDS, seen bits in the run's bitmap.  What
    is not in this number is the xGMI hop of a foreign line.  Parity: the first nchk reads against the oracle table."""
    stores = ga.DeviceKMerStore.striped(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, devices=(device,) * stripes)
    ms = ga.FastqKMerMatcher(stores[stripes // 2], ga.MatchConfig(profile=True))
    info = stores[0].info

    def launch():
        ms.reset()
        ms.submit(dseq, doff, 0, n_reads=n)

    kms, _ = _kernel_ms(ms, launch, 5)
    ms.reset()
    ms.submit(dseq, doff, 0, n_reads=nchk)
    table, _ = ms.finish()
    ms.close()
    for s in stores:
        s.close()
    return {"stripes": stripes, "stripe_bytes": int(info.stripe_bytes), "kernel_ms": round(kms, 4),
            "gbps": round(n * READ_LEN / kms / 1e6, 2), "kernel_ms_plain_store": round(plain_ms, 4),
            "over_plain": round(kms / plain_ms, 4), "where": "all stripes in this GPU's HBM (no xGMI hop in this number)",
            "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(table, otable))}}


def leg_long_reads(ga, synth, orc, torch, db, gen, m, dev, cores, read_len=1000, n=1_500_000, nchk=20_000):
    """The same store and the same number of bases as configs[1], as reads of 1000 bp: more than 128 k-mer positions, so they
    take gs_match_long_kernel (one wave per read, 128 positions per iteration, contig / vote state carried from iteration to
    iteration; queued by gs_match_kernel in chunks, drawn from a shared cursor).  Whole step (reset, submit, sync) by the
    host clock, best of 3.  Parity: the first nchk reads against the oracle table."""
    dseq = torch.empty(n * read_len, dtype=torch.uint8, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=read_len)
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        m.reset()
        m.sync()
        t0 = time.perf_counter()
        m.submit(dseq, doff, 0, n_reads=n)
        m.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    m.reset()
    m.submit(dseq, doff, 0, n_reads=nchk)
    table, _ = m.finish()
    seq, off = synth.reads_host(db.genomes, nchk, read_len=read_len)
    odb = orc.DB(K, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off, threads=cores, per_read=False)
    otable, _ = orun.finish()
    odb.close()
    del dseq, doff
    return {"workload": "match: %d reads x %d bp, k=%d, the configs[1] store" % (n, read_len, K), "kernel": "gs_match_kernel (queues) + gs_match_long_kernel",
            "ms_per_step": round(best * 1e3, 3), "gbps": round(n * read_len / best / 1e9, 2),
            "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(table, otable))}}


def leg_end_to_end(ga, synth, torch, db, m, n, dseq, doff):
    """configs[1] with the reads in page-locked HOST memory (what a JVM host holds after parsing): two batches under
    way through gs_match_submit_async, the H2D copy of one under the kernel of the other ("submit..finish
    with reads resident in pinned host memory").  Never `value`."""
    half = n // 2
    seq, off = synth.reads_host(db.genomes, n, read_len=READ_LEN)
    parts = []
    for a, b in ((0, half), (half, n)):
        ps = torch.from_numpy(seq[int(off[a]):int(off[b])].copy()).pin_memory().numpy()
        po = torch.from_numpy((off[a:b + 1] - off[a]).astype(np.int64)).pin_memory().numpy().view(off.dtype)
        parts.append((ps, po, a))
    del seq

    def run(rounds):
        last = []
        for _ in range(rounds):
            for ps, po, a in parts:
                last.append(m.submit_async(ps, po, a))
                if len(last) > 2:
                    m.wait(last[-3])
        m.sync()

    m.reset()
    run(1)  # allocates the staging buffers
    m.reset()
    rounds = 5
    t0 = time.perf_counter()
    run(rounds)
    dt = (time.perf_counter() - t0) / rounds
    m.reset()
    run(1)
    host_table, _ = m.finish()
    m.reset()
    m.submit(dseq, doff, 0, n_reads=n)
    dev_table, _ = m.finish()
    return {"pinned_host_gbps": round(n * READ_LEN / dt / 1e9, 2), "reads": n, "batches_in_flight": 2,
            "ms_per_batch_pair": round(dt * 1e3, 3), "h2d_gbs_of_sequence": round(n * READ_LEN / dt / 1e9, 2),
            "api": "gs_match_submit_async / gs_match_wait (GS_MEM_HOST, page-locked arrays)",
            "table_equals_device_resident_run": bool(np.array_equal(host_table, dev_table))}


def leg_db_build(ga, orc, torch, db, gen):
    """the compute core of FillDBGoal + DBGoal on the device (gs_dbbuild): the 47 M-k-mer store of the large-store
    leg from its 500 genomes -- every genome as a fill region and again as an update region (DBGoal walks the whole
    collection), one radix sort of the (k-mer, region) pairs, LCA fold -- with the genomes already in HBM; the arrays must equal
    the store the host recipe built.  Beside it the CPU restatement (one thread = one reader thread of the reference) on the
    first genomes."""
    g = db.genomes
    n_g, g_len = g.shape
    doff = (torch.arange(n_g + 1, dtype=torch.int64, device=gen.device) * g_len)
    flat = gen.reshape(-1)
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        b = ga.DeviceDbBuilder(K, db.n_values, db.parent_vi)
        b.add(flat, doff, db.species_vi, update=False)
        b.add(flat, doff, db.species_vi, update=True)
        keys, vals = b.finish()
        dt = time.perf_counter() - t0
        b.close()
        best = dt if best is None else min(best, dt)
    same = bool(np.array_equal(keys, db.kmers) and np.array_equal(vals, db.value_idx))
    ns = min(12, n_g)
    hseq = np.ascontiguousarray(g[:ns]).reshape(-1)
    hoff = (np.arange(ns + 1) * g_len).astype(np.uint64)
    t0 = time.perf_counter()
    ob = orc.DbBuild(K, db.n_values, db.parent_vi)
    ob.fill(hseq, hoff, db.species_vi[:ns])
    ob.optimize()
    ob.update(hseq, hoff, db.species_vi[:ns])
    ob.fetch()
    ob.close()
    dt_cpu = time.perf_counter() - t0
    bases = int(n_g) * int(g_len)
    return {"workload": "FillDBGoal + DBGoal: %d genomes x %d bp, k=31 -> %d distinct k-mers with LCA values" % (n_g, g_len, len(keys)),
            "genome_bases": bases, "pairs_sorted": 2 * bases, "seconds": round(best, 4),
            "mbases_per_s": round(bases / best / 1e6, 1), "includes": "k-mer kernel, radix sort, LCA fold, compaction, D2H of the result arrays",
            "arrays_equal_host_built_store": same,
            "cpu_baseline": {"mbases_per_s": round(ns * g_len / dt_cpu / 1e6, 2), "cores": 1, "kind": "port",
                             "sample": "the first %d genomes through orc_build_* (fill, sort, update)" % ns}}


def legs_large(ga, synth, orc, torch, dev, legs, cores):
    """BASELINE.json configs[2] / configs[3] per GPU: a 47 M-k-mer / 526-value store (1 GiB table, four times the
    Infinity Cache) and the XOR index filter over its species k-mers (~47 M keys, 1.8 G bits, 27 hashes)."""
    res = {}
    n, nchk = 10_000_000, 100_000
    t0 = time.perf_counter()
    db = synth.SynthDB(k=K, genera=25, species_per_genus=20)
    t_db = time.perf_counter() - t0
    gen = torch.from_numpy(db.genomes).to(dev)
    dseq = torch.empty(n * READ_LEN, dtype=torch.uint8, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=READ_LEN)
    torch.cuda.synchronize()
    seq, off = synth.reads_host(db.genomes, nchk, read_len=READ_LEN)
    if "dbbuild" in legs:
        res["db_build"] = leg_db_build(ga, orc, torch, db, gen)
    if "large" in legs:
        t0 = time.perf_counter()
        store = ga.DeviceKMerStore(K, db.kmers, db.value_idx, db.n_values, db.parent_vi)
        t_store = time.perf_counter() - t0
        info = store.info
        m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))

        def launch():
            m.reset()
            m.submit(dseq, doff, 0, n_reads=n)

        kms, wall = _kernel_ms(m, launch, 5)
        odb = orc.DB(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, bloom_gate=True)
        orun = orc.MatchRun(odb)
        orun.submit(seq, off, threads=cores, per_read=False)
        ot, _ = orun.finish()
        odb.close()
        m.reset()
        m.submit(dseq, doff, 0, n_reads=nchk)
        gt, _ = m.finish()
        traffic, src = _pmc_traffic("large_store")
        ach = n * BYTES_PER_READ / (kms * 1e-3) / 1e9
        res["large_store"] = {
            "workload": "match: %d reads x 150 bp, k=31, %d-k-mer / %d-value store" % (n, db.n_entries, db.n_values),
            "store_kmers": int(db.n_entries), "n_values": int(db.n_values), "record_bytes": int(info.rec_bytes), "table_bytes": int(info.table_bytes),
            "mgate_bytes": int(info.mgate_bytes), "kernel": "gs_match_kernel<global counters, k=31>",
            "kernel_ms": round(kms, 3), "ms_per_step": round(wall, 3), "gbps": round(n * READ_LEN / (kms * 1e-3) / 1e9, 2),
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "measured_frac": None if traffic is None else round(traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic_note": None if traffic is None else "rocprofv3 --pmc FETCH_SIZE+WRITE_SIZE per launch, %s" % src,
            "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(ot, gt))},
            "build_s": {"synthetic_arrays": round(t_db, 1), "gs_db_create": round(t_store, 1)}}
        m.close()
        store.close()
    if "filter" in legs:
        keys = db.kmers[np.isin(db.value_idx, db.species_vi)]  # BloomIndexGoal: k-mers of the requested taxa
        bits, hashes, factors = synth.xor_bloom_geometry(len(keys), 1e-8)
        dwords = torch.zeros((bits + 63) // 64, dtype=torch.int64, device=dev)
        synth.xor_bloom_device(torch.from_numpy(keys).to(dev), len(keys), bits, torch.from_numpy(factors).to(dev), hashes, dwords)
        words = dwords.cpu().numpy().view(np.uint64)
        del dwords
        bloom = ga.DeviceBloomFilter(ga.BLOOM_XOR, bits, factors, words)
        flt = ga.FastqBloomFilter(K, bloom, 1, 0.2, profile=True)
        acc = torch.empty(n, dtype=torch.uint8, device=dev)
        kms, wall = _kernel_ms(flt, lambda: flt.submit(dseq, doff, acc, n_reads=n), 5)
        ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)  # the oracle builds its own filter from the same keys
        ob.put_many(keys, threads=cores)
        same_filter = bool(ob.bits == bits and np.array_equal(ob.words, words))
        want = ob.filter_batch(K, 1, 0.2, seq, off, threads=cores)
        got = acc[:nchk].cpu().numpy()
        traffic, src = _pmc_traffic("filter")
        ach = n * BYTES_PER_READ / (kms * 1e-3) / 1e9
        res["filter"] = {
            "workload": "filter: %d reads x 150 bp, k=31, XOR index filter of %d keys (%d bits, %d hashes), minPosCount 1" % (n, len(keys), bits, hashes),
            "filter_keys": int(len(keys)), "filter_bytes": int(len(words) * 8), "n_hashes": int(hashes),
            "kernel": "gs_filter_kernel", "kernel_ms": round(kms, 3), "ms_per_step": round(wall, 3),
            "gbps": round(n * READ_LEN / (kms * 1e-3) / 1e9, 2), "accepted_frac": round(float(acc.float().mean()), 4),
            "bound": "fabric random 64-byte line rate (every filter bit is a random line of the bit array; members need 27)",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": traffic, "lines_per_read": None if traffic is None else round(traffic / 64 / n, 1),
            "measured_frac": None if traffic is None else round(traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic_note": None if traffic is None else "rocprofv3 --pmc FETCH_SIZE+WRITE_SIZE per launch, %s" % src,
            "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(want, got)), "filter_bits_equal_oracle": same_filter}}
        bloom.close()
    return res


if __name__ == "__main__":
    main()
