#!/usr/bin/env python3
"""bench.py -- Gbp/s of the `match` hot path on synthetic 150 bp reads, k=31 (BASELINE.json config 2).

A step = one pass of the hot path over one batch: gs_match_reset + gs_match_submit (reads already resident
in HBM) + the per-taxid table merge (RCCL all-reduce / bitmap all-gather when N > 1) + gs_match_finish.
Weak scaling: every rank classifies `--reads` reads of its own slice of the global read stream against a
full replica of the store.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--cpu-seconds S]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--check-reads", type=int, default=200_000, help="reads cross-checked against the oracle")
    ap.add_argument("--mode", choices=["sharded", "partitioned"], default="sharded",
                    help="sharded: store replicated, reads sharded (configs[1]/[3], the default bench line); "
                         "partitioned: store split over the ranks by key hash, k-mers routed by all-to-all (configs[4])")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import genestrip_amd as ga
    from genestrip_amd import synth
    from genestrip_amd.distributed import merge_run_state, partitioned_finish, partitioned_match_batch

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
    db = synth.SynthDB(k=K)
    store = ga.DeviceKMerStore(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, device=local_rank,
                               n_parts=world if partitioned else 1, part=rank if partitioned else 0)
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

    # measured memory-side traffic of one launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    # (profiles/r01_match_pmc_summary.csv, KiB units; 64-byte random requests are counted exactly).  Only valid for
    # the default workload the profile was taken on.
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_match_pmc_summary.csv")
    if n == 10_000_000 and not partitioned and os.path.exists(pmc):
        vals = dict(l.strip().split(",") for l in open(pmc) if l[0] not in "#c" and "," in l)
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            traffic = int((float(vals["FETCH_SIZE"]) + float(vals["WRITE_SIZE"])) * 1024)

    out = {
        "metric": "Gbp/s classified (match goal), k=31, 150bp reads; bit-exact CSV counts",
        "value": round(gbps, 3), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "match: %d synthetic 150 bp reads per GPU, k=31, %d-k-mer / %d-taxid store resident in HBM "
                               "(BASELINE.json configs[1])" % (n, db.n_entries, len(db.species_vi)),
                   "reads_per_gpu": n, "read_len": READ_LEN, "k": K, "store_kmers": int(db.n_entries),
                   "store_table_bytes": int(info.table_bytes), "gate_bytes": int(info.gate_bytes),
                   "parallelism": ("DB-partitioned x%d, k-mers routed by all-to-all" % world) if partitioned
                   else ("read-sharded x%d, store replicated" % world)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_note": "bytes per launch through the fabric (rocprofv3 --pmc FETCH_SIZE+WRITE_SIZE, "
                                     "profiles/r01_match_pmc_summary.csv); the 64 MiB table is Infinity-Cache resident",
                     "kernel": kernel_name, "kernel_ms": round(kern_ms, 4),
                     "algorithmic_bytes_per_launch": n * BYTES_PER_READ},
    }

    # ---- parity gate.  N = 1: the first --check-reads reads against the CPU oracle (bit-exact integer table).
    # N > 1: every rank must hold the same merged table, and rank 0 re-checks its own slice against the oracle.
    nchk = min(args.check_reads, n)
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
        seq, off = synth.reads_host(db.genomes, nchk, read_len=READ_LEN, first=first)
        orun = orc.MatchRun(odb)
        t1 = time.perf_counter()
        orun.submit(seq, off, first_read_no=first, threads=cores, per_read=False)
        dt_chk = time.perf_counter() - t1
        otable, _ = orun.finish()
        if partitioned and world > 1:
            gtable = otable  # a local re-check would need the other ranks' partitions; the digest check above covers it
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
            t1 = time.perf_counter()
            orun.submit(seq, off, first_read_no=first, threads=cores, per_read=False)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {
                "value": round(ns * READ_LEN / dt / 1e9, 5), "unit": "Gbp/s", "cores": cores, "kind": "port",
                "sample": "first %d reads of the same stream, C restatement of the Java path (sorted array + "
                          "Blocked-Bloom gate + binary search), %d OpenMP threads, %.1f s" % (ns, cores, dt)}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
