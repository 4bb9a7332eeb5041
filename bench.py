#!/usr/bin/env python3
"""bench.py -- Gbp/s of the `match` hot path on synthetic 150 bp reads, k=31 (BASELINE.json configs[1]).

A step = one pass of the hot path over one batch: gs_match_reset + gs_match_submit (reads already resident
in HBM) + the per-taxid table merge (RCCL all-reduce / bitmap all-gather when N > 1) + gs_match_finish.
Weak scaling: every rank classifies `--reads` reads of its own slice of the global read stream against a
full replica of the store.  Rank 0 prints ONE JSON line: `value` is configs[1].

What the line carries besides the contract's fields (N = 1):

  roofline      resource fractions of the dominant kernel, every CEILING measured in this run (gs_calibrate: VALU / SALU
                issue at the kernel's occupancy, random-line rate at the store's footprint) and every COUNTER collected in
                this run (rocprofv3 --pmc child passes of this same script, started before the parent touches the GPU;
                committed profiles/ summaries only when rocprofv3 is unavailable, and then tagged as such);
                `frac` = the largest resource fraction, `bound` names it (or says "latency" when none reaches 0.6);
                the SURVEY 8(d) convention figure is kept as `frac_survey_convention`
  parity        the table of the TIMED steps (all reads of all ranks) against the CPU oracle, bit-exact
  cpu_baseline  the oracle ("port") on the host cores, timed on that same pass
  large_store, huge_store, filter, table_only, reads_250bp, long_reads, end_to_end, striped_store, file_pipeline, db_build
                the workloads the headline does not show, each with its own parity gate

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--legs ...] [--pmc auto|off]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself, before any GPU call, and exits non-zero when
fewer than N devices are visible.  GS_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box (ranks share GPU 0, merge
over gloo).  GS_BENCH_FORCE_MERGE=1 runs the RCCL merge path even at N = 1.
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
K = 31
BYTES_PER_READ = READ_LEN + (READ_LEN - K + 1) * 64   # sequence bytes once + one 64 B line per k-mer
HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: 8 TB/s HBM3E
XGMI_LINKS, XGMI_GBS_PER_LINK_DIR = 7, 76.8            # per GPU: 7 links x 153.6 GB/s bidirectional
ALL_LEGS = "main,occ,large,huge,filter,c2file,tableonly,e2e,striped,dbbuild,long,r250,fasta,sweep,files"
# committed rocprofv3 --pmc summaries per workload, newest first: the fallback when the in-run passes cannot be taken
PROFILE_ROUNDS = {"match": ("r03", "r02"), "large_store": ("r03", "r02"), "filter": ("r03", "r02")}
PMC_GROUPS = (
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE",
    "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum",
    "FETCH_SIZE",
    "WRITE_SIZE",
    "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE",
)
PMC_KERNELS = {"match": r"gs_match_kernel<true", "large_store": r"gs_match_kernel<false, false, 31, false, false, 0>",
               "huge_store": r"gs_match_kernel<false, false, 31, false, false, 1>", "filter": r"gs_filter_kernel",
               "cal_lines": r"cal_random_lines"}
CAL_LINES_BYTES = 1 << 30


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


def _submit_reads(m, dseq, doff, first, n, read_len=None):
    """one resident batch to the matcher: the reads are all of one length and lie back to back, so no offsets array is handed over
    (gs_match_submit_fixed) unless GS_BENCH_OFFSETS=array asks for the general call"""
    if os.environ.get("GS_BENCH_OFFSETS", "fixed") == "array":
        m.submit(dseq, doff, first, n_reads=n)
    else:
        m.submit_fixed(dseq, READ_LEN if read_len is None else read_len, n, first)


class _DevArray:
    """zero-copy view of a raw device pointer for torch.as_tensor"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (n,), "typestr": typestr, "version": 2}


# ---------------------------------------------------------------------------------------------------------------
# counters: in-run rocprofv3 --pmc passes (children of this script), committed summaries as the fallback
# ---------------------------------------------------------------------------------------------------------------
def _pmc_child():
    """what the PMC passes profile: two full-size launches each of the configs[1] match kernel, the large-store match kernel
    and the filter kernel, plus one random-line calibration launch with a known line count (calibrates FETCH_SIZE for
    64-byte random lines, MI355X_MICROARCH.md 'HBM')"""
    import torch

    import genestrip_amd as ga
    from genestrip_amd import synth
    n = int(os.environ.get("GS_BENCH_PMC_READS", "10000000"))
    dev = torch.device("cuda", 0)
    for genera in (0, 25):
        db = synth.SynthDB(k=K, genera=genera, species_per_genus=20) if genera else synth.SynthDB(k=K)
        gen = torch.from_numpy(db.genomes).to(dev)
        dseq = torch.empty(n * READ_LEN, dtype=torch.uint8, device=dev)
        doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
        synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=READ_LEN)
        torch.cuda.synchronize()
        store = ga.DeviceKMerStore(K, db.kmers, db.value_idx, db.n_values, db.parent_vi)
        m = ga.FastqKMerMatcher(store)
        for _ in range(2):
            m.reset()
            _submit_reads(m, dseq, doff, 0, n)
        m.sync()
        m.close()
        store.close()
        if genera:
            bloom, _ = _index_filter(ga, synth, torch, dev, db)
            flt = ga.FastqBloomFilter(K, bloom, 1, 0.2)
            acc = torch.empty(n, dtype=torch.uint8, device=dev)
            for _ in range(2):
                flt.submit(dseq, doff, acc, n_reads=n)
            _sync_filter(ga, bloom)
            bloom.close()
        del gen, dseq, doff
        torch.cuda.empty_cache()
    if os.environ.get("GS_BENCH_PMC_HUGE", "1") != "0":
        # the 473 M-k-mer store of configs[4] (context-keyed gate: its own instantiation of the kernel), built on the device as leg_huge does
        store, gen, g = _huge_store(ga, synth, torch, dev)[:3]
        dseq = torch.empty(n * READ_LEN, dtype=torch.uint8, device=dev)
        doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
        synth.reads_device(gen, g.shape[0], g.shape[1], n, dseq, doff, read_len=READ_LEN)
        torch.cuda.synchronize()
        m = ga.FastqKMerMatcher(store)
        for _ in range(2):
            m.reset()
            _submit_reads(m, dseq, doff, 0, n)
        m.sync()
        m.close()
        store.close()
        del gen, dseq, doff
        torch.cuda.empty_cache()
    r = ga.calibrate(ga.CAL_RANDOM_LINES, CAL_LINES_BYTES)
    print("pmc-child done; cal lines per timed launch %d" % int(r["count"]), flush=True)


def _huge_store(ga, synth, torch, dev, genera=250):
    """~473 M k-mers / 5 251 values built ON THE DEVICE from 5 000 synthetic genomes (gs_dbbuild + the device layout builder)
    -> (store, genomes on the device, genomes, db, kmers, values, seconds)"""
    t0 = time.perf_counter()
    db = synth.SynthDB(k=K, genera=genera, species_per_genus=20, build=False)
    t_gen = time.perf_counter() - t0
    g = db.genomes
    gen = torch.from_numpy(g).to(dev)
    goff = torch.arange(g.shape[0] + 1, dtype=torch.int64, device=dev) * g.shape[1]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = ga.DeviceDbBuilder(K, db.n_values, db.parent_vi)
    b.add(gen.reshape(-1), goff, db.species_vi, update=False)
    b.add(gen.reshape(-1), goff, db.species_vi, update=True)
    kmers, vals = b.finish()
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    store = b.to_store()
    t_layout = time.perf_counter() - t0
    b.close()
    return store, gen, g, db, kmers, vals, (t_gen, t_build, t_layout)


def _sync_filter(ga, bloom):
    import ctypes as C
    ga.lib().gs_filter_sync(bloom.h)


def _index_filter(ga, synth, torch, dev, db):
    """BloomIndexGoal: the XOR index filter over the k-mers of the requested taxa (all species), built on the device"""
    keys = db.kmers[np.isin(db.value_idx, db.species_vi)]
    bits, hashes, factors = synth.xor_bloom_geometry(len(keys), 1e-8)
    dwords = torch.zeros((bits + 63) // 64, dtype=torch.int64, device=dev)
    synth.xor_bloom_device(torch.from_numpy(keys).to(dev), len(keys), bits, torch.from_numpy(factors).to(dev), hashes, dwords)
    words = dwords.cpu().numpy().view(np.uint64)
    del dwords
    return ga.DeviceBloomFilter(ga.BLOOM_XOR, bits, factors, words), (keys, bits, hashes, factors, words)


def _pmc_collect(timeout_s, reads):
    """run the PMC passes; -> ({workload: {counter: value per launch}}, note).  Every pass is its own rocprofv3 process
    (counters only, no trace domain) with this script as the program behind `--`."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="gsbench_pmc_")
    env = dict(os.environ, GS_BENCH_PMC_READS=str(reads), TMPDIR=tempfile.gettempdir())
    res = {w: {} for w in PMC_KERNELS}
    t_start = time.perf_counter()
    try:
        for gi, grp in enumerate(PMC_GROUPS):
            left = timeout_s - (time.perf_counter() - t_start)
            if left < 20:
                return None, "PMC passes ran out of their time budget (%d s)" % timeout_s
            out = os.path.join(tmp, "g%d" % gi)
            cmd = [exe, "--pmc"] + grp.split() + ["--output-format", "csv", "-d", out, "-o", "run", "--",
                                                  sys.executable, os.path.abspath(__file__), "--pmc-child"]
            try:
                p = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=min(left, 240))
            except subprocess.TimeoutExpired:
                return None, "PMC pass %d timed out" % gi
            if p.returncode != 0:
                return None, "PMC pass %d failed (rc %d): %s" % (gi, p.returncode, p.stdout.decode(errors="replace")[-300:])
            rows = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                rows += list(csv.DictReader(open(f)))
            if not rows:
                return None, "PMC pass %d wrote no counter file" % gi
            for w, pat in PMC_KERNELS.items():
                mine = [r for r in rows if re.search(pat, r["Kernel_Name"])]
                if not mine:
                    continue
                dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in mine}
                longest = max(dur.values())
                acc = {}
                for r in mine:  # the full-size launches only (warm-up launches of the calibration kernel are short)
                    if dur[r["Dispatch_Id"]] >= 0.8 * longest:
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                for name, vals in acc.items():
                    res[w][name] = sum(vals) / len(vals)
        return res, "rocprofv3 --pmc, %d passes of this script in this run (%.0f s)" % (len(PMC_GROUPS), time.perf_counter() - t_start)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _pmc_committed(name):
    """counters of a committed rocprofv3 --pmc summary of the same workload: (dict, source) or (None, None)"""
    for rnd in PROFILE_ROUNDS[name]:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}_pmc_summary.csv")
        if os.path.exists(path):
            vals = {}
            for l in open(path):
                if l[0] in "#c" or "," not in l:
                    continue
                f = l.strip().split(",")
                try:
                    vals[f[0]] = float(f[1])
                except ValueError:
                    pass
            head = [l for l in open(path) if l.startswith("#")][:1]
            return vals, "%s (committed summary, NOT measured in this run; %s)" % (os.path.relpath(path, ROOT), head[0][1:].strip() if head else "")
    return None, None


def _fetch_correction(pmc):
    """FETCH_SIZE calibrated on a known byte count of this access pattern (random 64-byte lines, 16 bytes per lane):
    expected = lines x 64 B over what the counter reported for the calibration kernel of the same pass"""
    c = (pmc or {}).get("cal_lines") or {}
    if "FETCH_SIZE" not in c or "lines" not in c or c["FETCH_SIZE"] <= 0:
        return 1.0, None
    want = c["lines"] * 64.0
    got = c["FETCH_SIZE"] * 1024.0
    return want / got, {"lines": c["lines"], "expected_bytes": want, "fetch_size_bytes": got, "factor": round(want / got, 4)}


def _resources(cnt, kern_ms, n_reads, ceil, footprint_key, src):
    """resource fractions of one kernel launch: counters per launch over the kernel's duration against ceilings measured in
    this run (gs_calibrate).  -> dict with `frac` (largest), `bound` (its name, or latency), the fractions and the raw rates"""
    sec = kern_ms * 1e-3
    out = {"counters_source": src, "kernel_ms": round(kern_ms, 4)}
    fr = {}
    if cnt is None:
        return out, fr
    fetch_b = cnt.get("FETCH_SIZE", 0.0) * 1024.0 * ceil.get("fetch_correction", 1.0)
    write_b = cnt.get("WRITE_SIZE", 0.0) * 1024.0
    if "FETCH_SIZE" in cnt:
        out["traffic"] = int(fetch_b + write_b)
        out["fabric_bytes"] = {"GBs": round((fetch_b + write_b) / sec / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                               "frac": round((fetch_b + write_b) / sec / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "FETCH_SIZE + WRITE_SIZE (memory side of L2; Infinity-Cache hits included) over the 8 TB/s HBM peak"}
        fr["hbm"] = out["fabric_bytes"]["frac"]
    if "TCC_EA0_RDREQ_sum" in cnt and footprint_key in ceil:
        rq = cnt["TCC_EA0_RDREQ_sum"] + cnt.get("TCC_EA0_WRREQ_sum", 0.0)
        out["line_requests"] = {"per_read": round(rq / n_reads, 2), "G_per_s": round(rq / sec / 1e9, 2),
                                "ceiling_G_per_s": round(ceil[footprint_key] / 1e9, 2),
                                "frac": round(rq / sec / ceil[footprint_key], 4),
                                "note": "L2 -> fabric requests over the random 64-byte line rate measured in this run for a table of the store's footprint"}
        fr["line_requests"] = out["line_requests"]["frac"]
    if "SQ_INSTS_VALU" in cnt:
        v = cnt["SQ_INSTS_VALU"]
        # the hardware-rate view: against the rate of full-rate 32-bit VALU ops measured in this run (and the guide's 2 cycles per
        # wave64 op and SIMD at 2.4 GHz); the fraction against the kernel's OWN mix (quarter-rate multiplies, 64-bit shifts) reads
        # high for any kernel with such ops and is kept as information only (not a demonstrated saturation)
        guide = ceil["n_cu"] * 4 * 2.4e9 / 2.0 if ceil.get("n_cu") else None
        out["valu_issue"] = {"per_read": round(v / n_reads, 1), "G_winst_per_s": round(v / sec / 1e9, 1),
                             "ceiling_G_winst_per_s": round(ceil.get("valu_pure", ceil["valu_mix"]) / 1e9, 1),
                             "frac": round(v / sec / ceil.get("valu_pure", ceil["valu_mix"]), 4),
                             "frac_vs_guide_2_cycles_per_op": round(v / sec / guide, 4) if guide else None,
                             "frac_vs_own_instruction_mix": round(v / sec / ceil["valu_mix"], 4),
                             "own_mix_ceiling_G_winst_per_s": round(ceil["valu_mix"] / 1e9, 1),
                             "note": "wave64 VALU instructions over the measured rate of full-rate 32-bit VALU ops at 8 waves per SIMD (gs_calibrate, "
                                     "this run); frac_vs_own_instruction_mix prices the same count against a calibration kernel with the match kernel's "
                                     "mix (27 of 64 instructions of the 2.3-cycle class -- and / or / xor / add / sub / lshr / mov on VGPRs and constants --, "
                                     "37 of the 4.3-cycle class: every compare, select, three-operand, 64-bit and cross-lane instruction; per-opcode "
                                     "rates: tools/valu_rates.hip, profiles/r04_valu_rates.txt): the share of the SIMDs' cycles the vector pipes are busy"}
        fr["valu_issue"] = out["valu_issue"]["frac"]
    if "SQ_INSTS_SALU" in cnt:
        s = cnt["SQ_INSTS_SALU"] + cnt.get("SQ_INSTS_SMEM", 0.0)
        out["salu_issue"] = {"per_read": round(s / n_reads, 1), "G_inst_per_s": round(s / sec / 1e9, 1),
                             "ceiling_G_inst_per_s": round(ceil["salu"] / 1e9, 1), "frac": round(s / sec / ceil["salu"], 4),
                             "note": "scalar ALU + scalar memory instructions over the measured scalar issue rate"}
        fr["salu_issue"] = out["salu_issue"]["frac"]
    if "SQ_WAIT_ANY" in cnt and cnt.get("SQ_WAVE_CYCLES"):
        out["wait_frac"] = round(cnt["SQ_WAIT_ANY"] / cnt["SQ_WAVE_CYCLES"], 3)
    if "SQ_INSTS_VMEM_RD" in cnt:
        out["vmem_loads_per_read"] = round(cnt["SQ_INSTS_VMEM_RD"] / n_reads, 2)
    if "GRBM_GUI_ACTIVE" in cnt:
        out["clock_ghz_under_pmc"] = round(cnt["GRBM_GUI_ACTIVE"] / 8 / sec / 1e9, 2)
    if "TA_TA_BUSY_sum" in cnt and cnt.get("GRBM_GUI_ACTIVE") and ceil.get("n_cu"):
        # one texture-address unit per CU takes the wave's vector-memory instructions apart into cache-line accesses; its busy
        # cycles (summed over the CUs) over the kernel's cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        f = cnt["TA_TA_BUSY_sum"] / (ceil["n_cu"] * cnt["GRBM_GUI_ACTIVE"] / 8.0)
        out["vmem_address"] = {"frac": round(f, 4), "l1_line_accesses_per_read": round(cnt.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / n_reads, 1),
                               "note": "TA_TA_BUSY_sum over CUs x kernel cycles: the share of the time a CU's vector-memory address unit is busy "
                                       "(scattered loads cost it up to 78 cycles per wave instruction, calibration.vmem_wave_loads_per_s_per_cu)"}
        fr["vmem_address"] = out["vmem_address"]["frac"]
    return out, fr


def _name_bound(fr):
    """largest resource fraction and what to call the kernel"""
    if not fr:
        return None, None, "unknown (no counters)"
    top = max(fr, key=fr.get)
    if fr[top] >= 0.6:
        return top, fr[top], top
    return top, fr[top], "latency / mixed (no resource reaches 0.6 of a hardware-rate ceiling; largest: %s)" % top


def _calibrate(ga, footprints):
    """ceilings of this device, measured now: issue rates at 8 waves per SIMD and random-line rates per footprint"""
    cal = {}
    n_cu = None
    for key, what in (("valu_pure", ga.CAL_VALU_PURE), ("valu_mix", ga.CAL_VALU_MIX), ("salu", ga.CAL_SALU), ("valu_salu", ga.CAL_VALU_SALU),
                      ("vmem_bytes", ga.CAL_VMEM_BYTES), ("vmem_words", ga.CAL_VMEM_WORDS), ("vmem_shared_lines", ga.CAL_VMEM_SHARED_LINES),
                      ("vmem_scattered", ga.CAL_VMEM_SCATTERED)):
        r = ga.calibrate(what)
        cal[key] = r["rate"]
        n_cu = r["n_cu"]
    for key, nbytes in footprints.items():
        cal[key] = ga.calibrate(ga.CAL_RANDOM_LINES, int(max(1 << 20, nbytes)))["rate"]
    simds = n_cu * 4
    cal["n_cu"] = n_cu
    rep = {
        "n_cu": n_cu,
        "valu_pure_cycles_per_wave64_inst_at_2.4GHz": round(2.4e9 * simds / cal["valu_pure"], 3),
        "valu_mix_cycles_per_wave64_inst_at_2.4GHz": round(2.4e9 * simds / cal["valu_mix"], 3),
        "salu_cycles_per_inst_per_cu_at_2.4GHz": round(2.4e9 * n_cu / cal["salu"], 3),
        "valu_salu_alternating_cycles_per_inst_per_simd": round(2.4e9 * simds / cal["valu_salu"], 3),
        "vmem_wave_loads_per_s_per_cu": {k[5:]: round(cal[k] / n_cu / 1e6, 1) for k in cal if k.startswith("vmem_")},
        "vmem_unit": "M wave-level load instructions per second and CU (cache resident)",
        "random_lines_G_per_s": {k: round(v / 1e9, 2) for k, v in cal.items() if k.startswith("lines_")},
        "note": "gs_calibrate kernels at 8 waves per SIMD, this run; MI355X_MICROARCH.md lines 54 / 473 give 2 cycles per wave64 VALU "
                "instruction on the SIMD-32 once a SIMD holds two or more waves (4 for one wave alone)",
    }
    return cal, rep


# ---------------------------------------------------------------------------------------------------------------
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _visible_gpus():
    """GPUs this process's children would see, counted WITHOUT loading or initialising the HIP runtime: the KFD topology nodes that
    have SIMDs and whose render node exists in /dev/dri, cut down by the *_VISIBLE_DEVICES lists.  -1: cannot tell (the ranks decide).
    (torch.cuda.device_count() only stays clear of hipInit when amdsmi initialises -- here it does not, error 34 -- and importing
    torch maps libamdhip64 into the process that is about to fork + exec the launcher.)"""
    topo = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.exists("/dev/kfd"):
        return 0
    try:
        nodes = sorted(os.listdir(topo))
    except OSError:
        return -1
    n = 0
    for node in nodes:
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(topo, node, "properties")) if len(line.split()) >= 2)
        except OSError:
            return -1
        if int(props.get("simd_count", "0")) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1"))
        if minor >= 0 and not os.path.exists("/dev/dri/renderD%d" % minor):
            continue  # (not handed to this container)
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _self_launch(args):
    """--gpus N without a launcher: start the N ranks (fresh processes).  This process never imports torch and never touches the HIP
    runtime: a fork + exec out of a process that has initialised the GPU is what must not happen (GS_BENCH_LAUNCH_HOOK=<file>: the
    parent records what it sees and whether libamdhip64 is mapped, for tests/test_distributed_cpu.py)."""
    rehearsal = os.environ.get("GS_BENCH_BACKEND", "nccl") == "gloo"
    ndev = _visible_gpus()
    hook = os.environ.get("GS_BENCH_LAUNCH_HOOK")
    if hook:
        with open("/proc/self/maps") as f:
            maps = f.read()
        with open(hook, "w") as f:
            json.dump({"visible": ndev, "hip_mapped": "libamdhip64" in maps, "torch_imported": "torch" in sys.modules}, f)
    if 0 <= ndev < args.gpus and not rehearsal:
        sys.stderr.write("bench.py --gpus %d: only %d device(s) visible -- refusing to print a line for fewer GPUs than asked for "
                         "(GS_BENCH_BACKEND=gloo rehearses the multi-rank flow on one GPU)\n" % (args.gpus, ndev))
        sys.exit(3)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg beyond the parity pass (0 = one pass)")
    ap.add_argument("--check-reads", type=int, default=0, help="reads per rank cross-checked against the oracle (0 = all timed reads)")
    ap.add_argument("--mode", choices=["sharded", "striped", "partitioned"], default="sharded",
                    help="sharded: store replicated, reads sharded (configs[1]/[3], the default bench line); "
                         "striped: ONE store, its record table split over the GPUs' HBM, foreign record lines loaded over "
                         "xGMI by the same fused kernel, reads sharded (configs[4]); "
                         "partitioned: round 1's split pipeline, store split by key hash, k-mers routed by all-to-all")
    ap.add_argument("--stripes", type=int, default=8,
                    help="--mode striped with ONE rank: stripes of the record table, all in this GPU's HBM (prices the "
                         "stripe arithmetic of the kernel; with N ranks there is one stripe per rank)")
    ap.add_argument("--genera", type=int, default=0, help="size of the synthetic store: genera of 20 species (0: configs[1]'s store)")
    ap.add_argument("--legs", default=ALL_LEGS, help="comma list of the extra N = 1 objects (main always runs)")
    ap.add_argument("--pmc", choices=["auto", "off"], default="auto",
                    help="auto: collect the roofline's counters in this run (rocprofv3 --pmc child passes); off: committed summaries")
    ap.add_argument("--pmc-seconds", type=int, default=420, help="time budget of the in-run PMC passes")
    ap.add_argument("--pmc-save", default="", help="directory for <workload>_pmc_summary.csv files of the in-run PMC passes (the format of profiles/rNN_*_pmc_summary.csv)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return _pmc_child()
    legs = set(args.legs.split(","))

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    partitioned = args.mode == "partitioned"
    striped = args.mode == "striped"
    default_workload = args.reads == 10_000_000 and not partitioned and not striped and not args.genera

    # ---- counters first: the PMC passes are child processes and must start before this process initialises the GPU
    pmc, pmc_note = None, "off (--pmc off)"
    if world == 1 and args.pmc == "auto" and default_workload:
        import torch  # noqa: F401  (pages the libraries in before the children time out on a cold box)
        pmc, pmc_note = _pmc_collect(args.pmc_seconds, args.reads)
        if pmc is not None and args.pmc_save:
            os.makedirs(args.pmc_save, exist_ok=True)
            for w, vals in pmc.items():
                if vals:
                    with open(os.path.join(args.pmc_save, w + "_pmc_summary.csv"), "w") as f:
                        f.write("# PMC counters of the kernel matching /%s/ per launch, mean over the full-size launches of `bench.py --pmc-child` (%d reads x 150 bp, k=31):\n"
                                "# %s; groups: %s\n" % (PMC_KERNELS[w], args.reads, pmc_note, " | ".join(PMC_GROUPS)))
                        f.write("counter,value_per_launch\n")
                        for name in sorted(vals):
                            f.write("%s,%.6g\n" % (name, vals[name]))

    import torch
    import torch.distributed as dist

    import genestrip_amd as ga
    from genestrip_amd import synth
    from genestrip_amd.distributed import merge_run_state, partitioned_finish, partitioned_match_batch, striped_store

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (genestrip_amd has no CPU fallback)")
    # GS_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box: several ranks share GPU 0 and the merge runs over
    # gloo on host copies, so the multi-rank control flow (barriers, collectives, rank-0-only legs) can be exercised
    # without a multi-GPU node.  Its numbers mean nothing; the real run uses RCCL ("nccl"), one GPU per rank.
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    rehearsal = backend == "gloo"
    if rehearsal:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        sys.stderr.write("bench.py: %d ranks but only %d device(s) visible\n" % (world, torch.cuda.device_count()))
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev  # where the small control tensors of the collectives live
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
        _submit_reads(m, dseq, doff, first, n_reads)
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
    conv = n * BYTES_PER_READ / (kern_ms * 1e-3) / 1e9  # GB/s by the SURVEY 8d convention

    out = {
        "metric": "Gbp/s classified (match goal), k=31, 150bp reads; bit-exact CSV counts",
        "value": round(gbps, 3), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": "match: %d synthetic 150 bp reads per GPU, k=31, %d-k-mer / %d-taxid store resident in HBM "
                               "(BASELINE.json %s)" % (n, db.n_entries, len(db.species_vi),
                                                       "configs[4]: a store spread over the GPUs' HBM; link-bound estimate in striped_store.xgmi_budget"
                                                       if striped else "configs[1]"),
                   "reads_per_gpu": n, "read_len": READ_LEN, "k": K, "store_kmers": int(db.n_entries),
                   "store_record_bytes": int(info.rec_bytes), "store_table_bytes": int(info.table_bytes),
                   "gate_bytes": int(info.mgate_bytes or info.gate_bytes),
                   "parallelism": ("DB-partitioned x%d, k-mers routed by all-to-all" % world) if partitioned
                   else ("reads sharded x%d, ONE store with its record table in %d stripes (%s)"
                         % (world, info.n_stripes, "one per GPU, foreign lines over xGMI" if world > 1 else "all in this GPU's HBM")) if striped
                   else ("read-sharded x%d, store replicated" % world)},
    }

    # ---- ceilings measured now, on this device (rank 0 of a one-GPU run: the roofline is a single-GPU statement)
    cal, cal_rep = (None, None)
    foot_main = int(info.rec_bytes + info.table_bytes + (info.mgate_bytes or info.gate_bytes))
    if world == 1:
        cal, cal_rep = _calibrate(ga, {"lines_main": foot_main, "lines_1GiB": 1 << 30, "lines_filter": 226 << 20,
                                       "lines_cal": CAL_LINES_BYTES})
        if pmc and pmc.get("cal_lines"):
            pmc["cal_lines"]["lines"] = ga.calibrate(ga.CAL_RANDOM_LINES, CAL_LINES_BYTES)["count"]
        corr, corr_rep = _fetch_correction(pmc)
        cal["fetch_correction"] = corr
        cal_rep["fetch_size_calibration"] = corr_rep or "not taken (no in-run PMC pass): FETCH_SIZE used as reported (64-byte requests)"
    cnt, src = (None, None)
    if world == 1 and default_workload:
        if pmc and pmc.get("match"):
            cnt, src = pmc["match"], pmc_note
        else:
            cnt, src = _pmc_committed("match")
            if src:
                src += "; in-run passes: " + pmc_note
    roof = {"kernel": kernel_name, "kernel_ms": round(kern_ms, 4),
            "algorithmic_bytes_per_launch": n * BYTES_PER_READ,
            "frac_survey_convention": round(conv / HBM_PEAK_GBS, 4),
            "survey_convention_note": "SURVEY 8(d): one 64-byte line per k-mer position + the sequence bytes, over the 8 TB/s HBM peak; above 1 "
                                      "because the kernel no longer requests those lines (minimizer gate, super-k-mer records) -- kept for "
                                      "continuity, NOT a resource fraction"}
    if cal is not None:
        res, fr = _resources(cnt, kern_ms, n, cal, "lines_main", src)
        top, frac, bound = _name_bound(fr)
        roof.update(res)
        roof["resource_fracs"] = fr
        roof["bound"] = "hbm" if top == "hbm" and frac >= 0.6 else bound
        roof["frac"] = frac
        if top == "hbm" or top is None:
            roof.update({"achieved": res.get("fabric_bytes", {}).get("GBs"), "peak": HBM_PEAK_GBS, "unit": "GB/s"})
        elif top == "line_requests":
            roof.update({"achieved": res["line_requests"]["G_per_s"], "peak": res["line_requests"]["ceiling_G_per_s"], "unit": "G lines/s"})
        elif top == "valu_issue":
            roof.update({"achieved": res["valu_issue"]["G_winst_per_s"], "peak": res["valu_issue"]["ceiling_G_winst_per_s"], "unit": "G wave-instructions/s"})
        elif top == "vmem_address":
            roof.update({"achieved": res["vmem_address"]["frac"], "peak": 1.0, "unit": "busy fraction of the CUs' vector-memory address units"})
        else:
            roof.update({"achieved": res["salu_issue"]["G_inst_per_s"], "peak": res["salu_issue"]["ceiling_G_inst_per_s"], "unit": "G instructions/s"})
        roof.setdefault("traffic", None)
        roof["calibration"] = cal_rep
    else:
        roof.update({"bound": "not priced at N > 1 (the roofline is the N = 1 line's)", "achieved": None, "peak": None, "unit": None,
                     "frac": None, "traffic": None})
    out["roofline"] = roof

    # ---- parity gate: the table of the TIMED steps against the CPU oracle over the same reads.
    # N > 1: every rank must hold the same merged table and rank 0 checks it against the oracle over ALL ranks' slices
    # (partitioned mode: a smaller re-run through the collective pipeline, the reads of every rank's slice head).
    nchk = n if args.check_reads <= 0 else min(args.check_reads, n)
    full = nchk == n
    ptable = None
    if (partitioned and world > 1) or (use_dist and world > 1 and not full):
        # a shorter pass of every rank through the same collective path, merged: what rank 0 compares
        m.reset()
        if partitioned:
            partitioned_match_batch(m, K, dseq, doff, nchk, first)
            ptable, _ = partitioned_finish(m, t_sums, t_max, t_dsum)
        else:
            ptable, _ = step(nchk)
    if use_dist and world > 1:
        digest = torch.tensor([int(np.asarray(table, dtype=np.int64).sum() % (1 << 62))], dtype=torch.int64, device=cdev)
        lo, hi = digest.clone(), digest.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out["parity"] = {"merged_table_identical_on_all_ranks": bool(lo.item() == hi.item())}
        # what the collective library itself saw: one contribution per rank summed by an all-reduce on the data-path backend, and the
        # distinct devices behind the ranks (a SCALE record with N copies of one GPU would show here)
        ones = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        devs = [None] * world
        prop = torch.cuda.get_device_properties(local_rank)
        dist.all_gather_object(devs, "%s:%s" % (socket.gethostname(), getattr(prop, "uuid", None) or getattr(prop, "pci_bus_id", local_rank)))
        out["rccl_ranks_seen"] = {"backend": backend, "torch_world_size": dist.get_world_size(), "all_reduce_sum_of_ones": int(ones.item()),
                                  "distinct_devices": len(set(devs))}
    if rank == 0:
        from oracle import gs_oracle as orc
        cores = _usable_cores()
        odb = orc.DB(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, bloom_gate=True)
        orun = orc.MatchRun(odb)
        dt_chk = 0.0
        for r in range(world):  # the slices of all ranks, one after the other (host memory: one slice at a time)
            seq, off = synth.reads_host(db.genomes, nchk, read_len=READ_LEN, first=r * n)
            t1 = time.perf_counter()
            orun.submit(seq, off, first_read_no=r * n, threads=cores, per_read=False)
            dt_chk += time.perf_counter() - t1
            if world > 1:
                del seq, off
        otable, _ = orun.finish()
        if ptable is not None:
            gtable = ptable
        elif full:
            gtable = table  # the table the last timed step returned
        else:
            gtable, _ = step(nchk)
        par = out.setdefault("parity", {})
        par["reads_checked"] = nchk * world
        par["reads_timed_per_step"] = n * world
        par["what"] = ("the table of the last timed step" if full and ptable is None else "a shorter pass through the same path") + \
                      " against the CPU oracle over the same reads (integer columns bit-exact)"
        par["bit_exact"] = bool(np.array_equal(otable, gtable))
        if not par["bit_exact"] or par.get("merged_table_identical_on_all_ranks") is False:
            out["value"] = None  # a throughput without parity does not count
        if world == 1:
            # the CPU baseline is the parity pass itself (timed), repeated while the budget lasts
            dt, passes = dt_chk, 1
            while dt < 0.6 * args.cpu_seconds and passes < 8:
                orun2 = orc.MatchRun(odb)
                t1 = time.perf_counter()
                orun2.submit(seq, off, first_read_no=0, threads=cores, per_read=False)
                dt += time.perf_counter() - t1
                passes += 1
                del orun2
            out["cpu_baseline"] = {
                "value": round(passes * nchk * READ_LEN / dt / 1e9, 5), "unit": "Gbp/s", "cores": cores, "kind": "port",
                "sample": "%s %d reads of the same stream%s, C restatement of the Java path (sorted array + "
                          "Blocked-Bloom gate + binary search), %d OpenMP threads, %.1f s"
                          % ("all" if full else "the first", nchk, " x %d passes" % passes if passes > 1 else "", cores, dt)}
            del seq, off
        odb.close()
        if world == 1 and default_workload:
            extra = {}
            lps = roof.get("line_requests", {}).get("per_read")
            if "occ" in legs:  # (one unique-counting run per store: the headline matcher makes room and comes back)
                m.close()
                out["roofline"]["occupancy"] = leg_occupancy(ga, store, dseq, doff, n)
                m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
            if "striped" in legs:
                extra["striped_store"] = leg_striped(ga, db, local_rank, dseq, doff, n, otable, kern_ms, lps)
            if "e2e" in legs:
                extra["end_to_end"] = leg_end_to_end(ga, synth, torch, db, m, min(n, 4_000_000), dseq, doff)
            if "long" in legs:
                extra["long_reads"] = leg_reads_of(ga, synth, orc, torch, db, gen, m, dev, cores, 1000, 1_500_000, 200_000,
                                                   "gs_match_long_kernel (the batch is its queue: reads of one length)")
            if "r250" in legs:
                extra["reads_250bp"] = leg_reads_of(ga, synth, orc, torch, db, gen, m, dev, cores, 250, 6_000_000, 1_000_000,
                                                    "gs_match_wide_kernel<4> (220 k-mer positions per read, one trip of four sub-rounds)")
            if "sweep" in legs:
                extra["read_len_sweep"] = leg_read_len_sweep(synth, torch, db, gen, m, dev)
            if "fasta" in legs:
                extra["fasta_records"] = leg_fasta_records(ga, orc, torch, db, m, dev, cores)
            if "files" in legs:  # (last: it closes the matcher)
                extra["file_pipeline"] = leg_files(ga, synth, orc, torch, dev, db, store, m, cores)
            m.close()
            store.close()
            del dseq, doff
            torch.cuda.empty_cache()
            if "tableonly" in legs:
                extra["table_only"] = leg_table_only(ga, synth, orc, torch, dev, cores)
            if legs & {"large", "filter", "dbbuild"}:
                extra.update(legs_large(ga, synth, orc, torch, dev, legs, cores, cal, pmc, pmc_note))
            if "huge" in legs:
                extra["huge_store"] = leg_huge(ga, synth, orc, torch, dev, cores, cal, pmc, pmc_note)
            out.update(extra)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def leg_occupancy(ga, store, dseq, doff, n):
    """kernel time of the headline launch at 8 and at 7 resident workgroups per CU (= waves per SIMD; 8 is the hardware's limit at 64
    VGPRs): how much the last wave is still worth says how far the kernel is from being bound by anything but latency"""
    res = {}
    for b in (8, 7):
        os.environ["GS_MATCH_BLOCKS_PER_CU"] = str(b)
        try:
            mo = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
        finally:
            del os.environ["GS_MATCH_BLOCKS_PER_CU"]

        def launch():
            mo.reset()
            _submit_reads(mo, dseq, doff, 0, n)

        kms, _ = _kernel_ms(mo, launch, 5)
        mo.close()
        res["kernel_ms_at_%d_blocks_per_cu" % b] = round(kms, 4)
    res["gain_from_the_8th_wave_per_simd"] = round(res["kernel_ms_at_7_blocks_per_cu"] / res["kernel_ms_at_8_blocks_per_cu"] - 1.0, 4)
    return res


def leg_striped(ga, db, device, dseq, doff, n, otable, plain_ms, lines_per_read, stripes=8):
    """BASELINE.json configs[4] (a store spread over 8 GPUs) as far as ONE GPU can show it: the same store with its record
    and overflow tables in 8 stripes (gs_db_create_striped, all stripes in this GPU's HBM), the same 10 M reads through the
    striped instantiation of the fused kernel -- stripe pointer per bucket from LDS, seen bits in the run's bitmap.  What
    is not in this number is the xGMI hop of a foreign line: `xgmi_budget` prices it.  Parity: all n reads."""
    stores = ga.DeviceKMerStore.striped(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, devices=(device,) * stripes)
    ms = ga.FastqKMerMatcher(stores[stripes // 2], ga.MatchConfig(profile=True))
    info = stores[0].info

    def launch():
        ms.reset()
        _submit_reads(ms, dseq, doff, 0, n)

    kms, _ = _kernel_ms(ms, launch, 5)
    ms.reset()
    _submit_reads(ms, dseq, doff, 0, n)
    table, _ = ms.finish()
    ms.close()
    for s in stores:
        s.close()
    res = {"stripes": stripes, "stripe_bytes": int(info.stripe_bytes), "kernel_ms": round(kms, 4),
           "gbps": round(n * READ_LEN / kms / 1e6, 2), "kernel_ms_plain_store": round(plain_ms, 4),
           "over_plain": round(kms / plain_ms, 4), "where": "all stripes in this GPU's HBM (no xGMI hop in this number)",
           "parity": {"reads_checked": n, "bit_exact": bool(np.array_equal(table, otable))}}
    if lines_per_read:
        # record / table lines per read = fabric requests minus the read's own sequence bytes (150 B = 2.34 lines), (N-1)/N of them foreign
        store_lines = max(0.0, lines_per_read - READ_LEN / 64.0)
        foreign = store_lines * 64.0 * (stripes - 1) / stripes
        link = XGMI_LINKS * XGMI_GBS_PER_LINK_DIR * 1e9
        reads_s = n / (kms * 1e-3)
        res["xgmi_budget"] = {
            "foreign_bytes_per_read": round(foreign, 1), "reads_per_s_kernel": round(reads_s / 1e9, 3),
            "inbound_GBs_needed_at_kernel_rate": round(foreign * reads_s / 1e9, 1),
            "inbound_GBs_available": round(link / 1e9, 1), "links": "%d x %.1f GB/s per direction" % (XGMI_LINKS, XGMI_GBS_PER_LINK_DIR),
            "link_bound_gbps_per_gpu": round(min(reads_s, link / max(foreign, 1e-9)) * READ_LEN / 1e9, 1),
            "basis": "fabric read requests per read of the PLAIN kernel on this store (PMC, this run) minus the read's own bases; the "
                     "striped kernel skips the second record line where the gate word's hint bit is clear (about a third fewer "
                     "record lines), which this estimate does not credit",
            "verdict": "LINK-BOUND ESTIMATE: at N = 8 the striped mode is limited by inbound xGMI before small-packet overhead; a "
                       "store that fits one GPU's HBM (a 473 M-k-mer store takes 9 GiB of 288 GB) is REPLICATED instead (the default "
                       "--mode sharded), striping is for stores beyond one GPU"}
    return res


def leg_reads_of(ga, synth, orc, torch, db, gen, m, dev, cores, read_len, n, nchk, kernel):
    """The configs[1] store and about the same number of bases as reads of another length (250 bp: 220 k-mer positions, the
    paired-end length; 1000 bp: the long-read kernel).  Whole step (reset, submit, sync) by the host clock, best of 4.
    Parity: the first nchk reads against the oracle table."""
    dseq = torch.empty(n * read_len, dtype=torch.uint8, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=read_len)
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        m.reset()
        m.sync()
        t0 = time.perf_counter()
        _submit_reads(m, dseq, doff, 0, n, read_len)
        m.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    m.reset()
    _submit_reads(m, dseq, doff, 0, nchk, read_len)
    table, _ = m.finish()
    seq, off = synth.reads_host(db.genomes, nchk, read_len=read_len)
    odb = orc.DB(K, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off, threads=cores, per_read=False)
    otable, _ = orun.finish()
    odb.close()
    del dseq, doff
    return {"workload": "match: %d reads x %d bp, k=%d, the configs[1] store" % (n, read_len, K), "kernel": kernel,
            "ms_per_step": round(best * 1e3, 3), "gbps": round(n * read_len / best / 1e9, 2),
            "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(table, otable))}}


def leg_read_len_sweep(synth, torch, db, gen, m, dev, lengths=(100, 125, 150, 158, 159, 190, 222, 250, 286, 287, 300), bases=1_200_000_000):
    """Gbp/s by read length on the configs[1] store at a constant number of bases: up to 128 k-mer positions (158 bp) a read is one
    trip of gs_match_kernel (two sub-rounds of 64 positions), up to 192 / 256 positions (222 / 286 bp) one trip of
    gs_match_wide_kernel (three / four sub-rounds), above that one iteration of gs_match_long_kernel per 128 positions -- a trip costs
    about the same however few positions of its last sub-round are live, hence the steps at 159, 223 and 287 bp.  Whole step by the host
    clock, best of 3, batches of one length without an offsets array (gs_match_submit_fixed) and, for three lengths, with one (the
    queueing pass of gs_match_kernel in front of the other kernels)."""
    rows = {}
    for L in lengths:
        n = bases // L
        dseq = torch.empty(n * L, dtype=torch.uint8, device=dev)
        doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
        synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=L)
        torch.cuda.synchronize()
        row = {}
        for how in ("fixed", "offsets") if L in (150, 159, 250) else ("fixed",):
            best = None
            for _ in range(3):
                m.reset()
                m.sync()
                t0 = time.perf_counter()
                if how == "fixed":
                    m.submit_fixed(dseq, L, n, 0)
                else:
                    m.submit(dseq, doff, 0, n_reads=n)
                m.sync()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            row[how] = round(n * L / best / 1e9, 1)
        rows[str(L)] = row["fixed"] if len(row) == 1 else row
        del dseq, doff
    m.reset()
    fixed = [v if not isinstance(v, dict) else v["fixed"] for v in rows.values()]
    steps = [abs(a - b) / max(a, b) for a, b in zip(fixed[:-1], fixed[1:])]
    return {"workload": "match: %.1f Gbases per length, k=%d, the configs[1] store" % (bases / 1e9, K), "gbps_by_read_length": rows,
            "min_gbps": min(fixed), "max_gbps": max(fixed), "largest_step_between_adjacent_lengths": round(max(steps), 3)}


def leg_fasta_records(ga, orc, torch, db, m, dev, cores):
    """Records of an assembly or a chromosome FASTA against the configs[1] store: one record of 5 Mbp, and 64 of 1 Mbp.  A record of
    32 768 k-mer positions and more is cut into chunks over many waves (gs_match_huge_kernel + gs_match_huge_finish_kernel); on one
    wave (the long-read path) the 5 Mbp record takes 150 ms.  Whole step (reset, submit, sync) by the host clock, best of 5.  Parity:
    table, class and flags of every record against the oracle."""
    rng = np.random.default_rng(5)
    g0 = db.genomes
    out = {"kernel": "gs_match_kernel (hands over) + gs_match_huge_kernel + gs_match_huge_finish_kernel"}
    for name, n, L in (("one_5mbp_record", 1, 5_000_000), ("records_64x1mbp", 64, 1_000_000)):
        recs = []
        for _ in range(n):  # pieces of 1 .. 60 kbp of the genomes, now and then an N
            parts, left = [], L
            while left > 0:
                sp = int(rng.integers(0, g0.shape[0]))
                c = min(left, int(rng.integers(1000, 60000)))
                p0 = int(rng.integers(0, g0.shape[1] - c))
                parts.append(g0[sp][p0:p0 + c])
                left -= c
            r = np.concatenate(parts)
            r[rng.integers(0, L, L // 50000)] = ord("N")
            recs.append(r)
        seq = np.concatenate(recs)
        off = np.arange(n + 1, dtype=np.int64) * L
        dseq, doff = torch.from_numpy(seq).to(dev), torch.from_numpy(off).to(dev)
        best = None
        for _ in range(5):
            m.reset()
            m.sync()
            t0 = time.perf_counter()
            m.submit(dseq, doff, 0, n_reads=n)
            m.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        m.reset()
        cv, fl = m.match_reads(seq, off, 0)
        table, _ = m.finish()
        odb = orc.DB(K, db.kmers, db.value_idx, db.n_values, db.parent_vi)
        orun = orc.MatchRun(odb)
        ocv, ofl = orun.submit(seq, off, threads=cores)
        otable, _ = orun.finish()
        # Kraken-style segments of the first record (two passes over its pieces of 32 iterations)
        t0 = time.perf_counter()
        seg_off, codes, _, lens = m.segments(seq[:L], off[:2])
        seg_ms = (time.perf_counter() - t0) * 1e3
        want = odb.segments(seq[:L], cap=1 << 22)
        seg_ok = list(zip(codes.tolist(), lens.tolist())) == want
        odb.close()
        out[name] = {"workload": "match: %d record(s) x %d bp, k=%d, the configs[1] store" % (n, L, K), "ms_per_step": round(best * 1e3, 3),
                     "gbps": round(n * L / best / 1e9, 2), "segments_of_one_record_ms_from_host_memory": round(seg_ms, 3),
                     "parity": {"records_checked": n, "bit_exact": bool(np.array_equal(table, otable) and np.array_equal(cv, ocv) and np.array_equal(fl, ofl)),
                                "segments_checked": len(want), "segments_equal": bool(seg_ok)}}
        del dseq, doff
    m.reset()
    return out


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
    _submit_reads(m, dseq, doff, 0, n)
    dev_table, _ = m.finish()
    return {"pinned_host_gbps": round(n * READ_LEN / dt / 1e9, 2), "reads": n, "batches_in_flight": 2,
            "ms_per_batch_pair": round(dt * 1e3, 3), "h2d_gbs_of_sequence": round(n * READ_LEN / dt / 1e9, 2),
            "api": "gs_match_submit_async / gs_match_wait (GS_MEM_HOST, page-locked arrays)",
            "table_equals_device_resident_run": bool(np.array_equal(host_table, dev_table))}


# ---------------------------------------------------------------------------------------------------------------
# file pipeline: FASTQ files written in this run -> gs_host_match_files / gs_host_filter_files
# ---------------------------------------------------------------------------------------------------------------
def _fastq_text(seq, n, read_len=READ_LEN):
    """four-line FASTQ of n fixed-length reads as one uint8 array (fixed-width descriptors '@r0001234')"""
    width = 2 + 8 + 1
    rec = width + read_len + 3 + read_len + 1
    a = np.empty((n, rec), dtype=np.uint8)
    a[:, 0] = ord("@")
    a[:, 1] = ord("r")
    idx = np.arange(n, dtype=np.int64)
    for d in range(8):
        a[:, 2 + 7 - d] = (idx % 10 + 48).astype(np.uint8)
        idx //= 10
    a[:, width - 1] = 10
    a[:, width:width + read_len] = seq.reshape(n, read_len)
    a[:, width + read_len:width + read_len + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    a[:, width + read_len + 3:rec - 1] = ord("I")
    a[:, rec - 1] = 10
    return a.reshape(-1)


def _deflate_piece(job):
    import zlib
    path, start, size, last, bgzf = job
    with open(path, "rb") as f:
        f.seek(start)
        c = f.read(size)
    if bgzf:  # members of <= 64 KiB that state their size, as bgzip writes them
        import struct
        parts = []
        for a in range(0, len(c), 65280):
            blk = c[a:a + 65280]
            z = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = z.compress(blk) + z.flush()
            parts.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body +
                         struct.pack("<II", zlib.crc32(blk), len(blk)))
        return b"".join(parts), 0, len(c)
    z = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = z.compress(c) + (z.flush(zlib.Z_FINISH) if last else z.flush(zlib.Z_FULL_FLUSH))
    return body, zlib.crc32(c), len(c)


def _write_gz(plain, out, bgzf, procs):
    """compress `plain` on a thread pool: ONE gzip member made of independently deflated 8 MiB pieces joined by full flushes
    (what pigz writes), or BGZF blocks"""
    from concurrent.futures import ThreadPoolExecutor
    import struct
    size = os.path.getsize(plain)
    piece = 8 << 20
    jobs = [(plain, a, min(piece, size - a), a + piece >= size, bgzf) for a in range(0, size, piece)]
    # THREADS, not processes: this runs after the GPU has been initialised, and a process that holds the HIP runtime must not fork
    # (round 3's profiled runs showed the forked workers die with SIGSEGV inside the profiler's inherited signal handler when the
    # pool was torn down).  zlib releases the interpreter lock while it compresses, so the threads run side by side.
    with ThreadPoolExecutor(max_workers=max(1, procs)) as pool:
        parts = list(pool.map(_deflate_piece, jobs))
    with open(out, "wb") as f:
        if bgzf:
            for body, _, _ in parts:
                f.write(body)
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")  # the empty end-of-file block
        else:
            f.write(b"\x1f\x8b\x08\0\0\0\0\0\0\xff")
            crc = 0
            for body, c, ln in parts:
                f.write(body)
                crc = _crc32_combine(crc, c, ln)
            f.write(struct.pack("<II", crc & 0xffffffff, size & 0xffffffff))


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (GF(2) matrix form), in Python"""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[i]) for i in range(32)]

    if len2 <= 0:
        return crc1
    odd = [0xedb88320] + [1 << i for i in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def leg_files(ga, synth, orc, torch, dev, db, store, m, cores, n=4_000_000):
    """SURVEY 8(f2) as the driver sees it: a four-line FASTQ of n reads written in this run (plain, .gz as pigz writes it, BGZF)
    -> gs_host_match_files (raw text blocks to the device, records found there; own parallel inflate) -> table, which must equal
    the table of the same reads resident in HBM; and the `filter` goal over the plain file with the accepted reads written
    (BASELINE.json configs[2]'s filtered-FASTQ writeback) against the resident filter run.  Files in the page cache."""
    from genestrip_amd import host
    seq, off = synth.reads_host(db.genomes, n, read_len=READ_LEN)
    tmp = tempfile.mkdtemp(prefix="gsbench_files_")
    res = {"reads": n, "tmp": tempfile.gettempdir()}
    try:
        plain = os.path.join(tmp, "reads.fastq")
        t0 = time.perf_counter()
        _fastq_text(seq, n).tofile(plain)
        gz, bz = plain + ".gz", os.path.join(tmp, "reads.bgzf.fastq.gz")
        _write_gz(plain, gz, False, cores)
        _write_gz(plain, bz, True, cores)
        res["inputs_written_s"] = round(time.perf_counter() - t0, 1)
        m.reset()
        m.submit(seq, off, 0)
        want, _ = m.finish()
        # a .gz of several batches at their natural size (the decoder takes up to 320 MiB of compressed data per batch): 12 M reads
        n_big = 12_000_000
        seq_b, off_b = synth.reads_host(db.genomes, n_big, read_len=READ_LEN)
        m.reset()
        half = n_big // 2  # (two submits: the kernel trace's largest match launches stay the timed steps of the bench line)
        cut = int(off_b[half])
        m.submit(seq_b[:cut], off_b[:half + 1], 0)
        m.submit(seq_b[cut:], off_b[half:] - off_b[half], half)
        want_b, _ = m.finish()
        m.close()  # (one unique-counting run per store: gs_host_match_files begins its own)
        t0 = time.perf_counter()
        plain_b = os.path.join(tmp, "reads_big.fastq")
        _fastq_text(seq_b, n_big).tofile(plain_b)
        del seq_b, off_b
        gz_b = plain_b + ".gz"
        _write_gz(plain_b, gz_b, False, cores)
        os.remove(plain_b)
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            table, _, tot = host.match_files(store, [gz_b])[:3]
            dt = time.perf_counter() - t1
            best = dt if best is None else min(best, dt)
        res["gz_large"] = {"reads": n_big, "file_bytes": os.path.getsize(gz_b), "seconds": round(best, 3), "gbps": round(n_big * READ_LEN / best / 1e9, 2),
                           "table_equals_resident_run": bool(np.array_equal(table, want_b)), "reads_seen": int(tot.reads),
                           "input_written_s": round(time.perf_counter() - t0 - 3 * best, 1),
                           "note": "three batches; the compressed bytes are uploaded whole by the decoder's own thread while the batches are decoded"}
        os.remove(gz_b)
        for label, path in (("plain", plain), ("gz", gz), ("bgzf", bz)):
            best, table = None, None
            for _ in range(3):  # (best of three: the files were written a moment ago, and their write-back now and then stalls a reader)
                t0 = time.perf_counter()
                table, _, tot = host.match_files(store, [path])[:3]
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            res[label] = {"file_bytes": os.path.getsize(path), "seconds": round(best, 3), "gbps": round(n * READ_LEN / best / 1e9, 2),
                          "file_GBs": round(os.path.getsize(path) / best / 1e9, 2),
                          "table_equals_resident_run": bool(np.array_equal(table, want)), "reads_seen": int(tot.reads)}
        # the same .gz with the device gunzip forced into four batches (a file of any size goes batch by batch: windows, CRC-32 and the
        # record that straddles a batch are carried from one to the next)
        os.environ["GS_GUNZIP_SLOTS"] = "1024"
        try:
            t0 = time.perf_counter()
            table, _, tot = host.match_files(store, [gz])[:3]
            dt = time.perf_counter() - t0
        finally:
            del os.environ["GS_GUNZIP_SLOTS"]
        res["gz_in_batches"] = {"batches_forced": "GS_GUNZIP_SLOTS=1024 (64 MiB of compressed data per batch)", "seconds": round(dt, 3),
                                "gbps": round(n * READ_LEN / dt / 1e9, 2), "table_equals_resident_run": bool(np.array_equal(table, want)),
                                "reads_seen": int(tot.reads)}
        # the filter goal with writeback
        bloom, (keys, bits, hashes, factors, words) = _index_filter(ga, synth, torch, dev, db)
        flt = ga.FastqBloomFilter(K, bloom, 1, 0.2)
        acc = flt.accept_reads(seq, off)
        outp = os.path.join(tmp, "filtered.fastq")
        best = None
        for _ in range(2):
            if os.path.exists(outp):
                os.remove(outp)  # (truncating 0.6 GB of page cache is not part of the pipeline)
            t0 = time.perf_counter()
            tot = host.filter_files(bloom, K, [plain], 1, 0.2, filtered_path=outp)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rec_len = (2 + 8 + 1) + READ_LEN + 3 + READ_LEN + 1
        res["filter_writeback"] = {"seconds": round(best, 3), "gbps": round(n * READ_LEN / best / 1e9, 2),
                                   "accepted_reads": int(tot.filtered_reads), "accepted_equals_resident_filter": bool(int(acc.sum()) == int(tot.filtered_reads)),
                                   "output_bytes": os.path.getsize(outp),
                                   "output_size_as_expected": bool(os.path.getsize(outp) == int(acc.sum()) * rec_len)}
        # ... and from the BGZF file: members inflated on the device, the filter on the device text, the text back once for the writers
        plain_out = open(outp, "rb").read() if os.path.getsize(outp) < (2 << 30) else None
        best = None
        for _ in range(2):
            os.remove(outp)
            t0 = time.perf_counter()
            tot = host.filter_files(bloom, K, [bz], 1, 0.2, filtered_path=outp)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res["filter_writeback_bgzf"] = {"seconds": round(best, 3), "gbps": round(n * READ_LEN / best / 1e9, 2), "accepted_reads": int(tot.filtered_reads),
                                        "output_equals_plain_input_run": bool(plain_out is not None and open(outp, "rb").read() == plain_out)}
        # ... and from the single-member .gz: inflated on the device as a whole (block starts found speculatively), handed on in slices
        best = None
        for _ in range(2):
            os.remove(outp)
            t0 = time.perf_counter()
            tot = host.filter_files(bloom, K, [gz], 1, 0.2, filtered_path=outp)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res["filter_writeback_gz"] = {"seconds": round(best, 3), "gbps": round(n * READ_LEN / best / 1e9, 2), "accepted_reads": int(tot.filtered_reads),
                                      "output_equals_plain_input_run": bool(plain_out is not None and open(outp, "rb").read() == plain_out)}
        # ... and in the reference's default shape, gzipFastqOutput (C/GSConfigKey.java:155): the accepted reads as a .gz file -- gathered
        # and DEFLATE-compressed on the device (gs_deflate_dev.hip), a sixth of the bytes to fetch and to write
        import gzip as _gzip
        outz = outp + ".gz"
        for label, src in (("plain", plain), ("bgzf", bz), ("gz", gz)):
            best = None
            for _ in range(2):
                if os.path.exists(outz):
                    os.remove(outz)
                t0 = time.perf_counter()
                tot = host.filter_files(bloom, K, [src], 1, 0.2, filtered_path=outz)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            res["filter_%s_to_gz" % label] = {"seconds": round(best, 3), "gbps": round(n * READ_LEN / best / 1e9, 2), "accepted_reads": int(tot.filtered_reads),
                                              "output_bytes": os.path.getsize(outz),
                                              "zcat_equals_plain_output_run": bool(plain_out is not None and _gzip.decompress(open(outz, "rb").read()) == plain_out)}
        bloom.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def leg_table_only(ga, synth, orc, torch, dev, cores, k=16, n=10_000_000, nchk=1_000_000):
    """A store without super-k-mer records (k < 19; the reference allows k = 15..31, C/GSConfigKey.java:70, and its own
    ComprehensiveMatchTest runs k = 16): every k-mer is a slot of the bucket table, gated by the word gate.  150-bp reads have
    135 k-mer positions at k = 16 -- more than 128.  Parity: the first nchk reads."""
    db = synth.SynthDB(k=k)
    gen = torch.from_numpy(db.genomes).to(dev)
    dseq = torch.empty(n * READ_LEN, dtype=torch.uint8, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=READ_LEN)
    torch.cuda.synchronize()
    store = ga.DeviceKMerStore(k, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    info = store.info
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
    best = None
    for _ in range(4):
        m.reset()
        m.sync()
        t0 = time.perf_counter()
        _submit_reads(m, dseq, doff, 0, n)
        m.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    m.reset()
    _submit_reads(m, dseq, doff, 0, nchk)
    gt, _ = m.finish()
    seq, off = synth.reads_host(db.genomes, nchk, read_len=READ_LEN)
    odb = orc.DB(k, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off, threads=cores, per_read=False)
    ot, _ = orun.finish()
    odb.close()
    m.close()
    store.close()
    return {"workload": "match: %d reads x 150 bp, k=%d, %d-k-mer store WITHOUT records (bucket table only)" % (n, k, db.n_entries),
            "table_bytes": int(info.table_bytes), "record_bytes": int(info.rec_bytes), "ms_per_step": round(best * 1e3, 3),
            "gbps": round(n * READ_LEN / best / 1e9, 2), "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(ot, gt))}}


def leg_huge(ga, synth, orc, torch, dev, cores, cal=None, pmc=None, pmc_note=None, genera=250, n=10_000_000, nchk=1_000_000):
    """BASELINE.json configs[4]'s store size on ONE GPU: ~473 M k-mers / 5 251 values built ON THE DEVICE from 5 000 synthetic
    genomes (gs_dbbuild: FillDBGoal + DBGoal, then the device layout builder), 10 M reads through the fused kernel -- plain and as
    8 stripes in this GPU's HBM --, oracle spot check over the first nchk reads against the arrays the builder returned."""
    store, gen, g, db, kmers, vals, (t_gen, t_build, t_layout) = _huge_store(ga, synth, torch, dev, genera)
    info = store.info
    dseq = torch.empty(n * READ_LEN, dtype=torch.uint8, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    synth.reads_device(gen, g.shape[0], g.shape[1], n, dseq, doff, read_len=READ_LEN)
    torch.cuda.synchronize()
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))

    def launch():
        m.reset()
        _submit_reads(m, dseq, doff, 0, n)

    kms, wall = _kernel_ms(m, launch, 4)
    m.reset()
    _submit_reads(m, dseq, doff, 0, nchk)
    gt, _ = m.finish()
    m.close()
    seq, off = synth.reads_host(g, nchk, read_len=READ_LEN)
    odb = orc.DB(K, kmers, vals, db.n_values, db.parent_vi)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off, threads=cores, per_read=False)
    ot, _ = orun.finish()
    odb.close()
    res = {"workload": "match: %d reads x 150 bp, k=31, %d-k-mer / %d-value store built on the device from %d genomes" % (n, len(kmers), db.n_values, g.shape[0]),
           "store_kmers": int(len(kmers)), "n_values": int(db.n_values), "record_bytes": int(info.rec_bytes), "table_bytes": int(info.table_bytes),
           "mgate_bytes": int(info.mgate_bytes), "in_records_frac": round(info.n_in_records / max(1, info.n_stored), 4),
           "kernel_ms": round(kms, 3), "ms_per_step": round(wall, 3), "gbps": round(n * READ_LEN / (kms * 1e-3) / 1e9, 2),
           "build_s": {"genomes_host": round(t_gen, 1), "gs_dbbuild_incl_fetch": round(t_build, 2), "layout_on_device": round(t_layout, 2)},
           "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(ot, gt))}}
    if cal is not None:  # the same pricing as the headline: counters of the in-run PMC passes, ceilings of this run
        foot = int(info.rec_bytes + info.table_bytes + info.mgate_bytes)
        if "lines_huge" not in cal:
            cal["lines_huge"] = ga.calibrate(ga.CAL_RANDOM_LINES, foot)["rate"]
        cnt = (pmc or {}).get("huge_store") or None
        res.update(_priced(cal, cnt, pmc_note if cnt else None, kms, n, "lines_huge"))
        res["random_line_ceiling_G_per_s"] = round(cal["lines_huge"] / 1e9, 2)
    store.close()
    torch.cuda.empty_cache()
    # the same store as 8 stripes in this GPU's HBM (what every GPU of configs[4] would run, minus the xGMI hop)
    stores = ga.DeviceKMerStore.striped(K, kmers, vals, db.n_values, db.parent_vi, devices=(dev.index or 0,) * 8)
    ms = ga.FastqKMerMatcher(stores[3], ga.MatchConfig(profile=True))

    def launch_s():
        ms.reset()
        _submit_reads(ms, dseq, doff, 0, n)

    kms_s, _ = _kernel_ms(ms, launch_s, 4)
    ms.reset()
    _submit_reads(ms, dseq, doff, 0, nchk)
    gts, _ = ms.finish()
    ms.close()
    for s in stores:
        s.close()
    res["striped_8"] = {"kernel_ms": round(kms_s, 3), "gbps": round(n * READ_LEN / (kms_s * 1e-3) / 1e9, 2),
                        "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(ot, gts))}}
    del gen, dseq, doff
    torch.cuda.empty_cache()
    return res


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


def _priced(cal, cnt, src, kms, n, footprint_key):
    """roofline fields of an extra leg (same pricing as the headline)"""
    if cal is None:
        return {}
    res, fr = _resources(cnt, kms, n, cal, footprint_key, src)
    top, frac, bound = _name_bound(fr)
    res.update({"resource_fracs": fr, "bound": "hbm" if top == "hbm" and frac is not None and frac >= 0.6 else bound, "frac": frac})
    return res


def _fastq_text_device(torch, dseq, first, n, read_len=READ_LEN):
    """the four-line FASTQ text of reads [first, first + n) of a resident read buffer, built on the device: @r<8 digits>, read, +, 'I' x L"""
    width = 2 + 8 + 1
    rec = width + read_len + 3 + read_len + 1
    a = torch.empty((n, rec), dtype=torch.uint8, device=dseq.device)
    a[:, 0] = ord("@")
    a[:, 1] = ord("r")
    idx = torch.arange(first, first + n, dtype=torch.int64, device=dseq.device)
    for d in range(8):
        a[:, 2 + 7 - d] = (idx % 10 + 48).to(torch.uint8)
        idx = idx // 10
    a[:, width - 1] = 10
    a[:, width:width + read_len] = dseq[first * read_len:(first + n) * read_len].view(n, read_len)
    a[:, width + read_len] = 10
    a[:, width + read_len + 1] = ord("+")
    a[:, width + read_len + 2] = 10
    a[:, width + read_len + 3:rec - 1] = ord("I")
    a[:, rec - 1] = 10
    return a.view(-1)


def _inflate_members_crc(path, threads):
    """every BGZF member of `path` inflated by zlib (threads side by side) -> (CRC-32 of the whole text, its length, members); the
    members' own trailers are checked on the way"""
    import mmap
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    members, o, n = [], 0, len(mm)
    while o + 28 <= n:
        bsize = struct.unpack_from("<H", mm, o + 16)[0] + 1
        members.append((o, bsize))
        o += bsize
    if o != n:
        raise ValueError("%s is not BGZF to its end" % path)

    def run(span):
        crc, size = 0, 0
        for at, bsize in span:
            text = zlib.decompress(mm[at + 18:at + bsize - 8], -15)
            c, isz = struct.unpack_from("<II", mm, at + bsize - 8)
            if zlib.crc32(text) != c or len(text) != isz:
                raise ValueError("member at %d: trailer does not match its text" % at)
            crc = zlib.crc32(text, crc)  # (running over the span's text; the spans are joined below)
            size += isz
        return crc, size
    per = max(1, (len(members) + threads * 8 - 1) // (threads * 8))
    spans = [members[i:i + per] for i in range(0, len(members), per)]
    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(run, spans))
    crc, size = 0, 0
    for c, ln in parts:
        crc = _crc32_combine(crc, c, ln) if size else c
        size += ln
    return crc, size, len(members)


def _file_crc(path, threads):
    """CRC-32 of a file's bytes, pieces on threads"""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    size = os.path.getsize(path)
    piece = 64 << 20

    def run(a):
        with open(path, "rb") as f:
            f.seek(a)
            return zlib.crc32(f.read(min(piece, size - a))), min(piece, size - a)
    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(run, range(0, size, piece)))
    crc, done = 0, 0
    for c, ln in parts:
        crc = _crc32_combine(crc, c, ln) if done else c
        done += ln
    return crc, size


def leg_configs2_file(ga, torch, bloom, dseq, n, acc, cores, check_plain=True):
    """BASELINE.json configs[2] END TO END in the reference's default shape (gzipFastqOutput, C/GSConfigKey.java:155; FilterGoal.java:76):
    n reads as a block-gzip FASTQ file written in this run (the text built and compressed on the device, 4 M reads at a time)
    -> gs_host_filter_files -> the accepted reads as a .gz file.  Input inflated on the device, filter on the device text, accepted
    records gathered and DEFLATE-compressed on the device, only compressed bytes cross PCIe either way.  Checks: accepted count =
    the resident run's; every member of the output inflated by zlib, trailer against text; the whole text's CRC-32 and length equal
    the PLAIN-output run's file (zcat-equal)."""
    from genestrip_amd import host
    tmp = tempfile.mkdtemp(prefix="gsbench_c2_")
    res = {"reads": n, "tmp": tempfile.gettempdir()}
    try:
        src = os.path.join(tmp, "reads.fastq.gz")
        slab = 4_000_000
        t0 = time.perf_counter()
        defl = ga.DeviceDeflater()
        rec = (2 + 8 + 1) + READ_LEN + 3 + READ_LEN + 1
        out = torch.empty(ga.deflate_bound(slab * rec), dtype=torch.uint8).pin_memory().numpy()
        with open(src, "wb") as f:
            for a in range(0, n, slab):
                k = min(slab, n - a)
                text = _fastq_text_device(torch, dseq, a, k)
                m = defl.pack(text, k * rec, out)
                f.write(out[:m].data)
                del text
            f.write(ga.BGZF_EOF)
        defl.close()
        del out
        torch.cuda.empty_cache()
        res["input"] = {"bytes": os.path.getsize(src), "text_bytes": n * rec, "written_s": round(time.perf_counter() - t0, 2),
                        "how": "text and BGZF members made on the device (gs_deflater_pack), 4 M reads per call"}
        outp = os.path.join(tmp, "filtered.fastq.gz")
        best, tot = None, None
        for _ in range(2):
            if os.path.exists(outp):
                os.remove(outp)
            t0 = time.perf_counter()
            tot = host.filter_files(bloom, K, [src], 1, 0.2, filtered_path=outp)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        want = int(acc.sum().item()) if hasattr(acc, "sum") else int(acc)
        res.update({"seconds": round(best, 3), "gbps": round(n * READ_LEN / best / 1e9, 2), "accepted_reads": int(tot.filtered_reads),
                    "accepted_equals_resident_run": bool(int(tot.filtered_reads) == want and int(tot.reads) == n),
                    "output_bytes": os.path.getsize(outp), "output_text_bytes": int(tot.filtered_reads) * rec})
        t0 = time.perf_counter()
        crc, size, members = _inflate_members_crc(outp, cores)
        res["output_check"] = {"members": members, "every_member_inflates_under_zlib_to_its_trailer": True,
                               "text_bytes_as_expected": bool(size == int(tot.filtered_reads) * rec), "seconds": round(time.perf_counter() - t0, 1)}
        if check_plain:
            plain = os.path.join(tmp, "filtered.fastq")
            t0 = time.perf_counter()
            tot_p = host.filter_files(bloom, K, [src], 1, 0.2, filtered_path=plain)
            dt = time.perf_counter() - t0
            pcrc, psize = _file_crc(plain, cores)
            res["plain_output"] = {"seconds": round(dt, 3), "gbps": round(n * READ_LEN / dt / 1e9, 2), "output_bytes": psize,
                                   "accepted_reads": int(tot_p.filtered_reads)}
            res["output_check"]["zcat_equal_to_the_plain_output_run"] = bool(pcrc == crc and psize == size)
            os.remove(plain)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


def legs_large(ga, synth, orc, torch, dev, legs, cores, cal, pmc, pmc_note):
    """BASELINE.json configs[2] / configs[3] per GPU: a 47 M-k-mer / 526-value store (1 GiB table, four times the
    Infinity Cache) and the XOR index filter over its species k-mers (~47 M keys, 1.8 G bits, 27 hashes).  Parity over the
    first 2 M reads of the timed stream; the filter also at configs[2]'s own read count (100 M reads resident in HBM)."""
    res = {}
    n, nchk = 10_000_000, 2_000_000
    t0 = time.perf_counter()
    db = synth.SynthDB(k=K, genera=25, species_per_genus=20)
    t_db = time.perf_counter() - t0
    gen = torch.from_numpy(db.genomes).to(dev)
    dseq = torch.empty(n * READ_LEN, dtype=torch.uint8, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=READ_LEN)
    torch.cuda.synchronize()
    seq, off = synth.reads_host(db.genomes, nchk, read_len=READ_LEN)

    def counters(name):
        if pmc and pmc.get(name):
            return pmc[name], pmc_note
        c, s = _pmc_committed(name)
        return c, (s + "; in-run passes: " + pmc_note) if s else None

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
            _submit_reads(m, dseq, doff, 0, n)

        kms, wall = _kernel_ms(m, launch, 5)
        odb = orc.DB(K, db.kmers, db.value_idx, db.n_values, db.parent_vi, bloom_gate=True)
        orun = orc.MatchRun(odb)
        orun.submit(seq, off, threads=cores, per_read=False)
        ot, _ = orun.finish()
        odb.close()
        m.reset()
        _submit_reads(m, dseq, doff, 0, nchk)
        gt, _ = m.finish()
        cnt, src = counters("large_store")
        ach = n * BYTES_PER_READ / (kms * 1e-3) / 1e9
        obj = {
            "workload": "match: %d reads x 150 bp, k=31, %d-k-mer / %d-value store" % (n, db.n_entries, db.n_values),
            "store_kmers": int(db.n_entries), "n_values": int(db.n_values), "record_bytes": int(info.rec_bytes), "table_bytes": int(info.table_bytes),
            "mgate_bytes": int(info.mgate_bytes), "kernel": "gs_match_kernel<global counters, k=31>",
            "kernel_ms": round(kms, 3), "ms_per_step": round(wall, 3), "gbps": round(n * READ_LEN / (kms * 1e-3) / 1e9, 2),
            "frac_survey_convention": round(ach / HBM_PEAK_GBS, 4),
            "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(ot, gt))},
            "build_s": {"synthetic_arrays": round(t_db, 1), "gs_db_create": round(t_store, 1)}}
        obj.update(_priced(cal, cnt, src, kms, n, "lines_1GiB"))
        res["large_store"] = obj
        m.close()
        store.close()
    if "filter" in legs:
        bloom, (keys, bits, hashes, factors, words) = _index_filter(ga, synth, torch, dev, db)
        flt = ga.FastqBloomFilter(K, bloom, 1, 0.2, profile=True)
        acc = torch.empty(n, dtype=torch.uint8, device=dev)
        kms, wall = _kernel_ms(flt, lambda: flt.submit(dseq, doff, acc, n_reads=n), 5)
        ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)  # the oracle builds its own filter from the same keys
        ob.put_many(keys, threads=cores)
        same_filter = bool(ob.bits == bits and np.array_equal(ob.words, words))
        want = ob.filter_batch(K, 1, 0.2, seq, off, threads=cores)
        got = acc[:nchk].cpu().numpy()
        cnt, src = counters("filter")
        ach = n * BYTES_PER_READ / (kms * 1e-3) / 1e9
        obj = {
            "workload": "filter: %d reads x 150 bp, k=31, XOR index filter of %d keys (%d bits, %d hashes), minPosCount 1" % (n, len(keys), bits, hashes),
            "filter_keys": int(len(keys)), "filter_bytes": int(len(words) * 8), "n_hashes": int(hashes),
            "kernel": "gs_filter_kernel", "kernel_ms": round(kms, 3), "ms_per_step": round(wall, 3),
            "gbps": round(n * READ_LEN / (kms * 1e-3) / 1e9, 2), "accepted_frac": round(float(acc.float().mean()), 4),
            "frac_survey_convention": round(ach / HBM_PEAK_GBS, 4),
            "parity": {"reads_checked": nchk, "bit_exact": bool(np.array_equal(want, got)), "filter_bits_equal_oracle": same_filter}}
        obj.update(_priced(cal, cnt, src, kms, n, "lines_filter"))
        # configs[2] at its own read count: 100 M reads resident in HBM (15 GB), one launch; the accept flags of the first
        # 10 M reads must be the ones the 10 M-read launch produced (same stream, same read numbers)
        n2 = 100_000_000
        del dseq, doff
        torch.cuda.empty_cache()
        dseq2 = torch.empty(n2 * READ_LEN, dtype=torch.uint8, device=dev)
        doff2 = torch.empty(n2 + 1, dtype=torch.int64, device=dev)
        synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n2, dseq2, doff2, read_len=READ_LEN)
        acc2 = torch.empty(n2, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        kms2, _ = _kernel_ms(flt, lambda: flt.submit(dseq2, doff2, acc2, n_reads=n2), 2, warm=1)
        obj["configs2_100M_reads"] = {"reads": n2, "kernel_ms": round(kms2, 2), "gbps": round(n2 * READ_LEN / (kms2 * 1e-3) / 1e9, 2),
                                      "accepted_frac": round(float(acc2.float().mean()), 4),
                                      "first_10M_flags_equal_the_10M_launch": bool(torch.equal(acc2[:n], acc)),
                                      "first_2M_flags_equal_oracle": bool(np.array_equal(want, acc2[:nchk].cpu().numpy()))}
        if "c2file" in legs:
            obj["configs2_file"] = leg_configs2_file(ga, torch, bloom, dseq2, n2, acc2, cores)
        res["filter"] = obj
        bloom.close()
        del dseq2, doff2, acc2
        torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
