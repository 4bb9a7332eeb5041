"""The multi-GPU paths on REAL devices (VERDICT r02 "Next round" 2): every test here needs at least two GPUs in this process'
sight and is skipped on the one-GPU box -- the same code runs there with replicas / stripes sharing device 0
(test_gpu_merge.py, test_gpu_striped.py, test_gpu_partitioned.py); on a multi-GPU node these switch themselves on and exercise
what one device cannot: ncclCommInitAll over distinct devices, hipDeviceEnablePeerAccess + record lines over xGMI, HIP IPC
between processes that own different GPUs.  Every result is compared with the CPU oracle.  Run with -m gpu."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import binding, host, synth
from oracle import gs_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(ga.device_count() < 2, reason="needs at least two GPUs")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _oracle(sdb, seq, off, **cfg):
    run = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi), **cfg)
    run.submit(seq, off, threads=8, per_read=False)
    return run.finish()[0]


def _devices():
    return list(range(min(ga.device_count(), 8)))


@pytest.mark.parametrize("cfg", [{}, dict(count_unique=False, threshold=3), dict(max_paths=128)])
def test_runs_on_distinct_devices_merge_over_rccl(sdb, cfg):
    """gs_match_merge: one replica + one run per device, the device leaders meet in a real ncclCommInitAll communicator"""
    devs = _devices()
    n = 4000 * len(devs)
    seq, off = synth.reads_host(sdb.genomes, n, read_len=150, seed=5)
    off = off.astype(np.uint64)
    want = _oracle(sdb, seq, off, **cfg)
    stores = [ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, device=d) for d in devs]
    ms = [ga.FastqKMerMatcher(s, ga.MatchConfig(**cfg)) for s in stores]
    cuts = np.linspace(0, n, len(devs) + 1).astype(int)
    for m, a, b in zip(ms, cuts[:-1], cuts[1:]):
        m.submit(seq[int(off[a]):int(off[b])], off[a:b + 1] - off[a], int(a), n_reads=int(b - a))
    binding.merge_runs(ms)
    for m in ms:
        t = m.finish()[0]
        assert np.array_equal(t, want), np.argwhere(t != want)[:6]
    for m in ms:
        m.close()
    for s in stores:
        s.close()


def _write_fastq(path, seq, off, start):
    with open(path, "wb") as f:
        for i in range(len(off) - 1):
            s = seq[int(off[i]):int(off[i + 1])].tobytes()
            f.write(b"@r%d\n%s\n+\n%s\n" % (start + i, s, b"F" * len(s)))


@pytest.mark.parametrize("container", ["plain", "gzip", "bgzf"])
def test_files_dealt_to_replicas_on_distinct_devices(sdb, tmp_path, container):
    """gs_host_match_files_multi over one replica per device; gzip / BGZF files are inflated on the device each replica lives on (every
    device its own inflater, staging buffers and events)"""
    import gzip
    from conftest import bgzf
    devs = _devices()
    seq, off = synth.reads_host(sdb.genomes, 15000, read_len=150, seed=23)
    off = off.astype(np.uint64)
    cuts = [0, 4000, 4001, 9000, 12500, 15000]
    paths = []
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        p = str(tmp_path / f"s{i}.fastq")
        _write_fastq(p, seq[int(off[a]):int(off[b])], off[a:b + 1] - off[a], a)
        if container != "plain":
            raw = open(p, "rb").read()
            p += ".gz"
            open(p, "wb").write(gzip.compress(raw, 6, mtime=0) if container == "gzip" else bgzf(raw, block=30000, level=1))
        paths.append(p)
    want = _oracle(sdb, seq, off)
    stores = [ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, device=d) for d in devs]
    table, _, tot = host.match_files_multi(stores, paths)
    assert (tot.reads, tot.bps, tot.kmers) == (15000, 15000 * 150, 15000 * 120)
    assert np.array_equal(table, want), np.argwhere(table != want)[:6]
    for s in stores:
        s.close()


def test_striped_store_over_distinct_devices_reads_foreign_lines_by_peer_access(sdb):
    """gs_db_create_striped(devices = one per GPU): every handle's kernel loads the record lines of the other stripes from
    another GPU's HBM; one run per handle, merged, against the oracle -- and every handle alone against the oracle of its reads"""
    devs = _devices()
    n = 3000 * len(devs)
    seq, off = synth.reads_host(sdb.genomes, n, read_len=150, seed=7)
    off = off.astype(np.uint64)
    stores = ga.DeviceKMerStore.striped(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, devices=tuple(devs))
    assert stores[0].info.n_stripes == len(devs)
    cuts = np.linspace(0, n, len(devs) + 1).astype(int)
    ms = [ga.FastqKMerMatcher(s) for s in stores]
    for m, a, b in zip(ms, cuts[:-1], cuts[1:]):
        sl, so = seq[int(off[a]):int(off[b])], off[a:b + 1] - off[a]
        cv, fl = m.match_reads(sl, so, int(a))
        orun = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
        ocv, ofl = orun.submit(sl, so, first_read_no=int(a))
        assert np.array_equal(cv, ocv) and np.array_equal(fl, ofl)
    binding.merge_runs(ms)
    want = _oracle(sdb, seq, off)
    for m in ms:
        assert np.array_equal(m.finish()[0], want)
    for m in ms:
        m.close()
    for s in stores:
        s.close()


@pytest.mark.parametrize("mode", ["sharded", "striped"])
def test_one_process_per_gpu_bench_flow_over_rccl(mode):
    """bench.py --gpus 2 started WITHOUT a launcher: it must start its two ranks itself, one GPU each, merge over RCCL (striped:
    each rank owns a stripe in its GPU's HBM and maps the other's through HIP IPC -- lines travel over xGMI), and print ONE line
    with n_gpus 2 whose merged table is bit-exact against the oracle over both ranks' reads"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GS_BENCH_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "400000",
           "--cpu-seconds", "0", "--mode", mode]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["parity"]["bit_exact"] is True and out["parity"]["reads_checked"] == 800000
    assert out["parity"]["merged_table_identical_on_all_ranks"] is True
