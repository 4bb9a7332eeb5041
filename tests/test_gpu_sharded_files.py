"""One process per GPU over the files of a sample (genestrip_amd.distributed.match_files_sharded): here two processes
share the one GPU of the test box and merge through gloo, which exercises everything but the RCCL transport -- file
split, raw-text ingest into separate runs, read numbers file << 32 | read, state merge, unique bitmap OR, max-contig
tie-break across processes.  The result must equal the single-process pipeline and the oracle."""
import gzip
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _files(tmp, sdb, n_files=5, per_file=700):
    from genestrip_amd import synth
    paths, blobs = [], []
    for f in range(n_files):
        seq, off = synth.reads_host(sdb.genomes, per_file, read_len=150, seed=50 + f)
        if f == 3:  # the same reads as file 1: ties in the max-contig column must go to the earlier file
            seq, off = synth.reads_host(sdb.genomes, per_file, read_len=150, seed=51)
        data = b"".join(b"@f%d_%d\n%s\n+\n%s\n" % (f, i, seq[int(off[i]):int(off[i + 1])].tobytes(), b"F" * 150)
                        for i in range(per_file))
        p = os.path.join(tmp, f"s{f}.fastq" + (".gz" if f % 2 else ""))
        with (gzip.open(p, "wb") if f % 2 else open(p, "wb")) as fh:
            fh.write(data)
        paths.append(p)
        blobs.append(data)
    return paths, b"".join(blobs)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import genestrip_amd as ga
    from genestrip_amd import synth
    from genestrip_amd.distributed import match_files_sharded
    sdb = synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)
    paths = sorted(os.path.join(tmp, x) for x in os.listdir(tmp) if x.startswith("s"))
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    table, _, tot = match_files_sharded(m, paths, via_host=True)
    np.save(os.path.join(tmp, f"rank{rank}.npy"), table)
    np.save(os.path.join(tmp, f"tot{rank}.npy"), np.array(tot))
    m.close()
    store.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_share_the_files_of_a_run(tmp_path):
    import genestrip_amd as ga
    from genestrip_amd import host, synth
    from oracle import gs_oracle as orc
    sdb = synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)
    paths, blob = _files(str(tmp_path), sdb)
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # single process, same files in the same order
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    single, _, tot = host.match_files(store, paths)
    store.close()
    p = orc.parse_fastq(blob, k=31)
    run = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    run.submit(p["seq"], p["seq_off"])
    want, _ = run.finish()
    assert np.array_equal(single, want)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npy"))
        assert np.array_equal(got, want), np.argwhere(got != want)[:6]
        assert tuple(np.load(os.path.join(str(tmp_path), f"tot{r}.npy"))) == (tot.reads, tot.kmers, tot.bps)


def test_single_rank_rccl_group(tmp_path):
    """the default (device buffers over RCCL) path with a one-rank group: same table as the plain pipeline"""
    import genestrip_amd as ga
    from genestrip_amd import host, synth
    from genestrip_amd.distributed import match_files_sharded
    sdb = synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)
    paths, _ = _files(str(tmp_path), sdb, n_files=3, per_file=400)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29900 + (os.getpid() % 90))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
        m = ga.FastqKMerMatcher(store)
        table, _, tot = match_files_sharded(m, paths)
        m.close()
        single, _, stot = host.match_files(store, paths)
        store.close()
    finally:
        dist.destroy_process_group()
    assert np.array_equal(table, single)
    assert tot == (stot.reads, stot.kmers, stot.bps)
