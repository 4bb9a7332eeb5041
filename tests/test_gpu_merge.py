"""gs_match_merge / gs_host_match_files_multi (VERDICT r01 "Next round" 5): the runs of ONE process -- what a JVM host
with one gs_run per GPU calls -- merged through the C ABI.  On the one-GPU box the replicas share device 0, which
exercises the intra-device stage with real kernels; GS_MERGE_FORCE_RCCL=1 sends the (single) device leader through the
RCCL collectives as well.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import binding, host, synth
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _oracle(sdb, seq, off, **cfg):
    run = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi), **cfg)
    run.submit(seq, off, threads=8, per_read=False)
    return run.finish()[0]


@pytest.mark.parametrize("n_runs,force_rccl,cfg", [(1, False, {}), (3, False, {}), (2, True, {}), (1, True, {}),
                                                   (3, False, dict(count_unique=False, threshold=3)), (4, True, dict(max_paths=128))])
def test_runs_of_one_process_merge_to_the_single_run_table(sdb, monkeypatch, n_runs, force_rccl, cfg):
    if force_rccl:
        monkeypatch.setenv("GS_MERGE_FORCE_RCCL", "1")
    seq, off = synth.reads_host(sdb.genomes, 12000, read_len=150, seed=5)
    off = off.astype(np.uint64)
    # a tie for the longest contig between the shards: the same read at the start of every shard (first read number wins)
    want = _oracle(sdb, seq, off, **cfg)
    stores = [ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi) for _ in range(n_runs)]
    ms = [ga.FastqKMerMatcher(s, ga.MatchConfig(**cfg)) for s in stores]
    cuts = np.linspace(0, 12000, n_runs + 1).astype(int)
    for m, a, b in zip(ms, cuts[:-1], cuts[1:]):
        m.submit(seq[int(off[a]):int(off[b])], off[a:b + 1] - off[a], int(a), n_reads=int(b - a))
    binding.merge_runs(ms)
    tables = [m.finish()[0] for m in ms]
    for t in tables:  # every run holds the global state
        assert np.array_equal(t, want), np.argwhere(t != want)[:6]
    # merging is not a one-shot: a reset run starts from zero again
    ms[0].reset()
    ms[0].submit(seq[:int(off[100])], off[:101], 0, n_reads=100)
    again = ms[0].finish()[0]
    assert np.array_equal(again, _oracle(sdb, seq[:int(off[100])], off[:101], **cfg))
    for m in ms:
        m.close()
    for s in stores:
        s.close()


def test_merge_refuses_runs_on_different_stores(sdb):
    a = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    b = ga.DeviceKMerStore(31, sdb.kmers[:-5], sdb.value_idx[:-5], sdb.n_values, sdb.parent_vi)
    ma, mb = ga.FastqKMerMatcher(a), ga.FastqKMerMatcher(b)
    with pytest.raises(ga.GsError) as e:
        binding.merge_runs([ma, mb])
    assert e.value.code == -1
    with pytest.raises(ga.GsError):
        binding.merge_runs([ma, ma])
    ma.close(), mb.close(), a.close(), b.close()


def _write_fastq(path, seq, off, start):
    with open(path, "wb") as f:
        for i in range(len(off) - 1):
            s = seq[int(off[i]):int(off[i + 1])].tobytes()
            f.write(b"@r%d\n%s\n+\n%s\n" % (start + i, s, b"F" * len(s)))


@pytest.mark.parametrize("n_replicas", [1, 2, 3])
def test_files_over_several_replicas_of_one_process(sdb, tmp_path, n_replicas):
    """gs_host_match_files_multi: five files dealt to the replicas, every replica on a thread of its own; the table (and
    the running read number of the max contig) must equal the single-replica pipeline and the oracle"""
    seq, off = synth.reads_host(sdb.genomes, 15000, read_len=150, seed=23)
    off = off.astype(np.uint64)
    cuts = [0, 4000, 4001, 9000, 12500, 15000]
    paths = []
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        p = str(tmp_path / f"s{i}.fastq")
        _write_fastq(p, seq[int(off[a]):int(off[b])], off[a:b + 1] - off[a], a)
        paths.append(p)
    want = _oracle(sdb, seq, off)
    stores = [ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi) for _ in range(n_replicas)]
    table, _, tot = host.match_files_multi(stores, paths)
    assert (tot.reads, tot.bps, tot.kmers) == (15000, 15000 * 150, 15000 * 120)
    assert np.array_equal(table, want), np.argwhere(table != want)[:6]
    single, _, _ = host.match_files(stores[0], paths)
    assert np.array_equal(single, want)
    for s in stores:
        s.close()
