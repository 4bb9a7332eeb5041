"""The BASELINE.json configurations at their own size on the MI355X (VERDICT r01 "Next round" 1):
* configs[0]  the reference's bundled sample.fastq.gz through the file pipeline (gs_host_match_files), totals of README.md:169
* configs[2]  filter against an index filter of ~47 M keys / 1.8 G bits / 27 hashes (64-bit abs-mod, word offsets > 2^25)
* configs[3]  match against a 47 M-k-mer / 526-value store whose 1 GiB table no longer fits the 256 MiB Infinity Cache
Each with an oracle spot check and the size-independent property that sharding the reads must not change the result.
Needs an MI355X: run with -m gpu."""
import gzip
import os

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import host, synth
from conftest import GOLDEN
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu

N_READS = 2_000_000
N_CHECK = 100_000


@pytest.fixture(scope="module")
def big():
    import torch
    db = synth.SynthDB(genera=25, species_per_genus=20)  # 47.1 M k-mers, 526 values
    assert db.n_entries > 45_000_000 and db.n_values == 526
    gen = torch.from_numpy(db.genomes).cuda()
    dseq = torch.empty(N_READS * 150, dtype=torch.uint8, device="cuda")
    doff = torch.empty(N_READS + 1, dtype=torch.int64, device="cuda")
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], N_READS, dseq, doff, seed=4242)
    seq, off = synth.reads_host(db.genomes, N_CHECK, seed=4242)
    assert np.array_equal(dseq[:N_CHECK * 150].cpu().numpy(), seq)
    return db, dseq, doff, seq, off


def test_match_on_the_47m_kmer_store(big):
    db, dseq, doff, seq, off = big
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    info = store.info
    assert info.n_stored == db.n_entries
    assert info.table_bytes + info.rec_bytes >= 1 << 29  # HBM resident: beyond the 256 MiB Infinity Cache
    m = ga.FastqKMerMatcher(store)
    m.submit(dseq, doff, 0, n_reads=N_READS)
    whole, _ = m.finish()
    # shards of uneven size with global read numbers: the table must not change (K4 of the reference, at size)
    m.reset()
    cuts = [0, 1, 700_001, 1_300_000, N_READS]
    for a, b in zip(cuts[:-1], cuts[1:]):
        m.submit(dseq, doff[a:].contiguous(), a, n_reads=b - a)
    parts, _ = m.finish()
    assert np.array_equal(whole, parts)
    # oracle (sorted array + Blocked-Bloom gate + binary search) on the first 100 k reads
    odb = orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi, bloom_gate=True)
    orun = orc.MatchRun(odb)
    ocv, ofl = orun.submit(seq, off, threads=16)
    ot, _ = orun.finish()
    m.reset()
    import torch
    cv = torch.empty(N_CHECK, dtype=torch.int32, device="cuda")
    fl = torch.empty(N_CHECK, dtype=torch.uint8, device="cuda")
    m.submit(dseq, doff, 0, n_reads=N_CHECK, class_vi=cv, flags=fl)
    gt, _ = m.finish()
    assert np.array_equal(ot, gt)
    assert np.array_equal(cv.cpu().numpy(), ocv) and np.array_equal(fl.cpu().numpy(), ofl)
    counts = np.bincount(db.value_idx, minlength=db.n_values)
    assert np.all(whole[:, 3] <= counts) and whole[:, 0].sum() > 0.4 * N_READS
    m.close()
    store.close()


def test_the_47m_kmer_store_striped_eight_ways(big):
    """configs[4]'s shape on one GPU: the record table in 8 stripes (what 8 GPUs would each hold), three runs on different
    handles over disjoint shards of the reads, merged -- against ONE run on the plain store, and the oracle"""
    from genestrip_amd import binding
    db, dseq, doff, seq, off = big
    plain = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    m = ga.FastqKMerMatcher(plain)
    m.submit(dseq, doff, 0, n_reads=N_READS)
    whole, _ = m.finish()
    m.close()
    pinfo = plain.info
    plain.close()
    stores = ga.DeviceKMerStore.striped(31, db.kmers, db.value_idx, db.n_values, db.parent_vi, devices=(0,) * 8)
    infos = [s.info for s in stores]
    # (the plain store's layout comes from the device builder, the stripes' from the host builder: same rules, other ties)
    assert sum(i.stripe_bytes for i in infos) == infos[0].rec_bytes + infos[0].table_bytes and infos[3].n_stored == pinfo.n_stored
    assert abs(infos[3].n_in_records - pinfo.n_in_records) < 0.05 * pinfo.n_stored
    assert max(i.stripe_bytes for i in infos) - min(i.stripe_bytes for i in infos) <= 64 + 4 * 64  # one record line, four buckets
    ms = [ga.FastqKMerMatcher(stores[i]) for i in (0, 3, 7)]
    cuts = [0, 650_001, 1_300_000, N_READS]
    for r, (a, b) in zip(ms, zip(cuts[:-1], cuts[1:])):
        r.submit(dseq, doff[a:].contiguous(), a, n_reads=b - a)
    binding.merge_runs(ms)
    for r in ms:
        t, _ = r.finish()
        assert np.array_equal(t, whole), np.argwhere(t != whole)[:6]
    odb = orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi, bloom_gate=True)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off, threads=16, per_read=False)
    ot, _ = orun.finish()
    ms[1].reset()
    ms[1].submit(dseq, doff, 0, n_reads=N_CHECK)
    gt, _ = ms[1].finish()
    assert np.array_equal(ot, gt)
    for r in ms:
        r.close()
    for s in stores:
        s.close()


def test_filter_on_the_full_size_index(big):
    import torch
    db, dseq, doff, seq, off = big
    keys = db.kmers[np.isin(db.value_idx, db.species_vi)]  # BloomIndexGoal: the k-mers of the requested taxa
    bits, hashes, factors = synth.xor_bloom_geometry(len(keys), 1e-8)
    assert len(keys) > 45_000_000 and bits > 1_800_000_000 and hashes == 27
    dwords = torch.zeros((bits + 63) // 64, dtype=torch.int64, device="cuda")
    synth.xor_bloom_device(torch.from_numpy(keys).cuda(), len(keys), bits, torch.from_numpy(factors).cuda(), hashes, dwords)
    words = dwords.cpu().numpy().view(np.uint64)
    del dwords
    # the oracle builds its own filter from the same keys: geometry, hash factors and every bit must agree
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys, threads=16)
    assert (ob.bits, ob.hashes) == (bits, hashes) and np.array_equal(ob.hash_factors, factors)
    assert np.array_equal(ob.words, words)
    bloom = ga.DeviceBloomFilter(ga.BLOOM_XOR, bits, factors, words)
    for min_pos, ratio in ((1, 0.2), (0, 0.2), (30, 0.0)):
        flt = ga.FastqBloomFilter(31, bloom, min_pos, ratio)
        acc = torch.empty(N_READS, dtype=torch.uint8, device="cuda")
        flt.submit(dseq, doff, acc, n_reads=N_READS)
        flt.sync()
        whole = acc.cpu().numpy()
        want = ob.filter_batch(31, min_pos, ratio, seq, off, threads=16)
        assert np.array_equal(whole[:N_CHECK], want)
        assert 0.3 * N_READS < whole.sum() < 0.6 * N_READS  # half of the stream comes from the indexed genomes
        # sharding invariance at size
        acc2 = torch.zeros(N_READS, dtype=torch.uint8, device="cuda")
        cuts = [0, 999_999, N_READS]
        for a, b in zip(cuts[:-1], cuts[1:]):
            flt.submit(dseq, doff[a:].contiguous(), acc2[a:], n_reads=b - a)
        flt.sync()
        assert np.array_equal(acc2.cpu().numpy(), whole)
    bloom.close()


def test_reference_sample_fastq_through_the_file_pipeline(tmp_path):
    """configs[0]: data/projects/human_virus/fastq/sample.fastq.gz (fixture K6) through gs_host_match_files -- gzip
    decoder, device-side record scan, match kernel -- against a store made of k-mers of the sample itself.  Totals must
    be the README's (6565 reads / 658255 bps / 461305 k-mers, README.md:169), table, Kraken lines and filtered reads
    the oracle's."""
    path = os.path.join(GOLDEN, "human_virus", "sample.fastq.gz")
    data = gzip.open(path).read()
    rd = orc.parse_fastq(data, k=31)
    seq, off = rd["seq"], rd["seq_off"]
    n = len(off) - 1
    assert (n, int(off[-1]), rd["total_kmers"]) == (6565, 658255, 461305)
    # store: k-mers of every 9th read under a small tree (root -> 2 genera -> 2 species each), value by read number
    parent = np.array([-1, 0, 0, 1, 1, 2, 2], dtype=np.int32)
    taxids = ["1", "10", "20", "11", "12", "21", "22"]
    d = {}
    for i in range(0, n, 9):
        r = seq[int(off[i]):int(off[i + 1])].tobytes()
        for x in orc.canonical_kmers(r, 31).tolist():
            d.setdefault(x, 3 + (i // 9) % 4 if i % 5 else 1 + (i // 9) % 2)
    keys = np.array(sorted(d), dtype=np.int64)
    vals = np.array([d[x] for x in keys.tolist()], dtype=np.int32)
    store = ga.DeviceKMerStore(31, keys, vals, 7, parent)
    kr, fq = str(tmp_path / "k.out"), str(tmp_path / "f.fastq")
    table, dtable, tot = host.match_files(store, [path], kraken_out_path=kr, filtered_path=fq, taxids=taxids)
    assert (tot.reads, tot.bps, tot.kmers) == (6565, 658255, 461305)
    odb = orc.DB(31, keys, vals, 7, parent)
    orun = orc.MatchRun(odb)
    ocv, ofl = orun.submit(seq, off)
    ot, _ = orun.finish()
    assert ot[:, orc.C_KMERS].sum() > 50_000
    assert np.array_equal(table, ot)
    desc = rd["desc"]
    doff_ = rd["desc_off"]
    names = {-1: "0", -2: "A"}
    want = []
    for i in range(n):
        r = seq[int(off[i]):int(off[i + 1])].tobytes()
        segs = odb.segments(r)
        if not segs:
            continue
        dline = desc[int(doff_[i]):int(doff_[i + 1])].tobytes().decode()
        name = dline[1:].split(" ")[0]
        segtxt = " ".join(f"{taxids[c] if c >= 0 else names[c]}:{m}" for c, m in segs)
        want.append(f"{'C' if ocv[i] >= 0 else 'U'}\t{name}\t{taxids[ocv[i]] if ocv[i] >= 0 else '0'}\t{len(r)}\t{segtxt}")
    assert open(kr).read().rstrip("\n").split("\n") == want
    assert tot.filtered_reads == int((ofl & orc.F_RETURNED != 0).sum()) > 700
    assert open(fq).read().count("\n") == 4 * tot.filtered_reads
    # the same file again without per-read outputs (text mode end to end) and in sharded form must agree
    table2, _, tot2 = host.match_files(store, [path])
    assert np.array_equal(table2, ot) and (tot2.reads, tot2.bps, tot2.kmers) == (6565, 658255, 461305)
    store.close()


def test_match_on_the_473m_kmer_store_built_on_the_device():
    """configs[4]'s store size on one GPU (VERDICT r03 weak 1 ii): ~473 M k-mers / 5 251 values built ON THE DEVICE from 5 000
    synthetic genomes (gs_dbbuild + the device layout builder: context-keyed gate, second-bucket hints, an overflow table that holds
    several per cent of the k-mers), 4 M reads through the fused kernel; table, classes and flags of the first 1 M reads against the
    oracle over the arrays the builder returned; sharding the reads must not change the table."""
    import torch
    n, nchk = 4_000_000, 1_000_000
    db = synth.SynthDB(k=31, genera=250, species_per_genus=20, build=False)
    g = db.genomes
    gen = torch.from_numpy(g).cuda()
    goff = torch.arange(g.shape[0] + 1, dtype=torch.int64, device="cuda") * g.shape[1]
    b = ga.DeviceDbBuilder(31, db.n_values, db.parent_vi)
    b.add(gen.reshape(-1), goff, db.species_vi, update=False)
    b.add(gen.reshape(-1), goff, db.species_vi, update=True)
    kmers, vals = b.finish()
    store = b.to_store()
    b.close()
    info = store.info
    assert len(kmers) > 450_000_000 and db.n_values == 5251 and info.n_stored == len(kmers)
    assert info.rec_bytes >= 1 << 33 and 0.85 < info.n_in_records / info.n_stored < 1.0  # (the rest lives in the overflow table)
    dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
    doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    synth.reads_device(gen, g.shape[0], g.shape[1], n, dseq, doff, seed=4242)
    m = ga.FastqKMerMatcher(store)
    m.submit(dseq, doff, 0, n_reads=n)
    whole, _ = m.finish()
    m.reset()
    cuts = [0, 3, 1_500_001, n]
    for a, c in zip(cuts[:-1], cuts[1:]):
        m.submit(dseq, doff[a:].contiguous(), a, n_reads=c - a)
    parts, _ = m.finish()
    assert np.array_equal(whole, parts)
    seq, off = synth.reads_host(g, nchk, seed=4242)
    odb = orc.DB(31, kmers, vals, db.n_values, db.parent_vi)
    orun = orc.MatchRun(odb)
    ocv, ofl = orun.submit(seq, off, threads=16)
    ot, _ = orun.finish()
    odb.close()
    m.reset()
    cv = torch.empty(nchk, dtype=torch.int32, device="cuda")
    fl = torch.empty(nchk, dtype=torch.uint8, device="cuda")
    m.submit(dseq, doff, 0, n_reads=nchk, class_vi=cv, flags=fl)
    gt, _ = m.finish()
    assert np.array_equal(ot, gt)
    assert np.array_equal(cv.cpu().numpy(), ocv) and np.array_equal(fl.cpu().numpy(), ofl)
    assert whole[:, 0].sum() > 0.4 * n
    m.close()
    store.close()
