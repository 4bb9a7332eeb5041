"""Parity of the HIP match path (through the C ABI) against the CPU oracle: bit-exact integer tables,
per-read class and flags.  Needs an MI355X: run with -m gpu."""
import os

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from conftest import GOLDEN
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu

TREE3 = np.array([-1, 0, 0], dtype=np.int32)


def _k2_arrays(pairs):
    d = {}
    for s, vi in pairs:
        d.setdefault(orc.kmer_canonical(s), vi)
    keys = sorted(d)
    return np.array(keys, dtype=np.int64), np.array([d[x] for x in keys], dtype=np.int32)


def _both(k, kmers, vidx, n_values, parent, reads, first_read_no=0, **cfg):
    """run the same batch through the oracle and the GPU; returns ((table, class, flags) x 2)"""
    seq, off = orc.pack_reads(reads) if not isinstance(reads, tuple) else reads
    odb = orc.DB(k, kmers, vidx, n_values, parent)
    orun = orc.MatchRun(odb, **cfg)
    ocv, ofl = orun.submit(seq, off, first_read_no)
    ot, od = orun.finish()
    store = ga.DeviceKMerStore(k, kmers, vidx, n_values, parent)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(**cfg))
    gcv, gfl = m.match_reads(seq, off, first_read_no)
    gt, gd = m.finish()
    m.close()
    store.close()
    return (ot, ocv, ofl, od), (gt, gcv, gfl, gd)


def _assert_same(o, g):
    ot, ocv, ofl, od = o
    gt, gcv, gfl, gd = g
    bad = np.argwhere(ot != gt)
    assert bad.size == 0, f"table differs at (vi, col) {bad[:10].tolist()}: oracle {ot[tuple(bad[0])]} gpu {gt[tuple(bad[0])]}"
    assert np.array_equal(ocv, gcv), f"class differs at reads {np.flatnonzero(ocv != gcv)[:10]}"
    assert np.array_equal(ofl, gfl), f"flags differ at reads {np.flatnonzero(ofl != gfl)[:10]}"
    assert np.allclose(od, gd, rtol=1e-9, atol=1e-9)  # double sums: order dependent, not part of the contract


# ------------------------------------------------------------------ reference KATs through the C ABI
K2_CASES = [
    (0.0, [("CCCC", 0), ("GAGAGA", -1), ("CCCG", 2), ("AGGGG", 1), ("CCCCCCT", 1)]),
    (1.0, [("CTCCT", 1), ("CTCTCCT", -1), ("TAGGGG", 1), ("TAGGGGT", -1)]),
    (0.5, [("CCA", 0), ("CCAA", -1)]),
    (0.1, [("CC", 0), ("CCA", -1), ("CCAA", -1)]),
    (0.99, [("TTTT", -1), ("CTTT", 1)]),
]


def test_k2_read_classification_table():
    kmers, vidx = _k2_arrays([("CC", 0), ("CT", 1), ("CG", 2)])
    store = ga.DeviceKMerStore(2, kmers, vidx, 3, TREE3)
    for err, cases in K2_CASES:
        m = ga.FastqKMerMatcher(store, ga.MatchConfig(max_paths=4, threshold=1, max_read_tax_err=err))
        seq, off = orc.pack_reads([c[0] for c in cases])
        cv, fl = m.match_reads(seq, off, first_read_no=1)
        assert cv.tolist() == [c[1] for c in cases], (err, cv.tolist())
        m.close()
    store.close()


def test_k3_k4_random_k2_reads():
    kmers, vidx = _k2_arrays([("CC", 0), ("GG", 1), ("TT", 1), ("AG", 2)])
    rnd = orc.JRandom(42)
    reads = [bytes(b"CGAT"[rnd.next_int(4)] for _ in range(500)) for _ in range(200)]
    rnd = orc.JRandom(4242)
    reads += [bytes(b"CGAT"[rnd.next_int(4)] for _ in range(200)) for _ in range(1000)]
    o, g = _both(2, kmers, vidx, 3, None, reads, first_read_no=1, classify=False, max_paths=4,
                 max_read_tax_err=0.0, max_read_class_err=0.0)
    _assert_same(o, g)
    assert np.all(g[0][:, ga.COLS.index("unique kmers")] == 1)


def test_k5_dengue():
    lines = open(os.path.join(GOLDEN, "dengue1", "dengue1.fasta")).read().split("\n")
    genome = "".join(l.strip() for l in lines if not l.startswith(">")).upper()
    keys = np.unique(orc.canonical_kmers(genome, 31))
    rd = orc.parse_fastq(open(os.path.join(GOLDEN, "dengue1", "test.fastq"), "rb").read(), k=31)
    o, g = _both(31, keys, np.zeros(len(keys), np.int32), 1, np.array([-1], np.int32), (rd["seq"], rd["seq_off"]))
    _assert_same(o, g)
    gt = g[0]
    assert gt[0, 2] == 7 and gt[0, 3] == 7 and gt[0, 4] == 1 and gt[0, 6] == 7 and g[1][0] == 0
    # every genome k-mer is found and unique counting is exact
    reads = [genome[i:i + 150] for i in range(0, len(genome) - 150, 37)] + [genome]
    o, g = _both(31, keys, np.zeros(len(keys), np.int32), 1, np.array([-1], np.int32), reads)
    _assert_same(o, g)
    assert g[0][0, 3] == len(keys)


# ------------------------------------------------------------------ synthetic store (config 2 recipe, small)
@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _mutate(seq, off, rng, frac_n=0.02, frac_lower=0.01):
    seq = seq.copy()
    n = len(off) - 1
    for r in rng.choice(n, int(n * frac_n), replace=False):
        L = int(off[r + 1] - off[r])
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(0, L))
            seq[int(off[r]) + p:int(off[r]) + min(L, p + int(rng.integers(1, 4)))] = ord("N")
    for r in rng.choice(n, int(n * frac_lower), replace=False):
        p = int(off[r]) + int(rng.integers(0, int(off[r + 1] - off[r])))
        seq[p] = ord(chr(seq[p]).lower())
    return seq


@pytest.mark.parametrize("cfg", [
    dict(),
    dict(classify=False),
    dict(count_unique=False),
    dict(max_read_tax_err=0.3),
    dict(max_read_tax_err=5.0, max_read_class_err=0.5),
    dict(max_read_class_err=40.0),
    dict(threshold=8),
    dict(threshold=200),
    dict(max_paths=1),
])
def test_synthetic_reads_match_oracle(sdb, cfg):
    seq, off = synth.reads_host(sdb.genomes, 20000, read_len=150, seed=99)
    seq = _mutate(seq, off, np.random.default_rng(5))
    o, g = _both(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, (seq, off), first_read_no=1000, **cfg)
    _assert_same(o, g)
    assert (g[2] & ga.F_FOUND).sum() > 5000


def test_ragged_short_long_and_empty_reads(sdb):
    rng = np.random.default_rng(17)
    g0 = sdb.genomes
    reads = [b"", b"A", b"ACGT" * 7, bytes(g0[0][:30]), bytes(g0[0][:31]), bytes(g0[1][100:132])]
    for _ in range(400):
        s = int(rng.integers(0, g0.shape[0]))
        L = int(rng.integers(20, 700))
        p = int(rng.integers(0, g0.shape[1] - L))
        r = bytearray(g0[s][p:p + L].tobytes())
        for _ in range(int(rng.integers(0, 6))):
            r[int(rng.integers(0, L))] = ord("N") if rng.random() < 0.3 else rng.choice(list(b"ACGT"))
        reads.append(bytes(r))
    # chimeras across species/genera exercise the path merge, ties and the LCA fold
    for _ in range(300):
        parts = []
        for _ in range(int(rng.integers(2, 6))):
            s = int(rng.integers(0, g0.shape[0]))
            L = int(rng.integers(31, 90))
            p = int(rng.integers(0, g0.shape[1] - L))
            parts.append(g0[s][p:p + L].tobytes())
        reads.append(b"".join(parts))
    reads.append(bytes(g0[2][:5000]))  # a long read: 4970 k-mer positions
    reads.append(b"N" * 40 + bytes(g0[2][:200]) + b"NN" + bytes(g0[4][300:640]) + b"N")
    for cfg in (dict(), dict(threshold=3, max_read_tax_err=0.9), dict(classify=False)):
        o, g = _both(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, reads, **cfg)
        _assert_same(o, g)


def test_values_without_tree_node_read_as_misses(sdb):
    parent = sdb.parent_vi.copy()
    leaf = int(sdb.species_vi[0])
    parent[leaf] = -2
    seq, off = synth.reads_host(sdb.genomes, 4000, read_len=150, seed=5)
    o, g = _both(31, sdb.kmers, sdb.value_idx, sdb.n_values, parent, (seq, off))
    _assert_same(o, g)
    assert g[0][leaf].tolist()[:9] == [0] * 9


def test_large_value_count_uses_global_counters(sdb):
    # > GS_NV_LDS value indices: the kernel variant with direct global atomics
    nv = 600
    parent = np.full(nv, 0, dtype=np.int32)
    parent[0] = -1
    parent[1:sdb.n_values] = sdb.parent_vi[1:]
    vidx = sdb.value_idx.copy()
    vidx[::7] = 300 + (np.arange(len(vidx[::7])) % 290)
    seq, off = synth.reads_host(sdb.genomes, 6000, read_len=150, seed=8)
    o, g = _both(31, sdb.kmers, vidx, nv, parent, (seq, off), first_read_no=7)
    _assert_same(o, g)


def test_multiple_submits_reset_and_device_batches(sdb):
    seq, off = synth.reads_host(sdb.genomes, 9000, read_len=150, seed=21)
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off)
    ot, _ = orun.finish()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    for a, b in ((0, 3000), (3000, 3001), (3001, 9000)):  # host batches with rebased offsets
        m.submit(seq, off[a:b + 1].copy(), first_read_no=a, n_reads=b - a)
    gt, _ = m.finish()
    assert np.array_equal(ot, gt)
    m.reset()
    import torch
    dseq = torch.from_numpy(seq).cuda()
    doff = torch.from_numpy(off.astype(np.int64)).cuda()
    dcv = torch.empty(9000, dtype=torch.int32, device="cuda")
    dfl = torch.empty(9000, dtype=torch.uint8, device="cuda")
    m.submit(dseq, doff, 0, dcv, dfl, n_reads=9000)
    gt2, _ = m.finish()
    assert np.array_equal(ot, gt2)
    ocv, ofl = orc.MatchRun(odb).submit(seq, off)
    assert np.array_equal(dcv.cpu().numpy(), ocv) and np.array_equal(dfl.cpu().numpy(), ofl)
    m.close()
    store.close()


def test_full_size_properties():
    """config-2 sized store (about 2 M k-mers); size-independent properties on 2 M reads:
    sharding invariance (two halves == whole), totals, and unique k-mers bounded by the store."""
    import torch
    db = synth.SynthDB()
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    n = 2_000_000
    gen = torch.from_numpy(db.genomes).cuda()
    dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
    doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, seed=4242)
    m = ga.FastqKMerMatcher(store)
    m.submit(dseq, doff, 0, n_reads=n)
    whole, _ = m.finish()
    m.reset()
    h = n // 2
    m.submit(dseq, doff, 0, n_reads=h)
    doff2 = doff[h:].contiguous()
    m.submit(dseq, doff2, h, n_reads=n - h)
    parts, _ = m.finish()
    assert np.array_equal(whole, parts)
    # oracle spot check on the first 20k reads of the same stream
    seq, off = synth.reads_host(db.genomes, 20000, seed=4242)
    assert np.array_equal(dseq[:20000 * 150].cpu().numpy(), seq)
    odb = orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off, threads=8)
    ot, _ = orun.finish()
    m.reset()
    m.submit(dseq, doff, 0, n_reads=20000)
    gt, _ = m.finish()
    assert np.array_equal(ot, gt)
    counts = np.bincount(db.value_idx, minlength=db.n_values)
    assert np.all(whole[:, 3] <= counts) and whole[:, 3].sum() > 0.5 * db.n_entries
    assert whole[:, 0].sum() <= n and whole[:, 7].sum() >= whole[:, 0].sum()
    m.close()
    store.close()


def test_max_kmer_res_counts(sdb):
    """maxKMerResCounts > 0: the largest per-k-mer hit counts per taxid and in total (experimental CSV column)"""
    seq, off = synth.reads_host(sdb.genomes, 12000, read_len=150, seed=77)
    # repeat one read many times so that some counters exceed a Java short and wrap
    hot = seq[:150].copy()
    seq = np.concatenate([seq, np.tile(hot, 40000)])
    off = np.concatenate([off, off[-1] + 150 * np.arange(1, 40001, dtype=np.uint64)])
    o = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi), max_kmer_res_counts=5)
    o.submit(seq, off, threads=4)
    ot, _ = o.finish()
    want = o.max_counts()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(max_kmer_res_counts=5))
    m.submit(seq, off, 0, n_reads=len(off) - 1)
    gt, _ = m.finish()
    got = m.max_counts()
    assert np.array_equal(ot, gt)
    assert np.array_equal(want, got), (want[-1], got[-1])
    assert got[-1, 0] > 0
    m.close()
    m2 = ga.FastqKMerMatcher(store)
    with pytest.raises(ga.GsError):
        m2.max_counts()
    m2.close()
    store.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_async_host_batches_equal_the_synchronous_ones(pinned):
    """gs_match_submit_async: batches queued back to back (two under way, buffer sets taken by turns, a slice with
    offsets[0] != 0, growing batch sizes) give the table and per-read outputs of the oracle"""
    import torch
    sdb = synth.SynthDB(genera=2, species_per_genus=3, genome_len=20_000, seed=7)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    seq, off = synth.reads_host(sdb.genomes, 9000, seed=77)
    off = off.astype(np.uint64)
    run = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    want_cv, want_fl = run.submit(seq, off)
    want_t, _ = run.finish()

    def hold(a):  # page-locked copies are what a C / Java host gets from gs_pinned_alloc
        return torch.from_numpy(a.copy()).pin_memory().numpy() if pinned else a.copy()

    m = ga.FastqKMerMatcher(store)
    cuts = [0, 500, 2500, 2501, 6000, 9000]
    keep, tickets = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == 2501:   # a slice of the big arrays: offsets[0] != 0
            s, o = hold(seq), hold(off[a:b + 1])
        else:
            s, o = hold(seq[int(off[a]):int(off[b])]), hold(off[a:b + 1] - off[a])
        cv, fl = hold(np.full(b - a, -7, np.int32)), hold(np.full(b - a, 99, np.uint8))
        keep.append((s, o, cv, fl))
        tickets.append(m.submit_async(s, o, a, cv, fl))
        if len(tickets) >= 2:
            m.wait(tickets[-2])
            pa, pb = cuts[len(tickets) - 2], cuts[len(tickets) - 1]
            assert np.array_equal(keep[-2][2], want_cv[pa:pb]) and np.array_equal(keep[-2][3], want_fl[pa:pb])
    m.wait(tickets[-1])
    got_t, _ = m.finish()
    assert np.array_equal(got_t, want_t)
    assert np.array_equal(np.concatenate([k[2] for k in keep]), want_cv)
    assert np.array_equal(np.concatenate([k[3] for k in keep]), want_fl)
    with pytest.raises(ga.GsError):
        m.wait(99)
    m.close()
    store.close()


def test_max_contig_reads_captured_per_batch_equal_the_final_answer(sdb):
    """gs_match_max_contig_reads: a host that cannot keep every descriptor asks after each batch which read holds a tax
    id's longest contig; what it captured when the answer fell into the batch it had just submitted must be the read
    gs_match_finish reports (the reference's maxContigDescriptor bookkeeping, FastqKMerMatcher.java:401-407)"""
    seq, off = synth.reads_host(sdb.genomes, 20000, read_len=150, seed=91)
    off = off.astype(np.uint64)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    captured = np.full(sdb.n_values, -1, dtype=np.int64)
    cuts = [0, 1500, 1501, 9000, 9000, 16000, 20000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            m.submit(seq[int(off[a]):int(off[b])], off[a:b + 1] - off[a], a, n_reads=b - a)
        now = m.max_contig_reads()
        inside = (now >= a) & (now < b)
        captured[inside] = now[inside]
    table, _ = m.finish()
    assert np.array_equal(captured, table[:, 9])
    assert (captured >= 0).sum() >= 5
    orun = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    orun.submit(seq, off, threads=4, per_read=False)
    assert np.array_equal(orun.finish()[0], table)
    m.close()
    store.close()


@pytest.mark.parametrize("blocks_per_cu", [None, "1"])
def test_many_long_reads_between_short_ones(sdb, monkeypatch, blocks_per_cu):
    """the long-read queue: gs_match_kernel files reads with more than 128 k-mer positions in chunks of 64 per wave (the rest of
    a wave's last chunk is padding), gs_match_long_kernel draws chunks from a shared cursor.  20 000 reads of 20 .. 3000 bases in
    random order -- far more long reads than waves, so chunks fill up and are reused across two submits of one run --, read by
    read against the oracle; the same with one workgroup per CU (GS_LONG_BLOCKS_PER_CU)"""
    if blocks_per_cu:
        monkeypatch.setenv("GS_LONG_BLOCKS_PER_CU", blocks_per_cu)
    rng = np.random.default_rng(77)
    g0 = sdb.genomes
    reads = []
    for i in range(20000):
        L = int(rng.choice([int(rng.integers(20, 158)), 158, 159, 160, int(rng.integers(161, 700)), int(rng.integers(700, 3000))],
                           p=[0.4, 0.02, 0.02, 0.02, 0.44, 0.1]))
        parts, left = [], L
        while left > 0:  # chimeras: several species in one read -> several distinct nodes per iteration and across iterations
            s = int(rng.integers(0, g0.shape[0]))
            n = min(left, int(rng.integers(40, 900)))
            p = int(rng.integers(0, g0.shape[1] - n))
            parts.append(g0[s][p:p + n].tobytes())
            left -= n
        r = bytearray(b"".join(parts))
        if i % 9 == 0:
            r[int(rng.integers(0, L))] = ord("N")
        reads.append(bytes(r))
    seq, off = orc.pack_reads(reads)
    for cfg in (dict(), dict(threshold=4, max_paths=3)):
        o, g = _both(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi, (seq, off), **cfg)
        _assert_same(o, g)
    # two submits of one run (the queue starts over with each), short-only batch in between
    odb = orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off)
    ot, _ = orun.finish()
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    cut = 9000
    m.match_reads(seq[:int(off[cut])], off[:cut + 1], 0)
    m.match_reads(seq[int(off[cut]):], off[cut:] - off[cut], cut)
    gt, _ = m.finish()
    assert np.array_equal(ot, gt)
    m.close()
    store.close()


def test_fixed_length_batches_without_offsets(sdb):
    """gs_match_submit_fixed: reads of one length back to back, no offsets array -- table, classes and flags as with offsets, for
    short reads (one iteration), reads of 129..256 positions and long ones, from device and from host memory"""
    import torch
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    for L, n in ((150, 20000), (31, 500), (30, 300), (200, 6000), (1000, 1500)):
        seq, off = synth.reads_host(sdb.genomes, n, read_len=L, seed=L)
        seq = seq.copy()
        seq[::977] = ord("N")
        m.reset()
        cv, fl = m.match_reads(seq, off)
        want, _ = m.finish()
        for dev in (False, True):
            m.reset()
            s = torch.from_numpy(seq).cuda() if dev else seq
            c2 = torch.empty(n, dtype=torch.int32, device="cuda") if dev else np.zeros(n, dtype=np.int32)
            f2 = torch.empty(n, dtype=torch.uint8, device="cuda") if dev else np.zeros(n, dtype=np.uint8)
            m.submit_fixed(s, L, n, class_vi=c2, flags=f2)
            got, _ = m.finish()
            assert np.array_equal(got, want), (L, dev)
            assert np.array_equal(c2.cpu().numpy() if dev else c2, cv) and np.array_equal(f2.cpu().numpy() if dev else f2, fl), (L, dev)
    m.close()
    store.close()
