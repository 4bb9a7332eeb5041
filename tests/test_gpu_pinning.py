"""Round-2 pinning gaps (VERDICT r01 "What's missing" 1-2, "What's weak" 7, ADVICE): the RadixKMerStore layout (SURVEY 8a
row a4), maxClassificationPaths up to 128 (C/GSConfigKey.java:350), the serial wrap of the long-read kernel and damaged
store files.  Needs an MI355X: run with -m gpu."""
import os
import struct

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


def _wide_tree(n_values, fan):
    parent = np.empty(n_values, dtype=np.int32)
    parent[0] = -1
    parent[1:fan + 1] = 0
    parent[fan + 1:] = 1 + (np.arange(n_values - fan - 1) % fan)
    return parent


def _mixed_reads(genomes, n, seed):
    """150 bp reads plus a tail of long ones (gs_match_long_kernel)"""
    seq, off = synth.reads_host(genomes, n, read_len=150, seed=seed)
    lseq, loff = synth.reads_host(genomes, n // 20, read_len=1400, seed=seed + 1)
    return np.concatenate([seq, lseq]), np.concatenate([off, off[-1] + loff[1:]]).astype(np.uint64)


@pytest.mark.parametrize("n_values,radix_bits", [(2500, 17), (70000, 17), (262144, 16), (500000, 17)])
def test_store_filled_in_radix_visit_order(n_values, radix_bits):
    """a4: a Java host with `useRadixStore` hands gs_db_create the (kmer, valueIndex) stream of RadixKMerStore.visit
    (:714-729) -- bucket by bucket, not ascending -- and up to 2^(2+radixBits) values (:160-164).  The device table
    must equal the oracle's radix getLong (:369-412) and, through it, the sorted layout."""
    db = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=40000, seed=5)
    rng = np.random.default_rng(n_values)
    parent = _wide_tree(n_values, 300)
    vals = rng.integers(0, n_values, len(db.kmers)).astype(np.int32)
    perm = rng.permutation(len(db.kmers))
    rdb = orc.DB(31, db.kmers[perm], vals[perm], n_values, parent, True, radix_bits=radix_bits)
    vk, vv = rdb.visit()
    assert not np.all(np.diff(vk) > 0)
    seq, off = _mixed_reads(db.genomes, 6000, 9)
    cfg = dict(threshold=2, max_kmer_res_counts=3)
    res = []
    for odb in (rdb, orc.DB(31, db.kmers, vals, n_values, parent, True)):
        run = orc.MatchRun(odb, **cfg)
        cv, fl = run.submit(seq, off, threads=8)
        mc = run.max_counts()
        res.append((run.finish()[0], cv, fl, mc))
    store = ga.DeviceKMerStore(31, vk, vv, n_values, parent)
    assert store.info.n_stored == len(vk)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(**cfg))
    gcv, gfl = m.match_reads(seq, off)
    gmc = m.max_counts()
    gt, _ = m.finish()
    m.close()
    store.close()
    assert res[0][0][:, orc.C_KMERS].sum() > 300000
    for want in res:
        assert np.array_equal(gt, want[0]) and np.array_equal(gcv, want[1]) and np.array_equal(gfl, want[2])
        assert np.array_equal(gmc, want[3])


def test_duplicate_kmers_are_refused_in_any_order():
    with pytest.raises(ga.GsError) as e:
        ga.DeviceKMerStore(31, np.array([9, 5, 9], dtype=np.int64), np.array([0, 0, 0], dtype=np.int32), 1, None)
    assert e.value.code == -1


@pytest.mark.parametrize("max_paths,threshold,fan", [(128, 1, 10), (128, 3, 10), (100, 2, 40), (65, 1, 4), (64, 1, 10), (128, 1, 1)])
def test_up_to_128_classification_paths(max_paths, threshold, fan):
    """every k-mer of the genomes maps to one of 400 leaves, so a 150 bp read walks ~60-120 unrelated leaves and a long
    read many more: the candidate list really fills up beyond 64 entries, and paths beyond max_paths are dropped
    silently (FastqKMerMatcher.java:568-586)"""
    db = synth.SynthDB(k=31, genera=2, species_per_genus=2, genome_len=30000, seed=21)
    rng = np.random.default_rng(max_paths * 7 + threshold)
    n_values = 1 + fan + 400
    parent = _wide_tree(n_values, fan)
    # runs of 1-3 consecutive genome positions share a leaf, so that counts differ and ties are partial
    vals = (1 + fan + (np.cumsum(rng.integers(0, 3, len(db.kmers)) == 0) * 7919) % 400).astype(np.int32)
    seq, off = _mixed_reads(db.genomes, 3000, 33)
    cfg = dict(max_paths=max_paths, threshold=threshold, max_read_class_err=0.95)
    orun = orc.MatchRun(orc.DB(31, db.kmers, vals, n_values, parent), **cfg)
    ocv, ofl = orun.submit(seq, off, threads=8)
    ot, _ = orun.finish()
    store = ga.DeviceKMerStore(31, db.kmers, vals, n_values, parent)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(**cfg))
    gcv, gfl = m.match_reads(seq, off)
    gt, _ = m.finish()
    m.close()
    if max_paths == 128:  # the limit matters: 64 paths give another classification for some reads
        m64 = ga.FastqKMerMatcher(store, ga.MatchConfig(**dict(cfg, max_paths=64)))
        cv64, _ = m64.match_reads(seq, off)
        m64.close()
        assert not np.array_equal(cv64, gcv) or fan == 1
    store.close()
    assert int((ocv >= 0).sum()) > 500
    assert np.array_equal(gt, ot), np.argwhere(gt != ot)[:6]
    assert np.array_equal(gcv, ocv) and np.array_equal(gfl, ofl)


def test_max_paths_outside_the_reference_range_is_refused():
    store = ga.DeviceKMerStore(31, np.array([5], dtype=np.int64), np.array([0], dtype=np.int32), 1, None)
    for bad in (0, 129):
        with pytest.raises(ga.GsError):
            ga.FastqKMerMatcher(store, ga.MatchConfig(max_paths=bad))
    store.close()


def test_long_read_serial_wrap(monkeypatch):
    """gs_match_long_kernel tags its per-wave scratch rows with a running serial; when the serial wraps, stale tags could
    pass for the new read's.  GS_TEST_LONG_SERIAL starts the serials three reads below the wrap and pre-fills the rows
    with the values the serials take right after it."""
    monkeypatch.setenv("GS_TEST_LONG_SERIAL", "0xFFFFFFFD")
    monkeypatch.setenv("GS_LONG_BLOCKS_PER_CU", "1")  # 1024 waves for 6000 long reads: every wave gets past the wrap
    db = synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)
    seq, off = synth.reads_host(db.genomes, 6000, read_len=900, seed=3)
    off = off.astype(np.uint64)
    # (more than GS_NV_LDS = 128 value indices: smaller taxonomies keep the rows in LDS, fresh with every launch)
    n_values = 300
    parent = np.zeros(n_values, dtype=np.int32)
    parent[:db.n_values] = db.parent_vi
    vidx = db.value_idx.copy()
    moved = np.flatnonzero(vidx == db.species_vi[0])[::3]
    vidx[moved] = 150 + (np.arange(len(moved)) % 150)  # leaves under the root that share reads with a species
    for cfg in (dict(), dict(threshold=3, max_paths=128)):
        orun = orc.MatchRun(orc.DB(31, db.kmers, vidx, n_values, parent), **cfg)
        ocv, ofl = orun.submit(seq, off, threads=8)
        ot, _ = orun.finish()
        store = ga.DeviceKMerStore(31, db.kmers, vidx, n_values, parent)
        m = ga.FastqKMerMatcher(store, ga.MatchConfig(**cfg))
        gcv, gfl = m.match_reads(seq, off)
        gt, _ = m.finish()
        m.close()
        store.close()
        assert np.array_equal(gt, ot) and np.array_equal(gcv, ocv) and np.array_equal(gfl, ofl)


def test_damaged_store_files_are_refused(tmp_path):
    """gs_db_load checks the header against the payload before anything reaches HBM: a stale or damaged image must end
    in GS_E_INVALID, not in out-of-range tree / value indices inside a kernel"""
    db = synth.SynthDB(k=31, genera=2, species_per_genus=3, genome_len=8000, seed=2)  # 9 values: 4 value bits
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    good = tmp_path / "good.gss"
    store.save(good)
    info = store.info
    store.close()
    raw = bytearray(good.read_bytes())
    ga.DeviceKMerStore.load(good).close()
    hdr = 8 + 96 + 8 + 8 + 8 + 8 + 8  # magic | gs_db_info | bucket_bits, vbits | gate_words | mgate_words | rec_buckets | checksum
    assert info.rec_bytes > 0 and info.n_in_records > 0.5 * info.n_stored
    assert len(raw) == hdr + info.table_bytes + info.gate_bytes + info.mgate_bytes + info.rec_bytes + 16 * info.n_values

    def refused(name, data):
        p = tmp_path / name
        p.write_bytes(bytes(data))
        with pytest.raises(ga.GsError) as e:
            ga.DeviceKMerStore.load(p)
        assert e.value.code == -1, name
        return str(e.value)

    refused("short.gss", raw[:len(raw) - 4])
    refused("long.gss", raw + b"\0" * 8)
    refused("magic.gss", b"GSSTORE5" + raw[8:])
    flip = bytearray(raw)
    flip[hdr + info.table_bytes // 2] ^= 0x10
    assert "checksum" in refused("flip.gss", flip)
    tree_at = len(raw) - 16 * info.n_values

    def with_checksum(data):
        """a DELIBERATE bad image: the checksum is made to fit, so only the semantic checks can catch it"""
        w = np.frombuffer(bytes(data[hdr:]), dtype=np.uint32).astype(np.uint64)
        a = (np.uint64(1) + np.cumsum(w, dtype=np.uint64))
        b = np.cumsum(a, dtype=np.uint64)[-1]
        cs = int(a[-1]) ^ ((int(b) << 1) & (2**64 - 1)) ^ (int(b) >> 63)
        out = bytearray(data)
        out[hdr - 8:hdr] = struct.pack("<Q", cs)
        return out

    assert with_checksum(raw) == raw  # the helper reproduces the library's checksum
    bad = bytearray(raw)
    bad[tree_at + 4:tree_at + 8] = struct.pack("<i", 10_000)  # parent of value 1 far outside the tree
    assert "tree" in refused("parent.gss", with_checksum(bad))
    bad = bytearray(raw)
    bad[8:12] = struct.pack("<i", 40)  # k
    assert "k outside" in refused("k.gss", bad)
    rec_at = hdr + info.table_bytes + info.gate_bytes + info.mgate_bytes
    rfirst = next(i for i in range(rec_at, rec_at + info.rec_bytes, 64) if any(raw[i + 8:i + 16]))
    bad = bytearray(raw)
    bad[rfirst + 16:rfirst + 64] = b"\xff" * 48  # every value of the record beyond n_values
    assert "records: value" in refused("recvalue.gss", with_checksum(bad))
    bad = bytearray(raw)
    bad[rfirst + 7] |= 0x80  # a seen bit
    assert "seen" in refused("recseen.gss", with_checksum(bad))
    bad = bytearray(raw)
    first = next((i for i in range(hdr, hdr + info.table_bytes, 8) if any(raw[i:i + 8])), None)
    assert first is not None  # some k-mers lose their record bucket and live in the table
    slot = struct.unpack("<Q", raw[first:first + 8])[0]
    vmask = (1 << info.value_bits) - 1
    slot = (slot & ~(vmask << 1)) | (vmask << 1)  # value index beyond n_values
    bad[first:first + 8] = struct.pack("<Q", slot)
    assert "slot value" in refused("value.gss", with_checksum(bad))


def test_device_and_host_layout_builds_are_reproducible_and_answer_alike(tmp_path, monkeypatch):
    """gs_db_create lays the store out on the device (gs_layout_build.hip); GS_BUILD_HOST=1 keeps the host builder.  Each of them
    gives the same bytes every time (the merge of runs on separately built replicas indexes the unique-k-mer bitmap by slot),
    and both layouts answer every read alike."""
    db = synth.SynthDB(k=31, genera=3, species_per_genus=4, genome_len=30000, seed=21)
    seq, off = synth.reads_host(db.genomes, 6000, read_len=150, seed=3)
    off = off.astype(np.uint64)
    images, tables = {}, {}
    for how in ("device", "device", "host", "host"):
        if how == "host":
            monkeypatch.setenv("GS_BUILD_HOST", "1")
        else:
            monkeypatch.delenv("GS_BUILD_HOST", raising=False)
        store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
        p = tmp_path / ("%s_%d.gss" % (how, len(images)))
        store.save(p)
        m = ga.FastqKMerMatcher(store)
        m.submit(seq, off, 0)
        t = m.finish()[0]
        m.close()
        store.close()
        raw = p.read_bytes()
        if how in images:
            assert images[how] == raw, how  # the same bytes again
            assert np.array_equal(tables[how], t)
        images[how], tables[how] = raw, t
    assert images["device"] != images["host"]  # two builders, two (valid) layouts
    assert np.array_equal(tables["device"], tables["host"]) and tables["device"][:, 3].sum() > 0
    ga.DeviceKMerStore.load(tmp_path / "device_0.gss").close()  # and the loader's checks accept the device's image


@pytest.mark.parametrize("seed", range(12))
def test_device_layout_builder_on_awkward_stores(seed, monkeypatch):
    """stores that stress the layout rules -- the same 15-mers in many contexts (more than two windows per minimizer, more than
    eight per minimizer: the clustering cap), low-complexity sequence, tiny stores, every k from 19 to 31 -- built by the device
    and by the host builder: reads from the same material must get identical tables (and the oracle's)"""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(19, 32))
    n_values = int(rng.integers(1, 40))
    parent = np.array([-1] + [int(rng.integers(0, i)) for i in range(1, n_values)], dtype=np.int32)
    core = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 400)
    genomes = []
    for g in range(int(rng.integers(1, 30))):
        L = int(rng.integers(k, 3000))
        body = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L)
        if L > 500 and g % 2 == 0:  # the same core in many different surroundings: one minimizer, many windows
            for _ in range(int(rng.integers(1, 12))):
                a = int(rng.integers(0, L - 120))
                body[a:a + 100] = core[:100]
                body[a + 40 + int(rng.integers(0, 20))] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8))
        if g % 5 == 1:
            body[:min(L, 200)] = np.frombuffer((b"AT" * 100)[:min(L, 200)], dtype=np.uint8)  # low complexity
        genomes.append(body)
    keys, vals = {}, {}
    for gi, body in enumerate(genomes):
        for x in orc.canonical_kmers(body.tobytes().decode(), k).tolist():
            keys.setdefault(int(x), int(rng.integers(0, n_values)))
    kk = np.array(sorted(keys), dtype=np.int64)
    vv = np.array([keys[int(x)] for x in kk], dtype=np.int32)
    reads = []
    for body in genomes:
        for _ in range(6):
            a = int(rng.integers(0, max(1, len(body) - 60)))
            r = body[a:a + int(rng.integers(k, 260))].copy()
            if len(r) > 5 and rng.random() < 0.3:
                r[int(rng.integers(0, len(r)))] = ord("N")
            reads.append(r.tobytes() if rng.random() < 0.5 else r.tobytes()[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA")))
    reads += [rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 150).tobytes() for _ in range(20)]
    seq, off = orc.pack_reads(reads)
    odb = orc.DB(k, kk, vv, n_values, parent)
    run = orc.MatchRun(odb)
    wcv, wfl = run.submit(seq, off, threads=4)
    want = run.finish()[0]
    odb.close()
    for how in ("device", "host"):
        if how == "host":
            monkeypatch.setenv("GS_BUILD_HOST", "1")
        else:
            monkeypatch.delenv("GS_BUILD_HOST", raising=False)
        store = ga.DeviceKMerStore(k, kk, vv, n_values, parent)
        assert store.info.n_stored == len(kk), how
        m = ga.FastqKMerMatcher(store)
        cv, fl = m.match_reads(seq, off, 0)
        t = m.finish()[0]
        assert np.array_equal(t, want), (how, k, len(kk), np.argwhere(t != want)[:5])
        assert np.array_equal(cv, wcv) and np.array_equal(fl, wfl), how
        m.close()
        store.close()
