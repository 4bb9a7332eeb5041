"""Device store: native file round trip, degenerate stores, small k, ABI error behaviour."""
import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import synth
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def test_store_file_round_trip(sdb, tmp_path):
    seq, off = synth.reads_host(sdb.genomes, 5000, read_len=150, seed=41)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m = ga.FastqKMerMatcher(store)
    cv, fl = m.match_reads(seq, off)
    want, _ = m.finish()
    with pytest.raises(ga.GsError) as e:  # seen bits are set while a unique-counting run is alive
        store.save(tmp_path / "busy.gss")
    assert e.value.code == -5
    m.close()
    path = tmp_path / "store.gss"
    store.save(path)
    i0 = store.info
    store.close()
    loaded = ga.DeviceKMerStore.load(path)
    i1 = loaded.info
    assert (i1.k, i1.n_values, i1.n_stored, i1.n_buckets, i1.gate_bytes) == (i0.k, i0.n_values, i0.n_stored, i0.n_buckets, i0.gate_bytes)
    m2 = ga.FastqKMerMatcher(loaded)
    cv2, fl2 = m2.match_reads(seq, off)
    got, _ = m2.finish()
    assert np.array_equal(want, got) and np.array_equal(cv, cv2) and np.array_equal(fl, fl2)
    m2.close()
    loaded.close()
    (tmp_path / "junk.gss").write_bytes(b"not a store")
    with pytest.raises(ga.GsError):
        ga.DeviceKMerStore.load(tmp_path / "junk.gss")


def test_second_unique_run_on_same_store_is_refused(sdb):
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    m1 = ga.FastqKMerMatcher(store)
    with pytest.raises(ga.GsError) as e:
        ga.FastqKMerMatcher(store)
    assert e.value.code == -5
    m3 = ga.FastqKMerMatcher(store, ga.MatchConfig(count_unique=False))  # fine: does not use the seen bits
    m3.close()
    m1.close()
    m4 = ga.FastqKMerMatcher(store)  # the owner is released on destroy
    m4.close()
    store.close()


def test_empty_and_tiny_stores():
    reads = [b"ACGTACGTACGTACGTACGTACGTACGTACGTACGT", b"", b"NNNN"]
    seq, off = orc.pack_reads(reads)
    store = ga.DeviceKMerStore(31, np.zeros(0, np.int64), np.zeros(0, np.int32), 1, np.array([-1], np.int32))
    m = ga.FastqKMerMatcher(store)
    cv, fl = m.match_reads(seq, off)
    t, _ = m.finish()
    assert not t.any() or t[0].tolist() == [0, 0, 0, 0, 0, 0, 0, 0, 0, -1]
    assert cv.tolist() == [-1, -1, -1] and fl.tolist() == [0, 0, 0]
    m.close()
    store.close()
    # a single-entry store
    key = orc.kmer_canonical(reads[0][:31])
    store = ga.DeviceKMerStore(31, [key], [0], 1, np.array([-1], np.int32))
    m = ga.FastqKMerMatcher(store)
    cv, fl = m.match_reads(seq, off)
    t, _ = m.finish()
    o = orc.MatchRun(orc.DB(31, [key], [0], 1, np.array([-1], np.int32)))
    ocv, ofl = o.submit(seq, off)
    ot, _ = o.finish()
    assert np.array_equal(t, ot) and np.array_equal(cv, ocv) and np.array_equal(fl, ofl) and t[0, 2] >= 1
    m.close()
    store.close()


@pytest.mark.parametrize("k", [5, 15, 16, 24, 30])
def test_small_k_against_oracle(k):
    rng = np.random.default_rng(k)
    genomes = [bytes(rng.choice(list(b"ACGT"), 3000).tolist()) for _ in range(4)]
    d = {}
    for vi, g in enumerate(genomes):
        for x in orc.canonical_kmers(g, k).tolist():
            d[x] = vi + 1 if x not in d else 0  # shared k-mers go to the root
    keys = np.array(sorted(d), dtype=np.int64)
    vidx = np.array([d[x] for x in keys.tolist()], dtype=np.int32)
    parent = np.array([-1, 0, 0, 0, 0], dtype=np.int32)
    reads = []
    for _ in range(600):
        g = genomes[int(rng.integers(0, 4))]
        L = int(rng.integers(1, 400))
        p = int(rng.integers(0, len(g) - L))
        r = bytearray(g[p:p + L])
        for _ in range(int(rng.integers(0, 4))):
            r[int(rng.integers(0, L))] = rng.choice(list(b"ACGTN"))
        reads.append(bytes(r))
    seq, off = orc.pack_reads(reads)
    store = ga.DeviceKMerStore(k, keys, vidx, 5, parent)
    m = ga.FastqKMerMatcher(store)
    cv, fl = m.match_reads(seq, off)
    t, _ = m.finish()
    o = orc.MatchRun(orc.DB(k, keys, vidx, 5, parent))
    ocv, ofl = o.submit(seq, off)
    ot, _ = o.finish()
    assert np.array_equal(t, ot), np.argwhere(t != ot)[:5]
    assert np.array_equal(cv, ocv) and np.array_equal(fl, ofl)
    m.close()
    store.close()


def test_argument_errors():
    with pytest.raises(ga.GsError) as e:
        ga.DeviceKMerStore(31, [5, 5], [0, 0], 1)  # the same k-mer twice
    assert e.value.code == -1
    with pytest.raises(ga.GsError):
        ga.DeviceKMerStore(32, [5], [0], 1)
    with pytest.raises(ga.GsError):
        ga.DeviceKMerStore(31, [5], [3], 2)  # value index out of range
    with pytest.raises(ga.GsError):
        ga.DeviceKMerStore(31, [5], [0], 2, np.array([1, 0], np.int32))  # cycle
    store = ga.DeviceKMerStore(31, [5], [0], 1)
    with pytest.raises(ga.GsError):
        ga.FastqKMerMatcher(store, ga.MatchConfig(max_paths=129))  # the reference allows 1..128
    store.close()
