"""The reference-side binding below the JVM: java/jni/gsgpu_jni.c compiled against the stand-in jni.h with a small FUNCTIONAL JNIEnv
(tests/native/jni_stub/jni_env.c; the image has no JDK) and its file-level entry points -- what GpuFastqKMerMatcher.runMatcher /
GpuFastqBloomFilter.runFilter call when every resource is a local file (C/match/FastqKMerMatcher.java:181-235,
C/bloom/FastqBloomFilter.java:80-89) -- driven exactly as a JVM would drive them: Strings, String[], direct ByteBuffers, long[].
Results must equal the ctypes path over the same C ABI."""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np
import pytest

import genestrip_amd as ga
from genestrip_amd import host, synth
from conftest import ROOT
from oracle import gs_oracle as orc

pytestmark = pytest.mark.gpu

PFX = "Java_org_metagene_genestrip_gpu_GsGpuNative_"


@pytest.fixture(scope="module")
def jni(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("jni") / "libgsjni_test.so")
    ga.lib()
    host.lib()
    cmd = ["gcc", "-shared", "-fPIC", "-Wall", "-I" + os.path.join(ROOT, "tests", "native", "jni_stub"), "-I" + os.path.join(ROOT, "include"), "-o", out,
           os.path.join(ROOT, "java", "jni", "gsgpu_jni.c"), os.path.join(ROOT, "tests", "native", "jni_stub", "jni_env.c"),
           "-L" + os.path.join(ROOT, "genestrip_amd"), "-lgshost", "-lgsgpu", "-Wl,-rpath," + os.path.join(ROOT, "genestrip_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    L = C.CDLL(out)
    vp, i64, i32, u8, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_uint8, C.c_double
    L.stub_env.restype = vp
    L.stub_string.restype, L.stub_string.argtypes = vp, [C.c_char_p]
    L.stub_string_chars.restype, L.stub_string_chars.argtypes = C.c_char_p, [vp]
    L.stub_buffer.restype, L.stub_buffer.argtypes = vp, [vp, i64]
    L.stub_long_array.restype, L.stub_long_array.argtypes = vp, [i32]
    L.stub_long_array_get.restype, L.stub_long_array_get.argtypes = i64, [vp, i32]
    L.stub_int_array.restype, L.stub_int_array.argtypes = vp, [i32]
    L.stub_int_array_set.argtypes = [vp, i32, i32]
    L.stub_object_array.restype, L.stub_object_array.argtypes = vp, [i32]
    L.stub_object_array_set.argtypes = [vp, i32, vp]
    L.stub_take_exception.restype = C.c_char_p
    f = getattr(L, PFX + "hostMatchFiles")
    f.restype, f.argtypes = None, [vp, vp, i64, u8, u8, i32, i32, dbl, dbl, i32, vp, vp, vp, u8, vp, u8, vp, vp, vp, i32, vp]
    f = getattr(L, PFX + "hostMatchRun")
    f.restype, f.argtypes = None, [vp, vp, i64, i64, vp, vp, vp, u8, vp, u8, vp, i32, vp]
    f = getattr(L, PFX + "hostMatchInto")
    f.restype, f.argtypes = None, [vp, vp, i64, i64, vp, vp, vp, vp]
    f = getattr(L, PFX + "hostFilterFiles")
    f.restype, f.argtypes = None, [vp, vp, i64, i32, i32, dbl, vp, vp, vp, u8, vp]
    f = getattr(L, PFX + "hostLastError")
    f.restype, f.argtypes = vp, [vp, vp]
    return L


def _strings(L, items):
    arr = L.stub_object_array(len(items))
    for i, s in enumerate(items):
        L.stub_object_array_set(arr, i, None if s is None else L.stub_string(s.encode() if isinstance(s, str) else s))
    return arr


def _str(L, s):
    return None if s is None else L.stub_string(str(s).encode())


def _buf(L, a):
    return L.stub_buffer(a.ctypes.data_as(C.c_void_p), a.nbytes)


@pytest.fixture(scope="module")
def sdb():
    return synth.SynthDB(k=31, genera=3, species_per_genus=3, genome_len=20000, seed=11)


def _files(sdb, tmp_path, n=9000):
    seq, off = synth.reads_host(sdb.genomes, n, read_len=150, seed=23)
    recs = [b"@read%d lane=%d\n%s\n+\n%s\n" % (i, i % 5, seq[int(off[i]):int(off[i + 1])].tobytes(), b"F" * 150) for i in range(n)]
    p1, p2 = str(tmp_path / "a.fastq.gz"), str(tmp_path / "b.fastq")
    open(p1, "wb").write(gzip.compress(b"".join(recs[:5000]), compresslevel=1, mtime=0))
    open(p2, "wb").write(b"".join(recs[5000:]))
    return seq, off, [p1, p2]


def test_run_matcher_over_local_files_through_the_jni_shim(jni, sdb, tmp_path):
    L = jni
    seq, off, paths = _files(sdb, tmp_path)
    store = ga.DeviceKMerStore(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi)
    nv = sdb.n_values
    # the ctypes path over the same C ABI
    f0, k0 = str(tmp_path / "flt0.fastq.gz"), str(tmp_path / "kr0.txt")
    want, wantd, wtot, wdesc = host.match_files(store, paths, filtered_path=f0, kraken_out_path=k0, taxids=sdb.taxids, max_contig_desc=True)
    # hostMatchFiles as GpuFastqKMerMatcher would call it
    table, dtable = np.zeros((nv, ga.N_COLS), dtype=np.int64), np.zeros((nv, ga.N_DCOLS), dtype=np.float64)
    descs = np.zeros((nv, 256), dtype=np.uint8)
    totals = L.stub_long_array(4)
    f1, k1 = str(tmp_path / "flt1.fastq.gz"), str(tmp_path / "kr1.txt")
    getattr(L, PFX + "hostMatchFiles")(L.stub_env(), None, store.h.value, 1, 1, 10, 1, -1.0, -1.0, 0, _strings(L, paths), _str(L, f1), _str(L, k1), 1,
                                       _strings(L, sdb.taxids), 0, _buf(L, table), _buf(L, dtable), _buf(L, descs), 256, totals)
    assert L.stub_take_exception() is None
    assert np.array_equal(table, want)
    assert [L.stub_long_array_get(totals, i) for i in range(4)] == [wtot.reads, wtot.kmers, wtot.bps, wtot.filtered_reads] and wtot.reads == 9000
    assert gzip.decompress(open(f1, "rb").read()) == gzip.decompress(open(f0, "rb").read()) and open(k1, "rb").read() == open(k0, "rb").read()
    names = [bytes(r).split(b"\0", 1)[0] for r in descs]
    assert names == wdesc
    # ... and the names are those of the reads the table points at (CountsPerTaxid.maxContigDescriptor: behind '@', up to the first blank)
    for v in range(nv):
        r = int(table[v, ga.C_MAX_CONTIG_READ_NO]) if hasattr(ga, "C_MAX_CONTIG_READ_NO") else int(table[v, 9])
        assert names[v] == (b"read%d" % r if r >= 0 else b"")
    # the oracle agrees with both
    orun = orc.MatchRun(orc.DB(31, sdb.kmers, sdb.value_idx, sdb.n_values, sdb.parent_vi))
    orun.submit(seq, off)
    otable, _ = orun.finish()
    assert np.array_equal(table, otable)
    # hostMatchRun: into the matcher's own run (matchReset before, matchFinish after)
    m = ga.FastqKMerMatcher(store)
    descs2 = np.zeros((nv, 256), dtype=np.uint8)
    f2 = str(tmp_path / "flt2.fastq.gz")
    getattr(L, PFX + "hostMatchRun")(L.stub_env(), None, m.h.value, store.h.value, _strings(L, paths), _str(L, f2), None, 1, _strings(L, sdb.taxids), 0,
                                     _buf(L, descs2), 256, totals)
    assert L.stub_take_exception() is None
    t2, _ = m.finish()
    assert np.array_equal(t2, want) and np.array_equal(descs2, descs)
    assert gzip.decompress(open(f2, "rb").read()) == gzip.decompress(open(f0, "rb").read())
    # hostMatchInto: the files as files 0 and 1 of a sample
    m.reset()
    idx = L.stub_int_array(2)
    L.stub_int_array_set(idx, 1, 1)
    rof = L.stub_long_array(2)
    getattr(L, PFX + "hostMatchInto")(L.stub_env(), None, m.h.value, store.h.value, _strings(L, paths), idx, rof, totals)
    assert L.stub_take_exception() is None
    assert [L.stub_long_array_get(rof, i) for i in range(2)] == [5000, 4000]
    t3, _ = m.finish()
    col = 9  # GS_C_MAX_CONTIG_READ_NO: (file << 32 | read) here
    conv = t3.copy()
    x = conv[:, col]
    conv[:, col] = np.where(x >= 0, (x >> 32) * 5000 + (x & 0xffffffff), x)
    assert np.array_equal(conv, want)
    # a failure surfaces as the RuntimeException's message
    getattr(L, PFX + "hostMatchRun")(L.stub_env(), None, m.h.value, store.h.value, _strings(L, [str(tmp_path / "missing.fastq")]), None, None, 1, None, 0, None, 0, totals)
    msg = L.stub_take_exception()
    assert msg is not None and b"missing.fastq" in msg
    err = getattr(L, PFX + "hostLastError")(L.stub_env(), None)
    assert b"missing.fastq" in L.stub_string_chars(err)
    m.close()
    store.close()


def test_run_filter_over_local_files_through_the_jni_shim(jni, sdb, tmp_path):
    L = jni
    _, _, paths = _files(sdb, tmp_path)
    keys = sdb.kmers[np.isin(sdb.value_idx, sdb.species_vi[:4])]
    ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
    ob.put_many(keys)
    gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
    a0, r0 = str(tmp_path / "acc0.fastq.gz"), str(tmp_path / "rest0.fastq.gz")
    wtot = host.filter_files(gb, 31, paths, filtered_path=a0, rest_path=r0)
    a1, r1 = str(tmp_path / "acc1.fastq.gz"), str(tmp_path / "rest1.fastq.gz")
    totals = L.stub_long_array(4)
    getattr(L, PFX + "hostFilterFiles")(L.stub_env(), None, gb.h.value, 31, 1, 0.2, _strings(L, paths), _str(L, a1), _str(L, r1), 0, totals)
    assert L.stub_take_exception() is None
    assert [L.stub_long_array_get(totals, i) for i in range(4)] == [wtot.reads, wtot.kmers, wtot.bps, wtot.filtered_reads]
    assert 0 < wtot.filtered_reads < 9000
    for x, y in ((a0, a1), (r0, r1)):
        assert gzip.decompress(open(x, "rb").read()) == gzip.decompress(open(y, "rb").read())
    gb.close()
