"""ctypes front end of the CPU oracle (oracle/libgsoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package ``genestrip_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

N_COLS = 10
N_DCOLS = 4
(C_READS, C_READS_KMERS, C_KMERS, C_UNIQUE_KMERS, C_CONTIGS, C_CONTIG_LEN_SQ_SUM, C_MAX_CONTIG_LEN,
 C_READS_1KMER, C_READS_BPS, C_MAX_CONTIG_READ_NO) = range(N_COLS)
F_FOUND, F_RETURNED, F_COUNTED = 1, 2, 4
BLOOM_XOR, BLOOM_MURMUR, BLOOM_BLOCKED = 0, 1, 2


class MatchCfg(C.Structure):
    _fields_ = [("classify", C.c_int32), ("count_unique", C.c_int32), ("max_paths", C.c_int32),
                ("threshold", C.c_int32), ("max_read_tax_err", C.c_double), ("max_read_class_err", C.c_double),
                ("max_kmer_res_counts", C.c_int32), ("pad", C.c_int32)]


class _Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("seq", C.c_void_p), ("seq_off", C.c_void_p), ("desc", C.c_void_p),
                ("desc_off", C.c_void_p), ("qual", C.c_void_p), ("qual_off", C.c_void_p),
                ("total_kmers", C.c_int64), ("total_bps", C.c_int64)]


def build():
    """(Re)build oracle/libgsoracle.so with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libgsoracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    sig = {
        "orc_jrandom_init": (None, [vp, i64]), "orc_jrandom_next_long": (i64, [vp]),
        "orc_jrandom_next_int": (i32, [vp, i32]),
        "orc_kmer_straight": (i64, [vp, C.c_int, C.c_int, vp]), "orc_kmer_reverse": (i64, [vp, C.c_int, C.c_int, vp]),
        "orc_next_straight": (i64, [i64, C.c_uint8, C.c_int]), "orc_next_reverse": (i64, [i64, C.c_uint8, C.c_int]),
        "orc_standard_kmer": (i64, [i64, i64]), "orc_kmer_canonical": (i64, [vp, C.c_int, C.c_int, vp]),
        "orc_bloom_create": (vp, [C.c_int, i64, dbl]), "orc_bloom_destroy": (None, [vp]),
        "orc_bloom_put": (None, [vp, i64]), "orc_bloom_contains": (C.c_int, [vp, i64]),
        "orc_bloom_put_many": (None, [vp, vp, i64]), "orc_bloom_put_many_mt": (None, [vp, vp, i64, C.c_int]),
        "orc_bloom_kind": (C.c_int, [vp]), "orc_bloom_bits": (i64, [vp]), "orc_bloom_hashes": (i32, [vp]),
        "orc_bloom_hash_factors": (vp, [vp]), "orc_bloom_words": (vp, [vp]), "orc_bloom_n_words": (i64, [vp]),
        "orc_murmur_hash64": (i64, [i64, i64]),
        "orc_filter_accept_read": (C.c_int, [vp, C.c_int, C.c_int, dbl, vp, C.c_int]),
        "orc_filter_batch": (None, [vp, C.c_int, C.c_int, dbl, vp, vp, i64, vp, C.c_int]),
        "orc_db_create": (vp, [C.c_int, i64, vp, vp, i32, vp, C.c_int]), "orc_db_destroy": (None, [vp]),
        "orc_db_create_radix": (vp, [C.c_int, C.c_int, i64, vp, vp, i32, vp, C.c_int]),
        "orc_radix_max_values": (i32, [C.c_int]), "orc_db_entries": (i64, [vp]), "orc_db_visit": (None, [vp, vp, vp]),
        "orc_db_get": (i32, [vp, i64, vp]), "orc_tree_lca": (i32, [vp, i32, i32]),
        "orc_tree_is_ancestor_of": (C.c_int, [vp, i32, i32]),
        "orc_match_begin": (vp, [vp, vp]),
        "orc_match_submit": (C.c_int, [vp, vp, vp, i64, i64, vp, vp, C.c_int]),
        "orc_match_finish": (C.c_int, [vp, vp, vp]), "orc_match_destroy": (None, [vp]),
        "orc_match_bitmap_words": (i64, [vp]), "orc_match_export": (C.c_int, [vp, vp, vp]),
        "orc_match_import": (C.c_int, [vp, vp, vp]), "orc_match_max_counts": (C.c_int, [vp, vp]),
        "orc_match_segments": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int]),
        "orc_parse_fastq": (C.POINTER(_Reads), [vp, C.c_size_t, C.c_int, C.c_int]),
        "orc_reads_free": (None, [C.POINTER(_Reads)]),
        "orc_build_begin": (vp, [C.c_int, i32, vp, C.c_int, C.c_int]),
        "orc_build_begin_dust": (vp, [C.c_int, i32, vp, C.c_int, C.c_int, C.c_int]),
        "orc_dust_passed": (i64, [C.c_int, C.c_int, vp, i64]), "orc_build_fill": (None, [vp, vp, vp, vp, i64]),
        "orc_build_optimize": (i64, [vp]), "orc_build_update": (None, [vp, vp, vp, vp, i64]),
        "orc_build_fetch": (None, [vp, vp, vp]), "orc_build_destroy": (None, [vp]),
        "orc_taxtree_lca": (i32, [i32, vp, i32, i32]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _LIB = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _seq(s):
    if isinstance(s, str):
        s = s.encode()
    return np.frombuffer(bytes(s), dtype=np.uint8)


class JRandom:
    """java.util.Random"""

    def __init__(self, seed):
        self._s = C.c_uint64(0)
        lib().orc_jrandom_init(C.byref(self._s), seed)

    def next_long(self):
        return lib().orc_jrandom_next_long(C.byref(self._s))

    def next_int(self, bound):
        return lib().orc_jrandom_next_int(C.byref(self._s), bound)


def kmer_straight(seq, start, k):
    a = _seq(seq)
    bad = C.c_int(-1)
    v = lib().orc_kmer_straight(_p(a), start, k, C.byref(bad))
    return v, bad.value


def kmer_reverse(seq, start, k):
    a = _seq(seq)
    bad = C.c_int(-1)
    v = lib().orc_kmer_reverse(_p(a), start, k, C.byref(bad))
    return v, bad.value


def kmer_canonical(seq, start=0, k=None):
    a = _seq(seq)
    if k is None:
        k = len(a) - start
    return lib().orc_kmer_canonical(_p(a), start, k, None)


def canonical_kmers(seq, k):
    """all canonical k-mers of an (upper-case ACGT) sequence, invalid windows skipped (numpy, vectorised)"""
    a = _seq(seq)
    lut = np.full(256, -1, dtype=np.int64)
    for ch, v in zip(b"CGAT", range(4)):
        lut[ch] = v
    codes = lut[a]
    n = len(a) - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.int64)
    bad = (codes < 0).astype(np.int64)
    cs = np.concatenate([[0], np.cumsum(bad)])
    ok = (cs[k:] - cs[:-k]) == 0
    c = np.where(codes < 0, 0, codes).astype(np.uint64)
    fwd = np.zeros(n, dtype=np.uint64)
    rev = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        fwd = (fwd << np.uint64(2)) | c[j:j + n]
        rev = rev | ((c[j:j + n] ^ np.uint64(1)) << np.uint64(2 * j))
    can = np.maximum(fwd, rev).astype(np.int64)
    return can[ok]


class Bloom:
    def __init__(self, kind, expected, fpp=0.01):
        self.h = lib().orc_bloom_create(kind, expected, fpp)
        self.kind = kind

    def close(self):
        if self.h:
            lib().orc_bloom_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the module globals may be gone already
            pass

    def put(self, key):
        lib().orc_bloom_put(self.h, int(key))

    def put_many(self, keys, threads=1):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        lib().orc_bloom_put_many_mt(self.h, _p(keys), len(keys), threads)

    def contains(self, key):
        return bool(lib().orc_bloom_contains(self.h, int(key)))

    @property
    def bits(self):
        return lib().orc_bloom_bits(self.h)

    @property
    def hashes(self):
        return lib().orc_bloom_hashes(self.h)

    @property
    def hash_factors(self):
        n = max(1, self.hashes)
        ptr = lib().orc_bloom_hash_factors(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(n,)).copy()

    @property
    def words(self):
        n = lib().orc_bloom_n_words(self.h)
        ptr = lib().orc_bloom_words(self.h)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n,)).copy()

    def accept_read(self, k, min_pos_count, positive_ratio, read):
        a = _seq(read)
        return bool(lib().orc_filter_accept_read(self.h, k, min_pos_count, positive_ratio, _p(a), len(a)))

    def filter_batch(self, k, min_pos_count, positive_ratio, seq, offsets, threads=1):
        n = len(offsets) - 1
        out = np.zeros(n, dtype=np.uint8)
        lib().orc_filter_batch(self.h, k, min_pos_count, positive_ratio, _p(seq), _p(offsets), n, _p(out), threads)
        return out


def radix_max_values(radix_bits):
    """RadixKMerStore.maxValuesForRadix (-1: radix_bits outside [16, 30])"""
    return lib().orc_radix_max_values(radix_bits)


class DB:
    """radix_bits = 0: KMerSortedArray layout (kmers ascending); 16..30: RadixKMerStore layout (kmers distinct, any
    order = putLong order)"""

    def __init__(self, k, kmers, value_idx, n_values, parent_vi=None, bloom_gate=False, radix_bits=0):
        kmers = np.ascontiguousarray(kmers, dtype=np.int64)
        value_idx = np.ascontiguousarray(value_idx, dtype=np.int32)
        pv = None if parent_vi is None else np.ascontiguousarray(parent_vi, dtype=np.int32)
        self.k, self.n_values, self.n = k, n_values, len(kmers)
        if radix_bits:
            self.h = lib().orc_db_create_radix(k, radix_bits, len(kmers), _p(kmers), _p(value_idx), n_values, _p(pv),
                                               int(bloom_gate))
            if not self.h:
                raise ValueError("radix_bits outside [16, 30] or more values than the radix store can hold")
        else:
            assert np.all(np.diff(kmers) > 0), "kmers must be sorted ascending and distinct"
            self.h = lib().orc_db_create(k, len(kmers), _p(kmers), _p(value_idx), n_values, _p(pv), int(bloom_gate))

    def visit(self):
        """(kmers, value_idx) in KMerStore.visit order: what a Java host hands to gs_db_create"""
        n = lib().orc_db_entries(self.h)
        km, vi = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int32)
        lib().orc_db_visit(self.h, _p(km), _p(vi))
        return km, vi

    def close(self):
        if self.h:
            lib().orc_db_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the module globals may be gone already
            pass

    def get(self, kmer):
        pos = C.c_int64(-1)
        vi = lib().orc_db_get(self.h, int(kmer), C.byref(pos))
        return vi, pos.value

    def lca(self, a, b):
        return lib().orc_tree_lca(self.h, a, b)

    def segments(self, read, cap=4096):
        a = _seq(read)
        codes = np.zeros(cap, dtype=np.int32)
        lens = np.zeros(cap, dtype=np.int32)
        n = lib().orc_match_segments(self.h, _p(a), len(a), _p(codes), _p(lens), cap)
        return list(zip(codes[:n].tolist(), lens[:n].tolist()))


class MatchRun:
    def __init__(self, db, classify=True, count_unique=True, max_paths=10, threshold=1,
                 max_read_tax_err=-1.0, max_read_class_err=-1.0, max_kmer_res_counts=0):
        self.db = db
        self.cfg = MatchCfg(int(classify), int(count_unique), max_paths, threshold, max_read_tax_err,
                            max_read_class_err, max_kmer_res_counts, 0)
        self.h = lib().orc_match_begin(db.h, C.byref(self.cfg))

    def close(self):
        if self.h:
            lib().orc_match_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the module globals may be gone already
            pass

    def submit(self, seq, offsets, first_read_no=0, threads=1, per_read=True):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        cv = np.full(n, -1, dtype=np.int32) if per_read else None
        fl = np.zeros(n, dtype=np.uint8) if per_read else None
        lib().orc_match_submit(self.h, _p(seq), _p(offsets), n, first_read_no, _p(cv), _p(fl), threads)
        return cv, fl

    def submit_reads(self, reads, first_read_no=0, threads=1):
        seq, off = pack_reads(reads)
        return self.submit(seq, off, first_read_no, threads)

    def finish(self):
        t = np.zeros((self.db.n_values, N_COLS), dtype=np.int64)
        d = np.zeros((self.db.n_values, N_DCOLS), dtype=np.float64)
        lib().orc_match_finish(self.h, _p(t), _p(d))
        return t, d

    def max_counts(self):
        n = self.cfg.max_kmer_res_counts
        out = np.zeros((self.db.n_values + 1, max(n, 1)), dtype=np.int16)
        lib().orc_match_max_counts(self.h, _p(out))
        return out

    def export_state(self):
        """(table int64 [nv, N_COLS], bitmap uint64[]) raw accumulators (multi-rank merge tests)"""
        t = np.zeros((self.db.n_values, N_COLS), dtype=np.int64)
        b = np.zeros(lib().orc_match_bitmap_words(self.h), dtype=np.uint64)
        lib().orc_match_export(self.h, _p(t), _p(b))
        return t, b

    def import_state(self, table, bitmap):
        t = np.ascontiguousarray(table, dtype=np.int64)
        b = np.ascontiguousarray(bitmap, dtype=np.uint64)
        lib().orc_match_import(self.h, _p(t), _p(b))


def pack_reads(reads):
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(b) for b in bs])
    seq = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8)[:-1].copy() if bs else np.zeros(0, np.uint8)
    return seq, off


def parse_fastq(data, fasta=False, k=31):
    """returns dict(seq, seq_off, desc, desc_off, qual, qual_off, n_reads, total_kmers, total_bps)"""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    r = lib().orc_parse_fastq(_p(buf), len(buf), int(fasta), k)
    rr = r.contents
    n = rr.n_reads

    def arr(ptr, ctype, cnt):
        if cnt == 0:
            return np.zeros(0, dtype=np.dtype(ctype))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(cnt,)).copy()
    so = arr(rr.seq_off, C.c_uint64, n + 1)
    do = arr(rr.desc_off, C.c_uint64, n + 1)
    qo = arr(rr.qual_off, C.c_uint64, n + 1)
    out = dict(n_reads=n, total_kmers=rr.total_kmers, total_bps=rr.total_bps, seq_off=so, desc_off=do,
               qual_off=qo, seq=arr(rr.seq, C.c_uint8, int(so[-1])), desc=arr(rr.desc, C.c_uint8, int(do[-1])),
               qual=arr(rr.qual, C.c_uint8, int(qo[-1])))
    lib().orc_reads_free(r)
    return out


def taxtree_lca(parent_vi, a, b):
    """TaxTree.getLowestCommonAncestor over value indices (-1 = null)"""
    pv = np.ascontiguousarray(parent_vi, dtype=np.int32)
    return int(lib().orc_taxtree_lca(len(pv), _p(pv), int(a), int(b)))


def dust_value(kmer):
    """CGATLongBuffer.getDustValue of a buffer of len(kmer) filled with kmer (streaming restatement, probed through the filter)"""
    s = _seq(kmer)
    lo, hi = -1, 1 << 22  # the smallest max_dust that lets the k-mer pass is its score
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if lib().orc_dust_passed(len(s), mid, _p(s), len(s)) == 1:
            hi = mid
        else:
            lo = mid
    return hi


class DbBuild:
    """FillDBGoal + store.optimize + DBGoal, one region after the other (the CPU restatement the device build is checked
    against): fill(regions) ..., optimize(), update(regions) ..., fetch() -> (kmers ascending, value_idx)"""

    def __init__(self, k, n_values, parent_vi, lower_case_bases=True, step_size=1, max_dust=-1):
        pv = np.ascontiguousarray(parent_vi, dtype=np.int32)
        self.h = C.c_void_p(lib().orc_build_begin_dust(k, n_values, _p(pv), int(lower_case_bases), step_size, max_dust))
        self.n = 0

    def _regions(self, seq, offsets, node_vi):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        if len(seq) == 0:
            seq = np.zeros(1, dtype=np.uint8)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        nv = np.ascontiguousarray(node_vi, dtype=np.int32)
        assert len(nv) == len(off) - 1
        return seq, off, nv

    def fill(self, seq, offsets, node_vi):
        seq, off, nv = self._regions(seq, offsets, node_vi)
        lib().orc_build_fill(self.h, _p(seq), _p(off), _p(nv), len(nv))

    def optimize(self):
        self.n = int(lib().orc_build_optimize(self.h))
        return self.n

    def update(self, seq, offsets, node_vi):
        seq, off, nv = self._regions(seq, offsets, node_vi)
        lib().orc_build_update(self.h, _p(seq), _p(off), _p(nv), len(nv))

    def fetch(self):
        k = np.zeros(self.n, dtype=np.int64)
        v = np.zeros(self.n, dtype=np.int32)
        if self.n:
            lib().orc_build_fetch(self.h, _p(k), _p(v))
        return k, v

    def close(self):
        if getattr(self, "h", None):
            lib().orc_build_destroy(self.h)
            self.h = None
