"""ctypes binding of include/gshost.h (libgshost.so): the C++ host layer above the C ABI -- FASTQ/FASTA ingest with
the reference's record semantics, the runMatcher / runFilter file pipelines and the CSV report."""
import ctypes as C
import os

import numpy as np

from . import binding as _b

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _ReadBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("seq", C.c_void_p), ("seq_off", C.c_void_p), ("desc", C.c_void_p),
                ("desc_off", C.c_void_p), ("qual", C.c_void_p), ("qual_off", C.c_void_p), ("first_read_no", C.c_int64)]


class _MatchOpts(C.Structure):
    _fields_ = [("filtered_path", C.c_char_p), ("kraken_out_path", C.c_char_p), ("write_all", C.c_int32),
                ("taxids", C.POINTER(C.c_char_p)), ("batch_reads", C.c_int64), ("with_probs", C.c_int32),
                ("max_contig_desc", C.c_void_p), ("max_contig_desc_stride", C.c_int32)]


class Totals(C.Structure):
    _fields_ = [("reads", C.c_int64), ("kmers", C.c_int64), ("bps", C.c_int64), ("filtered_reads", C.c_int64),
                ("seconds_total", C.c_double), ("seconds_parse", C.c_double), ("seconds_gpu", C.c_double)]


class _TaxInfo(C.Structure):
    _fields_ = [("n_values", C.c_int32), ("parent_vi", C.c_void_p), ("position", C.c_void_p),
                ("taxids", C.POINTER(C.c_char_p)), ("names", C.POINTER(C.c_char_p)), ("ranks", C.POINTER(C.c_char_p)),
                ("db_kmers", C.c_void_p), ("db_kmers_total", C.c_int64), ("max_contig_desc", C.POINTER(C.c_char_p)),
                ("max_kmer_counts", C.c_void_p), ("max_kmer_res_counts", C.c_int32)]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    _b.lib()  # loads libgsgpu.so (and the HIP runtime) first
    path = os.path.join(_HERE, "libgshost.so")
    if not os.path.exists(path):
        raise _b.GsError(-6, f"{path} is missing: build it with `make -C genestrip_amd/csrc`")
    L = C.CDLL(path)
    vp, i64, ci = C.c_void_p, C.c_int64, C.c_int
    sig = {
        "gs_fastq_open": (ci, [vp, C.c_char_p, ci, ci]), "gs_fastq_next": (ci, [vp, i64, i64, vp]),
        "gs_fastq_totals": (ci, [vp, vp, vp, vp]), "gs_fastq_close": (ci, [vp]),
        "gs_host_match_files": (ci, [vp, vp, vp, ci, vp, vp, vp, vp]),
        "gs_host_match_into": (ci, [vp, vp, vp, ci, vp, vp, vp]),
        "gs_host_match_run": (ci, [vp, vp, vp, ci, vp, vp]),
        "gs_host_match_files_multi": (ci, [vp, ci, vp, vp, ci, vp, vp, vp]),
        "gs_host_stat": (C.c_int64, [ci]),
        "gs_host_release_pools": (ci, []),
        "gs_host_filter_files": (ci, [vp, ci, ci, C.c_double, vp, ci, C.c_char_p, C.c_char_p, ci, vp]),
        "gs_host_write_csv": (ci, [C.c_char_p, vp, vp, vp, vp]),
        "gs_host_last_error": (C.c_char_p, []), "gs_host_java_double": (ci, [C.c_double, vp, ci]),
        "gs_host_gunzip": (ci, [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t]),
        "gs_host_gunzip_parallel": (ci, [vp, C.c_size_t, vp, C.c_size_t, vp, ci, C.c_size_t, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        msg = (lib().gs_host_last_error() or b"").decode(errors="replace")
        if not msg:
            msg = (_b.lib().gs_last_error() or b"").decode(errors="replace")
        raise _b.GsError(rc, msg)


def _cstr_array(items):
    if items is None:
        return None
    arr = (C.c_char_p * len(items))()
    for i, s in enumerate(items):
        arr[i] = None if s is None else (s if isinstance(s, bytes) else str(s).encode())
    return arr


class FastqReader:
    """AbstractFastqReader's record semantics (FASTQ or FASTA, gzip by content) as batches of numpy arrays"""

    def __init__(self, path, k=31, fasta=None):
        self.h = C.c_void_p()
        _check(lib().gs_fastq_open(C.byref(self.h), str(path).encode(), -1 if fasta is None else int(fasta), k))

    def next_batch(self, max_reads=1 << 20, max_bytes=1 << 30):
        b = _ReadBatch()
        _check(lib().gs_fastq_next(self.h, max_reads, max_bytes, C.byref(b)))
        n = b.n_reads
        if n == 0:
            return None

        def arr(ptr, ctype, cnt):
            if cnt == 0:
                return np.zeros(0, dtype=np.dtype(ctype))
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(cnt,)).copy()
        so, do, qo = (arr(p, C.c_uint64, n + 1) for p in (b.seq_off, b.desc_off, b.qual_off))
        return dict(n_reads=n, first_read_no=b.first_read_no, seq_off=so, desc_off=do, qual_off=qo,
                    seq=arr(b.seq, C.c_uint8, int(so[-1])), desc=arr(b.desc, C.c_uint8, int(do[-1])),
                    qual=arr(b.qual, C.c_uint8, int(qo[-1])))

    def totals(self):
        r, k, p = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(lib().gs_fastq_totals(self.h, C.byref(r), C.byref(k), C.byref(p)))
        return r.value, k.value, p.value

    def close(self):
        if getattr(self, "h", None):
            lib().gs_fastq_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def release_pools():
    """gs_host_release_pools: the page-locked blocks and device decoders the host layer keeps from call to call are freed"""
    _check(lib().gs_host_release_pools())


def match_files(store, paths, config=None, filtered_path=None, kraken_out_path=None, write_all=True, taxids=None,
                batch_reads=0, with_probs=False, max_contig_desc=False):
    """FastqKMerMatcher.runMatcher over files: returns (table, dtable, Totals); with_probs = the reference's withProbs
    (written reads keep their quality lines); max_contig_desc=True: a fourth result, the names of the reads that hold the longest
    contigs (list of bytes per value index)"""
    cfg = (config or _b.MatchConfig())._c()
    parr = _cstr_array(list(paths))
    tarr = _cstr_array(taxids)
    nv = store.n_values
    stride = 128
    descs = np.zeros((nv, stride), dtype=np.uint8) if max_contig_desc else None
    opts = _MatchOpts(None if filtered_path is None else str(filtered_path).encode(),
                      None if kraken_out_path is None else str(kraken_out_path).encode(), int(write_all),
                      tarr, batch_reads, int(with_probs), None if descs is None else descs.ctypes.data_as(C.c_void_p), stride)
    table = np.zeros((nv, _b.N_COLS), dtype=np.int64)
    dtable = np.zeros((nv, _b.N_DCOLS), dtype=np.float64)
    tot = Totals()
    _check(lib().gs_host_match_files(store.h, C.byref(cfg), parr, len(paths), C.byref(opts),
                                     table.ctypes.data_as(C.c_void_p), dtable.ctypes.data_as(C.c_void_p), C.byref(tot)))
    if descs is not None:
        return table, dtable, tot, [bytes(row).split(b"\0", 1)[0] for row in descs]
    return table, dtable, tot


def stat(which=0):
    """gs_host_stat: 0 = chunks that went through the general (multi-line) FASTQ device path in this process so far"""
    return int(lib().gs_host_stat(which))


def match_files_multi(stores, paths, config=None):
    """gs_host_match_files_multi: the files of a sample over several store replicas (one per device) of this process"""
    cfg = (config or _b.MatchConfig())._c()
    parr = _cstr_array(list(paths))
    darr = (C.c_void_p * len(stores))(*[s.h for s in stores])
    nv = stores[0].n_values
    table = np.zeros((nv, _b.N_COLS), dtype=np.int64)
    dtable = np.zeros((nv, _b.N_DCOLS), dtype=np.float64)
    tot = Totals()
    _check(lib().gs_host_match_files_multi(darr, len(stores), C.byref(cfg), parr, len(paths),
                                           table.ctypes.data_as(C.c_void_p), dtable.ctypes.data_as(C.c_void_p), C.byref(tot)))
    return table, dtable, tot


def match_files_into(matcher, paths, file_index):
    """the given files into `matcher`'s run (no finish): returns (reads per file, Totals); file_index = position of each
    file in the global file order (read numbers are file_index << 32 | read in file)"""
    parr = _cstr_array(list(paths))
    fi = np.ascontiguousarray(file_index, dtype=np.int32)
    counts = np.zeros(max(len(fi), 1), dtype=np.int64)
    tot = Totals()
    _check(lib().gs_host_match_into(matcher.h, matcher.store.h, parr, len(fi), fi.ctypes.data_as(C.c_void_p),
                                    counts.ctypes.data_as(C.c_void_p), C.byref(tot)))
    return counts[:len(fi)], tot


def filter_files(bloom, k, paths, min_pos_count=1, positive_ratio=0.2, filtered_path=None, rest_path=None,
                 with_probs=False):
    parr = _cstr_array(list(paths))
    tot = Totals()
    _check(lib().gs_host_filter_files(bloom.h, k, min_pos_count, positive_ratio, parr, len(paths),
                                      None if filtered_path is None else str(filtered_path).encode(),
                                      None if rest_path is None else str(rest_path).encode(), int(with_probs),
                                      C.byref(tot)))
    return tot


def write_csv(path, parent_vi, taxids, db_kmers, db_kmers_total, table, dtable, totals, names=None, ranks=None,
              position=None, max_contig_desc=None, max_kmer_counts=None):
    """MatchingResult.completeResults + ResultReporter.printMatchResult"""
    pv = np.ascontiguousarray(parent_vi, dtype=np.int32)
    dk = np.ascontiguousarray(db_kmers, dtype=np.int64)
    pos = None if position is None else np.ascontiguousarray(position, dtype=np.int32)
    keep = [_cstr_array(taxids), _cstr_array(names), _cstr_array(ranks), _cstr_array(max_contig_desc)]
    mc = None if max_kmer_counts is None else np.ascontiguousarray(max_kmer_counts, dtype=np.int16)
    info = _TaxInfo(len(pv), pv.ctypes.data_as(C.c_void_p), None if pos is None else pos.ctypes.data_as(C.c_void_p),
                    keep[0], keep[1], keep[2], dk.ctypes.data_as(C.c_void_p), db_kmers_total, keep[3],
                    None if mc is None else mc.ctypes.data_as(C.c_void_p), 0 if mc is None else mc.shape[1])
    t = np.ascontiguousarray(table, dtype=np.int64)
    d = np.ascontiguousarray(dtable, dtype=np.float64)
    _check(lib().gs_host_write_csv(str(path).encode(), C.byref(info), t.ctypes.data_as(C.c_void_p),
                                   d.ctypes.data_as(C.c_void_p), C.byref(totals)))


def gunzip(data, expected_size, block=1 << 20):
    """the ingest path's gzip decoder on bytes (test hook): returns the decoded bytes"""
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.empty(max(int(expected_size), 1), dtype=np.uint8)
    n = C.c_size_t(0)
    _check(lib().gs_host_gunzip(src.ctypes.data_as(C.c_void_p) if len(src) else None, len(src), out.ctypes.data_as(C.c_void_p),
                                int(expected_size), C.byref(n), int(block)))
    return out[:n.value].tobytes()


def gunzip_parallel(data, expected_size, threads=4, chunk=1 << 20, block=1 << 20):
    """the multi-threaded gzip decoder of the text path on bytes (test hook)"""
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.empty(max(int(expected_size), 1), dtype=np.uint8)
    n = C.c_size_t(0)
    _check(lib().gs_host_gunzip_parallel(src.ctypes.data_as(C.c_void_p) if len(src) else None, len(src),
                                         out.ctypes.data_as(C.c_void_p), int(expected_size), C.byref(n), int(threads), int(chunk), int(block)))
    return out[:n.value].tobytes()


def java_double(v):
    buf = C.create_string_buffer(64)
    _check(lib().gs_host_java_double(float(v), buf, 64))
    return buf.value.decode()
