"""ctypes binding of include/gsgpu.h (libgsgpu.so).

Host-side mirror of the reference interface for the hot path:

* ``DeviceKMerStore``   <- KMerStore + SmallTaxTree as the matcher sees them (C/store/KMerStore.java:45-317)
* ``FastqKMerMatcher``  <- C/match/FastqKMerMatcher.java (runMatcher :181-235, matchRead :327-535)
* ``DeviceBloomFilter`` <- KMerProbFilter (C/bloom/AbstractKMerBloomFilter.java, BlockedKMerBloomFilter.java)
* ``FastqBloomFilter``  <- C/bloom/FastqBloomFilter.java (runFilter :80-89, isAcceptRead :120-161)

numpy arrays are host batches (GS_MEM_HOST); objects exposing ``data_ptr()`` (torch tensors on the GPU) are
passed as device batches (GS_MEM_DEVICE).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

N_COLS, N_DCOLS, N_SUMS = 10, 4, 7
COLS = ("reads", "kmers from reads", "kmers", "unique kmers", "contigs", "contig len sq sum", "max contig length",
        "reads >=1 kmer", "reads bps", "max contig read no")
MEM_HOST, MEM_DEVICE = 0, 1
F_FOUND, F_RETURNED, F_COUNTED = 1, 2, 4
BLOOM_XOR, BLOOM_MURMUR, BLOOM_BLOCKED = 0, 1, 2

# every symbol include/gsgpu.h declares (checked by the CPU test-suite against the built library)
ABI_SYMBOLS = (
    "gs_last_error", "gs_strerror", "gs_device_cache_trim", "gs_abi_version", "gs_device_count",
    "gs_match_merge", "gs_match_max_contig_reads", "gs_db_create", "gs_db_get_info", "gs_db_destroy", "gs_db_save", "gs_db_load",
    "gs_match_begin", "gs_match_submit", "gs_match_submit_async", "gs_match_wait", "gs_match_sync", "gs_match_finish", "gs_match_reset", "gs_match_destroy",
    "gs_match_device_state", "gs_match_or_bitmap", "gs_match_kernel_time", "gs_match_segments",
    "gs_match_segments_fetch", "gs_match_max_counts", "gs_db_create_striped", "gs_db_create_stripe", "gs_db_stripe_export", "gs_db_stripe_attach", "gs_db_load_striped", "gs_db_load_stripe", "gs_dbbuild_begin", "gs_dbbuild_set_range", "gs_dbbuild_add", "gs_dbbuild_finish", "gs_dbbuild_fetch", "gs_dbbuild_to_db", "gs_dbbuild_destroy", "gs_db_create_part", "gs_match_encode", "gs_match_probe_keys", "gs_match_encode_route", "gs_match_route_geometry", "gs_unroute_region", "gs_match_reduce", "gs_route_keys",
    "gs_unroute_nodes",
    "gs_match_submit_text", "gs_match_submit_fasta", "gs_match_submit_fastq_ml", "gs_match_text_wait_copy", "gs_match_text_status", "gs_match_text_clear_error",
    "gs_match_text_select", "gs_match_segments_text", "gs_match_text_newlines", "gs_match_text_read_bounds", "gs_match_text_line_classes",
    "gs_pinned_alloc", "gs_pinned_free",
    "gs_filter_submit_text", "gs_filter_text_wait_copy", "gs_filter_text_status", "gs_filter_text_reset",
    "gs_filter_submit_fasta", "gs_filter_submit_fastq_ml", "gs_filter_text_read_bounds", "gs_filter_text_line_classes",
    "gs_bloom_create", "gs_bloom_build", "gs_bloom_get", "gs_bloom_destroy", "gs_filter_submit", "gs_filter_sync", "gs_filter_kernel_time",
    "gs_calibrate",
    "gs_match_get_device", "gs_inflate_members", "gs_inflater_create", "gs_inflater_feed", "gs_inflater_tail", "gs_gunzipper_open", "gs_gunzipper_reopen", "gs_gunzipper_next", "gs_gunzipper_info", "gs_gunzipper_first_span", "gs_gunzipper_park", "gs_gunzipper_close", "gs_gunzip_plan_device", "gs_gunzip_free", "gs_gunzip_device", "gs_text_cut_device", "gs_device_fetch", "gs_inflater_fetch", "gs_filter_get_device", "gs_inflater_reset", "gs_inflater_destroy", "gs_inflate_last_error",
    "gs_filter_compact_text", "gs_match_compact_text", "gs_deflater_create", "gs_deflater_pack", "gs_deflater_info", "gs_deflater_destroy", "gs_deflater_append", "gs_deflater_pending", "gs_deflater_flush",
    "gs_deflate_bound", "gs_deflate_host", "gs_deflate_host_reference", "gs_deflate_last_error", "gs_match_text_descriptors", "gs_match_submit_fixed",
)


class GsError(RuntimeError):
    """non-zero status from the C ABI (the JNI shim turns the same codes into RuntimeException)"""

    def __init__(self, code, msg):
        super().__init__(f"gsgpu error {code}: {msg}")
        self.code = code


class DbInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("n_values", C.c_int32), ("n_entries", C.c_int64), ("n_stored", C.c_int64),
                ("n_buckets", C.c_int64), ("table_bytes", C.c_int64), ("max_displacement", C.c_int32),
                ("value_bits", C.c_int32), ("gate_bytes", C.c_int64), ("mgate_bytes", C.c_int64),
                ("rec_bytes", C.c_int64), ("n_in_records", C.c_int64), ("n_stripes", C.c_int32), ("stripe", C.c_int32),
                ("stripe_bytes", C.c_int64)]


class _MatchCfg(C.Structure):
    _fields_ = [("classify", C.c_int32), ("count_unique", C.c_int32), ("max_paths", C.c_int32),
                ("threshold", C.c_int32), ("max_read_tax_err", C.c_double), ("max_read_class_err", C.c_double),
                ("profile", C.c_int32), ("max_kmer_res_counts", C.c_int32)]


def lib_path():
    # GS_LIBGSGPU: developer override used to A/B kernel build variants on the GPU box
    return os.environ.get("GS_LIBGSGPU") or os.path.join(_HERE, "libgsgpu.so")


def _preload_hip_runtime():
    """One HIP runtime per process.  libgsgpu.so needs `libamdhip64.so.7`; a PyTorch-ROCm wheel ships its own
    copy (same SONAME) next to libtorch_hip.so.  Whichever copy is loaded first serves both, so when torch is
    installed its copy is loaded first -- then torch (device memory, torch.distributed/RCCL in bench.py and
    the tests) and this library share a single runtime and a single set of device contexts.  Set
    GS_HIP_RUNTIME=system to skip this (e.g. under a JVM without torch)."""
    if os.environ.get("GS_HIP_RUNTIME", "") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """load libgsgpu.so; raises if it was not built (no fallback implementation exists)"""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise GsError(-6, f"{path} is missing: build it with `make -C genestrip_amd/csrc` (hipcc, gfx950); "
                          "genestrip_amd has no CPU fallback")
    _preload_hip_runtime()
    L = C.CDLL(path)
    vp, i32, i64, dbl, ci = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_int
    sig = {
        "gs_last_error": (C.c_char_p, []), "gs_strerror": (C.c_char_p, [ci]), "gs_device_cache_trim": (ci, []), "gs_abi_version": (ci, []),
        "gs_device_count": (ci, [vp]),
        "gs_db_create": (ci, [vp, ci, ci, i64, vp, vp, i32, vp]), "gs_db_get_info": (ci, [vp, vp]),
        "gs_db_destroy": (ci, [vp]), "gs_db_save": (ci, [vp, C.c_char_p]), "gs_db_load": (ci, [vp, ci, C.c_char_p]),
        "gs_match_begin": (ci, [vp, vp, vp]), "gs_match_submit": (ci, [vp, vp, vp, i64, i64, ci, vp, vp]),
        "gs_match_submit_async": (ci, [vp, vp, vp, i64, i64, vp, vp, vp]), "gs_match_wait": (ci, [vp, i64]),
        "gs_match_sync": (ci, [vp]), "gs_match_finish": (ci, [vp, vp, vp]), "gs_match_reset": (ci, [vp]),
        "gs_match_destroy": (ci, [vp]), "gs_match_device_state": (ci, [vp, vp, vp, vp, vp, vp]),
        "gs_match_or_bitmap": (ci, [vp, vp, i64]), "gs_match_merge": (ci, [vp, ci]), "gs_match_max_contig_reads": (ci, [vp, vp]), "gs_match_kernel_time": (ci, [vp, vp, vp]),
        "gs_db_create_part": (ci, [vp, ci, ci, i64, vp, vp, i32, vp, ci, ci]),
        "gs_db_create_striped": (ci, [vp, vp, ci, ci, i64, vp, vp, i32, vp]),
        "gs_db_create_stripe": (ci, [vp, ci, ci, ci, ci, i64, vp, vp, i32, vp]),
        "gs_db_stripe_export": (ci, [vp, vp]),
        "gs_db_stripe_attach": (ci, [vp, ci, vp]),
        "gs_db_load_striped": (ci, [vp, vp, ci, C.c_char_p]),
        "gs_db_load_stripe": (ci, [vp, ci, ci, ci, C.c_char_p]),
        "gs_dbbuild_begin": (ci, [vp, ci, ci, i32, vp, ci, ci, ci]),
        "gs_dbbuild_add": (ci, [vp, vp, vp, vp, i64, ci, ci]),
        "gs_dbbuild_set_range": (ci, [vp, C.c_uint64, C.c_uint64]),
        "gs_dbbuild_finish": (ci, [vp, vp]),
        "gs_dbbuild_fetch": (ci, [vp, vp, vp]),
        "gs_dbbuild_to_db": (ci, [vp, vp]),
        "gs_dbbuild_destroy": (ci, [vp]),
        "gs_match_encode": (ci, [vp, vp, vp, i64, vp, vp]), "gs_match_probe_keys": (ci, [vp, vp, i64, vp]),
        "gs_match_encode_route": (ci, [vp, vp, vp, i64, vp, ci, i64, vp, vp, vp, vp, vp]),
        "gs_match_route_geometry": (ci, [vp, i64, vp, vp]),
        "gs_unroute_region": (ci, [vp, vp, vp, i64, vp]),
        "gs_match_reduce": (ci, [vp, vp, vp, i64, i64, vp, vp, vp, vp]),
        "gs_route_keys": (ci, [vp, vp, i64, ci, vp, vp, vp, vp]), "gs_unroute_nodes": (ci, [vp, vp, vp, vp, i64, vp, i64]),
        "gs_match_max_counts": (ci, [vp, vp]),
        "gs_match_submit_text": (ci, [vp, vp, i64, i64, ci, i64, vp, vp, vp]),
        "gs_match_submit_fasta": (ci, [vp, vp, i64, i64, i64, ci, i64, vp, vp, vp]),
        "gs_match_submit_fastq_ml": (ci, [vp, vp, i64, i64, ci, i64, vp, vp, vp, vp, vp, vp]),
        "gs_match_text_line_classes": (ci, [vp, vp]),
        "gs_match_text_wait_copy": (ci, [vp, i64]), "gs_match_text_status": (ci, [vp, vp, vp, vp]),
        "gs_match_text_clear_error": (ci, [vp]), "gs_match_text_select": (ci, [vp, ci]),
        "gs_match_segments_text": (ci, [vp, vp]), "gs_match_text_newlines": (ci, [vp, vp]),
        "gs_match_text_read_bounds": (ci, [vp, vp]),
        "gs_pinned_alloc": (ci, [vp, C.c_size_t]), "gs_pinned_free": (ci, [vp]),
        "gs_filter_submit_text": (ci, [vp, ci, ci, dbl, vp, i64, i64, ci, vp, vp, ci, vp]),
        "gs_filter_text_wait_copy": (ci, [vp, i64]), "gs_filter_text_status": (ci, [vp, vp, vp, vp]),
        "gs_filter_text_reset": (ci, [vp, ci]),
        "gs_filter_submit_fasta": (ci, [vp, ci, ci, dbl, vp, i64, i64, i64, ci, vp, vp, vp]),
        "gs_filter_submit_fastq_ml": (ci, [vp, ci, ci, dbl, vp, i64, i64, ci, vp, vp, vp, vp, vp, vp]),
        "gs_filter_text_read_bounds": (ci, [vp, vp]), "gs_filter_text_line_classes": (ci, [vp, vp]),
        "gs_match_segments": (ci, [vp, vp, vp, i64, ci, vp]), "gs_match_segments_fetch": (ci, [vp, vp, vp]),
        "gs_bloom_create": (ci, [vp, ci, ci, i64, i32, vp, vp, i64]),
        "gs_bloom_build": (ci, [vp, ci, ci, vp, i64, ci, i64, C.c_double]),
        "gs_bloom_get": (ci, [vp, vp, vp, vp, vp, i64]), "gs_bloom_destroy": (ci, [vp]),
        "gs_filter_submit": (ci, [vp, ci, ci, dbl, vp, vp, i64, ci, vp, ci]), "gs_filter_sync": (ci, [vp]),
        "gs_filter_kernel_time": (ci, [vp, vp, vp]),
        "gs_calibrate": (ci, [ci, ci, i64, vp]),
        "gs_match_get_device": (ci, [vp, vp]),
        "gs_inflate_members": (ci, [ci, vp, vp, i64, vp, i64, vp]),
        "gs_inflater_create": (ci, [vp, ci]),
        "gs_inflater_feed": (ci, [vp, vp, vp, i64, i64, i64, ci, vp, vp, vp, vp]),
        "gs_inflater_tail": (ci, [vp, vp, i64, vp]),
        "gs_inflater_fetch": (ci, [vp, vp, i64]),
        "gs_gunzipper_open": (ci, [vp, ci, vp, i64]),
        "gs_gunzipper_reopen": (ci, [vp, vp, i64]),
        "gs_gunzipper_next": (ci, [vp, i64, vp, vp, vp]),
        "gs_gunzipper_info": (ci, [vp, vp]),
        "gs_gunzipper_first_span": (ci, [vp, i64]),
        "gs_gunzipper_park": (ci, [vp]),
        "gs_gunzipper_close": (ci, [vp]),
        "gs_gunzip_plan_device": (ci, [ci, vp, i64, vp, vp, vp]),
        "gs_gunzip_free": (ci, [ci, vp]),
        "gs_gunzip_device": (ci, [ci, vp, i64, vp, i64, vp, vp]),
        "gs_text_cut_device": (ci, [ci, vp, i64, vp, vp]),
        "gs_device_fetch": (ci, [ci, vp, vp, i64]),
        "gs_filter_get_device": (ci, [vp, vp]),
        "gs_inflater_reset": (ci, [vp]),
        "gs_inflater_destroy": (ci, [vp]),
        "gs_inflate_last_error": (C.c_char_p, []),
        "gs_filter_compact_text": (ci, [vp, ci, ci, ci, vp, vp, vp]),
        "gs_match_compact_text": (ci, [vp, ci, ci, vp, vp, vp]),
        "gs_deflater_create": (ci, [vp, ci]),
        "gs_deflater_pack": (ci, [vp, vp, i64, vp, i64, vp]),
        "gs_deflater_info": (ci, [vp, vp]),
        "gs_deflater_append": (ci, [vp, vp, i64]),
        "gs_deflater_pending": (i64, [vp]),
        "gs_deflater_flush": (ci, [vp, vp, i64, vp]),
        "gs_deflater_destroy": (ci, [vp]),
        "gs_deflate_bound": (i64, [i64]),
        "gs_deflate_host": (ci, [ci, vp, i64, vp, i64, vp]),
        "gs_deflate_host_reference": (ci, [vp, i64, vp, i64, vp]),
        "gs_deflate_last_error": (C.c_char_p, []),
        "gs_match_text_descriptors": (ci, [vp, vp, i32, vp, i32]),
        "gs_match_submit_fixed": (ci, [vp, vp, i32, i64, i64, ci, vp, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise GsError(rc, (lib().gs_last_error() or b"").decode(errors="replace"))


def abi_version():
    return lib().gs_abi_version()


def device_count():
    n = C.c_int(0)
    rc = lib().gs_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class InflateMember(C.Structure):
    _fields_ = [("payload_offset", C.c_int64), ("payload_len", C.c_uint32), ("isize", C.c_uint32), ("crc32", C.c_uint32),
                ("reserved", C.c_uint32)]


def bgzf_members(data):
    """the members of a BGZF buffer (bytes / numpy uint8), found from their headers alone: list of (payload_offset,
    payload_len, isize, crc32); stops at the first thing that is not a BGZF member -> (members, offset reached)"""
    import struct
    b = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    out, o, n = [], 0, len(b)
    while o + 28 <= n and b[o] == 0x1f and b[o + 1] == 0x8b and b[o + 2] == 8 and (b[o + 3] & 4):
        if b[o + 3] & ~4:  # name / comment / header CRC: not what bgzip writes
            break
        xlen = struct.unpack_from("<H", b, o + 10)[0]
        q, bsize = o + 12, None
        while q + 4 <= o + 12 + xlen:
            slen = struct.unpack_from("<H", b, q + 2)[0]
            if b[q:q + 2] == b"BC" and slen == 2:
                bsize = struct.unpack_from("<H", b, q + 4)[0]
            q += 4 + slen
        if bsize is None or o + bsize + 1 > n or bsize + 1 < 12 + xlen + 8:
            break
        end = o + bsize + 1
        crc, isize = struct.unpack_from("<II", b, end - 8)
        if isize > 65536:
            break
        out.append((o + 12 + xlen, end - 8 - (o + 12 + xlen), isize, crc))
        o = end
    return out, o


def inflate_members(data, members, device=0):
    """gs_inflate_members: -> (text bytes as numpy uint8, status int32[]); raises GsError when a member is corrupt"""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    arr = (InflateMember * max(1, len(members)))()
    total = 0
    for i, (po, pl, isz, crc) in enumerate(members):
        arr[i] = InflateMember(po, pl, isz, crc, 0)
        total += isz
    out = np.zeros(max(total, 1), dtype=np.uint8)
    st = np.full(max(len(members), 1), -1, dtype=np.int32)
    rc = lib().gs_inflate_members(device, buf.ctypes.data_as(C.c_void_p), arr, len(members), out.ctypes.data_as(C.c_void_p), total,
                                  st.ctypes.data_as(C.c_void_p))
    if rc != 0:
        e = GsError(rc, (lib().gs_inflate_last_error() or b"").decode(errors="replace"))
        e.status = st[:len(members)].copy()
        raise e
    return out[:total], st[:len(members)]


def gunzip_device(data, expect_bytes, device=0):
    """gs_gunzip_device: a single-member gzip stream inflated on the device -> (text as numpy uint8, info[4] = segments, chunks searched,
    batches, mirages); raises GsError (code -4: a stream this path does not take -- the host decoders do; -1: a damaged stream)"""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(max(int(expect_bytes), 1), dtype=np.uint8)
    n = C.c_int64(0)
    info = (C.c_int64 * 4)()
    rc = lib().gs_gunzip_device(device, buf.ctypes.data_as(C.c_void_p), len(buf), out.ctypes.data_as(C.c_void_p), int(expect_bytes), C.byref(n), info)
    if rc != 0:
        raise GsError(rc, (lib().gs_inflate_last_error() or b"").decode(errors="replace"))
    return out[:n.value], list(info)


def _fetch_device(device, p, n):
    out = np.zeros(max(n, 1), dtype=np.uint8)
    if n:
        rc = lib().gs_device_fetch(device, p, out.ctypes.data_as(C.c_void_p), n)
        if rc != 0:
            raise GsError(rc, (lib().gs_inflate_last_error() or b"").decode(errors="replace"))
    return out[:n]


def _deflate(fn, data, *head):
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    cap = int(lib().gs_deflate_bound(len(buf))) + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_int64(0)
    rc = fn(*head, buf.ctypes.data_as(C.c_void_p) if len(buf) else None, len(buf), out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    if rc != 0:
        raise GsError(rc, (lib().gs_deflate_last_error() or b"").decode(errors="replace"))
    return out[:n.value]


def deflate_device(data, device=0):
    """gs_deflate_host: bytes -> BGZF members written on the device (numpy uint8; no end-of-file block)"""
    return _deflate(lib().gs_deflate_host, data, device)


def deflate_reference(data):
    """gs_deflate_host_reference: the same format from the CPU loop over the same table builder (no device)"""
    return _deflate(lib().gs_deflate_host_reference, data)


class DeviceDeflater:
    """gs_deflater_*: device text -> BGZF members in a (page-locked) host buffer"""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        self.device = device
        rc = lib().gs_deflater_create(C.byref(self.h), device)
        if rc != 0:
            raise GsError(rc, (lib().gs_deflate_last_error() or b"").decode(errors="replace"))

    def pack(self, d_text, n, out):
        """d_text: device pointer (int) or tensor with data_ptr(); out: numpy uint8 of at least gs_deflate_bound(n) bytes -> bytes written"""
        p = C.c_void_p(d_text.data_ptr() if hasattr(d_text, "data_ptr") else int(d_text))
        _ready(d_text)
        n_out = C.c_int64(0)
        rc = lib().gs_deflater_pack(self.h, p, int(n), out.ctypes.data_as(C.c_void_p), int(out.shape[0]), C.byref(n_out))
        if rc != 0:
            raise GsError(rc, (lib().gs_deflate_last_error() or b"").decode(errors="replace"))
        return n_out.value

    def append(self, d_text, n):
        """gs_deflater_append: the text waits on the device for flush()"""
        p = C.c_void_p(d_text.data_ptr() if hasattr(d_text, "data_ptr") else int(d_text))
        _ready(d_text)
        rc = lib().gs_deflater_append(self.h, p, int(n))
        if rc != 0:
            raise GsError(rc, (lib().gs_deflate_last_error() or b"").decode(errors="replace"))

    def pending(self):
        return int(lib().gs_deflater_pending(self.h))

    def flush(self):
        """the waiting text as BGZF members (numpy uint8)"""
        out = np.empty(deflate_bound(self.pending()), dtype=np.uint8)
        n_out = C.c_int64(0)
        rc = lib().gs_deflater_flush(self.h, out.ctypes.data_as(C.c_void_p), int(out.shape[0]), C.byref(n_out))
        if rc != 0:
            raise GsError(rc, (lib().gs_deflate_last_error() or b"").decode(errors="replace"))
        return out[:n_out.value]

    def close(self):
        if getattr(self, "h", None):
            lib().gs_deflater_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def deflate_bound(n):
    return int(lib().gs_deflate_bound(int(n)))


BGZF_EOF = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


CAL_VALU_PURE, CAL_VALU_MIX, CAL_SALU, CAL_VALU_SALU = 0, 1, 2, 3
CAL_VMEM_BYTES, CAL_VMEM_WORDS, CAL_VMEM_SHARED_LINES, CAL_VMEM_SCATTERED, CAL_RANDOM_LINES = 4, 5, 6, 7, 8


def calibrate(what, arg=0, device=0):
    """gs_calibrate: a measured ceiling of the device (include/gsgpu.h) -> dict(rate, ms, count, n_cu)"""
    out = (C.c_double * 4)()
    _check(lib().gs_calibrate(device, what, arg, out))
    return {"rate": out[0], "ms": out[1], "count": out[2], "n_cu": int(out[3])}


def _ptr(a):
    """(pointer, mem kind) of a numpy array (host) or a tensor-like with data_ptr() (device)"""
    if a is None:
        return None, None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(C.c_void_p), MEM_HOST
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr()), (MEM_DEVICE if getattr(a, "is_cuda", True) else MEM_HOST)
    raise TypeError(f"unsupported buffer {type(a)}")


def _ready(*bufs):
    """The library works on its own HIP stream and takes device buffers as complete (include/gsgpu.h): wait for the
    producer's stream (torch's current one) before handing tensors over."""
    torch = sys.modules.get("torch")
    if torch is None:
        return
    for b in bufs:
        if b is not None and getattr(b, "is_cuda", False):
            torch.cuda.current_stream(b.device).synchronize()
            return


class DeviceKMerStore:
    """k-mer -> value-index store plus taxonomy, resident in HBM (gs_db)."""

    def __init__(self, k, kmers_sorted, value_idx, n_values, parent_vi=None, device=0, n_parts=1, part=0,
                 partition=None):
        """partition: build a partition store for the split pipeline encode -> probe_keys -> reduce (gs_db_create_part:
        every key in the table); default: n_parts > 1.  Otherwise the store serves the fused kernels (gs_db_create)."""
        kmers = np.ascontiguousarray(kmers_sorted, dtype=np.int64)
        vidx = np.ascontiguousarray(value_idx, dtype=np.int32)
        if len(kmers) != len(vidx):
            raise ValueError("kmers / value_idx length mismatch")
        pv = None if parent_vi is None else np.ascontiguousarray(parent_vi, dtype=np.int32)
        if pv is not None and len(pv) != n_values:
            raise ValueError("parent_vi must have n_values entries")
        self.h = C.c_void_p()
        self.k, self.n_values, self.device = k, n_values, device
        if partition is None:
            partition = n_parts > 1
        if partition:
            _check(lib().gs_db_create_part(C.byref(self.h), device, k, len(kmers), kmers.ctypes.data_as(C.c_void_p),
                                           vidx.ctypes.data_as(C.c_void_p), n_values,
                                           None if pv is None else pv.ctypes.data_as(C.c_void_p), n_parts, part))
        else:
            _check(lib().gs_db_create(C.byref(self.h), device, k, len(kmers), kmers.ctypes.data_as(C.c_void_p),
                                      vidx.ctypes.data_as(C.c_void_p), n_values,
                                      None if pv is None else pv.ctypes.data_as(C.c_void_p)))

    @staticmethod
    def _arrays(kmers, value_idx, n_values, parent_vi):
        kmers = np.ascontiguousarray(kmers, dtype=np.int64)
        vidx = np.ascontiguousarray(value_idx, dtype=np.int32)
        if len(kmers) != len(vidx):
            raise ValueError("kmers / value_idx length mismatch")
        pv = None if parent_vi is None else np.ascontiguousarray(parent_vi, dtype=np.int32)
        if pv is not None and len(pv) != n_values:
            raise ValueError("parent_vi must have n_values entries")
        return kmers, vidx, pv

    @classmethod
    def _wrap(cls, h, k, n_values, device):
        self = cls.__new__(cls)
        self.h, self.k, self.n_values, self.device = h, k, n_values, device
        return self

    @classmethod
    def striped(cls, k, kmers, value_idx, n_values, parent_vi=None, devices=(0, 0)):
        """gs_db_create_striped: ONE record table split over `devices` (stripe p in the HBM of devices[p]; a device may
        repeat), gates / overflow table / tree on each.  Returns one handle per stripe; every handle serves the fused
        kernels on its device and reads foreign stripes over peer access."""
        kmers, vidx, pv = cls._arrays(kmers, value_idx, n_values, parent_vi)
        n = len(devices)
        out = (C.c_void_p * n)()
        dev = (C.c_int * n)(*devices)
        _check(lib().gs_db_create_striped(out, dev, n, k, len(kmers), kmers.ctypes.data_as(C.c_void_p),
                                          vidx.ctypes.data_as(C.c_void_p), n_values,
                                          None if pv is None else pv.ctypes.data_as(C.c_void_p)))
        return [cls._wrap(C.c_void_p(out[p]), k, n_values, devices[p]) for p in range(n)]

    @classmethod
    def stripe(cls, k, kmers, value_idx, n_values, parent_vi=None, device=0, n_stripes=2, stripe=0):
        """gs_db_create_stripe: this process's stripe of a striped store (one process per GPU); the other stripes are
        attached from their owners' handles: export_stripe() -> all-gather -> attach_stripe()
        (genestrip_amd.distributed.striped_store does all of it)."""
        kmers, vidx, pv = cls._arrays(kmers, value_idx, n_values, parent_vi)
        h = C.c_void_p()
        _check(lib().gs_db_create_stripe(C.byref(h), device, n_stripes, stripe, k, len(kmers), kmers.ctypes.data_as(C.c_void_p),
                                         vidx.ctypes.data_as(C.c_void_p), n_values,
                                         None if pv is None else pv.ctypes.data_as(C.c_void_p)))
        return cls._wrap(h, k, n_values, device)

    @classmethod
    def load_striped(cls, path, devices=(0, 0)):
        """gs_db_load_striped: a store file (save() of a plain store) into the HBM of `devices`, one stripe each"""
        n = len(devices)
        out = (C.c_void_p * n)()
        _check(lib().gs_db_load_striped(out, (C.c_int * n)(*devices), n, str(path).encode()))
        res = []
        for p in range(n):
            s = cls._wrap(C.c_void_p(out[p]), 0, 0, devices[p])
            i = s.info
            s.k, s.n_values = i.k, i.n_values
            res.append(s)
        return res

    @classmethod
    def load_stripe(cls, path, device=0, n_stripes=2, stripe=0):
        """gs_db_load_stripe: this process's stripe of a store file (then export_stripe / attach_stripe as for stripe())"""
        h = C.c_void_p()
        _check(lib().gs_db_load_stripe(C.byref(h), device, n_stripes, stripe, str(path).encode()))
        s = cls._wrap(h, 0, 0, device)
        i = s.info
        s.k, s.n_values = i.k, i.n_values
        return s

    def export_stripe(self):
        buf = C.create_string_buffer(64)
        _check(lib().gs_db_stripe_export(self.h, buf))
        return buf.raw

    def attach_stripe(self, stripe, handle):
        _check(lib().gs_db_stripe_attach(self.h, stripe, C.c_char_p(bytes(handle))))

    @classmethod
    def load(cls, path, device=0):
        """open a native store file written by save() (gs_db_load)"""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        _check(lib().gs_db_load(C.byref(self.h), device, str(path).encode()))
        self.device = device
        i = self.info
        self.k, self.n_values = i.k, i.n_values
        return self

    def save(self, path):
        _check(lib().gs_db_save(self.h, str(path).encode()))

    @property
    def info(self):
        i = DbInfo()
        _check(lib().gs_db_get_info(self.h, C.byref(i)))
        return i

    def close(self):
        if getattr(self, "h", None):
            lib().gs_db_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kmer_ranges(k, n):
    """n ranges [lo, hi) of the canonical k-mer with about equal shares of the k-mers (a canonical k-mer is the larger of
    two strands: P(x <= t) ~ (t / 4^k)^2): for gs_dbbuild_set_range"""
    import math
    top = 1 << (2 * k)
    cuts = [math.isqrt(top * top * i // n) for i in range(n)] + [top]
    return [(cuts[i], cuts[i + 1]) for i in range(n) if cuts[i] < cuts[i + 1]]


class DeviceDbBuilder:
    """gs_dbbuild: distinct canonical k-mers of genome regions with the LCA of the nodes that hold them (FillDBGoal + DBGoal)"""

    def __init__(self, k, n_values, parent_vi, device=0, lower_case_bases=True, max_dust=-1, step_size=1):
        pv = np.ascontiguousarray(parent_vi, dtype=np.int32)
        if len(pv) != n_values:
            raise ValueError("parent_vi must have n_values entries")
        self.h = C.c_void_p()
        self.k, self.n_values = k, n_values
        _check(lib().gs_dbbuild_begin(C.byref(self.h), device, k, n_values, pv.ctypes.data_as(C.c_void_p), int(lower_case_bases),
                                      max_dust, step_size))

    def set_range(self, lo, hi):
        """keep only the canonical k-mers in [lo, hi) (before the first add): one builder per range of kmer_ranges()"""
        _check(lib().gs_dbbuild_set_range(self.h, lo, hi))

    def add(self, seq, offsets, node_vi, update=False):
        """regions: seq (uint8) + offsets (uint64, n + 1, from 0), both numpy or both device tensors; node_vi: numpy int32[n]"""
        ps, mem = _ptr(seq)
        po, mem2 = _ptr(offsets)
        assert mem == mem2, "seq and offsets must live in the same memory space"
        nv = np.ascontiguousarray(node_vi, dtype=np.int32)
        n = (offsets.shape[0] if hasattr(offsets, "shape") else len(offsets)) - 1
        if len(nv) != n:
            raise ValueError("node_vi must have one entry per region")
        _ready(seq, offsets)
        _check(lib().gs_dbbuild_add(self.h, ps, po, nv.ctypes.data_as(C.c_void_p), n, mem, int(update)))

    def finish(self):
        """-> (kmers int64 ascending, value_idx int32): what DeviceKMerStore takes"""
        n = C.c_int64(0)
        _check(lib().gs_dbbuild_finish(self.h, C.byref(n)))
        kmers = np.zeros(n.value, dtype=np.int64)
        vals = np.zeros(n.value, dtype=np.int32)
        _check(lib().gs_dbbuild_fetch(self.h, kmers.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p)))
        return kmers, vals

    def finish_count(self):
        """sort + fold only: the number of stored k-mers (the arrays stay on the device: fetch() or to_store())"""
        n = C.c_int64(0)
        _check(lib().gs_dbbuild_finish(self.h, C.byref(n)))
        return n.value

    def to_store(self, device=0):
        """gs_dbbuild_to_db: the store over the built arrays, laid out on the device, nothing through the host"""
        self.finish_count()
        h = C.c_void_p()
        _check(lib().gs_dbbuild_to_db(self.h, C.byref(h)))
        return DeviceKMerStore._wrap(h, self.k, self.n_values, device)

    def close(self):
        if getattr(self, "h", None):
            lib().gs_dbbuild_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_runs(matchers):
    """gs_match_merge: the runs of one process (one per GPU, or several on one GPU) into a global state held by each"""
    arr = (C.c_void_p * len(matchers))(*[m.h for m in matchers])
    _check(lib().gs_match_merge(arr, len(matchers)))


class MatchConfig:
    """the GSConfigKey values that parameterise matchRead (C/GSConfigKey.java:302-350)"""

    def __init__(self, classify=True, count_unique=True, max_paths=10, threshold=1, max_read_tax_err=-1.0,
                 max_read_class_err=-1.0, profile=False, max_kmer_res_counts=0):
        self.classify, self.count_unique, self.max_paths, self.threshold = classify, count_unique, max_paths, threshold
        self.max_read_tax_err, self.max_read_class_err, self.profile = max_read_tax_err, max_read_class_err, profile
        self.max_kmer_res_counts = max_kmer_res_counts

    def _c(self):
        return _MatchCfg(int(self.classify), int(self.count_unique), self.max_paths, self.threshold,
                         self.max_read_tax_err, self.max_read_class_err, int(self.profile), int(self.max_kmer_res_counts))


class FastqKMerMatcher:
    """one runMatcher() scope: begin -> submit batches -> finish."""

    def __init__(self, store, config=None):
        self.store = store
        self.config = config or MatchConfig()
        cfg = self.config._c()
        self.h = C.c_void_p()
        _check(lib().gs_match_begin(C.byref(self.h), store.h, C.byref(cfg)))

    def submit(self, seq, offsets, first_read_no=0, class_vi=None, flags=None, n_reads=None):
        ps, mem = _ptr(seq)
        po, mem2 = _ptr(offsets)
        assert mem == mem2, "seq and offsets must live in the same memory space"
        if n_reads is None:
            n_reads = (offsets.shape[0] if hasattr(offsets, "shape") else len(offsets)) - 1
        pc, _ = _ptr(class_vi)
        pf, _ = _ptr(flags)
        _ready(seq, offsets, class_vi, flags)
        _check(lib().gs_match_submit(self.h, ps, po, n_reads, first_read_no, mem, pc, pf))

    def submit_fixed(self, seq, read_len, n_reads, first_read_no=0, class_vi=None, flags=None):
        """gs_match_submit_fixed: n_reads reads of read_len bytes each, back to back in `seq` (numpy = host, tensor = device)"""
        ps, mem = _ptr(seq)
        pc, _ = _ptr(class_vi)
        pf, _ = _ptr(flags)
        _ready(seq, class_vi, flags)
        _check(lib().gs_match_submit_fixed(self.h, ps, int(read_len), int(n_reads), int(first_read_no), mem, pc, pf))

    def submit_async(self, seq, offsets, first_read_no=0, class_vi=None, flags=None, n_reads=None):
        """host batch (numpy arrays, best page-locked), queued: returns a ticket for wait(); the arrays must stay as they
        are until then.  Two batches can be under way (gs_match_submit_async)."""
        ps, mem = _ptr(seq)
        po, mem2 = _ptr(offsets)
        assert mem == mem2 == MEM_HOST, "submit_async takes host arrays"
        if n_reads is None:
            n_reads = (offsets.shape[0] if hasattr(offsets, "shape") else len(offsets)) - 1
        pc, _ = _ptr(class_vi)
        pf, _ = _ptr(flags)
        ticket = C.c_int64(-1)
        _check(lib().gs_match_submit_async(self.h, ps, po, n_reads, first_read_no, pc, pf, C.byref(ticket)))
        return ticket.value

    def wait(self, ticket):
        _check(lib().gs_match_wait(self.h, ticket))

    def match_reads(self, seq, offsets, first_read_no=0):
        """host batch -> (class_vi, flags) numpy arrays (the per-read outcome of matchRead)"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        cv = np.full(n, -1, dtype=np.int32)
        fl = np.zeros(n, dtype=np.uint8)
        if len(seq) == 0:
            seq = np.zeros(1, dtype=np.uint8)
        self.submit(seq, offsets, first_read_no, cv, fl)
        return cv, fl

    def submit_text(self, text, n_lines=None, first_read_no=0, class_vi=None, flags=None):
        """raw FASTQ text of whole 4-line records (numpy uint8 / bytes on the host, or a device tensor): the records
        are found on the device (gs_match_submit_text).  Returns the ticket."""
        if isinstance(text, (bytes, bytearray)):
            text = np.frombuffer(bytes(text), dtype=np.uint8)
        if n_lines is None:
            n_lines = int((text == 10).sum())
        n_bytes = int(text.shape[0])
        pt, mem = _ptr(text) if n_bytes else (None, MEM_HOST)
        pc, _ = _ptr(class_vi)
        pf, _ = _ptr(flags)
        _ready(text, class_vi, flags)
        ticket = C.c_int64(-1)
        self._text_keep = text  # the copy is asynchronous
        _check(lib().gs_match_submit_text(self.h, pt, n_bytes, int(n_lines), mem, first_read_no, pc, pf, C.byref(ticket)))
        return ticket.value

    def submit_fasta(self, text, n_lines=None, n_records=None, first_read_no=0, class_vi=None, flags=None):
        """raw FASTA text of whole records (gs_match_submit_fasta); returns the ticket"""
        if isinstance(text, (bytes, bytearray)):
            text = np.frombuffer(bytes(text), dtype=np.uint8)
        n_bytes = int(text.shape[0])
        if n_lines is None or n_records is None:
            host = text if isinstance(text, np.ndarray) else text.cpu().numpy()
            nl = np.flatnonzero(host == 10)
            starts = np.concatenate([[0], nl[:-1] + 1]) if len(nl) else np.zeros(0, dtype=np.int64)
            n_lines = len(nl) if n_lines is None else n_lines
            n_records = int((host[starts] == ord(">")).sum()) if n_records is None else n_records
        pt, mem = _ptr(text) if n_bytes else (None, MEM_HOST)
        pc, _ = _ptr(class_vi)
        pf, _ = _ptr(flags)
        _ready(text, class_vi, flags)
        ticket = C.c_int64(-1)
        self._text_keep = text
        _check(lib().gs_match_submit_fasta(self.h, pt, n_bytes, int(n_lines), int(n_records), mem, first_read_no, pc, pf,
                                           C.byref(ticket)))
        return ticket.value

    def submit_fastq_ml(self, text, n_lines=None, first_read_no=0):
        """general FASTQ text (sequence / quality over any number of lines) starting at a descriptor line: -> (records matched,
        bytes they cover); the rest belongs in front of the next chunk (gs_match_submit_fastq_ml)"""
        pt, mem = _ptr(text)
        n_bytes = int(text.shape[0] if hasattr(text, "shape") else len(text))
        if n_lines is None:
            n_lines = int(np.count_nonzero(np.asarray(text) == 10)) if mem == MEM_HOST else int((text == 10).sum().item())
        n_rec, used, ticket = C.c_int64(0), C.c_int64(0), C.c_int64(-1)
        _ready(text)
        _check(lib().gs_match_submit_fastq_ml(self.h, pt, n_bytes, n_lines, mem, first_read_no, None, None, C.byref(n_rec), C.byref(used),
                                              None, C.byref(ticket)))
        return max(n_rec.value, 0), used.value

    def compact_text(self, with_probs=False, slot=0):
        """gs_match_compact_text: the reads of the last four-line chunk that matchRead returned true for, as afterMatch writes them,
        gathered on the device -> (bytes as numpy uint8, records)"""
        p, nb, nr, d = C.c_void_p(), C.c_int64(0), C.c_int64(0), C.c_int(0)
        _check(lib().gs_match_compact_text(self.h, int(with_probs), int(slot), C.byref(p), C.byref(nb), C.byref(nr)))
        _check(lib().gs_match_get_device(self.h, C.byref(d)))
        return _fetch_device(d.value, p, nb.value), nr.value

    def text_wait_copy(self, ticket):
        _check(lib().gs_match_text_wait_copy(self.h, ticket))

    def text_status(self):
        """(failed_ticket or -1, first_bad_record or -1, (reads, kmers, bases) accepted so far); synchronises"""
        ft, fb = C.c_int64(-1), C.c_int64(-1)
        tot = (C.c_int64 * 3)()
        _check(lib().gs_match_text_status(self.h, C.byref(ft), C.byref(fb), tot))
        return ft.value, fb.value, tuple(tot)

    def text_clear_error(self):
        _check(lib().gs_match_text_clear_error(self.h))

    def text_select(self, bank):
        """chunks of files that are read side by side go to different banks (independent refusal state and totals)"""
        _check(lib().gs_match_text_select(self.h, bank))

    def sync(self):
        _check(lib().gs_match_sync(self.h))

    # ---- DB-partitioned mode (device tensors with data_ptr())
    def encode(self, seq, offsets, pos_off, keys, n_reads):
        _ready(seq, offsets, pos_off, keys)
        _check(lib().gs_match_encode(self.h, C.c_void_p(seq.data_ptr()), C.c_void_p(offsets.data_ptr()), n_reads,
                                     C.c_void_p(pos_off.data_ptr()), C.c_void_p(keys.data_ptr())))

    def probe_keys(self, keys, nodes, n_keys):
        _ready(keys, nodes)
        _check(lib().gs_match_probe_keys(self.h, C.c_void_p(keys.data_ptr()), n_keys, C.c_void_p(nodes.data_ptr())))

    def encode_route(self, seq, offsets, pos_off, n_reads, n_parts, cap, send_keys, send_idx, nodes):
        """gs_match_encode_route: returns (per-owner slot counts, overflow flag)"""
        counts = (C.c_int64 * n_parts)()
        over = C.c_int(0)
        _ready(seq, offsets, pos_off, send_keys, send_idx, nodes)
        _check(lib().gs_match_encode_route(self.h, C.c_void_p(seq.data_ptr()), C.c_void_p(offsets.data_ptr()), n_reads,
                                           C.c_void_p(pos_off.data_ptr()), n_parts, cap, C.c_void_p(send_keys.data_ptr()),
                                           C.c_void_p(send_idx.data_ptr()), C.c_void_p(nodes.data_ptr()), counts, C.byref(over)))
        return list(counts), bool(over.value)

    def route_geometry(self, n_reads):
        """gs_match_route_geometry: (waves gs_match_encode_route launches for n_reads reads, slots per chunk)"""
        w, c = C.c_int32(0), C.c_int32(0)
        _check(lib().gs_match_route_geometry(self.h, n_reads, C.byref(w), C.byref(c)))
        return int(w.value), int(c.value)

    def unroute_region(self, idx, back, n, nodes):
        _ready(idx, back, nodes)
        _check(lib().gs_unroute_region(self.h, C.c_void_p(idx.data_ptr()), C.c_void_p(back.data_ptr()), n, C.c_void_p(nodes.data_ptr())))

    def route_keys(self, keys, n_keys, n_parts, send_keys, idx, nodes=None):
        """device counting sort of the valid keys by owner rank; returns the per-owner counts (python list).
        nodes (int32 tensor, optional): the unrouted positions get their node (miss / invalid) here already"""
        counts = (C.c_int64 * n_parts)()
        _ready(keys, send_keys, idx, nodes)
        _check(lib().gs_route_keys(self.h, C.c_void_p(keys.data_ptr()), n_keys, n_parts, C.c_void_p(send_keys.data_ptr()),
                                   C.c_void_p(idx.data_ptr()), counts, None if nodes is None else C.c_void_p(nodes.data_ptr())))
        return list(counts)

    def unroute_nodes(self, keys, idx, back, n_routed, nodes, n_keys):
        """keys = None: route_keys(..., nodes) has written the unrouted positions already"""
        _ready(keys, idx, back, nodes)
        _check(lib().gs_unroute_nodes(self.h, None if keys is None else C.c_void_p(keys.data_ptr()), C.c_void_p(idx.data_ptr()),
                                      C.c_void_p(back.data_ptr()), n_routed, C.c_void_p(nodes.data_ptr()), n_keys))

    def reduce(self, seq, offsets, pos_off, nodes, n_reads, first_read_no=0, class_vi=None, flags=None):
        _ready(seq, offsets, pos_off, nodes, class_vi, flags)
        _check(lib().gs_match_reduce(self.h, C.c_void_p(seq.data_ptr()), C.c_void_p(offsets.data_ptr()), n_reads,
                                     first_read_no, C.c_void_p(pos_off.data_ptr()), C.c_void_p(nodes.data_ptr()),
                                     None if class_vi is None else C.c_void_p(class_vi.data_ptr()),
                                     None if flags is None else C.c_void_p(flags.data_ptr())))

    def segments(self, seq, offsets):
        """Kraken-style segments of a host batch: (seg_off uint64[n+1], codes int32[], starts int32[], lens int32[])"""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        if len(seq) == 0:
            seq = np.zeros(1, dtype=np.uint8)
        seg_off = np.zeros(n + 1, dtype=np.uint64)
        _check(lib().gs_match_segments(self.h, seq.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p), n,
                                       MEM_HOST, seg_off.ctypes.data_as(C.c_void_p)))
        total = int(seg_off[-1])
        codes = np.zeros(total, dtype=np.int32)
        starts = np.zeros(total, dtype=np.int32)
        _check(lib().gs_match_segments_fetch(self.h, codes.ctypes.data_as(C.c_void_p), starts.ctypes.data_as(C.c_void_p)))
        # a run ends where the next run of the same read starts, the last one at L - k + 1
        lens = np.zeros(total, dtype=np.int32)
        if total:
            nxt = np.empty(total, dtype=np.int64)
            nxt[:-1] = starts[1:]
            last = seg_off[1:][seg_off[1:] > seg_off[:-1]].astype(np.int64) - 1
            rl = (offsets[1:] - offsets[:-1]).astype(np.int64)[seg_off[1:] > seg_off[:-1]]
            nxt[last] = rl - self.store.k + 1
            lens = (nxt - starts).astype(np.int32)
        return seg_off, codes, starts, lens

    def finish(self):
        nv = self.store.n_values
        t = np.zeros((nv, N_COLS), dtype=np.int64)
        d = np.zeros((nv, N_DCOLS), dtype=np.float64)
        _check(lib().gs_match_finish(self.h, t.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p)))
        return t, d

    def max_counts(self):
        """(n_values + 1) x max_kmer_res_counts int16: largest per-k-mer hit counts per value index, last row = total"""
        n = self.config.max_kmer_res_counts
        out = np.zeros((self.store.n_values + 1, max(n, 1)), dtype=np.int16)
        _check(lib().gs_match_max_counts(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def reset(self):
        _check(lib().gs_match_reset(self.h))

    def max_contig_reads(self):
        """per value index: the read number that holds the longest contig so far (-1: none); synchronises"""
        out = np.zeros(self.store.n_values, dtype=np.int64)
        _check(lib().gs_match_max_contig_reads(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def device_state(self):
        """raw device pointers of the accumulators: dict(sums, max_keys, dsums, bitmap, bitmap_words)"""
        s, m, d, b = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        w = C.c_int64(0)
        _check(lib().gs_match_device_state(self.h, C.byref(s), C.byref(m), C.byref(d), C.byref(b), C.byref(w)))
        return dict(sums=s.value, max_keys=m.value, dsums=d.value, bitmap=b.value, bitmap_words=w.value)

    def or_bitmap(self, parts_ptr, n_parts):
        _check(lib().gs_match_or_bitmap(self.h, C.c_void_p(parts_ptr), n_parts))

    def kernel_time(self):
        n, ms = C.c_int64(0), C.c_double(0)
        _check(lib().gs_match_kernel_time(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def close(self):
        if getattr(self, "h", None):
            lib().gs_match_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBloomFilter:
    """device copy of a reference index filter (bits + hash factors replicated exactly)."""

    def __init__(self, kind, bits, hash_factors, words, n_hashes=None, device=0):
        hf = np.ascontiguousarray(hash_factors, dtype=np.int64)
        w = np.ascontiguousarray(words, dtype=np.uint64)
        if n_hashes is None:
            n_hashes = len(hf)
        self.h = C.c_void_p()
        _check(lib().gs_bloom_create(C.byref(self.h), device, kind, bits, n_hashes, hf.ctypes.data_as(C.c_void_p),
                                     w.ctypes.data_as(C.c_void_p), len(w)))

    @classmethod
    def build(cls, kmers, expected_insertions=None, fpp=1e-8, device=0, kind=BLOOM_XOR):
        """gs_bloom_build: the XOR (or Murmur) index filter over `kmers` (numpy int64 or a device tensor), sized like the
        reference's (BloomIndexGoal: indexBloomFilterFpp, default 1e-8) and filled on the device"""
        pk, mem = _ptr(kmers)
        n = int(kmers.shape[0] if hasattr(kmers, "shape") else len(kmers))
        _ready(kmers)
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        _check(lib().gs_bloom_build(C.byref(self.h), device, kind, pk, n, mem, int(expected_insertions or max(n, 1)), float(fpp)))
        return self

    @property
    def device(self):
        d = C.c_int(0)
        _check(lib().gs_filter_get_device(self.h, C.byref(d)))
        return d.value

    def get(self, with_words=True):
        """-> (bits, hash_factors int64[], words uint64[] or None)"""
        bits, nh = C.c_int64(0), C.c_int32(0)
        _check(lib().gs_bloom_get(self.h, C.byref(bits), C.byref(nh), None, None, 0))
        hf = np.zeros(nh.value, dtype=np.int64)
        words = np.zeros((bits.value + 63) // 64, dtype=np.uint64) if with_words else None
        _check(lib().gs_bloom_get(self.h, None, None, hf.ctypes.data_as(C.c_void_p),
                                  None if words is None else words.ctypes.data_as(C.c_void_p), 0 if words is None else len(words)))
        return bits.value, hf, words

    def close(self):
        if getattr(self, "h", None):
            lib().gs_bloom_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastqBloomFilter:
    def __init__(self, k, bloom, min_pos_count=1, positive_ratio=0.2, profile=False):
        self.k, self.bloom, self.min_pos_count, self.positive_ratio = k, bloom, min_pos_count, positive_ratio
        self.profile = profile

    def submit(self, seq, offsets, accept, n_reads=None):
        ps, mem = _ptr(seq)
        po, _ = _ptr(offsets)
        pa, _ = _ptr(accept)
        if n_reads is None:
            n_reads = (offsets.shape[0] if hasattr(offsets, "shape") else len(offsets)) - 1
        _ready(seq, offsets, accept)
        _check(lib().gs_filter_submit(self.bloom.h, self.k, self.min_pos_count, self.positive_ratio, ps, po, n_reads,
                                      mem, pa, int(self.profile)))

    def submit_text(self, text, accept, n_lines=None, newlines=None):
        """raw FASTQ text of whole 4-line records (gs_filter_submit_text); returns the ticket"""
        if isinstance(text, (bytes, bytearray)):
            text = np.frombuffer(bytes(text), dtype=np.uint8)
        if n_lines is None:
            n_lines = int((text == 10).sum())
        n_bytes = int(text.shape[0])
        pt, mem = _ptr(text) if n_bytes else (None, MEM_HOST)
        pa, _ = _ptr(accept)
        pn, _ = _ptr(newlines)
        _ready(text, accept, newlines)
        ticket = C.c_int64(-1)
        self._text_keep = text
        _check(lib().gs_filter_submit_text(self.bloom.h, self.k, self.min_pos_count, self.positive_ratio, pt, n_bytes,
                                           int(n_lines), mem, pa, pn, int(self.profile), C.byref(ticket)))
        return ticket.value

    def submit_fasta(self, text, accept, n_records=None, n_lines=None):
        """raw FASTA text of whole records (gs_filter_submit_fasta): accept[r] per record; -> read lengths"""
        text = np.frombuffer(bytes(text), dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text
        if n_lines is None:
            n_lines = int((text == 10).sum())
        if n_records is None:
            t = np.asarray(text)
            n_records = int(np.count_nonzero((t == 62) & np.concatenate(([True], t[:-1] == 10))))
        ticket = C.c_int64(-1)
        self._text_keep = text
        _check(lib().gs_filter_submit_fasta(self.bloom.h, self.k, self.min_pos_count, self.positive_ratio, _ptr(text)[0], int(text.shape[0]),
                                            int(n_lines), int(n_records), MEM_HOST, _ptr(accept)[0], None, C.byref(ticket)))
        return self._read_lengths(n_records)

    def submit_fastq_ml(self, text, accept, n_lines=None):
        """general FASTQ text starting at a descriptor line (gs_filter_submit_fastq_ml): -> (records filtered or -1 when refused,
        bytes they cover, read lengths)"""
        text = np.frombuffer(bytes(text), dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else text
        if n_lines is None:
            n_lines = int((text == 10).sum())
        n_rec, used, ticket = C.c_int64(0), C.c_int64(0), C.c_int64(-1)
        self._text_keep = text
        _check(lib().gs_filter_submit_fastq_ml(self.bloom.h, self.k, self.min_pos_count, self.positive_ratio, _ptr(text)[0],
                                               int(text.shape[0]), int(n_lines), MEM_HOST, _ptr(accept)[0], None, C.byref(n_rec),
                                               C.byref(used), None, C.byref(ticket)))
        return n_rec.value, used.value, (self._read_lengths(n_rec.value) if n_rec.value > 0 else np.zeros(0, dtype=np.int64))

    def _read_lengths(self, n_records):
        b = np.zeros(n_records + 1, dtype=np.uint64)
        if n_records > 0:
            _check(lib().gs_filter_text_read_bounds(self.bloom.h, _ptr(b)[0]))
        else:
            _check(lib().gs_filter_sync(self.bloom.h))
        return np.diff(b.astype(np.int64))

    def text_status(self):
        ft, fb = C.c_int64(-1), C.c_int64(-1)
        tot = (C.c_int64 * 3)()
        _check(lib().gs_filter_text_status(self.bloom.h, C.byref(ft), C.byref(fb), tot))
        return ft.value, fb.value, tuple(tot)

    def compact_text(self, which=1, with_probs=False, slot=0):
        """gs_filter_compact_text: the accepted (which = 1) / other records of the last four-line chunk as ReadEntry.write writes them,
        gathered on the device -> (bytes as numpy uint8, records)"""
        p, nb, nr = C.c_void_p(), C.c_int64(0), C.c_int64(0)
        _check(lib().gs_filter_compact_text(self.bloom.h, int(which), int(with_probs), int(slot), C.byref(p), C.byref(nb), C.byref(nr)))
        return _fetch_device(self.bloom.device, p, nb.value), nr.value

    def text_reset(self, clear_totals=False):
        _check(lib().gs_filter_text_reset(self.bloom.h, int(clear_totals)))

    def accept_reads(self, seq, offsets):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        acc = np.zeros(len(offsets) - 1, dtype=np.uint8)
        if len(seq) == 0:
            seq = np.zeros(1, dtype=np.uint8)
        self.submit(seq, offsets, acc)
        return acc

    def sync(self):
        _check(lib().gs_filter_sync(self.bloom.h))

    def kernel_time(self):
        n, ms = C.c_int64(0), C.c_double(0)
        _check(lib().gs_filter_kernel_time(self.bloom.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value
