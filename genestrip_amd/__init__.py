"""genestrip_amd -- MI355X-native engine for Genestrip's `match` / `filter` hot path.

The product is ``libgsgpu.so`` (C ABI, include/gsgpu.h: hand-written gfx950 kernels).  This package is the
thin ctypes binding used by tests and bench.py; it mirrors the reference's operator interface for the path
(FastqKMerMatcher.runMatcher, FastqBloomFilter.runFilter).  There is no CPU fallback: importing works
without a GPU (so the C ABI can be inspected), but every compute entry point fails loudly without one.
"""
from .binding import (  # noqa: F401
    GsError, lib, lib_path, device_count, abi_version, DeviceKMerStore, MatchConfig, FastqKMerMatcher,
    DeviceBloomFilter, FastqBloomFilter, DeviceDbBuilder, N_COLS, N_DCOLS, N_SUMS, COLS, MEM_HOST, MEM_DEVICE,
    F_FOUND, F_RETURNED, F_COUNTED, BLOOM_XOR, BLOOM_MURMUR, BLOOM_BLOCKED, ABI_SYMBOLS, calibrate, bgzf_members, inflate_members, gunzip_device, deflate_device, deflate_reference, BGZF_EOF, DeviceDeflater, deflate_bound,
    CAL_VALU_PURE, CAL_VALU_MIX, CAL_SALU, CAL_VALU_SALU, CAL_VMEM_BYTES, CAL_VMEM_WORDS, CAL_VMEM_SHARED_LINES,
    CAL_VMEM_SCATTERED, CAL_RANDOM_LINES,
)

__all__ = [
    "GsError", "lib", "lib_path", "device_count", "abi_version", "DeviceKMerStore", "MatchConfig",
    "FastqKMerMatcher", "DeviceBloomFilter", "FastqBloomFilter", "DeviceDbBuilder",
]
