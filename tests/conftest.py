import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bgzf(data, block=65280, level=6, eof_marker=True, extra_subfield=False):
    """BGZF as bgzip / htslib write it (SAM spec 4.1): gzip members of at most 64 KiB of text whose extra field says
    how long the member is"""
    import struct
    import zlib
    out = []
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_marker else [])
    for c in chunks:
        z = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = z.compress(c) + z.flush()
        extra = (b"XY" + struct.pack("<H", 3) + b"abc" if extra_subfield else b"")
        xlen = len(extra) + 6
        bsize = 12 + xlen + len(body) + 8 - 1
        assert bsize < 65536
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", xlen) + extra + b"BC" + struct.pack("<HH", 2, bsize) +
                   body + struct.pack("<II", zlib.crc32(c), len(c)))
    return b"".join(out)
