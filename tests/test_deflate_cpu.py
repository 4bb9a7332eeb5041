"""The table builder of the device DEFLATE writer (genestrip_amd/csrc/gs_deflate_dev.hip), without a device: gs_deflate_host_reference
runs a plain CPU loop over the SAME Huffman-code construction, dynamic-header writer and member framing the kernels use, and zlib must
inflate what it writes to the input (a multi-member gzip file, as java.util.zip.GZIPInputStream reads it: B/io/StreamProvider.java:92-100)."""
import gzip
import zlib

import numpy as np
import pytest

import genestrip_amd as ga


def _fastq(n, read_len=150, seed=1, probs=False):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), read_len))
        q = bytes(rng.integers(35, 74, read_len, dtype=np.uint8)) if probs else b"~" * read_len
        out.append(b"@read%d/1 lane=3\n%s\n+\n%s\n" % (i, seq, q))
    return b"".join(out)


CASES = {
    "one byte": b"A",
    "short line": b"@r1\nACGT\n+\n~~~~\n",
    "fastq without probs": _fastq(600),
    "fastq with probs": _fastq(500, probs=True),
    "several pieces": _fastq(1500, seed=3),
    "a run": b"~" * 100000,
    "every byte value": bytes(range(256)) * 40,
    "random bytes": np.random.default_rng(7).integers(0, 256, 70000, dtype=np.uint8).tobytes(),
    "period three": b"ACG" * 30000,
    "exactly one piece": _fastq(400, seed=5)[:16384],
    "one piece and a byte": _fastq(400, seed=6)[:16385],
}


def piece_bytes(n):
    """text per member: about one piece per wave slot of the device, 16 .. 63 KiB (gd_piece_bytes)"""
    return max(16384, min(64512, ((n + 4095) // 4096 + 1023) // 1024 * 1024))


def n_members(n):
    return (n + piece_bytes(n) - 1) // piece_bytes(n)


@pytest.mark.parametrize("name", list(CASES))
def test_reference_members_inflate_to_the_input(name):
    data = CASES[name]
    comp = ga.deflate_reference(data).tobytes()
    members, reached = ga.bgzf_members(comp)
    assert reached == len(comp) and len(members) == n_members(len(data))
    assert sum(m[2] for m in members) == len(data)
    assert gzip.decompress(comp + ga.BGZF_EOF) == data
    at = 0
    for po, pl, isz, crc in members:  # every member on its own: raw deflate, CRC-32 and ISIZE as announced
        text = zlib.decompress(comp[po:po + pl], -15)
        assert len(text) == isz and zlib.crc32(text) == crc and text == data[at:at + isz]
        at += isz


def test_empty_input_gives_no_member():
    assert len(ga.deflate_reference(b"")) == 0


def test_fastq_compresses_at_least_as_well_as_zlib_level_1():
    data = _fastq(1500, seed=11)
    ours = len(ga.deflate_reference(data))
    theirs = len(zlib.compress(data, 1))
    assert ours < theirs, (ours, theirs)
