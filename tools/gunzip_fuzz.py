"""randomised comparison of the device decoder for single-member gzip with zlib (developer tool; on the GPU box):
    python tools/gunzip_fuzz.py [cases] [seed] [bgzf]      (bgzf: the same texts as BGZF members through gs_inflate_members)
Texts of several kinds (FASTQ-like, runs, far copies, bytes of a small / the full alphabet, mixtures), levels 1 .. 9 and zlib
strategies, one or several members, with the decoder's geometry (chunk, wave slots, finder chunk, whole / per-batch upload, symbol
room) drawn at random.  Every case must give zlib's text or a refusal (GS_E_UNSUPPORTED / GS_E_NOMEM) -- never other text."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402


def fastq_like(rng, n):
    acgt = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n * 100).tobytes()
    qual = rng.choice(np.frombuffer(b"FFFFFFF:,#", dtype=np.uint8), n * 100).tobytes()
    return b"".join(b"@r%07d/1\n" % i + acgt[i * 100:(i + 1) * 100] + b"\n+\n" + qual[i * 100:(i + 1) * 100] + b"\n" for i in range(n))


def make_text(rng):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return fastq_like(rng, int(rng.integers(1, 30000)))
    if kind == 1:
        return (b"A" * int(rng.integers(1, 5000)) + b"\n") * int(rng.integers(1, 2000))
    if kind == 2:
        return bytes(rng.integers(65, 91, int(rng.integers(1, 60000)), dtype=np.uint8)) * int(rng.integers(1, 40))
    if kind == 3:
        return bytes(rng.integers(0, 256, int(rng.integers(0, 400000)), dtype=np.uint8))
    if kind == 4:
        return bytes(rng.integers(32, 40, int(rng.integers(0, 2000000)), dtype=np.uint8))
    if kind == 5:
        parts = [make_text(rng) for _ in range(int(rng.integers(2, 4)))]
        return b"".join(parts)
    return fastq_like(rng, int(rng.integers(1, 3000))) + bytes(rng.integers(0, 256, int(rng.integers(0, 100000)), dtype=np.uint8)) + fastq_like(rng, int(rng.integers(1, 3000)))


def gz(rng, data):
    level = int(rng.integers(1, 10))
    strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 6))]
    c = zlib.compressobj(level, zlib.DEFLATED, 31, int(rng.integers(1, 10)), strategy)
    out = bytearray()
    at = 0
    while at < len(data):  # now and then a flush: empty stored blocks, byte-aligned block starts
        n = int(rng.integers(1, 1 + max(1, len(data) // 3)))
        out += c.compress(data[at:at + n])
        at += n
        if rng.random() < 0.3:
            out += c.flush(zlib.Z_SYNC_FLUSH if rng.random() < 0.7 else zlib.Z_FULL_FLUSH)
    out += c.flush()
    return bytes(out)


def main(cases=200, seed=1, mode=""):
    rng = np.random.default_rng(seed)
    if mode == "bgzf":
        return main_bgzf(cases, rng)
    knobs = ("GS_GUNZIP_CHUNK", "GS_GUNZIP_SLOTS", "GS_GUNZIP_FIND_CHUNK", "GS_GUNZIP_WHOLE_MAX", "GS_GUNZIP_RATIO", "GS_GUNZIP_ANY_BYTES", "GS_GUNZIP_FIND_EVERY")
    refused = 0
    for case in range(cases):
        members = [make_text(rng) for _ in range(1 if rng.random() < 0.8 else int(rng.integers(2, 4)))]
        z = b"".join(gz(rng, m) for m in members)
        want = b"".join(members)
        env = {}
        if rng.random() < 0.7:
            env["GS_GUNZIP_CHUNK"] = str(int(rng.choice([4096, 8192, 16384, 65536, 1 << 20])))
        if rng.random() < 0.6:
            env["GS_GUNZIP_SLOTS"] = str(int(rng.choice([1, 2, 3, 7, 16, 64, 1024])))
        if rng.random() < 0.4:
            env["GS_GUNZIP_FIND_CHUNK"] = str(int(rng.choice([4096, 8192, 32768])))
        if rng.random() < 0.4:
            env["GS_GUNZIP_WHOLE_MAX"] = "0"
        if rng.random() < 0.2:
            env["GS_GUNZIP_RATIO"] = str(int(rng.choice([2, 6, 40])))
        if rng.random() < 0.5:
            env["GS_GUNZIP_ANY_BYTES"] = "1"
        if rng.random() < 0.3:
            env["GS_GUNZIP_FIND_EVERY"] = "1"
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            got, info = ga.gunzip_device(z, len(want))
            ok = got.tobytes() == want
        except ga.GsError as e:
            if e.code in (-4, -2):  # unsupported / no memory: the host decoders' turn
                refused += 1
                ok = True
            else:
                ok = False
                print("case", case, "error", e.code, str(e)[:200], flush=True)
        if not ok:
            print("MISMATCH in case %d: %d members, %d -> %d bytes, env %s" % (case, len(members), len(z), len(want), env), flush=True)
            os.makedirs("gpurun_out", exist_ok=True)
            open("gpurun_out/fuzz_case.gz", "wb").write(z)  # (tools/gunzip_debug.py takes it from there)
            return 1
        if case % 20 == 19:
            print("%d cases, %d refused" % (case + 1, refused), flush=True)
    for k in knobs:
        os.environ.pop(k, None)
    print("all %d cases equal zlib (%d refused to the host decoders)" % (cases, refused))
    return 0


def main_bgzf(cases, rng):
    """members of at most 64 KiB of text, every level and strategy, several deflate blocks per member"""
    import struct
    for case in range(cases):
        text = make_text(rng)[:int(rng.integers(1, 3_000_000))]
        file_bytes, at = bytearray(), 0
        while at < len(text):
            n = int(rng.integers(1, 65281))
            data = text[at:at + n]
            at += n
            level = int(rng.integers(1, 10))
            strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 6))]
            c = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.integers(1, 10)), strategy)
            body = c.compress(data[:len(data) // 2]) + (c.flush(zlib.Z_FULL_FLUSH) if rng.random() < 0.3 else b"") + c.compress(data[len(data) // 2:]) + c.flush()
            if 18 + len(body) + 8 - 1 >= 65536:  # (does not compress: a stored member)
                c = zlib.compressobj(0, zlib.DEFLATED, -15)
                data = data[:60000]
                at = at - n + len(data)
                body = c.compress(data) + c.flush()
            file_bytes += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(data), len(data))
        file_bytes = bytes(file_bytes)
        members, reached = ga.bgzf_members(file_bytes)
        got, st = ga.inflate_members(file_bytes, members)
        if reached != len(file_bytes) or st.any() or got.tobytes() != text[:at]:
            print("MISMATCH in BGZF case %d: %d members, status %s" % (case, len(members), st[st != 0][:8]), flush=True)
            os.makedirs("gpurun_out", exist_ok=True)
            open("gpurun_out/fuzz_case.bgzf", "wb").write(file_bytes)
            return 1
        if case % 20 == 19:
            print("%d BGZF cases" % (case + 1), flush=True)
    print("all %d BGZF cases equal zlib" % cases)
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1, sys.argv[3] if len(sys.argv) > 3 else ""))
