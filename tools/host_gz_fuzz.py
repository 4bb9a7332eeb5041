"""randomised comparison of the file pipeline over compressed FASTQ with the same pipeline over the plain file (developer tool; on
the GPU box):   python tools/host_gz_fuzz.py [cases] [seed]
Four-line FASTQ of random shape (read count, read lengths, with / without a final newline) written plain, as one gzip member, as
several members and as BGZF; `match` (table, totals, Kraken-style lines) and `filter` (accepted reads written) over each, with the
device decoders' geometry (batch / slice / chunk sizes, whole / per-batch upload, first span) drawn at random.  Every compressed
variant must give what the plain file gives."""
import os
import struct
import sys
import tempfile
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth, host  # noqa: E402
import bench  # noqa: E402

KNOBS = ("GS_GUNZIP_CHUNK", "GS_GUNZIP_SLOTS", "GS_GUNZIP_FIND_CHUNK", "GS_GUNZIP_WHOLE_MAX", "GS_HOST_BGZF_TEXT", "GS_HOST_GUNZIP_FIRST", "GS_GUNZIP_FIND_EVERY")


def fastq(rng, db):
    n = int(rng.integers(1, 40000))
    mode = int(rng.integers(0, 3))
    seq, off = synth.reads_host(db.genomes, n)
    seq = np.asarray(seq)
    recs = []
    for i in range(n):
        s = bytes(seq[off[i]:off[i + 1]])
        if mode == 1:
            s = s[:int(rng.integers(31, 151))]
        elif mode == 2 and rng.random() < 0.01:
            s = s * int(rng.integers(2, 30))
        q = bytes(rng.choice(np.frombuffer(b"FFFF:,#", dtype=np.uint8), len(s)))
        recs.append(b"@read%d some words\n" % i + s + b"\n+\n" + q + b"\n")
    text = b"".join(recs)
    if rng.random() < 0.3:
        text = text[:-1]  # no final newline
    return text, n


def gz_member(rng, data):
    c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, 31, int(rng.integers(4, 10)))
    out = bytearray()
    at = 0
    while at < len(data):
        n = int(rng.integers(1, 1 + max(1, len(data) // 2)))
        out += c.compress(data[at:at + n])
        at += n
        if rng.random() < 0.2:
            out += c.flush(zlib.Z_SYNC_FLUSH)
    return bytes(out + c.flush())


def bgzf(rng, data):
    out, at = bytearray(), 0
    while at < len(data):
        piece = data[at:at + int(rng.integers(1000, 65281))]
        at += len(piece)
        c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(piece), len(piece))
    return bytes(out + b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")


def main(cases=30, seed=1):
    rng = np.random.default_rng(seed)
    db = synth.SynthDB()
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    bloom, _ = bench._index_filter(ga, synth, torch, torch.device("cuda:0"), db)
    taxids = ["t%d" % i for i in range(db.n_values)]
    tmp = tempfile.mkdtemp(prefix="gsfz_")
    for case in range(cases):
        text, n = fastq(rng, db)
        for k in KNOBS:
            os.environ.pop(k, None)
        plain = os.path.join(tmp, "p.fastq")
        open(plain, "wb").write(text)
        kr0, fl0 = os.path.join(tmp, "k0.txt"), os.path.join(tmp, "f0.fastq")
        t0, _, tot0 = host.match_files(store, [plain], kraken_out_path=kr0, taxids=taxids)
        ftot0 = host.filter_files(bloom, 31, [plain], 1, 0.2, filtered_path=fl0)
        cut = int(rng.integers(0, len(text)))
        cut = text.rfind(b"\n@read", 0, cut) + 1 if b"\n@read" in text[:cut] else 0
        variants = {"one member": gz_member(rng, text), "bgzf": bgzf(rng, text)}
        if cut > 0:
            variants["two members"] = gz_member(rng, text[:cut]) + gz_member(rng, text[cut:])
        env = {}
        if rng.random() < 0.7:
            env["GS_GUNZIP_CHUNK"] = str(int(rng.choice([4096, 16384, 65536])))
        if rng.random() < 0.7:
            env["GS_GUNZIP_SLOTS"] = str(int(rng.choice([1, 3, 16, 256])))
        if rng.random() < 0.5:
            env["GS_HOST_BGZF_TEXT"] = str(int(rng.choice([65536, 300000, 4 << 20])))
        if rng.random() < 0.4:
            env["GS_GUNZIP_WHOLE_MAX"] = "0"
        if rng.random() < 0.5:
            env["GS_HOST_GUNZIP_FIRST"] = str(int(rng.choice([0, 4096, 100000])))
        os.environ.update(env)
        for name, z in variants.items():
            p = os.path.join(tmp, "c.fastq.gz")
            open(p, "wb").write(z)
            kr, fl = os.path.join(tmp, "k.txt"), os.path.join(tmp, "f.fastq")
            t1, _, tot1 = host.match_files(store, [p])
            t2, _, tot2 = host.match_files(store, [p], kraken_out_path=kr, taxids=taxids)
            ftot = host.filter_files(bloom, 31, [p], 1, 0.2, filtered_path=fl)
            ok = (np.array_equal(t0, t1) and np.array_equal(t0, t2) and tot1.reads == tot0.reads == tot2.reads and open(kr, "rb").read() == open(kr0, "rb").read()
                  and ftot.filtered_reads == ftot0.filtered_reads and open(fl, "rb").read() == open(fl0, "rb").read())
            if not ok:
                print("MISMATCH in case %d (%s): %d reads, %d bytes of text, env %s; tables %s %s, reads %d %d %d, kraken %s, filter %d %d %s" % (
                    case, name, n, len(text), env, np.array_equal(t0, t1), np.array_equal(t0, t2), tot0.reads, tot1.reads, tot2.reads,
                    open(kr, "rb").read() == open(kr0, "rb").read(), ftot0.filtered_reads, ftot.filtered_reads, open(fl, "rb").read() == open(fl0, "rb").read()), flush=True)
                os.makedirs("gpurun_out", exist_ok=True)
                open("gpurun_out/host_fuzz_case.gz", "wb").write(z)
                return 1
        if case % 5 == 4:
            print("%d cases" % (case + 1), flush=True)
    for k in KNOBS:
        os.environ.pop(k, None)
    print("all %d cases: every compressed variant gives what the plain file gives" % cases)
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
