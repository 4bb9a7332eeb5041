"""Rate of the device DEFLATE writer (gs_deflater_pack): n reads of rewritten FASTQ ('~' qualities, or quality lines with --probs)
resident in HBM -> BGZF members in page-locked host memory.  `rocprofv3 --kernel-trace --stats -- python3 tools/deflate_rate.py` for
the kernels' own times."""
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    probs = "--probs" in sys.argv
    db = synth.SynthDB()
    seq, off = synth.reads_host(db.genomes, n)
    L = 150
    rec = 2 + 8 + 1 + L + 3 + L + 1
    a = np.empty((n, rec), dtype=np.uint8)
    a[:, 0], a[:, 1] = ord("@"), ord("r")
    idx = np.arange(n, dtype=np.int64)
    for d in range(8):
        a[:, 9 - d] = (idx % 10 + 48).astype(np.uint8)
        idx //= 10
    a[:, 10] = 10
    a[:, 11:11 + L] = seq.reshape(n, L)
    a[:, 11 + L:14 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    if probs:
        a[:, 14 + L:rec - 1] = np.random.default_rng(1).choice(np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8), (n, L))
    else:
        a[:, 14 + L:rec - 1] = ord("~")
    a[:, rec - 1] = 10
    text = a.reshape(-1)
    d_text = torch.from_numpy(text).cuda()
    out = torch.empty(ga.deflate_bound(len(text)), dtype=torch.uint8).pin_memory().numpy()
    d = ga.DeviceDeflater()
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        m = d.pack(d_text, len(text), out)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    z1 = len(zlib.compress(text[:20_000_000].tobytes(), 1)) / 20e6
    print(f"{len(text) / 1e6:.0f} MB of text -> {m / 1e6:.0f} MB ({len(text) / m:.2f} x; zlib level 1: {1 / z1:.2f} x) in {best * 1e3:.1f} ms = "
          f"{len(text) / best / 1e9:.1f} GB/s of text = {n * L / best / 1e9:.1f} Gbp/s")
    import gzip
    assert gzip.decompress(out[:m].tobytes() + ga.BGZF_EOF) == text.tobytes()
    print("zlib inflates it to the text")
