"""rate of the device decoder for single-member gzip on FASTQ (developer tool): python tools/gunzip_device_rate.py [reads] [level] [keep_dir]
(keep_dir: the generated files stay there and are used again by the next run with the same arguments)
wall time of gs_gunzip_device (H2D of the compressed bytes + the five kernels + D2H of the text); run under rocprofv3 --kernel-trace --stats
for the kernels' own times"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    keep = sys.argv[3] if len(sys.argv) > 3 else None
    with tempfile.TemporaryDirectory() as tmp:  # gzip(1) itself: one member, one thread
        if keep:
            os.makedirs(keep, exist_ok=True)
            tmp = keep
        p = os.path.join(tmp, "r_%d_%d.fastq" % (n, level))
        if not os.path.exists(p + ".gz"):
            db = synth.SynthDB()
            seq, off = synth.reads_host(db.genomes, n)
            open(p, "wb").write(bench._fastq_text(seq, n).tobytes())
            subprocess.run(["gzip", "-%d" % level, "-k", p], check=True)
        text = open(p, "rb").read()
        data = open(p + ".gz", "rb").read()
    print(f"{len(text) / 1e6:.0f} MB text, {len(data) / 1e6:.0f} MB gzip -{level}", flush=True)
    for _ in range(3):
        t0 = time.perf_counter()
        out, info = ga.gunzip_device(data, len(text))
        dt = time.perf_counter() - t0
        print(f"gs_gunzip_device: {dt * 1e3:.1f} ms wall = {len(text) / dt / 1e9:.2f} GB/s of text incl. copies; segments {info[0]} of {info[1]} chunks", flush=True)
    assert out.tobytes() == text
    print("text equals the input")
