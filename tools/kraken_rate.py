"""end-to-end rate of gs_host_match_files with per-read outputs (Kraken-style lines, filtered FASTQ) (developer tool)"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
gz = ".gz" if len(sys.argv) > 2 and sys.argv[2] == "gz" else ""  # compressed outputs (multi-member gzip)
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n)
d = tempfile.mkdtemp(prefix="gskr")
path = os.path.join(d, "reads.fastq")
L = 150
blk = seq.tobytes()
with open(path, "wb") as f:
    f.write(b"".join(b"@r%d\n" % i + blk[i * L:(i + 1) * L] + b"\n+\n" + b"I" * L + b"\n" for i in range(n)))
for label, kw in (("table only", {}), ("kraken out", dict(kraken_out_path=os.path.join(d, "k.out" + gz), taxids=db.taxids)),
                  ("filtered fastq", dict(filtered_path=os.path.join(d, "f.fastq" + gz))),
                  ("both", dict(kraken_out_path=os.path.join(d, "k.out" + gz), taxids=db.taxids, filtered_path=os.path.join(d, "f.fastq" + gz)))):
    t0 = time.perf_counter()
    _, _, tot = host.match_files(store, [path], **kw)
    dt = time.perf_counter() - t0
    print(f"{label:15s}: {dt:.2f} s -> {n * 150 / dt / 1e9:.3f} Gbp/s (parse {tot.seconds_parse:.2f} s, gpu {tot.seconds_gpu:.2f} s)", flush=True)
shutil.rmtree(d)
