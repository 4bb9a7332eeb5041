import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import genestrip_amd as ga
from genestrip_amd import synth, host
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n)
tmp = tempfile.mkdtemp(prefix="gsb_")
plain = os.path.join(tmp, "r.fastq")
bench._fastq_text(seq, n).tofile(plain)
bz = os.path.join(tmp, "r.bgzf.fastq.gz")
bench._write_gz(plain, bz, True, 16)
for rep in range(3):
    t0 = time.perf_counter(); t, _, tot = host.match_files(store, [bz]); dt = time.perf_counter() - t0
    print(f"device inflate: {dt*1e3:.1f} ms = {n*150/dt/1e9:.2f} Gbp/s reads {tot.reads}", flush=True)
ko = os.path.join(tmp, "kraken.txt")
for env, label in (("1", "device inflate"), ("0", "host decoders")):  # with Kraken-style per-read lines: the text comes back once per feed
    os.environ["GS_DEVICE_INFLATE"] = env
    for rep in range(2):
        if os.path.exists(ko):
            os.remove(ko)
        t0 = time.perf_counter(); host.match_files(store, [bz], kraken_out_path=ko, taxids=[f"t{i}" for i in range(db.n_values)]); dt = time.perf_counter() - t0
        print(f"{label}, per-read lines written: {dt*1e3:.1f} ms = {n*150/dt/1e9:.2f} Gbp/s ({os.path.getsize(ko) / 1e6:.0f} MB)", flush=True)
os.environ["GS_DEVICE_INFLATE"] = "0"
t0 = time.perf_counter(); t2, _, tot = host.match_files(store, [bz]); dt = time.perf_counter() - t0
print(f"host inflate: {dt*1e3:.1f} ms = {n*150/dt/1e9:.2f} Gbp/s", np.array_equal(t, t2), flush=True)
shutil.rmtree(tmp)
