"""the filter goal on a BGZF FASTQ file with the accepted reads written (developer tool): device inflate against the host decoders
    python tools/filter_bgzf_rate.py [reads]      (GS_HOST_TRACE=1: per-feed times on stderr)"""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import genestrip_amd as ga
from genestrip_amd import synth, host
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
db = synth.SynthDB(genera=25, species_per_genus=20)
bloom, _ = bench._index_filter(ga, synth, torch, torch.device("cuda:0"), db)
seq, off = synth.reads_host(db.genomes, n)
tmp = tempfile.mkdtemp(prefix="gsf_")
try:
    plain = os.path.join(tmp, "r.fastq")
    bench._fastq_text(seq, n).tofile(plain)
    bz = os.path.join(tmp, "r.bgzf.fastq.gz")
    bench._write_gz(plain, bz, True, 16)
    outp = os.path.join(tmp, "acc.fastq")
    gz = os.path.join(tmp, "r.fastq.gz")
    bench._write_gz(plain, gz, False, 16)
    for label, path, env in (("plain input", plain, "1"), ("bgzf, device inflate", bz, "1"), ("bgzf, host decoders", bz, "0"), ("one .gz stream, device inflate", gz, "1")):
        os.environ["GS_DEVICE_INFLATE"] = env
        for rep in range(3):
            if os.path.exists(outp):
                os.remove(outp)  # (truncating 0.6 GB of page cache is not part of the pipeline)
            t0 = time.perf_counter()
            tot = host.filter_files(bloom, 31, [path], 1, 0.2, filtered_path=outp)
            dt = time.perf_counter() - t0
            print(f"{label}: {dt * 1e3:.1f} ms = {n * 150 / dt / 1e9:.2f} Gbp/s, accepted {tot.filtered_reads}", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
