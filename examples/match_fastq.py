"""The `match` goal end to end with the C++ host layer: FASTQ files (plain or .gz) -> per-taxid CSV in the reference's
layout (ResultReporter.printMatchResult), plus optional Kraken-style lines and filtered reads.

    python examples/match_fastq.py --store demo.gsstore --csv out.csv reads_1.fastq.gz reads_2.fastq.gz

Without --store a synthetic store is built (genestrip_amd.synth) and synthetic reads are written to a temporary file,
so the script runs as is on any MI355X box:

    python examples/match_fastq.py --demo

A real deployment creates the store once from the reference's database -- KMerStore.visit hands (k-mer, value index)
pairs and the tree's parent links to gs_db_create -- and keeps it with DeviceKMerStore.save().
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fastq", nargs="*")
    ap.add_argument("--store", help="native store file written by DeviceKMerStore.save()")
    ap.add_argument("--demo", action="store_true", help="synthetic store + synthetic reads")
    ap.add_argument("--csv", default="match.csv")
    ap.add_argument("--kraken-out")
    ap.add_argument("--filtered", help="FASTQ of the reads with a hit (a name ending in .gz is written as BGZF)")
    ap.add_argument("--with-probs", action="store_true", help="written reads keep their quality lines (withProbs)")
    args = ap.parse_args()
    if not args.demo and not (args.store and args.fastq):
        ap.error("give --demo, or --store and FASTQ files")

    db = synth.SynthDB()  # (the demo taxonomy also labels the rows of a loaded store in this example)
    if args.store and not args.demo:
        store = ga.DeviceKMerStore.load(args.store)
    else:
        store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    paths = list(args.fastq)
    tmp = None
    if args.demo:
        n, L = 1_000_000, 150
        seq, off = synth.reads_host(db.genomes, n)
        tmp = tempfile.NamedTemporaryFile(suffix=".fastq", delete=False)
        blk = seq.tobytes()
        tmp.write(b"".join(b"@demo%d\n" % i + blk[i * L:(i + 1) * L] + b"\n+\n" + b"F" * L + b"\n" for i in range(n)))
        tmp.close()
        paths = [tmp.name]

    table, dtable, tot = host.match_files(store, paths, kraken_out_path=args.kraken_out, filtered_path=args.filtered,
                                          taxids=db.taxids if args.kraken_out else None, with_probs=args.with_probs)
    print(f"{tot.reads} reads, {tot.bps} bases in {tot.seconds_total:.2f} s "
          f"({tot.bps / max(tot.seconds_total, 1e-9) / 1e9:.2f} Gbp/s end to end)")
    db_kmers = np.bincount(db.value_idx, minlength=db.n_values).astype(np.int64)
    host.write_csv(args.csv, db.parent_vi, db.taxids, db_kmers, int(db_kmers.sum()), table, dtable, tot)
    print(f"wrote {args.csv}")
    for line in open(args.csv).read().split("\n")[:4]:
        print("  " + line[:150])
    if tmp:
        os.unlink(tmp.name)
    store.close()


if __name__ == "__main__":
    main()
