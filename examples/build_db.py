"""DB construction end to end on the GPU: genome FASTA files -> (k-mer, lowest common ancestor) arrays -> store layout -> native
store file, the way FillDBGoal + DBGoal do it with two walks over the collection on the CPU.

    python examples/build_db.py --out my.gsstore --tree nodes.tsv genomes/*.fasta

nodes.tsv: one line per tax node, `taxid <tab> parent taxid` (the root's parent is itself or empty); every FASTA file is one
genome whose records all belong to the taxid in its first header field `>taxid|...` or, with --by-file, to the taxid that is the
file's basename.  Without arguments a synthetic collection is used (genestrip_amd.synth), so the script runs as is:

    python examples/build_db.py --demo

(What decides which regions count in the reference -- accession maps, RefSeq categories, per-taxid limits -- is host logic and not
part of this example; java/src/.../refseq/GpuStoreFastaReader.java shows the reference-side reader.)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402


def read_fasta(path):
    """-> [(header without '>', sequence bytes without line ends)]"""
    out, name, parts = [], None, []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    out.append((name, b"".join(parts)))
                name, parts = line[1:].strip().decode("latin1"), []
            elif name is not None:
                parts.append(line.rstrip(b"\r\n"))
    if name is not None:
        out.append((name, b"".join(parts)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fasta", nargs="*")
    ap.add_argument("--tree")
    ap.add_argument("--by-file", action="store_true")
    ap.add_argument("--out", default="built.gsstore")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--max-dust", type=int, default=-1)
    ap.add_argument("--demo", action="store_true")
    args = ap.parse_args()
    if not args.demo and not (args.tree and args.fasta):
        ap.error("give --demo, or --tree and FASTA files")

    t0 = time.time()
    if args.demo:
        db = synth.SynthDB(k=args.k, genera=5, species_per_genus=8, genome_len=200_000)
        parent_vi, n_values = db.parent_vi, db.n_values
        regions = [(db.genomes[i].tobytes(), int(db.species_vi[i])) for i in range(db.genomes.shape[0])]
        taxids = db.taxids
    else:
        ids, parents = [], {}
        for line in open(args.tree):
            f = line.rstrip("\n").split("\t")
            if f and f[0]:
                ids.append(f[0])
                parents[f[0]] = f[1] if len(f) > 1 and f[1] and f[1] != f[0] else None
        vi = {t: i for i, t in enumerate(ids)}
        parent_vi = np.array([vi[parents[t]] if parents[t] is not None else -1 for t in ids], dtype=np.int32)
        n_values, taxids = len(ids), ids
        regions = []
        for path in args.fasta:
            for name, seq in read_fasta(path):
                tax = os.path.splitext(os.path.basename(path))[0] if args.by_file else name.split("|")[0].split()[0]
                if tax in vi:
                    regions.append((seq, vi[tax]))
    t_read = time.time() - t0

    seq = np.frombuffer(b"".join(s for s, _ in regions), dtype=np.uint8)
    off = np.cumsum([0] + [len(s) for s, _ in regions]).astype(np.uint64)
    nodes = np.array([n for _, n in regions], dtype=np.int32)
    t0 = time.time()
    b = ga.DeviceDbBuilder(args.k, n_values, parent_vi, max_dust=args.max_dust)
    b.add(seq, off, nodes, update=False)  # FillDBGoal: the k-mers of these regions are stored ...
    b.add(seq, off, nodes, update=True)   # ... DBGoal: and every region moves the k-mers it shares towards the common ancestor
    n_kmers = b.finish_count()
    store = b.to_store()                  # the layout (records, overflow table, gate) without leaving the GPU
    b.close()
    t_build = time.time() - t0
    info = store.info
    store.save(args.out)
    print("%d regions, %.1f Mbases read in %.2f s; %d distinct k-mers with LCA values + store layout in %.2f s "
          "(%.0f Mbases/s); %d k-mers in records, %d in the overflow table; %s written (%.1f MB)" %
          (len(regions), len(seq) / 1e6, t_read, n_kmers, t_build, len(seq) / t_build / 1e6, info.n_in_records,
           info.n_stored - info.n_in_records, args.out, os.path.getsize(args.out) / 1e6))
    if args.demo:  # the store answers reads drawn from its genomes
        rs, ro = synth.reads_host(db.genomes, 20000, read_len=150, seed=1)
        m = ga.FastqKMerMatcher(store)
        m.submit(rs, ro.astype(np.uint64), 0)
        table, _ = m.finish()
        m.close()
        top = np.argsort(-table[:, 0])[:3]
        print("20000 reads from the genomes: %d classified; most reads: %s" %
              (int(table[:, 0].sum()), ", ".join("taxid %s: %d" % (taxids[i], table[i, 0]) for i in top)))
    store.close()


if __name__ == "__main__":
    main()
