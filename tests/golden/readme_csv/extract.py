"""Regenerates human_virus_match_sample_rows.csv: the header line and the 13 result rows that the reference's README
prints for `genestrip human_virus match -f sample.fastq.gz` (README.md:168-181).  Data only; run in the build container:
    python tests/golden/readme_csv/extract.py /root/reference/README.md"""
import os
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("pos;level;name;rank;taxid;"))
rows = []
for l in lines[start:]:
    if l.startswith("```"):
        break
    rows.append(l)
assert len(rows) == 14
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "human_virus_match_sample_rows.csv")
open(out, "w").write("\n".join(rows) + "\n")
print(out, len(rows), "lines")
