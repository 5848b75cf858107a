"""Run on the GPU box after rocprofv3: reduce the csv outputs to the rows of our kernels (small files for profiles/)."""
import csv
import glob
import sys

out_dir = sys.argv[1]
for path in glob.glob(out_dir + "/**/*.csv", recursive=True):
    rows = list(csv.reader(open(path, newline="")))
    if not rows:
        continue
    keep = [rows[0]] + [r for r in rows[1:] if any("zhip_" in c for c in r)]
    if "stats" in path:          # keep the whole (short) stats table, top 15 by time
        keep = rows[:16]
    with open(path.replace(".csv", ".zhip.csv"), "w", newline="") as fh:
        csv.writer(fh).writerows(keep)
    print(path, len(rows), "->", len(keep))
