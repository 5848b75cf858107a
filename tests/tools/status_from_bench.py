"""Markdown status table from ONE bench.py line (so that README.md / DESIGN.md quote the numbers the line carries, not remembered ones).
usage: python tests/tools/status_from_bench.py profiles/r03_bench_default.json"""
import json
import sys

line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def pct(x):
    return "%.2f %%" % (100.0 * x)


def kern(per):
    return ", ".join("%s %.2f ms (%s)" % (k.replace("zhip_", "").replace("_kernel", ""), v["avg_ms"], pct(v["frac"])) for k, v in per.items())


rows = []
r = line["roofline"]
cb = line.get("cpu_baseline", {})
rows.append(("[1] batch decompress, %d x 128 KiB level-3 frames" % line["config"]["frames_per_gpu"],
             "**%.1f GB/s** of uncompressed bytes, %.2f ms per step; end to end %s of the HBM peak (compressed + uncompressed bytes); per kernel, own bytes: %s"
             % (line["value"], line["ms_per_step"], pct(r["end_to_end"]["frac"]), kern(r["per_kernel"])),
             "%.1f GB/s (%s threads)" % (cb.get("value", 0), cb.get("cores", "?"))))
c = line.get("compress")
if c:
    cc = c.get("cpu_baseline", {})
    rows.append(("[2] batch compress, same inputs, every frame compared with libzstd's",
                 "**%.1f GB/s**, %.1f ms per step; end to end %s; %s" % (c["value"], c["ms_per_step"], pct(c["roofline"]["end_to_end"]["frac"]), kern(c["roofline"]["per_kernel"])),
                 "%.1f GB/s (%s threads)" % (cc.get("value", 0), cc.get("cores", "?"))))
d = line.get("dict")
if d and "error" not in d:
    dc, dd = d.get("cpu_baseline", {}), d["decompress"]
    rows.append(("[3] %d x 4 KiB JSON documents, shared %d-byte trained dictionary" % (d["config"]["docs_per_gpu"], d["config"]["dict_bytes"]),
                 "compress **%.1f GB/s** (%.1f ms; %s), decompress **%.1f GB/s** (%.1f ms); every frame libzstd's, every document back"
                 % (d["value"], d["ms_per_step"], kern(d["roofline"]["per_kernel"]), dd["value"], dd["ms_per_step"]),
                 "%.1f / %.1f GB/s" % (dc.get("value", 0), dd.get("cpu_baseline", {}).get("value", 0))))
t = line.get("roundtrip")
if t and "error" not in t:
    tc = t.get("cpu_baseline", {})
    rows.append(("[4] round trip, %d x 128 KiB per GPU generated in HBM" % t["config"]["frames_per_gpu"],
                 "**%.1f GB/s** per round trip (compress %.1f + decompress %.1f GB/s); N > 1 all-gatherv leg unmeasured (no multi-GPU hardware)"
                 % (t["value"], t["compress"]["value"], t["decompress"]["value"]),
                 "%.1f GB/s (compress %.1f, decompress %.1f)" % (tc.get("value", 0), tc.get("compress", 0), tc.get("decompress", 0))))
b = line.get("blocks")
if b and "error" not in b:
    rows.append(("[-] frames of several blocks: %d x 1 MiB (not a BASELINE config)" % b["config"]["frames_per_gpu"],
                 "decompress **%.1f GB/s** (%.1f ms per step: the phase-split kernels' several-block mode), compress %.1f GB/s (the generic kernel; large batches of "
                 "such sources take the flat search, DESIGN.md 4.2); every frame libzstd's, every byte back" % (b["value"], b["ms_per_step"], b["compress"]["value"]),
                 "24-30 / 6 GB/s (rows 1-2)"))
print("| workload (BASELINE.json config) | this backend, one MI355X, buffers resident in HBM | reference libzstd 1.5.7 on the same box's host cores |")
print("|---|---|---|")
for a, b, c_ in rows:
    print("| %s | %s | %s |" % (a, b, c_))
