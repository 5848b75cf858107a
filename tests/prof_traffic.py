"""Run on the GPU box after tests/run_profiles.sh's rocprofv3 passes: reduces the csv outputs to what profiles/ keeps --
  <tag>_<dir>_kt_zhip_kernels_summary.csv   per-kernel launch durations (kernel trace)
  <tag>_<dir>_{fetch,write,sq}_counters.csv  per-kernel counter sums per launch
  traffic.json                               FETCH_SIZE + WRITE_SIZE bytes per 128 KiB frame and kernel (what bench.py's roofline.traffic scales)
Usage: python tests/prof_traffic.py <prof dir> <tag> <frames per launch>"""
import collections
import csv
import glob
import json
import os
import sys

prof, tag, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
out_dir = os.path.join(os.path.dirname(prof.rstrip("/")), "summary")
os.makedirs(out_dir, exist_ok=True)


def rows_of(sub, pattern):
    """(the flat match kernel's three- and four-probe instantiations are reported under the kernel's one name, as the library's own timers do)"""
    for path in glob.glob(os.path.join(prof, sub, "**", pattern), recursive=True):
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                if "Kernel_Name" in r:
                    r["Kernel_Name"] = r["Kernel_Name"].replace("match_flat3_kernel", "match_flat_kernel").replace("match_flat4_kernel", "match_flat_kernel")
                yield r


traffic = collections.defaultdict(float)
requests = collections.defaultdict(dict)
for direction in ("decode", "compress", "dict"):              # dict: kernel trace only (bench.py --config dict, 262 144 x 4 KiB)
    # kernel trace -> durations
    dur = collections.defaultdict(list)
    for r in rows_of(direction + "_kt", "*kernel_trace.csv"):
        if "zhip_" in r.get("Kernel_Name", ""):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    # (the compress trace's average mixed in the table pick's probe launches -- the first launches of the flat match kernel in a fresh
    # context, up to three, on candidate allocations of which only the fastest is kept. They are dropped here: what is left is the steady state the bench times)
    # Since round 6's last session the probes are SHORT launches (the first 8 KiB of every source, eight of them): they are told from the real launches by their size --
    # anything below half the longest launch (kernel trace) / half the largest count (counter passes, below) is a probe.
    k_ = "zhip_encode_match_flat_kernel"
    if direction in ("compress", "dict") and dur.get(k_):
        top = max(dur[k_]); dur[k_] = [d for d in dur[k_] if d >= 0.5 * top]
        if direction == "dict" and len(dur[k_]) > 8: dur[k_] = dur[k_][8:]      # (a dictionary batch's probes are whole launches -- its documents are shorter than a probe's 8 KiB: the first eight launches of the context)
    if dur:
        with open(os.path.join(out_dir, "%s_%s_kt_zhip_kernels_summary.csv" % (tag, direction)), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v)])
    for kind in ("fetch", "write", "sq", "sq2", "sq3", "tcc"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows_of("%s_%s" % (direction, kind), "*counter_collection.csv"):
            if "zhip_" in r.get("Kernel_Name", ""):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if not acc:
            continue
        with open(os.path.join(out_dir, "%s_%s_%s_counters.csv" % (tag, direction, kind)), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Kernel", "Counter", "Launches", "MeanPerLaunch"])
            for k in sorted(acc):
                for c in sorted(acc[k]):
                    v = acc[k][c]
                    if k == k_ and v and max(v) > 0: v = [x for x in v if x >= 0.5 * max(v)]          # (the pick's probe launches: see above)
                    w.writerow([k, c, len(v), round(sum(v) / len(v), 3)])
                    if c in ("FETCH_SIZE", "WRITE_SIZE"):               # KiB per launch (rocprofv3's unit) -> bytes per frame
                        traffic[k] += sum(v) / len(v) * 1024.0 / frames
                    if c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):      # the L2's memory-side requests per frame
                        requests[k][{"TCC_EA0_RDREQ_sum": "read", "TCC_EA0_WRREQ_sum": "write", "TCC_EA0_WRREQ_64B_sum": "write64"}[c]] = sum(v) / len(v) / frames
json.dump({"round": tag, "frames_per_launch": frames,
           "source": "profiles/%s_{decode,compress}_{fetch,write}_counters.csv: FETCH_SIZE + WRITE_SIZE (KiB per launch of %d frames, one rocprofv3 --pmc pass each, "
                     "tests/run_profiles.sh); FETCH_SIZE uncorrected (the guide's x2 applies to wide coalesced reads; these kernels' reads are narrow)" % (tag, frames),
           "bytes_per_frame": {k: round(v, 1) for k, v in sorted(traffic.items())},
           "requests_per_frame": {k: {"read": round(v.get("read", 0.0), 1), "write": round(v.get("write", 0.0), 1),
                                      "write64_share": round(v.get("write64", 0.0) / v["write"], 4) if v.get("write") else None}
                                  for k, v in sorted(requests.items()) if v.get("read", 0) + v.get("write", 0) >= 1.0}},
          open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
print(open(os.path.join(out_dir, "traffic.json")).read())
