"""Static instruction count per source line of one kernel (from `hipcc --save-temps -gline-tables-only` assembly). Analysis aid.
usage: python tests/tools/isa_lines.py file.s kernel_name [first_line last_line]"""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
files = {}
inside = False
cur = (0, 0)
cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for line in open(path):
    if line.startswith(kern + ":"):
        inside = True; continue
    if not inside: continue
    if line.startswith(".Lfunc_end"): break
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', line)
    if m: files[int(m.group(1))] = m.group(3); continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", line)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r"\s+([sv]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|scratch_[a-z0-9_]+|flat_[a-z0-9_]+)\b", line)
    if m:
        op = m.group(1)
        k = "S" if op.startswith("s_") else "V" if op.startswith("v_") else "L" if op.startswith("ds_") else "M"
        cnt[cur] += 1; kinds[cur][k] += 1
tot = collections.Counter()
for (f, l), c in sorted(cnt.items()):
    if lo <= l <= hi:
        print("%5d %-28s %5d  %s" % (l, files.get(f, f), c, dict(kinds[(f, l)])))
    for k, v in kinds[(f, l)].items(): tot[k] += v
print("total", sum(cnt.values()), dict(tot))
