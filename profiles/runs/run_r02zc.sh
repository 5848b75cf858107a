# r02zc: host packing threads for the compress direction (8.6 GB of pageable input per call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
for h in 8 16 32 64; do
  ZHIP_PACK_THREADS=$h timeout 600 python tests/host_api_rate.py 65536 > $O/r02zc_pack$h.log 2>&1
  python - pack$h $O/r02zc_pack$h.log <<'PY'
import sys, json, re
t = open(sys.argv[2]).read(); m = re.search(r"\{.*\}", t.splitlines()[-1])
d = json.loads(m.group(0)); print(sys.argv[1], "decompress", d["decompress_GBps"], "compress", d["compress_GBps"])
PY
done
