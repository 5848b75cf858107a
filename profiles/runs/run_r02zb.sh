# r02zb: host-side compress chunking (items per chunk of the three-stream pipeline; a smaller first chunk)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
for h in 0 8192 16384 4096; do
  ZHIP_HCHUNK_E0=$h timeout 600 python tests/host_api_rate.py 65536 > $O/r02zb_first$h.log 2>&1
  python - first$h $O/r02zb_first$h.log <<'PY'
import sys, json, re
t = open(sys.argv[2]).read(); m = re.search(r"\{.*\}", t.splitlines()[-1])
d = json.loads(m.group(0)); print(sys.argv[1], "decompress", d["decompress_GBps"], "compress", d["compress_GBps"])
PY
done
