# r02zm: decode chunk default 65 536 -- decode and round-trip bench lines, decode GPU tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decompress.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline > $O/r02zm_decode.json 2> $O/r02zm_decode.err; python -c "
import json; d=json.load(open('$O/r02zm_decode.json')); print('decode', d['value'], d['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in d['kernels'].items()})"
timeout 900 python bench.py --config roundtrip --no-cpu-baseline > $O/r02zm_roundtrip.json 2> $O/r02zm_roundtrip.err; python -c "
import json; d=json.load(open('$O/r02zm_roundtrip.json')); print('roundtrip', d['value'], d['ms_per_step'], d['compress']['value'], d['decompress']['value'])"
