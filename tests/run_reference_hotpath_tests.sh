#!/bin/sh
# The reference's own hot-path tests (staged by tests/stage_reference_tests.sh) against the CPython extension and the Python mirror.
# Expected: 42 of 54 pass; the other 12 need zstd.train_dictionary, ZstdCompressionParameters, the magicless format or libzstd's
# multi-threaded frame layout (threads=2 changes the bytes) -- all outside the hot-path scope.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cd .reftmp || { echo "run tests/stage_reference_tests.sh first"; exit 1; }
for b in cext python; do
  echo "== backend $b"
  ( SHIM_BACKEND=$b PYTHONPATH="$PWD" timeout 300 python -m pytest -q -p no:cacheprovider tests 2>&1 | tail -30 )
done > ../gpurun_out/reference_hotpath_tests.log 2>&1
cat ../gpurun_out/reference_hotpath_tests.log
