#!/bin/sh
# The reference's own hot-path tests (staged by tests/stage_reference_tests.sh) against the CPython extension, on the GPU box.
# What cannot pass by construction: tests that expect libzstd's multi-threaded frame layout (threads=2 changes the bytes) or
# strategies / features outside the hot-path scope.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cd .reftmp || { echo "run tests/stage_reference_tests.sh first"; exit 1; }
( PYTHONPATH="$PWD" timeout 600 python -m pytest -q -p no:cacheprovider tests 2>&1 | tail -60 ) > ../gpurun_out/reference_hotpath_tests.log 2>&1
tail -40 ../gpurun_out/reference_hotpath_tests.log
