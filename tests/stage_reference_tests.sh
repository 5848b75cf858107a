#!/bin/sh
# Stages throw-away COPIES of the reference's own hot-path tests + a `zstandard` shim into .reftmp/ (git-ignored, never committed;
# it travels to the GPU box with gpurun, where /root/reference does not exist). Run here, then:
#   gpurun -- 'sh tests/run_reference_hotpath_tests.sh'
set -e
cd "$(dirname "$0")/.."
REF=${REF:-/root/reference}
rm -rf .reftmp; mkdir -p .reftmp/zstandard .reftmp/tests
cat > .reftmp/zstandard/__init__.py <<'PY'
# throw-away shim (never committed): `import zstandard` in a copy of the reference's tests resolves to this backend
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import zstandard_amd as _z
globals().update({k: getattr(_z, k) for k in dir(_z) if not k.startswith("_")})


def train_dictionary(dict_size, samples, **kw):
    """dictionary TRAINING is outside the hot path (SURVEY 8): the shim borrows it from the reference build (test infrastructure)"""
    import importlib.util
    root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
    spec = importlib.util.spec_from_file_location("_graft_reflib", os.path.join(root, "tests", "reflib.py"))     # `tests` here is the staged copy
    reflib = importlib.util.module_from_spec(spec); spec.loader.exec_module(reflib)
    return _z.ZstdCompressionDict(reflib.RefZstd().train_dictionary(dict_size, list(samples)))
PY
for f in __init__.py common.py test_buffer_util.py test_compressor_multi_compress_to_buffer.py test_decompressor_multi_decompress_to_buffer.py \
         test_compressor_compress.py test_decompressor_decompress.py test_decompressor_content_dict_chain.py; do
  cp "$REF/tests/$f" .reftmp/tests/
done
echo "staged $(ls .reftmp/tests | wc -l) files under .reftmp/"
