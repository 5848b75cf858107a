#!/bin/sh
# builds python-zstandard_amd/backend_hip.so : the CPython extension over the C ABI (binds csrc/libzstd_hip.so with dlopen at import)
set -e
cd "$(dirname "$0")"
PY=${PYTHON:-python3}
INC=$($PY -c "import sysconfig; print(sysconfig.get_paths()['include'])")
${CC:-gcc} -shared -fPIC -O2 -Wall -Wextra -Wno-missing-field-initializers -Wno-cast-function-type -I"$INC" -I../../include -o ../backend_hip.so backend_hip.c -ldl
