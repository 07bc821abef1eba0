#!/bin/bash
# CPU-side hygiene: the oracle's C restatement and the hand-written N-API pointer handling under
# AddressSanitizer + UBSan. GPU ASan / XNACK are not available on the pool, so this covers what runs on the CPU:
#   1. oracle/nd4_oracle.c   -> libnd4_oracle_asan.so, loaded by the CPU tests that use the oracle (ND4_ORACLE_SO)
#   2. csrc/napi_shim.c      -> nd4hip_napi_asan.node, loaded by tests/js/node_checks.js cpu (ND4HIP_NAPI_ADDON)
# usage: bash tools/check_sanitize.sh      (from anywhere; exits non-zero on any sanitizer report or test failure)
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
ASAN_LIB=$(gcc -print-file-name=libasan.so)
make -s -C oracle asan
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== CPU tests with the sanitized oracle"
LD_PRELOAD="$ASAN_LIB" ND4_ORACLE_SO="$ROOT/oracle/libnd4_oracle_asan.so" python3 -m pytest tests/test_oracle_golden.py tests/test_chain.py -x -q -m "not gpu" -p no:cacheprovider
if command -v node >/dev/null && [ -f /usr/include/node/node_api.h ]; then
  echo "== N-API shim under ASan/UBSan"
  gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared -std=c11 -Wall \
      -I/usr/include/node -Iinclude -o nd4js_amd/js/nd4hip_napi_asan.node nd4js_amd/csrc/napi_shim.c -ldl
  LD_PRELOAD="$ASAN_LIB" ND4HIP_NAPI_ADDON="$ROOT/nd4js_amd/js/nd4hip_napi_asan.node" node tests/js/node_checks.js cpu tests/golden | tail -1
else
  echo "== node or its headers missing: N-API shim not checked"
fi
echo "sanitize checks ok"
