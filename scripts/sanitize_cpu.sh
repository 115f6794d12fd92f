#!/bin/bash
# CPU-side AddressSanitizer + UndefinedBehaviorSanitizer job (round 4): the HOST code of the product library
# (smm_legacy.cpp: the legacy C ABI; the host parts of smm_api.hip: handles, pool, hashing, download, error paths) and
# the oracle's C restatement, built with -fsanitize=address,undefined and driven by the CPU test files.  GPU sanitizers
# are not available on this pool (no xnack+ code objects): the device code in the library is compiled as usual and
# never runs here.      bash scripts/sanitize_cpu.sh [outdir]      (exit code 0 = clean)
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/smm_san}
mkdir -p "$OUT"
CLANG_DIR=/opt/rocm/lib/llvm
RT=$(ls $CLANG_DIR/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libsan -g -O1"
cd "$ROOT/sparse_matrix_mult_amd/csrc"
/opt/rocm/bin/hipcc $SAN -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-option-ignored -shared \
    -o "$OUT/libsmm_hip_san.so" smm_api.hip -x hip smm_legacy.cpp
$CLANG_DIR/bin/clang $SAN -std=c11 -fPIC -ffp-contract=off -shared -o "$OUT/liboracle_san.so" "$ROOT/oracle/smm_oracle.c"
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD=$RT SMM_LIB_PATH="$OUT/libsmm_hip_san.so" SMM_ORACLE_LIB="$OUT/liboracle_san.so" \
    python -m pytest tests/test_abi.py tests/test_oracle_golden.py tests/test_host_logic.py -x -q -p no:cacheprovider
echo "sanitizers: clean ($(basename $RT), libsmm_hip_san.so + liboracle_san.so)"
