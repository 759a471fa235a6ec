#!/bin/bash
# Parity of the device code WITHOUT a GPU: builds erasor_hip.hip + kernels against the tests' CPU stand-in of the HIP runtime
# (tests/cpp/simt_emu) and runs the checks on it.  Test infrastructure; says nothing about timing.
#   tools/simt_check.sh quick        sort + kernel checks + two look-ahead steps                      (~30 s)
#   tools/simt_check.sh suite [-k E] tests/test_gpu_parity.py on the stand-in (default: all but the full-size cases, ~1 h 45 min)
#   tools/simt_check.sh asan [-k E]  the same under AddressSanitizer (heap blocks = device buffers: out-of-bounds accesses show)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
MODE=${1:-quick}; shift || true
OUT=${TMPDIR:-/tmp}/erasor_simt
mkdir -p $OUT
cd $ROOT
make -C oracle -s >/dev/null 2>&1 || python -c "from oracle import orc; orc.build()"
CXX="g++ -O1 -std=c++20 -pthread -ffp-contract=off -Itests/cpp/simt_emu"
EXPR="not (full_size or config4 or whole_map or long_segments or map_grows)"
if [ "$1" = "-k" ]; then EXPR="$2"; fi
case $MODE in
quick)
  $CXX -o $OUT/esort_simt_check tests/cpp/esort_simt_check.cpp && $OUT/esort_simt_check | tail -1
  $CXX -o $OUT/kernels_simt_check tests/cpp/kernels_simt_check.cpp -Loracle -lerasor_oracle -Wl,-rpath,$ROOT/oracle && $OUT/kernels_simt_check | tail -1
  $CXX -x c++ -fPIC -shared -DERASOR_HIP_TEST_HOOKS -o $OUT/liberasor_hip_simt.so erasor_amd/csrc/erasor_hip.hip
  python tests/simt_full_step.py $OUT/liberasor_hip_simt.so $ROOT 2 | tail -3
  ;;
suite)
  $CXX -x c++ -fPIC -shared -DERASOR_HIP_TEST_HOOKS -o $OUT/liberasor_hip_simt.so erasor_amd/csrc/erasor_hip.hip
  ERASOR_TEST_SIMT_LIB=$OUT/liberasor_hip_simt.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_hooks.py tests/test_golden.py -m gpu -q -p no:cacheprovider -o timeout=1800 -k "$EXPR"
  ;;
asan)
  $CXX -g -fsanitize=address -fno-omit-frame-pointer -x c++ -fPIC -shared -DERASOR_HIP_TEST_HOOKS -o $OUT/liberasor_hip_asan.so erasor_amd/csrc/erasor_hip.hip
  LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1 \
    ERASOR_TEST_SIMT_LIB=$OUT/liberasor_hip_asan.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_hooks.py -m gpu -q -x -p no:cacheprovider -o timeout=3600 -k "$EXPR"
  ;;
*) echo "usage: $0 quick | suite [-k expr] | asan [-k expr]"; exit 2;;
esac
