#!/bin/bash
# usage: tools/sanitize/run.sh [pytest args...]   (from the repo root; CPU only)
# Builds the ASan + UBSan host libraries (tools/sanitize/Makefile) and runs the CPU test-suite, the PRL fuzzer and the
# glTF-importer fuzzer against them.  Any sanitizer report aborts the process (halt_on_error), so a green run means none.
set -e
cd "$(dirname "$0")/../.."
make -s -C tools/sanitize -j3
export PINE_GPU_LIB=$PWD/build/asan/libpine_gpu.so PINE_PRL_LIB=$PWD/build/asan/libpine_prl.so PINE_ORACLE_LIB=$PWD/build/asan/liboracle.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export PINE_SANITIZER_RUN=1
if [ $# -eq 0 ]; then set -- tests/test_abi.py tests/test_prl.py tests/test_oracle_golden.py tests/test_embree_order.py -m "not gpu"; fi
python -m pytest -x -q -p no:cacheprovider "$@"
python tools/fuzz_prl.py ${PINE_FUZZ_MUTANTS:-300}
python tools/fuzz_gltf.py ${PINE_FUZZ_MUTANTS:-300}
