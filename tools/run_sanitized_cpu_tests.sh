#!/bin/bash
# the CPU-side native code (oracle restatement + C++ host classes) under AddressSanitizer + UBSan.
# Builds the two sanitizer variants and runs the whole `-m "not gpu"` suite over them; the summary goes to profiles/.
# GPU sanitizers are not available on this pool, so the HIP kernels are not covered by this run.
set -e
cd "$(dirname "$0")/.."
make -C oracle sanitize > /dev/null
make -C hybkinectfu_amd/host sanitize > /dev/null
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
out=${1:-profiles/r03_sanitizer_cpu.txt}
{
  echo "# $(date -u +%F) ASan+UBSan run of the CPU test suite: oracle/libkforacle_asan.so + hybkinectfu_amd/libhybkf_host_asan.so"
  echo "# g++ -fsanitize=address,undefined -fno-sanitize-recover=undefined; LD_PRELOAD=libasan.so libubsan.so; detect_leaks=0 (CPython itself leaks at exit)"
  LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    KF_ORACLE_SO=$PWD/oracle/libkforacle_asan.so KF_HOST_LIB=$PWD/hybkinectfu_amd/libhybkf_host_asan.so OMP_NUM_THREADS=4 \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
} | tee "$out"
