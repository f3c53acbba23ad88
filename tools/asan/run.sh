#!/bin/bash
# The sanitizer pass of the host code, with its log kept: profiles/<tag>/asan_ubsan_host.log  (CPU box only)
cd "$(dirname "$0")/../.." || exit 1
TAG=${TAG:-r06}
mkdir -p profiles/$TAG
LOG=profiles/$TAG/asan_ubsan_host.log
make -C tools/asan > /dev/null || exit 1
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0:detect_odr_violation=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
{
  echo "== g++/gcc -fsanitize=address,undefined builds of: sgm_plan_host.hpp (the library's host planners), oracle/sigma_oracle.c, tests/mock_rccl/mock_rccl.cpp (host memory)"
  echo "== $(gcc --version | head -1); $(date -u +%Y-%m-%dT%H:%MZ); commit $(git rev-parse --short HEAD)"
  for r in "1 2" "3 4" "8 6" "16 2"; do ASAN_OPTIONS=${ASAN_OPTIONS/detect_leaks=0/detect_leaks=1} ./tools/asan/mock_rccl_asan $r 2>&1; done
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) SGM_ASAN_HOOK=1 PYTHONPATH=tools/asan/pyhook:$PWD \
    python -m pytest -q -p no:cacheprovider tests/test_dist_cpu.py tests/test_oracle_golden.py tests/test_cabi_cpu.py \
      -k "not test_cabi_cpu or halo_plan or slice_schedule" 2>&1 | grep -v "^\[asan hook\]" 
  echo "== sanitizer reports in this log: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' $LOG 2>/dev/null || echo 0)"
} > $LOG 2>&1
REPORTS=$(grep -c 'ERROR: AddressSanitizer\|runtime error:' $LOG)
sed -i "s/^== sanitizer reports in this log: .*/== sanitizer reports in this log: $REPORTS/" $LOG
tail -5 $LOG
