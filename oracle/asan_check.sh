#!/bin/sh
# TEST INFRASTRUCTURE: builds the CPU oracle with AddressSanitizer + UndefinedBehaviorSanitizer (gcc) into a scratch directory and
# runs the oracle-only CPU tests against it (sanitizers are CPU-only on this pool).  sh oracle/asan_check.sh   (from the repo root)
set -e
D=$(mktemp -d)
F="-O1 -g -std=c99 -ffp-contract=off -mfma -fopenmp -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc $F -c -o $D/a.o oracle/sfm_oracle.c
gcc $F -c -o $D/b.o oracle/sift_oracle.c
gcc $F -fopenmp-simd -c -o $D/c.o oracle/sfm_oracle_fast.c
gcc -shared -fopenmp -fsanitize=address,undefined -o $D/libsfm_oracle.so $D/a.o $D/b.o $D/c.o -lm
cp oracle/libsfm_oracle.so $D/keep.so
trap 'cp $D/keep.so oracle/libsfm_oracle.so; rm -rf $D' EXIT
cp $D/libsfm_oracle.so oracle/libsfm_oracle.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
python -m pytest tests/test_oracle_golden.py tests/test_oracle_math.py tests/test_oracle_fast.py tests/test_oracle_sift.py tests/test_host_geom.py -x -q
