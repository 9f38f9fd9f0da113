#!/bin/sh
# build/old_lib/libsfm_amd.so = the product library of commit $1 (default HEAD), for profiles/r04_ab_libs.sh (same-box A/B of two builds)
set -e
REV=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/sfm_oldtree
git -C $R worktree add -f /tmp/sfm_oldtree $REV > /dev/null 2>&1
make -C /tmp/sfm_oldtree -j8 cuda-sfm_amd/lib/libsfm_amd.so > /dev/null 2>&1
mkdir -p $R/build/old_lib
cp /tmp/sfm_oldtree/cuda-sfm_amd/lib/libsfm_amd.so $R/build/old_lib/
git -C $R worktree remove --force /tmp/sfm_oldtree
git -C $R rev-parse --short $REV > $R/build/old_lib/REV
echo "build/old_lib = $(cat $R/build/old_lib/REV)"
