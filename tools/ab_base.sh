#!/bin/bash
# Builds the library of a git revision (default HEAD) into tools/ab/libcat_base.so for same-box A/B runs:
#   tools/ab_base.sh [rev];  on the GPU box: tools/ab_run.sh <command...> runs the command with the base library,
#   then with the working tree's library (the box's copy of the tree is scratch).
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
tmp=$(mktemp -d)
git archive $rev cat_amd/csrc include Makefile | tar -x -C $tmp
mkdir -p tools/ab
(cd $tmp && make -s -j8 cat_amd/lib/libcat_amd.so >/dev/null)
cp $tmp/cat_amd/lib/libcat_amd.so tools/ab/libcat_base.so
rm -rf $tmp
echo "tools/ab/libcat_base.so <- $rev"
