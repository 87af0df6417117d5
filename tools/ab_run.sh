#!/bin/bash
# Same-box A/B: runs "$@" with tools/ab/libcat_base.so in place of the library, then with the tree's own build.
cd "$(dirname "$0")/.."
cp cat_amd/lib/libcat_amd.so /tmp/libcat_new.so
cp tools/ab/libcat_base.so cat_amd/lib/libcat_amd.so
echo "== base"; "$@"
cp /tmp/libcat_new.so cat_amd/lib/libcat_amd.so
echo "== new"; "$@"
