#!/bin/bash
# A/B/... of several builds of the library on one box: tools/abn.sh "<kinds>" lib1.so lib2.so ...   (kernel lines of k_str_fused etc.)
kinds="$1"; shift
for lib in "$@"; do
  for k in $kinds; do
    echo "== $lib kind $k"; IXR_LIB=$PWD/indexr_amd/$lib tools/kinds.sh $k | head -${ABN_LINES:-4}
  done
done
