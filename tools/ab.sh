#!/bin/bash
# A/B of two builds of the library on one box: tools/ab.sh <kinds...>  (B = indexr_amd/libixrgpu_b.so)
for k in "$@"; do
  echo "== A kind $k"; tools/kinds.sh $k | head -4
  echo "== B kind $k"; IXR_LIB=$PWD/indexr_amd/libixrgpu_b.so tools/kinds.sh $k | head -4
done
