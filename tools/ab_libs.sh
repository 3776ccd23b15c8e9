#!/bin/bash
# Alternating A/B of several builds of the library on one shard geometry (tools/shard_ab.py), three rounds:
#   [CF_DEBUG_FLAGS=bits] bash tools/ab_libs.sh "hq hkv S" path/a.so path/b.so ...
# A variant: CF_EXTRA_HIPCC_FLAGS=-DCF_...=1 python -m clusterfusion_amd.build --force, copy the .so aside, rebuild the default.
ARGS="$1"; shift
for r in 1 2 3; do
  for L in "$@"; do
    echo -n "$(basename $L) : "
    CF_LIB_PATH=$PWD/$L CF_DEBUG_FLAGS=${CF_DEBUG_FLAGS:-0} python tools/shard_ab.py $ARGS 2>&1 | tail -1
  done
done
