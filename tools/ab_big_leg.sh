#!/bin/bash
# Same-call A/B of experiment builds (MI355OPT_BUILD_TAG builds, loaded through MI355OPT_LIB) on the beyond-cache leg
# (tools/big_leg.py: St(8e6,3)).  Usage (GPU box): tools/ab_big_leg.sh "" tag1 tag2 ""
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=optimization_amd/libmi355opt${tag:+_$tag}.so
  echo "== ${tag:-base}"
  MI355OPT_LIB=$PWD/$lib python tools/big_leg.py 200 100 2>/dev/null | tail -1
done
