#!/bin/bash
# r06: what the two far gathers of the window-form panel product cost -- the product lib against the
# timing-only builds libmi355opt_farl2.so (far rows read next to the wave's own chunk: no traffic beyond the L2) and
# libmi355opt_far0.so (no far loads at all), cfg5's 48-column product, with the L2<->fabric bytes of each.
#   MI355OPT_BUILD_TAG=farl2 MI355OPT_EXTRA_CFLAGS=-DMI_SPMM_ABLATE_FAR=1 python -m optimization_amd.build   (and far0 / =2)
cd "$(dirname "$0")/.."
REPO=$PWD
for tag in "" farl2 far0; do
  lib=$REPO/optimization_amd/libmi355opt${tag:+_$tag}.so
  [ -f "$lib" ] || continue
  for rep in 1 2; do
    echo "${tag:-product}: $(MI355OPT_LIB=$lib python tools/spmm_win_check.py 126 48 2>/dev/null | tail -1)"
  done
  MI355OPT_LIB=$lib PMC_CMD="python $REPO/tools/spmm_win_check.py 126 48" bash tools/pmc_bytes.sh gpurun_out/r06_pmc_spmm_${tag:-product} 2>/dev/null | grep spmm
done
