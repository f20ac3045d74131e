#!/bin/bash
# Same-call comparison of builds of the wide-row Hessian (tools/wide_hess_time.py): tools/wide_ablation.sh "" q3 c8 ...
# (tags of MI355OPT_BUILD_TAG builds; "" = the product build; NOQUAD = the product build with MI355OPT_WIDE_QUAD=0)
mkdir -p gpurun_out/r05b
for rep in 1 2; do
for t in "$@"; do
  if [ "$t" = NOQUAD ]; then
    MI355OPT_WIDE_QUAD=0 timeout 300 python tools/wide_hess_time.py 2>&1 | tail -4 | sed 's/^/lane-per-row /'
  elif [ "$t" = QUAD ]; then
    MI355OPT_WIDE_QUAD=1 timeout 300 python tools/wide_hess_time.py 2>&1 | tail -4 | sed 's/^/quad layout  /'
  else
    lib=optimization_amd/libmi355opt${t:+_$t}.so
    MI355OPT_LIB=$PWD/$lib timeout 300 python tools/wide_hess_time.py 2>&1 | tail -4
  fi
done; done
