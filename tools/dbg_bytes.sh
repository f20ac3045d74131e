# fabric bytes of the Hessian kernel for experiment builds (tools/pmc_bytes.sh per library)
for t in "$@"; do
  echo "== lib $t"
  MI355OPT_LIB=$PWD/optimization_amd/libmi355opt_$t.so bash tools/pmc_bytes.sh gpurun_out/pmc_lib_$t 2>&1 | grep "hess\|pupdate\|cg_update"
done
