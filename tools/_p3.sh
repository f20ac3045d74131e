R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 800 rocprofv3 --kernel-trace -d $R/gpurun_out/conv_db -o t -- python $R/tools/cfg5_converge.py max_iters=8000 > $R/gpurun_out/conv_db.log 2>&1
ls -la $R/gpurun_out/conv_db
