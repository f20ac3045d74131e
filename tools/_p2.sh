R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/lob_db -o t -- python $R/tools/prof_lobpcg.py 22 > $R/gpurun_out/lob_db.log 2>&1
ls $R/gpurun_out/lob_db
