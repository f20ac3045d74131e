mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_gpu_templates.py -x -q -m gpu -s -k "lsqr or tnls" 2>&1 | grep -E "^lsqr|^tnls|passed|failed|Error|assert" | tail -40) > gpurun_out/r04/t4_lsqr.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_user_operator.py -x -q -m gpu -s 2>&1 | tail -15) > gpurun_out/r04/t4_user.log 2>&1
(MI355OPT_BENCH_FORCE_COMM=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --no-cpu-baseline --no-roofline --ab-steps 1000 2> gpurun_out/r04/shard1.err > gpurun_out/r04/shard1.json)
bash tools/prof_lobpcg.sh > gpurun_out/r04/lob_prof.log 2>&1
python tools/trace_gaps.py gpurun_out/lob_trace > gpurun_out/r04/lob_gaps.log 2>&1
rm -rf gpurun_out/lob_trace
cat gpurun_out/r04/t4_lsqr.log gpurun_out/r04/t4_user.log gpurun_out/r04/lob_gaps.log
