mkdir -p gpurun_out/r04
(timeout 1200 python -m pytest tests/test_gpu_lobpcg.py -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r04/t5_lob.log 2>&1
(timeout 600 python tools/cfg5_converge.py max_iters=8000 --json-out gpurun_out/r04/cfg5_converge2.json 2>&1 | tail -3) > gpurun_out/r04/t5_conv.log 2>&1
(timeout 600 python tools/bench_extra.py cfg5 2>&1 | tail -3) > gpurun_out/r04/t5_extra.log 2>&1
bash tools/prof_lobpcg.sh > gpurun_out/r04/lob_prof2.log 2>&1
rm -rf gpurun_out/lob_trace
(MI355OPT_BENCH_FORCE_COMM=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --no-cpu-baseline --no-roofline --ab-steps 1000 2> gpurun_out/r04/shard2.err > gpurun_out/r04/shard2.json)
cat gpurun_out/r04/t5_lob.log gpurun_out/r04/t5_conv.log gpurun_out/r04/t5_extra.log; tail -22 gpurun_out/r04/lob_prof2.log
