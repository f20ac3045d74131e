mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests/test_gpu_comm.py -x -q -m gpu -k "cross_device" 2>&1 | tail -6) > gpurun_out/r04/t12_xdev.log 2>&1
(timeout 900 python -m pytest tests/test_gpu_lobpcg.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r04/t12_lob.log 2>&1
(timeout 600 python tools/cfg5_converge.py max_iters=8000 2>&1 | tail -1 | cut -c1-330) > gpurun_out/r04/t12_conv.log 2>&1
(timeout 600 python tools/bench_extra.py cfg5 2>&1 | tail -1 | grep -o '"lobpcg_ms_per_iteration": [0-9.]*') > gpurun_out/r04/t12_extra.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/t12_smoke.log 2>&1
cat gpurun_out/r04/t12_*.log
