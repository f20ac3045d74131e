#!/bin/bash
# Everything profiles/<tag>_* of a round is made from, on the GPU box (about 15 minutes):
#   bash tools/round_profiles.sh r04     then, in the build container:   python tools/profile_summary.py r04
# and copy gpurun_out/<tag>_keep/* into profiles/.  One artifact set per round (no per-experiment refreshes).
set -u
TAG=${1:-r06}
REPO=$(pwd)
K=$REPO/gpurun_out/${TAG}_keep
mkdir -p "$K"
# the full GPU suite first: parity is the first gate
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15) > "$K/${TAG}_gputest_full_suite.log"
# bench line, side benches, kernel trace, PMC bytes, fetch calibration
bash tools/profile_run.sh "$TAG" > "$K/${TAG}_profile_run.log" 2>&1
# cfg3: per-kernel figures + fabric bytes of the inner step, the pieces of the outer iteration and their fabric bytes
python tools/bench_extra.py cfg3 2>/dev/null | tail -1 > "$K/${TAG}_cfg3.json"
PMC_CMD="python $REPO/tools/bench_extra.py cfg3" bash tools/pmc_bytes.sh gpurun_out/${TAG}_pmc_cfg3 > /dev/null 2>&1
cp gpurun_out/${TAG}_pmc_cfg3/bytes.json "$K/${TAG}_pmc_traffic_cfg3.json"
python tools/time_so3_model.py 2>/dev/null | tail -1 > "$K/${TAG}_cfg3_outer_pieces.json"
PMC_CMD="python $REPO/tools/time_so3_model.py" bash tools/pmc_bytes.sh gpurun_out/${TAG}_pmc_cfg3_outer > /dev/null 2>&1
cp gpurun_out/${TAG}_pmc_cfg3_outer/bytes.json "$K/${TAG}_pmc_traffic_cfg3_outer.json"
# Stiefel(1e6, p) for p = 3 ... 8 (the wide-row one-pass Hessian), the Gram pair and the Ritz update A/B, the generalized LOBPCG
python tools/bench_extra.py wide 2>/dev/null > "$K/${TAG}_wide_rows.jsonl"
python tools/time_gram.py 2>/dev/null | tail -1 > "$K/${TAG}_gram_pair_ab.json"
python tools/time_update.py 2>/dev/null | tail -1 > "$K/${TAG}_update_ab.json"
python tools/lobpcg_gen.py 2>/dev/null | tail -1 > "$K/${TAG}_lobpcg_generalized.json"
# cfg5: side bench, the run to convergence, kernel stats of 22 iterations
python tools/bench_extra.py cfg5 2>/dev/null | tail -1 > "$K/${TAG}_cfg5.json"
python tools/cfg5_converge.py max_iters=8000 --json-out "$K/${TAG}_cfg5_converge.json" > /dev/null 2>&1
bash tools/prof_lobpcg.sh > "$K/${TAG}_lobpcg_prof.log" 2>&1
cp $(ls gpurun_out/lob_trace/*/*kernel_stats.csv | tail -1) "$K/${TAG}_lobpcg_kernel_stats.csv" 2>/dev/null
rm -rf gpurun_out/lob_trace
# beyond the Infinity Cache: fabric bytes of St(8e6, 3)
PMC_CMD="python $REPO/tools/big_leg.py 200 60" bash tools/pmc_bytes.sh gpurun_out/${TAG}_pmc_big > /dev/null 2>&1
cp gpurun_out/${TAG}_pmc_big/bytes.json "$K/${TAG}_pmc_traffic_beyond_cache.json"
# the sharded step at one rank: every exchange layer's leg (bench.py's own comm_ab_legs)
MI355OPT_BENCH_FORCE_COMM=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
  --master-port 29711 bench.py --gpus 1 --no-cpu-baseline --no-roofline --ab-steps 1000 2>/dev/null > "$K/${TAG}_sharded_one_rank.json"
# the bare N-rank command on this one GPU (functional rehearsal; timings mean nothing)
for n in 2 4 8; do
  python bench.py --gpus $n --steps 100 --warmup 5 --wakeup-steps 100 --ab-steps 100 2>/dev/null > "$K/${TAG}_rehearsal_n$n.json"
done
# a user-written HIP operator inside the fused STPCG
examples/bin/stpcg_user_stencil > "$K/${TAG}_user_operator.txt" 2>&1
# where k_cg_update's time goes (experiment build with in-kernel stamps, if it was built)
[ -f optimization_amd/libmi355opt_fstamp.so ] && MI355OPT_LIB=$PWD/optimization_amd/libmi355opt_fstamp.so \
  python tools/fold_stamps.py single > "$K/${TAG}_cg_update_stamps.txt" 2>&1
cp gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_tnt.json gpurun_out/${TAG}_lsqr.json "$K/" 2>/dev/null
cp gpurun_out/${TAG}_bench_legs_kernel_stats.csv gpurun_out/${TAG}_trace_legs.json "$K/" 2>/dev/null
# r06: the driver's own command, the parity curve, the long solves, what re-anchoring costs
(time python bench.py --steps 20 --warmup 5 --dry-run-layers 8) > "$K/${TAG}_bench_default.json" 2> "$K/${TAG}_bench_default.err"
python tools/parity_curve.py > "$K/${TAG}_parity_curve.json" 2> "$K/${TAG}_parity_curve_table.md"
python tools/reanchor_cost.py > "$K/${TAG}_reanchor_cost.json" 2>/dev/null
python -m pytest tests/test_gpu_long_solves.py -m gpu -q -s 2>&1 | grep -v "^$" | cut -c1-300 > "$K/${TAG}_long_solves.log"
cp gpurun_out/${TAG}_pmc/bytes.json "$K/${TAG}_pmc_traffic_cfg2.json" 2>/dev/null
ls -la "$K"; cat "$K/${TAG}_gputest_full_suite.log" | tail -3
