#!/bin/bash
# Everything profiles/<tag>_* is made from, in one go on the GPU box (about 6 minutes):
#   bash tools/profile_run.sh r01   then, back in the build container,   python tools/profile_summary.py r01
# Counter passes run on their own (--pmc with --kernel-trace only), one counter group per pass (tools/pmc_bytes.sh).
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
python tools/bench_extra.py cfg3 cfg5 > "$OUT/${TAG}_extra.json" 2> "$OUT/${TAG}_extra.err"
python tools/bench_lsqr.py > "$OUT/${TAG}_lsqr.json" 2> "$OUT/${TAG}_lsqr.err"
python tools/bench_tnt.py > "$OUT/${TAG}_tnt.json" 2> "$OUT/${TAG}_tnt.err"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-roofline --no-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- $BENCH --steps 500 --warmup 50 \
  > "$OUT/${TAG}_trace.log" 2>&1
# r06: the driver's own command WITH the cfg3 / cfg5 legs (no CPU legs: they are not kernels) under the kernel trace: the
# per-kernel averages its legs' dominant kernels must agree with (k_bsr3_spmv*, k_so3_model, k_spmm_colmajor_win,
# k_gram_pair_sym, k_lobpcg_update*)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_legs" -- \
  python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/${TAG}_trace_legs.json" 2> "$OUT/${TAG}_trace_legs.log"
cp $(ls -t "$OUT/${TAG}_trace_legs"/*/*kernel_stats.csv | head -1) "$OUT/${TAG}_bench_legs_kernel_stats.csv" 2>/dev/null
cd "$REPO"
# exact L2<->fabric bytes per launch (read requests by size + WRITE_SIZE, separate passes) and their calibration
bash tools/pmc_bytes.sh "gpurun_out/${TAG}_pmc" > "$OUT/${TAG}_pmc.log" 2>&1
bash tools/fetch_calib.sh > "$OUT/${TAG}_fetch_calib.log" 2>&1
cp "$OUT/fetch_calib/summary.json" "$OUT/${TAG}_fetch_calibration.json" 2>/dev/null
cd "$REPO"
tail -1 "$OUT/${TAG}_bench.json" | cut -c1-300
