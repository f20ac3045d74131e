#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on kernels of known byte counts (tools/microbench/fetch_calib.hip), one counter
# group per rocprofv3 pass.  On the GPU box:  bash tools/fetch_calib.sh   -> gpurun_out/fetch_calib/summary.json
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/fetch_calib
mkdir -p "$OUT"
BIN=$REPO/tools/microbench/fetch_calib
[ -x "$BIN" ] || hipcc --offload-arch=gfx950 -O3 -o "$BIN" "$REPO/tools/microbench/fetch_calib.hip" || exit 1
"$BIN" > "$OUT/bytes.json" || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_BUBBLE[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE" | sort -u > "$OUT/counters_available.txt"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/g$i" -- "$BIN" > "$OUT/g$i.log" 2>&1
done
cd "$REPO"
python tools/fetch_calib.py "$OUT"
