#!/bin/bash
# Hardware-counter passes over the hot kernels of `bench.py` (one rocprofv3 run per counter group).
# Usage (on the GPU box): bash tools/pmc_probe.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_probe}
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" \
  "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$REPO/$OUT/g$i" -- \
    python "$REPO/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-roofline > "$REPO/$OUT/g$i.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        for key in ("k_st_hess_fused", "k_st_spmm_gram", "k_st_finish", "k_cg_update", "k_cg_pupdate"):
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-44s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
