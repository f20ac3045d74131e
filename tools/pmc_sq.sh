#!/bin/bash
# Shader-side counters of one kernel (which resource a pass is waiting for): instruction mix, busy / wait cycles per
# class, LDS bank conflicts, texture-path busy.  Separate --pmc passes (no trace domains besides the kernel trace).
#   PMC_CMD="python tools/wide_hess_time.py 6" bash tools/pmc_sq.sh outdir kernel_regex
set -u
REPO=$(pwd)
OUT=$REPO/${1:-gpurun_out/pmc_sq}
KRE=${2:-k_st_hess}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-python $REPO/tools/wide_hess_time.py 6}
i=0
if [ "${PMC_SETS:-sq}" = "tcp" ]; then  # texture-path detail: who stalls whom, round-trip latencies, address translation
SETS=("TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
      "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum" \
      "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
      "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_WRITE_REQ_sum" \
      "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
      "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum" \
      "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
      "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_SERIALIZATION_STALL_sum")
for set in "${SETS[@]}"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- $CMD > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $set" >> "$OUT/failed.txt"
done
else
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- $CMD > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $set" >> "$OUT/failed.txt"
done
fi
cd "$REPO"
python - "$OUT" "$KRE" <<'PY'
import csv, glob, sys, collections, re
out, kre = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if not re.search(kre, r["Kernel_Name"]):
            continue
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        agg[m.group(1) if m else r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in sorted(agg.items()):
        for n, v in sorted(d.items()):
            line = "%-26s %-34s %16.1f  (n=%d)" % (k, n, sum(v) / len(v), len(v))
            print(line); fo.write(line + "\n")
PY
