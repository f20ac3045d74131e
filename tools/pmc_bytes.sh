#!/bin/bash
# Exact L2<->fabric bytes per launch of the bench kernels: read requests resolved by size (128/64/32 B, calibrated in
# tools/fetch_calib.sh: sum equals the known bytes of every stream pattern) and WRITE_SIZE.  On the GPU box:
#   bash tools/pmc_bytes.sh [outdir] [extra bench args]      (PMC_CMD="python tools/x.py ..." profiles that instead)
set -u
REPO=$(pwd)
OUT=$REPO/${1:-gpurun_out/pmc_bytes}
shift || true
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-roofline --no-legs --steps 60 --warmup 10 $*"
[ -n "${PMC_CMD:-}" ] && BENCH="$PMC_CMD"
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --kernel-trace --output-format csv -d "$OUT/rd_a" -- $BENCH > "$OUT/rd_a.log" 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$OUT/rd_b" -- $BENCH > "$OUT/rd_b.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/wr" -- $BENCH > "$OUT/wr.log" 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/l2" -- $BENCH > "$OUT/l2.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(agg.items()):
    c = {n: sum(v) / len(v) for n, v in d.items()}
    if "TCC_EA0_RDREQ_sum" not in c or c["TCC_EA0_RDREQ_sum"] + c.get("WRITE_SIZE", 0) < 1000:
        continue
    n128, n64, n32 = c.get("TCC_EA0_RDREQ_128B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_32B_sum", 0)
    rd = 128 * n128 + 64 * n64 + 32 * n32
    wr = c.get("WRITE_SIZE", 0) * 1024
    hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
    res[k] = dict(read_bytes=rd, write_bytes=wr, total=rd + wr, l2_hit=hit, launches=len(d["TCC_EA0_RDREQ_sum"]),
                  unclassified=c["TCC_EA0_RDREQ_sum"] - n128 - n64 - n32)
    print("%-28s read %8.1f MB  write %7.1f MB  total %8.1f MB  L2 hit %.2f  (n=%d)" % (k, rd / 1e6, wr / 1e6, (rd + wr) / 1e6, hit, res[k]["launches"]))
json.dump(res, open(out + "/bytes.json", "w"), indent=1)
PY
