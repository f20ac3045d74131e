#!/bin/bash
# r03 evidence in one GPU call (about 5 minutes): the bench line, rocprofv3 kernel stats of the same command, PMC
# traffic of the cfg2 kernels and of the cfg3 HVP (separate --pmc passes: tools/pmc_bytes.sh), cfg3 / cfg5 side
# measurements, TNT host-sync counts, the one-rank overhead of the sharded path.  Everything lands in gpurun_out/r03/.
set -u
REPO=$(pwd)
O=$REPO/gpurun_out/r03
mkdir -p $O
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- \
    python $REPO/bench.py --no-cpu-baseline --no-roofline --no-legs --steps 500 --warmup 50 > $O/trace.log 2>&1 )
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/r03_kernel_stats.csv \;
bash tools/pmc_bytes.sh gpurun_out/r03/pmc > $O/r03_pmc_cfg2.log 2>&1
cp $O/pmc/bytes.json $O/r03_pmc_traffic_cfg2.json 2>/dev/null
python tools/bench_extra.py cfg3 > $O/r03_cfg3.json 2> $O/cfg3.err
MI355OPT_BSR3_NT=0 python tools/bench_extra.py cfg3 > $O/r03_cfg3_without_nt.json 2>> $O/cfg3.err
python tools/bench_extra.py cfg5 > $O/r03_cfg5.json 2> $O/cfg5.err
python tools/bench_tnt.py 1e-2 > $O/r03_tnt.json 2> $O/tnt.err
PMC_CMD="python $REPO/tools/bench_extra.py cfg3" bash tools/pmc_bytes.sh gpurun_out/r03/pmc_cfg3 > $O/r03_pmc_cfg3.log 2>&1
cp $O/pmc_cfg3/bytes.json $O/r03_pmc_traffic_cfg3.json 2>/dev/null
tools/sharded_overhead.sh > $O/r03_sharded_overhead.jsonl 2>/dev/null
rm -rf $O/trace/*/*.db $O/pmc/*/*/*.db 2>/dev/null
du -sh $O; ls $O
