#!/bin/bash
# Collects the round's evidence on a GPU box into gpurun_out/round/ (copied into profiles/ afterwards):
#   bench JSON (full: cpu_baseline thread sweep + fp32 parity-mode side number), per-layer table, rocprofv3 kernel stats of the same
#   command, FETCH_SIZE / WRITE_SIZE passes for bench.py, kernel stats + traffic passes for C4 and C5 (299 / 512), other configs.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/round
mkdir -p $O
python bench.py --steps 20 --warmup 5 --per-layer > $O/bench.json 2> $O/per_layer.txt || exit 1
echo "bench done"; head -c 200 $O/bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-side --no-secondary --no-rccl-side > $O/bench_profiled.json 2> /dev/null || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-side --no-secondary --no-rccl-side --no-launch-timing > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-side --no-secondary --no-rccl-side --no-launch-timing > /dev/null 2>&1 || exit 1
echo "bench profiles done"
for cfg in c4 c5 c5x; do
  STEPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -o p -- python3 $R/tools/bench_configs.py $cfg > $O/$cfg.log 2>&1 || exit 1
  STEPS=2 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$cfg -o p --output-format csv -- python3 $R/tools/bench_configs.py $cfg > /dev/null 2>&1 || exit 1
  STEPS=2 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$cfg -o p --output-format csv -- python3 $R/tools/bench_configs.py $cfg > /dev/null 2>&1 || exit 1
  echo "$cfg profiles done"
done
cd $R
python tools/collect_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json > /dev/null
for cfg in c4 c5 c5x; do python tools/collect_traffic.py $O/pmc_fetch_$cfg $O/pmc_write_$cfg $O/traffic_$cfg.json all "tools/bench_configs.py $cfg (STEPS=2)" > /dev/null; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
STEPS=10 python tools/bench_configs.py > $O/other_configs.jsonl 2> $O/other_configs.err
grep -c config $O/other_configs.jsonl
ls $O
