#!/bin/bash
# usage (on the GPU box, from the repo root): tools/refresh_profiles.sh <tag>
# Produces under gpurun_out/<tag>/: bench.json (plain run), bench_under_rocprof.json + the rocprofv3 --kernel-trace
# --stats CSVs of the same command (plus the seg_hrnet3 and the W48 bf16 workloads), and the separate --pmc passes; copy what is to be judged into profiles/.
set -u
R=$PWD
TAG=${1:-prof}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_hrnet3 -- python3 $R/bench.py --variant seg_hrnet3 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_hrnet3_under_rocprof.json 2> $OUT/rocprof_hrnet3.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_w48bf16 -- python3 $R/bench.py --workload w48-bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_w48bf16_under_rocprof.json 2> $OUT/rocprof_w48bf16.err || exit 1
cd $R
bash tools/pmc_run.sh $TAG/pmc > $OUT/pmc.log 2>&1 || exit 1
python3 tools/make_traffic_json.py $OUT/pmc $OUT/traffic.json > $OUT/traffic.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
ls $OUT $OUT/stats/* | head -30
