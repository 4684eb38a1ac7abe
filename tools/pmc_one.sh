#!/bin/bash
# usage: tools/pmc_one.sh <tag> "<counters>"  — one rocprofv3 --pmc pass over tools/one_forward.py
set -u
R=$PWD; TAG=$1; C=$2
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- python3 $R/tools/one_forward.py --reps 2 > $OUT/log.txt 2>&1
echo rc=$?
