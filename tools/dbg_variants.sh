#!/bin/bash
# usage: dbg_variants.sh "<dbg_conv_case args>" "name:flags" ...  — rebuild with each flag set and run the op-level repro
args=$1; shift
for V in "$@"; do
  name=${V%%:*}; flags=${V#*:}
  ESA_HIPCC_FLAGS="$flags" python esa-pose-estimation_amd/build.py --force > gpurun_out/dbgv_build_$name.log 2>&1 || { echo "build $name failed"; continue; }
  echo "== $name ($flags)"
  python tools/dbg_conv_case.py $args 2>&1 | grep -E "rep|x % 16" | head -6
done
python esa-pose-estimation_amd/build.py --force > /dev/null 2>&1
