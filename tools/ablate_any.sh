#!/bin/bash
# usage: ablate_any.sh GREP_PATTERN "name:flags" ...   (build + per-op profile of each variant on the GPU box)
pat=$1; shift
mkdir -p gpurun_out
for V in "$@"; do
  name=${V%%:*}; flags=${V#*:}
  ESA_HIPCC_FLAGS="$flags" python esa-pose-estimation_amd/build.py --force > gpurun_out/abl_build_$name.log 2>&1 || { echo "build $name failed"; tail -5 gpurun_out/abl_build_$name.log; continue; }
  python tools/profile_ops.py --reps 5 > gpurun_out/abl_$name.txt 2>&1
  echo "== $name: $(tail -1 gpurun_out/abl_$name.txt)"
  grep -E "$pat" gpurun_out/abl_$name.txt
done
python esa-pose-estimation_amd/build.py --force > /dev/null 2>&1
