#!/bin/bash
# build + profile ablation variants of head_fused2 on the GPU box (timings only; outputs are wrong)
mkdir -p gpurun_out
for V in "$@"; do
  name=${V%%:*}; flags=${V#*:}
  ESA_HIPCC_FLAGS="$flags" python esa-pose-estimation_amd/build.py --force > gpurun_out/h2abl_build_$name.log 2>&1 || { echo "build $name failed"; tail -5 gpurun_out/h2abl_build_$name.log; continue; }
  python tools/profile_ops.py --reps 3 > gpurun_out/h2abl_$name.txt 2>&1
  echo "== $name: $(grep head_fused2 gpurun_out/h2abl_$name.txt)"
done
python esa-pose-estimation_amd/build.py --force > /dev/null 2>&1
