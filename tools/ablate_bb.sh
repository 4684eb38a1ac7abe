#!/bin/bash
mkdir -p gpurun_out
for V in "base:" "noxload:-DBB_DO_XLOAD=0" "nostore:-DBB_DO_STORE=0" "nores:-DBB_DO_RES=0" "nomem:-DBB_DO_XLOAD=0 -DBB_DO_STORE=0 -DBB_DO_RES=0"; do
  name=${V%%:*}; flags=${V#*:}
  ESA_HIPCC_FLAGS="$flags" python esa-pose-estimation_amd/build.py --force > gpurun_out/abb_build_$name.log 2>&1 || { echo "build $name failed"; tail -5 gpurun_out/abb_build_$name.log; continue; }
  python tools/profile_ops.py --reps 3 > gpurun_out/abb_$name.txt 2>&1
  echo "== $name: $(grep -E 'stage4.0.branches.0.3.conv1' gpurun_out/abb_$name.txt)"
done
python esa-pose-estimation_amd/build.py --force > /dev/null 2>&1
