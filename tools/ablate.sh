#!/bin/bash
# build + profile several ablation variants of the conv kernel on the GPU box
mkdir -p gpurun_out
for V in "base:" "nostore:-DESA_DO_STORE=0" "nomfma:-DESA_DO_MFMA=0" "noxload:-DESA_DO_XLOAD=0" "nomfma_nostore:-DESA_DO_MFMA=0 -DESA_DO_STORE=0"; do
  name=${V%%:*}; flags=${V#*:}
  ESA_HIPCC_FLAGS="$flags" python esa-pose-estimation_amd/build.py --force > gpurun_out/abl_build_$name.log 2>&1 || { echo "build $name failed"; tail -5 gpurun_out/abl_build_$name.log; continue; }
  python tools/profile_ops.py --reps 3 > gpurun_out/abl_$name.txt 2>&1
  echo "== $name: $(tail -1 gpurun_out/abl_$name.txt)"
  grep -E " layer1.1.conv1| layer1.1.conv2|stage4.0.branches.1.3.conv1|stage4.0.branches.2.3.conv1|stage4.0.branches.3.3.conv1| conv2 " gpurun_out/abl_$name.txt
done
python esa-pose-estimation_amd/build.py --force > /dev/null 2>&1
