#!/bin/bash
# usage (GPU box, repo root): tools/x6_ablate.sh "<mask> <mask> ..."   — phase trace of conv_x6 with parts compiled out
for m in $1; do
  ESA_HIPCC_FLAGS="-DX6_TRACE=1 -DX6_ABL=$m" python3 esa-pose-estimation_amd/build.py --force > /dev/null 2>&1 || { echo build failed $m; exit 1; }
  echo "== X6_ABL=$m"
  python3 tools/trace_x6.py 128 64 64 64 64 2>/dev/null | grep -A3 "workgroup 0"
done
