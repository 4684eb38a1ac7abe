#!/bin/bash
# usage: tools/isa.sh FILE.hip [extra hipcc flags]  -> /tmp/<name>.s (gfx950 device ISA) + register summary
f=$1; shift
n=$(basename $f .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o /tmp/$n.s /root/repo/esa-pose-estimation_amd/csrc/$n.hip 2>&1 | grep -E "error" -A3
grep -E "^    \.name:|^    \.vgpr_count|^    \.private_segment_fixed_size" /tmp/$n.s | paste - - - | awk '{print $2, "scratch", $4, "vgpr", $6}'
