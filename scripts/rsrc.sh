#!/bin/bash
# usage: scripts/rsrc.sh <file.hip> <mangled-name-substring>   -> register / LDS usage of matching kernels
cd /root/repo/torch_motion_correction_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -c "$1" -o /tmp/rsrc_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A10 "Function Name: .*$2" | grep "Name\|VGPRs\|Spill\|Scratch\|Occ\|LDS"
rm -f /tmp/rsrc_$$.o
