#!/bin/bash
# run one python script against several library variants: ab_run.sh <script.py> <tag> [<tag> ...]  ("base" = in-tree)
script=$1; shift
for tag in "$@"; do
  if [ "$tag" = base ]; then lib=torch_motion_correction_amd/libmcorr.so; else lib=variants/$tag/libmcorr.so; fi
  echo "== $tag"
  MCORR_LIB=$PWD/$lib python3 $script 2>&1 | grep -v amdgpu.ids
done
