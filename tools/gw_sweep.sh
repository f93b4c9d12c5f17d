#!/bin/bash
# tuning aid: queries of a gather pass below which every query gets a wave (GI_GATHER_WAVE_BELOW), on the frame and a 1/8 share
cd ${GRAFT_REPO_ROOT:-.}
for t in "$@"; do
  export GI_GATHER_WAVE_BELOW=$t
  echo "gather_wave_below $t"; timeout -k 5 200 python tools/stripe_probe.py 1 8 2>&1 | grep -v amdgpu
done
