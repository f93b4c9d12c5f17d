#!/bin/bash
# tuning aid: the finisher's hand-over point (GI_FINISH_THRESHOLD) and its one-path-per-wave cut-off (GI_WAVE_FACTOR) on the frame and on a 1/8 share
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "${@:-default}"; do
  if [ "$cfg" != default ]; then export GI_WAVE_FACTOR=${cfg%%:*} GI_FINISH_THRESHOLD=${cfg##*:}; fi
  echo "wave_factor:threshold $cfg"; timeout -k 5 200 python tools/stripe_probe.py 1 8 2>&1 | grep -v amdgpu
done
