#!/bin/bash
# tuning aid: one rank's 1/8 share of the benchmark frame (tools/stripe_probe.py 8) and the whole frame (1) for several finisher settings
run() { echo "== $*"; env "$@" python3 tools/stripe_probe.py 8 1 2>/dev/null | tail -2 | cut -c1-200; }
run GI_COOP_FACTOR=4
run GI_COOP_FACTOR=8
run GI_COOP_FACTOR=16
run GI_COOP_FACTOR=32
run GI_FINISH_THRESHOLD=65536 GI_COOP_FACTOR=8
run GI_FINISH_THRESHOLD=262144 GI_COOP_FACTOR=16
