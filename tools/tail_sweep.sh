#!/bin/bash
# tuning aid: one rank's 1/8 share of the benchmark frame (tools/stripe_probe.py 8) for several finisher settings
run() { echo "== $*"; env "$@" python3 tools/stripe_probe.py 8 2>/dev/null | tail -1; }
run GI_COOP_FACTOR=8
run GI_COOP_FACTOR=2
run GI_COOP_FACTOR=4
run GI_COOP_FACTOR=16
run GI_COOP_FACTOR=32
run GI_FINISH_THRESHOLD=32768
run GI_FINISH_THRESHOLD=65536
run GI_FINISH_THRESHOLD=262144
run GI_FINISH_THRESHOLD=32768 GI_COOP_FACTOR=16
