export PYTHONPATH=.
for t in 32768 131072 524288; do
  for plan in "0:1,0:1,0:1,0:1,0:2,0:2,0:4,0:8,0:65" "0:1,0:1,0:1,0:1,0:1,0:1,0:2,0:2,0:2,0:4,0:4,0:8,0:65" "0:2,0:2,0:4,0:8,0:65"; do
    echo "T=$t plan=$plan $(GI_FINISH_THRESHOLD=$t GI_FINISH_PLAN=$plan timeout -k 10 100 python tools/stripe_probe.py 8 2>&1 | tail -1 | cut -c1-22,40-200)"
  done
done
