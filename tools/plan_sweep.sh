export PYTHONPATH=.
for plan in "0:1,0:1,0:1,0:1,0:2,0:2,0:4,0:8,0:65" "0:1,0:1,0:1,0:2,0:4,0:65" "0:1,0:1,0:2,0:65" "0:2,0:2,0:65" "0:1,0:1,0:1,0:1,0:65" "0:1,0:1,0:1,0:1,0:2,0:4,0:8,0:16,0:65"; do
  echo "plan=$plan $(GI_FINISH_PLAN=$plan timeout -k 10 100 python tools/stripe_probe.py 1 8 2>&1 | tail -2 | cut -c1-14,60-200 | tr '\n' ' ')"
done
