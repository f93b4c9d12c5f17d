export PYTHONPATH=.
for t in 65536 131072 262144 524288 1048576; do
  echo "T=$t $(GI_FINISH_THRESHOLD=$t timeout -k 10 100 python tools/stripe_probe.py 1 8 2>&1 | tail -2 | cut -c1-14,60-200 | tr '\n' ' ')"
done
