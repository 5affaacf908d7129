# one gpurun call: which tier does every (weight set, rounding realisation) pair reach?  (is the "closest to the line" base the cheapest one?)
for s in 12 21 22 25 11 13 24; do
  for v in 0 1 2 3 4 5; do
    MLT_TUNING=1 MLT_ROUNDING=$v python scripts/tier_probe.py $s 2>&1 | grep "^seed" | cut -c1-170
  done
done
