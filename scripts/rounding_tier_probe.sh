# one gpurun call: does the BEST-measured rounding realisation of a weight set (profiles/r04u_roundings_probe.txt) lead to a cheaper tier than the default one?
for sv in "11 5" "21 3" "12 2" "24 2" "24 5" "13 4" "13 3"; do
  set -- $sv
  MLT_TUNING=1 MLT_ROUNDING=$2 python scripts/tier_probe.py $1 2>&1 | grep "^seed" | sed "s/^/forced rounding $2: /"
done
