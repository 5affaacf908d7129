# one gpurun call: could hi+lo weights in the stride-2 convs ALONE (the cheap units) admit the sets that need whole stages?  No: head-3 rms 2.0 - 3.6e-4
# against the 1.82e-4 the rule allows (profiles/r04t_units_force_probe.txt).
for s in 11 13 23 24; do
  for u in 0x55 0x54 0x50 0x44 0x14 0x15; do
    MLT_TUNING=1 MLT_W2_UNITS=$u MLT_CALIB_VERBOSE=1 python scripts/tier_probe.py $s 2>&1 | grep "units $u," | tail -1 | sed -E "s/.*hi\+lo weights in units (0x[0-9a-f]+), exact in units 0x0\): rms per class ([^|]*)\| per head ([^|]*)\| max ([0-9.e+-]+) = ([0-9.]+) x rms/seed $s units \1 heads \3 max \4 ratio \5/"
  done
done
