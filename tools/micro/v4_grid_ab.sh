#!/bin/bash
# headline step with the persistent GEMM on all CUs vs on half of them per launch (two lanes: two GEMMs side by side)
for g in 256 128 256 128 192; do
  VSC_GEMM_V4_GRID=$g python bench.py --no-search --no-swin --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid', $g, d['value'], d['ms_per_step'], d['roofline']['sustained']['package_power_w'], d['roofline']['sustained']['sclk_mhz'])"
done
