#!/bin/bash
# bench.py headline at 1 and 2 internal lanes, alternating (run on the GPU box)
for l in 1 2 1 2; do
  python bench.py --no-search --no-swin --no-cpu-baseline --steps 60 --lanes $l 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes', $l, d['value'], d['ms_per_step'], d['roofline']['sustained']['package_power_w'], d['roofline']['sustained']['sclk_mhz'])"
done
