#!/bin/bash
# Memory-side read bytes of the sweep kernel (rocprofv3 --pmc FETCH_SIZE, KiB; x2 on gfx950) at nq = 8192 x 1M:
# work order (VSC_KNN_XCD_MAP) : filter (VSC_KNN_ABL, 1 = none).
# Run through gpurun from the repo root.  CFGS="1:0 1:2" selects cases.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/knn_pmc_ab; mkdir -p $OUT
for cfg in ${CFGS:-0:0 1:0 0:1 1:1 1:2}; do
  IFS=: read map abl <<< "$cfg"
  tag=map${map}_abl${abl}
  (cd /tmp && VSC_KNN_XCD_MAP=$map VSC_KNN_ABL=$abl timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/$tag -o knn --output-format csv -- python $OLDPWD/tools/knn_bench.py ${NQ:-8192} 1000000 100 1 > /dev/null 2> $OUT/$tag.err)
  python tools/pmc_summarize.py $OUT/$tag.json $OUT/$tag 2>/dev/null | grep sweep | sed "s/^/$tag: /"
done
