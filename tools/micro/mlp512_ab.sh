mkdir -p gpurun_out/r05g
for r in 1 2; do
VSC_SWIN_MLP512=0 python tools/swin_bench.py 512 10 256 2>&1 | tail -1
python tools/swin_bench.py 512 10 256 2>&1 | tail -1
done > gpurun_out/r05g/swin_ab.txt 2>&1
cat gpurun_out/r05g/swin_ab.txt
timeout 300 python tools/micro/mlp512_variants.py 0,1,2,3,4,0,2,3 > gpurun_out/r05g/variants.txt 2>&1; cat gpurun_out/r05g/variants.txt
