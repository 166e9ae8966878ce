# Swin-V2-B at 2 x 256 frames on one box: stage-2 second half as (a) two GEMMs + proj_ln, (b) fused MLP + proj_ln, (c) proj + MLP fused
mkdir -p gpurun_out/r05n
for r in 1 2; do
echo "two-GEMM MLP, proj_ln launch:"; VSC_SWIN_MLP512=0 python tools/swin_bench.py 512 10 256 2>&1 | tail -1
echo "fused MLP, proj_ln launch:";    VSC_SWIN_PROJ512=0 python tools/swin_bench.py 512 10 256 2>&1 | tail -1
echo "proj + MLP fused:";             python tools/swin_bench.py 512 10 256 2>&1 | tail -1
done > gpurun_out/r05n/swin_ab.txt 2>&1
cat gpurun_out/r05n/swin_ab.txt
