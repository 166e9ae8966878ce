# Swin-V2-B at small call sizes: the stage-2 fused kernel (128-row tiles: 2 n tiles for n frames -- 80 of 256 CUs at 40 frames) against
# the GEMM launches it replaced (VSC_SWIN_MLP512=0), one call per step
mkdir -p gpurun_out/r05q
for n in 8 16 24 40 64 96 128 160 200; do
echo "n = $n fused:   $(python tools/swin_bench.py $n 20 256 2>&1 | tail -1)"
echo "n = $n GEMMs:   $(VSC_SWIN_MLP512=0 python tools/swin_bench.py $n 20 256 2>&1 | tail -1)"
done > gpurun_out/r05q/swin_small_batch_ab.txt 2>&1
cat gpurun_out/r05q/swin_small_batch_ab.txt
