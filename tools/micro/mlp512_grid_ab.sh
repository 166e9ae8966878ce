# Swin-V2-B at 2 x 256 frames on one box: the stage-2 fused kernel with one workgroup per tile (default), and as persistent workgroups
# (VSC_SWIN_MLP512_GRID = 256: two tiles each; 128: four tiles each, half the chip per launch so that the other lane's kernel runs beside it)
mkdir -p gpurun_out/r05q
for r in 1 2 3; do
for g in 0 256 128; do
echo "VSC_SWIN_MLP512_GRID=$g:"; VSC_SWIN_MLP512_GRID=$g python tools/swin_bench.py 512 10 256 2>&1 | tail -1
done
done > gpurun_out/r05q/mlp512_grid_ab.txt 2>&1
cat gpurun_out/r05q/mlp512_grid_ab.txt
