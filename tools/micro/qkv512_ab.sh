# Swin-V2-B at 2 x 256 frames on one box: stage 2 with (a) every block's own qkv GEMM, (b) blocks 1..17's qkv inside the previous block's kernel
mkdir -p gpurun_out/r05q
for r in 1 2 3; do
echo "own qkv launches (VSC_SWIN_QKV512=0):"; VSC_SWIN_QKV512=0 python tools/swin_bench.py 512 10 256 2>&1 | tail -1
echo "qkv inside the fused kernel:";          python tools/swin_bench.py 512 10 256 2>&1 | tail -1
done > gpurun_out/r05q/swin_qkv_ab.txt 2>&1
cat gpurun_out/r05q/swin_qkv_ab.txt
