# 1M x 1M x 512 top-100 (BASELINE.json configs[2]) with and without tail balancing, same box
mkdir -p gpurun_out/r05o
for r in 1 2; do
echo "one sweep (VSC_KNN_TAIL=0):"; VSC_KNN_TAIL=0 python tools/knn_bench.py 1000000 1000000 100 2 2>&1 | tail -1
echo "tail balanced:";              python tools/knn_bench.py 1000000 1000000 100 2 2>&1 | tail -1
done > gpurun_out/r05o/knn_tail_ab.txt 2>&1
cat gpurun_out/r05o/knn_tail_ab.txt
