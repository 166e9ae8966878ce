# whole-call rate of the top-100 search against a 1M x 512 bank over query counts (looking for cliffs in the work-splitting rules)
mkdir -p gpurun_out/r05q
for nq in 1000 4096 8192 12000 16384 20000 40000 65536 70000 100000 131072 200000 300000; do
python tools/knn_bench.py $nq 1000000 100 2 2>&1 | tail -1
done > gpurun_out/r05q/knn_nq_scan.txt
cat gpurun_out/r05q/knn_nq_scan.txt
