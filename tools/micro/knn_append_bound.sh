for a in 0 2 1; do echo "VSC_KNN_ABL=$a"; VSC_KNN_ABL=$a python tools/knn_bench.py 65536 1000000 100 3 2>&1 | tail -1; done
