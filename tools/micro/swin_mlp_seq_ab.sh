for r in 1 2; do
echo "interleaved (default):"; python tools/swin_bench.py 512 10 256 2>&1 | tail -1
echo "SEQ:"; VSC_SWIN_MLP_SEQ=1 python tools/swin_bench.py 512 10 256 2>&1 | tail -1
done
