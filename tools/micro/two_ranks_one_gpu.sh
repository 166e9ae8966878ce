set -x
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="--steps 5 --warmup 1 --search-nq 65536 --search-nr 1000000 --search-steps 1 --no-cpu-baseline"
# 1) RCCL with both ranks on device 0: expected to be refused
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 $B --share-device --backend nccl > gpurun_out/n2_nccl.json 2> gpurun_out/n2_nccl.err
echo "nccl rc=$?"; tail -c 1500 gpurun_out/n2_nccl.err | grep -i "error\|duplicate\|invalid" | tail -5
# 2) gloo with device tensors
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29562 bench.py --gpus 2 $B --share-device --backend gloo > gpurun_out/n2_gloo.json 2> gpurun_out/n2_gloo.err
echo "gloo rc=$?"; tail -c 800 gpurun_out/n2_gloo.err; head -c 1500 gpurun_out/n2_gloo.json
