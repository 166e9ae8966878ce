#!/bin/bash
# Reference-side extraction of the ensemble, one process per GPU (stands where the reference's infer/infer_ref.sh stands):
# every backbone over the train and test reference videos, then concat + PCA + score normalisation.
#   CKPT=../checkpoints ZIPS=../data/jpg_zips META=../data/meta bash infer_ref.sh
set -e
cd "$(dirname "$0")"
export PYTHONPATH=$PYTHONPATH:$PWD
CKPT=${CKPT:-../checkpoints}; ZIPS=${ZIPS:-../data/jpg_zips}; META=${META:-../data/meta}; OUT=${OUT:-./outputs}
GPUS=${GPUS:-$(rocm-smi --showid 2>/dev/null | grep -c "GPU\[" || echo 1)}; [ "$GPUS" -ge 1 ] || GPUS=1
PRECISION=${PRECISION:-fp16}     # MFMA operand type of the encoders; bf16 = the benchmarked configuration
# name : preset : weight naming  (the reference's four descriptor models, infer_ref.sh:7-8)
MODELS=("swinv2_v115:swinv2_base_256:swin_ref" "swinv2_v107:swinv2_base_256:swin_ref" "swinv2_v106:swinv2_base_256:swin_ref" "vit_v68:vit_v68:timm_vit")
for m in "${MODELS[@]}"; do
  IFS=: read -r name arch fmt <<< "$m"
  mkdir -p "$OUT/$name"
  for split in train test; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$GPUS" --master-addr 127.0.0.1 extract_ref_feats.py \
      --zip_prefix "$ZIPS" --input_file "$META/$split/${split}_ref_vids.txt" --save_file "$OUT/$name/${split}_refs" \
      --checkpoint_path "$CKPT/$name.torchscript.pt" --arch "$arch" --weights_format "$fmt" --batch_size 2 --precision "$PRECISION"
  done
done
python concat_pca_sn.py --root "$OUT" --models swinv2_v115 swinv2_v107 swinv2_v106 vit_v68 --pca_model "$CKPT/pca_model.pkl" ${FIT_PCA:+--fit_pca}
