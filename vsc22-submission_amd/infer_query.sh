#!/bin/bash
# Query-side extraction (stands where the reference's infer/infer_query.sh stands): ensemble + video-score gate + de-duplication + PCA +
# query score normalisation, per split.
#   CKPT=../checkpoints ZIPS=../data/jpg_zips META=../data/meta bash infer_query.sh
set -e
cd "$(dirname "$0")"
export PYTHONPATH=$PYTHONPATH:$PWD
CKPT=${CKPT:-../checkpoints}; ZIPS=${ZIPS:-../data/jpg_zips}; META=${META:-../data/meta}; OUT=${OUT:-./outputs}
PRECISION=${PRECISION:-fp16}
for split in ${SPLITS:-train val test}; do
  # the other split's references are the score-normalisation bank (extract_query_feats.py:47-50 of the reference)
  if [ "$split" = test ]; then NORM="$OUT/train_refs.npz"; else NORM="$OUT/test_refs.npz"; fi
  python extract_query_feats.py --split "$split" --precision "$PRECISION" \
    --models "swinv2_base_256:swin_ref:$CKPT/swinv2_v115.torchscript.pt" "swinv2_base_256:swin_ref:$CKPT/swinv2_v107.torchscript.pt" \
             "swinv2_base_256:swin_ref:$CKPT/swinv2_v106.torchscript.pt" "vit_v68:timm_vit:$CKPT/vit_v68.torchscript.pt" \
    --pca_model "$CKPT/pca_model.pkl" --zip_prefix "$ZIPS" --input_file "$META/$split/${split}_query_ids.txt" --norm_refs "$NORM" \
    --clip_checkpoint "$CKPT/clip.torchscript.pt" --vsm_checkpoint "$CKPT/vsm.torchscript.pt" --output_dir "$OUT"
done
