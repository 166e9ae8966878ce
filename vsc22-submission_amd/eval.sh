#!/bin/bash
# Candidate generation (+ localisation when the reference's vcsl package is importable) and, with GT=<matches csv>, the descriptor-track
# uAP (stands where the reference's infer/eval.sh stands).
set -e
cd "$(dirname "$0")"
export PYTHONPATH=$PYTHONPATH:$PWD
OUT=${OUT:-./outputs}; SPLIT=${SPLIT:-test}
python -m vsc.baseline.sscd_baseline --query_features "$OUT/${SPLIT}_query_sn.npz" --ref_features "$OUT/${SPLIT}_refs_sn.npz" \
  --output_path "$OUT/" --overwrite ${GT:+--ground_truth "$GT"}
