"""Reference-descriptor extraction, one process per GPU (reference: infer/extract_ref_feats.py).

    python -m torch.distributed.run --nproc-per-node N extract_ref_feats.py \
        --checkpoint_path vit.pth --arch vit_b16_224 --zip_prefix ../data/jpg_zips \
        --input_file ../data/meta/train/train_ref_vids.txt --save_file outputs/train_refs

Same outputs: <save_file>.npz in the reference layout, videos sorted by id.  Videos shard over
ranks; each rank writes <save_file>_<rank>.npz, rank 0 merges (as the reference does)."""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from src.dataset import ZipFrames, collate_fn, vit_transform_u8
from src.extractor import extract_vsc_feat
from vsc.storage import load_features, store_features
from src.model_zoo import DEFAULT_PRECISION, WEIGHT_FORMATS, load_encoder
from vsc_hip import distributed as vdist


def main(args):
    import torch.distributed as dist
    distributed = "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
    rank, world_size = vdist.world()
    model, image_size = load_encoder(args.arch, args.weights_format, args.checkpoint_path, args.max_batch,
                                     precision=getattr(args, "precision", DEFAULT_PRECISION))
    with open(args.input_file, encoding="utf-8") as f:
        vids = [x.strip() for x in f if x.strip()]
    lo, hi = vdist.shard_bounds(len(vids), rank, world_size)
    data = ZipFrames(vids[lo:hi], args.zip_prefix, vit_transform_u8(image_size, image_size))   # uint8 to the GPU; normalised in patchify
    loader = torch.utils.data.DataLoader(data, batch_size=args.batch_size, num_workers=4, collate_fn=collate_fn)
    ids, feats, stamps = extract_vsc_feat(model, loader, device)
    np.savez(f"{args.save_file}_{rank}.npz", video_ids=ids, features=feats, timestamps=stamps)
    if distributed:
        dist.barrier()
    if rank == 0:
        parts = [np.load(f"{args.save_file}_{i}.npz") for i in range(world_size)]
        # a rank whose shard held no video (more ranks than videos, or all of its zips missing) wrote a (0, 0) feature
        # block and an empty float id array: leave such parts out of the merge
        parts = [p for p in parts if len(p["features"])] or parts[:1]
        np.savez(args.save_file + ".npz", video_ids=np.concatenate([p["video_ids"] for p in parts]),
                 features=np.concatenate([p["features"] for p in parts]),
                 timestamps=np.concatenate([p["timestamps"] for p in parts]))
        for i in range(world_size):
            os.remove(f"{args.save_file}_{i}.npz")
        store_features(args.save_file + ".npz", sorted(load_features(args.save_file + ".npz"), key=lambda v: v.video_id))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--save_file", default="test_refs")
    ap.add_argument("--zip_prefix", default="")
    ap.add_argument("--input_file", default="test/test_reference.txt")
    ap.add_argument("--checkpoint_path", required=True)
    ap.add_argument("--arch", default="vit_b16_224", help="ViT preset (vsc_hip.config) or Swin-V2 preset (vsc_hip.swin_config)")
    ap.add_argument("--weights_format", default="hf_vit", choices=WEIGHT_FORMATS)
    ap.add_argument("--batch_size", type=int, default=2, help="videos per loader batch")
    ap.add_argument("--max_batch", type=int, default=None,
                    help="frames per encoder step; default: the backbone's tile-aligned batch (ViT-B/16: 332)")
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=["fp16", "bf16"],
                    help="16-bit type of the encoder's MFMA operands (same speed; fp16 = 8 x smaller rounding, DESIGN.md 3a)")
    main(ap.parse_args())
