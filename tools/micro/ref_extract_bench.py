"""Reference-side extraction loop (src.extractor.extract_vsc_feat, the body of extract_ref_feats.py) on decoded frames: 128 videos of
60 uint8 frames in loader batches of 2 videos through ViT-B/16 -- frames/s of the loop against the encoder's own rate.  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from tools import synth
from src.dataset import CLIP_MEAN, CLIP_STD, TensorFrames, collate_fn
from src.extractor import extract_vsc_feat
from vsc_hip.config import aligned_batch, get_config
from vsc_hip.encoder import HipEncoder
dev = torch.device("cuda:0")
cfg = get_config("vit_b16_224")
enc = HipEncoder(cfg, synth.encoder_weights(3, cfg), max_batch=aligned_batch(cfg.tokens), l2_normalize=True, u8_mean=CLIP_MEAN, u8_std=CLIP_STD)
base = torch.from_numpy(synth.uniform(1, (60, 224, 224, 3), 0.0, 256.0).astype(np.uint8))
vids = [(base.clone(), f"R{i:06d}") for i in range(128)]
x = base.to(dev).repeat(12, 1, 1, 1)[:664].contiguous()
enc(x); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): enc(x)
torch.cuda.synchronize(); rate = 664 * 5 / (time.perf_counter() - t0)
def per_batch(loader):      # the reference's loop: one encoder call, one pageable upload and one copy back per loader batch
    out = []
    for frames, mask, video_id in loader:
        m = mask.to(dev).bool()
        out.append(enc(frames.to(dev)[m]).float().cpu().numpy())
    return np.concatenate(out)


for _ in range(2):
    loader = torch.utils.data.DataLoader(TensorFrames(vids), batch_size=2, collate_fn=collate_fn)
    t0 = time.perf_counter()
    ref = per_batch(loader)
    dtb = time.perf_counter() - t0
print(f"per loader batch   : {ref.shape[0]} frames in {dtb * 1e3:.0f} ms = {ref.shape[0] / dtb:.0f} frames/s")
for _ in range(2):
    loader = torch.utils.data.DataLoader(TensorFrames(vids), batch_size=2, collate_fn=collate_fn)
    t0 = time.perf_counter()
    ids, feats, stamps = extract_vsc_feat(enc, loader, dev)
    dt = time.perf_counter() - t0
assert np.array_equal(feats, ref)
print(f"extract_vsc_feat   : {len(ids)} frames in {dt * 1e3:.0f} ms = {len(ids) / dt:.0f} frames/s; the encoder on resident frames: {rate:.0f} frames/s ({len(ids) / dt / rate:.2f})")
