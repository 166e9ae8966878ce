"""What torch.autocast of the SAME network costs against its own fp32 result -- a yardstick for 16-bit operand types, NOT a statement about
the reference: the reference runs its descriptor models in plain fp32 (infer/extract_query_feats.py:143-153, 171; infer/src/extractor.py:23);
only the CLIP video-score tower sits under `torch.cuda.amp.autocast()` (:159), which on CUDA means fp16.  (Round 5's DESIGN.md read that
line as "the reference runs its networks under bf16 autocast" and concluded that the HIP path was closer to fp32 than the reference's own
execution: wrong, withdrawn -- VERDICT r5 weak #2.)
transformers.Swinv2Model (the port the Swin fixtures come from) with the synthetic weights, on the noise frames and the structured frames
of the fixtures, CPU autocast to bf16 and to fp16.  (runs here, no GPU)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT,"vsc22-submission_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests/golden"))
torch.set_num_threads(8)
from sklearn.preprocessing import normalize
from tools import synth
from vsc_hip.swin_config import get_swin_config
import gen_swin_golden as gs
from transformers import Swinv2Config, Swinv2Model
cfg=get_swin_config("swinv2_base_256"); w=synth.swin_weights(gs.WEIGHT_SEED,cfg)
hf = Swinv2Model(Swinv2Config(image_size=cfg.image_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depths=list(cfg.depths), num_heads=list(cfg.heads), window_size=cfg.window_size, pretrained_window_sizes=list(cfg.pretrained_window_sizes), mlp_ratio=float(cfg.mlp_ratio), layer_norm_eps=cfg.ln_eps, hidden_act="gelu"), add_pooling_layer=False).eval()
hf.load_state_dict(gs.to_hf(w,cfg), strict=False)
def desc(x, ac):
    with torch.no_grad(), torch.autocast("cpu", dtype=ac or torch.bfloat16, enabled=ac is not None):
        tok = hf(pixel_values=x).last_hidden_state.float()
    p = tok.clamp(min=1e-6).pow(cfg.gem_p).mean(dim=1).pow(1.0/cfg.gem_p)
    return normalize((p @ torch.from_numpy(w["output_proj.weight"]).t() + torch.from_numpy(w["output_proj.bias"])).numpy())
for name, x in (("noise frames", synth.swin_frames(gs.FRAME_SEED, 2, cfg)), ("structured frames", synth.structured_frames(gs.FRAME_SEED, 6, cfg))):
    x=torch.from_numpy(x); a=desc(x,None)
    for dt in (torch.bfloat16, torch.float16):
        b=desc(x,dt)
        print(f"{name}: the port under torch.autocast({str(dt).split('.')[-1]}) against itself in fp32: max {np.abs(a-b).max():.2e}, mean |d| {np.abs(a-b).mean():.2e}")
