"""Generate the video-score model golden vectors (run in the build container, NOT on the GPU box).

    python tests/golden/gen_vsm_golden.py

The reference's ``MS`` (train/train_vid_score/video/model.py:63-99) wraps transformers' BertModel, a third-party
module that is installed here; MS itself imports only torch + transformers, but instantiates the encoder with
``AutoModel.from_pretrained(bert_path)`` -- a checkpoint directory that is not available -- so the encoder is built
from a BertConfig of the same architecture and MS.forward's own lines (:79-99) are applied verbatim below.
Weights and inputs are the deterministic tensors of tools/synth.py.

Output (small, committed): tests/golden/vsm_tiny_vsm.npz with weights_seed, feats_seed, n_valid [cases],
logits [cases], states_cls [cases, H] (last hidden state of [CLS])."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "vsc22-submission_amd"))

from tools import synth  # noqa: E402
from vsc_hip.vsm_config import get_vsm_config  # noqa: E402

WEIGHT_SEED, FEAT_SEED = 17, 23


def main(preset="tiny_vsm"):
    from transformers import BertConfig, BertModel
    cfg = get_vsm_config(preset)
    w = {k: torch.from_numpy(v) for k, v in synth.vsm_weights(WEIGHT_SEED, cfg).items()}
    bc = BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                    num_attention_heads=cfg.heads, intermediate_size=cfg.mlp_dim, hidden_act="gelu",
                    max_position_embeddings=cfg.max_position, type_vocab_size=2, layer_norm_eps=cfg.ln_eps,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bc._attn_implementation = "eager"
    bert = BertModel(bc, add_pooling_layer=False).eval()
    missing, unexpected = bert.load_state_dict({k[len("bert."):]: v for k, v in w.items() if k.startswith("bert.")}, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    frame_proj = torch.nn.Sequential(torch.nn.Linear(cfg.feat_dim, cfg.hidden), torch.nn.LayerNorm(cfg.hidden)).eval()
    frame_proj.load_state_dict({k[len("frame_proj."):]: v for k, v in w.items() if k.startswith("frame_proj.")})
    output_proj = torch.nn.Linear(2 * cfg.hidden, 1).eval()
    output_proj.load_state_dict({"weight": w["output_proj.weight"], "bias": w["output_proj.bias"]})

    n_valid = [1, 5, cfg.max_frames - 1, cfg.max_frames]
    feats = torch.zeros(len(n_valid), cfg.max_frames, cfg.feat_dim)
    for i, n in enumerate(n_valid):
        feats[i, :n] = torch.from_numpy(synth.normalish(FEAT_SEED + i, (n, cfg.feat_dim)))
    with torch.no_grad():  # MS.forward, model.py:79-99
        vision_feats = frame_proj(feats)
        masks = feats.abs().sum(dim=2).gt(0)
        bz = vision_feats.size(0)
        text = torch.tensor([101, 102], dtype=torch.long)[None]
        emb = bert.get_input_embeddings()
        text_emb = emb(text).expand((bz, -1, -1))
        cls_emb, sep_emb = text_emb[:, 0], text_emb[:, 1]
        inputs_embeds = torch.cat([cls_emb[:, None], vision_feats, sep_emb[:, None]], dim=1)
        masks = torch.cat([torch.ones((bz, 2)), masks], dim=1)
        states = bert(inputs_embeds=inputs_embeds, attention_mask=masks)[0]
        m = masks.to(states.dtype)
        avg_pool = (states * m[..., None]).sum(dim=1) / (m.sum(dim=1, keepdim=True) + 1e-5)
        logits = output_proj(torch.cat([states[:, 0], avg_pool], dim=1)).squeeze(1)
    out = os.path.join(HERE, f"vsm_{preset}.npz")
    np.savez(out, weights_seed=WEIGHT_SEED, feats_seed=FEAT_SEED, n_valid=np.array(n_valid), logits=logits.numpy(),
             states_cls=states[:, 0].numpy())
    print(out, logits.numpy())


if __name__ == "__main__":
    main()
