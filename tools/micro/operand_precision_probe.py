"""Max / mean |descriptor - fixture| of every golden fixture (tests/golden: outputs of the reference's own classes) through both builds of
the library: bf16 operands (libvsc_hip.so) and fp16 operands (libvsc_hip_f16.so).  Run on the GPU box: python tools/micro/operand_precision_probe.py
-> profiles/r06_operand_precision_probe.txt"""
import sys, os, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/vsc22-submission_amd')
from tools import synth
from vsc_hip.config import get_config
from vsc_hip.encoder import HipEncoder
from vsc_hip.swin_config import get_swin_config
from vsc_hip.swin_encoder import SwinHipEncoder
G='/root/repo/tests/golden/'
dev=torch.device('cuda:0')
def l2(x): return x/np.linalg.norm(x,axis=1,keepdims=True)
for prec in ('bf16','fp16'):
    for preset in ('tiny','tiny_clip','vit_b16_224','vit_v68'):
        g=np.load(G+f'vit_{preset}.npz'); cfg=get_config(preset)
        w=synth.encoder_weights(int(g['weights_seed']),cfg)
        enc=HipEncoder(cfg,w,max_batch=8,l2_normalize=True,precision=prec)
        x=torch.from_numpy(synth.frames(int(g['frames_seed']),int(g['n_frames']),cfg)).to(dev)
        d=enc(x).cpu().numpy(); ref=g['desc_l2'] if cfg.pool=='gem' or True else g['desc_l2']
        print(prec,'vit',preset,'max %.2e mean %.2e'%(np.abs(d-ref).max(),np.abs(d-ref).mean()),flush=True)
        if preset=='vit_b16_224':
            gs=np.load(G+'vit_vit_b16_224_structured.npz')
            xs=torch.from_numpy(synth.structured_frames(int(gs['frames_seed']),int(gs['n_frames']),cfg)).to(dev)
            d=enc(xs).cpu().numpy(); print(prec,'vit structured max %.2e mean %.2e'%(np.abs(d-gs['desc_l2']).max(),np.abs(d-gs['desc_l2']).mean()),flush=True)
        enc.close()
    for preset in ('tiny_swin','tiny_swin_w8','tiny_swin_w24','swinv2_base_256','swinv2_large_384'):
        g=np.load(G+f'swin_{preset}.npz'); cfg=get_swin_config(preset)
        w=synth.swin_weights(int(g['weights_seed']),cfg)
        enc=SwinHipEncoder(cfg,w,max_batch=8,l2_normalize=True,precision=prec)
        x=torch.from_numpy(synth.swin_frames(int(g['frames_seed']),int(g['n_frames']),cfg)).to(dev)
        d=enc(x).cpu().numpy(); ref=l2(g['desc'])
        print(prec,'swin',preset,'max %.2e mean %.2e'%(np.abs(d-ref).max(),np.abs(d-ref).mean()),flush=True)
        if preset=='swinv2_base_256':
            gs=np.load(G+'swin_swinv2_base_256_structured.npz')
            xs=torch.from_numpy(synth.structured_frames(int(gs['frames_seed']),int(gs['n_frames']),cfg)).to(dev)
            d=enc(xs).cpu().numpy(); print(prec,'swin structured max %.2e mean %.2e'%(np.abs(d-gs['desc_l2']).max(),np.abs(d-gs['desc_l2']).mean()),flush=True)
        enc.close()
