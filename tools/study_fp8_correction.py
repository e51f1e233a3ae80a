#!/usr/bin/env python
"""CPU study (no GPU): how much accuracy would the trunk keep if the two hi/lo CORRECTION terms of the 3-term split
(a_hi*w_lo + a_lo*w_hi) ran on the fp8 matrix cores (2x the f16 rate: 2 instead of 3 f16-MFMA-equivalents per
product, ceiling 1250 instead of 833 TFLOP/s-equivalent)?  Emulates per layer, with fp64 accumulation:
  f16x3      : the shipped arithmetic
  f16+fp8    : main term fp16 x fp16, correction terms e4m3 x e4m3 with one power-of-two scale per tensor
  f16+fp8e5  : same with e5m2
  f16only    : main term only
and prints the relative error of the VGG feature maps against fp32.  Result on the cfg3 weights (16 crops of 64x64):
1.5e-6 / 2.4e-5 / 4.8e-5 / 7.0e-4 rms at conv5_3.  The downstream GroupNorm chain amplifies feature errors 5-40x
into the scores (plain fp16 inputs: 0.003-0.03, budget 1e-3), so the fp8 variant would sit at 1e-4..1e-3 - at
the edge of the budget.  Not adopted this round; a finer scale granularity (per output channel / per pixel, with
the corrections in their own accumulator) is the open question.

    python tools/study_fp8_correction.py
"""
import sys, torch, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.nn.functional as F
from mmmot_amd import TrackingNet
from mmmot_amd.weights import init_module
from mmmot_amd.synth import make_pair
from mmmot_amd.pack import fold_bn, VGG_STAGES
import bench
torch.set_num_threads(8)
fusion, aff, sm, N, M, S, pts = bench.WORKLOADS['cfg3']
model = TrackingNet(**dict(bench.BASE_KW, score_fusion_arch=fusion, affinity_op=aff, softmax_mode=sm))
init_module(model, seed=0); sd = {k: v.detach() for k, v in model.state_dict().items()}
dets, info, _ = make_pair(8, 8, 64, 64, seed=1000)
x0 = dets  # [16,3,64,64]
def q8(t, fmt=torch.float8_e4m3fn):
    # per-tensor power-of-two scale into the fp8 range, round, scale back (emulates scaled fp8 operands)
    m = t.abs().max().clamp(min=1e-30)
    s = 2.0 ** torch.floor(torch.log2(torch.tensor(240.0) / m))
    return (t * s).to(fmt).to(torch.float32) / s
def h16(t): return t.half().float()
layers=[]
for s_, stage in enumerate(VGG_STAGES):
    for (idx,cin,cout,pool) in stage:
        p='appearance.layers.%d.'%s_
        w,b = fold_bn(sd[p+'%d.weight'%idx], sd[p+'%d.bias'%idx], sd[p+'%d.weight'%(idx+1)], sd[p+'%d.bias'%(idx+1)], sd[p+'%d.running_mean'%(idx+1)], sd[p+'%d.running_var'%(idx+1)], 1e-5)
        layers.append((w.float(), b.float(), pool))
def run(mode):
    x = x0.clone()
    feats=[]
    for li,(w,b,pool) in enumerate(layers):
        if mode=='fp32' or li==0:
            y = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
        else:
            ah, wh = h16(x), h16(w); al, wl = x-ah, w-wh
            main = F.conv2d(ah.double(), wh.double(), None, padding=1)
            if mode=='f16x3':
                c = F.conv2d(ah.double(), h16(wl).double(), None, padding=1) + F.conv2d(h16(al).double(), wh.double(), None, padding=1)
            elif mode=='f16+fp8':
                c = F.conv2d(q8(ah).double(), q8(wl).double(), None, padding=1) + F.conv2d(q8(al).double(), q8(wh).double(), None, padding=1)
            elif mode=='f16+fp8e5':
                f=torch.float8_e5m2
                c = F.conv2d(q8(ah,f).double(), q8(wl,f).double(), None, padding=1) + F.conv2d(q8(al,f).double(), q8(wh,f).double(), None, padding=1)
            elif mode=='f16only':
                c = 0
            y = (main + c + b.double().view(1,-1,1,1)).float()
        x = F.relu(y)
        if pool: x = F.max_pool2d(x,2,2)
        feats.append(x)
    return feats
ref = run('fp32')
for mode in ('f16x3','f16+fp8','f16+fp8e5','f16only'):
    out = run(mode)
    errs=[((o-r).abs().max()/r.abs().max()).item() for o,r in zip(out,ref)]
    rms=[(((o-r)**2).mean().sqrt()/ (r**2).mean().sqrt()).item() for o,r in zip(out,ref)]
    print(mode, 'max-rel per layer last: %.2e  rms-rel last: %.2e   (layer5 %.2e)'%(errs[-1], rms[-1], rms[5]))
