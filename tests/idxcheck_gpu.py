import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pronerf_amd import _lib, ops, synthetic
from oracle import pronerf_oracle as orc
name = sys.argv[1]
path = os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{name}.so') if name else _lib.LIB_PATH
lib = C.CDLL(path)
for fn,(res,args) in _lib.SIGNATURES.items():
    f=getattr(lib,fn); f.restype=res; f.argtypes=args
_lib._lib = lib
dev = torch.device('cuda:0')
tot = bad = 0; maxerr = 0
for seed, kind in ((0,'trained'),(2,'spread'),(3,'trained'),(5,'trained'),(1,'default')):
    w = synthetic.make_weights(seed, kind)
    scene = synthetic.make_scene(seed, H=96, W=128, rotate=True)
    fr = orc.frame_setup(scene)
    mm_rgb, add, mul, depth = orc.sampler_forward(w['sampler'], fr['mm_input'])
    ds, idx, adds, muls = orc.sort_gather(depth, add, mul, fr['rays'][:,6:7], fr['rays'][:,7:8])
    mlp = ops.PackedMLP(ops.NET_SAMPLER, w['sampler']['W'], w['sampler']['b'])
    g_ds, g_idx, *_ = ops.sampler_fwd(mlp, fr['rays'].to(dev))
    gap = (ds[:,1:]-ds[:,:-1]).min(1)[0]
    for thr in (1e-6, 3e-6):
        m = gap > thr
        nb = int((g_idx.cpu()[m] != idx[m]).any(1).sum())
        print(f'{kind} seed {seed}: rays {len(gap)} gap>{thr:g}: {int(m.sum())} mismatching rays {nb}; max depth err {float((g_ds.cpu()-ds).abs().max()):.2e}')
