"""Worker of test_train_gpu.py::test_weight_gradient_kernels_agree: one stage-2 forward + backward on a fixed batch, gradients to an
.npz.  argv[2] = weight-gradient tile to force through Trainer.set_dw_kernel: 64 | 128 (the per-layer kernels, 3072 rows), or — on a 1287-ray
batch (10 296 rows: the fine net then runs on the engine path whose weight gradients are ONE grouped launch; not a multiple of 32 rows, and the
skip layer's 319 input columns are not a multiple of 128) — 256 (grouped gradients on 256 x 128 tiles from row 1) | 255 (square tiles).  The
.npz also records Trainer.dw_group_info() so that the test can assert which tile shape really ran."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_train_gpu as T   # noqa: E402
from oracle import pronerf_oracle as orc   # noqa: E402  (batch construction only)
from pronerf_amd import ops   # noqa: E402

dev = torch.device('cuda:0')
tile = int(sys.argv[2])
big = tile in (255, 256)
b = T._batch(0, 33, 39, 7) if big else T._batch(0, 16, 24, 7)   # 1287 rays -> 10 296 rows (engine path) | 384 rays -> 3072 rows in the NeRF layers
layers = orc.trainer_layers(b['w'])
tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
tr.set_dw_kernel(tile, 1 if big else 1024)
img4 = ops.images_pack(T.cu(b['images'], dev))
tr.fwd_bwd(T.cu(b['rays'], dev), T.cu(b['or_rays'], dev), T.cu(b['target'], dev), img4, T.cu(b['poses'], dev), T.cu(b['K'], dev),
           b['ref_nos'].to(dev).contiguous(), jitter=T.cu(b['jitter'], dev), jitter_dir=1, raw_noise=T.cu(b['noise'], dev), want_rgb=False)
out = {}
for i in range(len(layers)):
    W, x = tr.read('grad', i)
    out[f'W{i}'], out[f'b{i}'] = W.cpu().numpy(), x.cpu().numpy()
out['group_info'] = np.array(tr.dw_group_info(), dtype=np.int64)
out['rows'] = np.int64(b['N'] * 8)
np.savez(sys.argv[1], **out)
