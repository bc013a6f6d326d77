"""Worker of test_train_gpu.py::test_weight_gradient_kernels_agree: one stage-2 forward + backward on a fixed batch, gradients to an
.npz.  argv[2] = weight-gradient tile to force (64 | 128) through Trainer.set_dw_kernel."""
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
b = T._batch(0, 16, 24, 7)                                    # 384 rays -> 3072 rows in the NeRF layers
layers = orc.trainer_layers(b['w'])
tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
tr.set_dw_kernel(int(sys.argv[2]), 1024)
img4 = ops.images_pack(T.cu(b['images'], dev))
tr.fwd_bwd(T.cu(b['rays'], dev), T.cu(b['or_rays'], dev), T.cu(b['target'], dev), img4, T.cu(b['poses'], dev), T.cu(b['K'], dev),
           b['ref_nos'].to(dev).contiguous(), jitter=T.cu(b['jitter'], dev), jitter_dir=1, raw_noise=T.cu(b['noise'], dev), want_rgb=False)
out = {}
for i in range(len(layers)):
    W, x = tr.read('grad', i)
    out[f'W{i}'], out[f'b{i}'] = W.cpu().numpy(), x.cpu().numpy()
np.savez(sys.argv[1], **out)
