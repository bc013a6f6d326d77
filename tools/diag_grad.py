import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import test_train_gpu as T
from oracle import pronerf_oracle as orc
from pronerf_amd import ops
dev = torch.device('cuda:0')
LOW = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cu = T.cu; rel = T.rel
for seed in range(0, 6):
  for (jdir, white, a) in [(1, False, 0.0), (-1, True, 1.0)]:
    b = T._batch(seed, 12, 16, 7)
    if LOW:
        T.low_frequency_nerf(b['w'], LOW)
    loss64, img64, o64, g64 = T._oracle_grads(b, jdir, white, a, torch.float64)
    _, _, _, g32 = T._oracle_grads(b, jdir, white, a, torch.float32)
    layers = orc.trainer_layers(b['w'])
    for products in ('f32', 'f16x2'):
        tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=b['N'], device=dev)
        tr.set_products(products)
        img4 = ops.images_pack(cu(b['images'], dev))
        L, rgb = tr.fwd_bwd(cu(b['rays'], dev), cu(b['or_rays'], dev), cu(b['target'], dev), img4, cu(b['poses'], dev), cu(b['K'], dev),
                            b['ref_nos'].to(dev).contiguous(), jitter=cu(b['jitter'], dev), jitter_dir=jdir, raw_noise=cu(b['noise'], dev), white_bkgd=white, a_mmrgb=a)
        es, cs = [], []
        for li in range(26):
            gW, gb = tr.read('grad', li)
            es += [rel(gW, g64[li][0]), rel(gb, g64[li][1])]; cs += [rel(g32[li][0], g64[li][0]), rel(g32[li][1], g64[li][1])]
        es, cs = np.array(es), np.array(cs)
        r = es / (cs + 1e-5)
        print(f'seed {seed} case {(jdir, white, a)} {products}: max e {es.max():.2e} (cpu32 max {cs.max():.2e}) max ratio {r.max():.2f} median ratio {np.median(r):.2f} '
              f'worst tensors {np.argsort(-r)[:3].tolist()} loss err {abs(float(L[0]) - loss64):.1e} margin {float(o64["edge_margin"].min()):.1e}', flush=True)
