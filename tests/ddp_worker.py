"""Worker of tests/test_train_gpu.py::test_data_parallel_gradients_match_single_process: rank r of 2 computes the stage-2
gradients on its half of the batch, the replicas are averaged with pronerf_amd.dist.allreduce_gradients (gloo here: both ranks
share the one GPU of the test box; on a node the backend is nccl = RCCL), one Adam step, and rank 0 saves the flat arrays."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main(out_path):
    import test_train_gpu as T
    from oracle import pronerf_oracle as orc
    from pronerf_amd import ops
    from pronerf_amd.dist import allreduce_gradients
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    dev = torch.device('cuda:0')
    b = T._batch(0, 12, 16, 7)
    n = b['N'] // world
    sl = slice(rank * n, (rank + 1) * n)
    layers = orc.trainer_layers(b['w'])
    tr = ops.Trainer([W for W, _ in layers], [x for _, x in layers], max_rays=n, device=dev)
    img4 = ops.images_pack(T.cu(b['images'], dev))
    L, _ = tr.fwd_bwd(T.cu(b['rays'][sl], dev), T.cu(b['or_rays'][sl], dev), T.cu(b['target'][sl], dev), img4, T.cu(b['poses'], dev), T.cu(b['K'], dev),
                      b['ref_nos'][sl].to(dev).contiguous(), jitter=T.cu(b['jitter'][sl], dev), jitter_dir=1, raw_noise=T.cu(b['noise'][sl], dev), want_rgb=False)
    allreduce_gradients(tr)
    tr.adam_step(5e-4, weight_decay=5e-8)
    torch.cuda.synchronize()
    # every rank must hold the same parameters now
    p = tr.flat('param').cpu()
    gathered = [torch.empty_like(p) for _ in range(world)]
    dist.all_gather(gathered, p)
    if rank == 0:
        np.savez(out_path, grad=tr.flat('grad').cpu().numpy(), param=p.numpy(), same=np.array([bool(torch.equal(gathered[0], g)) for g in gathered]),
                 loss=np.array([float(L[0])]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
