"""Worker of tests/test_dist_gpu.py: ONE rank on the box's one GPU with the `nccl` backend (= RCCL on ROCm).  Renders a few frames through
Renderer -> FrameGather (pipelined, the collective forced on at world size 1), i.e. the code path bench.py times at N > 1:
dist.all_gather_into_tensor(async_op=True) of device tensors on RCCL's stream while the next frame renders.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pronerf_amd import synthetic
    from pronerf_amd.dist import FrameGather
    from pronerf_amd.render import Renderer
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    H, W = 60, 84
    scene = synthetic.make_scene(0, H=H, W=W, rotate=True)
    rend = Renderer(synthetic.make_weights(0, 'trained'), max_rays=H * W, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    ref, _ = rend.render_rays(rays, or_rays)
    ref = ref.clone()
    fg = FrameGather(H * W, 4, device=dev, pipelined=True, collective=True)
    assert fg.collective and fg.pipelined and fg.depth == 2 and fg.fulls[0].is_cuda
    same = []
    for i in range(5):
        b = fg.acquire()
        fg.outs[b].zero_()
        rend.render_rays(rays, or_rays, out=fg.outs[b][:fg.count])
        fg.submit(b)
        if i >= 1:                       # the previous frame's gather ran beside this frame's kernels
            same.append(bool(torch.equal(fg.frame(1 - b), ref)))
    fg.fence()
    same.append(bool(torch.equal(fg.frame(b), ref)))
    # the stream-ordered reorder of a gathered buffer that is not in frame order (block-cyclic shards at N > 1): forced here with a permutation
    # as the gather index — frame(b) must be the gathered rows in that order, for the frame whose reorder ran beside the next frame's kernels too
    perm = torch.randperm(H * W, generator=torch.Generator().manual_seed(3)).to(dev)
    fg2 = FrameGather(H * W, 4, device=dev, pipelined=True, collective=True)
    fg2._set_index(perm, dev, torch.float32)
    assert fg2._eager and fg2.frames[0].is_cuda
    want = ref.index_select(0, perm)
    reordered = []
    for i in range(5):
        b = fg2.acquire()
        fg2.outs[b].zero_()
        rend.render_rays(rays, or_rays, out=fg2.outs[b][:fg2.count])
        fg2.submit(b)
        if i >= 1:
            reordered.append(bool(torch.equal(fg2.frame(1 - b), want)))
    fg2.fence()
    reordered.append(bool(torch.equal(fg2.frame(b), want)))
    # the plain collective on a device tensor, and an all-reduce (what allreduce_gradients issues), through the same communicator
    x = torch.arange(1024, device=dev, dtype=torch.float32)
    y = torch.empty_like(x)
    dist.all_gather_into_tensor(y, x)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    ok_plain = bool(torch.equal(y, torch.arange(1024, device=dev, dtype=torch.float32))) and bool(torch.equal(x, y))
    backend = dist.get_backend()
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({'backend': backend, 'world': 1, 'frames_equal': same, 'reordered_frames_equal': reordered, 'plain_collectives_ok': ok_plain,
                      'rccl': getattr(torch.cuda, 'nccl', None) is not None and list(torch.cuda.nccl.version())}))


if __name__ == '__main__':
    main()
