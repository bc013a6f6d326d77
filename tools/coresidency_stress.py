#!/usr/bin/env python3
"""Row-checking stress test of the SHIPPED fused launches beside foreign kernels that can share their CUs.

Round 4 found that two 4-wave fused-MLP workgroups of different kernels on one CU corrupt the refine stage's batch-head rows (NOTEBOOK §12);
since then a CU never holds two fused workgroups (wide ones exclude each other by registers, narrow ones by an 84 KiB LDS request).  What CAN
still share a CU with a fused workgroup is a kernel of another stream that fits into what is left — and at N > 1 there is one: the RCCL
all-gather and the index_select of ``FrameGather`` run beside the next frame.  This tool runs the shipped shapes beside such kernels
and compares EVERY output row of EVERY call with the rows of an undisturbed one-call frame:

  narrow   1024-ray calls (4-wave workgroups: 240 VGPRs, 84 KiB LDS) beside tools/foreign_kernels.hip kind 1 (random 16-byte gathers, 4 waves,
           48 VGPRs, 16 KiB LDS), kind 2 (LDS-DMA ring of 64 KiB, the weight stream's instruction) and kind 0 (1 wave, 16 VGPRs) on three streams;
           and again beside kind 3 (v_mfma_f32_16x16x32_bf16 loop, 240 VGPRs per wave: another stream's bf16 GEMM) on all three
  wide     8192-ray calls with the 8-wave shape forced beside kinds 0 and 2, and beside kind 0 on all three streams.  Until round 6 these kernels allocated
           2 x 240 VGPRs per SIMD lane and a 16-register wave of kind 0 fitted beside them: in THAT configuration the refine stage's slower wave half returned
           wrong rows 2e-4 of the calls (tools/wide_repro.py).  Wide fused kernels now allocate 256 registers per wave: nothing shares their SIMDs
  chunked  the frame as 745 calls on four streams (ChunkedRenderer) beside the three foreign streams
  gather   whole frames through FrameGather with the collective forced on in a one-rank RCCL group and a permutation as gather index:
           all_gather_into_tensor on RCCL's stream + index_select on the side stream beside the next frame's kernels (what bench.py times at N > 1)

    python tools/coresidency_stress.py [--calls 100000] [--variant dbg] [--out profiles/r05_coresidency_stress.json]

--variant dbg (python -m pronerf_amd.build --variant dbg -DPNRF_DEBUG_EHEAD: shipped launches + a placement record in the refine epilogue)
adds the evidence that foreign workgroups really were resident beside the fused ones: the share of refine waves whose physical VGPR / LDS
base is not zero (alone on a CU / SIMD a wave always starts at zero).  Prints one JSON line."""
import argparse
import collections
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--calls', type=int, default=100000, help='narrow 1024-ray calls beside foreign kernels (the other phases scale with it)')
    ap.add_argument('--variant', default=None)
    ap.add_argument('--out', default=None)
    ap.add_argument('--no-rccl', action='store_true')
    a = ap.parse_args(argv)

    from pronerf_amd import _lib
    dbg = None
    if a.variant:
        lib = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f'libpronerf_hip_{a.variant}.so'))
        for fn, (res, args) in _lib.SIGNATURES.items():
            f = getattr(lib, fn); f.restype = res; f.argtypes = args
        _lib._lib = lib
        if hasattr(lib, 'pnrf_debug_set_ehead'):
            lib.pnrf_debug_set_ehead.restype = C.c_int
            lib.pnrf_debug_set_ehead.argtypes = [C.c_void_p, C.c_void_p]
            dbg = lib
    from pronerf_amd import synthetic
    from pronerf_amd.render import ChunkedRenderer, Renderer

    fk = C.CDLL(os.path.join(ROOT, 'pronerf_amd', 'lib', 'libforeign_kernels.so'))
    fk.foreign_launch.restype = C.c_int
    fk.foreign_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]

    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    H, W = 756, 1008
    scene = synthetic.make_scene(0, H=H, W=W, focal=815.13, rotate=True)
    weights = synthetic.make_weights(0, 'trained')
    rend = Renderer(weights, max_rays=H * W, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    N = rays.shape[0]
    dbuf = torch.zeros(N, 2, 128, device=dev) if dbg else None
    if dbg:
        assert dbg.pnrf_debug_set_ehead(dbuf.data_ptr(), rays.data_ptr()) == 0
    ref, _ = rend.render_rays(rays, or_rays)
    ref = ref.clone()
    torch.cuda.synchronize()

    fbuf = torch.randint(0, 2 ** 31 - 1, (64 * 1024 * 1024 // 4,), dtype=torch.int32, device=dev)      # 64 MiB for the foreign kernels to read
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    where = torch.zeros(4 * 4096, dtype=torch.int32, device=dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(3)]

    def foreign(kind, stream, grid, iters, rec=False):
        rc = fk.foreign_launch(kind, C.c_void_p(stream.cuda_stream), grid, iters, C.c_void_p(fbuf.data_ptr()), fbuf.numel() * 4, C.c_void_p(sink.data_ptr()),
                               C.c_void_p(where.data_ptr()) if rec else None)
        assert rc == 0, rc

    # iterations for ~target microseconds per foreign kernel, measured alone
    def tune(kind, grid, target_us):
        it = 8
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            foreign(kind, side[0], grid, it); side[0].synchronize()
            e0.record(side[0]); foreign(kind, side[0], grid, it); e1.record(side[0]); side[0].synchronize()
            us = e0.elapsed_time(e1) * 1e3
            if us > 0.7 * target_us:
                break
            it = max(it + 1, int(it * min(8.0, target_us / max(us, 1.0))))
        return it, round(us, 1)

    GR = {0: 2048, 1: 512, 2: 256, 3: 256}
    iters = {k: tune(k, GR[k], 150.0) for k in (0, 1, 2, 3)}
    res = {'library': a.variant or 'shipped', 'foreign_kernels': {str(k): {'grid': GR[k], 'iters': iters[k][0], 'us_alone': iters[k][1]} for k in iters}}

    bad = torch.zeros((), dtype=torch.int64, device=dev)
    cur = torch.cuda.current_stream()

    def placement(tag):
        """share of refine waves (debug variant) that did not start at physical VGPR / LDS base 0 in the frames just rendered"""
        if dbuf is None:
            return
        di = dbuf.view(torch.int32)
        seen = di[:, :, 16] != 0
        lds = di[:, :, 21][seen] & 0xffffffff
        gpr = di[:, :, 22][seen] & 0xffffffff
        lv, lc = torch.unique(lds, return_counts=True); gv, gc = torch.unique(gpr, return_counts=True)
        res[tag + '_placement'] = {'lanes_recorded': int(seen.sum()),
                                   'LDS_ALLOC': {f'{int(v) & 0xffffffff:08x}': int(c) for v, c in zip(lv.tolist(), lc.tolist())},
                                   'GPR_ALLOC': {f'{int(v) & 0xffffffff:08x}': int(c) for v, c in sorted(zip(gv.tolist(), gc.tolist()), key=lambda t: -t[1])[:12]}}
        dbuf.zero_()

    def phase_calls(tag, rd, chunk, calls, kinds):
        """`calls` calls of `chunk` rays on the current stream, chunks walking through the frame, foreign kernels of `kinds` on the side streams"""
        nch = N // chunk
        outs = [torch.empty(chunk, 4, device=dev) for _ in range(2)]
        bad.zero_()
        t0 = time.time()
        for i in range(calls):
            c = (i * 7) % nch
            lo = c * chunk
            if i % 32 == 0:                      # throttle: the side streams never run more than 32 calls ahead of / behind the renderer
                ev = torch.cuda.Event(); ev.record(cur)
                for s in side:
                    s.wait_event(ev)
            for k, s in zip(kinds, side):
                foreign(k, s, GR[k], iters[k][0], rec=(i == calls // 2))
            o = outs[i & 1]
            rd.render_rays(rays[lo:lo + chunk], or_rays[lo:lo + chunk], out=o)
            bad.add_((o != ref[lo:lo + chunk]).any(1).sum())
            if i % 4096 == 4095:
                torch.cuda.synchronize()
                print(f'  {tag}: {i + 1} calls, {int(bad)} rows differ', file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        res[tag] = {'calls': calls, 'rays_per_call': chunk, 'rows_compared': calls * chunk, 'rows_differ': int(bad), 'foreign_kinds': list(kinds),
                    'seconds': round(time.time() - t0, 1)}
        w = where.view(-1, 4).cpu()
        res[tag]['foreign_LDS_ALLOC_last_kind'] = dict(collections.Counter(f'{int(x) & 0xffffffff:08x}' for x in w[:GR[kinds[-1]], 2].tolist()).most_common(6))
        placement(tag)

    # ---- narrow beside gather / dma / small
    phase_calls('narrow_1024', rend, 1024, a.calls, (1, 2, 0))
    # ---- narrow beside a bf16 MFMA kernel with 240 registers per wave (a GEMM of another stream): the aggressor of tools/pkf32_coexec_probe.hip — with packed
    # fp32 in the library this phase is where the SHIPPED narrow shape returns wrong rows (profiles/r05_coresidency_stress_packed_fp32.json); without, none
    phase_calls('narrow_1024_beside_mfma', rend, 1024, a.calls, (3, 3, 3))
    # ---- narrow beside three streams of the 1-wave small kernel (round 6: the kind beside which the WIDE refine stage failed; up to 17 of its waves fit a SIMD here)
    phase_calls('narrow_1024_beside_small', rend, 1024, a.calls // 2, (0, 0, 0))
    # ---- wide (forced) beside small / dma, and beside three streams of the small kernel.  Round 6: with 240 + 240 registers per SIMD a 16-register wave of the
    # small kernel fitted beside a wide pair and the refine stage returned wrong rows for its slower wave half 2e-4 of the calls (tools/wide_repro.py, NOTEBOOK 22);
    # wide fused kernels now allocate 256 registers per wave (own_the_simd): nothing shares a SIMD with them and these phases are clean
    rw = Renderer(weights, max_rays=8192, device=dev, shape='wide')
    rw.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    phase_calls('wide_8192', rw, 8192, max(200, a.calls // 8), (0, 2, 0))
    phase_calls('wide_8192_beside_small', rw, 8192, max(400, a.calls // 4), (0, 0, 0))
    del rw
    # ---- the frame as 745 calls on four streams beside the foreign streams
    ch = ChunkedRenderer(rend, 1024, 4)
    frames = max(3, a.calls // 2000)
    bad.zero_()
    t0 = time.time()
    out = torch.zeros_like(ref)
    for f in range(frames):
        ev = torch.cuda.Event(); ev.record(cur)
        for k, s in zip((1, 2, 0), side):
            s.wait_event(ev)
            for _ in range(24):
                foreign(k, s, GR[k], iters[k][0])
        ch.render_rays(rays, or_rays, out)
        bad.add_((out != ref).any(1).sum())
    torch.cuda.synchronize()
    res['chunked_4_streams'] = {'frames': frames, 'calls': frames * -(-N // 1024), 'rows_compared': frames * N, 'rows_differ': int(bad), 'seconds': round(time.time() - t0, 1)}
    placement('chunked_4_streams')
    del ch
    # ---- whole frames through FrameGather: RCCL all-gather (one-rank group) + index_select on the side stream beside the next frame
    if not a.no_rccl:
        import torch.distributed as dist
        from pronerf_amd.dist import FrameGather
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        fg = FrameGather(N, 4, device=dev, pipelined=True, collective=True)
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).to(dev)
        fg._set_index(perm, dev, torch.float32)
        want = ref.index_select(0, perm)
        frames = max(20, a.calls // 50)
        bad.zero_()
        t0 = time.time()
        for i in range(frames):
            b = fg.acquire()
            rend.render_rays(rays, or_rays, out=fg.outs[b][:fg.count])
            fg.submit(b)
            if i >= 1:
                bad.add_((fg.frame(1 - b) != want).any(1).sum())
        fg.fence()
        bad.add_((fg.frame(b) != want).any(1).sum())
        torch.cuda.synchronize()
        res['frame_gather_rccl_ws1'] = {'frames': frames, 'rows_compared': frames * N, 'rows_differ': int(bad), 'seconds': round(time.time() - t0, 1),
                                        'backend': dist.get_backend()}
        placement('frame_gather_rccl_ws1')
        dist.barrier(); dist.destroy_process_group()
    res['total_rows_differ'] = sum(v['rows_differ'] for v in res.values() if isinstance(v, dict) and 'rows_differ' in v)
    res['total_calls'] = sum(v.get('calls', v.get('frames', 0)) for v in res.values() if isinstance(v, dict) and 'rows_differ' in v)
    line = json.dumps(res)
    print(line)
    if a.out:
        open(a.out, 'w').write(json.dumps(res, indent=1) + '\n')
    return res


if __name__ == '__main__':
    r = main()
    sys.exit(1 if r['total_rows_differ'] else 0)
