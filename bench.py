#!/usr/bin/env python3
"""bench.py — rays/s of the ProNeRF inference hot path (render_rays) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): one 1008x756 LLFF-Fern-geometry frame =
762 048 rays, 8 samples/ray, 4 neighbour views, 48 ray-encoding points; synthetic poses/images
and seeded "trained-like" weights (no dataset/checkpoint ships).  A step = one pass of the hot
path (sampler MLP in two passes: plain fp16 for every ray + fp32-grade split fp16 for the rays whose depth order the first pass cannot
decide | neighbour projection + refine MLP fp16 | NeRF MLP bf16 + alpha compositing) over one frame: ONE pnrf_render_rays_fwd call renders the whole frame, as the reference
renders it in one render() call (run_S_eS_eN_alter_trt.py:329); the "1024-ray chunks" of BASELINE.json
configs[1] are four of the 256-column workgroup batches each persistent kernel walks through inside its
single launch.  Rays, images and weights are resident
in HBM before the timed region, exactly like the reference's timed loop
(run_S_eS_eN_alter_trt.py:327-332).  At N>1 rank r renders the 1024-ray blocks r, r + N, r + 2N, ... of the frame
(--partition cyclic, the default: contiguous ranges put the rays the sampler's second pass re-renders on one rank), and the per-rank
[n,4] rgb+depth tiles are all-gathered over RCCL inside the timed region, the gather of frame i behind the render of frame i + 1
("strong" scaling: the frame is fixed).

Rank 0 prints ONE JSON line (see the driver contract).  Every line carries `frame_sha256` — the assembled frame of the last timed step: the same
digest at every N — and, at N > 1, `per_rank` (device, rays, render ms alone, gather ms alone: measured behind the timed region, so that a scaling
line explains itself).  At N=1 it carries `roofline`, `cpu_baseline`
(the CPU oracle on the host cores, 65 536 rays of the same frame) and `gpu_eager_baseline`: the oracle's
eager fp32 torch graph on the same GPU, whole frame in one call, device events — BASELINE.md §4 item 2,
"the reference single-GPU PyTorch rays/s" that BASELINE.json's >= 10x target is measured against.  No
published number exists for the metric, so `vs_baseline` is value / that measured baseline
and `vs_baseline_kind` says so.  Both baselines run after the timed region; oracle/ is imported only there.

Also after the timed region, at N=1 (each can be switched off): `steady_state` — the timed loop again over >= 2 s of frames, with socket power /
package cap / shader clock polled from rocm-smi meanwhile (the path is power-limited: DESIGN.md 4.4); `shard_rehearsal` — this GPU's time on the
first / middle / last rank's share of the frame at N = 2, 4, 8 (contiguous and block-cyclic) and on 1024- / 4096-ray calls; `chunked_1024` — the
same frame as 745 calls of <= 1024 rays (the literal reading of configs[1]): on one stream, over four streams, and each as one hipGraph, next to
the one-call figure; `variants` — the frame with other operand types / the round-2 sampler; `weights_optimizer` — the frame on the only
optimizer-trained nets in the tree (the fixture the package's own stage-1 / stage-2 drivers produced: more second-pass rays, the NeRF-class fine net),
with its rgb PSNR against the eager oracle; `train` — the training iterations of configs[3] /
configs[4] (stage-2 iteration at 4096 rays x 17 views of 756x1008; stage-1 exploration iterations at 64 and 256 samples per ray) with their
algorithmic TFLOP/s, a per-kernel table measured in the run, HBM bytes from the newest committed profiles/r*_train_pmc_summary.json (quoted with
its provenance, refused when the trainer's sources changed since) and the same iteration as eager torch autograd + torch.optim.Adam on this GPU.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, FOCAL = 756, 1008, 815.13
# algorithmic dense-MLP FLOPs per ray: 2 x MAC, bias/activations/encodings excluded
MAC_SAMPLER = 288 * 256 + 5 * 256 * 256 + 256 * 27            # 408 320
MAC_REFINE = 144 * 256 + 5 * 256 * 256 + 256 * 35             # 373 504
MAC_NERF = 8 * (63 * 256 + 6 * 256 * 256 + 283 * 4)           # 8 x 410 476
FLOP_PER_RAY = 2 * (MAC_SAMPLER + MAC_REFINE + MAC_NERF)      # 8 131 264
PEAK_BF16 = 2500.0     # TFLOP/s dense, MI355X_MICROARCH.md chip table
PEAK_F32 = 157.3       # TFLOP/s, f32-input MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10, help='the chip needs ~5 frames from idle to its steady clock')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle timing (rank 0, N=1)')
    ap.add_argument('--cpu-sample-rays', type=int, default=65536, help='SURVEY.md §8(d): >= 65 536 rays')
    ap.add_argument('--no-gpu-eager-baseline', action='store_true', help='skip the eager-PyTorch-on-GPU baseline (rank 0, N=1)')
    ap.add_argument('--no-sustained', action='store_true', help='skip the pure-MFMA ceiling probe (a child process; skipped under rocprofv3)')
    ap.add_argument('--eager-reps', type=int, default=5)
    ap.add_argument('--no-chunked', action='store_true', help='skip the 745 x 1024-ray rendering of the frame (rank 0, N=1)')
    ap.add_argument('--no-variants', action='store_true', help='skip the frame with the single-pass split-fp16 sampler (rank 0, N=1)')
    ap.add_argument('--no-shard-rehearsal', action='store_true', help='skip the one-GPU rehearsal of the 1/2, 1/4, 1/8-frame shards and small calls (rank 0, N=1)')
    ap.add_argument('--steady-seconds', type=float, default=2.0, help='seconds of back-to-back frames after the timed region (steady_state block; 0 = skip)')
    ap.add_argument('--no-train', action='store_true', help='skip the training-iteration block (rank 0, N=1)')
    ap.add_argument('--no-optimizer-weights', action='store_true', help='skip the frame on the optimizer-trained nets (weights_optimizer block; rank 0, N=1)')
    ap.add_argument('--no-train-eager', action='store_true', help='training block without the eager-torch legs')
    ap.add_argument('--launch-timeout', type=float, default=900.0, help='N>1 started without a launcher: seconds before the ranks are stopped')
    ap.add_argument('--partition', default='cyclic', choices=('cyclic', 'contiguous'),
                    help="N>1: how the frame's rays are dealt to the ranks: blocks of 1024 rays round-robin (every rank sees the frame's average share of "
                         "second-pass rays) or contiguous ranges (SURVEY.md 8(e); the slowest shard then sets the frame rate)")
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend for N>1 ('nccl' = RCCL; 'gloo' to rehearse "
                    'the multi-rank path with several ranks sharing one GPU)')
    return ap.parse_args()


def host_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, capped at the GPU
    box's per-GPU CPU share (16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(weights, scene, n_rays, budget_s=20.0):
    """The CPU oracle (oracle/, a from-scratch port of the reference's torch path) timed on this
    box's host cores over a bounded sample of the same frame.  Checker-side code: imported here
    only, never on the measured path."""
    from oracle import pronerf_oracle as orc
    torch.set_num_threads(host_cores())
    fr = orc.frame_setup(scene)
    n_total = fr['rays'].shape[0]
    sel = torch.linspace(0, n_total - 1, n_rays).long()
    rays, or_rays = fr['rays'][sel].contiguous(), fr['or_rays'][sel].contiguous()
    best, reps, t_start = float('inf'), 0, time.perf_counter()
    with torch.no_grad():
        while reps < 3 or (time.perf_counter() - t_start < budget_s and reps < 10):
            t0 = time.perf_counter()
            orc.render_rays_infer(weights, rays, or_rays, fr['images'], fr['proj'])
            best = min(best, time.perf_counter() - t0)
            reps += 1
            if time.perf_counter() - t_start > budget_s:
                break
    return {'value': n_rays / best, 'unit': 'rays/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n_rays} rays evenly strided over the same 1008x756 frame, full-size neighbour images, '
                      f'best of {reps} passes of oracle.render_rays_infer (fp32 torch CPU)'}


def gpu_eager_baseline(weights, scene, dev, reps=5):
    """the reference-style eager PyTorch path on this GPU — the oracle's torch graph (validated against the
    reference's own outputs, tests/test_oracle_golden.py) with every tensor on `dev`, fp32 (TF32 off), unfused, the whole 762 048-ray
    frame in one call like run_S_eS_eN_alter_trt.py:329, `mm_input` precomputed outside the timed call as in the reference
    (trt.py:274-278), device events, `reps` timed calls after two warm-up calls.  Checker-side code, run after the timed region."""
    from oracle import pronerf_oracle as orc
    torch.backends.cuda.matmul.allow_tf32 = False
    wd = {k: {'W': [torch.as_tensor(w).to(dev) for w in v['W']], 'b': [torch.as_tensor(b).to(dev) for b in v['b']]} for k, v in weights.items()}
    fr = orc.frame_setup(scene)
    rays, or_rays, mm_input = fr['rays'].to(dev), fr['or_rays'].to(dev), fr['mm_input'].to(dev)
    images, proj = fr['images'].to(dev), fr['proj'].to(dev)
    n = rays.shape[0]
    t1, t2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    with torch.no_grad():
        for i in range(reps + 2):
            t1.record()
            out = orc.render_rays_infer(wd, rays, or_rays, images, proj, mm_input=mm_input)
            t2.record()
            torch.cuda.synchronize()
            if i >= 2:
                ms.append(t1.elapsed_time(t2))
            rgb = out['rgb']
            del out
    del wd, fr, rays, or_rays, mm_input, images, proj
    torch.cuda.empty_cache()
    mean = sum(ms) / len(ms)
    return {'value': n / mean * 1e3, 'unit': 'rays/s', 'ms_per_frame': mean, 'ms_per_frame_best': min(ms), 'reps': reps, 'rays': n, 'dtype': 'f32',
            'kind': 'port', 'what': 'oracle torch graph on the same GPU (eager, fp32, unfused, whole frame in one call; BASELINE.md §4 item 2)',
            'torch': torch.__version__}, rgb


def optimizer_weights_leg(scene, dev, n_total, frames=20, eager_reps=2, fixture='pictures'):
    """The frame on the only optimizer-shaped nets in the tree (tests/golden/trained_synth_scene.npz: this package's stage-1 + stage-2 drivers on the
    synthetic LLFF scene; sampler, refine and the NeRF-CLASS fine net the trainers save — run_S_eS_eN_alter_trt.py:468-481 loads exactly such a
    checkpoint): ms per frame, the share of rays the two-pass sampler sends through its second pass (27 % here against 15 % on the seeded set the
    headline times), per-kernel ms, and rgb PSNR against the oracle's eager graph on the same nets.  After the timed region; rank 0, N = 1."""
    from pronerf_amd import synthetic
    from pronerf_amd.render import Renderer
    from pronerf_amd.workloads import timed_ms
    try:
        w = synthetic.load_trained_fixture(fixture)
        if fixture == 'scene3d':      # the nets trained on the consistent 3-D scene, on THAT scene: hold-out pose 0 at the Fern frame size (tests/llff_synth.py Scene3D)
            scene = synthetic.scene3d_frame(0, 4)
            assert scene['H'] * scene['W'] == n_total
    except (OSError, KeyError, ImportError, AssertionError) as e:
        return {'skipped': f'fixture not usable: {e!r}'}
    rend = Renderer({k: w[k] for k in ('sampler', 'refine', 'nerf')}, max_rays=n_total, device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W)
    out = torch.empty(n_total, 4, device=dev)
    ms = timed_ms(lambda: rend.render_rays(rays, or_rays, out=out), frames, 5)[0]
    rend.ctx.profile_begin(frames)
    for _ in range(frames):
        rend.render_rays(rays, or_rays, out=out)
    torch.cuda.synchronize()
    prof, _ = rend.ctx.profile_end()
    n2, n3 = rend.ctx.sampler_stats(), rend.ctx.sampler_saturated()
    res = {'weights': f"tests/golden/{synthetic.FIXTURES[fixture]} (optimizer-trained: stage-1 {int(w['info']['stage1_iters'])} + stage-2 {int(w['info']['stage2_iters'])} iterations; NeRF-class fine net)"
                      + ('; frame = hold-out pose 0 of the consistent 3-D scene those nets were trained on, at 756 x 1008' if fixture == 'scene3d' else '; frame = the seeded bench scene'),
           'ms_per_frame': ms, 'rays_per_s': n_total / ms * 1e3, 'kernels_ms': prof,
           'sampler_two_pass': {'rays_second_pass': n2, 'fraction': n2 / n_total, 'rays_third_pass': n3}}
    if fixture == 'scene3d':          # the reference's quality figure on this frame: PSNR against the ground-truth picture (pixel (4j, 4i) is the ray of its pixel (j, i))
        small = out[:, :3].reshape(756, 1008, 3)[::4, ::4].reshape(-1, 3)
        gt = torch.as_tensor(scene['gt_small']).reshape(-1, 3).to(dev)
        res['psnr_vs_ground_truth_db'] = float(-10.0 * torch.log10(((small - gt) ** 2).mean()))
        vq = Renderer({k: w[k] for k in ('sampler', 'refine', 'nerf')}, max_rays=n_total, device=dev, preset='quality')
        vq.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
        outq = torch.empty(n_total, 4, device=dev)
        res['quality_preset_ms_per_frame'] = timed_ms(lambda: vq.render_rays(rays, or_rays, out=outq), frames, 5)[0]
        del vq
        keep = out.clone()
        frac, var = rend.calibrate(rays, or_rays)                      # Renderer.calibrate: the exact single-pass sampler once most rays are re-rendered anyway
        res['calibrated'] = {'second_pass_fraction_seen': frac, 'sampler_variant_chosen': var,
                             'ms_per_frame': timed_ms(lambda: rend.render_rays(rays, or_rays, out=out), frames, 5)[0]}
        out.copy_(keep)                                                # the comparisons below are the default preset's frame
        del keep
    if eager_reps > 0:
        from oracle import pronerf_oracle as orc                        # checker only, after every timed loop of this leg
        torch.backends.cuda.matmul.allow_tf32 = False
        td = lambda x: torch.as_tensor(x).to(dev)
        wd = {k: {'W': [td(x) for x in w[k]['W']], 'b': [td(x) for x in w[k]['b']]} for k in ('sampler', 'refine')}
        c = w['nerfcls']
        pair = lambda p: (td(p[0]), td(p[1]))
        wd['nerfcls'] = {'pts_linears': [pair(p) for p in c['pts_linears']], 'feature_linear': pair(c['feature_linear']), 'alpha_linear': pair(c['alpha_linear']),
                         'views_linears': [pair(c['views_linears'][0])], 'rgb_linear': pair(c['rgb_linear'])}
        fr = orc.frame_setup(scene)
        t1, t2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ems = []
        with torch.no_grad():
            for i in range(eager_reps + 1):
                t1.record()
                ref = orc.render_rays_infer(wd, rays, or_rays, fr['images'].to(dev), fr['proj'].to(dev), mm_input=fr['mm_input'].to(dev), nerf='nerfcls')
                t2.record(); torch.cuda.synchronize()
                if i >= 1:
                    ems.append(t1.elapsed_time(t2))
        mse = float(((out[:, :3].double() - ref['rgb'].double()) ** 2).mean())
        res['hip_vs_eager_rgb_psnr_db'] = (10.0 * float(np.log10(1.0 / mse))) if mse > 0 else float('inf')
        if fixture == 'scene3d':
            mq = float(((outq[:, :3].double() - ref['rgb'].double()) ** 2).mean())
            res['quality_preset_vs_eager_rgb_psnr_db'] = (10.0 * float(np.log10(1.0 / mq))) if mq > 0 else float('inf')
        res['eager_ms_per_frame'] = sum(ems) / len(ems)
        res['vs_eager'] = res['eager_ms_per_frame'] / ms
        del ref, wd, fr
    del rend, out
    torch.cuda.empty_cache()
    return res


def sustained_mfma_peak():
    """TFLOP/s of a pure v_mfma_f32_16x16x32_bf16 loop (no LDS, no VALU, no barrier; 2 waves per SIMD on every CU) on THIS GPU with random and
    with constant operand bits — tools/mfma_ceiling.hip, built by pronerf_amd.build into pronerf_amd/lib/mfma_ceiling.  A separate process,
    outside the timed region; None if the binary is not there.  `roofline.peak` stays the spec figure."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pronerf_amd', 'lib', 'mfma_ceiling')
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        d['what'] = 'pure MFMA loop, measured in this run by tools/mfma_ceiling.hip: what the pipes sustain at the power limit'
        return d
    except Exception:
        return None


def csrc_digest(scope='inference'):
    """Digest of the kernel sources + build flags (pronerf_amd.build._digest; scope 'inference' leaves out the trainer-only sources, 'all' is
    what the library's .sha256 stamp holds).  Profiles record it (tools/pmc_summary.py: inference, tools/train_pmc_summary.py: training); a number
    read from a profile is reported only while it still matches."""
    try:
        from pronerf_amd import build as b
        return b._digest(scope)
    except Exception:
        return None


def profile_provenance(summary, scope='inference'):
    """(usable, provenance) of a committed PMC summary: usable only if it was taken from the kernel sources of this tree."""
    have, now = summary.get('csrc_digest'), csrc_digest(scope)
    prov = {'file': summary.get('_file'), 'commit': summary.get('commit'), 'csrc_digest': have}
    if not have:
        return False, dict(prov, refused='the profile predates digest stamping (cannot tell whether csrc/ changed since)')
    if have != now:
        return False, dict(prov, refused='csrc/ changed since the profile was taken (digest mismatch): re-run tools/profile_round.sh')
    return True, prov


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc summary (profiles/r<round>_v<version>_pmc_summary.json,
    newest = highest (round, version), not lexicographic: v9 < v11): separate FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH_SIZE x2
    correction applied there.  -> (bytes or None, provenance): None when there is no summary, it does not list the kernel, or it was taken
    from other kernel sources than this tree's (the reason is in the provenance)."""
    import re

    def key(f):
        m = re.search(r'r(\d+)(?:_v(\d+))?_pmc_summary\.json$', os.path.basename(f))
        return (int(m.group(1)), int(m.group(2) or 0)) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.json')), key=key)
    files = [f for f in files if '_train_' not in os.path.basename(f)]
    if not files:
        return None, {'refused': 'no profiles/r*_pmc_summary.json'}
    try:
        d = json.load(open(files[-1]))
        d['_file'] = os.path.relpath(files[-1], ROOT)
        ok, prov = profile_provenance(d)
        return (d['per_kernel'][kernel]['hbm_bytes_per_launch'] if ok else None), prov
    except Exception as e:
        return None, {'refused': f'{type(e).__name__}: {e}'}



class PowerPoll:
    """`rocm-smi --showpower --showmaxpower --showclocks` every ~0.3 s on a thread while a loop runs on the GPU: socket power against the package cap and
    the shader clock, for the `steady_state` block (the frame is power-limited: DESIGN.md 4.4).  Best effort: without rocm-smi, or if its output does
    not parse, the block carries no `power` entry."""

    def __init__(self):
        import shutil
        import threading
        self.exe = shutil.which('rocm-smi') or ('/opt/rocm/bin/rocm-smi' if os.path.exists('/opt/rocm/bin/rocm-smi') else None)
        self.samples, self.cap, self.stop = [], None, threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True) if self.exe else None

    def _run(self):
        import re
        import subprocess
        while not self.stop.is_set():
            try:
                out = subprocess.run([self.exe, '--showpower', '--showmaxpower', '--showclocks'], capture_output=True, text=True, timeout=10).stdout
                w = re.search(r'GPU\[0\].*Current Socket Graphics Package Power \(W\):\s*([0-9.]+)', out)
                c = re.search(r'GPU\[0\].*Max Graphics Package Power \(W\):\s*([0-9.]+)', out)
                k = re.search(r'GPU\[0\].*sclk clock level:.*\((\d+)Mhz\)', out)
                if c:
                    self.cap = float(c.group(1))
                if w:
                    self.samples.append((float(w.group(1)), int(k.group(1)) if k else None))
            except Exception:
                return
            self.stop.wait(0.3)

    def __enter__(self):
        if self.thread:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.thread:
            self.thread.join(timeout=15)

    def result(self):
        if not self.samples:
            return None
        ws = [w for w, _ in self.samples]
        ks = [k for _, k in self.samples if k]
        return {'socket_w_mean': sum(ws) / len(ws), 'socket_w_max': max(ws), 'package_cap_w': self.cap, 'sclk_mhz_mean': (sum(ks) / len(ks)) if ks else None,
                'samples': len(ws), 'what': 'rocm-smi polled on a thread while the steady-state frames ran (GPU 0 of this process\'s view)'}


def chunked_1024(rend, rays, or_rays, ref, chunk=1024, reps=2, streams=4):
    """configs[1] read literally: the frame as ceil(n / 1024) pnrf_render_rays_fwd calls of <= 1024 rays each (744 x 1024 + 192 for the Fern
    frame), (a) launched back to back on ONE stream, (b) the same call sequence replayed as ONE hipGraph, (c) / (d) the same calls round-robin
    over `streams` HIP streams with a context each (pronerf_amd.render.ChunkedRenderer: a 1024-ray call is four dependent one-batch kernels on
    8 / 16 / 8 / 64 of the 256 CUs, so its time is latency and several chunks fit side by side), eager and as one hipGraph.  Every variant
    must equal the one-call frame `ref` bit for bit (rays are independent; a ray's instruction stream does not depend on the launch shape)."""
    from pronerf_amd.render import ChunkedRenderer
    n = rays.shape[0]
    out = torch.empty_like(ref)
    bounds = [(a, min(n, a + chunk)) for a in range(0, n, chunk)]

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) * 1e3 / reps

    def graphed(fn, key):
        try:
            out.zero_()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()                                       # warm-up on a side stream
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                fn()
            gms, gwall = timed(g.replay)
            res[key + 'graph_ms_per_frame'] = gms
            res[key + 'graph_host_ms_per_frame'] = gwall
            res[key + 'graph_bit_identical_to_one_call'] = bool(torch.equal(out, ref))
            del g
        except Exception as e:                                # capture is best effort: report why instead of failing the bench line
            res[key + 'graph_ms_per_frame'] = None
            res[key + 'graph_error'] = f'{type(e).__name__}: {e}'[:300]

    res = {'chunk_rays': chunk, 'calls_per_frame': len(bounds), 'reps': reps}
    one = ChunkedRenderer(rend, chunk, 1)
    ms, wall = timed(lambda: one.render_rays(rays, or_rays, out))
    res['calls_ms_per_frame'] = ms
    res['calls_host_ms_per_frame'] = wall
    res['calls_bit_identical_to_one_call'] = bool(torch.equal(out, ref))
    graphed(lambda: one.render_rays(rays, or_rays, out), '')
    del one
    if streams > 1:
        many = ChunkedRenderer(rend, chunk, streams)
        out.zero_()
        ms, wall = timed(lambda: many.render_rays(rays, or_rays, out))
        key = f'streams{streams}_'
        res['streams'] = streams
        res[key + 'calls_ms_per_frame'] = ms
        res[key + 'calls_host_ms_per_frame'] = wall
        res[key + 'calls_bit_identical_to_one_call'] = bool(torch.equal(out, ref))
        graphed(lambda: many.render_rays(rays, or_rays, out), key)
        del many
    return res


def train_pmc(workload):
    """HBM bytes per training iteration from the committed profile of tools/profile_train.sh (profiles/r<NN>_train_pmc_summary.json,
    newest round): sum over the iteration's kernels of 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction).  -> (entry or None, provenance)."""
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_train_pmc_summary.json')),
                   key=lambda f: int(re.search(r'r(\d+)_train', os.path.basename(f)).group(1)))
    if not files:
        return None, {'refused': 'no profiles/r*_train_pmc_summary.json'}
    try:
        d = json.load(open(files[-1]))
        d['_file'] = os.path.relpath(files[-1], ROOT)
        ok, prov = profile_provenance(d, 'training')
        return (d['workloads'][workload] if ok else None), prov
    except Exception as e:
        return None, {'refused': f'{type(e).__name__}: {e}'}


def train_kernel_table(fn, n_rays, samples, iters=5, top=8):
    """Per-kernel table of one training iteration, measured in this run: torch.profiler (roctracer) around `iters` iterations -> mean
    microseconds per iteration of every kernel, the `top` largest listed; for the three launches that carry the fine net (forward chain,
    backward chain, grouped weight gradients) the algorithmic bytes / FLOPs of the launch and what they amount to per second.
    Kernels run ~5-10 % slower under the tracer than in the timed loop (`ms` of the block is the un-traced figure)."""
    try:
        from torch.profiler import ProfilerActivity, profile
        fn(); torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
        rows = []
        for e in prof.key_averages():
            t = getattr(e, 'device_time_total', None)
            if t is None:
                t = getattr(e, 'cuda_time_total', 0.0)
            if t > 0:
                rows.append((e.key, e.count / iters, t / iters))
    except Exception as e:                                   # the tracer is optional: report why instead of failing the bench line
        return {'error': f'{type(e).__name__}: {e}'[:300]}
    rows.sort(key=lambda r: -r[2])
    R = n_rays * samples
    # fine net (NeRF class): per-row floats the launches move by design, and MACs (pronerf_amd.workloads.layer_macs)
    fine = [(63, 256)] + [(256, 256)] * 4 + [(319, 256)] + [(256, 256)] * 2 + [(256, 256), (256, 1), (283, 128), (128, 3)]
    macs = sum(a * b for a, b in fine)
    dx_macs = macs - 63 * 256                                # no input gradient through pts0's weights into ... the embedding: kept; through rgb: kept
    act_out = 8 * 256 + 256 + 128 + 4                        # activations the forward chain writes once (fp32): pts0..7, feature, views hidden, raw
    dz_out = 8 * 256 + 256 + 64                              # gradients the backward chain writes once: dZ of pts0..7, d feature, d embedding
    dw_in = 64 + 4 * 512 + (320 + 256) + 2 * 512 + 512 + (288 + 128) + 256      # X and dZ rows the nine 256-wide weight gradients + views read
    known = {
        'tchain_fwd_kernel': {'bytes': R * 4.0 * (90 + act_out) + R * 40.0, 'flop': 2.0 * macs * R, 'bound': 'stores (fp32 activations written once) / split-fp16 MFMA rate'},
        'tchain_bwd_kernel': {'bytes': R * 4.0 * (129 + dz_out) + R * 40.0, 'flop': 2.0 * dx_macs * R, 'bound': 'stores (fp32 gradients written once) / split-fp16 MFMA rate'},
        'dwh_group_kernel': {'bytes': R * 4.0 * dw_in, 'flop': 2.0 * (macs - 256 - 128 * 3) * R, 'bound': 'hbm (re-reads every saved activation and gradient)'},
    }
    out = {'iters': iters, 'traced_us_per_iter': sum(r[2] for r in rows), 'kernels': []}
    for name, launches, us in rows[:top]:
        short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].strip()
        e = {'kernel': short[:60], 'launches_per_iter': launches, 'us_per_iter': us}
        for k, v in known.items():
            if k in name:
                e.update(algorithmic_bytes=v['bytes'], algorithmic_tflop=v['flop'] / 1e12, TBps=v['bytes'] / (us * 1e-6) / 1e12,
                         tflops=v['flop'] / (us * 1e-6) / 1e12, frac_of_8TBps=v['bytes'] / (us * 1e-6) / 8e12, bound=v['bound'])
        out['kernels'].append(e)
    return out


def train_block(dev, eager=True):
    """Training iterations at configs[3] / configs[4] size (pronerf_amd.workloads): HIP trainer vs the oracle's eager torch autograd graph +
    torch.optim.Adam on the same GPU (checker-side, after the timed region).  `frac_of_hbm_roof` = measured HBM bytes per iteration / ms / 8 TB/s:
    the largest launch of an iteration (the grouped weight gradients) is HBM-bound, the two chain launches of the fine net are bound by
    their stores and the split-fp16 MFMA rate."""
    from pronerf_amd import workloads as wl
    HBM_PEAK = 8.0e12
    wk = wl.TrainWorkload(dev, max_samples=256)
    tr = wk.trainer
    out = {'what': 'one training iteration = forward (saved activations) + backward + Adam on one fixed synthetic batch; N_rand 4096 rays of one view, '
                   '17 training views of 756x1008, NeRF-class fine net; layer products in split fp16 (fp32-grade) unless noted',
           'hbm_peak_TBps': HBM_PEAK / 1e12}

    def entry(name, fn, iters, warm, flop, extra=None):
        ms, wall = wl.timed_ms(fn, iters, warm)
        e = {'ms': ms, 'host_ms': wall, 'iters': iters, 'rays_per_s': wk.n / (ms * 1e-3), 'algorithmic_tflop': flop / 1e12, 'tflops': flop / (ms * 1e-3) / 1e12}
        pm, prov = train_pmc(name)
        e['hbm_bytes_per_iter'] = pm['hbm_bytes_per_iter'] if pm else None
        e['hbm_bytes_source'] = prov
        if pm:
            e['hbm_TBps'] = pm['hbm_bytes_per_iter'] / (ms * 1e-3) / 1e12
            e['frac_of_hbm_roof'] = e['hbm_TBps'] * 1e12 / HBM_PEAK
            e['launches_per_iter'] = pm.get('launches_per_iter')
            e['pmc_source'] = pm.get('source')
        e.update(extra or {})
        e['per_kernel'] = train_kernel_table(fn, wk.n, (extra or {}).get('samples_per_ray', 8))
        out[name] = e
        return e

    s2 = entry('stage2_iteration', wk.stage2_step, 50, 5, wl.train_flop(wk.n, 8),
               {'workload': 'configs[3]: stage-2 refine iteration (run_S_eS_eN_alter_base_refine2.py:831-878), 8 samples per ray, inverse_warp projection into 4 of 17 views per ray'})
    tr.set_products('f32')
    s2['ms_f32_products'] = wl.timed_ms(wk.stage2_step, 30, 3)[0]
    tr.set_products('f16x2')
    for n_mult in (8, 32):
        S = 8 * n_mult
        entry(f'stage1_explore_{S}', lambda: wk.explore_step(n_mult), 20 if S <= 64 else 10, 3, wl.train_flop(wk.n, S, nerf_only=True),
              {'workload': f'configs[4]: stage-1 exploration iteration (run_S_eS_eN_alter_base.py:689-729, 929-940), {S} samples per ray'
                           + (' (the reference\'s cap: n_mult <= 8)' if S == 64 else ' (BASELINE.json configs[4]; n_mult 32)'), 'samples_per_ray': S})
    if eager:
        from oracle import pronerf_oracle as orc
        torch.backends.cuda.matmul.allow_tf32 = False
        tl = [(torch.tensor(W_, device=dev, requires_grad=True), torch.tensor(b, device=dev, requires_grad=True)) for W_, b in wk.layers]
        opt = torch.optim.Adam([p for pair in tl for p in pair], lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-8)
        opt_n = torch.optim.Adam([p for pair in tl[14:] for p in pair], lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-8)
        prev = torch.get_default_device() if hasattr(torch, 'get_default_device') else 'cpu'
        torch.set_default_device(dev)
        try:
            def eager2():
                opt.zero_grad()
                loss, _, _ = orc.stage2_loss(tl, wk.rays, wk.or_rays, wk.target, wk.images_nchw, wk.poses, wk.K, wk.ref_nos, jitter=wk.jitter, jitter_dir=1,
                                             raw_noise=wk.noise)
                loss.backward(); opt.step()
            ms = wl.timed_ms(eager2, 8, 2)[1]
            out['stage2_iteration'].update(eager_torch_gpu_ms=ms, speedup_vs_eager=ms / out['stage2_iteration']['host_ms'])
            for n_mult in (8, 32):
                jd = wk.explore_jitter(n_mult)

                def eagerx():
                    opt_n.zero_grad()
                    loss, _, _ = orc.stage1_loss(tl, wk.rays, wk.or_rays, wk.target, wk.images_nchw, wk.poses, wk.K, wk.ref_nos, False, n_mult=n_mult, dir1=1,
                                                 jitter=jd, dir2=-1)
                    loss.backward(); opt_n.step()
                ms = wl.timed_ms(eagerx, 3, 1)[1]
                e = out[f'stage1_explore_{8 * n_mult}']
                e.update(eager_torch_gpu_ms=ms, speedup_vs_eager=ms / e['host_ms'])
        finally:
            torch.set_default_device(prev)
        out['eager'] = 'oracle torch graph (fp32, TF32 off) with autograd + torch.optim.Adam on the same GPU, host wall time per iteration'
        del tl, opt, opt_n
    del wk, tr
    torch.cuda.empty_cache()
    return out


def dbg(msg):
    if os.environ.get('PNRF_BENCH_DEBUG'):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def under_profiler():
    """True when a rocprofv3 / rocprofiler-sdk tool library is preloaded into this process: child GPU programs would inherit it."""
    env = os.environ
    if any('rocprof' in env.get(k, '').lower() for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB')):
        return True
    return any(k.startswith(('ROCPROF', 'ROCPROFILER_', 'ROCP_')) for k in env)


def self_launch(args):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, exactly what torch.distributed.run would set), relay rank 0's JSON line, and return non-zero if any rank
    fails or the launch times out.  This parent never touches the GPU (torch.cuda.device_count() does not initialise it on this image) and
    never execs: the ranks are fresh processes."""
    import socket
    import subprocess
    n = args.gpus
    if args.backend == 'nccl':
        ndev = torch.cuda.device_count()
        if ndev < n:
            print(f'bench.py: --gpus {n} with the nccl (RCCL) backend needs {n} visible GPUs, found {ndev} '
                  f"(use --backend gloo to rehearse the multi-rank path with ranks sharing a GPU)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), PNRF_BENCH_CHILD='1')
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.monotonic() + args.launch_timeout
    rc, out0 = 0, ''
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                if r == 0:                           # drain rank 0's pipe while waiting for it
                    try:
                        o, _ = procs[0].communicate(timeout=0.2)
                        out0 += o or ''
                    except subprocess.TimeoutExpired:
                        continue
                elif procs[r].poll() is None:
                    continue
                pending.discard(r)
                if procs[r].returncode != 0:
                    raise RuntimeError(f'rank {r} exited with code {procs[r].returncode}')
            if time.monotonic() > deadline:
                raise RuntimeError(f'launch timed out after {args.launch_timeout} s')
            if pending and 0 not in pending:
                time.sleep(0.05)
    except RuntimeError as e:
        print(f'bench.py: {e}; stopping the other ranks', file=sys.stderr)
        rc = 1
    finally:
        for p in procs:                                  # exact PIDs of the children started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    lines = [l for l in out0.splitlines() if l.startswith('{')]
    if rc == 0 and not lines:
        print('bench.py: rank 0 printed no JSON line', file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'WORLD_SIZE={world} does not match --gpus {args.gpus}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    ndev = torch.cuda.device_count()
    if args.backend == 'nccl' and world > 1 and (local_rank >= ndev or ndev < int(os.environ.get('LOCAL_WORLD_SIZE', world))):
        raise SystemExit(f'the nccl (RCCL) backend needs one GPU per rank: LOCAL_RANK={local_rank}, {ndev} GPU(s) visible, {world} ranks '
                         '(--backend gloo rehearses the multi-rank path with ranks sharing a GPU)')
    dev = torch.device('cuda', local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    dbg('process group up')
    from pronerf_amd import synthetic
    from pronerf_amd.render import RayPartition, Renderer

    weights = synthetic.make_weights(0, 'trained')
    scene = synthetic.make_scene(0, H=H, W=W, focal=FOCAL, rotate=True)
    n_total = H * W
    part = RayPartition(n_total, world, args.partition)           # world == 1: the whole frame
    count = part.count(rank)
    rend = Renderer(weights, max_rays=max(count, 1), device=dev)
    rend.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
    rays, or_rays = rend.frame_rays(scene['K'], scene['c2w'], H, W, **part.frame_rays_args(rank))
    counts = list(part.counts)
    # N > 1: the gather of frame i runs on the collective's stream while frame i+1 renders (pronerf_amd.dist.FrameGather: two output /
    # frame buffers; the wait before a buffer is reused is a stream wait, the host never blocks).  Every frame is complete when fence() returns.
    from pronerf_amd.dist import FrameGather
    pipeline = world > 1 and os.environ.get('PNRF_BENCH_PIPELINE', '1') != '0'
    fg = FrameGather(n_total, 4, device=dev, pipelined=pipeline, partition=part)
    outs = fg.outs

    last_buf = [0]

    def step():
        b = fg.acquire()
        rend.render_rays(rays, or_rays, out=outs[b][:count])
        fg.submit(b)
        last_buf[0] = b

    def fence():
        fg.fence()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dbg('renderer + inputs ready')
    for _ in range(args.warmup):
        step()
    dbg('warm-up issued')
    fence()
    dbg('warm-up done')
    # per-kernel durations over the timed region itself: the library records a HIP event before / after each of the three
    # kernels of a frame on the launch stream (pnrf_ctx_profile_begin), four event records per frame, read after the region
    PROF_FRAMES = 256
    if world == 1:
        rend.ctx.profile_begin(min(args.steps, PROF_FRAMES))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof, prof_frames = rend.ctx.profile_end() if world == 1 else (None, 0)
    dbg(f'timed region done: {dt:.3f}s')
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = bool(all(torch.isfinite(o).all().item() for o in outs))
    # the assembled frame of the last timed step, as bytes: the same digest at every N (ray shards + gather change no row) — tests/test_dist_gpu.py,
    # tests/test_bench_gpu.py compare it with the N = 1 line's
    import hashlib
    frame_sha = hashlib.sha256(fg.frame(last_buf[0]).contiguous().cpu().numpy().tobytes()).hexdigest() if rank == 0 else None
    per_rank = None
    if world > 1:
        # after the timed region: where a frame's time goes on every rank — its render alone (device events) and the exchange alone (all-gather +
        # reorder of an already rendered tile, host clock around fence) — so that a SCALE line explains itself
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier(); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            rend.render_rays(rays, or_rays, out=outs[0][:count])
        e1.record(); torch.cuda.synchronize()
        render_ms = e0.elapsed_time(e1) / reps
        dist.barrier(); torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(reps):
            fg.submit(fg.acquire())
        fg.fence(); torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) / reps * 1e3
        mine = {'rank': rank, 'device': f'{dev} ({torch.cuda.get_device_name(dev)})', 'rays': count, 'render_ms': render_ms, 'gather_ms': gather_ms,
                'rays_second_pass': rend.ctx.sampler_stats()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    sampler_f32 = False                      # the product path: default kernel variants (the library reads no environment)
    res = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = n_total * args.steps / dt
        res = {
            'metric': 'rays/sec (and ms/1008x756 frame) LLFF Fern 8-sample infer',
            'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'ms_per_frame': ms, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None,
            'dtype': 'bf16 (NeRF MLP operands) + f16 (refine MLP operands); fp32 accumulate + ' + ('f32 (sampler MLP, exact f32 MFMA)' if sampler_f32 else
                                                                  'f16 | f16x2 (sampler MLP: plain fp16 pass for every ray, split fp16 hi+lo operands = fp32-grade for '
                                                                  'the rays whose depth order that pass cannot decide; fp32 accumulate)'),
            'data': 'synthetic',
            'backend': (args.backend if world > 1 else None),
            'rccl_ranks': (dist.get_world_size() if world > 1 and args.backend == 'nccl' else None),
            'ranks': world,
            'launcher': ('self (bench.py started its own ranks)' if os.environ.get('PNRF_BENCH_CHILD') else
                         'torch.distributed.run' if world > 1 else None),
            'config': {'workload': 'LLFF fern geometry 1008x756 frame (762048 rays), 8 samples/ray, 4 neighbour views, 48 ray-encoding points, '
                                   'bf16 NeRF MLP, fp16 refine MLP (all-fp16 and all-bf16 timed beside it: variants); one pnrf_render_rays_fwd call renders the whole frame, as the reference does (the 1024-ray chunks of '
                                   'configs[1] = 4 of the 256-column workgroup batches each persistent kernel walks inside its single launch; '
                                   'the frame as 745 separate 1024-ray calls is timed beside it: chunked_1024)',
                       'rays_per_step': n_total, 'rays_per_gpu': counts[0], 'rays_per_rank': counts,
                       'gather_bytes_per_rank_per_frame': (fg.cmax * 4 * 4 if world > 1 else 0),
                       'gather_bytes_per_frame': (fg.cmax * 4 * 4 * world if world > 1 else 0),
                       'gather_pipelined': bool(pipeline), 'ray_partition': (part.kind + (f' (blocks of {part.block} rays round-robin)' if part.kind == 'cyclic' else '')),
                       'parallelism': f'ray-sharded x{world}' + (' + RCCL all-gather of [n,4] rgb+depth' if world > 1 else '')},
            'outputs_finite': finite,
            'frame_sha256': frame_sha,
            'per_rank': per_rank,
            'algorithmic_flop_per_ray': FLOP_PER_RAY,
            'e2e_mfma_tflops': value * FLOP_PER_RAY / 1e12,
        }
        if world == 1:
            flops = {'sampler_kernel': 2 * MAC_SAMPLER, 'refine_kernel': 2 * MAC_REFINE, 'nerf_kernel': 2 * MAC_NERF}
            # the sampler's algorithmic FLOPs are priced against the peak of the MFMA dtype it runs on (f16 = bf16 rate)
            peaks = {'sampler_kernel': PEAK_F32 if sampler_f32 else PEAK_BF16, 'refine_kernel': PEAK_BF16, 'nerf_kernel': PEAK_BF16}
            kern = {}
            for k, ms_k in prof.items():
                kern[k] = {'ms': ms_k}
                if k in flops:
                    ach = flops[k] * n_total / (ms_k * 1e-3) / 1e12
                    kern[k].update(achieved_tflops=ach, peak_tflops=peaks[k], frac=ach / peaks[k])
            dom = max(flops, key=lambda k: prof[k])
            # symbols as rocprofv3 prints them in profiles/*_kernel_stats.csv (default build: split-fp16 sampler, 16x16x32 NeRF stage)
            symbols = {'sampler_kernel': 'sampler_kernel' if sampler_f32 else 'sampler_p1_kernel + sampler_h16_kernel',
                       'refine_kernel': 'refine_kernel<1, 8, 1, 1, (anonymous namespace)::PrecF16>',
                       'nerf_kernel': 'nerf16_kernel<false, 2, (anonymous namespace)::PrecBf16>'}
            for k in kern:
                kern[k]['symbol'] = symbols[k]
            traffic, traffic_src = pmc_traffic(dom)
            res['roofline'] = {'bound': 'mfma', 'kernel': symbols[dom], 'stage': dom, 'achieved': kern[dom]['achieved_tflops'], 'peak': peaks[dom],
                               'unit': 'TFLOP/s', 'frac': kern[dom]['frac'], 'traffic': traffic, 'traffic_source': traffic_src,
                               'launch_ms': prof[dom], 'flop_per_launch': flops[dom] * n_total,
                               'timing': f'HIP events on the launch stream around every kernel of the first {prof_frames} timed steps'}
            res['kernels'] = kern
            sus = None if (args.no_sustained or under_profiler()) else sustained_mfma_peak()
            if sus:                            # context for `frac`: the spec peak is not reachable at this chip's power limit with real operand bits
                res['roofline']['sustained'] = dict(sus, frac_of_sustained=kern[dom]['achieved_tflops'] / sus['random_operands_tflops'])
            n2 = rend.ctx.sampler_stats()
            res['sampler_two_pass'] = {'rays_second_pass': n2, 'fraction': n2 / n_total, 'kappa': rend.ctx.sampler_kappa(),
                                       'what': 'rays whose depth order the plain-fp16 pass could not decide (adjacent sorted gap <= kappa x its own '
                                               'error bound): re-rendered by the split-fp16 kernel'}
            one_call = outs[0][:count].clone()
            if args.steady_seconds > 0:
                # the driver fixes --steps (20 frames = 90 ms): the same loop over >= steady_seconds of back-to-back frames, where the chip sits at
                # the clock its power limit allows.  `value` stays the driver's K steps; DESIGN.md quotes the steady figure when they differ by > 3 %.
                nfr = max(args.steps, int(args.steady_seconds * 1e3 / ms) + 1)
                with PowerPoll() as pw:
                    t1 = time.perf_counter()
                    for _ in range(nfr):
                        step()
                    fence()
                    sdt = time.perf_counter() - t1
                sms = sdt / nfr * 1e3
                res['steady_state'] = {'ms_per_frame': sms, 'rays_per_s': n_total * nfr / sdt, 'frames': nfr, 'seconds': sdt,
                                       'vs_timed_region': sms / ms,
                                       'what': 'the timed loop again over >= %.1f s of back-to-back frames (host clock around barrier + synchronize)' % args.steady_seconds}
                if pw.result():
                    res['steady_state']['power'] = pw.result()
            if not args.no_shard_rehearsal:
                from pronerf_amd.workloads import shard_rehearsal
                res['shard_rehearsal'] = shard_rehearsal(weights, scene, H, W, dev, reps=40)
            if not args.no_chunked:
                res['chunked_1024'] = dict(chunked_1024(rend, rays, or_rays, one_call), one_call_ms_per_frame=ms)
            var_rgb = {}
            if not args.no_variants:
                from pronerf_amd.workloads import timed_ms
                res['variants'] = {}
                for vname, vset, what in (
                        ('sampler_split', {'sampler': 'sampler_split'}, 'split-fp16 sampler kernel for every ray (PNRF_VARIANT_SAMPLER_SPLIT, the round-2 sampler)'),
                        ('nerf_f16', {'nerf': 'f16'}, 'NeRF MLP on fp16 operands too (PNRF_VARIANT_F16; raw output within the authors\' FP16-engine tolerance)'),
                        ('refine_bf16', {'refine': 'bf16'}, 'refine MLP on bf16 operands too (PNRF_VARIANT_BF16: every MLP behind the sampler in bf16)'),
                        ('refine_16x16', {'refine': 'refine_16x16'}, 'refine stage on v_mfma_f32_16x16x32_f16, four lanes per ray, folded Pluecker inputs (PNRF_VARIANT_REFINE_16X16)'),
                        ('round2', {'sampler': 'sampler_split', 'refine': 'bf16'}, 'split sampler + bf16 refine: the kernels of the round-2 default')):
                    r2 = Renderer(weights, max_rays=count, device=dev, variants=vset)
                    r2.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
                    o2 = torch.empty_like(one_call)
                    vms = timed_ms(lambda: r2.render_rays(rays, or_rays, out=o2), 20, 5)[0]
                    dm = float(((o2[:, :3].double() - one_call[:, :3].double()) ** 2).mean())
                    res['variants'][vname] = {'ms_per_frame': vms, 'rays_per_s': n_total / vms * 1e3, 'what': what,
                                              'rgb_psnr_vs_default_db': (10.0 * float(np.log10(1.0 / dm))) if dm > 0 else float('inf')}
                    var_rgb[vname] = o2[:, :3].clone()
                    del r2, o2
            if not args.no_gpu_eager_baseline:
                last = outs[0][:count, :3].clone() if not pipeline else None
                eager, eager_rgb = gpu_eager_baseline(weights, scene, dev, args.eager_reps)
                if last is not None:             # the two paths rendered the same frame: error of the HIP path against the eager fp32 graph
                    mse = float(((last.double() - eager_rgb.double()) ** 2).mean())
                    eager['hip_vs_eager_rgb_psnr_db'] = (10.0 * float(np.log10(1.0 / mse))) if mse > 0 else float('inf')
                    for vname, vr in var_rgb.items():
                        mse2 = float(((vr.double() - eager_rgb.double()) ** 2).mean())
                        res['variants'][vname]['hip_vs_eager_rgb_psnr_db'] = (10.0 * float(np.log10(1.0 / mse2))) if mse2 > 0 else float('inf')
                del eager_rgb, var_rgb
                res['gpu_eager_baseline'] = eager
                res['vs_baseline'] = value / eager['value']
                res['vs_baseline_kind'] = ('value / gpu_eager_baseline.value, measured in this run (BASELINE.md §4 item 2: the denominator of the >= 10x '
                                           'target); the reference publishes no number for this metric')
            if not args.no_optimizer_weights:
                res['weights_optimizer'] = optimizer_weights_leg(scene, dev, n_total, eager_reps=0 if args.no_gpu_eager_baseline else min(2, args.eager_reps))
                res['weights_scene3d'] = optimizer_weights_leg(scene, dev, n_total, eager_reps=0 if args.no_gpu_eager_baseline else min(2, args.eager_reps), fixture='scene3d')
            if not args.no_train:
                del rend
                torch.cuda.empty_cache()
                res['train'] = train_block(dev, eager=not args.no_train_eager)
            if not args.no_cpu_baseline:
                res['cpu_baseline'] = cpu_baseline(weights, scene, args.cpu_sample_rays)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
