#!/usr/bin/env python3
"""CPU study behind the two-pass sampler (sampler): emulate pass 1 (fp16-rounded weights and activations, fp32 accumulate, layer 0
kept fp32-grade) on rays of the Fern-geometry frame, compare its depths with the fp32 oracle, and check the per-ray error model that decides
which rays are re-done by the split-fp16 kernel:

    rounding an operand to fp16 (RN, 11 significant bits) is a zero-mean error of variance <= c x^2, c = 2^-22 / 3;
    V_{l+1} = sum_i var(dx_{l+1,i}) <= C_l (2 c S_l + V_l),  S_l = |x_l|^2,  C_l = max_j sum_i W_l[i,j]^2   (ELU' <= 1)
    var(dlogit_k) <= M_k (2 c S_L + V_L),  M_k = max_j w_k[j]^2
    depth error std  s_k = span * d_k (1 - d_k) * sqrt(var(dlogit_k))

A ray is flagged when some adjacent sorted gap  g_i < KAPPA * (s_i + s_{i+1}).  Reported: flagged fraction, rays whose order pass 1 gets
wrong, and how many of those are NOT flagged (must be 0); max |error| / s over all samples (how conservative the model is).

    python tools/sampler_twopass_model.py [--rays 65536] [--kappa 6]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pronerf_oracle as orc          # noqa: E402
from oracle import synth                          # noqa: E402

H, W, FOCAL = 756, 1008, 815.13
C_RND = 2.0 ** -22 / 3.0


def f16(x):
    return x.to(torch.float16).to(torch.float64)


def elu(x):
    return torch.where(x > 0, x, torch.expm1(x))


def pass1(w, o, d):
    """-> logits [n,27] (float64 arithmetic on fp16-rounded operands), per-layer |x_l|^2 [n, L]."""
    Ws = [torch.as_tensor(x, dtype=torch.float64) for x in w['W']]
    bs = [torch.as_tensor(x, dtype=torch.float64) for x in w['b']]
    pl = orc.pluecker(o, d).double()                       # [n,6]: unit direction, moment (independent of t)
    Wf = Ws[0].reshape(256, 48, 6).sum(1)                  # folded first layer (pnrf_pack.hip)
    x = elu(pl @ Wf.T + bs[0])                             # layer 0 fp32-grade
    S = []
    for Wl, bl in zip(Ws[1:-1], bs[1:-1]):
        xh = f16(x)
        S.append((xh ** 2).sum(1))
        x = elu(xh @ f16(Wl).T + bl)
    xh = f16(x)
    S.append((xh ** 2).sum(1))
    return xh @ f16(Ws[-1]).T + bs[-1], torch.stack(S, 1)


def model_std(w, S, norm='max'):
    """per-ray std bound of the 8 depth logits from the layer norms S [n, 6] (x_1 .. x_6).  norm='max': C_l = the largest column norm (strict under
    independence: what the kernel uses); 'mean': C_l = |W_l|_F^2 / 256, the amplification of an error spread evenly over the inputs (an estimate, not a bound)."""
    Ws = [np.asarray(x, dtype=np.float64) for x in w['W']]
    V = torch.zeros(S.shape[0], dtype=torch.float64)
    for l, Wl in enumerate(Ws[1:-1]):
        C = float((Wl ** 2).sum(0).max()) if norm == 'max' else float((Wl ** 2).sum() / Wl.shape[1])
        V = C * (2 * C_RND * S[:, l] + V)
    M = torch.as_tensor((Ws[-1][:8] ** 2).max(1))           # [8]
    return torch.sqrt(M[None, :] * (2 * C_RND * S[:, -1] + V)[:, None])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=65536)
    ap.add_argument('--kappa', type=float, nargs='*', default=[4.0, 6.0, 8.0])
    ap.add_argument('--norm', default='max', choices=('max', 'mean'))
    ap.add_argument('--all-sets', action='store_true', help='also the heavy-tailed, x4 and optimizer-trained sets of tests/test_fullframe_gpu.py')
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    sets = [(0, 'trained'), (3, 'trained'), (2, 'spread'), (1, 'default')] + ([(0, 'heavy'), (0, 'x4'), (0, 'optimizer')] if args.all_sets else [])
    for seed, kind in sets:
        scene = synth.make_scene(seed, H=H, W=W, focal=FOCAL, rotate=True)
        w = synth.weight_set(seed, kind)['sampler']
        Wn = [np.asarray(x, dtype=np.float64) for x in w['W']][1:-1]
        print(f'({seed},{kind}) column norms^2 per hidden layer, max / mean: ' + ', '.join(f'{(x ** 2).sum(0).max():.2f} / {(x ** 2).sum() / x.shape[1]:.2f}' for x in Wn))
        ro, rd = orc.get_rays(H, W, scene['K'], scene['c2w'])
        o, d = orc.ndc_rays(H, W, float(scene['K'][0, 0]), 1.0, ro, rd)
        sel = torch.linspace(0, H * W - 1, args.rays).long()
        o, d = o.reshape(-1, 3)[sel], d.reshape(-1, 3)[sel]
        with torch.no_grad():
            _, _, _, depth = orc.sampler_forward(w, orc.mm_input_from_rays(o, d))      # fp32 oracle
            y1, S = pass1(w, o, d)
        d1 = torch.sigmoid(y1[:, :8])
        err = (d1 - depth.double()).abs()
        s = d1 * (1 - d1) * model_std(w, S, args.norm)                                            # span = far - near = 1
        ds, idx = torch.sort(depth, dim=1, stable=True)
        d1s, idx1 = torch.sort(d1.float(), dim=1, stable=True)
        s_sorted = torch.gather(s, 1, idx1)
        gap1 = (d1s[:, 1:] - d1s[:, :-1]).double()
        flipped = (idx1 != idx).any(1)
        tie = (ds[:, 1:] - ds[:, :-1]).min(1)[0] <= 1e-6
        ratio = (err / s.clamp_min(1e-30))
        print(f'({seed},{kind}) {args.rays} rays: max depth err {float(err.max()):.2e}, rms {float((err**2).mean().sqrt()):.2e}; model std median {float(s.median()):.2e} '
              f'max {float(s.max()):.2e}; max err/std {float(ratio.max()):.2f}, 99.9% {float(ratio.flatten().kthvalue(int(0.999 * ratio.numel()))[0]):.2f}; '
              f'flipped by pass 1: {int(flipped.sum())} ({float(flipped.float().mean()):.2%}), oracle tie set {int(tie.sum())}')
        for kappa in args.kappa:
            flag = (gap1 < kappa * (s_sorted[:, 1:] + s_sorted[:, :-1])).any(1)
            missed = flipped & ~flag & ~tie
            print(f'    kappa {kappa:g}: flagged {float(flag.float().mean()):.2%}, flipped and not flagged (outside the tie set): {int(missed.sum())}')
        # global threshold for comparison: gap < mult x max observed error
        for mult in (4.0, 8.0):
            flag = (gap1 < mult * float(err.max())).any(1)
            print(f'    global {mult:g} x max err: flagged {float(flag.float().mean()):.2%}, missed {int((flipped & ~flag & ~tie).sum())}')


if __name__ == '__main__':
    main()
