"""CPU: the error model behind the two-pass sampler, on the oracle.  Pass 1 is emulated in float64 arithmetic on fp16-rounded
weights and activations (tools/sampler_twopass_model.py); checked here on 4096 rays of the Fern-geometry frame per weight set:
  * the per-ray bound s_k really bounds the emulated error: max |error| / s stays below 2 = the shipped kappa (measured 0.8 .. 1.9 over 65 536 rays,
    i.e. s >= 2.7 sigma; over these 4096 rays 0.6 .. 1.6) — a single error never reaches the threshold two errors together would have to cross;
  * no ray whose order pass 1 gets wrong escapes the flag at the shipped kappa = 2, nor at kappa = 1 (outside the fp32 tie set);
  * the flagged fraction stays a minority.
Weight sets: the seeded draws and the optimizer-trained fixture.  The GPU side of the same statement (all 762 048 rays, 7 weight sets, the shipped
kappa and kappa = 0): tests/test_fullframe_gpu.py; tools/kappa_scan.py for the scan over kappa."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from oracle import pronerf_oracle as orc   # noqa: E402
from oracle import synth                   # noqa: E402

H, W, FOCAL = 756, 1008, 815.13


KAPPA = 2.0          # include/pronerf_hip.h PNRF_SAMPLER_KAPPA


def test_kappa_is_the_header_default():
    hdr = open(os.path.join(ROOT, 'include', 'pronerf_hip.h')).read()
    assert f'#define PNRF_SAMPLER_KAPPA {KAPPA:.1f}f' in hdr


@pytest.mark.parametrize('seed,kind', [(0, 'trained'), (1, 'default'), (2, 'spread'), (0, 'optimizer'), (0, 'x4')])
def test_error_bound_and_flag(seed, kind):
    import sampler_twopass_model as M
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    scene = synth.make_scene(seed, H=H, W=W, focal=FOCAL, rotate=True)
    w = synth.weight_set(seed, kind)['sampler']
    ro, rd = orc.get_rays(H, W, scene['K'], scene['c2w'])
    o, d = orc.ndc_rays(H, W, float(scene['K'][0, 0]), 1.0, ro, rd)
    sel = torch.linspace(0, H * W - 1, 4096).long()
    o, d = o.reshape(-1, 3)[sel], d.reshape(-1, 3)[sel]
    with torch.no_grad():
        _, _, _, depth = orc.sampler_forward(w, orc.mm_input_from_rays(o, d))
        y1, S = M.pass1(w, o, d)
    d1 = torch.sigmoid(y1[:, :8])
    err = (d1 - depth.double()).abs()
    s = d1 * (1 - d1) * M.model_std(w, S)
    assert float((err / s.clamp_min(1e-30)).max()) < KAPPA
    ds, idx = torch.sort(depth, dim=1, stable=True)
    d1s, idx1 = torch.sort(d1.float(), dim=1, stable=True)
    ss = torch.gather(s, 1, idx1)
    flipped = (idx1 != idx).any(1)
    tie = (ds[:, 1:] - ds[:, :-1]).min(1)[0] <= 1e-6
    for kappa in (KAPPA, 1.0):
        flag = ((d1s[:, 1:] - d1s[:, :-1]).double() < kappa * (ss[:, 1:] + ss[:, :-1]) + 2e-6).any(1)
        assert int((flipped & ~flag & ~tie).sum()) == 0, kappa
    flag = ((d1s[:, 1:] - d1s[:, :-1]).double() < KAPPA * (ss[:, 1:] + ss[:, :-1]) + 2e-6).any(1)
    assert float(flag.float().mean()) < 0.2
