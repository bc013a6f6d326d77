"""GPU: engine files — the packed weight stream serialized with pnrf_mlp_serialize, this build's counterpart of the reference's
serialized TensorRT engines (pronerf/cli.py:105-157 export, run_S_eS_eN_alter_trt.py:490-499 load).  An engine must reproduce the
network it was written from bit for bit, and a damaged or foreign file must be refused."""
import os
import struct

import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from pronerf_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def _cls_stack(wc):
    """NeRF-class weights in pack order: pts 0..7, feature, alpha, views, rgb."""
    order = list(wc['pts_linears']) + [wc['feature_linear'], wc['alpha_linear'], wc['views_linears'][0], wc['rgb_linear']]
    return {'W': [w for w, _ in order], 'b': [b for _, b in order]}


def _packed(kind):
    from pronerf_amd import ops
    if kind == 'nerfcls':
        w = _cls_stack(synth.make_nerfcls_weights(3))
        return ops.PackedMLP(ops.NET_NERFCLS, w['W'], w['b']), 63, 27
    w = synth.make_weights(3, 'trained')[kind]
    net = {'sampler': ops.NET_SAMPLER, 'refine': ops.NET_REFINE, 'nerf': ops.NET_NERF}[kind]
    return ops.PackedMLP(net, w['W'], w['b']), {'sampler': 288, 'refine': 144, 'nerf': 63}[kind], 27 if kind == 'nerf' else 0


@pytest.mark.parametrize('kind', ['sampler', 'refine', 'nerf', 'nerfcls'])
def test_engine_round_trip_is_bit_exact(dev, kind):
    from pronerf_amd import ops
    with torch.cuda.device(dev):
        a, n_in, n_v = _packed(kind)
        blob = a.serialize()
        assert blob[:8] == b'PNRFENG\0' and len(blob) > 128 + 16384
        b = ops.PackedMLP.deserialize(blob, expect_net=a.net)
        assert (b.net, b.in_dim, b.out_dim) == (a.net, a.in_dim, a.out_dim)
        assert b.serialize() == blob                                    # every section restored
        g = torch.Generator().manual_seed(0)
        x = (torch.rand(1000, n_in, generator=g) * 2 - 1).to(dev)
        v = (torch.rand(1000, n_v, generator=g) * 2 - 1).to(dev) if n_v else None
        for head in ((False, True) if kind in ('sampler', 'refine') else (False,)):
            assert torch.equal(a.forward(x, v, head_act=head), b.forward(x, v, head_act=head))
        with pytest.raises(ops.PnrfError, match='expected'):
            ops.PackedMLP.deserialize(blob, expect_net=(a.net + 1) % 4)


def test_damaged_and_foreign_engines_are_refused(dev):
    from pronerf_amd import ops
    with torch.cuda.device(dev):
        a, _, _ = _packed('refine')
        blob = bytearray(a.serialize())
        bad = bytearray(blob); bad[5000] ^= 0x10
        with pytest.raises(ops.PnrfError, match='checksum'):
            ops.PackedMLP.deserialize(bytes(bad))
        with pytest.raises(ops.PnrfError, match='truncated'):
            ops.PackedMLP.deserialize(bytes(blob[:-4]))
        with pytest.raises(ops.PnrfError, match='truncated or padded'):
            ops.PackedMLP.deserialize(bytes(blob) + b'\0' * 8)
        other = bytearray(blob); other[16:20] = struct.pack('<I', struct.unpack('<I', blob[16:20])[0] ^ 1)      # layout tag of another build
        with pytest.raises(ops.PnrfError, match='another build'):
            ops.PackedMLP.deserialize(bytes(other))
        ops.PackedMLP.deserialize(bytes(blob))                           # the untouched image still loads


def test_renderer_from_engine_directory(dev, tmp_path):
    from pronerf_amd import synthetic
    from pronerf_amd.render import Renderer
    scene = synthetic.make_scene(0, H=48, W=64, rotate=True)
    for w in (synthetic.make_weights(0, 'trained'), dict(synthetic.make_weights(0, 'trained'), nerf=_cls_stack(synth.make_nerfcls_weights(1, head_scale=0.3)))):
        a = Renderer(w, max_rays=48 * 64, device=dev)
        paths = a.save_engines(str(tmp_path / 'eng'))
        assert sorted(os.path.basename(p) for p in paths.values()) == ['minmaxrays_net.pnrf', 'nerf.pnrf', 'refine_net.pnrf']
        b = Renderer.from_engines(str(tmp_path / 'eng'), max_rays=48 * 64, device=dev)
        outs = []
        for r in (a, b):
            r.set_views(scene['c2w'], scene['poses'], scene['images'], scene['K'])
            rays, orr = r.frame_rays(scene['K'], scene['c2w'], 48, 64)
            outs.append(r.render_rays(rays, orr, want_idx=True))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('fine', ['donerf', 'nerfcls'])
def test_export_then_infer_from_engines(dev, tmp_path, fine):
    """`export-trt` + `infer --use-trt` of the CLI: engines written from a checkpoint, then a run that never sees the checkpoint
    renders the same PNGs as the checkpoint run."""
    import llff_synth
    from pronerf_amd import cli
    from pronerf_amd import run_nerf_helpers as h
    root = llff_synth.make_dataset(str(tmp_path / 'scene'), seed=1, n=10, H=24, W=32, factor=4)
    sds = synth.state_dicts(synth.make_weights(0, 'trained'))
    fine_sd = sds['nerf'] if fine == 'donerf' else synth.nerfcls_state_dict(synth.make_nerfcls_weights(0))
    ck = str(tmp_path / '000123.tar')
    torch.save({'global_step': 123, 'mmr_network_fn_state_dict': sds['sampler'], 'refine_net_state_dict': sds['refine'], 'network_fine_state_dict': fine_sd}, ck)
    body = (f'basedir = {tmp_path}/logs\ndatadir = {root}\nfactor = 4\nllffhold = 8\nN_samples = 8\nN_point_ray_enc = 48\nmmnetdepth = 6\n'
            'mmnetskips = [10000]\nnum_neighbor = 4\nuse_viewdirs = True\n')
    (tmp_path / 'ck.txt').write_text('expname = from_ckpt\n' + body)
    (tmp_path / 'eng.txt').write_text('expname = from_engine\n' + body)
    kw0 = cli.main(['infer', '--config', str(tmp_path / 'ck.txt'), '--checkpoint', ck, '--render-test'])
    kwx = cli.main(['export-trt', '--config', str(tmp_path / 'eng.txt'), '--checkpoint', ck, '--onnx-only'])
    eng_dir = tmp_path / 'logs' / 'from_engine'
    assert sorted(os.listdir(eng_dir)) == ['minmaxrays_net.pnrf', 'nerf.pnrf', 'refine_net.pnrf']       # nothing rendered on export
    assert sorted(kwx['engine_paths'].values()) == sorted(str(eng_dir / f) for f in os.listdir(eng_dir))
    kw1 = cli.main(['infer', '--config', str(tmp_path / 'eng.txt'), '--use-trt', '--render-test'])
    assert kw1['use_trt'] and isinstance(kw1['network_fine'], h.NeRF if fine == 'nerfcls' else h.DoNeRFTRT)
    assert all(m.engine_path for m in (kw1['network_fine'], kw1['min_max_ray_net'], kw1['refine_net']))
    d0 = tmp_path / 'logs' / 'from_ckpt' / 'renderonly_test_000123'
    d1 = eng_dir / 'renderonly_test_000000'                               # no checkpoint -> global step 0
    assert sorted(os.listdir(d0)) == sorted(os.listdir(d1)) == ['000.png', '001.png', 'depth_000.png', 'depth_001.png']
    for f in os.listdir(d0):
        assert (d0 / f).read_bytes() == (d1 / f).read_bytes(), f
    assert kw0['psnrs'] == kw1['psnrs']
    # per-net override + a module asked for use_trt without an engine
    from pronerf_amd import run_S_eS_eN_alter_trt as trt
    from pronerf_amd.ops import PnrfError
    with pytest.raises(PnrfError, match='expected'):
        trt.train(['--config', str(tmp_path / 'eng.txt'), '--use_trt', '--render_test', '--mm_engine_path', str(eng_dir / 'refine_net.pnrf')], device=dev)
    with pytest.raises(FileNotFoundError):
        trt.train(['--config', str(tmp_path / 'ck.txt'), '--use_trt', '--render_test'], device=dev)
