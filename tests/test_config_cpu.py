"""Config-file / CLI surface (pronerf_amd.config): the reference's own Fern configs must parse, command line wins."""
import glob
import os

import pytest

from pronerf_amd.config import config_parser, read_config_file

FERN_TRT = """expname = fern_8samples_trtinfer
ft_path = logs_minmax/fern_refine_8samples/370000.tar
datadir = data/nerf_llff_data/fern
dataset_type = llff
factor = 4
llffhold = 8
N_rand = 4096
N_samples = 8
use_viewdirs = True
raw_noise_std = 1e0
lrate = 5e-4
mmnetdepth = 6
mmnetskips = [10000]
N_point_ray_enc = 48
mm_emb = False
weight_decay = 5e-8
num_neighbor = 4
use_trt = False
"""


def test_config_file_then_command_line(tmp_path):
    p = tmp_path / 'fern_trt.txt'
    p.write_text(FERN_TRT)
    a = config_parser('trt').parse_args(['--config', str(p)])
    assert a.expname == 'fern_8samples_trtinfer' and a.factor == 4 and a.N_samples == 8 and a.N_point_ray_enc == 48
    assert a.mmnetskips == [10000] and a.use_viewdirs is True and a.mm_emb is False and a.use_trt is False
    assert a.raw_noise_std == 1.0 and a.weight_decay == 5e-8 and a.netdepth == 8 and a.basedir == './logs_trt/'
    b = config_parser('trt').parse_args(['--config', str(p), '--factor', '8', '--render_test', '--max_images', '2'])
    assert b.factor == 8 and b.render_test is True and b.max_images == 2 and b.N_samples == 8


def test_unknown_option_in_file_is_an_error(tmp_path):
    p = tmp_path / 'bad.txt'
    p.write_text('no_such_option = 3\n')
    with pytest.raises(SystemExit):
        config_parser('trt').parse_args(['--config', str(p)])


@pytest.mark.parametrize('variant', ['trt', 'refine2', 'base'])
def test_defaults(variant):
    a = config_parser(variant).parse_args([])
    assert a.N_samples == 64 and a.N_point_ray_enc == 32 and a.lrate_decay == 250 and a.num_neighbor == 4
    assert a.basedir == ('./logs_trt/' if variant == 'trt' else './logs_epi_RR/')
    assert hasattr(a, 'max_images') == (variant == 'trt') and hasattr(a, 'pretrain_path') == (variant == 'refine2')


def test_reference_configs_parse_if_present():
    """In the development container the reference's shipped configs are parsed as they are (skipped on the GPU box)."""
    files = glob.glob('/root/reference/configs/llff/fern/*.txt')
    if not files:
        pytest.skip('reference not present')
    for f in files:
        variant = 'trt' if f.endswith('_trt.txt') else ('refine2' if f.endswith('_refine.txt') else 'base')
        a = config_parser(variant).parse_args(['--config', f])
        assert a.N_samples == 8 and set(read_config_file(f)) <= set(vars(a))


def test_cli_subcommands_map_to_driver_arguments():
    """pronerf_amd.cli: the reference CLI's sub-commands / options (pronerf/cli.py:170-219) -> the drivers' argv."""
    from pronerf_amd import cli
    p = cli.build_parser()
    ns = p.parse_args(['train-stage1', '--config', 'a.txt', '--max-steps', '5', '--', '--N_rand', '1024'])
    assert cli.stage1_argv(ns) == ['--config', 'a.txt', '--max_steps', '5', '--N_rand', '1024']
    ns = p.parse_args(['train-stage2', '--config', 'b.txt', '--pretrain-path', 'x.tar', '--no-reload'])
    assert cli.stage2_argv(ns) == ['--config', 'b.txt', '--pretrain_path', 'x.tar', '--no_reload']
    ns = p.parse_args(['infer', '--config', 'c.txt', '--checkpoint', 'y.tar', '--render-test', '--max-images', '2'])
    assert cli.infer_argv(ns) == ['--config', 'c.txt', '--ft_path', 'y.tar', '--render_test', '--max_images', '2']
    ns = p.parse_args(['eval', '--checkpoint', 'y.tar'])
    assert ns.func is cli._eval and ns.config.endswith('fern_trt.txt')
    ns = p.parse_args(['export-trt', '--config', 'c.txt', '--checkpoint', 'y.tar', '--onnx-only', '--height', '378', '--', '--expname', 'e'])
    assert ns.func is cli._export_trt and cli.export_argv(ns) == ['--config', 'c.txt', '--export_only', '--ft_path', 'y.tar', '--expname', 'e']
    ns = p.parse_args(['infer', '--use-trt'])
    assert cli.infer_argv(ns)[-1] == '--use_trt'
    # every driver accepts what the CLI hands it
    for variant, argv in (('base', ['--max_steps', '5', '--N_rand', '1024']), ('refine2', ['--pretrain_path', 'x.tar', '--no_reload']),
                          ('trt', ['--ft_path', 'y.tar', '--render_test', '--max_images', '2']),
                          ('trt', ['--export_only', '--use_trt', '--nerf_engine_path', 'n.pnrf', '--mm_engine_path', 'm.pnrf', '--refine_engine_path', 'r.pnrf'])):
        config_parser(variant).parse_args(argv)


def test_reference_entry_point_name(monkeypatch):
    """`python -m pronerf.cli` — the module path of the reference's CLI (pronerf/cli.py:180-230; BASELINE.json configs[0]:
    `infer --render-test --max-images 1`): same parser, dispatched to this build's inference driver with the mapped argv."""
    import subprocess
    import sys
    import pronerf
    import pronerf.cli as rcli
    from pronerf_amd import run_S_eS_eN_alter_trt as drv
    assert pronerf.__version__
    p = rcli.build_parser()
    assert p.prog == 'python -m pronerf.cli'
    ns = p.parse_args(['infer', '--render-test', '--max-images', '1'])
    assert rcli.infer_argv(ns)[0] == '--config' and rcli.infer_argv(ns)[1].endswith(os.path.join('configs', 'llff', 'fern', 'fern_trt.txt'))
    assert rcli.infer_argv(ns)[2:] == ['--render_test', '--max_images', '1']
    seen = {}
    monkeypatch.setattr(drv, 'train', lambda argv: seen.setdefault('argv', list(argv)))
    rcli.main(['infer', '--render-test', '--max-images', '1', '--', '--chunk', '1024'])
    assert seen['argv'][2:] == ['--render_test', '--max_images', '1', '--chunk', '1024']
    seen.clear()
    rcli.main(['eval', '--checkpoint', 'z.tar'])
    assert seen['argv'][2:] == ['--ft_path', 'z.tar', '--render_test']
    # as a module, from another working directory, with both spellings (`-m pronerf.cli`, `-m pronerf`)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    for mod in ('pronerf.cli', 'pronerf'):
        r = subprocess.run([sys.executable, '-m', mod, 'infer', '--help'], cwd='/', env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and '--render-test' in r.stdout and '--max-images' in r.stdout, r.stderr
    r = subprocess.run([sys.executable, '-m', 'pronerf.cli'], cwd='/', env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and 'train-stage1' in r.stderr and 'export-trt' in r.stderr


def test_shipped_configs_carry_the_reference_hyperparameters():
    """configs/llff/fern/*.txt (the CLI's default --config files): parse with this package's parser; spot values of the reference's
    files (fern_trt.txt:10-34, fern_refine.txt, fern_epi.txt); equal to the reference's own files where those are present."""
    from pronerf_amd import cli
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'llff', 'fern')
    want = {'fern_trt.txt': ('trt', dict(expname='fern_8samples_trtinfer', factor=4, llffhold=8, N_samples=8, N_point_ray_enc=48, num_neighbor=4,
                                         mmnetdepth=6, mmnetwidth=256, mmnetskips=[10000], use_viewdirs=True, use_trt=False, k_ref=1, weight_decay=5e-8)),
            'fern_refine.txt': ('refine2', dict(expname='fern_refine_8samples_v2', N_rand=4096, lrate=3e-4, a_mmrgb=0.0, k_ref=1, weight_decay=0.0,
                                                pretrain_path='logs_epi_RR/fern_sampler_e2e_donerf_8samples/500000.tar')),
            'fern_epi.txt': ('base', dict(expname='fern_sampler_e2e_donerf_8samples_cc', N_rand=4096, lrate=5e-4, a_mmrgb=1.0, a_mmdisp=1.0, k_ref=0,
                                          mmnetskips=[1000], raw_noise_std=1.0))}
    for f, (variant, vals) in want.items():
        a = config_parser(variant).parse_args(['--config', os.path.join(root, f)])
        for k, v in vals.items():
            assert getattr(a, k) == v, (f, k, getattr(a, k))
        ref = os.path.join('/root/reference/configs/llff/fern', f)
        if os.path.exists(ref):
            b = config_parser(variant).parse_args(['--config', ref])
            diff = {k for k in vars(a) if getattr(a, k) != getattr(b, k)} - {'config', 'nerf_engine_path', 'mm_engine_path', 'refine_engine_path'}
            assert not diff, (f, diff)
    # the CLI finds them from any working directory
    ns = cli.build_parser().parse_args(['infer'])
    cwd = os.getcwd()
    try:
        os.chdir('/')
        assert cli.infer_argv(ns)[1] == os.path.join(root, 'fern_trt.txt')
    finally:
        os.chdir(cwd)
