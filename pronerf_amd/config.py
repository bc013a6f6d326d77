"""Command-line / config-file surface of the three driver scripts.

The reference builds its parsers with ``configargparse`` (run_S_eS_eN_alter_trt.py:44-183, ..._base_refine2.py:27-160,
..._base.py:31-164): every option can come from ``--config <file>`` (``key = value`` lines) and be overridden on the command
line.  ``configargparse`` is not a dependency here: ``config_parser(variant)`` returns a plain ``argparse`` parser whose
``parse_args`` first reads the config file and installs its entries as defaults.  Option names, types and defaults are the
reference's (so its ``configs/llff/fern/*.txt`` files parse unchanged); options that only steer TensorRT / ONNX are accepted
and ignored by the drivers.
"""
from __future__ import annotations

import argparse
import ast

# name, type ('flag' = store_true, 'ints' = list of int), default — shared by the three scripts
_COMMON = [
    ('expname', str, None), ('datadir', str, './data/llff/fern'),
    ('netdepth', int, 8), ('netwidth', int, 256), ('netskips', 'ints', [4]),
    ('a_mmrgb', float, 0), ('a_p', float, 0), ('a_mmdisp', float, 0),
    ('mmnetdepth', int, 8), ('mmnetwidth', int, 256), ('mmnetskips', 'ints', [4]),
    ('netdepth_fine', int, 8), ('netwidth_fine', int, 256),
    ('N_rand', int, 32 * 32 * 4), ('lrate', float, 5e-4), ('weight_decay', float, 0.0), ('lrate_decay', int, 250),
    ('chunk', int, 1024 * 32), ('netchunk', int, 1024 * 64),
    ('no_batching', 'flag', False), ('full_image', 'flag', False), ('no_reload', 'flag', False), ('ft_path', str, None),
    ('num_neighbor', int, 4), ('N_samples', int, 64), ('N_importance', int, 0), ('N_point_ray_enc', int, 32), ('k_ref', int, 4),
    ('rand_crop_size', int, 100), ('mm_emb', 'flag', False), ('perturb', float, 1.), ('use_viewdirs', 'flag', False),
    ('i_embed', int, 0), ('multires', int, 10), ('multires_views', int, 4), ('raw_noise_std', float, 0.),
    ('render_only', 'flag', False), ('render_test', 'flag', False), ('render_factor', int, 0),
    ('precrop_iters', int, 0), ('precrop_frac', float, .5),
    ('dataset_type', str, 'llff'), ('white_bkgd', 'flag', False), ('factor', int, 8), ('no_ndc', 'flag', False),
    ('lindisp', 'flag', False), ('spherify', 'flag', False), ('llffhold', int, 8),
    ('i_print', int, 5000), ('i_img', int, 10000), ('i_weights', int, 10000), ('i_testset', int, 10000), ('i_video', int, 10000),
]
_VARIANT = {
    # inference (run_S_eS_eN_alter_trt.py): TensorRT / ONNX switches are parsed for config compatibility only
    'trt': [('basedir', str, './logs_trt/'), ('use_trt', 'flag', False), ('export_only', 'flag', False), ('nerf_engine_path', str, None),
            ('mm_engine_path', str, None), ('refine_engine_path', str, None), ('max_images', int, None),
            # not in the reference: the renderer's operating point (pronerf_amd.render.PRESETS) — 'default', 'quality' (exact sampler + fp16 NeRF
            # operands) or 'auto' (decided from the first rendered frame: the exact single-pass sampler when the two-pass form would re-render most rays)
            ('pnrf_preset', str, 'default')],
    # stage 2 (run_S_eS_eN_alter_base_refine2.py)
    'refine2': [('basedir', str, './logs_epi_RR/'), ('pretrain_path', str, None), ('test_frames', 'ints', [3, 11]), ('max_steps', int, None)],
    # stage 1 (run_S_eS_eN_alter_base.py)
    'base': [('basedir', str, './logs_epi_RR/'), ('epi_nerf', 'flag', False), ('test_frames', 'ints', [3, 11]), ('max_steps', int, None)],
}


def _convert(kind, text):
    text = text.strip()
    if kind == 'flag':
        return text.lower() in ('true', '1', 'yes')
    if kind == 'ints':
        v = ast.literal_eval(text) if text.startswith('[') else [int(t) for t in text.replace(',', ' ').split()]
        return [int(x) for x in v]
    if text == 'None':
        return None
    return kind(text)


def read_config_file(path):
    """``key = value`` lines ('#' / ';' comments, blank lines ignored) -> dict of raw strings."""
    out = {}
    with open(path) as f:
        for line in f:
            line = line.split('#', 1)[0].split(';', 1)[0].strip()
            if not line:
                continue
            if '=' not in line:
                out[line] = 'True'
                continue
            k, v = line.split('=', 1)
            out[k.strip().lstrip('-')] = v.strip()
    return out


class ConfigArgumentParser(argparse.ArgumentParser):
    """argparse with ``--config``: file entries become defaults, explicit command-line values win."""

    def __init__(self, table, **kw):
        super().__init__(**kw)
        self._kinds = {}
        self.add_argument('--config', type=str, default=None, help='config file path (key = value lines)')
        for name, kind, default in table:
            self._kinds[name] = kind
            if kind == 'flag':
                self.add_argument('--' + name, action='store_true', default=default)
            elif kind == 'ints':
                self.add_argument('--' + name, type=int, nargs='*', default=default)
            else:
                self.add_argument('--' + name, type=kind, default=default)

    def parse_known_args(self, args=None, namespace=None):
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument('--config', type=str, default=None)
        known, _ = pre.parse_known_args(args)
        if known.config:
            cfg = read_config_file(known.config)
            unknown = sorted(set(cfg) - set(self._kinds))
            if unknown:
                self.error(f'{known.config}: unknown option(s) {unknown}')
            self.set_defaults(**{k: _convert(self._kinds[k], v) for k, v in cfg.items()})
        return super().parse_known_args(args, namespace)


def config_parser(variant='trt'):
    return ConfigArgumentParser(_COMMON + _VARIANT[variant], description=f'ProNeRF {variant} driver (MI355X HIP path)')
