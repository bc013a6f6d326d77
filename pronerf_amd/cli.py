"""``python -m pronerf_amd.cli {train-stage1,train-stage2,infer,eval,export-trt}`` — the sub-commands and options of the
reference's ``pronerf/cli.py`` (:170-219) on top of this package's three drivers.

Each sub-command builds the driver's argument list (``--config`` plus the mapped options plus whatever follows ``--``) and
calls the driver's ``train(argv)`` in-process; nothing is re-executed.  ``export-trt`` keeps its name but writes this build's
engine files (the packed weight streams, ``pnrf_mlp_serialize``) instead of ONNX / TensorRT plans; ``infer --use-trt`` /
``eval --use-trt`` then start from those files instead of packing the checkpoint.
"""
from __future__ import annotations

import argparse
import os
import sys

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _config(path):
    """A config path as given, or relative to the repository root when it does not exist under the working directory (the shipped
    configs/llff/fern/*.txt; the reference's `_repo_relative`, cli.py:22-26)."""
    if not os.path.exists(path) and not os.path.isabs(path) and os.path.exists(os.path.join(REPO_ROOT, path)):
        return os.path.join(REPO_ROOT, path)
    return path


def _extra(ns):
    extra = list(ns.extra)
    return extra[1:] if extra and extra[0] == '--' else extra


def stage1_argv(ns):
    argv = ['--config', _config(ns.config)]
    if ns.no_reload:
        argv.append('--no_reload')
    if ns.max_steps is not None:
        argv += ['--max_steps', str(ns.max_steps)]
    return argv + _extra(ns)


def stage2_argv(ns):
    argv = ['--config', _config(ns.config)]
    if ns.pretrain_path is not None:
        argv += ['--pretrain_path', ns.pretrain_path]
    if ns.no_reload:
        argv.append('--no_reload')
    if ns.max_steps is not None:
        argv += ['--max_steps', str(ns.max_steps)]
    return argv + _extra(ns)


def infer_argv(ns):
    argv = ['--config', _config(ns.config)]
    if ns.checkpoint is not None:
        argv += ['--ft_path', ns.checkpoint]
    if getattr(ns, 'render_test', False):
        argv.append('--render_test')
    if ns.use_trt:
        argv.append('--use_trt')
    if ns.max_images is not None:
        argv += ['--max_images', str(ns.max_images)]
    return argv + _extra(ns)


def _train_stage1(ns):
    from . import run_S_eS_eN_alter_base as m
    return m.train(stage1_argv(ns))


def _train_stage2(ns):
    from . import run_S_eS_eN_alter_base_refine2 as m
    return m.train(stage2_argv(ns))


def _infer(ns):
    from . import run_S_eS_eN_alter_trt as m
    return m.train(infer_argv(ns))


def _eval(ns):
    ns.render_test = True
    return _infer(ns)


def export_argv(ns):
    argv = ['--config', _config(ns.config), '--export_only']
    if ns.checkpoint is not None:
        argv += ['--ft_path', ns.checkpoint]
    return argv + _extra(ns)


def _export_trt(ns):
    """cli.py:105-157 writes nerf/minmaxrays_net/refine_net ONNX files and FP16 TensorRT engines into <basedir>/<expname>; here
    the driver's ``--export_only`` writes the three engine files of this build (the packed weight stream) there.  ``--onnx-only``,
    ``--height`` and ``--width`` are accepted for command-line compatibility: there is no ONNX stage, and the engines are not
    specialised to a batch size."""
    from . import run_S_eS_eN_alter_trt as m
    kw = m.train(export_argv(ns))
    print('Engine files written to:', ', '.join(sorted(kw['engine_paths'].values())))
    return kw


def build_parser():
    p = argparse.ArgumentParser(prog='python -m pronerf_amd.cli', description='ProNeRF LLFF pipeline on the MI355X HIP path')
    sub = p.add_subparsers(dest='command', required=True)

    def passthrough(q):
        q.add_argument('extra', nargs=argparse.REMAINDER, help='additional arguments for the underlying driver; prefix with --')

    q = sub.add_parser('train-stage1', help='alternating sampler / NeRF training')
    q.add_argument('--config', default='configs/llff/fern/fern_epi.txt')
    q.add_argument('--no-reload', action='store_true', dest='no_reload')
    q.add_argument('--max-steps', type=int, default=None, dest='max_steps')
    passthrough(q); q.set_defaults(func=_train_stage1)
    q = sub.add_parser('train-stage2', help='refinement training from a stage-1 checkpoint')
    q.add_argument('--config', default='configs/llff/fern/fern_refine.txt')
    q.add_argument('--pretrain-path', default=None, dest='pretrain_path')
    q.add_argument('--no-reload', action='store_true', dest='no_reload')
    q.add_argument('--max-steps', type=int, default=None, dest='max_steps')
    passthrough(q); q.set_defaults(func=_train_stage2)
    q = sub.add_parser('infer', help='render held-out / path views')
    q.add_argument('--config', default='configs/llff/fern/fern_trt.txt')
    q.add_argument('--checkpoint', default=None)
    q.add_argument('--render-test', action='store_true', dest='render_test')
    q.add_argument('--use-trt', action='store_true', dest='use_trt')
    q.add_argument('--max-images', type=int, default=None, dest='max_images')
    passthrough(q); q.set_defaults(func=_infer)
    q = sub.add_parser('eval', help='render the test split through the inference path')
    q.add_argument('--config', default='configs/llff/fern/fern_trt.txt')
    q.add_argument('--checkpoint', default=None)
    q.add_argument('--use-trt', action='store_true', dest='use_trt')
    q.add_argument('--max-images', type=int, default=None, dest='max_images')
    passthrough(q); q.set_defaults(func=_eval)
    q = sub.add_parser('export-trt', help='write the engine files (packed weight streams) of a checkpoint')
    q.add_argument('--config', default='configs/llff/fern/fern_trt.txt')
    q.add_argument('--checkpoint', default=None)
    q.add_argument('--onnx-only', action='store_true', dest='onnx_only')
    q.add_argument('--height', type=int, default=756)
    q.add_argument('--width', type=int, default=1008)
    passthrough(q); q.set_defaults(func=_export_trt)
    return p


def main(argv=None):
    ns = build_parser().parse_args(argv)
    return ns.func(ns)


if __name__ == '__main__':
    main(sys.argv[1:])
