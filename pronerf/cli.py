"""``python -m pronerf.cli`` — the reference's entry point name (pronerf/cli.py:170-230) on top of ``pronerf_amd.cli``: same
sub-commands and options, dispatched to this build's drivers in-process."""
import sys

from pronerf_amd.cli import build_parser as _build_parser
from pronerf_amd.cli import (export_argv, infer_argv, stage1_argv, stage2_argv)  # noqa: F401  (argv mapping, re-exported)


def build_parser():
    p = _build_parser()
    p.prog = 'python -m pronerf.cli'
    return p


def main(argv=None):
    ns = build_parser().parse_args(argv)
    return ns.func(ns)


if __name__ == '__main__':
    main(sys.argv[1:])
