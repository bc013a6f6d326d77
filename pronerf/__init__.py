"""``pronerf`` — the reference's release-facing package name (KAIST-VICLab/pronerf ``pronerf/__init__.py``), kept so that
``python -m pronerf.cli {train-stage1,train-stage2,infer,eval,export-trt}`` works unchanged.  Everything forwards to
``pronerf_amd`` (the MI355X HIP path); there is no code of its own here."""

__all__ = ['__version__']

__version__ = '0.1.0'
