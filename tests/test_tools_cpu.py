"""The two measurement aids that are tied to product sources must keep building: tools/hgemm_probe.hip includes the trainer's product kernels
(pnrf_hgemm.h, with PNRF_HG_PROBE defined), tools/mfma_ceiling.hip is what bench.py runs for roofline.sustained (pronerf_amd.build compiles
it).  hipcc cross-compiles for gfx950 without a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = next((c for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')) if c and os.path.exists(c)), None)


@pytest.mark.skipif(HIPCC is None, reason='hipcc not found')
@pytest.mark.parametrize('src,extra', [('tools/hgemm_probe.hip', ['-I', os.path.join(ROOT, 'pronerf_amd', 'csrc'), '-std=c++17']),
                                       ('tools/mfma_ceiling.hip', []),
                                       ('tools/foreign_kernels.hip', []),
                                       ('tools/pkf32_coexec_probe.hip', ['-DVICTIM_OVFL', '-DBIG_VGPR']),      # the packed-fp32 fault's micro-reproducer
                                       ('tools/valu_rate_probe.hip', []), ('tools/mfma_valu_overlap_probe.hip', []), ('tools/elu_chain_probe.hip', [])])
def test_probe_compiles(tmp_path, src, extra):
    out = str(tmp_path / 'probe.o')
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O1', '-c', os.path.join(ROOT, src), '-o', out] + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert os.path.getsize(out) > 0


def test_build_module_knows_the_ceiling_probe():
    from pronerf_amd import build as b
    assert os.path.exists(b.CEILING_SRC) and b.CEILING_BIN.startswith(b.LIBDIR)
    assert os.path.exists(b.FOREIGN_SRC) and b.FOREIGN_LIB.startswith(b.LIBDIR)        # the co-residency stress test's stand-in kernels
