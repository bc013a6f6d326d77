"""TEST INFRASTRUCTURE ONLY — the seeded synthetic inputs used by the oracle-side tooling.

The generator itself lives in ``pronerf_amd/synthetic.py`` (pure numpy) because ``bench.py`` and
``smoke()`` need the same inputs and may not import ``oracle/`` outside their checker legs.
"""
from pronerf_amd.synthetic import *  # noqa: F401,F403
from pronerf_amd.synthetic import (DIR_CH, MMNETDEPTH, MMNETWIDTH, MULTIRES, MULTIRES_VIEWS, N_POINT_RAY_ENC, N_SAMPLES,  # noqa: F401
                                   NETDEPTH, NETWIDTH, NUM_NEIGHBOR, POS_CH, make_nerfcls_weights, make_scene, make_weights,
                                   nerf_layer_dims, nerfcls_layer_dims, nerfcls_state_dict, state_dicts, load_trained_fixture, weight_set, scene_for, scene3d_frame)
