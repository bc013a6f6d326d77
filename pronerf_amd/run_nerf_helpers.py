"""Host-side mirror of the reference's ``run_nerf_helpers`` for the render hot path.

Same names, argument meaning and error behaviour as the live subset of the reference module
(run_nerf_helpers.py), every operator backed by a HIP kernel through the C ABI:

    get_embedder / Embedder        helpers:635-692   -> pnrf_posenc_fwd
    Pluecker                       helpers:613-632   -> pnrf_plucker_fwd
    MinMaxRay_Net                  helpers:1440-1471 -> pnrf_mlp_fwd (fp32 MFMA)
    MinMaxRaySamplerTRT_Net        helpers:1473-1507 -> pnrf_mlp_fwd (+ sigmoid heads)
    MinMaxRayEpiSamplerTRT_Net     helpers:1509-1540 -> pnrf_mlp_fwd (bf16 MFMA, sigmoid/tanh heads)
    DoNeRFTRT                      helpers:1186-1343 -> pnrf_mlp_fwd (bf16 MFMA)
    NeRF                           helpers:792-847   -> pnrf_mlp_fwd (bf16 MFMA, skip + view branch)
    get_rays / ndc_rays            helpers:2705-2714, 2776-2793 -> pnrf_frame_rays_fwd / pnrf_ndc_rays_fwd

The model classes keep the reference's ``state_dict`` keys (``fc_backbone.{i}.*``, ``fc_output.*``,
``layers.{i}.*``) so checkpoints interchange; their parameters are re-packed into the device weight
stream lazily whenever they change.  There is no eager-PyTorch fallback: on a machine without the
HIP library or a GPU these operators raise ``PnrfError``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .ops import PnrfError

# Misc (helpers:129-135) — frame-level metrics on finished images, not part of the per-ray path
img2mse = lambda x, y: torch.mean((x - y) ** 2)
mse2psnr = lambda x: -10. * torch.log10(x)
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)


# ------------------------------------------------------------------------------- encodings
class Embedder(nn.Module):
    """Positional encoding [x, sin(2^k x), cos(2^k x)] (helpers:635-671)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        d = kwargs['input_dims']
        n = kwargs['num_freqs']
        if not (kwargs.get('include_input', True) and kwargs.get('log_sampling', True) and d == 3
                and kwargs['max_freq_log2'] == n - 1):
            raise PnrfError('Embedder: the HIP kernel implements include_input=True, log_sampling=True, input_dims=3 '
                            '(the only configuration get_embedder() creates)')
        self.num_freqs = n
        self.freq_bands = 2. ** torch.linspace(0., kwargs['max_freq_log2'], steps=n) if n > 0 else torch.zeros(0)
        self.out_dim = d + 2 * d * n

    def embed(self, inputs):
        return ops.posenc(inputs, self.num_freqs)

    forward = embed


def get_embedder(multires, i=0):
    """(embed_fn, out_dim); ``i == -1`` -> (Identity, 3)  (helpers:677-692)."""
    if i == -1:
        return nn.Identity(), 3
    eo = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return (lambda x, eo=eo: eo.embed(x)), eo.out_dim


class Pluecker(nn.Module):
    """[normalize(d), o x normalize(d)] (helpers:613-632)."""

    def __init__(self, origin=None):
        super().__init__()
        self.in_channels = 6
        self.out_channels = 6
        self.direction_multiplier = 1.0
        self.moment_multiplier = 1.0
        self.origin = origin

    def forward(self, rays_o, rays_d):
        if rays_o.shape != rays_d.shape:
            rays_o, rays_d = torch.broadcast_tensors(rays_o, rays_d)
        return ops.plucker(rays_o.contiguous(), rays_d.contiguous())


# ------------------------------------------------------------------------------- networks
class _PackedNet(nn.Module):
    """nn.Linear parameters + a lazily refreshed packed copy for the MFMA kernels."""
    _NET = None
    engine_path = None          # set while the packed copy comes from an engine file (load_engine)

    def _linears(self):
        raise NotImplementedError

    def packed(self) -> ops.PackedMLP:
        lins = self._linears()
        key = self._param_key()
        if getattr(self, '_pack_key', None) != key:
            dev = lins[0].weight.device
            if dev.type != 'cuda':
                raise PnrfError(f'{type(self).__name__}: parameters live on {dev}; move the module to the GPU '
                                '(pronerf_amd has no CPU path)')
            with torch.cuda.device(dev):
                self._packed = ops.PackedMLP(self._NET, [l.weight for l in lins], [l.bias for l in lins])
            self._pack_key = key
            self.engine_path = None
        return self._packed

    def _param_key(self):
        return tuple((p.data_ptr(), p._version) for l in self._linears() for p in (l.weight, l.bias))

    def save_engine(self, path):
        """Write the packed weight stream of the current parameters to ``path`` — this build's counterpart of the reference's
        serialized TensorRT engine (pronerf/cli.py:131-156)."""
        self.packed().save(path)

    def load_engine(self, path):
        """Use the engine file at ``path`` instead of packing the parameters (the reference's ``NeRFEngine(path)`` etc.,
        run_S_eS_eN_alter_trt.py:497-499).  The nn.Linear parameters are left alone and no longer describe what runs; the engine
        stays in use until they are next modified (``load_state_dict``, an optimizer step), which repacks from them."""
        dev = self._linears()[0].weight.device
        if dev.type != 'cuda':
            raise PnrfError(f'{type(self).__name__}: move the module to the GPU before load_engine (pronerf_amd has no CPU path)')
        with torch.cuda.device(dev):
            self._packed = ops.PackedMLP.load(path, expect_net=self._NET)
        self._pack_key = self._param_key()
        self.engine_path = path

    def weights(self):
        lins = self._linears()
        return {'W': [l.weight.detach() for l in lins], 'b': [l.bias.detach() for l in lins]}


class MinMaxRay_Net(_PackedNet):
    """Sampler backbone, raw outputs (helpers:1440-1471).  Shapes the kernels take (round 6: the reference's free ``--mmnetdepth``,
    ``--N_point_ray_enc``, ``--num_neighbor``, run_S_eS_eN_alter_trt.py:62-82, 427-457): W = 256, no skip inside the stack, any depth D >= 2,
    sampler 6 * N_point_ray_enc -> 27 (any number of ray points), refine 48 + 24 * num_neighbor -> 35 (1 .. 8 neighbour views); N_samples = 8.
    The Fern configs are D = 6, 288 -> 27 and 144 -> 35."""

    def __init__(self, D=8, W=256, input_ch=3, output_ch=3, skips=[4]):
        super().__init__()
        self.D, self.W, self.input_ch, self.skips = D, W, input_ch, skips
        self.fc_backbone = nn.ModuleList([nn.Linear(input_ch, W)] +
                                         [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W) for i in range(D - 1)])
        self.fc_output = nn.Linear(W, output_ch)
        if output_ch == 27 and input_ch >= 6 and input_ch % 6 == 0:
            self._NET = ops.NET_SAMPLER
        elif output_ch == 35 and input_ch >= 72 and (input_ch - 48) % 24 == 0 and (input_ch - 48) // 24 <= 8:
            self._NET = ops.NET_REFINE
        self._supported = (2 <= D <= 32 and W == 256 and not any(0 <= s < D - 1 for s in skips) and self._NET is not None)

    def _linears(self):
        if not self._supported:
            raise PnrfError(f'{type(self).__name__}(D={self.D}, W={self.W}, input_ch={self.input_ch}, skips={self.skips}): '
                            'the HIP kernels take W=256, no skips inside the stack, 2 <= D <= 32, and 6*N_point_ray_enc -> 27 (sampler) or '
                            '48 + 24*num_neighbor -> 35 with 1 <= num_neighbor <= 8 (refine); N_samples = 8')
        return list(self.fc_backbone) + [self.fc_output]

    def forward(self, x):
        return self.packed().forward(x)


class MinMaxRaySamplerTRT_Net(MinMaxRay_Net):
    """-> (mm_rgb, mm_density_add, mm_density_mul, depth_values)  (helpers:1473-1507)."""

    def __init__(self, D=8, W=256, input_ch=3, output_ch=3, skips=[4], N_samples=8):
        super().__init__(D, W, input_ch, output_ch, skips)
        self.N_samples = N_samples

    def forward(self, x):
        S = self.N_samples
        y = self.packed().forward(x, head_act=True)
        return y[:, 3 * S:], y[:, S:2 * S], y[:, 2 * S:3 * S], y[:, :S]


class MinMaxRayEpiSamplerTRT_Net(MinMaxRay_Net):
    """-> (refine_depth_values, refine_rgb, points_offset)  (helpers:1509-1540)."""

    def __init__(self, D=8, W=256, input_ch=3, output_ch=3, skips=[4], N_samples=8):
        super().__init__(D, W, input_ch, output_ch, skips)
        self.N_samples = N_samples

    def forward(self, x):
        S = self.N_samples
        y = self.packed().forward(x, head_act=True)
        return y[:, :S], y[:, 4 * S:], y[:, S:4 * S]


class DoNeRFTRT(_PackedNet):
    """8-layer ReLU MLP, view encoding concatenated before the last layer (skip='auto')
    (helpers:1186-1343).  ``forward(input_pts[M,63], input_views[M,27]) -> [M,4]``."""
    _NET = ops.NET_NERF

    def __init__(self, D, W, skip, n_in, n_out):
        super().__init__()
        self.D, self.W, self.n_in, self.n_out = D, W, n_in, n_out
        pos_in = 63
        # skip='auto' (helpers:1190-1201): the view encoding enters at layer 7 D // 8 — the LAST layer for D <= 8 (the Fern configs: D = 8), a hidden
        # ReLU layer from D = 9 on.  The kernels take the first form at any depth 3 .. 8; the module is built as the reference builds it either way.
        view_at = D * 7 // 8
        self._supported = (3 <= D <= 8 and view_at == D - 1 and W == 256 and skip == 'auto' and n_in == 90 and n_out == 4)
        self.inputLocations = {0: (0, pos_in), view_at: (pos_in, n_in)}
        layers = [nn.Linear(pos_in, W)]
        for i in range(1, D):
            layers.append(nn.Linear((n_in - pos_in) + W if i == view_at else W, W if i != D - 1 else n_out))
        self.layers = nn.ModuleList(layers)
        for l in self.layers:
            nn.init.kaiming_normal_(l.weight)          # helpers:1243-1244

    def _linears(self):
        if not self._supported:
            raise PnrfError("DoNeRFTRT: the HIP kernels take W=256, skip='auto', n_in=90 (63 + 27), n_out=4 and depth 3 <= D <= 8 (from D = 9 on the "
                            "reference's skip='auto' feeds the view encoding into a hidden layer, helpers:1190-1201)")
        return list(self.layers)

    def forward(self, input_pts, input_views):
        return self.packed().forward(input_pts, input_views)


class NeRF(_PackedNet):
    """The fine network stages 1/2 train and save (helpers:792-847): 8 pts layers with a skip-concat of the
    position embedding after layer 4, alpha / feature heads, one view layer, rgb head.
    ``forward(x[M, 63+27]) -> [M,4] = [rgb, alpha]``.  Same ``state_dict`` keys as the reference."""
    _NET = ops.NET_NERFCLS

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.D, self.W, self.input_ch, self.input_ch_views, self.skips, self.use_viewdirs = D, W, input_ch, input_ch_views, skips, use_viewdirs
        self.pts_linears = nn.ModuleList([nn.Linear(input_ch, W)] +
                                         [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        self._supported = (D == 8 and W == 256 and input_ch == 63 and input_ch_views == 27 and list(skips) == [4] and use_viewdirs)

    def _linears(self):
        if not self._supported:
            raise PnrfError('NeRF: the HIP kernels are built for D=8, W=256, input_ch=63, input_ch_views=27, skips=[4], use_viewdirs=True')
        return list(self.pts_linears) + [self.feature_linear, self.alpha_linear, self.views_linears[0], self.rgb_linear]

    def forward(self, x, input_views=None):
        if input_views is None:
            x, input_views = x[..., :self.input_ch].contiguous(), x[..., self.input_ch:self.input_ch + self.input_ch_views].contiguous()
        return self.packed().forward(x, input_views)


def weights_from_modules(min_max_ray_net, refine_net, network_fine):
    """{'sampler','refine','nerf'} weight dict for ``pronerf_amd.render.Renderer``."""
    return {'sampler': min_max_ray_net.weights(), 'refine': refine_net.weights(), 'nerf': network_fine.weights()}


def weights_from_state_dicts(mmr_sd, refine_sd, fine_sd):
    """Checkpoint state dicts (keys of run_S_eS_eN_alter_trt.py:476-481) -> weight dict.
    ``fine_sd`` may carry DoNeRFTRT keys (``layers.{i}.*``) or the ``NeRF``-class keys (``pts_linears.*`` ...)
    that the released stage-2 trainer actually saves; the kernel is chosen from them."""
    def stack(sd):
        n = len([k for k in sd if k.startswith('fc_backbone.') and k.endswith('.weight')])
        return {'W': [sd[f'fc_backbone.{i}.weight'] for i in range(n)] + [sd['fc_output.weight']],
                'b': [sd[f'fc_backbone.{i}.bias'] for i in range(n)] + [sd['fc_output.bias']]}
    if any(k.startswith('pts_linears.') for k in fine_sd):        # NeRF class (what the released stage-2 trainer saves)
        names = [f'pts_linears.{i}' for i in range(8)] + ['feature_linear', 'alpha_linear', 'views_linears.0', 'rgb_linear']
        nerf = {'W': [fine_sd[f'{n}.weight'] for n in names], 'b': [fine_sd[f'{n}.bias'] for n in names]}
    elif any(k.startswith('layers.') for k in fine_sd):           # DoNeRFTRT
        n = len([k for k in fine_sd if k.endswith('.weight')])
        nerf = {'W': [fine_sd[f'layers.{i}.weight'] for i in range(n)], 'b': [fine_sd[f'layers.{i}.bias'] for i in range(n)]}
    else:
        raise PnrfError("fine-network state dict has neither 'layers.*' (DoNeRFTRT) nor 'pts_linears.*' (NeRF) keys")
    return {'sampler': stack(mmr_sd), 'refine': stack(refine_sd), 'nerf': nerf}


# ------------------------------------------------------------------------------- ray helpers
def get_rays(H, W, K, c2w):
    """rays_o, rays_d [H,W,3] on the GPU (helpers:2705-2714)."""
    dev = c2w.device if isinstance(c2w, torch.Tensor) and c2w.is_cuda else torch.device('cuda')
    _, orr = ops.frame_rays(K, c2w, H, W, device=dev)
    return orr[:, 0:3].reshape(H, W, 3), orr[:, 3:6].reshape(H, W, 3)


def get_rays_np(H, W, K, c2w):
    o, d = get_rays(H, W, K, c2w)
    return o.cpu().numpy(), d.cpu().numpy()


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """(helpers:2776-2793)."""
    return ops.ndc_rays(H, W, float(focal), float(near), rays_o.contiguous(), rays_d.contiguous())
