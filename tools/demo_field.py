"""tools/demo_field.py -- TEST / BENCH INFRASTRUCTURE: the smallest model that satisfies the protocol of
nerf_ray_query_march_occ (accel.ray_march, query_density, forward_density, forward).  16-level NGP LoTD encoder (the
HIP path) + two tiny MLP blocks with random weights; not part of the product."""
import os

import torch
import torch.nn as nn

from nr3d_lib_amd.graphics.raymarch.occgrid_raymarch import occgrid_raymarch
from nr3d_lib_amd.models.blocks import MLP
from nr3d_lib_amd.models.grid_encodings.lotd import LoTD, gen_ngp_cfg


class StaticOccGridAccel:
    """occupancy grid over ``roi`` with the marching parameters fixed at construction"""

    def __init__(self, occ_grid: torch.Tensor, step_size: float, max_steps: int = 512, roi: torch.Tensor = None):
        self.occ_grid, self.step_size, self.max_steps, self.roi = occ_grid, step_size, max_steps, roi

    def ray_march(self, rays_o, rays_d, near=None, far=None, perturb=False, **march_cfg):
        cfg = dict(step_size=self.step_size, max_steps=self.max_steps, roi=self.roi)
        cfg.update(march_cfg)
        return occgrid_raymarch(self.occ_grid, rays_o, rays_d, near, far, perturb=perturb, **cfg)


class DemoField(nn.Module):
    use_view_dirs = True
    # forward() is evaluated point by point and every per-sample output is a top-level tensor of its dict: the driver may query
    # the rendered samples along a Morton curve (nerf_ray_query.py: opt-in)
    pointwise_forward = True

    def __init__(self, occ_grid, step_size, max_steps=512, hidden=32, seed=0, device=None, precision="float"):
        """precision: "float" = fp32 end to end, like the headline benchmark; "half" = the reference's default storage -- half LoTD
        tables / features (lotd.py: dtype=torch.half) and half decoders (its tcnn FullyFusedMLP; here MLP(dtype=half) on the f16 MFMA)"""
        super().__init__()
        cfg = gen_ngp_cfg()
        dt = torch.half if precision == "half" else torch.float
        self.encoding = LoTD(3, cfg["lod_res"], cfg["lod_n_feats"], cfg["lod_types"], hashmap_size=cfg["hashmap_size"], dtype=dt)
        g = torch.Generator().manual_seed(seed)
        n_params = self.encoding.meta.n_params if hasattr(self.encoding, "meta") else self.encoding.n_params
        self.grid = nn.Parameter(torch.empty(n_params).uniform_(-1e-1, 1e-1, generator=g).to(dt))
        e = self.encoding.out_features
        # the decoder blocks of the package (fused MFMA kernels when they apply; tools/bench: NR3D_DEMO_TORCH_MLP=1 for A/B)
        if os.environ.get("NR3D_DEMO_TORCH_MLP") == "1":
            self.density = nn.Sequential(nn.Linear(e, hidden), nn.ReLU(), nn.Linear(hidden, 1 + 15))
            self.color = nn.Sequential(nn.Linear(15 + 3, hidden), nn.ReLU(), nn.Linear(hidden, 3))
        else:
            self.density = MLP(e, 1 + 15, D=1, W=hidden, dtype=dt)
            self.color = MLP(15 + 3, 3, D=1, W=hidden, dtype=dt)
        for p, s in zip(self.parameters(), range(100)):
            if p is not self.grid:
                with torch.no_grad():
                    p.copy_((torch.randn(p.shape, generator=g) * (0.5 if p.dim() > 1 else 0.1)).to(p.dtype))
        with torch.no_grad():                               # sigma = 20 softplus(h0 + 2): the shift lives in the bias
            last = [m for m in self.density.modules() if getattr(m, "bias", None) is not None][-1]
            last.bias[0] += 2.0
        self.register_buffer("_half", torch.tensor(0.5), persistent=False)
        self.accel = StaticOccGridAccel(occ_grid, step_size, max_steps)
        if device is not None:
            self.to(device)

    def _h(self, x):
        # [-1, 1]^3 -> [0, 1]^3 in one launch (the encoder clamps to [1e-6, 1 - 1e-6] itself, lotd.py)
        feat = self.encoding(torch.addcmul(self._half, x, self._half), self.grid)
        h = self.density(feat if feat.dtype == self.density.dtype else feat.float())
        # sigma in fp32; the geometry features stay in the decoder's dtype (half decoders: no [n, 16] half -> float pass)
        # (one split, not two slices: the backward of two slices of h is two zero-filled [n, 16] buffers, two copies and an add; the
        # backward of a split is one concatenation)
        h0, geo = h.split([1, h.shape[-1] - 1], dim=-1)
        return torch.nn.functional.softplus(h0[..., 0].float()) * 20.0, geo

    def query_density(self, x, **kw):
        if not torch.is_grad_enabled() and isinstance(self.density, MLP):
            # the pruning query (no grad): only the density column leaves the decoder (LoTD.forward_decoded: in one kernel with the
            # encoder when lotd.FUSE_DECODED is on, else the two calls with the decoder's last layer cut to that column)
            h0 = self.encoding.forward_decoded(torch.addcmul(self._half, x, self._half), self.grid, self.density, out_cols=1)
            return torch.nn.functional.softplus(h0[..., 0].float()) * 20.0
        return self._h(x)[0]

    def forward_density(self, x, **kw):
        return dict(sigma=self._h(x)[0])

    def forward(self, x, v=None, **kw):
        sigma, geo = self._h(x)
        vv = v if v is not None else torch.zeros_like(x)
        rgb = torch.sigmoid(self.color(torch.cat([geo, vv.to(geo.dtype)], -1)).float())
        return dict(sigma=sigma, rgb=rgb)


def pinhole_rays(side, device, fov=0.4, dist=4.0, shift=0.0):
    """side^2 rays from a pinhole at (shift, 0, -dist) looking along +z, with near/far from the [-1, 1]^3 box"""
    u = torch.linspace(-fov, fov, side)
    uu, vv = torch.meshgrid(u, u, indexing="ij")
    n = side * side
    d = torch.stack([uu.flatten(), vv.flatten(), torch.ones(n)], 1)
    d = (d / d.norm(dim=1, keepdim=True)).to(device)
    o = torch.tensor([float(shift), 0.0, -dist]).repeat(n, 1).to(device)
    t1, t2 = (-1 - o) / d, (1 - o) / d
    near = torch.minimum(t1, t2).amax(1).clamp_min(0).contiguous()
    far = torch.maximum(t1, t2).amin(1).contiguous()
    far = torch.where(far > near, far, near).contiguous()
    return o, d, near, far
