"""GPU: NeuS multi-stage up-sampling driver against a per-ray numpy restatement built on the
CPU oracle's pack ops."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


class SphereSDF(torch.nn.Module):
    """analytic model: sdf = |x| - r; colour = position-dependent; the protocol of the NeuS driver"""
    use_view_dirs = True

    def __init__(self, accel, radius=0.62, inv_s=48.0):
        super().__init__()
        self.accel, self.radius, self.inv_s = accel, radius, inv_s

    def forward_inv_s(self):
        return self.inv_s

    def forward_sdf(self, x, **kw):
        return dict(sdf=x.norm(dim=-1) - self.radius)

    def forward(self, x, v=None, nablas_has_grad=False, with_rgb=True, with_normal=True, **kw):
        out = dict(sdf=x.norm(dim=-1) - self.radius)
        if with_normal:
            out["nablas"] = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-10)
        if with_rgb:
            out["rgb"] = torch.sigmoid(x + (v if v is not None else 0))
        return out


def _scene(dev, side=12, res=32):
    from demo_field import StaticOccGridAccel, pinhole_rays
    c = (np.stack(np.meshgrid(*[np.arange(res)] * 3, indexing="ij"), -1) + 0.5) / res * 2 - 1
    r = np.linalg.norm(c, axis=-1)
    occ = (r > 0.45) & (r < 0.8)
    step = 0.03
    model = SphereSDF(StaticOccGridAccel(torch.from_numpy(occ).to(dev), step, max_steps=128)).eval()
    o, d, near, far = pinhole_rays(side, dev, fov=0.25)
    n = side * side
    return model, dict(num_rays=n, rays_o=o, rays_d=d, near=near, far=far, rays_inds=torch.arange(n, device=dev)), occ, step


def _reference_fine_depths(oracle, o, d, near, far, occ, step, radius, stages, n_fine, use_estimate):
    """per-ray restatement of the up-sampling loop on numpy + the oracle's alpha_to_vw / invert_cdf"""
    roi = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    pi, ts, te, ridx, _ = oracle.ray_marching(o, d, near, far, roi, occ, 0, np.float32(step), 1e10, 0.0, 128, True)[:5]
    hit = np.nonzero(pi[:, 1])[0]
    out = []
    for r in hit:
        b, n = pi[r]
        t = ts[b:b + n, 0].astype(np.float32)
        fines = []
        for f, nf in zip(stages, n_fine):
            x = o[r] + d[r] * t[:, None]
            s = (np.linalg.norm(x, axis=-1) - radius).astype(np.float32)
            inv_s = np.float32(64.0 * f)
            pinfo = np.array([[0, len(t)]], np.int64)
            if use_estimate:
                ds, dt = np.append(np.diff(s), 0).astype(np.float32), np.append(np.diff(t), 0).astype(np.float32)
                slope = ds / (dt + np.float32(1e-5))
                prev = np.roll(slope, 1); prev[0] = 0
                slope = np.clip(np.minimum(prev, slope), -10, 0).astype(np.float32)
                mid, half = s + ds * np.float32(0.5), slope * dt * np.float32(0.5)
                sig = lambda v: (1 / (1 + np.exp(-(v * inv_s).astype(np.float64)))).astype(np.float32)
                cp, cn = sig(mid - half), sig(mid + half)
                alpha = np.maximum((cp - cn) / (cp + np.float32(1e-5)), 0).astype(np.float32)
            else:
                cdf = (1 / (1 + np.exp(-(s * inv_s).astype(np.float64)))).astype(np.float32)
                alpha = np.maximum(-np.append(np.diff(cdf), 0).astype(np.float32) / (cdf + np.float32(1e-5)), 0).astype(np.float32)
            w = oracle.packed_alpha_to_vw_forward(alpha, pinfo, 1e-4, 0.0, False)[0]
            c = np.concatenate([[0], np.cumsum(w[:-1], dtype=np.float32)]).astype(np.float32)
            c = (c / max(c[-1], np.float32(1e-5))).astype(np.float32)
            u = np.linspace(0, 1, nf + 2, dtype=np.float32)[1:-1][None]
            fine = oracle.packed_invert_cdf(t, c, u, pinfo)[0][0]
            fines.append(fine)
            t = np.sort(np.concatenate([t, fine])).astype(np.float32)
        out.append(np.sort(np.concatenate(fines)))
    return hit, np.stack(out), pi


@pytest.mark.parametrize("use_estimate", [False, True])
def test_fine_buffer_matches_per_ray_restatement(oracle, dev, use_estimate):
    from nr3d_lib_amd.graphics.neus import neus_ray_query_march_occ_multi_upsample
    model, rays, occ, step = _scene(dev)
    stages, nf = [1, 4, 16], 8
    with torch.no_grad():
        vb, details = neus_ray_query_march_occ_multi_upsample(model, rays, num_fine=nf, upsample_inv_s_factors=stages,
                                                              upsample_use_estimate_alpha=use_estimate)
    o, d = rays["rays_o"].cpu().numpy(), rays["rays_d"].cpu().numpy()
    hit, fine_ref, pi = _reference_fine_depths(oracle, o, d, rays["near"].cpu().numpy(), rays["far"].cpu().numpy(), occ, step,
                                               model.radius, stages, [nf // 2 * 2 + 1] * 3, use_estimate)
    assert vb["type"] == "batched" and vb["num_per_hit"] == 3 * 9 - 1
    np.testing.assert_array_equal(vb["rays_inds_hit"].cpu().numpy(), hit)
    np.testing.assert_array_equal(details["march.num_per_ray"].cpu().numpy(), pi[hit, 1])
    mids_ref = fine_ref[:, :-1] + np.diff(fine_ref, axis=-1) / 2
    # the chain is fp32 on both sides; a last-bit difference in an sdf can move a sample inside its CDF bin
    close = np.isclose(vb["t"].cpu().numpy(), mids_ref, rtol=0, atol=2e-4)
    assert close.mean() > 0.995, f"only {close.mean():.4f} of the fine depths agree"
    assert vb["rgb"].shape == (*vb["t"].shape, 3) and vb["nablas"].shape == (*vb["t"].shape, 3)
    a = vb["opacity_alpha"].cpu().numpy()
    assert a.shape == mids_ref.shape and (a >= 0).all() and (a <= 1 + 1e-5).all()
    # the fine samples concentrate where the ray crosses the surface |x| = r
    x = o[hit][:, None] + d[hit][:, None] * vb["t"].cpu().numpy()[..., None]
    crossing = np.abs(np.linalg.norm(x, axis=-1) - model.radius).min(-1)
    assert np.median(crossing) < 0.02


def test_coarse_plus_fine_packed_buffer_and_coarse_only(dev):
    from nr3d_lib_amd.graphics.nerf import composite_packed_volume_buffer
    from nr3d_lib_amd.graphics.neus import neus_ray_query_march_occ_multi_upsample
    model, rays, occ, step = _scene(dev)
    n = rays["num_rays"]
    with torch.no_grad():
        vb, details = neus_ray_query_march_occ_multi_upsample(model, rays, num_coarse=16, num_fine=4,
                                                              upsample_inv_s_factors=[1, 4], debug_query_data=(dbg := {}))
    assert vb["type"] == "packed" and vb["pack_infos_hit"].shape == (n, 2)
    per_ray = vb["pack_infos_hit"][:, 1].cpu().numpy()
    hit = (details["march.num_per_ray"] > 0)
    assert set(np.unique(per_ray)) <= {17, 17 + 2 * 5} and int(vb["pack_infos_hit"][:, 1].sum()) == vb["t"].numel()
    t = vb["t"].cpu().numpy()
    for b, k in vb["pack_infos_hit"].cpu().numpy()[:: max(1, n // 16)]:
        assert (np.diff(t[b:b + k]) >= -1e-6).all()                     # depths sorted inside every pack
    assert set(dbg) == {"coarse", "fine"} and dbg["fine"]["upsample_stages"].min() >= 1
    out = composite_packed_volume_buffer(vb, n)
    m = out["mask_volume"].cpu().numpy()
    assert (m >= -1e-6).all() and (m <= 1 + 1e-4).all() and m.max() > 0.9    # rays through the sphere are opaque
    # nothing marched (empty grid): coarse-only batched buffer, or empty without coarse samples
    model.accel.occ_grid = torch.zeros_like(model.accel.occ_grid)
    with torch.no_grad():
        vb2, det2 = neus_ray_query_march_occ_multi_upsample(model, rays, num_coarse=16, coarse_step_cfg=dict(step_mode="sqrt_depth"))
        assert vb2["type"] == "batched" and vb2["t"].shape == (n, 16) and det2 == {"render.num_per_ray": 16}
        vb3, det3 = neus_ray_query_march_occ_multi_upsample(model, rays)
        assert vb3["type"] == "empty" and det3 == {}


def _expect_compressed(oracle, alpha, t, pack_infos, pack_rays):
    """the oracle's compaction of (alpha, t) laid out in packs -> kept samples, their packs, the rays of those packs"""
    _, cpi, sel = oracle.packed_alpha_to_vw_forward(alpha, pack_infos, 1e-4, 0.0, True)
    keep = cpi[:, 1] > 0
    return alpha[sel], t[sel], cpi[keep], pack_rays[keep]


def test_compressed_fine_buffer_is_the_compaction_of_the_batched_one(oracle, dev):
    from nr3d_lib_amd.graphics.nerf import composite_packed_volume_buffer
    from nr3d_lib_amd.graphics.neus import (neus_ray_query_march_occ_multi_upsample,
                                            neus_ray_query_march_occ_multi_upsample_compressed)
    model, rays, occ, step = _scene(dev)
    n = rays["num_rays"]
    kw = dict(num_fine=8, upsample_inv_s_factors=[1, 4, 16])
    with torch.no_grad():
        full, _ = neus_ray_query_march_occ_multi_upsample(model, rays, **kw)
        vb, details = neus_ray_query_march_occ_multi_upsample_compressed(model, rays, **kw)
    assert vb["type"] == "packed"
    a, t = full["opacity_alpha"].cpu().numpy(), full["t"].cpu().numpy()
    k = a.shape[1]
    pinfo = np.stack([np.arange(a.shape[0]) * k, np.full(a.shape[0], k)], 1).astype(np.int64)
    a_ref, t_ref, pi_ref, rays_ref = _expect_compressed(oracle, a.ravel(), t.ravel(), pinfo, full["rays_inds_hit"].cpu().numpy())
    np.testing.assert_array_equal(vb["pack_infos_hit"].cpu().numpy(), pi_ref)
    np.testing.assert_array_equal(vb["rays_inds_hit"].cpu().numpy(), rays_ref)
    np.testing.assert_array_equal(vb["opacity_alpha"].cpu().numpy(), a_ref)
    np.testing.assert_array_equal(vb["t"].cpu().numpy(), t_ref)
    assert 0 < vb["t"].numel() < a.size                              # something was pruned, something is left
    assert details["render.num_per_ray0"] == k
    np.testing.assert_array_equal(details["render.num_per_ray"].cpu().numpy(), pi_ref[:, 1])
    assert vb["rgb"].shape == (vb["t"].numel(), 3) and vb["nablas"].shape == (vb["t"].numel(), 3)
    # net_x are the kept samples' positions on their own rays
    o, d = rays["rays_o"].cpu().numpy(), rays["rays_d"].cpu().numpy()
    ridx = np.repeat(rays_ref, pi_ref[:, 1])
    np.testing.assert_allclose(vb["net_x"].cpu().numpy(), o[ridx] + d[ridx] * t_ref[:, None], rtol=0, atol=1e-6)
    # compression does not change the image: dropped samples carry (almost) no weight
    m = composite_packed_volume_buffer(vb, n)["mask_volume"].cpu().numpy()
    w = a * np.cumprod(np.concatenate([np.ones_like(a[:, :1]), 1 - a[:, :-1]], 1), 1)
    m_ref = np.zeros(n, np.float32); m_ref[full["rays_inds_hit"].cpu().numpy()] = w.sum(1)
    np.testing.assert_allclose(m, m_ref, rtol=0, atol=2e-4)


def test_compressed_coarse_plus_fine_and_coarse_only(oracle, dev):
    from nr3d_lib_amd.graphics.neus import (neus_ray_query_march_occ_multi_upsample,
                                            neus_ray_query_march_occ_multi_upsample_compressed,
                                            neus_ray_query_march_occ_multi_upsample_compressed_strategy)
    model, rays, occ, step = _scene(dev)
    n = rays["num_rays"]
    kw = dict(num_coarse=16, num_fine=4, upsample_inv_s_factors=[1, 4])
    with torch.no_grad():
        full, _ = neus_ray_query_march_occ_multi_upsample(model, rays, **kw)
        vb, details = neus_ray_query_march_occ_multi_upsample_compressed(model, rays, **kw)
    a_ref, t_ref, pi_ref, rays_ref = _expect_compressed(oracle, full["opacity_alpha"].cpu().numpy(), full["t"].cpu().numpy(),
                                                        full["pack_infos_hit"].cpu().numpy(), np.arange(n))
    np.testing.assert_array_equal(vb["pack_infos_hit"].cpu().numpy(), pi_ref)
    np.testing.assert_array_equal(vb["rays_inds_hit"].cpu().numpy(), rays_ref)
    np.testing.assert_array_equal(vb["opacity_alpha"].cpu().numpy(), a_ref)
    np.testing.assert_array_equal(vb["t"].cpu().numpy(), t_ref)
    np.testing.assert_array_equal(details["render.num_per_ray0"].cpu().numpy(), full["pack_infos_hit"][:, 1].cpu().numpy())
    assert len(rays_ref) < n                                         # rays that miss the sphere keep nothing
    # gradients reach the model through the kept samples only
    model.train()
    radius = torch.nn.Parameter(torch.tensor(model.radius, device=dev))
    model.radius = radius
    vb_t, _ = neus_ray_query_march_occ_multi_upsample_compressed(model, rays, **kw)
    vb_t["opacity_alpha"].sum().backward()
    assert radius.grad is not None and torch.isfinite(radius.grad) and radius.grad.abs() > 0
    del model.radius
    model.radius = float(radius.detach()); model.eval()
    # nothing marched: compressed coarse samples; without coarse samples, empty
    model.accel.occ_grid = torch.zeros_like(model.accel.occ_grid)
    with torch.no_grad():
        full2, _ = neus_ray_query_march_occ_multi_upsample(model, rays, num_coarse=16)
        vb2, det2 = neus_ray_query_march_occ_multi_upsample_compressed(model, rays, num_coarse=16)
        a2 = full2["opacity_alpha"].cpu().numpy()
        pinfo = np.stack([np.arange(n) * 16, np.full(n, 16)], 1).astype(np.int64)
        a_ref, t_ref, pi_ref, rays_ref = _expect_compressed(oracle, a2.ravel(), full2["t"].cpu().numpy().ravel(), pinfo, np.arange(n))
        assert vb2["type"] == "packed" and det2["render.num_per_ray0"] == 16
        np.testing.assert_array_equal(vb2["pack_infos_hit"].cpu().numpy(), pi_ref)
        np.testing.assert_array_equal(vb2["rays_inds_hit"].cpu().numpy(), rays_ref)
        np.testing.assert_array_equal(vb2["t"].cpu().numpy(), t_ref)
        vb3, det3 = neus_ray_query_march_occ_multi_upsample_compressed(model, rays)
        assert vb3["type"] == "empty" and det3 == {}
    with pytest.raises(NotImplementedError):
        neus_ray_query_march_occ_multi_upsample_compressed_strategy(model, rays)
