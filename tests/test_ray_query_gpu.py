"""GPU: the march -> prune -> encode -> composite driver against the CPU oracle chain."""
import os
import sys

import numpy as np
import pytest
import torch

from util import assert_close, assert_equal

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


def _scene(dev, side=24, res=64, seed=2):
    from demo_field import DemoField, pinhole_rays
    rng = np.random.default_rng(seed)
    c = (np.stack(np.meshgrid(*[np.arange(res)] * 3, indexing="ij"), -1) + 0.5) / res * 2 - 1
    r = np.linalg.norm(c, axis=-1)
    occ = ((r > 0.45) & (r < 0.8)) | (rng.random((res,) * 3) > 0.97)
    step = 2 * 3 ** 0.5 / 256
    model = DemoField(torch.from_numpy(occ).to(dev), step, max_steps=256, seed=seed, device=dev)
    o, d, near, far = pinhole_rays(side, dev)
    n = side * side
    rays = dict(num_rays=n, rays_o=o, rays_d=d, near=near, far=far, rays_inds=torch.arange(n, device=dev))
    return model, rays, occ, step


def test_volume_buffer_matches_oracle_chain(oracle, dev):
    from nr3d_lib_amd.graphics.nerf import nerf_ray_query_march_occ
    model, rays, occ, step = _scene(dev)
    with torch.no_grad():
        vb, details = nerf_ray_query_march_occ(model, rays, with_rgb=True, compression=False)
    assert vb["type"] == "packed"
    o, d = rays["rays_o"].cpu().numpy(), rays["rays_d"].cpu().numpy()
    roi = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    pi, ts, te, ridx, gidx = oracle.ray_marching(o, d, rays["near"].cpu().numpy(), rays["far"].cpu().numpy(), roi, occ, 0,
                                                 np.float32(step), 1e10, 0.0, 256, True)[:5]
    hit = np.nonzero(pi[:, 1])[0]
    assert_equal(vb["rays_inds_hit"], hit, "rays_inds_hit")
    assert_equal(vb["pack_infos_hit"], pi[hit].astype(np.int64), "pack_infos_hit")
    assert_equal(vb["t"], ts[:, 0], "t")                                       # marcher t-values: bit-exact
    assert_equal(details["march.num_per_ray"], pi[hit, 1].astype(np.int64), "march.num_per_ray")
    # encoder on the oracle + the same MLP weights on the CPU
    x = o[ridx] + d[ridx] * ts
    enc = model.encoding
    m_ref = oracle.lotd_create_meta(3, enc.params["lod_res"], enc.params["lod_n_feats"], enc.params["lod_types"],
                                    enc.params["hashmap_size"])
    x01 = np.clip((x + 1) * np.float32(0.5), 1e-6, 1 - 1e-6).astype(np.float32)
    feat, _ = oracle.lotd_fwd(m_ref, x01, model.grid.detach().cpu().numpy())
    import copy
    dens = copy.deepcopy(model.density).cpu()
    h = dens(torch.from_numpy(feat))
    sigma_ref = (torch.nn.functional.softplus(h[:, 0] + 2.0) * 20.0).detach().numpy()
    np.testing.assert_allclose(vb["sigma"].cpu().numpy(), sigma_ref, rtol=2e-4, atol=1e-4)
    alpha_ref = 1 - np.exp(-sigma_ref * (te - ts)[:, 0])
    np.testing.assert_allclose(vb["opacity_alpha"].cpu().numpy(), alpha_ref, rtol=2e-4, atol=1e-5)


def test_compression_prunes_without_changing_the_image(dev):
    from nr3d_lib_amd.graphics.nerf import composite_packed_volume_buffer, nerf_ray_query_march_occ
    model, rays, occ, step = _scene(dev, seed=3)
    n = rays["num_rays"]
    img = {}
    for comp in (False, True):
        model.zero_grad()
        vb, det = nerf_ray_query_march_occ(model, rays, with_rgb=True, compression=comp)
        out = composite_packed_volume_buffer(vb, n)
        (out["rgb_volume"].sum() + out["depth_volume"].sum() + out["mask_volume"].sum()).backward()
        img[comp] = (out, det, model.grid.grad.clone(), vb)
    full, pruned = img[False], img[True]
    assert int(pruned[1]["render.num_per_ray"].sum()) < int(full[1]["render.num_per_ray"].sum())
    assert int(pruned[1]["render.num_per_ray0"].sum()) == int(full[1]["march.num_per_ray"].sum())
    for k in ("mask_volume", "depth_volume", "rgb_volume"):
        assert_close(pruned[0][k], full[0][k].detach().cpu().numpy(), rel=2e-3, name=k)   # early_stop_eps = 1e-4
    g_full, g_pruned = full[2], pruned[2]
    assert torch.isfinite(g_pruned).all() and float(g_pruned.abs().max()) > 0
    rel = float((g_full - g_pruned).norm() / g_full.norm())
    assert rel < 2e-2, f"encoder gradient changed by {rel:.3e} under pruning"
    # the samples that survive are a sub-sequence of the marched ones
    assert set(pruned[3]["rays_inds_hit"].tolist()) <= set(full[3]["rays_inds_hit"].tolist())


def test_density_only_and_empty(dev):
    from nr3d_lib_amd.graphics.nerf import composite_packed_volume_buffer, nerf_ray_query_march_occ
    model, rays, occ, step = _scene(dev, side=8)
    with torch.no_grad():
        vb, _ = nerf_ray_query_march_occ(model, rays, with_rgb=False, compression=True)
        assert "rgb" not in vb and vb["sigma"].shape == vb["t"].shape
        out = composite_packed_volume_buffer(vb, rays["num_rays"], with_rgb=False)
        assert set(out) == {"mask_volume", "depth_volume"}
        model.accel.occ_grid = torch.zeros_like(model.accel.occ_grid)
        vb, det = nerf_ray_query_march_occ(model, rays)
        assert vb["type"] == "empty" and det == {}
        out = composite_packed_volume_buffer(vb, rays["num_rays"], device=dev)
        assert float(out["mask_volume"].abs().sum()) == 0.0
        none = dict(rays, num_rays=0)
        assert nerf_ray_query_march_occ(model, none)[0]["type"] == "empty"
