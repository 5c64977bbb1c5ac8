#!/usr/bin/env python
"""Regenerates tests/golden/ref_blocks.npz.  BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_blocks.py

Outputs of the reference's own pure-PyTorch host classes on fixed inputs: the MLP block (models/blocks/mlp.py: plain,
ragged widths, skip connection, no bias, output activation; randomly initialised by the reference, weights stored next to
the outputs), AABBSpace (models/spatial/aabb.py: normalisation, ray_test) and the occupancy-value helpers
(models/accelerations/occgrid/utils.py: binarize, sdf_to_occ_val; maths/common.py: normalized_logistic_density) and the
scatter-free parts of OccGridEma (ema_single.py: query, try_shrink, rescale_volume).
Data only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import numpy as np
import torch

from make_golden import import_reference   # noqa: E402

MLP_CASES = {
    "plain": dict(in_features=32, out_features=16, D=2, W=64, activation="relu"),
    "ragged_out_relu": dict(in_features=18, out_features=3, D=1, W=[40], activation="relu", output_activation="relu"),
    "skip": dict(in_features=6, out_features=3, D=3, W=[8, 10, 12], skips=[2], activation="relu", output_activation="sigmoid"),
    "nobias_softplus": dict(in_features=5, out_features=4, D=2, W=16, bias=False, last_bias=True, activation={"type": "softplus", "beta": 100.0}),
    "linear_only": dict(in_features=7, out_features=2, D=0, W=[], activation="relu"),
}


def main():
    blocks = import_reference("nr3d_lib.models.blocks.mlp")
    aabb = import_reference("nr3d_lib.models.spatial.aabb")
    occ = import_reference("nr3d_lib.models.accelerations.occgrid.utils")
    out = {}
    for name, kw in MLP_CASES.items():
        torch.manual_seed(hash(name) % 1000)
        m = blocks.MLP(**kw, dtype=torch.float, device="cpu")
        x = torch.randn(3, 11, kw["in_features"])
        with torch.no_grad():
            y = m(x) if kw["D"] == 0 else m(x, return_last=True)
        out[f"mlp_{name}_x"] = x.numpy()
        out[f"mlp_{name}_y"] = (y if kw["D"] == 0 else y[0]).numpy()
        if kw["D"] > 0:
            out[f"mlp_{name}_last"] = y[1].numpy()
        for k, v in m.state_dict().items():
            out[f"mlp_{name}_sd_{k}"] = v.numpy()
    sp = aabb.AABBSpace(aabb=[[-2., -1, 0], [2, 1, 4]], dtype=torch.float, device="cpu")
    torch.manual_seed(5)
    o = torch.randn(64, 3) * 3
    d = torch.nn.functional.normalize(torch.randn(64, 3), dim=-1)
    w = torch.randn(20, 3) * 2
    out["aabb_pts"] = w.numpy()
    out["aabb_norm"] = sp.normalize_coords(w).numpy()
    out["aabb_unnorm"] = sp.unnormalize_coords(sp.normalize_coords(w)).numpy()
    out["aabb_contains"] = sp.contains(w).numpy()
    out["aabb_o"], out["aabb_d"] = o.numpy(), d.numpy()
    for tag, kw in (("free", {}), ("clip", dict(near=0.5, far=6.0))):
        rt = sp.ray_test(o, d, **kw)
        out[f"aabb_rt_{tag}_inds"] = rt["rays_inds"].numpy()
        out[f"aabb_rt_{tag}_near"], out[f"aabb_rt_{tag}_far"] = rt["near"].numpy(), rt["far"].numpy()
        out[f"aabb_rt_{tag}_o"], out[f"aabb_rt_{tag}_d"] = rt["rays_o"].numpy(), rt["rays_d"].numpy()
    v = torch.linspace(-1, 1, 41)
    out["occ_v"] = v.numpy()
    out["occ_sdf10"] = occ.sdf_to_occ_val(v.clone(), inv_s=10.0).numpy()
    out["occ_bin"] = occ.binarize(v, 0.3).numpy()
    out["occ_bin_mean"] = occ.binarize(v, 0.3, consider_mean=True).numpy()
    out["occ_bin_const"] = occ.binarize(torch.full((5,), 0.2), 0.3, consider_mean=True).numpy()
    # OccGridEma: the pure-torch parts (no scatter): constant init, query, try_shrink, rescale_volume
    ema = import_reference("nr3d_lib.models.accelerations.occgrid.ema_single")
    g = ema.OccGridEma([8, 10, 6], occ_thre=0.5, init_cfg=dict(mode="constant", constant_value=0.0), device="cpu")
    g.init()
    torch.manual_seed(9)
    val = torch.zeros(8, 10, 6)
    val[2:5, 3:8, 1:4] = torch.rand(3, 5, 3) + 0.2
    g.occ_val_grid.copy_(val)
    g.occ_grid = occ.binarize(g.occ_val_grid, 0.5, False)
    q = torch.rand(50, 3) * 2.4 - 1.2
    old = torch.tensor([[-1., -2, -1], [1, 2, 3]])
    out["ema_val"], out["ema_q"] = val.numpy(), q.numpy()
    out["ema_occ"] = g.occ_grid.numpy()
    out["ema_query"] = g.query(q).numpy()
    out["ema_old_aabb"] = old.numpy()
    new = g.try_shrink(old)
    out["ema_shrink"] = new.numpy()
    g.rescale_volume(old, new)
    out["ema_rescaled_val"], out["ema_rescaled_occ"] = g.occ_val_grid.numpy(), g.occ_grid.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_blocks.npz"), **out)
    print(sorted(k for k in out if not k.startswith("mlp_") or k.endswith("_y")))


if __name__ == "__main__":
    main()
