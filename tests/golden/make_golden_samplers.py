#!/usr/bin/env python
"""Regenerates tests/golden/ref_samplers.json.  BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_samplers.py

Outputs of the reference's pure-PyTorch depth samplers (graphics/raysample.py: batch_sample_step_linear / _wrt_depth /
_wrt_sqrt_depth, unperturbed, with deltas) and NeuS opacity helpers (graphics/neus/neus_utils.py: neus_ray_cdf_to_alpha,
neus_ray_sdf_to_alpha, with and without the appended cdf) on fixed inputs.  Data only."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import torch

from make_golden import import_reference   # noqa: E402


def main():
    rs = import_reference("nr3d_lib.graphics.raysample")
    nu = import_reference("nr3d_lib.graphics.neus.neus_utils")
    near = torch.tensor([0.5, 1.0, 2.0, 0.05])
    far = torch.tensor([4.0, 1.5, 40.0, 9.0])
    out = dict(near=near.tolist(), far=far.tolist(), num_samples=9, samplers={})
    for name in ("batch_sample_step_linear", "batch_sample_step_wrt_depth", "batch_sample_step_wrt_sqrt_depth"):
        t, dt = getattr(rs, name)(near.clone(), far.clone(), 9, perturb=False, return_dt=True)
        out["samplers"][name] = dict(t=t.tolist(), dt=dt.tolist())
    torch.manual_seed(11)
    sdf = torch.randn(3, 12).cumsum(-1).flip(-1) * 0.1
    out["sdf"] = sdf.tolist()
    out["inv_s"] = 20.0
    out["neus"] = dict(
        ray_sdf_to_alpha=nu.neus_ray_sdf_to_alpha(sdf, 20.0).tolist(),
        ray_sdf_to_alpha_append=nu.neus_ray_sdf_to_alpha(sdf, 20.0, append_cdf_1=True).tolist(),
        ray_cdf_to_alpha=nu.neus_ray_cdf_to_alpha(torch.sigmoid(sdf * 20.0)).tolist())
    json.dump(out, open(os.path.join(HERE, "ref_samplers.json"), "w"), indent=1)
    print({k: list(v) if isinstance(v, dict) else v for k, v in out.items() if k in ("samplers", "neus")})


if __name__ == "__main__":
    main()
