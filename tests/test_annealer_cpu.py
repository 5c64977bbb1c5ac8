"""CPU: MultiresAnnealer (the coarse-to-fine max_level / window schedule that drives the hot path's max_level) against the
reference class's outputs over 1000 iterations (tests/golden/ref_annealer.json, make_golden_annealer.py), and its wiring
into LoTDEncoding (anneal_cfg / set_anneal_iter / space_cfg)."""
import json
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "ref_annealer.json")))


@pytest.mark.parametrize("case", CASES, ids=[f"{i}-{c['cfg']['type']}" for i, c in enumerate(CASES)])
def test_annealer_matches_reference(case):
    from nr3d_lib_amd.models.grid_encodings.multires_annealer import MultiresAnnealer
    a = MultiresAnnealer(**case["cfg"])
    assert a()[0] == case["default_max_level"]                  # without an iteration: fully annealed
    for it, want in enumerate(case["max_level"]):
        ml, w = a(it)
        assert ml == want, (it, ml, want)
        key = str(it)
        if key in case["windows"]:
            torch.testing.assert_close(w, torch.tensor(case["windows"][key]), rtol=0, atol=2e-6)
        if case["cfg"]["type"] == "hardmask":
            assert w is None
    a.set_iter(0)
    assert a()[0] == case["max_level"][0]
    with pytest.raises(RuntimeError, match="Invalid anneal_type"):
        MultiresAnnealer([2, 2], "linear", stop_it=10)


def test_lotd_encoding_anneal_and_space_cfg():
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTDEncoding
    cfg = dict(lod_res=[8, 12, 16, 24], lod_n_feats=[2, 2, 4, 2], lod_types=["Dense"] * 4)
    e = LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float, anneal_cfg=dict(type="cosine", start_it=0, stop_it=300, start_level=0),
                     space_cfg=dict(type="aabb", aabb=[[-1, -2, -3], [1, 2, 3]]))
    assert e.space is not None and e.space.aabb.tolist() == [[-1, -2, -3], [1, 2, 3]]
    assert e.max_level is None and e.window is None
    e.set_anneal_iter(0)
    assert e.max_level == 0 and tuple(e.window.shape) == (10,) and float(e.window[:2].min()) == 1.0 and float(e.window[2:].max()) == 0.0
    e.set_anneal_iter(150)
    assert e.max_level == 2 and 0.0 < float(e.window[4]) < 1.0 and float(e.window[8:].max()) == 0.0
    e.set_anneal_iter(10 ** 6)
    assert e.max_level == 3 and float(e.window.min()) == 1.0
    assert LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float, space_cfg=dict(type="unbounded")).space is None
    with pytest.raises(RuntimeError, match="Invalid space_type"):
        LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float, space_cfg=dict(type="sphere"))


def test_lotd_encoding_level_views_grad_clip_and_stats():
    """``lodN`` attribute views, ``clip_grad_and_update_ema`` (per-level norm clipping against a running norm, reference
    lotd_encoding.py:471-486) and ``stat_param`` (the reference logger's keys, :488-508) -- host logic, no kernel involved"""
    from nr3d_lib_amd.models.grid_encodings.lotd import LoTDEncoding
    cfg = dict(lod_res=[8, 12, 10], lod_n_feats=[2, 4, 2], lod_types=["Dense", "VM", "CP"])
    e = LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float, clip_level_grad_ema_factor=2.0)
    # attribute views are views into flattened_params
    assert e.lod0.shape == (512, 2) and e.lod0.data_ptr() == e.flattened_params.data_ptr()
    assert torch.equal(e.lod1_vec, e.get_level_param(1, 'vec')) and torch.equal(e.lod1_mat2, e.get_level_param(1, 'mat', 2))
    e.lod2 = torch.full_like(e.lod2, 0.25)
    assert float(e.get_level_param(2).min()) == 0.25 and 'lod2' not in e.__dict__
    with pytest.raises(AttributeError):
        e.lodestar
    # clipping: EMA moves 1 % towards the current norm, gradient norm capped at factor * EMA
    g = torch.randn_like(e.flattened_params)
    e.flattened_params.grad = g.clone()
    offs = e.lod_meta.level_offsets
    norms = [float(g[offs[l]:offs[l + 1]].norm()) for l in range(3)]
    e.clip_grad_and_update_ema()
    for l in range(3):
        ema = 0.99 * 0.1 + 0.01 * norms[l]
        assert abs(float(e.level_grad_norm_ema[l]) - ema) < 1e-5
        got = float(e.flattened_params.grad[offs[l]:offs[l + 1]].norm())
        assert abs(got - min(norms[l], 2.0 * ema)) < 1e-3 * norms[l]
    st = e.stat_param(with_grad=True, prefix='enc')
    for key in ('enc.total.mean', 'enc.grad_total.norm', 'enc.lv.0.absmax', 'enc.lv.1.vec.std', 'enc.lv.1.mat.max',
                'enc.grad.lv.2.norm', 'enc.grad.lv.1.ema'):
        assert key in st, key
    assert abs(st['enc.lv.2.mean'] - 0.25) < 1e-7
    # an encoder built without the factor: clipping is a no-op, no EMA keys
    e2 = LoTDEncoding(3, lotd_cfg=cfg, dtype=torch.float)
    e2.flattened_params.grad = g.clone()
    e2.clip_grad_and_update_ema()
    assert torch.equal(e2.flattened_params.grad, g) and not any(k.endswith('.ema') for k in e2.stat_param())
