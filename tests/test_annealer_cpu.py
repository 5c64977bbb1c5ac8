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
