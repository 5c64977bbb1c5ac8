"""CPU: the scalar schedules of nr3d_lib_amd.models.annealers against the reference's (tests/golden/
ref_scalar_annealers.json, make_golden_scalar_annealers.py), and the annealed inv_s of the occupancy conversion."""
import json
import os

import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_scalar_annealers.json")))


def test_functional_forms():
    from nr3d_lib_amd.models.annealers import get_anneal_val
    for rec in GOLD["functional"]:
        for it, want in zip(GOLD["its"], rec["vals"]):
            assert get_anneal_val(it=it, **rec["cfg"]) == pytest.approx(want, rel=1e-12, abs=1e-15), (rec["cfg"], it)
    with pytest.raises(RuntimeError, match="Invalid type"):
        get_anneal_val("cosine", it=0, stop_it=10)


def test_object_forms():
    from nr3d_lib_amd.models.annealers import get_annealer
    seen = set()
    for rec in GOLD["objects"]:
        assert "error" not in rec, rec
        a = get_annealer(**json.loads(json.dumps(rec["cfg"])))
        seen.add(a.type)
        for it, want in zip(GOLD["its"], rec["vals"]):
            a.set_iter(it)
            assert a.get_val() == pytest.approx(want, rel=1e-12, abs=1e-15), (rec["cfg"], it)
        if a.type != "constant":
            a.set_val(7.0)
            assert a(123) == 7.0
    assert seen == {"linear", "logspace", "milestones", "constant", "partitions"}


def test_annealed_inv_s_of_the_occupancy_conversion():
    import torch
    from nr3d_lib_amd.models.accelerations.occgrid.ema_single import get_occ_val_fn, normalized_logistic_density
    sdf = torch.linspace(-0.2, 0.2, 11)
    cfg = dict(type="logspace", stop_it=100, start_val=10.0, stop_val=1000.0)
    fn = get_occ_val_fn("sdf", inv_s_anneal_cfg=dict(cfg, it=50))
    torch.testing.assert_close(fn(sdf), normalized_logistic_density(sdf, 100.0))
