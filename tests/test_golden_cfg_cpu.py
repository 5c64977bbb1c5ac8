"""CPU: the LoTD level-layout generators against the outputs of the reference's own functions
(tests/golden/ref_lotd_cfg.json, made by tests/golden/make_golden_cfg.py from
nr3d_lib/models/grid_encodings/lotd/lotd_cfg.py).  Integer ladders: exact equality."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "ref_lotd_cfg.json")))


@pytest.mark.parametrize("case", CASES, ids=[f"{i}-{c['args']['type']}" for i, c in enumerate(CASES)])
def test_get_lotd_cfg_matches_reference(case):
    from nr3d_lib_amd.models.grid_encodings.lotd import get_lotd_cfg
    got = get_lotd_cfg(**case["args"])
    want = case["result"]
    assert set(got) == set(want)
    for k in want:
        g = got[k]
        g = g.tolist() if hasattr(g, "tolist") else g
        assert json.loads(json.dumps(g, default=int)) == want[k], (k, g, want[k])


def test_generated_ladders_build_metas():
    """every generated 3-D ladder is accepted by the meta builder of the HIP library (host side, no GPU)"""
    from nr3d_lib_amd.bindings import _lotd
    for c in CASES:
        a, r = c["args"], c["result"]
        if a["type"] == "ngp4d" or a["input_ch"] != 3:
            continue
        if len(r["lod_res"]) > 32:                   # the extension's level limit (lotd_torch_api.cu: same message)
            with pytest.raises(RuntimeError, match="exceeds maximum level"):
                _lotd.LoDMeta(3, r["lod_res"], r["lod_n_feats"], r["lod_types"], r.get("hashmap_size"))
            continue
        m = _lotd.LoDMeta(3, r["lod_res"], r["lod_n_feats"], r["lod_types"], r.get("hashmap_size"))
        assert m.n_levels == len(r["lod_res"]) and m.n_params > 0


def test_unknown_and_deprecated_types():
    from nr3d_lib_amd.models.grid_encodings.lotd import get_lotd_cfg
    with pytest.raises(RuntimeError, match="Invalid type"):
        get_lotd_cfg("nope")
    with pytest.raises(RuntimeError, match="deprecated"):
        get_lotd_cfg("lotd", stretch=1.0, target_num_params=1 << 20)
