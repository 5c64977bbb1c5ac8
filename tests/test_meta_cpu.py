import numpy as np
"""CPU: the C-ABI meta builder (nr3d_lotd_meta_create) against the oracle's restatement of
LoDMeta::create_meta (lotd_torch_api.cu:29-230) -- integer fields must match exactly."""
import pytest

from util import LOTD_CASES


@pytest.mark.parametrize("case", list(LOTD_CASES))
def test_meta_matches_oracle(oracle, hiplib, case):
    from nr3d_lib_amd.bindings import _lotd
    D, res, nf, types, T, smooth = LOTD_CASES[case]
    want = oracle.lotd_create_meta(D, res, nf, types, T, smooth).as_dict()
    m = _lotd.LoDMeta(D, res, nf, types, T, smooth)
    for k in ("level_res_multidim", "level_n_feats", "level_types", "level_n_params", "level_offsets", "level_sizes",
              "map_levels", "map_cnt", "n_levels", "n_pseudo_levels", "n_feat_per_pseudo_lvl", "n_dims_to_encode",
              "n_encoded_dims", "n_params"):
        assert getattr(m, k) == want[k], k


def test_regrouped_metas_cover_every_level_once(hiplib):
    """nr3d_lotd_meta_regroup: a mixed-width meta (configs[3]: n_feats 4,4,8,4,2,16,8,4 -> global gcd 2, 25 pseudo levels) as
    three metas of width 8 / 4 / 2 whose pseudo levels keep the ORIGINAL output columns and cover every feature once"""
    from nr3d_lib_amd.bindings import _lotd
    res = [[32, 24, 16], [64, 48, 32], [128, 96, 64], [256, 192, 128], [512, 384, 256], [1024, 768, 512], [2048, 1536, 1024],
           [4096, 3072, 2048]]
    feats = [4, 4, 8, 4, 2, 16, 8, 4]
    m = _lotd.LoDMeta(3, res, feats, ["Dense", "Dense", "VM", "VM", "VM", "CP", "CP", "CP"], None)
    assert m.n_feat_per_pseudo_lvl == 2 and m.n_pseudo_levels == 25
    assert [int(m._c.map_col[q]) for q in range(25)] == [2 * q for q in range(25)]
    assert m._groups is not None and [int(g.n_feat_per_pseudo_lvl) for g in m._groups] == [8, 4, 2]
    starts = np.cumsum([0] + feats)
    seen = np.zeros(sum(feats), int)
    expect = {8: [(2, 0), (5, 0), (5, 1), (6, 0)], 4: [(0, 0), (1, 0), (3, 0), (7, 0)], 2: [(4, 0)]}
    for g in m._groups:
        G = int(g.n_feat_per_pseudo_lvl)
        got = [(int(g.map_levels[q]), int(g.map_cnt[q])) for q in range(g.n_pseudo_levels)]
        assert got == expect[G]
        for q, (lv, cnt) in enumerate(got):
            col = int(g.map_col[q])
            assert col == starts[lv] + cnt * G
            seen[col:col + G] += 1
        assert int(g.n_params) == m.n_params and int(g.n_encoded_dims) == m.n_encoded_dims and int(g.n_levels) == m.n_levels
        assert [int(g.levels[l].offset) for l in range(8)] == m.level_offsets[:8]
    assert (seen == 1).all()
    # nothing to regroup: equal widths, or Dense / Hash only (the 2-feature pair kernels serve those)
    assert _lotd.LoDMeta(3, [8, 9], [4, 4], ["VM", "CP"], None)._groups is None
    assert _lotd.LoDMeta(3, [8, 9], [2, 8], ["Dense", "Hash"], 1 << 10)._groups is None
    assert _lotd.LoDMeta(3, [8, 9], [2, 8], ["Dense", "CP"], None)._groups is not None
