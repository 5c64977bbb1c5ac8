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
