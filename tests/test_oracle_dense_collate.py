"""CPU: the oracle's padded-batch builders (oracle/np_oracle.py, restating reference hodata/MaData.py:25-214) against the golden
vectors produced by running the reference (tests/golden/make_golden_dense.py)."""
import numpy as np

from conftest import load_golden
from oracle import np_oracle as O


# the reference constructs these MaskedTensors UNFILLED (padded slots keep the clamped gather's neighbours and its
# fill_masked(0) is a no-op because padvalue is already 0): the raw arrays are compared bit for bit, padding included
def test_to_dense_x_matches_reference():
    g = load_golden("dense_collate.npz")
    for tag in ("xf", "xi"):
        raw, mask = O.to_dense_x(g[tag], g["ptr"])
        assert np.array_equal(mask, g[tag + "_mask"])
        assert np.array_equal(raw, g[tag + "_raw"])
    raw, mask = O.to_dense_x(g["xf"], g["ptr"], 12)
    assert np.array_equal(mask, g["xf12_mask"]) and np.array_equal(raw, g["xf12_raw"])


def test_to_dense_tuplefeat_matches_reference():
    g = load_golden("dense_collate.npz")
    for tag in ("sq", "rect"):
        for kind in ("i", "v"):
            raw, mask = O.to_dense_tuplefeat(g[f"tf_{tag}_{kind}"], g[f"tf_{tag}_shape"], g[f"tf_{tag}_ptr"])
            assert np.array_equal(mask, g[f"tf_{tag}_{kind}_mask"])
            assert np.array_equal(raw, g[f"tf_{tag}_{kind}_raw"])
    raw, mask = O.to_dense_tuplefeat(g["tf3"], g["tf3_shape"], g["tf3_ptr"])
    assert np.array_equal(mask, g["tf3_mask"]) and np.array_equal(raw, g["tf3_raw"])


def test_to_dense_and_sparse_adj_match_reference():
    g = load_golden("dense_collate.npz")
    n, b = int(np.diff(g["ptr"]).max()), len(g["ptr"]) - 1
    for tag, attr, fill in (("ea", g["adj_ea"], 0.0), ("eai", g["adj_eai"], 0), ("ea_m1", g["adj_ea"], -1.0), ("ones", None, 0)):
        data, mask = O.to_dense_adj(g["adj_ei"], g["adj_eb"], attr, n, b, fill)
        assert np.array_equal(mask, g[f"adj_{tag}_mask"]) and np.array_equal(data, g[f"adj_{tag}_data"])
    ind, val, shape = O.to_sparse_adj(g["adj_ei"], g["adj_eb"], g["adj_ea"], n, b)
    assert np.array_equal(ind, g["spadj_ind"]) and np.array_equal(val, g["spadj_val"])
    assert list(shape) == list(g["spadj_shape"])
