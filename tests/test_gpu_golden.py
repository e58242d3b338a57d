"""HIP path vs the committed golden fixtures (tests/golden/, made by make_golden.py)."""
import numpy as np
import pytest

from tests.test_oracle import KEYS, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c1crop", "rand1", "rgb", "edge"])
def test_hip_reproduces_golden(name):
    from remotesensingproject_amd import depth as rs
    vol, meta, want = load_case(name)
    comp = rs.Depth1DComputer_pile(vol, meta["dmin"], meta["dmax"], meta["D"], meta["s_hat"], epi_scale_factor=1.0)
    comp.run()
    got = comp.results()
    for k in KEYS:
        if k == "disp_confidence":   # double arithmetic with a free summation order: 1e-5 (north_star)
            assert np.abs(got[k] - want[k]).max() <= 1e-5, (name, k)
        else:
            assert np.array_equal(got[k], want[k]), (name, k)


def test_c1_config_from_raw_crop():
    """BASELINE.json configs[0] plumbing: raw float32 TIFF values in, constructor normalisation by the
    given max (dc.hpp:474), 9 replicated views, 64 hypotheses."""
    import json
    import os
    from remotesensingproject_amd import depth as rs
    from tests.test_oracle import GOLD
    crop = np.load(os.path.join(GOLD, "c1_crop_000tif_rows400_424.npy"))
    z = np.load(os.path.join(GOLD, "c1crop.npz"))
    meta = json.loads(str(z["meta"]))
    epis = [np.ascontiguousarray(np.repeat(crop[v][None, :], 9, axis=0)) for v in range(crop.shape[0])]   # Vec<Mat>, each 9 x 540
    comp = rs.Depth1DComputer_pile(epis, meta["dmin"], meta["dmax"], meta["D"], epi_scale_factor=meta["tif_max"])
    comp.run()
    got = comp.results()
    assert np.array_equal(got["depth_idx"], z["depth_idx"])
    assert np.array_equal(got["edge_mask"], z["edge_mask"])
    assert np.array_equal(got["depth"], z["depth"])
    # identical views: wherever a pixel is scanned, disparity 0 scores 1.0 (index 16 of the [-2, 5.875] grid)
    m = z["depth_idx"] >= 0
    assert (got["score"][m] == 1.0).all()
