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


def test_c1_on_the_whole_frame():
    """BASELINE.json configs[0] on the real frame: data/000.tif (960 scanlines x 540 columns, decoded: a fixture), raw
    float values in, the constructor's normalisation by the image's maximum (dc.hpp:442-477), 9 identical views, 64
    hypotheses in [-2, 5.875].  223 271 of 518 400 pixels pass the edge test (SURVEY.md 8c's anchor) and are scanned;
    every plane equals the oracle's committed planes bit for bit (C_d within 1e-5)."""
    import json
    import os
    from remotesensingproject_amd import depth as rs
    from tests.test_oracle import GOLD
    img = np.load(os.path.join(GOLD, "c1_000tif_960x540_f32.npz"))["image"]
    z = np.load(os.path.join(GOLD, "c1full.npz"))
    meta = json.loads(str(z["meta"]))
    assert img.shape == (960, 540) and img.dtype == np.float32
    epis = [np.ascontiguousarray(np.repeat(img[v][None, :], meta["views"], axis=0)) for v in range(img.shape[0])]   # Vec<Mat>, each 9 x 540
    comp = rs.Depth1DComputer_pile(epis, meta["dmin"], meta["dmax"], meta["D"])     # default scale: the maximum over all EPIs
    comp.run()
    got = comp.results()
    assert comp.stats.pixels_scanned == 223271 and comp.stats.units == 223271 * 64
    assert int((got["edge_mask"] > 0).sum()) == 223271
    for k in KEYS:
        if k == "disp_confidence":
            assert np.abs(got[k] - z[k]).max() <= 1e-5, k
        else:
            assert np.array_equal(got[k], z[k]), k
    m = z["depth_idx"] >= 0
    assert (got["score"][m] == 1.0).all()   # identical views: disparity 0 (index 16) explains every pixel


def _check_against(got, want, label):
    for k in KEYS:
        if k == "disp_confidence":
            assert np.abs(got[k] - want[k]).max() <= 1e-5, (label, k)
        else:
            assert np.array_equal(got[k], want[k]), (label, k)


def test_c3_shape_noise_field_on_the_row_and_pixel_per_wave_kernels(hooks):
    """16 scanlines of NOISE at the c3 shape -- 101 views, 256 hypotheses in [-2, 5.96875], 1920 columns -- against the
    oracle's committed planes: k2_scan_reg<104,1> (dense row tiles, the bench kernel) and k2_scan_reg_px<104,1> (the
    sweep's sparse visits), bit for bit.  (The known-answer fields of the full-size tests score exactly 1.0 on the
    true line; here every hypothesis competes.)"""
    from remotesensingproject_amd import depth as rs
    vol, meta, want = load_case("c3noise")
    comp = rs.Depth1DComputer_pile(vol, meta["dmin"], meta["dmax"], meta["D"], meta["s_hat"], epi_scale_factor=1.0)
    comp.run()
    assert comp.stats.scan_kernel == 1 and comp.stats.s_pad == 104
    _check_against(comp.results(), want, "k2_scan_reg<104,1>")
    hooks(force_packed=1)
    hooks(px=1)
    comp.run()
    assert comp.stats.scan_kernel == 4 and comp.stats.s_pad == 104
    _check_against(comp.results(), want, "k2_scan_reg_px<104,1>")


def test_c5_shape_noise_field_on_the_chip_and_streaming_kernels(hooks):
    """2 scanlines of RGB noise at the c5 shape -- 201 views, 512 hypotheses in [-2, 5.984375], 4096 columns -- against
    the oracle's committed planes: k2_scan_chip (the c5 bench kernel), k2_scan_stream<3> and k2_scan_stream_px<3>."""
    from remotesensingproject_amd import depth as rs
    vol, meta, want = load_case("c5noise")
    comp = rs.Depth1DComputer_pile(vol, meta["dmin"], meta["dmax"], meta["D"], meta["s_hat"], epi_scale_factor=1.0)
    comp.run()
    assert comp.stats.scan_kernel == 3
    _check_against(comp.results(), want, "k2_scan_chip")
    hooks(force_scan="stream")
    comp.run()
    assert comp.stats.scan_kernel == 2
    _check_against(comp.results(), want, "k2_scan_stream<3>")
    hooks(force_packed=1)
    hooks(px=1)
    comp.run()
    assert comp.stats.scan_kernel == 5
    _check_against(comp.results(), want, "k2_scan_stream_px<3>")
