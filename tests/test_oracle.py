"""CPU tests of the oracle itself (no GPU): golden fixtures, the independent numpy
restatement, the analytic known-answer case, the c1 anchor, edge cases.

PARITY UNPINNED: the reference holds no golden vectors for this path; the goldens
are our oracle's own outputs (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "depth_idx", "score", "depth_raw")


# Noise fields at the BASELINE shapes (view count, hypothesis count and grid of configs[2] / configs[4]), a few scanlines
# each: name -> (V, S, U, C, D, dmin, dmax, seed).  Inputs come from the seed; the oracle's planes are committed.
NOISE_CASES = {
    "c3noise": (16, 101, 1920, 1, 256, -2.0, 5.96875, 20260403),
    "c5noise": (2, 201, 4096, 3, 512, -2.0, 5.984375, 20260405),
}


def noise_volume(name):
    V, S, U, C, _, _, _, seed = NOISE_CASES[name]
    return np.random.default_rng(seed).uniform(0.0, 1.0, size=(V, S, U, C)).astype(np.float32)


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    if name in NOISE_CASES:
        vol = noise_volume(name)
    elif name == "c1crop":
        import oracle
        crop = np.load(os.path.join(GOLD, "c1_crop_000tif_rows400_424.npy"))
        norm, _ = oracle.normalize_f32(crop, meta["tif_max"])
        vol = np.ascontiguousarray(np.repeat(norm[:, None, :, None], meta["views"], axis=1))
    elif name == "c1full":   # BASELINE.json configs[0]: the whole 960 x 540 frame of data/000.tif in 9 identical views
        import oracle
        img = np.load(os.path.join(GOLD, "c1_000tif_960x540_f32.npz"))["image"]
        norm, _ = oracle.normalize_f32(np.ascontiguousarray(img), meta["tif_max"])
        vol = np.ascontiguousarray(np.repeat(norm[:, None, :, None], meta["views"], axis=1))
    else:
        vol = np.load(os.path.join(GOLD, name + "_input.npy"))
    return vol, meta, {k: z[k] for k in KEYS}


@pytest.mark.parametrize("name", ["c1crop", "rand1", "rgb", "edge", "c1full", "c3noise", "c5noise"])
def test_oracle_reproduces_golden(oracle_mod, name):
    vol, meta, want = load_case(name)
    r = oracle_mod.depth1d_pile_run(vol, meta["dmin"], meta["dmax"], meta["D"], meta["s_hat"])
    for k in KEYS:
        assert np.array_equal(getattr(r, k), want[k]), (name, k)


@pytest.mark.parametrize("name,rows", [("rand1", slice(4, 8)), ("rgb", slice(2, 6)), ("edge", slice(0, 6)), ("c1crop", slice(10, 11))])
def test_numpy_restatement_agrees_bitwise(oracle_mod, name, rows):
    """Two independently written restatements (C, loop-restructured; numpy, matrix passes as
    in the reference) must agree bit for bit.  The median needs the neighbouring rows, so both
    sides run on the same row block."""
    from oracle import oracle_np as onp
    vol, meta, _ = load_case(name)
    sub = np.ascontiguousarray(vol[rows])
    r = oracle_mod.depth1d_pile_run(sub, meta["dmin"], meta["dmax"], meta["D"], meta["s_hat"])
    n = onp.depth1d_pile_run(sub, np.float32(meta["dmin"]), np.float32(meta["dmax"]), meta["D"], meta["s_hat"])
    for ko, kn in (("edge_confidence", "Ce"), ("edge_mask", "Ce_mask"), ("depth_idx", "idx"), ("score", "score"),
                   ("depth_raw", "depth_raw"), ("rbar", "rbar"), ("disp_confidence", "Cd"), ("depth", "depth")):
        assert np.array_equal(getattr(r, ko), n[kn]), (name, ko)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("name,rows", [("rand1", slice(4, 7)), ("rgb", slice(2, 5)), ("edge", slice(0, 4))])
def test_nearest_interpolation_restatements_agree(oracle_mod, name, rows, mode):
    """par_interpolation_class = Interpolation1DNearestNeighbour (core.hpp:77, interp.hpp:94-131): the C and numpy
    restatements agree bit for bit, for the class as stated (std::round) and as built (index = bits of x)."""
    from oracle import oracle_np as onp
    vol, meta, _ = load_case(name)
    sub = np.ascontiguousarray(vol[rows])
    pc = oracle_mod.default_params()
    pc.interpolation = mode
    pn = onp.default_params()
    pn["interpolation"] = mode
    r = oracle_mod.depth1d_pile_run(sub, meta["dmin"], meta["dmax"], meta["D"], meta["s_hat"], params=pc)
    n = onp.depth1d_pile_run(sub, np.float32(meta["dmin"]), np.float32(meta["dmax"]), meta["D"], meta["s_hat"], p=pn)
    for ko, kn in (("edge_confidence", "Ce"), ("edge_mask", "Ce_mask"), ("depth_idx", "idx"), ("score", "score"),
                   ("depth_raw", "depth_raw"), ("rbar", "rbar"), ("disp_confidence", "Cd"), ("depth", "depth")):
        assert np.array_equal(getattr(r, ko), n[kn]), (name, mode, ko)
    if mode == 2:
        # as built, a sample exists only where the position is exactly +0 (its bit pattern is column 0): R[s_hat][u]
        # is NaN for every u != 0, and a pixel survives only if some view's line passes through x = 0 exactly
        assert (r.edge_mask > 0).mean() < 0.05
        assert set(np.unique(r.score)) <= {np.float32(0.0), np.float32(1.0)}


def test_nearest_interpolation_known_answer(oracle_mod):
    """Integer disparities and integer view offsets: nearest and linear sampling read the same columns, so
    the stated nearest mode must reproduce the linear result wherever the whole line is in range."""
    from remotesensingproject_amd.synth import make_lightfield
    U, V, S, D = 96, 4, 9, 9
    deltas = np.array([-1, 0, 1, 2], np.float32)
    vol, _ = make_lightfield(U, V, S, 1, seed=9, deltas=deltas)
    lin = oracle_mod.depth1d_pile_run(vol, -2.0, 2.0, D)            # grid step 0.5
    p = oracle_mod.default_params()
    p.interpolation = 1
    nn = oracle_mod.depth1d_pile_run(vol, -2.0, 2.0, D, params=p)
    inner = slice(12, U - 12)
    m = lin.edge_mask[:, inner] > 0
    assert m.mean() > 0.9
    want = ((deltas + 2.0) / 0.5).astype(np.int32)
    for v in range(V):
        assert (nn.depth_idx[v, inner][m[v]] == want[v]).all()
        assert (nn.score[v, inner][m[v]] == 1.0).all()


DEGENERATE = [(1, 1, 1, 1, 2, -1.0, 1.0),     # one pixel, one view, the smallest hypothesis grid (core.hpp:548 divides by D-1)
              (1, 2, 2, 1, 2, -1.0, 1.0),
              (2, 3, 3, 3, 3, 0.0, 0.0),      # dmin == dmax: every hypothesis the same line, first index wins
              (1, 9, 5, 1, 4, -1.0, 1.0),     # U below the 9-tap edge filter: BORDER_REFLECT_101 folds more than once
              (3, 1, 40, 1, 5, -2.0, 2.0),    # a single view: R = r-bar, score 1 for every hypothesis
              (1, 5, 9, 1, 2, -1.0, 1.0),
              (2, 4, 17, 3, 7, 1.0, 1.0)]


@pytest.mark.parametrize("shape", DEGENERATE, ids=["%dx%dx%dx%d_D%d" % s[:5] for s in DEGENERATE])
def test_degenerate_shapes_restatements_agree(oracle_mod, shape):
    from oracle import oracle_np as onp
    V, S, U, C_, D, dmin, dmax = shape
    vol = np.random.default_rng(3 + U).uniform(0.0, 1.0, size=(V, S, U, C_)).astype(np.float32)
    r = oracle_mod.depth1d_pile_run(vol, dmin, dmax, D)
    n = onp.depth1d_pile_run(vol, np.float32(dmin), np.float32(dmax), D)
    for ko, kn in (("edge_confidence", "Ce"), ("edge_mask", "Ce_mask"), ("depth_idx", "idx"), ("score", "score"),
                   ("depth_raw", "depth_raw"), ("rbar", "rbar"), ("disp_confidence", "Cd"), ("depth", "depth")):
        assert np.array_equal(getattr(r, ko), n[kn]), (shape, ko)
    if dmin == dmax:
        assert set(np.unique(r.depth_idx)) <= {-1, 0}     # ties: cv::minMaxLoc keeps the first maximum


def test_structuring_elements_and_opening(oracle_mod):
    """cv::getStructuringElement as OpenCV 3.x builds it (the 5x5 ellipse is the documented
    [[0,0,1,0,0],[1,1,1,1,1],[1,1,1,1,1],[1,1,1,1,1],[0,0,1,0,0]]), and the opening in both restatements."""
    import ctypes as C
    from oracle import oracle_np as onp
    L = oracle_mod.lib()
    el = np.zeros(25, np.uint8)
    L.oracle_structuring_element(2, 5, el.ctypes.data_as(C.c_void_p))
    want = np.array([[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]], np.uint8)
    assert np.array_equal(el.reshape(5, 5), want)
    assert np.array_equal(onp.structuring_element(2, 5).astype(np.uint8), want)
    el3 = np.zeros(9, np.uint8)
    L.oracle_structuring_element(1, 3, el3.ctypes.data_as(C.c_void_p))
    assert np.array_equal(el3.reshape(3, 3), np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8))
    rng = np.random.default_rng(5)
    for shape in (0, 1, 2):
        for k in (2, 3, 4, 5, 7, 9):
            m = (rng.uniform(size=(17, 40)) < 0.7).astype(np.uint8) * 255
            c = m.copy()
            L.oracle_morph_open(c.ctypes.data_as(C.c_void_p), 17, 40, shape, k)
            n = onp.morph_open(m, shape, k)
            assert np.array_equal(c, n), (shape, k)
            c2 = c.copy()
            L.oracle_morph_open(c2.ctypes.data_as(C.c_void_p), 17, 40, shape, k)
            if k % 2 == 1:                                  # symmetric element about its anchor:
                assert (c <= m).all()                       #   an opening never adds pixels
                assert np.array_equal(c2, c), (shape, k)    #   and is idempotent
            # even sizes: the anchor k/2 is off centre and erosion and dilation use the same offsets (OpenCV's
            # MorphFilter does not reflect the element), so the "opening" also shifts by one pixel


@pytest.mark.parametrize("shape,k", [(2, 3), (0, 4), (1, 5)])
def test_opening_in_the_pile_restatements_agree(oracle_mod, shape, k):
    from oracle import oracle_np as onp
    rng = np.random.default_rng(8 + k)
    V, S, U = 10, 5, 60
    blobs = np.zeros((V, U), bool)                               # textured segments on a flat (C_e = 0) ground
    for v in range(V):
        for a in rng.integers(0, U - 10, size=2):
            blobs[v, a:a + int(rng.integers(1, 10))] = True
    blobs[2:V - 3, 8:30] = True                                   # and one block large enough to survive any element here
    tex = rng.uniform(0.2, 1.0, size=(V, S, U, 1)).astype(np.float32)
    vol = np.where(blobs[:, None, :, None], tex, np.float32(0.5)).astype(np.float32)
    pc = oracle_mod.default_params()
    pc.edge_confidence_opening_type, pc.edge_confidence_opening_size = shape, k
    pn = onp.default_params()
    pn["edge_confidence_opening_type"], pn["edge_confidence_opening_size"] = shape, k
    r = oracle_mod.depth1d_pile_run(vol, -1.0, 1.0, 7, params=pc)
    n = onp.depth1d_pile_run(vol, np.float32(-1.0), np.float32(1.0), 7, p=pn)
    for ko, kn in (("edge_confidence", "Ce"), ("edge_mask", "Ce_mask"), ("depth_idx", "idx"), ("score", "score"),
                   ("depth_raw", "depth_raw"), ("rbar", "rbar"), ("disp_confidence", "Cd"), ("depth", "depth")):
        assert np.array_equal(getattr(r, ko), n[kn]), (shape, k, ko)


def test_analytic_known_answer(oracle_mod):
    """Independent of every OpenCV-semantics assumption: integer true disparity on the grid =>
    all samples of the true line are identical => K == 1, score == 1.0 exactly, argmax known."""
    from remotesensingproject_amd.synth import make_lightfield
    U, V, S, D = 128, 8, 17, 33
    dmin, dmax = -2.0, 2.0
    deltas = np.array([-2, -1, 0, 1, 2, 1, 0, -1], np.float32)
    vol, _ = make_lightfield(U, V, S, 1, seed=5, deltas=deltas)
    r = oracle_mod.depth1d_pile_run(vol, dmin, dmax, D)
    want_idx = ((deltas - dmin) / ((dmax - dmin) / (D - 1))).astype(np.int32)
    m = r.edge_mask > 0
    assert m.mean() > 0.99
    for v in range(V):
        assert (r.depth_idx[v][m[v]] == want_idx[v]).all()
        assert (r.score[v][m[v]] == 1.0).all()
        assert (r.depth_raw[v][m[v]] == deltas[v]).all()


def test_c1_anchor_recorded():
    """after x * float(1/262.72208), 223 271 of 518 400 pixels of data/000.tif pass C_e > 0.02."""
    a = json.load(open(os.path.join(GOLD, "c1_anchor.json")))
    assert a["tif_shape"] == [960, 540]
    assert a["mask_count"] == 223271
    assert abs(a["tif_max"] - 262.72208) < 1e-4


@pytest.mark.skipif(not os.path.exists("/root/reference/data/000.tif"), reason="reference data file not present on this box")
def test_c1_anchor_recomputed(oracle_mod):
    from PIL import Image
    img = np.array(Image.open("/root/reference/data/000.tif"), dtype=np.float32)
    norm, scale = oracle_mod.normalize_f32(img[:, None, :, None], -1.0)
    _, m = oracle_mod.edge_confidence_pile(np.ascontiguousarray(norm), 0)
    assert int((m > 0).sum()) == 223271
    crop = np.load(os.path.join(GOLD, "c1_crop_000tif_rows400_424.npy"))
    assert np.array_equal(crop, img[400:424])


def test_normalisation(oracle_mod):
    u8 = np.arange(256, dtype=np.uint8)
    got = oracle_mod.normalize_u8(u8)
    assert np.array_equal(got, u8.astype(np.float32) * np.float32(1.0 / 255.0))       # dc.hpp:470
    x = np.array([1.0, 10.0, 262.72208], np.float32)
    got, s = oracle_mod.normalize_f32(x, -1.0)
    assert s == x.max()
    assert np.array_equal(got, x * np.float32(1.0 / np.float64(x.max())))              # dc.hpp:474


def test_border_pixels_have_partial_cardinality(oracle_mod):
    """u = 0 and u = U-1: half of every sloped line leaves the EPI, card_R < S, scores stay in [0,1]."""
    rng = np.random.default_rng(0)
    vol = rng.uniform(0.2, 1.0, size=(2, 9, 32, 1)).astype(np.float32)
    r = oracle_mod.depth1d_pile_run(vol, -2.0, 2.0, 9)
    assert (r.score >= 0).all() and (r.score <= 1.0).all()
    assert (r.depth_idx[:, 0] >= 0).all() and (r.depth_idx[:, -1] >= 0).all()


def test_first_maximum_wins_on_ties(oracle_mod):
    rng = np.random.default_rng(1)
    vol = rng.uniform(0.2, 1.0, size=(3, 9, 40, 1)).astype(np.float32)
    r = oracle_mod.depth1d_pile_run(vol, 0.5, 0.5, 7)     # dmin == dmax: all hypotheses identical
    assert (r.depth_idx[r.edge_mask > 0] == 0).all()


def test_scan_mask_is_anded_in_place(oracle_mod):
    rng = np.random.default_rng(2)
    vol = rng.uniform(0.0, 1.0, size=(3, 9, 50, 1)).astype(np.float32)
    Ce, cm = oracle_mod.edge_confidence_pile(vol, 4)
    mask = (rng.uniform(size=cm.shape) > 0.5).astype(np.uint8) * 255
    e = oracle_mod.depth_epi(vol[1], np.full(50, -1, np.float32), np.full(50, 1, np.float32), 8, 4, Ce[1], cm[1], mask_u=mask[1])
    assert np.array_equal(e["mask"], cm[1] & mask[1])            # core.hpp:511
    assert (e["idx"][(cm[1] & mask[1]) == 0] == -1).all()


@pytest.mark.parametrize("C_", [1, 3])
def test_sweep2d_numpy_restatement_agrees_bitwise(oracle_mod, C_):
    """compute_2D_edge_confidence + compute_2D_depth_epi (core.hpp:901-1133): C oracle vs the numpy restatement."""
    from oracle import oracle_np as onp
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(36, 4, 7, C_, seed=3, deltas=np.array([0, 1, -1, 0.5], np.float32))
    rng = np.random.default_rng(5)
    vol[2:] = rng.uniform(0, 1, size=vol[2:].shape).astype(np.float32)
    r = oracle_mod.depth2d_run(vol, -1.0, 1.0, 9)
    n = onp.depth2d_run(vol, np.float32(-1.0), np.float32(1.0), 9)
    for a, k in (("edge_confidence", "Ce"), ("edge_mask", "Ce_mask"), ("disp_confidence", "Cd"), ("depth", "depth"),
                 ("rbar", "rbar"), ("scan_mask", "scan_mask")):
        assert np.array_equal(getattr(r, a), n[k]), a
    assert (r.scan_mask == 0).sum() > (r.edge_mask == 0).sum()   # propagation painted something


@pytest.mark.parametrize("thr", [0.01, 0.2])
def test_sweep2d_with_the_disp_confidence_gate(oracle_mod, thr):
    """The reference's build switch _USE_DISP_CONFIDENCE_SCORE (core.hpp:35, :1097-1098): a pixel paints along its
    line only if its C_d exceeds par_disp_score_threshold.  Both restatements agree, and the gate changes the result."""
    from oracle import oracle_np as onp
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(40, 4, 7, 1, seed=4, deltas=np.array([0, 1, -1, 1], np.float32))
    rng = np.random.default_rng(6)
    vol[2:] = (vol[2:] + rng.normal(0, 0.05, size=vol[2:].shape)).clip(0, 1).astype(np.float32)
    pc = oracle_mod.default_params()
    pc.use_disp_confidence_score, pc.disp_score_threshold = 1, thr
    pn = onp.default_params()
    pn["use_disp_confidence_score"], pn["disp_score_threshold"] = True, np.float32(thr)
    r = oracle_mod.depth2d_run(vol, -1.0, 1.0, 9, params=pc)
    n = onp.depth2d_run(vol, np.float32(-1.0), np.float32(1.0), 9, p=pn)
    for a, k in (("edge_confidence", "Ce"), ("edge_mask", "Ce_mask"), ("disp_confidence", "Cd"), ("depth", "depth"),
                 ("rbar", "rbar"), ("scan_mask", "scan_mask")):
        assert np.array_equal(getattr(r, a), n[k]), a
    if thr > 0.1:   # a threshold most confidences miss: fewer pixels paint than under the edge-mask gate
        plain = oracle_mod.depth2d_run(vol, -1.0, 1.0, 9)
        assert not np.array_equal(plain.scan_mask, r.scan_mask) or not np.array_equal(plain.depth, r.depth)


def test_sweep2d_first_visit_is_the_pile_scan(oracle_mod):
    """The centre view's scan inside the 2-D sweep sees the untouched edge mask: its C_e, r_bar and C_d at
    scanned pixels equal Depth1DComputer_pile's (the stored depth differs: raw + painted, core.hpp:892)."""
    rng = np.random.default_rng(8)
    vol = rng.uniform(0, 1, size=(3, 5, 60, 1)).astype(np.float32)
    one = oracle_mod.depth1d_pile_run(vol, -1.0, 1.0, 8)
    two = oracle_mod.depth2d_run(vol, -1.0, 1.0, 8)
    s_hat = 2
    assert np.array_equal(two.edge_confidence[s_hat], one.edge_confidence)
    assert np.array_equal(two.edge_mask[s_hat], one.edge_mask)
    assert np.array_equal(two.rbar[s_hat], one.rbar)


def test_f2c_primitives_numpy_restatement_agrees_bitwise(oracle_mod):
    """downsample_EPIs, bound tightening and fuse_disp_maps (rslf_fine_to_coarse*.{hpp,cpp}): the C oracle vs an
    array-style numpy restatement, odd and even sizes."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(1)
    for (V, S, U, C_) in ((17, 3, 23, 1), (16, 2, 30, 3), (11, 2, 13, 1), (35, 1, 27, 1)):
        raw = rng.uniform(0, 250, size=(V, S, U, C_)).astype(np.float32)
        assert np.array_equal(oracle_mod.downsample_epis(raw), onp.downsample_epis(raw))
    for (S, Vu, Uu) in ((2, 17, 23), (3, 16, 30)):
        dep = rng.uniform(-2, 2, size=(S, Vu, Uu)).astype(np.float32)
        m = (rng.uniform(size=(S, Vu, Uu)) > 0.7).astype(np.uint8) * 255
        Vd, Ud = int(np.rint(Vu * .5)), int(np.rint(Uu * .5))
        lo = np.full((S, Vd, Ud), -3, np.float32); hi = np.full((S, Vd, Ud), 3, np.float32)
        a = oracle_mod.f2c_tighten_bounds(dep, m, lo, hi); b = onp.tighten_bounds(dep, m, lo, hi)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for dims in (((17, 23), (8, 12), (4, 6)), ((16, 30), (8, 15)), ((35, 27), (18, 14), (9, 7))):
        d = [rng.uniform(-2, 2, size=x).astype(np.float32) for x in dims]
        m = [(rng.uniform(size=x) > 0.5).astype(np.uint8) * 255 for x in dims]
        a = oracle_mod.f2c_fuse(d, m); b = onp.fuse(d, m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_f2c_known_structure(oracle_mod):
    """Halving a constant image keeps the constant (the Gaussian table sums to 1, the 2x2 mean is exact);
    cvRound level sizes; a pyramid stops above _MIN_SPATIAL_DIM."""
    raw = np.full((13, 2, 18, 1), 7.0, np.float32)
    out = oracle_mod.downsample_epis(raw)
    assert out.shape == (6, 2, 9, 1)           # cvRound(6.5) = 6 (ties to even), cvRound(9.0) = 9
    assert np.all(out == 7.0)
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(48, 40, 3, 1, seed=1, dmin=-1, dmax=1, band=8)
    r = oracle_mod.fine_to_coarse_run((vol * 100).astype(np.float32), -1.0, 1.0, 5)
    assert r["dims"] == [(40, 48), (20, 24), (10, 12)] or r["dims"] == [(40, 48), (20, 24)]
    assert r["fused_map"].shape == (3, 40, 48)


def test_downsample_u8_c_and_numpy_restatements_agree(oracle_mod):
    """downsample_EPIs on CV_8U Mats (uchar arithmetic, fine_to_coarse_core.cpp:22-41): the C and the numpy restatement
    are written independently and must agree bit for bit; results are uchar levels and stay close to the float pyramid."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(8)
    for (V, S, U, C_) in [(12, 3, 15, 1), (11, 2, 14, 3), (7, 2, 7, 1), (16, 2, 16, 3), (13, 1, 22, 1)]:
        lev = rng.integers(0, 256, size=(V, S, U, C_)).astype(np.float32)
        c = oracle_mod.downsample_epis_u8(lev)
        n = onp.downsample_epis_u8(lev)
        assert np.array_equal(c, n), (V, S, U, C_)
        assert np.array_equal(c, np.rint(c)) and c.min() >= 0 and c.max() <= 255
        assert np.abs(c - oracle_mod.downsample_epis(lev)).max() < 1.0
    flat = np.full((12, 2, 12, 1), 77.0, np.float32)       # a constant image stays constant: the taps sum to 256
    assert np.all(oracle_mod.downsample_epis_u8(flat) == 77.0)


def test_selective_median_of_a_pixel_with_no_candidate_is_zero(oracle_mod):
    """A masked pixel whose centre radiance is NaN passes nobody's radiance test, not even its own: the reference then reads
    buffer[0] of a vector it has just cleared (core.hpp:713-714, undefined; in practice a stale value).  Defined as 0 in
    both restatements and in K3 (found by tools/fuzz_parity.py seed 8302 case 4537)."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(12)
    V, S, U = 6, 3, 9
    vol = rng.uniform(0.0, 1.0, size=(V, S, U, 1)).astype(np.float32)
    vol[2, 1, 4, 0] = np.nan
    src = rng.uniform(-1.0, 1.0, size=(V, U)).astype(np.float32)
    mask = np.full((V, U), 255, np.uint8)
    c = oracle_mod.selective_median(src, vol, 1, mask, 5, np.float32(10.0))
    n = onp.selective_median(src, vol, 1, mask, 5, np.float32(10.0))
    assert c[2, 4] == 0.0 and n[2, 4] == 0.0
    assert np.array_equal(c, n)
    assert np.count_nonzero(c) == V * U - 1      # every other pixel has candidates (and a NaN neighbour is never one)
