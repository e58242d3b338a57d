"""The on-chip scan kernel (k2_scan_chip, remotesensingproject_amd/csrc/k2_chip.hpp): RGB light fields of 123 to 220
views -- BASELINE.json configs[4]'s 201 is the top rung of its ladder -- with all but three samples of a unit held in VGPRs +
AGPRs + LDS at one wave per SIMD, three fetched ahead on every pass, and the mean-shift passes in packed fp32; a volume
runs on the smallest rung that holds its views, the missing ones padded (round 4).  Bit-exact against the CPU oracle (core.hpp:480-661 restated) on small fields with
border and interior hypotheses, ragged rows, hypothesis groups and views beyond the tiers; bit-exact against the
streaming kernel at c5's real row length and hypothesis count."""
import numpy as np
import pytest

from tests.util import assert_pile_parity

pytestmark = pytest.mark.gpu

PLANES = ("edge_mask", "depth_idx", "edge_confidence", "score", "rbar", "depth_raw", "depth", "disp_confidence")


@pytest.fixture(scope="module")
def rs():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from remotesensingproject_amd import depth
    return depth


def _run(rs, vol, dmin, dmax, D, **debug):
    ctx = rs.default_context(0)
    ctx.set_debug(force_scan=0, force_groups=0, force_packed=-1)
    if debug:
        ctx.set_debug(**debug)
    try:
        comp = rs.Depth1DComputer_pile(vol, dmin, dmax, D, epi_scale_factor=1.0)
        comp.run()
        return comp.results(), comp.stats
    finally:
        ctx.set_debug(force_scan=0, force_groups=0, force_packed=-1)


@pytest.mark.parametrize("U,V,S,D,dmin,dmax,groups", [
    (200, 3, 201, 12, -0.3, 0.3, 0),     # tile 0 and the last tile are border, the middle one interior for every hypothesis
    (70, 2, 202, 9, -1.0, 1.0, 0),       # one view in the ragged tail; all border
    (131, 3, 220, 16, -0.25, 0.5, 0),    # 19 views beyond what the chip holds (fetched per pass, in pairs, the odd one last), a ragged last tile
    (140, 2, 207, 10, -0.5, 0.25, 0),    # six of them: a pair, a pair ahead of it, and one more trip
    (260, 2, 201, 24, -0.2, 0.2, 4),     # hypothesis groups: four workgroups share a tile, the last one merges
    (65, 1, 203, 8, 0.0, 0.0, 0),        # dmin == dmax
    # the ladder (round 4): rung = views in the AGPR tier / in the LDS tier; a volume is padded up to the next rung
    (200, 2, 127, 12, -0.3, 0.3, 0),     # <60, 0> exactly: no LDS tier, nothing padded
    (131, 2, 123, 10, -0.5, 0.6, 0),     # <60, 0>, four views missing (three of them the fetched-ahead slots... and one of the AGPR tier)
    (140, 2, 150, 9, -0.4, 0.4, 3),      # <84, 0>, one view missing (a fetched-ahead slot); three hypothesis groups (uneven slices)
    (200, 2, 152, 12, -0.3, 0.3, 0),     # <84, 8>: seven of the LDS tier's eight views missing
    (90, 3, 175, 8, -0.6, 0.3, 0),       # <84, 24> exactly
    (200, 2, 192, 10, -0.3, 0.3, 0),     # the top rung padded by nine views
    (70, 2, 200, 7, -1.0, 1.0, 0),       # ... by one, all border
])
def test_chip_kernel_against_the_oracle(rs, oracle_mod, U, V, S, D, dmin, dmax, groups):
    rng = np.random.default_rng(1000 + U + S)
    vol = rng.uniform(0.0, 1.0, size=(V, S, U, 3)).astype(np.float32)
    vol[:, :, U // 3: U // 3 + 5] *= np.float32(0.05)          # a dark band: the shadow cut leaves gaps in the pixel lists
    got, st = _run(rs, vol, dmin, dmax, D, **({"force_groups": groups} if groups else {}))
    assert st.scan_kernel == 3, "the on-chip kernel did not run (kernel %d)" % st.scan_kernel
    ref = oracle_mod.depth1d_pile_run(vol, dmin, dmax, D)
    assert_pile_parity(got, ref, label="chip %dx%dx%d" % (U, V, S))
    # ... and the streaming kernel, which the same shapes took until round 3, agrees with both
    other, st2 = _run(rs, vol, dmin, dmax, D, force_scan=2)
    assert st2.scan_kernel == 2
    for k in PLANES:
        assert np.array_equal(got[k], other[k]), k


@pytest.mark.parametrize("config,V", [("c5", 4), ("mansion_151", 6)])
def test_chip_kernel_on_the_synthetic_scene_matches_the_streaming_kernel(rs, config, V):
    """BASELINE.json configs[4]'s row length, view count and hypothesis grid on four scanlines (and the 151-view rung at the
    MansionLR row length on six): the known answer, and every plane bit-identical to the streaming kernel's (the oracle
    needs ~10 s per scanline at 512 hypotheses)."""
    from remotesensingproject_amd.synth import CONFIGS, make_lightfield
    from tests.test_gpu_fullsize import _check_known_answer
    c = dict(CONFIGS[config])
    vol, delta = make_lightfield(c["U"], V, c["S"], c["C"], seed=c["seed"], dmin=c["dmin"], dmax=c["dmax"], band=2)
    got, st = _run(rs, vol, c["dmin"], c["dmax"], c["D"])
    assert st.scan_kernel == 3
    if config == "c5":      # (its integer disparities are hypotheses of its grid; the 120-hypothesis grid over [0, 4] holds only 0 and 4)
        _check_known_answer(got, delta, dict(c, V=V))
    other, st2 = _run(rs, vol, c["dmin"], c["dmax"], c["D"], force_scan=2)
    assert st2.scan_kernel == 2 and st2.units == st.units
    for k in PLANES:
        assert np.array_equal(got[k], other[k]), k


def test_chip_kernel_leaves_other_launch_shapes_to_the_streaming_kernel(rs):
    """Per-pixel hypothesis ranges and packed pixel lists are not what the on-chip kernel is written for."""
    import torch
    rng = np.random.default_rng(4)
    U, V, S, D = 80, 2, 201, 8
    vol = rng.uniform(0.0, 1.0, size=(V, S, U, 3)).astype(np.float32)
    _, st = _run(rs, vol, -0.3, 0.3, D, force_packed=1)
    assert st.scan_kernel == 2
    _, st = _run(rs, vol[:, :200], -0.3, 0.3, D)          # fewer views than the top rung holds: padded (round 4)
    assert st.scan_kernel == 3
    _, st = _run(rs, vol[:, :122], -0.3, 0.3, D)          # fewer than the lowest rung pays for: two waves per SIMD are faster
    assert st.scan_kernel == 2
    _, st = _run(rs, vol[:, :123], -0.3, 0.3, D)
    assert st.scan_kernel == 3
    more = np.concatenate([vol, vol[:, :20]], axis=1)       # 221 views: beyond, the streaming kernel's tail is the cheaper one
    _, st = _run(rs, more, -0.3, 0.3, D)
    assert st.scan_kernel == 2
    _, st = _run(rs, more[:, :220], -0.3, 0.3, D)
    assert st.scan_kernel == 3
