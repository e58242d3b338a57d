"""The C++11 host wrapper (include/rslf_hip.hpp), compiled with g++ against librslf_hip.so and
run on the GPU, vs the oracle on the same inputs."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_wrapper_matches_oracle(tmp_path, oracle_mod):
    from remotesensingproject_amd import build as hb
    so = hb.build()
    exe = str(tmp_path / "test_host_wrapper")
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_host_wrapper.cpp"), "-o", exe,
                    "-L", os.path.dirname(so), "-lrslf_hip", "-Wl,-rpath," + os.path.dirname(so)], check=True)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    V, S, U, D = 7, 13, 150, 20
    for tag, C_, u8 in (("f32_1ch", 1, False), ("f32_3ch", 3, False), ("u8_3ch", 3, True)):
        raw = np.fromfile(tmp_path / (tag + "_input.f32"), np.float32).reshape(V, S, U, C_)
        if u8:
            vol = oracle_mod.normalize_u8(raw.astype(np.uint8))
        else:
            vol, _ = oracle_mod.normalize_f32(raw, -1.0)       # ctor: scale by the max over all EPIs
        ref = oracle_mod.depth1d_pile_run(vol, -1.5, 2.0, D)
        rd = lambda name, dt: np.fromfile(tmp_path / ("%s_%s" % (tag, name)), dt)
        assert np.array_equal(rd("mask.u8", np.uint8).reshape(V, U), ref.edge_mask), tag
        assert np.array_equal(rd("idx.i32", np.int32).reshape(V, U), ref.depth_idx), tag
        assert np.array_equal(rd("Ce.f32", np.float32).reshape(V, U), ref.edge_confidence), tag
        assert np.array_equal(rd("score.f32", np.float32).reshape(V, U), ref.score), tag
        assert np.array_equal(rd("depth.f32", np.float32).reshape(V, U), ref.depth), tag
        assert np.array_equal(rd("rbar.f32", np.float32).reshape(V, U, C_), ref.rbar), tag
        assert np.abs(rd("Cd.f32", np.float32).reshape(V, U) - ref.disp_confidence).max() <= 1e-5, tag
    # Depth2DComputer and FineToCoarse through the C++ classes (host pointers in, host planes out)
    V, S, U, D = 44, 5, 64, 9
    raw = np.fromfile(tmp_path / "sweep_input.f32", np.float32).reshape(V, S, U, 1)
    norm, _ = oracle_mod.normalize_f32(raw, -1.0)
    ref2 = oracle_mod.depth2d_run(norm, -1.0, 1.0, D)
    assert np.array_equal(np.fromfile(tmp_path / "d2_mask.u8", np.uint8).reshape(S, V, U), ref2.edge_mask)
    assert np.array_equal(np.fromfile(tmp_path / "d2_Ce.f32", np.float32).reshape(S, V, U), ref2.edge_confidence)
    assert np.array_equal(np.fromfile(tmp_path / "d2_depth.f32", np.float32).reshape(S, V, U), ref2.depth)
    reff = oracle_mod.fine_to_coarse_run(raw, -1.0, 1.0, D)
    assert reff["dims"] == [(44, 64), (22, 32), (11, 16)]
    assert np.array_equal(np.fromfile(tmp_path / "f2c_map.f32", np.float32).reshape(S, V, U), reff["fused_map"])
    assert np.array_equal(np.fromfile(tmp_path / "f2c_valid.u8", np.uint8).reshape(S, V, U), reff["fused_valid"])
