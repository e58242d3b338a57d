"""Multi-process scanline sharding on CPU: world_size 2 and 3 over gloo.

The per-rank compute is injected (the CPU oracle stands in for the HIP path: tests may use
the oracle, the product never does); what is under test is the partition, the recomputed
halo, the packed single-gather reassembly and its equality with the unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from remotesensingproject_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, V, S, U, C, D, out_path, use_gatherer):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from remotesensingproject_amd.synth import make_lightfield

        def local_compute(shard):
            # every rank regenerates the same scene and keeps only its rows (+halo), as bench.py does
            vol, _ = make_lightfield(U, V, S, C, seed=99, dmin=-1.0, dmax=2.0, band=3, rows=shard.rows)
            assert vol.shape[0] == shard.hi - shard.lo
            r = oracle.depth1d_pile_run(vol, -1.0, 2.0, D)
            return dict(edge_confidence=torch.from_numpy(r.edge_confidence), disp_confidence=torch.from_numpy(r.disp_confidence),
                        depth=torch.from_numpy(r.depth), depth_raw=torch.from_numpy(r.depth_raw), score=torch.from_numpy(r.score),
                        depth_idx=torch.from_numpy(r.depth_idx), rbar=torch.from_numpy(r.rbar), edge_mask=torch.from_numpy(r.edge_mask))

        if use_gatherer:
            shard = sharding.make_shard(V, rank, world, 5)
            out = sharding.PlaneGatherer(shard, U, C, "cpu")(local_compute(shard))
        else:
            out = sharding.run_sharded(local_compute, V, U, C, median_filter_size=5)
        if rank == 0:
            np.savez(out_path, **{k: v.numpy() for k, v in out.items()})
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,V,C,use_gatherer", [(2, 11, 1, False), (3, 10, 3, False), (2, 4, 1, False),
                                                    (2, 12, 1, True), (3, 11, 3, True)])
def test_sharded_equals_unsharded(tmp_path, oracle_mod, world, V, C, use_gatherer):
    from remotesensingproject_amd.synth import make_lightfield
    S, U, D = 9, 48, 12
    out_path = str(tmp_path / "stitched.npz")
    mp.spawn(_worker, args=(world, _free_port(), V, S, U, C, D, out_path, use_gatherer), nprocs=world, join=True)
    got = np.load(out_path)
    vol, _ = make_lightfield(U, V, S, C, seed=99, dmin=-1.0, dmax=2.0, band=3)
    ref = oracle_mod.depth1d_pile_run(vol, -1.0, 2.0, D)
    for k in ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "depth_idx", "score", "depth_raw"):
        assert np.array_equal(got[k], getattr(ref, k)), k


def test_partition_and_halo():
    parts = sharding.row_partition(1080, 8)
    assert parts[0] == (0, 135) and parts[-1] == (945, 1080) and all(b - a == 135 for a, b in parts)
    parts = sharding.row_partition(10, 3)
    assert parts == [(0, 4), (4, 7), (7, 10)]
    s = sharding.make_shard(1080, 3, 8, 5)
    assert (s.v0, s.v1, s.lo, s.hi) == (405, 540, 403, 542)
    assert s.interior == slice(2, 137)
    s0 = sharding.make_shard(1080, 0, 8, 5)
    assert (s0.lo, s0.hi) == (0, 137) and s0.interior == slice(0, 135)
    s7 = sharding.make_shard(1080, 7, 8, 5)
    assert (s7.lo, s7.hi) == (943, 1080)
    # more ranks than rows: empty blocks are legal
    parts = sharding.row_partition(2, 4)
    assert parts == [(0, 1), (1, 2), (2, 2), (2, 2)]
    # an opened edge mask widens the halo by twice the element radius (erosion, then dilation)
    assert sharding.halo_rows(5, 1) == 2 and sharding.halo_rows(5, 3) == 4 and sharding.halo_rows(5, 5) == 6
    assert sharding.halo_rows(7, 4) == 3 + 4 and sharding.halo_rows(1, 1) == 0
    so = sharding.make_shard(1080, 3, 8, 5, 5)
    assert (so.v0, so.v1, so.lo, so.hi) == (405, 540, 399, 546) and so.interior == slice(6, 141)


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    V, U, C = 7, 13, 3
    parts = sharding.row_partition(V, 3)
    max_rows = max(b - a for a, b in parts)
    full = {}
    for name, dt, per_c in sharding.PLANES:
        shape = (V, U, C) if per_c else (V, U)
        a = rng.integers(0, 200, size=shape)
        full[name] = torch.from_numpy(a.astype(np.float32)).to(dt)
    bufs = [sharding.pack_planes(full, slice(a, b), max_rows, U, C) for a, b in parts]
    assert len({b.numel() for b in bufs}) == 1
    out = sharding.unpack_planes(bufs, parts, max_rows, U, C)
    for k in full:
        assert torch.equal(out[k], full[k]), k


def _scale_worker(rank, world, port, V, S, U, D, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        raw = _raw_field(V, S, U)
        shard = sharding.make_shard(V, rank, world, 5)
        mine = raw[shard.rows]
        # the constructor's default scale (-1 = max over all EPIs): one MAX all-reduce makes it the same on every rank
        scale = sharding.global_epi_scale(float(mine.max()), -1.0)
        vol, used = oracle.normalize_f32(mine, scale)
        r = oracle.depth1d_pile_run(vol, -1.0, 2.0, D)
        planes = dict(edge_confidence=torch.from_numpy(r.edge_confidence), disp_confidence=torch.from_numpy(r.disp_confidence),
                      depth=torch.from_numpy(r.depth), depth_raw=torch.from_numpy(r.depth_raw), score=torch.from_numpy(r.score),
                      depth_idx=torch.from_numpy(r.depth_idx), rbar=torch.from_numpy(r.rbar), edge_mask=torch.from_numpy(r.edge_mask))
        out = sharding.gather_planes(planes, shard, U, 1)
        if rank == 0:
            np.savez(out_path, scale=np.float32(scale), **{k: v.numpy() for k, v in out.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _raw_field(V, S, U):
    """Un-normalised float radiances whose maximum sits in the LAST block, so a block-local maximum differs."""
    from remotesensingproject_amd.synth import make_lightfield
    vol, _ = make_lightfield(U, V, S, 1, seed=5, dmin=-1.0, dmax=2.0, band=3)
    raw = (vol * np.float32(180.0)).astype(np.float32)
    raw[:V // 2] *= np.float32(0.35)     # darker first half: its own maximum would rescale it by almost 3
    return raw


def test_sharded_default_normalisation_uses_the_global_maximum(tmp_path, oracle_mod):
    """with epi_scale_factor < 0 every rank must divide by the maximum over ALL EPIs (dc.hpp:442-474), not
    by the maximum of its own block -- else thresholds act on differently scaled radiances and the stitched planes
    show seams.  global_epi_scale() all-reduces the raw maxima; the stitched planes then equal the unsharded run."""
    V, S, U, D, world = 12, 9, 48, 12, 2
    out_path = str(tmp_path / "scaled.npz")
    mp.spawn(_scale_worker, args=(world, _free_port(), V, S, U, D, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    raw = _raw_field(V, S, U)
    vol, used = oracle_mod.normalize_f32(raw, -1.0)
    assert float(got["scale"]) == float(raw.max()) == used
    ref = oracle_mod.depth1d_pile_run(vol, -1.0, 2.0, D)
    for k in ("edge_confidence", "edge_mask", "disp_confidence", "depth", "rbar", "depth_idx", "score", "depth_raw"):
        assert np.array_equal(got[k], getattr(ref, k)), k
    # and the block-local maximum would indeed have given different planes (the test field is built for that)
    half, _ = oracle_mod.normalize_f32(raw[:V // 2 + 2], -1.0)
    assert not np.array_equal(oracle_mod.depth1d_pile_run(half, -1.0, 2.0, D).edge_confidence[:V // 2], ref.edge_confidence[:V // 2])
    with pytest.raises(ValueError):
        sharding.require_explicit_scale(-1.0, 2)
    sharding.require_explicit_scale(-1.0, 1)
    sharding.require_explicit_scale(255.0, 8)


def _halo_worker(rank, world, port, h, U, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = 6
        depth = torch.full((rows + 2 * h, U), float(rank), dtype=torch.float32)        # halo | own | halo, own rows tagged
        mask = torch.full((rows + 2 * h, U), rank, dtype=torch.uint8)
        depth[h:h + rows] += torch.arange(rows, dtype=torch.float32)[:, None] / 16.0
        a, b = h, h + rows
        top, bottom = (depth[a:a + h], mask[a:a + h]), (depth[b - h:b], mask[b - h:b])
        above = (depth[a - h:a], mask[a - h:a]) if rank > 0 else None
        below = (depth[b:b + h], mask[b:b + h]) if rank < world - 1 else None
        sharding.exchange_halo_rows(rank, world, top, bottom, above, below)
        np.savez(os.path.join(out_dir, "halo%d.npz" % rank), depth=depth.numpy(), mask=mask.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_row_exchange_between_neighbours(tmp_path, world):
    """The per-visit exchange of the sharded 2-D sweep (ShardedDepth2D.exchange): every rank's halo rows end up holding
    the neighbour's boundary rows, both planes, the field's outer halos untouched."""
    h, U, rows = 2, 9, 6
    mp.spawn(_halo_worker, args=(world, _free_port(), h, U, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / ("halo%d.npz" % r)) for r in range(world)]
    for r in range(world):
        d, m = got[r]["depth"], got[r]["mask"]
        own = d[h:h + rows]
        assert np.array_equal(own, r + np.arange(rows, dtype=np.float32)[:, None] / 16.0 + np.zeros((rows, U), np.float32))
        if r > 0:
            assert np.array_equal(d[:h], got[r - 1]["depth"][rows:rows + h]) and np.all(m[:h] == r - 1)
        else:
            assert np.all(d[:h] == 0.0) and np.all(m[:h] == 0)
        if r < world - 1:
            assert np.array_equal(d[h + rows:], got[r + 1]["depth"][h:2 * h]) and np.all(m[h + rows:] == r + 1)
        else:
            assert np.all(d[h + rows:] == r) and np.all(m[h + rows:] == r)


def test_sweep_order_centre_outwards():
    assert sharding.sweep_order(5) == [2, 3, 1, 4, 0]
    assert sharding.sweep_order(1) == [0]
    assert sharding.sweep_order(8) == [4, 5, 3, 6, 2, 7, 1]          # an even view count never visits view 0 (core.hpp:981-990)


def _gather_worker(rank, world, port, V, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        S, U = 3, 7
        full = torch.arange(S * V * U, dtype=torch.float32).reshape(S, V, U)
        a, b = sharding.row_partition(V, world)[rank]
        got = sharding._gather_rows(full[:, a:b].contiguous(), V, rank, world)
        gotb = sharding._gather_rows((full[:, a:b] % 251).to(torch.uint8).contiguous(), V, rank, world)
        np.savez(os.path.join(out_dir, "g%d.npz" % rank), f=got.numpy(), b=gotb.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,V", [(2, 10), (3, 11)])
def test_all_gather_of_ragged_row_blocks(tmp_path, world, V):
    """The per-level all-gather of the sharded fine-to-coarse run: every rank ends with the whole plane, blocks of
    unequal height included (f32 disparities and u8 validity)."""
    mp.spawn(_gather_worker, args=(world, _free_port(), V, str(tmp_path)), nprocs=world, join=True)
    want = np.arange(3 * V * 7, dtype=np.float32).reshape(3, V, 7)
    for r in range(world):
        q = np.load(tmp_path / ("g%d.npz" % r))
        assert np.array_equal(q["f"], want) and np.array_equal(q["b"], (want % 251).astype(np.uint8))
