"""Scanline sharding of the pile path over the GPUs of one node.

K1 (edge confidence) and K2 (the scan) are independent per scanline v
(rslf_depth_computation_core.hpp:743-757, :799-854); K3 (selective median,
:663-718) reads rows v +- (size-1)/2 of the depth plane, the mask and the
s_hat row of the volume.  So each rank takes a contiguous block of scanlines
plus a halo of (size-1)/2 rows on each side that it RECOMPUTES (no exchange),
and the only data-path collective is the reassembly of the output planes on
rank 0 (PlaneGatherer): per-plane gathers straight into the final planes when
the blocks are equal, else one gather of a packed, padded byte buffer.

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm.  The
partition/stitch logic is backend-agnostic and is exercised on CPU with gloo
(tests/test_sharding_gloo.py), where the per-rank compute is injected.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

# plane name -> (torch dtype, trailing channels multiplier: 1 or "C")
PLANES: List[Tuple[str, torch.dtype, bool]] = [
    ("edge_confidence", torch.float32, False),
    ("disp_confidence", torch.float32, False),
    ("depth", torch.float32, False),
    ("depth_raw", torch.float32, False),
    ("score", torch.float32, False),
    ("depth_idx", torch.int32, False),
    ("rbar", torch.float32, True),
    ("edge_mask", torch.uint8, False),
]


def row_partition(V: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous scanline blocks [v0, v1), sizes differing by at most one."""
    base, rem = divmod(V, world)
    out, v = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((v, v + n))
        v += n
    return out


@dataclass
class Shard:
    rank: int
    world: int
    V: int            # scanlines of the whole light field
    v0: int           # owned block [v0, v1)
    v1: int
    lo: int           # computed block [lo, hi) = owned + halo, clipped to [0, V)
    hi: int

    @property
    def rows(self) -> slice:
        """Rows of the whole volume this rank must hold."""
        return slice(self.lo, self.hi)

    @property
    def interior(self) -> slice:
        """Owned rows, in the local (computed-block) frame."""
        return slice(self.v0 - self.lo, self.v1 - self.lo)


def halo_rows(median_filter_size: int = 5, opening_size: int = 1) -> int:
    """Scanlines either side of a block that must be recomputed for the block's rows to come out exact: the
    selective median reads +-(size-1)/2 rows (core.hpp:686), and those rows' masks depend, through the optional
    opening (erosion then dilation, core.hpp:759-768), on +-2*(k/2) rows more."""
    halo = (median_filter_size - 1) // 2
    if opening_size > 1:
        halo += 2 * (opening_size // 2)
    return halo


def make_shard(V: int, rank: int, world: int, median_filter_size: int = 5, opening_size: int = 1) -> Shard:
    halo = halo_rows(median_filter_size, opening_size)
    v0, v1 = row_partition(V, world)[rank]
    return Shard(rank, world, V, v0, v1, max(0, v0 - halo), min(V, v1 + halo))


def global_epi_scale(local_max: float, epi_scale_factor: float, group=None) -> float:
    """The scale every rank must normalise its rows with so that the stitched planes equal the unsharded run.

    The constructor's default (epi_scale_factor < 0) divides by the maximum over ALL EPIs (dc.hpp:442-460, :474); a rank
    that built its volume from its own rows alone would divide by its block's maximum, and every threshold of the path
    would then act on differently scaled radiances from block to block.  So: one scalar all-reduce (MAX) of the raw
    maxima -- the only collective besides the reassembly, and exact in any order.  `local_max` is the maximum of this
    rank's raw rows (block + halo); a given scale (>= 0) is returned unchanged."""
    if epi_scale_factor >= 0:
        return float(epi_scale_factor)
    m = torch.tensor([float(local_max)], dtype=torch.float32)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            m = m.cuda()
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return float(m.item())


def require_explicit_scale(epi_scale_factor: float, world: int) -> None:
    """A sharded run must not be built with the 'max over my own rows' default: see global_epi_scale."""
    if world > 1 and epi_scale_factor < 0:
        raise ValueError("a sharded run needs one scale for all ranks: pass sharding.global_epi_scale(local_max, %g) "
                         "instead of epi_scale_factor=%g" % (epi_scale_factor, epi_scale_factor))


def _row_bytes(U: int, C: int) -> int:
    n = 0
    for _, dt, per_c in PLANES:
        n += U * (C if per_c else 1) * torch.empty((), dtype=dt).element_size()
    return n


def pack_planes(planes: Dict[str, torch.Tensor], rows: slice, max_rows: int, U: int, C: int) -> torch.Tensor:
    """Owned rows of every output plane, back to back, as one uint8 buffer of
    max_rows * row_bytes (short blocks are zero padded so every rank sends the same size)."""
    dev = planes["depth"].device
    buf = torch.zeros(max_rows * _row_bytes(U, C), dtype=torch.uint8, device=dev)
    off = 0
    for name, dt, per_c in PLANES:
        t = planes[name][rows].contiguous()
        assert t.dtype == dt, (name, t.dtype, dt)
        b = t.view(torch.uint8).reshape(-1)
        buf[off:off + b.numel()] = b
        off += max_rows * U * (C if per_c else 1) * t.element_size()
    return buf


def unpack_planes(bufs: List[torch.Tensor], parts: List[Tuple[int, int]], max_rows: int, U: int, C: int) -> Dict[str, torch.Tensor]:
    """Inverse of pack_planes over all ranks: the stitched [V, U(, C)] planes."""
    V = parts[-1][1]
    dev = bufs[0].device
    out: Dict[str, torch.Tensor] = {}
    off = 0
    for name, dt, per_c in PLANES:
        cc = C if per_c else 1
        es = torch.empty((), dtype=dt).element_size()
        full = torch.empty((V, U, cc) if per_c else (V, U), dtype=dt, device=dev)
        for (v0, v1), buf in zip(parts, bufs):
            n = (v1 - v0) * U * cc * es
            seg = buf[off:off + n].view(dt).reshape((v1 - v0, U, cc) if per_c else (v1 - v0, U))
            full[v0:v1] = seg
        out[name] = full
        off += max_rows * U * cc * es
    return out


class PlaneGatherer:
    """Reassembly of the depth map on rank `dst`, with its buffers allocated once.

    Equal scanline blocks (V divisible by the world size -- 1080 / 8 = 135): every plane is
    gathered straight into row-block views of the final [V, U] plane, no staging copy on either
    side (eight small collectives, ~5 MB per rank in total).  Unequal blocks: one gather of the
    zero-padded packed buffer (pack_planes / unpack_planes)."""

    def __init__(self, shard: Shard, U: int, C: int, device, group=None, dst: int = 0):
        self.shard, self.U, self.C, self.group, self.dst = shard, U, C, group, dst
        self.parts = row_partition(shard.V, shard.world)
        sizes = {b - a for a, b in self.parts}
        self.equal = len(sizes) == 1
        self.max_rows = max(sizes)
        self.full: Optional[Dict[str, torch.Tensor]] = None
        if shard.rank == dst:
            self.full = {}
            for name, dt, per_c in PLANES:
                shape = (shard.V, U, C) if per_c else (shard.V, U)
                self.full[name] = torch.empty(shape, dtype=dt, device=device)
            if not self.equal:
                n = self.max_rows * _row_bytes(U, C)
                self.recv = [torch.empty(n, dtype=torch.uint8, device=device) for _ in range(shard.world)]

    def __call__(self, planes: Dict[str, torch.Tensor]) -> Optional[Dict[str, torch.Tensor]]:
        sh = self.shard
        if sh.world == 1:
            return planes
        # rehearsal on a box without RCCL peers: gloo moves CPU tensors only
        via_host = dist.get_backend(self.group) == "gloo" and planes["depth"].is_cuda
        if self.equal:
            for name, _, _ in PLANES:
                src = planes[name][sh.interior]
                if via_host:
                    src = src.cpu()
                if sh.rank == self.dst:
                    if via_host:
                        recv = [torch.empty_like(src) for _ in self.parts]
                        dist.gather(src, recv, dst=self.dst, group=self.group)
                        for (a, b), r in zip(self.parts, recv):
                            self.full[name][a:b].copy_(r)
                    else:
                        dist.gather(src, [self.full[name][a:b] for a, b in self.parts], dst=self.dst, group=self.group)
                else:
                    dist.gather(src, None, dst=self.dst, group=self.group)
            return self.full
        if via_host:
            raise RuntimeError("gloo rehearsal supports equal scanline blocks only")
        buf = pack_planes(planes, sh.interior, self.max_rows, self.U, self.C)
        if sh.rank == self.dst:
            dist.gather(buf, self.recv, dst=self.dst, group=self.group)
            out = unpack_planes(self.recv, self.parts, self.max_rows, self.U, self.C)
            for k in out:
                self.full[k].copy_(out[k])
            return self.full
        dist.gather(buf, None, dst=self.dst, group=self.group)
        return None


def gather_planes(planes: Dict[str, torch.Tensor], shard: Shard, U: int, C: int, group=None,
                  dst: int = 0) -> Optional[Dict[str, torch.Tensor]]:
    """Reassemble the depth map on rank `dst`: ONE gather of the packed owned rows."""
    parts = row_partition(shard.V, shard.world)
    max_rows = max(b - a for a, b in parts)
    buf = pack_planes(planes, shard.interior, max_rows, U, C)
    if shard.world == 1:
        return unpack_planes([buf], parts, max_rows, U, C)
    if shard.rank == dst:
        recv = [torch.empty_like(buf) for _ in range(shard.world)]
        dist.gather(buf, recv, dst=dst, group=group)
        return unpack_planes(recv, parts, max_rows, U, C)
    dist.gather(buf, None, dst=dst, group=group)
    return None


def run_sharded(local_compute: Callable[[Shard], Dict[str, torch.Tensor]], V: int, U: int, C: int,
                median_filter_size: int = 5, group=None, opening_size: int = 1) -> Optional[Dict[str, torch.Tensor]]:
    """local_compute(shard) -> output planes for rows [shard.lo, shard.hi) (it must run the
    full K1+K2+K3 path on exactly those rows); returns the stitched planes on rank 0."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    shard = make_shard(V, rank, world, median_filter_size, opening_size)
    planes = local_compute(shard)
    return gather_planes(planes, shard, U, C, group)
