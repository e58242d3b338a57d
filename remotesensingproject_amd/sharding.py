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

import ctypes as C

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
    halo = max(0, int((median_filter_size - 1) / 2))   # C++ division truncates: size 0 is the 1 x 1 window too
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


# ---- the 2-D sweep over scanline shards ---------------------------------------------------------------------------------
#
# Depth2DComputer::run (dc.hpp:748-805): the scan and the propagation of a visit are local to a scanline, the selective
# median between them reads +-(size-1)/2 scanlines of the VISITED view's raw disparities and edge mask (core.hpp:686).
# Recomputing a halo does not work here as it does for the pile path -- every visit would widen it by two rows -- so this
# is the one place on the path with a real exchange step: after a visit's scan each rank sends its first and last
# (size-1)/2 own rows of those two planes to its neighbours (two small point-to-point messages per neighbour and visit,
# 2 x 1920 x 5 B each at c3), then runs the median, the claims and the apply pass on its own rows.

def sweep_order(S: int) -> List[int]:
    """core.hpp:981-990: centre view, then outwards, alternating."""
    s_mid = int(np.floor(S / 2.0))
    order = [s_mid]
    for off in range(1, S - s_mid):
        order.append(s_mid + off)
        if s_mid - off > -1:
            order.append(s_mid - off)
    return order


def exchange_halo_rows(rank: int, world: int, top, bottom, above, below, group=None) -> None:
    """Neighbour exchange of one visit: `top` / `bottom` (tuples of tensors: this rank's first / last own rows) go to
    ranks rank-1 / rank+1, whose bottom / top rows land in `above` / `below` (None where there is no neighbour).  One
    batch of point-to-point operations -- on RCCL they are stream-ordered device-to-device transfers over xGMI; gloo
    (CPU rehearsal, tests) goes through host copies."""
    if world == 1 or not dist.is_initialized():
        return
    via_host = dist.get_backend(group) == "gloo"
    ops, landing = [], []
    for peer, send, recv in ((rank - 1, top, above), (rank + 1, bottom, below)):
        if peer < 0 or peer >= world or recv is None:
            continue
        for t_send, t_recv in zip(send, recv):
            src = t_send.contiguous().cpu() if via_host else t_send.contiguous()
            dst = torch.empty_like(src)
            ops.append(dist.P2POp(dist.isend, src, peer, group))
            ops.append(dist.P2POp(dist.irecv, dst, peer, group))
            landing.append((t_recv, dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for t_recv, dst in landing:
        t_recv.copy_(dst)


class ShardedDepth2D:
    """One rank's share of Depth2DComputer::run: `vol` holds scanlines [shard.lo, shard.hi) of the light field (the own
    block plus halo_rows() either side, normalised with the GLOBAL scale -- global_epi_scale), planes are [S, rows, U].

    run() drives the visits with a neighbour exchange over torch.distributed (RCCL: stream-ordered, no host
    synchronisation); the steps are public so that a lock-step harness can drive several shards in one process
    (tests/test_gpu_sweep2d.py)."""

    def __init__(self, vol, shard: Shard, dmin, dmax, dim_d: int, parameters=None, group=None):
        """dmin / dmax: floats, or [S, rows, U] f32 planes over the volume's rows (a fine-to-coarse level's tightened
        per-pixel ranges, dc.hpp:201-203)."""
        from . import depth as rs
        self.rs, self.vol, self.shard, self.group = rs, vol, shard, group
        self.bounds = (dmin, dmax) if isinstance(dmin, torch.Tensor) else None
        self.p = parameters or rs.Depth1DParameters()
        self.dmin, self.dmax = (0.0, 0.0) if self.bounds else (float(dmin), float(dmax))
        self.dim_d = int(dim_d)
        self.h = halo_rows(int(self.p.par_median_filter_size), 1)         # rows the median reads either side
        self.own = shard.interior                                         # own rows in the local frame
        if vol.V != shard.hi - shard.lo:
            raise ValueError("the volume must hold rows [%d, %d) of the light field" % (shard.lo, shard.hi))
        if (shard.v0 > 0 and shard.v0 - shard.lo < self.h) or (shard.v1 < shard.V and shard.hi - shard.v1 < self.h):
            raise ValueError("the shard's halo is narrower than the median's reach")
        if shard.v1 - shard.v0 < self.h:
            raise ValueError("a block of %d scanlines cannot fill its neighbours' %d halo rows" % (shard.v1 - shard.v0, self.h))
        dev = vol.ctx.device
        S, V, U, C_ = vol.S, vol.V, vol.U, vol.C
        self.Ce = torch.zeros((S, V, U), dtype=torch.float32, device=dev)
        self.Cd = torch.zeros((S, V, U), dtype=torch.float32, device=dev)
        self.depth = torch.zeros((S, V, U), dtype=torch.float32, device=dev)
        self.rbar = torch.zeros((S, V, U, C_), dtype=torch.float32, device=dev)
        self.scan_mask = torch.empty((S, V, U), dtype=torch.uint8, device=dev)
        self.cem = None
        self.stats = None

    # -- steps -------------------------------------------------------------------------------------------------------------
    def prepare(self) -> None:
        rs, L = self.rs, self._L()
        self.cem = rs.compute_2D_edge_confidence(self.vol, self.Ce, self.p)     # every local row: the masks are row-local
        self.vol.ctx.use_current_stream()
        self._check(L.rslf_sweep_begin(self.vol.ctx._h, self.vol._h, self._ptr(self.cem), self._ptr(self.scan_mask), self.dim_d,
                                       self.own.start, self.own.stop), "rslf_sweep_begin")

    def visit_scan(self, s_hat: int) -> None:
        pc = self.p.to_c()
        self.vol.ctx.use_current_stream()
        lo, hi = self.bounds if self.bounds else (None, None)
        self._check(self._L().rslf_sweep_visit_scan(self.vol.ctx._h, self.vol._h, self._ptr(lo), self._ptr(hi), self.dmin, self.dmax, self.dim_d, int(s_hat),
                                                    self._ptr(self.Ce), self._ptr(self.cem), self._ptr(self.Cd), self._ptr(self.depth),
                                                    self._ptr(self.rbar), C.byref(pc)), "rslf_sweep_visit_scan")

    def boundary_rows(self, s_hat: int):
        """(top, bottom): this rank's first / last h own rows of the visited view's raw depths and edge mask."""
        a, b, h = self.own.start, self.own.stop, self.h
        return ((self.depth[s_hat, a:a + h], self.cem[s_hat, a:a + h]), (self.depth[s_hat, b - h:b], self.cem[s_hat, b - h:b]))

    def halo_slots(self, s_hat: int):
        """(above, below): where the upper / lower neighbour's boundary rows belong (None at the field's edge)."""
        a, b, h = self.own.start, self.own.stop, self.h
        above = (self.depth[s_hat, a - h:a], self.cem[s_hat, a - h:a]) if self.shard.v0 > 0 else None
        below = (self.depth[s_hat, b:b + h], self.cem[s_hat, b:b + h]) if self.shard.v1 < self.shard.V else None
        return above, below

    def visit_finish(self, s_hat: int) -> None:
        pc = self.p.to_c()
        self.vol.ctx.use_current_stream()
        self._check(self._L().rslf_sweep_visit_finish(self.vol.ctx._h, self.vol._h, int(s_hat), self._ptr(self.cem), self._ptr(self.Cd),
                                                      self._ptr(self.depth), self._ptr(self.rbar), C.byref(pc)), "rslf_sweep_visit_finish")

    def finish(self, ok: bool = True) -> None:
        from ._lib import RslfStats
        st = RslfStats()
        self._check(self._L().rslf_sweep_end(self.vol.ctx._h, 1 if ok else 0, self.dim_d, C.byref(st)), "rslf_sweep_end")
        self.stats = st

    # -- the distributed driver --------------------------------------------------------------------------------------------
    def exchange(self, s_hat: int) -> None:
        """Send the own boundary rows to the neighbours, receive theirs into the halo rows (two planes, 2 x h rows)."""
        top, bottom = self.boundary_rows(s_hat)
        above, below = self.halo_slots(s_hat)
        exchange_halo_rows(self.shard.rank, self.shard.world, top, bottom, above, below, self.group)

    def run(self) -> None:
        self.prepare()
        try:
            for s_hat in sweep_order(self.vol.S):
                self.visit_scan(s_hat)
                self.exchange(s_hat)
                self.visit_finish(s_hat)
        except Exception:
            self.finish(ok=False)
            raise
        self.finish()

    def own_planes(self) -> Dict[str, torch.Tensor]:
        o = self.own
        return dict(edge_confidence=self.Ce[:, o], edge_mask=self.cem[:, o], disp_confidence=self.Cd[:, o], depth=self.depth[:, o],
                    rbar=self.rbar[:, o], scan_mask=self.scan_mask[:, o])

    # -- helpers -------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _L():
        from . import _lib
        return _lib.lib()

    @staticmethod
    def _check(status, where):
        from . import _lib
        _lib.check(status, where)

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None


def run_lockstep_sweep(shards: List["ShardedDepth2D"], exchange: bool = True) -> None:
    """Several shards in ONE process, visit by visit, halo rows copied between them directly -- what the ranks do over
    RCCL, without the transport (tests; a one-GPU box).  exchange=False leaves the halo rows stale: what the tests use to
    show that the exchange is load-bearing."""
    for sh in shards:
        sh.prepare()
    for s_hat in sweep_order(shards[0].vol.S):
        for sh in shards:
            sh.visit_scan(s_hat)
        for i, sh in enumerate(shards if exchange else ()):
            above, below = sh.halo_slots(s_hat)
            if above is not None:
                for dst, src in zip(above, shards[i - 1].boundary_rows(s_hat)[1]):
                    dst.copy_(src)
            if below is not None:
                for dst, src in zip(below, shards[i + 1].boundary_rows(s_hat)[0]):
                    dst.copy_(src)
        for sh in shards:
            sh.visit_finish(s_hat)
    for sh in shards:
        sh.finish()


# ---- fine-to-coarse over scanline shards -------------------------------------------------------------------------------------

def _gather_rows(own: torch.Tensor, V: int, rank: int, world: int, group=None) -> torch.Tensor:
    """All-gather of row blocks: `own` is this rank's [S, rows, U] block of a [S, V, U] plane cut by row_partition(V, world);
    returns the whole plane on every rank.  Blocks differ by at most one row: every rank sends max_rows (zero padded)."""
    parts = row_partition(V, world)
    max_rows = max(b - a for a, b in parts)
    S, rows, U = own.shape
    via_host = dist.get_backend(group) == "gloo"
    send = torch.zeros((S, max_rows, U), dtype=own.dtype, device="cpu" if via_host else own.device)
    send[:, :rows] = own
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    full = torch.empty((S, V, U), dtype=own.dtype, device=own.device)
    for (a, b), t in zip(parts, recv):
        full[:, a:b] = t[:, :b - a].to(own.device)
    return full


class ShardedFineToCoarse:
    """One rank's share of rslf::FineToCoarse (rslf_fine_to_coarse.hpp:103-324).

    What is sharded is where the time goes: every level's 2-D sweep runs on this rank's block of scanlines
    (ShardedDepth2D: one neighbour exchange of 2 boundary rows per visit).  The pyramid itself -- Gaussian blur and
    halving, ~1 % of a run, and in need of the whole image's borders and odd-row rule -- is built on every rank from the
    whole light field, so a level's normalisation is the whole level's maximum as in the unsharded run; each level is cut
    by row_partition on its own.  After a level's sweep the ranks all-gather its disparity and validity planes (5 bytes
    per pixel and view), and the bound tightening for the next level and the final coarse-to-fine fusion -- cheap
    whole-image passes with non-local footprints (nearest valid column, bilinear upscaling, 3x3 median) -- run on the
    gathered planes.  Results are those of FineToCoarse bit for bit (tests/test_gpu_f2c.py)."""

    def __init__(self, epis, d_min: float, d_max: float, dim_d: int, rank: int, world: int, epi_scale_factor: float = -1.0,
                 parameters=None, max_pyr_depth: int = -1, accept_all_last_scale: bool = True, ctx=None, group=None):
        import copy
        from . import depth as rs
        from . import _lib
        self.rs, self.rank, self.world, self.group = rs, int(rank), int(world), group
        self.m_parameters = parameters or rs.Depth1DParameters.get_default()
        self.ctx = ctx or rs.default_context()
        dev = self.ctx.device
        a = np.stack([np.asarray(e) for e in epis]) if isinstance(epis, (list, tuple)) else np.asarray(epis)
        if a.ndim == 3:
            a = a[..., None]
        is_u8 = a.dtype == np.uint8
        raw = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
        start_dim_u = raw.shape[2]
        if max_pyr_depth < 1:
            max_pyr_depth = 1 << 30
        self.levels: List[dict] = []
        dim_v, dim_u, counter = raw.shape[0], raw.shape[2], 0
        L = _lib.lib()
        while dim_v > rs._MIN_SPATIAL_DIM and dim_u > rs._MIN_SPATIAL_DIM and counter < max_pyr_depth:   # f2c.hpp:130
            counter += 1
            par = copy.copy(self.m_parameters)
            par.par_slope_factor = float(np.float32((0.0 + dim_u) / start_dim_u))                          # f2c.hpp:139
            if is_u8:
                scale = 255.0
            elif epi_scale_factor < 0:      # the level's own maximum (dc.hpp:671-705) -- of the WHOLE level
                mx = C.c_float()
                self.ctx.use_current_stream()
                _lib.check(L.rslf_device_max_f32(self.ctx._h, C.c_void_p(raw.data_ptr()), raw.numel(), C.byref(mx)), "rslf_device_max_f32")
                scale = float(mx.value)
            else:
                scale = float(epi_scale_factor)
            # a coarse level whose blocks would be thinner than the halo they must fill is small enough to run whole
            # on every rank
            replicated = self.world == 1 or dim_v // self.world < max(1, halo_rows(par.par_median_filter_size, par.par_edge_confidence_opening_size))
            shard = make_shard(dim_v, 0, 1) if replicated else make_shard(dim_v, self.rank, self.world, par.par_median_filter_size,
                                                                          par.par_edge_confidence_opening_size)
            vol = rs.Volume.from_dense(raw[shard.rows].contiguous(), scale, self.ctx)
            self.levels.append(dict(V=dim_v, U=dim_u, par=par, shard=shard, vol=vol, sweep=None, depth=None, valid=None, accept_all=False,
                                    replicated=replicated))
            raw = rs.downsample_EPIs(raw, self.ctx, is_u8)                                                 # f2c.hpp:145-147
            dim_v, dim_u = raw.shape[0], raw.shape[2]
        if not self.levels:
            raise ValueError("light field smaller than _MIN_SPATIAL_DIM: no pyramid level")
        if accept_all_last_scale:
            self.levels[-1]["accept_all"] = True                                                           # f2c.hpp:157-158
        self._dmin, self._dmax, self._dim_d = float(d_min), float(d_max), int(dim_d)
        self.S = self.levels[0]["vol"].S

    # -- steps (a lock-step harness drives them for several ranks in one process) ---------------------------------------
    def begin_level(self, p: int) -> ShardedDepth2D:
        """The sweep of level p over this rank's rows; levels below the finest take their per-pixel ranges from the
        gathered planes of level p - 1 (f2c.hpp:202-294)."""
        from . import _lib
        lv = self.levels[p]
        if p == 0:
            lo, hi = self._dmin, self._dmax
        else:
            up = self.levels[p - 1]
            if up["depth"] is None:
                raise RuntimeError("level %d needs the gathered planes of level %d" % (p, p - 1))
            dev = self.ctx.device
            lo = torch.full((self.S, lv["V"], lv["U"]), self._dmin, dtype=torch.float32, device=dev)
            hi = torch.full((self.S, lv["V"], lv["U"]), self._dmax, dtype=torch.float32, device=dev)
            self.ctx.use_current_stream()
            vp = C.c_void_p
            _lib.check(_lib.lib().rslf_f2c_tighten_bounds(self.ctx._h, vp(up["depth"].data_ptr()), vp(up["valid"].data_ptr()), self.S, up["V"],
                                                          up["U"], vp(lo.data_ptr()), vp(hi.data_ptr()), lv["V"], lv["U"]), "rslf_f2c_tighten_bounds")
            rows = lv["shard"].rows
            lo, hi = lo[:, rows].contiguous(), hi[:, rows].contiguous()
        lv["sweep"] = ShardedDepth2D(lv["vol"], lv["shard"], lo, hi, self._dim_d, lv["par"], self.group)
        return lv["sweep"]

    def own_level_planes(self, p: int):
        """(disparities, validity) of this rank's own rows of level p: [S, rows, U] f32 / u8 (dc.hpp:893-915)."""
        lv = self.levels[p]
        sw = lv["sweep"]
        thr = -1.0 if lv["accept_all"] else float(np.float32(lv["par"].par_edge_score_threshold))
        o = sw.own
        return sw.depth[:, o].contiguous(), ((sw.Ce[:, o] > thr).to(torch.uint8) * 255).contiguous()

    def set_level_planes(self, p: int, depth_full: torch.Tensor, valid_full: torch.Tensor) -> None:
        self.levels[p]["depth"], self.levels[p]["valid"] = depth_full.contiguous(), valid_full.contiguous()

    def run(self) -> None:
        """f2c.hpp:171-299 over torch.distributed: per level one sharded sweep and one all-gather of two planes."""
        for p, lv in enumerate(self.levels):
            self.begin_level(p).run()
            depth, valid = self.own_level_planes(p)
            if not lv["replicated"]:
                depth = _gather_rows(depth, lv["V"], self.rank, self.world, self.group)
                valid = _gather_rows(valid, lv["V"], self.rank, self.world, self.group)
            self.set_level_planes(p, depth, valid)

    def get_results(self):
        """f2c.hpp:302-324 on the gathered planes -> (out_map_s_v_u, out_validity_s_v_u) at the finest scale."""
        from . import _lib
        P = len(self.levels)
        dp = (C.c_void_p * P)(*[lv["depth"].data_ptr() for lv in self.levels])
        vp = (C.c_void_p * P)(*[lv["valid"].data_ptr() for lv in self.levels])
        Vp = (C.c_int * P)(*[lv["V"] for lv in self.levels])
        Up = (C.c_int * P)(*[lv["U"] for lv in self.levels])
        dev = self.ctx.device
        out_map = torch.empty((self.S, self.levels[0]["V"], self.levels[0]["U"]), dtype=torch.float32, device=dev)
        out_valid = torch.empty((self.S, self.levels[0]["V"], self.levels[0]["U"]), dtype=torch.uint8, device=dev)
        self.ctx.use_current_stream()
        _lib.check(_lib.lib().rslf_f2c_fuse(self.ctx._h, dp, vp, Vp, Up, P, self.S, C.c_void_p(out_map.data_ptr()),
                                            C.c_void_p(out_valid.data_ptr())), "rslf_f2c_fuse")
        return out_map, out_valid

    @property
    def pixels_scanned(self) -> int:
        """Pixels this rank scanned over all levels (a replicated level counts on every rank)."""
        return sum(int(lv["sweep"].stats.pixels_scanned) for lv in self.levels if lv["sweep"] is not None and lv["sweep"].stats)


def run_lockstep_f2c(ranks: List["ShardedFineToCoarse"]) -> None:
    """Several ranks of a sharded fine-to-coarse run in ONE process: level by level, the sweeps in lock step and the
    all-gather as a concatenation (tests; a one-GPU box)."""
    for p in range(len(ranks[0].levels)):
        run_lockstep_sweep([r.begin_level(p) for r in ranks])
        planes = [r.own_level_planes(p) for r in ranks]
        if ranks[0].levels[p]["replicated"]:
            for r, (d, m) in zip(ranks, planes):
                r.set_level_planes(p, d, m)
            continue
        for r in ranks:
            dev = r.ctx.device
            r.set_level_planes(p, torch.cat([d.to(dev) for d, _ in planes], dim=1), torch.cat([m.to(dev) for _, m in planes], dim=1))
