// librslf_hip.so, unit 3 of 9: the on-chip scan kernel (k2_chip.hpp) -- one instantiation per rung of plan::kChipLadder, in
// three translation units of their own (this one, rslf_chip_b.hip, rslf_chip_c.hip: each instantiation is ~30 k
// instructions of unrolled code and ~20 s of hipcc; beside the other scan kernels they would be the library's whole
// build time in one process).  Here: the top rung -- BASELINE.json's c5 exactly, its ragged-tail form for up to 220 views,
// its padded form -- and the two rungs below it; and launch_scan_chip, which the hot path's unit (rslf_pile.hip) calls.
#include "rslf_internal.hpp"

#include "k2_scan.hpp"
#include "k2_chip.hpp"

namespace rslf {

RSLF_CHIP_PART_LAUNCHER(launch_chip_part_a, RSLF_CHIP_LADDER_A)

int launch_scan_chip(const ScanArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    const int S = a.vol.S;
    const int rung = a.vol.C == 3 ? plan::chip_rung_for(S) : -1;
    if (rung < 0)
        return fail(RSLF_ERR_INTERNAL, "launch_scan_chip: no rung for %d views, %d channels", S, a.vol.C);
    const plan::ChipRung r = plan::kChipLadder[rung];
    if (lds_bytes != (size_t)kScanWaves * plan::chip_wave_floats(S, r) * sizeof(float) || lds_bytes > kChipLdsBytes)
        return fail(RSLF_ERR_INTERNAL, "launch_scan_chip: %zu bytes of LDS planned for rung <%d, %d> at %d views", lds_bytes, r.na, r.nl, S);
    // every view has a place (or a fetched-ahead slot) on the TOP rung at exactly its view count -- BASELINE.json's c5: neither
    // padding nor a ragged tail compiled in (the tail's registers cost the 201-view kernel 0.5 %)
    if (S == plan::kChipTopS)
        return launch_chip_rung<false, plan::kChipNAMax, plan::kChipNLMax, false>(a, grid, lds_bytes, stream);
    if (S > plan::kChipTopS)
        return launch_chip_rung<true, plan::kChipNAMax, plan::kChipNLMax, false>(a, grid, lds_bytes, stream);
    int rc = launch_chip_part_a(r.na, r.nl, a, grid, lds_bytes, stream);
    if (rc == RSLF_ERR_UNSUPPORTED)
        rc = launch_chip_part_b(r.na, r.nl, a, grid, lds_bytes, stream);
    if (rc == RSLF_ERR_UNSUPPORTED)
        rc = launch_chip_part_c(r.na, r.nl, a, grid, lds_bytes, stream);
    if (rc == RSLF_ERR_UNSUPPORTED)
        return fail(RSLF_ERR_INTERNAL, "launch_scan_chip: rung <%d, %d> is not compiled in", r.na, r.nl);
    return rc;
}

}  // namespace rslf
