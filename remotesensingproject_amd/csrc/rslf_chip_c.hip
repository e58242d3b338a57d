// librslf_hip.so, unit 5 of 9: the on-chip scan kernel's instantiations, the lowest rungs (127 .. 143 views) -- see rslf_chip_a.hip.
#include "rslf_internal.hpp"

#include "k2_scan.hpp"
#include "k2_chip.hpp"

namespace rslf {

RSLF_CHIP_PART_LAUNCHER(launch_chip_part_c, RSLF_CHIP_LADDER_C)

}  // namespace rslf
