// librslf_hip.so, unit 4 of 9: the on-chip scan kernel's instantiations, the middle rungs (151 .. 175 views) -- see rslf_chip_a.hip.
#include "rslf_internal.hpp"

#include "k2_scan.hpp"
#include "k2_chip.hpp"

namespace rslf {

RSLF_CHIP_PART_LAUNCHER(launch_chip_part_b, RSLF_CHIP_LADDER_B)

}  // namespace rslf
