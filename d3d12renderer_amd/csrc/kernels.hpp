// kernels.hpp — HIP kernels of the rigid-body step for gfx950 (wave64).
//
// Stage names follow the reference's profiler blocks.  HBM layout is SoA with
// float4 rows: a wave touching 64 consecutive bodies / contacts issues 1 KiB coalesced transactions.
// No MFMA anywhere: there is no dense contraction on this path; the roofline is HBM bandwidth.
#pragma once
#include <type_traits>
#include <utility>
#include "dmath.hpp"
#include "narrow.hpp"

// (split by stage in round 6; one translation unit as before — the files below are meant to be read in this order)
#include "kernels_common.hpp"
#include "kernels_broad.hpp"
#include "kernels_narrow.hpp"
#include "kernels_integrate.hpp"
#include "kernels_schedule.hpp"
#include "kernels_contacts.hpp"
#include "kernels_solve.hpp"
#include "kernels_shard.hpp"
#include "kernels_scan.hpp"
