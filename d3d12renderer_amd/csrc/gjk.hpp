// gjk.hpp — GJK + EPA on the device (one pair per lane).  Filled in by the GJK/EPA milestone.
#pragma once
#include "kernels.hpp"
namespace mi {
__device__ inline bool intersectGjk(const Shape&, const Shape&, const HullSet&, Manifold&, int) { return false; }
}  // namespace mi
