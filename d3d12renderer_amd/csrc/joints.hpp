// joints.hpp — joint constraints (distance, ball, fixed, hinge, cone-twist, slider).  Filled in by
// the joints milestone.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mi_physics.h"
struct mi_world;
struct JointSet {
    int upload(mi_world&, hipStream_t) { return MI_OK; }
    int initialize(mi_world&, float, hipStream_t) { return MI_OK; }
    void solveIteration(mi_world&, hipStream_t) {}
    int add(mi_world&, uint32_t, uint32_t, uint32_t, const void*, uint32_t, uint32_t*) { return MI_ERR_UNSUPPORTED; }
    int update(uint32_t, uint32_t, const void*, uint32_t) { return MI_ERR_UNSUPPORTED; }
    int get(uint32_t, uint32_t, void*, uint32_t) { return MI_ERR_UNSUPPORTED; }
    int addFromGlobal(mi_world&, uint32_t, uint32_t, uint32_t, const float*, const float*, float, float, uint32_t*) { return MI_ERR_UNSUPPORTED; }
};
