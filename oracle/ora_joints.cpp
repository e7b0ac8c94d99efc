// ora_joints.cpp — TEST INFRASTRUCTURE (CPU oracle), see ora_world.h header note.
// Joint constraints (distance, ball, fixed, hinge, cone-twist, slider): storage + scalar
// initialize/solve routines of src/physics/constraints.cpp.  (Filled in by the joints milestone.)
#include "ora_world.h"

namespace ora {
struct JointStore { int unused = 0; };
JointStore* jointsCreate() { return new JointStore(); }
void jointsDestroy(JointStore* j) { delete j; }
int jointsAdd(World&, uint32_t, uint32_t, uint32_t, const void*, uint32_t, uint32_t*) { return MI_ERR_UNSUPPORTED; }
int jointsUpdate(World&, uint32_t, uint32_t, const void*, uint32_t) { return MI_ERR_UNSUPPORTED; }
int jointsGet(World&, uint32_t, uint32_t, void*, uint32_t) { return MI_ERR_UNSUPPORTED; }
int jointsAddFromGlobal(World&, uint32_t, uint32_t, uint32_t, const float*, const float*, float, float, uint32_t*) { return MI_ERR_UNSUPPORTED; }
void jointsInitialize(World&, float) {}
void jointsSolveIteration(World&) {}
uint32_t jointsCount(const World&) { return 0; }
}  // namespace ora
