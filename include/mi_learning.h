/*
 * mi_learning.h — the learning DLL's C ABI (libPhysics-Lib.so), SURVEY §8(b)(2) / §8(f).2.
 *
 * The first five functions are, name for name and signature for signature, the reference's only true C ABI:
 * the __declspec(dllexport) functions of src/learning/learned_locomotion.cpp:395-489 that learning/loco_env.py:8-47 binds
 * with ctypes (Physics-Lib.dll).  State = learned_locomotion::learning_state (66 floats), action = learning_action
 * (27 floats), src/learning/learned_locomotion.h:15-65.  The rest steps many such environments per call in ONE device world.
 * Implementation: d3d12renderer_amd/csrc/learning.cpp (host C++ over include/mi_physics.h).
 */
#ifndef MI_LEARNING_H
#define MI_LEARNING_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef MI_LEARNING_API
#define MI_LEARNING_API
#endif

/* learned_locomotion.cpp:395-396 */
MI_LEARNING_API int getPhysicsStateSize(void);
MI_LEARNING_API int getPhysicsActionSize(void);
/* learned_locomotion.cpp:398-427: state unbounded (-FLT_MAX / FLT_MAX); action bounds from the joint limits. */
MI_LEARNING_API void getPhysicsRanges(float* stateMin, float* stateMax, float* actionMin, float* actionMax);
/* learned_locomotion.cpp:429-450: ground slab + humanoid at hip height 1.25 m, episode state reset.  Unlike the reference it
 * also writes the initial state to outState (may be NULL). */
MI_LEARNING_API void resetPhysics(float* outState);
/* learned_locomotion.cpp:452-489: smooth + apply the action (position motors, 200 Nm), a random push with probability 0.02,
 * one physicsStep at 60 Hz, state, reward; returns 1 when the ragdoll has fallen (head below 1 m; reward 0 then). */
MI_LEARNING_API int updatePhysics(float* action, float* outState, float* outReward);

/* ---- batched environments: numEnvs ragdolls, each on its own ground slab, in one world on one GPU ---- */
/* (Re)creates the environments, writes [numEnvs][stateSize] initial states (may be NULL); 0 on success. */
MI_LEARNING_API int resetPhysicsBatch(int numEnvs, float* outStates);
/* actions [numEnvs][actionSize] -> states [numEnvs][stateSize], rewards [numEnvs], done [numEnvs]; an environment whose
 * ragdoll fell is reset in place after its terminal state was written; 0 on success. */
MI_LEARNING_API int updatePhysicsBatch(const float* actions, float* outStates, float* outRewards, int* outDone);
MI_LEARNING_API int getPhysicsNumEnvs(void);
MI_LEARNING_API void setPhysicsSeed(unsigned long long seed);   /* push RNG (the reference seeds with time(0)) */
MI_LEARNING_API void setPhysicsDevice(int device);              /* HIP device of the next (re)created world */
MI_LEARNING_API const char* getPhysicsError(void);              /* message of the last failure */
MI_LEARNING_API void shutdownPhysics(void);

#ifdef __cplusplus
}
#endif
#endif
