// kernels_integrate.hpp — force and velocity integration, manifold keys.
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

namespace mi {

// ------------------------------------------------------------------------------------------------
// Integrator
// ------------------------------------------------------------------------------------------------
// K9 "Integrate rigid body forces" (src/physics/rigid_body.cpp:95-124).  One lane per body; also
// zeroes the dummy body (physics.cpp:1279).  in ~112 B, out 112 B per body.
// (Measured and not kept, round 3: as GUEST workgroups of k_emit_manifolds — nothing between the two depends on the other, that kernel waits on random
// sectors and atomics, this one streams; interleaved every 4th workgroup.  k_emit_manifolds 54 -> 68 us for the 21 us saved: they compete for the same
// memory system; 938 vs 937 steps/s.  Nor as guests of the colouring rounds — launch-floor kernels between which nothing reads what this one writes: a slice of
// the bodies per round made every round 9-10 us instead of 4.8 (the body rows are a chain of dependent gathers: ~5 us however few bodies), 8 x 5 us for the 19 saved:
// 962 vs 987 steps/s.  The same with k_manifold_keys / k_manifold_place as guests of rounds 0 / 1: 17.6 + 10.9 us for the two rounds, i.e. guest time + the round's own
// ~4.7 us — a kernel's launch floor is start-up and drain in series with its work, not a window other work can hide in.  A guest only pays inside a kernel whose OWN work
// outlasts it (the statistics workgroup of k_emit_manifolds).  Round 5: as every fourth of the first workgroups of k_narrow_clip — a kernel that computes, a third of its cycles
// issuing, 1 TB/s of traffic —: that kernel 63 -> 84 us, i.e. exactly the 21 us saved; 1039.2 vs 1040.5 steps/s.  Its three waves per SIMD are what hides its own LDS latency:
// a guest wave takes a slot, it does not fill a gap.)
struct ForcesArgs {   // k_integrate_forces' arguments; no padding bytes (the launcher hashes arguments bytewise)
    const float4* bPos; const float4* bRot; const float4* bCogInvMass; const float4* bInvI; const float4* bParams; const float4* bLinVel;
    const float4* bAngVel; const float4* bForce; const float4* bTorque; float4* gPos; float4* gInvI; float4* gVel;
    float4* gVelL;                    // XCD-partitioned solver: cached copy for the XCD-local bodies, or null
    unsigned long long* bodyOwner;    // ... and the per-body XCD flags (8 bytes), cleared here
    const uint8_t* bodyActive;        // sharded world, or null
    const uint8_t* blockLive;         // ... and its per-block summary (see shardBlockRecent above), or null
    uint32_t nb; float dt; float globalForce[3]; uint32_t pad;
};
static_assert(sizeof(ForcesArgs) == 16 * 8 + 24, "ForcesArgs must not contain padding");
__device__ __forceinline__ void integrateForcesBody(const uint32_t i, const ForcesArgs& fa) {
    const uint32_t nb = fa.nb; const float dt = fa.dt; const float3 globalForce = make_float3(fa.globalForce[0], fa.globalForce[1], fa.globalForce[2]);
    const float4* __restrict__ bPos = fa.bPos; const float4* __restrict__ bRot = fa.bRot; const float4* __restrict__ bCogInvMass = fa.bCogInvMass; const float4* __restrict__ bInvI = fa.bInvI;
    const float4* __restrict__ bParams = fa.bParams; const float4* __restrict__ bLinVel = fa.bLinVel; const float4* __restrict__ bAngVel = fa.bAngVel;
    const float4* __restrict__ bForce = fa.bForce; const float4* __restrict__ bTorque = fa.bTorque;
    float4* __restrict__ gPos = fa.gPos; float4* __restrict__ gInvI = fa.gInvI; float4* __restrict__ gVel = fa.gVel; float4* __restrict__ gVelL = fa.gVelL;
    unsigned long long* __restrict__ bodyOwner = fa.bodyOwner; const uint8_t* __restrict__ bodyActive = fa.bodyActive;
    if (i > nb) return;
    if (bodyActive && i < nb && !bodyActive[i]) return;   // not simulated by this rank: no contact can reference it (nor its XCD flags: they are only ever read for
                                                          // bodies of this step's contacts and for owned bodies, all of which pass here first)
    if (bodyOwner) bodyOwner[i] = 0ull;
    if (i == nb) {
        float4 z = make_float4(0, 0, 0, 0);
        gPos[i] = z; gInvI[3 * i] = z; gInvI[3 * i + 1] = z; gInvI[3 * i + 2] = z; gVel[2 * i] = z; gVel[2 * i + 1] = z;
        if (gVelL) { gVelL[2 * i] = z; gVelL[2 * i + 1] = z; }
        return;
    }
    Q4 rot = toQ(bRot[i]);
    float4 ci = bCogInvMass[i];
    V3 cog = xyz(ci); float invMass = ci.w;
    V3 pos = xyz(bPos[i]) + rotate(rot, cog);
    M3 R = quatToMat(rot);
    float4 i0 = bInvI[3 * i], i1 = bInvI[3 * i + 1], i2 = bInvI[3 * i + 2];
    M3 I; I.m00 = i0.x; I.m01 = i0.y; I.m02 = i0.z; I.m10 = i1.x; I.m11 = i1.y; I.m12 = i1.z; I.m20 = i2.x; I.m21 = i2.y; I.m22 = i2.z;
    M3 W = mul(mul(R, I), transpose(R));
    float4 prm = bParams[i];
    V3 force = xyz(bForce[i]), torque = xyz(bTorque[i]);
    force = force + V3(globalForce.x, globalForce.y, globalForce.z);   // rb.forceAccumulator += globalForceField (physics.cpp:1273)
    if (invMass > 0.f) force.y += (kGravity / invMass * prm.x);
    V3 linAcc = force * invMass;
    V3 angAcc = mul(W, torque);
    V3 v = xyz(bLinVel[i]), w = xyz(bAngVel[i]);
    v = v + linAcc * dt;
    w = w + angAcc * dt;
    v = v * (1.f / (1.f + dt * prm.y));
    w = w * (1.f / (1.f + dt * prm.z));
    // persistent body state is NOT touched before k_integrate_velocities: a step can be re-run from scratch
    gPos[i] = f4(pos, invMass);
    gInvI[3 * i] = make_float4(W.m00, W.m01, W.m02, 0.f);
    gInvI[3 * i + 1] = make_float4(W.m10, W.m11, W.m12, 0.f);
    gInvI[3 * i + 2] = make_float4(W.m20, W.m21, W.m22, 0.f);
    gVel[2 * i] = f4(v, 0.f); gVel[2 * i + 1] = f4(w, 0.f);   // .w = update-version tag of the solver (0 at step start)
    if (gVelL) { gVelL[2 * i] = f4(v, 0.f); gVelL[2 * i + 1] = f4(w, 0.f); }
}
// workgroup `first` of `stride` workgroups: the body blocks first, first + stride, ... (one block each unless the world is sharded)
template <bool STRIDED>
__device__ __forceinline__ void integrateForcesBlocks(const uint32_t first, const uint32_t stride, const ForcesArgs& fa) {
    const uint32_t numBlocks = (fa.nb + 1u + 255u) / 256u;
    forLiveBlocks<STRIDED>(first, stride, numBlocks,
                  [&](uint32_t blk) { return !fa.blockLive || blk + 1u >= numBlocks || fa.blockLive[blk] != 0u; },   // (nothing simulated in it, now or in the previous step: skipped; the last block holds the dummy body: always visited)
                  [&](uint32_t blk) { integrateForcesBody(blk * 256u + threadIdx.x, fa); });
}
template <bool STRIDED>
__global__ __launch_bounds__(256) void k_integrate_forces(ForcesArgs fa) { integrateForcesBlocks<STRIDED>(blockIdx.x, gridDim.x, fa); }

// K13 "Integrate rigid body velocities" (src/physics/rigid_body.cpp:126-142).
// Writes the NEXT body state into the second buffer set (the host swaps the sets once the step is known to be valid).
__device__ __forceinline__ void integrateVelocitiesBody(const uint32_t i, uint32_t nb, float dt, const float4* __restrict__ gPos, const float4* __restrict__ gVel,
                                                        const float4* __restrict__ bCogInvMass, const float4* __restrict__ bRotIn,
                                                        float4* __restrict__ bPos, float4* __restrict__ bRot,
                                                        float4* __restrict__ bLinVel, float4* __restrict__ bAngVel, float4* __restrict__ bForce,
                                                        float4* __restrict__ bTorque,
                                                        const float4* __restrict__ gVelL, const unsigned long long* __restrict__ bodyOwner /* XCD-partitioned solver, or null */,
                                                        unsigned long long* __restrict__ bodyUsed, unsigned long long* __restrict__ bodyTop,
                                                        const uint8_t* __restrict__ bodyActive /* sharded world (1 = owned), or null */, const float4* __restrict__ bPosIn,
                                                        const float4* __restrict__ bLinVelIn, const float4* __restrict__ bAngVelIn, const float4* __restrict__ bForceIn,
                                                        const float4* __restrict__ bTorqueIn,
                                                        const uint8_t* __restrict__ bodyActivePrev /* the previous step's flags */) {
    if (i > nb) return;
    // a body this rank neither simulates now nor simulated in the previous step: nothing of it was touched, both state sets already agree
    const bool idle = bodyActive && i < nb && bodyActive[i] == 0u && bodyActivePrev[i] == 0u;
    if (idle) return;
    // the per-body colouring scratch of the NEXT step starts out cleared (saves two memset launches per step); launched over nb + 1
    bodyUsed[i] = 0ull; bodyTop[i] = 0ull; bodyTop[(size_t)nb + 1u + i] = 0ull;
    if (i == nb) return;
    if (bodyActive && bodyActive[i] != 1u) {   // sharded world: only the OWNER advances a body; ghosts and bodies elsewhere keep their state (the owner's arrives by exchange)
        bPos[i] = bPosIn[i]; bRot[i] = bRotIn[i]; bLinVel[i] = bLinVelIn[i]; bAngVel[i] = bAngVelIn[i]; bForce[i] = bForceIn[i]; bTorque[i] = bTorqueIn[i];
        return;
    }
    if (bodyOwner && __popcll(bodyOwner[i]) == 1) gVel = gVelL;   // a body only one XCD touched lives in the cached copy
    V3 v = xyz(gVel[2 * i]), w = xyz(gVel[2 * i + 1]);
    Q4 rot = toQ(bRotIn[i]);
    Q4 dq(0.5f * w.x, 0.5f * w.y, 0.5f * w.z, 0.f);
    dq = dq * rot;
    Q4 nr = normalize(Q4(rot.x + dq.x * dt, rot.y + dq.y * dt, rot.z + dq.z * dt, rot.w + dq.w * dt));
    V3 pos = xyz(gPos[i]) + v * dt;
    V3 cog = xyz(bCogInvMass[i]);
    bLinVel[i] = f4(v, 0.f); bAngVel[i] = f4(w, 0.f);
    float4 z = make_float4(0, 0, 0, 0);
    bForce[i] = z; bTorque[i] = z;
    bRot[i] = fromQ(nr);
    bPos[i] = f4(pos - rotate(nr, cog), 0.f);
}
template <bool STRIDED>
__global__ __launch_bounds__(256) void k_integrate_velocities(uint32_t nb, float dt, const float4* __restrict__ gPos, const float4* __restrict__ gVel,
                                                              const float4* __restrict__ bCogInvMass, const float4* __restrict__ bRotIn,
                                                              float4* __restrict__ bPos, float4* __restrict__ bRot,
                                                              float4* __restrict__ bLinVel, float4* __restrict__ bAngVel, float4* __restrict__ bForce,
                                                              float4* __restrict__ bTorque,
                                                              const float4* __restrict__ gVelL, const unsigned long long* __restrict__ bodyOwner /* XCD-partitioned solver, or null */,
                                                              unsigned long long* __restrict__ bodyUsed, unsigned long long* __restrict__ bodyTop,
                                                              const uint8_t* __restrict__ bodyActive /* sharded world (1 = owned), or null */, const float4* __restrict__ bPosIn,
                                                              const float4* __restrict__ bLinVelIn, const float4* __restrict__ bAngVelIn, const float4* __restrict__ bForceIn,
                                                              const float4* __restrict__ bTorqueIn,
                                                              const uint8_t* __restrict__ bodyActivePrev /* the previous step's flags */, const Shards* __restrict__ sh, StepScalars* sc,
                                                              const uint8_t* __restrict__ blockLive /* sharded world: body blocks with a body simulated in this step or the previous one (the others are skipped), or null */) {
    if (bodyActive && blockIdx.x == 0 && threadIdx.x < 3) {   // sharded world: this rank's owned bodies / manifolds / contacts, from the per-line counters
        uint32_t v = 0; for (uint32_t k = 0; k < kShards; ++k) v += sh->c[k].owned[threadIdx.x];
        sc->shardOwned[threadIdx.x] = v;
    }
    const uint32_t numBlocks = (nb + 1u + 255u) / 256u;
    forLiveBlocks<STRIDED>(blockIdx.x, gridDim.x, numBlocks,   // (one block per workgroup unless the world is sharded)
                  [&](uint32_t blk) { return !blockLive || blk + 1u >= numBlocks || blockLive[blk] != 0u; },   // (the last block holds the dummy body: always visited)
                  [&](uint32_t blk) {
        integrateVelocitiesBody(blk * 256u + threadIdx.x, nb, dt, gPos, gVel, bCogInvMass, bRotIn, bPos, bRot, bLinVel, bAngVel, bForce, bTorque, gVelL, bodyOwner, bodyUsed, bodyTop,
                                bodyActive, bPosIn, bLinVelIn, bAngVelIn, bForceIn, bTorqueIn, bodyActivePrev);
    });
}

__global__ __launch_bounds__(256) void k_iota(uint32_t n, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// Ghost-region exchange: pack / unpack 13-float body states (pos3, rot4, lin3, ang3) by body index.
__global__ __launch_bounds__(256) void k_gather_states(uint32_t n, const uint32_t* __restrict__ ids, const float4* __restrict__ bPos,
                                                       const float4* __restrict__ bRot, const float4* __restrict__ bLinVel,
                                                       const float4* __restrict__ bAngVel, float* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t b = ids[i];
    float4 p = bPos[b], q = bRot[b], v = bLinVel[b], w = bAngVel[b];
    float* o = out + 13 * (size_t)i;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    o[7] = v.x; o[8] = v.y; o[9] = v.z; o[10] = w.x; o[11] = w.y; o[12] = w.z;
}
__global__ __launch_bounds__(256) void k_scatter_states(uint32_t n, const uint32_t* __restrict__ ids, const float* __restrict__ in,
                                                        float4* __restrict__ bPos, float4* __restrict__ bRot, float4* __restrict__ bLinVel,
                                                        float4* __restrict__ bAngVel, uint8_t* __restrict__ shardKnown /* sharded world: the caller's state is authoritative; or null */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t b = ids[i];
    if (shardKnown) shardKnown[b] = 1u;
    const float* s = in + 13 * (size_t)i;
    bPos[b] = make_float4(s[0], s[1], s[2], 0.f); bRot[b] = make_float4(s[3], s[4], s[5], s[6]);
    bLinVel[b] = make_float4(s[7], s[8], s[9], 0.f); bAngVel[b] = make_float4(s[10], s[11], s[12], 0.f);
}

}  // namespace mi
