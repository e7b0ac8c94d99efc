// cloth.hpp — cloth_component on the device (src/physics/cloth.h:5-60, cloth.cpp:7-317; SURVEY §8(f).4 tail).
//
// A cloth is a small independent system (a few thousand particles, ~6 constraints per particle) that does not interact with
// the rigid bodies: ONE workgroup steps one cloth, all phases of cloth_component::applyWindForce + simulate in one launch,
// workgroup barriers between the phases.  The reference's Gauss-Seidel passes walk the constraints in creation order; here a
// pass walks 12 colours (constraint family x parity of the creating vertex: constraints of one colour share no particle)
// and solves a colour in parallel — the order the oracle's ORDER_CANONICAL replays, so results are bit-identical to it.
// The wind is accumulated per VERTEX in the order the reference's per-quad loop reaches that vertex (<= 6 triangle terms),
// which reproduces its sums exactly without atomics.
#pragma once
#include "dmath.hpp"

namespace mi {

struct ClothDev {
    float4* pos; float4* prev; float4* vel; float4* force;   // xyz (+ pos.w = inverse mass)
    const uint2* pairs; float2* restInvMass;                 // per constraint: (a, b), (rest distance, inverse mass sum)
    float4* temp;                                            // velocity pass: (gradient, inverse scaled gradient squared)
    const uint32_t* order;                                   // constraint indices, colour-major
    uint32_t colourOffsets[13];
    uint32_t gridX, gridY, numConstraints;
    float gravityFactor, damping;
};

__device__ __forceinline__ V3 clothTriangleForce(V3 a, V3 b, V3 c, V3 wind) {   // cloth.cpp:155-171: normal * dot(normalize(normal), force) / 3
    V3 normal = cross(b - a, c - a);
    V3 f = normal * dot(normalize(normal), wind);
    return f * (1.f / 3.f);
}
// solvePositions over one colour range (cloth.cpp:277-297)
__device__ __forceinline__ void clothSolvePositions(const ClothDev& c) {
    for (uint32_t col = 0; col < 12u; ++col) {
        for (uint32_t k = c.colourOffsets[col] + threadIdx.x; k < c.colourOffsets[col + 1]; k += blockDim.x) {
            const uint32_t i = c.order[k];
            const float2 rm = c.restInvMass[i];
            if (rm.y > 0.f) {
                const uint2 ab = c.pairs[i];
                float4 pa = c.pos[ab.x], pb = c.pos[ab.y];
                V3 delta = xyz(pb) - xyz(pa);
                float len = sqlen(delta);
                float sqRest = rm.x * rm.x;
                if (sqRest + len > 1e-5f) {
                    float kk = ((sqRest - len) / (rm.y * (sqRest + len)));
                    c.pos[ab.x] = f4(xyz(pa) - delta * (kk * pa.w), pa.w);
                    c.pos[ab.y] = f4(xyz(pb) + delta * (kk * pb.w), pb.w);
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_cloth_step(const ClothDev* __restrict__ cloths, float3 windIn, float dt, uint32_t velocityIterations, uint32_t positionIterations,
                                                    uint32_t driftIterations) {
    const ClothDev c = cloths[blockIdx.x];
    const uint32_t n = c.gridX * c.gridY, gx = c.gridX, gy = c.gridY;
    const V3 wind(windIn.x, windIn.y, windIn.z);
    // ---- applyWindForce (cloth.cpp:139-174), gathered per vertex in the reference's accumulation order
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t vx = i % gx, vy = i / gx;
        V3 f = xyz(c.force[i]);
        for (uint32_t dy = 0; dy < 2u; ++dy)
            for (uint32_t dx = 0; dx < 2u; ++dx) {
                // quad (qx, qy) with qy = vy - 1 + dy, qx = vx - 1 + dx: scan order y-major, x-minor
                if (vy + dy < 1u || vx + dx < 1u) continue;
                const uint32_t qx = vx + dx - 1u, qy = vy + dy - 1u;
                if (qx + 1u >= gx || qy + 1u >= gy) continue;
                const uint32_t tl = qy * gx + qx, tr = tl + 1u, bl = tl + gx, br = bl + 1u;
                const V3 ptl = xyz(c.pos[tl]), ptr_ = xyz(c.pos[tr]), pbl = xyz(c.pos[bl]), pbr = xyz(c.pos[br]);
                if (i != br) f = f + clothTriangleForce(ptl, pbl, ptr_, wind);   // first triangle: tl, tr, bl
                if (i != tl) f = f + clothTriangleForce(pbr, ptr_, pbl, wind);   // second triangle: br, tr, bl
            }
        c.force[i] = f4(f, 0.f);
    }
    __syncthreads();
    // ---- simulate (cloth.cpp:182-275)
    const float gravityVelocity = kGravity * dt * c.gravityFactor;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        float4 p = c.pos[i];
        V3 v = xyz(c.vel[i]), force = xyz(c.force[i]);
        if (p.w > 0.f) v.y += gravityVelocity;
        v = v + force * (p.w * dt);
        c.prev[i] = p;
        c.pos[i] = f4(xyz(p) + v * dt, p.w);
        c.vel[i] = f4(v, 0.f);
        c.force[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const float invDt = (dt > 1e-5f) ? (1.f / dt) : 1.f;
    if (velocityIterations > 0u) {
        for (uint32_t i = threadIdx.x; i < c.numConstraints; i += blockDim.x) {
            const uint2 ab = c.pairs[i]; const float2 rm = c.restInvMass[i];
            V3 g = xyz(c.prev[ab.y]) - xyz(c.prev[ab.x]);
            c.temp[i] = f4(g, (rm.y == 0.f) ? 0.f : (1.f / (sqlen(g) * rm.y)));
        }
        __syncthreads();
        for (uint32_t it = 0; it < velocityIterations; ++it)
            for (uint32_t col = 0; col < 12u; ++col) {
                for (uint32_t k = c.colourOffsets[col] + threadIdx.x; k < c.colourOffsets[col + 1]; k += blockDim.x) {
                    const uint32_t i = c.order[k];
                    const uint2 ab = c.pairs[i];
                    const float4 t = c.temp[i];
                    V3 va = xyz(c.vel[ab.x]), vb = xyz(c.vel[ab.y]), g = xyz(t);
                    float j = -dot(g, va - vb) * t.w;
                    c.vel[ab.x] = f4(va + g * (j * c.pos[ab.x].w), 0.f);
                    c.vel[ab.y] = f4(vb - g * (j * c.pos[ab.y].w), 0.f);
                }
                __syncthreads();
            }
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { float4 p = c.pos[i]; c.pos[i] = f4(xyz(c.prev[i]) + xyz(c.vel[i]) * dt, p.w); }
        __syncthreads();
    }
    if (positionIterations > 0u) {
        for (uint32_t it = 0; it < positionIterations; ++it) clothSolvePositions(c);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) c.vel[i] = f4((xyz(c.pos[i]) - xyz(c.prev[i])) * invDt, 0.f);
        __syncthreads();
    }
    if (driftIterations > 0u) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) c.prev[i] = c.pos[i];
        __syncthreads();
        for (uint32_t it = 0; it < driftIterations; ++it) clothSolvePositions(c);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) c.vel[i] = f4(xyz(c.vel[i]) + (xyz(c.pos[i]) - xyz(c.prev[i])) * invDt, 0.f);
        __syncthreads();
    }
    const float dampingFactor = 1.f / (1.f + dt * c.damping);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) c.vel[i] = f4(xyz(c.vel[i]) * dampingFactor, 0.f);
}

}  // namespace mi
