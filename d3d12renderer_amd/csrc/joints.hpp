// joints.hpp — joint constraints on the device: distance, ball, fixed, hinge, cone-twist, slider.
//
// Behavioural spec: the scalar initialize/solve routines of src/physics/constraints.cpp (distance 189-264,
// ball 460-528, fixed 736-823, hinge 1079-1307, cone-twist 1782-2070, slider 2638-2846) with PER-CONSTRAINT
// limit/motor gating (the reference's AVX2 path gates per batch of 8; SURVEY.md §8(a) explains why the scalar
// semantics are the ones to keep).  One lane per joint; joints of a type are greedily coloured on the host at
// upload (topology is static) so a colour's lanes own disjoint dynamic bodies; per iteration the types run in
// the reference's order distance -> ball -> fixed -> hinge -> cone-twist -> slider (constraints.cpp:3764-3769).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <algorithm>
#include <cstring>
#include "../../include/mi_physics.h"
#include "../../include/mi_constraints.h"
#include "dmath.hpp"

namespace mi {

constexpr float kBetaDistance = 0.1f, kBetaBall = 0.1f, kBetaSlider = 0.1f, kBetaHingeRot = 0.3f, kBetaHingeLimit = 0.1f,
                kBetaTwistLimit = 0.1f, kBetaSliderLimit = 0.1f, kDtThreshold = 1e-5f;   // constraints.cpp:9-17

struct BodyView { const float4* gPos; const float4* gInvI; float4* gVel; const float4* bRot; const float4* bCog; const uint8_t* active /* sharded world: 0 = body (and so its whole island) is not simulated by this rank; or null */; };
struct BodyState { Q4 rot; V3 cog, pos; M3 invI; float invMass; };
struct BodyVel { V3 v, w; float invMass; M3 invI; float tagV, tagW; };   // tags: the contact solver's update-version words in gVel[].w, carried through untouched

__device__ __forceinline__ M3 ldM3(const float4* __restrict__ p, uint32_t i) {
    float4 a = p[3 * i], b = p[3 * i + 1], c = p[3 * i + 2];
    M3 m; m.m00 = a.x; m.m01 = a.y; m.m02 = a.z; m.m10 = b.x; m.m11 = b.y; m.m12 = b.z; m.m20 = c.x; m.m21 = c.y; m.m22 = c.z;
    return m;
}
__device__ __forceinline__ BodyState loadState(const BodyView& bv, uint32_t i, uint32_t dummy) {
    BodyState s;
    float4 p = bv.gPos[i];
    s.pos = xyz(p); s.invMass = p.w; s.invI = ldM3(bv.gInvI, i);
    if (i < dummy) { s.rot = toQ(bv.bRot[i]); s.cog = xyz(bv.bCog[i]); } else { s.rot = Q4(0.f, 0.f, 0.f, 0.f); s.cog = V3(); }
    return s;
}
__device__ __forceinline__ BodyVel loadVel(const BodyView& bv, uint32_t i) {
    BodyVel b; float4 a = bv.gVel[2 * i], c = bv.gVel[2 * i + 1];
    b.v = xyz(a); b.invMass = bv.gPos[i].w; b.w = xyz(c); b.invI = ldM3(bv.gInvI, i);
    b.tagV = a.w; b.tagW = c.w;
    return b;
}
__device__ __forceinline__ void storeVel(const BodyView& bv, uint32_t i, const BodyVel& b) {
    if (b.invMass == 0.f) return;   // kinematic bodies are never changed by impulses
    bv.gVel[2 * i] = f4(b.v, b.tagV); bv.gVel[2 * i + 1] = f4(b.w, b.tagW);
}
__device__ __forceinline__ V3 ld3(const float* f) { return V3(f[0], f[1], f[2]); }
__device__ __forceinline__ Q4 ld4(const float* f) { return Q4(f[0], f[1], f[2], f[3]); }
__device__ __forceinline__ float inv0(float x) { return (x != 0.f) ? (1.f / x) : 0.f; }
__device__ __forceinline__ M3 ballInvEffMass(const BodyState& A, const BodyState& B, V3 rA, V3 rB) {
    M3 sA = skew(rA), sB = skew(rB);
    return add(add(mul(mul(sA, A.invI), transpose(sA)), mul(mul(sB, B.invI), transpose(sB))), scale(M3::identity(), A.invMass + B.invMass));
}
__device__ __forceinline__ void applyPoint(BodyVel& A, BodyVel& B, V3 rA, V3 rB, V3 P) {   // the ball / position block shared by 5 joint types
    A.v = A.v - A.invMass * P;
    A.w = A.w - mul(A.invI, cross(rA, P));
    B.v = B.v + B.invMass * P;
    B.w = B.w + mul(B.invI, cross(rB, P));
}
struct F2 { float x, y; };

// ------------------------------------------------------------------------------------------------ distance
struct DistanceJ {
    typedef mi_distance_constraint Pod;
    struct Upd { V3 rA, rB, iwA, iwB, u; float bias, effMass; };
    // impulses accumulated ACROSS sweeps (what a later sweep must see of an earlier one): none — every sweep is a plain velocity solve
    static constexpr int kAcc = 0;
    __device__ static void getAcc(const Upd&, float*) {}
    __device__ static void setAcc(Upd&, const float*) {}
    __device__ static void init(const Pod& in, const BodyState& A, const BodyState& B, float dt, Upd& o) {
        float invDt = 1.f / dt;
        o.rA = rotate(A.rot, ld3(in.local_anchor_a) - A.cog);
        o.rB = rotate(B.rot, ld3(in.local_anchor_b) - B.cog);
        V3 gA = A.pos + o.rA, gB = B.pos + o.rB;
        o.u = gB - gA;
        float l = len(o.u);
        o.u = (l > 0.001f) ? (o.u * (1.f / l)) : V3();
        V3 crAu = cross(o.rA, o.u), crBu = cross(o.rB, o.u);
        float im = A.invMass + dot(crAu, mul(A.invI, crAu)) + B.invMass + dot(crBu, mul(B.invI, crBu));
        o.effMass = inv0(im);
        o.bias = 0.f;
        if (dt > kDtThreshold) o.bias = (l - in.global_length) * (kBetaDistance * invDt);
        o.iwA = mul(A.invI, cross(o.rA, crAu));
        o.iwB = mul(B.invI, cross(o.rB, crBu));
    }
    __device__ static void solve(Upd& c, BodyVel& A, BodyVel& B) {
        V3 avA = A.v + cross(A.w, c.rA), avB = B.v + cross(B.w, c.rB);
        float Cdot = dot(c.u, avB - avA) + c.bias;
        float lambda = -c.effMass * Cdot;
        V3 P = lambda * c.u;
        A.v = A.v - A.invMass * P;
        A.w = A.w - c.iwA * lambda;
        B.v = B.v + B.invMass * P;
        B.w = B.w + c.iwB * lambda;
    }
};

// ------------------------------------------------------------------------------------------------ ball
struct BallJ {
    typedef mi_ball_constraint Pod;
    struct Upd { V3 rA, rB, bias; M3 invEff; };
    // impulses accumulated ACROSS sweeps (what a later sweep must see of an earlier one): none — every sweep is a plain velocity solve
    static constexpr int kAcc = 0;
    __device__ static void getAcc(const Upd&, float*) {}
    __device__ static void setAcc(Upd&, const float*) {}
    __device__ static void init(const Pod& in, const BodyState& A, const BodyState& B, float dt, Upd& o) {
        float invDt = 1.f / dt;
        o.rA = rotate(A.rot, ld3(in.local_anchor_a) - A.cog);
        o.rB = rotate(B.rot, ld3(in.local_anchor_b) - B.cog);
        V3 gA = A.pos + o.rA, gB = B.pos + o.rB;
        o.invEff = ballInvEffMass(A, B, o.rA, o.rB);
        o.bias = V3();
        if (dt > kDtThreshold) o.bias = (gB - gA) * (kBetaBall * invDt);
    }
    __device__ static void solve(Upd& c, BodyVel& A, BodyVel& B) {
        V3 avA = A.v + cross(A.w, c.rA), avB = B.v + cross(B.w, c.rB);
        V3 Cdot = avB - avA + c.bias;
        applyPoint(A, B, c.rA, c.rB, solve3(c.invEff, -Cdot));
    }
};

// ------------------------------------------------------------------------------------------------ fixed
struct FixedJ {
    typedef mi_fixed_constraint Pod;
    struct Upd { V3 rA, rB, tBias, rBias; M3 invEffT, invEffR; };
    // impulses accumulated ACROSS sweeps (what a later sweep must see of an earlier one): none — every sweep is a plain velocity solve
    static constexpr int kAcc = 0;
    __device__ static void getAcc(const Upd&, float*) {}
    __device__ static void setAcc(Upd&, const float*) {}
    __device__ static void init(const Pod& in, const BodyState& A, const BodyState& B, float dt, Upd& o) {
        float invDt = 1.f / dt;
        o.rA = rotate(A.rot, ld3(in.local_anchor_a) - A.cog);
        o.rB = rotate(B.rot, ld3(in.local_anchor_b) - B.cog);
        V3 gA = A.pos + o.rA, gB = B.pos + o.rB;
        o.invEffT = ballInvEffMass(A, B, o.rA, o.rB);
        o.invEffR = add(A.invI, B.invI);
        o.tBias = V3(); o.rBias = V3();
        if (dt > kDtThreshold) {
            o.tBias = (gB - gA) * (kBetaBall * invDt);
            Q4 err = B.rot * ld4(in.initial_inv_rotation_difference) * conj(A.rot);
            o.rBias = err.v() * (kBetaSlider * invDt * 2.f);
        }
    }
    __device__ static void solve(Upd& c, BodyVel& A, BodyVel& B) {
        {
            V3 Cdot = B.w - A.w;
            V3 rl = solve3(c.invEffR, -(Cdot + c.rBias));
            A.w = A.w - mul(A.invI, rl);
            B.w = B.w + mul(B.invI, rl);
        }
        {
            V3 avA = A.v + cross(A.w, c.rA), avB = B.v + cross(B.w, c.rB);
            V3 Cdot = avB - avA + c.tBias;
            applyPoint(A, B, c.rA, c.rB, solve3(c.invEffT, -Cdot));
        }
    }
};

// ------------------------------------------------------------------------------------------------ hinge
struct HingeJ {
    typedef mi_hinge_constraint Pod;
    struct Upd {
        V3 rA, rB, tBias; M3 invEffT; V3 bxa, cxa; float r00, r01, r10, r11; F2 rBias;
        uint32_t solveLimit, solveMotor; V3 axis; float limitImpulse, effAxial, limitSign, maxMotorImpulse, motorImpulse, motorVelocity, limitBias;
        V3 mlA, mlB;
    };
    static constexpr int kAcc = 2;   // the clamped accumulators: limit and motor impulse
    __device__ static void getAcc(const Upd& c, float* a) { a[0] = c.limitImpulse; a[1] = c.motorImpulse; }
    __device__ static void setAcc(Upd& c, const float* a) { c.limitImpulse = a[0]; c.motorImpulse = a[1]; }
    __device__ static void init(const Pod& in, const BodyState& A, const BodyState& B, float dt, Upd& o) {
        float invDt = 1.f / dt;
        o.rA = rotate(A.rot, ld3(in.local_anchor_a) - A.cog);
        o.rB = rotate(B.rot, ld3(in.local_anchor_b) - B.cog);
        V3 gA = A.pos + o.rA, gB = B.pos + o.rB;
        o.invEffT = ballInvEffMass(A, B, o.rA, o.rB);
        o.tBias = V3();
        if (dt > kDtThreshold) o.tBias = (gB - gA) * (kBetaBall * invDt);
        V3 axA = rotate(A.rot, ld3(in.local_hinge_axis_a)), axB = rotate(B.rot, ld3(in.local_hinge_axis_b));
        V3 tB = tangentOf(axB), btB = cross(axB, tB);
        V3 bxa = cross(tB, axA), cxa = cross(btB, axA);
        V3 iAbxa = mul(A.invI, bxa), iBbxa = mul(B.invI, bxa), iAcxa = mul(A.invI, cxa), iBcxa = mul(B.invI, cxa);
        o.r00 = dot(bxa, iAbxa) + dot(bxa, iBbxa);
        o.r01 = dot(bxa, iAcxa) + dot(bxa, iBcxa);
        o.r10 = dot(cxa, iAbxa) + dot(cxa, iBbxa);
        o.r11 = dot(cxa, iAcxa) + dot(cxa, iBcxa);
        o.bxa = bxa; o.cxa = cxa;
        o.rBias.x = 0.f; o.rBias.y = 0.f;
        if (dt > kDtThreshold) { float k = kBetaHingeRot * invDt; o.rBias.x = dot(axA, tB) * k; o.rBias.y = dot(axA, btB) * k; }
        o.solveLimit = 0; o.solveMotor = 0; o.axis = V3();
        o.limitImpulse = o.effAxial = o.limitSign = o.maxMotorImpulse = o.motorImpulse = o.motorVelocity = o.limitBias = 0.f;
        o.mlA = V3(); o.mlB = V3();
        if (in.min_rotation_limit <= 0.f || in.max_rotation_limit >= 0.f || in.max_motor_torque > 0.f) {
            V3 cmp = rotate(conj(A.rot), rotate(B.rot, ld3(in.local_hinge_tangent_b)));
            float angle = detAtan2(dot(cmp, ld3(in.local_hinge_bitangent_a)), dot(cmp, ld3(in.local_hinge_tangent_a)));
            bool minV = in.min_rotation_limit <= 0.f && angle <= in.min_rotation_limit;
            bool maxV = in.max_rotation_limit >= 0.f && angle >= in.max_rotation_limit;
            o.solveLimit = (minV || maxV) ? 1u : 0u;
            o.solveMotor = in.max_motor_torque > 0.f ? 1u : 0u;
            if (o.solveLimit || o.solveMotor) {
                o.axis = axA;
                float invAx = dot(axA, mul(A.invI, axA)) + dot(axA, mul(B.invI, axA));
                o.effAxial = inv0(invAx);
                o.limitSign = minV ? 1.f : -1.f;
                o.maxMotorImpulse = in.max_motor_torque * dt;
                o.mlA = mul(A.invI, o.axis); o.mlB = mul(B.invI, o.axis);
                o.motorVelocity = in.motor_velocity_or_target_angle;
                if (in.motor_type == MI_MOTOR_POSITION) {
                    float minL = (in.min_rotation_limit <= 0.f) ? in.min_rotation_limit : -kPi;
                    float maxL = (in.max_rotation_limit >= 0.f) ? in.max_rotation_limit : kPi;
                    float target = clampr(in.motor_velocity_or_target_angle, minL, maxL);
                    o.motorVelocity = (dt > kDtThreshold) ? ((target - angle) * invDt) : 0.f;
                }
                if (dt > kDtThreshold) {
                    float d = minV ? (angle - in.min_rotation_limit) : (in.max_rotation_limit - angle);
                    o.limitBias = d * kBetaHingeLimit * invDt;
                }
            }
        }
    }
    __device__ static void solve(Upd& c, BodyVel& A, BodyVel& B) {
        V3 axis = c.axis;
        if (c.solveMotor) {
            float aA = dot(axis, A.w), aB = dot(axis, B.w);
            float rel = (aB - aA);
            float cd = rel - c.motorVelocity;
            float l = -c.effAxial * cd;
            float old = c.motorImpulse;
            c.motorImpulse = clampr(c.motorImpulse + l, -c.maxMotorImpulse, c.maxMotorImpulse);
            l = c.motorImpulse - old;
            A.w = A.w - c.mlA * l;
            B.w = B.w + c.mlB * l;
        }
        if (c.solveLimit) {
            float s = c.limitSign;
            float aA = dot(axis, A.w), aB = dot(axis, B.w);
            float rel = s * (aB - aA);
            float cd = rel + c.limitBias;
            float l = -c.effAxial * cd;
            float imp = fmaxr(c.limitImpulse + l, 0.f);
            l = imp - c.limitImpulse;
            c.limitImpulse = imp;
            l *= s;
            A.w = A.w - c.mlA * l;
            B.w = B.w + c.mlB * l;
        }
        {
            V3 dw = B.w - A.w;
            float cx = dot(c.bxa, dw), cy = dot(c.cxa, dw);
            float sx = cx + c.rBias.x, sy = cy + c.rBias.y;
            float lx, ly;
            solve2(c.r00, c.r01, c.r10, c.r11, -sx, -sy, lx, ly);
            V3 P = c.bxa * lx + c.cxa * ly;
            A.w = A.w - mul(A.invI, P);
            B.w = B.w + mul(B.invI, P);
        }
        {
            V3 avA = A.v + cross(A.w, c.rA), avB = B.v + cross(B.w, c.rB);
            V3 cd = avB - avA + c.tBias;
            applyPoint(A, B, c.rA, c.rB, solve3(c.invEffT, -cd));
        }
    }
};

// ------------------------------------------------------------------------------------------------ cone twist
struct ConeJ {
    typedef mi_cone_twist_constraint Pod;
    struct Upd {
        V3 rA, rB, bias; M3 invEff;
        uint32_t solveSwingLimit, solveSwingMotor, solveTwistLimit, solveTwistMotor;
        float swingImpulse; V3 swingAxis; float effSwingLimit, swingLimitBias; V3 slA, slB;
        float maxSwingMotorImpulse, swingMotorImpulse, swingMotorVelocity, effSwingMotor; V3 swingMotorAxis, smA, smB;
        float twistImpulse; V3 twistAxis; float effTwist, twistLimitSign, maxTwistMotorImpulse, twistMotorImpulse, twistMotorVelocity, twistLimitBias;
        V3 tmA, tmB;
    };
    static constexpr int kAcc = 4;
    __device__ static void getAcc(const Upd& c, float* a) { a[0] = c.swingImpulse; a[1] = c.swingMotorImpulse; a[2] = c.twistImpulse; a[3] = c.twistMotorImpulse; }
    __device__ static void setAcc(Upd& c, const float* a) { c.swingImpulse = a[0]; c.swingMotorImpulse = a[1]; c.twistImpulse = a[2]; c.twistMotorImpulse = a[3]; }
    __device__ static void init(const Pod& in, const BodyState& A, const BodyState& B, float dt, Upd& o) {
        float invDt = 1.f / dt;
        o.solveSwingLimit = o.solveSwingMotor = o.solveTwistLimit = o.solveTwistMotor = 0;
        o.swingImpulse = o.effSwingLimit = o.swingLimitBias = 0.f; o.swingAxis = V3(); o.slA = V3(); o.slB = V3();
        o.maxSwingMotorImpulse = o.swingMotorImpulse = o.swingMotorVelocity = o.effSwingMotor = 0.f; o.swingMotorAxis = V3(); o.smA = V3(); o.smB = V3();
        o.twistImpulse = o.effTwist = o.twistLimitSign = o.maxTwistMotorImpulse = o.twistMotorImpulse = o.twistMotorVelocity = o.twistLimitBias = 0.f;
        o.twistAxis = V3(); o.tmA = V3(); o.tmB = V3();
        o.rA = rotate(A.rot, ld3(in.local_anchor_a) - A.cog);
        o.rB = rotate(B.rot, ld3(in.local_anchor_b) - B.cog);
        V3 gA = A.pos + o.rA, gB = B.pos + o.rB;
        o.invEff = ballInvEffMass(A, B, o.rA, o.rB);
        o.bias = V3();
        if (dt > kDtThreshold) o.bias = (gB - gA) * (kBetaBall * invDt);
        Q4 btoa = conj(A.rot) * B.rot;
        V3 axisA = ld3(in.local_limit_axis_a);
        V3 axisCmpA = rotate(btoa, ld3(in.local_limit_axis_b));
        Q4 swingRot = rotateFromTo(axisA, axisCmpA);
        V3 twT = rotate(swingRot, ld3(in.local_limit_tangent_a));
        V3 twB = rotate(swingRot, ld3(in.local_limit_bitangent_a));
        V3 tanCmpA = rotate(btoa, ld3(in.local_limit_tangent_b));
        float twistAngle = detAtan2(dot(tanCmpA, twB), dot(tanCmpA, twT));
        V3 swingAxis; float swingAngle;
        axisRotation(swingRot, swingAxis, swingAngle);
        if (swingAngle < 0.f) { swingAngle *= -1.f; swingAxis = swingAxis * -1.f; }
        o.solveSwingLimit = (in.swing_limit >= 0.f && swingAngle >= in.swing_limit) ? 1u : 0u;
        if (o.solveSwingLimit) {
            o.swingAxis = rotate(A.rot, swingAxis);
            float im = dot(o.swingAxis, mul(A.invI, o.swingAxis)) + dot(o.swingAxis, mul(B.invI, o.swingAxis));
            o.effSwingLimit = inv0(im);
            if (dt > kDtThreshold) o.swingLimitBias = (in.swing_limit - swingAngle) * (kBetaHingeLimit * invDt);
            o.slA = mul(A.invI, o.swingAxis); o.slB = mul(B.invI, o.swingAxis);
        }
        o.solveSwingMotor = in.max_swing_motor_torque > 0.f ? 1u : 0u;
        if (o.solveSwingMotor) {
            o.maxSwingMotorImpulse = in.max_swing_motor_torque * dt;
            float axisX, axisY;
            detSinCos(in.swing_motor_axis, axisY, axisX);
            V3 localMotorAxis = axisX * ld3(in.local_limit_tangent_a) + axisY * ld3(in.local_limit_bitangent_a);
            if (in.swing_motor_type == MI_MOTOR_VELOCITY) {
                o.swingMotorAxis = rotate(A.rot, localMotorAxis);
                o.swingMotorVelocity = in.swing_motor_velocity_or_target_angle;
            } else {
                float target = in.swing_motor_velocity_or_target_angle;
                if (in.swing_limit >= 0.f) target = clampr(target, -in.swing_limit, in.swing_limit);
                float sh, ch;
                detSinCos(target * 0.5f, sh, ch);
                Q4 tq(localMotorAxis.x * sh, localMotorAxis.y * sh, localMotorAxis.z * sh, ch);
                V3 localTargetDir = rotate(tq, axisA);
                V3 localMotorAxis2 = noz(cross(axisCmpA, localTargetDir));
                o.swingMotorAxis = rotate(A.rot, localMotorAxis2);
                float cosAngle = dot(localTargetDir, axisCmpA);
                float deltaAngle = detAcos(clamp01(cosAngle));
                o.swingMotorVelocity = (dt > kDtThreshold) ? (deltaAngle * invDt * 0.2f) : 0.f;
            }
            o.smA = mul(A.invI, o.swingMotorAxis); o.smB = mul(B.invI, o.swingMotorAxis);
            float im = dot(o.swingMotorAxis, mul(A.invI, o.swingMotorAxis)) + dot(o.swingMotorAxis, mul(B.invI, o.swingMotorAxis));
            o.effSwingMotor = inv0(im);
        }
        bool minTw = in.twist_limit >= 0.f && twistAngle <= -in.twist_limit;
        bool maxTw = in.twist_limit >= 0.f && twistAngle >= in.twist_limit;
        o.solveTwistLimit = (minTw || maxTw) ? 1u : 0u;
        o.solveTwistMotor = in.max_twist_motor_torque > 0.f ? 1u : 0u;
        if (o.solveTwistLimit || o.solveTwistMotor) {
            o.twistAxis = rotate(A.rot, axisA);
            float im = dot(o.twistAxis, mul(A.invI, o.twistAxis)) + dot(o.twistAxis, mul(B.invI, o.twistAxis));
            o.effTwist = inv0(im);
            o.twistLimitSign = minTw ? 1.f : -1.f;
            o.maxTwistMotorImpulse = in.max_twist_motor_torque * dt;
            o.tmA = mul(A.invI, o.twistAxis); o.tmB = mul(B.invI, o.twistAxis);
            o.twistMotorVelocity = in.twist_motor_velocity_or_target_angle;
            if (in.twist_motor_type == MI_MOTOR_POSITION) {
                float limit = (in.twist_limit >= 0.f) ? in.twist_limit : kPi;
                float target = clampr(in.twist_motor_velocity_or_target_angle, -limit, limit);
                o.twistMotorVelocity = (dt > kDtThreshold) ? ((target - twistAngle) * invDt) : 0.f;
            }
            if (dt > kDtThreshold) {
                float d = minTw ? (in.twist_limit + twistAngle) : (in.twist_limit - twistAngle);
                o.twistLimitBias = d * kBetaTwistLimit * invDt;
            }
        }
    }
    __device__ static void solve(Upd& c, BodyVel& A, BodyVel& B) {
        V3 tw = c.twistAxis;
        if (c.solveTwistMotor) {
            float aA = dot(tw, A.w), aB = dot(tw, B.w);
            float rel = (aB - aA);
            float cd = rel - c.twistMotorVelocity;
            float l = -c.effTwist * cd;
            float old = c.twistMotorImpulse;
            c.twistMotorImpulse = clampr(c.twistMotorImpulse + l, -c.maxTwistMotorImpulse, c.maxTwistMotorImpulse);
            l = c.twistMotorImpulse - old;
            A.w = A.w - c.tmA * l;
            B.w = B.w + c.tmB * l;
        }
        if (c.solveSwingMotor) {
            V3 ax = c.swingMotorAxis;
            float aA = dot(ax, A.w), aB = dot(ax, B.w);
            float rel = (aB - aA);
            float cd = rel - c.swingMotorVelocity;
            float l = -c.effSwingMotor * cd;
            float old = c.swingMotorImpulse;
            c.swingMotorImpulse = clampr(c.swingMotorImpulse + l, -c.maxSwingMotorImpulse, c.maxSwingMotorImpulse);
            l = c.swingMotorImpulse - old;
            A.w = A.w - c.smA * l;
            B.w = B.w + c.smB * l;
        }
        if (c.solveTwistLimit) {
            float s = c.twistLimitSign;
            float aA = dot(tw, A.w), aB = dot(tw, B.w);
            float rel = s * (aB - aA);
            float cd = rel + c.twistLimitBias;
            float l = -c.effTwist * cd;
            float imp = fmaxr(c.twistImpulse + l, 0.f);
            l = imp - c.twistImpulse;
            c.twistImpulse = imp;
            l *= s;
            A.w = A.w - c.tmA * l;
            B.w = B.w + c.tmB * l;
        }
        if (c.solveSwingLimit) {
            float aA = dot(c.swingAxis, A.w), aB = dot(c.swingAxis, B.w);
            float cd = aA - aB + c.swingLimitBias;
            float l = -c.effSwingLimit * cd;
            float imp = fmaxr(c.swingImpulse + l, 0.f);
            l = imp - c.swingImpulse;
            c.swingImpulse = imp;
            A.w = A.w + c.slA * l;
            B.w = B.w - c.slB * l;
        }
        {
            V3 avA = A.v + cross(A.w, c.rA), avB = B.v + cross(B.w, c.rB);
            V3 cd = avB - avA + c.bias;
            applyPoint(A, B, c.rA, c.rB, solve3(c.invEff, -cd));
        }
    }
};

// ------------------------------------------------------------------------------------------------ slider
struct SliderJ {
    typedef mi_slider_constraint Pod;
    struct Upd {
        V3 tangent, bitangent, rBxt, rBxb, rAuxt, rAuxb; float t00, t01, t10, t11; M3 invEffR; F2 tBias; V3 rBias;
        V3 axis; uint32_t solveLimit, solveMotor; float limitImpulse; V3 rAuxs, rBxs; float effAxial, limitSign, limitBias; V3 llA, llB;
        float maxMotorImpulse, motorImpulse, motorVelocity;
    };
    static constexpr int kAcc = 2;
    __device__ static void getAcc(const Upd& c, float* a) { a[0] = c.limitImpulse; a[1] = c.motorImpulse; }
    __device__ static void setAcc(Upd& c, const float* a) { c.limitImpulse = a[0]; c.motorImpulse = a[1]; }
    __device__ static void init(const Pod& in, const BodyState& A, const BodyState& B, float dt, Upd& o) {
        float invDt = 1.f / dt;
        V3 rA = rotate(A.rot, ld3(in.local_anchor_a) - A.cog);
        V3 rB = rotate(B.rot, ld3(in.local_anchor_b) - B.cog);
        V3 gA = A.pos + rA, gB = B.pos + rB;
        V3 axis = rotate(A.rot, ld3(in.local_axis_a));
        o.tangent = tangentOf(axis); o.bitangent = cross(axis, o.tangent);
        V3 u = gB - gA;
        V3 rAu = rA + u;
        o.rBxt = cross(rB, o.tangent); o.rBxb = cross(rB, o.bitangent);
        o.rAuxt = cross(rAu, o.tangent); o.rAuxb = cross(rAu, o.bitangent);
        V3 iArAuxt = mul(A.invI, o.rAuxt), iArAuxb = mul(A.invI, o.rAuxb), iBrBxt = mul(B.invI, o.rBxt), iBrBxb = mul(B.invI, o.rBxb);
        float ims = A.invMass + B.invMass;
        o.t00 = dot(o.rAuxt, iArAuxt) + dot(o.rBxt, iBrBxt) + ims;
        o.t01 = dot(o.rAuxt, iArAuxb) + dot(o.rBxt, iBrBxb);
        o.t10 = dot(o.rAuxb, iArAuxt) + dot(o.rBxb, iBrBxt);
        o.t11 = dot(o.rAuxb, iArAuxb) + dot(o.rBxb, iBrBxb) + ims;
        o.invEffR = add(A.invI, B.invI);
        o.tBias.x = 0.f; o.tBias.y = 0.f; o.rBias = V3();
        if (dt > kDtThreshold) {
            float a = dot(u, o.tangent), b = dot(u, o.bitangent);
            float k = kBetaSlider * invDt;
            o.tBias.x = a * k; o.tBias.y = b * k;
            Q4 err = B.rot * ld4(in.initial_inv_rotation_difference) * conj(A.rot);
            o.rBias = err.v() * (kBetaSlider * invDt * 2.f);
        }
        o.axis = axis;
        float dist = dot(u, axis);
        o.solveLimit = 0; o.limitImpulse = 0.f; o.rAuxs = V3(); o.rBxs = V3(); o.effAxial = o.limitSign = o.limitBias = 0.f; o.llA = V3(); o.llB = V3();
        if (in.neg_distance_limit <= 0.f || in.pos_distance_limit >= 0.f) {
            bool minV = (in.neg_distance_limit <= 0.f) && (dist < in.neg_distance_limit);
            bool maxV = (in.pos_distance_limit >= 0.f) && (dist > in.pos_distance_limit);
            if (minV || maxV) {
                o.solveLimit = 1;
                o.rAuxs = cross(rAu, axis); o.rBxs = cross(rB, axis);
                float invAx = ims + dot(o.rAuxs, mul(A.invI, o.rAuxs)) + dot(o.rBxs, mul(B.invI, o.rBxs));
                o.effAxial = inv0(invAx);
                o.limitSign = minV ? 1.f : -1.f;
                if (dt > kDtThreshold) {
                    float err = minV ? (dist - in.neg_distance_limit) : (in.pos_distance_limit - dist);
                    o.limitBias = err * (kBetaSliderLimit * invDt);
                }
                o.llA = mul(A.invI, o.rAuxs); o.llB = mul(B.invI, o.rBxs);
            }
        }
        o.solveMotor = 0; o.maxMotorImpulse = o.motorImpulse = o.motorVelocity = 0.f;
        if (in.max_motor_force > 0.f) {
            o.solveMotor = 1;
            o.maxMotorImpulse = in.max_motor_force * dt;
            o.motorVelocity = in.motor_velocity_or_target_distance;
            if (in.motor_type == MI_MOTOR_POSITION) {
                float minL = (in.neg_distance_limit <= 0.f) ? in.neg_distance_limit : -INFINITY;
                float maxL = (in.pos_distance_limit >= 0.f) ? in.pos_distance_limit : INFINITY;
                float target = clampr(in.motor_velocity_or_target_distance, minL, maxL);
                o.motorVelocity = (dt > kDtThreshold) ? ((target - dist) * invDt) : 0.f;
            }
        }
    }
    __device__ static void solve(Upd& c, BodyVel& A, BodyVel& B) {
        if (c.solveMotor) {
            float cd = dot(B.v, c.axis) - dot(A.v, c.axis) - c.motorVelocity;
            float mass = 1.f / (A.invMass + B.invMass);
            float l = -mass * cd;
            float old = c.motorImpulse;
            c.motorImpulse = clampr(c.motorImpulse + l, -c.maxMotorImpulse, c.maxMotorImpulse);
            l = c.motorImpulse - old;
            V3 P = l * c.axis;
            A.v = A.v - A.invMass * P;
            B.v = B.v + B.invMass * P;
        }
        if (c.solveLimit) {
            float cd = dot(B.v, c.axis) + dot(B.w, c.rBxs) - dot(A.v, c.axis) - dot(A.w, c.rAuxs);
            float l = -c.effAxial * (c.limitSign * cd + c.limitBias);
            float imp = fmaxr(c.limitImpulse + l, 0.f);
            l = imp - c.limitImpulse;
            c.limitImpulse = imp;
            l *= c.limitSign;
            V3 P = l * c.axis;
            A.v = A.v - A.invMass * P;
            A.w = A.w - c.llA * l;
            B.v = B.v + B.invMass * P;
            B.w = B.w + c.llB * l;
        }
        {
            V3 cd = B.w - A.w;
            V3 rl = solve3(c.invEffR, -(cd + c.rBias));
            A.w = A.w - mul(A.invI, rl);
            B.w = B.w + mul(B.invI, rl);
        }
        {
            float cx = dot(c.tangent, B.v) + dot(c.rBxt, B.w) - dot(c.tangent, A.v) - dot(c.rAuxt, A.w);
            float cy = dot(c.bitangent, B.v) + dot(c.rBxb, B.w) - dot(c.bitangent, A.v) - dot(c.rAuxb, A.w);
            float sx = cx + c.tBias.x, sy = cy + c.tBias.y;
            float lx, ly;
            solve2(c.t00, c.t01, c.t10, c.t11, -sx, -sy, lx, ly);
            V3 tb = c.tangent * lx + c.bitangent * ly;
            A.v = A.v - A.invMass * tb;
            A.w = A.w - mul(A.invI, c.rAuxt * lx + c.rAuxb * ly);
            B.v = B.v + B.invMass * tb;
            B.w = B.w + mul(B.invI, c.rBxt * lx + c.rBxb * ly);
        }
    }
};

template <class J>
__global__ __launch_bounds__(64) void k_joint_init(uint32_t n, uint32_t dummy, const typename J::Pod* __restrict__ pods, const uint2* __restrict__ bodies,
                                                   typename J::Upd* __restrict__ upd, float4* __restrict__ acc, BodyView bv, float dt) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint2 b = bodies[i];
    if (bv.active && !bv.active[b.x]) return;
    typename J::Pod in = pods[i];
    BodyState A = loadState(bv, b.x, dummy), B = loadState(bv, b.y, dummy);
    typename J::Upd o;
    J::init(in, A, B, dt, o);
    upd[i] = o;
    // accumulator granules of the fused solver (x, y = two accumulators, z = sweeps completed): start of the step
    if (J::kAcc > 0) { float a[4] = {0.f, 0.f, 0.f, 0.f}; J::getAcc(o, a); acc[2 * i] = make_float4(a[0], a[1], 0.f, 0.f); acc[2 * i + 1] = make_float4(a[2], a[3], 0.f, 0.f); }
}
// One colour of one joint type: lanes [s0, s1) of the colour-sorted order own disjoint dynamic bodies.
template <class J>
__global__ __launch_bounds__(64) void k_joint_solve(uint32_t s0, uint32_t s1, const uint32_t* __restrict__ order, const uint2* __restrict__ bodies,
                                                    typename J::Upd* __restrict__ upd, BodyView bv) {
    uint32_t s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= s1) return;
    uint32_t i = order[s];
    uint2 b = bodies[i];
    if (bv.active && !bv.active[b.x]) return;
    typename J::Upd c = upd[i];
    BodyVel A = loadVel(bv, b.x), B = loadVel(bv, b.y);
    J::solve(c, A, B);
    upd[i] = c;
    storeVel(bv, b.x, A); storeVel(bv, b.y, B);
}
template <class J>
__global__ void k_joint_solve_serial(uint32_t s0, uint32_t s1, const uint32_t* __restrict__ order, const uint2* __restrict__ bodies,
                                     typename J::Upd* __restrict__ upd, BodyView bv) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (uint32_t s = s0; s < s1; ++s) {
        uint32_t i = order[s];
        uint2 b = bodies[i];
        if (bv.active && !bv.active[b.x]) continue;
        typename J::Upd c = upd[i];
        BodyVel A = loadVel(bv, b.x), B = loadVel(bv, b.y);
        J::solve(c, A, B);
        upd[i] = c;
        storeVel(bv, b.x, A); storeVel(bv, b.y, B);
        __threadfence();
    }
}

// ---- articulated islands: ONE launch per sweep for all joint types ---------------------------------------------------
// Joints only couple the bodies of their own articulated island (a ragdoll, a vehicle): islands share no dynamic body, so
// their joint phases are independent and need no ordering against each other.  One wave per island runs the island's joints
// of a sweep in the canonical order — type-major (distance, ball, fixed, hinge, cone-twist, slider: constraints.cpp:3764-3769),
// colour-major inside a type — one lane per joint, active when its (type, colour) group is up, the island's body velocities
// staged in LDS between groups.  A sweep's joint phase is then one launch instead of one per (type, colour): 8-15 launches less per
// sweep on the ragdoll / vehicle configurations, where launch latency, not work, bounded the solver.  Islands with more than
// 64 bodies or more than 64 joints (or an overflow-coloured joint) stay on the per-colour kernels above.
struct IslandStep { uint32_t joint; uint16_t a, b, type, group; };   // joint index in its type's arrays; island-local body slots; (type, colour) group
struct IslandDesc { uint32_t bodyBegin, numBodies, stepBegin, numJoints; uint32_t typeGroups[7]; };   // groups of type t: [typeGroups[t], typeGroups[t + 1])
struct IslandUpd { DistanceJ::Upd* distance; BallJ::Upd* ball; FixedJ::Upd* fixed; HingeJ::Upd* hinge; ConeJ::Upd* cone; SliderJ::Upd* slider; };
constexpr uint32_t kIslandMaxBodies = 64, kIslandMaxJoints = 64;

struct IslandLds { float4 v[kIslandMaxBodies], w[kIslandMaxBodies], inertia[3 * kIslandMaxBodies]; float invMass[kIslandMaxBodies]; };
// One joint type of one island: the lanes owning a joint of this type keep its per-step data (Upd) in registers across the
// type's colour groups — one global round trip per type and sweep, not one per group.
template <class J>
__device__ __forceinline__ void islandType(uint32_t type, const IslandDesc& d, bool hasJoint, const IslandStep& st, typename J::Upd* __restrict__ upd, IslandLds& lds) {
    const uint32_t g0 = d.typeGroups[type], g1 = d.typeGroups[type + 1];
    if (g0 == g1) return;   // wave-uniform
    const bool mine = hasJoint && st.type == type;
    typename J::Upd c;
    if (mine) c = upd[st.joint];
    for (uint32_t g = g0; g < g1; ++g) {
        if (mine && st.group == g) {
            BodyVel A, B;
            { float4 a = lds.v[st.a], w = lds.w[st.a]; A.v = xyz(a); A.tagV = a.w; A.w = xyz(w); A.tagW = w.w; A.invMass = lds.invMass[st.a]; A.invI = ldM3(lds.inertia, st.a); }
            { float4 a = lds.v[st.b], w = lds.w[st.b]; B.v = xyz(a); B.tagV = a.w; B.w = xyz(w); B.tagW = w.w; B.invMass = lds.invMass[st.b]; B.invI = ldM3(lds.inertia, st.b); }
            J::solve(c, A, B);
            if (A.invMass != 0.f) { lds.v[st.a] = f4(A.v, A.tagV); lds.w[st.a] = f4(A.w, A.tagW); }   // kinematic bodies are never changed by impulses
            if (B.invMass != 0.f) { lds.v[st.b] = f4(B.v, B.tagV); lds.w[st.b] = f4(B.w, B.tagW); }
        }
        __syncthreads();
    }
    if (mine) upd[st.joint] = c;
}
__global__ __launch_bounds__(64) void k_joint_islands(const IslandDesc* __restrict__ islands, const IslandStep* __restrict__ steps,
                                                      const uint32_t* __restrict__ islandBodies, IslandUpd upd, BodyView bv) {
    __shared__ IslandLds lds;
    const IslandDesc d = islands[blockIdx.x];
    if (bv.active && !bv.active[islandBodies[d.bodyBegin]]) return;   // an island is simulated as a whole or not at all
    const uint32_t lane = threadIdx.x;
    uint32_t body = 0;
    if (lane < d.numBodies) {
        body = islandBodies[d.bodyBegin + lane];
        lds.v[lane] = bv.gVel[2 * body]; lds.w[lane] = bv.gVel[2 * body + 1]; lds.invMass[lane] = bv.gPos[body].w;
        lds.inertia[3 * lane] = bv.gInvI[3 * body]; lds.inertia[3 * lane + 1] = bv.gInvI[3 * body + 1]; lds.inertia[3 * lane + 2] = bv.gInvI[3 * body + 2];
    }
    const bool hasJoint = lane < d.numJoints;
    IslandStep st{};
    if (hasJoint) st = steps[d.stepBegin + lane];
    __syncthreads();
    islandType<DistanceJ>(0, d, hasJoint, st, upd.distance, lds);
    islandType<BallJ>(1, d, hasJoint, st, upd.ball, lds);
    islandType<FixedJ>(2, d, hasJoint, st, upd.fixed, lds);
    islandType<HingeJ>(3, d, hasJoint, st, upd.hinge, lds);
    islandType<ConeJ>(4, d, hasJoint, st, upd.cone, lds);
    islandType<SliderJ>(5, d, hasJoint, st, upd.slider, lds);
    if (lane < d.numBodies && lds.invMass[lane] != 0.f) { bv.gVel[2 * body] = lds.v[lane]; bv.gVel[2 * body + 1] = lds.w[lane]; }
}

// ---- contacts AND joints of all sweeps in one launch --------------------------------------------------------------------
// When every joint lives in an island, the joint phase joins the dataflow solver (k_contact_solve_flow): per sweep the grid
// holds the islands first, then the contact tiles.  A body in an island gets one more version per sweep: the island block of
// sweep `it` waits until each of its dynamic bodies carries tag it * (deg + 1) (deg = contact colours on the body: every
// contact update of the previous sweep has landed), solves the island's joints in canonical order and publishes tag + 1; the
// body's contact manifolds of that sweep then expect it * (deg + 1) + 1 + (colours below) — k_contact_init folds the extra
// version into their packed (base, deg).  The clamped joint accumulators (limit / motor impulses) travel between sweeps as
// tagged 16-byte granules like the contact impulses; everything else of a joint's per-step data is constant after
// k_joint_init.  No kernel boundary is left inside the solver: 2 x iterations launches less per step.
struct IslandAcc { float4* hinge; float4* cone; float4* slider; };
// the colour groups of one joint type; `c` (this lane's joint data, accumulators already merged) stays in registers
template <class J>
__device__ __forceinline__ void fusedGroups(uint32_t type, const IslandDesc& d, bool mine, const IslandStep& st, typename J::Upd& c, IslandLds& lds) {
    const uint32_t g0 = d.typeGroups[type], g1 = d.typeGroups[type + 1];
    for (uint32_t g = g0; g < g1; ++g) {   // wave-uniform bounds
        if (mine && st.group == g) {
            BodyVel A, B;
            { float4 a = lds.v[st.a], w = lds.w[st.a]; A.v = xyz(a); A.tagV = a.w; A.w = xyz(w); A.tagW = w.w; A.invMass = lds.invMass[st.a]; A.invI = ldM3(lds.inertia, st.a); }
            { float4 a = lds.v[st.b], w = lds.w[st.b]; B.v = xyz(a); B.tagV = a.w; B.w = xyz(w); B.tagW = w.w; B.invMass = lds.invMass[st.b]; B.invI = ldM3(lds.inertia, st.b); }
            J::solve(c, A, B);
            if (A.invMass != 0.f) { lds.v[st.a] = f4(A.v, A.tagV); lds.w[st.a] = f4(A.w, A.tagW); }
            if (B.invMass != 0.f) { lds.v[st.b] = f4(B.v, B.tagV); lds.w[st.b] = f4(B.w, B.tagW); }
        }
        __syncthreads();
    }
}
template <class J>
__device__ __forceinline__ void fusedPublish(bool mine, uint32_t it, const IslandStep& st, const typename J::Upd& c, float4* acc) {
    if (!mine) return;
    float a[4] = {0.f, 0.f, 0.f, 0.f}; J::getAcc(c, a);
    const float t = __uint_as_float(it + 1u);
    f32x4 q0 = {a[0], a[1], t, 0.f}, q1 = {a[2], a[3], t, 0.f};
    storeGranuleSc1(acc + 2 * (size_t)st.joint, q0);
    if (J::kAcc > 2) storeGranuleSc1(acc + 2 * (size_t)st.joint + 1, q1);
}
// A lane owns ONE joint of ONE type, but which type is only known at run time: six typed copies of the per-step joint data side by side (~250 registers) spilled
// to scratch inside the sweep loop (620 bytes per lane; every joint group then waited for scratch).  The data therefore lives in one untyped block sized for the
// largest type and is viewed as its type only inside that type's groups.
constexpr size_t kUpdMaxBytes = std::max({sizeof(DistanceJ::Upd), sizeof(BallJ::Upd), sizeof(FixedJ::Upd), sizeof(HingeJ::Upd), sizeof(ConeJ::Upd), sizeof(SliderJ::Upd)});
struct UpdRaw { uint32_t w[(kUpdMaxBytes + 3) / 4]; };
template <class J>
__device__ __forceinline__ void rawGroups(uint32_t type, const IslandDesc& d, bool mine, const IslandStep& st, UpdRaw& raw, IslandLds& lds) {
    if (d.typeGroups[type] == d.typeGroups[type + 1]) return;   // wave-uniform
    typename J::Upd c;
    if (mine) __builtin_memcpy(&c, raw.w, sizeof(c));
    fusedGroups<J>(type, d, mine, st, c, lds);
    if (mine) __builtin_memcpy(raw.w, &c, sizeof(c));
}
template <class J>
__device__ __forceinline__ void rawLoad(UpdRaw& raw, const typename J::Upd* __restrict__ upd, uint32_t joint, const float* a /* accumulators, or null */) {
    typename J::Upd c = upd[joint];
    if (a) J::setAcc(c, a);
    __builtin_memcpy(raw.w, &c, sizeof(c));
}
template <class J>
__device__ __forceinline__ void rawSetAcc(UpdRaw& raw, const float* a) { typename J::Upd c; __builtin_memcpy(&c, raw.w, sizeof(c)); J::setAcc(c, a); __builtin_memcpy(raw.w, &c, sizeof(c)); }
template <class J>
__device__ __forceinline__ void rawPublish(bool mine, uint32_t it, const IslandStep& st, const UpdRaw& raw, float4* acc) {
    if (!mine) return;
    typename J::Upd c; __builtin_memcpy(&c, raw.w, sizeof(c));
    fusedPublish<J>(true, it, st, c, acc);
}
__device__ __forceinline__ void fusedIsland(uint32_t it, const IslandDesc& d, const IslandStep* __restrict__ steps, const uint32_t* __restrict__ islandBodies, const IslandUpd& upd,
                                            const IslandAcc& acc, const BodyView& bv, const unsigned long long* __restrict__ bodyUsed, IslandLds& lds, StepScalars* sc) {
    const uint32_t lane = threadIdx.x;
    // everything this block needs is requested up front, so the waits below overlap in ONE memory round trip: the island's
    // bodies (tagged granules), this lane's joint data (constant after k_joint_init) and its accumulator granules
    const bool hasJoint = lane < d.numJoints;
    IslandStep st{};
    if (hasJoint) st = steps[d.stepBegin + lane];
    uint32_t body = 0, expect = 0; bool dynamic = false;
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, q0 = g0, q1 = g0;
    const bool isBody = lane < d.numBodies;
    if (isBody) {
        body = islandBodies[d.bodyBegin + lane];
        issueGranuleSc1(bv.gVel + 2 * (size_t)body, g0); issueGranuleSc1(bv.gVel + 2 * (size_t)body + 1, g1);
    }
    const uint32_t type = hasJoint ? st.type : 0xFFFFu;
    float4* accPtr = type == 3u ? acc.hinge : type == 4u ? acc.cone : type == 5u ? acc.slider : nullptr;
    const bool twoGranules = type == 4u;
    if (accPtr) { accPtr += 2 * (size_t)st.joint; issueGranuleSc1(accPtr, q0); if (twoGranules) issueGranuleSc1(accPtr + 1, q1); }
    UpdRaw raw;   // (one untyped block instead of six typed copies: see UpdRaw)
    switch (type) {
        case 0: rawLoad<DistanceJ>(raw, upd.distance, st.joint, nullptr); break;
        case 1: rawLoad<BallJ>(raw, upd.ball, st.joint, nullptr); break;
        case 2: rawLoad<FixedJ>(raw, upd.fixed, st.joint, nullptr); break;
        case 3: rawLoad<HingeJ>(raw, upd.hinge, st.joint, nullptr); break;
        case 4: rawLoad<ConeJ>(raw, upd.cone, st.joint, nullptr); break;
        case 5: rawLoad<SliderJ>(raw, upd.slider, st.joint, nullptr); break;
        default: break;
    }
    if (isBody) {
        const float im = bv.gPos[body].w;
        lds.invMass[lane] = im; dynamic = im != 0.f;
        lds.inertia[3 * lane] = bv.gInvI[3 * body]; lds.inertia[3 * lane + 1] = bv.gInvI[3 * body + 1]; lds.inertia[3 * lane + 2] = bv.gInvI[3 * body + 2];
        expect = it * ((uint32_t)__popcll(bodyUsed[body]) + 1u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed(g0); landed(g1); landed(q0); landed(q1);
    {   // dynamic bodies must carry the tag that says "every update of the previous sweep has landed"; accumulators the sweep count
        bool okBody = !isBody || !dynamic || (__float_as_uint(g0.w) == expect && __float_as_uint(g1.w) == expect);
        bool okAcc = !accPtr || (__float_as_uint(q0.z) == it && (!twoGranules || __float_as_uint(q1.z) == it));
        uint32_t budget = kSpinBudget;
        while (__ballot(!(okBody && okAcc)) != 0ull) {
            if (!okBody) {
                loadGranuleSc1(bv.gVel + 2 * (size_t)body, g0); loadGranuleSc1(bv.gVel + 2 * (size_t)body + 1, g1);
                okBody = __float_as_uint(g0.w) == expect && __float_as_uint(g1.w) == expect;
            }
            if (!okAcc) {
                loadGranuleSc1(accPtr, q0); if (twoGranules) loadGranuleSc1(accPtr + 1, q1);
                okAcc = __float_as_uint(q0.z) == it && (!twoGranules || __float_as_uint(q1.z) == it);
            }
            if (--budget == 0u) { sc->solveError = 1u; break; }
        }
        if (isBody) { lds.v[lane] = make_float4(g0.x, g0.y, g0.z, g0.w); lds.w[lane] = make_float4(g1.x, g1.y, g1.z, g1.w); }
    }
    { const float a[4] = {q0.x, q0.y, q1.x, q1.y}; if (type == 3u) rawSetAcc<HingeJ>(raw, a); else if (type == 4u) rawSetAcc<ConeJ>(raw, a); else if (type == 5u) rawSetAcc<SliderJ>(raw, a); }
    __syncthreads();
    rawGroups<DistanceJ>(0, d, type == 0u, st, raw, lds);
    rawGroups<BallJ>(1, d, type == 1u, st, raw, lds);
    rawGroups<FixedJ>(2, d, type == 2u, st, raw, lds);
    rawGroups<HingeJ>(3, d, type == 3u, st, raw, lds);
    rawPublish<HingeJ>(type == 3u, it, st, raw, acc.hinge);
    rawGroups<ConeJ>(4, d, type == 4u, st, raw, lds);
    rawPublish<ConeJ>(type == 4u, it, st, raw, acc.cone);
    rawGroups<SliderJ>(5, d, type == 5u, st, raw, lds);
    rawPublish<SliderJ>(type == 5u, it, st, raw, acc.slider);
    if (isBody && dynamic) {
        const float t = __uint_as_float(expect + 1u);
        const float4 v = lds.v[lane], w = lds.w[lane];
        f32x4 o0 = {v.x, v.y, v.z, t}, o1 = {w.x, w.y, w.z, t};
        storeGranuleSc1(bv.gVel + 2 * (size_t)body, o0); storeGranuleSc1(bv.gVel + 2 * (size_t)body + 1, o1);
    }
}
// ---- PRIVATE islands: joints AND contacts of all sweeps inside the island's one workgroup ------------------------------------
// An island is private in a step when every manifold that touches one of its dynamic bodies has no dynamic body outside the island (a ragdoll on the
// ground, limbs touching limbs; a vehicle on static hull tiles — every island of cfg4 / cfg5), there are at most kIslandMaxContacts of them and none
// is overflow-coloured.  Nothing outside then reads or writes the island's velocities during the solve, so there is nothing to hand over: workgroup
// (island, sweep 0) keeps the bodies in LDS, its joints' data, each lane ONE manifold's rows and accumulated impulses in registers, and runs ALL sweeps —
// per sweep the joint groups in canonical order, then the island's manifolds colour after colour (one lane per manifold; manifolds of a colour share
// no dynamic body) — and writes the velocities back once.  The arithmetic is the tile solver's (solveOnePk) in the canonical order restricted to the
// island, which is all that order means for bodies nobody else touches: bit-identical results, and the ~19 us island -> contacts -> island hand-over per
// sweep (the whole cost of the solve on these scenes) is gone.  The manifolds stay in their global tiles for k_contact_init (rows are computed there) but
// are marked invalid for the tile solver; k_contact_init appends (slot, first contact-tile, colour, contacts) to the island's list.
// after the colouring, before k_contact_init: which islands are private this step
__global__ __launch_bounds__(256) void k_island_classify(const StepScalars* __restrict__ sc, const uint4* __restrict__ colWork, const uint32_t* __restrict__ color, IslandPrivate ip) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= sc->numManifolds) return;
    const uint4 w = colWork[m];
    const bool dynA = (w.x >> 31) != 0u, dynB = (w.y >> 31) != 0u;
    const uint32_t iA = dynA ? ip.bodyIsland[w.x & 0x3FFFFFFFu] : 0xFFFFFFFFu, iB = dynB ? ip.bodyIsland[w.y & 0x7FFFFFFFu] : 0xFFFFFFFFu;
    const bool freeA = dynA && iA == 0xFFFFFFFFu, freeB = dynB && iB == 0xFFFFFFFFu;            // a dynamic body that belongs to no island
    const bool coupled = (dynA && dynB && iA != iB) || freeA || freeB || color[m] >= kOverflowColor;
    if (iA != 0xFFFFFFFFu) { atomicAdd(&ip.count[iA], 1u); if (coupled) ip.shared[iA] = 1u; }
    if (iB != 0xFFFFFFFFu && iB != iA) { atomicAdd(&ip.count[iB], 1u); if (coupled) ip.shared[iB] = 1u; }
}
struct PrivateLds { uint32_t bodyId[kIslandMaxBodies]; unsigned long long colours; };
__device__ __forceinline__ void privateIsland(uint32_t sweeps, uint32_t island, const IslandDesc& d, const IslandStep* __restrict__ steps, const uint32_t* __restrict__ islandBodies,
                                              const IslandUpd& upd, const IslandAcc& acc, const BodyView& bv, const IslandPrivate& ip,
                                              const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal, const float2* __restrict__ slotMass, const float4* __restrict__ rows,
                                              IslandLds& lds, PrivateLds& pl) {
    const uint32_t lane = threadIdx.x;
    const bool hasJoint = lane < d.numJoints, isBody = lane < d.numBodies;
    IslandStep st{};
    if (hasJoint) st = steps[d.stepBegin + lane];
    uint32_t body = 0; bool dynamic = false;
    if (lane == 0) pl.colours = 0ull;
    if (isBody) {
        body = islandBodies[d.bodyBegin + lane];
        pl.bodyId[lane] = body;
        lds.v[lane] = bv.gVel[2 * (size_t)body]; lds.w[lane] = bv.gVel[2 * (size_t)body + 1];
        const float im = bv.gPos[body].w;
        lds.invMass[lane] = im; dynamic = im != 0.f;
        lds.inertia[3 * lane] = bv.gInvI[3 * body]; lds.inertia[3 * lane + 1] = bv.gInvI[3 * body + 1]; lds.inertia[3 * lane + 2] = bv.gInvI[3 * body + 2];
    }
    const uint32_t type = hasJoint ? st.type : 0xFFFFu;
    UpdRaw raw;
    {   // the clamped accumulators start from what k_joint_init left in the granules (sweep tag 0) and then stay in registers
        const float4* accPtr = type == 3u ? acc.hinge : type == 4u ? acc.cone : type == 5u ? acc.slider : nullptr;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (accPtr) {
            const float4 q0 = accPtr[2 * (size_t)st.joint], q1 = type == 4u ? accPtr[2 * (size_t)st.joint + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
            a[0] = q0.x; a[1] = q0.y; a[2] = q1.x; a[3] = q1.y;
        }
        switch (type) {
            case 0: rawLoad<DistanceJ>(raw, upd.distance, st.joint, nullptr); break;
            case 1: rawLoad<BallJ>(raw, upd.ball, st.joint, nullptr); break;
            case 2: rawLoad<FixedJ>(raw, upd.fixed, st.joint, nullptr); break;
            case 3: rawLoad<HingeJ>(raw, upd.hinge, st.joint, a); break;
            case 4: rawLoad<ConeJ>(raw, upd.cone, st.joint, a); break;
            case 5: rawLoad<SliderJ>(raw, upd.slider, st.joint, a); break;
            default: break;
        }
    }
    // this lane's manifold
    const uint32_t nPriv = min(ip.fill[island], kIslandMaxContacts);
    const bool hasContact = lane < nPriv;
    uint32_t colour = 0xFFFFFFFFu, cnt = 0; bool perContactNormal = false;   // (a terrain manifold: heightmap.hpp, HmOut::put)
    uint4 meta = make_uint4(0u, 0u, 0u, 0u); float4 nf = make_float4(0.f, 0.f, 0.f, 0.f); float2 mass = make_float2(0.f, 0.f);
    ContactRows c[4];
    float2 im[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};   // no warm start (constraints.cpp:3312-3313)
    if (hasContact) {
        const uint4 e = ip.entries[(size_t)island * kIslandMaxContacts + lane];
        const uint32_t slot = e.x, ln = slot & 63u;
        colour = e.z & 0xFFu; cnt = (e.z >> 8) & 0xFFu; perContactNormal = ((e.z >> 16) & 1u) != 0u;
        meta = slotMeta[slot]; nf = slotNormal[slot]; mass = slotMass[slot];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) if (k < cnt) {
            const float4* __restrict__ row = rows + ((size_t)e.y + k) * (kRows * 64u) + ln;
#pragma unroll
            for (uint32_t r = 0; r < kRows; ++r) c[k].r[r] = row[r * 64u];
        }
    }
    __syncthreads();
    if (hasContact) atomicOr(&pl.colours, 1ull << colour);
    // island-local slots of the manifold's bodies; a body outside the island (static, kinematic: never changed by impulses) keeps its velocity in registers
    int la = -1, lb = -1;
    float4 extVA = make_float4(0.f, 0.f, 0.f, 0.f), extWA = extVA, extVB = extVA, extWB = extVA;
    if (hasContact) {
        for (uint32_t k = 0; k < d.numBodies; ++k) { const uint32_t id = pl.bodyId[k]; if (id == meta.x) la = (int)k; if (id == meta.y) lb = (int)k; }
        if (la < 0) { extVA = bv.gVel[2 * (size_t)meta.x]; extWA = bv.gVel[2 * (size_t)meta.x + 1]; }
        if (lb < 0) { extVB = bv.gVel[2 * (size_t)meta.y]; extWB = bv.gVel[2 * (size_t)meta.y + 1]; }
    }
    __syncthreads();
    const unsigned long long colours = pl.colours;
    const f32x2 sMass = pk2(-mass.x, mass.y);
    for (uint32_t it = 0; it < sweeps; ++it) {
        rawGroups<DistanceJ>(0, d, type == 0u, st, raw, lds);
        rawGroups<BallJ>(1, d, type == 1u, st, raw, lds);
        rawGroups<FixedJ>(2, d, type == 2u, st, raw, lds);
        rawGroups<HingeJ>(3, d, type == 3u, st, raw, lds);
        rawGroups<ConeJ>(4, d, type == 4u, st, raw, lds);
        rawGroups<SliderJ>(5, d, type == 5u, st, raw, lds);
        for (unsigned long long rest = colours; rest; rest &= rest - 1ull) {   // wave-uniform
            const uint32_t cur = (uint32_t)__ffsll((long long)rest) - 1u;
            if (hasContact && colour == cur) {
                const float4 a0 = la >= 0 ? lds.v[la] : extVA, a1 = la >= 0 ? lds.w[la] : extWA, b0 = lb >= 0 ? lds.v[lb] : extVB, b1 = lb >= 0 ? lds.w[lb] : extWB;
                P3 pv, pw;
                pv.x = pk2(a0.x, b0.x); pv.y = pk2(a0.y, b0.y); pv.z = pk2(a0.z, b0.z);
                pw.x = pk2(a1.x, b1.x); pw.y = pk2(a1.y, b1.y); pw.z = pk2(a1.z, b1.z);
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) if (k < cnt) solveOnePk(c[k], contactNormal(c[k], nf, perContactNormal), im[k], sMass, pv, pw);
                if (la >= 0 && mass.x != 0.f) { lds.v[la] = make_float4(pv.x.x, pv.y.x, pv.z.x, a0.w); lds.w[la] = make_float4(pw.x.x, pw.y.x, pw.z.x, a1.w); }
                if (lb >= 0 && mass.y != 0.f) { lds.v[lb] = make_float4(pv.x.y, pv.y.y, pv.z.y, b0.w); lds.w[lb] = make_float4(pw.x.y, pw.y.y, pw.z.y, b1.w); }
            }
            __syncthreads();
        }
    }
    if (isBody && dynamic) { bv.gVel[2 * (size_t)body] = lds.v[lane]; bv.gVel[2 * (size_t)body + 1] = lds.w[lane]; }
}
// Block b of sweep s = itBase + b / (numIslands + numTiles): island b' < numIslands, else contact tile b' - numIslands.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, MI_FLOW_WAVES))) void k_solve_flow_islands(
    uint32_t itBase, uint32_t sweeps, uint32_t numIslands, const IslandDesc* __restrict__ islands, const IslandStep* __restrict__ steps, const uint32_t* __restrict__ islandBodies,
    IslandUpd upd, IslandAcc acc, BodyView bv, const unsigned long long* __restrict__ bodyUsed,
    const uint2* __restrict__ tileDesc, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal, const float2* __restrict__ slotMass,
    const float4* __restrict__ rows, float4* imp, StepScalars* sc, IslandPrivate ip /* bodyIsland == null: every island goes through the dataflow */, uint32_t totalSweeps) {
    __shared__ IslandLds lds;
    __shared__ PrivateLds pl;
    const uint32_t numTiles = sc->totalTiles, per = numIslands + numTiles;
    if (blockIdx.x >= per * sweeps) return;
    if (stepIsVoid(sc)) return;
    const uint32_t it = itBase + blockIdx.x / per, idx = blockIdx.x % per;
    if (idx < numIslands) {
        if (bv.active && !bv.active[islandBodies[islands[idx].bodyBegin]]) return;
        if (ip.bodyIsland && islandIsPrivate(ip, idx)) {   // all sweeps in the block of sweep 0; the island's blocks of the later sweeps have nothing to do
            if (it == 0u) privateIsland(totalSweeps, idx, islands[idx], steps, islandBodies, upd, acc, bv, ip, slotMeta, slotNormal, slotMass, rows, lds, pl);
            return;
        }
        fusedIsland(it, islands[idx], steps, islandBodies, upd, acc, bv, bodyUsed, lds, sc); return;
    }
    const uint32_t tile = idx - numIslands, lane = threadIdx.x;
    const uint2 d = tileDesc[tile];
    switch (d.y) {
        case 1: flowTile<1>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, bv.gVel, sc); break;
        case 2: flowTile<2>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, bv.gVel, sc); break;
        case 3: flowTile<3>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, bv.gVel, sc); break;
        default: flowTile<4>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, bv.gVel, sc); break;
    }
}

}  // namespace mi

// ------------------------------------------------------------------------------------------------ host side
struct mi_world;

template <class J>
struct JointType {
    std::vector<typename J::Pod> pods;
    std::vector<uint2> bodies;          // rigid body indices (A, B)
    // dense arrays in EnTT pool order (append on create, swap-and-pop on destroy); public handles stay valid through denseOf
    std::vector<uint32_t> handleAt; std::vector<int32_t> denseOf; std::vector<uint2> ents; std::vector<uint64_t> seq;
    int dense(uint32_t handle) const { return handle < denseOf.size() ? denseOf[handle] : -1; }
    bool destroy(uint32_t handle) {
        int d = dense(handle);
        if (d < 0) return false;
        size_t last = pods.size() - 1;
        pods[d] = pods[last]; bodies[d] = bodies[last]; handleAt[d] = handleAt[last]; ents[d] = ents[last]; seq[d] = seq[last];
        denseOf[handleAt[d]] = d;
        pods.pop_back(); bodies.pop_back(); handleAt.pop_back(); ents.pop_back(); seq.pop_back();
        denseOf[handle] = -1;
        return true;
    }
    void clearAll() { pods.clear(); bodies.clear(); handleAt.clear(); ents.clear(); seq.clear(); std::fill(denseOf.begin(), denseOf.end(), -1); }
    std::vector<uint32_t> order;        // colour-major, index-minor; only the joints the per-colour kernels solve (see JointSet::buildIslands)
    std::vector<uint32_t> colorOffsets; // [0..65] boundaries into order (64 = overflow colour)
    std::vector<uint32_t> colorOf;      // colour of every joint of this type
    std::vector<uint8_t> inIsland;      // joint is solved by k_joint_islands
    typename J::Pod* dPods = nullptr; uint2* dBodies = nullptr; uint32_t* dOrder = nullptr; typename J::Upd* dUpd = nullptr; float4* dAcc = nullptr;
    size_t dCap = 0;
    ~JointType() { release(); }
    void release() {
        if (dPods) (void)hipFree(dPods); if (dBodies) (void)hipFree(dBodies); if (dOrder) (void)hipFree(dOrder); if (dUpd) (void)hipFree(dUpd); if (dAcc) (void)hipFree(dAcc);
        if (dIdentity) (void)hipFree(dIdentity); dIdentity = nullptr; identityCap = 0;
        dPods = nullptr; dBodies = nullptr; dOrder = nullptr; dUpd = nullptr; dAcc = nullptr; dCap = 0;
    }
    // Greedy colouring in descending hash32 priority with 64-bit per-body colour masks (same rule as the contact schedule).
    void computeOrder(const std::vector<float>& invMass) {
        uint32_t n = (uint32_t)bodies.size();
        order.resize(n);
        std::vector<uint32_t> prio(n), color(n, 64);
        for (uint32_t i = 0; i < n; ++i) { order[i] = i; prio[i] = i; }
        std::sort(prio.begin(), prio.end(), [](uint32_t a, uint32_t b) { return mi::hash32(a) > mi::hash32(b); });
        std::vector<unsigned long long> used(invMass.size(), 0ull);
        for (uint32_t j : prio) {
            uint2 bp = bodies[j];
            bool dynA = invMass[bp.x] != 0.f, dynB = invMass[bp.y] != 0.f;
            unsigned long long mask = (dynA ? used[bp.x] : 0ull) | (dynB ? used[bp.y] : 0ull);
            if (~mask == 0ull) continue;
            uint32_t c = (uint32_t)__builtin_ctzll(~mask);
            color[j] = c;
            if (dynA) used[bp.x] |= 1ull << c;
            if (dynB) used[bp.y] |= 1ull << c;
        }
        colorOf = color;
        inIsland.assign(n, 0);
    }
    // after JointSet::buildIslands has claimed its joints: the colour-sorted order of the rest
    void finishOrder() {
        order.clear();
        for (uint32_t i = 0; i < (uint32_t)bodies.size(); ++i) if (!inIsland[i]) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return colorOf[a] < colorOf[b]; });
        colorOffsets.assign(66, 0);
        for (uint32_t i : order) colorOffsets[colorOf[i] + 1]++;
        for (int c = 0; c < 65; ++c) colorOffsets[c + 1] += colorOffsets[c];
    }
    hipError_t upload(hipStream_t st) {
        size_t n = pods.size();
        if (!n) return hipSuccess;
        if (n > dCap) {
            release();
            hipError_t e;
            if ((e = hipMalloc((void**)&dPods, n * sizeof(typename J::Pod))) != hipSuccess) return e;
            if ((e = hipMalloc((void**)&dBodies, n * sizeof(uint2))) != hipSuccess) return e;
            if ((e = hipMalloc((void**)&dOrder, n * sizeof(uint32_t))) != hipSuccess) return e;
            if ((e = hipMalloc((void**)&dUpd, n * sizeof(typename J::Upd))) != hipSuccess) return e;
            // exchanged between workgroups as tagged sc1 granules: uncached memory serves those fastest (see world.hip, gVel)
            if ((e = hipExtMallocWithFlags((void**)&dAcc, 2 * n * sizeof(float4), hipDeviceMallocUncached)) != hipSuccess) return e;
            dCap = n;
        }
        hipError_t e;
        if ((e = hipMemcpyAsync(dPods, pods.data(), n * sizeof(typename J::Pod), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(dBodies, bodies.data(), n * sizeof(uint2), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        if (order.empty()) return hipSuccess;
        return hipMemcpyAsync(dOrder, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    }
    // mi_debug_set_solve_order: the reference's order — every joint of the type, one after the other, in pool order
    uint32_t* dIdentity = nullptr; size_t identityCap = 0;
    hipError_t launchSolveReference(mi::Launcher& L, const mi::BodyView& bv, hipStream_t st) {
        const uint32_t n = (uint32_t)pods.size();
        if (!n) return hipSuccess;
        if (n > identityCap) {
            if (dIdentity) (void)hipFree(dIdentity);
            dIdentity = nullptr; identityCap = 0;
            hipError_t e = hipMalloc((void**)&dIdentity, n * sizeof(uint32_t)); if (e != hipSuccess) return e;
            std::vector<uint32_t> id(n); for (uint32_t i = 0; i < n; ++i) id[i] = i;
            e = hipMemcpy(dIdentity, id.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice); if (e != hipSuccess) return e;
            identityCap = n;
        }
        L.launch(mi::k_joint_solve_serial<J>, dim3(1), dim3(64), 0, st, 0u, n, dIdentity, dBodies, dUpd, bv);
        return hipSuccess;
    }
    bool podsDirty = false;   // mi_constraint_update since the last upload: only the POD array changed (motors, limits), not the topology
    hipError_t uploadPods(hipStream_t st) {
        podsDirty = false;
        if (pods.empty() || !dPods) return hipSuccess;
        return hipMemcpyAsync(dPods, pods.data(), pods.size() * sizeof(typename J::Pod), hipMemcpyHostToDevice, st);
    }
    void launchInit(mi::Launcher& L, uint32_t dummy, const mi::BodyView& bv, float dt, hipStream_t st) {
        uint32_t n = (uint32_t)pods.size();
        if (n) L.launch(mi::k_joint_init<J>, dim3((n + 63) / 64), dim3(64), 0, st, n, dummy, dPods, dBodies, dUpd, dAcc, bv, dt);
    }
    void launchSolve(mi::Launcher& L, const mi::BodyView& bv, hipStream_t st) {
        if (order.empty()) return;
        for (int c = 0; c < 64; ++c) {
            uint32_t s0 = colorOffsets[c], s1 = colorOffsets[c + 1];
            if (s1 > s0) L.launch(mi::k_joint_solve<J>, dim3((s1 - s0 + 63) / 64), dim3(64), 0, st, s0, s1, dOrder, dBodies, dUpd, bv);
        }
        uint32_t o0 = colorOffsets[64], o1 = colorOffsets[65];
        if (o1 > o0) L.launch(mi::k_joint_solve_serial<J>, dim3(1), dim3(64), 0, st, o0, o1, dOrder, dBodies, dUpd, bv);
    }
};

struct JointSet {
    JointType<mi::DistanceJ> distance; JointType<mi::BallJ> ball; JointType<mi::FixedJ> fixed;
    JointType<mi::HingeJ> hinge; JointType<mi::ConeJ> cone; JointType<mi::SliderJ> slider;

    size_t count() const { return distance.pods.size() + ball.pods.size() + fixed.pods.size() + hinge.pods.size() + cone.pods.size() + slider.pods.size(); }
    int add(mi_world& w, uint32_t type, uint32_t ea, uint32_t eb, const void* pod, uint32_t bytes, uint32_t* out);
    int update(uint32_t type, uint32_t id, const void* pod, uint32_t bytes);
    uint64_t nextSeq = 0;
    int destroy(uint32_t type, uint32_t id);
    void destroyAll();
    void destroyOfEntity(uint32_t entity);
    int get(uint32_t type, uint32_t id, void* pod, uint32_t bytes);
    int addFromGlobal(mi_world& w, uint32_t type, uint32_t ea, uint32_t eb, const float* anchor, const float* axis, float l0, float l1, uint32_t* out);
    int upload(mi_world& w, hipStream_t st);
    // articulated islands (k_joint_islands)
    uint32_t numIslands = 0;
    mi::IslandDesc* dIslands = nullptr; mi::IslandStep* dSteps = nullptr; uint32_t* dIslandBodies = nullptr;
    uint8_t* dBodyJ = nullptr;          // [bodies + 1]: 1 = dynamic body of an island (one extra version per sweep in the fused solver)
    uint32_t* dBodyIsland = nullptr;    // [bodies + 1]: island of a dynamic island body, else 0xFFFFFFFF (private islands: IslandPrivate)
    uint32_t* dIslState = nullptr;      // [3][islands]: shared / count / fill of this step
    uint4* dIslEntries = nullptr;       // [islands][kIslandMaxContacts]
    mi::IslandPrivate islandPrivate() const { return mi::IslandPrivate{dBodyIsland, dIslState, dIslState ? dIslState + numIslands : nullptr, dIslState ? dIslState + 2 * (size_t)numIslands : nullptr, dIslEntries}; }
    bool allInIslands() const { return numIslands && distance.order.empty() && ball.order.empty() && fixed.order.empty() && hinge.order.empty() && cone.order.empty() && slider.order.empty(); }
    void buildIslands(const std::vector<float>& invMass, std::vector<mi::IslandDesc>& islands, std::vector<mi::IslandStep>& steps, std::vector<uint32_t>& islandBodies);
    ~JointSet() { releaseIslands(); }
    void releaseIslands() {
        if (dIslands) (void)hipFree(dIslands); if (dSteps) (void)hipFree(dSteps); if (dIslandBodies) (void)hipFree(dIslandBodies); if (dBodyJ) (void)hipFree(dBodyJ);
        if (dBodyIsland) (void)hipFree(dBodyIsland); if (dIslState) (void)hipFree(dIslState); if (dIslEntries) (void)hipFree(dIslEntries);
        dIslands = nullptr; dSteps = nullptr; dIslandBodies = nullptr; dBodyJ = nullptr; dBodyIsland = nullptr; dIslState = nullptr; dIslEntries = nullptr; numIslands = 0;
    }
    bool podsDirty() const { return distance.podsDirty || ball.podsDirty || fixed.podsDirty || hinge.podsDirty || cone.podsDirty || slider.podsDirty; }
    int uploadPods(hipStream_t st);
    int initialize(mi_world& w, float dt, hipStream_t st);
    void solveIteration(mi_world& w, hipStream_t st);
    int solveIterationReference(mi_world& w, hipStream_t st);   // mi_debug_set_solve_order: sequentially, type by type, pool order
};
