// kernels.hpp — HIP kernels of the rigid-body step for gfx950 (wave64).
//
// Stage names follow the reference's profiler blocks.  HBM layout is SoA with
// float4 rows: a wave touching 64 consecutive bodies / contacts issues 1 KiB coalesced transactions.
// No MFMA anywhere: there is no dense contraction on this path; the roofline is HBM bandwidth.
#pragma once
#include <type_traits>
#include "dmath.hpp"
#include "narrow.hpp"

namespace mi {

constexpr uint32_t kNoBody = 0xFFFFFFFFu;
constexpr uint32_t kMaxCells = 1u << 22;
constexpr uint32_t kOverflowColor = 64;
constexpr uint32_t kUncolored = 0xFFFFFFFFu;
constexpr uint32_t kSeamColors = 24;                                  // exact seam (include/mi_shard.h MI_SEAM_COLORS): the seam manifolds' colours
constexpr unsigned long long kSeamRange = (1ull << kSeamColors) - 1ull;
constexpr uint32_t kNumBuckets = 21;                          // 6 x 6 upper-triangular collider type pairs
constexpr uint32_t kColorBins = (kOverflowColor + 1) * 4;     // (colour, contacts per manifold) bins of the solver schedule
constexpr uint32_t kMaxColorRounds = 4094;                    // 12-bit round tag in the colouring keys
constexpr uint32_t kIndexBits = 26;                           // colliders per world < 2^26 (52-bit unique pair priorities)

enum : uint32_t { OBJ_RIGID_BODY = 0, OBJ_STATIC = 1, OBJ_FORCE_FIELD = 2, OBJ_TRIGGER = 3 };
constexpr float kDeadBox = 3.0e38f;                           // sharded world: min = +kDeadBox, max = -kDeadBox marks a collider that is not simulated here

struct GridParams {   // written by k_bp_grid_setup, read by the broad-phase kernels
    float origin[3];
    float invCell;
    uint32_t dims[3];
    uint32_t numCells;
    uint32_t numLarge;
    float cell;
    float largeThreshold;
    uint32_t pad;
};

struct StepScalars {  // device-resident per-step scalars
    double extentSum;
    int boundsMin[3];     // ordered-int encoded floats
    int boundsMax[3];
    uint32_t numLarge;
    uint32_t numPairs;        // broad-phase overlaps that passed pruning (collision pairs)
    uint32_t numOverlaps;     // all AABB overlaps (CPU_PROFILE_STAT "Num broadphase overlaps")
    uint32_t numManifolds;
    uint32_t numContacts;
    uint32_t solveError;      // set by the dataflow solver if a dependency wait ran out of its spin budget
    uint32_t axisCur;
    uint32_t axisNext;
    uint32_t bucketHist[24];       // collision pairs per narrow-phase bucket (type pair)
    uint32_t bucketCursor[24];     // running output cursors of the bucket partition
    uint32_t binStart[kColorBins + 4];   // first schedule slot of every (colour, contact count) bin; [kColorBins] = manifolds
    float largeThreshold;
    uint32_t bucketOffset[24];     // first pair of every bucket in the partitioned pair list
    uint32_t gjkLo, gjkHi;         // span of the partitioned pair list holding the GJK/EPA buckets
    uint32_t partitioned;          // 1: the narrow phase reads the partitioned copy of the pair list
    uint32_t totalTiles, totalCt;  // schedule: tiles and contact-tiles (k_build_tiles)
    uint32_t colorPending;         // manifolds still uncoloured after the last colouring round enqueued
    uint32_t specOverflow;         // a speculative bound (tiles / contact-tiles capacity) was exceeded on the device
    uint32_t numCells;             // cells of this step's broad-phase grid
    uint32_t numPairsFound;        // pair count of a step whose speculative pair bound was exceeded (numPairs is zeroed then)
    uint32_t tailRounds;           // colouring rounds k_bin_hist ran itself this step (colorTail; 0: the enqueued rounds were enough)
    uint32_t numHmManifolds;       // heightmap terrain: manifolds the terrain contacts were grouped into (up to four contacts of one collider each; k_emit_manifolds)
    uint32_t reserved14[14];
    uint32_t numEvents;            // collision begin / end events of this step (when events are enabled)
    uint32_t numInterPairs;        // AABB overlaps between a rigid-body collider and a trigger / force-field collider
    uint32_t numInteractions;      // ... of which the boolean overlap test passed (non_collision_interaction records)
    uint32_t numHmContacts;        // heightmap terrain: contacts of this step (one pair record each; four consecutive ones of a collider form a manifold) ...
    uint32_t numHmColliders;       // ... and the colliders they belong to (= the reference's collision count for the terrain)
    uint32_t numEpa;               // intersecting GJK pairs queued for k_narrow_epa
    uint32_t xcdCount[8];          // XCD-partitioned solver: tiles owned by each XCD (k_build_tiles)
    uint32_t xccOf[8];             // ... and the hardware XCC id the workgroups with blockIdx % 8 == i really ran on (0xFFFFFFFF = none yet)
    uint32_t numCellsNext;         // cells of the grid k_pair_finish prepared for the next step
    uint32_t numDead;              // sharded world: colliders of bodies this rank does not simulate this step (they take no part in the broad phase)
    uint32_t shardOwned[3];        // sharded world: bodies / manifolds / contacts OWNED by this rank (owner rule: the manifold's first dynamic body)
    uint32_t shardSent[8];         // sharded world: records packed for each neighbour this step (slot order of ShardParams::peers)
    uint32_t shardRecv[8];         // ... records the neighbours packed for this rank (the headers of the received messages); [.] = 0xFFFFFFFF: that message was cut short (library transport, adaptive sizes)
    uint32_t seamStats[3];         // exact seam (include/mi_shard.h): manifolds of the seam class, colours they use, violations of this step (k_seam_stats)
    unsigned long long axisSums[9]; // centre statistics of the colliders this world counts (k_pair_finish): S1[3], S2lo[3], S2hi[3]; a sharded world's are added over the ranks
};

// Sum-only counters are sharded over 16 cache lines: a same-address global atomic sustains only ~90 ops/us on this
// chip (one L2 channel), so thousands of workgroups adding to ONE word serialise a whole kernel behind it.
constexpr uint32_t kShards = 16;
struct ShardCounters { uint32_t numOverlaps; uint32_t bucketHist[24]; uint32_t owned[3]; uint32_t boxHits; uint32_t pad[3]; };   // one 128-byte line per shard; owned: sharded world (bodies / manifolds / contacts of this rank);
                                                                                                                                   // boxHits: box pairs that passed the SAT, per queue (k_narrow -> k_narrow_clip; queue q = shard q — its own line: the 16 counters side by side in
                                                                                                                                   // ONE line of StepScalars took every workgroup's returning atomic through one L2 line)
static_assert(sizeof(ShardCounters) == 128 && kShards == 16, "one line per shard; the box queues use the shards' lines");
struct Shards { ShardCounters c[kShards]; uint32_t extentHist[kShards][256]; };

// Sharded world, "a rank pays for what it simulates" (round 5): the per-body and per-collider passes of a step visit BLOCKS of 256 bodies / colliders, and skip the blocks
// in which this rank simulates nothing — in an 8-tile scene 7 of 8.  Per body block: `stamp` = the step in which it last held a simulated body or received a record
// (k_shard_classify, k_shard_unpack; a block is RECENT for kShardRecentSteps steps after that: classification and packing only look at recent blocks — a body can only
// become simulated here by moving while simulated or by a record arriving, and both body-flag arrays have seen their zeros by then), `live` = a body simulated in this
// step or the one before (what the integrators and the collider pass ask).  Per collider block: the range of body blocks its colliders' bodies lie in (static, from the
// upload) and `cbLive` = it holds a collider that is not dead (k_bp_prepare writes it; the sorted scatter and the centre statistics skip the rest).  A world that is not
// sharded passes null pointers and launches one workgroup per block as before.
constexpr uint32_t kShardRecentSteps = 2u, kShardGrid = 1024u;   // kShardGrid: workgroups of a pass that strides over the blocks
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* __restrict__ p, uint32_t value, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = value; }
__device__ __forceinline__ bool shardBlockRecent(const uint32_t* __restrict__ stamp, uint32_t blk, uint32_t step) { return !stamp || (int32_t)(step - stamp[blk]) <= (int32_t)kShardRecentSteps; }
// Workgroup `first` of `stride` visits the blocks first, first + stride, ... < numBlocks that pass `live`: the tests of up to 64 candidates are made by the lanes of a wave
// side by side (one round of loads instead of one dependent load per skipped block: 7 of 8 candidates are skipped in an 8-tile scene), then `visit(block)` runs for the
// survivors, in order.  Every wave of the workgroup computes the same mask from the same words, so `visit` may contain workgroup barriers.
// STRIDED = false: the launch has one workgroup per block (a world that is not sharded) — the pass is compiled without the loop (with it, k_bp_prepare needs 157 instead of
// 128 registers and loses a quarter of its occupancy: 25.8 -> 31.4 us).
template <bool STRIDED = true, class Live, class Visit>
__device__ __forceinline__ void forLiveBlocks(const uint32_t first, const uint32_t stride, const uint32_t numBlocks, Live live, Visit visit) {
    if constexpr (!STRIDED) { visit(first); return; }
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = first; base < numBlocks; base += 64u * stride) {
        const uint32_t cand = base + lane * stride;
        unsigned long long todo = __ballot(cand < numBlocks && live(cand));
        while (todo) { const uint32_t j = (uint32_t)__ffsll((long long)todo) - 1u; todo &= todo - 1ull; visit(base + j * stride); }
    }
}
__device__ __forceinline__ int orderedInt(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float fromOrderedInt(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__device__ __forceinline__ uint32_t bucketOf(uint32_t ta, uint32_t tb) { return ta * 6u - ta * (ta - 1u) / 2u + (tb - ta); }

// ------------------------------------------------------------------------------------------------
// K1 "Get world space colliders" (src/physics/physics.cpp:631-756)
// One lane per collider (world index = reverse creation order).  in: 48 B local shape + 32 B pose
// (gathered by body) ; out: 48 B world shape + 2 x 16 B AABB rows whose .w carry the type/object
// tags and the body index, so the broad phase never touches another array.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void growBox(V3& mn, V3& mx, V3 o) { mn = vmin(mn, o); mx = vmax(mx, o); }
__device__ inline void boxToAABB(V3 lmn, V3 lmx, Q4 rot, V3 tr, V3& mn, V3& mx) {  // bounding_volumes.cpp:58-70
    mn = V3(FLT_MAX); mx = V3(-FLT_MAX);
    growBox(mn, mx, rotate(rot, lmn) + tr);
    growBox(mn, mx, rotate(rot, V3(lmx.x, lmn.y, lmn.z)) + tr);
    growBox(mn, mx, rotate(rot, V3(lmn.x, lmx.y, lmn.z)) + tr);
    growBox(mn, mx, rotate(rot, V3(lmx.x, lmx.y, lmn.z)) + tr);
    growBox(mn, mx, rotate(rot, V3(lmn.x, lmn.y, lmx.z)) + tr);
    growBox(mn, mx, rotate(rot, V3(lmx.x, lmn.y, lmx.z)) + tr);
    growBox(mn, mx, rotate(rot, V3(lmn.x, lmx.y, lmx.z)) + tr);
    growBox(mn, mx, rotate(rot, lmx) + tr);
}

// one collider: world shape + AABB rows (written unless the collider is dead in this and the previous step: its rows already hold the dead box);
// mnOut / mxOut = the rows in either case
__device__ __forceinline__ void worldCollider(
    uint32_t k, uint32_t nb, const uint32_t* __restrict__ cTypeBody,  // [2*nc]: type, body (kNoBody = static)
    const uint32_t* __restrict__ cObject,   // colliders without a body: physics_object_type | object index << 8 (static / force field / trigger)
    const float4* __restrict__ cShape, const float4* __restrict__ cStaticPos, const float4* __restrict__ cStaticRot,
    const float4* __restrict__ bPos, const float4* __restrict__ bRot,
    const float4* __restrict__ hullAabb,  // [2*numHulls]
    float4* __restrict__ wShape, float4* __restrict__ aabbMin, float4* __restrict__ aabbMax,
    const uint8_t* __restrict__ bodyActive /* sharded world: 0 = body not simulated by this rank this step, or null */,
    const uint8_t* __restrict__ bodyActivePrev /* ... and in the previous step */, float4& mnOut, float4& mxOut) {
    uint32_t type = cTypeBody[2 * k], body = cTypeBody[2 * k + 1];
    if (bodyActive && body != kNoBody && !bodyActive[body]) {
        // a DEAD collider: inverted box (overlaps nothing, centre exactly 0 so the axis statistics are unaffected), skipped by the grid
        mnOut = make_float4(kDeadBox, kDeadBox, kDeadBox, __uint_as_float(type | (OBJ_RIGID_BODY << 8)));
        mxOut = make_float4(-kDeadBox, -kDeadBox, -kDeadBox, __uint_as_float(body));
        if (!bodyActivePrev[body]) return;   // dead before as well: its rows already hold this (most colliders of a many-tile scene, every step)
        wShape[3 * k] = make_float4(0, 0, 0, 0); wShape[3 * k + 1] = make_float4(0, 0, 0, 0); wShape[3 * k + 2] = make_float4(0, 0, 0, 1);
        aabbMin[k] = mnOut; aabbMax[k] = mxOut;
        return;
    }
    V3 tp; Q4 tr; uint32_t objType, objIndex;
    if (body != kNoBody) { tp = xyz(bPos[body]); tr = toQ(bRot[body]); objType = OBJ_RIGID_BODY; objIndex = body; }
    else {
        tp = xyz(cStaticPos[k]); tr = toQ(cStaticRot[k]);
        uint32_t o = cObject[k];
        objType = o & 0xFFu; objIndex = objType == OBJ_STATIC ? nb : (o >> 8);
    }
    float4 s0 = cShape[3 * k], s1 = cShape[3 * k + 1], s2 = cShape[3 * k + 2];
    float4 o0 = make_float4(0, 0, 0, 0), o1 = o0, o2 = make_float4(0, 0, 0, 1);
    V3 mn, mx;
    uint32_t wtype = type;
    switch (type) {
        case T_SPHERE: {
            V3 c = tp + rotate(tr, xyz(s0));
            mn = c - V3(s0.w); mx = c + V3(s0.w);
            o0 = f4(c, s0.w);
        } break;
        case T_CAPSULE: {
            V3 pa = rotate(tr, xyz(s0)) + tp, pb = rotate(tr, V3(s0.w, s1.x, s1.y)) + tp;
            float r = s1.z; V3 r3(r);
            mn = V3(FLT_MAX); mx = V3(-FLT_MAX);
            growBox(mn, mx, pa + r3); growBox(mn, mx, pa - r3); growBox(mn, mx, pb + r3); growBox(mn, mx, pb - r3);
            o0 = f4(pa, r); o1 = f4(pb, 0.f);
        } break;
        case T_CYLINDER: {
            V3 pa = rotate(tr, xyz(s0)) + tp, pb = rotate(tr, V3(s0.w, s1.x, s1.y)) + tp;
            float r = s1.z;
            V3 a = pb - pa; float aa = dot(a, a);
            float x = 1.f - a.x * a.x / aa, y = 1.f - a.y * a.y / aa, z = 1.f - a.z * a.z / aa;
            x = sqrtf(fmaxr(0.f, x)); y = sqrtf(fmaxr(0.f, y)); z = sqrtf(fmaxr(0.f, z));
            V3 e = r * V3(x, y, z);
            mn = vmin(pa - e, pb - e); mx = vmax(pa + e, pb + e);
            o0 = f4(pa, r); o1 = f4(pb, 0.f);
        } break;
        case T_AABB: {
            V3 lmn = xyz(s0), lmx(s0.w, s1.x, s1.y);
            boxToAABB(lmn, lmx, tr, tp, mn, mx);
            if (isIdentity(tr)) { o0 = f4(mn, 0.f); o1 = f4(mx, 0.f); }
            else {  // promoted to OBB (physics.cpp:725-733)
                wtype = T_OBB;
                o0 = f4(rotate(tr, (lmn + lmx) * 0.5f) + tp, 0.f);
                o1 = f4((lmx - lmn) * 0.5f, 0.f);
                o2 = fromQ(tr);
            }
        } break;
        case T_OBB: {
            Q4 lrot(s0.x, s0.y, s0.z, s0.w); V3 lc(s1.x, s1.y, s1.z), lr(s1.w, s2.x, s2.y);
            Q4 wrot = tr * lrot; V3 wc = rotate(tr, lc) + tp;
            boxToAABB(-lr, lr, wrot, wc, mn, mx);
            o0 = f4(wc, 0.f); o1 = f4(lr, 0.f); o2 = fromQ(wrot);
        } break;
        default: {  // hull
            Q4 lrot(s0.x, s0.y, s0.z, s0.w); V3 lp(s1.x, s1.y, s1.z);
            uint32_t geom = __float_as_uint(s1.w);
            Q4 wrot = tr * lrot; V3 wp = rotate(tr, lp) + tp;
            boxToAABB(xyz(hullAabb[2 * geom]), xyz(hullAabb[2 * geom + 1]), wrot, wp, mn, mx);
            o0 = f4(wp, s1.w); o2 = fromQ(wrot);
        } break;
    }
    wShape[3 * k] = o0; wShape[3 * k + 1] = o1; wShape[3 * k + 2] = o2;
    mnOut = f4(mn, __uint_as_float(wtype | (objType << 8)));
    mxOut = f4(mx, __uint_as_float(objIndex));
    aabbMin[k] = mnOut; aabbMax[k] = mxOut;
}
__global__ __launch_bounds__(256) void k_world_colliders(
    uint32_t nc, uint32_t nb, const uint32_t* __restrict__ cTypeBody, const uint32_t* __restrict__ cObject,
    const float4* __restrict__ cShape, const float4* __restrict__ cStaticPos, const float4* __restrict__ cStaticRot,
    const float4* __restrict__ bPos, const float4* __restrict__ bRot, const float4* __restrict__ hullAabb,
    float4* __restrict__ wShape, float4* __restrict__ aabbMin, float4* __restrict__ aabbMax, StepScalars* sc, uint32_t axisCur,
    const uint8_t* __restrict__ bodyActive, const uint8_t* __restrict__ bodyActivePrev,
    const uint32_t* __restrict__ axisDev /* sharded world: the sweep axis lives on the device (k_shard_axis, from the sums over all ranks), or null */) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { sc->axisCur = axisDev ? *axisDev : axisCur; }   // the SAP axis chosen at the end of the previous (successful) step
    if (k >= nc) return;
    float4 mn, mx;
    worldCollider(k, nb, cTypeBody, cObject, cShape, cStaticPos, cStaticRot, bPos, bRot, hullAabb, wShape, aabbMin, aabbMax, bodyActive, bodyActivePrev, mn, mx);
}

// ------------------------------------------------------------------------------------------------
// Broad phase.  The reference sorts AABB endpoints on the max-variance axis and sweeps
// (src/physics/collision_broad.cpp:297-447).  The pair SET it produces is "all AABB-overlapping
// pairs"; here that set comes from a uniform grid over collider centres (cell >= every "small"
// extent => 27 neighbour cells suffice) plus a brute-force pass for the few "large" colliders
// (ground, walls).  The SAP axis is still tracked because the reference's A/B orientation of a
// same-type pair depends on sweep order along it.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t extentBin(float ext) {
    float l = log2f(fmaxr(ext, 1e-6f)) * 8.f + 128.f;
    return (uint32_t)fminr(fmaxr(l, 0.f), 255.f);
}
__device__ __forceinline__ float extentBinUpper(uint32_t b) { return exp2f(((float)b + 1.f - 128.f) / 8.f); }

// Centre statistics for the next sorting axis (the reference sums centres and squared centres in float, sequentially:
// collision_broad.cpp:376-384, 443-444).  Here the statistic must not depend on the order of the sum — nor on the PARTITION: in a sharded
// world (include/mi_shard.h) every rank sums the colliders it owns and the sums are added over the ranks (one 72-byte all-reduce), so the
// axis is the single world's whatever the tiling.  The centre is quantised to 1/1024 m (clamped to +-2^20 m) and q and q^2 are added as
// INTEGERS: S1 (two's complement, 64 bits), q^2 split into its low 32 bits and the rest (S2lo, S2hi: 2^26 colliders overflow neither).
// k_pair_finish compares n * S2 - S1^2 exactly in 128 bits.  (Mirrored by the oracle: ora::axisTerms / ora::axisFromSums.)
constexpr uint32_t kAxisSums = 9;   // S1[3], S2lo[3], S2hi[3]
__device__ __forceinline__ void axisTerms(float c, unsigned long long& q, unsigned long long& lo, unsigned long long& hi) {
    const float lim = 1048576.f;
    c = (c > -lim) ? c : -lim;   // (a NaN centre counts as -2^20, like in the oracle)
    c = (c < lim) ? c : lim;
    const long long qi = (long long)rintf(c * 1024.f);
    const unsigned long long sq = (unsigned long long)(qi * qi);
    q = (unsigned long long)qi; lo = sq & 0xFFFFFFFFull; hi = sq >> 32;
}
// argmax of the variance n * S2 - S1^2 per axis, exactly (128-bit integers), in the shape of collision_broad.cpp:443-444
__host__ __device__ inline uint32_t axisFromSums(const unsigned long long s[kAxisSums], uint32_t n) {
    unsigned __int128 var[3];
    for (int a = 0; a < 3; ++a) {
        const long long s1 = (long long)s[a];
        const unsigned __int128 s2 = ((unsigned __int128)s[6 + a] << 32) + (unsigned __int128)s[3 + a];
        const unsigned __int128 m = (unsigned __int128)(s1 < 0 ? (unsigned long long)(-s1) : (unsigned long long)s1);
        const unsigned __int128 ns2 = (unsigned __int128)n * s2, sq = m * m;
        var[a] = ns2 > sq ? ns2 - sq : (unsigned __int128)0;   // (>= 0 by Cauchy-Schwarz; a rank's partial sums with the global n always satisfy it too)
    }
    return (var[0] > var[1]) ? ((var[0] > var[2]) ? 0u : 2u) : ((var[1] > var[2]) ? 1u : 2u);
}
// Does this world count collider (mn, mx) in its centre statistics?  All of them — or, sharded, those of the bodies this rank OWNS plus,
// on rank 0 only, the colliders without a rigid body (statics, triggers, force fields are replicated on every rank).
__device__ __forceinline__ bool axisCounted(const float4& mn, const float4& mx, const uint8_t* __restrict__ bodyActive, uint32_t countUnowned) {
    if (!bodyActive) return true;
    const uint32_t objType = (__float_as_uint(mn.w) >> 8) & 0xFFu;
    return objType == OBJ_RIGID_BODY ? bodyActive[__float_as_uint(mx.w)] == 1u : countUnowned != 0u;
}
__device__ __forceinline__ void axisAccumulate(bool counted, float cx, float cy, float cz, unsigned long long v[kAxisSums]) {
#pragma unroll
    for (uint32_t c = 0; c < kAxisSums; ++c) v[c] = 0ull;
    if (!counted) return;
    axisTerms(cx, v[0], v[3], v[6]); axisTerms(cy, v[1], v[4], v[7]); axisTerms(cz, v[2], v[5], v[8]);
}
// wave sums (any order: integers) -> sm[wave][9]; after a barrier thread 0 adds the four and writes the block's partial
__device__ __forceinline__ void axisWaveReduce(unsigned long long v[kAxisSums], unsigned long long (*sm)[kAxisSums]) {
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (uint32_t c = 0; c < kAxisSums; ++c) v[c] += (unsigned long long)__shfl_down((long long)v[c], off, 64);
    }
    if ((threadIdx.x & 63u) == 0u) { for (uint32_t c = 0; c < kAxisSums; ++c) sm[threadIdx.x >> 6][c] = v[c]; }
}
__global__ __launch_bounds__(256) void k_axis_partials(uint32_t nc, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
                                                       unsigned long long* __restrict__ partials, Shards* sh,
                                                       const uint8_t* __restrict__ bodyActive /* sharded world: this step's body flags, or null */, uint32_t countUnowned) {
    __shared__ unsigned long long sm[4][kAxisSums];
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long v[kAxisSums];
    bool counted = false; float cx = 0.f, cy = 0.f, cz = 0.f;
    if (i < nc) {
        float4 mn = aabbMin[i], mx = aabbMax[i];
        cx = (mn.x + mx.x) * 0.5f; cy = (mn.y + mx.y) * 0.5f; cz = (mn.z + mx.z) * 0.5f;
        counted = axisCounted(mn, mx, bodyActive, countUnowned);
        const float ext = fmaxr(fmaxr(mx.x - mn.x, mx.y - mn.y), mx.z - mn.z);
        if (!(mx.x < mn.x)) atomicAdd(&hist[extentBin(ext)], 1u);   // only steers the cell size, never results; dead colliders (sharded world) stay out: the histogram total = live colliders
    }
    axisAccumulate(counted, cx, cy, cz, v);
    axisWaveReduce(v, sm);
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&sh->extentHist[blockIdx.x & (kShards - 1u)][threadIdx.x], hist[threadIdx.x]);
    if (threadIdx.x < kAxisSums) partials[blockIdx.x * kAxisSums + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}
// Cell size = smallest extent bin edge that leaves at most `limit` colliders above it; those few "large"
// colliders (ground, walls, outliers) are handled by the brute-force pass.
__global__ __launch_bounds__(256) void k_bp_threshold(uint32_t nc, const Shards* __restrict__ sh, StepScalars* sc) {
    __shared__ uint32_t hist[256];
    uint32_t v = 0;
    for (uint32_t k = 0; k < kShards; ++k) v += sh->extentHist[k][threadIdx.x];
    hist[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x != 0) return;
    uint32_t limit = max(16u, nc / 16384u);
    uint32_t costCap = (uint32_t)(67108864ull / (uint64_t)max(nc, 1u));
    limit = max(8u, min(limit, costCap));
    uint32_t above = 0; int b = 255;
    for (; b >= 0; --b) { if (above + hist[b] > limit) break; above += hist[b]; }
    sc->largeThreshold = b < 0 ? 0.f : extentBinUpper((uint32_t)b);
}

__global__ __launch_bounds__(256) void k_bp_classify(uint32_t nc, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
                                                     StepScalars* sc, uint32_t* __restrict__ largeList, uint32_t* __restrict__ isLarge,
                                                     int* __restrict__ blockBounds) {
    __shared__ int sb[4][6];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float thr = sc->largeThreshold;
    int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    bool dead = false;
    if (i < nc) {
        float4 mn = aabbMin[i], mx = aabbMax[i];
        float ext = fmaxr(fmaxr(mx.x - mn.x, mx.y - mn.y), mx.z - mn.z);
        bool large = ext > thr;
        dead = mx.x < mn.x;                                   // sharded world: not simulated by this rank (k_world_colliders)
        isLarge[i] = dead ? 2u : large ? 1u : 0u;             // anything non-zero keeps the collider out of the grid
        if (dead) {}
        else if (large) { uint32_t slot = atomicAdd(&sc->numLarge, 1u); largeList[slot] = i; }
        else {
            lo[0] = hi[0] = orderedInt((mn.x + mx.x) * 0.5f);
            lo[1] = hi[1] = orderedInt((mn.y + mx.y) * 0.5f);
            lo[2] = hi[2] = orderedInt((mn.z + mx.z) * 0.5f);
        }
    }
    { const unsigned long long deadMask = __ballot(dead); if (deadMask && (threadIdx.x & 63u) == 0u) atomicAdd(&sc->numDead, (uint32_t)__popcll(deadMask)); }
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], __shfl_xor(lo[a], off, 64)); hi[a] = max(hi[a], __shfl_xor(hi[a], off, 64)); }
    uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; ++a) { sb[wv][a] = lo[a]; sb[wv][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        int v = sb[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? min(v, sb[w][threadIdx.x]) : max(v, sb[w][threadIdx.x]);
        blockBounds[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void k_bp_grid_setup(uint32_t nc, uint32_t numBlocks, uint32_t cellCap, const int* __restrict__ blockBounds, StepScalars* sc, GridParams* g) {
    __shared__ int red[4][6];
    int v[6] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (uint32_t b = threadIdx.x; b < numBlocks; b += 256)
        for (int a = 0; a < 6; ++a) { int x = blockBounds[b * 6 + a]; v[a] = a < 3 ? min(v[a], x) : max(v[a], x); }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v[a], d, 64); v[a] = a < 3 ? min(v[a], o) : max(v[a], o); }
    if ((threadIdx.x & 63u) == 0) for (int a = 0; a < 6; ++a) red[threadIdx.x >> 6][a] = v[a];
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int t = 1; t < 4; ++t) for (int a = 0; a < 6; ++a) v[a] = a < 3 ? min(v[a], red[t][a]) : max(v[a], red[t][a]);
    float thr = sc->largeThreshold;
    float cell = thr * 1.001f + 1e-6f;
    float lo[3], hi[3];
    bool any = v[0] != 0x7FFFFFFF;
    for (int a = 0; a < 3; ++a) { lo[a] = any ? fromOrderedInt(v[a]) : 0.f; hi[a] = any ? fromOrderedInt(v[3 + a]) : 0.f; }
    for (int it = 0; it < 64; ++it) {
        double cells = 1.0;
        for (int a = 0; a < 3; ++a) { uint32_t d = (uint32_t)((hi[a] - lo[a]) / cell) + 2u; g->dims[a] = d; cells *= (double)d; }
        if (cells <= (double)(cellCap - 1)) break;   // the host sized the cell table (histogram + scan) for cellCap cells
        cell *= 1.3f;
    }
    g->numCells = g->dims[0] * g->dims[1] * g->dims[2];
    sc->numCells = g->numCells;
    g->cell = cell; g->invCell = 1.f / cell;
    for (int a = 0; a < 3; ++a) g->origin[a] = lo[a];
    g->numLarge = sc->numLarge + sc->numDead;   // everything that is not in the cell-sorted arrays
    g->largeThreshold = thr;
}

__device__ __forceinline__ void cellOf(const GridParams& g, float cx, float cy, float cz, uint32_t& ix, uint32_t& iy, uint32_t& iz) {
    ix = min((uint32_t)fmaxr(0.f, (cx - g.origin[0]) * g.invCell), g.dims[0] - 1u);
    iy = min((uint32_t)fmaxr(0.f, (cy - g.origin[1]) * g.invCell), g.dims[1] - 1u);
    iz = min((uint32_t)fmaxr(0.f, (cz - g.origin[2]) * g.invCell), g.dims[2] - 1u);
}

// Cell id of every small collider + its arrival rank inside the cell (the returned value of the histogram atomic):
// after the exclusive scan of the histogram, sorted position = cellLower[key] + rank — a counting sort with no sort
// pass.  The order inside a cell is arbitrary; nothing downstream depends on it (pairs are keyed by collider index).
__global__ __launch_bounds__(256) void k_bp_cell_ids(uint32_t nc, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
                                                     const uint32_t* __restrict__ isLarge, const GridParams* __restrict__ gp,
                                                     uint32_t* __restrict__ keys, uint32_t* __restrict__ ranks, uint32_t* __restrict__ cellCount) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    GridParams g = *gp;
    uint32_t key = 0xFFFFFFFFu, rank = 0;
    if (!isLarge[i]) {
        float4 mn = aabbMin[i], mx = aabbMax[i];
        uint32_t ix, iy, iz;
        cellOf(g, (mn.x + mx.x) * 0.5f, (mn.y + mx.y) * 0.5f, (mn.z + mx.z) * 0.5f, ix, iy, iz);
        key = (ix * g.dims[1] + iy) * g.dims[2] + iz;
        rank = atomicAdd(&cellCount[key], 1u);
    }
    keys[i] = key; ranks[i] = rank;
}


// Steps after the first use the grid computed at the END OF THE PREVIOUS STEP (k_pair_finish): threshold, cell size, origin and dims
// only steer which colliders go through the grid and how fine it is, never the pair set — every small collider still has an extent
// <= the cell (it is classified against the same threshold), and centres outside the old bounds clamp to the rim cells, which keeps
// neighbours neighbours.  That takes k_bp_threshold and k_bp_grid_setup off the step's critical path and lets ONE kernel do what
// k_axis_partials, k_bp_classify and k_bp_cell_ids did: centre statistics (same fixed reduction shape), extent histogram,
// dead / large / small classification, bounds of the small centres, cell id + arrival rank.
template <bool STRIDED>
__global__ __launch_bounds__(256) void k_bp_prepare(uint32_t nc, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax, const GridParams* __restrict__ gp,
                                                    unsigned long long* __restrict__ partials, Shards* sh, StepScalars* sc, uint32_t* __restrict__ largeList, uint32_t* __restrict__ isLarge,
                                                    int* __restrict__ blockBounds, uint32_t* __restrict__ keys, uint32_t* __restrict__ ranks, uint32_t* __restrict__ cellCount,
                                                    const uint8_t* __restrict__ bodyActivePrev /* sharded world: the previous step's body flags, or null */,
                                                    const uint8_t* __restrict__ bodyActive /* sharded world: this step's body flags, or null */, uint32_t countUnowned,
                                                    // FUSED with k_world_colliders (cTypeBody non-null): the lane computes its collider's world shape and AABB first and goes on
                                                    // with them in registers — one launch and one pass over the AABB rows less (the grid it classifies against is the previous step's)
                                                    uint32_t nb, const uint32_t* __restrict__ cTypeBody, const uint32_t* __restrict__ cObject, const float4* __restrict__ cShape,
                                                    const float4* __restrict__ cStaticPos, const float4* __restrict__ cStaticRot, const float4* __restrict__ bPos, const float4* __restrict__ bRot,
                                                    const float4* __restrict__ hullAabb, float4* __restrict__ wShape, float4* __restrict__ aabbMinW, float4* __restrict__ aabbMaxW,
                                                    uint32_t axisCur, const uint32_t* __restrict__ axisDev,
                                                    const uint2* __restrict__ cbRange /* sharded world: per collider block the body blocks its colliders' bodies lie in (x > y: always visited), or null */,
                                                    const uint8_t* __restrict__ blockLive, uint8_t* __restrict__ cbLive) {
    __shared__ unsigned long long sm[4][kAxisSums];
    __shared__ uint32_t hist[256];
    __shared__ int sb[4][6];
  // the step's sweep axis: written by the first workgroup whatever blocks it goes on to visit (a sharded rank that simulates nothing in collider block 0 skips that block's
  // body below; the axis word must follow the exchange's axisDev all the same, or this rank orients its pairs along a stale axis)
  if (cTypeBody && blockIdx.x == 0 && threadIdx.x == 0) sc->axisCur = axisDev ? *axisDev : axisCur;
  // (one workgroup per collider block unless the world is sharded: then a collider block is skipped when nothing is simulated, now or in the previous step, in any body
  // block it refers to — its rows already say "dead", its partial results are empty)
  forLiveBlocks<STRIDED>(blockIdx.x, gridDim.x, (nc + 255u) / 256u, [&](uint32_t cb) {
        if (!cbRange) return true;
        const uint2 rg = cbRange[cb];
        bool any = rg.x > rg.y;
        for (uint32_t b = rg.x; b <= rg.y && !any; ++b) any = blockLive[b] != 0u;
        if (!any && cbLive[cb]) cbLive[cb] = 0u;
        return any;
    }, [&](uint32_t cb) {
    __syncthreads();   // (the previous block's shared results have been read)
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = cb * 256 + threadIdx.x;
    if (cTypeBody && bodyActive) {
        // sharded world: a workgroup whose colliders are all dead now and were dead in the previous step (7 of 8 workgroups of an 8-tile scene) has nothing to
        // compute, nothing to reduce and nothing to rewrite but its own (empty) partial results
        bool stale = true;
        if (i < nc) { const uint32_t body = cTypeBody[2 * i + 1]; stale = body != kNoBody && !bodyActive[body] && !bodyActivePrev[body]; }
        if (!__syncthreads_or(stale ? 0 : 1)) {
            if (threadIdx.x < kAxisSums) partials[cb * kAxisSums + threadIdx.x] = 0ull;
            if (threadIdx.x < 6) blockBounds[cb * 6 + threadIdx.x] = threadIdx.x < 3 ? 0x7FFFFFFF : (int)0x80000000;
            if (cbLive && threadIdx.x == 0) cbLive[cb] = 0u;
            return;
        }
    }
    const GridParams g = *gp;
    unsigned long long v[kAxisSums];
    int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    bool dead = false, counted = false; float cx = 0.f, cy = 0.f, cz = 0.f;
    if (i < nc) {
        float4 mn, mx;
        if (cTypeBody) worldCollider(i, nb, cTypeBody, cObject, cShape, cStaticPos, cStaticRot, bPos, bRot, hullAabb, wShape, aabbMinW, aabbMaxW, bodyActive, bodyActivePrev, mn, mx);
        else { mn = aabbMin[i]; mx = aabbMax[i]; }
        cx = (mn.x + mx.x) * 0.5f; cy = (mn.y + mx.y) * 0.5f; cz = (mn.z + mx.z) * 0.5f;
        counted = axisCounted(mn, mx, bodyActive, countUnowned);
        const float ext = fmaxr(fmaxr(mx.x - mn.x, mx.y - mn.y), mx.z - mn.z);
        dead = mx.x < mn.x;
        if (!dead) atomicAdd(&hist[extentBin(ext)], 1u);   // only steers the NEXT step's cell size; its total = live colliders (k_pair_finish derives numDead from it:
                                                            // a counter bumped once per wave of dead colliders cost 0.3 ms in an 8-tile world, ~90 same-address atomics per us)
        const bool large = ext > g.largeThreshold;
        // a collider that was dead in the previous step too already has (2, 0xFFFFFFFF, 0) in these rows
        const bool stale = dead && bodyActivePrev && !bodyActivePrev[__float_as_uint(mx.w)];
        if (!stale) isLarge[i] = dead ? 2u : large ? 1u : 0u;
        uint32_t key = 0xFFFFFFFFu, rank = 0;
        if (dead) {}
        else if (large) { uint32_t slot = atomicAdd(&sc->numLarge, 1u); largeList[slot] = i; }
        else {
            lo[0] = hi[0] = orderedInt(cx); lo[1] = hi[1] = orderedInt(cy); lo[2] = hi[2] = orderedInt(cz);
            uint32_t ix, iy, iz;
            cellOf(g, cx, cy, cz, ix, iy, iz);
            key = (ix * g.dims[1] + iy) * g.dims[2] + iz;
            rank = atomicAdd(&cellCount[key], 1u);
        }
        if (!stale) { keys[i] = key; ranks[i] = rank; }
    }
    axisAccumulate(counted, cx, cy, cz, v);
    axisWaveReduce(v, sm);
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], __shfl_xor(lo[a], off, 64)); hi[a] = max(hi[a], __shfl_xor(hi[a], off, 64)); }
    }
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { for (int a = 0; a < 3; ++a) { sb[wv][a] = lo[a]; sb[wv][3 + a] = hi[a]; } }
    const int anyAlive = __syncthreads_or((i < nc && !dead) ? 1 : 0);
    if (cbLive && threadIdx.x == 0) cbLive[cb] = anyAlive ? 1u : 0u;
    if (hist[threadIdx.x]) atomicAdd(&sh->extentHist[cb & (kShards - 1u)][threadIdx.x], hist[threadIdx.x]);
    if (threadIdx.x < kAxisSums) partials[cb * kAxisSums + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
    if (threadIdx.x < 6) {
        int x = sb[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) x = threadIdx.x < 3 ? min(x, sb[w][threadIdx.x]) : max(x, sb[w][threadIdx.x]);
        blockBounds[cb * 6 + threadIdx.x] = x;
    }
  });
}

// Cell-sorted copies of the AABB rows (a column scan reads contiguous memory), the cell key and the collider index.
// Positions [numSmall, nc) keep the key 0xFFFFFFFF written by the host-side fill.
template <bool STRIDED>
__global__ __launch_bounds__(256) void k_bp_scatter_sorted(uint32_t nc, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ranks,
                                                           const uint32_t* __restrict__ cellLower,
                                                           const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
                                                           uint32_t* __restrict__ keysS, uint32_t* __restrict__ valsS,
                                                           float4* __restrict__ sMin, float4* __restrict__ sMax, const uint8_t* __restrict__ cbLive /* sharded world: collider blocks with a live collider, or null */) {
    const uint32_t numCb = (nc + blockDim.x - 1u) / blockDim.x;
    forLiveBlocks<STRIDED>(blockIdx.x, gridDim.x, numCb, [&](uint32_t cb) { return !cbLive || cbLive[cb] != 0u; }, [&](uint32_t cb) {
        const uint32_t i = cb * blockDim.x + threadIdx.x;
        if (i >= nc) return;
        const uint32_t key = keys[i];
        if (key == 0xFFFFFFFFu) return;
        const uint32_t pos = cellLower[key] + ranks[i];
        keysS[pos] = key; valsS[pos] = i;
        sMin[pos] = aabbMin[i]; sMax[pos] = aabbMax[i];
    });
}

// Prune + orient + key (collision_narrow.cpp:2346-2395) fused into pair emission.
// i, j: collider world indices.  The SAP sweep emits {new, active}: new = later start on the axis;
// on a tie the later-created collider (smaller world index) is the newer endpoint.
// Returns false when the overlap generates no collision pair.
// `inter` (optional): AABB overlaps between a rigid-body collider and a trigger / force-field collider are appended there
// (rare; plain atomic append) for the boolean overlap tests of k_overlap.
struct InterSink { uint64_t* keys; uint32_t cap; uint32_t* count; };
__device__ __forceinline__ bool pairKey(uint32_t i, const float4& imn, const float4& imx, uint32_t j, const float4& jmn, const float4& jmx,
                                        uint32_t axis, uint64_t& key, const InterSink& inter = InterSink{nullptr, 0u, nullptr}) {
    uint32_t ti = __float_as_uint(imn.w), tj = __float_as_uint(jmn.w);
    uint32_t oi = (ti >> 8) & 0xFF, oj = (tj >> 8) & 0xFF;
    uint32_t bi = __float_as_uint(imx.w), bj = __float_as_uint(jmx.w);
    if (oi != OBJ_RIGID_BODY && oj != OBJ_RIGID_BODY) return false;
    if (oi == OBJ_RIGID_BODY && oj == OBJ_RIGID_BODY && bi == bj) return false;
    float mi_ = axis == 0 ? imn.x : (axis == 1 ? imn.y : imn.z);
    float mj_ = axis == 0 ? jmn.x : (axis == 1 ? jmn.y : jmn.z);
    bool iIsNew = (mi_ > mj_) || (mi_ == mj_ && i < j);
    uint32_t a = iIsNew ? i : j, b = iIsNew ? j : i;
    uint32_t ta = (iIsNew ? ti : tj) & 0xFF, tb = (iIsNew ? tj : ti) & 0xFF;
    uint32_t oa = iIsNew ? oi : oj, ob = iIsNew ? oj : oi;
    if (!(ta < tb)) { uint32_t t = a; a = b; b = t; t = ta; ta = tb; tb = t; t = oa; oa = ob; ob = t; }
    bool collision = (oa == OBJ_RIGID_BODY && ob == OBJ_RIGID_BODY) || oa == OBJ_STATIC || ob == OBJ_STATIC;
    key = ((uint64_t)bucketOf(ta, tb) << 58) | ((uint64_t)a << 29) | (uint64_t)b;
    if (!collision) {   // rigid body vs trigger / force field (collision_narrow.cpp:2385-2394): boolean overlap test later
        if (inter.keys) { uint32_t slot = atomicAdd(inter.count, 1u); if (slot < inter.cap) inter.keys[slot] = key; }
        return false;
    }
    return true;
}


// (A software version of a keyed LDS histogram increment — one atomic per distinct key of the wave, lanes ranked by ballot — was
// measured against plain same-address LDS atomics in k_manifold_keys / k_bin_hist / k_bin_scatter: 2-2.5x SLOWER; the LDS resolves
// the conflicts faster than a loop over the distinct keys does.)

// Wave-aggregated append: one atomic per wave per call (ballot + popcount prefix), not one per pair.
__device__ __forceinline__ void waveAppendKey(bool want, uint64_t key, uint64_t* __restrict__ pairKeys, uint32_t pairCap, uint32_t* counter) {
    unsigned long long mask = __ballot(want);
    if (!want) return;
    uint32_t lane = threadIdx.x & 63u;
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader, 64);
    uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (slot < pairCap) pairKeys[slot] = key;
}
__device__ __forceinline__ void waveAddCount(uint32_t v, uint32_t* counter) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(counter, v);
}

__device__ __forceinline__ bool aabbOverlap(const float4& amn, const float4& amx, const float4& bmn, const float4& bmx) {  // bounding_volumes.h:352-358
    if (amx.x < bmn.x || amn.x > bmx.x) return false;
    if (amx.y < bmn.y || amn.y > bmx.y) return false;
    if (amx.z < bmn.z || amn.z > bmx.z) return false;
    return true;
}

// Five lanes per small collider, one per forward neighbour COLUMN.  Colliders are sorted by cell key with z fastest,
// so the cells (x', y', z-1 .. z+1) of one neighbour column are ONE contiguous range of the sorted arrays:
// [cellLower[first], cellLower[last + 1]) with cellLower = exclusive prefix sum of the cell histogram.
//   column 0: (0,0,[z .. z+1]) starting after the collider itself     column 1: (0,+1,[z-1 .. z+1])
//   columns 2..4: (+1,{-1,0,+1},[z-1 .. z+1])            -> each unordered cell pair is visited once.
// A workgroup handles 256 CONSECUTIVE sorted colliders for ONE column, so its lanes walk (nearly) the same candidate
// range at the same time: loads are shared through L1 and candidates are fetched four at a time (8 loads in flight
// per lane) instead of one dependent load pair per loop trip.
constexpr uint32_t kPairOverflow = 512;   // block-shared overflow slots of k_bp_pairs_grid (4 KiB)
constexpr uint32_t kPairBuf = 6;      // LDS-staged pair keys per collider-column before the block-level flush
constexpr uint32_t kGridChunks = 2;   // a workgroup handles 2 x 256 consecutive sorted colliders for one column (24 KiB of staging: 6 workgroups per CU; measured 1: 168, 2: 148, 3: 154, 4: 171 us for the broad phase)

// Pair compaction: a same-address global atomic sustains only ~90 ops/us on this chip, so per-pair, per-wave or even
// per-256-lane-block atomics on one word bound the whole broad phase.  Each lane stages its hits in LDS (24 KiB per
// workgroup), the block prefix-sums the per-lane counts (wave shuffles), ONE returning atomic reserves the block's
// output range for its 512 colliders and the keys are copied out; a lane with more than kPairBuf hits in one column
// appends the excess directly (rare).  Sum-only counters go to the block's shard line.
// LDS of the pair kernels (one layout for both bodies, so that ONE launch can run them side by side: k_bp_pairs)
struct PairLds {
    uint64_t buf[kGridChunks * 256 * kPairBuf];
    uint64_t ovf[kPairOverflow];   // second chance of a lane whose own kPairBuf slots are full (dense piles: ~3 % of the lane-columns); LDS atomics,
    uint32_t ovfCount;             // not one same-address GLOBAL atomic per excess pair (that serialised the kernel in the settled pile: 473 us)
    uint32_t waveTotals[4];
    uint32_t blockBase;
    uint32_t bhist[32];            // [0..20] bucket histogram, [31] overlaps
};
__device__ __forceinline__ void bpPairsGridBody(PairLds& L, const uint32_t blockId /* workgroup of the grid pass */, uint32_t nc, uint32_t blocksPerColumn, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                const float4* __restrict__ sMin, const float4* __restrict__ sMax,
                                                const uint32_t* __restrict__ cellLower,
                                                const GridParams* __restrict__ gp, uint64_t* __restrict__ pairKeys, uint32_t pairCap,
                                                StepScalars* sc, Shards* sh, InterSink inter) {
    uint64_t* const buf = L.buf; uint64_t* const ovf = L.ovf; uint32_t& ovfCount = L.ovfCount; uint32_t* const waveTotals = L.waveTotals; uint32_t& blockBase = L.blockBase; uint32_t* const bhist = L.bhist;
    const uint32_t col = blockId / blocksPerColumn;
    // blocksPerColumn is a multiple of 8: the workgroups of one XCD (blockIdx % 8, a speed assumption only) walk ONE contiguous
    // eighth of the cell-sorted colliders instead of every eighth block of all of them, so the AABB rows they share stay in that XCD's L2
    const uint32_t inCol = blockId % blocksPerColumn;
    const uint32_t dy = gp->dims[1], dz = gp->dims[2], dx = gp->dims[0];
    const uint32_t axis = sc->axisCur;
    const uint32_t numSmall = cellLower[gp->numCells];   // total of the cell histogram = this step's small colliders = the filled part of the sorted arrays
    // The host sizes the launch for the small colliders it EXPECTS (previous step's count + slack; a sharded world holds mostly dead
    // colliders, and an idle workgroup still costs its dispatch slot: 0.27 ms in an 8-tile world); if there are more, the workgroups go round again.
    const uint32_t perRound = blocksPerColumn * (kGridChunks * 256u);
    for (uint32_t roundBase = 0; roundBase < numSmall; roundBase += perRound) {
    if (roundBase) __syncthreads();
    if (threadIdx.x < 32) bhist[threadIdx.x] = 0;
    if (threadIdx.x == 32) ovfCount = 0;
    __syncthreads();
    const uint32_t base = roundBase + ((inCol & 7u) * (blocksPerColumn >> 3) + (inCol >> 3)) * (kGridChunks * 256u);
    uint32_t overlaps = 0, nh[kGridChunks];
    uint32_t runBucket = 0, runCount = 0;   // this lane's hits go to the bucket histogram in runs (a pile: one bucket -> one LDS atomic per lane, not per hit)
#pragma unroll
    for (uint32_t ch = 0; ch < kGridChunks; ++ch) {
        const uint32_t i = base + ch * 256u + threadIdx.x;
        uint64_t* mybuf = buf + (ch * 256u + threadIdx.x) * kPairBuf;
        uint32_t nhit = 0;
        uint32_t key = i < numSmall ? keys[i] : 0xFFFFFFFFu;   // sorted positions [0, numSmall) hold the small colliders
        if (key != 0xFFFFFFFFu) {
            uint32_t iz = key % dz, iy = (key / dz) % dy, ix = key / (dz * dy);
            int x = (int)ix + (col >= 2 ? 1 : 0);
            int y = (int)iy + (col == 1 ? 1 : (col >= 2 ? (int)col - 3 : 0));
            if (x < (int)dx && y >= 0 && y < (int)dy) {
                int z0 = col == 0 ? (int)iz : (int)iz - 1, z1 = (int)iz + 1;
                if (z0 < 0) z0 = 0;
                if (z1 >= (int)dz) z1 = (int)dz - 1;
                uint32_t cbase = ((uint32_t)x * dy + (uint32_t)y) * dz;
                uint32_t s = col == 0 ? i + 1u : cellLower[cbase + (uint32_t)z0];
                uint32_t e = cellLower[cbase + (uint32_t)z1 + 1u];
                float4 amn = sMin[i], amx = sMax[i];
                uint32_t ci = vals[i];
                for (uint32_t j = s; j < e; j += 4u) {
                    float4 bmn[4], bmx[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) { uint32_t jj = min(j + u, e - 1u); bmn[u] = sMin[jj]; bmx[u] = sMax[jj]; }
#pragma unroll
                    for (uint32_t u = 0; u < 4; ++u) {
                        if (j + u >= e || !aabbOverlap(amn, amx, bmn[u], bmx[u])) continue;
                        ++overlaps;
                        uint64_t pk;
                        if (!pairKey(ci, amn, amx, vals[j + u], bmn[u], bmx[u], axis, pk, inter)) continue;
                        if (nhit < kPairBuf) mybuf[nhit] = pk;
                        else {
                            const uint32_t o = atomicAdd(&ovfCount, 1u);
                            if (o < kPairOverflow) ovf[o] = pk;
                            else { uint32_t slot = atomicAdd(&sc->numPairs, 1u); if (slot < pairCap) pairKeys[slot] = pk; }   // both stagings full: rare
                        }
                        { const uint32_t bk = (uint32_t)(pk >> 58); if (bk != runBucket && runCount) { atomicAdd(&bhist[runBucket], runCount); runCount = 0; } runBucket = bk; ++runCount; }
                        ++nhit;
                    }
                }
            }
        }
        nh[ch] = min(nhit, kPairBuf);
    }
    if (runCount) atomicAdd(&bhist[runBucket], runCount);
    // block exclusive scan of the staged counts
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t ch = 0; ch < kGridChunks; ++ch) mine += nh[ch];
    uint32_t incl = mine;
    uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += v; }
    if (lane == 63) waveTotals[wv] = incl;
    for (int off = 32; off >= 1; off >>= 1) overlaps += __shfl_xor(overlaps, off, 64);
    if (lane == 0 && overlaps) atomicAdd(&bhist[31], overlaps);
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < wv; ++w) wbase += waveTotals[w];
    const uint32_t staged = waveTotals[0] + waveTotals[1] + waveTotals[2] + waveTotals[3];
    const uint32_t nOvf = min(ovfCount, kPairOverflow);
    if (threadIdx.x == 0) {
        const uint32_t total = staged + nOvf;
        blockBase = total ? atomicAdd(&sc->numPairs, total) : 0u;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nOvf; k += 256u) { const uint32_t d = blockBase + staged + k; if (d < pairCap) pairKeys[d] = ovf[k]; }
    uint32_t dst = blockBase + wbase + incl - mine;
#pragma unroll
    for (uint32_t ch = 0; ch < kGridChunks; ++ch) {
        const uint64_t* mybuf = buf + (ch * 256u + threadIdx.x) * kPairBuf;
        for (uint32_t k = 0; k < nh[ch]; ++k, ++dst) if (dst < pairCap) pairKeys[dst] = mybuf[k];
    }
    ShardCounters* shard = &sh->c[blockId & (kShards - 1u)];
    if (threadIdx.x < kNumBuckets && bhist[threadIdx.x]) atomicAdd(&shard->bucketHist[threadIdx.x], bhist[threadIdx.x]);
    if (threadIdx.x == 31 && bhist[31]) atomicAdd(&shard->numOverlaps, bhist[31]);
    }
}

__global__ __launch_bounds__(256) void k_bp_pairs_grid(uint32_t nc, uint32_t blocksPerColumn, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                       const float4* __restrict__ sMin, const float4* __restrict__ sMax, const uint32_t* __restrict__ cellLower,
                                                       const GridParams* __restrict__ gp, uint64_t* __restrict__ pairKeys, uint32_t pairCap, StepScalars* sc, Shards* sh, InterSink inter) {
    __shared__ PairLds L;
    bpPairsGridBody(L, blockIdx.x, nc, blocksPerColumn, keys, vals, sMin, sMax, cellLower, gp, pairKeys, pairCap, sc, sh, inter);
}

// (Measured and not kept, round 3: the candidate rows of a workgroup's column staged in LDS — its colliders are consecutive in the cell-sorted order, so
// the union of their candidate ranges is one contiguous span; 32 KiB for 1024 rows, every lane then walks its own range in LDS.  The kernel waits on
// L2 round trips (15 % of its cycles issue, 91 % L2 hits: profiles/r03_pmc_kernels.json), but per-lane 16-byte LDS reads at unrelated addresses
// conflict and the 48 KiB cut the occupancy from 5 to 3 workgroups per CU: 55 -> 115 us.)
// Large colliders against everything: (large l) x (all colliders), grid-strided.  Large-large pairs
// are emitted once (from the lower index).
__device__ __forceinline__ void bpPairsLargeBody(PairLds& L, const uint32_t bx, const uint32_t by, const uint32_t gx, const uint32_t gy /* this workgroup in the (candidate chunk, large-list slice) grid of the pass */,
                                                 uint32_t nc, const uint32_t* __restrict__ largeList,
                                                 const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
                                                 const uint32_t* __restrict__ vals, const float4* __restrict__ sMin, const float4* __restrict__ sMax,
                                                 const uint32_t* __restrict__ cellLower, const GridParams* __restrict__ gp,
                                                 uint64_t* __restrict__ pairKeys, uint32_t pairCap, StepScalars* sc, Shards* sh, InterSink inter) {
    // hits are staged like in k_bp_pairs_grid (kLargeBuf slots per lane, then a block-shared overflow area, then — rare — a direct append)
    // and flushed with ONE returning atomic per workgroup: the ground of a settled pile touches tens of thousands of boxes, and a
    // same-address atomic per wave per hit-iteration made this kernel 40 us
    constexpr uint32_t kLargeBuf = 4;
    static_assert(256 * kLargeBuf * 8 + 2 * 64 * 16 + 64 * 4 <= sizeof(L.buf), "the large pass's staging + its slice of the large list live in the grid pass's staging area");
    uint64_t* const buf = L.buf; uint64_t* const ovf = L.ovf; uint32_t& ovfCount = L.ovfCount; uint32_t* const waveTotals = L.waveTotals; uint32_t& blockBase = L.blockBase; uint32_t* const bhist = L.bhist;
    if (threadIdx.x < 32) bhist[threadIdx.x] = 0;
    if (threadIdx.x == 32) ovfCount = 0;
    __syncthreads();
    uint32_t nl = sc->numLarge;
    uint32_t axis = sc->axisCur;
    const uint32_t numSmall = cellLower[gp->numCells];
    uint32_t overlaps = 0, nhit = 0, runBucket = 0, runCount = 0;
    uint64_t* mybuf = buf + threadIdx.x * kLargeBuf;
    // One lane per CANDIDATE — the cell-sorted small colliders [0, numSmall) (contiguous rows) and then the large list itself, not all
    // nc colliders: the dead ones of a sharded world are in neither — which it loads once and tests against every large collider,
    // 256 of them staged in LDS at a time (a few walls and a ground in a pile; hundreds of terrain tiles under vehicles).
    constexpr uint32_t kSlice = 64;
    float4* const lMin = reinterpret_cast<float4*>(L.buf + 256 * kLargeBuf); float4* const lMax = lMin + kSlice;
    uint32_t* const lIdx = reinterpret_cast<uint32_t*>(lMax + kSlice);
    for (uint32_t q0 = bx * blockDim.x; q0 < numSmall + nl; q0 += gx * blockDim.x) {
        const uint32_t q = q0 + threadIdx.x;
        const bool have = q < numSmall + nl, small = q < numSmall;
        uint32_t j = 0; float4 bmn = make_float4(0, 0, 0, 0), bmx = bmn;
        if (have) { j = small ? vals[q] : largeList[q - numSmall]; bmn = small ? sMin[q] : aabbMin[j]; bmx = small ? sMax[q] : aabbMax[j]; }
        for (uint32_t l0 = by * kSlice; l0 < nl; l0 += gy * kSlice) {   // by: a 64-wide slice of the large list (more workgroups, shorter loops)
            __syncthreads();
            if (threadIdx.x < kSlice && l0 + threadIdx.x < nl) { const uint32_t i = largeList[l0 + threadIdx.x]; lIdx[threadIdx.x] = i; lMin[threadIdx.x] = aabbMin[i]; lMax[threadIdx.x] = aabbMax[i]; }
            __syncthreads();
            const uint32_t n = min(kSlice, nl - l0);
            if (!have) continue;
            // pass 1: which of the staged large boxes overlap mine (a cheap, convergent loop); pass 2: only those — a hit costs ~10 x a
            // test, and with the hits handled inside the first loop every lane of a wave paid for every other lane's hits
            for (uint32_t w0 = 0; w0 < n; w0 += 64u) {
                unsigned long long hitMask = 0ull;
                const uint32_t m = min(64u, n - w0);
                for (uint32_t l = 0; l < m; ++l) {
                    const bool ok = small || j > lIdx[w0 + l];          // large-large pairs once, from the lower index
                    if (ok && aabbOverlap(lMin[w0 + l], lMax[w0 + l], bmn, bmx)) hitMask |= 1ull << l;
                }
                overlaps += (uint32_t)__popcll(hitMask);
                while (hitMask) {
                    const uint32_t l = w0 + (uint32_t)__ffsll((long long)hitMask) - 1u;
                    hitMask &= hitMask - 1ull;
                    uint64_t pk = 0;
                    if (!pairKey(lIdx[l], lMin[l], lMax[l], j, bmn, bmx, axis, pk, inter)) continue;
                    { const uint32_t bk = (uint32_t)(pk >> 58); if (bk != runBucket && runCount) { atomicAdd(&bhist[runBucket], runCount); runCount = 0; } runBucket = bk; ++runCount; }
                    if (nhit < kLargeBuf) mybuf[nhit] = pk;
                    else {
                        const uint32_t o = atomicAdd(&ovfCount, 1u);
                        if (o < kPairOverflow) ovf[o] = pk;
                        else { uint32_t slot = atomicAdd(&sc->numPairs, 1u); if (slot < pairCap) pairKeys[slot] = pk; }
                    }
                    ++nhit;
                }
            }
        }
    }
    if (runCount) atomicAdd(&bhist[runBucket], runCount);
    const uint32_t mine = min(nhit, kLargeBuf);
    uint32_t incl = mine;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += v; }
    if (lane == 63) waveTotals[wv] = incl;
    for (int off = 32; off >= 1; off >>= 1) overlaps += __shfl_xor(overlaps, off, 64);
    if (lane == 0 && overlaps) atomicAdd(&bhist[31], overlaps);
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < wv; ++w) wbase += waveTotals[w];
    const uint32_t staged = waveTotals[0] + waveTotals[1] + waveTotals[2] + waveTotals[3];
    const uint32_t nOvf = min(ovfCount, kPairOverflow);
    if (threadIdx.x == 0) { const uint32_t total = staged + nOvf; blockBase = total ? atomicAdd(&sc->numPairs, total) : 0u; }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nOvf; k += 256u) { const uint32_t d = blockBase + staged + k; if (d < pairCap) pairKeys[d] = ovf[k]; }
    uint32_t dst = blockBase + wbase + incl - mine;
    for (uint32_t k = 0; k < mine; ++k, ++dst) if (dst < pairCap) pairKeys[dst] = mybuf[k];
    ShardCounters* shard = &sh->c[(bx + by) & (kShards - 1u)];
    if (threadIdx.x < kNumBuckets && bhist[threadIdx.x]) atomicAdd(&shard->bucketHist[threadIdx.x], bhist[threadIdx.x]);
    if (threadIdx.x == 31 && bhist[31]) atomicAdd(&shard->numOverlaps, bhist[31]);
}

__global__ __launch_bounds__(256) void k_bp_pairs_large(uint32_t nc, const uint32_t* __restrict__ largeList, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
                                                        const uint32_t* __restrict__ vals, const float4* __restrict__ sMin, const float4* __restrict__ sMax,
                                                        const uint32_t* __restrict__ cellLower, const GridParams* __restrict__ gp,
                                                        uint64_t* __restrict__ pairKeys, uint32_t pairCap, StepScalars* sc, Shards* sh, InterSink inter) {
    __shared__ PairLds L;
    bpPairsLargeBody(L, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, nc, largeList, aabbMin, aabbMax, vals, sMin, sMax, cellLower, gp, pairKeys, pairCap, sc, sh, inter);
}
// Both passes in ONE launch: they only meet in the pair list's append counter and the sharded sum-only counters.  The grid pass waits on L2 round trips with its
// issue slots half empty, the large pass (a ground and four walls against every box of a pile) is 12 us of launch floor, gathers and a block flush: as the first workgroups
// of the grid pass's launch it runs beside that pass instead of behind it.  largeBlocks = gx * gy rounded up to a multiple of 8 (the grid pass's workgroups keep their
// blockIdx % 8 = XCD residue); workgroups in the padding find nothing to do.
__global__ __launch_bounds__(256) void k_bp_pairs(uint32_t largeBlocks, uint32_t gx, uint32_t gy, uint32_t nc, uint32_t blocksPerColumn, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                  const float4* __restrict__ sMin, const float4* __restrict__ sMax, const uint32_t* __restrict__ cellLower, const GridParams* __restrict__ gp,
                                                  const uint32_t* __restrict__ largeList, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
                                                  uint64_t* __restrict__ pairKeys, uint32_t pairCap, StepScalars* sc, Shards* sh, InterSink inter) {
    __shared__ PairLds L;
    if (blockIdx.x < largeBlocks) {
        if (blockIdx.x >= gx * gy) return;
        bpPairsLargeBody(L, blockIdx.x % gx, blockIdx.x / gx, gx, gy, nc, largeList, aabbMin, aabbMax, vals, sMin, sMax, cellLower, gp, pairKeys, pairCap, sc, sh, inter);
    } else bpPairsGridBody(L, blockIdx.x - largeBlocks, nc, blocksPerColumn, keys, vals, sMin, sMax, cellLower, gp, pairKeys, pairCap, sc, sh, inter);
}
// Shard totals -> StepScalars (read back by the host together with numPairs).
// `pairBound`: what the launches / scans / buffers downstream are sized for.  A speculative step that found more pairs is
// invalid as a whole (the host re-runs it synchronously): mark it and make everything downstream a no-op.
__host__ __device__ __forceinline__ int gjkMode(uint32_t ta, uint32_t tb);
// Bucket partition (replaces the reference's counting sort into [6][6] type-pair buckets, collision_narrow.cpp:2397-2453,
// and the former full 64-bit key sort): pairs are grouped by bucket so narrow-phase waves are type-uniform; the order
// inside a bucket is arbitrary — every later stage is keyed by the collider pair, not by the position of the pair.
// A block ranks its 1024 keys per bucket in LDS and reserves one output range per non-empty bucket.
__host__ __device__ __forceinline__ int gjkModeOfBucket(uint32_t bucket);
// One workgroup after the pair pass: the sharded counters summed (k_pair_totals), the bucket offsets / GJK span / "partition needed"
// (formerly k_pair_ranges) and — with `partials` — the next sweep axis (formerly k_axis_final): three single-workgroup launches in one.
// The part of the pair stage that nothing in the rest of the step waits for: the centre statistics -> next sweep axis, and the NEXT step's grid (threshold,
// cell size, origin, dims) from this step's extent histogram and centre bounds.  One workgroup of 256; it used to be the tail of k_pair_finish, i.e. ~10 us
// of single-workgroup work on the step's critical path — now an extra workgroup of k_emit_manifolds runs it beside that kernel's thousands.
__device__ inline void pairFinishStats(const Shards* __restrict__ sh, StepScalars* sc, uint32_t nc, uint32_t numBlocks, const unsigned long long* __restrict__ partials,
                                       const int* __restrict__ blockBounds, GridParams* gridNext, uint32_t cellCapNext, const uint8_t* __restrict__ cbLive = nullptr /* sharded world: only these blocks' rows are not empty */) {
    const uint32_t t = threadIdx.x;
    __shared__ unsigned long long sm[4][kAxisSums];
    unsigned long long v[kAxisSums];
#pragma unroll
    for (uint32_t c = 0; c < kAxisSums; ++c) v[c] = 0ull;
    // (sharded world: first WHICH of a thread's rows are not empty — 32 independent flag loads —, then those rows: a flag load in front of every row's loads made this
    // workgroup, 32 rows per thread in an 8-tile scene, the tail of k_emit_manifolds)
    for (uint32_t b0 = t; b0 < numBlocks; b0 += 256u * 32u) {
        uint32_t rowsLive = 0xFFFFFFFFu;
        if (cbLive) { rowsLive = 0u;
#pragma unroll
            for (uint32_t k = 0; k < 32u; ++k) { const uint32_t b = b0 + 256u * k; if (b < numBlocks && cbLive[b]) rowsLive |= 1u << k; } }
        for (uint32_t k = 0; k < 32u; ++k) {
            const uint32_t b = b0 + 256u * k;
            if (b >= numBlocks) break;
            if (!((rowsLive >> k) & 1u)) continue;
#pragma unroll
            for (uint32_t c = 0; c < kAxisSums; ++c) v[c] += partials[(size_t)b * kAxisSums + c];
        }
    }
    axisWaveReduce(v, sm);
    const uint32_t lane = t & 63, wv = t >> 6;
    __syncthreads();
    if (gridNext) {   // k_bp_threshold + k_bp_grid_setup for the next step, from this step's extent histogram and centre bounds
        __shared__ uint32_t hist[256];
        __shared__ int red[4][6];
        { uint32_t h = 0; for (uint32_t k = 0; k < kShards; ++k) h += sh->extentHist[k][t]; hist[t] = h; }
        int b6[6] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000};
        for (uint32_t b0 = t; b0 < numBlocks; b0 += 256u * 32u) {
            uint32_t rowsLive = 0xFFFFFFFFu;
            if (cbLive) { rowsLive = 0u;
#pragma unroll
                for (uint32_t k = 0; k < 32u; ++k) { const uint32_t b = b0 + 256u * k; if (b < numBlocks && cbLive[b]) rowsLive |= 1u << k; } }
            for (uint32_t k = 0; k < 32u; ++k) {
                const uint32_t b = b0 + 256u * k;
                if (b >= numBlocks) break;
                if (!((rowsLive >> k) & 1u)) continue;
                for (int a = 0; a < 6; ++a) { int x = blockBounds[b * 6 + a]; b6[a] = a < 3 ? min(b6[a], x) : max(b6[a], x); }
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(b6[a], d, 64); b6[a] = a < 3 ? min(b6[a], o) : max(b6[a], o); }
        if (lane == 0) for (int a = 0; a < 6; ++a) red[wv][a] = b6[a];
        __syncthreads();
        // threshold = upper edge of the highest bin b whose bins ABOVE hold <= limit colliders while b itself would exceed it: a suffix
        // sum over the 256 bins (wave shuffles + the 4 wave totals) instead of a serial walk
        __shared__ uint32_t wsum[4];
        __shared__ float thrShared, thrShared2;
        if (t == 0) { thrShared = 0.f; thrShared2 = 0.f; }
        uint32_t suf = hist[t];                                         // inclusive suffix sum within the wave
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { uint32_t o = (uint32_t)__shfl_down((int)suf, d, 64); if (lane + d < 64u) suf += o; }
        if (lane == 0) wsum[wv] = suf;
        __syncthreads();
        for (uint32_t w = wv + 1; w < 4; ++w) suf += wsum[w];
        {
            __shared__ uint32_t liveShared;
            if (t == 0) { liveShared = suf; sc->numDead = nc - suf; }   // histogram total = live colliders (dead ones of a sharded world are not in it)
            __syncthreads();
            const uint32_t live = liveShared;
            uint32_t limit = max(16u, live / 16384u);
            const uint32_t costCap = (uint32_t)(67108864ull / (uint64_t)max(live, 1u));
            limit = max(8u, min(limit, costCap));
            const uint32_t above = suf - hist[t];                       // colliders in bins > t
            if (above <= limit && suf > limit) thrShared = extentBinUpper(t);
            // A second candidate with a much larger budget of "large" colliders: worth it only where the sizes are bimodal — a few hundred
            // terrain tiles among tens of thousands of vehicle parts (cfg5) would otherwise set the cell size, every cell then holds a whole
            // vehicle and the column scans do 64 x the tests (k_bp_pairs_grid 102 us for 22 k colliders).  The brute-force pass over the
            // large ones streams numLarge x live boxes; it stays under the same cost cap.
            const uint32_t limit2 = max(limit, min(live / 32u, costCap));
            if (above <= limit2 && suf > limit2) thrShared2 = extentBinUpper(t);
        }
        __syncthreads();
        if (t == 0) {
            float thr = thrShared;
            if (thrShared2 > 0.f && thrShared2 <= 0.5f * thr) thr = thrShared2;   // only when the cells shrink at least 2 x (8 x fewer candidates each)
            for (int w = 1; w < 4; ++w) for (int a = 0; a < 6; ++a) b6[a] = a < 3 ? min(b6[a], red[w][a]) : max(b6[a], red[w][a]);
            float cell = thr * 1.001f + 1e-6f;
            float lo[3], hi[3];
            const bool any = b6[0] != 0x7FFFFFFF;
            for (int a = 0; a < 3; ++a) { lo[a] = any ? fromOrderedInt(b6[a]) : 0.f; hi[a] = any ? fromOrderedInt(b6[3 + a]) : 0.f; }
            for (int it = 0; it < 64; ++it) {
                double cells = 1.0;
                for (int a = 0; a < 3; ++a) { uint32_t d = (uint32_t)((hi[a] - lo[a]) / cell) + 2u; gridNext->dims[a] = d; cells *= (double)d; }
                if (cells <= (double)(cellCapNext - 1)) break;
                cell *= 1.3f;
            }
            gridNext->numCells = gridNext->dims[0] * gridNext->dims[1] * gridNext->dims[2];
            sc->numCellsNext = gridNext->numCells;
            gridNext->cell = cell; gridNext->invCell = 1.f / cell;
            for (int a = 0; a < 3; ++a) gridNext->origin[a] = lo[a];
            gridNext->numLarge = 0; gridNext->largeThreshold = thr;
        }
    }
    if (t != 0) return;
    unsigned long long s9[kAxisSums];
    for (uint32_t c = 0; c < kAxisSums; ++c) { s9[c] = sm[0][c] + sm[1][c] + sm[2][c] + sm[3][c]; sc->axisSums[c] = s9[c]; }
    sc->axisNext = axisFromSums(s9, nc);   // (sharded world: from this rank's own sums — the exchange replaces it by the axis of the sums over all ranks)
}
// The counts a step's pair list ends with (wave 0 of k_pair_finish; or, fused, wave 0 of the first workgroup of k_narrow): bucket histogram summed over the counter shards, bucket offsets, the GJK span, whether the list wants partitioning, and the
// speculative step's guards.  Returns the number of pairs the step goes on with (0: the step is void).
__device__ __forceinline__ uint32_t pairFinishCounts(const uint32_t t /* lane of wave 0 */, const Shards* __restrict__ sh, StepScalars* sc, uint32_t pairBound, uint32_t allowPartition, uint32_t& partitionedOut) {
    const uint32_t found = sc->numPairs;
    bool voidStep = found > pairBound;
    uint32_t v = 0;
    if (t < kNumBuckets) { for (uint32_t k = 0; k < kShards; ++k) v += sh->c[k].bucketHist[t]; sc->bucketHist[t] = v; }
    if (t == 31) { uint32_t o = 0; for (uint32_t k = 0; k < kShards; ++k) o += sh->c[k].numOverlaps; sc->numOverlaps = o; }
    uint32_t off = 0, nonEmpty = 0, lo = 0xFFFFFFFFu, hi = 0, largest = 0;
    for (uint32_t bk = 0; bk < kNumBuckets; ++bk) {   // every lane walks the buckets (the counts come over by shuffle), lane 0 writes
        const uint32_t n = (uint32_t)__shfl((int)v, (int)bk, 64);
        if (t == 0) sc->bucketOffset[bk] = off;
        if (n) { ++nonEmpty; largest = max(largest, n); if (gjkModeOfBucket(bk) >= 0) { lo = min(lo, off); hi = max(hi, off + n); } }
        off += n;
    }
    // the partition exists to make narrow-phase waves type-uniform and to give the GJK kernel its span; when nearly every pair is of
    // ONE type (a box pile: box-box, plus the boxes on the ground) it only costs (a pass over the keys + a reservation per workgroup):
    // partition if a GJK bucket is populated or more than an eighth of the pairs lies outside the largest bucket
    uint32_t want = (nonEmpty > 1u && (hi > lo || (off - largest) * 8u > off)) ? 1u : 0u;
    // a speculative step that left k_pair_partition out (its predecessor did not partition) must not go on with a partitioned list that was never
    // written: everything downstream becomes a no-op, the step is invalid as a whole and is re-run
    if (want && !allowPartition) { voidStep = true; want = 0u; }
    if (t == 0) {
        sc->gjkLo = (hi > lo && !voidStep) ? lo : 0u; sc->gjkHi = (hi > lo && !voidStep) ? hi : 0u;
        sc->partitioned = want;
        if (voidStep) { sc->specOverflow = 1u; sc->numPairsFound = found; sc->numPairs = 0u; }
    }
    partitionedOut = want;
    return voidStep ? 0u : found;
}
__global__ __launch_bounds__(256) void k_pair_finish(const Shards* __restrict__ sh, StepScalars* sc, uint32_t pairBound, uint32_t nc, uint32_t numBlocks, const unsigned long long* __restrict__ partials,
                                                     const int* __restrict__ blockBounds, GridParams* gridNext /* the NEXT step's grid (null: not wanted) */, uint32_t cellCapNext,
                                                     uint32_t allowPartition /* 0: k_pair_partition is not going to run in this step */,
                                                     uint32_t doStats /* 0: an extra workgroup of k_emit_manifolds runs pairFinishStats */) {
    const uint32_t t = threadIdx.x;
    if (t < 64u) { uint32_t part; (void)pairFinishCounts(t, sh, sc, pairBound, allowPartition, part); }
    if (!partials || !doStats) return;
    pairFinishStats(sh, sc, nc, numBlocks, partials, blockBounds, gridNext, cellCapNext);
}
__global__ __launch_bounds__(256) void k_pair_partition(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, StepScalars* sc) {
    __shared__ uint32_t cnt[32], base[32];
    const uint32_t n = sc->numPairs;
    if (!sc->partitioned || blockIdx.x * 1024u >= n) return;
    if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint64_t key[4]; uint32_t rank[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t p = blockIdx.x * 1024u + (uint32_t)k * 256u + threadIdx.x;
        key[k] = p < n ? in[p] : ~0ull;
        rank[k] = p < n ? atomicAdd(&cnt[(uint32_t)(key[k] >> 58)], 1u) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < kNumBuckets) base[threadIdx.x] = cnt[threadIdx.x] ? sc->bucketOffset[threadIdx.x] + atomicAdd(&sc->bucketCursor[threadIdx.x], cnt[threadIdx.x]) : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t p = blockIdx.x * 1024u + (uint32_t)k * 256u + threadIdx.x;
        if (p < n) out[base[(uint32_t)(key[k] >> 58)] + rank[k]] = key[k];
    }
}

// ------------------------------------------------------------------------------------------------
// Narrow phase: one lane per (bucket-sorted) collision pair; a wave is type-uniform except at bucket
// boundaries.  Writes a fixed 4-slot manifold per pair; compaction happens by prefix sums.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Shape loadShape(const float4* __restrict__ wShape, uint32_t k, uint32_t type) {
    Shape s; s.type = (int)type; s.radius = 0.f; s.hull = 0;
    float4 r0 = wShape[3 * k], r1 = wShape[3 * k + 1];
    s.a = xyz(r0); s.b = xyz(r1);
    if (type <= T_CYLINDER) s.radius = r0.w;
    if (type >= T_OBB) s.rot = toQ(wShape[3 * k + 2]);
    if (type == T_HULL) s.hull = __float_as_uint(r0.w);
    return s;
}

struct HullSet { const float4* verts; const uint32_t* ranges; };  // vertex pool + [first,count] per geometry

// Buckets that need GJK/EPA (and ~12 KB of per-lane scratch) run in k_narrow_gjk (gjk.hpp); -1 = primitive bucket.
// mode: 0 plain GJK+EPA single contact, 1 segment shape vs AABB, 2 segment shape vs OBB, 3 cylinder vs cylinder
__host__ __device__ __forceinline__ int gjkMode(uint32_t ta, uint32_t tb) {
    if (tb == T_HULL) return 0;
    if (ta == T_CAPSULE && tb == T_AABB) return 1;
    if (ta == T_CAPSULE && tb == T_OBB) return 2;
    if (ta == T_CYLINDER && tb == T_CYLINDER) return 3;
    if (ta == T_CYLINDER && tb == T_AABB) return 1;
    if (ta == T_CYLINDER && tb == T_OBB) return 2;
    return -1;
}

__host__ __device__ __forceinline__ int gjkModeOfBucket(uint32_t bucket) {
    uint32_t ta = 0, rem = bucket;
    while (rem >= 6u - ta) { rem -= 6u - ta; ++ta; }
    return gjkMode(ta, ta + rem);
}

__device__ inline bool intersectPair(const Shape& a, const Shape& b, const HullSet& hs, Manifold& out) {
    switch (a.type) {
        case T_SPHERE:
            switch (b.type) {
                case T_SPHERE: return sphereSphere(a.a, a.radius, b.a, b.radius, out);
                case T_CAPSULE: return sphereSphere(a.a, a.radius, closestOnSegment(a.a, b.a, b.b), b.radius, out);
                case T_CYLINDER: return sphereCylinder(a.a, a.radius, b.a, b.b, b.radius, out);
                case T_AABB: return sphereAABB(a.a, a.radius, b.a, b.b, out);
                case T_OBB: return sphereOBB(a.a, a.radius, b.rot, b.a, b.b, out);
                default: return false;  // GJK bucket: k_narrow_gjk
            }
        case T_CAPSULE:
            switch (b.type) {
                case T_CAPSULE: return capsuleVsSegmentShape(a, b, false, out);
                case T_CYLINDER: return capsuleVsSegmentShape(a, b, true, out);
                case T_AABB: return false;  // GJK bucket: k_narrow_gjk
                case T_OBB: return false;  // GJK bucket: k_narrow_gjk
                default: return false;  // GJK bucket: k_narrow_gjk
            }
        case T_CYLINDER:
            switch (b.type) {
                case T_CYLINDER: return false;  // GJK bucket: k_narrow_gjk
                case T_AABB: return false;  // GJK bucket: k_narrow_gjk
                case T_OBB: return false;  // GJK bucket: k_narrow_gjk
                default: return false;  // GJK bucket: k_narrow_gjk
            }
        case T_AABB:
            switch (b.type) {
                case T_AABB: return aabbAABB(a.a, a.b, b.a, b.b, out);
                case T_OBB: return false;  // box pair: SAT in k_narrow, contacts in k_narrow_clip
                default: return false;  // GJK bucket: k_narrow_gjk
            }
        case T_OBB:
            return false;  // OBB-OBB: SAT in k_narrow, contacts in k_narrow_clip; OBB-hull: k_narrow_gjk
        default:
            return false;  // GJK bucket: k_narrow_gjk
    }
}

// The pair list is read from `pairsA` (arrival order) or `pairsB` (bucket-partitioned) as StepScalars::partitioned says;
// lanes in [numPairs, scanLen) zero their scan input so the host can size the launch and the scan from an upper bound.
//
// Box-box pairs (OBB-OBB, AABB-OBB: the bulk of a box pile) run in two kernels.  k_narrow, every lane: the 15-axis SAT
// (cheap; ~57 % of the AABB-overlapping pairs of a settled pile are separated, and without clip polygons the kernel needs
// no LDS).  The lanes that overlap append (pair, SAT result) to one of 16 global queues (one reservation per workgroup,
// sharded so the reservations do not serialise on one word).  k_narrow_clip then runs the expensive half — incident-face
// clipping and the 4-point reduction, polygons in LDS — over the queues: every clipping wave has all 64 lanes busy
// instead of ~43 % of them (the clipping was 124 of the fused kernel's 170 us).  All other pair types finish in k_narrow.
struct BoxHit { uint32_t pair; float nx, ny, nz; uint32_t flags; };
__device__ __forceinline__ void writeManifold(uint32_t p, bool hit, const Manifold& m, uint64_t* __restrict__ npPacked, float4* __restrict__ npNormal,
                                              float4* __restrict__ npPoints) {
    uint32_t cnt = hit ? m.count : 0u;
    npPacked[p] = cnt ? ((1ull << 32) | (uint64_t)cnt) : 0ull;   // (manifold flag, contact count): one 64-bit scan compacts both
    if (cnt) {
        npNormal[p] = f4(m.n, 0.f);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) if (k < cnt) npPoints[4 * p + k] = f4(m.p[k], m.d[k]);   // (static indices: the manifold stays in registers, not in scratch)
    }
}
__device__ __forceinline__ void boxPairShapes(const float4* __restrict__ wShape, uint32_t a, uint32_t b, uint32_t ta,
                                              Q4& arot, V3& acen, V3& arad, Q4& brot, V3& bcen, V3& brad) {
    Shape sa = loadShape(wShape, a, ta), sb = loadShape(wShape, b, T_OBB);
    if (ta == T_AABB) { arot = Q4(0.f, 0.f, 0.f, 1.f); acen = (sa.a + sa.b) * 0.5f; arad = (sa.b - sa.a) * 0.5f; }
    else { arot = sa.rot; acen = sa.a; arad = sa.b; }
    brot = sb.rot; bcen = sb.a; brad = sb.b;
}
constexpr uint32_t kBoxQueues = 16;
// A word every workgroup reads, through the scalar cache — explicitly: after the stores of pairFinishCounts (other path, same kernel) the compiler no longer proves the
// word unclobbered and reads it with a vector load, and 3 000 workgroups' vector loads of one line that the same workgroups hit with atomics (boxHitCount) queue up behind
// those atomics in the L2: k_narrow 33 -> 99 us (measured, round 5).
__device__ __forceinline__ uint32_t scalarLoadU32(const uint32_t* p /* uniform */) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(256) void k_narrow(uint32_t scanLen, uint32_t queueRegion, StepScalars* sc, const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB,
                                                const float4* __restrict__ wShape,
                                                HullSet hs, uint64_t* __restrict__ npPacked, float4* __restrict__ npNormal,
                                                float4* __restrict__ npPoints, BoxHit* __restrict__ boxQueue,
                                                ulonglong2* __restrict__ clearTab /* the NEXT step's colour history, cleared here on the side (was a launch of its own) */, uint32_t clearSlots,
                                                const Shards* __restrict__ finishShards /* non-null: no k_pair_finish ran (a speculative step without k_pair_partition) — wave 0 of every workgroup
                                                                                           derives the pair list's final counts itself (pairFinishCounts), workgroup 0 writes them */,
                                                uint32_t finishBound, Shards* __restrict__ queueShards) {
    __shared__ BoxHit hits[256];
    __shared__ uint32_t numHits, queueBase, sNumPairs, sPartitioned;
    if (threadIdx.x == 0) numHits = 0;
    if (threadIdx.x < 64u) {   // (both modes leave the two words in LDS: a select between an LDS and a global ADDRESS compiles to a flat load, and that doubled this kernel's time)
        uint32_t n, part = 0u;
        if (finishShards) {
            // workgroup 0 does k_pair_finish's work and writes its results; the others only need the count, guarded like there.  (Should workgroup 0 find that the list
            // wants partitioning, the step is void anyway: the others walking the unpartitioned list in the meantime read valid memory and their output is discarded.)
            if (blockIdx.x == 0u) n = pairFinishCounts(threadIdx.x, finishShards, sc, finishBound, 0u, part);
            else { n = sc->numPairs; if (n > finishBound) n = 0u; }
        } else { n = scalarLoadU32(&sc->numPairs); part = scalarLoadU32(&sc->partitioned); }
        if (threadIdx.x == 0) { sNumPairs = n; sPartitioned = part; }
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < clearSlots; i += gridDim.x * blockDim.x) clearTab[i] = make_ulonglong2(0ull, 0ull);
    __syncthreads();
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t numPairs = sNumPairs;
    const uint64_t* __restrict__ pairKeys = sPartitioned ? pairsB : pairsA;
    bool boxHit = false; BoxHit mineHit{};
    if (p >= numPairs) { if (p < scanLen) npPacked[p] = 0ull; }
    else {
        uint64_t key = pairKeys[p];
        uint32_t bucket = (uint32_t)(key >> 58), a = (uint32_t)((key >> 29) & 0x1FFFFFFFu), b = (uint32_t)(key & 0x1FFFFFFFu);
        uint32_t ta = 0, rem = bucket;   // bucket -> (ta, tb)
        while (rem >= 6u - ta) { rem -= 6u - ta; ++ta; }
        uint32_t tb = ta + rem;
        if (gjkMode(ta, tb) >= 0) { /* handled by k_narrow_gjk */ }
        else if (tb == T_OBB && (ta == T_OBB || ta == T_AABB)) {
            Q4 arot, brot; V3 acen, arad, bcen, brad;
            boxPairShapes(wShape, a, b, ta, arot, acen, arad, brot, bcen, brad);
            ObbSat res;
            if (obbSat(arot, acen, arad, brot, bcen, brad, res)) { boxHit = true; mineHit = BoxHit{p, res.normal.x, res.normal.y, res.normal.z, (res.faceHit ? 1u : 0u) | (res.bFace ? 2u : 0u)}; }
            else npPacked[p] = 0ull;
        } else {
            Shape sa = loadShape(wShape, a, ta), sb = loadShape(wShape, b, tb);
            Manifold m; m.count = 0;
            bool hit = intersectPair(sa, sb, hs, m);
            writeManifold(p, hit, m, npPacked, npNormal, npPoints);
        }
    }
    {   // queue slots of the SAT hits: one LDS atomic per wave (ballot + popcount), not one per hitting lane
        const unsigned long long hm = __ballot(boxHit);
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (hm && lane == (uint32_t)__ffsll((long long)hm) - 1u) base = atomicAdd(&numHits, (uint32_t)__popcll(hm));
        base = (uint32_t)__shfl((int)base, hm ? __ffsll((long long)hm) - 1 : 0, 64);
        if (boxHit) hits[base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = mineHit;
    }
    __syncthreads();
    const uint32_t q = blockIdx.x & (kBoxQueues - 1u);
    if (threadIdx.x == 0 && numHits) queueBase = atomicAdd(&queueShards->c[q].boxHits, numHits);
    __syncthreads();
    if (threadIdx.x < numHits) boxQueue[(size_t)q * queueRegion + queueBase + threadIdx.x] = hits[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_narrow_clip(uint32_t queueRegion, const StepScalars* __restrict__ sc, const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB,
                                                     const float4* __restrict__ wShape, const BoxHit* __restrict__ boxQueue,
                                                     uint64_t* __restrict__ npPacked, float4* __restrict__ npNormal, float4* __restrict__ npPoints, const Shards* __restrict__ queueShards,
                                                     // non-null: workgroup 0 runs pairFinishStats (centre statistics -> next sweep axis, the NEXT step's grid) beside the clipping — this kernel computes,
                                                     // the statistics are a dozen dependent rounds of loads: beside k_emit_manifolds' atomics every round took several microseconds and that one
                                                     // workgroup became the kernel's tail in a sharded world (8 192 partial rows: 74 instead of 57 us)
                                                     const Shards* __restrict__ statsShards, StepScalars* statsSc, uint32_t statsNc, uint32_t statsBlocks, const unsigned long long* __restrict__ statsPartials,
                                                     const int* __restrict__ statsBounds, GridParams* statsGridNext, uint32_t statsCellCap, const uint8_t* __restrict__ statsCbLive) {
    if (statsShards && blockIdx.x == 0u) { pairFinishStats(statsShards, statsSc, statsNc, statsBlocks, statsPartials, statsBounds, statsGridNext, statsCellCap, statsCbLive); return; }
#ifdef MI_CLIP_PINGPONG
    __shared__ float4 polyMem[2 * kLdsPolyVerts * kLdsPolyStride];   // 64 KiB: two clip polygons per lane, [vertex][lane]
#else
    __shared__ float4 polyMem[kLdsPolyVerts * kLdsPolyStride];       // 32 KiB: ONE clip polygon per lane, [vertex][lane], clipped in place (narrow.hpp clipPolygonLds)
#endif
    const uint32_t t = (blockIdx.x - (statsShards ? 1u : 0u)) * blockDim.x + threadIdx.x;
    const uint32_t q = t / queueRegion, idx = t % queueRegion;       // queueRegion is a multiple of 256: a workgroup never straddles queues
    if (q >= kBoxQueues || idx >= queueShards->c[q].boxHits) return;
    const uint64_t* __restrict__ pairKeys = sc->partitioned ? pairsB : pairsA;
    BoxHit h = boxQueue[(size_t)q * queueRegion + idx];
    uint64_t key = pairKeys[h.pair];
    uint32_t bucket = (uint32_t)(key >> 58), a = (uint32_t)((key >> 29) & 0x1FFFFFFFu), b = (uint32_t)(key & 0x1FFFFFFFu);
    uint32_t ta = bucket == bucketOf(T_AABB, T_OBB) ? (uint32_t)T_AABB : (uint32_t)T_OBB;
    Q4 arot, brot; V3 acen, arad, bcen, brad;
    boxPairShapes(wShape, a, b, ta, arot, acen, arad, brot, bcen, brad);
    ObbSat res; res.normal = V3(h.nx, h.ny, h.nz); res.faceHit = (h.flags & 1u) != 0u; res.bFace = (h.flags & 2u) != 0u;
    Manifold m; m.count = 0;
#ifdef MI_CLIP_PINGPONG
    LdsPoly polyA{polyMem + threadIdx.x, 0u}, polyB{polyMem + kLdsPolyVerts * kLdsPolyStride + threadIdx.x, 0u};
    bool hit = obbContacts(arot, acen, arad, brot, bcen, brad, res, polyA, polyB, m);
#else
    LdsPoly poly{polyMem + threadIdx.x, 0u};
    bool hit = obbContactsLds(arot, acen, arad, brot, bcen, brad, res, poly, m);
#endif
    writeManifold(h.pair, hit, m, npPacked, npNormal, npPoints);
}

// Colouring priority of a manifold: a bijection on 52 bits of its oriented collider pair (same function in the oracle),
// so priorities are unique and do not depend on where the manifold sits in memory.
__device__ __forceinline__ uint64_t pairPriority(uint32_t a, uint32_t b) {
    const uint64_t M52 = (1ull << 52) - 1ull;
    uint64_t x = ((uint64_t)a << kIndexBits) | (uint64_t)b;
    x ^= x >> 25; x = (x * 0x9E3779B97F4A7ull) & M52;
    x ^= x >> 27; x = (x * 0xC2B2AE3D27D4Full) & M52;
    x ^= x >> 23;
    return x;
}

// Colour history: open-addressing table (linear probing, load <= 0.5) from the oriented collider pair of every manifold
// of the previous step to its colour.  A manifold that persists keeps its colour (still conflict-free: the manifolds it
// shared a body with kept theirs or vanished), so the Jones-Plassmann rounds only have to colour the NEW manifolds of a
// step — a few percent of them once a pile has settled.  Stored key = (A << 26 | B) + 1 (0 = empty slot).
// keyed by collider CREATION indices (world index = nc - 1 - creation index), so the history survives colliders being added
// The second index of a heightmap contact is virtual (kHeightmapVirtualBase + j, above every real collider index): it is its
// own "creation index".
constexpr uint32_t kHeightmapVirtualBase = (1u << kIndexBits) - 256u;
__device__ __forceinline__ uint64_t historyKey(uint32_t nc, uint32_t worldA, uint32_t worldB) {
    return (((uint64_t)(nc - 1u - worldA) << kIndexBits) | (uint64_t)(worldB >= kHeightmapVirtualBase ? worldB : nc - 1u - worldB)) + 1ull;
}
__device__ __forceinline__ uint32_t tableSlot(uint64_t key, uint32_t mask) {
    uint64_t x = key * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(x >> 40) & mask;
}
// One slot = one 16-byte row (key, colour): a probe touches ONE sector — key and colour used to live in two arrays, two random sectors per probe and
// two more per insert, and k_emit_manifolds is bound by exactly those.
//
// POSITION-STABLE entries (round 6).  A manifold that keeps its colour keeps its SLOT: k_emit_manifolds copies its entry into the next step's table at the index it found
// it at in the previous step's table — one plain 16-byte store, where a fresh insertion is a compare-and-swap on a cold line and a dependent store (measured with the
// knock-out harness: 22 of the kernel's 58 us).  Kept entries have distinct slots, and the new manifolds are entered afterwards (k_schedule_finish)
// by compare-and-swap into the first empty slot from their hash, so a table is a valid open-addressing table with ONE difference: an entry's probe
// chain may have holes where its old neighbours vanished.  A lookup therefore does not stop at an empty slot; it probes the `maxDisp + 1` slots from the hash, maxDisp =
// the largest displacement any insertion has ever had (one device word, only ever raised; zeroed when the history is dropped).  A hit ends at the first match as before
// (displacement 0-1 nearly always); a miss — a NEW manifold, a few per cent of a settled pile's — costs maxDisp + 1 contiguous 16-byte probes (a few cache lines).
// When the two tables differ in size (the manifold count crossed a power of two, a synchronous re-run) positions do not carry over and every entry is inserted afresh.
struct alignas(16) HistSlot { unsigned long long key; unsigned long long val; };
struct HistHit { uint32_t colour, slot; };
__device__ __forceinline__ HistHit tableFind(const HistSlot* __restrict__ tab, uint32_t mask, uint64_t key, uint32_t maxDisp) {
    uint32_t s = tableSlot(key, mask);
    for (uint32_t n = 0; n <= maxDisp && n <= mask; ++n, s = (s + 1u) & mask) {
        const ulonglong2 e = *reinterpret_cast<const ulonglong2*>(tab + s);
        if (e.x == key) return HistHit{(uint32_t)e.y, s};
    }
    return HistHit{kUncolored, 0u};
}
__device__ __forceinline__ uint32_t tableLookup(const HistSlot* __restrict__ tab, uint32_t mask, uint64_t key, uint32_t maxDisp) { return tableFind(tab, mask, key, maxDisp).colour; }
__device__ __forceinline__ void tableInsert(HistSlot* __restrict__ tab, uint32_t mask, uint64_t key, uint32_t val, uint32_t* __restrict__ maxDisp) {
    for (uint32_t s = tableSlot(key, mask), n = 0; n <= mask; s = (s + 1u) & mask, ++n) {
        const unsigned long long old = atomicCAS(&tab[s].key, 0ull, (unsigned long long)key);
        if (old == 0ull || old == key) {   // (the colour is read in the NEXT step only; the same key again — a schedule built twice in a synchronous step — overwrites)
            tab[s].val = val;
            if (n > __hip_atomic_load(maxDisp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxDisp, n);
            return;
        }
    }
}
// Collision events (handleCollisionCallbacks, src/physics/physics.cpp:1041-1178), device half.  A manifold whose oriented
// collider pair is not in the previous step's history table begins (k_emit_manifolds flags it); a pair of the previous table
// that is not in this step's table ended.  Begin records carry the mean contact point / normal and the relative point
// velocity from the solver-side body state after force integration (rbGlobal).  Appends are wave-aggregated.
struct DeviceEvent { uint32_t type, colliderA, colliderB, pad; float point[3]; float normal[3]; float relVel[3]; };
__device__ __forceinline__ uint32_t waveAppendSlot(bool want, uint32_t* counter) {
    unsigned long long mask = __ballot(want);
    uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)mask) - 1u, base = 0;
    if (!mask) return 0u;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader, 64);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
__global__ __launch_bounds__(256) void k_events_begin(uint32_t nc, uint32_t cap, StepScalars* sc, const uint8_t* __restrict__ isNew, const uint32_t* __restrict__ manPair,
                                                      const uint2* __restrict__ manBodies, const uint2* __restrict__ manInfo,
                                                      const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB,
                                                      const float4* __restrict__ npNormal, const float4* __restrict__ npPoints,
                                                      const float4* __restrict__ gPos, const float4* __restrict__ gVel, DeviceEvent* __restrict__ events) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    bool want = m < sc->numManifolds && isNew[m] != 0;
    uint32_t slot = waveAppendSlot(want, &sc->numEvents);
    if (!want) return;
    if (slot >= cap) { sc->specOverflow = 1u; return; }
    uint32_t p = manPair[m], n = manInfo[m].x & 7u;
    uint64_t pk = (sc->partitioned ? pairsB : pairsA)[p];
    uint2 bodies = manBodies[m];
    float norm = 1.f / (float)n;
    V3 point(0.f), normal(0.f), nrm = xyz(npNormal[p]);
    for (uint32_t i = 0; i < n; ++i) { point = point + xyz(npPoints[4 * p + i]); normal = normal + nrm; }
    point = point * norm; normal = normal * norm;
    V3 vA = xyz(gVel[2 * bodies.x]), wA = xyz(gVel[2 * bodies.x + 1]), vB = xyz(gVel[2 * bodies.y]), wB = xyz(gVel[2 * bodies.y + 1]);
    V3 velA = vA + cross(wA, point - xyz(gPos[bodies.x])), velB = vB + cross(wB, point - xyz(gPos[bodies.y]));
    V3 rel = velB - velA;
    DeviceEvent e; e.type = 0u; e.colliderA = nc - 1u - (uint32_t)((pk >> 29) & 0x1FFFFFFFull); e.colliderB = nc - 1u - (uint32_t)(pk & 0x1FFFFFFFull); e.pad = 0u;
    e.point[0] = point.x; e.point[1] = point.y; e.point[2] = point.z; e.normal[0] = normal.x; e.normal[1] = normal.y; e.normal[2] = normal.z;
    e.relVel[0] = rel.x; e.relVel[1] = rel.y; e.relVel[2] = rel.z;
    events[slot] = e;
}
__global__ __launch_bounds__(256) void k_events_end(uint32_t cap, StepScalars* sc, const HistSlot* __restrict__ prevTab, uint32_t prevMask,
                                                    const HistSlot* __restrict__ curTab, uint32_t curMask,
                                                    DeviceEvent* __restrict__ events, const uint32_t* __restrict__ histDisp) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long key = s <= prevMask ? prevTab[s].key : 0ull;
    bool want = key != 0ull && ((key - 1ull) & ((1ull << kIndexBits) - 1ull)) < kHeightmapVirtualBase;   // heightmap contacts raise no events
    if (want && prevMask == curMask && curTab[s].key == key) want = false;                                  // it kept its colour, hence its slot
    want = want && tableLookup(curTab, curMask, key, *histDisp) == kUncolored;
    uint32_t slot = waveAppendSlot(want, &sc->numEvents);
    if (!want) return;
    if (slot >= cap) { sc->specOverflow = 1u; return; }
    DeviceEvent e{};
    e.type = 1u; e.colliderA = (uint32_t)((key - 1ull) >> kIndexBits); e.colliderB = (uint32_t)((key - 1ull) & ((1ull << kIndexBits) - 1ull));
    events[slot] = e;
}

#ifdef MI_DBG_KNOCKOUT
// development (knock-out harness, tools/gpu_knockout.sh).  Bits 0-2: k_contact_solve_persist (see there).  Bits 8-12: k_emit_manifolds launched a first time with its
// read-modify-write targets redirected to scratch and parts removed: 8 no bodyUsed atomics, 9 no history insert, 10 no history probe, 11 no round-0 proposals, 12 no material gathers.
__device__ uint32_t g_dbgKnock = 0u;
#define MI_EMIT_KNOCK(bit) ((g_dbgKnock >> (bit)) & 1u)
#else
#define MI_EMIT_KNOCK(bit) 0u
#endif
constexpr uint32_t kSpatialKeys = 4096;   // levels of the manifolds' spatial counting sort (k_manifold_keys / k_manifold_place below)
// After the scans: manifold m <- pair p (count > 0).  colWork = (bodyA | dynA << 31, bodyB | dynB << 31, priority lo, hi):
// everything a colouring round needs in one 16-byte row.
__global__ __launch_bounds__(256) void k_emit_manifolds(uint32_t nc, uint32_t nb, const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB, const uint64_t* __restrict__ npPacked,
                                                        const uint64_t* __restrict__ npScan,
                                                        const float4* __restrict__ cEmit /* per collider, static between uploads: (restitution, friction, body index | nb for a static collider, 1 if that body is dynamic) —
                                                                                            ONE 16-byte gather per side instead of material + world box + body (three sectors, the last one dependent) */,
                                                        uint32_t* __restrict__ manPair, uint2* __restrict__ manBodies, uint2* __restrict__ manInfo,
                                                        uint4* __restrict__ colWork, uint32_t* __restrict__ color,
                                                        const HistSlot* __restrict__ prevTab, uint32_t prevMask,
                                                        unsigned long long* __restrict__ bodyUsed, uint8_t* __restrict__ isNew, StepScalars* sc,
                                                        float2 terrainMaterial /* (restitution, friction) of the heightmap */,
                                                        HistSlot* __restrict__ nextTab, uint32_t nextMask, uint8_t* __restrict__ manKept, uint32_t* __restrict__ histDisp /* the history's probe bound (tableFind) */,
                                                        const Shards* __restrict__ statsShards /* non-null: workgroup 0 runs pairFinishStats instead */, uint32_t statsBlocks,
                                                        const unsigned long long* __restrict__ statsPartials, const int* __restrict__ statsBounds, GridParams* statsGridNext, uint32_t statsCellCap, const uint8_t* __restrict__ statsCbLive,
                                                        const uint32_t* __restrict__ seamId /* exact seam: per body, the tile border it is shared across (0 = none); or null */,
                                                        unsigned long long* __restrict__ topRound1 /* non-null: colouring round 0 happens right here — an uncoloured manifold proposes itself on its bodies
                                                                                                       for round 1 (k_color_round's "lost" branch at round 0: every uncoloured manifold loses round 0) */,
                                                        uint32_t* __restrict__ roundFlags) {
    // (workgroup 0, not the last one: dispatched first, it runs beside all the others; as the last one its ~4 us — 12 us over the 8 192 partial rows
    // of a 2 M-collider sharded scene — started when the kernel was all but over and became its tail)
    if (statsShards && blockIdx.x == 0u) { pairFinishStats(statsShards, sc, nc, statsBlocks, statsPartials, statsBounds, statsGridNext, statsCellCap, statsCbLive); return; }
    uint32_t p = (blockIdx.x - (statsShards ? 1u : 0u)) * blockDim.x + threadIdx.x;
    const uint32_t numPairs = sc->numPairs;
    if (p >= numPairs) return;
    const uint64_t* __restrict__ pairKeys = sc->partitioned ? pairsB : pairsA;
    uint32_t cnt = (uint32_t)(npPacked[p] & 0xFFFFFFFFull);
    uint64_t sc64 = npScan[p];
    uint32_t m = (uint32_t)(sc64 >> 32), conOff = (uint32_t)(sc64 & 0xFFFFFFFFull);
    if (p == numPairs - 1) { sc->numManifolds = m + (cnt ? 1u : 0u); sc->numContacts = conOff + cnt; }
    if (!cnt) return;
    uint64_t key = pairKeys[p];
    uint32_t a = (uint32_t)((key >> 29) & 0x1FFFFFFFu), b = (uint32_t)(key & 0x1FFFFFFFu);
    const bool terrain = b >= kHeightmapVirtualBase;   // heightmap contact: body B = the static dummy, material of the heightmap
    float4 ma = cEmit[a], mb = terrain ? make_float4(terrainMaterial.x, terrainMaterial.y, __uint_as_float(nb), 0.f) : cEmit[b];
    if (MI_EMIT_KNOCK(12)) { ma = make_float4(0.5f, 0.5f, __uint_as_float(a % nb), __uint_as_float(1u)); mb = make_float4(0.5f, 0.5f, __uint_as_float(b % nb), __uint_as_float(1u)); }
    float friction = clamp01(sqrtf(ma.y * mb.y));                      // collision_narrow.cpp:2232-2238
    float restitution = clamp01(fmaxr(ma.x, mb.x));
    uint32_t fr = ((uint32_t)(friction * 0xFFFF) << 16) | (uint32_t)(restitution * 0xFFFF);
    uint32_t bA = __float_as_uint(ma.z), bB = __float_as_uint(mb.z);
    manPair[m] = p;
    manBodies[m] = make_uint2(bA, bB);
    manInfo[m] = make_uint2(cnt | (conOff << 3) | (terrain ? 0x80000000u : 0u), fr);   // bit 31: a terrain manifold — contact k and ITS normal live in pair record p + k (heightmap.hpp, HmOut::put)
    if (const unsigned long long tm = __ballot(terrain); tm != 0ull && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)tm) - 1u) atomicAdd(&sc->numHmManifolds, (uint32_t)__popcll(tm));   // (part of the wave has returned: count by ballot)
    uint32_t dynA = __float_as_uint(ma.w) ? 0x80000000u : 0u;
    uint32_t dynB = __float_as_uint(mb.w) ? 0x80000000u : 0u;
    uint64_t prio = pairPriority(a, b);
    // exact seam: a SEAM manifold (all of its dynamic bodies are shared across the same tile border) takes its colour from [0, kSeamColors), any other
    // one from the colours behind them; bit 30 of the first word tells the colouring rounds which
    uint32_t seam = 0u;
    if (seamId) {
        const uint32_t idA = dynA ? seamId[bA] : 0u, idB = dynB ? seamId[bB] : 0u;
        seam = ((dynA || dynB) && (!dynA || idA) && (!dynB || idB) && (!(dynA && dynB) || idA == idB)) ? 0x40000000u : 0u;
    }
    colWork[m] = make_uint4(bA | dynA | seam, bB | dynB, (uint32_t)prio, (uint32_t)(prio >> 32));
    // a manifold of the previous step keeps its colour (colour 64 = overflow is re-coloured)
    const uint64_t hk = historyKey(nc, a, b);
    HistHit hit{kUncolored, 0u};
    if (prevTab && !MI_EMIT_KNOCK(10)) hit = tableFind(prevTab, prevMask, hk, *histDisp);
    uint32_t c = hit.colour;
    const bool found = c != kUncolored;
    if (MI_EMIT_KNOCK(10)) c = (uint32_t)(prio & 7u);
    if (isNew) isNew[m] = (c == kUncolored && !terrain) ? 1u : 0u;   // not in the previous step's collision list: collision-begin event
    if (seamId && c < kOverflowColor && (c < kSeamColors) != (seam != 0u)) c = kOverflowColor;   // it changed class: re-coloured
    if (c < kOverflowColor) {
        if (dynA && !MI_EMIT_KNOCK(8)) atomicOr(&bodyUsed[bA], 1ull << c);
        if (dynB && !MI_EMIT_KNOCK(8)) atomicOr(&bodyUsed[bB], 1ull << c);
        // its colour is final: it enters the NEXT step's history right here (k_schedule_finish then only has the few new manifolds left)
        // (its old slot when the two tables have one size: kept entries have distinct slots, the new manifolds are entered after this kernel)
        if (MI_EMIT_KNOCK(9)) {}
        else if (found && prevMask == nextMask) { ulonglong2 e; e.x = hk; e.y = c; *reinterpret_cast<ulonglong2*>(nextTab + hit.slot) = e; }
        else tableInsert(nextTab, nextMask, hk, c, histDisp);
        manKept[m] = 1u;
    } else {
        c = kUncolored; manKept[m] = 0u;
        if (topRound1 && !MI_EMIT_KNOCK(11)) {   // round 0 of the colouring (one launch less: the host starts its rounds at 1)
            const unsigned long long key1 = (1ull << 52) | (unsigned long long)prio;
            if (dynA) atomicMax(&topRound1[bA], key1);
            if (dynB) atomicMax(&topRound1[bB], key1);
            roundFlags[0] = 1u;
        }
    }
    color[m] = c;
}

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

// ------------------------------------------------------------------------------------------------
// Contact schedule: Jones-Plassmann colouring of the manifold graph (replaces the serial greedy
// scheduleConstraintsSIMD, src/physics/constraints.cpp:51-184).  Two manifolds conflict when they
// share a body with invMass != 0 (the reference exempts its dummy body, constraints.cpp:81-83).
// Priority = pairPriority(colliderA, colliderB) (unique); a manifold colours itself once it is the
// top-priority uncoloured manifold on both of its bodies, taking the lowest colour free on both.
// The result equals sequential greedy colouring in descending priority order, which is what the
// oracle runs.  One launch per round: round r commits the winners of the proposals made in round
// r-1 (keys tagged r in top[r & 1]) and lets the losers propose for round r+1 (tag r+1 in
// top[(r+1) & 1]); a round whose predecessor left nothing uncoloured exits at once, so the host
// enqueues a fixed batch of rounds without reading anything back in between.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void colorRoundBody(const uint32_t m, const uint32_t round, const uint4* __restrict__ colWork, uint32_t* __restrict__ color,
                                               const unsigned long long* __restrict__ topCur, unsigned long long* __restrict__ topNext,
                                               unsigned long long* __restrict__ bodyUsed, uint32_t* __restrict__ roundFlags, const uint32_t seamMode) {
    if (color[m] != kUncolored) return;
    uint4 w = colWork[m];
    bool dynA = (w.x >> 31) != 0, dynB = (w.y >> 31) != 0;
    const bool seam = (w.x & 0x40000000u) != 0u;
    uint32_t bA = w.x & 0x3FFFFFFFu, bB = w.y & 0x7FFFFFFFu;
    unsigned long long prio = ((unsigned long long)w.w << 32) | (unsigned long long)w.z;
    bool lost = true;
    if (round > 0) {
        unsigned long long key = ((unsigned long long)round << 52) | prio;
        lost = (dynA && topCur[bA] != key) || (dynB && topCur[bB] != key);
    }
    if (!lost) {
        unsigned long long mask = (dynA ? bodyUsed[bA] : 0ull) | (dynB ? bodyUsed[bB] : 0ull);
        if (seamMode) mask |= seam ? ~kSeamRange : kSeamRange;
        uint32_t c = kOverflowColor;
        if (~mask != 0ull) {
            c = (uint32_t)__ffsll((long long)~mask) - 1u;
            if (dynA) bodyUsed[bA] |= (1ull << c);   // only one winner per body per round: no race
            if (dynB) bodyUsed[bB] |= (1ull << c);
        }
        color[m] = c;
    } else {
        unsigned long long key = ((unsigned long long)(round + 1) << 52) | prio;
        if (dynA) atomicMax(&topNext[bA], key);
        if (dynB) atomicMax(&topNext[bB], key);
        roundFlags[round] = 1u;
    }
}
__global__ __launch_bounds__(256) void k_color_round(const StepScalars* __restrict__ sc, uint32_t round, const uint4* __restrict__ colWork, uint32_t* __restrict__ color,
                                                     const unsigned long long* __restrict__ topCur, unsigned long long* __restrict__ topNext,
                                                     unsigned long long* __restrict__ bodyUsed, uint32_t* __restrict__ roundFlags, uint32_t seamMode /* exact seam: two colour ranges */) {
    if (round > 0 && roundFlags[round - 1] == 0) return;
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= sc->numManifolds) return;
    colorRoundBody(m, round, colWork, color, topCur, topNext, bodyUsed, roundFlags, seamMode);
}
// The colouring rounds the host did NOT enqueue (speculative steps): the host enqueues exactly as many k_color_round launches as the previous step needed; if the last of
// them still left losers — a growing scene — the first workgroups of the NEXT kernel (k_bin_hist) run the remaining rounds themselves, with a device-wide barrier between two
// rounds, and everybody else waits for them.  Same rounds, same results; what it replaces is a margin of launches that a steady scene paid every step (~4.5 us each) and the
// synchronous re-run a scene paid that outgrew the margin.  In a steady step this is one load per workgroup.
// Words behind the round flags: [kTailBar] barrier arrivals, [kTailDone] 1 + the last round run once the tail is through.
constexpr uint32_t kTailBar = kMaxColorRounds + 2u, kTailDone = kMaxColorRounds + 3u, kTailGroups = 256u, kRoundFlagWords = kMaxColorRounds + 4u;
struct ColorTail {   // (no padding: the launcher hashes arguments bytewise)
    const uint4* colWork; uint32_t* color; unsigned long long* top0; unsigned long long* top1; unsigned long long* bodyUsed; uint32_t* roundFlags;
    uint32_t from /* first round the host did not enqueue; 0: no tail */, seamMode;
};
static_assert(sizeof(ColorTail) == 6 * 8 + 8, "ColorTail must not contain padding");
__device__ __forceinline__ void colorTail(const ColorTail& ct, StepScalars* sc) {
    if (!ct.from || ct.roundFlags[ct.from - 1u] == 0u) return;   // the enqueued rounds coloured everything (the same word for every workgroup: written by the previous launch)
    const uint32_t P = min(gridDim.x, kTailGroups);
    uint32_t* flags = ct.roundFlags;
    if (blockIdx.x >= P) {   // not taking part: wait for the tail (its workgroups have lower indices, i.e. were dispatched before this one)
        if (threadIdx.x == 0) {
            uint32_t budget = 1u << 22;
            while (__hip_atomic_load(&flags[kTailDone], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) { __builtin_amdgcn_s_sleep(8); if (--budget == 0u) { sc->specOverflow = 1u; break; } }
        }
        __syncthreads();
        __threadfence();
        return;
    }
    const uint32_t nm = sc->numManifolds;
    uint32_t arrivals = 0u;
    bool failed = false;
    uint32_t r = ct.from;
    __shared__ uint32_t sMore;
    for (;; ++r) {
        const unsigned long long* topCur = (r & 1u) ? ct.top1 : ct.top0;
        unsigned long long* topNext = (r & 1u) ? ct.top0 : ct.top1;
        for (uint32_t m = blockIdx.x * 256u + threadIdx.x; m < nm; m += P * 256u) colorRoundBody(m, r, ct.colWork, ct.color, topCur, topNext, ct.bodyUsed, flags, ct.seamMode);
        // device-wide barrier: everything this round wrote is visible to everybody before the next one reads it (eight L2s: write back, then invalidate)
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            arrivals += P;
            atomicAdd(&flags[kTailBar], 1u);
            uint32_t budget = 1u << 22;
            while (__hip_atomic_load(&flags[kTailBar], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrivals) { __builtin_amdgcn_s_sleep(2); if (--budget == 0u) { failed = true; break; } }
            sMore = failed ? 2u : __hip_atomic_load(&flags[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        __threadfence();
        const uint32_t more = sMore;
        __syncthreads();
        if (more == 2u) { if (threadIdx.x == 0) sc->specOverflow = 1u; break; }   // (a participant never arrived: the step is void and re-run synchronously)
        if (more == 0u || r + 2u >= kMaxColorRounds) break;                          // round r left no loser: everything is coloured (or: give up, colorPending tells)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->tailRounds = r + 1u - ct.from; __hip_atomic_store(&flags[kTailDone], r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}
// which round's flag says whether the colouring is complete: the last one the host enqueued, or the last one the tail ran
__device__ __forceinline__ uint32_t colorPendingOf(const uint32_t* __restrict__ roundFlags, uint32_t lastRound) {
    const uint32_t t = roundFlags[kTailDone];
    return roundFlags[t ? t - 1u : lastRound];
}

// Schedule slots: manifolds grouped by (colour, contacts per manifold) bin — one stable-enough radix pass
// (block histograms -> exclusive scan -> scatter).  The order inside a bin is irrelevant to results
// (manifolds of one colour share no dynamic body); grouping by contact count makes solver waves uniform.
constexpr uint32_t kBinItems = 1024;
__device__ __forceinline__ uint32_t binOf(uint32_t color, uint32_t cnt) { return color * 4u + (cnt - 1u); }

// XCD-partitioned solver (k_contact_solve_persist<.., true>): the slots of every bin are laid out in ascending order of a
// spatial key (`perm` below), so a bin's tiles sweep the scene along its longest axis; tile `tl` of the bin's `nt` tiles
// belongs to XCD floor(8 tl / nt), i.e. every XCD gets an equal share of EVERY bin (balanced) and always the same slab of
// the scene (bodies away from the slab seams are only ever touched from one XCD).  Bins shorter than 8 tiles are dealt
// round-robin instead.  Results do not depend on any of this: which lane / wave / XCD runs a slot is invisible to the
// body-version dataflow.
// `single` (small piles, < 16384 manifolds): EVERY tile belongs to XCD 0 — the 128 waves of one XCD run the whole solve and every
// body is "local", i.e. all hand-overs go through one L2 instead of through memory (a third of the round trip, and the small
// piles are bound by exactly that: ~10 colours x sweeps hand-overs in a row, a handful of tiles per colour).
__host__ __device__ __forceinline__ uint32_t tileOwner(uint32_t tl, uint32_t nt, uint32_t bin, uint32_t single = 0u) {
    if (single) return 0u;
    return nt >= 8u ? (tl * 8u) / nt : ((tl * 8u) / nt + bin) & 7u;
}
// number of tiles tl' < tl of the same bin with the same owner
__host__ __device__ __forceinline__ uint32_t tileOwnerRank(uint32_t tl, uint32_t nt, uint32_t bin, uint32_t single = 0u) {
    if (single) return tl;
    if (nt >= 8u) { uint32_t x = (tl * 8u) / nt; return tl - (x * nt + 7u) / 8u; }
    uint32_t x = tileOwner(tl, nt, bin), r = 0;
    for (uint32_t k = 0; k < tl; ++k) r += tileOwner(k, nt, bin) == x ? 1u : 0u;
    return r;
}
__host__ __device__ __forceinline__ uint32_t tileOwnerCount(uint32_t x, uint32_t nt, uint32_t bin, uint32_t single = 0u) {
    if (single) return x == 0u ? nt : 0u;
    if (nt >= 8u) return ((x + 1u) * nt + 7u) / 8u - (x * nt + 7u) / 8u;
    uint32_t r = 0;
    for (uint32_t k = 0; k < nt; ++k) r += tileOwner(k, nt, bin) == x ? 1u : 0u;
    return r;
}
// Spatial order of the manifolds: a counting sort by the position of the manifold's (first dynamic) body along the
// longest axis of the broad-phase grid, kSpatialKeys levels; the order inside one level is arbitrary.  Two kernels:
//   k_manifold_keys   key + arrival rank.  Neighbouring manifolds mostly share a key, so the ranks are taken in an LDS
//                     histogram per workgroup and only one global atomic per (workgroup, key present) reserves the range;
//   k_manifold_place  every workgroup scans the 4096 counts itself (cheaper than a separate scan launch) and places its items.
constexpr uint32_t kKeyItems = 1024;   // manifolds per workgroup of k_manifold_keys
struct KeysArgs {   // k_manifold_keys' arguments; no padding bytes (the launcher hashes arguments bytewise)
    uint32_t n, nb; const StepScalars* sc; const GridParams* gp; const uint2* manBodies;
    const float4* bPos; const float4* bCogInvMass;   // the key comes from the body's origin at the start of the step (not from the centre of gravity k_integrate_forces computes: the two run side by side)
    uint32_t* keys; uint32_t* ranks; uint32_t* keyCount;
    // sharded world (bodyActive non-null): also what k_shard_count counts — this rank's manifolds / contacts by the owner rule — from the rows this kernel gathers anyway, one launch less
    const uint8_t* bodyActive; const uint2* manInfo; Shards* sh;
};
static_assert(sizeof(KeysArgs) == 8 + 11 * 8, "KeysArgs must not contain padding");
__device__ __forceinline__ void manifoldKeysBody(const uint32_t blockId, const KeysArgs& ka) {
    __shared__ uint32_t hist[kSpatialKeys];   // local count, then the global base of this workgroup's range
    __shared__ uint32_t ownedCnt[2];
    const uint2* __restrict__ manBodies = ka.manBodies; const float4* __restrict__ bPos = ka.bPos; const float4* __restrict__ bCogInvMass = ka.bCogInvMass;
    uint32_t* __restrict__ keys = ka.keys; uint32_t* __restrict__ ranks = ka.ranks; uint32_t* __restrict__ keyCount = ka.keyCount;
    const uint8_t* __restrict__ bodyActive = ka.bodyActive; const uint2* __restrict__ manInfo = ka.manInfo; const uint32_t nb = ka.nb;
    if (threadIdx.x < 2) ownedCnt[threadIdx.x] = 0u;
    uint32_t mine = 0, contacts = 0;
    for (uint32_t k = threadIdx.x; k < kSpatialKeys; k += 256) hist[k] = 0u;
    __syncthreads();
    const uint32_t nm = min(ka.n, ka.sc->numManifolds);
    const GridParams g = *ka.gp;
    const uint32_t axis = g.dims[0] >= g.dims[1] && g.dims[0] >= g.dims[2] ? 0u : g.dims[2] >= g.dims[1] ? 2u : 1u;
    const uint32_t dimA = axis == 0u ? g.dims[0] : axis == 1u ? g.dims[1] : g.dims[2];          // (selects, not g.dims[axis]: a dynamically indexed copy lives in scratch)
    const float originA = axis == 0u ? g.origin[0] : axis == 1u ? g.origin[1] : g.origin[2];
    const float scale = g.invCell * ((float)kSpatialKeys / (float)dimA);
    uint32_t key[kKeyItems / 256], local[kKeyItems / 256];
#pragma unroll
    for (uint32_t i = 0; i < kKeyItems / 256; ++i) {
        const uint32_t m = blockId * kKeyItems + i * 256 + threadIdx.x;
        key[i] = 0xFFFFFFFFu;
        if (m < nm) {
            uint2 b = manBodies[m];
            const bool dynA = b.x < nb && bCogInvMass[b.x].w != 0.f;
            const uint32_t first = dynA ? b.x : b.y;                     // the manifold's first dynamic body (a manifold has one)
            float c = 0.f;
            if (first < nb) { const float4 p = bPos[first]; c = axis == 0u ? p.x : axis == 1u ? p.y : p.z; }
            key[i] = (uint32_t)fminf(fmaxf((c - originA) * scale, 0.f), (float)(kSpatialKeys - 1u));
            local[i] = atomicAdd(&hist[key[i]], 1u);
            if (bodyActive && first < nb && bodyActive[first] == 1u) { ++mine; contacts += manInfo[m].x & 7u; }
        }
    }
    if (bodyActive) {
        for (int off = 32; off >= 1; off >>= 1) { mine += __shfl_xor(mine, off, 64); contacts += __shfl_xor(contacts, off, 64); }
        if ((threadIdx.x & 63u) == 0u && mine) { atomicAdd(&ownedCnt[0], mine); atomicAdd(&ownedCnt[1], contacts); }
    }
    __syncthreads();
    if (bodyActive && threadIdx.x < 2 && ownedCnt[threadIdx.x]) atomicAdd(&ka.sh->c[blockId & (kShards - 1u)].owned[1 + threadIdx.x], ownedCnt[threadIdx.x]);
    for (uint32_t k = threadIdx.x; k < kSpatialKeys; k += 256) { uint32_t c = hist[k]; if (c) hist[k] = atomicAdd(&keyCount[k], c); }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kKeyItems / 256; ++i) {
        const uint32_t m = blockId * kKeyItems + i * 256 + threadIdx.x;
        if (key[i] != 0xFFFFFFFFu) { keys[m] = key[i]; ranks[m] = hist[key[i]] + local[i]; }
    }
}
__global__ __launch_bounds__(256) void k_manifold_keys(KeysArgs ka) { manifoldKeysBody(blockIdx.x, ka); }
// k_integrate_forces and k_manifold_keys in one launch: neither reads what the other writes; the first `keyBlocks` workgroups take the keys (a chain of LDS and global
// atomics: started first), the others stream the bodies behind them.
template <bool STRIDED>
__global__ __launch_bounds__(256) void k_forces_keys(uint32_t keyBlocks, ForcesArgs fa, KeysArgs ka) {
    if (blockIdx.x < keyBlocks) manifoldKeysBody(blockIdx.x, ka);
    else integrateForcesBlocks<STRIDED>(blockIdx.x - keyBlocks, gridDim.x - keyBlocks, fa);
}
__global__ __launch_bounds__(256) void k_manifold_place(uint32_t n, const StepScalars* __restrict__ sc, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ranks,
                                                        const uint32_t* __restrict__ keyCount, uint32_t* __restrict__ perm) {
    __shared__ uint32_t lower[kSpatialKeys];
    __shared__ uint32_t part[256];
    constexpr uint32_t per = kSpatialKeys / 256;
    uint32_t v[per], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) { v[k] = keyCount[threadIdx.x * per + k]; sum += v[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) { lower[threadIdx.x * per + k] = run; run += v[k]; }
    __syncthreads();
    const uint32_t nm = min(n, sc->numManifolds);
#pragma unroll
    for (uint32_t i = 0; i < kKeyItems / 256; ++i) {
        const uint32_t m = blockIdx.x * kKeyItems + i * 256 + threadIdx.x;
        if (m < nm) perm[lower[keys[m]] + ranks[m]] = m;
    }
}
__global__ __launch_bounds__(256) void k_bin_hist(const StepScalars* __restrict__ sc, uint32_t numBlocks, const uint32_t* __restrict__ perm /* spatially sorted manifold ids, or null */,
                                                  const uint32_t* __restrict__ color, const uint2* __restrict__ manInfo, uint32_t* __restrict__ blockHist, ColorTail tail, StepScalars* scw) {
    __shared__ uint32_t h[kColorBins];
    colorTail(tail, scw);
    const uint32_t nm = sc->numManifolds;
    for (uint32_t b = threadIdx.x; b < kColorBins; b += 256) h[b] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kBinItems / 256; ++k) {
        uint32_t m = blockIdx.x * kBinItems + k * 256 + threadIdx.x;
        if (perm && m < nm) m = perm[m];
        if (m < nm) { uint32_t c = color[m]; if (c <= kOverflowColor) atomicAdd(&h[binOf(c, manInfo[m].x & 7u)], 1u); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < kColorBins; b += 256) blockHist[(size_t)b * numBlocks + blockIdx.x] = h[b];
}
__global__ __launch_bounds__(256) void k_bin_scatter(uint32_t lastRound, const uint32_t* __restrict__ roundFlags, uint32_t numBlocks, const uint32_t* __restrict__ perm,
                                                     const uint32_t* __restrict__ color, const uint2* __restrict__ manInfo,
                                                     const uint32_t* __restrict__ blockScan, uint32_t* __restrict__ order, StepScalars* sc) {
    __shared__ uint32_t cur[kColorBins];
    const uint32_t nm = sc->numManifolds;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->colorPending = colorPendingOf(roundFlags, lastRound);
    for (uint32_t b = threadIdx.x; b < kColorBins; b += 256) {
        uint32_t v = blockScan[(size_t)b * numBlocks + blockIdx.x];
        cur[b] = v;
        if (blockIdx.x == 0) sc->binStart[b] = v;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kBinItems / 256; ++k) {
        uint32_t m = blockIdx.x * kBinItems + k * 256 + threadIdx.x;
        if (perm && m < nm) m = perm[m];
        if (m < nm) { uint32_t c = color[m]; if (c <= kOverflowColor) order[atomicAdd(&cur[binOf(c, manInfo[m].x & 7u)], 1u)] = m; }
    }
    // End of the last bin = the number of SCHEDULED manifolds (the last workgroup's cursor of the last bin ends there).  It
    // equals numManifolds unless a speculative step left manifolds uncoloured or beyond the launched range; those must not
    // become slots (their `order` entries were never written), the step is then re-run anyway.
    __syncthreads();
    if (blockIdx.x == numBlocks - 1 && threadIdx.x == 0) sc->binStart[kColorBins] = cur[kColorBins - 1];
}

// Overflow colour (a body with > 64 incident manifolds) is solved sequentially, so its slots need a defined order:
// ascending pair key (bucket, A, B) like the oracle.  Rank sort by one workgroup; the overflow set is tiny or empty.
__global__ __launch_bounds__(256) void k_sort_overflow(uint32_t s0, uint32_t n, const uint32_t* __restrict__ manPair, const uint64_t* __restrict__ pairKeys,
                                                       const uint32_t* __restrict__ orderIn, uint32_t* __restrict__ orderOut) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t m = orderIn[s0 + i];
        uint64_t key = pairKeys[manPair[m]];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += pairKeys[manPair[orderIn[s0 + j]]] < key ? 1u : 0u;
        orderOut[s0 + rank] = m;
    }
}

// ------------------------------------------------------------------------------------------------
// Contact constraints.  The schedule groups manifolds into bins (colour, contacts per manifold); every
// bin is cut into TILES of 64 slots = one wave.  All constraint data of a tile is contiguous in HBM:
//   rows : [contact-tile ct][row r = 0..5][lane]  float4   (one contact-tile = 6 KiB)
//   imp  : [contact-tile ct][lane]                float4   (accumulated normal, tangent impulse, sweep tag, -)
//   meta : [tile][lane] uint4 = (bodyA, bodyB, friction|restitution, contacts; 0 = padding lane)
//   nrm  : [tile][lane] float4 = (normal, friction) — shared by the contacts of a manifold
// where tile T of a bin with k contacts per manifold owns contact-tiles ctStart + (T - tileStart) * k + 0..k-1,
// so a wave streams one contiguous 6.5 * k KiB block per PGS sweep (DRAM-page and TLB friendly) and
// its control flow is uniform (k is a template parameter).  96 B per contact (the reference's scalar
// collision_constraint is 104 B):
//   r0 = (rA, effMassN)  r1 = (rB, effMassT)  r2 = (tangent, bias)
//   r3 = (-tA.xyz, tB.x)  r4 = (tB.yz, -nA.xy)  r5 = (-nA.z, nB.xyz)
//   with tA = I_A^-1 (rA x t), tB = I_B^-1 (rB x t), nA = I_A^-1 (rA x n), nB = I_B^-1 (rB x n); body A's are stored negated (x - a * b == x + (-a) * b exactly)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kRows = 6;
constexpr uint32_t kSchedBins = kOverflowColor * 4 + 1;   // 256 regular bins + the overflow colour as one bin (stride 4)

struct BinInfo { uint32_t slotStart, count, tileStart, ctStart; };   // host-computed from StepScalars::binStart, uploaded every step

__device__ __forceinline__ M3 loadM3(const float4* __restrict__ p, uint32_t i) {
    float4 a = p[3 * i], b = p[3 * i + 1], c = p[3 * i + 2];
    M3 m; m.m00 = a.x; m.m01 = a.y; m.m02 = a.z; m.m10 = b.x; m.m11 = b.y; m.m12 = b.z; m.m20 = c.x; m.m21 = c.y; m.m22 = c.z;
    return m;
}

// Schedule bins -> tiles, on the device (so the host never has to read the bin sizes back before it can launch the
// constraint kernels): BinInfo per bin, tile -> bin and tile -> (first contact-tile, contacts per manifold) tables, totals.
// The host launches the consumers over an upper bound of tiles; tiles >= totalTiles exit.
// exclusive prefix sum over n values by ONE wave (64-value chunks, shuffle scan inside a chunk); returns the total
template <class Get, class Put>
__device__ __forceinline__ uint32_t waveExclusiveScan(uint32_t n, Get get, Put put) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t i = base + lane;
        const uint32_t v = i < n ? get(i) : 0u;
        uint32_t incl = v;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        if (i < n) put(i, carry + incl - v);
        carry += (uint32_t)__shfl((int)incl, 63, 64);
    }
    return carry;
}
__global__ __launch_bounds__(256) void k_build_tiles(uint32_t tilesCap, uint32_t ctCap, StepScalars* sc, BinInfo* __restrict__ binInfo, uint32_t* __restrict__ xcdBase /* [kSchedBins][8] or null */, uint32_t xcdSingle) {
    __shared__ BinInfo bins[kSchedBins];
    __shared__ uint32_t start[kColorBins + 4];
    __shared__ uint32_t totals[2];
    for (uint32_t b = threadIdx.x; b <= kColorBins; b += blockDim.x) start[b] = sc->binStart[b];
    __syncthreads();
    for (uint32_t bn = threadIdx.x; bn < kSchedBins; bn += blockDim.x) {
        const bool ovf = bn == kSchedBins - 1;
        bins[bn].slotStart = start[bn]; bins[bn].count = (ovf ? start[kColorBins] : start[bn + 1]) - start[bn];
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6;
    auto tilesOf = [&](uint32_t bn) { return (bins[bn].count + 63u) >> 6; };
    if (wave == 0) { uint32_t t = waveExclusiveScan(kSchedBins, tilesOf, [&](uint32_t bn, uint32_t v) { bins[bn].tileStart = v; }); if ((threadIdx.x & 63u) == 0) totals[0] = t; }
    if (wave == 1) { uint32_t t = waveExclusiveScan(kSchedBins, [&](uint32_t bn) { return tilesOf(bn) * (bn == kSchedBins - 1 ? 4u : (bn & 3u) + 1u); },
                                                    [&](uint32_t bn, uint32_t v) { bins[bn].ctStart = v; }); if ((threadIdx.x & 63u) == 0) totals[1] = t; }
    __syncthreads();
    const bool ok = totals[0] <= tilesCap && totals[1] <= ctCap;
    if (threadIdx.x == 0) {
        sc->totalTiles = ok ? totals[0] : 0u; sc->totalCt = ok ? totals[1] : 0u;
        if (!ok) sc->specOverflow = 1u;
    }
    for (uint32_t bn = threadIdx.x; bn < kSchedBins; bn += blockDim.x) binInfo[bn] = bins[bn];
    if (xcdBase) {   // per-XCD tile lists: first list position of every bin's share (wave w scans XCDs 2w and 2w + 1), and the list lengths
        for (uint32_t x = 2u * wave; x < 2u * wave + 2u; ++x) {
            uint32_t t = waveExclusiveScan(kSchedBins, [&](uint32_t bn) { return tileOwnerCount(x, tilesOf(bn), bn, xcdSingle); }, [&](uint32_t bn, uint32_t v) { xcdBase[bn * 8u + x] = v; });
            if ((threadIdx.x & 63u) == 0) sc->xcdCount[x] = ok && totals[0] ? t : 0u;
        }
    }
}
// tile -> bin (binary search over the bins' first tiles) and tile -> (first contact-tile, contacts per manifold)
__global__ __launch_bounds__(256) void k_fill_tiles(const StepScalars* __restrict__ sc, const BinInfo* __restrict__ binInfo,
                                                    uint4* __restrict__ tileInfo /* what k_contact_init needs of a tile in ONE load: (tile, first slot, count | stride << 8 | XCD << 12, first contact-tile) */,
                                                    uint2* __restrict__ tileDesc,
                                                    const uint32_t* __restrict__ xcdBase, uint32_t* __restrict__ xcdTiles /* [8][listCap] or null */, uint4* __restrict__ xcdInfo /* the same entries in list order */,
                                                    uint32_t listCap, uint32_t xcdSingle) {
    __shared__ uint32_t first[kSchedBins];
    for (uint32_t bn = threadIdx.x; bn < kSchedBins; bn += blockDim.x) first[bn] = binInfo[bn].tileStart;
    __syncthreads();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= sc->totalTiles) return;
    uint32_t lo = 0, hi = kSchedBins - 1;   // last bin whose first tile is <= t (empty bins share a first tile with their successor)
    while (lo < hi) { uint32_t mid = (lo + hi + 1u) >> 1; if (first[mid] <= t) lo = mid; else hi = mid - 1u; }
    BinInfo bi = binInfo[lo];
    uint32_t stride = lo == kSchedBins - 1 ? 4u : (lo & 3u) + 1u;
    const uint32_t tl = t - bi.tileStart, nt = (bi.count + 63u) >> 6;
    const uint32_t x = xcdTiles ? tileOwner(tl, nt, lo, xcdSingle) : 0u;
    const uint4 info = make_uint4(t, bi.slotStart + tl * 64u, min(64u, bi.count - tl * 64u) | (stride << 8) | (x << 12), bi.ctStart + tl * stride);
    tileInfo[t] = info;
    tileDesc[t] = make_uint2(bi.ctStart + tl * stride, stride);
    if (xcdTiles) {
        const uint32_t at = xcdBase[lo * 8u + x] + tileOwnerRank(tl, nt, lo, xcdSingle);
        if (at < listCap) { xcdTiles[(size_t)x * listCap + at] = t; xcdInfo[(size_t)x * listCap + at] = info; }   // (a longer list is reported by the solver kernel: solveError 2)
    }
}

// scatter + history insert of the new manifolds + k_build_tiles + k_fill_tiles in ONE launch (each was a 5-10 us launch doing ~1 us of work, one behind the
// other): workgroups [0, numBlocks) scatter their manifolds into the schedule slots and enter the newly coloured ones into the next step's colour
// history; workgroups [numBlocks, ...) turn the bins into tiles — every one of them derives the bin table itself (257 bins: three wave scans, from the
// first column of the block scan, i.e. without waiting for the scatter) and then fills its 256 tiles; the first of them also publishes the table.
__global__ __launch_bounds__(256) void k_schedule_finish(uint32_t lastRound, const uint32_t* __restrict__ roundFlags, uint32_t numBlocks, const uint32_t* __restrict__ perm,
                                                         const uint32_t* __restrict__ color, const uint2* __restrict__ manInfo, const uint32_t* __restrict__ blockHist,
                                                         const uint32_t* __restrict__ blockScan, uint32_t* __restrict__ order, StepScalars* sc,
                                                         uint32_t nc, const uint32_t* __restrict__ manPair, const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB,
                                                         HistSlot* __restrict__ tab, uint32_t tabMask, const uint8_t* __restrict__ manKept, uint32_t* __restrict__ histDisp,
                                                         uint32_t tilesCap, uint32_t ctCap, BinInfo* __restrict__ binInfo, uint32_t* __restrict__ xcdBase /* [kSchedBins][8] or null */, uint32_t xcdSingle,
                                                         uint4* __restrict__ tileInfo, uint2* __restrict__ tileDesc, uint32_t* __restrict__ xcdTiles /* [8][listCap] or null */, uint4* __restrict__ xcdInfo, uint32_t listCap) {
    __shared__ uint32_t cur[kColorBins + 4];
    if (blockIdx.x < numBlocks) {   // ---- scatter + history insert of the newly coloured manifolds
        const uint32_t nm = sc->numManifolds;
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->colorPending = colorPendingOf(roundFlags, lastRound);
        for (uint32_t b = threadIdx.x; b < kColorBins; b += 256) {
            uint32_t v = blockScan[(size_t)b * numBlocks + blockIdx.x];
            cur[b] = v;
            if (blockIdx.x == 0) sc->binStart[b] = v;
        }
        __syncthreads();
        const uint64_t* __restrict__ pairKeys = sc->partitioned ? pairsB : pairsA;
#pragma unroll
        for (uint32_t k = 0; k < kBinItems / 256; ++k) {
            uint32_t m = blockIdx.x * kBinItems + k * 256 + threadIdx.x;
            if (perm && m < nm) m = perm[m];
            if (m < nm) {
                const uint32_t c = color[m];
                if (c <= kOverflowColor) order[atomicAdd(&cur[binOf(c, manInfo[m].x & 7u)], 1u)] = m;
                if (!manKept[m]) {      // kept colours were entered by k_emit_manifolds
                    const uint64_t pk = pairKeys[manPair[m]];
                    tableInsert(tab, tabMask, historyKey(nc, (uint32_t)((pk >> 29) & 0x1FFFFFFFull), (uint32_t)(pk & 0x1FFFFFFFull)), c, histDisp);
                }
            }
        }
        __syncthreads();
        if (blockIdx.x == numBlocks - 1 && threadIdx.x == 0) sc->binStart[kColorBins] = cur[kColorBins - 1];   // = the number of SCHEDULED manifolds (see k_bin_scatter)
        return;
    }
    // ---- bins -> tiles (k_build_tiles, by every workgroup for itself) + the tile tables (k_fill_tiles)
    __shared__ BinInfo bins[kSchedBins];
    __shared__ uint32_t totals[2];
    __shared__ uint32_t xb[kSchedBins * 8u];
    const uint32_t tb = blockIdx.x - numBlocks;      // tile workgroup
    for (uint32_t b = threadIdx.x; b < kColorBins; b += blockDim.x) cur[b] = blockScan[(size_t)b * numBlocks];
    if (threadIdx.x == 0) { const size_t lastEl = (size_t)(kColorBins - 1u) * numBlocks + (numBlocks - 1u); cur[kColorBins] = blockScan[lastEl] + blockHist[lastEl]; }   // end of the last bin
    __syncthreads();
    for (uint32_t bn = threadIdx.x; bn < kSchedBins; bn += blockDim.x) {
        const bool ovf = bn == kSchedBins - 1;
        bins[bn].slotStart = cur[bn]; bins[bn].count = (ovf ? cur[kColorBins] : cur[bn + 1]) - cur[bn];
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6;
    auto tilesOf = [&](uint32_t bn) { return (bins[bn].count + 63u) >> 6; };
    if (wave == 0) { uint32_t t = waveExclusiveScan(kSchedBins, tilesOf, [&](uint32_t bn, uint32_t v) { bins[bn].tileStart = v; }); if ((threadIdx.x & 63u) == 0) totals[0] = t; }
    if (wave == 1) { uint32_t t = waveExclusiveScan(kSchedBins, [&](uint32_t bn) { return tilesOf(bn) * (bn == kSchedBins - 1 ? 4u : (bn & 3u) + 1u); },
                                                    [&](uint32_t bn, uint32_t v) { bins[bn].ctStart = v; }); if ((threadIdx.x & 63u) == 0) totals[1] = t; }
    __syncthreads();
    const bool ok = totals[0] <= tilesCap && totals[1] <= ctCap;
    if (xcdTiles) {
        for (uint32_t x = 2u * wave; x < 2u * wave + 2u; ++x) {
            uint32_t t = waveExclusiveScan(kSchedBins, [&](uint32_t bn) { return tileOwnerCount(x, tilesOf(bn), bn, xcdSingle); }, [&](uint32_t bn, uint32_t v) { xb[bn * 8u + x] = v; });
            if (tb == 0 && (threadIdx.x & 63u) == 0) sc->xcdCount[x] = ok && totals[0] ? t : 0u;
        }
    }
    if (tb == 0) {      // publish the table (the host mirrors binStart; the per-colour fallback kernels read binInfo)
        if (threadIdx.x == 0) { sc->totalTiles = ok ? totals[0] : 0u; sc->totalCt = ok ? totals[1] : 0u; if (!ok) sc->specOverflow = 1u; }
        for (uint32_t bn = threadIdx.x; bn < kSchedBins; bn += blockDim.x) binInfo[bn] = bins[bn];
    }
    __syncthreads();
    if (tb == 0 && xcdBase && xcdTiles) for (uint32_t i = threadIdx.x; i < kSchedBins * 8u; i += blockDim.x) xcdBase[i] = xb[i];
    const uint32_t t = tb * blockDim.x + threadIdx.x;
    if (!ok || t >= totals[0]) return;
    uint32_t lo = 0, hi = kSchedBins - 1;   // last bin whose first tile is <= t (empty bins share a first tile with their successor)
    while (lo < hi) { uint32_t mid = (lo + hi + 1u) >> 1; if (bins[mid].tileStart <= t) lo = mid; else hi = mid - 1u; }
    const BinInfo bi = bins[lo];
    const uint32_t stride = lo == kSchedBins - 1 ? 4u : (lo & 3u) + 1u;
    const uint32_t tl = t - bi.tileStart, nt = (bi.count + 63u) >> 6;
    const uint32_t x = xcdTiles ? tileOwner(tl, nt, lo, xcdSingle) : 0u;
    const uint4 info = make_uint4(t, bi.slotStart + tl * 64u, min(64u, bi.count - tl * 64u) | (stride << 8) | (x << 12), bi.ctStart + tl * stride);
    tileInfo[t] = info;
    tileDesc[t] = make_uint2(bi.ctStart + tl * stride, stride);
    if (xcdTiles) {
        const uint32_t at = xb[lo * 8u + x] + tileOwnerRank(tl, nt, lo, xcdSingle);
        if (at < listCap) { xcdTiles[(size_t)x * listCap + at] = t; xcdInfo[(size_t)x * listCap + at] = info; }   // (a longer list is reported by the solver kernel: solveError 2)
    }
}

// The constraint rows are written once here and read by the solver from memory: stored non-temporally they do not push the bodies this kernel gathers
// (one slab of the scene per XCD) out of that XCD's L2: 81 -> 76 us for the stage (A/B against a build with plain stores, same box; -DMI_NO_STREAM_ROWS).
typedef float mi_vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void storeStream(float4* p, float4 v) {
#ifdef MI_NO_STREAM_ROWS
    *p = v;
#else
    mi_vf4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<mi_vf4*>(p));
#endif
}
// private joint islands (joints.hpp "PRIVATE islands"): what k_contact_init needs of them
constexpr uint32_t kIslandMaxContacts = 64;
struct IslandPrivate {
    const uint32_t* bodyIsland;   // [bodies + 1]: island of a dynamic island body, else 0xFFFFFFFF
    uint32_t* shared;             // [islands] this step: 1 = some manifold couples the island to a dynamic body outside it (or is overflow-coloured)
    uint32_t* count;              // [islands] this step: manifolds touching the island
    uint32_t* fill;               // [islands] this step: entries appended by k_contact_init
    uint4* entries;               // [islands][kIslandMaxContacts]: (slot, first contact-tile, colour | contacts << 8 | per-contact normals << 16, manifold)
};
__device__ __forceinline__ bool islandIsPrivate(const IslandPrivate& ip, uint32_t island) { return ip.shared[island] == 0u && ip.count[island] <= kIslandMaxContacts; }
// K11 "Initialize collision constraints" (src/physics/constraints.cpp:3307-3379): one wave per tile, one lane per slot.
__global__ __launch_bounds__(64) void k_contact_init(const StepScalars* __restrict__ sc, uint32_t dummyBody, float dt, const uint4* __restrict__ tileInfo /* k_fill_tiles: per tile, or (XCD-partitioned) per entry of the XCD tile lists */,
                                                     const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ manPair, const uint2* __restrict__ manBodies,
                                                     const uint2* __restrict__ manInfo, const float4* __restrict__ npNormal,
                                                     const float4* __restrict__ npPoints, const float4* __restrict__ gPos,
                                                     const float4* __restrict__ gInvI, const float4* __restrict__ gVel,
                                                     const uint32_t* __restrict__ color, const unsigned long long* __restrict__ bodyUsed,
                                                     const uint8_t* __restrict__ bodyJ /* fused joint islands: 1 = the body gets one joint version per sweep, or null */,
                                                     float4* __restrict__ rows, float4* __restrict__ imp, uint4* __restrict__ slotMeta,
                                                     float4* __restrict__ slotNormal, float2* __restrict__ slotMass,
                                                     uint8_t* __restrict__ bodyOwner /* XCD-partitioned solver: [body][8] flags, 1 = a tile of that XCD touches the body; or null */,
                                                     uint32_t listCap /* with bodyOwner: workgroup b prepares entry b / 8 of XCD (b % 8)'s tile list */, uint32_t infoCap,
                                                     IslandPrivate ip /* bodyIsland non-null: manifolds of private islands are handed to their island's workgroup, invalid for the tile solver */) {
    // Measured and not kept: one wave per contact index (four waves per tile, the per-manifold gathers repeated): 52 -> 73 us; 5 or 6 waves per
    // SIMD instead of 4 by capping the registers (96 / 80 VGPRs, 96 / 164 bytes of scratch): 66 -> 84 / 94 us.  Everything the kernel needs of
    // its tile comes in ONE 16-byte entry (k_fill_tiles; was list -> tile -> bin -> bin info): no faster either — the kernel moves ~390 MB
    // (PMC) in 67 us, it is bound by the body gathers and the row stream, not by the length of its dependent-load chain.
    const uint32_t lane = threadIdx.x; const uint32_t kw = 0;
    // XCD-partitioned: the workgroups that land on XCD x (blockIdx % 8, a speed assumption only) prepare the tiles XCD x will solve,
    // i.e. gather the bodies of ONE slab of the scene — they fit that XCD's L2 instead of streaming all bodies through every L2
    const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t entry = bodyOwner ? x * listCap + j : blockIdx.x;
    const uint4 te = entry < infoCap ? tileInfo[entry] : make_uint4(0u, 0u, 0u, 0u);   // (requested before the validity checks below: their loads run beside it)
    if (bodyOwner) { if (!sc->totalTiles || j >= sc->xcdCount[x] || j >= listCap) return; }
    else if (entry >= sc->totalTiles) return;
    const uint32_t tile = te.x, count = te.z & 0xFFu, stride = (te.z >> 8) & 0xFu;
    if (kw >= stride) return;
    const size_t ctBase = te.w;
    if (lane >= count) {
        if (kw == 0) {
            slotMeta[(size_t)tile * 64u + lane] = make_uint4(dummyBody, dummyBody, 0u, 0u);
            slotMass[(size_t)tile * 64u + lane] = make_float2(0.f, 0.f);
        }
        return;
    }
    const uint32_t m = order[te.y + lane];
    uint32_t p = manPair[m];
    uint2 bodies = manBodies[m];
    uint2 info = manInfo[m];
    uint32_t cnt = info.x & 7u;
    const bool terrain = (info.x >> 31) != 0u;   // contact k and its OWN normal come from pair record p + k (heightmap.hpp, HmOut::put)
    float4 pa = gPos[bodies.x], pb = gPos[bodies.y];
    V3 xA = xyz(pa), xB = xyz(pb);
    float imA = pa.w, imB = pb.w;
    // Update-version bookkeeping for the dataflow solver: the manifolds of a body have distinct colours, so the number
    // of updates a body has received before this manifold's turn in a sweep = colours used on the body below this one.
    //   packed = baseA | degA << 7 | baseB << 14 | degB << 21   (deg = 0: body is never written, nothing to wait for)
    uint32_t packed = 0;
    if (kw == 0) {
        uint32_t c = color[m];
        unsigned long long below = c < 64u ? ((1ull << c) - 1ull) : ~0ull;
        // a body of a joint island is first updated by its island's block in every sweep (k_solve_flow_islands): one more version
        if (imA != 0.f) { unsigned long long u = bodyUsed[bodies.x]; uint32_t j = bodyJ ? bodyJ[bodies.x] : 0u; packed |= ((uint32_t)__popcll(u & below) + j) | (((uint32_t)__popcll(u) + j) << 7); }
        if (imB != 0.f) { unsigned long long u = bodyUsed[bodies.y]; uint32_t j = bodyJ ? bodyJ[bodies.y] : 0u; packed |= (((uint32_t)__popcll(u & below) + j) << 14) | (((uint32_t)__popcll(u) + j) << 21); }
    }
    bool priv = false;
    if (ip.bodyIsland) {
        const uint32_t iA = imA != 0.f ? ip.bodyIsland[bodies.x] : 0xFFFFFFFFu, iB = imB != 0.f ? ip.bodyIsland[bodies.y] : 0xFFFFFFFFu;
        const uint32_t isl = iA != 0xFFFFFFFFu ? iA : iB;
        if (isl != 0xFFFFFFFFu && islandIsPrivate(ip, isl)) {   // (a private island's manifolds have no dynamic body outside it: k_island_classify)
            priv = true;
            const uint32_t at = atomicAdd(&ip.fill[isl], 1u);
            if (at < kIslandMaxContacts) ip.entries[(size_t)isl * kIslandMaxContacts + at] = make_uint4(tile * 64u + lane, (uint32_t)ctBase, color[m] | (cnt << 8) | (terrain ? 1u << 16 : 0u), m);
        }
    }
    const uint32_t metaW = priv ? 0u : (cnt | (terrain ? 0x400u : 0u));   // (.w = 0: not a slot of the tile solver; bits 8 / 9: XCD-local bodies (the solver sets them); bit 10: per-contact normals)
    if (kw == 0) {
        slotMeta[(size_t)tile * 64u + lane] = make_uint4(bodies.x, bodies.y, packed, metaW);
        slotMass[(size_t)tile * 64u + lane] = make_float2(imA, imB);
    }
    if (bodyOwner && kw == 0) {   // one byte per (body, XCD): plain idempotent stores, no atomics
        const uint32_t xo = te.z >> 12;
        if (imA != 0.f) bodyOwner[(size_t)bodies.x * 8u + xo] = 1u;
        if (imB != 0.f) bodyOwner[(size_t)bodies.y * 8u + xo] = 1u;
    }
    M3 IA = loadM3(gInvI, bodies.x), IB = loadM3(gInvI, bodies.y);
    V3 vA = xyz(gVel[2 * bodies.x]), wA = xyz(gVel[2 * bodies.x + 1]);
    V3 vB = xyz(gVel[2 * bodies.y]), wB = xyz(gVel[2 * bodies.y + 1]);
    V3 n = xyz(npNormal[p]);   // (a terrain manifold: the normal of its first contact; the others follow in the loop)
    float invDt = 1.f / dt;
    float friction = (float)(info.y >> 16) / (float)0xFFFF;
    float restitution = (float)(info.y & 0xFFFF) / (float)0xFFFF;
    if (kw == 0) slotNormal[(size_t)tile * 64u + lane] = f4(n, friction);
    for (uint32_t k = 0; k < cnt; ++k) {
        float4 pd = terrain ? npPoints[4 * ((size_t)p + k)] : npPoints[4 * (size_t)p + k];
        if (terrain && k) n = xyz(npNormal[p + k]);
        V3 point = xyz(pd); float depth = pd.w;
        V3 rA = point - xA, rB = point - xB;
        V3 avA = vA + cross(wA, rA), avB = vB + cross(wB, rB);
        V3 rel = avB - avA;
        V3 t = rel - dot(n, rel) * n;
        t = noz(t);
        V3 crAt = cross(rA, t), crBt = cross(rB, t);
        V3 tA = mul(IA, crAt), tB = mul(IB, crBt);
        float invMT = imA + dot(crAt, tA) + imB + dot(crBt, tB);
        float effT = (invMT != 0.f) ? (1.f / invMT) : 0.f;
        V3 crAn = cross(rA, n), crBn = cross(rB, n);
        V3 nA = mul(IA, crAn), nB = mul(IB, crBn);
        float invMN = imA + dot(crAn, nA) + imB + dot(crBn, nB);
        float effN = (invMN != 0.f) ? (1.f / invMN) : 0.f;
        float bias = 0.f;
        if (dt > 1e-5f) {
            float vRel = dot(n, rel);
            const float slop = -0.001f;
            if (-depth < slop && vRel < 0.f) bias = -restitution * vRel - 0.1f * (-depth - slop) * invDt;
        }
        float4* __restrict__ row = rows + (ctBase + k) * (kRows * 64u) + lane;
        storeStream(row + 0 * 64, f4(rA, effN));
        // (terrain: body B is the static dummy — zero velocity, zero inverse mass and inertia — so its lever arm only ever meets zeros (v_B + w_B x r_B = +0 whatever
        // r_B is, finite); the contact's own normal travels in its place, where the solver's per-contact-normal path picks it up)
        storeStream(row + 1 * 64, f4(terrain ? n : rB, effT));
        storeStream(row + 2 * 64, f4(t, bias));
        storeStream(row + 3 * 64, make_float4(-tA.x, -tA.y, -tA.z, tB.x));
        storeStream(row + 4 * 64, make_float4(tB.y, tB.z, -nA.x, -nA.y));
        storeStream(row + 5 * 64, make_float4(-nA.z, nB.x, nB.y, nB.z));
        if (imp) imp[(ctBase + k) * 64u + lane] = make_float4(0.f, 0.f, 0.f, 0.f);   // no warm start (constraints.cpp:3312-3313); sweep tag 0 (null: the solver keeps the impulses in LDS)
    }
}

// One PGS update of one contact (src/physics/constraints.cpp:3381-3449): friction first (clamped with the
// previous normal impulse), then the normal row.
struct ContactRows { float4 r[kRows]; float4 imp; };   // imp = (normal, tangent, sweep tag, -)
// (normal, friction) of ONE contact: the slot's — every contact of a manifold shares the normal — except in a terrain manifold (slotMeta.w bit 10), whose contacts each
// carry their own in the place of body B's lever arm (k_contact_init; src/physics/heightmap_collision.cpp:575-594: one contact per triangle hit, each with its normal)
constexpr uint32_t kMetaPerContactNormal = 0x400u;
__device__ __forceinline__ float4 contactNormal(const ContactRows& c, const float4 nf, const bool perContact) {
    return perContact ? make_float4(c.r[1].x, c.r[1].y, c.r[1].z, nf.w) : nf;
}

__device__ __forceinline__ void solveOne(const ContactRows& c, const float4 nf, float2& im, float imA, float imB, V3& vA, V3& wA, V3& vB, V3& wB) {
    V3 rA = xyz(c.r[0]), rB = xyz(c.r[1]), t = xyz(c.r[2]), n = xyz(nf);
    V3 tA(c.r[3].x, c.r[3].y, c.r[3].z), tB(c.r[3].w, c.r[4].x, c.r[4].y), nA(c.r[4].z, c.r[4].w, c.r[5].x), nB(c.r[5].y, c.r[5].z, c.r[5].w);   // tA, nA: negated (k_contact_init)
    {
        V3 avA = vA + cross(wA, rA), avB = vB + cross(wB, rB);
        V3 rel = avB - avA;
        float vt = dot(rel, t);
        float lambda = -c.r[1].w * vt;
        float maxF = nf.w * im.x;
        float ni = clampr(im.y + lambda, -maxF, maxF);
        lambda = ni - im.y;
        im.y = ni;
        V3 P = lambda * t;
        vA = vA - imA * P;
        wA = wA + tA * lambda;
        vB = vB + imB * P;
        wB = wB + tB * lambda;
    }
    {
        V3 avA = vA + cross(wA, rA), avB = vB + cross(wB, rB);
        V3 rel = avB - avA;
        float vn = dot(rel, n);
        float lambda = -c.r[0].w * (vn - c.r[2].w);
        float ni = fmaxr(im.x + lambda, 0.f);
        lambda = ni - im.x;
        im.x = ni;
        V3 P = lambda * n;
        vA = vA - imA * P;
        wA = wA + nA * lambda;
        vB = vB + imB * P;
        wB = wB + nB * lambda;
    }
}

// The same update with bodies A and B side by side in packed-fp32 lanes (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations
// per instruction, bit-identical to the scalar ones).  The solve of a tile sits on the dependency chain between tiles, so
// its instruction count is latency, not just throughput.  Signs are folded into the operands: x - a*b == x + (-a)*b exactly.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct P3 { f32x2 x, y, z; };   // one vector per body: lane 0 = body A, lane 1 = body B
__device__ __forceinline__ f32x2 pk2(float a, float b) { f32x2 r = {a, b}; return r; }
__device__ __forceinline__ P3 pcross(const P3& a, const P3& b) { P3 r; r.x = a.y * b.z - a.z * b.y; r.y = a.z * b.x - a.x * b.z; r.z = a.x * b.y - a.y * b.x; return r; }
__device__ __forceinline__ void solveOnePk(const ContactRows& c, const float4 nf, float2& im, const f32x2 sMass /* (-imA, imB) */, P3& v, P3& w) {
    const V3 t = xyz(c.r[2]), n = xyz(nf);
    P3 r; r.x = pk2(c.r[0].x, c.r[1].x); r.y = pk2(c.r[0].y, c.r[1].y); r.z = pk2(c.r[0].z, c.r[1].z);
        // 4-wide shuffles that the backend then legalises THROUGH SCRATCH MEMORY — 16 bytes per contact stored and re-loaded on the tile's dependency chain)
    float tAx = c.r[3].x, tAy = c.r[3].y, tAz = c.r[3].z, nAx = c.r[4].z, nAy = c.r[4].w, nAz = c.r[5].x;
    asm("" : "+v"(tAx)); asm("" : "+v"(tAy)); asm("" : "+v"(tAz)); asm("" : "+v"(nAx)); asm("" : "+v"(nAy)); asm("" : "+v"(nAz));
    P3 T; T.x = pk2(tAx, c.r[3].w); T.y = pk2(tAy, c.r[4].x); T.z = pk2(tAz, c.r[4].y);
    P3 N; N.x = pk2(nAx, c.r[5].y); N.y = pk2(nAy, c.r[5].z); N.z = pk2(nAz, c.r[5].w);
    {
        P3 cr = pcross(w, r);
        f32x2 ax = v.x + cr.x, ay = v.y + cr.y, az = v.z + cr.z;
        V3 rel(ax.y - ax.x, ay.y - ay.x, az.y - az.x);
        float vt = dot(rel, t);
        float lambda = -c.r[1].w * vt;
        float maxF = nf.w * im.x;
        float ni = clampr(im.y + lambda, -maxF, maxF);
        lambda = ni - im.y;
        im.y = ni;
        V3 P = lambda * t;
        v.x = v.x + sMass * P.x; v.y = v.y + sMass * P.y; v.z = v.z + sMass * P.z;
        w.x = w.x + T.x * lambda; w.y = w.y + T.y * lambda; w.z = w.z + T.z * lambda;
    }
    {
        P3 cr = pcross(w, r);
        f32x2 ax = v.x + cr.x, ay = v.y + cr.y, az = v.z + cr.z;
        V3 rel(ax.y - ax.x, ay.y - ay.x, az.y - az.x);
        float vn = dot(rel, n);
        float lambda = -c.r[0].w * (vn - c.r[2].w);
        float ni = fmaxr(im.x + lambda, 0.f);
        lambda = ni - im.x;
        im.x = ni;
        V3 P = lambda * n;
        v.x = v.x + sMass * P.x; v.y = v.y + sMass * P.y; v.z = v.z + sMass * P.z;
        w.x = w.x + N.x * lambda; w.y = w.y + N.y * lambda; w.z = w.z + N.z * lambda;
    }
}

// The rows of one contact as the packed update wants them — (body A, body B) side by side in 64-bit register pairs — built BEFORE a tile waits for its bodies (packRows), and
// pinned there: the register moves that line the halves up then happen while the body loads are in flight, not between the bodies' arrival and the publish.
struct PkRows { f32x2 rx, ry, rz, Tx, Ty, Tz, Nx, Ny, Nz; float tx, ty, tz, effN, effT, bias, nx, ny, nz; };   // (nx, ny, nz: the contact's normal — the slot's, or its own in a terrain manifold)
__device__ __forceinline__ void pinPair(f32x2& p) { asm volatile("" : "+v"(p)); }
// PIN: the pairs are pinned where they are built (the persistent kernel, whose rows come out of its prefetch registers by inline asm).  NOT where the rows come from the
// compiler's own loads (flowTile): there the pinned form gave wrong results on the device in every run (round 5; the unpinned form and the pinned persistent kernel are
// bit-exact, the generated code of the failing form shows no hazard a static check finds) — not understood, so the dispatch-ordered kernels keep the compiler's placement.
template <bool PIN>
__device__ __forceinline__ PkRows packRows(const ContactRows& c, const float4 nf, const bool perContactNormal) {
    PkRows k;
    k.nx = perContactNormal ? c.r[1].x : nf.x; k.ny = perContactNormal ? c.r[1].y : nf.y; k.nz = perContactNormal ? c.r[1].z : nf.z;   // (before the wait for the bodies: off the dependency chain)
    k.rx = pk2(c.r[0].x, c.r[1].x); k.ry = pk2(c.r[0].y, c.r[1].y); k.rz = pk2(c.r[0].z, c.r[1].z);
    // (body A's halves come negated from k_contact_init.  They pass through an empty asm: left alone, the optimiser merges these element picks into 4-wide shuffles that the
    // backend legalises THROUGH SCRATCH MEMORY)
    float tAx = c.r[3].x, tAy = c.r[3].y, tAz = c.r[3].z, nAx = c.r[4].z, nAy = c.r[4].w, nAz = c.r[5].x;
    asm("" : "+v"(tAx)); asm("" : "+v"(tAy)); asm("" : "+v"(tAz)); asm("" : "+v"(nAx)); asm("" : "+v"(nAy)); asm("" : "+v"(nAz));
    k.Tx = pk2(tAx, c.r[3].w); k.Ty = pk2(tAy, c.r[4].x); k.Tz = pk2(tAz, c.r[4].y);
    k.Nx = pk2(nAx, c.r[5].y); k.Ny = pk2(nAy, c.r[5].z); k.Nz = pk2(nAz, c.r[5].w);
    if (PIN) { pinPair(k.rx); pinPair(k.ry); pinPair(k.rz); pinPair(k.Tx); pinPair(k.Ty); pinPair(k.Tz); pinPair(k.Nx); pinPair(k.Ny); pinPair(k.Nz); }
    k.tx = c.r[2].x; k.ty = c.r[2].y; k.tz = c.r[2].z; k.effN = c.r[0].w; k.effT = c.r[1].w; k.bias = c.r[2].w;
    return k;
}
// hi - lo of a pair (body B - body A) as ONE scalar subtraction each (left to itself the SLP vectoriser packs two of the three and pays three register moves for it)
__device__ __forceinline__ float subHiLo(const f32x2 p) { float d; asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(p.y), "v"(p.x)); return d; }
__device__ __forceinline__ void solveOnePkRows(const PkRows& c, const float4 nf, float2& im, const f32x2 sMass /* (-imA, imB) */, P3& v, P3& w) {
    const V3 t(c.tx, c.ty, c.tz), n(c.nx, c.ny, c.nz);
    P3 r; r.x = c.rx; r.y = c.ry; r.z = c.rz;
    {
        P3 cr = pcross(w, r);
        f32x2 ax = v.x + cr.x, ay = v.y + cr.y, az = v.z + cr.z;
        V3 rel(subHiLo(ax), subHiLo(ay), subHiLo(az));
        float vt = dot(rel, t);
        float lambda = -c.effT * vt;
        float maxF = nf.w * im.x;
        float ni = clampr(im.y + lambda, -maxF, maxF);
        lambda = ni - im.y;
        im.y = ni;
        V3 P = lambda * t;
        v.x = v.x + sMass * P.x; v.y = v.y + sMass * P.y; v.z = v.z + sMass * P.z;
        w.x = w.x + c.Tx * lambda; w.y = w.y + c.Ty * lambda; w.z = w.z + c.Tz * lambda;
    }
    {
        P3 cr = pcross(w, r);
        f32x2 ax = v.x + cr.x, ay = v.y + cr.y, az = v.z + cr.z;
        V3 rel(subHiLo(ax), subHiLo(ay), subHiLo(az));
        float vn = dot(rel, n);
        float lambda = -c.effN * (vn - c.bias);
        float ni = fmaxr(im.x + lambda, 0.f);
        lambda = ni - im.x;
        im.x = ni;
        V3 P = lambda * n;
        v.x = v.x + sMass * P.x; v.y = v.y + sMass * P.y; v.z = v.z + sMass * P.z;
        w.x = w.x + c.Nx * lambda; w.y = w.y + c.Ny * lambda; w.z = w.z + c.Nz * lambda;
    }
}

// One tile, CNT contacts per manifold.  Latency structure: a colour launch has < 1 wave per SIMD, so it is bound by
// dependent-load depth: all constraint rows are requested up front (they do not depend on the slot metadata), the
// body gathers follow the metadata — two memory round trips per sweep.
template <int CNT>
__device__ __forceinline__ void solveTile(uint32_t tile, uint32_t ctBase, uint32_t lane, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal,
                                          const float2* __restrict__ slotMass,
                                          const float4* __restrict__ rows, float4* __restrict__ imp, float4* __restrict__ gVel) {
    const uint4 meta = slotMeta[(size_t)tile * 64u + lane];
    const float4 nf = slotNormal[(size_t)tile * 64u + lane];
    const float2 mass = slotMass[(size_t)tile * 64u + lane];
    ContactRows c[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        const float4* __restrict__ row = rows + ((size_t)ctBase + k) * (kRows * 64u) + lane;
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) c[k].r[r] = row[r * 64u];
        c[k].imp = imp[((size_t)ctBase + k) * 64u + lane];
    }
    uint32_t bA = meta.x, bB = meta.y;
    float4 a0 = gVel[2 * bA], a1 = gVel[2 * bA + 1], b0 = gVel[2 * bB], b1 = gVel[2 * bB + 1];
    float imA = mass.x, imB = mass.y;
    // No early exit: a branch here would let the compiler sink the row loads below it and serialise four memory
    // round trips (meta -> bodies -> rows -> stores).  Padding lanes (meta.w == 0) and manifolds without a dynamic
    // body compute on whatever they loaded and simply do not store.
    const bool live = meta.w != 0u && (imA != 0.f || imB != 0.f);
    V3 vA = xyz(a0), wA = xyz(a1), vB = xyz(b0), wB = xyz(b1);
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        float2 im = make_float2(c[k].imp.x, c[k].imp.y);
        solveOne(c[k], contactNormal(c[k], nf, (meta.w & kMetaPerContactNormal) != 0u), im, imA, imB, vA, wA, vB, wB);
        if (live) imp[((size_t)ctBase + k) * 64u + lane] = make_float4(im.x, im.y, c[k].imp.z, c[k].imp.w);
    }
    if (live && imA != 0.f) { gVel[2 * bA] = f4(vA, a0.w); gVel[2 * bA + 1] = f4(wA, a1.w); }   // .w: version tags, untouched by this path
    if (live && imB != 0.f) { gVel[2 * bB] = f4(vB, b0.w); gVel[2 * bB + 1] = f4(wB, b1.w); }
}

__device__ __forceinline__ void solveTileK(uint32_t k, uint32_t tile, uint32_t ctBase, uint32_t lane, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal,
                                           const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* __restrict__ imp, float4* __restrict__ gVel) {
    switch (k) {
        case 1: solveTile<1>(tile, ctBase, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel); break;
        case 2: solveTile<2>(tile, ctBase, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel); break;
        case 3: solveTile<3>(tile, ctBase, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel); break;
        default: solveTile<4>(tile, ctBase, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel); break;
    }
}

// K12 "Solve collision constraints": one launch per colour, one wave per tile; lanes own disjoint dynamic bodies.
// Blocks are ordered 4-contact tiles first (longest first).  With `swizzle`, consecutive tiles (= spatially
// coherent manifolds, hence neighbouring bodies) are dealt to one XCD (block b runs on XCD b % 8) so the body
// velocity lines of a region stay in that XCD's L2.
struct ColorLaunch { uint32_t tileStart[4]; uint32_t blockEnd[4]; uint32_t ctStart[4]; uint32_t numBlocks; uint32_t swizzle; };
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_contact_solve(ColorLaunch cl, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal,
                                                      const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* __restrict__ imp, float4* __restrict__ gVel) {
    uint32_t b = blockIdx.x;
    if (cl.swizzle) {
        uint32_t per = (cl.numBlocks + 7u) >> 3;
        b = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if (b >= cl.numBlocks) return;
    }
    uint32_t lane = threadIdx.x;
    // blockEnd is cumulative over k = 4, 3, 2, 1
    if (b < cl.blockEnd[0]) { solveTile<4>(cl.tileStart[3] + b, cl.ctStart[3] + b * 4u, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel); return; }
    if (b < cl.blockEnd[1]) { uint32_t t = b - cl.blockEnd[0]; solveTile<3>(cl.tileStart[2] + t, cl.ctStart[2] + t * 3u, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel); return; }
    if (b < cl.blockEnd[2]) { uint32_t t = b - cl.blockEnd[1]; solveTile<2>(cl.tileStart[1] + t, cl.ctStart[1] + t * 2u, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel); return; }
    { uint32_t t = b - cl.blockEnd[2]; solveTile<1>(cl.tileStart[0] + t, cl.ctStart[0] + t, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel); }
}

// Trailing colours of the greedy colouring are tiny; one 256-lane workgroup runs colours [c0, c1) back to back
// with a workgroup barrier + workgroup-scope fence between them instead of one launch each (a colour costs one
// dependent-load chain, ~1.5 us, inside the kernel vs ~5.5 us as its own launch).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_contact_solve_tail(const BinInfo* __restrict__ binInfo, uint32_t c0, uint32_t c1, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal,
                                                             const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* __restrict__ imp, float4* __restrict__ gVel) {
    uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t c = c0; c < c1; ++c) {
        uint32_t g = 0;
        for (uint32_t k = 0; k < 4; ++k) {
            BinInfo bi = binInfo[c * 4u + k];
            uint32_t nt = (bi.count + 63u) >> 6;
            for (uint32_t tl = 0; tl < nt; ++tl, ++g)
                if ((g & 3u) == wave) solveTileK(k + 1u, bi.tileStart + tl, bi.ctStart + tl * (k + 1u), lane, slotMeta, slotNormal, slotMass, rows, imp, gVel);
        }
        __threadfence_block();   // one workgroup = one CU = one L1: workgroup scope is enough (an agent-scope fence costs ~3.5 us per lane here)
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Dataflow PGS sweep: ONE launch per solver iteration instead of one per colour.
//
// A colour launch costs ~8.5 us whatever its size (boundary + two dependent memory round trips + one
// wave's arithmetic + drain), and a sweep needs ~8 of them back to back: the solver is bound by the
// number of serial phases, not by bytes.  Here every tile of the sweep is in flight at once and waits
// only for ITS OWN bodies: gVel[2b] = (v, tag), gVel[2b+1] = (w, tag) where tag counts the updates the
// body has received this step.  The manifolds of a body have distinct colours, so "the colours used on
// the body below mine" (k_contact_init) says how many updates precede this manifold in a sweep; lane
// waits until both halves of the body carry tag = iteration * degree + base, solves, and publishes
// (v, w) with tag + 1.  The execution order is therefore exactly the sequential colour-major order the
// oracle replays — only the waiting is per body instead of per colour.
//
// Cross-CU visibility (MI355X_MICROARCH.md, "inter-workgroup visibility"): each half is ONE 16-byte
// `sc1` (agent-scope, write-through) store carrying its own tag and is read with `sc1` loads (L1
// bypass), so a reader that sees the tag sees the data of the same store: no fences, no separate flag.
// Forward progress: tile t depends only on tiles < t (lower colours) and workgroups are dispatched in
// index order, so the lowest unfinished tile is always resident and never waits on an undispatched one;
// every wait is bounded anyway (spin budget -> StepScalars::solveError -> MI_ERR_DEVICE, no hang).
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef MI_SC_LOAD
#define MI_SC_LOAD " sc1"    // cache-policy bits of the granule loads / stores (development experiments override them)
#endif
#ifndef MI_SC_STORE
#define MI_SC_STORE " sc1"
#endif
constexpr uint32_t kSpinBudget = 1u << 16;
#ifndef MI_FLOW_WAVES
#define MI_FLOW_WAVES 1   // resident waves per SIMD the flow kernel is compiled for (measured: 1 = 0.86 ms, 2 = 1.02 ms per 20 sweeps at 262144 bodies)
#endif

// single 16-byte granule: issue only (the next waiting asm block lands it), load + wait, store
__device__ __forceinline__ void issueGranuleSc1(const float4* p, f32x4& g) { asm volatile("global_load_dwordx4 %0, %1, off" MI_SC_LOAD : "=&v"(g) : "v"(p) : "memory"); }
__device__ __forceinline__ void landed(f32x4& g) { asm volatile("" : "+v"(g)); }   // orders every later use of g behind the waiting block
// the same into a register that already holds a value ("+v": the asm reads AND writes g, so a value merged from a divergent branch stays in ONE register — with a pure output
// the compiler may place a copy between the load's issue and the wait that lands it, and copy the old contents)
__device__ __forceinline__ void issueGranuleSc1Keep(const float4* p, f32x4& g) { asm volatile("global_load_dwordx4 %0, %1, off" MI_SC_LOAD : "+v"(g) : "v"(p) : "memory"); }
__device__ __forceinline__ void loadGranuleSc1(const float4* p, f32x4& g) { asm volatile("global_load_dwordx4 %0, %1, off" MI_SC_LOAD "\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory"); }
__device__ __forceinline__ void storeGranuleSc1(float4* p, f32x4 g) { asm volatile("global_store_dwordx4 %0, %1, off" MI_SC_STORE : : "v"(p), "v"(g) : "memory"); }

// Lane pairs (2i, 2i+1) move one body per instruction: the even lane touches granule 0 and the odd lane granule 1 of the
// SAME body, i.e. one contiguous, 32-byte-aligned transaction instead of two scattered 16-byte ones (scattered
// write-through stores are what bounds this kernel: 4 per manifold per sweep).  Pass 0 serves the even lane's body,
// pass 1 the odd lane's; a DPP quad swap hands each lane the half its partner moved for it.
// lane 2i <-> lane 2i+1 as a DPP quad permute [1,0,3,2]: one VALU move, no trip through the LDS crossbar (ds_bpermute) — these
// exchanges sit between a tile's body loads and its stores, i.e. on the dependency chain.  Both lanes of a pair are always active together.
__device__ __forceinline__ uint32_t swz1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float swz1(float v) { return __uint_as_float(swz1(__float_as_uint(v))); }
__device__ __forceinline__ f32x4 swz1(f32x4 g) { f32x4 r = {swz1(g.x), swz1(g.y), swz1(g.z), swz1(g.w)}; return r; }
__device__ __forceinline__ float4* swz1(float4* p) {
    unsigned long long v = (unsigned long long)p;
    uint32_t lo = swz1((uint32_t)v), hi = swz1((uint32_t)(v >> 32));
    return (float4*)(((unsigned long long)hi << 32) | lo);
}
struct PairBody {   // addresses this lane touches in pass 0 / pass 1 for one body slot (A or B)
    float4* q0; float4* q1;
    __device__ __forceinline__ PairBody(float4* mine, bool odd) {
        float4* partner = swz1(mine);
        q0 = (odd ? partner : mine) + (odd ? 1 : 0);
        q1 = (odd ? mine : partner) + (odd ? 1 : 0);
    }
};
// after both passes landed: r0 / r1 = what this lane loaded in pass 0 / 1 -> this lane's own (g0, g1)
__device__ __forceinline__ void pairGather(bool odd, f32x4 r0, f32x4 r1, f32x4& g0, f32x4& g1);   // (below)
// The lane-pair exchange of pairGather and of the publish in ONE instruction per word:  y0 = even lane ? x0 : the partner's x1,  y1 = odd lane ? x1 : the partner's x0
// (v_cndmask_b32 whose first source is DPP quad-permuted [1,0,3,2]).  The compiler's own code for `odd ? swz1(a) : b` is v_mov_b32_dpp + v_cndmask_b32_e64 — gfx9 has no
// VOP3 DPP, and it keeps the lane parity in an SGPR pair, not in VCC — i.e. three instructions per word where pairGather / storePair* need both directions; these sit between
// a tile's bodies arriving and its publish, where every instruction is ~4 cycles of the dependency chain.  Both lanes of a pair are always active together.
__device__ __forceinline__ void pairExchange(const f32x4 x0, const f32x4 x1, f32x4& y0, f32x4& y1) {
    float a0, a1, a2, a3, b0, b1, b2, b3;
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\ts_nop 1\n\t"
                 "v_cndmask_b32_dpp %0, %12, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %1, %13, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %2, %14, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %3, %15, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "s_not_b64 vcc, vcc\n\t"
                 "v_cndmask_b32_dpp %4, %8, %12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %5, %9, %13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %6, %10, %14, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %7, %11, %15, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                 : "v"(x0.x), "v"(x0.y), "v"(x0.z), "v"(x0.w), "v"(x1.x), "v"(x1.y), "v"(x1.z), "v"(x1.w) : "vcc", "scc");
    y0.x = a0; y0.y = a1; y0.z = a2; y0.w = a3; y1.x = b0; y1.y = b1; y1.z = b2; y1.w = b3;
}
__device__ __forceinline__ void pairGather(bool odd, f32x4 r0, f32x4 r1, f32x4& g0, f32x4& g1) {
    (void)odd; pairExchange(r0, r1, g0, g1);
}
__device__ __forceinline__ void loadPair4Sc1(const PairBody& A, const PairBody& B, f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1) {
    asm volatile("global_load_dwordx4 %0, %4, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %1, %5, off" MI_SC_LOAD "\n\t"
                 "global_load_dwordx4 %2, %6, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %3, %7, off" MI_SC_LOAD "\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(A.q0), "v"(A.q1), "v"(B.q0), "v"(B.q1) : "memory");
}
__device__ __forceinline__ void issuePair4Sc1(const PairBody& A, const PairBody& B, f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1) {   // no wait: waitVmcnt + landed follow
    asm volatile("global_load_dwordx4 %0, %4, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %1, %5, off" MI_SC_LOAD "\n\t"
                 "global_load_dwordx4 %2, %6, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %3, %7, off" MI_SC_LOAD
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(A.q0), "v"(A.q1), "v"(B.q0), "v"(B.q1) : "memory");
}
__device__ __forceinline__ void issuePair2Sc1(const PairBody& A, f32x4& a0, f32x4& a1) {   // no wait ("+v": lanes that do not take part keep their values)
    asm volatile("global_load_dwordx4 %0, %2, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %1, %3, off" MI_SC_LOAD
                 : "+v"(a0), "+v"(a1) : "v"(A.q0), "v"(A.q1) : "memory");
}
__device__ __forceinline__ void loadPair2Sc1(const PairBody& A, f32x4& a0, f32x4& a1) {
    asm volatile("global_load_dwordx4 %0, %2, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %1, %3, off" MI_SC_LOAD "\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1) : "v"(A.q0), "v"(A.q1) : "memory");
}
// publish one body slot: h0 / h1 = this lane's own granules, need = this lane's body is written at all
__device__ __forceinline__ void storePairSc1(const PairBody& X, bool odd, bool need, f32x4 h0, f32x4 h1) {
    f32x4 recv = swz1(odd ? h0 : h1);                 // even lane receives the odd lane's g0, odd lane the even lane's g1
    bool partnerNeed = swz1(need ? 1u : 0u) != 0u;
    f32x4 d0 = odd ? recv : h0, d1 = odd ? h1 : recv;
    if (odd ? partnerNeed : need) storeGranuleSc1(X.q0, d0);
    if (odd ? need : partnerNeed) storeGranuleSc1(X.q1, d1);
}
// the same with a choice per body: XCD-local bodies are published with plain stores (they stay in this XCD's L2)
__device__ __forceinline__ void storeGranulePlain(float4* p, f32x4 g) { asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(g) : "memory"); }
__device__ __forceinline__ void storePairXcd(const PairBody& X, bool odd, bool need, bool local, f32x4 h0, f32x4 h1) {
    f32x4 recv = swz1(odd ? h0 : h1);
    bool partnerNeed = swz1(need ? 1u : 0u) != 0u, partnerLocal = swz1(local ? 1u : 0u) != 0u;
    f32x4 d0 = odd ? recv : h0, d1 = odd ? h1 : recv;
    const bool n0 = odd ? partnerNeed : need, l0 = odd ? partnerLocal : local;     // pass 0 moves the even lane's body,
    const bool n1 = odd ? need : partnerNeed, l1 = odd ? local : partnerLocal;     // pass 1 the odd lane's
    if (n0 && l0) storeGranulePlain(X.q0, d0);
    if (n0 && !l0) storeGranuleSc1(X.q0, d0);
    if (n1 && l1) storeGranulePlain(X.q1, d1);
    if (n1 && !l1) storeGranuleSc1(X.q1, d1);
}

// the four stores of one body slot under precomputed EXEC masks (plain / write-through for pass 0, then for pass 1); the wave is fully active on entry and on exit
__device__ __forceinline__ void storePairMasked(float4* q0, float4* q1, f32x4 d0, f32x4 d1, unsigned long long plain0, unsigned long long sc0, unsigned long long plain1, unsigned long long sc1_) {
    asm volatile("s_mov_b64 exec, %4\n\tglobal_store_dwordx4 %0, %2, off\n\t"
                 "s_mov_b64 exec, %5\n\tglobal_store_dwordx4 %0, %2, off" MI_SC_STORE "\n\t"
                 "s_mov_b64 exec, %6\n\tglobal_store_dwordx4 %1, %3, off\n\t"
                 "s_mov_b64 exec, %7\n\tglobal_store_dwordx4 %1, %3, off" MI_SC_STORE "\n\t"
                 "s_mov_b64 exec, -1"
                 : : "v"(q0), "v"(q1), "v"(d0), "v"(d1), "s"(plain0), "s"(sc0), "s"(plain1), "s"(sc1_) : "memory");
}
// LDSIMP: the accumulated impulses live in LDS (`ldsImp`, [k][lane]) because the same wave runs this tile in every sweep
// (k_contact_solve_persist); otherwise they travel between sweeps as tagged granules in `imp`.
#ifdef MI_DBG_KNOCKOUT
// development (knock-out harness, tools/gpu_knockout.sh): the host launches k_contact_solve_persist a SECOND time per step on scratch copies of the velocity arrays with parts
// of a tile visit removed, to price them: bit 0 = no row stream at all (nothing is prefetched; the update runs on whatever the registers hold — the tag protocol does not
// depend on the values), bit 1 = every tile's rows come from contact-tile 0 (the same loads in the queue, served by the L2), bit 2 = no waiting for tags.
#define MI_KNOCK(bit) ((g_dbgKnockLocal >> (bit)) & 1u)
#else
#define MI_KNOCK(bit) 0u
#endif
#ifdef MI_DBG_TIMELINE
__device__ unsigned long long* g_dbgTimeline = nullptr;   // development: [wave][visit][8] wall-clock stamps of k_contact_solve_persist
__device__ __forceinline__ void dbgStamp(unsigned long long* rec, int i) { if (rec && threadIdx.x == 0) rec[i] = wall_clock64(); }
#define MI_STAMP(rec, i) dbgStamp(rec, i)
#else
#define MI_STAMP(rec, i) ((void)0)
#endif
// Hook of processTile: early() runs right after the body loads were issued and returns how many loads it issued itself (they
// may stay in flight across the first tag check); late(waited) runs once the tags are satisfied, waited = the tile had to poll.
struct NoHook { enum : bool { kPinRows = false }; unsigned long long* rec = nullptr; __device__ __forceinline__ uint32_t early() const { return 0u; } __device__ __forceinline__ void late(bool) const {}
                __device__ __forceinline__ bool knockNoWait() const { return false; } };
// wait until at most n of the newest vector-memory operations are outstanding (n = a count the caller issued itself)
__device__ __forceinline__ void waitVmcnt(uint32_t n) {
    switch (n) {
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
template <int CNT, bool LDSIMP, bool XCD = false, class Hook = NoHook>
__device__ __forceinline__ void processTile(uint32_t ctBase, uint32_t lane, uint32_t it, const uint4 meta, const float4 nf, const float2 mass, const ContactRows* c,
                                            float4* imp, float4* gVel, StepScalars* sc, float2* ldsImp, float4* gVelL = nullptr, Hook hook = Hook());
template <int CNT, bool LDSIMP = false>
__device__ __forceinline__ void flowTile(uint32_t tile, uint32_t ctBase, uint32_t lane, uint32_t it, const uint4* __restrict__ slotMeta,
                                         const float4* __restrict__ slotNormal, const float2* __restrict__ slotMass,
                                         const float4* __restrict__ rows, float4* imp, float4* gVel, StepScalars* sc, float2* ldsImp = nullptr) {
    const uint4 meta = slotMeta[(size_t)tile * 64u + lane];
    if (__ballot(meta.w != 0u) == 0ull) return;   // nothing of this tile is the tile solver's (manifolds of private joint islands: their island's workgroup solves them)
    const float4 nf = slotNormal[(size_t)tile * 64u + lane];
    const float2 mass = slotMass[(size_t)tile * 64u + lane];
    ContactRows c[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        const float4* __restrict__ row = rows + ((size_t)ctBase + k) * (kRows * 64u) + lane;
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) c[k].r[r] = row[r * 64u];
    }
    processTile<CNT, LDSIMP>(ctBase, lane, it, meta, nf, mass, c, imp, gVel, sc, ldsImp);
}
// The tile proper, from data already requested (flowTile) or prefetched (k_contact_solve_persist): wait for the bodies (and the
// impulse granules), solve, publish.
// The tile proper.  What shapes it is how few instructions sit between the
// arrival of a tile's bodies and its publish, the part of a visit that is on the dependency chain between tiles (~4 cycles per instruction at one wave per SIMD):
//   * which of the lane pair's four stores per body slot take place, and with which cache policy, is known from the slot's constants: four EXEC masks per body are
//     computed BEFORE the wait and the publish is four stores under `s_mov_b64 exec, mask` (was: the predicates recomputed and exchanged after the solve, ~12 per store);
//   * the lane-pair exchange is pairExchange (one v_cndmask_b32_dpp per word and direction), for the arriving bodies and for the publish;
//   * ONE gather / tag check site: every poll round gathers all lanes from the raw load registers (lanes that did not poll again find their old words there),
//     so the bodies the solve starts from are defined in one place and no copies merge two definitions at the loop's exit.
// The whole wave is active here (processTile is only reached through wave-uniform control flow).
template <int CNT, bool LDSIMP, bool XCD, class Hook>
__device__ __forceinline__ void processTile(uint32_t ctBase, uint32_t lane, uint32_t it, const uint4 meta, const float4 nf, const float2 mass, const ContactRows* c,
                                            float4* imp, float4* gVel, StepScalars* sc, float2* ldsImp, float4* gVelL, Hook hook) {
    const uint32_t bA = meta.x, bB = meta.y, pk = meta.z;
    const float imA = mass.x, imB = mass.y;
    const bool valid = meta.w != 0u;
#ifdef MI_DBG_ALLLOCAL
    const bool locA = XCD, locB = XCD;
#else
    const bool locA = XCD && (meta.w & 0x100u) != 0u, locB = XCD && (meta.w & 0x200u) != 0u;
#endif
    const bool live = valid && (imA != 0.f || imB != 0.f);
    const uint32_t degA = (pk >> 7) & 127u, degB = (pk >> 21) & 127u;
    const uint32_t expA = it * degA + (pk & 127u), expB = it * degB + ((pk >> 14) & 127u);
    const bool needA = valid && degA != 0u, needB = valid && degB != 0u;
    float4* pA = (locA ? gVelL : gVel) + 2 * (size_t)bA; float4* pB = (locB ? gVelL : gVel) + 2 * (size_t)bB;
    float4* pI = imp + (size_t)ctBase * 64u + lane;
    f32x4 ig[CNT], a0, a1, b0, b1;
    const bool odd = (lane & 1u) != 0u;
    const PairBody PA(pA, odd), PB(pB, odd);
    if (!LDSIMP) {
#pragma unroll
        for (int k = 0; k < CNT; ++k) issueGranuleSc1(pI + (size_t)k * 64u, ig[k]);
    }
    f32x4 ra0, ra1, rb0, rb1;   // raw load destinations (pass 0 / pass 1 of bodies A and B)
    issuePair4Sc1(PA, PB, ra0, ra1, rb0, rb1);
    MI_STAMP(hook.rec, 2);
    // (while the body loads are in flight) EXEC masks of the publish (pass 0 moves the even lane's body, pass 1 the odd lane's; XCD-local bodies are published with plain stores, the others write-through)
    unsigned long long mA[4], mB[4];
    {
        const unsigned long long E = 0x5555555555555555ull;
        const unsigned long long nA = __ballot(needA), lA = __ballot(locA), nB = __ballot(needB), lB = __ballot(locB);
        const unsigned long long nA0 = (nA & E) | ((nA & E) << 1), nA1 = (nA & ~E) | ((nA & ~E) >> 1), lA0 = (lA & E) | ((lA & E) << 1), lA1 = (lA & ~E) | ((lA & ~E) >> 1);
        const unsigned long long nB0 = (nB & E) | ((nB & E) << 1), nB1 = (nB & ~E) | ((nB & ~E) >> 1), lB0 = (lB & E) | ((lB & E) << 1), lB1 = (lB & ~E) | ((lB & ~E) >> 1);
        mA[0] = nA0 & lA0; mA[1] = nA0 & ~lA0; mA[2] = nA1 & lA1; mA[3] = nA1 & ~lA1;
        mB[0] = nB0 & lB0; mB[1] = nB0 & ~lB0; mB[2] = nB1 & lB1; mB[3] = nB1 & ~lB1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { asm volatile("" : "+s"(mA[k])); asm volatile("" : "+s"(mB[k])); }   // (pinned here: not recomputed behind the wait)
    }
    const uint32_t hookLoads = hook.early();   // (the persistent kernel: this tile's rows out of the prefetch registers, the next tile's requested)
    PkRows pkr[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) pkr[k] = packRows<std::remove_reference_t<Hook>::kPinRows>(c[k], nf, k != 0 && (meta.w & kMetaPerContactNormal) != 0u);   // (contact 0: the slot's normal IS its own)
    waitVmcnt(hookLoads);   // the hook's loads are younger than the body loads: they may stay in flight
    float2 imIn[CNT];   // accumulated impulses this tile starts from
    if (!LDSIMP) {
#pragma unroll
        for (int k = 0; k < CNT; ++k) landed(ig[k]);
    } else {
#pragma unroll
        for (int k = 0; k < CNT; ++k) imIn[k] = ldsImp[k * 64 + lane];
    }
    bool okA, okB, okI = true, polled = false;
    uint32_t budget = kSpinBudget;
    MI_STAMP(hook.rec, 3);
    for (;;) {
        landed(ra0); landed(ra1); landed(rb0); landed(rb1);
        pairExchange(ra0, ra1, a0, a1);
        pairExchange(rb0, rb1, b0, b1);
        okA = !needA || (__float_as_uint(a0.w) == expA && __float_as_uint(a1.w) == expA);
        okB = !needB || (__float_as_uint(b0.w) == expB && __float_as_uint(b1.w) == expB);
        if (!LDSIMP) {
            okI = true;
#pragma unroll
            for (int k = 0; k < CNT; ++k) okI = okI && (!live || __float_as_uint(ig[k].z) == it);
        }
#ifdef MI_DBG_KNOCKOUT
        if (hook.knockNoWait()) okA = okB = okI = true;   // development knock-out: timing without dependency waits (results are garbage)
#endif
        if (__ballot(!(okA && okB && okI)) == 0ull) break;
        polled = true;
        if (--budget == 0u) { sc->solveError = 1u; break; }
        // both lanes of a pair poll together (the exchange above needs both); tight polling measured fastest: only the pairs still waiting re-load, both bodies' polls
        // in flight together (one round trip per round, not two)
        const uint32_t partnerOkA = swz1(okA ? 1u : 0u), partnerOkB = swz1(okB ? 1u : 0u);
        const bool pollA = !okA || partnerOkA == 0u, pollB = !okB || partnerOkB == 0u;
        if (pollA) issuePair2Sc1(PA, ra0, ra1);
        if (pollB) issuePair2Sc1(PB, rb0, rb1);
        if (!LDSIMP && !okI) {
#pragma unroll
            for (int k = 0; k < CNT; ++k) issueGranuleSc1Keep(pI + (size_t)k * 64u, ig[k]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!LDSIMP) {
#pragma unroll
            for (int k = 0; k < CNT; ++k) landed(ig[k]);
        }
    }
    hook.late(polled);
    MI_STAMP(hook.rec, 4);
    P3 pv, pw;
    pv.x = pk2(a0.x, b0.x); pv.y = pk2(a0.y, b0.y); pv.z = pk2(a0.z, b0.z);
    pw.x = pk2(a1.x, b1.x); pw.y = pk2(a1.y, b1.y); pw.z = pk2(a1.z, b1.z);
    const f32x2 sMass = pk2(-imA, imB);
    float2 out[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        float2 im = LDSIMP ? imIn[k] : make_float2(ig[k].x, ig[k].y);
        solveOnePkRows(pkr[k], nf, im, sMass, pv, pw);
        out[k] = im;
    }
    // publish: bodies first (they are on the dependency chain), then the impulses; nothing to wait for afterwards
    {
        const float tA = __uint_as_float(expA + 1u), tB = __uint_as_float(expB + 1u);
        const f32x4 hA0 = {pv.x.x, pv.y.x, pv.z.x, tA}, hA1 = {pw.x.x, pw.y.x, pw.z.x, tA}, hB0 = {pv.x.y, pv.y.y, pv.z.y, tB}, hB1 = {pw.x.y, pw.y.y, pw.z.y, tB};
        MI_STAMP(hook.rec, 5);
        f32x4 dA0, dA1, dB0, dB1;
        pairExchange(hA0, hA1, dA0, dA1);
        pairExchange(hB0, hB1, dB0, dB1);
        storePairMasked(PA.q0, PA.q1, dA0, dA1, mA[0], mA[1], mA[2], mA[3]);
        storePairMasked(PB.q0, PB.q1, dB0, dB1, mB[0], mB[1], mB[2], mB[3]);
    }
    if (LDSIMP) {
#pragma unroll
        for (int k = 0; k < CNT; ++k) ldsImp[k * 64 + lane] = out[k];
    } else if (live) {
        float t = __uint_as_float(it + 1u);
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            f32x4 g = {out[k].x, out[k].y, t, 0.f};
            storeGranuleSc1(pI + (size_t)k * 64u, g);
        }
    }
}

// Block b runs sweep itBase + b / numTiles of tile b % numTiles (numTiles = StepScalars::totalTiles; schedule order, colour-major): with no joints between the
// sweeps ALL iterations are one launch, so the latency-bound small colours of sweep i overlap the bandwidth-bound large
// colours of sweep i + 1.  tileDesc[tile] = (first contact-tile, contacts per manifold).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, MI_FLOW_WAVES))) void k_contact_solve_flow(
    uint32_t itBase, uint32_t sweeps, const uint2* __restrict__ tileDesc, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal,
    const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* imp, float4* gVel, StepScalars* sc) {
    const uint32_t numTiles = sc->totalTiles;   // the grid is sized from an upper bound: surplus workgroups (all at the end) exit
    if (blockIdx.x >= numTiles * sweeps) return;
    const uint32_t it = itBase + blockIdx.x / numTiles, tile = blockIdx.x % numTiles, lane = threadIdx.x;
    const uint2 d = tileDesc[tile];
    switch (d.y) {
        case 1: flowTile<1>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, gVel, sc); break;
        case 2: flowTile<2>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, gVel, sc); break;
        case 3: flowTile<3>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, gVel, sc); break;
        default: flowTile<4>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, gVel, sc); break;
    }
}

// Persistent variant: `numWaves` workgroups (one per SIMD of the chip, all resident at once), workgroup w owns the tiles
// w, w + numWaves, ... in EVERY sweep and walks them in schedule order, sweep after sweep.  Because a tile never changes
// hands, its accumulated impulses stay in LDS: no impulse granules are read, polled or written (one 16-byte write-through
// store and one tagged load less per contact and sweep; 13 % fewer bytes).  Dependencies between tiles are the body tags as
// before.  Forward progress: every wave runs its tiles in ascending (sweep, tile) order and a tile only waits for smaller
// (sweep, tile) pairs, so the wave owning the smallest unfinished pair is never blocked — provided all workgroups are
// resident, which the host guarantees by launching at most one per SIMD (waits are bounded by the spin budget regardless).
#ifndef MI_PERSIST_WPE
#define MI_PERSIST_WPE 1
#endif
// a 16-byte load into four FIXED accumulator registers / reading them back (k_contact_solve_persist's row prefetch)
#define MI_ACC_LOAD(A0, A1, A2, A3, addr) asm volatile("global_load_dwordx4 a[" #A0 ":" #A3 "], %0, off" : : "v"(addr) : "memory", "a" #A0, "a" #A1, "a" #A2, "a" #A3)
#define MI_ACC_READ(dst, A0, A1, A2, A3) do { float x_, y_, z_, w_; \
    asm volatile("v_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 "\n\tv_accvgpr_read_b32 %2, a" #A2 "\n\tv_accvgpr_read_b32 %3, a" #A3 \
                 : "=v"(x_), "=v"(y_), "=v"(z_), "=v"(w_)); (dst) = make_float4(x_, y_, z_, w_); } while (0)

// RESIDENT ROWS (round 6).  The knock-out harness prices the row stream of this kernel at 14 % of its launch (no stream at all: 471 -> 405 us at the bench state;
// profiles/r06_knockout_solver_and_emit.txt): every wave's 24 x 1 KB row loads per tile sit in the same in-order memory queue as its body polls and publish stores.
// The accumulator registers a0..a143 are free (the ring is a160..a255), i.e. six contact-tiles ("positions") of 24 registers: the first tiles of a wave's list whose contacts fit
// are RESIDENT there — loaded once, at the top of the launch — and "prefetching" such a tile is 24 register moves per contact (v_accvgpr_mov_b32) into the ring,
// issued where the loads would have been: nothing enters the memory queue, and processTile still takes every tile's rows out of the ring, unchanged.
// Register numbers are literals ("n" operands): position q = a[24 q .. 24 q + 23], ring contact k = a[160 + 24 k .. ].  The compiler itself never allocates an
// accumulator register in this kernel (no spills: tests/test_capi_symbols.py reads the ISA); the ring's clobber lists make the descriptor cover a0..a255.
// Positions 6 and 7 live in the ARCHITECTURAL registers v208..v255 of the variants that fit into 208 allocatable VGPRs (slot data in LDS: amdgpu_num_vgpr keeps the
// compiler out of v208 and above): the same scheme with v_accvgpr_write_b32 as the move.  (Three such positions behind a cap of 184 were measured at 444 -> 433 us; the
// per-contact normals of terrain manifolds then took the nine registers that made 184 enough.)  The variants that need more registers keep six.
constexpr uint32_t kResidentAcc = 6, kResidentVgprBase = 208;
template <int DST, int SRC> __device__ __forceinline__ void accMov() { asm volatile("v_accvgpr_mov_b32 a[%0], a[%1]" : : "n"(DST), "n"(SRC)); }
template <int DST, int SRC> __device__ __forceinline__ void accFromV() { asm volatile("v_accvgpr_write_b32 a[%0], v[%1]" : : "n"(DST), "n"(SRC)); }
template <int Q, int K, int... R> __device__ __forceinline__ void accCopyContactImpl(std::integer_sequence<int, R...>) {
    if constexpr (Q < (int)kResidentAcc) (accMov<160 + 24 * K + R, 24 * Q + R>(), ...);
    else (accFromV<160 + 24 * K + R, (int)kResidentVgprBase + 24 * (Q - (int)kResidentAcc) + R>(), ...);
}
template <int NPOS, int Q, int K> __device__ __forceinline__ void accCopyContact() { if constexpr (Q < NPOS) accCopyContactImpl<Q, K>(std::make_integer_sequence<int, 24>()); }
template <int LO> __device__ __forceinline__ void accLoad4(const float4* p) { asm volatile("global_load_dwordx4 a[%0:%1], %2, off" : : "n"(LO), "n"(LO + 3), "v"(p) : "memory"); }
template <int LO> __device__ __forceinline__ void vgprLoad4(const float4* p) { asm volatile("global_load_dwordx4 v[%0:%1], %2, off" : : "n"(LO), "n"(LO + 3), "v"(p) : "memory", "v255"); }
template <int NPOS, int Q> __device__ __forceinline__ void accLoadContact(const float4* row) {   // the six rows of one contact of a tile -> position Q
    if constexpr (Q < (int)kResidentAcc) {
        accLoad4<24 * Q + 0>(row + 0u * 64u); accLoad4<24 * Q + 4>(row + 1u * 64u); accLoad4<24 * Q + 8>(row + 2u * 64u);
        accLoad4<24 * Q + 12>(row + 3u * 64u); accLoad4<24 * Q + 16>(row + 4u * 64u); accLoad4<24 * Q + 20>(row + 5u * 64u);
    } else if constexpr (Q < NPOS) {
        constexpr int B = (int)kResidentVgprBase + 24 * (Q - (int)kResidentAcc);
        vgprLoad4<B + 0>(row + 0u * 64u); vgprLoad4<B + 4>(row + 1u * 64u); vgprLoad4<B + 8>(row + 2u * 64u);
        vgprLoad4<B + 12>(row + 3u * 64u); vgprLoad4<B + 16>(row + 4u * 64u); vgprLoad4<B + 20>(row + 5u * 64u);
    }
}
template <int NPOS> __device__ __forceinline__ void accLoadResident(uint32_t q, const float4* row) {
    switch (q) {
        case 0: accLoadContact<NPOS, 0>(row); break; case 1: accLoadContact<NPOS, 1>(row); break; case 2: accLoadContact<NPOS, 2>(row); break;
        case 3: accLoadContact<NPOS, 3>(row); break; case 4: accLoadContact<NPOS, 4>(row); break; case 5: accLoadContact<NPOS, 5>(row); break;
        case 6: accLoadContact<NPOS, 6>(row); break; case 7: accLoadContact<NPOS, 7>(row); break; default: accLoadContact<NPOS, 8>(row); break;
    }
}
template <int NPOS, int Q> __device__ __forceinline__ void accCopyTile(uint32_t cnt) {   // resident positions Q .. Q + cnt - 1 -> ring contacts 0 .. cnt - 1
    accCopyContact<NPOS, Q, 0>();
    if (1u < cnt) accCopyContact<NPOS, Q + 1, 1>();
    if (2u < cnt) accCopyContact<NPOS, Q + 2, 2>();
    if (3u < cnt) accCopyContact<NPOS, Q + 3, 3>();
}
template <int NPOS> __device__ __forceinline__ void accCopyResident(uint32_t q, uint32_t cnt) {
    switch (q) {
        case 0: accCopyTile<NPOS, 0>(cnt); break; case 1: accCopyTile<NPOS, 1>(cnt); break; case 2: accCopyTile<NPOS, 2>(cnt); break;
        case 3: accCopyTile<NPOS, 3>(cnt); break; case 4: accCopyTile<NPOS, 4>(cnt); break; case 5: accCopyTile<NPOS, 5>(cnt); break;
        case 6: accCopyTile<NPOS, 6>(cnt); break; case 7: accCopyTile<NPOS, 7>(cnt); break; default: accCopyTile<NPOS, 8>(cnt); break;
    }
}

// METALDS = false (larger problems): only the impulses live in LDS (2060 B per slot instead of 4620); the constant slot data is
// prefetched from global memory together with the rows of the next tile.
// XCD = true (XCD-partitioned): workgroup w belongs to XCD w % 8 (verified against the hardware id: anything else is
// reported as solveError 3 and the host falls back) and owns entries w / 8, w / 8 + gridDim / 8, ... of THAT XCD's tile
// list (xcdTiles, ascending = schedule order).  Bodies only this XCD touches (bodyOwner has exactly this XCD's bit) are
// handed over through the XCD's L2 in the cached array gVelL; all others through memory in gVel as before.
// IMPLDS = false (piles beyond ~1.2 M manifolds): nothing per slot but a 20-byte descriptor stays in LDS; the accumulated
// impulses travel as tagged granules in `imp` exactly as in k_contact_solve_flow (no size limit left).
#define MI_PERSIST_PARAMS uint32_t sweeps, uint32_t maxSlots, const uint2* __restrict__ tileDesc, const uint4* slotMeta, const float4* __restrict__ slotNormal, \
    const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* gVel, StepScalars* sc, uint32_t xcdOnly, \
    const uint32_t* __restrict__ xcdTiles, uint32_t listCap, const unsigned long long* __restrict__ bodyOwner, float4* gVelL, uint4* slotMetaW, float4* imp, uint32_t xcdFault, \
    uint32_t resident /* 1: rows of the first tiles stay in a0..a143 (and v208..v255) */
#define MI_PERSIST_PASS sweeps, maxSlots, tileDesc, slotMeta, slotNormal, slotMass, rows, gVel, sc, xcdOnly, xcdTiles, listCap, bodyOwner, gVelL, slotMetaW, imp, xcdFault, resident
template <bool METALDS, bool XCD, bool IMPLDS>
__device__ __forceinline__ void persistSolveBody(MI_PERSIST_PARAMS) {
    // LDS per workgroup: [maxSlots] x { meta uint4[64], normal float4[64], mass float2[64] } (constant over the sweeps; METALDS only), then the
    // impulses float2[4 * maxSlots][64], then the per-slot (first contact-tile, contacts per manifold, impulse offset)
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    const size_t metaSlots = METALDS ? (size_t)maxSlots : 0;
    uint4* lMeta = reinterpret_cast<uint4*>(ldsRaw);
    float4* lNormal = reinterpret_cast<float4*>(lMeta + metaSlots * 64u);
    float2* lMass = reinterpret_cast<float2*>(lNormal + metaSlots * 64u);
    float2* lImp = lMass + metaSlots * 64u;
    uint32_t* lDesc = reinterpret_cast<uint32_t*>(lImp + (IMPLDS ? (size_t)maxSlots * 4u * 64u : 0));   // [maxSlots][3]
    if (xcdOnly && (blockIdx.x & 7u) != 0u) return;   // development experiment: only the workgroups of one XCD work
    const uint32_t lane = threadIdx.x;
    const uint32_t xcd = blockIdx.x & 7u;
    uint32_t numTiles = sc->totalTiles, numWaves = xcdOnly ? gridDim.x / 8u : gridDim.x, wid = xcdOnly ? blockIdx.x / 8u : blockIdx.x;
    if (XCD) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
        hw &= 15u;
        uint32_t seen = 0u;
        if (lane == 0) { seen = atomicCAS(&sc->xccOf[xcd], 0xFFFFFFFFu, hw); if (seen == 0xFFFFFFFFu) seen = hw; }
        seen = (uint32_t)__shfl((int)seen, 0, 64);
        if (seen != hw || (xcdFault && blockIdx.x == 9u)) { if (lane == 0) sc->solveError = 3u; return; }   // blockIdx % 8 does not identify the XCD on this device (xcdFault: test injection)
        numTiles = sc->totalTiles ? sc->xcdCount[xcd] : 0u; numWaves = gridDim.x / 8u; wid = blockIdx.x / 8u;
        xcdTiles += (size_t)xcd * listCap;
        if (numTiles > listCap) { if (lane == 0) sc->solveError = 2u; return; }
    }
    uint32_t* lTile = reinterpret_cast<uint32_t*>(lDesc + 3u * (size_t)maxSlots);   // [maxSlots] tile of every slot
    uint32_t* lCrit = lTile + maxSlots;                                             // [maxSlots] 1: the slot had to poll in the previous sweep
    constexpr int kResidentPositions = METALDS ? 8 : 6;
    uint32_t* lRes = lCrit + maxSlots;                                              // [maxSlots] first resident position of the slot's rows, or 0xFF: they stream
    uint32_t mySlots = 0, off = 0, resNext = 0;
    for (uint32_t li = wid; li < numTiles && mySlots < maxSlots; li += numWaves, ++mySlots) {
        const uint32_t tile = XCD ? xcdTiles[li] : li;
        const uint2 d = tileDesc[tile];
        const bool res = resident && resNext + d.y <= kResidentPositions;
        if (lane == 0) { lTile[mySlots] = tile; lCrit[mySlots] = 0u; lRes[mySlots] = res ? resNext : 0xFFu; }
        if (res) {   // (issued here, landed by the vmcnt(0) behind the loop)
            for (uint32_t k = 0; k < d.y; ++k) accLoadResident<kResidentPositions>(resNext + k, rows + ((size_t)d.x + k) * (kRows * 64u) + lane);
            resNext += d.y;
        }
        if (XCD) {   // which of this slot's two bodies are XCD-local -> bits 8 / 9 of meta.w (read back from LDS or global below)
            uint4 m = slotMeta[(size_t)tile * 64u + lane];
            const unsigned long long mine = 1ull << (8u * xcd);
            if (m.w != 0u) m.w |= (bodyOwner[m.x] == mine ? 0x100u : 0u) | (bodyOwner[m.y] == mine ? 0x200u : 0u);
            if (METALDS) lMeta[mySlots * 64u + lane] = m; else slotMetaW[(size_t)tile * 64u + lane] = m;
        }
        if (METALDS) {
            if (!XCD) lMeta[mySlots * 64u + lane] = slotMeta[(size_t)tile * 64u + lane];
            lNormal[mySlots * 64u + lane] = slotNormal[(size_t)tile * 64u + lane];
            lMass[mySlots * 64u + lane] = slotMass[(size_t)tile * 64u + lane];
        }
        if (lane == 0) { lDesc[3 * mySlots] = d.x; lDesc[3 * mySlots + 1] = d.y; lDesc[3 * mySlots + 2] = off; }
        if (IMPLDS) for (uint32_t k = 0; k < d.y; ++k) lImp[(size_t)(off + k) * 64u + lane] = make_float2(0.f, 0.f);
        off += d.y;
    }
    if (wid + (size_t)mySlots * numWaves < numTiles) { if (lane == 0) sc->solveError = 2u; return; }   // more tiles than the host sized LDS for
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!mySlots) return;
    // Software pipeline over (sweep, slot): the rows of the NEXT tile are requested while this tile waits for its bodies.
    // Loads retire in order, so the request order matters: this tile's body loads go first (even before its own rows are
    // taken out of the ACC registers: vmcnt(4)), the prefetch second, and the first tag check waits with vmcnt(number of
    // prefetch loads) — the bodies are back, the prefetch may still be in flight.
    // The prefetch is inline asm with its exact instruction count known, into FIXED accumulator registers a160..a255 that
    // the compiler never allocates (tests/test_capi_symbols.py checks the ISA for that); they are read back, again by
    // inline asm, after the explicit vmcnt(0) at the top of the next iteration.  (Compiler-allocated registers do not
    // work here: the allocator copies in-flight values around at loop boundaries.)
    uint4 nxMeta = make_uint4(0u, 0u, 0u, 0u); float4 nxNf = make_float4(0.f, 0.f, 0.f, 0.f); float2 nxMass = make_float2(0.f, 0.f);
#ifdef MI_DBG_KNOCKOUT
    const uint32_t g_dbgKnockLocal = __builtin_amdgcn_readfirstlane(g_dbgKnock);
#endif
    auto fetchRows = [&](uint32_t slot) __attribute__((always_inline)) -> uint32_t {
        uint32_t ct = lDesc[3 * slot]; const uint32_t cnt = lDesc[3 * slot + 1];
        if (MI_KNOCK(1)) ct = 0u;
        if (!METALDS) {
            const size_t at = (size_t)lTile[slot] * 64u + lane;
            nxMeta = XCD ? slotMetaW[at] : slotMeta[at]; nxNf = slotNormal[at]; nxMass = slotMass[at];
        }
        if (MI_KNOCK(0)) return 0u;
        if (const uint32_t q = lRes[slot]; q != 0xFFu) { accCopyResident<kResidentPositions>(q, cnt); return 0u; }   // resident: register moves, nothing enters the memory queue
        {
            const float4* row = rows + (size_t)ct * (kRows * 64u) + lane;   // row (k, r) of the tile at + (k * kRows + r) * 64
            if (0u < cnt) MI_ACC_LOAD(160, 161, 162, 163, row + 0u * 64u);
            if (0u < cnt) MI_ACC_LOAD(164, 165, 166, 167, row + 1u * 64u);
            if (0u < cnt) MI_ACC_LOAD(168, 169, 170, 171, row + 2u * 64u);
            if (0u < cnt) MI_ACC_LOAD(172, 173, 174, 175, row + 3u * 64u);
            if (0u < cnt) MI_ACC_LOAD(176, 177, 178, 179, row + 4u * 64u);
            if (0u < cnt) MI_ACC_LOAD(180, 181, 182, 183, row + 5u * 64u);
            if (1u < cnt) MI_ACC_LOAD(184, 185, 186, 187, row + 6u * 64u);
            if (1u < cnt) MI_ACC_LOAD(188, 189, 190, 191, row + 7u * 64u);
            if (1u < cnt) MI_ACC_LOAD(192, 193, 194, 195, row + 8u * 64u);
            if (1u < cnt) MI_ACC_LOAD(196, 197, 198, 199, row + 9u * 64u);
            if (1u < cnt) MI_ACC_LOAD(200, 201, 202, 203, row + 10u * 64u);
            if (1u < cnt) MI_ACC_LOAD(204, 205, 206, 207, row + 11u * 64u);
            if (2u < cnt) MI_ACC_LOAD(208, 209, 210, 211, row + 12u * 64u);
            if (2u < cnt) MI_ACC_LOAD(212, 213, 214, 215, row + 13u * 64u);
            if (2u < cnt) MI_ACC_LOAD(216, 217, 218, 219, row + 14u * 64u);
            if (2u < cnt) MI_ACC_LOAD(220, 221, 222, 223, row + 15u * 64u);
            if (2u < cnt) MI_ACC_LOAD(224, 225, 226, 227, row + 16u * 64u);
            if (2u < cnt) MI_ACC_LOAD(228, 229, 230, 231, row + 17u * 64u);
            if (3u < cnt) MI_ACC_LOAD(232, 233, 234, 235, row + 18u * 64u);
            if (3u < cnt) MI_ACC_LOAD(236, 237, 238, 239, row + 19u * 64u);
            if (3u < cnt) MI_ACC_LOAD(240, 241, 242, 243, row + 20u * 64u);
            if (3u < cnt) MI_ACC_LOAD(244, 245, 246, 247, row + 21u * 64u);
            if (3u < cnt) MI_ACC_LOAD(248, 249, 250, 251, row + 22u * 64u);
            if (3u < cnt) MI_ACC_LOAD(252, 253, 254, 255, row + 23u * 64u);
        }
        return cnt * kRows;
    };
    // this tile's rows out of the ACC registers; the body loads of the tile (4, issued just before) may still be in flight
    auto readRows = [&](ContactRows* cur, uint32_t cnt) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (0u < cnt) MI_ACC_READ(cur[0].r[0], 160, 161, 162, 163);
        if (0u < cnt) MI_ACC_READ(cur[0].r[1], 164, 165, 166, 167);
        if (0u < cnt) MI_ACC_READ(cur[0].r[2], 168, 169, 170, 171);
        if (0u < cnt) MI_ACC_READ(cur[0].r[3], 172, 173, 174, 175);
        if (0u < cnt) MI_ACC_READ(cur[0].r[4], 176, 177, 178, 179);
        if (0u < cnt) MI_ACC_READ(cur[0].r[5], 180, 181, 182, 183);
        if (1u < cnt) MI_ACC_READ(cur[1].r[0], 184, 185, 186, 187);
        if (1u < cnt) MI_ACC_READ(cur[1].r[1], 188, 189, 190, 191);
        if (1u < cnt) MI_ACC_READ(cur[1].r[2], 192, 193, 194, 195);
        if (1u < cnt) MI_ACC_READ(cur[1].r[3], 196, 197, 198, 199);
        if (1u < cnt) MI_ACC_READ(cur[1].r[4], 200, 201, 202, 203);
        if (1u < cnt) MI_ACC_READ(cur[1].r[5], 204, 205, 206, 207);
        if (2u < cnt) MI_ACC_READ(cur[2].r[0], 208, 209, 210, 211);
        if (2u < cnt) MI_ACC_READ(cur[2].r[1], 212, 213, 214, 215);
        if (2u < cnt) MI_ACC_READ(cur[2].r[2], 216, 217, 218, 219);
        if (2u < cnt) MI_ACC_READ(cur[2].r[3], 220, 221, 222, 223);
        if (2u < cnt) MI_ACC_READ(cur[2].r[4], 224, 225, 226, 227);
        if (2u < cnt) MI_ACC_READ(cur[2].r[5], 228, 229, 230, 231);
        if (3u < cnt) MI_ACC_READ(cur[3].r[0], 232, 233, 234, 235);
        if (3u < cnt) MI_ACC_READ(cur[3].r[1], 236, 237, 238, 239);
        if (3u < cnt) MI_ACC_READ(cur[3].r[2], 240, 241, 242, 243);
        if (3u < cnt) MI_ACC_READ(cur[3].r[3], 244, 245, 246, 247);
        if (3u < cnt) MI_ACC_READ(cur[3].r[4], 248, 249, 250, 251);
        if (3u < cnt) MI_ACC_READ(cur[3].r[5], 252, 253, 254, 255);
    };
    (void)fetchRows(0);
    for (uint32_t it = 0; it < sweeps; ++it)
        for (uint32_t slot = 0; slot < mySlots; ++slot) {
            ContactRows cur[4];
            unsigned long long* rec = nullptr;
#ifdef MI_DBG_TIMELINE
            if (g_dbgTimeline && it * mySlots + slot < 256u) rec = g_dbgTimeline + ((size_t)blockIdx.x * 256u + it * mySlots + slot) * 8u;
            if (rec && threadIdx.x == 0) { rec[6] = ((unsigned long long)it << 32) | slot; rec[7] = lTile[slot]; }
#endif
            MI_STAMP(rec, 0);
            const uint32_t ct = lDesc[3 * slot], cnt = lDesc[3 * slot + 1], io = lDesc[3 * slot + 2];
            uint4 meta; float4 nf; float2 mass;
            if (METALDS) { meta = lMeta[slot * 64u + lane]; nf = lNormal[slot * 64u + lane]; mass = lMass[slot * 64u + lane]; }
            else { meta = nxMeta; nf = nxNf; mass = nxMass; }
            const uint32_t nextSlot = slot + 1u < mySlots ? slot + 1u : 0u;
            const bool more = slot + 1u < mySlots || it + 1u < sweeps;
            // (a tile that had to poll in the previous sweep prefetching AFTER its wait, so that its polls do not queue behind the prefetch, was measured slower —
            // 0.68 vs 0.63 ms, round 3: the rows arriving late costs more)
            struct Prefetch {
                enum : bool { kPinRows = true };
                decltype(fetchRows)& fetch; decltype(readRows)& read; ContactRows* cur; uint32_t cnt; uint32_t* crit; uint32_t next; bool more; unsigned long long* rec; bool noWait;
                __device__ __forceinline__ bool knockNoWait() const { return noWait; }
                __device__ __forceinline__ uint32_t early() { read(cur, cnt); MI_STAMP(rec, 1); return more ? fetch(next) : 0u; }
                __device__ __forceinline__ void late(bool waited) { if (threadIdx.x == 0) *crit = waited ? 1u : 0u; }
            } prefetch{fetchRows, readRows, cur, cnt, &lCrit[slot], nextSlot, more, rec, MI_KNOCK(2) != 0u};
            float2* li = lImp + (size_t)io * 64u;
            // (the contact count of THIS tile as a literal in each case: the row read-back's `if (k < cnt)` guards fold, and no row register is "defined on some paths only" —
            // such values were kept alive around the loop: 61 register copies at the top of every visit)
            switch (cnt) {
                case 1: prefetch.cnt = 1u; processTile<1, IMPLDS, XCD, Prefetch&>(ct, lane, it, meta, nf, mass, cur, imp, gVel, sc, li, gVelL, prefetch); break;
                case 2: prefetch.cnt = 2u; processTile<2, IMPLDS, XCD, Prefetch&>(ct, lane, it, meta, nf, mass, cur, imp, gVel, sc, li, gVelL, prefetch); break;
                case 3: prefetch.cnt = 3u; processTile<3, IMPLDS, XCD, Prefetch&>(ct, lane, it, meta, nf, mass, cur, imp, gVel, sc, li, gVelL, prefetch); break;
                default: prefetch.cnt = 4u; processTile<4, IMPLDS, XCD, Prefetch&>(ct, lane, it, meta, nf, mass, cur, imp, gVel, sc, li, gVelL, prefetch); break;
            }
        }
}

// The kernels proper.  The two variants with the slot data in LDS fit into 208 architectural registers: theirs are capped there (amdgpu_num_vgpr takes no template
// argument, hence explicit specialisations) and v208..v255 hold two more resident positions.
#define MI_PERSIST_KERNEL_ATTRS __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, MI_PERSIST_WPE)))
template <bool METALDS, bool XCD, bool IMPLDS>
MI_PERSIST_KERNEL_ATTRS void k_contact_solve_persist(MI_PERSIST_PARAMS) { persistSolveBody<METALDS, XCD, IMPLDS>(MI_PERSIST_PASS); }
template <> MI_PERSIST_KERNEL_ATTRS __attribute__((amdgpu_num_vgpr(208))) void k_contact_solve_persist<true, true, true>(MI_PERSIST_PARAMS) { persistSolveBody<true, true, true>(MI_PERSIST_PASS); }
template <> MI_PERSIST_KERNEL_ATTRS __attribute__((amdgpu_num_vgpr(208))) void k_contact_solve_persist<true, false, true>(MI_PERSIST_PARAMS) { persistSolveBody<true, false, true>(MI_PERSIST_PASS); }
static_assert(kResidentVgprBase == 208, "the cap of the specialisations above");

// Overflow colour (a body with > 64 incident manifolds): sequential, one lane, slots in ascending pair-key order.
__global__ void k_contact_solve_serial(BinInfo bi, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal,
                                       const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* __restrict__ imp, float4* __restrict__ gVel) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (uint32_t j = 0; j < bi.count; ++j) {
        uint32_t tile = bi.tileStart + (j >> 6), lane = j & 63u, ctBase = bi.ctStart + (j >> 6) * 4u;
        uint4 meta = slotMeta[(size_t)tile * 64u + lane];
        solveTileK(meta.w & 7u, tile, ctBase, lane, slotMeta, slotNormal, slotMass, rows, imp, gVel);
        __threadfence();
    }
}



// ------------------------------------------------------------------------------------------------------------------------------
// Sharded world (multi-GPU): every rank holds the WHOLE scene (same body / collider indices everywhere — priorities,
// pair keys and colour history mean the same thing on every rank) and simulates one tile of an x-z grid: the bodies whose centre
// of gravity lies in its tile (OWNED: it integrates them) plus those within `margin` of the tile (GHOSTS: they take part in its
// collision detection and solve, their new state comes from their owner).  Tiles on the rim of the grid extend to infinity.
// Ownership follows the bodies: it is recomputed from the positions at the start of every step (migration needs no bookkeeping).
// Tiles are cut by BORDERS, uniform when sharding is enabled and moved by mi_world_shard_set_borders (load balance).  A rank only ever tests its own
// tile and its <= 8 neighbours', so it carries the borders of tile columns / rows (mine - 1) .. (mine + 2): bx[1] <= x < bx[2] is this rank's column,
// -inf / +inf beyond the rim of the grid (rim tiles are unbounded outwards).
struct ShardParams {
    float bx[4], bz[4], margin;
    uint32_t tilesX, tilesZ, myTile;
    uint32_t numPeers; uint32_t peers[8];     // neighbouring tiles (|dx| <= 1, |dz| <= 1), ascending tile index
};
__device__ __forceinline__ bool shardOwns(const ShardParams& sp, float x, float z) { return x >= sp.bx[1] && x < sp.bx[2] && z >= sp.bz[1] && z < sp.bz[2]; }
// is (x, z) inside tile `t` (this rank's or a neighbour's) grown by the margin?
__device__ __forceinline__ bool shardInExtended(const ShardParams& sp, uint32_t t, float x, float z) {
    const uint32_t dx = t % sp.tilesX + 1u - sp.myTile % sp.tilesX, dz = t / sp.tilesX + 1u - sp.myTile / sp.tilesX;   // 0 .. 2
    return x >= sp.bx[dx] - sp.margin && x < sp.bx[dx + 1u] + sp.margin && z >= sp.bz[dz] - sp.margin && z < sp.bz[dz + 1u] + sp.margin;
}
__device__ __forceinline__ V3 shardCog(float4 pos, float4 rot, float4 cogInvMass) { return xyz(pos) + rotate(toQ(rot), xyz(cogInvMass)); }

// start of a step: 1 = owned, 2 = ghost, 0 = not simulated here.  One workgroup per RECENT body block (see shardBlockRecent): the others hold nothing this rank
// simulates, and their flags already say so in both flag arrays.
__global__ __launch_bounds__(256) void k_shard_classify(uint32_t nb, ShardParams sp, const float4* __restrict__ bPos, const float4* __restrict__ bRot,
                                                        const float4* __restrict__ bCogInvMass, uint8_t* __restrict__ bodyActive, Shards* sh,
                                                        const uint32_t* __restrict__ root /* lowest body index of the body's articulated island: the island is classified as ONE */,
                                                        const uint8_t* __restrict__ known /* 1 = this rank's copy of the body is current (owned in the last step, or a record arrived) */,
                                                        const uint8_t* __restrict__ bodyActivePrev, uint32_t* __restrict__ blockStamp, uint8_t* __restrict__ blockLive, uint32_t step) {
    __shared__ uint32_t cnt;
    forLiveBlocks(blockIdx.x, gridDim.x, (nb + 255u) / 256u, [&](uint32_t blk) { return shardBlockRecent(blockStamp, blk, step); }, [&](uint32_t blk) {
        const uint32_t i = blk * 256u + threadIdx.x;
        bool owned = false;
        uint8_t flag = 0u;
        if (i < nb) {
            const uint32_t r = root[i];
            const bool k = known[r] != 0u;      // a copy that is not current says nothing about where the body is (it may lie in a tile that has since grown)
            if (k) {                            // (most bodies of a many-tile scene are not known here: 5 bytes read for them instead of 53)
                const V3 c = shardCog(bPos[r], bRot[r], bCogInvMass[r]);
                owned = shardOwns(sp, c.x, c.z);
                flag = owned ? 1u : shardInExtended(sp, sp.myTile, c.x, c.z) ? 2u : 0u;
            }
            bodyActive[i] = flag;
        }
        // counted per workgroup into one of kShards lines (summed by k_integrate_velocities): a same-address atomic per wave was 45 us of a 2 M-body scene
        __syncthreads();   // (the previous block's count has been added)
        if (threadIdx.x == 0) cnt = 0;
        const int anyNow = __syncthreads_or(flag != 0u ? 1 : 0);
        const int anyPrev = __syncthreads_or((i < nb && bodyActivePrev[i] != 0u) ? 1 : 0);
        const unsigned long long m = __ballot(owned);
        if (m && (threadIdx.x & 63u) == 0u) atomicAdd(&cnt, (uint32_t)__popcll(m));
        __syncthreads();
        if (threadIdx.x == 0) {
            if (cnt) atomicAdd(&sh->c[blk & (kShards - 1u)].owned[0], cnt);
            if (blockStamp) { if (anyNow) blockStamp[blk] = step; blockLive[blk] = (anyNow || anyPrev) ? 1u : 0u; }
        }
    });
}
// owner rule for the counts: a manifold belongs to the rank that owns its first dynamic body (A unless A has no inverse mass / is the static dummy)
__global__ __launch_bounds__(256) void k_shard_count(uint32_t nb, const uint2* __restrict__ manBodies, const uint2* __restrict__ manInfo,
                                                     const float4* __restrict__ bCogInvMass, const uint8_t* __restrict__ bodyActive, const StepScalars* __restrict__ sc, Shards* sh) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine = 0, contacts = 0;
    if (m < sc->numManifolds) {
        const uint2 b = manBodies[m];
        const uint32_t first = (b.x < nb && bCogInvMass[b.x].w != 0.f) ? b.x : b.y;
        if (first < nb && bodyActive[first] == 1u) { mine = 1u; contacts = manInfo[m].x & 7u; }
    }
    for (int off = 32; off >= 1; off >>= 1) { mine += __shfl_xor(mine, off, 64); contacts += __shfl_xor(contacts, off, 64); }
    if ((threadIdx.x & 63u) == 0u && mine) { atomicAdd(&cnt[0], mine); atomicAdd(&cnt[1], contacts); }
    __syncthreads();
    if (threadIdx.x < 2 && cnt[threadIdx.x]) atomicAdd(&sh->c[blockIdx.x & (kShards - 1u)].owned[1 + threadIdx.x], cnt[threadIdx.x]);   // (was: two same-address atomics per wave, 0.14 ms)
}
// after a valid step (body buffers already swapped: bPos = new state, bPosOld = state the step started from): the records this rank
// owes neighbour `slot` — every body it OWNED this step whose old or new centre of gravity lies in that neighbour's extended tile
// (old: so that the neighbour learns the body has left).  Record = (body index, 13 floats); record 0 of the buffer = (count, ...).
constexpr uint32_t kShardRecordFloats = 14;
struct ShardBufs { float* p[8]; };   // one message buffer per neighbour slot
__device__ __forceinline__ void shardPackWave(uint32_t i, bool owned, const ShardParams& sp, const ShardParams& spNext, uint32_t bordersPending,
                                              const float4* __restrict__ bPos, const float4* __restrict__ bRot, const float4* __restrict__ bLinVel,
                                              const float4* __restrict__ bAngVel, const float4* __restrict__ bPosOld, const float4* __restrict__ bRotOld,
                                              const float4* __restrict__ bCogInvMass, const ShardBufs& out, uint32_t capacity, StepScalars* sc,
                                              const uint32_t* __restrict__ root) {
    V3 cn(0.f, 0.f, 0.f), co(0.f, 0.f, 0.f);
    float4 p = make_float4(0, 0, 0, 0), q = p, v = p, w = p;
    if (owned) {
        const uint32_t r = root[i];
        const float4 cm = bCogInvMass[r];
        cn = shardCog(bPos[r], bRot[r], cm); co = shardCog(bPosOld[r], bRotOld[r], cm);
        p = bPos[i]; q = bRot[i]; v = bLinVel[i]; w = bAngVel[i];
    }
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t slot = 0; slot < sp.numPeers; ++slot) {
        // borders about to move: also what the neighbour simulates under the NEW borders (it classifies with them from the next step on)
        const bool want = owned && (shardInExtended(sp, sp.peers[slot], cn.x, cn.z) || shardInExtended(sp, sp.peers[slot], co.x, co.z) ||
                                    (bordersPending && shardInExtended(spNext, sp.peers[slot], cn.x, cn.z)));
        const unsigned long long mask = __ballot(want);
        if (!mask) continue;
        const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&sc->shardSent[slot], (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, (int)leader, 64);
        if (!want) continue;
        const uint32_t r = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (r >= capacity) continue;                                 // the count still grows: the host sees the overflow
        float* o = out.p[slot] + (size_t)(r + 1u) * kShardRecordFloats;
        o[0] = __uint_as_float(i); o[1] = p.x; o[2] = p.y; o[3] = p.z; o[4] = q.x; o[5] = q.y; o[6] = q.z; o[7] = q.w;
        o[8] = v.x; o[9] = v.y; o[10] = v.z; o[11] = w.x; o[12] = w.y; o[13] = w.z;
    }
}
// one launch for all neighbours (the record counts start at zero: k_reset_scalars)
__global__ __launch_bounds__(256) void k_shard_pack(uint32_t nb, ShardParams sp, ShardParams spNext, uint32_t bordersPending, uint8_t* __restrict__ known, const uint8_t* __restrict__ bodyActive,
                                                    const float4* __restrict__ bPos, const float4* __restrict__ bRot, const float4* __restrict__ bLinVel,
                                                    const float4* __restrict__ bAngVel, const float4* __restrict__ bPosOld, const float4* __restrict__ bRotOld,
                                                    const float4* __restrict__ bCogInvMass, ShardBufs out, uint32_t capacity, StepScalars* sc,
                                                    const uint32_t* __restrict__ root, const uint32_t* __restrict__ blockStamp /* recent body blocks only (all when null): the others hold neither an owned body nor a copy marked current */, uint32_t step) {
    forLiveBlocks(blockIdx.x, gridDim.x, (nb + 255u) / 256u, [&](uint32_t blk) { return shardBlockRecent(blockStamp, blk, step); }, [&](uint32_t blk) {
        const uint32_t i = blk * 256u + threadIdx.x;
        const bool owned = i < nb && bodyActive[i] == 1u;
        if (i < nb) known[i] = owned ? 1u : 0u;   // what this rank knows from here on: the bodies it owned; the records about to arrive add the neighbours' (k_shard_unpack)
        if (__ballot(owned)) shardPackWave(i, owned, sp, spNext, bordersPending, bPos, bRot, bLinVel, bAngVel, bPosOld, bRotOld, bCogInvMass, out, capacity, sc, root);
    });
}
// The next step's sweep axis of a sharded world, from centre statistics summed over all ranks (or, before / without that sum, this rank's own)
__global__ void k_shard_axis(const unsigned long long* __restrict__ sums9, uint32_t nc, uint32_t* __restrict__ axisDev) { if (threadIdx.x == 0 && blockIdx.x == 0) *axisDev = axisFromSums(sums9, nc); }
// (a done-ticket in k_shard_pack instead of this launch: 8 192 same-address atomics in a 2 M-body scene, ~90 us)
constexpr uint32_t kShardFlagsMagic = 0x5A4D0000u;   // header word 1 = magic | the sender's message-size policy (bit 0: adaptive sizes): ranks that disagree about it would post sends and receives of different lengths
__global__ void k_shard_pack_headers(uint32_t numPeers, const StepScalars* __restrict__ sc, ShardBufs out,
                                     uint32_t nc, uint32_t* __restrict__ axisOwn /* caller's transport: the next sweep axis from this rank's own sums (k_shard_axis), or null */, uint32_t flags,
                                     uint32_t* sentHost /* pinned host memory: the eight record counts, for the host's overflow check (was a copy of its own: a 4 us copy kernel) */) {
    if (threadIdx.x < numPeers) { out.p[threadIdx.x][0] = __uint_as_float(sc->shardSent[threadIdx.x]); out.p[threadIdx.x][1] = __uint_as_float(kShardFlagsMagic | flags); }
    if (sentHost && threadIdx.x < 8u) { sentHost[threadIdx.x] = sc->shardSent[threadIdx.x]; __threadfence_system(); }
    if (axisOwn && threadIdx.x == 63) *axisOwn = axisFromSums(sc->axisSums, nc);
}
// blockIdx.y = neighbour slot (a body has one owner: the messages never touch the same body)
struct ShardCaps { uint32_t c[8]; };   // records each received message can hold as it travelled (library transport: sized from the previous exchange)
__global__ __launch_bounds__(256) void k_shard_unpack(uint32_t nb, ShardBufs in, uint32_t capacity, float4* __restrict__ bPos, float4* __restrict__ bRot,
                                                      float4* __restrict__ bLinVel, float4* __restrict__ bAngVel, uint8_t* __restrict__ known,
                                                      ShardCaps caps = ShardCaps{{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}}, uint32_t* __restrict__ recvCounts = nullptr,
                                                      uint32_t myFlags = 0u /* library transport: this rank's message-size policy, held against the sender's (header word 1) */,
                                                      uint32_t* __restrict__ blockStamp = nullptr /* the body blocks that received a record become recent (for the step numbered `stampStep`) */, uint32_t stampStep = 0u) {
    const float* msg = in.p[blockIdx.y];
    const uint32_t sent = __float_as_uint(msg[0]), cap = min(capacity, caps.c[blockIdx.y]);
    if (recvCounts && blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t theirs = __float_as_uint(msg[1]);
        recvCounts[blockIdx.y] = (theirs & 0xFFFF0000u) == kShardFlagsMagic && (theirs & 0xFFFFu) != myFlags ? 0xFFFFFFFEu   // the neighbour sizes its messages by another rule (MI_SHARD_ADAPTIVE differs between the ranks)
                                 : sent > cap && sent <= capacity ? 0xFFFFFFFFu : sent;   // (more than travelled: the tail is missing — reported, never silent)
    }
    const uint32_t count = min(sent, cap);
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const float* s = msg + (size_t)(r + 1u) * kShardRecordFloats;
    const uint32_t b = __float_as_uint(s[0]);
    if (b >= nb) return;
    bPos[b] = make_float4(s[1], s[2], s[3], 0.f); bRot[b] = make_float4(s[4], s[5], s[6], s[7]);
    bLinVel[b] = make_float4(s[8], s[9], s[10], 0.f); bAngVel[b] = make_float4(s[11], s[12], s[13], 0.f);
    known[b] = 1u;
    if (blockStamp) blockStamp[b >> 8] = stampStep;
}
// ---- exact seam (include/mi_shard.h "Exact seam")
// Which tile border is v within the margin of?  b4 = the borders around this rank's column (ShardParams::bx / bz), mine = its index; 0 = none,
// else 1 + the border's index.  Same comparisons as shardInExtended; tiles are at least two margins wide (at most one border per axis).
__device__ __forceinline__ uint32_t seamNear(const float* b4, uint32_t mine, float m, float v) {
    const uint32_t c = v < b4[1] ? 0u : v < b4[2] ? 1u : 2u;      // the column v lies in, relative to mine - 1 (only bodies this rank simulates are asked about)
    const uint32_t t = mine + c;                                   // = (that column's index) + 1
    if (v < b4[c] + m) return t - 1u;                              // its lower border (index t - 2): id t - 1   (-inf at the rim: never)
    if (v >= b4[c + 1u] - m) return t;                             // its upper border (index t - 1): id t
    return 0u;
}
__device__ __forceinline__ uint32_t seamBorderOf(const ShardParams& sp, float x, float z) {
    return seamNear(sp.bx, sp.myTile % sp.tilesX, sp.margin, x) | (seamNear(sp.bz, sp.myTile / sp.tilesX, sp.margin, z) << 16);
}
// sharded world in exact mode: after k_shard_classify, for the bodies this rank simulates
__global__ __launch_bounds__(256) void k_seam_classify_shard(uint32_t nb, ShardParams sp, const float4* __restrict__ bPos, const float4* __restrict__ bRot, const float4* __restrict__ bCogInvMass,
                                                             const uint32_t* __restrict__ root, const uint8_t* __restrict__ bodyActive, uint32_t* __restrict__ seamId) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    uint32_t id = 0u;
    if (bodyActive[i]) { const uint32_t r = root[i]; const V3 c = shardCog(bPos[r], bRot[r], bCogInvMass[r]); id = seamBorderOf(sp, c.x, c.z); }
    seamId[i] = id;
}
// single world that was told a tiling (mi_world_set_seam_tiling): all borders, linear search (a handful of tiles per axis)
__global__ __launch_bounds__(256) void k_seam_classify_tiling(uint32_t nb, const float* __restrict__ bx, uint32_t nx, const float* __restrict__ bz, uint32_t nz, float m,
                                                              const float4* __restrict__ bPos, const float4* __restrict__ bRot, const float4* __restrict__ bCogInvMass,
                                                              const uint32_t* __restrict__ root, uint32_t* __restrict__ seamId) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const uint32_t r = root[i];
    const V3 c = shardCog(bPos[r], bRot[r], bCogInvMass[r]);
    auto near = [&](const float* b, uint32_t n, float v) -> uint32_t {
        uint32_t t = 0; while (t < n && v >= b[t]) ++t;
        if (t > 0u && v < b[t - 1u] + m) return t;
        if (t < n && v >= b[t] - m) return t + 1u;
        return 0u;
    };
    seamId[i] = near(bx, nx, c.x) | (near(bz, nz, c.z) << 16);
}
// after the colouring: seam manifolds, the colours they use, violations (a manifold outside the seam class that touches a ghost: the margin does not cover
// the reach of a contact; a seam manifold that found no colour among the kSeamColors: it would be solved after the interior)
__global__ __launch_bounds__(256) void k_seam_stats(StepScalars* sc, const uint4* __restrict__ colWork, const uint32_t* __restrict__ color, const uint8_t* __restrict__ bodyActive /* or null */) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    bool seam = false, bad = false; uint32_t c = 0u;
    if (m < sc->numManifolds) {
        const uint4 w = colWork[m];
        const bool dynA = (w.x >> 31) != 0u, dynB = (w.y >> 31) != 0u;
        seam = (w.x & 0x40000000u) != 0u;
        c = color[m];
        if (seam) bad = c >= kSeamColors;
        else if (bodyActive) bad = (dynA && bodyActive[w.x & 0x3FFFFFFFu] == 2u) || (dynB && bodyActive[w.y & 0x7FFFFFFFu] == 2u);
    }
    const unsigned long long ms = __ballot(seam), mb = __ballot(bad);
    if (seam && c < kSeamColors) atomicMax(&sc->seamStats[1], c + 1u);
    if ((threadIdx.x & 63u) == 0u) { if (ms) atomicAdd(&sc->seamStats[0], (uint32_t)__popcll(ms)); if (mb) atomicAdd(&sc->seamStats[2], (uint32_t)__popcll(mb)); }
}
// Per-sweep hand-over.  At the start of the step: for every neighbour slot the bodies this rank OWNS that the neighbour holds as ghosts (their centres, as
// classified, lie in its extended tile); after every sweep their velocities are gathered into one fixed-size message per neighbour (record 0 = count; a
// record = body index, linear velocity, angular velocity, pad) and the neighbours' are scattered into the ghost copies — the version tags in .w stay.
constexpr uint32_t kSweepRecordFloats = 8;
struct SweepLists { uint32_t* p[8]; };
__global__ __launch_bounds__(256) void k_seam_sweep_list(uint32_t nb, ShardParams sp, const uint8_t* __restrict__ bodyActive, const float4* __restrict__ bPos, const float4* __restrict__ bRot,
                                                         const float4* __restrict__ bCogInvMass, const uint32_t* __restrict__ root, SweepLists lists, uint32_t capacity, uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool owned = i < nb && bodyActive[i] == 1u;
    if (!__ballot(owned)) return;
    V3 c(0.f, 0.f, 0.f);
    if (owned) { const uint32_t r = root[i]; c = shardCog(bPos[r], bRot[r], bCogInvMass[r]); }
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t slot = 0; slot < sp.numPeers; ++slot) {
        const bool want = owned && shardInExtended(sp, sp.peers[slot], c.x, c.z);
        const unsigned long long mask = __ballot(want);
        if (!mask) continue;
        const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&counts[slot], (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, (int)leader, 64);
        if (!want) continue;
        const uint32_t r = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (r < capacity) lists.p[slot][r] = i;                       // (the count still grows: the host sees the overflow)
    }
}
// blockIdx.y = neighbour slot
__global__ __launch_bounds__(256) void k_seam_sweep_pack(SweepLists lists, const uint32_t* __restrict__ counts, uint32_t capacity, const float4* __restrict__ gVel, ShardBufs out) {
    const uint32_t slot = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = counts[slot];
    float* msg = out.p[slot];
    if (r == 0u) msg[0] = __uint_as_float(n);
    if (r >= min(n, capacity)) return;
    const uint32_t b = lists.p[slot][r];
    const float4 v = gVel[2 * (size_t)b], w = gVel[2 * (size_t)b + 1];
    float4* o = reinterpret_cast<float4*>(msg + (size_t)(r + 1u) * kSweepRecordFloats);
    o[0] = make_float4(__uint_as_float(b), v.x, v.y, v.z); o[1] = make_float4(w.x, w.y, w.z, 0.f);
}
__global__ __launch_bounds__(256) void k_seam_sweep_unpack(uint32_t nb, ShardBufs in, uint32_t capacity, const uint8_t* __restrict__ bodyActive, float4* __restrict__ gVel,
                                                           ShardCaps caps = ShardCaps{{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}}) {
    const float* msg = in.p[blockIdx.y];
    const uint32_t count = min(__float_as_uint(msg[0]), min(capacity, caps.c[blockIdx.y]));
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const float4* s = reinterpret_cast<const float4*>(msg + (size_t)(r + 1u) * kSweepRecordFloats);
    const float4 a = s[0], w = s[1];
    const uint32_t b = __float_as_uint(a.x);
    if (b >= nb || bodyActive[b] != 2u) return;                        // only a ghost's copy is replaced
    float4* g = gVel + 2 * (size_t)b;
    g[0] = make_float4(a.y, a.z, a.w, g[0].w); g[1] = make_float4(w.x, w.y, w.z, g[1].w);
}
// owned bodies per bin of [lo, hi) along x (axis 0) or z (1), by the centre their island was classified with; the end bins take what lies outside
__global__ __launch_bounds__(256) void k_shard_histogram(uint32_t nb, uint32_t axis, float lo, float scale, uint32_t bins, const uint8_t* __restrict__ bodyActive,
                                                         const float4* __restrict__ bPos, const float4* __restrict__ bRot, const float4* __restrict__ bCogInvMass,
                                                         const uint32_t* __restrict__ root, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb || bodyActive[i] != 1u) return;
    const uint32_t r = root[i];
    const V3 c = shardCog(bPos[r], bRot[r], bCogInvMass[r]);
    const float v = ((axis ? c.z : c.x) - lo) * scale;
    const uint32_t bin = v >= (float)bins ? bins - 1u : v > 0.f ? (uint32_t)(int)v : 0u;
    atomicAdd(&out[bin], 1u);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Device-wide exclusive prefix sum, ONE launch: chained scan with decoupled look-back.
//   * a workgroup takes its tile number from a ticket counter (so a tile's predecessors have always started), scans its
//     kScanTile items in registers / LDS and publishes first its AGGREGATE, then — after looking back over its predecessors'
//     records (one wave, 64 records at a time, until an inclusive prefix is found) — its INCLUSIVE PREFIX;
//   * a record is ONE 64-bit word (tag << 32 | 32-bit sum), tag = generation << 2 | state (1 aggregate, 2 inclusive), written
//     and read with single relaxed agent-scope accesses: a reader that sees the tag sees the sum of the same store — no fences
//     (a release / acquire pair at agent scope would write back / invalidate the L2 around every record);
//   * records are never reset: a reader ignores tags of older generations, and the ticket counter only ever grows
//     (`state[0]` = tickets handed out before this launch, `state[1]` = generation; the host clears everything long before the 30 generation bits wrap).
// T = uint32_t (W = 1) or a 64-bit word holding two independent 32-bit sums side by side (W = 2: the narrow phase's packed
// (manifold flag, contact count); both totals stay below 2^32, so the halves never carry into each other).
constexpr uint32_t kScanThreads = 256;
// items per lane: the long 64-bit scan (one item per collision pair) takes 16 — half the tiles, half the look-back chain (16.3 -> 13.7 us
// at 750 k pairs); the short 32-bit ones (cell histogram, schedule bins) are faster with 8 (6 vs 9 us)
template <typename T> struct ScanItems { static constexpr uint32_t N = sizeof(T) == 8 ? 16u : 8u; static constexpr uint32_t Tile = kScanThreads * N; };
__device__ __forceinline__ void scanPublish(unsigned long long* rec, uint32_t sum, uint32_t tag) {
    __hip_atomic_store(rec, ((unsigned long long)tag << 32) | (unsigned long long)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> struct ScanWords { static constexpr uint32_t W = sizeof(T) / 4; };
template <typename T>
__global__ __launch_bounds__(kScanThreads) void k_exclusive_scan(T* __restrict__ in, T* __restrict__ out, uint32_t n, unsigned long long* records,
                                                                 uint32_t* ticket, uint32_t* state /* [0] tickets handed out before this launch, [1] generation: kept ON THE DEVICE so that
                                                                 the launch has the same arguments every step (a captured HIP graph replays it) */, uint32_t zeroInput /* histograms: leave the input cleared for its next use */) {
    constexpr uint32_t W = ScanWords<T>::W;
    constexpr uint32_t kScanItems = ScanItems<T>::N, kScanTile = ScanItems<T>::Tile;
    __shared__ uint32_t sTile;
    __shared__ T sWave[kScanThreads / 64];
    __shared__ T sPrefix;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __shared__ uint32_t sGen;
    if (tid == 0) {
        // the state is read BEFORE the ticket is taken (the ticket address depends on it), and only the workgroup holding the launch's
        // LAST ticket advances it — by then every other workgroup of the launch has taken its ticket, i.e. has read the state
        const uint32_t ticketBase = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t g = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t t = atomicAdd(ticket + ((ticketBase ^ g) >> 31 >> 1), 1u) - ticketBase;
        sTile = t; sGen = g;
        if (t == gridDim.x - 1u) {
            __hip_atomic_store(&state[0], ticketBase + gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[1], (g + 1u) & 0x3FFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const uint32_t tile = sTile, gen = sGen;
    // striped loads (thread t takes items t, t + 256, ...: every load instruction reads one contiguous span), then blocked through LDS?  Not needed:
    // a prefix sum only needs each THREAD's items to be consecutive in the order it sums them, so thread t owns the kScanItems consecutive items
    // starting at base and reads them as 16-byte vectors
    const uint32_t base = tile * kScanTile + tid * kScanItems;
    T v[kScanItems];
    #pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) v[k] = base + k < n ? in[base + k] : T(0);
    if (zeroInput) {
#pragma unroll
        for (uint32_t k = 0; k < kScanItems; ++k) if (base + k < n) in[base + k] = T(0);
    }
    T local = 0;
    #pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) { T x = v[k]; v[k] = local; local += x; }      // exclusive within the thread
    T incl = local;                                                                         // inclusive across the wave
    #pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { T o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63u) sWave[wave] = incl;
    __syncthreads();
    T waveOff = 0, aggregate = 0;
    #pragma unroll
    for (uint32_t w = 0; w < kScanThreads / 64; ++w) { if (w < wave) waveOff += sWave[w]; aggregate += sWave[w]; }
    if (wave == 0) {
        T prefix = 0;
        const uint32_t tagAgg = (gen << 2) | 1u, tagInc = (gen << 2) | 2u;
        auto word = [](T x, uint32_t w) -> uint32_t { return (uint32_t)((unsigned long long)x >> (32u * w)); };
        if (tile == 0) {
            if (lane < W) scanPublish(&records[lane], word(aggregate, lane), tagInc);
        } else {
            if (lane < W) scanPublish(&records[(size_t)tile * W + lane], word(aggregate, lane), tagAgg);
            int32_t look = (int32_t)tile - 1;
            while (true) {                                   // 64 predecessors per round, nearest first
                const int32_t idx = look - (int32_t)lane;
                T val = 0; uint32_t state = idx < 0 ? 3u : 0u;                 // 3: before the first tile (contributes nothing, ends the search)
                while (state == 0u) {
                    unsigned long long r0 = __hip_atomic_load(&records[(size_t)idx * W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long r1 = W == 2 ? __hip_atomic_load(&records[(size_t)idx * W + (W - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : r0;
                    const uint32_t t0 = (uint32_t)(r0 >> 32), t1 = (uint32_t)(r1 >> 32);
                    if (t0 == t1 && (t0 == tagAgg || t0 == tagInc)) {      // both halves written by the same publication of this generation
                        state = t0 & 3u;
                        val = W == 2 ? (T)(((unsigned long long)(uint32_t)r1 << 32) | (unsigned long long)(uint32_t)r0) : (T)(uint32_t)r0;
                    } else __builtin_amdgcn_s_sleep(1);
                }
                const unsigned long long done = __ballot(state >= 2u);       // lanes holding an inclusive prefix (or the start of the array)
                const uint32_t first = done ? (uint32_t)__ffsll((long long)done) - 1u : 64u;
                T contrib = lane <= first ? val : T(0);                        // everything nearer than (and including) the first inclusive record
                #pragma unroll
                for (uint32_t d = 32; d >= 1; d >>= 1) contrib += __shfl_xor(contrib, d, 64);
                prefix += contrib;
                if (done) break;
                look -= 64;
            }
            if (lane < W) scanPublish(&records[(size_t)tile * W + lane], word(prefix + aggregate, lane), tagInc);
        }
        if (lane == 0) sPrefix = prefix;
    }
    __syncthreads();
    const T off = sPrefix + waveOff + (incl - local);
    #pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) if (base + k < n) out[base + k] = off + v[k];
}

// ------------------------------------------------------------------------------------------------ poses for the caller
// The transforms the caller reads after a step (transform_component of every entity with a rigid body), produced where the bodies live:
// one lane per entity, out as [n][3] positions followed by [n][4] rotations — the layout of mi_world_get_transforms — so that ONE
// device-to-host copy of 28 B per entity follows instead of 2-4 arrays of 16 B per body and a host pass over them.
// lerpT < 0: transform = physics_transform1 (physics.cpp:1408-1411); else lerp(transform0, transform1, t), nlerp on the rotation
// (physics.cpp:1392-1406, src/core/math.h:673-682) — the same expressions as the host path (download()), bit for bit.
// Entities without a rigid body keep their host-side transform: their rows are left alone here and filled in by the host.
__global__ __launch_bounds__(256) void k_entity_poses(uint32_t n, const int* __restrict__ entBody, const float4* __restrict__ pos, const float4* __restrict__ rot,
                                                      const float4* __restrict__ pos0, const float4* __restrict__ rot0, float lerpT, float* __restrict__ outP, float* __restrict__ outR,
                                                      const float4* __restrict__ lin, const float4* __restrict__ ang, float* __restrict__ outL, float* __restrict__ outA /* [n][3] each, or null: the velocities ride along */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = entBody[i];
    if (b < 0) return;
    if (outL) {
        const float4 l = lin[b], a = ang[b];
        outL[3 * (size_t)i] = l.x; outL[3 * (size_t)i + 1] = l.y; outL[3 * (size_t)i + 2] = l.z;
        outA[3 * (size_t)i] = a.x; outA[3 * (size_t)i + 1] = a.y; outA[3 * (size_t)i + 2] = a.z;
    }
    const float4 p1 = pos[b], r1 = rot[b];
    V3 ps(p1.x, p1.y, p1.z); Q4 rt(r1.x, r1.y, r1.z, r1.w);
    if (lerpT >= 0.f) {
        const float4 p0 = pos0[b], r0 = rot0[b]; const float t = lerpT;
        ps = lerp(V3(p0.x, p0.y, p0.z), ps, t);
        rt = normalize(Q4(r0.x + t * (r1.x - r0.x), r0.y + t * (r1.y - r0.y), r0.z + t * (r1.z - r0.z), r0.w + t * (r1.w - r0.w)));
    }
    outP[3 * (size_t)i] = ps.x; outP[3 * (size_t)i + 1] = ps.y; outP[3 * (size_t)i + 2] = ps.z;
    *reinterpret_cast<float4*>(outR + 4 * (size_t)i) = make_float4(rt.x, rt.y, rt.z, rt.w);
}

}  // namespace mi
