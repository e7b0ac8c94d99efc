// kernels_broad.hpp — world-space colliders and the broad phase (grid + counting sort, pair kernels, bucket partition, pair statistics).
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

namespace mi {

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

// Cell size >= `cell` (grown in steps of 1.3 x) and the grid dimensions over [lo, hi] that fit a table of cellCap cells.  The loop runs until they fit — a world
// whose bodies have flown apart (found by tools/gpu_fuzz.py: unstable joints carried a body 2e11 m away; 64 growth steps, a factor 2e7, were not enough and the cell
// histogram was indexed beyond its end) ends with few, huge cells, and one whose bounds are not numbers with a single cell; the indices are clamped to the dimensions anyway.
__device__ inline float gridFit(const float lo[3], const float hi[3], float cell, uint32_t cellCap, uint32_t dims[3]) {
    for (int it = 0; it < 512; ++it) {
        double cells = 1.0;
        for (int a = 0; a < 3; ++a) {
            float q = (hi[a] - lo[a]) / cell;
            if (!(q >= 0.f)) q = 0.f;                                   // (NaN bounds)
            const uint32_t d = (uint32_t)fminr(q, 4.0e9f) + 2u;
            dims[a] = d; cells *= (double)d;
        }
        if (cells <= (double)(cellCap - 1)) return cell;
        cell *= 1.3f;                                                   // (reaches +inf after ~340 steps at the latest: q = 0, two cells per axis)
    }
    dims[0] = dims[1] = dims[2] = 1u;
    return cell;
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
    cell = gridFit(lo, hi, cell, cellCap, g->dims);   // the host sized the cell table (histogram + scan) for cellCap cells
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
#ifndef MI_PAIR_FETCH
#define MI_PAIR_FETCH 4
#endif
constexpr uint32_t kPairFetch = MI_PAIR_FETCH;   // candidates of a column fetched per loop trip (2 loads each)
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
                if (MI_BP_KNOCK(16)) e = s;
                float4 amn = sMin[i], amx = sMax[i];
                uint32_t ci = vals[i];
                for (uint32_t j = s; j < e; j += kPairFetch) {
                    float4 bmn[kPairFetch], bmx[kPairFetch];
#pragma unroll
                    for (uint32_t u = 0; u < kPairFetch; ++u) { uint32_t jj = min(j + u, e - 1u); bmn[u] = sMin[jj]; bmx[u] = sMax[jj]; }
#pragma unroll
                    for (uint32_t u = 0; u < kPairFetch; ++u) {
                        if (j + u >= e || !aabbOverlap(amn, amx, bmn[u], bmx[u])) continue;
                        ++overlaps;
                        if (MI_BP_KNOCK(17)) continue;
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
    if (MI_BP_KNOCK(18)) continue;
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
            cell = gridFit(lo, hi, cell, cellCapNext, gridNext->dims);
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

}  // namespace mi
