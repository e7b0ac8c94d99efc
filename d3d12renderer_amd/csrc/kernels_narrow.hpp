// kernels_narrow.hpp — narrow phase of the primitive pairs, colour history, collision events, manifold emission.
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

namespace mi {

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
// chain may have holes where its old neighbours vanished.  A lookup therefore cannot stop at an empty slot.  What bounds it instead is a HINT per home slot: hint[h] = the
// largest displacement any key with home h was ever inserted at (one word per slot in a side array that lives as long as the tables keep their size; raised by the
// compare-and-swap insertions only, i.e. by the few NEW manifolds of a step).  A lookup loads the home slot and its hint together: a hit at home (four entries of five) and
// a miss with hint 0 (nearly every new manifold) are ONE round trip, as they were when lookups stopped at empty slots.  (A single global bound was tried first: 24 at the
// bench state, and every wave with a new manifold in it walked 25 dependent probes — the probe 10 -> 25 us.)
// When the two tables differ in size (the manifold count crossed a power of two, a synchronous re-run) positions do not carry over: every entry is inserted afresh, into a
// fresh hint array.
struct alignas(16) HistSlot { unsigned long long key; unsigned long long val; };
struct HistHit { uint32_t colour, slot; };
__device__ __forceinline__ HistHit tableFind(const HistSlot* __restrict__ tab, uint32_t mask, uint64_t key, const uint32_t* __restrict__ hint) {
    uint32_t s = tableSlot(key, mask);
    const ulonglong2 e0 = *reinterpret_cast<const ulonglong2*>(tab + s);
    const uint32_t bound = hint[s];
    if (e0.x == key) return HistHit{(uint32_t)e0.y, s};
    for (uint32_t n = 1; n <= bound && n <= mask; ++n) {
        s = (s + 1u) & mask;
        const ulonglong2 e = *reinterpret_cast<const ulonglong2*>(tab + s);
        if (e.x == key) return HistHit{(uint32_t)e.y, s};
    }
    return HistHit{kUncolored, 0u};
}
__device__ __forceinline__ uint32_t tableLookup(const HistSlot* __restrict__ tab, uint32_t mask, uint64_t key, const uint32_t* __restrict__ hint) { return tableFind(tab, mask, key, hint).colour; }
__device__ __forceinline__ void tableInsert(HistSlot* __restrict__ tab, uint32_t mask, uint64_t key, uint32_t val, uint32_t* __restrict__ hint) {
    const uint32_t home = tableSlot(key, mask);
    for (uint32_t s = home, n = 0; n <= mask; s = (s + 1u) & mask, ++n) {
        const unsigned long long old = atomicCAS(&tab[s].key, 0ull, (unsigned long long)key);
        if (old == 0ull || old == key) {   // (the colour is read in the NEXT step only; the same key again — a schedule built twice in a synchronous step — overwrites)
            tab[s].val = val;
            if (n && n > __hip_atomic_load(&hint[home], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&hint[home], n);
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
                                                    DeviceEvent* __restrict__ events, const uint32_t* __restrict__ curHint /* the probe hints of curTab (tableFind) */) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long key = s <= prevMask ? prevTab[s].key : 0ull;
    bool want = key != 0ull && ((key - 1ull) & ((1ull << kIndexBits) - 1ull)) < kHeightmapVirtualBase;   // heightmap contacts raise no events
    if (want && prevMask == curMask && curTab[s].key == key) want = false;                                  // it kept its colour, hence its slot
    want = want && tableLookup(curTab, curMask, key, curHint) == kUncolored;
    uint32_t slot = waveAppendSlot(want, &sc->numEvents);
    if (!want) return;
    if (slot >= cap) { sc->specOverflow = 1u; return; }
    DeviceEvent e{};
    e.type = 1u; e.colliderA = (uint32_t)((key - 1ull) >> kIndexBits); e.colliderB = (uint32_t)((key - 1ull) & ((1ull << kIndexBits) - 1ull));
    events[slot] = e;
}

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
                                                        HistSlot* __restrict__ nextTab, uint32_t nextMask, uint8_t* __restrict__ manKept, const uint32_t* __restrict__ prevHint, uint32_t* __restrict__ nextHint /* the tables' probe hints (tableFind; one array while the tables keep their size) */,
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
    if (prevTab && !MI_EMIT_KNOCK(10)) hit = tableFind(prevTab, prevMask, hk, prevHint);
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
        else if (found && prevMask == nextMask && prevHint == nextHint) { ulonglong2 e; e.x = hk; e.y = c; *reinterpret_cast<ulonglong2*>(nextTab + hit.slot) = e; }   // (same size AND the same hints: the slot means the same)
        else tableInsert(nextTab, nextMask, hk, c, nextHint);
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

}  // namespace mi
