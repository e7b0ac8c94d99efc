// kernels_schedule.hpp — contact schedule: Jones-Plassmann colouring, colouring tail, bins.
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

namespace mi {

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
constexpr uint32_t kTailSpinBudget = 1u << 18, kTailTimedOut = 0xFFFFFFFFu;   // (~15-50 ms of polling; StepScalars::tailRounds = kTailTimedOut tells the host why the step is void)
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
            uint32_t budget = kTailSpinBudget;
            while (__hip_atomic_load(&flags[kTailDone], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) { __builtin_amdgcn_s_sleep(8); if (--budget == 0u) { sc->specOverflow = 1u; sc->tailRounds = kTailTimedOut; break; } }
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
            uint32_t budget = kTailSpinBudget;
            while (__hip_atomic_load(&flags[kTailBar], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < arrivals) { __builtin_amdgcn_s_sleep(2); if (--budget == 0u) { failed = true; break; } }
            sMore = failed ? 2u : __hip_atomic_load(&flags[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        __threadfence();
        const uint32_t more = sMore;
        __syncthreads();
        if (more == 2u) { if (threadIdx.x == 0) { sc->specOverflow = 1u; sc->tailRounds = kTailTimedOut; __hip_atomic_store(&flags[kTailDone], r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } return; }   // (a participant never arrived — the barrier assumes its <= 256 workgroups are co-resident,
                                                                                                                          // which a shared or smaller device need not grant: the step is void, re-run synchronously, and the host
                                                                                                                          // goes back to a margin of enqueued rounds for the next 256 steps)
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

}  // namespace mi
