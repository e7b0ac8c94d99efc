// kernels_contacts.hpp — contact constraints: tiles, k_schedule_finish, k_contact_init, the PGS update, the per-colour solver kernels.
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

namespace mi {

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
                                                         HistSlot* __restrict__ tab, uint32_t tabMask, const uint8_t* __restrict__ manKept, uint32_t* __restrict__ histHint,
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
                    tableInsert(tab, tabMask, historyKey(nc, (uint32_t)((pk >> 29) & 0x1FFFFFFFull), (uint32_t)(pk & 0x1FFFFFFFull)), c, histHint);
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
#ifdef MI_EXP_PIN_NONVOLATILE   // (tools/exp/pinned_pairs_repro.sh: variants of the unexplained wrong-result case of round 5)
__device__ __forceinline__ void pinPair(f32x2& p) { asm("" : "+v"(p)); }
#else
__device__ __forceinline__ void pinPair(f32x2& p) { asm volatile("" : "+v"(p)); }
#endif
#ifndef MI_EXP_PIN_MASK
#define MI_EXP_PIN_MASK 0x1FF
#endif
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
    if (PIN) {
        constexpr uint32_t M = MI_EXP_PIN_MASK;
        if (M & 1u) pinPair(k.rx); if (M & 2u) pinPair(k.ry); if (M & 4u) pinPair(k.rz); if (M & 8u) pinPair(k.Tx); if (M & 16u) pinPair(k.Ty); if (M & 32u) pinPair(k.Tz);
        if (M & 64u) pinPair(k.Nx); if (M & 128u) pinPair(k.Ny); if (M & 256u) pinPair(k.Nz);
    }
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

}  // namespace mi
