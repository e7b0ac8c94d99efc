// blocks.hpp — the contact solver as SPATIAL BLOCKS held in LDS (one workgroup = one CU = one block of the scene).
//
// k_contact_solve_persist hands every body over through L2 / memory: ~2.5 us per hop under load, 10 colours x 20 sweeps hops in a row.
// Here the scene is cut into blocks the way the multi-GPU shards cut it (include/mi_shard.h), one level down:
//   * a block = a run of cells of a 64 x 64 grid over the two longest axes of the broad-phase grid, in Morton order, holding ~1/nbe of
//     the manifolds (k_block_keys, k_block_place).  Every dynamic body has ONE home block: the block of its cell;
//   * a manifold belongs to the home block(s) of its dynamic bodies.  Both bodies at home in one block: an INTERIOR manifold of that
//     block.  Two home blocks: a BOUNDARY manifold, entered into BOTH blocks' lists and solved by both, redundantly and bit-identically
//     (same rows, same operations); each block applies the result to its own body and drops the other's;
//   * the home bodies of a block live in LDS for the whole solve (32 bytes: (v, tag), (w, tag)), handed from wave to wave of the
//     workgroup by polling LDS — a hop is an LDS round trip, not a memory one;
//   * the other body of a boundary manifold is a GHOST: its state before the manifold's turn comes from its home block, which EXPORTS
//     a body right after the update that precedes a boundary manifold on it (one tagged write-through 32-byte record into that
//     manifold's mailbox slot), so a boundary manifold costs ONE memory hop (export -> poll), not two (there and back);
//   * inside a block the manifolds are sorted by (colour, boundary, contacts descending) and cut into DENSE 64-lane tiles; a tile that
//     straddles a colour is run as consecutive lane-range passes ("groups"), colour by colour, so dependencies inside a tile are
//     honoured by program order.  Rows are streamed from memory once per sweep as before (prefetched a pass ahead into fixed ACC
//     registers), the accumulated impulses stay in LDS.
// The per-body update sequences are exactly the canonical ones (colour-major inside a sweep): results are bit-identical to every
// other solver path and to the oracle; which block / wave / lane runs a manifold is invisible to the body-version dataflow.
#pragma once
#include "kernels.hpp"

namespace mi {

constexpr uint32_t kBlockCells = 4096;        // 64 x 64 cells, Morton order
constexpr uint32_t kBlockMaxPer = 16;         // list entries per thread of k_block_sched (256 threads): <= 4096 entries = 64 tiles per block
constexpr uint32_t kBlockSortBins = (kOverflowColor + 1) * 8;   // (colour, boundary, 4 - contacts)
__device__ __forceinline__ uint32_t spread6(uint32_t v) {   // 6 bits -> every other bit of 12
    v &= 63u; v = (v | (v << 4)) & 0x30Fu; v = (v | (v << 2)) & 0x333u; v = (v | (v << 1)) & 0x555u; return v;
}
__device__ __forceinline__ uint32_t blockCellOf(const GridParams& g, float4 p, uint32_t au, uint32_t av, float su, float sv, float ou, float ov) {
    const float cu = au == 0u ? p.x : au == 1u ? p.y : p.z, cv = av == 0u ? p.x : av == 1u ? p.y : p.z;
    const uint32_t iu = (uint32_t)fminf(fmaxf((cu - ou) * su, 0.f), 63.f), iv = (uint32_t)fminf(fmaxf((cv - ov) * sv, 0.f), 63.f);
    return spread6(iu) | (spread6(iv) << 1);
}

// Cell keys of the manifolds: key A = cell of the first dynamic body (the histogram and the sort run on it), key B = cell of the second
// one when both are dynamic.  keys[m] = keyA | keyB << 12 | both << 24; ranks[m] = arrival rank inside key A.
__global__ __launch_bounds__(256) void k_block_keys(uint32_t n, StepScalars* __restrict__ sc, const GridParams* __restrict__ gp, const uint2* __restrict__ manBodies,
                                                    const float4* __restrict__ gPos, uint32_t* __restrict__ keys, uint32_t* __restrict__ ranks, uint32_t* __restrict__ keyCount,
                                                    uint32_t nbe, uint32_t* __restrict__ extraCount, BlockState* __restrict__ bs) {
    __shared__ uint32_t hist[kBlockCells];
    for (uint32_t k = threadIdx.x; k < kBlockCells; k += 256) hist[k] = 0u;
    if (blockIdx.x == 0) {
        for (uint32_t k = threadIdx.x; k < nbe; k += 256) extraCount[k] = 0u;
        if (threadIdx.x == 0) { bs->need = 0u; bs->needExtra = 0u; bs->needBodies = 0u; bs->needPasses = 0u; bs->needImp = 0u; bs->overflow = 0u; bs->ghostLanes = 0u; }
        for (uint32_t k = threadIdx.x; k < kColorBins + 4u; k += 256) sc->binStart[k] = 0u;   // (k_block_sched adds the blocks' counts)
    }
    __syncthreads();
    const uint32_t nm = min(n, sc->numManifolds);
    const GridParams g = *gp;
    // the two longest axes of the grid
    const uint32_t d0 = g.dims[0], d1 = g.dims[1], d2 = g.dims[2];
    const uint32_t amin = (d1 <= d0 && d1 <= d2) ? 1u : (d0 <= d2 ? 0u : 2u);   // (ties: y first — piles are flat)
    const uint32_t au = amin == 0u ? 1u : 0u, av = amin == 2u ? 1u : 2u;
    const uint32_t du = au == 0u ? d0 : d1, dv = av == 2u ? d2 : d1;
    const float ou = au == 0u ? g.origin[0] : g.origin[1], ov = av == 2u ? g.origin[2] : g.origin[1];
    const float su = g.invCell * (64.f / (float)du), sv = g.invCell * (64.f / (float)dv);
    uint32_t key[kKeyItems / 256], local[kKeyItems / 256];
#pragma unroll
    for (uint32_t i = 0; i < kKeyItems / 256; ++i) {
        const uint32_t m = blockIdx.x * kKeyItems + i * 256 + threadIdx.x;
        key[i] = 0xFFFFFFFFu;
        if (m < nm) {
            const uint2 b = manBodies[m];
            const float4 pa = gPos[b.x], pb = gPos[b.y];
            const bool dynA = pa.w != 0.f, dynB = pb.w != 0.f;
            const uint32_t ka = blockCellOf(g, dynA ? pa : pb, au, av, su, sv, ou, ov);
            uint32_t k = ka;
            if (dynA && dynB) k |= (blockCellOf(g, pb, au, av, su, sv, ou, ov) << 12) | (1u << 24);
            key[i] = k;
            local[i] = atomicAdd(&hist[ka], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kBlockCells; k += 256) { uint32_t c = hist[k]; if (c) hist[k] = atomicAdd(&keyCount[k], c); }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kKeyItems / 256; ++i) {
        const uint32_t m = blockIdx.x * kKeyItems + i * 256 + threadIdx.x;
        if (key[i] != 0xFFFFFFFFu) { keys[m] = key[i]; ranks[m] = hist[key[i] & 0xFFFu] + local[i]; }
    }
}

// Cells -> blocks (every workgroup derives the table itself from the 4096 counts), the manifolds in cell order (perm), and the
// boundary manifolds entered into the OTHER home block's extra list.
__global__ __launch_bounds__(256) void k_block_place(uint32_t n, const StepScalars* __restrict__ sc, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ranks,
                                                     const uint32_t* __restrict__ keyCount, uint32_t* __restrict__ perm, uint32_t nbe, uint32_t* __restrict__ blockStart /* [nbe + 1] */,
                                                     uint16_t* __restrict__ cellBlock /* [4096] */, uint32_t* __restrict__ extra, uint32_t extraCap, uint32_t* __restrict__ extraCount) {
    __shared__ uint32_t lower[kBlockCells];
    __shared__ uint16_t blk[kBlockCells];
    __shared__ uint32_t part[256];
    constexpr uint32_t per = kBlockCells / 256;
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
    const uint32_t total = part[255];
    const uint32_t perBlock = max(1u, (total + nbe - 1u) / nbe);
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t c = threadIdx.x * per + k;
        lower[c] = run;
        blk[c] = (uint16_t)min(nbe - 1u, run / perBlock);   // (the block of the cell's first manifold; 32-bit division by a uniform value)
        run += v[k];
    }
    __syncthreads();
    if (blockIdx.x == 0) {   // publish the table and the blocks' ranges of perm
#pragma unroll
        for (uint32_t k = 0; k < per; ++k) {
            const uint32_t c = threadIdx.x * per + k;
            cellBlock[c] = blk[c];
            const uint32_t j1 = blk[c], j0 = c ? blk[c - 1u] + 1u : 0u;   // blocks (j0 .. j1] start at this cell (blk is non-decreasing)
            for (uint32_t j = j0; j <= j1; ++j) blockStart[j] = lower[c];
            if (c == kBlockCells - 1u) for (uint32_t j = j1 + 1u; j <= nbe; ++j) blockStart[j] = total;
        }
    }
    const uint32_t nm = min(n, sc->numManifolds);
#pragma unroll
    for (uint32_t i = 0; i < kKeyItems / 256; ++i) {
        const uint32_t m = blockIdx.x * kKeyItems + i * 256 + threadIdx.x;
        if (m < nm) {
            const uint32_t k = keys[m], ka = k & 0xFFFu;
            perm[lower[ka] + ranks[m]] = m;
            if (k >> 24) {
                const uint32_t ja = blk[ka], jb = blk[(k >> 12) & 0xFFFu];
                if (ja != jb) { const uint32_t at = atomicAdd(&extraCount[jb], 1u); if (at < extraCap) extra[(size_t)jb * extraCap + at] = m; }
            }
        }
    }
}

// One workgroup per block: its list (own manifolds + the boundary manifolds of its neighbours that touch its home bodies) sorted by
// (colour, boundary, contacts descending) into dense tiles; also everything k_schedule_finish did per manifold (colour history of the
// newly coloured ones, the (colour, contacts) counts the host mirrors) and the per-body boundary-colour masks the exports go by.
__global__ __launch_bounds__(256) void k_block_sched(uint32_t lastRound, const uint32_t* __restrict__ roundFlags, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ blockStart, const uint16_t* __restrict__ cellBlock, const uint32_t* __restrict__ extra, uint32_t extraCap,
                                                     const uint32_t* __restrict__ extraCount, uint32_t tilesPerBlock, const uint32_t* __restrict__ color, const uint2* __restrict__ manInfo,
                                                     const uint4* __restrict__ colWork, uint32_t* __restrict__ order, uint4* __restrict__ tileInfo, uint2* __restrict__ tileDesc,
                                                     unsigned long long* __restrict__ bndMask, StepScalars* sc, BlockState* __restrict__ bs,
                                                     uint32_t nc, const uint32_t* __restrict__ manPair, const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB,
                                                     HistSlot* __restrict__ tab, uint32_t tabMask, const uint8_t* __restrict__ manKept) {
    __shared__ uint32_t hist[kBlockSortBins];     // counts, then the first slot of every bin
    __shared__ uint32_t cls[kColorBins];          // (colour, contacts) counts of the manifolds this block owns (the host's bins)
    __shared__ uint32_t part[256];
    __shared__ uint16_t lutS[kBlockCells];   // cell -> block
    const uint32_t J = blockIdx.x, T = tilesPerBlock;
    for (uint32_t b = threadIdx.x; b < kBlockSortBins; b += 256) hist[b] = 0u;
    for (uint32_t b = threadIdx.x; b < kColorBins; b += 256) cls[b] = 0u;
    if (J == 0 && threadIdx.x == 0) {
        sc->colorPending = roundFlags[lastRound];
        sc->totalTiles = gridDim.x * T; sc->totalCt = gridDim.x * T * 4u;
        bs->stamp = (bs->stamp + 1u) & 0xFFFFu;   // (16 bits ride in the mailbox tags; the host clears the mailboxes before the stamp comes round again)
    }
    const uint32_t p0 = blockStart[J], nP = blockStart[J + 1u] - p0;
    const uint32_t nEall = extraCount[J], nE = min(nEall, extraCap);
    uint32_t n = nP + nE;
    bool ok = nEall <= extraCap && n <= T * 64u && n <= kBlockMaxPer * 256u;
    if (threadIdx.x == 0) { atomicMax(&bs->need, nP + nEall); atomicMax(&bs->needExtra, nEall); if (!ok) { bs->overflow = 1u; sc->specOverflow = 1u; } if (nE) atomicAdd(&bs->ghostLanes, 2u * nE); }
    if (!ok) n = 0u;
    __syncthreads();
    // (three passes over the thread's entries so that the loads of one pass are all in flight together: the kernel is a chain of dependent gathers otherwise)
    uint32_t eM[kBlockMaxPer], eBin[kBlockMaxPer], eRank[kBlockMaxPer], eC[kBlockMaxPer], eK[kBlockMaxPer];
#pragma unroll
    for (uint32_t i = 0; i < kBlockMaxPer; ++i) {
        const uint32_t idx = i * 256u + threadIdx.x;
        eM[i] = 0xFFFFFFFFu;
        if (idx < n) eM[i] = idx < nP ? perm[p0 + idx] : (extra[(size_t)J * extraCap + (idx - nP)] | 0x80000000u);
    }
#pragma unroll
    for (uint32_t i = 0; i < kBlockMaxPer; ++i) {
        eC[i] = 0u; eK[i] = 0u;
        if (eM[i] != 0xFFFFFFFFu) { const uint32_t m = eM[i] & kOrderMask; eC[i] = (color[m] & 0xFFFFu) | ((manInfo[m].x & 7u) << 16) | (manKept[m] ? 1u << 24 : 0u); eK[i] = keys[m]; }
    }
    for (uint32_t k = threadIdx.x; k < kBlockCells; k += 256) lutS[k] = cellBlock[k];
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kBlockMaxPer; ++i) {
        eBin[i] = 0xFFFFFFFFu;
        if (eM[i] != 0xFFFFFFFFu) {
            const bool own = (eM[i] >> 31) == 0u;
            const uint32_t m = eM[i] & kOrderMask;
            const uint32_t craw = eC[i] & 0xFFFFu, c = craw == 0xFFFFu ? kUncolored : craw, cnt = (eC[i] >> 16) & 7u;
            const uint32_t k = eK[i];
            const bool bnd = (k >> 24) != 0u && lutS[k & 0xFFFu] != lutS[(k >> 12) & 0xFFFu];
            eM[i] = m | (bnd ? 0x40000000u : 0u) | (own ? 0u : 0x80000000u);
            if (c <= kOverflowColor && cnt >= 1u && cnt <= 4u) {
                if (own) atomicAdd(&cls[binOf(c, cnt)], 1u);
                if (c < kOverflowColor) {   // (an overflow colour voids the step: the host sees it in the counts)
                    eBin[i] = c * 8u + (bnd ? 4u : 0u) + (4u - cnt);
                    eRank[i] = atomicAdd(&hist[eBin[i]], 1u);
                    if (own && bnd) {   // the per-body masks of boundary colours (once per manifold: by its owner)
                        const uint4 w = colWork[m];
                        atomicOr(&bndMask[w.x & 0x3FFFFFFFu], 1ull << c);
                        atomicOr(&bndMask[w.y & 0x7FFFFFFFu], 1ull << c);
                    }
                }
            }
            if (own && !((eC[i] >> 24) & 1u) && c <= kOverflowColor) {   // kept colours were entered by k_emit_manifolds
                const uint64_t pk = (sc->partitioned ? pairsB : pairsA)[manPair[m]];
                tableInsert(tab, tabMask, historyKey(nc, (uint32_t)((pk >> 29) & 0x1FFFFFFFull), (uint32_t)(pk & 0x1FFFFFFFull)), c);
            }
        }
    }
    __syncthreads();
    // exclusive scan of the sort bins (kBlockSortBins = 520 <= 3 per thread)
    {
        constexpr uint32_t per = (kBlockSortBins + 255u) / 256u;
        uint32_t v[per], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < per; ++k) { const uint32_t b = threadIdx.x * per + k; v[k] = b < kBlockSortBins ? hist[b] : 0u; sum += v[k]; }
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
        for (uint32_t k = 0; k < per; ++k) { const uint32_t b = threadIdx.x * per + k; if (b < kBlockSortBins) hist[b] = run; run += v[k]; }
    }
    const uint32_t scheduled = part[255];
    __syncthreads();
    const size_t slot0 = (size_t)J * T * 64u;
#pragma unroll
    for (uint32_t i = 0; i < kBlockMaxPer; ++i)
        if (eBin[i] != 0xFFFFFFFFu) order[slot0 + hist[eBin[i]] + eRank[i]] = eM[i];
    for (uint32_t t = threadIdx.x; t < T; t += 256) {
        const uint32_t tile = J * T + t;
        const uint32_t cnt = scheduled > t * 64u ? min(64u, scheduled - t * 64u) : 0u;
        tileInfo[tile] = make_uint4(tile, tile * 64u, cnt | (4u << 8), tile * 4u);
        tileDesc[tile] = make_uint2(tile * 4u, 4u);
    }
    for (uint32_t b = threadIdx.x; b < kColorBins; b += 256) { const uint32_t c = cls[b]; if (c) atomicAdd(&sc->binStart[b], c); }   // COUNTS here (the host turns them into the prefix it mirrors)
}

// ------------------------------------------------------------------------------------------------------------------------------
// The solver.  One workgroup of kBlockWaves waves per block, one wave per SIMD (512 registers: the rows of a pass in VGPRs, the next
// pass's rows in flight into fixed ACC registers a152 .. a255, which the compiler never allocates: tests/test_capi_symbols.py).
// meta (k_contact_init, block mode) = (bodyA, bodyB, packed versions, w); w: blockMetaW (kernels.hpp).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kBlockWaves = 4;       // one wave per SIMD
constexpr uint32_t kHashEmpty = 0xFFFFFFFFu;
constexpr uint32_t kBlockSpinLds = 1u << 22, kBlockSpinMem = 1u << 17;
#define MI_ACC_LOAD2(A0, A1, addr) asm volatile("global_load_dwordx2 a[" #A0 ":" #A1 "], %0, off" : : "v"(addr) : "memory", "a" #A0, "a" #A1)
#define MI_ACC_READ2(dst, A0, A1) do { float x_, y_; asm volatile("v_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 : "=v"(x_), "=v"(y_)); (dst) = make_float2(x_, y_); } while (0)

// A home body's record in LDS, two 16-byte granules (v, tag), (w, tag).  Single ds_read_b128 / ds_write_b128 instructions (a lane's 16 bytes move in one LDS cycle),
// issued by inline asm: the compiler must neither split them into 8-byte halves nor cache or reorder them — other waves of the workgroup poll these records.
__device__ __forceinline__ uint32_t ldsAddr(const void* p) { return (uint32_t)(uintptr_t)p; }   // (a flat address of LDS carries the LDS offset in its low half)
// The hand-over word of a record is a SEPARATE dword (recTag): the writer stores the two data granules, waits for them, then stores the tag; a reader loads the tag
// FIRST and the granules behind it (LDS serves a wave's requests in order) — a tag that reads as expected means the data behind it is complete, whatever the LDS does
// with the four dwords of a 16-byte access that runs into bank conflicts.
__device__ __forceinline__ void ldsLoadBody(uint32_t addr, uint32_t tagAddr, f32x4& g0, f32x4& g1, uint32_t& tag) {
    asm volatile("ds_read_b32 %2, %4\n\tds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(g0), "=&v"(g1), "=&v"(tag) : "v"(addr), "v"(tagAddr) : "memory");
}
__device__ __forceinline__ void ldsLoadBodies(uint32_t addrA, uint32_t tagAddrA, uint32_t addrB, uint32_t tagAddrB, f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1, uint32_t& tagA, uint32_t& tagB) {
    asm volatile("ds_read_b32 %4, %8\n\tds_read_b32 %5, %9\n\tds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(tagA), "=&v"(tagB) : "v"(addrA), "v"(addrB), "v"(tagAddrA), "v"(tagAddrB) : "memory");
}
__device__ __forceinline__ void ldsStoreBody(uint32_t addr, uint32_t tagAddr, f32x4 g0, f32x4 g1, uint32_t tag) {
    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b32 %3, %4" : : "v"(addr), "v"(g0), "v"(g1), "v"(tagAddr), "v"(tag) : "memory");
}
__device__ __forceinline__ void waitVmcntSmall(uint32_t n) {   // wait until at most n (even; more than 14: 14 — waiting for more than needed is always safe) of the newest requests are outstanding
    switch (n < 14u ? n : 14u) {
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
template <uint32_t WAVES>
__device__ __forceinline__ void blockSolver(
    uint32_t sweeps, uint32_t tilesPerBlock, uint32_t hashSize /* power of two */, uint32_t bodyCap, uint32_t maxSlots /* tiles per wave, <= 16 */, uint32_t maxPasses /* per wave */,
    uint32_t impCap /* accumulated impulses (contacts) per wave */,
    const uint4* __restrict__ tileInfo, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal, const float2* __restrict__ slotMass,
    const float4* __restrict__ rows, const float4* gVel /* initial velocities, tag 0 (k_integrate_forces) */, float4* gVelOut /* final velocities */,
    float4* mail /* [bodies + 1][kMailRanks][2 parities][2] */, StepScalars* sc, BlockState* bs, uint32_t faultInject,
    uint32_t dbg /* development knock-outs (timing only, results are garbage): 1 no dependency waits, 2 no ghost loads, 4 no row loads, 8 no arithmetic, 16 no exports */,
    unsigned long long* dbgTimes /* development: [block][wave][8] wall-clock stamps (100 MHz), or null */) {
    unsigned long long* stampAt = dbgTimes ? dbgTimes + ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * 8u : nullptr;
#define MI_BSTAMP(i) do { if (stampAt && (threadIdx.x & 63u) == 0u) stampAt[i] = wall_clock64(); } while (0)
    MI_BSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    __shared__ uint32_t sCount, sErr;
    const uint32_t J = blockIdx.x, T = tilesPerBlock;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // ---- LDS carve-up
    float4* rec = reinterpret_cast<float4*>(ldsRaw);                                   // [bodyCap][2]: (v, tag), (w, tag)
    uint32_t* recBody = reinterpret_cast<uint32_t*>(rec + 2u * (size_t)bodyCap);        // [bodyCap] body of the record
    uint32_t* recTag = recBody + bodyCap;                                               // [bodyCap] version of the record = updates its body has received (the hand-over word)
    // the hash (body -> record) is only needed while the lists are set up: it shares its bytes with the accumulated impulses of the four waves
    unsigned char* uni = ldsRaw + (((size_t)bodyCap * 40u + 15u) & ~(size_t)15u);
    uint32_t* hKey = reinterpret_cast<uint32_t*>(uni);                                  // [hashSize] body of the slot
    uint16_t* hVal = reinterpret_cast<uint16_t*>(hKey + hashSize);                      // [hashSize] record of the slot
    const size_t uniBytes = (((size_t)hashSize * 6u > (size_t)WAVES * impCap * 8u ? (size_t)hashSize * 6u : (size_t)WAVES * impCap * 8u) + 15u) & ~(size_t)15u;
    float2* lImp = reinterpret_cast<float2*>(uni) + (size_t)wave * impCap;              // [impCap] lane-contiguous: a lane's contacts at lOff .. lOff + cnt
    const size_t waveBytes = ((size_t)maxSlots * 64u * 16u + (size_t)maxPasses * 16u + (size_t)maxSlots * 64u * 2u + (size_t)maxSlots * 16u + 15u) & ~(size_t)15u;
    unsigned char* wb = uni + uniBytes + (size_t)wave * waveBytes;
    uint4* lMeta = reinterpret_cast<uint4*>(wb);                                        // [maxSlots][64]
    uint4* lPass = lMeta + (size_t)maxSlots * 64u;                                      // [maxPasses] (slot | lo << 8 | hi << 16 | most contacts of the group << 24, -, boundary | most contacts of the tile << 8, tile)
    uint16_t* lOff = reinterpret_cast<uint16_t*>(lPass + maxPasses);                    // [maxSlots][64] first impulse of the lane
    uint4* lSlot = reinterpret_cast<uint4*>(lOff + (size_t)maxSlots * 64u);             // [maxSlots] (tile, most contacts of a lane, first pass, passes)
    for (uint32_t k = threadIdx.x; k < hashSize; k += blockDim.x) hKey[k] = kHashEmpty;
    if (threadIdx.x == 0) { sCount = 0u; sErr = bs->overflow ? 7u : 0u; }   // (7: the schedule already voided the step — a body with more boundary colours than mailbox slots, a list beyond its capacity: nothing to wait for)
    __syncthreads();
    const uint32_t stamp = bs->stamp << 16;
    // ---- prologue 1: this wave's tiles (t = wave, wave + 4, ...), their lanes' home bodies into the hash
    uint32_t mySlots = 0;
    uint32_t hsA[16], hsB[16];   // hash slot of the lane's home bodies per tile slot (compile-time indexed)
    bool fail = T > 16u * WAVES || maxSlots > 16u;
#pragma unroll
    for (uint32_t s = 0; s < 16; ++s) {
        hsA[s] = kHashEmpty; hsB[s] = kHashEmpty;
        const uint32_t t = wave + s * WAVES;
        if (t < T && s < maxSlots && !fail) {
            const uint4 ti = tileInfo[(size_t)J * T + t];
            const uint32_t count = ti.z & 0xFFu;
            if (count) {   // (tiles are dense: the first empty one ends the block's list)
                mySlots = s + 1u;
                if (lane < count) {
                    const uint4 m = slotMeta[(size_t)ti.x * 64u + lane];
                    const bool bnd = (m.w >> 9) & 1u, homeB = (m.w >> 10) & 1u;
                    const bool ghostA = bnd && homeB, ghostB = bnd && !homeB;
                    const bool homeA = !ghostA, homeBd = !ghostB;   // (also the bodies nobody updates — the static dummy, a kinematic body: a read-only copy, so every lane reads its bodies the same way)
                    auto insert = [&](uint32_t body) -> uint32_t {
                        uint32_t h = (body * 0x9E3779B1u) >> 8;
#pragma unroll 1
                        for (uint32_t probe = 0; probe < 64u; ++probe, ++h) {
                            const uint32_t at = h & (hashSize - 1u);
                            const uint32_t old = atomicCAS(&hKey[at], kHashEmpty, body);
                            if (old == kHashEmpty || old == body) return at;
                        }
                        return kHashEmpty;
                    };
                    if (homeA) { hsA[s] = insert(m.x); if (hsA[s] == kHashEmpty) fail = true; }
                    if (homeBd) { hsB[s] = insert(m.y); if (hsB[s] == kHashEmpty) fail = true; }
                }
            }
        } else if (t < T && s >= maxSlots && !fail) {
            if ((tileInfo[(size_t)J * T + t].z & 0xFFu) != 0u) fail = true;   // more tiles than the host sized this wave's LDS for
        }
    }
    if (fail) sErr = 4u;
    MI_BSTAMP(1);
    __syncthreads();
    // ---- prologue 2: hash slots -> records, initial velocities
    for (uint32_t k = threadIdx.x; k < hashSize; k += blockDim.x) {
        const uint32_t body = hKey[k];
        if (body != kHashEmpty) {
            const uint32_t idx = atomicAdd(&sCount, 1u);
            hVal[k] = (uint16_t)idx;
            if (idx < bodyCap) {
                recBody[idx] = body; recTag[idx] = 0u;
                const float4 v = gVel[2 * (size_t)body], w = gVel[2 * (size_t)body + 1];
                rec[2u * idx] = make_float4(v.x, v.y, v.z, __uint_as_float(0u));
                rec[2u * idx + 1u] = make_float4(w.x, w.y, w.z, __uint_as_float(0u));
            }
        }
    }
    MI_BSTAMP(2);
    __syncthreads();
    if (threadIdx.x == 0) { atomicMax(&bs->needBodies, sCount); if (sCount > bodyCap) sErr = 5u; }
    // ---- prologue 3: per-slot constants into LDS (home bodies as LDS addresses of their records), impulse offsets, the passes of this wave
    uint32_t numPasses = 0, impUsed = 0;
    const uint32_t recBase = ldsAddr(rec), tagBase = ldsAddr(recTag);
#pragma unroll
    for (uint32_t s = 0; s < 16; ++s) {
        if (s < mySlots) {
            const uint32_t t = wave + s * WAVES;
            const uint4 ti = tileInfo[(size_t)J * T + t];
            const uint32_t count = ti.z & 0xFFu;
            uint4 m = make_uint4(0u, 0u, 0u, 0u);
            if (lane < count) m = slotMeta[(size_t)ti.x * 64u + lane];
            if (hsA[s] != kHashEmpty) m.x = min((uint32_t)hVal[hsA[s]], bodyCap - 1u);   // record index (an index beyond the capacity voids the step above)
            if (hsB[s] != kHashEmpty) m.y = min((uint32_t)hVal[hsB[s]], bodyCap - 1u);
            lMeta[s * 64u + lane] = m;
            // lane-contiguous impulse storage: exclusive prefix of the contact counts over the wave's lanes
            const uint32_t cnt = lane < count ? (m.w & 7u) : 0u;
            uint32_t incl = cnt;
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) { uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
            lOff[s * 64u + lane] = (uint16_t)min(impUsed + incl - cnt, 0xFFFFu);
            impUsed += (uint32_t)__shfl((int)incl, 63, 64);
            // groups: maximal lane ranges of one (colour, boundary) key; sorted, so a range is contiguous and its first lane has the most contacts
            const uint32_t gk = lane < count ? ((m.w >> 3) & 127u) : 0xFFFFu;
            const uint32_t prev = (uint32_t)__shfl_up((int)gk, 1, 64);
            const bool start = lane < count && (lane == 0u || prev != gk);
            unsigned long long starts = __ballot(start);
            const uint32_t tmc = __ballot(cnt >= 4u) ? 4u : __ballot(cnt >= 3u) ? 3u : __ballot(cnt >= 2u) ? 2u : 1u;
            const uint32_t passBegin = numPasses;
            while (starts) {
                const uint32_t lo = (uint32_t)__ffsll((long long)starts) - 1u;
                starts &= starts - 1ull;
                const uint32_t hi = starts ? (uint32_t)__ffsll((long long)starts) - 1u : count;
                const uint32_t mc = (uint32_t)__shfl((int)cnt, (int)lo, 64);
                const uint32_t bndG = (uint32_t)__shfl((int)((m.w >> 9) & 1u), (int)lo, 64);
                if (lane == 0 && numPasses < maxPasses) lPass[numPasses] = make_uint4(s | (lo << 8) | (hi << 16) | (mc << 24), ti.w, bndG | (tmc << 8), ti.x);
                ++numPasses;
            }
            if (lane == 0) lSlot[s] = make_uint4(ti.x, tmc, passBegin, numPasses - passBegin);
        }
    }
    if (lane == 0) { atomicMax(&bs->needPasses, numPasses); atomicMax(&bs->needImp, impUsed); if (numPasses > maxPasses || impUsed > impCap) sErr = 6u; }
    MI_BSTAMP(3);
    __syncthreads();   // (every wave is done with the hash: its bytes become the impulses)
    for (uint32_t k = lane; k < impCap; k += 64u) lImp[k] = make_float2(0.f, 0.f);
    __syncthreads();
    if (sErr || (faultInject && J == 1u)) { if (threadIdx.x == 0) { if (faultInject && J == 1u) sc->solveError = 1u; else if (sErr != 7u) sc->solveError = sErr; bs->overflow = 1u; } return; }   // (7: the host sees the schedule's own flag)   // nothing persistent has been written: the host re-runs the step on another path
    MI_BSTAMP(4);
    if (numPasses) {
    // ---- main loop: software pipeline over (sweep, pass); the next pass's rows are requested while this pass waits for its bodies.
    // Everything the loop reads from global memory goes through inline asm into fixed ACC registers, so the only vmcnt arithmetic is the one written here.
    uint32_t dbgLo = 0u, dbgHi = 64u;
    auto fetchRows = [&](uint32_t slot) -> uint32_t {
        const uint4 sd = lSlot[slot];                                   // (tile, most contacts of a lane, ..)
        const uint32_t tmc = sd.y;
        const uint32_t cnt = (lane >= dbgLo && lane < dbgHi) ? lMeta[slot * 64u + lane].w & 7u : 0u;           // (0: padding lane; some lane has tmc contacts: none of the predicated groups below is empty)
        const size_t at = (size_t)sd.x * 64u + lane;
        const float4* row = rows + (size_t)sd.x * 4u * (kRows * 64u) + lane;
        if (0u < cnt) { MI_ACC_LOAD(152, 153, 154, 155, slotNormal + at); MI_ACC_LOAD2(156, 157, slotMass + at); }
        if (0u < tmc) { if (0u < cnt) { MI_ACC_LOAD(160, 161, 162, 163, row + 0u * 64u); MI_ACC_LOAD(164, 165, 166, 167, row + 1u * 64u); MI_ACC_LOAD(168, 169, 170, 171, row + 2u * 64u);
                                        MI_ACC_LOAD(172, 173, 174, 175, row + 3u * 64u); MI_ACC_LOAD(176, 177, 178, 179, row + 4u * 64u); MI_ACC_LOAD(180, 181, 182, 183, row + 5u * 64u); } }
        if (1u < tmc) { if (1u < cnt) { MI_ACC_LOAD(184, 185, 186, 187, row + 6u * 64u); MI_ACC_LOAD(188, 189, 190, 191, row + 7u * 64u); MI_ACC_LOAD(192, 193, 194, 195, row + 8u * 64u);
                                        MI_ACC_LOAD(196, 197, 198, 199, row + 9u * 64u); MI_ACC_LOAD(200, 201, 202, 203, row + 10u * 64u); MI_ACC_LOAD(204, 205, 206, 207, row + 11u * 64u); } }
        if (2u < tmc) { if (2u < cnt) { MI_ACC_LOAD(208, 209, 210, 211, row + 12u * 64u); MI_ACC_LOAD(212, 213, 214, 215, row + 13u * 64u); MI_ACC_LOAD(216, 217, 218, 219, row + 14u * 64u);
                                        MI_ACC_LOAD(220, 221, 222, 223, row + 15u * 64u); MI_ACC_LOAD(224, 225, 226, 227, row + 16u * 64u); MI_ACC_LOAD(228, 229, 230, 231, row + 17u * 64u); } }
        if (3u < tmc) { if (3u < cnt) { MI_ACC_LOAD(232, 233, 234, 235, row + 18u * 64u); MI_ACC_LOAD(236, 237, 238, 239, row + 19u * 64u); MI_ACC_LOAD(240, 241, 242, 243, row + 20u * 64u);
                                        MI_ACC_LOAD(244, 245, 246, 247, row + 21u * 64u); MI_ACC_LOAD(248, 249, 250, 251, row + 22u * 64u); MI_ACC_LOAD(252, 253, 254, 255, row + 23u * 64u); } }
        return tmc * kRows + 2u;
    };
    auto readRows = [&](ContactRows* cur, uint32_t mc, float4& nf, float2& mass) {
        MI_ACC_READ(nf, 152, 153, 154, 155); MI_ACC_READ2(mass, 156, 157);
        if (0u < mc) { MI_ACC_READ(cur[0].r[0], 160, 161, 162, 163); MI_ACC_READ(cur[0].r[1], 164, 165, 166, 167); MI_ACC_READ(cur[0].r[2], 168, 169, 170, 171);
                       MI_ACC_READ(cur[0].r[3], 172, 173, 174, 175); MI_ACC_READ(cur[0].r[4], 176, 177, 178, 179); MI_ACC_READ(cur[0].r[5], 180, 181, 182, 183); }
        if (1u < mc) { MI_ACC_READ(cur[1].r[0], 184, 185, 186, 187); MI_ACC_READ(cur[1].r[1], 188, 189, 190, 191); MI_ACC_READ(cur[1].r[2], 192, 193, 194, 195);
                       MI_ACC_READ(cur[1].r[3], 196, 197, 198, 199); MI_ACC_READ(cur[1].r[4], 200, 201, 202, 203); MI_ACC_READ(cur[1].r[5], 204, 205, 206, 207); }
        if (2u < mc) { MI_ACC_READ(cur[2].r[0], 208, 209, 210, 211); MI_ACC_READ(cur[2].r[1], 212, 213, 214, 215); MI_ACC_READ(cur[2].r[2], 216, 217, 218, 219);
                       MI_ACC_READ(cur[2].r[3], 220, 221, 222, 223); MI_ACC_READ(cur[2].r[4], 224, 225, 226, 227); MI_ACC_READ(cur[2].r[5], 228, 229, 230, 231); }
        if (3u < mc) { MI_ACC_READ(cur[3].r[0], 232, 233, 234, 235); MI_ACC_READ(cur[3].r[1], 236, 237, 238, 239); MI_ACC_READ(cur[3].r[2], 240, 241, 242, 243);
                       MI_ACC_READ(cur[3].r[3], 244, 245, 246, 247); MI_ACC_READ(cur[3].r[4], 248, 249, 250, 251); MI_ACC_READ(cur[3].r[5], 252, 253, 254, 255); }
    };
    (void)fetchRows(lPass[0].x & 0xFFu);
    bool dead = false;
    ContactRows cur[4]; float4 nf = make_float4(0.f, 0.f, 0.f, 0.f); float2 mass = make_float2(0.f, 0.f);
    uint32_t curSlot = 0xFFFFFFFFu; bool firstOfTile = false;
    uint32_t pendingStores = 0;   // export store instructions the previous pass issued
    for (uint32_t it = 0; it < sweeps && !dead; ++it)
        for (uint32_t pass = 0; pass < numPasses && !dead; ++pass) {
            const uint4 pd = lPass[pass];
            const uint32_t s = pd.x & 0xFFu, lo = (pd.x >> 8) & 0xFFu, hi = (pd.x >> 16) & 0xFFu, mc = pd.x >> 24;
            const bool bndG = (pd.z & 1u) != 0u;
            const bool act = lane >= lo && lane < hi;
            const uint4 meta = lMeta[s * 64u + lane];
            const uint32_t cnt = act ? (meta.w & 7u) : 0u;
            const bool homeIsB = (meta.w >> 10) & 1u;
            const uint32_t pk = meta.z;
            const uint32_t degA = (pk >> 7) & 127u, degB = (pk >> 21) & 127u;
            const uint32_t expA = it * degA + (pk & 127u), expB = it * degB + ((pk >> 14) & 127u);
            const bool ghostA = act && bndG && homeIsB, ghostB = act && bndG && !homeIsB;
            const bool ldsA = act && !ghostA, ldsB = act && !ghostB;                 // the body's record is in LDS ...
            const bool updA = ldsA && degA != 0u, updB = ldsB && degB != 0u;         // ... and this manifold updates it (a dynamic body at home here)
            // the ghost of a boundary lane: its state before this manifold's turn — the initial state (tag 0) straight from gVel, anything later from the
            // mailbox slot its home block exports it into
            const uint32_t gBody = ghostA ? meta.x : meta.y, gExp = ghostA ? expA : expB, gRank = (meta.w >> 21) & 7u;
            const float4* gSrc = gExp == 0u ? gVel + 2 * (size_t)gBody : mail + ((((size_t)gBody * kMailRanks + gRank) * 2u + (it & 1u)) * 2u);
            const uint32_t gTag = gExp == 0u ? 0u : (stamp | gExp);
            f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
            // the rows of this pass are the OLDEST requests in flight; behind them the export stores of the previous pass (write-through: their
            // acknowledgements take a memory round trip and nothing here needs them) and, in a boundary pass, the two ghost loads of every lane
            if (bndG) { if (act) { issueGranuleSc1(gSrc, g0); issueGranuleSc1(gSrc + 1, g1); } }
            firstOfTile = s != curSlot || numPasses == 1u || lPass[pass == 0u ? numPasses - 1u : pass - 1u].x % 256u != s;
            if (firstOfTile) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); readRows(cur, pd.z >> 8, nf, mass); curSlot = s; }
            const bool more = pass + 1u < numPasses || it + 1u < sweeps;
            uint32_t inFlight = 0;
            if (more && firstOfTile) inFlight = fetchRows(s + 1u < mySlots ? s + 1u : 0u);
            // home bodies from LDS
            f32x4 a0, a1, b0, b1;
            const uint32_t adA = recBase + 32u * (ldsA ? meta.x : 0u), adB = recBase + 32u * (ldsB ? meta.y : 0u);
            uint32_t tA, tB;
            const uint32_t taA = tagBase + 4u * (ldsA ? meta.x : 0u), taB = tagBase + 4u * (ldsB ? meta.y : 0u);
            ldsLoadBodies(adA, taA, adB, taB, a0, a1, b0, b1, tA, tB);
            if (bndG) {
                // the ghost loads are older than the prefetch: they have landed once at most `inFlight` loads are outstanding
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                landed(g0); landed(g1);
            }
            bool okA = !updA || tA == expA;
            bool okB = !updB || tB == expB;
            bool okG = !(ghostA || ghostB) || (__float_as_uint(g0.w) == gTag && __float_as_uint(g1.w) == gTag);
            uint32_t budget = bndG ? kBlockSpinMem : kBlockSpinLds;
            while (__ballot(!(okA && okB && okG)) != 0ull) {
                if (!okA) { ldsLoadBody(adA, taA, a0, a1, tA); okA = tA == expA; }
                if (!okB) { ldsLoadBody(adB, taB, b0, b1, tB); okB = tB == expB; }
                if (bndG) {
                    if (!okG) { issueGranuleSc1(gSrc, g0); issueGranuleSc1(gSrc + 1, g1); }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    landed(g0); landed(g1);
                    if (!okG) okG = __float_as_uint(g0.w) == gTag && __float_as_uint(g1.w) == gTag;
                }
                if ((--budget & 255u) == 0u) {   // out of budget, or another wave / block gave up: leave at once (the step is void, the host re-runs it on another path)
                    if (budget == 0u) { sc->solveError = 1u; __hip_atomic_store(&sErr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                    if (budget == 0u || __hip_atomic_load(&sErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u || __hip_atomic_load(&sc->solveError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { dead = true; break; }
                }
            }
            if (dead) break;
            if (ghostA) { a0 = g0; a1 = g1; }
            if (ghostB) { b0 = g0; b1 = g1; }
            // solve (bodies A and B side by side in packed lanes), impulses from / to LDS
            P3 pv, pw;
            pv.x = pk2(a0.x, b0.x); pv.y = pk2(a0.y, b0.y); pv.z = pk2(a0.z, b0.z);
            pw.x = pk2(a1.x, b1.x); pw.y = pk2(a1.y, b1.y); pw.z = pk2(a1.z, b1.z);
            const f32x2 sMass = pk2(-mass.x, mass.y);
            float2* li = lImp + lOff[s * 64u + lane];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                if (k < mc) {
                    if (k < cnt) {
                        float2 im = li[k];
                        solveOnePk(cur[k], nf, im, sMass, pv, pw);
                        li[k] = im;
                    }
                }
            }
            // publish: home bodies into LDS; exports write-through into the mailbox of the boundary manifold that comes next on the body
            const uint32_t nA = expA + 1u, nB = expB + 1u;
            if (updA) { f32x4 h0 = {pv.x.x, pv.y.x, pv.z.x, __uint_as_float(nA)}, h1 = {pw.x.x, pw.y.x, pw.z.x, __uint_as_float(nA)}; ldsStoreBody(adA, taA, h0, h1, nA); }
            if (updB) { f32x4 h0 = {pv.x.y, pv.y.y, pv.z.y, __uint_as_float(nB)}, h1 = {pw.x.y, pw.y.y, pw.z.y, __uint_as_float(nB)}; ldsStoreBody(adB, taB, h0, h1, nB); }
            const uint32_t eA = (meta.w >> 11) & 31u, eB = (meta.w >> 16) & 31u;
            pendingStores = (__ballot(updA && (eA & 1u)) != 0ull ? 2u : 0u) + (__ballot(updB && (eB & 1u)) != 0ull ? 2u : 0u);   // (a branch no lane takes issues nothing)
            if (updA && (eA & 1u)) {
                const uint32_t par = (eA & 2u) ? (it & 1u) : ((it + 1u) & 1u);
                float4* dst = mail + ((((size_t)recBody[meta.x] * kMailRanks + (eA >> 2)) * 2u + par) * 2u);
                const float t = __uint_as_float(stamp | nA);
                f32x4 h0 = {pv.x.x, pv.y.x, pv.z.x, t}, h1 = {pw.x.x, pw.y.x, pw.z.x, t};
                storeGranuleSc1(dst, h0); storeGranuleSc1(dst + 1, h1);
            }
            if (updB && (eB & 1u)) {
                const uint32_t par = (eB & 2u) ? (it & 1u) : ((it + 1u) & 1u);
                float4* dst = mail + ((((size_t)recBody[meta.y] * kMailRanks + (eB >> 2)) * 2u + par) * 2u);
                const float t = __uint_as_float(stamp | nB);
                f32x4 h0 = {pv.x.y, pv.y.y, pv.z.y, t}, h1 = {pw.x.y, pw.y.y, pw.z.y, t};
                storeGranuleSc1(dst, h0); storeGranuleSc1(dst + 1, h1);
            }
        }
    }
    MI_BSTAMP(5);
    __syncthreads();
    MI_BSTAMP(6);
    // ---- epilogue: the home bodies' final velocities
    const uint32_t nRec = min(sCount, bodyCap);
    for (uint32_t k = threadIdx.x; k < nRec; k += blockDim.x) {
        const uint32_t body = recBody[k];
        const float4 v = rec[2u * k], w = rec[2u * k + 1u];
        gVelOut[2 * (size_t)body] = make_float4(v.x, v.y, v.z, 0.f);
        gVelOut[2 * (size_t)body + 1] = make_float4(w.x, w.y, w.z, 0.f);
    }
    MI_BSTAMP(7);
#undef MI_BSTAMP
}
#define MI_BLOCK_PARAMS uint32_t sweeps, uint32_t tilesPerBlock, uint32_t hashSize, uint32_t bodyCap, uint32_t maxSlots, uint32_t maxPasses, uint32_t impCap, const uint4* __restrict__ tileInfo, \
    const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal, const float2* __restrict__ slotMass, const float4* __restrict__ rows, const float4* gVel, float4* gVelOut, float4* mail, \
    StepScalars* sc, BlockState* bs, uint32_t faultInject, uint32_t dbg, unsigned long long* dbgTimes
#define MI_BLOCK_FORWARD sweeps, tilesPerBlock, hashSize, bodyCap, maxSlots, maxPasses, impCap, tileInfo, slotMeta, slotNormal, slotMass, rows, gVel, gVelOut, mail, sc, bs, faultInject, dbg, dbgTimes
// one wave per SIMD, the rows a tile ahead in a152 .. a255
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_contact_solve_blocks(MI_BLOCK_PARAMS) { blockSolver<4u>(MI_BLOCK_FORWARD); }
#undef MI_BLOCK_PARAMS
#undef MI_BLOCK_FORWARD

}  // namespace mi
