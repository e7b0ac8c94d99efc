// kernels_scan.hpp — device-wide exclusive prefix sum (chained scan with decoupled look-back) and the pose rows for the caller.
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

namespace mi {

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
