// kernels_shard.hpp — sharded world (multi-GPU): classification, packing, exact seam, load-balance histogram.
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

namespace mi {

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
                                                        const uint8_t* __restrict__ bodyActivePrev, uint32_t* __restrict__ blockStamp, uint8_t* __restrict__ blockLive,
                                                        const uint32_t* __restrict__ stepPtr /* the step's number, through memory: a launch argument would change a sharded step's graph signature every step */) {
    __shared__ uint32_t cnt;
    const uint32_t step = *stepPtr;
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

}  // namespace mi
