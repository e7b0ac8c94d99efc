// kernels_common.hpp — constants, StepScalars, counter shards, block-wise activity of a sharded world.
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

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

// A speculative step that is already known to be void by the time its solver starts — a bound was exceeded, the colouring did not finish, or an overflow colour appeared (only the
// per-colour path solves that one) — is run again synchronously by the host, whatever the solver does.  The dataflow kernels look here first and leave: they would otherwise wait,
// tile after tile, for updates of bodies that the missing manifolds never make, until the spin budget ends the launch (0.4-0.7 s; found by tools/gpu_fuzz.py on a heap of
// bodies spawned into one another).
__device__ __forceinline__ bool stepIsVoid(const StepScalars* sc) {
    return (sc->specOverflow | sc->colorPending | (sc->binStart[kColorBins] - sc->binStart[kOverflowColor * 4u])) != 0u;
}

#ifdef MI_DBG_KNOCKOUT
// development (knock-out harness, tools/gpu_knockout.sh).  Bits 0-2: k_contact_solve_persist (see there).  Bits 8-12: k_emit_manifolds launched a first time with its
// read-modify-write targets redirected to scratch and parts removed: 8 no bodyUsed atomics, 9 no history insert, 10 no history probe, 11 no round-0 proposals, 12 no material gathers.
// Bits 16-18: the grid part of k_bp_pairs a first time on a scratch pair list / scratch counters: 16 no candidate loop, 17 candidates tested but hits neither keyed nor
// staged, 18 no block flush (reservation + copy-out).
__device__ uint32_t g_dbgKnock = 0u;
#define MI_EMIT_KNOCK(bit) ((g_dbgKnock >> (bit)) & 1u)
#define MI_BP_KNOCK(bit) ((g_dbgKnock >> (bit)) & 1u)
#else
#define MI_EMIT_KNOCK(bit) 0u
#define MI_BP_KNOCK(bit) 0u
#endif
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
__global__ __launch_bounds__(256) void k_fill_u32_from(uint32_t* __restrict__ p, const uint32_t* __restrict__ value, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = *value; }
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

}  // namespace mi
