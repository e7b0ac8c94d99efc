// world.hip — host side of the MI355X rigid-body stepper: scene storage, per-step orchestration of
// the HIP kernels (kernels.hpp) on a world-owned stream, and the C ABI of include/mi_physics.h.
//
// Replaces physicsStep / physicsStepInternal (src/physics/physics.cpp:1180-1413) and the scene
// hooks that feed it (src/scene/scene.h:35-112).  There is NO CPU fallback: without a HIP device
// mi_world_create fails with MI_ERR_NO_DEVICE.
#include <cstring>
#include <limits>
#include <cstdio>
#include <cmath>
#include <string.h>
#include <vector>
#include <unordered_map>
#include <chrono>
#include <atomic>
#include <thread>
#include <string>
#include <algorithm>
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "../../include/mi_physics.h"
#include "../../include/mi_shard.h"
#include <dlfcn.h>
#include "../../include/mi_constraints.h"
#include "kernels.hpp"
#include "blocks.hpp"
#include "gjk.hpp"
#include "launcher.hpp"
#include "joints.hpp"
#include "heightmap.hpp"
#include "cloth.hpp"

using namespace mi;

static thread_local std::string g_lastError;
static int fail(int code, const std::string& msg) { g_lastError = msg; return code; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(MI_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Device buffer (grow-only)
// ------------------------------------------------------------------------------------------------
template <typename T>
struct DBuf {
    T* p = nullptr;
    size_t cap = 0;
    unsigned flags = 0;   // hipExtMallocWithFlags flags (0 = plain hipMalloc); development experiments only
    ~DBuf() { if (p) (void)hipFree(p); }
    hipError_t ensure(size_t n, bool keep = false, hipStream_t st = nullptr) {
        if (n <= cap) return hipSuccess;
        size_t ncap = std::max(n, cap + cap / 2);
        T* np = nullptr;
        hipError_t e = flags ? hipExtMallocWithFlags((void**)&np, ncap * sizeof(T), flags) : hipMalloc((void**)&np, ncap * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p && cap) { e = hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, st); if (e != hipSuccess) return e; (void)hipStreamSynchronize(st); }
        if (p) (void)hipFree(p);
        p = np; cap = ncap;
        return hipSuccess;
    }
};

// One scan site: the record arrays, ticket counter and generation of k_exclusive_scan (kernels.hpp).  Nothing is reset between
// launches; the records are zeroed when (re)allocated and when the 32-bit generation wraps.
template <typename T>
struct DeviceScan {
    DBuf<unsigned long long> records; DBuf<uint32_t> ticket;   // ticket.p[0] = ticket counter, [1] = tickets handed out before the next launch, [2] = generation
    uint32_t launches = 0;   // upper bound of the launches since the last clear (dry signature passes count too)
    bool clearPending = false;
    hipError_t run(Launcher& L, T* in, T* out, uint32_t n, hipStream_t st, bool zeroInput = false) {
        const uint32_t tiles = (n + ScanItems<T>::Tile - 1) / ScanItems<T>::Tile;
        if (!tiles) return hipSuccess;
        hipError_t e;
        if (!ticket.p) { if ((e = ticket.ensure(4)) != hipSuccess) return e; clearPending = true; }
        if (++launches >= (1u << 29)) clearPending = true;       // the tag holds 30 generation bits
        const size_t words = (size_t)tiles * ScanWords<T>::W;
        if (records.cap < words) { if ((e = records.ensure(words)) != hipSuccess) return e; clearPending = true; }
        if (clearPending) {
            if ((e = L.memsetAsync(records.p, 0, records.cap * sizeof(unsigned long long), st)) != hipSuccess) return e;
            if ((e = L.memsetAsync(ticket.p, 0, 4 * sizeof(uint32_t), st)) != hipSuccess) return e;
            if (!L.dry) { clearPending = false; launches = 1; }   // (a signature pass enqueues nothing: the clear is still owed)
        }
        L.launch(k_exclusive_scan<T>, dim3(tiles), dim3(kScanThreads), 0, st, in, out, n, records.p, ticket.p, ticket.p + 1, zeroInput ? 1u : 0u);
        return L.firstError;
    }
};

// ------------------------------------------------------------------------------------------------
// Host scene storage
// ------------------------------------------------------------------------------------------------
struct HEntity { V3 pos; Q4 rot; uint32_t kind; int rb = -1; std::vector<uint32_t> colliders; /* newest first */ V3 force; uint32_t kindIndex = 0; /* force fields, triggers */ };
struct HBody {
    uint32_t entity;
    V3 localCOG; float invMass; M3 invInertia;
    float gravityFactor, linDamp, angDamp;
    V3 linVel, angVel, force, torque;
    V3 p0, p1; Q4 r0, r1;
    uint8_t shardKnown = 1;   // sharded world: this rank's copy of the body is current (mirror of ShardState::known across re-uploads; follows the body through deletions)
};
struct HCollider { uint32_t entity; mi_collider_desc desc; };
struct HHull { std::vector<V3> verts; std::vector<uint32_t> tris; V3 mn, mx; };
struct MassProps { M3 inertia; V3 cog; float mass; };

static uint32_t divUp(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static float elapsedMs(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    hipError_t e = hipEventElapsedTime(&ms, a, b);
    if (e == hipErrorNotReady) { (void)hipGetLastError(); (void)hipEventSynchronize(a); (void)hipEventSynchronize(b); e = hipEventElapsedTime(&ms, a, b); }
    if (e != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
    return ms;
}
// Host loops over every body / entity (read-back conversions, interpolation) in slices on a few threads once they are long: at 57 k bodies
// the serial loop of download() — a quaternion normalisation per body — was 1.4 ms of a 4.5 ms batched learning step.  Independent iterations only.
template <class F>
static void hostParallelFor(uint32_t n, F&& fn) {
    const uint32_t kMinPerThread = 8192;
    uint32_t threads = std::min<uint32_t>({8u, std::max(1u, std::thread::hardware_concurrency()), n / kMinPerThread});
    if (threads <= 1) { for (uint32_t i = 0; i < n; ++i) fn(i); return; }
    std::vector<std::thread> pool; pool.reserve(threads - 1);
    const uint32_t per = (n + threads - 1) / threads;
    uint32_t started = 1;
    for (uint32_t t = 1; t < threads; ++t) {
        try { pool.emplace_back([&fn, t, per, n]() { const uint32_t lo = t * per, hi = std::min(n, lo + per); for (uint32_t i = lo; i < hi; ++i) fn(i); }); ++started; }
        catch (...) { break; }   // no more threads to be had: this thread takes the rest
    }
    for (uint32_t i = 0; i < std::min(n, per); ++i) fn(i);
    for (uint32_t i = std::min(n, started * per); i < n; ++i) fn(i);
    for (std::thread& th : pool) th.join();
}

// kernel arguments with padding bytes enter a step's signature field by field (launcher.hpp)
namespace mi {
template <> inline void sigMix<InterSink>(Launcher& L, const InterSink& v) { L.pod(v.keys); L.pod(v.cap); L.pod(v.count); }
template <> inline void sigMix<HmOut>(Launcher& L, const HmOut& v) { L.pod(v.sc); L.pod(v.pairCap); L.pod(v.pairsA); L.pod(v.pairsB); L.pod(v.npPacked); L.pod(v.npNormal); L.pod(v.npPoints); }
template <> inline void sigMix<HeightmapParams>(Launcher& L, const HeightmapParams& v) {
    L.pod(v.heights); L.pod(v.mips); L.pod(v.chunkSlot); L.pod(v.chunksPerDim); L.pod(v.chunkSize); L.pod(v.invChunkSize); L.pod(v.chunkScale); L.pod(v.heightScale);
    L.pod(v.invAmplitudeScale); L.pod(v.minX); L.pod(v.minY); L.pod(v.minZ); L.pod(v.restitution); L.pod(v.friction);
}
}

struct mi_world {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<HEntity> entities;
    std::vector<HBody> bodies;
    std::vector<HCollider> colliders;   // creation order
    std::vector<HHull> hulls;
    JointSet joints;
    bool topologyDirty = true;   // entities/colliders changed -> re-upload everything
    bool hostStale = false;      // device holds newer body state than host
    // sharded world (include/mi_shard.h): the rank's tile, per-body activity (1 owned, 2 ghost, 0 elsewhere), message buffers, transport
    struct ShardState {
        bool enabled = false, rccl = false;
        mi_shard_desc desc{}; ShardParams sp{};
        std::vector<float> bordersX, bordersZ;   // interior tile borders (tiles - 1 per axis): uniform at enable, moved by mi_world_shard_set_borders
        std::vector<float> nextX, nextZ; ShardParams spNext{}; bool bordersPending = false;   // ... in force after the next step's exchange
        DBuf<uint8_t> known;                     // per body: this rank's copy is current (owned in the last step, or a record arrived)
        DBuf<uint32_t> hist; DBuf<uint64_t> reduceBuf;
        DBuf<uint32_t> axisDev; DBuf<unsigned long long> axisGlobal;   // the sweep axis of the next step lives on the device (k_shard_axis: from the centre statistics summed over all ranks, include/mi_shard.h "Global sweep axis")
        bool axisHostCurrent = true;                 // sapAxis (host) equals *axisDev (not so after a library-transport exchange, until somebody asks)
        uint32_t capacity = 0; std::vector<uint32_t> peerRanks;
        DBuf<uint8_t> active, activePrev; bool prevValid = false, flagsSwapPending = false, stepOpen = false, flagsOfAStep = false;   // flagsOfAStep: `active` holds what the last step classified (not so right after enable / a re-upload)   // activePrev: the previous valid step's flags (k_integrate_velocities skips bodies idle in both)
        DBuf<float> sendBuf[8], recvBuf[8], importBuf;   // importBuf: staging of mi_world_shard_import (a tile without neighbours has no recvBuf)
        DBuf<uint32_t> root; size_t rootJoints = ~size_t(0), rootBodies = 0;   // island root of every body (union-find over the joints), rebuilt when the scene changes
        uint32_t* sentHost = nullptr;            // pinned: the records packed per slot in the previous exchange (overflow check)
        bool sentPending = false;
        hipEvent_t exEv[2] = {nullptr, nullptr}; bool exchangeTimed = false; double exchangeMsSum = 0.0; uint64_t exchangesTimed = 0;   // device time of the exchanges (pack -> send / receive -> unpack -> axis)
        uint32_t sentLast[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t sentSum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // library transport: a neighbour message travels as long as the previous exchange made it in EITHER direction (x 1.5 + 512 records) — both ends know both numbers, so
        // they agree on the size without talking; full size for the exchanges after anything that moves many bodies at once (enable, attach, new borders, a restore)
        uint32_t* recvHost = nullptr; uint32_t recvLast[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sizedLast[8] = {0, 0, 0, 0, 0, 0, 0, 0}; bool recvValid = false, adaptive = true; uint32_t fullExchanges = 2; uint64_t bytesSentSum = 0;
        // ... the sweep messages of the exact seam likewise, from the previous STEP's list lengths in both directions (x 1.5 + 64)
        uint32_t sweepPrevOwn[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sweepPeerHdr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sweepSized[8] = {0, 0, 0, 0, 0, 0, 0, 0}; bool sweepRecvValid = false, sweepCut = false; uint32_t sweepFullSteps = 2;
        uint32_t owned[3] = {0, 0, 0};
        void* comm = nullptr;                    // ncclComm_t
        size_t messageFloats() const { return (size_t)(capacity + 1u) * kShardRecordFloats; }
        // exact seam (include/mi_shard.h): per-sweep hand-over of the owners' velocities of the shared bodies
        bool exact = false; mi_shard_sweep_fn sweepFn = nullptr; void* sweepUser = nullptr;
        DBuf<float> sweepSend[8], sweepRecv[8], sweepImport; DBuf<uint32_t> sweepList[8], sweepCount;
        uint64_t sweepExchanges = 0;
        uint32_t sweepsDone = 0; uint32_t sweepCounts[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // records per neighbour message of this step (the lists are built once per step)
        size_t sweepFloats() const { return (size_t)(capacity + 1u) * kSweepRecordFloats; }
    } shard;
    // exact seam: a single world that was told a tiling orders its colours the same way (mi_world_set_seam_tiling)
    struct SeamTiling { bool on = false; std::vector<float> bx, bz; float margin = 0.f; DBuf<float> dBx, dBz; } seamTiling;
    DBuf<uint32_t> seamId; uint32_t seamLast[2] = {0, 0}; uint64_t seamViolations = 0;
    bool seamMode() const { return seamTiling.on || (shard.enabled && shard.exact); }
    int shardSweepExchange(uint32_t sweep);
    int shardExchange();
    int shardCheckOverflow(bool sync);
    int shardSyncAxis();
    void shardFillBorders(ShardParams& sp, const std::vector<float>& bx, const std::vector<float>& bz) const;
    int shardBuildRoots();
    void shardReleaseComm();
    bool transformsFollowPhysics = false;   // last stepped through mi_world_step_fixed: entity transforms = physics_transform1 at the next download
    // mi_world_step (physicsStep): physics_transform0 is kept ON THE DEVICE (bPos0 / bRot0, copied from transform1 before the sub-steps) and the
    // interpolated entity transforms are produced at the next download — the call itself moves nothing to the host (it used to download the
    // whole body state twice per call: at 57 k bodies that was most of a batched learning step)
    bool lerpPending = false, p0OnDevice = false; float lerpT = 0.f;
    float4* downloadStage = nullptr; size_t downloadStageCap = 0;   // pinned staging of download()
    // Poses for a caller that reads them after every step (a renderer): k_entity_poses writes the entity transforms in the caller's layout, ONE copy
    // per step brings them into pinned host memory in kPoseChunks pieces on a second stream, and once a caller has asked after a step the next
    // step enqueues both by itself, before it returns — the copy is then under way while control goes back to the caller.
    static constexpr uint32_t kPoseChunks = 4;
    struct PoseStream {
        bool enabled = true;                     // MI_POSE_STREAM=0: the per-array copies + host pass
        DBuf<int> entBody; DBuf<float> out;      // entity -> rigid body (or -1); [n][4] rotations followed by [n][3] positions
        float* host[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; size_t hostCap = 0;   // pinned; per flavour (entity transforms / physics transforms) two sets alternate: what mi_world_view_* hands out stays untouched by the next step
        hipStream_t copyStream = nullptr; hipEvent_t produced = nullptr; hipEvent_t chunkEv[kPoseChunks] = {nullptr, nullptr, nullptr, nullptr};
        size_t chunkOff[kPoseChunks + 1] = {0, 0, 0, 0, 0};
        std::vector<uint32_t> noBody;            // entities without a rigid body: their transform is the host's
        uint32_t tableCount = 0; bool tablesValid = false;
        bool valid = false, copyInFlight = false; uint64_t steps = 0; float t = 0.f; uint32_t n = 0; int cur[2] = {0, 0}, flavour = 0;   // the last production
        bool wanted = false, consumed = false, askedPhysics = false, retrySameStep = false;
        uint32_t produced_ahead = 0, produced_on_demand = 0;
    } pose;
    struct PoseArm { bool armed = false, done = false; float t = 0.f; } poseArm;   // the stepping call wants the LAST of its internal steps to enqueue the rows itself, behind its kernels and ahead of the host's wait
    bool posesPossible(bool physics, float* t) const;
    bool posesWantedAhead();
    void posesArm(bool lerpAfterwards, float lerpTAfterwards);
    int posesProduce(float t, bool fromNextState);
    int posesFetch(float* p, float* r, const float** viewP, const float** viewR);
    int posesAfterStep();
    float timer = 0.f;

    // device: bodies
    DBuf<float4> bPos, bRot, bLinVel, bAngVel, bForce, bTorque, bCogInvMass, bInvI, bParams;
    DBuf<float4> bPos0, bRot0;   // physics_transform0 (see lerpPending)
    DBuf<float4> bPosN, bRotN, bLinVelN, bAngVelN, bForceN, bTorqueN;   // second body-state set: written by k_integrate_velocities, swapped in when a step is valid
    DBuf<float4> gPos, gInvI, gVel;
    // XCD-partitioned persistent solver: cached velocity copy for XCD-local bodies, per-body XCD set, spatial sort of the manifolds, per-XCD tile lists
    DBuf<float4> gVelL; DBuf<unsigned long long> bodyOwner; DBuf<uint32_t> sortKeys[2], sortVals[2], xcdBase, xcdTiles, keyCount;
    bool privateIslandsEnabled = true;
    // spatial blocks in LDS (blocks.hpp): one workgroup per block of the scene, home bodies in LDS, boundary manifolds solved by both neighbours
    DBuf<uint32_t> blkKeys, blkRanks, blkPerm, blkStart, blkExtra, blkExtraCount; DBuf<uint16_t> blkCell; DBuf<unsigned long long> bndMask; DBuf<float4> mail;
    bool blockSolver = true, usedBlocks = false, blockFaultTest = false, blockFaultFired = false;
    struct BlockCaps { uint32_t nbe = 0, tiles = 0, extraCap = 0, bodyCap = 0, hashSize = 0, maxPasses = 0, impCap = 0; size_t lds = 0; } blkCaps;   // sticky: the same launches step after step (step graphs)
    BlockState lastBlk{}; bool haveBlkEstimate = false, blkLastFailed = false; uint32_t blkWaves = 4 /* waves per block: one per SIMD */, blkFailHistory = 0, blkFailures = 0, blkDisabledSteps = 0, blkLaunches = 0, blkMaxBlocks = 256, blkSteps = 0;
    bool planBlocks(uint32_t nmLast, uint32_t nbBodies);
    bool persistXcd = true, persistXcdSingle = true, usedXcd = false, usedXcdSingle = false, lastXcdSingle = false, haveXcdEstimate = false, xcdFaultFired = false, xcdFaultTest = false; uint32_t lastXcdMax = 0, xcdMinManifolds = 16384;
    // device: colliders
    DBuf<uint32_t> cTypeBody; DBuf<float4> cShape, cStaticPos, cStaticRot, cEmit;
    DBuf<float4> wShape, aabbMin, aabbMax, hullAabb, hullVerts; DBuf<uint32_t> hullRanges;
    // broad phase
    DBuf<unsigned long long> axisPartials;
    DBuf<uint32_t> largeList, isLarge, cellKeys, cellRanks, cellKeysS, cellValsS, cellCount, cellLower;
    DBuf<int> blockBounds;
    DBuf<float4> sMin, sMax;
    DBuf<GridParams> grid;   // [2]: the grid a step uses and the one its k_pair_finish prepares for the next step
    uint32_t gridCur = 0; bool gridValid = false; uint32_t gridNextCells = 0; DBuf<char> scalarsRaw; DBuf<Shards> shards;   // scalarsRaw = [StepScalars][colouring round flags]: one read-back
    StepScalars* scalarsPtr() { return reinterpret_cast<StepScalars*>(scalarsRaw.p); }
    uint32_t* roundFlagsPtr() { return reinterpret_cast<uint32_t*>(scalarsRaw.p + sizeof(StepScalars)); }
    DBuf<uint64_t> pairKeys, pairKeysS; DBuf<uint8_t> manKept; DBuf<uint32_t> epaQueue; DBuf<float4> epaSimplex;   // epaQueue / epaSimplex: k_narrow_gjk -> k_narrow_epa   // manKept: manifold kept its colour (already in the next step's history)
    DeviceScan<uint32_t> scanCells, scanBins; DeviceScan<unsigned long long> scanPairs, scanTerrain;   // one per scan site (own tickets / generations)
    // narrow phase
    DBuf<uint64_t> npPacked, npScan; DBuf<float4> npNormal, npPoints; DBuf<BoxHit> boxQueue;
    DBuf<uint32_t> manPair; DBuf<uint2> manBodies, manInfo; DBuf<uint4> colWork;
    // schedule + solver
    DBuf<uint32_t> color, order, orderTmp, blockHist, blockScan; DBuf<uint4> tileInfo, xcdInfo; DBuf<unsigned long long> bodyTop, bodyUsed;
    DBuf<BinInfo> binInfo;
    // colour history (pair -> colour of the previous step): two tables, the one written by a step becomes current only if the step is valid
    DBuf<HistSlot> tab[2]; uint32_t tabMask[2] = {0, 0}; int tabCur = 0; bool tabValid = false;
    // collision events (mi_world_enable_events / mi_world_poll_events)
    // triggers / force fields: entity lists by dense index, per-collider object tags, rotated forces, the pair pass's
    // rigid-body x (trigger | force field) AABB overlaps, the interactions that passed the boolean test, the per-step force accumulators
    std::vector<uint32_t> ffEntities, triggerEntities;
    bool usesInteractions = false; V3 globalForce;
    DBuf<uint32_t> cEntity, hullTris, hullTriRanges;   // ray tests (testPhysicsInteraction): entity of every collider, hull faces
    DBuf<uint32_t> cObject; DBuf<float4> localForce, bForceStep; DBuf<uint64_t> interKeys; DBuf<DeviceInteraction> interList; DBuf<uint2> fieldList;
    std::vector<uint64_t> prevTriggerOverlaps, nextTriggerOverlaps;
    int interactions(std::vector<mi_event>& triggerEvents);
    int interactionsDevice(uint32_t interPairBound);                                  // the speculative form: everything stays on the device
    int triggerEventsFrom(const std::vector<DeviceInteraction>& sortedList, std::vector<mi_event>& out);   // host half: trigger overlaps of a step, diffed against the previous step's
    DBuf<DeviceInteraction> interSorted;
    // cloth (cloth_component): host description + device state per cloth, one descriptor array for the single launch
    struct HCloth {
        mi_cloth_desc desc; float oldTotalMass, oldStiffness;
        std::vector<float> invMasses; std::vector<uint2> pairs; std::vector<float2> restInvMass; std::vector<uint32_t> order; uint32_t colourOffsets[13];
        DBuf<float4> pos, prev, vel, force, temp; DBuf<uint2> dPairs; DBuf<float2> dRestInvMass; DBuf<uint32_t> dOrder;
        bool constraintsDirty = true;
    };
    std::vector<HCloth*> cloths;
    DBuf<ClothDev> clothDescs; bool clothsDirty = true;
    uint32_t clothIterations[3] = {0, 1, 0};   // numClothVelocity / Position / DriftIterations (physics.h:390-392)
    int stepCloths(float dt);
    // heightmap terrain: host copy of the chunks (heights + min/max mips), device pool, per-collider contact counts
    struct HHeightmap {
        uint32_t chunksPerDim; float chunkSize, restitution, friction; V3 minCorner; float amplitudeScale = 1.f;
        std::vector<std::vector<uint16_t>> heights;    // per chunk: empty or 129 * 129
        bool dirty = true;
    };
    HHeightmap* heightmap = nullptr;
    HeightmapParams hmParams{};
    std::vector<uint16_t> hmHostHeights; std::vector<uint32_t> hmHostSlots;   // host mirror of the device pool (mi_heightmap_get_height)
    DBuf<uint16_t> hmHeights; DBuf<uint32_t> hmMips, hmChunkSlot;
    DBuf<uint32_t> hmStash;   // [colliders][kHmStash]: which triangles the counting pass of k_hm_contacts hit (the WRITE pass recomputes just those)
    DBuf<unsigned long long> hmPacked, hmScan; DBuf<uint8_t> hmSlow;   // per collider: contacts | touching << 32, its exclusive scan, sequential-walk flag
    uint32_t manifoldsLast = 0;   // device manifolds of the last step (heightmap contacts are one-contact manifolds; counts.num_collisions is per collider)
    int uploadHeightmap();
    bool eventsEnabled = false; DBuf<uint8_t> manIsNew; DBuf<DeviceEvent> devEvents; std::vector<mi_event> pendingEvents;
    DBuf<float4> rows, slotNormal; DBuf<float4> imp; DBuf<float2> slotMass; DBuf<uint4> slotMeta; DBuf<uint2> tileDesc;
    bool usedFlow = false, skippedPartition = false, havePartitionFlag = false, lastPartitioned = false;
    bool usedFused = false;
    bool persistSolver = true, persistMetaLds = true, persistImpLds = true, usedPersist = false; uint32_t xcdOnly = 0; uint32_t persistWaves = 1024;   // one resident workgroup per SIMD owns its tiles through all sweeps (k_contact_solve_persist)
    uint32_t flowLds = 0;                  // dynamic LDS bytes per 64-lane workgroup: caps resident waves per CU (160 KiB / flowLds)
    uint32_t flowFallbacks = 0;
    uint32_t launchFallbackSteps = 0;     // > 0: the dispatch-ordered dataflow kernel ran out of spin budget (shared device?) -> per-colour launches for this many steps
    bool flowFaultTest = false, flowFaultFired = false;   // MI_FLOW_FAULT: tests inject one such failure
    bool flowSolver = true;               // dataflow PGS sweep (one launch per iteration); MI_SOLVER=launch selects one launch per colour
    BinInfo bins[kSchedBins]{};           // host copy of the last step's schedule
    uint32_t totalTiles = 0;
    bool xcdSwizzle = false;

    StepScalars hs{};            // host copy of the last step's scalars
    struct Readback { StepScalars sc; uint32_t flags[96]; uint32_t seq; uint32_t pad[3]; };   // seq: written last by k_publish_readback (the host spins on it)
    uint32_t readbackSeq = 0; bool spinReadback = true, stageEvents = false /* time every stage */, stepEvents = false /* time the whole step and the solve stage */;
    Readback* hsPinned = nullptr; // pinned staging for the end-of-step read-back (one async copy, no pageable bounce)
    mi_stage_times timesSum{}; uint32_t timesSteps = 0; uint64_t contactUpdatesSum = 0;   // accumulated since the last mi_world_get_accumulated_stage_times(reset)
    mi_step_counts counts{};
    mi_stage_times times{};
    // two sets of step events, alternating per valid step: the elapsed times of step k are read at the START of step k + 1, after its first launches are
    // enqueued (three hipEventElapsedTime calls cost the host ~10 us it would otherwise spend with the device idle between two steps), or by whoever asks first
    hipEvent_t evSets[2][10]{}; hipEvent_t* ev = evSets[0]; int evSet = 0;
    bool timesPending = false; int timesPendingSet = 0; bool timesPendingStages = false; uint64_t timesPendingUpdates = 0;
    void finishTimes();
    uint32_t numColorsUsed = 0, solveLaunches = 0;
    bool profileSolve = false;            // per-launch HIP events around k_contact_solve (mi_world_step_profiled)
    std::vector<hipEvent_t> profEvents;   // pairs
    uint32_t profLaunches = 0; float profKernelMs = 0.f; uint64_t profSlots = 0, profContacts = 0;
    bool usesGjk = false;   // any capsule / cylinder / hull collider present (decided at upload)
    uint32_t lastNumCells = kMaxCells;   // cells covered by the histogram/scan (host-side bound)
    uint32_t lastCellCap = 0;            // sticky length of the cell scan (a few thousand cells more than the grid has: zeros)
    uint64_t* pairsIn = nullptr;          // bucket-partitioned pair keys of the last step (pairKeys or pairKeysS)

    int init(int dev);
    ~mi_world();
    int recalcProperties();
    int upload();
    int download();
    int stepInternal(const mi_step_settings& s, float dt);
    int runStep(const mi_step_settings& s, float dt, bool speculative);
    // Every enqueue of a step goes through L (launcher.hpp).  Steps of a small scene are replayed as ONE HIP graph when their
    // signature (every launch, pointer, size and scalar) equals that of a captured step: ~35 launches of 2-5 us kernels are bound by
    // the host's launch rate otherwise.
    Launcher L;
    struct StepGraph { uint64_t sig = 0; hipGraphExec_t exec = nullptr; uint64_t lastUse = 0; };
    std::vector<StepGraph> stepGraphs;   // at most kMaxStepGraphs; when full, ALL are dropped with the stream idle (never one next to live ones)
    static constexpr size_t kMaxStepGraphs = 64;
    bool graphsEnabled = true, graphsForAll = false, graphNoEvents = false, graphNoCapture = false; uint32_t graphMaxColliders = 32768;
    uint64_t graphLastSig = 0, graphPrevSig = 0, graphUseClock = 0;   // signatures of the last two steps (the buffer sets alternate: a steady scene repeats with period 2)
    uint32_t graphHits = 0, graphCaptures = 0, graphPlain = 0; bool graphDebug = false;
    std::vector<uint64_t> graphPrevOps, graphPrevOps2;
    DBuf<uint32_t> readbackSeqDev;   // the read-back sequence number lives on the device (k_publish_readback increments it): the launch has constant arguments
    void dropStepGraphs() { for (StepGraph& g : stepGraphs) if (g.exec) (void)hipGraphExecDestroy(g.exec); stepGraphs.clear(); graphLastSig = graphPrevSig = 0; }
    void mirrorSchedule();
    // speculative (single read-back) stepping: upper bounds come from the last valid step
    struct LastCounts { uint32_t numPairs = 0, numManifolds = 0, numContacts = 0, numCells = 0, colorRounds = 0, numSmall = 0, numLarge = 0, numInterPairs = 0, numInteractions = 0, gjkSpan = 0; } last;
    uint32_t gjkWaveMaxPairs = 16384;   // GJK bucket span up to which k_narrow_gjk_wave (one wave per pair) beats lanes + EPA queue
    bool specEnabled = true, haveEstimates = false;
    uint32_t specRetries = 0, specSteps = 0, totalSteps = 0, colorRoundsLaunched = 0;
    uint32_t sapAxis = 0;        // sorting axis for the next step (collision_broad.cpp:443-444), host copy
    // mi_debug_set_solve_order: the next internal step solves these oriented collider pairs (a << 29 | b) sequentially, in this order, and the joints in pool order
    std::vector<uint64_t> debugOrder; std::vector<uint32_t> debugRank; bool debugOrderPending = false;
    int applyDebugOrder();       // all manifolds into the sequential (overflow) colour; their slots in the caller's order
    int orientPairsLikeDebugOrder();   // equal-type pairs listed the other way round are turned (ties on the sweep axis: the reference's orientation follows its endpoint array's history)
};

int mi_world::init(int dev) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(MI_ERR_NO_DEVICE, "no HIP device visible: the stepper has no CPU fallback");
    if (dev < 0 || dev >= n) return fail(MI_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    device = dev;
    HIP_TRY(hipSetDevice(dev));
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (auto& set : evSets) for (auto& e : set) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(scalarsRaw.ensure(sizeof(StepScalars) + (kMaxColorRounds + 2) * sizeof(uint32_t)));
    HIP_TRY(grid.ensure(2));
    HIP_TRY(shards.ensure(1));
    HIP_TRY(hipMemsetAsync(scalarsRaw.p, 0, sizeof(StepScalars) + (kMaxColorRounds + 2) * sizeof(uint32_t), stream));
    HIP_TRY(binInfo.ensure(kSchedBins));
    if (hipHostMalloc((void**)&hsPinned, sizeof(Readback), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();   // no device-visible coherent host memory here: plain pinned staging and the copy path
        HIP_TRY(hipHostMalloc((void**)&hsPinned, sizeof(Readback)));
        spinReadback = false;
    }
    std::memset(hsPinned, 0, sizeof(Readback));
    HIP_TRY(readbackSeqDev.ensure(1)); HIP_TRY(hipMemsetAsync(readbackSeqDev.p, 0, sizeof(uint32_t), stream));
    {   // Replays diverged from plain launches under the HIP 7.0.x runtime (the one PyTorch 2.10 bundles) while the host read the state back
        // through pageable buffers every step; not reproduced since download() stages through pinned memory, cause not established:
        // graphs are used by default from 7.2 on (MI_GRAPH=force overrides; DESIGN.md "Step graphs").
        int ver = 0; if (hipRuntimeGetVersion(&ver) != hipSuccess) ver = 0;
        graphsEnabled = ver >= 70200000;
    }
    if (const char* g = getenv("MI_GRAPH")) { const std::string v(g); if (v == "0") graphsEnabled = false; if (v == "force") graphsEnabled = true; graphsForAll = v == "all"; }   // 0: never replay steps as HIP graphs; all: also the large scenes
    if (const char* g = getenv("MI_GRAPH_MAX_COLLIDERS")) graphMaxColliders = (uint32_t)strtoul(g, nullptr, 0);
    graphDebug = getenv("MI_GRAPH_DEBUG") != nullptr; graphNoEvents = getenv("MI_GRAPH_NOEVENTS") != nullptr; graphNoCapture = getenv("MI_GRAPH_NOCAPTURE") != nullptr;
    if (const char* ps = getenv("MI_POSE_STREAM")) pose.enabled = ps[0] != '0';
    if (const char* sr = getenv("MI_READBACK")) spinReadback = spinReadback && std::string(sr) != "copy";   // MI_READBACK=copy: hipMemcpyAsync + hipStreamSynchronize
    stageEvents = getenv("MI_STAGE_EVENTS") && getenv("MI_STAGE_EVENTS")[0] != '0';   // default: nothing is timed (mi_world_set_stage_timing)
    stepEvents = getenv("MI_STEP_EVENTS") && getenv("MI_STEP_EVENTS")[0] != '0';
    const char* sw = getenv("MI_XCD_SWIZZLE");
    xcdSwizzle = sw && sw[0] == '1';
    const char* sv = getenv("MI_SOLVER");
    flowSolver = !(sv && std::string(sv) == "launch");
    const char* as = getenv("MI_ASYNC");
    specEnabled = !(as && as[0] == '0');
    if (const char* fl = getenv("MI_FLOW_LDS")) flowLds = (uint32_t)strtoul(fl, nullptr, 0);
    // The solver-side body velocities (and the impulse granules) are exchanged between workgroups through 16-byte sc1
    // transactions: memory the L2 never caches (MTYPE_UC) serves them measurably faster than default device memory
    // (solve 0.86 -> 0.79 ms at 262144 bodies).  MI_GVEL_ALLOC / MI_IMP_ALLOC = plain | finegrained | uncached override.
    auto allocFlags = [](const char* env, unsigned dflt) {
        const char* v = getenv(env);
        if (!v) return dflt;
        return std::string(v) == "finegrained" ? (unsigned)hipDeviceMallocFinegrained : std::string(v) == "uncached" ? (unsigned)hipDeviceMallocUncached : 0u;
    };
    gVel.flags = allocFlags("MI_GVEL_ALLOC", hipDeviceMallocUncached);
    imp.flags = allocFlags("MI_IMP_ALLOC", 0u);
    { const char* sv = getenv("MI_SOLVER"); persistSolver = !sv || std::string(sv) == "persist" || std::string(sv) == "persist-global" || std::string(sv) == "blocks";   // default; MI_SOLVER=flow / launch select the other contact solvers
      persistImpLds = !sv || std::string(sv) != "persist-granules";   // persist-granules: impulses as tagged granules as well (what the largest piles get automatically)
      if (sv && std::string(sv) == "persist-granules") persistSolver = true;
      persistMetaLds = !sv || (std::string(sv) != "persist-global" && std::string(sv) != "persist-granules");   // persist-global: slot data always from global memory (the variant larger problems get automatically)
      hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) persistWaves = 4u * (uint32_t)prop.multiProcessorCount;
      if (const char* pw = getenv("MI_PERSIST_WAVES")) persistWaves = (uint32_t)strtoul(pw, nullptr, 0);
      xcdOnly = getenv("MI_PERSIST_XCD_ONLY") ? 1u : 0u;
      if (const char* px = getenv("MI_PERSIST_XCD")) persistXcd = px[0] != '0';
      if (const char* pi = getenv("MI_ISLAND_PRIVATE")) privateIslandsEnabled = pi[0] != '0';   // development / tests: every island through the dataflow
      if (const char* px = getenv("MI_PERSIST_XCD_SINGLE")) persistXcdSingle = px[0] != '0';   // 0: small piles on all XCDs, every body through memory
      xcdFaultTest = getenv("MI_PERSIST_XCD_FAULT") != nullptr;
      blockSolver = sv && std::string(sv) == "blocks";   // opt-in (MI_SOLVER=blocks): bit-identical to the persistent kernel, but not faster yet (DESIGN.md "Spatial blocks in LDS": measured)
      if (const char* bb = getenv("MI_BLOCKS")) blockSolver = blockSolver && bb[0] != '0';
      blockFaultTest = getenv("MI_BLOCK_FAULT") != nullptr;
      blkMaxBlocks = std::max(1u, persistWaves / 4u);   // one block per CU
      if (const char* bn = getenv("MI_BLOCKS_MAX")) blkMaxBlocks = std::max(1u, (uint32_t)strtoul(bn, nullptr, 0));
      mail.flags = hipDeviceMallocUncached;
      (void)hipFuncSetAttribute((const void*)k_contact_solve_blocks, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      flowFaultTest = getenv("MI_FLOW_FAULT") != nullptr;
      if (const char* pm = getenv("MI_PERSIST_XCD_MIN")) xcdMinManifolds = (uint32_t)strtoul(pm, nullptr, 0); }   // smallest manifold count that is partitioned (tests: 1)   // MI_PERSIST_XCD=0: no XCD partitioning (every body through memory)   // development experiment: one XCD's workgroups do all the work
    if (flowLds > 65536) (void)hipFuncSetAttribute((const void*)k_contact_solve_flow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flowLds);
    return MI_OK;
}
mi_world::~mi_world() {
    // nothing may still be writing into the pinned read-back record or reading the device buffers the members' destructors free
    if (stream) (void)hipStreamSynchronize(stream);
    (void)hipDeviceSynchronize();
    if (hsPinned) (void)hipHostFree(hsPinned);
    if (downloadStage) (void)hipHostFree(downloadStage);
    for (auto& hh : pose.host) for (float*& h : hh) if (h) (void)hipHostFree(h);
    if (pose.produced) (void)hipEventDestroy(pose.produced);
    for (hipEvent_t& e : pose.chunkEv) if (e) (void)hipEventDestroy(e);
    if (pose.copyStream) (void)hipStreamDestroy(pose.copyStream);
    if (shard.sentHost) (void)hipHostFree(shard.sentHost);
    if (shard.recvHost) (void)hipHostFree(shard.recvHost);
    for (hipEvent_t& e : shard.exEv) if (e) (void)hipEventDestroy(e);
    shardReleaseComm();
    if (graphDebug) std::fprintf(stderr, "[mi_physics] GJK bucket span of the last step (sticky bound): %u pairs\n", last.gjkSpan);
    if (graphDebug) std::fprintf(stderr, "[mi_physics] step graphs: %u replayed, %u captured, %u plain speculative steps, %llu steps in total\n", graphHits, graphCaptures, graphPlain, (unsigned long long)totalSteps);
    dropStepGraphs();
    if (stream) (void)hipStreamDestroy(stream);
    for (auto& e : profEvents) (void)hipEventDestroy(e);
    for (auto& set : evSets) for (auto& e : set) if (e) (void)hipEventDestroy(e);
    delete heightmap;
    for (HCloth* c : cloths) delete c;
}

// ------------------------------------------------------------------------------------------------
// Mass properties (setup time; src/physics/physics.cpp:1416-1588, src/physics/rigid_body.cpp:29-81)
// ------------------------------------------------------------------------------------------------
static float sphereVol(float r) { float sq = r * r; float sqpi = kPi * sq; return 4.f / 3.f * sqpi * r; }

static MassProps colliderMassProps(const mi_world& w, const mi_collider_desc& d) {
    MassProps r; r.inertia = M3::zero(); r.mass = 0.f;
    const float* f = d.shape;
    float density = d.density;
    switch (d.type) {
        case T_SPHERE: {
            float rad = f[3];
            r.mass = sphereVol(rad) * density;
            r.cog = V3(f[0], f[1], f[2]);
            r.inertia = scale(M3::identity(), 2.f / 5.f * r.mass * rad * rad);
        } break;
        case T_CAPSULE: case T_CYLINDER: {
            V3 a(f[0], f[1], f[2]), b(f[3], f[4], f[5]); float rad = f[6];
            V3 axis = a - b;
            if (axis.y < 0.f) axis = axis * -1.f;
            float height = len(axis);
            axis = axis * (1.f / height);
            M3 rot = quatToMat(rotateFromTo(V3(0.f, 1.f, 0.f), axis));
            float sqR = rad * rad;
            M3 I = M3::zero();
            if (d.type == T_CAPSULE) {
                float sqRpi = kPi * sqR;
                float volume = (4.f / 3.f * sqRpi * rad) + (sqRpi * len(a - b));
                r.mass = volume * density;
                float cylMass = density * sqRpi * height;
                float hemiMass = density * 2.f / 3.f * sqRpi * rad;
                float sqH = height * height;
                I.m11 = sqR * cylMass * 0.5f;
                I.m00 = I.m22 = I.m11 * 0.5f + cylMass * sqH / 12.f;
                float t0 = hemiMass * 2.f * sqR / 5.f;
                I.m11 += t0 * 2.f;
                float t1 = height * 0.5f;
                float t2 = t0 + hemiMass * (t1 * t1 + 3.f / 8.f * sqH);
                I.m00 += t2 * 2.f;
                I.m22 += t2 * 2.f;
            } else {
                float volume = (kPi * rad * rad) * len(a - b);
                r.mass = volume * density;
                float sqH = height * height;
                I.m11 = sqR * r.mass * 0.5f;
                I.m00 = I.m22 = 1.f / 12.f * r.mass * (3.f * sqR + sqH);
            }
            r.cog = (a + b) * 0.5f;
            r.inertia = mul(mul(transpose(rot), I), rot);
        } break;
        case T_AABB: {
            V3 mn(f[0], f[1], f[2]), mx(f[3], f[4], f[5]);
            V3 d0 = mx - mn;
            r.mass = (d0.x * d0.y * d0.z) * density;
            r.cog = (mn + mx) * 0.5f;
            V3 dia = ((mx - mn) * 0.5f) * 2.f;
            r.inertia.m00 = 1.f / 12.f * r.mass * (dia.y * dia.y + dia.z * dia.z);
            r.inertia.m11 = 1.f / 12.f * r.mass * (dia.x * dia.x + dia.z * dia.z);
            r.inertia.m22 = 1.f / 12.f * r.mass * (dia.x * dia.x + dia.y * dia.y);
        } break;
        case T_OBB: {
            Q4 q(f[0], f[1], f[2], f[3]); V3 c(f[4], f[5], f[6]), rad(f[7], f[8], f[9]);
            V3 dia = rad * 2.f;
            r.mass = (dia.x * dia.y * dia.z) * density;
            r.cog = c;
            M3 I = M3::zero();
            I.m00 = 1.f / 12.f * r.mass * (dia.y * dia.y + dia.z * dia.z);
            I.m11 = 1.f / 12.f * r.mass * (dia.x * dia.x + dia.z * dia.z);
            I.m22 = 1.f / 12.f * r.mass * (dia.x * dia.x + dia.y * dia.y);
            M3 rot = quatToMat(q);
            r.inertia = mul(mul(transpose(rot), I), rot);
        } break;
        default: {  // hull: tetrahedron covariance sum (physics.cpp:1517-1579)
            Q4 q(f[0], f[1], f[2], f[3]); V3 pos(f[4], f[5], f[6]);
            const HHull& g = w.hulls[d.hull_geometry];
            const float s60 = 1.f / 60.f, s120 = 1.f / 120.f;
            M3 Cc; Cc.m00 = s60; Cc.m01 = s120; Cc.m02 = s120; Cc.m10 = s120; Cc.m11 = s60; Cc.m12 = s120; Cc.m20 = s120; Cc.m21 = s120; Cc.m22 = s60;
            float totalMass = 0.f; M3 totalCov = M3::zero(); V3 totalCOG;
            for (size_t t = 0; t + 2 < g.tris.size(); t += 3) {
                V3 w1 = pos + rotate(q, g.verts[g.tris[t]]), w2 = pos + rotate(q, g.verts[g.tris[t + 1]]), w3 = pos + rotate(q, g.verts[g.tris[t + 2]]);
                M3 A; A.m00 = w1.x; A.m01 = w2.x; A.m02 = w3.x; A.m10 = w1.y; A.m11 = w2.y; A.m12 = w3.y; A.m20 = w1.z; A.m21 = w2.z; A.m22 = w3.z;
                float dA = det(A);
                M3 cov = mul(mul(scale(A, dA), Cc), transpose(A));
                float mass = 1.f / 6.f * dA;
                V3 cog = (w1 + w2 + w3) * 0.25f;
                totalMass += mass;
                totalCov = add(totalCov, cov);
                totalCOG = totalCOG + cog * mass;
            }
            totalCOG = totalCOG / totalMass;
            M3 Cp = sub(totalCov, scale(outer(totalCOG, totalCOG), totalMass));
            r.cog = totalCOG;
            r.mass = totalMass * density;
            float tr = Cp.m00 + Cp.m11 + Cp.m22;
            r.inertia = sub(scale(M3::identity(), tr), Cp);
            r.inertia = scale(r.inertia, density);
        } break;
    }
    return r;
}

int mi_world::recalcProperties() {
    for (HBody& rb : bodies) {
        if (entities[rb.entity].kind == MI_ENTITY_KINEMATIC) continue;
        const HEntity& e = entities[rb.entity];
        size_t n = e.colliders.size();
        if (!n) { rb.invMass = 1.f; rb.invInertia = M3::identity(); rb.localCOG = V3(); continue; }
        std::vector<MassProps> props(n);
        for (size_t i = 0; i < n; ++i) props[i] = colliderMassProps(*this, colliders[e.colliders[i]].desc);
        M3 inertia = M3::zero(); V3 cog; float mass = 0.f;
        for (size_t i = 0; i < n; ++i) { mass += props[i].mass; cog = cog + props[i].cog * props[i].mass; }
        rb.invMass = 1.f / mass;
        rb.localCOG = cog = cog * rb.invMass;
        for (size_t i = 0; i < n; ++i) {
            V3 r = props[i].cog - cog;
            M3 shift = scale(sub(scale(M3::identity(), dot(r, r)), outer(r, r)), props[i].mass);
            inertia = add(inertia, add(props[i].inertia, shift));
        }
        rb.invInertia = invert(inertia);
    }
    return MI_OK;
}

// ------------------------------------------------------------------------------------------------
// Upload / download
// ------------------------------------------------------------------------------------------------
static float4 h4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

int mi_world::upload() {
    recalcProperties();
    dropStepGraphs();
    pose.valid = false; pose.tablesValid = false;
    shard.prevValid = false; shard.flagsSwapPending = false; shard.flagsOfAStep = false;
    uint32_t nb = (uint32_t)bodies.size(), nc = (uint32_t)colliders.size();
    std::vector<float4> pos(nb), rot(nb), lv(nb), av(nb), fo(nb), to(nb), cim(nb), ii(3 * (size_t)nb), prm(nb);
    for (uint32_t i = 0; i < nb; ++i) {
        const HBody& b = bodies[i];
        pos[i] = h4(b.p1, 0.f); rot[i] = make_float4(b.r1.x, b.r1.y, b.r1.z, b.r1.w);
        lv[i] = h4(b.linVel, 0.f); av[i] = h4(b.angVel, 0.f); fo[i] = h4(b.force, 0.f); to[i] = h4(b.torque, 0.f);
        cim[i] = h4(b.localCOG, b.invMass);
        const M3& m = b.invInertia;
        ii[3 * i] = make_float4(m.m00, m.m01, m.m02, 0.f); ii[3 * i + 1] = make_float4(m.m10, m.m11, m.m12, 0.f); ii[3 * i + 2] = make_float4(m.m20, m.m21, m.m22, 0.f);
        prm[i] = make_float4(b.gravityFactor, b.linDamp, b.angDamp, 0.f);
    }
#define UP(buf, vec, count)                                                                                  \
    do {                                                                                                      \
        HIP_TRY(buf.ensure(std::max<size_t>((count), 1)));                                                    \
        if (count) HIP_TRY(hipMemcpyAsync(buf.p, vec.data(), (count) * sizeof(vec[0]), hipMemcpyHostToDevice, stream)); \
    } while (0)
    UP(bPos, pos, nb); UP(bRot, rot, nb); UP(bLinVel, lv, nb); UP(bAngVel, av, nb); UP(bForce, fo, nb); UP(bTorque, to, nb);
    UP(bCogInvMass, cim, nb); UP(bInvI, ii, 3 * (size_t)nb); UP(bParams, prm, nb);
    { size_t n1 = std::max<size_t>(nb, 1);
      HIP_TRY(bPosN.ensure(n1)); HIP_TRY(bRotN.ensure(n1)); HIP_TRY(bLinVelN.ensure(n1)); HIP_TRY(bAngVelN.ensure(n1)); HIP_TRY(bForceN.ensure(n1)); HIP_TRY(bTorqueN.ensure(n1)); }
    HIP_TRY(gPos.ensure(nb + 1)); HIP_TRY(gInvI.ensure(3 * ((size_t)nb + 1))); HIP_TRY(gVel.ensure(2 * ((size_t)nb + 1)));
    HIP_TRY(bodyTop.ensure(2 * ((size_t)nb + 1))); HIP_TRY(bodyUsed.ensure(nb + 1));
    HIP_TRY(hipMemsetAsync(bodyTop.p, 0, 2 * ((size_t)nb + 1) * sizeof(unsigned long long), stream));   // from here on k_integrate_velocities leaves both cleared for the next step
    HIP_TRY(hipMemsetAsync(bodyUsed.p, 0, ((size_t)nb + 1) * sizeof(unsigned long long), stream));
    HIP_TRY(bndMask.ensure(nb + 1)); HIP_TRY(hipMemsetAsync(bndMask.p, 0, ((size_t)nb + 1) * sizeof(unsigned long long), stream));   // (k_integrate_velocities leaves it cleared)
    HIP_TRY(gVelL.ensure(2 * ((size_t)nb + 1))); HIP_TRY(bodyOwner.ensure(nb + 1)); HIP_TRY(xcdBase.ensure(kSchedBins * 8)); HIP_TRY(keyCount.ensure(kSpatialKeys));

    usesGjk = false;
    for (const HCollider& c : colliders) if (c.desc.type == T_CAPSULE || c.desc.type == T_CYLINDER || c.desc.type == T_HULL) usesGjk = true;
    std::vector<uint32_t> tb(2 * (size_t)nc), obj(nc), cent(nc); std::vector<float4> sh(3 * (size_t)nc), sp(nc), sr(nc), emit(nc);
    // force fields (getForceFieldStates, physics.cpp:759-787): rotated force per field; fields without colliders are global
    usesInteractions = false; globalForce = V3();
    std::vector<float4> lf(ffEntities.size());
    for (size_t i = ffEntities.size(); i-- > 0;) {   // EnTT view order: back to front
        const HEntity& e = entities[ffEntities[i]];
        V3 f = rotate(e.rot, e.force);
        if (!e.colliders.empty()) lf[i] = h4(f, 0.f); else { lf[i] = make_float4(0, 0, 0, 0); globalForce = globalForce + f; }   // (a global field needs no interaction pass: k_integrate_forces adds it)
    }
    for (uint32_t k = 0; k < nc; ++k) {   // world index k <-> creation index nc-1-k (EnTT iterates back to front, physics.cpp:635-641)
        const HCollider& c = colliders[nc - 1 - k];
        const HEntity& e = entities[c.entity];
        tb[2 * k] = c.desc.type; tb[2 * k + 1] = e.rb >= 0 ? (uint32_t)e.rb : kNoBody;
        cent[k] = c.entity;
        obj[k] = e.kind == MI_ENTITY_FORCE_FIELD ? (OBJ_FORCE_FIELD | e.kindIndex << 8) : e.kind == MI_ENTITY_TRIGGER ? (OBJ_TRIGGER | e.kindIndex << 8) : OBJ_STATIC;
        if (e.kind == MI_ENTITY_FORCE_FIELD || e.kind == MI_ENTITY_TRIGGER) usesInteractions = true;
        float s[12]; std::memcpy(s, c.desc.shape, sizeof(s));
        if (c.desc.type == T_HULL) std::memcpy(&s[7], &c.desc.hull_geometry, 4);
        sh[3 * k] = make_float4(s[0], s[1], s[2], s[3]); sh[3 * k + 1] = make_float4(s[4], s[5], s[6], s[7]); sh[3 * k + 2] = make_float4(s[8], s[9], s[10], s[11]);
        sp[k] = h4(e.pos, 0.f); sr[k] = make_float4(e.rot.x, e.rot.y, e.rot.z, e.rot.w);
        // what k_emit_manifolds needs of a collider in one gather (objIndex as k_world_colliders writes it: the body, or the static dummy `nb`)
        const uint32_t obj = e.rb >= 0 ? (uint32_t)e.rb : e.kind == MI_ENTITY_FORCE_FIELD || e.kind == MI_ENTITY_TRIGGER ? e.kindIndex : nb;
        const uint32_t dyn = e.rb >= 0 && bodies[e.rb].invMass != 0.f ? 1u : 0u;
        emit[k] = make_float4(c.desc.restitution, c.desc.friction, 0.f, 0.f);
        std::memcpy(&emit[k].z, &obj, 4); std::memcpy(&emit[k].w, &dyn, 4);
    }
    UP(cObject, obj, nc); UP(localForce, lf, lf.size()); UP(cEntity, cent, nc);
    if (usesInteractions) HIP_TRY(bForceStep.ensure(std::max<size_t>(nb, 1)));
    UP(cTypeBody, tb, 2 * (size_t)nc); UP(cShape, sh, 3 * (size_t)nc); UP(cStaticPos, sp, nc); UP(cStaticRot, sr, nc); UP(cEmit, emit, nc);
    HIP_TRY(wShape.ensure(3 * (size_t)nc + 1)); HIP_TRY(aabbMin.ensure(nc + 1)); HIP_TRY(aabbMax.ensure(nc + 1));
    HIP_TRY(sMin.ensure(nc + 1)); HIP_TRY(sMax.ensure(nc + 1));
    HIP_TRY(largeList.ensure(nc + 1)); HIP_TRY(isLarge.ensure(nc + 1));
    HIP_TRY(cellKeys.ensure(nc + 1)); HIP_TRY(cellRanks.ensure(nc + 1)); HIP_TRY(cellKeysS.ensure(nc + 1)); HIP_TRY(cellValsS.ensure(nc + 1));
    HIP_TRY(cellCount.ensure(kMaxCells)); HIP_TRY(cellLower.ensure(kMaxCells));
    HIP_TRY(hipMemsetAsync(cellCount.p, 0, (size_t)kMaxCells * sizeof(uint32_t), stream));   // from here on every scan clears what it read
    HIP_TRY(blockBounds.ensure(6 * (size_t)divUp(std::max(nc, 1u), 256)));
    HIP_TRY(axisPartials.ensure(kAxisSums * (size_t)divUp(std::max(nc, 1u), 256)));
    // hull geometry pool
    std::vector<float4> ha(2 * hulls.size() + 1), hv; std::vector<uint32_t> hr(2 * hulls.size() + 2);
    for (size_t h = 0; h < hulls.size(); ++h) {
        ha[2 * h] = h4(hulls[h].mn, 0.f); ha[2 * h + 1] = h4(hulls[h].mx, 0.f);
        hr[2 * h] = (uint32_t)hv.size(); hr[2 * h + 1] = (uint32_t)hulls[h].verts.size();
        for (const V3& v : hulls[h].verts) hv.push_back(h4(v, 0.f));
    }
    if (hv.empty()) hv.push_back(make_float4(0, 0, 0, 0));
    std::vector<uint32_t> ht, htr(2 * hulls.size() + 2);
    for (size_t h = 0; h < hulls.size(); ++h) { htr[2 * h] = (uint32_t)(ht.size() / 3); htr[2 * h + 1] = (uint32_t)(hulls[h].tris.size() / 3); ht.insert(ht.end(), hulls[h].tris.begin(), hulls[h].tris.end()); }
    if (ht.empty()) ht.push_back(0u);
    UP(hullTris, ht, ht.size()); UP(hullTriRanges, htr, htr.size());
    UP(hullAabb, ha, ha.size()); UP(hullVerts, hv, hv.size()); UP(hullRanges, hr, hr.size());
#undef UP
    int rc = joints.upload(*this, stream);
    if (rc != MI_OK) return rc;
    if (heightmap) {
        HIP_TRY(hmPacked.ensure(nc + 1)); HIP_TRY(hmScan.ensure(nc + 1)); HIP_TRY(hmSlow.ensure(nc + 1));
        { static const bool stashOn = !(std::getenv("MI_HM_STASH") && std::getenv("MI_HM_STASH")[0] == '0'); if (stashOn) HIP_TRY(hmStash.ensure((size_t)(nc + 1) * kHmStash)); }
        rc = uploadHeightmap(); if (rc != MI_OK) return rc;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    topologyDirty = false; hostStale = false; haveEstimates = false; gridValid = false;
    if (seamTiling.on && !shard.enabled) { int rc = shardBuildRoots(); if (rc != MI_OK) return rc; }
    if (shard.enabled) {   // the previous step's counts say nothing about the new topology
        int rc = shardBuildRoots(); if (rc != MI_OK) return rc;
        std::vector<uint8_t> kn(std::max(nb, 1u), 1u);
        for (uint32_t i = 0; i < nb; ++i) kn[i] = bodies[i].shardKnown;
        HIP_TRY(shard.known.ensure(kn.size())); HIP_TRY(hipMemcpy(shard.known.p, kn.data(), kn.size(), hipMemcpyHostToDevice));
    }
    return MI_OK;
}

// heightmap_collider_chunk::setHeights (heightmap_collider.cpp:42-114): the min/max pyramid of every chunk that has heights,
// built on the host at upload time (setup cost, not per step) and pooled on the device next to the heights.
int mi_world::uploadHeightmap() {
    HHeightmap& h = *heightmap;
    const uint32_t nchunks = h.chunksPerDim * h.chunksPerDim;
    if (h.dirty) {
        hmHostSlots.assign(nchunks, 0xFFFFFFFFu);
        uint32_t used = 0;
        for (uint32_t c = 0; c < nchunks; ++c) if (!h.heights[c].empty()) hmHostSlots[c] = used++;
        hmHostHeights.assign((size_t)std::max(used, 1u) * kHmVerts * kHmVerts, 0);
        std::vector<uint32_t> mips((size_t)std::max(used, 1u) * kHmMipEntries, 0u);
        for (uint32_t c = 0; c < nchunks; ++c) {
            if (hmHostSlots[c] == 0xFFFFFFFFu) continue;
            const uint16_t* src = h.heights[c].data();
            std::copy(src, src + kHmVerts * kHmVerts, hmHostHeights.begin() + (size_t)hmHostSlots[c] * kHmVerts * kHmVerts);
            uint32_t* mp = mips.data() + (size_t)hmHostSlots[c] * kHmMipEntries;
            for (uint32_t z = 0; z < kHmSegs; ++z)
                for (uint32_t x = 0; x < kHmSegs; ++x) {
                    uint16_t v[4] = {src[kHmVerts * z + x], src[kHmVerts * (z + 1) + x], src[kHmVerts * z + x + 1], src[kHmVerts * (z + 1) + x + 1]};
                    mp[z * kHmSegs + x] = (uint32_t)*std::min_element(v, v + 4) | ((uint32_t)*std::max_element(v, v + 4) << 16);
                }
            for (uint32_t mip = 1; mip < 8; ++mip) {
                const uint32_t n = kHmSegs >> mip, rs = n * 2;
                const uint32_t* rd = mp + hmMipOffset(mip - 1); uint32_t* wr = mp + hmMipOffset(mip);
                for (uint32_t z = 0; z < n; ++z)
                    for (uint32_t x = 0; x < n; ++x) {
                        uint32_t q[4] = {rd[rs * (2 * z) + 2 * x], rd[rs * (2 * z + 1) + 2 * x], rd[rs * (2 * z) + 2 * x + 1], rd[rs * (2 * z + 1) + 2 * x + 1]};
                        uint32_t mn = 0xFFFFu, mx = 0u;
                        for (uint32_t k = 0; k < 4; ++k) { mn = std::min(mn, q[k] & 0xFFFFu); mx = std::max(mx, q[k] >> 16); }
                        wr[z * n + x] = mn | (mx << 16);
                    }
            }
        }
        HIP_TRY(hmHeights.ensure(hmHostHeights.size())); HIP_TRY(hmMips.ensure(mips.size())); HIP_TRY(hmChunkSlot.ensure(nchunks));
        HIP_TRY(hipMemcpyAsync(hmHeights.p, hmHostHeights.data(), hmHostHeights.size() * sizeof(uint16_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(hmMips.p, mips.data(), mips.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(hmChunkSlot.p, hmHostSlots.data(), nchunks * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));   // `mips` is a local
        h.dirty = false;
    }
    HeightmapParams& q = hmParams;   // heightmap_collider_component ctor + update (heightmap_collider.cpp:5-19)
    q.heights = hmHeights.p; q.mips = hmMips.p; q.chunkSlot = hmChunkSlot.p; q.chunksPerDim = h.chunksPerDim;
    q.chunkSize = h.chunkSize; q.invChunkSize = 1.f / h.chunkSize; q.chunkScale = h.chunkSize / (float)kHmSegs;
    q.heightScale = h.amplitudeScale / 65535.f; q.invAmplitudeScale = 1.f / h.amplitudeScale;
    q.minX = h.minCorner.x; q.minY = h.minCorner.y; q.minZ = h.minCorner.z; q.restitution = h.restitution; q.friction = h.friction;
    return MI_OK;
}

// ---- poses for the caller (PoseStream)
bool mi_world::posesPossible(bool physics, float* t) const {
    const bool follow = physics || transformsFollowPhysics, lerpNow = !follow && lerpPending;
    if (!pose.enabled || !hostStale || topologyDirty || bodies.empty() || shard.enabled) return false;
    if (!follow && !(lerpNow && p0OnDevice)) return false;
    *t = follow ? -1.f : lerpT;
    return true;
}
int mi_world::posesProduce(float t, bool fromNextState) {
    HIP_TRY(hipSetDevice(device));
    PoseStream& ps = pose;
    const uint32_t n = (uint32_t)entities.size();
    if (!ps.copyStream) {
        HIP_TRY(hipStreamCreateWithFlags(&ps.copyStream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ps.produced, hipEventDisableTiming));
        for (hipEvent_t& e : ps.chunkEv) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (ps.copyInFlight) { HIP_TRY(hipStreamWaitEvent(stream, ps.chunkEv[kPoseChunks - 1], 0)); ps.copyInFlight = false; }   // the device-side rows are about to be rewritten
    if (!ps.tablesValid || ps.tableCount != n) {
        std::vector<int> eb(n); ps.noBody.clear();
        for (uint32_t i = 0; i < n; ++i) { eb[i] = entities[i].rb; if (eb[i] < 0) ps.noBody.push_back(i); }
        HIP_TRY(ps.entBody.ensure(std::max<size_t>(n, 1)));
        HIP_TRY(hipMemcpyAsync(ps.entBody.p, eb.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));   // (eb is pageable and goes away)
        ps.tablesValid = true; ps.tableCount = n;
    }
    const size_t floats = 7u * (size_t)n;
    HIP_TRY(ps.out.ensure(std::max<size_t>(floats, 4)));
    if (floats > ps.hostCap) {
        HIP_TRY(hipStreamSynchronize(ps.copyStream));
        for (auto& hh : ps.host) for (float*& h : hh) { if (h) (void)hipHostFree(h); h = nullptr; }
        ps.hostCap = 0;
        const size_t cap = floats + floats / 4 + 16;
        for (auto& hh : ps.host) for (float*& h : hh) HIP_TRY(hipHostMalloc((void**)&h, cap * sizeof(float)));
        ps.hostCap = cap;
    }
    const int fl = t < 0.f ? 1 : 0;
    if (!(ps.valid == false && ps.retrySameStep && ps.flavour == fl)) ps.cur[fl] ^= 1;   // (a step that is re-run produces into the same rows again: the other set may still be in a caller's hands)
    ps.retrySameStep = false;
    float* outR = ps.out.p; float* outP = ps.out.p + 4u * (size_t)n;
    hipLaunchKernelGGL(k_entity_poses, dim3(divUp(n, 256u)), dim3(256), 0, stream, n, ps.entBody.p, fromNextState ? bPosN.p : bPos.p, fromNextState ? bRotN.p : bRot.p, bPos0.p, bRot0.p, t, outP, outR);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ps.produced, stream));
    HIP_TRY(hipStreamWaitEvent(ps.copyStream, ps.produced, 0));
    const size_t bytes = floats * sizeof(float);
    for (uint32_t k = 0; k <= kPoseChunks; ++k) ps.chunkOff[k] = k == kPoseChunks ? bytes : std::min(bytes, ((bytes * k / kPoseChunks) + 4095u) & ~(size_t)4095u);
    for (uint32_t k = 0; k < kPoseChunks; ++k) {
        const size_t off = ps.chunkOff[k], len = ps.chunkOff[k + 1] - off;
        if (len) HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(ps.host[fl][ps.cur[fl]]) + off, reinterpret_cast<const char*>(ps.out.p) + off, len, hipMemcpyDeviceToHost, ps.copyStream));
        HIP_TRY(hipEventRecord(ps.chunkEv[k], ps.copyStream));
    }
    ps.flavour = fl;
    ps.valid = true; ps.copyInFlight = true; ps.steps = totalSteps; ps.t = t; ps.n = n; ps.consumed = false;
    return MI_OK;
}
// p / r: the caller's arrays ([n][3], [n][4]; either may be null) — a few threads copy each piece as it lands; viewP / viewR: the pinned rows themselves
int mi_world::posesFetch(float* p, float* r, const float** viewP, const float** viewR) {
    PoseStream& ps = pose;
    const uint32_t n = ps.n;
    float* src = ps.host[ps.flavour][ps.cur[ps.flavour]];
    const size_t rBytes = 16u * (size_t)n;
    if (viewP || viewR) {
        HIP_TRY(hipEventSynchronize(ps.chunkEv[kPoseChunks - 1]));
        for (uint32_t i : ps.noBody) {
            const HEntity& e = entities[i];
            float* rr = src + 4u * (size_t)i; float* pp = src + 4u * (size_t)n + 3u * (size_t)i;
            rr[0] = e.rot.x; rr[1] = e.rot.y; rr[2] = e.rot.z; rr[3] = e.rot.w; pp[0] = e.pos.x; pp[1] = e.pos.y; pp[2] = e.pos.z;
        }
        if (viewP) *viewP = src + 4u * (size_t)n;
        if (viewR) *viewR = src;
    }
    if (p || r) {
        const size_t total = ps.chunkOff[kPoseChunks];
        const uint32_t threads = total >= (1u << 20) ? std::min<uint32_t>(4u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        std::atomic<int> err{0};
        auto work = [&](uint32_t tid) {
            for (uint32_t k = 0; k < kPoseChunks; ++k) {
                if (hipEventSynchronize(ps.chunkEv[k]) != hipSuccess) { err = 1; return; }
                const size_t lo0 = ps.chunkOff[k], len = ps.chunkOff[k + 1] - lo0;
                size_t lo = lo0 + ((len * tid / threads) & ~(size_t)63), hi = tid + 1 == threads ? lo0 + len : lo0 + ((len * (tid + 1) / threads) & ~(size_t)63);
                if (lo < rBytes && r) { const size_t e = std::min(hi, rBytes); std::memcpy(reinterpret_cast<char*>(r) + lo, reinterpret_cast<const char*>(src) + lo, e - lo); }
                if (hi > rBytes && p) { const size_t b = std::max(lo, rBytes); std::memcpy(reinterpret_cast<char*>(p) + (b - rBytes), reinterpret_cast<const char*>(src) + b, hi - b); }
            }
        };
        std::vector<std::thread> pool;
        for (uint32_t tdx = 1; tdx < threads; ++tdx) { try { pool.emplace_back(work, tdx); } catch (...) { for (std::thread& th : pool) th.join(); pool.clear(); for (uint32_t q = 1; q < threads; ++q) work(q); break; } }
        work(0);
        for (std::thread& th : pool) th.join();
        if (err) return fail(MI_ERR_DEVICE, "pose copy");
        for (uint32_t i : ps.noBody) {
            const HEntity& e = entities[i];
            if (p) { p[3 * (size_t)i] = e.pos.x; p[3 * (size_t)i + 1] = e.pos.y; p[3 * (size_t)i + 2] = e.pos.z; }
            if (r) { r[4 * (size_t)i] = e.rot.x; r[4 * (size_t)i + 1] = e.rot.y; r[4 * (size_t)i + 2] = e.rot.z; r[4 * (size_t)i + 3] = e.rot.w; }
        }
    }
    ps.consumed = true; ps.wanted = true;
    return MI_OK;
}
// end of mi_world_step / mi_world_step_fixed: a caller that asked for the poses after the previous step gets this step's under way now
bool mi_world::posesWantedAhead() {
    PoseStream& ps = pose;
    if (!ps.enabled || !ps.wanted) return false;
    if (ps.valid && !ps.consumed) { ps.wanted = false; return false; }   // nobody read the last ones
    return true;
}
// before the last internal step of a stepping call: what the poses will be afterwards is known (physics_transform1, or the lerp with the factor
// the accumulator will leave), so that step enqueues them itself — the host's ~0.1 ms of launches and copies hides behind the step's kernels
void mi_world::posesArm(bool lerpAfterwards, float lerpTAfterwards) {
    poseArm = PoseArm{};
    if (!posesWantedAhead() || topologyDirty || bodies.empty() || shard.enabled) return;
    const bool follow = pose.askedPhysics || !lerpAfterwards;
    poseArm.armed = true; poseArm.t = follow ? -1.f : lerpTAfterwards;
}
int mi_world::posesAfterStep() {
    PoseStream& ps = pose;
    const bool inside = poseArm.armed && poseArm.done;
    poseArm = PoseArm{};
    if (inside) return MI_OK;
    if (!posesWantedAhead()) return MI_OK;
    float t;
    if (!posesPossible(ps.askedPhysics, &t)) return MI_OK;
    if (ps.valid && ps.steps == totalSteps && ps.t == t && ps.n == (uint32_t)entities.size() && ps.tablesValid) return MI_OK;
    ++ps.produced_ahead;
    return posesProduce(t, false);
}

int mi_world::download() {
    if (!hostStale) return MI_OK;
    uint32_t nb = (uint32_t)bodies.size();
    if (nb) {
        // pinned staging, kept: pageable std::vectors made this 2.5 ms per call at 57 k bodies (allocation + staged copies), which was
        // the largest part of a batched learning step once mi_world_step stopped downloading
        const size_t rows = (p0OnDevice ? 8u : 6u) * (size_t)nb;
        if (rows > downloadStageCap) {
            if (downloadStage) (void)hipHostFree(downloadStage);
            downloadStage = nullptr; downloadStageCap = 0;
            HIP_TRY(hipHostMalloc((void**)&downloadStage, (rows + rows / 4) * sizeof(float4)));
            downloadStageCap = rows + rows / 4;
        }
        float4 *pos = downloadStage, *rot = pos + nb, *lv = rot + nb, *av = lv + nb, *fo = av + nb, *to = fo + nb, *pos0 = to + nb, *rot0 = pos0 + nb;
        HIP_TRY(hipMemcpyAsync(pos, bPos.p, nb * 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(rot, bRot.p, nb * 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(lv, bLinVel.p, nb * 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(av, bAngVel.p, nb * 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(fo, bForce.p, nb * 16, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(to, bTorque.p, nb * 16, hipMemcpyDeviceToHost, stream));
        if (p0OnDevice) {
            HIP_TRY(hipMemcpyAsync(pos0, bPos0.p, nb * 16, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(rot0, bRot0.p, nb * 16, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        hostParallelFor(nb, [&](uint32_t i) {   // a body writes its own HBody and its own entity only
            HBody& b = bodies[i];
            if (p0OnDevice) { b.p0 = V3(pos0[i].x, pos0[i].y, pos0[i].z); b.r0 = Q4(rot0[i].x, rot0[i].y, rot0[i].z, rot0[i].w); }
            b.p1 = V3(pos[i].x, pos[i].y, pos[i].z); b.r1 = Q4(rot[i].x, rot[i].y, rot[i].z, rot[i].w);
            b.linVel = V3(lv[i].x, lv[i].y, lv[i].z); b.angVel = V3(av[i].x, av[i].y, av[i].z);
            b.force = V3(fo[i].x, fo[i].y, fo[i].z); b.torque = V3(to[i].x, to[i].y, to[i].z);
            // after n x physicsStepInternal without interpolation (mi_world_step_fixed) the transform is physics_transform1 (physics.cpp:1408-1411)
            if (transformsFollowPhysics) { HEntity& e = entities[b.entity]; e.pos = b.p1; e.rot = b.r1; }
            else if (lerpPending) {   // lerp(trs): nlerp on the quaternion (src/core/math.h:673-682)
                HEntity& e = entities[b.entity]; const float t = lerpT;
                e.pos = lerp(b.p0, b.p1, t);
                e.rot = normalize(Q4(b.r0.x + t * (b.r1.x - b.r0.x), b.r0.y + t * (b.r1.y - b.r0.y), b.r0.z + t * (b.r1.z - b.r0.z), b.r0.w + t * (b.r1.w - b.r0.w)));
            }
        });
        p0OnDevice = false; lerpPending = false;
        if (shard.enabled && shard.known.p && shard.known.cap >= nb) {
            std::vector<uint8_t> kn(nb);
            HIP_TRY(hipMemcpy(kn.data(), shard.known.p, nb, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < nb; ++i) bodies[i].shardKnown = kn[i];
        }
    }
    hostStale = false;
    return MI_OK;
}

// ------------------------------------------------------------------------------------------------
// One internal step (physicsStepInternal, src/physics/physics.cpp:1180-1362)
// ------------------------------------------------------------------------------------------------
__global__ void k_reset_scalars(StepScalars* sc, Shards* sh, uint32_t* roundFlags, uint32_t* keyCount) {
    uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < kSpatialKeys; i += blockDim.x) keyCount[i] = 0u;
    for (uint32_t i = t; i < sizeof(Shards) / 4u; i += blockDim.x) reinterpret_cast<uint32_t*>(sh)[i] = 0u;
    for (uint32_t i = t; i < kMaxColorRounds + 2u; i += blockDim.x) roundFlags[i] = 0u;
    if (t == 0) {
        sc->numDead = 0; sc->shardOwned[0] = sc->shardOwned[1] = sc->shardOwned[2] = 0;
        for (int q = 0; q < 8; ++q) sc->shardSent[q] = 0;
        sc->seamStats[0] = sc->seamStats[1] = sc->seamStats[2] = 0;
        sc->extentSum = 0.0; sc->largeThreshold = 0.f; sc->numLarge = 0; sc->numPairs = 0; sc->numOverlaps = 0; sc->numManifolds = 0; sc->numContacts = 0; sc->solveError = 0;
        sc->specOverflow = 0; sc->totalTiles = 0; sc->totalCt = 0; sc->colorPending = 0; sc->partitioned = 0; sc->gjkLo = 0; sc->gjkHi = 0; sc->numCells = 0; sc->numPairsFound = 0; sc->numEvents = 0; sc->numInterPairs = 0; sc->numInteractions = 0; sc->numHmContacts = 0; sc->numHmColliders = 0; sc->numEpa = 0;
        for (int q = 0; q < 16; ++q) sc->boxHitCount[q] = 0;
        for (int q = 0; q < 8; ++q) { sc->xcdCount[q] = 0; sc->xccOf[q] = 0xFFFFFFFFu; }
        for (int a = 0; a < 3; ++a) { sc->boundsMin[a] = 0x7FFFFFFF; sc->boundsMax[a] = (int)0x80000000; }
    }
    if (t < 24) { sc->bucketHist[t] = 0; sc->bucketCursor[t] = 0; }
}
// End-of-step read-back without a copy engine round trip and without a driver wake-up: one workgroup writes the scalars and the
// colouring round flags straight into pinned host memory, fences at system scope and then publishes the step's sequence number,
// which the host thread spins on.
__global__ __launch_bounds__(256) void k_publish_readback(const uint32_t* __restrict__ src, uint32_t words, uint32_t* dstHost, uint32_t seqWord, uint32_t* seqDev) {
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dstHost[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t seq = *seqDev + 1u; if (seq == 0u) seq = 1u;   // never 0 (the host counts the same way)
        *seqDev = seq;
        __hip_atomic_store(dstHost + seqWord, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_reset_pair_counters(StepScalars* sc, Shards* sh) {
    uint32_t t = threadIdx.x;
    if (t == 0) { sc->numPairs = 0; sc->numOverlaps = 0; sc->numInterPairs = 0; }
    if (t < 24) sc->bucketHist[t] = 0;
    for (uint32_t k = 0; k < kShards; ++k) { if (t == 0) sh->c[k].numOverlaps = 0; if (t < 24) sh->c[k].bucketHist[t] = 0; }   // the pair pass's counters only (owned[] of a sharded world stays)
}

// One internal step.  Two ways to run it:
//   * synchronous: the host reads the pair count, the manifold count and the schedule back as they appear and sizes
//     every buffer and launch exactly (first step, after a topology change, per-colour solver path, or as the retry);
//   * speculative (default for every later step): no read-back until the very end.  Every kernel takes its sizes from
//     StepScalars on the device; the host only needs UPPER BOUNDS for launch grids, scan lengths and capacities, and
//     takes them from the previous step's counts (+12.5 %).  The end-of-step read-back validates the bounds (and that
//     the colouring converged, no overflow colour appeared, ...).  Nothing persistent is modified before
//     k_integrate_velocities, which writes into the second body-state buffer set, so an invalid speculation is simply
//     re-run synchronously from the untouched state.  This removes ~0.35 ms of idle GPU time per step at 262144 bodies.
enum { STEP_RETRY = 1 };

int mi_world::stepInternal(const mi_step_settings& settings, float dt) {
    HIP_TRY(hipSetDevice(device));
    if (topologyDirty) { int rc = download(); if (rc != MI_OK) return rc; rc = upload(); if (rc != MI_OK) return rc; haveEstimates = false; }
    else if (joints.podsDirty()) { int rc = joints.uploadPods(stream); if (rc != MI_OK) return rc; HIP_TRY(hipStreamSynchronize(stream)); }
    if (bodies.empty()) return cloths.empty() ? MI_OK : stepCloths(dt);   // physics.cpp:1184-1189: cloth alone still steps
    if (shard.stepOpen) shard.prevValid = false;   // the previous step ended in an error: what its kernels left behind is not what the flags describe
    shard.stepOpen = true;
    if (debugOrderPending && (heightmap || shard.enabled)) { debugOrderPending = false; debugOrder.clear(); shard.stepOpen = false; return fail(MI_ERR_UNSUPPORTED, "mi_debug_set_solve_order: not with heightmap terrain or sharding"); }   // (terrain / sharding came after the order was set: the order is dropped, later steps run)
    const bool exactSeam = shard.enabled && shard.exact;   // every rank must take the same path through the step (its sweeps end in an exchange): no speculation
    shard.sweepsDone = 0;
    const bool spec = specEnabled && haveEstimates && flowSolver && !launchFallbackSteps && !debugOrderPending && !exactSeam && (!usesInteractions || last.numInteractions <= 32768u);   // (triggers / force fields: ordered and applied on the device while there are at most 32 k interactions)
    ++totalSteps; if (spec) ++specSteps;
    int rc = runStep(settings, dt, spec);
    // a step that asks to be re-run has written nothing persistent; each re-run is synchronous and one rung further down the ladder
    // speculative -> exact sizes -> unpartitioned -> dispatch-ordered dataflow kernel -> one launch per colour
    for (int attempt = 0; rc == STEP_RETRY && attempt < 6; ++attempt) {
        if (poseArm.done) { poseArm.done = false; pose.valid = false; pose.retrySameStep = true; --pose.produced_ahead; }
        if (exactSeam && shard.sweepsDone) return fail(MI_ERR_DEVICE, "exact seam: the step would have to be re-run after sweeps were already exchanged with the neighbours");
        ++specRetries; rc = runStep(settings, dt, false);
    }
    if (rc == STEP_RETRY) return fail(MI_ERR_DEVICE, "step could not be completed on any solver path");
    if (rc == MI_OK && launchFallbackSteps) --launchFallbackSteps;
    if (rc == MI_OK && blkDisabledSteps) --blkDisabledSteps;
    debugOrderPending = false; debugOrder.clear();   // (one step only, whatever became of it)
    if (rc == MI_OK && shard.enabled) rc = shardExchange();
    if (rc == MI_OK && !cloths.empty()) rc = stepCloths(dt);   // after the rigid bodies (physics.cpp:1352-1358); once per VALID step: cloth state is updated in place
    return rc;
}

// cloth_component::applyWindForce(globalForceField) + simulate(...) for every cloth: one launch, one workgroup per cloth.
int mi_world::stepCloths(float dt) {
    for (HCloth* c : cloths) {
        if (c->desc.total_mass != c->oldTotalMass || c->desc.stiffness != c->oldStiffness) {   // recalculateProperties (cloth.cpp:299-317)
            const uint32_t n = c->desc.grid_size_x * c->desc.grid_size_y;
            const float invMassPerParticle = (float)n / c->desc.total_mass;
            for (float& im : c->invMasses) im = (im != 0.f) ? invMassPerParticle : 0.f;
            c->desc.stiffness = clampr(c->desc.stiffness, 0.01f, 1.f);
            const float invStiffness = 1.f / c->desc.stiffness;
            for (size_t k = 0; k < c->pairs.size(); ++k) c->restInvMass[k].y = (c->invMasses[c->pairs[k].x] + c->invMasses[c->pairs[k].y]) * invStiffness;
            c->oldTotalMass = c->desc.total_mass; c->oldStiffness = c->desc.stiffness;
            c->constraintsDirty = true;
            // the inverse masses ride in pos.w
            std::vector<float4> p(n);
            HIP_TRY(hipMemcpyAsync(p.data(), c->pos.p, n * sizeof(float4), hipMemcpyDeviceToHost, stream)); HIP_TRY(hipStreamSynchronize(stream));
            for (uint32_t i = 0; i < n; ++i) p[i].w = c->invMasses[i];
            HIP_TRY(hipMemcpyAsync(c->pos.p, p.data(), n * sizeof(float4), hipMemcpyHostToDevice, stream)); HIP_TRY(hipStreamSynchronize(stream));
        }
        if (c->constraintsDirty) {
            HIP_TRY(hipMemcpyAsync(c->dRestInvMass.p, c->restInvMass.data(), c->restInvMass.size() * sizeof(float2), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            c->constraintsDirty = false; clothsDirty = true;
        }
    }
    if (clothsDirty) {
        std::vector<ClothDev> d(cloths.size());
        for (size_t i = 0; i < cloths.size(); ++i) {
            HCloth& c = *cloths[i];
            ClothDev& o = d[i];
            o.pos = c.pos.p; o.prev = c.prev.p; o.vel = c.vel.p; o.force = c.force.p; o.pairs = c.dPairs.p; o.restInvMass = c.dRestInvMass.p; o.temp = c.temp.p; o.order = c.dOrder.p;
            std::memcpy(o.colourOffsets, c.colourOffsets, sizeof(o.colourOffsets));
            o.gridX = c.desc.grid_size_x; o.gridY = c.desc.grid_size_y; o.numConstraints = (uint32_t)c.pairs.size();
            o.gravityFactor = c.desc.gravity_factor; o.damping = c.desc.damping;
        }
        HIP_TRY(clothDescs.ensure(d.size()));
        HIP_TRY(hipMemcpyAsync(clothDescs.p, d.data(), d.size() * sizeof(ClothDev), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        clothsDirty = false;
    }
    k_cloth_step<<<(uint32_t)cloths.size(), 256, 0, stream>>>(clothDescs.p, make_float3(globalForce.x, globalForce.y, globalForce.z), dt, clothIterations[0], clothIterations[1], clothIterations[2]);
    HIP_TRY(hipGetLastError());
    return MI_OK;
}

// handleNonCollisionInteractions (physics.cpp:952-1039) for the AABB overlaps the pair pass collected between rigid-body colliders
// and trigger / force-field colliders.  Synchronous: the (few) interactions come back to the host, which puts them into the
// canonical order (body, other collider, body collider); localized force fields then add up per body on the device in that
// order, trigger overlaps are de-duplicated per (trigger entity, body entity) and diffed against the previous step.
int mi_world::interactions(std::vector<mi_event>& out) {
    const uint32_t nb = (uint32_t)bodies.size();
    hipStream_t st = stream;
    HIP_TRY(hipMemcpyAsync(bForceStep.p, bForce.p, (size_t)nb * sizeof(float4), hipMemcpyDeviceToDevice, st));
    std::vector<DeviceInteraction> list;
    if (hs.numInterPairs) {
        HIP_TRY(interList.ensure(hs.numInterPairs));
        HullSet hset{hullVerts.p, hullRanges.p};
        k_overlap<<<divUp(hs.numInterPairs, 64), 64, 0, st>>>(scalarsPtr(), hs.numInterPairs, interKeys.p, wShape.p, aabbMin.p, aabbMax.p, hset, interList.p);
        HIP_TRY(hipMemcpyAsync(&hs, scalarsPtr(), sizeof(StepScalars), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        list.resize(hs.numInteractions);
        if (!list.empty()) {
            HIP_TRY(hipMemcpyAsync(list.data(), interList.p, list.size() * sizeof(DeviceInteraction), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        std::sort(list.begin(), list.end(), [](const DeviceInteraction& x, const DeviceInteraction& y) {
            if (x.body != y.body) return x.body < y.body;
            if (x.otherCollider != y.otherCollider) return x.otherCollider < y.otherCollider;
            return x.rbCollider < y.rbCollider;
        });
    }
    std::vector<uint2> fields;
    for (const DeviceInteraction& in : list) if ((in.other >> 28) == OBJ_FORCE_FIELD) fields.push_back(make_uint2(in.body, in.other & 0x0FFFFFFFu));
    if (!fields.empty()) {
        HIP_TRY(fieldList.ensure(fields.size()));
        HIP_TRY(hipMemcpyAsync(fieldList.p, fields.data(), fields.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
        k_apply_fields<<<divUp((uint32_t)fields.size(), 256), 256, 0, st>>>((uint32_t)fields.size(), fieldList.p, localForce.p, bForceStep.p);
        HIP_TRY(hipStreamSynchronize(st));   // `fields` is pageable host memory
    }
    return triggerEventsFrom(list, out);
}
int mi_world::triggerEventsFrom(const std::vector<DeviceInteraction>& list, std::vector<mi_event>& out) {
    nextTriggerOverlaps.clear();
    for (const DeviceInteraction& in : list) {
        const uint32_t type = in.other >> 28, index = in.other & 0x0FFFFFFFu;
        if (type == OBJ_TRIGGER) nextTriggerOverlaps.push_back(((uint64_t)triggerEntities[index] << 32) | (uint64_t)bodies[in.body].entity);
    }
    std::sort(nextTriggerOverlaps.begin(), nextTriggerOverlaps.end());
    nextTriggerOverlaps.erase(std::unique(nextTriggerOverlaps.begin(), nextTriggerOverlaps.end()), nextTriggerOverlaps.end());
    if (!eventsEnabled) { nextTriggerOverlaps.clear(); return MI_OK; }
    auto emit = [&](uint64_t key, uint32_t type) {
        mi_event e{}; e.type = type; e.entity_a = (uint32_t)(key >> 32); e.entity_b = (uint32_t)key; e.collider_a = e.collider_b = 0xFFFFFFFFu;
        out.push_back(e);
    };
    const std::vector<uint64_t>& prev = prevTriggerOverlaps;
    size_t p = 0, t = 0;
    while (p < prev.size() && t < nextTriggerOverlaps.size()) {
        uint64_t pk = prev[p], tk = nextTriggerOverlaps[t];
        if (pk == tk) { ++p; ++t; }
        else if (pk < tk) { emit(pk, MI_EVENT_TRIGGER_LEAVE); ++p; }
        else { emit(tk, MI_EVENT_TRIGGER_ENTER); ++t; }
    }
    while (p < prev.size()) emit(prev[p++], MI_EVENT_TRIGGER_LEAVE);
    while (t < nextTriggerOverlaps.size()) emit(nextTriggerOverlaps[t++], MI_EVENT_TRIGGER_ENTER);
    return MI_OK;
}
// Speculative form: k_overlap over the bound of the candidate pairs, canonical order by a device rank sort, force fields applied from the
// sorted list — nothing comes back to the host inside the step; the trigger overlaps are taken from the sorted list after the read-back.
int mi_world::interactionsDevice(uint32_t interPairBound) {
    const uint32_t nb = (uint32_t)bodies.size();
    hipStream_t st = stream;
    HIP_TRY(L.memcpyAsync(bForceStep.p, bForce.p, (size_t)nb * sizeof(float4), hipMemcpyDeviceToDevice, st));
    if (!interPairBound) return MI_OK;
    const uint32_t cap = (uint32_t)interList.cap;
    HullSet hset{hullVerts.p, hullRanges.p};
    L.launch(k_overlap, dim3(divUp(interPairBound, 64)), dim3(64), 0, st, scalarsPtr(), cap, interKeys.p, wShape.p, aabbMin.p, aabbMax.p, hset, interList.p);
    L.launch(k_inter_sort, dim3(divUp(cap, 256)), dim3(256), 0, st, scalarsPtr(), cap, interList.p, interSorted.p);
    L.launch(k_apply_fields_sorted, dim3(divUp(cap, 256)), dim3(256), 0, st, scalarsPtr(), cap, interSorted.p, localForce.p, bForceStep.p);
    return MI_OK;
}

int mi_world::runStep(const mi_step_settings& settings, float dt, bool spec) {
    const uint32_t nb = (uint32_t)bodies.size(), nc = (uint32_t)colliders.size();
    const uint32_t B = 256;
    StepScalars* sc = scalarsPtr();
    hipStream_t st = stream;
    int evi = 0;
    static const bool debugSync = std::getenv("MI_DEBUG_SYNC") != nullptr;   // development: find the stage a device fault comes from
    // The step (events 0 / 8) and the solve stage (6 / 7) are always timed.  A recorded event is a barrier packet of its own (~6 us of
    // idle device per event: 24 us per step); where the stage is ONE kernel the events ride on that kernel's dispatch instead
    // (hipExtLaunchKernelGGL start / stop events): no packet, no gap.  `attached` = this step's 0 / 6 / 7 / 8 are attached ones.
    bool attached = !debugSync;   // (set per pass below: a graph cannot hold the attached form, it gets recorded events)
    // Timing is opt-in (mi_world_set_stage_timing): even ATTACHED events are not free — the start / stop events riding on the solver's dispatch cost ~11 us of idle
    // device per step (the kernels before / after wait for the signals), the step's two ~1.5 us: 12 us of a 1.0 ms step for numbers nobody asked for.
    const int stepEventsMode = stepEvents || stageEvents ? 2 : 0;   // (0 none; 2 step + solve stage)
    auto mark = [&]() {
        const int id = evi++;
        if (!stageEvents && (stepEventsMode == 0 || (stepEventsMode == 1 && (id == 0 || id == 8)))) return;
        if (attached && (id == 0 || id == 6 || id == 7 || id == 8)) return;
        if (!stageEvents && id != 0 && id != 6 && id != 7 && id != 8) return;   // level 2: only the step and the solve stage
        if (L.hashing && !stageEvents && graphNoEvents) return;
        L.eventRecord(ev[id], st);
        if (debugSync) { hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) std::fprintf(stderr, "[mi_physics] step %llu (%s): stage ending at mark %d: %s\n", (unsigned long long)totalSteps, spec ? "speculative" : "synchronous", evi - 1, hipGetErrorString(e)); }
    };
    auto bound = [](uint32_t last, uint32_t slack) { return last + last / 8u + slack; };
    auto readScalars = [&]() -> int { HIP_TRY(hipMemcpyAsync(&hs, sc, sizeof(StepScalars), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); return MI_OK; };

    // Passes over the enqueue section below (launcher.hpp): a speculative step of a small scene first runs it DRY (signature only); a known
    // signature is replayed from its captured graph, one seen in the previous step as well is captured now, anything else runs plainly.
    static const bool debugSyncG = debugSync;
    const bool graphStep = spec && graphsEnabled && !profileSolve && !xcdFaultTest && !flowFaultTest && !debugSyncG && !launchFallbackSteps &&
                           (graphsForAll || nc <= graphMaxColliders) && readbackSeqDev.p;
    enum { PASS_PLAIN, PASS_DRY, PASS_CAPTURE };
    int pass = graphStep ? PASS_DRY : PASS_PLAIN;
enqueue_section:
    evi = 0; skippedPartition = false;
    L.trace = graphDebug;
    L.begin(pass == PASS_DRY, pass != PASS_PLAIN);
    attached = !debugSync && pass == PASS_PLAIN;
    mark();  // 0
    if (attached && (stepEventsMode == 2 || stageEvents)) hipExtLaunchKernelGGL(k_reset_scalars, dim3(1), dim3(128), 0, st, ev[0], nullptr, 0, sc, shards.p, roundFlagsPtr(), keyCount.p);
    else L.launch(k_reset_scalars, dim3(1), dim3(128), 0, st, sc, shards.p, roundFlagsPtr(), keyCount.p);
    if (shard.enabled && nb) {
        HIP_TRY(shard.activePrev.ensure(std::max(nb, 1u)));
        if (shard.flagsSwapPending) { std::swap(shard.active.p, shard.activePrev.p); std::swap(shard.active.cap, shard.activePrev.cap); shard.flagsSwapPending = false; }   // (not on the synchronous re-run of a step)
        if (!shard.prevValid) { HIP_TRY(L.memsetAsync(shard.activePrev.p, 1, nb, st)); if (!L.dry) shard.prevValid = true; }   // after an upload / an outside write: every body is copied once
        L.launch(k_shard_classify, dim3(divUp(nb, B)), dim3(B), 0, st, nb, shard.sp, bPos.p, bRot.p, bCogInvMass.p, shard.active.p, shards.p, shard.root.p, shard.known.p);
    }
    const bool seamOn = seamMode() && nb;
    if (seamOn) {   // exact seam: which tile border every body is shared across (decides the class of its manifolds)
        HIP_TRY(seamId.ensure(nb));
        if (shard.enabled) L.launch(k_seam_classify_shard, dim3(divUp(nb, B)), dim3(B), 0, st, nb, shard.sp, bPos.p, bRot.p, bCogInvMass.p, shard.root.p, shard.active.p, seamId.p);
        else L.launch(k_seam_classify_tiling, dim3(divUp(nb, B)), dim3(B), 0, st, nb, seamTiling.dBx.p, (uint32_t)seamTiling.bx.size(), seamTiling.dBz.p, (uint32_t)seamTiling.bz.size(), seamTiling.margin,
                      bPos.p, bRot.p, bCogInvMass.p, shard.root.p, seamId.p);
    }
    if (shard.enabled && shard.exact && nb) {   // ... and the bodies whose velocities go to each neighbour after every sweep
        HIP_TRY(shard.sweepCount.ensure(8));
        HIP_TRY(L.memsetAsync(shard.sweepCount.p, 0, 8 * sizeof(uint32_t), st));
        SweepLists lists{}; for (uint32_t k = 0; k < shard.sp.numPeers; ++k) { HIP_TRY(shard.sweepList[k].ensure(shard.capacity)); lists.p[k] = shard.sweepList[k].p; }
        L.launch(k_seam_sweep_list, dim3(divUp(nb, B)), dim3(B), 0, st, nb, shard.sp, shard.active.p, bPos.p, bRot.p, bCogInvMass.p, shard.root.p, lists, shard.capacity, shard.sweepCount.p);
    }
    // with a grid prepared by the previous step the world colliders are computed INSIDE k_bp_prepare (one launch, one pass over the AABB rows less)
    static const bool fuseWorldEnabled = !(std::getenv("MI_FUSE_WORLD") && std::getenv("MI_FUSE_WORLD")[0] == '0');
    const bool fuseWorld = nc && gridValid && fuseWorldEnabled;
    bool prepared = false;
    if (nc) {
        if (fuseWorld) {   // (the cell histogram is all zero here: cleared once at upload, and every scan clears the cells it has read)
            L.launch(k_bp_prepare, dim3(divUp(nc, 256)), dim3(256), 0, st, nc, aabbMin.p, aabbMax.p, grid.p + gridCur, axisPartials.p, shards.p, sc, largeList.p, isLarge.p, blockBounds.p, cellKeys.p, cellRanks.p, cellCount.p,
                     shard.enabled ? shard.activePrev.p : nullptr, shard.enabled ? shard.active.p : nullptr, shard.enabled && shard.desc.rank != 0u ? 0u : 1u,
                     nb, cTypeBody.p, cObject.p, cShape.p, cStaticPos.p, cStaticRot.p, bPos.p, bRot.p, hullAabb.p, wShape.p, aabbMin.p, aabbMax.p, sapAxis, shard.enabled ? shard.axisDev.p : nullptr);
            prepared = true;
        } else
        L.launch(k_world_colliders, dim3(divUp(nc, B)), dim3(B), 0, st, nc, nb, cTypeBody.p, cObject.p, cShape.p, cStaticPos.p, cStaticRot.p, bPos.p, bRot.p, hullAabb.p,
                                                     wShape.p, aabbMin.p, aabbMax.p, sc, sapAxis, shard.enabled ? shard.active.p : nullptr, shard.activePrev.p, shard.enabled ? shard.axisDev.p : nullptr);
        if (heightmap) {   // terrain contacts per collider, their offsets and totals (they join the pair list after the collider-pair narrow phase)
            const HullSet hmHulls{hullVerts.p, hullRanges.p};
            L.launch(k_hm_contacts<false>, dim3(divUp(nc, 4)), dim3(256), 0, st, nc, hmParams, wShape.p, aabbMin.p, aabbMax.p, hmPacked.p, hmSlow.p, nullptr, HmOut{}, hmHulls, hmStash.p);
            L.launch(k_hm_slow<false>, dim3(divUp(nc, 64)), dim3(64), 0, st, nc, hmParams, wShape.p, aabbMin.p, aabbMax.p, hmPacked.p, hmSlow.p, nullptr, HmOut{}, hmHulls);
            HIP_TRY(scanTerrain.run(L, hmPacked.p, hmScan.p, nc, st));
            L.launch(k_hm_totals, dim3(1), dim3(1), 0, st, nc, hmPacked.p, hmScan.p, sc);
        }
    }
    mark();  // 1
    // ---------------------------------------------------------------------------------------------- broad phase
    uint32_t pairBound = 0;   // upper bound of this step's collision pairs that launches / scans are sized for
    GridParams* statsGridNext = nullptr; uint32_t statsCellCap = 0, statsBlocks = 0;   // (speculative steps: pairFinishStats runs beside k_emit_manifolds)
    if (nc) {
        uint32_t nblk = divUp(nc, 256);
        // the cell table (histogram + scan) covers cellCap cells; k_bp_grid_setup enlarges the cells if the grid would need more
        const uint32_t cellCapNext = std::min<uint32_t>(kMaxCells, std::max<uint32_t>(1u << 16, 4u * nc));
        const uint32_t cellCap = gridValid ? (lastCellCap = std::min<uint32_t>(kMaxCells, (lastCellCap > gridNextCells && lastCellCap - gridNextCells <= 8192u) ? lastCellCap : ((gridNextCells + 1u + 4095u) & ~4095u))) : spec ? std::min<uint32_t>(kMaxCells, std::max<uint32_t>(1u << 16, 2u * last.numCells)) : kMaxCells;
        GridParams* gridUse = grid.p + gridCur; GridParams* gridNext = grid.p + (gridCur ^ 1u);
        statsGridNext = gridNext; statsCellCap = cellCapNext; statsBlocks = nblk;
        if (prepared) {}   // k_bp_prepare ran with the world colliders
        else if (gridValid) {
            // the grid prepared at the end of the previous step (k_pair_finish): one fused kernel instead of five launches; the cell histogram
            // is all zero here (cleared once at upload, and every scan clears the cells it has read)
            L.launch(k_bp_prepare, dim3(nblk), dim3(256), 0, st, nc, aabbMin.p, aabbMax.p, gridUse, axisPartials.p, shards.p, sc, largeList.p, isLarge.p, blockBounds.p, cellKeys.p, cellRanks.p, cellCount.p, shard.enabled ? shard.activePrev.p : nullptr, shard.enabled ? shard.active.p : nullptr, shard.enabled && shard.desc.rank != 0u ? 0u : 1u,
                     nb, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr);
        } else {
            L.launch(k_axis_partials, dim3(nblk), dim3(256), 0, st, nc, aabbMin.p, aabbMax.p, axisPartials.p, shards.p, shard.enabled ? shard.active.p : nullptr, shard.enabled && shard.desc.rank != 0u ? 0u : 1u);
            L.launch(k_bp_threshold, dim3(1), dim3(256), 0, st, nc, shards.p, sc);
            L.launch(k_bp_classify, dim3(divUp(nc, B)), dim3(B), 0, st, nc, aabbMin.p, aabbMax.p, sc, largeList.p, isLarge.p, blockBounds.p);
            L.launch(k_bp_grid_setup, dim3(1), dim3(256), 0, st, nc, nblk, cellCap, blockBounds.p, sc, gridUse);
            L.launch(k_bp_cell_ids, dim3(divUp(nc, B)), dim3(B), 0, st, nc, aabbMin.p, aabbMax.p, isLarge.p, gridUse, cellKeys.p, cellRanks.p, cellCount.p);
        }
        HIP_TRY(scanCells.run(L, cellCount.p, cellLower.p, cellCap, st, true));
        L.launch(k_bp_scatter_sorted, dim3(divUp(nc, B)), dim3(B), 0, st, nc, cellKeys.p, cellRanks.p, cellLower.p, aabbMin.p, aabbMax.p, cellKeysS.p, cellValsS.p, sMin.p, sMax.p);
        if (pairKeys.cap == 0) { HIP_TRY(pairKeys.ensure(std::max<size_t>(1u << 16, 8 * (size_t)nc))); }
        if (spec) { HIP_TRY(pairKeys.ensure(bound(last.numPairs, 4096))); }
        if (usesInteractions && interKeys.cap == 0) HIP_TRY(interKeys.ensure(4096));
        if (usesInteractions && spec) HIP_TRY(interKeys.ensure(bound(last.numInterPairs, 1024)));
        for (int attempt = 0; attempt < 3; ++attempt) {
            uint32_t cap = (uint32_t)std::min<size_t>(pairKeys.cap, 0x7FFFFFFFu);
            const InterSink inter{usesInteractions ? interKeys.p : nullptr, (uint32_t)interKeys.cap, &sc->numInterPairs};
            // sized for the small (grid) colliders expected — in a sharded world most colliders are dead and in no list; more than expected: the workgroups loop
            const uint32_t smallBound = spec ? std::min(nc, bound(last.numSmall, 4096)) : nc;
            const uint32_t bpc = (divUp(smallBound, kGridChunks * 256u) + 7u) & ~7u;   // a multiple of 8 (XCD-contiguous block order in k_bp_pairs_grid)
            L.launch(k_bp_pairs_grid, dim3(5u * bpc), dim3(B), 0, st, nc, bpc, cellKeysS.p, cellValsS.p, sMin.p, sMax.p, cellLower.p, gridUse, pairKeys.p, cap, sc, shards.p, inter);
            L.launch(k_bp_pairs_large, dim3(std::min(divUp(smallBound + 1024u, B), 4096u), std::min(16u, std::max(1u, divUp(spec ? last.numLarge + last.numLarge / 4u : 1024u, 64u)))), dim3(B), 0, st, nc, largeList.p, aabbMin.p, aabbMax.p, cellValsS.p, sMin.p, sMax.p, cellLower.p, gridUse, pairKeys.p, cap, sc, shards.p, inter);
            // (a box pile: nearly every pair is of one type and k_pair_finish decides against partitioning — k_pair_partition then does nothing but cost its
            // launch slot: a speculative step whose predecessor was not partitioned leaves it out; if this step wants it after all, k_pair_finish voids the step)
            static const bool skipPartitionEnabled = !(std::getenv("MI_SKIP_PARTITION") && std::getenv("MI_SKIP_PARTITION")[0] == '0');
            skippedPartition = spec && skipPartitionEnabled && havePartitionFlag && !lastPartitioned;
            L.launch(k_pair_finish, dim3(1), dim3(256), 0, st, shards.p, sc, spec ? std::min(cap, bound(last.numPairs, 4096)) : 0xFFFFFFFFu, nc, nblk, attempt == 0 ? axisPartials.p : nullptr, blockBounds.p, attempt == 0 ? gridNext : nullptr, cellCapNext,
                     skippedPartition ? 0u : 1u, spec ? 0u : 1u /* speculative: an extra workgroup of k_emit_manifolds does the statistics, off the critical path */);
            if (spec) { pairBound = std::min(cap, bound(last.numPairs, 4096)); break; }
            int rc = readScalars(); if (rc != MI_OK) return rc;
            pairBound = hs.numPairs + hs.numHmContacts;   // the terrain contacts are appended to the pair list after the narrow phase
            if (pairBound <= cap && hs.numInterPairs <= interKeys.cap) { if (debugOrderPending) { int rco = orientPairsLikeDebugOrder(); if (rco != MI_OK) return rco; } break; }
            if (attempt == 2) return fail(MI_ERR_DEVICE, "pair pass did not settle");
            if (pairBound > cap) HIP_TRY(pairKeys.ensure((size_t)pairBound + pairBound / 4));   // overflow: grow and redo the pair pass
            if (hs.numInterPairs > interKeys.cap) HIP_TRY(interKeys.ensure((size_t)hs.numInterPairs + hs.numInterPairs / 4));
            L.launch(k_reset_pair_counters, dim3(1), dim3(32), 0, st, sc, shards.p);
        }
    }
    if (pass == PASS_PLAIN) finishTimes();   // the previous step's event times, now that this step's first kernels keep the device busy
    mark();  // 2
    // ---------------------------------------------------------------------------------------------- narrow phase
    if (pairBound) {
        HIP_TRY(pairKeysS.ensure(pairKeys.cap));
        if (!skippedPartition) L.launch(k_pair_partition, dim3(divUp(pairBound, 1024)), dim3(256), 0, st, pairKeys.p, pairKeysS.p, sc);
        HIP_TRY(npPacked.ensure(pairBound)); HIP_TRY(npScan.ensure(pairBound)); HIP_TRY(npNormal.ensure(pairBound)); HIP_TRY(npPoints.ensure(4 * (size_t)pairBound));
        HIP_TRY(manPair.ensure(pairBound)); HIP_TRY(manBodies.ensure(pairBound)); HIP_TRY(manInfo.ensure(pairBound));
        HIP_TRY(colWork.ensure(pairBound)); HIP_TRY(color.ensure(pairBound));
        // bodyUsed is all zero here (k_integrate_velocities of the previous step / upload cleared it); k_emit_manifolds seeds it with the kept colours
        HullSet hset{hullVerts.p, hullRanges.p};
        const uint32_t narrowBlocks = divUp(pairBound, B);
        const uint32_t queueRegion = divUp(narrowBlocks, kBoxQueues) * B;   // a queue can hold every pair of the workgroups that feed it
        HIP_TRY(boxQueue.ensure((size_t)kBoxQueues * queueRegion));
        uint32_t histCap = 1024;
        {   // the NEXT step's colour history: sized here, cleared by k_narrow on the side (k_emit_manifolds already enters the manifolds that keep their colour)
            const uint32_t histBound = spec ? std::min(pairBound, bound(last.numManifolds, 1024)) : pairBound;
            const int nt = tabCur ^ 1;
            while (histCap < 2u * histBound) histCap <<= 1;
            HIP_TRY(tab[nt].ensure(histCap)); HIP_TRY(manKept.ensure(pairBound));
            tabMask[nt] = histCap - 1u;
        }
        L.launch(k_narrow, dim3(narrowBlocks), dim3(B), 0, st, pairBound, queueRegion, sc, pairKeys.p, pairKeysS.p, wShape.p, hset, npPacked.p, npNormal.p, npPoints.p, boxQueue.p,
                 reinterpret_cast<ulonglong2*>(tab[tabCur ^ 1].p), histCap);
        L.launch(k_narrow_clip, dim3(kBoxQueues * (queueRegion / B)), dim3(B), 0, st, queueRegion, sc, pairKeys.p, pairKeysS.p, wShape.p, boxQueue.p, npPacked.p, npNormal.p, npPoints.p);
        // (a GJK-only kernel feeding a queue of hits to an EPA kernel was measured: no gain — the GJK half already needs ~250 VGPRs)
        if (usesGjk) {
            HIP_TRY(epaQueue.ensure(pairBound)); HIP_TRY(epaSimplex.ensure((size_t)pairBound * kEpaSimplexRows));
            // few GJK pairs (vehicles on hull tiles, a handful of capsules): one WAVE per pair for GJK as well; many: GJK by lanes, EPA by waves
            static const int gjkWaveMode = std::getenv("MI_GJK_WAVE") ? atoi(std::getenv("MI_GJK_WAVE")) : -1;   // 0 / 1 force a variant (tests, tuning)
            const bool gjkWave = gjkWaveMode >= 0 ? gjkWaveMode != 0 : (spec ? last.gjkSpan <= gjkWaveMaxPairs : false);
            if (gjkWave) L.launch(k_narrow_gjk_wave, dim3(std::min(pairBound, 16384u)), dim3(64), 0, st, sc, pairKeys.p, pairKeysS.p, wShape.p, hset, npPacked.p, npNormal.p, npPoints.p);
            else {
            L.launch(k_narrow_gjk, dim3(divUp(pairBound, 64)), dim3(64), 0, st, sc, pairKeys.p, pairKeysS.p, wShape.p, hset, npPacked.p, npNormal.p, npPoints.p, epaQueue.p, epaSimplex.p, pairBound);
            L.launch(k_narrow_epa, dim3(std::min(pairBound, 8192u)), dim3(64), 0, st, sc, pairKeys.p, pairKeysS.p, wShape.p, hset, epaQueue.p, epaSimplex.p, pairBound, npPacked.p, npNormal.p, npPoints.p);
            }
        }
        if (heightmap) {
            const HmOut hmOut{sc, pairBound, pairKeys.p, pairKeysS.p, npPacked.p, npNormal.p, npPoints.p};
            L.launch(k_hm_contacts<true>, dim3(divUp(nc, 4)), dim3(256), 0, st, nc, hmParams, wShape.p, aabbMin.p, aabbMax.p, hmPacked.p, hmSlow.p, hmScan.p, hmOut, hset, hmStash.p);
            L.launch(k_hm_slow<true>, dim3(divUp(nc, 64)), dim3(64), 0, st, nc, hmParams, wShape.p, aabbMin.p, aabbMax.p, hmPacked.p, hmSlow.p, hmScan.p, hmOut, hset);
            if (hmStash.p && pairBound) L.launch(k_hm_write_stashed, dim3(divUp(pairBound, B)), dim3(B), 0, st, nc, hmParams, wShape.p, aabbMin.p, aabbMax.p, hmPacked.p, hmSlow.p, hmScan.p, hmOut, hset, hmStash.p);   // (sized for all pairs: the terrain contacts are among them)
            L.launch(k_hm_finish, dim3(1), dim3(1), 0, st, sc, pairBound);
        }
        HIP_TRY(scanPairs.run(L, reinterpret_cast<unsigned long long*>(npPacked.p), reinterpret_cast<unsigned long long*>(npScan.p), pairBound, st));
        if (eventsEnabled) HIP_TRY(manIsNew.ensure(pairBound));
        const bool statsInEmit = spec && statsGridNext != nullptr;   // (a synchronous step: k_pair_finish did it)
        L.launch(k_emit_manifolds, dim3(divUp(pairBound, B) + (statsInEmit ? 1u : 0u)), dim3(B), 0, st, nc, nb, pairKeys.p, pairKeysS.p, npPacked.p, npScan.p, cEmit.p,
                                                        manPair.p, manBodies.p, manInfo.p, colWork.p, color.p,
                                                        tabValid ? tab[tabCur].p : nullptr, tabMask[tabCur], bodyUsed.p, eventsEnabled ? manIsNew.p : nullptr, sc,
                                                        heightmap ? make_float2(hmParams.restitution, hmParams.friction) : make_float2(0.f, 0.f),
                                                        tab[tabCur ^ 1].p, tabMask[tabCur ^ 1], manKept.p,
                                                        statsInEmit ? shards.p : nullptr, statsBlocks, axisPartials.p, blockBounds.p, statsGridNext, statsCellCap, seamOn ? seamId.p : nullptr);
    }
    // ---------------------------------------------------------------------------------------------- triggers / force fields
    std::vector<mi_event> triggerEvents;
    uint32_t interPairBound = 0;
    if (usesInteractions && spec) {
        interPairBound = (uint32_t)std::min<size_t>(interKeys.cap, bound(last.numInterPairs, 1024));
        HIP_TRY(interList.ensure(bound(last.numInteractions, 1024))); HIP_TRY(interSorted.ensure(interList.cap));
        int rc = interactionsDevice(interPairBound); if (rc != MI_OK) return rc;
    } else if (usesInteractions) { int rc = interactions(triggerEvents); if (rc != MI_OK) return rc; }
    mark();  // 3
    L.launch(k_integrate_forces, dim3(divUp(nb + 1, B)), dim3(B), 0, st, nb, dt, make_float3(globalForce.x, globalForce.y, globalForce.z), bPos.p, bRot.p, bCogInvMass.p, bInvI.p, bParams.p, bLinVel.p, bAngVel.p, usesInteractions ? bForceStep.p : bForce.p, bTorque.p,
                                                       gPos.p, gInvI.p, gVel.p, gVelL.p, bodyOwner.p, shard.enabled ? shard.active.p : nullptr);
    mark();  // 4
    // ---------------------------------------------------------------------------------------------- schedule
    uint32_t nmBound = 0, conBound = 0;
    if (pairBound) {
        if (spec) { nmBound = std::min(pairBound, bound(last.numManifolds, 1024)); conBound = bound(last.numContacts, 4096); }
        else { int rc = readScalars(); if (rc != MI_OK) return rc; nmBound = hs.numManifolds; conBound = hs.numContacts; }
    }
    uint32_t tilesCap = 0, ctCap = 0, eventCap = 0, xcdListCap = 0;
    bool shardCounted = false;   // sharded world: this rank's manifolds / contacts are counted inside k_manifold_keys when that runs, else by k_shard_count
    // XCD partitioning pays once the pile is big enough to keep eight L2s busy; it needs the persistent kernel (no joints)
    const bool exactSeamStep = shard.enabled && shard.exact;   // every sweep ends in an exchange with the neighbours: one launch per sweep (the generic dataflow path), nothing persistent
    // spatial blocks in LDS (blocks.hpp): speculative steps of a contact-only, unsharded world; sized from the previous block step (planBlocks); any other step takes the classic schedule
    const bool blockPlan = spec && flowSolver && blockSolver && !blkDisabledSteps && joints.count() == 0 && !xcdOnly && !debugOrderPending && !exactSeamStep && !shard.enabled && !seamOn &&
                           !launchFallbackSteps && nmBound && planBlocks(last.numManifolds, nb);
    const bool xcdAble = !blockPlan && flowSolver && persistSolver && persistXcd && !xcdOnly && joints.count() == 0 && (persistWaves & 7u) == 0u && !debugOrderPending && !exactSeamStep;
    // small piles: the 128 waves of ONE XCD run the whole solve, every body hand-over goes through that XCD's L2 (tileOwner(..., single))
    const bool xcdSingle = xcdAble && persistXcdSingle && nmBound && nmBound < xcdMinManifolds && divUp(divUp(nmBound, 64) + kSchedBins + 8, persistWaves / 8u) <= 16u;
    const bool xcdPlan = xcdAble && (nmBound >= xcdMinManifolds || xcdSingle);
    static const uint32_t colorMargin = std::getenv("MI_COLOR_MARGIN") ? (uint32_t)atoi(std::getenv("MI_COLOR_MARGIN")) : 1u;   // extra rounds enqueued beyond the previous step's count
    // converged rounds exit at once, but every enqueued round costs its launch slot (~4.6 us): a scene that replays its steps as graphs wants the same
    // launches step after step (a multiple of 4), a large one exactly what the previous step needed plus the margin
    uint32_t colorBatch = !spec ? 20u : graphStep ? std::min<uint32_t>(96u, (last.colorRounds + std::max(colorMargin, last.colorRounds / 4u) + 3u) & ~3u)
                                                  : std::min<uint32_t>(96u, last.colorRounds + std::max(colorMargin, last.colorRounds / 4u));
    if (nmBound) {
        tilesCap = divUp(nmBound, 64) + kSchedBins + 8; ctCap = divUp(conBound, 64) + 4 * kSchedBins + 8;
        if (blockPlan) { tilesCap = blkCaps.nbe * blkCaps.tiles; ctCap = tilesCap * 4u; }   // every block owns `tiles` tiles of 4 contact-tiles, dense from its first one
        const uint32_t binBlocks = divUp(nmBound, kBinItems);
        HIP_TRY(order.ensure(std::max((size_t)binBlocks * kBinItems, (size_t)tilesCap * 64u))); HIP_TRY(orderTmp.ensure((size_t)binBlocks * kBinItems));
        HIP_TRY(blockHist.ensure((size_t)kColorBins * binBlocks)); HIP_TRY(blockScan.ensure((size_t)kColorBins * binBlocks));
        HIP_TRY(tileInfo.ensure(tilesCap)); HIP_TRY(tileDesc.ensure(tilesCap));
        if (blockPlan) {   // cells -> blocks, the manifolds in cell order, boundary manifolds into the neighbour's list (blocks.hpp)
            const BlockCaps& bc = blkCaps;
            HIP_TRY(blkKeys.ensure(nmBound)); HIP_TRY(blkRanks.ensure(nmBound)); HIP_TRY(blkPerm.ensure(nmBound)); HIP_TRY(blkStart.ensure(bc.nbe + 1u)); HIP_TRY(blkCell.ensure(kBlockCells));
            HIP_TRY(blkExtra.ensure((size_t)bc.nbe * bc.extraCap)); HIP_TRY(blkExtraCount.ensure(bc.nbe));
            const size_t mailWords = ((size_t)nb + 1u) * kMailRanks * 4u;
            if (mail.cap < mailWords || blkLaunches >= 30000u) {   // fresh memory, or the 16-bit launch stamp of the tags is half way round: no record may look current
                HIP_TRY(mail.ensure(mailWords));
                HIP_TRY(L.memsetAsync(mail.p, 0xFF, mail.cap * sizeof(float4), st));
                if (!L.dry) blkLaunches = 0;
            }
            if (!L.dry) ++blkLaunches;
            L.launch(k_block_keys, dim3(divUp(nmBound, kKeyItems)), dim3(256), 0, st, nmBound, sc, grid.p + gridCur, manBodies.p, gPos.p, blkKeys.p, blkRanks.p, keyCount.p, bc.nbe, blkExtraCount.p, &sc->blk);
            L.launch(k_block_place, dim3(divUp(nmBound, kKeyItems)), dim3(256), 0, st, nmBound, sc, blkKeys.p, blkRanks.p, keyCount.p, blkPerm.p, bc.nbe, blkStart.p, blkCell.p, blkExtra.p, bc.extraCap, blkExtraCount.p);
        } else
        if (xcdPlan) {   // slots in spatial order inside every bin + per-XCD tile lists (k_contact_solve_persist<.., true>)
            xcdListCap = xcdSingle ? tilesCap + kSchedBins : divUp(tilesCap, 8) + kSchedBins;
            HIP_TRY(sortKeys[0].ensure(nmBound)); for (int k = 0; k < 2; ++k) HIP_TRY(sortVals[k].ensure(nmBound));
            HIP_TRY(xcdTiles.ensure((size_t)8 * xcdListCap)); HIP_TRY(xcdInfo.ensure((size_t)8 * xcdListCap));
            if (!xcdSingle) {   // (one XCD: nothing to keep apart, the emission order will do)
                L.launch(k_manifold_keys, dim3(divUp(nmBound, kKeyItems)), dim3(256), 0, st, nmBound, sc, grid.p + gridCur, manBodies.p, gPos.p, sortKeys[0].p, sortVals[0].p, keyCount.p,
                         nb, shard.enabled ? shard.active.p : nullptr, manInfo.p, shards.p);
                shardCounted = shard.enabled;
                L.launch(k_manifold_place, dim3(divUp(nmBound, kKeyItems)), dim3(256), 0, st, nmBound, sc, sortKeys[0].p, sortVals[0].p, keyCount.p, sortVals[1].p);
            }
        }
        static const bool xcdNoSort = std::getenv("MI_XCD_NOSORT") != nullptr;   // development: manifold order as emitted
        const uint32_t* perm = xcdPlan && !xcdSingle && !xcdNoSort ? sortVals[1].p : nullptr;
        unsigned long long* top[2] = {bodyTop.p, bodyTop.p + (nb + 1)};
        if (debugOrderPending) { int rc = applyDebugOrder(); if (rc != MI_OK) return rc; }   // (synchronous step: hs holds this step's counts) every manifold -> the sequential colour
        uint32_t round = 0;
        while (true) {
            for (uint32_t r = 0; r < colorBatch; ++r, ++round)
                L.launch(k_color_round, dim3(divUp(nmBound, B)), dim3(B), 0, st, sc, round, colWork.p, color.p, top[round & 1], top[(round + 1) & 1], bodyUsed.p, roundFlagsPtr(), seamOn ? 1u : 0u);
            if (blockPlan) {   // every block sorts its own list into dense tiles: no global bins, no scan (k_block_sched)
                const int ntab = tabCur ^ 1;
                L.launch(k_block_sched, dim3(blkCaps.nbe), dim3(256), 0, st, round - 1, roundFlagsPtr(), blkPerm.p, blkKeys.p, blkStart.p, blkCell.p, blkExtra.p, blkCaps.extraCap, blkExtraCount.p, blkCaps.tiles,
                         color.p, manInfo.p, colWork.p, order.p, tileInfo.p, tileDesc.p, bndMask.p, sc, &sc->blk, nc, manPair.p, pairKeys.p, pairKeysS.p, tab[ntab].p, tabMask[ntab], manKept.p);
                break;   // (speculative steps only)
            }
            // schedule bins -> tiles (valid once the last round left nothing uncoloured: StepScalars::colorPending)
            L.launch(k_bin_hist, dim3(binBlocks), dim3(256), 0, st, sc, binBlocks, perm, color.p, manInfo.p, blockHist.p);
            HIP_TRY(scanBins.run(L, blockHist.p, blockScan.p, kColorBins * binBlocks, st));
            {   // slots, colour history of the new manifolds, bins -> tiles, tile tables: one launch (k_schedule_finish)
                const int ntab = tabCur ^ 1;   // the NEXT step's colour history (sized and cleared in the narrow-phase stage; current only if this step turns out valid)
                L.launch(k_schedule_finish, dim3(binBlocks + divUp(tilesCap, B)), dim3(256), 0, st, round - 1, roundFlagsPtr(), binBlocks, perm, color.p, manInfo.p, blockHist.p, blockScan.p, order.p, sc,
                         nc, manPair.p, pairKeys.p, pairKeysS.p, tab[ntab].p, tabMask[ntab], manKept.p,
                         tilesCap, ctCap, binInfo.p, xcdPlan ? xcdBase.p : nullptr, xcdSingle ? 1u : 0u, tileInfo.p, tileDesc.p, xcdPlan ? xcdTiles.p : nullptr, xcdInfo.p, xcdListCap);
            }
            if (spec) break;
            int rc = readScalars(); if (rc != MI_OK) return rc;
            if (hs.colorPending == 0) break;
            if (round + 8 > kMaxColorRounds) return fail(MI_ERR_DEVICE, "colouring did not converge");
            colorBatch = 8;
        }
        colorRoundsLaunched = round;
        if (seamOn) L.launch(k_seam_stats, dim3(divUp(nmBound, B)), dim3(B), 0, st, sc, colWork.p, color.p, shard.enabled ? shard.active.p : nullptr);
        {   // colour history for the next step, into the OTHER table (it becomes current only if this step turns out valid)
            const int nt = tabCur ^ 1;   // sized and cleared before k_emit_manifolds (narrow phase stage)
            if (eventsEnabled) {   // begins: manifolds not in the previous table; ends: previous pairs not in this step's table
                eventCap = nmBound + (tabValid ? last.numManifolds : 0u) + 1024u;
                HIP_TRY(devEvents.ensure(eventCap));
                L.launch(k_events_begin, dim3(divUp(nmBound, B)), dim3(B), 0, st, nc, eventCap, sc, manIsNew.p, manPair.p, manBodies.p, manInfo.p, pairKeys.p, pairKeysS.p,
                                                               npNormal.p, npPoints.p, gPos.p, gVel.p, devEvents.p);
                if (tabValid) L.launch(k_events_end, dim3(divUp(tabMask[tabCur] + 1u, B)), dim3(B), 0, st, eventCap, sc, tab[tabCur].p, tabMask[tabCur], tab[nt].p, tabMask[nt], devEvents.p);
            }
        }
        if (!spec) {
            mirrorSchedule();
            const BinInfo& ob = bins[kSchedBins - 1];
            if (debugOrderPending) {   // the caller's order (applyDebugOrder left every manifold's rank in debugRank)
                if (ob.count != hs.numManifolds) return fail(MI_ERR_DEVICE, "mi_debug_set_solve_order: schedule did not put every manifold into the sequential bin");
                if (ob.count) {
                    std::vector<uint32_t> ord(ob.count);
                    for (uint32_t m = 0; m < ob.count; ++m) ord[debugRank[m]] = m;
                    HIP_TRY(hipMemcpyAsync(order.p + ob.slotStart, ord.data(), ob.count * sizeof(uint32_t), hipMemcpyHostToDevice, st));
                    HIP_TRY(hipStreamSynchronize(st));
                }
            } else
            if (ob.count > 1) {   // overflow colour: sequential solve in ascending pair-key order
                HIP_TRY(L.memcpyAsync(orderTmp.p + ob.slotStart, order.p + ob.slotStart, ob.count * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
                L.launch(k_sort_overflow, dim3(1), dim3(256), 0, st, ob.slotStart, ob.count, manPair.p, hs.partitioned ? pairKeysS.p : pairKeys.p, orderTmp.p, order.p);
            }
        }
    }
    if (debugOrderPending && !nmBound && !debugOrder.empty()) return fail(MI_ERR_INVALID_ARGUMENT, "mi_debug_set_solve_order: the step found no contact manifold, the list holds " + std::to_string(debugOrder.size()));
    mark();  // 5
    // ---------------------------------------------------------------------------------------------- constraints
    const uint32_t tilesLaunch = spec ? tilesCap : totalTiles;   // sync mode knows the exact tile count (mirrorSchedule)
    const bool useFlow = flowSolver && !launchFallbackSteps && !debugOrderPending && (spec || bins[kSchedBins - 1].count == 0 || !nmBound);   // the overflow colour needs the sequential kernel
    static const bool fuseEnabled = !(std::getenv("MI_FUSE_JOINTS") && std::getenv("MI_FUSE_JOINTS")[0] == '0');
    const bool fused = useFlow && fuseEnabled && joints.allInIslands() && !exactSeamStep;   // joints of all sweeps inside the dataflow launch
    // slots (tiles) one persistent workgroup must hold: exact in a synchronous step, from the previous step's lists (+ slack) in a speculative one
    auto persistSlots = [&](uint32_t tiles, bool xcd, bool speculative) -> uint32_t {
        if (!xcd) return divUp(tiles, xcdOnly ? persistWaves / 8u : persistWaves);
        uint32_t longest = 0;
        if (speculative) longest = (haveXcdEstimate && lastXcdSingle == xcdSingle) ? lastXcdMax + lastXcdMax / 8u + 16u : xcdSingle ? tiles + 16u : divUp(tiles, 8) + 64u;
        else for (uint32_t x = 0; x < 8u; ++x) { uint32_t n = 0; for (uint32_t bn = 0; bn < kSchedBins; ++bn) n += tileOwnerCount(x, divUp(bins[bn].count, 64), bn, xcdSingle ? 1u : 0u); longest = std::max(longest, n); }
        return divUp(std::max(longest, 1u), persistWaves / 8u);
    };
    // the persistent kernel keeps the accumulated impulses in LDS while they fit: k_contact_init then need not write the impulse granules
    const bool blocksRun = blockPlan && useFlow && !fused && tilesLaunch;
    const bool persistPlan = !blocksRun && !fused && useFlow && persistSolver && joints.count() == 0 && tilesLaunch && !exactSeamStep;
    const uint32_t persistMaxSlots = persistPlan ? persistSlots(tilesLaunch, xcdPlan, spec) : 0u;
    const bool privateIslands = fused && privateIslandsEnabled && !shard.enabled && joints.dBodyIsland && nmBound && tilesLaunch;
    const IslandPrivate islandPriv = privateIslands ? joints.islandPrivate() : IslandPrivate{nullptr, nullptr, nullptr, nullptr, nullptr};
    const bool impNeeded = !blocksRun && !(persistPlan && persistImpLds && persistMaxSlots * (4u * 512u + 20u) <= 38u * 1024u);
    if (nmBound) {
        HIP_TRY(slotMeta.ensure((size_t)tilesCap * 64)); HIP_TRY(slotNormal.ensure((size_t)tilesCap * 64)); HIP_TRY(slotMass.ensure((size_t)tilesCap * 64));
        HIP_TRY(rows.ensure((size_t)ctCap * kRows * 64)); HIP_TRY(imp.ensure((size_t)ctCap * 64));
        if (privateIslands) {   // which joint islands are private this step (joints.hpp "PRIVATE islands"): their manifolds go to the island's own workgroup
            HIP_TRY(L.memsetAsync(joints.dIslState, 0, 3 * (size_t)joints.numIslands * sizeof(uint32_t), st));
            L.launch(k_island_classify, dim3(divUp(nmBound, B)), dim3(B), 0, st, sc, colWork.p, color.p, islandPriv);
        }
        if (tilesLaunch)
            L.launch(k_contact_init, dim3(xcdPlan ? 8u * xcdListCap : blocksRun ? 8u * divUp(blkCaps.nbe, 8u) * blkCaps.tiles : tilesLaunch), dim3(64), 0, st, sc, nb, dt, xcdPlan ? xcdInfo.p : tileInfo.p, order.p, manPair.p, manBodies.p, manInfo.p, npNormal.p, npPoints.p,
                                                      gPos.p, gInvI.p, xcdPlan ? gVelL.p : gVel.p /* same content here; the cached copy */, color.p, bodyUsed.p, fused ? joints.dBodyJ : nullptr, rows.p, impNeeded ? imp.p : nullptr, slotMeta.p, slotNormal.p, slotMass.p,
                                                      xcdPlan ? reinterpret_cast<uint8_t*>(bodyOwner.p) : nullptr, blocksRun ? blkCaps.nbe : xcdListCap, xcdPlan ? 8u * xcdListCap : tilesCap, islandPriv,
                                                      blocksRun ? bndMask.p : nullptr, &sc->blk);
    }
    int rc = joints.initialize(*this, dt, st);   // (through L)
    if (rc != MI_OK) return rc;
    mark();  // 6
    bool solveAttached = false;
    const bool willPersist = blocksRun || (!fused && persistPlan && persistMaxSlots * 20u <= 38u * 1024u);
    if (attached && !willPersist) (void)hipEventRecord(ev[6], st);   // another solver path (several launches): classic recorded events
    const uint32_t iters = settings.num_rigid_solver_iterations;
    usedFlow = useFlow; usedPersist = false; usedFused = fused; usedXcd = false; usedXcdSingle = false; usedBlocks = false;
    uint64_t mainContacts = 0;
    if (fused) {
        // contacts and joint islands of every sweep in one launch (k_solve_flow_islands)
        const uint64_t per = (uint64_t)joints.numIslands + tilesLaunch;
        const uint32_t perLaunch = per * iters < 0x7FFFFFFFull ? iters : 1u;
        solveLaunches = (iters + perLaunch - 1) / perLaunch;
        BodyView bv{gPos.p, gInvI.p, gVel.p, bRot.p, bCogInvMass.p, shard.enabled ? shard.active.p : nullptr};
        const IslandUpd iu{joints.distance.dUpd, joints.ball.dUpd, joints.fixed.dUpd, joints.hinge.dUpd, joints.cone.dUpd, joints.slider.dUpd};
        const IslandAcc ia{joints.hinge.dAcc, joints.cone.dAcc, joints.slider.dAcc};
        for (uint32_t it = 0; it < iters; it += perLaunch) {
            if (profileSolve) {
                size_t e = 2 * (size_t)profLaunches;
                while (profEvents.size() < e + 2) { hipEvent_t ev_; HIP_TRY(hipEventCreate(&ev_)); profEvents.push_back(ev_); }
                (void)hipEventRecord(profEvents[e], st);
            }
            L.launch(k_solve_flow_islands, dim3((uint32_t)(per * perLaunch)), dim3(64), flowLds, st, it, perLaunch, joints.numIslands, joints.dIslands, joints.dSteps, joints.dIslandBodies, iu, ia, bv, bodyUsed.p,
                                                                                   tileDesc.p, slotMeta.p, slotNormal.p, slotMass.p, rows.p, imp.p, sc, islandPriv, iters);
            if (profileSolve) { (void)hipEventRecord(profEvents[2 * (size_t)profLaunches + 1], st); ++profLaunches; }
        }
    } else if (blocksRun) {
        // one workgroup per spatial block: home bodies, slot constants and impulses in LDS, boundary manifolds solved on both sides (k_contact_solve_blocks)
        const BlockCaps& bc = blkCaps;
        solveLaunches = 1; usedBlocks = true;
        if (profileSolve) {
            size_t e = 2 * (size_t)profLaunches;
            while (profEvents.size() < e + 2) { hipEvent_t ev_; HIP_TRY(hipEventCreate(&ev_)); profEvents.push_back(ev_); }
            (void)hipEventRecord(profEvents[e], st);
        }
        const uint32_t fault = blockFaultTest && !blockFaultFired ? 1u : 0u;   // tests: one block gives up once
        if (fault && !L.dry) blockFaultFired = true;
        const uint32_t maxSlots = divUp(bc.tiles, blkWaves);
        auto* blockKernel = k_contact_solve_blocks;
        static DBuf<unsigned long long> trapBuf;   // development (MI_BLOCK_MODE & 0x100): records of lanes with wild values, printed when the first ones appear
        static bool trapPrinted = false;
        if (std::getenv("MI_BLOCK_MODE") && (strtoul(std::getenv("MI_BLOCK_MODE"), nullptr, 0) & 0x100u)) {
            if (!trapBuf.p) { HIP_TRY(trapBuf.ensure(8 + 256 * 8)); HIP_TRY(hipMemsetAsync(trapBuf.p, 0, trapBuf.cap * 8, st)); }
            else if (!trapPrinted) {
                std::vector<unsigned long long> h(8 + 256 * 8); HIP_TRY(hipMemcpy(h.data(), trapBuf.p, h.size() * 8, hipMemcpyDeviceToHost));
                if (h[0]) { trapPrinted = true; std::fprintf(stderr, "[mi_physics] trap: %llu wild lanes by step %llu\n", h[0], (unsigned long long)totalSteps);
                    for (unsigned long long k = 0; k < std::min<unsigned long long>(h[0], 40ull); ++k) { const uint32_t* r = reinterpret_cast<const uint32_t*>(&h[8 + k * 8]);
                        std::fprintf(stderr, "  block %u wave %u lane %u pass %u it %u slot %u | bnd %u ghostA %u ghostB %u updA %u updB %u | wild inA %u inB %u out %u rows %u | mc %u cnt %u lo %u hi %u | meta %u %u %08x | a0.x %g b0.x %g g0.x %g g0.w %08x want %08x | row %g nf %g mass %g | tagA %u/%u tagB %u/%u\n",
                                     r[0] & 0xFFFF, (r[0] >> 16) & 0xFF, r[0] >> 24, r[1] & 0xFF, (r[1] >> 8) & 0xFF, r[1] >> 16, r[2] & 1, (r[2] >> 1) & 1, (r[2] >> 2) & 1, (r[2] >> 3) & 1, (r[2] >> 4) & 1, (r[2] >> 5) & 1, (r[2] >> 6) & 1, (r[2] >> 7) & 1, (r[2] >> 8) & 1,
                                     (r[2] >> 12) & 15, (r[2] >> 16) & 15, (r[2] >> 20) & 63, (r[2] >> 26) & 127, r[3], r[4], r[5], *(const float*)&r[6], *(const float*)&r[7], *(const float*)&r[8], r[9], r[10], *(const float*)&r[11], *(const float*)&r[12], *(const float*)&r[13],
                                     r[14] & 0xFFFF, r[14] >> 16, r[15] & 0xFFFF, r[15] >> 16); } }
            }
        }
        static const uint32_t blockMode = std::getenv("MI_BLOCK_MODE") ? (uint32_t)strtoul(std::getenv("MI_BLOCK_MODE"), nullptr, 0) : 0u;   // development: 0x20 rows fetched at every pass (no prefetch), 0x40 full waits
        static const std::vector<uint32_t> blockDbgList = [] { std::vector<uint32_t> v; if (const char* e = std::getenv("MI_BLOCK_DBG")) { const char* p = e; while (*p) { char* q; v.push_back((uint32_t)strtoul(p, &q, 0)); p = (*q == ',') ? q + 1 : q; if (q == p && *q) break; } } return v; }();
        hipEvent_t e6 = attached && (stepEventsMode || stageEvents) ? ev[6] : nullptr, e7 = attached && (stepEventsMode || stageEvents) ? ev[7] : nullptr;
        solveAttached = attached;
#define MI_BLOCK_ARGS iters, bc.tiles, bc.hashSize, bc.bodyCap, maxSlots, bc.maxPasses, bc.impCap, tileInfo.p, slotMeta.p, slotNormal.p, slotMass.p, rows.p, gVel.p, gVelL.p, mail.p, sc, &sc->blk, fault, blockMode, trapBuf.p
        if (attached) hipExtLaunchKernelGGL(blockKernel, dim3(bc.nbe), dim3(blkWaves * 64), (uint32_t)bc.lds, st, e6, e7, 0, MI_BLOCK_ARGS);
        else L.launch(blockKernel, dim3(bc.nbe), dim3(blkWaves * 64), bc.lds, st, MI_BLOCK_ARGS);
#undef MI_BLOCK_ARGS
        static const uint64_t blockDbgAfter = std::getenv("MI_BLOCK_DBG_AFTER") ? strtoull(std::getenv("MI_BLOCK_DBG_AFTER"), nullptr, 0) : 0ull;
        if (!blockDbgList.empty() && pass == PASS_PLAIN && totalSteps > blockDbgAfter) {   // development: the same launch once more with knock-outs, into scratch outputs (the step's results stay right), timed by its own event pair;
                                                              // MI_BLOCK_DBG=a,b,c: 20 steps with each value in turn
            static DBuf<unsigned long long> dbgTimes; HIP_TRY(dbgTimes.ensure((size_t)bc.nbe * blkWaves * 8u));
            static DBuf<float4> scratchVel, scratchMail; static hipEvent_t dbgEv[2] = {nullptr, nullptr}; static double dbgSum = 0; static uint32_t dbgN = 0;
            HIP_TRY(scratchVel.ensure(2 * ((size_t)nb + 1))); scratchMail.flags = hipDeviceMallocUncached;
            if (scratchMail.cap < mail.cap) { HIP_TRY(scratchMail.ensure(mail.cap)); HIP_TRY(hipMemsetAsync(scratchMail.p, 0xFF, scratchMail.cap * sizeof(float4), st)); }
            if (!dbgEv[0]) { HIP_TRY(hipEventCreate(&dbgEv[0])); HIP_TRY(hipEventCreate(&dbgEv[1])); }
            else { HIP_TRY(hipEventSynchronize(dbgEv[1])); float ms = 0; if (hipEventElapsedTime(&ms, dbgEv[0], dbgEv[1]) == hipSuccess) { dbgSum += ms; if (++dbgN % 20u == 0u) {
                std::fprintf(stderr, "[mi_physics] block knock-outs %u: extra launch %.1f us (mean of 20; step %llu, %u contacts)\n", blockDbgList[((dbgN - 1u) / 20u) % blockDbgList.size()], dbgSum / 20.0 * 1e3, (unsigned long long)totalSteps, last.numContacts); dbgSum = 0; } } }
            const uint32_t dbgNow = blockDbgList[(dbgN / 20u) % blockDbgList.size()];
            hipExtLaunchKernelGGL(blockKernel, dim3(bc.nbe), dim3(blkWaves * 64), (uint32_t)bc.lds, st, dbgEv[0], dbgEv[1], 0, iters, bc.tiles, bc.hashSize, bc.bodyCap, maxSlots, bc.maxPasses, bc.impCap,
                                  tileInfo.p, slotMeta.p, slotNormal.p, slotMass.p, rows.p, gVel.p, scratchVel.p, scratchMail.p, sc, &sc->blk, 0u, dbgNow | 0x200u, dbgTimes.p);
            if (dbgN % 20u == 19u) {   // phases of this launch, over all waves: mean and max of (hash, records, lists, impulses zeroed, main loop, barrier, epilogue) in us
                HIP_TRY(hipStreamSynchronize(st));
                std::vector<unsigned long long> h((size_t)bc.nbe * blkWaves * 8u);
                HIP_TRY(hipMemcpy(h.data(), dbgTimes.p, h.size() * 8u, hipMemcpyDeviceToHost));
                double mean[7] = {0}, mx[7] = {0}; unsigned long long t0 = ~0ull, t1 = 0;
                for (size_t w_ = 0; w_ < (size_t)bc.nbe * blkWaves; ++w_) { const unsigned long long* t = &h[w_ * 8u]; t0 = std::min(t0, t[0]); t1 = std::max(t1, t[7]);
                    for (int k = 0; k < 7; ++k) { const double d = (double)(t[k + 1] - t[k]) * 0.01; mean[k] += d / ((double)bc.nbe * blkWaves); mx[k] = std::max(mx[k], d); } }
                std::fprintf(stderr, "[mi_physics] block phases (knock-outs %u): span %.1f us; mean / max us: hash %.1f / %.1f, records %.1f / %.1f, lists %.1f / %.1f, zero %.1f / %.1f, main %.1f / %.1f, barrier %.1f / %.1f, epilogue %.1f / %.1f\n",
                             dbgNow, (double)(t1 - t0) * 0.01, mean[0], mx[0], mean[1], mx[1], mean[2], mx[2], mean[3], mx[3], mean[4], mx[4], mean[5], mx[5], mean[6], mx[6]);
            }
        }
        if (profileSolve) { (void)hipEventRecord(profEvents[2 * (size_t)profLaunches + 1], st); ++profLaunches; }
    } else if (persistPlan && persistMaxSlots * 20u <= 38u * 1024u) {
        // persistent waves: one workgroup per SIMD owns its tiles through all sweeps, impulses (and, while they fit, the constant slot data) in LDS (k_contact_solve_persist)
        const uint32_t maxSlots = persistMaxSlots;
        const bool metaLds = persistMetaLds && maxSlots * (64u * 40u + 4u * 512u + 20u) <= 38u * 1024u;
        usedXcd = xcdPlan; usedXcdSingle = xcdPlan && xcdSingle;
        const uint32_t xcdFault = xcdFaultTest && usedXcd && !xcdFaultFired ? 1u : 0u;   // tests: one workgroup reports a placement mismatch once
        if (xcdFault) xcdFaultFired = true;
        solveLaunches = 1; usedPersist = true;
        if (profileSolve) {
            size_t e = 2 * (size_t)profLaunches;
            while (profEvents.size() < e + 2) { hipEvent_t ev_; HIP_TRY(hipEventCreate(&ev_)); profEvents.push_back(ev_); }
            (void)hipEventRecord(profEvents[e], st);
        }
        const bool impLds = persistImpLds && maxSlots * (4u * 512u + 20u) <= 38u * 1024u;   // beyond that the impulses travel as granules in `imp` (no size limit)
        const uint32_t ldsMeta = maxSlots * (64u * 40u + 4u * 512u + 20u) + 16u, ldsImp = maxSlots * (4u * 512u + 20u) + 16u, ldsDesc = maxSlots * 20u + 16u;
#define MI_PERSIST_ARGS iters, maxSlots, tileDesc.p, slotMeta.p, slotNormal.p, slotMass.p, rows.p, gVel.p, sc, xcdOnly, xcdTiles.p, xcdListCap, bodyOwner.p, gVelL.p, slotMeta.p, imp.p, xcdFault
        // the solve stage IS this launch: its timing events ride on the dispatch (when this path is not taken they are recorded below)
        hipEvent_t e6 = attached && (stepEventsMode || stageEvents) ? ev[6] : nullptr, e7 = attached && (stepEventsMode || stageEvents) ? ev[7] : nullptr;
        solveAttached = attached;
#define MI_PERSIST_LAUNCH(A, B_, C_, LDS) do { if (attached) hipExtLaunchKernelGGL((k_contact_solve_persist<A, B_, C_>), dim3(persistWaves), dim3(64), LDS, st, e6, e7, 0, MI_PERSIST_ARGS); \
                                              else L.launch(k_contact_solve_persist<A, B_, C_>, dim3(persistWaves), dim3(64), LDS, st, MI_PERSIST_ARGS); } while (0)
        if (usedXcd) {
            if (metaLds && impLds) MI_PERSIST_LAUNCH(true, true, true, ldsMeta);
            else if (impLds) MI_PERSIST_LAUNCH(false, true, true, ldsImp);
            else MI_PERSIST_LAUNCH(false, true, false, ldsDesc);
        } else {
            if (metaLds && impLds) MI_PERSIST_LAUNCH(true, false, true, ldsMeta);
            else if (impLds) MI_PERSIST_LAUNCH(false, false, true, ldsImp);
            else MI_PERSIST_LAUNCH(false, false, false, ldsDesc);
        }
#undef MI_PERSIST_LAUNCH
#undef MI_PERSIST_ARGS
        if (profileSolve) { (void)hipEventRecord(profEvents[2 * (size_t)profLaunches + 1], st); ++profLaunches; }
    } else if (useFlow) {
        // no joints between the sweeps -> all sweeps in one launch; otherwise one launch per sweep (joints run in between)
        const uint32_t perLaunch = joints.count() == 0 && !exactSeamStep && (uint64_t)std::max(tilesLaunch, 1u) * iters < 0x7FFFFFFFull ? iters : 1u;
        solveLaunches = tilesLaunch ? (iters + perLaunch - 1) / perLaunch : 0;
        for (uint32_t it = 0; it < iters; it += perLaunch) {
            if (perLaunch == 1) joints.solveIteration(*this, st);   // distance, ball, fixed, hinge, cone-twist, slider (constraints.cpp:3764-3769)
            if (!tilesLaunch) { if (exactSeamStep) { int rcx = shardSweepExchange(it); if (rcx != MI_OK) return rcx; } continue; }
            if (profileSolve) {
                size_t e = 2 * (size_t)profLaunches;
                while (profEvents.size() < e + 2) { hipEvent_t ev_; HIP_TRY(hipEventCreate(&ev_)); profEvents.push_back(ev_); }
                (void)hipEventRecord(profEvents[e], st);
            }
            L.launch(k_contact_solve_flow, dim3(tilesLaunch * perLaunch), dim3(64), flowLds, st, it, perLaunch, tileDesc.p, slotMeta.p, slotNormal.p, slotMass.p, rows.p, imp.p, gVel.p, sc);
            if (profileSolve) { (void)hipEventRecord(profEvents[2 * (size_t)profLaunches + 1], st); ++profLaunches; }
            if (exactSeamStep) { int rcx = shardSweepExchange(it); if (rcx != MI_OK) return rcx; }   // exact seam: the owners' velocities of the shared bodies replace the ghost copies
        }
    } else {
        // one launch per colour per sweep (MI_SOLVER=launch, or an overflow colour is present); synchronous mode only
        auto colorCount = [&](uint32_t c) { return bins[4 * c].count + bins[4 * c + 1].count + bins[4 * c + 2].count + bins[4 * c + 3].count; };
        // colours [tailStart, tailEnd) are small (<= 512 manifolds each, a suffix of the used colours): one launch for all of them
        if (!nmBound) { numColorsUsed = 0; for (BinInfo& b : bins) b.count = 0; }   // no manifolds this step: the mirrored schedule is the previous step's
        uint32_t tailEnd = std::min(numColorsUsed, kOverflowColor), tailStart = tailEnd;
        while (tailStart > 0 && colorCount(tailStart - 1) <= 512u) --tailStart;
        if (tailEnd - tailStart < 2) tailStart = tailEnd;
        std::vector<ColorLaunch> launches(tailStart);
        for (uint32_t c = 0; c < tailStart; ++c) {
            ColorLaunch& cl = launches[c];
            uint32_t acc = 0;
            for (uint32_t k = 0; k < 4; ++k) { cl.tileStart[k] = bins[4 * c + k].tileStart; cl.ctStart[k] = bins[4 * c + k].ctStart; mainContacts += (uint64_t)bins[4 * c + k].count * (k + 1); }
            for (uint32_t i = 0; i < 4; ++i) { acc += divUp(bins[4 * c + (3 - i)].count, 64); cl.blockEnd[i] = acc; }
            cl.numBlocks = acc; cl.swizzle = xcdSwizzle ? 1u : 0u;
        }
        solveLaunches = iters * (tailStart + (tailStart < tailEnd ? 1u : 0u));
        for (uint32_t it = 0; it < iters; ++it) {
            if (debugOrderPending) { int rcj = joints.solveIterationReference(*this, st); if (rcj != MI_OK) return rcj; }
            else joints.solveIteration(*this, st);
            for (uint32_t c = 0; c < tailStart; ++c) {
                const ColorLaunch& cl = launches[c];
                if (!cl.numBlocks) continue;
                uint32_t grid_ = cl.swizzle ? divUp(cl.numBlocks, 8) * 8 : cl.numBlocks;
                if (profileSolve) {
                    size_t e = 2 * (size_t)profLaunches;
                    while (profEvents.size() < e + 2) { hipEvent_t ev_; HIP_TRY(hipEventCreate(&ev_)); profEvents.push_back(ev_); }
                    (void)hipEventRecord(profEvents[e], st);
                    L.launch(k_contact_solve, dim3(grid_), dim3(64), 0, st, cl, slotMeta.p, slotNormal.p, slotMass.p, rows.p, imp.p, gVel.p);
                    (void)hipEventRecord(profEvents[e + 1], st);
                    ++profLaunches;
                } else {
                    L.launch(k_contact_solve, dim3(grid_), dim3(64), 0, st, cl, slotMeta.p, slotNormal.p, slotMass.p, rows.p, imp.p, gVel.p);
                }
            }
            if (tailStart < tailEnd) L.launch(k_contact_solve_tail, dim3(1), dim3(256), 0, st, binInfo.p, tailStart, tailEnd, slotMeta.p, slotNormal.p, slotMass.p, rows.p, imp.p, gVel.p);
            if (bins[kSchedBins - 1].count) L.launch(k_contact_solve_serial, dim3(1), dim3(64), 0, st, bins[kSchedBins - 1], slotMeta.p, slotNormal.p, slotMass.p, rows.p, imp.p, gVel.p);
            if (exactSeamStep) { int rcx = shardSweepExchange(it); if (rcx != MI_OK) return rcx; }
        }
    }
    if (shard.enabled && pairBound && !shardCounted) L.launch(k_shard_count, dim3(divUp(nmBound ? nmBound : 1u, B)), dim3(B), 0, st, nb, manBodies.p, manInfo.p, bCogInvMass.p, shard.active.p, sc, shards.p);
    mark();  // 7
    if (attached && !solveAttached) (void)hipEventRecord(ev[7], st);
    if (attached && (stepEventsMode == 2 || stageEvents)) hipExtLaunchKernelGGL(k_integrate_velocities, dim3(divUp(nb + 1, B)), dim3(B), 0, st, nullptr, ev[8], 0, nb, dt, gPos.p, usedBlocks ? gVelL.p : gVel.p, bCogInvMass.p, bRot.p, bPosN.p, bRotN.p, bLinVelN.p, bAngVelN.p, bForceN.p, bTorqueN.p,
                                      gVelL.p, usedXcd ? bodyOwner.p : nullptr, bodyUsed.p, bodyTop.p,
                                      shard.enabled ? shard.active.p : nullptr, bPos.p, bLinVel.p, bAngVel.p, bForce.p, bTorque.p, shard.activePrev.p, shards.p, sc, bndMask.p);
    else L.launch(k_integrate_velocities, dim3(divUp(nb + 1, B)), dim3(B), 0, st, nb, dt, gPos.p, usedBlocks ? gVelL.p : gVel.p, bCogInvMass.p, bRot.p, bPosN.p, bRotN.p, bLinVelN.p, bAngVelN.p, bForceN.p, bTorqueN.p,
                                                      gVelL.p, usedXcd ? bodyOwner.p : nullptr, bodyUsed.p, bodyTop.p,
                                                      shard.enabled ? shard.active.p : nullptr, bPos.p, bLinVel.p, bAngVel.p, bForce.p, bTorque.p, shard.activePrev.p, shards.p, sc, bndMask.p);
    mark();  // 8
    // ---------------------------------------------------------------------------------------------- end of step: the one read-back
    {
        const uint32_t words = (uint32_t)(offsetof(Readback, seq) / 4u);
        if (spinReadback) L.launch(k_publish_readback, dim3(1), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(sc), words, reinterpret_cast<uint32_t*>(hsPinned), words, readbackSeqDev.p);
        else HIP_TRY(L.memcpyAsync(hsPinned, sc, offsetof(Readback, seq), hipMemcpyDeviceToHost, st));   // scalars + round flags are contiguous
    }
    if (pass == PASS_DRY) {
        const uint64_t sig = L.h ^ ((uint64_t)L.ops << 48);
        StepGraph* hit = nullptr;
        for (StepGraph& g : stepGraphs) if (g.sig == sig && g.exec) { hit = &g; break; }
        if (graphDebug) {
            if (!hit && !graphPrevOps2.empty()) {   // compare with the step before the previous one (same buffer parity)
                size_t k = 0; while (k < graphPrevOps2.size() && k < L.opHashes.size() && graphPrevOps2[k] == L.opHashes[k]) ++k;
                std::fprintf(stderr, "[mi_physics] step %llu: graph signature differs from that of two steps ago at operation %zu of %zu (then %zu)\n", (unsigned long long)totalSteps, k, L.opHashes.size(), graphPrevOps2.size());
            }
            graphPrevOps2 = graphPrevOps; graphPrevOps = L.opHashes;
        }
        if (hit) {
            hit->lastUse = ++graphUseClock; ++graphHits;
            if (hipGraphLaunch(hit->exec, st) != hipSuccess) { (void)hipGetLastError(); graphsEnabled = false; dropStepGraphs(); pass = PASS_PLAIN; goto enqueue_section; }
        } else if (!graphNoCapture && (sig == graphLastSig || sig == graphPrevSig)) {      // seen within the last two steps as well: capture it
            if (stepGraphs.size() >= kMaxStepGraphs) { HIP_TRY(hipStreamSynchronize(st)); dropStepGraphs(); }   // a long-lived scene keeps changing shape: start over (all at once, with the stream idle)
            graphPrevSig = graphLastSig; graphLastSig = sig;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); graphsEnabled = false; pass = PASS_PLAIN; }
            else pass = PASS_CAPTURE;
            goto enqueue_section;
        } else { graphPrevSig = graphLastSig; graphLastSig = sig; ++graphPlain; pass = PASS_PLAIN; goto enqueue_section; }
    } else if (pass == PASS_CAPTURE) {
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        bool ok = hipStreamEndCapture(st, &graph) == hipSuccess && graph && L.firstError == hipSuccess;
        if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (ok) ok = hipGraphLaunch(exec, st) == hipSuccess;
        if (!ok) {   // this runtime cannot hold the step in a graph: plain launches from now on (nothing has been enqueued yet)
            (void)hipGetLastError();
            if (exec) (void)hipGraphExecDestroy(exec);
            graphsEnabled = false; dropStepGraphs(); pass = PASS_PLAIN; goto enqueue_section;
        }
        const uint64_t sig = graphLastSig;
        stepGraphs.push_back(StepGraph{sig, exec, ++graphUseClock}); ++graphCaptures;
    }
    if (poseArm.armed) {   // the poses of the state this step is producing, enqueued behind it (a step that turns out void produces them again)
        ++pose.produced_ahead; poseArm.done = false;
        int rcp = posesProduce(poseArm.t, true); if (rcp != MI_OK) return rcp;
        poseArm.done = true;
    }
    if (spinReadback) {
        const uint32_t seq = readbackSeq + 1u ? readbackSeq + 1u : 1u;   // never 0; the device counts the same way (k_publish_readback)
        readbackSeq = seq;
        volatile uint32_t* flag = &hsPinned->seq;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                HIP_TRY(hipStreamSynchronize(st));   // a long step, a device fault or a lost store: let the runtime report it
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return fail(MI_ERR_DEVICE, "end-of-step read-back did not arrive");
                break;
            }
        }
    } else {
        HIP_TRY(hipStreamSynchronize(st));
    }
    hs = hsPinned->sc;
    const uint32_t* flagsHost = hsPinned->flags;
    HIP_TRY(hipGetLastError());
    if (blockPlan) {   // the blocks wrote COUNTS per (colour, contacts) bin: the prefix the host mirrors; and what the next block step is sized from
        uint32_t run = 0;
        for (uint32_t b = 0; b < kColorBins; ++b) { const uint32_t c = hs.binStart[b]; hs.binStart[b] = run; run += c; }
        hs.binStart[kColorBins] = run;
        lastBlk = hs.blk; haveBlkEstimate = true; ++blkSteps;
        static const bool blockDebug = std::getenv("MI_BLOCK_DEBUG") != nullptr;   // development: one line per block step
        if (blockDebug) std::fprintf(stderr, "[mi_physics] step %llu blocks %u x %u tiles (extra %u, bodies %u, hash %u, passes %u, impulses %u, lds %zu): needed entries %u extras %u bodies %u passes %u impulses %u; boundary entries %u; manifolds %u; overflow %u solveError %u specOverflow %u\n",
                                     (unsigned long long)totalSteps, blkCaps.nbe, blkCaps.tiles, blkCaps.extraCap, blkCaps.bodyCap, blkCaps.hashSize, blkCaps.maxPasses, blkCaps.impCap, blkCaps.lds,
                                     hs.blk.need, hs.blk.needExtra, hs.blk.needBodies, hs.blk.needPasses, hs.blk.needImp, hs.blk.ghostLanes, hs.numManifolds, hs.blk.overflow, hs.solveError, hs.specOverflow);
        const bool failed = hs.blk.overflow != 0u || (usedBlocks && hs.solveError != 0u);
        blkFailHistory = (blkFailHistory << 1) | (failed ? 1u : 0u);
        blkLastFailed = failed;
        if (failed) {
            if (usedBlocks && hs.solveError == 1u) blkDisabledSteps = 256u;          // a wait ran out of budget (shared device, or the test injection): the classic path for a while
            else if (__builtin_popcount(blkFailHistory & 0xFFFFu) >= 8) { blkDisabledSteps = 128u; blkFailHistory = 0u; }   // capacities that do not settle (a pile landing outgrows them step after step: the sizes follow with x 1.5 per failure)
            return STEP_RETRY;   // nothing persistent has been written: the synchronous re-run takes the classic schedule
        }
    }
    if (spec) {
        const uint32_t ovfCount = hs.binStart[kColorBins] - hs.binStart[kSchedBins - 1];
        const bool valid = hs.numPairs <= pairBound && hs.numManifolds <= nmBound && hs.specOverflow == 0 && hs.colorPending == 0 && ovfCount == 0 &&
                           (!usesInteractions || (hs.numInterPairs <= interPairBound && hs.numInteractions <= interList.cap && hs.numInteractions <= 32768u));
        if (!valid) return STEP_RETRY;   // nothing persistent was modified: run the same step synchronously
        mirrorSchedule();
    }
#ifdef MI_DBG_TIMELINE
    if (usedPersist && std::getenv("MI_DBG_TIMELINE_OUT")) {   // development: per-visit wall-clock stamps of the persistent solver, dumped after step MI_DBG_TIMELINE_STEP
        static unsigned long long* dbgBuf = nullptr;
        const size_t words = (size_t)persistWaves * 256 * 8;
        if (dbgBuf && totalSteps == (unsigned long long)atoll(std::getenv("MI_DBG_TIMELINE_STEP") ? std::getenv("MI_DBG_TIMELINE_STEP") : "3")) {
            std::vector<unsigned long long> h(words);
            HIP_TRY(hipMemcpy(h.data(), dbgBuf, words * 8, hipMemcpyDeviceToHost));
            FILE* f = fopen(std::getenv("MI_DBG_TIMELINE_OUT"), "wb"); if (f) { fwrite(h.data(), 8, words, f); fclose(f); }
        }
        if (!dbgBuf) {
            HIP_TRY(hipMalloc(&dbgBuf, words * 8)); HIP_TRY(hipMemset(dbgBuf, 0, words * 8));
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dbgTimeline), &dbgBuf, sizeof(dbgBuf)));
        }
    }
#endif
    if (usedXcd && !hs.solveError) {
        // partitioning is only trusted when the eight residue classes of blockIdx really sat on eight DIFFERENT XCDs (a device
        // that exposes fewer XCDs, or hides the id, runs unpartitioned)
        bool distinct = true;
        for (int a = 0; a < 8 && distinct; ++a) for (int b = a + 1; b < 8; ++b) if (hs.xccOf[a] == hs.xccOf[b]) { distinct = false; break; }
        if (!distinct) { persistXcd = false; return STEP_RETRY; }
    }
    if (hs.solveError && usedPersist) {
        // nothing persistent has been written yet, the step is simply run again:
        //   2 with XCD lists: a list outgrew the speculative LDS sizing -> the synchronous run sizes it exactly;
        //   3: blockIdx % 8 does not identify the XCD on this device -> no XCD partitioning from now on;
        //   otherwise: the persistent kernel needs all its workgroups resident at once and the device did not grant that (or
        //   a wait ran out of budget) -> the dispatch-ordered dataflow kernel for good
        if (hs.solveError == 2u && usedXcd && spec) { haveXcdEstimate = false; return STEP_RETRY; }
        if (usedXcd) persistXcd = false; else persistSolver = false;
        return STEP_RETRY;
    }
    if (flowFaultTest && !flowFaultFired && useFlow && !usedPersist && !hs.solveError) { flowFaultFired = true; hs.solveError = 1u; }   // test injection
    if (hs.solveError && useFlow) {
        // the dispatch-ordered kernels rely on workgroups being started in index order on a device that is not shared; when a wait
        // runs out of budget nothing persistent has been written: run the step again with one launch per colour, and stay there a while
        launchFallbackSteps = 256u; ++flowFallbacks;
        return STEP_RETRY;
    }
    if (hs.solveError) return fail(MI_ERR_DEVICE, "contact solver reported an error on the per-colour path");
    if (profileSolve) {
        if (useFlow) { mainContacts = 0; for (uint32_t bn = 0; bn + 1 < kSchedBins; ++bn) mainContacts += (uint64_t)bins[bn].count * ((bn & 3u) + 1u); }
        profContacts = mainContacts * iters;
        profKernelMs = 0.f;
        for (uint32_t l = 0; l < profLaunches; ++l) profKernelMs += elapsedMs(profEvents[2 * l], profEvents[2 * l + 1]);
    }
    if (spec && usesInteractions) {   // the trigger overlaps of this step, from the device-sorted interaction list (force fields were applied in-stream)
        std::vector<DeviceInteraction> list(eventsEnabled ? hs.numInteractions : 0u);
        if (!list.empty()) { HIP_TRY(hipMemcpyAsync(list.data(), interSorted.p, list.size() * sizeof(DeviceInteraction), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); }
        int rc = triggerEventsFrom(list, triggerEvents); if (rc != MI_OK) return rc;
    }
    if (eventsEnabled) {
        pendingEvents.insert(pendingEvents.end(), triggerEvents.begin(), triggerEvents.end());   // handleNonCollisionInteractions runs before the collision events
        if (!nmBound && tabValid) {   // no manifolds at all this step: every collision of the previous step ended
            uint32_t cap = last.numManifolds + 1024u;
            HIP_TRY(devEvents.ensure(cap));
            const int nt = tabCur ^ 1;
            HIP_TRY(tab[nt].ensure(1024)); tabMask[nt] = 1023u;
            HIP_TRY(hipMemsetAsync(tab[nt].p, 0, 1024 * sizeof(HistSlot), st));
            k_events_end<<<divUp(tabMask[tabCur] + 1u, B), B, 0, st>>>(cap, sc, tab[tabCur].p, tabMask[tabCur], tab[nt].p, tabMask[nt], devEvents.p);
            HIP_TRY(hipMemcpyAsync(hsPinned, sc, offsetof(Readback, seq), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            hs = hsPinned->sc;
        }
        if (hs.numEvents) {
            std::vector<DeviceEvent> ev_(hs.numEvents);
            HIP_TRY(hipMemcpyAsync(ev_.data(), devEvents.p, (size_t)hs.numEvents * sizeof(DeviceEvent), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            std::sort(ev_.begin(), ev_.end(), [](const DeviceEvent& x, const DeviceEvent& y) { return x.colliderA != y.colliderA ? x.colliderA < y.colliderA : x.colliderB < y.colliderB; });
            for (const DeviceEvent& d : ev_) {   // the reference's sorted merge visits the pairs in ascending (a, b) order
                mi_event e{}; e.type = d.type; e.collider_a = d.colliderA; e.collider_b = d.colliderB;
                e.entity_a = colliders[d.colliderA].entity; e.entity_b = colliders[d.colliderB].entity;
                for (int k = 0; k < 3; ++k) { e.point[k] = d.point[k]; e.normal[k] = d.normal[k]; e.relative_velocity[k] = d.relVel[k]; }
                pendingEvents.push_back(e);
            }
        }
    }
    // the step is valid: the freshly integrated state becomes the current one
    std::swap(bPos.p, bPosN.p); std::swap(bRot.p, bRotN.p); std::swap(bLinVel.p, bLinVelN.p); std::swap(bAngVel.p, bAngVelN.p);
    std::swap(bForce.p, bForceN.p); std::swap(bTorque.p, bTorqueN.p);
    if (usesInteractions) prevTriggerOverlaps.swap(nextTriggerOverlaps);
    sapAxis = hs.axisNext;
    if (nc) { gridCur ^= 1u; gridValid = true; gridNextCells = hs.numCellsNext; }   // the grid k_pair_finish prepared becomes the next step's
    if (nmBound) { tabCur ^= 1; tabValid = true; } else tabValid = false;
    hostStale = true;
    // The next step's launch sizes derive from these counts; they are kept as snug, STICKY upper bounds (12.5 % granules, unchanged while
    // they still fit) so that consecutive steps of a scene in a steady state enqueue identical work — which a captured graph can replay.
    auto sticky = [](uint32_t x, uint32_t prev, uint32_t minGranule) {
        const uint32_t g = std::max(minGranule, (x ? 1u << (31 - __builtin_clz(x)) : 1u) >> 3);
        if (prev >= x && prev - x <= 2u * g) return prev;
        return (x / g + 1u) * g;
    };
    last.numPairs = sticky(hs.numPairs, last.numPairs, 256); last.numManifolds = sticky(hs.numManifolds, last.numManifolds, 256);
    last.numContacts = sticky(hs.numContacts, last.numContacts, 256); last.numCells = sticky(hs.numCells, last.numCells, 1024);
    last.numSmall = sticky(nc - std::min(nc, hs.numLarge + hs.numDead), last.numSmall, 256); last.numLarge = sticky(hs.numLarge, last.numLarge, 16);
    last.gjkSpan = sticky(hs.gjkHi - hs.gjkLo, last.gjkSpan, 256);
    last.numInterPairs = sticky(hs.numInterPairs, last.numInterPairs, 256); last.numInteractions = sticky(hs.numInteractions, last.numInteractions, 256);
    for (int k = 0; k < 3; ++k) shard.owned[k] = hs.shardOwned[k];
    if (seamMode()) { seamLast[0] = hs.seamStats[0]; seamLast[1] = hs.seamStats[1]; seamViolations += hs.seamStats[2]; }
    shard.flagsSwapPending = shard.enabled; shard.stepOpen = false; shard.flagsOfAStep = shard.enabled;
    static const bool xcdStats = std::getenv("MI_XCD_STATS") != nullptr;   // development: how many bodies stayed XCD-local
    if (xcdStats && usedXcd && ((totalSteps % 50u) == 0u || std::getenv("MI_XCD_NOSORT"))) {
        std::vector<unsigned long long> own(nb);
        HIP_TRY(hipMemcpy(own.data(), bodyOwner.p, (size_t)nb * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        size_t loc = 0, shared = 0; for (unsigned long long o : own) { int c = __builtin_popcountll(o); loc += c == 1; shared += c > 1; }
        std::fprintf(stderr, "[mi_physics] step %llu: XCD-local bodies %zu, shared %zu, lists", (unsigned long long)totalSteps, loc, shared);
        for (int x = 0; x < 8; ++x) std::fprintf(stderr, " %u", hs.xcdCount[x]);
        std::fprintf(stderr, "\n");
    }
    haveXcdEstimate = usedXcd; lastXcdSingle = usedXcdSingle;
    if (usedXcd) { uint32_t m = 0; for (int x = 0; x < 8; ++x) m = std::max(m, hs.xcdCount[x]); lastXcdMax = sticky(m, lastXcdMax, 16); }
    last.colorRounds = 0;
    while (last.colorRounds < 96u && flagsHost[last.colorRounds]) ++last.colorRounds;   // rounds that still had work (+1 to commit) this step
    ++last.colorRounds;
    haveEstimates = true;
    pairsIn = hs.partitioned ? pairKeysS.p : pairKeys.p;
    lastPartitioned = hs.partitioned != 0u; havePartitionFlag = pairBound != 0u;

    counts.num_rigid_bodies = nb; counts.num_colliders = nc; counts.num_broadphase_overlaps = nc ? hs.numOverlaps : 0;
    manifoldsLast = pairBound ? hs.numManifolds : 0;
    counts.num_collisions = manifoldsLast - (manifoldsLast ? hs.numHmContacts - hs.numHmColliders : 0u);   // terrain: one collision per collider (heightmap_collision.cpp:582-594)
    counts.num_contacts = pairBound ? hs.numContacts : 0;
    counts.num_colors = numColorsUsed; counts.sorting_axis = hs.axisCur; counts.reserved = solveLaunches;
    // the step's device times: read from its events LATER (finishTimes), the next step records into the other set
    finishTimes();   // (normally done already, at the start of this step)
    timesPending = stepEventsMode == 2 || stageEvents; timesPendingSet = evSet; timesPendingStages = stageEvents; timesPendingUpdates = (uint64_t)counts.num_contacts * iters;
    if (!timesPending) { times = mi_stage_times{}; ++timesSteps; contactUpdatesSum += timesPendingUpdates; }   // timing off: no stale times, and the step / contact-update counts still add up
    evSet ^= 1; ev = evSets[evSet];
    static const bool eagerTimes = std::getenv("MI_EAGER_TIMES") != nullptr;   // development: read them right here, as before
    if (eagerTimes) finishTimes();
    return MI_OK;
}
// (the host gets here after the published read-back of the step the events belong to — but the HIP 7.0 runtime now and then still reports an event
// attached to a kernel as not ready, ~1 step in 1000: elapsedMs waits for it then instead of reporting 0 ms)
void mi_world::finishTimes() {
    if (!timesPending) return;
    timesPending = false;
    hipEvent_t* e = evSets[timesPendingSet];
    auto el = [&](int a, int b) { return elapsedMs(e[a], e[b]); };
    if (timesPendingStages) {
        times.world_colliders = el(0, 1); times.broadphase = el(1, 2); times.narrowphase = el(2, 3); times.integrate_forces = el(3, 4);
        times.schedule = el(4, 5); times.init_constraints = el(5, 6);
    } else { times.world_colliders = times.broadphase = times.narrowphase = times.integrate_forces = times.schedule = times.init_constraints = 0.f; }
    times.solve = el(6, 7); times.integrate_velocities = el(7, 8); times.total = el(0, 8);
    { float* a = &timesSum.world_colliders; const float* b = &times.world_colliders; for (int i = 0; i < 9; ++i) a[i] += b[i]; ++timesSteps;
      contactUpdatesSum += timesPendingUpdates; }
}

// mi_debug_set_solve_order, inside a synchronous step after the manifolds are known (hs = this step's counts): every manifold is given the
// sequential colour (kOverflowColor: one lane solves that bin slot by slot), and its rank in the caller's list is remembered for the slot order.
int mi_world::applyDebugOrder() {
    const uint32_t nm = hs.numManifolds;
    if (nm != debugOrder.size()) return fail(MI_ERR_INVALID_ARGUMENT, "mi_debug_set_solve_order: the step found " + std::to_string(nm) + " contact manifolds, the list holds " + std::to_string(debugOrder.size()));
    debugRank.assign(nm, 0u);
    if (!nm) return MI_OK;
    std::vector<uint32_t> mp(nm), col(nm, kOverflowColor);
    HIP_TRY(hipMemcpyAsync(mp.data(), manPair.p, nm * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    std::vector<uint64_t> keys(hs.numPairs);
    HIP_TRY(hipMemcpyAsync(keys.data(), hs.partitioned ? pairKeysS.p : pairKeys.p, keys.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::unordered_map<uint64_t, uint32_t> rank; rank.reserve(2 * (size_t)nm);
    for (uint32_t i = 0; i < nm; ++i) if (!rank.emplace(debugOrder[i], i).second) return fail(MI_ERR_INVALID_ARGUMENT, "mi_debug_set_solve_order: a collider pair is listed twice");
    for (uint32_t m = 0; m < nm; ++m) {
        if (mp[m] >= keys.size()) return fail(MI_ERR_DEVICE, "mi_debug_set_solve_order: manifold without a pair");
        const uint64_t k = keys[mp[m]] & ((1ull << 58) - 1ull);   // (bucket bits dropped: a << 29 | b)
        auto it = rank.find(k);
        if (it == rank.end()) return fail(MI_ERR_INVALID_ARGUMENT, "mi_debug_set_solve_order: the step found a contact manifold (colliders " + std::to_string(k >> 29) + ", " + std::to_string(k & 0x1FFFFFFFull) + ") that is not in the list");
        debugRank[m] = it->second;
    }
    HIP_TRY(hipMemcpyAsync(color.p, col.data(), nm * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return MI_OK;
}
// mi_debug_set_solve_order also ORIENTS: a pair of equal shape type whose AABB starts tie exactly on the sweep axis is oriented by the reference
// according to the history of its persistent, stably sorted endpoint array (collision_broad.cpp:386-398) — no rule of the current state reproduces
// that, the canonical rule (later created = new) is only the first frame's.  Synchronous step, right after the pair pass: pairs that the list holds
// the other way round (and not this way) are turned before the narrow phase sees them.
int mi_world::orientPairsLikeDebugOrder() {
    const uint32_t np = hs.numPairs;
    if (!np || debugOrder.empty()) return MI_OK;
    std::vector<uint64_t> keys(np);
    HIP_TRY(hipMemcpyAsync(keys.data(), pairKeys.p, (size_t)np * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::unordered_map<uint64_t, uint32_t> listed; listed.reserve(2 * debugOrder.size());
    for (uint64_t k : debugOrder) listed.emplace(k, 0u);
    bool turned = false;
    for (uint64_t& k : keys) {
        const uint64_t bucket = k >> 58, a = (k >> 29) & 0x1FFFFFFFull, b = k & 0x1FFFFFFFull;
        uint32_t ta = 0, rem = (uint32_t)bucket; while (rem >= 6u - ta) { rem -= 6u - ta; ++ta; }
        if (rem != 0u) continue;                                      // different shape types: ordered by type on both sides
        if (!listed.count((a << 29) | b) && listed.count((b << 29) | a)) { k = (bucket << 58) | (b << 29) | a; turned = true; }
    }
    if (turned) { HIP_TRY(hipMemcpyAsync(pairKeys.p, keys.data(), (size_t)np * sizeof(uint64_t), hipMemcpyHostToDevice, stream)); HIP_TRY(hipStreamSynchronize(stream)); }
    return MI_OK;
}
extern "C" {
MI_API int mi_debug_set_sweep_axis(mi_world* w, uint32_t axis) {
    if (!w || axis > 2u) return fail(MI_ERR_INVALID_ARGUMENT, "axis 0 | 1 | 2");
    w->sapAxis = axis;
    if (w->shard.enabled && w->shard.axisDev.p) { HIP_TRY(hipSetDevice(w->device)); HIP_TRY(hipMemcpy(w->shard.axisDev.p, &axis, sizeof(uint32_t), hipMemcpyHostToDevice)); w->shard.axisHostCurrent = true; }
    return MI_OK;
}
MI_API int mi_debug_set_solve_order(mi_world* w, const uint32_t* pairs, uint32_t count) {
    if (!w || (count && !pairs)) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    if (w->heightmap || w->shard.enabled) return fail(MI_ERR_UNSUPPORTED, "mi_debug_set_solve_order: not with heightmap terrain or sharding");   // (here, not in the step: a pending order would fail every later step)
    w->debugOrder.resize(count);
    for (uint32_t i = 0; i < count; ++i) {
        if (pairs[2 * i] >= (1u << 29) || pairs[2 * i + 1] >= (1u << 29)) return fail(MI_ERR_INVALID_ARGUMENT, "collider index out of range");
        w->debugOrder[i] = ((uint64_t)pairs[2 * i] << 29) | (uint64_t)pairs[2 * i + 1];
    }
    w->debugOrderPending = true;
    return MI_OK;
}
}

// Sizes of the next block step (blocks.hpp): blocks, tiles per block, LDS carve-up — from what the previous block step needed (BlockState), with slack, and STICKY:
// the numbers only move when a need outgrows them or falls far below, so consecutive steps enqueue identical launches (step graphs).  false: the scene does not fit the
// block path (more than 64 tiles or 160 KiB of LDS per block), the step takes the classic schedule.
bool mi_world::planBlocks(uint32_t nmLast, uint32_t nbBodies) {
    if (!nmLast) return false;
    BlockCaps c = blkCaps;
    const uint32_t per = c.nbe ? nmLast / c.nbe : 0u;
    if (!c.nbe || c.nbe > blkMaxBlocks || per < 160u || (per > 640u && c.nbe < blkMaxBlocks)) {
        const uint32_t nbe = std::min(blkMaxBlocks, std::max(1u, nmLast / 320u));
        if (nbe != c.nbe) { c = BlockCaps{}; c.nbe = nbe; haveBlkEstimate = false; }
    }
    const uint32_t perBlock = divUp(nmLast, c.nbe);
    auto sticky = [](uint32_t cap, uint32_t need, uint32_t slack, uint32_t quantum) {
        if (cap >= need + slack / 2u && cap <= 2u * need + 2u * slack) return cap;
        const uint32_t want = need + need / 8u + slack;
        return (want + quantum - 1u) / quantum * quantum;
    };
    const bool have = haveBlkEstimate;
    if (blkLastFailed) {   // the previous block step outgrew a capacity: the scene is changing fast (a pile landing), take a bigger stride than the usual slack
        BlockState& b = lastBlk;
        b.need += b.need / 2u; b.needExtra += b.needExtra / 2u + 16u; b.needBodies += b.needBodies / 2u; b.needPasses += b.needPasses / 2u + 4u; b.needImp += b.needImp / 2u;
        blkLastFailed = false;
    }
    const uint32_t needEntries = have && lastBlk.need ? lastBlk.need : perBlock + perBlock / 4u + 32u;
    c.tiles = sticky(c.tiles * 64u, needEntries, 64u, 64u) / 64u;
    if (c.tiles > 64u) return false;
    const uint32_t maxSlots = divUp(c.tiles, blkWaves);
    c.extraCap = sticky(c.extraCap, have ? lastBlk.needExtra : std::max(32u, perBlock / 6u), 32u, 32u);
    c.bodyCap = sticky(c.bodyCap, have && lastBlk.needBodies ? lastBlk.needBodies : std::min(nbBodies + 1u, nbBodies / c.nbe * 3u / 2u + 64u), 64u, 64u);
    if (c.bodyCap > 16384u) return false;
    c.hashSize = 256u; while (c.hashSize < 2u * c.bodyCap) c.hashSize <<= 1;
    c.maxPasses = sticky(c.maxPasses, have && lastBlk.needPasses ? lastBlk.needPasses : maxSlots * 3u + 16u, 8u, 8u);
    c.impCap = std::min(maxSlots * 256u, sticky(c.impCap, have && lastBlk.needImp ? lastBlk.needImp : (perBlock * 4u) / blkWaves + 128u, 64u, 64u));
    const size_t recBytes = ((size_t)c.bodyCap * 40u + 15u) & ~(size_t)15u;   // records (32 B), their bodies, their hand-over words
    const size_t uniBytes = (std::max((size_t)c.hashSize * 6u, (size_t)blkWaves * c.impCap * 8u) + 15u) & ~(size_t)15u;   // hash during set-up, impulses afterwards
    const size_t waveBytes = ((size_t)maxSlots * 64u * 16u + (size_t)c.maxPasses * 16u + (size_t)maxSlots * 64u * 2u + (size_t)maxSlots * 16u + 15u) & ~(size_t)15u;
    c.lds = recBytes + uniBytes + blkWaves * waveBytes;
    if (c.lds > 160u * 1024u - 64u) return false;
    blkCaps = c;
    return true;
}

// Host mirror of the device schedule (bins -> tiles), from the binStart table read back in StepScalars.
void mi_world::mirrorSchedule() {
    numColorsUsed = 0;
    uint32_t tiles = 0, ct = 0;
    for (uint32_t bn = 0; bn < kSchedBins; ++bn) {
        bool ovf = bn == kSchedBins - 1;
        uint32_t s0 = hs.binStart[bn], s1 = ovf ? hs.binStart[kColorBins] : hs.binStart[bn + 1];
        uint32_t stride = ovf ? 4u : (bn & 3u) + 1u;
        BinInfo bi{s0, s1 - s0, tiles, ct};
        uint32_t nt = divUp(bi.count, 64);
        tiles += nt; ct += nt * stride;
        bins[bn] = bi;
        if (bi.count) numColorsUsed = std::max(numColorsUsed, (ovf ? kOverflowColor : bn / 4u) + 1u);
    }
    totalTiles = tiles;
}

// ------------------------------------------------------------------------------------------------
// Joint storage (host) — addConstraint / add*ConstraintFromGlobalPoints (src/physics/physics.cpp:128-333)
// ------------------------------------------------------------------------------------------------
// A constraint POD may be handed over packed (sizeof(mi_*_constraint)) or AS THE REFERENCE'S STRUCT LIES IN MEMORY (src/physics/constraints.h): the
// fields are the same in the same order; a leading quat makes fixed_constraint and slider_constraint 16-byte aligned, i.e. 8 bytes of tail
// padding (40 -> 48, 72 -> 80: MI_REF_SIZEOF_*).  Only the fields are read / written, the padding is ignored / left untouched.
template <class P> constexpr uint32_t refSizeof() { return (std::is_same<P, mi_fixed_constraint>::value || std::is_same<P, mi_slider_constraint>::value) ? (uint32_t)((sizeof(P) + 15u) & ~15u) : (uint32_t)sizeof(P); }
static_assert(refSizeof<mi_distance_constraint>() == MI_REF_SIZEOF_DISTANCE_CONSTRAINT && refSizeof<mi_ball_constraint>() == MI_REF_SIZEOF_BALL_CONSTRAINT && refSizeof<mi_fixed_constraint>() == MI_REF_SIZEOF_FIXED_CONSTRAINT &&
              refSizeof<mi_hinge_constraint>() == MI_REF_SIZEOF_HINGE_CONSTRAINT && refSizeof<mi_cone_twist_constraint>() == MI_REF_SIZEOF_CONE_TWIST_CONSTRAINT && refSizeof<mi_slider_constraint>() == MI_REF_SIZEOF_SLIDER_CONSTRAINT, "reference struct sizes");
template <class P> static bool podSizeOk(uint32_t bytes) { return bytes == sizeof(P) || bytes == refSizeof<P>(); }
template <class JT>
static int jointAddTo(mi_world& w, JT& l, uint32_t ea, uint32_t eb, const void* pod, uint32_t bytes, uint32_t* out) {
    typedef typename std::remove_reference<decltype(l.pods[0])>::type P;
    if (!podSizeOk<P>(bytes)) return fail(MI_ERR_INVALID_ARGUMENT, "constraint pod size mismatch");
    if (ea >= w.entities.size() || eb >= w.entities.size() || w.entities[ea].rb < 0 || w.entities[eb].rb < 0)
        return fail(MI_ERR_INVALID_ARGUMENT, "both constraint entities must be rigid bodies");
    P p; std::memcpy(&p, pod, sizeof(P));
    const uint32_t handle = (uint32_t)l.denseOf.size();
    if (out) *out = handle;
    l.denseOf.push_back((int32_t)l.pods.size()); l.handleAt.push_back(handle);
    l.ents.push_back(make_uint2(ea, eb)); l.seq.push_back(w.joints.nextSeq++);
    l.pods.push_back(p);
    l.bodies.push_back(make_uint2((uint32_t)w.entities[ea].rb, (uint32_t)w.entities[eb].rb));
    return MI_OK;
}
// deleteConstraint / deleteAllConstraints / deleteAllConstraintsFromEntity — src/physics/physics.cpp:443-539
int JointSet::destroy(uint32_t type, uint32_t id) {
    bool ok = false;
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: ok = distance.destroy(id); break;
        case MI_CONSTRAINT_BALL: ok = ball.destroy(id); break;
        case MI_CONSTRAINT_FIXED: ok = fixed.destroy(id); break;
        case MI_CONSTRAINT_HINGE: ok = hinge.destroy(id); break;
        case MI_CONSTRAINT_CONE_TWIST: ok = cone.destroy(id); break;
        case MI_CONSTRAINT_SLIDER: ok = slider.destroy(id); break;
    }
    return ok ? MI_OK : fail(MI_ERR_INVALID_ARGUMENT, "bad constraint type or id");
}
void JointSet::destroyAll() { distance.clearAll(); ball.clearAll(); fixed.clearAll(); hinge.clearAll(); cone.clearAll(); slider.clearAll(); }
void JointSet::destroyOfEntity(uint32_t entity) {
    struct Hit { uint64_t seq; uint32_t type, handle; };
    std::vector<Hit> hits;
    auto scan = [&](uint32_t type, const auto& l) { for (size_t d = 0; d < l.pods.size(); ++d) if (l.ents[d].x == entity || l.ents[d].y == entity) hits.push_back(Hit{l.seq[d], type, l.handleAt[d]}); };
    scan(MI_CONSTRAINT_DISTANCE, distance); scan(MI_CONSTRAINT_BALL, ball); scan(MI_CONSTRAINT_FIXED, fixed);
    scan(MI_CONSTRAINT_HINGE, hinge); scan(MI_CONSTRAINT_CONE_TWIST, cone); scan(MI_CONSTRAINT_SLIDER, slider);
    std::sort(hits.begin(), hits.end(), [](const Hit& x, const Hit& y) { return x.seq > y.seq; });   // the entity's edge list is newest first
    for (const Hit& h : hits) (void)destroy(h.type, h.handle);
}
int JointSet::add(mi_world& w, uint32_t type, uint32_t ea, uint32_t eb, const void* pod, uint32_t bytes, uint32_t* out) {
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: return jointAddTo(w, distance, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_BALL: return jointAddTo(w, ball, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_FIXED: return jointAddTo(w, fixed, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_HINGE: return jointAddTo(w, hinge, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_CONE_TWIST: return jointAddTo(w, cone, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_SLIDER: return jointAddTo(w, slider, ea, eb, pod, bytes, out);
    }
    return fail(MI_ERR_INVALID_ARGUMENT, "bad constraint type");
}
template <class JT> static int jointCopy(JT& l, uint32_t id, void* dst, const void* src, uint32_t bytes) {
    typedef typename std::remove_reference<decltype(l.pods[0])>::type P;
    if (!podSizeOk<P>(bytes) || l.dense(id) < 0) return fail(MI_ERR_INVALID_ARGUMENT, "bad constraint id or pod size");
    if (src) std::memcpy(&l.pods[l.dense(id)], src, sizeof(P)); else std::memcpy(dst, &l.pods[l.dense(id)], sizeof(P));
    return MI_OK;
}
int JointSet::update(uint32_t type, uint32_t id, const void* pod, uint32_t bytes) {
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: distance.podsDirty = true; return jointCopy(distance, id, nullptr, pod, bytes);
        case MI_CONSTRAINT_BALL: ball.podsDirty = true; return jointCopy(ball, id, nullptr, pod, bytes);
        case MI_CONSTRAINT_FIXED: fixed.podsDirty = true; return jointCopy(fixed, id, nullptr, pod, bytes);
        case MI_CONSTRAINT_HINGE: hinge.podsDirty = true; return jointCopy(hinge, id, nullptr, pod, bytes);
        case MI_CONSTRAINT_CONE_TWIST: cone.podsDirty = true; return jointCopy(cone, id, nullptr, pod, bytes);
        case MI_CONSTRAINT_SLIDER: slider.podsDirty = true; return jointCopy(slider, id, nullptr, pod, bytes);
    }
    return fail(MI_ERR_INVALID_ARGUMENT, "bad constraint type");
}
int JointSet::uploadPods(hipStream_t st) {   // the host copy is authoritative for the PODs (the device never writes them)
    if (distance.podsDirty) HIP_TRY(distance.uploadPods(st));
    if (ball.podsDirty) HIP_TRY(ball.uploadPods(st));
    if (fixed.podsDirty) HIP_TRY(fixed.uploadPods(st));
    if (hinge.podsDirty) HIP_TRY(hinge.uploadPods(st));
    if (cone.podsDirty) HIP_TRY(cone.uploadPods(st));
    if (slider.podsDirty) HIP_TRY(slider.uploadPods(st));
    return MI_OK;
}
int JointSet::get(uint32_t type, uint32_t id, void* pod, uint32_t bytes) {
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: return jointCopy(distance, id, pod, nullptr, bytes);
        case MI_CONSTRAINT_BALL: return jointCopy(ball, id, pod, nullptr, bytes);
        case MI_CONSTRAINT_FIXED: return jointCopy(fixed, id, pod, nullptr, bytes);
        case MI_CONSTRAINT_HINGE: return jointCopy(hinge, id, pod, nullptr, bytes);
        case MI_CONSTRAINT_CONE_TWIST: return jointCopy(cone, id, pod, nullptr, bytes);
        case MI_CONSTRAINT_SLIDER: return jointCopy(slider, id, pod, nullptr, bytes);
    }
    return fail(MI_ERR_INVALID_ARGUMENT, "bad constraint type");
}
static void put3(float* f, V3 v) { f[0] = v.x; f[1] = v.y; f[2] = v.z; }
static void put4(float* f, Q4 q) { f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w; }
int JointSet::addFromGlobal(mi_world& w, uint32_t type, uint32_t ea, uint32_t eb, const float* anchor, const float* axisIn, float l0, float l1, uint32_t* out) {
    if (ea >= w.entities.size() || eb >= w.entities.size()) return fail(MI_ERR_INVALID_ARGUMENT, "entity out of range");
    const HEntity& A = w.entities[ea]; const HEntity& B = w.entities[eb];
    auto invPos = [](const HEntity& e, V3 p) { V3 r = rotate(conj(e.rot), p - e.pos); return V3(r.x / 1.f, r.y / 1.f, r.z / 1.f); };   // inverseTransformPosition (scale = 1)
    auto invDir = [](const HEntity& e, V3 d) { return rotate(conj(e.rot), d); };
    V3 ga(anchor[0], anchor[1], anchor[2]);
    V3 gx = axisIn ? V3(axisIn[0], axisIn[1], axisIn[2]) : V3();
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: {   // anchor = globalAnchorA, axis = globalAnchorB (physics.cpp:147-156)
            mi_distance_constraint c; put3(c.local_anchor_a, invPos(A, ga)); put3(c.local_anchor_b, invPos(B, gx)); c.global_length = len(ga - gx);
            return add(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_BALL: {
            mi_ball_constraint c; put3(c.local_anchor_a, invPos(A, ga)); put3(c.local_anchor_b, invPos(B, ga));
            return add(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_FIXED: {
            mi_fixed_constraint c; put3(c.local_anchor_a, invPos(A, ga)); put3(c.local_anchor_b, invPos(B, ga));
            put4(c.initial_inv_rotation_difference, conj(B.rot) * A.rot);
            return add(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_HINGE: {
            mi_hinge_constraint c;
            put3(c.local_anchor_a, invPos(A, ga)); put3(c.local_anchor_b, invPos(B, ga));
            V3 axA = invDir(A, gx), axB = invDir(B, gx);
            put3(c.local_hinge_axis_a, axA); put3(c.local_hinge_axis_b, axB);
            V3 t = tangentOf(axA), bt = cross(axA, t);
            put3(c.local_hinge_tangent_a, t); put3(c.local_hinge_bitangent_a, bt);
            put3(c.local_hinge_tangent_b, rotate(conj(B.rot), rotate(A.rot, t)));
            c.min_rotation_limit = l0; c.max_rotation_limit = l1;
            c.motor_type = MI_MOTOR_VELOCITY; c.motor_velocity_or_target_angle = 0.f; c.max_motor_torque = -1.f;
            return add(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_CONE_TWIST: {
            mi_cone_twist_constraint c;
            put3(c.local_anchor_a, invPos(A, ga)); put3(c.local_anchor_b, invPos(B, ga));
            c.swing_limit = l0; c.twist_limit = l1;
            V3 axA = invDir(A, gx), axB = invDir(B, gx);
            put3(c.local_limit_axis_a, axA); put3(c.local_limit_axis_b, axB);
            V3 t = tangentOf(axA), bt = cross(axA, t);
            put3(c.local_limit_tangent_a, t); put3(c.local_limit_bitangent_a, bt);
            put3(c.local_limit_tangent_b, rotate(conj(B.rot), rotate(A.rot, t)));
            c.swing_motor_type = MI_MOTOR_VELOCITY; c.swing_motor_velocity_or_target_angle = 0.f; c.max_swing_motor_torque = -1.f; c.swing_motor_axis = 0.f;
            c.twist_motor_type = MI_MOTOR_VELOCITY; c.twist_motor_velocity_or_target_angle = 0.f; c.max_twist_motor_torque = -1.f;
            return add(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_SLIDER: {
            mi_slider_constraint c;
            put3(c.local_anchor_a, invPos(A, ga)); put3(c.local_anchor_b, invPos(B, ga));
            put3(c.local_axis_a, invDir(A, gx));
            put4(c.initial_inv_rotation_difference, conj(B.rot) * A.rot);
            c.neg_distance_limit = l0; c.pos_distance_limit = l1;
            c.motor_type = MI_MOTOR_VELOCITY; c.motor_velocity_or_target_distance = 0.f; c.max_motor_force = -1.f;
            return add(w, type, ea, eb, &c, sizeof(c), out);
        }
    }
    return fail(MI_ERR_INVALID_ARGUMENT, "bad constraint type");
}
int JointSet::upload(mi_world& w, hipStream_t st) {
    if (!count()) {   // the last constraint may just have been deleted: nothing of the previous topology may survive
        releaseIslands();
        distance.order.clear(); ball.order.clear(); fixed.order.clear(); hinge.order.clear(); cone.order.clear(); slider.order.clear();
        return MI_OK;
    }
    std::vector<float> invMass(w.bodies.size());
    for (size_t i = 0; i < w.bodies.size(); ++i) invMass[i] = w.bodies[i].invMass;
    distance.computeOrder(invMass); ball.computeOrder(invMass); fixed.computeOrder(invMass);
    hinge.computeOrder(invMass); cone.computeOrder(invMass); slider.computeOrder(invMass);
    {
        std::vector<IslandDesc> islands; std::vector<IslandStep> steps; std::vector<uint32_t> islandBodies;
        static const bool useIslands = !(std::getenv("MI_JOINT_ISLANDS") && std::getenv("MI_JOINT_ISLANDS")[0] == '0');
        if (useIslands) buildIslands(invMass, islands, steps, islandBodies);
        releaseIslands();
        if (!islands.empty()) {
            HIP_TRY(hipMalloc((void**)&dIslands, islands.size() * sizeof(IslandDesc)));
            HIP_TRY(hipMalloc((void**)&dSteps, steps.size() * sizeof(IslandStep))); HIP_TRY(hipMalloc((void**)&dIslandBodies, islandBodies.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpyAsync(dIslands, islands.data(), islands.size() * sizeof(IslandDesc), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(dSteps, steps.data(), steps.size() * sizeof(IslandStep), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(dIslandBodies, islandBodies.data(), islandBodies.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            std::vector<uint8_t> bodyJ(invMass.size() + 1, 0);
            for (uint32_t b : islandBodies) if (b < invMass.size() && invMass[b] != 0.f) bodyJ[b] = 1;
            HIP_TRY(hipMalloc((void**)&dBodyJ, bodyJ.size()));
            HIP_TRY(hipMemcpyAsync(dBodyJ, bodyJ.data(), bodyJ.size(), hipMemcpyHostToDevice, st));
            // private islands (joints.hpp): island of every dynamic island body; per-step state and the islands' manifold lists
            std::vector<uint32_t> bodyIsland(invMass.size() + 1, 0xFFFFFFFFu);
            for (uint32_t i = 0; i < (uint32_t)islands.size(); ++i)
                for (uint32_t k = 0; k < islands[i].numBodies; ++k) { const uint32_t b = islandBodies[islands[i].bodyBegin + k]; if (b < invMass.size() && invMass[b] != 0.f) bodyIsland[b] = i; }
            HIP_TRY(hipMalloc((void**)&dBodyIsland, bodyIsland.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpyAsync(dBodyIsland, bodyIsland.data(), bodyIsland.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMalloc((void**)&dIslState, 3 * islands.size() * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void**)&dIslEntries, islands.size() * (size_t)mi::kIslandMaxContacts * sizeof(uint4)));
            HIP_TRY(hipStreamSynchronize(st));   // the staging vectors are locals
            numIslands = (uint32_t)islands.size();
        }
    }
    distance.finishOrder(); ball.finishOrder(); fixed.finishOrder(); hinge.finishOrder(); cone.finishOrder(); slider.finishOrder();
    HIP_TRY(distance.upload(st)); HIP_TRY(ball.upload(st)); HIP_TRY(fixed.upload(st));
    HIP_TRY(hinge.upload(st)); HIP_TRY(cone.upload(st)); HIP_TRY(slider.upload(st));
    distance.podsDirty = ball.podsDirty = fixed.podsDirty = hinge.podsDirty = cone.podsDirty = slider.podsDirty = false;
    return MI_OK;
}
// Connected components of the joint graph over DYNAMIC bodies; an island small enough for one wave gets a program of
// (type, colour) groups in canonical order, the rest stays with the per-colour kernels.
void JointSet::buildIslands(const std::vector<float>& invMass, std::vector<IslandDesc>& islands, std::vector<IslandStep>& steps, std::vector<uint32_t>& islandBodies) {
    const uint32_t nb = (uint32_t)invMass.size();
    std::vector<uint32_t> parent(nb);
    for (uint32_t i = 0; i < nb; ++i) parent[i] = i;
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    struct Ref { uint32_t type, joint, color; uint2 bodies; };
    std::vector<Ref> all;
    auto collect = [&](uint32_t type, const std::vector<uint2>& bodies, const std::vector<uint32_t>& colorOf) {
        for (uint32_t j = 0; j < (uint32_t)bodies.size(); ++j) {
            uint2 b = bodies[j];
            bool dynA = b.x < nb && invMass[b.x] != 0.f, dynB = b.y < nb && invMass[b.y] != 0.f;
            if (dynA && dynB) parent[find(b.x)] = find(b.y);
            all.push_back(Ref{type, j, colorOf[j], b});
        }
    };
    collect(0, distance.bodies, distance.colorOf); collect(1, ball.bodies, ball.colorOf); collect(2, fixed.bodies, fixed.colorOf);
    collect(3, hinge.bodies, hinge.colorOf); collect(4, cone.bodies, cone.colorOf); collect(5, slider.bodies, slider.colorOf);
    // joints by island root (a joint between two non-dynamic bodies does nothing; it stays with the per-colour kernels)
    std::vector<std::vector<uint32_t>> byRoot(nb);
    for (uint32_t r = 0; r < (uint32_t)all.size(); ++r) {
        const Ref& ref = all[r];
        bool dynA = ref.bodies.x < nb && invMass[ref.bodies.x] != 0.f, dynB = ref.bodies.y < nb && invMass[ref.bodies.y] != 0.f;
        if (dynA || dynB) byRoot[find(dynA ? ref.bodies.x : ref.bodies.y)].push_back(r);
    }
    uint8_t* flags[6] = {distance.inIsland.data(), ball.inIsland.data(), fixed.inIsland.data(), hinge.inIsland.data(), cone.inIsland.data(), slider.inIsland.data()};
    for (uint32_t root = 0; root < nb; ++root) {
        std::vector<uint32_t>& js = byRoot[root];
        if (js.empty()) continue;
        std::stable_sort(js.begin(), js.end(), [&](uint32_t x, uint32_t y) { return all[x].type != all[y].type ? all[x].type < all[y].type : all[x].color < all[y].color; });
        std::vector<uint32_t> slots;   // island-local body table (dynamic and static bodies alike; the static dummy is body index nb)
        auto slotOf = [&](uint32_t body) { for (uint32_t k = 0; k < (uint32_t)slots.size(); ++k) if (slots[k] == body) return k; slots.push_back(body); return (uint32_t)slots.size() - 1u; };
        bool fits = js.size() <= kIslandMaxJoints;
        std::vector<IslandStep> st;
        IslandDesc d{};
        uint32_t group = 0;   // groups are numbered type-major, colour-major within the island
        for (size_t k = 0; fits && k < js.size(); ++k) {
            const Ref& ref = all[js[k]];
            if (ref.color >= 64u) { fits = false; break; }
            if (k > 0 && (all[js[k - 1]].type != ref.type || all[js[k - 1]].color != ref.color)) ++group;
            st.push_back(IslandStep{ref.joint, (uint16_t)slotOf(ref.bodies.x), (uint16_t)slotOf(ref.bodies.y), (uint16_t)ref.type, (uint16_t)group});
            d.typeGroups[ref.type + 1] = group + 1;   // end of this type's groups so far
            if (slots.size() > kIslandMaxBodies) fits = false;
        }
        if (!fits) continue;
        for (uint32_t t = 1; t <= 6; ++t) d.typeGroups[t] = std::max(d.typeGroups[t], d.typeGroups[t - 1]);   // absent types: empty range
        d.bodyBegin = (uint32_t)islandBodies.size(); d.numBodies = (uint32_t)slots.size(); d.stepBegin = (uint32_t)steps.size(); d.numJoints = (uint32_t)st.size();
        steps.insert(steps.end(), st.begin(), st.end());
        islandBodies.insert(islandBodies.end(), slots.begin(), slots.end());
        islands.push_back(d);
        for (uint32_t r : js) flags[all[r].type][all[r].joint] = 1;
    }
}
static BodyView bodyView(mi_world& w) { return BodyView{w.gPos.p, w.gInvI.p, w.gVel.p, w.bRot.p, w.bCogInvMass.p, w.shard.enabled ? w.shard.active.p : nullptr}; }
int JointSet::initialize(mi_world& w, float dt, hipStream_t st) {
    if (!count()) return MI_OK;
    BodyView bv = bodyView(w);
    uint32_t dummy = (uint32_t)w.bodies.size();
    mi::Launcher& L = w.L;
    distance.launchInit(L, dummy, bv, dt, st); ball.launchInit(L, dummy, bv, dt, st); fixed.launchInit(L, dummy, bv, dt, st);
    hinge.launchInit(L, dummy, bv, dt, st); cone.launchInit(L, dummy, bv, dt, st); slider.launchInit(L, dummy, bv, dt, st);
    return MI_OK;
}
int JointSet::solveIterationReference(mi_world& w, hipStream_t st) {
    if (!count()) return MI_OK;
    BodyView bv = bodyView(w);
    mi::Launcher& L = w.L;
    HIP_TRY(distance.launchSolveReference(L, bv, st)); HIP_TRY(ball.launchSolveReference(L, bv, st)); HIP_TRY(fixed.launchSolveReference(L, bv, st));
    HIP_TRY(hinge.launchSolveReference(L, bv, st)); HIP_TRY(cone.launchSolveReference(L, bv, st)); HIP_TRY(slider.launchSolveReference(L, bv, st));
    return MI_OK;
}
void JointSet::solveIteration(mi_world& w, hipStream_t st) {
    if (!count()) return;
    BodyView bv = bodyView(w);
    mi::Launcher& L = w.L;
    if (numIslands) L.launch(k_joint_islands, dim3(numIslands), dim3(64), 0, st, dIslands, dSteps, dIslandBodies, IslandUpd{distance.dUpd, ball.dUpd, fixed.dUpd, hinge.dUpd, cone.dUpd, slider.dUpd}, bv);
    distance.launchSolve(L, bv, st); ball.launchSolve(L, bv, st); fixed.launchSolve(L, bv, st);
    hinge.launchSolve(L, bv, st); cone.launchSolve(L, bv, st); slider.launchSolve(L, bv, st);
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

MI_API const char* mi_last_error(void) { return g_lastError.c_str(); }
MI_API int mi_version(void) { return 1; }

MI_API int mi_world_create(const mi_world_desc* desc, mi_world** out) {
    if (!out) return fail(MI_ERR_INVALID_ARGUMENT, "out_world is null");
    mi_world* w = new mi_world();
    int rc = w->init(desc ? desc->device : 0);
    if (rc != MI_OK) { delete w; *out = nullptr; return rc; }
    *out = w;
    return MI_OK;
}
MI_API void mi_world_destroy(mi_world* w) { delete w; }

MI_API int mi_entities_create(mi_world* w, uint32_t count, const mi_entity_desc* descs, uint32_t* out_first) {
    if (!w || (count && !descs)) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    int rc = w->download(); if (rc != MI_OK) return rc;
    if (out_first) *out_first = (uint32_t)w->entities.size();
    for (uint32_t i = 0; i < count; ++i) {
        const mi_entity_desc& d = descs[i];
        HEntity e; e.pos = V3(d.position[0], d.position[1], d.position[2]); e.rot = Q4(d.rotation[0], d.rotation[1], d.rotation[2], d.rotation[3]); e.kind = d.kind;
        if (d.kind > MI_ENTITY_FORCE_FIELD) return fail(MI_ERR_INVALID_ARGUMENT, "bad entity kind");
        if (d.kind == MI_ENTITY_FORCE_FIELD) { e.kindIndex = (uint32_t)w->ffEntities.size(); w->ffEntities.push_back((uint32_t)w->entities.size()); }
        if (d.kind == MI_ENTITY_TRIGGER) { e.kindIndex = (uint32_t)w->triggerEntities.size(); w->triggerEntities.push_back((uint32_t)w->entities.size()); }
        if (d.kind == MI_ENTITY_DYNAMIC || d.kind == MI_ENTITY_KINEMATIC) {
            HBody b;
            b.entity = (uint32_t)w->entities.size();
            bool kin = d.kind == MI_ENTITY_KINEMATIC;   // rigid_body_component ctor, rigid_body.cpp:6-27
            b.invMass = kin ? 0.f : 1.f; b.invInertia = kin ? M3::zero() : M3::identity();
            b.gravityFactor = d.gravity_factor; b.linDamp = d.linear_damping; b.angDamp = d.angular_damping;
            b.linVel = V3(d.linear_velocity[0], d.linear_velocity[1], d.linear_velocity[2]);
            b.angVel = V3(d.angular_velocity[0], d.angular_velocity[1], d.angular_velocity[2]);
            b.p0 = b.p1 = e.pos; b.r0 = b.r1 = e.rot;
            e.rb = (int)w->bodies.size();
            w->bodies.push_back(b);
        }
        w->entities.push_back(e);
    }
    w->topologyDirty = true;
    return MI_OK;
}
MI_API int mi_entity_create(mi_world* w, const mi_entity_desc* d, uint32_t* out) { return mi_entities_create(w, 1, d, out); }
// game_scene::deleteEntity — src/scene/scene.cpp:124-150.  Host-side pool bookkeeping with EnTT's swap-and-pop (the LAST collider /
// rigid body / trigger / force field moves into the freed slot: world indices follow the pools, so the order downstream stages
// see changes exactly like the reference's); the scene is re-uploaded before the next step.  The colour history and the previous
// collision / trigger-overlap lists are keyed by pool positions and restart.
MI_API int mi_entity_destroy(mi_world* w, uint32_t entity) {
    if (!w || entity >= w->entities.size() || w->entities[entity].kind == MI_ENTITY_DESTROYED) return fail(MI_ERR_INVALID_ARGUMENT, "bad entity");
    int rc = w->download(); if (rc != MI_OK) return rc;
    const size_t mine = w->entities[entity].colliders.size();
    for (size_t k = 0; k < mine; ++k) {                       // the entity's colliders, newest first (removeColliderFromBroadphase + destroy)
        HEntity& e = w->entities[entity];
        const uint32_t id = e.colliders[0], last = (uint32_t)w->colliders.size() - 1u;
        e.colliders.erase(e.colliders.begin());
        if (id != last) {
            w->colliders[id] = w->colliders[last];
            for (uint32_t& c : w->entities[w->colliders[id].entity].colliders) if (c == last) c = id;
        }
        w->colliders.pop_back();
    }
    w->joints.destroyOfEntity(entity);                         // deleteAllConstraintsFromEntity
    HEntity& e = w->entities[entity];
    if (e.rb >= 0) {
        const uint32_t p = (uint32_t)e.rb, last = (uint32_t)w->bodies.size() - 1u;
        if (p != last) {
            w->bodies[p] = w->bodies[last]; w->entities[w->bodies[p].entity].rb = (int)p;
            JointSet& j = w->joints;   // the reference derives body pairs from the entities every step (physics.cpp:789-806): re-point the cached ones
            auto fix = [&](auto& t) { for (uint2& b : t.bodies) { if (b.x == last) b.x = p; if (b.y == last) b.y = p; } };
            fix(j.distance); fix(j.ball); fix(j.fixed); fix(j.hinge); fix(j.cone); fix(j.slider);
        }
        w->bodies.pop_back();
    }
    auto dropFrom = [&](std::vector<uint32_t>& pool) {
        const uint32_t p = e.kindIndex, last = (uint32_t)pool.size() - 1u;
        if (p != last) { pool[p] = pool[last]; w->entities[pool[p]].kindIndex = p; }
        pool.pop_back();
    };
    if (e.kind == MI_ENTITY_TRIGGER) dropFrom(w->triggerEntities);
    if (e.kind == MI_ENTITY_FORCE_FIELD) dropFrom(w->ffEntities);
    e.kind = MI_ENTITY_DESTROYED; e.rb = -1; e.colliders.clear(); e.pos = V3(0.f, 0.f, 0.f); e.rot = Q4(0.f, 0.f, 0.f, 1.f);
    w->tabValid = false; w->prevTriggerOverlaps.clear();
    w->topologyDirty = true; w->haveEstimates = false;
    return MI_OK;
}
// ---- cloth (cloth_component, src/physics/cloth.h:5-60)
static V3 clothParticlePosition(const mi_cloth_desc& d, float relX, float relY) {   // getParticlePosition, cloth.cpp:126-132
    V3 p(relX * d.width, -relY * d.height, 0.f);
    p.x -= d.width * 0.5f;
    float t = p.y; p.y = p.z; p.z = t;
    return p;
}
MI_API int mi_cloth_create(mi_world* w, const mi_cloth_desc* d, uint32_t* out) {   // cloth_component ctor, cloth.cpp:7-85
    if (!w || !d) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (d->grid_size_x < 2 || d->grid_size_y < 2 || !(d->total_mass > 0.f) || !(d->stiffness > 0.f)) return fail(MI_ERR_INVALID_ARGUMENT, "bad cloth description");
    HIP_TRY(hipSetDevice(w->device));
    mi_world::HCloth* c = new mi_world::HCloth();
    c->desc = *d; c->oldTotalMass = d->total_mass; c->oldStiffness = d->stiffness;
    const uint32_t gx = d->grid_size_x, gy = d->grid_size_y, n = gx * gy;
    const float invMassPerParticle = (float)n / d->total_mass;
    std::vector<float4> pos(n);
    c->invMasses.resize(n);
    for (uint32_t y = 0; y < gy; ++y)
        for (uint32_t x = 0; x < gx; ++x) {
            const float im = (y == 0) ? 0.f : invMassPerParticle;   // upper row locked
            V3 p = clothParticlePosition(*d, (float)x / (float)(gx - 1), (float)y / (float)(gy - 1));
            pos[y * gx + x] = make_float4(p.x, p.y, p.z, im); c->invMasses[y * gx + x] = im;
        }
    std::vector<uint32_t> colours;
    auto add = [&](uint32_t a, uint32_t b, uint32_t colour) {
        V3 d_ = V3(pos[a].x, pos[a].y, pos[a].z) - V3(pos[b].x, pos[b].y, pos[b].z);
        c->pairs.push_back(make_uint2(a, b));
        c->restInvMass.push_back(make_float2(len(d_), (c->invMasses[a] + c->invMasses[b]) / d->stiffness));
        colours.push_back(colour);
    };
    for (uint32_t y = 0; y < gy; ++y)
        for (uint32_t x = 0; x < gx; ++x) {   // creation order of cloth.cpp:46-80; colour = family x parity (cloth.hpp)
            const uint32_t i = y * gx + x;
            if (x < gx - 1) add(i, i + 1, 0 + (x & 1u));
            if (y < gy - 1) add(i, i + gx, 2 + (y & 1u));
            if (x < gx - 1 && y < gy - 1) { add(i, i + gx + 1, 4 + (x & 1u)); add(i + gx, i + 1, 6 + (x & 1u)); }
            if (x < gx - 2) add(i, i + 2, 8 + ((x >> 1) & 1u));
            if (y < gy - 2) add(i, i + gx * 2, 10 + ((y >> 1) & 1u));
        }
    const uint32_t nc = (uint32_t)c->pairs.size();
    c->order.resize(nc);
    for (uint32_t k = 0; k < nc; ++k) c->order[k] = k;
    std::stable_sort(c->order.begin(), c->order.end(), [&](uint32_t a, uint32_t b) { return colours[a] < colours[b]; });
    std::memset(c->colourOffsets, 0, sizeof(c->colourOffsets));
    for (uint32_t k = 0; k < nc; ++k) c->colourOffsets[colours[k] + 1]++;
    for (int k = 0; k < 12; ++k) c->colourOffsets[k + 1] += c->colourOffsets[k];
    int rc = MI_OK;
    auto up = [&]() -> int {
        HIP_TRY(c->pos.ensure(n)); HIP_TRY(c->prev.ensure(n)); HIP_TRY(c->vel.ensure(n)); HIP_TRY(c->force.ensure(n)); HIP_TRY(c->temp.ensure(nc));
        HIP_TRY(c->dPairs.ensure(nc)); HIP_TRY(c->dRestInvMass.ensure(nc)); HIP_TRY(c->dOrder.ensure(nc));
        HIP_TRY(hipMemcpy(c->pos.p, pos.data(), n * sizeof(float4), hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(c->prev.p, pos.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(c->vel.p, 0, n * sizeof(float4))); HIP_TRY(hipMemset(c->force.p, 0, n * sizeof(float4)));
        HIP_TRY(hipMemcpy(c->dPairs.p, c->pairs.data(), nc * sizeof(uint2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->dOrder.p, c->order.data(), nc * sizeof(uint32_t), hipMemcpyHostToDevice));
        return MI_OK;
    };
    rc = up();
    if (rc != MI_OK) { delete c; return rc; }
    if (out) *out = (uint32_t)w->cloths.size();
    w->cloths.push_back(c); w->clothsDirty = true;
    return MI_OK;
}
MI_API int mi_cloth_set_fixed_vertices(mi_world* w, uint32_t cloth, const float* p3, const float* r4, uint32_t moveRigid) {   // cloth.cpp:87-124
    if (!w || cloth >= w->cloths.size() || !p3 || !r4) return fail(MI_ERR_INVALID_ARGUMENT, "bad cloth / null argument");
    HIP_TRY(hipSetDevice(w->device));
    mi_world::HCloth& c = *w->cloths[cloth];
    const uint32_t gx = c.desc.grid_size_x, gy = c.desc.grid_size_y, n = gx * gy;
    std::vector<float4> pos(n);
    HIP_TRY(hipStreamSynchronize(w->stream));
    HIP_TRY(hipMemcpy(pos.data(), c.pos.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    const V3 tp(p3[0], p3[1], p3[2]); const Q4 tr(r4[0], r4[1], r4[2], r4[3]);
    auto xf = [&](V3 p) { return rotate(tr, p) + tp; };
    auto at = [&](uint32_t i) { return V3(pos[i].x, pos[i].y, pos[i].z); };
    if (moveRigid) {
        V3 pivot = (gx % 2 == 1) ? at(gx / 2) : (at(gx / 2) + at(gx / 2 - 1)) * 0.5f;
        V3 currentAxis = normalize(at(gx - 1) - at(0));
        V3 newAxis = normalize(xf(clothParticlePosition(c.desc, 1.f, 0.f)) - xf(clothParticlePosition(c.desc, 0.f, 0.f)));
        V3 newPivot = xf(clothParticlePosition(c.desc, 0.5f, 0.f));
        Q4 deltaRotation = rotateFromTo(currentAxis, newAxis);
        for (uint32_t y = 1; y < gy; ++y)
            for (uint32_t x = 0; x < gx; ++x) { V3 q = rotate(deltaRotation, at(y * gx + x) - pivot) + newPivot; float4& o = pos[y * gx + x]; o.x = q.x; o.y = q.y; o.z = q.z; }
    }
    for (uint32_t x = 0; x < gx; ++x) { V3 q = xf(clothParticlePosition(c.desc, (float)x / (float)(gx - 1), 0.f)); pos[x].x = q.x; pos[x].y = q.y; pos[x].z = q.z; }
    HIP_TRY(hipMemcpy(c.pos.p, pos.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    return MI_OK;
}
MI_API int mi_cloth_set_properties(mi_world* w, uint32_t cloth, float totalMass, float stiffness, float damping, float gravityFactor) {
    if (!w || cloth >= w->cloths.size()) return fail(MI_ERR_INVALID_ARGUMENT, "bad cloth");
    mi_cloth_desc& d = w->cloths[cloth]->desc;
    d.total_mass = totalMass; d.stiffness = stiffness; d.damping = damping; d.gravity_factor = gravityFactor;
    w->clothsDirty = true;
    return MI_OK;
}
MI_API int mi_cloth_get_state(mi_world* w, uint32_t cloth, float* outPos, float* outVel, uint32_t cap) {
    if (!w || cloth >= w->cloths.size()) return fail(MI_ERR_INVALID_ARGUMENT, "bad cloth");
    HIP_TRY(hipSetDevice(w->device));
    mi_world::HCloth& c = *w->cloths[cloth];
    const uint32_t n = c.desc.grid_size_x * c.desc.grid_size_y;
    if (cap < n) return fail(MI_ERR_CAPACITY, "capacity < particles");
    std::vector<float4> buf(n);
    HIP_TRY(hipStreamSynchronize(w->stream));
    if (outPos) { HIP_TRY(hipMemcpy(buf.data(), c.pos.p, n * sizeof(float4), hipMemcpyDeviceToHost)); for (uint32_t i = 0; i < n; ++i) { outPos[3 * i] = buf[i].x; outPos[3 * i + 1] = buf[i].y; outPos[3 * i + 2] = buf[i].z; } }
    if (outVel) { HIP_TRY(hipMemcpy(buf.data(), c.vel.p, n * sizeof(float4), hipMemcpyDeviceToHost)); for (uint32_t i = 0; i < n; ++i) { outVel[3 * i] = buf[i].x; outVel[3 * i + 1] = buf[i].y; outVel[3 * i + 2] = buf[i].z; } }
    return MI_OK;
}
MI_API int mi_world_set_cloth_iterations(mi_world* w, uint32_t v, uint32_t p, uint32_t d) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    w->clothIterations[0] = v; w->clothIterations[1] = p; w->clothIterations[2] = d;
    return MI_OK;
}

// ---- heightmap terrain (heightmap_collider_component, src/terrain/heightmap_collider.h:126-151)
MI_API int mi_heightmap_create(mi_world* w, uint32_t chunksPerDim, float chunkSize, float restitution, float friction) {
    if (!w || !chunksPerDim || !(chunkSize > 0.f)) return fail(MI_ERR_INVALID_ARGUMENT, "bad heightmap parameters");
    if (w->heightmap) return fail(MI_ERR_INVALID_ARGUMENT, "a world holds one heightmap");
    if (w->colliders.size() >= kHeightmapVirtualBase) return fail(MI_ERR_CAPACITY, "collider index space");
    w->heightmap = new mi_world::HHeightmap();
    w->heightmap->chunksPerDim = chunksPerDim; w->heightmap->chunkSize = chunkSize; w->heightmap->restitution = restitution; w->heightmap->friction = friction;
    w->heightmap->heights.resize((size_t)chunksPerDim * chunksPerDim);
    w->topologyDirty = true;
    return MI_OK;
}
MI_API int mi_heightmap_set_chunk_heights(mi_world* w, uint32_t x, uint32_t z, const uint16_t* heights) {
    if (!w || !w->heightmap || !heights) return fail(MI_ERR_INVALID_ARGUMENT, "no heightmap / null heights");
    if (x >= w->heightmap->chunksPerDim || z >= w->heightmap->chunksPerDim) return fail(MI_ERR_INVALID_ARGUMENT, "chunk out of range");
    w->heightmap->heights[(size_t)z * w->heightmap->chunksPerDim + x].assign(heights, heights + kHmVerts * kHmVerts);
    w->heightmap->dirty = true; w->topologyDirty = true;
    return MI_OK;
}
MI_API int mi_heightmap_update(mi_world* w, const float* minCorner, float amplitudeScale) {
    if (!w || !w->heightmap || !minCorner) return fail(MI_ERR_INVALID_ARGUMENT, "no heightmap / null corner");
    w->heightmap->minCorner = V3(minCorner[0], minCorner[1], minCorner[2]); w->heightmap->amplitudeScale = amplitudeScale;
    w->topologyDirty = true;
    return MI_OK;
}
MI_API int mi_heightmap_get_height(mi_world* w, float x, float z, float* out) {
    if (!w || !w->heightmap || !out) return fail(MI_ERR_INVALID_ARGUMENT, "no heightmap / null out");
    HIP_TRY(hipSetDevice(w->device));
    int rc = w->uploadHeightmap(); if (rc != MI_OK) return rc;   // refreshes the host mirror + parameters if needed
    HeightmapParams q = w->hmParams;
    q.heights = w->hmHostHeights.data(); q.chunkSlot = w->hmHostSlots.data(); q.mips = nullptr;
    *out = hmHeightAt(q, x, z);
    return MI_OK;
}
MI_API int mi_entity_set_force(mi_world* w, uint32_t entity, const float* force) {   // force_field_component::force (physics.h:35-38)
    if (!w || !force) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (entity >= w->entities.size() || w->entities[entity].kind != MI_ENTITY_FORCE_FIELD) return fail(MI_ERR_INVALID_ARGUMENT, "not a force-field entity");
    w->entities[entity].force = V3(force[0], force[1], force[2]);
    w->topologyDirty = true;   // the rotated forces are part of the uploaded topology
    return MI_OK;
}

MI_API int mi_colliders_add(mi_world* w, uint32_t count, const uint32_t* ents, const mi_collider_desc* descs) {
    if (!w || (count && (!ents || !descs))) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    int rc = w->download(); if (rc != MI_OK) return rc;
    for (uint32_t i = 0; i < count; ++i) {
        if (ents[i] >= w->entities.size()) return fail(MI_ERR_INVALID_ARGUMENT, "entity out of range");
        if (descs[i].type >= MI_COLLIDER_TYPE_COUNT) return fail(MI_ERR_INVALID_ARGUMENT, "bad collider type");
        if (descs[i].type == MI_COLLIDER_HULL && descs[i].hull_geometry >= w->hulls.size()) return fail(MI_ERR_INVALID_ARGUMENT, "bad hull geometry");
        HCollider c; c.entity = ents[i]; c.desc = descs[i];
        uint32_t id = (uint32_t)w->colliders.size();
        w->colliders.push_back(c);
        HEntity& e = w->entities[ents[i]];
        e.colliders.insert(e.colliders.begin(), id);   // linked-list prepend (src/scene/scene.h:52-54)
    }
    if (w->colliders.size() >= (1u << kIndexBits)) return fail(MI_ERR_CAPACITY, "collider index space is 26 bits per world");
    w->topologyDirty = true;
    return MI_OK;
}
MI_API int mi_collider_add(mi_world* w, uint32_t entity, const mi_collider_desc* d, uint32_t* out) {
    if (out && w) *out = (uint32_t)w->colliders.size();
    return mi_colliders_add(w, 1, &entity, d);
}

MI_API int mi_hull_geometry_create(mi_world* w, const float* v, uint32_t nv, const uint32_t* t, uint32_t nt, uint32_t* out) {
    if (!w || !v || !nv || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    HHull g; g.mn = V3(FLT_MAX); g.mx = V3(-FLT_MAX);
    for (uint32_t i = 0; i < nv; ++i) { V3 p(v[3 * i], v[3 * i + 1], v[3 * i + 2]); g.verts.push_back(p); g.mn = vmin(g.mn, p); g.mx = vmax(g.mx, p); }
    if (t) g.tris.assign(t, t + 3 * (size_t)nt);
    *out = (uint32_t)w->hulls.size();
    w->hulls.push_back(std::move(g));
    w->topologyDirty = true;
    return MI_OK;
}

MI_API int mi_constraint_create(mi_world* w, uint32_t type, uint32_t ea, uint32_t eb, const void* pod, uint32_t bytes, uint32_t* out) {
    if (!w || !pod) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    int rc = w->joints.add(*w, type, ea, eb, pod, bytes, out);
    if (rc == MI_OK) w->topologyDirty = true;
    return rc;
}
MI_API int mi_constraint_update(mi_world* w, uint32_t type, uint32_t id, const void* pod, uint32_t bytes) {
    if (!w || !pod) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    return w->joints.update(type, id, pod, bytes);   // motors / limits: the POD array is re-sent before the next step, nothing else changes
}
// Many constraints of one type at once (a policy writing the motor targets of thousands of ragdolls per step).
MI_API int mi_constraints_update(mi_world* w, uint32_t type, uint32_t count, const uint32_t* ids, const void* pods, uint32_t podBytes) {
    if (!w || (count && (!ids || !pods))) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    for (uint32_t i = 0; i < count; ++i) {
        int rc = w->joints.update(type, ids[i], (const char*)pods + (size_t)i * podBytes, podBytes);
        if (rc != MI_OK) return rc;
    }
    return MI_OK;
}
MI_API int mi_constraint_destroy(mi_world* w, uint32_t type, uint32_t id) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    int rc = w->joints.destroy(type, id);
    if (rc == MI_OK) w->topologyDirty = true;
    return rc;
}
MI_API int mi_constraints_destroy_all(mi_world* w) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    w->joints.destroyAll(); w->topologyDirty = true;
    return MI_OK;
}
MI_API int mi_entity_destroy_constraints(mi_world* w, uint32_t entity) {
    if (!w || entity >= w->entities.size()) return fail(MI_ERR_INVALID_ARGUMENT, "bad entity");
    w->joints.destroyOfEntity(entity); w->topologyDirty = true;
    return MI_OK;
}
MI_API int mi_constraint_get(mi_world* w, uint32_t type, uint32_t id, void* pod, uint32_t bytes) {
    if (!w || !pod) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    return w->joints.get(type, id, pod, bytes);
}
MI_API int mi_constraint_create_from_global(mi_world* w, uint32_t type, uint32_t ea, uint32_t eb, const float* anchor, const float* axis,
                                            float l0, float l1, uint32_t* out) {
    if (!w || !anchor) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    int rc = w->download(); if (rc != MI_OK) return rc;
    rc = w->joints.addFromGlobal(*w, type, ea, eb, anchor, axis, l0, l1, out);
    if (rc == MI_OK) w->topologyDirty = true;
    return rc;
}

__global__ void k_add_forces(uint32_t n, const uint32_t* __restrict__ bodies, const float* __restrict__ ft, float4* __restrict__ bForce, float4* __restrict__ bTorque) {
    // one lane, in order: several entries may name the same body and the sums must not depend on scheduling
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t b = bodies[i];
        if (b == 0xFFFFFFFFu) continue;   // a ray that hit nothing
        float4 f = bForce[b], t = bTorque[b];
        f.x += ft[6 * i]; f.y += ft[6 * i + 1]; f.z += ft[6 * i + 2]; t.x += ft[6 * i + 3]; t.y += ft[6 * i + 4]; t.z += ft[6 * i + 5];
        bForce[b] = f; bTorque[b] = t;
    }
}
static int ensureUploaded(mi_world* w) {
    HIP_TRY(hipSetDevice(w->device));
    if (w->topologyDirty) { int rc = w->download(); if (rc != MI_OK) return rc; return w->upload(); }
    return MI_OK;
}
// rb.forceAccumulator += f; rb.torqueAccumulator += tau for many bodies.  While the host copy is authoritative (topology edits
// pending) the sums go there; otherwise they are added on the device without a download / re-upload.
MI_API int mi_entities_apply_forces(mi_world* w, uint32_t count, const uint32_t* ents, const float* forces3, const float* torques3) {
    if (!w || (count && !ents)) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    for (uint32_t i = 0; i < count; ++i)
        if (ents[i] >= w->entities.size() || w->entities[ents[i]].rb < 0) return fail(MI_ERR_INVALID_ARGUMENT, "not a rigid body");
    if (!count) return MI_OK;
    if (w->topologyDirty) {
        int rc = w->download(); if (rc != MI_OK) return rc;
        for (uint32_t i = 0; i < count; ++i) {
            HBody& b = w->bodies[w->entities[ents[i]].rb];
            if (forces3) b.force = b.force + V3(forces3[3 * i], forces3[3 * i + 1], forces3[3 * i + 2]);
            if (torques3) b.torque = b.torque + V3(torques3[3 * i], torques3[3 * i + 1], torques3[3 * i + 2]);
        }
        return MI_OK;
    }
    HIP_TRY(hipSetDevice(w->device));
    std::vector<uint32_t> ids(count); std::vector<float> ft(6 * (size_t)count, 0.f);
    for (uint32_t i = 0; i < count; ++i) {
        ids[i] = (uint32_t)w->entities[ents[i]].rb;
        for (int k = 0; k < 3; ++k) { if (forces3) ft[6 * i + k] = forces3[3 * i + k]; if (torques3) ft[6 * i + 3 + k] = torques3[3 * i + k]; }
    }
    DBuf<uint32_t> dIds; DBuf<float> dFt;
    HIP_TRY(dIds.ensure(count)); HIP_TRY(dFt.ensure(6 * (size_t)count));
    HIP_TRY(hipMemcpyAsync(dIds.p, ids.data(), count * sizeof(uint32_t), hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipMemcpyAsync(dFt.p, ft.data(), ft.size() * sizeof(float), hipMemcpyHostToDevice, w->stream));
    k_add_forces<<<1, 1, 0, w->stream>>>(count, dIds.p, dFt.p, w->bForce.p, w->bTorque.p);
    w->shard.prevValid = false;
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->hostStale = true;
    return MI_OK;
}
MI_API int mi_entity_apply_force(mi_world* w, uint32_t entity, const float* f, const float* t) { return mi_entities_apply_forces(w, 1, &entity, f, t); }
// testPhysicsInteraction(scene, ray, strength) (src/physics/physics.cpp:555-629) for `count` rays, applied in order.
MI_API int mi_world_test_interactions(mi_world* w, uint32_t count, const float* origins, const float* directions, const float* strengths, const uint32_t* ranges) {
    if (!w || (count && (!origins || !directions))) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (!count) return MI_OK;
    int rc = ensureUploaded(w); if (rc != MI_OK) return rc;
    const uint32_t nc = (uint32_t)w->colliders.size();
    if (!nc || w->bodies.empty()) return MI_OK;
    std::vector<float> rays(8 * (size_t)count, 0.f); std::vector<uint32_t> rg(2 * (size_t)count);
    for (uint32_t r = 0; r < count; ++r) {
        for (int k = 0; k < 3; ++k) { rays[8 * r + k] = origins[3 * r + k]; rays[8 * r + 3 + k] = directions[3 * r + k]; }
        rays[8 * r + 6] = strengths ? strengths[r] : 1000.f;
        rg[2 * r] = ranges ? ranges[2 * r] : 0u; rg[2 * r + 1] = ranges ? ranges[2 * r + 1] : 0xFFFFFFFFu;
    }
    DBuf<float> dRays, dFT; DBuf<uint32_t> dRanges, dBody;
    HIP_TRY(dRays.ensure(rays.size())); HIP_TRY(dRanges.ensure(rg.size())); HIP_TRY(dBody.ensure(count)); HIP_TRY(dFT.ensure(6 * (size_t)count));
    HIP_TRY(hipMemcpyAsync(dRays.p, rays.data(), rays.size() * sizeof(float), hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipMemcpyAsync(dRanges.p, rg.data(), rg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, w->stream));
    HullFaces hf{w->hullVerts.p, w->hullRanges.p, w->hullTris.p, w->hullTriRanges.p};
    k_ray_interactions<<<count, 256, 0, w->stream>>>(nc, dRays.p, dRanges.p, w->cTypeBody.p, w->cEntity.p, w->cShape.p, w->bPos.p, w->bRot.p, w->bCogInvMass.p, hf, dBody.p, dFT.p);
    k_add_forces<<<1, 1, 0, w->stream>>>(count, dBody.p, dFT.p, w->bForce.p, w->bTorque.p);
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->hostStale = true;
    return MI_OK;
}

MI_API int mi_world_step_fixed(mi_world* w, const mi_step_settings* s, float dt, uint32_t n) {
    if (!w || !s) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (n > 1 && w->shard.enabled && !w->shard.rccl) return fail(MI_ERR_INVALID_ARGUMENT, "a sharded world with the caller's transport takes one internal step per call (exchange in between)");
    if (n) w->transformsFollowPhysics = true;
    for (uint32_t i = 0; i < n; ++i) { if (i + 1 == n) w->posesArm(false, 0.f); int rc = w->stepInternal(*s, dt); if (rc != MI_OK) { w->poseArm = mi_world::PoseArm{}; return rc; } }
    return n ? w->posesAfterStep() : MI_OK;
}

// One internal step with a HIP event pair around every k_contact_solve launch (roofline measurement; bench.py).
// out: number of profiled launches, their summed kernel time (ms), and the contact updates (contacts x iterations) they performed.
MI_API int mi_world_step_profiled(mi_world* w, const mi_step_settings* s, float dt, uint32_t* out_launches, float* out_kernel_ms, uint64_t* out_contact_updates) {
    if (!w || !s) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    w->profileSolve = true; w->profLaunches = 0; w->profSlots = 0; w->profKernelMs = 0.f; w->profContacts = 0;
    int rc = w->stepInternal(*s, dt);
    w->profileSolve = false;
    if (out_launches) *out_launches = w->profLaunches;
    if (out_kernel_ms) *out_kernel_ms = w->profKernelMs;
    if (out_contact_updates) *out_contact_updates = w->profContacts;
    return rc;
}

// physicsStep (src/physics/physics.cpp:1364-1413): accumulator, <= maxPhysicsIterationsPerFrame sub-steps, pose interpolation.
MI_API int mi_world_step(mi_world* w, const mi_step_settings* s, float dt) {
    if (!w || !s) return fail(MI_ERR_INVALID_ARGUMENT, "null argument");
    if (w->shard.enabled && !w->shard.rccl && s->fixed_frame_rate && s->max_physics_iterations_per_frame > 1u)
        return fail(MI_ERR_INVALID_ARGUMENT, "a sharded world with the caller's transport takes ONE internal step per call (the exchange lies in between): max_physics_iterations_per_frame = 1, or mi_world_step_fixed");
    if (w->transformsFollowPhysics) { int rc = w->download(); if (rc != MI_OK) return rc; w->transformsFollowPhysics = false; }   // settle what mi_world_step_fixed left pending
    if (s->fixed_frame_rate) {
        const float fixedDt = 1.f / (float)s->frame_rate;
        w->timer += dt;
        uint32_t iterations = 0;
        if (w->timer >= fixedDt) {
            // physics_transform0 = physics_transform1 (physics.cpp:1380-1384), on the device
            HIP_TRY(hipSetDevice(w->device));
            if (w->topologyDirty) { int rc = w->download(); if (rc != MI_OK) return rc; rc = w->upload(); if (rc != MI_OK) return rc; w->haveEstimates = false; }
            const uint32_t nb = (uint32_t)w->bodies.size();
            if (nb) {
                HIP_TRY(w->bPos0.ensure(nb)); HIP_TRY(w->bRot0.ensure(nb));
                HIP_TRY(hipMemcpyAsync(w->bPos0.p, w->bPos.p, (size_t)nb * sizeof(float4), hipMemcpyDeviceToDevice, w->stream));
                HIP_TRY(hipMemcpyAsync(w->bRot0.p, w->bRot.p, (size_t)nb * sizeof(float4), hipMemcpyDeviceToDevice, w->stream));
                w->p0OnDevice = true; w->hostStale = true;
            }
            uint32_t willRun = 0; float timerAfter = w->timer;   // the loop below, run ahead: how many internal steps, and the interpolation factor they leave
            { uint32_t it = 0; while (timerAfter >= fixedDt && it++ < s->max_physics_iterations_per_frame) { timerAfter -= fixedDt; ++willRun; } if (timerAfter >= fixedDt) timerAfter = fmodf(timerAfter, fixedDt); }
            while (w->timer >= fixedDt && iterations++ < s->max_physics_iterations_per_frame) {
                if (iterations == willRun && nb) w->posesArm(true, timerAfter / fixedDt);
                int rc = w->stepInternal(*s, fixedDt); if (rc != MI_OK) { w->poseArm = mi_world::PoseArm{}; return rc; }
                w->timer -= fixedDt;
            }
        }
        if (w->timer >= fixedDt) w->timer = fmodf(w->timer, fixedDt);
        // the interpolated transforms lerp(transform0, transform1, timer / fixedDt) are produced when somebody asks for them (download)
        w->lerpT = w->timer / fixedDt;
        if (w->hostStale) { w->lerpPending = true; return w->posesAfterStep(); }
        const float t = w->lerpT;   // nothing newer on the device (no sub-step in this call, host state current): interpolate right here
        for (HBody& b : w->bodies) {
            HEntity& e = w->entities[b.entity];
            e.pos = lerp(b.p0, b.p1, t);
            e.rot = normalize(Q4(b.r0.x + t * (b.r1.x - b.r0.x), b.r0.y + t * (b.r1.y - b.r0.y), b.r0.z + t * (b.r1.z - b.r0.z), b.r0.w + t * (b.r1.w - b.r0.w)));
        }
        return MI_OK;
    }
    w->posesArm(false, 0.f);
    int rc = w->stepInternal(*s, dt); if (rc != MI_OK) { w->poseArm = mi_world::PoseArm{}; return rc; }
    w->transformsFollowPhysics = true;   // transform = physics_transform1, at the next download
    return w->posesAfterStep();
}

// ================================================================================================ sharded world (include/mi_shard.h)
extern "C++" {
namespace {
// tile -> rank: tiles in ascending Morton code of (tx, tz); rank r simulates the r-th of them
uint32_t mortonCode(uint32_t x, uint32_t z) { uint32_t c = 0; for (uint32_t b = 0; b < 16; ++b) c |= ((x >> b) & 1u) << (2 * b) | ((z >> b) & 1u) << (2 * b + 1); return c; }
std::vector<uint32_t> tilesInRankOrder(uint32_t tx, uint32_t tz) {
    std::vector<uint32_t> t((size_t)tx * tz);
    for (uint32_t i = 0; i < t.size(); ++i) t[i] = i;
    std::sort(t.begin(), t.end(), [&](uint32_t a, uint32_t b) { uint32_t ca = mortonCode(a % tx, a / tx), cb = mortonCode(b % tx, b / tx); return ca != cb ? ca < cb : a < b; });
    return t;
}
struct Id128 { char bytes[128]; };   // ncclUniqueId
// RCCL, resolved at run time (the library has no link-time dependency on it; a process that already loaded librccl.so.1 — torch — shares it)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl() {
    static Rccl r; static bool tried = false;
    if (tried) return r.lib ? &r : nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (r.lib) break;
    }
    if (!r.lib) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
    if (!r.lib) return nullptr;
    auto sym = [&](const char* n) { return dlsym(r.lib, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId"); r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy"); r.Send = (decltype(r.Send))sym("ncclSend"); r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart"); r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd"); r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    if (!r.GetUniqueId || !r.CommInitRank || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) { r.lib = nullptr; return nullptr; }
    return &r;
}
constexpr int kNcclFloat32 = 7;   // ncclFloat32 (rccl.h)
constexpr int kNcclUint64 = 5, kNcclSum = 0;
}
}

// Pack the records every neighbour is owed (device), then — library transport — one RCCL group of sends / receives on the world's stream and
// the unpack kernels behind it; nothing is read back in between.  With the caller's transport the messages wait in sendBuf for mi_world_shard_export.
// An articulated island is owned / ghosted / ignored as ONE (its root = lowest body index decides): union-find over the joints' body pairs.
int mi_world::shardBuildRoots() {
    const uint32_t nb = (uint32_t)bodies.size();
    std::vector<uint32_t> parent(nb);
    for (uint32_t i = 0; i < nb; ++i) parent[i] = i;
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    auto link = [&](const auto& t) { for (const uint2& b : t.bodies) { uint32_t x = find(b.x), y = find(b.y); if (x != y) parent[std::max(x, y)] = std::min(x, y); } };
    link(joints.distance); link(joints.ball); link(joints.fixed); link(joints.hinge); link(joints.cone); link(joints.slider);
    for (uint32_t i = 0; i < nb; ++i) parent[i] = find(i);
    HIP_TRY(shard.root.ensure(std::max(nb, 1u))); HIP_TRY(shard.active.ensure(std::max(nb, 1u)));
    if (nb) HIP_TRY(hipMemcpy(shard.root.p, parent.data(), (size_t)nb * sizeof(uint32_t), hipMemcpyHostToDevice));
    shard.rootJoints = joints.count(); shard.rootBodies = nb;
    return MI_OK;
}
void mi_world::shardReleaseComm() { if (shard.comm) { if (Rccl* r = rccl()) if (r->CommDestroy) (void)r->CommDestroy(shard.comm); shard.comm = nullptr; } }
// A neighbour message that did not fit is an error, never a silent loss: the packed record counts of the last exchange are checked as soon as
// they are on the host — right after the exchange with the caller's transport (it synchronises anyway), and with the library transport at the
// next exchange or whenever the caller looks at the world in between (counts, owned entities, exchange statistics, checkpoint, detach).
int mi_world::shardCheckOverflow(bool sync) {
    ShardState& sh = shard;
    if (!sh.sentPending) return MI_OK;
    if (sync) HIP_TRY(hipStreamSynchronize(stream));
    sh.sentPending = false;
    if (sh.exchangeTimed) { sh.exchangeMsSum += (double)elapsedMs(sh.exEv[0], sh.exEv[1]); ++sh.exchangesTimed; sh.exchangeTimed = false; }
    for (uint32_t k = 0; k < sh.sp.numPeers; ++k) { sh.sentLast[k] = sh.sentHost[k]; sh.sentSum[k] += sh.sentHost[k]; }
    if (sh.rccl && sh.recvHost) {   // what the neighbours sent in that exchange (the headers of their messages)
        for (uint32_t k = 0; k < sh.sp.numPeers; ++k) {
            if (sh.recvHost[k] == 0xFFFFFFFFu) return fail(MI_ERR_CAPACITY, "a neighbour message outgrew the size both ranks had derived from the previous exchange (more than 1.5 x + 512 records in one step): set MI_SHARD_ADAPTIVE=0 on all ranks for fixed-size messages");
            sh.recvLast[k] = sh.recvHost[k];
        }
        sh.recvValid = true;
    }
    for (uint32_t k = 0; k < sh.sp.numPeers; ++k) if (sh.sentHost[k] > sh.capacity) return fail(MI_ERR_CAPACITY, "shard message overflow: raise mi_shard_desc::max_records (equal on all ranks)");
    return MI_OK;
}
int mi_world::shardExchange() {
    const uint32_t nb = (uint32_t)bodies.size(), nc = (uint32_t)colliders.size();
    if (!nb) return MI_OK;
    ShardState& sh = shard;
    { int rc = shardCheckOverflow(false); if (rc != MI_OK) return rc; }   // (the previous exchange's counts have long arrived: the end-of-step read-back came after them)
    hipStream_t st = stream;
    StepScalars* sc = scalarsPtr();
    ShardBufs sendBufs{}, recvBufs{};
    for (uint32_t k = 0; k < sh.sp.numPeers; ++k) { sendBufs.p[k] = sh.sendBuf[k].p; recvBufs.p[k] = sh.recvBuf[k].p; }
    if (!sh.exEv[0]) { HIP_TRY(hipEventCreate(&sh.exEv[0])); HIP_TRY(hipEventCreate(&sh.exEv[1])); }
    HIP_TRY(hipEventRecord(sh.exEv[0], st));
    // after a valid step the buffer sets are swapped: bPos = the new state, bPosN = the state the step started from; the record counts were cleared by k_reset_scalars
    k_shard_pack<<<divUp(nb, 256), 256, 0, st>>>(nb, sh.sp, sh.bordersPending ? sh.spNext : sh.sp, sh.bordersPending ? 1u : 0u, sh.known.p, sh.active.p, bPos.p, bRot.p, bLinVel.p, bAngVel.p, bPosN.p, bRotN.p, bCogInvMass.p, sendBufs, sh.capacity, sc, sh.root.p);
    k_shard_pack_headers<<<1, 64, 0, st>>>(sh.sp.numPeers, sc, sendBufs, nc, sh.rccl ? nullptr : sh.axisDev.p);   // caller's transport: until the caller hands in the centre statistics
                                                                                                                    // summed over all ranks (mi_world_shard_set_axis_sums), the next step's sweep axis is the one of this rank's own sums
    HIP_TRY(hipMemcpyAsync(sh.sentHost, &sc->shardSent[0], 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    sh.sentPending = true; sh.exchangeTimed = true;
    if (sh.bordersPending) { sh.sp = sh.spNext; sh.bordersX = sh.nextX; sh.bordersZ = sh.nextZ; sh.bordersPending = false; sh.fullExchanges = 2; sh.sweepFullSteps = 2; }   // the next step classifies with the new borders (this exchange hands bodies over: full-size messages, on every rank)
    if (!sh.rccl) {
        // (the axis of this rank's own sums is also what k_pair_finish computed: hs.axisNext, already in sapAxis)
        HIP_TRY(hipEventRecord(sh.exEv[1], st));
        HIP_TRY(hipStreamSynchronize(st));     // the messages are complete when this returns
        sh.axisHostCurrent = true;
        return shardCheckOverflow(false);
    }
    Rccl* r = rccl();
    ShardCaps caps{}; uint32_t maxRecs = 0;
    const bool sized = sh.adaptive && sh.fullExchanges == 0u && sh.recvValid;
    for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t m = std::max(sh.sentLast[k], sh.recvLast[k]);
        caps.c[k] = sized && k < sh.sp.numPeers ? std::min(sh.capacity, m + m / 2u + 512u) : sh.capacity;
        if (k < sh.sp.numPeers) { sh.sizedLast[k] = caps.c[k]; maxRecs = std::max(maxRecs, caps.c[k]); sh.bytesSentSum += (uint64_t)(caps.c[k] + 1u) * kShardRecordFloats * sizeof(float); }
    }
    if (sh.fullExchanges) --sh.fullExchanges;
    int e = r->GroupStart(); if (e) return fail(MI_ERR_DEVICE, "ncclGroupStart failed");
    for (uint32_t k = 0; k < sh.sp.numPeers && !e; ++k) {
        const size_t n = (size_t)(caps.c[k] + 1u) * kShardRecordFloats;   // header + the records both ends expect at most
        e = r->Send(sh.sendBuf[k].p, n, kNcclFloat32, (int)sh.peerRanks[k], sh.comm, st);
        if (!e) e = r->Recv(sh.recvBuf[k].p, n, kNcclFloat32, (int)sh.peerRanks[k], sh.comm, st);
    }
    const int e2 = r->GroupEnd();
    if (e || e2) return fail(MI_ERR_DEVICE, r->GetErrorString ? r->GetErrorString(e ? e : e2) : "RCCL send / receive failed");
    if (sh.sp.numPeers) {
        k_shard_unpack<<<dim3(divUp(std::max(maxRecs, 1u), 256), sh.sp.numPeers), 256, 0, st>>>(nb, recvBufs, sh.capacity, bPos.p, bRot.p, bLinVel.p, bAngVel.p, sh.known.p, caps, &sc->shardRecv[0]);
        if (!sh.recvHost) { HIP_TRY(hipHostMalloc((void**)&sh.recvHost, 8 * sizeof(uint32_t))); std::memset(sh.recvHost, 0, 8 * sizeof(uint32_t)); }
        HIP_TRY(hipMemcpyAsync(sh.recvHost, &sc->shardRecv[0], 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));   // (read at the next exchange, like the sent counts)
    }
    // global sweep axis: the centre statistics of the colliders every rank owns, summed over all ranks (72 bytes), stay on the device
    if (r->AllReduce) {
        const int e3 = r->AllReduce(sc->axisSums, sh.axisGlobal.p, kAxisSums, kNcclUint64, kNcclSum, sh.comm, st);
        if (e3) return fail(MI_ERR_DEVICE, r->GetErrorString ? r->GetErrorString(e3) : "ncclAllReduce failed");
        k_shard_axis<<<1, 64, 0, st>>>(sh.axisGlobal.p, nc, sh.axisDev.p);
    } else k_shard_axis<<<1, 64, 0, st>>>(sc->axisSums, nc, sh.axisDev.p);
    sh.axisHostCurrent = false;
    HIP_TRY(hipEventRecord(sh.exEv[1], st));
    hostStale = true;
    return MI_OK;
}
// Exact seam: the hand-over after sweep `sweep` (include/mi_shard.h).  The velocities of the bodies this rank owns and a neighbour holds as ghosts are gathered
// into one fixed-size message per neighbour; library transport: sent / received / scattered on the world's stream; caller's transport: the stream is
// drained and the caller's function moves mi_world_shard_export_sweep -> mi_world_shard_import_sweep.
int mi_world::shardSweepExchange(uint32_t sweep) {
    ShardState& sh = shard;
    hipStream_t st = stream;
    const uint32_t nb = (uint32_t)bodies.size();
    ShardBufs send{}, recv{}; SweepLists lists{};
    for (uint32_t k = 0; k < sh.sp.numPeers; ++k) {
        HIP_TRY(sh.sweepSend[k].ensure(sh.sweepFloats())); HIP_TRY(sh.sweepRecv[k].ensure(sh.sweepFloats()));
        send.p[k] = sh.sweepSend[k].p; recv.p[k] = sh.sweepRecv[k].p; lists.p[k] = sh.sweepList[k].p;
    }
    if (sh.sp.numPeers) k_seam_sweep_pack<<<dim3(divUp(sh.capacity, 256), sh.sp.numPeers), 256, 0, st>>>(lists, sh.sweepCount.p, sh.capacity, gVel.p, send);
    ++sh.sweepExchanges;
    if (sh.sweepsDone++ == 0u && sh.sp.numPeers) {   // once per step: did the lists fit?  (the step is synchronous in this mode anyway)
        uint32_t* counts = sh.sweepCounts;
        HIP_TRY(hipMemcpyAsync(counts, sh.sweepCount.p, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint32_t k = 0; k < sh.sp.numPeers; ++k) if (counts[k] > sh.capacity) return fail(MI_ERR_CAPACITY, "exact seam: more shared bodies than a neighbour message holds (mi_shard_desc::max_records)");
        if (sh.sweepCut) { sh.sweepCut = false; return fail(MI_ERR_CAPACITY, "exact seam: a sweep message outgrew the size both ranks had derived from the previous step (more than 1.5 x + 64 shared bodies in one step): set MI_SHARD_ADAPTIVE=0 on all ranks"); }
        if (sh.rccl) {   // this step's message sizes: both ends know both list lengths of the PREVIOUS step (their own, and the header of the last message they received)
            const bool sized = sh.adaptive && sh.sweepFullSteps == 0u && sh.sweepRecvValid;
            if (sh.sweepRecvValid) { for (uint32_t k = 0; k < sh.sp.numPeers; ++k) HIP_TRY(hipMemcpyAsync(&sh.sweepPeerHdr[k], sh.sweepRecv[k].p, sizeof(uint32_t), hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); }
            for (uint32_t k = 0; k < sh.sp.numPeers; ++k) {
                const uint32_t m = std::max(sh.sweepPrevOwn[k], sh.sweepPeerHdr[k]);
                sh.sweepSized[k] = sized ? std::min(sh.capacity, m + m / 2u + 64u) : sh.capacity;
                if (counts[k] > sh.sweepSized[k] || (sized && sh.sweepPeerHdr[k] > sh.capacity)) sh.sweepCut = true;   // (sent cut short all the same — the neighbour is already waiting for exactly that many floats —, reported at the next step)
                sh.sweepPrevOwn[k] = counts[k];
            }
            if (sh.sweepFullSteps) --sh.sweepFullSteps;
        }
    }
    if (sh.rccl) {
        Rccl* r = rccl();
        int e = r->GroupStart(); if (e) return fail(MI_ERR_DEVICE, "ncclGroupStart failed");
        ShardCaps caps{};
        for (uint32_t k = 0; k < 8u; ++k) caps.c[k] = k < sh.sp.numPeers ? sh.sweepSized[k] : sh.capacity;
        for (uint32_t k = 0; k < sh.sp.numPeers && !e; ++k) {
            const size_t n = (size_t)(sh.sweepSized[k] + 1u) * kSweepRecordFloats;   // header + the records both ends expect at most
            e = r->Send(sh.sweepSend[k].p, n, kNcclFloat32, (int)sh.peerRanks[k], sh.comm, st);
            if (!e) e = r->Recv(sh.sweepRecv[k].p, n, kNcclFloat32, (int)sh.peerRanks[k], sh.comm, st);
        }
        const int e2 = r->GroupEnd();
        if (e || e2) return fail(MI_ERR_DEVICE, r->GetErrorString ? r->GetErrorString(e ? e : e2) : "RCCL send / receive failed");
        if (sh.sp.numPeers) k_seam_sweep_unpack<<<dim3(divUp(sh.capacity, 256), sh.sp.numPeers), 256, 0, st>>>(nb, recv, sh.capacity, sh.active.p, gVel.p, caps);
        sh.sweepRecvValid = true;
        return MI_OK;
    }
    HIP_TRY(hipStreamSynchronize(st));   // the messages are complete
    if (!sh.sweepFn) return fail(MI_ERR_INVALID_ARGUMENT, "exact seam: neither the library transport is attached nor a sweep exchange callback set — the seam would silently run as block Jacobi");
    { const int rc = sh.sweepFn(sh.sweepUser, this, sweep); if (rc != MI_OK) return fail(MI_ERR_DEVICE, "exact seam: the caller's sweep exchange failed"); }
    return MI_OK;
}
MI_API int mi_world_set_seam_tiling(mi_world* w, const mi_shard_desc* d) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    HIP_TRY(hipSetDevice(w->device));
    w->tabValid = false; w->haveEstimates = false;   // the colour ranges mean something else from here on
    if (!d) { w->seamTiling.on = false; return MI_OK; }
    if (w->shard.enabled) return fail(MI_ERR_UNSUPPORTED, "a sharded world takes its tiling from mi_world_shard_enable (mi_world_shard_set_exact_seam)");
    if (!d->tiles_x || !d->tiles_z || !(d->tile_size_x > 0.f) || !(d->tile_size_z > 0.f) || !(d->ghost_margin > 0.f) || 2.f * d->ghost_margin > d->tile_size_x || 2.f * d->ghost_margin > d->tile_size_z)
        return fail(MI_ERR_INVALID_ARGUMENT, "tiles must be at least two ghost margins wide");
    mi_world::SeamTiling& t = w->seamTiling;
    t.bx.clear(); t.bz.clear(); t.margin = d->ghost_margin;
    for (uint32_t i = 1; i < d->tiles_x; ++i) t.bx.push_back((float)((double)d->origin_x + (double)i * (double)d->tile_size_x));
    for (uint32_t i = 1; i < d->tiles_z; ++i) t.bz.push_back((float)((double)d->origin_z + (double)i * (double)d->tile_size_z));
    HIP_TRY(t.dBx.ensure(std::max<size_t>(t.bx.size(), 1))); HIP_TRY(t.dBz.ensure(std::max<size_t>(t.bz.size(), 1)));
    if (!t.bx.empty()) HIP_TRY(hipMemcpy(t.dBx.p, t.bx.data(), t.bx.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!t.bz.empty()) HIP_TRY(hipMemcpy(t.dBz.p, t.bz.data(), t.bz.size() * sizeof(float), hipMemcpyHostToDevice));
    t.on = true;
    if (!w->topologyDirty) { int rc = w->shardBuildRoots(); if (rc != MI_OK) return rc; }   // (an upload builds them otherwise)
    return MI_OK;
}
MI_API int mi_world_shard_set_exact_seam(mi_world* w, uint32_t enable, mi_shard_sweep_fn fn, void* user) {
    if (!w || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    mi_world::ShardState& sh = w->shard;
    if (enable && sh.desc.tiles_x > 1u && sh.desc.tiles_z > 1u) return fail(MI_ERR_UNSUPPORTED, "exact seam: x- or z-slabs only (at a corner a shared body is seen by four tiles)");
    if (enable && (2.f * sh.desc.ghost_margin > sh.desc.tile_size_x || 2.f * sh.desc.ghost_margin > sh.desc.tile_size_z)) return fail(MI_ERR_INVALID_ARGUMENT, "exact seam: tiles must be at least two ghost margins wide");
    if (enable && !fn && !sh.rccl) return fail(MI_ERR_INVALID_ARGUMENT, "exact seam: with the caller's transport a sweep exchange callback is needed");
    if (enable) {   // ... also the tiles a load balance has cut (borders in force and pending): a body must never lie within the margin of two borders
        const float m2 = 2.f * sh.desc.ghost_margin;
        for (const std::vector<float>* b : {&sh.bordersX, &sh.bordersZ, &sh.nextX, &sh.nextZ})
            for (size_t i = 1; i < b->size(); ++i) if (!((*b)[i] - (*b)[i - 1] > m2)) return fail(MI_ERR_INVALID_ARGUMENT, "exact seam: a tile of the current borders is narrower than two ghost margins");
    }
    HIP_TRY(hipSetDevice(w->device));
    if (sh.exact != (enable != 0u)) { w->tabValid = false; w->haveEstimates = false; }   // the colour ranges mean something else from here on
    sh.exact = enable != 0u; sh.sweepFn = fn; sh.sweepUser = user; sh.sweepFullSteps = 2; sh.sweepRecvValid = false;
    if (sh.exact) { HIP_TRY(sh.sweepImport.ensure(sh.sweepFloats())); HIP_TRY(w->seamId.ensure(std::max<size_t>(w->bodies.size(), 1))); }
    return MI_OK;
}
MI_API int mi_world_shard_sweep_message_bytes(mi_world* w, uint64_t* out) {
    if (!w || !out || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    *out = (uint64_t)w->shard.sweepFloats() * sizeof(float); return MI_OK;
}
MI_API int mi_world_shard_export_sweep(mi_world* w, uint32_t slot, void* out) {
    if (!w || !out || !w->shard.enabled || !w->shard.exact || slot >= w->shard.sp.numPeers || !w->shard.sweepSend[slot].p) return fail(MI_ERR_INVALID_ARGUMENT, "no sweep message for this slot");
    HIP_TRY(hipSetDevice(w->device));
    // (only what the message holds: its header and `count` records; the rest of the caller's buffer is not touched)
    HIP_TRY(hipMemcpyAsync(out, w->shard.sweepSend[slot].p, (size_t)(std::min(w->shard.sweepCounts[slot], w->shard.capacity) + 1u) * kSweepRecordFloats * sizeof(float), hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    return MI_OK;
}
MI_API int mi_world_shard_import_sweep(mi_world* w, const void* msg) {
    if (!w || !msg || !w->shard.enabled || !w->shard.exact) return fail(MI_ERR_INVALID_ARGUMENT, "not an exact-seam world");
    if (w->shard.rccl) return fail(MI_ERR_INVALID_ARGUMENT, "the library transport exchanges the sweeps itself");
    mi_world::ShardState& sh = w->shard;
    uint32_t count; std::memcpy(&count, msg, 4);
    if (count > sh.capacity) return fail(MI_ERR_CAPACITY, "sweep message holds more records than max_records");
    HIP_TRY(hipSetDevice(w->device));
    HIP_TRY(sh.sweepImport.ensure(sh.sweepFloats()));
    HIP_TRY(hipMemcpyAsync(sh.sweepImport.p, msg, (size_t)(count + 1u) * kSweepRecordFloats * sizeof(float), hipMemcpyHostToDevice, w->stream));
    ShardBufs in{}; in.p[0] = sh.sweepImport.p;
    if (count) k_seam_sweep_unpack<<<dim3(divUp(count, 256), 1), 256, 0, w->stream>>>((uint32_t)w->bodies.size(), in, sh.capacity, sh.active.p, w->gVel.p);
    HIP_TRY(hipStreamSynchronize(w->stream));   // the staging buffer is free again
    return MI_OK;
}
MI_API int mi_world_seam_stats(mi_world* w, uint32_t* manifolds, uint32_t* colors, uint32_t* violations) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    if (manifolds) *manifolds = w->seamLast[0];
    if (colors) *colors = w->seamLast[1];
    if (violations) *violations = (uint32_t)std::min<uint64_t>(w->seamViolations, 0xFFFFFFFFull);
    return MI_OK;
}
// sapAxis (host) <- the device word, when a library-transport exchange has moved it on
int mi_world::shardSyncAxis() {
    if (!shard.enabled || shard.axisHostCurrent || !shard.axisDev.p) return MI_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpyAsync(&sapAxis, shard.axisDev.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    shard.axisHostCurrent = true;
    return MI_OK;
}

MI_API int mi_shard_tile_of_rank(uint32_t tx, uint32_t tz, uint32_t rank, uint32_t* out) {
    if (!out || !tx || !tz || tx > 65535u || tz > 65535u || rank >= tx * tz) return fail(MI_ERR_INVALID_ARGUMENT, "bad tile grid / rank");
    *out = tilesInRankOrder(tx, tz)[rank]; return MI_OK;
}
MI_API int mi_shard_rank_of_tile(uint32_t tx, uint32_t tz, uint32_t tile, uint32_t* out) {
    if (!out || !tx || !tz || tx > 65535u || tz > 65535u || tile >= tx * tz) return fail(MI_ERR_INVALID_ARGUMENT, "bad tile grid / tile");
    const std::vector<uint32_t> t = tilesInRankOrder(tx, tz);
    *out = (uint32_t)(std::find(t.begin(), t.end(), tile) - t.begin()); return MI_OK;
}
MI_API int mi_world_shard_enable(mi_world* w, const mi_shard_desc* d) {
    if (!w || !d) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    if (!d->tiles_x || !d->tiles_z || d->tiles_x > 65535u || d->tiles_z > 65535u || d->num_ranks != d->tiles_x * d->tiles_z || d->rank >= d->num_ranks) return fail(MI_ERR_INVALID_ARGUMENT, "num_ranks must equal tiles_x * tiles_z");
    if (!(d->tile_size_x > 0.f) || !(d->tile_size_z > 0.f) || !(d->ghost_margin > 0.f) || d->ghost_margin >= d->tile_size_x || d->ghost_margin >= d->tile_size_z) return fail(MI_ERR_INVALID_ARGUMENT, "0 < ghost_margin < tile size");
    HIP_TRY(hipSetDevice(w->device));
    mi_world::ShardState& sh = w->shard;
    if (w->seamTiling.on) { w->seamTiling.on = false; w->tabValid = false; w->haveEstimates = false; }   // a told tiling (mi_world_set_seam_tiling) ends here: a sharded world takes its seam colouring from mi_world_shard_set_exact_seam only
    sh.desc = *d;
    const std::vector<uint32_t> order = tilesInRankOrder(d->tiles_x, d->tiles_z);
    ShardParams& sp = sh.sp;
    sp.margin = d->ghost_margin;
    sp.tilesX = d->tiles_x; sp.tilesZ = d->tiles_z; sp.myTile = order[d->rank]; sp.numPeers = 0; sh.peerRanks.clear();
    sh.bordersX.clear(); sh.bordersZ.clear(); sh.bordersPending = false; sh.flagsOfAStep = false;
    for (uint32_t i = 1; i < d->tiles_x; ++i) sh.bordersX.push_back((float)((double)d->origin_x + (double)i * (double)d->tile_size_x));
    for (uint32_t i = 1; i < d->tiles_z; ++i) sh.bordersZ.push_back((float)((double)d->origin_z + (double)i * (double)d->tile_size_z));
    w->shardFillBorders(sp, sh.bordersX, sh.bordersZ);
    const int mx = (int)(sp.myTile % sp.tilesX), mz = (int)(sp.myTile / sp.tilesX);
    for (int z = mz - 1; z <= mz + 1; ++z) for (int x = mx - 1; x <= mx + 1; ++x) {          // ascending tile index
        if ((x == mx && z == mz) || x < 0 || z < 0 || x >= (int)sp.tilesX || z >= (int)sp.tilesZ) continue;
        const uint32_t t = (uint32_t)z * sp.tilesX + (uint32_t)x;
        sp.peers[sp.numPeers++] = t;
        sh.peerRanks.push_back((uint32_t)(std::find(order.begin(), order.end(), t) - order.begin()));
    }
    const uint32_t nb = (uint32_t)w->bodies.size();
    sh.capacity = d->max_records ? d->max_records : std::max(4096u, nb / d->num_ranks / 4u);   // a message always travels whole: (capacity + 1) records of 56 bytes
    { int rc = w->shardBuildRoots(); if (rc != MI_OK) return rc; }
    for (HBody& b : w->bodies) b.shardKnown = 1;                     // every rank was given the same scene
    HIP_TRY(sh.known.ensure(std::max(nb, 1u))); HIP_TRY(hipMemset(sh.known.p, 1, std::max(nb, 1u)));
    for (uint32_t k = 0; k < sp.numPeers; ++k) {
        HIP_TRY(sh.sendBuf[k].ensure(sh.messageFloats())); HIP_TRY(sh.recvBuf[k].ensure(sh.messageFloats()));
        HIP_TRY(hipMemset(sh.sendBuf[k].p, 0, sh.messageFloats() * sizeof(float))); HIP_TRY(hipMemset(sh.recvBuf[k].p, 0, sh.messageFloats() * sizeof(float)));
    }
    if (!sh.sentHost) HIP_TRY(hipHostMalloc((void**)&sh.sentHost, 8 * sizeof(uint32_t)));
    std::memset(sh.sentHost, 0, 8 * sizeof(uint32_t)); sh.sentPending = false; sh.exchangeTimed = false;
    sh.fullExchanges = 2; sh.recvValid = false; sh.sweepFullSteps = 2; sh.sweepRecvValid = false; if (const char* ad = getenv("MI_SHARD_ADAPTIVE")) sh.adaptive = ad[0] != '0';
    HIP_TRY(sh.axisDev.ensure(1)); HIP_TRY(sh.axisGlobal.ensure(kAxisSums)); HIP_TRY(sh.importBuf.ensure(sh.messageFloats()));
    HIP_TRY(hipMemcpy(sh.axisDev.p, &w->sapAxis, sizeof(uint32_t), hipMemcpyHostToDevice)); sh.axisHostCurrent = true;
    sh.enabled = true;
    w->haveEstimates = false;   // the first sharded step sizes itself exactly
    return MI_OK;
}
MI_API int mi_world_shard_neighbours(mi_world* w, uint32_t* out, uint32_t* count) {
    if (!w || !count || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    *count = w->shard.sp.numPeers;
    if (out) for (uint32_t k = 0; k < w->shard.sp.numPeers; ++k) out[k] = w->shard.peerRanks[k];
    return MI_OK;
}
MI_API int mi_world_shard_counts(mi_world* w, uint32_t* bodies, uint32_t* manifolds, uint32_t* contacts) {
    if (!w || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    if (bodies) *bodies = w->shard.owned[0];
    if (manifolds) *manifolds = w->shard.owned[1];
    if (contacts) *contacts = w->shard.owned[2];
    return w->shardCheckOverflow(true);   // (an overflow in the LAST exchange of a run is reported here at the latest)
}
// ---- load balance: the tile borders follow the bodies
extern "C++" {
namespace {
// What ONE change of the borders may do.  A body's new owner, and every rank that newly holds it as a ghost, must be the old owner's tile or one of
// its neighbours (only those exchange messages): new border i stays within [old border i-1 + margin, old border i+1 - margin]; and a tile stays
// wider than the margin (its ghost region must not reach past its neighbours).
bool shardBordersValid(const std::vector<float>& cur, const float* nb, uint32_t n, float m, float minWidth) {   // minWidth: m, or 2 m under the exact seam (a body within the margin of TWO borders has no seam class)
    for (uint32_t i = 0; i < n; ++i) {
        if (!(nb[i] == nb[i])) return false;
        if (i > 0 && !(nb[i] - nb[i - 1] > minWidth)) return false;
        if (i > 0 && nb[i] < cur[i - 1] + m) return false;
        if (i + 1 < n && nb[i] > cur[i + 1] - m) return false;
    }
    return true;
}
}
}
void mi_world::shardFillBorders(ShardParams& sp, const std::vector<float>& bx, const std::vector<float>& bz) const {
    const float inf = std::numeric_limits<float>::infinity();
    auto lower = [&](const std::vector<float>& b, int tile) { return tile <= 0 ? -inf : tile > (int)b.size() ? inf : b[(size_t)tile - 1]; };   // lower border of `tile`
    const int mx = (int)(sp.myTile % sp.tilesX), mz = (int)(sp.myTile / sp.tilesX);
    for (int k = 0; k < 4; ++k) { sp.bx[k] = lower(bx, mx - 1 + k); sp.bz[k] = lower(bz, mz - 1 + k); }
}
MI_API int mi_world_shard_set_borders(mi_world* w, const float* bx, const float* bz) {
    if (!w || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    mi_world::ShardState& sh = w->shard;
    const float m = sh.desc.ghost_margin;
    const float minWidth = sh.exact ? 2.f * m : m;
    if ((bx && !shardBordersValid(sh.bordersX, bx, (uint32_t)sh.bordersX.size(), m, minWidth)) || (bz && !shardBordersValid(sh.bordersZ, bz, (uint32_t)sh.bordersZ.size(), m, minWidth)))
        return fail(MI_ERR_INVALID_ARGUMENT, "borders: ascending, tiles wider than ghost_margin (two margins under the exact seam), and border i within [old border i-1 + margin, old border i+1 - margin]");
    sh.nextX = bx ? std::vector<float>(bx, bx + sh.bordersX.size()) : sh.bordersX;
    sh.nextZ = bz ? std::vector<float>(bz, bz + sh.bordersZ.size()) : sh.bordersZ;
    sh.spNext = sh.sp; w->shardFillBorders(sh.spNext, sh.nextX, sh.nextZ);
    sh.bordersPending = true;
    return MI_OK;
}
MI_API int mi_world_shard_get_borders(mi_world* w, float* bx, float* bz) {
    if (!w || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    if (bx) std::copy(w->shard.bordersX.begin(), w->shard.bordersX.end(), bx);
    if (bz) std::copy(w->shard.bordersZ.begin(), w->shard.bordersZ.end(), bz);
    return MI_OK;
}
MI_API int mi_world_shard_histogram(mi_world* w, uint32_t axis, float lo, float hi, uint32_t bins, uint32_t* out) {
    if (!w || !out || !w->shard.enabled || axis > 1u || !bins || !(hi > lo)) return fail(MI_ERR_INVALID_ARGUMENT, "axis 0 | 1, bins > 0, lo < hi");
    HIP_TRY(hipSetDevice(w->device));
    const uint32_t nb = (uint32_t)w->bodies.size();
    std::fill(out, out + bins, 0u);
    if (!nb || w->topologyDirty || !w->shard.flagsOfAStep) return MI_OK;   // no step since the scene was (re)built: nothing is owned yet
    HIP_TRY(w->shard.hist.ensure(bins));
    HIP_TRY(hipMemsetAsync(w->shard.hist.p, 0, bins * sizeof(uint32_t), w->stream));
    k_shard_histogram<<<divUp(nb, 256), 256, 0, w->stream>>>(nb, axis, lo, (float)bins / (hi - lo), bins, w->shard.active.p, w->bPos.p, w->bRot.p, w->bCogInvMass.p, w->shard.root.p, w->shard.hist.p);
    HIP_TRY(hipMemcpyAsync(out, w->shard.hist.p, bins * sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    return MI_OK;
}
// Borders that even out the body counts: hist = bodies per bin of [lo, hi) along one axis, summed over all ranks.  Border i goes where the cumulative
// count reaches i / tiles of the total (linear inside a bin), clamped to what one change may do; pure arithmetic, the same on every rank.
MI_API int mi_shard_balance_borders(const uint64_t* hist, uint32_t bins, float lo, float hi, uint32_t tiles, const float* cur, float margin, float* out) {
    if (!hist || !bins || !(hi > lo) || !tiles || (tiles > 1 && (!cur || !out))) return fail(MI_ERR_INVALID_ARGUMENT, "null / empty");
    const uint32_t n = tiles - 1u;
    if (!n) return MI_OK;
    const std::vector<float> c(cur, cur + n);
    std::vector<float> nb(c);
    double total = 0; for (uint32_t b = 0; b < bins; ++b) total += (double)hist[b];
    if (total > 0) {
        const double width = ((double)hi - (double)lo) / (double)bins;
        uint32_t b = 0; double below = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const double target = total * (double)(i + 1u) / (double)tiles;
            while (b + 1u < bins && below + (double)hist[b] < target) { below += (double)hist[b]; ++b; }
            const double frac = hist[b] ? std::min(1.0, std::max(0.0, (target - below) / (double)hist[b])) : 0.5;
            double v = (double)lo + ((double)b + frac) * width;
            if (i > 0) v = std::max(v, std::max((double)c[i - 1] + (double)margin, (double)nb[i - 1] + 1.25 * (double)margin));   // (x 1.25: room to move next time)
            if (i + 1u < n) v = std::min(v, (double)c[i + 1] - (double)margin);
            // ... and move by at most two margins: the hand-over rides in ONE neighbour message, whose capacity is sized in margin strips
            v = std::min(std::max(v, (double)c[i] - 2.0 * (double)margin), (double)c[i] + 2.0 * (double)margin);
            if (i > 0) v = std::max(v, (double)nb[i - 1] + 1.25 * (double)margin);
            nb[i] = (float)v;
        }
    }
    const bool ok = shardBordersValid(c, nb.data(), n, margin, margin);
    for (uint32_t i = 0; i < n; ++i) out[i] = ok ? nb[i] : c[i];
    return MI_OK;
}
// Sum of n 64-bit counters over all ranks: ONE ncclAllReduce on the world's stream (global counts; the histograms of the load balance)
MI_API int mi_world_shard_allreduce_u64(mi_world* w, uint64_t* inout, uint32_t n) {
    if (!w || !w->shard.enabled || (n && !inout)) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world / null");
    if (!w->shard.rccl) return fail(MI_ERR_UNSUPPORTED, "the library's all-reduce needs the library transport (mi_world_shard_attach_rccl); with the caller's transport reduce the values yourself");
    Rccl* r = rccl(); if (!r || !r->AllReduce) return fail(MI_ERR_UNSUPPORTED, "ncclAllReduce not found");
    if (!n) return MI_OK;
    HIP_TRY(hipSetDevice(w->device));
    HIP_TRY(w->shard.reduceBuf.ensure(n));
    HIP_TRY(hipMemcpyAsync(w->shard.reduceBuf.p, inout, n * sizeof(uint64_t), hipMemcpyHostToDevice, w->stream));
    const int e = r->AllReduce(w->shard.reduceBuf.p, w->shard.reduceBuf.p, n, kNcclUint64, kNcclSum, w->shard.comm, w->stream);
    if (e) return fail(MI_ERR_DEVICE, r->GetErrorString ? r->GetErrorString(e) : "ncclAllReduce failed");
    HIP_TRY(hipMemcpyAsync(inout, w->shard.reduceBuf.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    return MI_OK;
}
// One rebalancing round in one call (library transport): histograms of both axes over the extent of the tile grid as enabled, all-reduced, balanced, set.
MI_API int mi_world_shard_rebalance(mi_world* w, uint32_t bins) {
    if (!w || !w->shard.enabled || !bins || bins > 65536u) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world / bins");
    if (!w->shard.rccl) return fail(MI_ERR_UNSUPPORTED, "mi_world_shard_rebalance needs the library transport; with the caller's transport: mi_world_shard_histogram, your all-reduce, mi_shard_balance_borders, mi_world_shard_set_borders");
    const mi_shard_desc& d = w->shard.desc;
    const float lo[2] = {d.origin_x, d.origin_z};
    const float hi[2] = {(float)((double)d.origin_x + (double)d.tiles_x * (double)d.tile_size_x), (float)((double)d.origin_z + (double)d.tiles_z * (double)d.tile_size_z)};
    std::vector<uint32_t> h32(bins); std::vector<uint64_t> h((size_t)2 * bins);
    for (uint32_t a = 0; a < 2; ++a) {
        int rc = mi_world_shard_histogram(w, a, lo[a], hi[a], bins, h32.data()); if (rc != MI_OK) return rc;
        for (uint32_t b = 0; b < bins; ++b) h[(size_t)a * bins + b] = h32[b];
    }
    int rc = mi_world_shard_allreduce_u64(w, h.data(), 2u * bins); if (rc != MI_OK) return rc;
    std::vector<float> nx(w->shard.bordersX), nz(w->shard.bordersZ);
    rc = mi_shard_balance_borders(h.data(), bins, lo[0], hi[0], d.tiles_x, w->shard.bordersX.data(), d.ghost_margin, nx.data()); if (rc != MI_OK) return rc;
    rc = mi_shard_balance_borders(h.data() + bins, bins, lo[1], hi[1], d.tiles_z, w->shard.bordersZ.data(), d.ghost_margin, nz.data()); if (rc != MI_OK) return rc;
    if (w->shard.exact) {   // exact seam: a tile stays wider than TWO margins (a body within the margin of two borders has no seam class); a proposal that would not is not taken
        const float m2 = 2.f * d.ghost_margin;
        auto wide = [&](const std::vector<float>& b) { for (size_t i = 1; i < b.size(); ++i) if (!(b[i] - b[i - 1] > m2)) return false; return true; };
        if (!wide(nx)) nx = w->shard.bordersX;
        if (!wide(nz)) nz = w->shard.bordersZ;
    }
    return mi_world_shard_set_borders(w, nx.empty() ? nullptr : nx.data(), nz.empty() ? nullptr : nz.data());
}
MI_API int mi_world_shard_owned_entities(mi_world* w, uint32_t* out, uint32_t cap, uint32_t* count) {
    if (!w || !count || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    HIP_TRY(hipSetDevice(w->device));
    const uint32_t nb = (uint32_t)w->bodies.size();
    std::vector<uint8_t> act(nb, 0);
    HIP_TRY(hipStreamSynchronize(w->stream));
    { int rc = w->shardCheckOverflow(false); if (rc != MI_OK) return rc; }
    if (nb && w->shard.flagsOfAStep && !w->topologyDirty) HIP_TRY(hipMemcpy(act.data(), w->shard.active.p, nb, hipMemcpyDeviceToHost));   // (no step since the scene was (re)built: nothing is owned yet)
    uint32_t n = 0;
    for (uint32_t b = 0; b < nb; ++b) if (act[b] == 1u) { if (out && n < cap) out[n] = w->bodies[b].entity; ++n; }
    *count = n;
    return (out && n > cap) ? fail(MI_ERR_CAPACITY, "capacity < owned bodies") : MI_OK;
}
MI_API int mi_shard_get_unique_id(void* out) {
    if (!out) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    Rccl* r = rccl(); if (!r) return fail(MI_ERR_UNSUPPORTED, "librccl.so.1 not found");
    return r->GetUniqueId(out) == 0 ? MI_OK : fail(MI_ERR_DEVICE, "ncclGetUniqueId failed");
}
MI_API int mi_shard_library_transport_available(void) { Rccl* r = rccl(); return r && r->AllReduce && r->CommDestroy ? 1 : 0; }
MI_API int mi_world_shard_attach_rccl(mi_world* w, const void* id) {
    if (!w || !id || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "enable sharding first");
    Rccl* r = rccl(); if (!r) return fail(MI_ERR_UNSUPPORTED, "librccl.so.1 not found");
    HIP_TRY(hipSetDevice(w->device));
    Id128 uid; std::memcpy(uid.bytes, id, sizeof(uid.bytes));
    const int e = r->CommInitRank(&w->shard.comm, (int)w->shard.desc.num_ranks, uid, (int)w->shard.desc.rank);
    if (e) return fail(MI_ERR_DEVICE, r->GetErrorString ? r->GetErrorString(e) : "ncclCommInitRank failed");
    w->shard.rccl = true; w->shard.fullExchanges = 2; w->shard.recvValid = false; w->shard.sweepFullSteps = 2; w->shard.sweepRecvValid = false;
    return MI_OK;
}
// Development / tests: the library transport on ONE rank.  A one-rank communicator whose every neighbour is this rank itself: the exchange then runs
// ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, the unpack kernels and the all-reduce with real records on one GPU — each message comes back to its sender.
// (What a rank receives are the records it packed for the neighbouring tile: states of bodies it holds anyway, so the world goes on exactly like one on the
// caller's transport that is handed its own messages back: tests/test_gpu_sharding.py.)
MI_API int mi_debug_shard_attach_loopback(mi_world* w) {
    if (!w || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "enable sharding first");
    Rccl* r = rccl(); if (!r) return fail(MI_ERR_UNSUPPORTED, "librccl.so.1 not found");
    HIP_TRY(hipSetDevice(w->device));
    Id128 uid; std::memset(&uid, 0, sizeof(uid));
    int e = r->GetUniqueId(&uid);
    if (!e) e = r->CommInitRank(&w->shard.comm, 1, uid, 0);
    if (e) return fail(MI_ERR_DEVICE, r->GetErrorString ? r->GetErrorString(e) : "ncclCommInitRank failed");
    for (uint32_t& p : w->shard.peerRanks) p = 0u;
    w->shard.rccl = true; w->shard.fullExchanges = 2; w->shard.recvValid = false; w->shard.sweepFullSteps = 2; w->shard.sweepRecvValid = false;
    return MI_OK;
}
// ... and the message last RECEIVED in slot `slot` (library transport), so a test can hold it against what was sent
MI_API int mi_debug_shard_peek_received(mi_world* w, uint32_t slot, uint32_t sweep_message, void* out) {
    if (!w || !out || !w->shard.enabled || slot >= w->shard.sp.numPeers) return fail(MI_ERR_INVALID_ARGUMENT, "bad slot");
    HIP_TRY(hipSetDevice(w->device));
    HIP_TRY(hipStreamSynchronize(w->stream));
    const DBuf<float>& b = sweep_message ? w->shard.sweepRecv[slot] : w->shard.recvBuf[slot];
    const size_t n = sweep_message ? w->shard.sweepFloats() : w->shard.messageFloats();
    if (!b.p || b.cap < n) return fail(MI_ERR_INVALID_ARGUMENT, "nothing received in this slot yet");
    HIP_TRY(hipMemcpy(out, b.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return MI_OK;
}
MI_API int mi_world_shard_detach_rccl(mi_world* w) {
    if (!w || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    HIP_TRY(hipSetDevice(w->device));
    HIP_TRY(hipStreamSynchronize(w->stream));
    if (w->shard.exact && !w->shard.sweepFn) return fail(MI_ERR_INVALID_ARGUMENT, "exact seam: detaching the library transport needs a sweep exchange callback (mi_world_shard_set_exact_seam) first");
    w->shardReleaseComm(); w->shard.rccl = false;
    { int rc = w->shardSyncAxis(); if (rc != MI_OK) return rc; }
    return w->shardCheckOverflow(false);
}
MI_API int mi_world_shard_message_bytes(mi_world* w, uint64_t* out) {
    if (!w || !out || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    *out = (uint64_t)w->shard.messageFloats() * sizeof(float); return MI_OK;
}
MI_API int mi_world_shard_export(mi_world* w, uint32_t slot, void* out) {
    if (!w || !out || !w->shard.enabled || slot >= w->shard.sp.numPeers) return fail(MI_ERR_INVALID_ARGUMENT, "bad slot");
    HIP_TRY(hipSetDevice(w->device));
    HIP_TRY(hipMemcpy(out, w->shard.sendBuf[slot].p, w->shard.messageFloats() * sizeof(float), hipMemcpyDeviceToHost));
    return MI_OK;
}
MI_API int mi_world_shard_import(mi_world* w, const void* msg) {
    if (!w || !msg || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    if (w->shard.rccl) return fail(MI_ERR_INVALID_ARGUMENT, "mi_world_shard_import belongs to the caller's transport; this world exchanges through the library's (mi_world_shard_detach_rccl first)");
    HIP_TRY(hipSetDevice(w->device));
    uint32_t count; std::memcpy(&count, msg, 4);
    if (count > w->shard.capacity) return fail(MI_ERR_CAPACITY, "shard message overflow: raise mi_shard_desc::max_records (equal on all ranks)");
    const uint32_t nb = (uint32_t)w->bodies.size();
    HIP_TRY(w->shard.importBuf.ensure(w->shard.messageFloats()));   // its own staging: a tile without neighbours (1 x 1 grid) has no receive buffer
    HIP_TRY(hipMemcpyAsync(w->shard.importBuf.p, msg, (size_t)(count + 1u) * kShardRecordFloats * sizeof(float), hipMemcpyHostToDevice, w->stream));
    ShardBufs one{}; one.p[0] = w->shard.importBuf.p;
    if (count) k_shard_unpack<<<dim3(divUp(count, 256), 1), 256, 0, w->stream>>>(nb, one, w->shard.capacity, w->bPos.p, w->bRot.p, w->bLinVel.p, w->bAngVel.p, w->shard.known.p);
    HIP_TRY(hipStreamSynchronize(w->stream));     // `msg` is the caller's (possibly pageable) memory
    w->hostStale = true;
    return MI_OK;
}

// Global sweep axis with the caller's transport: this rank's centre statistics of the last internal step (the colliders of the bodies it owned;
// rank 0 also the colliders without a rigid body) — add them over all ranks and hand the sums to every rank before its next step.
MI_API int mi_world_shard_axis_sums(mi_world* w, uint64_t* out9) {
    if (!w || !out9 || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    for (uint32_t c = 0; c < kAxisSums; ++c) out9[c] = w->hs.axisSums[c];
    return MI_OK;
}
MI_API int mi_world_shard_set_axis_sums(mi_world* w, const uint64_t* global9) {
    if (!w || !global9 || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    if (w->shard.rccl) return fail(MI_ERR_INVALID_ARGUMENT, "with the library transport the exchange sums the statistics itself (ncclAllReduce)");
    HIP_TRY(hipSetDevice(w->device));
    unsigned long long s9[kAxisSums]; for (uint32_t c = 0; c < kAxisSums; ++c) s9[c] = global9[c];
    w->sapAxis = axisFromSums(s9, (uint32_t)w->colliders.size());
    HIP_TRY(hipMemcpyAsync(w->shard.axisDev.p, &w->sapAxis, sizeof(uint32_t), hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->shard.axisHostCurrent = true;
    return MI_OK;
}
// What the exchanges cost and moved (bench.py's N > 1 line): device time between the pack kernel and the end of the unpack / axis kernels.
MI_API int mi_world_shard_exchange_stats(mi_world* w, mi_shard_exchange_stats* out, uint32_t reset) {
    if (!w || !out || !w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "not a sharded world");
    HIP_TRY(hipSetDevice(w->device));
    mi_world::ShardState& sh = w->shard;
    int rc = w->shardCheckOverflow(true);
    std::memset(out, 0, sizeof(*out));
    out->exchanges = sh.exchangesTimed; out->device_ms_sum = sh.exchangeMsSum; out->num_neighbours = sh.sp.numPeers;
    out->message_bytes = (uint64_t)sh.messageFloats() * sizeof(float); out->library_transport = sh.rccl ? 1u : 0u;
    for (uint32_t k = 0; k < 8u; ++k) out->message_records_last[k] = k < sh.sp.numPeers ? (sh.rccl ? sh.sizedLast[k] : sh.capacity) : 0u;
    out->message_bytes_sum = sh.bytesSentSum;
    for (uint32_t k = 0; k < sh.sp.numPeers; ++k) { out->neighbour_rank[k] = sh.peerRanks[k]; out->records_last[k] = sh.sentLast[k]; out->records_sum[k] = sh.sentSum[k]; }
    const uint32_t nb = (uint32_t)w->bodies.size();
    if (nb && sh.flagsOfAStep && !w->topologyDirty) {
        std::vector<uint8_t> act(nb);
        HIP_TRY(hipMemcpy(act.data(), sh.active.p, nb, hipMemcpyDeviceToHost));
        for (uint8_t a : act) { out->owned_bodies += a == 1u; out->ghost_bodies += a == 2u; }
    }
    out->sweep_exchanges = sh.sweepExchanges; out->sweep_message_bytes = sh.exact ? (uint64_t)((sh.rccl && sh.sp.numPeers ? sh.sweepSized[0] : sh.capacity) + 1u) * kSweepRecordFloats * sizeof(float) : 0ull;
    for (uint32_t k = 0; k < sh.sp.numPeers; ++k) out->sweep_records_last[k] = sh.exact ? sh.sweepCounts[k] : 0u;
    if (reset) { sh.exchangesTimed = 0; sh.exchangeMsSum = 0.0; for (uint64_t& v : sh.sentSum) v = 0; sh.sweepExchanges = 0; sh.bytesSentSum = 0; }
    return rc;
}

// ---- checkpoint / resume (the solver-relevant state of a world)
// Everything a bit-identical continuation needs that is not the scene description itself: body states (physics_transform1,
// velocities, accumulators), physics_transform0 + the interpolated entity transforms, the step accumulator, the SAP axis chosen
// for the next step, the colour history (pair -> colour, which is also the previous step's collision list of the events), the
// previous step's trigger overlaps and the constraint PODs (motors / limits may have been edited).  The blob is tied to the
// topology: it can only be loaded into a world built from the same scene (same bodies, colliders, constraints).
extern "C++" {
namespace {
struct CheckpointHeader { uint32_t magic, version, numEntities, numBodies, numColliders, numHistory, numTriggerOverlaps, sapAxis; float timer; uint32_t eventsEnabled, jointCounts[6], reserved; };
constexpr uint32_t kCheckpointMagic = 0x4350494Du;   // "MIPC"
// CheckpointHeader::reserved bit 0: a SHARD SECTION follows the cloths — the blob is ONE RANK's view of a sharded world (include/mi_shard.h): which of
// its body copies are current (a rank only trusts a copy it owned in the last step or got a record for), the tile borders in force and the pending
// ones of a load-balance round.  Without it a restore to an earlier step would classify with the flags and borders of the LATER moment: bodies that
// migrated in between would be owned by nobody (or by two ranks) and silently drop out.  Such a blob only loads into the same rank of the same grid.
constexpr uint32_t kCheckpointHasShard = 1u;
struct CheckpointShard { uint32_t numRanks, rank, tilesX, tilesZ, bordersPending, knownBytes; };
template <class T> void put(std::vector<uint8_t>& out, const T* p, size_t n) { const uint8_t* b = reinterpret_cast<const uint8_t*>(p); out.insert(out.end(), b, b + n * sizeof(T)); }
template <class T> bool take(const uint8_t*& p, const uint8_t* end, T* out, size_t n) { if ((size_t)(end - p) < n * sizeof(T)) return false; std::memcpy(out, p, n * sizeof(T)); p += n * sizeof(T); return true; }
template <class JT> void putPods(std::vector<uint8_t>& out, const JT& j) { if (!j.pods.empty()) put(out, j.pods.data(), j.pods.size()); }
template <class JT> bool takePods(const uint8_t*& p, const uint8_t* end, JT& j) { return j.pods.empty() || take(p, end, j.pods.data(), j.pods.size()); }
}
}
MI_API int mi_world_save_checkpoint(mi_world* w, void* out, uint64_t capacity, uint64_t* out_size) {
    if (!w || !out_size) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    HIP_TRY(hipSetDevice(w->device));
    int rc = w->download(); if (rc != MI_OK) return rc;
    if (w->shard.enabled) { rc = w->shardCheckOverflow(true); if (rc != MI_OK) return rc; rc = w->shardSyncAxis(); if (rc != MI_OK) return rc; }
    std::vector<unsigned long long> keys; std::vector<uint32_t> vals;
    if (w->tabValid) {   // also with a pending topology edit: the keys are creation indices, the live world keeps the history across it
        const size_t cap = (size_t)w->tabMask[w->tabCur] + 1;
        std::vector<HistSlot> t(cap);
        HIP_TRY(hipMemcpy(t.data(), w->tab[w->tabCur].p, cap * sizeof(HistSlot), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cap; ++i) if (t[i].key) { keys.push_back(t[i].key); vals.push_back((uint32_t)t[i].val); }
    }
    CheckpointHeader h{};
    h.magic = kCheckpointMagic; h.version = 1; h.numEntities = (uint32_t)w->entities.size(); h.numBodies = (uint32_t)w->bodies.size();
    h.numColliders = (uint32_t)w->colliders.size(); h.numHistory = (uint32_t)keys.size(); h.numTriggerOverlaps = (uint32_t)w->prevTriggerOverlaps.size();
    h.sapAxis = w->sapAxis; h.timer = w->timer; h.eventsEnabled = w->eventsEnabled ? 1u : 0u;
    h.reserved = w->shard.enabled ? kCheckpointHasShard : 0u;
    const JointSet& j = w->joints;
    h.jointCounts[0] = (uint32_t)j.distance.pods.size(); h.jointCounts[1] = (uint32_t)j.ball.pods.size(); h.jointCounts[2] = (uint32_t)j.fixed.pods.size();
    h.jointCounts[3] = (uint32_t)j.hinge.pods.size(); h.jointCounts[4] = (uint32_t)j.cone.pods.size(); h.jointCounts[5] = (uint32_t)j.slider.pods.size();
    std::vector<uint8_t> blob;
    put(blob, &h, 1);
    for (const HEntity& e : w->entities) { put(blob, &e.pos, 1); put(blob, &e.rot, 1); }
    for (const HBody& b : w->bodies) { put(blob, &b.p0, 1); put(blob, &b.r0, 1); put(blob, &b.p1, 1); put(blob, &b.r1, 1); put(blob, &b.linVel, 1); put(blob, &b.angVel, 1); put(blob, &b.force, 1); put(blob, &b.torque, 1); }
    if (!keys.empty()) { put(blob, keys.data(), keys.size()); put(blob, vals.data(), vals.size()); }
    if (!w->prevTriggerOverlaps.empty()) put(blob, w->prevTriggerOverlaps.data(), w->prevTriggerOverlaps.size());
    putPods(blob, j.distance); putPods(blob, j.ball); putPods(blob, j.fixed); putPods(blob, j.hinge); putPods(blob, j.cone); putPods(blob, j.slider);
    {   // cloths: particle state (positions incl. inverse mass, previous positions, velocities, force accumulators) and the editable properties
        const uint32_t numCloths = (uint32_t)w->cloths.size();
        put(blob, &numCloths, 1);
        HIP_TRY(hipStreamSynchronize(w->stream));
        for (mi_world::HCloth* c : w->cloths) {
            const uint32_t n = c->desc.grid_size_x * c->desc.grid_size_y;
            put(blob, &c->desc, 1); put(blob, &c->oldTotalMass, 1); put(blob, &c->oldStiffness, 1);
            std::vector<float4> buf(4 * (size_t)n);
            HIP_TRY(hipMemcpy(buf.data(), c->pos.p, n * sizeof(float4), hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(buf.data() + n, c->prev.p, n * sizeof(float4), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(buf.data() + 2 * (size_t)n, c->vel.p, n * sizeof(float4), hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(buf.data() + 3 * (size_t)n, c->force.p, n * sizeof(float4), hipMemcpyDeviceToHost));
            put(blob, buf.data(), buf.size());
            put(blob, c->restInvMass.data(), c->restInvMass.size());
        }
    }
    if (w->shard.enabled) {   // this rank's view: current copies, borders (download() has mirrored the device's `known` flags into the host bodies)
        const mi_world::ShardState& sh = w->shard;
        const uint32_t nb = (uint32_t)w->bodies.size();
        CheckpointShard cs{sh.desc.num_ranks, sh.desc.rank, sh.desc.tiles_x, sh.desc.tiles_z, sh.bordersPending ? 1u : 0u, (nb + 3u) & ~3u};
        put(blob, &cs, 1);
        std::vector<uint8_t> known(cs.knownBytes, 0); for (uint32_t i = 0; i < nb; ++i) known[i] = w->bodies[i].shardKnown;
        put(blob, known.data(), known.size());
        const std::vector<float>& nx = sh.bordersPending ? sh.nextX : sh.bordersX; const std::vector<float>& nz = sh.bordersPending ? sh.nextZ : sh.bordersZ;
        if (!sh.bordersX.empty()) { put(blob, sh.bordersX.data(), sh.bordersX.size()); put(blob, nx.data(), nx.size()); }
        if (!sh.bordersZ.empty()) { put(blob, sh.bordersZ.data(), sh.bordersZ.size()); put(blob, nz.data(), nz.size()); }
    }
    *out_size = blob.size();
    if (!out) return MI_OK;
    if (capacity < blob.size()) return fail(MI_ERR_CAPACITY, "capacity < checkpoint size");
    std::memcpy(out, blob.data(), blob.size());
    return MI_OK;
}
MI_API int mi_world_load_checkpoint(mi_world* w, const void* data, uint64_t size) {
    if (!w || !data) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    HIP_TRY(hipSetDevice(w->device));
    // Transactional: the blob is validated against the scene and its own header BEFORE anything is allocated from its counts,
    // parsed into temporaries, and only committed to the world once it has been consumed completely.  Nothing throws across the ABI.
    try {
        const uint8_t* p = static_cast<const uint8_t*>(data); const uint8_t* end = p + size;
        CheckpointHeader h;
        if (!take(p, end, &h, 1) || h.magic != kCheckpointMagic || h.version != 1) return fail(MI_ERR_INVALID_ARGUMENT, "not a checkpoint of this library version");
        JointSet& j = w->joints;
        const uint32_t jc[6] = {(uint32_t)j.distance.pods.size(), (uint32_t)j.ball.pods.size(), (uint32_t)j.fixed.pods.size(), (uint32_t)j.hinge.pods.size(), (uint32_t)j.cone.pods.size(), (uint32_t)j.slider.pods.size()};
        if (h.numEntities != w->entities.size() || h.numBodies != w->bodies.size() || h.numColliders != w->colliders.size() || std::memcmp(jc, h.jointCounts, sizeof(jc)) != 0)
            return fail(MI_ERR_INVALID_ARGUMENT, "checkpoint belongs to a different scene (entity / body / collider / constraint counts differ)");
        // exact size the header implies (64-bit arithmetic; the cloth section is checked against the world's own cloths)
        const uint64_t entityBytes = sizeof(w->entities[0].pos) + sizeof(w->entities[0].rot);
        const uint64_t bodyBytes = sizeof(HBody::p0) + sizeof(HBody::r0) + sizeof(HBody::p1) + sizeof(HBody::r1) + sizeof(HBody::linVel) + sizeof(HBody::angVel) + sizeof(HBody::force) + sizeof(HBody::torque);
        auto podBytes = [](const auto& t) -> uint64_t { return t.pods.empty() ? 0ull : (uint64_t)t.pods.size() * sizeof(t.pods[0]); };
        uint64_t expect = sizeof(CheckpointHeader) + (uint64_t)h.numEntities * entityBytes + (uint64_t)h.numBodies * bodyBytes
                        + (uint64_t)h.numHistory * (sizeof(unsigned long long) + sizeof(uint32_t)) + (uint64_t)h.numTriggerOverlaps * sizeof(w->prevTriggerOverlaps[0])
                        + podBytes(j.distance) + podBytes(j.ball) + podBytes(j.fixed) + podBytes(j.hinge) + podBytes(j.cone) + podBytes(j.slider) + sizeof(uint32_t);
        for (const mi_world::HCloth* c : w->cloths) {
            const uint64_t n = (uint64_t)c->desc.grid_size_x * c->desc.grid_size_y;
            expect += sizeof(mi_cloth_desc) + 2 * sizeof(float) + 4 * n * sizeof(float4) + (uint64_t)c->restInvMass.size() * sizeof(c->restInvMass[0]);
        }
        const bool hasShard = (h.reserved & kCheckpointHasShard) != 0u;
        if (h.reserved & ~kCheckpointHasShard) return fail(MI_ERR_INVALID_ARGUMENT, "not a checkpoint of this library version");
        if (hasShard) {
            if (!w->shard.enabled) return fail(MI_ERR_INVALID_ARGUMENT, "the checkpoint is one rank's view of a sharded world: enable sharding (same grid, same rank) before loading it");
            expect += sizeof(CheckpointShard) + (((uint64_t)h.numBodies + 3u) & ~3ull) + 2ull * sizeof(float) * (w->shard.bordersX.size() + w->shard.bordersZ.size());
        }
        if (expect != size) return fail(MI_ERR_INVALID_ARGUMENT, "truncated or oversized checkpoint (size does not match its header and this scene)");
        // ---- parse into temporaries
        struct EntityState { decltype(HEntity::pos) pos; decltype(HEntity::rot) rot; };
        struct BodyState { decltype(HBody::p0) p0; decltype(HBody::r0) r0; decltype(HBody::p1) p1; decltype(HBody::r1) r1; decltype(HBody::linVel) linVel, angVel, force, torque; };
        std::vector<EntityState> es(h.numEntities); std::vector<BodyState> bs(h.numBodies);
        bool okay = true;
        for (EntityState& e : es) okay = okay && take(p, end, &e.pos, 1) && take(p, end, &e.rot, 1);
        for (BodyState& b : bs) okay = okay && take(p, end, &b.p0, 1) && take(p, end, &b.r0, 1) && take(p, end, &b.p1, 1) && take(p, end, &b.r1, 1) && take(p, end, &b.linVel, 1) && take(p, end, &b.angVel, 1) && take(p, end, &b.force, 1) && take(p, end, &b.torque, 1);
        std::vector<unsigned long long> keys(h.numHistory); std::vector<uint32_t> vals(h.numHistory);
        okay = okay && take(p, end, keys.data(), keys.size()) && take(p, end, vals.data(), vals.size());
        auto overlaps = w->prevTriggerOverlaps; overlaps.resize(h.numTriggerOverlaps);
        okay = okay && take(p, end, overlaps.data(), overlaps.size());
        auto pDistance = j.distance.pods; auto pBall = j.ball.pods; auto pFixed = j.fixed.pods; auto pHinge = j.hinge.pods; auto pCone = j.cone.pods; auto pSlider = j.slider.pods;
        auto takeVec = [&](auto& v) { return v.empty() || take(p, end, v.data(), v.size()); };
        okay = okay && takeVec(pDistance) && takeVec(pBall) && takeVec(pFixed) && takeVec(pHinge) && takeVec(pCone) && takeVec(pSlider);
        uint32_t numCloths = 0;
        okay = okay && take(p, end, &numCloths, 1);
        if (!okay) return fail(MI_ERR_INVALID_ARGUMENT, "truncated checkpoint");
        if (numCloths != w->cloths.size()) return fail(MI_ERR_INVALID_ARGUMENT, "checkpoint belongs to a different scene (cloth count differs)");
        struct ClothState { mi_cloth_desc d; float oldMass, oldStiff; std::vector<float4> buf; std::vector<float2> rest; };
        std::vector<ClothState> cs(w->cloths.size());
        for (size_t k = 0; k < cs.size(); ++k) {
            const mi_world::HCloth* c = w->cloths[k]; ClothState& t = cs[k];
            okay = take(p, end, &t.d, 1) && take(p, end, &t.oldMass, 1) && take(p, end, &t.oldStiff, 1);
            if (!okay) return fail(MI_ERR_INVALID_ARGUMENT, "truncated checkpoint");
            if (t.d.grid_size_x != c->desc.grid_size_x || t.d.grid_size_y != c->desc.grid_size_y) return fail(MI_ERR_INVALID_ARGUMENT, "checkpoint belongs to a different scene (cloth grid differs)");
            const size_t n = (size_t)c->desc.grid_size_x * c->desc.grid_size_y;
            t.buf.resize(4 * n); t.rest.resize(c->restInvMass.size());
            if (!take(p, end, t.buf.data(), t.buf.size()) || !take(p, end, t.rest.data(), t.rest.size())) return fail(MI_ERR_INVALID_ARGUMENT, "truncated checkpoint");
        }
        CheckpointShard shardHdr{}; std::vector<uint8_t> known; std::vector<float> curX, nextX, curZ, nextZ;
        if (hasShard) {
            const mi_world::ShardState& sh = w->shard;
            if (!take(p, end, &shardHdr, 1)) return fail(MI_ERR_INVALID_ARGUMENT, "truncated checkpoint");
            if (shardHdr.numRanks != sh.desc.num_ranks || shardHdr.rank != sh.desc.rank || shardHdr.tilesX != sh.desc.tiles_x || shardHdr.tilesZ != sh.desc.tiles_z || shardHdr.knownBytes != ((h.numBodies + 3u) & ~3u))
                return fail(MI_ERR_INVALID_ARGUMENT, "the checkpoint belongs to another rank or another tile grid of the sharded world");
            known.resize(shardHdr.knownBytes); curX.resize(sh.bordersX.size()); nextX.resize(sh.bordersX.size()); curZ.resize(sh.bordersZ.size()); nextZ.resize(sh.bordersZ.size());
            bool ok2 = take(p, end, known.data(), known.size());
            if (!curX.empty()) ok2 = ok2 && take(p, end, curX.data(), curX.size()) && take(p, end, nextX.data(), nextX.size());
            if (!curZ.empty()) ok2 = ok2 && take(p, end, curZ.data(), curZ.size()) && take(p, end, nextZ.data(), nextZ.size());
            if (!ok2) return fail(MI_ERR_INVALID_ARGUMENT, "truncated checkpoint");
            auto ascending = [](const std::vector<float>& b) { for (size_t i = 0; i < b.size(); ++i) { if (!(b[i] == b[i])) return false; if (i && !(b[i] > b[i - 1])) return false; } return true; };
            if (!ascending(curX) || !ascending(curZ) || !ascending(nextX) || !ascending(nextZ)) return fail(MI_ERR_INVALID_ARGUMENT, "corrupt checkpoint (tile borders)");
        }
        if (p != end) return fail(MI_ERR_INVALID_ARGUMENT, "oversized checkpoint");
        // colour history table: same open-addressing layout the kernels probe (tableSlot / linear probing)
        uint32_t cap = 1024; while ((uint64_t)cap < 2ull * h.numHistory) cap <<= 1;
        std::vector<unsigned long long> tk; std::vector<uint32_t> tv;
        if (h.numHistory) {
            tk.assign(cap, 0ull); tv.assign(cap, 0u);
            for (uint32_t i = 0; i < h.numHistory; ++i) {
                if (!keys[i]) return fail(MI_ERR_INVALID_ARGUMENT, "corrupt checkpoint (null history key)");
                uint32_t s_ = (uint32_t)((keys[i] * 0x9E3779B97F4A7C15ull) >> 40) & (cap - 1u);
                while (tk[s_]) s_ = (s_ + 1u) & (cap - 1u);
                tk[s_] = keys[i]; tv[s_] = vals[i];
            }
        }
        // ---- commit
        int rc = w->download(); if (rc != MI_OK) return rc;   // the host copy becomes authoritative; everything is re-sent before the next step
        if (h.numHistory) { const int c = w->tabCur; HIP_TRY(w->tab[c].ensure(cap)); }
        for (size_t i = 0; i < es.size(); ++i) { w->entities[i].pos = es[i].pos; w->entities[i].rot = es[i].rot; }
        for (size_t i = 0; i < bs.size(); ++i) { HBody& b = w->bodies[i]; b.p0 = bs[i].p0; b.r0 = bs[i].r0; b.p1 = bs[i].p1; b.r1 = bs[i].r1; b.linVel = bs[i].linVel; b.angVel = bs[i].angVel; b.force = bs[i].force; b.torque = bs[i].torque; }
        w->prevTriggerOverlaps = std::move(overlaps);
        j.distance.pods = std::move(pDistance); j.ball.pods = std::move(pBall); j.fixed.pods = std::move(pFixed); j.hinge.pods = std::move(pHinge); j.cone.pods = std::move(pCone); j.slider.pods = std::move(pSlider);
        for (size_t k = 0; k < cs.size(); ++k) {
            mi_world::HCloth* c = w->cloths[k]; ClothState& t = cs[k];
            const uint32_t n = c->desc.grid_size_x * c->desc.grid_size_y;
            c->desc = t.d; c->oldTotalMass = t.oldMass; c->oldStiffness = t.oldStiff; c->restInvMass.assign(t.rest.begin(), t.rest.end());
            for (uint32_t i = 0; i < n; ++i) c->invMasses[i] = t.buf[i].w;
            HIP_TRY(hipMemcpy(c->pos.p, t.buf.data(), n * sizeof(float4), hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(c->prev.p, t.buf.data() + n, n * sizeof(float4), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->vel.p, t.buf.data() + 2 * (size_t)n, n * sizeof(float4), hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(c->force.p, t.buf.data() + 3 * (size_t)n, n * sizeof(float4), hipMemcpyHostToDevice));
            c->constraintsDirty = true;
        }
        w->clothsDirty = true;
        w->timer = h.timer; w->sapAxis = h.sapAxis; w->eventsEnabled = h.eventsEnabled != 0; w->pendingEvents.clear();
        w->topologyDirty = true; w->haveEstimates = false;
        if (w->shard.enabled) {   // the rank's view of that moment (or, from a blob of an unsharded / fully synchronised world: every copy is current)
            mi_world::ShardState& sh = w->shard;
            for (size_t i = 0; i < w->bodies.size(); ++i) w->bodies[i].shardKnown = hasShard ? (known[i] ? 1 : 0) : 1;
            if (hasShard) {
                sh.bordersX = curX; sh.bordersZ = curZ; w->shardFillBorders(sh.sp, sh.bordersX, sh.bordersZ);
                sh.bordersPending = shardHdr.bordersPending != 0u;
                if (sh.bordersPending) { sh.nextX = nextX; sh.nextZ = nextZ; sh.spNext = sh.sp; w->shardFillBorders(sh.spNext, sh.nextX, sh.nextZ); }
            }
            sh.flagsOfAStep = false; sh.prevValid = false; sh.sentPending = false; sh.exchangeTimed = false;
            HIP_TRY(hipMemcpy(sh.axisDev.p, &w->sapAxis, sizeof(uint32_t), hipMemcpyHostToDevice)); sh.axisHostCurrent = true;
        }
        w->tabValid = h.numHistory != 0;
        if (w->tabValid) {
            const int c = w->tabCur; w->tabMask[c] = cap - 1u;
            std::vector<HistSlot> t(cap);
            for (uint32_t i = 0; i < cap; ++i) { t[i].key = tk[i]; t[i].val = tv[i]; }
            HIP_TRY(hipMemcpy(w->tab[c].p, t.data(), cap * sizeof(HistSlot), hipMemcpyHostToDevice));
            w->last.numManifolds = h.numHistory;
        }
        return MI_OK;
    } catch (const std::bad_alloc&) {
        return fail(MI_ERR_OUT_OF_MEMORY, "out of host memory while loading a checkpoint");
    } catch (...) {
        return fail(MI_ERR_INVALID_ARGUMENT, "checkpoint could not be loaded");
    }
}

MI_API int mi_world_enable_events(mi_world* w, uint32_t enable) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    w->eventsEnabled = enable != 0; w->pendingEvents.clear(); w->prevTriggerOverlaps.clear();
    w->tabValid = false;           // the event diff starts from an empty previous frame (so does the colour history, once)
    w->haveEstimates = false;
    return MI_OK;
}
MI_API int mi_world_poll_events(mi_world* w, mi_event* out, uint32_t cap, uint32_t* count) {
    if (!w || !count) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    *count = (uint32_t)w->pendingEvents.size();
    if (!out) return MI_OK;
    if (cap < w->pendingEvents.size()) return fail(MI_ERR_CAPACITY, "capacity < pending events");
    std::memcpy(out, w->pendingEvents.data(), w->pendingEvents.size() * sizeof(mi_event));
    w->pendingEvents.clear();
    return MI_OK;
}
MI_API int mi_world_get_step_mode_stats(mi_world* w, uint32_t* steps, uint32_t* spec, uint32_t* retries) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    if (steps) *steps = w->totalSteps;
    if (spec) *spec = w->specSteps;
    if (retries) *retries = w->specRetries;
    return MI_OK;
}

MI_API int mi_world_num_entities(mi_world* w, uint32_t* out) { if (!w || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null"); *out = (uint32_t)w->entities.size(); return MI_OK; }

static int getTransforms(mi_world* w, float* p, float* r, uint32_t cap, bool physics) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    uint32_t n = (uint32_t)w->entities.size();
    if (cap < n) return fail(MI_ERR_CAPACITY, "capacity < num entities");
    {   // the poses produced on the device in the caller's layout, one copy (PoseStream)
        float t;
        if (w->posesPossible(physics, &t)) {
            mi_world::PoseStream& ps = w->pose;
            if (!(ps.valid && ps.steps == w->totalSteps && ps.t == t && ps.n == n && ps.tablesValid && ps.tableCount == n)) { ++ps.produced_on_demand; int rc = w->posesProduce(t, false); if (rc != MI_OK) return rc; }
            ps.askedPhysics = physics;
            return w->posesFetch(p, r, nullptr, nullptr);
        }
    }
    {   // The caller that reads the poses after every step (a renderer): positions and rotations come straight from the device — 2 arrays (4 while an
        // interpolation is pending) instead of the 6-8 of a full download() — and the host mirror of the bodies is left alone (still stale: whoever
        // needs it downloads it).  Same arithmetic as download(), which keeps producing the same values later.
        const uint32_t nb = (uint32_t)w->bodies.size();
        const bool follow = physics || w->transformsFollowPhysics, lerpNow = !follow && w->lerpPending;
        if (w->hostStale && !w->topologyDirty && nb && (follow || lerpNow)) {
            HIP_TRY(hipSetDevice(w->device));
            const bool p0Dev = lerpNow && w->p0OnDevice;
            const size_t rows = (p0Dev ? 4u : 2u) * (size_t)nb;
            if (rows > w->downloadStageCap) {
                if (w->downloadStage) (void)hipHostFree(w->downloadStage);
                w->downloadStage = nullptr; w->downloadStageCap = 0;
                HIP_TRY(hipHostMalloc((void**)&w->downloadStage, (8u * (size_t)nb + 2u * (size_t)nb) * sizeof(float4)));   // what download() will ask for, once
                w->downloadStageCap = 8u * (size_t)nb + 2u * (size_t)nb;
            }
            float4 *pos = w->downloadStage, *rot = pos + nb, *pos0 = rot + nb, *rot0 = pos0 + nb;
            HIP_TRY(hipMemcpyAsync(pos, w->bPos.p, (size_t)nb * 16, hipMemcpyDeviceToHost, w->stream));
            HIP_TRY(hipMemcpyAsync(rot, w->bRot.p, (size_t)nb * 16, hipMemcpyDeviceToHost, w->stream));
            if (p0Dev) {
                HIP_TRY(hipMemcpyAsync(pos0, w->bPos0.p, (size_t)nb * 16, hipMemcpyDeviceToHost, w->stream));
                HIP_TRY(hipMemcpyAsync(rot0, w->bRot0.p, (size_t)nb * 16, hipMemcpyDeviceToHost, w->stream));
            }
            HIP_TRY(hipStreamSynchronize(w->stream));
            const float t = w->lerpT;
            hostParallelFor(n, [&](uint32_t i) {
                const HEntity& e = w->entities[i];
                V3 ps = e.pos; Q4 rt = e.rot;
                if (e.rb >= 0) {
                    const uint32_t b = (uint32_t)e.rb;
                    const V3 p1(pos[b].x, pos[b].y, pos[b].z); const Q4 r1(rot[b].x, rot[b].y, rot[b].z, rot[b].w);
                    if (follow) { ps = p1; rt = r1; }
                    else {   // lerp(trs): nlerp on the quaternion (src/core/math.h:673-682)
                        const V3 p0 = p0Dev ? V3(pos0[b].x, pos0[b].y, pos0[b].z) : w->bodies[b].p0;
                        const Q4 r0 = p0Dev ? Q4(rot0[b].x, rot0[b].y, rot0[b].z, rot0[b].w) : w->bodies[b].r0;
                        ps = lerp(p0, p1, t);
                        rt = normalize(Q4(r0.x + t * (r1.x - r0.x), r0.y + t * (r1.y - r0.y), r0.z + t * (r1.z - r0.z), r0.w + t * (r1.w - r0.w)));
                    }
                }
                if (p) { p[3 * i] = ps.x; p[3 * i + 1] = ps.y; p[3 * i + 2] = ps.z; }
                if (r) { r[4 * i] = rt.x; r[4 * i + 1] = rt.y; r[4 * i + 2] = rt.z; r[4 * i + 3] = rt.w; }
            });
            return MI_OK;
        }
    }
    int rc = w->download(); if (rc != MI_OK) return rc;
    hostParallelFor(n, [&](uint32_t i) {
        const HEntity& e = w->entities[i];
        V3 pos = e.pos; Q4 rot = e.rot;
        if (physics && e.rb >= 0) { pos = w->bodies[e.rb].p1; rot = w->bodies[e.rb].r1; }
        if (p) { p[3 * i] = pos.x; p[3 * i + 1] = pos.y; p[3 * i + 2] = pos.z; }
        if (r) { r[4 * i] = rot.x; r[4 * i + 1] = rot.y; r[4 * i + 2] = rot.z; r[4 * i + 3] = rot.w; }
    });
    return MI_OK;
}
MI_API int mi_world_get_transforms(mi_world* w, float* p, float* r, uint32_t cap) { return getTransforms(w, p, r, cap, false); }
MI_API int mi_world_get_physics_transforms(mi_world* w, float* p, float* r, uint32_t cap) { return getTransforms(w, p, r, cap, true); }
// The same values without the last copy: pointers to the library's pinned rows ([n][3] positions, [n][4] rotations), valid until the
// SECOND next stepping call on this world (two sets alternate).  MI_ERR_UNSUPPORTED when the poses are not coming from the device right now
// (nothing stepped since the last download, topology changed, sharded world): mi_world_get_transforms covers every case.
static int viewTransforms(mi_world* w, const float** p, const float** r, uint32_t* count, bool physics) {
    if (!w || (!p && !r)) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    const uint32_t n = (uint32_t)w->entities.size();
    float t;
    if (!w->posesPossible(physics, &t)) return fail(MI_ERR_UNSUPPORTED, "no device-side poses to view (nothing stepped since the last download, topology change pending, or sharded world): mi_world_get_transforms");
    mi_world::PoseStream& ps = w->pose;
    if (!(ps.valid && ps.steps == w->totalSteps && ps.t == t && ps.n == n && ps.tablesValid && ps.tableCount == n)) { ++ps.produced_on_demand; int rc = w->posesProduce(t, false); if (rc != MI_OK) return rc; }
    ps.askedPhysics = physics;
    if (count) *count = n;
    const float *vp = nullptr, *vr = nullptr;
    int rc = w->posesFetch(nullptr, nullptr, &vp, &vr); if (rc != MI_OK) return rc;
    if (p) *p = vp;
    if (r) *r = vr;
    return MI_OK;
}
MI_API int mi_world_view_transforms(mi_world* w, const float** p, const float** r, uint32_t* count) { return viewTransforms(w, p, r, count, false); }
MI_API int mi_world_view_physics_transforms(mi_world* w, const float** p, const float** r, uint32_t* count) { return viewTransforms(w, p, r, count, true); }
MI_API int mi_debug_pose_stream_stats(mi_world* w, uint32_t* ahead, uint32_t* on_demand) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    if (ahead) *ahead = w->pose.produced_ahead;
    if (on_demand) *on_demand = w->pose.produced_on_demand;
    return MI_OK;
}
MI_API int mi_world_get_velocities(mi_world* w, float* lin, float* ang, uint32_t cap) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    uint32_t n = (uint32_t)w->entities.size();
    if (cap < n) return fail(MI_ERR_CAPACITY, "capacity < num entities");
    const uint32_t nb = (uint32_t)w->bodies.size();
    if (w->hostStale && !w->topologyDirty && nb) {   // straight from the device, like the poses (getTransforms): two arrays, host mirror left alone
        HIP_TRY(hipSetDevice(w->device));
        if (2u * (size_t)nb > w->downloadStageCap) {
            if (w->downloadStage) (void)hipHostFree(w->downloadStage);
            w->downloadStage = nullptr; w->downloadStageCap = 0;
            HIP_TRY(hipHostMalloc((void**)&w->downloadStage, 10u * (size_t)nb * sizeof(float4)));
            w->downloadStageCap = 10u * (size_t)nb;
        }
        float4 *lv = w->downloadStage, *av = lv + nb;
        HIP_TRY(hipMemcpyAsync(lv, w->bLinVel.p, (size_t)nb * 16, hipMemcpyDeviceToHost, w->stream));
        HIP_TRY(hipMemcpyAsync(av, w->bAngVel.p, (size_t)nb * 16, hipMemcpyDeviceToHost, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
        hostParallelFor(n, [&](uint32_t i) {
            V3 v, a;
            const int b = w->entities[i].rb;
            if (b >= 0) { v = V3(lv[b].x, lv[b].y, lv[b].z); a = V3(av[b].x, av[b].y, av[b].z); }
            if (lin) { lin[3 * i] = v.x; lin[3 * i + 1] = v.y; lin[3 * i + 2] = v.z; }
            if (ang) { ang[3 * i] = a.x; ang[3 * i + 1] = a.y; ang[3 * i + 2] = a.z; }
        });
        return MI_OK;
    }
    int rc = w->download(); if (rc != MI_OK) return rc;
    hostParallelFor(n, [&](uint32_t i) {
        V3 v, a;
        if (w->entities[i].rb >= 0) { v = w->bodies[w->entities[i].rb].linVel; a = w->bodies[w->entities[i].rb].angVel; }
        if (lin) { lin[3 * i] = v.x; lin[3 * i + 1] = v.y; lin[3 * i + 2] = v.z; }
        if (ang) { ang[3 * i] = a.x; ang[3 * i + 1] = a.y; ang[3 * i + 2] = a.z; }
    });
    return MI_OK;
}
MI_API int mi_world_get_mass_properties(mi_world* w, float* invMass, float* invInertia, float* cog, uint32_t cap) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    w->recalcProperties();
    uint32_t n = (uint32_t)w->entities.size();
    if (cap < n) return fail(MI_ERR_CAPACITY, "capacity < num entities");
    for (uint32_t i = 0; i < n; ++i) {
        float im = 0.f; M3 ii = M3::zero(); V3 c;
        if (w->entities[i].rb >= 0) { const HBody& b = w->bodies[w->entities[i].rb]; im = b.invMass; ii = b.invInertia; c = b.localCOG; }
        if (invMass) invMass[i] = im;
        if (invInertia) {   // column-major like the reference's mat3 (src/core/math.h:390-397)
            float* o = invInertia + 9 * i;
            o[0] = ii.m00; o[1] = ii.m10; o[2] = ii.m20; o[3] = ii.m01; o[4] = ii.m11; o[5] = ii.m21; o[6] = ii.m02; o[7] = ii.m12; o[8] = ii.m22;
        }
        if (cog) { cog[3 * i] = c.x; cog[3 * i + 1] = c.y; cog[3 * i + 2] = c.z; }
    }
    return MI_OK;
}
MI_API int mi_world_get_counts(mi_world* w, mi_step_counts* out) { if (!w || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null"); *out = w->counts; return MI_OK; }
MI_API int mi_world_get_accumulated_stage_times(mi_world* w, mi_stage_times* out_sum, uint32_t* out_steps, uint64_t* out_contact_updates, uint32_t reset) {
    if (!w) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    w->finishTimes();
    if (out_sum) *out_sum = w->timesSum;
    if (out_steps) *out_steps = w->timesSteps;
    if (out_contact_updates) *out_contact_updates = w->contactUpdatesSum;
    if (reset) { w->timesSum = mi_stage_times{}; w->timesSteps = 0; w->contactUpdatesSum = 0; }
    return MI_OK;
}
// Which contact-solver kernel the last internal step ran: 0 k_contact_solve (one launch per colour per sweep), 1 k_contact_solve_flow,
// 2 k_contact_solve_persist, 3 k_solve_flow_islands (contacts + joint islands fused), 4 k_contact_solve_persist XCD-partitioned, 5 the same on one XCD,
// 6 k_contact_solve_blocks (spatial blocks in LDS).
MI_API int mi_debug_step_graph_stats(mi_world* w, uint32_t* out4) {
    if (!w || !out4) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    out4[0] = w->graphsEnabled ? 1u : 0u; out4[1] = w->graphHits; out4[2] = w->graphCaptures; out4[3] = w->graphPlain;
    return MI_OK;
}
// The block solver's sizes and what the last block step needed (blocks.hpp): out16 = { blocks, tiles per block, extra capacity, body capacity, hash slots, passes per wave,
// impulses per wave, LDS bytes, entries needed, extras needed, bodies needed, passes needed, impulses needed, boundary entries, block steps so far, steps the path is switched off for }.
MI_API int mi_debug_block_stats(mi_world* w, uint32_t* out16) {
    if (!w || !out16) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    const auto& c = w->blkCaps; const BlockState& b = w->lastBlk;
    const uint32_t v[16] = {c.nbe, c.tiles, c.extraCap, c.bodyCap, c.hashSize, c.maxPasses, c.impCap, (uint32_t)c.lds, b.need, b.needExtra, b.needBodies, b.needPasses, b.needImp, b.ghostLanes, w->blkSteps, w->blkDisabledSteps};
    for (int i = 0; i < 16; ++i) out16[i] = v[i];
    return MI_OK;
}
MI_API int mi_world_get_solver_kind(mi_world* w, uint32_t* out) {
    if (!w || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    *out = w->usedBlocks ? 6u : w->usedFused ? 3u : w->usedPersist ? (w->usedXcdSingle ? 5u : w->usedXcd ? 4u : 2u) : w->usedFlow ? 1u : 0u;
    return MI_OK;
}
// Host-side evaluation of the tile -> XCD assignment the XCD-partitioned solver uses on the device (tests: the per-XCD shares of
// a bin must add up, and (owner, rank) must enumerate every tile of a bin exactly once).  out = { owner, rank inside the owner's share, share of `query_xcd` }.
MI_API int mi_debug_tile_owner(uint32_t tile_in_bin, uint32_t tiles_in_bin, uint32_t bin, uint32_t query_xcd, uint32_t* out) {
    if (!out || tile_in_bin >= tiles_in_bin || query_xcd >= 8u) return fail(MI_ERR_INVALID_ARGUMENT, "tile_in_bin < tiles_in_bin, query_xcd < 8");
    out[0] = tileOwner(tile_in_bin, tiles_in_bin, bin); out[1] = tileOwnerRank(tile_in_bin, tiles_in_bin, bin); out[2] = tileOwnerCount(query_xcd, tiles_in_bin, bin);
    return MI_OK;
}
// Event pairs around every stage cost a few microseconds of device time per step each: NOTHING is timed by default (level 0: the stage times read 0, the
// accumulated step and contact-update counts still count every valid step); level 2 = the whole step and the solve stage, level 1 = every stage.
MI_API int mi_world_set_stage_timing(mi_world* w, uint32_t level) {
    if (!w || level > 2u) return fail(MI_ERR_INVALID_ARGUMENT, "level: 0 off, 1 every stage, 2 the whole step and the solve stage");
    w->stageEvents = level == 1u; w->stepEvents = level == 2u; return MI_OK;
}
MI_API int mi_world_get_stage_times(mi_world* w, mi_stage_times* out) { if (!w || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null"); w->finishTimes(); *out = w->times; return MI_OK; }

MI_API int mi_world_get_contacts(mi_world* w, mi_contact* out, uint32_t cap, uint32_t* count) {
    if (!w || !count) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    uint32_t nm = w->manifoldsLast, nc = w->counts.num_contacts;
    *count = nc;
    if (!out) return MI_OK;
    if (cap < nc) return fail(MI_ERR_CAPACITY, "capacity < num contacts");
    if (!nm) return MI_OK;
    uint32_t np = w->hs.numPairs;
    std::vector<uint32_t> mp(nm); std::vector<uint2> mb(nm), mi_(nm); std::vector<uint64_t> keys(np); std::vector<float4> nrm(np), pts(4 * (size_t)np);
    HIP_TRY(hipMemcpy(mp.data(), w->manPair.p, nm * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mb.data(), w->manBodies.p, nm * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mi_.data(), w->manInfo.p, nm * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(keys.data(), w->pairsIn, np * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(nrm.data(), w->npNormal.p, np * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pts.data(), w->npPoints.p, 4 * (size_t)np * 16, hipMemcpyDeviceToHost));
    // device manifolds are stored in arrival order; report them in ascending (bucket, colliderA, colliderB) key order
    std::vector<uint32_t> ord(nm);
    for (uint32_t m = 0; m < nm; ++m) ord[m] = m;
    std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return keys[mp[x]] < keys[mp[y]]; });
    uint32_t ci = 0;
    for (uint32_t mo = 0; mo < nm; ++mo) {
        uint32_t m = ord[mo];
        uint32_t p = mp[m], cnt = mi_[m].x & 7u;
        for (uint32_t k = 0; k < cnt && ci < nc; ++k, ++ci) {
            mi_contact& o = out[ci];
            float4 pd = pts[4 * (size_t)p + k];
            o.point[0] = pd.x; o.point[1] = pd.y; o.point[2] = pd.z; o.penetration_depth = pd.w;
            o.normal[0] = nrm[p].x; o.normal[1] = nrm[p].y; o.normal[2] = nrm[p].z;
            o.friction_restitution = mi_[m].y;
            o.collider_a = (uint32_t)((keys[p] >> 29) & 0x1FFFFFFFu); o.collider_b = (uint32_t)(keys[p] & 0x1FFFFFFFu);
            if (o.collider_b >= kHeightmapVirtualBase) o.collider_b = 0xFFFFFFFFu;   // terrain contact
            o.body_a = mb[m].x; o.body_b = mb[m].y;
        }
    }
    return MI_OK;
}

// ---- ghost-region exchange (multi-GPU sharding)
MI_API int mi_world_entities_to_bodies(mi_world* w, uint32_t n, const uint32_t* ents, uint32_t* out) {
    if (!w || (n && (!ents || !out))) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    for (uint32_t i = 0; i < n; ++i) {
        if (ents[i] >= w->entities.size() || w->entities[ents[i]].rb < 0) return fail(MI_ERR_INVALID_ARGUMENT, "not a rigid body");
        out[i] = (uint32_t)w->entities[ents[i]].rb;
    }
    return MI_OK;
}
static int statesDevice(mi_world* w, uint32_t n, const uint32_t* idsDev, float* outDev, const float* inDev, bool sync) {
    if (!w || (n && (!idsDev || (!outDev && !inDev)))) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    int rc = ensureUploaded(w); if (rc != MI_OK) return rc;
    if (n && outDev) k_gather_states<<<divUp(n, 256), 256, 0, w->stream>>>(n, idsDev, w->bPos.p, w->bRot.p, w->bLinVel.p, w->bAngVel.p, outDev);
    if (n && inDev) { k_scatter_states<<<divUp(n, 256), 256, 0, w->stream>>>(n, idsDev, inDev, w->bPos.p, w->bRot.p, w->bLinVel.p, w->bAngVel.p, w->shard.enabled ? w->shard.known.p : nullptr); w->hostStale = true; w->shard.prevValid = false; }
    if (sync) HIP_TRY(hipStreamSynchronize(w->stream));
    return MI_OK;
}
MI_API int mi_world_get_body_states_device(mi_world* w, uint32_t n, const uint32_t* idsDev, float* outDev) { return statesDevice(w, n, idsDev, outDev, nullptr, true); }
MI_API int mi_world_set_body_states_device(mi_world* w, uint32_t n, const uint32_t* idsDev, const float* inDev) { return statesDevice(w, n, idsDev, nullptr, inDev, true); }
// The same without a host synchronisation: the copy kernels are only ENQUEUED on the world's stream (mi_world_get_stream).  A
// caller that runs its collective on that stream (e.g. torch.cuda.ExternalStream + RCCL) gets gather -> exchange -> scatter ->
// next step ordered on the device with no host round trip in between.
MI_API int mi_world_get_body_states_device_async(mi_world* w, uint32_t n, const uint32_t* idsDev, float* outDev) { return statesDevice(w, n, idsDev, outDev, nullptr, false); }
MI_API int mi_world_set_body_states_device_async(mi_world* w, uint32_t n, const uint32_t* idsDev, const float* inDev) { return statesDevice(w, n, idsDev, nullptr, inDev, false); }
MI_API int mi_world_get_stream(mi_world* w, void** out) {
    if (!w || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    *out = (void*)w->stream;
    return MI_OK;
}
static int statesHost(mi_world* w, uint32_t n, const uint32_t* ents, float* out, const float* in) {
    if (!w || (n && !ents)) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    if (!n) return MI_OK;
    std::vector<uint32_t> ids(n);
    int rc = mi_world_entities_to_bodies(w, n, ents, ids.data()); if (rc != MI_OK) return rc;
    rc = ensureUploaded(w); if (rc != MI_OK) return rc;
    DBuf<uint32_t> dIds; DBuf<float> dSt;
    HIP_TRY(dIds.ensure(n)); HIP_TRY(dSt.ensure(13 * (size_t)n));
    HIP_TRY(hipMemcpy(dIds.p, ids.data(), n * 4, hipMemcpyHostToDevice));
    if (in) {
        HIP_TRY(hipMemcpy(dSt.p, in, 13 * (size_t)n * 4, hipMemcpyHostToDevice));
        return mi_world_set_body_states_device(w, n, dIds.p, dSt.p);
    }
    rc = mi_world_get_body_states_device(w, n, dIds.p, dSt.p); if (rc != MI_OK) return rc;
    HIP_TRY(hipMemcpy(out, dSt.p, 13 * (size_t)n * 4, hipMemcpyDeviceToHost));
    return MI_OK;
}
MI_API int mi_world_get_body_states(mi_world* w, uint32_t n, const uint32_t* ents, float* out) { return statesHost(w, n, ents, out, nullptr); }
MI_API int mi_world_set_body_states(mi_world* w, uint32_t n, const uint32_t* ents, const float* in) { return statesHost(w, n, ents, nullptr, in); }

// Stage dumps for parity bisecting.
MI_API int mi_world_get_aabbs(mi_world* w, float* out6, uint32_t cap) {
    if (!w || !out6) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    uint32_t nc = w->counts.num_colliders;
    if (cap < nc) return fail(MI_ERR_CAPACITY, "capacity < num colliders");
    std::vector<float4> mn(nc), mx(nc);
    if (nc) { HIP_TRY(hipMemcpy(mn.data(), w->aabbMin.p, nc * 16, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(mx.data(), w->aabbMax.p, nc * 16, hipMemcpyDeviceToHost)); }
    for (uint32_t i = 0; i < nc; ++i) { out6[6 * i] = mn[i].x; out6[6 * i + 1] = mn[i].y; out6[6 * i + 2] = mn[i].z; out6[6 * i + 3] = mx[i].x; out6[6 * i + 4] = mx[i].y; out6[6 * i + 5] = mx[i].z; }
    return MI_OK;
}
MI_API int mi_world_get_manifold_colors(mi_world* w, uint32_t* out, uint32_t cap) {
    if (!w || !out) return fail(MI_ERR_INVALID_ARGUMENT, "null");
    uint32_t nm = w->manifoldsLast;
    if (cap < nm) return fail(MI_ERR_CAPACITY, "capacity < num manifolds");
    if (!nm) return MI_OK;
    // same manifold order as mi_world_get_contacts: ascending (bucket, colliderA, colliderB)
    uint32_t np = w->hs.numPairs;
    std::vector<uint32_t> col(nm), mp(nm), ord(nm); std::vector<uint64_t> keys(np);
    HIP_TRY(hipMemcpy(col.data(), w->color.p, nm * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mp.data(), w->manPair.p, nm * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(keys.data(), w->pairsIn, np * 8, hipMemcpyDeviceToHost));
    for (uint32_t m = 0; m < nm; ++m) ord[m] = m;
    std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return keys[mp[x]] < keys[mp[y]]; });
    for (uint32_t m = 0; m < nm; ++m) out[m] = col[ord[m]];
    return MI_OK;
}

}  // extern "C"
