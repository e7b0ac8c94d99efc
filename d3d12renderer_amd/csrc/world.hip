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
#include "gjk.hpp"
#include "launcher.hpp"
#include "joints.hpp"
#include "heightmap.hpp"
#include "cloth.hpp"
#include "knobs.hpp"

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
    Knobs knobs;   // the environment, read once at creation (knobs.hpp)
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
        // anything that can put many bodies into a ghost strip at once (a restore, a re-upload, states written from outside) sends the next messages at full size again; with the
        // library transport such a call is COLLECTIVE: every rank makes it between the same two steps (include/mi_shard.h), or the ranks disagree about the message sizes
        void rearmFullSize() { fullExchanges = 2; sweepFullSteps = 2; }
        // "a rank pays for what it simulates" (kernels.hpp, shardBlockRecent): per 256-body block the step of its last activity / received record and whether anything in it was
        // simulated in this step or the previous one; per 256-collider block whether it holds a live collider.  blocksAll: every block counts as recent in the next step (set by
        // whatever can change a body's status behind the flags' back: enable, new borders; an upload / restore / outside state write does it through prevValid)
        DBuf<uint32_t> blockStamp; DBuf<uint8_t> blockLive, cbLive; bool blocksAll = true;
        DBuf<uint32_t> stepDev; uint32_t* stepHost = nullptr;   // the step's number for k_shard_classify: a pinned host word copied to the device at the top of a step (same operation every step: a step graph replays it)
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
        bool wantVel = false, hasVel = false, velConsumed = false;   // the velocities ride in the rows once a caller has read them after a step (and stop when nobody reads them)
        uint32_t produced_ahead = 0, produced_on_demand = 0;
        // per (flavour, set): what the set holds and the event behind its last piece — mi_world_view_transforms_landed hands out the newest set that is COMPLETE in host memory
        // without waiting for the one still on the bus (a renderer one frame behind: its frame then overlaps the copy of the step it has not seen yet)
        struct SetInfo { hipEvent_t landed = nullptr; bool valid = false; uint64_t steps = 0; float t = 0.f; uint32_t n = 0; } sets[2][2];
    } pose;
    struct PoseArm { bool armed = false, done = false; float t = 0.f; } poseArm;   // the stepping call wants the LAST of its internal steps to enqueue the rows itself, behind its kernels and ahead of the host's wait
    bool posesPossible(bool physics, float* t) const;
    bool posesWantedAhead();
    void posesArm(bool lerpAfterwards, float lerpTAfterwards);
    int posesProduce(float t, bool fromNextState, bool ahead);
    void posesAbort();
    int posesViewLanded(bool physics, const float** p, const float** r, uint32_t* count, uint64_t* ofStep);
    int posesFetch(float* p, float* r, const float** viewP, const float** viewR, float* lin = nullptr, float* ang = nullptr, const float** viewL = nullptr, const float** viewA = nullptr);
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
    bool persistXcd = true, persistXcdSingle = true, usedXcd = false, usedXcdSingle = false, lastXcdSingle = false, haveXcdEstimate = false, xcdFaultFired = false, xcdFaultTest = false; uint32_t lastXcdMax = 0, xcdMinManifolds = 16384;
    // device: colliders
    DBuf<uint32_t> cTypeBody; DBuf<float4> cShape, cStaticPos, cStaticRot, cEmit;
    DBuf<float4> wShape, aabbMin, aabbMax, hullAabb, hullVerts; DBuf<uint32_t> hullRanges;
    // Step-ahead (round 6): a speculative step enqueues the NEXT step's first kernel (k_bp_prepare: world colliders + grid classification, ~27 us) right behind its own end-of-step
    // record, on the state it has just produced, into the OTHER set of world-shape / AABB rows — the device then has work while the host validates this step, returns to the caller
    // and comes back with the next step's launches (~25 us the device used to idle).  The next step adopts the result if nothing its inputs depend on has changed (same arrays, same
    // grid, same axis, no outside write: `stale`), otherwise it clears what the kernel counted and starts as before.
    DBuf<float4> wShapeAlt, aabbMinAlt, aabbMaxAlt;
    struct Ahead { bool pending = false, stale = false; uint32_t nc = 0, nb = 0, gridIdx = 0, axis = 0; const void* pos = nullptr; const void* rot = nullptr; const void* shape = nullptr; } ahead; uint64_t aheadUsed = 0, aheadEnqueued = 0;
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
    DBuf<HistSlot> tab[2]; uint32_t tabMask[2] = {0, 0}; int tabCur = 0; bool tabValid = false; DBuf<uint32_t> histHint[2]; int hintOf[2] = {0, 0};   // histHint: the history tables' per-home-slot probe hints (kernels_narrow.hpp, tableFind); hintOf[t] = the array table t uses (both tables share one while they keep their size)
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
#if defined(MI_DBG_KNOCKOUT) || defined(MI_DBG_TIMELINE)
    bool knockPending = false, knockPendingEmit = false, knockPendingBp = false; double knockMsSum = 0.0; uint32_t knockLaunches = 0; unsigned long long* dbgTimelineBuf = nullptr;   // development builds only (world_step.inc)
#endif
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
    uint32_t readbackSeq = 0; bool spinReadback = true, stageEvents = false /* time every stage */, stepEvents = false /* time the whole step and the solve stage */, solveEventsOnly = false /* ... the solve stage alone */, timesPendingEnds = true;
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
    DBuf<uint2> cbRange;   // per 256-collider block: the range of 256-body blocks its colliders' bodies lie in (x > y: a static collider among them — always visited); upload()
    uint64_t tailSteps = 0, tailRoundsSum = 0;   // valid steps whose colouring was finished inside k_bin_hist, and the rounds it ran there (mi_debug_color_tail_stats)
    uint32_t colorBatchSticky = 0;   // colouring rounds a graph-replaying scene enqueues per step (runStep)
    uint32_t colorTailHold = 0;  // steps left during which the colouring tail stays off (its barrier timed out: kernels_schedule.hpp)
    bool scalarsClean = false;   // the device-side step scalars / counters are already cleared for the next attempt (k_publish_readback did k_reset_scalars' work)
    uint32_t sapAxis = 0;        // sorting axis for the next step (collision_broad.cpp:443-444), host copy
    // mi_debug_set_solve_order: the next internal step solves these oriented collider pairs (a << 29 | b) sequentially, in this order, and the joints in pool order
    std::vector<uint64_t> debugOrder; std::vector<uint32_t> debugRank; bool debugOrderPending = false, debugOrderDataflow = false, debugOrderLevelled = false; uint32_t debugOrderDepth = 0, debugOrderDepthLast = 0;
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
    HIP_TRY(scalarsRaw.ensure(sizeof(StepScalars) + kRoundFlagWords * sizeof(uint32_t)));
    HIP_TRY(grid.ensure(2));
    HIP_TRY(shards.ensure(1));
    HIP_TRY(hipMemsetAsync(scalarsRaw.p, 0, sizeof(StepScalars) + kRoundFlagWords * sizeof(uint32_t), stream));
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
    knobs = Knobs::fromEnvironment();
    const Knobs& kn = knobs;
    if (kn.graph == "0") graphsEnabled = false;   // never replay steps as HIP graphs
    if (kn.graph == "force") graphsEnabled = true;
    graphsForAll = kn.graph == "all";             // also the large scenes
    graphMaxColliders = kn.graphMaxColliders; graphDebug = kn.graphDebug; graphNoEvents = kn.graphNoEvents; graphNoCapture = kn.graphNoCapture;
    pose.enabled = kn.poseStream;
    spinReadback = spinReadback && kn.spinReadback;
    stageEvents = kn.stageEvents; stepEvents = kn.stepEvents;   // default: nothing is timed (mi_world_set_stage_timing)
    xcdSwizzle = kn.xcdSwizzle;
    const std::string& sv = kn.solver;
    flowSolver = sv != "launch";
    specEnabled = kn.speculative;
    if (kn.flowLds) flowLds = kn.flowLds;
    // The solver-side body velocities (and the impulse granules) are exchanged between workgroups through 16-byte sc1
    // transactions: memory the L2 never caches (MTYPE_UC) serves them measurably faster than default device memory
    // (solve 0.86 -> 0.79 ms at 262144 bodies).  MI_GVEL_ALLOC / MI_IMP_ALLOC = plain | finegrained | uncached override.
    auto allocFlags = [](const std::string& v, unsigned dflt) {
        if (v.empty()) return dflt;
        return v == "finegrained" ? (unsigned)hipDeviceMallocFinegrained : v == "uncached" ? (unsigned)hipDeviceMallocUncached : 0u;
    };
    gVel.flags = allocFlags(kn.gvelAlloc, hipDeviceMallocUncached);
    imp.flags = allocFlags(kn.impAlloc, 0u);
    persistSolver = sv.empty() || sv == "persist" || sv == "persist-global" || sv == "persist-granules" || sv == "blocks";   // default; MI_SOLVER=flow / launch select the other contact solvers
    persistImpLds = sv != "persist-granules";   // persist-granules: impulses as tagged granules as well (what the largest piles get automatically)
    persistMetaLds = sv != "persist-global" && sv != "persist-granules";   // persist-global: slot data always from global memory (the variant larger problems get automatically)
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) persistWaves = 4u * (uint32_t)prop.multiProcessorCount; }
    if (kn.persistWaves) persistWaves = kn.persistWaves;
    xcdOnly = kn.persistXcdOnly ? 1u : 0u;                                   // development experiment: one XCD's workgroups do all the work
    if (kn.persistXcd >= 0) persistXcd = kn.persistXcd != 0;                 // 0: no XCD partitioning (every body through memory)
    if (kn.islandPrivate >= 0) privateIslandsEnabled = kn.islandPrivate != 0;   // development / tests: every island through the dataflow
    if (kn.persistXcdSingle >= 0) persistXcdSingle = kn.persistXcdSingle != 0;   // 0: small piles on all XCDs, every body through memory
    xcdFaultTest = kn.xcdFault; flowFaultTest = kn.flowFault;
    if (kn.xcdMinManifolds >= 0) xcdMinManifolds = (uint32_t)kn.xcdMinManifolds;   // smallest manifold count that is partitioned (tests: 1)
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
    for (auto& ff : pose.sets) for (auto& si : ff) if (si.landed) (void)hipEventDestroy(si.landed);
    if (pose.copyStream) (void)hipStreamDestroy(pose.copyStream);
    if (shard.sentHost) (void)hipHostFree(shard.sentHost);
    if (shard.stepHost) (void)hipHostFree(shard.stepHost);
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

#include "world_setup.inc"   // mass properties (setup time), upload / download
#include "world_step.inc"   // one internal step: enqueue, read-back, validation, fallback ladder; poses for the caller
#include "world_joints.inc"   // joint storage on the host, islands
#include "world_capi.inc"   // the C ABI of include/mi_physics.h and include/mi_constraints.h
#include "world_shard.inc"   // sharded world: include/mi_shard.h (tiles, ghosts, RCCL transport, exact seam, load balance)
#include "world_state.inc"   // checkpoints, body states on the device, stage dumps, debug entry points
