// Every environment variable the library reads (30 since round 6: the switches whose A/B was settled became constants), in ONE place: read once PER WORLD (mi_world_create — not once per process: two worlds created under
// different environments differ), typed, documented.  Numeric knobs: 0 (or unset) means "the default", not the value 0 (MI_FLOW_LDS, MI_PERSIST_WAVES).
// None of them is needed in production — the defaults are what is measured and shipped; they select the fallback paths the tests
// pin against each other (every variant gives the same bits), inject the faults the fallback ladder is tested with, and switch
// development output on.  No reference counterpart.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

namespace mi {

struct Knobs {
    // ---- stepping
    bool speculative = true;            // MI_ASYNC=0: every step synchronous (sizes read back inside the step)
    bool spinReadback = true;           // MI_READBACK=copy: end-of-step read-back as hipMemcpyAsync + hipStreamSynchronize instead of the kernel-published record
    bool stageEvents = false;           // constant since round 6 (was MI_STAGE_EVENTS; its A/B is settled: EXPERIMENTS.md); 1 meant: per-stage events from the first step on (otherwise mi_world_set_stage_timing)
    bool stepEvents = false;            // constant since round 6 (was MI_STEP_EVENTS; its A/B is settled: EXPERIMENTS.md); 1 meant: step + solve-stage events from the first step on
    bool poseStream = true;             // MI_POSE_STREAM=0: poses for the caller through the per-array copies + host pass
    bool debugSync = false;             // MI_DEBUG_SYNC: synchronise after every stage and name the one a device fault comes from
    bool eagerTimes = false;            // constant since round 6 (was MI_EAGER_TIMES; its A/B is settled: EXPERIMENTS.md): read the step's event times at the end of the step (not one step later)
    bool stepAhead = true;              // MI_STEP_AHEAD=0: the next step's k_bp_prepare is not enqueued behind a speculative step's end-of-step record
    bool fuseReset = true;              // MI_FUSE_RESET=0: k_reset_scalars as the first launch of every step (otherwise its work rides at the end of k_publish_readback)
    // ---- step graphs (launcher.hpp)
    std::string graph;                  // MI_GRAPH=0 | force | all ("" = by runtime version)
    uint32_t graphMaxColliders = 32768; // constant since round 6 (was MI_GRAPH_MAX_COLLIDERS; its A/B is settled: EXPERIMENTS.md)
    bool graphDebug = false, graphNoEvents = false, graphNoCapture = false;   // MI_GRAPH_DEBUG (the other two: constants since round 6)
    // ---- broad / narrow phase
    bool fuseWorld = true;              // MI_FUSE_WORLD=0: k_world_colliders as its own launch
    bool fuseLarge = true;              // MI_FUSE_LARGE=0: k_bp_pairs_grid and k_bp_pairs_large as two launches (otherwise k_bp_pairs runs the large pass in the first workgroups of the grid pass's launch)
    bool finishInNarrow = true;         // MI_FINISH_IN_NARROW=0: k_pair_finish as its own launch also in steps without k_pair_partition
    bool statsInEmit = true;            // constant since round 6 (was MI_STATS_IN_EMIT; its A/B is settled: EXPERIMENTS.md); 0 meant: the centre statistics / next grid in k_pair_finish, on the step's critical path, instead of an extra workgroup of k_narrow_clip (development)
    bool fuseKeys = true;               // MI_FUSE_KEYS=0: k_integrate_forces and k_manifold_keys as two launches (otherwise k_forces_keys)
    bool skipPartition = true;          // constant since round 6 (was MI_SKIP_PARTITION; its A/B is settled: EXPERIMENTS.md); 0 meant: always launch k_pair_partition
    int gjkWave = -1;                   // MI_GJK_WAVE=0 / 1: force the lane / wave GJK variant
    bool hmStash = true;                // constant since round 6 (was MI_HM_STASH; its A/B is settled: EXPERIMENTS.md); 0 meant: terrain triangles recomputed instead of stashed
    // ---- schedule
    uint32_t colorMargin = 3;           // constant since round 6 (was MI_COLOR_MARGIN; its A/B is settled: EXPERIMENTS.md): colour rounds enqueued beyond the previous step's count (1 is ~3 us faster at the bench state and costs a synchronous re-run
                                        // whenever a growing scene needs two more rounds than the step before: measured in round 4, not kept)
    bool colorTail = true;              // MI_COLOR_TAIL=0: a margin of colouring rounds enqueued every step (MI_COLOR_MARGIN) and a synchronous re-run beyond it, instead of the rounds the
                                        // previous step needed + whatever is missing run inside k_bin_hist
    uint32_t colorTailMargin = 1;       // constant since round 6 (was MI_COLOR_TAIL_MARGIN; its A/B is settled: EXPERIMENTS.md): rounds enqueued beyond the previous step's count when the tail is on
    uint32_t colorRoundsMax = 0;        // MI_COLOR_ROUNDS_MAX: at most so many colouring rounds enqueued per speculative step, the tail runs the rest (tests; 0 = no cap)
    bool round0InEmit = true;           // MI_ROUND0_EMIT=0: colouring round 0 as its own launch (otherwise k_emit_manifolds makes the proposals of the manifolds it leaves uncoloured)
    bool xcdNoSort = false;             // constant since round 6 (was MI_XCD_NOSORT; its A/B is settled: EXPERIMENTS.md): manifold order as emitted (development)
    bool xcdStats = false;              // constant since round 6 (was MI_XCD_STATS; its A/B is settled: EXPERIMENTS.md): how many bodies stayed XCD-local (development)
    bool xcdSwizzle = false;            // constant since round 6 (was MI_XCD_SWIZZLE; its A/B is settled: EXPERIMENTS.md); 1 meant: 
    // ---- contact solver
    std::string solver;                 // MI_SOLVER=launch | flow | persist | persist-global | persist-granules ("" = persist)
    uint32_t flowLds = 0;               // constant since round 6 (was MI_FLOW_LDS; its A/B is settled: EXPERIMENTS.md) (bytes; 0 = default)
    uint32_t persistWaves = 0;          // MI_PERSIST_WAVES (0 = 4 per CU)
    bool persistXcdOnly = false;        // constant since round 6 (was MI_PERSIST_XCD_ONLY; its A/B is settled: EXPERIMENTS.md) (development)
    bool persistResident = true;        // MI_PERSIST_RESIDENT=0: every tile's rows stream (otherwise the first six contact-tiles of a wave stay in a0..a143)
    int persistXcd = -1, persistXcdSingle = -1;   // MI_PERSIST_XCD / _SINGLE = 0 / 1 (-1 = default)
    int xcdMinManifolds = -1;           // MI_PERSIST_XCD_MIN
    bool xcdFault = false, flowFault = false;   // MI_PERSIST_XCD_FAULT / MI_FLOW_FAULT: fault injection (tests)
    std::string gvelAlloc, impAlloc;    // constant since round 6 (was MI_GVEL_ALLOC; its A/B is settled: EXPERIMENTS.md)
    // ---- joints
    bool fuseJoints = true;             // constant since round 6 (was MI_FUSE_JOINTS; its A/B is settled: EXPERIMENTS.md); 0 meant: 
    bool jointIslands = true;           // constant since round 6 (was MI_JOINT_ISLANDS; its A/B is settled: EXPERIMENTS.md); 0 meant: 
    int islandPrivate = -1;             // MI_ISLAND_PRIVATE=0: every island through the dataflow
    // ---- sharding
    bool shardBlockSkip = true;         // MI_SHARD_BLOCK_SKIP=0: the per-body / per-collider passes of a sharded world visit every block of 256 (otherwise only those with something simulated in them)
    bool shardAdaptive = true;          // MI_SHARD_ADAPTIVE=0: neighbour messages always at full capacity
    // ---- development dumps
    std::string timelineOut; uint64_t timelineStep = 3;   // MI_DBG_TIMELINE_OUT / _STEP (-DMI_DBG_TIMELINE builds)
    uint32_t knockout = 0;              // MI_DBG_KNOCKOUT=bits (-DMI_DBG_KNOCKOUT builds): a second, knocked-out launch of the persistent solver per step on scratch arrays (kernels.hpp, g_dbgKnock)

    static Knobs fromEnvironment() {
        Knobs k;
        auto str = [](const char* n) { const char* v = std::getenv(n); return std::string(v ? v : ""); };
        auto set = [](const char* n) { return std::getenv(n) != nullptr; };
        auto off = [](const char* n) { const char* v = std::getenv(n); return v && v[0] == '0'; };   // "=0" switches a default-on feature off
        auto on = [](const char* n) { const char* v = std::getenv(n); return v && v[0] != '0'; };
        auto tri = [](const char* n) { const char* v = std::getenv(n); return !v ? -1 : (v[0] != '0' ? 1 : 0); };
        auto num = [](const char* n, uint64_t d) { const char* v = std::getenv(n); return v ? (uint64_t)strtoull(v, nullptr, 0) : d; };
        k.speculative = !off("MI_ASYNC"); k.spinReadback = str("MI_READBACK") != "copy";
        k.poseStream = !off("MI_POSE_STREAM"); k.debugSync = set("MI_DEBUG_SYNC");
        k.fuseReset = !off("MI_FUSE_RESET"); k.stepAhead = !off("MI_STEP_AHEAD");
        k.graph = str("MI_GRAPH");
        k.graphDebug = set("MI_GRAPH_DEBUG");
        k.fuseWorld = !off("MI_FUSE_WORLD"); k.fuseLarge = !off("MI_FUSE_LARGE"); k.finishInNarrow = !off("MI_FINISH_IN_NARROW"); k.fuseKeys = !off("MI_FUSE_KEYS");
        if (const char* v = std::getenv("MI_GJK_WAVE")) k.gjkWave = atoi(v);
        k.round0InEmit = !off("MI_ROUND0_EMIT"); k.colorTail = !off("MI_COLOR_TAIL"); k.colorRoundsMax = (uint32_t)num("MI_COLOR_ROUNDS_MAX", 0);
        k.solver = str("MI_SOLVER"); k.persistWaves = (uint32_t)num("MI_PERSIST_WAVES", 0); k.persistResident = !off("MI_PERSIST_RESIDENT");
        k.persistXcd = tri("MI_PERSIST_XCD"); k.persistXcdSingle = tri("MI_PERSIST_XCD_SINGLE");
        if (const char* v = std::getenv("MI_PERSIST_XCD_MIN")) k.xcdMinManifolds = (int)strtoul(v, nullptr, 0);
        k.xcdFault = set("MI_PERSIST_XCD_FAULT"); k.flowFault = set("MI_FLOW_FAULT");
        k.islandPrivate = tri("MI_ISLAND_PRIVATE");
        k.shardAdaptive = !off("MI_SHARD_ADAPTIVE"); k.shardBlockSkip = !off("MI_SHARD_BLOCK_SKIP");
        k.timelineOut = str("MI_DBG_TIMELINE_OUT"); k.timelineStep = num("MI_DBG_TIMELINE_STEP", 3); k.knockout = (uint32_t)num("MI_DBG_KNOCKOUT", 0);
        return k;
    }
};

}  // namespace mi
