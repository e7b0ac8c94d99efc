/*
 * mi_shard.h — one scene on several GPUs: spatial sharding of an mi_world (the reference has no multi-GPU path).
 *
 * Model.  One process per GPU.  EVERY rank creates the SAME scene (same entities, colliders, in the same order: body and
 * collider indices, pair priorities and colour-history keys then mean the same thing everywhere; 288 GB of HBM hold tens of
 * millions of bodies) and enables sharding with its own rank.  The x-z plane is cut into tiles_x * tiles_z tiles, one per rank
 * (tile -> rank in Morton order of the tile coordinates); rim tiles extend to infinity.  At the start of every internal step a
 * rank classifies every body by the position of its centre of gravity:
 *     OWNED  its tile contains the centre          -> simulated and integrated here
 *     GHOST  within `ghost_margin` of its tile     -> takes part in collision detection and in the solve here; its new state
 *                                                     is taken from its owner
 *     else                                         -> not simulated here this step (its colliders leave the broad phase)
 * so a body that crosses a tile border simply changes owner at the next step (MIGRATION needs no bookkeeping: the new owner has
 * had the body as a ghost).  The borders themselves can move to where the bodies are (load balance, below).  After the step every rank sends, to each of its <= 8 neighbour tiles, the new state (56-byte
 * record: body index + position, rotation, linear and angular velocity) of every body it owned whose OLD or NEW centre lies in
 * that neighbour's extended tile, and applies what it receives.  Counts follow an owner rule (a manifold belongs to the owner of
 * its first dynamic body), so the sum over ranks counts every manifold once.
 *
 * What this is numerically: block Jacobi across the tile seams — each tile solves its extended region for all PGS sweeps of a
 * step from the same start state, owners' results win.  Inside a tile everything is the single-GPU pipeline, bit for bit.  The
 * result of R ranks equals, bit for bit, R such worlds stepped one after the other in ONE process with the records copied by hand
 * (tests do exactly that, on the CPU oracle over gloo and on one GPU); it does NOT equal the unsharded world — the seam coupling
 * of a step is one step late — and tests bound that difference.  An ARTICULATED ISLAND (bodies connected by constraints: a ragdoll,
 * a vehicle) is classified as one — by the centre of its root body (lowest body index) — so no constraint ever spans ranks and islands
 * migrate whole; ghost_margin then has to cover an island's reach as well.  Heightmap terrain is static and replicated (a collider of a
 * body this rank does not simulate takes no part in it); cloths do not interact with rigid bodies, every rank steps all of them identically.
 *
 * Cost of the replicated scene.  A rank HOLDS every body (memory is O(scene) per rank) but its per-body and per-collider passes visit only the blocks of 256 in which it
 * simulates something: per block the library keeps the step of its last activity or received record (classification and packing look at blocks active within the last two
 * steps) and whether a body in it is simulated now or was in the previous step (the integrators, the collider pass).  One GPU as the middle rank of an 8-tile, 2 M-body scene
 * steps 7.4 % slower than the 262 144-body world alone.  Anything that moves bodies behind the flags' back — a re-upload, a restore, states written from outside,
 * new borders — makes every block count as active for the next step.
 *
 * Transport.  Either the library's own: RCCL point-to-point (ncclSend / ncclRecv in one group per step, on the world's stream,
 * fixed-size messages so that no host read-back sits between the step and the exchange) — mi_shard_get_unique_id on rank 0,
 * distribute the 128 bytes by any means, mi_world_shard_attach_rccl everywhere.  Or the caller's: mi_world_shard_export /
 * mi_world_shard_import move the same messages through host memory (MPI, gloo, a test harness).
 */
#ifndef MI_SHARD_H
#define MI_SHARD_H
#include "mi_physics.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_shard_desc {
    uint32_t rank, num_ranks;       /* num_ranks == tiles_x * tiles_z */
    float origin_x, origin_z;       /* min corner of the tile grid */
    float tile_size_x, tile_size_z;
    uint32_t tiles_x, tiles_z;
    float ghost_margin;             /* >= largest collider extent + the distance a body can travel in one step; < tile size */
    uint32_t max_records;           /* capacity of one neighbour message in records (0 = max(4096, bodies / num_ranks / 4)); equal on all ranks.
                                     * A message always travels whole — (max_records + 1) * 56 bytes per neighbour per step — so size it for the bodies in
                                     * one margin strip (the library reports an overflow as MI_ERR_CAPACITY, it never drops records silently) */
} mi_shard_desc;

#define MI_SHARD_RECORD_FLOATS 14   /* body index (bit pattern) + 13 state floats; record 0 of a message = (count, unused...) */

/* Tile of a rank / rank of a tile under the Morton mapping (pure functions; tiles are numbered tz * tiles_x + tx). */
MI_API int mi_shard_tile_of_rank(uint32_t tiles_x, uint32_t tiles_z, uint32_t rank, uint32_t* out_tile);
MI_API int mi_shard_rank_of_tile(uint32_t tiles_x, uint32_t tiles_z, uint32_t tile, uint32_t* out_rank);

MI_API int mi_world_shard_enable(mi_world* world, const mi_shard_desc* desc);
/* Neighbour ranks of this world's tile, ascending tile order (the order of the message slots); returns the count (<= 8). */
MI_API int mi_world_shard_neighbours(mi_world* world, uint32_t* out_ranks8, uint32_t* out_count);
/* Owned bodies / manifolds / contacts of the last internal step (sum over ranks = every body / manifold / contact once). */
MI_API int mi_world_shard_counts(mi_world* world, uint32_t* out_bodies, uint32_t* out_manifolds, uint32_t* out_contacts);
/* Entity ids of the bodies this rank owned in the last internal step (their read-backs are authoritative). */
MI_API int mi_world_shard_owned_entities(mi_world* world, uint32_t* out_entities, uint32_t capacity, uint32_t* out_count);

/* Load balance: the tile borders follow the bodies ("rebalanced every K steps by body count").  Tiles are cut by tiles_x - 1
 * interior borders along x and tiles_z - 1 along z (ascending; uniform — origin + i * tile_size — when sharding is enabled).  Every K steps, between
 * two steps and on every rank:
 *     mi_world_shard_histogram       this rank's owned bodies per bin along an axis          -> sum over the ranks (the caller's all-reduce: 2 KB)
 *     mi_shard_balance_borders       borders that even out the counts, clamped to one change  (pure arithmetic: every rank computes the same)
 *     mi_world_shard_set_borders     in force after the NEXT internal step's exchange
 * That next step still simulates under the old borders; its messages additionally carry every owned body the neighbour will own or hold as a ghost
 * under the new ones, so the switch needs no extra round trip and no rank ever classifies a body from a copy that is not current (a rank only trusts
 * its copy of a body it owned in the last step or got a record for).  What one change may do: border i stays within
 * [old border i-1 + ghost_margin, old border i+1 - ghost_margin] (a body's new owner is then the old owner's tile or a neighbour of it) and tiles stay
 * wider than ghost_margin; anything else is MI_ERR_INVALID_ARGUMENT.  mi_shard_balance_borders moves a border by at most two margins per round (the
 * hand-over has to fit the one neighbour message, whose capacity max_records is sized in margin strips): a badly laid out grid converges over a few rounds.  A null axis stays as it is.  Identical values on all ranks, between the same steps. */
MI_API int mi_world_shard_histogram(mi_world* world, uint32_t axis /* 0 = x, 1 = z */, float lo, float hi, uint32_t bins, uint32_t* out_counts /* the end bins take what lies outside [lo, hi) */);
MI_API int mi_shard_balance_borders(const uint64_t* hist, uint32_t bins, float lo, float hi, uint32_t tiles, const float* current_borders, float ghost_margin, float* out_borders /* tiles - 1 */);
MI_API int mi_world_shard_set_borders(mi_world* world, const float* borders_x /* tiles_x - 1, or null */, const float* borders_z /* tiles_z - 1, or null */);
MI_API int mi_world_shard_get_borders(mi_world* world, float* out_borders_x, float* out_borders_z);   /* the borders in force (either may be null) */

/* Global sweep axis.  The one global quantity of the pipeline is the broad phase's sweep axis (the reference: largest variance of the collider
 * centres, src/physics/collision_broad.cpp:376-384, 443-444): it orients pairs of equal shape type, so ranks that disagreed on it would give
 * last-bit differences from the single world even for bodies that never meet a seam.  The statistic is therefore a sum of INTEGERS (centres
 * quantised to 1/1024 m; q and q * q, the latter in two 32-bit halves: 9 counters), which does not depend on how the colliders are partitioned:
 * every rank sums the colliders of the bodies it OWNS (rank 0 also the colliders without a rigid body — statics, triggers, force fields are
 * replicated), the sums are added over all ranks and every rank derives the same axis as the unsharded world.  The library transport does
 * that inside the exchange (one 72-byte ncclAllReduce on the world's stream, the axis never leaves the device).  With the caller's transport:
 *     mi_world_shard_axis_sums      this rank's 9 counters of the last internal step        -> sum over the ranks (the caller's all-reduce)
 *     mi_world_shard_set_axis_sums  the global sums, on every rank, before its next step
 * (skipped: a rank falls back to the axis of its own sums — valid, but no longer the single world's orientation of equal-type pairs). */
MI_API int mi_world_shard_axis_sums(mi_world* world, uint64_t* out9);
MI_API int mi_world_shard_set_axis_sums(mi_world* world, const uint64_t* global9);

/* What the exchanges of this rank cost and moved (diagnostics; bench.py's N > 1 line).  device_ms_sum: time on the world's stream from the pack
 * kernel to the end of the unpack / axis kernels (library transport: including the RCCL sends and receives), summed over `exchanges`. */
typedef struct mi_shard_exchange_stats {
    uint64_t exchanges; double device_ms_sum;
    uint64_t message_bytes;               /* one neighbour message at full size (max_records + 1 records) */
    uint32_t num_neighbours, library_transport;
    uint32_t neighbour_rank[8], records_last[8]; uint64_t records_sum[8];   /* per neighbour slot: records (56 B) packed in the last exchange / since the last reset */
    uint32_t owned_bodies, ghost_bodies;  /* of the last internal step */
    uint64_t sweep_exchanges;             /* exact seam: hand-overs after a sweep since the last reset (iterations per internal step each) */
    uint64_t sweep_message_bytes;         /* ... one sweep message as the library transport sends it (fixed size) */
    uint32_t sweep_records_last[8];       /* ... records (32 B) per neighbour message of the last internal step */
    uint32_t message_records_last[8];     /* library transport: records each neighbour message of the last exchange could hold AS IT TRAVELLED: max(sent, received) of the
                                             exchange before x 1.5 + 512 — both ends derive the same number from the same two counts —, max_records right after enable / attach /
                                             new borders, always with MI_SHARD_ADAPTIVE=0, and with the caller's transport (mi_world_shard_export hands out whole messages) */
    uint64_t message_bytes_sum;           /* library transport: bytes sent since the last reset (all neighbours, headers included) */
} mi_shard_exchange_stats;
MI_API int mi_world_shard_exchange_stats(mi_world* world, mi_shard_exchange_stats* out, uint32_t reset);

/* Library transport: RCCL.  out_id128 / id128: the 128 bytes of an ncclUniqueId. */
MI_API int mi_shard_get_unique_id(void* out_id128);
/* Under the library transport three more calls are COLLECTIVE in effect: mi_world_load_checkpoint, a scene upload after an edit, and mi_world_set_body_states(_device).
 * Each re-arms full-size messages for the next two exchanges on the rank that makes it (message sizes are otherwise derived from the previous exchange, identically on both
 * ends, without talking) — so every rank of the grid has to make the same call between the same two steps, like mi_world_shard_set_borders.  A rank that did not sends and
 * expects the smaller size; the mismatch is detected (header word 1 carries the sender's sizing policy and count) and reported as MI_ERR_CAPACITY on both ends, never a silent cut. */
MI_API int mi_world_shard_attach_rccl(mi_world* world, const void* id128);
/* 1 if this process can use the library transport (librccl found, all entry points resolved) — a cheap, NON-collective probe: agree on it over
 * all ranks BEFORE any rank calls mi_world_shard_attach_rccl (ncclCommInitRank blocks until every rank has entered it). */
MI_API int mi_shard_library_transport_available(void);
/* Sum of `n` 64-bit counters over all ranks through the library transport: ONE ncclAllReduce on the world's stream (global body / manifold /
 * contact counts; the histograms of the load balance).  MI_ERR_UNSUPPORTED with the caller's transport (reduce them yourself). */
MI_API int mi_world_shard_allreduce_u64(mi_world* world, uint64_t* inout, uint32_t n);
/* One rebalancing round in one call (library transport; every rank, between the same two steps): histograms of both axes (`bins` each, over the
 * extent of the tile grid as enabled) -> all-reduce -> mi_shard_balance_borders -> mi_world_shard_set_borders. */
MI_API int mi_world_shard_rebalance(mi_world* world, uint32_t bins);
/* Back to the caller's transport (destroys the communicator; e.g. when another rank could not attach). */
MI_API int mi_world_shard_detach_rccl(mi_world* world);
/* Caller's transport: after mi_world_step_fixed(world, ..., 1) copy the message for neighbour slot `slot` out (host memory,
 * mi_world_shard_message_bytes bytes) and hand the neighbours' messages in; both may be called in any order across slots.  ONE internal step per
 * call in this mode (the exchange lies between two steps): mi_world_step_fixed(…, 1), or mi_world_step with max_physics_iterations_per_frame = 1;
 * anything else is MI_ERR_INVALID_ARGUMENT.  mi_world_shard_import is MI_ERR_INVALID_ARGUMENT on a world attached to the library transport.
 * Overflow: a message that did not fit is MI_ERR_CAPACITY from the step that packed it (caller's transport) or — library transport, which does
 * not wait for the counts — from the next step, mi_world_shard_counts, mi_world_shard_owned_entities, mi_world_shard_exchange_stats,
 * mi_world_save_checkpoint or mi_world_shard_detach_rccl, whichever comes first: call one of them after the last step of a run. */
MI_API int mi_world_shard_message_bytes(mi_world* world, uint64_t* out_bytes);
MI_API int mi_world_shard_export(mi_world* world, uint32_t slot, void* out_message);
MI_API int mi_world_shard_import(mi_world* world, const void* message);

/* Exact seam (an OPTION; the default stays block Jacobi).  What the block-Jacobi seam drops is the coupling, within one step, between the sweeps of
 * neighbouring tiles (the reference solves all constraints of a sweep one after the other: src/physics/constraints.cpp:3748-3770).  The exact mode
 * keeps it by ORDERING the solve around the tiling:
 *     shared body    its centre (its island root's) lies in the extended region of at least two tiles — every rank that sees it says so from
 *                    the same positions;
 *     seam manifold  every dynamic body of it is shared.  Both tiles of the seam see it with all of its bodies.
 * Seam manifolds are coloured among themselves and take the LEADING colours [0, MI_SEAM_COLORS), everything else the colours behind them (a manifold
 * that changes class is re-coloured).  A sweep is then: joints; the seam colours — every rank that sees a seam manifold solves it, redundantly, from
 * identical inputs, to identical results; the interior colours — each rank its own; and one exchange: every rank sends the velocities of the shared
 * bodies it OWNS (they alone changed through manifolds the neighbour does not see) to the neighbour that holds them as ghosts.  20 sweeps = 20 small
 * exchanges per step instead of one.
 * The result is, bit for bit, that of ONE world which was told the tiling (mi_world_set_seam_tiling: it classifies shared bodies the same way and
 * orders its colours the same way, nothing else changes) — tests assert that on virtual ranks of one GPU and on the CPU oracle.  Conditions, all
 * checked: x- or z-slabs only (tiles_x == 1 or tiles_z == 1: at a corner a shared body is seen by four tiles); ghost_margin covers the reach of a
 * contact and of an island (a manifold outside the seam class touching a ghost, or a seam manifold that found no colour among the MI_SEAM_COLORS,
 * is counted as a violation: mi_world_seam_stats; the step still completes, as block Jacobi would for that manifold).
 * Transport.  Library transport (RCCL attached): the per-sweep send / receive runs inside mi_world_step on the world's stream.  Caller's transport:
 * the library calls `exchange(user, world, sweep)` after every sweep of every internal step, on the stepping thread; inside it the caller moves
 * mi_world_shard_export_sweep(slot) of every rank to mi_world_shard_import_sweep of the neighbour (fixed-size messages of
 * mi_world_shard_sweep_message_bytes: count + records of MI_SHARD_SWEEP_FLOATS floats = body index, linear velocity, angular velocity; only the
 * header and the `count` records are written / read, so a transport may send just that prefix) and returns
 * MI_OK.  Every rank makes the same number of calls (the sweeps of the step), so a barrier inside the callback is safe.  Steps run one at a time
 * and synchronously in this mode (no speculation: every rank must take the same path through the step).  A checkpoint does not carry the mode
 * (nor a single world's tiling): set it before loading one — setting it drops the colour history, the checkpoint then brings its own. */
#define MI_SEAM_COLORS 24
#define MI_SHARD_SWEEP_FLOATS 8
typedef int (*mi_shard_sweep_fn)(void* user, mi_world* world, uint32_t sweep);
MI_API int mi_world_set_seam_tiling(mi_world* world, const mi_shard_desc* desc /* rank / num_ranks ignored; null: back to the plain schedule */);
MI_API int mi_world_shard_set_exact_seam(mi_world* world, uint32_t enable, mi_shard_sweep_fn exchange /* null with the library transport */, void* user);
MI_API int mi_world_shard_sweep_message_bytes(mi_world* world, uint64_t* out_bytes);
MI_API int mi_world_shard_export_sweep(mi_world* world, uint32_t slot, void* out_message);
MI_API int mi_world_shard_import_sweep(mi_world* world, const void* message);
/* Of the last internal step: manifolds in the seam class, colours they used, violations of the conditions above since the mode was set. */
MI_API int mi_world_seam_stats(mi_world* world, uint32_t* out_seam_manifolds, uint32_t* out_seam_colors, uint32_t* out_violations);

/* Development / tests: the library transport on ONE rank.  A one-rank communicator whose every neighbour is this rank itself, so the exchange runs
 * ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, the unpack kernels and the all-reduce with real records on one GPU: each message comes back to
 * its sender.  mi_debug_shard_peek_received copies the message last received in a slot (sweep_message != 0: the exact seam's sweep message).
 * No reference counterpart. */
MI_API int mi_debug_shard_attach_loopback(mi_world* world);
MI_API int mi_debug_shard_peek_received(mi_world* world, uint32_t slot, uint32_t sweep_message, void* out_message);

#ifdef __cplusplus
}
#endif
#endif /* MI_SHARD_H */
