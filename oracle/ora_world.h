// ora_world.h — TEST INFRASTRUCTURE (CPU oracle), see ora_math.h header note.
//
// Flat-array, uint32-index restatement of the reference's rigid-body step
// (src/physics/physics.cpp:1180-1413).  Parity status: PINNED to the reference's own code — oracle/refbuild/build_ref.py
// compiles the reference's physics-only source set into oracle/_ref/libref.so and tests/test_reference_pin.py steps the same
// scenes through its physicsStep and through this restatement (ORDER_REFERENCE): bit-identical poses, velocities, counts,
// contact lists, events and cloth particles.  (The reference ships no tests or golden vectors of its own.)  Also pinned by
// analytic known-answer tests (tests/test_oracle_*.py) and by its two independently ordered pipelines agreeing.
#pragma once
#include <vector>
#include <unordered_map>
#include <cstdint>
#include "ora_math.h"
#include "../include/mi_physics.h"
#include "../include/mi_constraints.h"
#include "../include/mi_shard.h"

namespace ora {

enum { T_SPHERE = 0, T_CAPSULE, T_CYLINDER, T_AABB, T_OBB, T_HULL, T_COUNT };

struct Material { float restitution, friction, density; };

struct HullGeometry {  // bounding_hull_geometry, src/physics/bounding_volumes.h:208-219
    std::vector<vec3> vertices;
    std::vector<uint32_t> tris;  // a,b,c per face
    vec3 aabbMin, aabbMax;
};

// collider_union (src/physics/physics.h:84-106) with named members instead of a C union.
struct Shape {
    int type = 0;
    vec3 a, b;          // sphere: a=center; capsule/cylinder: a,b; aabb: a=min,b=max; obb: a=center,b=radius; hull: a=position
    float radius = 0;   // sphere/capsule/cylinder
    quat rot;           // obb / hull
    uint32_t hull = 0;  // geometry index
};

struct Collider {  // collider_component
    Shape local;
    Material mat;
    uint32_t entity;
};

struct WorldCollider {  // one frame (physics.cpp:631-756)
    Shape s;
    Material mat;
    int objectType;
    uint32_t objectIndex;
};

struct AABB { vec3 mn, mx; };

struct Entity {
    vec3 position; quat rotation;       // transform_component
    int kind;
    int rb = -1;                        // rigid_body_component dense index or -1
    std::vector<uint32_t> colliders;    // newest first (linked list prepend, src/scene/scene.h:52-54)
    vec3 force;                         // force_field_component::force (kind MI_ENTITY_FORCE_FIELD)
    uint32_t kindIndex = 0;             // dense index among the entities of its kind (force fields / triggers)
};

struct Interaction { uint32_t rigidBodyIndex, otherIndex; int otherType; uint32_t rbCollider, otherCollider; };  // non_collision_interaction

struct RigidBody {  // rigid_body_component + physics_transform0/1 (src/physics/rigid_body.h:18-58)
    uint32_t entity;
    vec3 localCOG; float invMass; mat3 invInertia;
    float gravityFactor, linearDamping, angularDamping;
    vec3 linearVelocity, angularVelocity, forceAccumulator, torqueAccumulator;
    vec3 p0; quat r0;  // physics_transform0
    vec3 p1; quat r1;  // physics_transform1
    uint8_t shardKnown = 1;  // sharded world (include/mi_shard.h): this rank's copy of the body's state is current (it owned the body in the last step, or got a record for it)
};

struct GlobalState {  // rigid_body_global_state (src/physics/rigid_body.h:6-16)
    quat rotation; vec3 localCOG; vec3 position; mat3 invInertia; float invMass; vec3 linearVelocity; vec3 angularVelocity;
};

struct Contact {  // collision_contact (src/physics/physics.h:347-354)
    vec3 point; float penetrationDepth; vec3 normal; uint32_t friction_restitution;
};
struct Pair { uint32_t a, b; };

struct ContactManifold {  // src/physics/collision_narrow.cpp:40-46
    vec3 points[4]; float depths[4];
    vec3 normal; uint32_t numContacts;
};

struct CollisionConstraint {  // collision_constraint (src/physics/constraints.h:606-639)
    vec3 relGlobalAnchorA, relGlobalAnchorB, tangent;
    vec3 tangentImpulseToAngularVelocityA, tangentImpulseToAngularVelocityB;
    vec3 normalImpulseToAngularVelocityA, normalImpulseToAngularVelocityB;
    float impulseInNormalDir, impulseInTangentDir, effectiveMassInNormalDir, effectiveMassInTangentDir, bias;
};

// heightmap_collider_component + heightmap_collider_chunk (src/terrain/heightmap_collider.h:13-33, 126-151)
struct Heightmap {
    struct MinMax { uint16_t mn, mx; };
    struct Chunk { std::vector<uint16_t> heights; std::vector<std::vector<MinMax>> mips; };   // 129 x 129 heights or none
    uint32_t chunksPerDim; float chunkSize; Material material;
    vec3 minCorner; float invAmplitudeScale = 1.f, invChunkSize, chunkScale, heightScale = 0.f;
    std::vector<Chunk> chunks;
    Heightmap(uint32_t cpd, float size, Material m) : chunksPerDim(cpd), chunkSize(size), material(m), invChunkSize(1.f / size), chunkScale(size / 128.f), chunks((size_t)cpd * cpd) {}
    void setHeights(uint32_t x, uint32_t z, const uint16_t* heights);
    void update(vec3 minCorner, float amplitudeScale);
    float heightAt(float worldX, float worldZ) const;   // -FLT_MAX outside
};
// A heightmap contact is a one-contact manifold {collider, kHeightmapVirtualBase + j}: the virtual second index keeps pair
// keys, colouring priorities and the colour history unique (real collider indices stay below it).
static const uint32_t kHeightmapVirtualBase = (1u << 26) - 256u;

// cloth_component (src/physics/cloth.h:5-60)
struct Cloth {
    struct Constraint { uint32_t a, b; float restDistance, inverseMassSum; };
    float width, height; uint32_t gridSizeX, gridSizeY;
    float totalMass, gravityFactor, damping, stiffness, oldTotalMass, oldStiffness;
    std::vector<vec3> positions, prevPositions, velocities, forces;
    std::vector<float> invMasses;
    std::vector<Constraint> constraints;
    std::vector<uint32_t> colours, canonicalOrder;   // device order: colour-major (12 colours: family x parity)
    explicit Cloth(const mi_cloth_desc& d);
    void setFixedVertices(vec3 position, quat rotation, bool moveRigid);
    void applyWindForce(vec3 force);
    void recalculateProperties();
    void simulate(uint32_t velocityIterations, uint32_t positionIterations, uint32_t driftIterations, float dt, bool canonical);
};

struct SapEndpoint { float value; uint32_t creation; bool start; uint32_t colliderIndex; };

struct JointStore;  // ora_joints.cpp

struct World {
    std::vector<Entity> entities;
    std::vector<RigidBody> bodies;
    std::vector<Collider> colliders;  // creation order
    std::vector<HullGeometry> hulls;
    JointStore* joints = nullptr;

    // SAP context (src/physics/collision_broad.cpp:20-24)
    std::vector<SapEndpoint> endpoints;
    std::vector<uint32_t> startEndpoint, endEndpoint;  // per collider (creation index)
    uint32_t sortingAxis = 0;
    uint64_t axisSums[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // canonical mode: centre statistics of the last step (S1[3], S2lo[3], S2hi[3]) of the colliders this world counts

    std::vector<Pair> debugOrder; bool debugOrderPending = false, debugOrderError = false;   // ora_debug_set_solve_order (mirror of mi_debug_set_solve_order): the next step solves these manifolds in this order, joints in pool order
    int orderMode = 0;     // 0 = reference order (SAP sweep order, sequential PGS); 1 = canonical (GPU schedule replayed sequentially)
    bool dirtyProps = true;
    float timer = 0.f;

    // Last-step dumps
    std::vector<WorldCollider> wc;
    std::vector<AABB> aabbs;
    std::vector<Pair> bpPairs;            // broad-phase pairs in emission order
    std::vector<Pair> colliderPairs;      // manifolds
    std::vector<uint8_t> contactCounts;
    std::vector<Contact> contacts;
    std::vector<Pair> bodyPairs;          // per contact
    std::vector<uint32_t> manifoldColor;  // canonical mode
    std::vector<uint32_t> forceFieldEntities, triggerEntities;   // entity ids by dense index
    std::vector<Interaction> interactions;                        // last step
    std::vector<uint64_t> prevTriggerOverlaps;                    // sorted (triggerEntity << 32 | rbEntity)
    std::vector<Cloth*> cloths;
    uint32_t clothIterations[3] = {0, 1, 0};                      // velocity, position, drift (physics_settings defaults)
    Heightmap* heightmap = nullptr;                               // at most one per world
    uint32_t heightmapCollisions = 0, heightmapContacts = 0, heightmapManifolds = 0;     // last step: colliders touching the terrain, their contacts, the manifolds (of up to four) these form
    bool eventsEnabled = false;
    std::vector<uint64_t> prevCollisionKeys;   // sorted (creationA << 26 | creationB) of the previous step's manifolds
    std::vector<mi_event> events;              // since the last poll
    std::unordered_map<uint64_t, uint32_t> prevPairColor;   // canonical mode: colour of every manifold of the previous step, by (colliderA << 26 | colliderB)
    std::vector<GlobalState> rb;
    mi_step_counts counts{};
    // sharded world (include/mi_shard.h), canonical order only: this rank's tile, per-body activity (1 owned, 2 ghost, 0 elsewhere)
    struct Shard {
        bool enabled = false; mi_shard_desc desc{}; uint32_t myTile = 0, capacity = 0;
        std::vector<uint32_t> peers, peerRanks;
        std::vector<float> bordersX, bordersZ;        // interior tile borders (tiles - 1 per axis, ascending); uniform at enable, moved by ora_world_shard_set_borders
        std::vector<float> nextX, nextZ; bool bordersPending = false;   // set_borders: in force after the next step's exchange
        std::vector<uint8_t> active;
        std::vector<uint32_t> root;                   // lowest body index of the body's articulated island (itself without joints): the island is classified as one
        std::vector<std::vector<float>> sendBuf;      // one message per neighbour slot (record 0 = count)
        uint32_t owned[3] = {0, 0, 0};
    } shard;
    void shardClassify();
    void shardPack(const std::vector<vec3>& oldCog);
    // Exact seam (include/mi_shard.h): the contact schedule puts the SEAM manifolds (every dynamic body lies in the extended region of at least two
    // tiles) into the leading colours [0, MI_SEAM_COLORS) and everything else behind them.  A single world given the tiling orders its solve that way;
    // sharded worlds in exact mode do too, solve the seam manifolds redundantly and hand the owners' velocities of the shared bodies over after every sweep.
    struct Seam {
        bool tiling = false;                          // single world: virtual tiling (ora_world_set_seam_tiling)
        std::vector<float> bx, bz; float margin = 0.f;
        bool exact = false;                           // sharded world: exact mode (ora_world_shard_set_exact_seam)
        int (*fn)(void*, World*, uint32_t) = nullptr; void* user = nullptr;
        std::vector<vec3> cogStart;                   // island-root centres the step classified with
        std::vector<uint32_t> shared;                 // per body, this step: 0 = seen by its own tile only, else the border it is shared across (seamBorderOf)
        uint32_t manifolds = 0, colors = 0, violations = 0;
        int error = MI_OK;
    } seam;
    bool seamActive() const { return seam.tiling || (shard.enabled && seam.exact); }
    void seamClassify();

    World();
    ~World();
    void recalculateProperties();
    int destroyEntity(uint32_t entity);   // game_scene::deleteEntity (src/scene/scene.cpp:124-150)
    void stepInternal(const mi_step_settings& s, float dt);
    void step(const mi_step_settings& s, float dt);
};

// narrow phase (ora_narrow.cpp)
bool intersect(const World& w, const WorldCollider& A, const WorldCollider& B, ContactManifold& out);
bool overlapCheck(const World& w, const WorldCollider& A, const WorldCollider& B);   // boolean tests for triggers / force fields
vec3 closestPoint_PointSegment(vec3 q, vec3 la, vec3 lb);
float closestPoint_SegmentSegment(vec3 l1a, vec3 l1b, vec3 l2a, vec3 l2b, vec3& c1, vec3& c2);
bool rayVsCollider(const World& w, const Shape& localShape, vec3 localOrigin, vec3 localDirection, float& outT);   // ray::intersect*
void heightmapCollision(World& w);   // ora_heightmap.cpp: appends to colliderPairs / contactCounts / contacts / bodyPairs
// GJK / EPA (ora_gjk.cpp)
struct SupportShape { const Shape* s; const HullGeometry* g; };
vec3 support(const SupportShape& sh, vec3 dir);
struct GjkSimplexPoint { vec3 shapeAPoint, shapeBPoint, minkowski; };
struct GjkSimplex { GjkSimplexPoint a, b, c, d; uint32_t numPoints; };
bool gjkIntersectionTest(const SupportShape& A, const SupportShape& B, GjkSimplex& simplex);
struct EpaResult { vec3 point, normal; float penetrationDepth; };
int epaCollisionInfo(const GjkSimplex& simplex, const SupportShape& A, const SupportShape& B, EpaResult& out);

// joints (ora_joints.cpp)
JointStore* jointsCreate();
void jointsDestroy(JointStore*);
int jointsAdd(World& w, uint32_t type, uint32_t entityA, uint32_t entityB, const void* pod, uint32_t bytes, uint32_t* outId);
int jointsUpdate(World& w, uint32_t type, uint32_t id, const void* pod, uint32_t bytes);
int jointsGet(World& w, uint32_t type, uint32_t id, void* pod, uint32_t bytes);
int jointsDestroy(World& w, uint32_t type, uint32_t id);
void jointsDestroyAll(World& w);
int jointsDestroyOfEntity(World& w, uint32_t entity);
int jointsAddFromGlobal(World& w, uint32_t type, uint32_t ea, uint32_t eb, const float* anchor, const float* axis, float l0, float l1, uint32_t* outId);
void jointsInitialize(World& w, float dt);
void jointsMarkOrderDirty(World& w);
void jointsSolveIteration(World& w);
uint32_t jointsCount(const World& w);
void jointsRemapBody(World& w, uint32_t from, uint32_t to);
void jointsIslandRoots(const World& w, std::vector<uint32_t>& root);   // union-find over the joints' body pairs
int jointsLoadPods(World& w, const uint8_t*& p, const uint8_t* end, const uint32_t counts[6]);
void jointsSavePods(const World& w, std::vector<uint8_t>& out, uint32_t counts[6]);                 // ... appended to `out`, their numbers to `counts`   // checkpoint: PODs of all six types in pool order

void axisTerms(float centre, uint64_t out[3]);                 // canonical sweep-axis statistic: one centre coordinate -> (q, low 32 bits of q^2, the rest)
uint32_t axisFromSums(const uint64_t sums[9], uint32_t numColliders);
uint32_t hash32(uint32_t m);  // joint colouring priority
uint64_t pairPriority(uint32_t a, uint32_t b);  // contact-manifold colouring priority

}  // namespace ora
