// kernels_solve.hpp — the dataflow PGS solver: processTile, k_contact_solve_flow, the persistent XCD-partitioned kernel with resident rows, the one-lane kernel.
// Part of the ONE translation unit of the physics library (world.hip includes kernels.hpp, which includes the stage files in pipeline order).
#pragma once   // (included by kernels.hpp only, after the stage files before it)

namespace mi {

// ------------------------------------------------------------------------------------------------
// Dataflow PGS sweep: ONE launch per solver iteration instead of one per colour.
//
// A colour launch costs ~8.5 us whatever its size (boundary + two dependent memory round trips + one
// wave's arithmetic + drain), and a sweep needs ~8 of them back to back: the solver is bound by the
// number of serial phases, not by bytes.  Here every tile of the sweep is in flight at once and waits
// only for ITS OWN bodies: gVel[2b] = (v, tag), gVel[2b+1] = (w, tag) where tag counts the updates the
// body has received this step.  The manifolds of a body have distinct colours, so "the colours used on
// the body below mine" (k_contact_init) says how many updates precede this manifold in a sweep; lane
// waits until both halves of the body carry tag = iteration * degree + base, solves, and publishes
// (v, w) with tag + 1.  The execution order is therefore exactly the sequential colour-major order the
// oracle replays — only the waiting is per body instead of per colour.
//
// Cross-CU visibility (MI355X_MICROARCH.md, "inter-workgroup visibility"): each half is ONE 16-byte
// `sc1` (agent-scope, write-through) store carrying its own tag and is read with `sc1` loads (L1
// bypass), so a reader that sees the tag sees the data of the same store: no fences, no separate flag.
// Forward progress: tile t depends only on tiles < t (lower colours) and workgroups are dispatched in
// index order, so the lowest unfinished tile is always resident and never waits on an undispatched one;
// every wait is bounded anyway (spin budget -> StepScalars::solveError -> MI_ERR_DEVICE, no hang).
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef MI_SC_LOAD
#define MI_SC_LOAD " sc1"    // cache-policy bits of the granule loads / stores (development experiments override them)
#endif
#ifndef MI_SC_STORE
#define MI_SC_STORE " sc1"
#endif
constexpr uint32_t kSpinBudget = 1u << 16;
#ifndef MI_FLOW_WAVES
#define MI_FLOW_WAVES 1   // resident waves per SIMD the flow kernel is compiled for (measured: 1 = 0.86 ms, 2 = 1.02 ms per 20 sweeps at 262144 bodies)
#endif

// single 16-byte granule: issue only (the next waiting asm block lands it), load + wait, store
__device__ __forceinline__ void issueGranuleSc1(const float4* p, f32x4& g) { asm volatile("global_load_dwordx4 %0, %1, off" MI_SC_LOAD : "=&v"(g) : "v"(p) : "memory"); }
__device__ __forceinline__ void landed(f32x4& g) { asm volatile("" : "+v"(g)); }   // orders every later use of g behind the waiting block
// the same into a register that already holds a value ("+v": the asm reads AND writes g, so a value merged from a divergent branch stays in ONE register — with a pure output
// the compiler may place a copy between the load's issue and the wait that lands it, and copy the old contents)
__device__ __forceinline__ void issueGranuleSc1Keep(const float4* p, f32x4& g) { asm volatile("global_load_dwordx4 %0, %1, off" MI_SC_LOAD : "+v"(g) : "v"(p) : "memory"); }
__device__ __forceinline__ void loadGranuleSc1(const float4* p, f32x4& g) { asm volatile("global_load_dwordx4 %0, %1, off" MI_SC_LOAD "\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(p) : "memory"); }
__device__ __forceinline__ void storeGranuleSc1(float4* p, f32x4 g) { asm volatile("global_store_dwordx4 %0, %1, off" MI_SC_STORE : : "v"(p), "v"(g) : "memory"); }

// Lane pairs (2i, 2i+1) move one body per instruction: the even lane touches granule 0 and the odd lane granule 1 of the
// SAME body, i.e. one contiguous, 32-byte-aligned transaction instead of two scattered 16-byte ones (scattered
// write-through stores are what bounds this kernel: 4 per manifold per sweep).  Pass 0 serves the even lane's body,
// pass 1 the odd lane's; a DPP quad swap hands each lane the half its partner moved for it.
// lane 2i <-> lane 2i+1 as a DPP quad permute [1,0,3,2]: one VALU move, no trip through the LDS crossbar (ds_bpermute) — these
// exchanges sit between a tile's body loads and its stores, i.e. on the dependency chain.  Both lanes of a pair are always active together.
__device__ __forceinline__ uint32_t swz1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float swz1(float v) { return __uint_as_float(swz1(__float_as_uint(v))); }
__device__ __forceinline__ f32x4 swz1(f32x4 g) { f32x4 r = {swz1(g.x), swz1(g.y), swz1(g.z), swz1(g.w)}; return r; }
__device__ __forceinline__ float4* swz1(float4* p) {
    unsigned long long v = (unsigned long long)p;
    uint32_t lo = swz1((uint32_t)v), hi = swz1((uint32_t)(v >> 32));
    return (float4*)(((unsigned long long)hi << 32) | lo);
}
struct PairBody {   // addresses this lane touches in pass 0 / pass 1 for one body slot (A or B)
    float4* q0; float4* q1;
    __device__ __forceinline__ PairBody(float4* mine, bool odd) {
        float4* partner = swz1(mine);
        q0 = (odd ? partner : mine) + (odd ? 1 : 0);
        q1 = (odd ? mine : partner) + (odd ? 1 : 0);
    }
};
// after both passes landed: r0 / r1 = what this lane loaded in pass 0 / 1 -> this lane's own (g0, g1)
__device__ __forceinline__ void pairGather(bool odd, f32x4 r0, f32x4 r1, f32x4& g0, f32x4& g1);   // (below)
// The lane-pair exchange of pairGather and of the publish in ONE instruction per word:  y0 = even lane ? x0 : the partner's x1,  y1 = odd lane ? x1 : the partner's x0
// (v_cndmask_b32 whose first source is DPP quad-permuted [1,0,3,2]).  The compiler's own code for `odd ? swz1(a) : b` is v_mov_b32_dpp + v_cndmask_b32_e64 — gfx9 has no
// VOP3 DPP, and it keeps the lane parity in an SGPR pair, not in VCC — i.e. three instructions per word where pairGather / storePair* need both directions; these sit between
// a tile's bodies arriving and its publish, where every instruction is ~4 cycles of the dependency chain.  Both lanes of a pair are always active together.
__device__ __forceinline__ void pairExchange(const f32x4 x0, const f32x4 x1, f32x4& y0, f32x4& y1) {
    float a0, a1, a2, a3, b0, b1, b2, b3;
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\ts_nop 1\n\t"
                 "v_cndmask_b32_dpp %0, %12, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %1, %13, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %2, %14, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %3, %15, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "s_not_b64 vcc, vcc\n\t"
                 "v_cndmask_b32_dpp %4, %8, %12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %5, %9, %13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %6, %10, %14, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_cndmask_b32_dpp %7, %11, %15, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                 : "v"(x0.x), "v"(x0.y), "v"(x0.z), "v"(x0.w), "v"(x1.x), "v"(x1.y), "v"(x1.z), "v"(x1.w) : "vcc", "scc");
    y0.x = a0; y0.y = a1; y0.z = a2; y0.w = a3; y1.x = b0; y1.y = b1; y1.z = b2; y1.w = b3;
}
__device__ __forceinline__ void pairGather(bool odd, f32x4 r0, f32x4 r1, f32x4& g0, f32x4& g1) {
    (void)odd; pairExchange(r0, r1, g0, g1);
}
__device__ __forceinline__ void loadPair4Sc1(const PairBody& A, const PairBody& B, f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1) {
    asm volatile("global_load_dwordx4 %0, %4, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %1, %5, off" MI_SC_LOAD "\n\t"
                 "global_load_dwordx4 %2, %6, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %3, %7, off" MI_SC_LOAD "\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(A.q0), "v"(A.q1), "v"(B.q0), "v"(B.q1) : "memory");
}
__device__ __forceinline__ void issuePair4Sc1(const PairBody& A, const PairBody& B, f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1) {   // no wait: waitVmcnt + landed follow
    asm volatile("global_load_dwordx4 %0, %4, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %1, %5, off" MI_SC_LOAD "\n\t"
                 "global_load_dwordx4 %2, %6, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %3, %7, off" MI_SC_LOAD
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(A.q0), "v"(A.q1), "v"(B.q0), "v"(B.q1) : "memory");
}
__device__ __forceinline__ void issuePair2Sc1(const PairBody& A, f32x4& a0, f32x4& a1) {   // no wait ("+v": lanes that do not take part keep their values)
    asm volatile("global_load_dwordx4 %0, %2, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %1, %3, off" MI_SC_LOAD
                 : "+v"(a0), "+v"(a1) : "v"(A.q0), "v"(A.q1) : "memory");
}
__device__ __forceinline__ void loadPair2Sc1(const PairBody& A, f32x4& a0, f32x4& a1) {
    asm volatile("global_load_dwordx4 %0, %2, off" MI_SC_LOAD "\n\tglobal_load_dwordx4 %1, %3, off" MI_SC_LOAD "\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a0), "=&v"(a1) : "v"(A.q0), "v"(A.q1) : "memory");
}
// publish one body slot: h0 / h1 = this lane's own granules, need = this lane's body is written at all
__device__ __forceinline__ void storePairSc1(const PairBody& X, bool odd, bool need, f32x4 h0, f32x4 h1) {
    f32x4 recv = swz1(odd ? h0 : h1);                 // even lane receives the odd lane's g0, odd lane the even lane's g1
    bool partnerNeed = swz1(need ? 1u : 0u) != 0u;
    f32x4 d0 = odd ? recv : h0, d1 = odd ? h1 : recv;
    if (odd ? partnerNeed : need) storeGranuleSc1(X.q0, d0);
    if (odd ? need : partnerNeed) storeGranuleSc1(X.q1, d1);
}
// the same with a choice per body: XCD-local bodies are published with plain stores (they stay in this XCD's L2)
__device__ __forceinline__ void storeGranulePlain(float4* p, f32x4 g) { asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(g) : "memory"); }
__device__ __forceinline__ void storePairXcd(const PairBody& X, bool odd, bool need, bool local, f32x4 h0, f32x4 h1) {
    f32x4 recv = swz1(odd ? h0 : h1);
    bool partnerNeed = swz1(need ? 1u : 0u) != 0u, partnerLocal = swz1(local ? 1u : 0u) != 0u;
    f32x4 d0 = odd ? recv : h0, d1 = odd ? h1 : recv;
    const bool n0 = odd ? partnerNeed : need, l0 = odd ? partnerLocal : local;     // pass 0 moves the even lane's body,
    const bool n1 = odd ? need : partnerNeed, l1 = odd ? local : partnerLocal;     // pass 1 the odd lane's
    if (n0 && l0) storeGranulePlain(X.q0, d0);
    if (n0 && !l0) storeGranuleSc1(X.q0, d0);
    if (n1 && l1) storeGranulePlain(X.q1, d1);
    if (n1 && !l1) storeGranuleSc1(X.q1, d1);
}

// the four stores of one body slot under precomputed EXEC masks (plain / write-through for pass 0, then for pass 1); the wave is fully active on entry and on exit
__device__ __forceinline__ void storePairMasked(float4* q0, float4* q1, f32x4 d0, f32x4 d1, unsigned long long plain0, unsigned long long sc0, unsigned long long plain1, unsigned long long sc1_) {
    asm volatile("s_mov_b64 exec, %4\n\tglobal_store_dwordx4 %0, %2, off\n\t"
                 "s_mov_b64 exec, %5\n\tglobal_store_dwordx4 %0, %2, off" MI_SC_STORE "\n\t"
                 "s_mov_b64 exec, %6\n\tglobal_store_dwordx4 %1, %3, off\n\t"
                 "s_mov_b64 exec, %7\n\tglobal_store_dwordx4 %1, %3, off" MI_SC_STORE "\n\t"
                 "s_mov_b64 exec, -1"
                 : : "v"(q0), "v"(q1), "v"(d0), "v"(d1), "s"(plain0), "s"(sc0), "s"(plain1), "s"(sc1_) : "memory");
}
// LDSIMP: the accumulated impulses live in LDS (`ldsImp`, [k][lane]) because the same wave runs this tile in every sweep
// (k_contact_solve_persist); otherwise they travel between sweeps as tagged granules in `imp`.
#ifdef MI_DBG_KNOCKOUT
// development (knock-out harness, tools/gpu_knockout.sh): the host launches k_contact_solve_persist a SECOND time per step on scratch copies of the velocity arrays with parts
// of a tile visit removed, to price them: bit 0 = no row stream at all (nothing is prefetched; the update runs on whatever the registers hold — the tag protocol does not
// depend on the values), bit 1 = every tile's rows come from contact-tile 0 (the same loads in the queue, served by the L2), bit 2 = no waiting for tags.
#define MI_KNOCK(bit) ((g_dbgKnockLocal >> (bit)) & 1u)
#else
#define MI_KNOCK(bit) 0u
#endif
#ifdef MI_DBG_TIMELINE
__device__ unsigned long long* g_dbgTimeline = nullptr;   // development: [wave][visit][8] wall-clock stamps of k_contact_solve_persist
__device__ __forceinline__ void dbgStamp(unsigned long long* rec, int i) { if (rec && threadIdx.x == 0) rec[i] = wall_clock64(); }
#define MI_STAMP(rec, i) dbgStamp(rec, i)
#else
#define MI_STAMP(rec, i) ((void)0)
#endif
// Hook of processTile: early() runs right after the body loads were issued and returns how many loads it issued itself (they
// may stay in flight across the first tag check); late(waited) runs once the tags are satisfied, waited = the tile had to poll.
#ifdef MI_EXP_PIN_FLOW
#define MI_NOHOOK_PIN true    // (tools/exp/pinned_pairs_repro.sh: pinned register pairs in the dispatch-ordered kernels too — WRONG results there, round 5)
#else
#define MI_NOHOOK_PIN false
#endif
struct NoHook { enum : bool { kPinRows = MI_NOHOOK_PIN }; unsigned long long* rec = nullptr; __device__ __forceinline__ uint32_t early() const { return 0u; } __device__ __forceinline__ void late(bool) const {}
                __device__ __forceinline__ bool knockNoWait() const { return false; } };
// wait until at most n of the newest vector-memory operations are outstanding (n = a count the caller issued itself)
__device__ __forceinline__ void waitVmcnt(uint32_t n) {
    switch (n) {
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
template <int CNT, bool LDSIMP, bool XCD = false, class Hook = NoHook>
__device__ __forceinline__ void processTile(uint32_t ctBase, uint32_t lane, uint32_t it, const uint4 meta, const float4 nf, const float2 mass, const ContactRows* c,
                                            float4* imp, float4* gVel, StepScalars* sc, float2* ldsImp, float4* gVelL = nullptr, Hook hook = Hook());
template <int CNT, bool LDSIMP = false>
__device__ __forceinline__ void flowTile(uint32_t tile, uint32_t ctBase, uint32_t lane, uint32_t it, const uint4* __restrict__ slotMeta,
                                         const float4* __restrict__ slotNormal, const float2* __restrict__ slotMass,
                                         const float4* __restrict__ rows, float4* imp, float4* gVel, StepScalars* sc, float2* ldsImp = nullptr) {
    const uint4 meta = slotMeta[(size_t)tile * 64u + lane];
    if (__ballot(meta.w != 0u) == 0ull) return;   // nothing of this tile is the tile solver's (manifolds of private joint islands: their island's workgroup solves them)
    const float4 nf = slotNormal[(size_t)tile * 64u + lane];
    const float2 mass = slotMass[(size_t)tile * 64u + lane];
    ContactRows c[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        const float4* __restrict__ row = rows + ((size_t)ctBase + k) * (kRows * 64u) + lane;
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) c[k].r[r] = row[r * 64u];
    }
    processTile<CNT, LDSIMP>(ctBase, lane, it, meta, nf, mass, c, imp, gVel, sc, ldsImp);
}
// The tile proper, from data already requested (flowTile) or prefetched (k_contact_solve_persist): wait for the bodies (and the
// impulse granules), solve, publish.
// The tile proper.  What shapes it is how few instructions sit between the
// arrival of a tile's bodies and its publish, the part of a visit that is on the dependency chain between tiles (~4 cycles per instruction at one wave per SIMD):
//   * which of the lane pair's four stores per body slot take place, and with which cache policy, is known from the slot's constants: four EXEC masks per body are
//     computed BEFORE the wait and the publish is four stores under `s_mov_b64 exec, mask` (was: the predicates recomputed and exchanged after the solve, ~12 per store);
//   * the lane-pair exchange is pairExchange (one v_cndmask_b32_dpp per word and direction), for the arriving bodies and for the publish;
//   * ONE gather / tag check site: every poll round gathers all lanes from the raw load registers (lanes that did not poll again find their old words there),
//     so the bodies the solve starts from are defined in one place and no copies merge two definitions at the loop's exit.
// The whole wave is active here (processTile is only reached through wave-uniform control flow).
template <int CNT, bool LDSIMP, bool XCD, class Hook>
__device__ __forceinline__ void processTile(uint32_t ctBase, uint32_t lane, uint32_t it, const uint4 meta, const float4 nf, const float2 mass, const ContactRows* c,
                                            float4* imp, float4* gVel, StepScalars* sc, float2* ldsImp, float4* gVelL, Hook hook) {
    const uint32_t bA = meta.x, bB = meta.y, pk = meta.z;
    const float imA = mass.x, imB = mass.y;
    const bool valid = meta.w != 0u;
#ifdef MI_DBG_ALLLOCAL
    const bool locA = XCD, locB = XCD;
#else
    const bool locA = XCD && (meta.w & 0x100u) != 0u, locB = XCD && (meta.w & 0x200u) != 0u;
#endif
    const bool live = valid && (imA != 0.f || imB != 0.f);
    const uint32_t degA = (pk >> 7) & 127u, degB = (pk >> 21) & 127u;
    const uint32_t expA = it * degA + (pk & 127u), expB = it * degB + ((pk >> 14) & 127u);
    const bool needA = valid && degA != 0u, needB = valid && degB != 0u;
    float4* pA = (locA ? gVelL : gVel) + 2 * (size_t)bA; float4* pB = (locB ? gVelL : gVel) + 2 * (size_t)bB;
    float4* pI = imp + (size_t)ctBase * 64u + lane;
    f32x4 ig[CNT], a0, a1, b0, b1;
    const bool odd = (lane & 1u) != 0u;
    const PairBody PA(pA, odd), PB(pB, odd);
    if (!LDSIMP) {
#pragma unroll
        for (int k = 0; k < CNT; ++k) issueGranuleSc1(pI + (size_t)k * 64u, ig[k]);
    }
    f32x4 ra0, ra1, rb0, rb1;   // raw load destinations (pass 0 / pass 1 of bodies A and B)
    issuePair4Sc1(PA, PB, ra0, ra1, rb0, rb1);
    MI_STAMP(hook.rec, 2);
    // (while the body loads are in flight) EXEC masks of the publish (pass 0 moves the even lane's body, pass 1 the odd lane's; XCD-local bodies are published with plain stores, the others write-through)
    unsigned long long mA[4], mB[4];
    {
        const unsigned long long E = 0x5555555555555555ull;
        const unsigned long long nA = __ballot(needA), lA = __ballot(locA), nB = __ballot(needB), lB = __ballot(locB);
        const unsigned long long nA0 = (nA & E) | ((nA & E) << 1), nA1 = (nA & ~E) | ((nA & ~E) >> 1), lA0 = (lA & E) | ((lA & E) << 1), lA1 = (lA & ~E) | ((lA & ~E) >> 1);
        const unsigned long long nB0 = (nB & E) | ((nB & E) << 1), nB1 = (nB & ~E) | ((nB & ~E) >> 1), lB0 = (lB & E) | ((lB & E) << 1), lB1 = (lB & ~E) | ((lB & ~E) >> 1);
        mA[0] = nA0 & lA0; mA[1] = nA0 & ~lA0; mA[2] = nA1 & lA1; mA[3] = nA1 & ~lA1;
        mB[0] = nB0 & lB0; mB[1] = nB0 & ~lB0; mB[2] = nB1 & lB1; mB[3] = nB1 & ~lB1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { asm volatile("" : "+s"(mA[k])); asm volatile("" : "+s"(mB[k])); }   // (pinned here: not recomputed behind the wait)
    }
    const uint32_t hookLoads = hook.early();   // (the persistent kernel: this tile's rows out of the prefetch registers, the next tile's requested)
    PkRows pkr[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) pkr[k] = packRows<std::remove_reference_t<Hook>::kPinRows>(c[k], nf, k != 0 && (meta.w & kMetaPerContactNormal) != 0u);   // (contact 0: the slot's normal IS its own)
    waitVmcnt(hookLoads);   // the hook's loads are younger than the body loads: they may stay in flight
    float2 imIn[CNT];   // accumulated impulses this tile starts from
    if (!LDSIMP) {
#pragma unroll
        for (int k = 0; k < CNT; ++k) landed(ig[k]);
    } else {
#pragma unroll
        for (int k = 0; k < CNT; ++k) imIn[k] = ldsImp[k * 64 + lane];
    }
    bool okA, okB, okI = true, polled = false;
    uint32_t budget = kSpinBudget;
    MI_STAMP(hook.rec, 3);
    for (;;) {
        landed(ra0); landed(ra1); landed(rb0); landed(rb1);
        pairExchange(ra0, ra1, a0, a1);
        pairExchange(rb0, rb1, b0, b1);
        okA = !needA || (__float_as_uint(a0.w) == expA && __float_as_uint(a1.w) == expA);
        okB = !needB || (__float_as_uint(b0.w) == expB && __float_as_uint(b1.w) == expB);
        if (!LDSIMP) {
            okI = true;
#pragma unroll
            for (int k = 0; k < CNT; ++k) okI = okI && (!live || __float_as_uint(ig[k].z) == it);
        }
#ifdef MI_DBG_KNOCKOUT
        if (hook.knockNoWait()) okA = okB = okI = true;   // development knock-out: timing without dependency waits (results are garbage)
#endif
        if (__ballot(!(okA && okB && okI)) == 0ull) break;
        polled = true;
        if (--budget == 0u) { sc->solveError = 1u; break; }
        // both lanes of a pair poll together (the exchange above needs both); tight polling measured fastest: only the pairs still waiting re-load, both bodies' polls
        // in flight together (one round trip per round, not two)
        const uint32_t partnerOkA = swz1(okA ? 1u : 0u), partnerOkB = swz1(okB ? 1u : 0u);
        const bool pollA = !okA || partnerOkA == 0u, pollB = !okB || partnerOkB == 0u;
        if (pollA) issuePair2Sc1(PA, ra0, ra1);
        if (pollB) issuePair2Sc1(PB, rb0, rb1);
        if (!LDSIMP && !okI) {
#pragma unroll
            for (int k = 0; k < CNT; ++k) issueGranuleSc1Keep(pI + (size_t)k * 64u, ig[k]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!LDSIMP) {
#pragma unroll
            for (int k = 0; k < CNT; ++k) landed(ig[k]);
        }
    }
    hook.late(polled);
    MI_STAMP(hook.rec, 4);
    P3 pv, pw;
    pv.x = pk2(a0.x, b0.x); pv.y = pk2(a0.y, b0.y); pv.z = pk2(a0.z, b0.z);
    pw.x = pk2(a1.x, b1.x); pw.y = pk2(a1.y, b1.y); pw.z = pk2(a1.z, b1.z);
    const f32x2 sMass = pk2(-imA, imB);
    float2 out[CNT];
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        float2 im = LDSIMP ? imIn[k] : make_float2(ig[k].x, ig[k].y);
        solveOnePkRows(pkr[k], nf, im, sMass, pv, pw);
        out[k] = im;
    }
    // publish: bodies first (they are on the dependency chain), then the impulses; nothing to wait for afterwards
    {
        const float tA = __uint_as_float(expA + 1u), tB = __uint_as_float(expB + 1u);
        const f32x4 hA0 = {pv.x.x, pv.y.x, pv.z.x, tA}, hA1 = {pw.x.x, pw.y.x, pw.z.x, tA}, hB0 = {pv.x.y, pv.y.y, pv.z.y, tB}, hB1 = {pw.x.y, pw.y.y, pw.z.y, tB};
        MI_STAMP(hook.rec, 5);
        f32x4 dA0, dA1, dB0, dB1;
        pairExchange(hA0, hA1, dA0, dA1);
        pairExchange(hB0, hB1, dB0, dB1);
        storePairMasked(PA.q0, PA.q1, dA0, dA1, mA[0], mA[1], mA[2], mA[3]);
        storePairMasked(PB.q0, PB.q1, dB0, dB1, mB[0], mB[1], mB[2], mB[3]);
    }
    if (LDSIMP) {
#pragma unroll
        for (int k = 0; k < CNT; ++k) ldsImp[k * 64 + lane] = out[k];
    } else if (live) {
        float t = __uint_as_float(it + 1u);
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            f32x4 g = {out[k].x, out[k].y, t, 0.f};
            storeGranuleSc1(pI + (size_t)k * 64u, g);
        }
    }
}

// Block b runs sweep itBase + b / numTiles of tile b % numTiles (numTiles = StepScalars::totalTiles; schedule order, colour-major): with no joints between the
// sweeps ALL iterations are one launch, so the latency-bound small colours of sweep i overlap the bandwidth-bound large
// colours of sweep i + 1.  tileDesc[tile] = (first contact-tile, contacts per manifold).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, MI_FLOW_WAVES))) void k_contact_solve_flow(
    uint32_t itBase, uint32_t sweeps, const uint2* __restrict__ tileDesc, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal,
    const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* imp, float4* gVel, StepScalars* sc) {
    const uint32_t numTiles = sc->totalTiles;   // the grid is sized from an upper bound: surplus workgroups (all at the end) exit
    if (stepIsVoid(sc)) return;
    if (blockIdx.x >= numTiles * sweeps) return;
    const uint32_t it = itBase + blockIdx.x / numTiles, tile = blockIdx.x % numTiles, lane = threadIdx.x;
    const uint2 d = tileDesc[tile];
    switch (d.y) {
        case 1: flowTile<1>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, gVel, sc); break;
        case 2: flowTile<2>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, gVel, sc); break;
        case 3: flowTile<3>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, gVel, sc); break;
        default: flowTile<4>(tile, d.x, lane, it, slotMeta, slotNormal, slotMass, rows, imp, gVel, sc); break;
    }
}

// Persistent variant: `numWaves` workgroups (one per SIMD of the chip, all resident at once), workgroup w owns the tiles
// w, w + numWaves, ... in EVERY sweep and walks them in schedule order, sweep after sweep.  Because a tile never changes
// hands, its accumulated impulses stay in LDS: no impulse granules are read, polled or written (one 16-byte write-through
// store and one tagged load less per contact and sweep; 13 % fewer bytes).  Dependencies between tiles are the body tags as
// before.  Forward progress: every wave runs its tiles in ascending (sweep, tile) order and a tile only waits for smaller
// (sweep, tile) pairs, so the wave owning the smallest unfinished pair is never blocked — provided all workgroups are
// resident, which the host guarantees by launching at most one per SIMD (waits are bounded by the spin budget regardless).
#ifndef MI_PERSIST_WPE
#define MI_PERSIST_WPE 1
#endif
// a 16-byte load into four FIXED accumulator registers / reading them back (k_contact_solve_persist's row prefetch)
#define MI_ACC_LOAD(A0, A1, A2, A3, addr) asm volatile("global_load_dwordx4 a[" #A0 ":" #A3 "], %0, off" : : "v"(addr) : "memory", "a" #A0, "a" #A1, "a" #A2, "a" #A3)
#define MI_ACC_READ(dst, A0, A1, A2, A3) do { float x_, y_, z_, w_; \
    asm volatile("v_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 "\n\tv_accvgpr_read_b32 %2, a" #A2 "\n\tv_accvgpr_read_b32 %3, a" #A3 \
                 : "=v"(x_), "=v"(y_), "=v"(z_), "=v"(w_)); (dst) = make_float4(x_, y_, z_, w_); } while (0)

// RESIDENT ROWS (round 6).  The knock-out harness prices the row stream of this kernel at 14 % of its launch (no stream at all: 471 -> 405 us at the bench state;
// profiles/r06_knockout_solver_and_emit.txt): every wave's 24 x 1 KB row loads per tile sit in the same in-order memory queue as its body polls and publish stores.
// The accumulator registers a0..a143 are free (the ring is a160..a255), i.e. six contact-tiles ("positions") of 24 registers: the first tiles of a wave's list whose contacts fit
// are RESIDENT there — loaded once, at the top of the launch — and "prefetching" such a tile is 24 register moves per contact (v_accvgpr_mov_b32) into the ring,
// issued where the loads would have been: nothing enters the memory queue, and processTile still takes every tile's rows out of the ring, unchanged.
// Register numbers are literals ("n" operands): position q = a[24 q .. 24 q + 23], ring contact k = a[160 + 24 k .. ].  The compiler itself never allocates an
// accumulator register in this kernel (no spills: tests/test_capi_symbols.py reads the ISA); the ring's clobber lists make the descriptor cover a0..a255.
// Positions 6 and 7 live in the ARCHITECTURAL registers v208..v255 of the variants that fit into 208 allocatable VGPRs (slot data in LDS: amdgpu_num_vgpr keeps the
// compiler out of v208 and above): the same scheme with v_accvgpr_write_b32 as the move.  (Three such positions behind a cap of 184 were measured at 444 -> 433 us; the
// per-contact normals of terrain manifolds then took the nine registers that made 184 enough.)  The variants that need more registers keep six.
constexpr uint32_t kResidentAcc = 6, kResidentVgprBase = 208;
template <int DST, int SRC> __device__ __forceinline__ void accMov() { asm volatile("v_accvgpr_mov_b32 a[%0], a[%1]" : : "n"(DST), "n"(SRC)); }
template <int DST, int SRC> __device__ __forceinline__ void accFromV() { asm volatile("v_accvgpr_write_b32 a[%0], v[%1]" : : "n"(DST), "n"(SRC)); }
template <int Q, int K, int... R> __device__ __forceinline__ void accCopyContactImpl(std::integer_sequence<int, R...>) {
    if constexpr (Q < (int)kResidentAcc) (accMov<160 + 24 * K + R, 24 * Q + R>(), ...);
    else (accFromV<160 + 24 * K + R, (int)kResidentVgprBase + 24 * (Q - (int)kResidentAcc) + R>(), ...);
}
template <int NPOS, int Q, int K> __device__ __forceinline__ void accCopyContact() { if constexpr (Q < NPOS) accCopyContactImpl<Q, K>(std::make_integer_sequence<int, 24>()); }
template <int LO> __device__ __forceinline__ void accLoad4(const float4* p) { asm volatile("global_load_dwordx4 a[%0:%1], %2, off" : : "n"(LO), "n"(LO + 3), "v"(p) : "memory"); }
template <int LO> __device__ __forceinline__ void vgprLoad4(const float4* p) { asm volatile("global_load_dwordx4 v[%0:%1], %2, off" : : "n"(LO), "n"(LO + 3), "v"(p) : "memory", "v255"); }
template <int NPOS, int Q> __device__ __forceinline__ void accLoadContact(const float4* row) {   // the six rows of one contact of a tile -> position Q
    if constexpr (Q < (int)kResidentAcc) {
        accLoad4<24 * Q + 0>(row + 0u * 64u); accLoad4<24 * Q + 4>(row + 1u * 64u); accLoad4<24 * Q + 8>(row + 2u * 64u);
        accLoad4<24 * Q + 12>(row + 3u * 64u); accLoad4<24 * Q + 16>(row + 4u * 64u); accLoad4<24 * Q + 20>(row + 5u * 64u);
    } else if constexpr (Q < NPOS) {
        constexpr int B = (int)kResidentVgprBase + 24 * (Q - (int)kResidentAcc);
        vgprLoad4<B + 0>(row + 0u * 64u); vgprLoad4<B + 4>(row + 1u * 64u); vgprLoad4<B + 8>(row + 2u * 64u);
        vgprLoad4<B + 12>(row + 3u * 64u); vgprLoad4<B + 16>(row + 4u * 64u); vgprLoad4<B + 20>(row + 5u * 64u);
    }
}
template <int NPOS> __device__ __forceinline__ void accLoadResident(uint32_t q, const float4* row) {
    switch (q) {
        case 0: accLoadContact<NPOS, 0>(row); break; case 1: accLoadContact<NPOS, 1>(row); break; case 2: accLoadContact<NPOS, 2>(row); break;
        case 3: accLoadContact<NPOS, 3>(row); break; case 4: accLoadContact<NPOS, 4>(row); break; case 5: accLoadContact<NPOS, 5>(row); break;
        case 6: accLoadContact<NPOS, 6>(row); break; case 7: accLoadContact<NPOS, 7>(row); break; default: accLoadContact<NPOS, 8>(row); break;
    }
}
template <int NPOS, int Q> __device__ __forceinline__ void accCopyTile(uint32_t cnt) {   // resident positions Q .. Q + cnt - 1 -> ring contacts 0 .. cnt - 1
    accCopyContact<NPOS, Q, 0>();
    if (1u < cnt) accCopyContact<NPOS, Q + 1, 1>();
    if (2u < cnt) accCopyContact<NPOS, Q + 2, 2>();
    if (3u < cnt) accCopyContact<NPOS, Q + 3, 3>();
}
template <int NPOS> __device__ __forceinline__ void accCopyResident(uint32_t q, uint32_t cnt) {
    switch (q) {
        case 0: accCopyTile<NPOS, 0>(cnt); break; case 1: accCopyTile<NPOS, 1>(cnt); break; case 2: accCopyTile<NPOS, 2>(cnt); break;
        case 3: accCopyTile<NPOS, 3>(cnt); break; case 4: accCopyTile<NPOS, 4>(cnt); break; case 5: accCopyTile<NPOS, 5>(cnt); break;
        case 6: accCopyTile<NPOS, 6>(cnt); break; case 7: accCopyTile<NPOS, 7>(cnt); break; default: accCopyTile<NPOS, 8>(cnt); break;
    }
}

// METALDS = false (larger problems): only the impulses live in LDS (2060 B per slot instead of 4620); the constant slot data is
// prefetched from global memory together with the rows of the next tile.
// XCD = true (XCD-partitioned): workgroup w belongs to XCD w % 8 (verified against the hardware id: anything else is
// reported as solveError 3 and the host falls back) and owns entries w / 8, w / 8 + gridDim / 8, ... of THAT XCD's tile
// list (xcdTiles, ascending = schedule order).  Bodies only this XCD touches (bodyOwner has exactly this XCD's bit) are
// handed over through the XCD's L2 in the cached array gVelL; all others through memory in gVel as before.
// IMPLDS = false (piles beyond ~1.2 M manifolds): nothing per slot but a 20-byte descriptor stays in LDS; the accumulated
// impulses travel as tagged granules in `imp` exactly as in k_contact_solve_flow (no size limit left).
#define MI_PERSIST_PARAMS uint32_t sweeps, uint32_t maxSlots, const uint2* __restrict__ tileDesc, const uint4* slotMeta, const float4* __restrict__ slotNormal, \
    const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* gVel, StepScalars* sc, uint32_t xcdOnly, \
    const uint32_t* __restrict__ xcdTiles, uint32_t listCap, const unsigned long long* __restrict__ bodyOwner, float4* gVelL, uint4* slotMetaW, float4* imp, uint32_t xcdFault, \
    uint32_t resident /* 1: rows of the first tiles stay in a0..a143 (and v208..v255) */
#define MI_PERSIST_PASS sweeps, maxSlots, tileDesc, slotMeta, slotNormal, slotMass, rows, gVel, sc, xcdOnly, xcdTiles, listCap, bodyOwner, gVelL, slotMetaW, imp, xcdFault, resident
template <bool METALDS, bool XCD, bool IMPLDS>
__device__ __forceinline__ void persistSolveBody(MI_PERSIST_PARAMS) {
    // LDS per workgroup: [maxSlots] x { meta uint4[64], normal float4[64], mass float2[64] } (constant over the sweeps; METALDS only), then the
    // impulses float2[4 * maxSlots][64], then the per-slot (first contact-tile, contacts per manifold, impulse offset)
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
    const size_t metaSlots = METALDS ? (size_t)maxSlots : 0;
    uint4* lMeta = reinterpret_cast<uint4*>(ldsRaw);
    float4* lNormal = reinterpret_cast<float4*>(lMeta + metaSlots * 64u);
    float2* lMass = reinterpret_cast<float2*>(lNormal + metaSlots * 64u);
    float2* lImp = lMass + metaSlots * 64u;
    uint32_t* lDesc = reinterpret_cast<uint32_t*>(lImp + (IMPLDS ? (size_t)maxSlots * 4u * 64u : 0));   // [maxSlots][3]
    if (xcdOnly && (blockIdx.x & 7u) != 0u) return;   // development experiment: only the workgroups of one XCD work
    const uint32_t lane = threadIdx.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const bool isVoid = stepIsVoid(sc);   // (a speculative step already known to be void: no tiles, the workgroup leaves — kernels_common.hpp; loaded beside the tile counts, no round trip of its own)
    uint32_t numTiles = isVoid ? 0u : sc->totalTiles, numWaves = xcdOnly ? gridDim.x / 8u : gridDim.x, wid = xcdOnly ? blockIdx.x / 8u : blockIdx.x;
    if (XCD) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
        hw &= 15u;
        uint32_t seen = 0u;
        if (lane == 0) { seen = atomicCAS(&sc->xccOf[xcd], 0xFFFFFFFFu, hw); if (seen == 0xFFFFFFFFu) seen = hw; }
        seen = (uint32_t)__shfl((int)seen, 0, 64);
        if (seen != hw || (xcdFault && blockIdx.x == 9u)) { if (lane == 0) sc->solveError = 3u; return; }   // blockIdx % 8 does not identify the XCD on this device (xcdFault: test injection)
        numTiles = (sc->totalTiles && !isVoid) ? sc->xcdCount[xcd] : 0u; numWaves = gridDim.x / 8u; wid = blockIdx.x / 8u;
        xcdTiles += (size_t)xcd * listCap;
        if (numTiles > listCap) { if (lane == 0) sc->solveError = 2u; return; }
    }
    uint32_t* lTile = reinterpret_cast<uint32_t*>(lDesc + 3u * (size_t)maxSlots);   // [maxSlots] tile of every slot
    uint32_t* lCrit = lTile + maxSlots;                                             // [maxSlots] 1: the slot had to poll in the previous sweep
    constexpr int kResidentPositions = METALDS ? 8 : 6;
    uint32_t* lRes = lCrit + maxSlots;                                              // [maxSlots] first resident position of the slot's rows, or 0xFF: they stream
    uint32_t mySlots = 0, off = 0, resNext = 0;
    for (uint32_t li = wid; li < numTiles && mySlots < maxSlots; li += numWaves, ++mySlots) {
        const uint32_t tile = XCD ? xcdTiles[li] : li;
        const uint2 d = tileDesc[tile];
        const bool res = resident && resNext + d.y <= kResidentPositions;
        if (lane == 0) { lTile[mySlots] = tile; lCrit[mySlots] = 0u; lRes[mySlots] = res ? resNext : 0xFFu; }
        if (res) {   // (issued here, landed by the vmcnt(0) behind the loop)
            for (uint32_t k = 0; k < d.y; ++k) accLoadResident<kResidentPositions>(resNext + k, rows + ((size_t)d.x + k) * (kRows * 64u) + lane);
            resNext += d.y;
        }
        if (XCD) {   // which of this slot's two bodies are XCD-local -> bits 8 / 9 of meta.w (read back from LDS or global below)
            uint4 m = slotMeta[(size_t)tile * 64u + lane];
            const unsigned long long mine = 1ull << (8u * xcd);
            if (m.w != 0u) m.w |= (bodyOwner[m.x] == mine ? 0x100u : 0u) | (bodyOwner[m.y] == mine ? 0x200u : 0u);
            if (METALDS) lMeta[mySlots * 64u + lane] = m; else slotMetaW[(size_t)tile * 64u + lane] = m;
        }
        if (METALDS) {
            if (!XCD) lMeta[mySlots * 64u + lane] = slotMeta[(size_t)tile * 64u + lane];
            lNormal[mySlots * 64u + lane] = slotNormal[(size_t)tile * 64u + lane];
            lMass[mySlots * 64u + lane] = slotMass[(size_t)tile * 64u + lane];
        }
        if (lane == 0) { lDesc[3 * mySlots] = d.x; lDesc[3 * mySlots + 1] = d.y; lDesc[3 * mySlots + 2] = off; }
        if (IMPLDS) for (uint32_t k = 0; k < d.y; ++k) lImp[(size_t)(off + k) * 64u + lane] = make_float2(0.f, 0.f);
        off += d.y;
    }
    if (wid + (size_t)mySlots * numWaves < numTiles) { if (lane == 0) sc->solveError = 2u; return; }   // more tiles than the host sized LDS for
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!mySlots) return;
    // Software pipeline over (sweep, slot): the rows of the NEXT tile are requested while this tile waits for its bodies.
    // Loads retire in order, so the request order matters: this tile's body loads go first (even before its own rows are
    // taken out of the ACC registers: vmcnt(4)), the prefetch second, and the first tag check waits with vmcnt(number of
    // prefetch loads) — the bodies are back, the prefetch may still be in flight.
    // The prefetch is inline asm with its exact instruction count known, into FIXED accumulator registers a160..a255 that
    // the compiler never allocates (tests/test_capi_symbols.py checks the ISA for that); they are read back, again by
    // inline asm, after the explicit vmcnt(0) at the top of the next iteration.  (Compiler-allocated registers do not
    // work here: the allocator copies in-flight values around at loop boundaries.)
    uint4 nxMeta = make_uint4(0u, 0u, 0u, 0u); float4 nxNf = make_float4(0.f, 0.f, 0.f, 0.f); float2 nxMass = make_float2(0.f, 0.f);
#ifdef MI_DBG_KNOCKOUT
    const uint32_t g_dbgKnockLocal = __builtin_amdgcn_readfirstlane(g_dbgKnock);
#endif
    auto fetchRows = [&](uint32_t slot) __attribute__((always_inline)) -> uint32_t {
        uint32_t ct = lDesc[3 * slot]; const uint32_t cnt = lDesc[3 * slot + 1];
        if (MI_KNOCK(1)) ct = 0u;
        if (!METALDS) {
            const size_t at = (size_t)lTile[slot] * 64u + lane;
            nxMeta = XCD ? slotMetaW[at] : slotMeta[at]; nxNf = slotNormal[at]; nxMass = slotMass[at];
        }
        if (MI_KNOCK(0)) return 0u;
        if (const uint32_t q = lRes[slot]; q != 0xFFu) { accCopyResident<kResidentPositions>(q, cnt); return (METALDS || !IMPLDS) ? 0u : 3u; }   // resident: register moves, nothing enters the memory queue
                                                                                                                                   // (slot data not in LDS: its three loads, just issued, may stay in flight)
        {
            const float4* row = rows + (size_t)ct * (kRows * 64u) + lane;   // row (k, r) of the tile at + (k * kRows + r) * 64
            if (0u < cnt) MI_ACC_LOAD(160, 161, 162, 163, row + 0u * 64u);
            if (0u < cnt) MI_ACC_LOAD(164, 165, 166, 167, row + 1u * 64u);
            if (0u < cnt) MI_ACC_LOAD(168, 169, 170, 171, row + 2u * 64u);
            if (0u < cnt) MI_ACC_LOAD(172, 173, 174, 175, row + 3u * 64u);
            if (0u < cnt) MI_ACC_LOAD(176, 177, 178, 179, row + 4u * 64u);
            if (0u < cnt) MI_ACC_LOAD(180, 181, 182, 183, row + 5u * 64u);
            if (1u < cnt) MI_ACC_LOAD(184, 185, 186, 187, row + 6u * 64u);
            if (1u < cnt) MI_ACC_LOAD(188, 189, 190, 191, row + 7u * 64u);
            if (1u < cnt) MI_ACC_LOAD(192, 193, 194, 195, row + 8u * 64u);
            if (1u < cnt) MI_ACC_LOAD(196, 197, 198, 199, row + 9u * 64u);
            if (1u < cnt) MI_ACC_LOAD(200, 201, 202, 203, row + 10u * 64u);
            if (1u < cnt) MI_ACC_LOAD(204, 205, 206, 207, row + 11u * 64u);
            if (2u < cnt) MI_ACC_LOAD(208, 209, 210, 211, row + 12u * 64u);
            if (2u < cnt) MI_ACC_LOAD(212, 213, 214, 215, row + 13u * 64u);
            if (2u < cnt) MI_ACC_LOAD(216, 217, 218, 219, row + 14u * 64u);
            if (2u < cnt) MI_ACC_LOAD(220, 221, 222, 223, row + 15u * 64u);
            if (2u < cnt) MI_ACC_LOAD(224, 225, 226, 227, row + 16u * 64u);
            if (2u < cnt) MI_ACC_LOAD(228, 229, 230, 231, row + 17u * 64u);
            if (3u < cnt) MI_ACC_LOAD(232, 233, 234, 235, row + 18u * 64u);
            if (3u < cnt) MI_ACC_LOAD(236, 237, 238, 239, row + 19u * 64u);
            if (3u < cnt) MI_ACC_LOAD(240, 241, 242, 243, row + 20u * 64u);
            if (3u < cnt) MI_ACC_LOAD(244, 245, 246, 247, row + 21u * 64u);
            if (3u < cnt) MI_ACC_LOAD(248, 249, 250, 251, row + 22u * 64u);
            if (3u < cnt) MI_ACC_LOAD(252, 253, 254, 255, row + 23u * 64u);
        }
        return cnt * kRows;
    };
    // this tile's rows out of the ACC registers; the body loads of the tile (4, issued just before) may still be in flight
    auto readRows = [&](ContactRows* cur, uint32_t cnt) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (0u < cnt) MI_ACC_READ(cur[0].r[0], 160, 161, 162, 163);
        if (0u < cnt) MI_ACC_READ(cur[0].r[1], 164, 165, 166, 167);
        if (0u < cnt) MI_ACC_READ(cur[0].r[2], 168, 169, 170, 171);
        if (0u < cnt) MI_ACC_READ(cur[0].r[3], 172, 173, 174, 175);
        if (0u < cnt) MI_ACC_READ(cur[0].r[4], 176, 177, 178, 179);
        if (0u < cnt) MI_ACC_READ(cur[0].r[5], 180, 181, 182, 183);
        if (1u < cnt) MI_ACC_READ(cur[1].r[0], 184, 185, 186, 187);
        if (1u < cnt) MI_ACC_READ(cur[1].r[1], 188, 189, 190, 191);
        if (1u < cnt) MI_ACC_READ(cur[1].r[2], 192, 193, 194, 195);
        if (1u < cnt) MI_ACC_READ(cur[1].r[3], 196, 197, 198, 199);
        if (1u < cnt) MI_ACC_READ(cur[1].r[4], 200, 201, 202, 203);
        if (1u < cnt) MI_ACC_READ(cur[1].r[5], 204, 205, 206, 207);
        if (2u < cnt) MI_ACC_READ(cur[2].r[0], 208, 209, 210, 211);
        if (2u < cnt) MI_ACC_READ(cur[2].r[1], 212, 213, 214, 215);
        if (2u < cnt) MI_ACC_READ(cur[2].r[2], 216, 217, 218, 219);
        if (2u < cnt) MI_ACC_READ(cur[2].r[3], 220, 221, 222, 223);
        if (2u < cnt) MI_ACC_READ(cur[2].r[4], 224, 225, 226, 227);
        if (2u < cnt) MI_ACC_READ(cur[2].r[5], 228, 229, 230, 231);
        if (3u < cnt) MI_ACC_READ(cur[3].r[0], 232, 233, 234, 235);
        if (3u < cnt) MI_ACC_READ(cur[3].r[1], 236, 237, 238, 239);
        if (3u < cnt) MI_ACC_READ(cur[3].r[2], 240, 241, 242, 243);
        if (3u < cnt) MI_ACC_READ(cur[3].r[3], 244, 245, 246, 247);
        if (3u < cnt) MI_ACC_READ(cur[3].r[4], 248, 249, 250, 251);
        if (3u < cnt) MI_ACC_READ(cur[3].r[5], 252, 253, 254, 255);
    };
    (void)fetchRows(0);
    for (uint32_t it = 0; it < sweeps; ++it)
        for (uint32_t slot = 0; slot < mySlots; ++slot) {
            ContactRows cur[4];
            unsigned long long* rec = nullptr;
#ifdef MI_DBG_TIMELINE
            if (g_dbgTimeline && it * mySlots + slot < 256u) rec = g_dbgTimeline + ((size_t)blockIdx.x * 256u + it * mySlots + slot) * 8u;
            if (rec && threadIdx.x == 0) { rec[6] = ((unsigned long long)it << 32) | slot; rec[7] = lTile[slot]; }
#endif
            MI_STAMP(rec, 0);
            const uint32_t ct = lDesc[3 * slot], cnt = lDesc[3 * slot + 1], io = lDesc[3 * slot + 2];
            uint4 meta; float4 nf; float2 mass;
            if (METALDS) { meta = lMeta[slot * 64u + lane]; nf = lNormal[slot * 64u + lane]; mass = lMass[slot * 64u + lane]; }
            else { meta = nxMeta; nf = nxNf; mass = nxMass; }
            const uint32_t nextSlot = slot + 1u < mySlots ? slot + 1u : 0u;
            const bool more = slot + 1u < mySlots || it + 1u < sweeps;
            // (a tile that had to poll in the previous sweep prefetching AFTER its wait, so that its polls do not queue behind the prefetch, was measured slower —
            // 0.68 vs 0.63 ms, round 3: the rows arriving late costs more)
            struct Prefetch {
                enum : bool { kPinRows = true };
                decltype(fetchRows)& fetch; decltype(readRows)& read; ContactRows* cur; uint32_t cnt; uint32_t* crit; uint32_t next; bool more; unsigned long long* rec; bool noWait;
                __device__ __forceinline__ bool knockNoWait() const { return noWait; }
                __device__ __forceinline__ uint32_t early() { read(cur, cnt); MI_STAMP(rec, 1); return more ? fetch(next) : 0u; }
                __device__ __forceinline__ void late(bool waited) { if (threadIdx.x == 0) *crit = waited ? 1u : 0u; }
            } prefetch{fetchRows, readRows, cur, cnt, &lCrit[slot], nextSlot, more, rec, MI_KNOCK(2) != 0u};
            float2* li = lImp + (size_t)io * 64u;
            // (the contact count of THIS tile as a literal in each case: the row read-back's `if (k < cnt)` guards fold, and no row register is "defined on some paths only" —
            // such values were kept alive around the loop: 61 register copies at the top of every visit)
            switch (cnt) {
                case 1: prefetch.cnt = 1u; processTile<1, IMPLDS, XCD, Prefetch&>(ct, lane, it, meta, nf, mass, cur, imp, gVel, sc, li, gVelL, prefetch); break;
                case 2: prefetch.cnt = 2u; processTile<2, IMPLDS, XCD, Prefetch&>(ct, lane, it, meta, nf, mass, cur, imp, gVel, sc, li, gVelL, prefetch); break;
                case 3: prefetch.cnt = 3u; processTile<3, IMPLDS, XCD, Prefetch&>(ct, lane, it, meta, nf, mass, cur, imp, gVel, sc, li, gVelL, prefetch); break;
                default: prefetch.cnt = 4u; processTile<4, IMPLDS, XCD, Prefetch&>(ct, lane, it, meta, nf, mass, cur, imp, gVel, sc, li, gVelL, prefetch); break;
            }
        }
}

// The kernels proper.  The two variants with the slot data in LDS fit into 208 architectural registers: theirs are capped there (amdgpu_num_vgpr takes no template
// argument, hence explicit specialisations) and v208..v255 hold two more resident positions.
#define MI_PERSIST_KERNEL_ATTRS __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, MI_PERSIST_WPE)))
template <bool METALDS, bool XCD, bool IMPLDS>
MI_PERSIST_KERNEL_ATTRS void k_contact_solve_persist(MI_PERSIST_PARAMS) { persistSolveBody<METALDS, XCD, IMPLDS>(MI_PERSIST_PASS); }
template <> MI_PERSIST_KERNEL_ATTRS __attribute__((amdgpu_num_vgpr(208))) void k_contact_solve_persist<true, true, true>(MI_PERSIST_PARAMS) { persistSolveBody<true, true, true>(MI_PERSIST_PASS); }
template <> MI_PERSIST_KERNEL_ATTRS __attribute__((amdgpu_num_vgpr(208))) void k_contact_solve_persist<true, false, true>(MI_PERSIST_PARAMS) { persistSolveBody<true, false, true>(MI_PERSIST_PASS); }
static_assert(kResidentVgprBase == 208, "the cap of the specialisations above");

// Overflow colour (manifolds the 64 colours could not take: a body with that many incident manifolds — bodies spawned into one another): sequential, slots in ascending
// pair-key order, by ONE wave.  Round 6 (tools/gpu_fuzz.py: a 150-body heap of deeply overlapping shapes took 0.4 s per step): one lane used to walk the slots alone,
// ~13 us each (four dependent loads, a device-scope fence).  Now the 64 lanes fetch the constants of 64 consecutive slots together — one round trip per 64 slots —
// and then take their turns in slot order; a turn is the two bodies' loads, the solve, the stores and a workgroup-scope fence (the wave's L1 stays coherent with its own
// write-through stores; nobody else touches these bodies during this launch).  Same arithmetic, same order: same bits.
__global__ __launch_bounds__(64) void k_contact_solve_serial(BinInfo bi, const uint4* __restrict__ slotMeta, const float4* __restrict__ slotNormal,
                                                             const float2* __restrict__ slotMass, const float4* __restrict__ rows, float4* imp, float4* gVel) {
    if (blockIdx.x != 0) return;
    const uint32_t lane = threadIdx.x;
    for (uint32_t base = 0; base < bi.count; base += 64u) {
        const uint32_t tile = bi.tileStart + (base >> 6), ctBase = bi.ctStart + (base >> 6) * 4u;   // (the overflow bin's tiles hold four contact-tiles each)
        const uint32_t n = min(64u, bi.count - base);
        const uint4 meta = slotMeta[(size_t)tile * 64u + lane];
        const float4 nf = slotNormal[(size_t)tile * 64u + lane];
        const float2 mass = slotMass[(size_t)tile * 64u + lane];
        const uint32_t cnt = lane < n ? (meta.w & 7u) : 0u;
        ContactRows c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((uint32_t)k < cnt) {
                const float4* __restrict__ row = rows + ((size_t)ctBase + k) * (kRows * 64u) + lane;
#pragma unroll
                for (uint32_t r = 0; r < kRows; ++r) c[k].r[r] = row[r * 64u];
                c[k].imp = imp[((size_t)ctBase + k) * 64u + lane];
            }
        }
        const uint32_t bA = meta.x, bB = meta.y;
        const float imA = mass.x, imB = mass.y;
        const bool live = meta.w != 0u && (imA != 0.f || imB != 0.f);
        for (uint32_t turn = 0; turn < n; ++turn) {
            if (lane == turn && cnt) {
                const float4 a0 = gVel[2 * bA], a1 = gVel[2 * bA + 1], b0 = gVel[2 * bB], b1 = gVel[2 * bB + 1];
                V3 vA = xyz(a0), wA = xyz(a1), vB = xyz(b0), wB = xyz(b1);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if ((uint32_t)k < cnt) {
                        float2 im = make_float2(c[k].imp.x, c[k].imp.y);
                        solveOne(c[k], contactNormal(c[k], nf, (meta.w & kMetaPerContactNormal) != 0u), im, imA, imB, vA, wA, vB, wB);
                        if (live) imp[((size_t)ctBase + k) * 64u + lane] = make_float4(im.x, im.y, c[k].imp.z, c[k].imp.w);
                    }
                }
                if (live && imA != 0.f) { gVel[2 * bA] = f4(vA, a0.w); gVel[2 * bA + 1] = f4(wA, a1.w); }   // .w: version tags, untouched by this path
                if (live && imB != 0.f) { gVel[2 * bB] = f4(vB, b0.w); gVel[2 * bB + 1] = f4(wB, b1.w); }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // the next turn's loads come after these stores
        }
    }
    __threadfence();
}

}  // namespace mi
