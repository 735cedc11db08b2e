// jv_pqw_body.h — the persistent pool kernel with SEVERAL WAVES PER QUERY (round 3; instances in jv_kernels_pqw.hip).
//
// Same search as jv_pqp_body.h (GraphSearcher.search as called from J/JVectorReader.java:165-173, PQ provider, no filter,
// threshold <= 0, flat graph) and the same single sorted pool in LDS, but one query is run by a
// workgroup of W = pq_M / 16 waves:
//   * the PQ look-up table lives in registers, SPLIT BY CHUNK: wave w holds subspaces 16 w .. 16 w + 15 (64 VGPRs
//     instead of 128) and scores that chunk for every neighbour of the expanded node; the W chunk sums of a neighbour meet
//     in the adjacent-pair tree of the canonical order (oracle/jv_oracle.c jvo_pq_score: 16-subspace chunks summed left to
//     right, chunks combined pairwise) — for W = 2 that is ONE add, so scores stay bit-equal;
//   * wave 0 owns the pool (best unexpanded entry, rank search, insert, boundary); the other waves hand their chunk sums
//     over through 256 B of LDS per wave and two workgroup barriers per expansion;
//   * after the loop all waves share the visited-count pass (one LDS hash set, compare-and-swap from every wave; the log
//     groups are dealt round robin) and the exact rerank (64-entry batches dealt round robin);
// Register budget per wave: 64 (table) + the working set, 128 VGPRs at 4 waves per SIMD — the one-wave kernel needed
// 256 VGPRs + 320 B of scratch per lane.
#pragma once
#include "jv_pqp_body.h"
#include "jv_serve_claim.h"

// LDS-only workgroup barrier: waits for this wave's LDS operations, NOT for its global loads (a __syncthreads() would
// drain the prefetched fused block with vmcnt(0))
__device__ __forceinline__ void pqw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// order this wave's own LDS traffic (one wave's lanes exchanging data through LDS)
__device__ __forceinline__ void pqw_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// keeps a uniform base address on the scalar unit: without it the compiler folds the per-lane offset into a hoisted
// 64-bit VGPR base, which it then spills (every reload is a vmcnt(0) that drains the prefetched block as well)
typedef const __attribute__((address_space(1))) unsigned char* pqw_gptr;
__device__ __forceinline__ pqw_gptr pqw_scalar_base(const void* p) {
    const uint64_t u = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    pqw_gptr q = (pqw_gptr)(uintptr_t)(((uint64_t)hi << 32) | lo);
    asm volatile("" : "+s"(q));
    return q;
}
// (the empty asm keeps the 32-bit lane offset's zero-extension next to the load, where the instruction selector can fold
//  it into the "scalar base + 32-bit VGPR offset" form; hoisted out of the loop it becomes a 64-bit VGPR pair that spills)
__device__ __forceinline__ int pqw_ld_i32(pqw_gptr base, uint32_t off) {
    asm volatile("" : "+v"(off));
    return *(const __attribute__((address_space(1))) int*)(base + off);
}
__device__ __forceinline__ u32x4 pqw_ld_u32x4(pqw_gptr base, uint32_t off) {
    asm volatile("" : "+v"(off));
    return *(const __attribute__((address_space(1))) u32x4*)(base + off);
}

// Diagnostic build only (-DJV_STAMPS): this kernel keeps its per-phase cycle accumulators in LDS (behind the ctrl
// words) — sixteen 64-bit accumulators in scalar registers cost the loop its register budget and moved the very waits
// the stamps were meant to find.
#ifdef JV_STAMPS
#define PQW_STAMP_DECL unsigned long long* const stl = (unsigned long long*)(smem + a.pqp_scratch_off + W * 256 + 64); unsigned long long st_last = clock64();
#define PQW_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = clock64(); if (lane == 0) stl[i] += t_ - st_last; st_last = t_; __builtin_amdgcn_sched_barrier(0); }
#define PQW_STAMP_COUNT(i, v) { if (lane == 0) stl[i] += (unsigned long long)(v); }
#define PQW_STAMP_FLUSH if (lane == 0) { for (int i_ = 0; i_ < 16; i_++) { if (a.dbg) atomicAdd((unsigned long long*)a.dbg + i_, stl[i_]); stl[i_] = 0ull; } }
#else
#define PQW_STAMP_DECL
#define PQW_STAMP(i) {}
#define PQW_STAMP_COUNT(i, v) {}
#define PQW_STAMP_FLUSH {}
#endif

// KEY_MIN materialised where it is stored (hoisted out of the search loop the 64-bit constant becomes a spilled VGPR pair)
__device__ __forceinline__ int64_t pqw_key_min() {
    int lo = 0, hi = (int)0x80000000;
    asm volatile("" : "+v"(lo), "+v"(hi));
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}

typedef int i32x16 __attribute__((ext_vector_type(16)));
// ctrl words (ints) behind the exchange area
enum { PQW_C = 0, PQW_C2 = 1, PQW_WHY = 2, PQW_NP = 3, PQW_NEXP = 4, PQW_EXPANDED = 5, PQW_QI = 6, PQW_AGAIN = 7, PQW_C3 = 12, PQW_C4 = 13, PQW_CNT = 48 /* [W <= 16]: behind the diagnostic build's accumulators */ };

// One hash class of the visited-count pass in STEPS (round 5; the form jv_kernels_vis.hip's batch kernel was built around): a wave
// takes E = 1 024 / R log entries per step — one load of the log chunk, four 16-byte row loads, sixteen neighbour ids per lane —
// and sends all of them down the probe chains together: one full-width round (sixteen compare-and-swap instructions), then what
// is still pending is packed into the wave's list in LDS (each lane's ids behind those of the lanes below) and walked 64 ids at
// a time, each lane down its own chain.  The LDS charges per instruction, not per live lane, and a group-of-16-rows pass spent
// most of its instructions on the chains' tails with two or three lanes alive.  The rows of the step after next are requested
// before the current step probes.  Shapes: R = 4 * LPR in {16, 32, 64}, adjacency rows 16-byte aligned.
// Returns false when a chain ran past `chain_max` slots (the set is too full: the caller doubles the classes).
template <int LPR, int W>
__device__ __forceinline__ bool pqw_visited_steps(uint32_t* vh, const uint32_t vmask, const int vshift, const int pshift, const uint32_t pmask,
                                                  const uint32_t p, int32_t* lst, const int lcap, const int32_t* explog, const int nexp,
                                                  const int32_t* adj, const int wv, const int lane, int& cntl) {
    constexpr int R = LPR * 4, RPL = JV_WAVE / LPR, E = 1024 / R, NLD = E / RPL, IDS = NLD * 4;
    static_assert(NLD == 4 && IDS == 16, "four row loads, sixteen ids per lane and step");
    constexpr int chain_max = 96;
    const int nsteps = (nexp + E - 1) / E;
    auto load_log = [&](int s) -> int {
        if (s >= nsteps) return 0;
        return __hip_atomic_load(&explog[min(s * E + (lane & (E - 1)), nexp - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto load_rows = [&](int s, int cur, int (&dst)[IDS]) {
        if (s >= nsteps) return;
        int lo = lane;
        asm volatile("" : "+v"(lo));  // (nothing derived from the lane id is worth a register across the steps)
        const int lr4 = (lo / LPR) << 2, lc = (lo % LPR) * 4;
#pragma unroll
        for (int h = 0; h < NLD; h++) {
            const int node = __builtin_amdgcn_ds_bpermute(h * RPL * 4 + lr4, cur);  // (entries behind the log's end repeat its last node)
            const u32x4 v = *(const u32x4*)(adj + (size_t)node * R + lc);
            dst[4 * h] = (int)v.x, dst[4 * h + 1] = (int)v.y, dst[4 * h + 2] = (int)v.z, dst[4 * h + 3] = (int)v.w;
        }
    };
    bool fine = true;
    auto probe = [&](int s, const int (&q)[IDS]) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int lrow = lo / LPR;
        uint32_t pm = 0;  // bit u: id u of this lane is still on its way down a chain
#pragma unroll
        for (int h = 0; h < NLD; h++) {
            const bool ok = s * E + h * RPL + lrow < nexp;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int id = q[4 * h + j];
                if (ok && id >= 0 && (((uint32_t)id * 0x9E3779B1u >> pshift) & pmask) == p) pm |= 1u << (4 * h + j);  // (rows are padded with -1)
            }
        }
        int round = 0, npend = 0, first = 0;
        for (;; round++) {
            if (round >= chain_max) {
                fine = false;
                return;
            }
            uint32_t K = 0x9E3779B1u;
            asm volatile("" : "+s"(K));  // (a product per id kept across the rounds is sixteen registers)
#pragma unroll
            for (int c = 0; c < IDS; c += 4) {
                if (!__any((pm & (0xFu << c)) != 0)) continue;
                uint32_t oldv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    oldv[u] = 0;
                    if (pm & (1u << (c + u))) oldv[u] = atomicCAS(&vh[((((uint32_t)q[c + u] * K) >> vshift) + (uint32_t)round) & vmask], HASH_EMPTY, (uint32_t)q[c + u]);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (pm & (1u << (c + u))) {
                        const bool fresh = oldv[u] == HASH_EMPTY;
                        cntl += fresh ? 1 : 0;
                        if (fresh || oldv[u] == (uint32_t)q[c + u]) pm &= ~(1u << (c + u));
                    }
                }
            }
            const int mine = __popc(pm);
            int incl = mine;
#pragma unroll
            for (int o = 1; o < JV_WAVE; o <<= 1) {
                const int t = __shfl_up(incl, o, JV_WAVE);
                if (lo >= o) incl += t;
            }
            npend = __builtin_amdgcn_readlane(incl, JV_WAVE - 1);
            first = incl - mine;
            if (npend <= lcap) break;
        }
        round++;
        if (npend == 0) return;
#pragma unroll
        for (int u = 0; u < IDS; u++)
            if (pm & (1u << u)) lst[first + __popc(pm & ((1u << u) - 1u))] = q[u];
        for (int b = 0; b < npend; b += JV_WAVE) {
            bool pend = b + lo < npend;
            const uint32_t id = (uint32_t)lst[min(b + lo, npend - 1)];
            uint32_t slot = (((id * 0x9E3779B1u) >> vshift) + (uint32_t)round) & vmask;
            for (int chain = round; __any(pend); chain++) {
                if (chain >= chain_max) {
                    fine = false;
                    return;
                }
                if (pend) {
                    const uint32_t old = atomicCAS(&vh[slot], HASH_EMPTY, id);
                    cntl += old == HASH_EMPTY ? 1 : 0;
                    pend = !(old == HASH_EMPTY || old == id);
                    slot = (slot + 1) & vmask;
                }
            }
        }
    };
    int idsA[IDS], idsB[IDS];
#pragma unroll
    for (int u = 0; u < IDS; u++) idsA[u] = -1, idsB[u] = -1;
    int s0 = wv, curL;
    {
        const int c0 = load_log(s0), c1 = load_log(s0 + W);
        curL = load_log(s0 + 2 * W);
        load_rows(s0, c0, idsA);
        load_rows(s0 + W, c1, idsB);
    }
    for (; s0 < nsteps && fine; s0 += 2 * W) {
        probe(s0, idsA);
        load_rows(s0 + 2 * W, curL, idsA);
        curL = load_log(s0 + 3 * W);
        if (s0 + W >= nsteps || !fine) break;
        probe(s0 + W, idsB);
        load_rows(s0 + 3 * W, curL, idsB);
        curL = load_log(s0 + 4 * W);
    }
    return fine;
}

// NCHT: row length in 64-float chunks known at compile time (rerank), 0 = any d
// CAPK: pool capacity class: 0 -> <= 512 entries, 1 -> <= 1 024, 2 -> <= 2 048
// W:    waves per query = pq_M / 16 (2, 4; 12 = the reference's default 192 subspaces for 768-d .. 1 536-d fields)
// NL:   the first NL of a wave's 16 subspaces keep their table rows in LDS (plain gathers) instead of registers: every
//       look-up served from LDS saves four ds_bpermute — the LDS unit is this kernel's busiest resource — and four VGPRs
// FILT: the query has a doc filter (J/JVectorReader.java:157-163; round 4 — filtered searches ran one wave per query on
//       jv_pqp_body.h until now).  As there: every scored node is traversed, only accepted ones enter jvector's result queue, so
//       the pool holds every scored node whose score >= the rk-th best ACCEPTED one (~ rk / selectivity entries: capacity classes
//       up to 16 384 entries); key bit 2 = accepted, `bpos` tracks the rk-th best accepted entry.
template <int NCHT, int CAPK, int W, int NL, bool FILT = false>
__device__ __forceinline__ void search_one_pqw(const JvIndexDev& ix, const JvSearchArgs& a, const int qi, unsigned char* smem, int32_t* explog) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rk = a.rk, topK = a.topK;
    const int R = ix.R, cs = ix.pq_code_stride;
    float* qc_lds = (float*)(smem + a.pqp_qc_off);  // centred query, only during the LUT build (aliases the pool)
    float* xchg = (float*)(smem + a.pqp_scratch_off);             // [W][64] chunk sums (row 0 unused)
    int* ctrl = (int*)(smem + a.pqp_scratch_off + W * 256);       // PQW_* words
    const int log_cap = a.pqp_log_cap;
    int32_t* o_nodes = a.out_nodes + (size_t)qi * topK;
    int32_t* o_docs = a.out_docs ? a.out_docs + (size_t)qi * topK : nullptr;
    float* o_scores = a.out_scores + (size_t)qi * topK;
    const uint64_t* const accw = FILT ? a.accept + (size_t)qi * (size_t)a.accept_stride : nullptr;
    auto accepts = [&](int node) -> bool {  // the reference's acceptOrds lambda
        if (a.accept_ord) return (a.accept_ord[node >> 6] >> (node & 63)) & 1ull;  // (batch-wide filter, translated by the host's pre-pass)
        const int doc = ix.ord2doc ? ix.ord2doc[node] : node;
        return doc >= 0 && (int64_t)doc < a.accept_docs && ((accw[doc >> 6] >> (doc & 63)) & 1ull);
    };
    if (FILT) {
        // Rung choice only (never results), as in jv_pqp_body.h: the filter's selectivity from 256 sampled words; a pool of
        // ~ rerankK / selectivity entries that cannot fit this launch's capacity is handed on at once.  Every wave of the
        // workgroup computes the same estimate from the same words, so they all leave or all stay.
        const int64_t nwords = (a.accept_docs + 63) >> 6;
        int bits = 0;
        if (nwords > 0) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t h = ((uint64_t)(u * JV_WAVE + lane + 1) * 0x9E3779B97F4A7C15ull) >> 20;
                bits += __popcll(accw[nwords <= 4 * JV_WAVE ? (int64_t)((u * JV_WAVE + lane) % (int)nwords) : (int64_t)(h % (uint64_t)nwords)]);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) bits += __shfl_xor(bits, o, JV_WAVE);
        const float sel = fmaxf((float)bits, 1.0f) * (1.0f / 16384.0f);
        const float need = (float)rk / sel * 1.05f + 64.0f + (float)R;
        if (need > (float)a.cand_cap && (a.retry_only == 1 || (a.retry_only == 0 && a.cand_cap < 8 * rk + 256))) {
            if (threadIdx.x == 0) {
                a.out_flags[qi] = (int32_t)(JV_FLAG_OVERFLOW | (9u << 8));  // (9: estimated pool beyond this launch's)
                a.out_count[qi] = 0;
            }
            if (wv == 0) {
                for (int i = lane; i < topK; i += JV_WAVE) {
                    o_nodes[i] = -1;
                    if (o_docs) o_docs[i] = -1;
                    o_scores[i] = 0.0f;
                }
                if (a.done && a.done_all) {  // (server mode: the caller redoes a flagged row on the launch path)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                    if (lane == 0) __hip_atomic_store(&a.done[qi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }

    // ---- centred query -> this wave's 16 subspaces of the look-up table, in registers ----
    PQW_STAMP_DECL  // (diagnostic build: the clock starts before the table is built)
    const float* qg = a.queries + (size_t)qi * ix.d;
    for (int i = threadIdx.x; i < ix.nch * 64; i += JV_WAVE * W) {
        float v = i < ix.d ? qg[i] : 0.0f;
        if (ix.pq_centroid && i < ix.d) v = v - ix.pq_centroid[i];
        qc_lds[i] = v;
    }
    __syncthreads();
    float* const lutl = (float*)(smem + a.pqw_lut_off) + wv * NL * 256;  // [NL][256] this wave's LDS rows
    float lutr[NL < 16 ? 16 - NL : 1][4];  // lutr[i - NL][e], lane l = lut[16 wv + i][4 l + e] (NL = 16: the whole table is in LDS)
    {
        const bool l2 = ix.sim == 0;
        // Row length and subspace count known at compile time (d = 64 NCHT, M = 16 W, equal subspaces of WS = 4 NCHT / W dimensions):
        // one chain of PFS-dimension pieces over the wave's 16 subspaces, the NEXT piece's codebook rows in flight while the
        // current one is summed (two register sets, the chain unrolled so that none is copied).  The run-time form below waits
        // for one L2 round trip per piece: 48 of them per table at WS = 24 — 2.4 % of a query at rerankK 1 200, 14 % at 160.
        constexpr int WS = NCHT > 0 ? (NCHT * 4) / W : 0;
        constexpr int PFS = WS >= 8 && WS % 8 == 0 ? 8 : (WS >= 4 && WS % 4 == 0 ? 4 : 0);
        if (PFS > 0 && (NCHT * 4) % W == 0 && ix.pq_sub_off[16 * wv + 16] - ix.pq_sub_off[16 * wv] == 16 * WS && ix.pq_sub_off[16 * wv + 1] - ix.pq_sub_off[16 * wv] == WS) {
            constexpr int PFC = PFS > 0 ? PFS : 1;
            constexpr int NPC = PFS > 0 ? WS / PFC : 1;   // pieces per subspace
            const float* const cbase = ix.pq_cbT + (size_t)ix.pq_sub_off[16 * wv] * 256 + 4 * lane;
            const float* const qbase = qc_lds + ix.pq_sub_off[16 * wv];
            f32x4 cbA[PFC], cbB[PFC];
            auto ld = [&](int piece, f32x4 (&cb)[PFC]) {
#pragma unroll
                for (int u = 0; u < PFC; u++) cb[u] = *(const f32x4*)(cbase + (size_t)(piece * PFC + u) * 256);
            };
            float acc4[4] = {0.f, 0.f, 0.f, 0.f};
            auto sum = [&](int piece, const f32x4 (&cb)[PFC]) {
#pragma unroll
                for (int u = 0; u < PFC; u++) {
                    const float qc = qbase[piece * PFC + u];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (l2) {
                            const float df = qc - cb[u][e];
                            acc4[e] = fmaf(df, df, acc4[e]);
                        } else {
                            acc4[e] = fmaf(qc, cb[u][e], acc4[e]);
                        }
                    }
                }
            };
            ld(0, cbA);
#pragma unroll
            for (int pc = 0; pc < 16 * NPC; pc++) {   // (fully unrolled: pc, the subspace pc / NPC and the register set are constants)
                if (pc + 1 < 16 * NPC) {
                    if (pc & 1) ld(pc + 1, cbA);
                    else ld(pc + 1, cbB);
                }
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the loads below the sums to save registers)
                if (pc & 1) sum(pc, cbB);
                else sum(pc, cbA);
                if ((pc + 1) % NPC == 0) {
                    const int i = pc / NPC;
                    if (i < NL) {
                        *(f32x4*)(lutl + i * 256 + 4 * lane) = (f32x4){acc4[0], acc4[1], acc4[2], acc4[3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) lutr[(i < NL || NL >= 16) ? 0 : i - NL][e] = acc4[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) acc4[e] = 0.f;
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < 16; i++) {  // (same fmaf chains as build_lut)
            const int m = 16 * wv + i;
            const int d0 = ix.pq_sub_off[m], d1 = ix.pq_sub_off[m + 1];
            float acc4[4] = {0.f, 0.f, 0.f, 0.f};
            constexpr int PF = 8;
            for (int db = d0; db < d1; db += PF) {
                f32x4 cb[PF];
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    if (db + u < d1) cb[u] = *(const f32x4*)(ix.pq_cbT + (size_t)(db + u) * 256 + 4 * lane);
                    else cb[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    if (db + u < d1) {
                        const float qc = qc_lds[db + u];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (l2) {
                                const float df = qc - cb[u][e];
                                acc4[e] = fmaf(df, df, acc4[e]);
                            } else {
                                acc4[e] = fmaf(qc, cb[u][e], acc4[e]);
                            }
                        }
                    }
                }
            }
            if (i < NL) {
                *(f32x4*)(lutl + i * 256 + 4 * lane) = (f32x4){acc4[0], acc4[1], acc4[2], acc4[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) lutr[(i < NL || NL >= 16) ? 0 : i - NL][e] = acc4[e];
            }
        }
    }
    // Cosine (round 6) rides on the "any d" instances (NCHT = 0): the look-up table is the dot product's, the code vector's squared
    // norm comes with the neighbour (JvIndexDev.pq_fused_norm: summed once per node at index creation, canonical order) and the
    // query's |q|^2 is taken here, from the uncentred query (cosine never centres: no centroid), before the pool overwrites it.
    // score = (1 + dot / sqrt(|q|^2 |c|^2)) / 2 — the expression of every other kernel and of the oracle (jvo_pq_score).
    constexpr bool COSI = NCHT == 0;
    const bool cosine = COSI && ix.sim == 2;
    float qn2 = 0.0f;
    if (COSI && cosine && wv == 0) {
        qn2 = query_norm2(ix, qc_lds, lane);
        qn2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(qn2)));
    }
    __syncthreads();  // every wave is done with qc_lds: the pool may overwrite it

    // this wave's chunk of one code row: 16 look-ups summed left to right (the canonical order inside a chunk)
    auto adc_chunk_regs = [&](const u32x4 cw) -> float {
        float sum = 0.0f;
        // the LDS rows first: NL independent gathers, one round trip
        float tl[NL > 0 ? NL : 1];
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const uint32_t code = (cw[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
            tl[i] = lutl[i * 256 + code];
        }
#pragma unroll
        for (int i = 0; i < NL; i++) sum = sum + tl[i];
        // the 16 ds_bpermute of four look-ups are issued back to back before the first result is consumed
#pragma unroll
        for (int g4 = NL; g4 < 16; g4 += 4) {
            int t[4][4];
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                const int i = g4 + ii;
                const int w = (int)cw[i >> 2];
                const int addr = (int)(((uint32_t)w >> ((i & 3) * 8)) & 0xFFu);  // ds_bpermute reads lane (addr >> 2) & 63 = code >> 2
#pragma unroll
                for (int e = 0; e < 4; e++) t[ii][e] = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(lutr[NL >= 16 ? 0 : i - NL][e]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                const int i = g4 + ii;
                const int w = (int)cw[i >> 2];
                const int sh = (i & 3) * 8;
                const int m0 = __builtin_amdgcn_sbfe(w, sh, 1), m1 = __builtin_amdgcn_sbfe(w, sh + 1, 1);  // -1 / 0: code bits 0, 1
                const int s01 = (m0 & t[ii][1]) | (~m0 & t[ii][0]), s23 = (m0 & t[ii][3]) | (~m0 & t[ii][2]);
                sum = sum + __int_as_float((m1 & s23) | (~m1 & s01));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return sum;
    };
    // Round 6, the latency variant of two-wave queries (whole table in LDS, one to three queries per CU: instances NL = 16, W = 2, no
    // filter — small launches and the query server): the helper wave PRE-SCORES the pair of blocks requested ahead while the pool wave
    // ranks and inserts the current pair's keys.  The pair requested at a pass is the next pass's pair four times out of five
    // (counter 11 of the stamped build); such a pass then finds its 64 raw sums waiting in LDS and runs without
    // its own ADC, without the exchange and without barrier B — a quarter of the pool wave's chain.  Same arithmetic, same order:
    // with the whole table in LDS any wave can sum any chunk (sixteen gathers left to right), and for W = 2 the pair tree is the
    // one add chunk 0 + chunk 1, so the pre-scored value has the bits `combine` would produce.  Two buffers, alternating per pass:
    // the helper writes the one the pool wave reads NEXT pass; barrier A of that pass orders the two.
    // (W = 4 — PQ-64, C4's shape — the same with wave 1 as the one pre-scorer: it sums all four chunks and combines them in the
    //  pair tree ((0 + 1) + (2 + 3)) `combine` walks; waves 2 and 3 only skip their scoring on a pre-scored pass)
    constexpr bool PRE = NL == 16 && (W == 2 || W == 4) && !FILT;
    float* const psb = (float*)(smem + a.pqp_scratch_off + W * 256 + 256);  // [2][64] (plan_pqw_lds: variant 1, two waves)
    auto adc_chunk_lds = [&](const float* rows, const u32x4 cw) -> float {
        float tl[16];
#pragma unroll
        for (int i = 0; i < 16; i++) tl[i] = rows[i * 256 + ((cw[i >> 2] >> ((i & 3) * 8)) & 0xFFu)];
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) sum = sum + tl[i];
        return sum;
    };
    // chunk sums of lane l's node -> raw distance / dot product: adjacent-pair tree over the W chunks (lanes_tree_sum)
    auto combine = [&](float s0, int l) -> float {
        if (W == 1) return s0;
        // the W chunk sums padded with +0.0f to the next power of two, adjacent pairs first — what lanes_tree_sum does over
        // next_pow2(W) lanes (the padding adds are kept: x + 0.0f is not x for x = -0.0f)
        constexpr int PW = W <= 2 ? 2 : W <= 4 ? 4 : W <= 8 ? 8 : 16;
        float c[PW];
        c[0] = s0;
#pragma unroll
        for (int w = 1; w < PW; w++) c[w] = w < W ? xchg[w * 64 + l] : 0.0f;
#pragma unroll
        for (int span = 1; span < PW; span <<= 1)
#pragma unroll
            for (int w = 0; w < PW; w += 2 * span) c[w] = c[w] + c[w + span];
        return c[0];
    };

    // ---- the pool: sorted descending; bit 0 of a key = "not expanded yet"; slots [np, cap] hold the minimum key ----
    const int cap = a.cand_cap;  // entries; slot `cap` is a permanent sentinel
    int64_t* pool = (int64_t*)(smem + a.pqp_pool_off);
    for (int i = threadIdx.x; i <= cap; i += JV_WAVE * W) pool[i] = KEY_MIN;
    int np = 0, nexp = 0, expanded = 0;
    int why = 0;
    int nrej = 0;         // rejected entries in the pool (all at the boundary score)
    int bpos = -1, nacc = 0;  // FILT: position of the rk-th best accepted entry (-1: fewer than rk so far), accepted entries in the pool
    float bscore = 0.0f;  // score of the rk-th best entry once np >= rk
    {
        const int ep = ix.entry;
        u32x4 cw = (u32x4){0, 0, 0, 0};
        if (lane == 0) cw = *(const u32x4*)(ix.pq_codes + (size_t)ep * cs + wv * 16);
        const float s = adc_chunk_regs(cw);
        if (wv > 0 && lane == 0) xchg[wv * 64] = s;
        __syncthreads();
        if (wv == 0) {
            float sc;
            if (COSI && cosine) sc = map_score(2, combine(s, 0) / sqrtf(qn2 * ix.pq_node_norm[ep]));
            else sc = map_score(ix.sim == 0 ? 0 : 1, combine(s, 0));
            sc = __shfl(sc, 0, JV_WAVE);
            const bool acc_ep = FILT ? accepts(ep) : true;
            if (lane == 0) pool[0] = pqp_key<FILT>(sc, ep, acc_ep);
            if (FILT) {
                nacc = acc_ep ? 1 : 0;
                if (nacc >= rk) bpos = 0, bscore = sc;
            } else if (rk <= 1) {
                bscore = sc;
            }
        }
        np = 1;
        __syncthreads();
    }

    // fused block of the expanded node: lane j < R owns neighbour j (its ordinal: wave 0 only; its 16 code bytes of this
    // wave's chunk); the runner-up's block is requested one expansion ahead
    // The search loops run as lambdas over an OPAQUE copy of the lane id: everything derived from it (LDS addresses,
    // block offsets) is then computed after the table build instead of at kernel entry — hoisted that far, those values
    // live across the register peak of the table build, get spilled, and every reload inside the loop is a vmcnt(0)
    // that also drains the prefetched block.
    auto opaque_lane = [&]() -> int {
        int l = (int)(threadIdx.x & 63);
        asm volatile("" : "+v"(l));
        return l;
    };
    int pf_nn = -1;
    float pf_na = 0.0f;  // (cosine: the neighbours' code norms travel with their ordinals)
    u32x4 pf_cw = (u32x4){0, 0, 0, 0};
    if (wv == 0) PQW_STAMP(7)  // LUT build + entry point
    if (wv == 0) {
        // ================================ wave 0: the pool ================================
        // (it is the query's critical path: on a SIMD shared with other queries' helper waves it issues first)
        __builtin_amdgcn_s_setprio(2);
        [&](const int lane) {
        // Two blocks per scoring pass: lanes 0..31 hold the neighbours of the best unexpanded entry, lanes 32..63 those of
        // the runner-up (R <= 32).  The runner-up's scores wait in their lanes: when it is still the best entry one
        // expansion later (9 times out of 10) that expansion needs no pass and no barrier at all.
        const bool pair = 2 * R <= JV_WAVE;
        const int hf = pair ? lane >> 5 : 0;
        const int hl = pair ? lane & 31 : lane;
        const int jl = min(hl, R - 1);
        const uint32_t cw_off = (uint32_t)(jl * cs + wv * 16);
        const uint32_t adj_off = (uint32_t)(jl * 4);
        int sc_node0 = -1, sc_node1 = -1;  // nodes whose neighbours' scores sit in the lower / upper lanes
        int pf_node0 = -1, pf_node1 = -1;  // nodes whose blocks were requested ahead into the lower / upper lanes
        float score = 0.0f;
        int nn = -1;
        float na = 0.0f;   // cosine: |code vector of this lane's neighbour|^2
        bool accn = true;  // FILT: this lane's neighbour is accepted by the query's doc filter
        int64_t pv = KEY_MIN;  // lane t: last key of pool chunk t (low bits may be stale: they never decide a comparison with a new key)
        // ranks of BOTH halves' keys are taken in the pass iteration (the per-lane search costs the same for 32 or 64 keys); the
        // runner-up's half keeps them for its own expansion, corrected there for the keys the first half inserted meanwhile
        int rold_s = 0;
        bool cand_s = false;
        int ru_pos = -1, ru_lo = 0, ru_hi = 0;  // the runner-up's pool entry as the pass iteration found it
        // bit t: pool chunk t MAY hold an unexpanded entry (a clear bit: it holds none).  The best unexpanded entry is one
        // chunk read away wherever it is — a key that lands far ahead of the frontier no longer costs a walk through every
        // fully expanded chunk in between once it has been expanded itself (that walk was most of the "find" step)
        unsigned long long um = 1ull;
        constexpr int UGS = CAPK >= 5 ? 2 : (CAPK >= 4 ? 1 : 0);  // log2(pool chunks per mask bit): 64 bits cover every capacity class
        unsigned long long ins_lo = 0ull;  // lanes (lower half) whose keys went into the pool in the pass iteration
        int pass_no = 0;  // PRE: scoring passes so far (which of the two pre-score buffers belongs to this pass)
        while (true) {
            // (wave-uniform state, said so: the compiler's divergence analysis gives up on values that pass through LDS loads and
            //  the joins behind lane-level branches, and then runs this whole loop with vector compares and exec masks)
#define PQW_UNI(x) x = __builtin_amdgcn_readfirstlane(x)
            PQW_UNI(np); PQW_UNI(nexp); PQW_UNI(expanded); PQW_UNI(why); PQW_UNI(nrej); PQW_UNI(bpos); PQW_UNI(nacc);
            um = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(um >> 32)) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)um);
            PQW_UNI(sc_node0); PQW_UNI(sc_node1); PQW_UNI(pf_node0); PQW_UNI(pf_node1); PQW_UNI(ru_pos); PQW_UNI(ru_lo); PQW_UNI(ru_hi);
            bscore = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bscore)));
#undef PQW_UNI
            // ---- best unexpanded entry: first chunk whose bit is set (stale bits are cleared on the way; sentinels have bit 0 clear) ----
            int t1 = 0;
            int64_t e1 = 0;
            unsigned long long m1 = 0ull;
            while (um) {
                const int g1 = __ffsll((long long)um) - 1;
#pragma unroll
                for (int cc = 0; cc < (1 << UGS); cc++) {  // (pools beyond 4 096 entries: a mask bit stands for 2 / 4 chunks)
                    if (!m1) {
                        t1 = (g1 << UGS) + cc;
                        e1 = pool[min((t1 << 6) + lane, cap)];
                        m1 = __ballot((e1 & 1ll) != 0);
                    }
                }
                if (m1) break;
                um &= um - 1ull;
            }
            int c = -1, b1 = 0, idx = 0, pk_lo = 0, pk_hi = 0;
            bool reject = false;
            const int e1lo = (int)(uint32_t)(e1 & 0xFFFFFFFFll), e1hi = (int)(e1 >> 32);
            if (m1) {
                b1 = __ffsll((long long)m1) - 1;
                idx = (t1 << 6) + b1;
                pk_lo = __builtin_amdgcn_readlane(e1lo, b1);
                pk_hi = __builtin_amdgcn_readlane(e1hi, b1);
                const float sc = hi_score(pk_hi);
                c = lo_node<FILT>(pk_lo);
                if (sc < a.threshold) why = 1;  // a node the two-queue form would expand but not collect: general path
                else if (nexp >= log_cap) why = 2;
                else if (a.visit_limit > 0 && expanded >= a.visit_limit) why = 15;  // Lucene discards this search
                // strict admission (DESIGN.md "Single-pool search"): when the ADMITTED entries scoring >= the candidate
                // already fill the result queue (the worst result ties with the candidate), jvector expands the candidate
                // without admitting it
                if (FILT) {
                    // the same rule over ACCEPTED entries (jv_pqp_body.h): with t = accepted entries of the boundary's equal-score
                    // run at or ahead of bpos, rk - t accepted entries score higher; the candidate is rejected when those + the
                    // accepted, expanded entries of the run (minus the rejected ones) already fill rerankK.  A candidate the
                    // filter does not accept is expanded and never admitted.
                    if (why == 0 && (pk_lo & 4) && bpos >= 0 && sc == bscore) {
                        int tt = bpos >> 6;
                        while (tt > 0 && __builtin_amdgcn_readfirstlane((int)(pool[tt << 6] >> 32)) == pk_hi) tt--;
                        int t_le = 0, ex_acc = 0;
                        for (;;) {
                            const int64_t ee = pool[min((tt << 6) + lane, cap)];
                            const unsigned long long eq = __ballot((int)(ee >> 32) == pk_hi);
                            const unsigned long long am = eq & __ballot((ee & 4ll) != 0);
                            const unsigned long long un = __ballot((ee & 1ll) != 0);
                            const int last_in = bpos - (tt << 6);  // lanes <= last_in are at or ahead of bpos
                            const unsigned long long le = last_in >= 63 ? ~0ull : (last_in < 0 ? 0ull : ((2ull << last_in) - 1ull));
                            t_le += __popcll(am & le);
                            ex_acc += __popcll(am & ~un);
                            if (((tt + 1) << 6) >= np || (tt >= (bpos >> 6) && !(eq >> 63))) break;
                            tt++;
                        }
                        reject = ex_acc - nrej >= t_le;
                    }
                } else if (why == 0 && expanded >= rk && idx < rk + nrej) {
                    int ge = idx;
                    int64_t ee = e1;
                    for (int tt = t1;;) {
                        const unsigned long long eq = __ballot((int)(ee >> 32) == pk_hi);
                        unsigned long long ex = eq & ~__ballot((ee & 1ll) != 0);
                        if (tt == t1) ex &= ~((2ull << b1) - 1ull);  // positions behind the candidate only
                        ge += __popcll(ex);
                        tt++;
                        if (!(eq >> 63) || (tt << 6) >= np) break;  // the equal-score run ends inside this chunk
                        ee = pool[min((tt << 6) + lane, cap)];
                    }
                    if (np >= rk && sc == bscore) ge -= nrej;
                    reject = ge >= rk;
                }
            }
            c = __builtin_amdgcn_readfirstlane(c);  // (block addresses are computed on the scalar unit)
            const bool stop = !m1 || why != 0;
            int half = -1;
            if (!stop) half = c == sc_node0 ? 0 : (c == sc_node1 ? 1 : -1);
            bool fresh = false;  // this iteration runs a scoring pass
            PQW_STAMP(15)  // (diagnostic) find proper
            if (stop || half < 0) {
                // ---- scoring pass: the next three unexpanded entries as well (runner-up: scored now; the two after it:
                // their blocks are requested for the next pass) ----
                int bn1 = -1, bn2 = -1, bn3 = -1;
                ru_pos = -1;
                if (!stop) {
                    unsigned long long mm = m1 & (m1 - 1ull);
                    int elo = e1lo, ehi = e1hi, t2 = t1;
                    bool second = false;
#pragma unroll
                    for (int k = 1; k < 4; k++) {
                        const unsigned long long um2 = UGS == 0 ? (um & ~((2ull << t1) - 1ull)) : ((((t1 + 1) << 6) < np) ? 1ull : 0ull);  // chunks behind the first that may hold more
                        if (!mm && !second && um2) {
                            t2 = UGS == 0 ? __ffsll((long long)um2) - 1 : t1 + 1;
                            const int64_t e2 = pool[min((t2 << 6) + lane, cap)];
                            mm = __ballot((e2 & 1ll) != 0);
                            elo = (int)(uint32_t)(e2 & 0xFFFFFFFFll);
                            ehi = (int)(e2 >> 32);
                            second = true;
                        }
                        int nb = -1;
                        if (mm) {
                            const int ln = __ffsll((long long)mm) - 1;
                            const int klo = __builtin_amdgcn_readlane(elo, ln);
                            nb = lo_node<FILT>(klo);
                            if (k == 1) {  // the runner-up's entry: its key and position decide whether both entries are expanded at once
                                ru_lo = klo;
                                ru_hi = __builtin_amdgcn_readlane(ehi, ln);
                                ru_pos = (t2 << 6) + ln;
                            }
                            mm &= mm - 1ull;
                        }
                        if (k == 1) bn1 = nb;
                        if (k == 2) bn2 = nb;
                        if (k == 3) bn3 = nb;
                    }
                    if (!pair) bn3 = bn2, bn2 = bn1, bn1 = -1;  // one block per pass: the runner-up is only requested ahead
                }
                bn1 = __builtin_amdgcn_readfirstlane(bn1);
                bn2 = __builtin_amdgcn_readfirstlane(bn2);
                bn3 = __builtin_amdgcn_readfirstlane(bn3);
                if (W > 1) {
                    if (lane == 0) {
                        ctrl[PQW_C] = stop ? -1 : c;
                        ctrl[PQW_C2] = bn1;
                        ctrl[PQW_C3] = bn2;
                        ctrl[PQW_C4] = bn3;
                    }
                    pqw_barrier();  // A: the other waves learn which blocks to score
                }
                if (stop) break;
                PQW_STAMP_COUNT(11, (pf_node0 == c && pf_node1 == bn1) ? 1 : 0)
                PQW_STAMP_COUNT(12, (pf_node0 == c) ? 1 : 0)
                PQW_STAMP_COUNT(8, 1)
                PQW_STAMP(0)  // barrier A
                const int wy = bn1 >= 0 ? bn1 : c;  // (no runner-up: the upper lanes score the same block again, unused)
                // PRE: both blocks of this pass are the pair requested (and pre-scored by the helper wave) at the previous pass
                const bool pre_hit = PRE && pair && !a.no_prescore && pf_node0 == c && pf_node1 == wy;
                u32x4 cw;
                {
                    const bool hit = hf ? pf_node1 == wy : pf_node0 == c;
                    if (hit) {
                        nn = pf_nn;
                        cw = pf_cw;
                        if (COSI && cosine) na = pf_na;
                    } else {
                        const int node = hf ? wy : c;
                        nn = *(const int32_t*)((const unsigned char*)(ix.adj + (size_t)node * (size_t)R) + adj_off);
                        cw = JV_STREAM_LOAD((const u32x4*)(ix.pq_fused + (size_t)node * (size_t)R * (size_t)cs + cw_off));
                        if (COSI && cosine) na = *(const float*)((const unsigned char*)(ix.pq_fused_norm + (size_t)node * (size_t)R) + adj_off);
                    }
                }
                if (hl >= R) nn = -1;
                // FILT: the neighbours' accept bits are requested BEFORE the next pair's blocks — vmcnt retires in order, so a wait
                // for an accept word issued behind the prefetch would drain the prefetch on every pass
                if (FILT) {
                    accn = nn >= 0 ? accepts(nn) : false;
                    __builtin_amdgcn_sched_barrier(0);
                }
#ifdef JV_STAMPS
                asm volatile("" ::"v"(nn), "v"(cw));  // (diagnostic) the wait for this pass's blocks lands in phase 9
#endif
                PQW_STAMP(9)  // (diagnostic) wait for the blocks
                // the blocks of the two entries after the runner-up are requested now, UNCONDITIONALLY (clamped): they are
                // the likely pair of the next pass, two expansions from here
                pf_node0 = pair ? bn2 : bn2;
                pf_node1 = bn3;
                {
                    const int x = bn2 >= 0 ? bn2 : c, y = bn3 >= 0 ? bn3 : x;
                    const int node = hf ? y : x;
                    pf_nn = *(const int32_t*)((const unsigned char*)(ix.adj + (size_t)node * (size_t)R) + adj_off);
                    pf_cw = JV_STREAM_LOAD((const u32x4*)(ix.pq_fused + (size_t)node * (size_t)R * (size_t)cs + cw_off));
                    if (COSI && cosine) pf_na = *(const float*)((const unsigned char*)(ix.pq_fused_norm + (size_t)node * (size_t)R) + adj_off);
                }
                __builtin_amdgcn_sched_barrier(0);
                float raw;
                if (PRE && pre_hit) {
                    raw = psb[(pass_no & 1) * 64 + lane];  // (written by the helper wave before it reached barrier A of this pass)
                } else {
                    const float s0 = adc_chunk_regs(cw);
                    if (W > 1) pqw_barrier();  // B: the other waves' chunk sums are in xchg
                    raw = combine(s0, lane);
                }
                pass_no++;
                if (COSI && cosine) score = map_score(2, raw / sqrtf(qn2 * na));
                else score = map_score(ix.sim == 0 ? 0 : 1, raw);
#ifdef JV_STAMPS
                asm volatile("" ::"v"(score));
#endif
                sc_node0 = c;
                sc_node1 = pair ? bn1 : -1;
                half = 0;
                fresh = true;
                PQW_STAMP(2)  // ADC + exchange
            }
            // mark the entry expanded; log the node
            if (lane == b1) ((int*)pool)[2 * idx] = pk_lo & (reject ? ~3 : ~1);
            nrej += reject ? 1 : 0;
            if (lane == 0) explog[nexp] = c;
            nexp++;
            expanded++;
            if (half == 0) sc_node0 = -1;
            else sc_node1 = -1;
            PQW_STAMP(1)
            const int64_t v = pqp_key<FILT>(score, nn, accn);
            bool keep, dual = false;
            int rold;
            const bool have_b = FILT ? bpos >= 0 : np >= rk;  // a boundary exists
            if (fresh) {
                bool cand = nn >= 0;
                if (have_b && score < bscore) cand = false;  // below the boundary for good
                // ---- rank of the surviving keys in the pool; "same node" = equal up to the low bits ----
                int rold_a = 0;
                unsigned long long todo_m = __ballot(cand);
                if (__popcll(todo_m) > 6 || CAPK >= 4) {  // (pools beyond 4 096 entries: the chunk pivots do not fit one register across the wave)
                    // many survivors (the pool is still filling: every neighbour is a candidate): per-lane 8-ary search, all
                    // lanes at once, cost independent of their number
                    if (cand) {
                        int lo = 0;
                        if (CAPK == 5) lo = rank_level<2048, 8>(pool, lo, cap, v);
                        if (CAPK == 5) lo = rank_level<128, 16>(pool, lo, cap, v);
                        if (CAPK == 4) lo = rank_level<1024, 8>(pool, lo, cap, v);
                        if (CAPK == 3) lo = rank_level<512, 8>(pool, lo, cap, v);
                        // (classes 1, 2: the three pivots of the first level are last keys of pool chunks — lane 8k + 7 / 4k + 3 of `pv`,
                        //  up to the stale low bits that never decide a comparison with a new key: three register reads instead of
                        //  an LDS round trip)
                        if (CAPK == 2 || CAPK == 1) {
                            constexpr int CPB = CAPK == 2 ? 8 : 4;  // pool chunks per first-level block
                            const int pvl = (int)(uint32_t)(pv & 0xFFFFFFFFll), pvh = (int)(pv >> 32);
                            int c1 = 0;
    #pragma unroll
                            for (int k2 = 0; k2 < 3; k2++) {
                                const int64_t pk = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(pvh, CPB * k2 + CPB - 1) << 32) |
                                                             (uint64_t)(uint32_t)__builtin_amdgcn_readlane(pvl, CPB * k2 + CPB - 1));
                                c1 += pk > v ? 1 : 0;
                            }
                            lo = c1 * CPB * 64;
                        }
                        if (CAPK == 1) lo = rank_level<64, 4>(pool, lo, cap, v);
                        else if (CAPK == 4) lo = rank_level<64, 16>(pool, lo, cap, v);
                        else if (CAPK != 5) lo = rank_level<64, 8>(pool, lo, cap, v);
                        if (CAPK == 5) lo = rank_level<8, 16>(pool, lo, cap, v);
                        else lo = rank_level<8, 8>(pool, lo, cap, v);
                        int64_t p3[9];
    #pragma unroll
                        for (int k2 = 0; k2 < 9; k2++) p3[k2] = pool[min(lo + k2, cap)];
                        int c3 = 0;
                        bool dup = false;
    #pragma unroll
                        for (int k2 = 0; k2 < 9; k2++) {
                            if (k2 < 8) c3 += p3[k2] > v ? 1 : 0;
                            dup |= (p3[k2] | 3ll) == v;
                        }
                        rold_a = lo + c3;
                        if (dup) cand = false;
                    }
                } else {
                    const int vlo_ = (int)(uint32_t)(v & 0xFFFFFFFFll), vhi_ = (int)(v >> 32);
                    while (todo_m) {
                        int jj[4], tcs[4];
                        int64_t kk[4], ee[4];
    #pragma unroll
                        for (int u = 0; u < 4; u++) {
                            jj[u] = -1;
                            tcs[u] = 0;
                            kk[u] = 0;
                            ee[u] = 0;
                            if (todo_m) {
                                const int j = __ffsll((long long)todo_m) - 1;
                                todo_m &= todo_m - 1ull;
                                jj[u] = j;
                                kk[u] = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(vhi_, j) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(vlo_, j));
                                tcs[u] = __popcll(__ballot(pv > kk[u]));  // chunks whose last key ranks ahead of the new key
                                ee[u] = pool[min((tcs[u] << 6) + lane, cap)];
                            }
                        }
    #pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (jj[u] >= 0) {
                                const int r = (tcs[u] << 6) + __popcll(__ballot(ee[u] > kk[u]));
                                const bool dup = __ballot((ee[u] | 3ll) == kk[u]) != 0ull;  // same node => same score => same key up to the low bits
                                if (lane == jj[u]) {
                                    rold_a = r;
                                    if (dup) cand = false;
                                }
                            }
                        }
                    }
                }
                rold_s = rold_a;
                cand_s = cand;
                keep = cand && hf == 0;
                rold = rold_a;
                // ---- both entries at once.  The runner-up IS the next entry expanded unless one of this entry's new keys ranks
                // ahead of it; then its own new keys can go into the pool in the same insert (ranks, shift, boundary and trim
                // once for two expansions — the final pool is the one two inserts would leave: every key the first insert's
                // higher boundary would have refused falls off in the trim).  Only while nothing of jvector's pop-time logic
                // can apply to the runner-up: result queue not full (no strict admission), inside the best rerankK, score
                // above the threshold, room in the log and in the pool, no visit limit in reach.
                // (`expanded` and `nexp` already count this iteration's entry: they are what the runner-up's own pop would see)
                // (FILT: only while fewer than rerankK accepted entries exist — afterwards the runner-up's pop can meet the boundary's
                //  equal-score run, whose bookkeeping wants one insert at a time)
                if ((!FILT || bpos < 0) && sc_node1 >= 0 && ru_pos >= 0 && ru_pos < rk && nrej == 0 && expanded < rk && nexp < log_cap &&
                    hi_score(ru_hi) >= a.threshold && (a.visit_limit <= 0 || expanded < a.visit_limit)) {
                    const int64_t ruk = (int64_t)(((uint64_t)(uint32_t)ru_hi << 32) | (uint64_t)(uint32_t)ru_lo);
                    const unsigned long long all = __ballot(cand);
                    if (!__ballot(keep && v > ruk) && np + __popcll(all) <= cap) {
                        dual = true;
                        PQW_STAMP_COUNT(10, 1)
                        keep = cand;
                        if (lane == 0) {
                            ((int*)pool)[2 * ru_pos] = ru_lo & ~1;
                            explog[nexp] = sc_node1;
                        }
                        nexp++;
                        expanded++;
                        sc_node1 = -1;
                    }
                }
            } else {
                // the runner-up's half: ranked against the pool as it was before the first half's keys went in
                keep = cand_s && hf == 1;
                if (have_b && score < bscore) keep = false;  // (the boundary may have risen since)
                rold = rold_s;
                unsigned long long it = ins_lo;
                const int vlo2 = (int)(uint32_t)(v & 0xFFFFFFFFll), vhi2 = (int)(v >> 32);
                while (it) {
                    const int j = __ffsll((long long)it) - 1;
                    it &= it - 1ull;
                    const int64_t kj = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(vhi2, j) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(vlo2, j));
                    rold += kj > v ? 1 : 0;
                    if (kj == v) keep = false;  // the same node in both rows: it went in with the first half
                }
            }
            unsigned long long km = __ballot(keep);
            int nk = __popcll(km);
            if (fresh) ins_lo = dual ? 0ull : km;  // (twins inside the row are taken out below)
            PQW_STAMP(3)  // boundary test + rank search + duplicate test
            if (nk > 0) {
                int rnew = 0, r_min, r_max;
                if (nk == 1) {
                    // the common case: one new key, every entry behind it moves up by one
                    r_min = r_max = __builtin_amdgcn_readlane(rold, __ffsll((long long)km) - 1);
                    // (four chunks are read before the first is written back: one LDS round trip per four chunks instead of
                    //  one per chunk — the pool wave's time is mostly such round trips)
                    const int t_lo = r_min >> 6;
                    for (int t0 = (np - 1) >> 6; t0 >= t_lo; t0 -= 4) {
                        int64_t e4[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) e4[u] = pool[min((max(t0 - u, 0) << 6) + lane, cap)];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int pos = ((t0 - u) << 6) + lane;
                            if (t0 - u >= t_lo && pos < np && pos >= r_min) pool[pos + 1] = e4[u];
                        }
                    }
                } else {
                    // every kept key is read out of its lane in turn and compared by all lanes at once: rank among the new
                    // keys, twins (the same neighbour twice in one adjacency row of a malformed graph: keep the first)
                    const int vlo = (int)(uint32_t)(v & 0xFFFFFFFFll), vhi = (int)(v >> 32);
                    for (int attempt = 0; attempt < 2; attempt++) {
                        rnew = 0;
                        bool twin = false;
                        unsigned long long it = km;
                        while (it) {
                            const int j = __ffsll((long long)it) - 1;
                            it &= it - 1ull;
                            const int64_t kj = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(vhi, j) << 32) |
                                                         (uint64_t)(uint32_t)__builtin_amdgcn_readlane(vlo, j));
                            rnew += kj > v ? 1 : 0;
                            twin |= kj == v && j < lane;
                        }
                        const unsigned long long km2 = __ballot(keep && !twin);
                        if (km2 == km) break;
                        keep = keep && !twin;
                        km = km2;
                        nk = __popcll(km);
                        if (fresh) ins_lo = dual ? 0ull : km;
                    }
                    const int lane_first = __ffsll((long long)__ballot(keep && rnew == 0)) - 1;     // largest new key
                    const int lane_last = __ffsll((long long)__ballot(keep && rnew == nk - 1)) - 1;  // smallest new key
                    r_min = __builtin_amdgcn_readlane(rold, lane_first);
                    r_max = __builtin_amdgcn_readlane(rold, lane_last);
                    PQW_STAMP(14)  // (diagnostic) ranks among the new keys
                    // in-place shift, from the last occupied chunk down to the chunk of the first insertion point
                    const int t_mixed = r_max >> 6;  // chunks above it shift uniformly by nk
                    const int t_lo = r_min >> 6;
                    for (int t0 = (np - 1) >> 6; t0 >= t_lo; t0 -= 4) {
                        int64_t e4[4];  // (four chunks in flight, as above)
#pragma unroll
                        for (int u = 0; u < 4; u++) e4[u] = pool[min((max(t0 - u, 0) << 6) + lane, cap)];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int t = t0 - u;
                            if (t >= t_lo) {
                                const int pos = (t << 6) + lane;
                                int cnt = nk;
                                if (t <= t_mixed) {
                                    const int cs0 = t << 6;
                                    cnt = __popcll(__ballot(keep && rold <= cs0));
                                    unsigned long long inm = __ballot(keep && rold > cs0 && rold <= cs0 + 63);
                                    while (inm) {
                                        const int j = __ffsll((long long)inm) - 1;
                                        inm &= inm - 1ull;
                                        cnt += pos >= __builtin_amdgcn_readlane(rold, j) ? 1 : 0;
                                    }
                                }
                                if (pos < np && cnt > 0) pool[pos + cnt] = e4[u];
                            }
                        }
                    }
                }
                if (keep) pool[rold + rnew] = v;
                {
                    // entries from the first insertion point on moved up by at most nk <= 64 positions: into their own chunk or
                    // the next; the new keys are unexpanded where they landed
                    const unsigned long long lowm = (1ull << ((r_min >> 6) >> UGS)) - 1ull;
                    um = (um & lowm) | ((um | (um << 1)) & ~lowm);
                    const int npos = rold + rnew;
                    unsigned long long it = km;
                    while (it) {
                        const int j = __ffsll((long long)it) - 1;
                        it &= it - 1ull;
                        um |= 1ull << ((__builtin_amdgcn_readlane(npos, j) >> 6) >> UGS);
                    }
                }
                PQW_STAMP(4)  // ranks among the new keys + shift + insert
                // boundary = the rk-th best entry; entries behind it stay only while they tie with its score
                const int ntot = np + nk;
                np = ntot;
                if (FILT) {
                    // (jv_pqp_body.h's bookkeeping of the rk-th best ACCEPTED entry, on wave-uniform values)
                    const unsigned long long kacc = __ballot(keep && accn);
                    if (bpos >= 0) {
                        // new keys that landed ahead of the old boundary entry push it back; the accepted ones among them make
                        // the rk-th best accepted entry one of its predecessors: walk back over that many accepted entries
                        const unsigned long long before = __ballot(keep && rold <= bpos);
                        const int P = bpos + __popcll(before);
                        int need = __popcll(before & kacc);
                        bpos = P;
                        if (need > 0) {
                            int tt = P >> 6;
                            unsigned long long lim = (P & 63) ? ((1ull << (P & 63)) - 1ull) : 0ull;  // positions < P
                            if (!lim) tt--, lim = ~0ull;
                            for (;;) {
                                const int64_t ee = pool[(tt << 6) + lane];
                                const unsigned long long am = __ballot((ee & 4ll) != 0) & lim;
                                const int cnt = __popcll(am);
                                if (cnt >= need) {
                                    bpos = (tt << 6) + pqp_select_nth_bit(am, cnt - need + 1);
                                    break;
                                }
                                need -= cnt;
                                tt--;
                                lim = ~0ull;
                            }
                        }
                    } else {
                        nacc += __popcll(kacc);
                        if (nacc >= rk) {  // the pool holds rk accepted entries for the first time: find the rk-th from the front
                            int need = rk;
                            for (int tt = 0;; tt++) {
                                const int64_t ee = pool[min((tt << 6) + lane, cap)];
                                const unsigned long long am = __ballot((tt << 6) + lane < ntot && (ee & 4ll) != 0);
                                const int cnt = __popcll(am);
                                if (cnt >= need) {
                                    bpos = (tt << 6) + pqp_select_nth_bit(am, need);
                                    break;
                                }
                                need -= cnt;
                            }
                        }
                    }
                    if (bpos >= 0) {
                        const int bhi = __builtin_amdgcn_readfirstlane((int)(pool[bpos] >> 32));
                        const float nb = hi_score(bhi);
                        if (nb != bscore) nrej = 0;
                        bscore = nb;
                        // ties behind the boundary stay (slots left: cap - (bpos + 1) - R; none left = this launch's pool is too
                        // small for this filter's selectivity), everything below goes
                        const int slack = cap - (bpos + 1) - R;  // (with a boundary in place one expansion's keys arrive per insert)
                        int run = 0, acc_t = 0;
                        for (int p0 = bpos + 1; run < slack; p0 += JV_WAVE) {
                            const int64_t eb = pool[min(p0 + lane, cap)];
                            const unsigned long long mt = __ballot((int)(eb >> 32) == bhi);  // (a sentinel never matches)
                            const unsigned long long am = __ballot((eb & 4ll) != 0);
                            if (~mt) {
                                const int r = __ffsll((long long)~mt) - 1;
                                run += r;
                                acc_t += __popcll(am & ((1ull << r) - 1ull));
                                break;
                            }
                            run += JV_WAVE;
                            acc_t += __popcll(am);
                        }
                        if (run >= slack) {
                            why = 3;
                        } else {
                            np = bpos + 1 + run;
                            nacc = rk + acc_t;
                            for (int p0 = np; p0 < ntot; p0 += JV_WAVE)
                                if (p0 + lane < ntot) pool[p0 + lane] = pqw_key_min();
                        }
                    } else if (ntot > cap - 2 * R) {
                        why = 3;  // fewer than rk accepted entries among more scored nodes than this launch's pool holds
                    }
                } else if (ntot >= rk) {
                    const int bhi = __builtin_amdgcn_readfirstlane((int)(pool[rk - 1] >> 32));  // (wave-uniform, and said so: the boundary and the rejected count steer wave-level branches)
                    const float nb = hi_score(bhi);
                    if (nb != bscore) nrej = 0;  // the boundary rose: every rejected entry (they tied with the old one) falls off below
                    bscore = nb;
                    if (ntot > rk) {
                        const int slack = cap - rk - R;
                        int run = 0;
                        for (int p0 = rk;; p0 += JV_WAVE) {
                            const int64_t eb = pool[min(p0 + lane, cap)];
                            const unsigned long long mt = __ballot((int)(eb >> 32) == bhi);  // (a sentinel never matches)
                            if (~mt) {
                                run += __ffsll((long long)~mt) - 1;
                                break;
                            }
                            run += JV_WAVE;
                            if (run >= slack) break;
                        }
                        if (run >= slack) {
                            why = 3;  // more boundary ties than this launch tracks: announced at the top of the next round
                        } else {
                            np = rk + run;
                            for (int p0 = np; p0 < ntot; p0 += JV_WAVE)
                                if (p0 + lane < ntot) pool[p0 + lane] = pqw_key_min();
                        }
                    }
                }
                if (lane >= (r_min >> 6)) pv = pool[min((lane << 6) + 63, cap)];  // chunks from the first insertion point on changed
                um &= (2ull << (((np - 1) >> 6) >> UGS)) - 1ull;  // (nothing behind the pool's last chunk)
                PQW_STAMP(5)  // boundary + trim
                if (why != 0) {  // (why = 3) leave through the common exit so that every wave sees it
                    if (W > 1) {
                        if (lane == 0) ctrl[PQW_C] = -1;
                        pqw_barrier();
                    }
                    break;
                }
            }
        }
        PQW_STAMP(5)
        if (why == 0 && (FILT || nrej > 0)) {
            // take the rejected entries out (FILT: what the filter does not accept goes too): what remains in front is jvector's result queue
            int carry = 0;
            for (int t = 0; (t << 6) < np; t++) {
                const int pos = (t << 6) + lane;
                const int64_t e = pool[min(pos, cap)];
                const bool rej = pos < np && (!(e & 2ll) || (FILT && !(e & 4ll)));
                const unsigned long long rm = __ballot(rej);
                const int shift = carry + __popcll(rm & ((1ull << lane) - 1ull));
                if (pos < np && !rej && shift > 0) pool[pos - shift] = e;
                carry += __popcll(rm);
            }
            for (int p0 = np - carry; p0 < np; p0 += JV_WAVE)
                if (p0 + lane < np) pool[p0 + lane] = KEY_MIN;
            np -= carry;
        }
        if (lane == 0) {
            if (why == 0 && np > (((32 + W - 1) / W) * W) * 64) why = 4;  // (a boundary tie storm: more results than the waves can park in registers — next rung)
            ctrl[PQW_WHY] = why;
            ctrl[PQW_NP] = np;
            ctrl[PQW_NEXP] = nexp;
            ctrl[PQW_EXPANDED] = expanded;
        }
        // the log was written by lane 0 and is read back by every wave: drain the stores (read side: L1-bypassing loads)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }(opaque_lane());
        __builtin_amdgcn_s_setprio(0);
    } else {
        // ================================ waves 1 .. W-1: chunk sums ================================
        [&](const int lane) {
        const bool pair = 2 * R <= JV_WAVE;
        const int hf = pair ? lane >> 5 : 0;
        const int hl = pair ? lane & 31 : lane;
        const int jl = min(hl, R - 1);
        const uint32_t cw_off = (uint32_t)(jl * cs + wv * 16);
        int pf_node0 = -1, pf_node1 = -1;
        int pass_no = 0;
        const float* const lut0 = (const float*)(smem + a.pqw_lut_off);  // PRE: chunk 0's table rows (the pool wave's), all sixteen in LDS
        while (true) {
            pqw_barrier();  // A
            const int c = __builtin_amdgcn_readfirstlane(ctrl[PQW_C]);
            const int bn1 = __builtin_amdgcn_readfirstlane(ctrl[PQW_C2]);
            const int bn2 = __builtin_amdgcn_readfirstlane(ctrl[PQW_C3]);
            const int bn3 = __builtin_amdgcn_readfirstlane(ctrl[PQW_C4]);
            if (c < 0) break;
            const int wy = bn1 >= 0 ? bn1 : c;
            // (the pool wave takes the same decision from the same values: a pre-scored pass has no exchange and no barrier B)
            const bool pre_hit = PRE && pair && !a.no_prescore && pf_node0 == c && pf_node1 == wy;
            u32x4 cw = (u32x4){0, 0, 0, 0};
            if (!(PRE && pre_hit)) {
                const bool hit = hf ? pf_node1 == wy : pf_node0 == c;
                if (hit) {
                    cw = pf_cw;
                } else {
                    const int node = hf ? wy : c;
                    cw = JV_STREAM_LOAD((const u32x4*)(ix.pq_fused + (size_t)node * (size_t)R * (size_t)cs + cw_off));
                }
            }
            pf_node0 = bn2;
            pf_node1 = bn3;
            u32x4 pcw[PRE ? W : 1];  // PRE, wave 1: the other chunks' code bytes of the pair requested ahead (its own chunk: pf_cw)
            {
                const int x = bn2 >= 0 ? bn2 : c, y = bn3 >= 0 ? bn3 : x;
                const int node = hf ? y : x;
                pf_cw = JV_STREAM_LOAD((const u32x4*)(ix.pq_fused + (size_t)node * (size_t)R * (size_t)cs + cw_off));
                if constexpr (PRE) if (pair && !a.no_prescore && wv == 1) {
#pragma unroll
                    for (int w = 0; w < W; w++)
                        if (w != 1) pcw[w] = JV_STREAM_LOAD((const u32x4*)(ix.pq_fused + (size_t)node * (size_t)R * (size_t)cs + (uint32_t)(jl * cs + w * 16)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(PRE && pre_hit)) {
                const float s = adc_chunk_regs(cw);
                xchg[wv * 64 + lane] = s;
                pqw_barrier();  // B
            }
            if constexpr (PRE) if (pair && !a.no_prescore && wv == 1) {
                // the pair requested ahead, every chunk, canonical order: chunk sums left to right, then the adjacent-pair tree —
                // what `combine` gives the pool wave
                float t[W];
#pragma unroll
                for (int w = 0; w < W; w++) t[w] = adc_chunk_lds(lut0 + w * 16 * 256, w == 1 ? pf_cw : pcw[w]);
#pragma unroll
                for (int span = 1; span < W; span <<= 1)
#pragma unroll
                    for (int w = 0; w < W; w += 2 * span) t[w] = t[w] + t[w + span];
                psb[((pass_no + 1) & 1) * 64 + lane] = t[0];
            }
            pass_no++;
        }
        }(opaque_lane());
    }
    __syncthreads();
    why = __builtin_amdgcn_readfirstlane(ctrl[PQW_WHY]);
    np = __builtin_amdgcn_readfirstlane(ctrl[PQW_NP]);
    nexp = __builtin_amdgcn_readfirstlane(ctrl[PQW_NEXP]);
    expanded = __builtin_amdgcn_readfirstlane(ctrl[PQW_EXPANDED]);

    // the pool moves to registers (chunk t to wave t % W) so that the whole LDS allocation can serve as the hash set
    constexpr int PCH = (32 + W - 1) / W;
    int PL[PCH], PH[PCH];
    if (why == 0) {
#pragma unroll
        for (int u = 0; u < PCH; u++) {
            const int t = u * W + wv;
            PL[u] = 0;
            PH[u] = (int)0x80000000;
            if ((t << 6) < np) {
                const int64_t e = pool[min((t << 6) + lane, cap)];
                PL[u] = (int)(uint32_t)(e & 0xFFFFFFFFll);
                PH[u] = (int)(e >> 32);
            }
        }
    }
    __syncthreads();
    int visited = 0;
    // Round 5: a batch launch (no completion words) does not count here.  The pass below is two waves walking the log behind one
    // memory round trip per group while the query holds its slot of the CU; jv_visited_kernel (jv_kernels_vis.hip) counts the
    // whole batch afterwards from copies of the logs, many waves per query and one hash class.  A log the arena has no room for
    // is counted here as before.
    bool vis_later = false;
    if (why == 0 && a.vis_arena != nullptr && a.done == nullptr && nexp > 0) {
        if (threadIdx.x == 0) {
            const uint32_t units = ((uint32_t)nexp + 3u) >> 2;
            const uint32_t off = atomicAdd(a.vis_cursor, units);
            ctrl[PQW_AGAIN] = (off <= a.vis_cap_units && units <= a.vis_cap_units - off) ? (int)off : -1;
        }
        __syncthreads();
        const int off = __builtin_amdgcn_readfirstlane(ctrl[PQW_AGAIN]);
        if (off >= 0) {
            vis_later = true;
            int32_t* dst = a.vis_arena + (size_t)off * 4;
            for (int i = threadIdx.x; i < nexp; i += JV_WAVE * W) dst[i] = __hip_atomic_load(&explog[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) {
                a.vis_off[qi] = (uint32_t)off;
                a.vis_n[qi] = nexp;
            }
        }
        __syncthreads();
    }
    if (why == 0 && !vis_later) {
        // ---- jvector's visitedCount: distinct neighbours of the expanded nodes, entry point excluded; `parts` hash
        // classes are counted one after the other when one table cannot hold them all ----
        uint32_t* vh = (uint32_t*)smem;
        const int hash_bytes = a.pqp_scratch_off;  // everything in front of the exchange / ctrl words
        int vslots = 1;
        while (vslots * 2 * 4 <= hash_bytes) vslots <<= 1;
        const uint32_t vmask = (uint32_t)vslots - 1u;
        const int vshift = 32 - (31 - __clz(vslots));
        const int vlimit = (vslots / 16) * 13;
        const int vlimit_w = vlimit / W;  // fresh entries one wave may add per class
        int parts = 1;
        while (parts < 64 && (long long)nexp * 7 > (long long)vlimit * parts * 2) parts <<= 1;
        constexpr int VB = 8;             // neighbour ids per lane and group
        // Adjacency rows as 16-byte pieces where the shape allows it (R % 4 == 0, rows 16-byte aligned): R / 4 lanes per row,
        // 64 / (R / 4) rows per load instruction, two instructions per group (R = 32: 16 log entries, one bpermute + one load
        // per 8 rows instead of one per 2); else one id per lane and load, VB loads per group.
        const int lpr4 = max(1, R >> 2);
        const bool vec = (R & 3) == 0 && (64 % (2 * (JV_WAVE / lpr4))) == 0 && (((uintptr_t)ix.adj) & 15) == 0;
        const int lpr = vec ? lpr4 : R;                  // lanes per adjacency row
        const int rpl = JV_WAVE / lpr;                   // rows per load instruction
        const int G = vec ? rpl * 2 : rpl * VB;          // log entries per group (divides 64: a group sits in one log chunk)
        const int lrow = lane / lpr, lcol = (lane % lpr) * (vec ? 4 : 1);
        const bool lane_ok = lane < rpl * lpr;
        // the waves' lists of the step form sit behind the set, in what the power-of-two set leaves of the LDS
        const int lcap = min(256, ((hash_bytes - vslots * 4) / (4 * W)) & ~63);
        int32_t* const lst_w = (int32_t*)(smem + (size_t)vslots * 4) + wv * lcap;
        // (only the instances with the whole table in LDS — the latency variant and the query server, 200+ registers — take the step
        //  form: in the 128-register throughput instances its sixteen + sixteen id registers next to the pool's 32 pushed the
        //  kernel from 46 to 117 spills and the search loop lost 4 %; their batches count after the launch anyway)
        constexpr bool steps_inst = NL == 16 && !FILT;
        const bool steps_ok = steps_inst && vec && (R == 16 || R == 32 || R == 64) && lcap >= 64;
        bool again = true;
        while (again && why == 0) {
            again = false;
            visited = 0;
            int plog = 0;
            while ((1 << plog) < parts) plog++;
            const int pshift = vshift - plog;            // the class = the hash bits right below the slot index
            const uint32_t pmask = (uint32_t)parts - 1u;
            for (int p = 0; p < parts && !again; p++) {
                __syncthreads();
                for (int i = threadIdx.x; i < vslots; i += JV_WAVE * W) vh[i] = HASH_EMPTY;
                if (threadIdx.x == 0) ctrl[PQW_AGAIN] = 0;
                __syncthreads();
                if (threadIdx.x == 0 && ((((uint32_t)ix.entry * 0x9E3779B1u) >> pshift) & pmask) == (uint32_t)p)
                    visited_insert_lds(vh, vmask, vshift, (uint32_t)ix.entry);
                __syncthreads();
                int cntl = 0;  // fresh entries, counted per lane
                bool over = false;
                // The log is pulled into registers 1 024 entries at a time; this wave takes every W-th group of rows.  NPF groups
                // are in flight, each in registers of its own (the loop is unrolled NPF times: a copy from one group's registers
                // to the next waits for the YOUNGEST load, and the pass then ran one memory round trip per group; measured
                // NPF = 2 / 3 / 4: 3.12 / 3.10 / 2.81 M QPS at rerankK 160, 566 / 568 / 567 k at 1 200).  Loads are unconditional
                // with clamped indices.
                constexpr int NPF = 2;
                if constexpr (steps_inst) if (steps_ok) {
                    // (round 5: the common shapes take the log in steps of E rows per wave, see pqw_visited_steps)
                    bool fine;
                    if (R == 32) fine = pqw_visited_steps<8, W>(vh, vmask, vshift, pshift, pmask, (uint32_t)p, lst_w, lcap, explog, nexp, ix.adj, wv, lane, cntl);
                    else if (R == 16) fine = pqw_visited_steps<4, W>(vh, vmask, vshift, pshift, pmask, (uint32_t)p, lst_w, lcap, explog, nexp, ix.adj, wv, lane, cntl);
                    else fine = pqw_visited_steps<16, W>(vh, vmask, vshift, pshift, pmask, (uint32_t)p, lst_w, lcap, explog, nexp, ix.adj, wv, lane, cntl);
                    over = !fine;
                }
                for (int blk0 = 0; blk0 < nexp && !over && !steps_ok; blk0 += 1024) {
                    const int nblk = min(1024, nexp - blk0);
                    i32x16 logv;
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        logv[g] = 0;
                        if (g * 64 < nblk)
                            logv[g] = __hip_atomic_load(&explog[blk0 + min(g * 64 + lane, nblk - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const int e_last = (nblk - 1) / G * G;  // first entry of the last group
                    auto load_group = [&](int e0, int (&dst)[VB]) {
                        const int e0c = min(e0, e_last);
                        const int cur = logv[__builtin_amdgcn_readfirstlane(e0c >> 6)];
                        if (vec) {
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const int e = min(e0c + h * rpl + lrow, nblk - 1);
                                const int node = __builtin_amdgcn_ds_bpermute((e & 63) << 2, cur);
                                const u32x4 v = *(const u32x4*)(ix.adj + (size_t)node * R + lcol);
                                dst[4 * h] = (int)v.x, dst[4 * h + 1] = (int)v.y, dst[4 * h + 2] = (int)v.z, dst[4 * h + 3] = (int)v.w;
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const int e = min(e0c + u * rpl + lrow, nblk - 1);
                                const int node = __builtin_amdgcn_ds_bpermute((e & 63) << 2, cur);
                                dst[u] = ix.adj[(size_t)node * R + lcol];
                            }
                        }
                    };
                    // one group's ids into the set; false when the table's fill limit would be passed
                    auto probe_group = [&](const int (&q)[VB], int e0) -> bool {
                        uint32_t hh[VB];
                        bool pend[VB];
                        int pl = 0;
                        {
                            // which of this lane's ids exist: entry e0 + (u / 4 or u) * rpl + lrow of the block
                            const bool ok_a = lane_ok && e0 + lrow < nblk, ok_b = lane_ok && e0 + rpl + lrow < nblk;
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const bool ok = vec ? (u < 4 ? ok_a : ok_b) : (lane_ok && e0 + u * rpl + lrow < nblk);
                                const uint32_t prod = (uint32_t)q[u] * 0x9E3779B1u;   // one product: slot index on top, class below
                                hh[u] = prod >> vshift;
                                pend[u] = ok && q[u] >= 0 && ((prod >> pshift) & pmask) == (uint32_t)p;   // (rows are padded with -1)
                                pl += pend[u] ? 1 : 0;
                            }
                        }
                        // (fresh entries so far + the ids about to probe, one sum over the wave)
                        if (jv_wave_sum_int(cntl + pl) > vlimit_w) return false;
                        // Round one probes all VB batches (the compare-and-swaps overlap); most lanes are done after it — nine of
                        // ten neighbours are re-encounters that sit at or next to their home slot — and the later rounds only
                        // touch the batches that still have a lane on its way down a probe chain.
                        uint32_t live = 0;
                        {
                            uint32_t oldv[VB];
#pragma unroll
                            for (int u = 0; u < VB; u++) oldv[u] = pend[u] ? atomicCAS(&vh[hh[u]], HASH_EMPTY, (uint32_t)q[u]) : 0u;
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const bool fresh = pend[u] && oldv[u] == HASH_EMPTY;
                                cntl += fresh ? 1 : 0;
                                pend[u] = pend[u] && !fresh && oldv[u] != (uint32_t)q[u];
                                hh[u] = (hh[u] + 1) & vmask;
                                if (__any(pend[u])) live |= 1u << u;
                            }
                        }
                        while (live) {
                            uint32_t oldv[VB];
#pragma unroll
                            for (int u = 0; u < VB; u++)
                                if (live & (1u << u)) oldv[u] = pend[u] ? atomicCAS(&vh[hh[u]], HASH_EMPTY, (uint32_t)q[u]) : 0u;
#pragma unroll
                            for (int u = 0; u < VB; u++)
                                if (live & (1u << u)) {
                                    const bool fresh = pend[u] && oldv[u] == HASH_EMPTY;
                                    cntl += fresh ? 1 : 0;
                                    pend[u] = pend[u] && !fresh && oldv[u] != (uint32_t)q[u];
                                    hh[u] = (hh[u] + 1) & vmask;
                                    if (!__any(pend[u])) live &= ~(1u << u);
                                }
                        }
                        return true;
                    };
                    int qq[NPF][VB];
                    const int step = G * W;
#pragma unroll
                    for (int k = 0; k < NPF; k++) load_group(wv * G + k * step, qq[k]);
                    for (int e0 = wv * G; e0 < nblk && !over; e0 += NPF * step) {
#pragma unroll
                        for (int k = 0; k < NPF; k++) {
                            const int ek = e0 + k * step;
                            if (ek < nblk && !over) {
                                if (!probe_group(qq[k], ek)) over = true;
                                else load_group(ek + NPF * step, qq[k]);
                            }
                        }
                    }
                }
                const int cnt = jv_wave_sum_int(cntl);
                if (over && lane == 0) ctrl[PQW_AGAIN] = 1;
                visited += cnt;
                __syncthreads();
                again = __builtin_amdgcn_readfirstlane(ctrl[PQW_AGAIN]) != 0;
            }
            if (again) {
                parts <<= 1;
                if (parts > 64) why = 4;
            }
        }
        if (lane == 0) ctrl[PQW_CNT + wv] = visited;
        __syncthreads();
        visited = 0;
#pragma unroll
        for (int w2 = 0; w2 < W; w2++) visited += __builtin_amdgcn_readfirstlane(ctrl[PQW_CNT + w2]);
        // (the entry point went into its class's set before the counted inserts: it is never counted)
    }
    if (wv == 0) {
        PQW_STAMP(6)  // visited-count pass
        PQW_STAMP_FLUSH
    }
    // Lucene discards every search whose visited + expanded reached the visit limit (J/JVectorReader.java:202-207 reports the
    // sum, AbstractKnnVectorQuery tests it).  The loop above stops on expansions alone (visited is only known now): a search
    // that ran to its end but whose SUM reaches the limit is flagged the same way, with its real counters.
    int early_vis = -1;
    if (why == 0 && !vis_later && a.visit_limit > 0 && visited + expanded >= a.visit_limit) {  // (a row counted after the launch: jv_visited_kernel applies the same rule)
        why = 15;
        early_vis = visited;
    }
    const int nres = np < rk ? np : rk;
    // ---- rerank scratch: query, per-wave todo lists, the pool / exact keys ----
    float* q_lds = (float*)smem;
    size_t roff = (size_t)ix.nch * 64 * sizeof(float);
    float* todo_score = (float*)(smem + roff) + wv * JV_TODO;
    roff += (size_t)W * JV_TODO * sizeof(float);
    int32_t* todo = (int32_t*)(smem + roff) + wv * JV_TODO;
    roff += (size_t)W * JV_TODO * sizeof(int32_t);
    int64_t* fin = (int64_t*)(smem + roff);  // [np] the pool comes back here; the exact keys overwrite it in place
    __syncthreads();
    if (why == 0) {
#pragma unroll
        for (int u = 0; u < PCH; u++) {
            const int t = u * W + wv;
            if ((t << 6) < np) fin[(t << 6) + lane] = (int64_t)(((uint64_t)(uint32_t)PH[u] << 32) | (uint64_t)(uint32_t)PL[u]);
        }
        for (int i = threadIdx.x; i < ix.nch * 64; i += JV_WAVE * W) q_lds[i] = i < ix.d ? qg[i] : 0.0f;
    }
    __syncthreads();
    int above = 0;
    if (why == 0) {
        for (int i = lane; i < nres; i += JV_WAVE) above += key_score(fin[i]) >= a.rerank_floor ? 1 : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) above += __shfl_xor(above, o, JV_WAVE);
        // rerankFloor above every approximate score AND a tie at the best one: jvector rescores the first best entry of
        // its result heap's array, which only the HBM-scratch rung reconstructs (replay_first_best)
        if (above == 0 && nres >= 2 && key_score(fin[0]) == key_score(fin[1])) why = 6;
    }
    if (why != 0) {
        if (wv == 0) {
            if (lane == 0) {
                a.out_flags[qi] = why == 15 ? (int32_t)JV_FLAG_EARLY : (int32_t)(JV_FLAG_OVERFLOW | ((uint32_t)why << 8));
                a.out_count[qi] = 0;
                if (why == 15) {
                    int32_t* st = a.out_stats + (size_t)qi * 4;
                    st[0] = early_vis >= 0 ? early_vis : 0;
                    st[1] = 0;
                    st[2] = expanded;
                    st[3] = expanded;
                }
            }
            for (int i = lane; i < topK; i += JV_WAVE) {
                o_nodes[i] = -1;
                if (o_docs) o_docs[i] = -1;
                o_scores[i] = 0.0f;
            }
            if (a.done && (why == 15 || a.done_all)) {  // an early-terminated row is final (a flagged one is redone by a later launch — or, in server mode, by the caller)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                if (lane == 0) __hip_atomic_store(&a.done[qi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    // ---- rerank (NodeQueue.rerank) with the exact scorer: 64-entry batches dealt round robin to the waves ----
    float qnorm2 = 0.0f;
    if (ix.sim == 2) qnorm2 = query_norm2(ix, q_lds, lane), qnorm2 = __shfl(qnorm2, 0, JV_WAVE);
    __syncthreads();  // every wave has read fin[0], fin[1] and counted `above` before the first exact key lands
    for (int b0 = wv * JV_WAVE; b0 < nres; b0 += JV_WAVE * W) {
        const int i = b0 + lane;
        bool take = false;
        int node = 0;
        if (i < nres) {
            const int64_t k = fin[i];
            node = lo_node<FILT>((int)(uint32_t)(k & 0xFFFFFFFFll));
            take = above > 0 ? key_score(k) >= a.rerank_floor : i == 0;  // position 0 is the best approximate entry
        }
        const unsigned long long tm = __ballot(take);
        const int m = __popcll(tm);
        const int slot = __popcll(tm & ((1ull << lane) - 1ull));
        if (take) todo[slot] = node;
        pqw_wave_sync();
        if (m > 0) {
            // (rows in flight sized for 128 VGPRs; the two-wave LDS-table instances — the latency variant, the query server: 256 registers —
            //  keep a third group of four 768-float rows in flight)
            score_rows<NCHT, 1, true, (NL == 16 && W == 2 && NCHT == 12 && !FILT) ? 1 : 0>(ix, q_lds, todo, m, todo_score, qnorm2, 1.0f, lane);
            pqw_wave_sync();
        }
        if (i < nres) fin[i] = take ? make_key(todo_score[slot], node) : KEY_MIN;
        pqw_wave_sync();
    }
    __syncthreads();
    if (wv != 0) return;
    const int reranked = above > 0 ? above : (nres > 0 ? 1 : 0);
    int cnt = 0;
    for (; cnt < topK && cnt < reranked; cnt++) {
        int64_t bk;
        int bidx;
        scan_max(fin, nres, lane, bk, bidx);
        if (lane == 0) {
            const int node = key_node(bk);
            o_nodes[cnt] = node;
            if (o_docs) o_docs[cnt] = ix.ord2doc ? ix.ord2doc[node] : node;
            o_scores[cnt] = key_score(bk);
            fin[bidx] = KEY_MIN;
        }
        pqw_wave_sync();
    }
    for (int i = cnt + lane; i < topK; i += JV_WAVE) {
        o_nodes[i] = -1;
        if (o_docs) o_docs[i] = -1;
        o_scores[i] = 0.0f;
    }
    PQW_STAMP(13)  // rerank + top-K
    if (lane == 0) {
        a.out_count[qi] = cnt;
        int32_t* st = a.out_stats + (size_t)qi * 4;
        st[0] = vis_later ? -1 : visited;  // (-1: jv_visited_kernel's to fill in)
        st[1] = reranked;
        st[2] = expanded;
        st[3] = expanded;
        a.out_flags[qi] = 0;
    }
    // completion word (one-query-per-call API): the caller that owns this query is woken as soon as ITS row is final, not
    // when the slowest query of the combined launch is.  System-scope release: every store of the row precedes the word.
    if (a.done) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (lane == 0) __hip_atomic_store(&a.done[qi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Persistent grid: one workgroup (W waves) per resident LDS slot, queries dequeued in order.
// OCC = waves per SIMD the register budget is sized for (4: 128 VGPRs, 8 workgroups of 2 waves per CU)
// Device-resident query server: the same search, fed by single queries from a ring of slots in pinned host memory
// (JvServeSlot) instead of a batch — the reference's calling pattern is one query per call from many searcher threads
// (J/JVectorReader.java:129-210), and a kernel launch per call (or per small group of calls) runs into the HIP runtime's
// few hardware queues.  Callers publish slots in ticket order (host word TAIL); workgroups claim tickets with a
// compare-and-swap on HEAD once PUBLISHED has caught up with TAIL (one workgroup at a time reads the host word), answer into
// the slot and set its completion word.  The grid leaves when the host says STOP or nothing was claimed for serve_idle_ticks.
template <int NCHT, int CAPK, int W, int OCC, int NL = 4, bool FILT = false>
__global__ __launch_bounds__(JV_WAVE * W, OCC) void jv_serve_pqw_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* explog = a.pqp_log + (size_t)blockIdx.x * (size_t)a.pqp_log_cap;
    int* ctrl = (int*)(smem + a.pqp_scratch_off + W * 256);
#ifdef JV_STAMPS
    if (threadIdx.x < 16) ((unsigned long long*)(smem + a.pqp_scratch_off + W * 256 + 64))[threadIdx.x] = 0ull;
#endif
    for (;;) {
        if (threadIdx.x == 0) {
            const int ticket = jv_serve_claim(a);
            ctrl[PQW_QI] = ticket;
        }
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(ctrl[PQW_QI]);
        if (ticket < 0) break;
        unsigned char* const sp = a.serve_ring + (size_t)(ticket & (a.serve_slots - 1)) * (size_t)a.serve_slot_bytes;
        JvServeSlot* const slot = (JvServeSlot*)sp;
        if (!jv_serve_slot_current(slot, ticket)) {  // an abandoned ticket: nothing to answer
            __syncthreads();
            continue;
        }
        JvSearchArgs aq = a;
        aq.queries = (const float*)(sp + JV_SERVE_QUERY_OFF);
        aq.nq = 1;
        aq.topK = __builtin_amdgcn_readfirstlane(slot->topK);
        aq.rk = __builtin_amdgcn_readfirstlane(slot->rk);
        aq.visit_limit = __builtin_amdgcn_readfirstlane(slot->visit_limit);
        aq.rerank_floor = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(slot->rerank_floor)));
        aq.cand_cap = aq.rk + 64 + ix.R;  // (the pool a batch launch of this rerankK would use; the LDS plan covers the largest)
        if (FILT) {
            // a query with a doc filter: every request runs in the server's ONE pool (a.cand_cap); a filter whose estimated pool
            // does not fit is handed back at once (retry_only = 1 = "a wider rung follows": the caller takes the launch path)
            aq.cand_cap = a.cand_cap;
            aq.accept = (const uint64_t*)(uintptr_t)slot->accept;
            aq.accept_docs = slot->accept_docs;
            aq.accept_stride = 0;
            aq.accept_ord = nullptr;
            aq.retry_only = 1;
        }
        aq.out_nodes = slot->nodes;
        aq.out_docs = slot->docs;
        aq.out_scores = slot->scores;
        aq.out_count = &slot->count;
        aq.out_stats = slot->stats;
        aq.out_flags = &slot->flags;
        aq.done = &slot->done;
        aq.done_all = 1;
        search_one_pqw<NCHT, CAPK, W, NL, FILT>(ix, aq, 0, smem, explog);
        __syncthreads();
    }
    jv_serve_leave(a);
}

template <int NCHT, int CAPK, int W, int OCC, int NL = 4, bool FILT = false>
__global__ __launch_bounds__(JV_WAVE * W, OCC) void jv_search_pqw_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* explog = a.pqp_log + (size_t)blockIdx.x * (size_t)a.pqp_log_cap;
    int* ctrl = (int*)(smem + a.pqp_scratch_off + W * 256);
#ifdef JV_STAMPS
    if (threadIdx.x < 16) ((unsigned long long*)(smem + a.pqp_scratch_off + W * 256 + 64))[threadIdx.x] = 0ull;
#endif
    // later launches (a.retry_only: wider pool — the filtered rungs): only the queries an earlier one flagged, found up to 8 flags
    // at a time by the first wave (as jv_search_pqp_kernel does; the chunk shrinks with the number of flags per workgroup)
    const int chunk = max(1, min(8, a.nq / ((int)gridDim.x * 8)));
    const int lane = threadIdx.x & 63;
    int base = 0;
    unsigned long long todo = 0ull;
    for (;;) {
        if (FILT && a.retry_only) {  // (only the filtered rungs walk flags: the unfiltered instances do not carry this state across queries)
            if (threadIdx.x < JV_WAVE) {
                int next = a.nq;
                for (;;) {
                    if (todo) {
                        next = base + __ffsll((long long)todo) - 1;
                        todo &= todo - 1ull;
                        break;
                    }
                    if (lane == 0) base = atomicAdd(a.retry_counter, chunk);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= a.nq) break;
                    const int qf = (lane < chunk && base + lane < a.nq) ? a.out_flags[base + lane] : 0;
                    todo = __ballot(((uint32_t)qf & JV_FLAG_OVERFLOW) != 0);
                }
                if (lane == 0) ctrl[PQW_QI] = next;
            }
        } else if (threadIdx.x == 0) {
            ctrl[PQW_QI] = atomicAdd(a.pqp_counter, 1);
        }
        __syncthreads();
        const int qi = __builtin_amdgcn_readfirstlane(ctrl[PQW_QI]);
        if (qi >= a.nq) break;
        search_one_pqw<NCHT, CAPK, W, NL, FILT>(ix, a, qi, smem, explog);
        __syncthreads();
    }
}
