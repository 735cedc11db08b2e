// jv_kernels_pqr.hip — the headline kernel: PQ approximate search on the FUSED layout with the candidate pool held
// in REGISTERS (gfx950 / CDNA4), followed by jvector's visitedCount reconstruction and the exact rerank.
//
// What it computes is GraphSearcher.search as called from J/JVectorReader.java:165-173 for the
// case "PQ provider, no filter, threshold <= 0, flat graph"; results, scores and the four counters are bit-equal to
// oracle/jv_oracle.c (tests/test_gpu_parity.py).  Everything the kernel cannot hold is flagged and re-run by the
// generic ladder (jv_kernels.hip), so exactness never depends on a capacity.
//
// Why registers.  One wave runs one query; a lone wave per SIMD is instruction-issue bound, not bandwidth bound
// (DESIGN.md section 3).  The round-1 kernel kept the sorted pool in LDS: every expansion paid a 3-level 8-ary search
// of EVERY neighbour's key (23 dependent LDS reads), a full read-shift-write pass over the pool and a full pass to
// rebuild its masks — all proportional to the pool size (5 100 cycles per expansion at 256 entries, ~20 000 at
// 1 024).  Here the pool is a sorted array of 64-bit keys spread over two 32-dword register vectors (chunk t, lane l
// = position 64 t + l; empty slots hold the minimum key):
//   * rank of a key = one compare against the vector of chunk-first keys + one compare against ONE chunk (ballots),
//     the duplicate test (same node already in the pool: this kernel keeps no visited set while searching) falls out
//     of the same two compares;
//   * insertion = one DPP wave-shift per chunk behind the insertion point (most keys land near the tail);
//   * chunks are addressed with a wave-uniform index (s_set_gpr_idx / movrel), so the cost of an expansion does not
//     depend on the pool's capacity (2 048 entries = 64 VGPRs);
//   * LDS holds nothing but the query's look-up table: 32 KB at PQ-32 -> 5 resident queries per CU instead of 3-4.
// The expansion log goes to a small per-workgroup HBM scratch (persistent grid: one workgroup per LDS slot, queries
// are dequeued with an atomic counter).
#include "jv_dev_common.h"

typedef int i32x32 __attribute__((ext_vector_type(32)));

#define PQR_CHUNKS 32
#define PQR_CAP (64 * PQR_CHUNKS)
#define KEYMIN_HI ((int)0x80000000)

__device__ __forceinline__ int dpp_wave_shr1(int lane0_value, int v) {
    // lane i (i >= 1) <- v[i - 1]; lane 0 keeps `lane0_value` (bound_ctrl off: lanes without a source keep `old`)
    return __builtin_amdgcn_update_dpp(lane0_value, v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ int64_t mk64(int hi, int lo) { return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo); }
__device__ __forceinline__ float hi_score(int hi) { return __int_as_float(hi ^ ((hi >> 31) & 0x7fffffff)); }
__device__ __forceinline__ int lo_node(int lo) { return (int)((~((uint32_t)lo >> 1)) & 0x7FFFFFFFu); }

// NCHT: row length in 64-float chunks known at compile time (rerank), 0 = any d
// NP:   fused-block passes (1: R * lanes-per-node <= 64; 4: up to 4 passes)
// FAST: pq_M % 16 == 0 and not cosine (only the unmasked look-up is compiled)
template <int NCHT, int NP, bool FAST>
__device__ void search_one_pqr(const JvIndexDev& ix, const JvSearchArgs& a, const int qi, unsigned char* smem, int32_t* explog) {
    const int lane = threadIdx.x;
    const int rk = a.rk, topK = a.topK;
    const int M = ix.pq_M, R = ix.R, lpn = ix.pq_lanes, cs = ix.pq_code_stride;
    float* lut = (float*)smem;  // [M][256]; later: visited-count hash, then rerank scratch
    const int lut_bytes = M * 256 * (int)sizeof(float);
    float* qc_lds = (float*)(smem + a.pqr_qc_off);  // centred query, only during the LUT build (may alias the LUT's tail)
    const int log_cap = a.pqr_log_cap;
    const int pool_limit = a.cand_cap - R;  // one expansion adds at most R keys

    int32_t* o_nodes = a.out_nodes + (size_t)qi * topK;
    int32_t* o_docs = a.out_docs ? a.out_docs + (size_t)qi * topK : nullptr;
    float* o_scores = a.out_scores + (size_t)qi * topK;

    // ---- centred query -> LUT ----
    const float* qg = a.queries + (size_t)qi * ix.d;
    for (int i = lane; i < ix.nch * 64; i += JV_WAVE) {
        float v = i < ix.d ? qg[i] : 0.0f;
        if (ix.pq_centroid && i < ix.d) v = v - ix.pq_centroid[i];
        qc_lds[i] = v;
    }
    float qnorm2 = 0.0f;
    if (ix.sim == 2) {  // |q|^2 of the UNcentred query (cosine never centres: no centroid)
        __syncthreads();
        qnorm2 = query_norm2(ix, qc_lds, lane);
        qnorm2 = __shfl(qnorm2, 0, JV_WAVE);
    }
    __syncthreads();
    build_lut<24>(ix, qc_lds, lut, lane);
    __syncthreads();

    const int my_c = lane & (lpn - 1);
    const int jpp = JV_WAVE / lpn;
    const int npass = NP == 1 ? 1 : (R * lpn + JV_WAVE - 1) / JV_WAVE;
    const int my_slot = lane / lpn;
    const bool my_chunk = my_c * 16 < M;
    const bool full16 = (M & 15) == 0;
    auto adc_score = [&](const u32x4 cw, bool have) -> float {
        if (FAST) {
            const float s_ = adc_chunk<true>(lut, cw, my_c * 16, M);
            return map_score(ix.sim == 0 ? 0 : 1, lanes_tree_sum(have ? s_ : 0.0f, lpn));
        }
        float s = full16 ? adc_chunk<true>(lut, cw, my_c * 16, M) : adc_chunk<false>(lut, cw, my_c * 16, M);
        float na = 0.0f;
        if (ix.sim == 2) na = adc_chunk<false>(ix.pq_norm_lut, cw, my_c * 16, M);
        s = lanes_tree_sum(have ? s : 0.0f, lpn);
        if (ix.sim == 2) {
            na = lanes_tree_sum(have ? na : 0.0f, lpn);
            return map_score(2, s / sqrtf(qnorm2 * na));
        }
        return map_score(ix.sim, s);
    };

    // ---- the pool: sorted descending, position p = 64 * chunk + lane; bit 0 of a key = "not expanded yet" ----
    i32x32 L, H;
#pragma unroll
    for (int t = 0; t < PQR_CHUNKS; t++) {
        L[t] = 0;
        H[t] = KEYMIN_HI;
    }
    int fl = 0, fh = KEYMIN_HI;  // lane t: the FIRST (largest) key of chunk t (minimum key when the chunk is empty)
    int np = 0, nexp = 0, expanded = 0, lo_un = 0;
    int why = 0;
    float bscore = 0.0f;  // score of the rk-th best entry once np >= rk
    {
        const int ep = ix.entry;
        u32x4 cw = (u32x4){0, 0, 0, 0};
        if (lane < lpn && my_chunk) cw = *(const u32x4*)(ix.pq_codes + (size_t)ep * cs + my_c * 16);
        float s = adc_score(cw, lane < lpn && my_chunk);
        s = __shfl(s, 0, JV_WAVE);
        const int64_t k0 = make_pool_key(s, ep);
        if (lane == 0) {
            L[0] = (int)(uint32_t)(k0 & 0xFFFFFFFFll);
            H[0] = (int)(k0 >> 32);
            fl = L[0];
            fh = H[0];
        }
        np = 1;
        if (rk <= 1) bscore = s;
    }

    int pf_node = -1;
    int pf_nn[NP];
    u32x4 pf_cw[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) pf_nn[ps] = -1, pf_cw[ps] = (u32x4){0, 0, 0, 0};
    const int cb_rk = __builtin_amdgcn_readfirstlane((rk - 1) >> 6), lb_rk = __builtin_amdgcn_readfirstlane((rk - 1) & 63);
    STAMP_DECL
    STAMP(7)  // LUT build + entry point
    while (true) {
        // ---- best and runner-up unexpanded entries (every position < lo_un is expanded) ----
        const int topc = (np - 1) >> 6;
        int t1 = __builtin_amdgcn_readfirstlane(lo_un >> 6);
        unsigned long long m1 = 0ull;
        for (; t1 <= topc; t1 = __builtin_amdgcn_readfirstlane(t1 + 1)) {
            m1 = __ballot((L[t1] & 1) != 0);
            if (m1) break;
        }
        if (t1 > topc) break;
        const int b1 = __ffsll((long long)m1) - 1;
        const int idx = (t1 << 6) + b1;
        const int l1v = L[t1], h1v = H[t1];
        const int pk_lo = __builtin_amdgcn_readlane(l1v, b1), pk_hi = __builtin_amdgcn_readlane(h1v, b1);
        m1 &= m1 - 1ull;
        int t2 = t1;
        if (!m1) {
            for (t2 = __builtin_amdgcn_readfirstlane(t1 + 1); t2 <= topc; t2 = __builtin_amdgcn_readfirstlane(t2 + 1)) {
                m1 = __ballot((L[t2] & 1) != 0);
                if (m1) break;
            }
        }
        int c2 = -1;
        if (m1) c2 = lo_node(__builtin_amdgcn_readlane(L[t2], __ffsll((long long)m1) - 1));
        const float sc = hi_score(pk_hi);
        if (sc < a.threshold) {  // a node the two-queue form would expand but not collect: general path
            why = 1;
            break;
        }
        // strict-admission tie (DESIGN.md "Single-pool search"): the expanded entries scoring >= the candidate already
        // fill the result queue and the candidate ranks inside the top rerankK -> the two-queue form decides
        if (expanded >= rk && idx < rk) {
            int ge = idx;
            for (int tt = t1; tt <= topc; tt = __builtin_amdgcn_readfirstlane(tt + 1)) {
                const unsigned long long eq = __ballot(H[tt] == pk_hi);
                unsigned long long ex = eq & ~__ballot((L[tt] & 1) != 0);
                if (tt == t1) ex &= ~((2ull << b1) - 1ull);  // positions behind the candidate only
                ge += __popcll(ex);
                if (!(eq >> 63)) break;  // the equal-score run ends inside this chunk
            }
            if (ge >= rk) {
                why = 5;
                break;
            }
        }
        const int c = lo_node(pk_lo);
        int nnp[NP];
        u32x4 cwp[NP];
#pragma unroll
        for (int ps = 0; ps < NP; ps++) {
            nnp[ps] = -1;
            cwp[ps] = (u32x4){0, 0, 0, 0};
            if (ps < npass) {
                const int j = ps * jpp + my_slot;
                if (c == pf_node) {  // the prefetch loaded with clamped indices: mask here
                    nnp[ps] = j < R ? pf_nn[ps] : -1;
                    if (j < R && my_chunk) cwp[ps] = pf_cw[ps];
                } else {
                    nnp[ps] = j < R ? ix.adj[(size_t)c * R + j] : -1;
                    if (j < R && my_chunk) cwp[ps] = *(const u32x4*)(ix.pq_fused + ((size_t)c * R + j) * cs + my_c * 16);
                }
            }
        }
        STAMP(0)  // find + pool reads
        // start the runner-up's fetch now, UNCONDITIONALLY (clamped indices): a fixed number of younger loads lets the
        // wait for this expansion's block leave them in flight
        pf_node = c2;
        {
            const int c2e = c2 >= 0 ? c2 : c;
#pragma unroll
            for (int ps = 0; ps < NP; ps++) {
                if (NP == 1 || ps < npass) {
                    const int j = min(ps * jpp + my_slot, R - 1);
                    pf_nn[ps] = ix.adj[(size_t)c2e * R + j];
                    pf_cw[ps] = *(const u32x4*)(ix.pq_fused + ((size_t)c2e * R + j) * cs + (my_chunk ? my_c * 16 : 0));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nexp >= log_cap) {
            why = 2;
            break;
        }
        // mark the entry expanded; log the node
        L[t1] = (lane == b1) ? (l1v & ~1) : l1v;
        if (lane == 0) explog[nexp] = c;
        nexp++;
        lo_un = idx + 1;
        STAMP(1)
        // ---- ADC of all R stored neighbours; pass ps delivers its scores to the lanes whose chunk index is ps ----
        float score = 0.0f;
        int nn = -1;
#pragma unroll
        for (int ps = 0; ps < NP; ps++) {
            if (ps < npass) {
                const float sp = adc_score(cwp[ps], nnp[ps] >= 0 && my_chunk);
                const float sp_m = ps == 0 ? sp : __shfl(sp, lane - ps, JV_WAVE);
                const int nn_m = ps == 0 ? nnp[0] : __shfl(nnp[ps], lane - ps, JV_WAVE);
                if (my_c == ps) {
                    score = sp_m;
                    nn = nn_m;
                }
            }
        }
        expanded++;
#ifdef JV_STAMPS
        asm volatile("" ::"v"(score));
#endif
        STAMP(2)  // ADC + prefetch issue
        bool keep = nn >= 0 && my_c < npass;
        if (np >= rk && score < bscore) keep = false;  // below the boundary for good
        const int64_t v = make_pool_key(score, nn);
        const int vlo = (int)(uint32_t)(v & 0xFFFFFFFFll), vhi = (int)(v >> 32);
        unsigned long long km = __ballot(keep);
        bool inserted = false;
        STAMP_COUNT(8, __popcll(km))  // candidates that pass the boundary test
        while (km) {
            const int j = __ffsll((long long)km) - 1;
            km &= km - 1ull;
            const int klo = __builtin_amdgcn_readlane(vlo, j), khi = __builtin_amdgcn_readlane(vhi, j);
            const int64_t K = mk64(khi, klo), K0 = mk64(khi, klo & ~1);
            // level 1: chunk-first keys (lane t = chunk t).  "same node" = equal up to bit 0.
            const int64_t F = mk64(fh, fl);
            const unsigned long long gtp = __ballot(F > K), gep = __ballot(F >= K0);
            if (gtp != gep) {  // the node is already in the pool (as a chunk's first entry)
                STAMP_COUNT(9, 1)
                continue;
            }
            int c1 = __popcll(gtp) - 1;
            c1 = __builtin_amdgcn_readfirstlane(c1 < 0 ? 0 : c1);
            // level 2: the one chunk that holds the rank boundary
            const int pl = L[c1], ph = H[c1];
            const int64_t P = mk64(ph, pl);
            const unsigned long long gt = __ballot(P > K), ge = __ballot(P >= K0);
            if (gt != ge) {  // already in the pool
                STAMP_COUNT(10, 1)
                continue;
            }
            STAMP_COUNT(11, 1)                                                             // inserts
            STAMP_COUNT(12, __builtin_amdgcn_readfirstlane(np >> 6) - (c1 + (__popcll(gt) >> 6)))  // whole chunks shifted
            const int r = (c1 << 6) + __popcll(gt);
            // ---- insert at position r: chunks behind it shift by one lane (carry = the previous chunk's last entry);
            // chunk registers are addressed with a wave-uniform index (s_set_gpr_idx), in place ----
            const int tr = __builtin_amdgcn_readfirstlane(r >> 6), b = r & 63;  // (tr == c1 + 1 when all of chunk c1 ranks ahead)
            int t = __builtin_amdgcn_readfirstlane(np >> 6);
            if (t >= PQR_CHUNKS) t = PQR_CHUNKS - 1;  // (cannot happen: np <= pool_limit + R <= capacity)
            int curl = L[t], curh = H[t];
            while (t > tr) {
                const int tm1 = __builtin_amdgcn_readfirstlane(t - 1);
                const int prl = L[tm1], prh = H[tm1];
                const int cl = __builtin_amdgcn_readlane(prl, 63), ch = __builtin_amdgcn_readlane(prh, 63);
                L[t] = dpp_wave_shr1(cl, curl);
                H[t] = dpp_wave_shr1(ch, curh);
                if (lane == t) {
                    fl = cl;
                    fh = ch;
                }
                curl = prl;
                curh = prh;
                t = tm1;
            }
            {
                const int xl = dpp_wave_shr1(klo, curl), xh = dpp_wave_shr1(khi, curh);
                L[tr] = lane > b ? xl : (lane == b ? klo : curl);
                H[tr] = lane > b ? xh : (lane == b ? khi : curh);
                if (b == 0 && lane == tr) {
                    fl = klo;
                    fh = khi;
                }
            }
            np++;
            lo_un = lo_un < r ? lo_un : r;
            inserted = true;
        }
        STAMP(3)  // rank + duplicate test + insertion of the new keys
        if (inserted && np >= rk) {
            // boundary = the rk-th best entry; entries behind it stay only while they tie with its score
            const int bhi = __builtin_amdgcn_readlane(H[cb_rk], lb_rk);
            bscore = hi_score(bhi);
            if (np > rk) {
                const int oldtop = (np - 1) >> 6;
                int ties = 0;
                for (int tt = cb_rk; tt <= oldtop; tt = __builtin_amdgcn_readfirstlane(tt + 1)) {
                    const unsigned long long eq = __ballot(H[tt] == bhi);
                    unsigned long long behind = eq;
                    if (tt == cb_rk) behind &= ~((2ull << lb_rk) - 1ull);
                    ties += __popcll(behind);
                    if (!(eq >> 63)) break;
                }
                const int nnew = rk + ties;
                for (int tt = __builtin_amdgcn_readfirstlane(nnew >> 6); tt <= oldtop; tt = __builtin_amdgcn_readfirstlane(tt + 1)) {
                    const bool drop = (tt << 6) + lane >= nnew;
                    L[tt] = drop ? 0 : L[tt];
                    H[tt] = drop ? KEYMIN_HI : H[tt];
                    if ((tt << 6) >= nnew && lane == tt) {
                        fl = 0;
                        fh = KEYMIN_HI;
                    }
                }
                np = nnew;
                if (np > pool_limit) {
                    why = 3;
                    break;
                }
            }
        }
        STAMP(4)  // boundary + trim
    }
    STAMP(5)

    int visited = 0;
    if (why == 0) {
        // ---- jvector's visitedCount: distinct neighbours of the expanded nodes, entry point excluded.  The hash set
        // lives where the LUT was; node ids are split into `parts` hash classes counted one after the other when one
        // table cannot hold them all (adjacency rows are re-read once per class). ----
        uint32_t* vh = (uint32_t*)lut;
        int vslots = 1;
        while (vslots * 2 * 4 <= lut_bytes) vslots <<= 1;
        const uint32_t vmask = (uint32_t)vslots - 1u;
        const int vshift = 32 - (31 - __clz(vslots));
        const int vlimit = (vslots / 16) * 13;
        int parts = 1;
        while (parts < 64 && (long long)nexp * 6 > (long long)vlimit * parts) parts <<= 1;
        // the log was written by lane 0 and is read back by every lane: drain the stores, read with L1-bypassing loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        const int rows_per = JV_WAVE / R > 0 ? JV_WAVE / R : 1;
        bool again = true;
        while (again && why == 0) {
            again = false;
            visited = 0;
            for (int p = 0; p < parts && !again; p++) {
                __syncthreads();
                for (int i = lane; i < vslots; i += JV_WAVE) vh[i] = HASH_EMPTY;
                __syncthreads();
                auto part_of = [&](uint32_t node) -> int { return (int)(((node * 0x85EBCA6Bu) >> 20) & (uint32_t)(parts - 1)); };
                if (lane == 0 && part_of((uint32_t)ix.entry) == p) visited_insert_lds(vh, vmask, vshift, (uint32_t)ix.entry);
                __syncthreads();
                int cnt = 0;
                if (R <= JV_WAVE) {
                    constexpr int VB = 8;  // adjacency batches per group; the NEXT group's rows are in flight while this one probes
                    auto load_group = [&](int e0, int (&dst)[VB]) {
#pragma unroll
                        for (int u = 0; u < VB; u++) {
                            const int e = min(e0 + u * rows_per + lane / R, nexp - 1);
                            dst[u] = ix.adj[(size_t)__hip_atomic_load(&explog[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * R + (lane % R)];
                        }
                    };
                    int nb_next[VB];
                    load_group(0, nb_next);
                    for (int e0 = 0; e0 < nexp; e0 += rows_per * VB) {
                        int nb[VB];
#pragma unroll
                        for (int u = 0; u < VB; u++) {
                            const int e = e0 + u * rows_per + lane / R;
                            nb[u] = (e < nexp && lane < rows_per * R) ? nb_next[u] : -1;
                            if (nb[u] >= 0 && parts > 1 && part_of((uint32_t)nb[u]) != p) nb[u] = -1;
                        }
                        load_group(e0 + rows_per * VB, nb_next);
                        int pending = 0;
#pragma unroll
                        for (int u = 0; u < VB; u++) pending += __popcll(__ballot(nb[u] >= 0));
                        if (cnt + pending > vlimit) {
                            again = true;
                            break;
                        }
                        uint32_t hh[VB];
                        bool pend[VB];
#pragma unroll
                        for (int u = 0; u < VB; u++) {
                            pend[u] = nb[u] >= 0;
                            hh[u] = ((uint32_t)nb[u] * 0x9E3779B1u) >> vshift;
                        }
                        for (;;) {
                            uint32_t oldv[VB];
#pragma unroll
                            for (int u = 0; u < VB; u++) oldv[u] = pend[u] ? atomicCAS(&vh[hh[u]], HASH_EMPTY, (uint32_t)nb[u]) : 0u;
                            bool more = false;
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const bool fresh = pend[u] && oldv[u] == HASH_EMPTY;
                                cnt += __popcll(__ballot(fresh));
                                if (pend[u]) {
                                    if (fresh || oldv[u] == (uint32_t)nb[u]) pend[u] = false;
                                    else hh[u] = (hh[u] + 1) & vmask, more = true;
                                }
                            }
                            if (!__any(more)) break;
                        }
                    }
                } else {
                    for (int e0 = 0; e0 < nexp && !again; e0++) {
                        for (int cb = 0; cb < R; cb += JV_WAVE) {
                            if (cnt + JV_WAVE > vlimit) {
                                again = true;
                                break;
                            }
                            int nb = (cb + lane < R) ? ix.adj[(size_t)__hip_atomic_load(&explog[e0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * R + cb + lane] : -1;
                            if (nb >= 0 && parts > 1 && part_of((uint32_t)nb) != p) nb = -1;
                            bool is_new = false;
                            if (nb >= 0) is_new = visited_insert_lds(vh, vmask, vshift, (uint32_t)nb);
                            cnt += __popcll(__ballot(is_new));
                        }
                    }
                }
                visited += cnt;
            }
            if (again) {
                parts <<= 1;
                if (parts > 64) why = 4;
            }
        }
        __syncthreads();
    }
    STAMP(6)  // visited-count pass
    STAMP_FLUSH
    const int nres = np < rk ? np : rk;
    int above = 0;
    if (why == 0) {
        for (int t = 0; (t << 6) < nres; t = __builtin_amdgcn_readfirstlane(t + 1))
            above += __popcll(__ballot((t << 6) + lane < nres && hi_score(H[t]) >= a.rerank_floor));
        // rerankFloor above every approximate score AND a tie at the best one: jvector rescores the first best entry of
        // its result heap's array, which only the HBM-scratch rung reconstructs (replay_first_best)
        if (above == 0 && nres >= 2 && __builtin_amdgcn_readlane(H[0], 0) == __builtin_amdgcn_readlane(H[0], 1)) why = 6;
    }
    if (why != 0) {
        if (lane == 0) {
            a.out_flags[qi] = (int32_t)(JV_FLAG_OVERFLOW | ((uint32_t)why << 8));
            a.out_count[qi] = 0;
        }
        for (int i = lane; i < topK; i += JV_WAVE) {
            o_nodes[i] = -1;
            if (o_docs) o_docs[i] = -1;
            o_scores[i] = 0.0f;
        }
        return;
    }
    // ---- rerank (NodeQueue.rerank) with the exact scorer; scratch lives where the LUT was ----
    float* q_lds = (float*)smem;
    size_t roff = (size_t)ix.nch * 64 * sizeof(float);
    float* todo_score = (float*)(smem + roff);
    roff += JV_TODO * sizeof(float);
    int32_t* todo = (int32_t*)(smem + roff);
    roff += JV_TODO * sizeof(int32_t);
    int64_t* fin = (int64_t*)(smem + roff);  // [rk]
    roff += (size_t)((rk + 1) & ~1) * sizeof(int64_t);
    int32_t* cnodes = (int32_t*)(smem + roff);  // [nres] the nodes to rescore (-1 = below rerankFloor)
    // the pool leaves the registers here: the exact scorer below keeps up to 192 VGPRs of row data in flight
    for (int t = 0; (t << 6) < nres; t++) {
        const int tt = __builtin_amdgcn_readfirstlane(t);
        const int i = (tt << 6) + lane;
        const bool take = above > 0 ? hi_score(H[tt]) >= a.rerank_floor : i == 0;  // position 0 is the best approximate entry
        if (i < nres) cnodes[i] = take ? lo_node(L[tt]) : -1;
    }
    for (int i = lane; i < ix.nch * 64; i += JV_WAVE) q_lds[i] = i < ix.d ? qg[i] : 0.0f;
    __syncthreads();
    if (ix.sim == 2) qnorm2 = query_norm2(ix, q_lds, lane), qnorm2 = __shfl(qnorm2, 0, JV_WAVE);
    int nfin = 0, reranked = 0;
    for (int b0 = 0; b0 < nres; b0 += JV_WAVE) {
        const int i = b0 + lane;
        const int node = i < nres ? cnodes[i] : -1;
        const bool take = node >= 0;
        const unsigned long long tm = __ballot(take);
        const int m = __popcll(tm);
        if (take) todo[__popcll(tm & ((1ull << lane) - 1ull))] = node;
        __syncthreads();
        if (m > 0) {
            score_rows<NCHT, 1>(ix, q_lds, todo, m, todo_score, qnorm2, 1.0f, lane);
            __syncthreads();
            if (lane < m) fin[nfin + lane] = make_key(todo_score[lane], todo[lane]);
            nfin += m;
            reranked += m;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (; cnt < topK && nfin > 0; cnt++) {
        int64_t bk;
        int bidx;
        scan_max(fin, nfin, lane, bk, bidx);
        if (lane == 0) {
            const int node = key_node(bk);
            o_nodes[cnt] = node;
            if (o_docs) o_docs[cnt] = ix.ord2doc ? ix.ord2doc[node] : node;
            o_scores[cnt] = key_score(bk);
            fin[bidx] = fin[nfin - 1];
        }
        nfin--;
        __syncthreads();
    }
    for (int i = cnt + lane; i < topK; i += JV_WAVE) {
        o_nodes[i] = -1;
        if (o_docs) o_docs[i] = -1;
        o_scores[i] = 0.0f;
    }
    if (lane == 0) {
        a.out_count[qi] = cnt;
        int32_t* st = a.out_stats + (size_t)qi * 4;
        st[0] = visited;
        st[1] = reranked;
        st[2] = expanded;
        st[3] = expanded;
        a.out_flags[qi] = 0;
    }
}

// Persistent grid: one workgroup per resident LDS slot, queries dequeued in order.
template <int NCHT, int NP, bool FAST>
__global__ __launch_bounds__(JV_WAVE) void jv_search_pqr_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* explog = a.pqr_log + (size_t)blockIdx.x * (size_t)a.pqr_log_cap;
    for (;;) {
        int qi = 0;
        if (threadIdx.x == 0) qi = atomicAdd(a.pqr_counter, 1);
        qi = __builtin_amdgcn_readfirstlane(qi);
        if (qi >= a.nq) break;
        search_one_pqr<NCHT, NP, FAST>(ix, a, qi, smem, explog);
        __syncthreads();
    }
}

typedef void (*pqr_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQR_ROW(NP, FAST) \
    { jv_search_pqr_kernel<0, NP, FAST>, jv_search_pqr_kernel<2, NP, FAST>, jv_search_pqr_kernel<12, NP, FAST>, jv_search_pqr_kernel<24, NP, FAST> }
// [0 single-pass | 1 multi-pass | 2 single-pass FAST | 3 multi-pass FAST][nch slot]
static const pqr_kernel_t g_pqr_kernels[4][4] = {JV_PQR_ROW(1, false), JV_PQR_ROW(4, false), JV_PQR_ROW(1, true), JV_PQR_ROW(4, true)};

static int pqr_nch_slot(const JvIndexDev* ix) {
    if (ix->stride != ix->nch * 64) return 0;
    return ix->nch == 2 ? 1 : ix->nch == 12 ? 2 : ix->nch == 24 ? 3 : 0;
}

extern "C" hipError_t jvk_pqr_set_max_lds(int bytes) {
    for (int v = 0; v < 4; v++)
        for (int s = 0; s < 4; s++) {
            hipError_t e = hipFuncSetAttribute((const void*)g_pqr_kernels[v][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

extern "C" int jvk_pqr_capacity(void) { return PQR_CAP; }

// resident workgroups per CU for this index shape and LDS size (registers and LDS both limit it)
extern "C" int jvk_pqr_blocks_per_cu(const JvIndexDev* ix, int lds_bytes) {
    const int multi = ix->R * ix->pq_lanes > JV_WAVE ? 1 : 0;
    const int fast = (ix->pq_M % 16 == 0 && ix->sim != 2) ? 1 : 0;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)g_pqr_kernels[fast * 2 + multi][pqr_nch_slot(ix)], JV_WAVE,
                                                     (size_t)lds_bytes) != hipSuccess)
        return 1;
    return nb < 1 ? 1 : nb;
}

// blocks = resident workgroups (the host sizes the log scratch to it)
extern "C" hipError_t jvk_launch_search_pqr(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    const int multi = ix->R * ix->pq_lanes > JV_WAVE ? 1 : 0;
    const int fast = (ix->pq_M % 16 == 0 && ix->sim != 2) ? 1 : 0;
    g_pqr_kernels[fast * 2 + multi][pqr_nch_slot(ix)]<<<blocks, JV_WAVE, lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}
