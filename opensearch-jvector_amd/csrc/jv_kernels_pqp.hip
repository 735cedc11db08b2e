// jv_kernels_pqp.hip — the headline kernel: PQ approximate search on the FUSED layout, sorted candidate pool in LDS,
// persistent grid; followed by jvector's visitedCount reconstruction and the exact rerank (gfx950 / CDNA4).
//
// What it computes is GraphSearcher.search as called from J/JVectorReader.java:165-173 for the
// case "PQ provider, no filter, threshold <= 0, flat graph"; results, scores and the four counters are bit-equal to
// oracle/jv_oracle.c (tests/test_gpu_parity.py).  Everything the kernel cannot hold is flagged and re-run by the
// generic ladder (jv_kernels.hip), so exactness never depends on a capacity.
//
// One wave runs one query.  A lone wave per SIMD is bound by dependent-instruction latency, not by bytes: what an
// expansion costs is the length of its chain of dependent steps.  Measured alternatives:
//   * round 1 (jv_search_pqf_kernel): LDS pool, but every expansion touched the WHOLE pool — masks in scalar registers
//     over all 64-entry chunks, a full read-shift-write pass, a full pass to rebuild the masks: 5 100 cycles per
//     expansion at 256 entries, ~20 000 at 1 024 (rerankK = 900);
//   * pool in registers (jv_kernels_pqr.hip): insertion by DPP wave shifts is cheap, but the kernel keeps no visited
//     set while searching, so ~20 of the 32 neighbours of an expansion are re-encounters that must be recognised in
//     the pool — one at a time through scalar compares (registers cannot be indexed per lane): 340 cycles each;
//   * this kernel: the pool stays in LDS so that all 32 neighbours find their rank (and their duplicates) at once
//     with a 3-4 level 8-ary search per LANE, and everything else only touches the chunks it changes: the best
//     unexpanded entry is looked for from a moving lower bound, a merge shifts only the chunks behind the first
//     insertion point (uniform shift once past the last one), the boundary test reads one chunk.
// LDS per query = look-up table + pool; the expansion log goes to a per-workgroup HBM scratch (persistent grid: one
// workgroup per resident LDS slot, queries are dequeued with an atomic counter).
#include "jv_dev_common.h"

typedef int i32x32 __attribute__((ext_vector_type(32)));

__device__ __forceinline__ float hi_score(int hi) { return __int_as_float(hi ^ ((hi >> 31) & 0x7fffffff)); }
// pool keys of this kernel (n < 2^30): [63:32] sortable score | [31:2] ~node | bit 1 "not rejected" | bit 0 "not expanded yet".
// (A fresh key has both low bits set, so whatever entry the same node already has in the pool ranks at or behind the fresh
// key: the duplicate test finds it inside the window the rank search ends on.)
// The order between two different nodes is NodeQueue's (score desc, ordinal asc); the two low bits belong to the entry.
// "rejected" = jvector's strict admission (GraphSearcher.addTopCandidate): a popped candidate whose score merely EQUALS the
// worst result of a full result queue is expanded but not admitted.  Such an entry stays in the pool (re-encounters must
// still recognise the node) but is taken out before the results are read.  All rejected entries tie with the boundary
// score, so they neither move the boundary nor survive its next rise.
__device__ __forceinline__ int64_t pqp_key(float score, int node) {
    int32_t b = __float_as_int(score);
    int32_t s = b ^ ((b >> 31) & 0x7fffffff);
    return (int64_t)(((uint64_t)(uint32_t)s << 32) | ((uint64_t)((uint32_t)(~node) & 0x3FFFFFFFu) << 2) | 3ull);
}
__device__ __forceinline__ int lo_node(int lo) { return (int)((~((uint32_t)lo >> 2)) & 0x3FFFFFFFu); }

// one level of the per-lane rank search: FAN - 1 pivots at lo + k * BLK + BLK - 1; entries are sorted descending and
// slots beyond the pool hold the minimum key, so "pivot > v" is monotone and needs no bound check
template <int BLK, int FAN>
__device__ __forceinline__ int rank_level(const int64_t* pool, int lo, int last, int64_t v) {
    int64_t p[FAN - 1];
#pragma unroll
    for (int k = 0; k < FAN - 1; k++) p[k] = pool[min(lo + k * BLK + BLK - 1, last)];
    int c = 0;
#pragma unroll
    for (int k = 0; k < FAN - 1; k++) c += p[k] > v ? 1 : 0;
    return lo + c * BLK;
}

// NCHT: row length in 64-float chunks known at compile time (rerank), 0 = any d
// NP:   fused-block passes (1: R * lanes-per-node <= 64; 4: up to 4 passes)
// FAST: pq_M % 16 == 0 and not cosine (only the unmasked look-up is compiled)
// CAPK: pool capacity class: 0 -> <= 512 entries, 1 -> <= 1 024, 2 -> <= 2 048, 3 -> <= 4 096
// LUTR: the look-up table lives in REGISTERS (PQ-32, FAST, single pass only): lutr[m][e], lane l = lut[m][4 l + e]; a
//       look-up is ds_bpermute (lane = code >> 2) of the four e-registers + a bit-select by code & 3.  Costs ~3x the
//       instructions of an LDS gather, but LDS then only holds the pool: 8 resident queries per CU (two waves per SIMD
//       fill each other's stalls) instead of 3-4.
template <int NCHT, int NP, bool FAST, int CAPK, bool LUTR>
__device__ void search_one_pqp(const JvIndexDev& ix, const JvSearchArgs& a, const int qi, unsigned char* smem, int32_t* explog) {
    const int lane = threadIdx.x;
    const int rk = a.rk, topK = a.topK;
    const int M = ix.pq_M, R = ix.R, lpn = ix.pq_lanes, cs = ix.pq_code_stride;
    float* lut = (float*)smem;  // [M][256]; later: visited-count hash, then rerank scratch
    const int lut_bytes = M * 256 * (int)sizeof(float);
    float* qc_lds = (float*)(smem + a.pqp_qc_off);  // centred query, only during the LUT build (may alias the LUT's tail)
    const int log_cap = a.pqp_log_cap;

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
    float lutr[LUTR ? 32 : 1][4];
    if (LUTR) {
        const bool l2 = ix.sim == 0;
#pragma unroll
        for (int m = 0; m < 32; m++) {  // (same fmaf chains as build_lut, results kept in registers)
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
#pragma unroll
            for (int e = 0; e < 4; e++) lutr[LUTR ? m : 0][e] = acc4[e];
        }
    } else {
        build_lut<24>(ix, qc_lds, lut, lane);
    }
    __syncthreads();

    const int my_c = lane & (lpn - 1);
    const int jpp = JV_WAVE / lpn;
    const int npass = NP == 1 ? 1 : (R * lpn + JV_WAVE - 1) / JV_WAVE;
    const int my_slot = lane / lpn;
    const bool my_chunk = my_c * 16 < M;
    const bool full16 = (M & 15) == 0;
    auto adc_regs = [&](const u32x4 cw) -> float {  // this lane's 16 subspaces (chunk my_c), summed left to right like adc_chunk
        float sum = 0.0f;
        // four look-ups (32 ds_bpermute) are issued back to back before the first result is consumed: one LDS round trip
        // per group instead of one per register (the scheduler otherwise serialises them to save registers)
#pragma unroll
        for (int g4 = 0; g4 < 16; g4 += 4) {
            int t0[4][4], t1[4][4];
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                const int i = g4 + ii;
                const int w = (int)cw[i >> 2];
                const int addr = (int)(((uint32_t)w >> ((i & 3) * 8)) & 0xFFu);  // ds_bpermute reads lane (addr >> 2) & 63 = code >> 2
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    t0[ii][e] = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(lutr[LUTR ? i : 0][e]));
                    t1[ii][e] = __builtin_amdgcn_ds_bpermute(addr, __float_as_int(lutr[LUTR ? 16 + i : 0][e]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                const int i = g4 + ii;
                const int w = (int)cw[i >> 2];
                const int sh = (i & 3) * 8;
                const int m0 = __builtin_amdgcn_sbfe(w, sh, 1), m1 = __builtin_amdgcn_sbfe(w, sh + 1, 1);  // -1 / 0: code bits 0, 1
                int x[4];
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = my_c ? t1[ii][e] : t0[ii][e];
                const int s01 = (m0 & x[1]) | (~m0 & x[0]), s23 = (m0 & x[3]) | (~m0 & x[2]);
                sum = sum + __int_as_float((m1 & s23) | (~m1 & s01));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return sum;
    };
    auto adc_score = [&](const u32x4 cw, bool have) -> float {
        if (FAST) {
            const float s_ = LUTR ? adc_regs(cw) : adc_chunk<true>(lut, cw, my_c * 16, M);
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


    // ---- the pool: sorted descending; bit 0 of a key = "not expanded yet"; slots [np, cap] hold the minimum key ----
    const int cap = a.cand_cap;                   // entries; slot `cap` is a permanent sentinel
    int64_t* pool = (int64_t*)(smem + a.pqp_pool_off);
    for (int i = lane; i <= cap; i += JV_WAVE) pool[i] = KEY_MIN;
    int np = 0, nexp = 0, expanded = 0, lo_un = 0;
    int why = 0;
    int nrej = 0;         // rejected entries in the pool (all at the boundary score)
    float bscore = 0.0f;  // score of the rk-th best entry once np >= rk
    {
        const int ep = ix.entry;
        u32x4 cw = (u32x4){0, 0, 0, 0};
        if (lane < lpn && my_chunk) cw = *(const u32x4*)(ix.pq_codes + (size_t)ep * cs + my_c * 16);
        float s = adc_score(cw, lane < lpn && my_chunk);
        s = __shfl(s, 0, JV_WAVE);
        if (lane == 0) pool[0] = pqp_key(s, ep);
        np = 1;
        if (rk <= 1) bscore = s;
    }

    int pf_node = -1;
    int pf_nn[NP];
    u32x4 pf_cw[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) pf_nn[ps] = -1, pf_cw[ps] = (u32x4){0, 0, 0, 0};
    STAMP_DECL
    STAMP(7)  // LUT build + entry point
    while (true) {
        // ---- best and runner-up unexpanded entries (every position < lo_un is expanded; sentinels have bit 0 clear) ----
        int t1 = lo_un >> 6;
        int64_t e1 = 0;
        unsigned long long m1 = 0ull;
        for (; (t1 << 6) < np; t1++) {
            e1 = pool[min((t1 << 6) + lane, cap)];
            m1 = __ballot((e1 & 1ll) != 0);
            if (m1) break;
        }
        if (!m1) break;
        const int b1 = __ffsll((long long)m1) - 1;
        const int idx = (t1 << 6) + b1;
        const int e1lo = (int)(uint32_t)(e1 & 0xFFFFFFFFll), e1hi = (int)(e1 >> 32);
        const int pk_lo = __builtin_amdgcn_readlane(e1lo, b1), pk_hi = __builtin_amdgcn_readlane(e1hi, b1);
        int c2 = -1;
        {
            const unsigned long long m2 = m1 & (m1 - 1ull);
            if (m2) {
                c2 = lo_node(__builtin_amdgcn_readlane(e1lo, __ffsll((long long)m2) - 1));
            } else if (((t1 + 1) << 6) < np) {
                const int64_t e2 = pool[min(((t1 + 1) << 6) + lane, cap)];
                const unsigned long long m3 = __ballot((e2 & 1ll) != 0);
                if (m3) c2 = lo_node(__builtin_amdgcn_readlane((int)(uint32_t)(e2 & 0xFFFFFFFFll), __ffsll((long long)m3) - 1));
            }
        }
        const float sc = hi_score(pk_hi);
        if (sc < a.threshold) {  // a node the two-queue form would expand but not collect: general path
            why = 1;
            break;
        }
        // strict admission (DESIGN.md "Single-pool search"): when the ADMITTED entries scoring >= the candidate already fill
        // the result queue (the worst result ties with the candidate), jvector expands the candidate without admitting
        // it.  Admitted entries scoring >= sc = everything ahead of the candidate (all expanded) + the expanded entries of
        // its equal-score run behind it, minus the rejected ones (all of which tie with the boundary = sc here).
        bool reject = false;
        if (expanded >= rk && idx < rk + nrej) {
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
        if (a.visit_limit > 0 && expanded >= a.visit_limit) {  // Lucene discards this search (visited + expanded >= visitLimit)
            why = 15;
            break;
        }
        // mark the entry expanded; log the node
        if (lane == b1) ((int*)pool)[2 * idx] = pk_lo & (reject ? ~3 : ~1);
        nrej += reject ? 1 : 0;
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
        const int64_t v = pqp_key(score, nn);
        // ---- rank of every neighbour's key in the pool, all lanes at once; "same node" = equal up to bit 0 ----
        int rold;
        {
            int lo = 0;
            if (CAPK == 3) lo = rank_level<512, 8>(pool, lo, cap, v);
            if (CAPK == 2) lo = rank_level<512, 4>(pool, lo, cap, v);
            if (CAPK == 1) lo = rank_level<256, 4>(pool, lo, cap, v);
            if (CAPK == 1) lo = rank_level<64, 4>(pool, lo, cap, v);
            else lo = rank_level<64, 8>(pool, lo, cap, v);
            lo = rank_level<8, 8>(pool, lo, cap, v);
            int64_t p3[9];
#pragma unroll
            for (int k2 = 0; k2 < 9; k2++) p3[k2] = pool[min(lo + k2, cap)];
            int c3 = 0;
            bool dup = false;  // same node => same score => same key up to the expanded bit
#pragma unroll
            for (int k2 = 0; k2 < 9; k2++) {
                if (k2 < 8) c3 += p3[k2] > v ? 1 : 0;
                dup |= (p3[k2] | 3ll) == v;  // same node (whatever its expanded / rejected bits)
            }
            rold = lo + c3;
            if (dup) keep = false;
        }
        unsigned long long km = __ballot(keep);
        int nk = __popcll(km);
        STAMP_COUNT(8, __popcll(km))
        STAMP(3)  // boundary test + rank search + duplicate test
        if (nk == 0) { STAMP_COUNT(9, 1) }
        if (nk == 1) { STAMP_COUNT(10, 1) }
        if (nk > 0) {
            int rnew = 0, r_min, r_max;
            if (nk == 1) {
                // the common case: one new key, no ranks among new keys, every entry behind it moves up by one
                r_min = r_max = __builtin_amdgcn_readlane(rold, __ffsll((long long)km) - 1);
                for (int t = (np - 1) >> 6; t >= (r_min >> 6); t--) {
                    const int pos = (t << 6) + lane;
                    const int64_t e = pool[min(pos, cap)];
                    if (pos < np && pos >= r_min) pool[pos + 1] = e;
                }
            } else {
                // Bursts of ~8 new keys are the norm when there are several.  The kept keys (and their ranks among the old
                // entries) are compacted into a small LDS scratch, and every lane reads them back with wave-uniform
                // (broadcast) reads: rank among the new keys, twins (the same neighbour twice in one adjacency row of a
                // malformed graph: keep the first), and later the per-position shift counts — no scalar lane-by-lane loops.
                int64_t* const sk = (int64_t*)(smem + a.pqp_scratch_off);  // [64] kept keys (R <= 64), lane order
                int32_t* const sr = (int32_t*)(sk + 64);                    // [64] their ranks among the old entries
                for (int attempt = 0; attempt < 2; attempt++) {
                    const int my = __popcll(km & ((1ull << lane) - 1ull));
                    if (keep) {
                        sk[my] = v;
                        sr[my] = rold;
                    }
                    rnew = 0;
                    bool twin = false;
                    for (int j0 = 0; j0 < nk; j0 += 8) {
                        int64_t kj[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) kj[u] = sk[min(j0 + u, 63)];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const bool in = j0 + u < nk;
                            rnew += (in && kj[u] > v) ? 1 : 0;
                            twin |= in && kj[u] == v && j0 + u < my;
                        }
                    }
                    const unsigned long long km2 = __ballot(keep && !twin);
                    if (km2 == km) break;
                    keep = keep && !twin;
                    km = km2;
                    nk = __popcll(km);
                }
                const int lane_first = __ffsll((long long)__ballot(keep && rnew == 0)) - 1;     // largest new key
                const int lane_last = __ffsll((long long)__ballot(keep && rnew == nk - 1)) - 1;  // smallest new key
                r_min = __builtin_amdgcn_readlane(rold, lane_first);
                r_max = __builtin_amdgcn_readlane(rold, lane_last);
                STAMP(14)  // (diagnostic) ranks among the new keys
                // in-place shift, from the last occupied chunk down to the chunk of the first insertion point: an old
                // entry at position p moves up by the number of new keys that rank ahead of it
                const int t_mixed = r_max >> 6;  // chunks above it shift uniformly by nk
                for (int t = (np - 1) >> 6; t >= (r_min >> 6); t--) {
                    const int pos = (t << 6) + lane;
                    const int64_t e = pool[min(pos, cap)];
                    int cnt = nk;
                    if (t <= t_mixed) {
                        cnt = 0;
                        for (int j0 = 0; j0 < nk; j0 += 8) {
                            int rj[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) rj[u] = sr[min(j0 + u, 63)];
#pragma unroll
                            for (int u = 0; u < 8; u++) cnt += (j0 + u < nk && pos >= rj[u]) ? 1 : 0;
                        }
                    }
                    if (pos < np && cnt > 0) pool[pos + cnt] = e;
                }
            }
            if (keep) pool[rold + rnew] = v;
            STAMP_COUNT(11, nk)
            STAMP_COUNT(12, ((np - 1) >> 6) - (r_min >> 6) + 1)
            STAMP(4)  // ranks among the new keys + shift + insert
            // boundary = the rk-th best entry; entries behind it stay only while they tie with its score
            const int ntot = np + nk;
            np = ntot;
            if (ntot >= rk) {
                const int bhi = (int)(pool[rk - 1] >> 32);
                const float nb = hi_score(bhi);
                if (nb != bscore) nrej = 0;  // the boundary rose: every rejected entry (they tied with the old one) falls off below
                bscore = nb;
                if (ntot > rk) {
                    // ties directly behind the boundary stay; the pool has room for cap - rk - R of them (64 on the first
                    // launch: one chunk), the same query with more is redone by the wider second launch
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
                        why = 3;  // more boundary ties than this launch tracks
                        break;
                    }
                    np = rk + run;
                    for (int p0 = np; p0 < ntot; p0 += JV_WAVE)
                        if (p0 + lane < ntot) pool[p0 + lane] = KEY_MIN;
                }
            }
            lo_un = lo_un < r_min ? lo_un : r_min;
            STAMP(5)  // boundary + trim
        }
    }
    STAMP(5)
    if (why == 0 && nrej > 0) {
        // take the rejected entries out: what remains in front is jvector's result queue (ascending pass, every entry moves
        // down by the number of rejected entries ahead of it; a chunk is read completely before it is written)
        int carry = 0;
        for (int t = 0; (t << 6) < np; t++) {
            const int pos = (t << 6) + lane;
            const int64_t e = pool[min(pos, cap)];
            const bool rej = pos < np && !(e & 2ll);
            const unsigned long long rm = __ballot(rej);
            const int shift = carry + __popcll(rm & ((1ull << lane) - 1ull));
            if (pos < np && !rej && shift > 0) pool[pos - shift] = e;
            carry += __popcll(rm);
        }
        for (int p0 = np - carry; p0 < np; p0 += JV_WAVE)
            if (p0 + lane < np) pool[p0 + lane] = KEY_MIN;
        np -= carry;
    }

    // LUTR: the pool moves to registers so that the whole LDS allocation can serve as the visited-count hash set
    i32x32 PL, PH;
    if (LUTR && why == 0) {
#pragma unroll
        for (int t = 0; t < 32; t++) {
            PL[t] = 0;
            PH[t] = (int)0x80000000;
            if ((t << 6) < np) {
                const int64_t e = pool[min((t << 6) + lane, cap)];
                PL[t] = (int)(uint32_t)(e & 0xFFFFFFFFll);
                PH[t] = (int)(e >> 32);
            }
        }
        __syncthreads();
    }
    int visited = 0;
    if (why == 0) {
        // ---- jvector's visitedCount: distinct neighbours of the expanded nodes, entry point excluded.  The hash set
        // lives where the LUT was; node ids are split into `parts` hash classes counted one after the other when one
        // table cannot hold them all (adjacency rows are re-read once per class). ----
        uint32_t* vh = (uint32_t*)smem;
        const int hash_bytes = LUTR ? a.pqp_lds_bytes : lut_bytes;
        int vslots = 1;
        while (vslots * 2 * 4 <= hash_bytes) vslots <<= 1;
        const uint32_t vmask = (uint32_t)vslots - 1u;
        const int vshift = 32 - (31 - __clz(vslots));
        const int vlimit = (vslots / 16) * 13;
        // ~3.1 distinct neighbours per expansion are typical; a class that overflows its table doubles `parts` and starts over
        int parts = 1;
        while (parts < 64 && (long long)nexp * 7 > (long long)vlimit * parts * 2) parts <<= 1;
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
                if (R <= JV_WAVE && 64 % (rows_per * 8) == 0) {  // (a group of 8 batches must not straddle two 64-entry log chunks)
                    // The log is pulled into registers 2 048 entries at a time (coalesced loads, one latency), so a row
                    // fetch depends on ONE global load; two groups of rows are kept in flight ahead of the one that
                    // probes.  All loads are unconditional with clamped indices (fixed vmcnt distance).
                    constexpr int VB = 8;             // adjacency batches per group
                    const int G = rows_per * VB;      // log entries per group: 16 at R = 32 (divides 64: one log chunk)
                    for (int blk0 = 0; blk0 < nexp && !again; blk0 += 2048) {
                        const int nblk = min(2048, nexp - blk0);
                        i32x32 logv;
#pragma unroll
                        for (int g = 0; g < 32; g++) {
                            logv[g] = 0;
                            if (g * 64 < nblk)
                                logv[g] = __hip_atomic_load(&explog[blk0 + min(g * 64 + lane, nblk - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        const int e_last = (nblk - 1) / G * G;  // first entry of the last group
                        auto load_group = [&](int e0, int (&dst)[VB]) {
                            const int e0c = min(e0, e_last);
                            const int cur = logv[__builtin_amdgcn_readfirstlane(e0c >> 6)];
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const int e = min(e0c + u * rows_per + lane / R, nblk - 1);
                                const int node = __builtin_amdgcn_ds_bpermute((e & 63) << 2, cur);
                                dst[u] = ix.adj[(size_t)node * R + (lane % R)];
                            }
                        };
                        int q0[VB], q1[VB];
                        load_group(0, q0);
                        load_group(G, q1);
                        for (int e0 = 0; e0 < nblk; e0 += G) {
                            int nb[VB];
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const int e = e0 + u * rows_per + lane / R;
                                nb[u] = (e < nblk && lane < rows_per * R) ? q0[u] : -1;
                                if (nb[u] >= 0 && parts > 1 && part_of((uint32_t)nb[u]) != p) nb[u] = -1;
                                q0[u] = q1[u];
                            }
                            load_group(e0 + 2 * G, q1);
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
                    }
                } else {
                    for (int e0 = 0; e0 < nexp && !again; e0++) {  // (other row lengths: one row at a time)
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
    // ---- rerank scratch (where the LUT / the hash set was): query, todo lists, exact keys ----
    float* q_lds = (float*)smem;
    size_t roff = (size_t)ix.nch * 64 * sizeof(float);
    float* todo_score = (float*)(smem + roff);
    roff += JV_TODO * sizeof(float);
    int32_t* todo = (int32_t*)(smem + roff);
    roff += JV_TODO * sizeof(int32_t);
    int64_t* fin = (int64_t*)(smem + roff);  // [rk]
    const int64_t* rpool = pool;
    if (LUTR && why == 0) {
        // the pool returns from the registers; the exact keys overwrite it in place (entry i is written only after
        // the 64-entry batch containing position i has been read)
        int64_t* wp = (int64_t*)(smem + roff);
#pragma unroll
        for (int t = 0; t < 32; t++)
            if ((t << 6) < np) wp[(t << 6) + lane] = (int64_t)(((uint64_t)(uint32_t)PH[t] << 32) | (uint64_t)(uint32_t)PL[t]);
        rpool = wp;
        __syncthreads();
    }
    int above = 0;
    if (why == 0) {
        for (int i = lane; i < nres; i += JV_WAVE) above += key_score(rpool[i]) >= a.rerank_floor ? 1 : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) above += __shfl_xor(above, o, JV_WAVE);
        // rerankFloor above every approximate score AND a tie at the best one: jvector rescores the first best entry of
        // its result heap's array, which only the HBM-scratch rung reconstructs (replay_first_best)
        if (above == 0 && nres >= 2 && key_score(rpool[0]) == key_score(rpool[1])) why = 6;
    }
    if (why != 0) {
        if (lane == 0) {
            a.out_flags[qi] = why == 15 ? (int32_t)JV_FLAG_EARLY : (int32_t)(JV_FLAG_OVERFLOW | ((uint32_t)why << 8));
            a.out_count[qi] = 0;
            if (why == 15) {
                int32_t* st = a.out_stats + (size_t)qi * 4;
                st[0] = 0;
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
        return;
    }
    // ---- rerank (NodeQueue.rerank) with the exact scorer ----
    for (int i = lane; i < ix.nch * 64; i += JV_WAVE) q_lds[i] = i < ix.d ? qg[i] : 0.0f;
    __syncthreads();
    if (ix.sim == 2) qnorm2 = query_norm2(ix, q_lds, lane), qnorm2 = __shfl(qnorm2, 0, JV_WAVE);
    int nfin = 0, reranked = 0;
    for (int b0 = 0; b0 < nres; b0 += JV_WAVE) {
        const int i = b0 + lane;
        bool take = false;
        int node = 0;
        if (i < nres) {
            const int64_t k = rpool[i];
            node = lo_node((int)(uint32_t)(k & 0xFFFFFFFFll));
            take = above > 0 ? key_score(k) >= a.rerank_floor : i == 0;  // position 0 is the best approximate entry
        }
        const unsigned long long tm = __ballot(take);
        const int m = __popcll(tm);
        if (take) todo[__popcll(tm & ((1ull << lane) - 1ull))] = node;
        __syncthreads();
        if (m > 0) {
            score_rows<NCHT, JV_PQF_RERANK_UMUL>(ix, q_lds, todo, m, todo_score, qnorm2, 1.0f, lane);
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
    STAMP_DIRECT(13)  // rerank + top-K
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
template <int NCHT, int NP, bool FAST, int CAPK, bool LUTR = false>
__global__ __launch_bounds__(JV_WAVE, LUTR ? 2 : 1) void jv_search_pqp_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* explog = a.pqp_log + (size_t)blockIdx.x * (size_t)a.pqp_log_cap;
    // second launch (a.retry_only: wider pool, longer log): only the queries the first one flagged, found 8 flags at a time
    int base = 0;
    unsigned long long todo = 0ull;
    for (;;) {
        int qi = 0;
        if (a.retry_only) {
            while (!todo) {
                if (threadIdx.x == 0) base = atomicAdd(a.retry_counter, 8);  // (small chunks: two flagged queries rarely share one)
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= a.nq) return;
                const int qf = (threadIdx.x < 8 && base + (int)threadIdx.x < a.nq) ? a.out_flags[base + threadIdx.x] : 0;
                todo = __ballot(((uint32_t)qf & JV_FLAG_OVERFLOW) != 0);
            }
            qi = base + __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
        } else {
            if (threadIdx.x == 0) qi = atomicAdd(a.pqp_counter, 1);
            qi = __builtin_amdgcn_readfirstlane(qi);
            if (qi >= a.nq) break;
        }
        search_one_pqp<NCHT, NP, FAST, CAPK, LUTR>(ix, a, qi, smem, explog);
        __syncthreads();
    }
}

typedef void (*pqp_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQP_ROW(NP, FAST, CAPK) \
    { jv_search_pqp_kernel<0, NP, FAST, CAPK>, jv_search_pqp_kernel<2, NP, FAST, CAPK>, jv_search_pqp_kernel<12, NP, FAST, CAPK>, jv_search_pqp_kernel<24, NP, FAST, CAPK> }
#define JV_PQP_CAPS(NP, FAST) { JV_PQP_ROW(NP, FAST, 0), JV_PQP_ROW(NP, FAST, 1), JV_PQP_ROW(NP, FAST, 2), JV_PQP_ROW(NP, FAST, 3) }
// [0 single-pass | 1 multi-pass | 2 single-pass FAST | 3 multi-pass FAST][capacity class][nch slot]
static const pqp_kernel_t g_pqp_kernels[4][4][4] = {JV_PQP_CAPS(1, false), JV_PQP_CAPS(4, false), JV_PQP_CAPS(1, true), JV_PQP_CAPS(4, true)};

// register-LUT variants: PQ-32, FAST, single pass; [capacity class 0..2][nch slot]
#define JV_PQV_ROW(CAPK) \
    { jv_search_pqp_kernel<0, 1, true, CAPK, true>, jv_search_pqp_kernel<2, 1, true, CAPK, true>, jv_search_pqp_kernel<12, 1, true, CAPK, true>, jv_search_pqp_kernel<24, 1, true, CAPK, true> }
static const pqp_kernel_t g_pqv_kernels[3][4] = {JV_PQV_ROW(0), JV_PQV_ROW(1), JV_PQV_ROW(2)};

static int pqp_nch_slot(const JvIndexDev* ix) {
    if (ix->nvq_M > 0) return 0;  // the NVQ decoder lives in the "any d" instances only (score_rows)
    if (ix->stride != ix->nch * 64) return 0;
    return ix->nch == 2 ? 1 : ix->nch == 12 ? 2 : ix->nch == 24 ? 3 : 0;
}
static int pqp_capk(int cap) { return cap <= 512 ? 0 : cap <= 1024 ? 1 : cap <= 2048 ? 2 : 3; }
// lutr: look-up table in registers (jvk_pqp_lutr_ok shapes only)
extern "C" int jvk_pqp_lutr_ok(const JvIndexDev* ix, int cap) {
    return ix->pq_M == 32 && ix->sim != 2 && ix->R * ix->pq_lanes <= JV_WAVE && cap <= 2048 ? 1 : 0;
}
static pqp_kernel_t pqp_pick(const JvIndexDev* ix, int cap, int lutr) {
    const int multi = ix->R * ix->pq_lanes > JV_WAVE ? 1 : 0;
    const int fast = (ix->pq_M % 16 == 0 && ix->sim != 2) ? 1 : 0;
    if (lutr && jvk_pqp_lutr_ok(ix, cap)) return g_pqv_kernels[pqp_capk(cap)][pqp_nch_slot(ix)];
    return g_pqp_kernels[fast * 2 + multi][pqp_capk(cap)][pqp_nch_slot(ix)];
}

extern "C" hipError_t jvk_pqp_set_max_lds(int bytes) {
    for (int v = 0; v < 4; v++)
        for (int c = 0; c < 4; c++)
            for (int s = 0; s < 4; s++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_pqp_kernels[v][c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess && v == 0 && c < 3)
                    e = hipFuncSetAttribute((const void*)g_pqv_kernels[c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

extern "C" int jvk_pqp_max_entries(void) { return 4096; }

// resident workgroups per CU for this index shape, pool capacity and LDS size
extern "C" int jvk_pqp_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes, int lutr) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pqp_pick(ix, cap, lutr), JV_WAVE, (size_t)lds_bytes) != hipSuccess) return 1;
    return nb < 1 ? 1 : nb;
}

// blocks = resident workgroups (the host sizes the log scratch to it); a->cand_cap = pool entries
extern "C" hipError_t jvk_launch_search_pqp(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, int lutr, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    pqp_pick(ix, a->cand_cap, lutr)<<<blocks, JV_WAVE, lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}
