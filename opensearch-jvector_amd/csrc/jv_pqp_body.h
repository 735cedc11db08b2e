// jv_pqp_body.h — the persistent pool kernel's template (search_one_pqp + jv_search_pqp_kernel), included by
// jv_kernels_pqp.hip (unfiltered instances, launchers) and jv_kernels_pqpf.hip (instances with a doc filter) so that the
// two sets of instances compile in parallel.
#pragma once
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
// With a doc filter (FILT) one more bit: [31:3] ~node (n < 2^29) | bit 2 "accepted by the filter" | bits 1, 0 as above.
template <bool FILT>
__device__ __forceinline__ int64_t pqp_key(float score, int node, bool accepted) {
    int32_t b = __float_as_int(score);
    int32_t s = b ^ ((b >> 31) & 0x7fffffff);
    if (FILT)
        return (int64_t)(((uint64_t)(uint32_t)s << 32) | ((uint64_t)((uint32_t)(~node) & 0x1FFFFFFFu) << 3) | (accepted ? 4ull : 0ull) | 3ull);
    return (int64_t)(((uint64_t)(uint32_t)s << 32) | ((uint64_t)((uint32_t)(~node) & 0x3FFFFFFFu) << 2) | 3ull);
}
template <bool FILT>
__device__ __forceinline__ int lo_node(int lo) {
    return FILT ? (int)((~((uint32_t)lo >> 3)) & 0x1FFFFFFFu) : (int)((~((uint32_t)lo >> 2)) & 0x3FFFFFFFu);
}
// position of the n-th (1-based) set bit of m; popcount(m) >= n
__device__ __forceinline__ int pqp_select_nth_bit(unsigned long long m, int n) {
    int pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long low = m & ((1ull << w) - 1ull);
        const int c = __popcll(low);
        if (n > c) {
            n -= c;
            m >>= w;
            pos += w;
        } else {
            m = low;
        }
    }
    return pos;
}

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
// CAPK: pool capacity class: 0 -> <= 512 entries, 1 -> <= 1 024, 2 -> <= 2 048, 3 -> <= 4 096, 4 -> <= 8 192, 5 -> <= 16 384 (4, 5: filtered instances only)
// LUTR: the look-up table lives in REGISTERS (PQ-32, FAST, single pass only): lutr[m][e], lane l = lut[m][4 l + e]; a
//       look-up is ds_bpermute (lane = code >> 2) of the four e-registers + a bit-select by code & 3.  Costs ~3x the
//       instructions of an LDS gather, but LDS then only holds the pool: 8 resident queries per CU (two waves per SIMD
//       fill each other's stalls) instead of 3-4.
// FILT: the query has a doc filter (J/JVectorReader.java:157-163).  jvector's result queue only admits accepted nodes but
//       every node is traversed: the pool holds every scored node whose score >= the rerankK-th best ACCEPTED one
//       (~ rerankK / selectivity entries); `bpos` tracks that entry.  Single-pass blocks only (NP = 1).
template <int NCHT, int NP, bool FAST, int CAPK, bool LUTR, bool FILT = false>
__device__ __forceinline__ void search_one_pqp(const JvIndexDev& ix, const JvSearchArgs& a, const int qi, unsigned char* smem, int32_t* explog) {
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
    const uint64_t* const accw = FILT ? a.accept + (size_t)qi * (size_t)a.accept_stride : nullptr;
    auto accepts = [&](int node) -> bool {  // the reference's acceptOrds lambda
        if (a.accept_ord) return (a.accept_ord[node >> 6] >> (node & 63)) & 1ull;  // (batch-wide filter, translated by the host's pre-pass)
        const int doc = ix.ord2doc ? ix.ord2doc[node] : node;
        return doc >= 0 && (int64_t)doc < a.accept_docs && ((accw[doc >> 6] >> (doc & 63)) & 1ull);
    };
    if (FILT) {
        // Rung choice only (never results): estimate the filter's selectivity from 256 sampled words (16 384 bits: +-0.003 at
        // selectivity 0.2 — the rungs are 20-30 % apart); a pool of ~ rerankK / selectivity entries that cannot fit this
        // launch's capacity is handed on right away instead of after a wasted search.
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
        // (first launch: pools beyond 8 rerankK are not worth predicting; later rungs: skip whenever a wider one follows —
        //  retry_only = 2 marks the last on-chip rung, which takes whatever reaches it)
        if (need > (float)a.cand_cap && (a.retry_only == 1 || (a.retry_only == 0 && a.cand_cap < 8 * rk + 256))) {
            if (lane == 0) {
                a.out_flags[qi] = (int32_t)(JV_FLAG_OVERFLOW | (9u << 8));  // (9: estimated pool beyond this launch's — a property of (filter, rerankK))
                a.out_count[qi] = 0;
            }
            for (int i = lane; i < topK; i += JV_WAVE) {
                o_nodes[i] = -1;
                if (o_docs) o_docs[i] = -1;
                o_scores[i] = 0.0f;
            }
            return;
        }
    }

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
    int bpos = -1, nacc = 0;  // FILT: position of the rk-th best accepted entry (-1: fewer than rk so far), accepted entries in the pool
    float bscore = 0.0f;  // score of the rk-th best entry once np >= rk
    {
        const int ep = ix.entry;
        u32x4 cw = (u32x4){0, 0, 0, 0};
        if (lane < lpn && my_chunk) cw = *(const u32x4*)(ix.pq_codes + (size_t)ep * cs + my_c * 16);
        float s = adc_score(cw, lane < lpn && my_chunk);
        s = __shfl(s, 0, JV_WAVE);
        const bool acc_ep = FILT ? accepts(ep) : true;
        if (lane == 0) pool[0] = pqp_key<FILT>(s, ep, acc_ep);
        np = 1;
        if (FILT) {
            nacc = acc_ep ? 1 : 0;
            if (nacc >= rk) bpos = 0, bscore = s;
        } else if (rk <= 1) {
            bscore = s;
        }
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
                c2 = lo_node<FILT>(__builtin_amdgcn_readlane(e1lo, __ffsll((long long)m2) - 1));
            } else if (((t1 + 1) << 6) < np) {
                const int64_t e2 = pool[min(((t1 + 1) << 6) + lane, cap)];
                const unsigned long long m3 = __ballot((e2 & 1ll) != 0);
                if (m3) c2 = lo_node<FILT>(__builtin_amdgcn_readlane((int)(uint32_t)(e2 & 0xFFFFFFFFll), __ffsll((long long)m3) - 1));
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
        if (FILT) {
            // the same rule over ACCEPTED entries: with t = accepted entries of the boundary's equal-score run at or ahead of
            // bpos, rk - t accepted entries score higher; the candidate is rejected when those + the accepted, expanded
            // entries of the run (minus the rejected ones) already fill rerankK.  A candidate the filter does not accept
            // is expanded and never admitted.
            if ((pk_lo & 4) && bpos >= 0 && sc == bscore) {
                int tt = bpos >> 6;
                while (tt > 0 && (int)(pool[tt << 6] >> 32) == pk_hi) tt--;
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
        } else if (expanded >= rk && idx < rk + nrej) {
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
        const int c = lo_node<FILT>(pk_lo);
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
                    const int32_t* const arow = ix.adj + (size_t)c * (size_t)R;
                    const unsigned char* const blk = ix.pq_fused + (size_t)c * (size_t)R * (size_t)cs;
                    nnp[ps] = j < R ? arow[j] : -1;
                    if (j < R && my_chunk) cwp[ps] = *(const u32x4*)(blk + (uint32_t)(j * cs + my_c * 16));
                }
            }
        }
        STAMP(0)  // find + pool reads
        // FILT: the neighbours' accept bits are requested BEFORE the runner-up's block — vmcnt retires in order, so a wait for
        // an accept word issued behind the prefetch would drain the prefetch on every expansion
        bool accn = true;
        if (FILT) accn = nnp[0] >= 0 && my_c == 0 ? accepts(nnp[0]) : false;
        if (FILT) __builtin_amdgcn_sched_barrier(0);
        // start the runner-up's fetch now, UNCONDITIONALLY (clamped indices): a fixed number of younger loads lets the
        // wait for this expansion's block leave them in flight
        pf_node = c2;
        {
            const int c2e = c2 >= 0 ? c2 : c;
#pragma unroll
            for (int ps = 0; ps < NP; ps++) {
                if (NP == 1 || ps < npass) {
                    const int j = min(ps * jpp + my_slot, R - 1);
                    // (uniform block base + a 32-bit lane offset: scalar address arithmetic, saddr + voffset loads)
                    const int32_t* const arow = ix.adj + (size_t)c2e * (size_t)R;
                    const unsigned char* const blk = ix.pq_fused + (size_t)c2e * (size_t)R * (size_t)cs;
                    pf_nn[ps] = arow[j];
                    pf_cw[ps] = *(const u32x4*)(blk + (uint32_t)(j * cs + (my_chunk ? my_c * 16 : 0)));
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
        if ((FILT ? bpos >= 0 : np >= rk) && score < bscore) keep = false;  // below the boundary for good
        const int64_t v = pqp_key<FILT>(score, nn, accn);
        // ---- rank of every neighbour's key in the pool, all lanes at once; "same node" = equal up to bit 0 ----
        int rold;
        {
            int lo = 0;
            if (CAPK == 5) lo = rank_level<2048, 8>(pool, lo, cap, v);
            if (CAPK == 5) lo = rank_level<128, 16>(pool, lo, cap, v);
            if (CAPK == 4) lo = rank_level<1024, 8>(pool, lo, cap, v);
            if (CAPK == 3) lo = rank_level<512, 8>(pool, lo, cap, v);
            if (CAPK == 2) lo = rank_level<512, 4>(pool, lo, cap, v);
            if (CAPK == 1) lo = rank_level<256, 4>(pool, lo, cap, v);
            if (CAPK == 1) lo = rank_level<64, 4>(pool, lo, cap, v);
            else if (CAPK == 4) lo = rank_level<64, 16>(pool, lo, cap, v);
            else if (CAPK != 5) lo = rank_level<64, 8>(pool, lo, cap, v);
            if (CAPK == 5) lo = rank_level<8, 16>(pool, lo, cap, v);
            else lo = rank_level<8, 8>(pool, lo, cap, v);
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
                int t = (np - 1) >> 6;
                // (four chunks per round trip while whole chunks move: the reads of a batch are issued together — a wide
                //  filtered pool shifts ~50 chunks per expansion and one dependent LDS round trip per chunk was its cost)
                for (; t - 3 > (r_min >> 6); t -= 4) {
                    const int pos = (t << 6) + lane;
                    const int64_t e0 = pool[min(pos, cap)], e1 = pool[pos - 64], e2 = pool[pos - 128], e3 = pool[pos - 192];
                    if (pos < np) pool[pos + 1] = e0;
                    pool[pos - 63] = e1;
                    pool[pos - 127] = e2;
                    pool[pos - 191] = e3;
                }
                for (; t >= (r_min >> 6); t--) {
                    const int pos = (t << 6) + lane;
                    const int64_t e = pool[min(pos, cap)];
                    if (pos < np && pos >= r_min) pool[pos + 1] = e;
                }
            } else {
                // Bursts of ~6 new keys are the norm when there are several.  Every kept key is read out of its lane into
                // scalar registers in turn and compared by all lanes at once: rank among the new keys, twins (the same
                // neighbour twice in one adjacency row of a malformed graph: keep the first) — no LDS round trips.
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
                }
                const int lane_first = __ffsll((long long)__ballot(keep && rnew == 0)) - 1;     // largest new key
                const int lane_last = __ffsll((long long)__ballot(keep && rnew == nk - 1)) - 1;  // smallest new key
                r_min = __builtin_amdgcn_readlane(rold, lane_first);
                r_max = __builtin_amdgcn_readlane(rold, lane_last);
                STAMP(14)  // (diagnostic) ranks among the new keys
                // in-place shift, from the last occupied chunk down to the chunk of the first insertion point: an old
                // entry at position p moves up by the number of new keys that rank ahead of it
                const int t_mixed = r_max >> 6;  // chunks above it shift uniformly by nk
                // new keys ranked at or before a chunk's first position move the whole chunk; the (few) ranked inside it move
                // the entries at or behind them — two ballots and a short scalar loop, no LDS round trip
                auto moved_by = [&](int tt, int pos) -> int {
                    if (tt > t_mixed) return nk;
                    const int cs0 = tt << 6;
                    int cnt = __popcll(__ballot(keep && rold <= cs0));
                    unsigned long long inm = __ballot(keep && rold > cs0 && rold <= cs0 + 63);
                    while (inm) {
                        const int j = __ffsll((long long)inm) - 1;
                        inm &= inm - 1ull;
                        cnt += pos >= __builtin_amdgcn_readlane(rold, j) ? 1 : 0;
                    }
                    return cnt;
                };
                int t = (np - 1) >> 6;
                // four chunks per LDS round trip (all reads of a batch before its writes; destinations only move up)
                for (; t - 3 >= (r_min >> 6); t -= 4) {
                    const int pos = (t << 6) + lane;
                    const int64_t e0 = pool[min(pos, cap)], e1 = pool[pos - 64], e2 = pool[pos - 128], e3 = pool[pos - 192];
                    const int c0 = moved_by(t, pos), c1 = moved_by(t - 1, pos - 64), c2 = moved_by(t - 2, pos - 128), c3 = moved_by(t - 3, pos - 192);
                    if (pos < np && c0 > 0) pool[pos + c0] = e0;
                    if (c1 > 0) pool[pos - 64 + c1] = e1;
                    if (c2 > 0) pool[pos - 128 + c2] = e2;
                    if (c3 > 0) pool[pos - 192 + c3] = e3;
                }
                for (; t >= (r_min >> 6); t--) {
                    const int pos = (t << 6) + lane;
                    const int64_t e = pool[min(pos, cap)];
                    const int cnt = moved_by(t, pos);
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
            if (FILT) {
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
                    const int bhi = (int)(pool[bpos] >> 32);
                    const float nb = hi_score(bhi);
                    if (nb != bscore) nrej = 0;
                    bscore = nb;
                    // ties behind the boundary stay (slots left: cap - (bpos + 1) - R; none left = this launch's pool is too
                    // small for this filter's selectivity), everything below goes
                    const int slack = cap - (bpos + 1) - R;
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
                        break;
                    }
                    np = bpos + 1 + run;
                    nacc = rk + acc_t;
                    for (int p0 = np; p0 < ntot; p0 += JV_WAVE)
                        if (p0 + lane < ntot) pool[p0 + lane] = KEY_MIN;
                } else if (ntot > cap - R) {
                    why = 3;  // fewer than rk accepted entries among more scored nodes than this launch's pool holds
                    break;
                }
            } else if (ntot >= rk) {
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
    if (why == 0 && (FILT || nrej > 0)) {
        // take the rejected entries out: what remains in front is jvector's result queue (ascending pass, every entry moves
        // down by the number of rejected entries ahead of it; a chunk is read completely before it is written)
        int carry = 0;
        for (int t = 0; (t << 6) < np; t++) {
            const int pos = (t << 6) + lane;
            const int64_t e = pool[min(pos, cap)];
            const bool rej = pos < np && (!(e & 2ll) || (FILT && !(e & 4ll)));  // (FILT: what the filter does not accept goes too)
            const unsigned long long rm = __ballot(rej);
            const int shift = carry + __popcll(rm & ((1ull << lane) - 1ull));
            if (pos < np && !rej && shift > 0) pool[pos - shift] = e;
            carry += __popcll(rm);
        }
        for (int p0 = np - carry; p0 < np; p0 += JV_WAVE)
            if (p0 + lane < np) pool[p0 + lane] = KEY_MIN;
        np -= carry;
    }

    // LUTR: the pool moves to registers so that the whole LDS allocation can serve as the visited-count hash set.  Capacity
    // classes 3 and 4 (filtered instances; up to 8 192 entries do not fit the registers): the pool STAYS — by now it only holds
    // the result queue (~rerankK entries) — and the hash set, then the rerank scratch, take the LDS behind it.
    constexpr bool STAY = LUTR && CAPK >= 3;
    const int free_off = STAY ? ((a.pqp_pool_off + (np + 1) * 8 + 15) & ~15) : 0;
    if (STAY && why == 0) {
        const int rerank_need = ix.nch * 64 * 4 + JV_TODO * 8 + ((rk + 1) & ~1) * 8;
        if (a.pqp_lds_bytes - free_off < (rerank_need > 4096 ? rerank_need : 4096)) why = 4;  // (a boundary tie storm: next rung)
    }
    i32x32 PL, PH;
    if (LUTR && !STAY && why == 0) {
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
        uint32_t* vh = (uint32_t*)(smem + free_off);
        const int hash_bytes = LUTR ? a.pqp_lds_bytes - free_off : lut_bytes;
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
                int cnt = 0, cntl = 0;  // (cntl: fresh entries of the grouped form, per lane)
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
                        // one group's ids into the set; false when the table's fill limit would be passed.  (Round 4, as in
                        // jv_pqw_body.h: fresh entries counted per lane and summed once per group, the probe rounds after the first
                        // only for batches that still have a lane on its way down a chain, and two groups in flight in registers of
                        // their own — the old `q0 = q1` hand-over was a copy that waits for the YOUNGEST load.)
                        auto probe_group = [&](const int (&q)[VB], int e0) -> bool {
                            int nb[VB];
                            uint32_t hh[VB];
                            bool pend[VB];
                            int pl = 0;
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const int e = e0 + u * rows_per + lane / R;
                                nb[u] = (e < nblk && lane < rows_per * R) ? q[u] : -1;
                                if (nb[u] >= 0 && parts > 1 && part_of((uint32_t)nb[u]) != p) nb[u] = -1;
                                pend[u] = nb[u] >= 0;
                                hh[u] = ((uint32_t)nb[u] * 0x9E3779B1u) >> vshift;
                                pl += pend[u] ? 1 : 0;
                            }
                            if (jv_wave_sum_int(cntl + pl) > vlimit) return false;
                            uint32_t live = 0;
                            {
                                uint32_t oldv[VB];
#pragma unroll
                                for (int u = 0; u < VB; u++) oldv[u] = pend[u] ? atomicCAS(&vh[hh[u]], HASH_EMPTY, (uint32_t)nb[u]) : 0u;
#pragma unroll
                                for (int u = 0; u < VB; u++) {
                                    const bool fresh = pend[u] && oldv[u] == HASH_EMPTY;
                                    cntl += fresh ? 1 : 0;
                                    pend[u] = pend[u] && !fresh && oldv[u] != (uint32_t)nb[u];
                                    hh[u] = (hh[u] + 1) & vmask;
                                    if (__any(pend[u])) live |= 1u << u;
                                }
                            }
                            while (live) {
                                uint32_t oldv[VB];
#pragma unroll
                                for (int u = 0; u < VB; u++)
                                    if (live & (1u << u)) oldv[u] = pend[u] ? atomicCAS(&vh[hh[u]], HASH_EMPTY, (uint32_t)nb[u]) : 0u;
#pragma unroll
                                for (int u = 0; u < VB; u++)
                                    if (live & (1u << u)) {
                                        const bool fresh = pend[u] && oldv[u] == HASH_EMPTY;
                                        cntl += fresh ? 1 : 0;
                                        pend[u] = pend[u] && !fresh && oldv[u] != (uint32_t)nb[u];
                                        hh[u] = (hh[u] + 1) & vmask;
                                        if (!__any(pend[u])) live &= ~(1u << u);
                                    }
                            }
                            return true;
                        };
                        int qq[2][VB];
                        load_group(0, qq[0]);
                        load_group(G, qq[1]);
                        for (int e0 = 0; e0 < nblk && !again; e0 += 2 * G) {
#pragma unroll
                            for (int k = 0; k < 2; k++) {
                                const int ek = e0 + k * G;
                                if (ek < nblk && !again) {
                                    if (!probe_group(qq[k], ek)) again = true;
                                    else load_group(ek + 2 * G, qq[k]);
                                }
                            }
                        }
                    }
                    cnt += jv_wave_sum_int(cntl);
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
    // Lucene discards every search whose visited + expanded reached the visit limit (J/JVectorReader.java:202-207 reports the
    // sum, AbstractKnnVectorQuery tests it).  The loop above stops on expansions alone (visited is only known now): a search
    // that ran to its end but whose SUM reaches the limit is flagged the same way, with its real counters.
    int early_vis = -1;
    if (why == 0 && a.visit_limit > 0 && visited + expanded >= a.visit_limit) {
        why = 15;
        early_vis = visited;
    }
    const int nres = np < rk ? np : rk;
    // ---- rerank scratch (where the LUT / the hash set was): query, todo lists, exact keys ----
    float* q_lds = (float*)(smem + free_off);
    size_t roff = (size_t)free_off + (size_t)ix.nch * 64 * sizeof(float);
    float* todo_score = (float*)(smem + roff);
    roff += JV_TODO * sizeof(float);
    int32_t* todo = (int32_t*)(smem + roff);
    roff += JV_TODO * sizeof(int32_t);
    int64_t* fin = (int64_t*)(smem + roff);  // [rk]
    const int64_t* rpool = pool;
    if (LUTR && !STAY && why == 0) {
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
            node = lo_node<FILT>((int)(uint32_t)(k & 0xFFFFFFFFll));
            take = above > 0 ? key_score(k) >= a.rerank_floor : i == 0;  // position 0 is the best approximate entry
        }
        const unsigned long long tm = __ballot(take);
        const int m = __popcll(tm);
        if (take) todo[__popcll(tm & ((1ull << lane) - 1ull))] = node;
        __syncthreads();
        if (m > 0) {
            score_rows<NCHT, JV_PQF_RERANK_UMUL, true>(ix, q_lds, todo, m, todo_score, qnorm2, 1.0f, lane);
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
template <int NCHT, int NP, bool FAST, int CAPK, bool LUTR = false, bool FILT = false>
__global__ __launch_bounds__(JV_WAVE, (LUTR && CAPK < 3) ? 2 : 1) void jv_search_pqp_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* explog = a.pqp_log + (size_t)blockIdx.x * (size_t)a.pqp_log_cap;
    // later launches (a.retry_only: wider pool, longer log): only the queries an earlier one flagged, found up to 8 flags at a
    // time.  (Unfiltered: a handful of 65 536 flags are set and a chunk rarely holds two.  Filtered: a whole batch can land on
    // one rung with tens of milliseconds per query — chunks of 8 then left a third of the grid idle behind the last round, so
    // the chunk shrinks with the number of flags per workgroup.)
    const int chunk = max(1, min(8, a.nq / ((int)gridDim.x * 8)));
    int base = 0;
    unsigned long long todo = 0ull;
    for (;;) {
        int qi = 0;
        if (a.retry_only) {
            while (!todo) {
                if (threadIdx.x == 0) base = atomicAdd(a.retry_counter, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= a.nq) return;
                const int qf = ((int)threadIdx.x < chunk && base + (int)threadIdx.x < a.nq) ? a.out_flags[base + threadIdx.x] : 0;
                todo = __ballot(((uint32_t)qf & JV_FLAG_OVERFLOW) != 0);
            }
            qi = base + __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
        } else {
            if (threadIdx.x == 0) qi = atomicAdd(a.pqp_counter, 1);
            qi = __builtin_amdgcn_readfirstlane(qi);
            if (qi >= a.nq) break;
        }
        search_one_pqp<NCHT, NP, FAST, CAPK, LUTR, FILT>(ix, a, qi, smem, explog);
        __syncthreads();
    }
}

