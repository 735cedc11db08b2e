// jv_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4) for the jVector GraphSearcher hot path.
//
// One 64-lane wavefront (= one workgroup) runs one query's whole beam search:
//   * the query vector, the PQ look-up table, both NodeQueues and the visited set live in LDS;
//   * an expansion reads one adjacency row (coalesced 4R bytes), filters it through the LDS visited
//     set, and scores the survivors with a quarter-wave (16 lanes x 16 B = one 256-B burst) per
//     vector row, four rows per wave-instruction, many rows in flight;
//   * all queue operations are wave-parallel scans over small unsorted LDS arrays (a pop is an
//     arg-max), which yields exactly the order jvector's binary heaps define because NodeQueue keys
//     are unique and totally ordered.
// Thousands of such waves are resident at once; HBM bandwidth comes from the aggregate of their
// independent row gathers.  No MFMA: this path is gather/scan work (<= 0.5 flop per byte).
//
// Semantics follow jvector 4.0.0-rc.9 GraphSearcher as called from
// J/JVectorReader.java:165-173; the CPU oracle (oracle/jv_oracle.c) is the
// checker and uses the identical canonical fp32 order, so scores are bit-equal.
//
// Compile with -ffp-contract=off: every fused multiply-add is an explicit fmaf().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jv_device.h"

#include "jv_dev_common.h"

// POOL: level 0 runs on one sorted pool (exact when there is no filter and threshold <= 0, ties included:
// DESIGN.md "Single-pool search"); otherwise the two-queue form of jvector is executed literally.
// BIG: visited set = bitset in HBM scratch (exact whatever the search touches).  QLDS (with BIG): both queues still live
// in LDS (as many candidate slots as the workgroup's LDS holds) — a pop is an arg-max scan over the queue, so on-chip
// queues are ~20x faster than the HBM ones; only a query that outgrows them as well takes the HBM-queue form.
// LUTG: the PQ look-up table lives in this workgroup's HBM scratch instead of LDS (pq_M * 1 KB beyond the LDS: the reference's
// default 192 subspaces for d >= 768, J/JVectorIndexQuantization.java:428-446) — same arithmetic, gathers served by L1 / L2
template <bool PQ, bool BIG, bool POOL, int NCHT, bool QLDS = false, bool LUTG = false>
__device__ __forceinline__ void search_one(const JvIndexDev& ix, const JvSearchArgs& a, int qi, unsigned char* smem,
                           int64_t* big_cand, uint32_t* big_bits) {
    const int lane = threadIdx.x;
    const int rk = a.rk, topK = a.topK;

    // ---- LDS carve (all offsets multiples of 16 B) ----
    float* q_lds = (float*)smem;
    size_t off = (size_t)ix.nch * 64 * sizeof(float);
    float* todo_score = (float*)(smem + off);
    off += JV_TODO * sizeof(float);
    int32_t* todo = (int32_t*)(smem + off);
    off += JV_TODO * sizeof(int32_t);
    int64_t* newk = (int64_t*)(smem + off);
    off += JV_TODO * sizeof(int64_t);
    Tracker trk;
    trk.recent = trk.best = trk.best2 = nullptr;
    trk.idx = trk.obs = trk.nbest = 0;
    if (!POOL && a.threshold > 0.0f) {  // threshold queries: tracker state (host adds JV_TRACKER_LDS bytes)
        trk.recent = (float*)(smem + off);
        off += 512 * sizeof(float);
        trk.best = (float*)(smem + off);
        off += (TRK_BEST + JV_WAVE + 4) * sizeof(float);
        trk.best2 = (float*)(smem + off);
        off += (TRK_BEST + JV_WAVE + 4) * sizeof(float);
    }
    float* lut = nullptr;
    float* qc_lds = nullptr;  // PQ: the centred query q' = q - globalCentroid
    if (PQ) {
        if (LUTG) {
            lut = a.lut_scratch + (size_t)blockIdx.x * (size_t)ix.pq_M * 256;
        } else {
            lut = (float*)(smem + off);
            off += (size_t)ix.pq_M * 256 * sizeof(float);
        }
        qc_lds = (float*)(smem + off);
        off += (size_t)ix.nch * 64 * sizeof(float);
    }
    int64_t* res;
    int64_t* cand;
    uint32_t* hash = nullptr;
    int cand_cap;
    if (BIG && !QLDS) {
        res = big_cand;
        cand = big_cand + a.res_cap;
        cand_cap = a.big_cand_cap - a.res_cap;
    } else {
        res = (int64_t*)(smem + off);
        off += (size_t)a.res_cap * sizeof(int64_t);
        cand = (int64_t*)(smem + off);
        off += (size_t)a.cand_cap * sizeof(int64_t);
        hash = (uint32_t*)(smem + off);
        cand_cap = a.cand_cap;
    }
    // (vs.lds is set after `hash` is known; see below)
    const uint32_t hmask = (uint32_t)a.hash_slots - 1u;
    const int hshift = 32 - (31 - __clz(a.hash_slots));
    const int hash_limit = (a.hash_slots / 4) * 3;
    Visited vs;
    vs.lds = nullptr;
    vs.lmask = hmask;
    vs.lshift = hshift;
    vs.spill = nullptr;
    vs.smask = a.spill_slots > 0 ? (uint32_t)a.spill_slots - 1u : 0u;
    vs.sshift = a.spill_slots > 0 ? 32 - (31 - __clz(a.spill_slots)) : 0;
    vs.nspill = 0;
    const int spill_limit = (a.spill_slots / 4) * 3;

    // ---- stage the query; clear the visited set ----
    const float* qg = a.queries + (size_t)qi * ix.d;
    for (int i = lane; i < ix.nch * 64; i += JV_WAVE) q_lds[i] = i < ix.d ? qg[i] : 0.0f;
    if (BIG) {
        const int words = (ix.n + 31) >> 5;
        for (int i = lane; i < words; i += JV_WAVE) big_bits[i] = 0u;
    } else {
        for (int i = lane; i < a.hash_slots; i += JV_WAVE) hash[i] = HASH_EMPTY;
    }
    __syncthreads();
    float qnorm2 = 0.0f;
    if (ix.sim == 2) qnorm2 = query_norm2(ix, q_lds, lane), qnorm2 = __shfl(qnorm2, 0, 64);
    if (PQ) {
        for (int i = lane; i < ix.nch * 64; i += JV_WAVE) {
            float v = q_lds[i];
            if (ix.pq_centroid && i < ix.d) v = v - ix.pq_centroid[i];
            qc_lds[i] = v;
        }
        __syncthreads();
        build_lut<16>(ix, qc_lds, lut, lane);
        if (LUTG) {
            // the table was written through to L2; the L1 may still hold lines of the PREVIOUS query's table at these addresses
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // exact-provider path carries the Lucene MIP x2 wrap (J/JVectorReader.java:220-239,359-364);
    // the PQ provider's reranker is not wrapped (:353-356)
    const float search_scale = PQ ? 1.0f : ix.score_scale;

    int st_nvisited_lds = 0;  // nodes in the LDS table
    vs.lds = hash;
    // room for 64 more nodes?  LDS table -> freeze it and take a spill table -> give up (retry path)
    auto visited_room = [&]() -> bool {
        if (vs.spill == nullptr) {
            if (st_nvisited_lds + JV_WAVE <= hash_limit) return true;
            if (a.spill == nullptr || a.spill_slots <= 0) return false;
            int t = 0;
            if (lane == 0) t = atomicAdd(a.spill_counter, 1);
            t = __shfl(t, 0, JV_WAVE);
            if (t >= a.spill_tables) return false;
            uint32_t* tab = a.spill + (size_t)t * a.spill_slots;
            for (int i = lane; i < a.spill_slots; i += JV_WAVE) tab[i] = HASH_EMPTY;
            __syncthreads();
            vs.spill = tab;
            return true;
        }
        return vs.nspill + JV_WAVE <= spill_limit;
    };

    bool early = false;  // visit_limit reached
    QState st;
    st.ncand = st.nhand = st.nres = 0;
    st.worst = KEY_MAX;
    st.worst_idx = -1;
    st.visited = st.expanded = st.expanded_base = st.reranked = 0;
    st.overflow = false;

    auto score_todo = [&](int m) {
        if (PQ) score_nodes_pq(ix, lut, todo, m, todo_score, qnorm2, lane);
        else score_rows<NCHT>(ix, q_lds, todo, m, todo_score, qnorm2, search_scale, lane);
        __syncthreads();
    };

    // ---- initializeInternal: score the entry point, mark visited (not counted), push ----
    {
        const int ep = ix.entry;
        if (lane == 0) {
            todo[0] = ep;
            if (BIG) visited_insert_bits(big_bits, (uint32_t)ep);
            else visited_insert_lds(hash, hmask, hshift, (uint32_t)ep);
        }
        st_nvisited_lds = 1;
        __syncthreads();
        score_todo(1);
        if (lane == 0) cand[0] = make_key(todo_score[0], ep);
        st.ncand = 1;
        __syncthreads();
    }

    // ---- searchOneLayer for lvl = top .. 0 ----
    for (int lvl = ix.num_upper; lvl >= (POOL ? 1 : 0) && !st.overflow && !early; lvl--) {
        const int rk_cur = lvl > 0 ? 1 : rk;
        const float thr = lvl > 0 ? 0.0f : a.threshold;
        const bool accept_all = lvl > 0 || a.accept == nullptr;
        const uint64_t* const accw = a.accept ? a.accept + (size_t)qi * (size_t)a.accept_stride : nullptr;
        while (st.ncand > 0) {
            int64_t best;
            int bi;
            scan_max(cand, st.ncand, lane, best, bi);
            const float sc = key_score(best);
            if (st.nres >= rk_cur && sc < key_score(st.worst)) break;
            if (a.visit_limit > 0 && st.visited + st.expanded >= a.visit_limit) {  // Lucene discards this search: stop now
                early = true;
                break;
            }
            // when querying by threshold, also stop when more qualifying results are improbable
            if (thr > 0.0f && tracker_should_stop(trk, thr, lane)) break;
            // pop
            const int c = key_node(best);
            if (lane == 0) cand[bi] = cand[st.ncand - 1];
            st.ncand--;
            // adjacency row (issued early: it is the first dependent HBM access of the expansion)
            const int32_t* row;
            int deg;
            if (lvl == 0) {
                row = ix.adj + (size_t)c * ix.R;
                deg = ix.R;
            } else {
                const JvLayerDev& L = ix.upper[lvl - 1];
                int lo = 0, hi = L.count - 1, pos = -1;
                while (lo <= hi) {
                    int mid = (lo + hi) >> 1;
                    int v = L.nodes[mid];
                    if (v == c) {
                        pos = mid;
                        break;
                    }
                    if (v < c) lo = mid + 1;
                    else hi = mid - 1;
                }
                row = pos >= 0 ? L.adj + (size_t)pos * L.degree : nullptr;
                deg = pos >= 0 ? L.degree : 0;
            }
            int nb0 = (lane < deg) ? row[lane] : -1;
            // accept test (J/JVectorReader.java:157-163)
            bool acc = true;
            if (!accept_all) {
                int doc = ix.ord2doc ? ix.ord2doc[c] : c;
                acc = doc >= 0 && (int64_t)doc < a.accept_docs && ((accw[doc >> 6] >> (doc & 63)) & 1ull);
            }
            // addTopCandidate: a full queue only admits a STRICTLY better score
            if (acc && sc >= thr) {
                if (BIG && !QLDS && PQ && lvl == 0) {  // call log for replay_first_best (top of the candidate area, like the hand-back list)
                    if (st.ncand + st.nhand + 1 > cand_cap) {
                        st.overflow = true;
                        break;
                    }
                    if (lane == 0) cand[cand_cap - 1 - st.nhand] = best;
                    st.nhand++;
                }
                if (st.nres < rk_cur) {
                    if (lane == 0) res[st.nres] = best;
                    st.nres++;
                    __syncthreads();
                    if (st.nres == rk_cur) scan_min(res, st.nres, lane, st.worst, st.worst_idx);
                } else if (sc > key_score(st.worst)) {
                    if (lvl > 0) {  // evicted -> handed back to the next layer
                        if (lane == 0) cand[cand_cap - 1 - st.nhand] = st.worst;
                        st.nhand++;
                    }
                    if (lane == 0) res[st.worst_idx] = best;
                    __syncthreads();
                    scan_min(res, st.nres, lane, st.worst, st.worst_idx);
                } else if (lvl > 0) {
                    if (lane == 0) cand[cand_cap - 1 - st.nhand] = best;
                    st.nhand++;
                }
            }
            // neighbours, 64 at a time, in stored order
            for (int cb = 0; cb < deg; cb += JV_WAVE) {
                const int nn = cb == 0 ? nb0 : ((cb + lane < deg) ? row[cb + lane] : -1);
                if (!BIG && !visited_room()) {
                    st.overflow = true;
                    break;
                }
                bool is_new = false;
                if (nn >= 0) {
                    is_new = BIG ? visited_insert_bits(big_bits, (uint32_t)nn)
                                 : visited_insert2(vs, (uint32_t)nn);
                }
                const unsigned long long mask = __ballot(is_new);
                const int m = __popcll(mask);
                if (is_new) todo[__popcll(mask & ((1ull << lane) - 1ull))] = nn;
                __syncthreads();
                if (m == 0) continue;
                if (vs.spill == nullptr) st_nvisited_lds += m;
                else vs.nspill += m;
                st.visited += m;
                score_todo(m);
                if (thr > 0.0f) tracker_track(trk, todo_score, m, lane);
                // push (level 0: a candidate already below a full result queue's worst can never be
                // popped before the loop breaks, so it is not stored)
                bool keep = lane < m;
                float s = keep ? todo_score[lane] : 0.0f;
                if (keep && lvl == 0 && st.nres >= rk_cur && s < key_score(st.worst)) keep = false;
                const unsigned long long km = __ballot(keep);
                const int nk = __popcll(km);
                if (st.ncand + nk + st.nhand > cand_cap) {
                    // compact: drop candidates that can no longer be popped, then re-check
                    if (lvl == 0 && st.nres >= rk_cur) {
                        const float ws = key_score(st.worst);
                        int w = 0;
                        for (int b0 = 0; b0 < st.ncand; b0 += JV_WAVE) {
                            const int i = b0 + lane;
                            int64_t k = i < st.ncand ? cand[i] : KEY_MIN;
                            const bool kp = i < st.ncand && !(key_score(k) < ws);
                            const unsigned long long pm = __ballot(kp);
                            __syncthreads();
                            if (kp) cand[w + __popcll(pm & ((1ull << lane) - 1ull))] = k;
                            w += __popcll(pm);
                            __syncthreads();
                        }
                        st.ncand = w;
                    }
                    if (st.ncand + nk + st.nhand > cand_cap) {
                        st.overflow = true;
                        break;
                    }
                }
                if (keep) cand[st.ncand + __popcll(km & ((1ull << lane) - 1ull))] = make_key(s, todo[lane]);
                st.ncand += nk;
                __syncthreads();
            }
            if (st.overflow) break;
            st.expanded++;
            if (lvl == 0) st.expanded_base++;
        }
        if (lvl > 0 && !st.overflow) {
            // setEntryPointsFromPreviousLayer: results + evicted go back onto the candidate queue
            const int total = st.nres + st.nhand;
            if (st.ncand + total > cand_cap - st.nhand) {
                st.overflow = true;
                break;
            }
            __syncthreads();
            for (int i = lane; i < st.nhand; i += JV_WAVE) cand[st.ncand + i] = cand[cand_cap - 1 - i];
            for (int i = lane; i < st.nres; i += JV_WAVE) cand[st.ncand + st.nhand + i] = res[i];
            st.ncand += total;
            st.nhand = 0;
            st.nres = 0;
            st.worst = KEY_MAX;
            __syncthreads();
        }
    }

    if (POOL && !st.overflow && !early) {
        // ---- level 0 on a single sorted pool ----
        // The pool holds every scored node whose score >= the rk-th best score seen so far (so ties at the
        // boundary stay), in NodeQueue order, with an "unexpanded" bit.  jvector pops the best unexpanded
        // candidate and stops when it is worse than the rk-th best expanded one: with no filter and every
        // score >= threshold that is exactly "expand the first unexpanded pool entry until none is left".
        const int pool_limit = cand_cap - JV_WAVE;  // room for one chunk of 64 new keys beyond the trimmed pool
        int64_t* cur = res;
        int64_t* nxt = cand;
        int np = st.ncand;
        if (np > pool_limit) st.overflow = true;
        if (!st.overflow) {
            for (int i = lane; i < np; i += JV_WAVE) {
                const int64_t v = cand[i];
                int r = 0;
                for (int j = 0; j < np; j++) r += cand[j] > v ? 1 : 0;
                cur[r] = key_to_pool(v);
            }
            __syncthreads();
            np = pool_trim(cur, np, rk, lane);
        }
        // Rank-merge the `keep` lanes' (score, node) into the sorted pool, in place: new keys find their rank
        // with an 8-ary search, old entries at or below the first insertion point shift right chunk by chunk
        // from the top.  `lo_un` = index before which every entry is known to be expanded.
        int lo_un = 0;
        int64_t* newk_s = (int64_t*)todo_score;  // 64 keys: todo_score + todo are dead during a merge
        auto merge_keys = [&](bool keep, float s, int node) -> bool {
            if (keep && np >= rk && s < key_score(cur[rk - 1])) keep = false;  // below the boundary for good
            const unsigned long long km = __ballot(keep);
            const int nk = __popcll(km);
            if (nk == 0) return true;
            if (keep) newk[__popcll(km & ((1ull << lane) - 1ull))] = make_pool_key(s, node);
            __syncthreads();
            int64_t v = 0;
            int rnew = 0, rold = 0;
            if (lane < nk) {
                v = newk[lane];
                for (int j = 0; j < nk; j++) rnew += newk[j] > v ? 1 : 0;
                int lo = 0, hi = np;  // rold = #{pool entries > v}
                while (hi - lo > 8) {
                    const int step = (hi - lo + 7) >> 3;
                    int cgt = 0;
#pragma unroll
                    for (int k2 = 1; k2 < 8; k2++) {
                        const int pp = lo + k2 * step;
                        cgt += (pp < hi && cur[pp] > v) ? 1 : 0;
                    }
                    lo += cgt * step;
                    hi = lo + step < hi ? lo + step : hi;
                }
                rold = lo;
                for (int pp = lo; pp < hi; pp++) rold += cur[pp] > v ? 1 : 0;
                newk_s[rnew] = v;  // new keys in descending order
            }
            const unsigned long long firstm = __ballot(lane < nk && rnew == 0);
            const int r_min = __shfl(rold, __ffsll((long long)firstm) - 1, JV_WAVE);  // first insertion point
            __syncthreads();
            for (int t = (np - 1) >> 6; t >= (r_min >> 6); t--) {
                const int i = (t << 6) + lane;
                const bool mv = i < np && i >= r_min;
                const int64_t ov = mv ? cur[i] : 0;
                int cnt = 0;
                for (int j = 0; j < nk; j++) cnt += (mv && newk_s[j] > ov) ? 1 : 0;
                if (mv && cnt > 0) cur[i + cnt] = ov;  // every lane has read its entry of this chunk already
            }
            if (lane < nk) cur[rold + rnew] = v;
            __syncthreads();
            np = pool_trim(cur, np + nk, rk, lane);
            lo_un = lo_un < r_min ? lo_un : r_min;
            return np <= pool_limit;  // false: more boundary ties than the pool has room for
        };

        // Fused ADC layout: node u's block holds its R neighbours' PQ codes in adjacency order, so one
        // expansion is ONE dependent fetch (ids + codes) instead of two; scores are bit-identical.
        const int lpn = PQ ? ix.pq_lanes : 1;
        const bool fused = PQ && ix.pq_fused != nullptr && ix.R * lpn <= JV_WAVE;
        const int my_j = fused ? lane / lpn : lane;        // stored-order neighbour slot this lane serves
        const int my_c = fused ? lane & (lpn - 1) : 0;     // 16-subspace chunk this lane sums
        const bool my_chunk = PQ && my_c * 16 < ix.pq_M;
        // speculative prefetch of the runner-up's block: it is the next expansion unless a neighbour scored
        // in this one beats it; a wrong guess only costs the (tiny) load.  The prefetch is issued AFTER this
        // expansion's block has been consumed, so the wait for the current block never covers it.
        int pf_node = -1, pf_nn = -1;
        u32x4 pf_cw = (u32x4){0, 0, 0, 0};
        STAMP_DECL
        STAMP(7)  // everything before the loop: staging, LUT build, entry point
        while (!st.overflow) {
            int idx = -1, idx2 = -1;
            for (int b0 = lo_un; b0 < np && idx2 < 0; b0 += JV_WAVE) {
                const int i = b0 + lane;
                const bool un = i < np && (cur[i] & 1ll);
                unsigned long long um = __ballot(un);
                while (um && idx2 < 0) {
                    const int pp = b0 + __ffsll((long long)um) - 1;
                    um &= um - 1ull;
                    if (idx < 0) idx = pp;
                    else idx2 = pp;
                }
            }
            if (idx < 0) break;
            if (a.visit_limit > 0 && st.visited + st.expanded >= a.visit_limit) {
                early = true;
                break;
            }
            lo_un = idx + 1;
            STAMP(0)  // find the best / runner-up unexpanded entries
            const int64_t pk = cur[idx];
            const float sc = key_score(pk);
            if (sc < a.threshold) {  // a node the two-queue form would expand but not collect: take the general path
                st.overflow = true;
                break;
            }
            // jvector admits a popped candidate into a FULL result queue only if it is strictly better than the
            // worst result; the pool's key order would instead rank it ahead of equal-score, higher-ordinal nodes
            // that were expanded earlier.  The two disagree exactly when the candidate sits inside the top rerankK
            // and the expanded entries scoring >= it (all idx entries ahead + the equal-score expanded ones behind)
            // already fill the result queue, i.e. the worst result's score equals the candidate's.  That tie is
            // left to the literal two-queue form.
            if (st.expanded >= rk && idx < rk) {
                int ge = idx;
                for (int j = idx + 1; j < np && ge < rk; j++) {
                    const int64_t kj = cur[j];
                    if (key_score(kj) != sc) break;
                    ge += (kj & 1ll) ? 0 : 1;
                }
                const bool tie_bail = ge >= rk;
                if (tie_bail) {
                    st.overflow = true;
                    break;
                }
            }
            const int c = pool_node(pk);
            const int deg = ix.R;
            const int32_t* row = ix.adj + (size_t)c * ix.R;
            int nn;
            u32x4 cw = (u32x4){0, 0, 0, 0};
            if (c == pf_node) {
                nn = pf_nn;
                cw = pf_cw;
                STAMP_COUNT(6, 1)  // prefetch hits
            } else {
                nn = my_j < deg ? row[my_j] : -1;
                if (fused && my_j < deg && my_chunk)
                    cw = *(const u32x4*)(ix.pq_fused + ((size_t)c * ix.R + my_j) * ix.pq_code_stride + my_c * 16);
            }
            const int c2 = idx2 >= 0 ? pool_node(cur[idx2]) : -1;
            if (lane == 0) cur[idx] = pk & ~1ll;
#ifdef JV_STAMPS
            asm volatile("" ::"v"(nn), "v"(cw[0]), "v"(cw[3]));  // force the block to have arrived
#endif
            STAMP(1)  // wait for this expansion's block
            if (fused) {
                if (!visited_room()) {
                    st.overflow = true;
                    break;
                }
                // ADC for every stored neighbour at once (lanes of a visited neighbour just idle later)
                const bool have = nn >= 0 && my_chunk;
                float s = adc_chunk(lut, cw, my_c * 16, ix.pq_M);
                float na = 0.0f;
                if (ix.sim == 2) na = adc_chunk(ix.pq_norm_lut, cw, my_c * 16, ix.pq_M);
                // the current block is in registers: now start the runner-up's fetch
                __builtin_amdgcn_sched_barrier(0);
                pf_node = c2;
                if (c2 >= 0) {
                    pf_nn = my_j < deg ? ix.adj[(size_t)c2 * ix.R + my_j] : -1;
                    if (my_j < deg && my_chunk)
                        pf_cw = *(const u32x4*)(ix.pq_fused + ((size_t)c2 * ix.R + my_j) * ix.pq_code_stride + my_c * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                s = lanes_tree_sum(have ? s : 0.0f, lpn);
                float score;
                if (ix.sim == 2) {
                    na = lanes_tree_sum(have ? na : 0.0f, lpn);
                    score = map_score(2, s / sqrtf(qnorm2 * na));
                } else {
                    score = map_score(ix.sim, s);
                }
                STAMP(2)  // ADC: LUT gathers + lane tree + score map
                bool is_new = false;
                if (nn >= 0 && my_c == 0) is_new = visited_insert2(vs, (uint32_t)nn);
                const int m = __popcll(__ballot(is_new));
                if (vs.spill == nullptr) st_nvisited_lds += m;
                else vs.nspill += m;
                st.visited += m;
                __syncthreads();
                STAMP(3)  // visited-set inserts
                if (!merge_keys(is_new, score, nn)) {
                    st.overflow = true;
                    break;
                }
                STAMP(4)  // rank merge into the pool
            } else {
                for (int cb = 0; cb < deg; cb += JV_WAVE) {
                    const int nv = cb == 0 ? nn : ((cb + lane < deg) ? row[cb + lane] : -1);
                    if (!visited_room()) {
                        st.overflow = true;
                        break;
                    }
                    bool is_new = false;
                    if (nv >= 0) is_new = visited_insert2(vs, (uint32_t)nv);
                    const unsigned long long mask = __ballot(is_new);
                    const int m = __popcll(mask);
                    if (is_new) todo[__popcll(mask & ((1ull << lane) - 1ull))] = nv;
                    if (cb == 0) {  // adjacency consumed: start the runner-up's fetch under the scoring below
                        __builtin_amdgcn_sched_barrier(0);
                        pf_node = c2;
                        if (c2 >= 0) pf_nn = my_j < deg ? ix.adj[(size_t)c2 * ix.R + my_j] : -1;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __syncthreads();
                    if (m == 0) continue;
                    if (vs.spill == nullptr) st_nvisited_lds += m;
                else vs.nspill += m;
                    st.visited += m;
                    score_todo(m);
                    const bool keep = lane < m;
                    const float ks = keep ? todo_score[lane] : 0.0f;
                    const int kn = keep ? todo[lane] : 0;
                    __syncthreads();
                    if (!merge_keys(keep, ks, kn)) {
                        st.overflow = true;
                        break;
                    }
                }
            }
            if (st.overflow) break;
            st.expanded++;
            st.expanded_base++;
        }
        STAMP(5)  // non-fused expansions / loop exit
        STAMP_FLUSH
        if (!st.overflow && !early) {
            // approximateResults = the best rk expanded nodes = the first rk pool entries
            st.nres = np < rk ? np : rk;
            for (int i = lane; i < st.nres; i += JV_WAVE) nxt[i] = pool_to_key(cur[i]);
            __syncthreads();
            res = nxt;
            cand = cur;
        }
    }

    // a search that ended by itself with visited + expanded at or beyond Lucene's visit limit is discarded like the stopped ones
    if (!st.overflow && !early && a.visit_limit > 0 && st.visited + st.expanded >= a.visit_limit) early = true;
    // NodeQueue.rerank bookkeeping (PQ): how many results reach rerankFloor; if none, the single entry to rescore
    int above = 0, only_node = -1;
    if (PQ && !st.overflow && !early) {
        __syncthreads();
        for (int i = lane; i < st.nres; i += JV_WAVE) above += key_score(res[i]) >= a.rerank_floor ? 1 : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) above += __shfl_xor(above, o, 64);
        if (above == 0 && st.nres > 0) {
            int64_t bk;
            int bidx;
            scan_max(res, st.nres, lane, bk, bidx);
            only_node = key_node(bk);
            int ties = 0;
            for (int b0 = 0; b0 < st.nres; b0 += JV_WAVE) {
                const int i = b0 + lane;
                ties += __popcll(__ballot(i < st.nres && key_score(res[i]) == key_score(bk)));
            }
            if (ties > 1) {
                // several results share the best approximate score: jvector's pick is the first in its heap array
                if (!BIG || QLDS) {
                    st.overflow = true;  // -> the HBM-queue rung replays the heap
                } else if (st.nhand + rk + 2 > cand_cap) {
                    st.overflow = true;
                } else {
                    int nd = -1;
                    if (lane == 0) nd = replay_first_best(cand + (cand_cap - 1), st.nhand, rk, cand);
                    only_node = __shfl(nd, 0, JV_WAVE);
                    __syncthreads();
                }
            }
        }
    }

    int32_t* o_nodes = a.out_nodes + (size_t)qi * topK;
    int32_t* o_docs = a.out_docs ? a.out_docs + (size_t)qi * topK : nullptr;
    float* o_scores = a.out_scores + (size_t)qi * topK;
    if (st.overflow || early) {
        if (lane == 0) {
            if (early && !st.overflow) {
                a.out_flags[qi] = (int32_t)(JV_FLAG_EARLY | (BIG ? JV_FLAG_BIG : 0u));
                int32_t* s = a.out_stats + (size_t)qi * 4;
                s[0] = st.visited;
                s[1] = 0;
                s[2] = st.expanded;
                s[3] = st.expanded_base;
            } else {
                a.out_flags[qi] = (BIG && !QLDS) ? (int32_t)(JV_FLAG_FAILED | JV_FLAG_BIG) : (int32_t)JV_FLAG_OVERFLOW;
            }
            a.out_count[qi] = 0;
        }
        for (int i = lane; i < topK; i += JV_WAVE) {
            o_nodes[i] = -1;
            if (o_docs) o_docs[i] = -1;
            o_scores[i] = 0.0f;
        }
        return;
    }

    // ---- result assembly ----
    int64_t* fin = res;
    int nfin = st.nres;
    if (PQ) {
        // NodeQueue.rerank: exact-rescore entries with approx >= rerankFloor, or only the best one
        fin = cand;  // the candidate queue is dead now; cand_cap >= rk
        nfin = 0;
        for (int b0 = 0; b0 < st.nres; b0 += JV_WAVE) {
            const int i = b0 + lane;
            bool take = false;
            int node = 0;
            if (i < st.nres) {
                const int64_t k = res[i];
                node = key_node(k);
                take = above > 0 ? key_score(k) >= a.rerank_floor : node == only_node;
            }
            const unsigned long long tm = __ballot(take);
            const int m = __popcll(tm);
            if (take) todo[__popcll(tm & ((1ull << lane) - 1ull))] = node;
            __syncthreads();
            if (m > 0) {
                score_rows<NCHT>(ix, q_lds, todo, m, todo_score, qnorm2, 1.0f, lane);
                __syncthreads();
                if (lane < m) fin[nfin + lane] = make_key(todo_score[lane], todo[lane]);
                nfin += m;
                st.reranked += m;
            }
            __syncthreads();
        }
    }
    // top-K by key, descending (score desc, ordinal asc)
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
        int32_t* s = a.out_stats + (size_t)qi * 4;
        s[0] = st.visited;
        s[1] = st.reranked;
        s[2] = st.expanded;
        s[3] = st.expanded_base;
        a.out_flags[qi] = BIG ? (int32_t)JV_FLAG_BIG : 0;
    }
}


// =============================================================================================
// PQF: the headline path — PQ approximate search on the FUSED layout, single pool, level 0 only.
//
// Differences to the generic pool loop (results identical; tests/test_gpu_parity.py):
//  * no visited set inside the loop.  A re-encountered node has the same ADC score as before, so it is
//    either still in the pool (the rank-merge finds the identical key and drops the duplicate) or it was
//    dropped below the pool boundary, which only rises, so it is dropped again.  jvector's visitedCount
//    (= distinct nodes over all expanded rows, entry point excluded) is computed AFTER the search from the
//    log of expanded nodes, with a hash set that reuses the LUT's LDS.  That frees the 16-32 KB visited
//    table during the search: 38 KB per query instead of 73 KB -> 4 resident queries per CU instead of 2.
//  * merge bookkeeping lives in registers (v_readlane broadcasts), the unexpanded entries are tracked by
//    ballot bit masks, and the pool is merged in place.
// Bails out (JV_FLAG_OVERFLOW -> generic retry kernel) on: negative score vs threshold, > 64 boundary
// ties, expansion log overflow, visited-count table overflow.
// =============================================================================================
// CH = pool capacity in 64-entry chunks (template parameter): 4, 8 or 16 (rerankK + 64 + R <= 64 * CH)
// NP = compile-time bound on fused-block passes: 1 (R * lanes-per-node <= 64, the common case) or 4

// FILT = the query has a doc filter (J/JVectorReader.java:157-163): entries carry an "accepted" bit, the boundary is
// the rerankK-th best ACCEPTED entry (jvector's result queue only holds accepted nodes, but every node scoring at
// least as well as its worst entry stays a candidate), NP == 1 only.
// FAST = pq_M is a multiple of 16 and the similarity is not cosine: only the unmasked LDS look-up is compiled
// (the masked and norm-table variants cost ~70 scalar registers of hoisted lane predicates even when unused).
template <int NCHT, int CH, int NP, bool FILT, bool FAST>
__device__ __forceinline__ void search_one_pqf(const JvIndexDev& ix, const JvSearchArgs& a, int qi, unsigned char* smem) {
    const int lane = threadIdx.x;
    const uint64_t* const accw = FILT ? a.accept + (size_t)qi * (size_t)a.accept_stride : nullptr;
    auto accepts = [&](int node) -> bool {  // the reference's acceptOrds lambda
        const int doc = ix.ord2doc ? ix.ord2doc[node] : node;
        return doc >= 0 && (int64_t)doc < a.accept_docs && ((accw[doc >> 6] >> (doc & 63)) & 1ull);
    };
    auto pnode = [&](int64_t pk) -> int { return FILT ? pool_node_f(pk) : pool_node(pk); };
    if (FILT) {
        // Rung choice only (never results): estimate the filter's selectivity from 64 sampled words; a pool of
        // ~ rerankK / selectivity entries that cannot fit this launch's capacity is handed on right away instead
        // of after a wasted search.
        const int64_t nwords = (a.accept_docs + 63) >> 6;
        int bits = 0;
        if (nwords > 0) {
            const uint64_t h = ((uint64_t)(lane + 1) * 0x9E3779B97F4A7C15ull) >> 20;
            bits = __popcll(accw[nwords <= JV_WAVE ? (int64_t)(lane % (int)nwords) : (int64_t)(h % (uint64_t)nwords)]);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) bits += __shfl_xor(bits, o, JV_WAVE);
        const float sel = fmaxf((float)bits, 1.0f) * (1.0f / 4096.0f);
        const float need = (float)a.rk / sel * 1.05f + 64.0f + (float)ix.R;
        if (need > (float)a.cand_cap && a.cand_cap < 8 * a.rk + 256) {
            if (lane == 0) {
                a.out_flags[qi] = (int32_t)(JV_FLAG_OVERFLOW | (3u << 8));
                a.out_count[qi] = 0;
            }
            int32_t* o_n = a.out_nodes + (size_t)qi * a.topK;
            for (int i = lane; i < a.topK; i += JV_WAVE) {
                o_n[i] = -1;
                if (a.out_docs) a.out_docs[(size_t)qi * a.topK + i] = -1;
                a.out_scores[(size_t)qi * a.topK + i] = 0.0f;
            }
            return;
        }
    }
    const int rk = a.rk, topK = a.topK;
    const int M = ix.pq_M, R = ix.R, lpn = ix.pq_lanes, cs = ix.pq_code_stride;
    // ---- LDS carve ----
    float* lut = (float*)smem;                                   // [M][256]; later: visited-count hash, then rerank scratch
    size_t off = (size_t)M * 256 * sizeof(float);
    int64_t* pool = (int64_t*)(smem + off);                      // [cap]
    const int cap = a.cand_cap;                                  // rk + 64 boundary ties + R new keys, <= 64 * CH
    off += (size_t)cap * sizeof(int64_t);
    int32_t* explog = (int32_t*)(smem + off);                    // [a.res_cap] expanded nodes, in order
    const int log_cap = a.res_cap;
    float* qc_lds = (float*)pool;                                // LUT build only: aliases pool + log (host guarantees room)
    const int pool_limit = cap - ix.R;                           // one merge adds at most R keys

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

    // A node's fused block = R neighbours x lpn 16-B code chunks.  One pass covers 64/lpn neighbours
    // (lane = slot * lpn + chunk); blocks with R*lpn > 64 take npass <= lpn passes, and pass p's scores are
    // moved to the lanes with chunk index p so that all R new keys sit on distinct lanes for ONE merge.
    const int my_c = lane & (lpn - 1);
    const int jpp = JV_WAVE / lpn;                       // neighbours per pass
    const int npass = NP == 1 ? 1 : (R * lpn + JV_WAVE - 1) / JV_WAVE; // <= NP, <= lpn (host checks)
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

    // ---- entry point ----
    int np = 0, nexp = 0, expanded = 0;
    bool overflow = false, early = false;
    int why = 0;  // bail-out reason, reported in bits 8..11 of the flag word (diagnostics)
    float bscore = 0.0f;  // score of pool[rk-1] once the pool holds rk entries (the boundary)
    // level-1 pivots of the rank search (pool[63], pool[127], ...): wave-uniform, refreshed from the registers
    // of the batched pool read after every merge instead of being re-read from LDS in every expansion
    int64_t piv[CH - 1];
#pragma unroll
    for (int t = 0; t < CH - 1; t++) piv[t] = KEY_MIN;
    unsigned long long um[CH];
    unsigned long long am[FILT ? CH : 1];  // FILT: accepted entries
    bool have_b = false;                   // FILT: the pool holds >= rk accepted entries, bscore is their rk-th best
#pragma unroll
    for (int t = 0; t < CH; t++) um[t] = 0ull;
#pragma unroll
    for (int t = 0; t < (FILT ? CH : 1); t++) am[t] = 0ull;
    {
        const int ep = ix.entry;
        u32x4 cw = (u32x4){0, 0, 0, 0};
        if (lane < lpn && my_chunk) cw = *(const u32x4*)(ix.pq_codes + (size_t)ep * cs + my_c * 16);
        const float s = adc_score(cw, lane < lpn && my_chunk);
        bool acc_ep = true;
        if (FILT) acc_ep = accepts(ep);
        if (lane == 0) pool[0] = FILT ? make_pool_key_f(s, ep, acc_ep) : make_pool_key(s, ep);
        np = 1;
        um[0] = 1ull;
        __syncthreads();
        if (FILT) {
            am[0] = acc_ep ? 1ull : 0ull;
            have_b = acc_ep && rk <= 1;
            bscore = key_score(pool[0]);  // (s is only valid on the entry point's lanes)
        }
    }

    int pf_node = -1;
    int pf_nn[NP];
    u32x4 pf_cw[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ps++) pf_nn[ps] = -1, pf_cw[ps] = (u32x4){0, 0, 0, 0};
    STAMP_DECL
    STAMP(7)  // staging + LUT build + entry point
    while (true) {
#ifdef JV_STAMPS
        // diagnostic build: account the exposed part of the prefetch latency separately (slot 7)
        STAMP(5)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(7)
#endif
        // best and runner-up unexpanded entries from the masks: per-word first-set-bit + min, the best entry's bit is
        // cleared right away (it is expanded below or the query bails), then the same again for the runner-up.
        // (A single wave issues a scalar instruction every ~4 cycles: this form is half the instructions of
        // selecting the words through compare/select chains.)
        int idx = 0x7fffffff, idx2 = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < CH; t++) idx = min(idx, um[t] ? (t << 6) + __ffsll((long long)um[t]) - 1 : 0x7fffffff);
        if (idx == 0x7fffffff) break;
#pragma unroll
        for (int t = 0; t < CH; t++) um[t] &= ~((t == (idx >> 6)) ? (1ull << (idx & 63)) : 0ull);
#pragma unroll
        for (int t = 0; t < CH; t++) idx2 = min(idx2, um[t] ? (t << 6) + __ffsll((long long)um[t]) - 1 : 0x7fffffff);
        if (idx2 == 0x7fffffff) idx2 = -1;
        const int64_t pk = pool[idx];
        int64_t pk2 = pool[max(idx2, 0)];  // unconditional: both reads share one LDS round trip
        {   // (keeps the compiler from sinking the second read into the idx2 >= 0 branch, behind the first wait)
            int lo2 = (int)(uint32_t)(pk2 & 0xFFFFFFFFll);
            asm volatile("" : "+v"(lo2));
            pk2 = (int64_t)(((uint64_t)pk2 & 0xFFFFFFFF00000000ull) | (uint64_t)(uint32_t)lo2);
        }
        const int c2 = idx2 >= 0 ? pnode(pk2) : -1;
        const float sc = key_score(pk);
        if (sc < a.threshold) {
            overflow = true;
            why = 1;
            break;
        }
        // strict-admission tie (see the generic pool loop): the expanded entries scoring >= the candidate already
        // fill the result queue and the candidate ranks inside the top rerankK -> the two-queue form decides
        if (!FILT && expanded >= rk && idx < rk) {
            int ge = idx;
            for (int j = idx + 1; j < np && ge < rk; j++) {
                const int64_t kj = pool[j];
                if (key_score(kj) != sc) break;
                ge += (kj & 1ll) ? 0 : 1;
            }
            const bool tie_bail = ge >= rk;
            if (tie_bail) {
                overflow = true;
                why = 5;
                break;
            }
        }
        if (FILT && expanded >= rk && (pk & 2ll)) {
            // same rule over the ACCEPTED entries (only they enter jvector's result queue): all entries ahead of
            // the candidate are expanded; the accepted ones among them + the accepted, expanded, equal-score
            // entries behind it are the results scoring >= the candidate
            int ge = 0;
#pragma unroll
            for (int t = 0; t < CH; t++) {
                const int lo = t << 6;
                const unsigned long long below = idx >= lo + 64 ? ~0ull : (idx > lo ? ((1ull << (idx - lo)) - 1ull) : 0ull);
                ge += __popcll(am[t] & below);
            }
            if (ge < rk) {
                for (int j = idx + 1; j < np && ge < rk; j++) {
                    const int64_t kj = pool[j];
                    if (key_score(kj) != sc) break;
                    ge += ((kj & 3ll) == 2ll) ? 1 : 0;
                }
                if (ge >= rk) {
                    overflow = true;
                    why = 5;
                    break;
                }
            }
        }
        const int c = pnode(pk);
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
        // FILT: the neighbours' accept bits (ord -> doc -> bitset word) are fetched while the ADC runs
        bool accn = true;
        if (FILT) {
            accn = false;
            if (nnp[0] >= 0 && my_c == 0) accn = accepts(nnp[0]);
        }
#ifdef JV_STAMPS
        asm volatile("" ::"v"(c2), "v"(c));
#endif
        STAMP(0)  // masks -> idx, pool reads
#ifdef JV_STAMPS
        asm volatile("" ::"v"(nnp[0]), "v"(cwp[0][0]), "v"(cwp[0][3]));
#endif
        // start the runner-up's fetch now (this expansion's block is older in the load queue, so waiting for it
        // does not wait for the prefetch): ADC + merge (~3 500 cycles) cover its HBM latency
        // The loads are UNCONDITIONAL (clamped indices, every lane, also when there is no runner-up): a fixed number
        // of younger loads is what lets the wait for this expansion's block be vmcnt(2) instead of vmcnt(0).
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
        STAMP(1)  // wait for the block
        if (nexp >= log_cap) {
            overflow = true;
            why = 2;
            break;
        }
        if (a.visit_limit > 0 && expanded >= a.visit_limit) {  // (visited is only known after the loop: expansions alone reach it)
            early = true;
            break;
        }
        if (lane == 0) {
            pool[idx] = pk & ~1ll;
            explog[nexp] = c;
        }
        nexp++;
        // ADC of all R stored neighbours; pass ps delivers its scores to the lanes whose chunk index is ps
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
        // ---- merge the new keys (one per neighbour, on the lanes with my_c == 0) ----
        // All LDS reads below are unconditional (clamped index) and batched into registers first: a
        // conditional load would compile into an exec-masked branch with its own LDS round trip.
        bool keep = nn >= 0 && my_c < npass;
        if ((FILT ? have_b : np >= rk) && score < bscore) keep = false;  // below the boundary for good
        const int64_t v = FILT ? make_pool_key_f(score, nn, accn) : make_pool_key(score, nn);
        int rold;
        {   // rold = #{pool entries > v} = first index whose entry is <= v.  Uniform 3-level 8-ary search over
            // the pool: block sizes 64, 8, 1.  (Skipping the search when no key passes the boundary was measured:
            // the branch costs more than the skipped searches save.)
            const int last = cap - 1;
            int64_t p2[7], p3[9];
            int c1 = 0;
#pragma unroll
            for (int k2 = 0; k2 < CH - 1; k2++) c1 += ((k2 * 64 + 63 < np) & (piv[k2] > v)) ? 1 : 0;
            int lo = c1 * 64;
#pragma unroll
            for (int k2 = 0; k2 < 7; k2++) p2[k2] = pool[min(lo + k2 * 8 + 7, last)];
            int c2 = 0;
#pragma unroll
            for (int k2 = 0; k2 < 7; k2++) c2 += ((lo + k2 * 8 + 7 < np) & (p2[k2] > v)) ? 1 : 0;
            lo += c2 * 8;
#pragma unroll
            for (int k2 = 0; k2 < 9; k2++) p3[k2] = pool[min(lo + k2, last)];
            int c3 = 0;
            bool dup = false;  // same node => same score => same key up to the expanded bit
#pragma unroll
            for (int k2 = 0; k2 < 9; k2++) {
                const bool in = lo + k2 < np;
                if (k2 < 8) c3 += (in & (p3[k2] > v)) ? 1 : 0;
                dup |= in & ((p3[k2] | 1ll) == v);
            }
            rold = lo + c3;
            if (dup) keep = false;
        }
        unsigned long long km = __ballot(keep);
        int nk = __popcll(km);
        STAMP(3)  // boundary check + 3-level rank search + duplicate check
        if (nk > 0) {
            // rank among the kept new keys, the first insertion point, and per-chunk shift counts
            int rnew = 0;
            int cnt[CH];
            const int vlo = (int)(uint32_t)(v & 0xFFFFFFFFll), vhi = (int)(v >> 32);
            for (int attempt = 0; attempt < 2; attempt++) {
                rnew = 0;
#pragma unroll
                for (int t = 0; t < CH; t++) cnt[t] = 0;
                bool twin = false;  // the same neighbour twice in one adjacency row (malformed graph): keep one
                for (unsigned long long m = km; m;) {
                    const int j = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    const int64_t kj = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(vhi, j) << 32) |
                                                 (uint64_t)(uint32_t)__builtin_amdgcn_readlane(vlo, j));
                    const int rj = __builtin_amdgcn_readlane(rold, j);
                    rnew += kj > v ? 1 : 0;
                    twin |= kj == v && j < lane;
#pragma unroll
                    for (int t = 0; t < CH; t++) cnt[t] += ((t << 6) + lane >= rj) ? 1 : 0;  // key j precedes entry
                }
                const unsigned long long km2 = __ballot(keep && !twin);
                if (km2 == km) break;
                keep = keep && !twin;
                km = km2;
                nk = __popcll(km);
            }
            const unsigned long long firstm = __ballot(keep && rnew == 0);
            const int r_min = __builtin_amdgcn_readlane(rold, __ffsll((long long)firstm) - 1);
            // read every old entry first, then write the shifted ones: in place, no ordering hazard
            int64_t ov[CH];
#pragma unroll
            for (int t = 0; t < CH; t++) ov[t] = pool[min((t << 6) + lane, cap - 1)];
#pragma unroll
            for (int t = 0; t < CH; t++) {
                const int i = (t << 6) + lane;
                if (i < np && cnt[t] > 0) pool[i + cnt[t]] = ov[t];
            }
            if (keep) pool[rold + rnew] = v;
            __syncthreads();
            STAMP(4)  // ranks among new keys + in-place shift + insert
            // trim to the boundary (+ ties) and rebuild the unexpanded masks, one batched read of the pool
            const int ntot = np + nk;
#pragma unroll
            for (int t = 0; t < CH; t++) ov[t] = pool[min((t << 6) + lane, cap - 1)];
            int nnew = ntot;
            if (FILT) {
                int nacc = 0;
#pragma unroll
                for (int t = 0; t < CH; t++) {
                    am[t] = __ballot((t << 6) + lane < ntot && (ov[t] & 2ll));
                    nacc += __popcll(am[t]);
                }
                have_b = nacc >= rk;
                if (have_b) {
                    int need = rk, bpos = -1;
#pragma unroll
                    for (int t = 0; t < CH; t++) {
                        const int ct = __popcll(am[t]);
                        if (bpos < 0) {
                            if (need <= ct) bpos = (t << 6) + select_nth_bit(am[t], need);
                            else need -= ct;
                        }
                    }
                    bscore = key_score(pool[bpos]);
                    int ties = 0;
#pragma unroll
                    for (int t = 0; t < CH; t++) {
                        const int i = (t << 6) + lane;
                        ties += __popcll(__ballot(i > bpos && i < ntot && key_score(ov[t]) == bscore));
                    }
                    nnew = bpos + 1 + ties;
                }
            } else if (ntot > rk) {
                const int64_t bk = pool[rk - 1];
                bscore = key_score(bk);
                int ties = 0;  // entries beyond rk-1 that tie with the boundary score stay (they are contiguous)
#pragma unroll
                for (int t = 0; t < CH; t++) {
                    const int i = (t << 6) + lane;
                    ties += __popcll(__ballot(i >= rk && i < ntot && key_score(ov[t]) == bscore));
                }
                nnew = rk + ties;
            } else if (ntot == rk) {
                bscore = key_score(pool[rk - 1]);
            }
            np = nnew;
            if (np > pool_limit) {
                overflow = true;
                why = 3;
                break;
            }
#pragma unroll
            for (int t = 0; t < CH; t++) {
                const int i = (t << 6) + lane;
                um[t] = __ballot(i < np && (ov[t] & 1ll));
                if (FILT) am[t] &= __ballot(i < np);
                if (t < CH - 1) {  // pool[64t + 63] sits in lane 63 of ov[t]
                    const int plo = __builtin_amdgcn_readlane((int)(uint32_t)(ov[t] & 0xFFFFFFFFll), 63);
                    const int phi = __builtin_amdgcn_readlane((int)(ov[t] >> 32), 63);
                    piv[t] = (int64_t)(((uint64_t)(uint32_t)phi << 32) | (uint64_t)(uint32_t)plo);
                }
            }
            (void)r_min;
            STAMP(5)  // trim + mask rebuild
        }
    }
    STAMP(5)

    int32_t* o_nodes = a.out_nodes + (size_t)qi * topK;
    int32_t* o_docs = a.out_docs ? a.out_docs + (size_t)qi * topK : nullptr;
    float* o_scores = a.out_scores + (size_t)qi * topK;
    int visited = 0;
    if (early && !overflow) {
        if (lane == 0) {
            a.out_flags[qi] = (int32_t)JV_FLAG_EARLY;
            a.out_count[qi] = 0;
            int32_t* st = a.out_stats + (size_t)qi * 4;
            st[0] = 0;
            st[1] = 0;
            st[2] = expanded;
            st[3] = expanded;
        }
        for (int i = lane; i < topK; i += JV_WAVE) {
            o_nodes[i] = -1;
            if (o_docs) o_docs[i] = -1;
            o_scores[i] = 0.0f;
        }
        return;
    }
    if (!overflow) {
        // ---- jvector's visitedCount: distinct neighbours of the expanded nodes, entry point excluded ----
        uint32_t* vh = (uint32_t*)lut;
        int vslots = 1;
        while (vslots * 2 <= M * 256) vslots <<= 1;
        const uint32_t vmask = (uint32_t)vslots - 1u;
        const int vshift = 32 - (31 - __clz(vslots));
        const int vlimit = (vslots / 16) * 15;
        __syncthreads();
        for (int i = lane; i < vslots; i += JV_WAVE) vh[i] = HASH_EMPTY;
        __syncthreads();
        if (lane == 0) visited_insert_lds(vh, vmask, vshift, (uint32_t)ix.entry);
        __syncthreads();
        const int rows_per = JV_WAVE / R > 0 ? JV_WAVE / R : 1;  // adjacency rows per wave-instruction
        if (R <= JV_WAVE) {
            constexpr int VB = 8;  // adjacency batches per group; the NEXT group's rows are in flight while this one probes
            // unconditional loads with clamped indices (masked when consumed): a fixed number of loads per group is
            // what lets the wait for THIS group's rows leave the next group's in flight (vmcnt(VB), not vmcnt(0))
            auto load_group = [&](int e0, int (&dst)[VB]) {
#pragma unroll
                for (int u = 0; u < VB; u++) {
                    const int e = min(e0 + u * rows_per + lane / R, nexp - 1);
                    dst[u] = ix.adj[(size_t)explog[e] * R + (lane % R)];
                }
            };
            int nb_next[VB];
            load_group(0, nb_next);
            for (int e0 = 0; e0 < nexp && !overflow; e0 += rows_per * VB) {
                int nb[VB];
#pragma unroll
                for (int u = 0; u < VB; u++) {
                    const int e = e0 + u * rows_per + lane / R;
                    nb[u] = (e < nexp && lane < rows_per * R) ? nb_next[u] : -1;
                }
                load_group(e0 + rows_per * VB, nb_next);
                int pending = 0;
#pragma unroll
                for (int u = 0; u < VB; u++) pending += __popcll(__ballot(nb[u] >= 0));
                if (visited + pending > vlimit) {
                    overflow = true;
                    why = 4;
                    break;
                }
                // all VB batches probe together: the compare-and-swaps of one round are issued back to back and
                // their LDS latencies overlap (the distinct count does not depend on the insertion order)
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
                        visited += __popcll(__ballot(fresh));
                        if (pend[u]) {
                            if (fresh || oldv[u] == (uint32_t)nb[u]) pend[u] = false;
                            else hh[u] = (hh[u] + 1) & vmask, more = true;
                        }
                    }
                    if (!__any(more)) break;
                }
            }
        } else {
            for (int e0 = 0; e0 < nexp && !overflow; e0++) {
                for (int cb = 0; cb < R; cb += JV_WAVE) {
                    if (visited + JV_WAVE > vlimit) {
                        overflow = true;
                        break;
                    }
                    const int nb = (cb + lane < R) ? ix.adj[(size_t)explog[e0] * R + cb + lane] : -1;
                    bool is_new = false;
                    if (nb >= 0) is_new = visited_insert_lds(vh, vmask, vshift, (uint32_t)nb);
                    visited += __popcll(__ballot(is_new));
                }
            }
        }
        __syncthreads();
    }
    STAMP(6)  // visited-count pass
    STAMP_FLUSH
    if (overflow) {
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
    for (int i = lane; i < ix.nch * 64; i += JV_WAVE) q_lds[i] = i < ix.d ? qg[i] : 0.0f;
    __syncthreads();
    if (ix.sim == 2) qnorm2 = query_norm2(ix, q_lds, lane), qnorm2 = __shfl(qnorm2, 0, JV_WAVE);
    if (FILT) {
        // jvector's result queue = the accepted entries, best rerankK of them: compact them to the front, in order
        // (a chunk is read into registers before anything is written, and writes never pass the read position)
        int w = 0;
        for (int b0 = 0; b0 < np; b0 += JV_WAVE) {
            const int i = b0 + lane;
            const int64_t k = i < np ? pool[i] : 0;
            const bool t_ = i < np && (k & 2ll);
            const unsigned long long tm = __ballot(t_);
            __syncthreads();
            if (t_) pool[w + __popcll(tm & ((1ull << lane) - 1ull))] = k;
            w += __popcll(tm);
            __syncthreads();
        }
        np = w;
    }
    if (!overflow && a.visit_limit > 0 && visited + expanded >= a.visit_limit) {
        // the search ran to its end, but visited + expanded reaches Lucene's visit limit: discarded like the stopped ones,
        // with its real counters (include/jvgpu.h, jv_search_params.visit_limit)
        if (lane == 0) {
            a.out_flags[qi] = (int32_t)JV_FLAG_EARLY;
            a.out_count[qi] = 0;
            int32_t* st = a.out_stats + (size_t)qi * 4;
            st[0] = visited;
            st[1] = 0;
            st[2] = expanded;
            st[3] = expanded;
        }
        for (int i = lane; i < topK; i += JV_WAVE) {
            o_nodes[i] = -1;
            if (o_docs) o_docs[i] = -1;
            o_scores[i] = 0.0f;
        }
        return;
    }
    const int nres = np < rk ? np : rk;
    int nfin = 0, reranked = 0;
    int above = 0;
    for (int i = lane; i < nres; i += JV_WAVE) above += key_score(pool[i]) >= a.rerank_floor ? 1 : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) above += __shfl_xor(above, o, JV_WAVE);
    if (above == 0 && nres >= 2 && key_score(pool[0]) == key_score(pool[1])) {
        // rerankFloor above every approximate score AND a tie at the best one: jvector rescores the first best entry of
        // its result heap's array, which only the HBM-scratch rung reconstructs (replay_first_best)
        if (lane == 0) {
            a.out_flags[qi] = (int32_t)(JV_FLAG_OVERFLOW | (6u << 8));
            a.out_count[qi] = 0;
        }
        for (int i = lane; i < topK; i += JV_WAVE) {
            o_nodes[i] = -1;
            if (o_docs) o_docs[i] = -1;
            o_scores[i] = 0.0f;
        }
        return;
    }
    for (int b0 = 0; b0 < nres; b0 += JV_WAVE) {
        const int i = b0 + lane;
        bool take = false;
        int node = 0;
        if (i < nres) {
            const int64_t k = pool[i];
            node = pnode(k);
            take = above > 0 ? key_score(k) >= a.rerank_floor : i == 0;  // pool[0] is the best approximate entry
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

template <int NCHT, int CH, int NP, bool FAST>
__global__ __launch_bounds__(JV_WAVE) void jv_search_pqf_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int qi = blockIdx.x;
    if (qi >= a.nq) return;
    search_one_pqf<NCHT, CH, NP, false, FAST>(ix, a, qi, smem);
}

// the same search with a per-query doc filter (own name: profiles keep filtered launches apart)
template <int NCHT, int CH, bool FAST>
__global__ __launch_bounds__(JV_WAVE) void jv_search_pqf_filtered_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!a.retry_only) {
        const int qi = blockIdx.x;
        if (qi >= a.nq) return;
        search_one_pqf<NCHT, CH, 1, true, FAST>(ix, a, qi, smem);
        return;
    }
    // second rung (larger pool + log): only the queries whose pool or expansion log overflowed in the first launch
    const int lane = threadIdx.x;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(a.retry_counter, 8);  // small chunks: most queries may be flagged here
        base = __shfl(base, 0, JV_WAVE);
        if (base >= a.nq) break;
        const int qi = base + lane;
        bool flagged = false;
        if (lane < 8 && qi < a.nq) {
            const uint32_t f = (uint32_t)a.out_flags[qi];
            const uint32_t why = (f >> 8) & 0xFu;
            flagged = (f & JV_FLAG_OVERFLOW) && (why == 2u || why == 3u);
        }
        unsigned long long m = __ballot(flagged);
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            search_one_pqf<NCHT, CH, 1, true, FAST>(ix, a, base + j, smem);
            __syncthreads();
        }
    }
}

// Fast path: one query per workgroup, all scratch in LDS.
template <bool PQ, bool POOL, int NCHT>
__global__ __launch_bounds__(JV_WAVE) void jv_search_lds_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int qi = blockIdx.x;
    if (qi >= a.nq) return;
    search_one<PQ, false, POOL, NCHT>(ix, a, qi, smem, nullptr, nullptr);
}

// Escalation launch (same code, own name so profiles separate it from the main launch): only the queries
// an earlier, smaller launch flagged as overflowed, with a 4x larger visited table.
template <bool PQ, bool POOL, int NCHT>
__global__ __launch_bounds__(JV_WAVE) void jv_search_retry_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // A small resident grid walks the flag array 64 queries at a time (one flag per lane) and runs only the
    // flagged ones: a launch with nothing to redo costs microseconds instead of dispatching nq large-LDS blocks.
    const int lane = threadIdx.x;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(a.retry_counter, JV_WAVE);
        base = __shfl(base, 0, JV_WAVE);
        if (base >= a.nq) break;
        const int qi = base + lane;
        const bool flagged = qi < a.nq && ((uint32_t)a.out_flags[qi] & JV_FLAG_OVERFLOW);
        unsigned long long m = __ballot(flagged);
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            search_one<PQ, false, POOL, NCHT>(ix, a, base + j, smem, nullptr, nullptr);
            __syncthreads();
        }
    }
}

// The same search launched on behalf of the graph builder (index created with JV_DESC_BUILD_CLIENT): a
// separate kernel name keeps construction-time searches out of the query kernels' profile rows.
template <bool POOL, int NCHT>
__global__ __launch_bounds__(JV_WAVE) void jv_build_search_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int qi = blockIdx.x;
    if (qi >= a.nq) return;
    search_one<false, false, POOL, NCHT>(ix, a, qi, smem, nullptr, nullptr);
}

// Big path: queues and visited bitset in HBM scratch; each resident workgroup dequeues the queries the
// fast path flagged as overflowed.  Exact in all cases the fast path cannot hold on chip.
template <bool PQ, int NCHT, bool QLDS = false, bool LUTG = false>
__global__ __launch_bounds__(JV_WAVE) void jv_search_big_kernel(const JvIndexDev ix, const JvSearchArgs a,
                                                                 const int force_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int words = (ix.n + 31) >> 5;
    int64_t* my_cand = a.big_cand + (size_t)blockIdx.x * a.big_cand_cap;
    uint32_t* my_bits = a.big_visited + (size_t)blockIdx.x * words;
    const int lane = threadIdx.x;
    // dequeue up to 64 queries at a time, one flag per lane.  (A handful of 65 536 flags are set behind the pool kernels: big
    // chunks.  Where this rung carries a whole small batch — filters on tables beyond the LDS, forced runs — 64 queries per
    // workgroup would leave all but a few workgroups idle: the chunk shrinks with the number of flags per workgroup.)
    const int chunk = max(1, min(JV_WAVE, a.nq / ((int)gridDim.x * 2)));
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(a.work_counter, chunk);
        base = __shfl(base, 0, JV_WAVE);
        if (base >= a.nq) break;
        const int qi = base + lane;
        const bool todo_q = lane < chunk && qi < a.nq && (force_all || ((uint32_t)a.out_flags[qi] & JV_FLAG_OVERFLOW));
        unsigned long long m = __ballot(todo_q);
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            search_one<PQ, true, false, NCHT, QLDS, LUTG>(ix, a, base + j, smem, my_cand, my_bits);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Exact scorer over an ordinal list (JVectorVectorScorer.score, J/JVectorVectorScorer.java:36-53):
// one wave scores 64 ordinals per step with the same canonical accumulation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(JV_WAVE) void jv_score_ordinals_kernel(const JvIndexDev ix, const float* query,
                                                                     const int32_t* ordinals, int count,
                                                                     float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    float* q_lds = (float*)smem;
    size_t off = (size_t)ix.nch * 64 * sizeof(float);
    float* todo_score = (float*)(smem + off);
    off += JV_TODO * sizeof(float);
    int32_t* todo = (int32_t*)(smem + off);
    for (int i = lane; i < ix.nch * 64; i += JV_WAVE) q_lds[i] = i < ix.d ? query[i] : 0.0f;
    __syncthreads();
    float qnorm2 = 0.0f;
    if (ix.sim == 2) qnorm2 = query_norm2(ix, q_lds, lane), qnorm2 = __shfl(qnorm2, 0, 64);
    for (int base = blockIdx.x * JV_WAVE; base < count; base += gridDim.x * JV_WAVE) {
        const int i = base + lane;
        const int o = i < count ? ordinals[i] : -1;
        const bool ok = o >= 0 && o < ix.n;
        const unsigned long long mk = __ballot(ok);
        const int m = __popcll(mk);
        const int pos = __popcll(mk & ((1ull << lane) - 1ull));
        if (ok) todo[pos] = o;
        __syncthreads();
        if (m > 0) score_rows<0>(ix, q_lds, todo, m, todo_score, qnorm2, ix.score_scale, lane);
        __syncthreads();
        if (i < count) out[i] = ok ? todo_score[pos] : 0.0f;  // NO_VECTOR_OR_DELETED_DOC -> 0
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Merge of per-shard top-k lists (TopDocs.merge; the exchange step after the RCCL all-gather):
// one wave per query, lists*k candidates -> k best by (score desc, doc asc).
// ---------------------------------------------------------------------------------------------
// LISTMAJOR: input = (doc, score) 8-byte pairs laid out [lists][nq][k] (each shard's block as it arrives from its device)
// instead of separate doc / score arrays [nq][lists*k]
template <bool LISTMAJOR>
__global__ __launch_bounds__(JV_WAVE) void jv_merge_topk_kernel(const int32_t* docs, const float* scores, int nq,
                                                                 int lists, int k, int32_t* out_docs,
                                                                 float* out_scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int64_t* keys = (int64_t*)smem;
    const int lane = threadIdx.x;
    const int qi = blockIdx.x;
    if (qi >= nq) return;
    const int total = lists * k;
    for (int i = lane; i < total; i += JV_WAVE) {
        int doc;
        float sc;
        if (LISTMAJOR) {  // `docs` holds (doc, score bits) pairs, `scores` is unused
            const size_t src = ((size_t)(i / k) * (size_t)nq + (size_t)qi) * (size_t)k + (size_t)(i % k);
            doc = docs[2 * src];
            sc = __int_as_float(docs[2 * src + 1]);
        } else {
            doc = docs[(size_t)qi * total + i];
            sc = scores[(size_t)qi * total + i];
        }
        keys[i] = doc >= 0 ? make_key(sc, doc) : KEY_MIN;
    }
    __syncthreads();
    int n = total;
    for (int r = 0; r < k; r++) {
        int64_t bk;
        int bi;
        scan_max(keys, n, lane, bk, bi);
        if (lane == 0) {
            const bool ok = n > 0 && bk != KEY_MIN;
            out_docs[(size_t)qi * k + r] = ok ? key_node(bk) : -1;
            out_scores[(size_t)qi * k + r] = ok ? key_score(bk) : 0.0f;
            if (n > 0) keys[bi] = keys[n - 1];
        }
        if (n > 0) n--;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Fused ADC layout builder: fused[u][j][:] = codes[adj[u][j]][:] (zeros for -1), 16 B per thread.
// ---------------------------------------------------------------------------------------------
__global__ void jv_build_fused_kernel(const uint8_t* codes, const int32_t* adj, uint8_t* fused, long long n, int R,
                                      int cs) {
    const int cpn = cs / 16;
    const long long total = n * R * cpn;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long slot = i / cpn;
        const int ch = (int)(i % cpn);
        const int nb = adj[slot];
        u32x4 v = (u32x4){0, 0, 0, 0};
        if (nb >= 0) v = *(const u32x4*)(codes + (size_t)nb * cs + ch * 16);
        *(u32x4*)(fused + (size_t)slot * cs + ch * 16) = v;
    }
}

extern "C" hipError_t jvk_build_fused(const uint8_t* codes, const int32_t* adj, uint8_t* fused, long long n, int R, int cs,
                                      hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    jv_build_fused_kernel<<<4096, 256, 0, stream>>>(codes, adj, fused, n, R, cs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Cosine on the fused layout: |decoded code vector|^2 per node = the norm table's entries of its code row in the CANONICAL order
// (16-subspace chunks summed left to right, slots beyond M contribute +0.0f; chunk sums combined by the adjacent-pair tree over
// pq_lanes chunks: adc_chunk + lanes_tree_sum, oracle/jv_oracle.c jvo_pq_raw) — one thread per node — and the same value per
// adjacency slot, next to the fused codes' order.
// ---------------------------------------------------------------------------------------------
__global__ void jv_node_norm_kernel(const uint8_t* codes, const float* norm_lut, float* out, long long n, int M, int cs, int lpn) {
    for (long long node = (long long)blockIdx.x * blockDim.x + threadIdx.x; node < n; node += (long long)gridDim.x * blockDim.x) {
        const uint8_t* code = codes + (size_t)node * cs;
        float c[16];
        for (int w = 0; w < 16; w++) {
            float s = 0.0f;
            if (w < lpn && w * 16 < M) {
                for (int i = 0; i < 16; i++) {
                    const int mi = w * 16 + i;
                    const float t = mi < M ? norm_lut[mi * 256 + code[mi]] : 0.0f;
                    s = s + t;
                }
            }
            c[w] = s;
        }
        for (int span = 1; span < lpn; span <<= 1)
            for (int w = 0; w + span < 16; w += 2 * span) c[w] = c[w] + c[w + span];
        out[node] = c[0];
    }
}
__global__ void jv_fused_norm_kernel(const float* node_norm, const int32_t* adj, float* out, long long slots) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (long long)gridDim.x * blockDim.x) {
        const int nb = adj[i];
        out[i] = nb >= 0 ? node_norm[nb] : 0.0f;
    }
}
extern "C" hipError_t jvk_build_code_norms(const uint8_t* codes, const float* norm_lut, const int32_t* adj, float* node_norm, float* fused_norm,
                                           long long n, int R, int M, int cs, int lpn, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if (lpn > 16) return hipErrorInvalidValue;
    jv_node_norm_kernel<<<2048, 256, 0, stream>>>(codes, norm_lut, node_norm, n, M, cs, lpn);
    jv_fused_norm_kernel<<<4096, 256, 0, stream>>>(node_norm, adj, fused_norm, n * R);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from jv_abi.cpp)
// ---------------------------------------------------------------------------------------------
// kernel tables: [pq][pool][nch slot]; nch slots: 0 -> any d (runtime chunk loop), 1 -> d = 128,
// 2 -> d = 768, 3 -> d = 1536
typedef void (*lds_kernel_t)(const JvIndexDev, const JvSearchArgs);
typedef void (*big_kernel_t)(const JvIndexDev, const JvSearchArgs, const int);
#define JV_ROW(K, ...) { K<__VA_ARGS__, 0>, K<__VA_ARGS__, 2>, K<__VA_ARGS__, 12>, K<__VA_ARGS__, 24> }
static const lds_kernel_t g_lds_kernels[2][2][4] = {
    {JV_ROW(jv_search_lds_kernel, false, false), JV_ROW(jv_search_lds_kernel, false, true)},
    {JV_ROW(jv_search_lds_kernel, true, false), JV_ROW(jv_search_lds_kernel, true, true)}};
static const lds_kernel_t g_retry_kernels[2][2][4] = {
    {JV_ROW(jv_search_retry_kernel, false, false), JV_ROW(jv_search_retry_kernel, false, true)},
    {JV_ROW(jv_search_retry_kernel, true, false), JV_ROW(jv_search_retry_kernel, true, true)}};
static const lds_kernel_t g_build_kernels[2][4] = {JV_ROW(jv_build_search_kernel, false),
                                                   JV_ROW(jv_build_search_kernel, true)};
#define JV_PQF_ROW(CH, NP, FAST) \
    { jv_search_pqf_kernel<0, CH, NP, FAST>, jv_search_pqf_kernel<2, CH, NP, FAST>, jv_search_pqf_kernel<12, CH, NP, FAST>, jv_search_pqf_kernel<24, CH, NP, FAST> }
// [0 single-pass | 1 multi-pass | 2 single-pass FAST][pool size][nch slot]
static const lds_kernel_t g_pqf_kernels[3][3][4] = {{JV_PQF_ROW(4, 1, false), JV_PQF_ROW(8, 1, false), JV_PQF_ROW(16, 1, false)},
                                                    {JV_PQF_ROW(4, 4, false), JV_PQF_ROW(8, 4, false), JV_PQF_ROW(16, 4, false)},
                                                    {JV_PQF_ROW(4, 1, true), JV_PQF_ROW(8, 1, true), JV_PQF_ROW(16, 1, true)}};
#define JV_PQFF_ROW(CH, FAST) \
    { jv_search_pqf_filtered_kernel<0, CH, FAST>, jv_search_pqf_filtered_kernel<2, CH, FAST>, jv_search_pqf_filtered_kernel<12, CH, FAST>, jv_search_pqf_filtered_kernel<24, CH, FAST> }
// [FAST][pool size][nch slot]
static const lds_kernel_t g_pqff_kernels[2][2][4] = {{JV_PQFF_ROW(8, false), JV_PQFF_ROW(16, false)}, {JV_PQFF_ROW(8, true), JV_PQFF_ROW(16, true)}};
static const big_kernel_t g_big_kernels[2][4] = {JV_ROW(jv_search_big_kernel, false), JV_ROW(jv_search_big_kernel, true)};
#define JV_BIGQ_ROW(PQ) { jv_search_big_kernel<PQ, 0, true>, jv_search_big_kernel<PQ, 2, true>, jv_search_big_kernel<PQ, 12, true>, jv_search_big_kernel<PQ, 24, true> }
static const big_kernel_t g_bigq_kernels[2][4] = {JV_BIGQ_ROW(false), JV_BIGQ_ROW(true)};  // queues in LDS, visited bitset in HBM
// PQ look-up table in HBM scratch (queues and visited bitset in HBM too): pq_M beyond what LDS holds
static const big_kernel_t g_bigg_kernels[2][4] = {{jv_search_big_kernel<true, 0, false, true>, jv_search_big_kernel<true, 2, false, true>,
                                                   jv_search_big_kernel<true, 12, false, true>, jv_search_big_kernel<true, 24, false, true>},
                                                  {jv_search_big_kernel<true, 0, true, true>, jv_search_big_kernel<true, 2, true, true>,
                                                   jv_search_big_kernel<true, 12, true, true>, jv_search_big_kernel<true, 24, true, true>}};  // [queues in LDS]

static int nch_slot(const JvIndexDev* ix) {
    if (ix->nvq_M > 0) return 0;  // the NVQ decoder lives in the "any d" instances only (score_rows)
    if (ix->stride != ix->nch * 64) return 0;
    return ix->nch == 2 ? 1 : ix->nch == 12 ? 2 : ix->nch == 24 ? 3 : 0;
}

extern "C" hipError_t jvk_set_max_lds(int bytes) {
    for (int s = 0; s < 4; s++) {
        for (int a = 0; a < 2; a++) {
            for (int b = 0; b < 2; b++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_lds_kernels[a][b][s],
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute((const void*)g_retry_kernels[a][b][s],
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
            hipError_t e = hipFuncSetAttribute((const void*)g_build_kernels[a][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess && a == 0)
                for (int v = 0; v < 9 && e == hipSuccess; v++)
                    e = hipFuncSetAttribute((const void*)g_pqf_kernels[v / 3][v % 3][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess && a == 0)
                for (int v = 0; v < 4 && e == hipSuccess; v++)
                    e = hipFuncSetAttribute((const void*)g_pqff_kernels[v >> 1][v & 1][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)g_big_kernels[a][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)g_bigq_kernels[a][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)g_bigg_kernels[a][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

extern "C" hipError_t jvk_launch_search_pqf(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    const int multi = ix->R * ix->pq_lanes > JV_WAVE ? 1 : 0;
    const int fast = (ix->pq_M % 16 == 0 && ix->sim != 2) ? 1 : 0;
    if (a->accept) {  // filtered variant (the host only selects it for single-pass blocks)
        int grid = a->nq;
        if (a->retry_only) {
            grid = (a->nq + 7) / 8;
            const int resident = 256 * (lds_bytes > 0 ? (163840 / lds_bytes > 0 ? 163840 / lds_bytes : 1) : 8);
            if (grid > resident) grid = resident;
        }
        g_pqff_kernels[fast][a->cand_cap > 512 ? 1 : 0][nch_slot(ix)]<<<grid, JV_WAVE, lds_bytes, stream>>>(*ix, *a);
        return hipGetLastError();
    }
    g_pqf_kernels[multi ? 1 : (fast ? 2 : 0)][a->cand_cap > 512 ? 2 : a->cand_cap > 256 ? 1 : 0][nch_slot(ix)]<<<a->nq, JV_WAVE, lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}

// tag: 0 = query launch, 1 = escalation launch, 2 = graph-builder client
extern "C" hipError_t jvk_launch_search_lds(const JvIndexDev* ix, const JvSearchArgs* a, int pq, int pool, int tag,
                                            int lds_bytes, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    lds_kernel_t k;
    int grid = a->nq;
    if (tag == 1) {
        k = g_retry_kernels[pq ? 1 : 0][pool ? 1 : 0][nch_slot(ix)];
        grid = (a->nq + JV_WAVE - 1) / JV_WAVE;  // at most one block per 64 queries; blocks dequeue flag chunks
        if (grid > 2048) grid = 2048;
    } else if (tag == 2 && !pq) k = g_build_kernels[pool ? 1 : 0][nch_slot(ix)];
    else k = g_lds_kernels[pq ? 1 : 0][pool ? 1 : 0][nch_slot(ix)];
    k<<<grid, JV_WAVE, lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}

// qlds: 1 = queues in LDS (a->res_cap / a->cand_cap entries after the fixed part), visited bitset in HBM
extern "C" hipError_t jvk_launch_search_big(const JvIndexDev* ix, const JvSearchArgs* a, int pq, int blocks,
                                            int lds_bytes, int force_all, int qlds, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    (qlds ? g_bigq_kernels : g_big_kernels)[pq ? 1 : 0][nch_slot(ix)]<<<blocks, JV_WAVE, lds_bytes, stream>>>(*ix, *a, force_all);
    return hipGetLastError();
}
// stream marker: one store to host-visible memory.  The host polls the word instead of asking the runtime (hipEventQuery /
// hipStreamQuery never turn ready while a resident query-server grid runs, and waits for a result copy were seen to return
// only after the grid left: DESIGN.md "Resident grids and the rest of the runtime")
__global__ void jv_mark_kernel(int32_t* word, int32_t value) {
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" hipError_t jvk_launch_mark(int32_t* word, int32_t value, hipStream_t stream) {
    jv_mark_kernel<<<1, 1, 0, stream>>>(word, value);
    return hipGetLastError();
}

extern "C" hipError_t jvk_launch_search_big_lutg(const JvIndexDev* ix, const JvSearchArgs* a, int blocks, int lds_bytes, int force_all,
                                                 int qlds, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    g_bigg_kernels[qlds ? 1 : 0][nch_slot(ix)]<<<blocks, JV_WAVE, lds_bytes, stream>>>(*ix, *a, force_all);
    return hipGetLastError();
}

extern "C" hipError_t jvk_launch_score_ordinals(const JvIndexDev* ix, const float* d_query,
                                                const int32_t* d_ordinals, int count, float* d_out,
                                                hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    int blocks = (count + JV_WAVE - 1) / JV_WAVE;
    if (blocks > 2048) blocks = 2048;
    int lds = ix->nch * 64 * 4 + JV_TODO * 8;
    jv_score_ordinals_kernel<<<blocks, JV_WAVE, lds, stream>>>(*ix, d_query, d_ordinals, count, d_out);
    return hipGetLastError();
}

extern "C" hipError_t jvk_launch_merge_topk(const int32_t* d_docs, const float* d_scores, int nq, int lists,
                                            int k, int32_t* d_out_docs, float* d_out_scores,
                                            hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    int lds = lists * k * 8;
    jv_merge_topk_kernel<false><<<nq, JV_WAVE, lds, stream>>>(d_docs, d_scores, nq, lists, k, d_out_docs, d_out_scores);
    return hipGetLastError();
}

// (doc, score) -> one 8-byte pair per result, so that a shard's top-k lists travel in ONE peer copy
__global__ void jv_pack_pairs_kernel(const int32_t* docs, const float* scores, int32_t* pairs, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        pairs[2 * i] = docs[i];
        pairs[2 * i + 1] = __float_as_int(scores[i]);
    }
}
extern "C" hipError_t jvk_launch_pack_pairs(const int32_t* d_docs, const float* d_scores, int32_t* d_pairs, long long n,
                                            hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    jv_pack_pairs_kernel<<<blocks, 256, 0, stream>>>(d_docs, d_scores, d_pairs, n);
    return hipGetLastError();
}

extern "C" hipError_t jvk_launch_merge_topk_strided(const int32_t* d_docs, const float* d_scores, int nq, int lists,
                                                    int k, int32_t* d_out_docs, float* d_out_scores,
                                                    hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    int lds = lists * k * 8;
    jv_merge_topk_kernel<true><<<nq, JV_WAVE, lds, stream>>>(d_docs, d_scores, nq, lists, k, d_out_docs, d_out_scores);
    return hipGetLastError();
}
