// jv_dev_common.h — device helpers shared by the kernel translation units (jv_kernels.hip, jv_kernels_pqp.hip):
// NodeQueue keys, cross-lane helpers, the canonical exact / PQ scoring, visited sets, the threshold tracker.
// gfx950 only.  Compile with -ffp-contract=off: every fused multiply-add is an explicit fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jv_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DJV_STAMPS, lib/libjvgpu_stamps.so): per-phase cycle shares of the pool loop are
// accumulated into a.dbg; never compiled into the product library and never read by the kernel itself.
#ifdef JV_STAMPS
#define STAMP_DECL unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_last = clock64();
#define STAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = clock64(); st_acc[i] += t_ - st_last; st_last = t_; __builtin_amdgcn_sched_barrier(0); }
#define STAMP_FLUSH if (a.dbg && lane == 0) { for (int i_ = 0; i_ < 16; i_++) atomicAdd((unsigned long long*)a.dbg + i_, st_acc[i_]); }
#define STAMP_COUNT(i, v) { st_acc[i] += (v); }
#define STAMP_DIRECT(i) { unsigned long long t_ = clock64(); if (a.dbg && lane == 0) atomicAdd((unsigned long long*)a.dbg + (i), t_ - st_last); st_last = t_; }
#else
#define STAMP_DECL
#define STAMP(i) {}
#define STAMP_FLUSH {}
#define STAMP_COUNT(i, v) {}
#define STAMP_DIRECT(i) {}
#endif

// loads of data a launch touches ONCE (full-precision rows of the rerank, fused blocks): non-temporal, so that they do not evict
// what every query re-reads (the PQ codebook of the table builds: 786 KB per query from L2 instead of HBM)
#ifndef JV_NO_STREAM_LOADS
#define JV_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define JV_STREAM_LOAD(p) (*(p))
#endif
#define KEY_MIN ((int64_t)0x8000000000000000ll)
#define KEY_MAX ((int64_t)0x7fffffffffffffffll)
#define HASH_EMPTY 0xFFFFFFFFu
#ifndef JV_PQF_RERANK_UMUL
#define JV_PQF_RERANK_UMUL 2  // rows in flight per rerank step = 4 * RowsInFlight * this
#endif

// ---------------------------------------------------------------------------------------------
// NodeQueue keys (jvector NodeQueue.encode)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t make_key(float score, int node) {
    int32_t b = __float_as_int(score);
    int32_t s = b ^ ((b >> 31) & 0x7fffffff);
    return (int64_t)(((uint64_t)(uint32_t)s << 32) | (uint64_t)(uint32_t)(~node));
}
__device__ __forceinline__ float key_score(int64_t k) {
    int32_t s = (int32_t)(k >> 32);
    return __int_as_float(s ^ ((s >> 31) & 0x7fffffff));
}
__device__ __forceinline__ int key_node(int64_t k) { return ~(int32_t)(uint32_t)(k & 0xFFFFFFFFll); }

// ---------------------------------------------------------------------------------------------
// cross-lane helpers
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// adjacent-pair tree over the 16 lanes of a DPP row: lane^1, lane^2, then the sibling quads, then the
// sibling octets (mirrors deliver the sibling group's identical partial; fp add is commutative).
__device__ __forceinline__ float row16_tree_sum(float v) {
    v = v + dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v);  // row_half_mirror
    v = v + dpp_mov<0x140>(v);  // row_mirror
    return v;
}

// sum of an int over the 64 lanes, wave-uniform result: four DPP steps inside each row of 16, then the four row sums by readlane
__device__ __forceinline__ int jv_wave_sum_int(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ void wave_argmax(int64_t& k, int& idx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int64_t ok = __shfl_xor(k, off, 64);
        int oi = __shfl_xor(idx, off, 64);
        if (ok > k) {
            k = ok;
            idx = oi;
        }
    }
}
__device__ __forceinline__ void wave_argmin(int64_t& k, int& idx) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        int64_t ok = __shfl_xor(k, off, 64);
        int oi = __shfl_xor(idx, off, 64);
        if (ok < k) {
            k = ok;
            idx = oi;
        }
    }
}
__device__ __forceinline__ void scan_max(const int64_t* arr, int n, int lane, int64_t& best, int& bi) {
    best = KEY_MIN;
    bi = -1;
    for (int i = lane; i < n; i += JV_WAVE) {
        int64_t k = arr[i];
        if (k > best) {
            best = k;
            bi = i;
        }
    }
    wave_argmax(best, bi);
}
__device__ __forceinline__ void scan_min(const int64_t* arr, int n, int lane, int64_t& best, int& bi) {
    best = KEY_MAX;
    bi = -1;
    for (int i = lane; i < n; i += JV_WAVE) {
        int64_t k = arr[i];
        if (k < best) {
            best = k;
            bi = i;
        }
    }
    wave_argmin(best, bi);
}

__device__ __forceinline__ float map_score(int sim, float raw) {
    if (sim == 0) return 1.0f / (1.0f + raw);
    return (1.0f + raw) / 2.0f;
}

// ---------------------------------------------------------------------------------------------
// Exact scoring of up to 64 rows: the canonical accumulation.
// Lane t of a 16-lane group owns elements 64j + 4t .. 4t+3 of its row (one 16-B load per chunk j):
// 4 accumulators per lane = the 64 strided partials P[m], m = 4t + e; then (P0+P1)+(P2+P3) in the
// lane and the adjacent-pair tree across the 16 lanes — the same tree as oracle/jv_oracle.c tree64.
// SIM: 0 L2, 1 dot, 2 cosine.  Follows jvector VectorSimilarityFunction.compare.
// ---------------------------------------------------------------------------------------------
template <int SIM>
__device__ __forceinline__ void score_rows_t(const JvIndexDev& ix, const float* q_lds, const int32_t* todo,
                                             int m, float* todo_score, float qnorm2, float scale, int lane) {
    constexpr int U = 4;   // row-groups in flight: 4 x 4 = 16 rows
    constexpr int JU = 4;  // chunks in flight per row
    const int g = lane >> 4, t = lane & 15;
    const int nch = ix.nch, stride = ix.stride;
    for (int base = 0; base < m; base += 4 * U) {
        float acc[U][4], nrm[U][4];
        const float* rp[U];
        bool val[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int r = base + 4 * u + g;
            val[u] = r < m;
            int node = todo[val[u] ? r : 0];
            rp[u] = ix.vectors + (size_t)node * (size_t)stride + 4 * t;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc[u][e] = 0.0f;
                nrm[u][e] = 0.0f;
            }
        }
        for (int j0 = 0; j0 < nch; j0 += JU) {
            f32x4 v[U][JU];
#pragma unroll
            for (int jj = 0; jj < JU; jj++) {
                const int j = j0 + jj;
                const bool okj = j < nch && (j * 64 + 4 * t) < stride;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (okj && val[u]) v[u][jj] = *(const f32x4*)(rp[u] + j * 64);
                    else v[u][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int jj = 0; jj < JU; jj++) {
                const int j = j0 + jj;
                if (j < nch && (j * 64 + 4 * t) < stride) {
                    const f32x4 qv = *(const f32x4*)(q_lds + j * 64 + 4 * t);
#pragma unroll
                    for (int u = 0; u < U; u++) {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (SIM == 0) {
                                float df = qv[e] - v[u][jj][e];
                                acc[u][e] = fmaf(df, df, acc[u][e]);
                            } else {
                                acc[u][e] = fmaf(qv[e], v[u][jj][e], acc[u][e]);
                                if (SIM == 2) nrm[u][e] = fmaf(v[u][jj][e], v[u][jj][e], nrm[u][e]);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float s = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]);
            s = row16_tree_sum(s);
            float score;
            if (SIM == 2) {
                float nv = (nrm[u][0] + nrm[u][1]) + (nrm[u][2] + nrm[u][3]);
                nv = row16_tree_sum(nv);
                score = map_score(2, s / sqrtf(qnorm2 * nv));
            } else {
                score = map_score(SIM, s);
            }
            if (scale != 1.0f) score = score * scale;
            if (t == 0 && val[u]) todo_score[base + 4 * u + g] = score;
        }
    }
}

// Same arithmetic, specialised for rows of exactly NCH 64-float chunks: every 16-B load of U row-groups
// (4U rows) is issued before the first fma, so one pass costs one HBM round trip.
// STREAM: the rows are read once per launch (the rerank of a PQ search over a corpus far beyond the caches): non-temporal loads.
// The exact kernels re-read hot rows from L2 / MALL (C2: every row ~90 times per launch) and keep the default policy.
// the lane id, computed where it is used: a volatile asm cannot be hoisted, so nothing derived from it lives across a kernel's
// register peaks (a spilled lane constant reloaded between two row groups is a vmcnt(0): it waits for the first group's loads)
__device__ __forceinline__ int jv_lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
template <int SIM, int NCH, int U, bool FULL, bool STREAM = false>
__device__ __forceinline__ void score_rows_fixed(const JvIndexDev& ix, const float* q_lds, const int32_t* todo,
                                                 int m, float* todo_score, float qnorm2, float scale, int lane) {
    const int stride = ix.stride;  // FULL: stride == NCH * 64, no partial chunk
    for (int base = 0; base < m; base += 4 * U) {
        const int ln = jv_lane_now();
        const int g = ln >> 4, t = ln & 15;
        f32x4 v[U][NCH];
        bool val[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = base + 4 * u + g;
            val[u] = r < m;
            if (val[u]) {
                const float* rp = ix.vectors + (size_t)todo[r] * (size_t)stride + 4 * t;
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    if (FULL || (j * 64 + 4 * t) < stride) v[u][j] = STREAM ? JV_STREAM_LOAD((const f32x4*)(rp + j * 64)) : *(const f32x4*)(rp + j * 64);
                    else v[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            } else {
#pragma unroll
                for (int j = 0; j < NCH; j++) v[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        float acc[U][4], nrm[U][4];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[u][e] = 0.0f, nrm[u][e] = 0.0f;
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            if (FULL || (j * 64 + 4 * t) < stride) {
                const f32x4 qv = *(const f32x4*)(q_lds + j * 64 + 4 * t);
#pragma unroll
                for (int u = 0; u < U; u++) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (SIM == 0) {
                            const float df = qv[e] - v[u][j][e];
                            acc[u][e] = fmaf(df, df, acc[u][e]);
                        } else {
                            acc[u][e] = fmaf(qv[e], v[u][j][e], acc[u][e]);
                            if (SIM == 2) nrm[u][e] = fmaf(v[u][j][e], v[u][j][e], nrm[u][e]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float s = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]);
            s = row16_tree_sum(s);
            float score;
            if (SIM == 2) {
                float nv = (nrm[u][0] + nrm[u][1]) + (nrm[u][2] + nrm[u][3]);
                nv = row16_tree_sum(nv);
                score = map_score(2, s / sqrtf(qnorm2 * nv));
            } else {
                score = map_score(SIM, s);
            }
            if (scale != 1.0f) score = score * scale;
            if (t == 0 && val[u]) todo_score[base + 4 * u + g] = score;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// NVQ-inline vectors: the same canonical accumulation over the DEQUANTISED record (nvqDequantize,
// J/JVectorIndexQuantization.java:319-361; identical arithmetic to oracle/jv_oracle.c jvo_nvq_dequantize).
// A 16-lane group scores one row: lane t owns elements 64 j + 4 t .. + 3 (one dword of bytes per chunk).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int java_round_f(float x) {  // Math.round(float): closest int, ties towards +infinity
    const float r = floorf(x);
    return (int)r + ((x - r) >= 0.5f ? 1 : 0);
}
__device__ __forceinline__ float nvq_logistic(float value, float alpha, float x0) {
    float temp = fmaf(value, alpha, -alpha * x0);
    const int p = java_round_f(temp + 0.5f);
    const float f = fmaf(temp - (float)p, 0.5f, 1.0f);
    temp = __int_as_float((int)((uint32_t)__float_as_int(f) + ((uint32_t)p << 23)));
    return temp / (temp + 1.0f);
}
__device__ __forceinline__ float nvq_logit(float scaled, float inverse_alpha, float x0) {
    const float z = scaled / (1.0f - scaled);
    const int temp = __float_as_int(z);
    const int e = temp & 0x7f800000;
    const float p = (float)((e >> 23) - 128);
    const float m = __int_as_float((temp & 0x007fffff) + 0x3f800000);
    return (m + p) * inverse_alpha + x0;
}
template <int SIM>
__device__ __forceinline__ void score_rows_nvq_t(const JvIndexDev& ix, const float* q_lds, const int32_t* todo, int m,
                                                 float* todo_score, float qnorm2, float scale, int lane) {
    const int g = lane >> 4, t = lane & 15;
    const int nch = ix.nch, NM = ix.nvq_M;
    for (int base = 0; base < m; base += 4) {
        const int r = base + g;
        const bool val = r < m;
        const int node = todo[val ? r : 0];
        // per-subvector decode constants of this row
        float sc_[JV_NVQ_MAX_M], bias_[JV_NVQ_MAX_M], inv_[JV_NVQ_MAX_M], x0_[JV_NVQ_MAX_M];
#pragma unroll
        for (int s = 0; s < JV_NVQ_MAX_M; s++) {
            sc_[s] = bias_[s] = inv_[s] = x0_[s] = 0.0f;
            if (s < NM) {
                const float* pr = ix.nvq_params + ((size_t)node * NM + s) * 4;
                const float growth = pr[0], midpoint = pr[1], minv = pr[2], maxv = pr[3];
                const float delta = maxv - minv;
                const float sg = growth / delta;
                const float sm = midpoint * delta;
                bias_[s] = nvq_logistic(minv, sg, sm);
                sc_[s] = (nvq_logistic(maxv, sg, sm) - bias_[s]) / 255.0f;
                inv_[s] = 1.0f / sg;
                x0_[s] = sm;
            }
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, nrm[4] = {0.f, 0.f, 0.f, 0.f};
        const uint8_t* rowp = ix.nvq_bytes + (size_t)node * ix.nvq_stride;
        for (int j = 0; j < nch; j++) {
            const int i0 = j * 64 + 4 * t;
            if (i0 < ix.stride) {
                const uint32_t w = *(const uint32_t*)(rowp + i0);
                const f32x4 qv = *(const f32x4*)(q_lds + i0);
                const f32x4 mean = *(const f32x4*)(ix.nvq_mean + i0);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = i0 + e;
                    float v = 0.0f;
                    if (i < ix.d) {
                        int s = 0;
#pragma unroll
                        for (int s2 = 1; s2 < JV_NVQ_MAX_M; s2++) s += (s2 < NM && i >= ix.nvq_sub_off[s2]) ? 1 : 0;
                        float scs = sc_[0], bs = bias_[0], ivs = inv_[0], x0s = x0_[0];
#pragma unroll
                        for (int s2 = 1; s2 < JV_NVQ_MAX_M; s2++)
                            if (s == s2) scs = sc_[s2], bs = bias_[s2], ivs = inv_[s2], x0s = x0_[s2];
                        const float b = (float)((w >> (8 * e)) & 0xFFu);
                        v = nvq_logit(fmaf(b, scs, bs), ivs, x0s) + mean[e];
                    }
                    if (SIM == 0) {
                        const float df = qv[e] - v;
                        acc[e] = fmaf(df, df, acc[e]);
                    } else {
                        acc[e] = fmaf(qv[e], v, acc[e]);
                        if (SIM == 2) nrm[e] = fmaf(v, v, nrm[e]);
                    }
                }
            }
        }
        float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        s = row16_tree_sum(s);
        float score;
        if (SIM == 2) {
            float nv = (nrm[0] + nrm[1]) + (nrm[2] + nrm[3]);
            nv = row16_tree_sum(nv);
            score = map_score(2, s / sqrtf(qnorm2 * nv));
        } else {
            score = map_score(SIM, s);
        }
        if (scale != 1.0f) score = score * scale;
        if (t == 0 && val) todo_score[r] = score;
    }
}

// NCHT = number of 64-float chunks per row known at compile time (kernel template parameter; 0 = any d)
template <int NCHT>
struct RowsInFlight { static constexpr int U = NCHT <= 6 ? 4 : (NCHT <= 12 ? 2 : 1); };

template <int NCHT, int UMUL = 1, bool STREAM = false, int UADD = 0>   // UADD: more row groups in flight where the caller has the registers
__device__ __forceinline__ void score_rows(const JvIndexDev& ix, const float* q_lds, const int32_t* todo, int m,
                                           float* todo_score, float qnorm2, float scale, int lane) {
    // NVQ-inline field: exact scores against the dequantised records.  Only the "any d" instances (NCHT = 0) carry the
    // decoder — the launchers route NVQ indexes to them — so that the fixed-d kernels keep their register budget
    // (with the decoder inlined the C2 kernel went from 168 to 248 VGPRs + scratch and lost a third of its throughput).
    if (NCHT == 0 && ix.nvq_M > 0) {
        if (ix.sim == 0) score_rows_nvq_t<0>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
        else if (ix.sim == 1) score_rows_nvq_t<1>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
        else score_rows_nvq_t<2>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
        return;
    }
    if (NCHT == 0) {
        if (ix.sim == 0) score_rows_t<0>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
        else if (ix.sim == 1) score_rows_t<1>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
        else score_rows_t<2>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
    } else {
        constexpr int N = NCHT == 0 ? 1 : NCHT;
        constexpr int U = RowsInFlight<N>::U * UMUL + UADD;
        if (ix.sim == 0) score_rows_fixed<0, N, U, true, STREAM>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
        else if (ix.sim == 1) score_rows_fixed<1, N, U, true, STREAM>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
        else score_rows_fixed<2, N, U, true, STREAM>(ix, q_lds, todo, m, todo_score, qnorm2, scale, lane);
    }
}

// canonical dot(q,q) for cosine: the row is the query itself (in LDS)
__device__ __forceinline__ float query_norm2(const JvIndexDev& ix, const float* q_lds, int lane) {
    const int t = lane & 15;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < ix.nch; j++) {
        if (j * 64 + 4 * t < ix.stride) {
            const f32x4 qv = *(const f32x4*)(q_lds + j * 64 + 4 * t);
#pragma unroll
            for (int e = 0; e < 4; e++) a[e] = fmaf(qv[e], qv[e], a[e]);
        }
    }
    float s = (a[0] + a[1]) + (a[2] + a[3]);
    return row16_tree_sum(s);
}

// ---------------------------------------------------------------------------------------------
// PQ: per-query look-up table in LDS, ADC scoring (jvector PQVectors.precomputedScoreFunctionFor /
// PQDecoder).  lut[m][c] is a sequential fmaf chain over the subspace.
// ---------------------------------------------------------------------------------------------
template <int PF>  // PF = codebook rows in flight per lane (each a 1 KiB wave-wide read of the transposed codebook)
__device__ __forceinline__ void build_lut(const JvIndexDev& ix, const float* qc_lds, float* lut, int lane) {
    const int M = ix.pq_M;
    const bool l2 = ix.sim == 0;
    for (int m = 0; m < M; m++) {
        const int d0 = ix.pq_sub_off[m], d1 = ix.pq_sub_off[m + 1];
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int db = d0; db < d1; db += PF) {
            f32x4 cb[PF];
#pragma unroll
            for (int u = 0; u < PF; u++) {
                if (db + u < d1) cb[u] = *(const f32x4*)(ix.pq_cbT + (size_t)(db + u) * 256 + 4 * lane);
                else cb[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < PF; u++) {  // the fmaf chain stays in dimension order (canonical)
                if (db + u < d1) {
                    const float qc = qc_lds[db + u];  // q - globalCentroid (or q when there is no centroid)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (l2) {
                            float df = qc - cb[u][e];
                            a[e] = fmaf(df, df, a[e]);
                        } else {
                            a[e] = fmaf(qc, cb[u][e], a[e]);
                        }
                    }
                }
            }
        }
        *(f32x4*)(lut + m * 256 + 4 * lane) = (f32x4){a[0], a[1], a[2], a[3]};
    }
}

// tree over the `lpn` (power of two) adjacent lanes that share one node
__device__ __forceinline__ float lanes_tree_sum(float v, int lpn) {
    if (lpn >= 2) v = v + dpp_mov<0xB1>(v);
    if (lpn >= 4) v = v + dpp_mov<0x4E>(v);
    if (lpn >= 8) v = v + dpp_mov<0x141>(v);
    if (lpn >= 16) v = v + dpp_mov<0x140>(v);
    if (lpn >= 32) v = v + __shfl_xor(v, 16, 64);
    if (lpn >= 64) v = v + __shfl_xor(v, 32, 64);
    return v;
}

// 16 consecutive subspaces per lane, summed left to right; chunk sums combined by the lane tree.
// All 16 table reads are issued before the first add (independent ds_read_b32, one LDS round trip);
// slots beyond M contribute +0.0f, which leaves the sum bit-identical.
template <bool FULL = false>  // FULL: all 16 subspaces of the chunk exist (M % 16 == 0): no per-slot masking
__device__ __forceinline__ float adc_chunk(const float* lut, const u32x4 cw, int m0, int M) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int mi = m0 + i;
        const uint32_t code = (cw[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
        if (FULL) {
            v[i] = lut[mi * 256 + code];
        } else {
            const int mc = mi < M ? mi : m0;  // clamp the address, mask the value
            const float t = lut[mc * 256 + code];
            v[i] = mi < M ? t : 0.0f;
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) s = s + v[i];
    return s;
}

__device__ __forceinline__ void score_nodes_pq(const JvIndexDev& ix, const float* lut, const int32_t* todo, int m,
                                               float* todo_score, float qnorm2, int lane) {
    const int lpn = ix.pq_lanes;
    const int npp = JV_WAVE / lpn;  // nodes per pass
    const int c = lane & (lpn - 1);
    const int M = ix.pq_M, cs = ix.pq_code_stride;
    for (int base = 0; base < m; base += npp) {
        const int r = base + lane / lpn;
        const bool val = r < m;
        const int node = todo[val ? r : 0];
        u32x4 cw = (u32x4){0, 0, 0, 0};
        const bool have = val && c * 16 < M;
        if (have) cw = *(const u32x4*)(ix.pq_codes + (size_t)node * cs + c * 16);
        float s = have ? adc_chunk(lut, cw, c * 16, M) : 0.0f;
        s = lanes_tree_sum(s, lpn);
        float score;
        if (ix.sim == 2) {
            float na = have ? adc_chunk(ix.pq_norm_lut, cw, c * 16, M) : 0.0f;
            na = lanes_tree_sum(na, lpn);
            score = map_score(2, s / sqrtf(qnorm2 * na));
        } else {
            score = map_score(ix.sim, s);
        }
        if (c == 0 && val) todo_score[r] = score;
    }
}

// ---------------------------------------------------------------------------------------------
// visited set
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool visited_insert_lds(uint32_t* tab, uint32_t mask, int shift, uint32_t node) {
    uint32_t h = (node * 0x9E3779B1u) >> shift;
    for (;;) {
        uint32_t old = atomicCAS(&tab[h], HASH_EMPTY, node);
        if (old == HASH_EMPTY) return true;
        if (old == node) return false;
        h = (h + 1) & mask;
    }
}
// Two-level visited set: when the LDS table reaches its fill limit it is frozen (lookups only) and further
// nodes go to a per-query spill table in HBM (L2-resident), taken from a per-launch pool.  Exact in all
// cases; a query that also exhausts its spill table (or finds the pool empty) is flagged for the retry path.
struct Visited {
    uint32_t* lds;
    uint32_t lmask;
    int lshift;
    uint32_t* spill;   // nullptr until the LDS table is frozen
    uint32_t smask;
    int sshift;
    int nspill;
};
__device__ __forceinline__ bool visited_insert2(Visited& vs, uint32_t node) {
    if (vs.spill == nullptr) return visited_insert_lds(vs.lds, vs.lmask, vs.lshift, node);
    uint32_t h = (node * 0x9E3779B1u) >> vs.lshift;
    for (;;) {
        const uint32_t v = vs.lds[h];
        if (v == node) return false;
        if (v == HASH_EMPTY) break;
        h = (h + 1) & vs.lmask;
    }
    uint32_t g = (node * 0x85EBCA6Bu) >> vs.sshift;
    for (;;) {
        const uint32_t old = atomicCAS(&vs.spill[g], HASH_EMPTY, node);
        if (old == HASH_EMPTY) return true;
        if (old == node) return false;
        g = (g + 1) & vs.smask;
    }
}

__device__ __forceinline__ bool visited_insert_bits(uint32_t* bits, uint32_t node) {
    const uint32_t bit = 1u << (node & 31);
    uint32_t old = atomicOr(&bits[node >> 5], bit);
    return (old & bit) == 0;
}

// ---------------------------------------------------------------------------------------------
// jvector ScoreTracker.TwoPhaseTracker (threshold queries only; SURVEY App. A.2 "shouldStop"):
// window of the 500 most recent scores + the 100 best scores; evaluated when the observation count is
// a multiple of 100 (>= 500): stop iff percentile99(window) < threshold && worst-of-best < threshold.
// percentile = commons-math3 LEGACY estimate: pos = 99*(500+1)/100, lower + (pos-floor(pos))*(upper-lower).
// ---------------------------------------------------------------------------------------------
#define TRK_RECENT 500
#define TRK_BEST 100
struct Tracker {
    float* recent;  // [TRK_RECENT]
    float* best;    // [TRK_BEST + JV_WAVE] scratch for the rank merge
    float* best2;   // [TRK_BEST + JV_WAVE]
    int idx, obs, nbest;
};

// track m scores (todo_score[0..m), stored order)
__device__ __forceinline__ void tracker_track(Tracker& t, const float* scores, int m, int lane) {
    if (lane < m) t.recent[(t.idx + lane) % TRK_RECENT] = scores[lane];
    // best := top-TRK_BEST multiset of (best U new) by rank counting
    const int total = t.nbest + m;
    if (lane < m) t.best[t.nbest + lane] = scores[lane];
    __syncthreads();
    for (int i = lane; i < total; i += JV_WAVE) {
        const float v = t.best[i];
        int r = 0;
        for (int j = 0; j < total; j++) {
            const float w = t.best[j];
            r += (w > v || (w == v && j < i)) ? 1 : 0;
        }
        if (r < TRK_BEST) t.best2[r] = v;
    }
    __syncthreads();
    float* tmp = t.best;
    t.best = t.best2;
    t.best2 = tmp;
    t.nbest = total < TRK_BEST ? total : TRK_BEST;
    t.idx = (t.idx + m) % TRK_RECENT;
    t.obs += m;
}

__device__ __forceinline__ bool tracker_should_stop(const Tracker& t, float threshold, int lane) {
    if (t.obs < TRK_RECENT) return false;
    if (t.obs % 100 != 0) return false;
    // 5th and 6th largest of the window = sorted[495], sorted[494]
    float upper = 0.0f, lower = 0.0f;
    for (int i0 = 0; i0 < TRK_RECENT; i0 += JV_WAVE) {
        const int i = i0 + lane;
        int r = -1;
        float v = 0.0f;
        if (i < TRK_RECENT) {
            v = t.recent[i];
            r = 0;
            for (int j = 0; j < TRK_RECENT; j++) {
                const float w = t.recent[j];
                r += (w > v || (w == v && j < i)) ? 1 : 0;
            }
        }
        const unsigned long long m4 = __ballot(r == 4), m5 = __ballot(r == 5);
        if (m4) upper = __shfl(v, __ffsll((long long)m4) - 1, JV_WAVE);
        if (m5) lower = __shfl(v, __ffsll((long long)m5) - 1, JV_WAVE);
    }
    const double pos = 99.0 * (double)(TRK_RECENT + 1) / 100.0;
    const double dd = pos - floor(pos);
    const double pct = (double)lower + dd * ((double)upper - (double)lower);
    // best is sorted descending by construction (rank order): its last element is the worst of the best
    const double worst_best = (double)t.best[t.nbest - 1];
    return pct < (double)threshold && worst_best < (double)threshold;
}

// ---------------------------------------------------------------------------------------------
// the search
// ---------------------------------------------------------------------------------------------
struct QState {
    int ncand;       // live candidates: cand[0..ncand)
    int nhand;       // upper-layer hand-back entries: cand[cap-1-i]
    int nres;        // results: res[0..nres)
    int64_t worst;   // min key of res (valid when nres == rk_cur)
    int worst_idx;
    int visited, expanded, expanded_base, reranked;
    bool overflow;
};

// pool keys: NodeQueue key with the constant bit 31 of the low word dropped and bit 0 = "not yet expanded";
// the order between two different nodes is unchanged (score desc, ordinal asc)
__device__ __forceinline__ int64_t make_pool_key(float score, int node) {
    int32_t b = __float_as_int(score);
    int32_t s = b ^ ((b >> 31) & 0x7fffffff);
    return (int64_t)(((uint64_t)(uint32_t)s << 32) | ((uint64_t)((uint32_t)(~node) & 0x7FFFFFFFu) << 1) | 1ull);
}
__device__ __forceinline__ int64_t key_to_pool(int64_t k) {
    return (int64_t)(((uint64_t)k & 0xFFFFFFFF00000000ull) | (((uint64_t)k & 0x7FFFFFFFull) << 1) | 1ull);
}
__device__ __forceinline__ int64_t pool_to_key(int64_t pk) {
    return (int64_t)(((uint64_t)pk & 0xFFFFFFFF00000000ull) | 0x80000000ull | (((uint64_t)pk >> 1) & 0x7FFFFFFFull));
}
__device__ __forceinline__ int pool_node(int64_t pk) { return key_node(pool_to_key(pk)); }
// filtered pool keys (n < 2^30): one more constant bit of ~node is dropped to make room for bit 1 = "accepted by
// the query's doc filter" (a function of the node, so equal nodes still have equal keys up to bit 0)
__device__ __forceinline__ int64_t make_pool_key_f(float score, int node, bool acc) {
    int32_t b = __float_as_int(score);
    int32_t s = b ^ ((b >> 31) & 0x7fffffff);
    return (int64_t)(((uint64_t)(uint32_t)s << 32) | ((uint64_t)((uint32_t)(~node) & 0x3FFFFFFFu) << 2) | (acc ? 2ull : 0ull) | 1ull);
}
__device__ __forceinline__ int pool_node_f(int64_t pk) { return (int)((~(uint32_t)((uint64_t)pk >> 2)) & 0x3FFFFFFFu); }
// position of the n-th (1-based) set bit of m; popcount(m) >= n
__device__ __forceinline__ int select_nth_bit(unsigned long long m, int n) {
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

// keep the best rk entries of a descending pool plus every entry tied (equal score) with the rk-th
__device__ __forceinline__ int pool_trim(const int64_t* pool, int np, int rk, int lane) {
    if (np <= rk) return np;
    const float b = key_score(pool[rk - 1]);
    int extra = 0;
    for (int b0 = rk; b0 < np; b0 += JV_WAVE) {
        const int i = b0 + lane;
        const bool tie = i < np && key_score(pool[i]) == b;
        const unsigned long long tm = __ballot(tie);
        extra += __popcll(tm);
        if (tm != ~0ull) break;
    }
    return rk + extra;
}

// rerankFloor above EVERY approximate score with several results tied at the best approximate score: jvector's
// NodeQueue.rerank then rescores the first best entry in its result heap's ARRAY order, which depends on the
// push/replace history of that binary heap.  The HBM-scratch rung logs every addTopCandidate call of level 0 and this
// replays them through a literal BoundedLongHeap (same sift rules as oracle/jv_oracle.c lh_up / lh_down) to find
// that entry.  One lane, sequential: the corner is a degenerate use of rerankFloor, exactness is all that matters.
// log entry i lives at log_top[-i]; h has rk + 1 slots (1-based heap).
static __device__ __noinline__ int replay_first_best(const int64_t* log_top, int nlog, int rk, int64_t* h) {
    int size = 0;
    for (int i = 0; i < nlog; i++) {
        const int64_t v = log_top[-i];
        if (size < rk) {  // push: append + sift up
            int p = ++size;
            int j = p >> 1;
            while (j > 0 && v < h[j]) {
                h[p] = h[j];
                p = j;
                j >>= 1;
            }
            h[p] = v;
        } else if (key_score(v) > key_score(h[1])) {  // strictly better than the worst result: replace it, sift down
            int p = 1, j = 2, k2 = 3;
            if (k2 <= size && h[k2] < h[j]) j = k2;
            while (j <= size && h[j] < v) {
                h[p] = h[j];
                p = j;
                j = p << 1;
                k2 = j + 1;
                if (k2 <= size && h[k2] < h[j]) j = k2;
            }
            h[p] = v;
        }
    }
    float best = -__builtin_huge_valf();
    int bi = -1;
    for (int i = 1; i <= size; i++) {
        const float a = key_score(h[i]);
        if (a > best) {
            best = a;
            bi = i;
        }
    }
    return bi > 0 ? key_node(h[bi]) : -1;
}

