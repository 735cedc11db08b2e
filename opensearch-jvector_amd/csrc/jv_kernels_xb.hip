// jv_kernels_xb.hip — the BATCHED exact scorer: B queries against ONE shared candidate list (gfx950 / CDNA4).
//
// What it replaces.  Lucene's exact fallback — AbstractKnnVectorQuery.exactSearch driving
// JVectorFloatVectorValues.scorer(q) / JVectorVectorScorer.score (J/JVectorFloatVectorValues.java:189-191,
// J/JVectorVectorScorer.java:36-53) — scores ONE query against the docs of ONE filter.  It is the path the reference takes
// at exactly the selectivities where a graph search stops paying (JVectorReader.search reports visited + expanded,
// J/JVectorReader.java:202-207, and AbstractKnnVectorQuery discards the approximate result once that reaches the filter's
// cardinality).  B concurrent queries under the SAME filter (a tenant, an ACL, a facet) are  Q[B x d] . C^T[d x |filter|]
// — the one dense contraction with a shared operand on this path (BASELINE.json configs[4]).
//
// How.  The answer must equal the oracle's exact scan: ids, order (score desc, doc asc) and score BITS of the canonical
// fp32 accumulation (jv_dev_common.h score_rows).  So the matrix cores only PRE-FILTER:
//   1. a bf16 mirror of the vectors ([n][kp] bf16 + |v|^2 per row, built once per index, 288 GB of HBM make it affordable)
//      and the bf16 queries feed v_mfma_f32_32x32x16_bf16: a[q][c] ~ q.c with a PROVEN bound
//          |a - q.c| <= kappa |q| |c|,   kappa = (2u + u^2) + accumulation slack,  u = 2^-8 (bf16 round-to-nearest-even)
//      (Cauchy-Schwarz over the per-element relative errors; DESIGN.md "Batched exact scorer" has the derivation and
//      tests/test_gpu_xb.py checks  lower <= canonical value <= upper  against float64 for every pair it computes);
//   2. pass A runs the tile kernel over a strided SAMPLE of the candidates and a radix select takes, per query, the k-th
//      largest LOWER bound: at least k candidates are certainly that good;
//   3. pass B runs the tile kernel over ALL candidates and keeps those whose UPPER bound reaches that value — a superset of
//      the true top k, ties and fp32 rounding plateaus included (the slack terms);
//   4. the survivors (typically a few hundred per query) are re-scored in the canonical fp32 order and the top k taken by
//      (score desc, doc asc) — bit-equal to jvo_score_ordinals + a sort.  A query whose survivor list overflows is
//      re-scored against the whole list (exactness never depends on the bound being tight).
//
// Tile kernel: 128 candidates x 128 queries per 256-thread workgroup, K step 64, both operands staged with
// global_load_lds (16 B per lane, candidate rows GATHERED by ordinal through the per-lane source address), double-buffered
// LDS with an XOR-swizzled image (conflict-free ds_read_b128 fragment reads), 4 waves x (2 x 2) 32x32 accumulators.
// Bound: HBM (1.5 KB of bf16 per candidate, read once) against 256 FLOP per byte at B = 256 — both roofs in reach; bench.py
// reports flops vs the dense bf16 peak and bytes vs 8 TB/s.
//
// Compile with -ffp-contract=off (the canonical re-score); the pre-filter's arithmetic is covered by its slack terms.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jv_device.h"

#include "jv_dev_common.h"
#include "jv_xb.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define XB_TM 128          // candidates per tile
#define XB_TN 128          // queries per tile (panel)
#define XB_BK 64           // k per step (128 B of bf16 per row)
#define XB_GROUP 1152      // LDS bytes per 8-row group: 8 x 128 B + 128 B so that groups alternate between the two halves of the 256-B bank row
#define XB_OPER (16 * XB_GROUP)  // one operand tile (128 rows)
#define XB_BUF (2 * XB_OPER)     // candidates + queries of one k step
#define XB_AUX (2 * XB_BUF)      // byte offset of the per-row constants behind the two buffers
#define XB_LDS (XB_AUX + 3 * XB_TN * 4)

// ---------------------------------------------------------------------------------------------
// fp32 rows -> bf16 mirror (+ squared norms).  One wave per row, 8-B stores; used for the index's vectors (once per index)
// and for every batch's queries.  `scalar` = rows not 16-B aligned (the caller's query matrix with d % 4 != 0).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jvx_mirror_kernel(const float* __restrict__ src, long long rows, int d, long long src_stride,
                                                         int kp, uint16_t* __restrict__ dst, float* __restrict__ norm2, int scalar) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long r = wave; r < rows; r += nw) {
        const float* row = src + r * src_stride;
        float acc = 0.0f;
        for (int p = lane; p * 4 < kp; p += 64) {
            float x[4];
            if (!scalar && p * 4 + 3 < d) {
                const f32x4 v = *(const f32x4*)(row + p * 4);
                x[0] = v[0], x[1] = v[1], x[2] = v[2], x[3] = v[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = (p * 4 + e) < d ? row[p * 4 + e] : 0.0f;
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                o[e] = (__bf16)x[e];
                acc = fmaf(x[e], x[e], acc);
            }
            *(bf16x4*)(dst + r * (long long)kp + p * 4) = o;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) norm2[r] = acc;
    }
}

extern "C" hipError_t jvk_xb_mirror(const float* src, long long rows, int d, long long src_stride, int kp, uint16_t* dst, float* norm2,
                                    int scalar, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const long long want = (rows + 3) / 4;
    const int blocks = (int)(want < 65536 ? want : 65536);
    jvx_mirror_kernel<<<blocks, 256, 0, s>>>(src, rows, d, src_stride, kp, dst, norm2, scalar);
    return hipGetLastError();
}

// Deleted ordinals (ord2doc[ord] < 0: never returned, J/JVectorReader.java:157-163) carry norm2 = -1 in the index's mirror: the
// pre-filter's epilogues read the norm anyway, so "is this candidate alive" costs them one compare and no load — a dead row is
// then no candidate at all (lower bound -inf in the sample, never a survivor).  Round 5 let dead rows into the k-of-S sample:
// the bar could sit above the k-th best LIVE candidate and true neighbours were dropped.
__global__ __launch_bounds__(256) void jvx_mark_dead_kernel(const int32_t* __restrict__ ord2doc, long long n, float* __restrict__ norm2) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        if (ord2doc[i] < 0) norm2[i] = -1.0f;
}
extern "C" hipError_t jvk_xb_mark_dead(const int32_t* ord2doc, long long n, float* norm2, hipStream_t s) {
    if (!ord2doc || n <= 0) return hipSuccess;
    const long long want = (n + 255) / 256;
    jvx_mark_dead_kernel<<<(int)(want < 16384 ? want : 16384), 256, 0, s>>>(ord2doc, n, norm2);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// doc filter -> ascending ordinal list (the reference's acceptOrds lambda, J/JVectorReader.java:157-163, evaluated once for
// the whole batch): per-block counts, one-workgroup scan, scatter.
// ---------------------------------------------------------------------------------------------
#define XB_LIST_PER_BLOCK 2048
__device__ __forceinline__ bool xb_accepts(const int32_t* ord2doc, int n, const uint64_t* accept, long long accept_docs, int i) {
    if (i >= n) return false;
    const int doc = ord2doc ? ord2doc[i] : i;
    return doc >= 0 && (long long)doc < accept_docs && ((accept[doc >> 6] >> (doc & 63)) & 1ull);
}
__global__ __launch_bounds__(256) void jvx_list_count_kernel(const int32_t* __restrict__ ord2doc, int n, const uint64_t* __restrict__ accept,
                                                             long long accept_docs, int32_t* __restrict__ counts) {
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int c = 0;
    const int base = blockIdx.x * XB_LIST_PER_BLOCK;
    for (int j = 0; j < XB_LIST_PER_BLOCK / 256; j++) c += xb_accepts(ord2doc, n, accept, accept_docs, base + j * 256 + threadIdx.x) ? 1 : 0;
    c = jv_wave_sum_int(c);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt;
}
// counts[nb] -> exclusive offsets in place, total in counts[nb]
__global__ __launch_bounds__(1024) void jvx_list_scan_kernel(int32_t* __restrict__ counts, int nb) {
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    const int t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + t;
        const int v = i < nb ? counts[i] : 0;
        s_part[t] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int add = t >= off ? s_part[t - off] : 0;
            __syncthreads();
            s_part[t] += add;
            __syncthreads();
        }
        const int incl = s_part[t];
        const int carry = s_carry;
        if (i < nb) counts[i] = carry + incl - v;
        __syncthreads();
        if (t == 1023) s_carry = carry + incl;
        __syncthreads();
    }
    if (t == 0) counts[nb] = s_carry;
}
__global__ __launch_bounds__(256) void jvx_list_scatter_kernel(const int32_t* __restrict__ ord2doc, int n, const uint64_t* __restrict__ accept,
                                                               long long accept_docs, const int32_t* __restrict__ offsets,
                                                               int32_t* __restrict__ out) {
    __shared__ int s_wave[4];
    const int base = blockIdx.x * XB_LIST_PER_BLOCK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int pos = offsets[blockIdx.x];
    for (int j = 0; j < XB_LIST_PER_BLOCK / 256; j++) {
        const int i = base + j * 256 + threadIdx.x;
        const bool acc = xb_accepts(ord2doc, n, accept, accept_docs, i);
        const unsigned long long m = __ballot(acc);
        if (lane == 0) s_wave[w] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = s_wave[k];
            before += k < w ? c : 0;
            all += c;
        }
        if (acc) out[pos + before + __popcll(m & ((1ull << lane) - 1ull))] = i;
        pos += all;
        __syncthreads();
    }
}
extern "C" hipError_t jvk_xb_build_list(const JvIndexDev* ix, const uint64_t* d_accept, long long accept_docs, int32_t* d_counts,
                                        int32_t* d_list, hipStream_t s) {
    if (ix->n <= 0) return hipSuccess;
    const int nb = (ix->n + XB_LIST_PER_BLOCK - 1) / XB_LIST_PER_BLOCK;
    jvx_list_count_kernel<<<nb, 256, 0, s>>>(ix->ord2doc, ix->n, d_accept, accept_docs, d_counts);
    jvx_list_scan_kernel<<<1, 1024, 0, s>>>(d_counts, nb);
    jvx_list_scatter_kernel<<<nb, 256, 0, s>>>(ix->ord2doc, ix->n, d_accept, accept_docs, d_counts, d_list);
    return hipGetLastError();
}
extern "C" int jvk_xb_list_blocks(int n) { return (n + XB_LIST_PER_BLOCK - 1) / XB_LIST_PER_BLOCK; }

// ---------------------------------------------------------------------------------------------
// The tile kernel.  D[query][candidate] = sum_k Qb[query][k] Vb[ord(candidate)][k] on v_mfma_f32_32x32x16_bf16:
// A operand = queries (accumulator rows), B operand = candidates (accumulator column = lane & 31), so that a lane owns ONE
// candidate and 16 queries per accumulator: sample rows are written coalesced over candidates, and a candidate's norm is
// one register.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void xb_glds16(const char* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// quality value v (higher = better, the order of the final score) and half-width e of the interval that contains the
// CANONICAL fp32 raw value (oracle/jv_oracle.c jvo_raw_dot / jvo_raw_l2 / the cosine quotient): see the header.
__device__ __forceinline__ void xb_bounds(int sim, float a, float qn, float qn2, float cn, float cn2, float kappa, float& v, float& e) {
    const float nn = qn * cn;
    if (sim == 0) {
        const float ss = qn2 + cn2;
        v = -(ss - 2.0f * a);
        e = 2.0f * kappa * nn + 1.2e-5f * ss + 1e-6f;
    } else if (sim == 1) {
        v = a;
        e = kappa * nn + 1e-6f;
    } else {
        v = a / nn;               // nn == 0: NaN / inf — such a candidate always survives (the tests below are written for that)
        e = kappa + 2e-5f;
    }
    e = e * 1.000001f + 4e-7f * fabsf(v);
}

template <int MODE>  // 0: sample pass (write lower bounds), 1: filter pass (append survivors), 2: diagnostics (write UPPER bounds)
__global__ __launch_bounds__(256, 2) void jvx_tile_kernel(const JvXbTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // XCD-aware order: the blocks of one XCD (blockIdx % 8) take a contiguous run of (tile, panel) pairs, so the panels of a
    // candidate tile meet in one L2 (bijective for any grid size)
    const int nb = gridDim.x;
    int L;
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, q = nb >> 3, r = nb & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tile = L / a.panels, panel = L - tile * a.panels;
    const int kp = a.kp, nk = kp / XB_BK;

    // ---- staging addresses: lane l of an instruction writes LDS piece l of an 8-row group (row l >> 3, physical 16-B piece
    //      l & 7) and fetches the LOGICAL piece (l & 7) ^ (l >> 3) of that row — swizzle on the source, linear destination ----
    const int lr = lane >> 3, lp = (lane & 7) ^ lr;
    const char* cptr[4];
    const char* qptr[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = 8 * (4 * w + i) + lr;
        long long ci = (long long)tile * XB_TM + row;
        if (ci >= a.rows) ci = a.rows - 1;
        long long idx = ci * a.cstride;
        if (idx >= a.C) idx = a.C - 1;
        int ord = a.ords ? a.ords[idx] : (int)idx;
        if (ord < 0 || ord >= a.n) ord = 0;  // (an invalid list entry: row 0 is staged in its place, the epilogue drops the column)
        cptr[i] = (const char*)a.vb + (size_t)ord * (size_t)kp * 2 + lp * 16;
        qptr[i] = (const char*)a.qb + (size_t)(panel * XB_TN + row) * (size_t)kp * 2 + lp * 16;
    }
    // ---- per-row constants of the epilogue: |q|, |q|^2, threshold ----
    float* s_qn = (float*)(smem + XB_AUX);
    float* s_qn2 = s_qn + XB_TN;
    float* s_thr = s_qn2 + XB_TN;
    if (tid < XB_TN) {
        const int q = panel * XB_TN + tid;
        const float n2 = q < a.B ? a.qnorm2[q] : 0.0f;
        s_qn2[tid] = n2;
        s_qn[tid] = sqrtf(n2);
        s_thr[tid] = (MODE == 1 && q < a.B) ? a.thr[q] : 0.0f;
    }
    // ---- fragment read offsets ----
    const int wn = w & 1, wm = w >> 1, h = lane >> 5, r31 = lane & 31;
    int qoff[2], coff[2], qx[2], cx[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int rq = wn * 64 + t * 32 + r31, rc = wm * 64 + t * 32 + r31;
        qoff[t] = XB_OPER + (rq >> 3) * XB_GROUP + (rq & 7) * 128;
        coff[t] = (rc >> 3) * XB_GROUP + (rc & 7) * 128;
        qx[t] = rq & 7;
        cx[t] = rc & 7;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[n][m][e] = 0.0f;

    auto stage = [&](int s, int b) {
        unsigned char* base = smem + b * XB_BUF;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            xb_glds16(cptr[i] + (size_t)s * 128, base + (4 * w + i) * XB_GROUP);
            xb_glds16(qptr[i] + (size_t)s * 128, base + XB_OPER + (4 * w + i) * XB_GROUP);
        }
    };
    stage(0, 0);
    for (int s = 0; s < nk; s++) {
        const int b = s & 1;
        if (s + 1 < nk) {
            stage(s + 1, b ^ 1);
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const unsigned char* base = smem + b * XB_BUF;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            bf16x8 fq[2], fc[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                fq[t] = *(const bf16x8*)(base + qoff[t] + (((2 * kk + h) ^ qx[t]) << 4));
                fc[t] = *(const bf16x8*)(base + coff[t] + (((2 * kk + h) ^ cx[t]) << 4));
            }
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int m = 0; m < 2; m++) acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[n], fc[m], acc[n][m], 0, 0, 0);
        }
        // every fragment read of this buffer has RETURNED before any wave may restage it (the compiler places its own
        // lgkmcnt wait at the first use, which it is free to sink below a barrier it does not understand)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue ----
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int col = wm * 64 + m * 32 + r31;
        const long long ci = (long long)tile * XB_TM + col;
        const bool cin = ci < a.rows;
        long long idx = (cin ? ci : a.rows - 1) * a.cstride;
        if (idx >= a.C) idx = a.C - 1;
        int ord = a.ords ? a.ords[idx] : (int)idx;
        bool cval = cin && ord >= 0 && ord < a.n;
        if (!cval) ord = 0;
        const float cn2 = a.vnorm2[ord], cn = sqrtf(cn2);
        cval = cval && !(cn2 < 0.0f);   // (norm2 = -1: a deleted ordinal, jvx_mark_dead_kernel)
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = wn * 64 + n * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int q = panel * XB_TN + row;
                float v, e;
                xb_bounds(a.sim, acc[n][m][reg], s_qn[row], s_qn2[row], cn, cn2, a.kappa, v, e);
                if (MODE == 0 || MODE == 2) {
                    if (cin && q < a.B) a.sample[(size_t)q * a.sample_ld + ci] = !cval ? -__builtin_inff() : (MODE == 0 ? v - e : v + e);
                } else {
                    // NaN-proof: "certainly worse" must be TRUE to drop a candidate
                    if (cval && q < a.B && !(v + e < s_thr[row])) {
                        const int pos = atomicAdd(a.surv_cnt + q, 1);
                        if (pos < a.surv_cap) a.surv[(size_t)q * a.surv_cap + pos] = (int32_t)idx;
                    }
                }
            }
        }
    }
}

extern "C" hipError_t jvk_xb_tile(const JvXbTileArgs* a, int mode, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)jvx_tile_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, XB_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)jvx_tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, XB_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)jvx_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, XB_LDS);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (a->rows <= 0 || a->B <= 0) return hipSuccess;
    const long long tiles = ((long long)a->rows + XB_TM - 1) / XB_TM;
    const long long nb = tiles * a->panels;
    if (nb > 0x7fffffffll) return hipErrorInvalidValue;
    if (mode == 0) jvx_tile_kernel<0><<<(int)nb, 256, XB_LDS, s>>>(*a);
    else if (mode == 1) jvx_tile_kernel<1><<<(int)nb, 256, XB_LDS, s>>>(*a);
    else jvx_tile_kernel<2><<<(int)nb, 256, XB_LDS, s>>>(*a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The QUERY-STATIONARY tile kernel (kp <= 768): what the tile kernel above does, restructured for the regime the path lives
// in — a few hundred queries, candidates streamed ONCE.  Above, every 128-candidate tile re-reads the whole bf16 query panel
// from L2 (as many bytes as the candidates themselves) and fetches candidate rows in 128-B slivers one k step ahead: 2 x 16 KB
// of candidate bytes in flight per CU, 13-17 % of the HBM roof (profiles/r05_xb).  Here
//   * a wave keeps the A fragments of its 32 queries for the WHOLE k range in registers (NK16 x 4 = 192 VGPRs at kp = 768),
//     loaded once per workgroup: 8 waves = 256 queries per round, two waves per SIMD;
//   * candidates arrive as WHOLE rows (kp x 2 contiguous bytes: DRAM pages, not slivers), 32 rows per stage, three stages in
//     LDS — two in flight while the third is multiplied: ~100 KB of candidate bytes in flight per CU;
//   * EVERY vector-memory instruction of the loop is an LDS-DMA copy (global_load_lds): the stage's rows, its 32 ordinals
//     (one stage further ahead, read back from LDS to form the row addresses) and its 32 norms.  No instruction waits for a
//     register from memory, so nothing the compiler inserts can drain the copies in flight (vmcnt retires in order: one
//     ordinary load's wait would); the only waits are the two counted ones written below.
// One persistent workgroup per CU.  Same bounds, same outputs as the tile kernel (xb_bounds, restated as per-query and
// per-candidate factors so that the 16 accumulator elements of a lane share them).
// ---------------------------------------------------------------------------------------------
// LDS image of one 64-wide k chunk of a stage: four groups of 8 rows x 128 B.  A ds_read_b128 serves 16 lanes = 16 consecutive
// rows = two groups at a time, so the groups of a pair sit on opposite halves of the 256-B bank row: bases 0 / 1152 / 2176 / 3328
// (1 024 B each, no overlap).  3 stages x 12 chunks x 4 352 B + 7 168 B of per-query / per-stage words = 163 840 B: the whole LDS.
#define XQ_KCH 4352
#define XQ_PF 2       // fragment reads in flight ahead of the multiply
__device__ __forceinline__ int xq_group_off(int g) { return g * 1024 + (g == 0 ? 0 : (g == 3 ? 256 : 128)); }
__device__ __forceinline__ void xb_glds4(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
// v = a * (mq * mc) + (bq + bc),  e = keq * kec + (sq + sc) + c0   (see xb_bounds: the same interval, factored)
__device__ __forceinline__ f32x4 xq_query_factors(int sim, float qn2, float kappa) {
    const float qn = sqrtf(qn2);
    if (sim == 0) return (f32x4){2.0f, -qn2, 2.0f * kappa * qn, 1.2e-5f * qn2};
    if (sim == 1) return (f32x4){1.0f, 0.0f, kappa * qn, 0.0f};
    return (f32x4){1.0f / qn, 0.0f, kappa, 0.0f};
}

// The FILTER pass's test, folded.  A candidate survives unless its upper bound is certainly below the bar:  v + e >= bar  with
// v, e of xb_bounds.  Solved for the accumulator a, with e replaced by a slightly LARGER separable bound (|v| <= 2.01 (|q|^2 + |c|^2)
// for L2, <= 1.01 |q||c| for the dot product; 2 |q||c| <= |q|^2 + |c|^2) — more survivors, never fewer:
//     survives  <=>  !( a < P_q + Q_c - R_q * |c| )
// three vector instructions per accumulator element instead of ~15 and a branch.  Every constant is rounded towards "survive"
// (1e-6 relative on each term covers the fp32 evaluation of the right-hand side).  q >= B: P = +inf (never survives).
//   L2 :  P = (bar + |q|^2 (1 - s) - c) / 2,  Q = |c|^2 (1 - s) / 2,  R = k' |q|      s = 1.2e-5 (1 + 1e-6) + 4e-7 * 2.01
//   dot:  P = bar - c,                        Q = 0,                  R = (k' + 4.04e-7) |q|
//   cos:  P = 0,                              Q = 0,                  R = -(bar - E) |q|,  E = k' + 2.05e-5   (a >= (bar - E) |q||c|)
__device__ __forceinline__ f32x4 xq_filter_constants(int sim, float qn2, float bar, float kappa, bool live) {
    const float qn = sqrtf(qn2);
    const float k1 = kappa * 1.000001f;
    float P, R;
    if (sim == 0) {
        const float s_ = 1.2e-5f * 1.000001f + 4e-7f * 2.01f;
        P = 0.5f * (bar + qn2 * (1.0f - s_) - 1.000001e-6f);
        P = P - 1e-6f * (fabsf(bar) + qn2);
        R = k1 * qn * 1.000001f;
    } else if (sim == 1) {
        P = bar - 1.000001e-6f - 1e-6f * fabsf(bar);
        R = (k1 + 4.04e-7f) * qn * 1.000001f;
    } else {
        const float d = bar - (k1 + 2.05e-5f);
        P = 0.0f;
        R = -d * qn * (d >= 0.0f ? 0.999998f : 1.000002f);
    }
    if (!live) P = __builtin_inff(), R = 0.0f;
    return (f32x4){P, R, 0.0f, 0.0f};
}

template <int NK16, int MODE>
__global__ __launch_bounds__(512) void jvx_qs_kernel(const JvXbTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = 8;
    constexpr int NK64 = NK16 / 4;                 // 64-wide k chunks per row
    constexpr int STAGE = NK64 * XQ_KCH;           // LDS bytes of one stage (32 rows)
    constexpr int NCH = NK64 * 4;                  // 1-KB copies per stage
    constexpr int LPW = (NCH + NW - 1) / NW;       // copies per wave per stage
    constexpr int kp = NK16 * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r31 = lane & 31;
    f32x4* s_qc = (f32x4*)(smem + 3 * STAGE);                    // [256] per query: mq, bq, keq, sq
    float* s_thr = (float*)(smem + 3 * STAGE + 256 * 16);        // [256] bar
    int* s_ord = (int*)(smem + 3 * STAGE + 256 * 20);            // [2][64] ordinals, one stage further ahead than the rows
    float* s_cn2 = (float*)(smem + 3 * STAGE + 256 * 20 + 512);  // [3][64] norms of the stages in LDS
    int* s_val = (int*)(smem + 3 * STAGE + 256 * 20 + 512 + 768); // [3][64] 1 = the list entry is an ordinal of this index
    if (tid < 256) {
        const int q = a.qbase + tid;
        const float n2 = q < a.B ? a.qnorm2[q] : 0.0f;
        if (MODE == 1) s_qc[tid] = xq_filter_constants(a.sim, n2, q < a.B ? a.thr[q] : __builtin_inff(), a.kappa, q < a.B);
        else s_qc[tid] = xq_query_factors(a.sim, n2, a.kappa);
        s_thr[tid] = 0.0f;
    }
    const bool active = a.qbase + w * 32 < a.B;   // (a wave without queries still copies and meets the barriers)
    // A fragments: query row w * 32 + (lane & 31), k = 16 ks + 8 h ... + 7
    bf16x8 qf[NK16];
    {
        const char* qrow = (const char*)a.qb + (size_t)(a.qbase + w * 32 + r31) * (size_t)kp * 2 + h * 16;
#pragma unroll
        for (int ks = 0; ks < NK16; ks++) qf[ks] = *(const bf16x8*)(qrow + ks * 32);
        // a USE of every fragment here: the compiler waits for these loads now.  Left pending into the stage loop, its wait
        // insertion puts a vmcnt(0) in front of the first matrix instruction of EVERY stage — and drains the copies in flight.
#pragma unroll
        for (int ks = 0; ks < NK16; ks++) asm volatile("" ::"v"(qf[ks]));
    }
    const int nsub = (a.rows + 31) / 32;           // 32-row stages of this launch
    const int G = gridDim.x;
    const int lr = lane >> 3, lp = (lane & 7) ^ lr;
    // list position of row j of stage t (clamped into the list: rows behind its end are copies of the last one, dropped in the epilogue)
    auto list_pos = [&](int t, int j) -> long long {
        long long ci = (long long)t * 32 + j;
        if (ci >= a.rows) ci = a.rows - 1;
        const long long idx = ci * a.cstride;
        return idx >= a.C ? (long long)a.C - 1 : idx;
    };
    // wave 0: the 32 ordinals of stage t -> s_ord[slot] (lanes >= 32 repeat lane 31's; without a list the ordinal IS the position)
    auto issue_ord = [&](int t, int slot) {
        if (t >= nsub) return;
        const long long idx = list_pos(t, r31);
        if (a.ords) xb_glds4(a.ords + idx, s_ord + slot * 64);
        else s_ord[slot * 64 + lane] = (int)idx;
    };
    // stage t -> LDS buffer buf, ordinals from s_ord[slot]; wave 0 also requests the 32 norms
    auto issue_rows = [&](int buf, int slot, int nslot) {
        unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; i++) {
            const int c = w + NW * i;              // chunk: k chunk c >> 2, row group c & 3
            if (NCH % NW == 0 || c < NCH) {
                const int kc = c >> 2, g = c & 3;
                int ord = s_ord[slot * 64 + 8 * g + lr];
                ord = (ord < 0 || ord >= a.n) ? 0 : ord;
                xb_glds16((const char*)a.vb + (size_t)ord * (size_t)kp * 2 + kc * 128 + lp * 16, base + kc * XQ_KCH + xq_group_off(g));
            }
        }
        if (w == 0) {
            int ord = s_ord[slot * 64 + r31];
            const bool ok = ord >= 0 && ord < a.n;
            ord = ok ? ord : 0;
            s_val[nslot * 64 + lane] = ok ? 1 : 0;
            xb_glds4(a.vnorm2 + ord, s_cn2 + nslot * 64);
        }
    };
    // fragment read offset of candidate row r31 inside a k chunk
    const int foff = xq_group_off(r31 >> 3) + (r31 & 7) * 128, fx = r31 & 7;
    int sw[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sw[j] = foff + (((2 * j + h) ^ fx) << 4);

    // ---- prologue: ordinals of the first three stages, rows of the first two ----
    const int t0 = blockIdx.x;
    if (w == 0) issue_ord(t0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t0 < nsub) issue_rows(0, 0, 0);
    if (w == 0) issue_ord(t0 + G, 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (the loop's order — the ordinals two stages ahead BEFORE the rows one stage ahead — or the counted wait below would let
    //  them slip; slot 0 was read by every wave before the barrier above)
    if (w == 0) issue_ord(t0 + 2 * G, 0);
    if (t0 + G < nsub) issue_rows(1, 1, 1);
    int buf = 0, oslot = 0, nslot = 0;      // buffer / norm slot of stage t; ordinal slot of stage t + 2
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#define XQ_STAMP(i) if (a.stamps) { const unsigned long long t_ = __builtin_readcyclecounter(); st_acc[i] += t_ - st_last; st_last = t_; }
    for (int t = t0; t < nsub; t += G) {
        // stage t has landed once at most the copies of stage t + 1 (and wave 0's norm request) are outstanding; wave 0's
        // ordinals of stage t + 2 were requested before those, so they are in LDS too
        if (t + G < nsub) {
            if (w == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // A
        XQ_STAMP(0)
        // requests, oldest first: the ordinals of stage t + 3, then the rows and norms of stage t + 2
        auto requests = [&]() {
            const int ob = oslot ^ 1;
            if (w == 0) issue_ord(t + 3 * G, ob);   // (slot ob held stage t + 1's ordinals: read before barrier B of the previous iteration)
            if (t + 2 * G < nsub) {
                int b2 = buf + 2, n2 = nslot + 2;
                b2 = b2 >= 3 ? b2 - 3 : b2;
                n2 = n2 >= 3 ? n2 - 3 : n2;
                issue_rows(b2, oslot, n2);
            }
        };
        if (!(a.dbg & 4)) requests();
        XQ_STAMP(1)
        if (active) {
            const unsigned char* base = smem + buf * STAGE;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = 0.0f;
            // fragment reads run XQ_PF steps ahead of the matrix instructions they feed (the order is pinned: left to itself the
            // scheduler issues read - wait - multiply, and a stage is then 48 exposed LDS round trips: 6 us instead of ~1.5)
            bf16x8 fc[XQ_PF + 1];
            if (!(a.dbg & 1)) {
#pragma unroll
            for (int j = 0; j < XQ_PF; j++) fc[j] = *(const bf16x8*)(base + (j >> 2) * XQ_KCH + sw[j & 3]);
#pragma unroll
            for (int ks = 0; ks < NK16; ks++) {
                if (ks + XQ_PF < NK16) fc[(ks + XQ_PF) % (XQ_PF + 1)] = *(const bf16x8*)(base + ((ks + XQ_PF) >> 2) * XQ_KCH + sw[(ks + XQ_PF) & 3]);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], fc[ks % (XQ_PF + 1)], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            XQ_STAMP(2)
            if (!(a.dbg & 2)) {
            // ---- epilogue: lane = candidate r31 of this stage ----
            // (an OPAQUE copy of the lane id, made here: the sixteen per-row LDS addresses and predicates derived from it are then
            //  formed after the multiply instead of hoisted out of the stage loop, where they would be live — and spilled, every
            //  reload a vmcnt(0) that drains the copies in flight — across the 192 registers of the query)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int h = lane_o >> 5, r31 = lane_o & 31;
            const long long ci = (long long)t * 32 + r31;
            const bool cin = ci < a.rows;
            const float cn2 = s_cn2[nslot * 64 + r31];
            // (an invalid list entry: row 0 was staged in its place; norm2 = -1: a deleted ordinal, jvx_mark_dead_kernel)
            const bool cval = cin && s_val[nslot * 64 + r31] != 0 && !(cn2 < 0.0f);
            if (MODE == 1) {
                const float cn = sqrtf(cn2);
                const float Qc = a.sim == 0 ? 0.5f * cn2 * (1.0f - (1.2e-5f * 1.000001f + 4e-7f * 2.01f)) * 0.999999f : 0.0f;
                uint32_t keep = 0u;
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const int row = w * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    const f32x4 qc = s_qc[row];
                    const float bar = fmaf(-qc[1], cn, qc[0] + Qc);
                    keep |= !(acc[reg] < bar) ? (1u << reg) : 0u;   // (NaN-proof: "certainly worse" must be TRUE to drop a candidate)
                }
                keep = cval ? keep : 0u;
                while (keep) {   // rare: a few survivors per query per 10^5 candidates
                    const int reg = __ffs((int)keep) - 1;
                    keep &= keep - 1u;
                    const int q = a.qbase + w * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    const int pos = atomicAdd(a.surv_cnt + q, 1);
                    if (pos < a.surv_cap) a.surv[(size_t)q * a.surv_cap + pos] = (int32_t)list_pos(t, r31);
                }
            } else {
                float mc = 1.0f, bc = 0.0f, kec = 1.0f, sc = 0.0f, c0 = 2e-5f;
                if (a.sim == 0) bc = -cn2, kec = sqrtf(cn2), sc = 1.2e-5f * cn2, c0 = 1e-6f;
                else if (a.sim == 1) kec = sqrtf(cn2), c0 = 1e-6f;
                else mc = 1.0f / sqrtf(cn2);
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const int row = w * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    const int q = a.qbase + row;
                    const f32x4 qc = s_qc[row];
                    const float v = fmaf(acc[reg] * qc[0], mc, qc[1] + bc);
                    float e = fmaf(qc[2], kec, qc[3] + sc) + c0;
                    e = fmaf(e, 1.000001f, 4e-7f * fabsf(v));
                    if (cin && q < a.B) a.sample[(size_t)q * a.sample_ld + ci] = !cval ? -__builtin_inff() : (MODE == 0 ? v - e : v + e);
                    if ((reg & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
        }
        XQ_STAMP(3)
        if (a.dbg & 4) requests();
        // B: every fragment read of this buffer (and of the ordinal / norm slots) has returned before any wave restages it
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        XQ_STAMP(4)
        buf = buf == 2 ? 0 : buf + 1;
        nslot = nslot == 2 ? 0 : nslot + 1;
        oslot ^= 1;
    }
    if (a.stamps && w == 1 && lane == 0)
        for (int i = 0; i < 5; i++) atomicAdd(a.stamps + i, st_acc[i]);
#undef XQ_STAMP
}

template <int NK16>
static hipError_t xq_launch(const JvXbTileArgs* a, int mode, int blocks, hipStream_t s) {
    const int lds = 3 * (NK16 / 4) * XQ_KCH + 256 * 20 + 512 + 768 + 768;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)jvx_qs_kernel<NK16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)jvx_qs_kernel<NK16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)jvx_qs_kernel<NK16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (mode == 0) jvx_qs_kernel<NK16, 0><<<blocks, 512, lds, s>>>(*a);
    else if (mode == 1) jvx_qs_kernel<NK16, 1><<<blocks, 512, lds, s>>>(*a);
    else jvx_qs_kernel<NK16, 2><<<blocks, 512, lds, s>>>(*a);
    return hipGetLastError();
}
// one round of up to 256 queries (a->qbase ...)
extern "C" int jvk_xb_qs_ok(int kp) { return kp == 768 || kp == 512 || kp == 384 || kp == 256 || kp == 128; }
extern "C" hipError_t jvk_xb_qs(const JvXbTileArgs* a, int mode, int cus, hipStream_t s) {
    if (a->rows <= 0 || a->B <= 0) return hipSuccess;
    const int nsub = (a->rows + 31) / 32;
    const int blocks = nsub < cus ? nsub : cus;
    switch (a->kp) {
        case 768: return xq_launch<48>(a, mode, blocks, s);
        case 512: return xq_launch<32>(a, mode, blocks, s);
        case 384: return xq_launch<24>(a, mode, blocks, s);
        case 256: return xq_launch<16>(a, mode, blocks, s);
        case 128: return xq_launch<8>(a, mode, blocks, s);
        default: return hipErrorNotSupported;
    }
}

// ---------------------------------------------------------------------------------------------
// A per-query bar from the sample row: a value L such that AT LEAST k sample entries are >= L (any such value is a valid bar;
// the closer to the true k-th largest, the fewer survivors).  Three reads of the row (L2 hits): min / max of the finite
// entries, a 2 048-bucket histogram over [min, max] (linear buckets spread a bell-shaped score distribution over the LDS
// banks — a radix histogram on the float's top byte put nearly every entry on one word: 130 us per launch for 25 000-entry
// rows, a quarter of the whole call), and the smallest entry of the buckets that hold the top k.  That entry is a sample
// value with >= k entries at or above it by construction — no float edge reasoning.  Rows with NaN / +inf entries, or fewer
// than k finite ones, get -inf: every candidate survives (correct, merely slow).
// ---------------------------------------------------------------------------------------------
#define XB_KTH_BUCKETS 2048
#define XB_KT 1024   // threads: a row of 25 000 - 65 000 floats is read three times
__global__ __launch_bounds__(XB_KT) void jvx_kth_kernel(const float* __restrict__ sample, int ld, int S, int k, float* __restrict__ thr) {
    __shared__ int s_hist[XB_KTH_BUCKETS];
    __shared__ float s_red[2 * (XB_KT / 64)];
    __shared__ int s_cut, s_bad;
    const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* row = sample + (size_t)q * ld;
    const float inf = __builtin_inff();
    float lo = inf, hi = -inf;
    int bad = 0, fin = 0;
    for (int i0 = t; i0 < S; i0 += 4 * XB_KT) {   // four independent loads per step (one at a time, a pass is ~100 exposed L2 round trips)
        float vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vv[u] = i0 + XB_KT * u < S ? row[i0 + XB_KT * u] : -inf;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float v = vv[u];
            if (v != v || v == inf) bad = 1;
            else if (v != -inf) {
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
                fin++;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    bad = __ballot(bad != 0) != 0ull;
    fin = jv_wave_sum_int(fin);
    if (lane == 0) s_red[w] = lo, s_red[XB_KT / 64 + w] = hi;
    for (int i = t; i < XB_KTH_BUCKETS; i += XB_KT) s_hist[i] = 0;
    if (t == 0) s_cut = 0, s_bad = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&s_cut, fin);
    if (lane == 0 && bad) atomicOr(&s_bad, 1);
    lo = s_red[0], hi = s_red[XB_KT / 64];
#pragma unroll
    for (int i = 1; i < XB_KT / 64; i++) lo = fminf(lo, s_red[i]), hi = fmaxf(hi, s_red[XB_KT / 64 + i]);
    __syncthreads();
    const int total = s_cut;
    if (s_bad != 0 || total < k) {
        if (t == 0) thr[q] = -inf;
        return;
    }
    const float scale = hi > lo ? (float)(XB_KTH_BUCKETS - 1) / (hi - lo) : 0.0f;
    for (int i0 = t; i0 < S; i0 += 4 * XB_KT) {
        float vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vv[u] = i0 + XB_KT * u < S ? row[i0 + XB_KT * u] : -inf;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float v = vv[u];
            if (v != -inf) {
                int b = (int)((v - lo) * scale);
                b = b < 0 ? 0 : (b > XB_KTH_BUCKETS - 1 ? XB_KTH_BUCKETS - 1 : b);
                atomicAdd(&s_hist[b], 1);
            }
        }
    }
    __syncthreads();
    if (w == 0) {  // first bucket from the top at which the running count reaches k: one wave, 32 buckets per lane
        int mine = 0;
        const int b0 = XB_KTH_BUCKETS - 32 * (lane + 1);  // lane 0 owns the top 32 buckets
        for (int j = 0; j < 32; j++) mine += s_hist[b0 + j];
        int incl = mine;  // inclusive prefix over lanes (top first)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const int before = incl - mine;
        if (before < k && incl >= k) {
            int cum = before, b = b0 + 31;
            for (; b > b0; b--) {
                if (cum + s_hist[b] >= k) break;
                cum += s_hist[b];
            }
            s_cut = b;
        }
    }
    __syncthreads();
    const int cut = s_cut;
    float m = inf;
    for (int i0 = t; i0 < S; i0 += 4 * XB_KT) {
        float vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) vv[u] = i0 + XB_KT * u < S ? row[i0 + XB_KT * u] : -inf;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float v = vv[u];
            if (v != -inf) {
                int b = (int)((v - lo) * scale);
                b = b < 0 ? 0 : (b > XB_KTH_BUCKETS - 1 ? XB_KTH_BUCKETS - 1 : b);
                if (b >= cut) m = fminf(m, v);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
    __syncthreads();
    if (lane == 0) s_red[w] = m;
    __syncthreads();
    if (t == 0) {
        float r = s_red[0];
        for (int i = 1; i < XB_KT / 64; i++) r = fminf(r, s_red[i]);
        thr[q] = r;
    }
}
extern "C" hipError_t jvk_xb_kth(const float* sample, int ld, int S, int k, float* thr, int B, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    jvx_kth_kernel<<<B, XB_KT, 0, s>>>(sample, ld, S, k, thr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Canonical re-score of a query's survivors (or of the whole list) + top k by (score desc, doc asc): one 256-thread
// workgroup per query; a wave scores 64 rows per round with score_rows (the arithmetic of every exact score in this engine),
// keys that beat the running k-th best collect in an LDS buffer that is bitonic-sorted and cut back to k when it fills.
// ---------------------------------------------------------------------------------------------
#define XB_KCAP 4096
#define XB_RT 512   // threads of a re-score workgroup: 8 waves x 64 survivors per round (a 185-entry list is ONE round of two 16-row passes per wave, not four)
__device__ __forceinline__ void xb_sort_desc(int64_t* keys, int32_t* pay, int n_pow2, int tid) {
    for (int size = 2; size <= n_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (n_pow2 >> 1); i += XB_RT) {
                const int lo = (i / stride) * (stride << 1) + (i % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const int64_t x = keys[lo], y = keys[hi];
                if ((x < y) == desc) {
                    keys[lo] = y;
                    keys[hi] = x;
                    const int32_t p = pay[lo];
                    pay[lo] = pay[hi];
                    pay[hi] = p;
                }
            }
            __syncthreads();
        }
    }
}

template <int NCHT>
__global__ __launch_bounds__(XB_RT) void jvx_rescore_kernel(const JvIndexDev ix, const JvXbRescoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = blockIdx.x;
    float* q_lds = (float*)smem;
    size_t off = (size_t)ix.nch * 64 * sizeof(float);
    int64_t* keys = (int64_t*)(smem + off);
    off += (size_t)XB_KCAP * 8;
    int32_t* pay = (int32_t*)(smem + off);
    off += (size_t)XB_KCAP * 4;
    float* todo_score = (float*)(smem + off) + w * 64;
    off += (XB_RT / 64) * 64 * 4;
    int32_t* todo = (int32_t*)(smem + off) + w * 64;
    off += (XB_RT / 64) * 64 * 4;
    int* s_cnt = (int*)(smem + off);

    const float* query = a.queries + (size_t)q * ix.d;
    for (int i = tid; i < ix.nch * 64; i += XB_RT) q_lds[i] = i < ix.d ? query[i] : 0.0f;
    if (tid == 0) *s_cnt = 0;
    __syncthreads();
    float qnorm2 = 0.0f;
    if (ix.sim == 2) qnorm2 = query_norm2(ix, q_lds, lane), qnorm2 = __shfl(qnorm2, 0, 64);

    const int nsurv = a.surv_cnt ? a.surv_cnt[q] : 0;
    const bool all = a.force_all || !a.surv_cnt || nsurv > a.surv_cap;
    const int total = all ? a.C : nsurv;
    const int k = a.topK;
    int64_t thr_key = KEY_MIN;
    if (tid == 0 && a.out_info) {
        atomicAdd((unsigned long long*)a.out_info + 0, (unsigned long long)total);
        if (all && a.surv_cnt && !a.force_all) atomicAdd((unsigned long long*)a.out_info + 1, 1ull);
    }
    for (int base = 0; base < total; base += XB_RT) {
        const int idx = base + tid;
        int ord = -1, doc = -1;
        if (idx < total) {
            const int ci = all ? idx : a.surv[(size_t)q * a.surv_cap + idx];
            ord = a.ords ? a.ords[ci] : ci;
            if (ord >= 0 && ord < ix.n) doc = ix.ord2doc ? ix.ord2doc[ord] : ord;
        }
        const bool ok = doc >= 0;
        const unsigned long long mk = __ballot(ok);
        const int m = __popcll(mk);
        const int pos = __popcll(mk & ((1ull << lane) - 1ull));
        if (ok) todo[pos] = ord;
        __syncthreads();
        if (m > 0) score_rows<NCHT, 2>(ix, q_lds, todo, m, todo_score, qnorm2, ix.score_scale, lane);  // (16 rows in flight per wave at d = 768: a survivor list is a few passes of dependent HBM round trips)
        __syncthreads();
        int64_t key = KEY_MIN;
        if (ok) key = make_key(todo_score[pos], doc);
        const bool take = ok && key > thr_key;
        const unsigned long long tk = __ballot(take);
        int wbase = 0;
        if (lane == 0 && tk) wbase = atomicAdd(s_cnt, __popcll(tk));
        wbase = __shfl(wbase, 0, 64);
        if (take) {
            const int p = wbase + __popcll(tk & ((1ull << lane) - 1ull));
            keys[p] = key;
            pay[p] = ord;
        }
        __syncthreads();
        const int cnt = *s_cnt;
        if (cnt + XB_RT > XB_KCAP) {  // (workgroup-uniform) sort, keep the best k, raise the bar
            int np = 1;
            while (np < cnt) np <<= 1;
            for (int i = cnt + tid; i < np; i += XB_RT) keys[i] = KEY_MIN, pay[i] = -1;
            __syncthreads();
            xb_sort_desc(keys, pay, np, tid);
            const int keep = cnt < k ? cnt : k;
            if (keep >= k) thr_key = keys[k - 1];
            __syncthreads();
            if (tid == 0) *s_cnt = keep;
            __syncthreads();
        }
    }
    __syncthreads();
    const int cnt = *s_cnt;
    int np = 1;
    while (np < cnt) np <<= 1;
    for (int i = cnt + tid; i < np; i += XB_RT) keys[i] = KEY_MIN, pay[i] = -1;
    __syncthreads();
    if (cnt > 1) xb_sort_desc(keys, pay, np, tid);
    const int outn = cnt < k ? cnt : k;
    for (int i = tid; i < k; i += XB_RT) {
        const bool v = i < outn;
        const int64_t key = v ? keys[i] : 0;
        if (a.out_nodes) a.out_nodes[(size_t)q * k + i] = v ? pay[i] : -1;
        if (a.out_docs) a.out_docs[(size_t)q * k + i] = v ? key_node(key) : -1;
        if (a.out_scores) a.out_scores[(size_t)q * k + i] = v ? key_score(key) : 0.0f;
    }
    if (tid == 0 && a.out_count) a.out_count[q] = outn;
}

extern "C" int jvk_xb_rescore_lds(const JvIndexDev* ix) { return ix->nch * 64 * 4 + XB_KCAP * 12 + 2 * (XB_RT / 64) * 64 * 4 + 16; }
extern "C" hipError_t jvk_xb_rescore(const JvIndexDev* ix, const JvXbRescoreArgs* a, int nq, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    const int lds = jvk_xb_rescore_lds(ix);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)jvx_rescore_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)jvx_rescore_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)jvx_rescore_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)jvx_rescore_kernel<24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const bool fixed = ix->nvq_M == 0 && ix->vectors && ix->stride == ix->nch * 64;
    if (fixed && ix->nch == 2) jvx_rescore_kernel<2><<<nq, XB_RT, lds, s>>>(*ix, *a);
    else if (fixed && ix->nch == 12) jvx_rescore_kernel<12><<<nq, XB_RT, lds, s>>>(*ix, *a);
    else if (fixed && ix->nch == 24) jvx_rescore_kernel<24><<<nq, XB_RT, lds, s>>>(*ix, *a);
    else jvx_rescore_kernel<0><<<nq, XB_RT, lds, s>>>(*ix, *a);
    return hipGetLastError();
}
