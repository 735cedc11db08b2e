// jv_build_kernels.hip — GPU kernels of the WRITE side (index construction; not the graded hot path, see jv_build.h /
// builder_gpu.py): jvb_prune_rows_device (jvector-style diversity selection, GraphIndexBuilder's retainDiverse: alpha sweeps
// 1.0, 1.2, .. <= alpha; a candidate is kept unless an already-selected neighbour s has sim(c, s) > sim(c, centre) * a — one
// workgroup per row, from the vectors in HBM to the selected row), PQ training and encoding.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

// ---------------------------------------------------------------------------------------------------------------
// Fused diversity selection (round 4; SURVEY 8(f) row 3, J/JVectorWriter.java:1383-1422 addGraphNode / cleanup): everything
// robust_prune needs for ONE row in ONE workgroup — the centre's scores, the (score desc, id asc) order, duplicate removal,
// the candidate x candidate similarities and the alpha sweep — straight from the vectors in HBM.  Round 3 did the gather,
// the two batched GEMMs (torch.bmm -> Tensile), the sorts and the reorder of a [rows][Lc][d] tensor with torch ops: more
// than half of the build's GPU time and [rows][Lc][d] floats of temporary HBM.
//   workgroup = 256 threads = one row of up to JVB_PRUNE_MAX_LC candidates;
//   LDS: centre vector, the sorted ids / scores / squared norms, the Lc x Lc matrix of dot products, one k-tile of the
//        candidates' vectors ([Lc][KC + 1] floats);
//   the dot products are an LDS-tiled fp32 product: thread (ty, tx) of a 16 x 16 grid owns a TR x TR register tile of the
//   matrix (TR = ceil(Lc / 16) <= 10), k advances KC = 32 dimensions per tile; sums run in a fixed order, so two builds of
//   the same input select the same rows (tests/test_gpu_builder.py::test_gpu_builds_are_reproducible).
// ---------------------------------------------------------------------------------------------------------------
#define JVB_PRUNE_MAX_LC 160
#define JVB_PRUNE_KC 32
__device__ __forceinline__ float jvb_sim_from_dot(int sim, float dot, float sqa, float sqb) {
    if (sim == 0) {
        float d2 = sqa + sqb - 2.0f * dot;
        d2 = d2 < 0.0f ? 0.0f : d2;
        return 1.0f / (1.0f + d2);
    }
    if (sim == 1) return (1.0f + dot) * 0.5f;
    float den = sqa * sqb;
    den = den < 1e-30f ? 1e-30f : den;
    return (1.0f + dot / sqrtf(den)) * 0.5f;
}
template <int JVB_PRUNE_TR>  // register tile edge: 16 * TR >= Lc (5: <= 80 candidates, 7: <= 112, 10: <= 160)
__global__ __launch_bounds__(256) void jvb_prune_rows_kernel(const float* __restrict__ base, int d, int stride, int sim,
                                                             const long long* __restrict__ centers,   // [S]
                                                             const int32_t* __restrict__ cand, int cand_stride,  // [S][Lc], -1 = empty
                                                             int S, int Lc, int R, float alpha,
                                                             int32_t* __restrict__ sel, int sel_stride,   // [S][R] (-1 padded)
                                                             int32_t* __restrict__ nsel_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = blockIdx.x;
    if (row >= S) return;
    const int dpad = (d + 3) & ~3;
    float* cvec = (float*)smem;                         // [dpad] centre
    float* G = cvec + dpad;                             // [Lc][Lc] dot products (sorted order)
    float* tile = G + (size_t)Lc * Lc;                  // [Lc][KC + 1]
    float* scs = tile + (size_t)Lc * (JVB_PRUNE_KC + 1);   // [Lc] scores to the centre, sorted
    float* sqs = scs + Lc;                              // [Lc] squared norms, sorted
    float* sc0 = sqs + Lc;                              // [Lc] unsorted scratch
    float* sq0 = sc0 + Lc;                              // [Lc]
    int32_t* ids = (int32_t*)(sq0 + Lc);                // [Lc] sorted ids
    int32_t* id0 = ids + Lc;                            // [Lc] unsorted ids
    int32_t* sel_idx = id0 + Lc;                        // [R]
    uint8_t* state = (uint8_t*)(sel_idx + R);           // [Lc]: 0 = not a candidate, 1 = open, 2 = taken
    const long long c = centers[row];
    const float* cp = base + (size_t)c * (size_t)stride;
    for (int i = tid; i < dpad; i += 256) cvec[i] = i < d ? cp[i] : 0.0f;
    for (int i = tid; i < Lc; i += 256) id0[i] = cand[(size_t)row * cand_stride + i];
    __syncthreads();
    // ---- scores to the centre: one wave per candidate, lanes stride the row 4 floats at a time ----
    float sqc = 0.0f;
    {
        float a = 0.0f;
        for (int i = lane * 4; i < d; i += 256) {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (i + e < d) a = fmaf(cvec[i + e], cvec[i + e], a);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
        sqc = a;
    }
    for (int x = wv; x < Lc; x += 4) {
        const int id = id0[x];
        float dot = 0.0f, sq = 0.0f;
        if (id >= 0) {
            const float* vp = base + (size_t)id * (size_t)stride;
            for (int i = lane * 4; i < d; i += 256) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (i + e < d) {
                        const float v = vp[i + e];
                        dot = fmaf(v, cvec[i + e], dot);
                        sq = fmaf(v, v, sq);
                    }
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                dot += __shfl_xor(dot, o, 64);
                sq += __shfl_xor(sq, o, 64);
            }
        }
        if (lane == 0) {
            const bool ok = id >= 0 && (long long)id != c;
            sc0[x] = ok ? jvb_sim_from_dot(sim, dot, sq, sqc) : -__builtin_huge_valf();
            sq0[x] = sq;
        }
    }
    __syncthreads();
    // ---- order: score desc, id asc, position asc (rank by counting); duplicates (equal id => adjacent) lose ----
    for (int x = tid; x < Lc; x += 256) {
        const float sx = sc0[x];
        const int ix = id0[x];
        int r = 0;
        for (int y = 0; y < Lc; y++) {
            const float sy = sc0[y];
            const int iy = id0[y];
            r += (sy > sx || (sy == sx && (iy < ix || (iy == ix && y < x)))) ? 1 : 0;
        }
        ids[r] = ix;
        scs[r] = sx;
        sqs[r] = sq0[x];
    }
    __syncthreads();
    for (int x = tid; x < Lc; x += 256) {
        const bool ok = scs[x] > -__builtin_huge_valf() && !(x > 0 && ids[x] == ids[x - 1]);
        state[x] = ok ? 1 : 0;
    }
    // ---- dot products of every pair of candidates, LDS-tiled ----
    const int ty = tid >> 4, tx = tid & 15;
    float acc[JVB_PRUNE_TR][JVB_PRUNE_TR];
#pragma unroll
    for (int i = 0; i < JVB_PRUNE_TR; i++)
#pragma unroll
        for (int j = 0; j < JVB_PRUNE_TR; j++) acc[i][j] = 0.0f;
    for (int k0 = 0; k0 < d; k0 += JVB_PRUNE_KC) {
        __syncthreads();
        // tile[x][kk] = V[ids[x]][k0 + kk]: 8 threads per candidate row read 4 floats each (128 contiguous bytes per row)
        for (int e = tid; e < Lc * (JVB_PRUNE_KC / 4); e += 256) {
            const int x = e / (JVB_PRUNE_KC / 4), q4 = e % (JVB_PRUNE_KC / 4);
            const int id = ids[x];
            const int kk = q4 * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (id >= 0) {
                const float* vp = base + (size_t)id * (size_t)stride + k0 + kk;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (k0 + kk + t < d) v[t] = vp[t];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) tile[(size_t)x * (JVB_PRUNE_KC + 1) + kk + t] = v[t];
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < JVB_PRUNE_KC; kk++) {
            float a[JVB_PRUNE_TR], b[JVB_PRUNE_TR];
#pragma unroll
            for (int i = 0; i < JVB_PRUNE_TR; i++) {
                const int r = ty + 16 * i, cc = tx + 16 * i;
                a[i] = r < Lc ? tile[(size_t)r * (JVB_PRUNE_KC + 1) + kk] : 0.0f;
                b[i] = cc < Lc ? tile[(size_t)cc * (JVB_PRUNE_KC + 1) + kk] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < JVB_PRUNE_TR; i++)
#pragma unroll
                for (int j = 0; j < JVB_PRUNE_TR; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < JVB_PRUNE_TR; i++)
#pragma unroll
        for (int j = 0; j < JVB_PRUNE_TR; j++) {
            const int r = ty + 16 * i, cc = tx + 16 * j;
            if (r < Lc && cc < Lc) G[(size_t)r * Lc + cc] = acc[i][j];
        }
    __syncthreads();
    // ---- the alpha sweep (jvector's diversity selection), first wave ----
    if (wv == 0) {
        int nsel = 0;
        for (float a = 1.0f; a <= alpha + 1e-6f && nsel < R; a += 0.2f) {
            for (int x = 0; x < Lc && nsel < R; x++) {
                if (state[x] != 1) continue;
                const float thr = scs[x] * a;
                bool bad = false;
                for (int s0 = 0; s0 < nsel; s0 += WAVE) {
                    const int si = s0 + lane;
                    bool over = false;
                    if (si < nsel) {
                        const int y = sel_idx[si];
                        over = jvb_sim_from_dot(sim, G[(size_t)x * Lc + y], sqs[x], sqs[y]) > thr;
                    }
                    if (__ballot(over)) {
                        bad = true;
                        break;
                    }
                }
                if (!bad) {
                    if (lane == 0) {
                        state[x] = 2;
                        sel_idx[nsel] = x;
                    }
                    nsel++;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        for (int i = lane; i < R; i += WAVE) sel[(size_t)row * sel_stride + i] = i < nsel ? ids[sel_idx[i]] : -1;
        if (lane == 0) nsel_out[row] = nsel;
    }
}

extern "C" int jvb_prune_rows_device(const float* base, int d, int stride, int sim, const long long* centers, const int32_t* cand,
                                     int cand_stride, int S, int Lc, int R, float alpha, int32_t* sel, int sel_stride, int32_t* nsel,
                                     void* stream) {
    if (S <= 0) return 0;
    if (Lc < 1 || Lc > JVB_PRUNE_MAX_LC || R < 1 || R > 256) return -4;
    const int dpad = (d + 3) & ~3;
    size_t lds = (size_t)dpad * 4 + (size_t)Lc * Lc * 4 + (size_t)Lc * (JVB_PRUNE_KC + 1) * 4 + (size_t)Lc * 4 * 4 + (size_t)Lc * 4 * 2 +
                 (size_t)R * 4 + (size_t)Lc;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return -4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)jvb_prune_rows_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)jvb_prune_rows_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)jvb_prune_rows_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return -3;
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (Lc <= 80) jvb_prune_rows_kernel<5><<<S, 256, lds, st>>>(base, d, stride, sim, centers, cand, cand_stride, S, Lc, R, alpha, sel, sel_stride, nsel);
    else if (Lc <= 112) jvb_prune_rows_kernel<7><<<S, 256, lds, st>>>(base, d, stride, sim, centers, cand, cand_stride, S, Lc, R, alpha, sel, sel_stride, nsel);
    else jvb_prune_rows_kernel<10><<<S, 256, lds, st>>>(base, d, stride, sim, centers, cand, cand_stride, S, Lc, R, alpha, sel, sel_stride, nsel);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------------------------
// Back-links of one insertion batch (round 4; J/JVectorWriter.java:1383-1422: addGraphNode links both ways): for every new
// edge u -> s add s -> u.  Round 3 grouped the edges by target with a torch stable sort + unique_consecutive +
// repeat_interleave.  Here: (1) every edge pushes itself onto its target's list (atomicExch on a per-node head word; the
// first edge of a target also appends the target to a work list), (2) one wave per touched target walks its list, SORTS the
// sources by id — the order the atomics happened in is gone, two builds give the same rows — and either appends them to the
// row (it has room: neighborOverflow slack) or emits the row + the (smallest-id) new sources as candidates for a re-prune.
// ---------------------------------------------------------------------------------------------------------------
#define JVB_BL_MAX 1024  // sources of one target held in LDS for the sort (more than that: the smallest ids are kept)
__global__ __launch_bounds__(256) void jvb_backlink_push_kernel(const int32_t* __restrict__ sel, int sel_stride, int B, int R,
                                                                int32_t* __restrict__ head, int32_t* __restrict__ next,
                                                                int32_t* __restrict__ touched, int32_t* __restrict__ counters) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)B * R) return;
    const int p = (int)(e / R), j = (int)(e % R);
    const int s = sel[(size_t)p * sel_stride + j];
    if (s < 0) return;
    const int old = atomicExch(&head[s], (int)e);
    next[e] = old;
    if (old < 0) touched[atomicAdd(&counters[0], 1)] = s;
}
__global__ __launch_bounds__(64) void jvb_backlink_apply_kernel(const long long* __restrict__ u, int R, int32_t* __restrict__ adj, int Rcap,
                                                                int32_t* __restrict__ deg, int32_t* __restrict__ head,
                                                                const int32_t* __restrict__ next, const int32_t* __restrict__ touched,
                                                                int32_t* __restrict__ counters, int32_t* __restrict__ ov_nodes,
                                                                int32_t* __restrict__ ov_cand, int ov_rows, int Lc) {
    __shared__ int32_t src[JVB_BL_MAX];
    __shared__ int32_t srt[JVB_BL_MAX];
    const int lane = threadIdx.x;
    const int ntouched = counters[0];
    for (int t = blockIdx.x; t < ntouched; t += gridDim.x) {
        const int s = touched[t];
        // walk the list (one lane: a pointer chase), keep the sources; beyond the LDS array only the smallest ids stay
        int c = 0;
        if (lane == 0) {
            int e = head[s];
            head[s] = -1;
            int worst = -1, worst_i = -1;
            while (e >= 0) {
                const int su = (int)u[e / R];
                if (c < JVB_BL_MAX) {
                    src[c++] = su;
                } else {  // (rare: a hub node met by more than 1 024 sources of one batch)
                    if (worst_i < 0) {
                        for (int i = 0; i < JVB_BL_MAX; i++)
                            if (src[i] > worst) worst = src[i], worst_i = i;
                    }
                    if (su < worst) {
                        src[worst_i] = su;
                        worst = -1;
                        worst_i = -1;
                    }
                }
                e = next[e];
            }
        }
        c = __shfl(c, 0, 64);
        __syncthreads();
        // sort by id (rank by counting: ids of one target's sources are distinct)
        for (int i = lane; i < c; i += 64) {
            const int v = src[i];
            int r = 0;
            for (int k = 0; k < c; k++) r += src[k] < v ? 1 : 0;
            srt[r] = v;
        }
        __syncthreads();
        const int d0 = deg[s];
        if (d0 + c <= Rcap) {
            for (int i = lane; i < c; i += 64) adj[(size_t)s * Rcap + d0 + i] = srt[i];
            if (lane == 0) deg[s] = d0 + c;
        } else {
            int row = 0;
            if (lane == 0) row = atomicAdd(&counters[1], 1);
            row = __shfl(row, 0, 64);
            if (row < ov_rows) {
                if (lane == 0) ov_nodes[row] = s;
                for (int i = lane; i < Lc; i += 64) {
                    int v = -1;
                    if (i < Rcap) v = adj[(size_t)s * Rcap + i];
                    else if (i - Rcap < c) v = srt[i - Rcap];
                    ov_cand[(size_t)row * Lc + i] = v;
                }
            }
        }
        __syncthreads();
    }
}
// counters[0] = touched targets, counters[1] = rows that need a re-prune (both zeroed here); head[] must be -1 everywhere on
// entry and is -1 everywhere on return.  Returns 0; the caller reads counters[1] and prunes ov_cand's rows.
extern "C" int jvb_backlinks_device(const long long* u, int B, const int32_t* sel, int sel_stride, int R, int32_t* adj, int Rcap,
                                    int32_t* deg, int32_t* head, int32_t* next, int32_t* touched, int32_t* counters,
                                    int32_t* ov_nodes, int32_t* ov_cand, int ov_rows, int Lc, void* stream) {
    if (B <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counters, 0, 2 * sizeof(int32_t), st) != hipSuccess) return -3;
    const long long E = (long long)B * R;
    jvb_backlink_push_kernel<<<(unsigned)((E + 255) / 256), 256, 0, st>>>(sel, sel_stride, B, R, head, next, touched, counters);
    jvb_backlink_apply_kernel<<<4096, 64, 0, st>>>(u, R, adj, Rcap, deg, head, next, touched, counters, ov_nodes, ov_cand, ov_rows, Lc);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------------------------
// PQ encoding (J/JVectorIndexQuantization.java:114-140, merge re-encode J/JVectorWriter.java:1117-1124):
// code[i][m] = argmin_c sum_j (x'[i][off_m + j] - codebook[m][c][j])^2 with x' = x - globalCentroid, the sum being
// the SAME sequential fmaf chain over the subspace's dimensions that the search kernels use for a query's look-up
// table (jv_dev_common.h build_lut), ties -> lowest centroid index.  So encode(x) == argmin of x's own LUT row:
// the write side and the read side share one arithmetic, and the CPU encoder (jv_build_cpu.cpp) is bit-identical.
// One lane = one vector; the centroid components are wave-uniform (scalar loads); subspace of up to 64 dims.
// ---------------------------------------------------------------------------------------------------------------
#define PQE_MAX_DS 64
__global__ __launch_bounds__(256) void jvb_pq_encode_kernel(const float* __restrict__ vectors, long long n, int d, int stride,
                                                            int M, int K, const int32_t* __restrict__ sub_off,
                                                            const float* __restrict__ codebooks,      // concat over m of [K][ds_m]
                                                            const long long* __restrict__ cb_off,     // [M] float offset of subspace m
                                                            const float* __restrict__ centroid,       // [d] or nullptr
                                                            uint8_t* __restrict__ codes, int code_stride) {
    const int m = blockIdx.y;
    const int d0 = sub_off[m], ds = sub_off[m + 1] - d0;
    const float* cb = codebooks + cb_off[m];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float x[PQE_MAX_DS];
        const float* row = vectors + (size_t)i * (size_t)stride + d0;
#pragma unroll
        for (int j = 0; j < PQE_MAX_DS; j++) {
            x[j] = 0.0f;
            if (j < ds) x[j] = centroid ? row[j] - centroid[d0 + j] : row[j];
        }
        float best = 3.4028234663852886e38f;
        int bc = 0;
        for (int c = 0; c < K; c++) {
            const float* cv = cb + (size_t)c * ds;  // wave-uniform address
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < PQE_MAX_DS; j++) {
                if (j < ds) {
                    const float df = x[j] - cv[j];
                    acc = fmaf(df, df, acc);
                }
            }
            if (acc < best) {
                best = acc;
                bc = c;
            }
        }
        codes[(size_t)i * (size_t)code_stride + m] = (uint8_t)bc;
    }
}

// Subspaces wider than PQE_MAX_DS dimensions (few subspaces over a long vector: d = 768 with M = 8, d = 1536 with M = 16;
// the reference accepts any num_pq_subspaces, K/index/.../KNNVectorsFormatParams.java:143): the same sequential fmaf
// chain with the vector's components re-read per centroid (L1 / L2 hits) instead of held in registers.
__global__ __launch_bounds__(256) void jvb_pq_encode_wide_kernel(const float* __restrict__ vectors, long long n, int d, int stride,
                                                                 int M, int K, const int32_t* __restrict__ sub_off,
                                                                 const float* __restrict__ codebooks, const long long* __restrict__ cb_off,
                                                                 const float* __restrict__ centroid, uint8_t* __restrict__ codes,
                                                                 int code_stride) {
    const int m = blockIdx.y;
    const int d0 = sub_off[m], ds = sub_off[m + 1] - d0;
    const float* cb = codebooks + cb_off[m];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float* row = vectors + (size_t)i * (size_t)stride + d0;
        float best = 3.4028234663852886e38f;
        int bc = 0;
        for (int c = 0; c < K; c++) {
            const float* cv = cb + (size_t)c * ds;
            float acc = 0.0f;
            for (int j = 0; j < ds; j++) {
                const float xj = centroid ? row[j] - centroid[d0 + j] : row[j];
                const float df = xj - cv[j];
                acc = fmaf(df, df, acc);
            }
            if (acc < best) {
                best = acc;
                bc = c;
            }
        }
        codes[(size_t)i * (size_t)code_stride + m] = (uint8_t)bc;
    }
}

// device pointers everywhere; sub_off [M + 1] int32, cb_off [M] int64
extern "C" int jvb_pq_encode_device(const float* vectors, long long n, int d, int stride, int M, int K, const int32_t* sub_off,
                                    const float* codebooks, const long long* cb_off, const float* centroid, uint8_t* codes,
                                    int code_stride, int max_ds, void* stream) {
    if (n <= 0) return 0;
    if (max_ds <= 0 || M <= 0 || K <= 0 || K > 256) return -4;
    long long bx = (n + 255) / 256;
    if (bx > 8192) bx = 8192;
    dim3 grid((unsigned)bx, (unsigned)M);
    if (max_ds > PQE_MAX_DS)
        jvb_pq_encode_wide_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(vectors, n, d, stride, M, K, sub_off, codebooks, cb_off, centroid,
                                                                         codes, code_stride);
    else
        jvb_pq_encode_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(vectors, n, d, stride, M, K, sub_off, codebooks, cb_off, centroid, codes,
                                                                    code_stride);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}


// ---------------------------------------------------------------------------------------------------------------
// PQ training (J/JVectorIndexQuantization.java:114-140: ProductQuantization.compute on a sample of the
// vectors, K = min(256, n) clusters per subspace, global centring iff EUCLIDEAN).  Lloyd's algorithm with the ASSIGN step
// = jvb_pq_encode_kernel (the canonical fmaf-chain distance, ties to the lowest index: training and encoding agree on what
// "nearest" means) and the UPDATE step below: no float atomics anywhere, so two trainings of the same data give the same
// codebooks bit for bit.
// ---------------------------------------------------------------------------------------------------------------

// column means in two deterministic steps: per-block partial sums in f64 (fixed row order), then one thread per column
__global__ __launch_bounds__(256) void jvb_col_partial_kernel(const float* __restrict__ v, long long n, int d, int stride, int slabs,
                                                              double* __restrict__ partial) {  // [slabs][d]
    const int slab = blockIdx.y;
    const long long per = (n + slabs - 1) / slabs, r0 = slab * per, r1 = r0 + per < n ? r0 + per : n;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < d; j += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (long long i = r0; i < r1; i++) acc += (double)v[(size_t)i * (size_t)stride + j];
        partial[(size_t)slab * d + j] = acc;
    }
}
__global__ void jvb_col_mean_kernel(const double* __restrict__ partial, int slabs, int d, long long n, float* __restrict__ mean) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    double acc = 0.0;
    for (int s = 0; s < slabs; s++) acc += partial[(size_t)s * d + j];
    mean[j] = (float)(acc / (double)n);
}

// gathers the training sample (rows[i] of the corpus, centred) into a dense [nt][d] matrix
__global__ __launch_bounds__(256) void jvb_gather_rows_kernel(const float* __restrict__ v, int d, int stride, const long long* __restrict__ rows,
                                                              long long nt, const float* __restrict__ centroid, float* __restrict__ out) {
    for (long long i = blockIdx.x; i < nt; i += gridDim.x) {
        const float* src = v + (size_t)rows[i] * (size_t)stride;
        for (int j = threadIdx.x; j < d; j += blockDim.x) out[(size_t)i * d + j] = centroid ? src[j] - centroid[j] : src[j];
    }
}

// UPDATE: one workgroup per (subspace m, cluster c).  Thread t accumulates the members among rows t, t + 256, ... in row
// order; the 256 partial vectors meet in a fixed pairwise tree in LDS.  Empty clusters keep their centroid.
#define PQT_TILE 128  // columns of a subspace summed per pass (wider subspaces take several passes over the members)
__global__ __launch_bounds__(256) void jvb_pq_update_kernel(const float* __restrict__ x, long long nt, int d, const int32_t* __restrict__ sub_off,
                                                            const uint8_t* __restrict__ codes, int M, int K, float* __restrict__ codebooks,
                                                            const long long* __restrict__ cb_off) {
    extern __shared__ float red[];  // [256][tw + 1] (the last column counts members)
    const int m = blockIdx.y, c = blockIdx.x;
    const int d0 = sub_off[m], ds = sub_off[m + 1] - d0;
    const int t = threadIdx.x;
    for (int j0 = 0; j0 < ds; j0 += PQT_TILE) {
        const int tw = ds - j0 < PQT_TILE ? ds - j0 : PQT_TILE;
        float* mine = red + (size_t)t * (tw + 1);
        for (int j = 0; j <= tw; j++) mine[j] = 0.0f;
        for (long long i = t; i < nt; i += 256) {
            if (codes[(size_t)i * M + m] == (uint8_t)c) {
                const float* row = x + (size_t)i * d + d0 + j0;
                for (int j = 0; j < tw; j++) mine[j] += row[j];
                mine[tw] += 1.0f;
            }
        }
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {
            if (t < w) {
                float* a_ = red + (size_t)t * (tw + 1);
                const float* b_ = red + (size_t)(t + w) * (tw + 1);
                for (int j = 0; j <= tw; j++) a_[j] += b_[j];
            }
            __syncthreads();
        }
        const float cnt = red[tw];
        if (cnt > 0.0f) {
            float* cb = codebooks + cb_off[m] + (size_t)c * ds + j0;
            for (int j = t; j < tw; j += 256) cb[j] = red[j] / cnt;
        }
        __syncthreads();
    }
}

// codebooks[m][c][:] = x[init[c]][subspace m]
__global__ void jvb_pq_init_kernel(const float* __restrict__ x, int d, const int32_t* __restrict__ sub_off, const long long* __restrict__ init,
                                   int M, int K, float* __restrict__ codebooks, const long long* __restrict__ cb_off) {
    const int m = blockIdx.y, c = blockIdx.x;
    const int d0 = sub_off[m], ds = sub_off[m + 1] - d0;
    for (int j = threadIdx.x; j < ds; j += blockDim.x) codebooks[cb_off[m] + (size_t)c * ds + j] = x[(size_t)init[(size_t)m * K + c] * d + d0 + j];
}

// k-means++ seeding (J/JVectorIndexQuantization.java:122-131 -> jvector's KMeansPlusPlusClusterer; the CPU builder's
// restatement: jv_build_cpu.cpp "k-means++ seeding", same splitmix64 stream per subspace): one 1 024-thread workgroup per
// subspace picks K sample rows one after the other, each with probability proportional to its squared distance to the nearest
// centre picked so far.  Deterministic: thread t owns rows t, t + 1024, ...; a pick is the first row IN THAT ORDER (thread-major)
// whose running sum reaches r = u * total — per-thread sums in doubles, a fixed-order workgroup scan, no atomics.  Per pick
// the subspace's nt x ds floats are read once (12 MB at nt = 128 000, ds = 24: ~100 GB for 32 subspaces x 256 picks, tens of
// milliseconds at the 32 workgroups' share of the bandwidth).
__device__ __forceinline__ unsigned long long jvb_splitmix64(unsigned long long& s) {
    unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(1024) void jvb_pq_seed_kernel(const float* __restrict__ x, long long nt, int d, const int32_t* __restrict__ sub_off,
                                                           int K, unsigned long long seed, long long* __restrict__ init,
                                                           float* __restrict__ mind_all) {
    __shared__ float s_ctr[256];
    __shared__ double s_part[1024];
    __shared__ double s_tot;
    __shared__ long long s_pick;
    const int m = blockIdx.x, t = threadIdx.x;
    const int d0 = sub_off[m], ds = sub_off[m + 1] - d0;
    float* mind = mind_all + (size_t)m * (size_t)nt;
    unsigned long long st = seed * 0x9E3779B97F4A7C15ull + (unsigned long long)m * 1315423911ull + 7ull;
    const long long kk = K < nt ? K : nt;
    long long prev = (long long)(jvb_splitmix64(st) % (unsigned long long)nt);   // (every thread runs the same stream)
    if (t == 0) init[(size_t)m * K] = prev;
    for (long long i = t; i < nt; i += 1024) mind[i] = 3.0e38f;
    for (int c = 1; c < K; c++) {
        if (c >= kk) {  // fewer points than clusters: duplicates, as the CPU builder does
            if (t == 0) init[(size_t)m * K + c] = init[(size_t)m * K + (c % kk)];
            continue;
        }
        for (int j = t; j < ds; j += 1024) s_ctr[j & 255] = x[(size_t)prev * d + d0 + j];   // (ds <= 256)
        __syncthreads();
        double mine = 0.0;
        for (long long i = t; i < nt; i += 1024) {
            const float* r = x + (size_t)i * d + d0;
            float dd = 0.0f;
            for (int j = 0; j < ds; j++) {
                const float df = r[j] - s_ctr[j];
                dd = fmaf(df, df, dd);
            }
            const float mn = fminf(mind[i], dd);
            mind[i] = mn;
            mine += (double)mn;
        }
        s_part[t] = mine;
        __syncthreads();
        // inclusive scan of the 1 024 partial sums, fixed order
        for (int off = 1; off < 1024; off <<= 1) {
            const double add = t >= off ? s_part[t - off] : 0.0;
            __syncthreads();
            s_part[t] += add;
            __syncthreads();
        }
        const double u = (double)(jvb_splitmix64(st) >> 11) * (1.0 / 9007199254740992.0);
        if (t == 0) s_tot = s_part[1023], s_pick = -1;
        __syncthreads();
        const double r_ = u * s_tot;
        const double before = t > 0 ? s_part[t - 1] : 0.0;
        // the thread whose range holds r walks its rows (rounding may push r past the total: the last row with weight then)
        if (s_part[t] >= r_ && before < r_ || (t == 1023 && s_part[t] < r_)) {
            double run = before;
            long long pick = -1, last = -1;
            for (long long i = t; i < nt; i += 1024) {
                run += (double)mind[i];
                if (mind[i] > 0.0f) last = i;
                if (run >= r_) {
                    pick = i;
                    break;
                }
            }
            if (pick < 0) pick = last >= 0 ? last : (long long)t < nt ? (long long)t : 0;
            s_pick = pick;
        }
        __syncthreads();
        long long pk = s_pick;
        if (pk < 0) pk = prev;  // (all weights zero: every remaining point coincides with a centre)
        prev = pk;
        if (t == 0) init[(size_t)m * K + c] = pk;
        __syncthreads();
    }
}

// Trains M codebooks of K centroids on the rows `rows` [nt] of `vectors` (all device pointers; sub_off [M + 1] int32,
// cb_off [M] int64, init [M][K] int64 = sample positions of the initial centroids; scratch: sample [nt][d] floats,
// sample_codes [nt][M] bytes, partial [64][d] doubles).  centroid_out (or NULL: no centring) receives the corpus mean.
// init == NULL: k-means++ seeding on the device (jvb_pq_seed_kernel; `seed`, scratch init_scratch [M][K] int64 and
// mind_scratch [M][nt] floats); else the caller's sample positions.
extern "C" int jvb_pq_train_device2(const float* vectors, long long n, int d, int stride, int M, int K, const int32_t* sub_off,
                                    const long long* cb_off, const long long* rows, long long nt, const long long* init, int iters,
                                    float* centroid_out, float* sample, uint8_t* sample_codes, double* partial, float* codebooks,
                                    int max_ds, unsigned long long seed, long long* init_scratch, float* mind_scratch, void* stream_);
extern "C" int jvb_pq_train_device(const float* vectors, long long n, int d, int stride, int M, int K, const int32_t* sub_off,
                                   const long long* cb_off, const long long* rows, long long nt, const long long* init, int iters,
                                   float* centroid_out, float* sample, uint8_t* sample_codes, double* partial, float* codebooks,
                                   int max_ds, void* stream_) {
    if (!init) return -4;
    return jvb_pq_train_device2(vectors, n, d, stride, M, K, sub_off, cb_off, rows, nt, init, iters, centroid_out, sample, sample_codes, partial,
                                codebooks, max_ds, 0ull, nullptr, nullptr, stream_);
}
extern "C" int jvb_pq_train_device2(const float* vectors, long long n, int d, int stride, int M, int K, const int32_t* sub_off,
                                    const long long* cb_off, const long long* rows, long long nt, const long long* init, int iters,
                                    float* centroid_out, float* sample, uint8_t* sample_codes, double* partial, float* codebooks,
                                    int max_ds, unsigned long long seed, long long* init_scratch, float* mind_scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || nt <= 0 || M <= 0 || K <= 0 || K > 256 || max_ds <= 0) return -4;
    if (!init && (!init_scratch || !mind_scratch || max_ds > 256)) return -4;
    if (centroid_out) {
        const int slabs = 64;
        jvb_col_partial_kernel<<<dim3((unsigned)((d + 255) / 256), slabs), 256, 0, stream>>>(vectors, n, d, stride, slabs, partial);
        jvb_col_mean_kernel<<<(d + 255) / 256, 256, 0, stream>>>(partial, slabs, d, n, centroid_out);
    }
    jvb_gather_rows_kernel<<<(unsigned)(nt < 65535 ? nt : 65535), 256, 0, stream>>>(vectors, d, stride, rows, nt, centroid_out, sample);
    if (!init) {
        jvb_pq_seed_kernel<<<(unsigned)M, 1024, 0, stream>>>(sample, nt, d, sub_off, K, seed, init_scratch, mind_scratch);
        init = init_scratch;
    }
    jvb_pq_init_kernel<<<dim3((unsigned)K, (unsigned)M), 64, 0, stream>>>(sample, d, sub_off, init, M, K, codebooks, cb_off);
    const size_t lds = (size_t)256 * (size_t)((max_ds < PQT_TILE ? max_ds : PQT_TILE) + 1) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)jvb_pq_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -3;
        attr_set = true;
    }
    for (int it = 0; it < iters; it++) {
        // ASSIGN: the encoder on the (already centred) sample
        const int rc = jvb_pq_encode_device(sample, nt, d, d, M, K, sub_off, codebooks, cb_off, nullptr, sample_codes, M, max_ds, stream_);
        if (rc != 0) return rc;
        jvb_pq_update_kernel<<<dim3((unsigned)K, (unsigned)M), 256, lds, stream>>>(sample, nt, d, sub_off, sample_codes, M, K, codebooks, cb_off);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
