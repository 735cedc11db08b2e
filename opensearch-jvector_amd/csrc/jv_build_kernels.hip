// jv_build_kernels.hip — GPU helpers of the WRITE side (index construction; not the graded hot path,
// see jv_build.h / builder_gpu.py).  jvb_robust_prune_device: jvector-style diversity selection
// (GraphIndexBuilder's retainDiverse: alpha sweeps 1.0, 1.2, .. <= alpha; a candidate is kept unless an
// already-selected neighbour s has sim(c, s) > sim(c, centre) * a), one wavefront per row, the row's
// candidate-candidate similarity matrix staged in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64

__global__ __launch_bounds__(WAVE) void jvb_robust_prune_kernel(const float* __restrict__ scc,   // [S][Lc][Lc]
                                                                const float* __restrict__ sc,    // [S][Lc] desc
                                                                const int32_t* __restrict__ cd,  // [S][Lc]
                                                                const uint8_t* __restrict__ valid,  // [S][Lc]
                                                                int S, int Lc, int R, float alpha,
                                                                int32_t* __restrict__ sel,   // [S][R]
                                                                int32_t* __restrict__ nsel_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* m = (float*)smem;                      // [Lc][Lc]
    float* scl = m + (size_t)Lc * Lc;             // [Lc]
    int32_t* sel_idx = (int32_t*)(scl + Lc);      // [R]
    uint8_t* state = (uint8_t*)(sel_idx + R);     // [Lc]: 0 = not a candidate, 1 = open, 2 = taken
    const int lane = threadIdx.x;
    const int row = blockIdx.x;
    if (row >= S) return;
    const float* mg = scc + (size_t)row * Lc * Lc;
    for (int i = lane; i < Lc * Lc; i += WAVE) m[i] = mg[i];
    for (int i = lane; i < Lc; i += WAVE) {
        scl[i] = sc[(size_t)row * Lc + i];
        state[i] = valid[(size_t)row * Lc + i] ? 1 : 0;
    }
    __syncthreads();
    int nsel = 0;
    for (float a = 1.0f; a <= alpha + 1e-6f && nsel < R; a += 0.2f) {
        for (int c = 0; c < Lc && nsel < R; c++) {
            if (state[c] != 1) continue;
            const float thr = scl[c] * a;
            bool bad = false;
            for (int s0 = 0; s0 < nsel; s0 += WAVE) {
                const int s = s0 + lane;
                const bool over = s < nsel && m[(size_t)c * Lc + sel_idx[s]] > thr;
                if (__ballot(over)) {
                    bad = true;
                    break;
                }
            }
            if (!bad) {
                if (lane == 0) {
                    state[c] = 2;
                    sel_idx[nsel] = c;
                }
                nsel++;
                __syncthreads();
            }
        }
    }
    for (int i = lane; i < R; i += WAVE) sel[(size_t)row * R + i] = i < nsel ? cd[(size_t)row * Lc + sel_idx[i]] : -1;
    if (lane == 0) nsel_out[row] = nsel;
}

extern "C" int jvb_robust_prune_device(const float* scc, const float* sc, const int32_t* cd, const uint8_t* valid,
                                       int S, int Lc, int R, float alpha, int32_t* sel, int32_t* nsel,
                                       void* stream) {
    if (S <= 0) return 0;
    size_t lds = (size_t)Lc * Lc * 4 + (size_t)Lc * 4 + (size_t)R * 4 + (size_t)Lc;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return -4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)jvb_robust_prune_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return -3;
        attr_set = true;
    }
    jvb_robust_prune_kernel<<<S, WAVE, lds, (hipStream_t)stream>>>(scc, sc, cd, valid, S, Lc, R, alpha, sel, nsel);
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

// Trains M codebooks of K centroids on the rows `rows` [nt] of `vectors` (all device pointers; sub_off [M + 1] int32,
// cb_off [M] int64, init [M][K] int64 = sample positions of the initial centroids; scratch: sample [nt][d] floats,
// sample_codes [nt][M] bytes, partial [64][d] doubles).  centroid_out (or NULL: no centring) receives the corpus mean.
extern "C" int jvb_pq_train_device(const float* vectors, long long n, int d, int stride, int M, int K, const int32_t* sub_off,
                                   const long long* cb_off, const long long* rows, long long nt, const long long* init, int iters,
                                   float* centroid_out, float* sample, uint8_t* sample_codes, double* partial, float* codebooks,
                                   int max_ds, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || nt <= 0 || M <= 0 || K <= 0 || K > 256 || max_ds <= 0) return -4;
    if (centroid_out) {
        const int slabs = 64;
        jvb_col_partial_kernel<<<dim3((unsigned)((d + 255) / 256), slabs), 256, 0, stream>>>(vectors, n, d, stride, slabs, partial);
        jvb_col_mean_kernel<<<(d + 255) / 256, 256, 0, stream>>>(partial, slabs, d, n, centroid_out);
    }
    jvb_gather_rows_kernel<<<(unsigned)(nt < 65535 ? nt : 65535), 256, 0, stream>>>(vectors, d, stride, rows, nt, centroid_out, sample);
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
