// jv_kernels_pqpf.hip — the persistent pool kernel's instances WITH a doc filter (template in jv_pqp_body.h; FILT = true):
// filtered searches whose pool (~ rerankK / selectivity entries) outgrows round 1's filtered kernel (960 entries).
// The accept lambda is J/JVectorReader.java:157-163 (ord -> doc -> Bits.get), applied where jvector applies it: a
// popped candidate is admitted to the result queue only if accepted, every node is traversed.
#include "jv_pqp_body.h"

typedef void (*pqp_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQPF_ROW(FAST, CAPK) \
    { jv_search_pqp_kernel<0, 1, FAST, CAPK, false, true>, jv_search_pqp_kernel<2, 1, FAST, CAPK, false, true>, \
      jv_search_pqp_kernel<12, 1, FAST, CAPK, false, true>, jv_search_pqp_kernel<24, 1, FAST, CAPK, false, true> }
// [FAST][capacity class 1..5][nch slot]  (classes 4, 5 = up to 8 192 / 16 384 entries: selective filters at wide beams, 2 / 1 workgroups per CU)
static const pqp_kernel_t g_pqpf_kernels[2][5][4] = {{JV_PQPF_ROW(false, 1), JV_PQPF_ROW(false, 2), JV_PQPF_ROW(false, 3), JV_PQPF_ROW(false, 4), JV_PQPF_ROW(false, 5)},
                                                     {JV_PQPF_ROW(true, 1), JV_PQPF_ROW(true, 2), JV_PQPF_ROW(true, 3), JV_PQPF_ROW(true, 4), JV_PQPF_ROW(true, 5)}};
// register-table variants (PQ-32, FAST): [capacity class 1..5][nch slot]  (classes 3..5: the pool stays in LDS after the search)
#define JV_PQVF_ROW(CAPK) \
    { jv_search_pqp_kernel<0, 1, true, CAPK, true, true>, jv_search_pqp_kernel<2, 1, true, CAPK, true, true>, \
      jv_search_pqp_kernel<12, 1, true, CAPK, true, true>, jv_search_pqp_kernel<24, 1, true, CAPK, true, true> }
static const pqp_kernel_t g_pqvf_kernels[5][4] = {JV_PQVF_ROW(1), JV_PQVF_ROW(2), JV_PQVF_ROW(3), JV_PQVF_ROW(4), JV_PQVF_ROW(5)};

// fast: pq_M % 16 == 0 and not cosine; capk: capacity class (0..5, class 0 runs on class 1's instance); lutr: table in registers
extern "C" const void* jvk_pqpf_kernel(int fast, int capk, int nch_slot, int lutr) {
    if (capk < 1) capk = 1;
    if (lutr && fast) return (const void*)g_pqvf_kernels[capk - 1][nch_slot];
    return (const void*)g_pqpf_kernels[fast ? 1 : 0][capk - 1][nch_slot];
}

extern "C" hipError_t jvk_pqpf_set_max_lds(int bytes) {
    for (int f = 0; f < 2; f++)
        for (int c = 0; c < 5; c++)
            for (int s = 0; s < 4; s++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_pqpf_kernels[f][c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess && f == 0)
                    e = hipFuncSetAttribute((const void*)g_pqvf_kernels[c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

// A batch-wide doc filter translated to ORDINAL space once per launch (bit ord = acceptDocs.get(ord2doc[ord]), the reference's
// acceptOrds lambda, J/JVectorReader.java:157-163): inside the search the accept bit of a neighbour is then ONE load that can
// be issued before the next block's prefetch, instead of two dependent loads behind it (vmcnt is in-order: waiting for the
// younger accept word drained the prefetch on every expansion).  n * 4 B of ord2doc per launch: microseconds.
__global__ __launch_bounds__(256) void jv_accept_to_ord_kernel(const int32_t* __restrict__ ord2doc, int n, const uint64_t* __restrict__ accept,
                                                               long long accept_docs, uint64_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool acc = false;
    if (i < n) {
        const int doc = ord2doc ? ord2doc[i] : i;
        acc = doc >= 0 && (long long)doc < accept_docs && ((accept[doc >> 6] >> (doc & 63)) & 1ull);
    }
    const unsigned long long m = __ballot(acc);
    if ((threadIdx.x & 63) == 0 && i < n) out[i >> 6] = m;
}
extern "C" hipError_t jvk_launch_accept_to_ord(const JvIndexDev* ix, const uint64_t* accept, long long accept_docs, uint64_t* out, hipStream_t stream) {
    if (ix->n <= 0) return hipSuccess;
    jv_accept_to_ord_kernel<<<(ix->n + 255) / 256, 256, 0, stream>>>(ix->ord2doc, ix->n, accept, accept_docs, out);
    return hipGetLastError();
}
