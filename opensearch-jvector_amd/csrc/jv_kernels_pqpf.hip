// jv_kernels_pqpf.hip — the persistent pool kernel's instances WITH a doc filter (template in jv_pqp_body.h; FILT = true):
// filtered searches whose pool (~ rerankK / selectivity entries) outgrows round 1's filtered kernel (960 entries).
// The accept lambda is J/JVectorReader.java:157-163 (ord -> doc -> Bits.get), applied where jvector applies it: a
// popped candidate is admitted to the result queue only if accepted, every node is traversed.
#include "jv_pqp_body.h"

typedef void (*pqp_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQPF_ROW(FAST, CAPK) \
    { jv_search_pqp_kernel<0, 1, FAST, CAPK, false, true>, jv_search_pqp_kernel<2, 1, FAST, CAPK, false, true>, \
      jv_search_pqp_kernel<12, 1, FAST, CAPK, false, true>, jv_search_pqp_kernel<24, 1, FAST, CAPK, false, true> }
// [FAST][capacity class 1..3][nch slot]
static const pqp_kernel_t g_pqpf_kernels[2][3][4] = {{JV_PQPF_ROW(false, 1), JV_PQPF_ROW(false, 2), JV_PQPF_ROW(false, 3)},
                                                     {JV_PQPF_ROW(true, 1), JV_PQPF_ROW(true, 2), JV_PQPF_ROW(true, 3)}};
// register-table variants (PQ-32, FAST): [capacity class 1..2][nch slot]
#define JV_PQVF_ROW(CAPK) \
    { jv_search_pqp_kernel<0, 1, true, CAPK, true, true>, jv_search_pqp_kernel<2, 1, true, CAPK, true, true>, \
      jv_search_pqp_kernel<12, 1, true, CAPK, true, true>, jv_search_pqp_kernel<24, 1, true, CAPK, true, true> }
static const pqp_kernel_t g_pqvf_kernels[2][4] = {JV_PQVF_ROW(1), JV_PQVF_ROW(2)};

// fast: pq_M % 16 == 0 and not cosine; capk: capacity class (0..3, class 0 runs on class 1's instance); lutr: table in registers
extern "C" const void* jvk_pqpf_kernel(int fast, int capk, int nch_slot, int lutr) {
    if (capk < 1) capk = 1;
    if (lutr && fast && capk <= 2) return (const void*)g_pqvf_kernels[capk - 1][nch_slot];
    return (const void*)g_pqpf_kernels[fast ? 1 : 0][capk - 1][nch_slot];
}

extern "C" hipError_t jvk_pqpf_set_max_lds(int bytes) {
    for (int f = 0; f < 2; f++)
        for (int c = 0; c < 3; c++)
            for (int s = 0; s < 4; s++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_pqpf_kernels[f][c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess && f == 0 && c < 2)
                    e = hipFuncSetAttribute((const void*)g_pqvf_kernels[c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}
