// jv_kernels_pqw12.hip — the several-waves-per-query pool kernel (jv_pqw_body.h) with TWELVE waves per query: PQ-192, the
// reference's default subspace count for 768-d .. 1 536-d fields (J/JVectorIndexQuantization.java:428-446), and with EIGHT: PQ-128,
// the default for 512-d fields.  A translation unit
// of its own (the instances compile in parallel with jv_kernels_pqw.hip's).  One workgroup of 768 threads per CU: eight of a
// wave's 16 table rows live in LDS (96 KB), eight in registers.  gfx950 / CDNA4.
#include "jv_pqw_body.h"

typedef void (*pqw_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQW12_ROW(CAPK) \
    { jv_search_pqw_kernel<0, CAPK, 12, 4, 8>, jv_search_pqw_kernel<2, CAPK, 12, 4, 8>, jv_search_pqw_kernel<12, CAPK, 12, 4, 8>, jv_search_pqw_kernel<24, CAPK, 12, 4, 8> }
// [capacity class 0..2][nch slot]
static const pqw_kernel_t g_pqw12_kernels[3][4] = {JV_PQW12_ROW(0), JV_PQW12_ROW(1), JV_PQW12_ROW(2)};
// eight waves per query: PQ-128, the default for 512-d fields (d / 4); same split of the rows, two workgroups per CU
#define JV_PQW8_ROW(CAPK) \
    { jv_search_pqw_kernel<0, CAPK, 8, 4, 8>, jv_search_pqw_kernel<2, CAPK, 8, 4, 8>, jv_search_pqw_kernel<12, CAPK, 8, 4, 8>, jv_search_pqw_kernel<24, CAPK, 8, 4, 8> }
static const pqw_kernel_t g_pqw8_kernels[3][4] = {JV_PQW8_ROW(0), JV_PQW8_ROW(1), JV_PQW8_ROW(2)};

extern "C" const void* jvk_pqw12_kernel(int waves, int capk, int nch_slot) { return (const void*)(waves == 8 ? g_pqw8_kernels : g_pqw12_kernels)[capk][nch_slot]; }
extern "C" hipError_t jvk_pqw12_set_max_lds(int bytes) {
    for (int c = 0; c < 3; c++)
        for (int s = 0; s < 4; s++) {
            hipError_t e = hipFuncSetAttribute((const void*)g_pqw12_kernels[c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)g_pqw8_kernels[c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

// the device-resident query server for these shapes (jv_serve_pqw_kernel: one-query calls answered from the ring)
#define JV_PQS12_ROW(CAPK, W) \
    { jv_serve_pqw_kernel<0, CAPK, W, 4, 8>, jv_serve_pqw_kernel<2, CAPK, W, 4, 8>, jv_serve_pqw_kernel<12, CAPK, W, 4, 8>, jv_serve_pqw_kernel<24, CAPK, W, 4, 8> }
static const pqw_kernel_t g_pqs12_kernels[2][3][4] = {{JV_PQS12_ROW(0, 8), JV_PQS12_ROW(1, 8), JV_PQS12_ROW(2, 8)},
                                                      {JV_PQS12_ROW(0, 12), JV_PQS12_ROW(1, 12), JV_PQS12_ROW(2, 12)}};
extern "C" const void* jvk_pqs12_kernel(int waves, int capk, int nch_slot) { return (const void*)g_pqs12_kernels[waves == 8 ? 0 : 1][capk][nch_slot]; }
extern "C" hipError_t jvk_pqs12_set_max_lds(int bytes) {
    for (int w = 0; w < 2; w++)
        for (int c = 0; c < 3; c++)
            for (int s = 0; s < 4; s++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_pqs12_kernels[w][c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}
