// jv_kernels_pqwf.hip — the several-waves-per-query pool kernel WITH a doc filter (jv_pqw_body.h, FILT = true; round 4):
// filtered fused-PQ searches of PQ-32 / PQ-64 indexes (two / four waves per query, two fused blocks per scoring pass) on
// every pool class up to 16 384 entries (~ rerankK / selectivity).  The accept lambda is J/JVectorReader.java:157-163.
// gfx950 / CDNA4.
#include "jv_pqw_body.h"

typedef void (*pqwf_kernel_t)(const JvIndexDev, const JvSearchArgs);
// pools of up to 2 048 entries keep up to six workgroups resident per CU (three waves per SIMD, four of a wave's table rows in
// LDS); wider pools at most four, and what a CU keeps resident of them is decided by the pool's bytes alone: the whole table
// in registers (NL = 0; two waves per SIMD: 256 registers, nothing spills), LDS = pool + 1.3 KB
#define JV_PQWF_ROW(CAPK, W, OCC, NL) \
    { jv_search_pqw_kernel<0, CAPK, W, OCC, NL, true>, jv_search_pqw_kernel<2, CAPK, W, OCC, NL, true>, jv_search_pqw_kernel<12, CAPK, W, OCC, NL, true>, jv_search_pqw_kernel<24, CAPK, W, OCC, NL, true> }
// [PQ-64 ? 1 : 0][capacity class 1..5][nch slot]
static const pqwf_kernel_t g_pqwf_kernels[2][5][4] = {
    {JV_PQWF_ROW(1, 2, 3, 4), JV_PQWF_ROW(2, 2, 3, 4), JV_PQWF_ROW(3, 2, 2, 0), JV_PQWF_ROW(4, 2, 2, 0), JV_PQWF_ROW(5, 2, 2, 0)},
    {JV_PQWF_ROW(1, 4, 4, 4), JV_PQWF_ROW(2, 4, 4, 4), JV_PQWF_ROW(3, 4, 3, 0), JV_PQWF_ROW(4, 4, 2, 0), JV_PQWF_ROW(5, 4, 2, 0)},
};
// table rows a wave keeps in LDS for a pool of `cap` entries (PQ-128 / PQ-192: always eight, jv_kernels_pqw12f.hip)
extern "C" int jvk_pqwf_lds_rows(const JvIndexDev* ix, int cap) { return ix->pq_M >= 128 ? 8 : (cap <= 2048 ? 4 : 0); }
extern "C" const void* jvk_pqw12f_kernel(int waves, int capk, int nch);
extern "C" hipError_t jvk_pqw12f_set_max_lds(int bytes);

static int pqwf_nch_slot(const JvIndexDev* ix) {
    if (ix->nvq_M > 0) return 0;  // the NVQ decoder lives in the "any d" instances only (score_rows)
    if (ix->sim == 2) return 0;   // ... and so does cosine (jv_pqw_body.h COSI)
    if (ix->stride != ix->nch * 64) return 0;
    return ix->nch == 2 ? 1 : ix->nch == 12 ? 2 : ix->nch == 24 ? 3 : 0;
}
static int pqwf_capk(int cap) { return cap <= 1024 ? 1 : cap <= 2048 ? 2 : cap <= 4096 ? 3 : cap <= 8192 ? 4 : 5; }
extern "C" int jvk_pqwf_max_entries(const JvIndexDev* ix) { return ix->pq_M >= 128 ? 4096 : 16384; }
// shapes this kernel runs (jvk_pqw_ok's, with the filtered key's 29 ordinal bits)
extern "C" int jvk_pqwf_ok(const JvIndexDev* ix, int cap) {
    if (!(ix->pq_M == 32 || ix->pq_M == 64 || ix->pq_M == 128 || ix->pq_M == 192) || (ix->sim == 2 && !ix->pq_fused_norm) || !ix->pq_fused || ix->num_upper != 0) return 0;
    if (ix->R < 1 || ix->R > JV_WAVE || 64 % ((JV_WAVE / ix->R) * 8) != 0) return 0;
    if (ix->pq_M >= 128 && ix->nvq_M > 0) return 0;  // (the default-codec instances carry no NVQ decoder)
    return cap <= jvk_pqwf_max_entries(ix) && ix->n < (1 << 29) ? 1 : 0;
}
static pqwf_kernel_t pqwf_pick(const JvIndexDev* ix, int cap) {
    if (ix->pq_M >= 128) return (pqwf_kernel_t)jvk_pqw12f_kernel(ix->pq_M / 16, pqwf_capk(cap), (ix->stride == ix->nch * 64 && ix->sim != 2) ? ix->nch : 0);
    return g_pqwf_kernels[ix->pq_M == 64 ? 1 : 0][pqwf_capk(cap) - 1][pqwf_nch_slot(ix)];
}

extern "C" hipError_t jvk_pqwf_set_max_lds(int bytes) {
    {
        hipError_t e = jvk_pqw12f_set_max_lds(bytes);
        if (e != hipSuccess) return e;
    }
    for (int w = 0; w < 2; w++)
        for (int c = 0; c < 5; c++)
            for (int s = 0; s < 4; s++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_pqwf_kernels[w][c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

extern "C" int jvk_pqwf_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pqwf_pick(ix, cap), JV_WAVE * (ix->pq_M / 16), (size_t)lds_bytes) != hipSuccess) return 1;
    return nb < 1 ? 1 : nb;
}

// blocks = resident workgroups (the host sizes the log scratch to it); a->cand_cap = pool entries
extern "C" hipError_t jvk_launch_search_pqwf(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    pqwf_pick(ix, a->cand_cap)<<<blocks, JV_WAVE * (ix->pq_M / 16), lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}
