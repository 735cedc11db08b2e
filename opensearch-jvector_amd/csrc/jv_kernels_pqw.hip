// jv_kernels_pqw.hip — instances and launchers of the several-waves-per-query persistent pool kernel (jv_pqw_body.h):
// PQ-32 (two waves per query) and PQ-64 (four), unfiltered, flat graph, L2 / dot product.  gfx950 / CDNA4.
#include "jv_pqw_body.h"

typedef void (*pqw_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQW_ROW(CAPK, W, OCC, NL) \
    { jv_search_pqw_kernel<0, CAPK, W, OCC, NL>, jv_search_pqw_kernel<2, CAPK, W, OCC, NL>, jv_search_pqw_kernel<12, CAPK, W, OCC, NL>, jv_search_pqw_kernel<24, CAPK, W, OCC, NL> }
// [variant * 2 + (PQ-64 ? 1 : 0)][capacity class 0..2][nch slot]
//   variant 0 (throughput): 4 of a wave's 16 table rows in LDS, 12 in registers -> 8 workgroups of two waves per CU;
//   variant 1 (latency): the whole table in LDS (plain gathers: a third of the scoring pass's instructions) -> 3 workgroups
//   per CU: launches with few queries, and the device-resident query server
static const pqw_kernel_t g_pqw_kernels[4][3][4] = {
    {JV_PQW_ROW(0, 2, 4, 4), JV_PQW_ROW(1, 2, 4, 4), JV_PQW_ROW(2, 2, 4, 4)},
    {JV_PQW_ROW(0, 4, 4, 4), JV_PQW_ROW(1, 4, 4, 4), JV_PQW_ROW(2, 4, 4, 4)},
    {JV_PQW_ROW(0, 2, 2, 16), JV_PQW_ROW(1, 2, 2, 16), JV_PQW_ROW(2, 2, 2, 16)},  // (<= 3 workgroups per CU: two waves per SIMD, 256 registers — no scratch)
    {JV_PQW_ROW(0, 4, 3, 16), JV_PQW_ROW(1, 4, 3, 16), JV_PQW_ROW(2, 4, 3, 16)},
};

// Diagnostic instance (JV_PQW_OCC5=1, VERDICT r4 #1a): the register budget of FIVE waves per SIMD (96 VGPRs) with the whole table in
// registers (NL must be a multiple of 4 — the register rows are permuted four at a time — and NL = 4 leaves 19.8 KB of LDS per query:
// eight per CU again), so that nine two-wave workgroups fit a CU instead of eight — d = 768, PQ-32, pools beyond 1 024 entries only.
// tools/kernel_resources.py: 637 VGPR spills / 436 B of scratch per lane against 43 / 168 of the 128-register instance.
static const pqw_kernel_t g_pqw_occ5 = jv_search_pqw_kernel<12, 2, 2, 5, 0>;
static bool pqw_occ5_ok(const JvIndexDev* ix, int cap);

static int pqw_nch_slot(const JvIndexDev* ix) {
    if (ix->nvq_M > 0) return 0;  // the NVQ decoder lives in the "any d" instances only (score_rows)
    if (ix->sim == 2) return 0;   // ... and so does cosine (jv_pqw_body.h COSI)
    if (ix->stride != ix->nch * 64) return 0;
    return ix->nch == 2 ? 1 : ix->nch == 12 ? 2 : ix->nch == 24 ? 3 : 0;
}
static int pqw_capk(int cap) { return cap <= 512 ? 0 : cap <= 1024 ? 1 : 2; }
extern "C" int jvk_pqw_waves(const JvIndexDev* ix) { return ix->pq_M / 16; }
extern "C" const void* jvk_pqw12_kernel(int waves, int capk, int nch_slot);  // jv_kernels_pqw12.hip: twelve / eight waves per query (PQ-192 / PQ-128)
extern "C" hipError_t jvk_pqw12_set_max_lds(int bytes);
// NL of the instances: table rows per wave kept in LDS (PQ-192: eight, both variants)
extern "C" int jvk_pqw_lds_rows(const JvIndexDev* ix, int variant) { return ix->pq_M >= 128 ? 8 : (variant == 1 ? 16 : (variant == 2 ? 0 : 4)); }
// shapes this kernel runs: one wave per 16-subspace chunk, one lane per stored neighbour, whole log groups per 64-entry chunk
extern "C" int jvk_pqw_ok(const JvIndexDev* ix, int cap) {
    if (!(ix->pq_M == 32 || ix->pq_M == 64 || ix->pq_M == 128 || ix->pq_M == 192) || (ix->sim == 2 && !ix->pq_fused_norm) || !ix->pq_fused || ix->num_upper != 0) return 0;
    if (ix->R < 1 || ix->R > JV_WAVE || 64 % ((JV_WAVE / ix->R) * 8) != 0) return 0;
    return cap <= 2048 && ix->n < (1 << 30) ? 1 : 0;
}
static bool pqw_occ5_ok(const JvIndexDev* ix, int cap) { return ix->pq_M == 32 && pqw_capk(cap) == 2 && pqw_nch_slot(ix) == 2; }
extern "C" int jvk_pqw_occ5_ok(const JvIndexDev* ix, int cap) { return pqw_occ5_ok(ix, cap) ? 1 : 0; }
static pqw_kernel_t pqw_pick(const JvIndexDev* ix, int cap, int variant) {
    if (variant == 2) return g_pqw_occ5;
    if (ix->pq_M >= 128) return (pqw_kernel_t)jvk_pqw12_kernel(ix->pq_M / 16, pqw_capk(cap), pqw_nch_slot(ix));
    return g_pqw_kernels[(variant ? 2 : 0) + (ix->pq_M == 64 ? 1 : 0)][pqw_capk(cap)][pqw_nch_slot(ix)];
}

extern "C" hipError_t jvk_pqw_set_max_lds(int bytes) {
    {
        hipError_t e = jvk_pqw12_set_max_lds(bytes);
        if (e != hipSuccess) return e;
    }
    {
        hipError_t e = hipFuncSetAttribute((const void*)g_pqw_occ5, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
    }
    for (int w = 0; w < 4; w++)
        for (int c = 0; c < 3; c++)
            for (int s = 0; s < 4; s++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_pqw_kernels[w][c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

// resident workgroups per CU for this index shape, pool capacity and LDS size
extern "C" int jvk_pqw_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes, int variant) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pqw_pick(ix, cap, variant), JV_WAVE * jvk_pqw_waves(ix), (size_t)lds_bytes) != hipSuccess) return 1;
    return nb < 1 ? 1 : nb;
}

// blocks = resident workgroups (the host sizes the log scratch to it); a->cand_cap = pool entries
extern "C" hipError_t jvk_launch_search_pqw(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, int variant, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    pqw_pick(ix, a->cand_cap, variant)<<<blocks, JV_WAVE * jvk_pqw_waves(ix), lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}
