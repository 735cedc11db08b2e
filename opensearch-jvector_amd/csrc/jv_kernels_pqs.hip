// jv_kernels_pqs.hip — instances and launcher of the device-resident query server (jv_serve_pqw_kernel in jv_pqw_body.h):
// the several-waves pool kernel fed by single queries from a ring of host-visible slots.  gfx950 / CDNA4.
#include "jv_pqw_body.h"

typedef void (*pqs_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQS_ROW(CAPK, W, OCC, NL) \
    { jv_serve_pqw_kernel<0, CAPK, W, OCC, NL>, jv_serve_pqw_kernel<2, CAPK, W, OCC, NL>, jv_serve_pqw_kernel<12, CAPK, W, OCC, NL>, jv_serve_pqw_kernel<24, CAPK, W, OCC, NL> }
// [0 -> two waves per query (PQ-32), 1 -> four (PQ-64)][capacity class 0..2][nch slot]; the latency variant of
// jv_kernels_pqw.hip (whole table in LDS): a server query's time is its caller's latency
static const pqs_kernel_t g_pqs_kernels[2][3][4] = {
    {JV_PQS_ROW(0, 2, 2, 16), JV_PQS_ROW(1, 2, 2, 16), JV_PQS_ROW(2, 2, 2, 16)},  // (two workgroups per CU: registers to spare — a resident grid that needs no scratch never makes the runtime move scratch between queues)
    {JV_PQS_ROW(0, 4, 2, 16), JV_PQS_ROW(1, 4, 2, 16), JV_PQS_ROW(2, 4, 2, 16)},
};

static int pqs_nch_slot(const JvIndexDev* ix) {
    if (ix->nvq_M > 0) return 0;
    if (ix->sim == 2) return 0;   // (cosine lives in the "any d" instances: jv_pqw_body.h COSI)
    if (ix->stride != ix->nch * 64) return 0;
    return ix->nch == 2 ? 1 : ix->nch == 12 ? 2 : ix->nch == 24 ? 3 : 0;
}
static int pqs_capk(int cap) { return cap <= 512 ? 0 : cap <= 1024 ? 1 : 2; }
extern "C" const void* jvk_pqs12_kernel(int waves, int capk, int nch_slot);  // jv_kernels_pqw12.hip: eight / twelve waves per query
extern "C" hipError_t jvk_pqs12_set_max_lds(int bytes);
static pqs_kernel_t pqs_pick(const JvIndexDev* ix, int cap) {
    if (ix->pq_M >= 128) return (pqs_kernel_t)jvk_pqs12_kernel(ix->pq_M / 16, pqs_capk(cap), pqs_nch_slot(ix));
    return g_pqs_kernels[ix->pq_M == 64 ? 1 : 0][pqs_capk(cap)][pqs_nch_slot(ix)];
}

extern "C" hipError_t jvk_pqs_set_max_lds(int bytes) {
    {
        hipError_t e = jvk_pqs12_set_max_lds(bytes);
        if (e != hipSuccess) return e;
    }
    for (int w = 0; w < 2; w++)
        for (int c = 0; c < 3; c++)
            for (int s = 0; s < 4; s++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_pqs_kernels[w][c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

extern "C" int jvk_pqs_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pqs_pick(ix, cap), JV_WAVE * (ix->pq_M / 16), (size_t)lds_bytes) != hipSuccess) return 1;
    return nb < 1 ? 1 : nb;
}

// a->cand_cap = the LARGEST pool a request may need (the LDS plan and the capacity class follow it)
extern "C" hipError_t jvk_launch_serve_pqw(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, hipStream_t stream) {
    pqs_pick(ix, a->cand_cap)<<<blocks, JV_WAVE * (ix->pq_M / 16), lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}

// ---- one-query calls WITH a doc filter on the several-waves kernel (round 4; the one-wave filtered server is jv_kernels_pqsf.hip) ----
// pool class 4 (<= 8 192 entries: the host sizes the pool to what keeps two queries resident per CU next to the table in LDS)
#define JV_PQSWF_ROW(W, OCC) \
    { jv_serve_pqw_kernel<0, 4, W, OCC, 16, true>, jv_serve_pqw_kernel<2, 4, W, OCC, 16, true>, jv_serve_pqw_kernel<12, 4, W, OCC, 16, true>, jv_serve_pqw_kernel<24, 4, W, OCC, 16, true> }
static const pqs_kernel_t g_pqswf_kernels[2][4] = {JV_PQSWF_ROW(2, 2), JV_PQSWF_ROW(4, 2)};
static pqs_kernel_t pqswf_pick(const JvIndexDev* ix) { return g_pqswf_kernels[ix->pq_M == 64 ? 1 : 0][pqs_nch_slot(ix)]; }
extern "C" int jvk_pqswf_max_entries(void) { return 8192; }
extern "C" hipError_t jvk_pqswf_set_max_lds(int bytes) {
    for (int w = 0; w < 2; w++)
        for (int s = 0; s < 4; s++) {
            hipError_t e = hipFuncSetAttribute((const void*)g_pqswf_kernels[w][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}
extern "C" int jvk_pqswf_blocks_per_cu(const JvIndexDev* ix, int lds_bytes) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pqswf_pick(ix), JV_WAVE * (ix->pq_M / 16), (size_t)lds_bytes) != hipSuccess) return 1;
    return nb < 1 ? 1 : nb;
}
// a->cand_cap = the server's pool (4 097 .. 8 192 entries)
extern "C" hipError_t jvk_launch_serve_pqwf(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, hipStream_t stream) {
    pqswf_pick(ix)<<<blocks, JV_WAVE * (ix->pq_M / 16), lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}
