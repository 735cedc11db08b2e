// jv_kernels_pqp.hip — the headline kernel: PQ approximate search on the FUSED layout, sorted candidate pool in LDS,
// persistent grid; followed by jvector's visitedCount reconstruction and the exact rerank (gfx950 / CDNA4).
//
// What it computes is GraphSearcher.search as called from J/JVectorReader.java:165-173 for the
// case "PQ provider, no filter, threshold <= 0, flat graph"; results, scores and the four counters are bit-equal to
// oracle/jv_oracle.c (tests/test_gpu_parity.py).  Everything the kernel cannot hold is flagged and re-run by the
// generic ladder (jv_kernels.hip), so exactness never depends on a capacity.
//
// One wave runs one query.  A lone wave per SIMD is bound by dependent-instruction latency, not by bytes: what an
// expansion costs is the length of its chain of dependent steps.  Measured alternatives:
//   * round 1 (jv_search_pqf_kernel): LDS pool, but every expansion touched the WHOLE pool — masks in scalar registers
//     over all 64-entry chunks, a full read-shift-write pass, a full pass to rebuild the masks: 5 100 cycles per
//     expansion at 256 entries, ~20 000 at 1 024 (rerankK = 900);
//   * pool in registers (measured this round, then removed): insertion by DPP wave shifts is cheap, but the kernel keeps no visited
//     set while searching, so ~20 of the 32 neighbours of an expansion are re-encounters that must be recognised in
//     the pool — one at a time through scalar compares (registers cannot be indexed per lane): 340 cycles each;
//   * this kernel: the pool stays in LDS so that all 32 neighbours find their rank (and their duplicates) at once
//     with a 3-4 level 8-ary search per LANE, and everything else only touches the chunks it changes: the best
//     unexpanded entry is looked for from a moving lower bound, a merge shifts only the chunks behind the first
//     insertion point (uniform shift once past the last one), the boundary test reads one chunk.
// LDS per query = look-up table + pool; the expansion log goes to a per-workgroup HBM scratch (persistent grid: one
// workgroup per resident LDS slot, queries are dequeued with an atomic counter).
#include "jv_pqp_body.h"

typedef void (*pqp_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQP_ROW(NP, FAST, CAPK) \
    { jv_search_pqp_kernel<0, NP, FAST, CAPK>, jv_search_pqp_kernel<2, NP, FAST, CAPK>, jv_search_pqp_kernel<12, NP, FAST, CAPK>, jv_search_pqp_kernel<24, NP, FAST, CAPK> }
#define JV_PQP_CAPS(NP, FAST) { JV_PQP_ROW(NP, FAST, 0), JV_PQP_ROW(NP, FAST, 1), JV_PQP_ROW(NP, FAST, 2), JV_PQP_ROW(NP, FAST, 3) }
// [0 single-pass | 1 multi-pass | 2 single-pass FAST | 3 multi-pass FAST][capacity class][nch slot]
static const pqp_kernel_t g_pqp_kernels[4][4][4] = {JV_PQP_CAPS(1, false), JV_PQP_CAPS(4, false), JV_PQP_CAPS(1, true), JV_PQP_CAPS(4, true)};

// register-LUT variants: PQ-32, FAST, single pass; [capacity class 0..2][nch slot]
#define JV_PQV_ROW(CAPK) \
    { jv_search_pqp_kernel<0, 1, true, CAPK, true>, jv_search_pqp_kernel<2, 1, true, CAPK, true>, jv_search_pqp_kernel<12, 1, true, CAPK, true>, jv_search_pqp_kernel<24, 1, true, CAPK, true> }
static const pqp_kernel_t g_pqv_kernels[3][4] = {JV_PQV_ROW(0), JV_PQV_ROW(1), JV_PQV_ROW(2)};

static int pqp_nch_slot(const JvIndexDev* ix) {
    if (ix->nvq_M > 0) return 0;  // the NVQ decoder lives in the "any d" instances only (score_rows)
    if (ix->stride != ix->nch * 64) return 0;
    return ix->nch == 2 ? 1 : ix->nch == 12 ? 2 : ix->nch == 24 ? 3 : 0;
}
static int pqp_capk(int cap) { return cap <= 512 ? 0 : cap <= 1024 ? 1 : cap <= 2048 ? 2 : cap <= 4096 ? 3 : cap <= 8192 ? 4 : 5; }  // (4, 5: filtered instances only)
// lutr: look-up table in registers (jvk_pqp_lutr_ok shapes only)
extern "C" int jvk_pqp_lutr_ok(const JvIndexDev* ix, int cap) {
    return ix->pq_M == 32 && ix->sim != 2 && ix->R * ix->pq_lanes <= JV_WAVE && cap <= 2048 ? 1 : 0;
}
// (filtered instances: register-table variants for every pool class)
extern "C" int jvk_pqpf_lutr_ok(const JvIndexDev* ix, int cap) {
    return ix->pq_M == 32 && ix->sim != 2 && ix->R * ix->pq_lanes <= JV_WAVE && cap <= 16384 ? 1 : 0;
}
extern "C" const void* jvk_pqpf_kernel(int fast, int capk, int nch_slot, int lutr);  // jv_kernels_pqpf.hip
extern "C" hipError_t jvk_pqpf_set_max_lds(int bytes);
static pqp_kernel_t pqp_pick(const JvIndexDev* ix, int cap, int lutr, int filt = 0) {
    const int multi = ix->R * ix->pq_lanes > JV_WAVE ? 1 : 0;
    const int fast = (ix->pq_M % 16 == 0 && ix->sim != 2) ? 1 : 0;
    if (filt) return (pqp_kernel_t)jvk_pqpf_kernel(fast, pqp_capk(cap), pqp_nch_slot(ix), lutr && jvk_pqpf_lutr_ok(ix, cap));
    if (lutr && jvk_pqp_lutr_ok(ix, cap)) return g_pqv_kernels[pqp_capk(cap)][pqp_nch_slot(ix)];
    return g_pqp_kernels[fast * 2 + multi][pqp_capk(cap)][pqp_nch_slot(ix)];
}

extern "C" hipError_t jvk_pqp_set_max_lds(int bytes) {
    {
        hipError_t e = jvk_pqpf_set_max_lds(bytes);
        if (e != hipSuccess) return e;
    }
    for (int v = 0; v < 4; v++)
        for (int c = 0; c < 4; c++)
            for (int s = 0; s < 4; s++) {
                hipError_t e = hipFuncSetAttribute((const void*)g_pqp_kernels[v][c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e == hipSuccess && v == 0 && c < 3)
                    e = hipFuncSetAttribute((const void*)g_pqv_kernels[c][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

extern "C" int jvk_pqp_max_entries(void) { return 4096; }
extern "C" int jvk_pqp_max_entries_filtered(void) { return 16384; }

// resident workgroups per CU for this index shape, pool capacity and LDS size
extern "C" int jvk_pqp_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes, int lutr, int filt) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pqp_pick(ix, cap, lutr, filt), JV_WAVE, (size_t)lds_bytes) != hipSuccess) return 1;
    return nb < 1 ? 1 : nb;
}

// blocks = resident workgroups (the host sizes the log scratch to it); a->cand_cap = pool entries
extern "C" hipError_t jvk_launch_search_pqp(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, int lutr, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    pqp_pick(ix, a->cand_cap, lutr, a->accept ? 1 : 0)<<<blocks, JV_WAVE, lds_bytes, stream>>>(*ix, *a);  // (a->accept: the instances with a doc filter)
    return hipGetLastError();
}
