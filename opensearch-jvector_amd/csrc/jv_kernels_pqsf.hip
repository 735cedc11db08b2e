// jv_kernels_pqsf.hip — the device-resident query server for one-query calls WITH a doc filter: the persistent pool kernel's
// filtered instances (jv_pqp_body.h, FILT = true; one wave per query) fed from a ring of host-visible slots, as
// jv_kernels_pqs.hip feeds the several-waves kernel with unfiltered queries.  A filtered k-NN query reaches the reader as ONE
// query per call with its acceptDocs (J/JVectorReader.java:129-210, :157-163); served from the ring, a caller waits for ITS
// query only instead of for the combined batch it happened to share.  The slot carries a device pointer to the filter (the
// index's filter cache holds the bits in HBM while the call is in flight).  gfx950 / CDNA4.
#include "jv_pqp_body.h"
#include "jv_serve_claim.h"

template <int NCHT, bool FAST, int CAPK, bool LUTR>
__global__ __launch_bounds__(JV_WAVE, 1) void jv_serve_pqp_kernel(const JvIndexDev ix, const JvSearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* explog = a.pqp_log + (size_t)blockIdx.x * (size_t)a.pqp_log_cap;
    for (;;) {
        int t0 = 0;
        if (threadIdx.x == 0) t0 = jv_serve_claim(a);
        const int ticket = __builtin_amdgcn_readfirstlane(t0);  // (one wave per workgroup: lane 0's value, no LDS word needed)
        if (ticket < 0) break;
        unsigned char* const sp = a.serve_ring + (size_t)(ticket & (a.serve_slots - 1)) * (size_t)a.serve_slot_bytes;
        JvServeSlot* const slot = (JvServeSlot*)sp;
        if (!jv_serve_slot_current(slot, ticket)) {  // an abandoned ticket: nothing to answer
            __syncthreads();
            continue;
        }
        JvSearchArgs aq = a;  // (pool capacity, LDS plan and log length are the server's: every request runs in the largest pool)
        aq.queries = (const float*)(sp + JV_SERVE_QUERY_OFF);
        aq.nq = 1;
        aq.topK = __builtin_amdgcn_readfirstlane(slot->topK);
        aq.rk = __builtin_amdgcn_readfirstlane(slot->rk);
        aq.visit_limit = __builtin_amdgcn_readfirstlane(slot->visit_limit);
        aq.rerank_floor = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(slot->rerank_floor)));
        aq.accept = (const uint64_t*)(uintptr_t)slot->accept;
        aq.accept_docs = slot->accept_docs;
        aq.accept_stride = 0;
        aq.accept_ord = nullptr;
        aq.out_nodes = slot->nodes;
        aq.out_docs = slot->docs;
        aq.out_scores = slot->scores;
        aq.out_count = &slot->count;
        aq.out_stats = slot->stats;
        aq.out_flags = &slot->flags;
        // a filter whose estimated pool does not fit this one: hand the query back at once (retry_only = 1 = "a wider rung
        // follows": the caller takes the launch path and its rungs)
        aq.retry_only = 1;
        search_one_pqp<NCHT, 1, FAST, CAPK, LUTR, true>(ix, aq, 0, smem, explog);
        // completion word: every store of the row precedes it (flagged rows too: the caller redoes those on the launch path)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (threadIdx.x == 0) __hip_atomic_store(&slot->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
    }
    jv_serve_leave(a);
}

typedef void (*pqsf_kernel_t)(const JvIndexDev, const JvSearchArgs);
#define JV_PQSF_ROW(FAST, CAPK, LUTR) \
    { jv_serve_pqp_kernel<0, FAST, CAPK, LUTR>, jv_serve_pqp_kernel<2, FAST, CAPK, LUTR>, jv_serve_pqp_kernel<12, FAST, CAPK, LUTR>, jv_serve_pqp_kernel<24, FAST, CAPK, LUTR> }
// [0 table in LDS, any PQ shape | 1 table in LDS, FAST | 2 table in registers (PQ-32, FAST)][nch slot]; pool class 4 (<= 8 192 entries:
// the host sizes the pool to what keeps two queries resident per CU, ~5 800 entries next to a PQ-32 table)
static const pqsf_kernel_t g_pqsf_kernels[3][4] = {JV_PQSF_ROW(false, 4, false), JV_PQSF_ROW(true, 4, false), JV_PQSF_ROW(true, 4, true)};

static int pqsf_nch_slot(const JvIndexDev* ix) {
    if (ix->nvq_M > 0) return 0;
    if (ix->stride != ix->nch * 64) return 0;
    return ix->nch == 2 ? 1 : ix->nch == 12 ? 2 : ix->nch == 24 ? 3 : 0;
}
static pqsf_kernel_t pqsf_pick(const JvIndexDev* ix, int lutr) {
    const int fast = (ix->pq_M % 16 == 0 && ix->sim != 2) ? 1 : 0;
    return g_pqsf_kernels[lutr && fast ? 2 : fast][pqsf_nch_slot(ix)];
}
extern "C" int jvk_pqsf_max_entries(void) { return 8192; }
extern "C" hipError_t jvk_pqsf_set_max_lds(int bytes) {
    for (int v = 0; v < 3; v++)
        for (int s = 0; s < 4; s++) {
            hipError_t e = hipFuncSetAttribute((const void*)g_pqsf_kernels[v][s], hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}
extern "C" int jvk_pqsf_blocks_per_cu(const JvIndexDev* ix, int lds_bytes, int lutr) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pqsf_pick(ix, lutr), JV_WAVE, (size_t)lds_bytes) != hipSuccess) return 1;
    return nb < 1 ? 1 : nb;
}
// a->cand_cap = the server's pool (class 4: 4 097 .. 8 192 entries); lutr: table in registers (PQ-32, not cosine, single pass)
extern "C" hipError_t jvk_launch_serve_pqpf(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, int lutr, hipStream_t stream) {
    pqsf_pick(ix, lutr)<<<blocks, JV_WAVE, lds_bytes, stream>>>(*ix, *a);
    return hipGetLastError();
}
