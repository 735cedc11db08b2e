// jv_abi.cpp — implementation of the C ABI in include/jvgpu.h on top of the HIP kernels.
//
// jv_index_create copies the flattened FieldEntry (J/JVectorReader.java:284-337) into HBM once;
// jv_search* stage the query (4d bytes) and run the whole GraphSearcher.search
// (J/JVectorReader.java:165-173) on the GPU.  There is NO CPU fallback anywhere in this file: if the
// device or the kernels are unavailable every call fails with JV_EDEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <errno.h>
#include <sched.h>
#include <semaphore.h>
#include <time.h>

#include "../../include/jvgpu.h"
#include "jv_device.h"
#include "jv_xb.h"
#include "jv_serve_host.h"

extern "C" {
// batched exact scorer (jv_kernels_xb.hip)
hipError_t jvk_xb_mirror(const float* src, long long rows, int d, long long src_stride, int kp, uint16_t* dst, float* norm2, int scalar, hipStream_t s);
hipError_t jvk_xb_mark_dead(const int32_t* ord2doc, long long n, float* norm2, hipStream_t s);
hipError_t jvk_xb_build_list(const JvIndexDev* ix, const uint64_t* d_accept, long long accept_docs, int32_t* d_counts, int32_t* d_list, hipStream_t s);
int jvk_xb_list_blocks(int n);
hipError_t jvk_xb_tile(const JvXbTileArgs* a, int mode, hipStream_t s);
int jvk_xb_qs_ok(int kp);
hipError_t jvk_xb_qs(const JvXbTileArgs* a, int mode, int cus, hipStream_t s);
hipError_t jvk_xb_kth(const float* sample, int ld, int S, int k, float* thr, int B, hipStream_t s);
hipError_t jvk_xb_rescore(const JvIndexDev* ix, const JvXbRescoreArgs* a, int nq, hipStream_t s);
hipError_t jvk_set_max_lds(int bytes);
hipError_t jvk_build_fused(const uint8_t* codes, const int32_t* adj, uint8_t* fused, long long n, int R, int cs, hipStream_t s);
hipError_t jvk_build_code_norms(const uint8_t* codes, const float* norm_lut, const int32_t* adj, float* node_norm, float* fused_norm,
                                long long n, int R, int M, int cs, int lpn, hipStream_t s);
hipError_t jvk_launch_search_lds(const JvIndexDev* ix, const JvSearchArgs* a, int pq, int pool, int tag, int lds_bytes, hipStream_t s);
hipError_t jvk_launch_search_pqf(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, hipStream_t s);
hipError_t jvk_launch_search_big(const JvIndexDev* ix, const JvSearchArgs* a, int pq, int blocks, int lds_bytes,
                                 int force_all, int qlds, hipStream_t s);
hipError_t jvk_launch_mark(int32_t* word, int32_t value, hipStream_t stream);
hipError_t jvk_launch_search_big_lutg(const JvIndexDev* ix, const JvSearchArgs* a, int blocks, int lds_bytes, int force_all, int qlds, hipStream_t stream);
hipError_t jvk_launch_score_ordinals(const JvIndexDev* ix, const float* d_query, const int32_t* d_ordinals,
                                     int count, float* d_out, hipStream_t s);
hipError_t jvk_launch_merge_topk(const int32_t* d_docs, const float* d_scores, int nq, int lists, int k,
                                 int32_t* d_out_docs, float* d_out_scores, hipStream_t s);
hipError_t jvk_launch_merge_topk_strided(const int32_t* d_pairs, const float* unused, int nq, int lists, int k,
                                         int32_t* d_out_docs, float* d_out_scores, hipStream_t s);
hipError_t jvk_launch_pack_pairs(const int32_t* d_docs, const float* d_scores, int32_t* d_pairs, long long n, hipStream_t s);
// LDS-pool persistent kernel (jv_kernels_pqp.hip): the headline path
hipError_t jvk_pqp_set_max_lds(int bytes);
int jvk_pqp_max_entries(void);
int jvk_pqp_max_entries_filtered(void);
int jvk_pqp_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes, int lutr, int filt);
int jvk_pqp_lutr_ok(const JvIndexDev* ix, int cap);
int jvk_pqpf_lutr_ok(const JvIndexDev* ix, int cap);
hipError_t jvk_launch_accept_to_ord(const JvIndexDev* ix, const uint64_t* accept, long long accept_docs, uint64_t* out, hipStream_t stream);
hipError_t jvk_launch_search_pqp(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, int lutr, hipStream_t s);
// the same search with pq_M / 16 waves per query, the look-up table in registers split by chunk (jv_kernels_pqw.hip)
hipError_t jvk_pqw_set_max_lds(int bytes);
int jvk_pqw_ok(const JvIndexDev* ix, int cap);
int jvk_pqw_waves(const JvIndexDev* ix);
int jvk_pqw_lds_rows(const JvIndexDev* ix, int variant);
int jvk_pqw_occ5_ok(const JvIndexDev* ix, int cap);
int jvk_pqw_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes, int variant);
hipError_t jvk_launch_search_pqw(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, int variant, hipStream_t s);
int jvk_visited_blocks_per_cu(int slots);
int jvk_visited_lds_bytes(int slots);
hipError_t jvk_launch_visited(const JvVisArgs* a, int blocks, hipStream_t s);
// the same kernel with a doc filter (jv_kernels_pqwf.hip): pools of up to 16 384 entries
hipError_t jvk_pqwf_set_max_lds(int bytes);
int jvk_pqwf_ok(const JvIndexDev* ix, int cap);
int jvk_pqwf_max_entries(const JvIndexDev* ix);
int jvk_pqwf_lds_rows(const JvIndexDev* ix, int cap);
int jvk_pqwf_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes);
hipError_t jvk_launch_search_pqwf(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, hipStream_t s);
// device-resident query server (jv_kernels_pqs.hip)
hipError_t jvk_pqs_set_max_lds(int bytes);
int jvk_pqs_blocks_per_cu(const JvIndexDev* ix, int cap, int lds_bytes);
hipError_t jvk_launch_serve_pqw(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, hipStream_t s);
int jvk_pqsf_max_entries(void);
hipError_t jvk_pqsf_set_max_lds(int bytes);
int jvk_pqsf_blocks_per_cu(const JvIndexDev* ix, int lds_bytes, int lutr);
hipError_t jvk_launch_serve_pqpf(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, int lutr, hipStream_t s);
// the filtered server on the several-waves kernel (jv_kernels_pqs.hip, round 4)
int jvk_pqswf_max_entries(void);
hipError_t jvk_pqswf_set_max_lds(int bytes);
int jvk_pqswf_blocks_per_cu(const JvIndexDev* ix, int lds_bytes);
hipError_t jvk_launch_serve_pqwf(const JvIndexDev* ix, const JvSearchArgs* a, int lds_bytes, int blocks, hipStream_t s);
}

namespace {

constexpr int kMaxLds = 160 * 1024;

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? JV_ENOMEM : JV_EDEVICE, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                   \
    } while (0)

// Tunables.  Every index carries its own set (jv_index_set_option); jv_set_option only changes the DEFAULTS that
// indexes created afterwards start from — nothing process-wide is read at call time.
//   lds_visited_slots / lds_candidates   on-chip scratch geometry of the generic kernels (0 = auto)
//   force_big_path / force_general_path / no_escalation / no_pqf / no_pqp / no_pqw / no_lutr / pqf_only   rung selection (diagnostics)
//   pqw_min_queries                      launches with at least this many queries use the several-waves-per-query kernel
//   pqw_latency_queries                  launches with at most this many queries use its latency variant: whole table in LDS (-1 = 3 per CU)
//   spill_tables x spill_slots           per-context pool of visited-set spill tables (512 x 8192 x 4 B = 16 MB, allocated on first use)
//   big_blocks / big_cand_cap / big_budget_mb   HBM-scratch rung: resident blocks (0 = as many as fit the budget), candidate slots
//   combine / combine_leaders / combine_max_batch   group commit of concurrent jv_search calls
//   max_contexts                         cap of per-index launch contexts (callers beyond it wait)
//   async_contexts                       device-pointer API: contexts (scratch + counters) its calls on different streams may use side by side
//   filter_cache                         device-resident doc-filter bitsets kept per index (0 = off)
//   serve / serve_wgs_per_cu / serve_idle_ms   device-resident query servers for one-query calls — one grid for unfiltered calls, one for calls with a doc filter (resident workgroups per CU, idle time before they leave)
//   lazy_big_rung                        host-pointer calls enqueue the HBM-scratch rung only when a row came back flagged (it serialises batches otherwise)
//   serve_spin_waiters                   one-query calls: up to this many concurrent callers poll (sched_yield) for their completion word behind the first nap; 0 = naps only
//   visited_after                        several-waves batch launches without a visit limit copy their expansion logs to an arena and jv_visited_kernel counts visitedCount for the whole batch afterwards (0 = count inside the search kernel)
//   visited_slots / visited_arena_units  its hash slots per workgroup / arena size in 16-byte units (tests: small values exercise the several-classes path and the in-kernel fall-back)
//   time_search_kernel                   measurement: HIP events around the first search launch of every batch call (counters search_kernel_ns / search_kernel_timed)
//   direct_completion                    combined one-query calls: rows land in pinned memory, every caller is woken by its own query's completion word
//   lutr_min_queries                     launches with more queries keep the PQ look-up table in registers (-1 = 4 per CU)
//   dbg_ptr                              diagnostic build only
enum OptId { OPT_LDS_VISITED_SLOTS, OPT_LDS_CANDIDATES, OPT_FORCE_BIG, OPT_FORCE_GENERAL, OPT_NO_ESCALATION, OPT_DBG_PTR, OPT_NO_PQF, OPT_NO_PQP, OPT_NO_LUTR, OPT_LUTR_MIN_QUERIES, OPT_PQP_BLOCKS_PER_CU, OPT_NO_PQW, OPT_PQW_MIN_QUERIES, OPT_PQW_LAT_QUERIES, OPT_PQF_ONLY, OPT_SPILL_TABLES, OPT_SPILL_SLOTS, OPT_BIG_BLOCKS, OPT_BIG_CAND_CAP, OPT_BIG_BUDGET_MB, OPT_COMBINE, OPT_COMBINE_LEADERS, OPT_COMBINE_MAX_BATCH, OPT_MAX_CONTEXTS, OPT_FILTER_CACHE, OPT_DIRECT_COMPLETION, OPT_LAZY_BIG, OPT_SERVE, OPT_SERVE_WGS_PER_CU, OPT_SERVE_IDLE_MS, OPT_ASYNC_CONTEXTS, OPT_VISITED_AFTER, OPT_VISITED_SLOTS, OPT_VISITED_ARENA_UNITS, OPT_TIME_SEARCH_KERNEL, OPT_SERVE_SPIN_WAITERS, OPT_NO_PRESCORE, OPT_COUNT };
struct OptName { const char* name; int64_t def; };
const OptName kOptNames[OPT_COUNT] = {
    {"lds_visited_slots", 0},
    {"lds_candidates", 0},
    {"force_big_path", 0},
    {"force_general_path", 0},
    {"no_escalation", 0},
    {"dbg_ptr", 0},
    {"no_pqf", 0},
    {"no_pqp", 0},
    {"no_lutr", 0},
    {"lutr_min_queries", -1},
    {"pqp_blocks_per_cu", 0},
    {"no_pqw", 0},
    {"pqw_min_queries", 0},
    {"pqw_latency_queries", -1},
    {"pqf_only", 0},
    {"spill_tables", 512},
    {"spill_slots", 8192},
    {"big_blocks", 0},
    {"big_cand_cap", 65536},
    {"big_budget_mb", 192},
    {"combine", 1},
    {"combine_leaders", 2},
    {"combine_max_batch", 2048},
    {"max_contexts", 8},
    {"filter_cache", 8},
    {"direct_completion", 1},
    {"lazy_big_rung", 1},
    {"serve", 1},
    {"serve_wgs_per_cu", 2},
    {"serve_idle_ms", 100},
    {"async_contexts", 4},
    {"visited_after", 1},
    {"visited_slots", 16384},      // hash slots of jv_visited_kernel's set (tests: a small set sends every log through several classes)
    {"visited_arena_units", 0},    // > 0: the log arena's size in 16-byte units (tests: a small arena sends logs back to the in-kernel count)
    {"time_search_kernel", 0},     // measurement: HIP events around the first (main) search launch of every batch call -> counters search_kernel_ns / search_kernel_timed
    {"serve_spin_waiters", 4},     // one-query calls: up to this many concurrent callers poll for their completion word (sched_yield) instead of napping; 0 = naps only
    {"no_prescore", 0},            // diagnostics (A/B): 1 = the latency variant's helper wave does not pre-score the pair requested ahead
};
struct Opts {
    std::atomic<int64_t> v[OPT_COUNT];
    Opts() {
        for (int i = 0; i < OPT_COUNT; i++) v[i].store(kOptNames[i].def);
    }
    void copy_from(const Opts& o) {
        for (int i = 0; i < OPT_COUNT; i++) v[i].store(o.v[i].load());
    }
    int find(const char* name) const {
        for (int i = 0; i < OPT_COUNT; i++)
            if (strcmp(kOptNames[i].name, name) == 0) return i;
        return -1;
    }
};
Opts g_default_opts;

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct Ctx {
    hipStream_t stream = nullptr;
    hipEvent_t last_use = nullptr;
    hipStream_t last_stream = nullptr;
    uint64_t last_clock = 0;  // (device-pointer API: least-recently-used choice among its contexts)
    // device staging for the host-pointer API
    float* d_queries = nullptr;
    size_t queries_cap = 0;  // floats
    int32_t* d_flags = nullptr;  // flag words for device-pointer calls that pass no out_flags
    size_t nq_cap = 0;
    uint64_t* d_accept = nullptr;
    size_t accept_cap = 0;  // words
    uint64_t* d_accept_ord = nullptr;  // a batch-wide filter in ordinal space (persistent pool kernel's filtered instances)
    size_t accept_ord_cap = 0;         // words
    // host-pointer API: one device arena [nodes|docs|scores|count|stats|flags] + one pinned mirror, so a call
    // is one H2D (query) and ONE D2H instead of six staged pageable copies
    uint8_t* d_arena = nullptr;
    uint8_t* h_arena = nullptr;  // hipHostMalloc
    size_t arena_cap = 0;
    float* h_query = nullptr;    // pinned staging for small query batches
    size_t h_query_cap = 0;
    // combined one-query calls: the kernels write the rows straight into this pinned, device-visible arena
    // [nodes|docs|scores|count|stats|flags|done] and the owner of a query is woken when ITS completion word is set
    uint8_t* h_direct = nullptr;
    size_t direct_cap = 0;
    hipEvent_t ev_direct = nullptr;
    hipEvent_t ev_null = nullptr;  // device-pointer API without a caller stream: what the legacy default stream had in flight at call time
    int32_t* h_mark = nullptr;   // pinned: stream marker written by jv_mark_kernel, polled by the host (see wait_mark)
    int32_t mark_seq = 0;
    int32_t* work_counter = nullptr;  // 24 words: [0] big-path dequeue, [1] spill-table allocator, [2..7] rung counters, [8..15] filtered pool rungs, [16] log-arena cursor
    // pool of visited-set spill tables of the generic kernels (allocated on the first launch that can use it)
    uint32_t* spill = nullptr;
    int spill_tables = 0, spill_slots = 0;
    // register-pool kernel: per-resident-workgroup expansion logs
    int32_t* pqp_log = nullptr;
    size_t pqp_log_ints = 0;
    // visited counts after the launch (jv_kernels_vis.hip): copies of the batch's expansion logs + per-query offset / length
    int32_t* vis_arena = nullptr;
    size_t vis_arena_units = 0;  // 16-byte units
    uint32_t* vis_off = nullptr;
    int32_t* vis_n = nullptr;
    size_t vis_nq_cap = 0;
    // measurement (option time_search_kernel): event pairs around the main search launch, read back by jv_index_get_counter
    hipEvent_t kt[16][2] = {};
    bool kt_pending[16] = {};
    int kt_head = 0;
};

// One caller's jv_search waiting to be served.  Lives on the caller's stack.
struct PendingSearch {
    const float* query;
    int32_t topK, rerankK;
    float threshold, rerankFloor;
    int32_t* out_nodes;
    int32_t* out_docs;
    float* out_scores;
    int32_t* out_count;
    int32_t* out_stats;
    const uint64_t* accept;  // this query's doc filter (host words) or nullptr
    int64_t accept_docs;
    int64_t visit_limit;
    int32_t* out_flags;      // optional JV_QFLAG_* word
    int rc = 0;
    char err[256];
    enum { QUEUED, TAKEN } state = QUEUED;
    bool promoted = false;
    sem_t sem;  // posted once per promotion and once when another thread has served the request
};

// Group commit for the one-query-per-call API: a caller that finds a free leader slot takes every queued
// request with the same parameters (its own included), runs them as ONE batch launch and hands the answers
// back; callers arriving meanwhile queue up and form the next batch.
struct Combiner {
    std::mutex mu;
    std::deque<PendingSearch*> queue;
    int active = 0;  // leaders running or promoted
};

}  // namespace

namespace {
// The HBM-scratch rung's buffers (visited bitsets + candidate queues of its resident blocks): ONE set per device, shared
// by every index and context on it, allocated on the first search and grown on demand.  Launches that use it are
// ordered through `last_use` (they are the rare tail of a batch; the headline kernels never touch it).
struct DeviceScratch {
    std::mutex mu;
    hipEvent_t last_use = nullptr;
    uint32_t* big_visited = nullptr;  // every index carves [its blocks][its ceil(n / 32) words] out of this arena
    int64_t* big_cand = nullptr;      // ... and [its blocks][its candidate slots] out of this one
    size_t vis_bytes = 0, cand_bytes = 0;
    float* lut = nullptr;             // PQ tables too large for LDS: [blocks][pq_M][256] of the launch in flight
    size_t lut_bytes = 0;
    std::atomic<int64_t> bytes{0};
};
DeviceScratch g_scratch[64];
}  // namespace

namespace {
// device-resident copies of recently used doc-filter bitsets (per index), keyed by content hash + length
struct FilterEntry {
    uint64_t key = 0;
    size_t words = 0;
    uint64_t* d_words = nullptr;
    size_t cap_words = 0;
    uint64_t stamp = 0;
    int users = 0;  // launches in flight that read it
    int no_serve_rk = 0;  // > 0: the filtered query server's pool cannot hold this filter at rerankK >= this (its kernel said so): such calls skip the ring
    hipEvent_t ready = nullptr;  // recorded after the upload: consumers on other streams wait for it
    std::vector<uint64_t> host;  // the cached bits: a hit is served only after a memcmp against the caller's bitset (the key —
                                 // a 64-bit content hash or a caller-supplied number — only finds the candidate entry)
};

}  // namespace

// Batched exact scorer (jv_score_ordinals_batch): the bf16 mirror of the vectors (built by the first call that can use it)
// and the call's scratch.  One call at a time per index (`mu`); the kernels of a call fill the GPU by themselves.
struct XbState {
    std::mutex mu;
    hipStream_t stream = nullptr;
    int mirror_state = 0;           // 0 = not tried, 1 = ready, -1 = unavailable (NVQ-only field, or no HBM left for it)
    uint16_t* vb = nullptr;         // [n][kp] bf16
    float* vnorm2 = nullptr;        // [n]
    int kp = 0;
    int32_t* d_list = nullptr; size_t list_cap = 0;       // candidate ordinals
    int32_t* d_counts = nullptr; size_t counts_cap = 0;   // list construction: per-block counts / offsets
    float* d_queries = nullptr; size_t queries_cap = 0;   // fp32 queries (host-pointer API)
    uint16_t* d_qb = nullptr; size_t qb_cap = 0;          // bf16 queries of one round
    float* d_qn2 = nullptr; float* d_thr = nullptr; int32_t* d_surv_cnt = nullptr; size_t round_cap = 0;
    float* d_sample = nullptr; size_t sample_cap = 0;
    int32_t* d_surv = nullptr; size_t surv_cap = 0;
    uint8_t* d_out = nullptr; uint8_t* h_out = nullptr; size_t out_cap = 0;
    int64_t* d_info = nullptr;
    int64_t* h_info = nullptr;      // pinned: [0..1] kernel counters, [2] list length
    uint64_t* d_accept = nullptr; size_t accept_cap = 0;  // filter words when the cache has no slot
    int64_t bytes = 0;
    // one-query calls (jv_exact_search): group commit — while one batch runs, later arrivals queue; the next leader takes every
    // queued call with ITS filter and topK as one batch
    std::mutex cmu;
    std::deque<struct XbPending*> cqueue;
    bool cleader = false;
    std::atomic<int64_t> exact_calls{0}, exact_batches{0};
};

struct JvQueryServer;
struct jv_index {
    XbState xb;
    Opts opts;
    Combiner combiner;
    int device = 0;
    bool build_client = false;  // JV_DESC_BUILD_CLIENT: searches are launched under the builder's kernel name
    JvIndexDev dev{};
    int cu_count = 256;
    std::vector<int32_t> pq_sub_off;  // host copy of dev.pq_sub_off
    jv_index_info info{};
    std::vector<void*> owned;  // device allocations to free
    std::mutex mu;
    std::condition_variable ctx_cv;
    std::vector<Ctx*> free_ctx;
    std::vector<Ctx*> all_ctx;
    std::mutex filter_mu;
    std::vector<FilterEntry> filters;  // device-side doc-filter cache
    uint64_t filter_clock = 0;
    int64_t filter_hits = 0, filter_misses = 0;
    Ctx* async_ctx = nullptr;           // the device-pointer API's first context ...
    std::vector<Ctx*> async_ctxs;       // ... and the others: calls on DIFFERENT streams run side by side, each in a context of its own
    uint64_t async_clock = 0;
    std::mutex async_mu;
    // launches per kernel family since creation (jv_index_get_counter): which rung served a call is observable
    std::atomic<int64_t> launches[8] = {};
    std::atomic<int64_t> search_kernel_ns{0}, search_kernel_timed{0};  // (option time_search_kernel)
    std::atomic<int64_t> retry_rungs_skipped{0};  // redo launches a host-pointer batch call left out beside a live query-server grid (its flagged rows take the HBM-scratch rung)
    struct JvQueryServer* server = nullptr;  // device-resident query server (created by the first eligible one-query call)
    struct JvQueryServer* server_f = nullptr;  // the same for one-query calls WITH a doc filter (one-wave filtered pool kernel)
    std::mutex server_mu;
    std::condition_variable server_cv;  // a server's last caller has left (server_get waits for it before it rebuilds a ring)
};
enum { LAUNCH_PQW = 0, LAUNCH_PQP, LAUNCH_PQF, LAUNCH_LDS, LAUNCH_BIG, LAUNCH_SERVE, SERVED_QUERIES };

#define OPT(ixp, id) ((ixp)->opts.v[id].load(std::memory_order_relaxed))

namespace {

template <typename T>
int dev_alloc(jv_index* ix, T** out, size_t count) {
    void* p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIPCHK(hipMalloc(&p, bytes));
    ix->owned.push_back(p);
    ix->info.hbm_bytes += (int64_t)bytes;
    *out = (T*)p;
    return JV_OK;
}

// hipFree / hipHostFree synchronise the device (they drain every stream): resident query servers of that device must leave
// first — next to another index's grid under steady one-query load such a call would wait for ever.  The device is the
// pointer's OWNER (hipPointerGetAttributes), not whatever device the calling thread happens to have current.
void free_with_servers_paused(void* p, bool host);
static inline hipError_t jv_free(void* p) {
    free_with_servers_paused(p, false);
    return hipSuccess;
}
static inline void jv_host_free(void* p) { free_with_servers_paused(p, true); }

int grow(void** p, size_t* cap, size_t need, size_t elem) {
    if (need <= *cap && *p) return JV_OK;
    if (*p) jv_free(*p);
    *p = nullptr;
    size_t ncap = need < 16 ? 16 : need + need / 2;
    HIPCHK(hipMalloc(p, ncap * elem));
    *cap = ncap;
    return JV_OK;
}

int ctx_create(jv_index* ix, Ctx** out) {
    Ctx* c = new Ctx();
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->last_use, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&c->work_counter, 24 * sizeof(int32_t));
    if (e != hipSuccess) {
        delete c;
        return fail(JV_EDEVICE, "context creation failed: %s", hipGetErrorString(e));
    }
    ix->all_ctx.push_back(c);
    *out = c;
    return JV_OK;
}

void ctx_destroy(Ctx* c) {
    if (!c) return;
    jv_free(c->d_queries);
    jv_free(c->d_flags);
    jv_free(c->d_accept);
    jv_free(c->d_accept_ord);
    jv_host_free(c->h_mark);
    jv_free(c->d_arena);
    jv_host_free(c->h_arena);
    jv_host_free(c->h_query);
    jv_host_free(c->h_direct);
    if (c->ev_direct) hipEventDestroy(c->ev_direct);
    if (c->ev_null) hipEventDestroy(c->ev_null);
    jv_free(c->work_counter);
    jv_free(c->spill);
    jv_free(c->pqp_log);
    jv_free(c->vis_arena);
    jv_free(c->vis_off);
    jv_free(c->vis_n);
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 2; j++)
            if (c->kt[i][j]) hipEventDestroy(c->kt[i][j]);
    if (c->last_use) hipEventDestroy(c->last_use);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// a launch context: from the free list, a new one while the index is below its cap, else wait for a release
int ctx_acquire(jv_index* ix, Ctx** out) {
    std::unique_lock<std::mutex> lk(ix->mu);
    for (;;) {
        if (!ix->free_ctx.empty()) {
            *out = ix->free_ctx.back();
            ix->free_ctx.pop_back();
            return JV_OK;
        }
        const size_t cap = (size_t)std::max<int64_t>(1, OPT(ix, OPT_MAX_CONTEXTS));
        if (ix->all_ctx.size() < cap + std::max<size_t>(1, ix->async_ctxs.size())) return ctx_create(ix, out);  // (+ the device-pointer API's own contexts)
        ix->ctx_cv.wait(lk);
    }
}
void ctx_release(jv_index* ix, Ctx* c) {
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        ix->free_ctx.push_back(c);
    }
    ix->ctx_cv.notify_one();
}

uint64_t hash_words(const uint64_t* w, size_t n) {
    // 4 independent multiply-xorshift lanes over 64-bit words (~10 GB/s on one core), folded at the end
    uint64_t h0 = 0x9E3779B97F4A7C15ull ^ n, h1 = 0xC2B2AE3D27D4EB4Full, h2 = 0x165667B19E3779F9ull, h3 = 0x27D4EB2F165667C5ull;
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        h0 = (h0 ^ w[i]) * 0xFF51AFD7ED558CCDull; h0 ^= h0 >> 29;
        h1 = (h1 ^ w[i + 1]) * 0xC4CEB9FE1A85EC53ull; h1 ^= h1 >> 31;
        h2 = (h2 ^ w[i + 2]) * 0x9FB21C651E98DF25ull; h2 ^= h2 >> 30;
        h3 = (h3 ^ w[i + 3]) * 0xD6E8FEB86659FD93ull; h3 ^= h3 >> 28;
    }
    for (; i < n; i++) {
        h0 = (h0 ^ w[i]) * 0xFF51AFD7ED558CCDull; h0 ^= h0 >> 29;
    }
    uint64_t h = h0 ^ (h1 * 3) ^ (h2 * 5) ^ (h3 * 7);
    h ^= h >> 32;
    h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 29;
    return h ? h : 1;
}

struct Geometry {
    int hash_slots, cand_cap, res_cap;
    int lds_fast, lds_big;
    bool fast_ok;
    bool pool;
    bool lutg;  // the PQ look-up table does not fit LDS: HBM-scratch rung only, table in HBM scratch too
};

// LDS carve of the fast path (must mirror search_one in jv_kernels.hip)
Geometry plan_geometry(const jv_index* ix, int rk, bool pool_ok, int force_slots = 0, bool tracker = false) {
    Geometry g{};
    const JvIndexDev& d = ix->dev;
    const bool pq = d.pq_M > 0;
    int fixed = d.nch * 64 * 4 + JV_TODO * 16 + (pq ? d.pq_M * 256 * 4 + d.nch * 64 * 4 : 0) + (tracker ? JV_TRACKER_LDS : 0);
    // pq_M KB of table beyond the LDS (the reference's default for d >= 768 is 192 subspaces, J/JVectorIndexQuantization.java:428-446):
    // the HBM-scratch rung keeps the table in HBM scratch as well
    g.lutg = pq && fixed > kMaxLds;
    if (g.lutg) fixed -= d.pq_M * 256 * 4;
    g.lds_big = fixed;
    int64_t hs = OPT(ix, OPT_LDS_VISITED_SLOTS);
    int64_t cc = OPT(ix, OPT_LDS_CANDIDATES);
    g.pool = pool_ok && OPT(ix, OPT_FORCE_GENERAL) == 0;
    // visited set: ~10-25 x rerankK nodes at R=32; the table may fill to 75 %.  Overflow is not an error:
    // the query is re-run exactly on the HBM-scratch path.
    g.hash_slots = hs > 0 ? next_pow2((int)hs) : next_pow2(rk * 24 < 1024 ? 1024 : rk * 24);
    if (force_slots > 0) g.hash_slots = force_slots;
    if (g.hash_slots > 32768) g.hash_slots = 32768;
    if (g.pool) {
        // two pool buffers: rk entries + 64 boundary ties + one chunk of 64 new keys
        g.res_cap = (rk + 128 + 1) & ~1;
        g.cand_cap = g.res_cap;
        if (cc > 0) g.cand_cap = g.res_cap = ((int)cc + 1) & ~1;
    } else {
        g.res_cap = (rk + 1) & ~1;
        if (g.res_cap < 2) g.res_cap = 2;
        // two-queue form: the candidate queue can hold every visited node until the result queue fills
        g.cand_cap = cc > 0 ? (int)cc : (g.hash_slots / 4) * 3;
        if (g.cand_cap < rk) g.cand_cap = rk;
        g.cand_cap = (g.cand_cap + 1) & ~1;
    }
    auto total = [&]() { return (int64_t)fixed + (int64_t)g.res_cap * 8 + (int64_t)g.cand_cap * 8 + (int64_t)g.hash_slots * 4; };
    while (total() > kMaxLds && g.hash_slots > 256 && hs <= 0) {
        g.hash_slots >>= 1;
        if (!g.pool && cc <= 0) g.cand_cap = std::max(rk, (g.hash_slots / 4) * 3) & ~1;
    }
    g.fast_ok = !g.lutg && total() <= kMaxLds && fixed <= kMaxLds;
    g.lds_fast = (int)total();
    return g;
}

// size / grow the device's shared HBM-scratch (caller holds sc.mu)
int ensure_big(jv_index* ix, DeviceScratch& sc, int rk, int* cap_out, int* blocks_out) {
    int cap = (int)OPT(ix, OPT_BIG_CAND_CAP);
    int res_cap = ((rk + 1) & ~1);
    if (cap < 4 * rk) cap = 4 * rk;
    // the candidate queue never holds more than every node once; the addTopCandidate log of the PQ rungs (one entry per
    // expanded node at most) shares the area
    if (cap > 2 * ix->dev.n + 64) cap = 2 * ix->dev.n + 64;
    cap += res_cap;
    size_t words = ((size_t)ix->dev.n + 31) / 32;
    if (words == 0) words = 1;
    if (!sc.last_use) HIPCHK(hipEventCreateWithFlags(&sc.last_use, hipEventDisableTiming));
    int blocks = (int)OPT(ix, OPT_BIG_BLOCKS);
    if (blocks < 1) {
        // queries that outgrow the on-chip scratch (selective filters, huge rerankK) run here: the more resident
        // blocks, the more of them run side by side; sized to the budget
        const size_t per = words * sizeof(uint32_t) + (size_t)cap * sizeof(int64_t);
        const size_t budget = (size_t)std::max<int64_t>(16, OPT(ix, OPT_BIG_BUDGET_MB)) << 20;
        blocks = (int)std::min<size_t>(1024, std::max<size_t>(16, budget / per));
    }
    *cap_out = cap;        // this index's queue capacity and resident blocks (the shared arenas may be larger)
    *blocks_out = blocks;
    const size_t need_vis = (size_t)blocks * words * sizeof(uint32_t), need_cand = (size_t)blocks * (size_t)cap * sizeof(int64_t);
    if (sc.big_visited && sc.big_cand && sc.vis_bytes >= need_vis && sc.cand_bytes >= need_cand) return JV_OK;
    // grow: wait for the launches still using the old buffers
    HIPCHK(hipEventSynchronize(sc.last_use));
    if (sc.vis_bytes < need_vis || !sc.big_visited) {
        if (sc.big_visited) HIPCHK(jv_free(sc.big_visited));
        sc.big_visited = nullptr;
        sc.vis_bytes = 0;
        HIPCHK(hipMalloc((void**)&sc.big_visited, need_vis));
        sc.vis_bytes = need_vis;
    }
    if (sc.cand_bytes < need_cand || !sc.big_cand) {
        if (sc.big_cand) HIPCHK(jv_free(sc.big_cand));
        sc.big_cand = nullptr;
        sc.cand_bytes = 0;
        HIPCHK(hipMalloc((void**)&sc.big_cand, need_cand));
        sc.cand_bytes = need_cand;
    }
    sc.bytes = (int64_t)(sc.vis_bytes + sc.cand_bytes);
    return JV_OK;
}

int ensure_spill(jv_index* ix, Ctx* c) {
    const int st = (int)OPT(ix, OPT_SPILL_TABLES), ss = next_pow2((int)OPT(ix, OPT_SPILL_SLOTS));
    if (c->spill_tables != st || c->spill_slots != ss) {
        if (c->spill) HIPCHK(jv_free(c->spill));
        c->spill = nullptr;
        c->spill_tables = c->spill_slots = 0;
        if (st > 0 && ss > 0) {
            HIPCHK(hipMalloc((void**)&c->spill, (size_t)st * ss * sizeof(uint32_t)));
            c->spill_tables = st;
            c->spill_slots = ss;
        }
    }
    return JV_OK;
}

// LDS plan of the several-waves pool kernel for a pool of x.cand_cap entries:
// [pool | centred query | table rows | hash set | rerank scratch] + the waves' exchange rows + ctrl words
int plan_pqw_lds(const jv_index* ix, JvSearchArgs& x, int variant, int lds_rows = -1) {
    const int Wn = jvk_pqw_waves(&ix->dev);
    const int qc_b = ix->dev.nch * 64 * 4;
    const int pool_b = (x.cand_cap + 1) * 8;
    const int rr_b = qc_b + Wn * JV_TODO * 8 + (std::min(x.cand_cap, 2048) + 64) * 8;  // (what is reranked fits the waves' registers: <= 2 048 entries)
    x.pqw_lut_off = (std::max(pool_b, qc_b) + 15) & ~15;  // table rows kept in LDS, behind the pool / the centred query
    const int lut_end = x.pqw_lut_off + Wn * (lds_rows >= 0 ? lds_rows : jvk_pqw_lds_rows(&ix->dev, variant)) * 1024;
    const int front = (std::max(std::max(16384, lut_end), std::max(qc_b, rr_b)) + 15) & ~15;
    x.pqp_pool_off = 0;
    x.pqp_qc_off = 0;
    x.pqp_scratch_off = front;
    x.pqp_lds_bytes = front + Wn * 256 + 64 + 128 + 64;  // (ctrl words, the diagnostic build's phase accumulators, the waves' visited counts)
    // (round 6, the latency variant of two-wave queries: the helper wave pre-scores the pair requested ahead into two 64-float
    //  buffers behind the words above — jv_pqw_body.h PRE; four-wave queries — PQ-64 — the same)
    if ((Wn == 2 || Wn == 4) && variant == 1 && lds_rows < 0) x.pqp_lds_bytes += 512;
    return x.pqp_lds_bytes;
}

// Resident query-server grids hold their LDS for as long as they live; a launch whose workgroups need more than what they
// leave free on a CU would not start before the grids idle out (serve_idle_ms) — or never, under steady one-query traffic.
// Such a launch asks the grids to leave first (they come back with the next one-query call).  Defined with the servers below.
void servers_yield_lds(int device, int lds_needed);
bool servers_leave_room(int device, int lds_needed);  // true when a launch of that LDS size fits beside the live grids
int servers_free_lds(int device);  // LDS bytes per CU the live grids leave to other workgroups (the whole budget when none is alive)

// LDS plan of one launch of the one-wave pool kernel (jv_kernels_pqp.hip / jv_kernels_pqpf.hip / jv_kernels_pqsf.hip): offsets
// into the workgroup's LDS for a pool of x.cand_cap entries and beams of up to `rk`; regs = look-up table in registers
int plan_pqp_lds(const jv_index* ix, JvSearchArgs& x, bool regs, int rk) {
    const int lut_b = ix->dev.pq_M * 256 * 4;
    const int qc_b = ix->dev.nch * 64 * 4;
    bool alias = qc_b <= lut_b;
    const int off_f = ix->dev.pq_M * 256 - ix->dev.nch * 64;
    for (int m = 0; alias && m + 1 < ix->dev.pq_M; m++)
        if ((m + 1) * 256 - off_f > ix->pq_sub_off[(size_t)m + 1]) alias = false;
    const int rerank_b = qc_b + JV_TODO * 8 + ((rk + 1) & ~1) * 8;   // after the search, in front of the pool
    int lds;
    const int pool_b = (x.cand_cap + 1) * 8;
    if (regs) {
        // table in registers: LDS = the pool while searching, one hash set afterwards (>= 4 096 slots), then the
        // rerank scratch with the pool written back behind the query and todo lists
        x.pqp_pool_off = 0;
        x.pqp_qc_off = 0;
        x.pqp_scratch_off = (pool_b + 15) & ~15;
        lds = std::max(std::max(x.pqp_scratch_off + 768, qc_b), std::max(16384, qc_b + JV_TODO * 8 + pool_b));
        // (filtered classes 3, 4: the pool stays in LDS and everything after the search lives behind what is left of it)
        if (x.cand_cap > 2048) lds = x.pqp_scratch_off + 768;
    } else {
        x.pqp_pool_off = (std::max(lut_b, rerank_b) + 15) & ~15;
        x.pqp_scratch_off = (x.pqp_pool_off + pool_b + 15) & ~15;
        lds = x.pqp_scratch_off + 768;
        x.pqp_qc_off = alias ? lut_b - qc_b : ((lds + 15) & ~15);
        if (!alias) lds = x.pqp_qc_off + qc_b;
    }
    lds = (lds + 15) & ~15;
    x.pqp_lds_bytes = lds;
    return lds;
}

// Measurement (option time_search_kernel; bench.py's roofline): the duration of the batch call's FIRST search launch — the kernel
// that moves the bytes — from HIP events recorded around it on the stream it is launched on.  The later launches of a call (the
// visited-count kernels, the launch that redoes flagged rows) are not inside the pair.
void kt_harvest(jv_index* ix, Ctx* c) {
    for (int i = 0; i < 16; i++)
        if (c->kt_pending[i]) {
            float ms = 0.0f;
            if (hipEventSynchronize(c->kt[i][1]) == hipSuccess && hipEventElapsedTime(&ms, c->kt[i][0], c->kt[i][1]) == hipSuccess) {
                ix->search_kernel_ns += (int64_t)((double)ms * 1e6);
                ix->search_kernel_timed++;
            }
            c->kt_pending[i] = false;
        }
}
int kt_begin(jv_index* ix, Ctx* c, hipStream_t stream) {
    if (OPT(ix, OPT_TIME_SEARCH_KERNEL) == 0) return -1;
    const int slot = c->kt_head & 15;
    if (c->kt_pending[slot]) kt_harvest(ix, c);
    for (int j = 0; j < 2; j++)
        if (!c->kt[slot][j] && hipEventCreate(&c->kt[slot][j]) != hipSuccess) return -1;
    if (hipEventRecord(c->kt[slot][0], stream) != hipSuccess) return -1;
    return slot;
}
void kt_end(Ctx* c, int slot, hipStream_t stream) {
    if (slot < 0) return;
    if (hipEventRecord(c->kt[slot][1], stream) == hipSuccess) {
        c->kt_pending[slot] = true;
        c->kt_head++;
    }
}

// enqueue one batch on `stream`; all pointers are device pointers
void trace_point(Ctx* c, int i, hipStream_t stream);  // (JV_BATCH_TRACE diagnostics)
int enqueue_batch(jv_index* ix, Ctx* c, hipStream_t stream, const float* d_queries, int nq, int topK, int rk,
                  float thr, float floor_, const uint64_t* d_accept, int64_t accept_docs, int32_t* d_nodes,
                  int32_t* d_docs, float* d_scores, int32_t* d_count, int32_t* d_stats, int32_t* d_flags,
                  int64_t accept_stride = 0, int64_t visit_limit = 0, int32_t* done = nullptr, int phase = 0,
                  bool* big_deferred = nullptr) {
    // phase 0: the whole ladder (device-pointer API: nothing can be decided on the host between the rungs);
    // phase 1: the on-chip rungs only — the HBM-scratch rung lives in ONE arena per device and chains every batch behind the
    //          previous one through its event, so host-pointer callers enqueue it only when a row actually came back flagged
    //          (*big_deferred says whether that is still owed);  phase 2: the HBM-scratch rung alone, for the flagged rows.
    if (big_deferred) *big_deferred = false;
    const bool pq = ix->dev.pq_M > 0;
    // the single-pool form is exact only without a filter and with threshold <= 0 (kernel re-checks scores)
    Geometry g = plan_geometry(ix, rk, d_accept == nullptr && thr <= 0.0f, 0, thr > 0.0f);
    if (g.lds_big > kMaxLds)
        return fail(JV_EUNSUPPORTED, "query + PQ look-up table need %d B of LDS (> %d): pq_M=%d too large", g.lds_big,
                    kMaxLds, ix->dev.pq_M);
    int rc = ensure_spill(ix, c);
    if (rc != JV_OK) return rc;
    JvSearchArgs a{};
    a.queries = d_queries;
    a.qlist = nullptr;
    a.nq = nq;
    a.topK = topK;
    a.rk = rk;
    a.threshold = thr;
    a.rerank_floor = floor_;
    a.accept = d_accept;
    a.accept_docs = accept_docs;
    a.accept_stride = d_accept ? accept_stride : 0;
    a.out_nodes = d_nodes;
    a.out_docs = d_docs;
    a.out_scores = d_scores;
    a.out_count = d_count;
    a.out_stats = d_stats;
    a.out_flags = d_flags;
    a.hash_slots = g.hash_slots;
    a.cand_cap = g.cand_cap;
    a.res_cap = g.res_cap;
    a.visit_limit = visit_limit > 0 ? (int32_t)std::min<int64_t>(visit_limit, INT32_MAX) : 0;
    // (visited <= n distinct nodes and expanded <= n: a limit beyond 2 n can never be reached — such a call is a call without one,
    //  and takes the launches without the in-kernel count; Lucene's unfiltered searches carry Integer.MAX_VALUE)
    if (visit_limit > 2 * (int64_t)ix->dev.n) a.visit_limit = 0;
    a.done = done;  // (completion words: honoured by the several-waves pool kernel; rows of other kernels are final at stream end)
    a.work_counter = c->work_counter;
    a.retry_only = 0;
    a.no_prescore = OPT(ix, OPT_NO_PRESCORE) != 0 ? 1 : 0;
    a.spill = c->spill;
    a.spill_slots = c->spill_slots;
    a.spill_tables = c->spill_tables;
    a.spill_counter = c->work_counter + 1;
    a.retry_counter = c->work_counter + 2;
    a.dbg = (int64_t*)(uintptr_t)OPT(ix, OPT_DBG_PTR);  // always 0 unless a diagnostic run set it
    const bool force_big = OPT(ix, OPT_FORCE_BIG) != 0 || !g.fast_ok;
    trace_point(c, 3, stream);
    HIPCHK(hipMemsetAsync(c->work_counter, 0, 24 * sizeof(int32_t), stream));
    trace_point(c, 4, stream);
    // Launches that only REDO flagged rows (second pool launch, 4x-hash escalation) need large workgroups; beside a live query-
    // server grid they used to ask the grid to leave (servers_yield_lds) on EVERY batch call, flagged rows or not — a stop the
    // caller waits for, and a restart for the next one-query call (tools/grid_start_probe.py: 600 grid starts in 6 s, p99 of the
    // batch calls 17 ms, stalls of 0.6 s).  A host-pointer call (phase 1) now skips such a launch when it does not fit beside
    // the live grids: its rows stay flagged and take the HBM-scratch rung in phase 2, which the caller only enqueues when a row
    // actually came back flagged.  Unfiltered calls only: a selective filter's wide-pool rungs are its main path.
    // "Fits" counts every EXISTING server grid, running or not (servers_free_lds): a grid that starts between this check and the
    // kernel's dispatch takes its LDS on every CU first, and a workgroup that needs more than the rest is not placed until the grid
    // leaves — idle for serve_idle_ms, or never under steady traffic.  That race was the "call that overlaps a grid's START waits
    // until the grid idles out" of round 3 (JV_BATCH_TRACE=1: the stream stopped in front of the escalation rung's launch, whose
    // workgroups had no row to redo but each needed ~100 KB of LDS).
    auto retry_rung_ok = [&](int lds_bytes) -> bool {
        if (phase == 1 && d_accept == nullptr && servers_free_lds(ix->device) < lds_bytes) {
            ix->retry_rungs_skipped++;  // (observable: jv_index_get_counter "retry_rungs_skipped")
            return false;
        }
        servers_yield_lds(ix->device, lds_bytes);
        return true;
    };
    // Visited counts after the launch (jv_kernels_vis.hip, round 5): a several-waves launch without a visit limit and without
    // completion words copies every query's expansion log into an arena (2 x rerankK + 256 entries per query on average, at most
    // 8 GB; a log that finds no room is counted inside the search kernel as before) and jv_visited_kernel counts the batch.
    auto vis_attach = [&](JvSearchArgs& x) -> int {
        if (OPT(ix, OPT_VISITED_AFTER) == 0 || done != nullptr) return JV_OK;
        // (its workgroups need 78 KB of LDS: beside a query-server grid — existing, running or not, see retry_rung_ok above — they
        //  would wait for the grid to idle out; such a batch counts inside the search kernel
        //  — unless it is a throughput batch, which asks the grids to leave as the search launch itself does)
        if (nq >= 4 * ix->cu_count) servers_yield_lds(ix->device, jvk_visited_lds_bytes(16384));
        else if (servers_free_lds(ix->device) < jvk_visited_lds_bytes(16384)) return JV_OK;
        const size_t per_q = ((size_t)x.pqp_log_cap + 3) / 4;  // 16-byte units of the longest log
        if ((size_t)nq * per_q >= ((size_t)1 << 32)) return JV_OK;  // (the cursor is 32 bits wide)
        // (with a doc filter a search expands ~ rerankK / selectivity entries: room for a pool's worth of entries per query)
        const size_t typical = d_accept ? (size_t)x.cand_cap + 256 : (size_t)2 * rk + 256;
        size_t units = std::min<size_t>((size_t)nq * std::min<size_t>(per_q, (typical + 3) / 4), (size_t)1 << 29);
        if (OPT(ix, OPT_VISITED_ARENA_UNITS) > 0) units = (size_t)OPT(ix, OPT_VISITED_ARENA_UNITS);
        if (units > c->vis_arena_units) {
            if (c->vis_arena) HIPCHK(jv_free(c->vis_arena));
            c->vis_arena = nullptr;
            c->vis_arena_units = 0;
            if (hipMalloc((void**)&c->vis_arena, units * 16) != hipSuccess) {
                (void)hipGetLastError();  // no room for the arena: the search kernel counts
                c->vis_arena = nullptr;
                return JV_OK;
            }
            c->vis_arena_units = units;
        }
        if ((size_t)nq > c->vis_nq_cap) {
            if (c->vis_off) HIPCHK(jv_free(c->vis_off));
            if (c->vis_n) HIPCHK(jv_free(c->vis_n));
            c->vis_off = nullptr;
            c->vis_n = nullptr;
            c->vis_nq_cap = 0;
            HIPCHK(hipMalloc((void**)&c->vis_off, (size_t)nq * sizeof(uint32_t)));
            HIPCHK(hipMalloc((void**)&c->vis_n, (size_t)nq * sizeof(int32_t)));
            c->vis_nq_cap = (size_t)nq;
        }
        HIPCHK(hipMemsetAsync(c->vis_n, 0, (size_t)nq * sizeof(int32_t), stream));
        x.vis_arena = c->vis_arena;
        x.vis_cap_units = (uint32_t)std::min<size_t>(OPT(ix, OPT_VISITED_ARENA_UNITS) > 0 ? units : c->vis_arena_units, 0x7FFFFFFFull);
        x.vis_cursor = (uint32_t*)(c->work_counter + 16);
        x.vis_off = c->vis_off;
        x.vis_n = c->vis_n;
        return JV_OK;
    };
    auto vis_count = [&](const JvSearchArgs& x) -> int {
        if (!x.vis_arena) return JV_OK;
        JvVisArgs v{};
        v.adj = ix->dev.adj;
        v.R = ix->dev.R;
        v.entry = ix->dev.entry;
        v.arena = x.vis_arena;
        v.vis_off = x.vis_off;
        v.vis_n = x.vis_n;
        v.nq = nq;
        int slots = 16384;
        if (OPT(ix, OPT_VISITED_SLOTS) >= 256 && OPT(ix, OPT_VISITED_SLOTS) < 16384) slots = next_pow2((int)OPT(ix, OPT_VISITED_SLOTS));
        v.slots = slots;
        v.out_stats = x.out_stats;
        v.visit_limit = x.visit_limit;
        v.topK = x.topK;
        v.out_nodes = x.out_nodes;
        v.out_docs = x.out_docs;
        v.out_scores = x.out_scores;
        v.out_count = x.out_count;
        v.out_flags = x.out_flags;
        v.dbg = x.dbg ? (unsigned long long*)x.dbg + 16 : nullptr;  // (diagnostic runs: the words behind the search kernel's sixteen)
        static const int per_cu = jvk_visited_blocks_per_cu(16384);
        HIPCHK(jvk_launch_visited(&v, std::min(nq, ix->cu_count * per_cu), stream));
        return JV_OK;
    };
    // PQ tables beyond the LDS (g.lutg: the reference's default 192 subspaces): on the fused layout the several-waves kernel
    // takes them first — twelve waves hold the table in registers + LDS rows — and the HBM-scratch rung (table in HBM) only
    // redoes what comes back flagged; everything else about such shapes (filters, thresholds, cosine) is that rung's alone
    bool wide_first = false;
    if (phase != 2 && g.lutg && OPT(ix, OPT_FORCE_BIG) == 0 && OPT(ix, OPT_NO_PQW) == 0 && OPT(ix, OPT_NO_PQF) == 0 && OPT(ix, OPT_NO_PQP) == 0 &&
        d_accept == nullptr && thr <= 0.0f && !ix->build_client && jvk_pqw_ok(&ix->dev, rk + 64 + ix->dev.R) && nq >= OPT(ix, OPT_PQW_MIN_QUERIES)) {
        JvSearchArgs ap = a;
        ap.cand_cap = rk + 64 + ix->dev.R;
        ap.pqp_log_cap = (12 * rk + 1024 + 3) & ~3;
        const int lds = plan_pqw_lds(ix, ap, 0);
        // (a few queries — e.g. one row the query server handed back — are not worth asking a live server grid to leave for:
        //  they go straight to the HBM-table rung, which fits beside it)
        if (lds <= kMaxLds && (nq > 64 || servers_leave_room(ix->device, lds))) {
            int blocks = ix->cu_count * jvk_pqw_blocks_per_cu(&ix->dev, ap.cand_cap, lds, 0);
            if (blocks > nq) blocks = nq;
            const size_t need = (size_t)blocks * (size_t)ap.pqp_log_cap;
            if (need > c->pqp_log_ints) {
                if (c->pqp_log) HIPCHK(jv_free(c->pqp_log));
                c->pqp_log = nullptr;
                c->pqp_log_ints = 0;
                HIPCHK(hipMalloc((void**)&c->pqp_log, need * sizeof(int32_t)));
                c->pqp_log_ints = need;
            }
            ap.pqp_log = c->pqp_log;
            ap.pqp_counter = c->work_counter + 6;
            servers_yield_lds(ix->device, lds);
            if ((rc = vis_attach(ap)) != JV_OK) return rc;
            const int kts = kt_begin(ix, c, stream);
            HIPCHK(jvk_launch_search_pqw(&ix->dev, &ap, lds, blocks, 0, stream));
            kt_end(c, kts, stream);
            if ((rc = vis_count(ap)) != JV_OK) return rc;
            ix->launches[LAUNCH_PQW]++;
            wide_first = true;
        }
    }
    bool first_ran = wide_first;  // a pool kernel has run: the rungs below redo flagged rows only
    if (phase != 2 && !wide_first) {
    // headline path: PQ + fused layout + single pool + flat graph -> specialised kernel without an in-loop
    // visited set (jv_kernels.hip "PQF"); anything it cannot hold is flagged and falls through to the ladder
    bool pqf = false;
    const bool filtered = d_accept != nullptr;
    // filtered variant: single-pass blocks, ordinals below 2^30 (one key bit carries "accepted")
    const bool pqf_shape = filtered ? (thr <= 0.0f && ix->dev.R * ix->dev.pq_lanes <= JV_WAVE && ix->dev.n < (1 << 30)) : g.pool;
    const bool pqf_index = pq && ix->dev.pq_fused && ix->dev.num_upper == 0 && ix->dev.R <= JV_WAVE && ix->dev.R * ix->dev.pq_lanes <= 4 * JV_WAVE &&
                           (ix->dev.R * ix->dev.pq_lanes + JV_WAVE - 1) / JV_WAVE <= ix->dev.pq_lanes;
    // headline path: the persistent LDS-pool kernel (jv_kernels_pqp.hip): no filter, threshold <= 0; LDS = look-up table +
    // pool of rk + 64 boundary ties + one expansion's R new keys (+ 1 sentinel slot); the expansion log lives in HBM
    // (pools of <= 256 entries stay on the round-1 kernel below: its 256-entry variant keeps masks and pivots in scalar
    //  registers and measured 1.5-3 % faster there; its cost grows with the pool, this kernel's does not)
    // With a doc filter the same kernel (instances in jv_kernels_pqpf.hip) holds every scored node scoring >= the rerankK-th
    // best ACCEPTED one, ~ rerankK / selectivity entries: first launch 1 024..2 048 entries (register table where it
    // applies), second launch 4 096; each wave estimates its filter's selectivity first and skips a launch it cannot fit.
    // (Pools of <= 256 entries keep round 1's filtered kernel with its 448 / 960-entry launches where the register-table
    //  variant does not apply.)
    // (pools of <= 256 entries: the register-table variant where it applies — measured 2.10 M vs 1.92 M QPS for round 1's
    //  kernel at rerankK = 160 —, else round 1's kernel, which is the faster one against the LDS-table variant there)
    const int64_t lutr_min_q = OPT(ix, OPT_LUTR_MIN_QUERIES) >= 0 ? OPT(ix, OPT_LUTR_MIN_QUERIES) : 4 * (int64_t)ix->cu_count;
    const bool lutr_applies = OPT(ix, OPT_NO_LUTR) == 0 && jvk_pqp_lutr_ok(&ix->dev, rk + 64 + ix->dev.R) && nq > lutr_min_q;
    const bool pqw_applies = !filtered && OPT(ix, OPT_NO_PQW) == 0 && jvk_pqw_ok(&ix->dev, rk + 64 + ix->dev.R) && nq >= OPT(ix, OPT_PQW_MIN_QUERIES);
    const bool pqp_plain = !filtered && g.pool && ix->dev.n < (1 << 30) &&
                           (rk + 64 + ix->dev.R > 256 || OPT(ix, OPT_LUTR_MIN_QUERIES) == 0 || lutr_applies || pqw_applies);
    // (measured at rerankK = 160, 2M docs, 65 536 queries per launch, selectivity 0.9 / 0.5 / 0.3 / 0.15: round 1's filtered
    //  kernel 1.33 M / 0.90 M / 0.27 M / 37 k QPS, this kernel's register-table variant 1.64 M / 1.07 M / 0.68 M / 227 k)
    const bool pqwf_applies = filtered && OPT(ix, OPT_NO_PQW) == 0 && jvk_pqwf_ok(&ix->dev, rk + 64 + ix->dev.R) && nq >= OPT(ix, OPT_PQW_MIN_QUERIES);
    // (the several-waves filtered kernel also takes PQ-64 — four lanes per neighbour do not fit the one-wave kernel's single pass)
    const bool pqp_filt = filtered && thr <= 0.0f && (pqf_shape || pqwf_applies) && ix->dev.n < (1 << 29) &&
                          (rk + 64 + ix->dev.R > 256 || OPT(ix, OPT_LUTR_MIN_QUERIES) == 0 || lutr_applies || pqwf_applies);
    // (shapes whose table leaves the GENERIC kernel no room — g.fast_ok false, e.g. PQ-128 at wide beams — still run the
    //  several-waves kernel, which keeps the table in registers + LDS rows)
    // (round 5: the default codecs — PQ-128 / PQ-192, tables beyond the LDS — WITH a doc filter run the several-waves filtered
    //  kernel first as well (jv_kernels_pqw12f.hip); the HBM-table rung only redoes what comes back flagged)
    const bool pqwf_wide = pqwf_applies && ix->dev.pq_M >= 128 && thr <= 0.0f && !ix->build_client;
    if ((!force_big || (OPT(ix, OPT_FORCE_BIG) == 0 && ((!g.lutg && pqw_applies) || pqwf_wide))) && (pqp_plain || pqp_filt) && (pqf_index || pqwf_wide) && rk + 64 + ix->dev.R <= jvk_pqp_max_entries() && OPT(ix, OPT_NO_PQF) == 0 &&
        OPT(ix, OPT_NO_PQP) == 0) {
        JvSearchArgs ap = a;
        if (filtered && a.accept_stride == 0 && ix->dev.ord2doc && nq >= 16) {
            // one filter for the whole batch over a doc-id mapping: translate it to ordinal space first (jv_kernels_pqpf.hip)
            const size_t words = ((size_t)ix->dev.n + 63) / 64;
            if (words > c->accept_ord_cap) {
                if (c->d_accept_ord) HIPCHK(jv_free(c->d_accept_ord));
                c->d_accept_ord = nullptr;
                c->accept_ord_cap = 0;
                HIPCHK(hipMalloc((void**)&c->d_accept_ord, words * 8));
                c->accept_ord_cap = words;
            }
            HIPCHK(jvk_launch_accept_to_ord(&ix->dev, a.accept, a.accept_docs, c->d_accept_ord, stream));
            ap.accept_ord = c->d_accept_ord;
        }
        ap.cand_cap = filtered ? std::min(2048, std::max(1024, 2 * (rk + 64 + ix->dev.R))) : rk + 64 + ix->dev.R;
        ap.pqp_log_cap = filtered ? 3 * ap.cand_cap : ((3 * rk + 64 + 3) & ~3);
        // Table in registers (8 resident queries per CU, +34 % throughput at rerankK = 1 200, but 1.7x the latency of
        // one query): only when the launch has more queries than the LDS-table variant could keep resident anyway
        const int lutr = (OPT(ix, OPT_NO_LUTR) == 0 && jvk_pqp_lutr_ok(&ix->dev, ap.cand_cap) && nq > lutr_min_q) ? 1 : 0;
        // LDS plan of one launch: offsets into the workgroup's LDS for a pool of x.cand_cap entries
        auto plan = [&](JvSearchArgs& x, bool regs) { return plan_pqp_lds(ix, x, regs, rk); };
        // Several waves per query (jv_kernels_pqw.hip: PQ-32 / PQ-64, unfiltered): table in registers split by chunk, one
        // wave per chunk.  LDS = [pool | centred query | hash set | rerank scratch] + the waves' exchange rows + ctrl words.
        const bool pqw = !filtered && OPT(ix, OPT_NO_PQW) == 0 && jvk_pqw_ok(&ix->dev, ap.cand_cap) && nq >= OPT(ix, OPT_PQW_MIN_QUERIES);
        // few queries: every resident query has a CU (almost) to itself and its time is the launch's — the variant with the
        // whole table in LDS (a third of the scoring pass's instructions, 3 workgroups per CU)
        const int64_t lat_q = OPT(ix, OPT_PQW_LAT_QUERIES) >= 0 ? OPT(ix, OPT_PQW_LAT_QUERIES) : 3 * (int64_t)ix->cu_count;
        // (diagnostics, JV_PQW_OCC5=1: the 96-register instance — nine resident queries per CU instead of eight, jv_kernels_pqw.hip)
        static const bool occ5_env = getenv("JV_PQW_OCC5") != nullptr;
        const int pqw_variant = (pqw && nq <= lat_q) ? 1 : ((pqw && occ5_env && jvk_pqw_occ5_ok(&ix->dev, ap.cand_cap + 64)) ? 2 : 0);
        if (pqw) {
            // the first launch keeps what it can: an expansion log four times as long (it lives in HBM) and, where the LDS
            // budget of the same residency and the same capacity class allow it, 128 instead of 64 boundary-tie slots —
            // every query it does not have to hand over saves the second launch (whose length is its slowest query's)
            ap.pqp_log_cap = (12 * rk + 1024 + 3) & ~3;
            JvSearchArgs wide = ap;
            wide.cand_cap = ap.cand_cap + 64;
            const int lds_now = plan_pqw_lds(ix, ap, pqw_variant), lds_wide = plan_pqw_lds(ix, wide, pqw_variant);
            auto klass = [](int cap) { return cap <= 512 ? 0 : cap <= 1024 ? 1 : 2; };
            if (jvk_pqw_ok(&ix->dev, wide.cand_cap) && klass(wide.cand_cap) == klass(ap.cand_cap) && kMaxLds / lds_wide == kMaxLds / lds_now)
                ap.cand_cap = wide.cand_cap;
        }
        // Doc filters on the several-waves kernel (jv_kernels_pqwf.hip, round 4): the same launches and rungs as the one-wave
        // filtered kernel below — 2 x the pool first, then the largest pools that keep 4, 3, 2, 1 workgroups per CU — with two /
        // four waves per query, two fused blocks per scoring pass and the table's rows in registers + LDS.
        const bool pqwf = filtered && OPT(ix, OPT_NO_PQW) == 0 && jvk_pqwf_ok(&ix->dev, ap.cand_cap) && nq >= OPT(ix, OPT_PQW_MIN_QUERIES);
        auto planF = [&](JvSearchArgs& x, bool regs) { return pqwf ? plan_pqw_lds(ix, x, 0, jvk_pqwf_lds_rows(&ix->dev, x.cand_cap)) : plan(x, regs); };
        auto blocksF = [&](int cap, int lds_b, int lutr_) { return pqwf ? jvk_pqwf_blocks_per_cu(&ix->dev, cap, lds_b) : jvk_pqp_blocks_per_cu(&ix->dev, cap, lds_b, lutr_, 1); };
        auto launchF = [&](const JvSearchArgs& x, int lds_b, int blocks_, int lutr_) {
            return pqwf ? jvk_launch_search_pqwf(&ix->dev, &x, lds_b, blocks_, stream) : jvk_launch_search_pqp(&ix->dev, &x, lds_b, blocks_, lutr_, stream);
        };
        const int max_entries_f = pqwf ? jvk_pqwf_max_entries(&ix->dev) : jvk_pqp_max_entries_filtered();
        const int lds = pqw ? plan_pqw_lds(ix, ap, pqw_variant) : (pqwf ? planF(ap, false) : plan(ap, lutr != 0));
        // second launch for what outgrows the first (more than 63 ties at the rerankK boundary, a longer expansion log):
        // table in LDS, as many tie slots as the largest pool class allows, 4x the log; walks the flag array
        JvSearchArgs ap2 = ap;
        ap2.cand_cap = std::min(jvk_pqp_max_entries(), std::max(2 * ap.cand_cap, rk + ix->dev.R + 1024));
        ap2.pqp_log_cap = filtered ? 3 * ap2.cand_cap : ((12 * rk + 1024 + 3) & ~3);
        ap2.retry_only = 1;
        ap2.retry_counter = c->work_counter + 5;
        // (filtered: the later launches keep the table in registers too — twice the resident queries per CU at these pool sizes)
        const int lutr2 = filtered && OPT(ix, OPT_NO_LUTR) == 0 && jvk_pqpf_lutr_ok(&ix->dev, jvk_pqp_max_entries_filtered()) && nq > lutr_min_q ? 1 : 0;
        const int lds2 = pqwf ? planF(ap2, false) : plan(ap2, lutr2 != 0);
        const bool second = lds2 <= kMaxLds && ap2.cand_cap > ap.cand_cap && OPT(ix, OPT_PQF_ONLY) == 0;
        if (lds <= kMaxLds) {
            int per_cu = pqw ? jvk_pqw_blocks_per_cu(&ix->dev, ap.cand_cap, lds, pqw_variant)
                             : (pqwf ? blocksF(ap.cand_cap, lds, 0) : jvk_pqp_blocks_per_cu(&ix->dev, ap.cand_cap, lds, lutr, filtered ? 1 : 0));
            if (OPT(ix, OPT_PQP_BLOCKS_PER_CU) > 0) per_cu = (int)std::min<int64_t>(per_cu, OPT(ix, OPT_PQP_BLOCKS_PER_CU));  // diagnostics
            int blocks = ix->cu_count * per_cu;
            if (blocks > nq) blocks = nq;
            // (unfiltered: a handful of flagged queries; filtered: a selective filter sends the whole batch here)
            const int per_cu2 = filtered && second ? blocksF(ap2.cand_cap, lds2, lutr2) : 1;
            const int blocks2 = std::min(ix->cu_count * per_cu2, (nq + 7) / 8);
            // Filtered searches: a pool of ~ rerankK / selectivity entries.  What a launch costs is decided by how many of its
            // queries a CU keeps resident, and that is LDS / (pool bytes + fixed part).  Behind the second launch (twice the
            // first's pool) the rungs are therefore the LARGEST pools that still fit 4, 3, 2 (, 1) workgroups per CU — each
            // query runs on the first rung its estimated pool fits (the kernel's selectivity estimate skips the others at
            // once), e.g. selectivity 0.2 at rerankK 1 200 with three resident queries per CU instead of the 8 192-entry
            // rung's two.  (More than 4 per CU is not offered: the visited-count hash set lives in what LDS the final
            // compaction frees, and pools that small leave it too few slots — measured slower; and the wide-pool instances
            // are compiled for one wave per SIMD so that nothing spills to scratch.)
            struct Rung { JvSearchArgs a; int lds, blocks; };
            Rung rungs[8];
            int nrungs = 0;
            if (filtered && second) {
                Rung r0;
                r0.a = ap2;
                r0.lds = lds2;
                r0.blocks = std::min(ix->cu_count * per_cu2, (nq + 1) / 2);
                rungs[nrungs++] = r0;
                JvSearchArgs probe = ap2;
                probe.cand_cap = 4096;
                const int fixed = planF(probe, lutr2 != 0) - 4097 * 8;  // LDS bytes besides the pool (class 3 / 4 layout)
                int prev = ap2.cand_cap;
                for (int per = 4; per >= 1 && nrungs < 8; per--) {  // (these instances take 512 registers: one wave per SIMD)
                    // (measured with tools/lds_residency.hip: workgroups of 7 x 23 040 B and 3 x 53 248 B are resident together on a
                    //  gfx950 CU, 5 x 32 768 B, 10 x 16 384 B and 3 x 54 528 B are NOT although the occupancy API says so: what
                    //  several workgroups can share is a little less than 160 KB — budget 157.5 KB)
                    const int share = ((161280 / per) & ~255);
                    int cap = std::min(((share - fixed) / 8 - 1) & ~63, max_entries_f);
                    if (cap < prev + 512 || cap < rk + 64 + ix->dev.R) continue;  // (not worth a launch of its own)
                    Rung r;
                    r.a = ap2;
                    r.a.cand_cap = cap;
                    r.lds = planF(r.a, lutr2 != 0);
                    if (r.lds > kMaxLds) continue;
                    r.a.pqp_log_cap = 3 * cap;
                    r.a.retry_counter = c->work_counter + 8 + nrungs;
                    r.blocks = std::min(ix->cu_count * std::min(per, blocksF(cap, r.lds, lutr2)), (nq + 1) / 2);
                    rungs[nrungs++] = r;
                    prev = cap;
                    if (cap >= max_entries_f) break;
                }
                rungs[nrungs - 1].a.retry_only = 2;  // the last on-chip rung never skips on the selectivity estimate
            }
            size_t need = (size_t)blocks * (size_t)ap.pqp_log_cap;
            if (second && nrungs == 0) need = std::max(need, (size_t)blocks2 * (size_t)ap2.pqp_log_cap);
            for (int i = 0; i < nrungs; i++) need = std::max(need, (size_t)rungs[i].blocks * (size_t)rungs[i].a.pqp_log_cap);
            if (need > c->pqp_log_ints) {
                if (c->pqp_log) HIPCHK(jv_free(c->pqp_log));
                c->pqp_log = nullptr;
                c->pqp_log_ints = 0;
                HIPCHK(hipMalloc((void**)&c->pqp_log, need * sizeof(int32_t)));
                c->pqp_log_ints = need;
            }
            ap.pqp_log = c->pqp_log;
            ap.pqp_counter = c->work_counter + 6;
            const bool second_now = second && nrungs == 0 && retry_rung_ok(lds2);
            {
                int lds_max = lds;
                for (int i = 0; i < nrungs; i++) lds_max = std::max(lds_max, rungs[i].lds);
                servers_yield_lds(ix->device, lds_max);
            }
            if (pqw) {
                if ((rc = vis_attach(ap)) != JV_OK) return rc;
                const int kts = kt_begin(ix, c, stream);
                HIPCHK(jvk_launch_search_pqw(&ix->dev, &ap, lds, blocks, pqw_variant, stream));
                kt_end(c, kts, stream);
                if ((rc = vis_count(ap)) != JV_OK) return rc;
            } else {
                // (doc filters on the several-waves kernel: the first launch and every rung behind it hand their logs to the same arena)
                if (pqwf) {
                    if ((rc = vis_attach(ap)) != JV_OK) return rc;
                    auto share = [&](JvSearchArgs& y) {
                        y.vis_arena = ap.vis_arena, y.vis_cap_units = ap.vis_cap_units, y.vis_cursor = ap.vis_cursor, y.vis_off = ap.vis_off, y.vis_n = ap.vis_n;
                    };
                    share(ap2);
                    for (int i = 0; i < nrungs; i++) share(rungs[i].a);
                }
                const int kts = kt_begin(ix, c, stream);
                if (pqwf) HIPCHK(launchF(ap, lds, blocks, 0));
                else HIPCHK(jvk_launch_search_pqp(&ix->dev, &ap, lds, blocks, lutr, stream));
                kt_end(c, kts, stream);
            }
            ix->launches[(pqw || pqwf) ? LAUNCH_PQW : LAUNCH_PQP]++;
            trace_point(c, 5, stream);
            if (second_now) {
                ap2.pqp_log = c->pqp_log;
                if (pqwf) HIPCHK(launchF(ap2, lds2, blocks2, lutr2));
                else HIPCHK(jvk_launch_search_pqp(&ix->dev, &ap2, lds2, blocks2, lutr2, stream));
            }
            for (int i = 0; i < nrungs; i++) {
                rungs[i].a.pqp_log = c->pqp_log;
                HIPCHK(launchF(rungs[i].a, rungs[i].lds, rungs[i].blocks, lutr2));
            }
            if (pqwf && (rc = vis_count(ap)) != JV_OK) return rc;
            pqf = true;
        }
    }
    if (!pqf && !force_big && pqf_index && pqf_shape && rk + 64 + ix->dev.R <= 1024 && OPT(ix, OPT_NO_PQF) == 0) {
        JvSearchArgs ap = a;
        const int lut_b = ix->dev.pq_M * 256 * 4;
        ap.cand_cap = (rk + 64 + ix->dev.R + 1) & ~1;  // pool entries: rk + 64 boundary ties + one merge of <= R new keys
        ap.res_cap = (3 * rk + 64 + 3) & ~3;        // expansion log entries
        bool shape_ok = true;
        bool second_rung = false;
        if (filtered) {
            // the pool must hold every node scoring >= the rk-th best ACCEPTED one (~ rk / selectivity entries) and
            // the search expands about as many.  First launch: 448 entries (4 resident queries per CU with a 32 KB
            // LUT); queries that outgrow it are redone by a second launch with 960 entries before the generic ladder.
            const int need = ap.cand_cap;
            const bool big_lut = lut_b > 32768;
            const int want = (need <= 448) ? 448 : 960;
            shape_ok = need <= want && !(big_lut && want > 448);
            second_rung = want == 448 && !big_lut;
            ap.cand_cap = want;
            ap.res_cap = want == 448 ? 1024 : 2048;
        }
        const int qc_b = ix->dev.nch * 64 * 4;
        const int rerank_b = qc_b + JV_TODO * 8 + ((rk + 1) & ~1) * 8;                  // after the search, inside the LUT area
        auto loop_bytes = [&](const JvSearchArgs& x) {
            int b_ = lut_b + x.cand_cap * 8 + x.res_cap * 4;
            if (x.cand_cap * 8 + x.res_cap * 4 < qc_b) b_ = lut_b + qc_b;               // LUT build aliases pool + log
            return b_;
        };
        if (shape_ok && rerank_b <= lut_b && loop_bytes(ap) <= kMaxLds) {
            servers_yield_lds(ix->device, (loop_bytes(ap) + 15) & ~15);
            HIPCHK(jvk_launch_search_pqf(&ix->dev, &ap, (loop_bytes(ap) + 15) & ~15, stream));
            ix->launches[LAUNCH_PQF]++;
            pqf = true;
            if (second_rung && OPT(ix, OPT_PQF_ONLY) == 0) {
                JvSearchArgs ap2 = ap;
                ap2.cand_cap = 960;
                ap2.res_cap = 2048;
                ap2.retry_only = 1;
                ap2.retry_counter = c->work_counter + 5;
                if (loop_bytes(ap2) <= kMaxLds && retry_rung_ok((loop_bytes(ap2) + 15) & ~15)) HIPCHK(jvk_launch_search_pqf(&ix->dev, &ap2, (loop_bytes(ap2) + 15) & ~15, stream));
            }
        }
    }
    if (pqf && OPT(ix, OPT_PQF_ONLY) != 0) return JV_OK;
    first_ran = first_ran || pqf;
    if (!force_big) {
        if (!pqf) {
            servers_yield_lds(ix->device, g.lds_fast);
            const int kts = kt_begin(ix, c, stream);
            HIPCHK(jvk_launch_search_lds(&ix->dev, &a, pq ? 1 : 0, g.pool ? 1 : 0, ix->build_client ? 2 : 0, g.lds_fast, stream));
            kt_end(c, kts, stream);
            ix->launches[LAUNCH_LDS]++;
        }
        // (after a PQF launch the flagged queries go straight to the rung below: generic kernel, 4x visited table)
        // escalation: queries that overflowed the on-chip visited set are retried with a 4x larger table
        // (fewer resident queries, but only the flagged few run) before the HBM-scratch path
        if (OPT(ix, OPT_LDS_VISITED_SLOTS) <= 0 && OPT(ix, OPT_NO_ESCALATION) == 0) {
            Geometry g2 = plan_geometry(ix, rk, g.pool, g.hash_slots * 4, thr > 0.0f);
            if (g2.fast_ok && g2.hash_slots > g.hash_slots) {
                JvSearchArgs a2 = a;
                a2.hash_slots = g2.hash_slots;
                a2.cand_cap = g2.cand_cap;
                a2.res_cap = g2.res_cap;
                a2.retry_only = 1;
                a2.retry_counter = c->work_counter + 4;
                if (!pqf || retry_rung_ok(g2.lds_fast)) {  // (behind the generic first launch it stays: that one already needed the room)
                    servers_yield_lds(ix->device, g2.lds_fast);
                    HIPCHK(jvk_launch_search_lds(&ix->dev, &a2, pq ? 1 : 0, g2.pool ? 1 : 0, 1, g2.lds_fast, stream));
                }
            }
        }
    }
    }
    trace_point(c, 6, stream);
    if (phase == 1 && (!force_big || first_ran)) {
        if (big_deferred) *big_deferred = true;
        return JV_OK;
    }
    {
        // last rung: the device's shared HBM scratch; its users are ordered through the scratch event
        DeviceScratch& sc = g_scratch[ix->device & 63];
        std::lock_guard<std::mutex> lk(sc.mu);
        int my_cap = 0, my_blocks = 0;
        rc = ensure_big(ix, sc, rk, &my_cap, &my_blocks);
        if (rc != JV_OK) return rc;
        a.big_visited = sc.big_visited;  // (block b uses [b * words, ...) of the bitsets with words = ceil(n / 32) of THIS index)
        a.big_cand = sc.big_cand;
        a.big_cand_cap = my_cap;
        a.res_cap = (rk + 1) & ~1;  // the rung runs the two-queue form: rerankK results, the rest of my_cap are candidates
        if (g.lutg) {
            // the table in HBM scratch: 1 KB per subspace per resident workgroup (at most 1 024 of them: 192 MB at pq_M = 192)
            my_blocks = std::min(my_blocks, 1024);
            const size_t need = (size_t)my_blocks * (size_t)ix->dev.pq_M * 256 * sizeof(float);
            if (need > sc.lut_bytes) {
                HIPCHK(hipEventSynchronize(sc.last_use));
                if (sc.lut) HIPCHK(jv_free(sc.lut));
                sc.lut = nullptr;
                sc.lut_bytes = 0;
                HIPCHK(hipMalloc((void**)&sc.lut, need));
                sc.lut_bytes = need;
            }
            a.lut_scratch = sc.lut;
            HIPCHK(hipStreamWaitEvent(stream, sc.last_use, 0));
            // as below: first with both queues in LDS (the table is not there, so nearly all of it is free: four resident
            // queries per CU), then in HBM for what outgrew them
            const int fixed_g = (g.lds_big + 15) & ~15;
            const int qslots_g = (kMaxLds / 4 - fixed_g - 256) / 8;
            servers_yield_lds(ix->device, fixed_g + std::max(0, qslots_g) * 8);
            if (qslots_g >= a.res_cap + 4 * rk + 256 && OPT(ix, OPT_NO_ESCALATION) == 0) {
                JvSearchArgs aq = a;
                aq.cand_cap = qslots_g - aq.res_cap;
                aq.work_counter = c->work_counter + 7;
                const int all = (force_big && !first_ran && phase != 2) ? 1 : 0;  // (after the several-waves launch, or in phase 2: flagged rows only)
                HIPCHK(jvk_launch_search_big_lutg(&ix->dev, &aq, my_blocks, fixed_g + qslots_g * 8, all, 1, stream));
                HIPCHK(jvk_launch_search_big_lutg(&ix->dev, &a, my_blocks, fixed_g, 0, 0, stream));
            } else {
                HIPCHK(jvk_launch_search_big_lutg(&ix->dev, &a, my_blocks, fixed_g, (force_big && !first_ran && phase != 2) ? 1 : 0, 0, stream));
            }
            HIPCHK(hipEventRecord(sc.last_use, stream));
            return JV_OK;
        }
        HIPCHK(hipStreamWaitEvent(stream, sc.last_use, 0));
        // (every row when nothing ran before this rung; flagged rows only behind a pool kernel, and in phase 2 — whose phase 1
        //  either ran one or, forced here, did everything itself)
        const int all_rows = (force_big && !first_ran && phase != 2) ? 1 : 0;
        // first with both queues in LDS (every slot the workgroup's LDS has left; the visited set is the HBM bitset) ...
        // (this rung is enqueued behind every device-pointer batch, flagged rows or not: it must not wait for — nor displace —
        //  a resident query-server grid, so the LDS variant takes what the live grids leave free on a CU)
        const int fixed_lds = (g.lds_big + 15) & ~15;
        const int lds_q = std::min(kMaxLds, servers_free_lds(ix->device)) & ~15;
        const int qslots = (lds_q - fixed_lds) / 8;
        servers_yield_lds(ix->device, fixed_lds);
        if (qslots >= a.res_cap + 4 * rk + 256 && OPT(ix, OPT_NO_ESCALATION) == 0) {
            JvSearchArgs aq = a;
            aq.cand_cap = qslots - aq.res_cap;
            aq.work_counter = c->work_counter + 7;
            HIPCHK(jvk_launch_search_big(&ix->dev, &aq, pq ? 1 : 0, my_blocks, lds_q, all_rows, 1, stream));
            // ... then, for what outgrew them (and for the rerankFloor corner that needs the admission log), in HBM
            HIPCHK(jvk_launch_search_big(&ix->dev, &a, pq ? 1 : 0, my_blocks, g.lds_big, 0, 0, stream));
        } else {
            HIPCHK(jvk_launch_search_big(&ix->dev, &a, pq ? 1 : 0, my_blocks, g.lds_big, all_rows, 0, stream));
        }
        HIPCHK(hipEventRecord(sc.last_use, stream));
    }
    return JV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-resident query server (kernel: jv_serve_pqw_kernel, jv_pqw_body.h).  One-query calls on shapes the several-waves
// pool kernel runs (PQ-32 / PQ-64 fused, no filter, threshold <= 0) do not launch anything: the caller copies its query into
// a slot of a pinned ring, publishes it in ticket order and sleeps until the slot's completion word is set.  The grid is
// started on demand, leaves by itself after `serve_idle_ms` without work and is asked to leave (STOP) before anything in this
// library synchronises the device or frees memory.
// ---------------------------------------------------------------------------------------------------------------------
}  // namespace
struct JvQueryServer {
    jv_index* ix = nullptr;
    hipStream_t stream = nullptr;
    unsigned char* ring = nullptr;  // pinned, device-visible: [slots][slot_bytes]
    int32_t* h_words = nullptr;     // pinned: JV_SH_*
    int32_t* d_words = nullptr;     // device: JV_SV_*
    int32_t* log = nullptr;         // device: expansion logs of the resident workgroups
    int slots = 0, slot_bytes = 0, cap_max = 0, blocks = 0, lds = 0;
    int kind = 0;                   // 0: unfiltered queries on the several-waves kernel; 1: queries with a doc filter (one-wave filtered pool kernel)
    int lutr = 0;                   // kind 1: look-up table in registers
    int waves_f = 0;                // kind 1: the several-waves filtered kernel serves (PQ-32 / PQ-64), else the one-wave one
    hipStream_t up_stream = nullptr;  // kind 1: filter uploads of cache misses (the grid's own stream never drains)
    JvSearchArgs args{};
    std::atomic<uint32_t> reserve{0};
    std::atomic<uint32_t>* slot_free = nullptr;  // slot i may be filled by the call holding sequence number slot_free[i]
    std::mutex mu;                  // launch / stop
    std::atomic<int> inflight{0};
    std::atomic<int> lat_us{3000};  // running estimate of one query's latency (how long a caller sleeps before it polls)
    std::atomic<int> waiters{0};    // callers waiting for their completion word right now
    int spin_waiters = 4;           // up to this many of them poll with sched_yield() behind their first nap (option serve_spin_waiters)
    std::atomic<int64_t> last_call_ms{0};  // when a one-query call last took a slot (servers_free_lds: "may be restarted any moment")
};
namespace {
typedef JvQueryServer Server;

std::mutex g_servers_mu;
std::vector<Server*> g_servers;  // every live server of the process (device-wide synchronisation points pause them)

static bool serve_trace() {
    static const bool on = getenv("JV_SERVE_TRACE") != nullptr;  // diagnostics: who starts and stops the resident grids
    return on;
}
void server_stop_locked(Server* sv, const char* why = "pause") {  // sv->mu held: ask the grid to leave and wait until it has
    if (!sv->stream) return;
    const bool was_alive = __atomic_load_n(&sv->h_words[JV_SH_ALIVE], __ATOMIC_ACQUIRE) != 0;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    __atomic_store_n(&sv->h_words[JV_SH_STOP], 1, __ATOMIC_RELEASE);
    hipStreamSynchronize(sv->stream);
    __atomic_store_n(&sv->h_words[JV_SH_ALIVE], 0, __ATOMIC_RELEASE);
    __atomic_store_n(&sv->h_words[JV_SH_STOP], 0, __ATOMIC_RELEASE);
    if (serve_trace() && was_alive) {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "[jvgpu serve] grid kind %d stopped (%s) in %.2f ms\n", sv->kind, why,
                (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6);
    }
}

// RAII: no query server runs on `device` while this object lives (hipFree / hipDeviceSynchronize would otherwise wait for
// a grid that only leaves when it is idle).  Callers that arrive meanwhile wait on the servers' launch mutexes.
// Re-entrant per thread: a batched exact call keeps the device's servers paused while it runs (xb_pause_servers) and grows its
// buffers inside — jv_free -> free_with_servers_paused would take the same launch mutexes again on the same thread.
static thread_local int t_pause_depth = 0, t_pause_device = -1;
struct ServerPause {
    std::vector<Server*> held;
    enum { NESTED, OUTER, PLAIN } mode = PLAIN;   // NESTED: this thread already holds the device's pause; OUTER: this object marks the thread
    explicit ServerPause(int device) {
        if (t_pause_depth > 0 && t_pause_device == device) {
            mode = NESTED;
            t_pause_depth++;
            return;
        }
        std::lock_guard<std::mutex> g(g_servers_mu);
        for (Server* sv : g_servers)
            if (sv->ix->device == device) {
                sv->mu.lock();
                server_stop_locked(sv);
                held.push_back(sv);
            }
        if (t_pause_depth == 0) {   // (a pause on ANOTHER device inside a pause stays PLAIN: locks only)
            mode = OUTER;
            t_pause_depth = 1;
            t_pause_device = device;
        }
    }
    ~ServerPause() {
        if (mode == NESTED) {
            t_pause_depth--;
            return;
        }
        for (Server* sv : held) sv->mu.unlock();
        if (mode == OUTER) t_pause_depth = 0, t_pause_device = -1;
    }
    ServerPause(const ServerPause&) = delete;
    ServerPause& operator=(const ServerPause&) = delete;
};

void free_with_servers_paused(void* p, bool host) {
    if (!p) return;
    int dev = 0;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.device >= 0) dev = attr.device;
    else if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    bool any = false;
    // (a thread that already holds the device's pause must not touch g_servers_mu: servers_yield_lds takes g_servers_mu and THEN a
    //  server's launch mutex — a batch caller waiting there for the mutex this thread holds would never release the list lock)
    if (!(t_pause_depth > 0 && t_pause_device == dev)) {
        std::lock_guard<std::mutex> g(g_servers_mu);
        any = !g_servers.empty();
    }
    if (any) {
        ServerPause pause(dev);
        if (host) hipHostFree(p);
        else hipFree(p);
    } else {
        if (host) hipHostFree(p);
        else hipFree(p);
    }
}

// g_servers_mu held: LDS bytes per CU the server grids of this device occupy.  alive_only = false counts every EXISTING server:
// a grid that is not running now may be started by the next one-query call — before a kernel enqueued now gets its CUs
static int64_t now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}
// alive_only = false: also the grids that MAY start before a kernel enqueued now gets its CUs — servers that answered a call
// within the last second (a server nobody has called for longer is not about to be restarted: counting every EXISTING server,
// as round 4 did, switched the retry rungs off for the rest of the process after the first one-query call — ADVICE r4)
static int servers_held_lds_locked(int device, bool alive_only = true) {
    int held = 0;
    const int64_t now = alive_only ? 0 : now_ms();
    for (Server* sv : g_servers)
        if (sv->ix->device == device && sv->h_words &&
            (__atomic_load_n(&sv->h_words[JV_SH_ALIVE], __ATOMIC_ACQUIRE) != 0 || (!alive_only && now - sv->last_call_ms.load(std::memory_order_relaxed) < 1000)))
            held += (sv->blocks / std::max(1, sv->ix->cu_count)) * sv->lds;
    return held;
}
int servers_free_lds(int device) {
    std::lock_guard<std::mutex> g(g_servers_mu);
    const int held = servers_held_lds_locked(device, false);
    return held == 0 ? kMaxLds : std::max(0, 161280 - held - 8192);  // (8 KB of slack: allocation granules)
}
bool servers_leave_room(int device, int lds_needed) {
    std::lock_guard<std::mutex> g(g_servers_mu);
    const int held = servers_held_lds_locked(device);
    return held == 0 || lds_needed <= 161280 - held;  // (161 280: what a CU's workgroups can share, tools/lds_residency.hip)
}
void servers_yield_lds(int device, int lds_needed) {
    std::lock_guard<std::mutex> g(g_servers_mu);
    const int held = servers_held_lds_locked(device);
    if (held == 0 || lds_needed <= 161280 - held) return;
    for (Server* sv : g_servers)
        if (sv->ix->device == device) {
            std::lock_guard<std::mutex> lk(sv->mu);
            server_stop_locked(sv, "a launch needs its LDS");
        }
}

void server_destroy_one(jv_index* ix, Server*& ref) {
    Server* sv = ref;
    if (!sv) return;
    {
        std::lock_guard<std::mutex> g(g_servers_mu);
        g_servers.erase(std::remove(g_servers.begin(), g_servers.end(), sv), g_servers.end());
    }
    {
        std::lock_guard<std::mutex> lk(sv->mu);
        server_stop_locked(sv);
    }
    if (sv->stream) hipStreamDestroy(sv->stream);
    if (sv->up_stream) hipStreamDestroy(sv->up_stream);
    // (this server is off the list and its grid has left: the frees below pause the OTHER servers of the device)
    jv_host_free(sv->ring);
    jv_host_free(sv->h_words);
    jv_free(sv->d_words);
    jv_free(sv->log);
    delete[] sv->slot_free;
    delete sv;
    ref = nullptr;
}
void server_destroy(jv_index* ix) {
    server_destroy_one(ix, ix->server);
    server_destroy_one(ix, ix->server_f);
}

int server_launch_locked(Server* sv) {  // sv->mu held, grid not alive
    jv_index* ix = sv->ix;
    HIPCHK(hipSetDevice(ix->device));
    // tickets continue where the last grid stopped: HEAD / PUBLISHED stay, the exit count and the idle clock restart
    HIPCHK(hipMemsetAsync(sv->d_words + JV_SV_LOCK, 0, 4 * sizeof(int32_t), sv->stream));  // LOCK, EXITED, LAST_CLAIM, STOP_SEEN
    __atomic_store_n(&sv->h_words[JV_SH_ALIVE], 1, __ATOMIC_RELEASE);
    hipError_t e = sv->kind == 0 ? jvk_launch_serve_pqw(&ix->dev, &sv->args, sv->lds, sv->blocks, sv->stream)
                   : (sv->waves_f ? jvk_launch_serve_pqwf(&ix->dev, &sv->args, sv->lds, sv->blocks, sv->stream)
                                  : jvk_launch_serve_pqpf(&ix->dev, &sv->args, sv->lds, sv->blocks, sv->lutr, sv->stream));
    if (e != hipSuccess) {
        __atomic_store_n(&sv->h_words[JV_SH_ALIVE], 0, __ATOMIC_RELEASE);
        return fail(JV_EDEVICE, "query server launch failed: %s", hipGetErrorString(e));
    }
    ix->launches[LAUNCH_SERVE]++;
    if (serve_trace()) fprintf(stderr, "[jvgpu serve] grid kind %d launched (%d workgroups, %d B of LDS each)\n", sv->kind, sv->blocks, sv->lds);
    return JV_OK;
}

// the index's server, able to hold pools of `need_cap` entries; nullptr (with *rc set) when it cannot be provided
Server* server_get(jv_index* ix, int kind, int need_cap, int* rc) {
    *rc = JV_OK;
    std::unique_lock<std::mutex> lk(ix->server_mu);
    Server*& ref = kind == 0 ? ix->server : ix->server_f;
    // (the caller's reference is taken HERE, under ix->server_mu: a concurrent call that needs a larger pool waits for
    //  inflight == 0 under the same lock before it frees the server — counted after the lock was dropped, a caller could be
    //  left writing into a freed ring)
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
    for (;;) {
        if (ref && ref->cap_max >= need_cap) {
            ref->inflight++;
            return ref;
        }
        if (!ref) break;
        // a larger beam than the ring was planned for: rebuild it once nothing is in flight.  The wait RELEASES ix->server_mu
        // (round 4 spun on `inflight` with sched_yield while holding it: every one-query caller of the index stalled for a whole
        // in-flight query, without a bound); callers that arrive meanwhile still use the old ring.  Under steady traffic on the
        // old ring the deadline passes and THIS call takes the launch path — no caller waits for ever.
        if (ref->inflight.load() == 0) {
            server_destroy_one(ix, ref);
            break;
        }
        if (ix->server_cv.wait_until(lk, std::min(deadline, std::chrono::steady_clock::now() + std::chrono::milliseconds(2))) == std::cv_status::timeout &&
            std::chrono::steady_clock::now() >= deadline)
            return nullptr;  // (rc = JV_OK: "no server for this call" -> launch path)
    }
    if (hipSetDevice(ix->device) != hipSuccess) {
        *rc = fail(JV_EDEVICE, "hipSetDevice failed");
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> g(g_servers_mu);
        int same = 0;
        for (Server* o : g_servers) same += o->ix->device == ix->device ? 1 : 0;
        if (same >= 4) return nullptr;  // (one hardware queue per grid, BOTH kinds share the priority level: see the stream's creation below; such calls take the launch path)
    }
    Server* sv = new Server();
    sv->ix = ix;
    sv->kind = kind;
    sv->spin_waiters = (int)std::max<int64_t>(0, OPT(ix, OPT_SERVE_SPIN_WAITERS));
    auto bail = [&](const char* what, hipError_t e) -> Server* {
        *rc = fail(e == hipErrorOutOfMemory ? JV_ENOMEM : JV_EDEVICE, "query server: %s: %s", what, hipGetErrorString(e));
        ref = sv;
        server_destroy_one(ix, ref);
        return nullptr;
    };
    const int R = ix->dev.R;
    JvSearchArgs& a = sv->args;
    a = JvSearchArgs{};
    a.topK = 1;
    a.nq = 1;
    a.no_prescore = OPT(ix, OPT_NO_PRESCORE) != 0 ? 1 : 0;
    int per_cu;
    if (kind == 0) {
        sv->cap_max = need_cap <= 512 ? 512 : (need_cap <= 1024 ? 1024 : 2048);  // (the kernel's capacity classes: at most two rebuilds per index)
        a.cand_cap = sv->cap_max;
        a.rk = sv->cap_max - 64 - R;
        a.pqp_log_cap = (3 * a.rk + 64 + 3) & ~3;
        sv->lds = plan_pqw_lds(ix, a, 1);
        per_cu = std::min<int>(jvk_pqs_blocks_per_cu(&ix->dev, sv->cap_max, sv->lds), (int)std::max<int64_t>(1, OPT(ix, OPT_SERVE_WGS_PER_CU)));
    } else {
        // one pool of the filtered instances' class 4 (~5 800 entries ~ rerankK / selectivity next to a PQ-32 table; what does not
        // fit comes back flagged and takes the launch path's rungs).  Table in LDS, not in registers: a served query's time is its caller's latency, and
        // one wave gathers from an LDS table 1.4x faster than it permutes registers (measured: 7.4 vs 10.4 ms per query at
        // selectivity 0.5, rerankK 1 200) — two resident queries per CU instead of four, which only matters beyond 512 callers
        sv->lutr = 0;
        // Round 4: PQ-32 / PQ-64 indexes are served by the several-waves filtered kernel (two / four waves per query, whole table
        // in LDS: its latency variant) — the same one pool, sized the same way.
        sv->waves_f = (OPT(ix, OPT_NO_PQW) == 0 && jvk_pqwf_ok(&ix->dev, 4097)) ? 1 : 0;
        auto plan_f = [&](JvSearchArgs& x, int rk_) { return sv->waves_f ? plan_pqw_lds(ix, x, 1) : plan_pqp_lds(ix, x, false, rk_); };
        {
            // the largest pool that keeps TWO queries resident per CU next to this index's table (one if even 4 097 entries do not)
            JvSearchArgs probe = a;
            probe.cand_cap = 4097;
            const int fixed = plan_f(probe, 4097 - 64 - R) - 4098 * 8;
            int cap = 0;
            for (int per = 2; per >= 1 && cap < 4097; per--)
                cap = std::min(((161280 / per - fixed - 256) / 8 - 1) & ~63, sv->waves_f ? jvk_pqswf_max_entries() : jvk_pqsf_max_entries());
            sv->cap_max = cap;
        }
        a.cand_cap = sv->cap_max;
        a.rk = sv->cap_max - 64 - R;
        a.pqp_log_cap = 3 * sv->cap_max;
        sv->lds = sv->cap_max >= 4097 ? plan_f(a, a.rk) : kMaxLds + 1;
        if (sv->lds > kMaxLds) {
            delete sv;
            *rc = JV_OK;
            return nullptr;  // (this PQ shape's table leaves no room for the pool: launch path)
        }
        // (what a CU really keeps resident: tools/lds_residency.hip — a little less than 160 KB can be shared)
        per_cu = std::max(1, std::min<int>(std::min(sv->waves_f ? jvk_pqswf_blocks_per_cu(&ix->dev, sv->lds) : jvk_pqsf_blocks_per_cu(&ix->dev, sv->lds, sv->lutr), 161280 / sv->lds),
                                           (int)std::max<int64_t>(1, 2 * OPT(ix, OPT_SERVE_WGS_PER_CU))));  // (twice the unfiltered server's workgroups where the LDS allows)
    }
    sv->blocks = ix->cu_count * per_cu;
    sv->slots = next_pow2(std::max(1024, 2 * sv->blocks));
    const int qbytes = ((ix->dev.nch * 64 * 4) + 255) & ~255;
    sv->slot_bytes = (JV_SERVE_QUERY_OFF + qbytes + 255) & ~255;
    // The grid never ends, and the HIP runtime multiplexes streams onto a few hardware queues PER PRIORITY LEVEL
    // (GPU_MAX_HW_QUEUES = 4): on a default-priority stream it sat in front of every other stream that happened to share its
    // queue — a batch call then waited until the one-query traffic stopped.  The grids therefore run on streams of their own
    // priority level (the lowest; nothing else in this library uses it), and at most four per device — both kinds together —
    // are started (a fifth would share a queue with — and wait for ever behind — another).
    int prio_least = 0, prio_greatest = 0;
    hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t e = hipStreamCreateWithPriority(&sv->stream, hipStreamNonBlocking, prio_least);
    if (e != hipSuccess) return bail("stream", e);
    if (kind == 1 && (e = hipStreamCreateWithFlags(&sv->up_stream, hipStreamNonBlocking)) != hipSuccess) return bail("upload stream", e);
    if ((e = hipHostMalloc((void**)&sv->ring, (size_t)sv->slots * sv->slot_bytes, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail("ring", e);
    if ((e = hipHostMalloc((void**)&sv->h_words, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail("host words", e);
    memset(sv->h_words, 0, 64);
    memset(sv->ring, 0, (size_t)sv->slots * sv->slot_bytes);
    if ((e = hipMalloc((void**)&sv->d_words, 64)) != hipSuccess) return bail("device words", e);
    if ((e = hipMemset(sv->d_words, 0, 64)) != hipSuccess) return bail("device words", e);
    if ((e = hipMalloc((void**)&sv->log, (size_t)sv->blocks * (size_t)a.pqp_log_cap * sizeof(int32_t))) != hipSuccess) return bail("logs", e);
    sv->slot_free = new std::atomic<uint32_t>[(size_t)sv->slots];
    for (int i = 0; i < sv->slots; i++) sv->slot_free[i].store((uint32_t)i);
    a.pqp_log = sv->log;
    a.serve_ring = sv->ring;
    a.serve_slots = sv->slots;
    a.serve_slot_bytes = sv->slot_bytes;
    a.serve_dev = sv->d_words;
    a.serve_host = sv->h_words;
    a.serve_idle_ticks = (int32_t)std::min<int64_t>(2000000000, std::max<int64_t>(1, OPT(ix, OPT_SERVE_IDLE_MS)) * 100000);  // 100 MHz
    a.done_all = 1;
    ref = sv;
    {
        std::lock_guard<std::mutex> g(g_servers_mu);
        g_servers.push_back(sv);
    }
    sv->inflight++;  // (the caller's reference, as above)
    return sv;
}

// one query through the server.  Returns JV_OK with the row filled, a negative code, or +1 when the caller should take the
// launch path instead (not eligible, or the row came back flagged for the ladder).
int filter_acquire(jv_index* ix, const uint64_t* words, size_t nwords, uint64_t key, hipStream_t stream, const uint64_t** d_out, int* slot);
void filter_release(jv_index* ix, int slot);

// accept_words != nullptr: a query with a doc filter (the filtered server; the bits are served from the index's filter cache)
int serve_query(jv_index* ix, const float* query, int32_t topK, int32_t rerankK, float rerankFloor, int64_t visit_limit,
                int32_t* out_nodes, int32_t* out_docs, float* out_scores, int32_t* out_count, int32_t* out_stats, int32_t* out_flags,
                const uint64_t* accept_words = nullptr, int64_t accept_docs = 0, uint64_t accept_key = 0) {
    const int cap = rerankK + 64 + ix->dev.R;
    const int kind = accept_words ? 1 : 0;
    if (OPT(ix, OPT_SERVE) == 0 || OPT(ix, OPT_NO_PQF) != 0 || OPT(ix, OPT_NO_PQP) != 0 ||
        OPT(ix, OPT_FORCE_BIG) != 0 || OPT(ix, OPT_FORCE_GENERAL) != 0 || OPT(ix, OPT_PQF_ONLY) != 0)
        return 1;
    if (topK < 1 || topK > JV_SERVE_TOPK_MAX || ix->dev.pq_M <= 0 || !ix->dev.pq_fused || ix->build_client) return 1;
    if (kind == 0) {
        if (OPT(ix, OPT_NO_PQW) != 0 || !jvk_pqw_ok(&ix->dev, cap)) return 1;
    } else {
        // the shapes the filtered pool kernel takes (enqueue_batch): single-pass fused blocks, flat graph, ordinals below 2^29
        const JvIndexDev& dv = ix->dev;
        const bool wf = OPT(ix, OPT_NO_PQW) == 0 && jvk_pqwf_ok(&ix->dev, 4097);  // (the several-waves filtered server also takes PQ-64)
        if (OPT(ix, OPT_FILTER_CACHE) <= 0 || accept_docs <= 0 || dv.num_upper != 0 || dv.R > JV_WAVE || (!wf && dv.R * dv.pq_lanes > JV_WAVE) ||
            dv.n >= (1 << 29) || cap > jvk_pqsf_max_entries())
            return 1;
        if (ix->server_f && cap > ix->server_f->cap_max) return 1;  // (a beam wider than the server's pool: launch path)
    }
    int rc = JV_OK;
    Server* sv = server_get(ix, kind, kind == 0 ? cap : 0, &rc);  // (the filtered server has ONE pool size: what the LDS allows)
    if (!sv) return rc != JV_OK ? rc : 1;
    struct Leave {
        Server* sv;
        jv_index* ix;
        int fslot;
        ~Leave() {
            if (fslot >= 0) filter_release(ix, fslot);
            if (--sv->inflight == 0) ix->server_cv.notify_all();
        }
    } leave{sv, ix, -1};  // (server_get took the reference)
    if (kind == 1 && cap > sv->cap_max) return 1;
    const uint64_t* d_filter = nullptr;
    if (kind == 1) {
        // the filter's bits in HBM: from the cache (verified byte for byte), or uploaded now and waited for — the grid reads
        // them as soon as the slot is published
        if (hipSetDevice(ix->device) != hipSuccess) return fail(JV_EDEVICE, "hipSetDevice failed");
        const size_t nwords = ((size_t)accept_docs + 63) / 64;
        if ((rc = filter_acquire(ix, accept_words, nwords, accept_key, sv->up_stream, &d_filter, &leave.fslot)) != JV_OK) return rc;
        if (leave.fslot < 0) return 1;  // cache full of filters in use: launch path
        {
            std::lock_guard<std::mutex> lk(ix->filter_mu);
            const int nrk = ix->filters[(size_t)leave.fslot].no_serve_rk;
            if (nrk > 0 && rerankK >= nrk) return 1;  // (known not to fit the server's pool: straight to the launch path's rungs)
        }
        if (hipStreamSynchronize(sv->up_stream) != hipSuccess) return fail(JV_EDEVICE, "filter upload failed");
    }
    // the ticket protocol's caller side: csrc/jv_serve_host.h (shared with the sanitizer test, tests/native/serve_sim.cpp)
    int si = 0;
    sv->last_call_ms.store(now_ms(), std::memory_order_relaxed);
    const uint32_t seq = jvsh_take_slot(sv, &si);
    unsigned char* sp = sv->ring + (size_t)si * (size_t)sv->slot_bytes;
    JvServeSlot* slot = jvsh_slot(sv, si);
    slot->topK = topK;
    slot->rk = rerankK;
    slot->visit_limit = visit_limit > 0 ? (int32_t)std::min<int64_t>(visit_limit, INT32_MAX) : 0;
    slot->rerank_floor = rerankFloor;
    slot->accept = (uint64_t)(uintptr_t)d_filter;
    slot->accept_docs = accept_docs;
    slot->done = 0;
    slot->count = 0;
    slot->flags = 0;
    // (a grid only answers a slot whose content belongs to the ticket it claimed; an atomic store: a grid that claimed an
    //  ABANDONED ticket of this slot's previous occupant may be reading the word right now)
    __atomic_store_n(&slot->ticket, (int32_t)seq, __ATOMIC_RELAXED);
    memcpy(sp + JV_SERVE_QUERY_OFF, query, (size_t)ix->dev.d * sizeof(float));
    jvsh_publish(sv, seq);
    auto ensure_alive = [&](auto&& give_up) -> int {
        if (__atomic_load_n(&sv->h_words[JV_SH_ALIVE], __ATOMIC_ACQUIRE) != 0) return JV_OK;
        std::lock_guard<std::mutex> lk(sv->mu);
        if (__atomic_load_n(&sv->h_words[JV_SH_ALIVE], __ATOMIC_ACQUIRE) != 0) return JV_OK;
        hipStreamSynchronize(sv->stream);  // (the previous grid has signalled its exit: let its launch retire)
        const int r = server_launch_locked(sv);
        if (r != JV_OK) give_up();  // (under sv->mu: no other caller can start a grid before the slot is marked abandoned)
        return r;
    };
    if ((rc = jvsh_wait_done(sv, slot, seq, si, ensure_alive)) != JV_OK) return rc;
    const uint32_t f = (uint32_t)slot->flags;
    int ret = JV_OK;
    if (f & (JV_FLAG_OVERFLOW | JV_FLAG_FAILED)) {
        ret = 1;  // the ladder's business (boundary ties beyond the first launch's slack, rerankFloor corner, ...): launch path
        if (kind == 1 && ((f >> 8) & 0xFFu) == 9u) {  // the filter's estimated pool does not fit: remember it with the cached filter
            std::lock_guard<std::mutex> lk(ix->filter_mu);
            int& nrk = ix->filters[(size_t)leave.fslot].no_serve_rk;
            if (nrk == 0 || rerankK < nrk) nrk = rerankK;
        }
    } else {
        if (out_nodes) memcpy(out_nodes, slot->nodes, sizeof(int32_t) * (size_t)topK);
        if (out_docs) memcpy(out_docs, slot->docs, sizeof(int32_t) * (size_t)topK);
        if (out_scores) memcpy(out_scores, slot->scores, sizeof(float) * (size_t)topK);
        if (out_count) *out_count = slot->count;
        if (out_stats) memcpy(out_stats, slot->stats, sizeof(int32_t) * 4);
        if (out_flags) *out_flags = (int32_t)(f & (JV_FLAG_BIG | JV_FLAG_EARLY));
        ix->launches[SERVED_QUERIES]++;
    }
    jvsh_release(sv, seq, si);
    return ret;
}

int check_common(jv_index* index, const void* q, int nq, int topK, int rk, float thr) {
    if (!index) return fail(JV_EINVAL, "index is NULL");
    if (!q && nq > 0) return fail(JV_EINVAL, "query pointer is NULL");
    if (nq < 0 || topK < 0) return fail(JV_EINVAL, "negative nq/topK");
    // jvector: "rerankK %d must be >= topK %d" -> IllegalArgumentException
    if (rk < topK) return fail(JV_EINVAL, "rerankK %d must be >= topK %d", rk, topK);
    (void)thr;
    return JV_OK;
}

}  // namespace

extern "C" {

int jv_abi_version(void) { return JVGPU_ABI_VERSION; }

const char* jv_last_error(void) { return g_last_error.c_str(); }

int jv_set_option(const char* name, int64_t value) {
    if (!name) return fail(JV_EINVAL, "option name is NULL");
    const int id = g_default_opts.find(name);
    if (id < 0) return fail(JV_EINVAL, "unknown option '%s'", name);
    g_default_opts.v[id].store(value);
    return JV_OK;
}

int jv_index_set_option(jv_index* index, const char* name, int64_t value) {
    if (!index || !name) return fail(JV_EINVAL, "index/name is NULL");
    const int id = index->opts.find(name);
    if (id < 0) return fail(JV_EINVAL, "unknown option '%s'", name);
    index->opts.v[id].store(value);
    return JV_OK;
}

void jv_index_destroy(jv_index* ix) {
    if (!ix) return;
    hipSetDevice(ix->device);
    server_destroy(ix);
    {
        ServerPause pause(ix->device);  // (other handles' query servers on this device leave while it is synchronised)
        hipDeviceSynchronize();
    }
    for (Ctx* c : ix->all_ctx) ctx_destroy(c);
    for (FilterEntry& f : ix->filters) {
        jv_free(f.d_words);
        if (f.ready) hipEventDestroy(f.ready);
    }
    for (void* p : ix->owned) jv_free(p);
    {
        XbState& x = ix->xb;
        jv_free(x.vb); jv_free(x.vnorm2); jv_free(x.d_list); jv_free(x.d_counts); jv_free(x.d_queries); jv_free(x.d_qb);
        jv_free(x.d_qn2); jv_free(x.d_thr); jv_free(x.d_surv_cnt); jv_free(x.d_sample); jv_free(x.d_surv); jv_free(x.d_out);
        jv_free(x.d_info); jv_free(x.d_accept);
        jv_host_free(x.h_out); jv_host_free(x.h_info);
        if (x.stream) hipStreamDestroy(x.stream);
    }
    delete ix;
}

int jv_index_create(const jv_index_desc* desc, jv_index** out) {
    if (!desc || !out) return fail(JV_EINVAL, "desc/out is NULL");
    *out = nullptr;
    if (desc->struct_size != sizeof(jv_index_desc))
        return fail(JV_EINVAL, "jv_index_desc.struct_size %u != %zu (ABI mismatch)", desc->struct_size, sizeof(jv_index_desc));
    const int n = desc->n, d = desc->d, R = desc->R;
    if (n < 0 || d <= 0 || R <= 0) return fail(JV_EINVAL, "bad shape n=%d d=%d R=%d", n, d, R);
    // VectorSimilarityMapper.ordToDistFunc throws IllegalArgumentException on unknown ordinals (J/JVectorReader.java:407-413)
    if (desc->similarity < 0 || desc->similarity > 2) return fail(JV_EINVAL, "invalid similarity ordinal %d", desc->similarity);
    const int NM = desc->nvq_M;
    if (NM < 0 || NM > JV_NVQ_MAX_M || NM > d) return fail(JV_EUNSUPPORTED, "nvq_M %d not in [0,%d]", NM, JV_NVQ_MAX_M);
    if (NM > 0 && (!desc->nvq_global_mean || (n > 0 && (!desc->nvq_params || !desc->nvq_bytes)))) return fail(JV_EINVAL, "nvq arrays are NULL");
    if (n > 0 && ((!desc->vectors && NM == 0) || !desc->adj)) return fail(JV_EINVAL, "vectors/adj is NULL");
    if (desc->entry_node >= n) return fail(JV_EINVAL, "entry_node %d out of range", desc->entry_node);
    if (desc->num_upper_layers < 0 || desc->num_upper_layers > JV_MAX_UPPER_LAYERS)
        return fail(JV_EUNSUPPORTED, "num_upper_layers %d not in [0,%d]", desc->num_upper_layers, JV_MAX_UPPER_LAYERS);
    if (desc->num_upper_layers > 0 && !desc->upper_layers) return fail(JV_EINVAL, "upper_layers is NULL");
    if (desc->score_scale != 1.0f && desc->score_scale != 2.0f) return fail(JV_EINVAL, "score_scale must be 1 or 2");
    const int M = desc->pq_M;
    if (M < 0 || M > d) return fail(JV_EINVAL, "pq_M %d out of range", M);
    if (M > 0) {
        if (desc->pq_K <= 0 || desc->pq_K > 256) return fail(JV_EINVAL, "pq_K %d not in [1,256]", desc->pq_K);
        if (!desc->pq_codebooks || (n > 0 && !desc->pq_codes)) return fail(JV_EINVAL, "pq arrays are NULL");
        if ((M + 15) / 16 > 64) return fail(JV_EUNSUPPORTED, "pq_M %d > 1024", M);
    }
    const bool devptr = (desc->flags & JV_DESC_DEVICE_POINTERS) != 0;
    const bool borrow = devptr && (desc->flags & JV_DESC_BORROW) != 0;

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(JV_EDEVICE, "no HIP device available (%s): the GPU engine has no CPU fallback", hipGetErrorString(e));
    if (desc->device < 0 || desc->device >= ndev) return fail(JV_EINVAL, "device %d out of range [0,%d)", desc->device, ndev);
    HIPCHK(hipSetDevice(desc->device));

    jv_index* ix = new jv_index();
    ix->opts.copy_from(g_default_opts);
    ix->device = desc->device;
    ix->build_client = (desc->flags & JV_DESC_BUILD_CLIENT) != 0;
    JvIndexDev& D = ix->dev;
    D.n = n;
    D.d = d;
    D.R = R;
    D.stride = (d + 3) & ~3;
    D.nch = (D.stride + 63) / 64;
    D.sim = desc->similarity;
    D.score_scale = desc->score_scale;
    D.entry = n > 0 ? desc->entry_node : -1;
    D.num_upper = desc->num_upper_layers;
    const hipMemcpyKind kind = devptr ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    int rc = JV_OK;
#define TRY(x)                       \
    do {                             \
        rc = (x);                    \
        if (rc != JV_OK) goto error; \
    } while (0)
#define TRYHIP(x)                                                                          \
    do {                                                                                   \
        hipError_t e2 = (x);                                                               \
        if (e2 != hipSuccess) {                                                            \
            rc = fail(e2 == hipErrorOutOfMemory ? JV_ENOMEM : JV_EDEVICE, "%s failed: %s", #x, \
                      hipGetErrorString(e2));                                              \
            goto error;                                                                    \
        }                                                                                  \
    } while (0)
    {
        if (n > 0) {
            if (!desc->vectors) {
                D.vectors = nullptr;  // NVQ-only field: every exact score is taken against the dequantised record
            } else if (borrow && D.stride == d) {
                D.vectors = desc->vectors;
            } else {
                float* v = nullptr;
                TRY(dev_alloc(ix, &v, (size_t)n * D.stride));
                if (D.stride == d) {
                    TRYHIP(hipMemcpy(v, desc->vectors, (size_t)n * d * sizeof(float), kind));
                } else {
                    TRYHIP(hipMemset(v, 0, (size_t)n * D.stride * sizeof(float)));
                    TRYHIP(hipMemcpy2D(v, (size_t)D.stride * 4, desc->vectors, (size_t)d * 4, (size_t)d * 4, (size_t)n, kind));
                }
                D.vectors = v;
            }
            if (borrow) {
                D.adj = desc->adj;
            } else {
                int32_t* a = nullptr;
                TRY(dev_alloc(ix, &a, (size_t)n * R));
                TRYHIP(hipMemcpy(a, desc->adj, (size_t)n * R * sizeof(int32_t), kind));
                D.adj = a;
            }
            if (desc->ord2doc) {
                if (borrow) {
                    D.ord2doc = desc->ord2doc;
                } else {
                    int32_t* o = nullptr;
                    TRY(dev_alloc(ix, &o, (size_t)n));
                    TRYHIP(hipMemcpy(o, desc->ord2doc, (size_t)n * sizeof(int32_t), kind));
                    D.ord2doc = o;
                }
            }
        }
        // upper layers are always host arrays (tiny)
        for (int l = 0; l < D.num_upper; l++) {
            const jv_layer_desc& L = desc->upper_layers[l];
            if (L.count < 0 || L.degree <= 0 || (L.count > 0 && (!L.nodes || !L.adj))) {
                rc = fail(JV_EINVAL, "bad upper layer %d", l + 1);
                goto error;
            }
            int32_t *nd = nullptr, *ad = nullptr;
            TRY(dev_alloc(ix, &nd, (size_t)L.count));
            TRY(dev_alloc(ix, &ad, (size_t)L.count * L.degree));
            if (L.count > 0) {
                TRYHIP(hipMemcpy(nd, L.nodes, (size_t)L.count * sizeof(int32_t), hipMemcpyHostToDevice));
                TRYHIP(hipMemcpy(ad, L.adj, (size_t)L.count * L.degree * sizeof(int32_t), hipMemcpyHostToDevice));
            }
            D.upper[l].count = L.count;
            D.upper[l].degree = L.degree;
            D.upper[l].nodes = nd;
            D.upper[l].adj = ad;
        }
        if (M > 0) {
            D.pq_M = M;
            D.pq_K = desc->pq_K;
            D.pq_lanes = next_pow2((M + 15) / 16);
            D.pq_code_stride = (M + 15) & ~15;
            // subspace layout: jvector getSubvectorSizesAndOffsets (size d/M, first d%M get +1) unless given
            std::vector<int32_t> off(M + 1, 0);
            for (int m = 0; m < M; m++) {
                int s = desc->pq_sub_sizes ? desc->pq_sub_sizes[m] : d / M + (m < d % M ? 1 : 0);
                if (s <= 0) {
                    rc = fail(JV_EINVAL, "pq_sub_sizes[%d] = %d", m, s);
                    goto error;
                }
                off[m + 1] = off[m] + s;
            }
            if (off[M] != d) {
                rc = fail(JV_EINVAL, "pq subspace sizes sum to %d, expected d=%d", off[M], d);
                goto error;
            }
            int32_t* doff = nullptr;
            TRY(dev_alloc(ix, &doff, (size_t)M + 1));
            TRYHIP(hipMemcpy(doff, off.data(), (size_t)(M + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
            D.pq_sub_off = doff;
            ix->pq_sub_off = off;
            // codebooks (host) -> transposed [dim][256]
            std::vector<float> cbT((size_t)d * 256, 0.0f);
            std::vector<float> norm;
            if (D.sim == JV_SIM_COSINE) norm.assign((size_t)M * 256, 0.0f);
            const float* cb = desc->pq_codebooks;
            for (int m = 0; m < M; m++) {
                int s = off[m + 1] - off[m];
                for (int c = 0; c < desc->pq_K; c++) {
                    const float* cv = cb + (size_t)c * s;
                    float acc = 0.0f;
                    for (int i = 0; i < s; i++) {
                        cbT[(size_t)(off[m] + i) * 256 + c] = cv[i];
                        acc = std::fmaf(cv[i], cv[i], acc);  // same chain as the oracle's norm table
                    }
                    if (!norm.empty()) norm[(size_t)m * 256 + c] = acc;
                }
                cb += (size_t)desc->pq_K * s;
            }
            float* dcb = nullptr;
            TRY(dev_alloc(ix, &dcb, cbT.size()));
            TRYHIP(hipMemcpy(dcb, cbT.data(), cbT.size() * sizeof(float), hipMemcpyHostToDevice));
            D.pq_cbT = dcb;
            if (!norm.empty()) {
                float* dn = nullptr;
                TRY(dev_alloc(ix, &dn, norm.size()));
                TRYHIP(hipMemcpy(dn, norm.data(), norm.size() * sizeof(float), hipMemcpyHostToDevice));
                D.pq_norm_lut = dn;
            }
            if (desc->pq_centroid) {
                float* dc = nullptr;
                TRY(dev_alloc(ix, &dc, (size_t)d));
                TRYHIP(hipMemcpy(dc, desc->pq_centroid, (size_t)d * sizeof(float), hipMemcpyHostToDevice));
                D.pq_centroid = dc;
            }
            if (n > 0) {
                if (borrow && D.pq_code_stride == M) {
                    D.pq_codes = desc->pq_codes;
                } else {
                    uint8_t* codes = nullptr;
                    TRY(dev_alloc(ix, &codes, (size_t)n * D.pq_code_stride));
                    if (D.pq_code_stride == M) {
                        TRYHIP(hipMemcpy(codes, desc->pq_codes, (size_t)n * M, kind));
                    } else {
                        TRYHIP(hipMemset(codes, 0, (size_t)n * D.pq_code_stride));
                        TRYHIP(hipMemcpy2D(codes, (size_t)D.pq_code_stride, desc->pq_codes, (size_t)M, (size_t)M, (size_t)n, kind));
                    }
                    D.pq_codes = codes;
                }
            }
        }
        if (NM > 0) {
            D.nvq_M = NM;
            D.nvq_stride = (d + 3) & ~3;
            std::vector<int32_t> noff((size_t)NM + 1, 0);
            for (int m2 = 0; m2 < NM; m2++) {
                const int sz = desc->nvq_sub_sizes ? desc->nvq_sub_sizes[m2] : d / NM + (m2 < d % NM ? 1 : 0);
                if (sz <= 0) {
                    rc = fail(JV_EINVAL, "nvq_sub_sizes[%d] = %d", m2, sz);
                    goto error;
                }
                noff[(size_t)m2 + 1] = noff[(size_t)m2] + sz;
            }
            if (noff[(size_t)NM] != d) {
                rc = fail(JV_EINVAL, "nvq subvector sizes sum to %d, expected d=%d", noff[(size_t)NM], d);
                goto error;
            }
            int32_t* dno = nullptr;
            TRY(dev_alloc(ix, &dno, (size_t)NM + 1));
            TRYHIP(hipMemcpy(dno, noff.data(), ((size_t)NM + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
            D.nvq_sub_off = dno;
            std::vector<float> mean((size_t)D.nch * 64, 0.0f);
            memcpy(mean.data(), desc->nvq_global_mean, (size_t)d * sizeof(float));
            float* dmean = nullptr;
            TRY(dev_alloc(ix, &dmean, mean.size()));
            TRYHIP(hipMemcpy(dmean, mean.data(), mean.size() * sizeof(float), hipMemcpyHostToDevice));
            D.nvq_mean = dmean;
            if (n > 0) {
                float* dpr = nullptr;
                TRY(dev_alloc(ix, &dpr, (size_t)n * NM * 4));
                TRYHIP(hipMemcpy(dpr, desc->nvq_params, (size_t)n * NM * 4 * sizeof(float), kind));
                D.nvq_params = dpr;
                uint8_t* db = nullptr;
                TRY(dev_alloc(ix, &db, (size_t)n * D.nvq_stride));
                if (D.nvq_stride == d) {
                    TRYHIP(hipMemcpy(db, desc->nvq_bytes, (size_t)n * d, kind));
                } else {
                    TRYHIP(hipMemset(db, 0, (size_t)n * D.nvq_stride));
                    TRYHIP(hipMemcpy2D(db, (size_t)D.nvq_stride, desc->nvq_bytes, (size_t)d, (size_t)d, (size_t)n, kind));
                }
                D.nvq_bytes = db;
            }
        }
        if (M > 0 && n > 0 && (desc->flags & JV_DESC_FUSED_ADC)) {
            // fused layout: each node's neighbours' codes next to each other, in adjacency order
            uint8_t* fused = nullptr;
            TRY(dev_alloc(ix, &fused, (size_t)n * R * D.pq_code_stride));
            TRYHIP(jvk_build_fused(D.pq_codes, D.adj, fused, (long long)n, R, D.pq_code_stride, nullptr));
            if (D.sim == JV_SIM_COSINE && D.pq_norm_lut && D.pq_lanes <= 16) {
                // cosine on the several-waves kernels: the code vectors' squared norms, per node and per adjacency slot (jv_device.h)
                float *nn = nullptr, *fn = nullptr;
                TRY(dev_alloc(ix, &nn, (size_t)n));
                TRY(dev_alloc(ix, &fn, (size_t)n * R));
                TRYHIP(jvk_build_code_norms(D.pq_codes, D.pq_norm_lut, D.adj, nn, fn, (long long)n, R, M, D.pq_code_stride, D.pq_lanes, nullptr));
                D.pq_node_norm = nn;
                D.pq_fused_norm = fn;
            }
            TRYHIP(hipDeviceSynchronize());
            D.pq_fused = fused;
            ix->info.fused_adc = 1;
        }
        TRYHIP(jvk_set_max_lds(kMaxLds));
        TRYHIP(jvk_pqp_set_max_lds(kMaxLds));
        TRYHIP(jvk_pqw_set_max_lds(kMaxLds));
        TRYHIP(jvk_pqwf_set_max_lds(kMaxLds));
        TRYHIP(jvk_pqs_set_max_lds(kMaxLds));
        TRYHIP(jvk_pqsf_set_max_lds(kMaxLds));
        TRYHIP(jvk_pqswf_set_max_lds(kMaxLds));
        {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, desc->device) == hipSuccess && prop.multiProcessorCount > 0) ix->cu_count = prop.multiProcessorCount;
        }
        TRY(ctx_create(ix, &ix->async_ctx));
        ix->async_ctxs.push_back(ix->async_ctx);
    }
    ix->info.n = n;
    ix->info.d = d;
    ix->info.R = R;
    ix->info.similarity = D.sim;
    ix->info.pq_M = D.pq_M;
    ix->info.pq_K = D.pq_K;
    ix->info.num_upper_layers = D.num_upper;
    ix->info.device = ix->device;
    ix->info.row_stride_floats = D.stride;
    *out = ix;
    return JV_OK;
error:
    jv_index_destroy(ix);
    return rc;
#undef TRY
#undef TRYHIP
}

int jv_index_get_info(const jv_index* index, jv_index_info* out) {
    if (!index || !out) return fail(JV_EINVAL, "index/out is NULL");
    *out = index->info;
    jv_index* ix = const_cast<jv_index*>(index);
    int64_t scratch = g_scratch[index->device & 63].bytes.load();
    {
        std::lock_guard<std::mutex> lk(ix->mu);
        for (const Ctx* c : ix->all_ctx)
            scratch += (int64_t)(c->queries_cap * 4 + c->nq_cap * 4 + c->accept_cap * 8 + c->arena_cap + c->pqp_log_ints * 4 +
                                 (size_t)c->spill_tables * (size_t)c->spill_slots * 4 + 32 + c->vis_arena_units * 16 + c->vis_nq_cap * 8);
    }
    {
        std::lock_guard<std::mutex> lk(ix->filter_mu);
        for (const FilterEntry& f : ix->filters) scratch += (int64_t)f.cap_words * 8;
        out->filter_cache_hits = ix->filter_hits;
        out->filter_cache_misses = ix->filter_misses;
    }
    scratch += ix->xb.bytes;  // (the batched exact scorer's bf16 mirror of the vectors and its per-call buffers)
    out->scratch_bytes = scratch;
    return JV_OK;
}

int jv_index_get_counter(const jv_index* index, const char* name, int64_t* out) {
    if (!index || !name || !out) return fail(JV_EINVAL, "index/name/out is NULL");
    static const char* const kNames[] = {"launches_pqw", "launches_pqp", "launches_pqf", "launches_lds", "launches_big", "launches_serve", "served_queries"};
    for (int i = 0; i < 7; i++)
        if (strcmp(name, kNames[i]) == 0) {
            *out = index->launches[i].load();
            return JV_OK;
        }
    if (strcmp(name, "search_kernel_ns") == 0 || strcmp(name, "search_kernel_timed") == 0) {
        // (measurement: waits for the timed launches still in flight; not meant to be read while other threads search)
        jv_index* ix = const_cast<jv_index*>(index);
        {
            std::lock_guard<std::mutex> lk(ix->mu);
            for (Ctx* c : ix->all_ctx) kt_harvest(ix, c);
        }
        *out = name[14] == 'n' ? ix->search_kernel_ns.load() : ix->search_kernel_timed.load();
        return JV_OK;
    }
    if (strcmp(name, "retry_rungs_skipped") == 0) { *out = index->retry_rungs_skipped.load(); return JV_OK; }
    if (strcmp(name, "serve_alive") == 0) {
        // how many of this index's resident grids are running right now (the grid itself clears the word when it leaves idle:
        // jv_serve_claim.h) — lets a test DRIVE grid restarts instead of hoping for them
        int64_t alive = 0;
        jv_index* ix = const_cast<jv_index*>(index);
        std::lock_guard<std::mutex> lk(ix->server_mu);
        for (const JvQueryServer* sv : {ix->server, ix->server_f})
            if (sv && sv->h_words && __atomic_load_n(&sv->h_words[JV_SH_ALIVE], __ATOMIC_ACQUIRE) != 0) alive++;
        *out = alive;
        return JV_OK;
    }
    if (strcmp(name, "exact_calls") == 0) { *out = index->xb.exact_calls.load(); return JV_OK; }
    if (strcmp(name, "exact_batches") == 0) { *out = index->xb.exact_batches.load(); return JV_OK; }
    return fail(JV_EINVAL, "unknown counter '%s'", name);
}

static int search_batch_device_impl(jv_index* index, const float* d_queries, int32_t nq, int32_t topK, int32_t rerankK,
                                    float threshold, float rerankFloor, const uint64_t* d_accept_doc_words,
                                    int64_t accept_num_docs, int64_t visit_limit, int32_t* d_out_nodes, int32_t* d_out_docs,
                                    float* d_out_scores, int32_t* d_out_count, int32_t* d_out_stats,
                                    int32_t* d_out_flags, void* hip_stream);

int jv_search_batch_device(jv_index* index, const float* d_queries, int32_t nq, int32_t topK, int32_t rerankK,
                           float threshold, float rerankFloor, const uint64_t* d_accept_doc_words,
                           int64_t accept_num_docs, int32_t* d_out_nodes, int32_t* d_out_docs,
                           float* d_out_scores, int32_t* d_out_count, int32_t* d_out_stats,
                           int32_t* d_out_flags, void* hip_stream) {
    return search_batch_device_impl(index, d_queries, nq, topK, rerankK, threshold, rerankFloor, d_accept_doc_words, accept_num_docs, 0,
                                    d_out_nodes, d_out_docs, d_out_scores, d_out_count, d_out_stats, d_out_flags, hip_stream);
}

static int search_batch_device_impl(jv_index* index, const float* d_queries, int32_t nq, int32_t topK, int32_t rerankK,
                                    float threshold, float rerankFloor, const uint64_t* d_accept_doc_words,
                                    int64_t accept_num_docs, int64_t visit_limit, int32_t* d_out_nodes, int32_t* d_out_docs,
                                    float* d_out_scores, int32_t* d_out_count, int32_t* d_out_stats,
                                    int32_t* d_out_flags, void* hip_stream) {
    int rc = check_common(index, d_queries, nq, topK, rerankK, threshold);
    if (rc != JV_OK) return rc;
    if (nq == 0) return JV_OK;
    if (!d_out_nodes || !d_out_scores || !d_out_count || !d_out_stats) return fail(JV_EINVAL, "output pointer is NULL");
    HIPCHK(hipSetDevice(index->device));
    std::lock_guard<std::mutex> lk(index->async_mu);
    // Which context: the one this stream used last (its work is already ordered behind that use); else one whose last use has
    // completed; else a new one below the cap (async_contexts: batches on different streams then run side by side — a server that
    // receives the next batch of 256 queries while the previous one is being answered); else the least recently used one, ordered
    // behind its last use.
    Ctx* c = index->async_ctx;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (hip_stream) {
        c = nullptr;
        for (Ctx* x : index->async_ctxs)
            if (x->last_stream == s) { c = x; break; }
        if (!c)
            for (Ctx* x : index->async_ctxs)
                if (!x->last_stream || hipEventQuery(x->last_use) == hipSuccess) { c = x; break; }
        (void)hipGetLastError();  // (hipErrorNotReady of the queries above is not an error)
        if (!c && (int64_t)index->async_ctxs.size() < std::max<int64_t>(1, OPT(index, OPT_ASYNC_CONTEXTS))) {
            std::lock_guard<std::mutex> lk2(index->mu);
            rc = ctx_create(index, &c);
            if (rc != JV_OK) return rc;
            index->async_ctxs.push_back(c);
        }
        if (!c) {
            c = index->async_ctxs[0];
            for (Ctx* x : index->async_ctxs)
                if (x->last_clock < c->last_clock) c = x;
        }
    }
    c->last_clock = ++index->async_clock;
    // (always: a destroyed stream's handle value can come back as a NEW stream, which is not ordered behind the old one's use of
    //  this context — ADVICE r4; waiting for an event that has completed costs nothing)
    if (c->last_stream) HIPCHK(hipStreamWaitEvent(s, c->last_use, 0));
    if (!hip_stream) {
        // No caller stream: the library's own (non-blocking) stream.  A caller that produced the queries on the legacy default
        // stream (handle 0 — what a CUDA-style runtime hands out as "the current stream") expects to be ordered behind that work:
        // wait for whatever the default stream has in flight right now.  (Round 4's graph builder searched half-written query
        // rows without this.)
        if (!c->ev_null) HIPCHK(hipEventCreateWithFlags(&c->ev_null, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->ev_null, nullptr));
        HIPCHK(hipStreamWaitEvent(s, c->ev_null, 0));
    }
    // whatever happens below (also a failure after a partial enqueue), the next call on another stream must be ordered
    // behind the kernels that may already use this context's counters and scratch
    struct UseGuard {
        Ctx* c;
        hipStream_t s;
        ~UseGuard() {
            hipEventRecord(c->last_use, s);
            c->last_stream = s;
        }
    } use_guard{c, s};
    if (!d_out_flags) {
        rc = grow((void**)&c->d_flags, &c->nq_cap, (size_t)nq, sizeof(int32_t));
        if (rc != JV_OK) return rc;
        d_out_flags = c->d_flags;
    }
    if (topK == 0 || index->dev.n == 0 || index->dev.entry < 0) {
        HIPCHK(hipMemsetAsync(d_out_count, 0, (size_t)nq * 4, s));
        HIPCHK(hipMemsetAsync(d_out_stats, 0, (size_t)nq * 16, s));
        HIPCHK(hipMemsetAsync(d_out_flags, 0, (size_t)nq * 4, s));
        if (topK > 0) {
            HIPCHK(hipMemsetAsync(d_out_nodes, 0xFF, (size_t)nq * topK * 4, s));
            if (d_out_docs) HIPCHK(hipMemsetAsync(d_out_docs, 0xFF, (size_t)nq * topK * 4, s));
            HIPCHK(hipMemsetAsync(d_out_scores, 0, (size_t)nq * topK * 4, s));
        }
    } else {
        rc = enqueue_batch(index, c, s, d_queries, nq, topK, rerankK, threshold, rerankFloor, d_accept_doc_words,
                           accept_num_docs, d_out_nodes, d_out_docs, d_out_scores, d_out_count, d_out_stats, d_out_flags, 0, visit_limit);
        if (rc != JV_OK) return rc;
    }
    if (!hip_stream) HIPCHK(hipStreamSynchronize(s));
    return JV_OK;
}

}  // extern "C"

namespace {

// Doc-filter cache: a host bitset that was uploaded before (same 64-bit content hash — or caller-supplied key — and
// length) is served from HBM instead of crossing PCIe again (1.25 MB per call at 10M docs).  Returns slot >= 0 and the
// device pointer, or slot = -1 when the cache is off / full of filters in use (the caller then stages the words itself).
int filter_acquire(jv_index* ix, const uint64_t* words, size_t nwords, uint64_t key, hipStream_t stream, const uint64_t** d_out,
                   int* slot) {
    *slot = -1;
    const int cap = (int)OPT(ix, OPT_FILTER_CACHE);
    if (cap <= 0 || nwords == 0) return JV_OK;
    if (!key) key = hash_words(words, nwords);
    std::lock_guard<std::mutex> lk(ix->filter_mu);
    int victim = -1;
    for (size_t i = 0; i < ix->filters.size(); i++) {
        FilterEntry& f = ix->filters[i];
        // The bitset carries doc-level security and deletes: never trust the key alone.  A colliding hash or a re-used caller
        // key with different bits is a miss (and, entry idle, the entry is refilled below).
        if (f.key == key && f.words == nwords && f.d_words && f.host.size() == nwords &&
            memcmp(f.host.data(), words, nwords * 8) == 0) {
            f.users++;
            f.stamp = ++ix->filter_clock;
            ix->filter_hits++;
            HIPCHK(hipStreamWaitEvent(stream, f.ready, 0));
            *d_out = f.d_words;
            *slot = (int)i;
            return JV_OK;
        }
        if (f.users == 0 && (victim < 0 || f.stamp < ix->filters[(size_t)victim].stamp)) victim = (int)i;
    }
    if ((int)ix->filters.size() < cap) {
        ix->filters.emplace_back();
        victim = (int)ix->filters.size() - 1;
    }
    if (victim < 0) return JV_OK;
    FilterEntry& f = ix->filters[(size_t)victim];
    if (!f.ready) HIPCHK(hipEventCreateWithFlags(&f.ready, hipEventDisableTiming));
    if (f.cap_words < nwords) {
        // (the entry is not in use: every launch that read it has been synchronised by its caller)
        if (f.d_words) HIPCHK(jv_free(f.d_words));
        f.d_words = nullptr;
        f.cap_words = 0;
        f.key = 0;
        HIPCHK(hipMalloc((void**)&f.d_words, nwords * 8));
        f.cap_words = nwords;
    }
    f.key = 0;  // not valid until the copy is enqueued
    f.no_serve_rk = 0;
    try {
        f.host.assign(words, words + nwords);  // (also the staging source: the caller's buffer may be gone before the copy runs)
    } catch (const std::bad_alloc&) {
        return JV_OK;  // no cache entry: the caller uploads the bits itself
    }
    HIPCHK(hipMemcpyAsync(f.d_words, f.host.data(), nwords * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipEventRecord(f.ready, stream));
    f.key = key;
    f.words = nwords;
    f.users = 1;
    f.stamp = ++ix->filter_clock;
    ix->filter_misses++;
    *d_out = f.d_words;
    *slot = victim;
    return JV_OK;
}
void filter_release(jv_index* ix, int slot) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(ix->filter_mu);
    ix->filters[(size_t)slot].users--;
}

// Enqueue a marker behind everything on the context's stream and wait for it by polling host memory; falls back to the
// runtime's wait when the marker does not show up (a faulted kernel never writes it)
// diagnostics (JV_BATCH_TRACE=1): trace point i of the running call — a marker kernel writes the call's sequence number into
// word i of the marker block; a call that waits longer than 5 ms prints which points the stream has passed
static bool batch_trace() {
    static const bool on = getenv("JV_BATCH_TRACE") != nullptr;
    return on;
}
void trace_point(Ctx* c, int i, hipStream_t stream) {
    if (!batch_trace() || !c->h_mark || i < 1 || i > 15) return;
    jvk_launch_mark(c->h_mark + i, c->mark_seq + 1, stream);
}
int mark_enqueue(Ctx* c) {
    if (!c->h_mark) {
        HIPCHK(hipHostMalloc((void**)&c->h_mark, 64, hipHostMallocMapped | hipHostMallocCoherent));
        __atomic_store_n(c->h_mark, 0, __ATOMIC_RELEASE);
    }
    c->mark_seq = c->mark_seq == INT32_MAX ? 1 : c->mark_seq + 1;
    HIPCHK(jvk_launch_mark(c->h_mark, c->mark_seq, c->stream));
    return JV_OK;
}
bool mark_reached(const Ctx* c) { return __atomic_load_n(c->h_mark, __ATOMIC_ACQUIRE) == c->mark_seq; }
int mark_wait(Ctx* c) {
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int spins = 0; !mark_reached(c); spins++) {
        if (spins < 200) {
            sched_yield();
            continue;
        }
        struct timespec ts = {0, 20000};  // 20 us
        nanosleep(&ts, nullptr);
        if (batch_trace() && spins == 400) {  // ~ 8 ms in
            fprintf(stderr, "[jvgpu batch] call %d still waiting; trace points passed:", c->mark_seq);
            for (int i = 1; i < 16; i++) fprintf(stderr, " %d%s", i, __atomic_load_n(c->h_mark + i, __ATOMIC_ACQUIRE) == c->mark_seq ? "+" : "-");
            fprintf(stderr, "\n");
        }
        if ((spins & 1023) == 0) {
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (t1.tv_sec - t0.tv_sec > 20) {  // something is wrong: let the runtime report it
                HIPCHK(hipStreamSynchronize(c->stream));
                return JV_OK;
            }
        }
    }
    return JV_OK;
}

struct HostSearchExtras {
    int64_t visit_limit = 0;
    uint64_t accept_key = 0;
    int32_t* out_status = nullptr;  // [nq] per-query JV_OK / JV_ENOMEM
    int32_t* out_flags = nullptr;   // [nq] JV_QFLAG_*
    // direct delivery (combined one-query calls): on_launched() once the batch is on the GPU, then on_ready(i, ...) exactly
    // once per query, as soon as its row is final (the pointers address the row inside the pinned arena)
    void* user = nullptr;
    void (*on_launched)(void* user) = nullptr;
    void (*on_ready)(void* user, int i, const int32_t* nodes, const int32_t* docs, const float* scores, int32_t count,
                     const int32_t* stats, int32_t qflags, int status) = nullptr;
};

// host-pointer batch.  `accept_list` (optional, nq host pointers) gives every query its own doc filter of
// accept_num_docs bits; `accept_doc_words` is one filter shared by the whole batch.
// Returns JV_ENOMEM when at least one query could not be answered (its row is empty and out_status[i] says so); the
// rows of every other query are valid.
int search_batch_host(jv_index* index, const float* queries, int32_t nq, int32_t topK, int32_t rerankK,
                      float threshold, float rerankFloor, const uint64_t* accept_doc_words,
                      const uint64_t* const* accept_list, int64_t accept_num_docs, int32_t* out_nodes, int32_t* out_docs,
                      float* out_scores, int32_t* out_count, int32_t* out_stats, const HostSearchExtras& ex = HostSearchExtras()) {
    int rc = check_common(index, queries, nq, topK, rerankK, threshold);
    if (rc != JV_OK) return rc;
    if (nq == 0) return JV_OK;
    const size_t outn = (size_t)nq * (size_t)topK;
    if (ex.out_status) memset(ex.out_status, 0, sizeof(int32_t) * (size_t)nq);
    if (ex.out_flags) memset(ex.out_flags, 0, sizeof(int32_t) * (size_t)nq);
    if (topK == 0 || index->dev.n == 0 || index->dev.entry < 0) {
        for (size_t i = 0; i < outn; i++) {
            if (out_nodes) out_nodes[i] = -1;
            if (out_docs) out_docs[i] = -1;
            if (out_scores) out_scores[i] = 0.0f;
        }
        if (out_count) memset(out_count, 0, sizeof(int32_t) * (size_t)nq);
        if (out_stats) memset(out_stats, 0, sizeof(int32_t) * 4 * (size_t)nq);
        return JV_OK;
    }
    HIPCHK(hipSetDevice(index->device));
    Ctx* c = nullptr;
    rc = ctx_acquire(index, &c);
    if (rc != JV_OK) return rc;
    int filter_slot = -1;
    struct Releaser {
        jv_index* ix;
        Ctx* c;
        int* slot;
        ~Releaser() {
            // (a failed call may leave work in flight on the context's stream: drain it before the context or the
            // cached filter can be handed to another caller)
            hipStreamSynchronize(c->stream);
            filter_release(ix, *slot);
            ctx_release(ix, c);
        }
    } rel{index, c, &filter_slot};
    const int d = index->dev.d;
    if ((rc = grow((void**)&c->d_queries, &c->queries_cap, (size_t)nq * d, sizeof(float))) != JV_OK) return rc;
    // arena layout (4-byte units)
    const size_t o_nodes = 0, o_docs = outn, o_scores = 2 * outn, o_count = 3 * outn, o_stats = o_count + (size_t)nq,
                 o_flags = o_stats + 4 * (size_t)nq, total4 = o_flags + (size_t)nq;
    if (total4 * 4 > c->arena_cap) {
        jv_free(c->d_arena);
        c->d_arena = nullptr;
        jv_host_free(c->h_arena);
        c->h_arena = nullptr;
        c->arena_cap = 0;
        const size_t cap = total4 * 4 + total4;  // 25 % head room
        HIPCHK(hipMalloc((void**)&c->d_arena, cap));
        HIPCHK(hipHostMalloc((void**)&c->h_arena, cap, hipHostMallocDefault));
        c->arena_cap = cap;
    }
    const bool direct = ex.on_ready != nullptr;
    // small batches write their rows straight into pinned, device-visible memory as well: no result copy at all (next to a
    // starting query-server grid the wait for that copy was seen to last until the grid left — DESIGN.md "Resident grids")
    const bool pinned_out = !direct && nq <= 1024;
    int32_t* done = nullptr;
    if (direct || pinned_out) {
        const size_t need = (total4 + (size_t)nq) * 4;
        if (need > c->direct_cap) {
            jv_host_free(c->h_direct);
            c->h_direct = nullptr;
            c->direct_cap = 0;
            const size_t cap = std::max<size_t>(need + need / 4, 65536);
            HIPCHK(hipHostMalloc((void**)&c->h_direct, cap, hipHostMallocMapped | hipHostMallocCoherent));
            c->direct_cap = cap;
        }
        if (!c->ev_direct) HIPCHK(hipEventCreateWithFlags(&c->ev_direct, hipEventDisableTiming));
        if (direct) {
            done = (int32_t*)c->h_direct + total4;
            memset(done, 0, (size_t)nq * 4);
        }
    }
    int32_t* const a32 = (direct || pinned_out) ? (int32_t*)c->h_direct : (int32_t*)c->d_arena;
    int32_t* const dn = a32 + o_nodes;
    int32_t* const dd = a32 + o_docs;
    float* const dsc = (float*)(a32 + o_scores);
    int32_t* const dc = a32 + o_count;
    int32_t* const dst = a32 + o_stats;
    int32_t* const dfl = a32 + o_flags;
    const uint64_t* d_accept = nullptr;
    int64_t accept_stride = 0;
    if (accept_doc_words || accept_list) {
        if (accept_num_docs < 0) return fail(JV_EINVAL, "accept_num_docs < 0");
        const size_t copy_words = ((size_t)accept_num_docs + 63) / 64;
        size_t words = copy_words ? copy_words : 1;
        // one filter for the whole batch (given as such, or every caller of a combined batch passed the same bits)?
        const uint64_t* shared = accept_doc_words;
        uint64_t key = ex.accept_key;
        if (!shared && accept_list && copy_words && OPT(index, OPT_FILTER_CACHE) > 0) {
            // (same bits, not same hash: every query of a combined batch is answered under ITS caller's filter)
            bool same = true;
            for (int i = 1; i < nq && same; i++)
                same = accept_list[i] == accept_list[0] || memcmp(accept_list[i], accept_list[0], copy_words * 8) == 0;
            if (same) {
                shared = accept_list[0];
                key = hash_words(accept_list[0], copy_words);
            } else {
                key = 0;
            }
        }
        if (shared && copy_words) {
            if ((rc = filter_acquire(index, shared, copy_words, key, c->stream, &d_accept, &filter_slot)) != JV_OK) return rc;
        }
        if (!d_accept) {
            const size_t sets = shared ? 1 : (size_t)nq;
            if ((rc = grow((void**)&c->d_accept, &c->accept_cap, words * sets, 8)) != JV_OK) return rc;
            for (size_t i = 0; i < sets && copy_words; i++)
                HIPCHK(hipMemcpyAsync(c->d_accept + i * words, shared ? shared : accept_list[i], copy_words * 8,
                                      hipMemcpyHostToDevice, c->stream));
            d_accept = c->d_accept;
            accept_stride = shared ? 0 : (int64_t)words;
        }
    }
    const size_t qbytes = (size_t)nq * d * sizeof(float);
    trace_point(c, 1, c->stream);
    if (qbytes <= (1u << 20)) {  // small batches: stage through pinned memory (a pageable H2D is a blocking staged copy)
        if (qbytes > c->h_query_cap) {
            jv_host_free(c->h_query);
            c->h_query = nullptr;
            HIPCHK(hipHostMalloc((void**)&c->h_query, qbytes < 65536 ? 65536 : qbytes, hipHostMallocDefault));
            c->h_query_cap = qbytes < 65536 ? 65536 : qbytes;
        }
        memcpy(c->h_query, queries, qbytes);
        HIPCHK(hipMemcpyAsync(c->d_queries, c->h_query, qbytes, hipMemcpyHostToDevice, c->stream));
    } else {
        HIPCHK(hipMemcpyAsync(c->d_queries, queries, qbytes, hipMemcpyHostToDevice, c->stream));
    }
    trace_point(c, 2, c->stream);
    const int phase1 = OPT(index, OPT_LAZY_BIG) != 0 ? 1 : 0;
    bool big_owed = false;
    rc = enqueue_batch(index, c, c->stream, c->d_queries, nq, topK, rerankK, threshold, rerankFloor, d_accept,
                       accept_num_docs, dn, dd, dsc, dc, dst, dfl, accept_stride, ex.visit_limit, done, phase1, &big_owed);
    if (rc != JV_OK) return rc;
    auto enqueue_big = [&]() {  // the HBM-scratch rung for the rows the on-chip rungs flagged
        big_owed = false;
        return enqueue_batch(index, c, c->stream, c->d_queries, nq, topK, rerankK, threshold, rerankFloor, d_accept, accept_num_docs,
                             dn, dd, dsc, dc, dst, dfl, accept_stride, ex.visit_limit, nullptr, 2, nullptr);
    };
    if (direct) {
        // Rows land in pinned host memory.  A row whose completion word is set is final; every row is once the stream has
        // run dry (the rungs behind the first launch only touch flagged rows).  One poller per batch: this thread.
        if ((rc = mark_enqueue(c)) != JV_OK) return rc;
        if (ex.on_launched) ex.on_launched(ex.user);
        const int32_t* h32d = (const int32_t*)c->h_direct;
        std::vector<char> delivered((size_t)nq, 0);
        int remaining = nq, failed_d = 0, first_d = -1;
        struct timespec td0;
        clock_gettime(CLOCK_MONOTONIC, &td0);
        bool drained = false;  // the runtime confirmed that the stream ran dry (deadline path)
        for (int spins = 0; remaining > 0; spins++) {
            if (!drained && (spins & 4095) == 4095) {
                // same deadline as mark_wait: a marker that never lands (a kernel of the batch faulted) must not leave this
                // leader — and every combined caller blocked on its semaphore — spinning for ever
                struct timespec td1;
                clock_gettime(CLOCK_MONOTONIC, &td1);
                if (td1.tv_sec - td0.tv_sec > 20) {
                    const hipError_t se = hipStreamSynchronize(c->stream);
                    if (se != hipSuccess)  // (the caller hands this code to every owner that was not served: jv_search's combiner)
                        return fail(JV_EDEVICE, "batch did not complete: %s", hipGetErrorString(se));
                    drained = true;
                }
            }
            const hipError_t qe = (drained || mark_reached(c)) ? hipSuccess : hipErrorNotReady;  // (a marker in host memory, not hipEventQuery: see mark_enqueue)
            bool finished = qe == hipSuccess;
            if (finished && big_owed) {
                bool any = false;
                for (int i = 0; i < nq && !any; i++)
                    any = !delivered[(size_t)i] && ((uint32_t)h32d[o_flags + (size_t)i] & JV_FLAG_OVERFLOW) != 0;
                if (any) {
                    if ((rc = enqueue_big()) != JV_OK) return rc;
                    if ((rc = mark_enqueue(c)) != JV_OK) return rc;
                    drained = false;
                    clock_gettime(CLOCK_MONOTONIC, &td0);
                    finished = false;  // (rows without the flag are final and are handed out below)
                }
                big_owed = false;
            }
            for (int i = 0; i < nq; i++) {
                if (delivered[(size_t)i]) continue;
                const bool on_chip_done = qe == hipSuccess && ((uint32_t)h32d[o_flags + (size_t)i] & JV_FLAG_OVERFLOW) == 0;
                if (!finished && !on_chip_done && __atomic_load_n(&done[i], __ATOMIC_ACQUIRE) == 0) continue;
                const uint32_t f = (uint32_t)h32d[o_flags + (size_t)i];
                const bool bad = (f & (JV_FLAG_FAILED | JV_FLAG_OVERFLOW)) != 0;
                if (bad) {
                    failed_d++;
                    if (first_d < 0) first_d = i;
                }
                ex.on_ready(ex.user, i, h32d + o_nodes + (size_t)i * topK, h32d + o_docs + (size_t)i * topK,
                            (const float*)(h32d + o_scores) + (size_t)i * topK, h32d[o_count + (size_t)i], h32d + o_stats + 4 * (size_t)i,
                            (int32_t)(f & (JV_FLAG_BIG | JV_FLAG_EARLY)), bad ? JV_ENOMEM : JV_OK);
                delivered[(size_t)i] = 1;
                remaining--;
            }
            if (remaining > 0 && !finished) {
                if (spins < 64) sched_yield();
                else {
                    struct timespec ts = {0, 20000};  // 20 us
                    nanosleep(&ts, nullptr);
                }
            }
        }
        if (failed_d)
            return fail(JV_ENOMEM, "%d of %d queries (first: %d) overflowed the HBM scratch; raise option big_cand_cap (the other rows are valid)",
                        failed_d, nq, first_d);
        return JV_OK;
    }
    if (!pinned_out) HIPCHK(hipMemcpyAsync(c->h_arena, c->d_arena, total4 * 4, hipMemcpyDeviceToHost, c->stream));
    if ((rc = mark_enqueue(c)) != JV_OK || (rc = mark_wait(c)) != JV_OK) return rc;
    const int32_t* const res32 = pinned_out ? (const int32_t*)c->h_direct : (const int32_t*)c->h_arena;
    if (big_owed) {
        const int32_t* fl0 = res32 + o_flags;
        bool any = false;
        for (int i = 0; i < nq && !any; i++) any = ((uint32_t)fl0[i] & JV_FLAG_OVERFLOW) != 0;
        if (any) {
            if ((rc = enqueue_big()) != JV_OK) return rc;
            if (!pinned_out) HIPCHK(hipMemcpyAsync(c->h_arena, c->d_arena, total4 * 4, hipMemcpyDeviceToHost, c->stream));
            if ((rc = mark_enqueue(c)) != JV_OK || (rc = mark_wait(c)) != JV_OK) return rc;
        }
    }
    const int32_t* h32 = res32;
    if (out_nodes) memcpy(out_nodes, h32 + o_nodes, outn * 4);
    if (out_docs) memcpy(out_docs, h32 + o_docs, outn * 4);
    if (out_scores) memcpy(out_scores, h32 + o_scores, outn * 4);
    if (out_count) memcpy(out_count, h32 + o_count, (size_t)nq * 4);
    if (out_stats) memcpy(out_stats, h32 + o_stats, (size_t)nq * 16);
    const int32_t* flags = h32 + o_flags;
    int failed = 0, first = -1;
    for (int i = 0; i < nq; i++) {
        const uint32_t f = (uint32_t)flags[i];
        if (ex.out_flags) ex.out_flags[i] = (int32_t)(f & (JV_FLAG_BIG | JV_FLAG_EARLY));
        if (f & (JV_FLAG_FAILED | JV_FLAG_OVERFLOW)) {
            if (ex.out_status) ex.out_status[i] = JV_ENOMEM;
            if (first < 0) first = i;
            failed++;
        }
    }
    if (failed)
        return fail(JV_ENOMEM, "%d of %d queries (first: %d) overflowed the HBM scratch; raise option big_cand_cap (the other rows are valid)",
                    failed, nq, first);
    return JV_OK;
}

}  // namespace

extern "C" {

int jv_search_batch(jv_index* index, const float* queries, int32_t nq, int32_t topK, int32_t rerankK,
                    float threshold, float rerankFloor, const uint64_t* accept_doc_words, int64_t accept_num_docs,
                    int32_t* out_nodes, int32_t* out_docs, float* out_scores, int32_t* out_count, int32_t* out_stats) {
    return search_batch_host(index, queries, nq, topK, rerankK, threshold, rerankFloor, accept_doc_words, nullptr,
                             accept_num_docs, out_nodes, out_docs, out_scores, out_count, out_stats);
}

}  // extern "C"

namespace {

// (a JVM delivers signals to its threads: a wait interrupted by one is simply resumed)
void sem_wait_retry(sem_t* s) {
    while (sem_wait(s) == -1 && errno == EINTR) {
    }
}

// hand the leader slot to a queued request that has no leader yet, or give it up (combiner.mu held)
void pass_leadership(Combiner& cb) {
    for (PendingSearch* r : cb.queue) {
        if (!r->promoted) {
            r->promoted = true;
            sem_post(&r->sem);
            return;
        }
    }
    cb.active--;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// one query per call; concurrent calls on one handle are combined into batch launches (group commit)
int search_single(jv_index* index, const float* query, int32_t topK, int32_t rerankK, float threshold,
                  float rerankFloor, const uint64_t* accept_doc_words, int64_t accept_num_docs, int64_t visit_limit,
                  uint64_t accept_key, int32_t* out_nodes, int32_t* out_docs, float* out_scores, int32_t* out_count,
                  int32_t* out_stats, int32_t* out_flags) {
    int rc = check_common(index, query, 1, topK, rerankK, threshold);
    if (rc != JV_OK) return rc;
    if (accept_doc_words && accept_num_docs < 0) return fail(JV_EINVAL, "accept_num_docs < 0");
    if (visit_limit < 0) return fail(JV_EINVAL, "visit_limit < 0");
    // degenerate calls are not combined
    if (OPT(index, OPT_COMBINE) == 0 || topK == 0 || index->dev.n == 0 || index->dev.entry < 0) {
        HostSearchExtras ex;
        ex.visit_limit = visit_limit;
        ex.accept_key = accept_key;
        ex.out_flags = out_flags;
        return search_batch_host(index, query, 1, topK, rerankK, threshold, rerankFloor, accept_doc_words, nullptr,
                                 accept_num_docs, out_nodes, out_docs, out_scores, out_count, out_stats, ex);
    }
    if (threshold <= 0.0f) {
        // shapes the pool kernels run: no launch at all, a device-resident server answers (unfiltered: the several-waves kernel;
        // with a doc filter: the one-wave filtered instances, the bits served from the filter cache)
        rc = serve_query(index, query, topK, rerankK, rerankFloor, visit_limit, out_nodes, out_docs, out_scores, out_count, out_stats, out_flags,
                         accept_doc_words, accept_doc_words ? accept_num_docs : 0, accept_key);
        if (rc <= 0) return rc;  // (1 = not eligible / a row for the ladder: the launch path below)
    }
    Combiner& cb = index->combiner;
    PendingSearch me;
    me.query = query;
    me.topK = topK;
    me.rerankK = rerankK;
    me.threshold = threshold;
    me.rerankFloor = rerankFloor;
    me.out_nodes = out_nodes;
    me.out_docs = out_docs;
    me.out_scores = out_scores;
    me.out_count = out_count;
    me.out_stats = out_stats;
    me.accept = accept_doc_words;
    me.accept_docs = accept_doc_words ? accept_num_docs : 0;
    me.visit_limit = visit_limit;
    me.out_flags = out_flags;
    me.err[0] = 0;
    sem_init(&me.sem, 0, 0);
    auto finish = [&](int code) {
        sem_destroy(&me.sem);
        if (code != JV_OK) g_last_error = me.err;
        return code;
    };
    std::unique_lock<std::mutex> lk(cb.mu);
    cb.queue.push_back(&me);
    int leaders = (int)OPT(index, OPT_COMBINE_LEADERS);
    if (leaders < 1) leaders = 1;
    if (cb.active < leaders) {
        cb.active++;
        me.promoted = true;
    } else {
        lk.unlock();
        sem_wait_retry(&me.sem);  // promotion, or the answer
        lk.lock();
        if (!me.promoted) return finish(me.rc);
    }
    // ---- leader (lock held) ----
    if (me.state != PendingSearch::QUEUED) {
        // another leader took this request before the promotion was acted on: give the slot away and wait for it
        pass_leadership(cb);
        lk.unlock();
        sem_wait_retry(&me.sem);
        return finish(me.rc);
    }
    static thread_local std::vector<PendingSearch*> batch;
    static thread_local std::vector<float> qbuf, sbuf;
    static thread_local std::vector<int32_t> nbuf, dbuf, cbuf, stbuf, statusbuf, flagbuf;
    static thread_local std::vector<const uint64_t*> abuf;
    batch.clear();
    size_t max_batch = (size_t)std::max<int64_t>(1, OPT(index, OPT_COMBINE_MAX_BATCH));
    if (me.accept) {  // filtered calls batch with filtered calls over the same doc space; bound the staged filter bytes
        const size_t fbytes = (((size_t)me.accept_docs + 63) / 64) * 8 + 8;
        max_batch = std::min(max_batch, std::max<size_t>(1, ((size_t)64 << 20) / fbytes));
    }
    for (auto it = cb.queue.begin(); it != cb.queue.end();) {
        PendingSearch* r = *it;
        const bool same = r == &me || (batch.size() + 1 < max_batch && r->state == PendingSearch::QUEUED && r->topK == topK &&
                                       r->rerankK == rerankK && r->threshold == threshold && r->rerankFloor == rerankFloor &&
                                       (r->accept != nullptr) == (me.accept != nullptr) && r->accept_docs == me.accept_docs &&
                                       r->visit_limit == me.visit_limit);
        if (same) {
            r->state = PendingSearch::TAKEN;
            batch.push_back(r);
            it = cb.queue.erase(it);
        } else {
            ++it;
        }
    }
    lk.unlock();
    const size_t nb = batch.size(), d = (size_t)index->dev.d;
    bool passed = false;
    try {
    if (nb == 1) {
        HostSearchExtras ex;
        ex.visit_limit = visit_limit;
        ex.accept_key = accept_key;
        ex.out_flags = out_flags;
        rc = search_batch_host(index, query, 1, topK, rerankK, threshold, rerankFloor, me.accept, nullptr, me.accept_docs,
                               out_nodes, out_docs, out_scores, out_count, out_stats, ex);
        if (rc != JV_OK) snprintf(me.err, sizeof(me.err), "%s", g_last_error.c_str());
    } else if (OPT(index, OPT_DIRECT_COMPLETION) != 0) {
        // Direct delivery: the kernels write every row into pinned memory and set a completion word per query; this thread
        // (the batch's only poller) hands each answer to its owner the moment it is final — a caller no longer waits for
        // the slowest query of the launch it happened to share — and frees the leader slot as soon as the batch is on the GPU.
        qbuf.resize(nb * d);
        abuf.resize(nb);
        for (size_t i = 0; i < nb; i++) {
            memcpy(qbuf.data() + i * d, batch[i]->query, d * sizeof(float));
            abuf[i] = batch[i]->accept;
        }
        struct Deliver {
            std::vector<PendingSearch*>* batch;
            std::vector<char> served;
            PendingSearch* me;
            Combiner* cb;
            bool* passed;
            int topK;
            int my_rc;
        } dl{&batch, std::vector<char>(nb, 0), &me, &cb, &passed, topK, JV_OK};
        HostSearchExtras ex;
        ex.visit_limit = visit_limit;
        ex.user = &dl;
        ex.on_launched = [](void* u) {
            Deliver* dv = (Deliver*)u;
            std::lock_guard<std::mutex> g(dv->cb->mu);
            pass_leadership(*dv->cb);
            *dv->passed = true;
        };
        ex.on_ready = [](void* u, int i, const int32_t* nodes, const int32_t* docs, const float* scores, int32_t count,
                         const int32_t* stats, int32_t qflags, int status) {
            Deliver* dv = (Deliver*)u;
            PendingSearch* r = (*dv->batch)[(size_t)i];
            if (status == JV_OK) {
                if (r->out_nodes) memcpy(r->out_nodes, nodes, sizeof(int32_t) * dv->topK);
                if (r->out_docs) memcpy(r->out_docs, docs, sizeof(int32_t) * dv->topK);
                if (r->out_scores) memcpy(r->out_scores, scores, sizeof(float) * dv->topK);
                if (r->out_count) *r->out_count = count;
                if (r->out_stats) memcpy(r->out_stats, stats, sizeof(int32_t) * 4);
                if (r->out_flags) *r->out_flags = qflags;
            } else {
                snprintf(r->err, sizeof(r->err), "the query overflowed the HBM scratch; raise option big_cand_cap");
            }
            dv->served[(size_t)i] = 1;
            if (r != dv->me) {
                r->rc = status;
                sem_post(&r->sem);  // r's frame may be gone as soon as this returns
            } else {
                dv->my_rc = status;
            }
        };
        rc = search_batch_host(index, qbuf.data(), (int32_t)nb, topK, rerankK, threshold, rerankFloor, nullptr,
                               me.accept ? abuf.data() : nullptr, me.accept_docs, nullptr, nullptr, nullptr, nullptr, nullptr, ex);
        // a launch-level failure leaves owners unserved: they all get its code
        for (size_t i = 0; i < nb; i++) {
            if (dl.served[i]) continue;
            PendingSearch* r = batch[i];
            snprintf(r->err, sizeof(r->err), "%s", g_last_error.c_str());
            if (r != &me) {
                r->rc = rc;
                sem_post(&r->sem);
            } else {
                dl.my_rc = rc;
            }
        }
        rc = dl.my_rc;
    } else {
        qbuf.resize(nb * d);
        nbuf.resize(nb * topK);
        dbuf.resize(nb * topK);
        sbuf.resize(nb * topK);
        cbuf.resize(nb);
        stbuf.resize(nb * 4);
        abuf.resize(nb);
        for (size_t i = 0; i < nb; i++) {
            memcpy(qbuf.data() + i * d, batch[i]->query, d * sizeof(float));
            abuf[i] = batch[i]->accept;
        }
        statusbuf.assign(nb, 0);
        flagbuf.assign(nb, 0);
        HostSearchExtras ex;
        ex.visit_limit = visit_limit;
        ex.out_status = statusbuf.data();
        ex.out_flags = flagbuf.data();
        rc = search_batch_host(index, qbuf.data(), (int32_t)nb, topK, rerankK, threshold, rerankFloor, nullptr,
                               me.accept ? abuf.data() : nullptr, me.accept_docs, nbuf.data(), dbuf.data(), sbuf.data(),
                               cbuf.data(), stbuf.data(), ex);
        // every caller gets ITS query's status: a query that outgrew even the HBM scratch fails alone, like a
        // failing GraphSearcher.search call in the reference would; a launch-level error (rc set, no per-query
        // status) fails all of them
        const bool per_query = rc == JV_OK || rc == JV_ENOMEM;
        const int batch_rc = rc;
        int my_rc = JV_OK;
        // the answers are in this thread's buffers: free the leader slot BEFORE handing them out, so the next batch
        // is on the GPU while this one's owners are being woken (one futex wake per owner)
        lk.lock();
        pass_leadership(cb);
        lk.unlock();
        passed = true;
        for (size_t i = 0; i < nb; i++) {
            PendingSearch* r = batch[i];
            const int rrc = per_query ? statusbuf[i] : batch_rc;
            if (rrc == JV_OK) {
                if (r->out_nodes) memcpy(r->out_nodes, nbuf.data() + i * topK, sizeof(int32_t) * topK);
                if (r->out_docs) memcpy(r->out_docs, dbuf.data() + i * topK, sizeof(int32_t) * topK);
                if (r->out_scores) memcpy(r->out_scores, sbuf.data() + i * topK, sizeof(float) * topK);
                if (r->out_count) *r->out_count = cbuf[i];
                if (r->out_stats) memcpy(r->out_stats, stbuf.data() + i * 4, sizeof(int32_t) * 4);
                if (r->out_flags) *r->out_flags = flagbuf[i];
            } else if (per_query) {
                snprintf(r->err, sizeof(r->err), "the query overflowed the HBM scratch; raise option big_cand_cap");
            } else {
                snprintf(r->err, sizeof(r->err), "%s", g_last_error.c_str());
            }
            if (r != &me) {
                r->rc = rrc;
                sem_post(&r->sem);  // r's frame may be gone as soon as this returns: r is not touched afterwards
            } else {
                my_rc = rrc;
            }
        }
        rc = my_rc;
    }
    } catch (const std::exception& e) {
        // only the staging-vector resizes can throw, i.e. before any owner has been answered: nobody may be left
        // waiting, so every call of this batch fails with the same error
        rc = JV_ENOMEM;
        snprintf(me.err, sizeof(me.err), "jv_search: %s", e.what());
        if (!passed) {
            lk.lock();
            pass_leadership(cb);
            lk.unlock();
            passed = true;
        }
        for (PendingSearch* r : batch) {
            if (r == &me) continue;
            r->rc = rc;
            snprintf(r->err, sizeof(r->err), "%s", me.err);
            sem_post(&r->sem);
        }
    }
    if (!passed) {
        lk.lock();
        pass_leadership(cb);
        lk.unlock();
    }
    return finish(rc);
}

}  // namespace

extern "C" {

int jv_search(jv_index* index, const float* query, int32_t topK, int32_t rerankK, float threshold,
              float rerankFloor, const uint64_t* accept_doc_words, int64_t accept_num_docs, int32_t* out_nodes,
              int32_t* out_docs, float* out_scores, int32_t* out_count, int32_t* out_stats) {
    return search_single(index, query, topK, rerankK, threshold, rerankFloor, accept_doc_words, accept_num_docs, 0, 0, out_nodes,
                         out_docs, out_scores, out_count, out_stats, nullptr);
}

int jv_search_ex(jv_index* index, const float* query, const jv_search_params* p, int32_t* out_nodes, int32_t* out_docs,
                 float* out_scores, int32_t* out_count, int32_t* out_stats, int32_t* out_flags) {
    if (!p || p->struct_size != sizeof(jv_search_params)) return fail(JV_EINVAL, "jv_search_params is NULL or has the wrong struct_size");
    if (out_flags) *out_flags = 0;
    return search_single(index, query, p->topK, p->rerankK, p->threshold, p->rerankFloor, p->accept_doc_words,
                         p->accept_num_docs, p->visit_limit, p->accept_key, out_nodes, out_docs, out_scores, out_count, out_stats,
                         out_flags);
}

int jv_search_batch_ex(jv_index* index, const float* queries, int32_t nq, const jv_search_params* p, int32_t* out_nodes,
                       int32_t* out_docs, float* out_scores, int32_t* out_count, int32_t* out_stats, int32_t* out_status,
                       int32_t* out_flags) {
    if (!p || p->struct_size != sizeof(jv_search_params)) return fail(JV_EINVAL, "jv_search_params is NULL or has the wrong struct_size");
    if (p->visit_limit < 0) return fail(JV_EINVAL, "visit_limit < 0");
    HostSearchExtras ex;
    ex.visit_limit = p->visit_limit;
    ex.accept_key = p->accept_key;
    ex.out_status = out_status;
    ex.out_flags = out_flags;
    return search_batch_host(index, queries, nq, p->topK, p->rerankK, p->threshold, p->rerankFloor, p->accept_doc_words, nullptr,
                             p->accept_num_docs, out_nodes, out_docs, out_scores, out_count, out_stats, ex);
}

int jv_score_ordinals(jv_index* index, const float* query, const int32_t* ordinals, int32_t count, float* out_scores) {
    if (!index || !query || (count > 0 && (!ordinals || !out_scores))) return fail(JV_EINVAL, "NULL argument");
    if (count <= 0) return JV_OK;
    HIPCHK(hipSetDevice(index->device));
    if (index->dev.n == 0) {
        memset(out_scores, 0, sizeof(float) * (size_t)count);
        return JV_OK;
    }
    Ctx* c = nullptr;
    int rc = ctx_acquire(index, &c);
    if (rc != JV_OK) return rc;
    struct Releaser {
        jv_index* ix;
        Ctx* c;
        ~Releaser() { ctx_release(ix, c); }
    } rel{index, c};
    const int d = index->dev.d;
    // context-owned staging: query in d_queries, ordinals + scores in the result arena (no per-call allocation)
    if ((rc = grow((void**)&c->d_queries, &c->queries_cap, (size_t)d, sizeof(float))) != JV_OK) return rc;
    const size_t need = (size_t)count * 8;
    if (need > c->arena_cap) {
        jv_free(c->d_arena);
        c->d_arena = nullptr;
        jv_host_free(c->h_arena);
        c->h_arena = nullptr;
        c->arena_cap = 0;
        const size_t cap = need + need / 4;
        HIPCHK(hipMalloc((void**)&c->d_arena, cap));
        HIPCHK(hipHostMalloc((void**)&c->h_arena, cap, hipHostMallocDefault));
        c->arena_cap = cap;
    }
    float* dq = c->d_queries;
    int32_t* dord = (int32_t*)c->d_arena;
    float* dout = (float*)(c->d_arena + (size_t)count * 4);
    hipError_t e = hipMemcpyAsync(dq, query, (size_t)d * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dord, ordinals, (size_t)count * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = jvk_launch_score_ordinals(&index->dev, dq, dord, count, dout, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_scores, dout, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(JV_EDEVICE, "jv_score_ordinals: %s", hipGetErrorString(e));
    return JV_OK;
}

// ---------------------------------------------------------------------------------------------
// Batched exact scorer: nq queries against ONE shared candidate set (csrc/jv_kernels_xb.hip has the method and the bound).
// ---------------------------------------------------------------------------------------------
static bool xb_trace() {
    static const bool on = getenv("JV_XB_TRACE") != nullptr;
    return on;
}
static double xb_now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}
// diagnostics (JV_XB_TRACE=1): host-side points of a batched exact call with a clock (which step waits, and for how long)
#define XB_POINT(what)                                                              \
    do {                                                                            \
        if (xb_trace()) fprintf(stderr, "[jvgpu xb %.2f ms] %s\n", xb_now_ms(), what); \
    } while (0)
// diagnostics (JV_XB_TRACE=1): drain the stream after every step and say which one failed
#define XB_STEP(what)                                                                                        \
    do {                                                                                                     \
        if (xb_trace()) {                                                                                    \
            hipError_t e_ = hipStreamSynchronize(ix->xb.stream);                                             \
            fprintf(stderr, "[jvgpu xb %.2f ms] %s: %s\n", xb_now_ms(), what, hipGetErrorString(e_));        \
            if (e_ != hipSuccess) return fail(JV_EDEVICE, "%s: %s", what, hipGetErrorString(e_));            \
        }                                                                                                    \
    } while (0)
#define XB_ROUND_QUERIES 1024   /* queries per round (8 panels of 128): the candidates are read once per round */
#define XB_SURV_CAP 16384       /* survivor slots per query; a query that overflows scans the whole list (one workgroup: the backstop, not a path —
                                   at 5M candidates a 4 096-slot list overflowed for 2 of 256 queries and their scans took 320 ms, round 5) */

extern "C++" {
template <typename T>
static int xb_grow(XbState& x, T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return JV_OK;
    if (*p) {
        jv_free(*p);
        x.bytes -= (int64_t)(*cap * sizeof(T));
    }
    *p = nullptr;
    *cap = 0;
    const size_t ncap = need < 256 ? 256 : need + need / 4;
    HIPCHK(hipMalloc((void**)p, ncap * sizeof(T)));
    *cap = ncap;
    x.bytes += (int64_t)(ncap * sizeof(T));
    return JV_OK;
}
}  // extern "C++"

// the bf16 mirror: [n][kp] + |v|^2, built once (n * (2 kp + 4) bytes: 15.4 GB for 10M x 768)
static int xb_ensure_mirror(jv_index* ix) {
    XbState& x = ix->xb;
    if (x.mirror_state != 0) return JV_OK;
    const JvIndexDev& dv = ix->dev;
    x.mirror_state = -1;
    if (!dv.vectors || dv.n <= 0) return JV_OK;  // NVQ-only field: every batch takes the canonical scan
    const int kp = (dv.d + 63) / 64 * 64;
    uint16_t* vb = nullptr;
    float* vn = nullptr;
    if (hipMalloc((void**)&vb, (size_t)dv.n * (size_t)kp * 2) != hipSuccess || hipMalloc((void**)&vn, (size_t)dv.n * 4) != hipSuccess) {
        (void)hipGetLastError();
        if (vb) jv_free(vb);
        return JV_OK;  // no room: not an error, the canonical scan answers
    }
    hipError_t e = jvk_xb_mirror(dv.vectors, dv.n, dv.d, dv.stride, kp, vb, vn, 0, x.stream);
    if (e == hipSuccess) e = jvk_xb_mark_dead(dv.ord2doc, dv.n, vn, x.stream);   // (norm2 = -1: a deleted ordinal is no candidate of the pre-filter)
    if (e == hipSuccess) e = hipStreamSynchronize(x.stream);
    if (e != hipSuccess) {
        jv_free(vb);
        jv_free(vn);
        return fail(JV_EDEVICE, "bf16 mirror: %s", hipGetErrorString(e));
    }
    x.vb = vb;
    x.vnorm2 = vn;
    x.kp = kp;
    x.bytes += (int64_t)dv.n * ((int64_t)kp * 2 + 4);
    x.mirror_state = 1;
    return JV_OK;
}

// one pass of the matrix-core kernel over ta.rows candidates x ta.B queries: the query-stationary kernel (rounds of 256 queries,
// candidates streamed as whole rows) where the k range fits a wave's registers, else the LDS-tiled one (option "xb_no_qs" / env
// JV_XB_NO_QS force the latter: A/B runs)
static int xb_pass(jv_index* ix, JvXbTileArgs& ta, int mode) {
    static const bool no_qs = getenv("JV_XB_NO_QS") != nullptr;
    static const int dbg = getenv("JV_XB_DBG") ? atoi(getenv("JV_XB_DBG")) : 0;
    ta.dbg = dbg;
    // JV_XB_STAMPS=1: cycles per phase of the query-stationary kernel (wave 1 of every workgroup), printed per pass
    static const bool stamps = getenv("JV_XB_STAMPS") != nullptr;
    static unsigned long long* d_stamps = nullptr;
    if (stamps && !d_stamps) HIPCHK(hipMalloc((void**)&d_stamps, 64));
    if (stamps) HIPCHK(hipMemsetAsync(d_stamps, 0, 64, ix->xb.stream));
    ta.stamps = stamps ? d_stamps : nullptr;
    if (!no_qs && jvk_xb_qs_ok(ta.kp)) {
        for (int q0 = 0; q0 < ta.B; q0 += 256) {
            ta.qbase = q0;
            HIPCHK(jvk_xb_qs(&ta, mode, ix->cu_count, ix->xb.stream));
        }
        ta.qbase = 0;
        if (stamps) {
            unsigned long long h[8];
            HIPCHK(hipMemcpyAsync(h, d_stamps, 64, hipMemcpyDeviceToHost, ix->xb.stream));
            HIPCHK(hipStreamSynchronize(ix->xb.stream));
            const int nsub = (ta.rows + 31) / 32;
            const double per = (double)std::max(1, nsub);
            fprintf(stderr, "[jvgpu xb stamps] mode %d rows %d: cycles per stage (summed over workgroups / stages): wait+barrier A %.0f, requests %.0f, multiply %.0f, epilogue %.0f, barrier B %.0f\n",
                    mode, ta.rows, h[0] / per, h[1] / per, h[2] / per, h[3] / per, h[4] / per);
        }
        return JV_OK;
    }
    HIPCHK(jvk_xb_tile(&ta, mode, ix->xb.stream));
    return JV_OK;
}

// The matrix-core kernels take a whole CU's LDS (query-stationary: 163 840 B at kp = 768; LDS-tiled ~75 KB; re-score ~56 KB): beside
// a live query-server grid (serve_wgs_per_cu x its pool on EVERY CU) their workgroups are never placed until the grid idles out —
// under steady one-query traffic never (the default JVectorKnnFloatVectorQuery::exactSearch of the host mirror
// comes through here while other searcher threads keep the grid alive).  So for the length of a batched exact call the device's
// grids are asked to leave and kept from restarting: one-query callers that arrive meanwhile wait on the servers' launch
// mutexes for the milliseconds the scan takes, then restart their grid.  nullptr when the device has no server (the usual case).
static std::unique_ptr<ServerPause> xb_pause_servers(jv_index* ix) {
    {
        std::lock_guard<std::mutex> g(g_servers_mu);
        bool any = false;
        for (Server* sv : g_servers) any = any || sv->ix->device == ix->device;
        if (!any) return nullptr;
    }
    return std::unique_ptr<ServerPause>(new ServerPause(ix->device));
}

// d_* = device pointers on the index's device; d_ords = nullptr: every ordinal is a candidate (C = n).  Enqueues on x.stream.
static int xb_run(jv_index* ix, const float* d_queries, int nq, int topK, const int32_t* d_ords, int C, uint32_t flags,
                  int32_t* d_nodes, int32_t* d_docs, float* d_scores, int32_t* d_count, int64_t* info) {
    XbState& x = ix->xb;
    const JvIndexDev& dv = ix->dev;
    hipStream_t st = x.stream;
    if (!x.d_info) {
        HIPCHK(hipMalloc((void**)&x.d_info, 4 * sizeof(int64_t)));
        HIPCHK(hipHostMalloc((void**)&x.h_info, 4 * sizeof(int64_t), hipHostMallocDefault));
    }
    HIPCHK(hipMemsetAsync(x.d_info, 0, 4 * sizeof(int64_t), st));
    int S = 0;
    bool pre = !(flags & JV_XB_NO_PREFILTER) && (C >= 2048 || ((flags & JV_XB_FORCE_PREFILTER) && C >= 1));
    if (pre) {
        int rc = xb_ensure_mirror(ix);
        if (rc != JV_OK) return rc;
        pre = x.mirror_state == 1;
    }
    if (pre) {
        S = std::min(C, std::max(4096, std::min(65536, C / 8)));   // (about k * C / S candidates clear the bar of a k-of-S sample)
        if ((int64_t)topK * 4 > S && !(flags & JV_XB_FORCE_PREFILTER)) pre = false;
    }
    const int kp = x.kp;
    // kappa: (2u + u^2) with u = 2^-8 (bf16 round to nearest even, both operands) + fp32 accumulation of kp products in
    // the matrix pipe, whatever its order and rounding (4 kp 2^-24 of sum |q c|: four times the round-to-nearest chain) + the
    // fp32 norms' own error
    const float kappa = (float)((2.0 / 256.0 + 1.0 / 65536.0) * 1.001 + 4.0 * (double)kp / 16777216.0 + 1e-5);
    for (int q0 = 0; q0 < nq; q0 += XB_ROUND_QUERIES) {
        const int B = std::min(XB_ROUND_QUERIES, nq - q0);
        const float* dq = d_queries + (size_t)q0 * dv.d;
        JvXbRescoreArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.queries = dq;
        ra.ords = d_ords;
        ra.C = C;
        ra.topK = topK;
        ra.out_nodes = d_nodes ? d_nodes + (size_t)q0 * topK : nullptr;
        ra.out_docs = d_docs ? d_docs + (size_t)q0 * topK : nullptr;
        ra.out_scores = d_scores ? d_scores + (size_t)q0 * topK : nullptr;
        ra.out_count = d_count ? d_count + q0 : nullptr;
        ra.out_info = x.d_info;
        if (pre) {
            const int panels = (B + 127) / 128;
            const size_t qrows = (size_t)(B + 255) / 256 * 256;   // (the query-stationary kernel reads whole rounds of 256 rows)
            int rc;
            if ((rc = xb_grow(x, &x.d_qb, &x.qb_cap, qrows * kp)) != JV_OK) return rc;
            if (x.round_cap < (size_t)panels * 128) {
                size_t c1 = x.round_cap, c2 = x.round_cap, c3 = x.round_cap;
                if ((rc = xb_grow(x, &x.d_qn2, &c1, (size_t)panels * 128)) != JV_OK) return rc;
                if ((rc = xb_grow(x, &x.d_thr, &c2, (size_t)panels * 128)) != JV_OK) return rc;
                if ((rc = xb_grow(x, &x.d_surv_cnt, &c3, (size_t)panels * 128)) != JV_OK) return rc;
                x.round_cap = std::min(c1, std::min(c2, c3));
            }
            if ((rc = xb_grow(x, &x.d_sample, &x.sample_cap, (size_t)B * S)) != JV_OK) return rc;
            if ((rc = xb_grow(x, &x.d_surv, &x.surv_cap, (size_t)B * XB_SURV_CAP)) != JV_OK) return rc;
            HIPCHK(hipMemsetAsync(x.d_qb, 0, qrows * kp * 2, st));
            HIPCHK(jvk_xb_mirror(dq, B, dv.d, dv.d, kp, x.d_qb, x.d_qn2, 1, st));
            HIPCHK(hipMemsetAsync(x.d_surv_cnt, 0, (size_t)B * 4, st));
            XB_STEP("bf16 queries");
            JvXbTileArgs ta;
            memset(&ta, 0, sizeof(ta));
            ta.vb = x.vb;
            ta.vnorm2 = x.vnorm2;
            ta.kp = kp;
            ta.n = dv.n;
            ta.ords = d_ords;
            ta.C = C;
            ta.qb = x.d_qb;
            ta.qnorm2 = x.d_qn2;
            ta.B = B;
            ta.panels = panels;
            ta.sim = dv.sim;
            ta.kappa = kappa;
            // pass A: the strided sample -> per-query bar
            ta.cstride = C / S;
            ta.rows = S;
            ta.sample = x.d_sample;
            ta.sample_ld = S;
            if ((rc = xb_pass(ix, ta, 0)) != JV_OK) return rc;
            XB_STEP("sample pass");
            HIPCHK(jvk_xb_kth(x.d_sample, S, S, topK, x.d_thr, B, st));
            XB_STEP("k-th select");
            // pass B: every candidate against the bar
            ta.cstride = 1;
            ta.rows = C;
            ta.thr = x.d_thr;
            ta.surv_cnt = x.d_surv_cnt;
            ta.surv = x.d_surv;
            ta.surv_cap = XB_SURV_CAP;
            if ((rc = xb_pass(ix, ta, 1)) != JV_OK) return rc;
            XB_STEP("filter pass");
            ra.surv_cnt = x.d_surv_cnt;
            ra.surv = x.d_surv;
            ra.surv_cap = XB_SURV_CAP;
        } else {
            ra.force_all = 1;
        }
        HIPCHK(jvk_xb_rescore(&dv, &ra, B, st));
        XB_STEP("re-score");
    }
    if (info) {
        HIPCHK(hipMemcpyAsync(x.h_info, x.d_info, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        info[0] = C;
        info[1] = pre ? S : 0;
        info[2] = x.h_info[0];
        info[3] = x.h_info[1];
    }
    return JV_OK;
}

static int xb_check(jv_index* index, const void* queries, int32_t nq, const jv_exact_batch_params* p) {
    if (!index || !p) return fail(JV_EINVAL, "NULL argument");
    if (p->struct_size != sizeof(jv_exact_batch_params)) return fail(JV_EINVAL, "jv_exact_batch_params has the wrong struct_size");
    if (nq < 0 || (nq > 0 && !queries)) return fail(JV_EINVAL, "bad query batch");
    if (p->topK < 1 || p->topK > JV_XB_TOPK_MAX) return fail(p->topK < 1 ? JV_EINVAL : JV_EUNSUPPORTED, "topK must be 1..%d", JV_XB_TOPK_MAX);
    if (p->accept_doc_words && p->accept_num_docs <= 0) return fail(JV_EINVAL, "accept_num_docs <= 0");
    if (!p->accept_doc_words && p->count < 0) return fail(JV_EINVAL, "count < 0");
    if (!p->accept_doc_words && p->count > 0 && !p->ordinals) return fail(JV_EINVAL, "ordinals is NULL");
    return JV_OK;
}

// candidate list of a call: device list pointer (nullptr = identity) and its length
static int xb_candidates(jv_index* ix, const jv_exact_batch_params* p, bool device_ptrs, const int32_t** d_ords, int* C, int* filter_slot) {
    XbState& x = ix->xb;
    const JvIndexDev& dv = ix->dev;
    *filter_slot = -1;
    *d_ords = nullptr;
    int rc;
    if (p->accept_doc_words) {
        const size_t nwords = (size_t)((p->accept_num_docs + 63) / 64);
        const uint64_t* d_acc = nullptr;
        if (device_ptrs) {
            d_acc = p->accept_doc_words;
        } else {
            XB_POINT("candidates: filter_acquire");
            if ((rc = filter_acquire(ix, p->accept_doc_words, nwords, p->accept_key, x.stream, &d_acc, filter_slot)) != JV_OK) return rc;
            XB_POINT("candidates: filter acquired");
            if (*filter_slot < 0) {
                if ((rc = xb_grow(x, &x.d_accept, &x.accept_cap, nwords)) != JV_OK) return rc;
                HIPCHK(hipMemcpyAsync(x.d_accept, p->accept_doc_words, nwords * 8, hipMemcpyHostToDevice, x.stream));
                d_acc = x.d_accept;
            }
        }
        const int nb = jvk_xb_list_blocks(dv.n);
        if ((rc = xb_grow(x, &x.d_counts, &x.counts_cap, (size_t)nb + 1)) != JV_OK) return rc;
        if ((rc = xb_grow(x, &x.d_list, &x.list_cap, (size_t)std::max(dv.n, 1))) != JV_OK) return rc;
        if (!x.h_info) {
            HIPCHK(hipMalloc((void**)&x.d_info, 4 * sizeof(int64_t)));
            HIPCHK(hipHostMalloc((void**)&x.h_info, 4 * sizeof(int64_t), hipHostMallocDefault));
        }
        XB_POINT("candidates: list kernels");
        HIPCHK(jvk_xb_build_list(&dv, d_acc, p->accept_num_docs, x.d_counts, x.d_list, x.stream));
        HIPCHK(hipMemcpyAsync(x.h_info + 2, x.d_counts + nb, 4, hipMemcpyDeviceToHost, x.stream));
        XB_POINT("candidates: enqueued, waiting");
        HIPCHK(hipStreamSynchronize(x.stream));  // the grid of the tile kernel is sized by the filter's cardinality
        XB_POINT("candidates: list ready");
        *C = (int)(*(int32_t*)(x.h_info + 2));
        *d_ords = x.d_list;
    } else if (p->ordinals) {
        *C = p->count;
        if (device_ptrs) {
            *d_ords = p->ordinals;
        } else {
            if ((rc = xb_grow(x, &x.d_list, &x.list_cap, (size_t)std::max(p->count, 1))) != JV_OK) return rc;
            HIPCHK(hipMemcpyAsync(x.d_list, p->ordinals, (size_t)p->count * 4, hipMemcpyHostToDevice, x.stream));
            *d_ords = x.d_list;
        }
    } else {
        *C = dv.n;  // no filter, no list: every ordinal (brute force)
    }
    return JV_OK;
}

static int xb_prepare(jv_index* ix) {
    XbState& x = ix->xb;
    HIPCHK(hipSetDevice(ix->device));
    if (!x.stream) HIPCHK(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking));
    return JV_OK;
}

int jv_score_ordinals_batch(jv_index* index, const float* queries, int32_t nq, const jv_exact_batch_params* p, int32_t* out_nodes,
                            int32_t* out_docs, float* out_scores, int32_t* out_count, int64_t* out_info) {
    int rc = xb_check(index, queries, nq, p);
    if (rc != JV_OK) return rc;
    if (out_info) memset(out_info, 0, JV_XB_INFO_WORDS * sizeof(int64_t));
    if (nq == 0) return JV_OK;
    XbState& x = index->xb;
    std::lock_guard<std::mutex> lk(x.mu);
    XB_POINT("host call: lock taken");
    if ((rc = xb_prepare(index)) != JV_OK) return rc;
    const int topK = p->topK, d = index->dev.d;
    if (index->dev.n == 0) {
        if (out_count) memset(out_count, 0, (size_t)nq * 4);
        for (size_t i = 0; i < (size_t)nq * topK; i++) {
            if (out_nodes) out_nodes[i] = -1;
            if (out_docs) out_docs[i] = -1;
            if (out_scores) out_scores[i] = 0.0f;
        }
        return JV_OK;
    }
    // From the first kernel of the call (the filter's ordinal list) to the results on the host no grid runs or STARTS on this device:
    // measured (tools/exact_beside_server_probe.py, JV_XB_TRACE=1), work enqueued a moment after a grid's restart — the list kernels
    // of the call after the one whose pause had just ended — waited 12 s, until the one-query traffic stopped and the grid idled out.
    XB_POINT("host call: pausing the servers");
    std::unique_ptr<ServerPause> pause = xb_pause_servers(index);
    XB_POINT("host call: paused");
    const int32_t* d_ords = nullptr;
    int C = 0, slot = -1;
    rc = xb_candidates(index, p, false, &d_ords, &C, &slot);
    struct Rel {
        jv_index* ix;
        int* slot;
        ~Rel() { filter_release(ix, *slot); }
    } rel{index, &slot};
    if (rc != JV_OK) return rc;
    if ((rc = xb_grow(x, &x.d_queries, &x.queries_cap, (size_t)nq * d)) != JV_OK) return rc;
    const size_t rows = (size_t)nq * topK;
    const size_t need = rows * 12 + (size_t)nq * 4;
    if (need > x.out_cap || !x.d_out) {
        jv_free(x.d_out);
        jv_host_free(x.h_out);
        x.d_out = nullptr;
        x.h_out = nullptr;
        x.out_cap = 0;
        const size_t cap = need + need / 4;
        HIPCHK(hipMalloc((void**)&x.d_out, cap));
        HIPCHK(hipHostMalloc((void**)&x.h_out, cap, hipHostMallocDefault));
        x.out_cap = cap;
    }
    int32_t* dn = (int32_t*)x.d_out;
    int32_t* dd = dn + rows;
    float* ds = (float*)(dd + rows);
    int32_t* dc = (int32_t*)(ds + rows);
    HIPCHK(hipMemcpyAsync(x.d_queries, queries, (size_t)nq * d * 4, hipMemcpyHostToDevice, x.stream));
    int64_t info[4] = {0, 0, 0, 0};
    if ((rc = xb_run(index, x.d_queries, nq, topK, d_ords, C, p->flags, dn, dd, ds, dc, out_info ? info : nullptr)) != JV_OK) {
        hipStreamSynchronize(x.stream);
        return rc;
    }
    HIPCHK(hipMemcpyAsync(x.h_out, x.d_out, need, hipMemcpyDeviceToHost, x.stream));
    HIPCHK(hipStreamSynchronize(x.stream));
    XB_POINT("host call: results on the host");
    if (out_nodes) memcpy(out_nodes, x.h_out, rows * 4);
    if (out_docs) memcpy(out_docs, x.h_out + rows * 4, rows * 4);
    if (out_scores) memcpy(out_scores, x.h_out + rows * 8, rows * 4);
    if (out_count) memcpy(out_count, x.h_out + rows * 12, (size_t)nq * 4);
    if (out_info) memcpy(out_info, info, sizeof(info));
    return JV_OK;
}

int jv_score_ordinals_batch_device(jv_index* index, const float* d_queries, int32_t nq, const jv_exact_batch_params* p,
                                   int32_t* d_out_nodes, int32_t* d_out_docs, float* d_out_scores, int32_t* d_out_count,
                                   int64_t* out_info, void* hip_stream) {
    int rc = xb_check(index, d_queries, nq, p);
    if (rc != JV_OK) return rc;
    if (out_info) memset(out_info, 0, JV_XB_INFO_WORDS * sizeof(int64_t));
    if (nq == 0) return JV_OK;
    if (index->dev.n == 0) return fail(JV_EINVAL, "empty index: nothing to score on the device path");
    XbState& x = index->xb;
    std::lock_guard<std::mutex> lk(x.mu);
    if ((rc = xb_prepare(index)) != JV_OK) return rc;
    hipStream_t caller = (hipStream_t)hip_stream;
    // the library's stream starts behind what the caller's stream has enqueued (the queries), and the caller's stream
    // continues behind the library's work
    hipEvent_t ev = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    struct EvGuard {
        hipEvent_t e;
        ~EvGuard() { hipEventDestroy(e); }
    } evg{ev};
    HIPCHK(hipEventRecord(ev, caller));
    HIPCHK(hipStreamWaitEvent(x.stream, ev, 0));
    const int32_t* d_ords = nullptr;
    int C = 0, slot = -1;
    std::unique_ptr<ServerPause> pause = xb_pause_servers(index);   // (before the call's first kernel: see jv_score_ordinals_batch)
    if ((rc = xb_candidates(index, p, true, &d_ords, &C, &slot)) != JV_OK) return rc;
    if ((rc = xb_run(index, d_queries, nq, p->topK, d_ords, C, p->flags, d_out_nodes, d_out_docs, d_out_scores, d_out_count, out_info)) != JV_OK) {
        hipStreamSynchronize(x.stream);
        return rc;
    }
    HIPCHK(hipEventRecord(ev, x.stream));
    HIPCHK(hipStreamWaitEvent(caller, ev, 0));
    // (beside query servers the call is synchronous: a grid restarted between two of the enqueued passes would hold the LDS the
    //  later pass needs until it idles out)
    if (!hip_stream || pause) HIPCHK(hipStreamSynchronize(x.stream));
    return JV_OK;
}

// One caller's jv_exact_search waiting to be served (lives on the caller's stack)
struct XbPending {
    const float* query;
    int32_t topK;
    const uint64_t* words;
    int64_t ndocs;
    uint64_t key;
    int32_t* out_nodes;
    int32_t* out_docs;
    float* out_scores;
    int32_t* out_count;
    int rc = 0;
    char err[256];
    bool done = false, promoted = false;
    std::condition_variable cv;
};

// ONE query under a doc filter — what Lucene's exactSearch is per leaf and query (AbstractKnnVectorQuery.exactSearch over
// JVectorVectorScorer.score, J/JVectorVectorScorer.java:36-53) and how the reference issues it: one call per searcher thread
// (T/index/engine/JVectorConcurrentQueryTests.java:78-138).  Calls in flight at the same time that carry the SAME filter (same key or
// content hash, same length, bits compared) and the same topK are answered as ONE jv_score_ordinals_batch call: the first arrival
// leads; while its batch runs, later arrivals queue, and the oldest of them leads the next batch.  A lone caller pays no delay.
int jv_exact_search(jv_index* index, const float* query, const jv_exact_batch_params* p, int32_t* out_nodes, int32_t* out_docs,
                    float* out_scores, int32_t* out_count) {
    int rc = xb_check(index, query, 1, p);
    if (rc != JV_OK) return rc;
    if (!p->accept_doc_words) return jv_score_ordinals_batch(index, query, 1, p, out_nodes, out_docs, out_scores, out_count, nullptr);  // (lists are not combined)
    XbState& x = index->xb;
    x.exact_calls++;
    XbPending me;
    me.query = query;
    me.topK = p->topK;
    me.words = p->accept_doc_words;
    me.ndocs = p->accept_num_docs;
    const size_t nwords = (size_t)((p->accept_num_docs + 63) / 64);
    me.key = p->accept_key ? p->accept_key : hash_words(p->accept_doc_words, nwords);
    me.out_nodes = out_nodes;
    me.out_docs = out_docs;
    me.out_scores = out_scores;
    me.out_count = out_count;
    me.err[0] = 0;
    std::unique_lock<std::mutex> lk(x.cmu);
    if (x.cleader) {
        x.cqueue.push_back(&me);
        me.cv.wait(lk, [&] { return me.done || me.promoted; });
        if (me.done) {
            if (me.rc != JV_OK) g_last_error = me.err;
            return me.rc;
        }
    } else {
        x.cleader = true;
    }
    // ---- leader: every queued call with this filter and topK, this one included ----
    // Under the lock candidates are only FOUND (key, length, topK); the bits decide, and they are compared after the lock is
    // dropped (a 10M-doc filter is 1.25 MB: round 5 compared it for every queued caller while every arrival waited on x.cmu).
    // A call whose bits differ goes back to the head of the queue.  Nothing in the leader section may leave the lead taken: an
    // exception (std::bad_alloc from the staging vectors) becomes JV_ENOMEM and the hand-over below still runs.
    std::vector<XbPending*> mine{&me};
    for (auto it = x.cqueue.begin(); it != x.cqueue.end() && mine.size() < XB_ROUND_QUERIES;) {
        XbPending* o = *it;
        if (o->topK == me.topK && o->key == me.key && o->ndocs == me.ndocs) {
            mine.push_back(o);
            it = x.cqueue.erase(it);
        } else {
            ++it;
        }
    }
    lk.unlock();
    std::vector<XbPending*> back;
    for (size_t i = 1; i < mine.size();) {
        XbPending* o = mine[i];
        if (o->words != me.words && memcmp(o->words, me.words, nwords * 8) != 0) {
            back.push_back(o);
            mine.erase(mine.begin() + (long)i);
        } else {
            i++;
        }
    }
    if (!back.empty()) {
        lk.lock();
        x.cqueue.insert(x.cqueue.begin(), back.begin(), back.end());
        lk.unlock();
    }
    const int nq = (int)mine.size(), d = index->dev.d, k = me.topK;
    int brc = JV_OK;
    std::string berr;
    try {
        if (nq == 1) {
            brc = jv_score_ordinals_batch(index, query, 1, p, out_nodes, out_docs, out_scores, out_count, nullptr);
            if (brc != JV_OK) berr = g_last_error;
        } else {
            std::vector<float> q((size_t)nq * d);
            std::vector<int32_t> nodes((size_t)nq * k), docs((size_t)nq * k), count((size_t)nq);
            std::vector<float> scores((size_t)nq * k);
            for (int i = 0; i < nq; i++) memcpy(q.data() + (size_t)i * d, mine[(size_t)i]->query, (size_t)d * sizeof(float));
            jv_exact_batch_params bp = *p;
            bp.accept_key = me.key;
            brc = jv_score_ordinals_batch(index, q.data(), nq, &bp, nodes.data(), docs.data(), scores.data(), count.data(), nullptr);
            if (brc != JV_OK) berr = g_last_error;
            else
                for (int i = 0; i < nq; i++) {
                    XbPending* o = mine[(size_t)i];
                    if (o->out_nodes) memcpy(o->out_nodes, nodes.data() + (size_t)i * k, (size_t)k * 4);
                    if (o->out_docs) memcpy(o->out_docs, docs.data() + (size_t)i * k, (size_t)k * 4);
                    if (o->out_scores) memcpy(o->out_scores, scores.data() + (size_t)i * k, (size_t)k * 4);
                    if (o->out_count) *o->out_count = count[(size_t)i];
                }
        }
    } catch (const std::exception& e) {
        brc = JV_ENOMEM;
        berr = std::string("jv_exact_search: ") + e.what();
    }
    x.exact_batches++;
    lk.lock();
    for (size_t i = 1; i < mine.size(); i++) {
        XbPending* o = mine[i];
        o->rc = brc;
        snprintf(o->err, sizeof(o->err), "%s", berr.c_str());
        o->done = true;
        o->cv.notify_one();
    }
    if (!x.cqueue.empty()) {   // hand the lead to the oldest waiting call
        XbPending* nx = x.cqueue.front();
        x.cqueue.pop_front();
        nx->promoted = true;
        nx->cv.notify_one();
    } else {
        x.cleader = false;
    }
    lk.unlock();
    if (brc != JV_OK) g_last_error = berr;
    return brc;
}

// Diagnostics (tests/test_gpu_xb.py; not part of include/jvgpu.h): the interval [lower, upper] the matrix-core pass
// computes for every (query, list entry) pair — the canonical fp32 raw value (dot, -squared distance, cosine) must lie
// inside it, or the pre-filter could discard a true neighbour.  Host pointers, nq <= 1024, out_* [nq][count].
int jv_xb_debug_bounds(jv_index* index, const float* queries, int32_t nq, const int32_t* ordinals, int32_t count, float* out_lower,
                       float* out_upper, float* out_kappa) {
    if (!index || !queries || !ordinals || nq < 1 || nq > XB_ROUND_QUERIES || count < 1) return fail(JV_EINVAL, "bad argument");
    XbState& x = index->xb;
    std::lock_guard<std::mutex> lk(x.mu);
    int rc;
    if ((rc = xb_prepare(index)) != JV_OK) return rc;
    if ((rc = xb_ensure_mirror(index)) != JV_OK) return rc;
    if (x.mirror_state != 1) return fail(JV_EUNSUPPORTED, "no bf16 mirror for this index");
    const JvIndexDev& dv = index->dev;
    const int kp = x.kp, panels = (nq + 127) / 128;
    hipStream_t st = x.stream;
    if ((rc = xb_grow(x, &x.d_queries, &x.queries_cap, (size_t)nq * dv.d)) != JV_OK) return rc;
    if ((rc = xb_grow(x, &x.d_list, &x.list_cap, (size_t)count)) != JV_OK) return rc;
    const size_t qrows = (size_t)(nq + 255) / 256 * 256;
    if ((rc = xb_grow(x, &x.d_qb, &x.qb_cap, qrows * kp)) != JV_OK) return rc;
    if (x.round_cap < (size_t)panels * 128) {
        size_t c1 = x.round_cap, c2 = x.round_cap, c3 = x.round_cap;
        if ((rc = xb_grow(x, &x.d_qn2, &c1, (size_t)panels * 128)) != JV_OK) return rc;
        if ((rc = xb_grow(x, &x.d_thr, &c2, (size_t)panels * 128)) != JV_OK) return rc;
        if ((rc = xb_grow(x, &x.d_surv_cnt, &c3, (size_t)panels * 128)) != JV_OK) return rc;
        x.round_cap = std::min(c1, std::min(c2, c3));
    }
    if ((rc = xb_grow(x, &x.d_sample, &x.sample_cap, (size_t)nq * count)) != JV_OK) return rc;
    HIPCHK(hipMemcpyAsync(x.d_queries, queries, (size_t)nq * dv.d * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(x.d_list, ordinals, (size_t)count * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(x.d_qb, 0, qrows * kp * 2, st));
    HIPCHK(jvk_xb_mirror(x.d_queries, nq, dv.d, dv.d, kp, x.d_qb, x.d_qn2, 1, st));
    JvXbTileArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.vb = x.vb;
    ta.vnorm2 = x.vnorm2;
    ta.kp = kp;
    ta.n = dv.n;
    ta.ords = x.d_list;
    ta.C = count;
    ta.cstride = 1;
    ta.rows = count;
    ta.qb = x.d_qb;
    ta.qnorm2 = x.d_qn2;
    ta.B = nq;
    ta.panels = panels;
    ta.sim = dv.sim;
    ta.kappa = (float)((2.0 / 256.0 + 1.0 / 65536.0) * 1.001 + 4.0 * (double)kp / 16777216.0 + 1e-5);
    ta.sample = x.d_sample;
    ta.sample_ld = count;
    if (out_kappa) *out_kappa = ta.kappa;
    std::unique_ptr<ServerPause> pause = xb_pause_servers(index);
    for (int mode = 0; mode <= 2; mode += 2) {
        float* dst = mode == 0 ? out_lower : out_upper;
        if (!dst) continue;
        if ((rc = xb_pass(index, ta, mode)) != JV_OK) return rc;
        HIPCHK(hipMemcpyAsync(dst, x.d_sample, (size_t)nq * count * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return JV_OK;
}

int jv_merge_topk_device(int32_t device, const int32_t* d_docs, const float* d_scores, int32_t nq, int32_t lists,
                         int32_t k, int32_t* d_out_docs, float* d_out_scores, void* hip_stream) {
    if (nq < 0 || lists <= 0 || k <= 0) return fail(JV_EINVAL, "bad merge shape");
    if (nq == 0) return JV_OK;
    if (!d_docs || !d_scores || !d_out_docs || !d_out_scores) return fail(JV_EINVAL, "NULL argument");
    if ((int64_t)lists * k * 8 > 64 * 1024) return fail(JV_EUNSUPPORTED, "lists*k too large for one LDS tile");
    HIPCHK(hipSetDevice(device));
    HIPCHK(jvk_launch_merge_topk(d_docs, d_scores, nq, lists, k, d_out_docs, d_out_scores, (hipStream_t)hip_stream));
    if (!hip_stream) HIPCHK(hipStreamSynchronize(nullptr));
    return JV_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// doc-ID-range sharding inside one process; see include/jvgpu.h
// ---------------------------------------------------------------------------------------------------------------
}  // extern "C"

struct jv_shard_group {
    std::vector<jv_index*> shards;
    std::mutex mu;  // one sharded batch at a time per group (the buffers below are reused)
    struct PerShard {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        float* d_queries = nullptr;
        int32_t *d_nodes = nullptr, *d_docs = nullptr, *d_count = nullptr, *d_stats = nullptr, *d_flags = nullptr;
        float* d_scores = nullptr;
        int32_t* d_pairs = nullptr;  // [nq][k] (doc, score bits)
        uint64_t* d_accept = nullptr;  // the batch's doc filter (GLOBAL doc ids), copied to this shard's device
        size_t cap_q = 0, cap_out = 0, cap_nq = 0, cap_acc = 0;
    };
    std::vector<PerShard> per;
    // on shards[0]'s device: gathered (doc, score) pairs [G][nq][k] + merged output + per-shard stats / flags
    int32_t *g_docs = nullptr, *m_docs = nullptr;   // g_docs holds the pairs
    float *g_scores = nullptr, *m_scores = nullptr; // (g_scores unused: kept null)
    int32_t* g_stats = nullptr;  // [G][nq][4]
    int32_t* g_flags = nullptr;  // [G][nq]
    size_t cap_g = 0, cap_m = 0, cap_s = 0;
    hipStream_t merge_stream = nullptr;
    // optional: the per-shard (doc, score) lists travel by ONE RCCL all-gather over xGMI instead of one peer copy per shard
    // (jv_shard_group_set_option "gather" = 1; the north star's collective, owned by the library so that a one-JVM-per-node
    // host gets it without a process per GPU).  librccl is opened at run time: no link-time dependency.
    int gather_mode = 0;               // 0 peer copies, 1 RCCL all-gather
    std::vector<void*> comms;          // ncclComm_t per shard
    std::vector<int32_t*> d_gathered;  // per shard: [G][nq][k] pairs (every rank receives everything)
    size_t cap_gathered = 0;
};

namespace {
// the handful of RCCL entry points the shard group uses, resolved from librccl.so at first use
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.CommInitAll = (int (*)(void**, int, const int*))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
        api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
        api.GroupStart = (int (*)())dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (int (*)())dlsym(api.lib, "ncclGroupEnd");
        api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
        api.ok = api.CommInitAll && api.CommDestroy && api.AllGather && api.GroupStart && api.GroupEnd;
    });
    return api;
}
const int kNcclInt32 = 2;  // rccl.h: ncclInt32
}  // namespace

namespace {
template <typename T>
int regrow(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return JV_OK;
    if (*p) HIPCHK(jv_free(*p));
    *p = nullptr;
    *cap = 0;
    HIPCHK(hipMalloc((void**)p, (need ? need : 1) * sizeof(T)));
    *cap = need;
    return JV_OK;
}
}  // namespace

extern "C" {

int jv_shard_group_create(jv_index* const* shards, int32_t num_shards, jv_shard_group** out) {
    if (!shards || !out || num_shards <= 0) return fail(JV_EINVAL, "shards/out is NULL or num_shards <= 0");
    *out = nullptr;
    for (int g = 0; g < num_shards; g++) {
        if (!shards[g]) return fail(JV_EINVAL, "shard %d is NULL", g);
        if (shards[g]->dev.d != shards[0]->dev.d) return fail(JV_EINVAL, "shard %d has dimension %d, shard 0 has %d", g, shards[g]->dev.d, shards[0]->dev.d);
    }
    jv_shard_group* grp = new jv_shard_group();
    grp->shards.assign(shards, shards + num_shards);
    grp->per.resize((size_t)num_shards);
    for (int g = 0; g < num_shards; g++) {
        hipError_t e = hipSetDevice(shards[g]->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&grp->per[(size_t)g].stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&grp->per[(size_t)g].done, hipEventDisableTiming);
        if (e == hipSuccess && g > 0 && shards[g]->device != shards[0]->device) {
            // peer access both ways where the fabric offers it (xGMI); the gather falls back to staged copies otherwise
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, shards[g]->device, shards[0]->device) == hipSuccess && can) {
                hipError_t pe = hipDeviceEnablePeerAccess(shards[0]->device, 0);  // (current device = shard g's)
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
            can = 0;
            if (hipDeviceCanAccessPeer(&can, shards[0]->device, shards[g]->device) == hipSuccess && can &&
                hipSetDevice(shards[0]->device) == hipSuccess) {
                hipError_t pe = hipDeviceEnablePeerAccess(shards[g]->device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                e = hipSetDevice(shards[g]->device);
            }
        }
        if (e != hipSuccess) {
            jv_shard_group_destroy(grp);
            return fail(JV_EDEVICE, "shard group: %s", hipGetErrorString(e));
        }
    }
    hipError_t e = hipSetDevice(shards[0]->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&grp->merge_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        jv_shard_group_destroy(grp);
        return fail(JV_EDEVICE, "shard group: %s", hipGetErrorString(e));
    }
    *out = grp;
    return JV_OK;
}

void jv_shard_group_destroy(jv_shard_group* grp) {
    if (!grp) return;
    for (size_t g = 0; g < grp->per.size(); g++) {
        hipSetDevice(grp->shards[g]->device);
        jv_shard_group::PerShard& p = grp->per[g];
        if (p.stream) hipStreamSynchronize(p.stream);
        jv_free(p.d_queries);
        jv_free(p.d_nodes);
        jv_free(p.d_docs);
        jv_free(p.d_scores);
        jv_free(p.d_pairs);
        jv_free(p.d_accept);
        jv_free(p.d_count);
        jv_free(p.d_stats);
        jv_free(p.d_flags);
        if (p.done) hipEventDestroy(p.done);
        if (p.stream) hipStreamDestroy(p.stream);
    }
    for (size_t g = 0; g < grp->d_gathered.size(); g++) {
        hipSetDevice(grp->shards[g]->device);
        jv_free(grp->d_gathered[g]);
    }
    for (void* cm : grp->comms)
        if (cm && rccl_api().ok) rccl_api().CommDestroy(cm);
    if (!grp->shards.empty()) hipSetDevice(grp->shards[0]->device);
    jv_free(grp->g_docs);
    jv_free(grp->g_scores);
    jv_free(grp->m_docs);
    jv_free(grp->m_scores);
    jv_free(grp->g_stats);
    jv_free(grp->g_flags);
    if (grp->merge_stream) hipStreamDestroy(grp->merge_stream);
    delete grp;
}

int jv_shard_group_set_option(jv_shard_group* grp, const char* name, int64_t value) {
    if (!grp || !name) return fail(JV_EINVAL, "group/name is NULL");
    if (strcmp(name, "gather") != 0) return fail(JV_EINVAL, "unknown shard group option '%s'", name);
    std::lock_guard<std::mutex> lk(grp->mu);
    if (value == 0) {
        grp->gather_mode = 0;
        return JV_OK;
    }
    if (value != 1) return fail(JV_EINVAL, "gather must be 0 (peer copies) or 1 (RCCL all-gather)");
    const int G = (int)grp->shards.size();
    for (int a = 0; a < G; a++)
        for (int b2 = a + 1; b2 < G; b2++)
            if (grp->shards[(size_t)a]->device == grp->shards[(size_t)b2]->device)
                return fail(JV_EUNSUPPORTED, "RCCL gather needs every shard on its own device (shards %d and %d share device %d)", a, b2,
                            grp->shards[(size_t)a]->device);
    RcclApi& api = rccl_api();
    if (!api.ok) return fail(JV_EUNSUPPORTED, "librccl.so could not be opened: RCCL gather unavailable");
    if (grp->comms.empty()) {
        std::vector<int> devs((size_t)G);
        for (int g = 0; g < G; g++) devs[(size_t)g] = grp->shards[(size_t)g]->device;
        grp->comms.assign((size_t)G, nullptr);
        const int r = api.CommInitAll(grp->comms.data(), G, devs.data());
        if (r != 0) {
            grp->comms.clear();
            return fail(JV_EDEVICE, "ncclCommInitAll failed: %s", api.GetErrorString ? api.GetErrorString(r) : "?");
        }
        grp->d_gathered.assign((size_t)G, nullptr);
    }
    grp->gather_mode = 1;
    return JV_OK;
}

int jv_search_sharded_batch(jv_shard_group* grp, const float* queries, int32_t nq, int32_t topK, int32_t rerankK,
                            float threshold, float rerankFloor, int32_t* out_docs, float* out_scores,
                            int32_t* out_count, int32_t* out_stats) {
    jv_search_params p{};
    p.struct_size = sizeof(p);
    p.topK = topK;
    p.rerankK = rerankK;
    p.threshold = threshold;
    p.rerankFloor = rerankFloor;
    return jv_search_sharded_batch_ex(grp, queries, nq, &p, out_docs, out_scores, out_count, out_stats, nullptr, nullptr);
}

int jv_search_sharded_batch_ex(jv_shard_group* grp, const float* queries, int32_t nq, const jv_search_params* prm, int32_t* out_docs,
                               float* out_scores, int32_t* out_count, int32_t* out_stats, int32_t* out_status, int32_t* out_flags) {
    if (!grp) return fail(JV_EINVAL, "group is NULL");
    if (!prm || prm->struct_size != sizeof(jv_search_params)) return fail(JV_EINVAL, "jv_search_params is NULL or has the wrong struct_size");
    const int32_t topK = prm->topK, rerankK = prm->rerankK;
    const float threshold = prm->threshold, rerankFloor = prm->rerankFloor;
    const int G = (int)grp->shards.size();
    int rc = check_common(grp->shards[0], queries, nq, topK, rerankK, threshold);
    if (rc != JV_OK) return rc;
    if (prm->accept_doc_words && prm->accept_num_docs < 0) return fail(JV_EINVAL, "accept_num_docs < 0");
    if (prm->visit_limit < 0) return fail(JV_EINVAL, "visit_limit < 0");
    if (out_status && nq > 0) memset(out_status, 0, sizeof(int32_t) * (size_t)nq);
    if (out_flags && nq > 0) memset(out_flags, 0, sizeof(int32_t) * (size_t)nq);
    if (nq == 0) return JV_OK;
    if (!out_docs || !out_scores) return fail(JV_EINVAL, "output pointer is NULL");
    if ((int64_t)G * topK * 8 > 64 * 1024) return fail(JV_EUNSUPPORTED, "shards * topK too large for one merge tile");
    const size_t outn = (size_t)nq * (size_t)topK;
    if (topK == 0) {
        if (out_count) memset(out_count, 0, sizeof(int32_t) * (size_t)nq);
        if (out_stats) memset(out_stats, 0, sizeof(int32_t) * 4 * (size_t)nq);
        return JV_OK;
    }
    std::lock_guard<std::mutex> lk(grp->mu);
    const int d = grp->shards[0]->dev.d;
    const int dev0 = grp->shards[0]->device;
    HIPCHK(hipSetDevice(dev0));
    if ((rc = regrow(&grp->g_docs, &grp->cap_g, 2 * outn * (size_t)G)) != JV_OK) return rc;
    if (!grp->g_stats || grp->cap_s < (size_t)G * (size_t)nq) {
        if (grp->g_stats) HIPCHK(jv_free(grp->g_stats));
        grp->g_stats = nullptr;
        HIPCHK(hipMalloc((void**)&grp->g_stats, (size_t)G * (size_t)nq * 4 * sizeof(int32_t)));
        if (grp->g_flags) HIPCHK(jv_free(grp->g_flags));
        grp->g_flags = nullptr;
        HIPCHK(hipMalloc((void**)&grp->g_flags, (size_t)G * (size_t)nq * sizeof(int32_t)));
        grp->cap_s = (size_t)G * (size_t)nq;
    }
    if (!grp->m_docs || grp->cap_m < outn) {
        if (grp->m_docs) HIPCHK(jv_free(grp->m_docs));
        if (grp->m_scores) HIPCHK(jv_free(grp->m_scores));
        grp->m_docs = nullptr;
        grp->m_scores = nullptr;
        HIPCHK(hipMalloc((void**)&grp->m_docs, outn * sizeof(int32_t)));
        HIPCHK(hipMalloc((void**)&grp->m_scores, outn * sizeof(float)));
        grp->cap_m = outn;
    }
    // whatever happens below: nothing of this call may still be in flight on the group's buffers when it returns, and the
    // calling thread gets its device back
    struct Drain {
        jv_shard_group* grp;
        int dev_before;
        ~Drain() {
            for (size_t g = 0; g < grp->per.size(); g++) {
                if (hipSetDevice(grp->shards[g]->device) == hipSuccess && grp->per[g].stream) hipStreamSynchronize(grp->per[g].stream);
            }
            if (!grp->shards.empty() && hipSetDevice(grp->shards[0]->device) == hipSuccess && grp->merge_stream) hipStreamSynchronize(grp->merge_stream);
            if (dev_before >= 0) hipSetDevice(dev_before);
        }
    } drain{grp, -1};
    (void)hipGetDevice(&drain.dev_before);
    const size_t acc_words = prm->accept_doc_words ? ((size_t)prm->accept_num_docs + 63) / 64 : 0;
    // 1. every shard searches the whole batch on its own device and stream
    for (int g = 0; g < G; g++) {
        jv_index* ix = grp->shards[(size_t)g];
        jv_shard_group::PerShard& p = grp->per[(size_t)g];
        HIPCHK(hipSetDevice(ix->device));
        if ((rc = regrow(&p.d_queries, &p.cap_q, (size_t)nq * d)) != JV_OK) return rc;
        if (p.cap_out < outn) {
            jv_free(p.d_nodes);
            jv_free(p.d_docs);
            jv_free(p.d_scores);
            jv_free(p.d_pairs);
            p.d_nodes = p.d_docs = p.d_pairs = nullptr;
            p.d_scores = nullptr;
            p.cap_out = 0;
            HIPCHK(hipMalloc((void**)&p.d_nodes, outn * 4));
            HIPCHK(hipMalloc((void**)&p.d_docs, outn * 4));
            HIPCHK(hipMalloc((void**)&p.d_scores, outn * 4));
            HIPCHK(hipMalloc((void**)&p.d_pairs, outn * 8));
            p.cap_out = outn;
        }
        if (p.cap_nq < (size_t)nq) {
            jv_free(p.d_count);
            jv_free(p.d_stats);
            jv_free(p.d_flags);
            p.d_count = p.d_stats = p.d_flags = nullptr;
            p.cap_nq = 0;
            HIPCHK(hipMalloc((void**)&p.d_count, (size_t)nq * 4));
            HIPCHK(hipMalloc((void**)&p.d_stats, (size_t)nq * 16));
            HIPCHK(hipMalloc((void**)&p.d_flags, (size_t)nq * 4));
            p.cap_nq = (size_t)nq;
        }
        HIPCHK(hipMemcpyAsync(p.d_queries, queries, (size_t)nq * d * sizeof(float), hipMemcpyHostToDevice, p.stream));
        const uint64_t* d_acc = nullptr;
        if (prm->accept_doc_words) {
            // the filter is over GLOBAL doc ids (every shard's ord2doc maps into that space), so every shard gets the same bits
            if ((rc = regrow(&p.d_accept, &p.cap_acc, acc_words ? acc_words : 1)) != JV_OK) return rc;
            if (acc_words) HIPCHK(hipMemcpyAsync(p.d_accept, prm->accept_doc_words, acc_words * 8, hipMemcpyHostToDevice, p.stream));
            d_acc = p.d_accept;
        }
        rc = search_batch_device_impl(ix, p.d_queries, nq, topK, rerankK, threshold, rerankFloor, d_acc, d_acc ? prm->accept_num_docs : 0,
                                      prm->visit_limit, p.d_nodes, p.d_docs, p.d_scores, p.d_count, p.d_stats, p.d_flags, (void*)p.stream);
        if (rc != JV_OK) return rc;
        // 2. gather: this shard's [nq][k] (doc, score) pairs go to slot g of the [G][nq][k] buffer on device 0 in ONE
        //    peer copy over xGMI (+ the counters and flags)
        HIPCHK(jvk_launch_pack_pairs(p.d_docs, p.d_scores, p.d_pairs, (long long)outn, p.stream));
        if (grp->gather_mode == 0)
            HIPCHK(hipMemcpyPeerAsync(grp->g_docs + 2 * (size_t)g * outn, dev0, p.d_pairs, ix->device, outn * 8, p.stream));
        HIPCHK(hipMemcpyPeerAsync(grp->g_stats + (size_t)g * (size_t)nq * 4, dev0, p.d_stats, ix->device, (size_t)nq * 16, p.stream));
        HIPCHK(hipMemcpyPeerAsync(grp->g_flags + (size_t)g * (size_t)nq, dev0, p.d_flags, ix->device, (size_t)nq * 4, p.stream));
        HIPCHK(hipEventRecord(p.done, p.stream));
    }
    const int32_t* merge_src = grp->g_docs;
    if (grp->gather_mode == 1) {
        // 2b. ONE all-gather of the pair buffers: every device receives [G][nq][k] pairs (device 0's copy is merged)
        RcclApi& api = rccl_api();
        if (grp->cap_gathered < 2 * outn * (size_t)G) {
            for (int g = 0; g < G; g++) {
                HIPCHK(hipSetDevice(grp->shards[(size_t)g]->device));
                jv_free(grp->d_gathered[(size_t)g]);
                grp->d_gathered[(size_t)g] = nullptr;
                HIPCHK(hipMalloc((void**)&grp->d_gathered[(size_t)g], 2 * outn * (size_t)G * sizeof(int32_t)));
            }
            grp->cap_gathered = 2 * outn * (size_t)G;
        }
        int r = api.GroupStart();
        hipError_t he = hipSuccess;  // (no early return between GroupStart and GroupEnd: the group would stay open)
        for (int g = 0; g < G && r == 0 && he == hipSuccess; g++) {
            he = hipSetDevice(grp->shards[(size_t)g]->device);
            if (he != hipSuccess) break;
            r = api.AllGather(grp->per[(size_t)g].d_pairs, grp->d_gathered[(size_t)g], 2 * outn, kNcclInt32, grp->comms[(size_t)g], grp->per[(size_t)g].stream);
        }
        const int r2 = api.GroupEnd();
        if (he != hipSuccess) return fail(JV_EDEVICE, "hipSetDevice failed inside the all-gather group: %s", hipGetErrorString(he));
        if (r != 0 || r2 != 0) return fail(JV_EDEVICE, "ncclAllGather failed: %s", api.GetErrorString ? api.GetErrorString(r != 0 ? r : r2) : "?");
        for (int g = 0; g < G; g++) {
            HIPCHK(hipSetDevice(grp->shards[(size_t)g]->device));
            HIPCHK(hipEventRecord(grp->per[(size_t)g].done, grp->per[(size_t)g].stream));
        }
        merge_src = grp->d_gathered[0];
    }
    // 3. merge on device 0 once every shard's lists have arrived
    HIPCHK(hipSetDevice(dev0));
    for (int g = 0; g < G; g++) HIPCHK(hipStreamWaitEvent(grp->merge_stream, grp->per[(size_t)g].done, 0));
    HIPCHK(jvk_launch_merge_topk_strided(merge_src, grp->g_scores, nq, G, topK, grp->m_docs, grp->m_scores, grp->merge_stream));
    std::vector<int32_t> h_stats((size_t)G * (size_t)nq * 4), h_flags((size_t)G * (size_t)nq);
    HIPCHK(hipMemcpyAsync(out_docs, grp->m_docs, outn * 4, hipMemcpyDeviceToHost, grp->merge_stream));
    HIPCHK(hipMemcpyAsync(out_scores, grp->m_scores, outn * 4, hipMemcpyDeviceToHost, grp->merge_stream));
    HIPCHK(hipMemcpyAsync(h_stats.data(), grp->g_stats, h_stats.size() * 4, hipMemcpyDeviceToHost, grp->merge_stream));
    HIPCHK(hipMemcpyAsync(h_flags.data(), grp->g_flags, h_flags.size() * 4, hipMemcpyDeviceToHost, grp->merge_stream));
    HIPCHK(hipStreamSynchronize(grp->merge_stream));
    int failed = 0;
    for (int i = 0; i < nq; i++) {
        int32_t st[4] = {0, 0, 0, 0};
        uint32_t fl = 0;
        bool bad = false;
        for (int g = 0; g < G; g++) {
            for (int j = 0; j < 4; j++) st[j] += h_stats[((size_t)g * (size_t)nq + (size_t)i) * 4 + (size_t)j];
            const uint32_t f = (uint32_t)h_flags[(size_t)g * (size_t)nq + (size_t)i];
            if (f & (JV_FLAG_FAILED | JV_FLAG_OVERFLOW)) bad = true, failed++;
            fl |= f & (JV_FLAG_BIG | JV_FLAG_EARLY);  // (EARLY: at least one shard's search hit the visit limit and returned nothing)
        }
        if (out_status && bad) out_status[i] = JV_ENOMEM;
        if (out_flags) out_flags[i] = (int32_t)fl;
        if (out_stats) memcpy(out_stats + (size_t)i * 4, st, sizeof(st));
        if (out_count) {
            int c = 0;
            while (c < topK && out_docs[(size_t)i * topK + (size_t)c] >= 0) c++;
            out_count[i] = c;
        }
    }
    if (failed) return fail(JV_ENOMEM, "%d (query, shard) searches overflowed the HBM scratch; raise option big_cand_cap", failed);
    return JV_OK;
}

}  // extern "C"

