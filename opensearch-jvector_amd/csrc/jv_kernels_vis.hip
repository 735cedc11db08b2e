// jv_kernels_vis.hip — jvector's visitedCount for a whole batch, taken AFTER the search launch (round 5).  gfx950 / CDNA4.
//
// GraphSearcher counts every node it scores once (J/JVectorReader.java:202-207 reports the count through
// KnnCollector.incVisitedCount): the distinct neighbours of the expanded nodes, entry point excluded.  The
// pool kernels keep no visited set while they search — they rebuild the count from the expansion log.  Inside the several-waves
// search kernel that pass is two waves behind one memory round trip per group of rows, on a hash set that needs two classes
// (two walks over the log) because the query's LDS share is 19 KB: 11-16 % of the time a query holds its slot of the CU.  Here
// the same count is a throughput job, on copies of the batch's logs (JvSearchArgs.vis_*, jv_pqw_body.h): one workgroup of
// eight waves per query and a 64 KB set (one class for logs of up to ~3 800 expansions), two workgroups per CU.
//   jv_visited_fast_kernel   R = 16 / 32 / 64, 16-byte aligned rows, logs that fit one class (below: steps, packed tails)
//   jv_visited_kernel        any shape, any number of classes: what the fast kernel leaves at -1, or everything
// Both apply Lucene's visit limit to the rows they count (vis_publish).  The count does not depend on hash, probe order or
// classes; tests/test_gpu_pqw.py::test_visited_counts_taken_after_the_launch and tools/fuzz_parity.py hold every way through
// against the oracle's counters.
#include "jv_dev_common.h"

#define JV_VIS_WAVES 8
typedef int vis_i32x16 __attribute__((ext_vector_type(16)));

// one query's count goes out (called by a whole wave): word 0 of its stats row, and — with a visit limit — the row turns into
// an early-terminated one when visited + expanded reaches the limit (what jv_pqw_body.h does when it counts inside the search kernel)
__device__ __forceinline__ void vis_publish(const JvVisArgs& a, const int qi, const int v, const int lane) {
    int32_t* st = a.out_stats + (size_t)qi * 4;
    bool early = false;
    // (a row the search kernel left flagged for a later rung — e.g. why = 6, the rerankFloor tie, which returns after its log was
    //  copied and before its stats row is written — has no valid st[2]: the limit is the later rung's business, and its OVERFLOW flag
    //  must not be replaced by EARLY with count 0.  ADVICE r5, low)
    if (a.visit_limit > 0 && ((uint32_t)__builtin_amdgcn_readfirstlane(a.out_flags[qi]) & JV_FLAG_OVERFLOW) == 0u)
        early = v + __builtin_amdgcn_readfirstlane(st[2]) >= a.visit_limit;
    if (lane == 0) st[0] = v;
    if (early) {
        if (lane == 0) {
            st[1] = 0;
            a.out_flags[qi] = (int32_t)JV_FLAG_EARLY;
            a.out_count[qi] = 0;
        }
        for (int i = lane; i < a.topK; i += JV_WAVE) {
            a.out_nodes[(size_t)qi * a.topK + i] = -1;
            if (a.out_docs) a.out_docs[(size_t)qi * a.topK + i] = -1;
            a.out_scores[(size_t)qi * a.topK + i] = 0.0f;
        }
    }
}

__global__ __launch_bounds__(JV_WAVE * JV_VIS_WAVES) void jv_visited_kernel(const JvVisArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int W = JV_VIS_WAVES;
    uint32_t* vh = (uint32_t*)smem;
    int* ctrl = (int*)(smem + (size_t)a.slots * 4);  // [0] again, [1] query, [4 + w] per-wave counts
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int R = a.R;
    const int vslots = a.slots;
    const uint32_t vmask = (uint32_t)vslots - 1u;
    const int vshift = 32 - (31 - __clz(vslots));
    const int vlimit = (vslots / 16) * 13;
    const int vlimit_w = vlimit / W;  // fresh entries one wave may add per class
    constexpr int VB = 8;             // neighbour ids per lane and group
    // adjacency rows as 16-byte pieces where the shape allows it (R / 4 lanes per row), else one id per lane and load
    const int lpr4 = max(1, R >> 2);
    const bool vec = (R & 3) == 0 && (64 % (2 * (JV_WAVE / lpr4))) == 0 && (((uintptr_t)a.adj) & 15) == 0;
    const int lpr = vec ? lpr4 : R;
    const int rpl = JV_WAVE / lpr;                   // rows per load instruction
    const int G = vec ? rpl * 2 : rpl * VB;          // log entries per group (divides 64)
    const int lrow = lane / lpr, lcol = (lane % lpr) * (vec ? 4 : 1);
    const bool lane_ok = lane < rpl * lpr;
    // A workgroup takes a contiguous range of the batch; its threads look at one query each for what is still to count — a log in the
    // arena (else: counted by the search kernel, or a row a later rung redoes) whose stats row still says -1 (else: counted by
    // jv_visited_fast_kernel) — and the workgroup walks that list.  (One query at a time behind two dependent loads each cost
    // 1.2 ms per 262 144 queries with nothing to do.)
    int* const todo_q = (int*)(smem + (size_t)a.slots * 4 + 64);  // [JV_WAVE * W] (the fast kernel's list area)
    const int per = (a.nq + (int)gridDim.x - 1) / (int)gridDim.x;
    const int qb0 = (int)blockIdx.x * per, qb1 = min(a.nq, qb0 + per);
    for (int qc = qb0; qc < qb1; qc += JV_WAVE * W) {
    __syncthreads();
    if (threadIdx.x == 0) ctrl[1] = 0;
    __syncthreads();
    {
        const int qi = qc + (int)threadIdx.x;
        if (qi < qb1 && a.vis_n[qi] > 0 && __hip_atomic_load(&a.out_stats[(size_t)qi * 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0)
            todo_q[atomicAdd(&ctrl[1], 1)] = qi;
    }
    __syncthreads();
    const int ntodo = __builtin_amdgcn_readfirstlane(ctrl[1]);
    for (int ti = 0; ti < ntodo; ti++) {
        const int qi = __builtin_amdgcn_readfirstlane(todo_q[ti]);
        const int nexp = a.vis_n[qi];
        const int32_t* explog = a.arena + (size_t)a.vis_off[qi] * 4;
        int parts = 1;
        while ((long long)nexp * 7 > (long long)vlimit * parts * 2) parts <<= 1;
        int visited = 0;
        bool again = true;
        while (again) {
            again = false;
            visited = 0;
            int plog = 0;
            while ((1 << plog) < parts) plog++;
            const int pshift = vshift - plog;            // the class = the hash bits right below the slot index
            const uint32_t pmask = (uint32_t)parts - 1u;
            for (int p = 0; p < parts && !again; p++) {
                __syncthreads();
                for (int i = threadIdx.x; i < vslots; i += JV_WAVE * W) vh[i] = HASH_EMPTY;
                if (threadIdx.x == 0) ctrl[0] = 0;
                __syncthreads();
                if (threadIdx.x == 0 && ((((uint32_t)a.entry * 0x9E3779B1u) >> pshift) & pmask) == (uint32_t)p)
                    visited_insert_lds(vh, vmask, vshift, (uint32_t)a.entry);
                __syncthreads();
                int cntl = 0;  // fresh entries, counted per lane
                bool over = false;
                constexpr int NPF = 4;  // groups of rows in flight per wave, each in registers of its own
                for (int blk0 = 0; blk0 < nexp && !over; blk0 += 1024) {
                    const int nblk = min(1024, nexp - blk0);
                    vis_i32x16 logv;
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        logv[g] = 0;
                        if (g * 64 < nblk) logv[g] = explog[blk0 + min(g * 64 + lane, nblk - 1)];
                    }
                    const int e_last = (nblk - 1) / G * G;  // first entry of the last group
                    auto load_group = [&](int e0, int (&dst)[VB]) {
                        const int e0c = min(e0, e_last);
                        const int cur = logv[__builtin_amdgcn_readfirstlane(e0c >> 6)];
                        if (vec) {
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const int e = min(e0c + h * rpl + lrow, nblk - 1);
                                const int node = __builtin_amdgcn_ds_bpermute((e & 63) << 2, cur);
                                const u32x4 v = __builtin_nontemporal_load((const u32x4*)(a.adj + (size_t)node * R + lcol));
                                dst[4 * h] = (int)v.x, dst[4 * h + 1] = (int)v.y, dst[4 * h + 2] = (int)v.z, dst[4 * h + 3] = (int)v.w;
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const int e = min(e0c + u * rpl + lrow, nblk - 1);
                                const int node = __builtin_amdgcn_ds_bpermute((e & 63) << 2, cur);
                                dst[u] = a.adj[(size_t)node * R + lcol];
                            }
                        }
                    };
                    // one group's ids into the set; false when the table's fill limit would be passed
                    auto probe_group = [&](const int (&q)[VB], int e0) -> bool {
                        uint32_t hh[VB];
                        bool pend[VB];
                        int pl = 0;
                        {
                            const bool ok_a = lane_ok && e0 + lrow < nblk, ok_b = lane_ok && e0 + rpl + lrow < nblk;
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const bool ok = vec ? (u < 4 ? ok_a : ok_b) : (lane_ok && e0 + u * rpl + lrow < nblk);
                                const uint32_t prod = (uint32_t)q[u] * 0x9E3779B1u;   // one product: slot index on top, class below
                                hh[u] = prod >> vshift;
                                pend[u] = ok && q[u] >= 0 && ((prod >> pshift) & pmask) == (uint32_t)p;   // (rows are padded with -1)
                                pl += pend[u] ? 1 : 0;
                            }
                        }
                        if (jv_wave_sum_int(cntl + pl) > vlimit_w) return false;
                        uint32_t live = 0;
                        {
                            uint32_t oldv[VB];
#pragma unroll
                            for (int u = 0; u < VB; u++) oldv[u] = pend[u] ? atomicCAS(&vh[hh[u]], HASH_EMPTY, (uint32_t)q[u]) : 0u;
#pragma unroll
                            for (int u = 0; u < VB; u++) {
                                const bool fresh = pend[u] && oldv[u] == HASH_EMPTY;
                                cntl += fresh ? 1 : 0;
                                pend[u] = pend[u] && !fresh && oldv[u] != (uint32_t)q[u];
                                hh[u] = (hh[u] + 1) & vmask;
                                if (__any(pend[u])) live |= 1u << u;
                            }
                        }
                        while (live) {
                            uint32_t oldv[VB];
#pragma unroll
                            for (int u = 0; u < VB; u++)
                                if (live & (1u << u)) oldv[u] = pend[u] ? atomicCAS(&vh[hh[u]], HASH_EMPTY, (uint32_t)q[u]) : 0u;
#pragma unroll
                            for (int u = 0; u < VB; u++)
                                if (live & (1u << u)) {
                                    const bool fresh = pend[u] && oldv[u] == HASH_EMPTY;
                                    cntl += fresh ? 1 : 0;
                                    pend[u] = pend[u] && !fresh && oldv[u] != (uint32_t)q[u];
                                    hh[u] = (hh[u] + 1) & vmask;
                                    if (!__any(pend[u])) live &= ~(1u << u);
                                }
                        }
                        return true;
                    };
                    int qq[NPF][VB];
                    const int step = G * W;
#pragma unroll
                    for (int k = 0; k < NPF; k++) load_group(wv * G + k * step, qq[k]);
                    for (int e0 = wv * G; e0 < nblk && !over; e0 += NPF * step) {
#pragma unroll
                        for (int k = 0; k < NPF; k++) {
                            const int ek = e0 + k * step;
                            if (ek < nblk && !over) {
                                if (!probe_group(qq[k], ek)) over = true;
                                else load_group(ek + NPF * step, qq[k]);
                            }
                        }
                    }
                }
                const int cnt = jv_wave_sum_int(cntl);
                if (over && lane == 0) ctrl[0] = 1;
                visited += cnt;
                __syncthreads();
                again = __builtin_amdgcn_readfirstlane(ctrl[0]) != 0;
            }
            if (again) parts <<= 1;  // (a class holds at most vlimit ids: with enough classes every log fits)
        }
        if (lane == 0) ctrl[4 + wv] = visited;
        __syncthreads();
        if (wv == 0) {
            int v = 0;
#pragma unroll
            for (int w2 = 0; w2 < W; w2++) v += __builtin_amdgcn_readfirstlane(ctrl[4 + w2]);
            vis_publish(a, qi, v, lane);
        }
        __syncthreads();
    }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The common shapes (R = 16 / 32 / 64, 16-byte aligned rows, logs that fit ONE hash class): a wave takes 64 log entries per step
// (R = 64: 32) — one coalesced load of the log chunk, E / RPL 16-byte row loads, IDS = E * R / 64 neighbour ids per lane — and
// sends all of them down the probe chains together, so a step costs about as many LDS round trips as its longest chain instead
// of one chain per group of 16 rows.  The steps of a workgroup's queries form ONE stream per wave: rows of the step after next
// and the log chunk behind that are requested before the current step probes, across query boundaries (a query is ~2.4 steps
// per wave at rerankK 1 200: without that every query would pay two exposed memory round trips).  Two barriers per query: after
// the set is cleared and before the count is written.  The entry point is never pending (jvector does not count it).
// A chain longer than JV_VIS_CHAIN_MAX slots (a set close to full) gives the query back: word 0 of its stats row stays negative and
// jv_visited_kernel above — any shape, any number of classes — counts it.
#define JV_VIS_CHAIN_MAX 96
#define JV_VIS_LIST 256  /* ids a wave parks in LDS between the full-width rounds and the packed walk */
#define JV_VIS_META 512  /* queries of a workgroup's range whose log length / offset sit in LDS */
struct VisItem { int qi, s, nexp; uint32_t off; };
#ifdef JV_STAMPS
#define VIS_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = clock64(); vst[i] += t_ - vst_last; vst_last = t_; __builtin_amdgcn_sched_barrier(0); }
#define VIS_COUNT(i, v) { vst[i] += (unsigned long long)(v); }
#else
#define VIS_STAMP(i) {}
#define VIS_COUNT(i, v) {}
#endif

template <int LPR>  // lanes per adjacency row: R = 4 * LPR
__global__ __launch_bounds__(JV_WAVE * JV_VIS_WAVES, 4) void jv_visited_fast_kernel(const JvVisArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int W = JV_VIS_WAVES;
    constexpr int R = LPR * 4, RPL = JV_WAVE / LPR, E = (2048 / R < 64 ? 2048 / R : 64), NLD = E / RPL, IDS = NLD * 4;
    static_assert(IDS <= 32 && NLD >= 1, "ids per lane and step");
    constexpr int VCH = 4;  // compare-and-swaps in flight per lane in the full-width rounds
    uint32_t* vh = (uint32_t*)smem;
    int* ctrl = (int*)(smem + (size_t)a.slots * 4);  // [0] a chain ran too long, [4 + w] per-wave counts
    int32_t* lst = (int32_t*)(smem + (size_t)a.slots * 4 + 64) + (threadIdx.x >> 6) * JV_VIS_LIST;       // this wave's parked ids
    int32_t* mn = (int32_t*)(smem + (size_t)a.slots * 4 + 64 + W * JV_VIS_LIST * 4);                     // [JV_VIS_META] log length, 0 = not this kernel's
    uint32_t* mo = (uint32_t*)(mn + JV_VIS_META);                                                          // [JV_VIS_META] first unit of the log
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int vslots = a.slots;
    const uint32_t vmask = (uint32_t)vslots - 1u;
    const int vshift = 32 - (31 - __clz(vslots));
    const long long vlimit2 = (long long)((vslots / 16) * 13) * 2;
    const int lrow = lane / LPR, lcol = (lane % LPR) * 4;
    int cntl = 0;       // fresh ids, counted per lane
    bool over = false;  // a chain ran too long
#ifdef JV_STAMPS
    unsigned long long vst[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned long long vst_last = clock64();
#endif
    // a workgroup takes a contiguous range of the batch, JV_VIS_META queries at a time: their log lengths and offsets go to LDS
    // first (a step that had to find its query through dependent loads from global memory spent a third of its time there)
    const int per = (a.nq + (int)gridDim.x - 1) / (int)gridDim.x;
    const int qb0 = (int)blockIdx.x * per, qb1 = min(a.nq, qb0 + per);
    for (int qbase = qb0; qbase < qb1; qbase += JV_VIS_META) {
        const int cnt = min(JV_VIS_META, qb1 - qbase);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += JV_WAVE * W) {
            const int n = a.vis_n[qbase + k];
            mn[k] = (n > 0 && (long long)n * 7 <= vlimit2) ? n : 0;  // (a log in the arena that fits ONE class: the generic kernel's `parts` rule)
            mo[k] = a.vis_off[qbase + k];
        }
        __syncthreads();
        auto seek = [&](int k) -> int {
            while (k < cnt && __builtin_amdgcn_readfirstlane(mn[k]) == 0) k++;
            return k;
        };
        auto settle = [&](VisItem& it) {  // the first step at or behind (it.qi, it.s) that exists for this wave
            while (it.qi < cnt) {
                it.nexp = __builtin_amdgcn_readfirstlane(mn[it.qi]);
                if (it.s * E < it.nexp) {
                    it.off = __builtin_amdgcn_readfirstlane(mo[it.qi]);
                    return;
                }
                it.qi++;
                it.s = wv;
            }
        };
        auto next_item = [&](const VisItem& it) -> VisItem {
            VisItem n = it;
            if (n.qi < cnt) {
                n.s += W;
                settle(n);
            }
            return n;
        };
        auto load_log = [&](const VisItem& it) -> int {
            if (it.qi >= cnt) return 0;
            const int32_t* lg = a.arena + (size_t)it.off * 4;
            return lg[min(it.s * E + (lane & (E - 1)), it.nexp - 1)];
        };
        auto load_rows = [&](const VisItem& it, int cur, int (&dst)[IDS]) {
            if (it.qi >= cnt) return;
            // (lane-derived values from an opaque copy of the lane id: hoisted to kernel entry they are eight registers more, spilled,
            //  and every reload is an `s_waitcnt vmcnt(0)` that waits for the row load issued just before it)
            int lo = lane;
            asm volatile("" : "+v"(lo));
            const int lr4 = (lo / LPR) << 2, lc = (lo % LPR) * 4;
#pragma unroll
            for (int h = 0; h < NLD; h++) {
                const int node = __builtin_amdgcn_ds_bpermute(h * RPL * 4 + lr4, cur);  // (entries behind the log's end hold its last node)
                const u32x4 v = __builtin_nontemporal_load((const u32x4*)(a.adj + (size_t)node * R + lc));
                dst[4 * h] = (int)v.x, dst[4 * h + 1] = (int)v.y, dst[4 * h + 2] = (int)v.z, dst[4 * h + 3] = (int)v.w;
            }
        };
        // One step's ids into the set.  A compare-and-swap instruction over 64 random slots keeps the LDS busy for ~13 cycles
        // (two half-waves x ~3.5 lanes on the fullest bank x read-modify-write) whatever comes of it, and an instruction with three
        // live lanes still costs a third of that: after the full-width round(s) — every pending id one slot further, 32
        // instructions — what is still pending (~90 of a step's 2 048 ids) is packed into a list in LDS, each lane's ids behind
        // the ids of the lanes below it, and walked 64 ids at a time, each lane down its own chain.
        auto probe = [&](const VisItem& it, const int (&q)[IDS]) {
            uint32_t pm = 0;  // bit u: id u of this lane is still on its way down a chain
#pragma unroll
            for (int h = 0; h < NLD; h++) {
                const bool ok = it.s * E + h * RPL + lrow < it.nexp;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int id = q[4 * h + j];
                    if (ok && id >= 0 && id != a.entry) pm |= 1u << (4 * h + j);  // (rows are padded with -1; jvector does not count the entry point)
                }
            }
            VIS_STAMP(1)  // the step's rows arrive (the mask needs them)
            int round = 0, npend = 0, mine = 0, first = 0;
            for (;; round++) {
                if (round >= JV_VIS_CHAIN_MAX) {
                    over = true;
                    return;
                }
                uint32_t K = 0x9E3779B1u;
                asm volatile("" : "+s"(K));  // (a product per id hoisted out of the rounds is 32 registers this kernel does not have)
#pragma unroll
                for (int c = 0; c < IDS; c += VCH) {
                    if (!__any((pm & (((1u << VCH) - 1u) << c)) != 0)) continue;
                    uint32_t oldv[VCH];
#pragma unroll
                    for (int u = 0; u < VCH; u++) {
                        oldv[u] = 0;
                        if (pm & (1u << (c + u))) oldv[u] = atomicCAS(&vh[((((uint32_t)q[c + u] * K) >> vshift) + (uint32_t)round) & vmask], HASH_EMPTY, (uint32_t)q[c + u]);
                    }
#pragma unroll
                    for (int u = 0; u < VCH; u++) {
                        if (pm & (1u << (c + u))) {
                            const bool fresh = oldv[u] == HASH_EMPTY;
                            cntl += fresh ? 1 : 0;
                            if (fresh || oldv[u] == (uint32_t)q[c + u]) pm &= ~(1u << (c + u));
                        }
                    }
                }
                // pending ids per lane -> where this lane's ids start in the list
                mine = __popc(pm);
                int incl = mine;
#pragma unroll
                for (int o = 1; o < JV_WAVE; o <<= 1) {
                    const int t = __shfl_up(incl, o, JV_WAVE);
                    if (lane >= o) incl += t;
                }
                npend = __builtin_amdgcn_readlane(incl, JV_WAVE - 1);
                first = incl - mine;
                if (npend <= JV_VIS_LIST) break;
            }
            round++;
            VIS_STAMP(2)  // full-width rounds
            VIS_COUNT(8, round)
            if (npend == 0) {
                VIS_COUNT(7, 1)
                return;
            }
#pragma unroll
            for (int u = 0; u < IDS; u++)
                if (pm & (1u << u)) lst[first + __popc(pm & ((1u << u) - 1u))] = q[u];
            VIS_STAMP(3)  // packing
            VIS_COUNT(9, npend)
            for (int b = 0; b < npend; b += JV_WAVE) {
                bool pend = b + lane < npend;
                const uint32_t id = (uint32_t)lst[min(b + lane, npend - 1)];
                uint32_t slot = (((id * 0x9E3779B1u) >> vshift) + (uint32_t)round) & vmask;
                for (int chain = round; __any(pend); chain++) {
                    if (chain >= JV_VIS_CHAIN_MAX) {
                        over = true;
                        return;
                    }
                    VIS_COUNT(10, 1)
                    if (pend) {
                        const uint32_t old = atomicCAS(&vh[slot], HASH_EMPTY, id);
                        cntl += old == HASH_EMPTY ? 1 : 0;
                        pend = !(old == HASH_EMPTY || old == id);
                        slot = (slot + 1) & vmask;
                    }
                }
            }
            VIS_STAMP(4)  // packed walk
            VIS_COUNT(7, 1)
        };
        // the query whose set is in LDS; every wave opens and closes every query of the range in order, with or without steps of its own
        int p_k = -1;
        bool opened = false;
        auto goto_query = [&](int target) {  // target: the query of this wave's next step, or cnt = no step left
            while (!opened || p_k != target) {
                if (opened) {
                    const int c = jv_wave_sum_int(cntl);
                    if (lane == 0) ctrl[4 + wv] = c;
                    if (over && lane == 0) ctrl[0] = 1;
                    __syncthreads();
                    if (wv == 0) {
                        int v = 0;
#pragma unroll
                        for (int w2 = 0; w2 < W; w2++) v += __builtin_amdgcn_readfirstlane(ctrl[4 + w2]);
                        if (__builtin_amdgcn_readfirstlane(ctrl[0]) == 0) vis_publish(a, qbase + p_k, v, lane);  // (else the row stays at -1: the generic kernel's)
                    }
                    cntl = 0;
                    over = false;
                    opened = false;
                }
                p_k = seek(p_k + 1);
                if (p_k >= cnt) break;
                const u32x4 e4 = {HASH_EMPTY, HASH_EMPTY, HASH_EMPTY, HASH_EMPTY};
                for (int i = threadIdx.x; i < vslots / 4; i += JV_WAVE * W) ((u32x4*)vh)[i] = e4;
                if (threadIdx.x == 0) ctrl[0] = 0;
                __syncthreads();
                opened = true;
            }
        };
        // The steps of the range form ONE stream per wave: the rows of the step after next and the log chunk behind that are
        // requested before the current step probes, across query boundaries.
        VisItem i0, i1, i2;
        i0.qi = 0, i0.s = wv, i0.nexp = 0, i0.off = 0;
        settle(i0);
        i1 = next_item(i0);
        i2 = next_item(i1);
        int idsA[IDS], idsB[IDS];
#pragma unroll
        for (int u = 0; u < IDS; u++) idsA[u] = -1, idsB[u] = -1;
        int curL;
        {
            const int c0 = load_log(i0), c1 = load_log(i1);
            curL = load_log(i2);
            load_rows(i0, c0, idsA);
            load_rows(i1, c1, idsB);
        }
        for (;;) {
            // (two steps per trip: the id registers of a step are refilled for the step after next once it has probed)
            if (i0.qi >= cnt) break;
            VIS_STAMP(5)  // next steps found, their loads issued
            goto_query(i0.qi);
            VIS_STAMP(0)  // close / clear / barriers
            probe(i0, idsA);
            {
                const VisItem i3 = next_item(i2);
                load_rows(i2, curL, idsA);
                curL = load_log(i3);
                i0 = i1, i1 = i2, i2 = i3;
            }
            if (i0.qi >= cnt) break;
            VIS_STAMP(5)
            goto_query(i0.qi);
            VIS_STAMP(0)
            probe(i0, idsB);
            {
                const VisItem i3 = next_item(i2);
                load_rows(i2, curL, idsB);
                curL = load_log(i3);
                i0 = i1, i1 = i2, i2 = i3;
            }
        }
        goto_query(cnt);
        VIS_STAMP(6)  // the tail: closing the queries this wave has no step of
    }
#ifdef JV_STAMPS
    if (a.dbg && lane == 0)
        for (int i = 0; i < 12; i++) atomicAdd(a.dbg + i, vst[i]);
#endif
}

typedef void (*vis_fast_t)(const JvVisArgs);
static vis_fast_t vis_fast_pick(const JvVisArgs* a) {
    if ((((uintptr_t)a->adj) & 15) != 0) return nullptr;
    return a->R == 32 ? jv_visited_fast_kernel<8> : a->R == 16 ? jv_visited_fast_kernel<4> : a->R == 64 ? jv_visited_fast_kernel<16> : nullptr;
}

extern "C" int jvk_visited_lds_bytes(int slots) { return slots * 4 + 64 + JV_VIS_WAVES * JV_VIS_LIST * 4 + JV_VIS_META * 8; }
extern "C" hipError_t jvk_visited_set_max_lds(int bytes) {
    return hipFuncSetAttribute((const void*)jv_visited_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
extern "C" int jvk_visited_blocks_per_cu(int slots) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)jv_visited_kernel, JV_WAVE * JV_VIS_WAVES, (size_t)jvk_visited_lds_bytes(slots)) != hipSuccess) return 1;
    return nb < 1 ? 1 : nb;
}
extern "C" hipError_t jvk_launch_visited(const JvVisArgs* a, int blocks, hipStream_t stream) {
    if (a->nq <= 0) return hipSuccess;
    if (jvk_visited_lds_bytes(a->slots) > 65536) {
        const hipError_t e = jvk_visited_set_max_lds(jvk_visited_lds_bytes(a->slots));
        if (e != hipSuccess) return e;
    }
    // the common shapes first; whatever that leaves pending (or everything, for the other shapes) goes through the generic kernel
    const vis_fast_t fast = vis_fast_pick(a);
    if (fast) {
        if (jvk_visited_lds_bytes(a->slots) > 65536) {
            const hipError_t e = hipFuncSetAttribute((const void*)fast, hipFuncAttributeMaxDynamicSharedMemorySize, jvk_visited_lds_bytes(a->slots));
            if (e != hipSuccess) return e;
        }
        fast<<<blocks, JV_WAVE * JV_VIS_WAVES, jvk_visited_lds_bytes(a->slots), stream>>>(*a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    jv_visited_kernel<<<blocks, JV_WAVE * JV_VIS_WAVES, jvk_visited_lds_bytes(a->slots), stream>>>(*a);
    return hipGetLastError();
}
