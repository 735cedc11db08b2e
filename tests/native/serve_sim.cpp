// serve_sim.cpp — the query servers' ticket protocol under the sanitizers, on host threads (no GPU).
//
// What runs here is the PRODUCT's protocol code, not a model of it:
//   * the caller side   = csrc/jv_serve_host.h (jvsh_take_slot / jvsh_publish / jvsh_wait_done / jvsh_abandon / jvsh_release),
//                         the functions jv_abi.cpp's serve_query calls;
//   * the grid side     = csrc/jv_serve_claim.h (jv_serve_claim / jv_serve_slot_current / jv_serve_leave), compiled for the host
//                         through the shim below: every "workgroup" of the resident grid is a host thread.
// Around them the harness restates the lifecycle of csrc/jv_abi.cpp (server_launch_locked: clear LOCK / EXITED / LAST_CLAIM /
// STOP_SEEN, set ALIVE, start the grid; server_stop_locked: STOP, wait for the grid, clear ALIVE / STOP) and drives it the way
// the reference drives JVectorReader.search (T/index/engine/JVectorConcurrentQueryTests.java:78-138): many threads, one query per
// call — next to pauses (every hipFree / wide launch pauses the servers), grids that idle out and are restarted by the next
// caller, launches that fail (abandoned tickets) and callers with and without a "filter" word.  Every answered call is checked.
// tests/test_serve_protocol.py builds this file with -fsanitize=thread and with -fsanitize=address,undefined and runs both.
//
// Memory-model note: the device's agent-scope read-modify-writes (atomicCAS / atomicMax / atomicAdd) resolve in the L2, the
// coherence point of the GPU; they are modelled here as acq_rel operations (and __builtin_amdgcn_fence as nothing on top).  The
// test therefore checks the protocol's LOGIC and the host code's data races (slot hand-over, ticket order, abandon, stop /
// restart), not the GPU's cache behaviour — that is what the GPU stress tests (tests/test_gpu_pqw.py) are for.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <signal.h>
#include <time.h>
#include <unistd.h>

// ---- host shim for the device code of jv_serve_claim.h ----
#define __device__
#define __forceinline__ inline
#define __HIP_MEMORY_SCOPE_AGENT 1
#define __HIP_MEMORY_SCOPE_SYSTEM 2
// (every load is modelled as ACQUIRE: the grid reads PUBLISHED / HEAD relaxed and then issues an acquire FENCE before it touches the
//  slot — jv_serve_claim.h "order its reads of the slot behind the claim" — which C++ treats as synchronising with the release
//  it read from; ThreadSanitizer does not model fences, so the acquire is put on the loads instead)
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
static inline int atomicCAS(int* p, int cmp, int val) {
    __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    return cmp;
}
static inline int atomicMax(int* p, int v) {
    int cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) {
    }
    return cur;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline unsigned long long sim_ticks() {  // s_memrealtime: 100 MHz
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (unsigned long long)ts.tv_sec * 100000000ull + (unsigned long long)ts.tv_nsec / 10ull;
}
#define __builtin_amdgcn_s_memrealtime() sim_ticks()
#define __builtin_amdgcn_s_sleep(n) sched_yield()
#define __builtin_amdgcn_fence(order, scope) ((void)0)   /* (ThreadSanitizer does not model fences; the read-modify-writes above already carry acq_rel) */
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __syncthreads() ((void)0)
struct SimDim { unsigned x; };
static thread_local SimDim threadIdx = {0};
static SimDim gridDim = {1};

#include "jv_serve_claim.h"   // the grid side, as the kernels run it
#include "jv_serve_host.h"    // the caller side, as jv_abi.cpp runs it

// ---- the server object: the members jv_serve_host.h names, as in jv_abi.cpp's JvQueryServer ----
struct SimServer {
    unsigned char* ring = nullptr;
    int slots = 0, slot_bytes = 0;
    int32_t* h_words = nullptr;   // "pinned": JV_SH_TAIL / STOP / ALIVE
    int32_t* d_words = nullptr;   // "device": JV_SV_*
    std::atomic<uint32_t> reserve{0};
    std::atomic<uint32_t>* slot_free = nullptr;
    std::atomic<int> lat_us{200};
    std::atomic<int> waiters{0};
    int spin_waiters = 4;
    std::mutex mu;                // launch / stop
    std::vector<std::thread> grid;
    JvSearchArgs args{};
    int d = 16;
    std::atomic<long> launches{0}, served{0}, skipped{0};
};

static void workgroup(SimServer* sv) {  // one resident workgroup (thread 0's view: the other lanes only follow)
    const JvSearchArgs& a = sv->args;
    for (;;) {
        const int ticket = jv_serve_claim(a);
        if (ticket < 0) break;
        unsigned char* sp = a.serve_ring + (size_t)(ticket & (a.serve_slots - 1)) * (size_t)a.serve_slot_bytes;
        JvServeSlot* slot = (JvServeSlot*)sp;
        if (!jv_serve_slot_current(slot, ticket)) {  // an abandoned ticket: nothing to answer
            sv->skipped++;
            continue;
        }
        const float* q = (const float*)(sp + JV_SERVE_QUERY_OFF);
        const int topK = slot->topK;
        float acc = 0.0f;
        for (int i = 0; i < sv->d; i++) acc += q[i] * (float)(i + 1);
        for (int i = 0; i < topK; i++) {
            slot->nodes[i] = (int32_t)acc + i + (slot->accept ? 1000000 : 0);
            slot->docs[i] = slot->rk + i;
            slot->scores[i] = acc;
        }
        slot->count = topK;
        slot->stats[0] = ticket;
        slot->flags = 0;
        sv->served++;
        __hip_atomic_store(&slot->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    jv_serve_leave(a);
}

// server_launch_locked (csrc/jv_abi.cpp): sv->mu held, grid not alive
static int launch_locked(SimServer* sv, int blocks) {
    for (std::thread& t : sv->grid) t.join();   // "hipStreamSynchronize": the previous grid's launch retires
    sv->grid.clear();
    for (int i = JV_SV_LOCK; i <= JV_SV_STOP_SEEN; i++) __atomic_store_n(&sv->d_words[i], 0, __ATOMIC_RELAXED);
    __atomic_store_n(&sv->h_words[JV_SH_ALIVE], 1, __ATOMIC_RELEASE);
    gridDim.x = (unsigned)blocks;
    for (int b = 0; b < blocks; b++) sv->grid.emplace_back(workgroup, sv);
    sv->launches++;
    return 0;
}
// server_stop_locked: ask the grid to leave and wait until it has
static void stop_locked(SimServer* sv) {
    __atomic_store_n(&sv->h_words[JV_SH_STOP], 1, __ATOMIC_RELEASE);
    for (std::thread& t : sv->grid) t.join();
    sv->grid.clear();
    __atomic_store_n(&sv->h_words[JV_SH_ALIVE], 0, __ATOMIC_RELEASE);
    __atomic_store_n(&sv->h_words[JV_SH_STOP], 0, __ATOMIC_RELEASE);
}

int main(int argc, char** argv) {
    const int callers = argc > 1 ? atoi(argv[1]) : 64;
    const int per_caller = argc > 2 ? atoi(argv[2]) : 200;
    const int blocks = argc > 3 ? atoi(argv[3]) : 6;
    const int fail_every = argc > 4 ? atoi(argv[4]) : 37;  // every n-th launch attempt "fails" (0 = never)
    alarm(120);  // a deadlock is a failure, not a hang
    SimServer sv;
    sv.slots = 32;
    sv.slot_bytes = (int)((sizeof(JvServeSlot) + sv.d * sizeof(float) + 63) & ~63u);
    sv.ring = (unsigned char*)aligned_alloc(64, (size_t)sv.slots * sv.slot_bytes);
    memset(sv.ring, 0, (size_t)sv.slots * sv.slot_bytes);
    sv.h_words = (int32_t*)calloc(16, 4);
    sv.d_words = (int32_t*)calloc(16, 4);
    sv.slot_free = new std::atomic<uint32_t>[sv.slots];
    for (int i = 0; i < sv.slots; i++) sv.slot_free[i].store((uint32_t)i);
    for (int i = 0; i < sv.slots; i++) ((JvServeSlot*)(sv.ring + (size_t)i * sv.slot_bytes))->ticket = -1;
    sv.args.serve_ring = sv.ring;
    sv.args.serve_slots = sv.slots;
    sv.args.serve_slot_bytes = sv.slot_bytes;
    sv.args.serve_dev = sv.d_words;
    sv.args.serve_host = sv.h_words;
    sv.args.serve_idle_ticks = 30000;  // 0.3 ms without a claim: the grid leaves (the product: 100 ms)
    std::atomic<long> ok{0}, abandoned{0}, wrong{0}, attempts{0};
    std::atomic<bool> stop_pauser{false};
    std::thread pauser([&] {  // hipFree / wide launches: pause the server now and then
        while (!stop_pauser.load()) {
            {
                std::lock_guard<std::mutex> lk(sv.mu);
                stop_locked(&sv);
            }
            std::this_thread::sleep_for(std::chrono::microseconds(700));
        }
    });
    std::vector<std::thread> th;
    for (int c = 0; c < callers; c++) {
        th.emplace_back([&, c] {
            std::vector<float> q((size_t)sv.d);
            for (int it = 0; it < per_caller; it++) {
                for (int i = 0; i < sv.d; i++) q[(size_t)i] = (float)((c * 131 + it * 7 + i) % 17);
                const int topK = 1 + (c + it) % 8;
                const bool filt = ((c + it) % 3) == 0;
                int si = 0;
                const uint32_t seq = jvsh_take_slot(&sv, &si);
                JvServeSlot* slot = jvsh_slot(&sv, si);
                slot->topK = topK;
                slot->rk = 100 + it;
                slot->visit_limit = 0;
                slot->rerank_floor = 0.0f;
                slot->accept = filt ? 0x1234u : 0u;
                slot->accept_docs = 0;
                slot->done = 0;
                slot->count = 0;
                slot->flags = 0;
                __atomic_store_n(&slot->ticket, (int32_t)seq, __ATOMIC_RELAXED);
                memcpy((unsigned char*)slot + JV_SERVE_QUERY_OFF, q.data(), (size_t)sv.d * sizeof(float));
                jvsh_publish(&sv, seq);
                auto ensure_alive = [&](auto&& give_up) -> int {   // serve_query's lambda (csrc/jv_abi.cpp), launch failures injected
                    if (__atomic_load_n(&sv.h_words[JV_SH_ALIVE], __ATOMIC_ACQUIRE) != 0) return 0;
                    std::lock_guard<std::mutex> lk(sv.mu);
                    if (__atomic_load_n(&sv.h_words[JV_SH_ALIVE], __ATOMIC_ACQUIRE) != 0) return 0;
                    int r = 0;
                    if (fail_every > 0 && (attempts.fetch_add(1) + 1) % fail_every == 0) r = -3;  // "JV_EDEVICE": the launch failed
                    else r = launch_locked(&sv, blocks);
                    if (r != 0) give_up();
                    return r;
                };
                if (jvsh_wait_done(&sv, slot, seq, si, ensure_alive) != 0) {
                    abandoned++;
                    continue;
                }
                float acc = 0.0f;
                for (int i = 0; i < sv.d; i++) acc += q[(size_t)i] * (float)(i + 1);
                bool good = slot->count == topK && slot->stats[0] == (int32_t)seq;
                for (int i = 0; i < topK && good; i++)
                    good = slot->nodes[i] == (int32_t)acc + i + (filt ? 1000000 : 0) && slot->docs[i] == 100 + it + i && slot->scores[i] == acc;
                (good ? ok : wrong)++;
                jvsh_release(&sv, seq, si);
                if ((it & 15) == 15) std::this_thread::sleep_for(std::chrono::microseconds(900));  // a pause: the grid may idle out
            }
        });
    }
    for (std::thread& t : th) t.join();
    stop_pauser.store(true);
    pauser.join();
    {
        std::lock_guard<std::mutex> lk(sv.mu);
        stop_locked(&sv);
    }
    const long total = (long)callers * per_caller;
    printf("calls %ld: answered and verified %ld, abandoned (failed launches) %ld, wrong %ld | grid launches %ld, served %ld, skipped tickets %ld\n",
           total, ok.load(), abandoned.load(), wrong.load(), sv.launches.load(), sv.served.load(), sv.skipped.load());
    const bool pass = wrong.load() == 0 && ok.load() + abandoned.load() == total && sv.served.load() == ok.load() && sv.launches.load() >= 3;
    free(sv.ring);
    free(sv.h_words);
    free(sv.d_words);
    delete[] sv.slot_free;
    return pass ? 0 : 1;
}
