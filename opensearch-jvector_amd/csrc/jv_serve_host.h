// jv_serve_host.h — the CALLER side of the device-resident query servers' ticket protocol (the grid side is
// jv_serve_claim.h).  Pure C++, no HIP: jv_abi.cpp's serve_query runs it against pinned memory and a resident grid; the sanitizer
// test (tests/native/serve_sim.cpp, built with -fsanitize=thread and with address,undefined by tests/test_serve_protocol.py) runs
// the very same functions — and the very same jv_serve_claim.h, compiled for the host through a shim — against host threads
// that play the grid.
//
// One call = take a sequence number (it names both the ring slot, seq mod slots, and the ticket) -> wait until the slot's
// previous occupant has copied its row out -> fill the slot -> publish in ticket order (TAIL is release-stored from seq to
// seq + 1 by the owner of seq, so a grid that reads TAIL = t with acquire sees every slot below t) -> make sure a grid is alive
// -> sleep through most of the expected latency, then poll the slot's completion word -> copy the row out -> hand the slot on.
// A caller whose grid cannot be started marks its slot ABANDONED (another generation in slot->ticket) before handing it on:
// the ticket stays published, the grid that claims it later skips it.
//
// SV is any struct with: unsigned char* ring; int slots (a power of two), slot_bytes; int32_t* h_words (JV_SH_*);
// std::atomic<uint32_t> reserve; std::atomic<uint32_t>* slot_free; std::atomic<int> lat_us (running estimate of one query's
// latency in microseconds); std::atomic<int> waiters (callers inside jvsh_wait_done right now); int spin_waiters (up to this many
// of them poll with sched_yield() behind their first nap; 0 = naps only).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <sched.h>
#include <time.h>

#include "jv_device.h"

template <class SV>
inline uint32_t jvsh_take_slot(SV* sv, int* si_out) {
    const uint32_t seq = sv->reserve.fetch_add(1);
    const int si = (int)(seq & (uint32_t)(sv->slots - 1));
    for (int spins = 0; sv->slot_free[si].load(std::memory_order_acquire) != seq; spins++) {  // the slot's previous occupant is still reading its row
        if (spins > 64) sched_yield();
    }
    *si_out = si;
    return seq;
}
template <class SV>
inline JvServeSlot* jvsh_slot(SV* sv, int si) { return (JvServeSlot*)(sv->ring + (size_t)si * (size_t)sv->slot_bytes); }

// publish in ticket order
template <class SV>
inline void jvsh_publish(SV* sv, uint32_t seq) {
    for (int spins = 0; (uint32_t)__atomic_load_n(&sv->h_words[JV_SH_TAIL], __ATOMIC_ACQUIRE) != seq; spins++) {
        if (spins > 256) sched_yield();
    }
    __atomic_store_n(&sv->h_words[JV_SH_TAIL], (int32_t)(seq + 1), __ATOMIC_RELEASE);
}
// hand the slot to the call that holds sequence number seq + slots
template <class SV>
inline void jvsh_release(SV* sv, uint32_t seq, int si) { sv->slot_free[si].store(seq + (uint32_t)sv->slots, std::memory_order_release); }
// nobody answers this call: the ticket stays published, so the slot is marked abandoned before it is handed on — the grid that
// claims the ticket later finds another generation in the slot and skips it
template <class SV>
inline void jvsh_abandon(SV* sv, JvServeSlot* slot, uint32_t seq, int si) {
    __atomic_store_n(&slot->ticket, (int32_t)(seq ^ 0x40000000u), __ATOMIC_RELEASE);
    jvsh_release(sv, seq, si);
}
// sleep through most of the expected latency, then poll.  ensure_alive(give_up) returns 0 when a grid is (again) alive; when it
// cannot start one it calls give_up() WHILE IT STILL HOLDS THE LAUNCH LOCK and returns the reason.  The lock matters: a caller
// that marked its slot abandoned only after the lock was dropped raced with the NEXT caller's successful launch — the new grid
// could claim the still-unmarked ticket and write its row into a slot that was already handed to the next occupant (found by
// tests/native/serve_sim.cpp under ThreadSanitizer, round 5).  Returns 0 once slot->done is set (the row is final), else
// ensure_alive's code with the slot abandoned and handed on.
template <class SV, class Alive>
inline int jvsh_wait_done(SV* sv, JvServeSlot* slot, uint32_t seq, int si, Alive&& ensure_alive) {
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    bool answered = false;
    auto give_up = [&]() {
        // (the grid may have answered this very slot before it left: a row that is there is used, not recomputed)
        if (__atomic_load_n(&slot->done, __ATOMIC_ACQUIRE) != 0) answered = true;
        else jvsh_abandon(sv, slot, seq, si);
    };
    int rc = ensure_alive(give_up);
    if (rc != 0 && !answered) return rc;
    const int est = sv->lat_us.load(std::memory_order_relaxed);
    long nap_ns = (long)est * 600;  // 0.6 x
    // A few callers at a time (a lone searcher thread is the reference's latency case): after the first nap the caller polls with
    // sched_yield() instead of napping — a nanosleep of 60 us returns after ~110 (timer slack), which was ~0.06 ms of every
    // one-query call's p50.  With many callers waiting that would only burn the cores the other callers need, so they keep napping;
    // and a query that takes much longer than the estimate goes back to naps as well.
    const int waiting = sv->waiters.fetch_add(1, std::memory_order_relaxed) + 1;
    const bool spin = sv->spin_waiters > 0 && waiting <= sv->spin_waiters;
    for (int it = 0; !answered; it++) {
        if (__atomic_load_n(&slot->done, __ATOMIC_ACQUIRE) != 0) break;
        if (it < 3 && est < 200) {
            sched_yield();
            continue;
        }
        if (spin && it > 0) {
            struct timespec tn;
            clock_gettime(CLOCK_MONOTONIC, &tn);
            const int64_t waited_us = (int64_t)(tn.tv_sec - t0.tv_sec) * 1000000 + (tn.tv_nsec - t0.tv_nsec) / 1000;
            if (waited_us < 2 * (int64_t)est + 200) {
                for (int k = 0; k < 64 && __atomic_load_n(&slot->done, __ATOMIC_ACQUIRE) == 0; k++) sched_yield();
                if ((it & 63) == 63 && (rc = ensure_alive(give_up)) != 0 && !answered) {
                    sv->waiters.fetch_sub(1, std::memory_order_relaxed);
                    return rc;
                }
                continue;
            }
        }
        struct timespec ts = {0, std::max<long>(20000, std::min<long>(nap_ns, 5000000))};
        nanosleep(&ts, nullptr);
        nap_ns = std::max<long>(20000, (long)est * 25);  // then every est / 40
        if ((it & 7) == 7 && (rc = ensure_alive(give_up)) != 0 && !answered) {
            sv->waiters.fetch_sub(1, std::memory_order_relaxed);
            return rc;
        }
    }
    sv->waiters.fetch_sub(1, std::memory_order_relaxed);
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const int us = (int)std::min<int64_t>(1000000, (int64_t)(t1.tv_sec - t0.tv_sec) * 1000000 + (t1.tv_nsec - t0.tv_nsec) / 1000);
    sv->lat_us.store((est * 7 + us) / 8, std::memory_order_relaxed);
    return 0;
}
