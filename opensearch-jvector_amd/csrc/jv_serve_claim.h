// jv_serve_claim.h — ticket protocol of the device-resident query servers (jv_serve_pqw_kernel in jv_pqw_body.h: unfiltered
// queries on the several-waves kernel; jv_serve_pqp_kernel in jv_kernels_pqsf.hip: queries with a doc filter on the one-wave
// pool kernel).  Callers publish ring slots in ticket order (host word TAIL); workgroups claim tickets with a
// compare-and-swap on HEAD once PUBLISHED has caught up with TAIL.  gfx950 / CDNA4.
#pragma once
#include "jv_device.h"

// thread 0 of a workgroup: the next ticket, or -1 when the grid should leave (host STOP, or nothing claimed for
// a.serve_idle_ticks)
__device__ __forceinline__ int jv_serve_claim(const JvSearchArgs& a) {
    int ticket = -1;
    const uint32_t t_idle0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    int polls = 0, idle_iters = 0;
    for (;;) {
        // (the host's STOP must be honoured under load too: it used to be looked at by idle workgroups only, so a pause — hipFree,
        //  a launch that needs the grid's LDS — waited until the one-query traffic paused by itself)
        if (__hip_atomic_load(&a.serve_dev[JV_SV_STOP_SEEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        const int h = __hip_atomic_load(&a.serve_dev[JV_SV_HEAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int pb = __hip_atomic_load(&a.serve_dev[JV_SV_PUBLISHED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pb - h <= 0) {
            // nothing published that is not claimed: ONE workgroup at a time looks at the host's tail word
            // (the host words are read over PCIe: by the lock holder only — hundreds of idle workgroups polling them
            //  would queue in front of the working ones' query fetches and row stores)
            if (__hip_atomic_load(&a.serve_dev[JV_SV_LOCK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
                atomicCAS(&a.serve_dev[JV_SV_LOCK], 0, 1) == 0) {
                const int ht = __hip_atomic_load(&a.serve_host[JV_SH_TAIL], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (ht - pb > 0) {
                    atomicMax(&a.serve_dev[JV_SV_PUBLISHED], ht);
                    pb = ht;
                } else if ((++polls & 15) == 0 && __hip_atomic_load(&a.serve_host[JV_SH_STOP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
                    __hip_atomic_store(&a.serve_dev[JV_SV_STOP_SEEN], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(&a.serve_dev[JV_SV_LOCK], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (pb - h <= 0) {
                const uint32_t now = (uint32_t)__builtin_amdgcn_s_memrealtime();
                const uint32_t last = (uint32_t)__hip_atomic_load(&a.serve_dev[JV_SV_LAST_CLAIM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (signed differences: another workgroup's claim may carry a later time stamp than `now`)
                const bool idle = (int32_t)(now - last) > a.serve_idle_ticks && (int32_t)(now - t_idle0) > a.serve_idle_ticks;
                if (idle || __hip_atomic_load(&a.serve_dev[JV_SV_STOP_SEEN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                // back off the longer this workgroup has had nothing to do (3 us .. 55 us): hundreds of idle
                // workgroups polling at full rate slow the working ones down (one query alone: 9.8 ms instead of 3.2)
                idle_iters++;
                const int naps = idle_iters < 16 ? 1 : (idle_iters < 128 ? 4 : 16);
                for (int z = 0; z < naps; z++) __builtin_amdgcn_s_sleep(127);
                continue;
            }
        }
        if (atomicCAS(&a.serve_dev[JV_SV_HEAD], h, h + 1) == h) {
            ticket = h;
            // every 32nd ticket's owner reads the host's STOP word (one PCIe read per 32 queries); the unclaimed tickets stay
            // published and are served by the next grid
            if ((h & 31) == 0 && __hip_atomic_load(&a.serve_host[JV_SH_STOP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)
                __hip_atomic_store(&a.serve_dev[JV_SV_STOP_SEEN], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.serve_dev[JV_SV_LAST_CLAIM], (int)(uint32_t)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the slot was written by the host BEFORE its release-store of TAIL, which some workgroup read with acquire before it
            // advanced PUBLISHED; this workgroup only saw PUBLISHED / HEAD (relaxed): order its reads of the slot behind the claim
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            break;
        }
    }
    return ticket;
}

// every thread: true when the slot's content belongs to the claimed ticket (a caller that gave up on a grid that could not be
// started leaves its ticket published and marks the slot; the slot may even belong to a later call by now)
__device__ __forceinline__ bool jv_serve_slot_current(const JvServeSlot* slot, int ticket) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(&slot->ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) == ticket;  // (wave-uniform)
}

// the last workgroup out tells the host that the grid is gone (a caller that finds work pending launches it again)
__device__ __forceinline__ void jv_serve_leave(const JvSearchArgs& a) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&a.serve_dev[JV_SV_EXITED], 1) == (int)gridDim.x - 1) {
            __hip_atomic_store(&a.serve_dev[JV_SV_EXITED], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.serve_host[JV_SH_ALIVE], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
