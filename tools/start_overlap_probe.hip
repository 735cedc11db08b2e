// start_overlap_probe.hip — what makes a small launch that overlaps the START of a never-ending kernel wait until that kernel
// leaves?  (jv_abi.cpp: a host-pointer batch call that overlaps a query-server grid's start returned only after the grid idled
// out.)  A spinner is started on a lowest-priority stream while another thread launches a worker + marker kernel on a normal
// stream at a small offset around the start; the marker's latency is measured.  Variants: private-segment (scratch) use of
// the spinner / of the worker, a hipMemsetAsync in front of the spinner (as server_launch_locked does), an H2D copy in front
// of the worker (as search_batch_host does).  Development aid.
// build: hipcc --offload-arch=gfx950 -O2 -o build/exp/start_overlap_probe tools/start_overlap_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

template <int SCR>
__global__ void spin(volatile int* flag, int* sink) {
    int acc = 0;
    if (SCR) {
        volatile int priv[SCR ? SCR : 1];
        for (int i = 0; i < SCR; i++) priv[i] = i + threadIdx.x;
        acc = priv[(threadIdx.x * 7) % SCR];
    }
    while (*flag == 0) __builtin_amdgcn_s_sleep(64);
    if (acc == -1) *sink = acc;
}
template <int SCR>
__global__ void work(int* sink) {
    int acc = 0;
    if (SCR) {
        volatile int priv[SCR ? SCR : 1];
        for (int i = 0; i < SCR; i++) priv[i] = i + threadIdx.x;
        acc = priv[(threadIdx.x * 5) % SCR];
    }
    if (acc == -1) *sink = acc;
}
__global__ void mark(volatile int* out, int v) { *out = v; }

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

int main(int argc, char** argv) {
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    int *flag, *out, *d_sink, *d_words, *d_buf, *h_buf;
    hipHostMalloc((void**)&flag, 4, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostMalloc((void**)&out, 4, hipHostMallocMapped | hipHostMallocCoherent);
    hipHostMalloc((void**)&h_buf, 65536, hipHostMallocDefault);
    hipMalloc((void**)&d_sink, 64);
    hipMalloc((void**)&d_words, 64);
    hipMalloc((void**)&d_buf, 65536);
    hipStream_t sp, sw;
    hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, least);
    hipStreamCreateWithFlags(&sw, hipStreamNonBlocking);
    const int spin_ms = argc > 1 ? atoi(argv[1]) : 40;
    struct Variant { const char* name; int s_scr, w_scr, memset_first, h2d_first, big_grid; };
    const Variant vs[] = {
        {"512 workgroups, no scratch anywhere", 0, 0, 0, 0, 1},
        {"512 workgroups, worker uses  64 B of scratch per lane", 0, 2, 0, 0, 1},
        {"512 workgroups, worker uses 128 B of scratch per lane", 0, 3, 0, 0, 1},
        {"512 workgroups, worker uses 256 B of scratch per lane", 0, 1, 0, 0, 1},
        {"512 workgroups, worker uses 1200 B of scratch per lane", 0, 4, 0, 0, 1},
        {"512 workgroups with scratch, worker uses 256 B", 1, 1, 0, 0, 1},
    };
    for (const Variant& v : vs) {
        std::vector<double> lat;
        int seq = 0;
        for (int rep = 0; rep < 40; rep++) {
            *flag = 0;
            const int off_us = (rep % 8) * 60 - 120;  // worker launch relative to the spinner launch: -120 .. +300 us
            std::thread starter([&]() {
                if (off_us < 0) std::this_thread::sleep_for(std::chrono::microseconds(-off_us));
                if (v.memset_first) hipMemsetAsync(d_words, 0, 16, sp);
                const int blocks = v.big_grid ? 512 : 1;
                if (v.s_scr) spin<64><<<blocks, 128, 0, sp>>>(flag, d_sink);
                else spin<0><<<blocks, 128, 0, sp>>>(flag, d_sink);
                std::this_thread::sleep_for(std::chrono::milliseconds(spin_ms));
                *flag = 1;
            });
            if (off_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(off_us));
            const auto t0 = clk::now();
            *out = 0;
            ++seq;
            if (v.h2d_first) hipMemcpyAsync(d_buf, h_buf, 4096, hipMemcpyHostToDevice, sw);
            if (v.w_scr == 1) work<64><<<64, 128, 0, sw>>>(d_sink);
            else if (v.w_scr == 2) work<16><<<64, 128, 0, sw>>>(d_sink);
            else if (v.w_scr == 3) work<32><<<64, 128, 0, sw>>>(d_sink);
            else if (v.w_scr == 4) work<300><<<64, 128, 0, sw>>>(d_sink);
            else work<0><<<64, 128, 0, sw>>>(d_sink);
            mark<<<1, 1, 0, sw>>>(out, seq);
            while (*(volatile int*)out != seq && ms_since(t0) < 1000.0) std::this_thread::yield();
            lat.push_back(ms_since(t0));
            starter.join();
            hipStreamSynchronize(sp);
            hipStreamSynchronize(sw);
        }
        printf("   in order:"); for (size_t i = 0; i < 16 && i < lat.size(); i++) printf(" %.2f", lat[i]); printf("\n");
        std::sort(lat.begin(), lat.end());
        int slow = 0;
        for (double x : lat) slow += x > 5.0 ? 1 : 0;
        printf("%-70s p50 %.3f ms  max %.3f ms  calls > 5 ms: %d of %zu\n", v.name, lat[lat.size() / 2], lat.back(), slow, lat.size());
        fflush(stdout);
    }
    return 0;
}
