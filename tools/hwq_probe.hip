// hwq_probe.hip — which streams share a hardware queue with a never-ending kernel?  A spinning kernel is started on one stream
// (default priority, then highest, then lowest); empty kernels are then launched on 12 fresh default-priority streams and on the
// null stream, and polled for 300 ms.  Streams whose kernel does not finish sit behind the spinner in the same hardware queue.
// Development aid (the resident query-server grids of jv_abi.cpp must not block other streams).
// build: hipcc --offload-arch=gfx950 -O2 -o build/exp/hwq_probe tools/hwq_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void spin(volatile int* flag) { while (*flag == 0) __builtin_amdgcn_s_sleep(64); }
__global__ void noop(volatile int* out) { if (threadIdx.x == 0) *out = 1; }

int main() {
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    printf("stream priority range: least %d greatest %d\n", least, greatest);
    int* flag;
    hipHostMalloc((void**)&flag, 4, hipHostMallocMapped | hipHostMallocCoherent);
    int* out;  // [16] pinned: what a stream's kernel wrote is read by the host directly, whatever hipStreamQuery thinks
    hipHostMalloc((void**)&out, 64, hipHostMallocMapped | hipHostMallocCoherent);
    const int prios[3] = {0, greatest, least};
    const char* names[3] = {"default", "highest", "lowest"};
    for (int p = 0; p < 3; p++) {
        *flag = 0;
        hipStream_t sp;
        if (p == 0) hipStreamCreateWithFlags(&sp, hipStreamNonBlocking);
        else hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, prios[p]);
        hipStream_t s[12];
        int blocked = 0;
        char map[16] = {0};
        for (int i = 0; i < 12; i++) {  // the other streams exist and have run a kernel before the spinner starts
            hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
            noop<<<1, 64, 0, s[i]>>>(out + i);
        }
        hipDeviceSynchronize();
        for (int i = 0; i < 16; i++) out[i] = 0;
        spin<<<1, 64, 0, sp>>>(flag);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        for (int i = 0; i < 12; i++) noop<<<1, 64, 0, s[i]>>>(out + i);
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        for (int i = 0; i < 12; i++) {
            const bool done = ((volatile int*)out)[i] == 1;
            map[i] = done ? '.' : 'B';
            blocked += done ? 0 : 1;
        }
        // does the RUNTIME see those completions?  (hipStreamQuery / hipStreamSynchronize on a stream whose kernel has finished)
        int query_not_ready = 0;
        for (int i = 0; i < 12; i++) query_not_ready += (map[i] == '.' && hipStreamQuery(s[i]) != hipSuccess) ? 1 : 0;
        int first_done = -1;
        for (int i = 0; i < 12 && first_done < 0; i++) if (map[i] == '.') first_done = i;
        double sync_ms = -1.0;
        if (first_done >= 0) {
            const auto t0 = std::chrono::steady_clock::now();
            std::thread rel([&]() { std::this_thread::sleep_for(std::chrono::milliseconds(500)); *flag = 1; });  // (a safety net: release the spinner after 0.5 s)
            hipStreamSynchronize(s[first_done]);
            sync_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            rel.join();
        }
        printf("spinner on a %s-priority stream: %d of 12 default streams blocked  [%s]; finished kernels hipStreamQuery calls not ready: %d; "
               "hipStreamSynchronize of a finished stream took %.1f ms (500 = it waited for the spinner)\n", names[p], blocked, map, query_not_ready, sync_ms);
        *flag = 1;
        hipDeviceSynchronize();
        for (int i = 0; i < 12; i++) hipStreamDestroy(s[i]);
        hipStreamDestroy(sp);
    }
    return 0;
}
