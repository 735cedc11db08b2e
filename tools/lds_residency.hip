// lds_residency.hip — how many one-wave workgroups of a given dynamic-LDS size does a gfx950 CU really keep resident?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor answers from a formula; this measures.)  Every workgroup bumps a counter keyed by
// its (XCC, SE, CU) on entry, records the running maximum, idles ~200 us, leaves.  Development aid, not part of the product.
// build: hipcc --offload-arch=gfx950 -O2 -o build/exp/lds_residency tools/lds_residency.hip ; run: lds_residency <lds bytes>...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64) void probe(int* cur, int* peak, int* touched, long long ticks) {
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const int key = ((xcc & 15) << 8) | (se << 5) | (sh << 4) | cu;
    if (threadIdx.x == 0) {
        smem[0] = 1;
        const int now = atomicAdd(&cur[key], 1) + 1;
        atomicMax(&peak[key], now);
        touched[key] = 1;
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
        atomicSub(&cur[key], 1);
    }
}

int main(int argc, char** argv) {
    int *cur, *peak, *touched;
    hipMalloc(&cur, 4096 * 4); hipMalloc(&peak, 4096 * 4); hipMalloc(&touched, 4096 * 4);
    for (int i = 1; i < argc; i++) {
        const int lds = atoi(argv[i]);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipMemset(cur, 0, 4096 * 4); hipMemset(peak, 0, 4096 * 4); hipMemset(touched, 0, 4096 * 4);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)probe, 64, lds);
        probe<<<4096, 64, lds, 0>>>(cur, peak, touched, 20000);  // 100 MHz wall clock: 200 us
        hipError_t e = hipDeviceSynchronize();
        std::vector<int> p(4096), t(4096);
        hipMemcpy(p.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), touched, 4096 * 4, hipMemcpyDeviceToHost);
        int cus = 0, mx = 0, mn = 1 << 30; long long sum = 0;
        for (int k = 0; k < 4096; k++) if (t[k]) { cus++; mx = std::max(mx, p[k]); mn = std::min(mn, p[k]); sum += p[k]; }
        printf("lds %6d B: occupancy API says %d per CU; measured peak resident per CU: min %d max %d mean %.2f over %d CUs (%s)\n", lds, occ, mn, mx,
               cus ? (double)sum / cus : 0.0, cus, hipGetErrorString(e));
    }
    return 0;
}
