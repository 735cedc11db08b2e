// wave_placement_probe.hip — on which SIMD of its CU does wave w of a workgroup land?  The several-waves search kernels give
// wave 0 of every workgroup the serial part of a query (the pool); if the dispatcher hands a workgroup's waves to SIMDs in a
// fixed order, all pool waves of a CU share one or two SIMDs and the helper waves idle on the others.  The probe launches a
// persistent-grid-shaped kernel (W waves per workgroup, LDS sized so that OCC workgroups fit a CU) and prints, per wave index,
// how many waves landed on SIMD 0..3, plus how many distinct (SIMD) a CU's wave-0s cover.
// build: hipcc --offload-arch=gfx950 -O2 -o build/exp/wave_placement_probe tools/wave_placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void probe(unsigned* out, volatile int* go) {
    extern __shared__ int smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        out[(blockIdx.x * nw + w) * 2] = hw;
        out[(blockIdx.x * nw + w) * 2 + 1] = xcc;
    }
    // stay resident until every workgroup of the grid has been placed (a persistent grid's placement)
    if (threadIdx.x == 0) atomicAdd((int*)go, 1);
    while (*go < (int)gridDim.x) __builtin_amdgcn_s_sleep(32);
}

static void run(int W, int occ, int lds) {
    const int cus = 256, grid = cus * occ;
    unsigned* out;
    int* go;
    hipMalloc((void**)&out, (size_t)grid * W * 8);
    hipMalloc((void**)&go, 4);
    hipMemset(go, 0, 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    probe<<<grid, W * 64, lds>>>(out, go);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned> h((size_t)grid * W * 2);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    long hist[16][4];
    memset(hist, 0, sizeof hist);
    // per CU (xcc, se, sh, cu): how many wave-0s on each SIMD
    std::vector<int> percu(8 * 8 * 2 * 16 * 4, 0);
    for (int b = 0; b < grid; b++)
        for (int w = 0; w < W; w++) {
            unsigned hw = h[(size_t)(b * W + w) * 2], xcc = h[(size_t)(b * W + w) * 2 + 1] & 15;
            int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            hist[w][simd]++;
            if (w == 0) percu[(((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd]++;
        }
    printf("W=%d waves per workgroup, %d workgroups per CU (LDS %d B), grid %d\n", W, occ, lds, grid);
    for (int w = 0; w < W; w++) printf("  wave %d: SIMD0 %ld SIMD1 %ld SIMD2 %ld SIMD3 %ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    // the first two CUs in full: (workgroup, wave) -> SIMD / wave slot
    {
        unsigned first[2] = {~0u, ~0u};
        for (int b = 0; b < grid; b++) {
            unsigned hw0 = h[(size_t)(b * W) * 2], x0 = h[(size_t)(b * W) * 2 + 1] & 15;
            unsigned key = (x0 << 16) | (hw0 & 0xff00);
            if (first[0] == ~0u) first[0] = key;
            else if (first[1] == ~0u && key != first[0]) first[1] = key;
            for (int f = 0; f < 2; f++)
                if (key == first[f]) {
                    printf("  cu#%d wg %5d:", f, b);
                    for (int w = 0; w < W; w++) {
                        unsigned hw = h[(size_t)(b * W + w) * 2];
                        printf(" w%d=simd%u/slot%u", w, (hw >> 4) & 3, hw & 15);
                    }
                    printf("\n");
                }
        }
    }
    int cus_seen = 0, worst = 0;
    long spread[5] = {0, 0, 0, 0, 0};
    for (size_t c = 0; c < percu.size() / 4; c++) {
        int tot = 0, used = 0, mx = 0;
        for (int s = 0; s < 4; s++) { tot += percu[c * 4 + s]; used += percu[c * 4 + s] > 0; mx = percu[c * 4 + s] > mx ? percu[c * 4 + s] : mx; }
        if (!tot) continue;
        cus_seen++;
        spread[used]++;
        worst = mx > worst ? mx : worst;
    }
    printf("  CUs seen %d; CUs whose wave-0s cover 1/2/3/4 SIMDs: %ld/%ld/%ld/%ld; most wave-0s on one SIMD: %d\n", cus_seen, spread[1], spread[2],
           spread[3], spread[4], worst);
    hipFree(out);
    hipFree(go);
}

int main() {
    run(2, 8, 19 * 1024);   // PQ-32 throughput shape
    run(2, 3, 50 * 1024);   // PQ-32 latency shape
    run(4, 4, 38 * 1024);   // PQ-64
    run(4, 2, 76 * 1024);
    run(1, 8, 19 * 1024);
    return 0;
}
