// Where does the dispatcher put the blocks of a 256-thread / 67 KB-LDS launch (two resident blocks per CU), and when do they start?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/block_placement.hip -o /tmp/block_placement && /tmp/block_placement [blocks] [work]
// Every block records (XCC_ID, SE_ID, CU_ID) from HW_ID, its start and end on the constant 100 MHz clock; the host prints, per CU,
// the block indices in start order.  Answers: do blocks b and b + 256 share a CU (the assumption of the stagger knob)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>

struct Rec { unsigned hw, xcc; unsigned long long t0, t1; };

__global__ __launch_bounds__(256, 2) void probe(Rec* out, int work) {
    extern __shared__ float smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_readcyclecounter();
    float a = threadIdx.x;
    for (int i = 0; i < work; ++i) { a = a * 1.0001f + 0.5f; smem[threadIdx.x] = a; __syncthreads(); a += smem[(threadIdx.x + 1) & 255]; }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = Rec{hw, xcc, t0, t1}; if (a == 12345.f) out[0].hw = 0; }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1440, work = argc > 2 ? atoi(argv[2]) : 2000;
    Rec* d; hipMalloc(&d, blocks * sizeof(Rec));
    const int lds = 67 * 1024;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), lds, 0, d, work); hipDeviceSynchronize(); }
    std::vector<Rec> h(blocks); hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull; for (auto& r : h) tmin = std::min(tmin, r.t0);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < blocks; ++b) {
        const unsigned cu_id = (h[b].hw >> 8) & 15, sh = (h[b].hw >> 12) & 1, se = (h[b].hw >> 13) & 7, x = h[b].xcc & 15;
        cu[(x << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("%d blocks on %zu distinct (xcc, se, sh, cu)\n", blocks, cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        auto v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return h[a].t0 < h[b].t0; });
        if (shown++ < 12 || shown > (int)cu.size() - 2) {
            printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
            for (int b : v) printf("  b%-4d [%6llu, %6llu]", b, h[b].t0 - tmin, h[b].t1 - tmin);
            printf("\n");
        }
    }
    // histogram of (second block of a CU) - (first block of that CU)
    std::map<int, int> hist;
    for (auto& kv : cu) { auto v = kv.second; std::sort(v.begin(), v.end(), [&](int a, int b) { return h[a].t0 < h[b].t0; }); if (v.size() > 1) hist[v[1] - v[0]]++; }
    printf("index distance between the first two blocks of a CU:");
    for (auto& kv : hist) printf("  %d: %d", kv.first, kv.second);
    printf("\n");
    return 0;
}
