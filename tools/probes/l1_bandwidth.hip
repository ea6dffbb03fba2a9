// How many bytes per clock does a CU's vector-memory path deliver for 16-byte-per-lane loads whose data sits in the vector L1 / in the XCD's L2?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l1_bandwidth tools/probes/l1_bandwidth.hip && /tmp/l1_bandwidth
// Why: the F(4x4,3x3) Winograd kernel pulls its filter fragments (144 KB per block and 16-channel stage) and its staged pixels (36 KB) through this path; whether that
// is a bandwidth or a latency cost decides what can still be done about it (profiles/r06_notes.md §4).
// Each block (WAVES waves, one block per CU: grid = 256) sweeps a per-block window of WIN bytes ITER times with buffer_load_dwordx4 (64 lanes x 16 B = 1 KB per
// instruction, UNROLL independent loads in flight per wave); WIN = 16 KB stays in the 32 KB vector L1, WIN = 1 MB per block streams from L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(1024) void sweep(const float* __restrict__ buf, float* __restrict__ out, long long win_floats, int iters, long long* __restrict__ cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* base = buf + (long long)blockIdx.x * win_floats;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (unsigned)(win_floats * 4), 0x00020000);
    const unsigned step = (unsigned)nw * 1024u * UNROLL;                 // bytes covered by the block per loop trip
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        for (unsigned off = (unsigned)wave * 1024u * UNROLL + (unsigned)lane * 16u; off < (unsigned)(win_floats * 4); off += step) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off + u * 1024u, 0, 0));
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const int grid = 256;
    const long long max_win = 4LL << 20;                                   // bytes per block
    float *buf, *out; long long* cyc;
    (void)hipMalloc(&buf, grid * max_win); (void)hipMalloc(&out, grid * 1024 * sizeof(float)); (void)hipMalloc(&cyc, grid * sizeof(long long));
    hipMemset(buf, 0, grid * max_win);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<long long> h(grid);
    for (long long win : {16LL << 10, 64LL << 10, 1LL << 20, 4LL << 20}) {
        for (int waves : {4, 8, 16}) {
            const int iters = (int)((256LL << 20) / win / 4);              // 64 MB of loads per block
            sweep<8><<<grid, waves * 64>>>(buf, out, win / 4, 2, cyc); hipDeviceSynchronize();
            hipEventRecord(e0);
            sweep<8><<<grid, waves * 64>>>(buf, out, win / 4, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
            double avg = 0; for (auto c : h) avg += (double)c; avg /= grid;
            const double bytes = (double)win * iters;
            // s_memtime / readcyclecounter ticks at a fixed 100 MHz on gfx9: use the event time and report bytes per CU per ns; at ~2.1 GHz 1 B/ns = 0.48 B/clk
            printf("window %6lld KB/block, %2d waves/CU: %.3f ms, %.1f B/ns per CU = %.1f TB/s chip-wide (%.0f B/clk at 2.1 GHz); counter ticks %.0f\n", win >> 10, waves, ms,
                   bytes / (ms * 1e6), bytes * grid / (ms * 1e-3) / 1e12, bytes / (ms * 1e6) / 2.1, avg);
        }
    }
    return 0;
}
