// One element of the F(2x2, 3x3) filter transform U = G g G^T, shared by the dedicated pack kernel (conv_winograd.hip) and the
// table-driven multi-pack launch (layout.hip) so that both produce the same bits.  Computed in double, rounded once.
#pragma once

namespace vatl {

// Fragment order of the packed filter: [n_tile][step = c / 8][position][nh][lane = (c % 8 / 4) * 32 + n % 32][c % 4], n_tile = n / (32 NH).
// One block of 256 threads makes the 16 x 256 elements of (32 output channels, one 8-channel step): thread = (channel n % 32, c % 8) reads
// its filter taps once and writes one float per position — per position the block writes 1 KB contiguous.
//   bl = block index inside the filter = (n / 32) * (Cin / 8) + c / 8;   blocks per filter = CoutPad / 32 * Cin / 8 = elements / 4096
// mode 0: F(2x2,3x3), g = w[n][c] of a (Cout, Cin, 3, 3) filter (w_i = Cin);
// mode 1: its data gradient, g = rot180(w[o = c][i = n]) of the forward filter (O, I, 3, 3) (w_i = I);
// mode 2: F(3x3,2x2) of ConvTranspose2d(4,2,1), w = (Cin, Cout, 4, 4) (w_i = Cout): four phase filters one after another
//         (bl = phase * blocks per filter + ...), phase (py, px): g[a][b] = w[c][n][3 - py - 2a][3 - px - 2b], G = [1 0; .5 .5; .5 -.5; 0 -1].
// mode 3: data gradient of ConvTranspose2d(4,2,1) (a 4x4 / stride 2 conv over dz): w = (Cin_d, Cout_d, 4, 4) read as [n = Cin_d][c = Cout_d]
//         (w_i = Cout_d); four INPUT-phase filters one after another, phase (p, q): g[a][b] = w[n][c][2a + 1 - p][2b + 1 - q], same G as mode 2.
// n >= Cout: zero (padding rows of the last channel tile).  Computed in double, rounded once.
__device__ __forceinline__ void wino_pack_block(const float* __restrict__ w, float* __restrict__ out, int mode, int w_i, int Cout, int Cin, int NH,
                                                long long bl, int tid) {
    const int steps = Cin >> 3;
    int phase = 0;
    if (mode >= 2) {
        const long long per = (long long)((Cout + 32 * NH - 1) / (32 * NH)) * NH * steps;
        phase = (int)(bl / per);
        bl -= phase * per;
        out += phase * per * 4096;
    }
    const int n32 = (int)(bl / steps), step = (int)(bl - (long long)n32 * steps);
    const int nl = tid & 31, cc = tid >> 5;
    const int n = n32 * 32 + nl, c = step * 8 + cc;
    const int n_tile = n32 / NH, nh = n32 - n_tile * NH;
    const int lane = (cc >> 2) * 32 + nl, tt = cc & 3;
    double tg[4][3];
    if (mode >= 2) {
        const int py = phase >> 1, px = phase & 1;
        double g[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                double v = 0.0;
                if (n < Cout) v = mode == 2 ? (double)w[(((long long)c * w_i + n) * 4 + (3 - py - 2 * a)) * 4 + (3 - px - 2 * b)]
                                            : (double)w[(((long long)n * w_i + c) * 4 + (2 * a + 1 - py)) * 4 + (2 * b + 1 - px)];
                g[a][b] = v;
            }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            tg[0][s] = g[0][s];
            tg[1][s] = 0.5 * (g[0][s] + g[1][s]);
            tg[2][s] = 0.5 * (g[0][s] - g[1][s]);
            tg[3][s] = -g[1][s];
        }
    } else {
        double g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                double v = 0.0;
                if (n < Cout) v = mode == 0 ? (double)w[((long long)n * w_i + c) * 9 + r * 3 + s] : (double)w[((long long)c * w_i + n) * 9 + (2 - r) * 3 + (2 - s)];
                g[r][s] = v;
            }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            tg[0][s] = g[0][s];
            tg[1][s] = 0.5 * (g[0][s] + g[1][s] + g[2][s]);
            tg[2][s] = 0.5 * (g[0][s] - g[1][s] + g[2][s]);
            tg[3][s] = g[2][s];
        }
    }
    float* o = out + ((((long long)(n_tile * steps + step) * 16) * NH + nh) * 64 + lane) * 4 + tt;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double uu[4];
        if (mode >= 2) { uu[0] = tg[a][0]; uu[1] = 0.5 * (tg[a][0] + tg[a][1]); uu[2] = 0.5 * (tg[a][0] - tg[a][1]); uu[3] = -tg[a][1]; }
        else { uu[0] = tg[a][0]; uu[1] = 0.5 * (tg[a][0] + tg[a][1] + tg[a][2]); uu[2] = 0.5 * (tg[a][0] - tg[a][1] + tg[a][2]); uu[3] = tg[a][2]; }
#pragma unroll
        for (int b = 0; b < 4; ++b) o[(long long)(a * 4 + b) * NH * 256] = (float)uu[b];
    }
}

// One 16-byte item of the F(4x4,3x3) filter transform U = G g G^T (csrc/winograd_f4.hip), shared by its pack kernel and the table-driven multi-pack launch.
// Fragment order [stage = c / 16][position xi * 6 + nu][column block = n / 16][lane = (c % 16 / 4) * 16 + n % 16][k-step = c % 4]; item = (stage, position, block, lane).
// dgrad = 0: g = w[n][c] of the (Cout, Cin, 3, 3) filter (w_i = Cin); 1: the data gradient's filter g = rot180(w[o = c][i = n]) of the forward (O, I, 3, 3) filter
// (w_i = I = the packed Cout).  Float64 arithmetic, rounded once.
__device__ __forceinline__ void f4_pack_item(const float* __restrict__ w, float* __restrict__ u, int dgrad, int w_i, int Cout, int Cin, long long id) {
    const int nbg = Cout >> 4;
    const long long total = (long long)(Cin >> 4) * 36 * nbg * 64;
    if (id >= total) return;
    const int lane = (int)(id & 63);
    const int blk = (int)((id >> 6) % nbg);
    const int pos = (int)((id >> 6) / nbg % 36);
    const int stage = (int)((id >> 6) / nbg / 36);
    const int xi = pos / 6, nu = pos - 6 * xi;
    const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int n = 16 * blk + (lane & 15);
    float o[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 16 * stage + 4 * (lane >> 4) + s;
        const float* g = dgrad ? w + ((long long)c * w_i + n) * 9 : w + ((long long)n * w_i + c) * 9;
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) v += G[xi][i] * (double)(dgrad ? g[(2 - i) * 3 + (2 - j)] : g[i * 3 + j]) * G[nu][j];
        o[s] = (float)v;
    }
    float* dst = u + id * 4;
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
}

}  // namespace vatl
