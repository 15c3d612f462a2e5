// Micro-benchmark: how fast can one workgroup per CU push a 256x256 bf16 tile (128 KB) to HBM,
// by store pattern?  (a) the MFMA C/D pattern of gemm256 (lane = row, 16 B pieces strided by 32 B),
// (b) same rows but the two 16 B pieces of a lane quad contiguous (64 B runs), (c) fully coalesced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(char* out, int ld_bytes, int tiles_per_wg, int tilesN) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3, fr = lane & 15, fg = lane >> 4;
    f32x4 v = {1.f * tid, 2.f, 3.f, 4.f};
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int bm = tile / tilesN, bn = tile % tilesN;
        char* base = out + (size_t)bm * 256 * ld_bytes + bn * 512;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                char* r = base + (size_t)(wm * 128 + i * 16 + fr) * ld_bytes + wn * 128 + fg * 32;
                *(f32x4*)r = v; *(f32x4*)(r + 16) = v;
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                char* r = base + (size_t)(wm * 128 + i * 16 + fr) * ld_bytes + wn * 128 + fg * 16;
                *(f32x4*)r = v; *(f32x4*)(r + 64) = v;
            }
        } else if (MODE == 3) {
            // per-wave 128 x 64 sub-tile written as full 128 B lines: a wave instruction = 8 rows x 128 B
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                char* r = base + (size_t)(wm * 128 + i * 8 + (lane >> 3)) * ld_bytes + wn * 128 + (lane & 7) * 16;
                *(f32x4*)r = v;
            }
        } else {
            // coalesced: a wave instruction writes 2 rows x 512 B
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = w * 32 + i * 2 + (lane >> 5);
                char* r = base + (size_t)row * ld_bytes + (lane & 31) * 16;
                *(f32x4*)r = v;
            }
        }
    }
}
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256;
    const int M = 50432, N = 3072, tilesN = N / 256, ntiles = (M / 256) * tilesN;
    char* out; hipMalloc(&out, (size_t)M * N * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<G, 512>>>(out, N * 2, 9, tilesN);
            if (mode == 1) k<1><<<G, 512>>>(out, N * 2, 9, tilesN);
            if (mode == 2) k<2><<<G, 512>>>(out, N * 2, 9, tilesN);
            if (mode == 3) k<3><<<G, 512>>>(out, N * 2, 9, tilesN);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double bytes = 9.0 * G * 131072.0;
        printf("mode %d: %.1f us  %.2f TB/s  (%.1f us per 128KB tile per CU)\n", mode, best * 1e3, bytes / best / 1e9, best * 1e3 / 9);
    }
    return 0;
}
