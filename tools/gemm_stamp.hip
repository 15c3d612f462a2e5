// Diagnostic: per-tile timeline of the persistent 256x256 GEMM workgroups (s_memtime stamps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVITLORA_GEMM_STAMPS -I<csrc> tools/gemm_stamp.hip -o tools/gemm_stamp
//   tools/gemm_stamp N K epi      (epi: 0 h16 store, 2 GELU, 3 GELU_BWD, 7 none)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "gemm256.hip"
using namespace VLNS;      // the 16-bit sources live in vl_f16 / vl_bf16 (csrc/common.h)
Profiler* g_prof = nullptr;
__global__ void fill(unsigned short* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        float f = ((x & 0xffff) / 65535.f - 0.5f) * 0.25f; unsigned u = __builtin_bit_cast(unsigned, f); p[i] = (unsigned short)(u >> 16);
    }
}
int main(int argc, char** argv) {
    const int M = 50432, N = argc > 1 ? atoi(argv[1]) : 3072, K = argc > 2 ? atoi(argv[2]) : 768, epi = argc > 3 ? atoi(argv[3]) : 0;
    h16 *A, *W, *C, *C2, *R; float* bias;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&C2, (size_t)M * N * 2);
    hipMalloc(&R, (size_t)M * N * 4); hipMalloc(&bias, N * 4);
    fill<<<1024, 256>>>((unsigned short*)A, (size_t)M * K, 1); fill<<<1024, 256>>>((unsigned short*)W, (size_t)N * K, 2);
    fill<<<1024, 256>>>((unsigned short*)R, (size_t)M * N * 2, 3); hipMemset(bias, 0, N * 4);
    GemmArgs g; memset(&g, 0, sizeof g);
    g.A1 = A; g.lda1 = K; g.W1 = W; g.ldw1 = K; g.K1 = K; g.M = M; g.Mvalid = M; g.N = N;
    g.bias = bias; g.C = C; g.ldc = N; g.C2 = C2; g.ldc2 = N; g.R = R; g.ldr = N;
    g.dephase = argc > 4 ? atoi(argv[4]) : 0;
    gemm256_init();
    for (int i = 0; i < 3; ++i) launch_gemm256(g, epi, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); launch_gemm256(g, epi, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("N=%d K=%d epi=%d: %.1f us (big + small launch)\n", N, K, epi, ms * 1e3);
    std::vector<unsigned long long> st(256 * 8 * 16 * 4);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_gemm_stamps), st.size() * 8);
    // the stamps of the LAST launch that wrote them (the small launch overwrites tile 0 of its workgroups): use tiles 1..7
    for (int wv = 0; wv < 8; wv += 4) {
        double ml = 0, pro = 0, ep = 0, gap = 0; int n = 0;
        for (int blk = 0; blk < 256; ++blk)
            for (int it = 1; it < 7; ++it) {
                const unsigned long long* s = &st[((blk * 8 + wv) * 16 + it) * 4];
                const unsigned long long* nx = s + 4;
                if (!(s[3] > s[0]) || !(nx[0] >= s[3])) continue;
                ml += s[1] - s[0]; pro += s[2] - s[1]; ep += s[3] - s[2]; gap += nx[0] - s[3]; ++n;
            }
        if (n) printf("wave %d: main loop %.0f  issue next-tile loads %.0f  epilogue %.0f  (tile period %.0f cycles, n=%d)\n", wv, ml / n, pro / n, ep / n, (ml + pro + ep + gap) / n, n);
    }
    return 0;
}
