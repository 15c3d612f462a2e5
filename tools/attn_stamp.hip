// Diagnostic: where does a workgroup of the attention backward spend its time?  Builds the kernel file with
// s_memtime stamps (VITLORA_ATTN_STAMPS) and prints mean cycle counts per segment over all waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVITLORA_ATTN_STAMPS -I<csrc> tools/attn_stamp.hip -o tools/attn_stamp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "attention32.hip"
Profiler* g_prof = nullptr;
int main() {
    const int B = 256, T = 197, H = 12, D = 768;
    const size_t nq = (size_t)B * T * 3 * D, nc = (size_t)B * T * D;
    std::vector<unsigned short> hq(nq), hc(nc);
    srand(1);
    auto rnd = [] { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto& v : hq) v = rnd();
    for (auto& v : hc) v = rnd();
    bf16 *qkv, *ctx, *dctx, *dqkv; float* lse;
    hipMalloc(&qkv, nq * 2); hipMalloc(&dqkv, nq * 2); hipMalloc(&ctx, nc * 2); hipMalloc(&dctx, nc * 2); hipMalloc(&lse, (size_t)B * H * T * 4);
    hipMemcpy(qkv, hq.data(), nq * 2, hipMemcpyHostToDevice); hipMemcpy(dctx, hc.data(), nc * 2, hipMemcpyHostToDevice);
    attention32_init();
    k_attention32_fwd(qkv, ctx, lse, B, T, H, D, 0);
    for (int i = 0; i < 3; ++i) k_attention32_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k_attention32_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("bwd kernel %.1f us for %d workgroups\n", ms * 1e3, B * H);
    std::vector<unsigned long long> st(8192 * 8 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_attn_stamps), st.size() * 8);
    const int nblk = B * H < 8192 ? B * H : 8192;
    const char* names[] = {"issue loads (0->7)", "stage+barrier (0->1)", "B main loop (1->2)", "B stores (2->3)", "A main loop (4->5)", "A stores (5->6)", "whole wave (0->6)"};
    const int a[] = {0, 0, 1, 2, 4, 5, 0}, b[] = {7, 1, 2, 3, 5, 6, 6};
    for (int wv = 0; wv < 8; wv += (wv == 0 ? 6 : 1)) {       // wave 0 (two items), 6 (B only), 7 (A only)
        printf("wave %d:", wv);
        for (int k = 0; k < 7; ++k) {
            double sum = 0; int n = 0;
            for (int blk = 256; blk < nblk; ++blk) {          // skip the first wave of workgroups (cold start)
                const unsigned long long* s = &st[(blk * 8 + wv) * 8];
                if (wv == 6 && (k == 4 || k == 5)) continue;
                if (wv == 7 && (k == 2 || k == 3)) continue;
                if (s[b[k]] > s[a[k]]) { sum += (double)(s[b[k]] - s[a[k]]); ++n; }
            }
            if (n) printf("  %s %.0f", names[k], sum / n);
        }
        printf("\n");
    }
    return 0;
}
