// Diagnostic: per-phase cycle counts of the per-image attention backward (head 3 of every workgroup).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVITLORA_ATTN_STAMPS -I<csrc> tools/attn_img_stamp.hip -o tools/attn_img_stamp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "attention32.hip"
using namespace VLNS;      // the 16-bit sources live in vl_f16 / vl_bf16 (csrc/common.h)
Profiler* g_prof = nullptr;
int main(int argc, char** argv) {
    const int B = 256, T = 197, H = 12, D = 768;
    const bool with_lora = argc > 1;
    const size_t nq = (size_t)B * T * 3 * D, nc = (size_t)B * T * D;
    std::vector<unsigned short> hq(nq), hc(nc);
    srand(1);
    auto rnd = [] { _Float16 f = (_Float16)(rand() / (float)RAND_MAX - 0.5f); unsigned short u; memcpy(&u, &f, 2); return u; };
    for (auto& v : hq) v = rnd();
    for (auto& v : hc) v = rnd();
    h16 *qkv, *ctx, *dctx, *dqkv, *Bd, *u; float* lse;
    hipMalloc(&qkv, nq * 2); hipMalloc(&dqkv, nq * 2); hipMalloc(&ctx, nc * 2); hipMalloc(&dctx, nc * 2); hipMalloc(&lse, (size_t)B * H * T * 4);
    hipMalloc(&Bd, 64 * 3 * D * 2); hipMalloc(&u, (size_t)B * T * 64 * 2 + 4096);
    hipMemcpy(qkv, hq.data(), nq * 2, hipMemcpyHostToDevice); hipMemcpy(dctx, hc.data(), nc * 2, hipMemcpyHostToDevice);
    hipMemcpy(Bd, hq.data(), 64 * 3 * D * 2, hipMemcpyHostToDevice);
    attention32_init(0);
    k_attention_img_fwd(qkv, ctx, lse, B, T, H, D, nullptr, nullptr, 0, 0);
    for (int i = 0; i < 3; ++i) k_attention_img_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, with_lora ? Bd : nullptr, u, 8, 7u, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k_attention_img_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, with_lora ? Bd : nullptr, u, 8, 7u, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("bwd img kernel %.1f us for %d workgroups (lora %d)\n", ms * 1e3, B, (int)with_lora);
    std::vector<unsigned long long> st(8192 * 8 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_attn_stamps), st.size() * 8);
    hipEventRecord(e0); k_attention32_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("bwd per-head kernel %.1f us\n", ms * 1e3);
    const char* names[] = {"A loop (0->1)", "A tail (1->2)", "A barrier wait (2->3)", "B loop (3->4)", "B down+stores (4->7)", "B fetch_a (7->5)", "B barrier wait (5->6)", "head (0->6)"};
    const int a[] = {0, 1, 2, 3, 4, 7, 5, 0}, b2[] = {1, 2, 3, 4, 7, 5, 6, 6};
    for (int wv = 0; wv < 8; ++wv) {
        printf("wave %d:", wv);
        for (int k = 0; k < 8; ++k) {
            double sum = 0; int n = 0;
            for (int blk = 0; blk < B; ++blk) {
                const unsigned long long* s = &st[(blk * 8 + wv) * 8];
                if (s[b2[k]] > s[a[k]] && s[a[k]]) { sum += (double)(s[b2[k]] - s[a[k]]); ++n; }
            }
            if (n) printf("  %s %.0f", names[k], sum / n);
        }
        printf("\n");
    }
    return 0;
}
