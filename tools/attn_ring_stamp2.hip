// Diagnostic: where a head of the single-pass ("ring") attention backward spends its cycles (head 3 of every workgroup).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVITLORA_ATTN_STAMPS -I<csrc> tools/attn_ring_stamp.hip -o tools/attn_ring_stamp
// stamps: 0 head start | 1 step 3 start | 2 step 3 before its barrier | 3 after it | 4 after the last step | 5 before the stores |
//         6 before the end-of-head barrier | 7 after it          (wave 7 = loader)
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
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int ring = 1; ring >= 0; --ring) {
        g_attn_ring = ring;
        for (int i = 0; i < 3; ++i) k_attention_img_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, with_lora ? Bd : nullptr, u, 8, 7u, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) k_attention_img_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, with_lora ? Bd : nullptr, u, 8, 7u, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s backward: %.1f us per launch (batch %d, lora %d)\n", ring ? "ring" : "two-phase", ms * 1e3 / 5, B, (int)with_lora);
    }
    g_attn_ring = 1;
    k_attention_img_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, with_lora ? Bd : nullptr, u, 8, 7u, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(8192 * 8 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_attn_stamps), st.size() * 8);
    const char* names[] = {"step 0->7", "M1+frags (0->1)", "piece0 (1->2)", "piece1 (2->3)", "piece2 (3->4)", "piece3 (4->5)",
                           "E write+packs (5->6)", "barrier (6->7)", "-"};
    const int a[] = {0, 0, 1, 2, 3, 4, 5, 6, 0}, b2[] = {7, 1, 2, 3, 4, 5, 6, 7, 0};
    for (int wv = 0; wv < 8; ++wv) {
        printf("wave %d:", wv);
        for (int k = 0; k < 9; ++k) {
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
