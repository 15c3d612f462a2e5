// How much of the per-image attention kernels' time is the padding of T = 197 to 7 x 32 tokens?  Times forward and (ring) backward
// at batch 256, 12 heads of 64, for token counts around the tile boundaries: 160 (5 tiles), 192 (6 tiles, no padding), 193 (7 tiles,
// ONE token in the last), 197 (ViT-B/16 at 224 px), 224 (7 full tiles).  If the 7th tile's padding were what the time is made of,
// t(192) would sit a tile step below t(193) and t(193) = t(197) = t(224); measured (profiles/r04_attn_T_sweep.txt), the time
// follows the TOKEN count instead (us per token nearly constant), i.e. the bytes and per-row latencies, not the tile grid.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<csrc> tools/attn_T_sweep.hip -o tools/attn_T_sweep && tools/attn_T_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "attention32.hip"
using namespace VLNS;
Profiler* g_prof = nullptr;
int g_poison_lds = 0;
void vl_poison_lds(hipStream_t) {}
int main() {
    const int B = 256, H = 12, D = 768, TM = 224;
    const size_t nq = (size_t)B * TM * 3 * D, nc = (size_t)B * TM * D;
    std::vector<unsigned short> hq(nq), hc(nc);
    srand(1);
    auto rnd = [] { _Float16 f = (_Float16)(rand() / (float)RAND_MAX - 0.5f); unsigned short u; memcpy(&u, &f, 2); return u; };
    for (auto& v : hq) v = rnd();
    for (auto& v : hc) v = rnd();
    h16 *qkv, *ctx, *dctx, *dqkv; float* lse;
    hipMalloc(&qkv, nq * 2); hipMalloc(&dqkv, nq * 2); hipMalloc(&ctx, nc * 2); hipMalloc(&dctx, nc * 2); hipMalloc(&lse, (size_t)B * H * TM * 4);
    hipMemcpy(qkv, hq.data(), nq * 2, hipMemcpyHostToDevice); hipMemcpy(dctx, hc.data(), nc * 2, hipMemcpyHostToDevice);
    attention32_init(0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    double t192f = 0, t192b = 0;
    for (int T : {160, 192, 193, 197, 224}) {
        for (int i = 0; i < 3; ++i) { k_attention_img_fwd(qkv, ctx, lse, B, T, H, D, nullptr, nullptr, 0, 0); k_attention_img_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, nullptr, nullptr, 8, 0u, 0); }
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) k_attention_img_fwd(qkv, ctx, lse, B, T, H, D, nullptr, nullptr, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double tf = ms * 100.0;
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) k_attention_img_bwd(qkv, ctx, dctx, lse, dqkv, B, T, H, D, nullptr, nullptr, 8, 0u, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double tb = ms * 100.0;
        if (T == 192) { t192f = tf; t192b = tb; }
        printf("T %3d (%d tiles of 32): forward %6.1f us (%.3f us / token), backward %6.1f us (%.3f us / token) per launch%s\n", T, (T + 31) / 32,
               tf, tf / T, tb, tb / T, T == 192 ? "   [image stride 192 * 4608 B = 27 * 2^15: every workgroup starts on the same HBM channel]" : "");
        (void)t192f; (void)t192b;
    }
    return 0;
}
