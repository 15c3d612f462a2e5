// Fused MLP of a narrow transformer block (csrc/mlp_fused.hip): arguments and entry points.
#pragma once
#include "common.h"

namespace VLNS {

struct MlpArgs {
    // GEMM 1: Z1[M, HID] = X[M, K1] Wa[HID, K1]^T   (forward: x, fc1.W;  backward: g, fc2.W^T)
    const h16* X; int ldx;            // rows may be narrower than K1: the K tail is read from the next row against zero weight columns
    int K1, K1_algo;                  // padded depth (multiple of 64, <= 128) / true depth (profiling)
    const h16* Wa; int ldwa;
    const float* bias1;               // forward: fc1's bias [HID]; backward: unused
    int HID;                          // hidden width (multiple of 128)
    h16* S; int lds_;                 // gelu'(z) [M, HID]: WRITTEN by the forward, READ by the backward
    // GEMM 2: Y[M, n_store] = Act[M, HID] Wb[128, HID]^T   (forward: gelu(z), fc2.W;  backward: d(z), fc1.W^T)
    const h16* Wb; int ldwb;          // 128 rows (zero rows beyond n_store)
    const float* bias2;               // forward: fc2's bias (128 entries); backward: nullptr
    h16* Y; int ldy; int n_store;     // result rows of n_store <= 128 columns
    int M, Mvalid;
    // LoRA of fc2 (one adapted module, r <= 16): forward t = gelu(z) Ldown^T, y += t Lup^T; backward u = g Ldown^T, d(a) += u Lup^T
    int lora;
    const h16* Ldown; int ldd;        // forward: Ad [>= 16 rows][HID]; backward: Bd [>= 16 rows][K1]
    const h16* Lup;                   // forward: sB [128 rows][64]; backward: sA^T [HID rows][64]
    int r_algo;
};

int mlp_fused_init();                         // kernel attributes; 0 = ok
bool mlp_fused_supports(const MlpArgs& a);
void launch_mlp_fused(const MlpArgs& a, int backward, hipStream_t s);

}  // namespace VLNS
