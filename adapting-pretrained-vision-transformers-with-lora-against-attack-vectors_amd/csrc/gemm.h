// GEMM entry points shared between gemm.hip and the model driver.
#pragma once
#include "common.h"

namespace VLNS {

// C[M,N] = A1[M,K1] * W1[N,K1]^T + A2[M,K2] * W2[N,K2]^T  (+ epilogue)
// Both operands are K-contiguous h16 ("NT" form); the optional second pair is the
// rank-r LoRA update appended as extra K tiles:  [x | x A^T] * [W | s B]^T.
enum GemmEpilogue {
    EPI_STORE_H16 = 0,   // C(h16) = acc + bias
    EPI_RESID_F32 = 1,    // C(f32)  = acc + bias + R(f32)        (R may alias C)
    EPI_GELU = 2,         // z = acc + bias ; C(h16) = gelu(z) ; C2(h16) = gelu'(z)
    EPI_GELU_BWD = 3,     // C(h16) = acc * R(h16), R = the saved gelu'(z)
    EPI_PATCH_FWD = 4,    // C(h16)[row b*T + 1 + p] = acc + bias + pos[1+p]   (token rows of the 16-bit residual stream)
    EPI_PATCH_BWD = 5,    // image-layout scatter of d(patches), scaled by 1/std[c]
    EPI_STORE_F32 = 6,    // C(f32) = acc + bias
    EPI_NONE = 7,         // diagnostic: results kept live, nothing stored (timing of the main loop alone)
    EPI_DROP_ACC = 8,     // train-mode LoRA dgrad with dropout: C(h16) = (R(h16) + mask*acc) [* G(h16)]
    EPI_RESID_H16 = 10,   // C(h16) = round16(acc + bias + R(h16)): the residual add of the 16-bit stream in the o / fc2 epilogue (round 4)
    EPI_PATCH_PGD = 9,    // EPI_PATCH_BWD's pixel gradient consumed in registers: C(f32) = adv <- clamp(x0 + clamp(adv + alpha sign(g) - x0, +-eps), lo, hi), R = x0
};

struct GemmArgs {
    const h16* A1; const h16* W1; int lda1, ldw1, K1;
    const h16* A2; const h16* W2; int lda2, ldw2, K2;
    int M;            // rows computed (multiple of 128; buffers are padded to it)
    int Mvalid;       // rows that may be stored by the remapping epilogues
    int N;            // multiple of BN
    int n_store;      // 0, or the number of leading columns that are stored (a multiple of 16 < N: the result rows are NARROWER
                      // than the tile grid -- ldc may then be n_store; 128-row kernel only, launch_gemm routes accordingly)
    const float* bias;
    void* C; int ldc;
    void* C2; int ldc2;
    const void* R; int ldr;
    // patch epilogues
    const float* pos; int tokens; int patches; int grid; int psize; int img;
    float inv_std[3];
    const float* row_scale;   // EPI_PATCH_BWD: optional per-IMAGE factor (undoes the fp16 gradient scale), nullptr = 1
    int* err_flag;            // EPI_PATCH_BWD / EPI_PATCH_PGD: mapped host word, set to 2 when a pixel gradient is not finite (nullptr = unchecked)
    float pgd_eps, pgd_alpha, pgd_lo, pgd_hi;   // EPI_PATCH_PGD (K10 fused: whitebox_attacks.py:32-36 / the torchattacks PGD step)
    // A-row gather for the patch-embedding backward: GEMM row m = b*patches + p reads A row
    // b*tokens + 1 + p (the non-CLS rows of the token-major gradient); 0 = off
    int a_gather;
    // algorithmic sizes for profiling (0 = use N / K2): true LoRA rank columns, not the padded ones
    int n_algo, k2_algo;
    int k2_used;          // nonzero columns of the LoRA K tile (r * modules); <= 32 lets the 256-row kernel skip its upper half

    // gemm_pp only (gemm_pp_fuses_down): LoRA down projection t = A1 down_W^T computed inside the GEMM; down_W = [16 * groups
    // rows][K1] h16 (the rows of Ad in use, zero padded), down_out (optional) receives t (ld = down_ld); A2 is then unused
    const h16* down_W; int down_ldw; h16* down_out; int down_ld; int down_groups;
    int ones_col;     // gemm_pp with the down projection inside: column 63 of the LoRA K tile is set to 1 (W2 column 63 = the bias, `bias` null)
    int no_pp;        // 1: never the ping-pong kernel (the Swin stages 3-4 shapes -- M = 50 176 / 12 544 rows, N = 512 .. 3 072 -- run 1.5 % of a step
                      // faster on the 256-row kernel: 17.30 vs 17.56 ms, same box, round 5)
    int tile_group;   // gemm256: tile rows per group of the tile walk (0 = 8 for N >= 2048, else 1; 1 = row-major, the round-1 order)
    int dephase;      // gemm256: start offset unit (x 8128 cycles x (workgroup/8 mod 4)); 0 = off
    // EPI_DROP_ACC: dropout mask of element (m, n) of the module input = drop_scale(seed, stream, m*N + n)
    const h16* G; int ldg;          // optional extra factor (gelu'(z) for the fc2 input gradient)
    uint64_t drop_seed; uint32_t drop_stream; float drop_p, drop_inv_keep;
};

// Algorithmic HBM bytes of `rows` output rows of a GEMM launch (profiling scopes; bench.py's traffic ratio divides the PMC
// counter bytes by this): every operand element read once, every result element written once, in its storage type.
static inline double gemm_algo_bytes(const GemmArgs& a, int epi, double rows) {
    const double K = (double)a.K1 + (a.k2_algo ? a.k2_algo : a.K2);
    const double N = a.n_algo ? a.n_algo : a.N;
    double out = 2.0;                                   // h16 store
    if (epi == EPI_RESID_F32) out = 8.0;                // read + write fp32
    else if (epi == EPI_GELU) out = 4.0;                // gelu(z) and gelu'(z), h16 each
    else if (epi == EPI_GELU_BWD) out = 4.0;            // read the saved gelu'(z), write h16
    else if (epi == EPI_RESID_H16) out = 4.0;           // read the stream (h16), write the stream
    else if (epi == EPI_PATCH_BWD || epi == EPI_STORE_F32) out = 4.0;
    else if (epi == EPI_PATCH_PGD) out = 12.0;           // read adv, x0; write adv
    else if (epi == EPI_NONE) out = 0.0;
    return 2.0 * rows * K + 2.0 * N * K + rows * N * out + (a.down_W ? rows * 64 * 2.0 : 0.0);
}

// bn = 128 (default) or 64 (skinny LoRA-down GEMMs)
void launch_gemm(const GemmArgs& a, int epi, int bn, hipStream_t s);
int gemm_init(int device);
int gemm_force_small(int v);   // 1: every GEMM on the 128-row kernel (self-check reference); returns the previous setting
bool gemm_small_forced();      // that switch (or VITLORA_GEMM128=1) is on: no kernel may claim a fused form the 128-row kernel cannot run
void gemm256_set_cus(int n);   // persistent grid size of the 256-row kernel (default: the device's CU count)
void gemm_pp_set_cus(int n);
int gemm_pp_mode();           // gemm_pp.hip: 0 = off, 1 = every supported shape, 2 = epilogue-heavy shapes only
void gemm_pp_set_mode(int m);
int gemm_stream_set_mode(int mode);    // bit 0: streaming kernel on, bit 1: LoRA down projection inside it; returns the old mode
bool gemm_stream_fuses_down(const GemmArgs& a, int epi);   // gemm_stream.hip: a.down_W = [64 rows][K1] (down_ldw), W2 / K2 = 64 the LoRA K tile, A2 unused
bool gemm_pp_fuses_down(const GemmArgs& a, int epi);   // a.down_* set: can launch_gemm run this GEMM with the down projection inside?   // per-device kernel attributes (outside any stream capture); 0 = ok

}  // namespace VLNS
