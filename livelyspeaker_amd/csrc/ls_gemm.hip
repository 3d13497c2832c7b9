// nn.Linear-shaped products of the once-per-call stage and the SAG decoder:
//     C[m][n] = act( sum_k A[m][k] * W[n][k] + bias[n] ) (+ R[m][n])          (y = x W^T + b)
// They run on the strided fp32 MFMA GEMM of ls_train_gemm.hip (both operands K-contiguous): 128x128 tiles of
// v_mfma_f32_16x16x4_f32, register-prefetch software pipeline, and -- for full tiles with float4-legal operands, which is every
// SAG-decoder product at the callers' batch -- its fast staging path.
#include "ls_internal.h"
#include "ls_train.h"

namespace ls {

// act: 0 none, 1 SiLU, 2 exp(0.5 y), 3 exact GELU (F.gelu default, nn.TransformerDecoderLayer activation="gelu")
hipError_t launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr,
                          float* C, int ldc, int M, int N, int K, int act, hipStream_t st) {
    if (R && ldr != ldc) return hipErrorInvalidValue;           // the residual shares C's addressing
    GemmArgs a{};
    a.A = op_rows(A, lda, M, K);
    a.B = op_rows(W, ldw, N, K);
    a.C = C;
    a.cri = INT_MAX; a.cro = 0; a.crs = ldc; a.cns = 1;
    a.bias = bias; a.R = R; a.act = act;
    a.M = M; a.N = N; a.K = K;
    return launch_gemm_tr(a, true, true, 1, st);
}

}  // namespace ls
