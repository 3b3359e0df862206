// Library identification + the generic batched f32-MFMA GEMM entry point.
#include "hk_bgemm.h"
#include "../../include/hawkeye_hip.h"

using namespace hk;

extern "C" const char* hk_version(void) { return "hawkeye_hip 0.1 gfx950"; }

extern "C" int hk_bgemm_f32(const float* a, int lda, long long stride_a, int trans_a, const float* b, int ldb,
                            long long stride_b, int trans_b, float* c, int ldc, long long stride_c, int M, int N,
                            int K, int batch, float alpha, float beta, float diag, hk_stream_t stream) {
    if (!a || !b || !c) return HK_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    // memory views: A is [M][K] (k contiguous) or, transposed, [K][M]; B is [K][N] or, transposed, [N][K]
    const LdPlain la = trans_a ? make_plain(a, stride_a, lda, K, M) : make_plain(a, stride_a, lda, M, K);
    const LdPlain lb = trans_b ? make_plain(b, stride_b, ldb, N, K) : make_plain(b, stride_b, ldb, K, N);
    const EpAffine ep = make_affine(c, stride_c, ldc, alpha, nullptr, beta, diag);
    if (!trans_a && !trans_b) return bgemm_launch<true, false>(la, lb, ep, M, N, K, batch, st);
    if (!trans_a && trans_b) return bgemm_launch<true, true>(la, lb, ep, M, N, K, batch, st);
    if (trans_a && !trans_b) return bgemm_launch<false, false>(la, lb, ep, M, N, K, batch, st);
    return bgemm_launch<false, true>(la, lb, ep, M, N, K, batch, st);
}
