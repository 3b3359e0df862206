// Library identification + the generic batched f32-MFMA GEMM entry point.
#include <cstdlib>
#include <cstring>

#include "hk_bgemm.h"
#include "../../include/hawkeye_hip.h"

namespace hk {

namespace {
struct Knob {
    const char* name;
    const char* env;
    int Tuning::*field;
};
const Knob kKnobs[] = {
    {"bcnn_generic", "HK_BCNN_GENERIC", &Tuning::bcnn_generic}, {"cbp_bin", "HK_CBP_BIN", &Tuning::cbp_bin},
    {"roi_bwd", "HK_ROI_BWD", &Tuning::roi_bwd},                {"linear_slabs", "HK_LINEAR_SLABS", &Tuning::linear_slabs},
    {"ns_tn", "HK_NS_TN", &Tuning::ns_tn},                      {"bwd_v", "HK_BWD_V", &Tuning::bwd_v},
    {"ns_streams", "HK_NS_STREAMS", &Tuning::ns_streams},        {"sched_b", "HK_SCHED_B", &Tuning::sched_b},
    {"ns_sym", "HK_NS_SYM", &Tuning::ns_sym},
    {"lin_walk", "HK_LIN_WALK", &Tuning::lin_walk},           {"bwd_fold", "HK_BWD_FOLD", &Tuning::bwd_fold},
    {"fwd_fold", "HK_FWD_FOLD", &Tuning::fwd_fold},
};
Tuning from_env() {
    Tuning t;
    for (const Knob& k : kKnobs)
        if (const char* e = getenv(k.env)) t.*(k.field) = atoi(e);
    return t;
}
}  // namespace

Tuning& tuning() {
    static Tuning t = from_env();
    return t;
}

}  // namespace hk

using namespace hk;

extern "C" int hk_tuning_set(const char* name, int value) {
    if (!name) return HK_ERR_BAD_ARG;
    for (const Knob& k : kKnobs)
        if (!strcmp(name, k.name)) {
            tuning().*(k.field) = value;
            return HK_OK;
        }
    return HK_ERR_BAD_ARG;
}

extern "C" int hk_tuning_get(const char* name, int* value) {
    if (!name || !value) return HK_ERR_BAD_ARG;
    for (const Knob& k : kKnobs)
        if (!strcmp(name, k.name)) {
            *value = tuning().*(k.field);
            return HK_OK;
        }
    return HK_ERR_BAD_ARG;
}

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
