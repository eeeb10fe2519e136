// gf_api.hip -- whole-layer entry points: what GraphFilter.forward (reference graphML.py:2125-2144) and the
// autograd of LSIGF (graphML.py:152-175) amount to, as fixed launch sequences on one stream.
//
//   forward :  x[B,G,Nin] --layout--> Z[0] --(K-1)*E hops, op = S^T--> Z[1..T-1] --contract(h, bias)--> y[B,F,Nin]
//   backward:  dy[B,F,Nin] --layout--> P[0] --(K-1)*E hops, op = S--> P[1..T-1] --contract(h^T)--> dx[B,G,Nin]
//              dh, dbias = grad_taps(Z, P[0])
// The backward data path uses  dx = sum_{e,k} (dY H_{e,k}^T) (S_e^T)^k = sum_{e,k} ((S_e)^k-hop of dY) H_{e,k}^T :
// hopping the F-wide dY and contracting once costs 2(K-1)+K+1 signal passes instead of the K+3(K-1) of the
// Horner form on dZ, and reuses the forward kernels unchanged.
#include "gf_common.h"

extern "C" int gf_lsigf_forward(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias,
                                float* Z, float* y, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream) {
    GF_REQUIRE_ARG(plans && x && h && Z && y, "gf_lsigf_forward: NULL argument");
    GF_REQUIRE_SHAPE(E > 0 && B > 0 && G > 0 && F > 0 && K > 0 && Nin > 0, "gf_lsigf_forward: bad shape E=%d B=%d G=%d F=%d K=%d Nin=%d",
                     E, B, G, F, K, Nin);
    GF_REQUIRE_ARG(plans[0] != nullptr, "gf_lsigf_forward: plan 0 is NULL");
    const int N = plans[0]->n;
    GF_REQUIRE_SHAPE(Nin <= N, "gf_lsigf_forward: input has %d nodes, GSO has %d", Nin, N);  // graphML.py:2131 only pads
    int rc = gf_layout_bgn_to_bng(x, Z, B, G, Nin, N, stream);
    if (rc != GF_OK) return rc;
    rc = gf_khop(plans, E, GF_OP_FWD, Z, B, G, K, stream);
    if (rc != GF_OK) return rc;
    return gf_contract_launch(Z, h, bias, y, B, N, Nin, G, F, E, K, /*transpose_bank=*/0, gf_stream(stream));
}

extern "C" int gf_lsigf_backward(const gf_plan* const* plans, int32_t E, const float* dy, const float* Z, const float* h,
                                 float* P, float* dx, float* dh, float* dbias, void* workspace, size_t workspace_bytes,
                                 int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream) {
    GF_REQUIRE_ARG(plans && dy && h && P, "gf_lsigf_backward: NULL argument");
    GF_REQUIRE_SHAPE(E > 0 && B > 0 && G > 0 && F > 0 && K > 0 && Nin > 0, "gf_lsigf_backward: bad shape E=%d B=%d G=%d F=%d K=%d Nin=%d",
                     E, B, G, F, K, Nin);
    GF_REQUIRE_ARG(plans[0] != nullptr, "gf_lsigf_backward: plan 0 is NULL");
    const int N = plans[0]->n;
    GF_REQUIRE_SHAPE(Nin <= N, "gf_lsigf_backward: gradient has %d nodes, GSO has %d", Nin, N);
    int rc = gf_layout_bgn_to_bng(dy, P, B, F, Nin, N, stream);  // P[0] = dy, node-major, rows >= Nin zero
    if (rc != GF_OK) return rc;
    if (dh || dbias) {
        GF_REQUIRE_ARG(Z != nullptr, "gf_lsigf_backward: the saved tap stack Z is required for dh");
        rc = gf_grad_taps(Z, P, dh, dbias, workspace, workspace_bytes, B, N, G, F, E, K, stream);
        if (rc != GF_OK) return rc;
    }
    if (dx) {
        rc = gf_khop(plans, E, GF_OP_BWD, P, B, F, K, stream);
        if (rc != GF_OK) return rc;
        rc = gf_contract_launch(P, h, nullptr, dx, B, N, Nin, G, F, E, K, /*transpose_bank=*/1, gf_stream(stream));
    }
    return rc;
}
