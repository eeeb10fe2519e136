// gf_api.hip -- whole-layer entry points: what GraphFilter.forward (reference graphML.py:2125-2144) and the
// autograd of LSIGF (graphML.py:152-175) amount to, as fixed launch sequences on one stream.
//
//   forward :  x[B,G,Nin] --layout--> Z[0] --(K-1)*E hops, op = S^T--> Z[1..T-1] --contract(h, bias)--> y[B,F,Nin]
//   backward:  dy[B,F,Nin] --layout--> P[0] --(K-1)*E hops, op = S--> P[1..T-1] --contract(h^T)--> dx[B,G,Nin]
//              dh, dbias = grad_taps(Z, P[0])
// The backward data path uses  dx = sum_{e,k} (dY H_{e,k}^T) (S_e^T)^k = sum_{e,k} ((S_e)^k-hop of dY) H_{e,k}^T :
// hopping the F-wide dY and contracting once costs 2(K-1)+K+1 signal passes instead of the K+3(K-1) of the
// Horner form on dZ, and reuses the forward kernels unchanged.
//
// Two interchangeable pipelines expose the SAME INTERFACE (x, dy, y, dx in the reference layout; Z / P opaque).  Their floating-point
// summation orders differ (different ELL images / neighbour orders), and within the panel pipeline the chain kernel and the per-hop
// kernel (chosen from the panel count B * W / 4, see use_chain) differ too: results are bitwise reproducible for a given
// (shape, batch size), and agree across pipelines / batch sizes to fp32 rounding (tests compare those with the stated tolerance):
//   node-major  Z[T][B][N][G]        gathers served by L2            (gf_spmm.hip)     any N, any widths
//   panels      Z[T][B*G/4][N][4]    gathers served by LDS           (gf_panel.hip)    N <= 10239, G and F in {8,16,32,64,128}
// gf_lsigf_pipeline() tells which one a (plans, G, F, K) combination runs; forward and backward always agree because the rule
// depends only on those arguments (the experiment knob "pipeline" is refused outside GFHIP_EXPERIMENTS=1 processes).
#include "gf_common.h"

namespace {

int pick_pipeline(const gf_plan* const* plans, int E, int G, int F, int K) {  // 1 = node-major, 2 = panels, < 0 = error
    const bool ok = gf_panel_supported(plans, E, G, F, K);
    if (g_tune.pipeline == 2 && !ok) {
        gf_set_error("pipeline 2 (column panels) forced but unsupported here: needs N in [8, %d], G and F in {8, 16, 32, 64, 128}, "
                     "and the filter bank (T * G * F floats, both orientations) in LDS", kPanelMaxNodes);
        return GF_ERR_UNSUPPORTED;
    }
    if (g_tune.pipeline == 1) return 1;
    return ok ? 2 : 1;
}

bool use_chain(const gf_plan* plan, int op, int nPanels) {
    const gf_csr_dev& mm = plan->mat[op];
    const bool weighted_small = !(mm.pn_uniform && g_tune.panel_uniform) && mm.cn_np == 2;
    return gf_chain_available(plan, op) && (g_tune.panel_chain == 2 || (g_tune.panel_chain == 1 && nPanels * 4 > 256 && !weighted_small));
}

int khop_panel(const gf_plan* const* plans, int E, int op, float* Zp, int B, int W, int K, hipStream_t st) {
    const int64_t tap = (int64_t)B * plans[0]->n * W;
    const int nPanels = B * (W / 4);
    if (K < 2) return GF_OK;
    for (int e = 0; e < E; ++e) {
        float* first = Zp + (int64_t)(1 + e * (K - 1)) * tap;  // taps 1 + e(K-1) ... of this edge feature are consecutive
        // The K-1 hops of a panel inside LDS (one launch per edge feature) -- unless
        //   * there are so few panels that most CUs would idle: the per-hop kernel can put several workgroups on one panel, the
        //     chain cannot (its panel lives in one CU's LDS);
        //   * the GSO is weighted and small (two panels fit the LDS): the value stream is 4x the column stream and the per-hop
        //     kernel, which shares it between two panels with less bookkeeping, is faster there (N = 1682, config 3: 126 vs 170 us
        //     per chain of 2048 panels; from N ~ 5000 on the two are equal and the chain needs no start-stagger tuning).
        if (use_chain(plans[e], op, nPanels)) {
            const int rc = gf_spmm_chain_launch(plans[e], op, Zp, first, nPanels, K - 1, tap, st);
            if (rc != GF_OK) return rc;
            continue;
        }
        for (int k = 1; k < K; ++k) {
            const float* src = (k == 1) ? Zp : first + (int64_t)(k - 2) * tap;
            const int rc = gf_spmm_panel_launch(plans[e], op, src, first + (int64_t)(k - 1) * tap, nPanels, st);
            if (rc != GF_OK) return rc;
        }
    }
    return GF_OK;
}

int check_khop_panel_args(const gf_plan* const* plans, int32_t E, int32_t op, const float* Zp, int32_t B, int32_t W, int32_t K) {
    GF_REQUIRE_ARG(plans && Zp, "gf_khop_panel: NULL argument");
    GF_REQUIRE_ARG(op == GF_OP_FWD || op == GF_OP_BWD, "gf_khop_panel: op = %d", op);
    GF_REQUIRE_SHAPE(E > 0 && K > 0 && B > 0 && W > 0 && W % 4 == 0, "gf_khop_panel: bad shape E=%d K=%d B=%d W=%d (W %% 4 == 0)", E, K, B, W);
    for (int e = 0; e < E; ++e) {
        GF_REQUIRE_ARG(plans[e] != nullptr, "gf_khop_panel: plan %d is NULL", e);
        GF_REQUIRE_SHAPE(plans[e]->n == plans[0]->n, "gf_khop_panel: plan %d has %d nodes, plan 0 has %d", e, plans[e]->n, plans[0]->n);
        GF_REQUIRE_SHAPE(plans[e]->n <= kPanelMaxNodes && plans[e]->mat[op].pn_slices > 0,
                         "gf_khop_panel: N = %d has no panel image (limit %d)", plans[e]->n, kPanelMaxNodes);
    }
    return GF_OK;
}

}  // namespace

extern "C" int gf_lsigf_pipeline(const gf_plan* const* plans, int32_t E, int32_t G, int32_t F, int32_t K) {
    GF_REQUIRE_ARG(plans && E > 0 && plans[0], "gf_lsigf_pipeline: NULL plans");
    GF_REQUIRE_SHAPE(K > 0, "gf_lsigf_pipeline: K = %d", K);
    return pick_pipeline(plans, E, G, F, K);
}

extern "C" int gf_khop_panel_uses_chain(const gf_plan* plan, int32_t op, int32_t n_panels) {
    GF_REQUIRE_ARG(plan != nullptr && (op == GF_OP_FWD || op == GF_OP_BWD) && n_panels > 0, "gf_khop_panel_uses_chain: bad argument");
    return use_chain(plan, op, n_panels) ? 1 : (gf_panel_db_applies(plan, op, n_panels) ? 2 : 0);
}

extern "C" int gf_khop_panel(const gf_plan* const* plans, int32_t E, int32_t op, float* Zp, int32_t B, int32_t W, int32_t K, void* stream) {
    const int rc = check_khop_panel_args(plans, E, op, Zp, B, W, K);
    return rc != GF_OK ? rc : khop_panel(plans, E, op, Zp, B, W, K, gf_stream(stream));
}

extern "C" int gf_time_khop_panel(const gf_plan* const* plans, int32_t E, int32_t op, float* Zp, int32_t B, int32_t W, int32_t K,
                                  int32_t iters, void* stream, float* avg_ms) {
    GF_REQUIRE_ARG(avg_ms && iters > 0, "gf_time_khop_panel: bad iters / NULL avg_ms");
    int rc = check_khop_panel_args(plans, E, op, Zp, B, W, K);
    if (rc != GF_OK) return rc;
    hipStream_t st = gf_stream(stream);
    hipEvent_t e0, e1;
    GF_HIP(hipEventCreate(&e0));
    GF_HIP(hipEventCreate(&e1));
    rc = khop_panel(plans, E, op, Zp, B, W, K, st);  // warm-up
    if (rc == GF_OK) {
        GF_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < iters && rc == GF_OK; ++i) rc = khop_panel(plans, E, op, Zp, B, W, K, st);
        GF_HIP(hipEventRecord(e1, st));
        GF_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        GF_HIP(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// lx: layer-to-layer hand-over in the internal layout (gf_lsigf_forward_ex / gf_lsigf_backward_ex): bit 1 = tap 0 of the stack already
// holds the input in the pipeline's layout -- column panels, or node-major rows [B][N][G] on the node-major pipeline -- (the previous
// call wrote it there: no pack / transpose pass), bit 2 = the result goes out in that layout.
static int lsigf_forward_impl(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias, float* Z,
                              float* y, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream, int relu, int lx = 0) {
    GF_REQUIRE_ARG(plans && (x || (lx & 2)) && h && Z && y, "gf_lsigf_forward: NULL argument");
    GF_REQUIRE_SHAPE(E > 0 && B > 0 && G > 0 && F > 0 && K > 0 && Nin > 0, "gf_lsigf_forward: bad shape E=%d B=%d G=%d F=%d K=%d Nin=%d",
                     E, B, G, F, K, Nin);
    GF_REQUIRE_ARG(plans[0] != nullptr, "gf_lsigf_forward: plan 0 is NULL");
    const int N = plans[0]->n;
    GF_REQUIRE_SHAPE(Nin <= N, "gf_lsigf_forward: input has %d nodes, GSO has %d", Nin, N);  // graphML.py:2131 only pads
    for (int e = 0; e < E; ++e) {
        GF_REQUIRE_ARG(plans[e] != nullptr, "gf_lsigf_forward: plan %d is NULL", e);
        GF_REQUIRE_SHAPE(plans[e]->n == N, "gf_lsigf_forward: plan %d has %d nodes, plan 0 has %d", e, plans[e]->n, N);
    }
    const int pipe = pick_pipeline(plans, E, G, F, K);
    if (pipe < 0) return pipe;
    if ((lx & 6) && Nin != N) {
        gf_set_error("gf_lsigf_forward_ex: the hand-over in the internal layout needs Nin == N (Nin %d, N %d)", Nin, N);
        return GF_ERR_UNSUPPORTED;
    }
    if ((lx & 6) && pipe != 2 && (G % 4 != 0 || F % 4 != 0)) {
        gf_set_error("gf_lsigf_forward_ex: the hand-over of node-major rows needs widths that are multiples of 4 (G %d, F %d)", G, F);
        return GF_ERR_UNSUPPORTED;
    }
    if (pipe == 2) {
        int rc = (lx & 2) ? GF_OK : gf_pack_panels_launch(x, Z, B, G, Nin, N, gf_stream(stream), nullptr);
        if (rc != GF_OK) return rc;
        rc = khop_panel(plans, E, GF_OP_FWD, Z, B, G, K, gf_stream(stream));
        if (rc != GF_OK) return rc;
        return gf_contract_panel_launch(Z, h, bias, y, B, N, Nin, G, F, E, K, /*transpose_bank | relu << 1=*/relu << 1, gf_stream(stream),
                                        (lx & 4) ? 1 : 0, nullptr);
    }
    // large graphs, 32 columns, one edge feature: the layout pass rides in the fused chain launch (each XCD writes an entry's tap 0 right before it
    // walks the entry: gf_msweep.hip); everything else: layout kernel, then the hops
    int rc = (!(lx & 2) && E == 1) ? gf_khop_with_layout(plans[0], GF_OP_FWD, x, nullptr, Z, B, G, K, Nin, gf_stream(stream)) : GF_ERR_UNSUPPORTED;
    if (rc == GF_ERR_UNSUPPORTED) {
        rc = (lx & 2) ? GF_OK : gf_layout_bgn_to_bng(x, Z, B, G, Nin, N, stream);
        if (rc != GF_OK) return rc;
        rc = gf_khop(plans, E, GF_OP_FWD, Z, B, G, K, stream);
    }
    if (rc != GF_OK) return rc;
    return gf_contract_launch(Z, h, bias, y, B, N, Nin, G, F, E, K, /*transpose_bank | relu << 1=*/relu << 1, gf_stream(stream),
                              (lx & 4) ? 1 : 0, nullptr);
}

static int lsigf_backward_impl(const gf_plan* const* plans, int32_t E, const float* dy, const float* Z, const float* h, float* P,
                               float* dx, float* dh, float* dbias, void* workspace, size_t workspace_bytes, int32_t B, int32_t G,
                               int32_t F, int32_t K, int32_t Nin, void* stream, const float* y_relu, int lx = 0, const float* dx_mask = nullptr) {
    GF_REQUIRE_ARG(plans && (dy || (lx & 2)) && h && P, "gf_lsigf_backward: NULL argument");
    GF_REQUIRE_SHAPE(E > 0 && B > 0 && G > 0 && F > 0 && K > 0 && Nin > 0, "gf_lsigf_backward: bad shape E=%d B=%d G=%d F=%d K=%d Nin=%d",
                     E, B, G, F, K, Nin);
    GF_REQUIRE_ARG(plans[0] != nullptr, "gf_lsigf_backward: plan 0 is NULL");
    const int N = plans[0]->n;
    GF_REQUIRE_SHAPE(Nin <= N, "gf_lsigf_backward: gradient has %d nodes, GSO has %d", Nin, N);
    for (int e = 0; e < E; ++e) {
        GF_REQUIRE_ARG(plans[e] != nullptr, "gf_lsigf_backward: plan %d is NULL", e);
        GF_REQUIRE_SHAPE(plans[e]->n == N, "gf_lsigf_backward: plan %d has %d nodes, plan 0 has %d", e, plans[e]->n, N);
    }
    const int pipe = pick_pipeline(plans, E, G, F, K);
    if (pipe < 0) return pipe;
    if ((lx & 6) && Nin != N) {
        gf_set_error("gf_lsigf_backward_ex: the hand-over in the internal layout needs Nin == N (Nin %d, N %d)", Nin, N);
        return GF_ERR_UNSUPPORTED;
    }
    if ((lx & 6) && pipe != 2 && (G % 4 != 0 || F % 4 != 0)) {
        gf_set_error("gf_lsigf_backward_ex: the hand-over of node-major rows needs widths that are multiples of 4 (G %d, F %d)", G, F);
        return GF_ERR_UNSUPPORTED;
    }
    if (pipe == 2) {
        // P[0] = dy (masked by y > 0) as panels, rows >= Nin zero -- unless the next layer's backward already wrote it there (lx bit 1)
        int rc = (lx & 2) ? GF_OK : gf_pack_panels_launch(dy, P, B, F, Nin, N, gf_stream(stream), y_relu);
        if (rc != GF_OK) return rc;
        const int dxp = (lx & 4) ? 1 : 0;
        if (dx && dh && g_tune.bwd_fuse && gf_bwd_fused_supported(G, F, E, K)) {
            // dh_t = X0^T P_t: the tap gradient reads the adjoint stack the data path builds anyway (and tap 0 of the saved stack)
            GF_REQUIRE_ARG(Z != nullptr, "gf_lsigf_backward: the saved tap stack Z is required for dh");
            rc = khop_panel(plans, E, GF_OP_BWD, P, B, F, K, gf_stream(stream));
            if (rc != GF_OK) return rc;
            return gf_bwd_fused_panel_launch(P, Z, h, dx, dh, dbias, workspace, workspace_bytes, B, N, Nin, G, F, E, K, gf_stream(stream),
                                             /*node_major=*/0, dxp, dx_mask);
        }
        if (dh || dbias) {
            GF_REQUIRE_ARG(Z != nullptr, "gf_lsigf_backward: the saved tap stack Z is required for dh");
            rc = gf_grad_taps_panel(Z, P, dh, dbias, workspace, workspace_bytes, B, N, G, F, E, K, stream);
            if (rc != GF_OK) return rc;
        }
        if (dx) {
            rc = khop_panel(plans, E, GF_OP_BWD, P, B, F, K, gf_stream(stream));
            if (rc != GF_OK) return rc;
            rc = gf_contract_panel_launch(P, h, nullptr, dx, B, N, Nin, G, F, E, K, /*transpose_bank=*/1, gf_stream(stream), dxp, dx_mask);
        }
        return rc;
    }
    // P[0] = dy, node-major, rows >= Nin zero -- unless the next layer's backward already wrote it there (lx bit 1)
    // (as in the forward: where the adjoint chain is one fused sweep launch, dy's layout pass -- and the ReLU mask -- ride in it)
    const bool chain_first = dx && ((dh && g_tune.bwd_fuse && gf_bwd_fused_supported(G, F, E, K)) || !(dh || dbias));   // the hops are the first consumer of P[0]
    int rc = (!(lx & 2) && E == 1 && chain_first) ? gf_khop_with_layout(plans[0], GF_OP_BWD, dy, y_relu, P, B, F, K, Nin, gf_stream(stream)) : GF_ERR_UNSUPPORTED;
    const bool hopped = rc == GF_OK;
    if (rc == GF_ERR_UNSUPPORTED)
        rc = (lx & 2) ? GF_OK
             : y_relu ? gf_layout_masked_launch(dy, y_relu, P, B, F, Nin, N, gf_stream(stream))
                      : gf_layout_bgn_to_bng(dy, P, B, F, Nin, N, stream);
    if (rc != GF_OK) return rc;
    const int dxr = (lx & 4) ? 1 : 0;
    if (dx && dh && g_tune.bwd_fuse && gf_bwd_fused_supported(G, F, E, K)) {
        // one pass over the adjoint stack for dx and dh (dh_t = X0^T P_t), as in the panel pipeline: the separate tap-gradient kernel
        // re-reads the whole forward stack (8.2 GB at config 4)
        GF_REQUIRE_ARG(Z != nullptr, "gf_lsigf_backward: the saved tap stack Z is required for dh");
        rc = hopped ? GF_OK : gf_khop(plans, E, GF_OP_BWD, P, B, F, K, stream);
        if (rc != GF_OK) return rc;
        return gf_bwd_fused_panel_launch(P, Z, h, dx, dh, dbias, workspace, workspace_bytes, B, N, Nin, G, F, E, K, gf_stream(stream),
                                         /*node_major=*/1, dxr, dx_mask);
    }
    if (dh || dbias) {
        GF_REQUIRE_ARG(Z != nullptr, "gf_lsigf_backward: the saved tap stack Z is required for dh");
        rc = gf_grad_taps(Z, P, dh, dbias, workspace, workspace_bytes, B, N, G, F, E, K, stream);
        if (rc != GF_OK) return rc;
    }
    if (dx) {
        rc = hopped ? GF_OK : gf_khop(plans, E, GF_OP_BWD, P, B, F, K, stream);
        if (rc != GF_OK) return rc;
        rc = gf_contract_launch(P, h, nullptr, dx, B, N, Nin, G, F, E, K, /*transpose_bank=*/1, gf_stream(stream), dxr, dx_mask);
    }
    return rc;
}

extern "C" int gf_lsigf_forward(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias,
                                float* Z, float* y, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream) {
    return lsigf_forward_impl(plans, E, x, h, bias, Z, y, B, G, F, K, Nin, stream, 0);
}

extern "C" int gf_lsigf_backward(const gf_plan* const* plans, int32_t E, const float* dy, const float* Z, const float* h,
                                 float* P, float* dx, float* dh, float* dbias, void* workspace, size_t workspace_bytes,
                                 int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream) {
    return lsigf_backward_impl(plans, E, dy, Z, h, P, dx, dh, dbias, workspace, workspace_bytes, B, G, F, K, Nin, stream, nullptr);
}

// The filter followed by the nonlinearity of SelectionGNN's layers (architectures.py:286-289: GFL = [GraphFilter, sigma, rho]),
// for sigma = ReLU: y = max(0, LSIGF(...)) in the contraction's epilogue, and in backward the mask (y > 0) applied while dy is
// brought into the internal layout -- torch's separate relu / threshold_backward passes (5 signal passes) disappear.
extern "C" int gf_lsigf_forward_relu(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias,
                                     float* Z, float* y, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream) {
    return lsigf_forward_impl(plans, E, x, h, bias, Z, y, B, G, F, K, Nin, stream, 1);
}

extern "C" int gf_lsigf_backward_relu(const gf_plan* const* plans, int32_t E, const float* dy, const float* y, const float* Z,
                                      const float* h, float* P, float* dx, float* dh, float* dbias, void* workspace,
                                      size_t workspace_bytes, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream) {
    GF_REQUIRE_ARG(y != nullptr, "gf_lsigf_backward_relu: y (the saved forward output) is NULL");
    return lsigf_backward_impl(plans, E, dy, Z, h, P, dx, dh, dbias, workspace, workspace_bytes, B, G, F, K, Nin, stream, y);
}

// Consecutive filter layers on the same graph (SelectionGNN with NoPool, architectures.py:286-294: GFL = [filter, sigma, rho, filter, ...])
// hand their signals over in the internal layout (column panels, or node-major rows for graphs beyond the LDS panel limit): layer l's
// contraction writes sigma(y_l) straight into tap 0 of layer l+1's stack (no reference-layout round trip: one unpack-transpose in the epilogue and one pack pass less per boundary), and in
// the backward layer l+1 writes dx, masked by sigma'(y_l) = [tap 0 of its own stack > 0], straight into tap 0 of layer l's adjoint stack.
extern "C" int gf_lsigf_forward_ex(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias, float* Z, float* y,
                                   int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, int32_t flags, void* stream) {
    GF_REQUIRE_ARG((flags & ~7) == 0, "gf_lsigf_forward_ex: flags = %d", flags);
    return lsigf_forward_impl(plans, E, x, h, bias, Z, y, B, G, F, K, Nin, stream, flags & 1, flags & 6);
}

extern "C" int gf_lsigf_backward_ex(const gf_plan* const* plans, int32_t E, const float* dy, const float* y_relu, const float* Z, const float* h,
                                    float* P, float* dx, float* dh, float* dbias, void* workspace, size_t workspace_bytes, int32_t B, int32_t G,
                                    int32_t F, int32_t K, int32_t Nin, int32_t flags, const float* dx_mask, void* stream) {
    GF_REQUIRE_ARG((flags & ~6) == 0, "gf_lsigf_backward_ex: flags = %d", flags);
    GF_REQUIRE_ARG(!(flags & 2) || y_relu == nullptr, "gf_lsigf_backward_ex: a handed-over gradient is already masked (y_relu must be NULL)");
    GF_REQUIRE_ARG(dx_mask == nullptr || (flags & 4), "gf_lsigf_backward_ex: dx_mask is only defined for a dx in the internal layout");
    return lsigf_backward_impl(plans, E, dy, Z, h, P, dx, dh, dbias, workspace, workspace_bytes, B, G, F, K, Nin, stream, y_relu, flags & 6, dx_mask);
}
