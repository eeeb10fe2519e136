// gf_db.hip -- graph filters whose shift operator differs per sample and time step: the "_DB" (batch + delay) family of the
// reference (LSIGF_DB graphML.py:977-1094, GRNN_DB :1096-1290, GraphFilter_DB :3278-3393, HiddenState_DB :3395-3538) and the
// edge-gated recursion of GatedGRNN (:1394-1419, :1434-1456), where a gate multiplies the GSO entrywise per (b, t).
//
// The reference holds S as a dense [B, T, E, N, N] tensor (flocking: communication-radius graphs of N ~ 50-100 agents that move, so
// every (b, t) has its own operator) and runs, per tap, a time shift (split + cat of a zero row, :1062-1067) and a batched dense
// torch.matmul(x, S) (:1069), then cat / permute / reshape / matmul for the filter bank (:1073-1090).  Here:
//   * signals are node-major [B*T, N, W] (one contiguous feature row per (b, t, n)), taps written in place into the stack
//     Z[tap][B*T][N][W] (tap 0 shared by the edge features, as in gf_khop) -- no cat, no zero rows, the time shift is an index;
//   * one hop of every (b, t) is ONE launch: a workgroup takes (b, t, block of output rows), walks the source nodes in chunks of 32,
//     stages the S tile (read once per hop, coalesced in the orientation the op needs) and the source rows in LDS and accumulates
//     16 bytes per lane; op 0 = x @ S_t with the delayed input (forward), op 1 = its adjoint (backward of the hop);
//   * gradient with respect to S (needed only when the operator is a function of learnable gates: edge gating) is a batched outer
//     product over the feature axis, dS[b,t,m,n] = sum_w X[b,t,m,w] dOut[b,t,n,w];
//   * the filter bank, its gradients and the layouts are the kernels of the static-GSO path (gf_contract, gf_grad_taps,
//     gf_layout_*), run with batch B*T; the adjoint of the contraction per tap (dZ_t = dY H_t^T, needed by the Horner-form
//     backward) is the one new dense kernel.
// S is dense at the boundary because that is what the reference's API hands over; at these sizes (N^2 * 4 bytes = 10-40 KB per
// (b, t)) a tile is an LDS-resident block and the hop is bound by reading S once.  All sums run in a fixed order: deterministic.
#include "gf_common.h"

namespace {

constexpr int kT = 256;    // threads per workgroup
constexpr int kCh = 32;    // source nodes per chunk

// out[bt][r][:] = sum_c A(r, c) in[src(bt)][c][:]      A(r, c) = S[c][r] (op 0) or S[r][c] (op 1)
// thread -> (row r of the block, quad q of the W / 4 quads); RT = kT / (W / 4) rows per workgroup
__global__ __launch_bounds__(kT) void db_hop_kernel(const float* __restrict__ S, int64_t sb, int64_t st, const float* __restrict__ Xin,
                                                    float* __restrict__ Xout, int nt, int N, int W, int op, int shift) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Q = W >> 2, RT = kT / Q;
    float* sS = lds;                       // [kCh][RT + 1]
    float* sX = lds + kCh * (RT + 1);      // [kCh][W]
    const int bt = blockIdx.x, b = bt / nt, t = bt - b * nt;
    const int r0 = blockIdx.y * RT;
    const int tid = threadIdx.x, r = tid / Q, q = tid - r * Q;
    float4* out = reinterpret_cast<float4*>(Xout + ((int64_t)bt * N + (r0 + r)) * W) + q;
    const bool live_row = r < RT && r0 + r < N;
    // op 0: out(t) uses in(t - shift) and S(t);  op 1: out(t) uses in(t + shift) and S(t + shift)
    const int ts = op == 0 ? t - shift : t + shift;
    if (ts < 0 || ts >= nt) {
        if (live_row) *out = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float* Sm = S + (int64_t)b * sb + (int64_t)(op == 0 ? t : ts) * st;
    const float* Xs = Xin + ((int64_t)(b * nt + ts) * N) * W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = 0; c0 < N; c0 += kCh) {
        const int nc = min(kCh, N - c0);
        // S tile: element (i, j) = A(r0 + j, c0 + i)
        for (int e = tid; e < kCh * RT; e += kT) {
            int i, j;
            if (op == 0) { i = e / RT; j = e - i * RT; }       // S[c0 + i][r0 + j]: consecutive threads along j (a row of S)
            else { j = e / kCh; i = e - j * kCh; }             // S[r0 + j][c0 + i]: consecutive threads along i (a row of S)
            float v = 0.f;
            if (i < nc && r0 + j < N) v = op == 0 ? Sm[(int64_t)(c0 + i) * N + (r0 + j)] : Sm[(int64_t)(r0 + j) * N + (c0 + i)];
            sS[i * (RT + 1) + j] = v;
        }
        for (int e = tid; e < kCh * Q; e += kT) {
            const int i = e / Q, qq = e - i * Q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nc) v = reinterpret_cast<const float4*>(Xs + (int64_t)(c0 + i) * W)[qq];
            reinterpret_cast<float4*>(sX)[i * Q + qq] = v;
        }
        __syncthreads();
        if (live_row) {
#pragma unroll 8
            for (int i = 0; i < kCh; ++i) {    // ascending source index: fixed summation order
                const float s = sS[i * (RT + 1) + r];
                const float4 x = reinterpret_cast<const float4*>(sX)[i * Q + q];
                acc.x = fmaf(s, x.x, acc.x);
                acc.y = fmaf(s, x.y, acc.y);
                acc.z = fmaf(s, x.z, acc.z);
                acc.w = fmaf(s, x.w, acc.w);
            }
        }
        __syncthreads();
    }
    if (live_row) *out = acc;
}

// dS[b][t][m][n] (+)= sum_w X[src(bt)][m][w] * dOut[bt][n][w]        32 x 32 tile of (m, n) per workgroup, 4 n per thread
__global__ __launch_bounds__(kT) void db_grad_gso_kernel(const float* __restrict__ X, const float* __restrict__ dOut, float* __restrict__ dS,
                                                         int64_t sb, int64_t st, int nt, int N, int W, int shift, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sA = lds;                   // [32][W + 1]   rows m of X
    float* sB = lds + 32 * (W + 1);    // [32][W + 1]   rows n of dOut
    const int bt = blockIdx.x, b = bt / nt, t = bt - b * nt;
    const int tilesN = (N + 31) / 32;
    const int m0 = (blockIdx.y / tilesN) * 32, n0 = (blockIdx.y % tilesN) * 32;
    const int tid = threadIdx.x, mi = tid >> 3, nj = (tid & 7) * 4;
    const int ts = t - shift;
    float* out = dS + (int64_t)b * sb + (int64_t)t * st;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ts >= 0) {
        const float* Xs = X + ((int64_t)(b * nt + ts) * N) * W;
        const float* Ds = dOut + ((int64_t)bt * N) * W;
        for (int e = tid; e < 32 * W; e += kT) {
            const int i = e / W, w = e - i * W;
            sA[i * (W + 1) + w] = m0 + i < N ? Xs[(int64_t)(m0 + i) * W + w] : 0.f;
            sB[i * (W + 1) + w] = n0 + i < N ? Ds[(int64_t)(n0 + i) * W + w] : 0.f;
        }
        __syncthreads();
        for (int w = 0; w < W; ++w) {
            const float a = sA[mi * (W + 1) + w];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = fmaf(a, sB[(nj + u) * (W + 1) + w], acc[u]);
        }
    }
    if (m0 + mi < N)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (n0 + nj + u < N) {
                float* p = out + (int64_t)(m0 + mi) * N + (n0 + nj + u);
                *p = accumulate ? *p + acc[u] : acc[u];
            }
}

// dZ[t][b][n][g] = sum_f P0[b][n][f] * h[f][e(t)][k(t)][g]        (the adjoint of the filter-bank contraction, per tap)
// tap 0 (k = 0, shared by the edge features) sums h over e.  One thread per (b, n, quad of g); the bank slice of the tap in LDS.
__global__ __launch_bounds__(kT) void stack_adjoint_kernel(const float* __restrict__ P0, const float* __restrict__ h, float* __restrict__ dZ,
                                                           int64_t BN, int G, int F, int E, int K) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [F][G]
    const int tap = blockIdx.y;
    const int e0 = tap == 0 ? 0 : (tap - 1) / (K - 1), k = tap == 0 ? 0 : 1 + (tap - 1) % (K - 1);
    for (int i = threadIdx.x; i < F * G; i += kT) {
        const int f = i / G, g = i - f * G;
        float v = 0.f;
        if (tap == 0)
            for (int e = 0; e < E; ++e) v += h[(((int64_t)f * E + e) * K + 0) * G + g];
        else
            v = h[(((int64_t)f * E + e0) * K + k) * G + g];
        lds[i] = v;
    }
    __syncthreads();
    const int Q = G >> 2;
    const int64_t idx = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (idx >= BN * Q) return;
    const int64_t row = idx / Q;
    const int q = (int)(idx - row * Q);
    const float* p = P0 + row * F;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int f = 0; f < F; ++f) {
        const float s = p[f];
        const float4 hv = reinterpret_cast<const float4*>(lds + f * G)[q];
        acc.x = fmaf(s, hv.x, acc.x);
        acc.y = fmaf(s, hv.y, acc.y);
        acc.z = fmaf(s, hv.z, acc.z);
        acc.w = fmaf(s, hv.w, acc.w);
    }
    reinterpret_cast<float4*>(dZ + ((int64_t)tap * BN + row) * G)[q] = acc;
}

__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        float4 x = reinterpret_cast<float4*>(a)[i];
        const float4 y = reinterpret_cast<const float4*>(b)[i];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        reinterpret_cast<float4*>(a)[i] = x;
    }
}

int check_db(const char* who, int nb, int nt, int N, int W) {
    GF_REQUIRE_SHAPE(nb > 0 && nt > 0 && N > 0 && W > 0, "%s: bad shape nb=%d nt=%d N=%d W=%d", who, nb, nt, N, W);
    GF_REQUIRE_SHAPE(W % 4 == 0 && W <= 256, "%s: row width W = %d must be a multiple of 4, at most 256 (the host layer pads feature counts)", who, W);
    return GF_OK;
}

}  // namespace

extern "C" int gf_db_hop(const float* S, int64_t s_stride_b, int64_t s_stride_t, const float* Xin, float* Xout, int32_t nb, int32_t nt,
                         int32_t N, int32_t W, int32_t op, int32_t shift, void* stream) {
    GF_REQUIRE_ARG(S && Xin && Xout, "gf_db_hop: NULL argument");
    GF_REQUIRE_ARG(op == GF_OP_FWD || op == GF_OP_BWD, "gf_db_hop: op = %d", op);
    GF_REQUIRE_ARG(shift == 0 || shift == 1, "gf_db_hop: shift = %d (0 = same time step, 1 = delayed input)", shift);
    int rc = check_db("gf_db_hop", nb, nt, N, W);
    if (rc != GF_OK) return rc;
    const int Q = W / 4, RT = kT / Q;
    const size_t lds = sizeof(float) * ((size_t)kCh * (RT + 1) + (size_t)kCh * W);
    dim3 grid((unsigned)(nb * nt), (unsigned)((N + RT - 1) / RT));
    hipLaunchKernelGGL(db_hop_kernel, grid, dim3(kT), lds, gf_stream(stream), S, s_stride_b, s_stride_t, Xin, Xout, nt, N, W, op, shift);
    GF_LAUNCH_CHECK("db_hop_kernel");
    return GF_OK;
}

extern "C" int gf_db_grad_gso(const float* Xin, const float* dOut, float* dS, int64_t s_stride_b, int64_t s_stride_t, int32_t nb, int32_t nt,
                              int32_t N, int32_t W, int32_t shift, int32_t accumulate, void* stream) {
    GF_REQUIRE_ARG(Xin && dOut && dS, "gf_db_grad_gso: NULL argument");
    GF_REQUIRE_ARG(shift == 0 || shift == 1, "gf_db_grad_gso: shift = %d", shift);
    int rc = check_db("gf_db_grad_gso", nb, nt, N, W);
    if (rc != GF_OK) return rc;
    const int tiles = (N + 31) / 32;
    const size_t lds = sizeof(float) * 2 * 32 * (size_t)(W + 1);   // 65 792 bytes at the widest supported W = 256: past the 64 KiB default
    GF_HIP(gf_grant_lds((const void*)db_grad_gso_kernel, lds));
    hipLaunchKernelGGL(db_grad_gso_kernel, dim3((unsigned)(nb * nt), (unsigned)(tiles * tiles)), dim3(kT), lds, gf_stream(stream), Xin, dOut, dS,
                       s_stride_b, s_stride_t, nt, N, W, shift, accumulate);
    GF_LAUNCH_CHECK("db_grad_gso_kernel");
    return GF_OK;
}

extern "C" int gf_stack_adjoint(const float* P0, const float* h, float* dZ, int64_t BN, int32_t G, int32_t F, int32_t E, int32_t K, void* stream) {
    GF_REQUIRE_ARG(P0 && h && dZ, "gf_stack_adjoint: NULL argument");
    GF_REQUIRE_SHAPE(BN > 0 && G > 0 && F > 0 && E > 0 && K > 0 && G % 4 == 0, "gf_stack_adjoint: bad shape BN=%lld G=%d F=%d E=%d K=%d (G %% 4 == 0)",
                     (long long)BN, G, F, E, K);
    GF_REQUIRE_SHAPE((size_t)F * G * 4 <= 160 * 1024, "gf_stack_adjoint: bank slice F*G = %d floats exceeds 160 KiB of LDS", F * G);
    GF_HIP(gf_grant_lds((const void*)stack_adjoint_kernel, sizeof(float) * (size_t)F * G));   // H = 256: 256 KB > LDS is refused above, 128 x 128 and 256 x 128 need the grant
    const int T = gf_num_taps(E, K);
    const int64_t items = BN * (G / 4);
    hipLaunchKernelGGL(stack_adjoint_kernel, dim3((unsigned)((items + kT - 1) / kT), (unsigned)T), dim3(kT), sizeof(float) * (size_t)F * G,
                       gf_stream(stream), P0, h, dZ, BN, G, F, E, K);
    GF_LAUNCH_CHECK("stack_adjoint_kernel");
    return GF_OK;
}

// ---- whole layer: LSIGF_DB (graphML.py:977-1094) and its autograd ------------------------------------------------------------------
//   z_0(t) = x(t),  z_k(t) = z_{k-1}(t - shift) S_e(t)  (zero when t - shift < 0),  y(t) = sum_{e,k} z_k^e(t) h[:,e,k,:]^T + b
//   shift = 1: the reference's delayed filter; shift = 0: a filter with a per-(b, t) operator and no delay (edge-gated GSO, :1394-1419).
extern "C" int gf_lsigf_db_forward(const float* S, const float* x, const float* h, const float* bias, float* Z, float* y, int32_t B, int32_t T,
                                   int32_t G, int32_t F, int32_t E, int32_t K, int32_t N, int32_t shift, void* stream) {
    GF_REQUIRE_ARG(S && x && h && Z && y, "gf_lsigf_db_forward: NULL argument");
    GF_REQUIRE_SHAPE(B > 0 && T > 0 && G > 0 && F > 0 && E > 0 && K > 0 && N > 0, "gf_lsigf_db_forward: bad shape B=%d T=%d G=%d F=%d E=%d K=%d N=%d",
                     B, T, G, F, E, K, N);
    const int BT = B * T;
    const int64_t tap = (int64_t)BT * N * G, sb = (int64_t)T * E * N * N, st = (int64_t)E * N * N;
    int rc = gf_layout_bgn_to_bng(x, Z, BT, G, N, N, stream);   // tap 0: x in node-major rows
    if (rc != GF_OK) return rc;
    for (int e = 0; e < E; ++e)
        for (int k = 1; k < K; ++k) {
            const float* src = (k == 1) ? Z : Z + (int64_t)(1 + e * (K - 1) + (k - 2)) * tap;
            float* dst = Z + (int64_t)(1 + e * (K - 1) + (k - 1)) * tap;
            rc = gf_db_hop(S + (int64_t)e * N * N, sb, st, src, dst, B, T, N, G, GF_OP_FWD, shift, stream);
            if (rc != GF_OK) return rc;
        }
    return gf_contract_launch(Z, h, bias, y, BT, N, N, G, F, E, K, 0, gf_stream(stream));
}

// backward in Horner form: g_{K-1} = dZ_{K-1};  g_{k-1} = dZ_{k-1} + adjoint hop of g_k;  dx = sum_e g_0^e (tap 0 is shared);
// dS_e(t) += z_{k-1}(t - shift)^T g_k(t) when dS is requested.  dZ [T_taps, B*T, N, G] scratch (overwritten), P0 [B*T, N, F] scratch.
extern "C" int gf_lsigf_db_backward(const float* S, const float* dy, const float* Z, const float* h, float* P0, float* dZ, float* scratch, float* dx,
                                    float* dh, float* dbias, float* dS, void* workspace, size_t workspace_bytes, int32_t B, int32_t T, int32_t G,
                                    int32_t F, int32_t E, int32_t K, int32_t N, int32_t shift, void* stream) {
    GF_REQUIRE_ARG(S && dy && h && P0, "gf_lsigf_db_backward: NULL argument");
    GF_REQUIRE_SHAPE(B > 0 && T > 0 && G > 0 && F > 0 && E > 0 && K > 0 && N > 0, "gf_lsigf_db_backward: bad shape B=%d T=%d G=%d F=%d E=%d K=%d N=%d",
                     B, T, G, F, E, K, N);
    const int BT = B * T;
    const int64_t tap = (int64_t)BT * N * G, sb = (int64_t)T * E * N * N, st = (int64_t)E * N * N;
    hipStream_t hs = gf_stream(stream);
    int rc = gf_layout_bgn_to_bng(dy, P0, BT, F, N, N, stream);
    if (rc != GF_OK) return rc;
    if (dh || dbias) {
        GF_REQUIRE_ARG(Z != nullptr, "gf_lsigf_db_backward: the saved tap stack Z is required for dh");
        rc = gf_grad_taps(Z, P0, dh, dbias, workspace, workspace_bytes, BT, N, G, F, E, K, stream);
        if (rc != GF_OK) return rc;
    }
    if (!dx && !dS) return GF_OK;
    GF_REQUIRE_ARG(dZ && scratch, "gf_lsigf_db_backward: dZ / scratch are required for dx and dS");
    GF_REQUIRE_ARG(!dS || Z, "gf_lsigf_db_backward: the saved tap stack Z is required for dS");
    rc = gf_stack_adjoint(P0, h, dZ, (int64_t)BT * N, G, F, E, K, stream);
    if (rc != GF_OK) return rc;
    const int64_t n4 = tap / 4;
    const unsigned addBlocks = (unsigned)((n4 + 255) / 256);
    for (int e = 0; e < E; ++e)
        for (int k = K - 1; k >= 1; --k) {
            float* gk = dZ + (int64_t)(1 + e * (K - 1) + (k - 1)) * tap;                      // g_k^e (complete at this point)
            float* gkm1 = (k == 1) ? dZ : dZ + (int64_t)(1 + e * (K - 1) + (k - 2)) * tap;    // dZ_{k-1}: receives the adjoint hop
            if (dS) {
                const float* zkm1 = (k == 1) ? Z : Z + (int64_t)(1 + e * (K - 1) + (k - 2)) * tap;
                rc = gf_db_grad_gso(zkm1, gk, dS + (int64_t)e * N * N, sb, st, B, T, N, G, shift, /*accumulate=*/k != K - 1, stream);
                if (rc != GF_OK) return rc;
            }
            if (dx || k > 1) {
                rc = gf_db_hop(S + (int64_t)e * N * N, sb, st, gk, scratch, B, T, N, G, GF_OP_BWD, shift, stream);
                if (rc != GF_OK) return rc;
                hipLaunchKernelGGL(add_inplace_kernel, dim3(addBlocks), dim3(256), 0, hs, gkm1, scratch, n4);
                GF_LAUNCH_CHECK("add_inplace_kernel");
            }
        }
    if (dx) rc = gf_layout_bng_to_bgn(dZ, dx, BT, G, N, N, stream);   // g_0 (all edge features accumulated into tap 0) back to [B,T,G,N]
    return rc;
}
