// gf_nvgf.hip -- node-variant graph filter (reference NVGF, graphML.py:293-387): the same K-hop tap stack as LSIGF
// (gf_khop, node-major Z[T][B][N][G]) contracted with a filter bank that has its own taps at every node,
//
//     y[b,f,n] = bias[f] + sum_{e,k,g} h[f,e,k,g,n] * (x_g S_e^k)[b,n]
//
// instead of the shared [K*G, F] bank of gf_contract.  Per node this is a [B x T*G] x [T*G x F] product with its own
// right-hand side, so there is no operand reuse across nodes and the kernels are bound by streaming Z (B*T*G floats per
// node) and the bank (T*G*F floats per node) once each; they are plain FMA kernels, one workgroup per node.
//
// The bank is first brought to node-major  Ht[N][T][G][F]  (T = 1 + E(K-1); tap 0 is shared by the edge features,
// graphML.py:371 repeats x for every e, so its weights add up), which makes every operand stream unit-stride:
//     forward   Y [b][n][f]   = sum_{t,g} Z[t][b][n][g] * Ht[n][t][g][f]                    lanes along f
//     backward  dZ[t][b][n][g] = sum_f dY[b][n][f] * Ht[n][t][g][f]                          lanes along (t,g)
//               dHt[n][t][g][f] = sum_b Z[t][b][n][g] * dY[b][n][f]                          lanes along f
//     dX = dZ_0 + sum_e sum_{k>=1} (adjoint hop)^k dZ_{e,k}  in Horner form (K-1 hops per edge feature; the bank does not
//     commute with the shift, so the "hop dY, contract once" shortcut of the LSIGF backward does not apply here).
#include "gf_common.h"

namespace {

constexpr int TT = 32;  // transpose tile

// h[F][C][N] (C = E*K*G, reference layout) -> Ht[N][T*G][F]; tap 0 of every edge feature accumulates into t = 0.
__global__ __launch_bounds__(256) void nv_bank_in_kernel(const float* __restrict__ h, float* __restrict__ Ht, int F, int E, int K,
                                                         int G, int N) {
    __shared__ float tile[TT][TT + 1];
    const int TG = (1 + E * (K - 1)) * G;
    const int ct = blockIdx.z;  // output column c' = t*G + g
    const int t = ct / G, g = ct % G;
    const int n0 = blockIdx.x * TT, f0 = blockIdx.y * TT;
    const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
    for (int i = ty; i < TT; i += 8) {
        const int f = f0 + i, n = n0 + tx;
        float v = 0.f;
        if (f < F && n < N) {
            if (t == 0) {
                for (int e = 0; e < E; ++e) v += h[(((int64_t)f * E + e) * K * G + g) * N + n];
            } else {
                const int e = (t - 1) / (K - 1), k = (t - 1) % (K - 1) + 1;
                v = h[((((int64_t)f * E + e) * K + k) * G + g) * N + n];
            }
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < TT; i += 8) {
        const int n = n0 + i, f = f0 + tx;
        if (n < N && f < F) Ht[((int64_t)n * TG + ct) * F + f] = tile[tx][i];
    }
}

// dHt[N][T*G][F] -> dh[F][E][K][G][N]; the tap-0 gradient is the same for every edge feature.
__global__ __launch_bounds__(256) void nv_bank_out_kernel(const float* __restrict__ dHt, float* __restrict__ dh, int F, int E,
                                                          int K, int G, int N) {
    __shared__ float tile[TT][TT + 1];
    const int TG = (1 + E * (K - 1)) * G;
    const int c = blockIdx.z;  // reference column (e, k, g)
    const int e = c / (K * G), k = (c / G) % K, g = c % G;
    const int t = (k == 0) ? 0 : 1 + e * (K - 1) + (k - 1);
    const int n0 = blockIdx.x * TT, f0 = blockIdx.y * TT;
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int i = ty; i < TT; i += 8) {
        const int n = n0 + i, f = f0 + tx;
        tile[i][tx] = (n < N && f < F) ? dHt[((int64_t)n * TG + t * G + g) * F + f] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < TT; i += 8) {
        const int f = f0 + i, n = n0 + tx;
        if (f < F && n < N) dh[((int64_t)f * E * K * G + c) * N + n] = tile[tx][i];
    }
}

// One workgroup per node.  Thread -> (4 consecutive samples, one f); lanes along f: Ht rows are read unit-stride, the four Z
// values are wave-uniform per f-group (broadcast loads).
template <int BT>
__global__ __launch_bounds__(256) void nv_contract_kernel(const float* __restrict__ Z, const float* __restrict__ Ht,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int B, int N,
                                                          int G, int F, int T) {
    const int n = blockIdx.x;
    const int TG = T * G;
    const float* hn = Ht + (int64_t)n * TG * F;
    const int groups = (B + BT - 1) / BT;
    for (int idx = threadIdx.x; idx < groups * F; idx += blockDim.x) {
        const int f = idx % F, b0 = (idx / F) * BT;
        float acc[BT];
#pragma unroll
        for (int j = 0; j < BT; ++j) acc[j] = 0.f;
        for (int t = 0; t < T; ++t) {
            const float* zt = Z + (((int64_t)t * B + b0) * N + n) * G;
            const float* ht = hn + (int64_t)t * G * F + f;
            for (int g = 0; g < G; ++g) {
                const float w = ht[(int64_t)g * F];
#pragma unroll
                for (int j = 0; j < BT; ++j) {
                    const float z = (b0 + j < B) ? zt[(int64_t)j * N * G + g] : 0.f;
                    acc[j] = fmaf(z, w, acc[j]);
                }
            }
        }
        const float bf = bias ? bias[f] : 0.f;
#pragma unroll
        for (int j = 0; j < BT; ++j)
            if (b0 + j < B) Y[((int64_t)(b0 + j) * N + n) * F + f] = acc[j] + bf;
    }
}

// dZ[t][b][n][g] = sum_f dY[b][n][f] Ht[n][t][g][f]; thread -> (b, c = t*G + g), lanes along c: writes are unit-stride in g.
__global__ __launch_bounds__(256) void nv_dz_kernel(const float* __restrict__ dY, const float* __restrict__ Ht,
                                                    float* __restrict__ dZ, int B, int N, int G, int F, int T) {
    const int n = blockIdx.x;
    const int TG = T * G;
    const float* hn = Ht + (int64_t)n * TG * F;
    for (int idx = threadIdx.x; idx < B * TG; idx += blockDim.x) {
        const int c = idx % TG, b = idx / TG;
        const float* dy = dY + ((int64_t)b * N + n) * F;
        const float* hr = hn + (int64_t)c * F;
        float acc = 0.f;
        for (int f = 0; f < F; ++f) acc = fmaf(dy[f], hr[f], acc);
        const int t = c / G, g = c % G;
        dZ[(((int64_t)t * B + b) * N + n) * G + g] = acc;
    }
}

// dHt[n][c][f] = sum_b Z[t][b][n][g] dY[b][n][f]; thread -> (c, f), lanes along f; fixed summation order over b.
__global__ __launch_bounds__(256) void nv_dbank_kernel(const float* __restrict__ Z, const float* __restrict__ dY,
                                                       float* __restrict__ dHt, int B, int N, int G, int F, int T) {
    const int n = blockIdx.x;
    const int TG = T * G;
    for (int idx = threadIdx.x; idx < TG * F; idx += blockDim.x) {
        const int f = idx % F, c = idx / F;
        const int t = c / G, g = c % G;
        float acc = 0.f;
        for (int b = 0; b < B; ++b)
            acc = fmaf(Z[(((int64_t)t * B + b) * N + n) * G + g], dY[((int64_t)b * N + n) * F + f], acc);
        dHt[((int64_t)n * TG + c) * F + f] = acc;
    }
}

__global__ __launch_bounds__(256) void nv_add_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = i; j < count; j += stride) dst[j] += src[j];
}

int check_plans(const gf_plan* const* plans, int E, const char* who) {
    GF_REQUIRE_ARG(plans && E > 0 && plans[0], "%s: NULL plans", who);
    for (int e = 0; e < E; ++e) {
        GF_REQUIRE_ARG(plans[e] != nullptr, "%s: plan %d is NULL", who, e);
        GF_REQUIRE_SHAPE(plans[e]->n == plans[0]->n, "%s: plan %d has %d nodes, plan 0 has %d", who, e, plans[e]->n, plans[0]->n);
    }
    return GF_OK;
}

}  // namespace

// floats of device scratch gf_nvgf_forward (backward = 0) / gf_nvgf_backward (backward = 1) need
extern "C" size_t gf_nvgf_scratch_floats(int32_t B, int32_t N, int32_t G, int32_t F, int32_t E, int32_t K, int32_t backward) {
    if (B <= 0 || N <= 0 || G <= 0 || F <= 0 || E <= 0 || K <= 0) return 0;
    const size_t T = 1 + (size_t)E * (K - 1);
    const size_t bank = (size_t)N * T * G * F, sig = (size_t)B * N * F;
    if (!backward) return bank + sig;
    return 2 * bank + sig + T * B * N * G + (size_t)B * N * G;
}

extern "C" int gf_nvgf_forward(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias, float* Z,
                               float* y, float* scratch, size_t scratch_floats, int32_t B, int32_t G, int32_t F, int32_t K,
                               int32_t Nin, void* stream) {
    GF_REQUIRE_ARG(x && h && Z && y && scratch, "gf_nvgf_forward: NULL argument");
    int rc = check_plans(plans, E, "gf_nvgf_forward");
    if (rc != GF_OK) return rc;
    const int N = plans[0]->n;
    GF_REQUIRE_SHAPE(B > 0 && G > 0 && F > 0 && K > 0 && Nin > 0 && Nin <= N, "gf_nvgf_forward: bad shape B=%d G=%d F=%d K=%d Nin=%d N=%d",
                     B, G, F, K, Nin, N);
    GF_REQUIRE_SHAPE(scratch_floats >= gf_nvgf_scratch_floats(B, N, G, F, E, K, 0), "gf_nvgf_forward: scratch too small");
    const int T = 1 + E * (K - 1);
    hipStream_t st = gf_stream(stream);
    float* Ht = scratch;
    float* Y = scratch + (size_t)N * T * G * F;
    rc = gf_layout_bgn_to_bng(x, Z, B, G, Nin, N, stream);
    if (rc != GF_OK) return rc;
    rc = gf_khop(plans, E, GF_OP_FWD, Z, B, G, K, stream);
    if (rc != GF_OK) return rc;
    GF_REQUIRE_SHAPE(T * G <= 65535, "gf_nvgf_forward: T*G = %d > 65535", T * G);
    hipLaunchKernelGGL(nv_bank_in_kernel, dim3((N + TT - 1) / TT, (F + TT - 1) / TT, T * G), dim3(TT, 8), 0, st, h, Ht, F, E, K, G, N);
    GF_LAUNCH_CHECK("nv_bank_in_kernel");
    hipLaunchKernelGGL(nv_contract_kernel<4>, dim3(N), dim3(256), 0, st, Z, Ht, bias, Y, B, N, G, F, T);
    GF_LAUNCH_CHECK("nv_contract_kernel");
    return gf_layout_bng_to_bgn(Y, y, B, F, N, Nin, stream);
}

extern "C" int gf_nvgf_backward(const gf_plan* const* plans, int32_t E, const float* dy, const float* Z, const float* h, float* dx,
                                float* dh, float* scratch, size_t scratch_floats, int32_t B, int32_t G, int32_t F, int32_t K,
                                int32_t Nin, void* stream) {
    GF_REQUIRE_ARG(dy && Z && h && scratch, "gf_nvgf_backward: NULL argument");
    int rc = check_plans(plans, E, "gf_nvgf_backward");
    if (rc != GF_OK) return rc;
    const int N = plans[0]->n;
    GF_REQUIRE_SHAPE(B > 0 && G > 0 && F > 0 && K > 0 && Nin > 0 && Nin <= N, "gf_nvgf_backward: bad shape B=%d G=%d F=%d K=%d Nin=%d N=%d",
                     B, G, F, K, Nin, N);
    GF_REQUIRE_SHAPE(scratch_floats >= gf_nvgf_scratch_floats(B, N, G, F, E, K, 1), "gf_nvgf_backward: scratch too small");
    const int T = 1 + E * (K - 1);
    GF_REQUIRE_SHAPE(T * G <= 65535 && E * K * G <= 65535, "gf_nvgf_backward: too many bank columns");
    hipStream_t st = gf_stream(stream);
    const size_t bank = (size_t)N * T * G * F, tap = (size_t)B * N * G;
    float* Ht = scratch;
    float* dHt = Ht + bank;
    float* dY = dHt + bank;
    float* dZ = dY + (size_t)B * N * F;
    float* tmp = dZ + (size_t)T * tap;
    rc = gf_layout_bgn_to_bng(dy, dY, B, F, Nin, N, stream);  // rows >= Nin zero: dropped outputs carry no gradient
    if (rc != GF_OK) return rc;
    if (dh) {
        hipLaunchKernelGGL(nv_dbank_kernel, dim3(N), dim3(256), 0, st, Z, dY, dHt, B, N, G, F, T);
        GF_LAUNCH_CHECK("nv_dbank_kernel");
        hipLaunchKernelGGL(nv_bank_out_kernel, dim3((N + TT - 1) / TT, (F + TT - 1) / TT, E * K * G), dim3(TT, 8), 0, st, dHt, dh,
                           F, E, K, G, N);
        GF_LAUNCH_CHECK("nv_bank_out_kernel");
    }
    if (dx) {
        hipLaunchKernelGGL(nv_bank_in_kernel, dim3((N + TT - 1) / TT, (F + TT - 1) / TT, T * G), dim3(TT, 8), 0, st, h, Ht, F, E, K, G, N);
        GF_LAUNCH_CHECK("nv_bank_in_kernel");
        hipLaunchKernelGGL(nv_dz_kernel, dim3(N), dim3(256), 0, st, dY, Ht, dZ, B, N, G, F, T);
        GF_LAUNCH_CHECK("nv_dz_kernel");
        const int addBlocks = (int)((tap + 255) / 256 < 4096 ? (tap + 255) / 256 : 4096);
        for (int e = 0; e < E; ++e)
            for (int k = K - 1; k >= 1; --k) {  // Horner: fold tap (e,k) into tap (e,k-1) -- tap 0 for k = 1 -- through one adjoint hop
                const float* src = dZ + (size_t)(1 + e * (K - 1) + (k - 1)) * tap;
                float* dst = (k == 1) ? dZ : dZ + (size_t)(1 + e * (K - 1) + (k - 2)) * tap;
                rc = gf_spmm_hop(plans[e], GF_OP_BWD, src, tmp, B, G, stream);
                if (rc != GF_OK) return rc;
                hipLaunchKernelGGL(nv_add_kernel, dim3(addBlocks), dim3(256), 0, st, dst, tmp, (int64_t)tap);
                GF_LAUNCH_CHECK("nv_add_kernel");
            }
        rc = gf_layout_bng_to_bgn(dZ, dx, B, G, N, Nin, stream);
    }
    return rc;
}
