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

// ---- the three per-node products, operands staged in LDS ------------------------------------------------------------
// For node n let A = Z_n [B x TG] (A[b][c] = Z[t][b][n][g], c = t*G + g), H = Ht_n [TG x F], D = dY_n [B x F]:
//     forward   Y  = A H          backward   dZ_n = D H^T          dHt_n = A^T D
// One workgroup (256 threads) per (node, column chunk): H and the A / D rows of a batch chunk are brought to LDS with
// unit-stride reads (a Z row is G contiguous floats, a dY row F, the bank chunk CC*F), the FMAs then read LDS only:
// one operand is a wave-wide broadcast, the other is unit-stride (rows padded by one float where lanes walk down a column).
constexpr int NV_BC = 32;     // samples per batch chunk

// samples per batch chunk of nv_dz_kernel: its D tile [chunk][F] stays within 32 KiB of LDS
__host__ __device__ inline int nv_dz_batch(int F) {
    int b = (8192 / F) & ~3;
    return b < 4 ? 4 : (b > 128 ? 128 : b);
}

struct NvShape {
    int B, N, G, F, T, CC;    // CC = bank columns (t,g) per chunk
};

__device__ __forceinline__ const float* nv_zrow(const float* Z, const NvShape& s, int n, int c, int b) {
    const int t = c / s.G, g = c - t * s.G;
    return Z + (((int64_t)t * s.B + b) * s.N + n) * s.G + g;
}

// Every thread owns a 4 x 4 register tile of the output: per reduction step it reads 4 + 4 operands from LDS for 16 FMAs.
// Lanes are consecutive along the unit-stride operand (f for H / D rows, c for the padded H^T rows), the other operand is
// a broadcast within the lanes that share the tile row.

// Y[b][n][f] = bias[f] + sum_c A[b][c] H[c][f].  grid = N; the workgroup walks the column chunks itself (the sum runs over them).
// thread -> (bq, fq): samples b0 + 4 bq + i, features fq + k FQ (FQ = ceil(F / 4)); batch chunk = 4 * (256 / FQ) samples, at most 128.
__global__ __launch_bounds__(256) void nv_contract_kernel(const float* __restrict__ Z, const float* __restrict__ Ht,
                                                          const float* __restrict__ bias, float* __restrict__ Y, NvShape s) {
    extern __shared__ float sm[];
    const int TG = s.T * s.G, F = s.F, CC = s.CC;
    const int FQ = (F + 3) / 4;
    const int BCH = min(128, 4 * (256 / FQ));
    float* Hs = sm;                      // [CC][F]
    float* As = sm + (size_t)CC * F;     // [BCH][CC + 1]
    const int n = blockIdx.x, tid = threadIdx.x;
    const int fq = tid % FQ, bq = tid / FQ;
    const float* hn = Ht + (int64_t)n * TG * F;
    int fk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) fk[k] = min(fq + k * FQ, F - 1);
    for (int b0 = 0; b0 < s.B; b0 += BCH) {
        const int nb = min(BCH, s.B - b0);
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
        const bool active = 4 * bq < nb;
        for (int c0 = 0; c0 < TG; c0 += CC) {
            const int nc = min(CC, TG - c0);
            __syncthreads();
            for (int i = tid; i < nc * F; i += 256) Hs[i] = hn[(int64_t)c0 * F + i];
            for (int i = tid; i < nb * nc; i += 256) {
                const int b = i / nc, c = i - b * nc;
                As[b * (CC + 1) + c] = *nv_zrow(Z, s, n, c0 + c, b0 + b);
            }
            __syncthreads();
            if (active) {
                const float* a0 = As + min(4 * bq + 0, nb - 1) * (CC + 1);
                const float* a1 = As + min(4 * bq + 1, nb - 1) * (CC + 1);
                const float* a2 = As + min(4 * bq + 2, nb - 1) * (CC + 1);
                const float* a3 = As + min(4 * bq + 3, nb - 1) * (CC + 1);
                for (int c = 0; c < nc; ++c) {
                    const float av[4] = {a0[c], a1[c], a2[c], a3[c]};
                    const float* hr = Hs + c * F;
                    const float hv[4] = {hr[fk[0]], hr[fk[1]], hr[fk[2]], hr[fk[3]]};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[i][k] = fmaf(av[i], hv[k], acc[i][k]);
                }
            }
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = 4 * bq + i;
                if (b >= nb) break;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = fq + k * FQ;
                    if (f < F) Y[((int64_t)(b0 + b) * s.N + n) * F + f] = acc[i][k] + (bias ? bias[f] : 0.f);
                }
            }
        }
    }
}

// dZ[t][b][n][g] = sum_f D[b][f] H[c][f].  grid = (N, column chunks).
// thread -> (bq, cq): samples 4 bq + i of the batch chunk, columns cq + k CQ (CQ = ceil(nc / 4)): lanes consecutive along c.
__global__ __launch_bounds__(256) void nv_dz_kernel(const float* __restrict__ dY, const float* __restrict__ Ht,
                                                    float* __restrict__ dZ, NvShape s) {
    extern __shared__ float sm[];
    const int TG = s.T * s.G, F = s.F, CC = s.CC;
    float* Hs = sm;                            // [CC][F + 1]: lanes walk down a column of H^T
    float* Ds = sm + (size_t)CC * (F + 1);     // [BCH][F]
    const int n = blockIdx.x, c0 = blockIdx.y * CC, tid = threadIdx.x;
    const int nc = min(CC, TG - c0);
    const int CQ = (nc + 3) / 4;
    const int BCH = min(nv_dz_batch(F), 4 * (256 / CQ));
    const int cq = tid % CQ, bq = tid / CQ;
    const float* hn = Ht + ((int64_t)n * TG + c0) * F;
    for (int i = tid; i < nc * F; i += 256) {
        const int c = i / F, f = i - c * F;
        Hs[c * (F + 1) + f] = hn[i];
    }
    const float* h0 = Hs + min(cq + 0 * CQ, nc - 1) * (F + 1);
    const float* h1 = Hs + min(cq + 1 * CQ, nc - 1) * (F + 1);
    const float* h2 = Hs + min(cq + 2 * CQ, nc - 1) * (F + 1);
    const float* h3 = Hs + min(cq + 3 * CQ, nc - 1) * (F + 1);
    for (int b0 = 0; b0 < s.B; b0 += BCH) {
        const int nb = min(BCH, s.B - b0);
        __syncthreads();
        for (int i = tid; i < nb * F; i += 256) {
            const int b = i / F, f = i - b * F;
            Ds[i] = dY[((int64_t)(b0 + b) * s.N + n) * F + f];
        }
        __syncthreads();
        if (4 * bq < nb) {
            float acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
            const float* d0 = Ds + min(4 * bq + 0, nb - 1) * F;
            const float* d1 = Ds + min(4 * bq + 1, nb - 1) * F;
            const float* d2 = Ds + min(4 * bq + 2, nb - 1) * F;
            const float* d3 = Ds + min(4 * bq + 3, nb - 1) * F;
            for (int f = 0; f < F; ++f) {
                const float dv[4] = {d0[f], d1[f], d2[f], d3[f]};
                const float hv[4] = {h0[f], h1[f], h2[f], h3[f]};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i][k] = fmaf(dv[i], hv[k], acc[i][k]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = 4 * bq + i;
                if (b >= nb) break;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = cq + k * CQ;
                    if (c < nc) {
                        const int cc = c0 + c, t = cc / s.G, g = cc - t * s.G;
                        dZ[(((int64_t)t * s.B + b0 + b) * s.N + n) * s.G + g] = acc[i][k];
                    }
                }
            }
        }
    }
}

// dHt[n][c][f] = sum_b A[b][c] D[b][f], b ascending (fixed order).  grid = (N, column chunks); CC <= 4 * (256 / FQ).
// thread -> (cq, fq): columns 4 cq + i of the chunk, features fq + k FQ.
__global__ __launch_bounds__(256) void nv_dbank_kernel(const float* __restrict__ Z, const float* __restrict__ dY,
                                                       float* __restrict__ dHt, NvShape s) {
    extern __shared__ float sm[];
    const int TG = s.T * s.G, F = s.F, CC = s.CC;
    float* As = sm;                            // [NV_BC][CC]
    float* Ds = sm + (size_t)NV_BC * CC;       // [NV_BC][F]
    const int n = blockIdx.x, c0 = blockIdx.y * CC, tid = threadIdx.x;
    const int nc = min(CC, TG - c0);
    const int FQ = (F + 3) / 4;
    const int fq = tid % FQ, cq = tid / FQ;
    const bool active = 4 * cq < nc;
    int ci[4], fk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ci[i] = min(4 * cq + i, nc - 1);
        fk[i] = min(fq + i * FQ, F - 1);
    }
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
    for (int b0 = 0; b0 < s.B; b0 += NV_BC) {
        const int nb = min(NV_BC, s.B - b0);
        __syncthreads();
        for (int i = tid; i < nb * nc; i += 256) {
            const int b = i / nc, c = i - b * nc;
            As[b * CC + c] = *nv_zrow(Z, s, n, c0 + c, b0 + b);
        }
        for (int i = tid; i < nb * F; i += 256) {
            const int b = i / F, f = i - b * F;
            Ds[i] = dY[((int64_t)(b0 + b) * s.N + n) * F + f];
        }
        __syncthreads();
        if (active) {
            for (int b = 0; b < nb; ++b) {
                const float* ar = As + b * CC;
                const float* dr = Ds + b * F;
                const float av[4] = {ar[ci[0]], ar[ci[1]], ar[ci[2]], ar[ci[3]]};
                const float dv[4] = {dr[fk[0]], dr[fk[1]], dr[fk[2]], dr[fk[3]]};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i][k] = fmaf(av[i], dv[k], acc[i][k]);
            }
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * cq + i;
            if (c >= nc) break;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int f = fq + k * FQ;
                if (f < F) dHt[((int64_t)n * TG + c0 + c) * F + f] = acc[i][k];
            }
        }
    }
}

// out[r][m] = sum over the nodes n of group m (ascending) of dh[r][n]: the adjoint of NodeVariantGF's tap expansion
// h = weight[..., copyNodes] (graphML.py:2485), as a gather in fixed order instead of an atomic scatter.
__global__ __launch_bounds__(256) void nv_fold_kernel(const float* __restrict__ dh, const int32_t* __restrict__ grp_ptr,
                                                      const int32_t* __restrict__ grp_idx, float* __restrict__ out, int N, int M,
                                                      int staged) {
    extern __shared__ float rowbuf[];  // the row, read once with unit stride; the group gathers then hit LDS
    const float* row = dh + (int64_t)blockIdx.x * N;
    if (staged) {
        for (int i = threadIdx.x; i < N; i += 256) rowbuf[i] = row[i];
        __syncthreads();
    }
    for (int m = threadIdx.x; m < M; m += 256) {
        float a = 0.f;
        if (staged)
            for (int i = grp_ptr[m]; i < grp_ptr[m + 1]; ++i) a += rowbuf[grp_idx[i]];
        else
            for (int i = grp_ptr[m]; i < grp_ptr[m + 1]; ++i) a += row[grp_idx[i]];
        out[(int64_t)blockIdx.x * M + m] = a;
    }
}

__global__ __launch_bounds__(256) void nv_add_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = i; j < count; j += stride) dst[j] += src[j];
}

// bank columns per chunk: nv_dbank_kernel covers a chunk with 4 x 4 tiles, (CC / 4) * ceil(F / 4) <= 256 threads; balanced over the chunks
NvShape nv_shape(int B, int N, int G, int F, int T) {
    const int TG = T * G;
    int cap = 4 * (256 / ((F + 3) / 4));
    if (cap > 96) cap = 96;
    const int chunks = (TG + cap - 1) / cap;
    NvShape s{B, N, G, F, T, (TG + chunks - 1) / chunks};
    return s;
}
size_t nv_lds_contract(const NvShape& s) { return ((size_t)s.CC * s.F + (size_t)128 * (s.CC + 1)) * 4; }
size_t nv_lds_dz(const NvShape& s) { return ((size_t)s.CC * (s.F + 1) + (size_t)nv_dz_batch(s.F) * s.F) * 4; }
size_t nv_lds_dbank(const NvShape& s) { return ((size_t)NV_BC * s.CC + (size_t)NV_BC * s.F) * 4; }

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
    const NvShape shp = nv_shape(B, N, G, F, T);
    GF_REQUIRE_SHAPE(F <= 256, "gf_nvgf_forward: F = %d > 256", F);   // nv_contract_kernel keeps NV_BC * F / 256 outputs per thread
    hipLaunchKernelGGL(nv_contract_kernel, dim3(N), dim3(256), nv_lds_contract(shp), st, Z, Ht, bias, Y, shp);
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
    const NvShape shp = nv_shape(B, N, G, F, T);
    GF_REQUIRE_SHAPE(F <= 256, "gf_nvgf_backward: F = %d > 256", F);
    const int chunks = (T * G + shp.CC - 1) / shp.CC;
    GF_REQUIRE_SHAPE(chunks <= 65535, "gf_nvgf_backward: %d column chunks", chunks);
    if (dh) {
        hipLaunchKernelGGL(nv_dbank_kernel, dim3(N, chunks), dim3(256), nv_lds_dbank(shp), st, Z, dY, dHt, shp);
        GF_LAUNCH_CHECK("nv_dbank_kernel");
        hipLaunchKernelGGL(nv_bank_out_kernel, dim3((N + TT - 1) / TT, (F + TT - 1) / TT, E * K * G), dim3(TT, 8), 0, st, dHt, dh,
                           F, E, K, G, N);
        GF_LAUNCH_CHECK("nv_bank_out_kernel");
    }
    if (dx) {
        hipLaunchKernelGGL(nv_bank_in_kernel, dim3((N + TT - 1) / TT, (F + TT - 1) / TT, T * G), dim3(TT, 8), 0, st, h, Ht, F, E, K, G, N);
        GF_LAUNCH_CHECK("nv_bank_in_kernel");
        hipLaunchKernelGGL(nv_dz_kernel, dim3(N, chunks), dim3(256), nv_lds_dz(shp), st, dY, Ht, dZ, shp);
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

// dweight[R, M] from dh[R, N] (R = F*E*K*G rows): groups = the nodes that copy tap node m, DEVICE int32 CSR (grp_ptr [M+1], grp_idx)
extern "C" int gf_nvgf_fold_taps(const float* dh, const int32_t* grp_ptr, const int32_t* grp_idx, float* dweight, int64_t R,
                                 int32_t N, int32_t M, void* stream) {
    GF_REQUIRE_ARG(dh && grp_ptr && grp_idx && dweight, "gf_nvgf_fold_taps: NULL argument");
    GF_REQUIRE_SHAPE(R > 0 && R <= 2147483647 && N > 0 && M > 0, "gf_nvgf_fold_taps: bad shape R=%lld N=%d M=%d", (long long)R, N, M);
    const int staged = (size_t)N * 4 <= 64 * 1024;
    hipLaunchKernelGGL(nv_fold_kernel, dim3((unsigned)R), dim3(256), staged ? (size_t)N * 4 : 0, gf_stream(stream), dh, grp_ptr, grp_idx,
                       dweight, N, M, staged);
    GF_LAUNCH_CHECK("nv_fold_kernel");
    return GF_OK;
}
