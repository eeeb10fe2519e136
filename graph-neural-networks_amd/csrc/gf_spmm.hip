// gf_spmm.hip -- the K-hop GSO-signal product: X_k = op(S) X_{k-1} on node-major signals.
// Replaces the K-1 dense broadcast GEMMs `x = torch.matmul(x, S)` of reference graphML.py:158-161
// (O(B*G*N^2) each) with CSR SpMM (O(B*G*nnz)), and the growing torch.cat (graphML.py:161) with in-place
// writes into tap slot k of the stack Z[T,B,N,G].
//
// HBM-bound.  Algorithmic bytes per hop:  2*B*N*W*4 + nnz*8 + (N+1)*4.
//
// Kernel shape (W = row width in floats, LG = W/4 lanes per row, float4 per lane):
//   * a 256-thread workgroup owns RPB = 256/LG consecutive rows of the (degree-sorted) schedule for BT batch
//     entries; its CSR segment (col, val) is staged in LDS with coalesced loads, column indices pre-scaled
//     to float offsets; each LG-lane group then walks its own row out of LDS (broadcast reads), gathering one
//     full W*4-byte line per neighbour per batch entry and accumulating in registers -- a segmented reduction
//     with the segment (= row) pinned to the lane group, so no atomics and a fixed summation order.
//   * rows in one wavefront have near-equal degree (plan schedule), so the walk is nearly divergence-free.
//   * blockIdx -> (batch tile, row block) is XCD-aware: workgroup L runs on XCD L%8 (observed dispatch order),
//     so batch tile t is pinned to XCD t%8 and all row blocks of one batch tile are swept by one XCD: the
//     N*W*4*BT-byte gather panel is shared through ONE L2 instead of being replicated in eight.
#include <stdlib.h>

#include "gf_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 2048;  // CSR entries staged per pass: 16 KB of LDS

__device__ __forceinline__ void fma4(float4& a, float s, const float4& x) {
    a.x = fmaf(s, x.x, a.x);
    a.y = fmaf(s, x.y, a.y);
    a.z = fmaf(s, x.z, a.z);
    a.w = fmaf(s, x.w, a.w);
}

template <int LG, int BT>
__global__ __launch_bounds__(kThreads) void spmm_hop_vec_kernel(const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ col,
                                                                const float* __restrict__ val,
                                                                const int32_t* __restrict__ rowid,
                                                                const float* __restrict__ Xin, float* __restrict__ Xout,
                                                                int N, int B, int nRowBlocks, int nBTiles, int xcd_map) {
    constexpr int RPB = kThreads / LG;
    constexpr int W = LG * 4;
    __shared__ int32_t s_off[kChunk];  // neighbour row offset in floats (col * W)
    __shared__ float s_val[kChunk];

    int btile, rb;
    {
        const int L = blockIdx.x;
        if (xcd_map) {
            const int xcd = L & 7, slot = L >> 3;
            const int btl = slot / nRowBlocks;
            rb = slot - btl * nRowBlocks;
            btile = btl * 8 + xcd;
        } else {
            btile = L / nRowBlocks;
            rb = L - btile * nRowBlocks;
        }
    }
    if (btile >= nBTiles) return;  // uniform per block: no barrier is skipped by part of a block
    const int b0 = btile * BT;

    const int tid = threadIdx.x;
    const int grp = tid / LG, li = tid - grp * LG;
    const int p_lo = rb * RPB;
    const int p_hi = min(p_lo + RPB, N);
    const int p = p_lo + grp;
    int rs = 0, re = 0;
    if (p < N) {
        rs = rowptr[p];
        re = rowptr[p + 1];
    }
    const int seg_lo = rowptr[p_lo], seg_hi = rowptr[p_hi];

    const float* xb[BT];
    float4 acc[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        const int b = min(b0 + t, B - 1);  // clamp loads of a ragged last tile; its stores are masked below
        xb[t] = Xin + (int64_t)b * N * W + li * 4;
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    for (int base = seg_lo; base < seg_hi; base += kChunk) {
        const int cnt = min(kChunk, seg_hi - base);
        if (base != seg_lo) __syncthreads();  // everyone is done with the previous chunk
        for (int i = tid; i < cnt; i += kThreads) {
            s_off[i] = col[base + i] * W;
            s_val[i] = val[base + i];
        }
        __syncthreads();
        int q = max(rs, base) - base;
        const int hi = min(re, base + cnt) - base;
        for (; q + 3 < hi; q += 4) {
            const int o0 = s_off[q], o1 = s_off[q + 1], o2 = s_off[q + 2], o3 = s_off[q + 3];
            const float v0 = s_val[q], v1 = s_val[q + 1], v2 = s_val[q + 2], v3 = s_val[q + 3];
            float4 x0[BT], x1[BT], x2[BT], x3[BT];
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                x0[t] = *reinterpret_cast<const float4*>(xb[t] + o0);
                x1[t] = *reinterpret_cast<const float4*>(xb[t] + o1);
                x2[t] = *reinterpret_cast<const float4*>(xb[t] + o2);
                x3[t] = *reinterpret_cast<const float4*>(xb[t] + o3);
            }
#pragma unroll
            for (int t = 0; t < BT; ++t) {  // fixed order q, q+1, q+2, q+3: deterministic
                fma4(acc[t], v0, x0[t]);
                fma4(acc[t], v1, x1[t]);
                fma4(acc[t], v2, x2[t]);
                fma4(acc[t], v3, x3[t]);
            }
        }
        for (; q < hi; ++q) {
            const int o0 = s_off[q];
            const float v0 = s_val[q];
#pragma unroll
            for (int t = 0; t < BT; ++t) fma4(acc[t], v0, *reinterpret_cast<const float4*>(xb[t] + o0));
        }
    }

    if (p < N) {
        const int64_t orow = (int64_t)rowid[p] * W + li * 4;
#pragma unroll
        for (int t = 0; t < BT; ++t)
            if (b0 + t < B) *reinterpret_cast<float4*>(Xout + (int64_t)(b0 + t) * N * W + orow) = acc[t];
    }
}

// any width W (G = 1, odd G, N*W >= 2^31): one thread per output element, CSR read through the caches.
__global__ __launch_bounds__(kThreads) void spmm_hop_generic_kernel(const int32_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ col,
                                                                    const float* __restrict__ val,
                                                                    const int32_t* __restrict__ rowid,
                                                                    const float* __restrict__ Xin, float* __restrict__ Xout,
                                                                    int N, int W, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        const int w = (int)(idx % W);
        const int64_t t = idx / W;
        const int p = (int)(t % N);
        const int64_t b = t / N;
        const float* xb = Xin + b * N * W + w;
        float acc = 0.f;
        for (int q = rowptr[p]; q < rowptr[p + 1]; ++q) acc = fmaf(val[q], xb[(int64_t)col[q] * W], acc);
        Xout[(b * N + rowid[p]) * W + w] = acc;
    }
}

int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

template <int LG>
int launch_vec(const gf_csr_dev& m, const float* Xin, float* Xout, int N, int B, int bt, int xcd_map, hipStream_t st) {
    constexpr int RPB = kThreads / LG;
    const int nRowBlocks = (N + RPB - 1) / RPB;
    const int nBTiles = (B + bt - 1) / bt;
    const int64_t nblk = xcd_map ? (int64_t)((nBTiles + 7) / 8) * 8 * nRowBlocks : (int64_t)nBTiles * nRowBlocks;
    GF_REQUIRE_SHAPE(nblk < (int64_t)INT32_MAX, "gf_spmm_hop: grid of %lld blocks too large", (long long)nblk);
    dim3 grid((unsigned)nblk), block(kThreads);
#define GF_LAUNCH_BT(BTV)                                                                                           \
    hipLaunchKernelGGL((spmm_hop_vec_kernel<LG, BTV>), grid, block, 0, st, m.rowptr, m.col, m.val, m.rowid, Xin, Xout, \
                       N, B, nRowBlocks, nBTiles, xcd_map)
    switch (bt) {
        case 1: GF_LAUNCH_BT(1); break;
        case 2: GF_LAUNCH_BT(2); break;
        default: GF_LAUNCH_BT(4); break;
    }
#undef GF_LAUNCH_BT
    GF_LAUNCH_CHECK("spmm_hop_vec_kernel");
    return GF_OK;
}

}  // namespace

extern "C" int gf_spmm_hop(const gf_plan* plan, int32_t op, const float* Xin, float* Xout, int32_t B, int32_t W,
                           void* stream) {
    GF_REQUIRE_ARG(plan && Xin && Xout, "gf_spmm_hop: NULL argument");
    GF_REQUIRE_ARG(op == GF_OP_FWD || op == GF_OP_BWD, "gf_spmm_hop: op = %d", op);
    GF_REQUIRE_ARG(Xin != Xout, "gf_spmm_hop: in-place hop is not supported");
    GF_REQUIRE_SHAPE(B > 0 && W > 0, "gf_spmm_hop: bad shape B=%d W=%d", B, W);
    const gf_csr_dev& m = plan->mat[op];
    const int N = plan->n;
    hipStream_t st = gf_stream(stream);

    static const int env_bt = env_int("GFHIP_SPMM_BT", 0);
    static const int env_xcd = env_int("GFHIP_SPMM_XCD", 1);
    static const int env_generic = env_int("GFHIP_SPMM_GENERIC", 0);

    const int lg = W / 4;
    const bool pow2 = lg > 0 && (lg & (lg - 1)) == 0;
    const bool vec_ok = !env_generic && (W % 4 == 0) && pow2 && lg <= 64 && (int64_t)N * W < (int64_t)INT32_MAX;
    if (vec_ok) {
        // batch-tile heuristic: keep the BT-entry gather panel (N*W*4*BT bytes) inside one XCD's 4 MiB L2
        const int64_t panel = (int64_t)N * W * 4;
        int bt = 1;
        if (panel * 2 <= (5 << 19) && B >= 16) bt = 2;
        if (panel * 4 <= (5 << 19) && B >= 32) bt = 4;
        if (env_bt == 1 || env_bt == 2 || env_bt == 4) bt = env_bt;
        switch (lg) {
            case 1: return launch_vec<1>(m, Xin, Xout, N, B, bt, env_xcd, st);
            case 2: return launch_vec<2>(m, Xin, Xout, N, B, bt, env_xcd, st);
            case 4: return launch_vec<4>(m, Xin, Xout, N, B, bt, env_xcd, st);
            case 8: return launch_vec<8>(m, Xin, Xout, N, B, bt, env_xcd, st);
            case 16: return launch_vec<16>(m, Xin, Xout, N, B, bt, env_xcd, st);
            case 32: return launch_vec<32>(m, Xin, Xout, N, B, bt, env_xcd, st);
            default: return launch_vec<64>(m, Xin, Xout, N, B, bt, env_xcd, st);
        }
    }
    const int64_t total = (int64_t)B * N * W;
    const int64_t want = (total + kThreads - 1) / kThreads;
    dim3 grid((unsigned)(want < 65536 * 16 ? want : 65536 * 16)), block(kThreads);
    hipLaunchKernelGGL(spmm_hop_generic_kernel, grid, block, 0, st, m.rowptr, m.col, m.val, m.rowid, Xin, Xout, N, W, total);
    GF_LAUNCH_CHECK("spmm_hop_generic_kernel");
    return GF_OK;
}

extern "C" int gf_khop(const gf_plan* const* plans, int32_t E, int32_t op, float* Z, int32_t B, int32_t W, int32_t K,
                       void* stream) {
    GF_REQUIRE_ARG(plans && Z, "gf_khop: NULL argument");
    GF_REQUIRE_SHAPE(E > 0 && K > 0 && B > 0 && W > 0, "gf_khop: bad shape E=%d K=%d B=%d W=%d", E, K, B, W);
    for (int e = 0; e < E; ++e) {
        GF_REQUIRE_ARG(plans[e] != nullptr, "gf_khop: plan %d is NULL", e);
        GF_REQUIRE_SHAPE(plans[e]->n == plans[0]->n, "gf_khop: plan %d has %d nodes, plan 0 has %d", e, plans[e]->n,
                         plans[0]->n);
    }
    const int64_t tap = (int64_t)B * plans[0]->n * W;
    for (int e = 0; e < E; ++e)
        for (int k = 1; k < K; ++k) {
            const float* src = (k == 1) ? Z : Z + (int64_t)(1 + e * (K - 1) + (k - 2)) * tap;
            float* dst = Z + (int64_t)(1 + e * (K - 1) + (k - 1)) * tap;
            const int rc = gf_spmm_hop(plans[e], op, src, dst, B, W, stream);
            if (rc != GF_OK) return rc;
        }
    return GF_OK;
}

extern "C" int gf_time_spmm_hop(const gf_plan* plan, int32_t op, const float* Xin, float* Xout, int32_t B, int32_t W,
                                int32_t iters, void* stream, float* avg_ms) {
    GF_REQUIRE_ARG(avg_ms && iters > 0, "gf_time_spmm_hop: bad iters / NULL avg_ms");
    hipStream_t st = gf_stream(stream);
    hipEvent_t e0, e1;
    GF_HIP(hipEventCreate(&e0));
    GF_HIP(hipEventCreate(&e1));
    int rc = gf_spmm_hop(plan, op, Xin, Xout, B, W, stream);  // warm-up (also validates arguments)
    if (rc == GF_OK) {
        GF_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < iters && rc == GF_OK; ++i) rc = gf_spmm_hop(plan, op, Xin, Xout, B, W, stream);
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
