// gf_evgf.hip -- edge-variant graph filter with the filter matrices stored PER EDGE.
// Replaces reference graphML.py:389-488 (EVGF) as called by EdgeVariantGF.forward (graphML.py:2670-2698):
//     v_0^{fg} = Phi_0^{fg} x_g,   v_k^{fg} = Phi_k^{fg} v_{k-1}^{fg},   y_f = sum_{g,k} v_k^{fg} + b_f     (column convention)
// The reference materialises Phi = weightEV * pattern as a dense [F,E,K,G,N,N] tensor and runs a broadcast dense
// matmul per tap (3e13 bytes at N = 5e4); here only on-pattern entries exist: tap 0 is the diagonal wdiag[F,G,N], taps
// k >= 1 are wedge[F,K-1,G,nnzp] on one shared CSR pattern, and every chain (f,g) is a sparse product.
//
// Data layout in HBM: chain state V[c = f*G+g][n][b] -- batch index contiguous -- because the weight of an edge is one
// scalar for the whole batch: a hop is an axpy over b per pattern entry, the gather of v_{k-1}[col] is one contiguous
// B-float row, and all loads/stores of a wave are coalesced.  x / dy / y / dx cross the boundary in the reference layout
// [B,C,N] and are transposed once ([C][N][B] "node-batch" layout) by a tiled kernel.
//
// HBM-bound: per tap the algorithmic traffic is the weight stream F*G*nnzp*4 (read once) + the state 2*B*F*G*N*4.
// What limits a tap in practice is the gather (every state row is fetched once per incident edge): chain c runs on XCD c % 8
// (workgroups are dealt to the XCDs round-robin) so that each 4 MB L2 holds ONE gather panel (N*B*4 bytes) at a time, and the
// gathers are 16 bytes per lane when B % 4 == 0.
// Everything is deterministic: reductions over b use fixed-order wave shuffles, sums over g/k/f are sequential loops.
#include <algorithm>
#include <new>
#include <vector>

#include "gf_common.h"

struct gf_ev_plan {
    int32_t n = 0;
    int64_t nnzp = 0;
    int64_t device_bytes = 0;
    int32_t* rowptr = nullptr;    // [N+1]   pattern CSR: row i lists its on-pattern columns j (ascending); entry index = p
    int32_t* col = nullptr;       // [nnzp]
    int32_t* row = nullptr;       // [nnzp]  COO row of entry p (SDDMM)
    int32_t* t_rowptr = nullptr;  // [N+1]   transposed pattern: row j lists the entries p with col[p] == j
    int32_t* t_col = nullptr;     // [nnzp]  = row[p]   (ascending within a transposed row)
    int32_t* t_vidx = nullptr;    // [nnzp]  = p        (value index into wedge)
    // 16-bit copies of the node indices when N <= 65535: the index streams are re-read for every chain and share the 4 MB L2 with the
    // gather panel (config 5: 2.2 MB -> 1.1 MB per array)
    uint16_t* col16 = nullptr;
    uint16_t* row16 = nullptr;
    uint16_t* t_col16 = nullptr;
};

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(int64_t work_items) {
    int64_t blocks = (work_items + kThreads - 1) / kThreads;
    const int64_t cap = 256 * 64;  // grid-stride beyond this
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// ---- batched transposes between the reference layout and the node-batch layout -------------------------------------
// src element (c, r, q), r < R, q < Q at src[c*src_cs + r*src_rs + q];  dst element (c, q, r), q < Qpad at
// dst[c*dst_cs + q*dst_rs + r] = (q < Q ? src : 0).
__global__ __launch_bounds__(256) void ev_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Q,
                                                           int Qpad, int64_t src_cs, int64_t src_rs, int64_t dst_cs,
                                                           int64_t dst_rs) {
    __shared__ float tile[32][33];
    const int c = blockIdx.z;
    const int q0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* s = src + (int64_t)c * src_cs;
    float* d = dst + (int64_t)c * dst_cs;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, q = q0 + tx;
        tile[ty + i][tx] = (r < R && q < Q) ? s[(int64_t)r * src_rs + q] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int q = q0 + ty + i, r = r0 + tx;
        if (q < Qpad && r < R) d[(int64_t)q * dst_rs + r] = tile[tx][ty + i];
    }
}

int to_nodebatch(const float* x, float* Xt, int B, int Cn, int Nin, int N, hipStream_t st) {
    GF_REQUIRE_SHAPE(Cn <= 65535, "gf_evgf: %d feature planes exceed the grid limit", Cn);
    dim3 grid((N + 31) / 32, (B + 31) / 32, Cn);
    hipLaunchKernelGGL(ev_transpose_kernel, grid, dim3(256), 0, st, x, Xt, B, Nin, N, (int64_t)Nin, (int64_t)Cn * Nin,
                       (int64_t)N * B, (int64_t)B);
    GF_LAUNCH_CHECK("ev_transpose_kernel(to node-batch)");
    return GF_OK;
}

int from_nodebatch(const float* Yt, float* y, int B, int Cn, int N, int Nout, hipStream_t st) {
    GF_REQUIRE_SHAPE(Cn <= 65535, "gf_evgf: %d feature planes exceed the grid limit", Cn);
    dim3 grid((B + 31) / 32, (Nout + 31) / 32, Cn);
    hipLaunchKernelGGL(ev_transpose_kernel, grid, dim3(256), 0, st, Yt, y, Nout, B, B, (int64_t)N * B, (int64_t)B,
                       (int64_t)Nout, (int64_t)Cn * Nout);
    GF_LAUNCH_CHECK("ev_transpose_kernel(from node-batch)");
    return GF_OK;
}

// ---- tap 0: V0[c][n][b] = wdiag[c][n] * Xt[g(c)][n][b]   (Phi_0 is diagonal, graphML.py:2653-2668) -------------------
__global__ __launch_bounds__(kThreads) void ev_tap0_kernel(const float* __restrict__ wdiag, const float* __restrict__ Xt,
                                                           float* __restrict__ V0, int G, int64_t NB, int B, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        const int64_t c = idx / NB, nb = idx - c * NB;
        const int g = (int)(c % G);
        V0[idx] = wdiag[c * (NB / B) + nb / B] * Xt[(int64_t)g * NB + nb];
    }
}

// 4 batch entries per thread (B % 4 == 0); the chain rides on blockIdx.y so that the index arithmetic inside a chain is 32-bit (the
// first version split a 64-bit flat index three times per 16 bytes: 1.13 ms for a 3.3 GB write at config 5)
__global__ __launch_bounds__(kThreads) void ev_tap0_v4_kernel(const float* __restrict__ wdiag, const float* __restrict__ Xt,
                                                              float* __restrict__ V0, int G, int N, int B4) {
    const int c = blockIdx.y, g = c % G;
    const int NB4 = N * B4;
    const float4* X4 = reinterpret_cast<const float4*>(Xt) + (int64_t)g * NB4;
    float4* V4 = reinterpret_cast<float4*>(V0) + (int64_t)c * NB4;
    const float* wd = wdiag + (int64_t)c * N;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < NB4; i += gridDim.x * kThreads) {
        const float w = wd[i / B4];
        const float4 x = X4[i];
        V4[i] = make_float4(w * x.x, w * x.y, w * x.z, w * x.w);
    }
}

// ---- one tap of every chain: out[c][i][b] = add[c/add_div][i][b] + sum_{q in row i} W_c[vidx ? vidx[q] : q] * in[c/in_div][col[q]][b]
// W_c = wedge + ((f*K1 + kidx)*G + g)*nnzp for c = f*G + g.  Forward: (rowptr, col) = pattern, vidx = NULL.
// Backward: (rowptr, col, vidx) = transposed pattern, add = dy (chain-broadcast via add_div = G).
__global__ __launch_bounds__(kThreads) void ev_hop_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const int32_t* __restrict__ vidx, const float* __restrict__ wedge,
                                                          const float* __restrict__ in, const float* __restrict__ add,
                                                          float* __restrict__ out, int N, int B, int G, int K1, int kidx,
                                                          int64_t nnzp, int in_div, int add_div, int64_t total) {
    const int64_t NB = (int64_t)N * B;
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        const int64_t c = idx / NB;
        const int64_t nb = idx - c * NB;
        const int i = (int)(nb / B), b = (int)(nb - (int64_t)i * B);
        const int f = (int)(c / G), g = (int)(c - (int64_t)f * G);
        const float* W = wedge + ((int64_t)(f * K1 + kidx) * G + g) * nnzp;
        const float* vin = in + (c / in_div) * NB + b;
        float acc = add ? add[(c / add_div) * NB + nb] : 0.f;
        const int q1 = rowptr[i + 1];
        int q = rowptr[i];
        // 4 entries per round: all index / weight loads first, then the 4 gathers they address, then the FMAs in entry
        // order (fixed summation order) -- a one-entry loop is a chain of two dependent L2 round trips per entry
        for (; q + 3 < q1; q += 4) {
            int cj[4], pj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cj[u] = col[q + u];
                pj[u] = vidx ? vidx[q + u] : q + u;
            }
            float wv[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                wv[u] = W[pj[u]];
                xv[u] = vin[(int64_t)cj[u] * B];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fmaf(wv[u], xv[u], acc);
        }
        for (; q < q1; ++q) {
            const int p = vidx ? vidx[q] : q;
            acc = fmaf(W[p], vin[(int64_t)col[q] * B], acc);
        }
        out[idx] = acc;
    }
}

// ---- the same tap with the pattern segment of a row block staged in LDS (B <= 64): one workgroup = one chain x kRowsPerWG
// consecutive rows.  The (column, weight) pairs of those rows are loaded once, coalesced (3 of the 4 global loads per entry
// and thread of the kernel above were index / weight fetches with 16 useful bytes per wave-instruction), then every thread
// (row, b) walks its row from LDS and only the gathers of v_{k-1} go to memory, 4 in flight per thread.
#ifndef GF_EV_RPW
#define GF_EV_RPW 64
#endif
constexpr int kRowsPerWG = GF_EV_RPW;
constexpr int kEvChunk = 2048;  // staged entries per pass (16 KB of LDS)
#ifndef GF_EV_U
#define GF_EV_U 8
#endif
constexpr int kEvU = GF_EV_U;  // gathers in flight per row and thread (ev_hop_lds4_kernel)
#ifndef GF_EV_ROWS
#define GF_EV_ROWS 1
#endif
constexpr int kEvRows = GF_EV_ROWS;  // rows of a thread whose gathers are issued together

template <int LB>
__global__ __launch_bounds__(kThreads) void ev_hop_lds_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                              const int32_t* __restrict__ vidx, const float* __restrict__ wedge,
                                                              const float* __restrict__ in, const float* __restrict__ add,
                                                              float* __restrict__ out, int N, int B, int G, int K1, int kidx,
                                                              int64_t nnzp, int in_div, int add_div, int nRowBlocks, int nChains) {
    constexpr int RPP = kThreads / LB;  // rows per pass
    constexpr int RPW = RPP > kRowsPerWG ? RPP : kRowsPerWG;  // rows per workgroup
    __shared__ int32_t s_col[kEvChunk];
    __shared__ float s_w[kEvChunk];
    const int64_t NB = (int64_t)N * B;
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its own 4 MB L2: chain c lives on XCD
    // c % 8, whose workgroups walk its row blocks in order -- one gather panel (N*B*4 bytes) per L2 at a time instead of the
    // panels of all chains in flight in every L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int c = (slot / nRowBlocks) * 8 + xcd;
    const int rb = slot % nRowBlocks;
    if (c >= nChains) return;
    const int r0 = rb * RPW, r1 = min(N, r0 + RPW);
    const int f = c / G, g = c - f * G;
    const float* W = wedge + ((int64_t)(f * K1 + kidx) * G + g) * nnzp;
    const int tid = threadIdx.x, lr = tid / LB, b = tid - lr * LB;
    const float* vin = in + (int64_t)(c / in_div) * NB + b;
    const int seg_lo = rowptr[r0], seg_hi = rowptr[r1];
    float acc[RPW / RPP];
#pragma unroll
    for (int j = 0; j < RPW / RPP; ++j) acc[j] = 0.f;
    for (int base = seg_lo; base < seg_hi; base += kEvChunk) {
        const int cnt = min(kEvChunk, seg_hi - base);
        if (base != seg_lo) __syncthreads();
        for (int i = tid; i < cnt; i += kThreads) {
            s_col[i] = col[base + i];
            s_w[i] = W[vidx ? vidx[base + i] : base + i];
        }
        __syncthreads();
        if (b < B) {
#pragma unroll
            for (int j = 0; j < RPW / RPP; ++j) {
                const int r = r0 + j * RPP + lr;
                if (r >= r1) break;
                int q = max(rowptr[r], base) - base;
                const int qe = min(rowptr[r + 1], base + cnt) - base;
                float a = acc[j];
                for (; q + 3 < qe; q += 4) {
                    const float x0 = vin[(int64_t)s_col[q] * B], x1 = vin[(int64_t)s_col[q + 1] * B];
                    const float x2 = vin[(int64_t)s_col[q + 2] * B], x3 = vin[(int64_t)s_col[q + 3] * B];
                    a = fmaf(s_w[q], x0, a);
                    a = fmaf(s_w[q + 1], x1, a);
                    a = fmaf(s_w[q + 2], x2, a);
                    a = fmaf(s_w[q + 3], x3, a);
                }
                for (; q < qe; ++q) a = fmaf(s_w[q], vin[(int64_t)s_col[q] * B], a);
                acc[j] = a;
            }
        }
    }
    if (b < B) {
#pragma unroll
        for (int j = 0; j < RPW / RPP; ++j) {
            const int r = r0 + j * RPP + lr;
            if (r >= r1) break;
            const int64_t o = (int64_t)r * B + b;
            out[(int64_t)c * NB + o] = acc[j] + (add ? add[(int64_t)(c / add_div) * NB + o] : 0.f);
        }
    }
}

// ---- the same kernel with 4 batch entries per thread (B % 4 == 0): a lane fetches 16 bytes per gather instead of 4, so a
// wave-instruction moves 1 KiB instead of 256 B through the same address path (the gathers are L2 hits: what limits them is
// the per-lane request rate, not bytes -- the 4-byte version sat at 3.8 TB/s of gathered rows).  Per-element summation
// order is unchanged: results are bit-identical to ev_hop_lds_kernel.
template <int LQ, class IT>  // lanes per row = B / 4 rounded up to a power of two; IT = int32_t | uint16_t column indices
__global__ __launch_bounds__(kThreads) void ev_hop_lds4_kernel(const int32_t* __restrict__ rowptr, const IT* __restrict__ col,
                                                               const int32_t* __restrict__ vidx, const float* __restrict__ wedge,
                                                               const float* __restrict__ in, const float* __restrict__ add,
                                                               float* __restrict__ out, int N, int B, int G, int K1, int kidx,
                                                               int64_t nnzp, int in_div, int add_div, int nRowBlocks, int nChains) {
    constexpr int RPP = kThreads / LQ;
    constexpr int RPW = RPP > kRowsPerWG ? RPP : kRowsPerWG;
    __shared__ int32_t s_col[kEvChunk];
    __shared__ float s_w[kEvChunk];
    const int64_t NB = (int64_t)N * B;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;  // chain c on XCD c % 8 (see ev_hop_lds_kernel)
    const int c = (slot / nRowBlocks) * 8 + xcd;
    const int rb = slot % nRowBlocks;
    if (c >= nChains) return;
    const int r0 = rb * RPW, r1 = min(N, r0 + RPW);
    const int f = c / G, g = c - f * G;
    const float* W = wedge + ((int64_t)(f * K1 + kidx) * G + g) * nnzp;
    const int tid = threadIdx.x, lr = tid / LQ, b = (tid - lr * LQ) * 4;
    const float* vin = in + (int64_t)(c / in_div) * NB + b;
    const int seg_lo = rowptr[r0], seg_hi = rowptr[r1];
    float4 acc[RPW / RPP];
#pragma unroll
    for (int j = 0; j < RPW / RPP; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fma4 = [](float w, const float4& x, float4& a) {
        a.x = fmaf(w, x.x, a.x);
        a.y = fmaf(w, x.y, a.y);
        a.z = fmaf(w, x.z, a.z);
        a.w = fmaf(w, x.w, a.w);
    };
    for (int base = seg_lo; base < seg_hi; base += kEvChunk) {
        const int cnt = min(kEvChunk, seg_hi - base);
        if (base != seg_lo) __syncthreads();
        for (int i = tid; i < cnt; i += kThreads) {
            s_col[i] = (int32_t)col[base + i];
            s_w[i] = W[vidx ? vidx[base + i] : base + i];  // (non-temporal weight loads: adjoint taps 20 % slower; non-temporal output stores: no change)
        }
        __syncthreads();
        if (b < B) {
            // kEvU gathers in flight per row, kEvRows rows of a thread walked together (measured: 8 x 1 is as good as it gets, more
            // in flight is slower -- DESIGN.md section 3.3).  Entries are added in ascending order per row whatever the setting.
            constexpr int J = RPW / RPP, JI = J < kEvRows ? J : kEvRows;  // rows of a thread / rows walked together
#pragma unroll
            for (int j0 = 0; j0 < J; j0 += JI) {
                int q[JI], qe[JI];
                bool more = false;
#pragma unroll
                for (int j = 0; j < JI; ++j) {
                    const int r = r0 + (j0 + j) * RPP + lr;
                    q[j] = qe[j] = 0;
                    if (r < r1) {
                        q[j] = max(rowptr[r], base) - base;
                        qe[j] = min(rowptr[r + 1], base + cnt) - base;
                    }
                    more |= q[j] < qe[j];
                }
                while (more) {
                    float4 x[JI][kEvU];
#pragma unroll
                    for (int j = 0; j < JI; ++j)
#pragma unroll
                        for (int i = 0; i < kEvU; ++i)
                            if (q[j] + i < qe[j]) x[j][i] = *reinterpret_cast<const float4*>(vin + (int64_t)s_col[q[j] + i] * B);
                    more = false;
#pragma unroll
                    for (int j = 0; j < JI; ++j) {
#pragma unroll
                        for (int i = 0; i < kEvU; ++i)
                            if (q[j] + i < qe[j]) fma4(s_w[q[j] + i], x[j][i], acc[j0 + j]);
                        q[j] += kEvU;
                        more |= q[j] < qe[j];
                    }
                }
            }
        }
    }
    if (b < B) {
#pragma unroll
        for (int j = 0; j < RPW / RPP; ++j) {
            const int r = r0 + j * RPP + lr;
            if (r >= r1) break;
            const int64_t o = (int64_t)r * B + b;
            float4 v = acc[j];
            if (add) {
                const float4 ad = *reinterpret_cast<const float4*>(add + (int64_t)(c / add_div) * NB + o);
                v.x += ad.x, v.y += ad.y, v.z += ad.z, v.w += ad.w;
            }
            *reinterpret_cast<float4*>(out + (int64_t)c * NB + o) = v;
        }
    }
}

// ---- y: Yt[f][n][b] = bias[f] + sum_{k} sum_{g} V[k][f*G+g][n][b]      (graphML.py:481-487) ----------------------------
// VEC = 4: 16 bytes per lane and term (N*B % 4 == 0); the sum runs over k then g in the same order either way.
template <int VEC>
__global__ __launch_bounds__(kThreads) void ev_sum_kernel(const float* __restrict__ V, const float* __restrict__ bias,
                                                          float* __restrict__ Yt, int G, int K, int64_t NB, int64_t CNB,
                                                          int64_t total) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int64_t NBv = NB / VEC;
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        const int64_t f = idx / NBv, nb = (idx - f * NBv) * VEC;
        vec_t acc = bias ? bias[f] : 0.f;
        for (int k = 0; k < K; ++k) {
            const float* v = V + (int64_t)k * CNB + f * G * NB + nb;
#pragma unroll 8
            for (int g = 0; g < G; ++g) acc += *reinterpret_cast<const vec_t*>(v + (int64_t)g * NB);
        }
        *reinterpret_cast<vec_t*>(Yt + f * NB + nb) = acc;
    }
}

// ---- reductions over the batch: LB lanes (power of two <= 64) per output, fixed-order xor tree ------------------------
template <int LB>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LB / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// dW_c[p] = sum_b U[c/u_div][row[p]][b] * V[c][col[p]][b]        (SDDMM on the pattern: dPhi_k = u_k v_{k-1}^T)
template <int LB>
__global__ __launch_bounds__(kThreads) void ev_sddmm_kernel(const int32_t* __restrict__ row, const int32_t* __restrict__ col,
                                                            const float* __restrict__ U, const float* __restrict__ V,
                                                            float* __restrict__ dwedge, int N, int B, int G, int K1, int kidx,
                                                            int64_t nnzp, int u_div, int64_t nChains) {
    const int64_t NB = (int64_t)N * B;
    const int lane = threadIdx.x % LB;
    constexpr int EPG = 4;   // consecutive pattern entries of one chain per lane group and round: their (row, col) pairs are loaded
                             // first, then the 2*EPG gathers, then the fixed-order reductions
    constexpr int GPB = kThreads / LB, ROUNDS = 4;   // lane groups per workgroup, entry quads per lane group
    const int64_t epc = (nnzp + EPG - 1) / EPG;  // entry quads per chain
    const int64_t bpc = (epc + GPB * ROUNDS - 1) / (GPB * ROUNDS);  // workgroups per chain
    // chain c on XCD c % 8 (blockIdx % 8), its entry quads in order: one gather panel V[c] per L2 at a time (see ev_hop_lds_kernel)
    const int xcd = blockIdx.x & 7;
    const int64_t slot = blockIdx.x >> 3;
    const int64_t c = (slot / bpc) * 8 + xcd;
    if (c >= nChains) return;
    const int64_t q0 = (slot % bpc) * (GPB * ROUNDS) + threadIdx.x / LB;
    for (int r = 0; r < ROUNDS; ++r) {  // whole lane groups leave the loop together
        const int64_t quad = q0 + (int64_t)r * GPB;
        if (quad >= epc) break;
        const int64_t p0 = quad * EPG;
        const float* ub = U + (c / u_div) * NB;
        const float* vb = V + c * NB;
        int ri[EPG], ci[EPG];
#pragma unroll
        for (int e = 0; e < EPG; ++e) {
            const int64_t p = p0 + e < nnzp ? p0 + e : nnzp - 1;  // clamp: the surplus lanes of the last quad recompute its last entry
            ri[e] = (int)row[p];
            ci[e] = (int)col[p];
        }
        float acc[EPG];
#pragma unroll
        for (int e = 0; e < EPG; ++e) acc[e] = 0.f;
        for (int b = lane; b < B; b += LB) {
            float uu[EPG], vv[EPG];
#pragma unroll
            for (int e = 0; e < EPG; ++e) {
                uu[e] = ub[(int64_t)ri[e] * B + b];
                vv[e] = vb[(int64_t)ci[e] * B + b];
            }
#pragma unroll
            for (int e = 0; e < EPG; ++e) acc[e] = fmaf(uu[e], vv[e], acc[e]);
        }
        const int f = (int)(c / G), g = (int)(c - (int64_t)f * G);
        float* dst = dwedge + ((int64_t)(f * K1 + kidx) * G + g) * nnzp;
#pragma unroll
        for (int e = 0; e < EPG; ++e) {
            const float tot = group_sum<LB>(acc[e]);
            if (lane == 0 && p0 + e < nnzp) dst[p0 + e] = tot;
        }
    }
}

// 4 batch entries per lane (B % 4 == 0): 16-byte gathers of u and v, LQ = B/4 lanes per entry group (see ev_hop_lds4_kernel).
// Fixed summation order (4 in-lane FMAs, then the xor tree), different from the scalar kernel's by rounding only.
template <int LQ, class IT>
__global__ __launch_bounds__(kThreads) void ev_sddmm4_kernel(const IT* __restrict__ row, const IT* __restrict__ col,
                                                             const float* __restrict__ U, const float* __restrict__ V,
                                                             float* __restrict__ dwedge, int N, int B, int G, int K1, int kidx,
                                                             int64_t nnzp, int u_div, int64_t nChains) {
    const int64_t NB = (int64_t)N * B;
    const int lane = threadIdx.x % LQ;
    constexpr int EPG = 4;   // consecutive pattern entries of one chain per lane group and round: their (row, col) pairs are loaded
                             // first, then the 2*EPG gathers, then the fixed-order reductions
    constexpr int GPB = kThreads / LQ, ROUNDS = 4;   // lane groups per workgroup, entry quads per lane group
    const int64_t epc = (nnzp + EPG - 1) / EPG;  // entry quads per chain
    const int64_t bpc = (epc + GPB * ROUNDS - 1) / (GPB * ROUNDS);  // workgroups per chain
    // chain c on XCD c % 8 (blockIdx % 8), its entry quads in order: one gather panel V[c] per L2 at a time (see ev_hop_lds_kernel)
    const int xcd = blockIdx.x & 7;
    const int64_t slot = blockIdx.x >> 3;
    const int64_t c = (slot / bpc) * 8 + xcd;
    if (c >= nChains) return;
    const int64_t q0 = (slot % bpc) * (GPB * ROUNDS) + threadIdx.x / LQ;
    for (int r = 0; r < ROUNDS; ++r) {  // whole lane groups leave the loop together
        const int64_t quad = q0 + (int64_t)r * GPB;
        if (quad >= epc) break;
        const int64_t p0 = quad * EPG;
        const float* ub = U + (c / u_div) * NB;
        const float* vb = V + c * NB;
        int ri[EPG], ci[EPG];
#pragma unroll
        for (int e = 0; e < EPG; ++e) {
            const int64_t p = p0 + e < nnzp ? p0 + e : nnzp - 1;
            ri[e] = (int)row[p];
            ci[e] = (int)col[p];
        }
        float acc[EPG];
#pragma unroll
        for (int e = 0; e < EPG; ++e) acc[e] = 0.f;
        for (int b = lane * 4; b < B; b += LQ * 4) {
            float4 uu[EPG], vv[EPG];
#pragma unroll
            for (int e = 0; e < EPG; ++e) {
                uu[e] = *reinterpret_cast<const float4*>(ub + (int64_t)ri[e] * B + b);
                vv[e] = *reinterpret_cast<const float4*>(vb + (int64_t)ci[e] * B + b);
            }
#pragma unroll
            for (int e = 0; e < EPG; ++e) {
                acc[e] = fmaf(uu[e].x, vv[e].x, acc[e]);
                acc[e] = fmaf(uu[e].y, vv[e].y, acc[e]);
                acc[e] = fmaf(uu[e].z, vv[e].z, acc[e]);
                acc[e] = fmaf(uu[e].w, vv[e].w, acc[e]);
            }
        }
        const int f = (int)(c / G), g = (int)(c - (int64_t)f * G);
        float* dst = dwedge + ((int64_t)(f * K1 + kidx) * G + g) * nnzp;
#pragma unroll
        for (int e = 0; e < EPG; ++e) {
            const float tot = group_sum<LQ>(acc[e]);
            if (lane == 0 && p0 + e < nnzp) dst[p0 + e] = tot;
        }
    }
}

// dwdiag[c][n] = sum_b U0[c/u_div][n][b] * Xt[g(c)][n][b]
template <int LB>
__global__ __launch_bounds__(kThreads) void ev_dwdiag_kernel(const float* __restrict__ U0, const float* __restrict__ Xt,
                                                             float* __restrict__ dwdiag, int N, int B, int G, int u_div,
                                                             int64_t groups) {
    const int64_t NB = (int64_t)N * B;
    const int lane = threadIdx.x % LB;
    const int64_t g0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / LB;
    const int64_t gstep = (int64_t)gridDim.x * kThreads / LB;
    for (int64_t grp = g0; grp < groups; grp += gstep) {
        const int64_t c = grp / N, n = grp - c * N;
        const float* u = U0 + (c / u_div) * NB + n * B;
        const float* x = Xt + (c % G) * NB + n * B;
        float acc = 0.f;
        for (int b = lane; b < B; b += LB) acc = fmaf(u[b], x[b], acc);
        acc = group_sum<LB>(acc);
        if (lane == 0) dwdiag[grp] = acc;
    }
}

// 16-byte loads, LQ = B/4 lanes per output (B % 4 == 0)
template <int LQ>
__global__ __launch_bounds__(kThreads) void ev_dwdiag4_kernel(const float* __restrict__ U0, const float* __restrict__ Xt,
                                                              float* __restrict__ dwdiag, int N, int B, int G, int u_div,
                                                              int64_t groups) {
    const int64_t NB = (int64_t)N * B;
    const int lane = threadIdx.x % LQ;
    const int64_t g0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / LQ;
    const int64_t gstep = (int64_t)gridDim.x * kThreads / LQ;
    for (int64_t grp = g0; grp < groups; grp += gstep) {
        const int64_t c = grp / N, n = grp - c * N;
        const float* u = U0 + (c / u_div) * NB + n * B;
        const float* x = Xt + (c % G) * NB + n * B;
        float acc = 0.f;
        for (int b = lane * 4; b < B; b += LQ * 4) {
            const float4 uu = *reinterpret_cast<const float4*>(u + b), xx = *reinterpret_cast<const float4*>(x + b);
            acc = fmaf(uu.x, xx.x, acc);
            acc = fmaf(uu.y, xx.y, acc);
            acc = fmaf(uu.z, xx.z, acc);
            acc = fmaf(uu.w, xx.w, acc);
        }
        acc = group_sum<LQ>(acc);
        if (lane == 0) dwdiag[grp] = acc;
    }
}

// dXt[g][n][b] = sum_f wdiag[f*G+g][n] * U0[(f*G+g)/u_div][n][b]
// VEC batch entries per thread (16-byte loads when N*B % 4 == 0 and B % 4 == 0), g on blockIdx.y: 32-bit indices inside a plane.
template <int VEC>
__global__ __launch_bounds__(kThreads) void ev_dxt_kernel(const float* __restrict__ wdiag, const float* __restrict__ U0,
                                                          float* __restrict__ dXt, int N, int B, int G, int F, int u_div) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int g = blockIdx.y;
    const int64_t NB = (int64_t)N * B;
    const int NBv = (int)(NB / VEC), Bv = B / VEC;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < NBv; i += gridDim.x * kThreads) {
        const int n = i / Bv;
        vec_t acc = 0.f;
        for (int f = 0; f < F; ++f) {
            const int64_t c = (int64_t)f * G + g;
            acc += wdiag[c * N + n] * *reinterpret_cast<const vec_t*>(U0 + (c / u_div) * NB + (int64_t)i * VEC);
        }
        *reinterpret_cast<vec_t*>(dXt + (int64_t)g * NB + (int64_t)i * VEC) = acc;
    }
}

// ---- dbias[f] = sum_{n,b} Dyt[f][n][b]   (graphML.py:486-487 under autograd) ---------------------------------------------------
// Two fixed-order stages: blockIdx.y = one of kBiasSlices contiguous slices of the N*B sum (32 workgroups, one per feature, left
// 7/8 of the chip idle for 0.77 ms at config 5), then the slices' partial sums in index order.
constexpr int kBiasSlices = 64;
__global__ __launch_bounds__(kThreads) void ev_dbias_kernel(const float* __restrict__ Dyt, float* __restrict__ partial, int64_t NB) {
    __shared__ float part[kThreads];
    const int nS = (int)gridDim.y;
    const int64_t per = (NB + nS - 1) / nS;
    const int64_t lo = (int64_t)blockIdx.y * per, hi = lo + per < NB ? lo + per : NB;
    const float* d = Dyt + (int64_t)blockIdx.x * NB;
    float acc = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) acc += d[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x * nS + blockIdx.y] = part[0];
}

__global__ void ev_dbias_finish_kernel(const float* __restrict__ partial, float* __restrict__ dbias, int F, int nS) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float acc = 0.f;
    for (int s = 0; s < nS; ++s) acc += partial[f * nS + s];
    dbias[f] = acc;
}

int lanes_for_batch(int B) {
    int lb = 1;
    while (lb < B && lb < 64) lb <<= 1;
    return lb;
}


int launch_ev_hop(const int32_t* rowptr, const int32_t* col, const int32_t* vidx, const float* wedge, const float* in, const float* add,
                  float* out, int N, int B, int G, int C, int K1, int kidx, int64_t nnzp, int in_div, int add_div, hipStream_t st,
                  const uint16_t* col16 = nullptr) {
    const int64_t NB = (int64_t)N * B, CNB = (int64_t)C * NB;
    const int lbw = lanes_for_batch(B);
    const int rpw = std::max(kRowsPerWG, kThreads / lbw);  // rows per workgroup (RPW of ev_hop_lds_kernel)
    const int nRowBlocks = (N + rpw - 1) / rpw;
    const int64_t nblk = (int64_t)((C + 7) / 8) * 8 * nRowBlocks;
    const int env_generic = g_tune.evgf_generic;
    if (B % 4 == 0 && B <= 256 && env_generic == 0) {  // 16-byte gathers: 4 batch entries per thread
        const int lq = lanes_for_batch(B / 4);
        const int rpw4 = std::max(kRowsPerWG, kThreads / lq);
        const int nrb4 = (N + rpw4 - 1) / rpw4;
        const int64_t nblk4 = (int64_t)((C + 7) / 8) * 8 * nrb4;
        if (nblk4 < (int64_t)INT32_MAX) {
#define GF_EVHOP4(LQV)                                                                                                        \
    if (col16 && g_tune.evgf_idx16)                                                                                           \
        hipLaunchKernelGGL((ev_hop_lds4_kernel<LQV, uint16_t>), dim3((unsigned)nblk4), dim3(kThreads), 0, st, rowptr, col16, vidx, wedge, in, \
                           add, out, N, B, G, K1, kidx, nnzp, in_div, add_div, nrb4, C);                                      \
    else                                                                                                                      \
        hipLaunchKernelGGL((ev_hop_lds4_kernel<LQV, int32_t>), dim3((unsigned)nblk4), dim3(kThreads), 0, st, rowptr, col, vidx, wedge, in, \
                           add, out, N, B, G, K1, kidx, nnzp, in_div, add_div, nrb4, C)
            switch (lq) {
                case 1: GF_EVHOP4(1); break;
                case 2: GF_EVHOP4(2); break;
                case 4: GF_EVHOP4(4); break;
                case 8: GF_EVHOP4(8); break;
                case 16: GF_EVHOP4(16); break;
                case 32: GF_EVHOP4(32); break;
                default: GF_EVHOP4(64); break;
            }
#undef GF_EVHOP4
            GF_LAUNCH_CHECK("ev_hop_lds4_kernel");
            return GF_OK;
        }
    }
    if (B <= 64 && nblk < (int64_t)INT32_MAX && env_generic != 1) {
        const int lb = lanes_for_batch(B);
#define GF_EVHOP(LBV)                                                                                                         \
    hipLaunchKernelGGL((ev_hop_lds_kernel<LBV>), dim3((unsigned)nblk), dim3(kThreads), 0, st, rowptr, col, vidx, wedge, in, add, out, \
                       N, B, G, K1, kidx, nnzp, in_div, add_div, nRowBlocks, C)
        switch (lb) {
            case 1: GF_EVHOP(1); break;
            case 2: GF_EVHOP(2); break;
            case 4: GF_EVHOP(4); break;
            case 8: GF_EVHOP(8); break;
            case 16: GF_EVHOP(16); break;
            case 32: GF_EVHOP(32); break;
            default: GF_EVHOP(64); break;
        }
#undef GF_EVHOP
        GF_LAUNCH_CHECK("ev_hop_lds_kernel");
        return GF_OK;
    }
    hipLaunchKernelGGL(ev_hop_kernel, dim3(grid_for(CNB)), dim3(kThreads), 0, st, rowptr, col, vidx, wedge, in, add, out, N, B, G, K1,
                       kidx, nnzp, in_div, add_div, CNB);
    GF_LAUNCH_CHECK("ev_hop_kernel");
    return GF_OK;
}

template <class T>
int upload_vec(const std::vector<T>& h, T** d, int64_t& bytes) {
    const size_t nb = std::max<size_t>(h.size(), 1) * sizeof(T);
    GF_HIP(hipMalloc((void**)d, nb));
    if (!h.empty()) GF_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    bytes += (int64_t)nb;
    return GF_OK;
}

struct EvDims {
    int N;
    int64_t NB, CNB;
    int C;
};

int check_common(const char* who, const gf_ev_plan* plan, int B, int G, int F, int K, int Nin) {
    GF_REQUIRE_ARG(plan != nullptr, "%s: plan is NULL", who);
    GF_REQUIRE_SHAPE(B > 0 && G > 0 && F > 0 && K > 0 && Nin > 0, "%s: bad shape B=%d G=%d F=%d K=%d Nin=%d", who, B, G, F, K, Nin);
    GF_REQUIRE_SHAPE(Nin <= plan->n, "%s: signal has %d nodes, pattern has %d", who, Nin, plan->n);  // graphML.py:2678 only pads
    GF_REQUIRE_SHAPE((int64_t)F * G < (int64_t)INT32_MAX / 2, "%s: F*G too large", who);
    return GF_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
extern "C" int gf_ev_plan_create(int32_t n, int64_t nnzp, const int32_t* rowptr, const int32_t* colidx, gf_ev_plan** out) {
    GF_REQUIRE_ARG(out != nullptr, "gf_ev_plan_create: out_plan is NULL");
    *out = nullptr;
    GF_REQUIRE_ARG(rowptr && (nnzp == 0 || colidx), "gf_ev_plan_create: NULL CSR array");
    GF_REQUIRE_SHAPE(n > 0, "gf_ev_plan_create: n_nodes = %d must be positive", n);
    GF_REQUIRE_SHAPE(nnzp >= 0 && nnzp < (int64_t)INT32_MAX, "gf_ev_plan_create: nnzp = %lld outside [0, 2^31)", (long long)nnzp);
    GF_REQUIRE_SHAPE(rowptr[0] == 0 && rowptr[n] == nnzp, "gf_ev_plan_create: rowptr[0] = %d, rowptr[N] = %d, nnzp = %lld",
                     rowptr[0], rowptr[n], (long long)nnzp);
    for (int32_t i = 0; i < n; ++i) {
        GF_REQUIRE_SHAPE(rowptr[i] <= rowptr[i + 1], "gf_ev_plan_create: rowptr not monotone at row %d", i);
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
            GF_REQUIRE_SHAPE(colidx[q] >= 0 && colidx[q] < n, "gf_ev_plan_create: column %d at %d outside [0, %d)", colidx[q], q, n);
            GF_REQUIRE_SHAPE(q == rowptr[i] || colidx[q] > colidx[q - 1],
                             "gf_ev_plan_create: row %d is not strictly ascending (entry order defines the value index)", i);
        }
    }
    gf_ev_plan* pl = new (std::nothrow) gf_ev_plan();
    if (!pl) {
        gf_set_error("gf_ev_plan_create: out of host memory");
        return GF_ERR_NOMEM;
    }
    int rc = GF_OK;
    try {
        std::vector<int32_t> rp(rowptr, rowptr + n + 1), ci(colidx, colidx + nnzp), row(nnzp);
        for (int32_t i = 0; i < n; ++i)
            for (int32_t q = rp[i]; q < rp[i + 1]; ++q) row[q] = i;
        std::vector<int32_t> trp(n + 1, 0), tci(nnzp), tvi(nnzp);
        for (int64_t q = 0; q < nnzp; ++q) trp[ci[q] + 1]++;
        for (int32_t i = 0; i < n; ++i) trp[i + 1] += trp[i];
        std::vector<int32_t> cur(trp.begin(), trp.end() - 1);
        for (int64_t q = 0; q < nnzp; ++q) {  // ascending p => ascending original row within each transposed row
            const int32_t t = cur[ci[q]]++;
            tci[t] = row[q];
            tvi[t] = (int32_t)q;
        }
        pl->n = n;
        pl->nnzp = nnzp;
        if ((rc = upload_vec(rp, &pl->rowptr, pl->device_bytes)) == GF_OK && (rc = upload_vec(ci, &pl->col, pl->device_bytes)) == GF_OK &&
            (rc = upload_vec(row, &pl->row, pl->device_bytes)) == GF_OK && (rc = upload_vec(trp, &pl->t_rowptr, pl->device_bytes)) == GF_OK &&
            (rc = upload_vec(tci, &pl->t_col, pl->device_bytes)) == GF_OK)
            rc = upload_vec(tvi, &pl->t_vidx, pl->device_bytes);
        if (rc == GF_OK && n <= 65535) {
            std::vector<uint16_t> c16(ci.begin(), ci.end()), r16(row.begin(), row.end()), t16(tci.begin(), tci.end());
            if ((rc = upload_vec(c16, &pl->col16, pl->device_bytes)) == GF_OK && (rc = upload_vec(r16, &pl->row16, pl->device_bytes)) == GF_OK)
                rc = upload_vec(t16, &pl->t_col16, pl->device_bytes);
        }
    } catch (const std::bad_alloc&) {
        gf_set_error("gf_ev_plan_create: out of host memory");
        rc = GF_ERR_NOMEM;
    }
    if (rc != GF_OK) {
        gf_ev_plan_destroy(pl);
        return rc;
    }
    *out = pl;
    return GF_OK;
}

extern "C" int gf_ev_plan_destroy(gf_ev_plan* pl) {
    if (!pl) return GF_OK;
    for (int32_t* p : {pl->rowptr, pl->col, pl->row, pl->t_rowptr, pl->t_col, pl->t_vidx})
        if (p) (void)hipFree(p);
    for (uint16_t* p : {pl->col16, pl->row16, pl->t_col16})
        if (p) (void)hipFree(p);
    delete pl;
    return GF_OK;
}

extern "C" int gf_ev_plan_info(const gf_ev_plan* pl, int32_t* n, int64_t* nnzp, int64_t* device_bytes) {
    GF_REQUIRE_ARG(pl != nullptr, "gf_ev_plan_info: plan is NULL");
    if (n) *n = pl->n;
    if (nnzp) *nnzp = pl->nnzp;
    if (device_bytes) *device_bytes = pl->device_bytes;
    return GF_OK;
}

extern "C" size_t gf_evgf_scratch_floats(int32_t B, int32_t G, int32_t F, int32_t N, int32_t backward) {
    if (B <= 0 || G <= 0 || F <= 0 || N <= 0) return 0;
    const size_t NB = (size_t)N * B;
    return backward ? NB * ((size_t)2 * G + F + (size_t)2 * F * G) : NB * ((size_t)G + F);
}

extern "C" int gf_evgf_forward(const gf_ev_plan* plan, const float* x, const float* wdiag, const float* wedge, const float* bias,
                               float* V, float* scratch, float* y, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin,
                               void* stream) {
    int rc = check_common("gf_evgf_forward", plan, B, G, F, K, Nin);
    if (rc != GF_OK) return rc;
    GF_REQUIRE_ARG(x && wdiag && V && scratch && y && (K == 1 || wedge), "gf_evgf_forward: NULL tensor");
    const int N = plan->n, C = F * G;
    const int64_t NB = (int64_t)N * B, CNB = (int64_t)C * NB;
    hipStream_t st = gf_stream(stream);
    float* Xt = scratch;
    float* Yt = scratch + (int64_t)G * NB;
    if ((rc = to_nodebatch(x, Xt, B, G, Nin, N, st)) != GF_OK) return rc;
    if (B % 4 == 0 && C <= 65535 && NB < (int64_t)INT32_MAX)
        hipLaunchKernelGGL(ev_tap0_v4_kernel, dim3((unsigned)std::min<int64_t>(64, (NB / 4 + kThreads - 1) / kThreads), C), dim3(kThreads), 0, st,
                           wdiag, Xt, V, G, N, B / 4);
    else
        hipLaunchKernelGGL(ev_tap0_kernel, dim3(grid_for(CNB)), dim3(kThreads), 0, st, wdiag, Xt, V, G, NB, B, CNB);
    GF_LAUNCH_CHECK("ev_tap0_kernel");
    for (int k = 1; k < K; ++k) {
        rc = launch_ev_hop(plan->rowptr, plan->col, nullptr, wedge, V + (int64_t)(k - 1) * CNB, nullptr, V + (int64_t)k * CNB, N, B,
                           G, C, K - 1, k - 1, plan->nnzp, 1, 1, st, plan->col16);
        if (rc != GF_OK) return rc;
    }
    const int64_t FNB = (int64_t)F * NB;
    if (NB % 4 == 0)
        hipLaunchKernelGGL(ev_sum_kernel<4>, dim3(grid_for(FNB / 4)), dim3(kThreads), 0, st, V, bias, Yt, G, K, NB, CNB, FNB / 4);
    else
        hipLaunchKernelGGL(ev_sum_kernel<1>, dim3(grid_for(FNB)), dim3(kThreads), 0, st, V, bias, Yt, G, K, NB, CNB, FNB);
    GF_LAUNCH_CHECK("ev_sum_kernel");
    return from_nodebatch(Yt, y, B, F, N, Nin, st);
}

extern "C" int gf_evgf_backward(const gf_ev_plan* plan, const float* dy, const float* x, const float* wdiag, const float* wedge,
                                const float* V, float* scratch, float* dx, float* dwdiag, float* dwedge, float* dbias, int32_t B,
                                int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream) {
    int rc = check_common("gf_evgf_backward", plan, B, G, F, K, Nin);
    if (rc != GF_OK) return rc;
    GF_REQUIRE_ARG(dy && x && wdiag && V && scratch && (K == 1 || wedge), "gf_evgf_backward: NULL tensor");
    const int N = plan->n, C = F * G;
    const int64_t NB = (int64_t)N * B, CNB = (int64_t)C * NB;
    hipStream_t st = gf_stream(stream);
    float* Xt = scratch;
    float* dXt = Xt + (int64_t)G * NB;
    float* Dyt = dXt + (int64_t)G * NB;
    float* Ubuf[2] = {Dyt + (int64_t)F * NB, Dyt + (int64_t)F * NB + CNB};
    if ((rc = to_nodebatch(x, Xt, B, G, Nin, N, st)) != GF_OK) return rc;
    if ((rc = to_nodebatch(dy, Dyt, B, F, Nin, N, st)) != GF_OK) return rc;
    if (dbias) {
        // partial sums live in the (not yet used) dXt region of the scratch (G * N * B floats); tiny problems reduce in one stage
        int nS = kBiasSlices;
        if ((int64_t)F * nS > (int64_t)G * NB) nS = 1;
        hipLaunchKernelGGL(ev_dbias_kernel, dim3(F, nS), dim3(kThreads), 0, st, Dyt, nS > 1 ? dXt : dbias, NB);
        GF_LAUNCH_CHECK("ev_dbias_kernel");
        if (nS > 1) {
            hipLaunchKernelGGL(ev_dbias_finish_kernel, dim3((F + 63) / 64), dim3(64), 0, st, dXt, dbias, F, nS);
            GF_LAUNCH_CHECK("ev_dbias_finish_kernel");
        }
    }
    // u_{K-1}^{fg} = dy_f : read Dyt through the chain divisor G instead of materialising the broadcast
    const float* Ucur = Dyt;
    int udiv = G, pp = 0;
    const int lb = lanes_for_batch(B);
    for (int k = K - 1; k >= 1; --k) {
        if (dwedge) {
            const int64_t epc = (plan->nnzp + 3) / 4;  // entry quads per chain (EPG = 4 in ev_sddmm_kernel)
            const int64_t chainSlots = (int64_t)((C + 7) / 8) * 8;  // chain c runs on XCD c % 8
            auto sddmm_grid = [&](int lanes) {  // lane groups of `lanes` lanes, 4 quads each (ROUNDS)
                const int64_t perWG = (int64_t)(kThreads / lanes) * 4;
                return (unsigned)(chainSlots * ((epc + perWG - 1) / perWG));
            };
            GF_REQUIRE_SHAPE(chainSlots * ((epc + 3) / 4) < (int64_t)INT32_MAX, "gf_evgf_backward: %lld chains x %lld entries exceed the launch grid",
                             (long long)C, (long long)plan->nnzp);
            const int env_scalar = g_tune.evgf_generic;
            if (B % 4 == 0 && B <= 256 && env_scalar == 0) {
                const int lq = lanes_for_batch(B / 4);
                const unsigned grid4 = sddmm_grid(lq);
#define GF_SDDMM4(LQV)                                                                                                        \
    if (plan->col16 && g_tune.evgf_idx16)                                                                                     \
        hipLaunchKernelGGL((ev_sddmm4_kernel<LQV, uint16_t>), dim3(grid4), dim3(kThreads), 0, st, plan->row16, plan->col16, Ucur, \
                           V + (int64_t)(k - 1) * CNB, dwedge, N, B, G, K - 1, k - 1, plan->nnzp, udiv, (int64_t)C);          \
    else                                                                                                                      \
        hipLaunchKernelGGL((ev_sddmm4_kernel<LQV, int32_t>), dim3(grid4), dim3(kThreads), 0, st, plan->row, plan->col, Ucur,   \
                           V + (int64_t)(k - 1) * CNB, dwedge, N, B, G, K - 1, k - 1, plan->nnzp, udiv, (int64_t)C)
                switch (lq) {
                    case 1: GF_SDDMM4(1); break;
                    case 2: GF_SDDMM4(2); break;
                    case 4: GF_SDDMM4(4); break;
                    case 8: GF_SDDMM4(8); break;
                    case 16: GF_SDDMM4(16); break;
                    case 32: GF_SDDMM4(32); break;
                    default: GF_SDDMM4(64); break;
                }
#undef GF_SDDMM4
                GF_LAUNCH_CHECK("ev_sddmm4_kernel");
            } else {
            const unsigned grid = sddmm_grid(lb);
#define GF_SDDMM(LBV)                                                                                                         \
    hipLaunchKernelGGL((ev_sddmm_kernel<LBV>), dim3(grid), dim3(kThreads), 0, st, plan->row, plan->col, Ucur,                  \
                       V + (int64_t)(k - 1) * CNB, dwedge, N, B, G, K - 1, k - 1, plan->nnzp, udiv, (int64_t)C)
            switch (lb) {
                case 1: GF_SDDMM(1); break;
                case 2: GF_SDDMM(2); break;
                case 4: GF_SDDMM(4); break;
                case 8: GF_SDDMM(8); break;
                case 16: GF_SDDMM(16); break;
                case 32: GF_SDDMM(32); break;
                default: GF_SDDMM(64); break;
            }
#undef GF_SDDMM
            GF_LAUNCH_CHECK("ev_sddmm_kernel");
            }
        }
        if (k > 1 || dwdiag || dx) {  // u_{k-1} = Phi_k^T u_k + dy_f
            rc = launch_ev_hop(plan->t_rowptr, plan->t_col, plan->t_vidx, wedge, Ucur, Dyt, Ubuf[pp], N, B, G, C, K - 1, k - 1,
                               plan->nnzp, udiv, G, st, plan->t_col16);
            if (rc != GF_OK) return rc;
            Ucur = Ubuf[pp];
            udiv = 1;
            pp ^= 1;
        }
    }
    if (dwdiag) {
        const int64_t groups = (int64_t)C * N;
        if (B % 4 == 0 && B <= 256) {
            const int lq = lanes_for_batch(B / 4);
            const unsigned grid4 = grid_for(groups * lq);
#define GF_DWD4(LQV) \
    hipLaunchKernelGGL((ev_dwdiag4_kernel<LQV>), dim3(grid4), dim3(kThreads), 0, st, Ucur, Xt, dwdiag, N, B, G, udiv, groups)
            switch (lq) {
                case 1: GF_DWD4(1); break;
                case 2: GF_DWD4(2); break;
                case 4: GF_DWD4(4); break;
                case 8: GF_DWD4(8); break;
                case 16: GF_DWD4(16); break;
                case 32: GF_DWD4(32); break;
                default: GF_DWD4(64); break;
            }
#undef GF_DWD4
            GF_LAUNCH_CHECK("ev_dwdiag4_kernel");
        } else {
        const unsigned grid = grid_for(groups * lb);
#define GF_DWD(LBV) \
    hipLaunchKernelGGL((ev_dwdiag_kernel<LBV>), dim3(grid), dim3(kThreads), 0, st, Ucur, Xt, dwdiag, N, B, G, udiv, groups)
        switch (lb) {
            case 1: GF_DWD(1); break;
            case 2: GF_DWD(2); break;
            case 4: GF_DWD(4); break;
            case 8: GF_DWD(8); break;
            case 16: GF_DWD(16); break;
            case 32: GF_DWD(32); break;
            default: GF_DWD(64); break;
        }
#undef GF_DWD
        GF_LAUNCH_CHECK("ev_dwdiag_kernel");
        }
    }
    if (dx) {
        GF_REQUIRE_SHAPE(G <= 65535 && NB < (int64_t)INT32_MAX, "gf_evgf_backward: G = %d / N*B = %lld exceed the launch grid", G, (long long)NB);
        if (B % 4 == 0)
            hipLaunchKernelGGL(ev_dxt_kernel<4>, dim3((unsigned)std::min<int64_t>(256, (NB / 4 + kThreads - 1) / kThreads), G), dim3(kThreads), 0,
                               st, wdiag, Ucur, dXt, N, B, G, F, udiv);
        else
            hipLaunchKernelGGL(ev_dxt_kernel<1>, dim3((unsigned)std::min<int64_t>(256, (NB + kThreads - 1) / kThreads), G), dim3(kThreads), 0, st,
                               wdiag, Ucur, dXt, N, B, G, F, udiv);
        GF_LAUNCH_CHECK("ev_dxt_kernel");
        rc = from_nodebatch(dXt, dx, B, G, N, Nin, st);
    }
    return rc;
}
