// gf_layout.hip -- boundary layout kernels between the reference layout x[B,G,N] (node index contiguous,
// reference graphML.py:108-109) and the node-major layout X[B,N,G] the gather kernels want (one contiguous
// feature row per (b, n): a full 128-byte line at G = 32).
//
// bgn_to_bng also implements GraphFilter.forward's zero-padding to N nodes (reference graphML.py:2131-2135):
// rows n >= Nin of X are written as zeros.  bng_to_bgn keeps nodes n < Nout (graphML.py:2142-2143).
//
// HBM-bound: algorithmic bytes = 4*B*G*(Nin + N).  32x32 tiles through LDS, both sides coalesced.
#include "gf_common.h"

namespace {

constexpr int TILE = 32;
constexpr int TROWS = 8;  // 32 x 8 threads

// in:  [B, R, C] (C contiguous, only columns c < Cin exist, leading dim Cin)
// out: [B, Cout_rows, R] viewed as out[b][c][r] for c < Cout (leading dim R); c >= Cin reads as zero.
// With (R, C) = (G, N) this is x[B,G,Nin] -> X[B,N,G];  with (R, C) = (N, G) and swapped roles it is the inverse.
__global__ __launch_bounds__(TILE* TROWS) void transpose_pad_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                      int R, int Cin, int Cout, int64_t in_bstride,
                                                                      int64_t out_bstride, const float* __restrict__ mask) {
    __shared__ float tile[TILE][TILE + 1];
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * TILE;
    const int r0 = blockIdx.y * TILE;
    const float* ib = in + (int64_t)b * in_bstride;
    float* ob = out + (int64_t)b * out_bstride;
    const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int i = 0; i < TILE; i += TROWS) {
        const int r = r0 + ty + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < Cin) {
            v = ib[(int64_t)r * Cin + c];
            // mask = the saved ReLU output, same layout as `in`: gradient of the fused epilogue (dy where y > 0, else 0)
            if (mask != nullptr && !(mask[(int64_t)b * in_bstride + (int64_t)r * Cin + c] > 0.f)) v = 0.f;
        }
        tile[ty + i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TILE; i += TROWS) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (c < Cout && r < R) ob[(int64_t)c * R + r] = tile[tx][ty + i];
    }
}

// The same transposition with 16-byte accesses on both sides (Cin % 4 == 0, R % 4 == 0, 16-byte aligned tensors): a TR x TC tile,
// rows read as float4 along c (512-byte segments for TC = 128), columns written as float4 along r.  Config 4's layout passes
// (3.3 GB each): 784 -> ~600 us.
template <int TR, int TC>
__global__ __launch_bounds__(256) void transpose_pad_v4_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cin,
                                                               int Cout, int64_t in_bstride, int64_t out_bstride,
                                                               const float* __restrict__ mask) {
    __shared__ float tile[TR][TC + 1];
    constexpr int C4 = TC / 4, R4 = TR / 4;
    const int b = blockIdx.z, c0 = blockIdx.x * TC, r0 = blockIdx.y * TR, tid = threadIdx.x;
    const float* ib = in + (int64_t)b * in_bstride;
    float* ob = out + (int64_t)b * out_bstride;
#pragma unroll
    for (int idx = tid; idx < TR * C4; idx += 256) {
        const int rr = idx / C4, c4 = idx - rr * C4, r = r0 + rr, c = c0 + 4 * c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < Cin) {
            v = *reinterpret_cast<const float4*>(ib + (int64_t)r * Cin + c);
            if (mask != nullptr) {  // gradient of the fused ReLU epilogue: dy where the saved output is > 0
                const float4 m = *reinterpret_cast<const float4*>(mask + (int64_t)b * in_bstride + (int64_t)r * Cin + c);
                if (!(m.x > 0.f)) v.x = 0.f;
                if (!(m.y > 0.f)) v.y = 0.f;
                if (!(m.z > 0.f)) v.z = 0.f;
                if (!(m.w > 0.f)) v.w = 0.f;
            }
        }
        tile[rr][4 * c4] = v.x, tile[rr][4 * c4 + 1] = v.y, tile[rr][4 * c4 + 2] = v.z, tile[rr][4 * c4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int idx = tid; idx < TC * R4; idx += 256) {
        const int cc = idx / R4, r4 = idx - cc * R4, c = c0 + cc, r = r0 + 4 * r4;
        if (c < Cout && r < R)
            *reinterpret_cast<float4*>(ob + (int64_t)c * R + r) =
                make_float4(tile[4 * r4][cc], tile[4 * r4 + 1][cc], tile[4 * r4 + 2][cc], tile[4 * r4 + 3][cc]);
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int kMaxGridZ = 65535;

// in [nb, R, Cin] -> out [nb, Cout, R] (see transpose_pad_kernel); picks the 16-byte version when shapes and pointers allow
void launch_transpose(const float* in, float* out, int nb, int R, int Cin, int Cout, int64_t in_bstride, int64_t out_bstride,
                      const float* mask, hipStream_t st) {
    const bool v4 = Cin % 4 == 0 && R % 4 == 0 && in_bstride % 4 == 0 && out_bstride % 4 == 0 && aligned16(in) && aligned16(out) &&
                    (mask == nullptr || aligned16(mask));
    if (v4 && R <= 64) {  // few long rows (x[B,G,N] -> X[B,N,G])
        dim3 grid((Cout + 127) / 128, (R + 31) / 32, nb);
        hipLaunchKernelGGL((transpose_pad_v4_kernel<32, 128>), grid, dim3(256), 0, st, in, out, R, Cin, Cout, in_bstride, out_bstride, mask);
    } else if (v4 && Cout <= 64) {  // many short rows (X[B,N,G] -> x[B,G,N])
        dim3 grid((Cout + 31) / 32, (R + 127) / 128, nb);
        hipLaunchKernelGGL((transpose_pad_v4_kernel<128, 32>), grid, dim3(256), 0, st, in, out, R, Cin, Cout, in_bstride, out_bstride, mask);
    } else {
        dim3 grid((Cout + TILE - 1) / TILE, (R + TILE - 1) / TILE, nb), block(TILE, TROWS);
        hipLaunchKernelGGL(transpose_pad_kernel, grid, block, 0, st, in, out, R, Cin, Cout, in_bstride, out_bstride, mask);
    }
}

}  // namespace

extern "C" int gf_layout_bgn_to_bng(const float* x, float* X, int32_t B, int32_t G, int32_t Nin, int32_t N, void* stream) {
    GF_REQUIRE_ARG(x && X, "gf_layout_bgn_to_bng: NULL tensor");
    GF_REQUIRE_SHAPE(B > 0 && G > 0 && Nin > 0 && N >= Nin, "gf_layout_bgn_to_bng: bad shape B=%d G=%d Nin=%d N=%d", B, G,
                     Nin, N);
    // the batch rides on gridDim.z (<= 65535): recurrent layers fold B*T into the batch, so longer batches go in slices
    for (int32_t b0 = 0; b0 < B; b0 += kMaxGridZ) {
        const int32_t nb = B - b0 < kMaxGridZ ? B - b0 : kMaxGridZ;
        launch_transpose(x + (int64_t)b0 * G * Nin, X + (int64_t)b0 * N * G, nb, G, Nin, N, (int64_t)G * Nin, (int64_t)N * G, nullptr,
                         gf_stream(stream));
        GF_LAUNCH_CHECK("transpose_pad_kernel(bgn->bng)");
    }
    return GF_OK;
}

extern "C" int gf_layout_bng_to_bgn(const float* X, float* x, int32_t B, int32_t G, int32_t N, int32_t Nout, void* stream) {
    GF_REQUIRE_ARG(x && X, "gf_layout_bng_to_bgn: NULL tensor");
    GF_REQUIRE_SHAPE(B > 0 && G > 0 && Nout > 0 && N >= Nout, "gf_layout_bng_to_bgn: bad shape B=%d G=%d N=%d Nout=%d", B,
                     G, N, Nout);
    // in = X viewed [B, R = N(only first Nout rows used), C = G]; out[b][g][n] has leading dim Nout, so run the
    // kernel with R = Nout rows of the input (input batch stride still N*G).
    for (int32_t b0 = 0; b0 < B; b0 += kMaxGridZ) {
        const int32_t nb = B - b0 < kMaxGridZ ? B - b0 : kMaxGridZ;
        launch_transpose(X + (int64_t)b0 * N * G, x + (int64_t)b0 * G * Nout, nb, Nout, G, G, (int64_t)N * G, (int64_t)G * Nout, nullptr,
                         gf_stream(stream));
        GF_LAUNCH_CHECK("transpose_pad_kernel(bng->bgn)");
    }
    return GF_OK;
}

// dy [B,F,Nin] (reference layout) -> P0 [B,N,F] node-major with the ReLU mask of the saved output y applied on the way
int gf_layout_masked_launch(const float* dy, const float* y, float* X, int B, int G, int Nin, int N, hipStream_t st) {
    for (int b0 = 0; b0 < B; b0 += kMaxGridZ) {
        const int nb = B - b0 < kMaxGridZ ? B - b0 : kMaxGridZ;
        launch_transpose(dy + (int64_t)b0 * G * Nin, X + (int64_t)b0 * N * G, nb, G, Nin, N, (int64_t)G * Nin, (int64_t)N * G,
                         y ? y + (int64_t)b0 * G * Nin : nullptr, st);
        GF_LAUNCH_CHECK("transpose_pad_kernel(masked)");
    }
    return GF_OK;
}
