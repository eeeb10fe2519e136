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

constexpr int kMaxGridZ = 65535;

}  // namespace

extern "C" int gf_layout_bgn_to_bng(const float* x, float* X, int32_t B, int32_t G, int32_t Nin, int32_t N, void* stream) {
    GF_REQUIRE_ARG(x && X, "gf_layout_bgn_to_bng: NULL tensor");
    GF_REQUIRE_SHAPE(B > 0 && G > 0 && Nin > 0 && N >= Nin, "gf_layout_bgn_to_bng: bad shape B=%d G=%d Nin=%d N=%d", B, G,
                     Nin, N);
    // the batch rides on gridDim.z (<= 65535): recurrent layers fold B*T into the batch, so longer batches go in slices
    for (int32_t b0 = 0; b0 < B; b0 += kMaxGridZ) {
        const int32_t nb = B - b0 < kMaxGridZ ? B - b0 : kMaxGridZ;
        dim3 grid((N + TILE - 1) / TILE, (G + TILE - 1) / TILE, nb), block(TILE, TROWS);
        hipLaunchKernelGGL(transpose_pad_kernel, grid, block, 0, gf_stream(stream), x + (int64_t)b0 * G * Nin, X + (int64_t)b0 * N * G, G, Nin,
                           N, (int64_t)G * Nin, (int64_t)N * G, (const float*)nullptr);
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
        dim3 grid((G + TILE - 1) / TILE, (Nout + TILE - 1) / TILE, nb), block(TILE, TROWS);
        hipLaunchKernelGGL(transpose_pad_kernel, grid, block, 0, gf_stream(stream), X + (int64_t)b0 * N * G, x + (int64_t)b0 * G * Nout, Nout, G,
                           G, (int64_t)N * G, (int64_t)G * Nout, (const float*)nullptr);
        GF_LAUNCH_CHECK("transpose_pad_kernel(bng->bgn)");
    }
    return GF_OK;
}

// dy [B,F,Nin] (reference layout) -> P0 [B,N,F] node-major with the ReLU mask of the saved output y applied on the way
int gf_layout_masked_launch(const float* dy, const float* y, float* X, int B, int G, int Nin, int N, hipStream_t st) {
    for (int b0 = 0; b0 < B; b0 += kMaxGridZ) {
        const int nb = B - b0 < kMaxGridZ ? B - b0 : kMaxGridZ;
        dim3 grid((N + TILE - 1) / TILE, (G + TILE - 1) / TILE, nb), block(TILE, TROWS);
        hipLaunchKernelGGL(transpose_pad_kernel, grid, block, 0, st, dy + (int64_t)b0 * G * Nin, X + (int64_t)b0 * N * G, G, Nin, N,
                           (int64_t)G * Nin, (int64_t)N * G, y ? y + (int64_t)b0 * G * Nin : nullptr);
        GF_LAUNCH_CHECK("transpose_pad_kernel(masked)");
    }
    return GF_OK;
}
