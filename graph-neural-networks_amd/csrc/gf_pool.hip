// gf_pool.hip -- MaxPoolLocal: v[b,f,i] = max over the alpha-hop neighbourhood of node i, for the first Nout nodes.
// Replaces reference graphML.py:1996-2021: x.unsqueeze(3).repeat([1,1,1,maxNeighborhood]) (the signal copied maxNeighborhood
// times), torch.gather along the node axis and torch.max over the copies -- here one pass: every output element walks its
// neighbour list (nbh[i, 0..M), padded with i itself, graphTools.computeNeighborhood 'matrix' output) through the row
// x[b,f,:] it belongs to (L1/L2-resident) and keeps the FIRST maximum, torch.max's tie rule, so that the gradient goes to
// the same element as in the reference (ties are common: the input is a ReLU output).
// Backward is a gather, not a scatter: input node j sums dv[b,f,i] over the outputs i whose list contains j and whose
// recorded arg-max position is j's first position in that list (reverse lists built once by the host) -- no atomics, fixed
// order, bitwise deterministic.  HBM-bound, trivially small next to the filter (Nout <= Nin, M ~ tens).
#include "gf_common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void maxpool_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ nbh,
                                                               float* __restrict__ v, int32_t* __restrict__ arg, int Nin, int Nout,
                                                               int M, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        const int64_t bf = idx / Nout;
        const int i = (int)(idx - bf * Nout);
        const float* row = x + bf * Nin;
        const int32_t* list = nbh + (int64_t)i * M;
        float best = row[list[0]];
        int bp = 0;
        for (int p = 1; p < M; ++p) {
            const float c = row[list[p]];
            // first maximum wins; a NaN wins over everything that precedes it and is never displaced (torch.max propagates NaN)
            if (c > best || (c != c && best == best)) {
                best = c;
                bp = p;
            }
        }
        v[idx] = best;
        arg[idx] = bp;
    }
}

__global__ __launch_bounds__(kThreads) void maxpool_bwd_kernel(const float* __restrict__ dv, const int32_t* __restrict__ arg,
                                                               const int32_t* __restrict__ rev_ptr, const int32_t* __restrict__ rev_i,
                                                               const int32_t* __restrict__ rev_p, float* __restrict__ dx, int Nin,
                                                               int Nout, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        const int64_t bf = idx / Nin;
        const int j = (int)(idx - bf * Nin);
        const float* dvr = dv + bf * Nout;
        const int32_t* ar = arg + bf * Nout;
        float acc = 0.f;
        for (int q = rev_ptr[j]; q < rev_ptr[j + 1]; ++q) {  // ascending i: fixed summation order
            const int i = rev_i[q];
            if (ar[i] == rev_p[q]) acc += dvr[i];
        }
        dx[idx] = acc;
    }
}

unsigned grid_for(int64_t items) {
    int64_t blocks = (items + kThreads - 1) / kThreads;
    if (blocks > 256 * 32) blocks = 256 * 32;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int gf_maxpool_forward(const float* x, const int32_t* nbh, float* v, int32_t* arg, int32_t B, int32_t F, int32_t Nin,
                                  int32_t Nout, int32_t M, void* stream) {
    GF_REQUIRE_ARG(x && nbh && v && arg, "gf_maxpool_forward: NULL tensor");
    GF_REQUIRE_SHAPE(B > 0 && F > 0 && Nin > 0 && Nout > 0 && Nout <= Nin && M > 0,
                     "gf_maxpool_forward: bad shape B=%d F=%d Nin=%d Nout=%d M=%d", B, F, Nin, Nout, M);  // graphML.py:1972-1976
    const int64_t total = (int64_t)B * F * Nout;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(kThreads), 0, gf_stream(stream), x, nbh, v, arg, Nin, Nout, M,
                       total);
    GF_LAUNCH_CHECK("maxpool_fwd_kernel");
    return GF_OK;
}

extern "C" int gf_maxpool_backward(const float* dv, const int32_t* arg, const int32_t* rev_ptr, const int32_t* rev_i,
                                   const int32_t* rev_p, float* dx, int32_t B, int32_t F, int32_t Nin, int32_t Nout, void* stream) {
    GF_REQUIRE_ARG(dv && arg && rev_ptr && rev_i && rev_p && dx, "gf_maxpool_backward: NULL tensor");
    GF_REQUIRE_SHAPE(B > 0 && F > 0 && Nin > 0 && Nout > 0 && Nout <= Nin, "gf_maxpool_backward: bad shape B=%d F=%d Nin=%d Nout=%d", B,
                     F, Nin, Nout);
    const int64_t total = (int64_t)B * F * Nin;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(kThreads), 0, gf_stream(stream), dv, arg, rev_ptr, rev_i, rev_p,
                       dx, Nin, Nout, total);
    GF_LAUNCH_CHECK("maxpool_bwd_kernel");
    return GF_OK;
}
