// gather_width.hip -- cost of ONE gather instruction by width: a 128-byte row fetched by 32 lanes x dword (2 rows per instruction),
// 16 lanes x dwordx2 (4 rows) or 8 lanes x dwordx4 (8 rows); random rows of an L2-resident panel per XCD, 8 loads in flight per lane.
// (round 4: the register-accumulator sweep first used dword gathers and ran at the per-instruction rate, not the byte rate.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int W> struct vec;
template <> struct vec<1> { typedef float t; };
template <> struct vec<2> { typedef float t __attribute__((ext_vector_type(2))); };
template <> struct vec<4> { typedef float t __attribute__((ext_vector_type(4))); };

template <int W>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, int rows_mask, long panel_floats, int iters, float* __restrict__ sink) {
    constexpr int LPR = 32 / W;
    typedef typename vec<W>::t T;
    const int lane = threadIdx.x & 63, li = lane % LPR;
    const unsigned gid = (blockIdx.x * 256 + threadIdx.x) / LPR;
    const float* P = X + (long)(blockIdx.x & 7) * panel_floats + li * W;
    unsigned h = gid * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            v[u] = *reinterpret_cast<const T*>(P + (long)((h >> 8) & rows_mask) * 32);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (W == 1) acc += v[u];
            else if constexpr (W == 2) acc += v[u].x + v[u].y;
            else acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}
template <int W>
void run(const float* X, int rows, long pf, float* sink) {
    const int iters = 64, blocks = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<W><<<blocks, 256>>>(X, rows - 1, pf, iters, sink);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<W><<<blocks, 256>>>(X, rows - 1, pf, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double instr = (double)blocks * 4 * iters * 8;
    const double rowsn = instr * 2 * W;
    printf("  %d dword(s) per lane, %2d rows per instruction: %8.1f us  %6.2f ns/instr/CU  %6.2f ns/row/CU  %8.1f GB/s gathered\n", W, 2 * W, ms * 1e3,
           ms * 1e6 / (instr / 256), ms * 1e6 / (rowsn / 256), rowsn * 128 / ms / 1e6);
}
int main() {
    const int rows = 16384; const long pf = (long)rows * 32;
    float *X, *sink; CK(hipMalloc(&X, pf * 8 * sizeof(float))); CK(hipMalloc(&sink, 64)); CK(hipMemset(X, 0, pf * 8 * sizeof(float)));
    run<1>(X, rows, pf, sink); run<2>(X, rows, pf, sink); run<4>(X, rows, pf, sink);
    return 0;
}
