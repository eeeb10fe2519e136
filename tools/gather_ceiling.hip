// gather_ceiling.hip -- how fast can gfx950 serve random full-line (128 B) gathers out of an L2-resident panel?
// Standalone microbenchmark (hipcc --offload-arch=gfx950 tools/gather_ceiling.hip -o tools/gather_ceiling).
// Each 8-lane group loads one 128-byte row (float4 per lane) at a pseudo-random row of a panel; NLOAD loads are in
// flight per lane; block b uses panel b % 8 (one panel per XCD, as the SpMM tile mapping does).  No stores except one
// sink per thread, no LDS, no index traffic: an upper bound for the SpMM hop's gather stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int NLOAD>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ X, int rows_mask, long panel_floats, int iters,
                                                      float* __restrict__ sink) {
    const int li = threadIdx.x & 7;
    const unsigned gid = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const float* P = X + (long)(blockIdx.x & 7) * panel_floats + li * 4;
    unsigned h = gid * 2654435761u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[NLOAD];
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) {
            h = h * 1664525u + 1013904223u;
            const unsigned row = (h >> 8) & rows_mask;
            const f32x4* p = reinterpret_cast<const f32x4*>(P + (long)row * 32);
            if (MODE == 1) v[u] = __builtin_nontemporal_load(p);
            else v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) acc += v[u];
    }
    if (acc.x == 123.456f) sink[0] = acc.y + acc.z + acc.w;
}

// narrow rows: LPR lanes per row, 16 bytes per lane (LPR = 4: the 64-byte rows of the EVGF tap at B = 16; 2: 32-byte rows at B = 8)
template <int LPR, int NLOAD>
__global__ __launch_bounds__(256) void gather_narrow_kernel(const float* __restrict__ X, int rows_mask, long panel_floats, int iters,
                                                             float* __restrict__ sink) {
    const int li = threadIdx.x % LPR;
    const unsigned gid = (blockIdx.x * 256 + threadIdx.x) / LPR;
    const float* P = X + (long)(blockIdx.x & 7) * panel_floats + li * 4;
    unsigned h = gid * 2654435761u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[NLOAD];
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) {
            h = h * 1664525u + 1013904223u;
            const unsigned row = (h >> 8) & rows_mask;
            v[u] = *reinterpret_cast<const f32x4*>(P + (long)row * (LPR * 4));
        }
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) acc += v[u];
    }
    if (acc.x == 123.456f) sink[0] = acc.y + acc.z + acc.w;
}

template <int LPR, int NLOAD>
void run_narrow(const float* X, int rows, int blocks, float* sink) {
    const int iters = 64;
    const long pf = (long)rows * LPR * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    gather_narrow_kernel<LPR, NLOAD><<<blocks, 256>>>(X, rows - 1, pf, iters, sink);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) gather_narrow_kernel<LPR, NLOAD><<<blocks, 256>>>(X, rows - 1, pf, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double reqs = (double)blocks * (256 / LPR) * iters * NLOAD;
    printf("  rows of %3d B, %2d in flight   panel %7.2f MB per XCD  %8.1f us  %8.1f GB/s gathered  %6.1f G rows/s\n", LPR * 16, NLOAD,
           rows * LPR * 16.0 / 1e6, ms * 1e3, reqs * LPR * 16 / ms / 1e6, reqs / ms / 1e6);
}

// short-lived variant: what the hop kernel's waves look like -- ITERS rounds of NLOAD gathers, then ONE 128-byte row store
template <int NLOAD, int ITERS, int DEP>
__global__ __launch_bounds__(256) void gather_short_kernel(const float* __restrict__ X, int rows_mask, long panel_floats,
                                                            const int* __restrict__ idx, float* __restrict__ out) {
    const int li = threadIdx.x & 7;
    const unsigned gid = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const float* P = X + (long)(blockIdx.x & 7) * panel_floats + li * 4;
    unsigned h = gid * 2654435761u + 12345u;
    if (DEP) h += idx[gid & 1023];   // a dependent index load in front of the first gather (like kptr -> entries -> gather)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        f32x4 v[NLOAD];
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) {
            h = h * 1664525u + 1013904223u;
            const unsigned row = (h >> 8) & rows_mask;
            v[u] = *reinterpret_cast<const f32x4*>(P + (long)row * 32);
        }
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) acc += v[u];
    }
    *reinterpret_cast<f32x4*>(out + ((long)gid * 32 + li * 4)) = acc;
}

template <int NLOAD, int ITERS, int DEP>
void run_short(const float* X, int rows, long panel_floats, long groups, const int* idx, float* out) {
    const int blocks = (int)(groups / 32);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    gather_short_kernel<NLOAD, ITERS, DEP><<<blocks, 256>>>(X, rows - 1, panel_floats, idx, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) gather_short_kernel<NLOAD, ITERS, DEP><<<blocks, 256>>>(X, rows - 1, panel_floats, idx, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = (double)groups * ITERS * NLOAD * 128.0;
    printf("  short-lived: %2d x %2d gathers + 1 row store per lane group, dep=%d, %7ld groups (%6d blocks)  %8.1f us  %8.1f GB/s gathered\n",
           ITERS, NLOAD, DEP, groups, blocks, ms * 1e3, bytes / ms / 1e6);
}

template <int MODE, int NLOAD>
void run(const char* label, const float* X, int rows, long panel_floats, int blocks, float* sink) {
    const int iters = 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    gather_kernel<MODE, NLOAD><<<blocks, 256>>>(X, rows - 1, panel_floats, iters, sink);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) gather_kernel<MODE, NLOAD><<<blocks, 256>>>(X, rows - 1, panel_floats, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = (double)blocks * 32 * iters * NLOAD * 128.0;
    printf("  %-28s panel %7.2f MB  blocks %6d  %8.1f us  %8.1f GB/s gathered\n", label, rows * 128.0 / 1e6, blocks, ms * 1e3, bytes / ms / 1e6);
}

int main(int argc, char** argv) {
    if (argc > 2 && !strcmp(argv[1], "one")) {   // `gather_ceiling one <rows per panel>`: one shape only (counter collection)
        const int rows = atoi(argv[2]);
        const long pf = (long)rows * 32;
        float *X, *sink;
        CK(hipMalloc(&X, pf * 8 * sizeof(float))); CK(hipMalloc(&sink, 64));
        CK(hipMemset(X, 0, pf * 8 * sizeof(float)));
        const int blocks = argc > 3 ? atoi(argv[3]) : 4096;   // 1024 blocks = 4 per CU = 16 waves per CU (what an LDS-full kernel can hold)
        run<0, 4>("plain  4 in flight", X, rows, pf, blocks, sink);
        run<0, 8>("plain  8 in flight", X, rows, pf, blocks, sink);
        run<0, 16>("plain 16 in flight", X, rows, pf, blocks, sink);
        return 0;
    }
    if (argc > 1) {   // `gather_ceiling narrow`: row-width sweep for the EVGF tap analysis (profiles/r02_c_evgf)
        float *X, *sink;
        CK(hipMalloc(&X, (size_t)8 * 262144 * 128)); CK(hipMalloc(&sink, 64));
        CK(hipMemset(X, 0, (size_t)8 * 262144 * 128));
        for (int rows : {16384, 32768, 65536, 262144}) {
            run_narrow<8, 8>(X, rows, 4096, sink);
            run_narrow<4, 8>(X, rows, 4096, sink);
            run_narrow<2, 8>(X, rows, 4096, sink);
        }
        return 0;
    }
    {   // the cfg2 hop has 2.56M (row, batch entry) pairs of ~10 gathers each: same amount of work in different wave shapes
        const int rows = 8192; const long pf = (long)rows * 32; const long groups = 2560000 / 32 * 32;
        float *X, *out; int* idx;
        CK(hipMalloc(&X, pf * 8 * sizeof(float))); CK(hipMalloc(&out, groups * 128)); CK(hipMalloc(&idx, 4096));
        CK(hipMemset(X, 0, pf * 8 * sizeof(float))); CK(hipMemset(idx, 0, 4096));
        run_short<12, 1, 0>(X, rows, pf, groups, idx, out);
        run_short<12, 1, 1>(X, rows, pf, groups, idx, out);
        run_short<6, 2, 0>(X, rows, pf, groups, idx, out);
        run_short<4, 3, 0>(X, rows, pf, groups, idx, out);
        run_short<4, 3, 1>(X, rows, pf, groups, idx, out);
        run_short<12, 2, 0>(X, rows, pf, groups / 2, idx, out);
        run_short<12, 4, 0>(X, rows, pf, groups / 4, idx, out);
        run_short<12, 8, 0>(X, rows, pf, groups / 8, idx, out);
        run_short<12, 16, 0>(X, rows, pf, groups / 16, idx, out);
        CK(hipFree(X)); CK(hipFree(out)); CK(hipFree(idx));
    }
    const int sizes[] = {8192, 16384, 32768, 131072};   // rows per panel: 1, 2, 4, 16 MB
    for (int rows : sizes) {
        const long pf = (long)rows * 32;
        float* X; float* sink;
        CK(hipMalloc(&X, pf * 8 * sizeof(float))); CK(hipMalloc(&sink, 64));
        CK(hipMemset(X, 0, pf * 8 * sizeof(float)));
        for (int blocks : {4096}) {
            run<0, 4>("plain  4 in flight", X, rows, pf, blocks, sink);
            run<0, 8>("plain  8 in flight", X, rows, pf, blocks, sink);
            run<0, 16>("plain 16 in flight", X, rows, pf, blocks, sink);
            run<1, 8>("nt     8 in flight", X, rows, pf, blocks, sink);
            run<1, 16>("nt    16 in flight", X, rows, pf, blocks, sink);
        }
        CK(hipFree(X)); CK(hipFree(sink));
    }
    return 0;
}
