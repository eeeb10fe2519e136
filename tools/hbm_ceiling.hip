// hbm_ceiling.hip -- what the memory system sustains for the access mixes of the tap-stack kernels (context for the rooflines):
// pure read, pure write (plain / non-temporal stores), copy, and the chain kernel's mix (1 read : K-1 writes), 16 B per lane,
// every CU streaming.   hipcc -O3 --offload-arch=gfx950 tools/hbm_ceiling.hip -o tools/hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>  // 0 read, 1 write plain, 2 write nt, 3 copy (nt stores), 4 one read : four nt writes
__global__ __launch_bounds__(1024) void stream_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4, size_t tap4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        if (MODE == 0) acc += src[i];
        if (MODE == 1) dst[i] = (f32x4){1.f, 2.f, 3.f, (float)i};
        if (MODE == 2) __builtin_nontemporal_store((f32x4){1.f, 2.f, 3.f, (float)i}, dst + i);
        if (MODE == 3) __builtin_nontemporal_store(src[i], dst + i);
        if (MODE == 4) {
            const f32x4 v = src[i];
#pragma unroll
            for (int t = 0; t < 4; ++t) __builtin_nontemporal_store(v + (float)t, dst + t * tap4 + i);
        }
    }
    if (MODE == 0 && acc.x == 1.2345e30f) dst[0] = acc;
}

template <int MODE>
void run(const char* name, const f32x4* src, f32x4* dst, size_t n4, size_t tap4, double bytes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256 * 4, 256 * 8, 256 * 32}) {
        stream_kernel<MODE><<<grid, 256>>>(src, dst, n4, tap4);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 5; ++it) stream_kernel<MODE><<<grid, 256>>>(src, dst, n4, tap4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-34s grid %5d  %8.1f us  %7.0f GB/s\n", name, grid, ms * 1e3, bytes / ms / 1e6);
    }
}

// per-CU store / load rate when the chip is NOT saturated: `blocks` workgroups (one per CU) of `threads` threads stream
template <int MODE>
void run_percu(const char* name, const f32x4* src, f32x4* dst, size_t n4) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {32, 128, 256})
        for (int threads : {64, 128, 256, 1024}) {
            const size_t m4 = n4 / 256 * blocks;   // the same bytes per workgroup in every configuration
            hipLaunchKernelGGL(stream_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, src, dst, m4, n4);
            CK(hipEventRecord(e0));
            for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, src, dst, m4, n4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            printf("%-12s %3d CUs x %4d threads  %8.1f us  %6.1f GB/s per CU  %6.0f GB/s total\n", name, blocks, threads, ms * 1e3,
                   (double)m4 * 16 / ms / 1e6 / blocks, (double)m4 * 16 / ms / 1e6);
        }
}

int main() {
    const size_t tap = (size_t)256 * 32 * 10000 * 4;   // bytes of one config-2 tap (328 MB)
    f32x4 *src, *dst;
    CK(hipMalloc(&src, tap)); CK(hipMalloc(&dst, 4 * tap));
    CK(hipMemset(src, 0, tap)); CK(hipMemset(dst, 0, 4 * tap));
    const size_t n4 = tap / 16;
    run<0>("read 328 MB", src, dst, n4, n4, (double)tap);
    run<1>("write 328 MB (plain)", src, dst, n4, n4, (double)tap);
    run<2>("write 328 MB (nt)", src, dst, n4, n4, (double)tap);
    run<1>("write 1.31 GB (plain)", src, dst, 4 * n4, n4, 4.0 * tap);
    run<2>("write 1.31 GB (nt)", src, dst, 4 * n4, n4, 4.0 * tap);
    run<3>("copy 328 MB (nt)", src, dst, n4, n4, 2.0 * tap);
    run<4>("1 read : 4 nt writes (1.64 GB)", src, dst, n4, n4, 5.0 * tap);
    run_percu<0>("read", src, dst, n4);
    run_percu<2>("write nt", src, dst, n4);
    return 0;
}
