// residency_probe.hip -- how many workgroups of a given LDS size / thread count does a CU of this chip really hold at once?
// Launches 256 * k workgroups that each idle ~100 us and reports the spread of their start times (small spread = all co-resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <int T>
__global__ __launch_bounds__(T) void probe(unsigned long long* st, float* out) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        atomicMin(st, t0);
        atomicMax(st + 1, t0);
    }
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 10000) __builtin_amdgcn_s_sleep(32);   // 100 us at 100 MHz
    if (lds[(threadIdx.x + 1) % T] == -1.f) out[0] = 1.f;
}
int main() {
    unsigned long long* st;
    float* out;
    CK(hipMalloc(&st, 16));
    CK(hipMalloc(&out, 4));
    for (int threads : {512, 576, 1024})
        for (int k : {1, 2, 3})
            for (int kb : {40, 44, 46, 48, 52, 60, 64, 66, 68, 70, 72, 76, 79}) {
                const unsigned long long init[2] = {~0ull, 0ull};
                CK(hipMemcpy(st, init, 16, hipMemcpyHostToDevice));
                auto kern = threads == 512 ? probe<512> : threads == 576 ? probe<576> : probe<1024>;
                CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
                hipLaunchKernelGGL(kern, dim3(256 * k), dim3(threads), (size_t)kb * 1024, 0, st, out);
                CK(hipDeviceSynchronize());
                unsigned long long h[2];
                CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost));
                printf("threads=%4d  %d WG/CU requested  LDS %2d KB each (%3d KB per CU): start spread %8.1f us %s\n", threads, k, kb, k * kb,
                       (double)(h[1] - h[0]) / 100.0, (h[1] - h[0]) < 5000 ? "co-resident" : "NOT co-resident");
            }
    return 0;
}
