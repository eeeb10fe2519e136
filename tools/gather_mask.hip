// gather_mask.hip -- what does a PARTIALLY active gather instruction cost?  (round 4: sizing the padded steps of the stream kernel and
// the masked lane groups of a register-accumulator sweep.)  Same set-up as gather_ceiling.hip (one L2-resident panel per XCD, random
// 128-byte rows, 8 lanes per row, 8 loads in flight per lane) with
//   mode 0: only K of the 8 lane groups of every wave active (exec-masked), K = 8, 6, 4, 2, 1
//   mode 1: all 64 lanes active, but only K groups gather random rows -- the others all read row 0 (one shared line)
//   mode 2: all lanes read the SAME 16 bytes (the dummy gather of a store step)
// Reported: time per gather INSTRUCTION and useful bytes/s.   hipcc --offload-arch=gfx950 -O3 tools/gather_mask.hip -o tools/gather_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int K>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ X, int rows_mask, long panel_floats, int iters, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, li = lane & 7, g = lane >> 3;
    const unsigned gid = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const float* P = X + (long)(blockIdx.x & 7) * panel_floats + li * 4;
    unsigned h = gid * 2654435761u + 12345u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool active = g < K;
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;
            unsigned row = (h >> 8) & rows_mask;
            if (MODE == 0) {
                if (active) v[u] = *reinterpret_cast<const f32x4*>(P + (long)row * 32);
                else v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else if (MODE == 1) {
                if (!active) row = 0;
                v[u] = *reinterpret_cast<const f32x4*>(P + (long)row * 32);
            } else {
                v[u] = *reinterpret_cast<const f32x4*>(P - li * 4 + (long)((h >> 8) & 7) * 0);   // every lane: the same 16 bytes
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc.x == 123.456f) sink[0] = acc.y + acc.z + acc.w;
}

template <int MODE, int K>
void run(const float* X, int rows, long pf, float* sink) {
    const int iters = 64, blocks = 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_gather<MODE, K><<<blocks, 256>>>(X, rows - 1, pf, iters, sink);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k_gather<MODE, K><<<blocks, 256>>>(X, rows - 1, pf, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double instr = (double)blocks * 4 * iters * 8;                    // wave-level gather instructions
    const double useful = instr * (MODE == 2 ? 0 : K) * 128.0;
    // 256 CUs x 4 SIMDs: instructions per CU per microsecond -> clocks per instruction per CU at 2.4 GHz
    printf("  mode %d K=%d  %8.1f us  %7.2f ns/instr/CU  %8.1f GB/s useful\n", MODE, K, ms * 1e3, ms * 1e6 / (instr / 256), useful / ms / 1e6);
}

int main() {
    const int rows = 16384;   // 2 MB per XCD: L2-resident
    const long pf = (long)rows * 32;
    float *X, *sink;
    CK(hipMalloc(&X, pf * 8 * sizeof(float))); CK(hipMalloc(&sink, 64));
    CK(hipMemset(X, 0, pf * 8 * sizeof(float)));
    run<0, 8>(X, rows, pf, sink); run<0, 6>(X, rows, pf, sink); run<0, 4>(X, rows, pf, sink); run<0, 2>(X, rows, pf, sink); run<0, 1>(X, rows, pf, sink);
    run<1, 6>(X, rows, pf, sink); run<1, 4>(X, rows, pf, sink); run<1, 2>(X, rows, pf, sink); run<1, 1>(X, rows, pf, sink);
    run<2, 8>(X, rows, pf, sink);
    return 0;
}
