// mfma4x4_probe.hip -- operand layout and numerics of v_mfma_f32_4x4x1_16b_f32 on the device (what gf_msweep.hip relies on):
//   D_b[i][j] += A_b[i] * B_b[j], block b = lane / 4, A: i = lane % 4 (src0), B: j = lane % 4 (src1), D: register i, lane 4 b + j;
//   a zero A leaves the accumulator bit for bit; a product is one fmaf.
//   hipcc -O2 --offload-arch=gfx950 tools/mfma4x4_probe.hip -o /tmp/mfma4x4_probe && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, const float* c, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {c[l * 4], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}
int main() {
    float ha[64], hb[64], hc[256], hd[256];
    for (int l = 0; l < 64; ++l) {
        ha[l] = (l % 5 == 0) ? 0.f : 1.0f + 0.37f * l;      // some zero A lanes
        hb[l] = 3.0f + 1.1f * l + 1e-3f * l * l;
        for (int i = 0; i < 4; ++i) hc[l * 4 + i] = 0.123f * (l * 4 + i) - 7.f;
    }
    float *a, *b, *c, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 1024); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, c, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad_std = 0, bad_swapped = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int blk = l / 4, j = l % 4;
            const float e_std = fmaf(ha[blk * 4 + i], hb[blk * 4 + j], hc[l * 4 + i]);      // register = A's index, lane = B's index
            const float e_swp = fmaf(ha[blk * 4 + j], hb[blk * 4 + i], hc[l * 4 + i]);      // the transposed reading
            bad_std += memcmp(&e_std, &hd[l * 4 + i], 4) != 0;
            bad_swapped += memcmp(&e_swp, &hd[l * 4 + i], 4) != 0;
        }
    printf("mfma_f32_4x4x1_16b: layout D[reg i][lane j] = A[i] * B[j] + C: %s (%d mismatches); transposed reading: %d mismatches\n",
           bad_std == 0 ? "CONFIRMED bitwise (fmaf)" : "NOT confirmed", bad_std, bad_swapped);
    return bad_std != 0;
}
