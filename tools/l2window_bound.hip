// l2window_bound.hip -- can the node-major hop at N = 1e5 (config 4: ER, nnz ~ 1e6, 128 batch entries x 32 columns) be made faster
// by turning its Infinity-Cache gathers into L2 hits?   Standalone prototype + bound:
//     hipcc -O3 --offload-arch=gfx950 tools/l2window_bound.hip -o tools/l2window_bound && tools/l2window_bound
//
// Design under test ("source-sorted streams"): an XCD works on ONE batch entry at a time (12.8 MB of rows, 3x its L2).  The
// destination rows are cut into blocks of D rows, one block per workgroup, accumulators in LDS (D * 128 B).  Inside a workgroup
// each 8-lane group owns D/64 rows and walks ALL their edges sorted by SOURCE index; every lane group of every CU of the XCD
// therefore sweeps the source rows 0 .. N together, and the rows being gathered at any moment form a sliding window that fits
// the XCD's 4 MiB L2: a row is fetched from the fabric once per pass and hit ~(32 CUs * D / N * degree) times in L2.
// The accumulate is an LDS read-modify-write per edge (each row is owned by exactly one lane group: no atomics, fixed order).
// Variants: sorted (the design), unsorted (same code, edges in random order: what L2 does without the window), gather-only.
// Reported: time for all 128 batch entries = one hop; compare with spmm_sell_kernel's 1.90 ms (profiles/r01_p_fused/bench_cfg4.json).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr unsigned kPad = 0xffffffffu;
#ifndef KW
#define KW 8
#endif
constexpr int kWaves = KW;   // waves per workgroup

// stream[(wg * kWaves + wave) * L4 + s][lg] : u32x4 = 4 consecutive entries of lane group lg; entry = dest_local << 17 | src
template <int ACCUM>
__global__ __launch_bounds__(kWaves * 64) void window_kernel(const u32x4* __restrict__ stream, const float* __restrict__ X,
                                                             float* __restrict__ Y, int N, int D, int L4, int blocksPerEntry,
                                                             int passBlocks, float uval) {
    extern __shared__ __attribute__((aligned(16))) float4 accs[];  // [D] rows of 32 floats = 8 float4
    f32x4* acc4 = reinterpret_cast<f32x4*>(accs);
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 7, lg = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // hardware dispatch order: workgroup i runs on XCD i % 8; an XCD walks its batch entries one at a time, `passBlocks`
    // destination blocks (one per CU) per pass
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int passesPerEntry = (blocksPerEntry + passBlocks - 1) / passBlocks;
    const int pass = slot / passBlocks, cu = slot - pass * passBlocks;
    const int b = (pass / passesPerEntry) * 8 + xcd;
    const int blk = (pass % passesPerEntry) * passBlocks + cu;
    if (blk >= blocksPerEntry) return;
    const float* Xb = X + (int64_t)b * N * 32 + sub * 4;
    for (int i = tid; i < D * 8; i += kWaves * 64) acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const u32x4* st = stream + ((int64_t)(blk * kWaves + wave) * L4) * 8 + lg;
    u32x4 e = st[0];
    for (int s = 0; s < L4; ++s) {
        const u32x4 en = st[(int64_t)(s + 1 < L4 ? s + 1 : s) * 8];   // next 4 entries of this lane group (broadcast load)
        f32x4 x[4];
        const unsigned ee[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ee[u] != kPad) x[u] = *reinterpret_cast<const f32x4*>(Xb + (int64_t)(ee[u] & 0x1ffffu) * 32);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ee[u] != kPad) {
                if (ACCUM) {
                    const int d = (int)(ee[u] >> 17) * 8 + sub;
                    acc4[d] += x[u];   // the row belongs to this lane group alone: plain read-modify-write, program order
                } else if (x[u].x == 1.2345e30f) {
                    acc4[sub] = x[u];
                }
            }
        e = en;
    }
    __syncthreads();
    float* Yb = Y + (int64_t)b * N * 32 + (int64_t)blk * D * 32;
    const int rows = min(D, N - blk * D);
    for (int i = tid; i < rows * 8; i += kWaves * 64) {
        f32x4 v = acc4[i];
        v *= uval;
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Yb) + i);
    }
}

// Persistent variant with an XCD-local barrier at the start of every pass (and at nSync - 1 evenly spaced points inside it): the
// workgroups of an XCD (blockIdx % 8) walk the passes of that XCD together, so that their sweeps over the source rows stay aligned
// and the rows in flight really are one sliding window.  grid = 8 * passBlocks workgroups, all resident.
template <int ACCUM>
__global__ __launch_bounds__(kWaves * 64) void window_sync_kernel(const u32x4* __restrict__ stream, const float* __restrict__ X,
                                                                  float* __restrict__ Y, int N, int D, int L4, int blocksPerEntry,
                                                                  int passBlocks, float uval, int entriesPerXcd, int nSync,
                                                                  unsigned* __restrict__ counters, unsigned epoch0) {
    extern __shared__ __attribute__((aligned(16))) float4 accs[];
    f32x4* acc4 = reinterpret_cast<f32x4*>(accs);
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 7, lg = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;      // cu: 0 .. passBlocks-1
    const int passesPerEntry = (blocksPerEntry + passBlocks - 1) / passBlocks;
    unsigned* ctr = counters + xcd * 32;                       // one counter per XCD, 128 bytes apart
    unsigned arrived = epoch0;
    auto xcd_barrier = [&]() {
        __syncthreads();
        arrived += (unsigned)passBlocks;
        if (tid == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int guard = 0;
            while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - arrived) < 0 && ++guard < (1 << 22))
                __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    };
    for (int pass = 0; pass < entriesPerXcd * passesPerEntry; ++pass) {
        const int b = (pass / passesPerEntry) * 8 + xcd;
        const int blk = (pass % passesPerEntry) * passBlocks + cu;
        const bool live = blk < blocksPerEntry;
        const float* Xb = X + (int64_t)b * N * 32 + sub * 4;
        for (int i = tid; i < D * 8; i += kWaves * 64) acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const u32x4* st = stream + ((int64_t)((live ? blk : 0) * kWaves + wave) * L4) * 8 + lg;
        for (int seg = 0; seg < nSync; ++seg) {
            xcd_barrier();
            if (!live) continue;
            const int s0 = (int)((int64_t)L4 * seg / nSync), s1 = (int)((int64_t)L4 * (seg + 1) / nSync);
            u32x4 e = st[(int64_t)s0 * 8];
            for (int s = s0; s < s1; ++s) {
                const u32x4 en = st[(int64_t)(s + 1 < L4 ? s + 1 : s) * 8];
                f32x4 x[4];
                const unsigned ee[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ee[u] != kPad) x[u] = *reinterpret_cast<const f32x4*>(Xb + (int64_t)(ee[u] & 0x1ffffu) * 32);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ee[u] != kPad) {
                        if (ACCUM) {
                            const int d = (int)(ee[u] >> 17) * 8 + sub;
                            acc4[d] += x[u];
                        } else if (x[u].x == 1.2345e30f) {
                            acc4[sub] = x[u];
                        }
                    }
                e = en;
            }
        }
        __syncthreads();
        if (live) {
            float* Yb = Y + (int64_t)b * N * 32 + (int64_t)blk * D * 32;
            const int rows = min(D, N - blk * D);
            for (int i = tid; i < rows * 8; i += kWaves * 64) {
                f32x4 v = acc4[i];
                v *= uval;
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Yb) + i);
            }
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int N = 100000, B = argc > 1 ? atoi(argv[1]) : 128;
    const int64_t nnzTarget = 1000000;
    std::mt19937 rng(0);
    std::vector<std::vector<int>> nbr(N);   // nbr[dest] = sources
    {
        std::uniform_int_distribution<int> pick(0, N - 1);
        for (int64_t k = 0; k < nnzTarget / 2; ++k) {
            const int i = pick(rng), j = pick(rng);
            if (i == j) continue;
            nbr[i].push_back(j);
            nbr[j].push_back(i);
        }
        for (auto& v : nbr) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
    }
    int64_t nnz = 0;
    for (auto& v : nbr) nnz += (int64_t)v.size();
    const float uval = 0.1f;
    printf("ER graph N=%d nnz=%lld, B=%d batch entries x 32 columns (row = 128 B); algorithmic bytes/hop = %.3f GB\n", N, (long long)nnz, B,
           (2.0 * B * N * 32 * 4 + nnz * 8.0 + (N + 1) * 4.0) / 1e9);

    float *X, *Y;
    CK(hipMalloc(&X, (size_t)B * N * 32 * 4));
    CK(hipMalloc(&Y, (size_t)B * N * 32 * 4));
    {
        std::vector<float> h((size_t)N * 32);
        for (auto& v : h) v = (float)((rng() & 0xffff) / 65536.0 - 0.5);
        for (int b = 0; b < B; ++b) CK(hipMemcpy(X + (size_t)b * N * 32, h.data(), h.size() * 4, hipMemcpyHostToDevice));  // same signal in every entry
    }
    std::vector<float> hx((size_t)N * 32);
    CK(hipMemcpy(hx.data(), X, hx.size() * 4, hipMemcpyDeviceToHost));

    for (int D : {520, 1040}) {
        const int blocksPerEntry = (N + D - 1) / D;
        int wgPerCU = (160 * 1024) / (D * 128);
        if (wgPerCU * kWaves > 32) wgPerCU = 32 / kWaves;
        const int passBlocks = 32 * wgPerCU;            // destination blocks an XCD works on concurrently
        for (int sorted = 1; sorted >= (getenv("L2W_SHUFFLED") ? 0 : 1); --sorted) {
            // build the streams: lane group (wave, lg) of block blk owns rows blk*D + (wave*8 + lg) + kWaves*8*t
            int L = 0;
            std::vector<std::vector<unsigned>> lists((size_t)blocksPerEntry * kWaves * 8);
            for (int blk = 0; blk < blocksPerEntry; ++blk)
                for (int q = 0; q < kWaves * 8; ++q) {
                    auto& li = lists[(size_t)blk * kWaves * 8 + q];
                    for (int t = 0; q + kWaves * 8 * t < D; ++t) {
                        const int dl = q + kWaves * 8 * t, row = blk * D + dl;
                        if (row >= N) break;
                        for (int src : nbr[row]) li.push_back(((unsigned)dl << 17) | (unsigned)src);
                    }
                    if (sorted)
                        std::sort(li.begin(), li.end(), [](unsigned a, unsigned c) { return (a & 0x1ffffu) < (c & 0x1ffffu); });
                    else
                        std::shuffle(li.begin(), li.end(), rng);
                    L = std::max<int>(L, (int)li.size());
                }
            const int L4 = (L + 3) / 4;
            std::vector<unsigned> st((size_t)blocksPerEntry * kWaves * L4 * 8 * 4, kPad);
            int64_t slots = 0;
            for (int blk = 0; blk < blocksPerEntry; ++blk)
                for (int w = 0; w < kWaves; ++w)
                    for (int lg = 0; lg < 8; ++lg) {
                        const auto& li = lists[(size_t)blk * kWaves * 8 + w * 8 + lg];
                        for (size_t k = 0; k < li.size(); ++k)
                            st[((((size_t)(blk * kWaves + w) * L4 + k / 4) * 8 + lg) * 4) + (k & 3)] = li[k];
                        slots += L4 * 4;
                    }
            unsigned* dst;
            CK(hipMalloc(&dst, st.size() * 4));
            CK(hipMemcpy(dst, st.data(), st.size() * 4, hipMemcpyHostToDevice));
            const int passesPerEntry = (blocksPerEntry + passBlocks - 1) / passBlocks;
            const int entriesPerXcd = (B + 7) / 8;
            const int grid = 8 * entriesPerXcd * passesPerEntry * passBlocks;
            const size_t lds = (size_t)D * 128;
            for (int accum = 1; accum >= 0; --accum) {
                auto kern = accum ? window_kernel<1> : window_kernel<0>;
                CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0));
                CK(hipEventCreate(&e1));
                CK(hipMemset(Y, 0xff, (size_t)B * N * 32 * 4));
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), lds, 0, (const u32x4*)dst, X, Y, N, D, L4, blocksPerEntry, passBlocks, uval);
                CK(hipDeviceSynchronize());
                const int iters = 5;
                CK(hipEventRecord(e0));
                for (int it = 0; it < iters; ++it)
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), lds, 0, (const u32x4*)dst, X, Y, N, D, L4, blocksPerEntry, passBlocks, uval);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= iters;
                double err = -1.0;
                if (accum && B % 8 == 0) {   // check the last batch entry against the host
                    std::vector<float> hy((size_t)N * 32);
                    CK(hipMemcpy(hy.data(), Y + (size_t)(B - 1) * N * 32, hy.size() * 4, hipMemcpyDeviceToHost));
                    err = 0.0;
                    for (int row = 0; row < N; row += 97)
                        for (int c = 0; c < 32; ++c) {
                            double s = 0.0;
                            for (int src : nbr[row]) s += hx[(size_t)src * 32 + c];
                            err = std::max(err, std::fabs(s * uval - hy[(size_t)row * 32 + c]));
                        }
                }
                const double bytes = 2.0 * B * N * 32 * 4 + nnz * 8.0 + (N + 1) * 4.0;
                printf("D=%4d (%d WG/CU, %d passes/entry) %-8s %-11s L=%3d fill=%.2f  %8.3f ms/hop  %6.0f GB/s algorithmic = %4.1f %% of 8 TB/s  gathered %5.1f TB/s  maxerr=%.2e\n",
                       D, wgPerCU, passesPerEntry, sorted ? "sorted" : "shuffled", accum ? "gather+acc" : "gather-only", L, (double)nnz / slots, ms,
                       bytes / ms / 1e6, bytes / ms / 1e6 / 80.0, (double)nnz * B * 128.0 / ms / 1e9, err);
            }
            if (sorted && B % 8 == 0) {
                unsigned* ctrs;
                CK(hipMalloc(&ctrs, 8 * 32 * 4));
                CK(hipMemset(ctrs, 0, 8 * 32 * 4));
                unsigned epoch = 0;
                const int nSegTotalPerLaunch = entriesPerXcd * passesPerEntry;
                for (int nSync : {1, 2, 4}) {
                    auto kern = window_sync_kernel<1>;
                    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0));
                    CK(hipEventCreate(&e1));
                    CK(hipMemset(Y, 0xff, (size_t)B * N * 32 * 4));
                    auto launch = [&]() {
                        hipLaunchKernelGGL(kern, dim3(8 * passBlocks), dim3(kWaves * 64), lds, 0, (const u32x4*)dst, X, Y, N, D, L4, blocksPerEntry,
                                           passBlocks, uval, entriesPerXcd, nSync, ctrs, epoch);
                        epoch += (unsigned)passBlocks * (unsigned)nSegTotalPerLaunch * (unsigned)nSync;
                    };
                    launch();
                    CK(hipDeviceSynchronize());
                    const int iters = 5;
                    CK(hipEventRecord(e0));
                    for (int it = 0; it < iters; ++it) launch();
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    ms /= iters;
                    std::vector<float> hy((size_t)N * 32);
                    CK(hipMemcpy(hy.data(), Y + (size_t)(B - 1) * N * 32, hy.size() * 4, hipMemcpyDeviceToHost));
                    double err = 0.0;
                    for (int row = 0; row < N; row += 97)
                        for (int c = 0; c < 32; ++c) {
                            double sacc = 0.0;
                            for (int src : nbr[row]) sacc += hx[(size_t)src * 32 + c];
                            err = std::max(err, std::fabs(sacc * uval - hy[(size_t)row * 32 + c]));
                        }
                    const double bytes = 2.0 * B * N * 32 * 4 + nnz * 8.0 + (N + 1) * 4.0;
                    printf("D=%4d persistent, XCD barrier x%d per pass (%d passes per XCD)  %8.3f ms/hop  %6.0f GB/s algorithmic = %4.1f %% of 8 TB/s  maxerr=%.2e\n",
                           D, nSync, nSegTotalPerLaunch, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0, err);
                }
                CK(hipFree(ctrs));
            }
            CK(hipFree(dst));
        }
    }
    return 0;
}
