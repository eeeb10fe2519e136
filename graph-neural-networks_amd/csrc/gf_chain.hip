// gf_chain.hip -- the K-1 hops of the tap stack, one column panel at a time, WITHOUT the panel leaving LDS between hops.
// Same operation as gf_panel.hip / gf_spmm.hip (reference graphML.py:158-161: `x = torch.matmul(x, S)` per tap, the taps
// concatenated at :161); same column-panel layout Xp[P][N][4] (gf_panel.hip).
//
// What changes against the per-hop panel kernel:
//   * A workgroup loads a panel ONCE (N*16 bytes, coalesced), runs hop 1 .. K-1 on it inside LDS and stores every tap with
//     coalesced full-line stores.  HBM traffic per chain = (1 + (K-1)) * panel instead of 2 * (K-1) * panel: the K-2 re-reads
//     of the algorithmic count (SURVEY.md 8d counts a read and a write per hop) never happen.
//   * A hop's outputs stay in registers (4 floats per row per lane, <= kChainSets rows per lane) until every wave has finished
//     gathering from the panel; then they are written over the panel in natural row order and the panel is the next hop's
//     source.  Because outputs pass through registers, the rows a wave computes together need not be neighbours in memory:
//     the plan sorts rows by degree (gf_plan.hip, upload_chain) and lane = row lockstep wastes ~10 % of the gathers on
//     padding instead of the 39 % of natural-order octets.
//   * per hop two workgroup barriers (gathers done | panel rewritten); the tap's HBM store reads the rewritten panel and runs
//     under the next hop's gathers; the next panel is requested by LDS-DMA chunk by chunk inside the last tap's store sweep.
// Determinism: each row's sum runs in the plan's fixed neighbour order in one lane; no atomics.
#include "gf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
template <int UNIFORM> struct ChainCol { typedef u32x2 type; };   // weighted stream: 4 x 16-bit columns (+ 4 values)
template <> struct ChainCol<1> { typedef u32x4 type; };            // value-free stream: 4 LDS byte offsets

constexpr unsigned kNoRow = 0xffffffffu;

template <int UNIFORM>
__global__ __launch_bounds__(1024) void spmm_chain_kernel(const int32_t* __restrict__ gtab, const uint32_t* __restrict__ rowoff,
                                                          const void* __restrict__ cols, const float4* __restrict__ vals, float uval,
                                                          const float* __restrict__ Xin, float* __restrict__ Xout, int N, int nPanels,
                                                          int R, int nHops, int64_t tapStride, int store_mode) {
    extern __shared__ __attribute__((aligned(16))) float4 panel[];  // [N + 1]: the panel + one zero slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (int)blockDim.x;
    f32x4* lds4 = reinterpret_cast<f32x4*>(panel);
    // gathers address LDS absolutely (see gf_panel.hip): the dynamic panel is the only LDS object, so it starts at address 0
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();
    typedef typename ChainCol<UNIFORM>::type colw;
    const colw* col4 = reinterpret_cast<const colw*>(cols) + lane;
    const f32x4* val4 = reinterpret_cast<const f32x4*>(vals) + lane;

    unsigned ro[kChainSets];  // LDS byte offset of the row this lane computes in set r
#pragma unroll
    for (int r = 0; r < kChainSets; ++r) ro[r] = r < R ? rowoff[(int64_t)r * T + tid] : kNoRow;
    const int gv = gtab[wave * 16 + (lane & 15)];          // lane i (< 16) holds entry i of this wave's table
    const int gbeg = __builtin_amdgcn_readlane(gv, 0);
    if (tid == 0) lds4[N] = (f32x4){0.f, 0.f, 0.f, 0.f};   // the zero slot empty ELL slots gather from
    const int nj = (N + T - 1) / T;                        // rows per thread in the load / store phases (<= kChainSets)
    const int64_t pstride4 = (int64_t)N;                   // float4 per panel

    // Panel loads go through LDS-DMA (global_load_lds_dwordx4): a wave's 64 rows land as 1 KiB at a wave-uniform LDS base
    // (+ lane * 16) without passing through registers, so the NEXT panel can be requested while the last tap of the current one
    // is still being stored -- chunk by chunk, each wave overwriting only the rows it has just read back itself.
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma_chunk = [&](const f32x4* src, int j) {  // rows wave*64 + j*T ... + 63 of the panel at src
        const int row0 = wave * 64 + j * T;          // wave-uniform
        if (row0 + lane < N)
            __builtin_amdgcn_global_load_lds(src + row0 + lane, (lds_void*)(uintptr_t)((unsigned)row0 * 16u), 16, 0, 0);
    };

    int p = (int)blockIdx.x;
    if (p >= nPanels) return;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(Xin) + (int64_t)p * pstride4;
        for (int j = 0; j < nj; ++j) dma_chunk(src, j);
    }
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the panel has landed
        __syncthreads();                                   // ... and everybody else's
        const int pn = p + (int)gridDim.x;
        for (int h = 0; h < nHops; ++h) {
            // The wave's stream of group-rows, two in flight in two register sets (A, B) that alternate strictly per group-row --
            // also across block boundaries, so a block starts on A or on B (`par`, wave-uniform) and exists in both variants.
            // Each block is its own loop and assigns its set's accumulators exactly once, outside any loop: the accumulators of
            // the other sets are merely live across it (a single stream loop that commits into acc[r] through a switch made the
            // register allocator shuffle all 40 accumulator registers on every iteration).
            f32x4 acc[kChainSets];
            {
                int g = gbeg;
                int par = 0;
                colw cA = col4[(int64_t)g * 64], cB = col4[(int64_t)(g + 1) * 64];
                f32x4 vA, vB;
                if (!UNIFORM) {
                    vA = val4[(int64_t)g * 64];
                    vB = val4[(int64_t)(g + 1) * 64];
                }
                f32x4 a0, a1;
                auto step = [&](colw& cc, f32x4& vv) {
                    unsigned o0, o1, o2, o3;
                    if constexpr (UNIFORM) {
                        o0 = cc.x, o1 = cc.y, o2 = cc.z, o3 = cc.w;
                    } else {
                        o0 = (cc.x & 0xffffu) << 4, o1 = (cc.x >> 16) << 4, o2 = (cc.y & 0xffffu) << 4, o3 = (cc.y >> 16) << 4;
                    }
                    const f32x4 x0 = *reinterpret_cast<lds_f32x4*>(o0);
                    const f32x4 x1 = *reinterpret_cast<lds_f32x4*>(o1);
                    const f32x4 x2 = *reinterpret_cast<lds_f32x4*>(o2);
                    const f32x4 x3 = *reinterpret_cast<lds_f32x4*>(o3);
                    const f32x4 w = vv;
                    cc = col4[(int64_t)(g + 2) * 64];  // refill this register set (the stream ends with two sentinel group-rows)
                    if (!UNIFORM) vv = val4[(int64_t)(g + 2) * 64];
                    if (UNIFORM) {
                        a0 += x0;
                        a1 += x1;
                        a0 += x2;
                        a1 += x3;
                    } else {
                        a0 += w.x * x0;
                        a1 += w.y * x1;
                        a0 += w.z * x2;
                        a1 += w.w * x3;
                    }
                    ++g;
                };
                auto block = [&](int ge) -> f32x4 {  // group-rows [g, ge) of the stream, ge > g
                    a0 = a1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (par == 0) {
                        for (;;) {
                            step(cA, vA);
                            if (g == ge) { par = 1; break; }
                            step(cB, vB);
                            if (g == ge) { par = 0; break; }
                        }
                    } else {
                        for (;;) {
                            step(cB, vB);
                            if (g == ge) { par = 0; break; }
                            step(cA, vA);
                            if (g == ge) { par = 1; break; }
                        }
                    }
                    return a0 + a1;
                };
#pragma unroll
                for (int r = 0; r < kChainSets; ++r) {
                    acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (r < R) acc[r] = block(__builtin_amdgcn_readlane(gv, r + 1));
                }
            }
            __syncthreads();  // every wave has finished gathering from the panel
#pragma unroll
            for (int r = 0; r < kChainSets; ++r)
                if (ro[r] != kNoRow) {
                    f32x4 v = acc[r];
                    if (UNIFORM) v *= uval;
                    *reinterpret_cast<lds_f32x4*>(ro[r]) = v;
                }
            __syncthreads();  // the panel now holds tap h + 1
            // store tap h + 1 from the rewritten panel (full lines, natural order).  After the chain's last hop the same sweep
            // requests the next panel: each 64-row chunk is overwritten by the wave that has just read it back (no barrier).
            const bool next = (h + 1 == nHops) && pn < nPanels;
            f32x4* out = reinterpret_cast<f32x4*>(Xout + (int64_t)h * tapStride) + (int64_t)p * pstride4;
            const f32x4* nsrc = reinterpret_cast<const f32x4*>(Xin) + (int64_t)pn * pstride4;
#pragma unroll 2
            for (int j = 0; j < nj; ++j) {
                const int idx = tid + j * T;
                if (idx < N) {
                    const f32x4 v = lds4[idx];
                    if (store_mode == 2)
                        __builtin_nontemporal_store(v, out + idx);
                    else
                        out[idx] = v;
                }
                if (next) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the chunk is in registers before the DMA may overwrite it
                    dma_chunk(nsrc, j);
                }
            }
        }
        p = pn;
        if (p >= nPanels) break;
    }
}

int chain_num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace

bool gf_chain_available(const gf_plan* plan, int op) { return plan->mat[op].cn_waves > 0; }

// nHops successive hops of every panel: tap h + 1 = op(S) tap h, tap 0 = Xin, tap h + 1 at Xout + h * tapStride.
int gf_spmm_chain_launch(const gf_plan* plan, int op, const float* Xin, float* Xout, int nPanels, int nHops, int64_t tapStride,
                         hipStream_t st) {
    const gf_csr_dev& m = plan->mat[op];
    const int N = plan->n;
    GF_REQUIRE_ARG(m.cn_waves > 0, "gf_khop_panel: the plan has no chain image (N = %d > %d?)", N, kPanelMaxNodes);
    const bool uniform = m.pn_uniform && g_tune.panel_uniform;
    const size_t lds = (size_t)(N + 1) * 16;
    const int thr = m.cn_waves * 64;
    int wgPerCU = (int)((160 * 1024) / (lds < 1024 ? 1024 : lds));
    const int waveCap = 32 / m.cn_waves;
    if (wgPerCU > waveCap) wgPerCU = waveCap;
    if (wgPerCU > 8) wgPerCU = 8;
    if (wgPerCU < 1) wgPerCU = 1;
    int64_t grid = (int64_t)chain_num_cus() * wgPerCU;
    if (grid > nPanels) grid = nPanels;
    auto kern = uniform ? spmm_chain_kernel<1> : spmm_chain_kernel<0>;
    if (lds > 64 * 1024) GF_HIP(gf_grant_lds((const void*)kern, lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(thr), lds, st, m.cn_gtab, m.cn_rowoff,
                       uniform ? (const void*)m.cn_col4 : (const void*)m.cn_col2, m.cn_val4, m.pn_uval, Xin, Xout, N, nPanels, m.cn_sets,
                       nHops, tapStride, g_tune.spmm_store);
    GF_LAUNCH_CHECK("spmm_chain_kernel");
    return GF_OK;
}
