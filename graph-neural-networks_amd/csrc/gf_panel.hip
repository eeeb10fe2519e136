// gf_panel.hip -- the K-hop GSO-signal product with the gather served from LDS (column-panel pipeline).
// Same operation as gf_spmm.hip (reference graphML.py:158-161, x = torch.matmul(x, S) per tap), different data layout:
//
//   column panels   Xp[P][N][4],  P = B*C/4:  panel p holds 4 consecutive signal columns (b, c..c+3) of every node,
//                   16 bytes per node, N*16 bytes per panel -- at most 160 KiB, i.e. ONE panel fits a CU's LDS.
//
// Why: a hop gathers nnz*B*C*4 bytes (each node's row once per neighbour, ~10x the algorithmic read).  Served from L2
// that stream is capped by the L2 itself (34 TB/s peak for 8 TB/s of HBM: <= 85 % of the HBM roofline even at 100 % hits,
// 39 % measured, profiles/r01_g_*).  Here a workgroup stages one panel in LDS (one coalesced read of its N*16 bytes =
// exactly the algorithmic X read), every gather becomes a ds_read_b128, and the only thing streamed per panel besides
// the panel itself is the (column, value) list -- 2 or 6 bytes per edge, coalesced, L2-resident.
//
// Kernel (persistent, one workgroup per LDS-full; 16 / 8 / 4 waves by N):
//   * waves are specialised: the first half compute, the second half are loaders.  A wave's vector-memory results
//     return in issue order, so a wave that had the next panel's HBM loads in flight would stall its L2-latency entry
//     loads behind them; the loader waves hold the next panel in registers (20 x 16 B per lane), sleep on it, and copy
//     it into LDS between two workgroup barriers once the compute waves are done with the current panel.
//   * compute: lane = row, wave = slice of 64 consecutive rows in NATURAL order (the output store of a slice is one
//     contiguous 1 KiB, no node permutation exists anywhere).  A lane's neighbours come in groups of 4 (one 8-byte
//     column load + one 16-byte value load per 4 gathers); rows of a slice differ in length: group-step j holds only the
//     lanes that still have neighbours, compacted (ballot + mbcnt give a lane its slot), so at most 3 padding entries per
//     row are read; the groups of the next 8 steps -- across slice boundaries -- are requested before the current 8 are gathered.
//   * plan-time neighbour order (gf_plan.hip) spreads the 16 lanes of each ds_read_b128 service group over distinct
//     16-byte bank quads, cutting the random-access LDS conflicts.
//   * per-row sums run in the plan's fixed neighbour order: bitwise run-to-run deterministic, no atomics.
#include "gf_common.h"

namespace {

constexpr int kGC = 2;   // entry groups (of 4 neighbours) per lane gathered per round
constexpr int kNV = 20;  // float4 per loader lane: 64 * kNV * 16 B per loader wave
typedef float f32x4 __attribute__((ext_vector_type(4)));


template <int UNIFORM, int PACE, int UNIFIED>
__global__ __launch_bounds__(1024) void spmm_panel_kernel(const int2* __restrict__ slice, const uint16_t* __restrict__ degs, const int32_t* __restrict__ rows,
                                                          const uint2* __restrict__ cols, const float4* __restrict__ vals,
                                                          float uval, const float* __restrict__ Xin, float* __restrict__ Xout,
                                                          int N, int nSlices, int nPanels, int sentinel, int store_mode, int debug, int stagger) {
    extern __shared__ __attribute__((aligned(16))) float4 panel[];  // [N + 1]: the panel + one zero slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // UNIFIED: every wave loads its share of the panel, then every wave computes (load and compute phases alternate per CU;
    // other CUs are in other phases).  Otherwise: half the waves are loaders that prefetch the next panel (see above).
    const int CW = UNIFIED ? (int)(blockDim.x >> 6) : (int)(blockDim.x >> 7);
    const int64_t pstride = (int64_t)N * 4;  // floats per panel
    int p = blockIdx.x;
    if (p >= nPanels) return;  // whole workgroup

    if (!UNIFIED && wave >= CW) {
        // ------------------------------------------------------------------ loader waves
        const int lt = (wave - CW) * 64 + lane, nl = CW * 64;
        // every lane always loads kNV rows (index clamped into the panel: unconditional loads keep pre[] in registers)
        f32x4 pre[kNV];
        int idxs[kNV];
#pragma unroll
        for (int j = 0; j < kNV; ++j) idxs[j] = min(lt + j * nl, N - 1);
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(Xin + (int64_t)p * pstride);
#pragma unroll
            for (int j = 0; j < kNV; ++j) pre[j] = src[idxs[j]];
        }
        f32x4* lds4 = reinterpret_cast<f32x4*>(panel);
        if (lt == 0) lds4[N] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the zero slot exhausted rows gather from
        for (;;) {
#pragma unroll
            for (int j = 0; j < kNV; ++j)
                if (lt + j * nl < N) lds4[idxs[j]] = pre[j];
            __syncthreads();  // B1: panel p is in LDS
            const int pn = p + (int)gridDim.x;
            const bool more = pn < nPanels;
            const f32x4* src = reinterpret_cast<const f32x4*>(Xin + (int64_t)(more ? pn : p) * pstride);
            // The next panel streams in while the compute waves work on panel p.  PACE bounds the HBM loads a loader wave
            // keeps in flight: dumped all at once, the 160 KiB fill the CU's L1 miss queue and the compute waves' L2-resident
            // entry loads wait behind them (measured: hop 265 us unpaced).
#pragma unroll
            for (int j = 0; j < kNV; ++j) {
                pre[j] = src[idxs[j]];
                if (PACE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (PACE == 2 && (j & 1) == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                if (PACE == 4 && (j & 1) == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                if (PACE == 8 && (j & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            __syncthreads();  // B2: compute waves are done reading panel p
            if (!more) break;
            p = pn;
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    // Two register sets (A, B) ping-pong: while the gathers of the set in hand run, the other set is being filled with the
    // NEXT chunk's entry groups -- across slice boundaries.  No register copies between the sets (a copy would read the
    // freshly requested registers and force a full vmcnt(0) at the loop's back edge), and every request is issued on every
    // path: a skipped load would make the in-order vmcnt bookkeeping of the other set unknowable (again vmcnt(0)).
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const f32x4* lds4 = reinterpret_cast<const f32x4*>(panel);
    const u32x2* col4 = reinterpret_cast<const u32x2*>(cols);
    const f32x4* val4 = reinterpret_cast<const f32x4*>(vals);

    // request group-steps [jj, jj + kGC) for a lane that owns gl groups; pb = offset of group-step jj's first group.
    // Lanes without a group at a step read the sentinel group {columns N = the panel's zero slot, values 0}.
    auto load_chunk = [&](u32x2 (&cc)[kGC], f32x4 (&vv)[kGC], int gl, int jj, int& pb) {
#pragma unroll
        for (int g = 0; g < kGC; ++g) {
            const bool act = (jj + g) < gl;
            const unsigned long long m = __ballot(act);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const int e = act ? pb + rank : sentinel;
            cc[g] = col4[e];
            if (!UNIFORM) vv[g] = val4[e];
            pb += __popcll(m);
        }
    };

    if (UNIFIED && tid == 0) reinterpret_cast<f32x4*>(panel)[N] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the zero slot
    if (UNIFIED) {
        // De-synchronise the workgroups once: every CU alternates an HBM-bound load phase and an LDS/issue-bound compute phase;
        // started together they stay in lock step (HBM idle while everyone computes: measured 224 us = 53 load + 138 compute
        // + 33 lost), started spread over one period the phases of different CUs interleave.
        const int phase = (int)((blockIdx.x >> 3) & 7);   // workgroups b, b+8, ... share an XCD: spread within each XCD
        for (int i = 0; i < phase * stagger; ++i) __builtin_amdgcn_s_sleep(64);  // 64 * 64 cycles ~ 2 us each
    }
    for (;;) {
        if (UNIFIED && debug != 2) {
            constexpr int kNVU = 10;  // N <= 10 * blockDim.x
            const f32x4* src = reinterpret_cast<const f32x4*>(Xin + (int64_t)p * pstride);
            f32x4* dst = reinterpret_cast<f32x4*>(panel);
            f32x4 tmp[kNVU];
            const int nthr = (int)blockDim.x;
#pragma unroll
            for (int j = 0; j < kNVU; ++j) tmp[j] = src[min(tid + j * nthr, N - 1)];
#pragma unroll
            for (int j = 0; j < kNVU; ++j)
                if (tid + j * nthr < N) dst[tid + j * nthr] = tmp[j];
        }
        __syncthreads();  // B1
        if (wave < nSlices && debug != 1) {
            float* outp = Xout + (int64_t)p * pstride;
            int s = wave;                       // slice in hand
            int2 si = slice[s];                 // {group offset, group-steps}
            int gl = ((int)degs[s * 64 + lane] + 3) >> 2;   // this lane's groups
            int sn = s + CW;                    // next slice of this wave (its header is fetched one slice ahead)
            int2 sin = make_int2(0, 0);
            int gln = 0;
            if (sn < nSlices) {
                sin = slice[sn];
                gln = ((int)degs[sn * 64 + lane] + 3) >> 2;
            }
            int pb = si.x;                      // group offset up to which requests have been issued
            int j0 = 0;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            u32x2 cA[kGC], cB[kGC];
            f32x4 vA[kGC], vB[kGC];
            load_chunk(cA, vA, gl, 0, pb);

            // gathers + FMAs of the chunk held in (cc, vv) = group-steps [j0, j0 + kGC) of slice s, after requesting the
            // following chunk into (cn, vn).  Returns true when the wave has no slice left.
            auto process = [&](u32x2 (&cc)[kGC], f32x4 (&vv)[kGC], u32x2 (&cn)[kGC], f32x4 (&vn)[kGC]) -> bool {
                const bool same = (j0 + kGC) < si.y;  // wave-uniform: the next chunk belongs to the same slice
                const int gn = same ? gl : gln;       // gln = 0 when the wave has no further slice: all-sentinel requests
                const int jn = same ? j0 + kGC : 0;
                pb = same ? pb : sin.x;
                load_chunk(cn, vn, gn, jn, pb);
#pragma unroll
                for (int g = 0; g < kGC; ++g) {
                    const unsigned c01 = cc[g].x, c23 = cc[g].y;
                    const f32x4 x0 = lds4[c01 & 0xffffu];
                    const f32x4 x1 = lds4[c01 >> 16];
                    const f32x4 x2 = lds4[c23 & 0xffffu];
                    const f32x4 x3 = lds4[c23 >> 16];
                    if (UNIFORM) {
                        acc0 += x0;
                        acc1 += x1;
                        acc0 += x2;
                        acc1 += x3;
                    } else {
                        acc0 += vv[g].x * x0;   // contracted to FMAs; two accumulators: fixed order, shorter dependency chain
                        acc1 += vv[g].y * x1;
                        acc0 += vv[g].z * x2;
                        acc1 += vv[g].w * x3;
                    }
                }
                if (same) {
                    j0 += kGC;
                    return false;
                }
                const int row = rows ? rows[s * 64 + lane] : (s * 64 + lane < N ? s * 64 + lane : -1);
                if (row >= 0) {
                    f32x4 acc = acc0 + acc1;
                    if (UNIFORM) acc *= uval;
                    f32x4* dst = reinterpret_cast<f32x4*>(outp + (int64_t)row * 4);
                    if (store_mode == 2)
                        __builtin_nontemporal_store(acc, dst);
                    else
                        *dst = acc;
                }
                acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                s = sn;
                if (s >= nSlices) return true;
                si = sin;
                gl = gln;
                j0 = 0;
                sn = s + CW;
                gln = 0;
                if (sn < nSlices) {
                    sin = slice[sn];
                    gln = ((int)degs[sn * 64 + lane] + 3) >> 2;
                }
                return false;
            };
            for (;;) {
                if (process(cA, vA, cB, vB)) break;
                if (process(cB, vB, cA, vA)) break;
            }
        }
        __syncthreads();  // B2
        p += (int)gridDim.x;
        if (p >= nPanels) break;
    }
}

// x[B, C, Nin] (reference layout, node index contiguous) -> Xp[B*C/4][N][4]; rows n >= Nin are zero
// (== GraphFilter.forward's zero padding, graphML.py:2131-2135).  C % 4 == 0.
__global__ __launch_bounds__(256) void pack_panels_kernel(const float* __restrict__ x, float* __restrict__ Xp, int Nin, int N,
                                                          int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t p = idx / N;
        const int n = (int)(idx - p * N);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < Nin) {
            const float* src = x + p * 4 * Nin + n;
            v.x = src[0];
            v.y = src[Nin];
            v.z = src[2 * (int64_t)Nin];
            v.w = src[3 * (int64_t)Nin];
        }
        reinterpret_cast<float4*>(Xp)[idx] = v;
    }
}

// Xp[B*C/4][N][4] -> x[B, C, Nout], nodes n < Nout (tests / debugging; the contraction writes the reference layout itself)
__global__ __launch_bounds__(256) void unpack_panels_kernel(const float* __restrict__ Xp, float* __restrict__ x, int N, int Nout,
                                                            int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t p = idx / Nout;
        const int n = (int)(idx - p * Nout);
        const float4 v = reinterpret_cast<const float4*>(Xp)[p * N + n];
        float* dst = x + p * 4 * Nout + n;
        dst[0] = v.x;
        dst[Nout] = v.y;
        dst[2 * (int64_t)Nout] = v.z;
        dst[3 * (int64_t)Nout] = v.w;
    }
}

int num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

unsigned grid_for(int64_t items) {
    int64_t blocks = (items + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

bool gf_panel_supported(const gf_plan* const* plans, int E, int G, int F) {
    if (G % 8 != 0 || F % 8 != 0 || G > 128 || F > 128) return false;  // hop width % 4 and the MFMA contraction's Cin % 8
    for (int e = 0; e < E; ++e) {
        if (!plans[e] || plans[e]->n > kPanelMaxNodes || plans[e]->n < 8) return false;
        if (plans[e]->mat[0].pn_slices == 0 || plans[e]->mat[1].pn_slices == 0) return false;
    }
    return true;
}

int gf_pack_panels_launch(const float* x, float* Xp, int B, int C, int Nin, int N, hipStream_t st) {
    const int64_t total = (int64_t)B * (C / 4) * N;
    hipLaunchKernelGGL(pack_panels_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, Xp, Nin, N, total);
    GF_LAUNCH_CHECK("pack_panels_kernel");
    return GF_OK;
}

int gf_spmm_panel_launch(const gf_plan* plan, int op, const float* Xin, float* Xout, int nPanels, hipStream_t st) {
    const gf_csr_dev& m = plan->mat[op];
    const int N = plan->n;
    GF_REQUIRE_ARG(m.pn_slices > 0, "gf_spmm_hop_panel: the plan has no panel image (N = %d > %d?)", N, kPanelMaxNodes);
    const int threads = N > 5120 ? 1024 : (N > 2560 ? 512 : 256);
    const size_t lds = (size_t)(N + 1) * 16;
    int wgPerCU = (int)((160 * 1024) / (lds < 1024 ? 1024 : lds));
    const int waveCap = 32 / (threads / 64);
    if (wgPerCU > waveCap) wgPerCU = waveCap;
    if (wgPerCU < 1) wgPerCU = 1;
    int64_t grid = (int64_t)num_cus() * wgPerCU;
    if (grid > nPanels) grid = nPanels;
    const bool uniform = m.pn_uniform && g_tune.panel_uniform;
    typedef void (*kern_t)(const int2*, const uint16_t*, const int32_t*, const uint2*, const float4*, float, const float*, float*, int, int, int, int, int, int, int);
    kern_t kern;
    if (g_tune.panel_mode == 0) {
        kern = uniform ? spmm_panel_kernel<1, 0, 1> : spmm_panel_kernel<0, 0, 1>;
    } else {
        switch (g_tune.panel_pace) {
            case 1: kern = uniform ? spmm_panel_kernel<1, 1, 0> : spmm_panel_kernel<0, 1, 0>; break;
            case 0: kern = uniform ? spmm_panel_kernel<1, 0, 0> : spmm_panel_kernel<0, 0, 0>; break;
            default: kern = uniform ? spmm_panel_kernel<1, 4, 0> : spmm_panel_kernel<0, 4, 0>; break;
        }
    }
    if (lds > 64 * 1024) GF_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, st, m.pn_slice, m.pn_deg, g_tune.panel_sort ? m.pn_row : nullptr, m.pn_col4, m.pn_val4, m.pn_uval, Xin,
                       Xout, N, m.pn_slices, nPanels, m.pn_sentinel, g_tune.spmm_store, g_tune.panel_debug, g_tune.panel_stagger);
    GF_LAUNCH_CHECK("spmm_panel_kernel");
    return GF_OK;
}

extern "C" int gf_pack_panels(const float* x, float* Xp, int32_t B, int32_t C, int32_t Nin, int32_t N, void* stream) {
    GF_REQUIRE_ARG(x && Xp, "gf_pack_panels: NULL tensor");
    GF_REQUIRE_SHAPE(B > 0 && C > 0 && C % 4 == 0 && Nin > 0 && N >= Nin, "gf_pack_panels: bad shape B=%d C=%d (C %% 4 == 0) Nin=%d N=%d",
                     B, C, Nin, N);
    return gf_pack_panels_launch(x, Xp, B, C, Nin, N, gf_stream(stream));
}

extern "C" int gf_unpack_panels(const float* Xp, float* x, int32_t B, int32_t C, int32_t N, int32_t Nout, void* stream) {
    GF_REQUIRE_ARG(x && Xp, "gf_unpack_panels: NULL tensor");
    GF_REQUIRE_SHAPE(B > 0 && C > 0 && C % 4 == 0 && Nout > 0 && N >= Nout, "gf_unpack_panels: bad shape B=%d C=%d (C %% 4 == 0) N=%d Nout=%d",
                     B, C, N, Nout);
    const int64_t total = (int64_t)B * (C / 4) * Nout;
    hipLaunchKernelGGL(unpack_panels_kernel, dim3(grid_for(total)), dim3(256), 0, gf_stream(stream), Xp, x, N, Nout, total);
    GF_LAUNCH_CHECK("unpack_panels_kernel");
    return GF_OK;
}

extern "C" int gf_spmm_hop_panel(const gf_plan* plan, int32_t op, const float* Xin, float* Xout, int32_t n_panels, void* stream) {
    GF_REQUIRE_ARG(plan && Xin && Xout, "gf_spmm_hop_panel: NULL argument");
    GF_REQUIRE_ARG(op == GF_OP_FWD || op == GF_OP_BWD, "gf_spmm_hop_panel: op = %d", op);
    GF_REQUIRE_ARG(Xin != Xout, "gf_spmm_hop_panel: in-place hop is not supported");
    GF_REQUIRE_SHAPE(n_panels > 0, "gf_spmm_hop_panel: n_panels = %d", n_panels);
    GF_REQUIRE_SHAPE(plan->n <= kPanelMaxNodes, "gf_spmm_hop_panel: N = %d exceeds the LDS panel limit %d", plan->n, kPanelMaxNodes);
    return gf_spmm_panel_launch(plan, op, Xin, Xout, n_panels, gf_stream(stream));
}

extern "C" int gf_time_spmm_hop_panel(const gf_plan* plan, int32_t op, const float* Xin, float* Xout, int32_t n_panels, int32_t iters,
                                      void* stream, float* avg_ms) {
    GF_REQUIRE_ARG(avg_ms && iters > 0, "gf_time_spmm_hop_panel: bad iters / NULL avg_ms");
    hipStream_t st = gf_stream(stream);
    hipEvent_t e0, e1;
    GF_HIP(hipEventCreate(&e0));
    GF_HIP(hipEventCreate(&e1));
    int rc = gf_spmm_hop_panel(plan, op, Xin, Xout, n_panels, stream);  // warm-up (also validates arguments)
    if (rc == GF_OK) {
        GF_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < iters && rc == GF_OK; ++i) rc = gf_spmm_hop_panel(plan, op, Xin, Xout, n_panels, stream);
        GF_HIP(hipEventRecord(e1, st));
        GF_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        GF_HIP(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
