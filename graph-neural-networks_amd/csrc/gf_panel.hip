// gf_panel.hip -- the K-hop GSO-signal product with the gather served from LDS (column-panel pipeline).
// Same operation as gf_spmm.hip (reference graphML.py:158-161, x = torch.matmul(x, S) per tap), different data layout:
//
//   column panels   Xp[P][N][4],  P = B*C/4:  panel p holds 4 consecutive signal columns (b, c..c+3) of every node,
//                   16 bytes per node, N*16 bytes per panel -- at most 160 KiB, i.e. ONE panel fits a CU's LDS.
//
// Why: a hop gathers nnz*B*C*4 bytes (each node's row once per neighbour, ~10x the algorithmic read).  Served from L2
// that stream goes through the CU's L1 miss path (~64 lines in flight x 128 B / ~250 cycles ~ 33 B/clk/CU: the node-major
// kernel sits at 39 % of the HBM roofline and every variant of it clustered there, profiles/r01_b..g).  Here a workgroup
// stages one panel in LDS (one coalesced read of its N*16 bytes = exactly the algorithmic X read), every gather is a
// ds_read_b128, and the only thing streamed per panel besides the panel itself is the (column, value) list -- 2 or 6 bytes
// per edge, coalesced, L2-resident.
//
// Kernel (persistent; 16 / 8 / 4 waves per workgroup by N, as many workgroups per CU as LDS allows):
//   * per panel: every wave loads its share of the panel into LDS (HBM-bound phase), barrier, every wave computes its slices
//     (LDS / issue-bound phase), barrier.  (Round 1 started the workgroups of a full-LDS launch staggered by a timing-tuned
//     s_sleep ladder so that the phases of different CUs interleave: 224 -> 189 us at N = 1e4.  Full-LDS panels now run in the
//     chain kernel, gf_chain.hip, which needs no such constant; this kernel serves launches with several workgroups per CU,
//     whose phases interleave by themselves, and launches with few panels.)
//     (A variant with dedicated loader waves prefetching the next panel into registers overlapped perfectly but left only
//     2 compute waves per SIMD: 255 us.  A wave's vector-memory results return in issue order, so a computing wave cannot
//     keep HBM loads in flight without stalling its own L2-latency entry loads behind them.)
//   * compute: work unit = octet (8 consecutive rows = one 128-byte output line); the plan sorts octets by their longest row,
//     a slice = 8 octets = one wave, lane = row.  A lane's neighbours come in groups of 4 (one 8-byte load of 4 x 16-bit
//     columns + one 16-byte load of 4 values per 4 gathers) laid out as an ELL block per slice: the address of a group is
//     base + j*64 + lane -- no per-lane bookkeeping; empty slots point at a zero slot of the panel.  Two register sets
//     ping-pong so that the groups of the next 8 steps -- across slice boundaries -- are in flight while the current 8 steps
//     gather; every path issues the same number of requests (a skipped request would make the in-order vmcnt bookkeeping
//     of the other set unknowable and force vmcnt(0) waits).
//   * plan-time neighbour order (gf_plan.hip) spreads the 16 lanes of each ds_read_b128 service group over distinct
//     16-byte bank quads, cutting the random-access LDS conflicts.
//   * per-row sums run in the plan's fixed neighbour order: bitwise run-to-run deterministic, no atomics.
#include "gf_common.h"

namespace {

#ifndef GF_PANEL_KGC
#define GF_PANEL_KGC 2
#endif
#ifndef GF_PANEL_BATCH
#define GF_PANEL_BATCH 0
#endif
constexpr int kGC = GF_PANEL_KGC;   // entry groups (of 4 neighbours) per lane gathered per round
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int UNIFORM> struct ColWord { typedef u32x2 type; };   // weighted stream: 4 x 16-bit columns
template <> struct ColWord<1> { typedef u32x4 type; };            // value-free stream: 4 LDS byte offsets

template <int UNIFORM, int NP>
__global__ __launch_bounds__(1024) void spmm_panel_kernel(const int2* __restrict__ slice, const int32_t* __restrict__ octs,
                                                          const void* __restrict__ cols, const float4* __restrict__ vals,
                                                          float uval, const float* __restrict__ Xin, float* __restrict__ Xout,
                                                          int N, int nSlices, int nPanels, int sentinel, int store_mode,
                                                          int ush, int rotate, int split) {
    extern __shared__ __attribute__((aligned(16))) float4 panel[];  // [N + 1]: the panel + one zero slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CW = (int)(blockDim.x >> 6);
    const int64_t pstride = (int64_t)N * 4;  // floats per panel
    const int nGroups = (nPanels + NP - 1) / NP;  // passes: NP panels each
    // split > 1 (fewer passes than CUs): `split` workgroups share a pass -- each loads the panel (the second copy comes from L2) and
    // computes every split-th slice, so a small batch still occupies every CU in the compute phase (2/3 of the time)
    const int part = (int)(blockIdx.x % (unsigned)split);
    int p = (int)(blockIdx.x / (unsigned)split);
    if (p >= nGroups) return;  // whole workgroup
    f32x4* lds4 = reinterpret_cast<f32x4*>(panel);
    typedef typename ColWord<UNIFORM>::type colw;
    const colw* col4 = reinterpret_cast<const colw*>(cols) + lane;
    // gathers address LDS absolutely: the dynamic panel is this kernel's only LDS object, so it starts at LDS address 0 and a
    // stored byte offset IS the ds_read address (through the `panel` symbol the compiler adds a relocatable base -- a useless
    // v_add per gather in an issue-bound loop).  A static __shared__ object would shift the panel: the launcher asks the runtime that there
    // is none (gf_require_no_static_lds) and refuses the launch otherwise
    const unsigned lds_zero = 0u;
    const f32x4* val4 = reinterpret_cast<const f32x4*>(vals) + lane;

    // NP panels per pass live side by side in LDS (region k = float4 index k*(N+1), each with its own zero slot): one entry word
    // then drives NP gathers -- the (column, value) stream and its bookkeeping are amortised over NP x 16 bytes per edge.
    const unsigned regionB = (unsigned)(N + 1) * 16u;
    if (tid < NP) lds4[tid * (N + 1) + N] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the zero slots empty ELL slots gather from
    const int rot = rotate ? (int)(((blockIdx.x / (unsigned)split) * 41u) % (unsigned)nSlices) : 0;  // the same for the workgroups sharing a pass
    // request group-rows [g0, g0 + kGC) of an ELL block (g0 = absolute group-row index)
    auto load_chunk = [&](colw (&cc)[kGC], f32x4 (&vv)[kGC], int g0) {
#pragma unroll
        for (int g = 0; g < kGC; ++g) {
            cc[g] = col4[(int64_t)(g0 + g) * 64];
            if (!UNIFORM) vv[g] = val4[(int64_t)(g0 + g) * 64];
        }
    };

    for (;;) {
        const float* srcp = Xin + (int64_t)p * NP * pstride;
        float* outp = Xout + (int64_t)p * NP * pstride;
        const int nvalid = min(NP, nPanels - p * NP);  // panels of this pass (the last pass may be short)
        {
            // HBM-bound phase: every wave loads its share of the panel (N <= kNVU * blockDim.x rows of 16 bytes).
            // (Requesting the NEXT panel from inside the compute phase instead -- registers, one HBM-latency stall per wave
            // and panel -- was measured and gave nothing: 184 vs 181 us; a CU pulls at most ~22 GB/s from HBM and the
            // entry streams compete for the same L1 miss queue.)
            constexpr int kNVU = 10;
            const int nthr = (int)blockDim.x;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (k >= nvalid) break;
                const f32x4* src = reinterpret_cast<const f32x4*>(srcp + (int64_t)k * pstride);
                f32x4 tmp[kNVU];
#pragma unroll
                for (int j = 0; j < kNVU; ++j) tmp[j] = src[min(tid + j * nthr, N - 1)];  // (a non-temporal hint here: no effect)
#pragma unroll
                for (int j = 0; j < kNVU; ++j)
                    if (tid + j * nthr < N) lds4[k * (N + 1) + tid + j * nthr] = tmp[j];
            }
        }
        __syncthreads();  // B1: panel p is in LDS
        if (wave + part * CW < nSlices) {
            const int CWS = CW * split;         // slice stride of this wave
            int s = wave + part * CW;           // slice in hand
            // Workgroups walk the slice list rotated by a workgroup-specific offset: at any moment the 256 CUs read different parts
            // of the (shared, L2-resident) entry arrays instead of queueing on the same L2 channels.
            auto rs = [&](int q) { const int t = q + rot; return t >= nSlices ? t - nSlices : t; };
            int2 si = slice[rs(s)];             // {group-row offset, group-rows}
            int oc = octs[(rs(s) << (6 - ush)) + (lane >> ush)];
            int sn = s + CWS;                   // next slice of this wave (its header is fetched one slice ahead)
            int2 sin = make_int2(sentinel, 0);
            int ocn = -1;
            if (sn < nSlices) {
                sin = slice[rs(sn)];
                ocn = octs[(rs(sn) << (6 - ush)) + (lane >> ush)];
            }
            int j0 = 0;
            f32x4 acc0[NP], acc1[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) acc0[k] = acc1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            colw cA[kGC], cB[kGC];
            f32x4 vA[kGC], vB[kGC];
            load_chunk(cA, vA, si.y > 0 ? si.x : sentinel);

            // gathers + FMAs of the chunk held in (cc, vv) = group-rows [j0, j0 + kGC) of slice s, after requesting the
            // following chunk into (cn, vn).  Returns true when the wave has no slice left.
            auto process = [&](colw (&cc)[kGC], f32x4 (&vv)[kGC], colw (&cn)[kGC], f32x4 (&vn)[kGC]) -> bool {
                const bool same = (j0 + kGC) < si.y;  // wave-uniform: the next chunk belongs to the same slice
                // next chunk: same slice (group-rows per slice are even: a chunk never straddles two blocks), or the head of
                // the next slice (the sentinel rows when it is empty / absent)
                const int gnext = same ? si.x + j0 + kGC : (sin.y > 0 ? sin.x : sentinel);
                load_chunk(cn, vn, gnext);
#pragma unroll
                for (int g = 0; g < kGC; ++g) {
                    // a slice with an odd number of group-rows: the round's second group belongs to the next block -- its
                    // (already requested) entries are dropped by a wave-uniform branch instead of gathering padding
                    if (g > 0 && j0 + g >= si.y) break;
                    unsigned o0, o1, o2, o3;  // LDS byte offsets of the 4 gathered rows
                    if constexpr (UNIFORM) {
                        o0 = cc[g].x, o1 = cc[g].y, o2 = cc[g].z, o3 = cc[g].w;
                    } else {
                        o0 = (cc[g].x & 0xffffu) << 4, o1 = (cc[g].x >> 16) << 4, o2 = (cc[g].y & 0xffffu) << 4, o3 = (cc[g].y >> 16) << 4;
                    }
                    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const unsigned rb = lds_zero + k * regionB;
                        const f32x4 x0 = *reinterpret_cast<lds_f32x4*>(o0 + rb);
                        const f32x4 x1 = *reinterpret_cast<lds_f32x4*>(o1 + rb);
                        const f32x4 x2 = *reinterpret_cast<lds_f32x4*>(o2 + rb);
                        const f32x4 x3 = *reinterpret_cast<lds_f32x4*>(o3 + rb);
                        if (UNIFORM) {
                            acc0[k] += x0;
                            acc1[k] += x1;
                            acc0[k] += x2;
                            acc1[k] += x3;
                        } else {
                            acc0[k] += vv[g].x * x0;   // contracted to FMAs; two accumulators: fixed order, shorter dependency chain
                            acc1[k] += vv[g].y * x1;
                            acc0[k] += vv[g].z * x2;
                            acc1[k] += vv[g].w * x3;
                        }
                    }
                }
                if (same) {
                    j0 += kGC;
                    return false;
                }
                const int row = (oc << ush) + (lane & ((1 << ush) - 1));
                if (oc >= 0 && row < N) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        if (k >= nvalid) break;
                        f32x4 acc = acc0[k] + acc1[k];
                        if (UNIFORM) acc *= uval;
                        f32x4* dst = reinterpret_cast<f32x4*>(outp + (int64_t)k * pstride + (int64_t)row * 4);
                        if (store_mode == 2)
                            __builtin_nontemporal_store(acc, dst);
                        else
                            *dst = acc;
                    }
                }
#pragma unroll
                for (int k = 0; k < NP; ++k) acc0[k] = acc1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                s = sn;
                if (s >= nSlices) return true;
                si = sin;
                oc = ocn;
                j0 = 0;
                sn = s + CWS;
                sin = make_int2(sentinel, 0);
                ocn = -1;
                if (sn < nSlices) {
                    sin = slice[rs(sn)];
                    ocn = octs[(rs(sn) << (6 - ush)) + (lane >> ush)];
                }
                return false;
            };
            for (;;) {
                if (process(cA, vA, cB, vB)) break;
                if (process(cB, vB, cA, vA)) break;
            }
        }
        __syncthreads();  // B2: every wave is done reading panel p
        p += (int)(gridDim.x / (unsigned)split);
        if (p >= nGroups) break;
    }
}

// ---- double-buffered variant (round 4): the load of the NEXT pass's panels runs under the gathers of the current one ----------------
// spmm_panel_kernel alternates a load phase (HBM-bound) and a gather phase (LDS-bound) per workgroup and relies on the two or three
// workgroups of a CU being in different phases.  Measured at N = 1682 (config 3: weighted kNN, width 64, tools/panel_w_probe.py): load +
// store skeleton alone 35 us per hop, LDS busy 30 us per hop (SQ_LDS_IDX_ACTIVE; a third of it bank conflicts, which is what random
// columns cost with 16 lanes per service group), the hop 65 us: the phases add up instead of overlapping.  Here ONE workgroup owns two
// buffers of NP panels: loader waves fetch pass i + 1 with LDS-DMA (no registers, and their vmcnt queue holds nothing else) while the
// gather waves work on pass i and store its results from registers; one LDS-only barrier per pass (stores stay in flight across it).
// Slices are CLAIMED from an LDS counter in the plan's order (longest first): waves that finish early take the next slice instead of
// waiting at the barrier for the wave that drew the long ones.  Columns come from the 16-bit stream for weighted and equal-weight
// GSOs alike (the buffer base is folded into the shift: v_lshl_add_u32), so the equal-weight stream is half the per-hop kernel's.
// Arithmetic per row is the per-hop kernel's: same ELL image, same two accumulators, same order -- bitwise the same results.
template <int UNIFORM, int NP>
__global__ __launch_bounds__(1024) void spmm_panel_db_kernel(const int2* __restrict__ slice, const int32_t* __restrict__ octs,
                                                             const uint2* __restrict__ cols, const float4* __restrict__ vals, float uval,
                                                             const float* __restrict__ Xin, float* __restrict__ Xout, int N, int nSlices,
                                                             int nPanels, int sentinel, int store_mode, int ush, int nLoaders) {
    extern __shared__ __attribute__((aligned(16))) float4 panel[];  // [2 buffers][NP regions][N + 1] float4, then two claim counters
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nW = (int)(blockDim.x >> 6), Wg = nW - nLoaders;
    const bool gatherer = wave < Wg;
    // the panels start at LDS address 0 (absolute ds_read addresses; no static LDS: checked by the launcher, gf_require_no_static_lds)
    f32x4* lds4 = reinterpret_cast<f32x4*>(panel);
    const int region4 = N + 1;
    const unsigned regionB = (unsigned)region4 * 16u, bufB = NP * regionB;
    unsigned* ctr = reinterpret_cast<unsigned*>(lds4 + 2 * NP * region4);
    const int64_t pstride = (int64_t)N * 4;
    const int nGroups = (nPanels + NP - 1) / NP;
    const int nChunks = (N + 63) >> 6;
    int p = (int)blockIdx.x;
    if (p >= nGroups) return;
    if (tid < 2 * NP) lds4[tid * region4 + N] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tid == 0) ctr[0] = ctr[1] = 0u;
    typedef __attribute__((address_space(3))) void lds_void;
    // the panels of pass `pass` into buffer `buf`, 64-row chunks w, w + nw, ... (LDS-DMA: 1 KiB per instruction at a wave-uniform base)
    auto dma_pass = [&](int pass, int buf, int w, int nw) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (pass * NP + k >= nPanels) break;
            const f32x4* src = reinterpret_cast<const f32x4*>(Xin + ((int64_t)pass * NP + k) * pstride);
            const unsigned base = (unsigned)buf * bufB + k * regionB;
            for (int c = w; c < nChunks; c += nw) {
                const int row0 = c * 64;
                if (row0 + lane < N) __builtin_amdgcn_global_load_lds(src + row0 + lane, (lds_void*)(uintptr_t)(base + (unsigned)row0 * 16u), 16, 0, 0);
            }
        }
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // (the waits on the DMA go through the builtin: the compiler's wait-count pass tracks LDS-DMA as "may write any LDS address" and,
    //  not seeing a wait it understands, would put vmcnt(0) in front of every ds_read of the gather loop -- i.e. behind the entry
    //  loads it has just issued for the NEXT chunk)
    constexpr int kWaitVm0 = 0x0F70;   // s_waitcnt vmcnt(0), gfx9 encoding (expcnt / lgkmcnt fields at their maxima)
    dma_pass(p, 0, wave, nW);
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    __syncthreads();

    const uint2* col2 = cols + lane;
    const f32x4* val4 = reinterpret_cast<const f32x4*>(vals) + lane;
    auto load_chunk = [&](uint2 (&cc)[kGC], f32x4 (&vv)[kGC], int g0) {
#pragma unroll
        for (int g = 0; g < kGC; ++g) {
            cc[g] = col2[(int64_t)(g0 + g) * 64];
            if (!UNIFORM) vv[g] = val4[(int64_t)(g0 + g) * 64];
        }
    };
    int cur = 0;
    for (;;) {
        const int pn = p + (int)gridDim.x;
        if (!gatherer) {
            if (pn < nGroups) dma_pass(pn, cur ^ 1, wave - Wg, nLoaders);
            if (wave == Wg && lane == 0) ctr[cur ^ 1] = 0u;   // last used in the previous pass, which ended at the barrier
            __builtin_amdgcn_s_waitcnt(kWaitVm0);   // the next pass's panels have landed before the barrier lets anyone read them
        } else {
            // a ticket = the counter before lane 0's increment.  Through inline asm: the compiler brackets a workgroup-scope atomic with
            // vmcnt(0) -- a full round trip of the entry loads and of the result stores in flight, at every slice boundary.  (One lane
            // only: 64 lanes adding to one word are 64 serialised LDS atomics, 5 us per pass.)
            const unsigned ctr_addr = (unsigned)(2 * NP * region4) * 16u + (unsigned)cur * 4u;
            auto claim = [&]() -> int {
                unsigned v = 0;
                if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ctr_addr), "v"(1u) : "memory");
                return __builtin_amdgcn_readfirstlane((int)v);
            };
            int s = claim();
            if (s < nSlices) {
                float* outp = Xout + (int64_t)p * NP * pstride;
                const int nvalid = min(NP, nPanels - p * NP);
                const unsigned bbase = (unsigned)cur * bufB;
                int2 si = slice[s];
                int oc = octs[(s << (6 - ush)) + (lane >> ush)];
                int sn = claim();
                int2 sin = make_int2(sentinel, 0);
                int ocn = -1;
                if (sn < nSlices) {
                    sin = slice[sn];
                    ocn = octs[(sn << (6 - ush)) + (lane >> ush)];
                }
                int j0 = 0;
                f32x4 acc0[NP], acc1[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) acc0[k] = acc1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                uint2 cA[kGC], cB[kGC];
                f32x4 vA[kGC], vB[kGC];
                load_chunk(cA, vA, si.y > 0 ? si.x : sentinel);
                auto process = [&](uint2 (&cc)[kGC], f32x4 (&vv)[kGC], uint2 (&cn)[kGC], f32x4 (&vn)[kGC]) -> bool {
                    const bool same = (j0 + kGC) < si.y;
                    const int gnext = same ? si.x + j0 + kGC : (sin.y > 0 ? sin.x : sentinel);
                    load_chunk(cn, vn, gnext);
#pragma unroll
                    for (int g = 0; g < kGC; ++g) {
                        if (g > 0 && j0 + g >= si.y) break;
                        typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
#if GF_PANEL_BATCH
                        f32x4 xx[NP][4];
#pragma unroll
                        for (int k = 0; k < NP; ++k) {
                            const unsigned rb = bbase + k * regionB;
                            xx[k][0] = *reinterpret_cast<lds_f32x4*>(((cc[g].x & 0xffffu) << 4) + rb);
                            xx[k][1] = *reinterpret_cast<lds_f32x4*>(((cc[g].x >> 16) << 4) + rb);
                            xx[k][2] = *reinterpret_cast<lds_f32x4*>(((cc[g].y & 0xffffu) << 4) + rb);
                            xx[k][3] = *reinterpret_cast<lds_f32x4*>(((cc[g].y >> 16) << 4) + rb);
                        }
                        asm volatile("" ::: "memory");   // all reads of the group are issued before the first FMA waits for one
#endif
#pragma unroll
                        for (int k = 0; k < NP; ++k) {
#if GF_PANEL_BATCH
                            const f32x4 x0 = xx[k][0], x1 = xx[k][1], x2 = xx[k][2], x3 = xx[k][3];
#else
                            const unsigned rb = bbase + k * regionB;
                            const unsigned o0 = ((cc[g].x & 0xffffu) << 4) + rb, o1 = ((cc[g].x >> 16) << 4) + rb;
                            const unsigned o2 = ((cc[g].y & 0xffffu) << 4) + rb, o3 = ((cc[g].y >> 16) << 4) + rb;
                            const f32x4 x0 = *reinterpret_cast<lds_f32x4*>(o0);
                            const f32x4 x1 = *reinterpret_cast<lds_f32x4*>(o1);
                            const f32x4 x2 = *reinterpret_cast<lds_f32x4*>(o2);
                            const f32x4 x3 = *reinterpret_cast<lds_f32x4*>(o3);
#endif
                            if (UNIFORM) {
                                acc0[k] += x0;
                                acc1[k] += x1;
                                acc0[k] += x2;
                                acc1[k] += x3;
                            } else {
                                acc0[k] += vv[g].x * x0;
                                acc1[k] += vv[g].y * x1;
                                acc0[k] += vv[g].z * x2;
                                acc1[k] += vv[g].w * x3;
                            }
                        }
                    }
                    if (same) {
                        j0 += kGC;
                        return false;
                    }
                    const int row = (oc << ush) + (lane & ((1 << ush) - 1));
                    if (oc >= 0 && row < N) {
#pragma unroll
                        for (int k = 0; k < NP; ++k) {
                            if (k >= nvalid) break;
                            f32x4 acc = acc0[k] + acc1[k];
                            if (UNIFORM) acc *= uval;
                            f32x4* dst = reinterpret_cast<f32x4*>(outp + (int64_t)k * pstride + (int64_t)row * 4);
                            if (store_mode == 2)
                                __builtin_nontemporal_store(acc, dst);
                            else
                                *dst = acc;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NP; ++k) acc0[k] = acc1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    s = sn;
                    if (s >= nSlices) return true;
                    si = sin;
                    oc = ocn;
                    j0 = 0;
                    sn = claim();
                    sin = make_int2(sentinel, 0);
                    ocn = -1;
                    if (sn < nSlices) {
                        sin = slice[sn];
                        ocn = octs[(sn << (6 - ush)) + (lane >> ush)];
                    }
                    return false;
                };
                for (;;) {
                    if (process(cA, vA, cB, vB)) break;
                    if (process(cB, vB, cA, vA)) break;
                }
            }
        }
        lds_barrier();   // every gatherer is done with buffer cur, the loaders' DMA into the other one has landed
        p = pn;
        cur ^= 1;
        if (p >= nGroups) break;
    }
}

// x[B, C, Nin] (reference layout, node index contiguous) -> Xp[B*C/4][N][4]; rows n >= Nin are zero
// (== GraphFilter.forward's zero padding, graphML.py:2131-2135).  C % 4 == 0.
__global__ __launch_bounds__(256) void pack_panels_kernel(const float* __restrict__ x, float* __restrict__ Xp, int Nin, int N,
                                                          int64_t total, const float* __restrict__ mask) {
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
            if (mask != nullptr) {  // the saved ReLU output, same layout as x: gradient of the fused epilogue
                const float* m = mask + p * 4 * Nin + n;
                if (!(m[0] > 0.f)) v.x = 0.f;
                if (!(m[Nin] > 0.f)) v.y = 0.f;
                if (!(m[2 * (int64_t)Nin] > 0.f)) v.z = 0.f;
                if (!(m[3 * (int64_t)Nin] > 0.f)) v.w = 0.f;
            }
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

bool gf_panel_supported(const gf_plan* const* plans, int E, int G, int F, int K) {
    auto ok = [](int w) { return w == 8 || w == 16 || w == 32 || w == 64 || w == 128; };  // the MFMA contraction's Cin tiles
    if (!ok(G) || !ok(F)) return false;
    // the panel contraction keeps its whole bank in LDS: both the forward bank (Cin = G) and the transposed one the backward
    // uses (Cin = F) must fit, otherwise the layer runs node-major (whose contraction has a generic fallback)
    const int T = gf_num_taps(E, K);
    if (!gf_contract_panel_fits(G, F, T) || !gf_contract_panel_fits(F, G, T)) return false;
    for (int e = 0; e < E; ++e) {
        if (!plans[e] || plans[e]->n > kPanelMaxNodes || plans[e]->n < 8) return false;
        if (plans[e]->mat[0].pn_slices == 0 || plans[e]->mat[1].pn_slices == 0) return false;
    }
    return true;
}

int gf_pack_panels_launch(const float* x, float* Xp, int B, int C, int Nin, int N, hipStream_t st, const float* mask) {
    const int64_t total = (int64_t)B * (C / 4) * N;
    hipLaunchKernelGGL(pack_panels_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, Xp, Nin, N, total, mask);
    GF_LAUNCH_CHECK("pack_panels_kernel");
    return GF_OK;
}

// does a per-hop launch for nPanels panels run the double-buffered kernel?  (the launcher and gf_khop_panel_uses_chain share the rule)
bool gf_panel_db_applies(const gf_plan* plan, int op, int nPanels) {
    const gf_csr_dev& m = plan->mat[op];
    const size_t lds_db = 4 * (size_t)(plan->n + 1) * 16 + 16;
    const bool one = lds_db > 80 * 1024;
    return g_tune.panel_db != 0 && (one || g_tune.panel_db == 2) && lds_db <= 160 * 1024 && nPanels >= 2 * num_cus() && m.pn_slices >= 2 &&
           g_tune.panel_np == 0;
}

int gf_spmm_panel_launch(const gf_plan* plan, int op, const float* Xin, float* Xout, int nPanels, hipStream_t st) {
    const gf_csr_dev& m = plan->mat[op];
    const int N = plan->n;
    GF_REQUIRE_ARG(m.pn_slices > 0, "gf_spmm_hop_panel: the plan has no panel image (N = %d > %d?)", N, kPanelMaxNodes);
    // NP panels per pass while the LDS still takes two workgroups per CU (their load and compute phases then overlap);
    // knob panel_np forces 1 / 2
    // (measured, tools/gpu_round25.sh: value-free stream +2-4 % for N <= 2559, mixed above; weighted stream +13-15 % up to N = 5119,
    // where the pair fills the LDS and only one workgroup fits a CU)
    const bool uniform = m.pn_uniform && g_tune.panel_uniform;
    const size_t pair = 2 * (size_t)(N + 1) * 16;
    int np = (pair <= 80 * 1024 || (!uniform && pair <= 160 * 1024)) ? 2 : 1;
    if (g_tune.panel_np == 1 || (g_tune.panel_np == 2 && pair <= 160 * 1024)) np = g_tune.panel_np;
    if (g_tune.panel_np == 4 && 2 * pair <= 160 * 1024) np = 4;   // experiments: four panels per pass (one workgroup per CU from N = 1279 on)
    if (nPanels < np * num_cus()) np = 1;  // not enough panels to keep every CU busy with pairs
    const size_t lds = (size_t)np * (N + 1) * 16;
    const int threads = N > 5120 ? 1024 : ((N > 2560 || (np >= 2 && N > 1280)) ? 512 : 256);
    const int thr = (lds > 80 * 1024) ? 1024 : threads;  // one workgroup per CU: give it all 16 waves
    int wgPerCU = (int)((160 * 1024) / (lds < 1024 ? 1024 : lds));
    const int waveCap = 32 / (thr / 64);
    if (wgPerCU > waveCap) wgPerCU = waveCap;
    if (wgPerCU < 1) wgPerCU = 1;
    const int nGroups = (nPanels + np - 1) / np;
    int64_t grid = (int64_t)num_cus() * wgPerCU;
    int split = 1;
    if (grid > nGroups) {
        // fewer passes than workgroup slots: several workgroups per pass, each taking every split-th slice (knob panel_split: 1 = off)
        const int waves = thr / 64;
        // (measured, tools/gpu_split.sh: 8-32 passes 1.1-2.1x faster, from ~128 passes on the duplicated panel loads cost more than the
        // idle CUs)
        split = nGroups * 4 <= num_cus() ? (int)(grid / nGroups) : 1;
        while (split > 1 && (split * waves > m.pn_slices || split > 8)) --split;
        if (g_tune.panel_split > 0 && split > g_tune.panel_split) split = g_tune.panel_split;
        grid = (int64_t)nGroups * split;
    }
    if (g_tune.panel_grid > 0 && grid > g_tune.panel_grid) grid = g_tune.panel_grid / split * split;  // experiments: fewer workgroups than CUs
    // double-buffered variant (spmm_panel_db_kernel): two buffers of two panels in one workgroup
    const size_t lds_db = 4 * (size_t)(N + 1) * 16 + 16;
    // measured (tools/panel_w_probe.py, profiles/r04_f_panel_db): N = 1682 weighted 64.2 -> 60.2 us per hop at width 64 (equal weights
    // 59.8 -> 58.6), N = 2500 47.6 -> 43.0 (44.5 -> 41.7) at width 32; N = 1000 (two workgroups per CU either way) 20.6 vs 20.8: the
    // default takes it where one workgroup fills the CU (1280 <= N <= 2559); knob panel_db: 0 = never, 2 = wherever four panels fit
    const bool one = lds_db > 80 * 1024;   // one workgroup per CU: all 16 waves
    if (gf_panel_db_applies(plan, op, nPanels)) {
        const int thr_db = g_tune.panel_thr > 0 ? g_tune.panel_thr : (one ? 1024 : 512);
        // (loader waves of 16: on a tap stack -- every hop reads what the last one wrote, nothing comes from the Infinity Cache -- 2: 75.4,
        //  3: 74.1, 4: 71.0 us per hop at N = 1682, width 64, weighted; the per-hop kernel 79.4; tools/panel_w_probe.py with PROBE_KHOP=1)
        const int loaders = g_tune.panel_loaders > 0 ? g_tune.panel_loaders : (thr_db >= 1024 ? 4 : (thr_db >= 512 ? 2 : 1));
        const int nG = (nPanels + 1) / 2;
        int64_t gdb = (int64_t)num_cus() * (one ? 1 : 2);
        if (gdb > nG) gdb = nG;
        auto kdb = uniform ? spmm_panel_db_kernel<1, 2> : spmm_panel_db_kernel<0, 2>;
        if (lds_db > 64 * 1024) GF_HIP(gf_grant_lds((const void*)kdb, lds_db));
        if (const int rc = gf_require_no_static_lds((const void*)kdb, "spmm_panel_db_kernel")) return rc;
        hipLaunchKernelGGL(kdb, dim3((unsigned)gdb), dim3(thr_db), lds_db, st, m.pn_slice, m.pn_oct, (const uint2*)m.pn_col2, m.pn_val4, m.pn_uval,
                           Xin, Xout, N, m.pn_slices, nPanels, m.pn_sentinel, g_tune.spmm_store, m.pn_ushift, loaders);
        GF_LAUNCH_CHECK("spmm_panel_db_kernel");
        return GF_OK;
    }
    typedef void (*kern_t)(const int2*, const int32_t*, const void*, const float4*, float, const float*, float*, int, int, int, int, int,
                           int, int, int);
    kern_t kern = np == 4 ? (uniform ? (kern_t)spmm_panel_kernel<1, 4> : (kern_t)spmm_panel_kernel<0, 4>)
                  : np == 2 ? (uniform ? (kern_t)spmm_panel_kernel<1, 2> : (kern_t)spmm_panel_kernel<0, 2>)
                            : (uniform ? (kern_t)spmm_panel_kernel<1, 1> : (kern_t)spmm_panel_kernel<0, 1>);
    if (lds > 64 * 1024) GF_HIP(gf_grant_lds((const void*)kern, lds));
    if (const int rc = gf_require_no_static_lds((const void*)kern, "spmm_panel_kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(thr), lds, st, m.pn_slice, m.pn_oct, uniform ? (const void*)m.pn_col4 : (const void*)m.pn_col2, m.pn_val4, m.pn_uval, Xin,
                       Xout, N, m.pn_slices, nPanels, m.pn_sentinel, g_tune.spmm_store, m.pn_ushift, g_tune.panel_rotate, split);
    GF_LAUNCH_CHECK("spmm_panel_kernel");
    return GF_OK;
}

extern "C" int gf_pack_panels(const float* x, float* Xp, int32_t B, int32_t C, int32_t Nin, int32_t N, void* stream) {
    GF_REQUIRE_ARG(x && Xp, "gf_pack_panels: NULL tensor");
    GF_REQUIRE_SHAPE(B > 0 && C > 0 && C % 4 == 0 && Nin > 0 && N >= Nin, "gf_pack_panels: bad shape B=%d C=%d (C %% 4 == 0) Nin=%d N=%d",
                     B, C, Nin, N);
    return gf_pack_panels_launch(x, Xp, B, C, Nin, N, gf_stream(stream), nullptr);
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
