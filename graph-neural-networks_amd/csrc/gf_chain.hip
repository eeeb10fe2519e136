// gf_chain.hip -- the K-1 hops of the tap stack, one column panel (or two) at a time, WITHOUT the panel leaving LDS between hops.
// Same operation as gf_panel.hip / gf_spmm.hip (reference graphML.py:158-161: `x = torch.matmul(x, S)` per tap, the taps
// concatenated at :161); same column-panel layout Xp[P][N][4] (gf_panel.hip).
//
// What changes against the per-hop panel kernel:
//   * A workgroup loads a panel ONCE (N*16 bytes, coalesced, LDS-DMA), runs hop 1 .. K-1 on it inside LDS and stores every tap
//     with coalesced full-line stores.  HBM traffic per chain = (1 + (K-1)) * panel instead of 2 * (K-1) * panel: the K-2
//     re-reads of the algorithmic count (SURVEY.md 8d counts a read and a write per hop) never happen.
//   * A hop's outputs stay in registers (4 floats per row per lane, kChainSets / NP rows per lane) until every wave has finished
//     gathering from the panel; then they are written over the panel in natural row order and the panel is the next hop's
//     source.  Because outputs pass through registers, the rows a wave computes together need not be neighbours in memory:
//     the plan sorts rows by degree (gf_plan.hip, upload_chain) and lane = row lockstep wastes ~10-15 % of the gathers on
//     padding instead of the 39 % of natural-order octets.
//   * Roles: waves 0 .. Wc-1 gather; the last one or two waves are storers: during hop h+1 they stream the panel (= tap h, which
//     the gatherers are reading) to HBM.  The gatherers never have a store in flight (loads and stores share vmcnt and complete
//     out of order with each other -- a wave with both pending can only wait for vmcnt(0), i.e. for the stores), and the tap stores
//     (160 KB per hop for a full-LDS panel) run under the gathers instead of between them.  One wave moves ~20 GB/s of stores
//     whatever the rest of the chip does (tools/hbm_ceiling.hip), so a full-LDS workgroup has two.  History (profiles/r02_a_chain):
//     every wave storing its share after the rewrite 250 us per hop at config 2 (the per-hop kernel: 150 us); one storer 131 us
//     (it needs 10.3 us per hop, the gatherers 6.7 us); two storers 122-127 us (raising their priority with s_setprio: no change).
//   * NP = 2 panels side by side in LDS when two fit (N <= 5119): one entry word drives two gathers, the (column, value) stream --
//     which, for graphs this small, is several times the panel itself -- is read once per pair.
//   * per hop two workgroup barriers (gathers done | panel rewritten), LDS-only: global stores stay in flight across them.
//     After the last hop every wave takes part in one sweep that stores the last tap and requests the next panel chunk by chunk
//     (a wave overwrites only what it has just read back).
// What bounds it at config 2: HBM writes.  A chain moves 1 read + 4 writes of 328 MB; the chip sustains ~4.9 TB/s of writes
// (tools/hbm_ceiling.hip), i.e. ~290 us for the 1.64 GB against 480-510 us measured: the CUs store ~60 % of the time (nothing to
// store during hop 0 and the panel load; the LDS is full, so the next panel cannot be fetched early).  Inside a hop the slowest of the
// 14 gather waves finishes ~40 % after the fastest: two SIMDs carry 4 gather waves, two carry 3 + a storer, and a SIMD's issue
// slots go to its oldest wave first (per-wave trace; static s_setprio by wave age only inverts the order, the spread stays; the
// balanced split -- 12 gather waves + 4 storers, one per SIMD, 14 sets per lane -- is slower: 490 vs 475 us per launch).
// Determinism: each row's sum runs in the plan's fixed neighbour order in one lane; no atomics.
#include "gf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr unsigned kNoRow = 0xffffffffu;

#ifdef GF_CHAIN_TRACE  // experiment builds only (make variant EXTRA=-DGF_CHAIN_TRACE): per-phase s_memtime stamps of the first workgroups
__device__ unsigned long long* g_chain_trace = nullptr;
#define GF_STAMP(ev)                                                                                                  \
    do {                                                                                                              \
        if (g_chain_trace && blockIdx.x < 8 && p == (int)blockIdx.x + (int)gridDim.x && lane == 0 && (wave == 0 || !gatherer)) \
            g_chain_trace[((blockIdx.x * 8 + h) * 2 + (gatherer ? 0 : 1)) * 8 + (ev)] = __builtin_amdgcn_s_memtime();     \
    } while (0)
#else
#define GF_STAMP(ev) do { } while (0)
#endif
constexpr int kWaitVm0 = 0x0F70;  // s_waitcnt vmcnt(0) (expcnt / lgkmcnt fields at their maxima) in the gfx9 encoding

#ifndef GF_CHAIN_MAXT
#define GF_CHAIN_MAXT 1024
#endif
template <int UNIFORM, int NP>
__global__ __launch_bounds__(GF_CHAIN_MAXT) void spmm_chain_kernel(const int32_t* __restrict__ gtab, const uint32_t* __restrict__ rowoff,
                                                          const void* __restrict__ cols, const float4* __restrict__ vals, float uval,
                                                          const float* __restrict__ Xin, float* __restrict__ Xout, int N, int nPanels,
                                                          int R, int nHops, int64_t tapStride, int store_mode, int nStorers) {
    constexpr int kSets = (kChainSets / NP) & ~1;   // row sets per lane: the accumulators of a hop are kSets x NP x 4 registers
    extern __shared__ __attribute__((aligned(16))) float4 panel[];  // NP regions of [N + 1]: a panel + one zero slot each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nW = (int)(blockDim.x >> 6);          // waves of the workgroup
    const int Wc = nW - nStorers;                   // gatherers; the last nStorers waves store
    const int Tc = Wc * 64;
    const bool gatherer = wave < Wc;
    f32x4* lds4 = reinterpret_cast<f32x4*>(panel);
    // gathers address LDS absolutely (see gf_panel.hip): the dynamic panel is the only LDS object, so it starts at address 0 (the launcher
    // asks the runtime that the kernel holds no static LDS: gf_require_no_static_lds)
    const unsigned regionB = (unsigned)(N + 1) * 16u;  // bytes between the two panels of a pass
    const int region4 = N + 1;                         // ... in float4
    const u32x4* col4 = reinterpret_cast<const u32x4*>(cols) + lane;  // [word][lane]: 8 x 16-bit columns = two group-rows
    const f32x4* val4 = reinterpret_cast<const f32x4*>(vals) + lane;

    // the row this lane computes in set r, two 16-bit row indices per register (0xffff = none); kept in registers: re-deriving them
    // per hop made the compiler hoist a dozen 64-bit addresses out of the loops and spill them
    unsigned ro2[kSets / 2];
#pragma unroll
    for (int r = 0; r < kSets; r += 2) {
        const unsigned lo = (gatherer && r < R) ? rowoff[(int64_t)r * Tc + tid] : kNoRow;
        const unsigned hi = (gatherer && r + 1 < R) ? rowoff[(int64_t)(r + 1) * Tc + tid] : kNoRow;
        ro2[r / 2] = (lo == kNoRow ? 0xffffu : lo >> 4) | ((hi == kNoRow ? 0xffffu : hi >> 4) << 16);
    }
    static_assert(kSets % 2 == 0, "row indices are packed in pairs");
    // lane i < 16 holds entry i of this wave's table (0: first word, 1 + r: end of block r), lane 16 + r: block r has an odd number of group-rows
    const int gv = gatherer ? gtab[wave * 32 + (lane & 31)] : 0;
    const int gbeg = __builtin_amdgcn_readlane(gv, 0);
    if (tid < NP) lds4[tid * region4 + N] = (f32x4){0.f, 0.f, 0.f, 0.f};   // the zero slots empty ELL slots gather from
    const int nChunks = (N + 63) >> 6;                     // 64-row chunks = 1 KiB of a panel
    const int64_t pstride4 = (int64_t)N;                   // float4 per panel
    const int nPasses = (nPanels + NP - 1) / NP;           // a pass = NP panels (the last one may hold fewer)

    // Panel loads go through LDS-DMA (global_load_lds_dwordx4): a wave's 64 rows land as 1 KiB at a wave-uniform LDS base
    // (+ lane * 16) without passing through registers.
    auto dma_chunk = [&](const f32x4* src, int k, int c) {  // rows 64c .. 64c+63 of the panel at src into region k
        const int row0 = c * 64;                            // wave-uniform
        if (row0 + lane < N)
            __builtin_amdgcn_global_load_lds(src + row0 + lane, (lds_void*)(uintptr_t)((unsigned)(k * region4 + row0) * 16u), 16, 0, 0);
    };
    auto store4 = [&](const f32x4& v, f32x4* dst) {
        if (store_mode == 2)
            __builtin_nontemporal_store(v, dst);
        else
            *dst = v;
    };
    // Barrier for the LDS hand-offs inside a pass: LDS operations complete, global stores stay in flight (__syncthreads() would
    // wait for vmcnt(0): the storers would sit out the HBM latency of their last stores at every barrier; nobody in this kernel
    // reads what they store).
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    int p = (int)blockIdx.x;   // pass
    if (p >= nPasses) return;
#pragma unroll
    for (int k = 0; k < NP; ++k)
        if (p * NP + k < nPanels) {
            const f32x4* src = reinterpret_cast<const f32x4*>(Xin) + (int64_t)(p * NP + k) * pstride4;
            for (int c = wave; c < nChunks; c += nW) dma_chunk(src, k, c);
        }
    for (;;) {
        __builtin_amdgcn_s_waitcnt(kWaitVm0);  // this wave's share of the pass has landed (and its stores of the last sweep are out)
        __syncthreads();                       // ... and everybody else's
        const int pn = p + (int)gridDim.x;
        const int nvalid = min(NP, nPanels - p * NP);
        f32x4* outp = reinterpret_cast<f32x4*>(Xout) + (int64_t)p * NP * pstride4;
        for (int h = 0; h < nHops; ++h) {
            f32x4 acc[kSets][NP];
            GF_STAMP(0);
            if (!gatherer) {
                // storer: the panels hold tap h (h >= 1: the output of the previous hop; tap 0 is the caller's) -- stream them out
                // while the gatherers read them.  The LDS reads run one round (kSt chunks) ahead of the stores in a second
                // register set (an LDS read takes several hundred cycles to come back while the other waves gather).
                if (h > 0) {
                    constexpr int kSt = 4;
                    const int sw = wave - Wc;                        // which storer this is
                    const int cPer = (nChunks + nStorers - 1) / nStorers;
                    const int cLo = sw * cPer, cHi = min(nChunks, cLo + cPer);  // its contiguous share of the chunks
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        if (k >= nvalid) break;
                        f32x4* out = outp + (int64_t)(h - 1) * (tapStride / 4) + (int64_t)k * pstride4;
                        const f32x4* reg = lds4 + k * region4;
                        f32x4 v0[kSt], v1[kSt];
                        auto rd = [&](f32x4 (&v)[kSt], int c0) {
#pragma unroll
                            for (int u = 0; u < kSt; ++u) v[u] = reg[min((c0 + u) * 64 + lane, N)];  // N = the zero slot (reads past the share are dropped)
                        };
                        auto wr = [&](const f32x4 (&v)[kSt], int c0) {
#pragma unroll
                            for (int u = 0; u < kSt; ++u) {
                                const int idx = (c0 + u) * 64 + lane;
                                if (c0 + u < cHi && idx < N) store4(v[u], out + idx);
                            }
                        };
                        rd(v0, cLo);
                        for (int c0 = cLo; c0 < cHi; c0 += 2 * kSt) {
                            rd(v1, c0 + kSt);
                            wr(v0, c0);
                            rd(v0, c0 + 2 * kSt);
                            wr(v1, c0 + kSt);
                        }
                    }
                }
            } else {
                // The wave's entry stream, block by block (block r = the 64 rows of set r).  One 16-byte word per lane = the columns
                // of TWO group-rows (8 neighbours, 16 bits each; the weighted stream adds 2 x 16 bytes of values).  Two words are in
                // flight in two register sets (A, B) that alternate per word, also across block boundaries: a block starts on A or
                // on B (`par`, wave-uniform) and exists in both variants; each block is its own loop and assigns its set's
                // accumulators once, outside any loop.  Blocks are padded to an even number of group-rows in storage only: the odd
                // half is skipped, not gathered.
                // (Measured alternatives: sums committed through a switch inside one stream loop -- the register allocator shuffles
                // all accumulator registers on every iteration; a static goto state machine over the blocks -- 440 bytes of scratch
                // per lane; register banks holding a whole block, prefetched one block ahead with exact vmcnt waits -- 2-4 narrower
                // loads per block, slower; the gather phase is LDS-bound at ~6.3 cycles per ds_read_b128: its time follows the
                // degree with exactly that slope.)
                __builtin_amdgcn_s_waitcnt(kWaitVm0);  // nothing of this wave is pending here (tells the waitcnt pass so)
                int u = gbeg;   // word index in the stream
                int par = 0;
                u32x4 cA = col4[(int64_t)u * 64], cB = col4[(int64_t)(u + 1) * 64];
                f32x4 vA[UNIFORM ? 1 : 2], vB[UNIFORM ? 1 : 2];
                if (!UNIFORM) {
                    vA[0] = val4[(int64_t)(2 * u) * 64], vA[1] = val4[(int64_t)(2 * u + 1) * 64];
                    vB[0] = val4[(int64_t)(2 * u + 2) * 64], vB[1] = val4[(int64_t)(2 * u + 3) * 64];
                }
                f32x4 a0[NP];   // one running sum per panel (a second, interleaved one costs 4 registers per panel the kernel does not have:
                                // 131 VGPRs against the 128 of a 1024-thread workgroup; the phase is LDS-bound, not add-latency-bound)
                auto gather4 = [&](unsigned lo, unsigned hi, const f32x4& w) {
                    const unsigned o0 = (lo & 0xffffu) << 4, o1 = (lo >> 16) << 4, o2 = (hi & 0xffffu) << 4, o3 = (hi >> 16) << 4;
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const unsigned rb = k * regionB;
                        const f32x4 x0 = *reinterpret_cast<lds_f32x4*>(o0 + rb);
                        const f32x4 x1 = *reinterpret_cast<lds_f32x4*>(o1 + rb);
                        const f32x4 x2 = *reinterpret_cast<lds_f32x4*>(o2 + rb);
                        const f32x4 x3 = *reinterpret_cast<lds_f32x4*>(o3 + rb);
                        if (UNIFORM) {
                            a0[k] += x0;
                            a0[k] += x1;
                            a0[k] += x2;
                            a0[k] += x3;
                        } else {
                            a0[k] += w.x * x0;
                            a0[k] += w.y * x1;
                            a0[k] += w.z * x2;
                            a0[k] += w.w * x3;
                        }
                    }
                };
                // one word: its first group-row always exists, its second unless the block has an odd count and this is its last word
                auto step = [&](u32x4& cc, f32x4 (&vv)[UNIFORM ? 1 : 2], bool both) {
                    const u32x4 c = cc;
                    f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
                    if (!UNIFORM) w0 = vv[0], w1 = vv[1];
                    gather4(c.x, c.y, w0);
                    cc = col4[(int64_t)(u + 2) * 64];  // refill this register set (the stream ends with two sentinel words)
                    if (!UNIFORM) vv[0] = val4[(int64_t)(2 * u + 4) * 64], vv[1] = val4[(int64_t)(2 * u + 5) * 64];
                    if (both) gather4(c.z, c.w, w1);
                    ++u;
                };
                auto block = [&](int ue, int odd, f32x4 (&out)[NP]) {  // words [u, ue) of the stream, ue > u; odd: the last word is half empty
#pragma unroll
                    for (int k = 0; k < NP; ++k) a0[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (par == 0) {
                        for (;;) {
                            step(cA, vA, !(odd && u + 1 == ue));
                            if (u == ue) { par = 1; break; }
                            step(cB, vB, !(odd && u + 1 == ue));
                            if (u == ue) { par = 0; break; }
                        }
                    } else {
                        for (;;) {
                            step(cB, vB, !(odd && u + 1 == ue));
                            if (u == ue) { par = 0; break; }
                            step(cA, vA, !(odd && u + 1 == ue));
                            if (u == ue) { par = 1; break; }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NP; ++k) out[k] = a0[k];
                };
#pragma unroll
                for (int r = 0; r < kSets; ++r) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) acc[r][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (r < R) block(__builtin_amdgcn_readlane(gv, r + 1), __builtin_amdgcn_readlane(gv, 16 + r), acc[r]);
                }
            }
            GF_STAMP(1);
            lds_barrier();  // every gatherer has finished reading the panels, the storers have read tap h out of them
            GF_STAMP(2);
            if (gatherer) {
#pragma unroll
                for (int r = 0; r < kSets; ++r) {
                    const unsigned row = (r & 1) ? ro2[r / 2] >> 16 : ro2[r / 2] & 0xffffu;
                    if (row != 0xffffu) {
#pragma unroll
                        for (int k = 0; k < NP; ++k) {
                            f32x4 v = acc[r][k];
                            if (UNIFORM) v *= uval;
                            *reinterpret_cast<lds_f32x4*>((row << 4) + k * regionB) = v;
                        }
                    }
                }
            }
            GF_STAMP(3);
            lds_barrier();  // the panels now hold tap h + 1
            GF_STAMP(4);
        }
        // last tap: every wave stores its chunks and, right behind each, requests the same chunk of the next pass
        {
            const bool next = pn < nPasses;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const bool have = k < nvalid;                                  // this pass has a panel k (to store)
                const bool want = next && pn * NP + k < nPanels;               // the next pass has one (to load)
                f32x4* out = outp + (int64_t)(nHops - 1) * (tapStride / 4) + (int64_t)k * pstride4;
                const f32x4* nsrc = reinterpret_cast<const f32x4*>(Xin) + (int64_t)(pn * NP + k) * pstride4;
                const f32x4* reg = lds4 + k * region4;
                for (int j0 = 0; wave + j0 * nW < nChunks; j0 += 4) {  // 4 of this wave's chunks per round (all reads, then store + DMA each)
                    f32x4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = reg[min((wave + (j0 + q) * nW) * 64 + lane, N)];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the chunks are in registers before the DMA may overwrite them
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = wave + (j0 + q) * nW;
                        const int idx = c * 64 + lane;
                        if (have && c < nChunks && idx < N) store4(v[q], out + idx);
                        if (want && c < nChunks) dma_chunk(nsrc, k, c);
                    }
                }
            }
        }
        p = pn;
        if (p >= nPasses) break;
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

#ifdef GF_CHAIN_TRACE
extern "C" int gf_chain_trace_set(unsigned long long* buf) {
    GF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), &buf, sizeof(buf)));
    return GF_OK;
}
#endif

// nHops successive hops of every panel: tap h + 1 = op(S) tap h, tap 0 = Xin, tap h + 1 at Xout + h * tapStride.
int gf_spmm_chain_launch(const gf_plan* plan, int op, const float* Xin, float* Xout, int nPanels, int nHops, int64_t tapStride,
                         hipStream_t st) {
    const gf_csr_dev& m = plan->mat[op];
    const int N = plan->n;
    GF_REQUIRE_ARG(m.cn_waves > 0, "gf_khop_panel: the plan has no chain image (N = %d > %d?)", N, kPanelMaxNodes);
    GF_REQUIRE_ARG(tapStride % 4 == 0, "gf_khop_panel: tap stride %lld is not a multiple of 4 floats", (long long)tapStride);
    const bool uniform = m.pn_uniform && g_tune.panel_uniform;
    const int np = m.cn_np;
    const size_t lds = (size_t)np * (N + 1) * 16;
    // gatherers + storers: one wave moves ~20 GB/s of stores whatever the rest of the chip does (tools/hbm_ceiling.hip), two cover the
    // 160 KB per hop of a full-LDS workgroup within the gather time
    const int storers = m.cn_waves >= kChainBigW ? kChainStorers : 1;
    const int waves = m.cn_waves + storers;
    const int thr = waves * 64;
    int wgPerCU = (int)((160 * 1024) / (lds < 1024 ? 1024 : lds));
    const int waveCap = 32 / waves;
    if (wgPerCU > waveCap) wgPerCU = waveCap;
    if (wgPerCU > 8) wgPerCU = 8;
    if (wgPerCU < 1) wgPerCU = 1;
    const int nPasses = (nPanels + np - 1) / np;
    int64_t grid = (int64_t)chain_num_cus() * wgPerCU;
    if (grid > nPasses) grid = nPasses;
    typedef void (*kern_t)(const int32_t*, const uint32_t*, const void*, const float4*, float, const float*, float*, int, int, int, int,
                           int64_t, int, int);
    const kern_t kern = np == 2 ? (uniform ? (kern_t)spmm_chain_kernel<1, 2> : (kern_t)spmm_chain_kernel<0, 2>)
                                : (uniform ? (kern_t)spmm_chain_kernel<1, 1> : (kern_t)spmm_chain_kernel<0, 1>);
    if (lds > 64 * 1024) GF_HIP(gf_grant_lds((const void*)kern, lds));
    if (const int rc = gf_require_no_static_lds((const void*)kern, "spmm_chain_kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(thr), lds, st, m.cn_gtab, m.cn_rowoff, (const void*)m.cn_col8, m.cn_val4, m.pn_uval,
                       Xin, Xout, N, nPanels, m.cn_sets, nHops, tapStride, g_tune.spmm_store, storers);
    GF_LAUNCH_CHECK("spmm_chain_kernel");
    return GF_OK;
}
