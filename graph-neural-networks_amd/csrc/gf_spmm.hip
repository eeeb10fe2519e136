// gf_spmm.hip -- the K-hop GSO-signal product: X_k = op(S) X_{k-1} on node-major signals.
// Replaces the K-1 dense broadcast GEMMs `x = torch.matmul(x, S)` of reference graphML.py:158-161
// (O(B*G*N^2) each) with sparse products (O(B*G*nnz)), and the growing torch.cat (graphML.py:161) with in-place
// writes into tap slot k of the stack Z[T,B,N,G].
//
// HBM-bound by the algorithmic count:  bytes/hop = 2*B*N*W*4 + nnz*8 + (N+1)*4.
// What the hardware actually has to serve is the gather stream nnz*B*W*4 (each signal row is re-read once per
// neighbour, ~10x the algorithmic read); it can only come out of cache, so the design goal is: keep the gather
// panel L2-resident and keep the vector-memory pipe (64 B/clk/CU) saturated with 128-byte line gathers.
//
// Kernel A (default) -- SELL-8, wave-autonomous, dispatch-ordered:
//   * the plan stores the degree-sorted rows as SELL-8 slices (8 rows padded to the slice's longest row, entries
//     k-major).  One wavefront owns one slice at a time: lane = row*8 + sub-lane, 8 sub-lanes x float4 = one 128-byte
//     line per row per neighbour (W = 32); W = 64..256 -> several float4 per lane, W < 32 -> spare sub-lanes take more
//     batch entries.  All 8 rows of a slice walk the same (wave-uniform) number of neighbours: no divergence, and all
//     of a slice's gathers (up to 16 per lane) are issued before the first FMA -- one memory round trip per slice.
//   * the slice's (col, val) stream is staged through a per-wave LDS ring (coalesced 512-byte reads, then broadcast
//     ds_read_b64 per neighbour): index traffic stays off the vector-memory return path that the gathers need, and the
//     next slice's entries are fetched while the current slice is gathered.  No workgroup barrier exists at all.
//   * blockIdx -> (XCD role, batch tile, slice group) follows the hardware dispatch order: workgroup L runs on XCD L%8
//     (observed; a wrong guess costs speed, never correctness), batch tile t is pinned to XCD t%8 and the dispatcher
//     hands an XCD its tiles strictly in order, so at most ~2 gather panels (N*W*4*BT bytes each) are live in that
//     XCD's 4 MiB L2.  (A persistent variant with statically strided waves was measured and rejected: waves drift apart
//     by many tiles, 30 % L2 hits vs 76 % -- profiles/r01_c_l2_residency.)  Output rows are stored write-through (sc1)
//     so they do not evict the panel.
//   * each (row, batch entry) sum runs over the row's entries in ascending column order in one lane group: fixed
//     summation order, no atomics -> bitwise deterministic.
// Kernel B -- CSR, one workgroup per 256/LG rows with its segment staged in LDS (the first version; kept for A/B).
// Kernel C -- generic: any W (G = 1, odd widths), one thread per output element.
#include <stdlib.h>
#include <type_traits>

#include "gf_common.h"
#include <utility>

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 2048;  // kernel B: CSR entries staged per pass (16 KB of LDS)
constexpr int kCK = 32;       // kernel A: neighbours (k) per LDS ring slot -> 32*8 entries * 8 B = 2 KB per slot

__device__ __forceinline__ void fma4(float4& a, float s, const float4& x) {
    a.x = fmaf(s, x.x, a.x);
    a.y = fmaf(s, x.y, a.y);
    a.z = fmaf(s, x.z, a.z);
    a.w = fmaf(s, x.w, a.w);
}

// Output rows are written once and never re-read by this kernel; a plain store leaves the line in the XCD's L2
// where it competes with the gather panel for the 4 MiB.  mode 1 = write-through (sc1: the line leaves L2),
// mode 2 = non-temporal hint, mode 0 = plain.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_row(float* p, const float4& v, int mode) {
    if (mode == 3) {   // experiments only (gf_tune spmm_store=3): NO store unless the value is an impossible one -- an upper bound on what
        if (v.x == 1.2345e30f) *reinterpret_cast<float4*>(p) = v;   // the output stores cost; results are wrong
        return;
    }
    if (mode == 1) {
        const f32x4 d = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    } else if (mode == 2) {
        __builtin_nontemporal_store(v.x, p);
        __builtin_nontemporal_store(v.y, p + 1);
        __builtin_nontemporal_store(v.z, p + 2);
        __builtin_nontemporal_store(v.w, p + 3);
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}

// gather load of one float4: NTL = 1 adds the non-temporal hint (the line is not kept in the CU's L1 -- a gathered
// line is never reused by the same CU, so caching it only costs L1 fill bandwidth).
template <int NTL>
__device__ __forceinline__ float4 load_row(const float* p) {
    if (NTL) {
        const f32x4 d = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return make_float4(d.x, d.y, d.z, d.w);
    }
    return *reinterpret_cast<const float4*>(p);
}

// ------------------------------------------------------------------------------------------------------------------
// Kernel A
//   LPR = sub-lanes per row (1, 2, 4, 8), VPL = float4 per lane per neighbour, W = 4 * LPR * VPL
//   BL  = 8 / LPR batch entries side by side in one wave, BT = batch entries per lane (register tile)
//   SPW = consecutive slices one wave walks (the second slice's entries are fetched while the first is gathered)
// ------------------------------------------------------------------------------------------------------------------
//   UNI = 1: every stored value equals uval (S = A / lambda_max of an unweighted graph): the entry stream is the columns only (4 bytes
//         instead of 8 per entry: at config 4 it is re-read for each of the 128 batch entries, 1 GB per hop through the same fabric
//         and the same L2 as the gathers), padding = column -1, rows are summed and scaled once (v * sum(x), as the panel kernels do)
template <int LPR, int VPL, int BT, int SPW, int NTL, int UCAP, int UNI>
__global__ __launch_bounds__(kThreads) void spmm_sell_kernel(const int32_t* __restrict__ kptr, const void* __restrict__ ent_v,
                                                             const int32_t* __restrict__ rowid, const float* __restrict__ Xin,
                                                             float* __restrict__ Xout, int N, int B, int nSlices, int nBTiles,
                                                             int lanes, int parts, int blocksPerTile, int store_mode, int pf_blocks,
                                                             float uval) {
    typedef typename std::conditional<UNI != 0, int32_t, int2>::type ent_t;
    const ent_t* __restrict__ ent = static_cast<const ent_t*>(ent_v);
    constexpr int BL = 8 / LPR;
    constexpr int W = 4 * LPR * VPL;
    constexpr int BTW = BL * BT;  // batch entries per tile
    constexpr int UMAX = (UCAP / (BT * VPL)) > 0 ? (UCAP / (BT * VPL)) : 1;  // neighbours whose gathers are in flight together
    __shared__ ent_t s_ent[kThreads / 64][2][kCK * 8];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane >> 3, sub = lane & 7;
    const int bl = sub / LPR, li = sub - bl * LPR;

    // ---- work assignment, in HARDWARE DISPATCH ORDER -----------------------------------------------------------------
    // workgroup L runs on XCD L % 8 (observed; a wrong guess costs speed only).  XCD x serves batch-tile lane x % lanes
    // (tiles lane, lane + lanes, ... in dispatch order, so an XCD has at most ~2 gather panels live in its L2 at a time)
    // and slice partition x / lanes of `parts`.  lanes * parts <= 8.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lane_t = xcd % lanes, part = xcd / lanes;
    if (part >= parts) return;  // all exits are wave-uniform; this kernel has no workgroup barrier
    const int tl = slot / blocksPerTile;
    int blk = slot - tl * blocksPerTile;
    const int tile = lane_t + tl * lanes;
    if (tile >= nBTiles) return;
    if (blk < pf_blocks) {
        // ---- prefetch role: the first pf_blocks workgroups of every tile stream the NEXT tile's gather panel into this
        // XCD's L2 with sequential full-line reads, so the gathers of that tile hit L2 instead of waiting out HBM latency
        // on a demand miss (every vmcnt(0) batch of ~100 random lines contained at least one).  Part 0 only.
        const int ntile = tile + lanes;
        if (part != 0 || ntile >= nBTiles) return;
        const int nb0 = ntile * BTW;
        const int nb = min(BTW, B - nb0);
        const int64_t n4 = (int64_t)nb * N * W / 4;  // float4 count of the panel (batch entries of a tile are contiguous)
        const f32x4* src = reinterpret_cast<const f32x4*>(Xin + (int64_t)nb0 * N * W);
        f32x4 sink = {0.f, 0.f, 0.f, 0.f};
        const int64_t step = (int64_t)pf_blocks * kThreads;
        for (int64_t i = (int64_t)blk * kThreads + threadIdx.x; i < n4; i += step * 16) {
            f32x4 v[16];  // 16 independent 16-byte loads per lane in flight: 64 KB per workgroup per round
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int64_t j = i + (int64_t)u * step;
                v[u] = (j < n4) ? src[j] : sink;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) sink += v[u];
        }
        if (sink.x == 1.2345e30f && sink.y == -7.0f) Xout[0] = sink.z + sink.w;  // keeps the loads alive; never true in practice
        return;
    }
    blk -= pf_blocks;
    const int nSlP = (nSlices - part + parts - 1) / parts;  // slices of this partition: part, part + parts, ...
    const int i0 = (blk * (kThreads / 64) + wave) * SPW;
    if (i0 >= nSlP) return;
    const int ns = min(SPW, nSlP - i0);

    ent_t* ring0 = &s_ent[wave][0][0];
    ent_t* ring1 = &s_ent[wave][1][0];
    auto pad_entry = []() {
        if constexpr (UNI != 0) return (int32_t)-1;
        else return make_int2(0, 0);
    };
    auto col_of = [](const ent_t& e) {
        if constexpr (UNI != 0) return e;
        else return e.x;
    };

    // entries of k-range [ka, kb) (kb - ka <= kCK) -> 4 registers per lane, coalesced 512-byte reads
    ent_t pre[4];
    auto issue_entries = [&](int ka, int kb) {
        const int cnt = (kb - ka) * 8;
        const ent_t* src = ent + (int64_t)ka * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = j * 64 + lane;
            pre[j] = (e < cnt) ? src[e] : pad_entry();
        }
    };
    auto commit_entries = [&](ent_t* dst) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j * 64 + lane] = pre[j];
    };

    const int b0 = tile * BTW;
    const float* xb[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        const int b = min(b0 + t * BL + bl, B - 1);  // clamp loads of a ragged last tile; stores are masked
        xb[t] = Xin + (int64_t)b * N * W + li * 4;
    }

    int s = part + parts * i0;
    int k0 = kptr[s], k1 = kptr[s + 1];
    issue_entries(k0, min(k1, k0 + kCK));
    commit_entries(ring0);
    int cur = 0;

    for (int j = 0; j < ns; ++j) {
        const bool has_next = j + 1 < ns;
        int k0n = 0, k1n = 0;
        if (has_next) {
            k0n = kptr[s + parts];
            k1n = kptr[s + parts + 1];
        }
        const int orow = rowid[s * 8 + r];  // -1 past the last row; issued early, consumed at the store
        float4 acc[BT][VPL];
#pragma unroll
        for (int t = 0; t < BT; ++t)
#pragma unroll
            for (int v = 0; v < VPL; ++v) acc[t][v] = make_float4(0.f, 0.f, 0.f, 0.f);

        for (int kc = k0;;) {  // ring slots of this slice (one for ordinary rows, many for hub rows)
            const int kend = min(k1, kc + kCK);
            // -- prefetch the next ring slot: this slice's next chunk, or the next slice's first chunk
            bool fetch = true;
            if (kend < k1)
                issue_entries(kend, min(k1, kend + kCK));
            else if (has_next)
                issue_entries(k0n, min(k1n, k0n + kCK));
            else
                fetch = false;

            // -- gather + accumulate the nk neighbours staged in the current slot, UMAX neighbours at a time: all their
            //    gathers are issued before the first FMA (cnt is wave-uniform: scalar branches, exact counts, no padding)
            const ent_t* eb = (cur ? ring1 : ring0) + r;
            const int nk = kend - kc;
            for (int k = 0; k < nk; k += UMAX) {
                const int cnt = min(UMAX, nk - k);
                // all of the chunk's (col, val) pairs first -- unconditional, back-to-back broadcast reads and ONE lgkmcnt wait
                // (the slot is always fully written, zero-padded), then the gathers, each behind a scalar branch
                ent_t e[UMAX];
#pragma unroll
                for (int u = 0; u < UMAX; ++u) e[u] = eb[(k + u) * 8];
                float4 x[UMAX][BT][VPL];
#pragma unroll
                for (int u = 0; u < UMAX; ++u)
                    if (u < cnt) {
                        const unsigned off = __umul24((unsigned)max(col_of(e[u]), 0), (unsigned)W);
#pragma unroll
                        for (int t = 0; t < BT; ++t)
#pragma unroll
                            for (int v = 0; v < VPL; ++v)
                                x[u][t][v] = load_row<NTL>(xb[t] + off + v * (LPR * 4));
                    }
#pragma unroll
                for (int u = 0; u < UMAX; ++u)  // ascending k: fixed summation order
                    if (u < cnt) {
                        float val;
                        if constexpr (UNI != 0) val = e[u] >= 0 ? 1.f : 0.f;   // (padding: row 0 times zero, as the {0, 0.0f} entries)
                        else val = __int_as_float(e[u].y);
#pragma unroll
                        for (int t = 0; t < BT; ++t)
#pragma unroll
                            for (int v = 0; v < VPL; ++v) fma4(acc[t][v], val, x[u][t][v]);
                    }
            }

            if (fetch) commit_entries(cur ? ring0 : ring1);  // LDS ops of one wave execute in order: no barrier needed
            cur ^= 1;
            if (kend >= k1) break;
            kc = kend;
        }

        if (orow >= 0) {
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                const int b = b0 + t * BL + bl;
                if (b < B) {
                    float* o = Xout + (int64_t)b * N * W + (int64_t)orow * W + li * 4;
#pragma unroll
                    for (int v = 0; v < VPL; ++v) {
                        if constexpr (UNI != 0) acc[t][v].x *= uval, acc[t][v].y *= uval, acc[t][v].z *= uval, acc[t][v].w *= uval;
                        store_row(o + v * (LPR * 4), acc[t][v], store_mode);
                    }
                }
            }
        }
        s += parts;
        k0 = k0n;
        k1 = k1n;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Kernel B: CSR, workgroup-staged segment
// ------------------------------------------------------------------------------------------------------------------
template <int LG, int BT>
__global__ __launch_bounds__(kThreads) void spmm_hop_vec_kernel(const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ col,
                                                                const float* __restrict__ val,
                                                                const int32_t* __restrict__ rowid,
                                                                const float* __restrict__ Xin, float* __restrict__ Xout,
                                                                int N, int B, int nRowBlocks, int nBTiles, int xcd_map,
                                                                int store_mode) {
    constexpr int RPB = kThreads / LG;
    constexpr int W = LG * 4;
    __shared__ int32_t s_off[kChunk];  // neighbour row offset in floats (col * W)
    __shared__ float s_val[kChunk];

    int btile, rb;
    {
        const int L = blockIdx.x;
        if (xcd_map) {
            const int xcd = L & 7, slot = L >> 3;
            const int btl = slot / nRowBlocks;
            rb = slot - btl * nRowBlocks;
            btile = btl * 8 + xcd;
        } else {
            btile = L / nRowBlocks;
            rb = L - btile * nRowBlocks;
        }
    }
    if (btile >= nBTiles) return;  // uniform per block: no barrier is skipped by part of a block
    const int b0 = btile * BT;

    const int tid = threadIdx.x;
    const int grp = tid / LG, li = tid - grp * LG;
    const int p_lo = rb * RPB;
    const int p_hi = min(p_lo + RPB, N);
    const int p = p_lo + grp;
    int rs = 0, re = 0;
    if (p < N) {
        rs = rowptr[p];
        re = rowptr[p + 1];
    }
    const int seg_lo = rowptr[p_lo], seg_hi = rowptr[p_hi];

    const float* xb[BT];
    float4 acc[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        const int b = min(b0 + t, B - 1);
        xb[t] = Xin + (int64_t)b * N * W + li * 4;
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    for (int base = seg_lo; base < seg_hi; base += kChunk) {
        const int cnt = min(kChunk, seg_hi - base);
        if (base != seg_lo) __syncthreads();
        for (int i = tid; i < cnt; i += kThreads) {
            s_off[i] = col[base + i] * W;
            s_val[i] = val[base + i];
        }
        __syncthreads();
        int q = max(rs, base) - base;
        const int hi = min(re, base + cnt) - base;
        for (; q + 3 < hi; q += 4) {
            const int o0 = s_off[q], o1 = s_off[q + 1], o2 = s_off[q + 2], o3 = s_off[q + 3];
            const float v0 = s_val[q], v1 = s_val[q + 1], v2 = s_val[q + 2], v3 = s_val[q + 3];
            float4 x0[BT], x1[BT], x2[BT], x3[BT];
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                x0[t] = *reinterpret_cast<const float4*>(xb[t] + o0);
                x1[t] = *reinterpret_cast<const float4*>(xb[t] + o1);
                x2[t] = *reinterpret_cast<const float4*>(xb[t] + o2);
                x3[t] = *reinterpret_cast<const float4*>(xb[t] + o3);
            }
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                fma4(acc[t], v0, x0[t]);
                fma4(acc[t], v1, x1[t]);
                fma4(acc[t], v2, x2[t]);
                fma4(acc[t], v3, x3[t]);
            }
        }
        for (; q < hi; ++q) {
            const int o0 = s_off[q];
            const float v0 = s_val[q];
#pragma unroll
            for (int t = 0; t < BT; ++t) fma4(acc[t], v0, *reinterpret_cast<const float4*>(xb[t] + o0));
        }
    }

    if (p < N) {
        const int64_t orow = (int64_t)rowid[p] * W + li * 4;
#pragma unroll
        for (int t = 0; t < BT; ++t)
            if (b0 + t < B) store_row(Xout + (int64_t)(b0 + t) * N * W + orow, acc[t], store_mode);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Kernel C: any width W (G = 1, odd G, N*W >= 2^24): one thread per output element, CSR read through the caches.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void spmm_hop_generic_kernel(const int32_t* __restrict__ rowptr,
                                                                    const int32_t* __restrict__ col,
                                                                    const float* __restrict__ val,
                                                                    const int32_t* __restrict__ rowid,
                                                                    const float* __restrict__ Xin, float* __restrict__ Xout,
                                                                    int N, int W, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
        const int w = (int)(idx % W);
        const int64_t t = idx / W;
        const int p = (int)(t % N);
        const int64_t b = t / N;
        const float* xb = Xin + b * N * W + w;
        float acc = 0.f;
        for (int q = rowptr[p]; q < rowptr[p + 1]; ++q) acc = fmaf(val[q], xb[(int64_t)col[q] * W], acc);
        Xout[(b * N + rowid[p]) * W + w] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
int pick_bt(int N, int W, int B, int bl) {
    if (g_tune.spmm_bt == 1 || g_tune.spmm_bt == 2 || g_tune.spmm_bt == 4) return g_tune.spmm_bt;
    // Measured (profiles/r01_g_evgf_and_prefetch/sweep.log): while the gather panel of a tile (N*W*4*bl*bt bytes) plus the
    // prefetched next one fit the XCD's 4 MiB L2, two batch entries per lane amortise the (col, val) stream best;
    // a panel larger than L2 (N = 1e5: 12.8 MB) wants the smallest tile.
    const int64_t panel = (int64_t)N * W * 4 * bl;
    int bt = 1;
    if (panel * 2 <= (3 << 20) && B >= 16 * bl) bt = 2;
    if (panel * 4 * 2 <= (3 << 20) && B >= 32 * bl) bt = 4;
    return bt;
}

static inline int tilesPerLaneOf(int nBTiles, int lanes) { return (nBTiles + lanes - 1) / lanes; }

template <int LPR, int VPL>
int launch_sell(const gf_csr_dev& m, const float* Xin, float* Xout, int N, int B, hipStream_t st) {
    constexpr int BL = 8 / LPR, W = 4 * LPR * VPL;
    int bt = pick_bt(N, W, B, BL);
    while (bt * VPL > 8) bt >>= 1;  // bound the register tile
    int nBTiles = (B + BL * bt - 1) / (BL * bt);
    while (nBTiles < 8 && bt > 1) {  // too few batch tiles to give every XCD one: shrink the register tile first
        bt >>= 1;
        nBTiles = (B + BL * bt - 1) / (BL * bt);
    }
    // XCD roles: `lanes` batch-tile lanes x `parts` slice partitions (lanes * parts <= 8); spmm_xcd = 0 ignores the XCD
    // structure (every XCD works on the same tile).
    // (knob spmm_lanes = 1 | 2 | 4: fewer batch tiles in flight, 8 / lanes XCDs share a tile's slices -- a smaller active source set
    //  in the Infinity Cache against a lower L2 hit rate per XCD)
    int lanes = !g_tune.spmm_xcd ? 1 : (nBTiles >= 8 ? 8 : nBTiles);
    if (g_tune.spmm_lanes == 1 || g_tune.spmm_lanes == 2 || g_tune.spmm_lanes == 4) lanes = std::min(lanes, g_tune.spmm_lanes);
    const int parts = 8 / lanes;
    const int64_t tileBytes = (int64_t)N * W * 4 * BL * bt;
    const bool l2_resident = tileBytes * 2 <= (6 << 20);  // this tile's panel + the prefetched next one
    const int spw = (g_tune.spmm_spw == 1 || g_tune.spmm_spw == 2 || g_tune.spmm_spw == 4) ? g_tune.spmm_spw : (l2_resident ? 1 : 2);
    const int sliceShare = (m.n_slices + parts - 1) / parts;
    // spmm_pf: -1 = heuristic (24 prefetch workgroups per tile while panels are L2-resident, none otherwise), >= 0 = forced
    const int pf_want = g_tune.spmm_pf >= 0 ? g_tune.spmm_pf : (l2_resident ? 24 : 0);
    const int pf = tilesPerLaneOf(nBTiles, lanes) > 1 ? pf_want : 0;
    const int blocksPerTile = (sliceShare + 4 * spw - 1) / (4 * spw) + pf;
    const int tilesPerLane = (nBTiles + lanes - 1) / lanes;
    const int64_t nblk = (int64_t)8 * tilesPerLane * blocksPerTile;
    GF_REQUIRE_SHAPE(nblk < (int64_t)INT32_MAX, "gf_spmm_hop: grid of %lld blocks too large", (long long)nblk);
    dim3 grid((unsigned)nblk), block(kThreads);
    const bool uni = m.sell_uniform && m.sell_col && g_tune.panel_uniform;
#define GF_SELL_ARGS(ENT) m.sell_kptr, (const void*)(ENT), m.sell_rowid, Xin, Xout, N, B, m.n_slices, nBTiles, lanes, parts, blocksPerTile, g_tune.spmm_store, pf, m.sell_uval
#define GF_SELL(BTV, SPWV)                                                                                          \
    if (g_tune.spmm_load)                                                                                           \
        hipLaunchKernelGGL((spmm_sell_kernel<LPR, VPL, BTV, SPWV, 1, 16, 0>), grid, block, 0, st, GF_SELL_ARGS(m.sell_ent));    \
    else if (g_tune.spmm_ucap == 8)                                                                                 \
        hipLaunchKernelGGL((spmm_sell_kernel<LPR, VPL, BTV, SPWV, 0, 8, 0>), grid, block, 0, st, GF_SELL_ARGS(m.sell_ent));     \
    else if (uni)                                                                                                   \
        hipLaunchKernelGGL((spmm_sell_kernel<LPR, VPL, BTV, SPWV, 0, 16, 1>), grid, block, 0, st, GF_SELL_ARGS(m.sell_col));    \
    else                                                                                                            \
        hipLaunchKernelGGL((spmm_sell_kernel<LPR, VPL, BTV, SPWV, 0, 16, 0>), grid, block, 0, st, GF_SELL_ARGS(m.sell_ent))
#define GF_SELL_BT(SPWV)             \
    switch (bt) {                    \
        case 1: GF_SELL(1, SPWV); break; \
        case 2: GF_SELL(2, SPWV); break; \
        default: GF_SELL(4, SPWV); break; \
    }
    switch (spw) {
        case 1: GF_SELL_BT(1); break;
        case 2: GF_SELL_BT(2); break;
        default: GF_SELL_BT(4); break;
    }
#undef GF_SELL_BT
#undef GF_SELL
#undef GF_SELL_ARGS
    GF_LAUNCH_CHECK("spmm_sell_kernel");
    return GF_OK;
}

template <int LG>
int launch_vec(const gf_csr_dev& m, const float* Xin, float* Xout, int N, int B, hipStream_t st) {
    constexpr int RPB = kThreads / LG;
    const int bt = pick_bt(N, LG * 4, B, 1);
    const int xcd_map = g_tune.spmm_xcd;
    const int nRowBlocks = (N + RPB - 1) / RPB;
    const int nBTiles = (B + bt - 1) / bt;
    const int64_t nblk = xcd_map ? (int64_t)((nBTiles + 7) / 8) * 8 * nRowBlocks : (int64_t)nBTiles * nRowBlocks;
    GF_REQUIRE_SHAPE(nblk < (int64_t)INT32_MAX, "gf_spmm_hop: grid of %lld blocks too large", (long long)nblk);
    dim3 grid((unsigned)nblk), block(kThreads);
#define GF_LAUNCH_BT(BTV)                                                                                           \
    hipLaunchKernelGGL((spmm_hop_vec_kernel<LG, BTV>), grid, block, 0, st, m.rowptr, m.col, m.val, m.rowid, Xin, Xout, \
                       N, B, nRowBlocks, nBTiles, xcd_map, g_tune.spmm_store)
    switch (bt) {
        case 1: GF_LAUNCH_BT(1); break;
        case 2: GF_LAUNCH_BT(2); break;
        default: GF_LAUNCH_BT(4); break;
    }
#undef GF_LAUNCH_BT
    GF_LAUNCH_CHECK("spmm_hop_vec_kernel");
    return GF_OK;
}

}  // namespace

extern "C" int gf_spmm_hop(const gf_plan* plan, int32_t op, const float* Xin, float* Xout, int32_t B, int32_t W,
                           void* stream) {
    GF_REQUIRE_ARG(plan && Xin && Xout, "gf_spmm_hop: NULL argument");
    GF_REQUIRE_ARG(op == GF_OP_FWD || op == GF_OP_BWD, "gf_spmm_hop: op = %d", op);
    GF_REQUIRE_ARG(Xin != Xout, "gf_spmm_hop: in-place hop is not supported");
    GF_REQUIRE_SHAPE(B > 0 && W > 0, "gf_spmm_hop: bad shape B=%d W=%d", B, W);
    const gf_csr_dev& m = plan->mat[op];
    const int N = plan->n;
    hipStream_t st = gf_stream(stream);

    const bool fits24 = (int64_t)N < (1 << 24) && (int64_t)N * W < (int64_t)INT32_MAX;  // __umul24 offsets
    if (!g_tune.spmm_generic && fits24) {
        if (g_tune.spmm_algo == 0 && N >= kMsDefaultMinNodes && gf_msweep_applicable(m, N, B, W)) return gf_msweep_launch(m, Xin, Xout, 0, 1, N, B, W, st);
        if (g_tune.spmm_algo == 5) {   // experiments: the MFMA sweep or an error (never a silent fallback)
            if (!gf_msweep_applicable(m, N, B, W)) {
                gf_set_error("gf_spmm_hop: spmm_algo = 5 but the MFMA sweep does not apply (W = %d, B = %d, N = %d, fill = %.3f)", W, B, N, m.ms_fill);
                return GF_ERR_UNSUPPORTED;
            }
            return gf_msweep_launch(m, Xin, Xout, 0, 1, N, B, W, st);
        }
        if (g_tune.spmm_algo != 1) {
            switch (W) {
                case 4: return launch_sell<1, 1>(m, Xin, Xout, N, B, st);
                case 8: return launch_sell<2, 1>(m, Xin, Xout, N, B, st);
                case 16: return launch_sell<4, 1>(m, Xin, Xout, N, B, st);
                case 32: return launch_sell<8, 1>(m, Xin, Xout, N, B, st);
                case 64: return launch_sell<8, 2>(m, Xin, Xout, N, B, st);
                case 96: return launch_sell<8, 3>(m, Xin, Xout, N, B, st);
                case 128: return launch_sell<8, 4>(m, Xin, Xout, N, B, st);
                case 256: return launch_sell<8, 8>(m, Xin, Xout, N, B, st);
                default: break;
            }
        } else {
            switch (W) {
                case 4: return launch_vec<1>(m, Xin, Xout, N, B, st);
                case 8: return launch_vec<2>(m, Xin, Xout, N, B, st);
                case 16: return launch_vec<4>(m, Xin, Xout, N, B, st);
                case 32: return launch_vec<8>(m, Xin, Xout, N, B, st);
                case 64: return launch_vec<16>(m, Xin, Xout, N, B, st);
                case 128: return launch_vec<32>(m, Xin, Xout, N, B, st);
                case 256: return launch_vec<64>(m, Xin, Xout, N, B, st);
                default: break;
            }
        }
    }
    const int64_t total = (int64_t)B * N * W;
    const int64_t want = (total + kThreads - 1) / kThreads;
    dim3 grid((unsigned)(want < 65536 * 16 ? want : 65536 * 16)), block(kThreads);
    hipLaunchKernelGGL(spmm_hop_generic_kernel, grid, block, 0, st, m.rowptr, m.col, m.val, m.rowid, Xin, Xout, N, W, total);
    GF_LAUNCH_CHECK("spmm_hop_generic_kernel");
    return GF_OK;
}

// 1 when gf_spmm_hop / gf_khop run the MFMA source sweep for this call (the default from kMsDefaultMinNodes nodes on, or spmm_algo = 5)
static bool gf_hop_uses_msweep(const gf_plan* plan, int op, int B, int W) {
    const gf_csr_dev& m = plan->mat[op];
    const int N = plan->n;
    const bool fits24 = (int64_t)N < (1 << 24) && (int64_t)N * W < (int64_t)INT32_MAX;
    if (g_tune.spmm_generic || !fits24 || !gf_msweep_applicable(m, N, B, W)) return false;
    return g_tune.spmm_algo == 5 || (g_tune.spmm_algo == 0 && N >= kMsDefaultMinNodes);
}

extern "C" int gf_spmm_hop_kernel(const gf_plan* plan, int32_t op, int32_t B, int32_t W) {
    GF_REQUIRE_ARG(plan != nullptr && (op == GF_OP_FWD || op == GF_OP_BWD) && B > 0 && W > 0, "gf_spmm_hop_kernel: bad argument");
    return gf_hop_uses_msweep(plan, op, B, W) ? 1 : 0;
}

extern "C" int gf_khop(const gf_plan* const* plans, int32_t E, int32_t op, float* Z, int32_t B, int32_t W, int32_t K,
                       void* stream) {
    GF_REQUIRE_ARG(plans && Z, "gf_khop: NULL argument");
    GF_REQUIRE_ARG(op == GF_OP_FWD || op == GF_OP_BWD, "gf_khop: op = %d", op);
    GF_REQUIRE_SHAPE(E > 0 && K > 0 && B > 0 && W > 0, "gf_khop: bad shape E=%d K=%d B=%d W=%d", E, K, B, W);
    for (int e = 0; e < E; ++e) {
        GF_REQUIRE_ARG(plans[e] != nullptr, "gf_khop: plan %d is NULL", e);
        GF_REQUIRE_SHAPE(plans[e]->n == plans[0]->n, "gf_khop: plan %d has %d nodes, plan 0 has %d", e, plans[e]->n,
                         plans[0]->n);
    }
    const int64_t tap = (int64_t)B * plans[0]->n * W;
    for (int e = 0; e < E; ++e) {
        // the MFMA sweep runs the K - 1 hops of an edge feature in ONE launch, batch entry by batch entry (gf_msweep.hip)
        const gf_csr_dev& m = plans[e]->mat[op];
        if (K > 2 && g_tune.spmm_fuse && gf_hop_uses_msweep(plans[e], op, B, W) && gf_msweep_fusion_allowed()) {
            const int rc = gf_msweep_launch(m, Z, Z + (int64_t)(1 + e * (K - 1)) * tap, tap, K - 1, plans[e]->n, B, W, gf_stream(stream));
            if (rc != GF_OK) return rc;
            continue;
        }
        for (int k = 1; k < K; ++k) {
            const float* src = (k == 1) ? Z : Z + (int64_t)(1 + e * (K - 1) + (k - 2)) * tap;
            float* dst = Z + (int64_t)(1 + e * (K - 1) + (k - 1)) * tap;
            const int rc = gf_spmm_hop(plans[e], op, src, dst, B, W, stream);
            if (rc != GF_OK) return rc;
        }
    }
    return GF_OK;
}

int gf_khop_with_layout(const gf_plan* plan, int op, const float* xref, const float* xmask, float* Z, int B, int W, int K, int Nin, hipStream_t st) {
    if (!(K > 1 && W == 32 && g_tune.spmm_fuse && g_tune.spmm_xlayout && gf_hop_uses_msweep(plan, op, B, W) && gf_msweep_fusion_allowed())) return GF_ERR_UNSUPPORTED;
    const int64_t tap = (int64_t)B * plan->n * W;
    return gf_msweep_launch(plan->mat[op], Z, Z + tap, tap, K - 1, plan->n, B, W, st, xref, xmask, Nin);
}

extern "C" int gf_time_khop(const gf_plan* const* plans, int32_t E, int32_t op, float* Z, int32_t B, int32_t W, int32_t K, int32_t iters,
                            void* stream, float* avg_ms) {
    GF_REQUIRE_ARG(avg_ms && iters > 0, "gf_time_khop: bad iters / NULL avg_ms");
    hipStream_t st = gf_stream(stream);
    hipEvent_t e0, e1;
    GF_HIP(hipEventCreate(&e0));
    GF_HIP(hipEventCreate(&e1));
    int rc = gf_khop(plans, E, op, Z, B, W, K, stream);  // warm-up (also validates arguments)
    if (rc == GF_OK) {
        GF_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < iters && rc == GF_OK; ++i) rc = gf_khop(plans, E, op, Z, B, W, K, stream);
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

extern "C" int gf_time_spmm_hop(const gf_plan* plan, int32_t op, const float* Xin, float* Xout, int32_t B, int32_t W,
                                int32_t iters, void* stream, float* avg_ms) {
    GF_REQUIRE_ARG(avg_ms && iters > 0, "gf_time_spmm_hop: bad iters / NULL avg_ms");
    hipStream_t st = gf_stream(stream);
    hipEvent_t e0, e1;
    GF_HIP(hipEventCreate(&e0));
    GF_HIP(hipEventCreate(&e1));
    int rc = gf_spmm_hop(plan, op, Xin, Xout, B, W, stream);  // warm-up (also validates arguments)
    if (rc == GF_OK) {
        GF_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < iters && rc == GF_OK; ++i) rc = gf_spmm_hop(plan, op, Xin, Xout, B, W, stream);
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
