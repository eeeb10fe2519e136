// gf_common.h -- internal declarations shared by the libgfhip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gfhip.h"

// ---------------------------------------------------------------------------------------------------
// errors: thread-local message + status codes; nothing throws across the C boundary.
// ---------------------------------------------------------------------------------------------------
void gf_set_error(const char* fmt, ...);

#define GF_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            gf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);   \
            return GF_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

#define GF_LAUNCH_CHECK(name)                                                                          \
    do {                                                                                               \
        hipError_t e_ = hipGetLastError();                                                             \
        if (e_ != hipSuccess) {                                                                        \
            gf_set_error("launch of %s failed: %s", name, hipGetErrorString(e_));                      \
            return GF_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

#define GF_REQUIRE_SHAPE(cond, ...)                                                                    \
    do {                                                                                               \
        if (!(cond)) {                                                                                 \
            gf_set_error(__VA_ARGS__);                                                                 \
            return GF_ERR_SHAPE;                                                                       \
        }                                                                                              \
    } while (0)

#define GF_REQUIRE_ARG(cond, ...)                                                                      \
    do {                                                                                               \
        if (!(cond)) {                                                                                 \
            gf_set_error(__VA_ARGS__);                                                                 \
            return GF_ERR_ARG;                                                                         \
        }                                                                                              \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// plan: device CSR of one sparse operator in both orientations, rows stored in a degree-sorted schedule.
//   mat[GF_OP_FWD] = CSR of S^T  (row n lists {(i, S[i,n])}):  X_out[n] = sum_i S[i,n] X_in[i]   (x @ S)
//   mat[GF_OP_BWD] = CSR of S    (row i lists {(j, S[i,j])}):  X_out[i] = sum_j S[i,j] X_in[j]
// Row p of the stored CSR is original row rowid[p]; within a row entries keep ascending column order,
// so the per-row summation order is fixed (determinism).
// ---------------------------------------------------------------------------------------------------
struct gf_csr_dev {
    int32_t* rowptr = nullptr;  // [N+1]  (int32: nnz < 2^31)
    int32_t* col = nullptr;     // [nnz]
    float* val = nullptr;       // [nnz]
    int32_t* rowid = nullptr;   // [N]    stored position -> original row
    int32_t max_deg = 0;
    // SELL-8 image of the same scheduled rows (slices of 8 consecutive stored rows, padded to the slice's longest row):
    // entry (k, r) of slice s lives at sell_ent[(sell_kptr[s] + k) * 8 + r] = {col, bits(val)}; padding = {0, 0.0f}.
    int32_t n_slices = 0;
    int32_t* sell_kptr = nullptr;   // [n_slices + 1]  k-offsets (sum of slice widths)
    int2* sell_ent = nullptr;       // [sell_kptr[n_slices] * 8]
    int32_t* sell_rowid = nullptr;  // [n_slices * 8]  stored position -> original row, -1 past the last row
    int64_t sell_pad_entries = 0;   // padding entries (wasted gathers), for diagnostics
    int32_t* sell_col = nullptr;    // uniform values only: the columns of sell_ent alone, padding = -1 (half the entry stream)
    int32_t sell_uniform = 0;       // every stored value equals sell_uval
    float sell_uval = 0.f;
    // MSWEEP image (gf_msweep_image.h, round 5): the source sweep with the partial sums of a batch entry in the XCD's registers and an
    // fp32 MFMA as scatter-accumulate (spmm_msweep_kernel); built for graphs whose gather panel does not fit an XCD's L2
    int32_t ms_sets = 0, ms_passes = 0, ms_rounds = 0;   // 0 = no image
    int32_t ms_hub_rows = 0, ms_hub_split_rows = 0, ms_hub_limit = 0, ms_hub_split = 0;   // rows computed outside the groups (gf_msweep_image.h); rows longer than ms_hub_split are summed as 32 partial chains
    int64_t ms_hub_entries = 0;
    int32_t ms_uniform = 0;
    uint32_t* ms_ent = nullptr;     // [passes][128 waves][rounds + 2][8 positions][sets rounded up to 4]
    float* ms_val = nullptr;        // same shape (weighted GSOs only)
    uint32_t* ms_rows = nullptr;    // [passes][128 waves][sets][32]  output byte offsets
    uint32_t* ms_hub = nullptr;     // hub rows (rows too long for a group): {block offsets [passes * 128 + 1], total words, blocks, (weighted) values}; nullptr = none
    uint32_t* ms_gate = nullptr;    // XCD barrier counters, one slot per launch in rotation
    double ms_fill = 0.0;
    // Panel (LDS-resident) SpMM image, built when N <= kPanelMaxNodes.  Work unit = OCTET (8 consecutive rows = one 128-byte
    // line of a column panel); octets are sorted by their longest row and a slice = 8 octets = one wavefront (lane l handles
    // row pn_oct[8s + l/8] * 8 + l%8), so the rows a wave walks together have similar lengths and every octet still stores a
    // full line.  Entries are an ELL block per slice: group-row j (steps 4j .. 4j+3) x 64 lanes, one 16-byte word of
    // 4 LDS byte offsets (column * 16: no unpacking in the kernel) and one 16-byte word of 4 fp32 values per lane; empty
    // slots = {column N = the LDS zero slot, 0}; group-rows per slice are padded to an even count.
    //   address of (slice s, group-row j, lane l) = (pn_slice[s].x + j) * 64 + l      -- no per-lane bookkeeping at all.
    // pn_uniform: every stored value equals pn_uval (adjacency / lambda_max of an unweighted graph): the value stream is not read.
    int32_t pn_slices = 0;          // 0 = no panel image
    int2* pn_slice = nullptr;       // [pn_slices]  {group-row offset, group-rows = ceil(longest row / 4)}
    int32_t* pn_oct = nullptr;      // [pn_slices * (64 >> pn_ushift)]  unit handled by lanes (i << ushift) .. of the slice (-1 = none)
    int32_t pn_ushift = 3;          // log2(rows per unit): 3 = octets (full 128-byte lines), 2 = quads (64 bytes), 1 = pairs
    uint4* pn_col4 = nullptr;       // [(group-rows + 2) * 64]  value-free stream: LDS byte offsets (column * 16), no unpacking in the kernel
    uint2* pn_col2 = nullptr;       // [(group-rows + 2) * 64]  weighted stream: 4 x 16-bit columns (the entry bytes bound that kernel)
    float4* pn_val4 = nullptr;      // [(group-rows + 2) * 64]  weighted stream only
    int32_t pn_sentinel = 0;        // group-row index of two all-sentinel group-rows after the last real one
    double pn_fill = 1.0;           // real entries / ELL slots (diagnostic)
    int32_t pn_uniform = 0;
    float pn_uval = 0.f;
    double pn_conflict = 0.0;       // expected LDS cycles per ds_read_b128 step after the bank-aware ordering (diagnostic)
    // Chain image (gf_chain.hip): the K-1 hops of one panel run inside LDS, outputs are held in registers until every wave has
    // finished gathering and are then written over the panel.  Because outputs pass through registers, rows need not be
    // computed in natural order: rows are sorted by degree, the chunks (64 consecutive sorted rows) are dealt to the cn_waves
    // gather waves in boustrophedon order (set r of wave w = chunk r * W + w for even r, r * W + W-1-w for odd r: every wave gets
    // the same mix of long and short rows), <= kChainSets sets per wave, so the 64 rows a wave walks together have (almost) equal
    // length -- the ELL fill that lane = row lockstep wastes in the natural-order octets above (0.61 on the SBM of config 2)
    // goes to ~0.9.
    // A wave's blocks are stored back to back as 16-byte words of 8 x 16-bit columns = two group-rows (blocks padded to an even number
    // of group-rows): word u of the stream lives at cn_col8[u * 64 + lane], the values of its group-rows at cn_val4[(2u + {0,1}) * 64
    // + lane];  cn_gtab[w * 32 + 0] = first word of wave w, [w * 32 + 1 + r] = end of its block r, [w * 32 + 16 + r] = 1 when block r
    // has an odd number of group-rows (the second half of its last word is padding and is not gathered).
    int32_t cn_waves = 0;           // gather waves per workgroup (1, 2, 4, 8, 14; one or two more waves store); 0 = no chain image
    int32_t cn_sets = 0;            // row sets per wave (<= kChainSets / cn_np)
    int32_t cn_np = 1;              // panels per workgroup pass: 2 while two panels fit the LDS (N <= 5119), else 1
    uint32_t* cn_rowoff = nullptr;  // [cn_sets][cn_waves * 64]  LDS byte offset (row * 16) of the row of (set, thread); 0xffffffff = none
    int32_t* cn_gtab = nullptr;     // [cn_waves * 32]
    uint4* cn_col8 = nullptr;       // 8 x 16-bit columns per lane and word
    float4* cn_val4 = nullptr;      // 4 values (read when the stored values are not all equal)
    double cn_fill = 1.0;
    double cn_conflict = 0.0;
};
#ifndef GF_CHAIN_SETS   // experiments (make variant): sets per lane / gather waves and storer waves of a full-LDS workgroup
#define GF_CHAIN_SETS 12
#define GF_CHAIN_BIGW 14
#define GF_CHAIN_STORERS 2
#endif
constexpr int kChainSets = GF_CHAIN_SETS;      // 14 gather waves x 12 sets x 64 rows = 10752 >= kPanelMaxNodes (waves 15 and 16 store)
constexpr int kChainBigW = GF_CHAIN_BIGW, kChainStorers = GF_CHAIN_STORERS;
constexpr int32_t kPanelMaxNodes = 10239;   // 16 bytes per node + one zero slot in 160 KiB of LDS
constexpr int32_t kPanelMaxDeg = 65535;

struct gf_plan {
    int32_t n = 0;
    int64_t nnz = 0;
    int64_t device_bytes = 0;
    gf_csr_dev mat[2];
};

static inline hipStream_t gf_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// tap-stack index helpers (see gfhip.h "Layouts")
static inline int gf_num_taps(int E, int K) { return 1 + E * (K - 1); }

// debugging / tuning knobs (gf_tune): not part of the numerical contract, every setting gives identical results
struct gf_tuning {
    int spmm_bt = 0;            // 0 = heuristic, else 1 / 2 / 4 batch entries per lane
    int spmm_spw = 0;           // 0 = default (2), else 1 / 2 / 4 consecutive slices per wave
    int spmm_generic = 0;       // 1 = force the generic one-thread-per-element kernel
    int spmm_algo = 0;          // 0 = default (the MFMA source sweep where gf_msweep_applicable says so, else SELL-8), 1 = CSR workgroup-staged kernel
                                // (first version), 3 = SELL-8 always, 5 = the MFMA source sweep or GF_ERR_UNSUPPORTED
    int spmm_bar = 0;           // MFMA sweep: XCD barrier between batch entries too (0 = only between the hops of an entry, where the next hop reads what this one stored)
    int spmm_pfd = 0;           // MFMA sweep: scalar prefetch of the source rows this many loop iterations (of two rounds) ahead of the sweep's nominal position; 0 = auto:
                                // (rounds + 7) / 14 -- a row's first use comes up to ~ rounds / 8 early (config 4: 42 rounds -> 3 iterations; from 4 on the rows start to
                                // fall out of the 4 MiB L2 again; a degree-4 graph with 18 rounds wants 1, a degree-20 one with 84 rounds 6); -1 = off
    int spmm_passes = 2;        // MFMA sweep image: passes (sweeps of the sources) allowed per batch entry -- 1: N <= 102 400, 2: up to 204 800 (set BEFORE gf_plan_create)
    int spmm_xlayout = 1;       // MFMA sweep: the boundary layout pass (x / dy -> tap 0) inside the fused chain launch (32-column rows, one edge feature); 0 = separate kernel
    int spmm_hublim = 0;        // MFMA sweep image, experiments: rows longer than this are hub rows (0 = the builder's cost model; set BEFORE gf_plan_create)
    int spmm_minwork = 5;       // MFMA sweep: fewest (batch entry, 32-column slab) pairs it takes (fewer: SELL-8 is faster -- an XCD walks a pair in ~65 us per hop alone or not)
    int spmm_census = 0;        // MFMA sweep, experiments (tests of the abandon-and-repair path): 1 = the census is called bad, 2 = one workgroup claims the
                                // next XCC (a 33 / 31 census), 3 = one workgroup never arrives (the others run into the time limit: the slot is poisoned)
    int spmm_tmo_ms = 0;        // MFMA sweep, experiments: time limit of the census / the barriers in ms (0 = 2000)
    int spmm_trace = 0;         // MFMA sweep, experiments: record phase time stamps (gf_debug_msweep_trace)
    int spmm_fuse = 1;          // MFMA sweep: the K - 1 hops of gf_khop in one launch, entry by entry (0 = one launch per hop)
    int spmm_depth = 0;         // MFMA sweep: gathers in flight per wave, 0 = default (10 from 15 sets per wave on, 5 at 10 sets), 5
    int spmm_srcmask = 0;       // MFMA sweep, timing experiments only (results wrong): AND mask on the gathered source offsets (confines them to a window)
    int spmm_slack = 5;         // MFMA sweep image: rounds beyond the mean group length, in percent (set BEFORE gf_plan_create)
    int spmm_xcd = 1;           // 1 = XCD-aware tile order
    int spmm_ucap = 0;          // 0/16 = up to 16 gathers in flight per lane, 8 = up to 8 (fewer registers, more waves)
    int spmm_pf = -1;           // workgroups per tile that prefetch the next tile's gather panel into L2 (-1 = heuristic, 0 = off)
    int spmm_group = 1;         // 1 = rows of large graphs scheduled in locality groups (set BEFORE gf_plan_create)
    int spmm_load = 0;          // gather loads: 0 = plain, 1 = non-temporal
    int spmm_store = 2;         // output rows: 0 = plain stores, 1 = write-through (sc1), 2 = non-temporal
    int contract_generic = 0;   // 1 = force the generic contraction kernel
    int gradw_lds = 1;          // panel tap gradients: 1 = operands transposed through LDS (coalesced loads), 0 = direct strided loads
    int pipeline = 0;           // 0 = auto, 1 = node-major (gather through L2), 2 = column panels through LDS (needs N <= 10239, G%8 == F%8 == 0)
    int panel_uniform = 1;      // 1 = use the value-free stream when the plan detected uniform values
    int panel_order = 1;        // 1 = bank-aware neighbour order at plan creation (set BEFORE gf_plan_create)
    int panel_chain = 1;        // the K-1 hops of a panel inside LDS (gf_chain.hip): 1 = when there are enough panels to fill the CUs,
                                // 2 = always, 0 = never (one launch per hop, gf_panel.hip)
    int panel_rotate = 1;       // 1 = each workgroup walks the slice list from its own starting offset
    int panel_grid = 0;         // experiments: cap on the panel kernel's grid (0 = one workgroup per LDS-full)
    int bwd_fuse64 = 1;         // ... also for F = 64 (G <= 32): 0 = tap-gradient kernel + contraction, as in round 3
    int bwd_fuse = 1;           // panel pipeline backward: 1 = dx and dh from the adjoint stack in one kernel (G, F <= 32), 0 = separate
    int panel_split = 0;        // workgroups per pass when there are fewer passes than CUs: 0 = as many as fit (<= 8), 1 = off
    int panel_unit = 8;         // rows per work unit of the panel image: 8 | 4 | 2 (set BEFORE gf_plan_create)
    int panel_np = 0;           // panels per workgroup pass: 0 = heuristic (2 while two workgroups still fit a CU's LDS), 1, 2
    int spmm_lanes = 0;         // experiments: batch tiles in flight in the SELL kernels (0 = one per XCD)
    int panel_db = 1;           // the double-buffered per-hop kernel (spmm_panel_db_kernel): 1 = where one workgroup fills a CU's LDS
                                // with its four panels (1280 <= N <= 2559), 2 = wherever four panels fit, 0 = never
    int panel_thr = 0;          // experiments: threads of its workgroup (0 = 1024 when one workgroup fills a CU, else 512)
    int panel_loaders = 0;      // experiments: its loader waves (0 = 2 of 16 waves, 1 of 8)
    int panel_even = 0;         // 1 = pad every slice to an even number of group-rows (set BEFORE gf_plan_create)
    int panel_sort = 1;         // 1 = octets sorted by their longest row (set BEFORE gf_plan_create)
    int evgf_idx16 = 1;         // EVGF: 16-bit node indices when N <= 65535 (half the index streams that share L2 with the gather panel)
    int evgf_generic = 0;       // EVGF taps: 0 = best kernel, 1 = one thread per output, 2 = LDS-staged 4-byte gathers (no 16-byte version)
};
extern gf_tuning g_tune;

// internal launchers shared between translation units
bool gf_msweep_applicable(const gf_csr_dev& m, int N, int B, int W);
// nhops hops in one launch: hop h reads Xin (h = 0) or Xtaps + (h - 1) * tapStride floats and writes Xtaps + h * tapStride floats
// xref != nullptr (W == 32): the boundary layout pass is part of the launch -- every batch entry's x[b][0..32)[0..Nin) (reference layout; xmask: entries
// whose mask is <= 0 give 0) is written to Xin[b] as node-major rows (rows >= Nin zero) by the XCD that then walks the entry through its hops
int gf_msweep_launch(const gf_csr_dev& m, const float* Xin, float* Xtaps, int64_t tapStride, int nhops, int N, int B, int W, hipStream_t st,
                     const float* xref = nullptr, const float* xmask = nullptr, int Nin = 0);
// gf_khop for one edge feature with the layout pass folded into the launch where the sweep runs fused (else GF_ERR_UNSUPPORTED: the caller
// runs the layout kernel and gf_khop)
int gf_khop_with_layout(const gf_plan* plan, int op, const float* xref, const float* xmask, float* Z, int B, int W, int K, int Nin, hipStream_t st);
size_t gf_msweep_gate_bytes();
unsigned* gf_msweep_status_word();          // pinned host word the repair kernel reports into (allocated on first use: call at plan creation)
bool gf_msweep_fusion_allowed();            // false once a launch had to be repaired, or with GFHIP_MSWEEP_FUSE=0
void gf_msweep_status_reset();              // experiments
constexpr int32_t kMsMinNodes = 32768;      // GFHIP_EXPERIMENTS=1 processes build an image from here on (below, a batch entry's rows -- N x 128 bytes -- fit the 4 MiB L2 of an XCD)
constexpr int32_t kMsDefaultMinNodes = 49152;   // the default hop uses it from here on, and every process builds the image from here on (measured: SELL-8 wins at 33k / 40k, the sweep from 50k on)
int gf_contract_launch(const float* Z, const float* h, const float* bias, float* out, int B, int N, int Nout, int G,
                       int F, int E, int K, int transpose_bank, hipStream_t st, int out_rows = 0, const float* mask = nullptr);
// column-panel pipeline (gf_panel.hip / gf_contract.hip / gf_gradw.hip)
bool gf_bwd_fused_supported(int G, int F, int E, int K);
int gf_bwd_fused_panel_launch(const float* Pp, const float* X0p, const float* h, float* dx, float* dh, float* dbias, void* workspace,
                              size_t workspace_bytes, int B, int N, int Nout, int G, int F, int E, int K, hipStream_t st, int node_major = 0,
                              int dx_panels = 0, const float* maskp = nullptr);
bool gf_panel_supported(const gf_plan* const* plans, int E, int G, int F, int K);
bool gf_contract_panel_fits(int Cin, int Cout, int T);
bool gf_chain_available(const gf_plan* plan, int op);
int gf_spmm_chain_launch(const gf_plan* plan, int op, const float* Xin, float* Xout, int nPanels, int nHops, int64_t tapStride,
                         hipStream_t st);
// kernels that need more than 64 KiB of dynamic LDS: raise the limit once per (device, kernel), not on every launch
hipError_t gf_grant_lds(const void* kernel, size_t lds_bytes);
// kernels that address their dynamic LDS panel ABSOLUTELY (it must start at LDS address 0): GF_OK when the kernel's code object holds no static
// __shared__ object, GF_ERR_UNSUPPORTED (with a message) otherwise -- asked of the runtime once per (device, kernel), before the launch
int gf_require_no_static_lds(const void* kernel, const char* name);
int gf_pack_panels_launch(const float* x, float* Xp, int B, int C, int Nin, int N, hipStream_t st, const float* mask);
int gf_layout_masked_launch(const float* dy, const float* y, float* X, int B, int G, int Nin, int N, hipStream_t st);
int gf_spmm_panel_launch(const gf_plan* plan, int op, const float* Xin, float* Xout, int nPanels, hipStream_t st);
bool gf_panel_db_applies(const gf_plan* plan, int op, int nPanels);
int gf_contract_panel_launch(const float* Zp, const float* h, const float* bias, float* out, int B, int N, int Nout, int G,
                             int F, int E, int K, int transpose_bank, hipStream_t st, int out_panels = 0, const float* maskp = nullptr);
