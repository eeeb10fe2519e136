/* gfhip.h -- C ABI of libgfhip.so: the MI355X (gfx950) implementation of the LSIGF / GraphFilter hot path
 * of alelab-upenn/graph-neural-networks ("alegnn").
 *
 * The reference has no FFI / plugin registry: callers bind the Python symbols
 *     alegnn/utils/graphML.py:83     LSIGF(h, S, x, b)
 *     alegnn/utils/graphML.py:2036   class GraphFilter  (addGSO :2116, forward :2125)
 *     alegnn/utils/graphML.py:389    EVGF / :2511 class EdgeVariantGF
 * by name.  This header is what a Python (ctypes) / C / C++ host binds instead; each entry point says which
 * reference lines it replaces.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - plain C types only; every tensor argument is a DEVICE pointer to fp32 data unless it says HOST.
 *   - the library never allocates or frees tensor storage; the caller passes outputs and workspaces.
 *     The only library-owned object is the opaque plan (device CSR of S and S^T + row schedule).
 *   - every launch goes to the hipStream_t passed as `stream` (void*; NULL = default stream);
 *     no internal synchronisation, no host callbacks.
 *   - return value: 0 = GF_OK, negative = error (see enum); gf_last_error() gives a thread-local message.
 *     Shape violations that the reference reports with `assert` (graphML.py:135-140, 2118-2122) come back
 *     as GF_ERR_SHAPE so the Python layer can re-raise AssertionError.
 *   - results are bitwise run-to-run deterministic for a given call shape (no floating-point atomics).  The kernel
 *     (and with it the summation order) is chosen from the shapes, batch size included: the same sample evaluated at
 *     B = 1 and inside a batch of 256 agrees to fp32 rounding, not bit for bit.
 *
 * Layouts
 *   reference layout   x [B, G, N]  (node index contiguous)          graphML.py:108-109
 *   node-major layout  X [B, N, G]  (feature index contiguous)       internal; one 128-byte line per (b, n) at G = 32
 *   tap stack          Z [T, B, N, G], T = 1 + E*(K-1):  tap 0 = X_0 (shared by all e, graphML.py:154),
 *                      tap 1 + e*(K-1) + (k-1) = x S_e^k  for k = 1..K-1      (replaces the cat at graphML.py:161)
 *   filter taps        h [F, E, K, G]   reference parameter layout    graphML.py:2101  (read in place, never repacked)
 */
#ifndef GFHIP_H
#define GFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFHIP_VERSION 100 /* major*10000 + minor*100 + patch */

enum {
    GF_OK = 0,
    GF_ERR_SHAPE = -1,       /* the reference would have raised AssertionError            */
    GF_ERR_ARG = -2,         /* null pointer / bad enum / unsupported value               */
    GF_ERR_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString) */
    GF_ERR_UNSUPPORTED = -4, /* valid request this build cannot serve                      */
    GF_ERR_NOMEM = -5
};

/* which operator a hop applies on the node axis */
enum {
    GF_OP_FWD = 0, /* X_out = S^T X_in  == reference row-vector product x @ S (graphML.py:159)          */
    GF_OP_BWD = 1  /* X_out = S   X_in  == its adjoint, used by the backward pass (autograd of :159)    */
};

typedef struct gf_plan gf_plan; /* opaque */

int gf_version(void);
const char* gf_last_error(void);

/* ---- GSO ingest: replaces holding the dense [E,N,N] tensor of GraphFilter.addGSO (graphML.py:2116-2123) ------
 * Builds the device plan for ONE edge feature S_e from HOST CSR arrays of S_e (row i lists S_e[i, :]).
 * Duplicate (i,j) entries are summed in order; explicit zeros are kept.  vals may be fp32 or fp64 (vals_is_f64).
 * flags: bit 0 = disable the degree-sorted row schedule (debug). */
int gf_plan_create(int32_t n_nodes, int64_t nnz, const int32_t* rowptr_host, const int32_t* colidx_host,
                   const void* vals_host, int32_t vals_is_f64, uint32_t flags, gf_plan** out_plan);
int gf_plan_destroy(gf_plan* plan);
/* n_nodes, nnz, and the device bytes held by the plan */
int gf_plan_info(const gf_plan* plan, int32_t* n_nodes, int64_t* nnz, int64_t* device_bytes);

/* ---- boundary layout kernels (replace nothing in the reference: they are the price of node-major gathers) ---- */
/* x [B,G,Nin] -> X [B,N,G]; rows n >= Nin are zero-filled == GraphFilter.forward zero-padding (graphML.py:2131-2135) */
int gf_layout_bgn_to_bng(const float* x, float* X, int32_t B, int32_t G, int32_t Nin, int32_t N, void* stream);
/* X [B,N,G] -> x [B,G,Nout], keeping nodes n < Nout == the index_select at graphML.py:2142-2143 */
int gf_layout_bng_to_bgn(const float* X, float* x, int32_t B, int32_t G, int32_t N, int32_t Nout, void* stream);

/* ---- one hop: X_out[b] = op(S) X_in[b] for b < B, rows of width W floats (replaces torch.matmul, graphML.py:159) */
int gf_spmm_hop(const gf_plan* plan, int32_t op, const float* X_in, float* X_out, int32_t B, int32_t W, void* stream);

/* ---- K-hop tap stack for E plans: fills taps 1..T-1 of Z from tap 0 (replaces the loop graphML.py:158-161) ---- */
int gf_khop(const gf_plan* const* plans, int32_t E, int32_t op, float* Z, int32_t B, int32_t W, int32_t K,
            void* stream);

/* ---- filter-bank contraction (replaces permute+matmul+permute+bias, graphML.py:170-175)
 * transpose_bank = 0 (forward):   out[b, f, n] = bias[f] + sum_{t,g} Z[t,b,n,g] * h[f, e(t), k(t), g]      Cin = G, Cout = F
 * transpose_bank = 1 (backward):  out[b, g, n] =           sum_{t,f} Z[t,b,n,f] * h[f, e(t), k(t), g]      Cin = F, Cout = G
 * Z [T,B,N,Cin] node-major tap stack; out [B,Cout,Nout] in the REFERENCE layout, nodes n < Nout only; bias nullable [F]. */
int gf_contract(const float* Z, const float* h, const float* bias, float* out, int32_t B, int32_t N, int32_t Nout,
                int32_t G, int32_t F, int32_t E, int32_t K, int32_t transpose_bank, void* stream);

/* ---- filter-tap gradient (autograd of graphML.py:170-175 wrt h and b):
 * dh[f,e,k,g] = sum_{b,n} Z[t(e,k),b,n,g] * P0[b,n,f]   (for k = 0 the same value is written for every e)
 * dbias[f]    = sum_{b,n} P0[b,n,f]                      (dbias nullable)
 * P0 [B,N,F] is dy in node-major layout.  workspace: gf_grad_taps_workspace_bytes() bytes of device scratch. */
size_t gf_grad_taps_workspace_bytes(int32_t B, int32_t N, int32_t G, int32_t F, int32_t E, int32_t K);
int gf_grad_taps(const float* Z, const float* P0, float* dh, float* dbias, void* workspace, size_t workspace_bytes,
                 int32_t B, int32_t N, int32_t G, int32_t F, int32_t E, int32_t K, void* stream);

/* ---- whole-layer entry points: what GraphFilter.forward / its autograd call (graphML.py:2125-2144) -------------
 * forward:  x [B,G,Nin], h [F,E,K,G], bias [F]|NULL  ->  y [B,F,Nin];  Z [T,B,N,G] is written (save it for backward).
 * backward: dy [B,F,Nin], Z (saved), h -> dx [B,G,Nin] (NULL = skip), dh [F,E,K,G] (NULL = skip), dbias [F] (NULL = skip);
 *           P [T,B,N,F] scratch, workspace as for gf_grad_taps.  */
int gf_lsigf_forward(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias,
                     float* Z, float* y, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream);
int gf_lsigf_backward(const gf_plan* const* plans, int32_t E, const float* dy, const float* Z, const float* h,
                      float* P, float* dx, float* dh, float* dbias, void* workspace, size_t workspace_bytes,
                      int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream);

/* ---- column-panel pipeline: the same hot path with the gathers served from LDS (N <= 10239 nodes; G and F in {8, 16, 32, 64, 128}).
 * Layout: Xp[P][N][4], P = B*C/4 panels of 4 consecutive signal columns (b, c..c+3): 16 bytes per node, one panel fills
 * at most 160 KiB = one CU's LDS.  Tap stack Zp[T][B*C/4][N][4] (same size as the node-major stack).
 * gf_lsigf_forward / _backward pick this pipeline by themselves (gf_lsigf_pipeline tells which: 1 node-major, 2 panels);
 * the entry points below expose its stages for tests and profiling. */
int gf_lsigf_pipeline(const gf_plan* const* plans, int32_t E, int32_t G, int32_t F, int32_t K);
/* x [B,C,Nin] -> Xp [B*C/4][N][4], nodes n >= Nin zero (GraphFilter.forward's padding, graphML.py:2131-2135); C % 4 == 0 */
int gf_pack_panels(const float* x, float* Xp, int32_t B, int32_t C, int32_t Nin, int32_t N, void* stream);
/* Xp [B*C/4][N][4] -> x [B,C,Nout], nodes n < Nout (graphML.py:2142-2143) */
int gf_unpack_panels(const float* Xp, float* x, int32_t B, int32_t C, int32_t N, int32_t Nout, void* stream);
/* one hop on n_panels panels: Xout[p] = op(S) Xin[p]   (replaces torch.matmul, graphML.py:159) */
int gf_spmm_hop_panel(const gf_plan* plan, int32_t op, const float* Xin, float* Xout, int32_t n_panels, void* stream);
int gf_time_spmm_hop_panel(const gf_plan* plan, int32_t op, const float* Xin, float* Xout, int32_t n_panels, int32_t iters,
                           void* stream, float* avg_ms);
/* the whole tap stack in panel layout, Zp [T][B*W/4][N][4]: tap 0 is the caller's (gf_pack_panels), taps 1 + e(K-1) + (k-1) =
 * op(S_e)^k tap 0 are written -- the loop at graphML.py:158-161 (matmul per tap + cat).  Each panel is loaded once and walks
 * its K-1 hops inside LDS (gf_chain.hip); W % 4 == 0.  gf_time_khop_panel: the same call `iters` times between two HIP
 * events on `stream`, average milliseconds per call (per K-1 hops of every edge feature). */
int gf_khop_panel(const gf_plan* const* plans, int32_t E, int32_t op, float* Zp, int32_t B, int32_t W, int32_t K, void* stream);
/* 1 when gf_khop_panel runs the chain kernel (spmm_chain_kernel, one launch per edge feature) for n_panels panels on this plan, 0 when it
 * runs one spmm_panel_kernel launch per hop (few panels; small weighted GSOs), 2 when those per-hop launches are the double-buffered
 * spmm_panel_db_kernel (1280 <= N <= 2559): which kernel a profile of the call shows */
int gf_khop_panel_uses_chain(const gf_plan* plan, int32_t op, int32_t n_panels);
int gf_time_khop_panel(const gf_plan* const* plans, int32_t E, int32_t op, float* Zp, int32_t B, int32_t W, int32_t K, int32_t iters,
                       void* stream, float* avg_ms);
/* as gf_contract / gf_grad_taps with Z (and P0) in panel layout */
int gf_contract_panel(const float* Zp, const float* h, const float* bias, float* out, int32_t B, int32_t N, int32_t Nout,
                      int32_t G, int32_t F, int32_t E, int32_t K, int32_t transpose_bank, void* stream);
int gf_grad_taps_panel(const float* Zp, const float* P0p, float* dh, float* dbias, void* workspace, size_t workspace_bytes,
                       int32_t B, int32_t N, int32_t G, int32_t F, int32_t E, int32_t K, void* stream);
/* panel image of the plan: slices of 64 rows (0 = none, N too large), whether all stored values are equal (value-free
 * stream), and the modelled LDS cycles per ds_read_b128 step after the bank-aware neighbour ordering (4.0 = conflict-free) */
int gf_plan_panel_info(const gf_plan* plan, int32_t op, int32_t* n_slices, int32_t* uniform, double* lds_cycles_per_step,
                       double* fill);

/* ---- edge-variant graph filter, per-edge storage: EVGF (graphML.py:389-488) as called by EdgeVariantGF.forward
 * (graphML.py:2670-2698).  ONE edge feature per call (the host sums over e; EVGF is linear in e).  The reference holds
 * Phi = weightEV * sparsityPatternFull as a dense [F,E,K,G,N,N] tensor; here only the entries its mask keeps exist:
 *   wdiag [F,G,N]         = Phi[:, e, 0, :, n, n]        tap 0 is diagonal (identity & hybrid mask, graphML.py:2653-2668)
 *   wedge [F,K-1,G,nnzp]  = Phi[:, e, 1:, :, i_p, j_p]   taps k >= 1 on the pattern (|S|+I > 1e-9 & hybrid mask, :2617-2643)
 * gf_ev_plan: device image of the pattern CSR (HOST arrays; row i lists its columns j strictly ascending; the entry
 * order defines the value index p) and of its transpose.  Column convention v_k = Phi_k v_{k-1} (graphML.py:464, 475).
 *   forward : x [B,G,Nin] (zero-padded to N, :2678-2680) -> y [B,F,Nin] = sum_{g,k} v_k^{fg} + bias[f];
 *             V [K, F*G, N, B] receives every chain state (save it for backward).
 *   backward: dy [B,F,Nin], x, V -> dx [B,G,Nin], dwdiag [F,G,N], dwedge [F,K-1,G,nnzp], dbias [F]   (each nullable = skip)
 * scratch: gf_evgf_scratch_floats(B, G, F, N, backward) floats of device memory. */
typedef struct gf_ev_plan gf_ev_plan; /* opaque */
int gf_ev_plan_create(int32_t n_nodes, int64_t nnzp, const int32_t* rowptr_host, const int32_t* colidx_host,
                      gf_ev_plan** out_plan);
int gf_ev_plan_destroy(gf_ev_plan* plan);
int gf_ev_plan_info(const gf_ev_plan* plan, int32_t* n_nodes, int64_t* nnzp, int64_t* device_bytes);
size_t gf_evgf_scratch_floats(int32_t B, int32_t G, int32_t F, int32_t N, int32_t backward);
int gf_evgf_forward(const gf_ev_plan* plan, const float* x, const float* wdiag, const float* wedge, const float* bias, float* V,
                    float* scratch, float* y, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream);
int gf_evgf_backward(const gf_ev_plan* plan, const float* dy, const float* x, const float* wdiag, const float* wedge,
                     const float* V, float* scratch, float* dx, float* dwdiag, float* dwedge, float* dbias, int32_t B, int32_t G,
                     int32_t F, int32_t K, int32_t Nin, void* stream);

/* ---- MaxPoolLocal (graphML.py:1890-2028), the pooling that follows the filter in every SelectionGNN layer
 * (architectures.py:286-294).  nbh [Nout, M] int32 DEVICE: row i = the alpha-hop neighbourhood of node i, padded with i
 * (graphTools.computeNeighborhood(..., 'matrix'), graphML.py:1953-1959).  forward: x [B,F,Nin] -> v [B,F,Nout] = max over the
 * list (replaces repeat + gather + max, :2003-2018), arg [B,F,Nout] = list position of the FIRST maximum (torch.max's rule).
 * backward: dx [B,F,Nin] from dv [B,F,Nout] through reverse lists (DEVICE int32 CSR over input nodes j: rev_ptr [Nin+1],
 * rev_i = output node, rev_p = first position of j in nbh[rev_i]); a gather in fixed order, no atomics. */
int gf_maxpool_forward(const float* x, const int32_t* nbh, float* v, int32_t* arg, int32_t B, int32_t F, int32_t Nin, int32_t Nout,
                       int32_t M, void* stream);
int gf_maxpool_backward(const float* dv, const int32_t* arg, const int32_t* rev_ptr, const int32_t* rev_i, const int32_t* rev_p,
                        float* dx, int32_t B, int32_t F, int32_t Nin, int32_t Nout, void* stream);

/* ---- Node-variant graph filter, NVGF (graphML.py:293-387) / NodeVariantGF.forward (:2475-2498): the LSIGF tap stack contracted
 * with a bank that has its own taps at every node:  y[b,f,n] = bias[f] + sum_{e,k,g} h[f,e,k,g,n] (x_g S_e^k)[b,n].
 * h [F,E,K,G,N] in the reference layout (already expanded over copyNodes, :2485); x [B,G,Nin] zero-padded to N, y [B,F,Nin].
 * forward writes the node-major tap stack Z [T,B,N,G], T = 1+E(K-1) (save it for backward); backward: dx [B,G,Nin] and/or
 * dh [F,E,K,G,N] (NULL = skip).  scratch: gf_nvgf_scratch_floats(..., backward) floats of device memory. */
size_t gf_nvgf_scratch_floats(int32_t B, int32_t N, int32_t G, int32_t F, int32_t E, int32_t K, int32_t backward);
int gf_nvgf_forward(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias, float* Z, float* y,
                    float* scratch, size_t scratch_floats, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream);
int gf_nvgf_backward(const gf_plan* const* plans, int32_t E, const float* dy, const float* Z, const float* h, float* dx, float* dh,
                     float* scratch, size_t scratch_floats, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream);
/* adjoint of NodeVariantGF's tap expansion h = weight[..., copyNodes] (graphML.py:2485; autograd's index_add there):
 * dweight [R, M] = sum of dh [R, N] over the nodes that copy tap node m, R = F*E*K*G.  DEVICE int32 CSR of the groups:
 * grp_ptr [M+1], grp_idx [N] (nodes of group m ascending) -- a gather in fixed order, no atomics. */
int gf_nvgf_fold_taps(const float* dh, const int32_t* grp_ptr, const int32_t* grp_idx, float* dweight, int64_t R, int32_t N,
                      int32_t M, void* stream);

/* ---- GraphFilter followed by sigma = ReLU (SelectionGNN layers, architectures.py:286-289): y = max(0, LSIGF(...)) fused into the
 * contraction's epilogue; backward takes the saved output y [B,F,Nin] and applies the mask (y > 0) to dy on the way in. */
int gf_lsigf_forward_relu(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias,
                          float* Z, float* y, int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream);
int gf_lsigf_backward_relu(const gf_plan* const* plans, int32_t E, const float* dy, const float* y, const float* Z, const float* h,
                           float* P, float* dx, float* dh, float* dbias, void* workspace, size_t workspace_bytes,
                           int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, void* stream);

/* ---- filters whose GSO differs per sample and time step: the reference's "_DB" family and edge gating ----------------------------
 * S is the reference's dense tensor [B, T, E, N, N] (LSIGF_DB graphML.py:977-1094: `torch.matmul(x, S)` at :1069 with the time
 * shift of :1062-1067), on the device; signals are node-major [nb * nt, N, W] rows (W % 4 == 0, W <= 256).
 * gf_db_hop: one hop of every (b, t) in one launch.  S points at edge feature e of (b = 0, t = 0); s_stride_b / s_stride_t are the
 *   element strides between samples / time steps (T*E*N*N and E*N*N for the full tensor; pass a pointer offset by t0 time steps and
 *   nt = 1 for the per-time-step hop of GRNN_DB :1224-1262 or of the edge-gated recursion :1434-1456).
 *     op = GF_OP_FWD:  X_out[b,t] = X_in[b,t-shift] @ S[b,t]       (zero for t - shift < 0)            x S_t of :1069 / :1407 / :1445
 *     op = GF_OP_BWD:  X_out[b,t] = X_in[b,t+shift] @ S[b,t+shift]^T (zero for t + shift >= nt)          its adjoint (autograd)
 * gf_db_grad_gso: dS[b,t,m,n] (+)= sum_w X_in[b,t-shift,m,w] * dOut[b,t,n,w] -- gradient of the hop with respect to the operator,
 *   needed only when S is a function of learnable gates (edge gating: S = q * S_e, :1397-1399).
 * gf_stack_adjoint: dZ[t,b,n,g] = sum_f P0[b,n,f] h[f,e(t),k(t),g] -- adjoint of gf_contract per tap (tap 0 sums h over e). */
int gf_db_hop(const float* S, int64_t s_stride_b, int64_t s_stride_t, const float* X_in, float* X_out, int32_t nb, int32_t nt, int32_t N,
              int32_t W, int32_t op, int32_t shift, void* stream);
int gf_db_grad_gso(const float* X_in, const float* dOut, float* dS, int64_t s_stride_b, int64_t s_stride_t, int32_t nb, int32_t nt, int32_t N,
                   int32_t W, int32_t shift, int32_t accumulate, void* stream);
int gf_stack_adjoint(const float* P0, const float* h, float* dZ, int64_t BN, int32_t G, int32_t F, int32_t E, int32_t K, void* stream);
/* whole layer (GraphFilter_DB.forward graphML.py:3356-3369 -> LSIGF_DB :977-1094, and its autograd):
 *   forward:  S [B,T,E,N,N], x [B,T,G,N], h [F,E,K,G], bias [F]|NULL -> y [B,T,F,N]; Z [1+E(K-1), B*T, N, G] is written (save it).
 *             shift = 1: the delayed filter z_k(t) = z_{k-1}(t-1) S(t); shift = 0: per-(b,t) operator without delay (edge gating).
 *   backward: dy [B,T,F,N] -> dx [B,T,G,N], dh, dbias, dS [B,T,E,N,N] (each nullable = skipped); scratch: P0 [B*T,N,F],
 *             dZ [1+E(K-1), B*T, N, G], hop [B*T,N,G]; workspace as for gf_grad_taps with batch B*T. */
int gf_lsigf_db_forward(const float* S, const float* x, const float* h, const float* bias, float* Z, float* y, int32_t B, int32_t T, int32_t G,
                        int32_t F, int32_t E, int32_t K, int32_t N, int32_t shift, void* stream);
int gf_lsigf_db_backward(const float* S, const float* dy, const float* Z, const float* h, float* P0, float* dZ, float* hop_scratch, float* dx,
                         float* dh, float* dbias, float* dS, void* workspace, size_t workspace_bytes, int32_t B, int32_t T, int32_t G,
                         int32_t F, int32_t E, int32_t K, int32_t N, int32_t shift, void* stream);

/* ---- layer-to-layer hand-over in the internal layout (no reference counterpart: the reference permutes at every layer,
 * graphML.py:170-171; SelectionGNN strings [filter, sigma, rho] blocks together, architectures.py:286-294).  "Internal layout" = the
 * layout of the pipeline gf_lsigf_pipeline() names for the layer: column panels [B*C/4][N][4] (2), or node-major rows [B][N][C] (1,
 * widths that are multiples of 4); neighbouring layers must run the same pipeline.  Nin == N, else GF_ERR_UNSUPPORTED.
 *   flags bit 0 (forward): fused ReLU epilogue, as gf_lsigf_forward_relu
 *   flags bit 1: tap 0 of the stack (Z for forward, P for backward) ALREADY holds the input in the internal layout -- x / dy are
 *                ignored (NULL allowed): the neighbouring layer's call wrote it there
 *   flags bit 2: the result (y, resp. dx) is written in the internal layout -- pass the neighbouring layer's tap 0;
 *                backward: masked by dx_mask (same shape and layout, nullable; entries <= 0 give 0 = the ReLU of the layer below,
 *                whose activation is tap 0 of THIS layer's forward stack). */
int gf_lsigf_forward_ex(const gf_plan* const* plans, int32_t E, const float* x, const float* h, const float* bias, float* Z, float* y,
                        int32_t B, int32_t G, int32_t F, int32_t K, int32_t Nin, int32_t flags, void* stream);
int gf_lsigf_backward_ex(const gf_plan* const* plans, int32_t E, const float* dy, const float* y_relu, const float* Z, const float* h,
                         float* P, float* dx, float* dh, float* dbias, void* workspace, size_t workspace_bytes, int32_t B, int32_t G,
                         int32_t F, int32_t K, int32_t Nin, int32_t flags, const float* dx_mask, void* stream);

/* ---- measurement hook: run ONE hop `iters` times on `stream` bracketed by HIP events on that stream and return
 * the average milliseconds per launch (bench.py's roofline leg; hipEvents see the launch stream, torch events may not). */
/* which kernel a node-major hop of this shape runs: 1 = spmm_msweep_kernel (the MFMA source sweep, gf_msweep.hip: W = 32 / 64 / 96 / 128 -- wide
 * rows as slabs of 32 columns --, graphs from 49 152 nodes on whose row groups balance, B * W / 32 >= 5; gf_khop then runs the K-1 hops of an
 * edge feature in ONE launch), 0 = spmm_sell_kernel / the others */
int gf_spmm_hop_kernel(const gf_plan* plan, int32_t op, int32_t B, int32_t W);
/* State of the fused chains of this process (no reference counterpart).  A fused launch opens with a census of its workgroups (32 on each of
 * 8 XCCs, all resident); a launch whose census fails, or that runs into its time limit, stores nothing, and the repair kernel queued behind it
 * recomputes the chain with the same bits -- no trap, no wrong result, no host synchronisation.  flags: bit 0 = a census failed, bit 1 = a
 * time limit expired (since the process started); fusion_on = 0 once that has been seen by the host (gf_khop then runs one launch per hop)
 * or when GFHIP_MSWEEP_FUSE=0 is set in the environment. */
int gf_msweep_status(uint32_t* flags, int32_t* fusion_on);
int gf_time_khop(const gf_plan* const* plans, int32_t E, int32_t op, float* Z, int32_t B, int32_t W, int32_t K, int32_t iters,
                 void* stream, float* avg_ms);   /* the same for one gf_khop call (the K-1 hops of every edge feature on a tap stack) */
int gf_time_spmm_hop(const gf_plan* plan, int32_t op, const float* X_in, float* X_out, int32_t B, int32_t W,
                     int32_t iters, void* stream, float* avg_ms);

/* ---- tuning / debugging knobs (no reference counterpart; results are identical for every setting):
 * "spmm_bt" (0 = heuristic | 1 | 2 | 4), "spmm_spw" (0 = default | 1 | 2 | 4 slices per wave), "spmm_generic" (0/1),
 * "spmm_algo" (0 = default: the MFMA source sweep (gf_msweep.hip) where it applies, else the SELL-8 wave kernel | 1 = CSR workgroup
 * kernel | 3 = SELL-8 always | 5 = the MFMA source sweep or GF_ERR_UNSUPPORTED where it does not apply), the sweep's own knobs:
 * "spmm_fuse" (0/1 the K-1 hops of gf_khop in one launch), "spmm_bar" (0/1 XCD barrier between batch entries too), "spmm_pfd" (scalar
 * prefetch lead in loop iterations: 0 = auto from the image's rounds, -1 = off), "spmm_depth" (0 = 10 gathers in flight | 5), "spmm_slack" / "spmm_passes" (image: rounds
 * beyond the mean group length in percent, passes allowed per batch entry; read by gf_plan_create), timing-only: "spmm_srcmask",
 * "spmm_trace"; tests of the abandon-and-repair path: "spmm_census" (1 = census called bad | 2 = one workgroup claims the next XCC |
 * 3 = one workgroup never arrives), "spmm_tmo_ms" (time limit of census / barriers, 0 = 2000), "spmm_status_reset"; "spmm_xlayout" (0/1 the boundary layout pass
 * inside the fused chain launch), "spmm_hublim" (image: rows longer than this are hub rows, 0 = the builder's cost model; read by gf_plan_create); "spmm_minwork" (fewest (batch entry,
 * 32-column slab) pairs the sweep takes, default 5: below, SELL-8 is faster); "spmm_xcd" (0/1),
 * "spmm_group" (0/1 locality groups in the row schedule of graphs with N >= 32768; read by gf_plan_create),
 * "spmm_pf" (workgroups per tile prefetching the next gather panel, -1 = heuristic, 0 = off), "spmm_ucap" (0 | 8 | 16 gathers in flight per lane), "spmm_load" (0 = plain | 1 = non-temporal gather loads),
 * "spmm_store" (0 = plain | 1 = write-through sc1 | 2 = non-temporal output stores), "contract_generic" (0/1),
 * "pipeline" (0 = auto | 1 = node-major | 2 = column panels), "panel_uniform" (0/1 use the value-free stream),
 * "panel_order" (0/1 bank-aware neighbour order; read by gf_plan_create), "panel_sort" (0/1 octets sorted by longest row; read by gf_plan_create),
 * "panel_rotate" (0/1 per-workgroup rotated slice walk), "panel_np" (0 = heuristic | 1 | 2 panels per pass), "panel_split" (workgroups per pass
 * for small batches: 0 = as many as fit, 1 = off), "panel_chain" (0/1 the K-1 hops of a panel inside LDS, gf_chain.hip), "gradw_lds" (0/1),
 * "panel_db" (double-buffered per-hop panel kernel: 1 = where one workgroup fills a CU's LDS with four panels, 1280 <= N <= 2559 | 2 = wherever
 * four panels fit | 0 = never), "panel_thr" / "panel_loaders" (its workgroup size and loader waves, 0 = default), "spmm_lanes" (SELL kernels:
 * batch tiles in flight, 0 = one per XCD | 1 | 2 | 4),
 * "bwd_fuse" (0/1 dx and dh in one pass over the adjoint stack, either pipeline, when F <= 32 and G <= 32 or G = 64 / 128),
 * "bwd_fuse64" (0/1 the same for F = 64, G <= 32),
 * "evgf_generic" (0 = best EVGF tap kernel | 1 = one thread per output | 2 = LDS-staged 4-byte gathers), "evgf_idx16" (0/1 16-bit node
 * indices in the EVGF kernels when N <= 65535).
 * EXPERIMENTS ONLY: the knobs are process-global and not thread-safe, so gf_tune is refused (GF_ERR_UNSUPPORTED) unless the
 * environment variable GFHIP_EXPERIMENTS=1 was set when the library was loaded; without it the tuning state is the built-in
 * default and immutable, i.e. the product path reads no mutable global state.  Every setting gives identical results. */
int gf_tune(const char* key, int32_t value);
/* experiments: the phase time stamps of the last spmm_msweep_kernel launch made with gf_tune("spmm_trace", 1) -- out[8 XCDs][64 (entry, hop)
 * slots][8] shader-clock stamps of the XCD's first wave: {body start, round loop start, round loop end, stores issued, barrier passed, 0, 0, 0}
 * (tools/msweep_trace.py).  Synchronises the device. */
int gf_debug_msweep_trace(unsigned long long* out);
/* the sweep image of one orientation of a plan: out[8] = {accumulator sets per wave (0 = no image), passes, rounds, fill x 1000, hub rows (rows too long
 * for a group: computed by the waves between store phase and hand-over), hub rows summed as 32 partial chains, the two length limits} */
int gf_debug_msweep_info(const gf_plan* plan, int32_t op, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* GFHIP_H */
