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
};

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
    int spmm_algo = 0;          // 0 = SELL-8 persistent wave kernel, 1 = CSR workgroup-staged kernel (first version)
    int spmm_xcd = 1;           // 1 = XCD-aware tile order
    int spmm_ucap = 0;          // 0/16 = up to 16 gathers in flight per lane, 8 = up to 8 (fewer registers, more waves)
    int spmm_pf = 0;            // workgroups per tile that prefetch the next tile's gather panel into L2 (0 = off)
    int spmm_load = 0;          // gather loads: 0 = plain, 1 = non-temporal
    int spmm_store = 1;         // output rows: 0 = plain stores, 1 = write-through (sc1), 2 = non-temporal
    int contract_generic = 0;   // 1 = force the generic contraction kernel
};
extern gf_tuning g_tune;

// internal launchers shared between translation units
int gf_contract_launch(const float* Z, const float* h, const float* bias, float* out, int B, int N, int Nout, int G,
                       int F, int E, int K, int transpose_bank, hipStream_t st);
