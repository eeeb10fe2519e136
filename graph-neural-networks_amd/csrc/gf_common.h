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

// internal launchers shared between translation units
int gf_contract_launch(const float* Z, const float* h, const float* bias, float* out, int B, int N, int Nout, int G,
                       int F, int E, int K, int transpose_bank, hipStream_t st);
