// gf_plan.hip -- GSO ingest: host CSR(S_e) -> device plan (CSR of S^T and of S, degree-sorted row schedule).
// Replaces holding the dense [E,N,N] tensor handed to GraphFilter.addGSO (reference graphML.py:2116-2123):
// at N = 1e5 that tensor is 40 GB, the plan is ~25 MB.
#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <new>
#include <numeric>
#include <vector>

#include "gf_common.h"

// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void gf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gf_last_error(void) { return g_err; }

gf_tuning g_tune;

extern "C" int gf_tune(const char* key, int32_t value) {
    GF_REQUIRE_ARG(key != nullptr, "gf_tune: key is NULL");
    if (!strcmp(key, "spmm_bt")) g_tune.spmm_bt = value;
    else if (!strcmp(key, "spmm_spw")) g_tune.spmm_spw = value;
    else if (!strcmp(key, "spmm_generic")) g_tune.spmm_generic = value;
    else if (!strcmp(key, "spmm_algo")) g_tune.spmm_algo = value;
    else if (!strcmp(key, "spmm_xcd")) g_tune.spmm_xcd = value;
    else if (!strcmp(key, "spmm_store")) g_tune.spmm_store = value;
    else if (!strcmp(key, "spmm_load")) g_tune.spmm_load = value;
    else if (!strcmp(key, "spmm_pf")) g_tune.spmm_pf = value;
    else if (!strcmp(key, "spmm_ucap")) g_tune.spmm_ucap = value;
    else if (!strcmp(key, "contract_generic")) g_tune.contract_generic = value;
    else {
        gf_set_error("gf_tune: unknown key '%s'", key);
        return GF_ERR_ARG;
    }
    return GF_OK;
}
extern "C" int gf_version(void) { return GFHIP_VERSION; }

// ---------------------------------------------------------------------------------------------------
namespace {

struct HostCsr {
    std::vector<int32_t> rowptr, col;
    std::vector<float> val;
};

// rows sorted by column, duplicates summed in input order
HostCsr canonicalize(int32_t n, const int32_t* rp, const int32_t* ci, const void* vals, bool f64) {
    HostCsr out;
    out.rowptr.assign(n + 1, 0);
    out.col.reserve(rp[n]);
    out.val.reserve(rp[n]);
    std::vector<int32_t> order;
    for (int32_t i = 0; i < n; ++i) {
        const int32_t lo = rp[i], hi = rp[i + 1];
        order.resize(hi - lo);
        std::iota(order.begin(), order.end(), lo);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return ci[a] < ci[b]; });
        int32_t last = -1;
        for (int32_t p : order) {
            const float v = f64 ? (float)((const double*)vals)[p] : ((const float*)vals)[p];
            if (ci[p] == last) {
                out.val.back() += v;
            } else {
                out.col.push_back(ci[p]);
                out.val.push_back(v);
                last = ci[p];
            }
        }
        out.rowptr[i + 1] = (int32_t)out.col.size();
    }
    return out;
}

HostCsr transpose(int32_t n, const HostCsr& a) {
    HostCsr t;
    const int32_t nnz = a.rowptr[n];
    t.rowptr.assign(n + 1, 0);
    t.col.resize(nnz);
    t.val.resize(nnz);
    for (int32_t p = 0; p < nnz; ++p) t.rowptr[a.col[p] + 1]++;
    for (int32_t i = 0; i < n; ++i) t.rowptr[i + 1] += t.rowptr[i];
    std::vector<int32_t> cursor(t.rowptr.begin(), t.rowptr.end() - 1);
    for (int32_t i = 0; i < n; ++i)
        for (int32_t p = a.rowptr[i]; p < a.rowptr[i + 1]; ++p) {
            const int32_t q = cursor[a.col[p]]++;
            t.col[q] = i;  // ascending i within each transposed row
            t.val[q] = a.val[p];
        }
    return t;
}

// Row schedule: within windows of `window` consecutive rows, order rows by descending degree so that the
// rows one workgroup (and one wavefront) walks together have near-equal length; windows keep whatever
// coarse locality the node numbering has.  Returns the permuted CSR + rowid (stored position -> row).
constexpr int32_t kScheduleWindow = 8192;

void schedule(int32_t n, const HostCsr& a, bool sorted, HostCsr& s, std::vector<int32_t>& rowid, int32_t& max_deg) {
    rowid.resize(n);
    std::iota(rowid.begin(), rowid.end(), 0);
    if (sorted) {
        for (int32_t w0 = 0; w0 < n; w0 += kScheduleWindow) {
            const int32_t w1 = std::min(n, w0 + kScheduleWindow);
            std::stable_sort(rowid.begin() + w0, rowid.begin() + w1, [&](int32_t x, int32_t y) {
                return (a.rowptr[x + 1] - a.rowptr[x]) > (a.rowptr[y + 1] - a.rowptr[y]);
            });
        }
    }
    s.rowptr.assign(n + 1, 0);
    s.col.resize(a.col.size());
    s.val.resize(a.val.size());
    max_deg = 0;
    int32_t q = 0;
    for (int32_t p = 0; p < n; ++p) {
        const int32_t r = rowid[p];
        const int32_t lo = a.rowptr[r], hi = a.rowptr[r + 1];
        max_deg = std::max(max_deg, hi - lo);
        std::copy(a.col.begin() + lo, a.col.begin() + hi, s.col.begin() + q);
        std::copy(a.val.begin() + lo, a.val.begin() + hi, s.val.begin() + q);
        q += hi - lo;
        s.rowptr[p + 1] = q;
    }
}

template <class T>
int upload(const std::vector<T>& h, T** d, int64_t& bytes) {
    const size_t nb = std::max<size_t>(h.size(), 1) * sizeof(T);
    GF_HIP(hipMalloc((void**)d, nb));
    if (!h.empty()) GF_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    bytes += (int64_t)nb;
    return GF_OK;
}

int upload_csr(int32_t n, const HostCsr& a, bool sorted, gf_csr_dev& d, int64_t& bytes) {
    HostCsr s;
    std::vector<int32_t> rowid;
    schedule(n, a, sorted, s, rowid, d.max_deg);
    int rc;
    if ((rc = upload(s.rowptr, &d.rowptr, bytes))) return rc;
    if ((rc = upload(s.col, &d.col, bytes))) return rc;
    if ((rc = upload(s.val, &d.val, bytes))) return rc;
    if ((rc = upload(rowid, &d.rowid, bytes))) return rc;

    // SELL-8: slices of 8 consecutive scheduled rows; the degree-sorted schedule keeps padding small.
    const int32_t ns = (n + 7) / 8;
    std::vector<int32_t> kptr(ns + 1, 0), rid(ns * 8, -1);
    for (int32_t sl = 0; sl < ns; ++sl) {
        int32_t w = 0;
        for (int32_t r = 0; r < 8 && sl * 8 + r < n; ++r) w = std::max(w, s.rowptr[sl * 8 + r + 1] - s.rowptr[sl * 8 + r]);
        kptr[sl + 1] = kptr[sl] + w;
    }
    std::vector<int2> ent((size_t)kptr[ns] * 8, make_int2(0, 0));
    for (int32_t sl = 0; sl < ns; ++sl)
        for (int32_t r = 0; r < 8 && sl * 8 + r < n; ++r) {
            const int32_t p = sl * 8 + r;
            rid[p] = rowid[p];
            for (int32_t q = s.rowptr[p]; q < s.rowptr[p + 1]; ++q) {
                int32_t bits;
                std::memcpy(&bits, &s.val[q], 4);
                ent[((size_t)kptr[sl] + (q - s.rowptr[p])) * 8 + r] = make_int2(s.col[q], bits);
            }
        }
    d.n_slices = ns;
    d.sell_pad_entries = (int64_t)kptr[ns] * 8 - (int64_t)s.col.size();
    if ((rc = upload(kptr, &d.sell_kptr, bytes))) return rc;
    if ((rc = upload(ent, &d.sell_ent, bytes))) return rc;
    if ((rc = upload(rid, &d.sell_rowid, bytes))) return rc;
    return GF_OK;
}

void free_csr(gf_csr_dev& d) {
    if (d.rowptr) (void)hipFree(d.rowptr);
    if (d.col) (void)hipFree(d.col);
    if (d.val) (void)hipFree(d.val);
    if (d.rowid) (void)hipFree(d.rowid);
    if (d.sell_kptr) (void)hipFree(d.sell_kptr);
    if (d.sell_ent) (void)hipFree(d.sell_ent);
    if (d.sell_rowid) (void)hipFree(d.sell_rowid);
    d = gf_csr_dev{};
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C" int gf_plan_create(int32_t n, int64_t nnz, const int32_t* rowptr, const int32_t* colidx, const void* vals,
                              int32_t vals_is_f64, uint32_t flags, gf_plan** out) {
    GF_REQUIRE_ARG(out != nullptr, "gf_plan_create: out_plan is NULL");
    *out = nullptr;
    GF_REQUIRE_ARG(rowptr && (nnz == 0 || (colidx && vals)), "gf_plan_create: NULL CSR array");
    GF_REQUIRE_SHAPE(n > 0, "gf_plan_create: n_nodes = %d must be positive", n);
    GF_REQUIRE_SHAPE(nnz >= 0 && nnz < (int64_t)INT32_MAX, "gf_plan_create: nnz = %lld outside [0, 2^31)", (long long)nnz);
    GF_REQUIRE_SHAPE(rowptr[0] == 0 && rowptr[n] == nnz, "gf_plan_create: rowptr[0] = %d, rowptr[N] = %d, nnz = %lld",
                     rowptr[0], rowptr[n], (long long)nnz);
    for (int32_t i = 0; i < n; ++i)
        GF_REQUIRE_SHAPE(rowptr[i] <= rowptr[i + 1], "gf_plan_create: rowptr not monotone at row %d", i);
    for (int64_t p = 0; p < nnz; ++p)
        GF_REQUIRE_SHAPE(colidx[p] >= 0 && colidx[p] < n, "gf_plan_create: column index %d at %lld outside [0, %d)",
                         colidx[p], (long long)p, n);

    gf_plan* pl = new (std::nothrow) gf_plan();
    if (!pl) {
        gf_set_error("gf_plan_create: out of host memory");
        return GF_ERR_NOMEM;
    }
    const bool sorted = !(flags & 1u);
    try {
        HostCsr S = canonicalize(n, rowptr, colidx, vals, vals_is_f64 != 0);
        HostCsr St = transpose(n, S);
        pl->n = n;
        pl->nnz = (int64_t)S.col.size();
        int rc = upload_csr(n, St, sorted, pl->mat[GF_OP_FWD], pl->device_bytes);
        if (rc == GF_OK) rc = upload_csr(n, S, sorted, pl->mat[GF_OP_BWD], pl->device_bytes);
        if (rc != GF_OK) {
            gf_plan_destroy(pl);
            return rc;
        }
    } catch (const std::bad_alloc&) {
        gf_plan_destroy(pl);
        gf_set_error("gf_plan_create: out of host memory");
        return GF_ERR_NOMEM;
    }
    *out = pl;
    return GF_OK;
}

extern "C" int gf_plan_destroy(gf_plan* pl) {
    if (!pl) return GF_OK;
    free_csr(pl->mat[0]);
    free_csr(pl->mat[1]);
    delete pl;
    return GF_OK;
}

extern "C" int gf_plan_info(const gf_plan* pl, int32_t* n, int64_t* nnz, int64_t* device_bytes) {
    GF_REQUIRE_ARG(pl != nullptr, "gf_plan_info: plan is NULL");
    if (n) *n = pl->n;
    if (nnz) *nnz = pl->nnz;
    if (device_bytes) *device_bytes = pl->device_bytes;
    return GF_OK;
}
