// gf_plan.hip -- GSO ingest: host CSR(S_e) -> device plan (CSR of S^T and of S, degree-sorted row schedule).
// Replaces holding the dense [E,N,N] tensor handed to GraphFilter.addGSO (reference graphML.py:2116-2123):
// at N = 1e5 that tensor is 40 GB, the plan is ~25 MB.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unistd.h>
#include <set>
#include <utility>
#include <new>
#include <numeric>
#include <vector>

#include "gf_common.h"
#include "gf_msweep_image.h"

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

// The knobs exist for experiments (A/B timing, forcing a pipeline in tests).  They are process-global, so they are refused unless
// the process opted in through the environment BEFORE the library was loaded: without GFHIP_EXPERIMENTS=1 g_tune is the built-in
// default for the life of the process and the product path reads no mutable global state.
static const bool g_experiments = [] {
    const char* e = getenv("GFHIP_EXPERIMENTS");
    return e != nullptr && atoi(e) != 0;
}();

hipError_t gf_grant_lds(const void* kernel, size_t lds_bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> granted;
    if (lds_bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (granted.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) granted.insert({dev, kernel});
    return e;
}

int gf_require_no_static_lds(const void* kernel, const char* name) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> checked;
    int dev = 0;
    GF_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (checked.count({dev, kernel})) return GF_OK;
    hipFuncAttributes attr;
    GF_HIP(hipFuncGetAttributes(&attr, kernel));
    if (attr.sharedSizeBytes != 0) {
        gf_set_error("%s: the kernel holds %zu bytes of static LDS; its gathers address the dynamic panel from LDS address 0 (build error, not a data error)",
                     name, (size_t)attr.sharedSizeBytes);
        return GF_ERR_UNSUPPORTED;
    }
    checked.insert({dev, kernel});
    return GF_OK;
}

extern "C" int gf_tune(const char* key, int32_t value) {
    GF_REQUIRE_ARG(key != nullptr, "gf_tune: key is NULL");
    if (!g_experiments) {
        gf_set_error("gf_tune('%s'): tuning knobs are for experiments only; set GFHIP_EXPERIMENTS=1 before loading libgfhip.so", key);
        return GF_ERR_UNSUPPORTED;
    }
    if (!strcmp(key, "spmm_bt")) g_tune.spmm_bt = value;
    else if (!strcmp(key, "spmm_spw")) g_tune.spmm_spw = value;
    else if (!strcmp(key, "spmm_generic")) g_tune.spmm_generic = value;
    else if (!strcmp(key, "spmm_algo")) g_tune.spmm_algo = value;
    else if (!strcmp(key, "spmm_bar")) g_tune.spmm_bar = value;
    else if (!strcmp(key, "spmm_pfd")) g_tune.spmm_pfd = value;
    else if (!strcmp(key, "spmm_depth")) g_tune.spmm_depth = value;
    else if (!strcmp(key, "spmm_fuse")) g_tune.spmm_fuse = value;
    else if (!strcmp(key, "spmm_trace")) g_tune.spmm_trace = value;
    else if (!strcmp(key, "spmm_xlayout")) g_tune.spmm_xlayout = value;
    else if (!strcmp(key, "spmm_hublim")) g_tune.spmm_hublim = value;
    else if (!strcmp(key, "spmm_census")) g_tune.spmm_census = value;
    else if (!strcmp(key, "spmm_minwork")) g_tune.spmm_minwork = value < 1 ? 1 : value;
    else if (!strcmp(key, "spmm_tmo_ms")) g_tune.spmm_tmo_ms = value;
    else if (!strcmp(key, "spmm_status_reset")) gf_msweep_status_reset();
    else if (!strcmp(key, "spmm_passes")) g_tune.spmm_passes = value;
    else if (!strcmp(key, "spmm_srcmask")) g_tune.spmm_srcmask = value;
    else if (!strcmp(key, "spmm_slack")) g_tune.spmm_slack = value;
    else if (!strcmp(key, "spmm_xcd")) g_tune.spmm_xcd = value;
    else if (!strcmp(key, "spmm_store")) g_tune.spmm_store = value;
    else if (!strcmp(key, "spmm_load")) g_tune.spmm_load = value;
    else if (!strcmp(key, "spmm_pf")) g_tune.spmm_pf = value;
    else if (!strcmp(key, "spmm_group")) g_tune.spmm_group = value;
    else if (!strcmp(key, "spmm_ucap")) g_tune.spmm_ucap = value;
    else if (!strcmp(key, "contract_generic")) g_tune.contract_generic = value;
    else if (!strcmp(key, "pipeline")) g_tune.pipeline = value;
    else if (!strcmp(key, "gradw_lds")) g_tune.gradw_lds = value;
    else if (!strcmp(key, "panel_uniform")) g_tune.panel_uniform = value;
    else if (!strcmp(key, "panel_order")) g_tune.panel_order = value;
    else if (!strcmp(key, "panel_sort")) g_tune.panel_sort = value;
    else if (!strcmp(key, "panel_even")) g_tune.panel_even = value;
    else if (!strcmp(key, "panel_np")) g_tune.panel_np = value;
    else if (!strcmp(key, "spmm_lanes")) g_tune.spmm_lanes = value;
    else if (!strcmp(key, "panel_db")) g_tune.panel_db = value;
    else if (!strcmp(key, "panel_thr")) g_tune.panel_thr = value;
    else if (!strcmp(key, "panel_loaders")) g_tune.panel_loaders = value;
    else if (!strcmp(key, "panel_unit")) g_tune.panel_unit = value;
    else if (!strcmp(key, "panel_split")) g_tune.panel_split = value;
    else if (!strcmp(key, "bwd_fuse")) g_tune.bwd_fuse = value;
    else if (!strcmp(key, "bwd_fuse64")) g_tune.bwd_fuse64 = value;
    else if (!strcmp(key, "panel_grid")) g_tune.panel_grid = value;
    else if (!strcmp(key, "panel_rotate")) g_tune.panel_rotate = value;
    else if (!strcmp(key, "panel_chain")) g_tune.panel_chain = value;
    else if (!strcmp(key, "evgf_generic")) g_tune.evgf_generic = value;
    else if (!strcmp(key, "evgf_idx16")) g_tune.evgf_idx16 = value;
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

// Locality groups for graphs whose gather panel cannot live in an XCD's L2 (config 4: 12.8 MB of 128-byte rows against 4 MB, hit rate
// 0.31 = what an LRU gives a random order): rows that gather the same sources should run close in time.  The order in which rows are
// PROCESSED is free (every row writes its own output line), so no data is renumbered: rows are clustered -- recursive spectral
// bisection, then balanced label propagation -- and the schedule walks group after group.  An LRU of 32 768 rows over the resulting
// gather sequence of config 4's ER graph hits 0.44 instead of 0.32 (label propagation alone: 0.38 simulated, 0.37 measured); graphs
// with real community structure gain more.  Deterministic: fixed start vector, fixed sweep order, ties to the lowest group.
constexpr int32_t kGroupMinNodes = 32768;   // below this a 128-byte-row panel fits L2 anyway
constexpr int32_t kGroupRows = 6250;        // rows per group: their own 128-byte rows are 0.8 MB of the 4 MB L2
constexpr int kGroupPowerIters = 50;        // power-iteration steps per bisection level
constexpr int64_t kGroupWorkCap = 400'000'000;   // entry visits per bisection level: large graphs get fewer power iterations (>= 8)

std::vector<int32_t> locality_groups(int32_t n, const HostCsr& a, int32_t& P) {
    // Start: recursive spectral bisection.  Per level, kGroupPowerIters steps of power iteration on (d_max I - L) of the symmetrised
    // pattern restricted to each part, constant vector deflated -- not a converged Fiedler vector, but smooth enough: on config 4's ER
    // graph the 16 parts cut 0.71 of the entries (a random partition: 0.94) -- then a median split of every part.
    int levels = 1;
    while ((n >> (levels + 1)) >= kGroupRows * 3 / 4 && levels < 6) ++levels;
    P = 1 << levels;
    std::vector<int32_t> label(n, 0);
    {
        std::vector<double> x(n), y(n), dsum(n);
        std::vector<int32_t> order(n);
        for (int lev = 0; lev < levels; ++lev) {
            const int32_t parts = 1 << lev;
            std::fill(dsum.begin(), dsum.end(), 0.0);
            for (int32_t v = 0; v < n; ++v)
                for (int32_t q = a.rowptr[v]; q < a.rowptr[v + 1]; ++q) {
                    const int32_t u = a.col[q];
                    if (u != v && label[u] == label[v]) dsum[v] += 1.0, dsum[u] += 1.0;
                }
            std::vector<double> dmax(parts, 0.0), mean(parts), norm(parts);
            std::vector<int64_t> cntp(parts, 0);
            for (int32_t v = 0; v < n; ++v) dmax[label[v]] = std::max(dmax[label[v]], dsum[v]), cntp[label[v]]++;
            for (int32_t v = 0; v < n; ++v) x[v] = std::sin(0.7 * v + 0.3);
            auto deflate = [&](std::vector<double>& z) {
                std::fill(mean.begin(), mean.end(), 0.0);
                for (int32_t v = 0; v < n; ++v) mean[label[v]] += z[v];
                for (int32_t pp = 0; pp < parts; ++pp) mean[pp] /= (double)std::max<int64_t>(1, cntp[pp]);
                std::fill(norm.begin(), norm.end(), 0.0);
                for (int32_t v = 0; v < n; ++v) z[v] -= mean[label[v]], norm[label[v]] += z[v] * z[v];
                for (int32_t v = 0; v < n; ++v) z[v] /= std::sqrt(std::max(norm[label[v]], 1e-300));
            };
            deflate(x);
            const int iters = (int)std::max<int64_t>(8, std::min<int64_t>(kGroupPowerIters, kGroupWorkCap / std::max<int64_t>(1, (int64_t)a.col.size())));
            for (int it = 0; it < iters; ++it) {
                for (int32_t v = 0; v < n; ++v) y[v] = (dmax[label[v]] - dsum[v]) * x[v];
                for (int32_t v = 0; v < n; ++v)
                    for (int32_t q = a.rowptr[v]; q < a.rowptr[v + 1]; ++q) {
                        const int32_t u = a.col[q];
                        if (u != v && label[u] == label[v]) y[v] += x[u], y[u] += x[v];
                    }
                deflate(y);
                x.swap(y);
            }
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int32_t p0, int32_t p1) {
                if (label[p0] != label[p1]) return label[p0] < label[p1];
                return x[p0] < x[p1];
            });
            std::vector<int64_t> seen(parts, 0);
            for (int32_t i = 0; i < n; ++i) {
                const int32_t v = order[i], pp = label[v];
                const int32_t child = seen[pp]++ < cntp[pp] / 2 ? 0 : 1;
                y[v] = (double)(2 * pp + child);   // (labels are read by the sort above: written after the pass)
            }
            for (int32_t v = 0; v < n; ++v) label[v] = (int32_t)y[v];
        }
    }
    // Refinement: balanced label propagation -- a row moves to the group that holds most of its columns while that group has room.
    std::vector<int32_t> size(P, 0), cnt(P, 0);
    for (int32_t i = 0; i < n; ++i) size[label[i]]++;
    const int32_t cap = (int32_t)((int64_t)(n + P - 1) / P * 103 / 100) + 1;
    for (int sweep = 0; sweep < 6; ++sweep) {
        int64_t moved = 0;
        for (int32_t v = 0; v < n; ++v) {
            const int32_t lo = a.rowptr[v], hi = a.rowptr[v + 1];
            if (lo == hi) continue;
            for (int32_t q = lo; q < hi; ++q) cnt[label[a.col[q]]]++;
            const int32_t cur = label[v];
            int32_t best = cur;
            for (int32_t q = lo; q < hi; ++q) {
                const int32_t l = label[a.col[q]];
                if (cnt[l] > cnt[best] || (cnt[l] == cnt[best] && l < best && cnt[l] > cnt[cur])) best = l;
            }
            if (best != cur && cnt[best] > cnt[cur] && size[best] < cap) {
                label[v] = best;
                size[best]++;
                size[cur]--;
                ++moved;
            }
            for (int32_t q = lo; q < hi; ++q) cnt[label[a.col[q]]] = 0;
            cnt[cur] = 0;
        }
        if (moved == 0) break;
    }
    return label;
}

// The clustering costs ~2 s of one host core at N = 1e5 / nnz = 1e6 and depends on the sparsity pattern only.  It is therefore
// computed once per pattern and process (every device of a multi-GPU process, every edge feature with the same pattern, every
// deep copy / unpickled module re-creates plans from the same CSR), and -- when the caller names a directory in
// GFHIP_PLAN_CACHE_DIR -- once per pattern and machine (the ranks of a data-parallel job, repeated runs): labels are stored as
// <dir>/groups_<hash>_<n>_<nnz>.bin, written to a temporary name and renamed.
uint64_t pattern_hash(int32_t n, const HostCsr& a) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t bytes) {
        const unsigned char* c = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < bytes; ++i) h = (h ^ c[i]) * 1099511628211ull;
    };
    mix(&n, sizeof(n));
    mix(a.rowptr.data(), a.rowptr.size() * sizeof(int32_t));
    mix(a.col.data(), a.col.size() * sizeof(int32_t));
    return h;
}

std::vector<int32_t> locality_groups_cached(int32_t n, const HostCsr& a, int32_t& P) {
    struct Entry { uint64_t h; int32_t n; int64_t nnz; int32_t P; std::vector<int32_t> label; };
    static std::mutex mu;
    static std::vector<Entry> cache;   // newest last, at most 16 patterns
    const uint64_t h = pattern_hash(n, a);
    const int64_t nnz = (int64_t)a.col.size();
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Entry& e : cache)
            if (e.h == h && e.n == n && e.nnz == nnz) {
                P = e.P;
                return e.label;
            }
    }
    std::vector<int32_t> label;
    char path[1024] = "";
    if (const char* dir = getenv("GFHIP_PLAN_CACHE_DIR")) {
        snprintf(path, sizeof(path), "%s/groups_%016llx_%d_%lld.bin", dir, (unsigned long long)h, n, (long long)nnz);
        if (FILE* f = fopen(path, "rb")) {
            int32_t hdr[2] = {0, 0};
            label.resize(n);
            if (fread(hdr, sizeof(int32_t), 2, f) == 2 && hdr[0] == n && hdr[1] > 0 && fread(label.data(), sizeof(int32_t), n, f) == (size_t)n) {
                P = hdr[1];
                for (int32_t v = 0; v < n && !label.empty(); ++v)
                    if (label[v] < 0 || label[v] >= P) label.clear();   // a damaged file is recomputed, never trusted
            } else {
                label.clear();
            }
            fclose(f);
        }
    }
    if (label.empty()) {
        label = locality_groups(n, a, P);
        if (path[0]) {
            char tmp[1100];
            snprintf(tmp, sizeof(tmp), "%s.%d.tmp", path, (int)getpid());
            if (FILE* f = fopen(tmp, "wb")) {
                const int32_t hdr[2] = {n, P};
                const bool ok = fwrite(hdr, sizeof(int32_t), 2, f) == 2 && fwrite(label.data(), sizeof(int32_t), n, f) == (size_t)n;
                fclose(f);
                if (!ok || rename(tmp, path) != 0) remove(tmp);
            }
        }
    }
    std::lock_guard<std::mutex> lock(mu);
    if (cache.size() >= 16) cache.erase(cache.begin());
    cache.push_back(Entry{h, n, nnz, P, label});
    return label;
}

void schedule(int32_t n, const HostCsr& a, bool sorted, const std::vector<int32_t>* groups, HostCsr& s, std::vector<int32_t>& rowid,
              int32_t& max_deg) {
    rowid.resize(n);
    std::iota(rowid.begin(), rowid.end(), 0);
    if (sorted && groups && (int32_t)groups->size() == n) {
        const std::vector<int32_t>& label = *groups;
        std::stable_sort(rowid.begin(), rowid.end(), [&](int32_t x, int32_t y) {
            if (label[x] != label[y]) return label[x] < label[y];
            return (a.rowptr[x + 1] - a.rowptr[x]) > (a.rowptr[y + 1] - a.rowptr[y]);
        });
    } else if (sorted) {
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

int upload_panel(int32_t n, const HostCsr& a, gf_csr_dev& d, int64_t& bytes);

int upload_csr(int32_t n, const HostCsr& a, bool sorted, const std::vector<int32_t>* groups, gf_csr_dev& d, int64_t& bytes) {
    HostCsr s;
    std::vector<int32_t> rowid;
    schedule(n, a, sorted, groups, s, rowid, d.max_deg);
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
    bool uni = !s.val.empty();
    for (size_t q = 1; uni && q < s.val.size(); ++q) uni = s.val[q] == s.val[0];
    d.sell_uniform = uni ? 1 : 0;
    d.sell_uval = uni ? s.val[0] : 0.f;
    if (uni) {
        std::vector<int32_t> colv(ent.size(), -1);
        for (int32_t sl = 0; sl < ns; ++sl)
            for (int32_t r = 0; r < 8 && sl * 8 + r < n; ++r) {
                const int32_t p = sl * 8 + r;
                for (int32_t q = s.rowptr[p]; q < s.rowptr[p + 1]; ++q) colv[((size_t)kptr[sl] + (q - s.rowptr[p])) * 8 + r] = s.col[q];
            }
        if ((rc = upload(colv, &d.sell_col, bytes))) return rc;
    }
    // MSWEEP image (gf_msweep_image.h): graphs whose gather panel does not fit L2, when the groups balance (no hub rows).  A product
    // process builds it only where the default hop uses it (kMsDefaultMinNodes); between kMsMinNodes and there the kernel can only be
    // asked for through gf_tune (spmm_algo = 5), i.e. in GFHIP_EXPERIMENTS=1 processes -- only those pay for that image.
    if (n >= (g_experiments ? kMsMinNodes : kMsDefaultMinNodes)) {
        MsweepImage ms = build_msweep_image(n, a.rowptr.data(), a.col.data(), a.val.data(), uni, g_tune.spmm_slack, g_tune.spmm_passes > 0 ? g_tune.spmm_passes : 1, g_tune.spmm_hublim);
        d.ms_fill = ms.fill();
        if (ms.passes >= 1 && ms.passes <= 2 && ms.fill() >= 0.6 && ms.hub_entries * 5 <= (int64_t)a.rowptr[n] * 2) {   // (hub rows: at most 40 % of the entries)
            d.ms_hub_rows = ms.hub_rows;
            d.ms_hub_split_rows = ms.hub_split_rows;
            d.ms_hub_limit = ms.hub_limit;
            d.ms_hub_split = ms.hub_split;
            d.ms_hub_entries = ms.hub_entries;
            d.ms_sets = ms.sets;
            d.ms_passes = ms.passes;
            d.ms_rounds = ms.rounds;
            d.ms_uniform = uni ? 1 : 0;
            if ((rc = upload(ms.ent, &d.ms_ent, bytes))) return rc;
            if (!uni && (rc = upload(ms.val, &d.ms_val, bytes))) return rc;
            if ((rc = upload(ms.rows, &d.ms_rows, bytes))) return rc;
            if (ms.hub_rows) {   // hub rows: one packed buffer {block offsets, total words, blocks, values}; the kernel's loads run up to four steps ahead: padded
                const size_t hw = ms.hub.size() + 192;
                std::vector<uint32_t> pack(ms.hubptr);
                pack.push_back((uint32_t)hw);
                pack.insert(pack.end(), ms.hub.begin(), ms.hub.end());
                pack.resize(pack.size() + 192, kMsPad);
                if (!uni) {
                    const size_t at = pack.size();
                    pack.resize(at + hw, 0u);
                    memcpy(pack.data() + at, ms.hubval.data(), ms.hubval.size() * sizeof(float));
                }
                if ((rc = upload(pack, &d.ms_hub, bytes))) return rc;
            }
            GF_HIP(hipMalloc((void**)&d.ms_gate, gf_msweep_gate_bytes()));
            GF_HIP(hipMemset(d.ms_gate, 0, gf_msweep_gate_bytes()));
            (void)gf_msweep_status_word();   // (pinned word for the repair kernel's report: allocated here, never under a launch)
            bytes += (int64_t)gf_msweep_gate_bytes();
        }
    }
    return upload_panel(n, a, d, bytes);
}

// ---- panel (LDS) image: natural row order, slices of 64 rows, steps compacted (see gf_common.h) -------------------------
// ds_read_b128 is serviced in four 16-lane groups; within a group, lanes whose 16-byte slots fall on the same bank quad
// (column mod 16) with different addresses serialise.  The order in which a row visits its neighbours is free (it only
// fixes the summation order), so step by step each lane picks, among its remaining neighbours, one whose quad is still
// unused in its group (or an address another lane already reads: broadcast).  Greedy, most-constrained lanes first.
const int kB128Group[64] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
                            2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3};

// ELL block of up to 64 rows (lane l walks row rows[l]; rows[l] >= n: no row): appends gmax = max(min_groups, ceil(longest / 4))
// group-rows of 64 lanes to the three entry streams; empty slots = {column N = the LDS zero slot, value 0}.
struct EllStreams {
    std::vector<uint4> col4;   // [group-row][lane] -> 4 LDS byte offsets (column * 16) of steps 4j .. 4j+3 (value-free stream)
    std::vector<uint2> col2;   // the same as 4 x 16-bit columns (weighted stream: 8 + 16 bytes per group instead of 16 + 16)
    std::vector<float4> val4;
    double cycles = 0.0;       // modelled LDS cycles of the ds_read_b128 steps
    int64_t steps = 0, slots = 0;
    int32_t group_rows() const { return (int32_t)(col4.size() / 64); }
    void sentinel_rows(int32_t n, int count) {
        const uint32_t nn = (uint32_t)n * 16u;
        for (int i = 0; i < count * 64; ++i) {
            col4.push_back(make_uint4(nn, nn, nn, nn));
            col2.push_back(make_uint2((uint32_t)n | ((uint32_t)n << 16), (uint32_t)n | ((uint32_t)n << 16)));
            val4.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
};

int32_t emit_ell_block(int32_t n, const HostCsr& a, const int32_t* rows, bool reorder, bool even, int32_t min_groups, EllStreams& o) {
    std::vector<int32_t> rest[64];  // remaining entry indices (into a.col / a.val) per lane
    int32_t w = 0;
    for (int l = 0; l < 64; ++l) {
        const int32_t r = rows[l];
        if (r >= 0 && r < n) {
            for (int32_t q = a.rowptr[r]; q < a.rowptr[r + 1]; ++q) rest[l].push_back(q);
            w = std::max<int32_t>(w, (int32_t)rest[l].size());
        }
    }
    int32_t gmax = even ? ((((w + 3) / 4) + 1) & ~1) : (w + 3) / 4;  // (even: a two-group round never straddles blocks)
    gmax = std::max(gmax, min_groups);
    std::vector<int32_t> picks((size_t)gmax * 4 * 64, -1);  // [step][lane] chosen entry (-1 = row exhausted)
    for (int32_t k = 0; k < w; ++k) {
        int quadCols[4][16][4];  // per service group, per bank quad: distinct columns already read (up to 4 tracked)
        int quadCnt[4][16];
        for (int g = 0; g < 4; ++g)
            for (int qd = 0; qd < 16; ++qd) quadCnt[g][qd] = 0;
        int order[64], no = 0;
        for (int l = 0; l < 64; ++l)
            if (!rest[l].empty()) order[no++] = l;
        if (reorder)  // most-constrained first: lanes with the fewest remaining neighbours choose first
            std::stable_sort(order, order + no, [&](int x, int y) { return rest[x].size() < rest[y].size(); });
        for (int oi = 0; oi < no; ++oi) {
            const int l = order[oi], g = kB128Group[l];
            size_t best = 0;
            if (reorder) {
                int bestCost = 1 << 30;
                for (size_t j = 0; j < rest[l].size(); ++j) {
                    const int c = a.col[rest[l][j]], qd = c & 15;
                    int cost = quadCnt[g][qd];
                    for (int t = 0; t < quadCnt[g][qd] && t < 4; ++t)
                        if (quadCols[g][qd][t] == c) cost = 0;  // same address: broadcast
                    if (cost < bestCost) {
                        bestCost = cost;
                        best = j;
                        if (cost == 0) break;
                    }
                }
            }
            const int32_t q = rest[l][best];
            rest[l].erase(rest[l].begin() + best);
            picks[(size_t)k * 64 + l] = q;
            const int c = a.col[q], qd = c & 15;
            bool seen = false;
            for (int t = 0; t < quadCnt[g][qd] && t < 4; ++t) seen = seen || quadCols[g][qd][t] == c;
            if (!seen) {
                if (quadCnt[g][qd] < 4) quadCols[g][qd][quadCnt[g][qd]] = c;
                quadCnt[g][qd]++;
            }
        }
        for (int g = 0; g < 4; ++g) {
            int mx = 1;
            for (int qd = 0; qd < 16; ++qd) mx = std::max(mx, quadCnt[g][qd]);
            o.cycles += mx;
        }
        o.steps++;
    }
    for (int32_t j = 0; j < gmax; ++j)
        for (int l = 0; l < 64; ++l) {
            uint32_t c[4];
            float v[4];
            for (int i = 0; i < 4; ++i) {
                const int32_t q = picks[(size_t)(4 * j + i) * 64 + l];
                c[i] = (uint32_t)(q >= 0 ? a.col[q] : n) * 16u;
                v[i] = q >= 0 ? a.val[q] : 0.f;
            }
            o.col4.push_back(make_uint4(c[0], c[1], c[2], c[3]));
            o.col2.push_back(make_uint2((c[0] >> 4) | ((c[1] >> 4) << 16), (c[2] >> 4) | ((c[3] >> 4) << 16)));
            o.val4.push_back(make_float4(v[0], v[1], v[2], v[3]));
        }
    o.slots += (int64_t)gmax * 4 * 64;
    return gmax;
}

int upload_chain(int32_t n, const HostCsr& a, gf_csr_dev& d, int64_t& bytes);

int upload_panel(int32_t n, const HostCsr& a, gf_csr_dev& d, int64_t& bytes) {
    if (n > kPanelMaxNodes) return GF_OK;
    for (int32_t i = 0; i < n; ++i)
        if (a.rowptr[i + 1] - a.rowptr[i] > kPanelMaxDeg) return GF_OK;  // hub row longer than a 16-bit count: no panel image
    auto rdeg = [&](int32_t r) { return r < n ? a.rowptr[r + 1] - a.rowptr[r] : 0; };
    // Octets (8 consecutive rows = one 128-byte line of a panel) are the unit of work assignment: sorted by their longest
    // row, so that the 8 octets of a slice need about the same number of steps, while every octet still stores one full line.
    const int ush = (g_tune.panel_unit == 4) ? 2 : (g_tune.panel_unit == 2 ? 1 : 3);  // rows per unit = 8 (full line) | 4 | 2
    const int U = 1 << ush, UPS = 64 >> ush;                                          // rows per unit, units per slice
    const int32_t noct = (n + U - 1) / U;
    const int32_t ns = (noct + UPS - 1) / UPS;
    std::vector<int32_t> oct((size_t)ns * UPS, -1), omax(noct, 0);
    for (int32_t o = 0; o < noct; ++o) {
        oct[o] = o;
        for (int r = 0; r < U; ++r) omax[o] = std::max(omax[o], rdeg(o * U + r));
    }
    if (g_tune.panel_sort)
        std::stable_sort(oct.begin(), oct.begin() + noct, [&](int32_t x, int32_t y) { return omax[x] > omax[y]; });
    std::vector<int2> slice(ns);
    EllStreams ell;
    const bool reorder = g_tune.panel_order != 0;
    for (int32_t sl = 0; sl < ns; ++sl) {
        int32_t rows[64];
        for (int l = 0; l < 64; ++l) {
            const int32_t o = oct[(size_t)sl * UPS + (l >> ush)];
            rows[l] = o >= 0 ? o * U + (l & (U - 1)) : n;
        }
        const int32_t g0 = ell.group_rows();
        const int32_t gmax = emit_ell_block(n, a, rows, reorder, g_tune.panel_even != 0, 0, ell);
        slice[sl] = make_int2(g0, gmax);   // group-row offset (x 64 lanes), group-steps
    }
    d.pn_uniform = 0;
    d.pn_uval = 0.f;
    if (!a.val.empty()) {
        bool uni = true;
        for (size_t i = 1; i < a.val.size() && uni; ++i) uni = std::memcmp(&a.val[i], &a.val[0], 4) == 0;
        d.pn_uniform = uni ? 1 : 0;
        d.pn_uval = a.val[0];
    }
    d.pn_conflict = ell.steps ? ell.cycles / (double)ell.steps : 0.0;
    d.pn_fill = ell.slots ? (double)a.col.size() / (double)ell.slots : 1.0;
    d.pn_sentinel = ell.group_rows();   // two all-sentinel group-rows: what a wave without further work requests
    ell.sentinel_rows(n, 2);
    int rc;
    if ((rc = upload(slice, &d.pn_slice, bytes))) return rc;
    if ((rc = upload(oct, &d.pn_oct, bytes))) return rc;
    if (d.pn_uniform && (rc = upload(ell.col4, &d.pn_col4, bytes))) return rc;
    if ((rc = upload(ell.col2, &d.pn_col2, bytes))) return rc;   // kept for uniform plans too (knob panel_uniform = 0)
    if ((rc = upload(ell.val4, &d.pn_val4, bytes))) return rc;
    d.pn_slices = ns;
    d.pn_ushift = ush;
    return upload_chain(n, a, d, bytes);
}

// ---- chain image (gf_chain.hip; layout described at gf_csr_dev::cn_*) --------------------------------------------------
int upload_chain(int32_t n, const HostCsr& a, gf_csr_dev& d, int64_t& bytes) {
    const int32_t nChunks = (n + 63) / 64;
    // two panels per pass while both fit the LDS: the accumulators of a hop are sets x panels x 4 registers per lane, so a pair
    // halves the sets a wave may take (more, shorter-lived waves for the same rows)
    const int np = (2 * (size_t)(n + 1) * 16 <= 160 * 1024) ? 2 : 1;
    const int maxSets = (kChainSets / np) & ~1;
    static const int kGatherWaves[5] = {1, 2, 4, 8, kChainBigW};  // + the storer wave(s): workgroups of 128 ... 1024 threads
    int W = kChainBigW;
    for (int i = 4; i >= 0; --i)
        if ((nChunks + kGatherWaves[i] - 1) / kGatherWaves[i] <= maxSets) W = kGatherWaves[i];
    const int R = (nChunks + W - 1) / W;
    if (R > maxSets) return GF_OK;  // cannot happen for n <= kPanelMaxNodes
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        return (a.rowptr[x + 1] - a.rowptr[x]) > (a.rowptr[y + 1] - a.rowptr[y]);
    });
    const int T = W * 64;
    std::vector<uint32_t> rowoff((size_t)R * T, 0xffffffffu);
    std::vector<int32_t> gtab((size_t)W * 32, 0);
    EllStreams ell;
    const bool reorder = g_tune.panel_order != 0;
    for (int w = 0; w < W; ++w) {
        gtab[(size_t)w * 32] = ell.group_rows() / 2;
        for (int r = 0; r < R; ++r) {
            const int32_t c = r * W + ((r & 1) ? W - 1 - w : w);  // chunk: sorted rows 64c .. 64c+63, dealt boustrophedon
            int32_t rows[64];
            for (int l = 0; l < 64; ++l) {
                const int64_t pos = (int64_t)c * 64 + l;
                rows[l] = (c < nChunks && pos < n) ? order[pos] : n;
                if (rows[l] < n) rowoff[(size_t)r * T + w * 64 + l] = (uint32_t)rows[l] * 16u;
            }
            // every block has at least one group-row: the kernel commits a set when its stream position reaches the block's end
            const int32_t gr = emit_ell_block(n, a, rows, reorder, false, 1, ell);
            if (gr & 1) ell.sentinel_rows(n, 1);  // storage padding to a whole word; flagged in the table, never gathered
            gtab[(size_t)w * 32 + 1 + r] = ell.group_rows() / 2;
            gtab[(size_t)w * 32 + 16 + r] = gr & 1;
        }
    }
    ell.sentinel_rows(n, 4);  // the two-word prefetch of the last wave runs past its stream
    std::vector<uint4> col8(ell.col2.size() / 2);  // [word][lane] = {group-row 2u, group-row 2u + 1}
    for (size_t u = 0; u < col8.size() / 64; ++u)
        for (int l = 0; l < 64; ++l) {
            const uint2 lo = ell.col2[(2 * u) * 64 + l], hi = ell.col2[(2 * u + 1) * 64 + l];
            col8[u * 64 + l] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    d.cn_fill = ell.slots ? (double)a.col.size() / (double)ell.slots : 1.0;
    d.cn_conflict = ell.steps ? ell.cycles / (double)ell.steps : 0.0;
    int rc;
    if ((rc = upload(rowoff, &d.cn_rowoff, bytes))) return rc;
    if ((rc = upload(gtab, &d.cn_gtab, bytes))) return rc;
    if ((rc = upload(col8, &d.cn_col8, bytes))) return rc;
    if ((rc = upload(ell.val4, &d.cn_val4, bytes))) return rc;   // kept for uniform plans too (knob panel_uniform = 0)
    d.cn_waves = W;
    d.cn_sets = R;
    d.cn_np = np;
    return GF_OK;
}

void free_csr(gf_csr_dev& d) {
    if (d.rowptr) (void)hipFree(d.rowptr);
    if (d.col) (void)hipFree(d.col);
    if (d.val) (void)hipFree(d.val);
    if (d.rowid) (void)hipFree(d.rowid);
    if (d.sell_kptr) (void)hipFree(d.sell_kptr);
    if (d.sell_ent) (void)hipFree(d.sell_ent);
    if (d.sell_col) (void)hipFree(d.sell_col);
    if (d.sell_rowid) (void)hipFree(d.sell_rowid);
    if (d.ms_ent) (void)hipFree(d.ms_ent);
    if (d.ms_val) (void)hipFree(d.ms_val);
    if (d.ms_rows) (void)hipFree(d.ms_rows);
    if (d.ms_hub) (void)hipFree(d.ms_hub);
    if (d.ms_gate) (void)hipFree(d.ms_gate);
    if (d.pn_slice) (void)hipFree(d.pn_slice);
    if (d.pn_oct) (void)hipFree(d.pn_oct);
    if (d.pn_col4) (void)hipFree(d.pn_col4);
    if (d.pn_col2) (void)hipFree(d.pn_col2);
    if (d.pn_val4) (void)hipFree(d.pn_val4);
    if (d.cn_rowoff) (void)hipFree(d.cn_rowoff);
    if (d.cn_gtab) (void)hipFree(d.cn_gtab);
    if (d.cn_col8) (void)hipFree(d.cn_col8);
    if (d.cn_val4) (void)hipFree(d.cn_val4);
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
        // one set of locality groups for both orientations (the spectral start symmetrises the pattern anyway)
        std::vector<int32_t> groups;
        if (sorted && n >= kGroupMinNodes && g_tune.spmm_group) {
            int32_t P = 0;
            groups = locality_groups_cached(n, S, P);
        }
        const std::vector<int32_t>* gp = groups.empty() ? nullptr : &groups;
        int rc = upload_csr(n, St, sorted, gp, pl->mat[GF_OP_FWD], pl->device_bytes);
        if (rc == GF_OK) rc = upload_csr(n, S, sorted, gp, pl->mat[GF_OP_BWD], pl->device_bytes);
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

extern "C" int gf_plan_panel_info(const gf_plan* pl, int32_t op, int32_t* n_slices, int32_t* uniform, double* lds_cycles_per_step, double* fill) {
    GF_REQUIRE_ARG(pl != nullptr && (op == GF_OP_FWD || op == GF_OP_BWD), "gf_plan_panel_info: bad plan / op");
    const gf_csr_dev& m = pl->mat[op];
    if (n_slices) *n_slices = m.pn_slices;
    if (uniform) *uniform = m.pn_uniform;
    if (lds_cycles_per_step) *lds_cycles_per_step = m.pn_conflict;
    if (fill) *fill = m.pn_fill;
    return GF_OK;
}

extern "C" int gf_plan_info(const gf_plan* pl, int32_t* n, int64_t* nnz, int64_t* device_bytes) {
    GF_REQUIRE_ARG(pl != nullptr, "gf_plan_info: plan is NULL");
    if (n) *n = pl->n;
    if (nnz) *nnz = pl->nnz;
    if (device_bytes) *device_bytes = pl->device_bytes;
    return GF_OK;
}
