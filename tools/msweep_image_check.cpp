// msweep_image_check.cpp -- CPU check of the MSWEEP image (graph-neural-networks_amd/csrc/gf_msweep_image.h): builds it for random graphs,
// interprets it the way spmm_msweep_kernel does and compares BITWISE with the row-by-row fmaf sums in ascending column order
// (graphML.py:158-161 per batch entry); prints the fill and the lock-step LRU hit rate of the gathers.
//   g++ -O2 -std=c++17 -Igraph-neural-networks_amd/csrc tools/msweep_image_check.cpp -o /tmp/msweep_image_check && /tmp/msweep_image_check
// (run by tests/test_host_logic.py::test_msweep_image_on_cpu)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "gf_msweep_image.h"

static int check(int32_t n, double avg_deg, int hubs, uint32_t seed, int W, bool uniform, int slack, int max_passes, bool sim = false) {
    std::mt19937 rng(seed);
    std::poisson_distribution<int> pd(avg_deg);
    std::uniform_int_distribution<int32_t> un(0, n - 1);
    std::normal_distribution<float> nd;
    std::vector<int32_t> rp(n + 1, 0), col;
    std::vector<float> val;
    for (int32_t i = 0; i < n; ++i) {
        int d = (i % 17 == 3) ? 0 : pd(rng);
        if (i < hubs) d = std::min(n, 60 + 40 * i);
        if (hubs < 0) {                               // power-law degrees, P(d > k) = (m / k)^2 with m = avg_deg / 2: the tail of a Barabasi-Albert graph
            const double u = std::max(1e-9, std::generate_canonical<double, 32>(rng));
            d = (int)std::min<double>(n / 4, 0.5 * avg_deg / std::sqrt(u));
        }
        std::vector<int32_t> c(d);
        for (auto& x : c) x = un(rng);
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        col.insert(col.end(), c.begin(), c.end());
        rp[i + 1] = (int32_t)col.size();
    }
    val.resize(col.size());
    for (auto& v : val) v = uniform ? 0.37f : nd(rng);
    MsweepImage im = build_msweep_image(n, rp.data(), col.data(), val.data(), uniform, slack, max_passes);
    if (!im.passes) {
        printf("n=%d deg=%.1f hubs=%d: no image\n", n, avg_deg, hubs);
        return 0;
    }
    std::vector<float> X((size_t)n * W), Y((size_t)n * W, NAN);
    for (auto& x : X) x = nd(rng);
    interpret_msweep_image(im, uniform, 0.37f, X.data(), Y.data(), W);
    int bad = 0;
    for (int32_t i = 0; i < n; ++i)
        for (int w = 0; w < W; ++w) {
            float a = 0.f, mag = 0.f;
            for (int32_t q = rp[i]; q < rp[i + 1]; ++q) {
                a = fmaf(uniform ? 1.f : val[q], X[(size_t)col[q] * W + w], a);
                mag += std::fabs((uniform ? 1.f : val[q]) * X[(size_t)col[q] * W + w]);
            }
            if (uniform) a *= 0.37f, mag *= 0.37f;
            const bool split = im.hub_split_rows && rp[i + 1] - rp[i] > im.hub_split;      // a split hub row: 32 partial chains + a fixed tree
            if (split ? std::fabs(a - Y[(size_t)i * W + w]) > 2e-6f * mag : memcmp(&a, &Y[(size_t)i * W + w], 4) != 0) ++bad;
        }
    printf("n=%d deg=%.1f hubs=%d %s slack=%d: sets=%d passes=%d rounds=%d fill=%.4f image=%.1f MB", n, avg_deg, hubs, uniform ? "uniform" : "weighted",
           slack, im.sets, im.passes, im.rounds, im.fill(), (im.ent.size() + im.val.size() + im.rows.size()) * 4 / 1e6);
    if (im.hub_rows) {
        size_t steps = 0, mx = 0;
        for (size_t w = 0; w + 1 < im.hubptr.size(); ++w) {
            size_t st = 0;
            for (size_t at = im.hubptr[w]; at < im.hubptr[w + 1]; at += 36 + (size_t)im.hub[at] * 32) st += im.hub[at];
            steps += st, mx = std::max(mx, st);
        }
        printf("  hubs: limit %d, %d rows (%d split, longer than %d), %.1f %% of the entries, hub steps per wave mean %.1f max %zu", im.hub_limit, im.hub_rows,
               im.hub_split_rows, im.hub_split, 100.0 * im.hub_entries / std::max<int64_t>(1, rp[n]), (double)steps / (im.hubptr.size() - 1), mx);
    }
    if (sim) printf("  LRU hit rate: %.3f (28k lines) %.3f (12k) %.3f (6k)", simulate_msweep_hits(im, n, 28000), simulate_msweep_hits(im, n, 12000), simulate_msweep_hits(im, n, 6000));
    printf("  %s\n", bad ? "MISMATCH" : "ok");
    return bad;
}

int main(int argc, char** argv) {
    int bad = 0;
    if (argc > 2 && !strcmp(argv[1], "csr")) {   // a CSR dumped as int32: n, nnz, rowptr[n + 1], col[nnz]  (tools/msweep_image_check csr <file> [slack])
        FILE* f = fopen(argv[2], "rb");
        int32_t n = 0, nnz = 0;
        if (!f || fread(&n, 4, 1, f) != 1 || fread(&nnz, 4, 1, f) != 1) return 2;
        std::vector<int32_t> rp(n + 1), col(nnz);
        if (fread(rp.data(), 4, n + 1, f) != (size_t)n + 1 || fread(col.data(), 4, nnz, f) != (size_t)nnz) return 2;
        fclose(f);
        std::vector<float> val(nnz, 1.f);
        for (int i = 3; i < std::max(argc, 4); ++i) {
            const int slack = i < argc ? atoi(argv[i]) : 5;
            MsweepImage im = build_msweep_image(n, rp.data(), col.data(), val.data(), true, slack, 2);
            printf("n=%d nnz=%d slack=%d: sets=%d passes=%d rounds=%d fill=%.4f  LRU hit rate %.3f (28k lines)\n", n, nnz, slack, im.sets, im.passes, im.rounds, im.fill(),
                   im.passes ? simulate_msweep_hits(im, n, 28000) : 0.0);
        }
        return 0;
    }
    if (argc > 1) {   // exploration: n slack...
        const int32_t n = atoi(argv[1]);
        for (int i = 2; i < argc; ++i) bad += check(n, 10.0, 0, 4, 1, true, atoi(argv[i]), 1, true);
        return bad != 0;
    }
    bad += check(100000, 10.0, -1, 9, 1, true, 5, 1, true);    // power-law degrees: hub rows computed outside the groups
    bad += check(60000, 8.0, -1, 10, 2, false, 5, 1);
    bad += check(50000, 200.0, 0, 12, 1, true, 5, 1);          // every row long: no hub rows, many rounds
    bad += check(52000, 2.0, 40, 13, 2, false, 5, 1);          // very sparse with 40 rows of 60 .. 1620 entries
    bad += check(1, 0.0, 0, 1, 2, true, 15, 1);             // (10 sets is the smallest geometry)
    bad += check(203, 6.0, 0, 2, 2, false, 15, 1);
    bad += check(4099, 10.0, 3, 3, 2, true, 15, 1);
    bad += check(40000, 8.0, 0, 6, 1, false, 10, 1);
    bad += check(100000, 10.0, 0, 4, 1, true, 15, 1, true);     // config 4's shape: 25 sets, one pass
    bad += check(131071, 6.0, 2, 5, 1, true, 20, 2);           // two passes
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad != 0;
}
