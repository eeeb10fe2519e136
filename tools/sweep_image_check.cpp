// sweep_image_check.cpp -- CPU check of the SWEEP image (graph-neural-networks_amd/csrc/gf_sweep_image.h): builds it for random graphs,
// interprets it the way spmm_sweep_kernel does and compares BITWISE with the row-by-row sums in ascending column order.
//   g++ -O2 -std=c++17 -Igraph-neural-networks_amd/csrc tools/sweep_image_check.cpp -o /tmp/sweep_image_check && /tmp/sweep_image_check
// (run by tests/test_host_logic.py::test_sweep_image_on_cpu)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "gf_sweep_image.h"

static int check(int32_t n, double avg_deg, int hubs, uint32_t seed, int W) {
    std::mt19937 rng(seed);
    std::poisson_distribution<int> pd(avg_deg);
    std::uniform_int_distribution<int32_t> un(0, n - 1);
    std::normal_distribution<float> nd;
    std::vector<int32_t> rp(n + 1, 0), col;
    for (int32_t i = 0; i < n; ++i) {
        int d = (i % 17 == 3) ? 0 : pd(rng);
        if (i < hubs) d = std::min(n, 300 + 500 * i);
        std::vector<int32_t> c(d);
        for (auto& x : c) x = un(rng);
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        col.insert(col.end(), c.begin(), c.end());
        rp[i + 1] = (int32_t)col.size();
    }
    SweepImage im = build_sweep_image(n, rp.data(), col.data());
    std::vector<float> X((size_t)n * W), Y((size_t)n * W, NAN);
    for (auto& x : X) x = nd(rng);
    interpret_sweep_image(im, n, 0.37f, X.data(), Y.data(), W);
    int bad = 0;
    for (int32_t i = 0; i < n; ++i)
        for (int w = 0; w < W; ++w) {
            float a = 0.f;
            for (int32_t q = rp[i]; q < rp[i + 1]; ++q) a += X[(size_t)col[q] * W + w];
            a *= 0.37f;
            if (memcmp(&a, &Y[(size_t)i * W + w], 4) != 0) ++bad;
        }
    const double fill = (double)im.real_entries / ((double)im.passes * kSweepWavesPerXcd * im.steps);
    printf("n=%d deg=%.1f hubs=%d: passes=%d steps=%d fill=%.4f image=%.1f MB  %s\n", n, avg_deg, hubs, im.passes, im.steps, fill,
           (im.ent.size() + im.rows.size()) * 4 / 1e6, bad ? "MISMATCH" : "ok");
    return bad;
}

int main() {
    int bad = 0;
    bad += check(1, 0.0, 0, 1, 2);
    bad += check(203, 6.0, 0, 2, 2);
    bad += check(4099, 10.0, 3, 3, 2);
    bad += check(100000, 10.0, 0, 4, 1);      // config 4's shape: two passes of ~1000 steps
    bad += check(131071, 6.0, 2, 5, 1);       // three passes
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad != 0;
}
