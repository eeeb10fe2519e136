// gf_sweep_image.h -- host-side builder of the SWEEP image (pure C++; gf_plan.hip uploads it, tools/sweep_image_check.cpp interprets
// it on the CPU the way spmm_sweep_kernel executes it).
//
// The idea (round 4).  A node-major hop at N = 1e5 re-gathers every 128-byte signal row ~10 times, and an XCD's
// 4 MiB L2 holds a third of a batch entry's 12.8 MB of rows: 58 % of the gathers miss and are served over the fabric at a third of
// the L2 rate.  The misses go away if all waves of an XCD walk the SOURCE rows in the same order at the same time -- then every row
// is fetched from HBM once and its other ~9 uses hit L2 -- but a wave that walks sources has to keep the partial sums of ALL its
// destination rows until the end.  The only on-chip memory large enough for the 12.8 MB of partial sums of a batch entry is the
// register file: 32 CUs x 512 KB = 16 MB per XCD.  So:
//   * the XCD's 512 wavefronts (16 per CU, 128 registers each) OWN the rows of one batch entry: each half of a wavefront (32 lanes x
//     one float = one 128-byte row) owns up to kSweepSlots = 100 destination rows, one REGISTER per row: slot j of a half-wave is
//     register ACC0 + j of its 32 lanes.  1024 half-waves x 100 slots = 102 400 rows per pass;
//   * a half-wave's work is the list of its rows' entries (source row, slot) SORTED BY SOURCE; the two halves of a wave walk their
//     lists side by side: step t = one buffer_load_dword whose lanes 0-31 fetch source row A_t and lanes 32-63 source row B_t, then
//     `v_add_f32 v[ACC0 + slot], ...` once per half with the slot as a relative register index (s_set_gpr_idx_on: the index is
//     wave-uniform, which is why a HALF-wave owns a row and the two halves are added under complementary exec masks);
//   * rows are dealt to the half-waves longest first in boustrophedon order, so every list has the same length up to a few entries:
//     all waves of the XCD advance through the sources at the same rate, and a coarse XCD-wide progress gate keeps them within a
//     window that fits L2;
//   * within a row the entries keep ascending column order: bit for bit the sums of spmm_sell_kernel.
// An entry is one 32-bit word: bits 0..23 the byte offset of the source row inside the tap (column * 128 < 2^24: N <= 131071 in this
// compact format), bits 24..31 the slot.  "Nothing" (the tail of a shorter list) = offset 0xffff80, past the end of every such tap:
// the buffer load's range check returns 0.0f for it.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <vector>

constexpr int32_t kSweepSlots = 100;                 // destination rows per half-wave = accumulator registers per lane
constexpr int32_t kSweepDepth = 8;                   // gather steps in flight per wave (ring registers)
constexpr int32_t kSweepWavesPerXcd = 512;           // 32 CUs x 16 waves (4 per SIMD at 128 registers)
constexpr uint32_t kSweepNothing = 0x00ffff80u;      // entry whose byte offset (0xffff80) lies past every tap of <= 131071 rows, slot 0
constexpr uint32_t kSweepNoRow = 0xffffff80u;        // output offset of a slot without a row: the buffer store drops it
constexpr int32_t kSweepMaxNodes = 131071;
constexpr int32_t kSweepBlock = 64;                  // steps per pass are a multiple of this (one LDS read of the entry stream per wave)

struct SweepImage {
    int32_t passes = 0;                              // ceil(N / (1024 * kSweepSlots))
    int32_t steps = 0;                               // steps per pass and wave, a multiple of kSweepDepth; the stream holds steps + kSweepDepth
    std::vector<uint32_t> ent;                       // [passes][kSweepWavesPerXcd][steps + kSweepDepth][2]   (half A, half B)
    std::vector<uint32_t> rows;                      // [passes][kSweepWavesPerXcd][kSweepSlots][2]          output byte offsets (row * 128)
    int64_t real_entries = 0, slots_total = 0;       // fill = real_entries / (passes * 1024 * steps)
};

// rowptr / col: CSR of the operator in ORIGINAL row order (row i lists its columns ascending); values are not part of the image
// (uniform GSOs only: the kernel sums the gathered rows and scales once).
inline SweepImage build_sweep_image(int32_t n, const int32_t* rowptr, const int32_t* col) {
    SweepImage im;
    const int32_t halves = kSweepWavesPerXcd * 2;
    im.passes = (n + halves * kSweepSlots - 1) / (halves * kSweepSlots);
    const int32_t bins = im.passes * halves;
    // deal the rows, longest first, over the bins in boustrophedon order: equal row counts (+-1) and near-equal entry counts
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b]; });
    std::vector<std::vector<int32_t>> binrows(bins);
    for (int32_t i = 0; i < n; ++i) {
        const int32_t round = i / bins, pos = i % bins;
        binrows[(round & 1) ? bins - 1 - pos : pos].push_back(order[i]);
    }
    int64_t longest = 0;
    for (auto& b : binrows) {
        int64_t c = 0;
        for (int32_t r : b) c += rowptr[r + 1] - rowptr[r];
        longest = std::max(longest, c);
        im.real_entries += c;
    }
    im.steps = (int32_t)((std::max<int64_t>(longest, 1) + kSweepBlock - 1) / kSweepBlock * kSweepBlock);
    const size_t stride = (size_t)(im.steps + kSweepDepth) * 2;
    im.ent.assign((size_t)im.passes * kSweepWavesPerXcd * stride, kSweepNothing);
    im.rows.assign((size_t)im.passes * kSweepWavesPerXcd * kSweepSlots * 2, kSweepNoRow);
    std::vector<std::pair<uint32_t, uint32_t>> list;   // (column, slot)
    for (int32_t b = 0; b < bins; ++b) {
        // bin b -> (pass, wave, half): consecutive bins alternate halves so that the two halves of a wave hold lists of equal length
        const int32_t pass = b / halves, wave = (b % halves) / 2, half = b & 1;
        list.clear();
        for (size_t j = 0; j < binrows[b].size(); ++j) {
            const int32_t r = binrows[b][j];
            im.rows[(((size_t)pass * kSweepWavesPerXcd + wave) * kSweepSlots + j) * 2 + half] = (uint32_t)r << 7;
            for (int32_t q = rowptr[r]; q < rowptr[r + 1]; ++q) list.push_back({(uint32_t)col[q], (uint32_t)j});
        }
        im.slots_total += (int64_t)binrows[b].size();
        std::stable_sort(list.begin(), list.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        // spread the list over the steps so that position ~ source: entry i sits at step floor(i * steps / size) or later
        uint32_t* e = im.ent.data() + ((size_t)pass * kSweepWavesPerXcd + wave) * stride;
        const size_t m = list.size();
        for (size_t i = 0; i < m; ++i) {
            const size_t at = m == (size_t)im.steps ? i : (size_t)((double)i * im.steps / (double)m);
            e[at * 2 + half] = (list[i].first << 7) | (list[i].second << 24);   // byte offset | slot << 24
        }
    }
    return im;
}

// What spmm_sweep_kernel computes for ONE batch entry (uniform GSO: sum, scale once by uval) in its own order of operations.
inline void interpret_sweep_image(const SweepImage& im, int32_t n, float uval, const float* X, float* Y, int32_t W) {
    const size_t stride = (size_t)(im.steps + kSweepDepth) * 2;
    std::vector<float> acc((size_t)kSweepSlots * W);
    for (int32_t pass = 0; pass < im.passes; ++pass)
        for (int32_t wave = 0; wave < kSweepWavesPerXcd; ++wave)
            for (int32_t half = 0; half < 2; ++half) {
                std::fill(acc.begin(), acc.end(), 0.f);
                const uint32_t* e = im.ent.data() + ((size_t)pass * kSweepWavesPerXcd + wave) * stride;
                for (int32_t t = 0; t < im.steps + kSweepDepth; ++t) {
                    const uint32_t w = e[(size_t)t * 2 + half], off = w & 0x00ffffffu, slot = w >> 24;
                    if ((uint64_t)off + 128 > (uint64_t)n * 128) continue;                     // out of range: + 0.0f
                    for (int32_t k = 0; k < W; ++k) acc[(size_t)slot * W + k] += X[(size_t)(off >> 7) * W + k];
                }
                for (int32_t j = 0; j < kSweepSlots; ++j) {
                    const uint32_t ro = im.rows[(((size_t)pass * kSweepWavesPerXcd + wave) * kSweepSlots + j) * 2 + half];
                    if (ro == kSweepNoRow) continue;
                    for (int32_t k = 0; k < W; ++k) Y[(size_t)(ro >> 7) * W + k] = acc[(size_t)j * W + k] * uval;
                }
            }
}
