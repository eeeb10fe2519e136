// gf_sweep_image.h -- host-side builder of the SWEEP image (pure C++; gf_plan.hip uploads it, tools/sweep_image_check.cpp interprets
// it on the CPU the way spmm_sweep_kernel executes it).
//
// The idea (round 4).  A node-major hop at N = 1e5 re-gathers every 128-byte signal row ~10 times, and an XCD's
// 4 MiB L2 holds a third of a batch entry's 12.8 MB of rows: 58 % of the gathers miss and are served over the fabric at a third of
// the L2 rate.  The misses go away if all waves of an XCD walk the SOURCE rows in the same order at the same time -- then every row
// is fetched from HBM once per pass and its other uses hit L2 -- but a wave that walks sources has to keep the partial sums of ALL
// its destination rows until the end, and the only on-chip memory large enough for that is the register file (512 KB per CU).  So:
//   * an XCD works on TWO batch entries at a time: lanes 0-31 of every wavefront belong to entry b, lanes 32-63 to entry b + 8 (32
//     lanes x one float = one 128-byte row of either entry);
//   * each of the XCD's 512 wavefronts (16 per CU, 128 registers) OWNS up to kSweepRows = 99 destination rows per pass: row j = ONE
//     accumulator register v[ACC0 + j] -- its low half holds the sum for entry b, its high half the sum for entry b + 8.  512 x 99 rows
//     per pass: N = 1e5 takes two passes;
//   * a wave's work in a pass is the list of its rows' entries (source row, slot) SORTED BY SOURCE.  A step = one buffer_load_dword (every
//     lane: its entry's tap + source offset + lane offset: the SAME source row of both entries) and, kSweepDepth steps later, ONE
//     `v_add_f32 v[ACC0 + slot], ...` with the slot as relative register index (s_set_gpr_idx_on: the index is wave-uniform -- which is
//     why a step serves one destination row, and two batch entries share it to fill the 64 lanes);
//   * rows are dealt to the (pass, wave) bins longest first in boustrophedon order: all lists are equally long up to a few entries,
//     every wave of the XCD advances through the sources at the same rate; shorter lists are spread evenly (entry i of m sits at step
//     i * steps / m) and the gaps gather row 0 into a TRASH slot (register ACC0 + 99);
//   * within a row the entries keep ascending column order: bit for bit the sums of spmm_sell_kernel.
// An entry is one 32-bit word: bits 8..31 the byte offset of the source row inside the tap (column * 128 < 2^24: N <= 131071 in this
// compact format), bits 0..7 the slot (s_set_gpr_idx_on takes the low byte of an SGPR as the index: no decoding).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <vector>

constexpr int32_t kSweepSlots = 100;                 // accumulator registers per lane: kSweepRows destination rows + the trash slot
constexpr int32_t kSweepRows = kSweepSlots - 1;
constexpr int32_t kSweepDepth = 8;                   // gather steps in flight per wave (ring registers)
constexpr int32_t kSweepWavesPerXcd = 512;           // 32 CUs x 16 waves (4 per SIMD at 128 registers)
constexpr uint32_t kSweepNothing = kSweepRows;       // entry of a gap: source row 0 into the trash slot
constexpr uint32_t kSweepNoRow = 0xffffffffu;        // output offset of a slot without a row
constexpr int32_t kSweepMaxNodes = 131071;
constexpr int32_t kSweepBlock = 64;                  // steps per pass are a multiple of this (one LDS read of the entry stream per wave)

struct SweepImage {
    int32_t passes = 0;                              // ceil(N / (512 * kSweepRows))
    int32_t steps = 0;                               // steps per pass and wave, a multiple of kSweepBlock; the stream holds steps + kSweepDepth
    std::vector<uint32_t> ent;                       // [kSweepWavesPerXcd][passes][steps + kSweepDepth]
    std::vector<uint32_t> rows;                      // [kSweepWavesPerXcd][passes][kSweepSlots]   output byte offsets (row * 128), kSweepNoRow = none
    int64_t real_entries = 0;                        // fill = real_entries / (passes * 512 * steps)
};

// rowptr / col: CSR of the operator in ORIGINAL row order (row i lists its columns ascending); values are not part of the image
// (uniform GSOs only: the kernel sums the gathered rows and scales once).
inline SweepImage build_sweep_image(int32_t n, const int32_t* rowptr, const int32_t* col) {
    SweepImage im;
    im.passes = (n + kSweepWavesPerXcd * kSweepRows - 1) / (kSweepWavesPerXcd * kSweepRows);
    const int32_t bins = im.passes * kSweepWavesPerXcd;
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
    const size_t stride = (size_t)(im.steps + kSweepDepth);
    im.ent.assign((size_t)bins * stride, kSweepNothing);
    im.rows.assign((size_t)bins * kSweepSlots, kSweepNoRow);
    std::vector<std::pair<uint32_t, uint32_t>> list;   // (column, slot)
    for (int32_t b = 0; b < bins; ++b) {
        const int32_t wave = b % kSweepWavesPerXcd, pass = b / kSweepWavesPerXcd;   // (consecutive bins = consecutive waves: the passes are balanced too)
        list.clear();
        for (size_t j = 0; j < binrows[b].size(); ++j) {
            const int32_t r = binrows[b][j];
            im.rows[((size_t)wave * im.passes + pass) * kSweepSlots + j] = (uint32_t)r << 7;
            for (int32_t q = rowptr[r]; q < rowptr[r + 1]; ++q) list.push_back({(uint32_t)col[q], (uint32_t)j});
        }
        std::stable_sort(list.begin(), list.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        uint32_t* e = im.ent.data() + ((size_t)wave * im.passes + pass) * stride;
        const size_t m = list.size();
        for (size_t i = 0; i < m; ++i) {   // position ~ source: entry i sits at step floor(i * steps / m)
            const size_t at = m == (size_t)im.steps ? i : (size_t)((double)i * im.steps / (double)m);
            e[at] = (list[i].first << 15) | list[i].second;   // (byte offset << 8) | slot
        }
    }
    return im;
}

// What spmm_sweep_kernel computes for ONE batch entry (uniform GSO: sum, scale once by uval) in its own order of operations.
inline void interpret_sweep_image(const SweepImage& im, int32_t n, float uval, const float* X, float* Y, int32_t W) {
    const size_t stride = (size_t)(im.steps + kSweepDepth);
    std::vector<float> acc((size_t)kSweepSlots * W);
    for (int32_t wave = 0; wave < kSweepWavesPerXcd; ++wave)
        for (int32_t pass = 0; pass < im.passes; ++pass) {
            std::fill(acc.begin(), acc.end(), 0.f);
            const uint32_t* e = im.ent.data() + ((size_t)wave * im.passes + pass) * stride;
            for (int32_t t = 0; t < im.steps + kSweepDepth; ++t) {
                const uint32_t w = e[t], off = w >> 8, slot = w & 0xffu;
                for (int32_t k = 0; k < W; ++k) acc[(size_t)slot * W + k] += X[(size_t)(off >> 7) * W + k];
            }
            for (int32_t j = 0; j < kSweepRows; ++j) {
                const uint32_t ro = im.rows[((size_t)wave * im.passes + pass) * kSweepSlots + j];
                if (ro == kSweepNoRow) continue;
                for (int32_t k = 0; k < W; ++k) Y[(size_t)(ro >> 7) * W + k] = acc[(size_t)j * W + k] * uval;
            }
        }
    (void)n;
}
