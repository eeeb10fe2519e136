// gf_stream_image.h -- host-side builder of the STREAM image of a scheduled CSR (pure C++, no HIP: gf_plan.hip uploads the result,
// tools/stream_image_check.cpp interprets it on the CPU exactly as spmm_stream_kernel does and compares with the CSR product).
//
// Why another image.  The SELL-8 kernel of round 1-3 (spmm_sell_kernel) gives every wavefront two slices and lets it die: kptr ->
// entries -> LDS ring -> gathers -> store are four dependent round trips for ~20 gathers, so a CU has gathers in flight for a third
// of the time (band-graph probe: 14 TB/s of gathered rows where the chip serves 33).  spmm_stream_kernel keeps waves alive for the
// whole launch and feeds them ONE flat instruction stream in which nothing has to be looked up or decoded:
//   * a STEP is 8 entries, one per lane group (8 lanes = one 128-byte signal row): every step is exactly one gather instruction
//     of the wave, so the gather ring has a compile-time depth and `s_waitcnt vmcnt` counts are exact;
//   * an entry IS the byte offset of the row to gather (column * 128); "nothing" (a short row's padding, a run's tail) is an offset
//     past the end of the tap: the gathers are BUFFER loads, whose range check returns zeros for it without touching memory -- no
//     compare / select per step, a padded step adds +0.0f;
//   * the rows of a SELL-8 slice (8 consecutive scheduled rows, padded to the longest, at least one step) are max(len, 1) steps; the
//     last one is flagged in the run's 32-bit LAST mask (a scalar: `s_bitcmp1 + s_cbranch` per step), the output rows of the run's
//     slices sit in a table beside it -- slice boundaries and row ids cost nothing on the steps that are not boundaries;
//   * steps are packed into RUNS of exactly kStreamRun = 32 steps = 1 KiB: one coalesced 16-byte load per lane fetches a run
//     (lane (g, i) holds steps 4i .. 4i+3 of lane group g; a step's entry reaches the other lanes of its group with one ds_swizzle).
//     A run holds whole slices only (first-fit decreasing inside every unit of the schedule; <= 16 slices), so ANY wave may take ANY
//     run: runs are handed out in schedule order and the set of rows in flight stays a narrow front of the schedule (what the
//     locality groups need) although the waves never exit;
//   * slices longer than a run (hub rows) stay in a residual SELL image for spmm_sell_kernel;
//   * PREFETCH RUNS (graphs with locality only).  What bounds a gather kernel on this chip is the ~64-100 L2 requests a CU's vector
//     cache keeps in flight times their latency, and a wave's loads return IN ORDER: one L2 miss (first touch of a row: HBM, ~7x the
//     latency of a hit) holds back every hit the wave issued after it.  With 10 % first touches more than half of the 8-row gather
//     instructions contain one, and the whole kernel runs at the miss rate (tools/pmc_hop.sh: 12-14 TB/s gathered on a band graph
//     whose reads hit L2 88 % of the time, where all-hit gathers run at 31).  So the first touches are taken out of the gather runs:
//     a unit's source rows that are not already resident (a small LRU model of the XCD's L2 over the units) are listed in PREFETCH
//     runs -- ordinary runs whose gathers nobody adds up: LAST on the final step with no output rows -- placed one unit AHEAD in
//     the stream.  Tickets are handed out in order, so whichever waves draw those runs eat the misses while the others gather hits;
//     no new kernel role, no synchronisation.  A unit whose new rows are more than half of its references (a random graph: the
//     prefetch would be the gather) gets none.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

constexpr int32_t kStreamRun = 32;                  // steps per run
constexpr int32_t kStreamSlices = 16;               // slices per run at most (two row-id words per lane)
constexpr uint32_t kStreamNothing = 0xffffff80u;    // byte offset past every tap (taps are < 2^32 - 256 bytes): the buffer load returns 0
constexpr uint32_t kStreamNoRow = 0xffffffffu;
constexpr int32_t kStreamBlock = 784;               // slices packed together when the schedule has no units
constexpr int32_t kStreamResidentRows = 20480;      // L2 model for the prefetch runs: rows (128 bytes) an XCD's 4 MiB L2 is trusted to keep
constexpr double kStreamPrefetchMaxRatio = 0.25;    // prefetch a unit's new rows only if they are at most this fraction of its references

struct StreamImage {
    int32_t n_runs = 0;
    std::vector<uint32_t> ent;                      // [n_runs][8 lane groups][kStreamRun steps]   byte offsets (column * 128)
    std::vector<float> val;                         // same shape; 0.0f where there is nothing
    std::vector<uint32_t> last;                     // [n_runs]  bit u: step u is the last step of a slice
    std::vector<uint32_t> rows;                     // [n_runs][64 lanes][2]  lane (g, i): output rows of lane group g in slices i and i + 8 of the run
    std::vector<int32_t> hub_slices;                // indices of the SELL slices that are NOT in the stream (longer than a run)
    int64_t pad_steps = 0, gather_steps = 0;
    int32_t pf_runs = 0;                            // prefetch runs among n_runs
    int64_t pf_rows = 0;
};

// rowptr / col / val: the SCHEDULED CSR (row p of it is output row rowid[p]); slices = 8 consecutive scheduled rows.
// unit_starts: scheduled-row positions at which a unit of the schedule begins (locality group / sort window; ascending, first = 0);
// nullptr = fixed blocks of kStreamBlock slices.
inline StreamImage build_stream_image(int32_t n, const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* rowid,
                                      const std::vector<int32_t>* unit_starts = nullptr, int prefetch_lead = 2) {
    StreamImage im;
    const int32_t ns = (n + 7) / 8;
    std::vector<int32_t> len(ns, 0);
    for (int32_t sl = 0; sl < ns; ++sl)
        for (int32_t r = 0; r < 8 && sl * 8 + r < n; ++r) len[sl] = std::max(len[sl], rowptr[sl * 8 + r + 1] - rowptr[sl * 8 + r]);
    auto steps_of = [&](int32_t sl) { return std::max(len[sl], 1); };   // an all-empty slice still needs its (padding) step to carry LAST
    auto emit = [&](int32_t sl, int32_t run, int32_t at, int32_t nth) {
        uint32_t* e = im.ent.data() + (size_t)run * 8 * kStreamRun;
        float* v = im.val.data() + (size_t)run * 8 * kStreamRun;
        for (int32_t r = 0; r < 8; ++r) {
            const int32_t p = sl * 8 + r;
            if (p >= n) continue;                                   // lane group without a row: nothing to gather, nothing to store
            for (int32_t q = rowptr[p]; q < rowptr[p + 1]; ++q) {
                e[r * kStreamRun + at + (q - rowptr[p])] = (uint32_t)col[q] << 7;
                v[r * kStreamRun + at + (q - rowptr[p])] = val[q];
            }
            im.rows[((size_t)run * 64 + r * 8 + (nth & 7)) * 2 + (nth >> 3)] = (uint32_t)rowid[p];
        }
        im.last[run] |= 1u << (at + steps_of(sl) - 1);
        im.gather_steps += steps_of(sl);
    };
    // The units of the schedule (a locality group / a sort window of gf_plan.hip's schedule(): every unit holds rows of all lengths,
    // longest first) are packed on their own -- the runs of a unit stay together in the stream, so the front of rows in flight
    // follows the schedule -- by first-fit decreasing: longest slice first into the first run of the unit with room.
    std::vector<int32_t> order, bounds;                             // block boundaries in slices
    if (unit_starts && !unit_starts->empty()) {
        for (int32_t p : *unit_starts) {
            const int32_t sl = p / 8;                               // a slice that straddles two units goes with the earlier one
            if (sl > 0 && sl < ns && (bounds.empty() || sl > bounds.back())) bounds.push_back(sl);
        }
    } else {
        for (int32_t sl = kStreamBlock; sl < ns; sl += kStreamBlock) bounds.push_back(sl);
    }
    bounds.push_back(ns);
    // prefetch lists: per unit, the source rows it references that the L2 model does not hold (rows referenced or prefetched by the
    // units before it, most recent kStreamResidentRows of them)
    std::vector<std::vector<int32_t>> pf(bounds.size());
    if (prefetch_lead > 0 && n > kStreamResidentRows) {      // (a graph whose whole gather panel fits L2 needs none)
        std::vector<int32_t> stamp(n, -1), fifo;     // stamp[c] = position in fifo of the latest touch, -1 = not resident
        size_t head = 0;                             // fifo[head ..] are the resident touches
        int32_t lo = 0;
        for (size_t u = 0; u < bounds.size(); ++u) {
            const int32_t p0 = lo * 8, p1 = std::min(n, bounds[u] * 8);
            std::vector<int32_t>& list = pf[u];
            int64_t refs = 0;
            for (int32_t p = p0; p < p1; ++p) {
                for (int32_t q = rowptr[p]; q < rowptr[p + 1]; ++q) {
                    ++refs;
                    const int32_t c = col[q];
                    if (stamp[c] < 0) list.push_back(c);
                    stamp[c] = (int32_t)fifo.size();
                    fifo.push_back(c);
                }
            }
            std::sort(list.begin(), list.end());
            list.erase(std::unique(list.begin(), list.end()), list.end());
            if ((double)list.size() > kStreamPrefetchMaxRatio * (double)refs) list.clear();
            // evict: keep the most recent kStreamResidentRows distinct rows (approximately: count touches whose stamp is current)
            size_t live = 0;
            for (size_t i = fifo.size(); i > head; --i)
                if (stamp[fifo[i - 1]] == (int32_t)(i - 1) && ++live > (size_t)kStreamResidentRows) {
                    for (size_t j = head; j < i; ++j)
                        if (stamp[fifo[j]] == (int32_t)j) stamp[fifo[j]] = -1;
                    head = i;
                    break;
                }
            lo = bounds[u];
        }
    }
    auto emit_prefetch = [&](const std::vector<int32_t>& list) {
        for (size_t i = 0; i < list.size(); i += 8 * kStreamRun) {
            const int32_t run = im.n_runs++;
            im.ent.resize((size_t)im.n_runs * 8 * kStreamRun, kStreamNothing);
            im.val.resize((size_t)im.n_runs * 8 * kStreamRun, 0.f);
            im.last.resize(im.n_runs, 0u);
            im.rows.resize((size_t)im.n_runs * 64 * 2, kStreamNoRow);
            for (size_t j = i; j < std::min(list.size(), i + 8 * kStreamRun); ++j)
                im.ent[(size_t)run * 8 * kStreamRun + ((j - i) & 7) * kStreamRun + ((j - i) >> 3)] = (uint32_t)list[j] << 7;
            im.last[run] = 1u << (kStreamRun - 1);   // the sums of a prefetch run are dropped: LAST with no output rows
            ++im.pf_runs;
            im.pf_rows += (int64_t)std::min(list.size() - i, (size_t)8 * kStreamRun);
        }
    };
    int32_t b0 = 0;
    size_t unit = 0;
    for (size_t u = 0; u < pf.size() && u < (size_t)prefetch_lead; ++u) emit_prefetch(pf[u]);
    for (int32_t b1 : bounds) {
        if (unit + prefetch_lead < pf.size()) emit_prefetch(pf[unit + prefetch_lead]);   // prefetch_lead units ahead of the runs that gather these rows
        ++unit;
        order.clear();
        for (int32_t sl = b0; sl < b1; ++sl) {
            if (steps_of(sl) > kStreamRun) im.hub_slices.push_back(sl);
            else order.push_back(sl);
        }
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return len[a] > len[b]; });
        const int32_t first = im.n_runs;
        std::vector<int32_t> used, cnt;                             // steps used / slices placed per run of this unit
        size_t lo = 0;                                              // runs before `lo` are full
        for (int32_t sl : order) {
            const int32_t need = steps_of(sl);
            int32_t slot = -1;
            while (lo < used.size() && (used[lo] == kStreamRun || cnt[lo] == kStreamSlices)) ++lo;
            for (size_t o = lo; o < used.size(); ++o)
                if (used[o] + need <= kStreamRun && cnt[o] < kStreamSlices) {
                    slot = (int32_t)o;
                    break;
                }
            if (slot < 0) {
                slot = (int32_t)used.size();
                used.push_back(0);
                cnt.push_back(0);
                ++im.n_runs;
                im.ent.resize((size_t)im.n_runs * 8 * kStreamRun, kStreamNothing);
                im.val.resize((size_t)im.n_runs * 8 * kStreamRun, 0.f);
                im.last.resize(im.n_runs, 0u);
                im.rows.resize((size_t)im.n_runs * 64 * 2, kStreamNoRow);
            }
            emit(sl, first + slot, used[slot], cnt[slot]);
            used[slot] += need;
            cnt[slot] += 1;
        }
        b0 = b1;
    }
    im.pad_steps = (int64_t)(im.n_runs - im.pf_runs) * kStreamRun - im.gather_steps;
    return im;
}

// What spmm_stream_kernel computes for ONE batch entry from the stream alone (uniform = 1: sum the rows, scale once by uval, as
// the kernel does; else fma with the value stream), in the kernel's own order of operations.  X, Y: [n][W].  Offsets past the tap
// read as zeros (the buffer load's range check).
inline void interpret_stream_image(const StreamImage& im, int32_t n, int uniform, float uval, const float* X, float* Y, int32_t W) {
    std::vector<float> acc((size_t)8 * W);
    const std::vector<float> zeros(W, 0.f);
    for (int32_t run = 0; run < im.n_runs; ++run) {
        std::fill(acc.begin(), acc.end(), 0.f);
        const uint32_t* e = im.ent.data() + (size_t)run * 8 * kStreamRun;
        const float* v = im.val.data() + (size_t)run * 8 * kStreamRun;
        int32_t nth = 0;
        for (int32_t st = 0; st < kStreamRun; ++st) {
            for (int32_t g = 0; g < 8; ++g) {
                const uint32_t eg = e[g * kStreamRun + st];
                float* a = acc.data() + (size_t)g * W;
                const float* x = (uint64_t)eg + 128 <= (uint64_t)n * 128 ? X + (size_t)(eg >> 7) * W : zeros.data();
                for (int32_t w = 0; w < W; ++w) a[w] = uniform ? a[w] + x[w] : fmaf(v[g * kStreamRun + st], x[w], a[w]);
                if (im.last[run] >> st & 1u) {
                    const uint32_t row = im.rows[((size_t)run * 64 + g * 8 + (nth & 7)) * 2 + (nth >> 3)];
                    if (row != kStreamNoRow)
                        for (int32_t w = 0; w < W; ++w) Y[(size_t)row * W + w] = uniform ? a[w] * uval : a[w];
                    for (int32_t w = 0; w < W; ++w) a[w] = 0.f;
                }
            }
            if (im.last[run] >> st & 1u) ++nth;
        }
    }
}
