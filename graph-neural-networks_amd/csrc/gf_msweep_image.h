// gf_msweep_image.h -- host-side builder of the MSWEEP image (pure C++: gf_plan.hip uploads it, tools/msweep_image_check.cpp
// interprets it on the CPU the way spmm_msweep_kernel executes it).
//
// What the image is for (round 5).  A node-major hop (graphML.py:158-161, one `x = torch.matmul(x, S)`) at N = 1e5
// re-gathers every 128-byte signal row ~10 times; an XCD's 4 MiB L2 holds a third of a batch entry's 12.8 MB of rows, so a row-by-row
// kernel misses on 58 % of its gathers.  The misses go away when all waves of an XCD walk the SOURCE rows together: a row is then
// fetched over the fabric once and its other ~9 uses hit L2.  A wave that walks sources must keep the partial sums of ALL its
// destination rows to the end -- 12.8 MB per batch entry, which only the XCD's register files hold (32 CUs x 512 KB) -- and must add
// each gathered row to an accumulator that the DATA selects.  Round 4's sweep selected it by relative register indexing and was bound
// by the scalar instructions that takes.  Here the selection is an MFMA operand:
//
//   v_mfma_f32_4x4x1_16b_f32   D_b[i][j] += A_b[i] * B_b[j]     16 blocks b = lane / 4, i = accumulator register 0..3, j = lane % 4
//
//   * a gather instruction loads 8 source rows (lane = 8 * position + fg, 16 bytes per lane: the four VGPRs hold features 4 fg + v);
//     VGPR v is the B operand of MFMA v: position p feeds blocks 2p and 2p + 1, i.e. the 8 lanes of position p again;
//   * the A operand is ONE VGPR: the edge weight in the lanes whose i = lane % 4 equals the destination SLOT, zero elsewhere.  The four
//     MFMAs of a step add the gathered row of position p to slot i of position p in one accumulator SET (4 x 4 registers): a set holds
//     8 positions x 4 slots = 32 destination rows x 32 features.  The slot is data; the set is code (static registers);
//   * a wave owns S sets (S = 25: 800 rows, 400 of its 512 registers), an XCD's 128 waves (one per SIMD) own 102 400 rows: one batch
//     entry in one pass.  The wave's program is T ROUNDS of S steps: round t, step s serves set s.  The group of 4 destination rows
//     behind (set, position) lists the (source, slot) pairs of its rows sorted by source and spreads them over the T rounds so that
//     round t holds sources near t * N / T: every wave of the XCD is at the same place of the source range at the same time, whatever
//     set it is serving.  Rows are dealt to the groups by degree so that all groups hold (almost) the same number of entries (bands of
//     consecutive row ids = sets, lowest band = last set: stored last, gathered first by the next hop);
//   * fp32 MFMA is an exact fmaf chain and the rounds visit a row's sources in ascending order: bit for bit the sums of spmm_sell_kernel.
//
// Entry word (32 bits): bits 7..31 = byte offset of the source row inside the tap (row * 128), bits 0..3 = one-hot slot; a gap is
// kMsPad: an offset beyond the tap (the buffer load's range check returns zeros without a memory request) and no slot bit.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <queue>
#include <vector>

constexpr int32_t kMsDepth = 5;                      // granule of the geometry: sets per wave are a multiple of it (the kernel keeps 5 or 10 gathers in flight per wave)
constexpr int32_t kMsWavesPerXcd = 128;              // 32 CUs x 4 SIMDs x one 512-register wave
constexpr int32_t kMsMaxSets = 25;                   // accumulator sets per wave: 25 x 16 = 400 registers
constexpr uint32_t kMsPad = 0xffffff80u;             // entry of a gap / row offset of a slot without a row
constexpr int32_t kMsMaxRounds = 4096;

struct MsweepImage {
    int32_t sets = 0;                                // S: 10, 15, 20 or 25 (a multiple of kMsDepth)
    int32_t s4 = 0;                                  // S rounded up to a multiple of 4 (entries of a position are loaded 4 at a time)
    int32_t passes = 0;                              // ceil(groups / (128 * S * 8)); one pass = one sweep of the sources per batch entry
    int32_t rounds = 0;                              // T; the streams hold T + 2 rounds (the kernel's entry loads run two rounds ahead)
    std::vector<uint32_t> ent;                       // [passes][128 waves][T + 2][8 positions][8 quads][4]: a round is 1 KB -- lane 8 p + i of the wave loads
                                                     // quad i (steps 4 i .. 4 i + 3) of position p with ONE 16-byte load per round; the kernel hands
                                                     // a step's entry to the position's 8 lanes with two DPP moves
    std::vector<float> val;                          // same shape, weighted GSOs only (empty when uniform)
    std::vector<uint32_t> rows;                      // [passes][128 waves][S][32]  output byte offset (row * 128) of (set, position * 4 + slot), kMsPad = none
    int64_t real_entries = 0;                        // fill = real_entries / (passes * 128 * S * 8 * T)   (entries of the rows IN groups)
    // Hub rows (round 6): rows longer than hub_limit are in no group -- one of them would set the number of rounds every wave walks.  Each
    // wave computes its share of them straight from these streams between its store phase and the hand-over (ascending columns, one
    // fmaf per entry: the same chain).  A wave's block = QUADS of 4 octets (32 rows of similar length, longest first), back to back:
    //   word 0: Lq (steps of the quad = its longest row), word 1: 1 = the quad is ONE row split over its 32 slots, words 2..3: 0
    //   words 4..35:  output byte offset (row * 128, kMsPad = none) of (position p, octet o) at 4 + 4 p + o
    //   then Lq x 32 words: source byte offset (row * 128, kMsPad = gap) of step j, (position p, octet o) at 36 + 32 j + 4 p + o
    // (a lane -- position p -- reads its four octets' words of a step with one 16-byte load); hubval has the same shape (weighted GSOs).
    int32_t hub_limit = 0, hub_rows = 0, hub_split = 0, hub_split_rows = 0;
    int64_t hub_entries = 0;
    std::vector<uint32_t> hub;
    std::vector<float> hubval;
    std::vector<uint32_t> hubptr;                    // [passes * 128 + 1] word offsets of the waves' blocks in hub
    double fill() const { return passes ? (double)real_entries / ((double)passes * kMsWavesPerXcd * sets * 8 * rounds) : 0.0; }
    size_t stream_words() const { return (size_t)(rounds + 2) * 256; }   // per (pass, wave)
    size_t at(int32_t t, int32_t p, int32_t step) const { return (size_t)t * 256 + (size_t)p * 32 + (step / 4) * 4 + (step & 3); }
};

// rowptr / col / val: CSR of the operator in ORIGINAL row order, columns ascending inside a row.  uniform: the values are not stored
// (the kernel sums the gathered rows and scales once, as spmm_sell_kernel<UNI = 1> does).  slack_pct: rounds beyond the MEAN
// group length, in percent -- the room the placement has to keep round ~ source (config 4: 5 % -> T = 42 for groups of 39.1: measured best).
inline MsweepImage build_msweep_image(int32_t n, const int32_t* rowptr, const int32_t* col, const float* val, bool uniform,
                                      int32_t slack_pct = 15, int32_t max_passes = 1, int32_t force_hub_limit = 0) {
    MsweepImage im;
    if (n <= 0 || (int64_t)n * 128 >= (int64_t)kMsPad) return im;
    // Hub rows: candidates for the limit are multiples of the mean group length; the cost of a candidate = steps of the sweep (rounds x sets:
    // the rounds must hold the longest group -- at least the longest row in a group plus three short ones -- and the mean with its slack)
    // + three sweep steps per hub step (a hub step is four dependent-latency gathers, out of sweep order).  A graph whose longest row fits the
    // rounds anyway (ER, SBM, kNN) gets no hubs and the image it always had.
    std::vector<int32_t> degs(n);
    for (int32_t r = 0; r < n; ++r) degs[r] = rowptr[r + 1] - rowptr[r];
    int32_t hub_limit = 0;
    {
        std::vector<int32_t> sorted(degs);
        std::sort(sorted.begin(), sorted.end());
        const int32_t maxdeg = sorted.empty() ? 0 : sorted.back(), dq = sorted[(size_t)n / 4];
        const double g0 = 4.0 * rowptr[n] / std::max(1, n);
        double best = 0.0;
        const double cands[] = {1e9, 8, 6, 4, 3, 2, 1.5, 1.25, 1.0, 0.75};
        for (double c : cands) {
            const int32_t H = c > 1e8 ? maxdeg : (int32_t)std::ceil(c * g0);
            if (c < 1e8 && H >= maxdeg) continue;
            int64_t eh = 0, nh = 0;
            double hubsteps = 0.0;
            for (int32_t i = n - 1; i >= 0 && sorted[i] > H; --i) ++nh, eh += sorted[i];
            for (int64_t i = 0; i < nh; i += 32) hubsteps += sorted[n - 1 - i];          // a quad takes its longest row's steps
            const double gl = 4.0 * (rowptr[n] - eh) / std::max<int64_t>(1, n - nh);
            const double T = std::max(gl * (100 + slack_pct) / 100.0, (double)H + 3.0 * dq);
            const double cost = T * kMsMaxSets + 3.0 * hubsteps / kMsWavesPerXcd;
            if (hub_limit == 0 || cost < best) best = cost, hub_limit = H;
        }
        if (force_hub_limit > 0) hub_limit = force_hub_limit;      // (experiments: the cost model's choice against fixed limits)
    }
    int32_t n_in = 0;                                // rows in groups
    int64_t e_in = 0;
    for (int32_t r = 0; r < n; ++r)
        if (degs[r] <= hub_limit) ++n_in, e_in += degs[r];
    if (n_in == 0) return im;
    const int32_t groups = (n_in + 3) / 4;
    // passes: as few as the largest geometry (25 sets per wave) allows -- every pass sweeps the sources again; sets per wave: the smallest
    // geometry that holds every group in that many passes
    const int32_t passes = (int32_t)((groups + (int64_t)kMsWavesPerXcd * kMsMaxSets * 8 - 1) / ((int64_t)kMsWavesPerXcd * kMsMaxSets * 8));
    int32_t S = 0;
    for (int32_t s = 2 * kMsDepth; s <= kMsMaxSets; s += kMsDepth)   // (the kernel's ring needs >= 10 steps per round)
        if ((int64_t)passes * kMsWavesPerXcd * s * 8 >= groups) { S = s; break; }
    if (!S || passes > max_passes) return im;
    // rows -> sets: a BAND of consecutive row ids is one accumulator set of the whole XCD (128 waves x 8 positions x up to 4 slots), and the
    // LOWEST bands get the LAST sets: a wave stores its sets in order, so the rows written last -- the ones still in the XCD's L2 when the
    // next hop of the fused chain starts its sweep at source row 0 -- are the rows that sweep gathers first.
    // rows of a band -> groups of 3 or 4: longest row first, each to the group with the fewest entries so far that still has a free slot (the
    // first rows seed one group each): every group ends with (almost) the same entry total -- the longest one sets T
    const int32_t cap = (int32_t)std::ceil((double)e_in / std::max<int64_t>(1, std::min<int64_t>((int64_t)passes * S * kMsWavesPerXcd * 8, n_in)) * (100 + slack_pct) / 100.0);   // rounds the slack asks for, over the mean group (before rounding to even)
    // (the rows are spread EVENLY over the bands the geometry has: a band then holds a few groups of 3 rows instead of leaving the last
    // band half empty, the mean group is shorter and so is the longest -- config 4: 25 600 groups of 39.1 entries instead of 25 000 of 40.0)
    const int32_t band_groups = kMsWavesPerXcd * 8;
    const int32_t bands = passes * S;
    // (bands are ranges of the rows IN groups, in row order: inrow[] lists them)
    std::vector<int32_t> inrow;
    inrow.reserve(n_in);
    for (int32_t r = 0; r < n; ++r)
        if (degs[r] <= hub_limit) inrow.push_back(r);
    const int32_t band_rows = std::min(band_groups * 4, ((n_in + bands - 1) / bands + 3) / 4 * 4);
    const int32_t groups_all = bands * band_groups;       // (group slots; the last band may leave some empty)
    std::vector<int32_t> grow((size_t)groups_all * 4, -1);
    std::vector<int32_t> glen(groups_all, 0), gcnt(groups_all, 0);
    std::vector<int32_t> order;
    for (int32_t j = 0; j < bands; ++j) {
        const int32_t r0 = std::min(n_in, j * band_rows), r1 = std::min(n_in, r0 + band_rows);
        const int32_t ng = std::min(band_groups, r1 - r0);        // every group slot of the band gets a seed row while rows last
        order.assign(inrow.begin() + r0, inrow.begin() + r1);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return rowptr[a + 1] - rowptr[a] > rowptr[b + 1] - rowptr[b]; });
        std::priority_queue<std::pair<int32_t, int32_t>, std::vector<std::pair<int32_t, int32_t>>, std::greater<std::pair<int32_t, int32_t>>> open;
        for (int32_t i = 0; i < r1 - r0; ++i) {
            const int32_t r = order[i], d = rowptr[r + 1] - rowptr[r];
            int32_t g;
            if (i < ng) g = j * band_groups + i;
            else {
                g = open.top().second;
                open.pop();
            }
            grow[(size_t)g * 4 + gcnt[g]++] = r;
            glen[g] += d;
            if (gcnt[g] < 4) open.push({glen[g], g});
        }
        // The greedy deal leaves a few groups of a band above the others (its last rows have no choice left), and the LONGEST group of all
        // bands sets the number of rounds every wave walks.  Local repair: while a group of the band is longer than the cap (the rounds the
        // slack asks for anyway), swap one of its rows with a shorter row of the band's shortest group if that lowers the longer of the two.
        for (int32_t iter = 0; iter < 4096; ++iter) {
            int32_t hi = j * band_groups, lo = hi;
            for (int32_t g = j * band_groups; g < j * band_groups + ng; ++g) {
                if (glen[g] > glen[hi]) hi = g;
                if (glen[g] < glen[lo]) lo = g;
            }
            if (glen[hi] <= cap || glen[hi] - glen[lo] < 2) break;
            int32_t best = 0, ba = -1, bb = -1;
            const int32_t gap = glen[hi] - glen[lo];
            for (int32_t a = 0; a < gcnt[hi]; ++a)
                for (int32_t b = 0; b < gcnt[lo]; ++b) {
                    const int32_t ra = grow[(size_t)hi * 4 + a], rb = grow[(size_t)lo * 4 + b];
                    const int32_t delta = (rowptr[ra + 1] - rowptr[ra]) - (rowptr[rb + 1] - rowptr[rb]);
                    if (delta > 0 && delta < gap && std::min(delta, gap - delta) > best) best = std::min(delta, gap - delta), ba = a, bb = b;
                }
            if (ba < 0) break;
            const int32_t ra = grow[(size_t)hi * 4 + ba], rb = grow[(size_t)lo * 4 + bb];
            const int32_t delta = (rowptr[ra + 1] - rowptr[ra]) - (rowptr[rb + 1] - rowptr[rb]);
            std::swap(grow[(size_t)hi * 4 + ba], grow[(size_t)lo * 4 + bb]);
            glen[hi] -= delta;
            glen[lo] += delta;
        }
    }
    int64_t total = 0;
    int32_t longest = 0;
    for (int32_t g = 0; g < groups_all; ++g) total += glen[g], longest = std::max(longest, glen[g]);
    int32_t T = std::max(longest, cap);
    T = std::max(T, 1);
    T += T & 1;                                      // (the kernel's loop body spans two rounds when S is odd)
    if (T > kMsMaxRounds) return im;
    im.sets = S;
    im.s4 = (S + 3) / 4 * 4;
    im.passes = passes;
    im.rounds = T;
    im.real_entries = total;
    const size_t sw = im.stream_words();
    im.ent.assign((size_t)passes * kMsWavesPerXcd * sw, kMsPad);
    if (!uniform) im.val.assign(im.ent.size(), 0.f);
    im.rows.assign((size_t)passes * kMsWavesPerXcd * S * 32, kMsPad);
    // groups of a band -> (wave, position): consecutive groups go to consecutive WAVES first, then positions
    struct Ent { int32_t src; int32_t slot; float v; };
    std::vector<Ent> list;
    std::vector<int32_t> f, bk;
    for (int32_t g = 0; g < groups_all; ++g) {
        if (gcnt[g] == 0) continue;
        const int32_t band = g / band_groups, gl = g % band_groups;
        const int32_t wave = gl % kMsWavesPerXcd, p = gl / kMsWavesPerXcd;
        const int32_t pass = band / S, set = S - 1 - band % S;            // low bands = last sets
        list.clear();
        for (int32_t slot = 0; slot < 4; ++slot) {
            const int32_t r = grow[(size_t)g * 4 + slot];
            if (r < 0) continue;
            im.rows[(((size_t)pass * kMsWavesPerXcd + wave) * S + set) * 32 + p * 4 + slot] = (uint32_t)r << 7;
            for (int32_t q = rowptr[r]; q < rowptr[r + 1]; ++q) list.push_back({col[q], slot, uniform ? 1.f : val[q]});
        }
        std::stable_sort(list.begin(), list.end(), [](const Ent& x, const Ent& y) { return x.src < y.src; });
        const int32_t L = (int32_t)list.size();
        // placement: round ~ source * T / n, strictly increasing.  f = earliest-feasible walk from the front (>= ideal), bk = latest-
        // feasible walk from the back (<= ideal), both clamped to [k, T - (L - k)]; their midpoint is strictly increasing and halves the
        // displacement of either.
        f.resize(L);
        bk.resize(L);
        int32_t prev = -1;
        for (int32_t k = 0; k < L; ++k) {
            const int32_t ideal = (int32_t)((int64_t)list[k].src * T / n);
            prev = std::max(ideal, prev + 1);
            f[k] = prev;
        }
        int32_t nxt = T;
        for (int32_t k = L - 1; k >= 0; --k) {
            nxt = std::min(f[k], nxt - 1);
            f[k] = nxt;
        }
        nxt = T;
        for (int32_t k = L - 1; k >= 0; --k) {
            const int32_t ideal = (int32_t)((int64_t)list[k].src * T / n);
            nxt = std::min(ideal, nxt - 1);
            bk[k] = nxt;
        }
        prev = -1;
        for (int32_t k = 0; k < L; ++k) {
            prev = std::max(bk[k], prev + 1);
            bk[k] = prev;
        }
        uint32_t* e = im.ent.data() + ((size_t)pass * kMsWavesPerXcd + wave) * sw;
        float* v = uniform ? nullptr : im.val.data() + ((size_t)pass * kMsWavesPerXcd + wave) * sw;
        for (int32_t k = 0; k < L; ++k) {
            const int32_t t = (f[k] + bk[k]) >> 1;
            const size_t at = im.at(t, p, set);
            e[at] = ((uint32_t)list[k].src << 7) | (1u << list[k].slot);
            if (v) v[at] = list[k].v;
        }
    }
    // hub rows -> quads.  Ordinary hub rows: longest first, 32 per quad.  A row longer than hub_split is SPLIT over the 32 slots of a quad of
    // its own (slot s = 4 p + o takes the row's entries s Lq .. s Lq + Lq - 1, Lq = ceil(length / 32)): a wave must not spend 20 times the
    // other waves' hub time on one row; its 32 partial chains are added in a fixed tree (octets in order, then positions by xor 1, 2, 4): the
    // one place where a row's sum is not the single ascending-column chain (deterministic; ~1e-7 relative to it).  Quads go to the waves
    // longest first, each to the wave with the fewest hub steps so far.
    im.hub_limit = hub_limit;
    std::vector<int32_t> hubs;
    for (int32_t r = 0; r < n; ++r)
        if (degs[r] > hub_limit) hubs.push_back(r), im.hub_entries += degs[r];
    im.hub_rows = (int32_t)hubs.size();
    if (!hubs.empty()) {
        std::stable_sort(hubs.begin(), hubs.end(), [&](int32_t a, int32_t b) { return degs[a] > degs[b]; });
        const int32_t nw = passes * kMsWavesPerXcd;
        im.hub_split = std::max<int32_t>(32, (int32_t)(2 * (im.hub_entries / 32 / nw)));
        struct Quad { int32_t L, first, count, split; };          // rows hubs[first .. first + count)
        std::vector<Quad> quads;
        size_t i = 0;
        for (; i < hubs.size() && degs[hubs[i]] > im.hub_split; ++i) quads.push_back({(degs[hubs[i]] + 31) / 32, (int32_t)i, 1, 1}), ++im.hub_split_rows;
        for (; i < hubs.size(); i += 32) quads.push_back({degs[hubs[i]], (int32_t)i, (int32_t)std::min<size_t>(32, hubs.size() - i), 0});
        std::stable_sort(quads.begin(), quads.end(), [](const Quad& x, const Quad& y) { return x.L > y.L; });
        std::vector<std::vector<int32_t>> quads_of(nw);
        std::priority_queue<std::pair<int64_t, int32_t>, std::vector<std::pair<int64_t, int32_t>>, std::greater<std::pair<int64_t, int32_t>>> load;
        for (int32_t w = 0; w < nw; ++w) load.push({0, w});
        for (int32_t q = 0; q < (int32_t)quads.size(); ++q) {
            auto [l, w] = load.top();
            load.pop();
            quads_of[w].push_back(q);
            load.push({l + quads[q].L + 2, w});
        }
        im.hubptr.assign(nw + 1, 0);
        for (int32_t w = 0; w < nw; ++w) {
            im.hubptr[w] = (uint32_t)im.hub.size();
            for (int32_t q : quads_of[w]) {
                const Quad& Q = quads[q];
                const size_t base = im.hub.size();
                im.hub.resize(base + 36 + (size_t)Q.L * 32, kMsPad);
                if (!uniform) im.hubval.resize(im.hub.size(), 0.f);
                im.hub[base] = (uint32_t)Q.L;
                im.hub[base + 1] = (uint32_t)Q.split;
                im.hub[base + 2] = im.hub[base + 3] = 0u;
                if (Q.split) {
                    const int32_t r = hubs[Q.first];
                    im.hub[base + 4] = (uint32_t)r << 7;                       // (position 0, octet 0 stores the row)
                    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                        const int32_t idx = k - rowptr[r], slot = idx / Q.L, j = idx % Q.L;
                        const size_t at = base + 36 + (size_t)j * 32 + slot;
                        im.hub[at] = (uint32_t)col[k] << 7;
                        if (!uniform) im.hubval[at] = val[k];
                    }
                    continue;
                }
                for (int32_t ii = 0; ii < Q.count; ++ii) {
                    const int32_t r = hubs[(size_t)Q.first + ii], o = ii / 8, pp = ii % 8;   // octet o of the quad, position pp
                    im.hub[base + 4 + 4 * pp + o] = (uint32_t)r << 7;
                    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                        const size_t at = base + 36 + (size_t)(k - rowptr[r]) * 32 + 4 * pp + o;
                        im.hub[at] = (uint32_t)col[k] << 7;
                        if (!uniform) im.hubval[at] = val[k];
                    }
                }
            }
        }
        im.hubptr[nw] = (uint32_t)im.hub.size();
    }
    return im;
}

// What spmm_msweep_kernel computes for ONE batch entry, in its own order of operations (per accumulator: fmaf in round order; the
// slots a step does not address see fmaf(0, x, acc) = acc).  uniform: sum, then scale once by uval.
inline void interpret_msweep_image(const MsweepImage& im, bool uniform, float uval, const float* X, float* Y, int32_t W) {
    const size_t sw = im.stream_words();
    std::vector<float> acc((size_t)im.sets * 32 * W);
    for (int32_t pass = 0; pass < im.passes; ++pass)
        for (int32_t wave = 0; wave < kMsWavesPerXcd; ++wave) {
            std::fill(acc.begin(), acc.end(), 0.f);
            const uint32_t* e = im.ent.data() + ((size_t)pass * kMsWavesPerXcd + wave) * sw;
            const float* v = uniform ? nullptr : im.val.data() + ((size_t)pass * kMsWavesPerXcd + wave) * sw;
            for (int32_t t = 0; t < im.rounds + 2; ++t)
                for (int32_t s = 0; s < im.sets; ++s)
                    for (int32_t p = 0; p < 8; ++p) {
                        const size_t at = im.at(t, p, s);
                        const uint32_t w = e[at];
                        if (!(w & 15u)) continue;
                        const float a = uniform ? 1.f : v[at];
                        const float* x = X + (size_t)(w >> 7) * W;
                        for (int32_t slot = 0; slot < 4; ++slot)
                            if (w & (1u << slot)) {
                                float* d = acc.data() + ((size_t)s * 32 + p * 4 + slot) * W;
                                for (int32_t k = 0; k < W; ++k) d[k] = fmaf(a, x[k], d[k]);
                            }
                    }
            for (int32_t s = 0; s < im.sets; ++s)
                for (int32_t j = 0; j < 32; ++j) {
                    const uint32_t ro = im.rows[(((size_t)pass * kMsWavesPerXcd + wave) * im.sets + s) * 32 + j];
                    if (ro == kMsPad) continue;
                    const float* d = acc.data() + ((size_t)s * 32 + j) * W;
                    for (int32_t k = 0; k < W; ++k) Y[(size_t)(ro >> 7) * W + k] = uniform ? d[k] * uval : d[k];
                }
        }
    // hub phase: per wave its quads, per (octet, position) the steps in order (gaps add 0 * 0)
    const int32_t nw = im.hubptr.empty() ? 0 : (int32_t)im.hubptr.size() - 1;
    std::vector<float> a((size_t)32 * W);
    for (int32_t w = 0; w < nw; ++w)
        for (size_t at = im.hubptr[w]; at < im.hubptr[w + 1];) {
            const int32_t Lq = (int32_t)im.hub[at];
            const bool split = im.hub[at + 1] != 0;
            std::fill(a.begin(), a.end(), 0.f);
            for (int32_t slot = 0; slot < 32; ++slot)              // slot = 4 p + o
                for (int32_t j = 0; j < Lq; ++j) {
                    const size_t e = at + 36 + (size_t)j * 32 + slot;
                    if (im.hub[e] == kMsPad) continue;
                    const float* x = X + (size_t)(im.hub[e] >> 7) * W;
                    const float v = uniform ? 1.f : im.hubval[e];
                    for (int32_t k = 0; k < W; ++k) a[(size_t)slot * W + k] = fmaf(v, x[k], a[(size_t)slot * W + k]);
                }
            if (split) {                                            // octets in order, then positions by xor 1, 2, 4 (both partners compute a + b: commutative)
                std::vector<float> ps((size_t)8 * W);
                for (int32_t pp = 0; pp < 8; ++pp)
                    for (int32_t k = 0; k < W; ++k)
                        ps[(size_t)pp * W + k] = ((a[(size_t)(4 * pp) * W + k] + a[(size_t)(4 * pp + 1) * W + k]) + a[(size_t)(4 * pp + 2) * W + k]) + a[(size_t)(4 * pp + 3) * W + k];
                for (int32_t m = 1; m < 8; m <<= 1) {
                    std::vector<float> nx(ps);
                    for (int32_t pp = 0; pp < 8; ++pp)
                        for (int32_t k = 0; k < W; ++k) nx[(size_t)pp * W + k] = ps[(size_t)pp * W + k] + ps[(size_t)(pp ^ m) * W + k];
                    ps.swap(nx);
                }
                const uint32_t ro = im.hub[at + 4];
                for (int32_t k = 0; k < W; ++k) Y[(size_t)(ro >> 7) * W + k] = uniform ? ps[k] * uval : ps[k];
            } else {
                for (int32_t slot = 0; slot < 32; ++slot) {
                    const uint32_t ro = im.hub[at + 4 + slot];
                    if (ro == kMsPad) continue;
                    for (int32_t k = 0; k < W; ++k) Y[(size_t)(ro >> 7) * W + k] = uniform ? a[(size_t)slot * W + k] * uval : a[(size_t)slot * W + k];
                }
            }
            at += 36 + (size_t)Lq * 32;
        }
}

// Diagnostic: L2 hit rate of the row gathers of one batch entry if the XCD's 128 waves advance in lock step (round by round, set by
// set) and the L2 is an LRU cache of `lines` 128-byte lines; the entry stream's own lines pass through the same cache.
inline double simulate_msweep_hits(const MsweepImage& im, int32_t n, int32_t lines) {
    const size_t sw = im.stream_words();
    std::vector<int64_t> stamp(n, -1);               // time of the row's last use while it is resident, -1 = not resident
    std::vector<std::pair<int64_t, int32_t>> fifo;   // (stamp, row) in time order; stale pairs (stamp changed since) are skipped at eviction
    size_t head = 0;
    int64_t resident = 0, now = 0, hits = 0, misses = 0;
    auto evict = [&]() {
        while (resident > lines && head < fifo.size()) {
            const auto [t, r] = fifo[head++];
            if (r < 0) { --resident; continue; }     // a line of the entry stream
            if (stamp[r] == t) stamp[r] = -1, --resident;
        }
    };
    const int64_t stream_per_round = (int64_t)kMsWavesPerXcd * 1024 / 128;
    for (int32_t pass = 0; pass < im.passes; ++pass)
        for (int32_t t = 0; t < im.rounds; ++t) {
            for (int64_t i = 0; i < stream_per_round; ++i) fifo.push_back({++now, -1}), ++resident;
            for (int32_t s = 0; s < im.sets; ++s)
                for (int32_t wave = 0; wave < kMsWavesPerXcd; ++wave) {
                    const uint32_t* e = im.ent.data() + ((size_t)pass * kMsWavesPerXcd + wave) * sw;
                    for (int32_t p = 0; p < 8; ++p) {
                        const uint32_t w = e[im.at(t, p, s)];
                        if (!(w & 15u)) continue;
                        const int32_t r = (int32_t)(w >> 7);
                        if (stamp[r] >= 0) ++hits;
                        else ++misses, ++resident;
                        stamp[r] = ++now;
                        fifo.push_back({now, r});
                    }
                }
            evict();
        }
    return hits + misses ? (double)hits / (double)(hits + misses) : 0.0;
}
