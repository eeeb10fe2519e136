// xcd_sweep.hip -- go / no-go prototype for the node-major hop at N = 1e5 (config 4): "XCD-synchronous source sweep".
//     hipcc -O3 --offload-arch=gfx950 tools/xcd_sweep.hip -o tools/xcd_sweep && tools/xcd_sweep [key=value ...]
//
// What round 2 left open ("next round" 1d): the gathers of a hop (nnz * B rows of 128 bytes = 16.4 GB) miss the 4 MiB L2
// because a batch entry's rows are 12.8 MB; tools/l2window_bound.hip showed that source-sorted edge streams alone form only half a
// window (hit rate 0.45) and that XCD BARRIERS cost more than they give (they serialise the store / zero / gather phases).
// This prototype keeps that kernel's mechanics (destination tile accumulators in LDS, each 8-lane group owns rows and walks their
// entries sorted by source, LDS read-modify-write, fixed order, no atomics) and changes what decides the hit rate:
//   * PERSISTENT workgroups (LDS-limited residency: exactly wgPerCU per CU), an XCD walks its batch entries one at a time, P passes
//     per entry, one destination tile per workgroup and pass;
//   * padding is spread EVENLY over a stream (entry with source j sits near slot j * L / N), so "step s" means "sources around s*N/L"
//     for every lane group of every workgroup, whatever its list length;
//   * BOUNDED LAG instead of barriers: every workgroup publishes its step counter (one write-through dword), every wave polls the
//     XCD's counters one step ahead of use (a 256/512-byte read) and only a wave that is more than `lag` steps ahead of the slowest
//     workgroup waits (bounded spin: correctness never depends on the protocol, a broken protocol costs time only);
//   * two COHORTS half a sweep apart (optional), so that one workgroup of a CU stores its tile while the other gathers;
//   * optional FIRST-TOUCH PREFETCH: every wave reads one coalesced KB of the source block `pf` steps ahead, so the first gather of a
//     row finds it in L2 (the fabric carries the same bytes, as full sequential lines instead of random demand misses).
// Compared against: tools/l2window_bound (1.73 ms), the library's spmm_sell_kernel (1.60 ms), the algorithmic bound (0.41 ms).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr unsigned kPad = 0xffffffffu;
constexpr int kWaves = 7;         // WORKER waves per workgroup (+ one sync wave = 512 threads: 2 waves per SIMD and workgroup)
constexpr int kThreads = (kWaves + 1) * 64;   // kWaves worker waves + the sync wave
constexpr int kSlots = 128;       // progress slots per XCD (>= workgroups per XCD)
constexpr int kMaxSpin = 3000;

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned wave_min(unsigned v) {
#pragma unroll
    for (int m = 32; m; m >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, m));
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

// stream[((tile * kWaves + wave) * L4 + s) * 8 + lg] : u32x4 = 4 consecutive slots of lane group lg; slot = dest_local << 17 | src, or kPad
// LDS accumulators: row dl at byte dl * 128; plan-time rule "lane groups {0,1,4,5} own even rows, {2,3,6,7} odd rows": the two rows a
// ds_read_b128 / ds_write_b128 service group touches then lie in different bank halves (conflict-free).
// ACCUM: 0 = gather only, 1 = read-modify-write per entry in program order, 3 = the four rows of a word are read together, duplicates
//        (two entries of a word with the same destination) are merged in registers, then written in order (the last write carries
//        every contribution).  (ds_add_f32 was measured: 165 clocks per wave instruction, 26 ms per hop -- LDS float atomics serialise.)
constexpr int kRow = 32;
template <int ACCUM, int MINW, int DEPTH>
__global__ __launch_bounds__(kThreads, MINW) void sweep_kernel(const u32x4* __restrict__ stream, const float* __restrict__ X,
                                                                  float* __restrict__ Y, int N, int D, int L4, int P, int wgPerXcd,
                                                                  int nTiles, int entriesPerXcd, int B, float uval,
                                                                  unsigned* __restrict__ prog, unsigned base, int lag,
                                                                  unsigned* __restrict__ stats, int shift) {
    extern __shared__ __attribute__((aligned(16))) float accs[];   // [(D + 8) * kRow]
    __shared__ unsigned s_prog[kWaves + 1];   // [kWaves] = the XCD's floor as last seen by a polling wave of this workgroup
    auto lds_st = [&](int i, unsigned v) { __hip_atomic_store(&s_prog[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_ld = [&](int i) { return __hip_atomic_load(&s_prog[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    const int tid = threadIdx.x, lane = tid & 63, sub = lane & 7, lg = lane >> 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    unsigned short* slots = reinterpret_cast<unsigned short*>(prog) + xcd * kSlots;   // 16-bit progress (steps since `base`) per workgroup
    const unsigned* pollp = prog + xcd * (kSlots / 2) + (lane & (wgPerXcd > 64 ? 63 : 31));   // one dword = two slots per lane: 1-2 lines
    unsigned* xfloor = prog + 8 * (kSlots / 2) + xcd * 32;   // the XCD's floor, kept by the sync waves: one word, own line
    // two COHORTS: odd workgroups count their steps from `shift` (half a sweep), i.e. they are let go once the even ones are half a sweep
    // in, and from then on one cohort stores its tiles while the other gathers (all workgroups storing at the same moment stalls the
    // TCP's store-data path for 27 % of the kernel: TCP_TCP_TA_DATA_STALL_CYCLES, profiles/r03_a_sweep/v6_tcc_counters.log)
    unsigned gi = (j & 1) ? (unsigned)shift : 0u;   // index (since launch, in the shifted coordinate) of the step whose gathers are issued next
    unsigned pv = 0;
    unsigned waited = 0, gaveup = 0;

    // progress: a worker wave writes its own counter to LDS (lgkmcnt traffic) and polls the XCD's counters with ONE unconditional
    // L1-bypassing load per step, issued ahead of its gathers and used a step later: its vmcnt queue holds loads only and every path
    // issues the same number of them, so the compiler counts exactly and two steps of gathers stay in flight (a publish STORE inside the
    // loop, or a load behind a branch, makes it wait with vmcnt(0..1)).  The SYNC wave (wave kWaves) of the workgroup does nothing but
    // publish the workgroup's minimum whenever it changes (workgroup-scope 16-bit store: the line stays in this XCD's L2).
    auto publish = [&]() {
        if (lane == 0) lds_st(wave, gi);
    };
    auto wg_min = [&]() -> unsigned {
        unsigned v = lds_ld(min(lane & 7, kWaves - 1));
        v = min(v, (unsigned)__shfl_xor((int)v, 1));
        v = min(v, (unsigned)__shfl_xor((int)v, 2));
        v = min(v, (unsigned)__shfl_xor((int)v, 4));
        return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    // A worker polls ONE word (every lane the same address: one request, no reduction): the minimum over the XCD's counters is taken
    // by the sync waves.  (Every worker reducing 64-128 counters itself costs 6 dependent ds_bpermute per step: with 14 worker
    // waves per CU the per-step latency chain, not memory, bounded the prototype -- 62 L2 requests in flight per CU, counters.)
    auto issue_poll = [&]() { pv = ld_sc1(xfloor); };
    auto floor_of = [&]() { return (unsigned)__builtin_amdgcn_readfirstlane((int)pv); };
    auto wait_floor = [&]() {   // gate the gathers of step gi: at most `lag` steps ahead of the slowest workgroup of the XCD
        unsigned fl = floor_of();   // the poll issued a step ago (always consumed: keeps the load count uniform)
        issue_poll();
        if (lag < 0) return;
        int guard = 0;
        while ((int)(gi - fl) > lag) {
            if (++guard > kMaxSpin) { lag = -1; gaveup = 1; break; }
            __builtin_amdgcn_s_sleep(8);
            fl = floor_of();
            issue_poll();
        }
        waited += (unsigned)guard;
    };
    auto gather = [&](const float* Xb, const u32x4& w, f32x4 (&x)[4]) {
        const unsigned ee[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)   // unconditional: a padding slot gathers row 0 (an L1 hit) into the lane group's trash row, so every
            x[u] = *reinterpret_cast<const f32x4*>(Xb + ((ee[u] & 0x1ffffu) * 32u + (unsigned)sub * 4u));   // path issues the same number of loads and the compiler can count vmcnt exactly
    };
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    f32x4* acc4 = reinterpret_cast<f32x4*>(accs);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto pack_dst = [&](const u32x4& w) -> u32x2 {   // the 4 destination rows of a word in 2 registers: a step in flight keeps 18 VGPRs, not 20
        return (u32x2){(w.x >> 17) | ((w.y >> 17) << 16), (w.z >> 17) | ((w.w >> 17) << 16)};
    };
    auto accumulate = [&](const u32x2& dp, const f32x4 (&x)[4]) {
        const int d[4] = {(int)(dp.x & 0xffffu) * 8 + sub, (int)(dp.x >> 16) * 8 + sub, (int)(dp.y & 0xffffu) * 8 + sub, (int)(dp.y >> 16) * 8 + sub};
        if (ACCUM == 3) {
            f32x4 a[4], sx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = acc4[d[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sx[u] = x[u];
#pragma unroll
                for (int v = 0; v < u; ++v) {   // fixed order: contributions of earlier slots first
                    const float m = (d[v] == d[u]) ? 1.f : 0.f;
                    sx[u] += x[v] * m;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc4[d[u]] = a[u] + sx[u];
        } else if (ACCUM == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc4[d[u]] += x[u];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) sink += x[u];
        }
    };

    if (stats && tid == 0) {   // residency check: spread of the workgroups' start times (100 MHz ticks)
        const unsigned long long t0 = wall_clock64();
        atomicMin(reinterpret_cast<unsigned long long*>(stats + 4), t0);
        atomicMax(reinterpret_cast<unsigned long long*>(stats + 6), t0);
    }
    for (int i = tid; i < (D + 8) * kRow; i += kThreads) accs[i] = 0.f;
    if (lane == 0 && wave < kWaves) lds_st(wave, gi);
    __syncthreads();
    issue_poll();
    unsigned long long tsync = 0, rounds = 0;

    for (int e = 0; e < entriesPerXcd; ++e) {
        const int b = e * 8 + xcd;
        for (int p = 0; p < P; ++p) {
            const int tile = p * wgPerXcd + j;
            const bool live = tile < nTiles && b < B;
            if (wave == kWaves) {   // ---- the sync wave: publish until every worker has issued the last gathers of this sweep
                const unsigned target = gi + (unsigned)L4;
                if (lag >= 0) {
                    unsigned last = 0xffffffffu;
                    const unsigned long long t0 = wall_clock64();
                    unsigned lastfl = 0xffffffffu;
                    for (int guard = 0; guard < (1 << 22); ++guard) {
                        const unsigned v = wg_min();
                        if (v != last && lane == 0)
                            __hip_atomic_store(slots + j, (unsigned short)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        last = v;
                        ++rounds;
                        {   // every sync wave refreshes the XCD's floor (a stale lower value overwriting a newer one is conservative, never wrong)
                            const unsigned q = ld_sc1(pollp);
                            const unsigned fl = wave_min(min(q & 0xffffu, q >> 16));
                            if (fl != lastfl && lane == 0) __hip_atomic_store(xfloor, fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            lastfl = fl;
                        }
                        if ((int)(v - target) >= 0) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    tsync += wall_clock64() - t0;
                }
                gi = target;
            } else if (!live) {     // nothing to do in this pass: do not hold the others back
                gi += (unsigned)L4;
                publish();
            } else {
                const float* Xb = X + (int64_t)b * N * 32;
                const u32x4* st = stream + ((int64_t)(tile * kWaves + wave) * L4) * 8 + lg;
                if (DEPTH == 4) {   // three steps of gathers in flight behind the one being accumulated (16 loads per lane); a step in
                    // flight keeps its 16 data registers and its destinations packed in 2 -- the source words are dead once the gathers are issued
                    f32x4 xa[4], xb[4], xc[4], xd[4];
                    u32x2 da, db, dc, dd;
                    {
                        const u32x4 w0 = st[0], w1 = st[8], w2 = st[16];
                        wait_floor(); gather(Xb, w0, xa); da = pack_dst(w0); gi += 1u; publish();
                        wait_floor(); gather(Xb, w1, xb); db = pack_dst(w1); gi += 1u; publish();
                        wait_floor(); gather(Xb, w2, xc); dc = pack_dst(w2); gi += 1u; publish();
                    }
                    u32x4 w3 = st[24];
                    for (int s = 0; s < L4; s += 4) {   // L4 is a multiple of 4
                        const u32x4 w4 = st[(int64_t)(s + 4 < L4 ? s + 4 : L4 - 1) * 8];
                        const u32x4 w5 = st[(int64_t)(s + 5 < L4 ? s + 5 : L4 - 1) * 8];
                        wait_floor(); gather(Xb, w3, xd); dd = pack_dst(w3); gi += 1u; publish();
                        accumulate(da, xa);
                        const u32x4 w6 = st[(int64_t)(s + 6 < L4 ? s + 6 : L4 - 1) * 8];
                        wait_floor(); gather(Xb, w4, xa); da = pack_dst(w4); gi += (s + 4 < L4) ? 1u : 0u; publish();
                        accumulate(db, xb);
                        const u32x4 w7 = st[(int64_t)(s + 7 < L4 ? s + 7 : L4 - 1) * 8];
                        wait_floor(); gather(Xb, w5, xb); db = pack_dst(w5); gi += (s + 5 < L4) ? 1u : 0u; publish();
                        accumulate(dc, xc);
                        wait_floor(); gather(Xb, w6, xc); dc = pack_dst(w6); gi += (s + 6 < L4) ? 1u : 0u; publish();
                        accumulate(dd, xd);
                        w3 = w7;
                    }
                } else {
                u32x4 w0 = st[0], w1 = st[8];
                f32x4 xa[4], xb[4];
                wait_floor();
                gather(Xb, w0, xa);
                gi += 1u;
                publish();
                for (int s = 0; s < L4; s += 2) {   // L4 is even; two steps per round, two gather buffers, no register copies of x and
                    // no conditional loads (the last round gathers the last word once more and drops it): exact vmcnt counting
                    const u32x4 w2 = st[(int64_t)(s + 2 < L4 ? s + 2 : L4 - 1) * 8];
                    const u32x4 w3 = st[(int64_t)(s + 3 < L4 ? s + 3 : L4 - 1) * 8];
                    wait_floor();
                    gather(Xb, w1, xb);              // step s + 1 in flight ...
                    gi += 1u;
                    publish();
                    accumulate(pack_dst(w0), xa);    // ... while step s is accumulated
                    wait_floor();
                    gather(Xb, w2, xa);
                    gi += (s + 2 < L4) ? 1u : 0u;
                    publish();
                    accumulate(pack_dst(w1), xb);
                    w0 = w2;
                    w1 = w3;
                }
                }
            }
            if (!live) continue;
            __syncthreads();
            float* Yb = Y + (int64_t)b * N * 32 + (int64_t)tile * D * 32;
            const int rows = min(D, N - tile * D);
            for (int i = tid; i < rows * 8; i += kThreads) {
                f32x4 v = acc4[i];
                acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                v *= uval;
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Yb) + i);
            }
            __syncthreads();
        }
    }
    __syncthreads();   // finished: never the minimum again (the host resets the slots before the next launch)
    if (wave == kWaves && lane == 0) __hip_atomic_store(slots + j, (unsigned short)0xffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    sink.x += (float)(pv & 1u) * 1e-30f;
    if (stats && wave == kWaves && lane == 0) {
        atomicAdd(stats + 2, (unsigned)rounds);
        atomicAdd(stats + 3, (unsigned)(tsync >> 4));
    }
    if (sink.x == 1.2345e30f && sink.y == -7.f) Y[0] = sink.z + sink.w;
    if (stats && lane == 0) {
        atomicAdd(stats + 0, waited);
        atomicAdd(stats + 1, gaveup);
    }
}

struct Graph {
    int N;
    std::vector<std::vector<int>> nbr;   // nbr[dest] = sorted sources
    int64_t nnz;
};

static Graph make_er(int N, int64_t nnzTarget, unsigned seed) {
    Graph G;
    G.N = N;
    G.nbr.resize(N);
    std::mt19937 rng(seed);
    std::uniform_int_distribution<int> pick(0, N - 1);
    for (int64_t k = 0; k < nnzTarget / 2; ++k) {
        const int i = pick(rng), j = pick(rng);
        if (i == j) continue;
        G.nbr[i].push_back(j);
        G.nbr[j].push_back(i);
    }
    G.nnz = 0;
    for (auto& v : G.nbr) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        G.nnz += (int64_t)v.size();
    }
    return G;
}

struct Streams {
    int D, P, wgPerXcd, nTiles, L4;
    double fill;
    std::vector<unsigned> words;
};

// distinct = 1: the 4 slots of a word never hold the same destination twice (ACCUM == 2 may batch its LDS reads)
static Streams build_streams(const Graph& G, int wgPerCU, int P, bool distinct) {
    Streams S;
    const int N = G.N;
    S.wgPerXcd = 32 * wgPerCU;
    S.P = P;
    S.D = ((N + P * S.wgPerXcd - 1) / (P * S.wgPerXcd) + 1) & ~1;   // even: the trash rows D .. D + 7 keep the bank rule
    S.nTiles = (N + S.D - 1) / S.D;
    const int D = S.D, nLG = kWaves * 8;
    // rows of a tile -> lane groups: even local rows to lane groups lg in {0,1,4,5}, odd ones to {2,3,6,7} (bank rule of the LDS
    // accumulators), dealt by degree, boustrophedon
    std::vector<std::vector<unsigned>> lists((size_t)S.nTiles * nLG);
    int maxLen = 0;
    for (int t = 0; t < S.nTiles; ++t) {
        for (int par = 0; par < 2; ++par) {
            std::vector<int> rows;
            for (int dl = par; dl < D && t * D + dl < N; dl += 2) rows.push_back(dl);
            std::stable_sort(rows.begin(), rows.end(), [&](int a, int c) { return G.nbr[t * D + a].size() > G.nbr[t * D + c].size(); });
            std::vector<int> groups;
            for (int w = 0; w < kWaves; ++w)
                for (int lgx : {0, 1, 4, 5}) groups.push_back(w * 8 + lgx + 2 * par);
            const int ng = (int)groups.size();
            for (size_t k = 0; k < rows.size(); ++k) {
                const int round = (int)(k / ng), pos = (int)(k % ng);
                const int q = groups[(round & 1) ? ng - 1 - pos : pos];
                for (int src : G.nbr[t * D + rows[k]]) lists[(size_t)t * nLG + q].push_back(((unsigned)rows[k] << 17) | (unsigned)src);
            }
        }
        for (int q = 0; q < nLG; ++q) {
            auto& li = lists[(size_t)t * nLG + q];
            std::sort(li.begin(), li.end(), [](unsigned a, unsigned c) { return (a & 0x1ffffu) < (c & 0x1ffffu); });
            maxLen = std::max(maxLen, (int)li.size());
        }
    }
    // even placement: entry with source j near slot j * alpha * (Lslots - 4) / N, never before its predecessor; alpha = the largest
    // of 1, 0.96, 0.92, ... for which the list fits (a list that is dense near the end runs slightly ahead of its position)
    for (int L4 = ((maxLen + 3) / 4 + 4) & ~3;; L4 += 4) {
        const int Ls = L4 * 4;
        std::vector<unsigned> words((size_t)S.nTiles * kWaves * L4 * 8 * 4, kPad);
        bool ok = true;
        std::vector<int> slotOf;
        for (int t = 0; t < S.nTiles && ok; ++t)
            for (int q = 0; q < nLG && ok; ++q) {
                const auto& li = lists[(size_t)t * nLG + q];
                const int w = q / 8, lgx = q % 8;
                bool placed = false;
                const int n = (int)li.size();
                for (int ai = 25; ai >= 0 && !placed; --ai) {   // word by word: up to 4 pending entries whose nominal slot has come, distinct
                    slotOf.assign(n, -1);                      // destinations inside a word (look-ahead of 12 pending entries)
                    int head = 0, left = n;
                    for (int wd = 0; wd < L4 && left > 0; ++wd) {
                        unsigned dests[4];
                        int c = 0;
                        for (int k = head, seen = 0; k < n && c < 4 && seen < 12; ++k) {
                            if (slotOf[k] >= 0) continue;
                            ++seen;
                            if ((int64_t)(li[k] & 0x1ffffu) * (Ls - 4) * ai / 25 / N > wd * 4 + 3) break;
                            bool clash = false;
                            if (distinct)
                                for (int u = 0; u < c; ++u) clash |= dests[u] == (li[k] >> 17);
                            if (clash) continue;
                            dests[c] = li[k] >> 17;
                            slotOf[k] = wd * 4 + c++;
                            --left;
                        }
                        while (head < n && slotOf[head] >= 0) ++head;
                    }
                    if (left > 0) continue;
                    placed = true;
                    for (int k = 0; k < n; ++k)
                        words[((((size_t)(t * kWaves + w) * L4 + slotOf[k] / 4) * 8 + lgx) * 4) + (slotOf[k] & 3)] = li[k];
                }
                if (!placed) ok = false;
            }
        if (ok) {
            for (size_t i = 0; i < words.size(); ++i)   // padding slot of lane group lg: gather row 0 into trash row D + lg
                if (words[i] == kPad) {
                    static const int trash[8] = {0, 2, 1, 3, 4, 6, 5, 7};   // even trash rows for lane groups 0,1,4,5, odd ones for 2,3,6,7
                    words[i] = (unsigned)(D + trash[(i / 4) % 8]) << 17;
                }
            S.L4 = L4;
            S.words.swap(words);
            break;
        }
    }
    S.fill = (double)G.nnz / ((double)S.nTiles * nLG * S.L4 * 4);
    return S;
}

int main(int argc, char** argv) {
    int N = 100000, B = 128, iters = 5, only = -1, hostcheck = 0, depth = 4;
    int64_t nnzTarget = 1000000;
    std::vector<int> wgs = {2, 3}, lags = {-1, 1, 2, 3, 4, 6, 8}, shifts = {0}, pfs = {0}, accums = {3, 1, 0};
    int P = 3;
    auto parse_list = [](const char* s) {
        std::vector<int> v;
        for (const char* p = s; *p;) {
            v.push_back(atoi(p));
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        return v;
    };
    for (int a = 1; a < argc; ++a) {
        std::string kv = argv[a];
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
        if (k == "N") N = atoi(v.c_str());
        else if (k == "B") B = atoi(v.c_str());
        else if (k == "nnz") nnzTarget = atoll(v.c_str());
        else if (k == "iters") iters = atoi(v.c_str());
        else if (k == "P") P = atoi(v.c_str());
        else if (k == "wg") wgs = parse_list(v.c_str());
        else if (k == "lag") lags = parse_list(v.c_str());
        else if (k == "shift") shifts = parse_list(v.c_str());
        else if (k == "pf") pfs = parse_list(v.c_str());
        else if (k == "accum") accums = parse_list(v.c_str());
        else if (k == "only") only = atoi(v.c_str());
        else if (k == "depth") depth = atoi(v.c_str());
        else if (k == "hostcheck") hostcheck = atoi(v.c_str());
    }
    Graph G = make_er(N, nnzTarget, 0);
    const float uval = 0.1f;
    const double algBytes = 2.0 * B * N * 32 * 4 + G.nnz * 8.0 + (N + 1) * 4.0;
    printf("ER graph N=%d nnz=%lld, B=%d batch entries x 32 columns (row = 128 B); algorithmic bytes/hop = %.3f GB (0.41 ms at 8 TB/s)\n", N,
           (long long)G.nnz, B, algBytes / 1e9);

    if (hostcheck) {   // no GPU needed: interpret the streams on the host (one column) and compare with the direct sums
        for (int wgPerCU : wgs)
            for (int distinct = 0; distinct >= 0; --distinct) {
                Streams S = build_streams(G, wgPerCU, P, distinct != 0);
                std::vector<double> xs(N), y(N, 0.0), yr(N, 0.0);
                for (int i = 0; i < N; ++i) xs[i] = std::sin(0.001 * i) + 1.0;
                int64_t cnt = 0, clash = 0;
                double posErr = 0.0;
                for (int t = 0; t < S.nTiles; ++t)
                    for (int w = 0; w < kWaves; ++w)
                        for (int s = 0; s < S.L4; ++s)
                            for (int lgx = 0; lgx < 8; ++lgx) {
                                const unsigned* wd = &S.words[((((size_t)(t * kWaves + w) * S.L4 + s) * 8 + lgx) * 4)];
                                for (int u = 0; u < 4; ++u) {
                                    if ((int)(wd[u] >> 17) >= S.D) continue;
                                    const int dl = (int)(wd[u] >> 17), src = (int)(wd[u] & 0x1ffffu);
                                    if ((dl & 1) != ((lgx >> 1) & 1)) { printf("bank-rule violation\n"); return 1; }
                                    y[(size_t)t * S.D + dl] += xs[src];
                                    ++cnt;
                                    posErr = std::max(posErr, std::fabs((double)src / N - (double)(s * 4 + u) / (S.L4 * 4)));
                                    for (int u2 = 0; u2 < u; ++u2) if ((wd[u2] >> 17) == (unsigned)dl) ++clash;
                                }
                            }
                // LRU model of one XCD's L2 over one batch entry (P passes): every workgroup at step s + offset(wg), offsets uniform in
                // [0, lagSim]; capacity in 128-byte lines
                for (int lagSim : {0, 2, 4, 8})
                    for (int capLines : {16384, 24576, 32768}) {
                        std::vector<int> last(N, -1);          // last access time (in accesses)
                        std::vector<int> order;                // access sequence
                        std::mt19937 r2(7);
                        for (int p = 0; p < S.P; ++p) {
                            std::vector<int> off(S.wgPerXcd);
                            for (auto& o : off) o = lagSim ? (int)(r2() % (unsigned)(lagSim + 1)) : 0;
                            for (int s = 0; s < S.L4 + lagSim; ++s)
                                for (int jj = 0; jj < S.wgPerXcd; ++jj) {
                                    const int t = p * S.wgPerXcd + jj, ss = s - off[jj];
                                    if (t >= S.nTiles || ss < 0 || ss >= S.L4) continue;
                                    for (int w = 0; w < kWaves; ++w)
                                        for (int lgx = 0; lgx < 8; ++lgx) {
                                            const unsigned* wd = &S.words[((((size_t)(t * kWaves + w) * S.L4 + ss) * 8 + lgx) * 4)];
                                            for (int u = 0; u < 4; ++u)
                                                if ((int)(wd[u] >> 17) < S.D) order.push_back((int)(wd[u] & 0x1ffffu));
                                        }
                                }
                        }
                        // exact LRU via reuse distance is costly; approximate with "distinct lines since last access" using a
                        // time-stamped Fenwick tree
                        const int M = (int)order.size();
                        std::vector<int> fen(M + 1, 0);
                        auto add = [&](int i, int v) { for (++i; i <= M; i += i & -i) fen[i] += v; };
                        auto sum = [&](int i) { int r = 0; for (++i; i > 0; i -= i & -i) r += fen[i]; return r; };
                        int64_t hits = 0;
                        for (int i = 0; i < M; ++i) {
                            const int a = order[i];
                            if (last[a] >= 0) {
                                const int distinctSince = sum(i - 1) - sum(last[a]);
                                if (distinctSince < capLines) ++hits;
                                add(last[a], -1);
                            }
                            add(i, 1);
                            last[a] = i;
                        }
                        printf("   LRU model: lag %d steps, %5d lines (%.1f MB): hit rate %.3f\n", lagSim, capLines, capLines * 128 / 1048576.0, (double)hits / M);
                    }
                double err = 0.0;
                for (int i = 0; i < N; ++i) {
                    for (int src : G.nbr[i]) yr[i] += xs[src];
                    err = std::max(err, std::fabs(yr[i] - y[i]));
                }
                printf("hostcheck wgPerCU=%d distinct=%d: D=%d tiles=%d L4=%d fill=%.3f entries=%lld (nnz %lld) same-word clashes=%lld maxerr=%.2e max|src/N - slot/L|=%.3f\n", wgPerCU,
                       distinct, S.D, S.nTiles, S.L4, S.fill, (long long)cnt, (long long)G.nnz, (long long)clash, err, posErr);
            }
        return 0;
    }
    float *X, *Y;
    CK(hipMalloc(&X, (size_t)B * N * 32 * 4));
    CK(hipMalloc(&Y, (size_t)B * N * 32 * 4));
    std::vector<float> hx((size_t)N * 32);
    {
        std::mt19937 rng(1);
        for (auto& v : hx) v = (float)((rng() & 0xffff) / 65536.0 - 0.5);
        std::vector<float> hb(hx.size());
        for (int b = 0; b < B; ++b) {   // entry b = (1 + b / 256) * signal (exact in fp32 for these values up to rounding of the product)
            const float sc = 1.0f + (float)b / 256.0f;
            for (size_t i = 0; i < hx.size(); ++i) hb[i] = hx[i] * sc;
            CK(hipMemcpy(X + (size_t)b * N * 32, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        }
    }
    unsigned *prog, *stats;
    CK(hipMalloc(&prog, 8 * kSlots * 4 + 8 * 32 * 4));
    CK(hipMalloc(&stats, 32));
    int cfg = 0;
    for (int wgPerCU : wgs) {
        for (int distinct = 1; distinct >= 0; --distinct) {
            bool need = false;
            if (distinct) continue;
            Streams S = build_streams(G, wgPerCU, P, distinct != 0);
            printf("# wgPerCU=%d P=%d D=%d tiles=%d L4=%d (steps per sweep) fill=%.3f stream=%.1f MB lds=%d B distinct=%d\n", wgPerCU, S.P, S.D, S.nTiles, S.L4,
                   S.fill, S.words.size() * 4 / 1e6, (S.D + 8) * kRow * 4, distinct);
            unsigned* dst;
            CK(hipMalloc(&dst, S.words.size() * 4));
            CK(hipMemcpy(dst, S.words.data(), S.words.size() * 4, hipMemcpyHostToDevice));
            const int entriesPerXcd = (B + 7) / 8;
            const size_t lds = (size_t)(S.D + 8) * kRow * 4;
            const int rowsPerStep = (N + S.L4 - 1) / S.L4;
            for (int ac : accums) {
                if (distinct) continue;
                for (int lag : lags)
                    for (int sh : shifts)
                        for (int pf : pfs) {
                            const int myc = cfg++;
                            if (only >= 0 && myc != only) continue;
                            if (lag < 0 && sh != 0) continue;   // the cohort offset only exists through the lag protocol
                            const int shift = sh ? S.L4 / 2 : 0;
                            const unsigned span = (unsigned)(entriesPerXcd * S.P * S.L4 + shift + 64);
                            auto launch = [&](unsigned epoch) {
                                std::vector<unsigned short> init(8 * kSlots, 0xffffu);
                                for (int x = 0; x < 8; ++x)
                                    for (int jj = 0; jj < S.wgPerXcd; ++jj) init[x * kSlots + jj] = (jj & 1) ? (unsigned short)shift : 0;
                                CK(hipMemcpyAsync(prog, init.data(), init.size() * 2, hipMemcpyHostToDevice, 0));
                                CK(hipMemsetAsync(reinterpret_cast<char*>(prog) + 8 * (kSlots / 2) * 4, 0, 8 * 32 * 4, 0));
                                { const unsigned init8[8] = {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0, 0}; CK(hipMemcpyAsync(stats, init8, 32, hipMemcpyHostToDevice, 0)); }
                                CK(hipStreamSynchronize(0));
                                return epoch * span;
                            };
                            auto run = [&](unsigned base) {
#define LAUNCH(AC, MW, DP) hipLaunchKernelGGL((sweep_kernel<AC, MW, DP>), dim3(8 * S.wgPerXcd), dim3(kThreads), lds, 0, (const u32x4*)dst, X, Y, N, S.D, \
                                          S.L4, S.P, S.wgPerXcd, S.nTiles, entriesPerXcd, B, uval, prog, base, lag, stats, shift)
                                if (wgPerCU >= 3) {
                                    if (ac == 1) LAUNCH(1, 6, 2); else LAUNCH(0, 6, 2);
                                } else {
                                    if (depth == 4) {
                                        if (ac == 3) LAUNCH(3, 4, 4); else if (ac == 1) LAUNCH(1, 4, 4); else LAUNCH(0, 4, 4);
                                    } else {
                                        if (ac == 3) LAUNCH(3, 4, 2); else if (ac == 1) LAUNCH(1, 4, 2); else LAUNCH(0, 4, 2);
                                    }
                                }
                            };
                            static bool attr = false;
                            if (!attr) {
                                attr = true;
#define SETATTR(AC, MW, DP) CK(hipFuncSetAttribute((const void*)sweep_kernel<AC, MW, DP>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024))
                                SETATTR(3, 4, 4); SETATTR(1, 4, 4); SETATTR(0, 4, 4); SETATTR(3, 4, 2); SETATTR(1, 4, 2); SETATTR(0, 4, 2);
                                SETATTR(1, 6, 2); SETATTR(0, 6, 2);
                            }
                            CK(hipMemset(Y, 0xff, (size_t)B * N * 32 * 4));
                            unsigned epoch = 1;
                            run(launch(epoch++));
                            CK(hipDeviceSynchronize());
                            CK(hipGetLastError());
                            hipEvent_t e0, e1;
                            CK(hipEventCreate(&e0));
                            CK(hipEventCreate(&e1));
                            float msTot = 0.f;
                            unsigned hstats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                            for (int it = 0; it < iters; ++it) {
                                const unsigned base = launch(epoch++);
                                CK(hipEventRecord(e0));
                                run(base);
                                CK(hipEventRecord(e1));
                                CK(hipEventSynchronize(e1));
                                float ms;
                                CK(hipEventElapsedTime(&ms, e0, e1));
                                msTot += ms;
                            }
                            CK(hipMemcpy(hstats, stats, 32, hipMemcpyDeviceToHost));
                            const float ms = msTot / iters;
                            double err = -1.0;
                            if (ac) {
                                err = 0.0;
                                for (int b : {0, B - 1}) {
                                    std::vector<float> hy((size_t)N * 32);
                                    CK(hipMemcpy(hy.data(), Y + (size_t)b * N * 32, hy.size() * 4, hipMemcpyDeviceToHost));
                                    const double sc = 1.0 + b / 256.0;
                                    for (int row = 0; row < N; row += 7)
                                        for (int c = 0; c < 32; ++c) {
                                            double s = 0.0;
                                            for (int src : G.nbr[row]) s += (double)(hx[(size_t)src * 32 + c] * (float)sc);
                                            err = std::max(err, std::fabs(s * uval - hy[(size_t)row * 32 + c]));
                                        }
                                }
                            }
                            printf("cfg %d wg/CU=%d accum=%d lag=%d shift=%d pf=%d : %.3f ms/hop  %.1f %% of 8 TB/s  spins/wave=%.1f gaveup=%u startspread=%.1fus syncround=%.2fus  maxerr %.2e\n", myc,
                                   wgPerCU, ac, lag, shift, pf, ms, algBytes / ms / 1e6 / 80.0, (double)hstats[0] / (8.0 * S.wgPerXcd * (kWaves + 1)), hstats[1],
                                   (double)((((unsigned long long)hstats[7] << 32) | hstats[6]) - (((unsigned long long)hstats[5] << 32) | hstats[4])) / 100.0,
                                   hstats[2] ? (double)hstats[3] * 16.0 / 100.0 / hstats[2] : 0.0, err);
                            fflush(stdout);
                            CK(hipEventDestroy(e0));
                            CK(hipEventDestroy(e1));
                        }
            }
            CK(hipFree(dst));
        }
    }
    return 0;
}
