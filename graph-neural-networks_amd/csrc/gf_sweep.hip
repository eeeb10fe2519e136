// gf_sweep.hip -- spmm_sweep_kernel: the node-major hop (graphML.py:158-161, one `x = torch.matmul(x, S)`) as a SOURCE SWEEP with the
// partial sums of two batch entries held in the XCD's vector registers (image and rationale: gf_sweep_image.h).
//
// Register map of a wavefront (128 VGPRs, 4 waves per SIMD), fixed by hand because the accumulators are addressed RELATIVELY
// (s_set_gpr_idx_on: M0 holds the slot) and the ring is written by loads the compiler must not see as finished:
//     v0  .. v13    the compiler's (lane offsets, the LDS block registers) -- tools/check_sweep_isa.py checks on every build that no
//                   compiler-chosen operand, inside or outside the asm blocks, reaches v14
//     v16 .. v23    ring: the gathered dwords of the kSweepDepth steps in flight
//     v28 .. v127   accumulators: slot j = v[28 + j] (lanes 0-31: batch entry b, lanes 32-63: entry b + 8); slot 99 = trash
// One step = one buffer_load_dword (every lane reads tap(its entry) + source offset + its lane offset) and, kSweepDepth steps later,
// one v_add_f32 with the slot as relative register index.  All vector-memory loads of the loop are issued by inline asm, one per step:
// the s_waitcnt counts are exact by construction.
#include <stdlib.h>
#include <utility>

#include "gf_common.h"
#include "gf_sweep_image.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRing0 = 16, kAcc0 = 28;
constexpr int kSweepBarrierWords = 34 * 16;   // per XCD: 32 shard counters, the XCD counter, the release word -- a 64-byte line each
constexpr int kSweepRoundsPerBlock = kSweepBlock / kSweepDepth;
static_assert(kRing0 + kSweepDepth <= kAcc0 && kAcc0 + kSweepSlots <= 128, "register map");

// gather of ring slot S: every lane reads rsrc.base + voff(lane) + soff  (voff = lane offset inside the row, + the distance to the
// second batch entry's tap in lanes 32-63; soff = the source row: an SGPR -- no address arithmetic, no VGPR that changes per step)
template <int S>
__device__ __forceinline__ void sweep_issue(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dword v[%0], %1, %2, %3 offen" : : "i"(kRing0 + S), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// ring slot S has landed (kSweepDepth - 1 younger gathers may still be in flight): add it to the slot in the low byte of `entry`
template <int S>
__device__ __forceinline__ void sweep_consume(unsigned entry) {
    asm volatile("s_waitcnt vmcnt(%1) lgkmcnt(0)\n\t"
                 "s_set_gpr_idx_on %0, gpr_idx(SRC0,DST)\n\t"
                 "v_add_f32 v[%3], v[%3], v[%2]\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_nop 1"   // (a VALU instruction within two slots behind s_set_gpr_idx_off still runs indexed: measured)
                 :
                 : "s"(entry), "i"(kSweepDepth - 1), "i"(kRing0 + S), "i"(kAcc0)
                 : "memory");
}

template <int J>
__device__ __forceinline__ void sweep_zero_one() {
    asm volatile("v_mov_b32 v[%0], 0" : : "i"(kAcc0 + J) : "memory");
}
template <int... Js>
__device__ __forceinline__ void sweep_zero(std::integer_sequence<int, Js...>) {
    (sweep_zero_one<Js>(), ...);
}

// slot J: scale and store both entries' row (the row offset travels in an SGPR: soffset; a slot without a row is skipped by a scalar
// branch -- soffset is not range-checked).  The accumulators are zeroed by the caller once the stores have drained.
template <int J>
__device__ __forceinline__ void sweep_store_reg(__amdgpu_buffer_rsrc_t ro, unsigned voff, unsigned row_off, float uval) {
    if (row_off != kSweepNoRow)
        asm volatile("v_mul_f32 v[%0], %1, v[%0]\n\tbuffer_store_dword v[%0], %2, %3, %4 offen nt"
                     : : "i"(kAcc0 + J), "s"(uval), "v"(voff), "s"(ro), "s"(row_off) : "memory");
}
// slots J0 .. J0 + 7 with the row table in two registers (lane j of R0 / R1 holds the offset of slot j / 64 + j): the offsets are
// taken first (v_readlane writes SGPRs that the asm's instructions read: five wait states, see sweep_take), then the stores
template <int J0, int... Ks>
__device__ __forceinline__ void sweep_store_group(__amdgpu_buffer_rsrc_t ro, unsigned voff, unsigned R0, unsigned R1, float uval,
                                                  std::integer_sequence<int, Ks...>) {
    unsigned a[sizeof...(Ks)];
    ((a[Ks] = (J0 + Ks) < kSweepRows ? (unsigned)__builtin_amdgcn_readlane((int)((J0 + Ks) < 64 ? R0 : R1), (J0 + Ks) & 63) : kSweepNoRow), ...);
    static_assert(sizeof...(Ks) == 8, "groups of eight slots");
    asm volatile("s_nop 4" : "+s"(a[0]), "+s"(a[1]), "+s"(a[2]), "+s"(a[3]), "+s"(a[4]), "+s"(a[5]), "+s"(a[6]), "+s"(a[7]));
    (sweep_store_reg<(J0 + Ks) < kSweepSlots ? (J0 + Ks) : 0>(ro, voff, (J0 + Ks) < kSweepRows ? a[Ks] : kSweepNoRow, uval), ...);
}
template <int... Gs>
__device__ __forceinline__ void sweep_store(__amdgpu_buffer_rsrc_t ro, unsigned voff, unsigned R0, unsigned R1, float uval, std::integer_sequence<int, Gs...>) {
    (sweep_store_group<8 * Gs>(ro, voff, R0, R1, uval, std::make_integer_sequence<int, 8>{}), ...);
}

// entries of the round that starts at step RB * kSweepDepth of a block: lanes of the block register.  The look-ahead of a block's
// last round comes from the next block's register, which is read kSweepDepth lanes early (lane l holds step l - kSweepDepth of it).
template <int RB, int... Ss>
__device__ __forceinline__ void sweep_take(unsigned (&nxt)[kSweepDepth], unsigned blkA, unsigned blkB, std::integer_sequence<int, Ss...>) {
    constexpr int base = RB < kSweepRoundsPerBlock - 1 ? kSweepDepth * (RB + 1) : kSweepDepth;
    const unsigned blk = RB < kSweepRoundsPerBlock - 1 ? blkA : blkB;
    ((nxt[Ss] = (unsigned)__builtin_amdgcn_readlane((int)blk, base + Ss)), ...);
    // The SGPRs above are written by VALU instructions (v_readlane) and read by the VMEM / SALU instructions of the asm blocks that
    // follow, where the compiler's hazard recogniser cannot see them: all of them are forced to exist here, five wait states before
    // the first asm block of the round.
    static_assert(kSweepDepth == 8, "operand list below");
    asm volatile("s_nop 4" : "+s"(nxt[0]), "+s"(nxt[1]), "+s"(nxt[2]), "+s"(nxt[3]), "+s"(nxt[4]), "+s"(nxt[5]), "+s"(nxt[6]), "+s"(nxt[7]));
}
template <int... Ss>
__device__ __forceinline__ void sweep_prologue(__amdgpu_buffer_rsrc_t rs, unsigned voff, const unsigned (&e)[kSweepDepth], std::integer_sequence<int, Ss...>) {
    (sweep_issue<Ss>(rs, voff, e[Ss] >> 8), ...);
}
template <int... Ss>
__device__ __forceinline__ void sweep_round(__amdgpu_buffer_rsrc_t rs, unsigned voff, const unsigned (&cur)[kSweepDepth], const unsigned (&nxt)[kSweepDepth],
                                            std::integer_sequence<int, Ss...>) {
    ((sweep_consume<Ss>(cur[Ss]), sweep_issue<Ss>(rs, voff, nxt[Ss] >> 8)), ...);
}

__global__ __launch_bounds__(kThreads, 4) void spmm_sweep_kernel(const unsigned* __restrict__ ent, const unsigned* __restrict__ rows,
                                                                 const float* __restrict__ Xin, float* __restrict__ Xout, int N, int B,
                                                                 int passes, int steps, unsigned* __restrict__ gates, int use_barrier,
                                                                 float uval) {
    extern __shared__ unsigned s_streams[];   // [4 waves][passes][steps + kSweepBlock + 128]: entry lists and output-row tables, one copy per launch
    // every register above v13 is this kernel's by hand: make the allocation cover them
    asm volatile("; sweep register map: v16-v23 ring, v28-v127 accumulators" ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v127");
    const unsigned lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned wid = (blockIdx.x >> 3) * (kThreads / 64) + wv;   // wave of this XCD (dispatch order)
    if (wid >= (unsigned)kSweepWavesPerXcd) return;
    const unsigned tapBytes = (unsigned)N * 128u;
    const int stride = steps + kSweepDepth;                  // image
    const int lstride = steps + kSweepBlock + 128;            // LDS, per pass
    // XCD BARRIER (once per pass of an entry pair): what keeps the rows in use inside the 4 MiB L2 is that the XCD's waves walk the
    // sources together; the lists are equally long, so a common start is enough.  Two-level arrival (32 shard counters of 16 waves,
    // then one XCD counter), the last arriver publishes the epoch, everybody polls that word with a sleep in between; agent-scope
    // vector atomics of lane 0 -- placed where no gather is in flight (vmcnt = 0).  Bounded spin: a barrier that does not open in
    // time is passed anyway (results never depend on it; a launch whose waves are not all resident cannot hang).
    unsigned* bar = gates + (size_t)xcd * kSweepBarrierWords;
    unsigned epoch = 0;
    const int bar_every = use_barrier > 1 ? max(1, steps / kSweepBlock / use_barrier) : 0;
    auto xcd_barrier = [&]() {
        ++epoch;
        if (lane == 0) {
            unsigned* shard = bar + (wid / 16u) * 16u;
            unsigned* top = bar + 32 * 16;
            unsigned* rel = bar + 33 * 16;
            if (__hip_atomic_fetch_add(shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == 16u * epoch)
                if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == 32u * epoch)
                    __hip_atomic_store(rel, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spin = 0;
            while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spin < 20000) __builtin_amdgcn_s_sleep(8);
        }
    };

    // wave-private copies (LDS operations of one wave execute in order: no barrier)
    unsigned* my = s_streams + (size_t)wv * (size_t)passes * lstride;
    for (int p = 0; p < passes; ++p) {
        const unsigned* ep = ent + ((size_t)wid * passes + p) * stride;
        for (int i = (int)lane; i < steps + kSweepBlock; i += 64) my[p * lstride + i] = i < stride ? ep[i] : kSweepNothing;
        const unsigned* rp = rows + ((size_t)wid * passes + p) * kSweepSlots;
        for (int i = (int)lane; i < 128; i += 64) my[p * lstride + steps + kSweepBlock + i] = i < kSweepRows ? rp[i] : kSweepNoRow;
    }

    for (int b = xcd; b < B; b += 16) {   // the pair (b, b + 8); a lone last entry is paired with itself (both halves compute and store the same bytes)
        const unsigned second = b + 8 < B ? 8u * tapBytes : 0u;
        const unsigned voff = (lane & 31u) * 4u + (lane >= 32u ? second : 0u);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(Xin) + (size_t)b * tapBytes), 0,
                                                                            (int)(second + tapBytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(Xout) + (size_t)b * tapBytes), 0,
                                                                            (int)(second + tapBytes), 0x00020000);
        for (int p = 0; p < passes; ++p) {
            const unsigned* me = my + p * lstride;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous pass's stores have read their registers
            xcd_barrier();
            sweep_zero(std::make_integer_sequence<int, kSweepSlots>{});
            unsigned cur[kSweepDepth], nxt[kSweepDepth];
            unsigned blkA, blkB = me[(int)lane - kSweepDepth < 0 ? 0 : (int)lane - kSweepDepth];   // lane l holds step l - kSweepDepth
            sweep_take<kSweepRoundsPerBlock - 1>(nxt, blkB, blkB, std::make_integer_sequence<int, kSweepDepth>{});   // steps 0 .. kSweepDepth - 1
            sweep_prologue(rs, voff, nxt, std::make_integer_sequence<int, kSweepDepth>{});
            int blocks_to_barrier = bar_every;
            for (int t0 = 0; t0 < steps; t0 += kSweepBlock) {
                if (use_barrier > 1 && --blocks_to_barrier < 0) {   // (experiments: further barriers inside a pass; the atomics' vmcnt(0) drains the ring, whose registers stay valid)
                    xcd_barrier();
                    blocks_to_barrier = bar_every - 1;
                }
                blkA = me[t0 + (int)lane];
                blkB = me[t0 + kSweepBlock - kSweepDepth + (int)lane];
#define GF_SWEEP_ROUND(RB)                                                                          \
                {                                                                                   \
                    _Pragma("unroll") for (int s = 0; s < kSweepDepth; ++s) cur[s] = nxt[s];        \
                    sweep_take<RB>(nxt, blkA, blkB, std::make_integer_sequence<int, kSweepDepth>{}); \
                    sweep_round(rs, voff, cur, nxt, std::make_integer_sequence<int, kSweepDepth>{}); \
                }
                static_assert(kSweepRoundsPerBlock == 8, "rounds per block");
                GF_SWEEP_ROUND(0) GF_SWEEP_ROUND(1) GF_SWEEP_ROUND(2) GF_SWEEP_ROUND(3)
                GF_SWEEP_ROUND(4) GF_SWEEP_ROUND(5) GF_SWEEP_ROUND(6) GF_SWEEP_ROUND(7)
#undef GF_SWEEP_ROUND
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the kSweepDepth gap gathers behind the last step
            const unsigned R0 = me[steps + kSweepBlock + (int)lane], R1 = me[steps + kSweepBlock + 64 + (int)lane];
            static_assert(kSweepSlots <= 104, "13 groups of eight slots");
            sweep_store(ro, voff, R0, R1, uval, std::make_integer_sequence<int, 13>{});
        }
    }
}

size_t sweep_lds_bytes(const gf_csr_dev& m) {
    return (size_t)(kThreads / 64) * (size_t)m.sw_passes * (size_t)(m.sw_steps + kSweepBlock + 128) * sizeof(unsigned);
}

}  // namespace

bool gf_sweep_applicable(const gf_csr_dev& m, int N, int B, int W) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    // uniform GSO, 128-byte rows, the compact entry format, the entry lists of all passes in the LDS of four workgroups per CU, a buffer
    // descriptor that spans two batch entries 8 apart
    return W == 32 && m.sw_ent && m.sw_rows && m.sell_uniform && g_tune.panel_uniform && N <= kSweepMaxNodes && B >= 1 &&
           cus / 8 * 16 == kSweepWavesPerXcd && sweep_lds_bytes(m) <= 40 * 1024 && (int64_t)9 * N * 128 < ((int64_t)1 << 31);
}

int gf_sweep_launch(const gf_csr_dev& m, const float* Xin, float* Xout, int N, int B, hipStream_t st) {
    dim3 grid(8 * (kSweepWavesPerXcd / (kThreads / 64))), block(kThreads);
    // barriers per pass: 1 = at its start; > 1: experiments (more of them cost more than their hit rate buys: 1.61 / 1.74 / 2.05 ms for
    // 1 / 2 / 4).  There is no "none": without the common start the hit rate falls to 0.5 AND -- not understood -- about 10 rows in a
    // million come out wrong (whole accumulators, both batch entries of a pair), which no run with the barrier has shown.
    const int use_barrier = g_tune.spmm_lag > 1 ? g_tune.spmm_lag : 1;
    GF_HIP(hipMemsetAsync(m.sw_gate, 0, 8 * kSweepBarrierWords * sizeof(unsigned), st));
    hipLaunchKernelGGL(spmm_sweep_kernel, grid, block, sweep_lds_bytes(m), st, m.sw_ent, m.sw_rows, Xin, Xout, N, B, m.sw_passes, m.sw_steps, m.sw_gate,
                       use_barrier, m.sell_uval);
    GF_LAUNCH_CHECK("spmm_sweep_kernel");
    return GF_OK;
}
