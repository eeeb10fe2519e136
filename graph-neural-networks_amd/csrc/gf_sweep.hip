// gf_sweep.hip -- spmm_sweep_kernel: the node-major hop (graphML.py:158-161, one `x = torch.matmul(x, S)`) as a SOURCE SWEEP with the
// partial sums of a whole batch entry held in the XCD's vector registers (image and rationale: gf_sweep_image.h).
//
// Register map of a wavefront (128 VGPRs, 4 waves per SIMD), fixed by hand because the accumulators are addressed RELATIVELY
// (s_set_gpr_idx_on: M0 holds the slot) and the ring is written by loads the compiler must not see as finished:
//     v0  .. v15    the compiler's (lane offsets, address temporaries)            -- checked by tools/check_sweep_isa.py on every build
//     v16 .. v23    ring: the gathered dwords of the kSweepDepth steps in flight
//     v28 .. v127   accumulators: slot j of the half-wave = v[28 + j] of its 32 lanes
// One step = one buffer_load_dword (lanes 0-31: source row A_t, lanes 32-63: source row B_t; offsets past the tap return 0.0f) and,
// kSweepDepth steps later, two v_add_f32 under complementary exec masks with the slot as relative register index.  All vector-memory
// loads of the loop are issued by inline asm, one per step, so the s_waitcnt counts are exact by construction.
#include <stdlib.h>
#include <utility>

#include "gf_common.h"
#include "gf_sweep_image.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRing0 = 16, kAcc0 = 28;
constexpr int kSweepBarrierWords = 34 * 16;   // per XCD: 32 shard counters, the XCD counter, the release word -- a 64-byte line each
static_assert(kRing0 + kSweepDepth <= kAcc0 && kAcc0 + kSweepSlots <= 128, "register map");

struct SweepCtx {
    unsigned lane4;      // (lane & 31) * 4: byte offset of the lane inside its 128-byte row
    unsigned maskhi;     // 0 in lanes 0-31, ~0 in lanes 32-63
    unsigned masklo;     // ~0 in lanes 0-31, 0 in lanes 32-63
    unsigned long long lomask;   // exec mask of lanes 0-31
};

// gather of ring slot S: lanes 0-31 read tap + offA + lane4, lanes 32-63 tap + offA + delta + lane4
template <int S>
__device__ __forceinline__ void sweep_issue(const SweepCtx& c, __amdgpu_buffer_rsrc_t rs, unsigned offA, unsigned delta) {
    // The address is built IN the ring register the load returns to: a vector-memory instruction may read its address VGPR well after it
    // was issued (measured: with a shared temporary, overwritten by the next step three instructions later, one gather in ~10^5 used the
    // next step's half-built address when the TA queue was full), and nothing touches v[RING + S] again before the load has landed.
    asm volatile("v_and_b32 v[%0], %1, %2\n\tv_add3_u32 v[%0], v[%0], %3, %4\n\tbuffer_load_dword v[%0], v[%0], %5, 0 offen"
                 :
                 : "i"(kRing0 + S), "s"(delta), "v"(c.maskhi), "s"(offA), "v"(c.lane4), "s"(rs)
                 : "memory");
}

// ring slot S has landed (D - 1 younger gathers may still be in flight): add it to slot jA of lanes 0-31 and slot jB of lanes 32-63
template <int S>
__device__ __forceinline__ void sweep_consume(const SweepCtx& c, unsigned jA, unsigned jB) {
    // No exec switching: the gathered dword is split into "lanes 0-31, zeros above" and "zeros below, lanes 32-63" (v_and with a lane
    // mask: +0.0f where masked) and each part is added with ALL lanes active -- the other half's slot of the same number gets + 0.0f.
    // (A version that narrowed exec to one half per v_add delivered, once in ~10^5 gathers, a gather whose 32 lanes had all read lane
    // 0's address: the first VALU instructions after the exec restore are the address computation of the next gather.)
    unsigned xa, xb;
    asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)\n\t"
                 "v_and_b32 %1, v[%6], %5\n\t"               // lanes 32-63 of the gathered dword, zeros below
                 "v_and_b32 %0, v[%6], %8\n\t"               // lanes 0-31, zeros above
                 "s_set_gpr_idx_on %2, gpr_idx(SRC0,DST)\n\t"
                 "v_add_f32 v[%7], v[%7], %0\n\t"
                 "s_set_gpr_idx_idx %3\n\t"
                 "v_add_f32 v[%7], v[%7], %1\n\t"
                 "s_set_gpr_idx_off\n\t"
                 "s_nop 1"                                   // (a VALU instruction within two slots behind s_set_gpr_idx_off still runs indexed)
                 : "=&v"(xa), "=&v"(xb)
                 : "s"(jA), "s"(jB), "i"(kSweepDepth - 1), "v"(c.maskhi), "i"(kRing0 + S), "i"(kAcc0), "v"(c.masklo)
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

// slot J: scale and store lanes 0-31 to output offset roA, lanes 32-63 to roB.  The row offset travels in an SGPR (soffset) and the
// vector offset is the constant lane4 -- no address VGPR changes between stores (see sweep_issue); a slot without a row is skipped by a
// scalar branch (soffset is not range-checked).  The accumulators are zeroed by the caller once the stores have drained.
template <int J>
__device__ __forceinline__ void sweep_store_reg(const SweepCtx& c, __amdgpu_buffer_rsrc_t ro, unsigned roA, unsigned roB, float uval) {
    asm volatile("v_mul_f32 v[%0], %1, v[%0]" : : "i"(kAcc0 + J), "s"(uval) : "memory");
    if (roA != kSweepNoRow)
        asm volatile("s_mov_b64 exec, %4\n\tbuffer_store_dword v[%0], %1, %2, %3 offen nt\n\ts_mov_b64 exec, -1"
                     : : "i"(kAcc0 + J), "v"(c.lane4), "s"(ro), "s"(roA), "s"(c.lomask) : "memory");
    if (roB != kSweepNoRow)
        asm volatile("s_not_b64 exec, %4\n\tbuffer_store_dword v[%0], %1, %2, %3 offen nt\n\ts_mov_b64 exec, -1"
                     : : "i"(kAcc0 + J), "v"(c.lane4), "s"(ro), "s"(roB), "s"(c.lomask) : "memory");
}
template <int J>
__device__ __forceinline__ void sweep_store_one(const SweepCtx& c, __amdgpu_buffer_rsrc_t ro, const uint2* __restrict__ rows, float uval) {
    const uint2 r = rows[J];
    sweep_store_reg<J>(c, ro, r.x, r.y, uval);
}
template <int... Js>
__device__ __forceinline__ void sweep_store(const SweepCtx& c, __amdgpu_buffer_rsrc_t ro, const uint2* __restrict__ rows, float uval,
                                            std::integer_sequence<int, Js...>) {
    (sweep_store_one<Js>(c, ro, rows, uval), ...);
}

template <int... Ss>
__device__ __forceinline__ void sweep_prologue(const SweepCtx& c, __amdgpu_buffer_rsrc_t rs, const uint2 (&e)[kSweepDepth],
                                               std::integer_sequence<int, Ss...>) {
    (sweep_issue<Ss>(c, rs, e[Ss].x & 0x00ffffffu, (e[Ss].y & 0x00ffffffu) - (e[Ss].x & 0x00ffffffu)), ...);
}
template <int... Ss>
__device__ __forceinline__ void sweep_round(const SweepCtx& c, __amdgpu_buffer_rsrc_t rs, const uint2 (&cur)[kSweepDepth],
                                            const uint2 (&nxt)[kSweepDepth], std::integer_sequence<int, Ss...>) {
    ((sweep_consume<Ss>(c, cur[Ss].x >> 24, cur[Ss].y >> 24),
      sweep_issue<Ss>(c, rs, nxt[Ss].x & 0x00ffffffu, (nxt[Ss].y & 0x00ffffffu) - (nxt[Ss].x & 0x00ffffffu))),
     ...);
}

// slots J0 .. J0 + 3 with the row table in two registers (lane j of R0 / R1 holds the offsets of slot j / 64 + j): the offsets are taken
// first (v_readlane writes SGPRs that the asm's instructions read: see sweep_take), then the stores
template <int J0, int... Ks>
__device__ __forceinline__ void sweep_store_lds_group(const SweepCtx& c, __amdgpu_buffer_rsrc_t ro, const uint2& R0, const uint2& R1, float uval,
                                                      std::integer_sequence<int, Ks...>) {
    unsigned a[sizeof...(Ks)], d[sizeof...(Ks)];
    ((a[Ks] = (unsigned)__builtin_amdgcn_readlane((int)((J0 + Ks) < 64 ? R0.x : R1.x), (J0 + Ks) & 63),
      d[Ks] = (unsigned)__builtin_amdgcn_readlane((int)((J0 + Ks) < 64 ? R0.y : R1.y), (J0 + Ks) & 63)),
     ...);
    static_assert(sizeof...(Ks) == 4, "groups of four slots");
    asm volatile("s_nop 4" : "+s"(a[0]), "+s"(d[0]), "+s"(a[1]), "+s"(d[1]), "+s"(a[2]), "+s"(d[2]), "+s"(a[3]), "+s"(d[3]));
    (sweep_store_reg<J0 + Ks>(c, ro, a[Ks], d[Ks], uval), ...);
}
template <int... Gs>
__device__ __forceinline__ void sweep_store_lds(const SweepCtx& c, __amdgpu_buffer_rsrc_t ro, const uint2& R0, const uint2& R1, float uval,
                                                std::integer_sequence<int, Gs...>) {
    (sweep_store_lds_group<4 * Gs>(c, ro, R0, R1, uval, std::make_integer_sequence<int, 4>{}), ...);
}

// entries of the round that starts at step RB * kSweepDepth of a kSweepBlock-step block: lanes of the block register.  The look-ahead of a
// block's last round comes from the next block's register, which is read kSweepDepth lanes early (lane l holds step l - kSweepDepth
// of that block): lane 0 is never a source.
constexpr int kSweepRoundsPerBlock = kSweepBlock / kSweepDepth;
template <int RB, int... Ss>
__device__ __forceinline__ void sweep_take(uint2 (&nxt)[kSweepDepth], const uint2& blkA, const uint2& blkB, std::integer_sequence<int, Ss...>) {
    constexpr int base = RB < kSweepRoundsPerBlock - 1 ? kSweepDepth * (RB + 1) : kSweepDepth;
    const uint2& blk = RB < kSweepRoundsPerBlock - 1 ? blkA : blkB;
    ((nxt[Ss].x = (unsigned)__builtin_amdgcn_readlane((int)blk.x, base + Ss), nxt[Ss].y = (unsigned)__builtin_amdgcn_readlane((int)blk.y, base + Ss)), ...);
    // The SGPRs above are written by VALU instructions (v_readlane) and read by the VALU / VMEM instructions of the asm blocks that
    // follow, where the compiler's hazard recogniser cannot see them: all of them are forced to exist here, five wait states before the
    // first asm block of the round.
    static_assert(kSweepDepth == 8, "operand list below");
    asm volatile("s_nop 4" : "+s"(nxt[0].x), "+s"(nxt[0].y), "+s"(nxt[1].x), "+s"(nxt[1].y), "+s"(nxt[2].x), "+s"(nxt[2].y), "+s"(nxt[3].x), "+s"(nxt[3].y),
                 "+s"(nxt[4].x), "+s"(nxt[4].y), "+s"(nxt[5].x), "+s"(nxt[5].y), "+s"(nxt[6].x), "+s"(nxt[6].y), "+s"(nxt[7].x), "+s"(nxt[7].y));
}

// LDSS = 1: the wave's entry list and output-row table live in LDS for the whole launch (one pass only: they are the same for every
// batch entry, and re-reading 4 MB of them per batch entry through the scalar cache pushes the source window out of the 4 MiB L2).
template <int LDSS>
__global__ __launch_bounds__(kThreads, 4) void spmm_sweep_kernel(const uint2* __restrict__ ent, const uint2* __restrict__ rows,
                                                                 const float* __restrict__ Xin, float* __restrict__ Xout, int N, int B,
                                                                 int passes, int steps, unsigned* __restrict__ gates, int gate_lag,
                                                                 float uval) {
    extern __shared__ uint2 s_streams[];   // LDSS: [4 waves][steps + 64 + 128]
    // every register above v15 is this kernel's by hand: make the allocation cover them
    asm volatile("; sweep register map: v16-v23 ring, v28-v127 accumulators" ::: "v16", "v17", "v18", "v19", "v20", "v21",
                 "v22", "v23", "v24", "v25", "v26", "v27", "v127");
    const unsigned lane = threadIdx.x & 63;
    SweepCtx c;
    c.lane4 = (lane & 31u) * 4u;
    c.maskhi = lane >= 32u ? 0xffffffffu : 0u;
    c.masklo = ~c.maskhi;
    c.lomask = 0x00000000ffffffffull;
    const int xcd = blockIdx.x & 7;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned wid = (blockIdx.x >> 3) * (kThreads / 64) + wv;   // wave of this XCD
    if (wid >= (unsigned)kSweepWavesPerXcd) return;
    const unsigned tapBytes = (unsigned)N * 128u;
    const size_t stride = (size_t)(steps + kSweepDepth);
    // XCD BARRIERS.  The lists are sorted by source and equally long, so "step t" means "sources around t * N / steps" for every wave
    // of the XCD; what keeps the rows in use inside the 4 MiB L2 is that the waves stay together.  `nbar` times per batch entry (at
    // its start and at equal distances inside it) all 512 waves of the XCD meet: two-level arrival (32 shard counters of 16 waves,
    // then one XCD counter: a single word serves only ~30-90 returning atomics per microsecond), the last arriver publishes the
    // epoch, everybody else polls that word -- a different line than the counters -- with a sleep in between.  Scalar atomics and
    // loads, each waited for on the spot: no scalar-memory operation is in flight while the gather loop runs.  Bounded spin: a
    // barrier that does not open in time is passed anyway and switches the barriers off for this wave (results never depend on them,
    // and a launch whose waves are not all resident cannot hang).
    unsigned* bar = gates + (size_t)xcd * kSweepBarrierWords;
    unsigned epoch = 0;
    const bool bars_on = gate_lag > 0;
    auto xcd_barrier = [&]() {   // only where no gather is in flight (vmcnt = 0): the atomics below are ordinary vector-memory operations
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

    uint2* my = s_streams + (size_t)wv * (size_t)(steps + kSweepBlock + 128);
    if constexpr (LDSS != 0) {   // wave-private copies: LDS operations of one wave execute in order, no barrier
        const uint2* ep = ent + (size_t)wid * stride;
        for (int i = (int)lane; i < steps + kSweepBlock; i += 64) my[i] = i < (int)stride ? ep[i] : make_uint2(kSweepNothing, kSweepNothing);
        const uint2* rp = rows + (size_t)wid * kSweepSlots;
        for (int i = (int)lane; i < 128; i += 64) my[steps + kSweepBlock + i] = i < kSweepSlots ? rp[i] : make_uint2(kSweepNoRow, kSweepNoRow);
    }

    for (int b = xcd; b < B; b += 8) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(Xin) + (size_t)b * tapBytes), 0, (int)tapBytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro =
            __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(Xout) + (size_t)b * tapBytes), 0, (int)tapBytes, 0x00020000);
        if constexpr (LDSS != 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous batch entry's stores have read their registers
            if (bars_on) xcd_barrier();
            sweep_zero(std::make_integer_sequence<int, kSweepSlots>{});
            uint2 cur[kSweepDepth], nxt[kSweepDepth];
            uint2 blkA, blkB = my[(int)lane - kSweepDepth < 0 ? 0 : (int)lane - kSweepDepth];   // lane l holds step l - kSweepDepth
            sweep_take<kSweepRoundsPerBlock - 1>(nxt, blkB, blkB, std::make_integer_sequence<int, kSweepDepth>{});   // steps 0 .. kSweepDepth - 1
            sweep_prologue(c, rs, nxt, std::make_integer_sequence<int, kSweepDepth>{});
            for (int t0 = 0; t0 < steps; t0 += kSweepBlock) {
                blkA = my[t0 + (int)lane];
                blkB = my[t0 + kSweepBlock - kSweepDepth + (int)lane];
#define GF_SWEEP_ROUND(R8)                                                                        \
                {                                                                                   \
                    _Pragma("unroll") for (int s = 0; s < kSweepDepth; ++s) cur[s] = nxt[s];        \
                    sweep_take<R8>(nxt, blkA, blkB, std::make_integer_sequence<int, kSweepDepth>{}); \
                    sweep_round(c, rs, cur, nxt, std::make_integer_sequence<int, kSweepDepth>{});    \
                }
                static_assert(kSweepRoundsPerBlock == 8, "rounds per block");
                GF_SWEEP_ROUND(0) GF_SWEEP_ROUND(1) GF_SWEEP_ROUND(2) GF_SWEEP_ROUND(3)
                GF_SWEEP_ROUND(4) GF_SWEEP_ROUND(5) GF_SWEEP_ROUND(6) GF_SWEEP_ROUND(7)
#undef GF_SWEEP_ROUND
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the kSweepDepth padding gathers behind the last step
            const uint2 R0 = my[steps + kSweepBlock + (int)lane], R1 = my[steps + kSweepBlock + 64 + (int)lane];
            static_assert(kSweepSlots % 4 == 0, "groups of four slots");
            sweep_store_lds(c, ro, R0, R1, uval, std::make_integer_sequence<int, kSweepSlots / 4>{});
        } else {
            for (int p = 0; p < passes; ++p) {
                const uint2* ep = ent + ((size_t)p * kSweepWavesPerXcd + wid) * stride;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous pass's stores have read their registers
                if (bars_on) xcd_barrier();
                sweep_zero(std::make_integer_sequence<int, kSweepSlots>{});
                uint2 cur[kSweepDepth], nxt[kSweepDepth];
#pragma unroll
                for (int s = 0; s < kSweepDepth; ++s) nxt[s] = ep[s];
                sweep_prologue(c, rs, nxt, std::make_integer_sequence<int, kSweepDepth>{});
                for (int t = 0; t < steps; t += kSweepDepth) {
#pragma unroll
                    for (int s = 0; s < kSweepDepth; ++s) {
                        cur[s] = nxt[s];
                        nxt[s] = ep[t + kSweepDepth + s];
                    }
                    sweep_round(c, rs, cur, nxt, std::make_integer_sequence<int, kSweepDepth>{});
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the kSweepDepth padding gathers behind the last step
                sweep_store(c, ro, rows + ((size_t)p * kSweepWavesPerXcd + wid) * kSweepSlots, uval, std::make_integer_sequence<int, kSweepSlots>{});
            }
        }
    }
}

}  // namespace

bool gf_sweep_applicable(const gf_csr_dev& m, int N, int B, int W) {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    // one pass whose streams fit the LDS of four workgroups per CU (the variant that streams them from global memory, spmm_sd = 1, is an
    // experiment: it delivered sporadic wrong gathers that the LDS variant never showed)
    const size_t lds = (size_t)(kThreads / 64) * (size_t)(m.sw_steps + kSweepBlock + 128) * sizeof(uint2);
    return W == 32 && m.sw_ent && m.sw_rows && m.sell_uniform && g_tune.panel_uniform && N <= kSweepMaxNodes && B >= 1 &&
           cus / 8 * 16 == kSweepWavesPerXcd && ((m.sw_passes == 1 && lds <= 40 * 1024) || g_tune.spmm_sd == 1);
}

int gf_sweep_launch(const gf_csr_dev& m, const float* Xin, float* Xout, int N, int B, hipStream_t st) {
    dim3 grid(8 * (kSweepWavesPerXcd / (kThreads / 64))), block(kThreads);
    const int gates_per_pass = g_tune.spmm_lag;   // experiments: 0 = no gates
    if (gates_per_pass > 0) GF_HIP(hipMemsetAsync(m.sw_gate, 0, 8 * kSweepBarrierWords * sizeof(unsigned), st));
    const size_t lds = (size_t)(kThreads / 64) * (size_t)(m.sw_steps + kSweepBlock + 128) * sizeof(uint2);
    if (getenv("GFHIP_SWEEP_DEBUG")) {
        int n1 = -1, n0 = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, spmm_sweep_kernel<1>, kThreads, lds);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n0, spmm_sweep_kernel<0>, kThreads, 0);
        fprintf(stderr, "sweep: lds = %zu B, occupancy query: %d workgroups per CU with LDS streams, %d without; grid %u\n", lds, n1, n0, grid.x);
    }
    if (m.sw_passes == 1 && lds <= 40 * 1024 && g_tune.spmm_sd != 1)   // (4 workgroups per CU: 160 KB of LDS; spmm_sd = 1: experiments, streams from global memory)
        hipLaunchKernelGGL(spmm_sweep_kernel<1>, grid, block, lds, st, m.sw_ent, m.sw_rows, Xin, Xout, N, B, m.sw_passes, m.sw_steps, m.sw_gate,
                           gates_per_pass, m.sell_uval);
    else
        hipLaunchKernelGGL(spmm_sweep_kernel<0>, grid, block, 0, st, m.sw_ent, m.sw_rows, Xin, Xout, N, B, m.sw_passes, m.sw_steps, m.sw_gate,
                           gates_per_pass, m.sell_uval);
    GF_LAUNCH_CHECK("spmm_sweep_kernel");
    return GF_OK;
}
