// gf_msweep.hip -- spmm_msweep_kernel: the node-major hop (graphML.py:158-161, one `x = torch.matmul(x, S)`) as a SOURCE SWEEP whose
// scatter-accumulate is an fp32 multi-block MFMA (image, geometry and rationale: gf_msweep_image.h).
//
// One workgroup of four waves per CU (one 512-register wave per SIMD), 256 workgroups; workgroup L runs on XCD L % 8 (observed
// dispatch order: a wrong guess costs speed, never correctness).  XCD x works through batch entries x, x + 8, ...: its 128 waves hold
// the entry's whole output (S sets x 32 rows x 32 features per wave) in accumulator registers, walk the entry's source rows together
// (T rounds of S steps; a step = one 8-row gather + four v_mfma_f32_4x4x1_16b_f32), store, and meet at an XCD barrier before the
// next entry.
//
// The body of a (batch entry, pass) is ONE inline-asm block written with assembler macros (MS_* below): hipcc's scheduler and register
// allocator cannot express this kernel -- given the same program as C++ with builtins it hoisted the gathers into vmcnt(0) groups,
// kept most accumulators in VGPRs and moved every one of them through a[0:3] around its MFMA (two v_accvgpr_read/write quads per
// MFMA), and spilled at 25 sets.  Register map of a wave (the asm block lists v24-v255 and a0-a255 as clobbers, so whatever the
// compiler keeps across the block lives in v0-v23):
//     a0  .. a255          accumulators 0..255:   accumulator (set s, quad q, slot i) = number 16 s + 4 q + i
//     v112 .. v255         accumulators 256..399  (MFMAs take their C/D operand from either file)
//     v24 .. v43           ring: the 16 bytes per lane of the kMsDepth = 5 gathers in flight (store phase: four output quads + addresses)
//     v44 .. v48           their A operands (edge weight in the lanes of the destination slot, zero elsewhere)
//     v49, v50, v108, v109 temporaries (gather offset / slot mask, alternating between steps)
//     v52 .. v79           entries of this lane's position, one per step of a round (reloaded four at a time, a round ahead)
//     v80 .. v107          their values (weighted GSOs)
// Vector-memory operations of the loop are issued in a fixed order, loads return in order: the s_waitcnt counts are computed by the
// assembler from that order (MS_RLCOUNT).  Wait states the hardware does not interlock (VALU write -> MFMA read: 2; MFMA write ->
// VALU / VMEM read: up to 19) are covered by distance: an A operand is written five steps before its MFMAs, accumulators are read
// only after the loop (s_nop block in MS_BODY).
#include <stdlib.h>
#include <atomic>

#include "gf_common.h"
#include "gf_msweep_image.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMsGateWords = 64;      // per XCD: one arrival counter on a 256-byte line of its own
constexpr int kMsGateSlots = 16;      // launches whose barrier counters may be live at once (slots rotate)

// Assembler macros (a basic asm statement: no operand substitution, `%` and `|` are the assembler's).  Defined once per module.
#define GF_MS_MACROS R"(
.ifndef MS_MACROS_DEFINED
.set MS_MACROS_DEFINED, 1
.set MS_R0, 24
.set MS_A0, 44
.set MS_E0, 52
.set MS_V0, 80
.set MS_ACCV, 112
.macro MS_MFMA s, q, k
  .if ((\s)*16 + (\q)*4) < 256
    v_mfma_f32_4x4x1_16b_f32 a[(\s)*16+(\q)*4:(\s)*16+(\q)*4+3], v[MS_A0+(\k)], v[MS_R0+4*(\k)+(\q)], a[(\s)*16+(\q)*4:(\s)*16+(\q)*4+3]
  .else
    v_mfma_f32_4x4x1_16b_f32 v[MS_ACCV+(\s)*16+(\q)*4-256:MS_ACCV+(\s)*16+(\q)*4-256+3], v[MS_A0+(\k)], v[MS_R0+4*(\k)+(\q)], v[MS_ACCV+(\s)*16+(\q)*4-256:MS_ACCV+(\s)*16+(\q)*4-256+3]
  .endif
.endm
.macro MS_ZERO S
  .set MS_I, 0
  .rept (\S)*16
    .if MS_I < 256
      v_accvgpr_write_b32 a[MS_I], 0
    .else
      v_mov_b32 v[MS_ACCV+MS_I-256], 0
    .endif
    .set MS_I, MS_I+1
  .endr
.endm
.macro MS_RLCOUNT j, S
  .set MS_CNT, 0
  .irp d, 0,1,2,3,4
    .set MS_SP, ((\j)+\d) % (\S)
    .if ((MS_SP & 3) == 3) || (MS_SP == (\S)-1)
      .set MS_CNT, MS_CNT+1
    .endif
  .endr
.endm
.macro MS_ISSUE sp, k, par, S, UNI, rs, re, rv, vfg, vslot, vevoff, smask, snext
  .if \par
    .set MS_TA, 108
  .else
    .set MS_TA, 49
  .endif
  v_and_or_b32 v[MS_TA], v[MS_E0+(\sp)], \smask, \vfg
  buffer_load_dwordx4 v[MS_R0+4*(\k):MS_R0+4*(\k)+3], v[MS_TA], \rs, 0 offen
  v_bfe_i32 v[MS_TA+1], v[MS_E0+(\sp)], \vslot, 1
  .if \UNI
    v_and_b32 v[MS_A0+(\k)], 1.0, v[MS_TA+1]
  .else
    v_and_b32 v[MS_A0+(\k)], v[MS_V0+(\sp)], v[MS_TA+1]
  .endif
  .if (((\sp) & 3) == 3) || ((\sp) == (\S)-1)
    buffer_load_dwordx4 v[MS_E0+((\sp)/4)*4:MS_E0+((\sp)/4)*4+3], \vevoff, \re, \snext offen offset:((\sp)/4)*16
    .if (\UNI) == 0
      buffer_load_dwordx4 v[MS_V0+((\sp)/4)*4:MS_V0+((\sp)/4)*4+3], \vevoff, \rv, \snext offen offset:((\sp)/4)*16
    .endif
  .endif
.endm
.macro MS_BODY S, UNI, RB, rs, ro, re, rv, vfg, vslot, vevoff, vrow, smask, suval, scur, snxt, sit
  MS_ZERO \S
  .set MS_Q, 0
  .rept ((\S)+3)/4
    buffer_load_dwordx4 v[MS_E0+4*MS_Q:MS_E0+4*MS_Q+3], \vevoff, \re, 0 offen offset:MS_Q*16
    .if (\UNI) == 0
      buffer_load_dwordx4 v[MS_V0+4*MS_Q:MS_V0+4*MS_Q+3], \vevoff, \rv, 0 offen offset:MS_Q*16
    .endif
    .set MS_Q, MS_Q+1
  .endr
  s_waitcnt vmcnt(0)
  .set MS_J, 0
  .rept 5
    MS_ISSUE MS_J, MS_J, (MS_J & 1), \S, \UNI, \rs, \re, \rv, \vfg, \vslot, \vevoff, \smask, \scur
    .set MS_J, MS_J+1
  .endr
MS_LOOP_\@:
  .set MS_J, 0
  .rept \S
    MS_RLCOUNT MS_J, \S
    s_waitcnt vmcnt(4 + MS_CNT*(2-(\UNI)))
    MS_MFMA MS_J, 0, (MS_J % 5)
    MS_MFMA MS_J, 1, (MS_J % 5)
    MS_MFMA MS_J, 2, (MS_J % 5)
    MS_MFMA MS_J, 3, (MS_J % 5)
    .if MS_J + 5 < \S
      MS_ISSUE (MS_J+5), (MS_J % 5), (MS_J & 1), \S, \UNI, \rs, \re, \rv, \vfg, \vslot, \vevoff, \smask, \scur
    .else
      MS_ISSUE (MS_J+5-(\S)), (MS_J % 5), (MS_J & 1), \S, \UNI, \rs, \re, \rv, \vfg, \vslot, \vevoff, \smask, \snxt
    .endif
    .set MS_J, MS_J+1
  .endr
  s_add_u32 \scur, \scur, \RB
  s_add_u32 \snxt, \snxt, \RB
  s_sub_u32 \sit, \sit, 1
  s_cmp_lg_u32 \sit, 0
  s_cbranch_scc1 MS_LOOP_\@
  s_waitcnt vmcnt(0)
  s_nop 7
  s_nop 7
  s_nop 7
  ds_read_b128 v[MS_E0:MS_E0+3], \vrow
  .set MS_S, 0
  .rept \S
    .if MS_S + 1 < \S
      ds_read_b128 v[MS_E0+4*((MS_S+1)&1):MS_E0+4*((MS_S+1)&1)+3], \vrow offset:(MS_S+1)*128
      s_waitcnt lgkmcnt(1)
    .else
      s_waitcnt lgkmcnt(0)
    .endif
    .set MS_I, 0
    .rept 4
      .set MS_Q, 0
      .rept 4
        .set MS_RR, MS_S*16 + MS_Q*4 + MS_I
        .if MS_RR < 256
          v_accvgpr_read_b32 v[MS_R0+4*MS_I+MS_Q], a[MS_RR]
          .if \UNI
            v_mul_f32 v[MS_R0+4*MS_I+MS_Q], \suval, v[MS_R0+4*MS_I+MS_Q]
          .endif
        .else
          .if \UNI
            v_mul_f32 v[MS_R0+4*MS_I+MS_Q], \suval, v[MS_ACCV+MS_RR-256]
          .else
            v_mov_b32 v[MS_R0+4*MS_I+MS_Q], v[MS_ACCV+MS_RR-256]
          .endif
        .endif
        .set MS_Q, MS_Q+1
      .endr
      v_add_u32 v[MS_R0+16+MS_I], v[MS_E0+4*(MS_S&1)+MS_I], \vfg
      buffer_store_dwordx4 v[MS_R0+4*MS_I:MS_R0+4*MS_I+3], v[MS_R0+16+MS_I], \ro, 0 offen nt
      .set MS_I, MS_I+1
    .endr
    .set MS_S, MS_S+1
  .endr
.endm
.endif
)"

#define GF_MS_V8(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define GF_MS_A8(a) "a" #a "0", "a" #a "1", "a" #a "2", "a" #a "3", "a" #a "4", "a" #a "5", "a" #a "6", "a" #a "7", "a" #a "8", "a" #a "9"
#define GF_MS_CLOBBERS                                                                                                             \
    "memory", "scc", "v24", "v25", "v26", "v27", "v28", "v29", GF_MS_V8(3), GF_MS_V8(4), GF_MS_V8(5), GF_MS_V8(6), GF_MS_V8(7),    \
        GF_MS_V8(8), GF_MS_V8(9), GF_MS_V8(10), GF_MS_V8(11), GF_MS_V8(12), GF_MS_V8(13), GF_MS_V8(14), GF_MS_V8(15), GF_MS_V8(16), \
        GF_MS_V8(17), GF_MS_V8(18), GF_MS_V8(19), GF_MS_V8(20), GF_MS_V8(21), GF_MS_V8(22), GF_MS_V8(23), GF_MS_V8(24), "v250",     \
        "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", GF_MS_A8(1),           \
        GF_MS_A8(2), GF_MS_A8(3), GF_MS_A8(4), GF_MS_A8(5), GF_MS_A8(6), GF_MS_A8(7), GF_MS_A8(8), GF_MS_A8(9), GF_MS_A8(10),      \
        GF_MS_A8(11), GF_MS_A8(12), GF_MS_A8(13), GF_MS_A8(14), GF_MS_A8(15), GF_MS_A8(16), GF_MS_A8(17), GF_MS_A8(18),            \
        GF_MS_A8(19), GF_MS_A8(20), GF_MS_A8(21), GF_MS_A8(22), GF_MS_A8(23), GF_MS_A8(24), "a250", "a251", "a252", "a253",        \
        "a254", "a255"

template <int S, int UNI>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1)))
void spmm_msweep_kernel(const uint32_t* __restrict__ ent, const float* __restrict__ val, const uint32_t* __restrict__ rows,
                        const float* __restrict__ Xin, float* __restrict__ Xout, int N, int B, int passes, int rounds,
                        unsigned* __restrict__ gate, int use_barrier, float uval, unsigned src_mask) {
    constexpr int S4 = (S + 3) / 4 * 4;
    constexpr unsigned kRoundBytes = 8u * S4 * 4u;
    static_assert(S % kMsDepth == 0 && S >= 2 * kMsDepth && S <= kMsMaxSets, "ring slots are static; an entry quad is reloaded a round ahead");
    __shared__ unsigned s_rows[kThreads / 64][S * 32];      // per wave: output byte offsets of (set, position, slot)
    asm volatile(GF_MS_MACROS);
    const unsigned lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wid = (unsigned)__builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 3) * (kThreads / 64) + wv));   // wave of this XCD
    const unsigned pos = lane >> 3, fg16 = (lane & 7u) * 16u, slotbit = lane & 3u;
    const unsigned tapBytes = (unsigned)N * 128u;
    const size_t streamWords = (size_t)(rounds + 2) * 8 * S4;
    const unsigned evoff = pos * (S4 * 4u);                 // this lane's position inside a round of the entry stream
    const unsigned rowlds = (unsigned)(size_t)(&s_rows[wv][0]) + pos * 16u;   // (an LDS address is the low half of the generic pointer)
    const unsigned smask = 0xffffff80u & src_mask;
    unsigned* ctr = gate + (size_t)xcd * kMsGateWords;
    unsigned epoch = 0;
    int table_of = -1;

    for (int b = xcd; b < B; b += 8) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(Xin) + (size_t)b * tapBytes), 0, (int)tapBytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<char*>(Xout) + (size_t)b * tapBytes), 0, (int)tapBytes, 0x00020000);
        for (int pass = 0; pass < passes; ++pass) {
            const size_t pw = (size_t)pass * kMsWavesPerXcd + wid;
            if (table_of != pass) {   // wave-private copy (LDS operations of one wave execute in order: no barrier)
                for (int i = (int)lane; i < S * 32; i += 64) s_rows[wv][i] = rows[pw * (size_t)(S * 32) + i];
                table_of = pass;
            }
            const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(ent + pw * streamWords), 0, (int)(streamWords * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)((UNI ? reinterpret_cast<const float*>(ent) : val) + pw * streamWords), 0, (int)(streamWords * 4), 0x00020000);
            unsigned scur = kRoundBytes, snxt = 2u * kRoundBytes, sit = (unsigned)rounds;
            asm volatile("MS_BODY %13, %14, %15, %3, %4, %5, %6, %7, %8, %9, %10, %11, %12, %0, %1, %2"
                         : "+s"(scur), "+s"(snxt), "+s"(sit)
                         : "s"(rs), "s"(ro), "s"(re), "s"(rv), "v"(fg16), "v"(slotbit), "v"(evoff), "v"(rowlds), "s"(smask), "s"(uval),
                           "n"(S), "n"(UNI), "n"(kRoundBytes)
                         : GF_MS_CLOBBERS);

            if (use_barrier) {
                // XCD barrier: one scalar atomic per workgroup on the XCD's counter (monotonic over the launch), then the first wave polls
                // with returning scalar atomics (they execute in the L2: coherent, and they wait on lgkmcnt, not on the stores' vmcnt).
                // Bounded: a barrier that does not open in time is passed anyway -- results never depend on it.
                ++epoch;
                __builtin_amdgcn_s_barrier();
                if (wv == 0) {
                    unsigned t = 1u;
                    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(ctr) : "memory");
                    const unsigned target = 32u * epoch;
                    for (int spin = 0; t + 1u < target && spin < 4000; ++spin) {
                        __builtin_amdgcn_s_sleep(16);
                        t = 0u;
                        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(ctr) : "memory");
                        t -= 1u;   // (compared as "arrivals before mine", like the first read)
                    }
                }
                __builtin_amdgcn_s_barrier();
            }
        }
    }
}

int cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    return cus;
}

}  // namespace

bool gf_msweep_applicable(const gf_csr_dev& m, int N, int B, int W) {
    // 128-byte rows, an image, one workgroup per CU on a 256-CU device (8 XCDs x 32 CUs x 4 SIMDs = the 128 waves per XCD of the
    // image), 32-bit byte offsets inside a tap, enough batch entries to give every XCD one
    return W == 32 && m.ms_ent && m.ms_rows && m.ms_sets >= 2 * kMsDepth && (m.ms_uniform || m.ms_val) && cu_count() == 256 && B >= 8 &&
           (int64_t)N * 128 < (int64_t)kMsPad;
}

int gf_msweep_launch(const gf_csr_dev& m, const float* Xin, float* Xout, int N, int B, hipStream_t st) {
    static std::atomic<unsigned> next_slot{0};
    unsigned* gate = m.ms_gate + (size_t)(next_slot.fetch_add(1) % kMsGateSlots) * 8 * kMsGateWords;
    const int use_barrier = g_tune.spmm_bar;
    if (use_barrier) GF_HIP(hipMemsetAsync(gate, 0, 8 * kMsGateWords * sizeof(unsigned), st));
    dim3 grid(256), block(kThreads);
    const unsigned src_mask = g_tune.spmm_srcmask ? (unsigned)g_tune.spmm_srcmask : 0xffffffffu;   // experiments (timing only): confine the gathers to a window
#define GF_MS(SV, UV)                                                                                                              \
    hipLaunchKernelGGL((spmm_msweep_kernel<SV, UV>), grid, block, 0, st, m.ms_ent, m.ms_val, m.ms_rows, Xin, Xout, N, B, m.ms_passes, \
                       m.ms_rounds, gate, use_barrier, m.sell_uval, src_mask)
#define GF_MS_S(UV)                                   \
    switch (m.ms_sets) {                              \
        case 10: GF_MS(10, UV); break;                \
        case 15: GF_MS(15, UV); break;                \
        case 20: GF_MS(20, UV); break;                \
        default: GF_MS(25, UV); break;                \
    }
    if (m.ms_uniform) {
        GF_MS_S(1);
    } else {
        GF_MS_S(0);
    }
#undef GF_MS_S
#undef GF_MS
    GF_LAUNCH_CHECK("spmm_msweep_kernel");
    return GF_OK;
}

size_t gf_msweep_gate_bytes() { return (size_t)kMsGateSlots * 8 * kMsGateWords * sizeof(unsigned); }
