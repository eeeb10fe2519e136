// gf_msweep.hip -- spmm_msweep_kernel: the node-major hop (graphML.py:158-161, one `x = torch.matmul(x, S)`) as a SOURCE SWEEP whose
// scatter-accumulate is an fp32 multi-block MFMA (image, geometry and rationale: gf_msweep_image.h).
//
// One workgroup of four waves per CU (one 512-register wave per SIMD), 256 workgroups.  A fused chain takes its TEAMS from the hardware,
// not from the dispatch order: every workgroup reads the XCC it runs on (s_getreg HW_REG_XCC_ID), draws its rank among that XCC's
// workgroups from a census (agent-scope atomics, once per launch) and, when the census shows 32 workgroups on each of 8 XCCs, works as
// wave rank * 4 + w of team XCC -- the hand-over between the hops of an entry then runs through ONE physical L2 by construction.  Any
// other census (or a grid that does not become resident within the time limit) ends the launch without a store, and the repair
// kernel queued behind it (spmm_msweep_repair_kernel, gated on the launch's flag) computes the chain row by row: slower, same bits,
// no trap, no host synchronisation.  A single hop (no hand-over) keeps workgroup L on team L % 8: there a wrong guess costs speed only.
// XCD x works through the (batch entry, 32-column slab) pairs x, x + 8, ... (rows of 64 / 96 / 128 columns are 2 / 3 / 4 slabs of the same image,
// addressed {row, column} through a strided buffer resource): its 128 waves hold
// the entry's whole output (S sets x 32 rows x 32 features per wave) in accumulator registers, walk the entry's source rows together
// (T rounds of S steps; a step = one 8-row gather + four v_mfma_f32_4x4x1_16b_f32) and store; the K - 1 hops of an entry run back to
// back in one launch with an XCD barrier between them (hop h + 1 gathers what hop h stored -- with plain stores, so that the rows
// stored last, the lowest row bands, are still in the XCD's L2 when the next hop's sweep starts with them).  Rows too long for a group (hub
// rows, gf_msweep_image.h) are summed by the waves between store phase and hand-over, in compiler code (template HUB).
//
// The body of a (batch entry, pass) is ONE inline-asm block written with assembler macros (MS_* below): hipcc's scheduler and register
// allocator cannot express this kernel -- given the same program as C++ with builtins it hoisted the gathers into vmcnt(0) groups,
// kept most accumulators in VGPRs and moved every one of them through a[0:3] around its MFMA (two v_accvgpr_read/write quads per
// MFMA), and spilled at 25 sets.  Register map of a wave (the asm block lists v24-v255 and a0-a255 as clobbers, so whatever the
// compiler keeps across the block lives in v0-v23):
//     a0  .. a255          accumulators 0..255:   accumulator (set s, quad q, slot i) = number 16 s + 4 q + i
//     v112 .. v255         accumulators 256..399  (MFMAs take their C/D operand from either file)
//     v24 .. v63           ring: the 16 bytes per lane of the D = 10 gathers in flight (store phase: four output quads + addresses)
//     v64 .. v73           their A operands (edge weight in the lanes of the destination slot, zero elsewhere)
//     v74 .. v81           temporaries {entry, value, gather offset, slot mask} x step parity
//     v82 .. v89           entry buffers of the even / odd rounds: ONE 16-byte load per lane and round (lane 8 p + i: quad i = steps 4 i .. 4 i + 3
//                          of position p); a step's entry reaches the position's 8 lanes through two DPP moves (quad broadcast, then quad copy)
//     v90 .. v97           value buffers (weighted GSOs)
// Vector-memory operations of the loop are issued in a fixed order, loads return in order: the s_waitcnt counts are computed by the
// assembler from that order (MS_RLCOUNT).  Wait states the hardware does not interlock (VALU write -> MFMA read: 2; MFMA write ->
// VALU / VMEM read: up to 19) are covered by distance: an A operand is written D steps before its MFMAs, the two DPP moves of an entry
// have two MFMAs between them (VALU write -> DPP read: 2), accumulators are read only after the loop (s_nop block in MS_BODY).
// The store phase zeroes each accumulator behind its read: the body's registers carry state between asm statements, and
// tools/check_msweep_isa.py (a CPU test) fails the build when a compiler-emitted instruction names one of them or anything spills.
#include <stdlib.h>
#include <atomic>
#include <mutex>

#include "gf_common.h"
#include "gf_msweep_image.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMsGateWords = 64;      // per XCD: the arrival counter (word 0) of its barrier, 256 bytes of its own
constexpr int kMsGateSlots = 16;      // launches whose barriers may be live at once: one slot per stream (zeroed once, at plan creation)
// behind the 8 team lines of a slot: the census block of its launches (agent-scope atomics only)
constexpr int kMsCensusWord = 8 * kMsGateWords;
constexpr int kMsSlotWords = kMsCensusWord + 64;
enum : int { kCsCount = 0 /* [8] workgroups per XCC */, kCsArrive = 8, kCsGen = 9 /* += 2 per launch; bit 0 = this launch is abandoned */,
             kCsRepair = 10 /* read by the repair kernel behind the launch */, kCsPoison = 11 /* sticky: a launch on this slot timed out */ };
enum : unsigned { kMsStatusCensus = 1u /* a census did not show 8 x 32 */, kMsStatusTimeout = 2u /* a grid did not become resident / a barrier did not open */ };

__device__ __forceinline__ unsigned ag_load(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ag_add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Assembler macros (a basic asm statement: no operand substitution, `%` and `|` are the assembler's).  Defined once per module.
#ifndef GF_MS_NT   // experiments (make variant): gathers with the non-temporal hint
#define GF_MS_NT 0
#endif
#ifndef GF_MS_STPLAIN   // output rows with plain stores (1, default) or non-temporal ones (0; make variant): plain stores leave the rows in the XCD's
#define GF_MS_STPLAIN 1 // L2, and the rows stored LAST are the lowest bands -- the ones the next hop of the fused chain gathers first (1.24 -> 1.13 ms at config 4)
#endif
#ifndef GF_MS_EXP      // experiment (make msvariant; TIMING ONLY, results wrong): 8 = two of the four MFMAs of a step (round 6's other timing builds -- no MFMAs,
#define GF_MS_EXP 0    // plain moves for the DPP broadcasts, global_load for buffer_load -- are in git history at e5e8983; profiles/r06_a_decompose)
#endif
#ifndef GF_MS_PFP      // scalar prefetch: GF_MS_PFP source rows (two s_loads each) per GF_MS_PFQ steps of a wave.  3 per 4: config 4's sweep needs 0.74 rows per
#define GF_MS_PFP 3    // step and wave (N / (rounds x 128 waves x 25 steps)) and a wave can keep 15 scalar loads outstanding: ~1.8 per step at the ~850 clocks a
#define GF_MS_PFQ 4    // first touch takes -- 2 per step (round 5) ran into that limit and held the wave's whole instruction stream back
#endif
#define GF_MS_STR2(x) #x
#define GF_MS_STR(x) GF_MS_STR2(x)
#define GF_MS_MACROS R"(
.ifndef MS_MACROS_DEFINED
.set MS_MACROS_DEFINED, 1
.set MS_ACCV, 112
.set MS_GATHER_NT, GF_MS_NT_VALUE
.set MS_STORE_PLAIN, GF_MS_STPLAIN_VALUE
.set MS_EXP, GF_MS_EXP_VALUE
.set MS_PFP, GF_MS_PFP_VALUE
.set MS_PFQ, GF_MS_PFQ_VALUE
.set MS_R0, 24
.macro MS_SETMAP D
  .set MS_A0, 24 + 4*(\D)
  .set MS_T0, (24 + 5*(\D) + 1) & 0xfffe  // 8 temporaries: {entry, value, offset, mask} x step parity (register tuples start at even numbers)
  .set MS_E0, MS_T0 + 8              // entry buffers: rounds of even / odd parity, 4 registers each (lane 8 p + i holds quad i of position p)
  .set MS_V0, MS_E0 + 8              // value buffers (weighted GSOs)
.endm
.macro MS_MFMA s, q, k
  .if (MS_EXP & 8) && ((\q) & 1)
  .elseif ((\s)*16 + (\q)*4) < 256
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
// MS_CNT = entry-buffer reloads issued behind the gather of step n up to the gather of step n + D - 1 (one reload per round, behind the
// gather of the round's last step S - 1)
.macro MS_RLCOUNT n, S, D
  .set MS_CNT, 0
  .set MS_DD, 0
  .rept \D
    .if (((\n)+MS_DD) % (\S)) == (\S)-1
      .set MS_CNT, MS_CNT+1
    .endif
    .set MS_DD, MS_DD+1
  .endr
.endm
// Step sp of a round whose entries sit in buffer `buf` (round parity): lane 8 p + i of the buffer's register sp % 4 holds the entry of
// position p for step 4 i + sp % 4.  MS_BCAST1 broadcasts lane (sp / 4) % 4 inside every quad, MS_BCAST2 copies the quad that holds it
// over the position's other quad (rows of 16 lanes = two positions; banks = quads): two DPP moves, with >= 2 instructions between them.
.macro MS_BCAST1 dst, src, sp
  v_mov_b32_dpp v[\dst], v[\src] quad_perm:[((\sp)/4)%4,((\sp)/4)%4,((\sp)/4)%4,((\sp)/4)%4] row_mask:0xf bank_mask:0xf
.endm
.macro MS_BCAST2 dst, sp
  .if ((\sp)/16) == 0
    v_mov_b32_dpp v[\dst], v[\dst] row_shr:4 row_mask:0xf bank_mask:0xa
  .else
    v_mov_b32_dpp v[\dst], v[\dst] row_shl:4 row_mask:0xf bank_mask:0x5
  .endif
.endm
// gather of step sp into ring slot k and its A operand (the entry / value were broadcast into the temporaries of parity `par`), and --
// behind the last step of a round -- the reload of that round's buffer with the round two later
.macro MS_ISSUE sp, k, par, buf, rho, S, UNI, rs, re, rv, vfg, vslot, vevoff, smask, scur, xptr
  .set MS_TA, MS_T0 + 4*(\par)
  .if MS_IDX                                   // wide rows (W = 64 / 96 / 128): address = row index x stride (buffer resource) + constant column offset in v[MS_TA+3]
    v_lshrrev_b32 v[MS_TA+2], 7, v[MS_TA]
    buffer_load_dwordx4 v[MS_R0+4*(\k):MS_R0+4*(\k)+3], v[MS_TA+2:MS_TA+3], \rs, 0 idxen offen
    v_bfe_i32 v[MS_A0+(\k)], v[MS_TA], \vslot, 1
    .if \UNI
      v_and_b32 v[MS_A0+(\k)], 1.0, v[MS_A0+(\k)]
    .else
      v_and_b32 v[MS_A0+(\k)], v[MS_TA+1], v[MS_A0+(\k)]
    .endif
  .else
  v_and_or_b32 v[MS_TA+2], v[MS_TA], \smask, \vfg
  .if MS_GATHER_NT
    buffer_load_dwordx4 v[MS_R0+4*(\k):MS_R0+4*(\k)+3], v[MS_TA+2], \rs, 0 offen nt
  .else
    buffer_load_dwordx4 v[MS_R0+4*(\k):MS_R0+4*(\k)+3], v[MS_TA+2], \rs, 0 offen
  .endif
  v_bfe_i32 v[MS_TA+3], v[MS_TA], \vslot, 1
  .if \UNI
    v_and_b32 v[MS_A0+(\k)], 1.0, v[MS_TA+3]
  .else
    v_and_b32 v[MS_A0+(\k)], v[MS_TA+1], v[MS_TA+3]
  .endif
  .endif
  .if (\sp) == (\S)-1
    buffer_load_dwordx4 v[MS_E0+4*(\buf):MS_E0+4*(\buf)+3], \vevoff, \re, \scur offen offset:((\rho)+2)*1024
    .if (\UNI) == 0
      buffer_load_dwordx4 v[MS_V0+4*(\buf):MS_V0+4*(\buf)+3], \vevoff, \rv, \scur offen offset:((\rho)+2)*1024
    .endif
  .endif
.endm
// store phase: slot i of position p of set s = the row whose byte offset is word 4p + i of the set's line in the LDS row table
// (staging: v[MS_R0 .. MS_R0+19] = four output quads + their addresses, row-table lines in v[MS_E0 .. MS_E0+7])
.macro MS_STORE S, UNI, ro, vfg, vrow, suval
  .if MS_IDX                                   // (wide rows: store address = {row index, column offset} pairs in the dead A-operand registers)
    v_mov_b32 v[MS_A0+1], \vfg
    v_mov_b32 v[MS_A0+3], \vfg
    v_mov_b32 v[MS_A0+5], \vfg
    v_mov_b32 v[MS_A0+7], \vfg
  .endif
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
        .if MS_RR < 256                          // (read, then zero: the next (entry, hop) finds its accumulators cleared)
          v_accvgpr_read_b32 v[MS_R0+4*MS_I+MS_Q], a[MS_RR]
          v_accvgpr_write_b32 a[MS_RR], 0
          .if \UNI
            v_mul_f32 v[MS_R0+4*MS_I+MS_Q], \suval, v[MS_R0+4*MS_I+MS_Q]
          .endif
        .else
          .if \UNI
            v_mul_f32 v[MS_R0+4*MS_I+MS_Q], \suval, v[MS_ACCV+MS_RR-256]
          .else
            v_mov_b32 v[MS_R0+4*MS_I+MS_Q], v[MS_ACCV+MS_RR-256]
          .endif
          v_mov_b32 v[MS_ACCV+MS_RR-256], 0
        .endif
        .set MS_Q, MS_Q+1
      .endr
      .if MS_IDX
        v_lshrrev_b32 v[MS_A0+2*MS_I], 7, v[MS_E0+4*(MS_S&1)+MS_I]
        buffer_store_dwordx4 v[MS_R0+4*MS_I:MS_R0+4*MS_I+3], v[MS_A0+2*MS_I:MS_A0+2*MS_I+1], \ro, 0 idxen offen
      .else
      v_add_u32 v[MS_R0+16+MS_I], v[MS_E0+4*(MS_S&1)+MS_I], \vfg
      .if MS_STORE_PLAIN
        buffer_store_dwordx4 v[MS_R0+4*MS_I:MS_R0+4*MS_I+3], v[MS_R0+16+MS_I], \ro, 0 offen
      .else
        buffer_store_dwordx4 v[MS_R0+4*MS_I:MS_R0+4*MS_I+3], v[MS_R0+16+MS_I], \ro, 0 offen nt
      .endif
      .endif
      .set MS_I, MS_I+1
    .endr
    .set MS_S, MS_S+1
  .endr
.endm
// S sets, ring depth D, two rounds per loop iteration (the entry buffers alternate by round parity; 2 S steps must be a multiple of D: ring
// slots are static); scur = byte offset of the iteration's first round in the entry stream, sit = iterations left
.macro MS_BODY S, UNI, IDX, PF, D, rs, ro, re, rv, vfg, vslot, vevoff, vrow, smask, suval, xptr, smaxpf, schunk, scur, sit, spfr, spfc, sdummy, stl0, stl1, uid
  MS_SETMAP \D
  .set MS_IDX, \IDX
  s_nop 4                                      // (an SGPR operand the compiler has just written with a VALU instruction -- v_readfirstlane / v_readlane -- needs five wait states before
                                               //  a VMEM instruction reads it, and the hazard recogniser does not look into this block)
  buffer_load_dwordx4 v[MS_E0:MS_E0+3], \vevoff, \re, 0 offen
  buffer_load_dwordx4 v[MS_E0+4:MS_E0+7], \vevoff, \re, 0 offen offset:1024
  .if (\UNI) == 0
    buffer_load_dwordx4 v[MS_V0:MS_V0+3], \vevoff, \rv, 0 offen
    buffer_load_dwordx4 v[MS_V0+4:MS_V0+7], \vevoff, \rv, 0 offen offset:1024
  .endif
  .if MS_IDX
    v_mov_b32 v[MS_T0+3], \vfg
    v_mov_b32 v[MS_T0+7], \vfg
  .endif
  s_waitcnt vmcnt(0)
  s_memtime \stl0
  .set MS_N, 0
  .rept \D
    MS_BCAST1 (MS_T0+4*(MS_N&1)), (MS_E0+(MS_N%4)), MS_N
    .if (\UNI) == 0
      MS_BCAST1 (MS_T0+4*(MS_N&1)+1), (MS_V0+(MS_N%4)), MS_N
    .endif
    s_nop 1
    MS_BCAST2 (MS_T0+4*(MS_N&1)), MS_N
    .if (\UNI) == 0
      MS_BCAST2 (MS_T0+4*(MS_N&1)+1), MS_N
    .endif
    MS_ISSUE MS_N, MS_N, (MS_N & 1), 0, 0, \S, \UNI, \rs, \re, \rv, \vfg, \vslot, \vevoff, \smask, \scur, \xptr
    .set MS_N, MS_N+1
  .endr
MS_LOOP_\uid:
  .set MS_N, 0
  .rept 2*(\S)
    .set MS_SP, (MS_N+(\D)) % (\S)              // the step whose gather this step issues, its round (0 .. 2 past the iteration's first) and buffer
    .set MS_RHO, (MS_N+(\D)) / (\S)
    MS_RLCOUNT MS_N, \S, \D
    s_waitcnt vmcnt((\D) - 1 + MS_CNT*(2-(\UNI)))
    MS_BCAST1 (MS_T0+4*(MS_N&1)), (MS_E0+4*(MS_RHO&1)+(MS_SP%4)), MS_SP
    .if (\UNI) == 0
      MS_BCAST1 (MS_T0+4*(MS_N&1)+1), (MS_V0+4*(MS_RHO&1)+(MS_SP%4)), MS_SP
    .endif
    MS_MFMA (MS_N % (\S)), 0, (MS_N % (\D))
    MS_MFMA (MS_N % (\S)), 1, (MS_N % (\D))
    MS_BCAST2 (MS_T0+4*(MS_N&1)), MS_SP
    .if (\UNI) == 0
      MS_BCAST2 (MS_T0+4*(MS_N&1)+1), MS_SP
    .endif
    MS_MFMA (MS_N % (\S)), 2, (MS_N % (\D))
    MS_MFMA (MS_N % (\S)), 3, (MS_N % (\D))
    .if (\PF) && ((((MS_N+1)*MS_PFP)/MS_PFQ) > ((MS_N*MS_PFP)/MS_PFQ))
      s_load_dword \sdummy, \xptr, \spfc offset:((MS_N*MS_PFP)/MS_PFQ)*0x4000
      s_load_dword \sdummy, \xptr, \spfc offset:((MS_N*MS_PFP)/MS_PFQ)*0x4000+0x40
    .endif
    MS_ISSUE MS_SP, (MS_N % (\D)), (MS_N & 1), (MS_RHO & 1), MS_RHO, \S, \UNI, \rs, \re, \rv, \vfg, \vslot, \vevoff, \smask, \scur, \xptr
    .set MS_N, MS_N+1
  .endr
  s_add_u32 \scur, \scur, 2048
  .if \PF
    s_add_u32 \spfr, \spfr, \schunk
    s_min_u32 \spfc, \spfr, \smaxpf
  .endif
  s_sub_u32 \sit, \sit, 1
  s_cmp_lg_u32 \sit, 0
  s_cbranch_scc1 MS_LOOP_\uid
  s_memtime \stl1
  // the D gathers in flight belong to round T (gaps); they and the last entry reloads must land before their registers are reused
  s_waitcnt vmcnt(0) lgkmcnt(0)
  s_nop 7
  s_nop 7
  s_nop 7
  MS_STORE \S, \UNI, \ro, \vfg, \vrow, \suval
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

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PH: phases of compiler code around the asm bodies -- bit 0: hub rows (after the store phase), bit 1: the boundary layout pass x[B, 32, Nin] ->
// tap 0 as a pre-phase of every batch entry (before its first hop).  Instantiations with PH != 0 zero their accumulators before every body.
template <int S, int UNI, int PF, int D, int IDX, int PH>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1)))
void spmm_msweep_kernel(const uint32_t* __restrict__ ent, const float* __restrict__ val, const uint32_t* __restrict__ rows,
                        const float* __restrict__ Xin, float* __restrict__ Xtaps, size_t tapStrideBytes, int nhops, int N, int B, int passes,
                        int rounds, unsigned* __restrict__ gate, int use_barrier, float uval, unsigned src_mask, int pf_lead,
                        int nostore, unsigned long long* __restrict__ trace, int census, unsigned tmo_ticks, int W,
                        const uint32_t* __restrict__ hubpack, const float* __restrict__ xref, const float* __restrict__ xmask, int Nin) {
    constexpr int HUB = PH & 1, XP = (PH >> 1) & 1;
    static_assert(XP == 0 || IDX == 0, "layout pre-phase: 32-column rows");
    constexpr int U = 2;                                    // rounds per loop iteration (the two entry buffers alternate by round parity)
    static_assert(IDX == 0 || PF == 0, "wide rows: no scalar prefetch");
    static_assert((U * S) % D == 0 && D < S && S <= kMsMaxSets && S <= 32,
                  "ring slots are static: 2 S steps are a multiple of the depth; a round's buffer is reloaded (for the round two later) behind "
                  "its last use and has been waited for (in-order returns) before the ring reaches that round; D < S keeps that reload "
                  "inside the two rounds of an iteration (12-bit instruction offsets)");
    __shared__ unsigned s_rows[kThreads / 64][S * 32];      // per wave: output byte offsets of (set, position, slot)
    __shared__ unsigned s_ctl[4];                           // census result {XCC, rank, abandoned}, [3] = a barrier timed out
    __shared__ float s_tile[XP ? kThreads / 64 : 1][XP ? 32 * 65 : 1];   // layout pre-phase: per wave a 32 x 64 (+1) transpose tile
    asm volatile(".set GF_MS_NT_VALUE, " GF_MS_STR(GF_MS_NT) "\n\t.set GF_MS_STPLAIN_VALUE, " GF_MS_STR(GF_MS_STPLAIN)
                 "\n\t.set GF_MS_EXP_VALUE, " GF_MS_STR(GF_MS_EXP) "\n\t.set GF_MS_PFP_VALUE, " GF_MS_STR(GF_MS_PFP) "\n\t.set GF_MS_PFQ_VALUE, " GF_MS_STR(GF_MS_PFQ));
    asm volatile(GF_MS_MACROS);
    const unsigned lane = threadIdx.x & 63;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned* cs = gate + kMsCensusWord;
    int xcd = blockIdx.x & 7;
    unsigned rank = blockIdx.x >> 3;
    if (census) {
        // Census (launches with a hand-over between workgroups: cooperative, so the grid is resident or the launch waits).  Tickets and the
        // arrival count are agent-scope atomics: coherent whatever XCD a workgroup runs on.  The last of the 256 arrivers reads the eight
        // counts, takes every word back to zero (a captured launch replays as is) and publishes the verdict in the generation word; the
        // others poll the generation they read BEFORE arriving.  census = 2 / 3 / 4 (experiments): the last arriver calls the census bad /
        // workgroup 5 claims the next XCC / workgroup 7 never arrives (the others run into the time limit).
        if (threadIdx.x == 0) {
            unsigned xcc, bad = 0u, rk = 0u;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7u;
            if (census == 3 && blockIdx.x == 5) xcc = (xcc + 1u) & 7u;
            if (ag_load(cs + kCsPoison) || (census == 4 && blockIdx.x == 7)) {
                ag_store(cs + kCsRepair, 1u);
                bad = 1u;
            } else {
                const unsigned g0 = ag_load(cs + kCsGen);
                rk = ag_add(cs + kCsCount + xcc, 1u);
                if (ag_add(cs + kCsArrive, 1u) == 255u) {
                    bad = census == 2 ? 1u : 0u;
                    for (int i = 0; i < 8; ++i) {
                        bad |= ag_load(cs + kCsCount + i) != 32u ? 1u : 0u;
                        ag_store(cs + kCsCount + i, 0u);
                    }
                    ag_store(cs + kCsArrive, 0u);
                    bad |= ag_load(cs + kCsPoison) ? 1u : 0u;
                    ag_store(cs + kCsRepair, bad);
                    ag_store(cs + kCsGen, ((g0 & ~1u) + 2u) | bad);
                } else {
                    const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
                    unsigned g = g0;
                    while ((g >> 1) == (g0 >> 1)) {
                        __builtin_amdgcn_s_sleep(8);
                        g = ag_load(cs + kCsGen);
                        if ((g >> 1) == (g0 >> 1) && __builtin_amdgcn_s_memrealtime() - tb > (unsigned long long)tmo_ticks) {
                            ag_store(cs + kCsPoison, 1u);   // the slot's words are no longer trustworthy: every later launch on it goes to the repair kernel
                            ag_store(cs + kCsRepair, 1u);
                            g = g0 | 1u;
                            break;
                        }
                    }
                    bad = g & 1u;
                }
            }
            s_ctl[0] = xcc;
            s_ctl[1] = rk;
            s_ctl[2] = bad;
            s_ctl[3] = 0u;
        }
        __syncthreads();
        if (s_ctl[2]) return;                               // abandoned before the first store: the repair kernel behind this launch does the work
        xcd = __builtin_amdgcn_readfirstlane((int)s_ctl[0]);
        rank = (unsigned)__builtin_amdgcn_readfirstlane((int)s_ctl[1]);
    }
    const unsigned wid = (unsigned)__builtin_amdgcn_readfirstlane((int)(rank * (kThreads / 64) + wv));   // wave of this XCD's team
    // accumulators start at zero; every store phase leaves them zero again (the compiler keeps out of v24+ / a0+ between the bodies:
    // tools/check_msweep_isa.py, a CPU test)
    asm volatile("MS_ZERO %0" ::"n"(S) : GF_MS_CLOBBERS);
    const unsigned pos = lane >> 3, fg16 = (lane & 7u) * 16u, slotbit = lane & 3u;
    const unsigned tapBytes = (unsigned)N * 128u;
    const size_t streamWords = (size_t)(rounds + 2) * 256;
    const unsigned evoff = lane * 16u;                      // this lane's 16 bytes of a round of the entry stream (lane 8 p + i: quad i of position p)
    const unsigned rowlds = (unsigned)(size_t)(&s_rows[wv][0]) + pos * 16u;   // (an LDS address is the low half of the generic pointer)
    const unsigned smask = 0xffffff80u & src_mask;
    const unsigned chunkBytes = (unsigned)((N + rounds - 1) / rounds) * 128u;   // source rows per round
    unsigned* ctr = gate + (size_t)xcd * kMsGateWords;
    int table_of = -1;

    // Batch entry b runs through all its hops before the XCD takes the next entry (hop h reads the tap hop h - 1 wrote: Xin for the
    // first, then Xtaps + (h - 1) taps; it writes Xtaps + h taps): the rows a hop gathers first were written by this XCD a moment ago
    // and are still in its L2 (the image stores the lowest row bands last).
    // Team barrier (used between the hops of an entry, and behind the layout pre-phase): see the comment at its first use below.
    auto team_barrier = [&](bool drain) -> bool {
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wv == 0) {
            // ONE monotonic counter per team: arrival = fetch-add, the barrier opens when the count reaches the next multiple of 32 (exactly 32
            // arrivals per barrier, so launches start on a multiple and nothing has to be reset; the last arriver is through after one L2
            // round trip, the others after their next poll).  Wrap-around: compared as a signed difference.
            unsigned t = 1u;
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(ctr) : "memory");
            if ((t & 31u) != 31u) {
                const unsigned target = (t & ~31u) + 32u;
                const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
                bool open = false;
                while (!open) {
                    __builtin_amdgcn_s_sleep(2);
                    unsigned v = 0u;
                    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
                    open = (int)(v - target) >= 0;
                    if (!open && __builtin_amdgcn_s_memrealtime() - tb > (unsigned long long)tmo_ticks) break;
                }
                if (!open && lane == 0) {           // a team mate is gone: abandon the launch (the repair kernel redoes the whole chain)
                    ag_store(cs + kCsPoison, 1u);
                    ag_store(cs + kCsRepair, 1u);
                    s_ctl[3] = 1u;
                }
            }
        }
        __syncthreads();
        return s_ctl[3] == 0u;
    };

    // Wide rows (IDX: W = 64 / 96 / 128 floats): a batch entry is W / 32 independent SLABS of 32 columns -- the same image, the same
    // sums; a gather addresses {row, slab column} through a buffer resource whose stride is the row (range check: row < N).  The XCD's
    // work list is the (entry, slab) pairs x, x + 8, ...
    const int nslab = IDX ? (W >> 5) : 1;
    const unsigned rowBytes = IDX ? (unsigned)W * 4u : 128u;
    const size_t entryBytes = (size_t)N * rowBytes;
    for (int ve = xcd; ve < B * nslab; ve += 8)
      for (int hop = 0; hop < nhops; ++hop) {
        const int b = IDX ? ve / nslab : ve;
        const unsigned fgs = IDX ? fg16 + (unsigned)(ve - b * nslab) * 128u : fg16;
        const char* src = hop == 0 ? reinterpret_cast<const char*>(Xin) : reinterpret_cast<const char*>(Xtaps) + (size_t)(hop - 1) * tapStrideBytes;
        char* dst = reinterpret_cast<char*>(Xtaps) + (size_t)hop * tapStrideBytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)b * entryBytes), IDX ? (short)rowBytes : (short)0, IDX ? N : (int)tapBytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (size_t)b * entryBytes), IDX ? (short)rowBytes : (short)0, nostore ? 0 : (IDX ? N : (int)tapBytes), 0x00020000);
        if constexpr (XP) {
            if (hop == 0) {
                // Boundary layout pass of this entry (graphML.py:2131-2135 zero-pad + the permute the reference does at :170): x[b][g][n] (reference layout,
                // n < Nin) -> tap 0 rows X0[b][n][0..32), rows n >= Nin zero; xmask (backward behind a fused ReLU): entries whose mask is <= 0 give 0.
                // The team's 128 waves take blocks of 64 nodes, HIGHEST first: the rows written last are the ones hop 0's sweep gathers first and
                // are still in the XCD's L2 -- what made the later hops of an entry faster than its first.  Three blocks of loads in flight per
                // wave; transposed through a wave-private LDS tile (LDS operations of one wave execute in order: no barrier).
                // (loads through ONE buffer resource with per-lane byte offsets: 32 row base pointers in SGPRs, hoisted out of the block loop, would be
                //  live across the asm bodies and spill; the row pitch passes through an opaque asm per block so that its multiples are not hoisted)
                const size_t xbytes = (size_t)32u * (size_t)Nin * 4u;
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xref + (size_t)b * 32u * (size_t)Nin), 0, (int)xbytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)((xmask ? xmask : xref) + (size_t)b * 32u * (size_t)Nin), 0, (int)xbytes, 0x00020000);
                const bool masked = xmask != nullptr;
                float* tile = &s_tile[wv][0];
                const int nblk = (N + 63) >> 6;
                unsigned lx = lane;
                asm volatile("; layout phase" : "+v"(lx));
                const unsigned px = lx >> 3, fx = lx & 7u;
                float va[32], vb[32], vc[32];
                auto ld = [&](float (&v)[32], int blk) {
                    unsigned pitch = (unsigned)Nin * 4u;
                    asm volatile("" : "+s"(pitch));
                    const int n = blk * 64 + (int)lx;
                    unsigned off = (blk >= 0 && n < Nin) ? (unsigned)n * 4u : 0xfffffff0u;     // (out of range: the load returns 0 without a memory request)
#pragma unroll
                    for (int g = 0; g < 32; ++g) {
                        float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
                        if (masked && !(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, off, 0, 0)) > 0.f)) t = 0.f;
                        v[g] = t;
                        off = off < 0xfffffff0u ? off + pitch : off;
                    }
                };
                auto put = [&](const float (&v)[32], int blk) {   // one block through the wave's LDS tile, out as full 128-byte rows
#pragma unroll
                    for (int g = 0; g < 32; ++g) tile[g * 65 + lx] = v[g];
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const unsigned r = (unsigned)it * 8u + px;
                        f32x4 o;
                        o.x = tile[(4 * fx + 0) * 65 + r], o.y = tile[(4 * fx + 1) * 65 + r], o.z = tile[(4 * fx + 2) * 65 + r], o.w = tile[(4 * fx + 3) * 65 + r];
                        // (range check: rows >= N dropped.  Issued through asm: the compiler's wait-count pass then counts loads only -- with a store it
                        //  can see pending it waits for vmcnt(0), i.e. for the blocks of loads just issued; the s_nop: a 16-byte store reads its data
                        //  registers up to two cycles after issue)
                        const unsigned so = ((unsigned)blk * 64u + r) * 128u + fx * 16u;
                        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(o), "v"(so), "s"(rs) : "memory");
                    }
                };
                // three register sets in rotation by code position (a copy between sets would wait for the loads it reads): two blocks of loads are in
                // flight while the third is transposed and stored
                constexpr int kStep = kMsWavesPerXcd;
                int blk = nblk - 1 - (int)wid;
                ld(va, blk);
                ld(vb, blk - kStep);
                for (; blk >= 0; blk -= 3 * kStep) {
                    ld(vc, blk - 2 * kStep);
                    put(va, blk);
                    if (blk - kStep < 0) break;
                    ld(va, blk - 3 * kStep);
                    put(vb, blk - kStep);
                    if (blk - 2 * kStep < 0) break;
                    ld(vb, blk - 4 * kStep);
                    put(vc, blk - 2 * kStep);
                }
                if (!team_barrier(true)) return;       // hop 0 gathers rows other CUs of the team stored
            }
        }
        for (int pass = 0; pass < passes; ++pass) {
            const size_t pw = (size_t)pass * kMsWavesPerXcd + wid;
            if (table_of != pass) {   // wave-private copy (LDS operations of one wave execute in order: no barrier)
                for (int i = (int)lane; i < S * 32; i += 64) s_rows[wv][i] = rows[pw * (size_t)(S * 32) + i];
                table_of = pass;
            }
            const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(ent + pw * streamWords), 0, (int)(streamWords * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)((UNI ? reinterpret_cast<const float*>(ent) : val) + pw * streamWords), 0, (int)(streamWords * 4), 0x00020000);
            unsigned scur = 0, sit = (unsigned)(rounds / U);
            // scalar prefetch (PF): in every step each wave touches one source row of the iteration pf_lead iterations ahead with two
            // s_loads (both 64-byte halves) -- the first touch of a row then travels through the scalar cache's queue, not through the
            // vector cache's, whose in-order returns would hold the wave's other gathers behind an HBM round trip
            const unsigned smaxpf = tapBytes - (unsigned)(U * S) * 0x4000u - 128u;
            unsigned spfr = (unsigned)pf_lead * (U * chunkBytes) + wid * 128u, spfc = spfr < smaxpf ? spfr : smaxpf, sdummy;
            const char* xptr = src + (size_t)b * entryBytes;
            unsigned long long tl0, tl1;
            if constexpr (PH != 0) asm volatile("MS_ZERO %0" ::"n"(S) : GF_MS_CLOBBERS);   // (the phases of compiler code may have used any register)
            if constexpr (PF) {
                asm volatile("MS_BODY %20, %21, %22, 1, %23, %7, %8, %9, %10, %11, %12, %13, %14, %15, %16, %17, %18, %19, %0, %1, %2, %3, %4, %5, %6, %="
                             : "+s"(scur), "+s"(sit), "+s"(spfr), "+s"(spfc), "=&s"(sdummy), "=&s"(tl0), "=&s"(tl1)
                             : "s"(rs), "s"(ro), "s"(re), "s"(rv), "v"(fgs), "v"(slotbit), "v"(evoff), "v"(rowlds), "s"(smask), "s"(uval),
                               "s"(xptr), "s"(smaxpf), "s"(U * chunkBytes), "n"(S), "n"(UNI), "n"(IDX), "n"(D)
                             : GF_MS_CLOBBERS);
            } else {   // (no scalar prefetch: its six operands are not materialised)
                (void)spfr; (void)spfc; (void)sdummy; (void)xptr; (void)smaxpf;
                asm volatile("MS_BODY %15, %16, %17, 0, %18, %4, %5, %6, %7, %8, %9, %10, %11, %12, %13, %14, %14, %14, %0, %1, %14, %14, %14, %2, %3, %="
                             : "+s"(scur), "+s"(sit), "=&s"(tl0), "=&s"(tl1)
                             : "s"(rs), "s"(ro), "s"(re), "s"(rv), "v"(fgs), "v"(slotbit), "v"(evoff), "v"(rowlds), "s"(smask), "s"(uval),
                               "s"(0), "n"(S), "n"(UNI), "n"(IDX), "n"(D)
                             : GF_MS_CLOBBERS);
            }
            if constexpr (HUB) {
                // Hub rows (gf_msweep_image.h): this wave's quads of 4 octets x 8 positions -- rows too long for a group, each an ascending-column
                // fmaf chain of its own (a split quad: 32 partial chains of ONE row, added in a fixed tree).  Compiler code between the asm
                // bodies; gathers and stores through the same buffer resources (range check: gaps and missing rows cost no memory request).
                // hubpack = {offsets of the passes x 128 waves' blocks [nw + 1], total words, the blocks, (weighted: the values, same shape)}
                const int nw = passes * kMsWavesPerXcd;
                const uint32_t* hw = hubpack + nw + 2;
                const uint32_t hwords = hubpack[nw + 1];
                uint32_t at = hubpack[pw];
                const uint32_t end = hubpack[pw + 1];
                // (per-lane values of this phase are re-derived here from an opaque copy of the lane id: hoisted out of the entry / hop loops they
                //  would be live ACROSS the asm body, where the compiler has only v0-v23, and spill to scratch)
                unsigned lh = lane;
                asm volatile("; hub phase" : "+v"(lh));
                const unsigned fgh = (lh & 7u) * 16u + (IDX ? (unsigned)(ve - b * nslab) * 128u : 0u), p4 = (lh >> 3) * 4u;
                // wide rows: the body's resources address {row, column}; this phase uses byte offsets (row index capped so that a gap's offset cannot wrap into range)
                const __amdgpu_buffer_rsrc_t rsh = IDX ? __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)b * entryBytes), 0, (int)entryBytes, 0x00020000) : rs;
                const __amdgpu_buffer_rsrc_t roh = IDX ? __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (size_t)b * entryBytes), 0, nostore ? 0 : (int)entryBytes, 0x00020000) : ro;
#define GF_HUB_OFF(E) (IDX ? ((((E) >> 7) < 0x7fffffu ? ((E) >> 7) : 0x7fffffu)) * rowBytes + fgh : (E) + fgh)
                while (at < end) {
                    const int Lq = (int)hw[at];
                    const bool split = hw[at + 1] != 0u;
                    const u32x4 ro4 = *reinterpret_cast<const u32x4*>(hw + at + 4 + p4);
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                    // two steps in flight: the gathers of step j + 1 are issued before the sums of step j, the entries of step j + 2 before those
                    // (loads return in order, so an entry word has always arrived by the time its gathers are due; the stream is padded past the
                    // last quad, and what is read beyond a quad's end is gathered and dropped)
                    const uint32_t* ep = hw + at + 36 + p4;
#define GF_HUB_LD(E, V, P)                                                                                             \
    do {                                                                                                               \
        E = *reinterpret_cast<const u32x4*>(P);                                                                        \
        if (!UNI) V = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>((P) + hwords));                   \
    } while (0)
#define GF_HUB_GATHER(X, E)                                                                                            \
    do {                                                                                                               \
        X[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsh, GF_HUB_OFF(E.x), 0, 0));                  \
        X[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsh, GF_HUB_OFF(E.y), 0, 0));                  \
        X[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsh, GF_HUB_OFF(E.z), 0, 0));                  \
        X[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsh, GF_HUB_OFF(E.w), 0, 0));                  \
    } while (0)
#define GF_HUB_FMA1(A, X, V) A.x = __builtin_fmaf(V, X.x, A.x), A.y = __builtin_fmaf(V, X.y, A.y), A.z = __builtin_fmaf(V, X.z, A.z), A.w = __builtin_fmaf(V, X.w, A.w)
#define GF_HUB_SUM(X, V) GF_HUB_FMA1(a0, X[0], V.x), GF_HUB_FMA1(a1, X[1], V.y), GF_HUB_FMA1(a2, X[2], V.z), GF_HUB_FMA1(a3, X[3], V.w)
                    u32x4 eA, eB;
                    f32x4 vA = {1.f, 1.f, 1.f, 1.f}, vB = vA, vC = vA, vD = vA, xA[4], xB[4];
                    GF_HUB_LD(eA, vA, ep);
                    GF_HUB_LD(eB, vB, ep + 32);
                    GF_HUB_GATHER(xA, eA);
                    GF_HUB_LD(eA, vC, ep + 64);
                    int j = 0;
                    for (; j + 1 < Lq; j += 2) {            // steps j (xA, vA) and j + 1 (xB, vB); eB = entries of j + 1, eA = entries of j + 2
                        GF_HUB_GATHER(xB, eB);
                        GF_HUB_LD(eB, vD, ep + 96);
                        GF_HUB_SUM(xA, vA);
                        GF_HUB_GATHER(xA, eA);
                        GF_HUB_LD(eA, vA, ep + 128);
                        GF_HUB_SUM(xB, vB);
                        vB = vD;
                        { const f32x4 t = vA; vA = vC; vC = t; }
                        ep += 64;
                    }
                    if (j < Lq) GF_HUB_SUM(xA, vA);
#undef GF_HUB_SUM
#undef GF_HUB_FMA1
#undef GF_HUB_GATHER
#undef GF_HUB_LD
                    if (split) {                            // octets in order, then the eight positions by xor 1, 2, 4 (lanes 8 p + fg: xor 8, 16, 32)
                        f32x4 t = ((a0 + a1) + a2) + a3;
#pragma unroll
                        for (int m = 8; m < 64; m <<= 1) {
                            f32x4 o;
                            o.x = __shfl_xor(t.x, m), o.y = __shfl_xor(t.y, m), o.z = __shfl_xor(t.z, m), o.w = __shfl_xor(t.w, m);
                            t = t + o;
                        }
                        if (UNI) t = t * uval;
                        if (p4 == 0u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), roh, GF_HUB_OFF(ro4.x), 0, 0);
                    } else {
                        if (UNI) a0 = a0 * uval, a1 = a1 * uval, a2 = a2 * uval, a3 = a3 * uval;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a0), roh, GF_HUB_OFF(ro4.x), 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a1), roh, GF_HUB_OFF(ro4.y), 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a2), roh, GF_HUB_OFF(ro4.z), 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a3), roh, GF_HUB_OFF(ro4.w), 0, 0);
                    }
                    at += 36u + (uint32_t)Lq * 32u;
                }
            }
#undef GF_HUB_OFF
            const unsigned long long t1 = trace ? __builtin_amdgcn_s_memtime() : 0ull;

            const bool dependent = nhops > 1 && pass == passes - 1 && hop + 1 < nhops;   // the next hop gathers what this one stores
            if (use_barrier || dependent) {
                // XCD barrier (no state to zero between launches: a captured launch replays as is).  Scalar atomics:
                // they execute in the XCD's L2 and wait on lgkmcnt, not on the stores' vmcnt.  Between the hops of an entry the barrier
                // orders this hop's stores (acknowledged by the L2: vmcnt(0)) before the next hop's gathers from the other CUs of the team --
                // the team IS the set of workgroups on one XCC (census above), so counters and rows live in the one L2 they all use.  The
                // census has seen all 256 workgroups resident; a barrier that still does not open within the time limit (a team mate died)
                // abandons the launch the same way a bad census does.
                if (!team_barrier(dependent)) return;
            }
            const int slot = ((ve >> 3) * nhops + hop) * passes + pass;   // (entry, hop, pass) bodies done by this XCD
            if (trace && wid == 0 && lane == 0 && slot < 64) {    // (experiments) phase stamps of the XCD's first wave: previous stores drained + entries
                unsigned long long* t = trace + ((size_t)xcd * 64 + slot) * 8;   // loaded (= loop start), loop end, stores issued, barrier passed (= the next body's start)
                t[0] = 1ull; t[1] = tl0; t[2] = tl1; t[3] = t1; t[4] = __builtin_amdgcn_s_memtime();
                t[5] = __builtin_amdgcn_s_memrealtime();   // (100 MHz: the shader clock under this load = d t[4] / d t[5] x 100 MHz)
            }
        }
      }
}

// The chain, row by row, for launches the sweep abandoned (bad census, time limit): gated on the launch's flag -- when the sweep ran, every
// workgroup leaves at once (~2 us per chain).  Workgroup = one batch entry at a time, through all its hops (a hop of an entry needs only
// that entry's previous tap: the hand-over stays inside the workgroup -- __syncthreads + an agent-scope fence -- and needs no co-residency).
// Per row: the stored CSR's entries in ascending column order, one fmaf per entry (uniform GSOs: sum, then scale once) = the sums of
// spmm_msweep_kernel and spmm_sell_kernel, bit for bit.
__global__ __launch_bounds__(512) void spmm_msweep_repair_kernel(unsigned* __restrict__ cs, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                 const float* __restrict__ val, const int32_t* __restrict__ rowid, const float* __restrict__ Xin,
                                                                 float* __restrict__ Xtaps, size_t tapStride, int nhops, int N, int B, int W, int uniform, float uval,
                                                                 unsigned* __restrict__ status, const float* __restrict__ xref, const float* __restrict__ xmask, int Nin, int split_above) {
    if (ag_load(cs + kCsRepair) == 0u) return;
    if (xref)   // the abandoned launch would have written tap 0 itself (layout pre-phase): x[b][g][n] -> X0[b][n][g], rows >= Nin zero, masked
        for (int b = blockIdx.x; b < B; b += gridDim.x) {
            float* x0 = const_cast<float*>(Xin) + (size_t)b * N * W;
            for (int64_t idx = threadIdx.x; idx < (int64_t)N * W; idx += blockDim.x) {
                const int n = (int)(idx / W), g = (int)(idx - (int64_t)n * W);
                float t = 0.f;
                if (n < Nin) {
                    const size_t at = ((size_t)b * W + g) * (size_t)Nin + n;
                    t = xref[at];
                    if (xmask && !(xmask[at] > 0.f)) t = 0.f;
                }
                x0[idx] = t;
            }
            __syncthreads();
            __threadfence();
            __syncthreads();
        }
    const int W4 = W >> 2;                                  // float4 columns per row
    for (int b = blockIdx.x; b < B; b += gridDim.x)
        for (int hop = 0; hop < nhops; ++hop) {
            const float* src = (hop == 0 ? Xin : Xtaps + (size_t)(hop - 1) * tapStride) + (size_t)b * N * W;
            float* dst = Xtaps + (size_t)hop * tapStride + (size_t)b * N * W;
            for (int64_t idx = threadIdx.x; idx < (int64_t)N * W4; idx += blockDim.x) {
                const int p = (int)(idx / W4), c4 = (int)(idx - (int64_t)p * W4) * 4;
                const int q0 = rowptr[p], q1 = rowptr[p + 1];
                auto chain = [&](int a, int b2) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int q = a; q < b2; ++q) {
                        const float4 x = *reinterpret_cast<const float4*>(src + (size_t)col[q] * W + c4);
                        const float v = uniform ? 1.f : val[q];
                        acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y); acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
                    }
                    return acc;
                };
                auto add4 = [](const float4& a, const float4& b2) { return make_float4(a.x + b2.x, a.y + b2.y, a.z + b2.z, a.w + b2.w); };
                float4 acc;
                if (split_above > 0 && q1 - q0 > split_above) {
                    // a SPLIT hub row of the sweep (gf_msweep_image.h): 32 partial chains over contiguous runs of L entries, added octets first, then
                    // the eight positions as a binary tree -- the sweep's order, so that a repaired launch has its bits
                    const int L = (q1 - q0 + 31) / 32;
                    float4 ps[8];
                    for (int pp = 0; pp < 8; ++pp) {
                        float4 t = chain(std::min(q1, q0 + (4 * pp) * L), std::min(q1, q0 + (4 * pp + 1) * L));
                        for (int o = 1; o < 4; ++o) t = add4(t, chain(std::min(q1, q0 + (4 * pp + o) * L), std::min(q1, q0 + (4 * pp + o + 1) * L)));
                        ps[pp] = t;
                    }
                    acc = add4(add4(add4(ps[0], ps[1]), add4(ps[2], ps[3])), add4(add4(ps[4], ps[5]), add4(ps[6], ps[7])));
                } else {
                    acc = chain(q0, q1);
                }
                if (uniform) { acc.x *= uval; acc.y *= uval; acc.z *= uval; acc.w *= uval; }
                *reinterpret_cast<float4*>(dst + (size_t)rowid[p] * W + c4) = acc;
            }
            __syncthreads();
            __threadfence();
            __syncthreads();
        }
    if (blockIdx.x == 0 && threadIdx.x == 0 && status) {   // host-visible (pinned) word: the library reads it at its next call and stops fusing
        const unsigned old = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (plain load + store: no PCIe atomics needed; one writer per launch)
        __hip_atomic_store(status, old | (ag_load(cs + kCsPoison) ? kMsStatusTimeout : kMsStatusCensus), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

unsigned long long* g_trace = nullptr;
constexpr size_t kTraceBytes = 8 * 64 * 8 * sizeof(unsigned long long);

int cu_count() {   // of the CURRENT device (a process may drive several)
    static std::mutex mu;
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (cus[dev] == 0) {
        int n = 0;
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus[dev] = n > 0 ? n : -1;
    }
    return cus[dev];
}

// One gate slot per stream: two launches of one stream never overlap, and launches on different streams never share barrier words (the
// first 15 streams a process uses get a slot of their own, the rest share the last one).
int slot_of(hipStream_t st) {
    static std::mutex mu;
    static hipStream_t seen[kMsGateSlots];
    static int nseen = 0;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == st) return i;
    if (nseen < kMsGateSlots - 1) {
        seen[nseen] = st;
        return nseen++;
    }
    return kMsGateSlots - 1;
}

// Host-visible status of the fused chains of this process: a pinned word the repair kernel ORs its reason into.  Allocated at plan
// creation (gf_msweep_status_word), never during a launch (a launch may be under stream capture).
std::atomic<unsigned*> g_status{nullptr};
std::atomic<int> g_fuse_off{0};   // 1 = a repair was seen (or GFHIP_MSWEEP_FUSE=0): gf_khop runs one launch per hop from here on

}  // namespace

unsigned* gf_msweep_status_word() {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    unsigned* w = g_status.load();
    if (!w) {
        if (hipHostMalloc((void**)&w, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
        *w = 0u;
        g_status.store(w);
    }
    return w;
}

// 1 while gf_khop may fuse the K - 1 hops into one launch.  Off: GFHIP_MSWEEP_FUSE=0 in the environment, or -- from the first call after
// the fact -- a launch of this process had to be repaired (reported once on stderr; gf_msweep_status tells which).
bool gf_msweep_fusion_allowed() {
    static const int env_off = [] { const char* e = getenv("GFHIP_MSWEEP_FUSE"); return e && atoi(e) == 0 ? 1 : 0; }();
    if (env_off || g_fuse_off.load()) return false;
    const unsigned* w = g_status.load();
    const unsigned st = w ? *reinterpret_cast<const volatile unsigned*>(w) : 0u;
    if (st) {
        if (!g_fuse_off.exchange(1))
            fprintf(stderr, "gfhip: a fused K-hop chain was abandoned (%s%s) and recomputed by the repair kernel; results are unaffected, "
                            "the chain runs one launch per hop from here on\n",
                    (st & kMsStatusCensus) ? "workgroups not spread 32 per XCD" : "", (st & kMsStatusTimeout) ? " grid not resident within the time limit" : "");
        return false;
    }
    return true;
}

extern "C" int gf_msweep_status(uint32_t* flags, int32_t* fusion_on) {
    const unsigned* w = g_status.load();
    if (flags) *flags = w ? *reinterpret_cast<const volatile unsigned*>(w) : 0u;
    if (fusion_on) *fusion_on = gf_msweep_fusion_allowed() ? 1 : 0;
    return GF_OK;
}

void gf_msweep_status_reset() {   // experiments (gf_tune("spmm_status_reset", 1)): tests of the repair path start from a clean process state
    unsigned* w = g_status.load();
    if (w) *reinterpret_cast<volatile unsigned*>(w) = 0u;
    g_fuse_off.store(0);
}

bool gf_msweep_applicable(const gf_csr_dev& m, int N, int B, int W) {
    // rows of 32 columns, or of 2 / 3 / 4 slabs of 32 (the image is the same: a slab is a 128-byte column block of the row); an image; one
    // workgroup per CU on a 256-CU device (8 XCDs x 32 CUs x 4 SIMDs = the 128 waves per XCD of the image); 32-bit byte offsets inside a
    // tap; enough (entry, slab) pairs: an XCD walks a pair in ~65 us per hop however many of the 8 are busy, SELL-8 takes 0.02 ms per pair -- from 5 pairs
    // on the sweep wins (5 .. 7 pairs: 0.074-0.076 against 0.094-0.107 ms per hop at config 4's graph; profiles/r06_l_share/share_short.log)
    return (W == 32 || W == 64 || W == 96 || W == 128) && m.ms_ent && m.ms_rows && m.ms_sets >= 2 * kMsDepth && (m.ms_uniform || m.ms_val) &&
           cu_count() == 256 && B * (W / 32) >= g_tune.spmm_minwork && (int64_t)N * 128 < (int64_t)kMsPad;
}

int gf_msweep_launch(const gf_csr_dev& m, const float* Xin, float* Xtaps, int64_t tapStride, int nhops, int N, int B, int W, hipStream_t st,
                     const float* xref, const float* xmask, int Nin) {
    unsigned* gate = m.ms_gate + (size_t)slot_of(st) * kMsSlotWords;
    const int use_barrier = g_tune.spmm_bar;
    dim3 grid(256), block(kThreads);
    const unsigned src_mask = g_tune.spmm_srcmask ? (unsigned)g_tune.spmm_srcmask : 0xffffffffu;   // experiments (timing only): confine the gathers to a window
    unsigned long long* trace = nullptr;
    if (g_tune.spmm_trace) {   // (experiments) 8 XCDs x 64 (entry, hop) slots x 8 stamps, read back with gf_debug_msweep_trace
        if (!g_trace) GF_HIP(hipMalloc((void**)&g_trace, kTraceBytes));
        GF_HIP(hipMemsetAsync(g_trace, 0, kTraceBytes, st));
        trace = g_trace;
    }
    const bool wide = W != 32;                  // slabs of 32 columns: {row, column} addressing, no scalar prefetch
    const int pfd = g_tune.spmm_pfd > 0 ? g_tune.spmm_pfd : (g_tune.spmm_pfd == 0 ? std::max(1, (m.ms_rounds + 7) / 14) : 0);
    const bool pf = pfd > 0 && !wide;
    const bool deep = g_tune.spmm_depth != 5;   // ring of 10 gathers; 5: experiments
    // A fused chain depends on its XCD barriers (hop h + 1 gathers what hop h stored): it is launched COOPERATIVELY -- the runtime starts the
    // grid only when all 256 workgroups can be resident together -- and opens with the census (kernel); the repair kernel behind it runs
    // only if the sweep abandoned the launch.  A single hop has no such dependence and takes the plain launch.
    const bool chained = nhops > 1 || use_barrier || xref != nullptr;   // (the layout pre-phase hands rows over inside the team too)
    GF_REQUIRE_ARG(xref == nullptr || W == 32, "gf_msweep_launch: the layout pre-phase takes 32-column rows");
    const uint32_t* a_ent = m.ms_ent;
    const float* a_val = m.ms_val;
    const uint32_t* a_rows = m.ms_rows;
    size_t a_stride = (size_t)tapStride * 4;
    int a_nhops = nhops, a_N = N, a_B = B, a_passes = m.ms_passes, a_rounds = m.ms_rounds, a_bar = use_barrier, a_pfd = pfd,
        a_nostore = g_tune.spmm_store == 3, a_census = chained ? 1 + (g_tune.spmm_census > 0 ? g_tune.spmm_census : 0) : 0;
    float a_uval = m.sell_uval;
    unsigned a_mask = src_mask;
    int a_W = W;
    const uint32_t* a_hub = m.ms_hub;
    int a_Nin = Nin;
    unsigned a_tmo = (unsigned)(g_tune.spmm_tmo_ms > 0 ? g_tune.spmm_tmo_ms : 2000) * 100000u;   // s_memrealtime ticks (100 MHz)
    void* args[] = {&a_ent, &a_val, &a_rows, &Xin, &Xtaps, &a_stride, &a_nhops, &a_N, &a_B, &a_passes, &a_rounds, &gate, &a_bar, &a_uval, &a_mask,
                    &a_pfd, &a_nostore, &trace, &a_census, &a_tmo, &a_W, &a_hub, &xref, &xmask, &a_Nin};
    hipError_t lerr = hipSuccess;
#define GF_MS(SV, UV, PV, DV, XV) \
    do {                                                                                                                               \
        if (xref && (XV) == 0) {                                                                                                       \
            if (m.ms_hub) GF_MSH(SV, UV, PV, DV, 0, 3);                                                                                \
            else GF_MSH(SV, UV, PV, DV, 0, 2);                                                                                         \
        } else if (m.ms_hub) GF_MSH(SV, UV, PV, DV, XV, 1);                                                                            \
        else GF_MSH(SV, UV, PV, DV, XV, 0);                                                                                            \
    } while (0)
#define GF_MSH(SV, UV, PV, DV, XV, HV)                                                                                                    \
    do {                                                                                                                               \
        if (chained)                                                                                                                   \
            lerr = hipLaunchCooperativeKernel((const void*)spmm_msweep_kernel<SV, UV, PV, DV, XV, HV>, grid, block, args, 0, st); \
        else                                                                                                                           \
            hipLaunchKernelGGL((spmm_msweep_kernel<SV, UV, PV, DV, XV, HV>), grid, block, 0, st, a_ent, a_val, a_rows, Xin, Xtaps, a_stride, a_nhops, a_N, a_B, \
                               a_passes, a_rounds, gate, a_bar, a_uval, a_mask, a_pfd, a_nostore, trace, a_census, a_tmo, a_W, a_hub, xref, xmask, a_Nin); \
    } while (0)
#define GF_MS_P(SV, UV, DV)                            \
    do {                                              \
        if (pf) GF_MS(SV, UV, 1, DV, 0);              \
        else GF_MS(SV, UV, 0, DV, 0);                 \
    } while (0)
#define GF_MS_D(SV, UV)                               \
    do {                                              \
        if (wide) GF_MS(SV, UV, 0, 10, 1);            \
        else if (deep) GF_MS_P(SV, UV, 10);           \
        else GF_MS_P(SV, UV, 5);                      \
    } while (0)
    if (m.ms_uniform) {
        switch (m.ms_sets) {
            case 10: if (wide) GF_MS(10, 1, 0, 5, 1); else GF_MS_P(10, 1, 5); break;
            case 15: GF_MS_D(15, 1); break;
            case 20: GF_MS_D(20, 1); break;
            default: GF_MS_D(25, 1); break;
        }
    } else {
        switch (m.ms_sets) {
            case 10: if (wide) GF_MS(10, 0, 0, 5, 1); else GF_MS_P(10, 0, 5); break;
            case 15: GF_MS_D(15, 0); break;
            case 20: GF_MS_D(20, 0); break;
            default: GF_MS_D(25, 0); break;
        }
    }
#undef GF_MS_D
#undef GF_MS_P
#undef GF_MS
#undef GF_MSH
    GF_HIP(lerr);
    GF_LAUNCH_CHECK("spmm_msweep_kernel");
    if (chained && !a_nostore && src_mask == 0xffffffffu) {
        hipLaunchKernelGGL(spmm_msweep_repair_kernel, dim3((unsigned)(B < 512 ? B : 512)), dim3(512), 0, st, gate + kMsCensusWord, m.rowptr, m.col, m.val, m.rowid,
                           Xin, Xtaps, (size_t)tapStride, nhops, N, B, W, m.ms_uniform, m.sell_uval, g_status.load(), xref, xmask, Nin,
                           m.ms_hub ? std::max(m.ms_hub_limit, m.ms_hub_split) : 0);
        GF_LAUNCH_CHECK("spmm_msweep_repair_kernel");
    }
    return GF_OK;
}

extern "C" int gf_debug_msweep_info(const gf_plan* plan, int32_t op, int32_t* out) {   // {sets, passes, rounds, fill x 1000, hub rows, split hub rows, hub limit, split limit}
    GF_REQUIRE_ARG(plan != nullptr && out != nullptr && (op == GF_OP_FWD || op == GF_OP_BWD), "gf_debug_msweep_info: bad argument");
    const gf_csr_dev& m = plan->mat[op];
    out[0] = m.ms_sets, out[1] = m.ms_passes, out[2] = m.ms_rounds, out[3] = (int32_t)(m.ms_fill * 1000.0 + 0.5);
    out[4] = m.ms_hub_rows, out[5] = m.ms_hub_split_rows, out[6] = m.ms_hub_limit, out[7] = m.ms_hub_split;
    return GF_OK;
}

extern "C" int gf_debug_msweep_trace(unsigned long long* out) {   // [8][64][8]; experiments only (filled when gf_tune("spmm_trace", 1))
    GF_REQUIRE_ARG(out != nullptr && g_trace != nullptr, "gf_debug_msweep_trace: no trace");
    GF_HIP(hipDeviceSynchronize());
    GF_HIP(hipMemcpy(out, g_trace, kTraceBytes, hipMemcpyDeviceToHost));
    return GF_OK;
}

size_t gf_msweep_gate_bytes() { return (size_t)kMsGateSlots * kMsSlotWords * sizeof(unsigned); }
