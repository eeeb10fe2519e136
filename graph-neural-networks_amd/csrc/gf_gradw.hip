// gf_gradw.hip -- gradient of the filter taps and the bias (autograd of reference graphML.py:170-175 wrt h, b):
//     dh[f,e,k,g] = sum_{b,n} Z[t(e,k), b, n, g] * dY[b, n, f]         dbias[f] = sum_{b,n} dY[b, n, f]
// A tall-skinny reduction GEMM  dHm[T*G, F] = Z^T[T*G, R] * P0[R, F]  over R = B*N rows.  HBM-bound: reads Z and P0
// once (4*R*(T*G + F) bytes), 2*R*T*G*F flops.
//
// Stage 1 (MFMA, v_mfma_f32_32x32x2_f32): each wavefront owns a contiguous strip of rows and keeps up to 8
//   32x32 output tiles (c-tile x f-tile) in accumulators; both operands are read straight from HBM, fully
//   coalesced, with no LDS: A[i = g][k = row] is 32 consecutive floats of one Z row per half-wave, B[k = row][j = f]
//   32 consecutive floats of one P0 row.  Each wave writes its partial tiles to the workspace.
// Stage 2/3: fixed-order tree over the per-wave partials (no atomics -> bitwise deterministic), then scatter into the
//   reference parameter layout dh[F,E,K,G] (tap 0 is shared by all e, so its gradient is written for every e).
#include "gf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kMaxWaves = 2048;  // stage-1 partial producers
constexpr int kSlices = 32;      // stage-2 fan-in

struct Geo {
    int T, numGI, numCT, numFT, ctp, cpasses, passes, rowsPerWave, numWaves, strips;
    int64_t R;
    size_t off_partial_b, off_mid, off_mid_b, bytes;  // float offsets into the workspace (off_* in floats)
    int64_t tileOutputs, biasOutputs;
};

Geo make_geo(int B, int N, int G, int F, int E, int K) {
    Geo g{};
    g.T = gf_num_taps(E, K);
    g.R = (int64_t)B * N;
    g.numGI = (G + 31) / 32;
    g.numCT = g.T * g.numGI;
    g.numFT = (F + 31) / 32;
    g.cpasses = (g.numCT + 7) / 8;
    g.ctp = (g.numCT + g.cpasses - 1) / g.cpasses;
    g.passes = g.cpasses * g.numFT;
    int64_t rpw = (g.R + kMaxWaves - 1) / kMaxWaves;
    if (rpw < 64) rpw = 64;
    rpw = (rpw + 7) & ~(int64_t)7;  // whole 8-row steps
    g.rowsPerWave = (int)rpw;
    g.numWaves = (int)((g.R + rpw - 1) / rpw);
    g.strips = (g.numWaves + kWaves - 1) / kWaves;
    const int wavesPadded = g.strips * kWaves;
    g.tileOutputs = (int64_t)g.passes * g.ctp * 1024;
    g.biasOutputs = (int64_t)g.numFT * 32;
    size_t off = (size_t)wavesPadded * g.tileOutputs;  // partial tiles
    g.off_partial_b = off;
    off += (size_t)wavesPadded * g.biasOutputs;
    g.off_mid = off;
    off += (size_t)kSlices * g.tileOutputs;
    g.off_mid_b = off;
    off += (size_t)kSlices * g.biasOutputs;
    g.bytes = off * sizeof(float);
    return g;
}

template <int CTP>
__global__ __launch_bounds__(kThreads) void grad_taps_kernel(const float* __restrict__ Z, const float* __restrict__ P0,
                                                             float* __restrict__ partial, float* __restrict__ partial_b,
                                                             int64_t R, int G, int F, int numGI, int numCT, int numFT,
                                                             int passes, int rowsPerWave) {
    const int pass = blockIdx.y;
    const int ft = pass % numFT, cpass = pass / numFT;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wg = blockIdx.x * kWaves + wave;
    const int64_t r_begin = (int64_t)wg * rowsPerWave;
    const int64_t r_end = min(R, r_begin + rowsPerWave);
    const int64_t tapStride = R * G;

    const int f = ft * 32 + l31;
    const bool fvalid = f < F;
    const float* zp[CTP];
    bool gvalid[CTP];
#pragma unroll
    for (int j = 0; j < CTP; ++j) {
        const int ct = cpass * CTP + j;
        const int t = ct / numGI, gi = ct - t * numGI;
        const int g = gi * 32 + l31;
        gvalid[j] = (ct < numCT) && (g < G);
        zp[j] = Z + (gvalid[j] ? (int64_t)t * tapStride + g : 0);
    }
    f32x16 acc[CTP];
#pragma unroll
    for (int j = 0; j < CTP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;

    // lane half h takes rows r0 + h, +2, +4, +6 of each 8-row step (k index of the 32x32x2 MFMA = row parity);
    // the 4 x (1 + CTP) loads of a step are issued before its 4 x CTP MFMAs.
    for (int64_t r0 = r_begin; r0 < r_begin + rowsPerWave; r0 += 8) {
        float p[4], a[4][CTP];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t r = r0 + 2 * s + half;
            const bool rv = r < r_end;
            p[s] = (rv && fvalid) ? P0[r * F + f] : 0.f;
#pragma unroll
            for (int j = 0; j < CTP; ++j) a[s][j] = (rv && gvalid[j]) ? zp[j][r * G] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bsum += p[s];
#pragma unroll
            for (int j = 0; j < CTP; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][j], p[s], acc[j], 0, 0, 0);
        }
    }

    float* pt = partial + ((int64_t)wg * passes + pass) * CTP * 1024;
#pragma unroll
    for (int j = 0; j < CTP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            pt[j * 1024 + i * 32 + l31] = acc[j][r];
        }
    if (cpass == 0) {
        const float other = __shfl_xor(bsum, 32, 64);
        if (half == 0) partial_b[((int64_t)wg * numFT + ft) * 32 + l31] = bsum + other;  // even rows + odd rows
    }
}

// out[s][o] = sum over rows w in slice s (ascending) of in[w][o];  slices partition [0, rows) evenly.
__global__ __launch_bounds__(kThreads) void reduce_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int rows, int64_t width, int slices) {
    const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int s = blockIdx.y;
    if (o >= width) return;
    const int per = (rows + slices - 1) / slices;
    const int w0 = s * per, w1 = min(rows, w0 + per);
    float acc = 0.f;
    for (int w = w0; w < w1; ++w) acc += in[(int64_t)w * width + o];
    out[(int64_t)s * width + o] = acc;
}

__global__ __launch_bounds__(kThreads) void finalize_taps_kernel(const float* __restrict__ mid, const float* __restrict__ mid_b,
                                                                 float* __restrict__ dh, float* __restrict__ dbias,
                                                                 int64_t tileOutputs, int biasOutputs, int slices, int G, int F,
                                                                 int E, int K, int numGI, int numCT, int numFT, int ctp) {
    const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (o < tileOutputs) {
        float acc = 0.f;
        for (int s = 0; s < slices; ++s) acc += mid[(int64_t)s * tileOutputs + o];
        const int jj = (int)(o & 31), i = (int)((o >> 5) & 31);
        const int64_t tileIdx = o >> 10;  // pass * ctp + j
        const int j = (int)(tileIdx % ctp), pass = (int)(tileIdx / ctp);
        const int ft = pass % numFT, cpass = pass / numFT;
        const int ct = cpass * ctp + j;
        const int t = ct / numGI, gi = ct - t * numGI;
        const int g = gi * 32 + i, f = ft * 32 + jj;
        if (dh != nullptr && ct < numCT && g < G && f < F) {
            if (t == 0) {
                for (int e = 0; e < E; ++e) dh[((int64_t)(f * E + e) * K) * G + g] = acc;
            } else {
                const int e = (t - 1) / (K - 1), k = (t - 1) % (K - 1) + 1;
                dh[((int64_t)(f * E + e) * K + k) * G + g] = acc;
            }
        }
    } else if (o < tileOutputs + biasOutputs && dbias != nullptr) {
        const int fb = (int)(o - tileOutputs);
        float acc = 0.f;
        for (int s = 0; s < slices; ++s) acc += mid_b[(int64_t)s * biasOutputs + fb];
        if (fb < F) dbias[fb] = acc;
    }
}

template <int CTP>
void launch_stage1(const Geo& g, const float* Z, const float* P0, float* ws, int G, int F, hipStream_t st) {
    hipLaunchKernelGGL((grad_taps_kernel<CTP>), dim3(g.strips, g.passes), dim3(kThreads), 0, st, Z, P0, ws,
                       ws + g.off_partial_b, g.R, G, F, g.numGI, g.numCT, g.numFT, g.passes, g.rowsPerWave);
}


// ---- column-panel variant of stage 1: Z is Zp[T][B][G/4][N][4], P0 is P0p[B][F/4][N][4] (gf_panel.hip).  Same strips, same
// accumulators, same partial layout (stages 2/3 are shared); only the operand addresses change: lane l31 = g (or f) reads
// component g % 4 of panel g / 4, so the 32 lanes of a half wave cover 8 panels x 16 bytes of one node and the four steps
// of an 8-row round walk each panel's 128-byte line once (L1-resident between the four loads).
template <int CTP>
__global__ __launch_bounds__(kThreads) void grad_taps_panel_kernel(const float* __restrict__ Zp, const float* __restrict__ P0p,
                                                                   float* __restrict__ partial, float* __restrict__ partial_b,
                                                                   int R, int N, int B, int G, int F, int numGI, int numCT, int numFT,
                                                                   int passes, int rowsPerWave) {
    const int pass = blockIdx.y;
    const int ft = pass % numFT, cpass = pass / numFT;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wg = blockIdx.x * kWaves + wave;
    const int r_begin = wg * rowsPerWave;  // R < 2^31 (checked by the launcher)
    const int r_end = min(R, r_begin + rowsPerWave);
    const int QG = G / 4, QF = F / 4;
    const int64_t N4 = (int64_t)N * 4;

    const int f = ft * 32 + l31;
    const bool fvalid = f < F;
    const float* pp = P0p + (fvalid ? (int64_t)(f >> 2) * N4 + (f & 3) : 0);
    const float* zp[CTP];
    bool gvalid[CTP];
#pragma unroll
    for (int j = 0; j < CTP; ++j) {
        const int ct = cpass * CTP + j;
        const int t = ct / numGI, gi = ct - t * numGI;
        const int g = gi * 32 + l31;
        gvalid[j] = (ct < numCT) && (g < G);
        zp[j] = Zp + (gvalid[j] ? ((int64_t)t * B * QG + (g >> 2)) * N4 + (g & 3) : 0);
    }
    f32x16 acc[CTP];
#pragma unroll
    for (int j = 0; j < CTP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;

    for (int r0 = r_begin; r0 < r_begin + rowsPerWave; r0 += 8) {
        float p[4], a[4][CTP];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = r0 + 2 * s + half;
            const bool rv = r < r_end;
            const int b = rv ? r / N : 0;
            const int n = rv ? r - b * N : 0;
            const int64_t zoff = (int64_t)b * QG * N4 + (int64_t)n * 4;
            const int64_t poff = (int64_t)b * QF * N4 + (int64_t)n * 4;
            p[s] = (rv && fvalid) ? pp[poff] : 0.f;
#pragma unroll
            for (int j = 0; j < CTP; ++j) a[s][j] = (rv && gvalid[j]) ? zp[j][zoff] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bsum += p[s];
#pragma unroll
            for (int j = 0; j < CTP; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][j], p[s], acc[j], 0, 0, 0);
        }
    }

    float* pt = partial + ((int64_t)wg * passes + pass) * CTP * 1024;
#pragma unroll
    for (int j = 0; j < CTP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            pt[j * 1024 + i * 32 + l31] = acc[j][r];
        }
    if (cpass == 0) {
        const float other = __shfl_xor(bsum, 32, 64);
        if (half == 0) partial_b[((int64_t)wg * numFT + ft) * 32 + l31] = bsum + other;
    }
}

// ---- stage 1, column panels, operands transposed through LDS.  In panel layout a node's 4 consecutive columns are 16 contiguous
// bytes, but the MFMA wants lane = column: the direct kernel above therefore issues, per 8-row round and tile, four 4-byte
// loads that each touch 8 x 32-byte segments (TCP-bound: 0.53 ms at config 2 against 0.19 ms of MFMA time).  Here a wave
// loads a tile of 8 nodes x 8 panels with ONE coalesced 16-byte load per lane (128 contiguous bytes per panel), writes it to
// a wave-private LDS tile [panel][node][4] with the panel stride padded to 36 dwords, and reads the operand of step s as
// LDS[q*36 + (2s+half)*4 + g%4] -- bank (4q + g%4 + const) % 32: conflict-free.  The next round's global loads are in
// flight during the 4 x CTP MFMAs of the current one.  Same strips, accumulators and partial layout as above.
constexpr int kTileDw = 8 * 36;  // dwords per staged tile

template <int CTP>
__global__ __launch_bounds__(kThreads) void grad_taps_panel_lds_kernel(const float* __restrict__ Zp, const float* __restrict__ P0p,
                                                                       float* __restrict__ partial, float* __restrict__ partial_b,
                                                                       int R, int N, int B, int G, int F, int numGI, int numCT,
                                                                       int numFT, int passes, int rowsPerWave) {
    __shared__ __attribute__((aligned(16))) float s_tiles[kWaves][(CTP + 1) * kTileDw];
    const int pass = blockIdx.y;
    const int ft = pass % numFT, cpass = pass / numFT;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wg = blockIdx.x * kWaves + wave;
    const int r_begin = wg * rowsPerWave;  // R < 2^31 (checked by the launcher)
    const int r_end = min(R, r_begin + rowsPerWave);
    const int QG = G / 4, QF = F / 4;
    const int64_t N4 = (int64_t)N * 4;
    float* tiles = s_tiles[wave];

    // load role: lane = (panel q of the tile, node j of the round)
    const int lq = lane >> 3, lj = lane & 7;
    const float* lbase[CTP + 1];  // tile j < CTP: Z tap/column tile; tile CTP: the P0 tile
    bool lvalid[CTP + 1];
    int lQ[CTP + 1];              // panels per batch entry of that operand
#pragma unroll
    for (int j = 0; j < CTP; ++j) {
        const int ct = cpass * CTP + j;
        const int t = ct / numGI, gi = ct - t * numGI;
        const int q = gi * 8 + lq;
        lvalid[j] = (ct < numCT) && (q < QG);
        lbase[j] = Zp + (lvalid[j] ? ((int64_t)t * B * QG + q) * N4 : 0);
        lQ[j] = QG;
    }
    {
        const int q = ft * 8 + lq;
        lvalid[CTP] = q < QF;
        lbase[CTP] = P0p + (lvalid[CTP] ? (int64_t)q * N4 : 0);
        lQ[CTP] = QF;
    }
    // MFMA role: lane = column (g for A tiles, f for the B tile), half = node parity
    const int rd = (l31 >> 2) * 36 + (l31 & 3) + half * 4;  // + 8*s per step

    f32x16 acc[CTP];
#pragma unroll
    for (int j = 0; j < CTP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;

    float4 cur[CTP + 1], nxt[CTP + 1];
    auto issue = [&](int r0, float4 (&dst)[CTP + 1]) {
        const int r = r0 + lj;
        const bool rv = r < r_end;
        const int b = rv ? r / N : 0;
        const int n = rv ? r - b * N : 0;
#pragma unroll
        for (int j = 0; j <= CTP; ++j)
            dst[j] = (rv && lvalid[j]) ? *reinterpret_cast<const float4*>(lbase[j] + (int64_t)b * lQ[j] * N4 + (int64_t)n * 4)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    issue(r_begin, cur);
    for (int r0 = r_begin; r0 < r_begin + rowsPerWave; r0 += 8) {
#pragma unroll
        for (int j = 0; j <= CTP; ++j) *reinterpret_cast<float4*>(tiles + j * kTileDw + lq * 36 + lj * 4) = cur[j];
        if (r0 + 8 < r_begin + rowsPerWave) issue(r0 + 8, nxt);  // in flight during this round's MFMAs
        float p[4], a[4][CTP];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            p[s] = tiles[CTP * kTileDw + rd + 8 * s];
#pragma unroll
            for (int j = 0; j < CTP; ++j) a[s][j] = tiles[j * kTileDw + rd + 8 * s];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bsum += p[s];
#pragma unroll
            for (int j = 0; j < CTP; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][j], p[s], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j <= CTP; ++j) cur[j] = nxt[j];
    }

    float* pt = partial + ((int64_t)wg * passes + pass) * CTP * 1024;
#pragma unroll
    for (int j = 0; j < CTP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            pt[j * 1024 + i * 32 + l31] = acc[j][r];
        }
    if (cpass == 0) {
        const float other = __shfl_xor(bsum, 32, 64);
        if (half == 0) partial_b[((int64_t)wg * numFT + ft) * 32 + l31] = bsum + other;
    }
}

template <int CTP>
void launch_stage1_panel(const Geo& g, const float* Zp, const float* P0p, float* ws, int N, int B, int G, int F, hipStream_t st) {
    if (g_tune.gradw_lds)
        hipLaunchKernelGGL((grad_taps_panel_lds_kernel<CTP>), dim3(g.strips, g.passes), dim3(kThreads), 0, st, Zp, P0p, ws,
                           ws + g.off_partial_b, (int)g.R, N, B, G, F, g.numGI, g.numCT, g.numFT, g.passes, g.rowsPerWave);
    else
        hipLaunchKernelGGL((grad_taps_panel_kernel<CTP>), dim3(g.strips, g.passes), dim3(kThreads), 0, st, Zp, P0p, ws,
                           ws + g.off_partial_b, (int)g.R, N, B, G, F, g.numGI, g.numCT, g.numFT, g.passes, g.rowsPerWave);
}

// ---- backward of the panel pipeline in one pass over the adjoint tap stack -----------------------------------------------------
// With P_t = (adjoint hop)^t dY (the stack the data path already builds) the tap gradient is  dh_t[g][f] = sum_rows X0[row][g] *
// P_t[row][f]  (= Z_t^T dY, moved through the adjoint), so BOTH outputs of the backward read the same P tiles:
//     dx[row][g]   = sum_{t,f} P_t[row][f] * h[f, t, g]        (the transposed-bank contraction of gf_contract.hip)
//     dh_t[g][f]  += X0[row][g] * P_t[row][f]                   (a reduction over rows)
// One wave owns a strip of rows (the Geo of the tap-gradient kernels) and walks it in 32-row tiles: per tap it loads the P tile once
// (16 bytes per lane, the contraction's B-operand layout: lane = row, 4 consecutive f), feeds the contraction MFMAs from registers,
// drops the tile into a wave-private LDS buffer [row][f] and reads it back with lane = f for the reduction MFMAs against the X0
// tile staged the same way.  HBM: the P stack + X0 once, dx once -- the separate tap-gradient kernel re-read the whole
// forward stack Z (T*R*G*4 bytes).  G, F <= 32 (one 32x32 accumulator tile per tap), T <= 6 (the dispatcher's limit).
// NM = 1: the same pass over NODE-MAJOR stacks P[T][B][N][F], X0[B][N][G] (the pipeline of graphs beyond the LDS panel limit): a lane
// still holds 4 consecutive f (g) of its row -- its four 16-byte loads per tap then come from one 128-byte row instead of four
// panels, 32 bytes contiguous per pair of lanes; everything after the loads is identical.
// F = 64 (FIN8 = 8; round 4: the 8 -> 64 first layers of config 3's class): the P tile is 64 columns wide, the contraction walks all 8
// column groups and the tap gradients are kept for BOTH 32-wide f blocks (T x 2 accumulator tiles: 160 registers at T = 5 -- one
// workgroup per CU, accumulators in AGPRs); G <= 32 only (a second g block would re-read the 64-wide P tiles).
template <int T, int GIN8, int FIN8, int NM>
__global__ __launch_bounds__(kThreads, (FIN8 > 4 ? 1 : 2)) void bwd_fused_panel_kernel(const float* __restrict__ Pp, const float* __restrict__ X0p,
                                                                      const float* __restrict__ h, float* __restrict__ dx,
                                                                      float* __restrict__ partial, float* __restrict__ partial_b, int R,
                                                                      int N, int Nout, int B, int E, int K, int rowsPerWave, int dx_panels,
                                                                      const float* __restrict__ maskp, int Gtot, int numGI, int passes, int ctp) {
    // dx_panels = 1 (layer-to-layer hand-over): dx goes out in the layout of the stacks -- column panels dx[b * G/4 + g/4][n][g % 4], or
    // (NM) node-major rows dx[b][n][g] -- masked by maskp (nullable; the activation of the layer the gradient is handed to, same layout:
    // entries <= 0 give 0) -- see contract_panel_kernel / contract_mfma_kernel
    //
    // LDS operands are laid out for 16-BYTE READS: every MFMA of this kernel takes an A (or B) operand out of LDS, and with one
    // ds_read_b32 per MFMA issued one or two MFMAs ahead the 64-cycle MFMAs of a wave waited for the LDS round trip every second
    // instruction (58-65 % MFMA busy at two waves per SIMD, profiles/r03_d_final).  The bank is stored [t][g][f] (a lane reads the four
    // f of four consecutive contraction MFMAs at once), the X0 and P tiles are stored TRANSPOSED [column][row order], rows in the order
    // the reduction MFMAs consume them (row 2s + half at position 16 half + s): four reads give a lane all 16 operands of a tap.
    // Gtot > 32 (64, 128: the widths of config 3's layers): blockIdx.y = one 32-wide block of the input features g -- its own X0 columns,
    // bank rows, dx columns and tap-gradient tiles; the P tiles are read once per block.
    constexpr int G = GIN8 * 8, F = FIN8 * 8, QF = F / 4;
    constexpr int FB = FIN8 > 4 ? FIN8 / 4 : 1;   // 32-wide f blocks of the tap gradients
    const int gblk = blockIdx.y, g0 = gblk * 32, QG = Gtot / 4, q0 = gblk * 8;
    constexpr int TS = 36;      // row stride of the transposed tiles (floats): 16-byte aligned, 36/4 odd -> conflict-free 16-byte reads
    constexpr int FS = F + 4;   // row stride of the bank (same rule: (F + 4) / 4 is odd for F = 8, 16, 32)
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    float* s_w = s_dyn;                                   // Hm[t][g][f], g padded to 32: T * 32 * FS floats
    constexpr int WT = (1 + FB) * 32 * TS;               // per wave: X0 tile [32 columns][TS], P tile [32 FB columns][TS] (transposed)
    float* s_tiles = s_dyn + T * 32 * FS;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < T * 32 * F; idx += kThreads) {
        const int f = idx % F, g = (idx / F) & 31, t = idx / (F * 32);
        float v = 0.f;
        if (g < G) {
            if (t == 0) {
                for (int e = 0; e < E; ++e) v += h[((int64_t)(f * E + e) * K) * Gtot + g0 + g];   // tap 0 is shared by the edge features
            } else {
                const int e = (t - 1) / (K - 1), k = (t - 1) % (K - 1) + 1;
                v = h[((int64_t)(f * E + e) * K + k) * Gtot + g0 + g];
            }
        }
        s_w[(t * 32 + g) * FS + f] = v;
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wg = blockIdx.x * kWaves + wave;
    const int r_begin = wg * rowsPerWave, r_end = min(R, r_begin + rowsPerWave);
    const int64_t N4 = (int64_t)N * 4;
    const int64_t tapStride = (int64_t)B * QF * N4;     // floats per tap (either layout: B * N * F)
    const int64_t pstep = NM ? 8 : 2 * N4;              // from one 8-column group of a row to the next: +8 floats | +2 panels
    const int64_t xstep = NM ? 8 : 2 * N4;
    float* xs = s_tiles + wave * WT;
    float* ps = xs + 32 * TS;
    // write side of a tile: this lane holds row l31, columns 8u + 4 half + (0..3); read side: column l31, rows 2s + half, s = 0..15
    const int wpos = (l31 & 1) * 16 + (l31 >> 1) + 4 * half * TS;   // + (8u + j) * TS
    const int rpos = l31 * TS + 16 * half;                          // 16 consecutive floats
    auto put_tile = [&](float* tile, const float4& v, int u) {
        float* q = tile + wpos + 8 * u * TS;
        q[0] = v.x;
        q[TS] = v.y;
        q[2 * TS] = v.z;
        q[3 * TS] = v.w;
    };

    f32x16 acc_h[T * FB];
#pragma unroll
    for (int t = 0; t < T * FB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_h[t][r] = 0.f;
    float bsum[FB];
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) bsum[fb] = 0.f;

    // ---- the operand stream of a strip: per tile the X0 tile and the T tap tiles, element e of tile i in ring slot (i (T + 1) + e) % 3.
    // TWO elements are in flight while one is consumed, across tile boundaries (one tap ahead left 32 KB per CU on the way: at the
    // loaded HBM latency that is 4.3 TB/s, and every tap ended in a wait).  The slot numbers are compile-time when 3 divides T + 1
    // (T = 5: the K = 5 filters of the configurations; T = 2); the other tap counts keep the one-ahead schedule below.
    // (Node-major stacks only: on column panels the same schedule measured 2 % slower than one-ahead, tools/ab_same_box.sh.)
    constexpr bool kDeep = (T + 1) % 3 == 0 && NM == 1 && FB == 1;
    constexpr int MAXW = GIN8 > FIN8 ? GIN8 : FIN8;
    if constexpr (kDeep) {
        f32x4 slot[3][MAXW];
        const float one = (float)(R > 0);   // 1.0f the compiler cannot fold
        int b = 0, n = 0, nb = 0, nn = 0;
        bool rv = false, nrv = false;
        const float *pb = Pp, *xb = X0p, *npb = Pp, *nxb = X0p;
        // every load is UNCONDITIONAL (rows past the strip's end re-read its last row and are zeroed when the tile is consumed): a load
        // behind a lane-validity branch made the compiler wait with vmcnt(0) instead of counting the younger loads
#define GF_TILE_ADDR(R0, RV, BB, NN, PB, XB)                                                                                 \
        do {                                                                                                                 \
            RV = (R0) + l31 < r_end;                                                                                         \
            const int r_ = min((R0) + l31, r_end - 1);                                                                       \
            BB = r_ / N;                                                                                                     \
            NN = r_ - BB * N;                                                                                                \
            PB = NM ? Pp + (int64_t)r_ * F + 4 * half : Pp + ((int64_t)BB * QF + half) * N4 + (int64_t)NN * 4;               \
            XB = NM ? X0p + (int64_t)r_ * Gtot + g0 + 4 * half : X0p + ((int64_t)BB * QG + q0 + half) * N4 + (int64_t)NN * 4; \
        } while (0)
#define GF_LOAD_X0(S, XB)                                                                                                    \
        _Pragma("unroll") for (int u = 0; u < GIN8; ++u) slot[S][u] = *reinterpret_cast<const f32x4*>(XB + (int64_t)u * xstep)
#define GF_LOAD_P(S, PB, TT)                                                                                                 \
        _Pragma("unroll") for (int u = 0; u < FIN8; ++u)                                                                     \
            slot[S][u] = *reinterpret_cast<const f32x4*>(PB + (int64_t)(TT) * tapStride + (int64_t)u * pstep)
        auto put_tile4 = [&](float* tile, const f32x4& v, int u) {
            float* q = tile + wpos + 8 * u * TS;
            q[0] = v[0];
            q[TS] = v[1];
            q[2 * TS] = v[2];
            q[3 * TS] = v[3];
        };
        if (r_begin < r_end) {
            GF_TILE_ADDR(r_begin, rv, b, n, pb, xb);
            GF_LOAD_X0(0, xb);
            GF_LOAD_P(1, pb, 0);
        }
        for (int r0 = r_begin; r0 < r_end; r0 += 32) {
            f32x16 acc_x;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_x[i] = 0.f;
#pragma unroll
            for (int e = 0; e <= T; ++e) {
                // request element e + 2 (its slot held element e - 1)
                if (e + 2 <= T) {
                    GF_LOAD_P((e + 2) % 3, pb, e + 1);
                } else if (e + 2 == T + 1) {   // X0 of the next tile (past the strip's end: its last row again, never consumed)
                    GF_TILE_ADDR(r0 + 32, nrv, nb, nn, npb, nxb);
                    GF_LOAD_X0(0, nxb);
                } else {
                    GF_LOAD_P(1, npb, 0);
                }
                __builtin_amdgcn_sched_barrier(0);   // the requests stay HERE: the scheduler otherwise sinks them to their first use, two elements later
                if (e == 0) {
#pragma unroll
                    for (int u = 0; u < GIN8; ++u) put_tile4(xs, rv ? slot[0][u] : f32x4{0.f, 0.f, 0.f, 0.f}, u);
                    continue;
                }
                const int t = e - 1;
                f32x4 cur[FIN8];
#pragma unroll
                for (int u = 0; u < FIN8; ++u) cur[u] = rv ? slot[e % 3][u] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < FIN8; ++u) put_tile4(ps, cur[u], u);
                const float* wt = s_w + (t * 32 + l31) * FS + 4 * half;
#pragma unroll
                for (int u = 0; u < FIN8; ++u) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + 8 * u);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc_x = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[s], cur[u][s], acc_x, 0, 0, 0);
                }
                if (t == T - 1) {   // dx of this tile is complete: stored ahead of the last 16 tap-gradient MFMAs, through inline asm
                                    // (see store_f32_hidden in gf_contract.hip: the compiler then keeps exact vmcnt counts for the ring).
                                    // Every value passes through a VALU instruction the compiler sees (* one, exact): the wait
                                    // states between an MFMA and a store reading its result are only inserted for instructions
                                    // the hazard recogniser knows, not for inline asm.
                    if (rv && dx_panels) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int q = 2 * j + half;   // g = 4q .. 4q + 3
                            if (q < G / 4) {
                                const int64_t at = NM ? ((int64_t)b * N + n) * Gtot + g0 + 4 * q : (((int64_t)b * QG + q0 + q) * N + n) * 4;
                                f32x4 v = {acc_x[4 * j] * one, acc_x[4 * j + 1] * one, acc_x[4 * j + 2] * one, acc_x[4 * j + 3] * one};
                                if (maskp) {
                                    const float4 m = *reinterpret_cast<const float4*>(maskp + at);
                                    v[0] = m.x > 0.f ? v[0] : 0.f;
                                    v[1] = m.y > 0.f ? v[1] : 0.f;
                                    v[2] = m.z > 0.f ? v[2] : 0.f;
                                    v[3] = m.w > 0.f ? v[3] : 0.f;
                                }
                                asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dx + at), "v"(v) : "memory");   // (s_nop: > 8-byte store data hazard, see gf_contract.hip)
                            }
                        }
                    } else if (rv && n < Nout) {
                        float* ob = dx + ((int64_t)b * Gtot + g0) * Nout + n;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int g = (i & 3) + 8 * (i >> 2) + 4 * half;
                            const float v = acc_x[i] * one;
                            if (g < G) asm volatile("global_store_dword %0, %1, off" ::"v"(ob + (int64_t)g * Nout), "v"(v) : "memory");
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(xs + rpos + 4 * q);
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(ps + rpos + 4 * q);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (t == 0) bsum[0] += pv[s];
                        acc_h[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], pv[s], acc_h[t], 0, 0, 0);
                    }
                }
            }
            rv = nrv, b = nb, n = nn, pb = npb, xb = nxb;
        }
#undef GF_TILE_ADDR
#undef GF_LOAD_X0
#undef GF_LOAD_P
    } else
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
        const int r = r0 + l31;
        const bool rv = r < r_end;
        const int b = rv ? r / N : 0;
        const int n = rv ? r - b * N : 0;
        const float* pb = NM ? Pp + (int64_t)r * F + 4 * half                     // + 8u floats, + t taps
                             : Pp + ((int64_t)b * QF + half) * N4 + (int64_t)n * 4;   // + (2u) panels, + t taps
        const float* xb = NM ? X0p + (int64_t)r * Gtot + g0 + 4 * half : X0p + ((int64_t)b * QG + q0 + half) * N4 + (int64_t)n * 4;
        float4 x0[GIN8], cur[FIN8], nxt[FIN8];
#pragma unroll
        for (int u = 0; u < GIN8; ++u)
            x0[u] = rv ? *reinterpret_cast<const float4*>(xb + (int64_t)u * xstep) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < FIN8; ++u)
            cur[u] = rv ? *reinterpret_cast<const float4*>(pb + (int64_t)u * pstep) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < GIN8; ++u) put_tile(xs, x0[u], u);

        f32x16 acc_x;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_x[i] = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T) {
#pragma unroll
                for (int u = 0; u < FIN8; ++u)
                    nxt[u] = rv ? *reinterpret_cast<const float4*>(pb + (int64_t)(t + 1) * tapStride + (int64_t)u * pstep)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // P tile -> LDS (transposed) for the reduction (the previous tap's reads are complete: same wave, program order)
#pragma unroll
            for (int u = 0; u < FIN8; ++u) put_tile(ps, cur[u], u);
            // data path: acc_x[g][row] += Hm[t][g][f] * P_t[row][f]
            const float* wt = s_w + (t * 32 + l31) * FS + 4 * half;
#pragma unroll
            for (int u = 0; u < FIN8; ++u) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + 8 * u);
                const float bs[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
#pragma unroll
                for (int s = 0; s < 4; ++s) acc_x = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[s], bs[s], acc_x, 0, 0, 0);
            }
            // tap gradient: acc_h[t][g][f] += X0[row][g] * P_t[row][f], two rows per MFMA (k = half)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(xs + rpos + 4 * q);
#pragma unroll
                for (int fb = 0; fb < FB; ++fb) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(ps + fb * 32 * TS + rpos + 4 * q);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (t == 0) bsum[fb] += pv[s];
                        acc_h[t * FB + fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], pv[s], acc_h[t * FB + fb], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < FIN8; ++u) cur[u] = nxt[u];
        }
        if (rv && dx_panels) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 2 * j + half;   // g = 4q .. 4q + 3
                if (q < G / 4) {
                    const int64_t at = NM ? ((int64_t)b * N + n) * Gtot + g0 + 4 * q : (((int64_t)b * QG + q0 + q) * N + n) * 4;
                    float4 v = make_float4(acc_x[4 * j], acc_x[4 * j + 1], acc_x[4 * j + 2], acc_x[4 * j + 3]);
                    if (maskp) {
                        const float4 m = *reinterpret_cast<const float4*>(maskp + at);
                        v.x = m.x > 0.f ? v.x : 0.f;
                        v.y = m.y > 0.f ? v.y : 0.f;
                        v.z = m.z > 0.f ? v.z : 0.f;
                        v.w = m.w > 0.f ? v.w : 0.f;
                    }
                    *reinterpret_cast<float4*>(dx + at) = v;
                }
            }
        } else if (rv && n < Nout) {
            float* ob = dx + ((int64_t)b * Gtot + g0) * Nout + n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int g = (i & 3) + 8 * (i >> 2) + 4 * half;
                if (g < G) ob[(int64_t)g * Nout] = acc_x[i];
            }
        }
    }

    // tap-gradient tile (t, gblk) is tile ct = t * numGI + gblk of the shared partial layout [wave][pass][ctp] (make_geo; one pass when Gtot <= 32)
    // The four waves of the workgroup add their tiles up through LDS, in wave order (fixed: deterministic), and the workgroup writes ONE
    // set of partial tiles: row blockIdx.x of the partial matrix (round 3 wrote one per wave: at config 3 the 2048 x 40 KB of partials
    // were 84 MB written here and read back by reduce_rows_kernel, 29 us per call, four calls per step).  Tile by tile: every wave
    // drops its tile into its slot, one barrier, the 256 threads add the four slots for four elements each and store them coalesced;
    // two sets of slots alternate, so the next tile is dropped while stragglers still read.  The wave tiles are dead by now: their LDS
    // (4 x (1 + FB) x 32 x 36 >= 9216 floats) holds the 2 x 4 slots of 1024 and the bias sums.
    // (pass = c-pass * numFT + f block, as grad_taps_kernel numbers them; FB = 2 runs with numGI = 1: one c-pass)
    static_assert(kWaves * WT >= 2 * kWaves * 1024 + kWaves * FB * 32, "the wave tiles hold the slots of the workgroup reduction");
    float* slots = s_tiles;
    float* bslot = s_tiles + 2 * kWaves * 1024;
    __syncthreads();   // every wave is done with its tiles
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
        const float both = bsum[fb] + __shfl_xor(bsum[fb], 32, 64);   // even rows + odd rows
        if (half == 0) bslot[(wave * FB + fb) * 32 + l31] = both;
    }
#pragma unroll
    for (int k = 0; k < T * FB; ++k) {
        float* set = slots + (k & 1) * kWaves * 1024;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int g = (i & 3) + 8 * (i >> 2) + 4 * half;
            set[wave * 1024 + g * 32 + l31] = acc_h[k][i];
        }
        __syncthreads();
        const int t = k / FB, fb = k - t * FB;
        const int ct = t * numGI + gblk;
        float* pt = partial + (((int64_t)blockIdx.x * passes + (ct / ctp) * FB + fb) * ctp + ct % ctp) * 1024;
#pragma unroll
        for (int j = 0; j < 1024 / kThreads; ++j) {
            const int e = tid + j * kThreads;
            pt[e] = ((set[e] + set[1024 + e]) + set[2048 + e]) + set[3072 + e];
        }
    }
    if (gblk == 0 && tid < FB * 32)   // (the bias slots were complete at the first barrier of the loop)
        partial_b[(int64_t)blockIdx.x * FB * 32 + tid] = ((bslot[tid] + bslot[FB * 32 + tid]) + bslot[2 * FB * 32 + tid]) + bslot[3 * FB * 32 + tid];
}

// per_wg: the stage-1 kernel wrote one row of partials per WORKGROUP (bwd_fused_panel_kernel) instead of one per wave
int finish_taps(const Geo& g, float* ws, float* dh, float* dbias, int G, int F, int E, int K, hipStream_t st, bool per_wg = false) {
    const int wavesPadded = per_wg ? g.strips : g.strips * kWaves;
    const int slices = wavesPadded < kSlices ? wavesPadded : kSlices;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((g.tileOutputs + kThreads - 1) / kThreads), slices), dim3(kThreads), 0,
                       st, ws, ws + g.off_mid, wavesPadded, g.tileOutputs, slices);
    GF_LAUNCH_CHECK("reduce_rows_kernel(taps)");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((g.biasOutputs + kThreads - 1) / kThreads), slices), dim3(kThreads), 0,
                       st, ws + g.off_partial_b, ws + g.off_mid_b, wavesPadded, g.biasOutputs, slices);
    GF_LAUNCH_CHECK("reduce_rows_kernel(bias)");
    const int64_t tot = g.tileOutputs + g.biasOutputs;
    hipLaunchKernelGGL(finalize_taps_kernel, dim3((unsigned)((tot + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       ws + g.off_mid, ws + g.off_mid_b, dh, dbias, g.tileOutputs, (int)g.biasOutputs, slices, G, F, E, K,
                       g.numGI, g.numCT, g.numFT, g.ctp);
    GF_LAUNCH_CHECK("finalize_taps_kernel");
    return GF_OK;
}

}  // namespace

// dx and (dh, dbias) from the adjoint tap stack in one kernel; false if the shape is not covered (the caller then runs the
// contraction and the tap-gradient kernel separately)
bool gf_bwd_fused_supported(int G, int F, int E, int K) {
    const int T = gf_num_taps(E, K);
    auto ok = [](int w) { return w == 8 || w == 16 || w == 32; };
    // G = 64, 128: one 32-wide block of g per blockIdx.y.  F = 64: the tap gradients of two f blocks in one wave's registers (G <= 32).
    if (F == 64) return ok(G) && T >= 1 && T <= 6 && g_tune.bwd_fuse64 != 0;
    return (ok(G) || G == 64 || G == 128) && ok(F) && T >= 1 && T <= 6;  // 16 accumulator registers per tap: T = 7, 8 need > 256 VGPRs
}

int gf_bwd_fused_panel_launch(const float* Pp, const float* X0p, const float* h, float* dx, float* dh, float* dbias, void* workspace,
                              size_t workspace_bytes, int B, int N, int Nout, int G, int F, int E, int K, hipStream_t st, int node_major,
                              int dx_panels, const float* maskp) {
    const Geo g = make_geo(B, N, G, F, E, K);
    GF_REQUIRE_SHAPE(!dx_panels || Nout == N, "gf_lsigf_backward: the hand-over of dx in the internal layout needs Nin == N");
    GF_REQUIRE_SHAPE(g.R < (int64_t)INT32_MAX - 4096, "gf_lsigf_backward: B*N = %lld too large", (long long)g.R);
    GF_REQUIRE_ARG(workspace && workspace_bytes >= g.bytes, "gf_lsigf_backward: workspace %zu bytes < required %zu", workspace_bytes, g.bytes);
    GF_REQUIRE_SHAPE((g.numFT == 1 && (G > 32 || (g.passes == 1 && g.ctp == g.T))) || (F == 64 && G <= 32 && g.passes == 2 && g.ctp == g.T),
                     "gf_lsigf_backward: fused backward geometry");
    float* ws = (float*)workspace;
    const size_t lds = ((size_t)g.T * 32 * (F + 4) + (size_t)kWaves * (1 + (F > 32 ? F / 32 : 1)) * 32 * 36) * sizeof(float);  // bank [t][g][F + 4] + wave tiles (TS = 36)
    hipError_t attr = hipSuccess;
#define GF_BF(TT, GG, FF)                                                                                                      \
    do {                                                                                                                        \
        auto kern = node_major ? bwd_fused_panel_kernel<TT, GG, FF, 1> : bwd_fused_panel_kernel<TT, GG, FF, 0>;                 \
        if (lds > 64 * 1024) attr = gf_grant_lds((const void*)kern, lds);                                                       \
        if (attr == hipSuccess)   /* (a refused LDS grant is reported below, not turned into an invalid launch) */               \
            hipLaunchKernelGGL(kern, dim3(g.strips, g.numGI), dim3(kThreads), lds, st, Pp, X0p, h, dx, ws, ws + g.off_partial_b, (int)g.R, N, \
                               Nout, B, E, K, g.rowsPerWave, dx_panels, maskp, G, g.numGI, g.passes, g.ctp);                    \
    } while (0)
#define GF_BF64(TT, GG)                                                                                                         \
    do {                                                                                                                        \
        if constexpr ((TT) <= 6) {                                                                                              \
            GF_BF((TT) <= 6 ? (TT) : 1, GG, 8);                                                                                 \
        }                                                                                                                       \
    } while (0)
#define GF_BF_F(TT, GG)                                                                                                         \
    switch (F / 8) {                                                                                                            \
        case 1: GF_BF(TT, GG, 1); break;                                                                                        \
        case 2: GF_BF(TT, GG, 2); break;                                                                                        \
        case 8: GF_BF64(TT, GG); break;                                                                                         \
        default: GF_BF(TT, GG, 4); break;                                                                                       \
    }
#define GF_BF_G(TT)                                                                                                             \
    switch (G >= 32 ? 4 : G / 8) {                                                                                              \
        case 1: GF_BF_F(TT, 1); break;                                                                                          \
        case 2: GF_BF_F(TT, 2); break;                                                                                          \
        default: GF_BF_F(TT, 4); break;                                                                                         \
    }
    switch (g.T) {
        case 1: GF_BF_G(1); break;
        case 2: GF_BF_G(2); break;
        case 3: GF_BF_G(3); break;
        case 4: GF_BF_G(4); break;
        case 5: GF_BF_G(5); break;
        case 6: GF_BF_G(6); break;
        case 7: GF_BF_G(7); break;
        default: GF_BF_G(8); break;
    }
    GF_HIP(attr);
#undef GF_BF_G
#undef GF_BF_F
#undef GF_BF64
#undef GF_BF
    GF_LAUNCH_CHECK("bwd_fused_panel_kernel");
    return finish_taps(g, ws, dh, dbias, G, F, E, K, st, /*per_wg=*/true);
}

extern "C" int gf_grad_taps_panel(const float* Zp, const float* P0p, float* dh, float* dbias, void* workspace, size_t workspace_bytes,
                                  int32_t B, int32_t N, int32_t G, int32_t F, int32_t E, int32_t K, void* stream) {
    GF_REQUIRE_ARG(Zp && P0p && workspace, "gf_grad_taps_panel: NULL tensor");
    GF_REQUIRE_ARG(dh || dbias, "gf_grad_taps_panel: nothing to compute (dh and dbias both NULL)");
    GF_REQUIRE_SHAPE(B > 0 && N > 0 && G > 0 && F > 0 && E > 0 && K > 0 && G % 4 == 0 && F % 4 == 0,
                     "gf_grad_taps_panel: bad shape B=%d N=%d G=%d F=%d E=%d K=%d (G, F multiples of 4)", B, N, G, F, E, K);
    const Geo g = make_geo(B, N, G, F, E, K);
    GF_REQUIRE_SHAPE(g.R < (int64_t)INT32_MAX - 4096, "gf_grad_taps_panel: B*N = %lld too large", (long long)g.R);
    GF_REQUIRE_ARG(workspace_bytes >= g.bytes, "gf_grad_taps_panel: workspace %zu bytes < required %zu", workspace_bytes, g.bytes);
    GF_REQUIRE_SHAPE(g.passes <= 65535, "gf_grad_taps_panel: %d passes exceed the grid limit", g.passes);
    hipStream_t st = gf_stream(stream);
    float* ws = (float*)workspace;
    switch (g.ctp) {
        case 1: launch_stage1_panel<1>(g, Zp, P0p, ws, N, B, G, F, st); break;
        case 2: launch_stage1_panel<2>(g, Zp, P0p, ws, N, B, G, F, st); break;
        case 3: launch_stage1_panel<3>(g, Zp, P0p, ws, N, B, G, F, st); break;
        case 4: launch_stage1_panel<4>(g, Zp, P0p, ws, N, B, G, F, st); break;
        case 5: launch_stage1_panel<5>(g, Zp, P0p, ws, N, B, G, F, st); break;
        case 6: launch_stage1_panel<6>(g, Zp, P0p, ws, N, B, G, F, st); break;
        case 7: launch_stage1_panel<7>(g, Zp, P0p, ws, N, B, G, F, st); break;
        default: launch_stage1_panel<8>(g, Zp, P0p, ws, N, B, G, F, st); break;
    }
    GF_LAUNCH_CHECK("grad_taps_panel_kernel");
    return finish_taps(g, ws, dh, dbias, G, F, E, K, st);
}

namespace {
}  // namespace

extern "C" size_t gf_grad_taps_workspace_bytes(int32_t B, int32_t N, int32_t G, int32_t F, int32_t E, int32_t K) {
    if (B <= 0 || N <= 0 || G <= 0 || F <= 0 || E <= 0 || K <= 0) return 0;
    return make_geo(B, N, G, F, E, K).bytes;
}

extern "C" int gf_grad_taps(const float* Z, const float* P0, float* dh, float* dbias, void* workspace, size_t workspace_bytes,
                            int32_t B, int32_t N, int32_t G, int32_t F, int32_t E, int32_t K, void* stream) {
    GF_REQUIRE_ARG(Z && P0 && workspace, "gf_grad_taps: NULL tensor");
    GF_REQUIRE_ARG(dh || dbias, "gf_grad_taps: nothing to compute (dh and dbias both NULL)");
    GF_REQUIRE_SHAPE(B > 0 && N > 0 && G > 0 && F > 0 && E > 0 && K > 0, "gf_grad_taps: bad shape B=%d N=%d G=%d F=%d E=%d K=%d",
                     B, N, G, F, E, K);
    const Geo g = make_geo(B, N, G, F, E, K);
    GF_REQUIRE_ARG(workspace_bytes >= g.bytes, "gf_grad_taps: workspace %zu bytes < required %zu", workspace_bytes, g.bytes);
    GF_REQUIRE_SHAPE(g.passes <= 65535, "gf_grad_taps: %d passes exceed the grid limit", g.passes);
    hipStream_t st = gf_stream(stream);
    float* ws = (float*)workspace;
    switch (g.ctp) {
        case 1: launch_stage1<1>(g, Z, P0, ws, G, F, st); break;
        case 2: launch_stage1<2>(g, Z, P0, ws, G, F, st); break;
        case 3: launch_stage1<3>(g, Z, P0, ws, G, F, st); break;
        case 4: launch_stage1<4>(g, Z, P0, ws, G, F, st); break;
        case 5: launch_stage1<5>(g, Z, P0, ws, G, F, st); break;
        case 6: launch_stage1<6>(g, Z, P0, ws, G, F, st); break;
        case 7: launch_stage1<7>(g, Z, P0, ws, G, F, st); break;
        default: launch_stage1<8>(g, Z, P0, ws, G, F, st); break;
    }
    GF_LAUNCH_CHECK("grad_taps_kernel");
    return finish_taps(g, ws, dh, dbias, G, F, E, K, st);
}
