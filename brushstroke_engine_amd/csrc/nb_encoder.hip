// Geometry encoder of the painting engine for gfx950.
//
// Reference: forger/experimental/autoenc/simple_autoencoder.py:88-121, 155-199, 251-261 -- a stack of
//   conv (reflect padding) -> eval-mode BatchNorm -> LeakyReLU(0.01):
//   stem 1->64 7x7 | 64->128, 128->256, 256->256 3x3 stride 2 | 256->32, 32->16 3x3 | bilinear x2 + 16->256 3x3.
// BatchNorm is folded into the conv weights / bias on the host, so every layer is conv + bias + LeakyReLU.
//
// Three kernels:
//   enc_stem7x7_kernel   C_in = 1: no channel contraction, so the taps are the K dimension -- ordered column by column (K = 8 b + a,
//                        tap row a and column b padded from 7 to 8 with zero weights), which makes a lane's 8 K-values eight
//                        vertically adjacent pixels: each wave keeps, per output row, a column-major hi/lo f16 strip of its
//                        seven input rows in LDS and every B fragment is one aligned ds_read_b128.  Split-f16 products (3
//                        v_mfma_f32_32x32x16_f16 per K step, fp32-grade) instead of the exact-fp32 MFMA of rounds 1-2, which runs
//                        at a sixteenth of the rate and was half of the launch: 96 instead of 200 four-times-longer MFMAs per wave.
//   enc_conv3x3_h3_kernel  3x3, stride 1 or 2, split-f16 products (see nb_modconv_h3.hip for the arithmetic and
//                        the H2 activation format).  K loop = (16-channel chunk, tap row); per step the activation
//                        rows that tap row needs are gathered by LDS-DMA with per-lane source addresses -- for
//                        stride 2 the even and odd input columns land in two planes, so every MFMA B fragment is a
//                        stride-1 ds_read_b128 whatever the conv stride, and reflect padding is just address
//                        arithmetic in the gather.  Steps are double buffered (DMA of step t+1 under the MFMAs of t).
//   enc_upsample2x_h2_kernel  bilinear x2, align_corners=True, fp32 NCHW -> H2 (input of the decoder conv).
// Layer outputs go out as H2 (next layer's input) or fp32 NCHW (what the generator consumes).
#include "nb_common.h"
#include "nb_h3_common.h"
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define NB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

extern "C" const float* nb_zero_page_ptr(void);

__device__ __forceinline__ int nb_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ float nb_lrelu(float v, float slope) { return v < 0.f ? v * slope : v; }

// ------------------------------------------------------------------------------------------------
// stem: 1 -> 64 channels, 7x7, reflect padding 3; fp32 [n,1,h,w] -> H2 [n,8,2,h,w,8]
// ------------------------------------------------------------------------------------------------
// four values -> four e4m3 bytes, saturated to the format's range (the same conversion as nb_pk4_fp8 of nb_modconv_h3.hip)
__device__ __forceinline__ unsigned nb_enc_pk4_fp8(float a, float b, float c, float d) {
    auto cl = [](float v) { return fminf(fmaxf(v, -448.f), 448.f); };
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), w, true);
    return (unsigned)w;
}

// ... in a wave whose MODE.FP16_OVFL is set the conversion itself saturates (nb_pk4_fp8_sat of nb_modconv_h3.hip): the stem's
// epilogue is bound by its vector instructions, 4 of ~15 per value were these clamps
__device__ __forceinline__ unsigned nb_enc_pk4_fp8_sat(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ void nb_enc_set_fp16_ovfl() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }

struct StemParams {
    const float* x; const float* w50; const float* bias; _Float16* y;
    int h, w, tiles_x, preproc;
    float slope;
    int out_f8;             // lo planes in the "f8" operand format (fp8 correction operands) instead of f16 residuals
};

__global__ __launch_bounds__(256) void enc_stem7x7_kernel(const StemParams p) {
    constexpr int TR = 16, TC = 32, PR = TR + 6, PC = TC + 6, NBW = 4, KS = 4;
    constexpr int SC = 40;                                                                   // strip columns (38 + the zero-weight tap column 7)
    if (p.out_f8) nb_set_fp16_ovfl();                                                        // the fp8 (and f16) conversions saturate
    __shared__ float tile[PR * PC];
    __shared__ __attribute__((aligned(16))) float s_one[64], s_bias[64];                     // the epilogue's per-channel tables
    __shared__ __attribute__((aligned(16))) h8 wfrag[2][KS][2][64];                          // weight fragments [mb][K step][hi/lo][lane]
    __shared__ __attribute__((aligned(16))) h8 strip[4][NBW][2][SC];                          // per wave and output row: [hi/lo][column] x 8 input rows
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lh = lane >> 5, l31 = lane & 31;
    const int n = blockIdx.y;
    const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x - ty * p.tiles_x;
    const int y0 = ty * TR, x0 = tx * TC;
    const float* xn = p.x + (size_t)n * p.h * p.w;
    if (tid < 64) { s_one[tid] = 1.f; s_bias[tid] = p.bias[tid]; }
    for (int e = tid; e < PR * PC; e += 256) {
        const int r = e / PC, c = e - r * PC;
        float v = xn[(size_t)nb_reflect(y0 + r - 3, p.h) * p.w + nb_reflect(x0 + c - 3, p.w)];
        if (p.preproc == 1) v = (1.f - v) * 2.f - 1.f;          // '-11inverse' (autoenc/base.py:30-52)
        else if (p.preproc == 2) v = 1.f - v;                   // 'inverse'
        tile[e] = v;
    }
    // A fragments: lane (row = c_out l31, k group lh) of K step s holds w[c_out][tap rows 0..7][tap column 2s + lh] (row / column 7: zero), hi and lo.
    // All four waves need the same 16 fragments per lane: each wave builds a quarter (one (mb, s-pair): 16 loads + splits per lane instead
    // of 64) and they trade through LDS.
    {
        const int mb = wv >> 1, s0 = (wv & 1) * 2;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int b = 2 * (s0 + ss) + lh;
            h8 hi8, lo8;
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float wv_ = (a < 7 && b < 7) ? p.w50[(mb * 32 + l31) * 50 + a * 7 + b] : 0.f;
                const _Float16 hi = (_Float16)wv_;
                hi8[a] = hi; lo8[a] = (_Float16)(wv_ - (float)hi);
            }
            wfrag[mb][s0 + ss][0][lane] = hi8;
            wfrag[mb][s0 + ss][1][lane] = lo8;
        }
    }
    __syncthreads();
    h8 wah[2][KS], wal[2][KS];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int s = 0; s < KS; ++s) { wah[mb][s] = wfrag[mb][s][0][lane]; wal[mb][s] = wfrag[mb][s][1][lane]; }
    // this wave's strips: lane = column, eight input rows of output row (wv NBW + nb) as one 16-byte slot per half
    if (lane < SC) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            h8 hi, lo;
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float v = (a < 7 && lane < PC) ? tile[(wv * NBW + nb + a) * PC + lane] : 0.f;
                const _Float16 hh = (_Float16)v;
                hi[a] = hh; lo[a] = (_Float16)(v - (float)hh);
            }
            strip[wv][nb][0][lane] = hi;
            strip[wv][nb][1][lane] = lo;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): wave-private strips, LDS executes a wave's accesses in order
    __builtin_amdgcn_wave_barrier();

    f32x16 acc[2][NBW];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const h8 bh = strip[wv][nb][0][l31 + 2 * s + lh], bl = strip[wv][nb][1][l31 + 2 * s + lh];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wah[mb][s], bh, acc[mb][nb], 0, 0, 0);
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wah[mb][s], bl, acc[mb][nb], 0, 0, 0);
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wal[mb][s], bh, acc[mb][nb], 0, 0, 0);
            }
        }
    }
    // epilogue: bias, LeakyReLU, hi/lo split -- the generator's hand-off epilogue (nb_h3_common.h: straight from the accumulators, packed
    // arithmetic, lane halves trade channel groups, whole 16-byte slots) with demodulation 1, no noise, gain 1, no clamp, scale 1
    const float nz0[NBW] = {};
    nb_up1_handoff_epilogue<2, NBW>(nb_handoff_args(p.y, 8, 64, p.h, p.w, p.out_f8, 0, p.slope, 1.f, -1.f), acc, nz0, s_one, s_bias, s_one, 0, wv * NBW, 0, n, y0, x0, lh, l31);
}

extern "C" int nb_enc_stem7x7_f32_h2_ex(const float* x, const float* w50, const float* bias, void* y_h2, int out_fmt, int n, int h, int w,
                                        int preproc, float slope, void* stream);
extern "C" int nb_enc_stem7x7_f32_h2(const float* x, const float* w50, const float* bias, void* y_h2, int n, int h, int w,
                                     int preproc, float slope, void* stream) {
    return nb_enc_stem7x7_f32_h2_ex(x, w50, bias, y_h2, 0, n, h, w, preproc, slope, stream);
}
extern "C" int nb_enc_stem7x7_f32_h2_ex(const float* x, const float* w50, const float* bias, void* y_h2, int out_fmt, int n, int h, int w,
                                        int preproc, float slope, void* stream) {
    NB_REQUIRE(out_fmt == 0 || out_fmt == 1, "enc_stem7x7: output format must be 0 (H2) or 1 (f8)");
    NB_REQUIRE(x && w50 && bias && y_h2, "enc_stem7x7: null pointer");
    NB_REQUIRE(n >= 1 && n <= 65535 && h % 16 == 0 && w % 32 == 0 && h >= 16 && w >= 32,
               "enc_stem7x7: needs h %% 16 == 0 and w %% 32 == 0 (got %dx%d)", h, w);
    NB_REQUIRE(preproc >= 0 && preproc <= 2, "Unknown preprocessing type %d", preproc);
    NB_REQUIRE(slope >= 0.f && slope <= 1.f, "enc_stem7x7: leaky-ReLU slope must lie in [0, 1] (got %g)", slope);
    NB_REQUIRE((uintptr_t)y_h2 % 16 == 0, "enc_stem7x7: output must be 16-byte aligned");
    StemParams p{x, w50, bias, (_Float16*)y_h2, h, w, w / 32, preproc, slope, out_fmt};
    hipLaunchKernelGGL(enc_stem7x7_kernel, dim3((w / 32) * (h / 16), n), dim3(256), 0, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("enc_stem7x7");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// 3x3 conv, stride 1 or 2, reflect padding 1, split-f16
// ------------------------------------------------------------------------------------------------
struct EncConvParams {
    const _Float16* x;      // H2 [n][c8][2][hin][win][8]
    const _Float16* wts;    // [nchunks][3][3][2][2][co_ld][8], co_ld % 128 == 0
    const float* bias;      // [c_out]
    float* y32;             // fp32 NCHW [n][c_out][hout][wout]      (OUT == 0)
    _Float16* yh2;          // H2 [n][c_out/8][2][hout][wout][8]     (OUT == 1)
    const float* zeros;
    int c8, nchunks, c_out, co_ld, hin, win, hout, wout, tiles_x, tiles_y, slices;
    float slope;
    // hand-off into a CONSUMER's operand tensor (OUT == 1): the result times oscale[n][co] goes into channel groups
    // cg0 .. of a tensor with c8_total groups, in H2 (hi/lo f16) or "f8" (hi f16 + fp8 correction operands) format
    const float* oscale;
    int oscale_stride, c8_total, cg0, out_f8;
    // shift = 1 (stride 2 only): no padding -- the window of output (i, j) starts at input (2i, 2j) of a (2 hout + 1) x (2 wout + 1)
    // input, i.e. the padded form's taps moved one pixel down / right (the training path's stride-2 correlations)
    int shift;
    unsigned long long* tstamps;   // debug: per-workgroup phase timestamps [workgroup][8] (nb_debug_set_enc_timestamps), else null
    // FUSED (stem + first stride-2 stage in one kernel): the fp32 image [n][1][hin][win], the stem's weights [64][50] and bias [64], its
    // input preprocessing and LeakyReLU slope; x is unused
    const float* img; const float* stem_w; const float* stem_b;
    int preproc;
};

static unsigned long long* g_enc_tstamps = nullptr;
static int g_enc_tstamps_cap = 0;
// Debug hook (not part of the product ABI): phase timestamps of the next enc_conv3x3_h3 launches, s_memrealtime ticks (100 MHz):
// slot 0 start, 1 first step may begin, 2 K loop done, 3 epilogue values in LDS, 4 end
extern "C" void nb_debug_set_enc_timestamps(void* buf, int capacity_workgroups) { g_enc_tstamps = (unsigned long long*)buf; g_enc_tstamps_cap = capacity_workgroups; }

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));


// F8: the input activations and the weights carry their corrections as fp8 (the "f8" operand format of nb_modconv_h3.hip:
// lo slot of an even channel group = fp8(xl 2^9) of the 16-channel chunk, of the odd group = fp8(x/4); weights fp8(w) /
// fp8(wl 2^11)); per tap one f16 MFMA for the main product and, per PAIR of taps, one block-scaled K=64 fp8 MFMA for both
// correction products (the third tap of a step pairs with the third tap of the next step): 768 instead of 1152 matrix
// cycles per step and tile row.
// FUSED (with F8, stride 2, 32-wide tiles, hand-off output; round 5): the 64-channel input is not read -- it is the STEM's output (1 -> 64,
// 7 x 7, reflect padding 3, LeakyReLU), and the workgroup computes the slab of it a chunk needs itself, on the matrix pipe, straight into the slab
// buffers in operand format: the stem's output is the largest tensor of the encoder (16.8 MB per 256 x 256 tile in "f8" format), written once
// and read once -- 1.1 GB of HBM traffic per batch of 32 that bounded the stem launch (133 us) and cost this one its cold reads.
//   * prologue: the (2 TH + 7) x 71 window of the image the tile's stem outputs need, split into f16 hi / lo and stored COLUMN-major (26 halves
//     per column, twice: starting at window row 0 and at row 1), so that the eight vertically adjacent pixels of a stem row's taps are 16 bytes
//     at a 4-byte-aligned address whatever the row: K = (tap column b, tap row a) as in enc_stem7x7_kernel, b = 4 s + (lane >> 4).
//   * a stem phase = one slab (odd rows: TH + 1, even rows: TH) of one 16-channel chunk: v_mfma_f32_16x16x32_f16 tiles of 16 c_out x 16 slab
//     positions (one row, one column parity, half a row; the 33rd odd column of all rows is one more tile), three split-f16 products x two K
//     steps; a lane then holds four consecutive channels of one position: bias, LeakyReLU, f16 hi + the fp8 correction operands, three LDS
//     stores into the slab.  O_c+1 is computed behind step (c, 1) -- into the buffer E_c has just left --, E_c+1 behind step (c, 2).
//   * the K loop is the S2F loop below with weight pieces only.
// Per stem output the sum runs in another order than enc_stem7x7_kernel's (three accumulators of 16x16x32 products): not bit-identical to the
// two-kernel path, same fp32-grade products (tests/test_hip_encoder.py compares both with the oracle).
template <int STRIDE, int LW, int OUT, bool F8 = false, bool FUSED = false>
__global__ __launch_bounds__(512) void enc_conv3x3_h3_kernel(const EncConvParams p) {
    static_assert(!FUSED || (F8 && STRIDE == 2 && LW == 5 && OUT == 1), "the fused stem feeds the stride-2 f8 loop with the hand-off epilogue");
    constexpr int NW = 8, NWN = 4, MB = 2, NBW = 2, CO_WG = 128;
    constexpr int WT = 1 << LW, RPB = 32 / WT, TH = NWN * NBW * RPB, PW = WT + 2;      // output tile TH x WT = 256 pixels
    // S2F (stride 2, f8 operands): activations are staged per CHUNK, not per step.  The slab of tap row 2 (input rows 2r+1) is the
    // slab of tap row 0 (rows 2r-1) moved down one row, so a chunk needs 2 TH + 1 distinct input rows -- an "odd" slab of TH + 1 rows
    // for tap rows 0 and 2 and an "even" slab of TH rows for tap row 1 -- instead of three slabs of TH rows: 76 instead of 108
    // activation pieces per chunk (the K loop of this kernel is bound by what goes through the CU's LDS).
    // Slab layout [row][even cols 2j | odd cols 2j-1][c], so that the even slab is a prefix of the same piece list.
    constexpr bool S2F = F8 && STRIDE == 2;
    constexpr int ROWPITCH = S2F ? 2 * PW : PW;
    constexpr int SLOTS = S2F ? (TH + 1) * ROWPITCH : STRIDE * TH * PW;             // 16-byte slots of one (cgroup, hi/lo) plane of a step (S2F: of a slab)
    constexpr int PP = (SLOTS + 63) / 64, XPL = FUSED ? SLOTS : PP * 64;                // (FUSED: no 1-KiB pieces to pad a plane for)
    constexpr int NXP = 4 * PP, NXPW = (NXP + NW - 1) / NW;
    constexpr int NXA = FUSED ? 0 : NXPW;                                               // activation pieces per wave and slab (S2F loop)
    constexpr int WSLOTS = 12 * CO_WG, NWP = WSLOTS / 64, NWPW = NWP / NW;
    constexpr int KX0 = S2F ? PW : (STRIDE == 2 ? TH * PW : 0), KX1 = STRIDE == 2 ? 0 : 1, KX2 = S2F ? PW + 1 : (STRIDE == 2 ? TH * PW + 1 : 2);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_enc[];
    h8* xbuf = reinterpret_cast<h8*>(smem_enc);         // [2][4][XPL]
    h8* wbuf = xbuf + 2 * 4 * XPL;                      // [2][WSLOTS]
    // FUSED: the image window as column-major f16 strips [start row parity][column][hi | lo: 26 halves each] and the stem's weight fragments
    constexpr int SCOLS = 2 * WT + 8, SLO = 52, SPITCH = 108, STRIP_BYTES = 2 * SCOLS * SPITCH;     // column = 26 halves hi | 26 halves lo | 4 bytes: 27 dwords
    static_assert(!FUSED || (STRIP_BYTES % 16 == 0 && 2 * TH + 7 + 1 <= SLO / 2), "strip geometry");
    [[maybe_unused]] unsigned char* strips = reinterpret_cast<unsigned char*>(wbuf + 2 * WSLOTS);
    [[maybe_unused]] h8* stemw = reinterpret_cast<h8*>(strips + STRIP_BYTES);           // [chunk 4][K step 2][hi/lo][lane]
    NB_TSTAMP(0);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    const int wm = wv / NWN, wn = wv - wm * NWN;
    int b = blockIdx.x;
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x; const int tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int y0 = tile_y * TH, x0 = tile_x * WT, co0 = slice * CO_WG;
    const size_t HW8 = (size_t)p.hin * p.win * 8;
    const _Float16* xn = p.x + (size_t)n * p.c8 * 2 * HW8;

    // OUT == 1 on 32-wide tiles: the hand-off epilogue of the generator's up=1 kernel (nb_up1_handoff_epilogue: straight from the
    // accumulators, packed arithmetic, lane-half trade, whole 16-byte slots) with demodulation 1, no noise, gain 1, no clamp --
    // the same expression as the staged epilogue below.  Its per-channel tables are fetched now, under the first LDS-DMA round trip.
    constexpr bool DIRECT = OUT == 1 && LW == 5;
    __shared__ __attribute__((aligned(16))) float s_one[CO_WG], s_bias[CO_WG], s_osc[CO_WG];
    {
        if (tid < CO_WG) {
            const int co = co0 + tid;
            s_one[tid] = 1.f;
            s_bias[tid] = co < p.c_out ? p.bias[co] : 0.f;
            s_osc[tid] = (p.oscale && co < p.c_out) ? p.oscale[(size_t)n * p.oscale_stride + co] : 1.f;
        }
    }

    // gather descriptors of this wave's activation pieces
    int xcol[NXPW], xrow[NXPW], xpl[NXPW], xdst[NXPW];
    [[maybe_unused]] int xcol_e[NXPW];                  // S2F: the column offsets of the even slab (rows < TH only)
#pragma unroll
    for (int i = 0; i < NXPW; ++i) {
        int q = i * NW + wv;
        q = q < NXP ? q : NXP - 1;
        const int pl4 = q / PP, part = q - pl4 * PP;
        const int e = part * 64 + lane;
        xpl[i] = pl4;
        xdst[i] = pl4 * XPL + part * 64;
        bool valid = e < SLOTS;
        int r, ix;
        if constexpr (S2F) {
            r = e / ROWPITCH;
            const int rem = e - r * ROWPITCH, pl = rem / PW, c = rem - pl * PW;
            ix = pl ? nb_reflect(2 * (x0 + c) - 1, p.win) : 2 * (x0 + c);               // odd columns 2j-1 | even columns 2j
            valid = valid && (pl ? c <= WT : c < WT);
            xrow[i] = 2 * (y0 + r) - 1;                                                 // odd slab; the even slab is one row below
            xcol_e[i] = (valid && r < TH) ? ix * 8 : -1;
        } else if (STRIDE == 1) {
            r = e / PW;
            ix = nb_reflect(x0 + (e - r * PW) - 1, p.win);
            xrow[i] = y0 + r - 1;
        } else {
            const int pl = e / (TH * PW), rem = e - pl * (TH * PW);
            r = rem / PW;
            const int c = rem - r * PW;
            ix = pl ? nb_reflect(2 * (x0 + c) - 1 + p.shift, p.win) : 2 * (x0 + c) + p.shift;        // odd columns 2j-1 | even columns 2j  (+ shift)
            valid = valid && (pl ? c <= WT : c < WT);
            xrow[i] = 2 * (y0 + r) - 1 + p.shift;
        }
        xcol[i] = valid ? ix * 8 : -1;
    }
    auto issue = [&](int t, int buf) {                   // t = chunk * 3 + ky
        const int c = t / 3, ky = t - 3 * c;
        h8* xd = xbuf + buf * 4 * XPL;
#pragma unroll
        for (int i = 0; i < NXPW; ++i) {
            const int cg = 2 * c + (xpl[i] >> 1);
            const _Float16* src = reinterpret_cast<const _Float16*>(p.zeros);
            if (xcol[i] >= 0 && cg < p.c8)
                src = xn + (size_t)(4 * c + xpl[i]) * HW8 + (size_t)nb_reflect(xrow[i] + ky, p.hin) * p.win * 8 + xcol[i];
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(xd + xdst[i]), 16, 0, 0);
        }
        h8* wd = wbuf + buf * WSLOTS;
#pragma unroll
        for (int i = 0; i < NWPW; ++i) {
            const int e = (i * NW + wv) * 64 + lane;
            const int row = e / CO_WG, j = e - row * CO_WG;                      // row = kx*4 + cg*2 + hl
            const _Float16* src = p.wts + (((size_t)t * 12 + row) * p.co_ld + co0 + j) * 8;
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(wd + (i * NW + wv) * 64), 16, 0, 0);
        }
    };

    // one LDS-DMA piece of step t into buffer `buf`: k < NXPW activations, else weights (the pieces of issue(), singly)
    auto piece = [&](auto kk, int t, int buf) {
        constexpr int k = decltype(kk)::value;
#ifdef NB_ENC_ABL_NODMA     // developer ablation (tools/build_variant.sh; timing only, wrong results): the steps re-read what the prologue staged
        (void)t; (void)buf;
        return;
#endif
        if constexpr (k < NXPW) {
            // (36 activation pieces over 8 waves: the fifth round exists for waves 0-3 only.  The others skip it instead of re-copying
            //  the last piece -- allowed for pieces of the SECOND half of a step only: the counted wait at the top of a step counts
            //  the first-half pieces issued after them, which every wave issues in full)
            if constexpr (k >= (NXPW + NWPW) / 2 && k * NW + NW - 1 >= NXP) { if (k * NW + wv >= NXP) return; }
            const int c = t / 3, ky = t - 3 * c;
            const int cg = 2 * c + (xpl[k] >> 1);
            const _Float16* src = reinterpret_cast<const _Float16*>(p.zeros);
            if (xcol[k] >= 0 && cg < p.c8)
                src = xn + (size_t)(4 * c + xpl[k]) * HW8 + (size_t)nb_reflect(xrow[k] + ky, p.hin) * p.win * 8 + xcol[k];
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(xbuf + buf * 4 * XPL + xdst[k]), 16, 0, 0);
        } else {
            constexpr int i = k - NXPW;
            const int e = (i * NW + wv) * 64 + lane;
            const int row = e / CO_WG, j = e - row * CO_WG;
            const _Float16* src = p.wts + (((size_t)t * 12 + row) * p.co_ld + co0 + j) * 8;
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(wbuf + buf * WSLOTS + (i * NW + wv) * 64), 16, 0, 0);
        }
    };

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    const int T = p.nchunks * 3;
    const int a_base = lh * 2 * CO_WG + wm * 64 + l31;
    const int b_base = lh * 2 * XPL + ((wn * NBW) * RPB + (l31 >> LW)) * ROWPITCH + (l31 & (WT - 1));
    if constexpr (!S2F) issue(0, 0);
    if constexpr (S2F) {
        // ---- stride 2, f8 operands: slabs per chunk ---------------------------------------------------------------------------
        // Step (c, S), S = tap row: S = 0 reads the odd slab O_c (rows 0 .. TH-1), S = 1 the even slab E_c, S = 2 O_c one row down.
        // Two slab buffers; O_c lives in buffer c & 1, E_c in the other.  What is issued FOR a step is a list of pieces per wave:
        //   kind 0 (for a step S = 0):  the NXPW pieces of O + NWPW weight pieces      kind 1: the pieces of E + weights
        //   kind 2: weights only
        // and, as in the loop below, a list's first half goes out in the second half of the step two before its consumer (behind the
        // mid-step barrier that frees the target: E_c's buffer takes O_c+1 once step (c, 1) has read it, O_c's buffer takes E_c+1
        // once step (c, 2) has), the second half in the first half of the step before.  The counted wait at the top of a step
        // leaves exactly the first half of the NEXT step's list in flight.  Weight buffers alternate with the step as before.
        constexpr int NPC = NXA + NWPW;                                // pieces per wave of kinds 0 and 1
        constexpr int FH01 = NPC / 2, FH2 = (NWPW + 1) / 2;           // first halves
        static_assert(NXP == NXPW * NW, "every wave issues the same number of activation pieces (no skipping: counted waits)");
        const int NC = p.nchunks, T = NC * 3;
        // The pieces go out from inline assembly (nb_lds_dma16_m / _s, nb_h3_common.h): the builtin pins every later wait for a
        // fragment read at lgkmcnt(0), i.e. a step could not start its matrix work before ALL of its reads were back.  Per-lane
        // source of chunk 0 + a per-chunk stride (0 for the lanes that read the zero page); weights: uniform base + lane offset.
        const unsigned lds0 = (unsigned)(uintptr_t)NB_LDS_PTR(smem_enc);
        const char* xs_o[NXPW]; const char* xs_e[NXPW];
        unsigned xst_o[NXPW], xst_e[NXPW], wof[NWPW];
#pragma unroll
        for (int i = 0; i < NXPW; ++i) {
            const bool vo = xcol[i] >= 0, ve = xcol_e[i] >= 0;
            const char* base = reinterpret_cast<const char*>(xn + (size_t)xpl[i] * HW8);
            xs_o[i] = vo ? base + ((size_t)nb_reflect(xrow[i], p.hin) * p.win * 8 + xcol[i]) * 2 : reinterpret_cast<const char*>(p.zeros);
            xs_e[i] = ve ? base + ((size_t)nb_reflect(xrow[i] + 1, p.hin) * p.win * 8 + xcol_e[i]) * 2 : reinterpret_cast<const char*>(p.zeros);
            xst_o[i] = vo ? (unsigned)(4 * HW8 * 2) : 0u;
            xst_e[i] = ve ? (unsigned)(4 * HW8 * 2) : 0u;
        }
#pragma unroll
        for (int i = 0; i < NWPW; ++i) {
            const int e = (i * NW + wv) * 64 + lane;
            const int row = e / CO_WG, j = e - row * CO_WG;
            wof[i] = (unsigned)(((size_t)row * p.co_ld + co0 + j) * 16);
        }
        const size_t wstep = (size_t)12 * p.co_ld * 16;               // bytes of one (chunk, tap row) weight sub-chunk
        // piece k of the list of kind KIND for step (cc, KIND); past the end: the last chunk again (harmless re-copies keep the counts uniform)
        auto piece2 = [&](auto kind_, auto kk, int cc) {
            constexpr int KIND = decltype(kind_)::value, k = decltype(kk)::value;
            constexpr int NA = KIND == 2 ? 0 : NXA;
            const int src_c = cc < NC ? cc : NC - 1;
            const int t = 3 * cc + KIND, src_t = 3 * src_c + KIND;
            if constexpr (k < NA) {
                const int buf = KIND == 0 ? (cc & 1) : ((cc + 1) & 1);
                if constexpr (KIND == 0) nb_lds_dma16_m(xs_o[k] + (size_t)src_c * xst_o[k], lds0 + (unsigned)(buf * 4 * XPL + xdst[k]) * 16u, ~0ull);
                else nb_lds_dma16_m(xs_e[k] + (size_t)src_c * xst_e[k], lds0 + (unsigned)(buf * 4 * XPL + xdst[k]) * 16u, ~0ull);
            } else {
                constexpr int i = k - NA;
                nb_lds_dma16_s(reinterpret_cast<const char*>(p.wts) + (size_t)src_t * wstep, wof[i],
                               lds0 + (unsigned)(2 * 4 * XPL + (t & 1) * WSLOTS + (i * NW + wv) * 64) * 16u);
            }
        };
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
        // before step 0: all of list 0 of chunk 0 and the first half of list 1
        nb_static_for<0, NPC>([&](auto k) { piece2(K0{}, k, 0); });
        nb_static_for<0, FH01>([&](auto k) { piece2(K1{}, k, 0); });
#define NB_SB __builtin_amdgcn_sched_barrier(0)
#define NB_Q(v, q, src) { const i32x4 t_ = __builtin_bit_cast(i32x4, (src)); v[4 * (q)] = t_[0]; v[4 * (q) + 1] = t_[1]; v[4 * (q) + 2] = t_[2]; v[4 * (q) + 3] = t_[3]; }
        const int sa = lh ? 116 : 127, sb = lh ? 129 : 118;        // E8M0 block scales: fp8(w) fp8(xl 2^9) 2^-9 | fp8(wl 2^11) 2^-11 fp8(x/4) 2^2
        h8 ah0[MB], ah1[MB], ah2[MB], bh0[NBW], bh1[NBW], bh2[NBW];
        i32x8 al01[MB], bl01[NBW], al2[MB], bl2[NBW];          // fp8 tuples: (tap 0 | tap 1), (tap 2 of an even step | of the odd step after it)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { ah2[mb] = h8{}; al2[mb] = i32x8{}; al01[mb] = i32x8{}; }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) { bh2[nb] = h8{}; bl2[nb] = i32x8{}; bl01[nb] = i32x8{}; }
        constexpr int NM = MB * NBW, NF = MB + NBW;       // MFMAs per group; fragment reads per (tap, hi or lo)
        // one fragment read: i < MB: weights of block i, else activations of pixel block i - MB; KX = the tap's slot offset in the slab
        auto rd_hi = [&](auto i_, h8 (&a)[MB], h8 (&b)[NBW], const h8* wb, const h8* xb, auto kx_) {
            constexpr int i = decltype(i_)::value, kx = decltype(kx_)::value;
            constexpr int ko = kx == 0 ? KX0 : (kx == 1 ? KX1 : KX2);
            if constexpr (i < MB) a[i] = wb[a_base + kx * 4 * CO_WG + i * 32];
            else b[i - MB] = xb[b_base + (i - MB) * RPB * ROWPITCH + ko];
        };
        auto rd_lo = [&](auto i_, auto q_, i32x8 (&a)[MB], i32x8 (&b)[NBW], const h8* wb, const h8* xb, auto kx_) {
            constexpr int i = decltype(i_)::value, q = decltype(q_)::value, kx = decltype(kx_)::value;
            constexpr int ko = kx == 0 ? KX0 : (kx == 1 ? KX1 : KX2);
            if constexpr (i < MB) { NB_Q(a[i], q, wb[a_base + kx * 4 * CO_WG + CO_WG + i * 32]); }
            else { NB_Q(b[i - MB], q, xb[b_base + XPL + (i - MB) * RPB * ROWPITCH + ko]); }
        };
        // a group of NM MFMAs (tile k = (k / NBW, k % NBW)) with NFILL fillers dealt evenly into the gaps behind them
        // (sf: what else rides in gap k of the group -- the fused stem's operations, see stem_op)
        auto group = [&](auto nfill_, auto&& mf, auto&& ff, auto&& sf) {
            constexpr int NFILL = decltype(nfill_)::value;
            nb_static_for<0, NM>([&](auto k_) {
                constexpr int k = decltype(k_)::value;
                mf(std::integral_constant<int, k / NBW>{}, std::integral_constant<int, k % NBW>{});
                NB_SB;
                nb_static_for<0, NFILL>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    if constexpr (i * NM / (NFILL > 0 ? NFILL : 1) == k) ff(i_);
                });
                NB_SB;
                sf(k_);
                NB_SB;
            });
        };
        auto mf_f16 = [&](h8 (&a)[MB], h8 (&b)[NBW]) {
            return [&](auto mb_, auto nb_) {
                constexpr int mb = decltype(mb_)::value, nb = decltype(nb_)::value;
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
            };
        };
        auto mf_fp8 = [&](i32x8 (&a)[MB], i32x8 (&b)[NBW]) {
            return [&](auto mb_, auto nb_) {
                constexpr int mb = decltype(mb_)::value, nb = decltype(nb_)::value;
                acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[mb], b[nb], acc[mb][nb], 0, 0, 0, sa, 0, sb);
            };
        };
        auto no_mf = [&](auto, auto) {};
        // ---- FUSED: the stem on the matrix pipe (see the kernel's header) ----
        __shared__ __attribute__((aligned(16))) float s_bias0[64];
        [[maybe_unused]] auto fused_prologue = [&]() {
            if (tid < 64) s_bias0[tid] = p.stem_b[tid];
            const float* img = p.img + (size_t)n * p.hin * p.win;
            // window row j <-> image row 2 y0 - 4 + j, column w <-> image column 2 x0 - 4 + w (the stem's reflect padding); row 23 = zeros (the
            // zero-weight eighth tap of the last slab row)
            constexpr int NE = (SCOLS * 24 + 511) / 512;
            float v_[NE];
#pragma unroll
            for (int q = 0; q < NE; ++q) {                                  // (all loads first: one round trip)
                const int e = tid + 512 * q, j = e / SCOLS, w = e - j * SCOLS;
                v_[q] = 0.f;
                if (e < SCOLS * 24 && j < 2 * TH + 7)
                    v_[q] = img[(size_t)nb_reflect(2 * y0 - 4 + j, p.hin) * p.win + nb_reflect(2 * x0 - 4 + w, p.win)];
            }
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                const int e = tid + 512 * q, j = e / SCOLS, w = e - j * SCOLS;
                if (e < SCOLS * 24) {
                    float v = v_[q];
                    if (j < 2 * TH + 7) {
                        if (p.preproc == 1) v = (1.f - v) * 2.f - 1.f;          // '-11inverse' (autoenc/base.py:30-52)
                        else if (p.preproc == 2) v = 1.f - v;                   // 'inverse'
                    }
                    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                    *reinterpret_cast<_Float16*>(strips + w * SPITCH + 2 * j) = hi;
                    *reinterpret_cast<_Float16*>(strips + w * SPITCH + SLO + 2 * j) = lo;
                    if (j >= 1) {
                        *reinterpret_cast<_Float16*>(strips + (SCOLS + w) * SPITCH + 2 * (j - 1)) = hi;
                        *reinterpret_cast<_Float16*>(strips + (SCOLS + w) * SPITCH + SLO + 2 * (j - 1)) = lo;
                    }
                }
            }
            // A fragments of v_mfma_f32_16x16x32_f16: lane (row = c_out l15, K group kg) of K step s holds w[c_out][tap rows 0..7][tap column 4 s + kg]
            {
                const int chunk = tid >> 7, s_ = (tid >> 6) & 1, l15 = lane & 15, kg = lane >> 4;
                const int b = 4 * s_ + kg;
                h8 hi8, lo8;
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    const float wv_ = (a < 7 && b < 7) ? p.stem_w[(chunk * 16 + l15) * 50 + a * 7 + b] : 0.f;
                    const _Float16 hi = (_Float16)wv_;
                    hi8[a] = hi; lo8[a] = (_Float16)(wv_ - (float)hi);
                }
                stemw[((chunk * 2 + s_) * 2 + 0) * 64 + lane] = hi8;
                stemw[((chunk * 2 + s_) * 2 + 1) * 64 + lane] = lo8;
            }
        };
        // One slab of chunk C: KIND 0 = the odd slab O_C (image rows 2 (y0 + r) - 1, r = 0 .. TH) into buffer C & 1, KIND 1 = the even slab E_C (rows
        // 2 (y0 + r), r < TH) into the other.  Tiles (16 slab positions) go round-robin over the waves; a wave's five tiles are a pipeline of
        // NSOPS = 15 operations on registers that live across the steps -- L(k): the fragment reads of tile k (+ the chunk's weight fragments
        // before the first); M(k): its six products; S(k): bias (the accumulator's initial value), LeakyReLU, conversion, three LDS stores --
        // in the order  L0 L1 M0 | L2 M1 S0 | L3 M2 S1 | L4 M3 S2 | M4 S3 S4  (two fragment buffers, two accumulators: a tile's reads are two or
        // three operations ahead of its products, its conversion two or three behind them -- an operation that waits stalls the wave's K loop).  The chunk-0 slabs run
        // them back to back in the prologue; the others are dealt into the gaps behind the matrix instructions of the steps (see step()).
        constexpr int NSOPS = 15;
        [[maybe_unused]] h8 st_wh[2], st_wl[2];                          // the chunk's weight fragments of the two K steps
        [[maybe_unused]] h8 st_fh[2][2], st_fl[2][2];                    // image fragments (hi, lo) [tile & 1][K step]
        [[maybe_unused]] f32x4 st_a[2], st_bias;
        // a wave's tiles 0-3 are rows r0 + 2 k of ONE (column parity, half row): their addresses are those of tile 0 + constants
        [[maybe_unused]] const unsigned char* st_bp = nullptr;
        [[maybe_unused]] unsigned char* st_hi = nullptr;
        [[maybe_unused]] unsigned char* st_lo = nullptr;
        [[maybe_unused]] auto stem_op = [&](auto kind_, auto c_, auto j_) {
            constexpr int KIND = decltype(kind_)::value, C = decltype(c_)::value, J = decltype(j_)::value;
            constexpr int NR = KIND == 0 ? TH + 1 : TH, NBLKS = 4 * NR + 1, NIT = (NBLKS + NW - 1) / NW;
            static_assert(NIT == 5 && (NIT - 1) * NW <= 4 * NR, "five tiles per wave; only the last can be the extra one (the 33rd odd column of all rows) or missing");
            // operation J: type (0 = L, 1 = M, 2 = S) and tile:  L0 L1 M0 | L2 M1 S0 | L3 M2 S1 | L4 M3 S2 | M4 S3 S4
            constexpr int TYPES[NSOPS] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2, 1, 2, 2}, TILES[NSOPS] = {0, 1, 0, 2, 1, 0, 3, 2, 1, 4, 3, 2, 4, 3, 4};
            constexpr int TYPE = TYPES[J], K = TILES[J];
            typedef int i32x4a __attribute__((ext_vector_type(4), aligned(4)));
            const int l15 = lane & 15, kg = lane >> 4;
            h8* sb = xbuf + (KIND == 0 ? (C & 1) : ((C + 1) & 1)) * 4 * XPL;
            const unsigned char* sbase = strips + (KIND * SCOLS) * SPITCH;          // start rows: even (odd slab) | odd (even slab)
            // the conv's own reflect padding: image row -1 = row 1 (odd slab of the first tile row), image column -1 = column 1
            if constexpr (J == 0) {
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) { st_wh[s_] = stemw[((C * 2 + s_) * 2 + 0) * 64 + lane]; st_wl[s_] = stemw[((C * 2 + s_) * 2 + 1) * 64 + lane]; }
                st_bias = *reinterpret_cast<const f32x4*>(s_bias0 + 16 * C + 4 * kg);
                const int par = (wv >> 1) & 1, cl = 16 * (wv & 1) + l15, r0 = wv >> 2;
                int col0 = 2 * cl + (par ? 0 : 1);
                if (par && cl == 0 && x0 == 0) col0 += 2;
                st_bp = sbase + (col0 + kg) * SPITCH + 4 * r0;
                const int slot = r0 * ROWPITCH + par * PW + cl;
                // channels 4 kg .. + 3 of the chunk: half of a hi slot of group kg >> 1, bytes 4 kg .. of the chunk's two lo slots
                st_hi = reinterpret_cast<unsigned char*>(sb + 2 * (kg >> 1) * XPL + slot) + 8 * (kg & 1);
                st_lo = reinterpret_cast<unsigned char*>(sb + 1 * XPL + slot) + 4 * kg;
            }
            // the last tile of a wave (regular, the extra one, or none): column parity, column and row of the lane's position, whether it stores
            [[maybe_unused]] int par4 = 0, cl4 = 0, rl4 = 0; [[maybe_unused]] bool ok4 = false;
            if constexpr (K == NIT - 1) {
                const int blk = wv + NW * K;
                const bool sp = blk >= 4 * NR;
                par4 = sp ? 1 : (blk >> 1) & 1;
                cl4 = sp ? WT : 16 * (blk & 1) + l15;
                rl4 = sp ? (l15 < NR ? l15 : NR - 1) : (blk >> 2);
                ok4 = blk < NBLKS && (!sp || l15 < NR);
            }
#if defined(NB_ENC_ABL_ST_NOREAD) || defined(NB_ENC_ABL_ST_NOMMA) || defined(NB_ENC_ABL_ST_NOSTORE) || defined(NB_ENC_ABL_ST_NONE)
            // developer ablations of the fused stem (tools/build_variant.sh; timing only, wrong results)
            if constexpr (J == 0) {
                for (int b_ = 0; b_ < 2; ++b_) { st_fh[b_][0] = st_wh[0]; st_fh[b_][1] = st_wh[1]; st_fl[b_][0] = st_wl[0]; st_fl[b_][1] = st_wl[1]; st_a[b_] = st_bias; }
            }
#endif
#ifdef NB_ENC_ABL_ST_NONE
            return;
#endif
            if constexpr (TYPE == 0) {
#ifdef NB_ENC_ABL_ST_NOREAD
                return;
#endif
                const unsigned char* bp;
                if constexpr (K < NIT - 1) {
                    bp = st_bp + 8 * K;
                    if constexpr (KIND == 0 && K == 0) bp += (y0 == 0 && wv < 4) ? 4 : 0;
                } else {
                    const int re = (KIND == 0 && y0 == 0 && rl4 == 0) ? 1 : rl4;
                    int col0 = 2 * cl4 + (par4 ? 0 : 1);
                    if (par4 && cl4 == 0 && x0 == 0) col0 += 2;
                    bp = sbase + (col0 + kg) * SPITCH + 4 * re;
                }
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    st_fh[K & 1][s_] = __builtin_bit_cast(h8, *reinterpret_cast<const i32x4a*>(bp + s_ * 4 * SPITCH));
                    st_fl[K & 1][s_] = __builtin_bit_cast(h8, *reinterpret_cast<const i32x4a*>(bp + s_ * 4 * SPITCH + SLO));
                }
            } else if constexpr (TYPE == 1) {
#ifdef NB_ENC_ABL_ST_NOMMA
                return;
#endif
                f32x4 a = st_bias;
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(st_wh[s_], st_fh[K & 1][s_], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(st_wh[s_], st_fl[K & 1][s_], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(st_wl[s_], st_fh[K & 1][s_], a, 0, 0, 0);
                }
                st_a[K & 1] = a;
            } else {
#ifdef NB_ENC_ABL_ST_NOSTORE
                return;
#endif
                f32x4 t = st_a[K & 1];
                const f32x4 ta = t * p.slope;
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = fmaxf(t[i], ta[i]);               // LeakyReLU, 0 <= slope <= 1
                const h2 h01 = __builtin_convertvector(f32x2{t[0], t[1]}, h2), h23 = __builtin_convertvector(f32x2{t[2], t[3]}, h2);
                const f32x4 xl = {nb_sub_f16(t[0], h01, false), nb_sub_f16(t[1], h01, true), nb_sub_f16(t[2], h23, false), nb_sub_f16(t[3], h23, true)};
                const unsigned lo_xl = nb_pk4_fp8_sat_scaled(xl[0], xl[1], xl[2], xl[3], 0x1p-9f);       // (FP16_OVFL is set: the conversions saturate)
                const unsigned lo_w = nb_pk4_fp8_sat_scaled(t[0], t[1], t[2], t[3], 4.f);
                const u32x2 hi2 = {__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
                if constexpr (K < NIT - 1) {
                    constexpr int OFF = K * 2 * ROWPITCH * 16;
                    *reinterpret_cast<u32x2*>(st_hi + OFF) = hi2;
                    *reinterpret_cast<unsigned*>(st_lo + OFF) = lo_xl;
                    *reinterpret_cast<unsigned*>(st_lo + OFF + 2 * XPL * 16) = lo_w;
                } else if (ok4) {
                    const int slot = rl4 * ROWPITCH + par4 * PW + cl4;
                    *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(sb + 2 * (kg >> 1) * XPL + slot) + 8 * (kg & 1)) = hi2;
                    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(sb + 1 * XPL + slot) + 4 * kg) = lo_xl;
                    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(sb + 3 * XPL + slot) + 4 * kg) = lo_w;
                }
            }
        };
        // a whole slab, back to back (fences: left alone the scheduler puts every tile's reads right in front of its products)
        [[maybe_unused]] auto stem_phase = [&](auto kind_, auto c_) {
#ifdef NB_ENC_STEM_STAMPS   // developer build: wave 0's clock ticks (s_memtime) per operation type of this slab -> stamp slots 5 (L), 6 (M), 7 (S)
            unsigned long long tt[3] = {0, 0, 0};
            constexpr int TYPES_[NSOPS] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 0, 1, 2, 1, 2, 2};
            nb_static_for<0, NSOPS>([&](auto j_) {
                const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
                NB_SB; stem_op(kind_, c_, j_); NB_SB;
                tt[TYPES_[decltype(j_)::value]] += __builtin_amdgcn_s_memtime() - t0_;
            });
            if (p.tstamps && tid == 0)
                for (int i = 0; i < 3; ++i) p.tstamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + 5 + i] = tt[i];
#else
            nb_static_for<0, NSOPS>([&](auto j_) { stem_op(kind_, c_, j_); NB_SB; });
#endif
        };
        using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
        // step (c, S); ODD = parity of t = 3 c + S (selects the half of the tap-2 tuple).  Per accumulator tile the products arrive in the
        // order of the per-step loop below: tap 2 of step t-1, on even t the tap-2 corrections of t-2 and t-1, tap 0, tap 1, corrections
        // of taps 0 + 1.  Reads and pieces ride behind the MFMAs: a step opens with matrix work on registers it already holds.
        // FUSED: CC = the chunk as a constant (else -1).  The stem's slabs of the NEXT chunk ride in the gaps of the steps: O_c+1 -- into the buffer
        // E_c leaves at the mid-step barrier of (c, 1) -- operations 0-2 behind the last group of step (c, 1), 3-14 in the groups of step (c, 2) before
        // its barrier; E_c+1 -- into the buffer O_c leaves at the barrier of (c, 2) -- operations 0-2 behind the last group of (c, 2), 3-14 in step
        // (c + 1, 0).  Each is complete (stores counted out at the top barrier) a step before its first reader; E_0 runs whole in step (0, 0).
        auto no_sf = [](auto) {};
        auto step = [&](auto s_, auto odd_, auto first_, int c, auto cc_) {
            constexpr int S = decltype(s_)::value, ODD = decltype(odd_)::value, FIRST = decltype(first_)::value, CC = decltype(cc_)::value;
            static_assert(!FUSED || CC >= 0, "the fused loop is unrolled over its four chunks");
            constexpr bool PRE = FUSED && ((S == 2 && CC + 1 < 4) || S == 0);                  // operations 3-14 of O_CC+1 (S = 2) / E_CC (S = 0; all 15 of E_0)
            constexpr int PRE0 = (S == 0 && CC == 0) ? 0 : 3;
            constexpr bool POST = FUSED && CC + 1 < 4 && S != 0;                               // operations 0-2 of O_CC+1 (S = 1) / E_CC+1 (S = 2)
            using PRE_K = std::integral_constant<int, S == 2 ? 0 : 1>; using PRE_C = std::integral_constant<int, S == 2 ? CC + 1 : CC>;
            using POST_K = std::integral_constant<int, S == 1 ? 0 : 1>; using POST_C = std::integral_constant<int, CC + 1>;
            constexpr int G3B = FIRST ? 0 : (ODD ? 4 : 8), NGPRE = G3B + 8;                    // gaps before the mid-step barrier (four per group)
            auto sf_pre = [&](auto base_) {
                return [&](auto k_) {
                    if constexpr (PRE) {
                        constexpr int g = decltype(base_)::value + decltype(k_)::value;
                        nb_static_for<0, NSOPS - PRE0>([&](auto q_) {
                            constexpr int q = decltype(q_)::value;
                            if constexpr (q * NGPRE / (NSOPS - PRE0) == g) { stem_op(PRE_K{}, PRE_C{}, std::integral_constant<int, PRE0 + q>{}); NB_SB; }
                        });
                    }
                };
            };
            auto sf_post = [&](auto k_) {
                if constexpr (POST && decltype(k_)::value < 3) { stem_op(POST_K{}, POST_C{}, k_); NB_SB; }
            };
            using KN = std::integral_constant<int, (S + 1) % 3>;          // kind of the next step's list, of the one after
            using KNN = std::integral_constant<int, (S + 2) % 3>;
            constexpr int LEN_N = KN::value == 2 ? NWPW : NPC, FH_N = KN::value == 2 ? FH2 : FH01;
            constexpr int FH_NN = KNN::value == 2 ? FH2 : FH01;
            const int cn = S == 2 ? c + 1 : c, cnn = S == 0 ? c : c + 1;  // their chunks
            const int t = 3 * c + S;
            if constexpr (FUSED) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(FH_N) : "memory");      // (+ the slab stores of the stem's operations)
            } else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(FH_N) : "memory");      // this step's data has landed, everybody's
            NB_SB;
            const h8* xb = xbuf + (S == 1 ? ((c + 1) & 1) : (c & 1)) * 4 * XPL + (S == 2 ? ROWPITCH : 0);
            const h8* wb = wbuf + (t & 1) * WSLOTS;
            auto rd_a = [&](auto i_) {                      // 2 NF reads: hi fragments of taps 0 and 1
                constexpr int i = decltype(i_)::value;
                if constexpr (i < NF) rd_hi(i_, ah0, bh0, wb, xb, T0{});
                else rd_hi(std::integral_constant<int, i - NF>{}, ah1, bh1, wb, xb, T1{});
            };
            auto rd_b = [&](auto i_) {                      // 2 NF reads: lo tuples of taps 0 and 1
                constexpr int i = decltype(i_)::value;
                if constexpr (i < NF) rd_lo(i_, T0{}, al01, bl01, wb, xb, T0{});
                else rd_lo(std::integral_constant<int, i - NF>{}, T1{}, al01, bl01, wb, xb, T1{});
            };
            auto dma_n = [&](auto i_) { piece2(KN{}, std::integral_constant<int, FH_N + decltype(i_)::value>{}, cn); };
            auto dma_nn = [&](auto i_) { piece2(KNN{}, i_, cnn); };
            if constexpr (FIRST) {
                nb_static_for<0, 2 * NF>(rd_a); nb_static_for<0, 2 * NF>(rd_b); NB_SB;
            } else {
                group(std::integral_constant<int, 2 * NF>{}, mf_f16(ah2, bh2), rd_a, sf_pre(std::integral_constant<int, 0>{}));                       // tap 2 of step t-1
                if constexpr (!ODD) group(std::integral_constant<int, 2 * NF>{}, mf_fp8(al2, bl2), rd_b, sf_pre(std::integral_constant<int, 4>{}));   // tap-2 corrections of t-2, t-1
            }
            if constexpr (!FIRST && ODD) {
                group(std::integral_constant<int, 2 * NF + LEN_N - FH_N>{}, mf_f16(ah0, bh0), [&](auto i_) {  // tap 0
                    constexpr int i = decltype(i_)::value;
                    if constexpr (i < 2 * NF) rd_b(i_); else dma_n(std::integral_constant<int, i - 2 * NF>{});
                }, sf_pre(std::integral_constant<int, G3B>{}));
            } else {
                group(std::integral_constant<int, LEN_N - FH_N>{}, mf_f16(ah0, bh0), dma_n, sf_pre(std::integral_constant<int, G3B>{}));                // tap 0 | second half of the next list
            }
            group(std::integral_constant<int, 2 * NF>{}, mf_f16(ah1, bh1), [&](auto i_) {                   // tap 1 | this step's tap-2 operands
                constexpr int i = decltype(i_)::value;
                if constexpr (i < NF) rd_hi(i_, ah2, bh2, wb, xb, T2{});
                else rd_lo(std::integral_constant<int, i - NF>{}, std::integral_constant<int, ODD>{}, al2, bl2, wb, xb, T2{});
            }, sf_pre(std::integral_constant<int, G3B + 4>{}));
            // the step's last fragment reads are in registers for every wave: the buffers it was the last reader of take new data
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            NB_SB;
            group(std::integral_constant<int, FH_NN>{}, mf_fp8(al01, bl01), dma_nn, sf_post);              // corrections 0 + 1 | first half of the list after the next
            (void)no_mf; (void)no_sf;
        };
        using O0 = std::integral_constant<int, 0>; using O1 = std::integral_constant<int, 1>;
        using CN = std::integral_constant<int, -1>;
        if constexpr (FUSED) {
            nb_set_fp16_ovfl();
            fused_prologue();
            __syncthreads();
            NB_TSTAMP(5);                              // (debug: the image strips and the stem's weight fragments are in LDS)
            stem_phase(K0{}, std::integral_constant<int, 0>{});          // O_0; E_0 rides in step (0, 0), whose buffers it does not touch
            NB_SB;
        }
        if (p.tstamps) { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(FH01) : "memory"); NB_TSTAMP(1); }      // (debug: when the first step could begin)
        if constexpr (FUSED) {
            using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;
            using C3 = std::integral_constant<int, 3>;
            step(K0{}, O0{}, O1{}, 0, C0{}); step(K1{}, O1{}, O0{}, 0, C0{}); step(K2{}, O0{}, O0{}, 0, C0{});
            step(K0{}, O1{}, O0{}, 1, C1{}); step(K1{}, O0{}, O0{}, 1, C1{}); step(K2{}, O1{}, O0{}, 1, C1{});
            step(K0{}, O0{}, O0{}, 2, C2{}); step(K1{}, O1{}, O0{}, 2, C2{}); step(K2{}, O0{}, O0{}, 2, C2{});
            step(K0{}, O1{}, O0{}, 3, C3{}); step(K1{}, O0{}, O0{}, 3, C3{}); step(K2{}, O1{}, O0{}, 3, C3{});
        } else {
        step(K0{}, O0{}, O1{}, 0, CN{}); step(K1{}, O1{}, O0{}, 0, CN{}); step(K2{}, O0{}, O0{}, 0, CN{});
        for (int c = 1; c < NC; c += 2) {
            step(K0{}, O1{}, O0{}, c, CN{}); step(K1{}, O0{}, O0{}, c, CN{}); step(K2{}, O1{}, O0{}, c, CN{});
            if (c + 1 < NC) { step(K0{}, O0{}, O0{}, c + 1, CN{}); step(K1{}, O1{}, O0{}, c + 1, CN{}); step(K2{}, O0{}, O0{}, c + 1, CN{}); }
        }
        }
        NB_SB;
        // the last step's tap 2 and the last tuple of tap-2 corrections (an odd number of steps: it holds one tap only)
        nb_static_for<0, NM>([&](auto k_) { constexpr int k = decltype(k_)::value; mf_f16(ah2, bh2)(std::integral_constant<int, k / NBW>{}, std::integral_constant<int, k % NBW>{}); NB_SB; });
        if (T & 1) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 4; r < 8; ++r) al2[mb][r] = 0;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 4; r < 8; ++r) bl2[nb][r] = 0;
        }
        nb_static_for<0, NM>([&](auto k_) { constexpr int k = decltype(k_)::value; mf_fp8(al2, bl2)(std::integral_constant<int, k / NBW>{}, std::integral_constant<int, k % NBW>{}); NB_SB; });
#undef NB_Q
#undef NB_SB
    } else if constexpr (F8) {
        // Two buffers, and the LDS-DMA pieces of a step spread over the MFMA groups of the TWO steps before it -- a burst of eight
        // pieces right after a barrier stalls both waves of a SIMD at the same moment, with nobody feeding the matrix pipe (the
        // up=2 generator kernel's finding; there it was a third of the loop).  Step t's buffers are free once every wave has done
        // the step's last fragment read (mid-step barrier): pieces 0 .. HP-1 of step t+2 follow in the second half of step t,
        // pieces HP .. of step t+1 went out in its first half, two at a time between groups of four MFMAs.  The wait at the top of
        // a step is counted: everything but the HP pieces issued in the second half of the step before.
        constexpr int NPC = NXPW + NWPW, HP = NPC / 2;          // LDS-DMA pieces a wave issues per step; its first half
        static_assert(NPC % 2 == 0 && NPC <= 8, "the placement below is written for six or eight pieces");
        constexpr int G1 = (HP + 1) / 2;                       // pieces per group: [0, G1) [G1, HP) | [HP, HP + G1) [HP + G1, NPC)
        auto pieces = [&](auto lo_, auto hi_, int t, int buf) {
            constexpr int lo = decltype(lo_)::value, hi = decltype(hi_)::value;
            if constexpr (lo + 0 < hi) piece(std::integral_constant<int, lo + 0>{}, t, buf);
            if constexpr (lo + 1 < hi) piece(std::integral_constant<int, lo + 1>{}, t, buf);
            if constexpr (lo + 2 < hi) piece(std::integral_constant<int, lo + 2>{}, t, buf);
            if constexpr (lo + 3 < hi) piece(std::integral_constant<int, lo + 3>{}, t, buf);
        };
        using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, G1>; using P2 = std::integral_constant<int, HP>;
        using P3 = std::integral_constant<int, HP + G1>; using P4 = std::integral_constant<int, NPC>;
        pieces(P0{}, P2{}, 1 < T ? 1 : T - 1, 1);             // before step 0: all of step 0 is in flight (issue(0, 0) above) and the first half of step 1
#define NB_SB __builtin_amdgcn_sched_barrier(0)
#define NB_Q(v, q, src) { const i32x4 t_ = __builtin_bit_cast(i32x4, (src)); v[4 * (q)] = t_[0]; v[4 * (q) + 1] = t_[1]; v[4 * (q) + 2] = t_[2]; v[4 * (q) + 3] = t_[3]; }
        const int sa = lh ? 116 : 127, sb = lh ? 129 : 118;        // E8M0 block scales: fp8(w) fp8(xl 2^9) 2^-9 | fp8(wl 2^11) 2^-11 fp8(x/4) 2^2
        h8 ah0[MB], ah1[MB], ah2[MB], bh0[NBW], bh1[NBW], bh2[NBW];
        i32x8 al01[MB], bl01[NBW], al2[MB], bl2[NBW];          // fp8 tuples: (tap 0 | tap 1), (tap 2 of an even step | of the odd step after it)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 8; ++r) al2[mb][r] = 0;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 8; ++r) bl2[nb][r] = 0;
        auto main4 = [&](h8 (&a)[MB], h8 (&b)[NBW]) {
#ifdef NB_ENC_ABL_NOMFMA    // developer ablation: no matrix work (the operands are kept alive)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) asm volatile("" :: "v"(a[mb]));
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) asm volatile("" :: "v"(b[nb]));
            return;
#endif
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) { acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb], b[nb], acc[mb][nb], 0, 0, 0); NB_SB; }
        };
        auto corr4 = [&](i32x8 (&a)[MB], i32x8 (&b)[NBW]) {
#ifdef NB_ENC_ABL_NOMFMA
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) asm volatile("" :: "v"(a[mb]));
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) asm volatile("" :: "v"(b[nb]));
            return;
#endif
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[mb], b[nb], acc[mb][nb], 0, 0, 0, sa, 0, sb);
                    NB_SB;
                }
        };
        auto step = [&](int t, int odd, bool first) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HP) : "memory");      // my share of step t has landed (the first half of step t+1's may be in flight)
            __builtin_amdgcn_s_barrier();                         // ... everybody's has
            NB_SB;
            const int tn = t + 1 < T ? t + 1 : T - 1, tnn = t + 2 < T ? t + 2 : T - 1;      // (past the end: harmless re-copies keep the counts uniform)
            const h8* xb = xbuf + (t & 1) * 4 * XPL;
            const h8* wb = wbuf + (t & 1) * WSLOTS;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                ah0[mb] = wb[a_base + mb * 32];
                NB_Q(al01[mb], 0, wb[a_base + CO_WG + mb * 32]);
                ah1[mb] = wb[a_base + 4 * CO_WG + mb * 32];
                NB_Q(al01[mb], 1, wb[a_base + 4 * CO_WG + CO_WG + mb * 32]);
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                bh0[nb] = xb[b_base + nb * RPB * PW + KX0];
                NB_Q(bl01[nb], 0, xb[b_base + XPL + nb * RPB * PW + KX0]);
                bh1[nb] = xb[b_base + nb * RPB * PW + KX1];
                NB_Q(bl01[nb], 1, xb[b_base + XPL + nb * RPB * PW + KX1]);
            }
            NB_SB;
            if (!first) main4(ah2, bh2);                           // tap 2 of the previous step
            pieces(P2{}, P3{}, tn, (t + 1) & 1); NB_SB;             // second half of step t+1's pieces
            if (!odd && !first) corr4(al2, bl2);                   // tap-2 corrections of the two previous steps
            pieces(P3{}, P4{}, tn, (t + 1) & 1); NB_SB;
            main4(ah0, bh0);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                ah2[mb] = wb[a_base + 2 * 4 * CO_WG + mb * 32];
                if (odd) { NB_Q(al2[mb], 1, wb[a_base + 2 * 4 * CO_WG + CO_WG + mb * 32]); } else { NB_Q(al2[mb], 0, wb[a_base + 2 * 4 * CO_WG + CO_WG + mb * 32]); }
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                bh2[nb] = xb[b_base + nb * RPB * PW + KX2];
                if (odd) { NB_Q(bl2[nb], 1, xb[b_base + XPL + nb * RPB * PW + KX2]); } else { NB_Q(bl2[nb], 0, xb[b_base + XPL + nb * RPB * PW + KX2]); }
            }
            NB_SB;
            // the step's last fragment reads are in registers for every wave: its buffers take step t+2 (past the end: a harmless
            // re-copy keeps the piece counts uniform)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            pieces(P0{}, P1{}, tnn, t & 1); NB_SB;                  // first half of step t+2's pieces
            main4(ah1, bh1);
            pieces(P1{}, P2{}, tnn, t & 1); NB_SB;
            corr4(al01, bl01);
        };
        for (int t = 0; t < T; t += 2) {
            step(t, 0, t == 0);
            if (t + 1 < T) step(t + 1, 1, false);
        }
        NB_SB;
        main4(ah2, bh2);                                           // the last step's tap 2
        if (T & 1) {                                               // odd number of steps: the last tuple holds one tap only
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 4; r < 8; ++r) al2[mb][r] = 0;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 4; r < 8; ++r) bl2[nb][r] = 0;
        }
        corr4(al2, bl2);
#undef NB_Q
#undef NB_SB
    } else
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my share of step t has landed
        __builtin_amdgcn_s_barrier();                             // ... everybody's has, and step t-1 is fully consumed
        issue(t + 1 < T ? t + 1 : T - 1, (t + 1) & 1);            // past the end: a harmless re-copy keeps the flow uniform
        __builtin_amdgcn_sched_barrier(0);
        const h8* xb = xbuf + (t & 1) * 4 * XPL;
        const h8* wb = wbuf + (t & 1) * WSLOTS;
        h8 ah[2][MB], al[2][MB], bh[2][NBW], bl[2][NBW];
        auto fetch = [&](int kx, h8 (&fah)[MB], h8 (&fal)[MB], h8 (&fbh)[NBW], h8 (&fbl)[NBW]) {
            const int ko = kx == 0 ? KX0 : (kx == 1 ? KX1 : KX2);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                fah[mb] = wb[a_base + kx * 4 * CO_WG + mb * 32];
                fal[mb] = wb[a_base + kx * 4 * CO_WG + CO_WG + mb * 32];
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                fbh[nb] = xb[b_base + nb * RPB * PW + ko];
                fbl[nb] = xb[b_base + XPL + nb * RPB * PW + ko];
            }
        };
        fetch(0, ah[0], al[0], bh[0], bl[0]);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int cu = kx & 1;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cu][0], bh[cu][0], acc[0][0], 0, 0, 0);
            if (kx + 1 < 3) fetch(kx + 1, ah[cu ^ 1], al[cu ^ 1], bh[cu ^ 1], bl[cu ^ 1]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    if (mb + nb > 0)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cu][mb], bh[cu][nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cu][mb], bl[cu][nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cu][mb], bh[cu][nb], acc[mb][nb], 0, 0, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (kx + 1 < 3) __builtin_amdgcn_sched_group_barrier(0x100, 2 * MB + 2 * NBW, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * MB * NBW - 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // staging buffers are dead from here on
    NB_TSTAMP(2);

    // ---- epilogue: + bias, LeakyReLU; D[row = c_out, col = pixel]; tile pixel index = block * 32 + l31 ----
    if constexpr (DIRECT) {
        if (p.out_f8) nb_set_fp16_ovfl();                         // f8 hand-off: the fp8 (and f16) conversions saturate
        const float nz0[NBW] = {};
        const size_t OHW8 = (size_t)p.hout * p.wout * 8;
        nb_up1_handoff_epilogue<MB, NBW>(nb_handoff_args(p.yh2 + (size_t)p.cg0 * 2 * OHW8, p.c8_total, p.c_out, p.hout, p.wout, p.out_f8, 0, p.slope, 1.f, -1.f),
                                         acc, nz0, s_one, s_bias, s_osc, wm * 64, wn * NBW, co0, n, y0, x0, lh, l31);
        NB_TSTAMP(3);
    } else if (OUT == 0) {
        constexpr int PIX_WG = 256;
        float* ot = reinterpret_cast<float*>(smem_enc);           // [CO_WG][PIX_WG]
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int co = co0 + col;
                    float v = 0.f;
                    if (co < p.c_out) {
                        v = nb_lrelu(acc[mb][nb][r] + s_bias[col], p.slope);
                        if (p.oscale) v *= s_osc[col];
                    }
                    ot[col * PIX_WG + (wn * NBW + nb) * 32 + l31] = v;
                }
        __syncthreads();
        NB_TSTAMP(3);
        for (int e = tid; e < CO_WG * (PIX_WG / 4); e += 512) {
            const int col = e / (PIX_WG / 4), q4 = e - col * (PIX_WG / 4);
            const int co = co0 + col;
            if (co < p.c_out) {
                const int pix = q4 * 4, row = pix >> LW, cx = pix & (WT - 1);
                const f32x4 v = *reinterpret_cast<const f32x4*>(ot + col * PIX_WG + pix);
                *reinterpret_cast<f32x4*>(p.y32 + (((size_t)n * p.c_out + co) * p.hout + y0 + row) * p.wout + x0 + cx) = v;
            }
        }
    } else {
        constexpr int CP = CO_WG + 8;                             // padded channel pitch (halves): 2-way conflicts at most
        _Float16* sh = reinterpret_cast<_Float16*>(smem_enc);     // [256 pixels][CP] hi, then lo
        _Float16* sl = sh + 256 * CP;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int pix = (wn * NBW + nb) * 32 + l31;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = wm * 64 + mb * 32 + 8 * g + 4 * lh;
                    h4 vh, vl;
                    float vv[4], xl[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = co0 + col + j;
                        float v = 0.f;
                        if (co < p.c_out) {
                            v = nb_lrelu(acc[mb][nb][4 * g + j] + s_bias[col + j], p.slope);
                            if (p.oscale) v *= s_osc[col + j];
                        }
                        const _Float16 hi = (_Float16)v;
                        vv[j] = v; xl[j] = v - (float)hi;
                        vh[j] = hi; vl[j] = (_Float16)xl[j];
                    }
                    *reinterpret_cast<h4*>(sh + pix * CP + col) = vh;
                    if (p.out_f8) {
                        // lo image as bytes: per 16-channel chunk 16 x fp8(xl 2^9) then 16 x fp8(v/4) (see nb_modconv_h3.hip)
                        unsigned char* sb = reinterpret_cast<unsigned char*>(sl) + (size_t)pix * (CP * 2) + (col >> 4) * 32 + (col & 15);
                        *reinterpret_cast<unsigned*>(sb) = nb_enc_pk4_fp8(xl[0] * 512.f, xl[1] * 512.f, xl[2] * 512.f, xl[3] * 512.f);
                        *reinterpret_cast<unsigned*>(sb + 16) = nb_enc_pk4_fp8(vv[0] * 0.25f, vv[1] * 0.25f, vv[2] * 0.25f, vv[3] * 0.25f);
                    } else {
                        *reinterpret_cast<h4*>(sl + pix * CP + col) = vl;
                    }
                }
        }
        __syncthreads();
        NB_TSTAMP(3);
        const int c8o = (p.c_out + 7) / 8;
        const size_t OHW8 = (size_t)p.hout * p.wout * 8;
        _Float16* yn = p.yh2 + ((size_t)n * p.c8_total + p.cg0) * 2 * OHW8;
        for (int e = tid; e < (CO_WG / 8) * 2 * 256; e += 512) {  // [cg 16][hl 2][pixel 256]
            const int pix = e & 255, hl = (e >> 8) & 1, cgl = e >> 9;
            const int cg = co0 / 8 + cgl;
            if (cg < c8o) {
                const int row = pix >> LW, cx = pix & (WT - 1);
                const h8 v = *reinterpret_cast<const h8*>((hl ? sl : sh) + pix * CP + cgl * 8);
                *reinterpret_cast<h8*>(yn + (size_t)(cg * 2 + hl) * OHW8 + ((size_t)(y0 + row) * p.wout + x0 + cx) * 8) = v;
            }
        }
    }
    NB_TSTAMP(4);
}

template <int STRIDE, int LW, int OUT, bool F8 = false, bool FUSED = false>
static int launch_enc_conv(EncConvParams p, int n, hipStream_t st) {
    constexpr int WT = 1 << LW, TH = 8 * (32 / WT), PW = WT + 2, SLOTS = (F8 && STRIDE == 2) ? (TH + 1) * 2 * PW : STRIDE * TH * PW;
    constexpr int XPL = FUSED ? SLOTS : ((SLOTS + 63) / 64) * 64;
    constexpr size_t fused = FUSED ? (size_t)2 * (2 * WT + 8) * 108 + (size_t)4 * 2 * 2 * 64 * 16 : 0;     // image strips + stem weight fragments
    constexpr size_t staging = (size_t)(2 * 4 * XPL + 2 * 12 * 128) * 16 + fused;
    constexpr size_t epi = OUT == 0 ? (size_t)128 * 256 * 4 : (LW == 5 ? 0 : (size_t)2 * 256 * (128 + 8) * 2);       // (32-wide hand-off tiles: no staging)
    constexpr size_t lds = staging > epi ? staging : epi;
    static_assert(lds + 3 * 128 * 4 + 64 * 4 <= 160 * 1024, "LDS budget (dynamic staging + the hand-off epilogue's static tables + the stem's bias)");
    p.tiles_x = p.wout / WT; p.tiles_y = p.hout / TH; p.slices = (p.c_out + 127) / 128;
    p.tstamps = (g_enc_tstamps && (long long)p.tiles_x * p.tiles_y * p.slices * n <= g_enc_tstamps_cap) ? g_enc_tstamps : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)enc_conv3x3_h3_kernel<STRIDE, LW, OUT, F8, FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      // (+ static tables: the sum must fit 160 KB)
        attr_set = true;
    }
    hipLaunchKernelGGL((enc_conv3x3_h3_kernel<STRIDE, LW, OUT, F8, FUSED>), dim3(p.tiles_x * p.tiles_y * p.slices, n), dim3(512), lds, st, p);
    NB_CHECK_LAUNCH("enc_conv3x3_h3");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// The same layer for launches the 128 c_out x 256 pixel tiles above cannot fill the chip with (one interactive stroke =
// ONE tile: 4 - 32 workgroups that each walk up to 48 double-buffered steps, i.e. ~2 us of exposed DMA latency per
// step).  Structure of modconv3x3_up1_small_h3_kernel (nb_modconv_small.hip): 4 waves = one tile of 32 c_out x 32
// output positions, the waves SPLIT K (wave w owns the 16-channel chunks w, w+4, ...) and meet only in the final LDS
// reduction; weight fragments go from global memory straight into registers, a whole chunk ahead; the chunk's H2
// activations (tile + halo, reflect padding = address arithmetic) are gathered by LDS-DMA into a wave-private,
// double-buffered region - for stride 2 simply the (2 rows + 1) x (2 cols + 1) input window, tap (ky, kx) of output
// (ty, tx) reads slot (2 ty + ky, 2 tx + kx).
// ------------------------------------------------------------------------------------------------
struct EncSmallParams {
    const _Float16* x; const _Float16* wts; const float* bias; float* y32; _Float16* yh2;
    int c8, nchunks, c_out, co_ld, hin, win, hout, wout, rows, cols, tiles_x, slices;
    float slope;
};

template <int STRIDE, int OUT>
__global__ __launch_bounds__(256) void enc_conv3x3_small_h3_kernel(const EncSmallParams p) {
    constexpr int PP = STRIDE == 2 ? 4 : 2, NHP = PP * 64;         // 64-slot DMA pieces / slots per (cg, hi/lo) plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_es[];     // [wave 4][buf 2][plane 4][NHP] slots
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x, tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int y0 = tile_y * p.rows, x0 = tile_x * p.cols, co0 = slice * 32;
    const int HR = STRIDE * p.rows + (3 - STRIDE), HC = STRIDE * p.cols + (3 - STRIDE), NH = HR * HC;
    const size_t HW8 = (size_t)p.hin * p.win * 8;
    const _Float16* xn = p.x + (size_t)n * p.c8 * 2 * HW8;

    // gather descriptors: piece q of a plane, lane -> halo slot e = q*64 + lane -> reflected input pixel
    int xsrc[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        int e = q * 64 + lane;
        e = e < NH ? e : NH - 1;                                     // (tail lanes re-copy the last slot: never read)
        const int hy = e / HC, hx = e - hy * HC;
        // (positions of a ragged tile past the image reach further than the one reflected pixel: clamped, never stored)
        int iy = nb_reflect(STRIDE * y0 - 1 + hy, p.hin), ix = nb_reflect(STRIDE * x0 - 1 + hx, p.win);
        iy = iy < 0 ? 0 : (iy >= p.hin ? p.hin - 1 : iy);
        ix = ix < 0 ? 0 : (ix >= p.win ? p.win - 1 : ix);
        xsrc[q] = (iy * p.win + ix) * 8;
    }
    h8* mybuf = reinterpret_cast<h8*>(smem_es) + wv * (2 * 4 * NHP);
    auto issue_x = [&](int c, h8* buf) {
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                const _Float16* src = xn + (size_t)(4 * c + pl) * HW8 + xsrc[q];
                __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(buf + pl * NHP + q * 64), 16, 0, 0);
            }
    };
    const unsigned wstep = (unsigned)(p.co_ld * 8);
    const unsigned wl = (unsigned)((lh * 2 * p.co_ld + co0 + l31) * 8);
    auto load_w = [&](int c, h8 (&wa)[9][2]) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
                wa[tap][hl] = *reinterpret_cast<const h8*>(p.wts + (wl + (unsigned)((c * 9 + tap) * 4 + hl) * wstep));
    };
    // this lane's output position and the slot of its tap (0, 0)
    const int pty = l31 / p.cols, ptx = l31 - pty * p.cols;
    const int pbase = lh * 2 * NHP + (STRIDE * pty) * HC + STRIDE * ptx;     // plane (cg = lh, hi); lo = + NHP

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto mfma_chunk = [&](const h8 (&wa)[9][2], const h8* buf) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int off = pbase + (tap / 3) * HC + (tap % 3);
            const h8 bh = buf[off], bl = buf[off + NHP];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[tap][0], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[tap][0], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[tap][1], bh, acc, 0, 0, 0);
        }
    };
    h8 wa0[9][2], wa1[9][2];
    const int NC = p.nchunks;
    int c = wv;
    if (c < NC) { issue_x(c, mybuf); load_w(c, wa0); }
    while (c < NC) {
        int cn = c + 4;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // chunk c: slots in LDS, fragments in registers
        __builtin_amdgcn_wave_barrier();
        if (cn < NC) { issue_x(cn, mybuf + 4 * NHP); load_w(cn, wa1); }
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(wa0, mybuf);
        c = cn;
        if (c >= NC) break;
        cn = c + 4;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (cn < NC) { issue_x(cn, mybuf); load_w(cn, wa0); }
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(wa1, mybuf + 4 * NHP);
        c = cn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_es);                  // [wave 4][reg 16][lane 64]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wv * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    const int oy = y0 + pty, ox = x0 + ptx;
    if (oy >= p.hout || ox >= p.wout) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wv * 4 + j;
        const int co = co0 + j + 8 * wv + 4 * lh;
        const float sum = red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane] + red[(2 * 16 + r) * 64 + lane] + red[(3 * 16 + r) * 64 + lane];
        v[j] = co < p.c_out ? nb_lrelu(sum + p.bias[co], p.slope) : 0.f;
    }
    if (OUT == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + j + 8 * wv + 4 * lh;
            if (co < p.c_out) p.y32[(((size_t)n * p.c_out + co) * p.hout + oy) * p.wout + ox] = v[j];
        }
    } else {
        // H2: channel group cg = (co0 + 8 wv) / 8, this lane's 4 channels are halves 4*lh .. 4*lh+3 of the pixel's slot
        const int c8o = (p.c_out + 7) / 8, cg = (co0 >> 3) + wv;
        if (cg < c8o) {
            h4 vh, vl;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const _Float16 hi = (_Float16)v[j]; vh[j] = hi; vl[j] = (_Float16)(v[j] - (float)hi); }
            const size_t OHW8 = (size_t)p.hout * p.wout * 8;
            _Float16* dst = p.yh2 + ((size_t)n * c8o + cg) * 2 * OHW8 + ((size_t)oy * p.wout + ox) * 8 + 4 * lh;
            *reinterpret_cast<h4*>(dst) = vh;
            *reinterpret_cast<h4*>(dst + OHW8) = vl;
        }
    }
}

template <int STRIDE, int OUT>
static int launch_enc_small(EncSmallParams p, int n, hipStream_t st) {
    constexpr int NHP = (STRIDE == 2 ? 4 : 2) * 64;
    constexpr size_t lds = (size_t)4 * 2 * 4 * NHP * 16;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)enc_conv3x3_small_h3_kernel<STRIDE, OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int tiles_y = (p.hout + p.rows - 1) / p.rows;          // (ragged last tile row: masked at the store)
    hipLaunchKernelGGL((enc_conv3x3_small_h3_kernel<STRIDE, OUT>), dim3(p.tiles_x * tiles_y * p.slices, n), dim3(256), lds, st, p);
    NB_CHECK_LAUNCH("enc_conv3x3_small_h3");
    return NB_OK;
}

static int g_enc_small = -1;
// developer / test hook: -1 = automatic choice between the two tile forms of enc_conv3x3, 0 = large tiles, 1 = small tiles
extern "C" void nb_debug_set_enc_small(int mode) { g_enc_small = mode; }

static int nb_enc_conv3x3_impl(const void* x_h2, int c_in, const void* w_h3, const float* bias, float* y_f32, void* y_h2,
                               const float* oscale, int oscale_stride, int c8_total, int cg0, int out_fmt,
                               int n, int h_in, int w_in, int c_out, int stride, float slope, void* stream, int in_fmt = 0, int shift = 0) {
    NB_REQUIRE(in_fmt == 0 || (in_fmt == 1 && c_in % 16 == 0), "enc_conv3x3_h3: operand format must be 0 (H2) or 1 (f8, c_in %% 16 == 0)");
    NB_REQUIRE(x_h2 && w_h3 && bias && ((y_f32 != nullptr) != (y_h2 != nullptr)), "enc_conv3x3_h3: need x, w, bias and exactly one output");
    NB_REQUIRE(n >= 1 && n <= 65535 && c_in >= 1 && c_out >= 1 && (stride == 1 || stride == 2), "enc_conv3x3_h3: bad sizes");
    NB_REQUIRE(shift == 0 || (shift == 1 && stride == 2 && in_fmt == 0 && y_f32 && h_in % 2 == 1 && w_in % 2 == 1 && h_in >= 3 && w_in >= 3),
               "conv3x3_s2_valid_h3: needs H2 operands, an fp32 destination and an odd input size (got %dx%d)", h_in, w_in);
    NB_REQUIRE(shift || (h_in % stride == 0 && w_in % stride == 0 && h_in >= 2 && w_in >= 2), "enc_conv3x3_h3: bad input size %dx%d", h_in, w_in);
    const int ho = shift ? (h_in - 1) / 2 : h_in / stride, wo = shift ? (w_in - 1) / 2 : w_in / stride;
    // (16-wide tiles for every width that is a multiple of 16 but not of 32: 16 itself -- R = 128 -- and e.g. 48 = the inner layers at R = 384)
    const bool wide = wo % 32 == 0 && ho % 8 == 0, narrow = !wide && wo % 16 == 0 && ho % 16 == 0;
    // the 32-position split-K tiles take any output whose width is a power of two >= 4 (4- and 8-wide images: the encoder's
    // inner layers at patch sizes 32 and 64; rows that do not fill the last tile are masked)
    const bool pow2 = (wo & (wo - 1)) == 0 && wo >= 4;
    const bool small_ok = pow2 && c_in % 16 == 0 && in_fmt == 0 && !shift;
    NB_REQUIRE(wide || narrow || small_ok, "enc_conv3x3_h3: output must be a multiple of 32 wide (rows %% 8 == 0), a multiple of 16 wide (rows %% 16 == 0), or -- H2 "
               "operands, c_in %% 16 == 0 -- a power of two >= 4 wide; got %dx%d", ho, wo);
    NB_REQUIRE(y_f32 || c_out % 8 == 0, "enc_conv3x3_h3: H2 output needs c_out %% 8 == 0");
    NB_REQUIRE(((uintptr_t)x_h2 | (uintptr_t)w_h3 | (uintptr_t)y_f32 | (uintptr_t)y_h2) % 16 == 0, "enc_conv3x3_h3: pointers must be 16-byte aligned");
    EncConvParams p;
    p.x = (const _Float16*)x_h2; p.wts = (const _Float16*)w_h3; p.bias = bias; p.y32 = y_f32; p.yh2 = (_Float16*)y_h2;
    p.zeros = nb_zero_page_ptr();
    p.img = nullptr; p.stem_w = nullptr; p.stem_b = nullptr; p.preproc = 0;
    NB_REQUIRE(p.zeros, "enc_conv3x3_h3: could not allocate the zero page");
    p.c8 = (c_in + 7) / 8; p.nchunks = (c_in + 15) / 16; p.c_out = c_out; p.co_ld = (c_out + 127) / 128 * 128;
    p.hin = h_in; p.win = w_in; p.hout = ho; p.wout = wo; p.slope = slope;
    const bool handoff = !shift && (oscale != nullptr || c8_total > 0 || out_fmt != 0);
    p.shift = shift;
    NB_REQUIRE(!handoff || wide || narrow, "enc_conv3x3_h3: the hand-off into a consumer's operand tensor needs the large tiles (output %dx%d)", ho, wo);
    NB_REQUIRE(!handoff || (y_h2 && c8_total >= cg0 + (c_out + 7) / 8 && cg0 >= 0 && (out_fmt == 0 || (out_fmt == 1 && c_out % 16 == 0 && cg0 % 2 == 0))
                            && (!oscale || oscale_stride >= c_out)),
               "enc_conv3x3_h3: bad hand-off arguments (needs an H2 destination with room for the channel groups; f8: whole 16-channel chunks)");
    p.oscale = oscale; p.oscale_stride = oscale_stride; p.c8_total = handoff ? c8_total : (c_out + 7) / 8; p.cg0 = handoff ? cg0 : 0;
    p.out_f8 = out_fmt;
    hipStream_t st = (hipStream_t)stream;
    if (!handoff && in_fmt == 0 && !shift) {
        // under-filled launch (interactive strokes, small batches): the 32 x 32 split-K tiles instead.  Needs whole
        // 16-channel chunks and an output width that is a power of two >= 8 (32 positions = 32 / w rows).
        const long big_wgs = (long)n * (wo / (wide ? 32 : 16)) * (ho / (wide ? 8 : 16)) * ((c_out + 127) / 128);
        const int force = g_enc_small;                 // (test hook nb_debug_set_enc_small: -1 = this rule)
        // ... and layers with <= 32 output channels at any batch: the large tile has 128 c_out rows, three quarters of them empty then
        // (256 -> 32 and 32 -> 16 of a batch of 32 at R=256: 82 -> 39 us together)
        const bool small = !(wide || narrow) || (force >= 0 ? force != 0 : ((big_wgs <= 48 || c_out <= 32) && c_in >= 32));
        if (small && small_ok) {
            EncSmallParams q;
            q.x = p.x; q.wts = p.wts; q.bias = bias; q.y32 = y_f32; q.yh2 = (_Float16*)y_h2;
            q.c8 = p.c8; q.nchunks = p.nchunks; q.c_out = c_out; q.co_ld = p.co_ld; q.hin = h_in; q.win = w_in; q.hout = ho; q.wout = wo;
            q.cols = wo >= 32 ? 32 : wo; q.rows = 32 / q.cols; q.tiles_x = wo / q.cols; q.slices = (c_out + 31) / 32; q.slope = slope;
            if (stride == 2) return y_h2 ? launch_enc_small<2, 1>(q, n, st) : launch_enc_small<2, 0>(q, n, st);
            return y_h2 ? launch_enc_small<1, 1>(q, n, st) : launch_enc_small<1, 0>(q, n, st);
        }
    }
    const int key = (stride == 2 ? 4 : 0) | (wide ? 2 : 0) | (y_h2 ? 1 : 0);
    if (in_fmt == 1) {
        switch (key) {
            case 0: return launch_enc_conv<1, 4, 0, true>(p, n, st);
            case 1: return launch_enc_conv<1, 4, 1, true>(p, n, st);
            case 2: return launch_enc_conv<1, 5, 0, true>(p, n, st);
            case 3: return launch_enc_conv<1, 5, 1, true>(p, n, st);
            case 4: return launch_enc_conv<2, 4, 0, true>(p, n, st);
            case 5: return launch_enc_conv<2, 4, 1, true>(p, n, st);
            case 6: return launch_enc_conv<2, 5, 0, true>(p, n, st);
            default: return launch_enc_conv<2, 5, 1, true>(p, n, st);
        }
    }
    switch (key) {
        case 0: return launch_enc_conv<1, 4, 0>(p, n, st);
        case 1: return launch_enc_conv<1, 4, 1>(p, n, st);
        case 2: return launch_enc_conv<1, 5, 0>(p, n, st);
        case 3: return launch_enc_conv<1, 5, 1>(p, n, st);
        case 4: return launch_enc_conv<2, 4, 0>(p, n, st);
        case 5: return launch_enc_conv<2, 4, 1>(p, n, st);
        case 6: return launch_enc_conv<2, 5, 0>(p, n, st);
        default: return launch_enc_conv<2, 5, 1>(p, n, st);
    }
}

extern "C" int nb_enc_conv3x3_h3(const void* x_h2, int c_in, const void* w_h3, const float* bias, float* y_f32, void* y_h2,
                                 int n, int h_in, int w_in, int c_out, int stride, float slope, void* stream) {
    return nb_enc_conv3x3_impl(x_h2, c_in, w_h3, bias, y_f32, y_h2, nullptr, 0, 0, 0, 0, n, h_in, w_in, c_out, stride, slope, stream);
}

extern "C" int nb_enc_conv3x3_h3_handoff(const void* x_h2, int c_in, const void* w_h3, const float* bias, void* y_h2,
                                         const float* oscale, int oscale_stride, int c8_total, int cg0, int out_fmt,
                                         int n, int h_in, int w_in, int c_out, int stride, float slope, void* stream) {
    NB_REQUIRE(y_h2 && c8_total > 0, "enc_conv3x3_h3_handoff: needs the consumer's tensor and its channel-group count");
    return nb_enc_conv3x3_impl(x_h2, c_in, w_h3, bias, nullptr, y_h2, oscale, oscale_stride, c8_total, cg0, out_fmt, n, h_in, w_in,
                               c_out, stride, slope, stream);
}

// Stride-2 3x3 correlation WITHOUT padding on the same kernel (training path: the discriminator's down-sampling convolution
// after its FIR, and the input gradient of the generator's up=2 layers -- conv2d_resample.py:96-113, :124-147): x H2
// [n][c8][2][2ho+1][2wo+1][8], y fp32 [n][c_out][ho][wo] = oscale[n][co] * sum x[.., 2i+a, 2j+b] w[co][ci][a][b]; oscale may be NULL.
extern "C" int nb_conv3x3_s2_valid_h3(const void* x_h2, int c_in, const void* w_h3, const float* bias, const float* oscale, int oscale_stride,
                                      float* y_f32, int n, int h_in, int w_in, int c_out, void* stream) {
    NB_REQUIRE(!oscale || oscale_stride >= c_out, "conv3x3_s2_valid_h3: output scale rows shorter than c_out");
    return nb_enc_conv3x3_impl(x_h2, c_in, w_h3, bias, y_f32, nullptr, oscale, oscale_stride, 0, 0, 0, n, h_in, w_in, c_out, 2, 1.0f,
                               stream, 0, 1);
}

extern "C" int nb_enc_conv3x3_ex(const void* x, int c_in, const void* wts, const float* bias, float* y_f32, void* y_h2,
                                 const float* oscale, int oscale_stride, int c8_total, int cg0, int in_fmt, int out_fmt,
                                 int n, int h_in, int w_in, int c_out, int stride, float slope, void* stream) {
    return nb_enc_conv3x3_impl(x, c_in, wts, bias, y_f32, y_h2, oscale, oscale_stride, c8_total, cg0, out_fmt, n, h_in, w_in, c_out,
                               stride, slope, stream, in_fmt);
}

// Stem (1 -> 64, 7 x 7, reflect padding 3, LeakyReLU) + the first stride-2 stage (64 -> c_out, 3 x 3, reflect padding 1, LeakyReLU) in ONE
// launch (enc_conv3x3_h3_kernel<.., FUSED>): what nb_enc_stem7x7_f32_h2_ex(out_fmt 1) followed by nb_enc_conv3x3_ex(in_fmt 1, stride 2) computes,
// without the 64-channel full-resolution tensor between them.  x fp32 [n][1][h][w]; w50 / bias0: the stem's parameters as nb_enc_stem7x7 takes
// them; wts1: the stage's weights in "f8" format (c_in = 64); y_h2: its output [n][c_out / 8][2][h / 2][w / 2][8] in format out_fmt
// (0 = H2, 1 = f8).  Needs h %% 16 == 0, w %% 64 == 0, c_out %% 16 == 0.
extern "C" int nb_enc_stem_conv3x3_f8(const float* x, const float* w50, const float* bias0, int preproc, const void* wts1, const float* bias1,
                                      void* y_h2, int out_fmt, int n, int h, int w, int c_out, float slope, void* stream) {
    NB_REQUIRE(x && w50 && bias0 && wts1 && bias1 && y_h2, "enc_stem_conv3x3_f8: null pointer");
    NB_REQUIRE(out_fmt == 0 || out_fmt == 1, "enc_stem_conv3x3_f8: output format must be 0 (H2) or 1 (f8)");
    NB_REQUIRE(n >= 1 && n <= 65535 && h >= 16 && w >= 64 && h % 16 == 0 && w % 64 == 0, "enc_stem_conv3x3_f8: needs h %% 16 == 0 and w %% 64 == 0 (got %dx%d)", h, w);
    NB_REQUIRE(c_out >= 16 && c_out % 16 == 0, "enc_stem_conv3x3_f8: c_out %% 16 == 0 (got %d)", c_out);
    NB_REQUIRE(preproc >= 0 && preproc <= 2, "Unknown preprocessing type %d", preproc);
    NB_REQUIRE(slope >= 0.f && slope <= 1.f, "enc_stem_conv3x3_f8: leaky-ReLU slope must lie in [0, 1] (got %g)", slope);
    NB_REQUIRE(((uintptr_t)wts1 | (uintptr_t)y_h2) % 16 == 0, "enc_stem_conv3x3_f8: pointers must be 16-byte aligned");
    EncConvParams p;
    p.x = nullptr; p.wts = (const _Float16*)wts1; p.bias = bias1; p.y32 = nullptr; p.yh2 = (_Float16*)y_h2;
    p.zeros = nb_zero_page_ptr();
    NB_REQUIRE(p.zeros, "enc_stem_conv3x3_f8: could not allocate the zero page");
    p.c8 = 8; p.nchunks = 4; p.c_out = c_out; p.co_ld = (c_out + 127) / 128 * 128;
    p.hin = h; p.win = w; p.hout = h / 2; p.wout = w / 2; p.slope = slope;
    p.shift = 0; p.oscale = nullptr; p.oscale_stride = 0; p.c8_total = c_out / 8; p.cg0 = 0; p.out_f8 = out_fmt;
    p.img = x; p.stem_w = w50; p.stem_b = bias0; p.preproc = preproc;
    return launch_enc_conv<2, 5, 1, true, true>(p, n, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// bilinear x2 upsampling, align_corners=True (nn.Upsample in ScaleUp, simple_autoencoder.py:106-121):
// fp32 NCHW [n,c,h,w] -> H2 [n,c/8,2,2h,2w,8]
// ------------------------------------------------------------------------------------------------
__global__ NB_NO_PACKED_F32 __launch_bounds__(256) void enc_upsample2x_h2_kernel(const float* __restrict__ x, _Float16* __restrict__ y,
                                                                int c, int h, int w, long long total, int out_f8) {
    const int oh = 2 * h, ow = 2 * w, c8 = c / 8;
    const float sy = (float)(h - 1) / (float)(oh - 1), sx = (float)(w - 1) / (float)(ow - 1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % ow);
        long long r = i / ow;
        const int oy = (int)(r % oh); r /= oh;
        const int cg = (int)(r % c8);
        const int n = (int)(r / c8);
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
        h8 vh, vl;
        float vv[8], xl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* xp = x + ((size_t)n * c + cg * 8 + j) * ((size_t)h * w);
            const float v = hy * (hx * xp[y0 * w + x0] + lx * xp[y0 * w + x1]) + ly * (hx * xp[y1 * w + x0] + lx * xp[y1 * w + x1]);
            const _Float16 hi = (_Float16)v;
            vv[j] = v; xl[j] = v - (float)hi;
            vh[j] = hi; vl[j] = (_Float16)xl[j];
        }
        const size_t OHW8 = (size_t)oh * ow * 8;
        _Float16* yp = y + ((size_t)(n * c8 + cg) * 2) * OHW8 + ((size_t)oy * ow + ox) * 8;
        *reinterpret_cast<h8*>(yp) = vh;
        if (out_f8) {
            // the 16-channel chunk's two lo slots: (even group, lo) = fp8(xl 2^9), (odd group, lo) = fp8(v/4); this group's 8
            // channels are bytes 8 (cg & 1) .. + 7 of both
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            _Float16* lo_xl = y + ((size_t)(n * c8 + (cg & ~1)) * 2 + 1) * OHW8 + ((size_t)oy * ow + ox) * 8 + (cg & 1) * 4;
            *reinterpret_cast<u32x2*>(lo_xl) = u32x2{nb_enc_pk4_fp8(xl[0] * 512.f, xl[1] * 512.f, xl[2] * 512.f, xl[3] * 512.f),
                                                     nb_enc_pk4_fp8(xl[4] * 512.f, xl[5] * 512.f, xl[6] * 512.f, xl[7] * 512.f)};
            *reinterpret_cast<u32x2*>(lo_xl + 2 * OHW8) = u32x2{nb_enc_pk4_fp8(vv[0] * 0.25f, vv[1] * 0.25f, vv[2] * 0.25f, vv[3] * 0.25f),
                                                                nb_enc_pk4_fp8(vv[4] * 0.25f, vv[5] * 0.25f, vv[6] * 0.25f, vv[7] * 0.25f)};
        } else {
            *reinterpret_cast<h8*>(yp + OHW8) = vl;
        }
    }
}

extern "C" int nb_enc_upsample2x_h2_ex(const float* x, void* y_h2, int out_fmt, int n, int c, int h, int w, void* stream);
extern "C" int nb_enc_upsample2x_h2(const float* x, void* y_h2, int n, int c, int h, int w, void* stream) {
    return nb_enc_upsample2x_h2_ex(x, y_h2, 0, n, c, h, w, stream);
}
extern "C" int nb_enc_upsample2x_h2_ex(const float* x, void* y_h2, int out_fmt, int n, int c, int h, int w, void* stream) {
    NB_REQUIRE(x && y_h2, "enc_upsample2x: null pointer");
    NB_REQUIRE(out_fmt == 0 || (out_fmt == 1 && c % 16 == 0), "enc_upsample2x: output format must be 0 (H2) or 1 (f8, c %% 16 == 0)");
    NB_REQUIRE(n >= 1 && c >= 8 && c % 8 == 0 && h >= 2 && w >= 2, "enc_upsample2x: bad sizes (c must be a multiple of 8)");
    NB_REQUIRE((uintptr_t)y_h2 % 16 == 0, "enc_upsample2x: output must be 16-byte aligned");
    const long long total = (long long)n * (c / 8) * 4 * h * w;
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(enc_upsample2x_h2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)y_h2, c, h, w, total, out_fmt);
    NB_CHECK_LAUNCH("enc_upsample2x");
    return NB_OK;
}
