// up = 2 split-f16 convolution, "wide" form for gfx950: 64 c_out x (12 x 16 quads) per workgroup, ONE wave per SIMD.
//
// Same math as modconv3x3_up2_h3_kernel (nb_modconv_h3.hip: the 4-phase transposed convolution of
// torch_utils/ops/conv2d_resample.py:124-142 + the fused polyphase FIR of upfirdn2d.cu:97-200, split-f16 / "f8" products),
// same per-output summation order -- the two kernels are bit-identical -- but a different tiling:
//
//   * 4 waves, each with the whole 512-entry register file of its SIMD: wave tile = 64 c_out x 64 positions x 4 output
//     phases = 256 accumulator registers (2 x 2 MFMA tiles per phase).  A fragment pair now feeds two MFMAs in both
//     directions: 52 ds_read_b128 per 56 MFMAs and chunk (the 32 c_out x 64 position wave tile of the 8-wave kernel: 34 per 28).
//   * all 64 c_out of a slice in one workgroup: the haloed input tile is staged once per 64 (not 32) output channels.
//   * staging: activations in a THREE-stage LDS ring (HBM latency: two chunks of flight time), the 36.9 KB of weights per chunk
//     in a TWO-stage ring (L2 hits; issued early in the interval) -- 3 x 18 240 + 2 x 36 864 = 128 448 B.
//   * software pipeline rotated by one MFMA group: the barrier of chunk c sits before its LAST group (taps 2, 0), whose operands
//     are in registers by then; behind the barrier the first operands of chunk c+1 are read under that group's 12 MFMAs, so no
//     chunk starts with an exposed LDS round trip (with one wave per SIMD nobody else would cover it).
#include "nb_h3_common.h"

namespace {
constexpr int TQH = 12, TQW = 16, NW = 4, NT = NW * 64;
constexpr int PH = TQH + 2, PW = TQW + 2, NPOS = PH * PW;            // 14 x 18 = 252 halo'd quad positions
constexpr int NBLK = (NPOS + 31) / 32, NBJ = 2;                      // 8 position blocks of 32, two per wave
constexpr int XR = TQH + 3, XS = TQW + 3, XPL = XR * XS;             // 15 x 19 input pixels = 285 slots per (cg, hi/lo) plane
constexpr int PP = (XPL + 63) / 64, NXP = 4 * PP;                    // 5 pieces per plane, 20 per chunk
constexpr int CO_WG = 64, WROWS = 36, WSLOTS = WROWS * CO_WG;        // weights of a chunk: [tap 9][cg 2][hl 2] rows x 64 c_out
constexpr int NWPC = WROWS / NW, NXPC = NXP / NW;                    // 9 weight + 5 activation pieces per wave and chunk
constexpr int ASTAGE = 4 * XPL, NSTA = 3, NSTW = 2;
constexpr int RING_SLOTS = NSTA * ASTAGE + NSTW * WSLOTS;            // 8 028 slots = 128 448 B
constexpr int NW2 = 4;                                               // weight pieces of chunk c+2 issued under the last group of chunk c
static_assert(NBLK == NBJ * NW && WROWS % NW == 0 && NXP % NW == 0, "piece and block counts must be uniform over the waves");
}

template <bool F8, int OUTM>
__global__ __launch_bounds__(NT) void modconv3x3_up2w_kernel(const H3Up2Params p) {
    static_assert(F8, "the wide kernel is built for the f8 operand format so far");
    NB_TSTAMP(0);
    if constexpr (OUTM == 2) nb_set_fp16_ovfl();
    nb_stagger(p.stagger_ticks, 256);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    h8* aring = reinterpret_cast<h8*>(smem_w);           // [NSTA][4 planes x XPL]
    h8* wring = aring + NSTA * ASTAGE;                   // [NSTW][36 rows x 64]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    const int H = p.h, W = p.w;
    int b = blockIdx.x;
    // XCD-aware order (see modconv3x3_up2_h3_kernel): the c_out slices of one input tile share an XCD's L2
    if (gridDim.x % 8 == 0 && !(p.dbg & 8)) b = (((b & 7) + blockIdx.y) & 7) * (gridDim.x >> 3) + (b >> 3);
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x; const int tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int I0 = tile_y * TQH, J0 = tile_x * TQW;
    const int co0 = slice * CO_WG;
    const size_t HW8 = (size_t)H * W * 8;
    const _Float16* xn = p.x + (size_t)n * p.c8 * 2 * HW8;

    __shared__ __attribute__((aligned(16))) float s_dco[CO_WG], s_bias[CO_WG], s_nst[CO_WG];
    __shared__ __attribute__((aligned(16))) float s_noise[2 * TQH * 2 * TQW];
    if (tid < CO_WG) {
        const int co = co0 + tid;
        s_dco[tid] = co < p.c_out ? p.dcoefs[(size_t)n * p.c_out + co] * p.gain : 0.f;
        s_bias[tid] = co < p.c_out ? p.bias[co] * p.gain : 0.f;
        s_nst[tid] = (p.yh2 && co < p.c_out) ? p.next_styles[(size_t)n * p.next_stride + co] : 0.f;
    }
    // ---- LDS-DMA pieces.  Weights: piece k of a wave = row 4 k + wv of the chunk's 36 (one row = 64 c_out x 16 B = 1 KiB,
    //      contiguous in memory): uniform base + lane offset.  Activations: piece i = 64 slots of plane (4 i + wv) / 5; per-lane
    //      source (the haloed tile's rows are 19 slots of an image row each), out-of-image slots read the zero page (their
    //      address does not move with the chunk: per-lane stride 0); lanes past the plane's 285 slots do not copy (uniform
    //      mask).  Every wave issues exactly 9 + 5 pieces per chunk: the waits below count them. ----
    const unsigned lds0 = (unsigned)(uintptr_t)NB_LDS_PTR(smem_w);
    const unsigned wlane = (unsigned)(((size_t)wv * p.co_ld + co0 + lane) * 16);
    const size_t wrow4 = (size_t)4 * p.co_ld * 16, wchunk = (size_t)WROWS * p.co_ld * 16;     // bytes
    // (wslot / aslot: first slot of the destination stage within the ring)
    auto issue_w = [&](auto kk, int c, int wslot) {
        constexpr int k = decltype(kk)::value;
        const char* base = reinterpret_cast<const char*>(p.wts) + (size_t)c * wchunk + (size_t)k * wrow4;
        nb_lds_dma16_s(base, wlane, lds0 + (unsigned)(wslot + (4 * k + wv) * CO_WG) * 16u);
    };
    const char* xsrc0[NXPC];
    unsigned xstr[NXPC];
    int xdst[NXPC];
    unsigned long long xmask[NXPC];
#pragma unroll
    for (int i = 0; i < NXPC; ++i) {
        const int q = i * NW + wv;
        const int pl = q / PP, part = q - pl * PP;
        const int e = part * 64 + lane;
        xdst[i] = pl * XPL + part * 64;
        xmask[i] = part == PP - 1 ? (1ull << (XPL - (PP - 1) * 64)) - 1 : ~0ull;
        xsrc0[i] = reinterpret_cast<const char*>(p.zeros);
        xstr[i] = 0;
        if (e < XPL) {
            const int r = e / XS, c = e - r * XS;
            const int gy = I0 - 1 + r, gx = J0 - 1 + c;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                xsrc0[i] = reinterpret_cast<const char*>(xn + (size_t)pl * HW8 + (size_t)(gy * W + gx) * 8);
                xstr[i] = (unsigned)(4 * HW8 * 2);
            }
        }
    }
    auto issue_x = [&](auto ii, int c, int aslot) {
        constexpr int i = decltype(ii)::value;
        nb_lds_dma16_m(xsrc0[i] + (size_t)c * xstr[i], lds0 + (unsigned)(aslot + xdst[i]) * 16u, xmask[i]);
    };

    // fragment offsets (16-byte slots).  B: position (r, c) of block (wv + 4 j): slot r XS + c of plane (lh 2 + hl);
    // A: row tap 4 + lh 2 + hl, column mb 32 + l31
    int boff[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        int pidx = (wv + NW * j) * 32 + l31;
        pidx = pidx < NPOS ? pidx : NPOS - 1;
        const int r = pidx / PW, c = pidx - r * PW;
        boff[j] = lh * 2 * XPL + r * XS + c;
    }
    const int aoff = lh * 2 * CO_WG + l31;            // + tap * 256 + hl * 64 + mb * 32

    f32x16 acc[2][NBJ][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < NBJ; ++j)
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][j][ph][r] = 0.f;

    // ---- prologue: chunk 0 (weights, activations), then what the first interval finds in flight: activations of chunk 1 and the
    //      first NW2 weight pieces of chunk 1 ----
    const int NC = p.nchunks;
    constexpr int WRING = NSTA * ASTAGE;              // first slot of the weight ring
    nb_static_for<0, NWPC>([&](auto k) { issue_w(k, 0, WRING); });
    nb_static_for<0, NXPC>([&](auto i) { issue_x(i, 0, 0); });
    if (NC > 1) {
        nb_static_for<0, NXPC>([&](auto i) { issue_x(i, 1, ASTAGE); });
        nb_static_for<0, NW2>([&](auto k) { issue_w(k, 1, WRING + WSLOTS); });
    }
    // the tile's noise values (epilogue operand), computed or fetched while chunk 0 is on its way
    for (int e = tid; e < 2 * TQH * 2 * TQW; e += NT) {
        const int r = e / (2 * TQW), c = e - r * (2 * TQW);
        const int oy = 2 * I0 + r, ox = 2 * J0 + c;
        float v = (p.noise && oy < 2 * H) ? p.noise[(size_t)n * p.noise_stride_n + (size_t)oy * (2 * W) + ox] : 0.f;
        if (p.nsrc.const_t && oy < 2 * H) {
            float np0, np1, wx0, wx1, wy0, wy1;
            int sx0, sy0;
            nb_noise_np(p.nsrc, n, np0, np1);
            nb_noise_axis(p.nsrc, oy, np0, sx0, wx0, wx1);
            nb_noise_axis(p.nsrc, ox, np1, sy0, wy0, wy1);
            v = nb_noise_value(p.nsrc, p.nsrc.strength[0], sx0, wx0, wx1, sy0, wy0, wy1);
        }
        s_noise[e] = v * p.gain;
    }
    if (NC > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXPC + NW2) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    NB_TSTAMP(1);

    // ---- fragment registers (loop-carried: a chunk's first operands are read under the previous chunk's last group).
    //      f16 operands: one h8 per fragment.  fp8 operands: the (first tap | second tap) halves of a block-scaled MFMA's 32-byte
    //      operand are read straight into ONE 8-register tuple each -- a lo fragment that served in two different tuples would
    //      have to be copied by v_mov (4 per copy: more issue slots than the second ds_read) ----
    h8 ah_a[2][2], ah_n[2][2], ah_m[2];               // A hi: sets a / n [tap of the pair][mb], m = the lone tap 4
    i32x8 al_a[2], al_n[2], al_m[2];                  // A lo tuples [mb]; al_m: (tap 4 | zeros)
    h8 bh0[NBJ], bh1[NBJ], bh2[NBJ], bh3[NBJ];        // B hi at input offsets 0, 1, XS, XS + 1
    i32x8 bl01[NBJ], bl02[NBJ], bl23[NBJ];            // B lo tuples (offset 0 | 1), (0 | XS), (XS | XS + 1)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        ah_m[i] = h8{}; al_a[i] = i32x8{}; al_n[i] = i32x8{}; al_m[i] = i32x8{};
        bh0[i] = h8{}; bh1[i] = h8{}; bh2[i] = h8{}; bh3[i] = h8{}; bl01[i] = i32x8{}; bl02[i] = i32x8{}; bl23[i] = i32x8{};
#pragma unroll
        for (int j = 0; j < 2; ++j) { ah_a[i][j] = h8{}; ah_n[i][j] = h8{}; }
    }
    const int sa_ = lh ? 116 : 127, sb_ = lh ? 129 : 118;      // E8M0 block scales (see modconv3x3_up1_h3_kernel)

#define NB_FENCE() __builtin_amdgcn_sched_barrier(0)
    // tuple halves: t.lo4 = *p (first tap), t.hi4 = *p (second tap)
    auto set_lo = [](i32x8& t, const h8& v) { const i32x4 x = __builtin_bit_cast(i32x4, v); t[0] = x[0]; t[1] = x[1]; t[2] = x[2]; t[3] = x[3]; };
    auto set_hi = [](i32x8& t, const h8& v) { const i32x4 x = __builtin_bit_cast(i32x4, v); t[4] = x[0]; t[5] = x[1]; t[6] = x[2]; t[7] = x[3]; };
    // one chunk.  NBE = position blocks this wave multiplies (blocks wholly below the image are skipped);
    // MODE 2: steady state (c + 2 < NC), 1: last but one chunk, 0: last chunk.
    // sa / sw: this chunk's stages; san / swn: the next chunk's; w1 / a2 / w2: first slots of the stages the DMA pieces of chunk
    // c + 1 (weights), c + 2 (activations) and c + 2 (weights, behind the barrier) go to
    auto chunk = [&](auto mode_, auto nbe_, int c, const h8* sa, const h8* sw, const h8* san, const h8* swn, int w1, int a2, int w2) {
        constexpr int MODE = decltype(mode_)::value, NBE = decltype(nbe_)::value;
        // single fragment reads.  A: hi of tap `tap` into ah[ti][mb]; lo into the first / second half of the tuple al[mb]
        auto rAh = [&](h8 (&ah)[2][2], int ti, int mb, int tap, const h8* w) { ah[ti][mb] = w[aoff + tap * 256 + mb * 32]; };
        auto rAl = [&](i32x8 (&al)[2], int half, int mb, int tap, const h8* w) {
            if (half) set_hi(al[mb], w[aoff + tap * 256 + 64 + mb * 32]); else set_lo(al[mb], w[aoff + tap * 256 + 64 + mb * 32]);
        };
        // B: hi at slot offset `del` into bh[j]; lo into a half of bl[j]  (blocks this wave does not multiply: nothing)
        auto rBh = [&](h8 (&bh)[NBJ], auto j_, int del, const h8* a) { constexpr int j = decltype(j_)::value; if constexpr (j < NBE) bh[j] = a[boff[j] + del]; };
        auto rBl = [&](i32x8 (&bl)[NBJ], int half, auto j_, int del, const h8* a) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j < NBE) { if (half) set_hi(bl[j], a[boff[j] + XPL + del]); else set_lo(bl[j], a[boff[j] + XPL + del]); }
        };
        // DMA slot s of the interval (5 groups x 4 slots, counted from the last group of the previous chunk): slots 0 .. 8 carry the
        // weight pieces of the NEXT chunk (0 .. NW2-1 under the previous chunk's last group), slots 10, 12, .. 18 the activation
        // pieces of the next but one
        auto dma = [&](auto s_) {
            constexpr int s = decltype(s_)::value;
#ifndef NB_ABL_NODMA
            if constexpr (s < NW2) {
                if constexpr (MODE == 2) issue_w(std::integral_constant<int, s>{}, c + 2, w2);
            } else if constexpr (s < NWPC) {
                if constexpr (MODE >= 1) issue_w(std::integral_constant<int, s>{}, c + 1, w1);
            } else if constexpr (s >= 10 && (s - 10) % 2 == 0 && (s - 10) / 2 < NXPC) {
                if constexpr (MODE == 2) issue_x(std::integral_constant<int, (s - 10) / 2>{}, c + 2, a2);
            }
#endif
        };
        using J0_ = std::integral_constant<int, 0>; using J1_ = std::integral_constant<int, 1>;
        // One group of a tap pair = 4 x (f16, f16, fp8) MFMAs on accumulator (mb, block) = (t >> 1, t & 1) of phase ph.  Behind
        // every MFMA a FILLER: gap g = 3 t + position.  One wave per SIMD: nobody else covers a bubble, so the fragment reads of the
        // NEXT group are dealt out one or two per gap instead of in a burst between the groups (a burst of 8-16 ds_read_b128 takes
        // longer to issue than the last MFMA of the group runs), and an LDS-DMA piece (~70 cycles of issue) goes behind an fp8 MFMA
        // (64 cycles).  A scheduling fence after every statement: program order is issue order.
        auto group_pair = [&](auto ph_, h8 (&ah)[2][2], i32x8 (&al)[2], h8 (&Ba)[NBJ], h8 (&Bb)[NBJ], i32x8 (&bl)[NBJ], auto&& filler) {
            constexpr int ph = decltype(ph_)::value;
            nb_static_for<0, 4>([&](auto t_) {
                constexpr int t = decltype(t_)::value;
                constexpr int mb = t >> 1, j = t & 1;
                f32x16& a_ = acc[mb][j][ph];
                if constexpr (j < NBE) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][mb], Ba[j], a_, 0, 0, 0);
                NB_FENCE(); filler(std::integral_constant<int, 3 * t>{}); NB_FENCE();
                if constexpr (j < NBE) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][mb], Bb[j], a_, 0, 0, 0);
                NB_FENCE(); filler(std::integral_constant<int, 3 * t + 1>{}); NB_FENCE();
                if constexpr (j < NBE) a_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(al[mb], bl[j], a_, 0, 0, 0, sa_, 0, sb_);
                NB_FENCE(); filler(std::integral_constant<int, 3 * t + 2>{}); NB_FENCE();
            });
        };
        // the 8 A fragments of a tap pair (t0, t1) as fillers 0 .. 7: (hi t0, hi t1, lo t0, lo t1) of mb 0, then of mb 1
        auto rA_pair = [&](auto i_, h8 (&ah)[2][2], i32x8 (&al)[2], int t0, int t1, const h8* w) {
            constexpr int i = decltype(i_)::value;
            constexpr int mb = i >> 2, k = i & 3;
            if constexpr (k == 0) rAh(ah, 0, mb, t0, w);
            else if constexpr (k == 1) rAh(ah, 1, mb, t1, w);
            else if constexpr (k == 2) rAl(al, 0, mb, t0, w);
            else rAl(al, 1, mb, t1, w);
        };
        NB_FENCE();
        // G0: taps 8, 6 -> phase 0 (operands in the a set, bh0, bh1, bl01).  Fillers: the A fragments of G1; weight pieces 4 .. 7
        group_pair(std::integral_constant<int, 0>{}, ah_a, al_a, bh0, bh1, bl01, [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g % 3 == 2) dma(std::integral_constant<int, 4 + g / 3>{});
            else rA_pair(std::integral_constant<int, (g / 3) * 2 + g % 3>{}, ah_n, al_n, 5, 3, sw);
        });
        // G1: taps 5, 3 -> phase 2.  Fillers: tap 4 (G2), the row-below fragments (G3); weight piece 8, activation piece 0
        group_pair(std::integral_constant<int, 2>{}, ah_n, al_n, bh0, bh1, bl01, [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g == 0) ah_m[0] = sw[aoff + 4 * 256];
            else if constexpr (g == 1) set_lo(al_m[0], sw[aoff + 4 * 256 + 64]);
            else if constexpr (g == 2) dma(std::integral_constant<int, 8>{});
            else if constexpr (g == 3) ah_m[1] = sw[aoff + 4 * 256 + 32];
            else if constexpr (g == 4) set_lo(al_m[1], sw[aoff + 4 * 256 + 64 + 32]);
            else if constexpr (g == 5) { rBh(bh2, J0_{}, XS, sa); rBh(bh2, J1_{}, XS, sa); }
            else if constexpr (g == 6) rBl(bl02, 0, J0_{}, 0, sa);
            else if constexpr (g == 7) rBl(bl02, 1, J0_{}, XS, sa);
            else if constexpr (g == 8) dma(std::integral_constant<int, 10>{});
            else if constexpr (g == 9) rBl(bl02, 0, J1_{}, 0, sa);
            else if constexpr (g == 10) rBl(bl02, 1, J1_{}, XS, sa);
        });
        // G2: tap 4 -> phase 3: 4 x (f16, fp8).  A = (tap 4 | zeros), so the second half of B may hold anything finite (bl01 holds
        // offset 1's bytes there): every product of that half is an exact zero, as with the zero-filled B half of
        // modconv3x3_up2_h3_kernel.  Fillers: the A fragments of G3, the diagonal fragments (G4); activation pieces 1, 2
        nb_static_for<0, 4>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            constexpr int mb = t >> 1, j = t & 1;
            f32x16& a_ = acc[mb][j][3];
            if constexpr (j < NBE) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_m[mb], bh0[j], a_, 0, 0, 0);
            NB_FENCE();
            rA_pair(std::integral_constant<int, 2 * t>{}, ah_a, al_a, 7, 1, sw);
            rA_pair(std::integral_constant<int, 2 * t + 1>{}, ah_a, al_a, 7, 1, sw);
            NB_FENCE();
            if constexpr (j < NBE) a_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(al_m[mb], bl01[j], a_, 0, 0, 0, sa_, 0, sb_);
            NB_FENCE();
            if constexpr (t == 0) dma(std::integral_constant<int, 12>{});
            else if constexpr (t == 1) { rBh(bh3, J0_{}, XS + 1, sa); rBl(bl23, 0, J0_{}, XS, sa); rBl(bl23, 1, J0_{}, XS + 1, sa); }
            else if constexpr (t == 2) dma(std::integral_constant<int, 14>{});
            else { rBh(bh3, J1_{}, XS + 1, sa); rBl(bl23, 0, J1_{}, XS, sa); rBl(bl23, 1, J1_{}, XS + 1, sa); }
            NB_FENCE();
        });
        // G3: taps 7, 1 -> phase 1.  Fillers: the A fragments of G4; activation pieces 3, 4
        group_pair(std::integral_constant<int, 1>{}, ah_a, al_a, bh0, bh2, bl02, [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g % 3 == 2) dma(std::integral_constant<int, 16 + g / 3>{});
            else rA_pair(std::integral_constant<int, (g / 3) * 2 + g % 3>{}, ah_n, al_n, 2, 0, sw);
        });
        // everything of this chunk's stages has been read (G4's operands are on their way: waited for here); the next chunk has
        // landed: all but the NXPC youngest pieces (the activations of chunk c + 2)
        if constexpr (MODE == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NXPC) : "memory");
        else if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        NB_FENCE();
        // G4: taps 2, 0 -> phase 0.  Fillers: the NEXT chunk's first operands (two per f16 gap, in the order G0 takes them);
        // weight pieces 0 .. 3 of the chunk after next
        group_pair(std::integral_constant<int, 0>{}, ah_n, al_n, bh2, bh3, bl23, [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g % 3 == 2) dma(std::integral_constant<int, g / 3>{});
            else if constexpr (MODE >= 1) {
                if constexpr (g == 0) { rAh(ah_a, 0, 0, 8, swn); rBh(bh0, J0_{}, 0, san); }
                else if constexpr (g == 1) { rAh(ah_a, 1, 0, 6, swn); rBh(bh1, J0_{}, 1, san); }
                else if constexpr (g == 3) { rAl(al_a, 0, 0, 8, swn); rAl(al_a, 1, 0, 6, swn); }
                else if constexpr (g == 4) { rBl(bl01, 0, J0_{}, 0, san); rBl(bl01, 1, J0_{}, 1, san); }
                else if constexpr (g == 6) { rBh(bh0, J1_{}, 0, san); rBh(bh1, J1_{}, 1, san); }
                else if constexpr (g == 7) { rBl(bl01, 0, J1_{}, 0, san); rBl(bl01, 1, J1_{}, 1, san); }
                else if constexpr (g == 9) { rAh(ah_a, 0, 1, 8, swn); rAh(ah_a, 1, 1, 6, swn); }
                else if constexpr (g == 10) { rAl(al_a, 0, 1, 8, swn); rAl(al_a, 1, 1, 6, swn); }
            }
        });
        NB_FENCE();
    };
    auto kloop = [&](auto nbe) {
        constexpr int NBE = decltype(nbe)::value;
        int sa_i = 0, sw_i = 0;                       // stages of chunk c: c % 3, c % 2
        // the first chunk's first operands
        {
            const h8* sw = wring; const h8* sa = aring;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                ah_a[0][mb] = sw[aoff + 8 * 256 + mb * 32]; ah_a[1][mb] = sw[aoff + 6 * 256 + mb * 32];
                set_lo(al_a[mb], sw[aoff + 8 * 256 + 64 + mb * 32]); set_hi(al_a[mb], sw[aoff + 6 * 256 + 64 + mb * 32]);
            }
#pragma unroll
            for (int j = 0; j < NBE; ++j) {
                bh0[j] = sa[boff[j]]; bh1[j] = sa[boff[j] + 1];
                set_lo(bl01[j], sa[boff[j] + XPL]); set_hi(bl01[j], sa[boff[j] + XPL + 1]);
            }
        }
        int c = 0;
        for (; c + 2 < NC; ++c) {
            const int sa_n = sa_i == 2 ? 0 : sa_i + 1, sa_nn = sa_i == 0 ? 2 : sa_i - 1;
            chunk(std::integral_constant<int, 2>{}, nbe, c, aring + sa_i * ASTAGE, wring + sw_i * WSLOTS, aring + sa_n * ASTAGE, wring + (sw_i ^ 1) * WSLOTS,
                  WRING + (sw_i ^ 1) * WSLOTS, sa_nn * ASTAGE, WRING + sw_i * WSLOTS);
            sa_i = sa_n; sw_i ^= 1;
        }
        if (c + 1 < NC) {
            const int sa_n = sa_i == 2 ? 0 : sa_i + 1;
            chunk(std::integral_constant<int, 1>{}, nbe, c, aring + sa_i * ASTAGE, wring + sw_i * WSLOTS, aring + sa_n * ASTAGE, wring + (sw_i ^ 1) * WSLOTS,
                  WRING + (sw_i ^ 1) * WSLOTS, 0, 0);
            sa_i = sa_n; sw_i ^= 1; ++c;
        }
        chunk(std::integral_constant<int, 0>{}, nbe, c, aring + sa_i * ASTAGE, wring + sw_i * WSLOTS, nullptr, nullptr, 0, 0, 0);
    };
    const unsigned long long t_loop0 = p.tstamps ? __builtin_amdgcn_s_memtime() : 0;
    // position blocks of this wave that hold rows feeding stored pixels (a ragged last tile row: blocks wholly below the image
    // cost no MFMAs and no fragment reads; their accumulators stay zero and nothing stored reads them)
    const int nvalid_blk = (min(TQH, H - I0) + 2) * PW;
    int nbe_w = 0;
#pragma unroll
    for (int j = 0; j < NBJ; ++j) nbe_w += (wv + NW * j) * 32 < nvalid_blk;
    if (nbe_w == 2) kloop(std::integral_constant<int, 2>{});
    else if (nbe_w == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
#undef NB_FENCE
    if (p.tstamps && tid == 0) {
        unsigned long long* ts = p.tstamps + (size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8;
        ts[6] = 0; ts[7] = (__builtin_amdgcn_s_memtime() - t_loop0) << 32;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // drain before the staging LDS is reused
    __builtin_amdgcn_s_barrier();

    NB_TSTAMP(2);
    if (p.dbg & 4) { if (acc[0][0][0][0] == 123.456f) p.y[0] = 0.f; return; }      // ablation: main loop only
    // ---- epilogue: 4 rounds of 16 c_out, the arithmetic of modconv3x3_up2_h3_kernel's epilogue statement for statement (see the
    //      comments there): accumulators -> LDS as 4-channel slots y4[g][lh][phase][position]; one item = one quad x 4 channels:
    //      25 slot reads, separable polyphase FIR packed over channel pairs, activation, hi/lo split / fp8 conversion in registers,
    //      row trade between lanes l and l+32, direct slot stores. ----
    const int Wo = 2 * W, Ho = 2 * H;
    constexpr int nquads = TQH * TQW;
    constexpr int Y1P = NBLK * 32;
    f32x4* y4 = reinterpret_cast<f32x4*>(smem_w);
    const float clampv = p.clamp >= 0.f ? p.clamp : __builtin_inff();
    unsigned long long te_w = 0, te_f = 0, te0 = 0;
#pragma unroll
    for (int R = 0; R < 4; ++R) {
        if (p.tstamps) te0 = __builtin_amdgcn_s_memtime();
        if (R) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int gs = 0; gs < 2; ++gs) {
            const int r0 = (2 * (R & 1) + gs) * 4;
#pragma unroll
            for (int j = 0; j < NBJ; ++j) {
                const int pidx = (wv + NW * j) * 32 + l31;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const f32x16& a_ = acc[R >> 1][j][ph];
                    y4[((gs * 2 + lh) * 4 + ph) * Y1P + pidx] = f32x4{a_[r0], a_[r0 + 1], a_[r0 + 2], a_[r0 + 3]};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (p.tstamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); te_w += t_ - te0; te0 = t_; }
        auto quad_item = [&](const int wi) {
            const int hq = wi * 32 + l31;
            const int gs = hq / nquads, qd = hq - gs * nquads;
            const int ti = qd / TQW, tj = qd - ti * TQW;
            if (I0 + ti >= H) return;
            const int c4 = 16 * R + 8 * gs + 4 * lh;  // the lane's four channels within the slice
            const f32x4* ee = y4 + ((gs * 2 + lh) * 4) * Y1P + ti * PW + tj;
            const f32x4* eo = ee + 1 * Y1P;
            const f32x4* oe = ee + 2 * Y1P;
            const f32x4* oo = ee + 3 * Y1P;
            auto fir4 = [](f32x4 a, f32x4 b, f32x4 c, f32x4 d) {
                f32x4 q75, q25;
                q75 = 0.75f; q25 = 0.25f;
                return __builtin_elementwise_fma(q25, d, __builtin_elementwise_fma(q75, c, __builtin_elementwise_fma(q75, b, 0.25f * a)));
            };
            f32x4 ve[2][2], vo[3][2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c < 2) {
                    const f32x4 e0 = ee[c], e1 = ee[PW + c], o0 = oe[c], o1 = oe[PW + c], o2 = oe[2 * PW + c];
                    ve[c][0] = fir4(o0, e0, o1, e1); ve[c][1] = fir4(e0, o1, e1, o2);
                }
                const f32x4 e0 = eo[c], e1 = eo[PW + c], o0 = oo[c], o1 = oo[PW + c], o2 = oo[2 * PW + c];
                vo[c][0] = fir4(o0, e0, o1, e1); vo[c][1] = fir4(e0, o1, e1, o2);
            }
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(s_dco + c4), b4 = *reinterpret_cast<const f32x4*>(s_bias + c4);
            const int qi = I0 + ti, qj = J0 + tj;
            auto act4 = [&](f32x4 o, float nz) {
                f32x4 t = __builtin_elementwise_fma(o, d4, b4 + nz);
                const f32x4 ta = t * p.alpha;
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(t[i], ta[i], __builtin_inff()), -clampv, clampv);
                return t;
            };
            f32x4 v[2][2];                            // [dy][px]
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const f32x2 nz = *reinterpret_cast<const f32x2*>(s_noise + (2 * ti + dy) * (2 * TQW) + 2 * tj);
                // (the two noise values get registers of their own: see the note on v_pk_add_f32 op_sel in modconv3x3_up2_h3_kernel)
                float nz0 = nz[0], nz1 = nz[1];
                asm volatile("v_mov_b32 %0, %0" : "+v"(nz0));
                asm volatile("v_mov_b32 %0, %0" : "+v"(nz1));
                v[dy][0] = act4(fir4(vo[0][dy], ve[0][dy], vo[1][dy], ve[1][dy]), nz0);
                v[dy][1] = act4(fir4(ve[0][dy], vo[1][dy], ve[1][dy], vo[2][dy]), nz1);
            }
            if constexpr (OUTM == 0) {
                if (qi < H && !(p.dbg & 1)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = co0 + c4 + i;
                        if (co < p.c_out) {
                            float* dst = p.y + ((size_t)n * p.c_out + co) * ((size_t)Ho * Wo) + (size_t)(2 * qi) * Wo + 2 * qj;
                            *reinterpret_cast<f32x2*>(dst) = f32x2{v[0][0][i], v[0][1][i]};
                            *reinterpret_cast<f32x2*>(dst + Wo) = f32x2{v[1][0][i], v[1][1][i]};
                        }
                    }
                }
            } else {
                const f32x4 ns4 = *reinterpret_cast<const f32x4*>(s_nst + c4);
                unsigned hi[2][2][2], lo[2][2][2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        f32x4 w = v[dy][px] * ns4;
                        const h2 h01 = __builtin_convertvector(f32x2{w[0], w[1]}, h2), h23 = __builtin_convertvector(f32x2{w[2], w[3]}, h2);
                        const f32x4 xl = {nb_sub_f16(w[0], h01, false), nb_sub_f16(w[1], h01, true), nb_sub_f16(w[2], h23, false), nb_sub_f16(w[3], h23, true)};
                        hi[dy][px][0] = __builtin_bit_cast(unsigned, h01); hi[dy][px][1] = __builtin_bit_cast(unsigned, h23);
                        if constexpr (OUTM == 1) {
                            const h2 l01 = __builtin_convertvector(f32x2{xl[0], xl[1]}, h2), l23 = __builtin_convertvector(f32x2{xl[2], xl[3]}, h2);
                            lo[dy][px][0] = __builtin_bit_cast(unsigned, l01); lo[dy][px][1] = __builtin_bit_cast(unsigned, l23);
                        } else {
                            // (x 2^9 and x 2^-2 inside the conversions: nb_pk4_fp8_sat_scaled)
                            lo[dy][px][0] = nb_pk4_fp8_sat_scaled(xl[0], xl[1], xl[2], xl[3], 0x1p-9f);
                            lo[dy][px][1] = nb_pk4_fp8_sat_scaled(w[0], w[1], w[2], w[3], 4.f);
                        }
                    }
                unsigned ha[2][2], hb[2][2], la[2][2], lb[2][2];
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        ha[px][k] = hi[0][px][k]; hb[px][k] = hi[1][px][k]; la[px][k] = lo[0][px][k]; lb[px][k] = lo[1][px][k];
                        nb_swap32(ha[px][k], hb[px][k]);
                        nb_swap32(la[px][k], lb[px][k]);
                    }
                const int cg = co0 / 8 + 2 * R + gs;
                const int oy = 2 * qi + lh, ox = 2 * qj;
                if (qi < H && cg * 8 < p.c_out && !(p.dbg & 1)) {
                    const size_t OHW8 = (size_t)Ho * Wo * 8;
                    _Float16* yn = p.yh2 + ((size_t)n * p.c8_next + cg) * 2 * OHW8;
                    const size_t opix8 = ((size_t)oy * Wo + ox) * 8;
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        *reinterpret_cast<u32x4*>(yn + opix8 + px * 8) = u32x4{ha[px][0], ha[px][1], hb[px][0], hb[px][1]};
                        if constexpr (OUTM == 1) {
                            *reinterpret_cast<u32x4*>(yn + OHW8 + opix8 + px * 8) = u32x4{la[px][0], la[px][1], lb[px][0], lb[px][1]};
                        } else {
                            _Float16* lo_xl = p.yh2 + ((size_t)n * p.c8_next + (cg & ~1)) * 2 * OHW8 + OHW8 + opix8 + px * 8 + (cg & 1) * 4;
                            *reinterpret_cast<u32x2*>(lo_xl) = u32x2{la[px][0], lb[px][0]};
                            *reinterpret_cast<u32x2*>(lo_xl + 2 * OHW8) = u32x2{la[px][1], lb[px][1]};
                        }
                    }
                }
            }
        };
        constexpr int NWI = nquads * 2 / 32;          // 12 wave-iterations of 32 quads x both channel halves per round
        static_assert(nquads * 2 % 32 == 0 && NWI % NW == 0, "tile quads must fill whole waves");
#pragma unroll
        for (int k = 0; k < NWI / NW; ++k) quad_item(wv + k * NW);
        if (p.tstamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); te_f += t_ - te0; }
    }
    NB_TSTAMP(4);
    if (p.tstamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NB_TSTAMP(5);
        if (threadIdx.x == 0) p.tstamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + 3] = (te_w & 0x1fffff) | ((te_f & 0x1fffff) << 21);
    }
}

template <bool F8, int OUTM>
static int nb_up2w_launch1(const H3Up2Params& p, int n, void* stream) {
    constexpr size_t lds_ring = (size_t)RING_SLOTS * 16, lds_epi = (size_t)16 * NBLK * 32 * 16;
    constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_up2w_kernel<F8, OUTM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(p.tiles_x * p.tiles_y * p.slices, n);
    hipLaunchKernelGGL((modconv3x3_up2w_kernel<F8, OUTM>), grid, dim3(NT), lds, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("modconv3x3_up2w");
    return NB_OK;
}

// shapes the wide form takes: f8 operands, whole 64-channel output slices, 16-column tiles
bool nb_up2w_eligible(int in_fmt, int c_in, int c_out, int h, int w) {
    return in_fmt == 1 && c_in % 16 == 0 && c_out % CO_WG == 0 && w % TQW == 0 && h >= 8;
}
long nb_up2w_workgroups(int n, int c_out, int h, int w) { return (long)n * (w / TQW) * ((h + TQH - 1) / TQH) * (c_out / CO_WG); }

// p as filled in by nb_up2_h3_impl (nb_modconv_h3.hip); tiles and slices are set here
int nb_up2w_launch(H3Up2Params p, int n, int in_fmt, void* stream, unsigned long long* tstamps, int tstamps_cap) {
    NB_REQUIRE(nb_up2w_eligible(in_fmt, p.nchunks * 16, p.c_out, p.h, p.w) && p.co_ld % CO_WG == 0, "modconv3x3_up2w: shape not supported by the wide form");
    p.tiles_x = p.w / TQW;
    p.tiles_y = (p.h + TQH - 1) / TQH;
    p.slices = p.c_out / CO_WG;
    p.tstamps = (tstamps && (long long)p.tiles_x * p.tiles_y * p.slices * n <= tstamps_cap) ? tstamps : nullptr;
    const int outm = p.yh2 ? (p.out_f8 ? 2 : 1) : 0;
    return outm == 2 ? nb_up2w_launch1<true, 2>(p, n, stream) : outm == 1 ? nb_up2w_launch1<true, 1>(p, n, stream) : nb_up2w_launch1<true, 0>(p, n, stream);
}
