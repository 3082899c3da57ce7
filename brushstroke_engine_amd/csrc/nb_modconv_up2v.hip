// up = 2 split-f16 convolution ("f8" operands), 8-wave form with a SOFTWARE-PIPELINED K loop for gfx950.
//
// Same math, same tile (12 x 32 quads x 32 c_out, two position blocks and all four output phases per wave = 128 accumulator
// registers, two waves per SIMD), same three-stage LDS ring and the same epilogue as modconv3x3_up2_h3_kernel
// (nb_modconv_h3.hip) -- and the same per-output summation order: bit-identical results.  What differs is the K loop, which is the
// one measured on the one-wave-per-SIMD kernel of round 4 (nb_modconv_up2w.hip, removed in round 6 -- docs/perf_history.md: 97 % of its cycles were matrix work):
//   * one basic block per chunk: LDS-DMA pieces are issued from statements that set their lane mask themselves (an `if` around a
//     piece is a branch, and the compiler moves MFMAs across the resulting blocks);
//   * the pipeline is rotated by one MFMA group: the chunk's barrier stands before its LAST group (taps 2, 0), whose operands are
//     in registers by then, and the first operands of the next chunk are read under that group -- no chunk opens with both waves
//     of a SIMD waiting for the same burst of sixteen fragment reads;
//   * fragment reads and DMA pieces are dealt out between the MFMAs (one or two reads per gap, a piece behind an fp8 MFMA)
//     instead of in bursts between the groups, a scheduling fence after every statement: program order is issue order;
//   * the two 16-byte halves of a block-scaled fp8 operand are read straight into one 8-register tuple (no v_mov assembling).
#include "nb_h3_common.h"

#ifdef NB_ABL6_NOMFMA
#define NB_ABL6_F16 false
#define NB_ABL6_FP6 false
#elif defined(NB_ABL6_NOFP6)
#define NB_ABL6_F16 true
#define NB_ABL6_FP6 false
#else
#define NB_ABL6_F16 true
#define NB_ABL6_FP6 true
#endif
#ifdef NB_ABL6_NODMA
#define NB_ABL_NODMA 1
#endif
namespace {
constexpr int TQH = 12, TQW = 32, NW = 8, NT = NW * 64;
constexpr int PH = TQH + 2, PW = TQW + 2, NPOS = PH * PW;            // 14 x 34 = 476 halo'd quad positions
constexpr int NBLK = (NPOS + 31) / 32, NBJ = 2;                      // 15 position blocks of 32: wave w takes w and w + 8
constexpr int XR = TQH + 3, XS = TQW + 3, XPL = XR * XS;             // 15 x 35 input pixels = 525 slots per (cg, hi/lo) plane
constexpr int PP = (XPL + 63) / 64, NXP = 4 * PP;                    // 9 pieces per plane, 36 per chunk
constexpr int CO_WG = 32, WROWS = 36, WSLOTS = WROWS * CO_WG, NWP = WSLOTS / 64;     // 18 weight pieces per chunk
constexpr int NPC = (NXP + NWP + NW - 1) / NW;                       // 7 pieces per wave and chunk (54 dealt as one list, 2 re-copies)
constexpr int STAGE = 4 * XPL + WSLOTS, NST = 3;                     // 3 x 52 032 B
constexpr int N4 = 2;                                                // pieces of chunk c + 3 issued under the last group of chunk c
constexpr int KMIX = NXP / NW;                                       // the round of the list where the kind changes (4: waves 0-3 activations)
static_assert(NBLK <= NBJ * NW && KMIX * NW <= NXP && (KMIX + 1) * NW > NXP && NPC == 7, "piece list layout");
}

// F8: "f8" operands (hi f16 + fp8 correction operands: 14 matrix instructions per tap-chunk and block) or H2 operands (hi / lo f16:
// three f16 MFMAs per tap, 27 per chunk and block -- the `h3` arithmetic mode)
// F6 (with F8): the "f6" operand format (nb_h3_common.h): the correction products on fp6 MFMAs, 14 x 32 = 448 matrix cycles per chunk and block
// instead of 9 x 32 + 5 x 64 = 608.  A lane's 32 K values of a correction instruction = ONE tap at ONE position, both terms (one block
// scale per lane): lane half 0 takes the first tap of a pair, lane half 1 the second -- per-lane lo-slot addresses --, and the operand
// tuple is the chunk's two lo slots as they stand (quad 0 = (cg 0, lo), quad 1 = (cg 1, lo): six dwords of fields, the scale dword =
// the instruction's scale operand).  The lone tap 4: lane half 1 multiplies a zero slot on the weights side.
// HO (with F8, round 6): "hi only" -- the f8 loop WITHOUT its correction products: one f16 MFMA per tap on the hi halves, the plain
// single-f16 evaluation (~3e-3 from fp32 end to end: outside the 1e-3 budget; the reference's own shipped arithmetic for its blocks
// >= 32^2, training/networks.py:634-638).  Generator(conv_mode="f16"): a data point that separates the cost of the split scheme from
// the cost of the kernel structure, NOT a parity mode.  Same staging, same fillers, same epilogue; the lo-fragment reads have no
// consumer and are dropped by the compiler.
// PERSIST = false: the same body WITHOUT the tile loop (item_next is the list's end at compile time: no next tile, no prefetch, nothing carried) -- one
// workgroup per tile, as until round 5 (see modconv3x3_up1_h3_kernel's PERSIST and profiles/r06_ab_variants.txt for why both exist).
template <bool F8, int OUTM, bool F6 = false, bool HO = false, bool PERSIST = true>
__global__ __launch_bounds__(NT, 2) void modconv3x3_up2v_kernel(const H3Up2Params p) {
    static_assert(!F6 || F8, "the f6 form is a variant of the f8 loop");
    static_assert(!HO || (F8 && !F6), "the hi-only form is a variant of the f8 loop");
    NB_TSTAMP(0);
    if constexpr (OUTM == 2) nb_set_fp16_ovfl();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_v[];
    h8* ring = reinterpret_cast<h8*>(smem_v);         // [NST][ x: 4 planes x XPL | w: 36 rows x 32 ]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    const int H = p.h, W = p.w;
    // The per-tile set-up below reads its launch parameters through a LAUNDERED pointer to the kernarg segment: read through `p` they are
    // loop invariants that the compiler keeps in ~40 SGPRs across the K loop (12 pointers of the epilogue tables and the noise source,
    // the tile-list geometry ...), which spilled 112 SGPRs and 60 VGPRs; re-read per tile they cost a handful of scalar loads.
    typedef const __attribute__((address_space(4))) H3Up2Params* kparams_t;
    auto fresh_params = [&]() -> kparams_t {
        kparams_t q = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(q));
        return q;
    };
    // PERSISTENT workgroups (round 6): the launch has gridDim.x = min(items, 4 x CUs) workgroups and workgroup w renders items w, w + gridDim.x,
    // ... of the list [sample][tile][c_out slice] (p.items of them, p.items_x per sample).  What that buys is the prologue: the first
    // chunk of the NEXT tile is on its way (LDS-DMA) while this tile's epilogue runs -- a workgroup used to spend 4.4-5 of its 31-50 us
    // waiting for that round trip with the matrix pipes idle, and one workgroup per CU (156 KB of LDS) means nobody else could use them.
    // XCD-aware order.  Hardware workgroup ids go round-robin to the 8 XCDs, each with its own L2.  The whole list is renumbered so that
    // an XCD works through a contiguous eighth of it: the c_out slices of a tile (which re-read the same activations) run side by side on
    // one XCD, its neighbours in the image (shared halo rows, shared 128-byte lines at the tile's left and right edge) right after
    // them, and every XCD gets whole samples, i.e. the same mix of full and ragged tiles; gridDim.x is a multiple of 8 whenever the
    // list is, so a workgroup's items all sit on its own XCD.  (dbg & 32: the round-3 order, per sample with a rotating XCD share --
    // there 6 or 11 workgroups of a sample per XCD cut through the slices of a tile and the tile was fetched by two L2s.)
    const unsigned total = (unsigned)p.items;
    int n = 0, I0 = 0, J0 = 0, co0 = 0;
    const size_t HW8 = (size_t)H * W * 8;
    auto tile_coords = [&](unsigned l) {
        kparams_t q_ = fresh_params();
        const unsigned total = (unsigned)q_->items, gx = (unsigned)q_->items_x;
        const int dbg = q_->dbg, slices = q_->slices, tiles_x = q_->tiles_x;
        unsigned b = l % gx;
        unsigned nn = l / gx;
        if (total % 8 == 0 && gridDim.x % 8 == 0 && !(dbg & (8 | 32))) {
            const unsigned q = (l & 7) * (total >> 3) + (l >> 3);
            nn = q / gx; b = q - nn * gx;
        } else if (gx % 8 == 0 && !(dbg & 8)) b = (((b & 7) + nn) & 7) * (gx >> 3) + (b >> 3);
        const int slice = b % slices; b /= slices;
        const int tile_x = b % tiles_x; const int tile_y = b / tiles_x;
        // (uniform values: the divisions above run on the vector unit -- back into scalar registers)
        n = __builtin_amdgcn_readfirstlane((int)nn); I0 = __builtin_amdgcn_readfirstlane(tile_y * TQH);
        J0 = __builtin_amdgcn_readfirstlane(tile_x * TQW); co0 = __builtin_amdgcn_readfirstlane(slice * CO_WG);
    };
    unsigned item = __builtin_amdgcn_readfirstlane(blockIdx.x);
    tile_coords(item);
    const int nblk = wv < NBLK - (NBJ - 1) * NW ? NBJ : NBJ - 1;      // blocks wv, wv + 8 (< 15)

    __shared__ __attribute__((aligned(16))) float s_dco[CO_WG], s_bias[CO_WG], s_nst[CO_WG];
    __shared__ __attribute__((aligned(16))) float s_noise[2 * TQH * 2 * TQW];
    // per-tile epilogue operands (written at the top of a tile's iteration: the previous tile's epilogue has passed its closing barrier)
    auto load_channel_tables = [&]() {
        kparams_t q_ = fresh_params();
        if (tid < CO_WG) {
            const int co = co0 + tid, c_out = q_->c_out;
            const float gain = q_->gain;
            s_dco[tid] = co < c_out ? q_->dcoefs[(size_t)n * c_out + co] * gain : 0.f;
            s_bias[tid] = co < c_out ? q_->bias[co] * gain : 0.f;
            s_nst[tid] = (q_->yh2 && co < c_out) ? q_->next_styles[(size_t)n * q_->next_stride + co] : 0.f;
        }
    };
    // ---- LDS-DMA pieces: position u = 8 k + wv of ONE list per chunk -- activation piece u (plane u / 9, 64 slots from
    //      (u % 9) 64; per-lane source, out-of-image slots read the zero page with stride 0, the ninth piece of a plane is
    //      partial: uniform lane mask) if u < 36, else weight piece u - 36 (64 slots = two rows of 32 c_out; uniform base + lane
    //      offset).  Every wave issues exactly NPC pieces per chunk: the waits below count them. ----
    const unsigned lds0 = (unsigned)(uintptr_t)NB_LDS_PTR(smem_v);
    const size_t wchunk = (size_t)WROWS * p.co_ld * 16;               // bytes of a chunk's weights
    const char* xsrc0[KMIX];
    unsigned xstr[KMIX];
    unsigned woff[NPC - KMIX - 1];                     // lane's byte offset within a chunk's weights (rounds > KMIX): slot e = q 64 + lane -> row e / 32, c_out e % 32
    const char* msrc0 = nullptr;
    unsigned mstr = 0;
    // (wave-uniform, the same for every tile: kept in SCALAR registers -- computed through the vector unit's divisions they sat in vector
    //  registers across the tile loop, ten of them spilled)
    int xdst_s[KMIX + 1];
    unsigned xlast_bits = 0;                           // bit k: round k's activation piece is the partial ninth piece of its plane
#pragma unroll
    for (int k = 0; k <= KMIX; ++k) {
        int q = k * NW + wv; q = q < NXP ? q : NXP - 1;
        const int pl = q / PP, part = q - pl * PP;
        xdst_s[k] = __builtin_amdgcn_readfirstlane(pl * XPL + part * 64);
        xlast_bits |= (part == PP - 1 ? 1u : 0u) << k;
    }
    xlast_bits = __builtin_amdgcn_readfirstlane(xlast_bits);
    auto xdst = [&](int k) { return xdst_s[k]; };
    auto xmask = [&](int k) -> unsigned long long { return (xlast_bits >> k) & 1u ? (1ull << (XPL - (PP - 1) * 64)) - 1 : ~0ull; };
    // weight piece of list round k (k >= KMIX): q = 8 k + wv - 36 (clamped: the list's last two positions re-copy piece 17)
    auto wq = [&](int k) { int q = k * NW + wv - NXP; return __builtin_amdgcn_readfirstlane(q < 0 ? 0 : (q < NWP ? q : NWP - 1)); };
    // the round of the list where the kind changes (waves 0-3: an activation piece, 4-7: a weight piece): ONE form for both --
    // per-lane source of chunk 0 and per-chunk stride, uniform destination and mask -- set up here, so that the K loop holds
    // neither a branch nor a select for it (a uniform `cond ? a : b` of two address computations compiles to a branch, the
    // branch splits the chunk into two basic blocks, and the compiler sinks MFMAs from the first into the second)
    const bool mix_isx = KMIX * NW + wv < NXP;
    const int mdst = mix_isx ? xdst(KMIX) : 4 * XPL + wq(KMIX) * 64;
    const unsigned long long mmask = mix_isx ? xmask(KMIX) : ~0ull;
    // the per-lane sources of the CURRENT tile's pieces (I0, J0, n, co0 as tile_coords left them)
    auto piece_sources = [&]() {
        kparams_t q_ = fresh_params();
        const char* zeros = reinterpret_cast<const char*>(q_->zeros);
        const int co_ld = q_->co_ld;
        const _Float16* xn = q_->x + (size_t)n * q_->c8 * 2 * HW8;        // (the sample's activations: not carried from tile to tile either)
        // (I0, J0, co0 through an opaque statement: the sources are RECOMPUTED by the call behind the epilogue, not carried through it)
        int i0_ = I0, j0_ = J0, c0_ = co0, lane_ = lane;             // (the lane too: slot row / column of a piece are tile-independent and would be hoisted -- ten spilled registers)
        asm volatile("" : "+v"(i0_), "+v"(j0_), "+v"(c0_), "+v"(lane_));
        i0_ = __builtin_amdgcn_readfirstlane(i0_); j0_ = __builtin_amdgcn_readfirstlane(j0_); c0_ = __builtin_amdgcn_readfirstlane(c0_);
        // (round KMIX of the list -- an activation piece on waves 0-3, a weight piece on 4-7 -- only lives in msrc0 / mstr)
        const char* xs_mix = zeros;
        unsigned xt_mix = 0, wo_mix = 0;
#pragma unroll
        for (int k = 0; k <= KMIX; ++k) {
            int q = k * NW + wv;
            q = q < NXP ? q : NXP - 1;
            const int pl = q / PP, part = q - pl * PP;
            const int e = part * 64 + lane_;
            const char* xs_ = zeros;
            unsigned xt_ = 0;
            if (e < XPL) {
                const int r = e / XS, c = e - r * XS;
                const int gy = i0_ - 1 + r, gxx = j0_ - 1 + c;
                if (gy >= 0 && gy < H && gxx >= 0 && gxx < W) {
                    xs_ = reinterpret_cast<const char*>(xn + (size_t)pl * HW8 + (size_t)(gy * W + gxx) * 8);
                    xt_ = (unsigned)(4 * HW8 * 2);
                }
            }
            if (k < KMIX) { xsrc0[k] = xs_; xstr[k] = xt_; } else { xs_mix = xs_; xt_mix = xt_; }
        }
#pragma unroll
        for (int k = KMIX; k < NPC; ++k) {
            const int e = wq(k) * 64 + lane_;
            const unsigned wo_ = (unsigned)(((size_t)(e >> 5) * co_ld + c0_ + (e & 31)) * 16);
            if (k > KMIX) woff[k - KMIX - 1] = wo_; else wo_mix = wo_;
        }
        msrc0 = mix_isx ? xs_mix : reinterpret_cast<const char*>(q_->wts) + wo_mix;
        mstr = mix_isx ? xt_mix : (unsigned)wchunk;
    };
    piece_sources();
    auto issue_piece = [&](auto kk, int c, int stage_slot) {
        constexpr int k = decltype(kk)::value;
        if constexpr (k < KMIX) {
            nb_lds_dma16_m(xsrc0[k] + (size_t)c * xstr[k], lds0 + (unsigned)(stage_slot + xdst(k)) * 16u, xmask(k));
        } else if constexpr (k > KMIX) {
            const char* wbase = reinterpret_cast<const char*>(p.wts) + (size_t)c * wchunk;
            nb_lds_dma16_s(wbase, woff[k - KMIX - 1], lds0 + (unsigned)(stage_slot + 4 * XPL + wq(k) * 64) * 16u);
        } else {
            nb_lds_dma16_m(msrc0 + (size_t)c * mstr, lds0 + (unsigned)(stage_slot + mdst) * 16u, mmask);
        }
    };

    // fragment offsets (16-byte slots).  B: position (r, c) of block (wv + 8 j): slot r XS + c of plane (lh 2 + hl);
    // A: row tap 4 + lh 2 + hl, column l31
    int boff[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        int pidx = (wv + NW * j) * 32 + l31;
        pidx = pidx < NPOS ? pidx : NPOS - 1;
        const int r = pidx / PW, c = pidx - r * PW;
        boff[j] = lh * 2 * XPL + r * XS + c;
    }
    const int aoff = 4 * XPL + lh * 2 * CO_WG + l31;  // + tap * 128 + hl * 32
    // f6: lo slot q of tap t = row 4 t + 2 q + 1 of the weights; of position offset d = plane 2 q + 1 of the activations
    const int a6 = 4 * XPL + CO_WG + l31;             // + tap * 128 + q * 64
    int b6[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) b6[j] = boff[j] - lh * 2 * XPL + XPL;     // + q * 2 XPL + offset
    __shared__ __attribute__((aligned(16))) h8 s_zero6[1];
    if constexpr (F6) { if (tid == 0) s_zero6[0] = h8{}; }

    f32x16 acc[NBJ][4];
    const int NC = p.nchunks;

    // ---- fragment registers (loop-carried).  A: sets a / n (a tap pair: two hi fragments + ONE lo tuple), m (the lone tap 4, lo
    //      tuple = (tap 4 | zeros)).  B: hi fragments at input offsets 0, 1, XS, XS + 1; lo tuples bl01 = (offset 0 | offset 1) --
    //      whose second half is reloaded with offset XS's bytes once taps 6 and 3 are through: it then is the operand of taps
    //      (7, 1); the lone tap in between multiplies that half by A's zeros, whatever it holds -- and bl23 = (XS | XS + 1). ----
    h8 ah_a[2], ah_n[2], ah_m;
    i32x8 al_a, al_n, al_m;
    h8 bh0[NBJ], bh1[NBJ], bh2[NBJ], bh3[NBJ];
    i32x8 bl01[NBJ], bl23[NBJ];
    const int sa_ = lh ? 116 : 127, sb_ = lh ? 129 : 118;      // E8M0 block scales (see modconv3x3_up1_h3_kernel)
    // f6 operands: six registers + the scale dword each (A: pairs a / n, the lone tap m; B: positions (0 | 1) -- later (0 | XS) --, (XS | XS + 1))
    v6i al6_a, al6_n, al6_m, bl6_01[NBJ], bl6_23[NBJ];
    int sa6_a, sa6_n, sa6_m, sb6_01[NBJ], sb6_23[NBJ];
    // H2 operands: separate lo fragments (no tuples): A sets a / n [tap of the pair], m; B lo at the four input offsets
    h8 hl_a[2], hl_n[2], hl_m, bl0[NBJ], bl1[NBJ], bl2[NBJ], bl3[NBJ];
    // (cleared at the top of every tile: the ~100 fragment registers then are NOT values carried from one tile to the next -- and
    //  through its epilogue, where the register file is full of accumulators and FIR operands)
    auto zero_fragments = [&]() {
        ah_m = h8{}; al_a = i32x8{}; al_n = i32x8{}; al_m = i32x8{};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah_a[i] = h8{}; ah_n[i] = h8{};
            bh0[i] = h8{}; bh1[i] = h8{}; bh2[i] = h8{}; bh3[i] = h8{}; bl01[i] = i32x8{}; bl23[i] = i32x8{};
        }
        al6_a = v6i{}; al6_n = v6i{}; al6_m = v6i{}; sa6_a = 0; sa6_n = 0; sa6_m = 0;
#pragma unroll
        for (int i = 0; i < NBJ; ++i) { bl6_01[i] = v6i{}; bl6_23[i] = v6i{}; sb6_01[i] = 0; sb6_23[i] = 0; }
        hl_m = h8{};
#pragma unroll
        for (int i = 0; i < 2; ++i) { hl_a[i] = h8{}; hl_n[i] = h8{}; bl0[i] = h8{}; bl1[i] = h8{}; bl2[i] = h8{}; bl3[i] = h8{}; }
    };

#define NB_FENCE() __builtin_amdgcn_sched_barrier(0)
    auto set_lo = [](i32x8& t, const h8& v) { const i32x4 x = __builtin_bit_cast(i32x4, v); t[0] = x[0]; t[1] = x[1]; t[2] = x[2]; t[3] = x[3]; };
    auto set_hi = [](i32x8& t, const h8& v) { const i32x4 x = __builtin_bit_cast(i32x4, v); t[4] = x[0]; t[5] = x[1]; t[6] = x[2]; t[7] = x[3]; };
    // one chunk.  NBE = position blocks this wave multiplies; MODE = min(3, NC - 1 - c): 3 steady state, 2 / 1 / 0 the last three
    // chunks.  sa: this chunk's stage, san: the next chunk's; d2 / d3: first slots of the stages of chunks c + 2 / c + 3
    auto chunk = [&](auto mode_, auto nbe_, int c, const h8* sa, const h8* san, int d2, int d3) {
        constexpr int MODE = decltype(mode_)::value, NBE = decltype(nbe_)::value;
#ifdef NB_ABL6_NOREAD
        const h8 zero_frag = h8{};
        auto rA = [&](int tap, int hl, const h8* s) -> const h8& { return zero_frag; };
        auto rBh = [&](h8 (&bh)[NBJ], auto j_, int del, const h8* s) {};
#else
        auto rA = [&](int tap, int hl, const h8* s) -> const h8& { return s[aoff + tap * 128 + hl * 32]; };
        auto rBh = [&](h8 (&bh)[NBJ], auto j_, int del, const h8* s) { constexpr int j = decltype(j_)::value; if constexpr (j < NBE) bh[j] = s[boff[j] + del]; };
#endif
        auto rBl = [&](i32x8 (&bl)[NBJ], int half, auto j_, int del, const h8* s) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j < NBE) { if (half) set_hi(bl[j], s[boff[j] + XPL + del]); else set_lo(bl[j], s[boff[j] + XPL + del]); }
        };
        // piece k of the interval: 0 .. N4-1 belong to chunk c + 3 (under this chunk's last group), N4 .. 6 to chunk c + 2
        auto dma = [&](auto k_) {
            constexpr int k = decltype(k_)::value;
#ifndef NB_ABL_NODMA
            if constexpr (k < N4) { if constexpr (MODE == 3) issue_piece(k_, c + 3, d3); }
            else { if constexpr (MODE >= 2) issue_piece(k_, c + 2, d2); }
#endif
        };
        using J0_ = std::integral_constant<int, 0>; using J1_ = std::integral_constant<int, 1>;
        if constexpr (!F8) {
            // ---- H2 operands: per tap (hi x hi), (hi x lo), (lo x hi) in that order (the order of modconv3x3_up2_h3_kernel) ----
            auto rBl1 = [&](h8 (&bl)[NBJ], auto j_, int del, const h8* s_) { constexpr int j = decltype(j_)::value; if constexpr (j < NBE) bl[j] = s_[boff[j] + XPL + del]; };
            // a tap pair on both blocks: 2 x (3 + 3) f16 MFMAs; filler(g) behind MFMA g = 6 j + position
            auto group6 = [&](auto ph_, h8 (&ah)[2], h8 (&al)[2], h8 (&Bha)[NBJ], h8 (&Bla)[NBJ], h8 (&Bhb)[NBJ], h8 (&Blb)[NBJ], auto&& filler) {
                constexpr int ph = decltype(ph_)::value;
                nb_static_for<0, 2>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    f32x16& a_ = acc[j][ph];
                    nb_static_for<0, 6>([&](auto q_) {
                        constexpr int q = decltype(q_)::value;
                        constexpr int t = q / 3, k = q % 3;
                        if constexpr (j < NBE) {
                            h8 (&Bh)[NBJ] = t ? Bhb : Bha;
                            h8 (&Bl)[NBJ] = t ? Blb : Bla;
                            if constexpr (k == 0) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], Bh[j], a_, 0, 0, 0);
                            else if constexpr (k == 1) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], Bl[j], a_, 0, 0, 0);
                            else a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], Bh[j], a_, 0, 0, 0);
                        }
                        NB_FENCE(); filler(std::integral_constant<int, 6 * j + q>{}); NB_FENCE();
                    });
                });
            };
            NB_FENCE();
            // G0: taps 8, 6 -> phase 0.  Fillers: the A fragments of G1; pieces 2, 3
            group6(std::integral_constant<int, 0>{}, ah_a, hl_a, bh0, bl0, bh1, bl1, [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g == 0) ah_n[0] = rA(5, 0, sa);
                else if constexpr (g == 2) hl_n[0] = rA(5, 1, sa);
                else if constexpr (g == 4) dma(std::integral_constant<int, 2>{});
                else if constexpr (g == 6) ah_n[1] = rA(3, 0, sa);
                else if constexpr (g == 8) hl_n[1] = rA(3, 1, sa);
                else if constexpr (g == 10) dma(std::integral_constant<int, 3>{});
            });
            // G1: taps 5, 3 -> phase 2.  Fillers: tap 4, the row-below fragments; piece 4
            group6(std::integral_constant<int, 2>{}, ah_n, hl_n, bh0, bl0, bh1, bl1, [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g == 0) ah_m = rA(4, 0, sa);
                else if constexpr (g == 2) hl_m = rA(4, 1, sa);
                else if constexpr (g == 4) dma(std::integral_constant<int, 4>{});
                else if constexpr (g == 6) rBh(bh2, J0_{}, XS, sa);
                else if constexpr (g == 7) rBl1(bl2, J0_{}, XS, sa);
                else if constexpr (g == 9) rBh(bh2, J1_{}, XS, sa);
                else if constexpr (g == 10) rBl1(bl2, J1_{}, XS, sa);
            });
            // G2: tap 4 -> phase 3: 3 f16 MFMAs per block.  Fillers: the A fragments of G3; piece 5
            nb_static_for<0, 2>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                f32x16& a_ = acc[j][3];
                if constexpr (j < NBE) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_m, bh0[j], a_, 0, 0, 0);
                NB_FENCE();
                if constexpr (j == 0) ah_a[0] = rA(7, 0, sa); else ah_a[1] = rA(1, 0, sa);
                NB_FENCE();
                if constexpr (j < NBE) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_m, bl0[j], a_, 0, 0, 0);
                NB_FENCE();
                if constexpr (j == 0) hl_a[0] = rA(7, 1, sa); else hl_a[1] = rA(1, 1, sa);
                NB_FENCE();
                if constexpr (j < NBE) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl_m, bh0[j], a_, 0, 0, 0);
                NB_FENCE();
                if constexpr (j == 0) dma(std::integral_constant<int, 5>{});
                NB_FENCE();
            });
            // G3: taps 7, 1 -> phase 1.  Fillers: the A fragments of G4, the diagonal fragments; piece 6
            group6(std::integral_constant<int, 1>{}, ah_a, hl_a, bh0, bl0, bh2, bl2, [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g == 0) ah_n[0] = rA(2, 0, sa);
                else if constexpr (g == 2) hl_n[0] = rA(2, 1, sa);
                else if constexpr (g == 4) dma(std::integral_constant<int, 6>{});
                else if constexpr (g == 5) ah_n[1] = rA(0, 0, sa);
                else if constexpr (g == 6) hl_n[1] = rA(0, 1, sa);
                else if constexpr (g == 7) rBh(bh3, J0_{}, XS + 1, sa);
                else if constexpr (g == 8) rBl1(bl3, J0_{}, XS + 1, sa);
                else if constexpr (g == 9) rBh(bh3, J1_{}, XS + 1, sa);
                else if constexpr (g == 10) rBl1(bl3, J1_{}, XS + 1, sa);
            });
            if constexpr (MODE >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPC) : "memory");
            else if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            NB_FENCE();
            // G4: taps 2, 0 -> phase 0.  Fillers: the NEXT chunk's first operands; pieces 0, 1 of chunk c + 3
            group6(std::integral_constant<int, 0>{}, ah_n, hl_n, bh2, bl2, bh3, bl3, [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g == 4) dma(std::integral_constant<int, 0>{});
                if constexpr (g == 10) dma(std::integral_constant<int, 1>{});
                if constexpr (MODE >= 1) {
                    if constexpr (g == 0) ah_a[0] = rA(8, 0, san);
                    else if constexpr (g == 1) { rBh(bh0, J0_{}, 0, san); rBl1(bl0, J0_{}, 0, san); }
                    else if constexpr (g == 2) hl_a[0] = rA(8, 1, san);
                    else if constexpr (g == 3) ah_a[1] = rA(6, 0, san);
                    else if constexpr (g == 5) { rBh(bh1, J0_{}, 1, san); rBl1(bl1, J0_{}, 1, san); }
                    else if constexpr (g == 6) hl_a[1] = rA(6, 1, san);
                    else if constexpr (g == 7) { rBh(bh0, J1_{}, 0, san); rBl1(bl0, J1_{}, 0, san); }
                    else if constexpr (g == 8) { rBh(bh1, J1_{}, 1, san); rBl1(bl1, J1_{}, 1, san); }
                }
            });
            NB_FENCE();
            return;
        }
        if constexpr (F6) {
            // An fp6 operand = SIX registers + the scale dword.  Read as 16 + 8 + 4 bytes (ds_read_b128 / b64 / b32, volatile so that
            // the compiler does not fuse them): a 16-byte read cannot land across the end of the six-register operand, and two
            // 16-byte reads into an eight-register tuple cost two v_mov per operand and enough extra live registers to spill.
            auto rd6 = [](v6i& t, int& sc, const h8* slot0, const h8* slot1) {
                // (explicit LDS address space: through a generic pointer the volatile reads become flat loads, which count in vmcnt)
                typedef const volatile __attribute__((address_space(3))) int* lds_vint;
                typedef const volatile __attribute__((address_space(3))) i32x2* lds_vint2;
                const i32x4 q0 = __builtin_bit_cast(i32x4, *slot0);
                const i32x2 q1 = *(lds_vint2)NB_LDS_PTR(slot1);
                sc = ((lds_vint)NB_LDS_PTR(slot1))[2];
                t[0] = q0[0]; t[1] = q0[1]; t[2] = q0[2]; t[3] = q0[3]; t[4] = q1[0]; t[5] = q1[1];
            };
            // weights of pair (ta, tb): the lane's own tap; of the lone tap 4 (lane half 1: the zero slot)
            auto rA6 = [&](v6i& al, int& sc, int ta, int tb, const h8* s_) {
#ifdef NB_ABL6_NOREAD
                return;
#endif
                const h8* b_ = s_ + a6 + (lh ? tb : ta) * 128;
                rd6(al, sc, b_, b_ + 64);
            };
            auto rA6m = [&](v6i& al, int& sc, const h8* s_) {
#ifdef NB_ABL6_NOREAD
                return;
#endif
                const h8* b_ = lh ? &s_zero6[0] : s_ + a6 + 4 * 128;
                rd6(al, sc, b_, lh ? &s_zero6[0] : b_ + 64);
            };
            // activations of block j at offsets (da | db)
            auto rB6 = [&](v6i (&bl)[NBJ], int (&sc)[NBJ], auto j_, int da, int db, const h8* s_) {
                constexpr int j = decltype(j_)::value;
#ifdef NB_ABL6_NOREAD
                return;
#endif
                if constexpr (j < NBE) {
                    const h8* b_ = s_ + b6[j] + (lh ? db : da);
                    rd6(bl[j], sc[j], b_, b_ + 2 * XPL);
                }
            };
            auto wide = [](const v6i& t) { return i32x8{t[0], t[1], t[2], t[3], t[4], t[5], 0, 0}; };
            auto group_pair6 = [&](auto ph_, h8 (&ah)[2], v6i& al, int sa6, h8 (&Ba)[NBJ], h8 (&Bb)[NBJ], v6i (&bl)[NBJ], int (&sb6)[NBJ], auto&& filler) {
                constexpr int ph = decltype(ph_)::value;
                nb_static_for<0, 2>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    f32x16& a_ = acc[j][ph];
                    if constexpr (j < NBE && NB_ABL6_F16) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], Ba[j], a_, 0, 0, 0);
                    NB_FENCE(); filler(std::integral_constant<int, 3 * j>{}); NB_FENCE();
                    if constexpr (j < NBE && NB_ABL6_F16) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], Bb[j], a_, 0, 0, 0);
                    NB_FENCE(); filler(std::integral_constant<int, 3 * j + 1>{}); NB_FENCE();
                    if constexpr (j < NBE && NB_ABL6_FP6) a_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wide(al), wide(bl[j]), a_, 2, 2, 0, sa6, 0, sb6[j]);
                    NB_FENCE(); filler(std::integral_constant<int, 3 * j + 2>{}); NB_FENCE();
                });
            };
            NB_FENCE();
            // G0: taps 8, 6 -> phase 0 (positions 0 | 1).  Fillers: the A fragments of G1; pieces 2, 3
            group_pair6(std::integral_constant<int, 0>{}, ah_a, al6_a, sa6_a, bh0, bh1, bl6_01, sb6_01, [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g == 0) ah_n[0] = rA(5, 0, sa);
                else if constexpr (g == 1) ah_n[1] = rA(3, 0, sa);
                else if constexpr (g == 2) dma(std::integral_constant<int, 2>{});
                else if constexpr (g == 3) rA6(al6_n, sa6_n, 5, 3, sa);
                else if constexpr (g == 5) dma(std::integral_constant<int, 3>{});
            });
            // G1: taps 5, 3 -> phase 2 (positions 0 | 1).  Fillers: tap 4, the row-below hi fragments; piece 4
            group_pair6(std::integral_constant<int, 2>{}, ah_n, al6_n, sa6_n, bh0, bh1, bl6_01, sb6_01, [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g == 0) ah_m = rA(4, 0, sa);
                else if constexpr (g == 1) rA6m(al6_m, sa6_m, sa);
                else if constexpr (g == 2) dma(std::integral_constant<int, 4>{});
                else if constexpr (g == 3) rBh(bh2, J0_{}, XS, sa);
                else if constexpr (g == 4) rBh(bh2, J1_{}, XS, sa);
            });
            // G2: tap 4 -> phase 3: (f16, fp6) x 2; lane half 1 of A = zeros.  Fillers: the A fragments of G3; piece 5; behind the fp6
            // MFMA that was the last to read a block's (0 | 1) operand, the (0 | XS) slots into it
            nb_static_for<0, 2>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                f32x16& a_ = acc[j][3];
                if constexpr (j < NBE && NB_ABL6_F16) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_m, bh0[j], a_, 0, 0, 0);
                NB_FENCE();
                if constexpr (j == 0) { ah_a[0] = rA(7, 0, sa); ah_a[1] = rA(1, 0, sa); } else rA6(al6_a, sa6_a, 7, 1, sa);
                NB_FENCE();
                if constexpr (j < NBE && NB_ABL6_FP6) a_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wide(al6_m), wide(bl6_01[j]), a_, 2, 2, 0, sa6_m, 0, sb6_01[j]);
                NB_FENCE();
                if constexpr (j == 0) dma(std::integral_constant<int, 5>{});
                rB6(bl6_01, sb6_01, j_, 0, XS, sa);
                NB_FENCE();
            });
            // G3: taps 7, 1 -> phase 1 (positions 0 | XS).  Fillers: the A fragments of G4, the diagonal fragments, the (XS | XS + 1) slots; piece 6
            group_pair6(std::integral_constant<int, 1>{}, ah_a, al6_a, sa6_a, bh0, bh2, bl6_01, sb6_01, [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g == 0) { ah_n[0] = rA(2, 0, sa); ah_n[1] = rA(0, 0, sa); }
                else if constexpr (g == 1) rB6(bl6_23, sb6_23, J0_{}, XS, XS + 1, sa);
                else if constexpr (g == 2) dma(std::integral_constant<int, 6>{});
                else if constexpr (g == 3) { rA6(al6_n, sa6_n, 2, 0, sa); rBh(bh3, J0_{}, XS + 1, sa); }
                else if constexpr (g == 4) rBh(bh3, J1_{}, XS + 1, sa);
                else rB6(bl6_23, sb6_23, J1_{}, XS, XS + 1, sa);
            });
            if constexpr (MODE >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPC) : "memory");
            else if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            NB_FENCE();
            // G4: taps 2, 0 -> phase 0 (positions XS | XS + 1).  Fillers: the NEXT chunk's first operands; pieces 0, 1 of chunk c + 3
            group_pair6(std::integral_constant<int, 0>{}, ah_n, al6_n, sa6_n, bh2, bh3, bl6_23, sb6_23, [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                if constexpr (g == 2) dma(std::integral_constant<int, 0>{});
                if constexpr (g == 5) dma(std::integral_constant<int, 1>{});
                if constexpr (MODE >= 1) {
                    if constexpr (g == 0) { ah_a[0] = rA(8, 0, san); rBh(bh0, J0_{}, 0, san); }
                    else if constexpr (g == 1) { ah_a[1] = rA(6, 0, san); rBh(bh1, J0_{}, 1, san); }
                    else if constexpr (g == 2) { rBh(bh0, J1_{}, 0, san); rBh(bh1, J1_{}, 1, san); }
                    else if constexpr (g == 3) rA6(al6_a, sa6_a, 8, 6, san);
                    else if constexpr (g == 4) rB6(bl6_01, sb6_01, J0_{}, 0, 1, san);
                    else rB6(bl6_01, sb6_01, J1_{}, 0, 1, san);
                }
            });
            NB_FENCE();
            return;
        }
        // a tap pair on both blocks: (f16, f16, fp8) x 2; filler(g) behind MFMA g = 3 j + position
        auto group_pair = [&](auto ph_, h8 (&ah)[2], i32x8& al, h8 (&Ba)[NBJ], h8 (&Bb)[NBJ], i32x8 (&bl)[NBJ], auto&& filler) {
            constexpr int ph = decltype(ph_)::value;
            nb_static_for<0, 2>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                f32x16& a_ = acc[j][ph];
                if constexpr (j < NBE) a_ = NB_MFMA_F16(ah[0], Ba[j], a_, 1);
                NB_FENCE(); filler(std::integral_constant<int, 3 * j>{}); NB_FENCE();
                if constexpr (j < NBE) a_ = NB_MFMA_F16(ah[1], Bb[j], a_, 0);
                NB_FENCE(); filler(std::integral_constant<int, 3 * j + 1>{}); NB_FENCE();
                if constexpr (j < NBE && !HO) a_ = NB_MFMA_FP8(al, bl[j], a_, sa_, sb_, 1);
                NB_FENCE(); filler(std::integral_constant<int, 3 * j + 2>{}); NB_FENCE();
            });
        };
        NB_FENCE();
        // G0: taps 8, 6 -> phase 0.  Fillers: the A fragments of G1; pieces 2, 3
        group_pair(std::integral_constant<int, 0>{}, ah_a, al_a, bh0, bh1, bl01, [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g == 0) ah_n[0] = rA(5, 0, sa);
            else if constexpr (g == 1) ah_n[1] = rA(3, 0, sa);
            else if constexpr (g == 2) dma(std::integral_constant<int, 2>{});
            else if constexpr (g == 3) set_lo(al_n, rA(5, 1, sa));
            else if constexpr (g == 4) set_hi(al_n, rA(3, 1, sa));
            else dma(std::integral_constant<int, 3>{});
        });
        // G1: taps 5, 3 -> phase 2.  Fillers: tap 4, the row-below hi fragments, and -- behind the fp8 MFMA that was the last to
        // read offset 1's lo bytes of a block -- offset XS's into that half of bl01; piece 4
        group_pair(std::integral_constant<int, 2>{}, ah_n, al_n, bh0, bh1, bl01, [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g == 0) { ah_m = rA(4, 0, sa); set_lo(al_m, rA(4, 1, sa)); }
            else if constexpr (g == 1) { rBh(bh2, J0_{}, XS, sa); rBh(bh2, J1_{}, XS, sa); }
            else if constexpr (g == 2) { dma(std::integral_constant<int, 4>{}); rBl(bl01, 1, J0_{}, XS, sa); }
            else if constexpr (g == 5) rBl(bl01, 1, J1_{}, XS, sa);
        });
        // G2: tap 4 -> phase 3: (f16, fp8) x 2; A = (tap 4 | zeros).  Fillers: the A fragments of G3; piece 5
        nb_static_for<0, 2>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            f32x16& a_ = acc[j][3];
            if constexpr (j < NBE) a_ = NB_MFMA_F16(ah_m, bh0[j], a_, 1);
            NB_FENCE();
            if constexpr (j == 0) { ah_a[0] = rA(7, 0, sa); ah_a[1] = rA(1, 0, sa); } else { set_lo(al_a, rA(7, 1, sa)); set_hi(al_a, rA(1, 1, sa)); }
            NB_FENCE();
            if constexpr (j < NBE && !HO) a_ = NB_MFMA_FP8(al_m, bl01[j], a_, sa_, sb_, 1);
            NB_FENCE();
            if constexpr (j == 0) dma(std::integral_constant<int, 5>{});
            NB_FENCE();
        });
        // G3: taps 7, 1 -> phase 1 (bl01 = (offset 0 | XS) now).  Fillers: the A fragments of G4, the diagonal fragments; piece 6
        group_pair(std::integral_constant<int, 1>{}, ah_a, al_a, bh0, bh2, bl01, [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g == 0) ah_n[0] = rA(2, 0, sa);
            else if constexpr (g == 1) ah_n[1] = rA(0, 0, sa);
            else if constexpr (g == 2) { dma(std::integral_constant<int, 6>{}); rBl(bl23, 0, J0_{}, XS, sa); rBl(bl23, 1, J0_{}, XS + 1, sa); }
            else if constexpr (g == 3) { set_lo(al_n, rA(2, 1, sa)); set_hi(al_n, rA(0, 1, sa)); }
            else if constexpr (g == 4) { rBh(bh3, J0_{}, XS + 1, sa); rBh(bh3, J1_{}, XS + 1, sa); }
            else { rBl(bl23, 0, J1_{}, XS, sa); rBl(bl23, 1, J1_{}, XS + 1, sa); }
        });
        // everything of this chunk's stage has been read (G4's operands are on their way: waited for here); the next chunk has
        // landed: all but the NPC youngest pieces (chunk c + 2)
        if constexpr (MODE >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NPC) : "memory");
        else if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        NB_FENCE();
        // G4: taps 2, 0 -> phase 0.  Fillers: the NEXT chunk's first operands in the order G0 takes them; pieces 0, 1 of chunk c + 3
        group_pair(std::integral_constant<int, 0>{}, ah_n, al_n, bh2, bh3, bl23, [&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (g == 2) dma(std::integral_constant<int, 0>{});
            if constexpr (g == 5) dma(std::integral_constant<int, 1>{});
            if constexpr (MODE >= 1) {
                if constexpr (g == 0) { ah_a[0] = rA(8, 0, san); rBh(bh0, J0_{}, 0, san); }
                else if constexpr (g == 1) { ah_a[1] = rA(6, 0, san); rBh(bh1, J0_{}, 1, san); }
                else if constexpr (g == 2) { rBh(bh0, J1_{}, 0, san); rBh(bh1, J1_{}, 1, san); }
                else if constexpr (g == 3) { set_lo(al_a, rA(8, 1, san)); set_hi(al_a, rA(6, 1, san)); }
                else if constexpr (g == 4) { rBl(bl01, 0, J0_{}, 0, san); rBl(bl01, 1, J0_{}, 1, san); }
                else { rBl(bl01, 0, J1_{}, 0, san); rBl(bl01, 1, J1_{}, 1, san); }
            }
        });
        NB_FENCE();
    };
    auto kloop = [&](auto nbe) {
        constexpr int NBE = decltype(nbe)::value;
        // the first chunk's first operands
        ah_a[0] = ring[aoff + 8 * 128]; ah_a[1] = ring[aoff + 6 * 128];
        if constexpr (F6) {
            const h8* b_ = ring + a6 + (lh ? 6 : 8) * 128;
            const i32x4 q0 = __builtin_bit_cast(i32x4, b_[0]); const i32x4 q1 = __builtin_bit_cast(i32x4, b_[64]);
            al6_a = v6i{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1]}; sa6_a = q1[2];
        } else if constexpr (F8) { set_lo(al_a, ring[aoff + 8 * 128 + 32]); set_hi(al_a, ring[aoff + 6 * 128 + 32]); }
        else { hl_a[0] = ring[aoff + 8 * 128 + 32]; hl_a[1] = ring[aoff + 6 * 128 + 32]; }
#pragma unroll
        for (int j = 0; j < NBE; ++j) {
            bh0[j] = ring[boff[j]]; bh1[j] = ring[boff[j] + 1];
            if constexpr (F6) {
                const h8* b_ = ring + b6[j] + lh;
                const i32x4 q0 = __builtin_bit_cast(i32x4, b_[0]); const i32x4 q1 = __builtin_bit_cast(i32x4, b_[2 * XPL]);
                bl6_01[j] = v6i{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1]}; sb6_01[j] = q1[2];
            } else if constexpr (F8) { set_lo(bl01[j], ring[boff[j] + XPL]); set_hi(bl01[j], ring[boff[j] + XPL + 1]); }
            else { bl0[j] = ring[boff[j] + XPL]; bl1[j] = ring[boff[j] + XPL + 1]; }
        }
        int c = 0, s = 0;                             // s = stage of chunk c
        auto nxt = [](int s_) { return s_ == 2 ? 0 : s_ + 1; };
        for (; c + 3 < NC; ++c) {                     // stage of chunk c + 3 = stage of chunk c (overwritten behind the barrier)
            chunk(std::integral_constant<int, 3>{}, nbe, c, ring + s * STAGE, ring + nxt(s) * STAGE, nxt(nxt(s)) * STAGE, s * STAGE);
            s = nxt(s);
        }
        if (c + 2 < NC) { chunk(std::integral_constant<int, 2>{}, nbe, c, ring + s * STAGE, ring + nxt(s) * STAGE, nxt(nxt(s)) * STAGE, 0); s = nxt(s); ++c; }
        if (c + 1 < NC) { chunk(std::integral_constant<int, 1>{}, nbe, c, ring + s * STAGE, ring + nxt(s) * STAGE, 0, 0); s = nxt(s); ++c; }
        chunk(std::integral_constant<int, 0>{}, nbe, c, ring + s * STAGE, nullptr, 0, 0);
    };
    // ---- the first tile's chunk 0 (goes out alone and FIRST: a CU's LDS-DMA fill rate is ~25-35 GB/s, and pieces in flight together
    //      share it -- with chunks 1 and 2 issued in the same breath the first MFMA waited 1.5 us longer for chunk 0) ----
    nb_static_for<0, NPC>([&](auto k) { issue_piece(k, 0, 0); });
  for (;;) {                                            // one iteration per tile of this workgroup (see PERSISTENT above)
#undef NB_TSTAMP
// (the row address from a laundered copy of `item`: computed at the stamp, not carried in two vector registers across the K loop)
#define NB_TSTAMP(k) do { if (p.tstamps && threadIdx.x == 0) { unsigned it_ = item; asm volatile("" : "+v"(it_)); p.tstamps[(size_t)it_ * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
    // ---- the tile's prologue: chunk 0 is on its way (the first tile's was issued above; a later tile's activations went out before
    //      the previous tile's epilogue, its weights behind it); now chunk 1 and the two pieces of chunk 2 that the steady state
    //      issues under the previous chunk's last group.  The tile's epilogue operands and noise values are written meanwhile. ----
    load_channel_tables();
    {
        kparams_t q_ = fresh_params();
        const float* noise = q_->noise;
        const long long nstride = q_->noise_stride_n;
        const float gain = q_->gain;
        const NbNoiseSrcDev nsrc{q_->nsrc.const_t, q_->nsrc.lin, q_->nsrc.strength, q_->nsrc.norm_pos, q_->nsrc.positions, q_->nsrc.res, q_->nsrc.img_res};
        for (int e = tid; e < 2 * TQH * 2 * TQW; e += NT) {
            const int r = e / (2 * TQW), c = e - r * (2 * TQW);
            const int oy = 2 * I0 + r, ox = 2 * J0 + c;
            float v = (noise && oy < 2 * H) ? noise[(size_t)n * nstride + (size_t)oy * (2 * W) + ox] : 0.f;
            if (nsrc.const_t && oy < 2 * H) {
                float np0, np1, wx0, wx1, wy0, wy1;
                int sx0, sy0;
                nb_noise_np(nsrc, n, np0, np1);
                nb_noise_axis(nsrc, oy, np0, sx0, wx0, wx1);
                nb_noise_axis(nsrc, ox, np1, sy0, wy0, wy1);
                v = nb_noise_value(nsrc, nsrc.strength[0], sx0, wx0, wx1, sy0, wy0, wy1);
            }
            s_noise[e] = v * gain;
        }
    }
    if (NC > 1) nb_static_for<0, NPC>([&](auto k) { issue_piece(k, 1, STAGE); });
    if (NC > 2) nb_static_for<0, N4>([&](auto k) { issue_piece(k, 2, 2 * STAGE); });
    // (vmcnt counts the previous tile's epilogue stores too -- older than every piece of this tile: "all but the youngest N" covers them)
    if (NC > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC + N4) : "memory");
    else if (NC > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    NB_TSTAMP(1);

#pragma unroll
    for (int j = 0; j < NBJ; ++j)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][ph][r] = 0.f;
    zero_fragments();
    const unsigned t_loop0 = p.tstamps ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;      // (32 bits: a K loop is < 2^32 cycles)
    // position blocks of this wave that hold rows feeding stored pixels (see modconv3x3_up2_h3_kernel)
    const int nvalid_blk = (min(TQH, H - I0) + 2) * PW;
    int nbe_w = 0;
#pragma unroll
    for (int j = 0; j < NBJ; ++j) nbe_w += (wv + NW * j) * 32 < nvalid_blk && j < nblk;
    if (nbe_w == 2) kloop(std::integral_constant<int, 2>{});
    else if (nbe_w == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});
#undef NB_FENCE
    if (p.tstamps && tid == 0) {
        unsigned it_ = item; asm volatile("" : "+v"(it_));
        unsigned long long* ts = p.tstamps + (size_t)it_ * 8;
        ts[6] = 0; ts[7] = (unsigned long long)((unsigned)__builtin_amdgcn_s_memtime() - t_loop0) << 32;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // drain before the staging LDS is reused
    __builtin_amdgcn_s_barrier();
    // (the K loop's per-lane piece sources end here: an opaque "definition" tells the compiler so -- piece_sources() rewrites them all
    //  before their next use, but under a condition, and 21 registers would otherwise stay reserved through the epilogue)
#pragma unroll
    for (int k = 0; k < KMIX; ++k) { asm volatile("" : "=v"(xsrc0[k])); asm volatile("" : "=v"(xstr[k])); }
#pragma unroll
    for (int k = 0; k < NPC - KMIX - 1; ++k) asm volatile("" : "=v"(woff[k]));
    asm volatile("" : "=v"(msrc0)); asm volatile("" : "=v"(mstr));

    NB_TSTAMP(2);
    if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) p.y[0] = 0.f; return; }      // ablation: main loop only
    // ---- the NEXT tile's chunk 0, activations: into stage 0's activation planes (the first 33 600 bytes of the ring), which the
    //      epilogue below leaves alone (its phase slots start behind them) -- they travel while the epilogue computes and stores.
    //      The epilogue keeps THIS tile's coordinates (e_*); the coordinates and piece sources move on to the next tile. ----
    const int e_n = n, e_I0 = I0, e_J0 = J0, e_co0 = co0;
    const unsigned item_next = PERSIST ? __builtin_amdgcn_readfirstlane(item + gridDim.x) : total;
    const bool has_next = item_next < total && !(p.dbg & 64);          // (dbg & 64: no prefetch -- the next tile's chunk 0 goes out after the epilogue)
    if (item_next < total) {
        tile_coords(item_next);
        piece_sources();
        if (has_next) {
            nb_static_for<0, KMIX>([&](auto k) { issue_piece(k, 0, 0); });
            if (mix_isx) issue_piece(std::integral_constant<int, KMIX>{}, 0, 0);
        }
    }
    // ---- epilogue: modconv3x3_up2_h3_kernel's, statement for statement (2 rounds of 16 c_out; comments there) ----
    // (the lane / wave coordinates through an opaque statement: everything the epilogue derives from them -- slot addresses, quad
    //  coordinates, channel offsets: ~45 per-lane values -- is the same for every tile, so the compiler would hoist it all out of the
    //  tile loop and keep it in (spilled) registers across the K loop; recomputed per tile it is a few dozen integer instructions)
    int l31e = l31, lhe = lh, wve = wv;
    asm volatile("" : "+v"(l31e), "+v"(lhe), "+s"(wve));
    const int Wo = 2 * W, Ho = 2 * H;
    constexpr int nquads = TQH * TQW;
    constexpr int Y1P = NBLK * 32;
    // (the phase slots start behind stage 0's activation planes, where the next tile's first chunk is landing)
    f32x4* y4 = reinterpret_cast<f32x4*>(smem_v + (size_t)4 * XPL * 16);
    const float clampv = p.clamp >= 0.f ? p.clamp : __builtin_inff();
    unsigned long long te_w = 0, te_f = 0, te0 = 0;
#pragma unroll
    for (int R = 0; R < 2; ++R) {
        if (p.tstamps) te0 = __builtin_amdgcn_s_memtime();
        if (R) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int gs = 0; gs < 2; ++gs) {
            const int r0 = (2 * R + gs) * 4;
#pragma unroll
            for (int j = 0; j < NBJ; ++j) {
                if (j >= nblk) continue;
                const int pidx = (wve + NW * j) * 32 + l31e;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const f32x16& a_ = acc[j][ph];
                    y4[((gs * 2 + lhe) * 4 + ph) * Y1P + pidx] = f32x4{a_[r0], a_[r0 + 1], a_[r0 + 2], a_[r0 + 3]};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (p.tstamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); te_w += t_ - te0; te0 = t_; }
        auto quad_item = [&](const int wi) {
            const int hq = wi * 32 + l31e;
            const int gs = hq / nquads, qd = hq - gs * nquads;
            const int ti = qd / TQW, tj = qd - ti * TQW;
            if (e_I0 + ti >= H) return;
            const int c4 = 16 * R + 8 * gs + 4 * lhe;  // the lane's four channels within the slice
            const f32x4* ee = y4 + ((gs * 2 + lhe) * 4) * Y1P + ti * PW + tj;
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
            const int qi = e_I0 + ti, qj = e_J0 + tj;
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
                        const int co = e_co0 + c4 + i;
                        if (co < p.c_out) {
                            float* dst = p.y + ((size_t)e_n * p.c_out + co) * ((size_t)Ho * Wo) + (size_t)(2 * qi) * Wo + 2 * qj;
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
                const int cg = e_co0 / 8 + 2 * R + gs;
                const int oy = 2 * qi + lhe, ox = 2 * qj;
                if (qi < H && cg * 8 < p.c_out && !(p.dbg & 1)) {
                    const size_t OHW8 = (size_t)Ho * Wo * 8;
                    _Float16* yn = p.yh2 + ((size_t)e_n * p.c8_next + cg) * 2 * OHW8;
                    const size_t opix8 = ((size_t)oy * Wo + ox) * 8;
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        *reinterpret_cast<u32x4*>(yn + opix8 + px * 8) = u32x4{ha[px][0], ha[px][1], hb[px][0], hb[px][1]};
                        if constexpr (OUTM == 1) {
                            *reinterpret_cast<u32x4*>(yn + OHW8 + opix8 + px * 8) = u32x4{la[px][0], la[px][1], lb[px][0], lb[px][1]};
                        } else {
                            _Float16* lo_xl = p.yh2 + ((size_t)e_n * p.c8_next + (cg & ~1)) * 2 * OHW8 + OHW8 + opix8 + px * 8 + (cg & 1) * 4;
                            *reinterpret_cast<u32x2*>(lo_xl) = u32x2{la[px][0], lb[px][0]};
                            *reinterpret_cast<u32x2*>(lo_xl + 2 * OHW8) = u32x2{la[px][1], lb[px][1]};
                        }
                    }
                }
            }
        };
        constexpr int NWI = nquads * 2 / 32;          // 24 wave-iterations of 32 quads x both channel halves per round
        static_assert(nquads * 2 % 32 == 0 && NWI % NW == 0, "tile quads must fill whole waves");
#pragma unroll
        for (int k = 0; k < NWI / NW; ++k) quad_item(wve + k * NW);
        if (p.tstamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); te_f += t_ - te0; }
    }
    NB_TSTAMP(4);
    if (p.tstamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NB_TSTAMP(5);
        if (threadIdx.x == 0) { unsigned it_ = item; asm volatile("" : "+v"(it_)); p.tstamps[(size_t)it_ * 8 + 3] = (te_w & 0x1fffff) | ((te_f & 0x1fffff) << 21); }
    }
    if (item_next >= total) break;
    // ---- on to the next tile: every wave has read its phase slots -- the rest of chunk 0 (its weights, which land on them) goes out ----
    item = __builtin_amdgcn_readfirstlane(item_next);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    NB_TSTAMP(0);
    piece_sources();            // (again: computed here, the 21 registers of per-lane sources are not alive through the epilogue)
    if (has_next) {
        if (!mix_isx) issue_piece(std::integral_constant<int, KMIX>{}, 0, 0);
        nb_static_for<KMIX + 1, NPC>([&](auto k) { issue_piece(k, 0, 0); });
    } else {
        nb_static_for<0, NPC>([&](auto k) { issue_piece(k, 0, 0); });
    }
  }
#undef NB_TSTAMP
#define NB_TSTAMP(k)                                                                                         \
    do {                                                                                                     \
        if (p.tstamps && threadIdx.x == 0) p.tstamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
}

extern int g_persist_wgs_per_cu;
#ifndef NB_UP2V_PERSIST_DEFAULT
#define NB_UP2V_PERSIST_DEFAULT 1          // (one stream: persistent 0.664-0.668 ms per step, round 5's kernel 0.667-0.672, the loop-less instantiation 0.679-0.686; three streams: equal)
#endif
static int g_up2v_persist = -1;
// developer / test hook: -1 / 1 = persistent workgroups (one per CU, next tile's first chunk prefetched under the epilogue), 0 = one workgroup per tile
extern "C" void nb_debug_set_up2v_persistent(int mode) { g_up2v_persist = mode; }

template <bool F8, int OUTM, bool F6 = false, bool HO = false, bool PERSIST = true>
static int nb_up2v_launch1(const H3Up2Params& p, int n, void* stream) {
    // (epilogue phase slots behind stage 0's activation planes: 33 600 + 122 880 = 156 480 B, 384 more than the ring)
    constexpr size_t lds_ring = (size_t)NST * STAGE * 16, lds_epi = (size_t)4 * XPL * 16 + (size_t)16 * NBLK * 32 * 16;
    constexpr size_t lds = lds_ring > lds_epi ? lds_ring : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_up2v_kernel<F8, OUTM, F6, HO, PERSIST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    // persistent workgroups: one per CU (fewer items than CUs: one each), each walking its items with the next tile's first chunk
    // prefetched under the current tile's epilogue; g_up2v_persist == 0 (test hook): one workgroup per item, as until round 5
    static int ncu = 0;
    if (!ncu) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        ncu = v;
    }
    const int items = p.items;
    const long want = (long)ncu * g_persist_wgs_per_cu;                  // (workgroups per CU: NB_PERSIST_WGS_PER_CU in nb_modconv_h3.hip)
    dim3 grid(PERSIST && items > want ? (unsigned)want : (unsigned)items);
    hipLaunchKernelGGL((modconv3x3_up2v_kernel<F8, OUTM, F6, HO, PERSIST>), grid, dim3(NT), lds, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("modconv3x3_up2v");
    return NB_OK;
}

// shapes this form takes: whole 16-channel chunks (f8 or H2 operands), 32-column tiles of 12 quad rows
bool nb_up2v_eligible(int in_fmt, int c_in, int h, int w) { return in_fmt >= 0 && in_fmt <= 3 && c_in % 16 == 0 && w % TQW == 0 && h >= 8; }

// p as filled in by nb_up2_h3_impl (nb_modconv_h3.hip); tiles are set here
int nb_up2v_launch(H3Up2Params p, int n, int in_fmt, void* stream, unsigned long long* tstamps, int tstamps_cap) {
    NB_REQUIRE(nb_up2v_eligible(in_fmt, p.nchunks * 16, p.h, p.w) && p.co_ld % CO_WG == 0, "modconv3x3_up2v: shape not supported");
    p.tiles_x = p.w / TQW;
    p.tiles_y = (p.h + TQH - 1) / TQH;
    p.slices = (p.c_out + CO_WG - 1) / CO_WG;
    p.items_x = p.tiles_x * p.tiles_y * p.slices;
    p.items = p.items_x * n;
    p.tstamps = (tstamps && (long long)p.items <= tstamps_cap) ? tstamps : nullptr;
    const int outm = p.yh2 ? (p.out_f8 ? 2 : 1) : 0;
    // (persistent workgroups: NB_UP2V_PERSIST_DEFAULT; nb_debug_set_up2v_persistent(0 / 1) = never / always)
    const bool persist = g_up2v_persist >= 0 ? g_up2v_persist != 0 : NB_UP2V_PERSIST_DEFAULT != 0;
    auto go = [&](auto f8_, auto f6_, auto ho_) {
        constexpr bool F8_ = decltype(f8_)::value, F6_ = decltype(f6_)::value, HO_ = decltype(ho_)::value;
        if (persist) return outm == 2 ? nb_up2v_launch1<F8_, 2, F6_, HO_, true>(p, n, stream) : outm == 1 ? nb_up2v_launch1<F8_, 1, F6_, HO_, true>(p, n, stream) : nb_up2v_launch1<F8_, 0, F6_, HO_, true>(p, n, stream);
        return outm == 2 ? nb_up2v_launch1<F8_, 2, F6_, HO_, false>(p, n, stream) : outm == 1 ? nb_up2v_launch1<F8_, 1, F6_, HO_, false>(p, n, stream) : nb_up2v_launch1<F8_, 0, F6_, HO_, false>(p, n, stream);
    };
    using T = std::true_type; using F = std::false_type;
    if (in_fmt == 3) return go(T{}, F{}, T{});
    if (in_fmt == 2) return go(T{}, T{}, F{});
    if (in_fmt) return go(T{}, F{}, F{});
    return go(F{}, F{}, F{});
}
