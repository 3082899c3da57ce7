// Split-f16 ("h3") modulated 3x3 convolution, up = 1, for gfx950: fp32-accurate products on the f16 matrix
// cores at ~5x the fp32-MFMA rate.
//
// Every fp32 operand v is carried as two halves  v = hi + lo,  hi = f16(v), lo = f16(v - hi)  (22 significant
// bits); a product is evaluated as  x*w ~= xh*wh + xl*wh + xh*wl  with three v_mfma_f32_32x32x16_f16 (products
// of two f16 are exact in fp32, accumulation is fp32); the dropped xl*wl term is 2^-22 relative.  End to end
// (14 stacked layers, R=128) this differs from the all-fp32 evaluation by 5e-6 on pixels, against 3e-3 for a
// plain single-f16 evaluation (tools/split_precision_check.py) -- far inside the 1e-3 parity budget.
//
// The style modulation is applied by the PRODUCER of the activation tensor (x * s[n,c] is what gets split and
// stored), so the weights are static and pre-split once at load time; the demodulation d[n,o] stays in the
// fp32 epilogue.  Reference arithmetic: training/networks.py:67-76 (non-fused form of modulated_conv2d).
//
// Activation format "H2":  _Float16 [N][C/8][2 (hi, lo)][H][W][8]   -- 8 channels of one pixel = 16 bytes =
// one lane's MFMA fragment (k = 8 consecutive channels), so halo-tile rows are contiguous 16-byte slots in HBM
// and LDS and every staging transfer is a global_load_lds_dwordx4 (no VGPRs, no alignment games).
// Weight format:  _Float16 [C_in/16][ky 3][kx 3][cg 2][hi/lo 2][C_out_ld][8].
//
// Workgroup = 8 waves as MW (c_out) x 8/MW (pixels), wave tile 64 c_out x 64 pixels (2x2 MFMA tiles, 64
// accumulator registers), i.e. 128 x 256 (MW=2) or 64 x 512 (MW=1) per workgroup; pixel blocks are image rows
// of 32.  K loop: 16-channel chunks x 3 tap rows; the weights of one (chunk, tap row) form a 4-deep LDS ring,
// the activation halo tile of a chunk is double buffered; counted vmcnt + raw s_barrier keep 2 sub-chunks of
// LDS-DMA in flight behind the MFMAs.
#include "nb_h3_common.h"
#include "nb_torgb.h"

// `noise` / `noise_stride_n` of the entry points -> kernel fields: a [n or 1][H][W] tensor, or (NB_NOISE_IN_KERNEL) a host
// NbNoiseSrc describing how the kernel computes the noise itself
static int nb_noise_src_setup(const float* noise, int64_t stride_n, int ho, int wo, const float** k_noise, long long* k_stride, NbNoiseSrcDev* k_src) {
    *k_src = NbNoiseSrcDev{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (stride_n != NB_NOISE_IN_KERNEL) { *k_noise = noise; *k_stride = stride_n; return NB_OK; }
    const NbNoiseSrc* s = reinterpret_cast<const NbNoiseSrc*>(noise);
    if (!s || !s->noise_const_t || !s->noise_lin || !s->noise_strength || ((s->norm_pos != nullptr) == (s->positions != nullptr))) return NB_EINVAL;
    if (s->res != ho || s->res != wo || (s->positions && s->img_resolution < 2)) return NB_EINVAL;
    *k_src = NbNoiseSrcDev{s->noise_const_t, s->noise_lin, s->noise_strength, s->norm_pos, (const long long*)s->positions, s->res, s->img_resolution};
    *k_noise = nullptr; *k_stride = 0;
    return NB_OK;
}

struct H3Params {
    const _Float16* x;      // H2 [n][c8][2][H][W][8], already multiplied by this layer's styles
    const _Float16* wts;    // [nchunks][3][3][2][2][co_ld][8]
    const float* dcoefs;    // [n][c_out]
    const float* noise;     // [n or 1][H][W] or null
    NbNoiseSrcDev nsrc;     // nsrc.const_t != null: the noise is computed here (NbNoiseSrc), `noise` is null
    const float* bias;      // [c_out]
    float* y;               // fp32 NCHW [n][c_out][H][W] or null
    const float* zeros;
    long long noise_stride_n;
    int c8, nchunks, c_out, co_ld, h, w;
    int tiles_x, tiles_y, slices, dbg, stagger_ticks;
    int items, items_x;            // persistent workgroups (8-wave kernel): items = tiles x slices x samples of the launch, items_x = per sample
    float alpha, gain, clamp;
    unsigned long long* tstamps;   // debug: per-workgroup phase timestamps (nb_debug_set_timestamps), else null
    // H2 output (y == null): the result, multiplied by the CONSUMER's styles, goes straight into the consumer's H2
    // input tensor [n][c8_next][2][H][W][8] (channel groups 0 .. c_out/8-1), so no separate packing pass is needed
    _Float16* yh2;
    const float* next_styles;      // [n][next_stride]
    int next_stride, c8_next;
    int out_f8;                    // H2 output in the "f8" operand format (the consumer's corrections run on fp8 MFMAs)
    // fused triad ToRGB (last layer; the workgroup holds all c_out channels of its pixels): tg.c != 0 enables it,
    // y may then be null (the fp32 activations are only needed when somebody taps them)
    TorgbParams tg;
};

static unsigned long long* g_tstamps = nullptr;
static int g_tstamps_cap = 0;
// Debug hook (not part of the product ABI): phase timestamps [workgroup][8] of the next h3 launches, s_memrealtime ticks
extern "C" void nb_debug_set_timestamps(void* buf, int capacity_workgroups) { g_tstamps = (unsigned long long*)buf; g_tstamps_cap = capacity_workgroups; }


// Activation of the split-f16 kernels: lrelu(g (a d + noise + bias)) clamped, with the gain g > 0 folded into the three
// addends (lrelu(g t) = g lrelu(t)).  Every output path of a kernel -- fp32, H2, f8 -- evaluates exactly this expression
// (dg = d g, nbg = bias g + noise g), so that a fused hand-off equals the fp32 output packed afterwards bit for bit.
__device__ __forceinline__ float nb_h3_act(float a, float dg, float nbg, float alpha, float clampv) {
    const float t = __builtin_fmaf(a, dg, nbg);
    // lrelu = max(t, alpha t) for 0 <= alpha <= 1 (the launchers check); clampv = +inf for "no clamp"
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(t, t * alpha, __builtin_inff()), -clampv, clampv);
}



// (the hand-off epilogue of the up=1 kernels, nb_up1_handoff_epilogue, lives in nb_h3_common.h: the geometry encoder's conv kernel uses it too)

// NBW_ = 32-pixel rows per wave: 2 for throughput, 1 (half the pixels per workgroup, twice the workgroups) when the
// launch would otherwise leave most of the chip idle - the batch-1 / interactive configuration.
// V2: the K loop in the form of modconv3x3_up2v_kernel (nb_modconv_up2v.hip) -- LDS-DMA from inline assembly
// (counted lgkmcnt waits), the step's barrier in front of its last two MFMA groups with the next step's first operands read under
// them, fragment reads and DMA pieces dealt out between the MFMAs.  Same per-accumulator order of products: bit-identical.
// F6 (with F8 and V2): the "f6" operand format (nb_h3_common.h) -- the K loop of the f8 V2 form with fp6 correction products.
// PP (with F8 and V2, round 5): the f8 K loop as a PING-PONG of the two waves of a SIMD (waves w and w + 4): a step is a LOAD segment -- all
// of the step's fragment reads and its LDS-DMA pieces, no matrix instruction -- and a COMPUTE segment -- the step's 18 MFMAs back to back,
// every operand in registers --, waves 4-7 run one segment behind waves 0-3, a barrier after every segment: while one wave of a SIMD
// computes, its partner loads.  Same per-accumulator order of products as the other f8 loops: bit-identical.
// HO (with PPK, round 6): "hi only" -- the ping-pong f8 loop WITHOUT its correction products (plain single-f16 evaluation, ~3e-3 from
// fp32: Generator(conv_mode="f16"), a timing data point at the reference's shipped precision, not a parity mode; see modconv3x3_up2v_kernel).
// PERSIST = false: the same body WITHOUT the tile loop (item_next is the list's end at compile time: no next tile, no prefetch, no carried state;
// 208-210 instead of 256 registers, no spills) -- one workgroup per tile as until round 5.  DEFAULT for the ping-pong launches: against round 5's
// kernels linked into the same library (profiles/r06_ab_variants*.txt) the persistent form is 4.4 % SLOWER at 64 channels (b256.conv1 + ToRGB:
// 0.347 -> 0.362 ms per step), equal at 128, and costs the three-streams schedule 1.4 % -- while its own "one workgroup per tile" switch,
// which kept the loop, had said -3 ... -4 %.  The persistent instantiation stays behind nb_debug_set_up1_persistent(1).
template <int MW, bool F8 = false, int NBW_ = 2, bool V2 = false, bool F6 = false, bool PPK = false, bool HO = false, bool PERSIST = true>
__global__ __launch_bounds__(512) void modconv3x3_up1_h3_kernel(const H3Params p) {
    static_assert(!HO || PPK, "the hi-only form is a variant of the ping-pong f8 loop");
    static_assert(!F6 || (F8 && V2), "the f6 form is a variant of the software-pipelined f8 loop");
    static_assert(!PPK || (F8 && V2 && !F6), "the ping-pong form is a variant of the software-pipelined f8 loop");
    NB_TSTAMP(0);
    if (p.out_f8) nb_set_fp16_ovfl();             // f8 hand-off: the fp8 (and f16) conversions saturate
    nb_stagger(p.stagger_ticks, 256);
    constexpr int NW = 8, NWN = NW / MW;          // waves along pixels
    constexpr int MB = 2, NBW = NBW_;             // 32x32 MFMA tiles per wave: 64 c_out x 64 (32) pixels
    constexpr int CO_WG = MW * 64, TH = NWN * NBW, TWP = 34;     // tile rows (32 pixels each), halo tile width
    constexpr int SLOTS = (TH + 2) * TWP;         // 16-byte slots of one (cgroup, hi/lo) plane of the halo tile
    constexpr int PP = (SLOTS + 63) / 64;         // 1-KiB DMA pieces per plane
    constexpr int XPL = PP * 64;                  // slots reserved per plane
    constexpr int NXP = 4 * PP, NXPW = (NXP + NW - 1) / NW;      // activation pieces per chunk / per wave
    constexpr int WSLOTS = 12 * CO_WG;            // slots of one (chunk, tap row) weight sub-chunk: [kx][cg][hl][c_out]
    constexpr int NWP = WSLOTS / 64, NWPW = (NWP + NW - 1) / NW;
    constexpr int WST = 4;                        // weight ring depth
    static_assert(WST == 4, "the ring indices below are written as (t & 3)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h3[];
    h8* xbuf = reinterpret_cast<h8*>(smem_h3);                    // [2][4 planes][XPL]
    h8* wring = xbuf + 2 * 4 * XPL;                               // [WST][WSLOTS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    const int wm = wv / NWN, wn = wv - wm * NWN;  // wave coordinates (c_out, pixel rows)
    const int H = p.h, W = p.w;
    // PERSISTENT workgroups (round 6): gridDim.x = min(items, CUs); workgroup w renders items w, w + gridDim.x, ... of the list
    // [sample][tile][c_out slice] (p.items of them, p.items_x per sample), and the NEXT tile's prologue -- its halo tile and first three
    // weight sub-chunks, ~4 us of LDS-DMA round trip -- is issued BEFORE this tile's epilogue, which works straight from the
    // accumulators (hand-off / fused ToRGB outputs) and leaves the staging LDS alone.
    // XCD-aware order (as in the up=2 kernel below): hardware workgroup ids go round-robin to the 8 XCDs, each with its own L2;
    // renumbered, an XCD works on a contiguous run of a sample's row-major tile list -- whole tile rows -- so that the halo columns
    // and rows neighbouring tiles share are L2 hits instead of second HBM reads (gridDim.x is a multiple of 8 whenever the per-sample
    // list is, so a workgroup's items keep to its XCD)
    typedef const __attribute__((address_space(4))) H3Params* kparams_t;
    auto fresh_params = [&]() -> kparams_t {           // (per-tile reads of the launch parameters: not carried in registers across the K loop)
        kparams_t q = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(q));
        return q;
    };
    int n = 0, y0 = 0, x0 = 0, co0 = 0;
    bool sample_lead = false;                      // exactly one item per sample: it writes the sample's ToRGB colors
    const size_t HW8 = (size_t)H * W * 8;
    auto tile_coords = [&](unsigned l) {
        kparams_t q_ = fresh_params();
        const unsigned gx = (unsigned)q_->items_x;
        unsigned b = l % gx;
        sample_lead = b == 0;
        const unsigned nn = l / gx;
        if (gx % 8 == 0 && gridDim.x % 8 == 0 && !(q_->dbg & 8)) b = (b & 7) * (gx >> 3) + (b >> 3);
        const int slice = b % q_->slices; b /= q_->slices;
        const int tile_x = b % q_->tiles_x; const int tile_y = b / q_->tiles_x;
        n = __builtin_amdgcn_readfirstlane((int)nn); y0 = __builtin_amdgcn_readfirstlane(tile_y * TH);
        x0 = __builtin_amdgcn_readfirstlane(tile_x * 32); co0 = __builtin_amdgcn_readfirstlane(slice * CO_WG);
    };
    unsigned item = __builtin_amdgcn_readfirstlane(blockIdx.x);
    const unsigned total = (unsigned)p.items;
    tile_coords(item);

    // epilogue operands are fetched at the top of a tile, under the prologue DMA: per-channel demodulation / bias into LDS, the
    // lane's noise values into registers (fetching them in the epilogue costs ~10 us of exposed latency per tile)
    __shared__ __attribute__((aligned(16))) float s_dco[CO_WG], s_bias[CO_WG], s_nst[CO_WG];
    __shared__ __attribute__((aligned(16))) float s_tw[3 * CO_WG];
    __shared__ float s_tcol[9], s_tcol01[9];
    float nzr[NBW];
    int tg_n = -1;                                 // sample whose ToRGB tables (weights x styles) are in LDS
    auto load_tile_tables = [&]() {
        kparams_t q_ = fresh_params();
        if (tid < CO_WG) {
            const int co = co0 + tid, c_out = q_->c_out;
            s_dco[tid] = co < c_out ? q_->dcoefs[(size_t)n * c_out + co] : 0.f;
            s_bias[tid] = co < c_out ? q_->bias[co] : 0.f;
            s_nst[tid] = (q_->yh2 && co < c_out) ? q_->next_styles[(size_t)n * q_->next_stride + co] : 0.f;
        }
        // (weights x styles of the sample into LDS; the sample's lead item also writes its colors -- every item of the lead's sample
        //  that this workgroup renders goes through here with the tables in place, so the lead must not be skipped)
        if (p.tg.c && (n != tg_n || sample_lead)) { nb_torgb_setup(p.tg, n, s_tw, s_tcol, s_tcol01, tid, 512, sample_lead); tg_n = n; }
        const float* noise = q_->noise;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
            nzr[nb] = noise ? noise[(size_t)n * q_->noise_stride_n + (size_t)(y0 + wn * NBW + nb) * W + x0 + l31] : 0.f;
        if (q_->nsrc.const_t) {
            // the lane's pixels (row y0 + wn NBW + nb, column x0 + l31): column parameters once, row parameters per row
            const NbNoiseSrcDev nsrc{q_->nsrc.const_t, q_->nsrc.lin, q_->nsrc.strength, q_->nsrc.norm_pos, q_->nsrc.positions, q_->nsrc.res, q_->nsrc.img_res};
            float np0, np1, wy0, wy1;
            int sy0;
            nb_noise_np(nsrc, n, np0, np1);
            nb_noise_axis(nsrc, x0 + l31, np1, sy0, wy0, wy1);
            const float strength = nsrc.strength[0];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                float wx0, wx1;
                int sx0;
                nb_noise_axis(nsrc, y0 + wn * NBW + nb, np0, sx0, wx0, wx1);
                nzr[nb] = nb_noise_value(nsrc, strength, sx0, wx0, wx1, sy0, wy0, wy1);
            }
        }
    };

    // ---- DMA descriptors of this wave's activation pieces ----
    int xsp[NXPW], xpl[NXPW], xdst[NXPW];
    const _Float16* xn = p.x;
#pragma unroll
    for (int i = 0; i < NXPW; ++i) {
        int q = i * NW + wv;
        q = q < NXP ? q : NXP - 1;
        const int pl = q / PP, part = q - pl * PP;      // pl = cg_local*2 + hi/lo
        xpl[i] = __builtin_amdgcn_readfirstlane(pl);
        xdst[i] = __builtin_amdgcn_readfirstlane(pl * XPL + part * 64);
    }
    // the CURRENT tile's per-lane sources (y0, x0, n as tile_coords left them; through an opaque statement: recomputed per tile)
    auto piece_offsets = [&]() {
        int y0_ = y0, x0_ = x0, lane_ = lane;
        asm volatile("" : "+v"(y0_), "+v"(x0_), "+v"(lane_));
        y0_ = __builtin_amdgcn_readfirstlane(y0_); x0_ = __builtin_amdgcn_readfirstlane(x0_);
        kparams_t q_ = fresh_params();
        xn = q_->x + (size_t)n * q_->c8 * 2 * HW8;
#pragma unroll
        for (int i = 0; i < NXPW; ++i) {
            int q = i * NW + wv;
            q = q < NXP ? q : NXP - 1;
            const int part = q % PP;
            const int e = part * 64 + lane_;
            xsp[i] = -1;
            if (e < SLOTS) {
                const int r = e / TWP, c = e - r * TWP;
                const int gy = y0_ - 1 + r, gx = x0_ - 1 + c;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) xsp[i] = (gy * W + gx) * 8;
            }
        }
    };
    piece_offsets();
    auto issue_x = [&](int c, h8* dst) {
#pragma unroll
        for (int i = 0; i < NXPW; ++i) {
            const int cg = 2 * c + (xpl[i] >> 1);
            const _Float16* src = reinterpret_cast<const _Float16*>(p.zeros);
            if (xsp[i] >= 0 && cg < p.c8) src = xn + (size_t)(4 * c + xpl[i]) * HW8 + xsp[i];
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(dst + xdst[i]), 16, 0, 0);
        }
    };
    auto issue_w = [&](int t, h8* dst) {          // t = chunk*3 + ky
#pragma unroll
        for (int i = 0; i < NWPW; ++i) {
            int q = i * NW + wv;
            q = q < NWP ? q : NWP - 1;
            const int e = q * 64 + lane;
            const int row = e / CO_WG, j = e - row * CO_WG;     // row = kx*4 + cg*2 + hl
            const _Float16* src = p.wts + (((size_t)t * 12 + row) * p.co_ld + co0 + j) * 8;
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(dst + q * 64), 16, 0, 0);
        }
    };

    f32x16 acc[MB][NBW];

    const int NC = p.nchunks, T = NC * 3;
    auto clampt = [&](int t) { return t < T ? t : T - 1; };
    // V2: the same pieces from inline assembly (a builtin LDS-DMA anywhere in the kernel pins every fragment-read wait at
    // lgkmcnt(0), see nb_lds_dma16): per-lane source of chunk 0 + per-chunk stride (0 for out-of-image slots: the zero page),
    // weights as a uniform base per step + a lane offset
    const unsigned lds0_v2 = (unsigned)(uintptr_t)NB_LDS_PTR(smem_h3);
    const char* xs0_v2[NXPW];
    unsigned xst_v2[NXPW], wof_v2[NWPW];
    auto piece_sources_v2 = [&]() {
        kparams_t q_ = fresh_params();
        int c0_ = co0, lane_ = lane;
        asm volatile("" : "+v"(c0_), "+v"(lane_));
        c0_ = __builtin_amdgcn_readfirstlane(c0_);
#pragma unroll
        for (int i = 0; i < NXPW; ++i) {
            const bool in = xsp[i] >= 0;
            xs0_v2[i] = in ? reinterpret_cast<const char*>(xn + (size_t)xpl[i] * HW8 + xsp[i]) : reinterpret_cast<const char*>(q_->zeros);
            xst_v2[i] = in ? (unsigned)(4 * HW8 * 2) : 0u;
        }
#pragma unroll
        for (int i = 0; i < NWPW; ++i) {
            int q = i * NW + wv;
            q = q < NWP ? q : NWP - 1;
            const int e = q * 64 + lane_;
            const int row = e / CO_WG, j = e - row * CO_WG;
            wof_v2[i] = (unsigned)(((size_t)row * q_->co_ld + c0_ + j) * 16);
        }
    };
    if constexpr (V2) piece_sources_v2();
    const size_t wstep_v2 = (size_t)12 * p.co_ld * 16;                 // bytes of one (chunk, tap row) weight sub-chunk
    constexpr int WRING_SLOT0 = 2 * 4 * XPL;
    auto issue_x_v2 = [&](auto ii, int c, int buf) {                   // piece ii of chunk c's halo tile into activation buffer buf
        constexpr int i = decltype(ii)::value;
        nb_lds_dma16_m(xs0_v2[i] + (size_t)c * xst_v2[i], lds0_v2 + (unsigned)(buf * 4 * XPL + xdst[i]) * 16u, ~0ull);
    };
    auto issue_w_v2 = [&](auto ii, int t, int slot) {                  // piece ii of weight sub-chunk t into ring slot `slot`
        constexpr int i = decltype(ii)::value;
        int q = i * NW + wv;
        q = q < NWP ? q : NWP - 1;
        nb_lds_dma16_s(reinterpret_cast<const char*>(p.wts) + (size_t)t * wstep_v2, wof_v2[i], lds0_v2 + (unsigned)(WRING_SLOT0 + slot * WSLOTS + q * 64) * 16u);
    };
    // a tile's prologue: chunk 0's halo tile and the first three weight sub-chunks
    auto prologue_issue = [&]() {
        if constexpr (V2) {
            nb_static_for<0, NXPW>([&](auto i) { issue_x_v2(i, 0, 0); });
            nb_static_for<0, NWPW>([&](auto i) { issue_w_v2(i, 0, 0); });
            nb_static_for<0, NWPW>([&](auto i) { issue_w_v2(i, clampt(1), 1); });
            nb_static_for<0, NWPW>([&](auto i) { issue_w_v2(i, clampt(2), 2); });
        } else {
            issue_x(0, xbuf);
            issue_w(0, wring);
            issue_w(clampt(1), wring + WSLOTS);
            issue_w(clampt(2), wring + 2 * WSLOTS);
        }
    };
    NB_TSTAMP(5);                              // (the hand-off epilogues leave slot 5 free: time that passes before the first LDS-DMA piece goes out)
    prologue_issue();
  for (;;) {                                   // one iteration per tile of this workgroup (see PERSISTENT above)
#undef NB_TSTAMP
#define NB_TSTAMP(k) do { if (p.tstamps && threadIdx.x == 0) { unsigned it_ = item; asm volatile("" : "+v"(it_)); p.tstamps[(size_t)it_ * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
    load_tile_tables();
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    // step 0 needs the halo tile and sub-chunk 0 only: sub-chunks 1 and 2 may still be in flight (the loop's invariant)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NWPW) : "memory");
    __builtin_amdgcn_s_barrier();
    NB_TSTAMP(1);

    // fragment addresses (in 16-byte slots)
    const int a_base = lh * 2 * CO_WG + wm * 64 + l31;           // + kx*4*CO_WG + hl*CO_WG + mb*32
    const int b_base = lh * 2 * XPL + (wn * NBW) * TWP + l31;    // + hl*XPL + (nb + ky)*TWP + kx

    unsigned long long t_dma = 0, t_bar = 0;
    const unsigned long long t_loop0 = p.tstamps ? __builtin_amdgcn_s_memtime() : 0;
    if constexpr (F6) {
        // ---- f6 operands: the f8 loop below with the correction products on fp6 MFMAs (32 instead of 64 cycles each) -----------
        // A lane's 32 K values of a correction instruction = ONE tap of ONE pixel / c_out, both terms (one block scale per lane):
        // lane half 0 takes tap 0 of the pair (0, 1), lane half 1 tap 1 -- the lo-slot addresses are per lane --, and its operand is
        // the chunk's two lo slots: six registers of fields (read as 16 + 8 bytes) and the scale dword (4 bytes) = the instruction's
        // scale operand.  Tap 2 has no partner in its step: its corrections run alone every step, lane half 1 multiplying a zero slot
        // (weights side).  Per step and tile 3 f16 + 2 fp6 MFMAs = 160 cycles (f8: 3 x 32 + 1.5 x 64 = 192).  Step t:
        //   part A:  tap 2 main (t-1) | tap-2 corrections (t-1) | tap 0 (t)      fillers: the step's LDS-DMA pieces, its tap-2 fragments
        //   wait (counted) + barrier
        //   part B:  tap 1 (t) | corrections 0 + 1 (t)                           fillers: hi fragments of taps 0, 1 of step t+1;
        //            behind the last fp6 MFMA the lo slots of step t+1
        // Measured (profiles/r05_f6_ab.txt): not faster than the f8 loop -- neither loop is bound by the matrix pipe.
#define NB_SB __builtin_amdgcn_sched_barrier(0)
        __shared__ __attribute__((aligned(16))) h8 s_zero6[1];
        if (tid == 0) s_zero6[0] = h8{};
        __builtin_amdgcn_s_barrier();
        h8 ah0[MB], ah1[MB], ah2[MB], bh0[NBW], bh1[NBW], bh2[NBW];
        v6i al01[MB], bl01[NBW], al2[MB], bl2[NBW];
        int sa01[MB], sb01[NBW], sa2[MB], sb2[NBW];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { ah2[mb] = h8{}; al2[mb] = v6i{}; al01[mb] = v6i{}; sa01[mb] = 0; sa2[mb] = 0; }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) { bh2[nb] = h8{}; bl2[nb] = v6i{}; bl01[nb] = v6i{}; sb01[nb] = 0; sb2[nb] = 0; }
        constexpr int NM = MB * NBW, NF = MB + NBW;
        const int a6 = wm * 64 + l31 + CO_WG, a6_01 = a6 + lh * 4 * CO_WG;          // slot (cg 0, lo) of tap 0 / of the lane's tap; (cg 1, lo) = + 2 CO_WG
        const int b6 = (wn * NBW) * TWP + l31 + XPL, b6_01 = b6 + lh;                 // plane (cg 0, lo); (cg 1, lo) = + 2 XPL
        auto rd_hi = [&](auto i_, h8 (&a)[MB], h8 (&b)[NBW], const h8* wb, const h8* xb, int ky, int kx) {
            constexpr int i = decltype(i_)::value;
#ifdef NB_ABL6_NOREAD
            return;
#endif
            if constexpr (i < MB) a[i] = wb[a_base + kx * 4 * CO_WG + i * 32];
            else b[i - MB] = xb[b_base + (i - MB + ky) * TWP + kx];
        };
        // an fp6 operand: 16 + 8 bytes of fields, 4 bytes of scale (volatile: the compiler must not fuse them into a 16-byte read, which
        // cannot land across the end of the six-register operand)
        auto rd6 = [](v6i& t, int& sc, const h8* slot0, const h8* slot1) {
            // (explicit LDS address space: through a generic pointer the volatile reads become flat loads, which count in vmcnt)
            typedef const volatile __attribute__((address_space(3))) int* lds_vint;
            typedef const volatile __attribute__((address_space(3))) i32x2* lds_vint2;
            const i32x4 q0 = __builtin_bit_cast(i32x4, *slot0);
            const i32x2 q1 = *(lds_vint2)NB_LDS_PTR(slot1);
            sc = ((lds_vint)NB_LDS_PTR(slot1))[2];
            t[0] = q0[0]; t[1] = q0[1]; t[2] = q0[2]; t[3] = q0[3]; t[4] = q1[0]; t[5] = q1[1];
        };
        // fragment i of the pair (0, 1) -- the lane's own tap -- or of the lone tap 2 (weights of lane half 1: the zero slot)
        auto rd_lo01 = [&](auto i_, const h8* wb, const h8* xb, int ky) {
            constexpr int i = decltype(i_)::value;
#ifdef NB_ABL6_NOREAD
            return;
#endif
            if constexpr (i < MB) { const h8* s_ = wb + a6_01 + i * 32; rd6(al01[i], sa01[i], s_, s_ + 2 * CO_WG); }
            else { const h8* s_ = xb + b6_01 + (i - MB + ky) * TWP; rd6(bl01[i - MB], sb01[i - MB], s_, s_ + 2 * XPL); }
        };
        auto rd_lo2 = [&](auto i_, const h8* wb, const h8* xb, int ky) {
            constexpr int i = decltype(i_)::value;
#ifdef NB_ABL6_NOREAD
            return;
#endif
            if constexpr (i < MB) { const h8* s_ = wb + a6 + 8 * CO_WG + i * 32; rd6(al2[i], sa2[i], lh ? &s_zero6[0] : s_, lh ? &s_zero6[0] : s_ + 2 * CO_WG); }
            else { const h8* s_ = xb + b6 + (i - MB + ky) * TWP + 2; rd6(bl2[i - MB], sb2[i - MB], s_, s_ + 2 * XPL); }
        };
        auto group = [&](auto nfill_, auto&& mf, auto&& ff) {
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
            });
        };
        auto mf_f16 = [&](h8 (&a)[MB], h8 (&b)[NBW]) {
            return [&](auto mb_, auto nb_) {
                constexpr int mb = decltype(mb_)::value, nb = decltype(nb_)::value;
#ifdef NB_ABL6_NOMFMA
                return;
#endif
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
            };
        };
        auto wide = [](const v6i& t) { return i32x8{t[0], t[1], t[2], t[3], t[4], t[5], 0, 0}; };
        auto mf_fp6 = [&](v6i (&a)[MB], int (&sa6)[MB], v6i (&b)[NBW], int (&sb6)[NBW]) {
            return [&](auto mb_, auto nb_) {
                constexpr int mb = decltype(mb_)::value, nb = decltype(nb_)::value;
#if defined(NB_ABL6_NOFP6) || defined(NB_ABL6_NOMFMA)
                return;
#endif
                acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wide(a[mb]), wide(b[nb]), acc[mb][nb], 2, 2, 0, sa6[mb], 0, sb6[nb]);
            };
        };
        auto step = [&](auto ky_, int t, int c) {
            constexpr int KY = decltype(ky_)::value;
            constexpr int NDMA = NWPW + (KY == 0 ? NXPW : 0);
            const h8* xb = xbuf + (c & 1) * 4 * XPL;
            const h8* wb = wring + (t & 3) * WSLOTS;
            const int tn = t + 1, cn = KY == 2 ? c + 1 : c;
            constexpr int KYN = KY == 2 ? 0 : KY + 1;
            const h8* xbn = xbuf + (cn & 1) * 4 * XPL;
            const h8* wbn = wring + (tn & 3) * WSLOTS;
            const int t3 = clampt(t + 3), c1 = c + 1 < NC ? c + 1 : NC - 1;
            auto dma = [&](auto i_) {
                constexpr int i = decltype(i_)::value;
#ifdef NB_ABL6_NODMA
                return;
#endif
                if constexpr (i < NWPW) issue_w_v2(i_, t3, (t + 3) & 3);
                else issue_x_v2(std::integral_constant<int, i - NWPW>{}, c1, (c + 1) & 1);
            };
            NB_SB;
            // part A
            group(std::integral_constant<int, NDMA>{}, mf_f16(ah2, bh2), dma);                       // tap 2 of step t-1
            group(std::integral_constant<int, NF>{}, mf_fp6(al2, sa2, bl2, sb2),                      // its corrections
                  [&](auto i_) { rd_hi(i_, ah2, bh2, wb, xb, KY, 2); });
            group(std::integral_constant<int, NF>{}, mf_f16(ah0, bh0), [&](auto i_) { rd_lo2(i_, wb, xb, KY); });       // tap 0
#ifdef NB_ABL6_NODMA
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
            if constexpr (KY == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NWPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NWPW + NXPW) : "memory");
#endif
            NB_SB;
            // part B
            group(std::integral_constant<int, NF>{}, mf_f16(ah1, bh1),                                // tap 1
                  [&](auto i_) { rd_hi(i_, ah0, bh0, wbn, xbn, KYN, 0); });
            group(std::integral_constant<int, NF>{}, mf_fp6(al01, sa01, bl01, sb01),                  // corrections of taps 0 + 1
                  [&](auto i_) { rd_hi(i_, ah1, bh1, wbn, xbn, KYN, 1); });
            nb_static_for<0, NF>([&](auto i_) { rd_lo01(i_, wbn, xbn, KYN); });
            NB_SB;
        };
        // operands of step 0
        nb_static_for<0, NF>([&](auto i_) { rd_hi(i_, ah0, bh0, wring, xbuf, 0, 0); });
        nb_static_for<0, NF>([&](auto i_) { rd_hi(i_, ah1, bh1, wring, xbuf, 0, 1); });
        nb_static_for<0, NF>([&](auto i_) { rd_lo01(i_, wring, xbuf, 0); });
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
        for (int c = 0; c < NC; ++c) { step(K0{}, 3 * c, c); step(K1{}, 3 * c + 1, c); step(K2{}, 3 * c + 2, c); }
        NB_SB;
        // the last step's tap 2 and its corrections
        nb_static_for<0, NM>([&](auto k_) { constexpr int k = decltype(k_)::value; mf_f16(ah2, bh2)(std::integral_constant<int, k / NBW>{}, std::integral_constant<int, k % NBW>{}); NB_SB; });
        nb_static_for<0, NM>([&](auto k_) { constexpr int k = decltype(k_)::value; mf_fp6(al2, sa2, bl2, sb2)(std::integral_constant<int, k / NBW>{}, std::integral_constant<int, k % NBW>{}); NB_SB; });
#undef NB_SB
    } else if constexpr (F8 && V2) {
        // ---- f8 operands, software-pipelined over the barrier ------------------------------------------------------
        // Step t = (chunk c, tap row ky) multiplies, per accumulator tile and in this order: tap 2 of step t-1 (f16), on even t
        // the tap-2 corrections of steps t-2 and t-1 (ONE fp8 K = 64), tap 0 (f16), tap 1 (f16), the corrections of taps 0 + 1
        // (fp8) -- the order of the loop below.  What differs is where the barrier stands and when operands are read:
        //   part A:  tap 2 (t-1) | [tap-2 corrections] | tap 0 (t)      fillers: the step's LDS-DMA pieces, its tap-2 fragments
        //   wait (counted) + barrier: every wave has read ALL of stage t; stage t+1 has landed
        //   part B:  tap 1 (t) | corrections 0 + 1 (t)                  fillers: hi fragments of taps 0, 1 of step t+1;
        //            behind the last fp8 MFMA the lo halves of step t+1 (their tuples were its operands)
        // so no step opens with a burst of sixteen reads behind a barrier that both waves of a SIMD reach together.
        // All compile-time: the loop body is two chunks = six steps (tap row and the parity that selects the tuple quad).
#define NB_SB __builtin_amdgcn_sched_barrier(0)
#define NB_Q(v, q, src) { const i32x4 t_ = __builtin_bit_cast(i32x4, (src)); v[4 * (q)] = t_[0]; v[4 * (q) + 1] = t_[1]; v[4 * (q) + 2] = t_[2]; v[4 * (q) + 3] = t_[3]; }
        const int sa = lh ? 116 : 127, sb = lh ? 129 : 118;
        h8 ah0[MB], ah1[MB], ah2[MB], bh0[NBW], bh1[NBW], bh2[NBW];
        i32x8 al01[MB], bl01[NBW], al2[MB], bl2[NBW];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { ah2[mb] = h8{}; al2[mb] = i32x8{}; al01[mb] = i32x8{}; }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) { bh2[nb] = h8{}; bl2[nb] = i32x8{}; bl01[nb] = i32x8{}; }
        constexpr int NM = MB * NBW, NF = MB + NBW;       // MFMAs per group; fragment reads per (tap, hi or lo)
        // one fragment read: index i < MB: weights of block i, else activations of pixel row i - MB; hl = 0 hi, 1 lo
        auto rd_hi = [&](auto i_, h8 (&a)[MB], h8 (&b)[NBW], const h8* wb, const h8* xb, int ky, int kx) {
            constexpr int i = decltype(i_)::value;
            if constexpr (i < MB) a[i] = wb[a_base + kx * 4 * CO_WG + i * 32];
            else b[i - MB] = xb[b_base + (i - MB + ky) * TWP + kx];
        };
        auto rd_lo = [&](auto i_, auto q_, i32x8 (&a)[MB], i32x8 (&b)[NBW], const h8* wb, const h8* xb, int ky, int kx) {
            constexpr int i = decltype(i_)::value, q = decltype(q_)::value;
            if constexpr (i < MB) { NB_Q(a[i], q, wb[a_base + kx * 4 * CO_WG + CO_WG + i * 32]); }
            else { NB_Q(b[i - MB], q, xb[b_base + XPL + (i - MB + ky) * TWP + kx]); }
        };
        // a group of NM MFMAs (tile k = (k / NBW, k % NBW)) with NFILL fillers dealt evenly into the gaps behind them
        auto group = [&](auto nfill_, auto&& mf, auto&& ff) {
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
            });
        };
        // (q: which half of the accumulator tile the NB_MOCK16 timing build writes; unused otherwise)
        auto mf_f16 = [&](h8 (&a)[MB], h8 (&b)[NBW], int q = 0) {
            return [&, q](auto mb_, auto nb_) {
                constexpr int mb = decltype(mb_)::value, nb = decltype(nb_)::value;
                acc[mb][nb] = NB_MFMA_F16(a[mb], b[nb], acc[mb][nb], q);
            };
        };
        auto mf_fp8 = [&](i32x8 (&a)[MB], i32x8 (&b)[NBW], int q = 0) {
            return [&, q](auto mb_, auto nb_) {
                constexpr int mb = decltype(mb_)::value, nb = decltype(nb_)::value;
                if constexpr (!HO) acc[mb][nb] = NB_MFMA_FP8(a[mb], b[nb], acc[mb][nb], sa, sb, q);
            };
        };
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
        if constexpr (PPK) {
            // ---- ping-pong: LOAD(t) = the pieces of sub-chunk t + 3 (on KY = 0 also the next chunk's halo tile), then every fragment of
            //      step t; COMPUTE(t) = tap 0, tap 1, corrections 0 + 1, tap 2, on odd t the tap-2 corrections of steps t-1 and t.
            //   segment      0        1        2        3        4
            //   waves 0-3   LOAD 0   COMP 0   LOAD 1   COMP 1   LOAD 2  ...        (one barrier after every segment)
            //   waves 4-7   (wait)   LOAD 0   COMP 0   LOAD 1   COMP 1  ...
            // Stage t (weight ring slot t & 3, halo tile of its chunk) is read in segments 2t (waves 0-3) and 2t + 1 (waves 4-7); the
            // barrier that ends segment 2t + 1 is the one in front of which every wave waits for its pieces up to sub-chunk t + 1
            // (all but the youngest two sub-chunks, + the halo tile unless it is due), so stage t + 1 has landed when segment 2t + 2
            // starts to read it, and slot (t + 3) & 3 = (t - 1) & 3, which LOAD(t) refills, was last read in segment 2t - 1.
            auto load_seg = [&](auto ky_, auto odd_, int t, int c) {
                constexpr int KY = decltype(ky_)::value, ODD = decltype(odd_)::value;
                const h8* xb = xbuf + (c & 1) * 4 * XPL;
                const h8* wb = wring + (t & 3) * WSLOTS;
                NB_SB;
                nb_static_for<0, NF>([&](auto i_) { rd_hi(i_, ah0, bh0, wb, xb, KY, 0); });
                nb_static_for<0, NF>([&](auto i_) { rd_hi(i_, ah1, bh1, wb, xb, KY, 1); });
                nb_static_for<0, NF>([&](auto i_) { rd_lo(i_, std::integral_constant<int, 0>{}, al01, bl01, wb, xb, KY, 0); });
                nb_static_for<0, NF>([&](auto i_) { rd_lo(i_, std::integral_constant<int, 1>{}, al01, bl01, wb, xb, KY, 1); });
                nb_static_for<0, NF>([&](auto i_) { rd_hi(i_, ah2, bh2, wb, xb, KY, 2); });
                nb_static_for<0, NF>([&](auto i_) { rd_lo(i_, std::integral_constant<int, ODD>{}, al2, bl2, wb, xb, KY, 2); });
                NB_SB;
            };
            // COMPUTE(t): the step's MFMAs; its LDS-DMA pieces (sub-chunk t + 3, on KY = 0 the next chunk's halo tile) ride in the gaps
            // behind the first two groups (slot (t + 3) & 3 was last read in segment 2t - 1; this is segment 2t + 1 or 2t + 2)
            auto comp_seg = [&](auto ky_, auto odd_, int t, int c) {
                constexpr int KY = decltype(ky_)::value, ODD = decltype(odd_)::value;
                constexpr int NDMA = NWPW + (KY == 0 ? NXPW : 0);
                const int t3 = clampt(t + 3), c1 = c + 1 < NC ? c + 1 : NC - 1;
                auto dma = [&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    if constexpr (i < NWPW) issue_w_v2(i_, t3, (t + 3) & 3);
                    else issue_x_v2(std::integral_constant<int, i - NWPW>{}, c1, (c + 1) & 1);
                };
                NB_SB;
                // (a piece costs the wave ~60 cycles of issue: behind a 64-cycle fp8 MFMA it is hidden, behind a 32-cycle f16 one the pipe waits)
                constexpr int N1 = NDMA < NM ? NDMA : NM, R = NDMA - N1;
                auto dma_r = [&](auto i_) { dma(std::integral_constant<int, decltype(i_)::value + N1>{}); };
                group(std::integral_constant<int, 0>{}, mf_f16(ah0, bh0, 0), [](auto) {});
                group(std::integral_constant<int, 0>{}, mf_f16(ah1, bh1, 1), [](auto) {});
                group(std::integral_constant<int, N1>{}, mf_fp8(al01, bl01, 0), dma);
                group(std::integral_constant<int, ODD ? 0 : R>{}, mf_f16(ah2, bh2, 1), dma_r);
                if constexpr (ODD) group(std::integral_constant<int, R>{}, mf_fp8(al2, bl2, 0), dma_r);
                NB_SB;
            };
            const bool grp_b = wv >= 4;
            // one step of a wave: LOAD, sync, COMPUTE, sync.  The counted vmcnt wait stands in front of the barrier that ends an ODD
            // segment -- behind COMPUTE(t) for waves 0-3 (pieces up to sub-chunk t + 3 issued: the youngest two sub-chunks may be in flight,
            // and the halo tile unless it is due), behind LOAD(t) for waves 4-7 (issued up to t + 2: the youngest one).
            auto pp_step = [&](auto b_, auto ky_, auto odd_, int t, int c) {
                constexpr bool B = decltype(b_)::value;
                constexpr int KY = decltype(ky_)::value;
                constexpr int NVA = KY == 2 ? 2 * NWPW : 2 * NWPW + NXPW, NVB = KY == 1 ? NWPW + NXPW : NWPW;
#ifdef NB_PP_STAMPS
                const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
                load_seg(ky_, odd_, t, c);
#ifdef NB_PP_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long s1 = __builtin_amdgcn_s_memtime();
#endif
                if constexpr (B) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NVB) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef NB_PP_STAMPS
                const unsigned long long s2 = __builtin_amdgcn_s_memtime();
#endif
                comp_seg(ky_, odd_, t, c);
#ifdef NB_PP_STAMPS
                const unsigned long long s3 = __builtin_amdgcn_s_memtime();
#endif
                if constexpr (B) asm volatile("s_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NVA) : "memory");
#ifdef NB_PP_STAMPS
                const unsigned long long s4 = __builtin_amdgcn_s_memtime();
                t_dma += ((s1 - s0) & 0xffffffffull) | ((s3 - s2) << 32);          // (load | compute) cycles
                t_bar += ((s2 - s1) & 0xffffffffull) | ((s4 - s3) << 32);          // (wait behind load | wait behind compute)
#endif
            };
            auto pp_loop = [&](auto b_) {
                int c = 0;
                for (; c + 1 < NC; c += 2) {
                    pp_step(b_, K0{}, K0{}, 3 * c, c); pp_step(b_, K1{}, K1{}, 3 * c + 1, c); pp_step(b_, K2{}, K0{}, 3 * c + 2, c);
                    pp_step(b_, K0{}, K1{}, 3 * c + 3, c + 1); pp_step(b_, K1{}, K0{}, 3 * c + 4, c + 1); pp_step(b_, K2{}, K1{}, 3 * c + 5, c + 1);
                }
                if (c < NC) { pp_step(b_, K0{}, K0{}, 3 * c, c); pp_step(b_, K1{}, K1{}, 3 * c + 1, c); pp_step(b_, K2{}, K0{}, 3 * c + 2, c); }
            };
            if (grp_b) {
                asm volatile("s_barrier" ::: "memory");
                pp_loop(std::true_type{});
            } else {
                pp_loop(std::false_type{});
                asm volatile("s_barrier" ::: "memory");
            }
            NB_SB;
            if (T & 1) {                                   // an odd number of steps: the last tuple of tap-2 corrections holds one tap only
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int r = 4; r < 8; ++r) al2[mb][r] = 0;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int r = 4; r < 8; ++r) bl2[nb][r] = 0;
                group(std::integral_constant<int, 0>{}, mf_fp8(al2, bl2), [](auto) {});
            }
        } else {
        // step t: KY = tap row, ODD = t & 1 (compile time); reads stage (c, slot t & 3), issues sub-chunk t + 3 (and on KY = 0 the
        // next chunk's halo tile), pre-reads step t + 1
        auto step = [&](auto ky_, auto odd_, int t, int c) {
            constexpr int KY = decltype(ky_)::value, ODD = decltype(odd_)::value;
            constexpr int NDMA = NWPW + (KY == 0 ? NXPW : 0);
            const h8* xb = xbuf + (c & 1) * 4 * XPL;
            const h8* wb = wring + (t & 3) * WSLOTS;
            const int tn = t + 1, cn = KY == 2 ? c + 1 : c;
            constexpr int KYN = KY == 2 ? 0 : KY + 1;
            const h8* xbn = xbuf + (cn & 1) * 4 * XPL;
            const h8* wbn = wring + (tn & 3) * WSLOTS;
            const int t3 = clampt(t + 3), c1 = c + 1 < NC ? c + 1 : NC - 1;
            auto dma = [&](auto i_) {
                constexpr int i = decltype(i_)::value;
                if constexpr (i < NWPW) issue_w_v2(i_, t3, (t + 3) & 3);
                else issue_x_v2(std::integral_constant<int, i - NWPW>{}, c1, (c + 1) & 1);
            };
            NB_SB;
            // part A
            group(std::integral_constant<int, NDMA>{}, mf_f16(ah2, bh2), dma);                       // tap 2 of step t-1
            if constexpr (!ODD) {
                group(std::integral_constant<int, NF>{}, mf_fp8(al2, bl2),                            // tap-2 corrections of t-2, t-1
                      [&](auto i_) { rd_hi(i_, ah2, bh2, wb, xb, KY, 2); });
                group(std::integral_constant<int, NF>{}, mf_f16(ah0, bh0),                            // tap 0
                      [&](auto i_) { rd_lo(i_, std::integral_constant<int, 0>{}, al2, bl2, wb, xb, KY, 2); });
            } else {
                group(std::integral_constant<int, 2 * NF>{}, mf_f16(ah0, bh0), [&](auto i_) {         // tap 0
                    constexpr int i = decltype(i_)::value;
                    if constexpr (i < NF) rd_hi(i_, ah2, bh2, wb, xb, KY, 2);
                    else rd_lo(std::integral_constant<int, i - NF>{}, std::integral_constant<int, 1>{}, al2, bl2, wb, xb, KY, 2);
                });
            }
            // everything issued before sub-chunk t-1 has landed (it is what step t+1 reads); all reads of stage t are done
            if constexpr (KY == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NWPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NWPW + NXPW) : "memory");
            NB_SB;
            // part B
            group(std::integral_constant<int, NF>{}, mf_f16(ah1, bh1),                                // tap 1
                  [&](auto i_) { rd_hi(i_, ah0, bh0, wbn, xbn, KYN, 0); });
            group(std::integral_constant<int, NF>{}, mf_fp8(al01, bl01),                              // corrections of taps 0 + 1
                  [&](auto i_) { rd_hi(i_, ah1, bh1, wbn, xbn, KYN, 1); });
            nb_static_for<0, NF>([&](auto i_) { rd_lo(i_, std::integral_constant<int, 0>{}, al01, bl01, wbn, xbn, KYN, 0); });
            nb_static_for<0, NF>([&](auto i_) { rd_lo(i_, std::integral_constant<int, 1>{}, al01, bl01, wbn, xbn, KYN, 1); });
            NB_SB;
        };
        // operands of step 0
        nb_static_for<0, NF>([&](auto i_) { rd_hi(i_, ah0, bh0, wring, xbuf, 0, 0); });
        nb_static_for<0, NF>([&](auto i_) { rd_hi(i_, ah1, bh1, wring, xbuf, 0, 1); });
        nb_static_for<0, NF>([&](auto i_) { rd_lo(i_, std::integral_constant<int, 0>{}, al01, bl01, wring, xbuf, 0, 0); });
        nb_static_for<0, NF>([&](auto i_) { rd_lo(i_, std::integral_constant<int, 1>{}, al01, bl01, wring, xbuf, 0, 1); });
        int c = 0;
        for (; c + 1 < NC; c += 2) {                   // t = 3 c is even here
            step(K0{}, K0{}, 3 * c, c); step(K1{}, K1{}, 3 * c + 1, c); step(K2{}, K0{}, 3 * c + 2, c);
            step(K0{}, K1{}, 3 * c + 3, c + 1); step(K1{}, K0{}, 3 * c + 4, c + 1); step(K2{}, K1{}, 3 * c + 5, c + 1);
        }
        if (c < NC) { step(K0{}, K0{}, 3 * c, c); step(K1{}, K1{}, 3 * c + 1, c); step(K2{}, K0{}, 3 * c + 2, c); }
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
        }
#undef NB_Q
#undef NB_SB
    } else if constexpr (V2) {
        // ---- H2 operands (the `h3` arithmetic: hi x hi, hi x lo, lo x hi per tap), software-pipelined over the barrier like the
        //      f8 loop above.  Per accumulator tile the nine products of a step arrive in the order of the loop below (tap by tap:
        //      hh, hl, lh): bit-identical.  Step t:
        //   part A:  taps 0 and 1 (six groups of four MFMAs: one product kind of one tap over the four tiles -- no two MFMAs in a row on
        //            one accumulator)            fillers: the step's LDS-DMA pieces, its tap-2 fragments
        //   wait (counted) + barrier: every wave has read ALL of stage t; stage t+1 has landed
        //   part B:  tap 2 (three groups)         fillers: the sixteen fragments of taps 0 and 1 of step t+1
#define NB_SB __builtin_amdgcn_sched_barrier(0)
        h8 ah[3][MB], al[3][MB], bh[3][NBW], bl[3][NBW];           // [tap][block]
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) { ah[k][mb] = h8{}; al[k][mb] = h8{}; }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) { bh[k][nb] = h8{}; bl[k][nb] = h8{}; }
        }
        constexpr int NM = MB * NBW, NF = MB + NBW;
        // fragment read j of a tap (0 .. 2 NF - 1): hi fragments (weights of block i, then activations of pixel row i - MB), then lo
        auto rd = [&](auto j_, auto kx_, const h8* wb, const h8* xb, int ky) {
            constexpr int j = decltype(j_)::value, kx = decltype(kx_)::value, hl = j / NF, i = j % NF;
            if constexpr (i < MB) (hl ? al : ah)[kx][i] = wb[a_base + kx * 4 * CO_WG + hl * CO_WG + i * 32];
            else (hl ? bl : bh)[kx][i - MB] = xb[b_base + hl * XPL + (i - MB + ky) * TWP + kx];
        };
        // group g of NG: NM MFMAs of one product kind with the fillers [g L / NG, (g + 1) L / NG) of a list of L dealt into its gaps
        auto group = [&](auto g_, auto ng_, auto l_, auto&& mf, auto&& item) {
            constexpr int g = decltype(g_)::value, NG = decltype(ng_)::value, L = decltype(l_)::value;
            constexpr int lo = g * L / NG, hi = (g + 1) * L / NG, NFILL = hi - lo;
            nb_static_for<0, NM>([&](auto k_) {
                constexpr int k = decltype(k_)::value;
                mf(std::integral_constant<int, k / NBW>{}, std::integral_constant<int, k % NBW>{});
                NB_SB;
                nb_static_for<0, NFILL>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    if constexpr (i * NM / (NFILL > 0 ? NFILL : 1) == k) item(std::integral_constant<int, lo + i>{});
                });
                NB_SB;
            });
        };
        // product kind q of tap kx: 0 = hi x hi, 1 = hi x lo, 2 = lo x hi
        auto mf = [&](auto kx_, auto q_) {
            return [&](auto mb_, auto nb_) {
                constexpr int kx = decltype(kx_)::value, q = decltype(q_)::value, mb = decltype(mb_)::value, nb = decltype(nb_)::value;
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q == 2 ? al[kx][mb] : ah[kx][mb], q == 1 ? bl[kx][nb] : bh[kx][nb], acc[mb][nb], 0, 0, 0);
            };
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
        using I6 = std::integral_constant<int, 6>;
        auto step = [&](auto ky_, int t, int c) {
            constexpr int KY = decltype(ky_)::value;
            constexpr int NDMA = NWPW + (KY == 0 ? NXPW : 0);
            const h8* xb = xbuf + (c & 1) * 4 * XPL;
            const h8* wb = wring + (t & 3) * WSLOTS;
            const int tn = t + 1, cn = KY == 2 ? c + 1 : c;
            constexpr int KYN = KY == 2 ? 0 : KY + 1;
            const h8* xbn = xbuf + (cn & 1) * 4 * XPL;
            const h8* wbn = wring + (tn & 3) * WSLOTS;
            const int t3 = clampt(t + 3), c1 = c + 1 < NC ? c + 1 : NC - 1;
            // part A's fillers: the pieces, then this step's tap-2 fragments
            using LA = std::integral_constant<int, NDMA + 2 * NF>;
            auto item_a = [&](auto i_) {
                constexpr int i = decltype(i_)::value;
                if constexpr (i < NWPW) issue_w_v2(i_, t3, (t + 3) & 3);
                else if constexpr (i < NDMA) issue_x_v2(std::integral_constant<int, i - NWPW>{}, c1, (c + 1) & 1);
                else rd(std::integral_constant<int, i - NDMA>{}, I2{}, wb, xb, KY);
            };
            // part B's: taps 0 and 1 of step t + 1
            using LB = std::integral_constant<int, 4 * NF>;
            auto item_b = [&](auto i_) {
                constexpr int i = decltype(i_)::value;
                if constexpr (i < 2 * NF) rd(i_, I0{}, wbn, xbn, KYN);
                else rd(std::integral_constant<int, i - 2 * NF>{}, I1{}, wbn, xbn, KYN);
            };
            NB_SB;
            group(I0{}, I6{}, LA{}, mf(I0{}, I0{}), item_a); group(I1{}, I6{}, LA{}, mf(I0{}, I1{}), item_a); group(I2{}, I6{}, LA{}, mf(I0{}, I2{}), item_a);
            group(I3{}, I6{}, LA{}, mf(I1{}, I0{}), item_a); group(I4{}, I6{}, LA{}, mf(I1{}, I1{}), item_a); group(I5{}, I6{}, LA{}, mf(I1{}, I2{}), item_a);
            // everything issued before sub-chunk t-1 has landed (it is what step t+1 reads); all reads of stage t are done
            if constexpr (KY == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NWPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NWPW + NXPW) : "memory");
            NB_SB;
            group(I0{}, I3{}, LB{}, mf(I2{}, I0{}), item_b); group(I1{}, I3{}, LB{}, mf(I2{}, I1{}), item_b); group(I2{}, I3{}, LB{}, mf(I2{}, I2{}), item_b);
            NB_SB;
        };
        // operands of step 0
        nb_static_for<0, 2 * NF>([&](auto j_) { rd(j_, I0{}, wring, xbuf, 0); });
        nb_static_for<0, 2 * NF>([&](auto j_) { rd(j_, I1{}, wring, xbuf, 0); });
        for (int c = 0; c < NC; ++c) { step(I0{}, 3 * c, c); step(I1{}, 3 * c + 1, c); step(I2{}, 3 * c + 2, c); }
        NB_SB;
#undef NB_SB
    } else if constexpr (F8) {
        // ---- f8 operands: explicitly ordered, software-pipelined step -------------------------------------------
        // The compiler, left alone, sinks every fragment read to just before its first use and waits with lgkmcnt(0)
        // right after issuing it (five exposed LDS latencies per step with both waves of a SIMD in lockstep).  Here
        // every statement group is followed by a scheduling fence, so the program order below is the issue order:
        //     after barrier(t-1):  reads tap 0, tap 1 of step t | tap 2 main of step t-1 (4 x 32 cycles of cover) |
        //                          on even t: tap-2 corrections of steps t-2 and t-1 together (4 x 64 cycles)
        //     tap 0 main | reads tap 2 | tap 1 main | corrections taps 0+1 | DMA issues | wait | barrier(t)
        // i.e. a fragment is read at least one matrix group (128-256 cycles) before the group that consumes it.
        // The third tap of a row has no partner within its step, so its corrections ride with the third tap of the NEXT step
        // in one K = 64 fp8 instruction (quad 0 = even step, quad 1 = odd step): 7 fp8 instructions per two steps instead of
        // 8, 768 instead of 896 matrix cycles per step and tile row.
#define NB_SB __builtin_amdgcn_sched_barrier(0)
#define NB_Q(v, q, src) { const i32x4 t_ = __builtin_bit_cast(i32x4, (src)); v[4 * (q)] = t_[0]; v[4 * (q) + 1] = t_[1]; v[4 * (q) + 2] = t_[2]; v[4 * (q) + 3] = t_[3]; }
        const int sa = lh ? 116 : 127, sb = lh ? 129 : 118;
        h8 ah0[MB], ah1[MB], ah2[MB], bh0[NBW], bh1[NBW], bh2[NBW];
        i32x8 al01[MB], bl01[NBW], al2[MB], bl2[NBW];          // fp8 operand tuples: (tap 0 | tap 1), (tap 2 of an even step | of the odd step after it)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 8; ++r) al2[mb][r] = 0;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 8; ++r) bl2[nb][r] = 0;
        auto rd_tap01 = [&](const h8* wb, const h8* xb, int ky, int kx) {       // kx = 0 or 1: hi fragments + quad kx of the lo tuples
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                (kx ? ah1 : ah0)[mb] = wb[a_base + kx * 4 * CO_WG + mb * 32];
                if (kx) { NB_Q(al01[mb], 1, wb[a_base + kx * 4 * CO_WG + CO_WG + mb * 32]); } else { NB_Q(al01[mb], 0, wb[a_base + CO_WG + mb * 32]); }
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                (kx ? bh1 : bh0)[nb] = xb[b_base + (nb + ky) * TWP + kx];
                if (kx) { NB_Q(bl01[nb], 1, xb[b_base + XPL + (nb + ky) * TWP + kx]); } else { NB_Q(bl01[nb], 0, xb[b_base + XPL + (nb + ky) * TWP]); }
            }
            NB_SB;
        };
        auto rd_tap2 = [&](const h8* wb, const h8* xb, int ky, int odd) {       // odd = quad of the tap-2 tuples
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                ah2[mb] = wb[a_base + 2 * 4 * CO_WG + mb * 32];
                if (odd) { NB_Q(al2[mb], 1, wb[a_base + 2 * 4 * CO_WG + CO_WG + mb * 32]); } else { NB_Q(al2[mb], 0, wb[a_base + 2 * 4 * CO_WG + CO_WG + mb * 32]); }
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                bh2[nb] = xb[b_base + (nb + ky) * TWP + 2];
                if (odd) { NB_Q(bl2[nb], 1, xb[b_base + XPL + (nb + ky) * TWP + 2]); } else { NB_Q(bl2[nb], 0, xb[b_base + XPL + (nb + ky) * TWP + 2]); }
            }
            NB_SB;
        };
        auto main4 = [&](h8 (&a)[MB], h8 (&b)[NBW]) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) { acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mb], b[nb], acc[mb][nb], 0, 0, 0); NB_SB; }
        };
        auto corr4 = [&](i32x8 (&a)[MB], i32x8 (&b)[NBW]) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[mb], b[nb], acc[mb][nb], 0, 0, 0, sa, 0, sb);
                    NB_SB;
                }
        };
        // the loop runs over PAIRS of steps (t = 2 tp, 2 tp + 1) so that the tuple quad is a compile-time constant
        auto step = [&](int t, int odd, bool first) {
            const int c = t / 3, ky = t - 3 * c;
            const h8* xb = xbuf + (c & 1) * 4 * XPL;
            const h8* wb = wring + (t & 3) * WSLOTS;
            NB_SB;
            rd_tap01(wb, xb, ky, 0);
            rd_tap01(wb, xb, ky, 1);
            if (!first) main4(ah2, bh2);               // tap 2 of the previous step
            if (!odd && !first) corr4(al2, bl2);       // tap-2 corrections of the two previous steps
            main4(ah0, bh0);
            rd_tap2(wb, xb, ky, odd);
            main4(ah1, bh1);
            corr4(al01, bl01);
            // keep the LDS-DMA stream 3 sub-chunks ahead (past the end: harmless re-copies keep the counts uniform)
            issue_w(clampt(t + 3), wring + ((t + 3) & 3) * WSLOTS);
            if (ky == 0) issue_x(c + 1 < NC ? c + 1 : NC - 1, xbuf + ((c + 1) & 1) * 4 * XPL);
            NB_SB;
            // everything issued before sub-chunk t-1 must have landed (it is what sub-chunk t+1 reads)
            if (ky == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NWPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NWPW + NXPW) : "memory");
            __builtin_amdgcn_s_barrier();
        };
        for (int t = 0; t < T; t += 2) {
            step(t, 0, t == 0);
            if (t + 1 < T) step(t + 1, 1, false);
        }
        NB_SB;
        main4(ah2, bh2);                               // the last step's tap 2
        if (T & 1) {                                   // odd number of steps: the last tuple holds one tap only
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
    for (int c = 0; c < NC; ++c) {
        const h8* xb = xbuf + (c & 1) * 4 * XPL;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int t = c * 3 + ky;
            // keep the LDS-DMA stream 3 sub-chunks ahead (past the end: harmless re-copies keep the counts uniform)
            if (!(p.dbg & 2)) {
                issue_w(clampt(t + 3), wring + ((t + 3) & 3) * WSLOTS);
                if (ky == 0) issue_x(c + 1 < NC ? c + 1 : NC - 1, xbuf + ((c + 1) & 1) * 4 * XPL);
            }
            __builtin_amdgcn_sched_barrier(0);
            const h8* wb = wring + (t & 3) * WSLOTS;
            h8 ah[2][MB], al[2][MB], bh[2][NBW], bl[2][NBW];
            auto fetch = [&](int kx, h8 (&fah)[MB], h8 (&fal)[MB], h8 (&fbh)[NBW], h8 (&fbl)[NBW]) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    fah[mb] = wb[a_base + kx * 4 * CO_WG + mb * 32];
                    fal[mb] = wb[a_base + kx * 4 * CO_WG + CO_WG + mb * 32];
                }
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    fbh[nb] = xb[b_base + (nb + ky) * TWP + kx];
                    fbl[nb] = xb[b_base + XPL + (nb + ky) * TWP + kx];
                }
            };
            fetch(0, ah[0], al[0], bh[0], bl[0]);
            if (F8) {
                // block scales (E8M0) of this lane's 32-element K block: lh = 0 -> fp8(w) * fp8(xl 2^9), lh = 1 -> fp8(wl 2^11) * fp8(x/4)
                const int sa = lh ? 116 : 127, sb = lh ? 129 : 118;
                const h8 z8 = {};
                // tap 0: main products; tap 1's fragments arrive meanwhile
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) {
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][mb], bh[0][nb], acc[mb][nb], 0, 0, 0);
                        if (mb + nb == 0) fetch(1, ah[1], al[1], bh[1], bl[1]);
                    }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * MB + 2 * NBW, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MB * NBW - 1, 0);
                // taps 0+1: corrections (one fp8 MFMA per tile), then tap 1's main products; tap 2's fragments replace tap 0's
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(nb_cat8(al[0][mb], al[1][mb]), nb_cat8(bl[0][nb], bl[1][nb]),
                                                                                      acc[mb][nb], 0, 0, 0, sa, 0, sb);
                fetch(2, ah[0], al[0], bh[0], bl[0]);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][mb], bh[1][nb], acc[mb][nb], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MB * NBW, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * MB + 2 * NBW, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MB * NBW, 0);
                // tap 2: main products and its corrections (second half of the fp8 K block empty)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) {
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][mb], bh[0][nb], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(nb_cat8(al[0][mb], z8), nb_cat8(bl[0][nb], z8),
                                                                                      acc[mb][nb], 0, 0, 0, sa, 0, sb);
                    }
            } else {
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
            }
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tw0 = 0;
            if (p.tstamps) tw0 = __builtin_amdgcn_s_memtime();
            // everything issued before sub-chunk t-1 must have landed (it is what sub-chunk t+1 reads)
            if (ky == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NWPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NWPW + NXPW) : "memory");
            if (p.tstamps) { const unsigned long long tw1 = __builtin_amdgcn_s_memtime(); t_dma += tw1 - tw0; tw0 = tw1; }
            __builtin_amdgcn_s_barrier();
            if (p.tstamps) t_bar += __builtin_amdgcn_s_memtime() - tw0;
        }
    }
#ifdef NB_PP_STAMPS
    if (p.tstamps && tid == 0) {
        unsigned long long* ts = p.tstamps + (size_t)item * 8;
        ts[6] = t_dma; ts[7] = t_bar;
    }
#else
    if (p.tstamps && tid == 0) {
        unsigned it_ = item; asm volatile("" : "+v"(it_));
        unsigned long long* ts = p.tstamps + (size_t)it_ * 8;
        ts[6] = t_dma; ts[7] = t_bar | ((__builtin_amdgcn_s_memtime() - t_loop0) << 32);
    }
#endif
    // drain the tail re-copies before the staging LDS is reused: every wave waits for ITS pieces, and the barrier makes
    // sure no other wave's late piece lands on top of epilogue data (without it the outcome depended on DMA timing)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    NB_TSTAMP(2);
    if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) p.y[0] = 0.f; return; }      // ablation: main loop only
    // ---- the NEXT tile's prologue goes out now, ahead of this tile's epilogue, where that epilogue works straight from the accumulators
    //      (hand-off output; the fused ToRGB of a 64-channel layer): its ~4 us of LDS-DMA round trip then pass under the epilogue's
    //      arithmetic and stores.  The fp32-output path stages its tile in the LDS the prologue fills: there the prologue follows it.
    //      The epilogue keeps THIS tile's coordinates (e_*). ----
    const int e_n = n, e_y0 = y0, e_x0 = x0, e_co0 = co0;
    const unsigned item_next = PERSIST ? __builtin_amdgcn_readfirstlane(item + gridDim.x) : total;
    const bool lds_free_epilogue = p.yh2 || (MW == 1 && p.tg.c && !p.y && p.c_out % 8 == 0);
    const bool early = item_next < total && lds_free_epilogue && !(p.dbg & 64);      // (dbg & 64: no prefetch)
    if (item_next < total) {
        tile_coords(item_next);
        piece_offsets();
        if constexpr (V2) piece_sources_v2();
        if (early) prologue_issue();
    }
    auto epilogue = [&]() {

    // ---- epilogue: *d, +noise, +bias, lrelu, gain, clamp -> fp32 NCHW; D[row = c_out, col = pixel] ----
    // The finished values go through LDS (the staging buffers are dead now) as an [c_out][pixel] image so that each
    // lane can store 16 bytes (4 consecutive pixels of one channel row): 4x fewer store instructions than storing
    // the accumulator registers directly, and whole 512-byte row segments per wave-instruction.
    constexpr int PIX_WG = TH * 32;
    if (p.yh2) {
        // (the gain through a vector register of its own: see gain_t below)
        float gain_h = p.gain;
        asm volatile("" : "+v"(gain_h));
        const H3HandoffArgs ha_ = nb_handoff_args(p.yh2, p.c8_next, p.c_out, p.h, p.w, p.out_f8, p.dbg, p.alpha, gain_h, p.clamp);
        if constexpr (F8 && V2) {          // (the f6 output form exists for the software-pipelined f8 / f6 kernels only: the launcher checks)
            if (p.out_f8 == 2) nb_up1_handoff_epilogue<MB, NBW, true>(ha_, acc, nzr, s_dco, s_bias, s_nst, wm * 64, wn * NBW, e_co0, e_n, e_y0, e_x0, lh, l31);
            else nb_up1_handoff_epilogue<MB, NBW, false>(ha_, acc, nzr, s_dco, s_bias, s_nst, wm * 64, wn * NBW, e_co0, e_n, e_y0, e_x0, lh, l31);
        } else nb_up1_handoff_epilogue<MB, NBW, false>(ha_, acc, nzr, s_dco, s_bias, s_nst, wm * 64, wn * NBW, e_co0, e_n, e_y0, e_x0, lh, l31);
        NB_TSTAMP(3);
        NB_TSTAMP(4);
        return;
    }
    if constexpr (MW == 1) {
        if (p.tg.c && !p.y && p.c_out % 8 == 0) {
            // ToRGB straight from the accumulators (nobody taps the fp32 activations): the lane holds 32 of its pixel's 64
            // channels (4 lh .. + 3 of every group), lane l + 32 the other 32; partial sums in the order of nb_torgb_dot
            // (j = the four registers of a group, summed over (mb, g)), then the two lane halves meet through one
            // v_permlane32_swap per output.  No LDS image, no barrier.
            const float clampv = p.clamp >= 0.f ? p.clamp : __builtin_inff();
            // (the gain through a vector register of its own: straight from the kernarg pair (alpha, gain) the packed `* gain` below broadcast
            //  the pair's HIGH dword through op_sel:[1,0] in the loop-less instantiations -- the form tests/test_abi.py bans)
            float gain_t = p.gain;
            asm volatile("" : "+v"(gain_t));
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const float nzg = nzr[nb] * gain_t;
                f32x4 s0, s1, s2;
                s0 = 0.f; s1 = 0.f; s2 = 0.f;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = mb * 32 + 8 * g + 4 * lh;
                        if (col < p.c_out) {
                            const f32x4 d4 = *reinterpret_cast<const f32x4*>(s_dco + col) * gain_t;
                            const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + col) * gain_t + nzg;
                            const f32x4 a4 = {acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1], acc[mb][nb][4 * g + 2], acc[mb][nb][4 * g + 3]};
                            f32x4 t = __builtin_elementwise_fma(a4, d4, b4);
                            const f32x4 ta = t * p.alpha;
#pragma unroll
                            for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(t[i], ta[i], __builtin_inff()), -clampv, clampv);
                            s0 = __builtin_elementwise_fma(t, *reinterpret_cast<const f32x4*>(s_tw + col), s0);
                            s1 = __builtin_elementwise_fma(t, *reinterpret_cast<const f32x4*>(s_tw + p.c_out + col), s1);
                            s2 = __builtin_elementwise_fma(t, *reinterpret_cast<const f32x4*>(s_tw + 2 * p.c_out + col), s2);
                        }
                    }
                float pk[3] = {(s0[0] + s0[1]) + (s0[2] + s0[3]), (s1[0] + s1[1]) + (s1[2] + s1[3]), (s2[0] + s2[1]) + (s2[2] + s2[3])};
                float a[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // swap(x, 0): the lower lanes get (their own, the upper lanes') partial sums
                    unsigned u0 = __builtin_bit_cast(unsigned, pk[k]), u1 = 0;
                    nb_swap32(u0, u1);
                    a[k] = __builtin_bit_cast(float, u0) + __builtin_bit_cast(float, u1);
                }
                if (lh == 0) nb_torgb_pixel(p.tg, e_n, (e_y0 + wn * NBW + nb) * W + e_x0 + l31, a[0], a[1], a[2], s_tcol, s_tcol01);
            }
            NB_TSTAMP(3);
            NB_TSTAMP(4);
            return;
        }
    }
    float* ot = reinterpret_cast<float*>(smem_h3);               // [CO_WG][PIX_WG] floats (<= 128 KiB)
    // (the gain through a vector register of its own: read from the kernarg pair (alpha, gain) the SLP vectoriser pairs the two `* gain` below
    //  into a v_pk_mul_f32 with the gain broadcast through op_sel:[1,0] -- a swizzled packed form, see NB_NO_PACKED_F32 in nb_common.h)
    float gain_v = p.gain;
    asm volatile("" : "+v"(gain_v));
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int trow = wn * NBW + nb;
        const float nzg = nzr[nb] * gain_v, clampv = p.clamp >= 0.f ? p.clamp : __builtin_inff();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;      // c_out within the workgroup
                // (channels past c_out carry dco = bias = 0 and are never stored)
                ot[col * PIX_WG + trow * 32 + l31] = nb_h3_act(acc[mb][nb][r], s_dco[col] * gain_v, s_bias[col] * gain_v + nzg, p.alpha, clampv);
            }
        }
    }
    __syncthreads();
    NB_TSTAMP(3);
    if (p.tg.c) {
        // ToRGB on the tile while it sits in LDS: 3 dot products over the channels per pixel, then the triad tail
        for (int pix = tid; pix < PIX_WG; pix += 512) {
            float a0, a1, a2;
            nb_torgb_dot(p.c_out, s_tw, [&](int ch) { return ot[ch * PIX_WG + pix]; }, a0, a1, a2);
            nb_torgb_pixel(p.tg, e_n, (e_y0 + (pix >> 5)) * W + e_x0 + (pix & 31), a0, a1, a2, s_tcol, s_tcol01);
        }
        if (!p.y) return;
    }
    if (!(p.dbg & 1)) {
        constexpr int V4_PER_ROW = PIX_WG / 4;                   // float4 per c_out row
        for (int e = tid; e < CO_WG * V4_PER_ROW; e += 512) {
            const int col = e / V4_PER_ROW, q4 = e - col * V4_PER_ROW;
            const int co = e_co0 + col;
            if (co < p.c_out) {
                const int trow = q4 >> 3, px = (q4 & 7) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(ot + col * PIX_WG + q4 * 4);
                *reinterpret_cast<f32x4*>(p.y + ((size_t)e_n * p.c_out + co) * ((size_t)H * W) + (size_t)(e_y0 + trow) * W + e_x0 + px) = v;
            }
        }
    }
    NB_TSTAMP(4);
    };
    epilogue();
    if (item_next >= total) break;
    item = item_next;
    // every wave is through with the tile's tables (and, on the fp32-output path, with the staged tile)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    NB_TSTAMP(0);
    if (!early) prologue_issue();
  }
#undef NB_TSTAMP
#define NB_TSTAMP(k)                                                                                         \
    do {                                                                                                     \
        if (p.tstamps && threadIdx.x == 0) p.tstamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
}

// ------------------------------------------------------------------------------------------------
// Small-workgroup form of the same layer: 4 waves, 64 c_out x 256 px, <= 75 KB of LDS, so TWO workgroups share a
// CU and one's prologue / epilogue (DMA latency, VALU, stores) runs under the other's MFMAs.  K loop = (16-channel
// chunk, tap row) steps, double buffered: the rows a step needs are gathered by LDS-DMA (zero page outside the
// image), the next step's DMA is in flight under the current step's MFMAs (the scheme of nb_encoder.hip).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void modconv3x3_up1_h3s_kernel(const H3Params p) {
    if (p.out_f8) nb_set_fp16_ovfl();             // f8 hand-off: the fp8 (and f16) conversions saturate
    constexpr int NW = 4, MB = 2, NBW = 2, CO_WG = 64, TH = NW * NBW, PW = 34, PIX_WG = TH * 32;
    constexpr int SLOTS = TH * PW, PP = (SLOTS + 63) / 64, XPL = PP * 64, NXP = 4 * PP, NXPW = (NXP + NW - 1) / NW;
    constexpr int WSLOTS = 12 * CO_WG, NWP = WSLOTS / 64, NWPW = NWP / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h3[];
    h8* xbuf = reinterpret_cast<h8*>(smem_h3);                    // [2][4][XPL]
    h8* wbuf = xbuf + 2 * 4 * XPL;                                // [2][WSLOTS]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    const int H = p.h, W = p.w;
    int b = blockIdx.x;
    if (gridDim.x % 8 == 0 && p.slices > 1) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);       // slices of a tile share an XCD
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x; const int tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int y0 = tile_y * TH, x0 = tile_x * 32, co0 = slice * CO_WG;
    const size_t HW8 = (size_t)H * W * 8;
    const _Float16* xn = p.x + (size_t)n * p.c8 * 2 * HW8;

    __shared__ __attribute__((aligned(16))) float s_dco[CO_WG], s_bias[CO_WG], s_nst[CO_WG];
    __shared__ float s_tw[3 * 128], s_tcol[9], s_tcol01[9];
    if (tid < CO_WG) {
        const int co = co0 + tid;
        s_dco[tid] = co < p.c_out ? p.dcoefs[(size_t)n * p.c_out + co] : 0.f;
        s_bias[tid] = co < p.c_out ? p.bias[co] : 0.f;
        s_nst[tid] = (p.yh2 && co < p.c_out) ? p.next_styles[(size_t)n * p.next_stride + co] : 0.f;
    }
    if (p.tg.c) nb_torgb_setup(p.tg, n, s_tw, s_tcol, s_tcol01, tid, 256, blockIdx.x == 0);
    float nzr[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
        nzr[nb] = p.noise ? p.noise[(size_t)n * p.noise_stride_n + (size_t)(y0 + wn * NBW + nb) * W + x0 + l31] : 0.f;
    if (p.nsrc.const_t) {
        // the lane's pixels (row y0 + wn NBW + nb, column x0 + l31): column parameters once, row parameters per row
        float np0, np1, wy0, wy1;
        int sy0;
        nb_noise_np(p.nsrc, n, np0, np1);
        nb_noise_axis(p.nsrc, x0 + l31, np1, sy0, wy0, wy1);
        const float strength = p.nsrc.strength[0];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            float wx0, wx1;
            int sx0;
            nb_noise_axis(p.nsrc, y0 + wn * NBW + nb, np0, sx0, wx0, wx1);
            nzr[nb] = nb_noise_value(p.nsrc, strength, sx0, wx0, wx1, sy0, wy0, wy1);
        }
    }

    int xcol[NXPW], xrow[NXPW], xpl[NXPW], xdst[NXPW];
#pragma unroll
    for (int i = 0; i < NXPW; ++i) {
        const int q = i * NW + wn;                    // NXP = NW * NXPW exactly
        const int pl4 = q / PP, part = q - pl4 * PP;
        const int e = part * 64 + lane;
        const int r = e / PW, c = e - r * PW;
        const int gx = x0 - 1 + c;
        xpl[i] = pl4;
        xdst[i] = pl4 * XPL + part * 64;
        xrow[i] = y0 - 1 + r;
        xcol[i] = (e < SLOTS && gx >= 0 && gx < W) ? gx * 8 : -1;
    }
    auto issue = [&](int t, int buf) {                // t = chunk * 3 + ky
        const int c = t / 3, ky = t - 3 * c;
        h8* xd = xbuf + buf * 4 * XPL;
#pragma unroll
        for (int i = 0; i < NXPW; ++i) {
            const int cg = 2 * c + (xpl[i] >> 1), gy = xrow[i] + ky;
            const _Float16* src = reinterpret_cast<const _Float16*>(p.zeros);
            if (xcol[i] >= 0 && gy >= 0 && gy < H && cg < p.c8)
                src = xn + (size_t)(4 * c + xpl[i]) * HW8 + (size_t)gy * W * 8 + xcol[i];
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(xd + xdst[i]), 16, 0, 0);
        }
        h8* wd = wbuf + buf * WSLOTS;
#pragma unroll
        for (int i = 0; i < NWPW; ++i) {
            const int e = (i * NW + wn) * 64 + lane;
            const int row = e / CO_WG, j = e - row * CO_WG;                      // row = kx*4 + cg*2 + hl
            const _Float16* src = p.wts + (((size_t)t * 12 + row) * p.co_ld + co0 + j) * 8;
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(wd + (i * NW + wn) * 64), 16, 0, 0);
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
    const int a_base = lh * 2 * CO_WG + l31;
    const int b_base = lh * 2 * XPL + (wn * NBW) * PW + l31;
    issue(0, 0);
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(t + 1 < T ? t + 1 : T - 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const h8* xb = xbuf + (t & 1) * 4 * XPL;
        const h8* wb = wbuf + (t & 1) * WSLOTS;
        h8 ah[2][MB], al[2][MB], bh[2][NBW], bl[2][NBW];
        auto fetch = [&](int kx, h8 (&fah)[MB], h8 (&fal)[MB], h8 (&fbh)[NBW], h8 (&fbl)[NBW]) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                fah[mb] = wb[a_base + kx * 4 * CO_WG + mb * 32];
                fal[mb] = wb[a_base + kx * 4 * CO_WG + CO_WG + mb * 32];
            }
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                fbh[nb] = xb[b_base + nb * PW + kx];
                fbl[nb] = xb[b_base + XPL + nb * PW + kx];
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
    __syncthreads();                                  // staging buffers are dead from here on
    if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) p.y[0] = 0.f; return; }      // ablation: main loop only

    if (p.yh2) {
        nb_up1_handoff_epilogue<MB, NBW>(nb_handoff_args(p.yh2, p.c8_next, p.c_out, p.h, p.w, p.out_f8, p.dbg, p.alpha, p.gain, p.clamp), acc, nzr, s_dco, s_bias, s_nst, 0, wn * NBW, co0, n, y0, x0, lh, l31);
        return;
    }
    float* ot = reinterpret_cast<float*>(smem_h3);               // [CO_WG][PIX_WG]
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int trow = wn * NBW + nb;
        const float nzg = nzr[nb] * p.gain, clampv = p.clamp >= 0.f ? p.clamp : __builtin_inff();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ot[col * PIX_WG + trow * 32 + l31] = nb_h3_act(acc[mb][nb][r], s_dco[col] * p.gain, s_bias[col] * p.gain + nzg, p.alpha, clampv);
            }
    }
    __syncthreads();
    if (p.tg.c) {
        for (int pix = tid; pix < PIX_WG; pix += 256) {
            float a0, a1, a2;
            nb_torgb_dot(p.c_out, s_tw, [&](int ch) { return ot[ch * PIX_WG + pix]; }, a0, a1, a2);
            nb_torgb_pixel(p.tg, n, (y0 + (pix >> 5)) * W + x0 + (pix & 31), a0, a1, a2, s_tcol, s_tcol01);
        }
        if (!p.y) return;
    }
    constexpr int V4_PER_ROW = PIX_WG / 4;
    for (int e = tid; e < CO_WG * V4_PER_ROW; e += 256) {
        const int col = e / V4_PER_ROW, q4 = e - col * V4_PER_ROW;
        const int co = co0 + col;
        if (co < p.c_out) {
            const int trow = q4 >> 3, px = (q4 & 7) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ot + col * PIX_WG + q4 * 4);
            *reinterpret_cast<f32x4*>(p.y + ((size_t)n * p.c_out + co) * ((size_t)H * W) + (size_t)(y0 + trow) * W + x0 + px) = v;
        }
    }
}

#define NB_PERSIST_WGS_PER_CU 4
// Workgroups per CU of the persistent launches (both large kernels).  1 would keep every workgroup resident from the start; with 4 a
// workgroup still walks 2-4 tiles of the BASELINE launches (the prefetch pays from the second tile on) but CUs come free four times per
// launch, which is what lets another stream's kernels in: same-box, 3 streams in flight, 18 415 (1) / 18 490 (2) / 18 577 (4) patches/s
// against 18 311 with one workgroup per tile; on ONE stream all three give +2.8 % (profiles/r06_ab_persistent_grid.txt).
int g_persist_wgs_per_cu = NB_PERSIST_WGS_PER_CU;
// developer / test hook: <= 0 restores the default
extern "C" void nb_debug_set_persistent_wgs_per_cu(int k) { g_persist_wgs_per_cu = k > 0 ? k : NB_PERSIST_WGS_PER_CU; }
static int g_up1_persist = -1;
// developer / test hook: 1 = persistent workgroups of the 8-wave up=1 kernel (next tile's prologue ahead of the epilogue), -1 (default) / 0 = one workgroup per tile
extern "C" void nb_debug_set_up1_persistent(int mode) { g_up1_persist = mode; }

static int launch_h3s(H3Params p, int n, hipStream_t st) {
    constexpr size_t lds_stage = (size_t)2 * (4 * 320 + 12 * 64) * 16, lds_h2 = (size_t)2 * 256 * 72 * 2;
    const size_t lds = p.yh2 ? lds_h2 : lds_stage;
    p.tiles_x = p.w / 32; p.tiles_y = p.h / 8; p.slices = (p.c_out + 63) / 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_up1_h3s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h2);
        attr_set = true;
    }
    hipLaunchKernelGGL(modconv3x3_up1_h3s_kernel, dim3(p.tiles_x * p.tiles_y * p.slices, n), dim3(256), lds, st, p);
    NB_CHECK_LAUNCH("modconv3x3_up1_h3s");
    return NB_OK;
}

template <int MW, bool F8 = false, int NBW = 2, bool V2 = false, bool F6 = false, bool PP = false, bool HO = false, bool PERSIST = true>
static int launch_h3(H3Params p, int n, hipStream_t st) {
    constexpr int NWN = 8 / MW, TH = NWN * NBW, CO_WG = MW * 64;
    constexpr int SLOTS = (TH + 2) * 34, XPL = ((SLOTS + 63) / 64) * 64;
    constexpr size_t lds_stage = (size_t)(2 * 4 * XPL + 4 * 12 * CO_WG) * 16, lds_h2 = (size_t)2 * TH * 32 * (CO_WG + 8) * 2;
    const size_t lds = lds_stage > lds_h2 ? lds_stage : lds_h2;
    p.tiles_x = p.w / 32; p.tiles_y = p.h / TH; p.slices = (p.c_out + CO_WG - 1) / CO_WG;
    p.items_x = p.tiles_x * p.tiles_y * p.slices; p.items = p.items_x * n;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_up1_h3_kernel<MW, F8, NBW, V2, F6, PP, HO, PERSIST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    // persistent workgroups: one per CU (fewer items than CUs: one each), the next tile's prologue issued ahead of the current tile's
    // epilogue; g_up1_persist == 0 (test hook): one workgroup per item, as until round 5
    static int ncu = 0;
    if (!ncu) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        ncu = v;
    }
    const long want = (long)ncu * g_persist_wgs_per_cu;                 // (workgroups per CU: see NB_PERSIST_WGS_PER_CU)
    dim3 grid(PERSIST && p.items > want ? (unsigned)want : (unsigned)p.items);
    hipLaunchKernelGGL((modconv3x3_up1_h3_kernel<MW, F8, NBW, V2, F6, PP, HO, PERSIST>), grid, dim3(512), lds, st, p);
    NB_CHECK_LAUNCH("modconv3x3_up1_h3");
    return NB_OK;
}

extern "C" const float* nb_zero_page_ptr(void);

static int g_force_up1_v2 = -1;
// developer / test hook: -1 = automatic (on), 0 / 1 = the round-3 / the software-pipelined K loop of the f8 up=1 kernel
extern "C" void nb_debug_set_up1_v2(int mode) { g_force_up1_v2 = mode; }
#ifndef NB_UP1_PP_DEFAULT
#define NB_UP1_PP_DEFAULT 1          // round 5: +1.0 % / +0.5 % per step on a slow / fast box (three alternating pairs each), bit-identical
#endif
static int g_force_up1_pp = -1;
// developer / test hook: -1 = automatic, 0 / 1 = the software-pipelined / the ping-pong K loop of the f8 up=1 kernel
extern "C" void nb_debug_set_up1_pp(int mode) { g_force_up1_pp = mode; }
static int g_force_nbw = 0;
// developer / test hook: 0 = automatic, 1 / 2 = force that many 32-pixel rows per wave in the 8-wave up=1 kernel
extern "C" void nb_debug_set_up1_rows(int nbw) { g_force_nbw = nbw; }
static int g_force_up1_small = -1;
// developer / test hook: -1 = automatic (images <= 32 x 32), 0 / 1 = never / always the 4-wave two-workgroups-per-CU form of the H2 up=1 kernel
extern "C" void nb_debug_set_up1_small(int mode) { g_force_up1_small = mode; }

static int nb_up1_h3_impl(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                          int64_t noise_stride_n, const float* bias, float* y, void* y_h2, const float* next_styles,
                          int next_stride, int c_next, int n, int h, int w, int c_out, float alpha, float gain, float clamp,
                          void* stream, const TorgbParams* tg = nullptr, int in_fmt = 0, int out_fmt = 0) {
    const bool f8 = in_fmt != 0, f6 = in_fmt == 2;          // (the f6 form is a variant of the f8 loop: same containers, same staging)
    const bool hi_only = in_fmt == 3;                       // f8 containers, correction products skipped (conv_mode "f16": where the ping-pong loop runs)
    NB_REQUIRE(out_fmt == 0 || ((out_fmt == 1 || out_fmt == 2) && y_h2 && c_out % 16 == 0 && c_next % 16 == 0),
               "modconv3x3_up1_h3: f8 / f6 output needs an H2 destination and c_out, c_next %% 16 == 0");
    NB_REQUIRE(x_h2 && w_h3 && dcoefs && bias && (tg ? !y_h2 : ((y != nullptr) != (y_h2 != nullptr))), "modconv3x3_up1_h3: null pointer");
    NB_REQUIRE(!f8 || c_in % 16 == 0, "modconv3x3_up1_h3: the f8 operand format needs c_in %% 16 == 0 (got %d)", c_in);
    NB_REQUIRE(!y_h2 || (next_styles && c_out % 8 == 0 && c_next >= c_out && next_stride >= c_out && (uintptr_t)y_h2 % 16 == 0),
               "modconv3x3_up1_h3: H2 output needs the consumer's styles, c_out %% 8 == 0 and c_next >= c_out");
    NB_REQUIRE(n > 0 && n <= 65535 && c_in > 0 && c_out > 0, "modconv3x3_up1_h3: bad sizes");
    NB_REQUIRE(alpha >= 0.f && alpha <= 1.f && gain > 0.f, "modconv3x3_up1_h3: leaky-ReLU slope must lie in [0, 1] and the gain be positive (got %g, %g)", alpha, gain);
    NB_REQUIRE(w % 32 == 0 && h % 16 == 0, "modconv3x3_up1_h3: needs w %% 32 == 0 and h %% 16 == 0 (got %dx%d)", h, w);
    NB_REQUIRE(((uintptr_t)x_h2 | (uintptr_t)w_h3 | (uintptr_t)y) % 16 == 0, "modconv3x3_up1_h3: pointers must be 16-byte aligned");
    H3Params p;
    p.x = (const _Float16*)x_h2; p.wts = (const _Float16*)w_h3; p.dcoefs = dcoefs; p.noise = noise; p.bias = bias; p.y = y;
    p.zeros = nb_zero_page_ptr();
    NB_REQUIRE(p.zeros, "modconv3x3_up1_h3: could not allocate the zero page");
    p.noise_stride_n = noise_stride_n;
    NB_REQUIRE(nb_noise_src_setup(noise, noise_stride_n, h, w, &p.noise, &p.noise_stride_n, &p.nsrc) == NB_OK, "modconv3x3_up1_h3: bad NbNoiseSrc (needs the "
               "transposed constant, the grid row, the strength, exactly one of norm_pos / positions, and res = the %dx%d output)", h, w);
    p.c8 = (c_in + 7) / 8; p.nchunks = (c_in + 15) / 16; p.c_out = c_out; p.co_ld = (c_out + 63) / 64 * 64; p.h = h; p.w = w;
    p.dbg = g_nb_debug_flags; p.stagger_ticks = g_nb_stagger_ticks;            // (developer hooks: nb_debug_set_flags / nb_debug_set_stagger)
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.yh2 = (_Float16*)y_h2; p.next_styles = next_styles; p.next_stride = next_stride; p.c8_next = (c_next + 7) / 8;
    p.out_f8 = out_fmt;
    p.tg = TorgbParams{};
    if (tg) p.tg = *tg;
    p.tstamps = nullptr;
    if (g_tstamps) {
        const long long wgs = (long long)(w / 32) * (h / (c_out > 64 ? 8 : 16)) * ((c_out + (c_out > 64 ? 127 : 63)) / (c_out > 64 ? 128 : 64)) * n;
        if (wgs <= g_tstamps_cap) p.tstamps = g_tstamps;
    }
    // the 4-wave / 2-workgroups-per-CU form wins on small images (fewer, larger workgroups leave CUs idle there); on the
    // large layers both forms run at the same rate -- the chip is power-limited in these loops, not latency-limited
    { const bool small = g_force_up1_small >= 0 ? g_force_up1_small != 0 : (h * w <= 32 * 32);
      if (small && !f8 && (!tg || c_out <= 64)) return launch_h3s(p, n, (hipStream_t)stream); }
    // half-height tiles when the full ones leave the chip mostly idle (batch 1); g_force_nbw: test hook
    const long wgs_full = (long)n * (w / 32) * (h / (c_out > 64 ? 8 : 16)) * ((c_out + (c_out > 64 ? 127 : 63)) / (c_out > 64 ? 128 : 64));
    const bool half = g_force_nbw ? g_force_nbw == 1 : wgs_full < 160;
    hipStream_t st = (hipStream_t)stream;
    // the software-pipelined K loop (V2; both operand formats) unless switched off (test hook nb_debug_set_up1_v2)
    // (its pieces walk the chunks with a fixed per-chunk stride: whole 16-channel chunks only -- f8 operands always are; H2 operands
    //  with an odd number of channel groups keep the round-3 loop, whose last chunk reads the missing group from the zero page)
    const bool v2 = g_force_up1_v2 != 0 && (f8 || p.c8 % 2 == 0);
    NB_REQUIRE(out_fmt != 2 || (f8 && v2), "modconv3x3_up1_h3: the f6 output format is written by the software-pipelined f8 / f6 kernels only");
    // Persistent workgroups: automatic = NO (see the kernel's PERSIST); nb_debug_set_up1_persistent(1) = always -- the tests and
    // tools/stress_persistent.py keep that instantiation honest.  Half-height launches (< 160 workgroups) have nothing to walk: loop-less always.
    const bool persist = g_up1_persist > 0;
#define NB_H3_GO(MW, F8, NBW, V2, F6, PP, HO) \
    (persist ? launch_h3<MW, F8, NBW, V2, F6, PP, HO, true>(p, n, st) : launch_h3<MW, F8, NBW, V2, F6, PP, HO, false>(p, n, st))
#define NB_H3_GO1(MW, F8, NBW, V2, F6) launch_h3<MW, F8, NBW, V2, F6, false, false, false>(p, n, st)
    if (f6) {                                               // (the software-pipelined loop only)
        if (half) return c_out > 64 ? NB_H3_GO1(2, true, 1, true, true) : NB_H3_GO1(1, true, 1, true, true);
        return c_out > 64 ? NB_H3_GO(2, true, 2, true, true, false, false) : NB_H3_GO(1, true, 2, true, true, false, false);
    }
    if (half) {
        if (f8 && v2) return c_out > 64 ? NB_H3_GO1(2, true, 1, true, false) : NB_H3_GO1(1, true, 1, true, false);
        if (f8) return c_out > 64 ? NB_H3_GO1(2, true, 1, false, false) : NB_H3_GO1(1, true, 1, false, false);
        if (v2) return c_out > 64 ? NB_H3_GO1(2, false, 1, true, false) : NB_H3_GO1(1, false, 1, true, false);
        return c_out > 64 ? NB_H3_GO1(2, false, 1, false, false) : NB_H3_GO1(1, false, 1, false, false);
    }
    // the ping-pong form of that loop (full-height tiles): nb_debug_set_up1_pp
    const bool pp = (g_force_up1_pp >= 0 ? g_force_up1_pp : NB_UP1_PP_DEFAULT) != 0;
    if (f8 && v2 && pp && hi_only) return c_out > 64 ? NB_H3_GO(2, true, 2, true, false, true, true) : NB_H3_GO(1, true, 2, true, false, true, true);
    if (f8 && v2 && pp) return c_out > 64 ? NB_H3_GO(2, true, 2, true, false, true, false) : NB_H3_GO(1, true, 2, true, false, true, false);
    if (f8 && v2) return c_out > 64 ? NB_H3_GO(2, true, 2, true, false, false, false) : NB_H3_GO(1, true, 2, true, false, false, false);
    if (f8) return c_out > 64 ? NB_H3_GO(2, true, 2, false, false, false, false) : NB_H3_GO(1, true, 2, false, false, false, false);
    if (v2) return c_out > 64 ? NB_H3_GO(2, false, 2, true, false, false, false) : NB_H3_GO(1, false, 2, true, false, false, false);
    return c_out > 64 ? NB_H3_GO(2, false, 2, false, false, false, false) : NB_H3_GO(1, false, 2, false, false, false, false);
#undef NB_H3_GO
#undef NB_H3_GO1
}

extern "C" int nb_modconv3x3_up1_h3_ex(const void* x, int c_in, const void* wts, const float* dcoefs, const float* noise,
                                      int64_t noise_stride_n, const float* bias, float* y_f32, void* y_h2,
                                      const float* next_styles, int next_stride, int c_next, const NbTorgbArgs* t, int in_fmt,
                                      int out_fmt, int n, int h, int w, int c_out, float alpha, float gain, float clamp,
                                      void* stream) {
    NB_REQUIRE(in_fmt >= 0 && in_fmt <= 3, "modconv3x3_up1_h3: operand format must be 0 (H2), 1 (f8), 2 (f6) or 3 (f8 containers, hi x hi products only)");
    if (!t)
        return nb_up1_h3_impl(x, c_in, wts, dcoefs, noise, noise_stride_n, bias, y_f32, y_h2, next_styles, next_stride, c_next,
                              n, h, w, c_out, alpha, gain, clamp, stream, nullptr, in_fmt, out_fmt);
    NB_REQUIRE(t->styles && t->w && t->bias && t->color_bias && !y_h2, "modconv3x3_up1_h3_ex: bad ToRGB arguments");
    NB_REQUIRE(c_out <= 128, "modconv3x3_up1_h3_ex: the fused ToRGB needs all channels in one workgroup (c_out <= 128)");
    NB_REQUIRE(t->styles_stride_n >= c_out + 9, "torgb_triad: styles rows must hold 9 color scalars + c styles");
    NB_REQUIRE(t->render_mode == 0 || t->render_mode == 1, "Unknown render mode for TriadGanPaintEngine: %d", t->render_mode);
    TorgbParams tg;
    tg.x = nullptr; tg.styles = t->styles; tg.w = t->w; tg.bias = t->bias; tg.color_bias = t->color_bias;
    tg.logits = t->logits; tg.uvs = t->uvs; tg.img = t->img; tg.colors_out = t->colors_out; tg.user_colors = t->user_colors;
    tg.sfactor = t->sfactor; tg.rgba_f32 = t->rgba_f32; tg.rgba_u8 = t->rgba_u8;
    tg.styles_stride_n = t->styles_stride_n; tg.c = c_out; tg.hw = h * w; tg.render_mode = t->render_mode; tg.clamp = t->clamp;
    return nb_up1_h3_impl(x, c_in, wts, dcoefs, noise, noise_stride_n, bias, y_f32, nullptr, nullptr, 0, 0, n, h, w, c_out,
                          alpha, gain, clamp, stream, &tg, in_fmt, 0);
}

extern "C" int nb_modconv3x3_up1_h3(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                                    int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w, int c_out,
                                    float alpha, float gain, float clamp, void* stream) {
    return nb_up1_h3_impl(x_h2, c_in, w_h3, dcoefs, noise, noise_stride_n, bias, y, nullptr, nullptr, 0, 0, n, h, w, c_out,
                          alpha, gain, clamp, stream);
}

extern "C" int nb_modconv3x3_up1_h3_torgb(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                                          int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w, int c_out,
                                          float alpha, float gain, float clamp, const NbTorgbArgs* t, void* stream) {
    NB_REQUIRE(t && t->styles && t->w && t->bias && t->color_bias, "modconv3x3_up1_h3_torgb: null pointer");
    NB_REQUIRE(c_out <= 128, "modconv3x3_up1_h3_torgb: the fused ToRGB needs all channels in one workgroup (c_out <= 128)");
    NB_REQUIRE(t->styles_stride_n >= c_out + 9, "torgb_triad: styles rows must hold 9 color scalars + c styles");
    NB_REQUIRE(t->render_mode == 0 || t->render_mode == 1, "Unknown render mode for TriadGanPaintEngine: %d", t->render_mode);
    TorgbParams tg;
    tg.x = nullptr; tg.styles = t->styles; tg.w = t->w; tg.bias = t->bias; tg.color_bias = t->color_bias;
    tg.logits = t->logits; tg.uvs = t->uvs; tg.img = t->img; tg.colors_out = t->colors_out; tg.user_colors = t->user_colors;
    tg.sfactor = t->sfactor; tg.rgba_f32 = t->rgba_f32; tg.rgba_u8 = t->rgba_u8;
    tg.styles_stride_n = t->styles_stride_n; tg.c = c_out; tg.hw = h * w; tg.render_mode = t->render_mode; tg.clamp = t->clamp;
    return nb_up1_h3_impl(x_h2, c_in, w_h3, dcoefs, noise, noise_stride_n, bias, y, nullptr, nullptr, 0, 0, n, h, w, c_out,
                          alpha, gain, clamp, stream, &tg);
}

extern "C" int nb_modconv3x3_up1_h3_h2(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                                       int64_t noise_stride_n, const float* bias, const float* next_styles, int next_stride,
                                       void* y_h2, int c_next, int n, int h, int w, int c_out, float alpha, float gain,
                                       float clamp, void* stream) {
    return nb_up1_h3_impl(x_h2, c_in, w_h3, dcoefs, noise, noise_stride_n, bias, nullptr, y_h2, next_styles, next_stride,
                          c_next, n, h, w, c_out, alpha, gain, clamp, stream);
}

// ------------------------------------------------------------------------------------------------
// up = 2 on the f16 matrix cores: the 4-phase transposed convolution + fused polyphase FIR of
// nb_modconv.hip (same math, same quad grid: see the comment block there), with split-f16 products.
//
// Workgroup = 8 waves, one 12 x 32 quad tile (14 x 34 = 476 halo'd positions = 15 MFMA column blocks of 32, two
// per wave) x 32 c_out.  All four phases share the staged halo tile (9 taps per staged byte), so every wave
// carries 2 blocks x 4 phases x 16 = 128 accumulator registers; the A/B fragments are streamed per tap.
// (12 rows do not divide the image height: the last tile row overhangs and is masked.)  Input H2 (pre-modulated), weights [chunk][tap 9][cg 2][hi/lo 2][c_out_ld][8], output fp32 NCHW.
// ------------------------------------------------------------------------------------------------
#ifndef NB_H3_TQH
#define NB_H3_TQH 12
#endif
#define NB_H3_TQH_SMALL 5       // tile height of the under-filled (batch-1) launches
#define NB_H3_TQH_MID 8         // 10 x 34 = 340 positions = 11 blocks (3 / 3 / 3 / 2 per SIMD): for launches whose 12-row tiles end in a mostly empty round of workgroups
#define NB_H3_STAGES 3          // LDS-DMA stages of the up=2 kernel (planes at their exact size: 3 x 52 032 B for the 12-row tile)

// TQH = quad rows per tile: NB_H3_TQH (12) for throughput; 5 (7 x 34 = 238 positions = 8 column blocks, one per wave)
// when the large tiles would leave most of the chip idle - the batch-1 / interactive configuration.
// OUTM = output mode: 0 = fp32 NCHW, 1 = the consumer's H2 tensor, 2 = the consumer's tensor in the "f8" operand format.
// TQW_ = quad columns per tile: 32, or 16 for 16-wide inputs (b32.conv0 at large batch: 8 x 16 tiles = 10 x 18 = 180
// positions = 6 blocks, one per wave).
// (Tried and dropped, DESIGN.md 6: a one-wave-per-SIMD form with 256 accumulators per wave and a hand-ordered K loop, and two
//  4-wave workgroups per CU on 5-row tiles -- neither was faster.)
// NW_ / NST_ = waves per workgroup / LDS-DMA stages: 8 / 3 (one workgroup per CU), or 4 / 2 on 12 x 16 tiles (14 x 18 = 252
// positions = 8 column blocks, two per wave; 2 x 36.7 KB of staging) so that TWO workgroups share a CU and one's prologue
// (first-chunk DMA latency) and epilogue (LDS round trip, FIR, conversions, stores: a quarter to a third of a workgroup's
// life, no matrix work) run under the other's K loop.
template <bool F8, int TQH, int OUTM = 0, int TQW_ = 32, int NW_ = 8, int NST_ = NB_H3_STAGES>
__global__ __launch_bounds__(NW_ * 64, 2) void modconv3x3_up2_h3_kernel(const H3Up2Params p) {
    NB_TSTAMP(0);
    if constexpr (OUTM == 2) nb_set_fp16_ovfl();
    nb_stagger(p.stagger_ticks, 256);
    constexpr int NW = NW_, NT = NW_ * 64, TQW = TQW_, PH = TQH + 2, PW = TQW + 2, NPOS = PH * PW;     // 476
    constexpr int NBLK = (NPOS + 31) / 32;            // 15 position blocks
    constexpr int NBJ = (NBLK + NW - 1) / NW;         // blocks per wave (2)
    constexpr int XR = TQH + 3, XS = TQW + 3;         // halo tile 15 x 35 input pixels
    // planes are packed at their exact size (the last 1-KiB DMA piece of a plane is partial: lanes past the plane do not
    // copy), which is what lets THREE stages of the 12-row tile fit the 160 KiB of LDS
    constexpr int SLOTS = XR * XS, PP = (SLOTS + 63) / 64, XPL = SLOTS;         // 525 slots -> 9 pieces
    constexpr int NXP = 4 * PP, NXPW = (NXP + NW - 1) / NW;                     // 36 -> 5 per wave
    constexpr int WSLOTS = 36 * 32, NWP = WSLOTS / 64, NWPW = (NWP + NW - 1) / NW;   // 18 -> 3 per wave
    // LDS-DMA pieces a wave issues per chunk: the 36 activation + 18 weight pieces are ONE list dealt round-robin to the waves
    // (7 each, 2 re-copies; dealt per kind it was 5 + 3 = 8 each with 10 re-copies -- and a piece is a KiB through the CU's
    // vector-memory path whether anybody needs it or not)
    constexpr int NPC = (NXP + NWP + NW - 1) / NW;
    constexpr int STAGE = 4 * XPL + WSLOTS;           // 16-byte slots per stage
    constexpr int NST = NST_;                         // 3 (2: the two-workgroups-per-CU form)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h3[];
    h8* ring = reinterpret_cast<h8*>(smem_h3);        // [NST][ x: 4 planes x XPL | w: 36 rows x 32 ]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    const int H = p.h, W = p.w;
    int b = blockIdx.x;
    // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so the c_out
    // slices of one input tile -- which re-read the same activations -- are renumbered to share an XCD
    // (the XCD's share of the tile list rotates with the sample: the cheap tiles of a ragged last tile row -- see the K loop's
    //  copies below -- would otherwise all sit on the last XCDs, which then idle while the others finish)
    if (gridDim.x % 8 == 0 && !(p.dbg & 8)) b = (((b & 7) + blockIdx.y) & 7) * (gridDim.x >> 3) + (b >> 3);
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x; const int tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int I0 = tile_y * TQH, J0 = tile_x * TQW;
    const int co0 = slice * 32;
    const size_t HW8 = (size_t)H * W * 8;
    const _Float16* xn = p.x + (size_t)n * p.c8 * 2 * HW8;
    const int nblk = wv < NBLK - (NBJ - 1) * NW ? NBJ : NBJ - 1;      // blocks wv, wv+8 (< 15)

    // epilogue operands fetched under the prologue DMA: demodulation / bias per channel and the tile's noise
    __shared__ __attribute__((aligned(16))) float s_dco[32], s_bias[32], s_nst[32];
    __shared__ __attribute__((aligned(16))) float s_noise[2 * TQH * 2 * TQW];
    if (tid < 32) {
        const int co = co0 + tid;
        // the activation gain is folded into the three addends: lrelu(g t) = g lrelu(t) for g > 0 (the launcher checks)
        float dg = co < p.c_out ? p.dcoefs[(size_t)n * p.c_out + co] * p.gain : 0.f;
        float bv = co < p.c_out ? p.bias[co] : 0.f;
        // (the two `* gain` must not meet in one packed instruction: the SLP vectoriser pairs them as v_pk_mul_f32 with the gain
        //  broadcast through op_sel:[1,0] in some instantiations -- a swizzled packed form, see NB_NO_PACKED_F32 in nb_common.h)
        asm volatile("" : "+v"(dg), "+v"(bv));
        s_dco[tid] = dg;
        s_bias[tid] = bv * p.gain;
        s_nst[tid] = (p.yh2 && co < p.c_out) ? p.next_styles[(size_t)n * p.next_stride + co] : 0.f;
    }
    // LDS-DMA descriptors of this wave's pieces.  Activation piece i: plane xpl (= cg_local*2 + hi/lo), 64 slots from
    // `part`; xsp = element offset of the lane's source slot inside a plane (-1: outside the image -> zero page),
    // xok = the lane's slot belongs to the plane (the last piece of a plane is partial)
    int xsp[NXPW], xpl[NXPW], xdst[NXPW];
    bool xok[NXPW];
#pragma unroll
    for (int i = 0; i < NXPW; ++i) {
        int q = i * NW + wv;
        q = q < NXP ? q : NXP - 1;
        const int pl = q / PP, part = q - pl * PP;
        const int e = part * 64 + lane;
        xpl[i] = pl;
        xdst[i] = pl * XPL + part * 64;
        xsp[i] = -1;
        xok[i] = e < SLOTS;
        if (e < SLOTS) {
            const int r = e / XS, c = e - r * XS;
            const int gy = I0 - 1 + r, gx = J0 - 1 + c;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) xsp[i] = (gy * W + gx) * 8;
        }
    }
    // piece k of chunk c into stage st: k < NXPW activations, else weights
    auto issue_piece = [&](auto kk, int c, h8* st) {
        constexpr int k = decltype(kk)::value;
        // list position u = k NW + wv: activation piece u (descriptor k of this wave) if u < NXP, else weight piece u - NXP
        constexpr bool ALLX = k * NW + NW - 1 < NXP, ALLW = k * NW >= NXP;
        const _Float16* xsrc = reinterpret_cast<const _Float16*>(p.zeros);
        h8* xd = st;
        bool xo = false;
        if constexpr (!ALLW) {
            const int cg = 2 * c + (xpl[k] >> 1);
            if (xsp[k] >= 0 && cg < p.c8) xsrc = xn + (size_t)(4 * c + xpl[k]) * HW8 + xsp[k];
            xd = st + xdst[k]; xo = xok[k];
        }
        if constexpr (ALLX) {
            if (xo) nb_lds_dma16(xsrc, xd);
        } else {
            int q = k * NW + wv - NXP;
            q = q < 0 ? 0 : (q < NWP ? q : NWP - 1);
            const int e = q * 64 + lane;
            const int row = e >> 5, j = e & 31;           // row = tap*4 + cg*2 + hl
            const _Float16* wsrc = p.wts + (((size_t)c * 36 + row) * p.co_ld + co0 + j) * 8;
            h8* wd = st + 4 * XPL + q * 64;
            if constexpr (ALLW) {
                nb_lds_dma16(wsrc, wd);
            } else {                                      // the round where the list changes kind: a wave issues ONE of the two (the
                const bool isx = k * NW + wv < NXP;       // other has no lane enabled, and such an instruction does not exist for vmcnt)
                if (isx && xo) nb_lds_dma16(xsrc, xd);
                if (!isx) nb_lds_dma16(wsrc, wd);
            }
        }
    };
    auto issue = [&](int c, h8* st) { nb_static_for<0, NPC>([&](auto k) { issue_piece(k, c, st); }); };
    // the pieces of one chunk spread over S points of the MFMA stream: point s issues pieces [s NPC / S, (s+1) NPC / S)
    auto issue_at = [&](auto ss, auto SS, int c, h8* st) {
        constexpr int s_ = decltype(ss)::value, S_ = decltype(SS)::value;
        nb_static_for<s_ * NPC / S_, (s_ + 1) * NPC / S_>([&](auto k) { issue_piece(k, c, st); });
    };

    // B-fragment base slots: position (r, c) of block (wv + 8j); X(r, c) = slot r*XS + c of plane (lh*2 + hl)
    int boff[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        int pidx = (wv + NW * j) * 32 + l31;
        pidx = pidx < NPOS ? pidx : NPOS - 1;
        const int r = pidx / PW, c = pidx - r * PW;
        boff[j] = lh * 2 * XPL + r * XS + c;
    }
    const int aoff = 4 * XPL + lh * 2 * 32 + l31;     // + tap*128 + hl*32

    f32x16 acc[NBJ][4];
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][ph][r] = 0.f;

    // ---- K loop: 16-channel chunks through a THREE-stage LDS ring.  Chunk c is read from stage c % 3 while the LDS-DMA
    //      pieces of chunk c+2 are issued one or two at a time BETWEEN the chunk's MFMA groups (a piece costs the issuing
    //      wave ~60 cycles among MFMAs, 100-185 in a burst next to the fragment reads -- and after a barrier both waves of a
    //      SIMD would burst at the same moment, with nobody feeding the matrix pipe: that burst was a third of the K loop
    //      of the 2-stage form).  One wait + barrier per chunk: vmcnt(NPC) = everything but the pieces just issued, i.e.
    //      chunk c+1 (issued a whole chunk ago) has landed; the barrier also frees stage (c-1) % 3 = (c+2) % 3 for the
    //      next chunk's pieces.  The last two chunks issue nothing (own copies of the body: no branch inside it). ----
    const int NC = p.nchunks;
    static_assert(NST == 3 || NST == 2, "ring indices below are written for three (or two) stages");
    issue(0, ring);
    // the tile's noise values (epilogue operand), computed or fetched while chunk 0 is on its way: with the in-kernel noise
    // (NbNoiseSrc) that is ~150 VALU instructions per thread, which cost 1 us of prologue when they ran ahead of the first DMA
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

    if (NC > 1 && NST == 3) {
        issue(1, ring + STAGE);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    NB_TSTAMP(1);
    // tap (a,b) row-major -> (slot offset of its input pixel, output phase): x11 = X(r,c), x10 = X(r,c+1),
    // x01 = X(r+1,c), x00 = X(r+1,c+1).  Only four distinct offsets occur, so the taps are walked grouped by offset:
    // one pair of B fragments (hi, lo) per block serves up to four taps, and the A fragments of the next tap (and the
    // B fragments of the next group) are fetched under the six MFMAs of the current tap.
    constexpr int kOrd[9] = {4, 5, 7, 8, 3, 6, 1, 2, 0};
    constexpr int kDel[9] = {0, 0, 0, 0, 1, 1, XS, XS, XS + 1};          // slot offset, in walk order
    constexpr int kGrp[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
    constexpr int kPha[9] = {3, 2, 1, 0, 2, 0, 1, 0, 0};                 // kTapPhase[kOrd[i]]
    unsigned long long t_dma = 0, t_bar = 0;
    const unsigned long long t_loop0 = p.tstamps ? __builtin_amdgcn_s_memtime() : 0;
    // one chunk: reads from `st`; DMA = issue the pieces of chunk cn into `sn` between the MFMA groups
    // (Measured, round 3 -- an ablation build without the in-loop pieces, and one that also drops the barrier: the pieces cost
    //  ~480 of a chunk's ~3 550 cycles wherever they sit in the chunk, also with the two waves of a SIMD issuing theirs at
    //  different points; the barrier costs nothing; the loop without DMA still needs ~3 050 cycles for 2 432 cycles of MFMA.
    //  A variant with masked out-of-image lanes and pre-zeroed halo slots saved registers but is WRONG: a piece whose lanes
    //  are all masked is skipped, and the counted vmcnt waits below rely on every wave issuing exactly NPC pieces per chunk.)
    auto chunk = [&](auto dma_, auto nbe_, const h8* st, int cn, h8* sn) {
        constexpr bool DMA = decltype(dma_)::value;
        constexpr int NBE = decltype(nbe_)::value & 3;      // column blocks this wave multiplies (a wave whose second block does not exist skips its MFMAs and reads)
        // ... and how many of them take all four phases.  The LAST block of a full tile lies wholly in the last halo row
        // (positions 448 .. 475 of the 14 x 34), which the FIR reads in the odd-row phases only (oe, oo: rows ti .. ti+2 of a quad;
        // ee, eo: ti, ti+1): its six even-row taps and their fragment reads are left out (nbe_ & 4).
        constexpr int NB01 = (decltype(nbe_)::value & 4) ? NBE - 1 : NBE;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (F8) {
            // "f8" operands (see modconv3x3_up1_h3_kernel): one f16 MFMA per tap for the main product, and the two
            // correction products of a PAIR of taps that feed the same output phase on one block-scaled fp8 MFMA:
            //   phase 0 (ee): taps (8,6) and (2,0)   phase 1 (eo): (7,1)   phase 2 (oe): (5,3)   phase 3 (oo): 4 alone
            // walked so that the B fragments of input offsets 0 and 1 are loaded once and those of XS / XS+1 replace them.
            const int sa = lh ? 116 : 127, sb = lh ? 129 : 118;
            const h8 z8 = {};
            h8 b0h[NBJ] = {}, b0l[NBJ] = {}, b1h[NBJ] = {}, b1l[NBJ] = {}, b2h[NBJ] = {}, b2l[NBJ] = {};
            h8 a1h = {}, a1l = {}, a2h = {}, a2l = {}, n1h = {}, n1l = {}, n2h = {}, n2l = {};
            using S4 = std::integral_constant<int, 4>;
#ifdef NB_ABL_NOREAD      // developer ablation (tools/build_variant.sh; timing only, wrong results): no fragment reads
#define NB_RD(dst, src) asm volatile("" : "+v"(dst))
#else
#define NB_RD(dst, src) dst = (src)
#endif
#define NB_LDA(tap, hi, lo) { NB_RD(hi, st[aoff + (tap) * 128]); NB_RD(lo, st[aoff + (tap) * 128 + 32]); }
#define NB_LDB(del, hi, lo) { _Pragma("unroll") for (int j = 0; j < NB01; ++j) { NB_RD(hi[j], st[boff[j] + (del)]); NB_RD(lo[j], st[boff[j] + XPL + (del)]); } }
#define NB_PAIR(ph, ah_a, al_a, bha, bla, ah_b, al_b, bhb, blb)                                                                   \
            { _Pragma("unroll") for (int j = 0; j < ((ph) >= 2 ? NBE : NB01); ++j) {                                               \
                f32x16& a_ = acc[j][ph];                                                                                           \
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_a, bha[j], a_, 0, 0, 0);                                            \
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_b, bhb[j], a_, 0, 0, 0);                                            \
                a_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(nb_cat8(al_a, al_b), nb_cat8(bla[j], blb[j]), a_, 0, 0, 0, sa, 0, sb); } }
#ifdef NB_ABL_NODMA       // developer ablation: no LDS-DMA inside the K loop (the chunks read what the prologue staged)
#define NB_DMA(slot) {}
#else
#define NB_DMA(slot) { if constexpr (DMA) { __builtin_amdgcn_sched_barrier(0); issue_at(std::integral_constant<int, slot>{}, S4{}, cn, sn); __builtin_amdgcn_sched_barrier(0); } }
#endif
            // Program order = issue order (a scheduling fence after every group): left to itself the scheduler sinks each
            // fragment read to just before its first use (register pressure), and every MFMA group then starts with an
            // exposed `s_waitcnt lgkmcnt(0)`.  With the reads of the NEXT group issued before the current group's MFMAs the
            // compiler's own wait bookkeeping emits counted waits (lgkmcnt(4), (2), (8), ...) and the reads return under
            // 256 cycles of matrix work.
#define NB_FENCE() __builtin_amdgcn_sched_barrier(0)
            // first group's operands in the order its MFMAs take them (the counted waits then release the first MFMA after
            // three reads, not after all sixteen), then the second group's A fragments
            NB_RD(a1h, st[aoff + 8 * 128]);
#pragma unroll
            for (int j = 0; j < NBE; ++j) NB_RD(b0h[j], st[boff[j]]);
            NB_RD(a2h, st[aoff + 6 * 128]);
#pragma unroll
            for (int j = 0; j < NBE; ++j) NB_RD(b1h[j], st[boff[j] + 1]);
            NB_RD(a1l, st[aoff + 8 * 128 + 32]); NB_RD(a2l, st[aoff + 6 * 128 + 32]);
#pragma unroll
            for (int j = 0; j < NBE; ++j) { NB_RD(b0l[j], st[boff[j] + XPL]); NB_RD(b1l[j], st[boff[j] + XPL + 1]); }
            NB_LDA(5, n1h, n1l); NB_LDA(3, n2h, n2l);                             // (for the second group)
            NB_FENCE();
            NB_PAIR(0, a1h, a1l, b0h, b0l, a2h, a2l, b1h, b1l);                   // taps 8, 6
            NB_FENCE();
            NB_DMA(0);
            NB_LDA(4, a1h, a1l);                                                  // (for the third group)
            NB_FENCE();
            NB_PAIR(2, n1h, n1l, b0h, b0l, n2h, n2l, b1h, b1l);                   // taps 5, 3
            NB_FENCE();
            NB_DMA(1);
            NB_LDB(XS, b1h, b1l); NB_LDA(7, n1h, n1l); NB_LDA(1, n2h, n2l);       // (for the fourth group)
            NB_FENCE();
#pragma unroll
            for (int j = 0; j < NBE; ++j) {                                       // tap 4 alone
                f32x16& a_ = acc[j][3];
                a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0h[j], a_, 0, 0, 0);
                a_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(nb_cat8(a1l, z8), nb_cat8(b0l[j], z8), a_, 0, 0, 0, sa, 0, sb);
            }
            NB_FENCE();
            NB_DMA(2);
            NB_LDB(XS + 1, b2h, b2l); NB_LDA(2, a1h, a1l); NB_LDA(0, a2h, a2l);   // (for the fifth group)
            NB_FENCE();
            NB_PAIR(1, n1h, n1l, b0h, b0l, n2h, n2l, b1h, b1l);                   // taps 7, 1
            NB_FENCE();
            NB_DMA(3);
            NB_PAIR(0, a1h, a1l, b1h, b1l, a2h, a2l, b2h, b2l);                   // taps 2, 0
#undef NB_FENCE
#undef NB_LDA
#undef NB_LDB
#undef NB_PAIR
#undef NB_DMA
        } else {
            h8 ah[2], al[2], bh[2][NBJ], bl[2][NBJ];
            ah[0] = st[aoff + kOrd[0] * 128]; al[0] = st[aoff + kOrd[0] * 128 + 32];
#pragma unroll
            for (int j = 0; j < NBE; ++j) { bh[0][j] = st[boff[j] + kDel[0]]; bl[0][j] = st[boff[j] + XPL + kDel[0]]; }
            nb_static_for<0, 9>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                constexpr int ca = i & 1, cb = kGrp[i] & 1;
                constexpr int NBT = kPha[i] >= 2 ? NBE : NB01;            // blocks that take this tap
                constexpr int NBF = kGrp[i + 1 < 9 ? i + 1 : i] >= 2 ? NB01 : NBE;      // blocks that need the next offset group's B fragments (groups 2, 3 = taps 1, 2, 0: even-row phases only)
                f32x16& a0 = acc[0][kPha[i]];
                if constexpr (NBT > 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bh[cb][0], a0, 0, 0, 0);
                constexpr int nfetch = i + 1 < 9 ? (kGrp[i + 1 < 9 ? i + 1 : i] != kGrp[i] ? 2 + 2 * NBF : 2) : 0;
                if constexpr (i + 1 < 9) {
                    ah[ca ^ 1] = st[aoff + kOrd[i + 1] * 128]; al[ca ^ 1] = st[aoff + kOrd[i + 1] * 128 + 32];
                    if constexpr (kGrp[i + 1] != kGrp[i]) {
#pragma unroll
                        for (int j = 0; j < NBF; ++j) {
                            bh[cb ^ 1][j] = st[boff[j] + kDel[i + 1]]; bl[cb ^ 1][j] = st[boff[j] + XPL + kDel[i + 1]];
                        }
                    }
                }
                if constexpr (NBT > 0) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bl[cb][0], a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ca], bh[cb][0], a0, 0, 0, 0);
                }
#pragma unroll
                for (int j = 1; j < NBT; ++j) {
                    f32x16& aj = acc[j][kPha[i]];
                    aj = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bh[cb][j], aj, 0, 0, 0);
                    aj = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bl[cb][j], aj, 0, 0, 0);
                    aj = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ca], bh[cb][j], aj, 0, 0, 0);
                }
                // next tap's fragment reads go out BEFORE this tap's six MFMAs (their registers are free: the previous tap
                // has issued), which gives the LDS the whole tap to answer
                if constexpr (nfetch > 0) __builtin_amdgcn_sched_group_barrier(0x100, nfetch, 0);
                if constexpr (NBT > 0) __builtin_amdgcn_sched_group_barrier(0x008, 3 * NBT, 0);
                // this chunk's share of the next-but-one chunk's LDS-DMA pieces, behind the tap's MFMAs (taps 0..7)
                if constexpr (DMA && i < 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_at(ii, std::integral_constant<int, 8>{}, cn, sn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto kloop = [&](auto nbe) {
    int c = 0;
    int s_cur = 0;                                    // stage of chunk c
    if constexpr (NST == 2) {
        // two stages: chunk c+1 lands in the other stage while chunk c is multiplied; the wait at the end of a chunk is for
        // pieces issued during it -- exposed latency that the co-resident workgroup's matrix work covers
        for (; c + 1 < NC; ++c) {
            chunk(std::true_type{}, nbe, ring + s_cur * STAGE, c + 1, ring + (s_cur ^ 1) * STAGE);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            s_cur ^= 1;
        }
    }
    for (; NST == 3 && c + 2 < NC; ++c) {
        const int s_nn = s_cur == 0 ? 2 : s_cur - 1;  // (c + 2) % 3
        chunk(std::true_type{}, nbe, ring + s_cur * STAGE, c + 2, ring + s_nn * STAGE);
        // chunk c+1 has landed (the pieces of c+2 may stay in flight); everybody is done reading chunk c
        // (no timestamp reads in here: the branches around them split the loop body into several basic blocks, and the
        //  compiler then sinks MFMAs past the wait and the barrier)
#ifdef NB_ABL_NODMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");
#endif
#ifndef NB_ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
        s_cur = s_cur == 2 ? 0 : s_cur + 1;
    }
    for (; c < NC; ++c) {
        chunk(std::false_type{}, nbe, ring + s_cur * STAGE, 0, nullptr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        s_cur = s_cur == NST - 1 ? 0 : s_cur + 1;
    }
    };
    // (copies of the loop per number of column blocks a wave really has to multiply: blocks that do not exist -- the 16th of the
    //  15-block tile -- or lie wholly below the image -- the last tile row of a height that 12 does not divide -- cost no MFMAs and
    //  no fragment reads; the DMA pieces and barriers are the same in every copy.  Their accumulators stay zero; nothing that is
    //  stored reads them.)
    const int nvalid_blk = (min(TQH, H - I0) + 2) * PW;                  // positions of the rows that feed stored pixels ...
    int nbe_w = 0;                                                       // ... and this wave's blocks that hold any of them
#pragma unroll
    for (int j = 0; j < NBJ; ++j) nbe_w += (wv + NW * j) * 32 < nvalid_blk && j < nblk;
    // the wave whose last multiplied block is the tile's last one, when that block holds nothing but the last halo row
    constexpr bool LASTROW_BLK = (NBLK - 1) * 32 >= (PH - 1) * PW;
    const bool half_last = LASTROW_BLK && !(p.dbg & 16) && wv == (NBLK - 1) % NW && nbe_w == (NBLK - 1) / NW + 1;
    if constexpr (NBJ > 1) {
        if (nbe_w == 2) { if (half_last) kloop(std::integral_constant<int, 2 | 4>{}); else kloop(std::integral_constant<int, 2>{}); }
        else if (nbe_w == 1) { if (half_last) kloop(std::integral_constant<int, 1 | 4>{}); else kloop(std::integral_constant<int, 1>{}); }
        else kloop(std::integral_constant<int, 0>{});
    } else {
        if (nbe_w == 1) { if (half_last) kloop(std::integral_constant<int, 1 | 4>{}); else kloop(std::integral_constant<int, 1>{}); }
        else kloop(std::integral_constant<int, 0>{});
    }
    if (p.tstamps && tid == 0) {
        unsigned long long* ts = p.tstamps + (size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8;
        ts[6] = t_dma; ts[7] = t_bar | ((__builtin_amdgcn_s_memtime() - t_loop0) << 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // drain before the staging LDS is reused
    __builtin_amdgcn_s_barrier();

    NB_TSTAMP(2);
    if (p.dbg & 4) { if (acc[0][0][0] == 123.456f) p.y[0] = 0.f; return; }      // ablation: main loop only
    // ---- epilogue: 2 rounds of 16 c_out.  Accumulator register rho <-> c_out row (rho&3) + 8(rho>>2) + 4*lh, so the four
    //      registers rho = 4g..4g+3 of a lane are four CONSECUTIVE channels: they go to LDS as one 16-byte slot
    //      y4[g&1][lh][phase][position].  Then one item = one quad (2 x 2 output pixels) x those 4 channels: 25 slot reads,
    //      the separable polyphase FIR with every instruction packed over channel pairs (no lane-half shuffles), activation,
    //      hi/lo split and fp8 conversion in registers.  Lanes l and l+32 hold the two channel halves of the same quad and
    //      trade rows with v_permlane32_swap, after which a lane owns 8 channels of one output row of the quad = whole
    //      16-byte H2 slots (8-byte halves of the f8 slots) that go straight to global memory: no staging buffer, no
    //      transposition stage, 4 barriers instead of 12. ----
    const int Wo = 2 * W, Ho = 2 * H;
    constexpr int nquads = TQH * TQW;
    constexpr int Y1P = NBLK * 32;                    // slots per (g, lh, phase) plane
    f32x4* y4 = reinterpret_cast<f32x4*>(smem_h3);
    const float clampv = p.clamp >= 0.f ? p.clamp : __builtin_inff();     // (no clamp: med3 against +-inf)
    unsigned long long te_w = 0, te_f = 0, te0 = 0;   // debug: cycles in (phase write + sync), (items)
#pragma unroll
    for (int R = 0; R < 2; ++R) {
        if (p.tstamps) te0 = __builtin_amdgcn_s_memtime();
        // LDS-only barriers: __syncthreads() would also wait (vmcnt(0)) for the previous round's global stores to retire
        if (R) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the previous round's reads are done
#pragma unroll
        for (int gs = 0; gs < 2; ++gs) {
            const int r0 = (2 * R + gs) * 4;
#pragma unroll
            for (int j = 0; j < NBJ; ++j) {
                if (j < nblk) {
                    const int pidx = (wv + NW * j) * 32 + l31;
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
                        y4[((gs * 2 + lh) * 4 + ph) * Y1P + pidx] = f32x4{acc[j][ph][r0], acc[j][ph][r0 + 1], acc[j][ph][r0 + 2], acc[j][ph][r0 + 3]};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (p.tstamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); te_w += t_ - te0; te0 = t_; }
        auto quad_item = [&](const int wi) {
            const int hq = wi * 32 + l31;             // (g, quad); lane half = channel half
            const int gs = hq / nquads, qd = hq - gs * nquads;
            const int ti = qd / TQW, tj = qd - ti * TQW;
            if (I0 + ti >= H) return;                 // quad row below the image (ragged last tile row): nothing of it is stored
            const int c4 = 8 * (2 * R + gs) + 4 * lh; // the lane's four channels within the slice
            const f32x4* ee = y4 + ((gs * 2 + lh) * 4) * Y1P + ti * PW + tj;
            const f32x4* eo = ee + 1 * Y1P;
            const f32x4* oe = ee + 2 * Y1P;
            const f32x4* oo = ee + 3 * Y1P;
            // 4-tap polyphase FIR 0.25 a + 0.75 b + 0.75 c + 0.25 d as one multiply + three fused multiply-adds (the
            // reference's upfirdn2d is a convolution whose summation order and fusing are the backend's)
            auto fir4 = [](f32x4 a, f32x4 b, f32x4 c, f32x4 d) {
                f32x4 q75, q25;
                q75 = 0.75f; q25 = 0.25f;
                return __builtin_elementwise_fma(q25, d, __builtin_elementwise_fma(q75, c, __builtin_elementwise_fma(q75, b, 0.25f * a)));
            };
            // vertical pass at the quad's 5 intermediate columns: even ones (ve) from (ee, oe) at quad columns tj, tj+1,
            // odd ones (vo) from (eo, oo) at tj .. tj+2; [dy] = output row of the quad
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
            // g (o d + noise + bias), lrelu, clamp with g folded into d, noise and bias (s_dco, s_noise, s_bias hold g d,
            // g noise, g bias)
            auto act4 = [&](f32x4 o, float nz) {
                f32x4 t = __builtin_elementwise_fma(o, d4, b4 + nz);
                const f32x4 ta = t * p.alpha;
                // lrelu = max(t, alpha t) for 0 <= alpha <= 1 (the launcher checks); med3 with +inf is a max without the NaN
                // canonicalisation instructions fmaxf() costs
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(t[i], ta[i], __builtin_inff()), -clampv, clampv);
                return t;
            };
            f32x4 v[2][2];                            // [dy][px]
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const f32x2 nz = *reinterpret_cast<const f32x2*>(s_noise + (2 * ti + dy) * (2 * TQW) + 2 * tj);
                // horizontal pass: even pixel = fir4(vo[0], ve[0], vo[1], ve[1]), odd pixel = fir4(ve[0], vo[1], ve[1], vo[2])
                // The two noise values get registers of their own.  Left in the loaded pair, the compiler adds the second one
                // as `v_pk_add_f32 d, bias, v[pair] op_sel:[0,1]` (low result reads the pair's HIGH dword), and that
                // instruction returned, for 8-16 lanes of the upper half-wave and run-to-run differently, the sum with the
                // pair's LOW dword in its low result (measured: tools/determinism_up2.py; waits forced to zero and s_nops
                // did not change it, this did).
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
                unsigned hi[2][2][2], lo[2][2][2];    // [dy][px][dword]: hi = 4 x f16; lo = H2: 4 x f16 residual, f8: {fp8(xl 2^9) x 4, fp8(v/4) x 4}
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        f32x4 w = v[dy][px] * ns4;
                        const h2 h01 = __builtin_convertvector(f32x2{w[0], w[1]}, h2), h23 = __builtin_convertvector(f32x2{w[2], w[3]}, h2);
                        // residual w - f16(w) as a mixed-precision fma (reads the f16 half directly)
                        const f32x4 xl = {nb_sub_f16(w[0], h01, false), nb_sub_f16(w[1], h01, true), nb_sub_f16(w[2], h23, false), nb_sub_f16(w[3], h23, true)};
                        hi[dy][px][0] = __builtin_bit_cast(unsigned, h01); hi[dy][px][1] = __builtin_bit_cast(unsigned, h23);
                        if constexpr (OUTM == 1) {
                            const h2 l01 = __builtin_convertvector(f32x2{xl[0], xl[1]}, h2), l23 = __builtin_convertvector(f32x2{xl[2], xl[3]}, h2);
                            lo[dy][px][0] = __builtin_bit_cast(unsigned, l01); lo[dy][px][1] = __builtin_bit_cast(unsigned, l23);
                        } else {
                            // (conversions saturate: FP16_OVFL is set for this kernel)
                            // (x 2^9 and x 2^-2 inside the conversions: nb_pk4_fp8_sat_scaled)
                            lo[dy][px][0] = nb_pk4_fp8_sat_scaled(xl[0], xl[1], xl[2], xl[3], 0x1p-9f);
                            lo[dy][px][1] = nb_pk4_fp8_sat_scaled(w[0], w[1], w[2], w[3], 4.f);
                        }
                    }
                // lanes l (channels 0-3 of the group) and l+32 (channels 4-7) trade rows: afterwards a lane holds output row
                // dy = lh of the quad with all 8 channels -- a = channels 0-3, b = channels 4-7
                unsigned ha[2][2], hb[2][2], la[2][2], lb[2][2];     // [px][dword]
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
                            // the 16-channel chunk's two lo slots: (even group, lo) = fp8(xl 2^9) of its 16 channels, (odd group,
                            // lo) = fp8(v/4); this group's 8 channels are bytes 8 (cg & 1) .. + 7 of both
                            _Float16* lo_xl = p.yh2 + ((size_t)n * p.c8_next + (cg & ~1)) * 2 * OHW8 + OHW8 + opix8 + px * 8 + (cg & 1) * 4;
                            *reinterpret_cast<u32x2*>(lo_xl) = u32x2{la[px][0], lb[px][0]};
                            *reinterpret_cast<u32x2*>(lo_xl + 2 * OHW8) = u32x2{la[px][1], lb[px][1]};
                        }
                    }
                }
            }
        };
        constexpr int NWI = nquads * 2 / 32;          // wave-iterations of 32 quads x both channel halves per round
        static_assert(nquads * 2 % 32 == 0, "tile quads must fill whole waves");
        if constexpr (NWI % NW == 0) {
#pragma unroll
            for (int k = 0; k < NWI / NW; ++k) quad_item(wv + k * NW);
        } else {
            for (int wi = wv; wi < NWI; wi += NW) quad_item(wi);
        }
        if (p.tstamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); te_f += t_ - te0; }
    }
    const unsigned long long te_s = 0;
    NB_TSTAMP(4);
    if (p.tstamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NB_TSTAMP(5);
        if (threadIdx.x == 0) p.tstamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + 3] = (te_w & 0x1fffff) | ((te_f & 0x1fffff) << 21) | ((te_s & 0x1fffff) << 42);
    }
}

static int g_force_tqh = -1;
// developer / test hook: 0 = automatic tile choice, NB_H3_TQH, NB_H3_TQH_MID or NB_H3_TQH_SMALL = force that tile height
extern "C" void nb_debug_set_up2_tile(int tqh) { g_force_tqh = tqh; }
// the 8-wave form with the software-pipelined K loop (nb_modconv_up2v.hip)
bool nb_up2v_eligible(int in_fmt, int c_in, int h, int w);
int nb_up2v_launch(H3Up2Params p, int n, int in_fmt, void* stream, unsigned long long* tstamps, int tstamps_cap);
#ifndef NB_UP2V_AUTO
#define NB_UP2V_AUTO 1          // 1: f8 launches on the 12-row throughput tiles run on the software-pipelined kernel
#endif
static int g_force_v2 = -1;
// developer / test hook: -1 = automatic, 0 = never, 1 = wherever the shape allows (f8 operands, w % 32 == 0; always 12-row tiles)
extern "C" void nb_debug_set_up2_v2(int mode) { g_force_v2 = mode; }
static int g_force_pair = -1;
// developer / test hook: -1 = automatic, 0 / 1 = never / always the two-workgroups-per-CU form (4 waves, 12 x 16 tiles, 2 stages)
extern "C" void nb_debug_set_up2_pair(int mode) { g_force_pair = mode; }
template <int TQH, int TQW, bool F8, int OUTM, int NW, int NST>
static int nb_up2_h3_launch1(const H3Up2Params& p, int n, size_t lds, void* stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_up2_h3_kernel<F8, TQH, OUTM, TQW, NW, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(p.tiles_x * p.tiles_y * p.slices, n);
    hipLaunchKernelGGL((modconv3x3_up2_h3_kernel<F8, TQH, OUTM, TQW, NW, NST>), grid, dim3(NW * 64), lds, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("modconv3x3_up2_h3");
    return NB_OK;
}

template <int TQH, int TQW = 32, int NW = 8, int NST = NB_H3_STAGES>
static int nb_up2_h3_launch(H3Up2Params p, int n, int in_fmt, void* stream) {
    p.tiles_x = p.w / TQW;
    p.tiles_y = (p.h + TQH - 1) / TQH;
    p.tstamps = (g_tstamps && (long long)p.tiles_x * p.tiles_y * p.slices * n <= g_tstamps_cap) ? g_tstamps : nullptr;
    constexpr int XPL = (TQH + 3) * (TQW + 3);
    constexpr int NBLK_ = ((TQH + 2) * (TQW + 2) + 31) / 32;
    constexpr size_t lds_stage = (size_t)NST * (4 * XPL + 36 * 32) * 16;
    static_assert(NBLK_ <= 2 * NW, "two column blocks per wave at most");
    constexpr size_t lds_epi = (size_t)16 * NBLK_ * 32 * 16;       // epilogue: [2 groups][2 halves][4 phases][positions] x 4 channels fp32
    const size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
    const int outm = p.yh2 ? (p.out_f8 ? 2 : 1) : 0;
    if constexpr (NW == 4) {      // the two-per-CU form exists for H2 operands only (its f8 form spills 1-3 registers: not with counted waits)
        NB_REQUIRE(!in_fmt, "modconv3x3_up2_h3: the two-workgroups-per-CU form takes H2 operands only");
        return outm == 2 ? nb_up2_h3_launch1<TQH, TQW, false, 2, NW, NST>(p, n, lds, stream) : outm == 1 ? nb_up2_h3_launch1<TQH, TQW, false, 1, NW, NST>(p, n, lds, stream)
                                                                                               : nb_up2_h3_launch1<TQH, TQW, false, 0, NW, NST>(p, n, lds, stream);
    } else
    if (in_fmt)
        return outm == 2 ? nb_up2_h3_launch1<TQH, TQW, true, 2, NW, NST>(p, n, lds, stream) : outm == 1 ? nb_up2_h3_launch1<TQH, TQW, true, 1, NW, NST>(p, n, lds, stream)
                                                                                              : nb_up2_h3_launch1<TQH, TQW, true, 0, NW, NST>(p, n, lds, stream);
    return outm == 2 ? nb_up2_h3_launch1<TQH, TQW, false, 2, NW, NST>(p, n, lds, stream) : outm == 1 ? nb_up2_h3_launch1<TQH, TQW, false, 1, NW, NST>(p, n, lds, stream)
                                                                                           : nb_up2_h3_launch1<TQH, TQW, false, 0, NW, NST>(p, n, lds, stream);
}

// which kernel a split-f16 up=2 launch runs on (the debug hooks apply; UP2_WIDE: the one-wave-per-SIMD form of round 4, removed in round 6)
enum Up2Form { UP2_BIG = 0, UP2_MID, UP2_SMALL, UP2_W16, UP2_PAIR, UP2_WIDE, UP2_V2, UP2_W8 };
static Up2Form nb_up2_h3_select(int in_fmt, int c_in, int c_out, int n, int h, int w) {
    const int tiles_x = w / 32, slices = (c_out + 31) / 32;
    if (w == 16) return UP2_W16;                      // 16-wide inputs: 8 x 16 quad tiles
    if (w == 8) return UP2_W8;                        // 8 x 8 inputs: the whole image is one 8 x 8 quad tile (4 position blocks)
    // tile height: the 12-row tiles unless they leave the chip mostly idle (batch-1 / interactive), then 5-row tiles
    const int force_tqh = g_force_tqh >= 0 ? g_force_tqh : 0;
    const long wgs_big = (long)n * tiles_x * ((h + NB_H3_TQH - 1) / NB_H3_TQH) * slices;
    const bool small_tiles = force_tqh ? force_tqh == NB_H3_TQH_SMALL : wgs_big < 160;
    // 8-row tiles when the 12-row tiles' last round of workgroups would leave most of the chip idle.  Estimate = rounds of 256
    // workgroups x cost of one (critical SIMD's position blocks 4 / 3, + prologue and epilogue ~ quads): b64.conv0 at batch 32
    // (32 input rows) is 384 workgroups = 1.5 rounds of 12-row tiles, but 512 = 2 full rounds of 8-row tiles at 0.73 the cost.
    bool mid_tiles = force_tqh == NB_H3_TQH_MID;
    if (!force_tqh && !small_tiles) {
        constexpr long ncu = 256;                     // MI355X (one workgroup of this kernel per CU)
        const long wgs_mid = (long)n * tiles_x * ((h + NB_H3_TQH_MID - 1) / NB_H3_TQH_MID) * slices;
        const double est_big = (double)((wgs_big + ncu - 1) / ncu) * 5.4, est_mid = (double)((wgs_mid + ncu - 1) / ncu) * 3.93;
        mid_tiles = est_mid < 0.9 * est_big;
    }
    // two 4-wave workgroups per CU on 12 x 16 tiles (see the kernel's NW_ / NST_) when that launch fills the chip as well
    const bool pair = !in_fmt && g_force_pair > 0;          // (opt-in through nb_debug_set_up2_pair: not faster than one 8-wave workgroup per CU)
    if (pair) return UP2_PAIR;
    // the 12-row throughput tiles of an f8 launch: the kernel with the software-pipelined K loop (same tile, same results)
    const int force_v2 = g_force_v2;
    if (force_v2 != 0 && nb_up2v_eligible(in_fmt, c_in, h, w) &&
        (force_v2 > 0 ? force_tqh == 0 || force_tqh == NB_H3_TQH : (NB_UP2V_AUTO && !mid_tiles && !small_tiles && (force_tqh == 0 || force_tqh == NB_H3_TQH))))
        return UP2_V2;
    if (mid_tiles) return UP2_MID;
    return small_tiles ? UP2_SMALL : UP2_BIG;
}

static int nb_up2_h3_impl(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                          int64_t noise_stride_n, const float* bias, float* y, void* y_h2, const float* next_styles,
                          int next_stride, int c_next, int n, int h, int w, int c_out, float alpha, float gain, float clamp,
                          void* stream, int in_fmt = 0, int out_fmt = 0) {
    NB_REQUIRE(in_fmt >= 0 && in_fmt <= 3 && (out_fmt == 0 || out_fmt == 1), "modconv3x3_up2_h3: input operand format must be 0 (H2), 1 (f8), 2 (f6) or 3 (f8, hi only), "
               "output format 0 or 1 (the up=2 epilogue does not write the f6 format)");
    NB_REQUIRE(in_fmt == 0 || c_in % 16 == 0, "modconv3x3_up2_h3: the f8 / f6 operand formats need c_in %% 16 == 0 (got %d)", c_in);
    NB_REQUIRE(out_fmt == 0 || (y_h2 && c_out % 16 == 0 && c_next % 16 == 0), "modconv3x3_up2_h3: f8 output needs an H2 destination and c_out, c_next %% 16 == 0");
    NB_REQUIRE(x_h2 && w_h3 && dcoefs && bias && ((y != nullptr) != (y_h2 != nullptr)), "modconv3x3_up2_h3: null pointer");
    NB_REQUIRE(!y_h2 || (next_styles && c_out % 8 == 0 && c_next >= c_out && next_stride >= c_out && (uintptr_t)y_h2 % 16 == 0),
               "modconv3x3_up2_h3: H2 output needs the consumer's styles, c_out %% 8 == 0 and c_next >= c_out");
    NB_REQUIRE(n > 0 && n <= 65535 && c_in > 0 && c_out > 0, "modconv3x3_up2_h3: bad sizes");
    NB_REQUIRE((w % 32 == 0 || w == 16 || w == 8) && h >= 8, "modconv3x3_up2_h3: needs w %% 32 == 0, w == 16 or w == 8 (got %dx%d)", h, w);
    NB_REQUIRE(alpha >= 0.f && alpha <= 1.f && gain > 0.f, "modconv3x3_up2_h3: leaky-ReLU slope must lie in [0, 1] and the gain be positive (got %g, %g)", alpha, gain);
    NB_REQUIRE(((uintptr_t)x_h2 | (uintptr_t)w_h3 | (uintptr_t)y) % 16 == 0, "modconv3x3_up2_h3: pointers must be 16-byte aligned");
    H3Up2Params p;
    p.x = (const _Float16*)x_h2; p.wts = (const _Float16*)w_h3; p.dcoefs = dcoefs; p.noise = noise; p.bias = bias; p.y = y;
    p.zeros = nb_zero_page_ptr();
    NB_REQUIRE(p.zeros, "modconv3x3_up2_h3: could not allocate the zero page");
    p.noise_stride_n = noise_stride_n;
    NB_REQUIRE(nb_noise_src_setup(noise, noise_stride_n, 2 * h, 2 * w, &p.noise, &p.noise_stride_n, &p.nsrc) == NB_OK, "modconv3x3_up2_h3: bad NbNoiseSrc (needs the "
               "transposed constant, the grid row, the strength, exactly one of norm_pos / positions, and res = the %dx%d output)", 2 * h, 2 * w);
    p.c8 = (c_in + 7) / 8; p.nchunks = (c_in + 15) / 16; p.c_out = c_out; p.co_ld = (c_out + 63) / 64 * 64; p.h = h; p.w = w;
    p.dbg = g_nb_debug_flags; p.stagger_ticks = g_nb_stagger_ticks;            // (developer hooks: nb_debug_set_flags / nb_debug_set_stagger)
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.tiles_x = w / 32; p.slices = (c_out + 31) / 32;
    p.yh2 = (_Float16*)y_h2; p.next_styles = next_styles; p.next_stride = next_stride; p.c8_next = (c_next + 7) / 8;
    p.out_f8 = out_fmt;
    const Up2Form form = nb_up2_h3_select(in_fmt, c_in, c_out, n, h, w);
    NB_REQUIRE(in_fmt != 2 || form == UP2_V2, "modconv3x3_up2_h3: f6 operands are taken by the 12-row software-pipelined kernel only (this launch: %dx%d, batch %d)", h, w, n);
    switch (form) {
        case UP2_W16: return nb_up2_h3_launch<8, 16>(p, n, in_fmt, stream);
        case UP2_W8: return nb_up2_h3_launch<8, 8>(p, n, in_fmt, stream);
        case UP2_PAIR: return nb_up2_h3_launch<NB_H3_TQH, 16, 4, 2>(p, n, in_fmt, stream);
        case UP2_V2: return nb_up2v_launch(p, n, in_fmt, stream, g_tstamps, g_tstamps_cap);
        case UP2_MID: return nb_up2_h3_launch<NB_H3_TQH_MID>(p, n, in_fmt, stream);
        case UP2_SMALL: return nb_up2_h3_launch<NB_H3_TQH_SMALL>(p, n, in_fmt, stream);
        default: return nb_up2_h3_launch<NB_H3_TQH>(p, n, in_fmt, stream);
    }
}

extern "C" int nb_modconv3x3_up2_h3_variant(int in_fmt, int c_in, int c_out, int n, int h, int w, char* buf, int buflen) {
    NB_REQUIRE(buf && buflen > 0, "modconv3x3_up2_h3_variant: bad buffer");
    NB_REQUIRE(in_fmt >= 0 && in_fmt <= 3 && n > 0 && c_in > 0 && c_out > 0 && h >= 8 && (w % 32 == 0 || w == 16 || w == 8), "modconv3x3_up2_h3_variant: bad shape");
    static const char* const names[] = {"modconv3x3_up2_h3_kernel", "modconv3x3_up2_h3_kernel", "modconv3x3_up2_h3_kernel", "modconv3x3_up2_h3_kernel",
                                        "modconv3x3_up2_h3_kernel", "modconv3x3_up2_h3_kernel", "modconv3x3_up2v_kernel", "modconv3x3_up2_h3_kernel"};
    snprintf(buf, buflen, "%s", names[nb_up2_h3_select(in_fmt, c_in, c_out, n, h, w)]);
    return NB_OK;
}

extern "C" int nb_modconv3x3_up2_h3(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                                    int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w, int c_out,
                                    float alpha, float gain, float clamp, void* stream) {
    return nb_up2_h3_impl(x_h2, c_in, w_h3, dcoefs, noise, noise_stride_n, bias, y, nullptr, nullptr, 0, 0, n, h, w, c_out,
                          alpha, gain, clamp, stream);
}

extern "C" int nb_modconv3x3_up2_h3_ex(const void* x, int c_in, const void* wts, const float* dcoefs, const float* noise,
                                      int64_t noise_stride_n, const float* bias, float* y_f32, void* y_h2,
                                      const float* next_styles, int next_stride, int c_next, int in_fmt, int out_fmt, int n,
                                      int h, int w, int c_out, float alpha, float gain, float clamp, void* stream) {
    return nb_up2_h3_impl(x, c_in, wts, dcoefs, noise, noise_stride_n, bias, y_f32, y_h2, next_styles, next_stride, c_next,
                          n, h, w, c_out, alpha, gain, clamp, stream, in_fmt, out_fmt);
}

extern "C" int nb_modconv3x3_up2_h3_h2(const void* x_h2, int c_in, const void* w_h3, const float* dcoefs, const float* noise,
                                       int64_t noise_stride_n, const float* bias, const float* next_styles, int next_stride,
                                       void* y_h2, int c_next, int n, int h, int w, int c_out, float alpha, float gain,
                                       float clamp, void* stream) {
    return nb_up2_h3_impl(x_h2, c_in, w_h3, dcoefs, noise, noise_stride_n, bias, nullptr, y_h2, next_styles, next_stride,
                          c_next, n, h, w, c_out, alpha, gain, clamp, stream);
}

// ------------------------------------------------------------------------------------------------
// fp32 NCHW -> H2 packing (optionally x two concatenated inputs, x per-(n,c) scale = the consumer's styles)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_h2_kernel(const float* __restrict__ x1, int c1, const float* __restrict__ x2, int c2,
                                                      const float* __restrict__ scale, _Float16* __restrict__ out, int c8, int hw) {
    const int n = blockIdx.z, cg = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const int c_in = c1 + c2;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cg * 8 + j;
        float v = 0.f;
        if (ch < c_in) {
            v = ch < c1 ? x1[((size_t)n * c1 + ch) * hw + pix] : x2[((size_t)n * c2 + (ch - c1)) * hw + pix];
            if (scale) v *= scale[(size_t)n * c_in + ch];
        }
        const _Float16 hh = (_Float16)v;
        hi[j] = hh;
        lo[j] = (_Float16)(v - (float)hh);
    }
    h8* o = reinterpret_cast<h8*>(out) + ((size_t)(n * c8 + cg) * 2) * hw + pix;
    o[0] = hi;
    o[hw] = lo;
}

extern "C" int nb_pack_h2_f32(const float* x1, int c1, const float* x2, int c2, const float* scale, void* out_h2, int n,
                              int hw, void* stream) {
    NB_REQUIRE(x1 && out_h2 && c1 > 0 && c2 >= 0 && (c2 == 0 || x2) && n > 0 && n <= 65535 && hw > 0, "pack_h2: bad arguments");
    const int c8 = (c1 + c2 + 7) / 8;
    dim3 grid((hw + 255) / 256, c8, n);
    hipLaunchKernelGGL(pack_h2_kernel, grid, dim3(256), 0, (hipStream_t)stream, x1, c1, x2, c2, scale, (_Float16*)out_h2, c8, hw);
    NB_CHECK_LAUNCH("pack_h2");
    return NB_OK;
}

// fp32 NCHW (x1 ++ x2) * scale -> the "f8" activation format (see modconv3x3_up1_h3_kernel): per 16-channel chunk the
// f16 high halves of both channel groups, fp8(xl * 2^9) and fp8(x / 4)
__global__ __launch_bounds__(256) void pack_h2f8_kernel(const float* __restrict__ x1, int c1, const float* __restrict__ x2, int c2,
                                                        const float* __restrict__ scale, _Float16* __restrict__ out, int c8, int hw) {
    const int n = blockIdx.z, chunk = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const int c_in = c1 + c2;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ch = chunk * 16 + j;
        float t = 0.f;
        if (ch < c_in) {
            t = ch < c1 ? x1[((size_t)n * c1 + ch) * hw + pix] : x2[((size_t)n * c2 + (ch - c1)) * hw + pix];
            if (scale) t *= scale[(size_t)n * c_in + ch];
        }
        v[j] = t;
    }
    h8 hi0, hi1;
    float xl[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const _Float16 hh = (_Float16)v[j];
        if (j < 8) hi0[j & 7] = hh; else hi1[j & 7] = hh;
        xl[j] = (v[j] - (float)hh) * 512.f;
    }
    i32x4 l8, h8v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        l8[q] = (int)nb_pk4_fp8(xl[4 * q], xl[4 * q + 1], xl[4 * q + 2], xl[4 * q + 3]);
        h8v[q] = (int)nb_pk4_fp8(v[4 * q] * 0.25f, v[4 * q + 1] * 0.25f, v[4 * q + 2] * 0.25f, v[4 * q + 3] * 0.25f);
    }
    h8* o = reinterpret_cast<h8*>(out) + ((size_t)(n * c8 + 2 * chunk) * 2) * hw + pix;
    o[0] = hi0;
    o[hw] = __builtin_bit_cast(h8, l8);
    o[2 * (size_t)hw] = hi1;
    o[3 * (size_t)hw] = __builtin_bit_cast(h8, h8v);
}

extern "C" int nb_pack_h2f8_f32(const float* x1, int c1, const float* x2, int c2, const float* scale, void* out, int n, int hw,
                                void* stream) {
    NB_REQUIRE(x1 && out && c1 > 0 && c2 >= 0 && (c2 == 0 || x2) && n > 0 && n <= 65535 && hw > 0, "pack_h2f8: bad arguments");
    NB_REQUIRE((c1 + c2) % 16 == 0, "pack_h2f8: the f8 operand format needs a multiple of 16 channels (got %d)", c1 + c2);
    const int c8 = (c1 + c2) / 8;
    dim3 grid((hw + 255) / 256, c8 / 2, n);
    hipLaunchKernelGGL(pack_h2f8_kernel, grid, dim3(256), 0, (hipStream_t)stream, x1, c1, x2, c2, scale, (_Float16*)out, c8, hw);
    NB_CHECK_LAUNCH("pack_h2f8");
    return NB_OK;
}

// fp32 NCHW (x1 ++ x2) * scale -> the "f6" activation format (nb_h3_common.h): per 16-channel chunk the f16 high halves of both channel
// groups and, in the two lo slots, 32 e2m3 fields (xl 2^11 / S, x / S interleaved) + the chunk's scale byte.  With cg0 / c8_total:
// into channel groups cg0.. of a tensor with c8_total groups (the geometry features behind a producer that wrote its own groups).
__global__ __launch_bounds__(256) void pack_h2f6_kernel(const float* __restrict__ x1, int c1, const float* __restrict__ x2, int c2,
                                                        const float* __restrict__ scale, int scale_stride, _Float16* __restrict__ out,
                                                        int c8_total, int cg0, int hw) {
    const int n = blockIdx.z, chunk = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const int c_in = c1 + c2;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ch = chunk * 16 + j;
        float t = 0.f;
        if (ch < c_in) {
            t = ch < c1 ? x1[((size_t)n * c1 + ch) * hw + pix] : x2[((size_t)n * c2 + (ch - c1)) * hw + pix];
            if (scale) t *= scale[(size_t)n * scale_stride + ch];
        }
        v[j] = t;
    }
    h8 hi0, hi1;
    i32x4 l0, l1;
    nb_f6_encode16(v, hi0, hi1, l0, l1);
    h8* o = reinterpret_cast<h8*>(out) + ((size_t)(n * c8_total + cg0 + 2 * chunk) * 2) * hw + pix;
    o[0] = hi0;
    o[hw] = __builtin_bit_cast(h8, l0);
    o[2 * (size_t)hw] = hi1;
    o[3 * (size_t)hw] = __builtin_bit_cast(h8, l1);
}

extern "C" int nb_pack_h2f6_f32(const float* x1, int c1, const float* x2, int c2, const float* scale, void* out, int n, int hw,
                                void* stream) {
    NB_REQUIRE(x1 && out && c1 > 0 && c2 >= 0 && (c2 == 0 || x2) && n > 0 && n <= 65535 && hw > 0, "pack_h2f6: bad arguments");
    NB_REQUIRE((c1 + c2) % 16 == 0, "pack_h2f6: the f6 operand format needs a multiple of 16 channels (got %d)", c1 + c2);
    const int c8 = (c1 + c2) / 8;
    dim3 grid((hw + 255) / 256, c8 / 2, n);
    hipLaunchKernelGGL(pack_h2f6_kernel, grid, dim3(256), 0, (hipStream_t)stream, x1, c1, x2, c2, scale, c1 + c2, (_Float16*)out, c8, 0, hw);
    NB_CHECK_LAUNCH("pack_h2f6");
    return NB_OK;
}

extern "C" int nb_pack_h2f6_part_f32(const float* x, int c, const float* scale, int scale_stride, void* out, int c8_total,
                                     int cg0, int n, int hw, void* stream) {
    NB_REQUIRE(x && out && c > 0 && c % 16 == 0 && cg0 % 2 == 0 && n > 0 && n <= 65535 && hw > 0 && cg0 >= 0 && cg0 + c / 8 <= c8_total,
               "pack_h2f6_part: bad arguments (whole 16-channel chunks only)");
    dim3 grid((hw + 255) / 256, c / 16, n);
    hipLaunchKernelGGL(pack_h2f6_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, c, (const float*)nullptr, 0, scale, scale_stride, (_Float16*)out,
                       c8_total, cg0, hw);
    NB_CHECK_LAUNCH("pack_h2f6_part");
    return NB_OK;
}

// Pack c channels of an fp32 NCHW tensor into channel groups cg0.. of an H2 tensor with c8_total groups (the geometry
// features that the consumer concatenates behind a producer that wrote its own groups directly)
__global__ __launch_bounds__(256) void pack_h2_part_kernel(const float* __restrict__ x, int c, const float* __restrict__ scale,
                                                           int scale_stride, _Float16* __restrict__ out, int c8_total, int cg0, int hw) {
    const int n = blockIdx.z, cgl = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cgl * 8 + j;
        float v = 0.f;
        if (ch < c) {
            v = x[((size_t)n * c + ch) * hw + pix];
            if (scale) v *= scale[(size_t)n * scale_stride + ch];
        }
        const _Float16 hh = (_Float16)v;
        hi[j] = hh;
        lo[j] = (_Float16)(v - (float)hh);
    }
    h8* o = reinterpret_cast<h8*>(out) + ((size_t)(n * c8_total + cg0 + cgl) * 2) * hw + pix;
    o[0] = hi;
    o[hw] = lo;
}

__global__ __launch_bounds__(256) void pack_h2f8_part_kernel(const float* __restrict__ x, int c, const float* __restrict__ scale,
                                                             int scale_stride, _Float16* __restrict__ out, int c8_total, int cg0, int hw) {
    const int n = blockIdx.z, chunk = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    float v[16], xl[16];
    h8 hi0, hi1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ch = chunk * 16 + j;
        float t = 0.f;
        if (ch < c) {
            t = x[((size_t)n * c + ch) * hw + pix];
            if (scale) t *= scale[(size_t)n * scale_stride + ch];
        }
        const _Float16 hh = (_Float16)t;
        if (j < 8) hi0[j & 7] = hh; else hi1[j & 7] = hh;
        v[j] = t; xl[j] = (t - (float)hh) * 512.f;
    }
    i32x4 l8, h8v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        l8[q] = (int)nb_pk4_fp8(xl[4 * q], xl[4 * q + 1], xl[4 * q + 2], xl[4 * q + 3]);
        h8v[q] = (int)nb_pk4_fp8(v[4 * q] * 0.25f, v[4 * q + 1] * 0.25f, v[4 * q + 2] * 0.25f, v[4 * q + 3] * 0.25f);
    }
    h8* o = reinterpret_cast<h8*>(out) + ((size_t)(n * c8_total + cg0 + 2 * chunk) * 2) * hw + pix;
    o[0] = hi0;
    o[hw] = __builtin_bit_cast(h8, l8);
    o[2 * (size_t)hw] = hi1;
    o[3 * (size_t)hw] = __builtin_bit_cast(h8, h8v);
}

extern "C" int nb_pack_h2f8_part_f32(const float* x, int c, const float* scale, int scale_stride, void* out, int c8_total,
                                     int cg0, int n, int hw, void* stream) {
    NB_REQUIRE(x && out && c > 0 && c % 16 == 0 && cg0 % 2 == 0 && n > 0 && n <= 65535 && hw > 0 && cg0 >= 0 && cg0 + c / 8 <= c8_total,
               "pack_h2f8_part: bad arguments (whole 16-channel chunks only)");
    dim3 grid((hw + 255) / 256, c / 16, n);
    hipLaunchKernelGGL(pack_h2f8_part_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, c, scale, scale_stride, (_Float16*)out, c8_total, cg0, hw);
    NB_CHECK_LAUNCH("pack_h2f8_part");
    return NB_OK;
}

extern "C" int nb_pack_h2_part_f32(const float* x, int c, const float* scale, int scale_stride, void* out_h2, int c8_total,
                                   int cg0, int n, int hw, void* stream) {
    NB_REQUIRE(x && out_h2 && c > 0 && n > 0 && n <= 65535 && hw > 0 && cg0 >= 0 && cg0 + (c + 7) / 8 <= c8_total, "pack_h2_part: bad arguments");
    dim3 grid((hw + 255) / 256, (c + 7) / 8, n);
    hipLaunchKernelGGL(pack_h2_part_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, c, scale, scale_stride, (_Float16*)out_h2, c8_total, cg0, hw);
    NB_CHECK_LAUNCH("pack_h2_part");
    return NB_OK;
}

// Device-side weight packing (same layout as the host helper below): one thread per (chunk, tap, cg, co) writes the hi and lo
// slots of its 8 channels (zero padding included)
// tf != 0: w is the weight of the convolution whose INPUT gradient is being computed, [c_in][c_out][3][3] in this call's terms; the
// packed weight is its transpose with the taps reversed (the transposed convolution's kernel) -- no flipped copy in between.
__global__ __launch_bounds__(256) void pack_conv_weight_h3_kernel(const float* __restrict__ w, int c_out, int c_in, int co_ld, int nch,
                                                                  h8* __restrict__ out, int tf) {
    const int idx = blockIdx.x * 256 + threadIdx.x;             // ((chunk * 9 + tap) * 2 + cg) * co_ld + co
    if (idx >= nch * 18 * co_ld) return;
    const int co = idx % co_ld;
    int r = idx / co_ld;
    const int cg = r & 1; r >>= 1;
    const int tap = r % 9, ch = r / 9;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ch * 16 + cg * 8 + j;
        const float v = (co < c_out && ci < c_in) ? (tf ? w[((size_t)ci * c_out + co) * 9 + 8 - tap] : w[((size_t)co * c_in + ci) * 9 + tap]) : 0.f;
        const _Float16 hh = (_Float16)v;
        hi[j] = hh;
        lo[j] = (_Float16)(v - (float)hh);
    }
    const size_t base = ((((size_t)ch * 9 + tap) * 2 + cg) * 2) * co_ld;
    out[base + co] = hi;
    out[base + co_ld + co] = lo;
}

extern "C" int nb_pack_conv_weight_h3_dev(const float* w, int c_out, int c_in, int co_align, int transpose_flip, void* out, void* stream) {
    NB_REQUIRE(w && out && c_out > 0 && c_in > 0 && (uintptr_t)out % 16 == 0 && (co_align == 64 || co_align == 128), "pack_conv_weight_h3_dev: bad arguments");
    const int nch = (c_in + 15) / 16, co_ld = (c_out + co_align - 1) / co_align * co_align;
    const int total = nch * 18 * co_ld;
    hipLaunchKernelGGL(pack_conv_weight_h3_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, c_out, c_in, co_ld, nch,
                       (h8*)out, transpose_flip);
    NB_CHECK_LAUNCH("pack_conv_weight_h3_dev");
    return NB_OK;
}

// Host-side weight packing: W[c_out][c_in][3][3] fp32 -> hi/lo f16 [ceil16(c_in)/16][3][3][2][2][ceil64(c_out)][8]
extern "C" int nb_pack_conv_weight_h3(const float* w, int c_out, int c_in, void* out) {
    NB_REQUIRE(w && out && c_out > 0 && c_in > 0, "pack_conv_weight_h3: bad arguments");
    const int nch = (c_in + 15) / 16, co_ld = (c_out + 63) / 64 * 64;
    _Float16* o = (_Float16*)out;
    const size_t total = (size_t)nch * 9 * 4 * co_ld * 8;
    for (size_t i = 0; i < total; ++i) o[i] = (_Float16)0.f;
    for (int co = 0; co < c_out; ++co)
        for (int ci = 0; ci < c_in; ++ci)
            for (int t = 0; t < 9; ++t) {
                const float v = w[((size_t)co * c_in + ci) * 9 + t];
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                const int ch = ci / 16, cg = (ci % 16) / 8, j = ci % 8;
                const size_t base = ((((size_t)ch * 9 + t) * 2 + cg) * 2) * co_ld;
                o[(base + co) * 8 + j] = hi;
                o[(base + co_ld + co) * 8 + j] = lo;
            }
    return NB_OK;
}
