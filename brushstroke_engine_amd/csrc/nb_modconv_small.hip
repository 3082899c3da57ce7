// Modulated 3x3 convolution (up = 1) for SMALL images on the f16 matrix cores of gfx950: the <= 64x64 layers of the
// generator (b4 ... b64 conv1), which the large-tile split-f16 kernels of nb_modconv_h3.hip cannot fill the chip with and
// which the fp32-MFMA kernels of nb_modconv.hip run at the fp32 matrix rate (1/16 of f16) -- at batch 1 their launches
// are half of the interactive step, at batch 32 a sixth.
//
// Same math as everywhere (training/networks.py:30-88, :362-391):
//     y = clamp(lrelu(conv(x * s[n,c]) * d[n,o] + noise + bias[o]) * gain)
// in the "scale activations, shared weights" form with the split-f16 products of nb_modconv_h3.hip
//     x w  ~=  xh wh + xh wl + xl wh      (xh = f16(x s), xl = f16(x s - xh); same for w; error ~2^-22 relative).
//
// Workgroup = 4 waves = one tile of 32 c_out x 32 output positions, the four waves SPLIT K: wave w owns the 16-channel
// chunks w, w+4, ... of c_in.  Nothing is shared between the waves until the final reduction, so there is no barrier in
// the K loop:
//   * A fragments (weights, static hi/lo f16 in the layout of nb_pack_conv_weight_h3) go straight from global memory
//     into registers: lane (c_out = l & 31, channel group = l >> 5) reads one 16-byte slot per (tap, hi/lo), 512
//     contiguous bytes per half wave; the next chunk's 18 fragments are in flight under the current chunk's MFMAs.
//   * B fragments: the wave loads its chunk's fp32 activations for the tile + halo (lanes walk pixels: coalesced),
//     multiplies by the styles, splits into hi/lo f16 and writes 16-byte [position][8 channels] slots into its private
//     LDS region (zeros outside the image); the tap fragments are then single ds_read_b128 at halo offsets.
//   * 27 v_mfma_f32_32x32x16_f16 per chunk per wave.
// The four partial accumulators meet in LDS; wave w finishes c_out rows {8w + 4*lh + j} (demodulation, noise, bias,
// leaky ReLU, gain, clamp) and stores fp32 NCHW.
//
// A tile's 32 positions are 32 pixels of one image row segment (W >= 32), 32 / W whole rows (W < 32, H*W >= 32) or, for
// 4x4 images, the 16 pixels of TWO samples (per-position styles / demodulation / noise make that free).
#include "nb_common.h"
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NB_SM_NHP 104            // slots per (channel group, hi/lo) plane of the halo tile (max 3 x 34 = 102)
#define NB_SM_NHP2 136           // ... of a two-block tile (max 4 x 34 = 136)
#define NB_SM_MAX_CIN 512

struct SmallParams {
    const float* x;          // [n][c1][H][W]
    const float* x2;         // [n][c_in - c1][H][W] (channels c1 .. c_in-1; c1 % 16 == 0) or null
    const _Float16* wts;     // [up*up phases][nchunks][3][3][2 cg][2 hi/lo][co_ld][8]
    const float* styles;     // [n][c_in]
    const float* dcoefs;     // [n][c_out]
    const float* noise;      // [n or 1][H][W] or null
    const float* bias;       // [c_out]
    float* y;                // [n][c_out][up*H][up*W]
    long long noise_stride_n;
    int n, c_in, c1, nchunks, c_out, co_ld, h, w, up;
    int rows, cols, spt, tiles_x, slices;       // tile = spt samples x rows x cols pixels (= 32 positions)
    float alpha, gain, clamp;
};

__device__ __forceinline__ float nb_sm_epilogue(float v, float bias, float alpha, float gain, float clamp) {
    v += bias;
    v = v < 0.f ? v * alpha : v;
    v *= gain;
    if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    return v;
}

// OCC = workgroups per CU the register budget is sized for: 2 (256 VGPRs: 20 of them spill, but two workgroups share a CU
// when a launch has more than one round of them) or 1 (no spills: 1-2 us less per launch when every CU gets at most one
// workgroup anyway -- batch 1 and the <= 8x8 layers of a batch)
// NWV = waves that split K: 4, or 8 (one workgroup per CU) -- with the 128-channel layers' eight chunks every wave then has ONE
// chunk: one weight round trip per launch instead of two, which is most of what such a launch spends.
// TB = 32-position blocks per tile: 1, or 2 (twice the rows; one sample per tile) when the launch has more than a round of
// workgroups: the weight fragments a wave fetched serve both blocks, half as many workgroups fetch them at all.  Per-position
// arithmetic is the same in both forms (bit-identical results).
template <int OCC, int NWV = 4, int TB = 1>
__global__ __launch_bounds__(NWV * 64, OCC) void modconv3x3_up1_small_h3_kernel(const SmallParams p) {
    static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");
    static_assert(TB == 1 || TB == 2, "1 or 2 position blocks");
    constexpr int NT = NWV * 64, RPW = 16 / NWV;                                              // threads; accumulator rows a wave finishes
    constexpr int NHP = TB == 1 ? NB_SM_NHP : NB_SM_NHP2, NR = (2 * NHP + 63) / 64;           // plane slots; staging rounds of 64 tasks
    __shared__ __attribute__((aligned(16))) unsigned char smem[NWV * 4 * NHP * 16];         // [wave][plane][slot] (4 waves: 26 KB, fits next to a large-tile workgroup)
    __shared__ float s_sty[2 * NB_SM_MAX_CIN];
    __shared__ float s_epi[96];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    const int H = p.h, W = p.w, P = H * W;
    int b = blockIdx.x;
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x, tile_y = b / p.tiles_x;
    const int n0 = blockIdx.y * p.spt;
    const int y0 = tile_y * p.rows, x0 = tile_x * p.cols;
    const int co0 = slice * 32;
    const int phase = blockIdx.z, py = phase >> 1, px = phase & 1;          // up = 2: output pixel (2 oy + py, 2 ox + px)
    const int HR = p.rows + 2, HC = p.cols + 2, RC = p.rows * p.cols, NH = p.spt * HR * HC;

    // ---- staging tasks of this lane (chunk-invariant): task = round * 64 + lane -> (channel group, halo slot) ----
    // xoff / xoff2 = 32-bit element offset of the slot's pixel in channel cg*8 of the sample in x / x2, tslot = LDS slot
    // (-1: no such task - the slot lies outside the image or the tile and keeps the zero it is given once below), tsty = style row
    unsigned xoff[NR], xoff2[NR];
    const int c2 = p.c_in - p.c1;
    int tslot[NR], tsty[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int task = r * 64 + lane;
        const int cg = task / NHP, hp = task - cg * NHP;
        xoff[r] = 0; xoff2[r] = 0; tslot[r] = -1; tsty[r] = 0;
        if (cg < 2) {
            if (hp < NH) {
                const int s = hp / (HR * HC), rem = hp - s * (HR * HC);
                const int hy = rem / HC, hx = rem - hy * HC;
                const int gy = y0 + hy - 1, gx = x0 + hx - 1;
                if (n0 + s < p.n && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    xoff[r] = (unsigned)(((n0 + s) * p.c1 + cg * 8) * P + gy * W + gx);
                    xoff2[r] = (unsigned)(((n0 + s) * c2 + cg * 8) * P + gy * W + gx);
                    tsty[r] = s * NB_SM_MAX_CIN + cg * 8;
                    tslot[r] = cg * 2 * NHP + hp;               // plane (cg, hi); lo = + NHP
                }
            }
        }
    }
    // ---- this lane's output position(s) -- block j holds positions 32 j .. 32 j + 31 of the tile -- and their halo base slots ----
    int ps[TB], pty[TB], ptx[TB], pbase[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int pp = j * 32 + l31;
        ps[j] = pp / RC;
        const int prem = pp - ps[j] * RC;
        pty[j] = prem / p.cols; ptx[j] = prem - pty[j] * p.cols;
        const int hb = ps[j] * (HR * HC) + pty[j] * HC + ptx[j];        // + ky * HC + kx
        pbase[j] = lh * 2 * NHP + hb;                                   // plane (cg = lh, hi); lo = + NHP
    }

    h8* mybuf = reinterpret_cast<h8*>(smem) + wv * (4 * NHP);
    {   // zero padding: slots outside the image are never staged, they keep this zero
        const h8 z8 = {};
        for (int i = lane; i < 4 * NHP; i += 64) mybuf[i] = z8;
    }
    const unsigned wstep = (unsigned)(p.co_ld * 8);
    // + ((chunk*9 + tap)*4 + hl) * co_ld * 8  (halves); one weight set per output phase
    const unsigned wl = (unsigned)((lh * 2 * p.co_ld + co0 + l31) * 8) + (unsigned)phase * (unsigned)(p.nchunks * 36) * wstep;

    auto load_w = [&](int c, h8 (&wa)[9][2]) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
                wa[tap][hl] = *reinterpret_cast<const h8*>(p.wts + (wl + (unsigned)((c * 9 + tap) * 4 + hl) * wstep));
    };
    auto load_x = [&](int c, float (&xr)[NR][8]) {
        const bool second = c * 16 >= p.c1;                      // (wave-uniform: c1 % 16 == 0)
        const float* src = second ? p.x2 : p.x;
        const unsigned cbase = (unsigned)((c * 16 - (second ? p.c1 : 0)) * P);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const unsigned o = (second ? xoff2[r] : xoff[r]) + cbase;       // (tasks without a slot read element 0: unused)
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[r][j] = src[o + (unsigned)(j * P)];
        }
    };
    auto stage = [&](int c, const float (&xr)[NR][8], h8* buf) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (tslot[r] >= 0) {
                h8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = xr[r][j] * s_sty[tsty[r] + c * 16 + j];
                    const _Float16 hh = (_Float16)v;
                    hi[j] = hh;
                    lo[j] = (_Float16)(v - (float)hh);
                }
                buf[tslot[r]] = hi;
                buf[tslot[r] + NHP] = lo;
            }
        }
    };
    f32x16 acc[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    auto mfma_chunk = [&](const h8 (&wa)[9][2], const h8* buf) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int off = pbase[j] + (tap / 3) * HC + (tap % 3);
                const h8 bh = buf[off], bl = buf[off + NHP];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[tap][0], bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[tap][0], bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[tap][1], bh, acc[j], 0, 0, 0);
            }
        }
    };
    // wave-private LDS hand-over: all lanes' slot writes must have landed before any lane's fragment reads.  ONE buffer
    // suffices: a wave's LDS instructions execute in program order, so the next chunk's slot writes (issued after this
    // chunk's fragment reads) cannot overtake them
    auto wave_sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };

    h8 wa0[9][2];
    float xr0[NR][8];
    const int NC = p.nchunks;
    int c = wv;
    if (c < NC) { load_w(c, wa0); load_x(c, xr0); }
    // everything else the workgroup needs from global memory is requested now, under the first chunk's loads: the styles
    // (to LDS) and the epilogue's operands (registers) - a launch of this kernel is a few microseconds, so every exposed
    // round trip counts
    const int Wo = p.up * W;
    bool ok[TB];
    int ns[TB], oyo[TB], oxo[TB];
    float nz[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int oy = y0 + pty[j], ox = x0 + ptx[j];
        ns[j] = n0 + ps[j];
        ok[j] = ns[j] < p.n && j * 32 + l31 < p.spt * RC && oy < H && ox < W;
        oyo[j] = p.up * oy + py; oxo[j] = p.up * ox + px;
        nz[j] = (ok[j] && p.noise) ? p.noise[(size_t)ns[j] * p.noise_stride_n + (size_t)oyo[j] * Wo + oxo[j]] : 0.f;
    }
    if (tid < 96) {                                             // [2 samples][32 c_out] demodulation, [32] bias
        const int co = co0 + (tid & 31), sidx = tid >> 5;
        float v = 0.f;
        if (co < p.c_out) {
            if (sidx < 2) { if (n0 + sidx < p.n) v = p.dcoefs[(size_t)(n0 + sidx) * p.c_out + co]; }
            else v = p.bias[co];
        }
        s_epi[tid] = v;
    }
    for (int i = tid; i < p.spt * p.c_in; i += NT) {
        const int s = i / p.c_in, cc = i - s * p.c_in;
        s_sty[s * NB_SM_MAX_CIN + cc] = n0 + s < p.n ? p.styles[(size_t)(n0 + s) * p.c_in + cc] : 0.f;
    }
    __syncthreads();
    if constexpr (NWV == 8) {
        // one chunk per wave for up to 128 channels; a wider layer's further chunks are fetched after the current one's products
        // (no second register set: eight waves are two per SIMD, 256 registers each)
        while (c < NC) {
            stage(c, xr0, mybuf);
            wave_sync();
            mfma_chunk(wa0, mybuf);
            c += NWV;
            if (c < NC) { load_w(c, wa0); load_x(c, xr0); }
        }
    } else {
        h8 wa1[9][2];
        float xr1[NR][8];
        while (c < NC) {
            int cn = c + NWV;
            stage(c, xr0, mybuf);
            if (cn < NC) { load_w(cn, wa1); load_x(cn, xr1); }
            wave_sync();
            mfma_chunk(wa0, mybuf);
            c = cn;
            if (c >= NC) break;
            cn = c + NWV;
            stage(c, xr1, mybuf);
            if (cn < NC) { load_w(cn, wa0); load_x(cn, xr0); }
            wave_sync();
            mfma_chunk(wa1, mybuf);
            c = cn;
        }
    }

    // ---- split-K reduction through LDS (the staging buffers are dead after the barrier) ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                // [block][wave][reg 16][lane 64]
    static_assert(TB * NWV * 16 * 64 * 4 <= NWV * 4 * NHP * 16, "the reduction buffer must fit the staging buffers");
#pragma unroll
    for (int jb = 0; jb < TB; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((jb * NWV + wv) * 16 + r) * 64 + lane] = acc[jb][r];
    __syncthreads();
    const size_t Po = (size_t)p.up * p.up * P;
#pragma unroll
    for (int jb = 0; jb < TB; ++jb) {
    if (!ok[jb]) continue;
    const float* redb = red + (size_t)jb * NWV * 16 * 64;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int r = wv * RPW + j;                             // accumulator register r <-> c_out row (r & 3) + 8 (r >> 2) + 4 lh
        const int col = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = co0 + col;
        if (co < p.c_out) {
            // (the four-wave sum keeps its association; eight waves add the second four the same way)
            float sum = redb[(0 * 16 + r) * 64 + lane] + redb[(1 * 16 + r) * 64 + lane] + redb[(2 * 16 + r) * 64 + lane] + redb[(3 * 16 + r) * 64 + lane];
            if constexpr (NWV == 8)
                sum += redb[(4 * 16 + r) * 64 + lane] + redb[(5 * 16 + r) * 64 + lane] + redb[(6 * 16 + r) * 64 + lane] + redb[(7 * 16 + r) * 64 + lane];
            const float v = nb_sm_epilogue(sum * s_epi[ps[jb] * 32 + col] + nz[jb], s_epi[64 + col], p.alpha, p.gain, p.clamp);
            p.y[((size_t)ns[jb] * p.c_out + co) * Po + (size_t)oyo[jb] * Wo + oxo[jb]] = v;
        }
    }
    }
}

static int g_force_small_waves = 0, g_force_small_blocks = 0;
// developer / test hooks: 0 = automatic; 4 / 8 = that many K-splitting waves per workgroup; 1 / 2 = position blocks per tile
extern "C" void nb_debug_set_small_waves(int waves) { g_force_small_waves = waves; }
extern "C" void nb_debug_set_small_blocks(int blocks) { g_force_small_blocks = blocks; }

static int nb_small_h3_impl(const float* x, int c1, const float* x2, int c2, const void* w_h3, const float* styles, const float* dcoefs,
                            const float* noise, int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w, int c_out,
                            int up, float alpha, float gain, float clamp, void* stream) {
    const int c_in = c1 + c2;
    NB_REQUIRE(x && w_h3 && styles && dcoefs && bias && y && (c2 == 0 || x2), "modconv3x3_small_h3: null pointer");
    NB_REQUIRE(n > 0 && n <= 65535 && c1 > 0 && c2 >= 0 && c_in <= NB_SM_MAX_CIN && c1 % 16 == 0 && c2 % 16 == 0 && c_out > 0,
               "modconv3x3_small_h3: bad sizes (c1, c2 %% 16 == 0, c_in <= %d)", NB_SM_MAX_CIN);
    NB_REQUIRE(h >= 4 && w >= 4 && (w & (w - 1)) == 0 && (h & (h - 1)) == 0, "modconv3x3_small_h3: h, w must be powers of two >= 4 (got %dx%d)", h, w);
    NB_REQUIRE((long long)n * (c1 > c2 ? c1 : c2) * h * w < (1ll << 31), "modconv3x3_small_h3: input too large for 32-bit offsets");
    NB_REQUIRE(((uintptr_t)w_h3) % 16 == 0, "modconv3x3_small_h3: weights must be 16-byte aligned");
    SmallParams p;
    p.x = x; p.x2 = x2; p.wts = (const _Float16*)w_h3; p.styles = styles; p.dcoefs = dcoefs; p.noise = noise; p.bias = bias; p.y = y;
    p.noise_stride_n = noise_stride_n;
    p.n = n; p.c_in = c_in; p.c1 = c1; p.nchunks = c_in / 16; p.c_out = c_out; p.co_ld = (c_out + 63) / 64 * 64; p.h = h; p.w = w; p.up = up;
    NB_REQUIRE((long long)up * up * p.nchunks * 36 * p.co_ld * 8 < (1ll << 31), "modconv3x3_small_h3: weights too large for 32-bit offsets");
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    int tiles_y;
    if (w >= 32) { p.cols = 32; p.rows = 1; p.spt = 1; p.tiles_x = w / 32; tiles_y = h; }
    else if (h * w >= 32) {
        p.cols = w; p.rows = 32 / w; p.spt = 1; p.tiles_x = 1;
        NB_REQUIRE(h % p.rows == 0, "modconv3x3_small_h3: %dx%d images do not tile into 32-pixel row groups", h, w);
        tiles_y = h / p.rows;
    } else { p.cols = w; p.rows = h; p.spt = 32 / (h * w); p.tiles_x = 1; tiles_y = 1; }
    p.slices = (c_out + 31) / 32;
    NB_REQUIRE(p.spt <= 2 && p.spt * (p.rows + 2) * (p.cols + 2) <= NB_SM_NHP, "modconv3x3_small_h3: unsupported image size %dx%d", h, w);
    dim3 grid(p.tiles_x * tiles_y * p.slices, (n + p.spt - 1) / p.spt, up * up);
    const long wgs = (long)grid.x * grid.y * grid.z;
    // eight K-splitting waves when the chunks give every one of them work.  The rule looks at the layer only, never at the batch:
    // the two forms associate the partial sums differently, and a sample's result must not depend on what it is batched with.
    // (nb_debug_set_small_waves(4 / 8) forces the form; 0 = this rule)
    const int force_waves = g_force_small_waves > 0 ? g_force_small_waves : 0;
    const bool eight = force_waves ? force_waves == 8 : p.nchunks >= 8;
    // two position blocks per tile (twice the rows) when the launch is more than a round of workgroups: half as many
    // workgroups fetch the weights (nb_debug_set_small_blocks(1 / 2) forces the form where it exists)
    const int force_blocks = g_force_small_blocks > 0 ? g_force_small_blocks : 0;
    const bool can_two = eight && p.spt == 1 && tiles_y % 2 == 0 && (2 * p.rows + 2) * (p.cols + 2) <= NB_SM_NHP2;
    if (can_two && (force_blocks ? force_blocks == 2 : wgs > 256)) {
        p.rows *= 2;
        grid.x = p.tiles_x * (tiles_y / 2) * p.slices;
        hipLaunchKernelGGL((modconv3x3_up1_small_h3_kernel<1, 8, 2>), grid, dim3(512), 0, (hipStream_t)stream, p);
    } else
    if (eight) hipLaunchKernelGGL((modconv3x3_up1_small_h3_kernel<1, 8>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (wgs <= 256) hipLaunchKernelGGL((modconv3x3_up1_small_h3_kernel<1, 4>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((modconv3x3_up1_small_h3_kernel<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("modconv3x3_small_h3");
    return NB_OK;
}

extern "C" int nb_modconv3x3_up1_small_h3(const float* x, int c_in, const void* w_h3, const float* styles, const float* dcoefs,
                                          const float* noise, int64_t noise_stride_n, const float* bias, float* y, int n, int h,
                                          int w, int c_out, float alpha, float gain, float clamp, void* stream) {
    return nb_small_h3_impl(x, c_in, nullptr, 0, w_h3, styles, dcoefs, noise, noise_stride_n, bias, y, n, h, w, c_out, 1, alpha, gain, clamp, stream);
}

// up = 2 through the same kernel: the stride-2 transposed convolution followed by the 4x4 FIR (conv2d_resample.py:124-142,
// upfirdn2d pad 1 gain 4) is, per output phase (py, px), a plain 3x3 correlation of the INPUT grid with an effective kernel
//     Keff[py,px][o,c,di+1,dj+1] = sum_{a,b} W[o,c,a,b] * g[a - (py - 2 di) + 1][b - (px - 2 dj) + 1]     (g = flip(4 f), 0 outside)
// (a 6x6 kernel at stride 2 has 3x3 taps per phase).  The host folds the FIR into four weight sets (static: no styles
// involved) and the kernel runs the four phases as grid.z, each writing its quarter of the (2H x 2W) output.  That is 4x
// the matrix work of the 9-tap phase decomposition the large-tile kernels use - irrelevant at these sizes, where launches
// are latency bound - and no FIR epilogue.  w_h3_phases = 4 x nb_pack_conv_weight_h3(Keff[phase]), phase = 2 py + px.
extern "C" int nb_modconv3x3_up2_small_h3(const float* x1, int c1, const float* x2, int c2, const void* w_h3_phases, const float* styles,
                                          const float* dcoefs, const float* noise, int64_t noise_stride_n, const float* bias, float* y,
                                          int n, int h, int w, int c_out, float alpha, float gain, float clamp, void* stream) {
    return nb_small_h3_impl(x1, c1, x2, c2, w_h3_phases, styles, dcoefs, noise, noise_stride_n, bias, y, n, h, w, c_out, 2, alpha, gain, clamp, stream);
}
