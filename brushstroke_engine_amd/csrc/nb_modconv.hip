// Modulated 3x3 convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores.
//
// Math (reference: training/networks.py:30-88 modulated_conv2d, :362-391 SynthesisLayer.forward,
// torch_utils/ops/conv2d_resample.py:124-147):
//     y = clamp(lrelu(conv(x * s[n,c]) * d[n,o] + noise + bias[o]) * gain)
// i.e. the "scale activations, shared weights, scale outputs" form (networks.py:67-76), which is
// algebraically the fused per-sample-weight form (networks.py:55-64, 78-88) but turns the whole
// batch into ONE implicit GEMM against a single shared weight matrix:
//     A = weights  [M = c_out]            (MFMA A operand, from LDS, layout [k][tap][c_out])
//     B = s-scaled activations [N = pixels] (MFMA B operand, from an LDS halo tile [k][row][col])
//     K = c_in x taps, walked in chunks of KC input channels that are double-buffered in LDS.
// Output D[c_out, pixel] has the pixel on the lane, so NCHW stores are 128-B contiguous per row.
//
// up = 1 kernel: v_mfma_f32_32x32x2_f32, 4 waves, wave tile = (MB x 32 c_out) x (NBW x 32 pixels).
// up = 2 kernel: the stride-2 transposed convolution is evaluated as its 4 output phases
//     (even/odd row x even/odd col; 4+2+2+1 = 9 non-zero taps, so no multiply by stuffed zeros),
//     on a quad grid with a one-quad halo, v_mfma_f32_16x16x4_f32; the phase images y1 are then
//     passed through LDS to the fused 4x4 FIR ([1,3,3,1]x[1,3,3,1]/64 * 4, evaluated separably,
//     polyphase) + epilogue, so the (2H+1)^2 intermediate never reaches HBM.
#include "nb_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ModconvParams {
    const float* x1;
    const float* x2;
    const float* wpk;      // [c_in][9][c_out]
    const float* styles;   // [n][c_in]
    const float* dcoefs;   // [n][c_out]
    const float* noise;    // [n or 1][Hout][Wout] or null
    const float* bias;     // [c_out]
    float* y;              // [n][c_out][Hout][Wout]
    long long noise_stride_n;
    int c1, c2, c_in, c_out;
    int h, w;              // input resolution
    int log2_tw;           // up1: tile width = 1 << log2_tw
    int th;                // up1: tile rows actually staged (<= h); up2: quad rows per tile
    int tw;                // up2: quad cols per tile
    int tiles_x, tiles_y, slices;
    float alpha, gain, clamp;
};

__device__ __forceinline__ float nb_epilogue(float v, float bias, float alpha, float gain, float clamp) {
    v += bias;
    v = v < 0.f ? v * alpha : v;
    v *= gain;
    if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    return v;
}

// ------------------------------------------------------------------------------------------------
// up = 1
// ------------------------------------------------------------------------------------------------
template <int MB, int NBW, int KC>
__global__ __launch_bounds__(256) void modconv3x3_up1_kernel(const ModconvParams p) {
    constexpr int CO_WG = MB * 32;
    constexpr int XPLANE_MAX = 352;                 // >= (th+2)*(tw+2) for every tile shape (10x34 = 340, 18x18 = 324)
    constexpr int XBUF = KC * XPLANE_MAX;
    constexpr int WBUF = KC * 9 * CO_WG;
    constexpr int XE = (XPLANE_MAX + 255) / 256;    // halo elements per thread per channel
    constexpr int WE = (WBUF / 4 + 255) / 256;      // float4 weight elements per thread per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                 // [2][KC][plane]
    float* wsm = smem + 2 * XBUF;     // [2][KC][9][CO_WG]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, lh = lane >> 5, l31 = lane & 31;
    const int H = p.h, W = p.w;
    const int TW = 1 << p.log2_tw, XS = TW + 2;
    const int th = p.th;
    const int plane = (th + 2) * XS;

    int b = blockIdx.x;
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x; const int tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int y0 = tile_y * th, x0 = tile_x * TW;
    const int co0 = slice * CO_WG;
    const int HW = H * W;

    // ---- per-thread staging geometry (identical for every channel chunk) ----
    int xoff[XE];      // offset inside one channel plane of the global image, -1 = zero fill
    int xlds[XE];      // offset inside one channel plane of the LDS tile, -1 = not mine
#pragma unroll
    for (int i = 0; i < XE; ++i) {
        const int e = tid + i * 256;
        xoff[i] = -1; xlds[i] = -1;
        if (e < plane) {
            const int r = e / XS, c = e - r * XS;
            const int gy = y0 - 1 + r, gx = x0 - 1 + c;
            xlds[i] = e;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) xoff[i] = gy * W + gx;
        }
    }
    const float* sty = p.styles + (size_t)n * p.c_in;

    float xreg[KC][XE], sreg[KC];
    f32x4 wreg[WE];

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int ch = c0 + k;
            const bool chv = ch < p.c_in;
            const float* src = nullptr;
            float s = 0.f;
            if (chv) {
                src = ch < p.c1 ? p.x1 + ((size_t)n * p.c1 + ch) * HW : p.x2 + ((size_t)n * p.c2 + (ch - p.c1)) * HW;
                s = sty[ch];
            }
            sreg[k] = s;      // the style multiply happens at store_chunk, so no wait on these loads sits before the MFMAs
#pragma unroll
            for (int i = 0; i < XE; ++i) {
                float v = 0.f;
                if (chv && xoff[i] >= 0) v = src[xoff[i]];
                xreg[k][i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < WE; ++i) {
            const int e4 = tid + i * 256;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e4 < WBUF / 4) {
                const int row = e4 / (CO_WG / 4), j4 = e4 - row * (CO_WG / 4);   // row = k*9 + tap
                const int ch = c0 + row / 9;
                if (ch < p.c_in && co0 + j4 * 4 < p.c_out)
                    v = *reinterpret_cast<const f32x4*>(p.wpk + ((size_t)c0 * 9 + row) * p.c_out + co0 + j4 * 4);
            }
            wreg[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        float* xd = xs + buf * XBUF;
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int i = 0; i < XE; ++i)
                if (xlds[i] >= 0) xd[k * plane + xlds[i]] = xreg[k][i] * sreg[k];
        float* wd = wsm + buf * WBUF;
#pragma unroll
        for (int i = 0; i < WE; ++i) {
            const int e4 = tid + i * 256;
            if (e4 < WBUF / 4) *reinterpret_cast<f32x4*>(wd + e4 * 4) = wreg[i];
        }
    };

    // ---- B-fragment base offsets (pixel -> halo tile position) ----
    int boff[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int q = (wv * NBW + nb) * 32 + l31;
        int ty = q >> p.log2_tw;
        const int tx = q & (TW - 1);
        ty = ty < th ? ty : th - 1;       // rows past the tile are computed on clamped data and never stored
        boff[nb] = ty * XS + tx;
    }

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    const int nchunks = (p.c_in + KC - 1) / KC;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ck = 0; ck < nchunks; ++ck) {
        const int buf = ck & 1;
        if (ck + 1 < nchunks) load_chunk((ck + 1) * KC);
        const float* xb = xs + buf * XBUF + lh * plane;
        const float* wb = wsm + buf * WBUF + lh * 9 * CO_WG + l31;
        // software pipeline: the fragments of step s+1 are read from LDS right after the first MFMA of
        // step s has issued, so their latency rides under the remaining MB*NBW-1 MFMAs (64 cycles each)
        constexpr int STEPS = (KC / 2) * 9;
        float af[2][MB], bfr[2][NBW];
        auto fetch = [&](int step, float (&a)[MB], float (&b)[NBW]) {
            const int kk = step / 9, tap = step - kk * 9;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = wb[(kk * 2 * 9 + tap) * CO_WG + mb * 32];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) b[nb] = xb[kk * 2 * plane + boff[nb] + ky * XS + kx];
        };
        fetch(0, af[0], bfr[0]);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int cur = step & 1;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][0], bfr[cur][0], acc[0][0], 0, 0, 0);
            if (step + 1 < STEPS) fetch(step + 1, af[cur ^ 1], bfr[cur ^ 1]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    if (mb + nb > 0)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mb], bfr[cur][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (step + 1 < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, MB + NBW, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MB * NBW - 1, 0);
        }
        if (ck + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: *d, +noise, +bias, lrelu, gain, clamp; D[row = c_out, col = pixel] ----
    const int Wo = W, Ho = H;
    const float* dco = p.dcoefs + (size_t)n * p.c_out;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int q = (wv * NBW + nb) * 32 + l31;
        const int ty = q >> p.log2_tw, tx = q & (TW - 1);
        const int oy = y0 + ty, ox = x0 + tx;
        const bool ok = ty < th && oy < Ho;
        float nz = 0.f;
        if (ok && p.noise) nz = p.noise[(size_t)n * p.noise_stride_n + (size_t)oy * Wo + ox];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ok && co < p.c_out) {
                    float v = acc[mb][nb][r] * dco[co] + nz;
                    v = nb_epilogue(v, p.bias[co], p.alpha, p.gain, p.clamp);
                    p.y[((size_t)n * p.c_out + co) * ((size_t)Ho * Wo) + (size_t)oy * Wo + ox] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// up = 2  (transposed conv as 4 phases on a haloed quad grid + fused polyphase FIR)
//
// Quad grid of a tile with origin (I0, J0) and TQH x TQW interior quads: positions (r, c),
// r in [0, TQH+2), c in [0, TQW+2).  For the ODD row phase position r is quad I0-1+r, for the EVEN
// row phase it is quad I0+r (same for columns), so that one LDS address set serves all phases:
//   phase(ee) = W[0,0]*X(r+1,c+1) + W[0,2]*X(r+1,c) + W[2,0]*X(r,c+1) + W[2,2]*X(r,c)
//   phase(eo) = W[0,1]*X(r+1,c)   + W[2,1]*X(r,c)
//   phase(oe) = W[1,0]*X(r,c+1)   + W[1,2]*X(r,c)
//   phase(oo) = W[1,1]*X(r,c)
// with X(t,u) = input pixel (I0-1+t, J0-1+u), zero outside the image (this also yields the zero rows
// y1[-1] and y1[2H+1] the FIR padding needs).  y1 phase images:  ee[r][c] = y1[2(I0+r), 2(J0+c)],
// oo[r][c] = y1[2(I0-1+r)+1, 2(J0-1+c)+1], etc.  FIR (per axis taps [1,3,3,1]/4):
//   y[2i]   = .25*o[ti]  + .75*e[ti]   + .75*o[ti+1] + .25*e[ti+1]
//   y[2i+1] = .25*e[ti]  + .75*o[ti+1] + .75*e[ti+1] + .25*o[ti+2]         (i = I0 + ti)
// ------------------------------------------------------------------------------------------------
template <int NBP, int KC>
__global__ __launch_bounds__(256) void modconv3x3_up2_kernel(const ModconvParams p) {
    constexpr int CO_WG = 16;
    constexpr int XPLANE_MAX = 19 * 35;             // (TQH+3)*(TQW+3) for TQ = 16 x 32
    constexpr int XBUF = KC * XPLANE_MAX;
    constexpr int WBUF = KC * 9 * CO_WG;
    constexpr int XE = (XPLANE_MAX + 255) / 256;
    constexpr int NPOS_MAX = 4 * NBP * 16;
    constexpr int Y1_PHASE = NPOS_MAX;               // floats per phase image
    constexpr int Y1_SLOT = 4 * Y1_PHASE + 16;       // +16: keep the 4 c_out slots on different banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                 // [2][KC][plane]
    float* wsm = smem + 2 * XBUF;     // [2][KC][9][16]
    float* y1s = smem;                // epilogue reuse: [4 slots][4 phases][NPOS_MAX]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, lq = lane >> 4, l15 = lane & 15;
    const int H = p.h, W = p.w;
    const int TQH = p.th, TQW = p.tw;
    const int PH = TQH + 2, PW = TQW + 2, NPOS = PH * PW;
    const int XS = TQW + 3;
    const int plane = (TQH + 3) * XS;

    int b = blockIdx.x;
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x; const int tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int I0 = tile_y * TQH, J0 = tile_x * TQW;
    const int co0 = slice * CO_WG;
    const int HW = H * W;

    int xoff[XE], xlds[XE];
#pragma unroll
    for (int i = 0; i < XE; ++i) {
        const int e = tid + i * 256;
        xoff[i] = -1; xlds[i] = -1;
        if (e < plane) {
            const int r = e / XS, c = e - r * XS;
            const int gy = I0 - 1 + r, gx = J0 - 1 + c;
            xlds[i] = e;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) xoff[i] = gy * W + gx;
        }
    }
    const float* sty = p.styles + (size_t)n * p.c_in;

    float xreg[KC][XE], sreg[KC];
    float wreg[(WBUF + 255) / 256];
    constexpr int WE = (WBUF + 255) / 256;

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int ch = c0 + k;
            const bool chv = ch < p.c_in;
            const float* src = nullptr;
            float s = 0.f;
            if (chv) {
                src = ch < p.c1 ? p.x1 + ((size_t)n * p.c1 + ch) * HW : p.x2 + ((size_t)n * p.c2 + (ch - p.c1)) * HW;
                s = sty[ch];
            }
            sreg[k] = s;      // the style multiply happens at store_chunk, so no wait on these loads sits before the MFMAs
#pragma unroll
            for (int i = 0; i < XE; ++i) {
                float v = 0.f;
                if (chv && xoff[i] >= 0) v = src[xoff[i]];
                xreg[k][i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < WE; ++i) {
            const int e = tid + i * 256;
            float v = 0.f;
            if (e < WBUF) {
                const int row = e >> 4, j = e & 15;            // row = k*9 + tap
                const int ch = c0 + row / 9;
                if (ch < p.c_in && co0 + j < p.c_out) v = p.wpk[((size_t)c0 * 9 + row) * p.c_out + co0 + j];
            }
            wreg[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        float* xd = xs + buf * XBUF;
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int i = 0; i < XE; ++i)
                if (xlds[i] >= 0) xd[k * plane + xlds[i]] = xreg[k][i] * sreg[k];
        float* wd = wsm + buf * WBUF;
#pragma unroll
        for (int i = 0; i < WE; ++i) {
            const int e = tid + i * 256;
            if (e < WBUF) wd[e] = wreg[i];
        }
    };

    // position blocks of this wave: block index wv*NBP + j, 16 positions each
    int boff[NBP];
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        int pidx = (wv * NBP + j) * 16 + l15;
        pidx = pidx < NPOS ? pidx : NPOS - 1;
        const int r = pidx / PW, c = pidx - r * PW;
        boff[j] = r * XS + c;
    }

    f32x4 acc[NBP][4];
#pragma unroll
    for (int j = 0; j < NBP; ++j)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) acc[j][ph] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = (p.c_in + KC - 1) / KC;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ck = 0; ck < nchunks; ++ck) {
        const int buf = ck & 1;
        if (ck + 1 < nchunks) load_chunk((ck + 1) * KC);
        const float* xb = xs + buf * XBUF + lq * plane;
        const float* wb = wsm + buf * WBUF + lq * 9 * CO_WG + l15;
        // software pipeline over (k-step, position block): block j+1's four B fragments (and, at the last
        // block of a k-step, the next k-step's nine A fragments) are read right after block j's first MFMA
        constexpr int KS = KC / 4;
        float af[2][9], xf[2][4];
        auto fetchA = [&](int ks, float (&a)[9]) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) a[tap] = wb[(ks * 4 * 9 + tap) * CO_WG];
        };
        auto fetchB = [&](int ks, int j, float (&x)[4]) {
            const float* xp = xb + ks * 4 * plane + boff[j];
            x[3] = xp[0];            // X(r, c)
            x[2] = xp[1];            // X(r, c+1)
            x[1] = xp[XS];           // X(r+1, c)
            x[0] = xp[XS + 1];       // X(r+1, c+1)
        };
        fetchA(0, af[0]);
        fetchB(0, 0, xf[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int j = 0; j < NBP; ++j) {
                const int it = ks * NBP + j;
                const int cb = it & 1, ca = ks & 1;
                const float x00 = xf[cb][0], x01 = xf[cb][1], x10 = xf[cb][2], x11 = xf[cb][3];
                const float(&a)[9] = af[ca];
                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], x00, acc[j][0], 0, 0, 0);
                int nreads = 0;
                if (j + 1 < NBP) { fetchB(ks, j + 1, xf[cb ^ 1]); nreads = 4; }
                else if (ks + 1 < KS) { fetchA(ks + 1, af[ca ^ 1]); fetchB(ks + 1, 0, xf[cb ^ 1]); nreads = 13; }
                // accumulator order keeps consecutive MFMAs on different accumulators (40-cycle dependent latency)
                acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], x01, acc[j][1], 0, 0, 0);
                acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], x10, acc[j][2], 0, 0, 0);
                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], x01, acc[j][0], 0, 0, 0);
                acc[j][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4], x11, acc[j][3], 0, 0, 0);
                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[6], x10, acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[7], x11, acc[j][1], 0, 0, 0);
                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[8], x11, acc[j][0], 0, 0, 0);
                acc[j][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[5], x11, acc[j][2], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (nreads == 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                if (nreads == 13) __builtin_amdgcn_sched_group_barrier(0x100, 13, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
        if (ck + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: 4 rounds of 4 c_out (accumulator register g <-> c_out rows {g, 4+g, 8+g, 12+g}) ----
    const int Wo = 2 * W, Ho = 2 * H;
    const float* dco = p.dcoefs + (size_t)n * p.c_out;
    const int nquads = TQH * TQW;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // D layout of v_mfma_f32_16x16x4_f32: col = lane & 15 (position), row = 4*(lane>>4) + reg (c_out)
#pragma unroll
        for (int j = 0; j < NBP; ++j) {
            const int pidx = (wv * NBP + j) * 16 + l15;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                y1s[lq * Y1_SLOT + ph * Y1_PHASE + pidx] = acc[j][ph][g];
            }
        }
        __syncthreads();
        // slot s <-> c_out row 4*s + g
        for (int it = tid; it < 4 * nquads; it += 256) {
            const int s = it / nquads, qd = it - s * nquads;
            const int ti = qd / TQW, tj = qd - ti * TQW;
            const int co = co0 + 4 * s + g;
            const float* ee = y1s + s * Y1_SLOT + 0 * Y1_PHASE + ti * PW + tj;
            const float* eo = y1s + s * Y1_SLOT + 1 * Y1_PHASE + ti * PW + tj;
            const float* oe = y1s + s * Y1_SLOT + 2 * Y1_PHASE + ti * PW + tj;
            const float* oo = y1s + s * Y1_SLOT + 3 * Y1_PHASE + ti * PW + tj;
            // vertical pass -> 2 output rows x (even cols c..c+1, odd cols c..c+2)
            float ve0[2], ve1[2], vo0[3], vo1[3];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float e0 = ee[c], e1 = ee[PW + c];
                const float o0 = oe[c], o1 = oe[PW + c], o2 = oe[2 * PW + c];
                ve0[c] = 0.25f * o0 + 0.75f * e0 + 0.75f * o1 + 0.25f * e1;
                ve1[c] = 0.25f * e0 + 0.75f * o1 + 0.75f * e1 + 0.25f * o2;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float e0 = eo[c], e1 = eo[PW + c];
                const float o0 = oo[c], o1 = oo[PW + c], o2 = oo[2 * PW + c];
                vo0[c] = 0.25f * o0 + 0.75f * e0 + 0.75f * o1 + 0.25f * e1;
                vo1[c] = 0.25f * e0 + 0.75f * o1 + 0.75f * e1 + 0.25f * o2;
            }
            float out[2][2];
            out[0][0] = 0.25f * vo0[0] + 0.75f * ve0[0] + 0.75f * vo0[1] + 0.25f * ve0[1];
            out[0][1] = 0.25f * ve0[0] + 0.75f * vo0[1] + 0.75f * ve0[1] + 0.25f * vo0[2];
            out[1][0] = 0.25f * vo1[0] + 0.75f * ve1[0] + 0.75f * vo1[1] + 0.25f * ve1[1];
            out[1][1] = 0.25f * ve1[0] + 0.75f * vo1[1] + 0.75f * ve1[1] + 0.25f * vo1[2];
            const int qi = I0 + ti, qj = J0 + tj;
            if (qi < H && qj < W && co < p.c_out) {
                const float d = dco[co], bs = p.bias[co];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int oy = 2 * qi + dy, ox = 2 * qj;
                    float n0 = 0.f, n1 = 0.f;
                    if (p.noise) {
                        const float* np_ = p.noise + (size_t)n * p.noise_stride_n + (size_t)oy * Wo + ox;
                        n0 = np_[0]; n1 = np_[1];
                    }
                    float2 o;
                    o.x = nb_epilogue(out[dy][0] * d + n0, bs, p.alpha, p.gain, p.clamp);
                    o.y = nb_epilogue(out[dy][1] * d + n1, bs, p.alpha, p.gain, p.clamp);
                    *reinterpret_cast<float2*>(p.y + ((size_t)n * p.c_out + co) * ((size_t)Ho * Wo) + (size_t)oy * Wo + ox) = o;
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int MB, int NBW, int KC>
static int launch_up1(ModconvParams p, int n, hipStream_t st) {
    constexpr int PIX_WG = 4 * NBW * 32;
    const int TW = p.w < 32 ? p.w : 32;
    int l2 = 0; while ((1 << l2) < TW) ++l2;
    p.log2_tw = l2;
    int th = PIX_WG / TW; if (th > p.h) th = p.h;
    p.th = th;
    if ((th + 2) * (TW + 2) > 352) { nb_set_error("modconv up1: tile %dx%d does not fit the LDS plane", th, TW); return NB_EINVAL; }
    p.tiles_x = p.w / TW; p.tiles_y = nb_cdiv(p.h, th); p.slices = nb_cdiv(p.c_out, MB * 32);
    const size_t lds = (size_t)(2 * KC * 352 + 2 * KC * 9 * MB * 32) * sizeof(float);
    dim3 grid(p.tiles_x * p.tiles_y * p.slices, n);
    hipLaunchKernelGGL((modconv3x3_up1_kernel<MB, NBW, KC>), grid, dim3(256), lds, st, p);
    NB_CHECK_LAUNCH("modconv3x3_up1");
    return NB_OK;
}

template <int NBP, int KC>
static int launch_up2(ModconvParams p, int n, hipStream_t st) {
    p.slices = nb_cdiv(p.c_out, 16);
    const size_t lds_main = (size_t)(2 * KC * 19 * 35 + 2 * KC * 9 * 16) * sizeof(float);
    const size_t lds_epi = (size_t)(4 * (4 * 4 * NBP * 16 + 16)) * sizeof(float);
    const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    dim3 grid(p.tiles_x * p.tiles_y * p.slices, n);
    hipLaunchKernelGGL((modconv3x3_up2_kernel<NBP, KC>), grid, dim3(256), lds, st, p);
    NB_CHECK_LAUNCH("modconv3x3_up2");
    return NB_OK;
}

extern "C" int nb_modconv3x3_f32(const float* x1, int c1, const float* x2, int c2, const float* wpk,
                                 const float* styles, const float* dcoefs, const float* noise,
                                 int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w,
                                 int c_out, int up, float alpha, float gain, float clamp, void* stream) {
    NB_REQUIRE(x1 && wpk && styles && dcoefs && bias && y, "modconv3x3: null pointer");
    NB_REQUIRE(c1 > 0 && c2 >= 0 && (c2 == 0 || x2), "modconv3x3: bad channel split c1=%d c2=%d", c1, c2);
    NB_REQUIRE(n > 0 && n <= 65535, "modconv3x3: batch %d out of range", n);
    NB_REQUIRE(h >= 1 && w >= 1 && (w & (w - 1)) == 0 && (h & (h - 1)) == 0, "modconv3x3: h,w must be powers of two (got %dx%d)", h, w);
    NB_REQUIRE(c_out > 0 && c_out % 4 == 0, "modconv3x3: c_out=%d must be a multiple of 4", c_out);
    NB_REQUIRE(up == 1 || up == 2, "modconv3x3: up=%d unsupported", up);
    hipStream_t st = (hipStream_t)stream;
    ModconvParams p;
    p.x1 = x1; p.x2 = x2; p.wpk = wpk; p.styles = styles; p.dcoefs = dcoefs; p.noise = noise; p.bias = bias; p.y = y;
    p.noise_stride_n = noise_stride_n;
    p.c1 = c1; p.c2 = c2; p.c_in = c1 + c2; p.c_out = c_out; p.h = h; p.w = w;
    p.log2_tw = 0; p.th = 0; p.tw = 0; p.tiles_x = p.tiles_y = p.slices = 1;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    if (up == 1) {
        // pick the c_out slice / pixels per workgroup so that small layers still give >= ~2 workgroups per CU
        const long pixels = (long)n * h * w;
        if (c_out % 128 == 0 && pixels >= 512L * 256) return launch_up1<4, 2, 4>(p, n, st);
        if (c_out % 64 == 0 && pixels >= 256L * 256) return launch_up1<2, 2, 4>(p, n, st);
        if (pixels >= 64L * 256) return launch_up1<1, 2, 4>(p, n, st);
        return launch_up1<1, 1, 4>(p, n, st);
    }
    const int tqw = w < 32 ? w : 32, tqh = h < 16 ? h : 16;
    p.th = tqh; p.tw = tqw; p.tiles_x = w / tqw; p.tiles_y = h / tqh;
    const int nblk = nb_cdiv((tqh + 2) * (tqw + 2), 16);
    const int nbp = nb_cdiv(nblk, 4);
    if (nbp <= 1) return launch_up2<1, 4>(p, n, st);
    if (nbp <= 2) return launch_up2<2, 4>(p, n, st);
    if (nbp <= 6) return launch_up2<6, 4>(p, n, st);
    return launch_up2<10, 4>(p, n, st);
}

extern "C" int nb_pack_conv_weight(const float* w, int c_out, int c_in, float* wpk, float* wsq) {
    NB_REQUIRE(w && c_out > 0 && c_in > 0, "pack_conv_weight: bad arguments");
    for (int o = 0; o < c_out; ++o)
        for (int i = 0; i < c_in; ++i) {
            float sq = 0.f;
            for (int t = 0; t < 9; ++t) {
                const float v = w[((size_t)o * c_in + i) * 9 + t];
                sq += v * v;
                if (wpk) wpk[((size_t)i * 9 + t) * c_out + o] = v;
            }
            if (wsq) wsq[(size_t)i * c_out + o] = sq;
        }
    return NB_OK;
}
