// Modulated 3x3 convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores.
//
// Math (reference: training/networks.py:30-88 modulated_conv2d, :362-391 SynthesisLayer.forward,
// torch_utils/ops/conv2d_resample.py:124-147):
//     y = clamp(lrelu(conv(x * s[n,c]) * d[n,o] + noise + bias[o]) * gain)
// i.e. the "scale activations, shared weights, scale outputs" form (networks.py:67-76), which is
// algebraically the fused per-sample-weight form (networks.py:55-64, 78-88) but turns the whole
// batch into ONE implicit GEMM against a single shared weight matrix:
//     A = weights  [M = c_out]              (MFMA A operand, LDS image [k][tap][c_out], filled by LDS-DMA)
//     B = s-scaled activations [N = pixels] (MFMA B operand, LDS halo tile [k][row][col], 16-byte staged)
//     K = c_in x taps, walked in chunks of KC input channels that are double-buffered in LDS.
// Output D[c_out, pixel] has the pixel on the lane, so NCHW stores are 128-B contiguous per row.
//
// up = 1 kernel: v_mfma_f32_32x32x2_f32, NW waves, wave tile = (MB x 32 c_out) x (NBW x 32 pixels).
// up = 2 kernel: the stride-2 transposed convolution is evaluated as its 4 output phases
//     (even/odd row x even/odd col; 4+2+2+1 = 9 non-zero taps, so no multiply by stuffed zeros),
//     on a quad grid with a one-quad halo, v_mfma_f32_16x16x4_f32; the phase images y1 are then
//     passed through LDS to the fused 4x4 FIR ([1,3,3,1]x[1,3,3,1]/64 * 4, evaluated separably,
//     polyphase) + epilogue, so the (2H+1)^2 intermediate never reaches HBM.
//
// Staging: the activation halo tile is laid out so that LDS column u <-> image column X0 - 4 + u; every
// 16-byte group is then aligned in global memory AND in LDS (one global_load_dwordx4 + one
// ds_write_b128 per 4 pixels, style multiply in between).  Loads for chunk k+1 are issued before the
// MFMA loop of chunk k and consumed after it.  Weights are zero-padded on the host to
// [ceil8(c_in)][9][ceil32(c_out)] so their chunk is a pure linear copy done by global_load_lds_dwordx4
// (no VGPRs, no ds_write).  LDS fragment reads are software-pipelined one k-step ahead of the MFMAs.
#include "nb_common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ModconvParams {
    const float* x1;
    const float* x2;
    const float* wpk;      // [c_in_pad][9][c_out_ld]
    const float* styles;   // [n][c_in]
    const float* dcoefs;   // [n][c_out]
    const float* noise;    // [n or 1][Hout][Wout] or null
    const float* bias;     // [c_out]
    float* y;              // [n][c_out][Hout][Wout]
    const float* out_scale;// H2 output mode: [n][c_out] per-channel scale applied after the epilogue (the consumer's styles), or null
    const float* zeros;    // >= 16 bytes of zeros in device memory (LDS-DMA source for out-of-image halo groups)
    long long noise_stride_n;
    int c1, c2, c_in, c_out, c_out_ld;
    int dbg;               // developer ablation flags (env NB_DEBUG): 1 = no output stores, 2 = no LDS-DMA in the K loop
    int sty_floats;        // LDS floats reserved for the styles of one sample (c_in rounded up to the chunk grid)
    int h, w;              // input resolution
    int log2_tw;           // up1: tile width = 1 << log2_tw
    int th;                // up1: tile rows actually staged (<= h); up2: quad rows per tile
    int tw;                // up2: quad cols per tile
    int tiles_x, tiles_y, slices;
    float alpha, gain, clamp;
    unsigned long long* tstamps;   // debug: per-workgroup phase timestamps (nb_debug_set_timestamps_f32), or null
};

static unsigned long long* g_ts32 = nullptr;
static int g_ts32_cap = 0;
// Debug hook (not part of the product ABI): phase timestamps [workgroup][8] of the next fp32 launches, 100 MHz ticks
extern "C" void nb_debug_set_timestamps_f32(void* buf, int capacity_workgroups) { g_ts32 = (unsigned long long*)buf; g_ts32_cap = capacity_workgroups; }
#define NB_TS32(k)                                                                                           \
    do {                                                                                                     \
        if (p.tstamps && threadIdx.x == 0) p.tstamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

__device__ __forceinline__ float nb_epilogue(float v, float bias, float alpha, float gain, float clamp) {
    v += bias;
    v = v < 0.f ? v * alpha : v;
    v *= gain;
    if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    return v;
}

// LDS floats reserved for one weight chunk: whole 1-KiB LDS-DMA pieces (tail lanes of the last piece re-copy
// the last element, so the piece must lie inside the buffer)
__host__ __device__ constexpr int nb_wbuf_floats(int kc, int co_wg) { return ((kc * 9 * co_wg / 4 + 63) / 64) * 256; }

#define NB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define NB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ------------------------------------------------------------------------------------------------
// up = 1
// ------------------------------------------------------------------------------------------------
// LDS geometry of the up=1 kernel (shared by kernel and launcher): 16-byte groups of the largest halo tile
__host__ __device__ constexpr int nb_imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int nb_imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ constexpr int nb_up1_plane4(int pix_wg) {
    return nb_imax(nb_imax((pix_wg / 32 + 2) * 10, (nb_imin(pix_wg / 16, 16) + 2) * 6),
                   nb_imax((nb_imin(pix_wg / 8, 8) + 2) * 4, 6 * 3));
}
__host__ __device__ constexpr int nb_up1_ppc(int pix_wg) { return (nb_up1_plane4(pix_wg) + 63) / 64; }   // 1-KiB pieces per channel
__host__ __device__ constexpr int nb_up1_stage_floats(int pix_wg, int kc, int co_wg) {
    return kc * nb_up1_ppc(pix_wg) * 256 + nb_wbuf_floats(kc, co_wg);
}

// SK = split-K: all NW waves work on the SAME (MB x 32 c_out) x (NBW x 32 pixel) tile, wave w taking the k-pairs
// kk = w (mod NW) of every chunk; the partial accumulators are summed through LDS before the epilogue.  This is for
// the small layers / the batch-1 configuration, where the length of one wave's dependent MFMA chain (not the
// chip's MFMA rate) bounds the launch.
template <int NW, int MB, int NBW, int KC, int NST, bool SK>
__global__ __launch_bounds__(NW * 64) void modconv3x3_up1_kernel(const ModconvParams p) {
    constexpr int CO_WG = MB * 32;
    constexpr int PIX_WG = (SK ? 1 : NW) * NBW * 32;
    static_assert(!SK || (KC / 2) % NW == 0, "split-K needs a whole number of k-pairs per wave");
    constexpr int PPC = nb_up1_ppc(PIX_WG);
    constexpr int XCH = PPC * 256;                   // floats per channel of the halo tile (lanes 0-31 / 32-63 read k / k+1 in separate LDS passes)
    constexpr int NXP = KC * PPC;                    // activation pieces per chunk
    constexpr int NXPW = (NXP + NW - 1) / NW;
    constexpr int NWP = nb_wbuf_floats(KC, CO_WG) / 256;
    constexpr int NWPW = (NWP + NW - 1) / NW;
    constexpr int NPW = NXPW + NWPW;                 // LDS-DMA instructions per wave per chunk (same for every wave)
    constexpr int STAGE = nb_up1_stage_floats(PIX_WG, KC, CO_WG);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    NB_TS32(0);
    float* sty = smem;                // [sty_floats] styles of this sample (zero padded to the chunk grid)
    float* ring = smem + p.sty_floats;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lh = lane >> 5, l31 = lane & 31;
    const int H = p.h, W = p.w;
    const int TW = 1 << p.log2_tw, XS = TW + 8, G = XS >> 2;
    const int th = p.th;
    const int plane4 = (th + 2) * G;

    int b = blockIdx.x;
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x; const int tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int y0 = tile_y * th, x0 = tile_x * TW;
    const int co0 = slice * CO_WG;
    const int HW = H * W;

    for (int i = tid; i < p.sty_floats; i += NW * 64) sty[i] = i < p.c_in ? p.styles[(size_t)n * p.c_in + i] : 0.f;

    // ---- LDS-DMA descriptors of this wave's activation pieces (fixed for all chunks), see the up=2 kernel ----
    int xsp[NXPW], xk[NXPW], xdst[NXPW];
#pragma unroll
    for (int i = 0; i < NXPW; ++i) {
        int q = i * NW + wv;
        q = q < NXP ? q : NXP - 1;
        const int k = q / PPC, part = q - k * PPC;
        const int e4 = part * 64 + lane;
        xk[i] = k;
        xdst[i] = k * XCH + part * 256;
        xsp[i] = -1;
        if (e4 < plane4) {
            const int r = e4 / G, g = e4 - r * G;
            const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * g;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) xsp[i] = gy * W + gx;
        }
    }
    auto issue = [&](int ck, float* st) {
        const int c0 = ck * KC;
#pragma unroll
        for (int i = 0; i < NXPW; ++i) {
            const int ch = c0 + xk[i];
            const float* src = p.zeros;
            if (xsp[i] >= 0 && ch < p.c_in)
                src = (ch < p.c1 ? p.x1 + ((size_t)n * p.c1 + ch) * HW : p.x2 + ((size_t)n * p.c2 + (ch - p.c1)) * HW) + xsp[i];
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(st + xdst[i]), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NWPW; ++i) {
            int q = i * NW + wv;
            q = q < NWP ? q : NWP - 1;
            int e4 = q * 64 + lane;
            e4 = e4 < KC * 9 * CO_WG / 4 ? e4 : KC * 9 * CO_WG / 4 - 1;
            const int row = e4 / (CO_WG / 4), j4 = e4 - row * (CO_WG / 4);
            const float* src = p.wpk + ((size_t)c0 * 9 + row) * p.c_out_ld + co0 + j4 * 4;
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(st + KC * XCH + q * 256), 16, 0, 0);
        }
    };

    // ---- B-fragment base offsets (pixel -> halo tile position; +3: column u = tx + kx + 3) ----
    int boff[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int q = ((SK ? 0 : wv) * NBW + nb) * 32 + l31;
        int ty = q >> p.log2_tw;
        const int tx = q & (TW - 1);
        ty = ty < th ? ty : th - 1;       // rows past the tile are computed on clamped data and never stored
        boff[nb] = ty * XS + tx + 3;
    }

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    // split-K (the latency-bound launches): the epilogue's operands - demodulation / bias of the 4 c_out rows this wave
    // stores, the lane's noise value - are fetched now, so that their latency hides under the K loop
    float pre_d[MB][4], pre_b[MB][4], pre_nz[NBW];
    if constexpr (SK) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + mb * 32 + j + 8 * wv + 4 * lh;
                pre_d[mb][j] = co < p.c_out ? p.dcoefs[(size_t)n * p.c_out + co] : 0.f;
                pre_b[mb][j] = co < p.c_out ? p.bias[co] : 0.f;
            }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int q = nb * 32 + l31;
            const int ty = q >> p.log2_tw, tx = q & (TW - 1);
            const int oy = y0 + ty, ox = x0 + tx;
            pre_nz[nb] = (ty < th && oy < H && p.noise) ? p.noise[(size_t)n * p.noise_stride_n + (size_t)oy * W + ox] : 0.f;
        }
    }

    const int nchunks = (p.c_in + KC - 1) / KC;
    // NST-stage ring: chunks ck+1 .. ck+NST-1 are in flight while chunk ck is multiplied.  A chunk is complete
    // for this wave when all but the (chunks still allowed in flight) x NPW youngest LDS-DMA operations have retired
    // (counted vmcnt), and for the workgroup after the raw s_barrier that follows (__syncthreads() would drain the queue).
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nchunks) issue(st, ring + st * STAGE);
    {
        // wait for chunk 0: at most min(nchunks, NST-1) - 1 younger chunks may stay in flight
        const int younger = (nchunks < NST - 1 ? nchunks : NST - 1) - 1;
#pragma unroll
        for (int v = NST - 2; v >= 0; --v)
            if (younger == v) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(v * NPW) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    NB_TS32(1);
    int cur = 0;
    for (int ck = 0; ck < nchunks; ++ck) {
        if (ck + NST - 1 < nchunks && !(p.dbg & 2)) {
            const int nxt = cur == 0 ? NST - 1 : cur - 1;       // (cur + NST - 1) % NST
            issue(ck + NST - 1, ring + nxt * STAGE);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* xb = ring + cur * STAGE + lh * XCH;
        const float* wb = ring + cur * STAGE + KC * XCH + lh * 9 * CO_WG + l31;
        const float* sb = sty + ck * KC + lh;
        // software pipeline: the fragments of step s+1 are read from LDS right after the first MFMA of
        // step s has issued, so their latency rides under the remaining MB*NBW-1 MFMAs (64 cycles each).
        // The style modulates the weight fragments on their way from LDS to the MFMA (networks.py:59-60).
        constexpr int STEPS = (KC / 2) / (SK ? NW : 1) * 9;
        // PF = prefetch distance in steps (2 was measured for the single-MFMA-per-step split-K variant, as were two
        // alternating accumulators: no faster - tools/phase_times_f32.py: its K loop runs at ~55 ns per MFMA step either way)
        constexpr int PF = 1;
        float af[PF + 1][MB], bfr[PF + 1][NBW], sv[PF + 1];
        auto fetch = [&](int step, float (&a)[MB], float (&bb)[NBW], float& s) {
            const int kq = step / 9, tap = step - kq * 9;
            const int kk = SK ? kq * NW + wv : kq;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = wb[(kk * 2 * 9 + tap) * CO_WG + mb * 32];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) bb[nb] = xb[kk * 2 * XCH + boff[nb] + ky * XS + kx];
            s = sb[kk * 2];
        };
#pragma unroll
        for (int f = 0; f < PF; ++f)
            if (f < STEPS) fetch(f, af[f], bfr[f], sv[f]);
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int cur_f = step % (PF + 1), nxt_f = (step + PF) % (PF + 1);
            float a[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = af[cur_f][mb] * sv[cur_f];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bfr[cur_f][0], acc[0][0], 0, 0, 0);
            if (step + PF < STEPS) fetch(step + PF, af[nxt_f], bfr[nxt_f], sv[nxt_f]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    if (mb + nb > 0)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], bfr[cur_f][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (step + PF < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, MB + NBW + 1, 0);
            if (MB * NBW > 1) __builtin_amdgcn_sched_group_barrier(0x008, MB * NBW - 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            // chunk ck+1 must have landed: chunks ck+2 .. min(nchunks-1, ck+NST-1) may stay in flight
            const int last = nchunks - 1 < ck + NST - 1 ? nchunks - 1 : ck + NST - 1;
            const int younger = last - (ck + 1);
#pragma unroll
            for (int v = NST - 2; v >= 1; --v)
                if (younger == v) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(v * NPW) : "memory");
            if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        cur = cur == NST - 1 ? 0 : cur + 1;
    }

    NB_TS32(2);
    // ---- split-K: sum the NW partial accumulators through LDS; wave w then owns registers r with (r >> 2) == w ----
    if constexpr (SK) {
        static_assert(!SK || NW == 4, "split-K register ownership assumes 4 waves");
        float* red = ring;                                   // [NW][MB*NBW*16][64]; the ring is idle now
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wv * (MB * NBW * 16) + (mb * NBW + nb) * 16 + r) * 64 + lane] = acc[mb][nb][r];
        __syncthreads();
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sum = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) sum += red[(w2 * (MB * NBW * 16) + (mb * NBW + nb) * 16 + r) * 64 + lane];
                    acc[mb][nb][r] = sum;
                }
    }

    NB_TS32(3);
    // ---- epilogue: *d, +noise, +bias, lrelu, gain, clamp; D[row = c_out, col = pixel] ----
    const int Wo = W, Ho = H;
    const float* dco = p.dcoefs + (size_t)n * p.c_out;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int q = ((SK ? 0 : wv) * NBW + nb) * 32 + l31;
        const int ty = q >> p.log2_tw, tx = q & (TW - 1);
        const int oy = y0 + ty, ox = x0 + tx;
        const bool ok = ty < th && oy < Ho;
        float nz = 0.f;
        if constexpr (SK) nz = pre_nz[nb];
        else if (ok && p.noise) nz = p.noise[(size_t)n * p.noise_stride_n + (size_t)oy * Wo + ox];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (SK && (r >> 2) != wv) continue;          // split-K: each wave stores a quarter of the rows
                const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ok && co < p.c_out && !(p.dbg & 1)) {
                    float v = acc[mb][nb][r] * (SK ? pre_d[mb][r & 3] : dco[co]) + nz;
                    v = nb_epilogue(v, SK ? pre_b[mb][r & 3] : p.bias[co], p.alpha, p.gain, p.clamp);
                    p.y[((size_t)n * p.c_out + co) * ((size_t)Ho * Wo) + (size_t)oy * Wo + ox] = v;
                }
            }
        }
    }
    NB_TS32(4);
    if (p.tstamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); NB_TS32(5); }
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
// LDS geometry of the up=2 kernel (shared by kernel and launcher)
__host__ __device__ constexpr int nb_up2_ppc(int nbp) { return nbp <= 2 ? 1 : (nbp <= 6 ? 2 : 3); }   // 1-KiB pieces per channel
__host__ __device__ constexpr int nb_up2_xch(int nbp) { return nb_up2_ppc(nbp) * 256 + 16; }            // floats per channel (+16: k and k+1 on different banks)
__host__ __device__ constexpr int nb_up2_stage_floats(int nbp, int kc) { return kc * nb_up2_xch(nbp) + nb_wbuf_floats(kc, 16); }
#define NB_UP2_STAGES 3
#define NB_STY_MAX 1024

// H2OUT: write the output in the split-f16 "H2" activation format of nb_modconv_h3.hip ([N][C/8][hi,lo][H][W][8] f16,
// multiplied by out_scale = the consuming layer's styles) instead of fp32 NCHW.
template <int NBP, int KC, bool H2OUT>
__global__ __launch_bounds__(256) void modconv3x3_up2_kernel(const ModconvParams p) {
    constexpr int NW = 4;
    constexpr int CO_WG = 16;
    constexpr int PPC = nb_up2_ppc(NBP);
    constexpr int XCH = nb_up2_xch(NBP);
    constexpr int NXP = KC * PPC;                    // activation pieces per chunk
    constexpr int NXPW = (NXP + NW - 1) / NW;        // ... issued per wave
    constexpr int NWP = nb_wbuf_floats(KC, CO_WG) / 256;   // weight pieces per chunk
    constexpr int NWPW = (NWP + NW - 1) / NW;
    constexpr int NPW = NXPW + NWPW;                 // LDS-DMA instructions per wave per chunk (same for every wave)
    constexpr int STAGE = nb_up2_stage_floats(NBP, KC);
    constexpr int NPOS_MAX = 4 * NBP * 16;
    constexpr int Y1_PHASE = NPOS_MAX;               // floats per phase image
    constexpr int Y1_SLOT = 4 * Y1_PHASE + 16;       // +16: keep the 4 c_out slots on different banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sty = smem;                // [NB_STY_MAX] styles of this sample (zero padded)
    float* ring = smem + NB_STY_MAX;  // [3 stages][ x: KC channels x XCH | w: KC x 9 x 16 ]
    float* y1s = ring;                // epilogue reuse: [4 slots][4 phases][NPOS_MAX]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lq = lane >> 4, l15 = lane & 15;
    const int H = p.h, W = p.w;
    const int TQH = p.th, TQW = p.tw;
    const int PH = TQH + 2, PW = TQW + 2, NPOS = PH * PW;
    const int XS = TQW + 8, G = XS >> 2;
    const int plane4 = (TQH + 3) * G;                // 16-byte groups per channel of the halo tile

    int b = blockIdx.x;
    const int slice = b % p.slices; b /= p.slices;
    const int tile_x = b % p.tiles_x; const int tile_y = b / p.tiles_x;
    const int n = blockIdx.y;
    const int I0 = tile_y * TQH, J0 = tile_x * TQW;
    const int co0 = slice * CO_WG;
    const int HW = H * W;

    // styles of this sample -> LDS (the A fragments are modulated on their way from LDS to the MFMA)
    for (int i = tid; i < NB_STY_MAX; i += 256) sty[i] = i < p.c_in ? p.styles[(size_t)n * p.c_in + i] : 0.f;
    // small-tile variants (the latency-bound launches): the epilogue's operands - demodulation, bias, hand-off scale of
    // the 16 channels and the tile's noise - go to LDS now instead of being fetched in each of the 4 epilogue rounds
    constexpr bool PRE = NBP <= 3;
    constexpr int NZ_MAX = !PRE ? 1 : NBP == 3 ? 512 : NBP == 2 ? 256 : 64;
    __shared__ float s_pd[16], s_pb[16], s_psc[16], s_pnz[NZ_MAX];
    if constexpr (PRE) {
        if (tid < 16) {
            const int co = co0 + tid;
            const bool v = co < p.c_out;
            s_pd[tid] = v ? p.dcoefs[(size_t)n * p.c_out + co] : 0.f;
            s_pb[tid] = v ? p.bias[co] : 0.f;
            s_psc[tid] = v ? (p.out_scale ? p.out_scale[(size_t)n * p.c_out + co] : 1.f) : 0.f;
        }
        for (int e = tid; e < 4 * TQH * TQW; e += 256) {
            const int r = e / (2 * TQW), c = e - r * (2 * TQW);
            s_pnz[e] = p.noise ? p.noise[(size_t)n * p.noise_stride_n + (size_t)(2 * I0 + r) * (2 * W) + 2 * J0 + c] : 0.f;
        }
    }

    // ---- LDS-DMA descriptors of this wave's activation pieces (fixed for all chunks) ----
    // piece q = i*NW + wv  ->  channel k = q / PPC of the chunk, part = q % PPC; lane l copies the 16-byte group
    // e4 = part*64 + l of that channel's halo tile (row r = e4 / G, group g = e4 % G), or zeros outside the image.
    int xsp[NXPW], xk[NXPW], xdst[NXPW];
#pragma unroll
    for (int i = 0; i < NXPW; ++i) {
        int q = i * NW + wv;
        q = q < NXP ? q : NXP - 1;
        const int k = q / PPC, part = q - k * PPC;
        const int e4 = part * 64 + lane;
        xk[i] = k;
        xdst[i] = k * XCH + part * 256;
        xsp[i] = -1;
        if (e4 < plane4) {
            const int r = e4 / G, g = e4 - r * G;
            const int gy = I0 - 1 + r, gx = J0 - 4 + 4 * g;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) xsp[i] = gy * W + gx;
        }
    }
    auto issue = [&](int ck, float* st) {
        const int c0 = ck * KC;
#pragma unroll
        for (int i = 0; i < NXPW; ++i) {
            const int ch = c0 + xk[i];
            const float* src = p.zeros;
            if (xsp[i] >= 0 && ch < p.c_in)
                src = (ch < p.c1 ? p.x1 + ((size_t)n * p.c1 + ch) * HW : p.x2 + ((size_t)n * p.c2 + (ch - p.c1)) * HW) + xsp[i];
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(st + xdst[i]), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NWPW; ++i) {
            int q = i * NW + wv;
            q = q < NWP ? q : NWP - 1;
            int e4 = q * 64 + lane;
            e4 = e4 < KC * 9 * CO_WG / 4 ? e4 : KC * 9 * CO_WG / 4 - 1;
            const int row = e4 / (CO_WG / 4), j4 = e4 - row * (CO_WG / 4);
            const float* src = p.wpk + ((size_t)c0 * 9 + row) * p.c_out_ld + co0 + j4 * 4;
            __builtin_amdgcn_global_load_lds(NB_GLOBAL_PTR(src), NB_LDS_PTR(st + KC * XCH + q * 256), 16, 0, 0);
        }
    };

    // position blocks of this wave: block index wv*NBP + j, 16 positions each; X(r,c) sits at column c + 3
    int boff[NBP];
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        int pidx = (wv * NBP + j) * 16 + l15;
        pidx = pidx < NPOS ? pidx : NPOS - 1;
        const int r = pidx / PW, c = pidx - r * PW;
        boff[j] = r * XS + c + 3;
    }

    f32x4 acc[NBP][4];
#pragma unroll
    for (int j = 0; j < NBP; ++j)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) acc[j][ph] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = (p.c_in + KC - 1) / KC;
    // 3-stage ring: chunk ck+2 is in flight while chunk ck is multiplied.  A chunk is complete for this wave when
    // all but its NPW youngest LDS-DMA operations have retired (counted vmcnt), and for the workgroup after the
    // barrier that follows.  Raw s_barrier: __syncthreads() would drain the DMA queue (vmcnt(0)).
    issue(0, ring);
    if (nchunks > 1) issue(1, ring + STAGE);
    if (nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int ck = 0; ck < nchunks; ++ck) {
        if (ck + 2 < nchunks && !(p.dbg & 2)) {
            const int nxt = cur >= 1 ? cur - 1 : 2;       // (cur + 2) % 3
            issue(ck + 2, ring + nxt * STAGE);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* xb = ring + cur * STAGE + lq * XCH;
        const float* wb = ring + cur * STAGE + KC * XCH + lq * 9 * CO_WG + l15;
        const float* sb = sty + ck * KC + lq;
        // software pipeline over (k-step, position block): block j+1's four B fragments (and, at the last
        // block of a k-step, the next k-step's nine A fragments + style) are read right after block j's first MFMA
        constexpr int KS = KC / 4;
        float af[2][9], sv[2], xf[2][4];
        auto fetchA = [&](int ks, float (&a)[9], float& s) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) a[tap] = wb[(ks * 4 * 9 + tap) * CO_WG];
            s = sb[ks * 4];
        };
        auto fetchB = [&](int ks, int j, float (&x)[4]) {
            const float* xp = xb + ks * 4 * XCH + boff[j];
            x[3] = xp[0];            // X(r, c)
            x[2] = xp[1];            // X(r, c+1)
            x[1] = xp[XS];           // X(r+1, c)
            x[0] = xp[XS + 1];       // X(r+1, c+1)
        };
        fetchA(0, af[0], sv[0]);
        fetchB(0, 0, xf[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ca = ks & 1;
            float a[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) a[tap] = af[ca][tap] * sv[ca];     // weight modulation (networks.py:59-60)
#pragma unroll
            for (int j = 0; j < NBP; ++j) {
                const int it = ks * NBP + j;
                const int cb = it & 1;
                const float x00 = xf[cb][0], x01 = xf[cb][1], x10 = xf[cb][2], x11 = xf[cb][3];
                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], x00, acc[j][0], 0, 0, 0);
                int nreads = 0;
                if (j + 1 < NBP) { fetchB(ks, j + 1, xf[cb ^ 1]); nreads = 4; }
                else if (ks + 1 < KS) { fetchA(ks + 1, af[ca ^ 1], sv[ca ^ 1]); fetchB(ks + 1, 0, xf[cb ^ 1]); nreads = 14; }
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
                if (nreads == 14) __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // chunk ck+1 must have landed (this wave's share) before the barrier publishes it to the workgroup
        if (ck + 2 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
    }

    // ---- epilogue: 4 rounds of 4 c_out through LDS -> polyphase FIR -> bias/lrelu/clamp -> store ----
    // D layout of v_mfma_f32_16x16x4_f32: col = lane & 15 (position), row = 4*(lane>>4) + reg (c_out).
    // fp32 NCHW output: round g publishes accumulator register g of every lane (c_out rows {g, 4+g, 8+g, 12+g}).
    // H2 output:        round g publishes all 4 registers of lane group g (c_out rows 4g .. 4g+3, consecutive),
    //                   so that one thread owns 4 consecutive channels of a pixel = one 8-byte f16x4 store.
    const int Wo = 2 * W, Ho = 2 * H;
    const float* dco = p.dcoefs + (size_t)n * p.c_out;
    const int nquads = TQH * TQW;
    auto fir_quad = [&](const float* base, float (&out)[2][2]) {
        const float* ee = base;
        const float* eo = base + 1 * Y1_PHASE;
        const float* oe = base + 2 * Y1_PHASE;
        const float* oo = base + 3 * Y1_PHASE;
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
        out[0][0] = 0.25f * vo0[0] + 0.75f * ve0[0] + 0.75f * vo0[1] + 0.25f * ve0[1];
        out[0][1] = 0.25f * ve0[0] + 0.75f * vo0[1] + 0.75f * ve0[1] + 0.25f * vo0[2];
        out[1][0] = 0.25f * vo1[0] + 0.75f * ve1[0] + 0.75f * vo1[1] + 0.25f * ve1[1];
        out[1][1] = 0.25f * ve1[0] + 0.75f * vo1[1] + 0.75f * ve1[1] + 0.25f * vo1[2];
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (H2OUT) {
            if (lq == g) {
#pragma unroll
                for (int j = 0; j < NBP; ++j) {
                    const int pidx = (wv * NBP + j) * 16 + l15;
#pragma unroll
                    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y1s[r * Y1_SLOT + ph * Y1_PHASE + pidx] = acc[j][ph][r];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NBP; ++j) {
                const int pidx = (wv * NBP + j) * 16 + l15;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) y1s[lq * Y1_SLOT + ph * Y1_PHASE + pidx] = acc[j][ph][g];
            }
        }
        __syncthreads();
        if constexpr (H2OUT) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const int cb = co0 + 4 * g;                          // 4 consecutive channels cb .. cb+3
            const int c8o = (p.c_out + 7) >> 3;
            for (int qd = tid; qd < nquads; qd += 256) {
                const int ti = qd / TQW, tj = qd - ti * TQW;
                const int qi = I0 + ti, qj = J0 + tj;
                float o[4][2][2];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) fir_quad(y1s + s4 * Y1_SLOT + ti * PW + tj, o[s4]);
                if (qi < H && qj < W && cb < p.c_out && !(p.dbg & 1)) {
                    float d4[4], b4[4], sc4[4];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int co = cb + s4;
                        const bool v = co < p.c_out;
                        if constexpr (PRE) {
                            d4[s4] = s_pd[4 * g + s4]; b4[s4] = s_pb[4 * g + s4]; sc4[s4] = s_psc[4 * g + s4];
                        } else {
                            d4[s4] = v ? dco[co] : 0.f;
                            b4[s4] = v ? p.bias[co] : 0.f;
                            sc4[s4] = v ? (p.out_scale ? p.out_scale[(size_t)n * p.c_out + co] : 1.f) : 0.f;
                        }
                    }
                    _Float16* ob = reinterpret_cast<_Float16*>(p.y) + (((size_t)n * c8o + (cb >> 3)) * 2) * ((size_t)Ho * Wo * 8) + (cb & 7);
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const int oy = 2 * qi + dy, ox = 2 * qj + dx;
                            float nz = 0.f;
                            if constexpr (PRE) nz = s_pnz[(2 * ti + dy) * (2 * TQW) + 2 * tj + dx];
                            else if (p.noise) nz = p.noise[(size_t)n * p.noise_stride_n + (size_t)oy * Wo + ox];
                            h4 hi, lo;
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                float v = nb_epilogue(o[s4][dy][dx] * d4[s4] + nz, b4[s4], p.alpha, p.gain, p.clamp) * sc4[s4];
                                // (one value, materialised once: the hi and the lo half must be split from the SAME fp32 number --
                                //  built without SLP vectorisation the two uses were computed separately and disagreed in the last
                                //  bit on ~5e-5 of the elements, which costs a whole f16 ulp when the rounding of hi flips)
                                asm volatile("" : "+v"(v));
                                const _Float16 hh = (_Float16)v;
                                hi[s4] = hh;
                                lo[s4] = (_Float16)(v - (float)hh);
                            }
                            _Float16* op = ob + ((size_t)oy * Wo + ox) * 8;
                            *reinterpret_cast<h4*>(op) = hi;
                            *reinterpret_cast<h4*>(op + (size_t)Ho * Wo * 8) = lo;
                        }
                }
            }
        } else {
            // slot s <-> c_out row 4*s + g
            for (int it = tid; it < 4 * nquads; it += 256) {
                const int s = it / nquads, qd = it - s * nquads;
                const int ti = qd / TQW, tj = qd - ti * TQW;
                const int co = co0 + 4 * s + g;
                float out[2][2];
                fir_quad(y1s + s * Y1_SLOT + ti * PW + tj, out);
                const int qi = I0 + ti, qj = J0 + tj;
                if (qi < H && qj < W && co < p.c_out && !(p.dbg & 1)) {
                    const float d = PRE ? s_pd[4 * s + g] : dco[co], bs = PRE ? s_pb[4 * s + g] : p.bias[co];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        const int oy = 2 * qi + dy, ox = 2 * qj;
                        float n0 = 0.f, n1 = 0.f;
                        if constexpr (PRE) {
                            n0 = s_pnz[(2 * ti + dy) * (2 * TQW) + 2 * tj]; n1 = s_pnz[(2 * ti + dy) * (2 * TQW) + 2 * tj + 1];
                        } else if (p.noise) {
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
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int NW, int MB, int NBW, int KC, int NST, bool SK = false>
static int launch_up1(ModconvParams p, int n, hipStream_t st) {
    constexpr int PIX_WG = (SK ? 1 : NW) * NBW * 32;
    const int TW = p.w < 32 ? p.w : 32;
    int l2 = 0; while ((1 << l2) < TW) ++l2;
    p.log2_tw = l2;
    int th = PIX_WG / TW; if (th > p.h) th = p.h;
    p.th = th;
    if ((th + 2) * ((TW + 8) / 4) > nb_up1_ppc(PIX_WG) * 64) { nb_set_error("modconv up1: tile %dx%d does not fit the LDS plane", th, TW); return NB_EINVAL; }
    p.tiles_x = p.w / TW; p.tiles_y = nb_cdiv(p.h, th); p.slices = nb_cdiv(p.c_out, MB * 32);
    p.sty_floats = (p.c_in + 31) / 32 * 32;
    size_t ring_floats = (size_t)NST * nb_up1_stage_floats(PIX_WG, KC, MB * 32);
    if (SK && ring_floats < (size_t)NW * MB * NBW * 16 * 64) ring_floats = (size_t)NW * MB * NBW * 16 * 64;
    const size_t lds = (p.sty_floats + ring_floats) * sizeof(float);
    if (lds > 160 * 1024) { nb_set_error("modconv up1: c_in=%d needs %zu bytes of LDS", p.c_in, lds); return NB_EINVAL; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_up1_kernel<NW, MB, NBW, KC, NST, SK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(p.tiles_x * p.tiles_y * p.slices, n);
    hipLaunchKernelGGL((modconv3x3_up1_kernel<NW, MB, NBW, KC, NST, SK>), grid, dim3(NW * 64), lds, st, p);
    NB_CHECK_LAUNCH("modconv3x3_up1");
    return NB_OK;
}

template <int NBP, int KC, bool H2OUT = false>
static int launch_up2(ModconvParams p, int n, hipStream_t st) {
    p.slices = nb_cdiv(p.c_out, 16);
    if (p.c_in > NB_STY_MAX) { nb_set_error("modconv up2: c_in=%d exceeds %d", p.c_in, NB_STY_MAX); return NB_EINVAL; }
    const size_t lds_main = (size_t)(NB_UP2_STAGES * nb_up2_stage_floats(NBP, KC)) * sizeof(float);
    const size_t lds_epi = (size_t)(4 * (4 * 4 * NBP * 16 + 16)) * sizeof(float);
    const size_t lds = NB_STY_MAX * sizeof(float) + (lds_main > lds_epi ? lds_main : lds_epi);
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_up2_kernel<NBP, KC, H2OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(p.tiles_x * p.tiles_y * p.slices, n);
    hipLaunchKernelGGL((modconv3x3_up2_kernel<NBP, KC, H2OUT>), grid, dim3(256), lds, st, p);
    NB_CHECK_LAUNCH("modconv3x3_up2");
    return NB_OK;
}

// Kernel variant selection.  Preference order = efficiency order; a less efficient (smaller-tile) variant is
// taken when the preferred one would leave the chip under-filled (< ~1.5 workgroups per CU), which is what
// matters for the small layers and for the batch-1 interactive configuration.
static const char* const kVariantNames[] = {
    "modconv3x3_up1_kernel<8, 4, 1, 4, 3, false>", "modconv3x3_up1_kernel<8, 2, 2, 4, 3, false>", "modconv3x3_up1_kernel<4, 2, 2, 8, 4, false>",
    "modconv3x3_up1_kernel<4, 1, 2, 8, 4, false>", "modconv3x3_up1_kernel<4, 1, 1, 8, 4, false>",
    "modconv3x3_up2_kernel<10, 4>", "modconv3x3_up2_kernel<6, 4>", "modconv3x3_up2_kernel<3, 8>",
    "modconv3x3_up2_kernel<2, 8>", "modconv3x3_up2_kernel<1, 8>", "modconv3x3_up1_kernel<4, 1, 1, 8, 4, true>"};
static const int kUp2Tiles[5][2] = {{16, 32}, {16, 16}, {8, 16}, {8, 8}, {4, 4}};      // quad rows x cols per workgroup

static int nb_select_variant(int n, int h, int w, int c_out, int up, int* tq = nullptr) {
    const long target = 384;
    if (up == 1) {
        static const int cand[6][2] = {{128, 256}, {64, 512}, {64, 256}, {32, 256}, {32, 128}, {32, 32}};   // c_out x pixels per workgroup (last: split-K)
        const long hw = (long)h * w;
        if (hw <= 32) return 10;       // 4x4 images: one 32-pixel block per sample, latency-bound -> split K over the waves
        int best = 4; long best_wgs = -1;
        for (int i = 0; i < 6; ++i) {
            const int co_wg = cand[i][0], pix = cand[i][1];
            if (i == 0 && c_out <= 64) continue;
            if (i == 1 && (c_out <= 32 || c_out > 64)) continue;
            if (i == 2 && c_out <= 32) continue;
            const long wgs = (long)n * ((hw + pix - 1) / pix) * ((c_out + co_wg - 1) / co_wg);
            if (wgs >= target) return i == 5 ? 10 : i;
            if (wgs > best_wgs) { best_wgs = wgs; best = i == 5 ? 10 : i; }
        }
        return best;
    }
    int best = 9; long best_wgs = -1;
    for (int i = 0; i < 5; ++i) {
        const int tqh = kUp2Tiles[i][0] < h ? kUp2Tiles[i][0] : h, tqw = kUp2Tiles[i][1] < w ? kUp2Tiles[i][1] : w;
        const int nbp = ((tqh + 2) * (tqw + 2) + 63) / 64;          // 16-position blocks per wave (4 waves)
        const int v = nbp > 6 ? 5 : nbp > 3 ? 6 : nbp > 2 ? 7 : nbp > 1 ? 8 : 9;
        const long wgs = (long)n * (h / tqh) * (w / tqw) * ((c_out + 15) / 16);
        if (tq) { tq[0] = tqh; tq[1] = tqw; }
        if (wgs >= target) return v;
        if (wgs > best_wgs) { best_wgs = wgs; best = v; }
    }
    if (tq) {      // nothing reached the target: the last (smallest) candidate has the most workgroups
        tq[0] = kUp2Tiles[4][0] < h ? kUp2Tiles[4][0] : h; tq[1] = kUp2Tiles[4][1] < w ? kUp2Tiles[4][1] : w;
    }
    return best;
}

// Lazily created per-device page of zeros (256 B): the LDS-DMA source for halo groups outside the image.
// Created on the first call for a device (outside any stream capture: run one warm-up call before capturing).
extern "C" const float* nb_zero_page_ptr(void) {
    static float* pages[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!pages[dev]) {
        float* ptr = nullptr;
        if (hipMalloc((void**)&ptr, 256) != hipSuccess) return nullptr;
        if (hipMemset(ptr, 0, 256) != hipSuccess) return nullptr;
        pages[dev] = ptr;
    }
    return pages[dev];
}

static int nb_modconv3x3_impl(const float* x1, int c1, const float* x2, int c2, const float* wpk,
                             const float* styles, const float* dcoefs, const float* noise,
                             int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w,
                             int c_out, int up, float alpha, float gain, float clamp, void* stream,
                             bool h2out, const float* out_scale) {
    NB_REQUIRE(x1 && wpk && styles && dcoefs && bias && y, "modconv3x3: null pointer");
    NB_REQUIRE(c1 > 0 && c2 >= 0 && (c2 == 0 || x2), "modconv3x3: bad channel split c1=%d c2=%d", c1, c2);
    NB_REQUIRE(n > 0 && n <= 65535, "modconv3x3: batch %d out of range", n);
    NB_REQUIRE(h >= 4 && w >= 4 && (w & (w - 1)) == 0 && (h & (h - 1)) == 0, "modconv3x3: h,w must be powers of two >= 4 (got %dx%d)", h, w);
    NB_REQUIRE(c_out > 0, "modconv3x3: c_out=%d", c_out);
    NB_REQUIRE(up == 1 || up == 2, "modconv3x3: up=%d unsupported", up);
    NB_REQUIRE(((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)wpk | (uintptr_t)y) % 16 == 0, "modconv3x3: x1, x2, wpk and y must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    ModconvParams p;
    p.x1 = x1; p.x2 = x2; p.wpk = wpk; p.styles = styles; p.dcoefs = dcoefs; p.noise = noise; p.bias = bias; p.y = y;
    p.noise_stride_n = noise_stride_n;
    p.out_scale = out_scale;
    p.zeros = nb_zero_page_ptr();
    NB_REQUIRE(p.zeros, "modconv3x3: could not allocate the zero page");
    p.c1 = c1; p.c2 = c2; p.c_in = c1 + c2; p.c_out = c_out; p.c_out_ld = (c_out + 31) / 32 * 32; p.h = h; p.w = w;
    p.dbg = g_nb_debug_flags;            // (developer hook: nb_debug_set_flags)
    p.sty_floats = 0; p.log2_tw = 0; p.th = 0; p.tw = 0; p.tiles_x = p.tiles_y = p.slices = 1;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.tstamps = g_ts32;
    int tq[2] = {0, 0};
    const int v = nb_select_variant(n, h, w, c_out, up, tq);
    switch (v) {
        case 0: return launch_up1<8, 4, 1, 4, 3>(p, n, st);
        case 1: return launch_up1<8, 2, 2, 4, 3>(p, n, st);
        case 2: return launch_up1<4, 2, 2, 8, 4>(p, n, st);
        case 3: return launch_up1<4, 1, 2, 8, 4>(p, n, st);
        case 4: return launch_up1<4, 1, 1, 8, 4>(p, n, st);
        case 10: return launch_up1<4, 1, 1, 8, 4, true>(p, n, st);    // (ring depths 6 / 8 and chunks of 16 / 32 channels measured at batch 1: no faster)
        default: break;
    }
    p.th = tq[0]; p.tw = tq[1];
    p.tiles_x = w / p.tw; p.tiles_y = h / p.th;
    if (h2out) {
        switch (v) {
            case 5: return launch_up2<10, 4, true>(p, n, st);
            case 6: return launch_up2<6, 4, true>(p, n, st);
            case 7: return launch_up2<3, 8, true>(p, n, st);
            case 8: return launch_up2<2, 8, true>(p, n, st);
            default: return launch_up2<1, 8, true>(p, n, st);
        }
    }
    switch (v) {
        case 5: return launch_up2<10, 4>(p, n, st);
        case 6: return launch_up2<6, 4>(p, n, st);
        case 7: return launch_up2<3, 8>(p, n, st);
        case 8: return launch_up2<2, 8>(p, n, st);
        default: return launch_up2<1, 8>(p, n, st);
    }
}

extern "C" int nb_modconv3x3_f32(const float* x1, int c1, const float* x2, int c2, const float* wpk,
                                 const float* styles, const float* dcoefs, const float* noise,
                                 int64_t noise_stride_n, const float* bias, float* y, int n, int h, int w,
                                 int c_out, int up, float alpha, float gain, float clamp, void* stream) {
    return nb_modconv3x3_impl(x1, c1, x2, c2, wpk, styles, dcoefs, noise, noise_stride_n, bias, y, n, h, w, c_out, up,
                              alpha, gain, clamp, stream, false, nullptr);
}

extern "C" int nb_modconv3x3_up2_f32_h2(const float* x1, int c1, const float* x2, int c2, const float* wpk,
                                        const float* styles, const float* dcoefs, const float* noise,
                                        int64_t noise_stride_n, const float* bias, const float* out_scale, void* y_h2,
                                        int n, int h, int w, int c_out, float alpha, float gain, float clamp,
                                        void* stream) {
    return nb_modconv3x3_impl(x1, c1, x2, c2, wpk, styles, dcoefs, noise, noise_stride_n, bias, (float*)y_h2, n, h, w,
                              c_out, 2, alpha, gain, clamp, stream, true, out_scale);
}

extern "C" int nb_modconv3x3_variant(int n, int h, int w, int c_out, int up, char* buf, int buflen) {
    NB_REQUIRE(buf && buflen > 0, "modconv3x3_variant: bad buffer");
    NB_REQUIRE(n > 0 && h >= 4 && w >= 4 && c_out > 0 && (up == 1 || up == 2), "modconv3x3_variant: bad shape");
    snprintf(buf, buflen, "%s", kVariantNames[nb_select_variant(n, h, w, c_out, up)]);
    return NB_OK;
}

// Host-side repack: W[c_out,c_in,3,3] -> wpk[ceil8(c_in)][9][ceil32(c_out)] (zero padded), wsq[c_in][c_out] = sum_k W^2
extern "C" int nb_pack_conv_weight(const float* w, int c_out, int c_in, float* wpk, float* wsq) {
    NB_REQUIRE(w && c_out > 0 && c_in > 0, "pack_conv_weight: bad arguments");
    const int ci_pad = (c_in + 7) / 8 * 8, co_ld = (c_out + 31) / 32 * 32;
    if (wpk) for (size_t i = 0; i < (size_t)ci_pad * 9 * co_ld; ++i) wpk[i] = 0.f;
    for (int o = 0; o < c_out; ++o)
        for (int i = 0; i < c_in; ++i) {
            float sq = 0.f;
            for (int t = 0; t < 9; ++t) {
                const float v = w[((size_t)o * c_in + i) * 9 + t];
                sq += v * v;
                if (wpk) wpk[((size_t)i * 9 + t) * co_ld + o] = v;
            }
            if (wsq) wsq[(size_t)i * c_out + o] = sq;
        }
    return NB_OK;
}
