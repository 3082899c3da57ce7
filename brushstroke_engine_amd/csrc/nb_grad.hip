// Gradient building blocks of the modulated convolution for gfx950 (second slice): what
// `conv2d_gradfix` (torch_utils/ops/conv2d_gradfix.py:107-168) obtains from cuDNN in the reference -
//   * the gradient w.r.t. the input of a strided convolution = another convolution (nb_conv2d_f32: generic stride /
//     padding / kernel size; the up = 1 layers reuse the fused forward kernel with swapped roles instead), and
//   * the gradient w.r.t. the weights = a correlation of two feature maps over all pixels (nb_conv2d_wgrad_f32),
// both exact fp32 on v_mfma_f32_32x32x2_f32 as implicit GEMMs whose operands are read straight from global memory
// (L1/L2 serve the reuse).  These are FIRST, correct versions: no LDS tiling yet - the training path is not what this
// round's performance work is about.
#include "nb_common.h"
#include <cstdlib>

static int g_wgrad_wgs = 256;
// developer hook (tools/bench_wgrad.py): workgroups the weight-gradient launches aim for (256 = one per CU; <= 0 restores it)
extern "C" void nb_debug_set_wgrad_wgs(int wgs) { g_wgrad_wgs = wgs > 0 ? wgs : 256; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// y[n,co,oy,ox] = out_scale[n,co] * sum_{ci,a,b} (x[n,ci,oy*st+a-pad,ox*st+b-pad] * in_scale[n,ci]) * w[co,ci,a,b]
// (cross-correlation, zero padding; scales optional).  Workgroup = 4 waves, wave tile = 32 c_out x 32 output pixels.
// ------------------------------------------------------------------------------------------------
struct ConvParams {
    const float* x; const float* w; const float* in_scale; const float* out_scale; float* y;
    int n, ci, h, wd, co, kh, kw, stride, pad, ho, wo;
};

// KS = compile-time kernel size (1 or 3: fully unrolled taps, so a channel pair's 2 x KS^2 operand loads are all in flight
// before its MFMAs), 0 = run-time kh x kw.
template <int KS>
__global__ __launch_bounds__(256) void conv2d_f32_kernel(const ConvParams p) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, lk = lane >> 5;
    const int n = blockIdx.z, co0 = blockIdx.y * 32;
    const int pix = (blockIdx.x * 4 + wv) * 32 + l31;
    const int npix = p.ho * p.wo;
    const bool pvalid = pix < npix;
    const int oy = pvalid ? pix / p.wo : 0, ox = pvalid ? pix - oy * p.wo : 0;
    const int co = co0 + l31;
    const bool cvalid = co < p.co;
    const int kh = KS ? KS : p.kh, kw = KS ? KS : p.kw;
    const int ktaps = kh * kw;
    const float* xn = p.x + (size_t)n * p.ci * p.h * p.wd;
    const float* wr = p.w + (size_t)(cvalid ? co : 0) * p.ci * ktaps;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int c0 = 0; c0 < p.ci; c0 += 2) {
        const int c = c0 + lk;
        const bool chv = c < p.ci;
        const float sc = (chv && p.in_scale) ? p.in_scale[(size_t)n * p.ci + c] : 1.f;
        const float* xc = xn + (size_t)(chv ? c : 0) * p.h * p.wd;
        const float* wc = wr + (size_t)(chv ? c : 0) * ktaps;
        if constexpr (KS != 0) {
            float av[KS * KS], bv[KS * KS];
#pragma unroll
            for (int a = 0; a < KS; ++a)
#pragma unroll
                for (int b = 0; b < KS; ++b) {
                    const int iy = oy * p.stride + a - p.pad, ix = ox * p.stride + b - p.pad;
                    const bool ok = chv && pvalid && iy >= 0 && iy < p.h && ix >= 0 && ix < p.wd;
                    bv[a * KS + b] = ok ? xc[(size_t)iy * p.wd + ix] : 0.f;
                    av[a * KS + b] = (chv && cvalid) ? wc[a * KS + b] : 0.f;
                }
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t] * sc, acc, 0, 0, 0);
        } else {
            for (int a = 0; a < kh; ++a) {
                const int iy = oy * p.stride + a - p.pad;
                for (int b = 0; b < kw; ++b) {
                    const int ix = ox * p.stride + b - p.pad;
                    float bv = 0.f, av = 0.f;
                    if (chv && pvalid && iy >= 0 && iy < p.h && ix >= 0 && ix < p.wd) bv = xc[(size_t)iy * p.wd + ix] * sc;
                    if (chv && cvalid) av = wc[a * kw + b];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                }
            }
        }
    }
    if (!pvalid) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = co0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (o < p.co) {
            const float s = p.out_scale ? p.out_scale[(size_t)n * p.co + o] : 1.f;
            p.y[((size_t)n * p.co + o) * npix + pix] = acc[r] * s;
        }
    }
}

extern "C" int nb_conv2d_f32(const float* x, const float* w, const float* in_scale, const float* out_scale, float* y, int n,
                             int c_in, int h, int wd, int c_out, int kh, int kw, int stride, int pad, void* stream) {
    NB_REQUIRE(x && w && y, "conv2d: null pointer");
    NB_REQUIRE(n >= 1 && n <= 65535 && c_in >= 1 && c_out >= 1 && h >= 1 && wd >= 1, "conv2d: bad sizes");
    NB_REQUIRE(kh >= 1 && kw >= 1 && kh <= 7 && kw <= 7 && stride >= 1 && pad >= 0, "conv2d: kernel 1..7, stride >= 1, pad >= 0");
    ConvParams p{x, w, in_scale, out_scale, y, n, c_in, h, wd, c_out, kh, kw, stride, pad, 0, 0};
    p.ho = (h + 2 * pad - kh) / stride + 1; p.wo = (wd + 2 * pad - kw) / stride + 1;
    NB_REQUIRE(p.ho >= 1 && p.wo >= 1, "conv2d: empty output");
    dim3 grid(nb_cdiv(p.ho * p.wo, 128), nb_cdiv(c_out, 32), n);
    NB_REQUIRE(grid.y <= 65535, "conv2d: too many output channels");
    if (kh == 3 && kw == 3) hipLaunchKernelGGL(conv2d_f32_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (kh == 1 && kw == 1) hipLaunchKernelGGL(conv2d_f32_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv2d_f32_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("conv2d");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient correlation:  A[n,cu,cv,a,b] = sum_{i,j} U[n,cu, i*st+a-pad, j*st+b-pad] * V[n,cv,i,j]
// (a, b in [0,3); zero outside U); the row slices of a launch add their partial sums atomically into A (zeroed first).
//   up = 1 layers: U = modulated input (pad 1, st 1), V = dL/d(conv output)  ->  A[n, c_in, c_out]
//   up = 2 layers: U = FIR-adjoint of dL/dy on the (2H+1)^2 grid (pad 0, st 2), V = modulated input -> A[n, c_out, c_in]
// ------------------------------------------------------------------------------------------------
struct WgradParams {
    const float* u; const float* v; float* a;
    int n, cu, hu, wu, cv, hv, wv, stride, pad, rows_per_wg, nslices;
    float* part;            // split-f16 kernel: partial results [nslices][n][cu][cv][9] written with plain stores (NULL: atomics into a)
    int scales_are_absmax;  // split-f16 kernel: `scales` holds max|u|, max|v| (nb_absmax_f32 slots), not the two powers of two
};

// The power of two that brings a tensor whose largest magnitude is mx near `target` (every thread of every kernel that derives
// a range scale from the same absmax slot computes the same value: the scale only has to be a power of two, not a particular one).
__device__ __forceinline__ float nb_pow2_scale(float target, float mx) {
    mx = fmaxf(mx, 1e-30f);
    float e = floorf(log2f(target / mx));
    e = fminf(fmaxf(e, -100.f), 100.f);
    return ldexpf(1.f, (int)e);
}

// Workgroup = 4 waves = 32 (cu) x 128 (cv) of all 9 taps (wave w owns cv columns 32w .. 32w+31, so no cross-wave
// reduction), for one sample and a slice of V's rows.  Per V row and 64-pixel column chunk the operands are staged in
// LDS with coalesced loads (lanes along pixels): V[128 cv][64] and the three U rows the taps need, U[3][32 cu][64 st + 2]
// (zeros outside the image), odd row pitches so that the channel-on-the-lane fragment reads are conflict free; then per
// pixel pair 1 + 9 single-dword LDS reads feed 9 v_mfma_f32_32x32x2_f32.  Row slices add their partial sums atomically.
#define NB_WG_CW 64
__global__ __launch_bounds__(256) void conv2d_wgrad_f32_kernel(const WgradParams p) {
    extern __shared__ float smem_wg[];
    const int st = p.stride;
    const int UW = NB_WG_CW * st + 3;                       // staged U columns per row (odd pitch)
    constexpr int VP = NB_WG_CW + 1;                        // V row pitch (odd)
    float* sv = smem_wg;                                    // [128][VP]
    float* su = smem_wg + 128 * VP;                         // [3][32][UW]
    const int tid = threadIdx.x, lane = tid & 63, wvid = tid >> 6, l31 = lane & 31, lk = lane >> 5;
    const int cu0 = blockIdx.x * 32, cv0 = blockIdx.y * 128;
    const int n = blockIdx.z / p.nslices, sl = blockIdx.z - n * p.nslices;
    const float* un = p.u + (size_t)n * p.cu * p.hu * p.wu;
    const float* vn = p.v + (size_t)n * p.cv * p.hv * p.wv;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool wave_active = cv0 + wvid * 32 < p.cv;
    const int i1 = min(p.hv, (sl + 1) * p.rows_per_wg);
    for (int i = sl * p.rows_per_wg; i < i1; ++i) {
        for (int jc = 0; jc < p.wv; jc += NB_WG_CW) {
            __syncthreads();                                 // previous chunk's fragment reads are done
            for (int idx = tid; idx < 128 * NB_WG_CW; idx += 256) {
                const int c = idx / NB_WG_CW, j = idx - c * NB_WG_CW;
                float val = 0.f;
                if (cv0 + c < p.cv && jc + j < p.wv) val = vn[((size_t)(cv0 + c) * p.hv + i) * p.wv + jc + j];
                sv[c * VP + j] = val;
            }
            const int ucols = NB_WG_CW * st + 2;
            for (int idx = tid; idx < 3 * 32 * ucols; idx += 256) {
                const int a = idx / (32 * ucols), rem = idx - a * (32 * ucols);
                const int c = rem / ucols, t = rem - c * ucols;
                const int y = i * st + a - p.pad, x = jc * st - p.pad + t;
                float val = 0.f;
                if (cu0 + c < p.cu && y >= 0 && y < p.hu && x >= 0 && x < p.wu) val = un[((size_t)(cu0 + c) * p.hu + y) * p.wu + x];
                su[(a * 32 + c) * UW + t] = val;
            }
            __syncthreads();
            if (wave_active) {
                const float* svw = sv + (wvid * 32 + l31) * VP + lk;
                const float* suw = su + l31 * UW + lk * st;
#pragma unroll 4
                for (int j0 = 0; j0 < NB_WG_CW; j0 += 2) {
                    const float bv = svw[j0];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b)
                            acc[a * 3 + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(suw[a * 32 * UW + j0 * st + b], bv, acc[a * 3 + b], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active) return;
    const int cvl = cv0 + wvid * 32 + l31;
    if (cvl >= p.cv) return;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = cu0 + (r & 3) + 8 * (r >> 2) + 4 * lk;      // cu row of the D tile; this lane's column = cv
            if (m < p.cu) atomicAdd(p.a + (((size_t)n * p.cu + m) * p.cv + cvl) * 9 + t, acc[t][r]);
        }
}

extern "C" int nb_conv2d_wgrad_f32(const float* u, const float* v, float* a, int n, int cu, int hu, int wu, int cv, int hv,
                                   int wv, int stride, int pad, void* stream) {
    NB_REQUIRE(u && v && a, "conv2d_wgrad: null pointer");
    NB_REQUIRE(n >= 1 && cu >= 1 && cv >= 1 && hu >= 1 && wu >= 1 && hv >= 1 && wv >= 1 && (stride == 1 || stride == 2) && pad >= 0,
               "conv2d_wgrad: bad sizes (stride 1 or 2)");
    WgradParams p{u, v, a, n, cu, hu, wu, cv, hv, wv, stride, pad, 0, 0, nullptr, 0};
    // enough workgroups to fill the chip: slice V's rows when there are few (n, tile) combinations
    const long tiles = (long)n * nb_cdiv(cu, 32) * nb_cdiv(cv, 128);
    const int wg_target = g_wgrad_wgs;     // 256 = one workgroup per CU: more row slices only add atomic traffic (tools/bench_wgrad.py)
    int slices = (int)((wg_target + tiles - 1) / tiles);
    if (slices > hv) slices = hv;
    if (slices < 1) slices = 1;
    p.rows_per_wg = nb_cdiv(hv, slices);
    p.nslices = nb_cdiv(hv, p.rows_per_wg);
    NB_REQUIRE((long)n * p.nslices <= 65535 && nb_cdiv(cv, 128) <= 65535, "conv2d_wgrad: grid too large");
    const size_t lds = (size_t)(128 * (NB_WG_CW + 1) + 3 * 32 * (NB_WG_CW * stride + 3)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv2d_wgrad_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    (void)hipMemsetAsync(a, 0, (size_t)n * cu * cv * 9 * sizeof(float), (hipStream_t)stream);
    dim3 grid(nb_cdiv(cu, 32), nb_cdiv(cv, 128), n * p.nslices);
    hipLaunchKernelGGL(conv2d_wgrad_f32_kernel, grid, dim3(256), lds, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("conv2d_wgrad");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// The same correlation on the f16 matrix cores with the split products of the inference kernels (hi/lo f16 halves of both
// operands, x y ~= xh yh + xh yl + xl yh, fp32 accumulate: ~2^-22 relative) - 27 v_mfma_f32_32x32x16_f16 per 16 pixels
// instead of 72 fp32 MFMAs.  Gradients can be far below the f16 range, so both operands are multiplied by a power of two
// (scales[0] for U, scales[1] for V, computed by the caller from the tensors' max-abs so that the largest value sits
// near 2^10) while they are split on their way into LDS, and the result is divided by the product.
// LDS: V hi/lo [128][72] f16 and, per tap (a, b), U hi/lo [32][72] f16 holding U[row a][(j)*stride + b - pad] - one
// pre-shifted (and, for stride 2, decimated) copy per tap, so that every MFMA fragment is one aligned ds_read_b128.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h8g __attribute__((ext_vector_type(8)));
typedef _Float16 h2g __attribute__((ext_vector_type(2)));
#define NB_WH_P 72             // row pitch in halves (144 B: 16-byte aligned rows, conflict-free b128 reads over 8 rows)

#ifndef NB_WHC
#define NB_WHC 32             // V pixels per column chunk of the split-f16 kernel
#endif
#define NB_WHP (NB_WHC + 8)    // LDS row pitch in halves: 16-byte aligned rows, conflict-free b128 reads over 8 rows
// Pipeline: column chunks of NB_WHC V-pixels outermost, V rows inside.  Moving one V row down needs only `stride` new U rows, so
// U lives in a 4-slot ring of rows (slot = row & 3), each with its three pre-shifted copies; while the MFMAs of row i run,
// the fp32 values of row i+1 (V row + the new U rows) are already on their way into registers, and are split / written to
// LDS after the MFMAs - global latency hides under the matrix work even with one workgroup per CU.
__global__ __launch_bounds__(256) void conv2d_wgrad_h3_kernel(const WgradParams p, const float* __restrict__ scales) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wh[];
    _Float16* svh = reinterpret_cast<_Float16*>(smem_wh);                 // [128][P]
    _Float16* svl = svh + 128 * NB_WHP;
    _Float16* suh = svl + 128 * NB_WHP;                                  // [4 ring rows][3 shifts][32][P]
    _Float16* sul = suh + 12 * 32 * NB_WHP;
    const int st = p.stride;
    const int tid = threadIdx.x, lane = tid & 63, wvid = tid >> 6, l31 = lane & 31, lk = lane >> 5;
    const int cu0 = blockIdx.x * 32, cv0 = blockIdx.y * 128;
    const int n = blockIdx.z / p.nslices, sl = blockIdx.z - n * p.nslices;
    const float* un = p.u + (size_t)n * p.cu * p.hu * p.wu;
    const float* vn = p.v + (size_t)n * p.cv * p.hv * p.wv;
    const float scu = p.scales_are_absmax ? nb_pow2_scale(1024.f, scales[0]) : scales[0];
    const float scv = p.scales_are_absmax ? nb_pow2_scale(1024.f, scales[1]) : scales[1];
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool wave_active = cv0 + wvid * 32 < p.cv;
    constexpr int HP = NB_WHC / 2;                                      // pixel pairs per row chunk
    constexpr int VPT = 128 * HP / 256, UPT = 3 * 32 * HP / 256;          // pairs per thread: V row chunk (16), one U row x 3 shifts (12)
    auto split2 = [](float a, float b, h2g& hi, h2g& lo) {
        const _Float16 ah = (_Float16)a, bh = (_Float16)b;
        hi[0] = ah; hi[1] = bh;
        lo[0] = (_Float16)(a - (float)ah); lo[1] = (_Float16)(b - (float)bh);
    };
    // Operand fetch without a branch per element: which of a thread's pixel pairs exist (channel tail, image border, padding
    // columns) depends only on the column chunk, so per chunk every pair gets an always-valid element offset and two mask bits;
    // the row loop then issues plain loads from (row base + offset) and selects zeros.  (The bounds-checked form compiled to 336
    // branches per row and was bound by instruction issue: ~6 us per 32-pixel row chunk for 0.8 us of matrix work.)
    int voff[VPT], uoff0[UPT], uoff1[UPT];
    unsigned vmask = 0, umask = 0;                                       // bits 2k / 2k+1: first / second pixel of pair k exists
    auto setup_chunk = [&](int jc) {
        vmask = 0; umask = 0;
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const int idx = k * 256 + tid, c = idx / HP, j = 2 * (idx - c * HP);
            const bool okc = cv0 + c < p.cv, b0 = okc && jc + j < p.wv, b1 = okc && jc + j + 1 < p.wv;
            voff[k] = b0 ? (cv0 + c) * p.hv * p.wv + jc + j : 0;
            vmask |= (b0 ? 1u : 0u) << (2 * k) | (b1 ? 1u : 0u) << (2 * k + 1);
        }
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            const int idx = k * 256 + tid, b = idx / (32 * HP), rem = idx - b * (32 * HP);
            const int c = rem / HP, j = 2 * (rem - c * HP);
            const int x0 = (jc + j) * st + b - p.pad, x1 = x0 + st;
            const bool okc = cu0 + c < p.cu;
            const bool b0 = okc && x0 >= 0 && x0 < p.wu && jc + j < p.wv, b1 = okc && x1 >= 0 && x1 < p.wu && jc + j + 1 < p.wv;
            uoff0[k] = b0 ? (cu0 + c) * p.hu * p.wu + x0 : 0;
            uoff1[k] = b1 ? (cu0 + c) * p.hu * p.wu + x1 : 0;
            umask |= (b0 ? 1u : 0u) << (2 * k) | (b1 ? 1u : 0u) << (2 * k + 1);
        }
    };
    auto load_v = [&](int i, float (&r)[VPT][2]) {                       // V row i (always inside the image)
        const float* row = vn + (size_t)i * p.wv;
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const float a0 = row[voff[k]], a1 = row[voff[k] + ((vmask >> (2 * k + 1)) & 1)];
            r[k][0] = (vmask >> (2 * k)) & 1 ? a0 : 0.f;
            r[k][1] = (vmask >> (2 * k + 1)) & 1 ? a1 : 0.f;
        }
    };
    auto store_v = [&](const float (&r)[VPT][2]) {
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const int idx = k * 256 + tid, c = idx / HP, j = 2 * (idx - c * HP);
            h2g hi, lo;
            split2(r[k][0] * scv, r[k][1] * scv, hi, lo);
            *reinterpret_cast<h2g*>(svh + c * NB_WHP + j) = hi;
            *reinterpret_cast<h2g*>(svl + c * NB_WHP + j) = lo;
        }
    };
    auto load_u = [&](int y, int part, float (&r)[UPT][2]) {             // U row y (may be a padding row), its three shifted / decimated copies
        (void)part;
        if (y >= 0 && y < p.hu) {                                         // (uniform)
            const float* row = un + (size_t)y * p.wu;
#pragma unroll
            for (int k = 0; k < UPT; ++k) {
                const float a0 = row[uoff0[k]], a1 = row[uoff1[k]];
                r[k][0] = (umask >> (2 * k)) & 1 ? a0 : 0.f;
                r[k][1] = (umask >> (2 * k + 1)) & 1 ? a1 : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < UPT; ++k) r[k][0] = r[k][1] = 0.f;
        }
    };
    auto store_u = [&](int y, const float (&r)[UPT][2]) {
        const int slot = y & 3;
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            const int idx = k * 256 + tid, b = idx / (32 * HP), rem = idx - b * (32 * HP);
            const int c = rem / HP, j = 2 * (rem - c * HP);
            h2g hi, lo;
            split2(r[k][0] * scu, r[k][1] * scu, hi, lo);
            *reinterpret_cast<h2g*>(suh + ((slot * 3 + b) * 32 + c) * NB_WHP + j) = hi;
            *reinterpret_cast<h2g*>(sul + ((slot * 3 + b) * 32 + c) * NB_WHP + j) = lo;
        }
    };
    const int i0 = sl * p.rows_per_wg, i1 = min(p.hv, (sl + 1) * p.rows_per_wg);
    float rv[VPT][2], ru0[UPT][2], ru1[UPT][2];
    for (int jc = 0; jc < p.wv; jc += NB_WHC) {
        if (i0 >= i1) break;
        __syncthreads();                                                  // previous column chunk's last MFMAs are done
        // prologue of the chunk: V row i0 and the three U rows it needs
        setup_chunk(jc);
        load_v(i0, rv); store_v(rv);
        for (int a = 0; a < 3; ++a) { load_u(i0 * st + a - p.pad, 0, ru0); store_u(i0 * st + a - p.pad, ru0); }
        __syncthreads();
        for (int i = i0; i < i1; ++i) {
            const bool more = i + 1 < i1;
            if (more) {                                                   // next row's operands -> registers (in flight under the MFMAs)
                load_v(i + 1, rv);
                if (st == 1) load_u((i + 1) + 2 - p.pad, 0, ru0);
                else { load_u((i + 1) * 2 + 1 - p.pad, 0, ru0); load_u((i + 1) * 2 + 2 - p.pad, 1, ru1); }
            }
            if (wave_active) {
                const int vo = (wvid * 32 + l31) * NB_WHP + 8 * lk, uo = l31 * NB_WHP + 8 * lk;
#pragma unroll
                for (int ks = 0; ks < NB_WHC / 16; ++ks) {
                    const h8g bh = *reinterpret_cast<const h8g*>(svh + vo + ks * 16);
                    const h8g bl = *reinterpret_cast<const h8g*>(svl + vo + ks * 16);
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const int slot = (i * st + a - p.pad) & 3;
#pragma unroll
                        for (int b_ = 0; b_ < 3; ++b_) {
                            const int ro = (slot * 3 + b_) * 32 * NB_WHP + uo + ks * 16;
                            const h8g ah = *reinterpret_cast<const h8g*>(suh + ro);
                            const h8g al = *reinterpret_cast<const h8g*>(sul + ro);
                            f32x16& c_ = acc[a * 3 + b_];
                            c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c_, 0, 0, 0);
                            c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c_, 0, 0, 0);
                            c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c_, 0, 0, 0);
                        }
                    }
                }
            }
            if (more) {
                __syncthreads();                                          // every wave has read row i's fragments
                store_v(rv);
                if (st == 1) store_u((i + 1) + 2 - p.pad, ru0);
                else { store_u((i + 1) * 2 + 1 - p.pad, ru0); store_u((i + 1) * 2 + 2 - p.pad, ru1); }
                __syncthreads();
            }
        }
    }
    const float inv = 1.f / (scu * scv);
    if (p.part) {
        // Partial results leave through LDS as whole rows of the [cu][cv][9] block (16 cu rows per pass: 16 x 1152 floats),
        // stored with 16-byte coalesced writes into this (slice, sample)'s own block: no atomics, no zero fill, and the sum over
        // slices (wgrad_reduce_kernel) has a fixed order.  (The atomic form below issued 144 scattered 36-byte-stride atomic
        // instructions per lane: ~300 us per call whatever the image size -- most of a training step's wgrad time.)
        constexpr int RUNP = 128 * 9 + 4;
        float* so = reinterpret_cast<float*>(smem_wh);                   // [16][RUNP]
        const int ncv = min(128, p.cv - cv0), run = ncv * 9;
        float* dst = p.part + (((size_t)sl * p.n + n) * p.cu) * p.cv * 9;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();                                              // fragment reads / the previous pass's copies are done
            if (wave_active) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int ml = (rr & 3) + 8 * (rr >> 2) + 4 * lk;  // row within the pass's 16
                        so[ml * RUNP + (wvid * 32 + l31) * 9 + t] = acc[t][half * 8 + rr] * inv;
                    }
            }
            __syncthreads();
            if ((p.cv & 3) == 0 && (ncv & 3) == 0) {
                const int run4 = run >> 2;
                for (int e = tid; e < 16 * run4; e += 256) {
                    const int row = e / run4, q = e - row * run4, m = cu0 + half * 16 + row;
                    if (m < p.cu)
                        *reinterpret_cast<f32x4*>(dst + ((size_t)m * p.cv + cv0) * 9 + 4 * q) = *reinterpret_cast<const f32x4*>(so + row * RUNP + 4 * q);
                }
            } else {
                for (int e = tid; e < 16 * run; e += 256) {
                    const int row = e / run, q = e - row * run, m = cu0 + half * 16 + row;
                    if (m < p.cu) dst[((size_t)m * p.cv + cv0) * 9 + q] = so[row * RUNP + q];
                }
            }
        }
        return;
    }
    if (!wave_active) return;
    const int cvl = cv0 + wvid * 32 + l31;
    if (cvl >= p.cv) return;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = cu0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < p.cu) atomicAdd(p.a + (((size_t)n * p.cu + m) * p.cv + cvl) * 9 + t, acc[t][r] * inv);
        }
}

// a[i] = sum over the `parts` partial blocks (each `count` floats apart), in a fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ a, long long count, int parts, int vec) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i * 4 >= count) return;
        f32x4 s = *reinterpret_cast<const f32x4*>(part + i * 4);
        for (int k = 1; k < parts; ++k) s += *reinterpret_cast<const f32x4*>(part + (size_t)k * count + i * 4);
        *reinterpret_cast<f32x4*>(a + i * 4) = s;
    } else {
        if (i >= count) return;
        float s = part[i];
        for (int k = 1; k < parts; ++k) s += part[(size_t)k * count + i];
        a[i] = s;
    }
}

static void nb_wgrad_h3_slices(int n, int cu, int cv, int hv, int* rows_per_wg, int* nslices) {
    const long tiles = (long)n * nb_cdiv(cu, 32) * nb_cdiv(cv, 128);
    const int wg_target = g_wgrad_wgs;     // 256 = one workgroup per CU
    int slices = (int)((wg_target + tiles - 1) / tiles);
    if (slices > hv) slices = hv;
    if (slices < 1) slices = 1;
    *rows_per_wg = nb_cdiv(hv, slices);
    *nslices = nb_cdiv(hv, *rows_per_wg);
}

extern "C" long long nb_conv2d_wgrad_h3_ws_bytes(int n, int cu, int cv, int hv, int sum_n) {
    if (n < 1 || cu < 1 || cv < 1 || hv < 1) return 0;
    int rows, nsl;
    nb_wgrad_h3_slices(n, cu, cv, hv, &rows, &nsl);
    const long long parts = (long long)nsl * (sum_n ? n : 1);
    return parts > 1 ? (long long)nsl * n * cu * cv * 9 * (long long)sizeof(float) : 0;
}

static int nb_wgrad_h3_impl(const float* u, const float* v, const float* scales, int scales_are_absmax, float* a, float* ws, long long ws_bytes, int sum_n, bool staged,
                            int n, int cu, int hu, int wu, int cv, int hv, int wv, int stride, int pad, void* stream) {
    NB_REQUIRE(u && v && a && scales, "conv2d_wgrad_h3: null pointer");
    NB_REQUIRE(n >= 1 && cu >= 1 && cv >= 1 && hu >= 1 && wu >= 1 && hv >= 1 && wv >= 1 && (stride == 1 || stride == 2) && pad >= 0,
               "conv2d_wgrad_h3: bad sizes (stride 1 or 2)");
    WgradParams p{u, v, a, n, cu, hu, wu, cv, hv, wv, stride, pad, 0, 0, nullptr, 0};
    p.scales_are_absmax = scales_are_absmax;
    nb_wgrad_h3_slices(n, cu, cv, hv, &p.rows_per_wg, &p.nslices);
    NB_REQUIRE((long)n * p.nslices <= 65535 && nb_cdiv(cv, 128) <= 65535, "conv2d_wgrad_h3: grid too large");
    const size_t lds = (size_t)(2 * 128 + 2 * 12 * 32) * NB_WHP * sizeof(_Float16);
    static_assert((size_t)(2 * 128 + 2 * 12 * 32) * NB_WHP * sizeof(_Float16) >= (size_t)16 * (128 * 9 + 4) * sizeof(float), "the output staging rows must fit the operand LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv2d_wgrad_h3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const long long need = nb_conv2d_wgrad_h3_ws_bytes(n, cu, cv, hv, sum_n);
    const int parts = p.nslices * (sum_n ? n : 1);
    if (staged) {
        NB_REQUIRE(need == 0 || (ws && ws_bytes >= need && (uintptr_t)ws % 16 == 0), "conv2d_wgrad_h3: workspace of %lld bytes (16-byte aligned) needed, got %lld", need, ws_bytes);
        NB_REQUIRE((uintptr_t)a % 16 == 0, "conv2d_wgrad_h3: output must be 16-byte aligned");
        p.part = parts > 1 ? ws : a;
    } else {
        (void)hipMemsetAsync(a, 0, (size_t)n * cu * cv * 9 * sizeof(float), (hipStream_t)stream);
    }
    dim3 grid(nb_cdiv(cu, 32), nb_cdiv(cv, 128), n * p.nslices);
    hipLaunchKernelGGL(conv2d_wgrad_h3_kernel, grid, dim3(256), lds, (hipStream_t)stream, p, scales);
    NB_CHECK_LAUNCH("conv2d_wgrad_h3");
    if (p.part && parts > 1) {
        // partial blocks are [slice][sample][cu][cv][9]: summing over slices only keeps the sample axis (count = n cu cv 9,
        // blocks one slice apart); summing over samples as well folds it (count = cu cv 9, nslices * n blocks)
        const long long count = (long long)(sum_n ? 1 : n) * cu * cv * 9;
        const int vec = count % 4 == 0;
        const long long items = vec ? count / 4 : count;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, a, count, parts, vec);
        NB_CHECK_LAUNCH("wgrad_reduce");
    }
    return NB_OK;
}

extern "C" int nb_conv2d_wgrad_h3(const float* u, const float* v, const float* scales, float* a, int n, int cu, int hu, int wu,
                                  int cv, int hv, int wv, int stride, int pad, void* stream) {
    return nb_wgrad_h3_impl(u, v, scales, 0, a, nullptr, 0, 0, false, n, cu, hu, wu, cv, hv, wv, stride, pad, stream);
}

extern "C" int nb_conv2d_wgrad_h3_ws(const float* u, const float* v, const float* scales, int scales_are_absmax, float* a, float* ws, long long ws_bytes,
                                     int sum_n, int n, int cu, int hu, int wu, int cv, int hv, int wv, int stride, int pad, void* stream) {
    return nb_wgrad_h3_impl(u, v, scales, scales_are_absmax, a, ws, ws_bytes, sum_n, true, n, cu, hu, wu, cv, hv, wv, stride, pad, stream);
}

// ------------------------------------------------------------------------------------------------
// Range scaling of the split-f16 training path without host round trips or strings of small launches: one launch leaves
// max|.| of the operands in two device slots, and the kernels that consume the operands derive the power-of-two scale from
// the slots themselves (nb_pow2_scale).
// ------------------------------------------------------------------------------------------------
// slots[0] = max(slots[0], max|a|, max|b|), slots[1] = max(slots[1], max|c|) -- as IEEE bit patterns (non-negative floats
// order like unsigned integers), so the caller zero-fills the slots once and atomicMax does the rest.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ a, long long na, const float* __restrict__ b, long long nb_,
                                                     const float* __restrict__ c, long long nc, unsigned* __restrict__ slots) {
    __shared__ float red[2][4];
    float m0 = 0.f, m1 = 0.f;
    const long long step = (long long)gridDim.x * 256, t0 = (long long)blockIdx.x * 256 + threadIdx.x;
    auto scan = [&](const float* p, long long count, float& m) {
        if (!p) return;
        const long long n4 = ((uintptr_t)p % 16 == 0) ? count / 4 : 0;
        for (long long i = t0; i < n4; i += step) {
            const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        for (long long i = n4 * 4 + t0; i < count; i += step) m = fmaxf(m, fabsf(p[i]));
    };
    scan(a, na, m0); scan(b, nb_, m0); scan(c, nc, m1);
    for (int o = 32; o > 0; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = m0; red[1][threadIdx.x >> 6] = m1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m0 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        m1 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        if (m0 > 0.f) atomicMax(slots, __float_as_uint(m0));
        if (m1 > 0.f) atomicMax(slots + 1, __float_as_uint(m1));
    }
}

extern "C" int nb_absmax_f32(const float* a, long long na, const float* b, long long nb_, const float* c, long long nc, void* slots, void* stream) {
    NB_REQUIRE(slots && na >= 0 && nb_ >= 0 && nc >= 0 && (a || na == 0) && (b || nb_ == 0) && (c || nc == 0), "absmax: bad arguments");
    const long long most = na > nb_ ? (na > nc ? na : nc) : (nb_ > nc ? nb_ : nc);
    long long blocks = (most / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, na, b, nb_, c, nc, (unsigned*)slots);
    NB_CHECK_LAUNCH("absmax");
    return NB_OK;
}

typedef _Float16 h8r __attribute__((ext_vector_type(8)));
// nb_pack_h2_f32 with the range scale folded in: out = H2((x1 ++ x2) * scale[n,c] * k), k = the power of two that brings
// slots[0] * slots[1] (= max|x| max|scale|) near `target`; workgroup (0,0,0) also leaves dco_out = dco_in / k (the consumer
// kernel's output coefficients undo the scale).
__global__ __launch_bounds__(256) void pack_h2_ranged_kernel(const float* __restrict__ x1, int c1, const float* __restrict__ x2, int c2,
                                                             const float* __restrict__ scale, _Float16* __restrict__ out, int c8, int hw,
                                                             const float* __restrict__ slots, float target, const float* __restrict__ dco_in,
                                                             float* __restrict__ dco_out, int dco_count) {
    const float s1 = slots[1];                                           // (a zero second slot = no second operand)
    const float k = nb_pow2_scale(target, slots[0] * (s1 > 0.f ? s1 : 1.f));
    if (dco_out && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        const float ik = 1.f / k;
        for (int i = threadIdx.x; i < dco_count; i += 256) dco_out[i] = dco_in[i] * ik;
    }
    const int n = blockIdx.z, cg = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const int c_in = c1 + c2;
    h8r hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = cg * 8 + j;
        float v = 0.f;
        if (ch < c_in) {
            v = ch < c1 ? x1[((size_t)n * c1 + ch) * hw + pix] : x2[((size_t)n * c2 + (ch - c1)) * hw + pix];
            v *= scale[(size_t)n * c_in + ch] * k;
        }
        const _Float16 hh = (_Float16)v;
        hi[j] = hh;
        lo[j] = (_Float16)(v - (float)hh);
    }
    h8r* o = reinterpret_cast<h8r*>(out) + ((size_t)(n * c8 + cg) * 2) * hw + pix;
    o[0] = hi;
    o[hw] = lo;
}

extern "C" int nb_pack_h2_ranged_f32(const float* x1, int c1, const float* x2, int c2, const float* scale, void* out_h2, int n, int hw,
                                     const void* slots, float target, const float* dco_in, float* dco_out, int dco_count, void* stream) {
    NB_REQUIRE(x1 && out_h2 && scale && slots && c1 > 0 && c2 >= 0 && (c2 == 0 || x2) && n > 0 && n <= 65535 && hw > 0 && target > 0.f,
               "pack_h2_ranged: bad arguments");
    NB_REQUIRE(dco_count == 0 || (dco_in && dco_out), "pack_h2_ranged: output coefficients need both pointers");
    const int c8 = (c1 + c2 + 7) / 8;
    dim3 grid((hw + 255) / 256, c8, n);
    hipLaunchKernelGGL(pack_h2_ranged_kernel, grid, dim3(256), 0, (hipStream_t)stream, x1, c1, x2, c2, scale, (_Float16*)out_h2, c8, hw,
                       (const float*)slots, target, dco_in, dco_count ? dco_out : nullptr, dco_count);
    NB_CHECK_LAUNCH("pack_h2_ranged");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// The tail of modulated_conv2d's backward pass in three small launches instead of ~40 torch operators per layer
// (einsum decompositions, element-wise passes over full activation tensors):
//   dd[n,o]   = sum_pix dy[n,o,p] * (y[n,o,p] - noise[n,p])                       (nb_modconv_bwd_dot_f32)
//   dW[o,c,t] = sum_n s[n,c] A[n,o,c,t] + 2 W[o,c,t] sum_n dq[n,o] s[n,c]^2       (nb_modconv_bwd_finish_f32)
//   ds[n,c]   = sum_{o,t} W[o,c,t] A[n,o,c,t] + 2 s[n,c] sum_o dq[n,o] Wsq[o,c]
// with A the per-sample weight-gradient correlation in either of the two layouts the wgrad kernel leaves it in
// ([n][c][o][9] for up = 1, [n][o][c][9] for up = 2: strides are arguments).  Fixed summation orders (no atomics).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void modconv_bwd_dot_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ noise,
                                                              long long noise_stride_n, float* __restrict__ out, int o, int hw) {
    __shared__ float red[4];
    const int plane = blockIdx.x, n = plane / o;
    const float* a = dy + (size_t)plane * hw;
    const float* b = y + (size_t)plane * hw;
    const float* nz = noise ? noise + (size_t)n * noise_stride_n : nullptr;
    float acc = 0.f;
    if ((hw & 3) == 0) {
        for (int i = threadIdx.x; i < hw / 4; i += 256) {
            const f32x4 u = reinterpret_cast<const f32x4*>(a)[i];
            f32x4 v = reinterpret_cast<const f32x4*>(b)[i];
            if (nz) v -= reinterpret_cast<const f32x4*>(nz)[i];
            acc += (u[0] * v[0] + u[1] * v[1]) + (u[2] * v[2] + u[3] * v[3]);
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += 256) acc += a[i] * (b[i] - (nz ? nz[i] : 0.f));
    }
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[plane] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int nb_modconv_bwd_dot_f32(const float* dy, const float* y, const float* noise, long long noise_stride_n, float* out,
                                      int n, int o, int hw, void* stream) {
    NB_REQUIRE(dy && y && out && n >= 1 && o >= 1 && hw >= 1 && (long long)n * o <= 0x7fffffffLL, "modconv_bwd_dot: bad arguments");
    NB_REQUIRE(!noise || (hw & 3) || (((uintptr_t)noise | (uintptr_t)(noise_stride_n * 4)) % 16 == 0), "modconv_bwd_dot: noise planes must be 16-byte aligned");
    NB_REQUIRE((hw & 3) || (((uintptr_t)dy | (uintptr_t)y) % 16 == 0), "modconv_bwd_dot: tensors must be 16-byte aligned");
    hipLaunchKernelGGL(modconv_bwd_dot_kernel, dim3(n * o), dim3(256), 0, (hipStream_t)stream, dy, y, noise, noise_stride_n, out, o, hw);
    NB_CHECK_LAUNCH("modconv_bwd_dot");
    return NB_OK;
}

// one thread per (o, c): the nine taps of dW
__global__ __launch_bounds__(256) void modconv_bwd_dw_kernel(const float* __restrict__ A, long long sn, long long so, long long sc, const float* __restrict__ s,
                                                             const float* __restrict__ W, const float* __restrict__ dq, float* __restrict__ dW, int n, int o, int c) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= o * c) return;
    const int io = idx / c, ic = idx - io * c;
    float acc[9] = {};
    float q = 0.f;
    for (int in = 0; in < n; ++in) {
        const float sv = s[(size_t)in * c + ic];
        const float* a = A + in * sn + io * so + ic * sc;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += sv * a[t];
        if (dq) q += dq[(size_t)in * o + io] * (sv * sv);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) dW[(size_t)idx * 9 + t] = dq ? acc[t] + 2.f * W[(size_t)idx * 9 + t] * q : acc[t];
}

// ds: one workgroup per (n, 32 channels); 8 threads share a channel, each summing every 8th output channel, then a fixed-order
// reduction through LDS
__global__ __launch_bounds__(256) void modconv_bwd_ds_kernel(const float* __restrict__ A, long long sn, long long so, long long sc, const float* __restrict__ s,
                                                            const float* __restrict__ W, const float* __restrict__ dq, float* __restrict__ ds, int n, int o, int c) {
    __shared__ float part[2][8][32];
    const int ctiles = (c + 31) / 32;
    const int in = blockIdx.x / ctiles, ic = (blockIdx.x - in * ctiles) * 32 + (threadIdx.x & 31), og = threadIdx.x >> 5;
    float acc = 0.f, q = 0.f;
    if (ic < c) {
        for (int io = og; io < o; io += 8) {
            const float* a = A + in * sn + io * so + ic * sc;
            const float* w = W + ((size_t)io * c + ic) * 9;
            float t1 = 0.f, wsq = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) { t1 += w[t] * a[t]; wsq += w[t] * w[t]; }
            acc += t1;
            if (dq) q += dq[(size_t)in * o + io] * wsq;
        }
    }
    part[0][og][threadIdx.x & 31] = acc; part[1][og][threadIdx.x & 31] = q;
    __syncthreads();
    if (og == 0 && ic < c) {
        float a_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) { a_ += part[0][g][threadIdx.x]; q_ += part[1][g][threadIdx.x]; }
        const size_t idx = (size_t)in * c + ic;
        ds[idx] = dq ? a_ + 2.f * s[idx] * q_ : a_;
    }
}

extern "C" int nb_modconv_bwd_finish_f32(const float* A, long long a_stride_n, long long a_stride_o, long long a_stride_c, const float* s,
                                         const float* W, const float* dq, float* dW, float* ds, int n, int o, int c, void* stream) {
    NB_REQUIRE(A && s && W && (dW || ds) && n >= 1 && o >= 1 && c >= 1, "modconv_bwd_finish: bad arguments");
    if (dW) {
        hipLaunchKernelGGL(modconv_bwd_dw_kernel, dim3((o * c + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, a_stride_n, a_stride_o, a_stride_c, s, W, dq, dW, n, o, c);
        NB_CHECK_LAUNCH("modconv_bwd_dw");
    }
    if (ds) {
        hipLaunchKernelGGL(modconv_bwd_ds_kernel, dim3(n * ((c + 31) / 32)), dim3(256), 0, (hipStream_t)stream, A, a_stride_n, a_stride_o, a_stride_c, s, W, dq, ds, n, o, c);
        NB_CHECK_LAUNCH("modconv_bwd_ds");
    }
    return NB_OK;
}
