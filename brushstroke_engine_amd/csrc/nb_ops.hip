// Non-contraction kernels of the NeuBE generator path for gfx950: bias_act, upfirdn2d (standalone
// operator parity with the reference plugins), mapping network, per-layer styles + demodulation
// coefficients, constant-noise resampling, triad ToRGB epilogue, feature blending.
// All are HBM- or latency-bound; they use 64-wide waves, 16-byte accesses where the layout allows,
// and LDS only for per-block constants.
#include "nb_common.h"
#include "nb_torgb.h"
#include <cmath>
#include <cstring>

// developer / test hooks shared by the conv launchers (declared in nb_common.h; include/neube_hip_debug.h): a bit mask that switches
// single features of the kernels off for A/B runs (8 = no XCD-aware workgroup order, 32 = the round-3 order of the up=2 launches, ...:
// the uses of p.dbg), and the first-round stagger of the large kernels in s_sleep ticks.  Both 0 on the product path.
int g_nb_debug_flags = 0, g_nb_stagger_ticks = 0;
extern "C" void nb_debug_set_flags(int flags) { g_nb_debug_flags = flags; }
extern "C" void nb_debug_set_stagger(int ticks) { g_nb_stagger_ticks = ticks; }
static bool g_upfirdn_generic = false;
extern "C" void nb_debug_set_upfirdn_generic(int on) { g_upfirdn_generic = on != 0; }


static thread_local char g_err[512] = "";

void nb_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* nb_last_error(void) { return g_err; }
extern "C" int nb_abi_version(void) { return NB_ABI_VERSION; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// bias_act  (reference: torch_utils/ops/bias_act.cu:23-147; forward and the two gradient modes, fp32)
//   grad 0: y = clamp(act(x + b) * gain)
//   grad 1: x carries the incoming gradient; y = x * act'(.) * gain, zero where the forward clamped
//   grad 2: x carries the gradient w.r.t. the grad-1 output, dy the original gradient;
//           y = x * dy * act''(.) * gain, zero where the forward clamped
//   act' / act'' are expressed through yref (= forward output, divided by gain) for every activation but
//   swish, which needs xref (+ b) instead - exactly the tensors the reference saves (bias_act.py:22-32 `ref`).
// HBM-bound streaming kernels: 16-byte accesses when sizes / alignment / bias step allow.
// ------------------------------------------------------------------------------------------------
#define NB_SELU_SCALE 1.0507009873554804934193349852946f
#define NB_SELU_ALPHA 1.6732632423543772848170429916717f

__device__ __forceinline__ float nb_act(float x, int act, float alpha) {
    switch (act) {
        case NB_ACT_RELU: return x > 0.f ? x : 0.f;
        case NB_ACT_LRELU: return x > 0.f ? x : x * alpha;
        case NB_ACT_TANH: return tanhf(x);
        case NB_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        case NB_ACT_ELU: return x >= 0.f ? x : expm1f(x);
        case NB_ACT_SELU: return x >= 0.f ? NB_SELU_SCALE * x : (NB_SELU_SCALE * NB_SELU_ALPHA) * expm1f(x);
        case NB_ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
        case NB_ACT_SWISH: return x / (1.f + expf(-x));
        default: return x;
    }
}

// one element of the gradient modes: g = incoming gradient, yy = forward output / gain, xr = xref + b
template <int G>
__device__ __forceinline__ float nb_act_grad(float g, float yy, float xr, int act, float alpha) {
    switch (act) {
        case NB_ACT_LINEAR: return G == 1 ? g : 0.f;
        case NB_ACT_RELU: return G == 1 ? (yy > 0.f ? g : 0.f) : 0.f;
        case NB_ACT_LRELU: return G == 1 ? (yy > 0.f ? g : g * alpha) : 0.f;
        case NB_ACT_TANH: { const float d = g * (1.f - yy * yy); return G == 1 ? d : d * (-2.f * yy); }
        case NB_ACT_SIGMOID: { const float d = g * yy * (1.f - yy); return G == 1 ? d : d * (1.f - 2.f * yy); }
        case NB_ACT_ELU: return yy >= 0.f ? (G == 1 ? g : 0.f) : g * (yy + 1.f);
        case NB_ACT_SELU: return yy >= 0.f ? (G == 1 ? g * NB_SELU_SCALE : 0.f) : g * (yy + NB_SELU_SCALE * NB_SELU_ALPHA);
        case NB_ACT_SOFTPLUS: { const float c = expf(-yy); return G == 1 ? g * (1.f - c) : g * c * (1.f - c); }
        case NB_ACT_SWISH: {
            if (xr > 40.f) return G == 1 ? g : 0.f;
            const float c = expf(xr), d = c + 1.f;
            return G == 1 ? g * c * (xr + d) / (d * d) : g * c * (xr * (2.f - d) + 2.f * d) / (d * d * d);
        }
        default: return 0.f;
    }
}

struct BiasActParams {
    const float* x; const float* b; const float* xref; const float* yref; const float* dy; float* y;
    long long size_x; int size_b, step_b, act; float alpha, gain, clamp;
};

template <int G>
__device__ __forceinline__ float nb_bias_act_elem(const BiasActParams& p, float x, float bb, float xref, float yref, float dy) {
    float y;
    if (G == 0) {
        y = nb_act(x + bb, p.act, p.alpha) * p.gain;
        if (p.clamp >= 0.f) y = fminf(fmaxf(y, -p.clamp), p.clamp);
        return y;
    }
    const float xr = xref + bb;
    const float yy = p.gain != 0.f ? yref / p.gain : 0.f;
    y = nb_act_grad<G>(x, yy, xr, p.act, p.alpha) * (p.gain * dy);
    if (p.clamp >= 0.f) {
        // where did the forward clamp?  swish saves x, not y: recompute its forward output
        const float yf = p.act == NB_ACT_SWISH ? nb_act(xr, NB_ACT_SWISH, 0.f) * p.gain : yref;
        if (!(yf > -p.clamp && yf < p.clamp)) y = 0.f;
    }
    return y;
}

template <int G, bool VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(const BiasActParams p) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (VEC) {
        const long long n4 = p.size_x >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            f32x4 v = reinterpret_cast<const f32x4*>(p.x)[i];
            f32x4 xr = {0.f, 0.f, 0.f, 0.f}, yr = xr, dv = {1.f, 1.f, 1.f, 1.f};
            if (G > 0 && p.xref) xr = reinterpret_cast<const f32x4*>(p.xref)[i];
            if (G > 0 && p.yref) yr = reinterpret_cast<const f32x4*>(p.yref)[i];
            if (G > 0 && p.dy) dv = reinterpret_cast<const f32x4*>(p.dy)[i];
            const float bb = p.size_b ? p.b[((i << 2) / p.step_b) % p.size_b] : 0.f;   // step_b % 4 == 0: one bias per vector
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = nb_bias_act_elem<G>(p, v[j], bb, xr[j], yr[j], dv[j]);
            reinterpret_cast<f32x4*>(p.y)[i] = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.size_x; i += stride) {
            const float bb = p.size_b ? p.b[(i / p.step_b) % p.size_b] : 0.f;
            p.y[i] = nb_bias_act_elem<G>(p, p.x[i], bb, (G > 0 && p.xref) ? p.xref[i] : 0.f,
                                         (G > 0 && p.yref) ? p.yref[i] : 0.f, (G > 0 && p.dy) ? p.dy[i] : 1.f);
        }
    }
}

extern "C" int nb_bias_act_grad_f32(const float* x, const float* b, const float* xref, const float* yref, const float* dy,
                                    float* y, int64_t size_x, int size_b, int step_b, int grad, int act, float alpha,
                                    float gain, float clamp, void* stream) {
    NB_REQUIRE(x && y, "bias_act: null pointer");
    NB_REQUIRE(size_x >= 0, "bias_act: negative size");
    NB_REQUIRE(size_b == 0 || (b && step_b >= 1), "bias_act: bias given without a valid step");
    NB_REQUIRE(act >= NB_ACT_LINEAR && act <= NB_ACT_SWISH, "bias_act: unsupported activation %d", act);
    NB_REQUIRE(grad >= 0 && grad <= 2, "bias_act: grad must be 0, 1 or 2");
    if (grad > 0) {
        const bool needs_x = act == NB_ACT_SWISH;
        const bool needs_y = !needs_x && (act != NB_ACT_LINEAR || clamp >= 0.f);
        NB_REQUIRE(!needs_x || xref, "bias_act: grad >= 1 of swish needs xref");
        NB_REQUIRE(!needs_y || yref, "bias_act: grad >= 1 of this activation / clamp needs yref");
    }
    if (size_x == 0) return NB_OK;
    uintptr_t al = (uintptr_t)x | (uintptr_t)y;
    if (grad > 0) al |= (uintptr_t)xref | (uintptr_t)yref | (uintptr_t)dy;          // null pointers do not disturb the test
    const bool vec = (size_x % 4 == 0) && (al % 16 == 0) && (size_b == 0 || step_b % 4 == 0);
    const long long work = vec ? size_x / 4 : size_x;
    int grid = (int)((work + 255) / 256);
    if (grid > 2048 * 4) grid = 2048 * 4;
    BiasActParams p;
    p.x = x; p.b = size_b ? b : nullptr; p.xref = xref; p.yref = yref; p.dy = dy; p.y = y; p.size_x = size_x;
    p.size_b = size_b; p.step_b = step_b; p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    hipStream_t st = (hipStream_t)stream;
#define NB_BA_LAUNCH(G) do { if (vec) hipLaunchKernelGGL((bias_act_kernel<G, true>), dim3(grid), dim3(256), 0, st, p); \
                             else hipLaunchKernelGGL((bias_act_kernel<G, false>), dim3(grid), dim3(256), 0, st, p); } while (0)
    if (grad == 0) NB_BA_LAUNCH(0); else if (grad == 1) NB_BA_LAUNCH(1); else NB_BA_LAUNCH(2);
#undef NB_BA_LAUNCH
    NB_CHECK_LAUNCH("bias_act");
    return NB_OK;
}

extern "C" int nb_bias_act_f32(const float* x, const float* b, float* y, int64_t size_x, int size_b, int step_b,
                               int act, float alpha, float gain, float clamp, void* stream) {
    return nb_bias_act_grad_f32(x, b, nullptr, nullptr, nullptr, y, size_x, size_b, step_b, 0, act, alpha, gain, clamp, stream);
}

// ------------------------------------------------------------------------------------------------
// upfirdn2d  (reference: torch_utils/ops/upfirdn2d.cu:29-92 generic kernel semantics, fp32, NCHW)
// ------------------------------------------------------------------------------------------------
struct UpfirdnParams {
    const float* x; const float* f; float* y;
    int major, in_h, in_w, out_h, out_w, f_h, f_w, upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
};

__global__ __launch_bounds__(256) void upfirdn2d_kernel(const UpfirdnParams p) {
    extern __shared__ float sf[];
    for (int i = threadIdx.x; i < p.f_h * p.f_w; i += blockDim.x) {
        // store the filter so that tap (fy, fx) below is always "filter as correlated with the padded input"
        const int fy = i / p.f_w, fx = i % p.f_w;
        const int sy = p.flip ? fy : p.f_h - 1 - fy, sx = p.flip ? fx : p.f_w - 1 - fx;
        sf[i] = p.f[sy * p.f_w + sx] * p.gain;
    }
    __syncthreads();
    const long long total = (long long)p.major * p.out_h * p.out_w;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % p.out_w);
        const int oy = (int)((idx / p.out_w) % p.out_h);
        const int m = (int)(idx / ((long long)p.out_w * p.out_h));
        // position of the window's first tap in the upsampled (zero-stuffed), un-padded image
        const int ux0 = ox * p.downx - p.padx0, uy0 = oy * p.downy - p.pady0;
        const float* xm = p.x + (size_t)m * p.in_h * p.in_w;
        float v = 0.f;
        // only every upy-th (upx-th) tap meets a sample of the zero-stuffed image: start at the first one, step by the
        // up-sampling factor (no modulo per tap), and clip the tap range to the image once
        const int fy0 = uy0 >= 0 ? (p.upy - uy0 % p.upy) % p.upy : -uy0;      // first tap with uy >= 0 and uy % upy == 0 (uy = 0 when uy0 < 0)
        const int fx0 = ux0 >= 0 ? (p.upx - ux0 % p.upx) % p.upx : -ux0;
        for (int fy = fy0; fy < p.f_h; fy += p.upy) {
            const int iy = (uy0 + fy) / p.upy;
            if (iy >= p.in_h) break;
            const float* xr = xm + iy * p.in_w;
            const float* fr = sf + fy * p.f_w;
            for (int fx = fx0; fx < p.f_w; fx += p.upx) {
                const int ix = (ux0 + fx) / p.upx;
                if (ix >= p.in_w) break;
                v += xr[ix] * fr[fx];
            }
        }
        p.y[idx] = v;
    }
}

// The configurations the networks actually use (4x4 FIR at up / down 1 or 2, zero stuffing, the separable 1-D passes of the
// augmentation pipe) with the factors -- and for the 4x4 / 1x1 filters the tap loops -- fixed at compile time: one
// workgroup = 256 consecutive outputs of one image plane, 32-bit index arithmetic, no division per tap.  (The generic
// kernel above spends most of its time in 64-bit index divisions: 0.45 TB/s on a 256x256 x 512-plane FIR; this form is
// bound by the cache path.)  FH = FW = 0: filter size at run time.
template <int UPX, int UPY, int DX, int DY, int FH, int FW>
__global__ __launch_bounds__(256) void upfirdn2d_t_kernel(const UpfirdnParams p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if constexpr (FH * FW > 0) {
        // filter taps in registers (uniform loads), "as correlated with the padded input" and times the gain, like sf[] below
        float w[FH][FW];
#pragma unroll
        for (int fy = 0; fy < FH; ++fy)
#pragma unroll
            for (int fx = 0; fx < FW; ++fx) w[fy][fx] = p.f[(p.flip ? fy : FH - 1 - fy) * FW + (p.flip ? fx : FW - 1 - fx)] * p.gain;
        if (idx >= p.out_h * p.out_w) return;
        const int oy = idx / p.out_w, ox = idx - oy * p.out_w;
        const int ux0 = ox * DX - p.padx0, uy0 = oy * DY - p.pady0;
        // taps that meet a sample of the zero-stuffed image: f = phase + UP t; their input pixel (u0 + f) / UP = base + t may
        // lie outside the image on either side (masked below), which walks the same taps in the same order as the generic kernel
        const int phx = ((-ux0) % UPX + UPX) % UPX, phy = ((-uy0) % UPY + UPY) % UPY;
        const int ixb = (ux0 + phx) / UPX, iyb = (uy0 + phy) / UPY;     // (exact divisions)
        constexpr int NTY = (FH + UPY - 1) / UPY, NTX = (FW + UPX - 1) / UPX;
        float wt[NTY][NTX];                                               // this output's taps
        bool okx[NTX], oky[NTY];
        int cx[NTX], cy[NTY];
#pragma unroll
        for (int tx = 0; tx < NTX; ++tx) { okx[tx] = phx + UPX * tx < FW && ixb + tx >= 0 && ixb + tx < p.in_w; cx[tx] = min(max(ixb + tx, 0), p.in_w - 1); }
#pragma unroll
        for (int ty = 0; ty < NTY; ++ty) { oky[ty] = phy + UPY * ty < FH && iyb + ty >= 0 && iyb + ty < p.in_h; cy[ty] = min(max(iyb + ty, 0), p.in_h - 1) * p.in_w; }
#pragma unroll
        for (int ty = 0; ty < NTY; ++ty)
#pragma unroll
            for (int tx = 0; tx < NTX; ++tx) {
                float t = 0.f;
#pragma unroll
                for (int a = 0; a < UPY; ++a)
#pragma unroll
                    for (int b = 0; b < UPX; ++b)
                        if (a + UPY * ty < FH && b + UPX * tx < FW) t = (phy == a && phx == b) ? w[a + UPY * ty][b + UPX * tx] : t;
                wt[ty][tx] = t;
            }
        for (int m = blockIdx.y; m < p.major; m += gridDim.y) {
            const float* xm = p.x + (size_t)m * p.in_h * p.in_w;
            float xv[NTY][NTX];
#pragma unroll
            for (int ty = 0; ty < NTY; ++ty)
#pragma unroll
                for (int tx = 0; tx < NTX; ++tx) xv[ty][tx] = xm[cy[ty] + cx[tx]];
            float v = 0.f;
#pragma unroll
            for (int ty = 0; ty < NTY; ++ty)
#pragma unroll
                for (int tx = 0; tx < NTX; ++tx) {
                    const float t = xv[ty][tx] * wt[ty][tx];
                    v = (oky[ty] && okx[tx]) ? v + t : v;
                }
            p.y[(size_t)m * p.out_h * p.out_w + idx] = v;
        }
    } else {
        __shared__ float sf[1024];
        const int fh = p.f_h, fw = p.f_w;
        for (int i = threadIdx.x; i < fh * fw; i += 256) {
            const int fy = i / fw, fx = i - fy * fw;
            const int sy = p.flip ? fy : fh - 1 - fy, sx = p.flip ? fx : fw - 1 - fx;
            sf[i] = p.f[sy * fw + sx] * p.gain;
        }
        __syncthreads();
        if (idx >= p.out_h * p.out_w) return;
        const int oy = idx / p.out_w, ox = idx - oy * p.out_w;
        const int ux0 = ox * DX - p.padx0, uy0 = oy * DY - p.pady0;
        // first tap that meets a sample of the zero-stuffed image (see the generic kernel)
        const int fy0 = uy0 >= 0 ? (UPY == 1 ? 0 : (UPY - uy0 % UPY) % UPY) : -uy0;
        const int fx0 = ux0 >= 0 ? (UPX == 1 ? 0 : (UPX - ux0 % UPX) % UPX) : -ux0;
        const int iy0 = (uy0 + fy0) / UPY, ix0 = (ux0 + fx0) / UPX;      // (numerators >= 0)
        for (int m = blockIdx.y; m < p.major; m += gridDim.y) {
            const float* xm = p.x + (size_t)m * p.in_h * p.in_w;
            float v = 0.f;
            for (int fy = fy0, iy = iy0; fy < fh && iy < p.in_h; fy += UPY, ++iy)
                for (int fx = fx0, ix = ix0; fx < fw && ix < p.in_w; fx += UPX, ++ix) v += xm[iy * p.in_w + ix] * sf[fy * fw + fx];
            p.y[(size_t)m * p.out_h * p.out_w + idx] = v;
        }
    }
}

template <int UPX, int UPY, int DX, int DY, int FH, int FW>
static void nb_upfirdn2d_launch_t(const UpfirdnParams& p, hipStream_t st) {
    const int per_plane = (p.out_h * p.out_w + 255) / 256;
    // each thread walks several planes (set-up amortised) as long as the launch still has a few thousand workgroups
    int gy = p.major;
    while (gy > 1 && (long long)per_plane * gy > 8192 && gy * 4 > p.major) gy = (gy + 1) / 2;
    if (gy > 65535) gy = 65535;
    hipLaunchKernelGGL((upfirdn2d_t_kernel<UPX, UPY, DX, DY, FH, FW>), dim3(per_plane, gy), dim3(256), 0, st, p);
}

extern "C" int nb_upfirdn2d_f32(const float* x, const float* f, float* y, int major, int in_h, int in_w, int f_h,
                                int f_w, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0,
                                int pady1, int flip, float gain, void* stream) {
    NB_REQUIRE(x && f && y, "upfirdn2d: null pointer");
    NB_REQUIRE(major >= 1 && in_h >= 1 && in_w >= 1, "upfirdn2d: empty input");
    NB_REQUIRE(f_h >= 1 && f_w >= 1 && f_h * f_w <= 1024, "upfirdn2d: filter must be between 1x1 and 1024 taps");
    NB_REQUIRE(upx >= 1 && upy >= 1, "upfirdn2d: upsampling factor must be at least 1");
    NB_REQUIRE(downx >= 1 && downy >= 1, "upfirdn2d: downsampling factor must be at least 1");
    UpfirdnParams p;
    p.x = x; p.f = f; p.y = y; p.major = major; p.in_h = in_h; p.in_w = in_w; p.f_h = f_h; p.f_w = f_w;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0; p.flip = flip; p.gain = gain;
    p.out_w = (in_w * upx + padx0 + padx1 - f_w + downx) / downx;
    p.out_h = (in_h * upy + pady0 + pady1 - f_h + downy) / downy;
    NB_REQUIRE(p.out_w >= 1 && p.out_h >= 1, "upfirdn2d: output must be at least 1x1");
    const long long total = (long long)major * p.out_h * p.out_w;
    hipStream_t st = (hipStream_t)stream;
    const bool generic_only = g_upfirdn_generic;      // developer switch (nb_debug_set_upfirdn_generic): the run-time-everything kernel
    const long long plane = (long long)p.out_h * p.out_w, in_plane = (long long)in_h * in_w;
    // (factors of 4 and more would alias the 2-bit fields of the key -- down = (1, 5) reads as <1, 1, 2, 1> -- : generic kernel)
    const bool small_factors = upx <= 3 && upy <= 3 && downx <= 3 && downy <= 3;
    const int key = generic_only || !small_factors || plane >= (1LL << 30) || in_plane >= (1LL << 30) ? -1 : ((upx * 4 + upy) * 4 + downx) * 4 + downy;
    const bool f44 = f_h == 4 && f_w == 4, f11 = f_h == 1 && f_w == 1;
    bool done = true;
    switch (key) {
    case ((1 * 4 + 1) * 4 + 1) * 4 + 1:
        if (f44) nb_upfirdn2d_launch_t<1, 1, 1, 1, 4, 4>(p, st); else nb_upfirdn2d_launch_t<1, 1, 1, 1, 0, 0>(p, st);
        break;
    case ((2 * 4 + 2) * 4 + 1) * 4 + 1:
        if (f44) nb_upfirdn2d_launch_t<2, 2, 1, 1, 4, 4>(p, st); else if (f11) nb_upfirdn2d_launch_t<2, 2, 1, 1, 1, 1>(p, st); else nb_upfirdn2d_launch_t<2, 2, 1, 1, 0, 0>(p, st);
        break;
    case ((1 * 4 + 1) * 4 + 2) * 4 + 2:
        if (f44) nb_upfirdn2d_launch_t<1, 1, 2, 2, 4, 4>(p, st); else nb_upfirdn2d_launch_t<1, 1, 2, 2, 0, 0>(p, st);
        break;
    case ((2 * 4 + 1) * 4 + 1) * 4 + 1: nb_upfirdn2d_launch_t<2, 1, 1, 1, 0, 0>(p, st); break;
    case ((1 * 4 + 2) * 4 + 1) * 4 + 1: nb_upfirdn2d_launch_t<1, 2, 1, 1, 0, 0>(p, st); break;
    case ((1 * 4 + 1) * 4 + 2) * 4 + 1: nb_upfirdn2d_launch_t<1, 1, 2, 1, 0, 0>(p, st); break;
    case ((1 * 4 + 1) * 4 + 1) * 4 + 2: nb_upfirdn2d_launch_t<1, 1, 1, 2, 0, 0>(p, st); break;
    default: done = false;
    }
    if (!done) {
        int grid = (int)((total + 255) / 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(upfirdn2d_kernel, dim3(grid), dim3(256), f_h * f_w * sizeof(float), st, p);
    }
    NB_CHECK_LAUNCH("upfirdn2d");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// mapping network  (reference: training/networks.py:255-290, :109-122, :24-26)
// one workgroup per latent; activations ping-pong through LDS; one output feature per thread
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mapping_kernel(const float* __restrict__ z, const float* __restrict__ fc_w,
                                                      const float* __restrict__ fc_b, float* __restrict__ w_out,
                                                      int z_dim, int w_dim, int num_layers, float lr_mul, int num_ws) {
    __shared__ float xa[512], xb[512], red[8];
    const int n = blockIdx.x, t = threadIdx.x;
    // normalize_2nd_moment
    float v = t < z_dim ? z[(size_t)n * z_dim + t] : 0.f;
    float sq = v * v;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((t & 63) == 0) red[t >> 6] = sq;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < 8; ++i) tot += red[i];
    xa[t] = v * rsqrtf(tot / (float)z_dim + 1e-8f);
    __syncthreads();
    float* cur = xa; float* nxt = xb;
    const float* wl = fc_w;
    for (int l = 0; l < num_layers; ++l) {
        const int in = l == 0 ? z_dim : w_dim;
        const float wg = lr_mul / sqrtf((float)in);
        if (t < w_dim) {
            const float* wr = wl + (size_t)t * in;
            float acc = 0.f;
            for (int i = 0; i < in; ++i) acc += cur[i] * (wr[i] * wg);
            acc += fc_b[l * w_dim + t] * lr_mul;
            acc = (acc > 0.f ? acc : acc * 0.2f) * 1.41421356237309515f;
            nxt[t] = acc;
        }
        __syncthreads();
        wl += (size_t)w_dim * in;
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    if (t < w_dim)
        for (int k = 0; k < num_ws; ++k) w_out[((size_t)n * num_ws + k) * w_dim + t] = cur[t];
}

// The same network for the style1 shapes (z_dim = w_dim = 64, <= 8 layers), latency-oriented (it heads the batch-1
// step): 8 lanes share one output feature (8 inputs each, two 16-byte loads of the weight row), every layer's weight
// slice is fetched into registers before the first layer starts, and the partial sums meet through three lane shuffles.
#define NB_MAP_MAXL 8
__global__ __launch_bounds__(512) void mapping64_kernel(const float* __restrict__ z, const float* __restrict__ fc_w,
                                                        const float* __restrict__ fc_b, float* __restrict__ w_out,
                                                        int num_layers, float lr_mul, int num_ws) {
    __shared__ float xs[2][64];
    const int n = blockIdx.x, t = threadIdx.x, o = t >> 3, part = t & 7;
    f32x4 wr[NB_MAP_MAXL][2];
    float bs[NB_MAP_MAXL];
#pragma unroll
    for (int l = 0; l < NB_MAP_MAXL; ++l) {
        if (l < num_layers) {
            const f32x4* src = reinterpret_cast<const f32x4*>(fc_w + ((size_t)l * 64 + o) * 64 + part * 8);
            wr[l][0] = src[0]; wr[l][1] = src[1];
            bs[l] = fc_b[l * 64 + o];
        }
    }
    if (t < 64) {                                    // normalize_2nd_moment (one wave holds the whole latent)
        const float v = z[(size_t)n * 64 + t];
        float sq = v * v;
        for (int d = 32; d > 0; d >>= 1) sq += __shfl_xor(sq, d);
        xs[0][t] = v * rsqrtf(sq / 64.f + 1e-8f);
    }
    __syncthreads();
    const float wg = lr_mul / 8.f;                   // lr_mul / sqrt(64)
#pragma unroll
    for (int l = 0; l < NB_MAP_MAXL; ++l) {
        if (l < num_layers) {
            const float* cur = xs[l & 1] + part * 8;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += cur[i] * (wr[l][i >> 2][i & 3] * wg);
            acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
            if (part == 0) {
                acc += bs[l] * lr_mul;
                xs[(l + 1) & 1][o] = (acc > 0.f ? acc : acc * 0.2f) * 1.41421356237309515f;
            }
            __syncthreads();
        }
    }
    const float* fin = xs[num_layers & 1];
    for (int e = t; e < num_ws * 64; e += 512) w_out[(size_t)n * num_ws * 64 + e] = fin[e & 63];
}

static int nb_mapping_impl(const float* z, const float* fc_w, const float* fc_b, float* w_out, int n, int z_dim, int w_dim,
                           int num_layers, float lr_mul, int num_ws, void* stream) {
    NB_REQUIRE(z && fc_w && fc_b && w_out, "mapping: null pointer");
    NB_REQUIRE(n >= 1, "mapping: empty batch");
    NB_REQUIRE(z_dim >= 1 && z_dim <= 512 && w_dim >= 1 && w_dim <= 512, "mapping: z_dim/w_dim must be in [1,512]");
    NB_REQUIRE(num_layers >= 1 && num_ws >= 1, "mapping: need at least one layer and one ws row");
    if (z_dim == 64 && w_dim == 64 && num_layers <= NB_MAP_MAXL && ((uintptr_t)fc_w % 16) == 0)
        hipLaunchKernelGGL(mapping64_kernel, dim3(n), dim3(512), 0, (hipStream_t)stream, z, fc_w, fc_b, w_out, num_layers, lr_mul, num_ws);
    else
        hipLaunchKernelGGL(mapping_kernel, dim3(n), dim3(512), 0, (hipStream_t)stream, z, fc_w, fc_b, w_out, z_dim, w_dim, num_layers, lr_mul, num_ws);
    NB_CHECK_LAUNCH("mapping");
    return NB_OK;
}

extern "C" int nb_mapping_f32(const float* z, const float* fc_w, const float* fc_b, float* w_out, int n, int z_dim,
                              int w_dim, int num_layers, float lr_mul, void* stream) {
    return nb_mapping_impl(z, fc_w, fc_b, w_out, n, z_dim, w_dim, num_layers, lr_mul, 1, stream);
}

extern "C" int nb_mapping_ws_f32(const float* z, const float* fc_w, const float* fc_b, float* ws_out, int n, int z_dim,
                                 int w_dim, int num_layers, float lr_mul, int num_ws, void* stream) {
    return nb_mapping_impl(z, fc_w, fc_b, ws_out, n, z_dim, w_dim, num_layers, lr_mul, num_ws, stream);
}

// ------------------------------------------------------------------------------------------------
// per-layer styles (affine) + demodulation coefficients, all layers in one launch
// grid = (n_layers, n); reference: networks.py:366 (affine), :59-62 (dcoefs), :458-460 (ToRGB split/scale)
// ------------------------------------------------------------------------------------------------
#define NB_MAX_AFF 1024
__global__ __launch_bounds__(256) void styles_kernel(const NbLayerDesc* __restrict__ layers, const float* __restrict__ ws,
                                                     int num_ws, int w_dim) {
    __shared__ float wv[512];
    __shared__ float s2[NB_MAX_AFF];
    const NbLayerDesc L = layers[blockIdx.x];
    const int n = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < w_dim; i += 256) wv[i] = ws[((size_t)n * num_ws + L.w_index) * w_dim + i];
    __syncthreads();
    const float wg = 1.f / sqrtf((float)w_dim);
    for (int c = t; c < L.c_aff; c += 256) {
        const float* wr = L.affine_w + (size_t)c * w_dim;
        float acc = 0.f;
        for (int i = 0; i < w_dim; ++i) acc += wv[i] * (wr[i] * wg);
        acc += L.affine_b[c];
        if (c >= L.n_plain) acc *= L.style_scale;
        L.styles[(size_t)n * L.c_aff + c] = acc;
        s2[c] = acc * acc;
    }
    __syncthreads();
    if (L.wsq) {
        const int c_in = L.c_aff - L.n_plain;
        for (int o = t; o < L.c_out; o += 256) {
            float acc = 0.f;
            for (int i = 0; i < c_in; ++i) acc += s2[L.n_plain + i] * L.wsq[(size_t)i * L.c_out + o];
            L.dcoefs[(size_t)n * L.c_out + o] = rsqrtf(acc + 1e-8f);
        }
    }
}

// Latency-oriented variant (w_dim % 16 == 0, every c_out % 4 == 0; grid = (n_layers, n, NB_STY_PARTS)): four lanes share
// one affine output (16-byte loads of the weight row, two shuffles), and the demodulation sum of a layer is split over
// NB_STY_PARTS workgroups (each recomputes the cheap affine and takes a slice of the c_out outputs) and, inside a
// workgroup, over c_in slices whose partial sums meet in LDS - instead of one thread walking all c_in rows.
#define NB_STY_PARTS 4
// With grid.z > NB_STY_PARTS the extra z-slices compute the layer's position-shifted noise image of sample n (the
// noise kernel's work, independent of the styles): one launch instead of two at the head of the step.
__device__ __forceinline__ void nb_noise_sample(const NbLayerDesc& L, const float* __restrict__ norm_pos,
                                                const long long* __restrict__ positions, int img_res, int n, int idx0, int stride);
__global__ __launch_bounds__(256) void styles_fast_kernel(const NbLayerDesc* __restrict__ layers, const float* __restrict__ ws,
                                                          int num_ws, int w_dim, const float* __restrict__ norm_pos,
                                                          const long long* __restrict__ positions, int img_res) {
    __shared__ __attribute__((aligned(16))) float wv[512];
    __shared__ float s2[NB_MAX_AFF];
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const NbLayerDesc L = layers[blockIdx.x];
    const int n = blockIdx.y, part = blockIdx.z, t = threadIdx.x;
    if (part >= NB_STY_PARTS) {
        if (L.noise_const) {
            const int nz = gridDim.z - NB_STY_PARTS;
            nb_noise_sample(L, norm_pos, positions, img_res, n, (part - NB_STY_PARTS) * 256 + t, nz * 256);
        }
        return;
    }
    if (!L.wsq && part > 0) return;                                 // ToRGB: no demodulation, one workgroup does the affine
    for (int i = t; i < w_dim; i += 256) wv[i] = ws[((size_t)n * num_ws + L.w_index) * w_dim + i];
    __syncthreads();
    const float wg = 1.f / sqrtf((float)w_dim);
    const int q = t & 3, per = w_dim >> 2;                          // lane q of 4 covers columns q*per .. +per
    for (int c0 = 0; c0 < L.c_aff; c0 += 64) {
        const int c = c0 + (t >> 2);
        float acc = 0.f;
        if (c < L.c_aff) {
            const f32x4* wr = reinterpret_cast<const f32x4*>(L.affine_w + (size_t)c * w_dim + q * per);
            const f32x4* xv = reinterpret_cast<const f32x4*>(wv + q * per);
            for (int i = 0; i < per / 4; ++i) {
                const f32x4 a = wr[i], b = xv[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += b[j] * (a[j] * wg);
            }
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
        if (c < L.c_aff && q == 0) {
            acc += L.affine_b[c];
            if (c >= L.n_plain) acc *= L.style_scale;
            if (part == 0) L.styles[(size_t)n * L.c_aff + c] = acc;
            s2[c] = acc * acc;
        }
    }
    __syncthreads();
    if (!L.wsq) return;
    const int c_in = L.c_aff - L.n_plain;
    const int slice = ((L.c_out + 4 * NB_STY_PARTS - 1) / (4 * NB_STY_PARTS)) * 4;     // outputs per workgroup (multiple of 4)
    const int o_lo = part * slice;
    const int ng = slice >> 2;                                      // float4 groups across the slice
    if (ng > 64 || o_lo >= L.c_out) {
        if (o_lo < L.c_out)                                         // very wide layer: plain loop (not a style1 shape)
            for (int o = o_lo + t; o < min(o_lo + slice, L.c_out); o += 256) {
                float acc = 0.f;
                for (int i = 0; i < c_in; ++i) acc += s2[L.n_plain + i] * L.wsq[(size_t)i * L.c_out + o];
                L.dcoefs[(size_t)n * L.c_out + o] = rsqrtf(acc + 1e-8f);
            }
        return;
    }
    const int nks = 256 / ng, g = t % ng, ks = t / ng;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int o4 = o_lo + 4 * g;
    if (ks < nks && o4 < L.c_out) {
#pragma unroll 4
        for (int i = ks; i < c_in; i += nks) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(L.wsq + (size_t)i * L.c_out + o4);
            acc += s2[L.n_plain + i] * w4;
        }
    }
    if (ks < nks) *reinterpret_cast<f32x4*>(red + (ks * ng + g) * 4) = acc;
    __syncthreads();
    if (t < slice && o_lo + t < L.c_out) {
        float sum = 0.f;
        for (int k = 0; k < nks; ++k) sum += red[(k * ng + (t >> 2)) * 4 + (t & 3)];
        L.dcoefs[(size_t)n * L.c_out + o_lo + t] = rsqrtf(sum + 1e-8f);
    }
}

extern "C" int nb_styles_f32(const NbLayerDesc* layers_dev, int n_layers, const float* ws, int num_ws, int w_dim, int n,
                             void* stream) {
    NB_REQUIRE(layers_dev && ws, "styles: null pointer");
    NB_REQUIRE(n_layers >= 1 && n >= 1 && n <= 65535, "styles: bad sizes");
    NB_REQUIRE(w_dim >= 1 && w_dim <= 512, "styles: w_dim must be in [1,512]");
    hipLaunchKernelGGL(styles_kernel, dim3(n_layers, n), dim3(256), 0, (hipStream_t)stream, layers_dev, ws, num_ws, w_dim);
    NB_CHECK_LAUNCH("styles");
    return NB_OK;
}

extern "C" int nb_styles_fast_f32(const NbLayerDesc* layers_dev, int n_layers, const float* ws, int num_ws, int w_dim, int n,
                                  void* stream) {
    NB_REQUIRE(layers_dev && ws, "styles: null pointer");
    NB_REQUIRE(n_layers >= 1 && n >= 1 && n <= 65535, "styles: bad sizes");
    NB_REQUIRE(w_dim >= 16 && w_dim <= 512 && w_dim % 16 == 0, "styles_fast: w_dim must be a multiple of 16 in [16,512]");
    hipLaunchKernelGGL(styles_fast_kernel, dim3(n_layers, n, NB_STY_PARTS), dim3(256), 0, (hipStream_t)stream, layers_dev, ws, num_ws, w_dim,
                       (const float*)nullptr, (const long long*)nullptr, 0);
    NB_CHECK_LAUNCH("styles_fast");
    return NB_OK;
}

extern "C" int nb_styles_noise_f32(const NbLayerDesc* layers_dev, int n_layers, const float* ws, int num_ws, int w_dim,
                                   const float* norm_pos, const int64_t* positions, int img_resolution, int n, void* stream) {
    NB_REQUIRE(layers_dev && ws, "styles_noise: null pointer");
    NB_REQUIRE(n_layers >= 1 && n >= 1 && n <= 65535, "styles_noise: bad sizes");
    NB_REQUIRE(w_dim >= 16 && w_dim <= 512 && w_dim % 16 == 0, "styles_noise: w_dim must be a multiple of 16 in [16,512]");
    NB_REQUIRE((norm_pos != nullptr) != (positions != nullptr), "styles_noise: pass exactly one of norm_pos / positions (per-sample noise)");
    NB_REQUIRE(!positions || img_resolution >= 2, "styles_noise: positions need img_resolution >= 2");
    hipLaunchKernelGGL(styles_fast_kernel, dim3(n_layers, n, NB_STY_PARTS + 64), dim3(256), 0, (hipStream_t)stream, layers_dev, ws, num_ws,
                       w_dim, norm_pos, (const long long*)positions, img_resolution);
    NB_CHECK_LAUNCH("styles_noise");
    return NB_OK;
}

// standalone demodulation coefficients (networks.py:59-62): d[n,o] = rsqrt(sum_i s[n,i]^2 * wsq[i,o] + 1e-8)
__global__ __launch_bounds__(256) void demod_kernel(const float* __restrict__ styles, const float* __restrict__ wsq,
                                                    float* __restrict__ dcoefs, int c_in, int c_out) {
    const int n = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
    if (o >= c_out) return;
    float acc = 0.f;
    for (int i = 0; i < c_in; ++i) {
        const float s = styles[(size_t)n * c_in + i];
        acc += (s * s) * wsq[(size_t)i * c_out + o];
    }
    dcoefs[(size_t)n * c_out + o] = rsqrtf(acc + 1e-8f);
}

extern "C" int nb_demod_coefs_f32(const float* styles, const float* wsq, float* dcoefs, int n, int c_in, int c_out,
                                  void* stream) {
    NB_REQUIRE(styles && wsq && dcoefs, "demod_coefs: null pointer");
    NB_REQUIRE(n >= 1 && n <= 65535 && c_in >= 1 && c_out >= 1, "demod_coefs: bad sizes");
    hipLaunchKernelGGL(demod_kernel, dim3(nb_cdiv(c_out, 256), n), dim3(256), 0, (hipStream_t)stream, styles, wsq, dcoefs, c_in, c_out);
    NB_CHECK_LAUNCH("demod_coefs");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// constant noise, optionally position-shifted (networks.py:371-382, SURVEY note C), all layers at once
// ------------------------------------------------------------------------------------------------
// one sample's noise image of one layer, pixels idx0, idx0 + stride, ... (shared by the stand-alone and the fused launch)
__device__ __forceinline__ void nb_noise_sample(const NbLayerDesc& L, const float* __restrict__ norm_pos,
                                                const long long* __restrict__ positions, int img_res, int n, int idx0, int stride) {
    const int r = L.res;
    const float strength = L.noise_strength[0];
    float np0 = 0.f, np1 = 0.f;
    if (positions) {
        // networks_modified.py:351-353: (positions % R) / (R - 1), python-style modulo, IEEE float32 division
        // (done here, not with a torch GPU op: torch's device division is not correctly rounded and the
        // wrap `% 1` below is discontinuous, so a 1-ulp difference moves whole noise rows)
        const long long R = img_res;
        const long long p0 = ((positions[2 * n + 0] % R) + R) % R, p1 = ((positions[2 * n + 1] % R) + R) % R;
        np0 = (float)p0 / (float)(img_res - 1);
        np1 = (float)p1 / (float)(img_res - 1);
    } else if (norm_pos) {
        np0 = norm_pos[2 * n + 0];
        np1 = norm_pos[2 * n + 1];
    }
    for (int idx = idx0; idx < r * r; idx += stride) {
        const int i = idx / r, j = idx - i * r;
        // grid[i,j] = (lin[i] + pos0, lin[j] + pos1); channel 0 is the COLUMN coordinate, channel 1 the ROW
        const float g0 = nb_fmod1(L.noise_lin[i] + np0) * 2.f - 1.f;
        const float g1 = nb_fmod1(L.noise_lin[j] + np1) * 2.f - 1.f;
        const float cx = ((g0 + 1.f) / 2.f) * (float)(r - 1);
        const float cy = ((g1 + 1.f) / 2.f) * (float)(r - 1);
        const float x0 = floorf(cx), y0 = floorf(cy);
        const int x0i = (int)x0, y0i = (int)y0, x1i = x0i + 1, y1i = y0i + 1;
        const float wx1 = cx - x0, wy1 = cy - y0, wx0 = (x0 + 1.f) - cx, wy0 = (y0 + 1.f) - cy;
        auto tap = [&](int yi, int xi) -> float {
            return (yi >= 0 && yi < r && xi >= 0 && xi < r) ? L.noise_const[yi * r + xi] : 0.f;
        };
        float v = tap(y0i, x0i) * (wx0 * wy0);
        v += tap(y0i, x1i) * (wx1 * wy0);
        v += tap(y1i, x0i) * (wx0 * wy1);
        v += tap(y1i, x1i) * (wx1 * wy1);
        L.noise_out[(size_t)n * r * r + idx] = v * strength;
    }
}

// The sampling grid is separable and TRANSPOSING -- the source column depends on the output row i, the source row on the
// output column j (SURVEY note C) -- so a straight gather has either its loads or its stores strided by a whole image row.
// A block takes a 32 x 32 output tile of one (layer, sample): the bilinear parameters of its 32 rows and 32 columns are
// computed once, the 32 x 2 x 32 x 2 source taps are fetched with lanes along the source COLUMN (coalesced) into LDS, and
// the outputs are formed with lanes along the output column (coalesced stores, conflict-free 8-byte LDS reads).  Same
// expressions in the same order per pixel as nb_noise_sample (bit-identical).
#define NB_NOISE_T 32
__global__ __launch_bounds__(256) void noise_kernel(const NbLayerDesc* __restrict__ layers, const float* __restrict__ norm_pos,
                                                    const long long* __restrict__ positions, int img_res, int n_total) {
    const NbLayerDesc L = layers[blockIdx.y];
    if (!L.noise_const) return;
    const int r = L.res;
    const int tiles = (r + NB_NOISE_T - 1) / NB_NOISE_T;
    const int t = threadIdx.x;
    // a block walks tiles blockIdx.x, + gridDim.x, ... (few blocks per (layer, sample): most layers have 1-4 tiles, and
    // tens of thousands of empty blocks cost more than the work)
    for (int tile = blockIdx.x; tile < tiles * tiles; tile += gridDim.x) {
    const int i0 = (tile / tiles) * NB_NOISE_T, j0 = (tile % tiles) * NB_NOISE_T;
    __syncthreads();                                         // (the previous tile's LDS tables are done with)
    if (!norm_pos && !positions) {
        if (blockIdx.z == 0)
            for (int e = t; e < NB_NOISE_T * NB_NOISE_T; e += 256) {
                const int i = i0 + e / NB_NOISE_T, j = j0 + e % NB_NOISE_T;
                if (i < r && j < r) L.noise_out[i * r + j] = L.noise_const[i * r + j] * L.noise_strength[0];
            }
        continue;
    }
    const int n = blockIdx.z;
    const float strength = L.noise_strength[0];
    float np0 = 0.f, np1 = 0.f;
    if (positions) {
        // networks_modified.py:351-353: (positions % R) / (R - 1), python-style modulo, IEEE float32 division
        const long long R = img_res;
        const long long p0 = ((positions[2 * n + 0] % R) + R) % R, p1 = ((positions[2 * n + 1] % R) + R) % R;
        np0 = (float)p0 / (float)(img_res - 1);
        np1 = (float)p1 / (float)(img_res - 1);
    } else {
        np0 = norm_pos[2 * n + 0];
        np1 = norm_pos[2 * n + 1];
    }
    constexpr int PITCH = 2 * NB_NOISE_T * 2 + 2;            // floats per output column jj: [dy 2][ii 32][dx 2] (+2: bank spread)
    __shared__ int s_x0[NB_NOISE_T], s_y0[NB_NOISE_T];
    __shared__ float s_wx0[NB_NOISE_T], s_wx1[NB_NOISE_T], s_wy0[NB_NOISE_T], s_wy1[NB_NOISE_T];
    __shared__ __attribute__((aligned(8))) float s_tap[NB_NOISE_T * PITCH];
    if (t < 2 * NB_NOISE_T) {
        // grid[i,j] = (lin[i] + pos0, lin[j] + pos1); channel 0 is the COLUMN coordinate, channel 1 the ROW
        const int k = t & (NB_NOISE_T - 1), idx = (t < NB_NOISE_T ? i0 : j0) + k;
        if (idx < r) {
            const float g = nb_fmod1(L.noise_lin[idx] + (t < NB_NOISE_T ? np0 : np1)) * 2.f - 1.f;
            const float cc = ((g + 1.f) / 2.f) * (float)(r - 1);
            const float c0 = floorf(cc);
            if (t < NB_NOISE_T) { s_x0[k] = (int)c0; s_wx1[k] = cc - c0; s_wx0[k] = (c0 + 1.f) - cc; }
            else { s_y0[k] = (int)c0; s_wy1[k] = cc - c0; s_wy0[k] = (c0 + 1.f) - cc; }
        } else if (t < NB_NOISE_T) { s_x0[k] = -2; s_wx0[k] = s_wx1[k] = 0.f; }
        else { s_y0[k] = -2; s_wy0[k] = s_wy1[k] = 0.f; }
    }
    __syncthreads();
    // taps: lanes along the source column (output row ii)
    for (int e = t; e < NB_NOISE_T * 2 * NB_NOISE_T * 2; e += 256) {
        const int dx = e & 1, ii = (e >> 1) & (NB_NOISE_T - 1), dy = (e >> 6) & 1, jj = e >> 7;
        const int xi = s_x0[ii] + dx, yi = s_y0[jj] + dy;
        s_tap[jj * PITCH + (dy * NB_NOISE_T + ii) * 2 + dx] = (yi >= 0 && yi < r && xi >= 0 && xi < r) ? L.noise_const[yi * r + xi] : 0.f;
    }
    __syncthreads();
    const int jj = t & (NB_NOISE_T - 1);
    if (j0 + jj >= r) continue;
    const float wy0 = s_wy0[jj], wy1 = s_wy1[jj];
    for (int ii = t >> 5; ii < NB_NOISE_T && i0 + ii < r; ii += 8) {
        const float wx0 = s_wx0[ii], wx1 = s_wx1[ii];
        const float2 r0 = *reinterpret_cast<const float2*>(s_tap + jj * PITCH + ii * 2);
        const float2 r1 = *reinterpret_cast<const float2*>(s_tap + jj * PITCH + (NB_NOISE_T + ii) * 2);
        float v = r0.x * (wx0 * wy0);
        v += r0.y * (wx1 * wy0);
        v += r1.x * (wx0 * wy1);
        v += r1.y * (wx1 * wy1);
        L.noise_out[(size_t)n * r * r + (size_t)(i0 + ii) * r + j0 + jj] = v * strength;
    }
    }
}

extern "C" int nb_noise_f32(const NbLayerDesc* layers_dev, int n_layers, int max_res, const float* norm_pos,
                            const int64_t* positions, int img_resolution, int n, void* stream) {
    NB_REQUIRE(layers_dev, "noise: null pointer");
    NB_REQUIRE(n_layers >= 1 && n_layers <= 65535 && max_res >= 1 && n >= 1 && n <= 65535, "noise: bad sizes");
    NB_REQUIRE(!(norm_pos && positions), "noise: pass either norm_pos or positions, not both");
    NB_REQUIRE(!positions || img_resolution >= 2, "noise: positions need img_resolution >= 2");
    const int tiles = nb_cdiv(max_res, NB_NOISE_T);
    dim3 grid(tiles * tiles, n_layers, (norm_pos || positions) ? n : 1);
    hipLaunchKernelGGL(noise_kernel, grid, dim3(256), 0, (hipStream_t)stream, layers_dev, norm_pos,
                       (const long long*)positions, img_resolution, n);
    NB_CHECK_LAUNCH("noise");
    return NB_OK;
}

// Integer patch positions -> the normalised float32 positions the noise arithmetic starts from (nb_noise_np: python-style modulo,
// correctly rounded division), once per batch.  The convolutions that compute their noise themselves (NbNoiseSrc) evaluate
// nb_noise_np at the top of EVERY tile -- with `positions` that is four 64-bit modulo operations per lane, ~1 us of a 23-33 us tile
// (44 us of a 1.93 ms step at batch 32, R=256); with `norm_pos` two loads.  Same function, same result bits.
__global__ __launch_bounds__(64) void norm_positions_kernel(const long long* __restrict__ positions, int img_res, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const NbNoiseSrcDev s{nullptr, nullptr, nullptr, nullptr, positions, 0, img_res};
    float np0, np1;
    nb_noise_np(s, i, np0, np1);
    out[2 * i] = np0; out[2 * i + 1] = np1;
}

extern "C" int nb_norm_positions_f32(const int64_t* positions, int img_resolution, float* norm_pos_out, int n, void* stream) {
    NB_REQUIRE(positions && norm_pos_out && n >= 1 && img_resolution >= 2, "norm_positions: bad arguments");
    hipLaunchKernelGGL(norm_positions_kernel, dim3(nb_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, (const long long*)positions, img_resolution, norm_pos_out, n);
    NB_CHECK_LAUNCH("norm_positions");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// triad ToRGB epilogue (networks.py:451-485) + paint-engine compositing (forger/ui/brush.py:763-792)
// HBM-bound: reads x once (16 B per lane per channel), writes the 3-channel results.
// ------------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void torgb_triad_kernel(const TorgbParams p) {
    extern __shared__ float sw[];          // [3][c] modulated weights, then 9 colors, 9 col01
    float* scol = sw + 3 * p.c;
    float* scol01 = scol + 9;
    const int n = blockIdx.y, t = threadIdx.x;
    const float* st = p.styles + (size_t)n * p.styles_stride_n;
    for (int i = t; i < 3 * p.c; i += 256) {
        const int o = i / p.c, ch = i - o * p.c;
        sw[i] = p.w[o * p.c + ch] * st[9 + ch];
    }
    if (t < 9) {
        const float col = tanhf(st[t] + p.color_bias[t]);
        scol[t] = col;
        float c01 = (col + 1.f) / 2.f;
        if (p.user_colors) {
            const float u = p.user_colors[n * 9 + t];
            if (!(u != u)) c01 = u;
        }
        scol01[t] = c01;
        if (p.colors_out && blockIdx.x == 0) p.colors_out[n * 9 + t] = col;
    }
    __syncthreads();
    const int pix = (blockIdx.x * 256 + t) * V;
    if (pix >= p.hw) return;
    // the summation order of nb_torgb_dot (nb_torgb.h), V pixels at a time
    float a0[V], a1[V], a2[V];
    const float* xp = p.x + (size_t)n * p.c * p.hw + pix;
    {
        float s[3][2][4][V];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int v = 0; v < V; ++v) s[k][h][j][v] = 0.f;
        for (int mg = 0; mg * 8 < p.c; ++mg) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = mg * 8 + 4 * h + j;
                    if (ch < p.c) {
                        float xv[V];
                        if constexpr (V == 4) {
                            const f32x4 q = *reinterpret_cast<const f32x4*>(xp + (size_t)ch * p.hw);
                            xv[0] = q[0]; xv[1] = q[1]; xv[2] = q[2]; xv[3] = q[3];
                        } else {
                            xv[0] = xp[(size_t)ch * p.hw];
                        }
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float wk = sw[k * p.c + ch];
#pragma unroll
                            for (int v = 0; v < V; ++v) s[k][h][j][v] = __builtin_fmaf(xv[v], wk, s[k][h][j][v]);
                        }
                    }
                }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float a[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                a[k] = ((s[k][0][0][v] + s[k][0][1][v]) + (s[k][0][2][v] + s[k][0][3][v])) + ((s[k][1][0][v] + s[k][1][1][v]) + (s[k][1][2][v] + s[k][1][3][v]));
            a0[v] = a[0]; a1[v] = a[1]; a2[v] = a[2];
        }
    }
    const float b0 = p.bias[0], b1 = p.bias[1], b2 = p.bias[2];
    const float sf = p.sfactor ? p.sfactor[n] : 0.f;
    float lg[3][V], uv[3][V], im[3][V], rg[4][V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        float l0 = a0[j] + b0, l1 = a1[j] + b1, l2 = a2[j] + b2;
        if (p.clamp >= 0.f) {
            l0 = fminf(fmaxf(l0, -p.clamp), p.clamp); l1 = fminf(fmaxf(l1, -p.clamp), p.clamp); l2 = fminf(fmaxf(l2, -p.clamp), p.clamp);
        }
        const float m = fmaxf(l0, fmaxf(l1, l2));
        const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
        const float inv = 1.f / (e0 + e1 + e2);
        const float u = e0 * inv, v = e1 * inv, s = e2 * inv;
        lg[0][j] = l0; lg[1][j] = l1; lg[2][j] = l2;
        uv[0][j] = u; uv[1][j] = v; uv[2][j] = s;
        // StyleUVSMapper._map_style_s (forger/ui/mapper.py:52-72): stretch the background weight S so that clear
        // background becomes fully transparent, rescale U, V to keep the triple on the simplex
        float um = u, vm = v, sm = s;
        if (p.sfactor) {
            sm = fminf(sf * s, 1.f);
            const float delta = 1.f - sm;
            const float f = delta <= 0.000001f ? 0.f : delta / (u + v);
            um = f * u; vm = f * v;
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            im[ch][j] = u * scol[ch * 3 + 0] + v * scol[ch * 3 + 1] + s * scol[ch * 3 + 2];
            rg[ch][j] = um * scol01[ch * 3 + 0] + vm * scol01[ch * 3 + 1] + sm * scol01[ch * 3 + 2];
        }
        rg[3][j] = p.render_mode == 0 ? um + vm : 1.f;
    }
    auto put = [&](float* base, int nch, int ch, const float (&vals)[V]) {
        float* dst = base + ((size_t)n * nch + ch) * p.hw + pix;
        if constexpr (V == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{vals[0], vals[1], vals[2], vals[3]};
        else dst[0] = vals[0];
    };
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        if (p.logits) put(p.logits, 3, ch, lg[ch]);
        if (p.uvs) put(p.uvs, 3, ch, uv[ch]);
        if (p.img) put(p.img, 3, ch, im[ch]);
    }
    if (p.rgba_f32) {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) put(p.rgba_f32, 4, ch, rg[ch]);
    }
    if (p.rgba_u8) {
        uint32_t pk[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            pk[j] = 0;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const float q = fminf(fmaxf(rg[ch][j] * 255.f, 0.f), 255.f);   // (x*255).clip(0,255).to(uint8): truncation
                pk[j] |= ((uint32_t)q & 0xffu) << (8 * ch);
            }
        }
        uint32_t* dst = reinterpret_cast<uint32_t*>(p.rgba_u8) + (size_t)n * p.hw + pix;    // [n][hw][4] bytes (HWC)
        if constexpr (V == 4) *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        else dst[0] = pk[0];
    }
}

extern "C" int nb_torgb_triad_f32(const float* x, const float* styles, int styles_stride_n, const float* w,
                                  const float* bias, const float* color_bias, float clamp, float* logits, float* uvs,
                                  float* img, float* colors_out, const float* user_colors, const float* sfactor,
                                  int render_mode, float* rgba_f32, uint8_t* rgba_u8, int n, int c, int hw, void* stream) {
    NB_REQUIRE(x && styles && w && bias && color_bias, "torgb_triad: null pointer");
    NB_REQUIRE(n >= 1 && n <= 65535 && c >= 1 && c <= 4096 && hw >= 1, "torgb_triad: bad sizes");
    NB_REQUIRE(styles_stride_n >= c + 9, "torgb_triad: styles rows must hold 9 color scalars + c styles");
    NB_REQUIRE(render_mode == 0 || render_mode == 1, "Unknown render mode for TriadGanPaintEngine: %d", render_mode);
    TorgbParams p;
    p.x = x; p.styles = styles; p.w = w; p.bias = bias; p.color_bias = color_bias; p.logits = logits; p.uvs = uvs; p.img = img;
    p.colors_out = colors_out; p.user_colors = user_colors; p.sfactor = sfactor; p.rgba_f32 = rgba_f32; p.rgba_u8 = rgba_u8;
    p.styles_stride_n = styles_stride_n; p.c = c; p.hw = hw; p.render_mode = render_mode; p.clamp = clamp;
    const size_t lds = (size_t)(3 * c + 18) * sizeof(float);
    const bool vec = (hw % 4 == 0) && ((uintptr_t)x % 16 == 0);
    if (vec) {
        dim3 grid(nb_cdiv(hw, 1024), n);
        hipLaunchKernelGGL(torgb_triad_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, p);
    } else {
        dim3 grid(nb_cdiv(hw, 256), n);
        hipLaunchKernelGGL(torgb_triad_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, p);
    }
    NB_CHECK_LAUNCH("torgb_triad");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// feature blending (forger/train/stitching.py:24-25)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ feat, int nf, const float* __restrict__ alpha,
                                                    int na, const float* __restrict__ x, float* __restrict__ y,
                                                    int c, int hw, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int pix = (int)(i % hw);
        const long long nc = i / hw;
        const int ch = (int)(nc % c);
        const int n = (int)(nc / c);
        const float a = alpha[(size_t)(na == 1 ? 0 : n) * hw + pix];
        const float f = feat[((size_t)(nf == 1 ? 0 : n) * c + ch) * hw + pix];
        y[i] = a * f + (1.f - a) * x[i];
    }
}

extern "C" int nb_blend_f32(const float* features, int nf, const float* alpha, int na, const float* x, float* y, int n,
                            int c, int hw, void* stream) {
    NB_REQUIRE(features && alpha && x && y, "blend: null pointer");
    NB_REQUIRE(n >= 1 && c >= 1 && hw >= 1, "blend: bad sizes");
    NB_REQUIRE((nf == 1 || nf == n) && (na == 1 || na == n), "blend: features/alpha batch must be 1 or n");
    const long long total = (long long)n * c * hw;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(blend_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, features, nf, alpha, na, x, y, c, hw, total);
    NB_CHECK_LAUNCH("blend");
    return NB_OK;
}
