// Triad ToRGB epilogue shared by the standalone kernel (nb_ops.hip) and the fused conv1+ToRGB kernel
// (nb_modconv_h3.hip).  Reference: ToRGBColorTriadLayer.forward, training/networks.py:451-485; the RGBA compositing of
// TriadGanPaintEngine._render_stroke_torch, forger/ui/brush.py:763-792; StyleUVSMapper._map_style_s, mapper.py:52-72.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct TorgbParams {
    const float* x; const float* styles; const float* w; const float* bias; const float* color_bias;
    float* logits; float* uvs; float* img; float* colors_out; const float* user_colors; const float* sfactor; float* rgba_f32; uint8_t* rgba_u8;
    int styles_stride_n, c, hw, render_mode;
    float clamp;
};


// Per-sample constants: sw[3][c] = w * styles, scol[9] = tanh(affine + color_bias), scol01[9] = compositing colors.
// Call with every thread of the block, then synchronise.
__device__ __forceinline__ void nb_torgb_setup(const TorgbParams& p, int n, float* sw, float* scol, float* scol01, int t,
                                               int nthreads, bool write_colors) {
    const float* st = p.styles + (size_t)n * p.styles_stride_n;
    for (int i = t; i < 3 * p.c; i += nthreads) {
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
        if (p.colors_out && write_colors) p.colors_out[n * 9 + t] = col;
    }
}

// The 3 x c dot products of the 1x1 conv in ONE summation order, shared by every implementation (standalone kernel, LDS
// image of the fused kernel, accumulator registers of the fused kernel) so that they agree bit for bit.  The order follows
// the MFMA accumulator layout: channel ch = 32 m + 8 g + 4 h + j (h = lane half, j = register within a group of 4); the
// partial sums s[h][j] run over (m, g) ascending with fused multiply-adds, and
//   total = ((s[0][0] + s[0][1]) + (s[0][2] + s[0][3])) + ((s[1][0] + s[1][1]) + (s[1][2] + s[1][3])).
// x(ch) returns channel ch of the pixel; sw = [3][c] modulated weights.
template <typename GetX>
__device__ __forceinline__ void nb_torgb_dot(int c, const float* sw, GetX x, float& a0, float& a1, float& a2) {
    float s[3][2][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[k][h][j] = 0.f;
    for (int mg = 0; mg * 8 < c; ++mg) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = mg * 8 + 4 * h + j;
                if (ch < c) {
                    const float xv = x(ch);
#pragma unroll
                    for (int k = 0; k < 3; ++k) s[k][h][j] = __builtin_fmaf(xv, sw[k * c + ch], s[k][h][j]);
                }
            }
    }
    float a[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        a[k] = ((s[k][0][0] + s[k][0][1]) + (s[k][0][2] + s[k][0][3])) + ((s[k][1][0] + s[k][1][1]) + (s[k][1][2] + s[k][1][3]));
    a0 = a[0]; a1 = a[1]; a2 = a[2];
}

// One pixel: (a0, a1, a2) = the 1x1 modulated conv before the bias -> every requested output.
__device__ __forceinline__ void nb_torgb_pixel(const TorgbParams& p, int n, int pix, float a0, float a1, float a2,
                                               const float* scol, const float* scol01) {
    float l0 = a0 + p.bias[0], l1 = a1 + p.bias[1], l2 = a2 + p.bias[2];
    if (p.clamp >= 0.f) {
        l0 = fminf(fmaxf(l0, -p.clamp), p.clamp); l1 = fminf(fmaxf(l1, -p.clamp), p.clamp); l2 = fminf(fmaxf(l2, -p.clamp), p.clamp);
    }
    const float m = fmaxf(l0, fmaxf(l1, l2));
    const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
    const float inv = 1.f / (e0 + e1 + e2);
    const float u = e0 * inv, v = e1 * inv, s = e2 * inv;
    float um = u, vm = v, sm = s;
    if (p.sfactor) {
        sm = fminf(p.sfactor[n] * s, 1.f);
        const float delta = 1.f - sm;
        const float f = delta <= 0.000001f ? 0.f : delta / (u + v);
        um = f * u; vm = f * v;
    }
    const size_t b3 = (size_t)n * 3 * p.hw + pix, b4 = (size_t)n * 4 * p.hw + pix;
    if (p.logits) { p.logits[b3] = l0; p.logits[b3 + p.hw] = l1; p.logits[b3 + 2 * (size_t)p.hw] = l2; }
    if (p.uvs) { p.uvs[b3] = u; p.uvs[b3 + p.hw] = v; p.uvs[b3 + 2 * (size_t)p.hw] = s; }
    float rg[4];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        if (p.img) p.img[b3 + (size_t)ch * p.hw] = u * scol[ch * 3 + 0] + v * scol[ch * 3 + 1] + s * scol[ch * 3 + 2];
        rg[ch] = um * scol01[ch * 3 + 0] + vm * scol01[ch * 3 + 1] + sm * scol01[ch * 3 + 2];
    }
    rg[3] = p.render_mode == 0 ? um + vm : 1.f;
    if (p.rgba_f32) {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) p.rgba_f32[b4 + (size_t)ch * p.hw] = rg[ch];
    }
    if (p.rgba_u8) {
        uint32_t pk = 0;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const float q = fminf(fmaxf(rg[ch] * 255.f, 0.f), 255.f);
            pk |= ((uint32_t)q & 0xffu) << (8 * ch);
        }
        reinterpret_cast<uint32_t*>(p.rgba_u8)[(size_t)n * p.hw + pix] = pk;
    }
}
