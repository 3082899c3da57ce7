// Shared helpers for the gfx950 NeuBE kernels (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/neube_hip.h"

#define NB_ABI_VERSION 11

// Kernels whose scalar fp32 arithmetic the SLP vectoriser pairs into packed instructions are compiled WITHOUT packed fp32 ops.
// Reason: the pairing produces forms that swizzle register halves -- `v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]`,
// `v_pk_mul_f32 ... op_sel:[1,0] op_sel_hi:[0,1]` -- and on MI355X those returned, sporadically and only while waves of another
// kernel shared the SIMD, a result computed from the OTHER half of the pair for 16-32 lanes of a wave: first seen in the up=2
// split-f16 epilogue (round 3, tools/determinism_up2.py), then as wrong `img` / RGBA values of torgb_triad_kernel<4> (element 2 of a
// thread's 4 pixels, one channel) in ~1 of 6 rounds of tools/gen_race_hunt.py.  Waits forced to zero and s_nops between the
// producer and the packed op did not help; not emitting the form does.  Kernels that use packed fp32 on purpose (explicit
// f32x4 / f32x2 arithmetic in the split-f16 epilogues) only ever use the un-swizzled forms and keep them.  The attribute
// below is applied to enc_upsample2x_h2_kernel; nb_ops.hip (ToRGB, bias_act, ...) and nb_modconv.hip (exact-fp32 conv kernels: 4-8
// swizzled adds in the up=2 FIR code, never seen to misbehave) are compiled without SLP vectorisation instead (build.py FILE_FLAGS),
// which removes the swizzled forms without costing scratch.
#define NB_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))

void nb_set_error(const char* fmt, ...);

#define NB_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) {                                \
            nb_set_error(__VA_ARGS__);                \
            return NB_EINVAL;                         \
        }                                             \
    } while (0)

#define NB_CHECK_LAUNCH(what)                                                      \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            nb_set_error("%s: launch failed: %s", what, hipGetErrorString(e_));    \
            return NB_ELAUNCH;                                                     \
        }                                                                          \
    } while (0)

static inline int nb_cdiv(int a, int b) { return (a + b - 1) / b; }

// developer / test hooks (include/neube_hip_debug.h; defined in nb_ops.hip, both 0 on the product path): feature-off bit mask of the conv
// kernels (p.dbg) and their first-round stagger in s_sleep ticks
extern int g_nb_debug_flags, g_nb_stagger_ticks;

// ---- position-shifted constant noise (networks.py:371-382), the per-pixel arithmetic shared by nb_noise_f32's kernels
//      (nb_ops.hip) and the convolutions that compute their noise themselves (NbNoiseSrc) ----
#ifdef __HIPCC__
// fmodf(a, 1.f) without the library's general-divisor loop: a - trunc(a) is exact in fp32 for every finite a (the fraction is
// the low bits of a; |a| >= 2^23 is an integer) and has a's sign, which is what fmod returns (the two differ only in the sign
// of a zero result, which the `* 2 - 1` that follows erases).  ~40 VALU instructions per call less in the conv prologues.
__device__ __forceinline__ float nb_fmod1(float a) { return a - truncf(a); }

struct NbNoiseSrcDev {
    const float* const_t; const float* lin; const float* strength; const float* norm_pos; const long long* positions;
    int res, img_res;
};
// normalised position of sample n: (positions % R) / (R - 1), python-style modulo, IEEE float32 division
__device__ __forceinline__ void nb_noise_np(const NbNoiseSrcDev& s, int n, float& np0, float& np1) {
    if (s.positions) {
        const long long R = s.img_res;
        const long long p0 = ((s.positions[2 * n + 0] % R) + R) % R, p1 = ((s.positions[2 * n + 1] % R) + R) % R;
        np0 = (float)p0 / (float)(s.img_res - 1);
        np1 = (float)p1 / (float)(s.img_res - 1);
    } else {
        np0 = s.norm_pos[2 * n + 0];
        np1 = s.norm_pos[2 * n + 1];
    }
}
// bilinear parameters of one axis: output index idx -> first source index c0 and the two weights
__device__ __forceinline__ void nb_noise_axis(const NbNoiseSrcDev& s, int idx, float np, int& c0, float& w0, float& w1) {
    const float g = nb_fmod1(s.lin[idx] + np) * 2.f - 1.f;
    const float cc = ((g + 1.f) / 2.f) * (float)(s.res - 1);
    const float f0 = floorf(cc);
    c0 = (int)f0; w1 = cc - f0; w0 = (f0 + 1.f) - cc;
}
// noise of output pixel (row i, column j): the SOURCE COLUMN parameters (x0, wx0, wx1) come from row i, the source row
// parameters (y0, wy0, wy1) from column j (the reference's grid transposes); const_t[x * res + y] = noise_const[y, x]
__device__ __forceinline__ float nb_noise_value(const NbNoiseSrcDev& s, float strength, int x0, float wx0, float wx1, int y0, float wy0, float wy1) {
    const int r = s.res;
    auto tap = [&](int yi, int xi) -> float { return (yi >= 0 && yi < r && xi >= 0 && xi < r) ? s.const_t[xi * r + yi] : 0.f; };
    float v = tap(y0, x0) * (wx0 * wy0);
    v += tap(y0, x0 + 1) * (wx1 * wy0);
    v += tap(y0 + 1, x0) * (wx0 * wy1);
    v += tap(y0 + 1, x0 + 1) * (wx1 * wy1);
    return v * strength;
}
#endif
