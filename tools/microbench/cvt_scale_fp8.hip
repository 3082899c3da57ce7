// Does v_cvt_scalef32_pk_fp8_f32 (gfx950) equal "multiply by a power of two, then v_cvt_pk_fp8_f32" bit for bit -- and with which
// meaning of its scale operand?  The f8 epilogues spend two packed multiplies per four values on exactly such scalings (x 512 for the
// residual, x 0.25 for the value).  Prints, for scale operands 2^k, the multiplier m for which the two agree on every input.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/cvt_scale_fp8.hip -o tools/microbench/bin/cvt_scale_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* a, unsigned* b, float scale, float mul, int ovfl, int n) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    float v0 = x[2 * i], v1 = x[2 * i + 1];
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v0 * mul, v1 * mul, w, false);
    a[i] = (unsigned)w & 0xffff;
    s16x2 o = {0, 0};
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, v0, v1, scale, false);
    b[i] = (unsigned)(unsigned short)o[0];
}
int main() {
    const int n = 1 << 20;
    std::vector<float> h(2 * n);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        const float m = (float)rand() / RAND_MAX * 2.f - 1.f;
        const int e = rand() % 40 - 28;                        // 2^-28 .. 2^11: denormal fp8 results up to saturation after x 512
        h[i] = ldexpf(m, e);
    }
    // every 4-bit mantissa at every exponent of interest, i.e. all e4m3 values AND all midpoints between neighbours (round-to-even ties),
    // pre-divided by both scalings so that the scaled value hits them exactly
    int at = 8;
    for (float mul : {512.f, 0.25f})
        for (int e = -12; e <= 9; ++e)
            for (int m = 0; m < 32; ++m)
                for (float sg : {1.f, -1.f}) h[at++] = sg * ldexpf(1.f + m / 32.f, e) / mul;
    h[0] = 0.f; h[1] = -0.f; h[2] = 1e30f; h[3] = -1e30f; h[4] = INFINITY; h[5] = NAN;
    float* dx; unsigned *da, *db;
    hipMalloc(&dx, 2 * n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, h.data(), 2 * n * 4, hipMemcpyHostToDevice);
    std::vector<unsigned> a(n), b(n);
    for (int ovfl = 0; ovfl < 2; ++ovfl)
        for (float mul : {512.f, 0.25f})
            for (float scale : {mul, 1.f / mul}) {
                hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, scale, mul, ovfl, n);
                hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
                int bad = 0, first = -1;
                for (int i = 0; i < n; ++i) if (a[i] != b[i]) { if (first < 0) first = i; ++bad; }
                printf("FP16_OVFL %d  multiply by %g then cvt  vs  cvt_scalef32(scale = %g): %d of %d pairs differ", ovfl, mul, scale, bad, n);
                if (first >= 0) printf("  (first: inputs %g %g -> %04x vs %04x)", h[2 * first], h[2 * first + 1], a[first], b[first]);
                printf("\n");
            }
    return 0;
}
