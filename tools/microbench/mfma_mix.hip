// Microbenchmark: sustained rate of the split-f16 inner loop (6 x v_mfma_f32_32x32x16_f16 per 32 channels) against a
// variant whose two correction products ride on ONE fp8 MFMA (2 x f16 + 1 x v_mfma_f32_32x32x64_f8f6f4), operands in
// registers, 2 waves per SIMD, whole chip.  Answers: what would fp8 corrections buy under the power limit?
//   hipcc -O3 --offload-arch=gfx950 mfma_mix.hip -o mfma_mix && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    i32x8 a8, b8;
    for (int j = 0; j < 8; ++j) { a8[j] = 0x38383838 + threadIdx.x; b8[j] = 0x3c3c3c3c; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            } else {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, iters, clk);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[256]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
        // per iteration and wave: 4 accumulators x 32 channels of a 32x32 tile: algorithmic 4*2*32*32*32 FLOP
        const double alg = (double)iters * 4 * 2 * 32 * 32 * 32 * 256 * 8;
        printf("mode %d (%s): %.3f ms, %.1f algorithmic TFLOP/s (fp32-equivalent), shader clock %.2f GHz, cycles/iter/wave %.1f\n", mode,
               mode == 0 ? "6 x f16 MFMA" : "2 x f16 + 1 x fp8(K=64) MFMA", ms, alg / ms / 1e9, c[0] / (ms * 1e6), (double)c[0] / iters);
    }
    return 0;
}
