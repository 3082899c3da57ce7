// Sustained rate of v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 (and the K = 64 / K = 128 fp8 forms) on random
// operands, whole chip, 2 waves per SIMD, registers only: which shape delivers more FLOP/s under the power limit?
//   hipcc -O3 --offload-arch=gfx950 mfma_shapes.hip -o mfma_shapes && ./mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const h8* ops, float* out, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ops[(2 * i) * 64 + lane]; b[i] = ops[(2 * i + 1) * 64 + lane]; }
    i32x8 a8[2], b8[2];
    for (int i = 0; i < 2; ++i) { a8[i] = __builtin_bit_cast(i32x8, __builtin_shufflevector(a[2 * i], a[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
                                  b8[i] = __builtin_bit_cast(i32x8, __builtin_shufflevector(b[2 * i], b[2 * i + 1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)); }
    float s = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + q) & 3], b[q], acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else if (MODE == 1) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + q) & 3], b[q], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    } else if (MODE == 2) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[(i + q) & 1], b8[q], acc[i], 0, 0, 0, 127, 0, 127);
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[(i + q) & 1], b8[q], acc[i], 0, 0, 0, 127, 0, 127);
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    h8* ops; float* out; unsigned long long* clk;
    hipMalloc(&ops, 8 * 64 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    _Float16 h[8 * 64 * 8];
    srand(5);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    hipMemcpy(ops, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    const char* names[4] = {"32x32x16 f16", "16x16x32 f16", "32x32x64 fp8 (block-scaled)", "16x16x128 fp8 (block-scaled)"};
    const double macs[4] = {16.0 * 32 * 32 * 16, 32.0 * 16 * 16 * 32, 8.0 * 32 * 32 * 64, 16.0 * 16 * 16 * 128};      // per wave and iteration
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, ops, out, iters, clk); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, ops, out, iters, clk); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, ops, out, iters, clk); break;
                default: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, ops, out, iters, clk); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[256]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
            const double fl = 2.0 * macs[mode] * iters * 256 * 8;
            if (rep) printf("%-30s %.3f ms  %.0f TFLOP/s  shader clock %.2f GHz\n", names[mode], ms, fl / ms / 1e9, c[0] / (ms * 1e6));
        }
    return 0;
}
