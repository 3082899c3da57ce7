// Microbenchmark: f16 MFMA rate when the fragments come from LDS (ds_read_b128), at the read:MFMA ratios of the
// split-f16 conv kernels (8 reads : 12 MFMAs per tap today; 12 : 24 with 2x larger wave tiles), 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 mfma_lds.hip -o mfma_lds && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NREAD, int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ h8 buf[4096];                                     // 64 KiB
    for (int i = threadIdx.x; i < 4096; i += 512) for (int j = 0; j < 8; ++j) buf[i][j] = (_Float16)(i * 0.001f + j);
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int base = wv * 64 + lane;
    h8 f[NREAD];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NREAD; ++q) f[q] = buf[(base + q * 512 + it * 64) & 4095];
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[(i + q) % NREAD], f[(i + 2 * q + 1) % NREAD], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NREAD, int NACC>
void run(float* out, const char* what) {
    const int iters = 30000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NREAD, NACC>), dim3(256), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double mf = (double)iters * NACC * 3 * 256 * 8 * 2.0 * 32 * 32 * 16;
    printf("%-44s %7.3f ms  %7.1f TFLOP/s executed f16 (%.0f%% of 2500), LDS %5.1f TB/s\n", what, ms, mf / ms / 1e9, mf / ms / 1e9 / 25.0,
           (double)iters * NREAD * 256 * 8 * 1024 / ms / 1e9);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<1, 4>(out, "1 read : 12 MFMA (register resident)");
    run<8, 4>(out, "8 reads : 12 MFMA (today, 64x64 wave tile)");
    run<12, 8>(out, "12 reads : 24 MFMA (64x128 wave tile)");
    run<6, 4>(out, "6 reads : 12 MFMA");
    run<4, 4>(out, "4 reads : 12 MFMA");
    return 0;
}
