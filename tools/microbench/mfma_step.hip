// Microbenchmark: the K-loop skeleton of modconv3x3_up1_h3_kernel without any global traffic -- per step 3 taps x
// (8 fragment reads + 12 MFMAs) and one workgroup barrier -- against variants: no barrier; first fragments of the
// next step fetched BEFORE the barrier; one barrier per 3 steps.  8 waves, 2 per SIMD, whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int steps) {
    __shared__ h8 buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) for (int j = 0; j < 8; ++j) buf[i][j] = (_Float16)(i * 0.001f + j);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int base = wv * 64 + lane;
    h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
    auto fetch = [&](int t, int kx, int cu) {
        const int o = (base + t * 192 + kx * 64) & 2047;
        ah[cu][0] = buf[o]; ah[cu][1] = buf[o + 32]; al[cu][0] = buf[o + 512]; al[cu][1] = buf[o + 544];
        bh[cu][0] = buf[o + 2048]; bh[cu][1] = buf[(o + 2082) & 4095]; bl[cu][0] = buf[(o + 3072) & 4095]; bl[cu][1] = buf[(o + 3106) & 4095];
    };
    if (MODE == 2) fetch(0, 0, 0);
    for (int t = 0; t < steps; ++t) {
        if (MODE != 2) fetch(t, 0, 0);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int cu = kx & 1;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cu][0], bh[cu][0], acc[0][0], 0, 0, 0);
            if (kx + 1 < 3) fetch(t, kx + 1, cu ^ 1);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    if (mb + nb > 0) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cu][mb], bh[cu][nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cu][mb], bl[cu][nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cu][mb], bh[cu][nb], acc[mb][nb], 0, 0, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (kx + 1 < 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 2) fetch(t + 1, 0, 1);            // kx = 2 used buffer 0; next step's kx = 0 goes to buffer 1
        if (MODE == 0 || MODE == 2 || (MODE == 3 && t % 3 == 2)) __builtin_amdgcn_s_barrier();
        if (MODE == 2) {                               // rotate: next step starts from buffer 1 -> copy (register moves are free here)
#pragma unroll
            for (int m = 0; m < 2; ++m) { ah[0][m] = ah[1][m]; al[0][m] = al[1][m]; bh[0][m] = bh[1][m]; bl[0][m] = bl[1][m]; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i >> 1][i & 1][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, const char* what) {
    const int steps = 12000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, steps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double mf = (double)steps * 36 * 256 * 8 * 2.0 * 32 * 32 * 16;
    printf("%-58s %7.3f ms  %7.1f TFLOP/s executed (%.0f%% of 2500)\n", what, ms, mf / ms / 1e9, mf / ms / 1e9 / 25.0);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<0>(out, "barrier every step (today)");
    run<1>(out, "no barrier");
    run<2>(out, "next step's first fragments fetched before the barrier");
    run<3>(out, "barrier every 3rd step");
    return 0;
}
