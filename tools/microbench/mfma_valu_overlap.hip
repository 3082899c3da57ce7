// Microbenchmark: does a VALU-bound epilogue overlap with an MFMA-dense K loop on the same CU, or does the power limit
// give the time back?  One 8-wave workgroup per CU, random f16 operands (data toggling matters for power).
//   mode 0: waves 0-3 run the MFMA loop, waves 4-7 exit                (matrix work alone, one wave per SIMD)
//   mode 1: waves 4-7 run the packed-fp32 FMA loop, waves 0-3 exit     (vector work alone, one wave per SIMD)
//   mode 2: both at once                                                (wave-specialised workgroup)
//   mode 3: all 8 waves run half the MFMA iterations, then half the VALU iterations (today's kernel structure)
//   hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mfma_loop(const h8* ops, int iters, int lane) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ops[(i * 2) * 64 + lane]; b[i] = ops[(i * 2 + 1) * 64 + lane]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + q) & 3], b[q], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    return s;
}

__device__ __forceinline__ float valu_loop(const float* seed, int iters, int tid) {
    f32x2 v[16];
    for (int i = 0; i < 16; ++i) { v[i][0] = seed[(tid * 32 + 2 * i) & 4095]; v[i][1] = seed[(tid * 32 + 2 * i + 1) & 4095]; }
    const f32x2 m = {0.999f, 1.001f}, c = {0.01f, -0.01f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c + v[(i + 1) & 15] * 0.f);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
    return s;
}

// one wave interleaves: after every MFMA, NV independent packed-fp32 instructions (the software-pipelined form: the
// previous tile's epilogue arithmetic inside this tile's K loop)
template <int NV>
__device__ __forceinline__ float mixed_loop(const h8* ops, const float* seed, int iters, int tid) {
    const int lane = tid & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ops[(i * 2) * 64 + lane]; b[i] = ops[(i * 2 + 1) * 64 + lane]; }
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) { v[i][0] = seed[(tid * 32 + 2 * i) & 4095]; v[i][1] = seed[(tid * 32 + 2 * i + 1) & 4095]; }
    const f32x2 m = {0.999f, 1.001f}, c = {0.01f, -0.01f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + q) & 3], b[q], acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_elementwise_fma(v[j & 7], m, c);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    return s;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const h8* ops, const float* seed, float* out, int mi, int vi, unsigned long long* clk, int prio) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (MODE >= 4) {
        s = MODE == 4 ? mixed_loop<0>(ops, seed, mi / 2, tid) : MODE == 5 ? mixed_loop<2>(ops, seed, mi / 2, tid) : MODE == 6 ? mixed_loop<4>(ops, seed, mi / 2, tid) : mixed_loop<8>(ops, seed, mi / 2, tid);
    } else if (MODE == 3) {
        s = mfma_loop(ops, mi / 2, lane);
        __syncthreads();
        s += valu_loop(seed, vi / 2, tid);
    } else if (wv < 4) {
        if (MODE == 0 || MODE == 2) s = mfma_loop(ops, mi, lane);
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        if (MODE == 1 || MODE == 2) s = valu_loop(seed, vi, tid);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 || tid == 256) clk[blockIdx.x * 2 + (tid >> 8)] = t1 - t0;
}

int main(int argc, char** argv) {
    const int mi = argc > 1 ? atoi(argv[1]) : 20000, vi = argc > 2 ? atoi(argv[2]) : 20000, prio = argc > 3 ? atoi(argv[3]) : 0;
    h8* ops; float *seed, *out; unsigned long long* clk;
    hipMalloc(&ops, 8 * 64 * 16); hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 16);
    _Float16 h[8 * 64 * 8]; float hs[4096];
    srand(3);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    for (auto& x : hs) x = rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(ops, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
    const char* names[8] = {"MFMA alone (4 waves)", "VALU alone (4 waves)", "MFMA waves + VALU waves at once", "8 waves: MFMA phase then VALU phase", "8 waves: MFMA only (half iters each)", "8 waves: MFMA + 2 pk VALU each", "8 waves: MFMA + 4 pk VALU each", "8 waves: MFMA + 8 pk VALU each"};
    for (int mode = 0; mode < 8; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, ops, seed, out, mi, vi, clk, prio); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, ops, seed, out, mi, vi, clk, prio); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, ops, seed, out, mi, vi, clk, prio); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, ops, seed, out, mi, vi, clk, prio); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, ops, seed, out, mi, vi, clk, prio); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, ops, seed, out, mi, vi, clk, prio); break;
                case 7: hipLaunchKernelGGL(k<7>, dim3(256), dim3(512), 0, 0, ops, seed, out, mi, vi, clk, prio); break;
                default: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, ops, seed, out, mi, vi, clk, prio); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[512]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
        printf("mode %d %-40s %.3f ms   wave 0: %.0f Mcycles (%.2f GHz if it ran the whole launch), wave 4: %.0f Mcycles\n", mode, names[mode], ms,
               c[0] / 1e6, c[0] / (ms * 1e6), c[1] / 1e6);
    }
    return 0;
}
