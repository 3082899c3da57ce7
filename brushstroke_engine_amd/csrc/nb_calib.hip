// Box calibration: what THIS device sustains on the instruction the split-f16 conv kernels are built on.
//
// A registers-only loop of back-to-back v_mfma_f32_32x32x16_f16 on random operands, one wave per SIMD on every CU, for a
// few tens of milliseconds.  MI355X boards differ by several percent in the clock they hold under matrix load (the chip
// lowers its clock under load; MI355X_MICROARCH.md "DVFS give-back"), so a benchmark line that wants to be comparable
// across boxes carries this figure next to its own (bench.py: box_calibration, roofline.frac_of_sustained).
// Not on the product path: nothing in the generator calls it.
#include "nb_h3_common.h"
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void calib_mfma_f16_kernel(float* __restrict__ sink, unsigned long long* __restrict__ clk, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    // pseudo-random f16 operands in [-1, 1) from the lane / wave id (data toggling matters for power; zeros would run at 2.4 GHz)
    unsigned s = (unsigned)(blockIdx.x * 256 + tid) * 2654435761u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (_Float16)(((float)(s >> 8) * (1.f / 8388608.f)) - 1.f); };
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[i][j] = rnd(); b[i][j] = rnd(); }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + q) & 3], b[q], acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * 256 + tid] = t;                 // (keeps the loop alive; also orders the second pair of stamps behind it)
    if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    (void)lane;
}

// Runs the loop for about target_ms on every CU of the current device (blocking) and reports the sustained rate in TFLOP/s,
// the duration of the measured launch, and the median in-kernel clock (shader cycles per 100 MHz reference tick x 100 MHz).
extern "C" int nb_calibrate_mfma_f16(double target_ms, double* tflops, double* ms, double* clock_mhz, void* stream) {
    NB_REQUIRE(target_ms > 0.0 && target_ms <= 2000.0 && tflops, "calibrate_mfma_f16: bad arguments");
    int dev = 0;
    hipDeviceProp_t prop;
    NB_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess, "calibrate_mfma_f16: no device");
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    float* sink = nullptr;
    unsigned long long* clk = nullptr;
    NB_REQUIRE(hipMalloc(&sink, (size_t)cus * 256 * sizeof(float)) == hipSuccess && hipMalloc(&clk, (size_t)cus * 2 * sizeof(unsigned long long)) == hipSuccess,
               "calibrate_mfma_f16: allocation failed");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](int iters) {
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(calib_mfma_f16_kernel, dim3(cus), dim3(256), 0, st, sink, clk, iters);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        return (double)t;
    };
    run(64);                                          // code object load, clocks up
    const double probe = run(4096);                   // ~1.4 ms at 2 GHz
    int iters = (int)(4096.0 * target_ms / (probe > 1e-3 ? probe : 1e-3));
    iters = iters < 256 ? 256 : iters;
    const double t = run(iters);
    int rc = NB_OK;
    if (hipGetLastError() != hipSuccess) { nb_set_error("calibrate_mfma_f16: launch failed"); rc = NB_ELAUNCH; }
    if (rc == NB_OK) {
        const double flops = (double)cus * 4 /*waves*/ * iters * 16 /*MFMAs*/ * (2.0 * 32 * 32 * 16);
        *tflops = flops / (t * 1e-3) / 1e12;
        if (ms) *ms = t;
        if (clock_mhz) {
            std::vector<unsigned long long> h(2 * (size_t)cus);
            (void)hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::vector<double> f;
            for (int i = 0; i < cus; ++i) if (h[2 * i + 1]) f.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
            std::sort(f.begin(), f.end());
            *clock_mhz = f.empty() ? 0.0 : f[f.size() / 2];
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(sink); (void)hipFree(clk);
    return rc;
}
