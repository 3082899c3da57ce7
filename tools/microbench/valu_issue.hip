// Microbenchmark: VALU issue cost per instruction with ONE wave per SIMD (256-thread workgroups) vs TWO (512), for the
// instruction kinds of the up=2 epilogue: v_fma_f32, v_pk_fma_f32, v_med3_f32, v_cvt_pk_f16_f32 (cvt_pkrtz), ds_read_b128 + use.
// Cycles per instruction per wave via s_memtime around an unrolled loop of independent instructions.
//   hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o bin/valu_issue && ./bin/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f32x2{v[2 * i], v[2 * i + 1]};
    const float m = 0.999f, c = 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f32x2{m, m}), "v"(f32x2{c, c}));
        } else if constexpr (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
        } else if constexpr (KIND == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
        } else if constexpr (KIND == 4) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f32x2{m, m}));
        } else if constexpr (KIND == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c));
        } else if constexpr (KIND == 6) {     // dependent chain of v_fma (latency)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(m), "v"(c));
        } else if constexpr (KIND == 7) {     // v_mov_b32 dpp (row_shl:1)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long h[256];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
        printf("%-22s %4d threads/CU (%d waves/SIMD): %.2f cycles per instruction per wave\n", name, threads, threads >= 256 ? threads / 256 : 0, s / 256 / iters / 16);
    }
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("v_fma_f32", out, cyc); run<1>("v_pk_fma_f32", out, cyc); run<4>("v_pk_mul_f32", out, cyc); run<2>("v_med3_f32", out, cyc);
    run<3>("v_cvt_pk_f16_f32", out, cyc); run<5>("v_cvt_pk_fp8_f32", out, cyc); run<6>("v_fma_f32 dependent", out, cyc); run<7>("v_mov_b32_dpp row_shl", out, cyc);
    return 0;
}
