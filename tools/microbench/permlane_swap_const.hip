// Reproducer: __builtin_amdgcn_permlane32_swap with hipcc (ROCm 7.2, -O3, gfx950).  When the two results are used in one
// expression (r[0] + r[1]), or one operand is a constant, the generated code reads the FIRST result for both
// (v_permlane32_swap v4, v2; v_add_f32 v2, v4, v4): lane 1 prints 2 instead of 1 + 33.  The kernels use inline assembly
// with two read-write registers instead (nb_swap32, nb_modconv_h3.hip).
//   hipcc -O3 --offload-arch=gfx950 permlane_swap_const.hip -o permlane_swap_const && ./permlane_swap_const
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, const float* in) {
    float x = in[threadIdx.x], y = in[64 + threadIdx.x];
    const u2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    out[threadIdx.x] = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    unsigned z = 0; asm volatile("" : "+v"(z));
    const u2 q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), z, false, false);
    out[64 + threadIdx.x] = __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1]);
    const u2 w = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), z, false, false);
    out[128 + threadIdx.x] = __builtin_bit_cast(float, w[0]); out[192 + threadIdx.x] = __builtin_bit_cast(float, w[1]);
}
int main() {
    float h[128]; for (int i = 0; i < 64; ++i) { h[i] = i; h[64 + i] = 1000 + i; }
    float *o, *in; hipMalloc(&o, 256 * 4); hipMalloc(&in, 512); hipMemcpy(in, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, in);
    float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("x,y distinct: sum lane0 %g (expect 0+32=32) lane1 %g (34) lane32 %g (1000+1032) \n", r[0], r[1], r[32]);
    printf("x,opaque 0:   sum lane0 %g (expect 32) lane1 %g (34)\n", r[64], r[65]);
    printf("separate: r0 lane1 %g r1 lane1 %g\n", r[129], r[193]);
}
