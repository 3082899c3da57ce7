// Attempt at a stand-alone reproducer of the round-2 finding: v_pk_add_f32 d, a, v[pair] op_sel:[0,1]
// (both results read the pair's HIGH dword) right after the pair was loaded from LDS, 8 waves per workgroup, other LDS
// traffic in flight, the whole chip busy with two launches on two streams.  Counts results whose low half is not
// a.lo + pair.hi.
//   hipcc -O3 --offload-arch=gfx950 pk_add_opsel.hip -o pk_add_opsel && ./pk_add_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(const float* src, unsigned* bad, int iters) {
    __shared__ __attribute__((aligned(16))) float s_noise[2 * 12 * 2 * 32];
    __shared__ __attribute__((aligned(16))) f32x4 pad[4096];
    const int tid = threadIdx.x;
    for (int e = tid; e < 2 * 12 * 2 * 32; e += 512) s_noise[e] = src[(blockIdx.x * 1536 + e) & 65535];
    for (int e = tid; e < 4096; e += 512) pad[e] = f32x4{(float)e, 1.f, 2.f, 3.f};
    __syncthreads();
    unsigned nbad = 0;
    const int ti = (tid >> 5) % 12, tj = tid & 31;
    const unsigned addr = (unsigned)(size_t)(s_noise + (2 * ti) * 64 + 2 * tj);     // LDS byte address of the lane's noise pair
    const float want_hi = s_noise[(2 * ti + 1) * 64 + 2 * tj + 1];      // (offset1:32 x 8 bytes = the next noise row)
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        const f32x4 q = pad[(tid * 7 + it * 13) & 4095];           // other LDS traffic in flight
        const f32x2 a = {q[0] * 0.5f + it, q[1] + it};
        f32x2 r;
        asm volatile("ds_read2_b64 v[100:103], %1 offset1:32\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_pk_add_f32 %0, %2, v[102:103] op_sel:[0,1]"
                     : "=&v"(r) : "v"(addr), "v"(a) : "memory", "v100", "v101", "v102", "v103");
        if (r[0] != a[0] + want_hi || r[1] != a[1] + want_hi) ++nbad;
        sink += r[0];
    }
    if (sink == 12345.f) bad[1] = 1;
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    float* src; unsigned* bad;
    hipMalloc(&src, 65536 * 4); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = rand() / (float)RAND_MAX * 4.f - 2.f;
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    hipStream_t s[2]; hipStreamCreate(&s[0]); hipStreamCreate(&s[1]);
    for (int rep = 0; rep < 20; ++rep)
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(2048), dim3(512), 0, s[i], src, bad, 2000);
    hipDeviceSynchronize();
    unsigned hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("results with a wrong half: %u of %llu\n", hb[0], 40ull * 2048 * 512 * 2000);
    return 0;
}
