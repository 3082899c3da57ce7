// Microbenchmark (round 5): two questions that the round-4 numbers answered against the platform guide.
//   (1) what does ONE wave pay per independent plain VALU instruction (guide: 4 cycles for v_fma_f32 / v_add_f32 / v_max3, 4-5 for
//       v_cvt_pk; round 4's valu_issue.hip: 6.5-7.7 "cycles")?  Here every figure is given three ways -- s_memtime ticks,
//       s_memrealtime (100 MHz) and the host's HIP-event wall time -- over runs of >= 5 ms with exactly one workgroup per CU
//       (64 KiB of LDS per workgroup... 3 workgroups would fit: the grid has 256 workgroups of 100 KiB), so that a tick that is not
//       a shader cycle, or a clock that has not settled, shows up as a disagreement between the three.
//   (2) how many UNPACKED single-issue VALU instructions hide in the gap of a back-to-back v_mfma_f32_32x32x16_f16 stream
//       (guide: <= 5 per gap nearly free, packed f32 the exception), at one and at two waves per SIMD?
//   hipcc -O3 --offload-arch=gfx950 issue_rates.hip -o bin/issue_rates && ./bin/issue_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { K_FMA = 0, K_ADD, K_MED3, K_CVT, K_PKFMA, K_NOP, K_MAX3, K_NONE };

template <int KIND>
__device__ __forceinline__ void filler(float& v, f32x2& p, float m, float c) {
    if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c));
    else if constexpr (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(c));
    else if constexpr (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c));
    else if constexpr (KIND == K_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c));
    else if constexpr (KIND == K_CVT) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v) : "v"(m));
    else if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(f32x2{m, m}), "v"(f32x2{c, c}));
    else if constexpr (KIND == K_NOP) asm volatile("s_nop 0");
}

struct Stamp { unsigned long long ticks, real; };

// (1) pure issue: 16 independent instructions per iteration
template <int KIND, int UNROLL>
__global__ __launch_bounds__(1024) void k_issue(float* out, Stamp* st, int iters) {
    __shared__ float pad[25 * 1024];                          // 100 KiB: one workgroup per CU
    float v[16]; f32x2 p[16];
    for (int i = 0; i < 16; ++i) { v[i] = (float)(threadIdx.x + i) * 1e-3f; p[i] = f32x2{v[i], v[i] + 1.f}; }
    const float m = 0.999f, c = 1e-3f;
    ((volatile float*)pad)[threadIdx.x * 25] = v[0];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // UNROLL x 16 independent instructions per loop iteration: with 16 (round 4's valu_issue.hip, and the first version of this file)
    // the taken branch at the end of the body -- ~25 cycles -- was a quarter of the "cycles per instruction"
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) filler<KIND>(v[i], p[i], m, c);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = ((volatile float*)pad)[threadIdx.x * 25];
    for (int i = 0; i < 16; ++i) s += v[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t1 - t0, r1 - r0};
}

// (2) MFMA stream with NF fillers behind every MFMA; 16 MFMAs per iteration on four accumulators.  MFMA + its fillers are ONE asm
// statement (between separate statements hipcc put an `s_nop 0` behind most MFMAs: 4 cycles of issue that are not the question).
#define I_FMA(r) "v_fma_f32 %[" #r "], %[" #r "], %[m], %[c]\n"
#define I_MED3(r) "v_med3_f32 %[" #r "], %[" #r "], %[m], %[c]\n"
#define I_CVT(r) "v_cvt_pk_f16_f32 %[" #r "], %[" #r "], %[m]\n"
#define I_PK(r) "v_pk_fma_f32 %[" #r "], %[" #r "], %[m], %[c]\n"
#define SEQ0(I) ""
#define SEQ1(I) I(v0)
#define SEQ2(I) SEQ1(I) I(v1)
#define SEQ3(I) SEQ2(I) I(v2)
#define SEQ4(I) SEQ3(I) I(v3)
#define SEQ5(I) SEQ4(I) I(v4)
#define SEQ6(I) SEQ5(I) I(v5)
#define SEQ8(I) SEQ6(I) I(v6) I(v7)
#define MF(i, q, SEQ) "v_mfma_f32_32x32x16_f16 %[acc" #i "], %[a" #q "], %[b" #q "], %[acc" #i "]\n" SEQ
#define MF4(q0, q1, q2, q3, SEQ) MF(0, q0, SEQ) MF(1, q1, SEQ) MF(2, q2, SEQ) MF(3, q3, SEQ)
// the whole iteration (16 MFMAs, each followed by its fillers) is ONE asm statement: between separate statements hipcc
// (ROCm 7.2) puts an `s_nop 0` -- 4 cycles of issue that are not the question -- behind most MFMAs
#define DEFINE_MIX(NAME, SEQ, VT, VINIT)                                                                                              \
    __global__ __launch_bounds__(512) void NAME(const h8* ops, float* out, Stamp* st, int iters) {                                   \
        __shared__ float pad[25 * 1024];                                                                                              \
        const int lane = threadIdx.x & 63;                                                                                            \
        f32x16 acc[4];                                                                                                                \
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;                                                    \
        h8 a[4], b[4];                                                                                                                \
        for (int i = 0; i < 4; ++i) { a[i] = ops[(i * 2) * 64 + lane]; b[i] = ops[(i * 2 + 1) * 64 + lane]; }                       \
        VT v[8];                                                                                                                      \
        for (int i = 0; i < 8; ++i) v[i] = VINIT((float)(threadIdx.x + i) * 1e-3f);                                                   \
        const VT m = VINIT(0.999f), c = VINIT(1e-3f);                                                                                 \
        ((volatile float*)pad)[threadIdx.x * 50] = (float)threadIdx.x;                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                                  \
        __syncthreads();                                                                                                              \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                            \
        for (int it = 0; it < iters; ++it) {                                                                                          \
            asm volatile(MF4(0, 1, 2, 3, SEQ) MF4(1, 2, 3, 0, SEQ) MF4(2, 3, 0, 1, SEQ) MF4(3, 0, 1, 2, SEQ)                          \
                         : [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [acc3] "+v"(acc[3]), [v0] "+v"(v[0]),       \
                           [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]), [v6] "+v"(v[6]),      \
                           [v7] "+v"(v[7])                                                                                            \
                         : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]),            \
                           [b2] "v"(b[2]), [b3] "v"(b[3]), [m] "v"(m), [c] "v"(c));                                                   \
        }                                                                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                            \
        float s = ((volatile float*)pad)[threadIdx.x * 50];                                                                           \
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];                                                     \
        for (int i = 0; i < 8; ++i) s += sum_of(v[i]);                                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                               \
        if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t1 - t0, r1 - r0};                                                               \
    }
__device__ __forceinline__ float sum_of(float x) { return x; }
__device__ __forceinline__ float sum_of(f32x2 x) { return x[0] + x[1]; }
#define ID_F(x) (x)
#define ID_P(x) f32x2{(x), (x)}
#define DEFINE_KIND(K, I, VT, VI) DEFINE_MIX(mix_##K##_1, SEQ1(I), VT, VI) DEFINE_MIX(mix_##K##_2, SEQ2(I), VT, VI) DEFINE_MIX(mix_##K##_3, SEQ3(I), VT, VI) \
    DEFINE_MIX(mix_##K##_4, SEQ4(I), VT, VI) DEFINE_MIX(mix_##K##_5, SEQ5(I), VT, VI) DEFINE_MIX(mix_##K##_6, SEQ6(I), VT, VI) DEFINE_MIX(mix_##K##_8, SEQ8(I), VT, VI)
DEFINE_MIX(mix_bare, SEQ0(I_FMA), float, ID_F)
DEFINE_KIND(fma, I_FMA, float, ID_F)
DEFINE_KIND(med3, I_MED3, float, ID_F)
DEFINE_KIND(cvt, I_CVT, float, ID_F)
DEFINE_KIND(pk, I_PK, f32x2, ID_P)

static float* g_out; static Stamp* g_st; static h8* g_ops;

struct Res { double ticks_per, real_ns_per, wall_ns_per, clock_mhz; };

template <typename F>
static Res measure(F launch, int iters, double per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {                       // the last of three back-to-back runs is reported (settled clock)
        hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<Stamp> h(256);
    hipMemcpy(h.data(), g_st, 256 * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> t, r;
    for (auto& s : h) { t.push_back((double)s.ticks); r.push_back((double)s.real); }
    std::sort(t.begin(), t.end()); std::sort(r.begin(), r.end());
    const double tm = t[128], rm = r[128];
    hipEventDestroy(e0); hipEventDestroy(e1);
    return Res{tm / (iters * per_iter), rm * 10.0 / (iters * per_iter), ms * 1e6 / (iters * per_iter), tm / (rm * 10.0) * 1e3};
}

// (the stamps are wave 0's: the OLDEST wave of its SIMD, which wins every issue arbitration -- with more than one wave per SIMD its
//  figure is what a wave gets when it is never the loser, not a share; the last column is the whole kernel by the host's clock:
//  wall time x tick clock / instructions issued per SIMD)
template <int KIND, int UNROLL>
static void issue_row(const char* name) {
    for (int threads : {256, 512, 1024}) {
        const int iters = 6400000 / (16 * UNROLL) / (threads / 256);
        Res x = measure([&](int it) { hipLaunchKernelGGL((k_issue<KIND, UNROLL>), dim3(256), dim3(threads), 0, 0, g_out, g_st, it); }, iters, 16.0 * UNROLL);
        printf("issue  %-18s %3d per loop body, %d wave/SIMD: %6.2f ticks per instruction (oldest wave); tick clock %.0f MHz; SIMD throughput by wall time: %.2f cycles per instruction\n",
               name, 16 * UNROLL, threads / 256, x.ticks_per, x.clock_mhz, x.wall_ns_per * x.clock_mhz * 1e-3 / (threads / 256));
    }
}

typedef void (*MixK)(const h8*, float*, Stamp*, int);
static void mix_row(const char* name, int nf, MixK kern, int threads) {
    const int iters = 60000 / (threads / 256);
    Res x = measure([&](int it) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, g_ops, g_out, g_st, it); }, iters, 16.0);
    printf("mix    %-18s x%d per MFMA, %d wave/SIMD: %6.2f ticks per MFMA per wave (%6.2f per SIMD)  wall %.3f ns  tick clock %.0f MHz\n",
           name, nf, threads / 256, x.ticks_per, x.ticks_per / (threads / 256), x.wall_ns_per, x.clock_mhz);
}
#define MIX_ROWS(K, NAME) for (int threads : {256, 512}) { mix_row(NAME, 1, mix_##K##_1, threads); mix_row(NAME, 2, mix_##K##_2, threads); \
    mix_row(NAME, 3, mix_##K##_3, threads); mix_row(NAME, 4, mix_##K##_4, threads); mix_row(NAME, 5, mix_##K##_5, threads);              \
    mix_row(NAME, 6, mix_##K##_6, threads); mix_row(NAME, 8, mix_##K##_8, threads); }

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4); hipMalloc(&g_st, 256 * sizeof(Stamp)); hipMalloc(&g_ops, 8 * 64 * 16);
    _Float16 h[8 * 64 * 8]; srand(3);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    hipMemcpy(g_ops, h, sizeof(h), hipMemcpyHostToDevice);
    issue_row<K_FMA, 1>("v_fma_f32"); issue_row<K_FMA, 16>("v_fma_f32"); issue_row<K_ADD, 16>("v_add_f32"); issue_row<K_MAX3, 16>("v_max3_f32");
    issue_row<K_MED3, 16>("v_med3_f32"); issue_row<K_CVT, 16>("v_cvt_pk_f16_f32"); issue_row<K_PKFMA, 16>("v_pk_fma_f32"); issue_row<K_NOP, 1>("s_nop 0");
    issue_row<K_NOP, 16>("s_nop 0");
    mix_row("(bare MFMA)", 0, mix_bare, 256); mix_row("(bare MFMA)", 0, mix_bare, 512);
    MIX_ROWS(fma, "v_fma_f32") MIX_ROWS(med3, "v_med3_f32") MIX_ROWS(cvt, "v_cvt_pk_f16_f32") MIX_ROWS(pk, "v_pk_fma_f32")
    return 0;
}
