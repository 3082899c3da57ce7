// fp6 (e2m3) operands for the correction products (round 5): what the hardware does, checked against a host evaluation.
//   (1) v_cvt_scalef32_2xpk16_fp6_f32 (32 floats -> 32 x 6 bit in 6 registers): element order in the bit stream, what the scale operand
//       does (divide? power of two only?), rounding and saturation against a host round-to-nearest-even onto the e2m3 grid;
//   (2) v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 2 (both operands e2m3): lane l holds row/col l % 32 and the 32 K elements
//       32 (l / 32) .. + 31 in its first SIX registers (element j at bits 6 j .. 6 j + 5), one E8M0 scale byte per lane; operands built
//       by the converter on the device, per-lane scales 2^-3 .. 2^3, result against a float64 dot product;
//   (3) cycles per instruction back to back on one SIMD: f16 32x32x16, fp8 and fp6 32x32x64.
//   hipcc -O3 --offload-arch=gfx950 mfma_f6_check.hip -o bin/mfma_f6_check && ./bin/mfma_f6_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void k_cvt(const float* in, unsigned* out, float* back, const float* scale) {
    v16f a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x * 32 + i]; b[i] = in[threadIdx.x * 32 + 16 + i]; }
    const float s = scale[threadIdx.x];
    v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, s);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
    v32f d = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(r, 1.0f);
    for (int i = 0; i < 32; ++i) back[threadIdx.x * 32 + i] = d[i];
}

// A: [64 lanes][32] floats (lane l: row l % 32, K block l / 32), B likewise (col l % 32); per-lane scale exponents
__global__ void k_mfma(const float* A, const float* B, const int* sa, const int* sb, float* D) {
    const int l = threadIdx.x;
    v16f a0, a1, b0, b1;
    for (int i = 0; i < 16; ++i) { a0[i] = A[l * 32 + i]; a1[i] = A[l * 32 + 16 + i]; b0[i] = B[l * 32 + i]; b1[i] = B[l * 32 + 16 + i]; }
    const v6u ra = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, 1.0f), rb = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(b0, b1, 1.0f);
    i32x8 ta = {(int)ra[0], (int)ra[1], (int)ra[2], (int)ra[3], (int)ra[4], (int)ra[5], 0x7fffffff, -1};      // (registers 6, 7 must not matter)
    i32x8 tb = {(int)rb[0], (int)rb[1], (int)rb[2], (int)rb[3], (int)rb[4], (int)rb[5], -1, 0x7fffffff};
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ta, tb, acc, 2, 2, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

struct Stamp { unsigned long long ticks, real; };
#define RATE_KERNEL(NAME, INSTR, TA, TB)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(const i32x8* ops, float* out, Stamp* st, int iters) {                      \
        __shared__ float pad[25 * 1024];                                                                                   \
        const int lane = threadIdx.x & 63;                                                                                 \
        f32x16 acc[4];                                                                                                     \
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;                                         \
        const i32x8 a8 = ops[lane], b8 = ops[64 + lane];                                                                   \
        TA a; TB b;                                                                                                        \
        for (int i = 0; i < (int)(sizeof(TA) / 4); ++i) a[i] = a8[i];                                                      \
        for (int i = 0; i < (int)(sizeof(TB) / 4); ++i) b[i] = b8[i];                                                      \
        int sc = 127;                                                                                                      \
        ((volatile float*)pad)[threadIdx.x * 50] = (float)threadIdx.x;                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                       \
        __syncthreads();                                                                                                   \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
        for (int it = 0; it < iters; ++it)                                                                                 \
            asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(0) INSTR(1) INSTR(2) INSTR(3) \
                         : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]) : [a] "v"(a), [b] "v"(b), [s] "v"(sc)); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
        float s = ((volatile float*)pad)[threadIdx.x * 50];                                                                \
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                    \
        if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t1 - t0, r1 - r0};                                                    \
    }
// (the f16 form reads the first four registers of the tuples)
#define I_F16(i) "v_mfma_f32_32x32x16_f16 %[c" #i "], %[a], %[b], %[c" #i "]\n"
#define I_FP8(i) "v_mfma_scale_f32_32x32x64_f8f6f4 %[c" #i "], %[a], %[b], %[c" #i "], %[s], %[s] op_sel_hi:[0,0,0]\n"
#define I_FP6(i) "v_mfma_scale_f32_32x32x64_f8f6f4 %[c" #i "], %[a], %[b], %[c" #i "], %[s], %[s] op_sel_hi:[0,0,0] cbsz:2 blgp:2\n"
#define I_F8A6B(i) "v_mfma_scale_f32_32x32x64_f8f6f4 %[c" #i "], %[a], %[b], %[c" #i "], %[s], %[s] op_sel_hi:[0,0,0] blgp:2\n"
__global__ __launch_bounds__(256) void rate_f16(const i32x8* ops, float* out, Stamp* st, int iters) {
    __shared__ float pad[25 * 1024];
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const h8 a = __builtin_bit_cast(h8, ((const float4*)ops)[lane]), b = __builtin_bit_cast(h8, ((const float4*)ops)[64 + lane]);
    ((volatile float*)pad)[threadIdx.x * 50] = (float)threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it)
        asm volatile(I_F16(0) I_F16(1) I_F16(2) I_F16(3) I_F16(0) I_F16(1) I_F16(2) I_F16(3) I_F16(0) I_F16(1) I_F16(2) I_F16(3) I_F16(0) I_F16(1) I_F16(2) I_F16(3)
                     : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]) : [a] "v"(a), [b] "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = ((volatile float*)pad)[threadIdx.x * 50];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t1 - t0, r1 - r0};
}
RATE_KERNEL(rate_fp8, I_FP8, i32x8, i32x8)
RATE_KERNEL(rate_fp6, I_FP6, v6i, v6i)
RATE_KERNEL(rate_f8a6b, I_F8A6B, i32x8, v6i)

static const float GRID[32] = {0, .125f, .25f, .375f, .5f, .625f, .75f, .875f, 1, 1.125f, 1.25f, 1.375f, 1.5f, 1.625f, 1.75f, 1.875f,
                               2, 2.25f, 2.5f, 2.75f, 3, 3.25f, 3.5f, 3.75f, 4, 4.5f, 5, 5.5f, 6, 6.5f, 7, 7.5f};
static int enc_host(float x) {                    // round to nearest even onto the grid, saturating; code = s eee? no: s | e(2) | m(3)
    const int s = std::signbit(x) ? 32 : 0;
    float a = fabsf(x);
    if (!(a == a)) return s | 31;
    if (a > 7.5f) a = 7.5f;
    const float step = a < 2 ? .125f : a < 4 ? .25f : .5f;
    const float q = nearbyintf(a / step) * step;   // (ties to even under the default rounding mode)
    for (int c = 0; c < 32; ++c) if (GRID[c] == q) return s | c;
    return -1;
}
static float dec_host(int c) { return (c & 32) ? -GRID[c & 31] : GRID[c & 31]; }
static int field(const unsigned* w, int j) {     // 6 bits at bit offset 6 j of a 192-bit little-endian stream
    const int bit = 6 * j, d = bit >> 5, o = bit & 31;
    unsigned long long v = w[d];
    if (d + 1 < 6) v |= (unsigned long long)w[d + 1] << 32;
    return (int)((v >> o) & 63);
}

int main() {
    srand(5);
    // ---- (1) the converter ----
    std::vector<float> in(64 * 32), sc(64);
    for (int l = 0; l < 64; ++l) {
        sc[l] = l < 16 ? 1.f : l < 32 ? ldexpf(1.f, (l % 7) - 3) : l < 48 ? 3.f : 1.f;          // lanes 32-47: a scale that is no power of two
        for (int j = 0; j < 32; ++j) {
            float v;
            if (l == 0) v = GRID[j];                                      // lane 0: the grid itself, ascending
            else if (l == 1) v = -GRID[31 - j];
            else if (l >= 48) v = ((rand() / (float)RAND_MAX) - .5f) * (l >= 56 ? 40.f : 4.f);    // incl. values beyond 7.5 (saturation)
            else v = ((rand() / (float)RAND_MAX) - .5f) * 15.f * sc[l];
            in[l * 32 + j] = v;
        }
    }
    in[2 * 32 + 0] = 0.0625f; in[2 * 32 + 1] = 0.1875f; in[2 * 32 + 2] = 1.9375f; in[2 * 32 + 3] = 0.3125f; in[2 * 32 + 4] = 7.75f; in[2 * 32 + 5] = 1e30f;   // ties, top
    float *d_in, *d_back, *d_sc; unsigned* d_out;
    hipMalloc(&d_in, in.size() * 4); hipMalloc(&d_back, in.size() * 4); hipMalloc(&d_sc, 256); hipMalloc(&d_out, 64 * 6 * 4);
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_sc, sc.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, d_in, d_out, d_back, d_sc);
    std::vector<unsigned> out(64 * 6); std::vector<float> back(64 * 32);
    hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(back.data(), d_back, back.size() * 4, hipMemcpyDeviceToHost);
    printf("lane 0 (inputs = the e2m3 grid ascending): %08x %08x %08x %08x %08x %08x\n", out[0], out[1], out[2], out[3], out[4], out[5]);
    // input element e (src0[0..15] = e 0..15, src1[0..15] = e 16..31) sits in field F(e) = 2 (e % 16) + e / 16: the two sources INTERLEAVED
    auto F = [](int e) { return 2 * (e & 15) + (e >> 4); };
    int in_order = 1, inter = 1, dec_field = 1;
    for (int j = 0; j < 32; ++j) {
        in_order &= field(&out[0], j) == j; inter &= field(&out[0], F(j)) == j; dec_field &= back[F(j)] == GRID[j];
        if (field(&out[0], F(j)) != j || back[F(j)] != GRID[j]) printf("  lane 0 element %d: field %d holds %d, decoded %g (expected %d, %g)\n", j, F(j), field(&out[0], F(j)), back[F(j)], j, GRID[j]);
    }
    printf("code = s|e(2)|m(3) in 6-bit fields of a little-endian stream; element e in field e: %s; in field 2 (e %% 16) + e / 16 (src0 / src1 interleaved): %s; "
           "v_cvt_scalef32_pk32_f32_fp6 returns field f as element f: %s\n", in_order ? "YES" : "NO", inter ? "YES" : "NO", dec_field ? "YES" : "NO");
    int bad_div = 0, bad_mul = 0, bad_p2 = 0, n_p2 = 0, n_np2 = 0, bad_np2_floor = 0;
    for (int l = 1; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            const int got = field(&out[l * 6], F(j));
            const float x = in[l * 32 + j], s = sc[l];
            const bool p2 = (l < 32 || l >= 48);
            if (p2) {
                ++n_p2; bad_div += got != enc_host(x / s); bad_mul += got != enc_host(x * s);
                if (got != enc_host(x / s) && bad_div <= 12) printf("  lane %d element %d: x %.9g scale %g -> code %d (%g); host RNE(x / scale) = %d (%g)\n", l, j, x, s, got, dec_host(got), enc_host(x / s), dec_host(enc_host(x / s)));
            }
            else { ++n_np2; bad_p2 += got != enc_host(x / s); bad_np2_floor += got != enc_host(x / 2.f); }     // 3.0 -> exponent only = 2.0 ?
        }
    printf("power-of-two scales: %d values; mismatches vs host RNE(x / scale) %d, vs RNE(x * scale) %d\n", n_p2, bad_div, bad_mul);
    printf("scale 3.0: %d values; mismatches vs RNE(x / 3) %d, vs RNE(x / 2) [exponent of the scale only] %d\n", n_np2, bad_p2, bad_np2_floor);
    printf("ties / top (lane 2): in 0.0625 0.1875 1.9375 0.3125 7.75 1e30 -> %g %g %g %g %g %g\n", back[64 + F(0)], back[64 + F(1)], back[64 + F(2)], back[64 + F(3)], back[64 + F(4)], back[64 + F(5)]);

    // ---- (2) the MFMA ----
    std::vector<float> A(64 * 32), B(64 * 32); std::vector<int> sa(64), sb(64);
    for (auto& v : A) v = dec_host(rand() & 63);
    for (auto& v : B) v = dec_host(rand() & 63);
    for (int l = 0; l < 64; ++l) { sa[l] = 127 + (rand() % 7) - 3; sb[l] = 127 + (rand() % 7) - 3; }
    float *dA, *dB, *dD; int *dsa, *dsb;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            double ref = 0;
            for (int kb = 0; kb < 2; ++kb)
                for (int j = 0; j < 32; ++j)
                    ref += (double)A[(kb * 32 + i) * 32 + j] * ldexp(1.0, sa[kb * 32 + i] - 127) * (double)B[(kb * 32 + n) * 32 + j] * ldexp(1.0, sb[kb * 32 + n] - 127);
            maxerr = fmax(maxerr, fabs(ref - D[i * 32 + n])); maxref = fmax(maxref, fabs(ref));
        }
    printf("fp6 x fp6 MFMA (converter-built operands, per-lane scales 2^-3..2^3, junk in registers 6-7): max |ref| %.4g, max abs err %.4g (%s)\n", maxref, maxerr,
           maxerr <= 1e-5 * maxref ? "OK: layout and scale semantics as assumed" : "MISMATCH");

    // ---- (3) rates ----
    std::vector<int> ops(128 * 8);
    for (auto& v : ops) v = (rand() & 0x3f3f3f3f) | 0x10101010;           // bytes valid as e4m3 and as f16 pairs of moderate size
    i32x8* d_ops; float* d_o; Stamp* d_st;
    hipMalloc(&d_ops, ops.size() * 4); hipMalloc(&d_o, 256 * 256 * 4); hipMalloc(&d_st, 256 * sizeof(Stamp));
    hipMemcpy(d_ops, ops.data(), ops.size() * 4, hipMemcpyHostToDevice);
    typedef void (*RK)(const i32x8*, float*, Stamp*, int);
    struct { const char* name; RK k; double flop; } rk[4] = {{"f16 32x32x16", rate_f16, 2.0 * 32 * 32 * 16}, {"fp8 32x32x64 (e4m3 x e4m3)", rate_fp8, 2.0 * 32 * 32 * 64},
                                                           {"fp6 32x32x64 (e2m3 x e2m3)", rate_fp6, 2.0 * 32 * 32 * 64}, {"A e4m3 x B e2m3 32x32x64", rate_f8a6b, 2.0 * 32 * 32 * 64}};
    for (auto& r : rk) {
        const int iters = 40000;
        float ms = 0.f; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(r.k, dim3(256), dim3(256), 0, 0, d_ops, d_o, d_st, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
        std::vector<Stamp> h(256); hipMemcpy(h.data(), d_st, 256 * sizeof(Stamp), hipMemcpyDeviceToHost);
        double t = 0, rr = 0; for (auto& s : h) { t += s.ticks; rr += s.real; } t /= 256; rr /= 256;
        printf("rate  %-28s %6.2f ticks per instruction (one wave per SIMD, whole chip), tick clock %.0f MHz, %.0f TFLOP/s (wall %.2f ms)\n", r.name, t / (iters * 16.0),
               t / (rr * 10.0) * 1e3, r.flop * iters * 16.0 * 4 * 256 / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
