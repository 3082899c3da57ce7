// Semantics check of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands, per-lane E8M0 block scales) against a host
// evaluation: lane l holds row/col l%32 and the 32 K-elements 32*(l/32) .. +31 (byte j of the 8 dwords = element j).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void k(const uint8_t* A, const uint8_t* B, const int* sa, const int* sb, float* D) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = ((const int*)(A + lane * 32))[j];
        b[j] = ((const int*)(B + lane * 32))[j];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa[lane], 0, sb[lane]);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        D[row * 32 + col] = acc[r];
    }
}

static float dec(uint8_t v) {          // e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    uint8_t hA[64 * 32], hB[64 * 32]; int hsa[64], hsb[64];
    srand(1);
    for (int i = 0; i < 64 * 32; ++i) {
        do { hA[i] = rand() & 255; } while ((hA[i] & 0x7f) == 0x7f);      // skip NaN
        do { hB[i] = rand() & 255; } while ((hB[i] & 0x7f) == 0x7f);
    }
    for (int l = 0; l < 64; ++l) { hsa[l] = l < 32 ? 127 : 116; hsb[l] = l < 32 ? 118 : 129; }
    uint8_t *A, *B; int *sa, *sb; float* D;
    (void)hipMalloc(&A, sizeof(hA)); (void)hipMalloc(&B, sizeof(hB)); (void)hipMalloc(&sa, sizeof(hsa)); (void)hipMalloc(&sb, sizeof(hsb)); (void)hipMalloc(&D, 32 * 32 * 4);
    (void)hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    (void)hipMemcpy(sa, hsa, sizeof(hsa), hipMemcpyHostToDevice); (void)hipMemcpy(sb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, sa, sb, D);
    float hD[32 * 32];
    (void)hipMemcpy(hD, D, sizeof(hD), hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            double ref = 0;
            for (int kb = 0; kb < 2; ++kb)
                for (int j = 0; j < 32; ++j)
                    ref += (double)dec(hA[(kb * 32 + i) * 32 + j]) * ldexp(1.0, hsa[kb * 32 + i] - 127) *
                           (double)dec(hB[(kb * 32 + n) * 32 + j]) * ldexp(1.0, hsb[kb * 32 + n] - 127);
            maxerr = fmax(maxerr, fabs(ref - hD[i * 32 + n]));
            maxref = fmax(maxref, fabs(ref));
        }
    printf("max |ref| %.4g, max abs err %.4g (%s)\n", maxref, maxerr, maxerr <= 1e-5 * maxref ? "OK: layout and scale semantics as assumed" : "MISMATCH");
    return 0;
}
