// Shared device helpers of the split-f16 ("h3" / "f8") convolution kernels: vector types, LDS-DMA, counted waits, fp8 packing,
// lane-half exchange, the up=2 kernels' parameter block.  Included by nb_modconv_h3.hip and nb_modconv_up2w.hip.
#pragma once
#include "nb_common.h"
#include <cstdlib>
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define NB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define NB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))


#define NB_TSTAMP(k)                                                                                         \
    do {                                                                                                     \
        if (p.tstamps && threadIdx.x == 0) p.tstamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// De-synchronise the chip: without this every CU runs the same tile schedule in lockstep, so all epilogue store
// bursts (and all prologue DMA bursts) hit HBM at the same moments while the matrix pipes idle.  The workgroups of
// the first dispatch round (one per CU) start `phase/16` of a tile time apart; later workgroups inherit the offsets
// as slots free up.  `stagger_ticks` = tile time / 16 in 100 MHz s_memrealtime ticks (0 = off).
__device__ __forceinline__ void nb_stagger(int stagger_ticks, int first_round) {
    const int bid = blockIdx.x + blockIdx.y * gridDim.x;
    if (stagger_ticks > 0 && bid < first_round) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)(bid & 15) * stagger_ticks;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
}

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// s_waitcnt vmcnt(N) as the BUILTIN, not inline assembly: the compiler's own wait bookkeeping (SIInsertWaitcnts) then sees
// the LDS-DMA operations complete.  With asm waits it never does, keeps every LDS-DMA "pending" for the rest of the kernel and
// -- because an LDS-DMA counts as a flat access that may touch LDS -- turns every wait for a fragment read into
// `s_waitcnt lgkmcnt(0)`, also where only the oldest of sixteen outstanding reads is needed.
// gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
#define NB_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))

// One LDS-DMA piece (64 lanes x 16 bytes, global -> LDS at lds + 16 * lane) as INLINE ASSEMBLY.  The builtin
// (__builtin_amdgcn_global_load_lds) is a flat-segment access that may touch LDS as far as the compiler's wait bookkeeping is
// concerned: while one is pending -- and with counted inline-asm vmcnt waits it never sees them complete -- every wait for an
// LDS fragment read is forced to `s_waitcnt lgkmcnt(0)`, also where only the oldest of sixteen reads in flight is needed.
// Hidden in an asm statement the copy is nobody's business but ours (the K loops count their vmcnt themselves anyway) and the
// fragment reads get counted waits.  M0 (the LDS destination base) is saved and restored around the copy.
__device__ __forceinline__ void nb_lds_dma16(const void* src, const void* lds) {
    const unsigned dst = (unsigned)(uintptr_t)NB_LDS_PTR(lds);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
}

// LDS-DMA pieces of the one-wave-per-SIMD / software-pipelined up=2 kernels (see nb_lds_dma16): the LDS destination is a byte address (no generic -> LDS pointer cast per
// piece).  _s: uniform 64-bit base (SGPR pair) + 32-bit lane offset: no per-piece vector address arithmetic.
__device__ __forceinline__ void nb_lds_dma16_s(const void* sbase, unsigned voff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
// _m: per-lane 64-bit source, lanes chosen by a uniform mask set INSIDE the statement: an `if` around the copy is a branch, and a
// K loop of several basic blocks is no longer scheduled as written (the compiler sinks MFMAs across the blocks, past the fences)
__device__ __forceinline__ void nb_lds_dma16_m(const void* src, unsigned lds_byte, unsigned long long mask) {
    unsigned keep;
    unsigned long long ex;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, %4\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(ex) : "v"(src), "s"(lds_byte), "s"(mask) : "memory");
}

// compile-time loop: f(std::integral_constant<int, K0>{}) ... f(std::integral_constant<int, K1 - 1>{})
template <int K0, int K1, class F>
__device__ __forceinline__ void nb_static_for(F&& f) {
    if constexpr (K0 < K1) {
        f(std::integral_constant<int, K0>{});
        nb_static_for<K0 + 1, K1>(f);
    }
}

// "f8" operand format (F8 = true): the two correction products run on ONE block-scaled fp8 MFMA per tap pair.
//   activations: the (cg, lo) slots of a 16-channel chunk hold, instead of the f16 low halves,
//       (cg 2k,   lo) = fp8 e4m3( xl * 2^9 ) of the chunk's 16 channels     (xl = x - f16(x))
//       (cg 2k+1, lo) = fp8 e4m3( x / 4 )    of the chunk's 16 channels
//   weights likewise: (cg 0, lo) = fp8(w), (cg 1, lo) = fp8((w - f16(w)) * 2^11)
//   v_mfma_scale_f32_32x32x64_f8f6f4: lane half lh = 0 contracts  fp8(w) . fp8(xl 2^9) * 2^-9,  lh = 1 contracts
//   fp8(wl 2^11) 2^-11 . fp8(x/4) 2^2; its K = 64 = 2 taps x 16 channels x 2 terms (the third tap of a row rides alone).
// Same containers, same staging, same fragment reads as the f16 lo halves; 5 instead of 9 matrix instructions per tap row.
__device__ __forceinline__ i32x8 nb_cat8(h8 a, h8 b) {
    const i32x4 x = __builtin_bit_cast(i32x4, a), y = __builtin_bit_cast(i32x4, b);
    return i32x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
}

// two values -> two fp8 bytes (low 16 bits).  CLAMP: saturate to the e4m3 range first (needed for x/4 when |x| > 1792;
// xl * 2^9 = (x - f16(x)) * 512 <= |x| / 4 is covered by the same bound and is clamped as well where x is not)
template <bool CLAMP>
__device__ __forceinline__ unsigned nb_pk2_fp8(float a, float b) {
    if (CLAMP) { a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f); }
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}

// four values -> four fp8 bytes in a wave whose MODE.FP16_OVFL is set (nb_set_fp16_ovfl): the conversion itself saturates
// to +-448 (and f32 -> f16 to +-65504), no clamp instructions
__device__ __forceinline__ unsigned nb_pk4_fp8_sat(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ void nb_set_fp16_ovfl() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }

// v_permlane32_swap: exchanges a[lanes 32..63] with b[lanes 0..31] in place.  Afterwards lanes 0..31 hold (a, b) = (their
// own a, the upper lanes' a) and lanes 32..63 hold (the lower lanes' b, their own b).  Written as inline assembly with both
// registers read-write: the compiler's builtin (__builtin_amdgcn_permlane32_swap, ROCm 7.2) returned the FIRST result for
// both elements whenever its two results met again in one expression or one of the operands was a constant
// (tools/microbench/permlane_swap_const.hip).  The two wait states cover a VALU write of an operand right before it
// (the hazard the compiler pads with s_nop when it emits the instruction itself).  Needs every lane active.
__device__ __forceinline__ void nb_swap32(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// v - (float)h[hi ? 1 : 0] in one instruction
__device__ __forceinline__ float nb_sub_f16(float v, h2 h, bool hi) {
    float r;
    if (hi) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(v));
    return r;
}

__device__ __forceinline__ unsigned nb_pk4_fp8(float a, float b, float c, float d) {
    auto cl = [](float v) { return fminf(fmaxf(v, -448.f), 448.f); };
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), w, true);
    return (unsigned)w;
}

struct H3Up2Params {
    const _Float16* x;      // H2 [n][c8][2][H][W][8]
    const _Float16* wts;    // [nchunks][9][2][2][co_ld][8]
    const float* dcoefs; const float* noise; const float* bias; float* y; const float* zeros;
    NbNoiseSrcDev nsrc;     // see H3Params
    long long noise_stride_n;
    int c8, nchunks, c_out, co_ld, h, w;
    int tiles_x, tiles_y, slices, dbg, stagger_ticks;
    float alpha, gain, clamp;
    unsigned long long* tstamps;
    _Float16* yh2; const float* next_styles; int next_stride, c8_next, out_f8;      // H2 output (see H3Params)
};
