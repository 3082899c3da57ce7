// Shared device helpers of the split-f16 ("h3" / "f8") convolution kernels: vector types, LDS-DMA, counted waits, fp8 packing,
// lane-half exchange, the up=2 kernels' parameter block.  Included by nb_modconv_h3.hip and nb_modconv_up2v.hip.
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
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int v6i __attribute__((ext_vector_type(6)));

// The matrix instructions of the f8 K loops.  -DNB_MOCK16 (developer TIMING experiment, wrong results; tools/build_variant.sh, profiles/
// r06_ab_mfma16.txt): every 32x32 instruction is replaced by TWO 16x16 instructions of half its MACs each on the same operand and
// accumulator registers -- `v_mfma_f32_16x16x32_f16` (2 x 16 cycles for 32) and `v_mfma_scale_f32_16x16x128_f8f6f4` (2 x 32 for 64) --,
// i.e. the instruction stream a 16x16-shaped body with the same 64 x 64 wave tile would execute (same fragment reads per MAC, same
// LDS-DMA, same matrix cycles), to measure what the shape is worth in wall time (the clock the chip holds: MI355X_MICROARCH.md, DVFS
// give-back item 7) BEFORE the operand layouts, pairings and epilogues of such a body are written.
#ifdef NB_MOCK16
__device__ __forceinline__ f32x16 nb_mock16_f16(h8 a, h8 b, f32x16 c, int q) {
    f32x4 c0, c1;
    const int o = (q & 1) * 8;
#pragma unroll
    for (int r = 0; r < 4; ++r) { c0[r] = c[o + r]; c1[r] = c[o + 4 + r]; }
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { c[o + r] = c0[r]; c[o + 4 + r] = c1[r]; }
    return c;
}
__device__ __forceinline__ f32x16 nb_mock16_fp8(i32x8 a, i32x8 b, f32x16 c, int sa, int sb, int q) {
    f32x4 c0, c1;
    const int o = (q & 1) * 8;
#pragma unroll
    for (int r = 0; r < 4; ++r) { c0[r] = c[o + r]; c1[r] = c[o + 4 + r]; }
    c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, sa, 0, sb);
    c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, sa, 0, sb);
#pragma unroll
    for (int r = 0; r < 4; ++r) { c[o + r] = c0[r]; c[o + 4 + r] = c1[r]; }
    return c;
}
#define NB_MFMA_F16(a, b, c, q) nb_mock16_f16(a, b, c, q)
#define NB_MFMA_FP8(a, b, c, sa, sb, q) nb_mock16_fp8(a, b, c, sa, sb, q)
#else
#define NB_MFMA_F16(a, b, c, q) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define NB_MFMA_FP8(a, b, c, sa, sb, q) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb)
#endif

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
// ... of a x 2^k, b x 2^k, ... with the scaling inside the conversion (v_cvt_scalef32_pk_fp8_f32 divides by its scale operand: pass
// 2^-k).  Bit-identical to "multiply, then nb_pk4_fp8_sat" for every finite input, denormal results and saturation included
// (tools/microbench/cvt_scale_fp8.hip: 1M random pairs per scaling, with and without FP16_OVFL; only +-inf differs -- 448 here, NaN
// there -- and the epilogues' values are clamped), at two packed multiplies less per four values.
typedef short nb_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned nb_pk4_fp8_sat_scaled(float a, float b, float c, float d, float inv_scale) {
    nb_s16x2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, a, b, inv_scale, false);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, c, d, inv_scale, true);
    return __builtin_bit_cast(unsigned, w);
}
// ---- "f6" operand format (round 5): the two correction products on ONE block-scaled fp6 (e2m3) MFMA per tap pair, at the f16
// instruction's cycles (the fp8 form takes twice as many).  e2m3 has the three mantissa bits of e4m3 but a range of only 2^6, so
// every 16-channel chunk of a pixel carries its own power-of-two scale S (E8M0 byte), chosen so that the chunk's largest |x| lands
// in [4, 8) (7.5 is the format's top: values in (7.5 S, 8 S) saturate -- an error of at most one step at the top of the range):
//   lo slots of a chunk (32 bytes per pixel: (cg 2k, lo) ++ (cg 2k+1, lo)) = six dwords of 32 six-bit fields, the scale byte, zeros;
//   field 2 i = e2m3(xl[ch(i)] 2^11 / S),  field 2 i + 1 = e2m3(x[ch(i)] / S),   ch(i) = channels 0-3, 8-11, 4-7, 12-15 of the chunk
//   (the order in which the two lane halves of a 32x32 accumulator hold a chunk's rows);
//   weights likewise per (c_out, tap, chunk): field 2 i = e2m3(w[ch(i)] / Sw), field 2 i + 1 = e2m3(wl[ch(i)] 2^11 / Sw), byte = Sw 2^-11.
// v_mfma_scale_f32_32x32x64_f8f6f4 (cbsz = blgp = 2): a lane's 32 K values = ONE pixel / c_out, ONE tap, both terms (the hardware
// applies one scale per lane); lane half 0 contracts the first tap of a pair, lane half 1 the second.  The operand tuple is read
// straight from the two slots (8 registers: six fields, the scale dword -- the instruction's scale operand --, one unused).
// Layouts, scale semantics and the cycle count are checked by tools/microbench/mfma_f6_check.hip.
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
// 32 floats -> 32 e2m3 codes: field 2 i = a[i] / 2^e, field 2 i + 1 = b[i] / 2^e (e = the exponent of `scale`; round to nearest even,
// saturating at +-7.5).  Inline assembly with an EARLY-CLOBBER destination: through the builtin, hipcc (ROCm 7.2) allocated the six
// result registers on top of the last two registers of the second source (v[32:37] <- v[2:17], v[18:33]) and the last two inputs
// came out as garbage (tools/microbench/mfma_f6_check.hip, first version).
__device__ __forceinline__ u32x6 nb_cvt_fp6x32(f32x16 a, f32x16 b, float scale) {
    u32x6 r;
    asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(scale));
    return r;
}
// ... of 8 + 8 values (fields 0 .. 15 = the first three dwords; the upper halves of the sources are left undefined)
__device__ __forceinline__ u32x6 nb_cvt_fp6x16(f32x8 a, f32x8 b, float scale) {
    return nb_cvt_fp6x32(__builtin_shufflevector(a, a, 0, 1, 2, 3, 4, 5, 6, 7, -1, -1, -1, -1, -1, -1, -1, -1),
                         __builtin_shufflevector(b, b, 0, 1, 2, 3, 4, 5, 6, 7, -1, -1, -1, -1, -1, -1, -1, -1), scale);
}
// the converter's scale operand for a block whose largest magnitude is m (only its exponent counts: m / 2^e lands in [4, 8)), and the
// E8M0 byte that undoes it in the MFMA
__device__ __forceinline__ float nb_f6_scale(float maxabs) { return maxabs * 0.25f; }
__device__ __forceinline__ unsigned nb_f6_scale_byte(float scale) { return (__builtin_bit_cast(unsigned, scale) >> 23) & 0xffu; }
// channel of a 16-channel chunk behind field pair i
__device__ __forceinline__ constexpr int nb_f6_ch(int i) { return (i & 3) + 8 * ((i >> 2) & 1) + 4 * (i >> 3); }
// 16 channels of one pixel (already times the consumer's style) -> the chunk's four slots: hi f16 of channels 0-7 / 8-15, lo slots
__device__ __forceinline__ void nb_f6_encode16(const float (&v)[16], h8& hi0, h8& hi1, i32x4& l0, i32x4& l1) {
    f32x16 a, b;
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const _Float16 hh = (_Float16)v[j];
        if (j < 8) hi0[j & 7] = hh; else hi1[j & 7] = hh;
        m = fmaxf(m, fabsf(v[j]));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ch = nb_f6_ch(i);
        const _Float16 hh = ch < 8 ? hi0[ch & 7] : hi1[ch & 7];
        a[i] = (v[ch] - (float)hh) * 2048.f;
        b[i] = v[ch];
    }
    const float sc = nb_f6_scale(m);
    const u32x6 r = nb_cvt_fp6x32(a, b, sc);
    l0 = i32x4{(int)r[0], (int)r[1], (int)r[2], (int)r[3]};
    l1 = i32x4{(int)r[4], (int)r[5], (int)nb_f6_scale_byte(sc), 0};
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

// Hand-off epilogue of the up=1 kernels (H2 or "f8" output into the consumer's operand tensor), straight from the
// accumulators: no LDS image, no barrier.  D[row = c_out, col = pixel]: a lane holds, for ITS pixel, four consecutive
// channels (4 lh .. 4 lh + 3) of each 8-channel group g; activation / consumer style / hi-lo split run packed over those
// four; then lanes l and l+32 (same pixel, the two halves of every group) trade groups with v_permlane32_swap -- the lower
// lane ends up with all 8 channels of the even groups, the upper lane with the odd ones -- and every lane stores whole
// 16-byte slots (8-byte halves of the f8 lo slots) at consecutive pixels: 512 contiguous bytes per half-wave.
// s_dco / s_bias / s_nst: the workgroup's per-channel tables (16-byte aligned); colbase = first channel of the wave's
// 64-row band within them; trow0 = the wave's first tile row.
// (The kernel parameters it needs come by value: handing the __global__ function's parameter struct on by reference makes
//  the compiler keep a copy of it in scratch memory.)
struct H3HandoffArgs {
    _Float16* yh2;
    int c8_next, c_out, h, w, out_f8, dbg;
    float alpha, gain, clamp;
};
__device__ __forceinline__ H3HandoffArgs nb_handoff_args(_Float16* yh2, int c8_next, int c_out, int h, int w, int out_f8, int dbg, float alpha, float gain,
                                                         float clamp) {
    return H3HandoffArgs{yh2, c8_next, c_out, h, w, out_f8, dbg, alpha, gain, clamp};
}
// F6OUT: the f6 output format is a compile-time form (its extra live values cost the encoder's stem, which shares this epilogue and never
// writes f6, 135 -> 205 us when it was a run-time branch)
template <int MB, int NBW, bool F6OUT = false>
__device__ __forceinline__ void nb_up1_handoff_epilogue(const H3HandoffArgs p, const f32x16 (&acc)[MB][NBW], const float (&nzr)[NBW], const float* s_dco,
                                                        const float* s_bias, const float* s_nst, int colbase, int trow0, int co0, int n, int y0, int x0,
                                                        int lh, int l31) {
    const int W = p.w;
    const size_t HW8 = (size_t)p.h * W * 8;
    const float clampv = p.clamp >= 0.f ? p.clamp : __builtin_inff();
    _Float16* yn = p.yh2 + (size_t)n * p.c8_next * 2 * HW8;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            // g (acc d + noise + bias), lrelu, clamp with the gain g folded into d, noise and bias (lrelu(g t) = g lrelu(t))
            f32x4 d4[2], b4[2], ns4[2];
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int col = colbase + mb * 32 + 8 * (2 * gp + gi) + 4 * lh;
                d4[gi] = *reinterpret_cast<const f32x4*>(s_dco + col) * p.gain;
                b4[gi] = *reinterpret_cast<const f32x4*>(s_bias + col) * p.gain;
                ns4[gi] = *reinterpret_cast<const f32x4*>(s_nst + col);
            }
            const int cg = (co0 + colbase + mb * 32) / 8 + 2 * gp + lh;     // the group this lane owns after the trade
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const float nzg = nzr[nb] * p.gain;
                unsigned hi[2][2], lo[2][2];              // [group of the pair][dword]
                f32x4 wv[2], xlv[2];                      // the values themselves (f6 output converts both groups at once)
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    const int r0 = 4 * (2 * gp + gi);
                    const f32x4 a4 = {acc[mb][nb][r0], acc[mb][nb][r0 + 1], acc[mb][nb][r0 + 2], acc[mb][nb][r0 + 3]};
                    f32x4 t = __builtin_elementwise_fma(a4, d4[gi], b4[gi] + nzg);
                    const f32x4 ta = t * p.alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(t[i], ta[i], __builtin_inff()), -clampv, clampv);
                    const f32x4 w = t * ns4[gi];
                    const h2 h01 = __builtin_convertvector(f32x2{w[0], w[1]}, h2), h23 = __builtin_convertvector(f32x2{w[2], w[3]}, h2);
                    const f32x4 xl = {nb_sub_f16(w[0], h01, false), nb_sub_f16(w[1], h01, true), nb_sub_f16(w[2], h23, false), nb_sub_f16(w[3], h23, true)};
                    hi[gi][0] = __builtin_bit_cast(unsigned, h01); hi[gi][1] = __builtin_bit_cast(unsigned, h23);
                    if constexpr (F6OUT) { wv[gi] = w; xlv[gi] = xl; }
                    if constexpr (F6OUT) {
                        lo[gi][0] = 0; lo[gi][1] = 0;             // (assembled below, once both groups are known)
                    } else if (p.out_f8) {
                        // (conversions saturate: FP16_OVFL is set when out_f8)
                        // (x 2^9 and x 2^-2 inside the conversions: nb_pk4_fp8_sat_scaled)
                        lo[gi][0] = nb_pk4_fp8_sat_scaled(xl[0], xl[1], xl[2], xl[3], 0x1p-9f);
                        lo[gi][1] = nb_pk4_fp8_sat_scaled(w[0], w[1], w[2], w[3], 4.f);
                    } else {
                        const h2 l01 = __builtin_convertvector(f32x2{xl[0], xl[1]}, h2), l23 = __builtin_convertvector(f32x2{xl[2], xl[3]}, h2);
                        lo[gi][0] = __builtin_bit_cast(unsigned, l01); lo[gi][1] = __builtin_bit_cast(unsigned, l23);
                    }
                }
                if constexpr (F6OUT) {
                    // f6 output: this lane holds 8 of the chunk's 16 channels (4 lh .. + 3 and 8 + 4 lh .. + 3: field pairs 8 lh .. 8 lh + 7
                    // of the chunk's 32-field stream), lane l ^ 32 the other 8.  Chunk maximum over both lanes, one conversion of the lane's
                    // 8 + 8 values (three dwords), one dword handed to the lower lane: it stores slot (cg even, lo) = dwords 0-3, the upper
                    // lane slot (cg odd, lo) = dwords 4, 5, the scale byte.
                    float m = 0.f;
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                        for (int i = 0; i < 4; ++i) m = fmaxf(m, fabsf(wv[gi][i]));
                    unsigned m0 = __builtin_bit_cast(unsigned, m), m1 = m0;
                    nb_swap32(m0, m1);                            // (own, partner's) on both halves
                    const float sc = nb_f6_scale(fmaxf(__builtin_bit_cast(float, m0), __builtin_bit_cast(float, m1)));
                    const u32x6 r = nb_cvt_fp6x16(f32x8{xlv[0][0] * 2048.f, xlv[0][1] * 2048.f, xlv[0][2] * 2048.f, xlv[0][3] * 2048.f,
                                                        xlv[1][0] * 2048.f, xlv[1][1] * 2048.f, xlv[1][2] * 2048.f, xlv[1][3] * 2048.f},
                                                  f32x8{wv[0][0], wv[0][1], wv[0][2], wv[0][3], wv[1][0], wv[1][1], wv[1][2], wv[1][3]}, sc);
                    unsigned d3 = r[0], dummy = 0;                // the upper lane's first dword = dword 3 of the stream
                    nb_swap32(d3, dummy);                         // lower lanes: dummy = the upper lane's r[0]
                    lo[0][0] = lh ? r[1] : r[0]; lo[0][1] = lh ? r[2] : r[1];
                    lo[1][0] = lh ? nb_f6_scale_byte(sc) : r[2]; lo[1][1] = lh ? 0u : dummy;
                }
                unsigned ha[2], hb[2], la[2], lb[2];      // a = channels 0-3, b = channels 4-7 of the owned group
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    ha[k] = hi[0][k]; hb[k] = hi[1][k]; la[k] = lo[0][k]; lb[k] = lo[1][k];
                    nb_swap32(ha[k], hb[k]);
                    if constexpr (!F6OUT) nb_swap32(la[k], lb[k]);
                }
                if (cg * 8 < p.c_out && !(p.dbg & 1)) {
                    const size_t pix8 = ((size_t)(y0 + trow0 + nb) * W + x0 + l31) * 8;
                    *reinterpret_cast<u32x4*>(yn + (size_t)(cg * 2) * HW8 + pix8) = u32x4{ha[0], ha[1], hb[0], hb[1]};
                    if constexpr (F6OUT) {
                        // (this lane's own four dwords: lo[0][0], lo[0][1], lo[1][0], lo[1][1] as assembled above)
                        *reinterpret_cast<u32x4*>(yn + (size_t)(cg * 2 + 1) * HW8 + pix8) = u32x4{la[0], la[1], lb[0], lb[1]};
                    } else if (p.out_f8) {
                        // the 16-channel chunk's two lo slots: (even group, lo) = fp8(xl 2^9), (odd group, lo) = fp8(v/4); this
                        // group's 8 channels are bytes 8 (cg & 1) .. + 7 of both
                        _Float16* lo_xl = yn + (size_t)((cg & ~1) * 2 + 1) * HW8 + pix8 + (cg & 1) * 4;
                        *reinterpret_cast<u32x2*>(lo_xl) = u32x2{la[0], lb[0]};
                        *reinterpret_cast<u32x2*>(lo_xl + 2 * HW8) = u32x2{la[1], lb[1]};
                    } else {
                        *reinterpret_cast<u32x4*>(yn + (size_t)(cg * 2 + 1) * HW8 + pix8) = u32x4{la[0], la[1], lb[0], lb[1]};
                    }
                }
            }
        }
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
    int items, items_x;     // up2v (persistent workgroups): items = tiles x slices x samples of the launch, items_x = per sample
};
