/*
 * neube_hip_debug.h -- developer / test hooks of libneube_hip.so.  NOT part of the drop-in boundary
 * (include/neube_hip.h): nothing in the product path calls these.  They are process-global switches that the
 * parity tests and the profiling tools use to force a kernel variant or to collect in-kernel timestamps; a
 * caller sets one, runs, and restores the default (0 / -1 / NULL) afterwards.  Thread-compatible only in the
 * sense that no product call changes them.
 */
#ifndef NEUBE_HIP_DEBUG_H
#define NEUBE_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Tile form of the 8-wave split-f16 up=1 kernel: 0 = automatic (by workgroup count), 1 / 2 = that many 32-pixel rows per
 * wave.  tests/test_hip_f8.py asserts both forms bit-identical. */
void nb_debug_set_up1_rows(int nbw);

/* K loop of the 8-wave split-f16 up=1 kernel with f8 operands: -1 = automatic (the software-pipelined loop of round 4), 0 = the
 * round-3 loop, 1 = the software-pipelined loop.  tests/test_hip_f8.py asserts both bit-identical. */
void nb_debug_set_up1_v2(int mode);
/* ... its ping-pong form (round 5: waves 4-7 one segment behind waves 0-3, a load segment against a compute segment on every SIMD):
 * -1 = automatic (ON since the end of round 5: launches -2.5 %, step +0.5 ... +1.0 %), 0 / 1 = off / on.  Bit-identical
 * (tests/test_hip_f8.py). */
void nb_debug_set_up1_pp(int mode);

/* K-splitting waves per workgroup of the small-image kernel (modconv3x3_up1_small_h3): 0 = automatic (8 for layers of >= 8
 * sixteen-channel chunks), 4 or 8 = force. */
void nb_debug_set_small_waves(int waves);
/* ... and 32-position blocks per tile: 0 = automatic (2 when the launch is more than a round of workgroups), 1 or 2 = force
 * (2 only where the form exists: eight waves, one sample per tile). */
void nb_debug_set_small_blocks(int blocks);

/* Tile height of the split-f16 up=2 kernel: 0 = automatic, 12 = throughput tiles, 8 = the tiles of launches whose 12-row tiles
 * would end in a mostly empty round of workgroups, 5 = the under-filled (batch-1) tiles. */
void nb_debug_set_up2_tile(int tqh);

/* Workgroup form of the split-f16 up=2 kernel: -1 = automatic, 0 = 8 waves / 12 x 32 tiles / 3 LDS stages (one workgroup per
 * CU), 1 = 4 waves / 12 x 16 tiles / 2 stages (two per CU).  tests/test_hip_f8.py asserts both bit-identical. */
void nb_debug_set_up2_pair(int mode);

/* The 8-wave split-f16 up=2 kernel with the software-pipelined K loop (csrc/nb_modconv_up2v.hip; f8 operands, 12 x 32 tiles):
 * -1 = automatic (every f8 launch that takes the 12-row tiles), 0 = never (the round-3 kernel), 1 = wherever the shape allows.
 * tests/test_hip_f8.py asserts the two bit-identical. */
void nb_debug_set_up2_v2(int mode);

/* Workgroups of the 8-wave split-f16 up=1 kernel: 1 = persistent -- a few per CU, each walking its share of the launch's tiles with
 * the NEXT tile's prologue (halo tile + three weight sub-chunks) issued ahead of the current tile's epilogue --, -1 (default) / 0 = one
 * workgroup per tile, on a loop-less instantiation of the same kernel (template parameter PERSIST: against round 5's kernels in the
 * same library the persistent form measured 4.4 % slower at 64 channels, equal at 128).  tests/test_hip_f8.py asserts the two bit-identical. */
void nb_debug_set_up1_persistent(int mode);

/* Workgroups of the 12-row software-pipelined up=2 kernel: -1 (default) / 1 = persistent -- a few per CU, each walking its share of the
 * launch's tiles with the NEXT tile's first chunk prefetched under the current tile's epilogue --, 0 = one workgroup per tile on the
 * loop-less instantiation (PERSIST = false).  tests/test_hip_f8.py asserts the two bit-identical. */
void nb_debug_set_up2v_persistent(int mode);

/* Workgroups per CU of the persistent launches (both kernels above): default 4 (<= 0 restores it) -- a workgroup walks 2-4 tiles of the
 * BASELINE launches and CUs come free four times per launch for other streams' kernels; 1 = every workgroup resident from the start
 * (`profiles/r06_ab_persistent_grid.txt`). */
void nb_debug_set_persistent_wgs_per_cu(int k);

/* Tile form of nb_enc_conv3x3_h3: -1 = automatic, 0 = large tiles, 1 = small split-K tiles. */
void nb_debug_set_enc_small(int mode);

/* Feature-off bit mask of the conv kernels for same-box A/B runs (0 on the product path): 8 = no XCD-aware workgroup order,
 * 32 = the round-3 order of the up=2 launches, ... (the uses of `p.dbg` in csrc/).  Until round 5 the launchers read the environment
 * variable NB_DEBUG themselves; since round 6 the library reads NO environment variable -- tools/nb_debug_env.py maps the developer
 * variables (NB_DEBUG, NB_STAGGER, NB_UP1_PP, NB_UP2_TQH, ...) onto these setters for the A/B scripts. */
void nb_debug_set_flags(int flags);
/* First-round stagger of the large split-f16 kernels: workgroup b sleeps (b % 8) * ticks s_sleep units before its prologue; 0 = off. */
void nb_debug_set_stagger(int ticks);
/* The 4-wave two-workgroups-per-CU form of the H2 up=1 kernel: -1 = automatic (images <= 32 x 32), 0 / 1 = never / always. */
void nb_debug_set_up1_small(int mode);
/* Workgroups the weight-gradient launches aim for (default 256 = one per CU; <= 0 restores it); tools/bench_wgrad.py. */
void nb_debug_set_wgrad_wgs(int wgs);
/* 1 = every nb_upfirdn2d_f32 call on the run-time-everything kernel (tests compare it with the specialised ones), 0 = automatic. */
void nb_debug_set_upfirdn_generic(int on);

/* Per-workgroup phase timestamps of the next split-f16 (resp. fp32 split-K) conv launches: buf = device pointer to
 * capacity_workgroups x 8 uint64 (s_memrealtime ticks; layout: tools/phase_times.py), NULL = off. */
void nb_debug_set_timestamps(void* buf, int capacity_workgroups);
void nb_debug_set_timestamps_f32(void* buf, int capacity_workgroups);
/* The same for nb_enc_conv3x3_h3's large-tile kernel (slots: 0 start, 1 first step may begin -- f8 operands, stride 2 only --, 2 K loop
 * done, 3 epilogue values staged in LDS, 4 end; tools/phase_times_enc.py). */
void nb_debug_set_enc_timestamps(void* buf, int capacity_workgroups);

/* The library's 64 KiB device page of zeros (source of out-of-image halo slots for the LDS-DMA staging); lazily
 * allocated, one per process. */
const float* nb_zero_page_ptr(void);

#ifdef __cplusplus
}
#endif
#endif
