#!/bin/bash
# Collect the round's judged artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1800 -- "bash tools/collect_profiles.sh r03 $(git rev-parse --short HEAD)"   (no git on the GPU box)
# Writes gpurun_out/<tag>/...; copy what is to be judged into profiles/ (tools/copy_profiles.sh <tag>).
TAG=${1:-r06}; GIT_HEAD=${2:-unknown}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the driver's command (one line carrying the arithmetic modes; `value` on the concurrent schedule, roofline on its single-stream leg), before anything else touches the GPU; it runs once more at the
# very end, when the counter passes have written the traffic file for THESE kernel sources (a bench.py run right after the
# rocprofv3 --pmc passes measured its auxiliary three-streams leg 19 % low, twice: the counters leave the clocks in a state of
# their own for a while)
(cd /tmp && python3 $R/bench.py --detail $O/bench_pre_detail.json > $O/bench_pre.json 2> $O/bench_pre.err)
python3 $R/bench.py --res 128 --no-cpu --modes primary --detail $O/bench_r128_f8_detail.json > $O/bench_r128_f8.json 2> $O/bench_r128.err
# per-kernel averages of the same command (timing pass: kernel trace + stats only), per mode
for m in f8 h3 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-latency --schedule single --modes primary --conv-mode $m > $O/stats_$m.log 2>&1
  cp $(ls $O/stats_$m/*/*kernel_stats.csv | head -1) $O/kernel_stats_$m.csv 2>/dev/null
done
# counters: each group in its own pass, with the kernel trace only; memory counters per mode, MFMA-busy for the split modes
for m in f8 h3 f32; do
  for c in "hit:TCC_HIT_sum TCC_MISS_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    n=${c%%:*}
    rocprofv3 --kernel-trace --pmc ${c#*:} --output-format csv -d $O/pmc_${n}_$m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-latency --schedule single --modes primary --conv-mode $m > $O/pmc_${n}_$m.log 2>&1
  done
  (cd $R && python tools/pmc_mem_summary.py $O/pmc_hit_$m $O/pmc_fetch_$m $O/pmc_write_$m $O/pmc_mem_$m.json > $O/pmc_mem_$m.txt 2>&1)
done
for m in f8 h3; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_$m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-latency --schedule single --modes primary --conv-mode $m > $O/pmc_mfma_$m.log 2>&1
  (cd $R && python tools/pmc_summary.py $O/pmc_mfma_$m > $O/pmc_mfma_busy_$m.txt 2>&1)
done
cd $R
python tools/make_hbm_traffic.py $O/hbm_traffic.json "${GIT_HEAD:-unknown}" f8=$O/pmc_mem_f8.json h3=$O/pmc_mem_h3.json f32=$O/pmc_mem_f32.json > $O/hbm_traffic.log 2>&1
cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json
NB_PHASE_FMT=1 NB_PHASE_H2OUT=1 python tools/phase_times.py > $O/phase_times.txt 2>&1
python tools/phase_times_enc.py > $O/phase_times_encoder.txt 2>&1
# round 6: launch times of the four large layers (40-launch loops), the persistent workgroups against one workgroup per tile (same box,
# alternating: per-kernel ms of bench.py's calibration pass, both schedules' step rates; phase stamps of the up=1 launches), the number of
# streams of the concurrent schedule, the first-round stagger (no-go)
NB_FMTS=1 python tools/bench_f6_layers.py 2>&1 | grep "^up" > $O/large_layers.txt
bash tools/ab_persist.sh > $O/ab_persistent.txt 2>&1
bash tools/ab_streams.sh > $O/ab_streams.txt 2>&1
# library-level A/B against the two features that were measured as gains by in-build switches and reverted (tools/build_variant_at.sh builds them from e618c53)
[ -f brushstroke_engine_amd/csrc/libneube_r05k.so ] && bash tools/ab_variants.sh "shipped r05k wrap early" 2>/dev/null | grep "patches/s" > $O/ab_variants.txt      # (r05k: tools/build_variant_at.sh r05k c6a9fb7 nb_modconv_h3.hip,nb_modconv_up2v.hip,nb_modconv_up2w.hip,nb_h3_common.h,nb_common.h,nb_torgb.h)
python tools/ab_positions_once.py 2>/dev/null | grep "per tile" > $O/ab_positions_once.txt     # integer positions normalised once per batch instead of at the top of every tile
python tools/ab_noise_in_kernel.py 2>/dev/null | grep "noise images" > $O/ab_noise_in_kernel.txt  # the large layers' noise: computed in the kernels (default) against the noise launch's images    # up2v: the next tile's epilogue operands and noise under the epilogue (NB_DEBUG=512: off)
PAIRS=2 bash tools/ab_env.sh NB_STAGGER=200 > $O/ab_stagger.txt 2>&1
python tools/stress_persistent.py > $O/stress_persistent.txt 2>&1           # race hunt: 2 400 persistent launches against the one-workgroup-per-tile results
# the N > 1 code through RCCL at world size 1 (NB_FORCE_PG=1): bench.py, the lamali canvas, the training step
NB_FORCE_PG=1 python bench.py --modes primary --no-cpu --no-latency --detail $O/rccl_world1_bench_detail.json > $O/rccl_world1_bench.json 2> $O/rccl_world1_bench.err
NB_FORCE_PG=1 python tools/bench_lamali.py --steps 3 2> $O/rccl_world1_lamali.err | grep "^{" > $O/rccl_world1_lamali.json
NB_FORCE_PG=1 python tools/bench_train.py --iters 8 --warmup 2 2> $O/rccl_world1_train.err | grep "^{" > $O/rccl_world1_train.json
bash $R/tools/microbench/run_all.sh > $O/microbench.txt 2>&1
python tools/latency_stroke.py > $O/latency_stroke.txt 2>&1
NB_SUBS="1" NB_STEPS=40 bash tools/run_step_trace.sh > $O/step_trace.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && cd $R && NB_SUB=1 NB_STEPS=40 rocprofv3 --kernel-trace -d $O/steptl -o st --output-format csv -- python3 tools/trace_step_loop.py > /dev/null 2>&1; python3 tools/trace_step_timeline.py $O/steptl 20 > $O/step_timeline.txt 2>&1; rm -rf $O/steptl)
(cd /tmp && rocprofv3 --kernel-trace -d $O/b1trace -o b1 --output-format csv -- python3 $R/tools/trace_b1.py > $O/b1trace.log 2>&1)
python tools/trace_b1_summary.py $O/b1trace > $O/b1_trace_summary.txt 2>&1; rm -rf $O/b1trace
(cd /tmp && rocprofv3 --kernel-trace -d $O/enctrace -o enc --output-format csv -- python3 $R/tools/trace_encoder.py > $O/enctrace.log 2>&1)
(grep '^encoder' $O/enctrace.log; python tools/trace_encoder_summary.py $O/enctrace) > $O/encoder_trace.txt 2>&1; rm -rf $O/enctrace
python tools/bench_canvas.py --size 4096 --res 256 --level 2 --steps 3 --breakdown > $O/canvas_4096_r256_l2.json 2>/dev/null
# the same job with 1, 2 and the probed number of batch streams, one process
python tools/bench_canvas.py --size 4096 --res 256 --level 2 --steps 3 --streams ab > $O/canvas_4096_r256_l2_streams_ab.json 2>/dev/null
python tools/bench_canvas.py --size 4096 --res 256 --level 0 --steps 3 --breakdown > $O/canvas_4096_r256_l0.json 2>/dev/null
python tools/bench_canvas.py --size 1024 --res 128 --level 2 --steps 3 --breakdown > $O/canvas_1024_r128_l2.json 2>/dev/null
python tools/bench_lamali.py > $O/lamali.json 2> $O/lamali.err
# the N > 1 launch path of the canvas job on this one-GPU box: two ranks share device 0 over gloo (functional evidence: halo
# exchange, pieces replay, gather; the rate means nothing)
NB_BENCH_SHARE_GPU=1 NB_BENCH_BACKEND=gloo python tools/bench_lamali.py --gpus 2 --steps 3 2> $O/lamali_2ranks.err | grep "^{" > $O/lamali_2ranks_shared_gpu.json
python tools/bench_train.py > $O/train_bench.json 2> $O/train_bench.err
bash tools/trace_train.sh 400 > /dev/null 2>&1; head -70 gpurun_out/train_trace_summary.txt > $O/train_trace.txt; python tools/trace_train_agg.py > $O/train_trace_by_kernel.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
(cd /tmp && python3 $R/bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err)
rm -rf $O/stats_* $O/pmc_hit_* $O/pmc_fetch_* $O/pmc_write_* $O/pmc_mfma_f8 $O/pmc_mfma_h3
ls $O; tail -c 400 $O/bench.json; tail -2 $O/smoke.txt
