#!/bin/bash
# Collect the round's judged artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r02'
# Writes gpurun_out/<tag>/...; copy what is to be judged into profiles/ (tools/copy_profiles.sh <tag>).
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_f8.json 2> $O/bench_f8.err
python3 $R/bench.py --conv-mode h3 --no-cpu > $O/bench_h3.json 2> $O/bench_h3.err
python3 $R/bench.py --conv-mode f32 --no-cpu --no-latency > $O/bench_f32.json 2> $O/bench_f32.err
python3 $R/bench.py --res 128 --no-cpu > $O/bench_r128_f8.json 2> $O/bench_r128.err
# per-kernel averages of the same command (timing pass: kernel trace + stats only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-latency > $O/stats.log 2>&1
# counters: each group in its own pass, with the kernel trace only
for c in "hit:TCC_HIT_sum TCC_MISS_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  n=${c%%:*}
  rocprofv3 --kernel-trace --pmc ${c#*:} --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-latency > $O/pmc_$n.log 2>&1
done
cd $R
python tools/pmc_mem_summary.py $O/pmc_hit $O/pmc_fetch $O/pmc_write $O/pmc_mem.json > $O/pmc_mem.txt 2>&1
python tools/make_hbm_traffic.py $O/pmc_mem.json $O/hbm_traffic.json > $O/hbm_traffic.log 2>&1
python tools/pmc_summary.py $O/pmc_mfma > $O/pmc_mfma_busy.txt 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
NB_PHASE_F8=1 NB_PHASE_H2OUT=1 python tools/phase_times.py > $O/phase_times.txt 2>&1
python tools/latency_stroke.py > $O/latency_stroke.txt 2>&1
NB_SUBS="1 2" NB_STEPS=40 bash tools/run_step_trace.sh > $O/step_trace.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace -d $O/b1trace -o b1 --output-format csv -- python3 $R/tools/trace_b1.py > $O/b1trace.log 2>&1)
python tools/trace_b1_summary.py $O/b1trace > $O/b1_trace_summary.txt 2>&1; rm -rf $O/b1trace
python tools/bench_canvas.py --size 4096 --res 256 --level 2 --steps 3 --breakdown > $O/canvas_4096_r256_l2.json 2>/dev/null
python tools/bench_canvas.py --size 4096 --res 256 --level 0 --steps 3 --breakdown > $O/canvas_4096_r256_l0.json 2>/dev/null
python tools/bench_canvas.py --size 1024 --res 128 --level 2 --steps 3 --breakdown > $O/canvas_1024_r128_l2.json 2>/dev/null
python tools/bench_lamali.py > $O/lamali.json 2> $O/lamali.err
python tools/bench_train.py > $O/train_bench.json 2> $O/train_bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
rm -rf $O/stats $O/pmc_hit $O/pmc_fetch $O/pmc_write $O/pmc_mfma
ls $O; tail -c 400 $O/bench_f8.json; tail -2 $O/smoke.txt
