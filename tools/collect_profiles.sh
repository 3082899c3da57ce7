#!/bin/bash
# Collect the round's judged artefacts on the GPU box (run through gpurun from the repo root).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_f8.json 2> $O/bench_f8.err
python3 $R/bench.py --conv-mode h3 --no-cpu > $O/bench_h3.json 2> $O/bench_h3.err
python3 $R/bench.py --conv-mode f32 --no-cpu > $O/bench_f32.json 2> $O/bench_f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-latency > $O/stats.log 2>&1
for c in "hit:TCC_HIT_sum TCC_MISS_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  n=${c%%:*}
  rocprofv3 --kernel-trace --pmc ${c#*:} --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-latency > $O/pmc_$n.log 2>&1
done
cd $R
python tools/pmc_mem_summary.py $O/pmc_hit $O/pmc_fetch $O/pmc_write $O/pmc_mem.json > $O/pmc_mem.txt 2>&1
python tools/pmc_summary.py $O/pmc_mfma > $O/pmc_mfma_busy.txt 2>&1
python tools/phase_times.py > $O/phase_times.txt 2>&1
python tools/phase_times_f32.py > $O/phase_times_f32.txt 2>&1
python tools/latency_stroke.py > $O/latency_stroke.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace -d $O/b1trace -o b1 --output-format csv -- python3 $R/tools/trace_b1.py > $O/b1trace.log 2>&1)
python tools/trace_b1_summary.py $O/b1trace > $O/b1_trace_summary.txt 2>&1; rm -rf $O/b1trace
python tools/graph_b32.py > $O/graph_b32.txt 2>&1
python tools/bench_canvas.py --size 4096 --res 256 --level 2 --steps 3 --conv-mode f8 > $O/canvas_4096_r256_l2_f8.json 2>/dev/null
python tools/bench_canvas.py --size 4096 --res 256 --level 2 --steps 3 --breakdown > $O/canvas_4096_r256_l2.json 2>/dev/null
python tools/bench_canvas.py --size 4096 --res 256 --level 0 --steps 3 --breakdown > $O/canvas_4096_r256_l0.json 2>/dev/null
python tools/bench_canvas.py --size 1024 --res 128 --level 2 --steps 3 --breakdown > $O/canvas_1024_r128_l2.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
ls $O/stats/*/ | head; tail -1 $O/bench_f8.json | cut -c1-300; cat $O/smoke.txt | tail -2
