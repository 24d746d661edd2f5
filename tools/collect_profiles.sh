#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/prof_$1 (copy what is to be
# judged into profiles/ afterwards):  kernel stats of the bench command, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes), SQ wave-state counters, big-FFT timings.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r05'
set -u
tag=${1:-r06}
out=gpurun_out/prof_$tag
export TMPDIR=/tmp
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
python3 bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 > $out/bench_driver_style.json 2> /dev/null
# per-launch kernel times: ONE buffer per launch, one launch behind the other (what rounds 1-4 listed), then the benchmarked
# form (four buffers per launch, overlapped)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -- python3 bench.py --no-oracle --no-extra --no-pipeline --batch 1 > $out/bench_profiled.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace_batch -- python3 bench.py --no-oracle --no-extra --no-pipeline > $out/bench_profiled_batch.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace_pipelined -- python3 bench.py --no-oracle --no-extra > $out/bench_profiled_pipelined.json 2> /dev/null
K="chain_batch4 chain_fft convert shift_gain conv chain_c64 beamform downsample scale fftbig16 fftbig18"
REPS=6 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 tools/prof_kernels.py $K > /dev/null 2>&1
REPS=6 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 tools/prof_kernels.py $K > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/traffic.json > /dev/null
REPS=6 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq1 -- python3 tools/prof_kernels.py chain chain_fft conv chain_c64 > /dev/null 2>&1
REPS=6 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/sq2 -- python3 tools/prof_kernels.py chain chain_fft conv chain_c64 > /dev/null 2>&1
python3 tools/pmc_sq.py $out/sq1 $out/sq2 > $out/sq_counters.txt
REPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fft_trace -- python3 tools/prof_kernels.py fft1024 fft4096 fftbig13 fftbig14 fftbig15 fftbig16 fftbig18 fftbig20 > /dev/null 2>&1
REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kern_trace -- python3 tools/prof_kernels.py $K chain shift rotate > /dev/null 2>&1
python3 tools/host_path_bench.py > $out/host_path.txt 2>&1
tools/bin/fir_ablate > $out/fir_ablate.txt 2>&1
# the matrix pipe's share of the kernel: one buffer per launch, and the benchmarked four
REPS=6 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/sq3 -- python3 tools/prof_kernels.py chain > /dev/null 2>&1
echo "== one buffer per launch" >> $out/sq_counters.txt
python3 tools/pmc_sq.py $out/sq3 >> $out/sq_counters.txt
REPS=6 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/sq4 -- python3 tools/prof_kernels.py chain_batch4 > /dev/null 2>&1
echo "== four buffers per launch (hzsdr_chain_run_batch)" >> $out/sq_counters.txt
python3 tools/pmc_sq.py $out/sq4 >> $out/sq_counters.txt
tools/bin/mfma_fir > $out/mfma_fir.txt 2>&1
BISECT=1 tools/bin/mfma_fir2 > $out/mfma_fir2.txt 2>&1
AB=1 tools/bin/mfma_fir2 > $out/mfma_fir2_ab.txt 2>&1
BATCH=4 tools/bin/mfma_fir2 > $out/mfma_fir2_batch4.txt 2>&1
tools/bin/mfma_rate > $out/mfma_rate.txt 2>&1
# round 5: what an epilogue's instructions cost beside the partner's matrix loop; the streaming kernels' access forms
tools/bin/epi_cost > $out/epi_cost.txt 2>&1
tools/bin/stream_rate > $out/stream_rate.txt 2>&1
tools/bin/copy_rate > $out/copy_rate.txt 2>&1
python3 tools/pipeline_time.py > $out/pipeline_time.txt 2>&1
# the packed-float32 hazard of round 4 (the instruction alone; the kernel as built)
tools/bin/pk_glitch 50000 > $out/pk_glitch.txt 2>&1
tools/bin/mm2_glitch 20000 2>&1 | cut -c1-600 > $out/mm2_glitch.txt
tools/bin/nco_ablate > $out/nco_ablate.txt 2>&1
python3 tools/conv_time.py 2>&1 | grep " us" > $out/conv_time.txt
timeout 900 python3 tools/repeat_check.py 300 > $out/repeat_check.txt 2>&1
PIPELINE=1 timeout 900 python3 tools/repeat_check.py 200 > $out/repeat_check_pipelined_run.txt 2>&1
BATCH=1 timeout 900 python3 tools/repeat_check.py 200 > $out/repeat_check_batch_run.txt 2>&1
python3 tools/firmm_probe.py 2>/dev/null | grep "path\|mean\|calls" > $out/firmm_probe.txt
# round 6: per-pass stamps of the benchmarked form, the clock the chip holds (random bytes / constant input), what the chip
# sustains of bare MFMAs by operand content and duty, the round's A/Bs, config 2 and Downsample from HBM
(cd tools; BATCH=4 PASSES=1 ./bin/mfma_fir2 > ../$out/pass_breakdown.txt 2>&1; ZERO=1 BATCH=4 PASSES=1 ./bin/mfma_fir2 > ../$out/pass_breakdown_zero.txt 2>&1
 for v in "" ZERO_IN=1 ZERO_TAB=1 ZERO=1; do echo "== $v"; env $v BATCH=4 PASSES=1 SPLIT=1 ./bin/mfma_fir2 2>&1 | grep "MIX\|shader clock"; done > ../$out/power_split.txt 2>&1
 ./bin/mfma_power > ../$out/mfma_power.txt 2>&1
 BATCH=4 AB6=1 ./bin/mfma_fir2 > ../$out/ab6_batch4.txt 2>&1; BATCH=4 AB7=1 ./bin/mfma_fir2 2>&1 | grep MIX > ../$out/ab7.txt; BATCH=8 AB6=1 ./bin/mfma_fir2 2>&1 | grep MIX > ../$out/ab6_batch8.txt)
python3 tools/shift_time.py 2>&1 | grep -v amdgpu > $out/shift_time_final.txt
python3 tools/downsample_time.py 2>&1 | grep -v amdgpu > $out/downsample_time_final.txt
for d in 16; do for t in 1024 512 256; do for impl in 0 2; do PROBE_D=$d PROBE_TAPS=$t PROBE_IMPL=$impl python3 tools/firmm_probe.py 2>/dev/null | head -2; done; done; done > $out/firmm_probe_d16.txt
python3 tools/fftbig_time.py 2>&1 | grep "N =" > $out/fftbig_time.txt
(python3 tools/clock_watch.py; HZ_QUIET_INPUT=1 python3 tools/clock_watch.py) 2>&1 | grep -v amdgpu > $out/clock_watch.txt
for f in $out/bench_trace/*/*kernel_stats.csv $out/bench_trace_batch/*/*kernel_stats.csv $out/kern_trace/*/*kernel_stats.csv; do echo "== $f"; cut -d, -f1-4 $f | cut -c1-160 | head -14; done
cat $out/sq_counters.txt | tail -30
tail -3 $out/bench.err
