#!/bin/bash
# Copies what tools/collect_profiles.sh gathered (gpurun_out/prof_<tag>) into profiles/<tag>_* (tracked).
set -u
tag=${1:-r06}
src=gpurun_out/prof_$tag
cp $src/bench.json profiles/${tag}_bench.json
# (a directory collects one file set per profiled process and run: the newest is this collection's)
newest() { ls -t $1 | head -1; }
cp $(newest "$src/bench_trace/*/*kernel_stats.csv") profiles/${tag}_bench_kernel_stats.csv
[ -d $src/bench_trace_pipelined ] && cp $(newest "$src/bench_trace_pipelined/*/*kernel_stats.csv") profiles/${tag}_bench_pipelined_kernel_stats.csv
[ -d $src/bench_trace_batch ] && cp $(newest "$src/bench_trace_batch/*/*kernel_stats.csv") profiles/${tag}_bench_batch_kernel_stats.csv
python3 tools/split_launches.py $src/bench_trace $src/bench_trace_batch $src/bench_trace_pipelined > profiles/${tag}_bench_launch_split.txt 2>/dev/null || true
[ -f $src/bench_driver_style.json ] && cp $src/bench_driver_style.json profiles/${tag}_bench_driver_style.json
cp $(newest "$src/kern_trace/*/*kernel_stats.csv") profiles/${tag}_kernels_kernel_stats.csv
cp $(newest "$src/fft_trace/*/*kernel_stats.csv") profiles/${tag}_fft_kernel_stats.csv
cp $src/traffic.json profiles/${tag}_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do
  d=$src/pmc_fetch; [ $c = WRITE_SIZE ] && d=$src/pmc_write
  f=$(newest "$d/*/*_counter_collection.csv")
  (head -1 $f; grep "hz::" $f) > profiles/${tag}_pmc_$c.csv
done
for t in sq_counters fir_ablate host_path mfma_fir mfma_fir2 mfma_fir2_ab mfma_fir2_batch4 mfma_rate firmm_probe pk_glitch mm2_glitch mm2_glitch_unpatched_build mfma_hazard repeat_check repeat_check_pipelined_run repeat_check_batch_run nco_ablate copy_rate issue_rate conv_time shift_in_place epi_cost stream_rate pipeline_time pass_breakdown pass_breakdown_zero power_split mfma_power ab6_batch4 ab6_batch8 ab7 shift_time_final downsample_time_final firmm_probe_d16 fftbig_time clock_watch; do [ -f $src/$t.txt ] && cp $src/$t.txt profiles/${tag}_$t.txt; done
ls -la profiles/${tag}_*
