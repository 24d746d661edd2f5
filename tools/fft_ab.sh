#!/bin/bash
# A/B of two-step FFT builds: tools/fft_ab.sh <tag>... ; each tag = go-sdr_amd/libhzsdr_fft_<tag>.so ("hip" = the library).
# Prints the column and row kernels' median durations (rocprofv3 kernel trace), two rounds, interleaved.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
for v in "$@"; do
  if [ $v = hip ]; then unset HZSDR_LIB; else export HZSDR_LIB=$R/go-sdr_amd/libhzsdr_fft_$v.so; fi
  rm -rf /tmp/ft_$v
  REPS=${REPS:-20} rocprofv3 --kernel-trace --output-format csv -d /tmp/ft_$v -- python3 $R/tools/prof_kernels.py ${SIZES:-fftbig14 fftbig16 fftbig18 fftbig20} > /dev/null 2>&1
  echo "== $v (round $round)"; python3 $R/tools/fft_pairs.py /tmp/ft_$v
done
done
