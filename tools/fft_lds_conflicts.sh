#!/bin/bash
# LDS bank conflicts of the two-step FFT's kernels (N = 2^16): the library against a build with -DHZ_FFT2_LDS_SKEW=0
# (go-sdr_amd/libhzsdr_fft_noskew.so).  SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS cycles.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in hip ${VARIANTS:-noskew}; do
  if [ $v = hip ]; then unset HZSDR_LIB; else export HZSDR_LIB=$R/go-sdr_amd/libhzsdr_fft_$v.so; fi
  rm -rf /tmp/fl_$v
  REPS=6 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/fl_$v -- python3 $R/tools/prof_kernels.py fftbig16 > /dev/null 2>&1
  echo "== $v"; python3 $R/tools/pmc_sq.py /tmp/fl_$v | grep -A4 "fft2_"
done
