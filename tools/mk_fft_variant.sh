#!/bin/bash
# An A/B build of the FFT unit alone: tools/mk_fft_variant.sh <tag> [compiler flags...]
#   -> go-sdr_amd/libhzsdr_fft_<tag>.so = the library's objects (build/csrc, `make -C go-sdr_amd/csrc` first) with
#      hz_fft.hip recompiled under the flags (-DHZ_FFT2_ABL=…, -DHZ_FFT2_TW=0, -DHZ_FFT2_XCD=0, -DHZ_FFT2_LDS_SKEW=0,
#      -DHZ_FFT2_SKEW_UNIT=64, …); tools/fft_ab.sh <tag>… and HZSDR_LIB=… pick it up.  Git-ignored like every .so.
set -e
cd "$(dirname "$0")/../go-sdr_amd/csrc"
tag=$1; shift
mkdir -p ../../build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c hz_fft.hip -o ../../build/ab/hz_fft_$tag.o
objs=$(ls ../../build/csrc/*.o | grep -v "/hz_fft.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhzsdr_fft_$tag.so $objs ../../build/ab/hz_fft_$tag.o -ldl
python3 ../../tools/fix_pk_opsel.py ../libhzsdr_fft_$tag.so | tail -1
