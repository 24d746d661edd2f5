"""ctypes declarations for libhzsdr_hip.so (include/hzsdr.h).

The product path is the HIP library: if it is missing this module raises at
import time -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HZSDR_LIB: another build of the same library (A/B measurements of two kernels in one GPU call)
LIB_PATH = os.environ.get("HZSDR_LIB") or os.path.join(_HERE, "libhzsdr_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libhzsdr_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C go-sdr_amd/csrc` (needs hipcc); there is no CPU fallback")



def _one_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64
    (same SONAME as /opt/rocm's).  If this library is loaded first it binds the
    system copy, a later `import torch` maps the bundled copy as well, and the
    second runtime to initialise finds no GPU ("No HIP GPUs are available").
    So when torch is installed but not imported yet, map ITS runtime first: the
    dynamic loader then satisfies our DT_NEEDED libamdhip64.so.7 with it, and
    torch's own import later reuses the same mapping.  Without torch (the cgo /
    C++ case) nothing happens and the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        with open("/proc/self/maps") as f:
            if any("libamdhip64" in line for line in f):
                return
    except OSError:
        pass
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)


_one_hip_runtime()
lib = C.CDLL(LIB_PATH)

FMT_C64, FMT_U8, FMT_I16, FMT_I8 = 1, 2, 3, 4
MEM_HOST, MEM_DEVICE = 0, 1
FFT_BACKWARD, FFT_FORWARD = 0, 1
FIR_PATH_NONE, FIR_PATH_TRANSFORM, FIR_PATH_MATRIX = 0, 1, 2
FIR_KERNEL_NONE, FIR_KERNEL_TRANSFORM, FIR_KERNEL_MATRIX_CHUNKS, FIR_KERNEL_MATRIX_PASSES = 0, 1, 2, 3
FIR_IMPL_AUTO, FIR_IMPL_TRANSFORMS, FIR_IMPL_MATRIX_CHUNKS = 0, 1, 2  # hzsdr_chain_fir_options
CONV_CONVOLVE, CONV_CROSS_CORRELATE = 0, 1

(OK, ERR_FORMAT_MISMATCH, ERR_FORMAT_UNKNOWN, ERR_DST_TOO_SMALL, ERR_CONVERSION_NOT_IMPLEMENTED,
 ERR_LENGTH_MISMATCH, ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY) = range(10)


class NcoSegment(C.Structure):
    _fields_ = [("first", C.c_uint64), ("count", C.c_uint64), ("t0", C.c_double),
                ("step", C.c_double)]


vp, sz, i32, u32, i64, u64, f32, f64 = (C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_int64,
                                        C.c_uint64, C.c_float, C.c_double)
psz = C.POINTER(C.c_size_t)
pvp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol include/hzsdr.h declares
SIGNATURES = {
    "hzsdr_backend": (C.c_char_p, []),
    "hzsdr_version": (C.c_char_p, []),
    "hzsdr_strerror": (C.c_char_p, [i32]),
    "hzsdr_format_size": (i32, [i32]),
    "hzsdr_device_count": (i32, [C.POINTER(i32)]),
    "hzsdr_open": (i32, [i32, i32, pvp]),
    "hzsdr_close": (i32, [vp]),
    "hzsdr_last_error": (C.c_char_p, [vp]),
    "hzsdr_memspace": (i32, [vp]),
    "hzsdr_set_stream": (i32, [vp, vp]),
    "hzsdr_use_own_stream": (i32, [vp]),
    "hzsdr_get_stream": (vp, [vp]),
    "hzsdr_synchronize": (i32, [vp]),
    "hzsdr_call_count": (i32, [vp, C.POINTER(C.c_ulonglong)]),
    "hzsdr_malloc_device": (i32, [vp, sz, pvp]),
    "hzsdr_free_device": (i32, [vp, vp]),
    "hzsdr_malloc_pinned": (i32, [vp, sz, pvp]),
    "hzsdr_free_pinned": (i32, [vp, vp]),
    "hzsdr_memcpy_h2d": (i32, [vp, vp, vp, sz]),
    "hzsdr_memcpy_d2h": (i32, [vp, vp, vp, sz]),
    "hzsdr_convert": (i32, [vp, i32, vp, sz, i32, vp, sz, psz]),
    "hzsdr_i16_shift_lsb_to_msb": (i32, [vp, vp, sz, i32]),
    "hzsdr_lut_create": (i32, [vp, i32, i32, vp, sz, pvp]),
    "hzsdr_lut_lookup": (i32, [vp, i32, vp, sz, i32, vp, sz, psz]),
    "hzsdr_lut_free": (i32, [vp]),
    "hzsdr_lut_identity": (i32, [vp]),
    "hzsdr_scale": (i32, [vp, vp, sz, f32]),
    "hzsdr_rotate": (i32, [vp, vp, sz, f32, f32]),
    "hzsdr_add": (i32, [vp, vp, sz, vp, sz, vp, sz]),
    "hzsdr_sum": (i32, [vp, i32, vp, pvp, i32, sz]),
    "hzsdr_rotlut_create": (i32, [vp, i32, f32, f32, pvp]),
    "hzsdr_rotlut_set_multiplier": (i32, [vp, f32, f32]),
    "hzsdr_rotlut_apply": (i32, [vp, vp, sz]),
    "hzsdr_rotlut_free": (i32, [vp]),
    "hzsdr_nco_create": (i32, [vp, u64, pvp]),
    "hzsdr_nco_shift": (i32, [vp, f64, vp, sz]),
    "hzsdr_nco_get_time": (i32, [vp, C.POINTER(f64)]),
    "hzsdr_nco_set_time": (i32, [vp, f64]),
    "hzsdr_nco_set_ulp1": (i32, [vp, i32]),
    "hzsdr_nco_free": (i32, [vp]),
    "hzsdr_nco_segments": (i32, [u64, f64, u64, C.POINTER(NcoSegment), sz, psz, C.POINTER(f64)]),
    "hzsdr_decimate": (i32, [vp, i32, vp, sz, i32, vp, sz, u32, i64, psz]),
    "hzsdr_downsample": (i32, [vp, i32, vp, sz, i32, vp, sz, u32, i64, psz]),
    "hzsdr_fft_plan": (i32, [vp, vp, sz, vp, sz, i32, pvp]),
    "hzsdr_fft_plan_batch": (i32, [vp, vp, vp, sz, sz, i32, pvp]),
    "hzsdr_fft_transform": (i32, [vp]),
    "hzsdr_fft_free": (i32, [vp]),
    "hzsdr_convolve_create": (i32, [vp, vp, sz, vp, sz, vp, sz, i32, pvp]),
    "hzsdr_convolve_freq_create": (i32, [vp, vp, sz, vp, sz, vp, sz, pvp]),
    "hzsdr_conv_exec": (i32, [vp]),
    "hzsdr_conv_set_filter": (i32, [vp, vp, sz]),
    "hzsdr_conv_free": (i32, [vp]),
    "hzsdr_convolution_blocks": (i32, [vp, vp, sz, vp, sz, vp, sz, psz]),
    "hzsdr_beamform_angles_2d": (i32, [f64, f64, C.POINTER(f64), C.POINTER(f64), i32,
                                       C.POINTER(f32)]),
    "hzsdr_beamform_angles": (i32, [f64, f64, C.POINTER(f64), i32, C.POINTER(f32)]),
    "hzsdr_beamform": (i32, [vp, vp, i32, pvp, C.POINTER(f32), i32, sz]),
    "hzsdr_beamform_partial": (i32, [vp, vp, i32, pvp, C.POINTER(f32), i32, sz, i32]),
    "hzsdr_peak_lag": (i32, [vp, vp, sz, C.POINTER(i64)]),
    "hzsdr_mean_phase": (i32, [vp, vp, vp, sz, C.POINTER(f64)]),
    "hzsdr_fftshift_scale": (i32, [vp, vp, sz, f32]),
    "hzsdr_graft": (i32, [vp, vp, sz, pvp, i32, sz]),
    "hzsdr_byteswap": (i32, [vp, i32, vp, sz]),
    "hzsdr_convert_foreign": (i32, [vp, i32, vp, sz, i32, i32, vp, sz, i32, psz]),
    "hzsdr_chain_create": (i32, [vp, i32, u64, pvp]),
    "hzsdr_chain_shift": (i32, [vp, f64]),
    "hzsdr_chain_gain": (i32, [vp, f32]),
    "hzsdr_chain_rotate": (i32, [vp, f32, f32]),
    "hzsdr_chain_decimate": (i32, [vp, u32]),
    "hzsdr_chain_downsample": (i32, [vp, u32]),
    "hzsdr_chain_convolution": (i32, [vp, vp, sz, u32]),
    "hzsdr_chain_fir_decimate": (i32, [vp, C.POINTER(f32), sz, u32]),
    "hzsdr_chain_mix_in_order": (i32, [vp, i32]),
    "hzsdr_chain_shift_ulp1": (i32, [vp, i32]),
    "hzsdr_chain_fir_options": (i32, [vp, i32, u32, i32]),
    "hzsdr_chain_pipeline": (i32, [vp, i32]),
    "hzsdr_chain_plan": (i32, [vp, sz, psz, psz]),
    "hzsdr_chain_run": (i32, [vp, vp, sz, vp, sz, psz, psz]),
    "hzsdr_chain_run_after": (i32, [vp, vp, sz, vp, sz, psz, psz, vp]),
    "hzsdr_chain_run_batch": (i32, [vp, vp, vp, sz, sz, sz, psz, psz]),
    "hzsdr_chain_run_batch_after": (i32, [vp, vp, vp, sz, sz, sz, psz, psz, vp]),
    "hzsdr_mgpu_open": (i32, [C.POINTER(C.c_int), i32, pvp]),
    "hzsdr_mgpu_close": (i32, [vp]),
    "hzsdr_mgpu_shards": (i32, [vp]),
    "hzsdr_mgpu_ctx": (i32, [vp, i32, pvp]),
    "hzsdr_mgpu_shard_channels": (i32, [i32, i32, i32, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hzsdr_mgpu_beamform": (i32, [vp, vp, i32, i32, C.POINTER(C.c_void_p), C.POINTER(C.c_float), i32, sz, i32]),
    "hzsdr_mgpu_synchronize": (i32, [vp]),
    "hzsdr_mgpu_last_error": (C.c_char_p, [vp]),
    "hzsdr_mgpu_peer_pairs": (i32, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hzsdr_chain_reset": (i32, [vp]),
    "hzsdr_chain_set_time": (i32, [vp, f64]),
    "hzsdr_chain_time": (i32, [vp, C.POINTER(f64)]),
    "hzsdr_chain_last_fir_path": (i32, [vp, C.POINTER(i32)]),
    "hzsdr_chain_last_fir_kernel": (i32, [vp, C.POINTER(i32)]),
    "hzsdr_chain_free": (i32, [vp]),
    "hzsdr_ring_create": (i32, [vp, sz, i32, pvp]),
    "hzsdr_ring_iq_buffer": (i32, [vp, pvp, psz, psz]),
    "hzsdr_ring_acquire": (i32, [vp, C.POINTER(i32), pvp]),
    "hzsdr_ring_submit": (i32, [vp, i32, sz]),
    "hzsdr_ring_submit_many": (i32, [vp, i32, i32, sz]),
    "hzsdr_ring_release": (i32, [vp, i32]),
    "hzsdr_ring_pop": (i32, [vp, pvp, psz]),
    "hzsdr_ring_in_flight": (i32, [vp]),
    "hzsdr_ring_free": (i32, [vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library disagree
    _fn.restype = _res
    _fn.argtypes = _args
