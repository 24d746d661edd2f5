"""Shared helpers for the parity tests: formats, seeded synthetic inputs
(splitmix64) and ULP distance."""
import numpy as np

FMT = {"c64": 1, "u8": 2, "i16": 3, "i8": 4}
DT = {"c64": np.complex64, "u8": np.uint8, "i16": np.int16, "i8": np.int8}


def samples(fmt, data):
    """Build a sample buffer of format `fmt` from [[I, Q], ...] pairs."""
    if fmt == "c64":
        a = np.asarray(data, np.float32).reshape(-1, 2)
        return np.ascontiguousarray(a).view(np.complex64).reshape(-1)
    return np.ascontiguousarray(np.asarray(data).reshape(-1, 2).astype(DT[fmt]))


def filled(fmt, n, pair):
    if fmt == "c64":
        return np.full(n, np.complex64(complex(pair[0], pair[1])), np.complex64)
    a = np.empty((n, 2), DT[fmt])
    a[:, 0], a[:, 1] = pair[0], pair[1]
    return a


def zeros(fmt, n):
    return np.zeros(n, np.complex64) if fmt == "c64" else np.zeros((n, 2), DT[fmt])


def splitmix64(seed, n):
    """n uint64 values of the splitmix64 sequence (vectorised)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def rand_u8(seed, n):
    return (splitmix64(seed, 2 * n) >> np.uint64(56)).astype(np.uint8).reshape(n, 2)


def rand_i8(seed, n):
    return rand_u8(seed, n).view(np.int8)


def rand_i16(seed, n):
    return (splitmix64(seed, 2 * n) >> np.uint64(48)).astype(np.uint16).view(np.int16).reshape(n, 2)


def rand_c64(seed, n):
    """re/im uniform in [-1, 1) as float32."""
    u = (splitmix64(seed, 2 * n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    f = (u * 2.0 - 1.0).astype(np.float32)
    return f.view(np.complex64).reshape(n)


def ulp_diff(a, b):
    """Per-component ULP distance between two float32/complex64 arrays."""
    a = np.ascontiguousarray(a).view(np.float32).ravel()
    b = np.ascontiguousarray(b).view(np.float32).ravel()
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def in_epsilon(expected, actual, eps):
    """testify assert.InEpsilon: |e - a| / |e| <= eps."""
    e = np.asarray(expected, np.float64)
    a = np.asarray(actual, np.float64)
    return bool(np.all(np.abs(e - a) <= eps * np.abs(e)))


# ---- the FIR-decimate bound (one definition for every test, smoke() and bench.py repeat it) ----
# float32 FFT overlap-save against a float64 direct form: max-abs error per output
# <= FIR_ABS * sum|h| * max|x| AND relative L2 error <= FIR_REL_L2.  Round 1 used 4e-6 for the
# max-abs term, 50-100x what the kernels achieve; these are ~3x the worst case observed.
FIR_ABS = 6e-7
FIR_REL_L2 = 3e-7
# The two mixer orders against EACH OTHER.  Each is a float32 FFT pipeline of ~45 sequential
# roundings (1.6e-7 .. 2.0e-7 relative L2 from the oracle, tests/accuracy_probe.py) whose
# rounding errors are independent of the other's, so their distance measures sqrt(2) times
# that: 2.1e-7 .. 2.5e-7 on the cases here.  A late-mixer defect (a wrong modulated-tap
# spectrum, a phase slip of 1e-6 rad) shows up as 1e-6 or more.
CROSS_REL_L2 = 3e-7


def fir_errors(got, want, taps, xmax):
    """-> (max abs error, its bound, relative L2 error) of a FIR-decimate output."""
    g = np.asarray(got).astype(np.complex128)
    w = np.asarray(want).astype(np.complex128)
    err = float(np.abs(g - w).max()) if len(w) else 0.0
    bound = FIR_ABS * float(np.abs(np.asarray(taps)).sum()) * max(float(xmax), 1e-30)
    den = float(np.linalg.norm(w))
    rel = float(np.linalg.norm(g - w)) / den if den > 0 else 0.0
    return err, bound, rel


def assert_fir_close(got, want, taps, xmax, what=""):
    err, bound, rel = fir_errors(got, want, taps, xmax)
    assert err <= bound, (what, "max abs", err, bound)
    # very short / very quiet outputs: the L2 ratio is dominated by the float32 rounding of a
    # handful of outputs, the max-abs bound is the meaningful one there
    if len(np.asarray(want)) >= 64:
        assert rel <= FIR_REL_L2, (what, "rel L2", rel, FIR_REL_L2)
    return err, bound, rel
