// A one-wave kernel that samples the shader clock while OTHER kernels run (diagnostic, not product).
//
// s_memtime counts shader-clock cycles, s_memrealtime a constant 100 MHz: one lane spins on the second and writes
// both every `period` ticks, so the difference quotient of consecutive samples is the clock the chip's power
// management held over that microsecond.  Launched on a stream of its own; it occupies one wave slot of one CU and
// ends by itself after n samples (never waits for anybody: it cannot hang a launch that follows it).
// Built as a shared library (tools/Makefile -> tools/bin/libclock_witness.so); tools/clock_watch.py drives it.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_witness_kernel(uint64_t *out, int n, uint32_t period) {
    if (threadIdx.x != 0) return;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i++) {
        const uint64_t target = t0 + (uint64_t)i * period;
        while (__builtin_amdgcn_s_memrealtime() < target) __builtin_amdgcn_s_sleep(2);
        const uint64_t c = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
        out[2 * i] = c, out[2 * i + 1] = r;
    }
}

__global__ void clock_mark_kernel(uint64_t *slot) {
    if (threadIdx.x == 0) *slot = __builtin_amdgcn_s_memrealtime();
}

// the 100 MHz counter's value when the stream reaches this point: brackets the work on ITS stream
extern "C" int clock_witness_mark(void *stream, uint64_t *slot_dev) {
    hipLaunchKernelGGL(clock_mark_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot_dev);
    return (int)hipGetLastError();
}

extern "C" int clock_witness_launch(void *stream, uint64_t *out_dev, int n, unsigned period_ticks) {
    hipLaunchKernelGGL(clock_witness_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev, n, period_ticks);
    return (int)hipGetLastError();
}
