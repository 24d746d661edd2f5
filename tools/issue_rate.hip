// issue_rate.hip -- what one VALU instruction costs on gfx950, measured: SIMD cycles
// per wave-instruction for the ops the f64 NCO and the f32 FFT cores are made of.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/issue_rate.hip -o build/issue_rate && build/issue_rate
// Each kernel runs ITER x 8 independent chains of one op per lane, 4 waves per SIMD
// resident (enough to cover the dependent-issue latency), every SIMD of the chip busy.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITER 4096

template <int OP> __global__ __launch_bounds__(256) void k(double *out, double seed) {
    double a[8];
    float f[8];
    for (int i = 0; i < 8; i++) {
        a[i] = seed + threadIdx.x * 1e-9 + i;
        f[i] = (float)a[i];
    }
    const double c = seed * 0.999, d = seed * 1e-3;
    const float cf = (float)c, df = (float)d;
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (OP == 0) a[i] = a[i] * c;                       // v_mul_f64
            if constexpr (OP == 1) a[i] = a[i] + d;                       // v_add_f64
            if constexpr (OP == 2) a[i] = __builtin_fma(a[i], c, d);      // v_fma_f64
            if constexpr (OP == 3) f[i] = f[i] * cf;                      // v_mul_f32
            if constexpr (OP == 4) f[i] = __builtin_fmaf(f[i], cf, df);   // v_fma_f32
            if constexpr (OP == 5) a[i] = (double)(float)a[i];            // v_cvt_f32_f64 + v_cvt_f64_f32
            if constexpr (OP == 6) a[i] = (double)(int)a[i];              // v_cvt_i32_f64 + v_cvt_f64_i32
            if constexpr (OP == 7) a[i] = a[i] > d ? a[i] * c : a[i];     // v_cmp + 2 x v_cndmask + mul
            if constexpr (OP == 8) a[i] = __builtin_floor(a[i]) + d;      // v_floor_f64 + add
            if constexpr (OP == 9) a[i] = __builtin_fabs(a[i] * c);       // mul with |.| modifier
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// packed f32: two floats per lane per instruction
template <int OP> __global__ __launch_bounds__(256) void kp(float2 *out, float seed) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a[8];
    for (int i = 0; i < 8; i++) a[i] = (v2){seed + threadIdx.x * 1e-6f + i, seed - i};
    const v2 c = {seed * 0.999f, seed * 1.001f}, d = {seed * 1e-3f, seed * 2e-3f};
#pragma unroll 1
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (OP == 0) a[i] = a[i] * c;                             // v_pk_mul_f32
            if constexpr (OP == 1) a[i] = a[i] + d;                             // v_pk_add_f32
            if constexpr (OP == 2) a[i] = __builtin_elementwise_fma(a[i], c, d);  // v_pk_fma_f32
        }
    }
    v2 s = {0, 0};
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = make_float2(s.x, s.y);
}

template <class F> static double run(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    const int blocks = cus * 4;  // 4 x 256 lanes per CU = 4 waves per SIMD
    double *out;
    hipMalloc(&out, (size_t)blocks * 256 * 16);
    printf("%s: %d CUs, %.2f GHz (clockRate)\n", p.gcnArchName, cus, ghz);
    const char *names[] = {"v_mul_f64", "v_add_f64", "v_fma_f64", "v_mul_f32", "v_fma_f32", "cvt f64->f32->f64 (2 ops)",
                           "cvt f64->i32->f64 (2 ops)", "cmp + select + mul f64", "floor_f64 + add_f64", "mul_f64 |.|"};
    const char *pnames[] = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"};
    // per SIMD: 4 waves x ITER x 8 statements; cycles = ms * 1e-3 * ghz * 1e9
#define REPORT(name, ms) \
    printf("%-30s %8.3f ms  %6.2f SIMD-cycles per wave-statement\n", name, ms, (ms) * 1e-3 * ghz * 1e9 / (4.0 * ITER * 8))
#define RUN(OP) { double ms = run([&] { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001); }); REPORT(names[OP], ms); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
#define RUNP(OP) { double ms = run([&] { hipLaunchKernelGGL(kp<OP>, dim3(blocks), dim3(256), 0, 0, (float2 *)out, 1.0000001f); }); REPORT(pnames[OP], ms); }
    RUNP(0) RUNP(1) RUNP(2)
    hipFree(out);
    return 0;
}
