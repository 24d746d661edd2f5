// pk_glitch.hip -- a packed-float32 instruction that now and then loses one half of its result in lanes 48-63
// while the other wave of its SIMD issues MFMAs: which instructions, which half, lost how (zero / not written),
// and what the partner has to be doing.
//
// The trail: tools/mm2_glitch.hip (the non-repeatable outputs of hz::mm2::fir_mm2_kernel are the mixer's first
// step factor, lanes 48-63, the term of a v_pk_mul_f32 missing), tools/pk_hazard.hip and tools/lds_jit.hip (not a
// wait-state hazard of the MFMA, not late LDS data: the same happens behind a global load, behind a read that
// landed long ago, with one wait state in between).  Here the instruction under test runs on registers that
// are written by plain moves well ahead; its destination pair holds a poison.
//   test instruction T:  0 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]   1 v_pk_mul_f32   2 v_pk_fma_f32   3 v_pk_add_f32
//                        4 v_mul_f32 + v_mul_f32 (not packed)         5 v_fma_f64        6 v_pk_mul_f32 op_sel, neg_lo
//                        7 .. 13: other op_sel patterns and instructions (see the table in main)
//   partner P:           5 sleep + ONE MFMA   6 sleep + 64 float32 vector instructions (no MFMA)   7 sleep + 4 MFMAs of 16x16x64
//                        0 idle   1 MFMAs back to back   2 bursts: sleep, 16 MFMAs   3 bursts of 4 MFMAs
//                        4 bursts with ten ds_write_b128 and forty v_xor in front (the kernel's landing)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Res {
    unsigned long long bad_lo, bad_hi, quarter[4];
    unsigned long long lo_zero, lo_poison, lo_other, hi_zero, hi_poison, hi_other;
    unsigned sample_got, sample_want;
};

template <int T>
__global__ __launch_bounds__(512) void glitch_kernel(Res *res, int iters, int partner) {
    extern __shared__ int lds[];
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    for (int i = tid; i < 8192; i += 512) lds[8192 + i] = i * 2654435761u;
    if (tid == 0) lds[0] = 0;
    __syncthreads();
    if (wave >= 4) {
        v16i c0{}, c1{}, c2{}, c3{};
        v4i *base = reinterpret_cast<v4i *>(lds + 8192) + l;
        v4i a0 = base[0], a1 = base[64], b0 = base[128], b1 = base[192];
        unsigned rng = 12345u + (unsigned)tid * 7919u + blockIdx.x * 104729u;
        auto done = [&]() { return __hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 4; };
        if (partner == 1) {
            while (!done()) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c3, 0, 0, 0);
                }
            }
        } else if (partner >= 2) {
            while (!done()) {
                rng = rng * 1664525u + 1013904223u;
                const int nap = (int)__builtin_amdgcn_readfirstlane((int)(rng >> 27));  // 0 .. 31 x 64 clocks
                for (int q = 0; q < nap; q++) __builtin_amdgcn_s_sleep(1);
                if (partner == 4) {
                    v4i t[10];
#pragma unroll
                    for (int u = 0; u < 10; u++) t[u] = v4i{a0[0] ^ (int)0x80808080, a0[1] ^ (int)0x80808080, a1[2] ^ (int)0x80808080, b1[3] ^ (int)0x80808080} + u;
#pragma unroll
                    for (int u = 0; u < 10; u++) base[64 * (40 + u)] = t[u];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
                if (partner == 5) {
                    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c0, 0, 0, 0);
                    continue;
                }
                if (partner == 6) {
                    float xx = (float)rng;
#pragma unroll
                    for (int u = 0; u < 64; u++) xx = __builtin_fmaf(xx, 1.0001f, 0.5f);
                    c0[1] += (int)xx;
                    continue;
                }
                if (partner == 7) {
                    v4i d0{c0[0], c0[1], c0[2], c0[3]};
#pragma unroll
                    for (int u = 0; u < 4; u++) d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, d0, 0, 0, 0);
                    c0[0] = d0[0], c0[1] = d0[1], c0[2] = d0[2], c0[3] = d0[3];
                    continue;
                }
                const int burst = partner == 3 ? 1 : 4;
                for (int u = 0; u < burst; u++) {
                    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c3, 0, 0, 0);
                }
            }
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] == 0x12345678) res[0].bad_lo = 1;
        return;
    }
    unsigned long long bad_lo = 0, bad_hi = 0, z[6] = {0, 0, 0, 0, 0, 0};
    unsigned s_got = 0, s_want = 0;
    const float s0 = 0.25f + (float)l * 0.001f, s1 = 0.75f - (float)l * 0.002f;
    for (int it = 0; it < iters; it++) {
        const float poison = -7777.0f - (float)(it & 255);
        const float f0 = 1.0f + (float)((it * 7) & 1023) * (1.0f / 8192.0f), f1 = 1.5f + (float)((it * 13) & 1023) * (1.0f / 8192.0f);
        float lo, hi, wlo, whi;
#define PRE "v_mov_b32 v14, %[f0]\n v_mov_b32 v15, %[f1]\n v_mov_b32 v30, %[s0]\n v_mov_b32 v31, %[s1]\n v_mov_b32 v26, %[poi]\n v_mov_b32 v27, %[poi]\n s_nop 7\n"
#define POST "s_nop 3\n v_mov_b32 %[lo], v26\n v_mov_b32 %[hi], v27\n"
#define OPSX : [lo] "=&v"(lo), [hi] "=&v"(hi) : [poi] "v"(poison), [f0] "v"(f0), [f1] "v"(f1), [s0] "v"(s0), [s1] "v"(s1) : "v14", "v15", "v26", "v27", "v30", "v31"
        if constexpr (T == 0) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n" POST OPSX);
            wlo = s0 * f1, whi = s1 * f0;
        } else if constexpr (T == 1) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15]\n" POST OPSX);
            wlo = s0 * f0, whi = s1 * f1;
        } else if constexpr (T == 2) {
            asm volatile(PRE "v_pk_fma_f32 v[26:27], v[30:31], v[14:15], v[30:31]\n" POST OPSX);
            wlo = __fmaf_rn(s0, f0, s0), whi = __fmaf_rn(s1, f1, s1);
        } else if constexpr (T == 3) {
            asm volatile(PRE "v_pk_add_f32 v[26:27], v[30:31], v[14:15]\n" POST OPSX);
            wlo = s0 + f0, whi = s1 + f1;
        } else if constexpr (T == 4) {
            asm volatile(PRE "v_mul_f32 v26, v30, v14\n v_mul_f32 v27, v31, v15\n" POST OPSX);
            wlo = s0 * f0, whi = s1 * f1;
        } else if constexpr (T == 5) {
            asm volatile(PRE "v_fma_f64 v[26:27], v[30:31], v[14:15], v[30:31]\n" POST OPSX);
            const double a = __hiloint2double((int)__float_as_uint(s1), (int)__float_as_uint(s0)),
                         b = __hiloint2double((int)__float_as_uint(f1), (int)__float_as_uint(f0));
            const double r = __fma_rn(a, b, a);
            wlo = __uint_as_float((unsigned)__double2loint(r)), whi = __uint_as_float((unsigned)__double2hiint(r));
        } else if constexpr (T == 6) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n" POST OPSX);
            wlo = s0 * -f1, whi = s1 * f0;
        } else if constexpr (T == 7) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[1,0] op_sel_hi:[0,1]\n" POST OPSX);
            wlo = s1 * f0, whi = s0 * f1;
        } else if constexpr (T == 8) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[1,1] op_sel_hi:[0,0]\n" POST OPSX);
            wlo = s1 * f1, whi = s0 * f0;
        } else if constexpr (T == 9) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[0,0] op_sel_hi:[0,0]\n" POST OPSX);
            wlo = s0 * f0, whi = s0 * f0;
        } else if constexpr (T == 10) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[1,1] op_sel_hi:[1,1]\n" POST OPSX);
            wlo = s1 * f1, whi = s1 * f1;
        } else if constexpr (T == 11) {
            asm volatile(PRE "v_pk_fma_f32 v[26:27], v[30:31], v[14:15], v[30:31] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n" POST OPSX);
            wlo = __fmaf_rn(s0, f1, s0), whi = __fmaf_rn(s1, f0, s1);
        } else if constexpr (T == 12) {
            asm volatile(PRE "v_pk_add_f32 v[26:27], v[30:31], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n" POST OPSX);
            wlo = s0 + f1, whi = s1 + f0;
        } else if constexpr (T == 13) {
            asm volatile(PRE "v_pk_fma_f32 v[26:27], v[30:31], v[14:15], v[30:31] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n" POST OPSX);
            wlo = __fmaf_rn(s0, f0, s1), whi = __fmaf_rn(s1, f1, s0);
        } else if constexpr (T == 14) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel_hi:[1,0]\n" POST OPSX);
            wlo = s0 * f0, whi = s1 * f0;
        } else if constexpr (T == 15) {
            asm volatile(PRE "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[0,1]\n" POST OPSX);
            wlo = s0 * f1, whi = s1 * f1;
        } else if constexpr (T == 16) {  // the packed MOVE: lo <- src0 half op_sel[0], hi <- src1 half op_sel[1]
            asm volatile(PRE "v_pk_mov_b32 v[26:27], v[30:31], v[14:15] op_sel:[0,1]\n" POST OPSX);
            wlo = s0, whi = f1;
        } else if constexpr (T == 17) {  // packed float16 (VOP3P too): lo result from the HIGH half of src1's register
            asm volatile(PRE "v_pk_add_f16 v26, v30, v14 op_sel:[0,1] op_sel_hi:[1,0]\n v_mov_b32 v27, v26\n" POST OPSX);
            const unsigned a = __float_as_uint(s0), b = __float_as_uint(f0);
            const _Float16 alo = __builtin_bit_cast(_Float16, (unsigned short)(a & 0xffff)), ahi = __builtin_bit_cast(_Float16, (unsigned short)(a >> 16));
            const _Float16 blo = __builtin_bit_cast(_Float16, (unsigned short)(b & 0xffff)), bhi = __builtin_bit_cast(_Float16, (unsigned short)(b >> 16));
            const _Float16 rlo = alo + bhi, rhi = ahi + blo;
            const unsigned r = (unsigned)__builtin_bit_cast(unsigned short, rlo) | ((unsigned)__builtin_bit_cast(unsigned short, rhi) << 16);
            wlo = whi = __uint_as_float(r);
        } else {  // a 64-bit shift: a 64-bit VALU operand pair that is not a packed instruction
            asm volatile(PRE "v_lshlrev_b64 v[26:27], 1, v[14:15]\n" POST OPSX);
            const unsigned long long xx = ((unsigned long long)__float_as_uint(f1) << 32) | __float_as_uint(f0), yy = xx << 1;
            wlo = __uint_as_float((unsigned)yy), whi = __uint_as_float((unsigned)(yy >> 32));
        }
        const unsigned ulo = __float_as_uint(lo), uhi = __float_as_uint(hi), up = __float_as_uint(poison);
        if (ulo != __float_as_uint(wlo)) {
            bad_lo++;
            z[ulo == 0 || ulo == 0x80000000u ? 0 : (ulo == up ? 1 : 2)]++;
            s_got = ulo, s_want = __float_as_uint(wlo);
        }
        if (uhi != __float_as_uint(whi)) {
            bad_hi++;
            z[3 + (uhi == 0 || uhi == 0x80000000u ? 0 : (uhi == up ? 1 : 2))]++;
            s_got = uhi, s_want = __float_as_uint(whi);
        }
    }
    if (bad_lo | bad_hi) {
        atomicAdd(&res->bad_lo, bad_lo);
        atomicAdd(&res->bad_hi, bad_hi);
        atomicAdd(&res->quarter[l >> 4], bad_lo + bad_hi);
        atomicAdd(&res->lo_zero, z[0]);
        atomicAdd(&res->lo_poison, z[1]);
        atomicAdd(&res->lo_other, z[2]);
        atomicAdd(&res->hi_zero, z[3]);
        atomicAdd(&res->hi_poison, z[4]);
        atomicAdd(&res->hi_other, z[5]);
        res->sample_got = s_got;
        res->sample_want = s_want;
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (l == 0) atomicAdd(&lds[0], 1);
}

static Res *g_res;

template <int T> static void run_one(int partner, int iters) {
    static const char *tn[19] = {"v_pk_mul_f32 op_sel", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "2 x v_mul_f32", "v_fma_f64", "v_pk_mul_f32 op_sel neg_lo",
                                 "pk_mul op_sel:[1,0] hi:[0,1]", "pk_mul op_sel:[1,1] hi:[0,0]", "pk_mul op_sel:[0,0] hi:[0,0]", "pk_mul op_sel:[1,1] hi:[1,1]",
                                 "pk_fma op_sel:[0,1,0] hi:[1,0,1]", "pk_add op_sel:[0,1] hi:[1,0]", "pk_fma op_sel:[0,0,1] hi:[1,1,0]", "pk_mul op_sel_hi:[1,0]", "pk_mul op_sel:[0,1]",
                                 "v_pk_mov_b32 op_sel:[0,1]", "v_pk_add_f16 op_sel:[0,1]", "v_lshlrev_b64"};
    static const char *pn[8] = {"idle", "MFMAs back to back", "sleep + 16 MFMAs", "sleep + 4 MFMAs", "sleep + landing + 16 MFMAs", "sleep + 1 MFMA", "sleep + 64 v_fma_f32", "sleep + 4 MFMA 16x16x64"};
    CK(hipMemset(g_res, 0, sizeof(Res)));
    auto k = glitch_kernel<T>;
    const int lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, g_res, iters, partner);
    CK(hipDeviceSynchronize());
    Res r;
    CK(hipMemcpy(&r, g_res, sizeof r, hipMemcpyDeviceToHost));
    printf("  %-27s partner %-27s: wrong lo %5llu hi %5llu of %.3g  quarters %llu %llu %llu %llu | lo: zero %llu untouched %llu other %llu | hi: zero %llu untouched %llu other %llu",
           tn[T], pn[partner], r.bad_lo, r.bad_hi, 256.0 * 4 * 64 * iters, r.quarter[0], r.quarter[1], r.quarter[2], r.quarter[3], r.lo_zero, r.lo_poison,
           r.lo_other, r.hi_zero, r.hi_poison, r.hi_other);
    if (r.bad_lo | r.bad_hi) {
        float a, b;
        memcpy(&a, &r.sample_got, 4);
        memcpy(&b, &r.sample_want, 4);
        printf("  (e.g. got %g want %g)", a, b);
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    CK(hipMalloc(&g_res, sizeof(Res)));
    printf("%d iterations x 256 CUs x 4 waves x 64 lanes\n", iters);
    for (int p = 0; p < 5; p++) run_one<0>(p, iters);
    for (int p = 1; p < 5; p++) run_one<1>(p, iters);
    for (int p = 1; p < 5; p++) run_one<2>(p, iters);
    for (int p = 1; p < 5; p++) run_one<3>(p, iters);
    for (int p = 1; p < 5; p++) run_one<4>(p, iters);
    for (int p = 1; p < 5; p++) run_one<5>(p, iters);
    for (int p = 1; p < 5; p++) run_one<6>(p, iters);
    // what the partner has to do (T = 0)
    for (int p = 5; p < 8; p++) run_one<0>(p, iters);
    // which operand selections lose a half (partner: sleep + 4 MFMAs)
    run_one<7>(3, iters);
    run_one<8>(3, iters);
    run_one<9>(3, iters);
    run_one<10>(3, iters);
    run_one<11>(3, iters);
    run_one<12>(3, iters);
    run_one<13>(3, iters);
    run_one<14>(3, iters);
    run_one<15>(3, iters);
    // neighbours of the three instructions: the packed move, packed float16, a 64-bit operand that is not packed
    run_one<16>(3, iters);
    run_one<17>(3, iters);
    run_one<18>(3, iters);
    return 0;
}
