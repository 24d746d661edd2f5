// lds_jit.hip -- a register that a ds_read has "just" landed, read by the instruction right behind the s_waitcnt:
// is its last quarter-wave there?  (It is not, now and then, while the SIMD's other wave issues MFMAs.)
//
// tools/mm2_glitch.hip traced the non-repeatable outputs of hz::mm2::fir_mm2_kernel to the mixer's first step
// factor and tools/pk_hazard.hip reproduced it with the compiler's instruction sequence: ds_read_b128, s_waitcnt
// lgkmcnt, and a vector instruction that reads the landed registers AT ONCE sees the OLD contents in lanes
// 48-63 -- a few times per 10^9, and only while the second wave of the SIMD keeps the matrix pipe busy.  This
// program sweeps the conditions: the read's width, one address for all lanes or one per lane, the consumer
// (packed multiply with op_sel, plain move, add), the number of wait states between the s_waitcnt and the
// consumer, and what the partner wave does (MFMAs + LDS reads, MFMAs alone, LDS reads alone, nothing).
// Destination registers hold a poison before every read, the LDS a known pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Res {
    unsigned long long bad, quarter[4];
    unsigned sample_got[4], sample_want[4];
};

// READ: 0 = ds_read_b128, lgkmcnt(0); 1 = four ds_read_b128, lgkmcnt(3); 2 = ds_read_b64; 3 = ds_read_b32
// CONS: 0 = v_pk_mul_f32 with op_sel on registers 0, 1 (b128 / b64); 1 = v_mov from the LAST register of the read;
//       2 = v_add_f32 of the first and the last register; 3 = v_pk_mul_f32 WITHOUT op_sel; 4 = v_add_f64 of the pair
//       with itself (a 64-bit operand that is not packed float32); 5 = v_pk_add_f32
// READ 4: ds_read_b128 that has landed LONG before the s_waitcnt (64 wait states in between): the wait does not stall;
// READ 5: global_load_dwordx2 + s_waitcnt vmcnt(0)
// GAP:  wait states between the s_waitcnt and the consumer
template <int READ, int CONS, int GAP, int BCAST>
__global__ __launch_bounds__(512) void jit_kernel(Res *res, int iters, int hammer, const float *gpat) {
    extern __shared__ int lds[];
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    float *fl = reinterpret_cast<float *>(lds);
    // LDS: [0] the done counter; floats 64 .. 64 + 4096: the pattern f[i] = 1 + i / 8192; from 32 KB: the partner's operands
    for (int i = tid; i < 4096; i += 512) fl[64 + i] = 1.0f + (float)i * (1.0f / 8192.0f);
    for (int i = tid; i < 8192; i += 512) lds[8192 + i] = i * 2654435761u;
    if (tid == 0) lds[0] = 0;
    __syncthreads();
    if (wave >= 4) {
        v16i c0{}, c1{}, c2{}, c3{};
        const v4i *base = reinterpret_cast<const v4i *>(lds + 8192) + l;
        v4i a0 = base[0], a1 = base[64], b0 = base[128], b1 = base[192];
        if (hammer == 1) {  // the product's matrix loop in miniature: four ds_read_b128 and four MFMAs per step
            while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    a0 = base[64 * ((4 * u) & 31)], a1 = base[64 * ((4 * u + 1) & 31)], b0 = base[64 * ((4 * u + 2) & 31)], b1 = base[64 * ((4 * u + 3) & 31)];
                    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c3, 0, 0, 0);
                }
            }
        } else if (hammer == 2) {  // MFMAs alone
            while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c3, 0, 0, 0);
                }
            }
        } else if (hammer == 3) {  // LDS reads alone
            v4i s{};
            while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
                for (int u = 0; u < 32; u++) s += base[64 * u];
            }
            c0[0] = s[0] + s[1] + s[2] + s[3];
        } else if (hammer == 4) {  // float64 vector instructions
            double xx = (double)l, yy = 1.0000001, zz = 0.5;
            while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
                for (int u = 0; u < 32; u++) {
                    xx = __builtin_fma(xx, yy, zz);
                    zz = __builtin_fma(zz, yy, xx);
                }
            }
            c0[0] = (int)(xx + zz);
        }
        if (c0[0] + c1[1] + c2[2] + c3[3] == 0x12345678) res[0].bad = 1;
        return;
    }
    unsigned long long bad = 0;
    unsigned s_got = 0, s_want = 0;
    const float s0 = 0.25f + (float)l * 0.001f;
    for (int it = 0; it < iters; it++) {
        const float poison = -7777.0f - (float)(it & 255);
        const int idx = BCAST ? ((it * 7 + wave * 64) & 1023) : ((it * 7 + l) & 1023);  // in 16-byte entries
        const int addr = 4 * (64 + 4 * idx);
        float got;
        const float *gaddr = gpat + 4 * idx;
#define READ_STR_0 "ds_read_b128 v[14:17], %[a]\n s_waitcnt lgkmcnt(0)\n"
#define READ_STR_1 "ds_read_b128 v[14:17], %[a]\n ds_read_b128 v[18:21], %[a] offset:16\n ds_read_b128 v[22:25], %[a] offset:32\n ds_read_b128 v[44:47], %[a] offset:48\n s_waitcnt lgkmcnt(3)\n"
#define READ_STR_2 "ds_read_b64 v[14:15], %[a]\n s_waitcnt lgkmcnt(0)\n"
#define READ_STR_3 "ds_read_b32 v14, %[a]\n s_waitcnt lgkmcnt(0)\n"
#define CONS_STR_0 "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 1\n v_add_f32 %[got], v26, v27\n"
#define READ_STR_4 "ds_read_b128 v[14:17], %[a]\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_waitcnt lgkmcnt(0)\n"
#define READ_STR_5 "global_load_dwordx2 v[14:15], %[ga], off\n s_waitcnt vmcnt(0)\n"
#define CONS_STR_3 "v_pk_mul_f32 v[26:27], v[30:31], v[14:15]\n s_nop 1\n v_add_f32 %[got], v26, v27\n"
#define CONS_STR_4 "v_add_f64 v[26:27], v[14:15], v[14:15]\n s_nop 1\n v_xor_b32 %[got], v26, v27\n"
#define CONS_STR_5 "v_pk_add_f32 v[26:27], v[30:31], v[14:15]\n s_nop 1\n v_add_f32 %[got], v26, v27\n"
#define CONS_B128_1 "v_mov_b32 %[got], v17\n"
#define CONS_B64_1 "v_mov_b32 %[got], v15\n"
#define CONS_B32_1 "v_mov_b32 %[got], v14\n"
#define CONS_B128_2 "v_add_f32 %[got], v14, v17\n"
#define CONS_B64_2 "v_add_f32 %[got], v14, v15\n"
#define SEQ(RD, GP, CN)                                                                                                                       \
    asm volatile("v_mov_b32 v14, %[poi]\n v_mov_b32 v15, %[poi]\n v_mov_b32 v16, %[poi]\n v_mov_b32 v17, %[poi]\n v_mov_b32 v30, %[s0]\n"          \
                 "v_mov_b32 v31, %[s0]\n s_nop 3\n" RD GP CN "s_nop 3\n s_waitcnt lgkmcnt(0)\n"                                                  \
                 : [got] "=&v"(got)                                                                                                            \
                 : [poi] "v"(poison), [a] "v"(addr), [s0] "v"(s0), [ga] "v"(gaddr)                                                             \
                 : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v30", "v31", "v44", "v45", \
                   "v46", "v47", "memory")
#define SEQ_GAP(RD, CN)                                  \
    if constexpr (GAP == 0) SEQ(RD, "", CN);             \
    else if constexpr (GAP == 1) SEQ(RD, "s_nop 0\n", CN); \
    else if constexpr (GAP == 2) SEQ(RD, "s_nop 1\n", CN); \
    else if constexpr (GAP == 4) SEQ(RD, "s_nop 3\n", CN); \
    else SEQ(RD, "s_nop 7\n", CN)
        const float f0 = 1.0f + (float)(4 * idx) * (1.0f / 8192.0f), f1 = 1.0f + (float)(4 * idx + 1) * (1.0f / 8192.0f),
                    f3 = 1.0f + (float)(4 * idx + 3) * (1.0f / 8192.0f);
        float want;
        if constexpr (READ == 0 || READ == 1) {
            if constexpr (CONS == 0) {
                if constexpr (READ == 0) { SEQ_GAP(READ_STR_0, CONS_STR_0); } else { SEQ_GAP(READ_STR_1, CONS_STR_0); }
                want = s0 * f1 + s0 * f0;
            } else if constexpr (CONS == 1) {
                if constexpr (READ == 0) { SEQ_GAP(READ_STR_0, CONS_B128_1); } else { SEQ_GAP(READ_STR_1, CONS_B128_1); }
                want = f3;
            } else {
                if constexpr (READ == 0) { SEQ_GAP(READ_STR_0, CONS_B128_2); } else { SEQ_GAP(READ_STR_1, CONS_B128_2); }
                want = f0 + f3;
            }
        } else if constexpr (READ == 4 || READ == 5) {
            if constexpr (READ == 4) { SEQ_GAP(READ_STR_4, CONS_STR_0); } else { SEQ_GAP(READ_STR_5, CONS_STR_0); }
            want = s0 * f1 + s0 * f0;
        } else if constexpr (READ == 2 && CONS >= 3) {
            if constexpr (CONS == 3) {
                SEQ_GAP(READ_STR_2, CONS_STR_3);
                want = s0 * f0 + s0 * f1;
            } else if constexpr (CONS == 4) {
                SEQ_GAP(READ_STR_2, CONS_STR_4);
                const double dd = __hiloint2double((int)__float_as_uint(f1), (int)__float_as_uint(f0));
                const double d2 = dd + dd;
                want = __uint_as_float((unsigned)__double2loint(d2) ^ (unsigned)__double2hiint(d2));
            } else {
                SEQ_GAP(READ_STR_2, CONS_STR_5);
                want = (s0 + f0) + (s0 + f1);
            }
        } else if constexpr (READ == 2) {
            if constexpr (CONS == 0) {
                SEQ_GAP(READ_STR_2, CONS_STR_0);
                want = s0 * f1 + s0 * f0;
            } else if constexpr (CONS == 1) {
                SEQ_GAP(READ_STR_2, CONS_B64_1);
                want = f1;
            } else {
                SEQ_GAP(READ_STR_2, CONS_B64_2);
                want = f0 + f1;
            }
        } else {
            SEQ_GAP(READ_STR_3, CONS_B32_1);
            want = f0;
        }
        if (__float_as_uint(got) != __float_as_uint(want)) {
            bad++;
            s_got = __float_as_uint(got);
            s_want = __float_as_uint(want);
        }
    }
    if (bad) {
        atomicAdd(&res->bad, bad);
        atomicAdd(&res->quarter[l >> 4], bad);
        res->sample_got[l >> 4] = s_got;
        res->sample_want[l >> 4] = s_want;
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (l == 0) atomicAdd(&lds[0], 1);
}

static Res *g_res;
static float *g_pat;
static const char *hn[5] = {"idle", "MFMA+LDS", "MFMA", "LDS", "f64 VALU"};

template <int READ, int CONS, int GAP, int BCAST> static void run_one(int hammer, int iters) {
    static const char *rn[6] = {"1 x b128, lgkmcnt(0)", "4 x b128, lgkmcnt(3)", "1 x b64, lgkmcnt(0)", "1 x b32, lgkmcnt(0)", "b128 landed long ago", "global x2, vmcnt(0)"};
    static const char *cn[6] = {"v_pk_mul op_sel", "v_mov last reg", "v_add first+last", "v_pk_mul", "v_add_f64", "v_pk_add"};
    CK(hipMemset(g_res, 0, sizeof(Res)));
    auto k = jit_kernel<READ, CONS, GAP, BCAST>;
    const int lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, g_res, iters, hammer, (const float *)g_pat);
    CK(hipDeviceSynchronize());
    Res r;
    CK(hipMemcpy(&r, g_res, sizeof r, hipMemcpyDeviceToHost));
    float sg = 0, sw = 0;
    for (int q = 0; q < 4; q++)
        if (r.quarter[q]) {
            unsigned a = r.sample_got[q], b = r.sample_want[q];
            memcpy(&sg, &a, 4);
            memcpy(&sw, &b, 4);
        }
    printf("  %-21s %-9s -> %d wait states -> %-17s partner %-8s: wrong %5llu of %.3g  quarters %llu %llu %llu %llu", rn[READ], BCAST ? "one addr" : "per lane", GAP,
           cn[CONS], hn[hammer], r.bad, 256.0 * 4 * 64 * iters, r.quarter[0], r.quarter[1], r.quarter[2], r.quarter[3]);
    if (r.bad) printf("   (e.g. got %g, want %g)", sg, sw);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    CK(hipMalloc(&g_res, sizeof(Res)));
    {
        float h[4096];
        for (int i = 0; i < 4096; i++) h[i] = 1.0f + (float)i * (1.0f / 8192.0f);
        CK(hipMalloc(&g_pat, sizeof h));
        CK(hipMemcpy(g_pat, h, sizeof h, hipMemcpyHostToDevice));
    }
    printf("%d iterations x 256 CUs x 4 waves x 64 lanes; the destination registers hold -7777 - (it & 255) before each read\n", iters);
    // what the partner does (the kernel's read, consumer and gap)
    for (int h = 0; h < 5; h++) run_one<1, 0, 0, 1>(h, iters);
    // the gap
    run_one<1, 0, 1, 1>(1, iters);
    run_one<1, 0, 2, 1>(1, iters);
    run_one<1, 0, 4, 1>(1, iters);
    run_one<1, 0, 8, 1>(1, iters);
    // the consumer
    run_one<1, 1, 0, 1>(1, iters);
    run_one<1, 2, 0, 1>(1, iters);
    // the read
    run_one<0, 0, 0, 1>(1, iters);
    run_one<0, 1, 0, 1>(1, iters);
    run_one<0, 2, 0, 1>(1, iters);
    run_one<2, 0, 0, 1>(1, iters);
    run_one<2, 1, 0, 1>(1, iters);
    run_one<3, 1, 0, 1>(1, iters);
    // one address per lane
    run_one<1, 0, 0, 0>(1, iters);
    run_one<0, 1, 0, 0>(1, iters);
    run_one<2, 1, 0, 0>(1, iters);
    run_one<3, 1, 0, 0>(1, iters);
    // with a gap, the other reads
    run_one<0, 1, 2, 1>(1, iters);
    run_one<0, 1, 4, 1>(1, iters);
    run_one<2, 1, 2, 1>(1, iters);
    // other consumers with a 64-bit operand, right behind the s_waitcnt
    run_one<2, 3, 0, 1>(1, iters);
    run_one<2, 4, 0, 1>(1, iters);
    run_one<2, 5, 0, 1>(1, iters);
    run_one<2, 3, 1, 1>(1, iters);
    run_one<2, 4, 1, 1>(1, iters);
    // a wait that does not stall; a global load
    run_one<4, 0, 0, 1>(1, iters);
    run_one<5, 0, 0, 1>(1, iters);
    run_one<5, 0, 1, 1>(1, iters);
    return 0;
}
