// pk_hazard.hip -- is a packed-float32 result safe to read one wait state later?  And is the last beat of a
// ds_read_b128 in its registers when s_waitcnt lgkmcnt lets the wave go?
//
// The non-repeatable outputs of hz::mm2::fir_mm2_kernel (tools/mm2_glitch.hip) sit in the mixer's FIRST step
// factor: lanes 48-63 (the last quarter-wave) of one output row, a wave's first pass, few ulp.  The compiler's
// code for that factor is
//       ds_read_b128 v[14:17], ...            (the factor: cos hi, sin hi, cos lo, sin lo; all lanes one address)
//       3 vector instructions, s_waitcnt lgkmcnt(3)
//       v_pk_mul_f32 v[26:27], v[30:31], v[14:15] ...
//       v_fmac_f32   v13, v36, v30                        <- ONE independent instruction
//       v_pk_fma_f32 v[14:15], v[32:33], v[14:15], v[26:27]  <- reads the v_pk_mul's result
// and for the later factors the same with `s_nop 0` in the place of the v_fmac (LLVM's hazard recognizer
// wants one wait state behind a packed float32 instruction).  This program runs both sequences -- and the
// bare ds_read_b128 -> s_waitcnt -> use -- in inline assembly with poisoned destination registers, on every
// CU, with the SIMD's second wave idle / issuing MFMAs and LDS reads back to back, and counts wrong lanes
// by quarter-wave.
//   mode 0: v_pk_mul -> FILL -> v_pk_fma (reads it as C)      FILL = nothing | s_nop 0 | v_fmac_f32 | 2 x v_fmac_f32
//   mode 1: v_pk_mul -> FILL -> v_add_f32 reading the HIGH half
//   mode 2: ds_read_b128 (one address for all lanes) -> s_waitcnt lgkmcnt(0) -> v_mov from the 4th register
//   mode 3: four ds_read_b128, lgkmcnt(3) -> use of the first (as the kernel does)
//   mode 4: the kernel's instruction sequence for its first factor, verbatim (registers renamed): four ds_read_b128,
//           three vector instructions, lgkmcnt(3), v_pk_mul with op_sel / neg_lo on the just-landed pair, v_fmac,
//           v_pk_fma; FILL = 1: s_nop 3 between the s_waitcnt and the v_pk_mul
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Res {
    unsigned long long bad, quarter[4], sample[4];
};

template <int MODE, int FILL>
__global__ __launch_bounds__(512) void pk_kernel(const v4i *__restrict__ ab, Res *res, int iters, int hammer) {
    extern __shared__ int lds[];
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    float *fl = reinterpret_cast<float *>(lds);
    // LDS: [0] the done counter; floats 64 .. 64 + 4 * 256: the "factors"; from 16 KB: the partner's operands
    for (int i = tid; i < 1024; i += 512) fl[64 + i] = 1.0f + (float)i * (1.0f / 4096.0f);
    for (int i = tid; i < 8192; i += 512) lds[4096 + i] = i * 2654435761u;
    if (tid == 0) lds[0] = 0;
    __syncthreads();
    if (wave >= 4) {
        if (hammer == 1) {  // the product's matrix loop in miniature: four ds_read_b128 and four MFMAs per step
            v16i c0{}, c1{}, c2{}, c3{};
            const v4i *base = reinterpret_cast<const v4i *>(lds + 4096) + l;
            while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const v4i a0 = base[64 * ((4 * u) & 31)], a1 = base[64 * ((4 * u + 1) & 31)], b0 = base[64 * ((4 * u + 2) & 31)], b1 = base[64 * ((4 * u + 3) & 31)];
                    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, c3, 0, 0, 0);
                }
            }
            if (c0[0] + c1[1] + c2[2] + c3[3] == 0x12345678) res[0].bad = 1;
        }
        return;
    }
    unsigned long long bad = 0;
    unsigned sample = 0;
    const float x = 1.0f + (float)l * 0.001f, y = 0.5f + (float)wave * 0.01f;
    for (int it = 0; it < iters; it++) {
        const float p = 3.0f + (float)(it & 15), poison = 1000.0f + (float)it;
        float got = 0.f, want = 0.f;
        if constexpr (MODE == 0 || MODE == 1) {
            // v[26:27] poisoned; v[26:27] = (x p, y p); then C-read (mode 0: pk_fma -> lo + hi; mode 1: v_add of the hi half)
            float lo, hi, t = y;
#define PKSEQ(FILLSTR)                                                                                         \
    asm volatile("v_mov_b32 v26, %[poi]\n v_mov_b32 v27, %[poi]\n v_mov_b32 v30, %[x]\n v_mov_b32 v31, %[y]\n"    \
                 "v_mov_b32 v32, %[p]\n v_mov_b32 v33, %[p]\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n s_nop 7\n"    \
                 "v_pk_mul_f32 v[26:27], v[30:31], v[32:33]\n" FILLSTR                                          \
                 "v_pk_fma_f32 v[14:15], v[14:15], v[14:15], v[26:27]\n s_nop 7\n"                               \
                 "v_mov_b32 %[lo], v14\n v_mov_b32 %[hi], v15\n"                                                \
                 : [lo] "=&v"(lo), [hi] "=&v"(hi), [t] "+v"(t)                                                  \
                 : [poi] "v"(poison), [x] "v"(x), [y] "v"(y), [p] "v"(p)                                        \
                 : "v14", "v15", "v26", "v27", "v30", "v31", "v32", "v33")
#define PKSEQ1(FILLSTR)                                                                                        \
    asm volatile("v_mov_b32 v26, %[poi]\n v_mov_b32 v27, %[poi]\n v_mov_b32 v30, %[x]\n v_mov_b32 v31, %[y]\n"    \
                 "v_mov_b32 v32, %[p]\n v_mov_b32 v33, %[p]\n s_nop 7\n"                                         \
                 "v_pk_mul_f32 v[26:27], v[30:31], v[32:33]\n" FILLSTR                                          \
                 "v_add_f32 %[hi], v27, v27\n v_mov_b32 %[lo], 0\n s_nop 7\n"                                    \
                 : [lo] "=&v"(lo), [hi] "=&v"(hi), [t] "+v"(t)                                                  \
                 : [poi] "v"(poison), [x] "v"(x), [y] "v"(y), [p] "v"(p)                                        \
                 : "v14", "v15", "v26", "v27", "v30", "v31", "v32", "v33")
            if constexpr (MODE == 0) {
                if constexpr (FILL == 0) PKSEQ("");
                else if constexpr (FILL == 1) PKSEQ("s_nop 0\n");
                else if constexpr (FILL == 2) PKSEQ("v_fmac_f32 %[t], v30, v31\n");
                else PKSEQ("v_fmac_f32 %[t], v30, v31\n v_fmac_f32 %[t], v31, v30\n");
                got = lo + hi;
                want = x * p + y * p;
            } else {
                if constexpr (FILL == 0) PKSEQ1("");
                else if constexpr (FILL == 1) PKSEQ1("s_nop 0\n");
                else if constexpr (FILL == 2) PKSEQ1("v_fmac_f32 %[t], v30, v31\n");
                else PKSEQ1("v_fmac_f32 %[t], v30, v31\n v_fmac_f32 %[t], v31, v30\n");
                got = hi;
                want = 2.0f * (y * p);
            }
            if (t == 12345.0f) res->bad = 1;
        } else if constexpr (MODE == 4) {
            const int idx = (it * 7 + wave) & 255;
            const int addr = 4 * (64 + 4 * idx);
            const float s0 = 0.25f + (float)l * 0.001f, c0 = 0.9f + (float)(it & 7) * 0.01f;
            float lo, hi, t0 = x, t1 = y, t2 = x;
#define SEQ4(GAP)                                                                                                                    \
    asm volatile("v_mov_b32 v14, %[poi]\n v_mov_b32 v15, %[poi]\n v_mov_b32 v16, %[poi]\n v_mov_b32 v17, %[poi]\n v_mov_b32 v26, %[poi]\n"   \
                 "v_mov_b32 v27, %[poi]\n v_mov_b32 v30, %[s0]\n v_mov_b32 v31, %[s0]\n v_mov_b32 v32, %[c0]\n v_mov_b32 v33, %[c0]\n s_nop 3\n" \
                 "ds_read_b128 v[14:17], %[a]\n ds_read_b128 v[18:21], %[a] offset:16\n ds_read_b128 v[22:25], %[a] offset:32\n"         \
                 "ds_read_b128 v[44:47], %[a] offset:48\n"                                                                            \
                 "v_mul_f32 %[t0], %[t1], %[t2]\n v_mul_f32 %[t1], %[t2], %[t0]\n v_fmac_f32 %[t2], %[t0], %[t1]\n"                    \
                 "s_waitcnt lgkmcnt(3)\n" GAP                                                                                         \
                 "v_pk_mul_f32 v[26:27], v[30:31], v[14:15] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"                              \
                 "v_fmac_f32 %[t1], %[t0], v30\n"                                                                                     \
                 "v_pk_fma_f32 v[14:15], v[32:33], v[14:15], v[26:27]\n s_nop 7\n s_waitcnt lgkmcnt(0)\n"                               \
                 "v_mov_b32 %[lo], v14\n v_mov_b32 %[hi], v15\n"                                                                      \
                 : [lo] "=&v"(lo), [hi] "=&v"(hi), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2)                                        \
                 : [poi] "v"(poison), [a] "v"(addr), [s0] "v"(s0), [c0] "v"(c0)                                                       \
                 : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v30", "v31", "v32", \
                   "v33", "v44", "v45", "v46", "v47", "memory")
            if constexpr (FILL == 0) SEQ4("");
            else SEQ4("s_nop 3\n");
            if (t0 + t1 + t2 == 12345.0f) res->bad = 1;
            const float wx = 1.0f + (float)(4 * idx) * (1.0f / 4096.0f), wy = 1.0f + (float)(4 * idx + 1) * (1.0f / 4096.0f);
            got = lo + hi * 3.0f;
            want = __fmaf_rn(c0, wx, -(s0 * wy)) + __fmaf_rn(c0, wy, s0 * wx) * 3.0f;
        } else {
            const int idx = (it * 7 + wave) & 255;  // the factor, one address for all lanes
            const int addr = 4 * (64 + 4 * idx);
            float o0, o3;
            if constexpr (MODE == 2) {
                asm volatile("v_mov_b32 v20, %[poi]\n v_mov_b32 v21, %[poi]\n v_mov_b32 v22, %[poi]\n v_mov_b32 v23, %[poi]\n s_nop 3\n"
                             "ds_read_b128 v[20:23], %[a]\n s_waitcnt lgkmcnt(0)\n"
                             "v_mov_b32 %[o3], v23\n v_mov_b32 %[o0], v20\n"
                             : [o0] "=&v"(o0), [o3] "=&v"(o3)
                             : [poi] "v"(poison), [a] "v"(addr)
                             : "v20", "v21", "v22", "v23", "memory");
            } else {
                asm volatile("v_mov_b32 v20, %[poi]\n v_mov_b32 v21, %[poi]\n v_mov_b32 v22, %[poi]\n v_mov_b32 v23, %[poi]\n s_nop 3\n"
                             "ds_read_b128 v[20:23], %[a]\n ds_read_b128 v[40:43], %[a] offset:16\n ds_read_b128 v[44:47], %[a] offset:32\n"
                             "ds_read_b128 v[48:51], %[a] offset:48\n s_waitcnt lgkmcnt(3)\n"
                             "v_mov_b32 %[o3], v23\n v_mov_b32 %[o0], v20\n s_waitcnt lgkmcnt(0)\n"
                             : [o0] "=&v"(o0), [o3] "=&v"(o3)
                             : [poi] "v"(poison), [a] "v"(addr)
                             : "v20", "v21", "v22", "v23", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory");
            }
            got = o0 + o3;
            want = (1.0f + (float)(4 * idx) * (1.0f / 4096.0f)) + (1.0f + (float)(4 * idx + 3) * (1.0f / 4096.0f));
        }
        if (got != want) {
            bad++;
            sample = __float_as_uint(got);
        }
    }
    if (bad) {
        atomicAdd(&res->bad, bad);
        atomicAdd(&res->quarter[l >> 4], bad);
        res->sample[l >> 4] = sample;
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (l == 0) atomicAdd(&lds[0], 1);
}

static v4i *g_ab;
static Res *g_res;

template <int MODE, int FILL> static void run_one(const char *what, int hammer, int iters) {
    CK(hipMemset(g_res, 0, sizeof(Res)));
    auto k = pk_kernel<MODE, FILL>;
    const int lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, g_ab, g_res, iters, hammer);
    CK(hipDeviceSynchronize());
    Res r;
    CK(hipMemcpy(&r, g_res, sizeof r, hipMemcpyDeviceToHost));
    printf("  %-58s partner %-6s: wrong %.4g of %.4g  by quarter-wave %llu %llu %llu %llu\n", what, hammer ? "MFMA+LDS" : "idle", (double)r.bad,
           256.0 * 4 * 64 * iters, r.quarter[0], r.quarter[1], r.quarter[2], r.quarter[3]);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    CK(hipMalloc(&g_ab, 4096));
    CK(hipMemset(g_ab, 1, 4096));
    CK(hipMalloc(&g_res, sizeof(Res)));
    printf("%d iterations x 256 CUs x 4 waves x 64 lanes\n", iters);
    for (int h = 0; h < 2; h++) {
        run_one<0, 0>("v_pk_mul -> v_pk_fma (C operand), back to back", h, iters);
        run_one<0, 1>("v_pk_mul -> s_nop 0 -> v_pk_fma", h, iters);
        run_one<0, 2>("v_pk_mul -> v_fmac_f32 -> v_pk_fma (the kernel's first factor)", h, iters);
        run_one<0, 3>("v_pk_mul -> 2 x v_fmac_f32 -> v_pk_fma", h, iters);
        run_one<1, 0>("v_pk_mul -> v_add_f32 of the high half, back to back", h, iters);
        run_one<1, 1>("v_pk_mul -> s_nop 0 -> v_add_f32 of the high half", h, iters);
        run_one<1, 2>("v_pk_mul -> v_fmac_f32 -> v_add_f32 of the high half", h, iters);
        run_one<2, 0>("ds_read_b128 (one address) -> lgkmcnt(0) -> 4th register", h, iters);
        run_one<3, 0>("4 x ds_read_b128 -> lgkmcnt(3) -> the first one's 4th register", h, iters);
        run_one<4, 0>("the kernel's sequence: 4 reads, lgkmcnt(3), pk_mul, fmac, pk_fma", h, iters);
        run_one<4, 1>("the same with s_nop 3 behind the s_waitcnt", h, iters);
    }
    return 0;
}
