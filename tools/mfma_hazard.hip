// mfma_hazard.hip -- how many wait states v_mfma_i32_32x32x32_i8 REALLY needs on this chip, alone and with a
// second wave on the same SIMD keeping the matrix pipe (or the vector pipe) busy.
//
// Why: hz::mm2::fir_mm2_kernel (csrc/hz_firmm2.h) showed, on some boxes, one pass in ~10^5 whose result
// differed in a quarter-wave of one accumulator register by the last window step's term.  Quarter-wave + last
// step reads like a register-file hazard behind the matrix loop's last MFMAs.  The compiler pads such hazards
// with s_nop from a table (12 wait states from an 8-pass XDL write to a VALU read on gfx950, 1 from an MFMA
// to a VALU write of its A/B registers, 7 to a write of its C registers); whether the table holds with TWO
// waves per SIMD sharing the pipe is what this program measures, with the hazards written in inline assembly
// (the compiler's hazard recognizer does not look inside an asm block) and the distance K swept:
//   mode 0  RAW:   4 MFMAs, K wait states, v_mov from the LAST MFMA's destination (first and last register)
//                  and from the third's -- compared with the same registers read after a long wait
//   mode 1  WAR-B: 4 MFMAs, K wait states, VALU write of the last MFMA's B registers -- results compared with
//                  the clean ones
//   mode 2  WAR-A: the same for the A registers
//   mode 3  WAR-C: MFMA with a separate C operand, K wait states, VALU write of C
//   mode 4  RAW-C: VALU write of C, K wait states, MFMA reading it
// One workgroup of 8 waves per CU (the product kernel's shape: wave w and w + 4 share a SIMD); waves 0-3 run
// the test, waves 4-7 idle / issue independent MFMAs back to back / issue float64 vector instructions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Res {
    unsigned long long bad[4];   // mismatching (iteration, lane) pairs per probe
    unsigned long long quarter[4];  // ... of probe 0 by quarter-wave
};


// registers: A v[40:43], B v[44:47], C0 v[48:63], C1 v[64:79], C2 v[80:95], C3 v[96:111], scratch C v[112:127]
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", \
             "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", \
             "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", \
             "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
             "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

#define SETUP                                                                                                  \
    "v_mov_b32 v40, %[a0]\n v_mov_b32 v41, %[a1]\n v_mov_b32 v42, %[a2]\n v_mov_b32 v43, %[a3]\n"              \
    "v_mov_b32 v44, %[b0]\n v_mov_b32 v45, %[b1]\n v_mov_b32 v46, %[b2]\n v_mov_b32 v47, %[b3]\n"
#define WAIT_LONG "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"

// zero v48 .. v127
#define ZERO_ALL                                                                                               \
    "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n"   \
    "v_mov_b32 v55, 0\n v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n"   \
    "v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n"   \
    "v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n"   \
    "v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n v_mov_b32 v82, 0\n"   \
    "v_mov_b32 v83, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n v_mov_b32 v88, 0\n v_mov_b32 v89, 0\n"   \
    "v_mov_b32 v90, 0\n v_mov_b32 v91, 0\n v_mov_b32 v92, 0\n v_mov_b32 v93, 0\n v_mov_b32 v94, 0\n v_mov_b32 v95, 0\n v_mov_b32 v96, 0\n"   \
    "v_mov_b32 v97, 0\n v_mov_b32 v98, 0\n v_mov_b32 v99, 0\n v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n" \
    "v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n" \
    "v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n" \
    "v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n v_mov_b32 v124, 0\n" \
    "v_mov_b32 v125, 0\n v_mov_b32 v126, 0\n v_mov_b32 v127, 0\n"

#define MFMA4                                                            \
    "v_mfma_i32_32x32x32_i8 v[48:63], v[40:43], v[44:47], v[48:63]\n"    \
    "v_mfma_i32_32x32x32_i8 v[64:79], v[40:43], v[44:47], v[64:79]\n"    \
    "v_mfma_i32_32x32x32_i8 v[80:95], v[40:43], v[44:47], v[80:95]\n"    \
    "v_mfma_i32_32x32x32_i8 v[96:111], v[40:43], v[44:47], v[96:111]\n"

#define OPS                                                                                                                         \
    : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2), [f3] "=&v"(f3) \
    : [a0] "v"(A[0]), [a1] "v"(A[1]), [a2] "v"(A[2]), [a3] "v"(A[3]), [b0] "v"(B[0]), [b1] "v"(B[1]), [b2] "v"(B[2]), [b3] "v"(B[3]), \
      [junk] "v"(junk)                                                                                                              \
    : CLOB

template <int MODE, int K>
__device__ __forceinline__ void probe(const v4i A, const v4i B, int junk, int &o0, int &o1, int &o2, int &o3, int &f0, int &f1, int &f2, int &f3) {
#define BODY(NOP)                                                                                                              \
    if constexpr (MODE == 0) {                                                                                                \
        asm volatile(SETUP ZERO_ALL WAIT_LONG MFMA4 NOP                                                                        \
                     "v_mov_b32 %[o0], v96\n v_mov_b32 %[o1], v111\n v_mov_b32 %[o2], v80\n v_mov_b32 %[o3], v95\n" WAIT_LONG    \
                     "v_mov_b32 %[f0], v96\n v_mov_b32 %[f1], v111\n v_mov_b32 %[f2], v80\n v_mov_b32 %[f3], v95\n" OPS);      \
    } else if constexpr (MODE == 1) {                                                                                         \
        /* clean results first, then the same with the last MFMA's B registers overwritten K wait states behind it */         \
        asm volatile(SETUP ZERO_ALL WAIT_LONG MFMA4 WAIT_LONG                                                                  \
                     "v_mov_b32 %[f0], v96\n v_mov_b32 %[f1], v111\n v_mov_b32 %[f2], v100\n v_mov_b32 %[f3], v105\n" ZERO_ALL WAIT_LONG MFMA4 NOP \
                     "v_mov_b32 v44, %[junk]\n v_mov_b32 v45, %[junk]\n v_mov_b32 v46, %[junk]\n v_mov_b32 v47, %[junk]\n" WAIT_LONG \
                     "v_mov_b32 %[o0], v96\n v_mov_b32 %[o1], v111\n v_mov_b32 %[o2], v100\n v_mov_b32 %[o3], v105\n" OPS);    \
    } else if constexpr (MODE == 2) {                                                                                         \
        asm volatile(SETUP ZERO_ALL WAIT_LONG MFMA4 WAIT_LONG                                                                  \
                     "v_mov_b32 %[f0], v96\n v_mov_b32 %[f1], v111\n v_mov_b32 %[f2], v100\n v_mov_b32 %[f3], v105\n" ZERO_ALL WAIT_LONG MFMA4 NOP \
                     "v_mov_b32 v40, %[junk]\n v_mov_b32 v41, %[junk]\n v_mov_b32 v42, %[junk]\n v_mov_b32 v43, %[junk]\n" WAIT_LONG \
                     "v_mov_b32 %[o0], v96\n v_mov_b32 %[o1], v111\n v_mov_b32 %[o2], v100\n v_mov_b32 %[o3], v105\n" OPS);    \
    } else if constexpr (MODE == 3) {                                                                                         \
        /* C = v[112:127] (zeros), D = v[96:111]; C overwritten K wait states behind the MFMA */                               \
        asm volatile(SETUP ZERO_ALL WAIT_LONG                                                                                  \
                     "v_mfma_i32_32x32x32_i8 v[96:111], v[40:43], v[44:47], v[112:127]\n" WAIT_LONG                             \
                     "v_mov_b32 %[f0], v96\n v_mov_b32 %[f1], v111\n v_mov_b32 %[f2], v100\n v_mov_b32 %[f3], v105\n" ZERO_ALL WAIT_LONG \
                     "v_mfma_i32_32x32x32_i8 v[64:79], v[40:43], v[44:47], v[64:79]\n"                                         \
                     "v_mfma_i32_32x32x32_i8 v[96:111], v[40:43], v[44:47], v[112:127]\n" NOP                                   \
                     "v_mov_b32 v112, %[junk]\n v_mov_b32 v127, %[junk]\n v_mov_b32 v116, %[junk]\n v_mov_b32 v121, %[junk]\n" WAIT_LONG \
                     "v_mov_b32 %[o0], v96\n v_mov_b32 %[o1], v111\n v_mov_b32 %[o2], v100\n v_mov_b32 %[o3], v105\n" OPS);    \
    } else {                                                                                                                  \
        /* C written (to a known non-zero value) K wait states in front of the MFMA that reads it */                          \
        asm volatile(SETUP ZERO_ALL WAIT_LONG                                                                                  \
                     "v_mov_b32 v112, %[junk]\n v_mov_b32 v127, %[junk]\n v_mov_b32 v116, %[junk]\n v_mov_b32 v121, %[junk]\n" WAIT_LONG \
                     "v_mfma_i32_32x32x32_i8 v[96:111], v[40:43], v[44:47], v[112:127]\n" WAIT_LONG                             \
                     "v_mov_b32 %[f0], v96\n v_mov_b32 %[f1], v111\n v_mov_b32 %[f2], v100\n v_mov_b32 %[f3], v105\n" ZERO_ALL WAIT_LONG \
                     "v_mfma_i32_32x32x32_i8 v[64:79], v[40:43], v[44:47], v[64:79]\n"                                         \
                     "v_mov_b32 v112, %[junk]\n v_mov_b32 v127, %[junk]\n v_mov_b32 v116, %[junk]\n v_mov_b32 v121, %[junk]\n" NOP \
                     "v_mfma_i32_32x32x32_i8 v[96:111], v[40:43], v[44:47], v[112:127]\n" WAIT_LONG                             \
                     "v_mov_b32 %[o0], v96\n v_mov_b32 %[o1], v111\n v_mov_b32 %[o2], v100\n v_mov_b32 %[o3], v105\n" OPS);    \
    }
    if constexpr (K == 0) { BODY("") }
    else if constexpr (K == 1) { BODY("s_nop 0\n") }
    else if constexpr (K == 2) { BODY("s_nop 1\n") }
    else if constexpr (K == 3) { BODY("s_nop 2\n") }
    else if constexpr (K == 4) { BODY("s_nop 3\n") }
    else if constexpr (K == 5) { BODY("s_nop 4\n") }
    else if constexpr (K == 6) { BODY("s_nop 5\n") }
    else if constexpr (K == 7) { BODY("s_nop 6\n") }
    else if constexpr (K == 8) { BODY("s_nop 7\n") }
    else if constexpr (K == 9) { BODY("s_nop 8\n") }
    else if constexpr (K == 10) { BODY("s_nop 9\n") }
    else if constexpr (K == 11) { BODY("s_nop 10\n") }
    else if constexpr (K == 12) { BODY("s_nop 11\n") }
    else if constexpr (K == 13) { BODY("s_nop 12\n") }
    else if constexpr (K == 14) { BODY("s_nop 13\n") }
    else if constexpr (K == 16) { BODY("s_nop 15\n") }
    else if constexpr (K == 20) { BODY("s_nop 15\n s_nop 3\n") }
    else { BODY("s_nop 15\n s_nop 15\n") }  // 32
#undef BODY
}

template <int MODE, int K>
__global__ __launch_bounds__(512) void hazard_kernel(const v4i *__restrict__ ab, Res *res, int iters, int hammer) {
    extern __shared__ int lds[];
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    if (tid == 0) lds[0] = 0;
    __syncthreads();
    if (wave >= 4) {
        if (hammer == 1) {  // independent MFMAs back to back until the test waves are done
            v4i A = ab[l], B = ab[64 + l];
            v16i c0{}, c1{}, c2{}, c3{};
            while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, c3, 0, 0, 0);
                }
            }
            if (c0[0] + c1[1] + c2[2] + c3[3] == 0x12345678) res[0].bad[3] = 1;  // (keeps the loop alive)
        } else if (hammer == 2) {  // float64 vector instructions (the epilogue's kind)
            double x = (double)l, y = 1.0000001, z = 0.5;
            while (__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
#pragma unroll
                for (int u = 0; u < 32; u++) {
                    x = __builtin_fma(x, y, z);
                    z = __builtin_fma(z, y, x);
                }
            }
            if (x + z == 0.123) res[0].bad[3] = 1;
        }
        return;
    }
    const v4i A = ab[l + 64 * (wave & 1)], B = ab[128 + l + 64 * (wave >> 1)];
    unsigned long long bad[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        int o0, o1, o2, o3, f0, f1, f2, f3;
        probe<MODE, K>(A, B, 0x7f7f7f7f ^ it, o0, o1, o2, o3, f0, f1, f2, f3);
        bad[0] += o0 != f0;
        bad[1] += o1 != f1;
        bad[2] += o2 != f2;
        bad[3] += o3 != f3;
    }
    for (int p = 0; p < 4; p++)
        if (bad[p]) atomicAdd(&res->bad[p], bad[p]);
    if (bad[0] | bad[1] | bad[2] | bad[3]) atomicAdd(&res->quarter[l >> 4], bad[0] + bad[1] + bad[2] + bad[3]);
    __builtin_amdgcn_s_waitcnt(0);
    if (l == 0) atomicAdd(&lds[0], 1);
}

static v4i *g_ab;
static Res *g_res;

template <int MODE, int K> static void run_one(int hammer, int iters) {
    CK(hipMemset(g_res, 0, sizeof(Res)));
    auto k = hazard_kernel<MODE, K>;
    const int lds = 100 * 1024;  // one workgroup per CU
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, g_ab, g_res, iters, hammer);
    CK(hipDeviceSynchronize());
    Res r;
    CK(hipMemcpy(&r, g_res, sizeof r, hipMemcpyDeviceToHost));
    const double tot = 256.0 * 4 * 64 * iters;
    printf("  K %2d: bad %.3g %.3g %.3g %.3g of %.3g   quarter-waves %llu %llu %llu %llu\n", K, (double)r.bad[0], (double)r.bad[1], (double)r.bad[2],
           (double)r.bad[3], tot, r.quarter[0], r.quarter[1], r.quarter[2], r.quarter[3]);
}

template <int MODE> static void sweep(const char *name, int iters) {
    static const char *hn[3] = {"partner wave idle", "partner wave: MFMAs back to back", "partner wave: float64 vector instructions"};
    for (int hammer = 0; hammer < 3; hammer++) {
        printf("%s -- %s\n", name, hn[hammer]);
        run_one<MODE, 0>(hammer, iters);
        run_one<MODE, 1>(hammer, iters);
        run_one<MODE, 2>(hammer, iters);
        run_one<MODE, 3>(hammer, iters);
        run_one<MODE, 4>(hammer, iters);
        run_one<MODE, 5>(hammer, iters);
        run_one<MODE, 6>(hammer, iters);
        run_one<MODE, 7>(hammer, iters);
        run_one<MODE, 8>(hammer, iters);
        run_one<MODE, 9>(hammer, iters);
        run_one<MODE, 10>(hammer, iters);
        run_one<MODE, 11>(hammer, iters);
        run_one<MODE, 12>(hammer, iters);
        run_one<MODE, 13>(hammer, iters);
        run_one<MODE, 14>(hammer, iters);
        run_one<MODE, 16>(hammer, iters);
        run_one<MODE, 20>(hammer, iters);
        run_one<MODE, 32>(hammer, iters);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    std::vector<int> h(4 * 256);
    unsigned s = 12345;
    for (auto &v : h) {
        s = s * 1664525u + 1013904223u;
        v = (int)s;
    }
    CK(hipMalloc(&g_ab, h.size() * 4));
    CK(hipMemcpy(g_ab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&g_res, sizeof(Res)));
    printf("v_mfma_i32_32x32x32_i8 hazards, K = wait states between the two instructions; probes: RAW = (last MFMA reg 0, reg 15, third MFMA reg 0, "
           "reg 15); WAR = the last MFMA's (reg 0, 15, 4, 9); %d iterations x 256 CUs x 4 waves x 64 lanes\n", iters);
    sweep<0>("RAW  (MFMA result read by a VALU instruction; the compiler pads to 12)", iters);
    sweep<1>("WAR-B (VALU write of the MFMA's B registers; the compiler pads to 1)", iters);
    sweep<2>("WAR-A (VALU write of the MFMA's A registers)", iters);
    sweep<3>("WAR-C (VALU write of the MFMA's C registers; the compiler pads to 7)", iters);
    sweep<4>("RAW-C (VALU write of C in front of the MFMA that reads it)", iters);
    return 0;
}
