// issue_probe.hip — scalar-ALU issue rate of an MI355X SIMD, alone and beside vector work (tools/probe/issue_probe.py).
// Every wave runs `iters` x 64 instructions:
//   mode 0: s_mov_b32 with a literal (independent)            mode 1: v_pk_add_f32 on 8 independent register pairs
//   mode 2: the two alternating inside ONE wave               mode 3: even waves of a workgroup run mode 0, odd waves mode 1
//   mode 6: 512-thread workgroups, waves 0-3 run mode 0 and waves 4-7 mode 1 (one of each per SIMD)
//   mode 4: s_mov_b64 (SGPR pair copy)                        mode 5: s_add_u32 / s_addc_u32 pairs (dependent through SCC)
// The caller picks waves per SIMD (workgroups of 256 threads = one wave per SIMD; blocks = 256 CUs x waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float F2 __attribute__((ext_vector_type(2)));
#define S8 "s_mov_b32 s20, 0x3f800001\n s_mov_b32 s21, 0x3f800002\n s_mov_b32 s22, 0x3f800003\n s_mov_b32 s23, 0x3f800004\n" \
           "s_mov_b32 s24, 0x3f800005\n s_mov_b32 s25, 0x3f800006\n s_mov_b32 s26, 0x3f800007\n s_mov_b32 s27, 0x3f800008\n"
#define V8 "v_pk_add_f32 v[20:21], v[20:21], v[36:37]\n v_pk_add_f32 v[22:23], v[22:23], v[36:37]\n v_pk_add_f32 v[24:25], v[24:25], v[36:37]\n v_pk_add_f32 v[26:27], v[26:27], v[36:37]\n" \
           "v_pk_add_f32 v[28:29], v[28:29], v[36:37]\n v_pk_add_f32 v[30:31], v[30:31], v[36:37]\n v_pk_add_f32 v[32:33], v[32:33], v[36:37]\n v_pk_add_f32 v[34:35], v[34:35], v[36:37]\n"
#define SV8 "s_mov_b32 s20, 0x3f800001\n v_pk_add_f32 v[20:21], v[20:21], v[36:37]\n s_mov_b32 s21, 0x3f800002\n v_pk_add_f32 v[22:23], v[22:23], v[36:37]\n" \
            "s_mov_b32 s22, 0x3f800003\n v_pk_add_f32 v[24:25], v[24:25], v[36:37]\n s_mov_b32 s23, 0x3f800004\n v_pk_add_f32 v[26:27], v[26:27], v[36:37]\n"
#define M8 "s_mov_b64 s[20:21], s[28:29]\n s_mov_b64 s[22:23], s[28:29]\n s_mov_b64 s[24:25], s[28:29]\n s_mov_b64 s[26:27], s[28:29]\n" \
           "s_mov_b64 s[20:21], s[28:29]\n s_mov_b64 s[22:23], s[28:29]\n s_mov_b64 s[24:25], s[28:29]\n s_mov_b64 s[26:27], s[28:29]\n"
#define A8 "s_add_u32 s20, s20, 16\n s_addc_u32 s21, s21, 0\n s_add_u32 s22, s22, 16\n s_addc_u32 s23, s23, 0\n" \
           "s_add_u32 s24, s24, 16\n s_addc_u32 s25, s25, 0\n s_add_u32 s26, s26, 16\n s_addc_u32 s27, s27, 0\n"
#define CLOB "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37"
#define X8(B) B B B B B B B B
__global__ void __launch_bounds__(512) issue_probe(int mode, int iters, float *out) {
    const int wave = threadIdx.x >> 6;
    int m = mode;
    if (mode == 3) m = (wave & 1);       // (wave w runs on SIMD w % 4: SALU-only and VALU-only SIMDs)
    if (mode == 6) m = (wave >> 2) & 1;  // 512 threads: every SIMD holds one wave of each kind
    asm volatile("v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n s_mov_b64 s[28:29], 0\n s_mov_b64 s[20:21], 0\n s_mov_b64 s[22:23], 0\n s_mov_b64 s[24:25], 0\n s_mov_b64 s[26:27], 0" ::: CLOB);
    for (int it = 0; it < iters; it++) {
        if (m == 0) asm volatile(X8(S8) ::: CLOB);
        else if (m == 1) asm volatile(X8(V8) ::: CLOB);
        else if (m == 2) asm volatile(X8(SV8) ::: CLOB);
        else if (m == 4) asm volatile(X8(M8) ::: CLOB);
        else asm volatile(X8(A8) ::: CLOB);
    }
    if (iters < 0) out[threadIdx.x] = 1.0f;
}
extern "C" int issue_probe_run(int mode, int iters, int blocks, int threads, float *ms) {
    float *d = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(issue_probe, dim3(blocks), dim3(threads), 0, 0, mode, iters, d);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return 2;
    }
    (void)hipEventElapsedTime(ms, e0, e1);
    (void)hipFree(d);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
