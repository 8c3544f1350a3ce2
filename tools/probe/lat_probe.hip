// lat_probe.hip — unloaded latencies behind one dispatch of the threaded interpreter on MI355X (tools/probe/lat_probe.py):
//   0: s_load_dwordx4 pointer chase through a 4 KB ring (scalar data cache hits)         -> cycles per dependent scalar load
//   1: the same chase with a stride that never re-uses a line within 64 KB (misses to L2)  -> scalar load that misses the scalar cache
//   2: ds_read_b128 dependent chain (address from the loaded value)                         -> LDS read latency
//   3: s_setpc_b64 chain over 32 blocks spaced 512 bytes (instruction-cache hits after the first pass) -> taken indirect jump
//   4: 3 + an independent s_load_dwordx4 per block that the NEXT block waits for (the interpreter's dispatch with an empty body)
// One wave; time from s_memtime around `iters` repetitions; s_memrealtime (100 MHz) alongside calibrates the s_memtime unit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
typedef uint32_t U4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(64) lat_probe(int mode, int iters, const uint64_t *ring, uint64_t *out) {
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0;
    __syncthreads();
    uint64_t t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    uint64_t p = (uint64_t)ring;
    uint32_t sink = 0;
    if (mode == 0 || mode == 1) {
        // warm the ring
        for (int i = 0; i < (mode == 0 ? 64 : 0); i++) { U4 q; asm volatile("s_load_dwordx4 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(q) : "s"(p) : "memory"); p = ((uint64_t)q.y << 32) | q.x; }
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) { U4 q; asm volatile("s_load_dwordx4 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(q) : "s"(p) : "memory"); p = ((uint64_t)q.y << 32) | q.x; sink += q.z; }
        }
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r1), "=s"(t1)::"memory");
    } else if (mode == 2) {
        uint32_t a = threadIdx.x * 16;
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) { U4 q; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a) : "memory"); a = (a + q.x) & 0x3ff0; sink += q.y; }
        }
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r1), "=s"(t1)::"memory");
    } else {
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
        for (int i = 0; i < iters; i++) {
            if (mode == 3) {
                asm volatile(
                    "s_getpc_b64 s[20:21]\n"
                    "Lb3_%=:\n"
                    "s_add_u32 s20, s20, Lk3_%= - Lb3_%=\n"
                    "s_addc_u32 s21, s21, 0\n"
                    "s_setpc_b64 s[20:21]\n"
                    ".p2align 9\n"
                    "Lk3_%=:\n"
                    ".rept 31\n"
                    "s_add_u32 s20, s20, 512\n"
                    "s_addc_u32 s21, s21, 0\n"
                    "s_setpc_b64 s[20:21]\n"
                    ".p2align 9\n"
                    ".endr\n"
                    ::: "s20", "s21", "scc", "memory");
            } else {
                asm volatile(
                    "s_getpc_b64 s[20:21]\n"
                    "Lb4_%=:\n"
                    "s_add_u32 s20, s20, Lk4_%= - Lb4_%=\n"
                    "s_addc_u32 s21, s21, 0\n"
                    "s_load_dwordx4 s[24:27], %0, 0x0\n"
                    "s_setpc_b64 s[20:21]\n"
                    ".p2align 9\n"
                    "Lk4_%=:\n"
                    ".rept 31\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    "s_load_dwordx4 s[24:27], %0, 0x0\n"
                    "s_add_u32 s20, s20, 512\n"
                    "s_addc_u32 s21, s21, 0\n"
                    "s_setpc_b64 s[20:21]\n"
                    ".p2align 9\n"
                    ".endr\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    :: "s"(p) : "s20", "s21", "s24", "s25", "s26", "s27", "scc", "memory");
            }
        }
        asm volatile("s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r1), "=s"(t1)::"memory");
    }
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = sink + (uint32_t)p; }
}

extern "C" int lat_probe_run(int mode, int iters, int blocks, uint64_t *res3, float *ms) {
    // ring: mode 0 = 256 entries of 16 bytes (4 KB, consecutive); mode 1 = stride 4160 bytes over 8 MB (every load a new line and set)
    const size_t bytes = 16u << 20;
    uint8_t *h = (uint8_t *)calloc(bytes, 1), *d = nullptr;
    uint64_t *o = nullptr, ho[3 * 1024] = {0};
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, sizeof ho) != hipSuccess) return 1;
    const size_t stride = mode == 1 ? 4160 : 16, n = mode == 1 ? 2000 : 256;
    for (size_t i = 0; i < n; i++) { uint64_t nxt = (uint64_t)d + ((i + 1) % n) * stride; memcpy(h + i * stride, &nxt, 8); }
    hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(lat_probe, dim3(blocks), dim3(64), 0, 0, mode, iters, (const uint64_t *)d, o);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return 2;
    }
    hipEventElapsedTime(ms, e0, e1);
    hipMemcpy(ho, o, 24, hipMemcpyDeviceToHost);
    memcpy(res3, ho, 24);
    hipFree(d); hipFree(o); free(h);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
