// hw_trig.hip — accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) on gfx950 (tools/probe/hw_trig.py)
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k(const float *x, float *c, float *s, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    c[i] = __builtin_amdgcn_cosf(x[i]); // v_cos_f32: cos(2 pi x)
    s[i] = __builtin_amdgcn_sinf(x[i]);
}
extern "C" int hw_trig(const float *hx, float *hc, float *hs, int n) {
    float *x, *c, *s;
    if (hipMalloc(&x, n * 4) || hipMalloc(&c, n * 4) || hipMalloc(&s, n * 4)) return 1;
    (void)hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, x, c, s, n);
    (void)hipMemcpy(hc, c, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs, s, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(x); (void)hipFree(c); (void)hipFree(s);
    return hipDeviceSynchronize() != hipSuccess;
}
