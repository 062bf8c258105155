// PROBE: what v_exp_legacy_f32 returns on gfx950 for the inputs the attention kernel feeds its exponential
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float* x, float* a, float* b, int n) {
  int i = threadIdx.x;
  if (i < n) {
    float p, q = __builtin_amdgcn_exp2f(x[i]);
    asm volatile("v_exp_legacy_f32 %0, %1" : "=v"(p) : "v"(x[i]));
    a[i] = p; b[i] = q;
  }
}
int main() {
  float h[] = {0.f, 1.f, -1.f, 0.5f, -0.5f, 10.f, -10.f, -50.5f, -100.f, -126.f, -127.f, -140.f, -150.f, -1000.f, -1e30f, -INFINITY, 3.3f, -23.7f, 1e-30f, -1e-30f, 127.f, 128.f};
  const int n = sizeof(h) / sizeof(float);
  float *x, *a, *b, ra[64], rb[64];
  hipMalloc(&x, 256); hipMalloc(&a, 256); hipMalloc(&b, 256);
  hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, a, b, n);
  hipMemcpy(ra, a, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(rb, b, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("x = %-12g legacy %-14.8g v_exp_f32 %-14.8g exact %-14.8g rel %.2e\n", h[i], ra[i], rb[i], exp2(h[i]), (ra[i] - exp2(h[i])) / exp2(h[i]));
  // precision sweep
  double worst = 0, worst2 = 0;
  for (int rep = 0; rep < 40; ++rep) {
    float hx[64];
    for (int i = 0; i < 64; ++i) hx[i] = -(float)(rep * 64 + i) * 0.0137f - 0.001f;
    hipMemcpy(x, hx, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, a, b, 64);
    hipMemcpy(ra, a, 256, hipMemcpyDeviceToHost); hipMemcpy(rb, b, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) { double e = exp2((double)hx[i]); worst = fmax(worst, fabs(ra[i] - e) / e); worst2 = fmax(worst2, fabs(rb[i] - e) / e); }
  }
  printf("max rel err on [-35, 0]: legacy %.3e  v_exp_f32 %.3e\n", worst, worst2);
  return 0;
}
