// EXPERIMENT: operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 (one wave).  A [32][64] e4m3, B [32][64] e4m3 (row = output
// row / column, K contiguous), scale words per (row, khalf) with 4 bytes each; C [32][32] fp32 = (A.sa)(B.sb)^T.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OPS>
__global__ void probe(const unsigned char* A, const unsigned char* B, const unsigned* sa, const unsigned* sb, float* C, int layout) {
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  v8i a, b;
  for (int j = 0; j < 8; ++j) {
    int k;
    if (layout == 0) k = hi * 32 + j * 4;                       // lane half = K half, 32 contiguous bytes
    else k = (j >> 2) * 32 + hi * 16 + (j & 3) * 4;             // two 16-byte groups, one from each K half
    a[j] = *(const int*)(A + l31 * 64 + k);
    b[j] = *(const int*)(B + l31 * 64 + k);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPS, sa[l31 * 2 + hi], OPS, sb[l31 * 2 + hi]);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;            // C/D layout of the 32x32 shapes: row from reg, col = lane & 31
    C[row * 32 + l31] = c[r];
  }
}
extern "C" int mxprobe(int opsel, int layout, const void* A, const void* B, const void* sa, const void* sb, void* C, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (opsel == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, s, (const unsigned char*)A, (const unsigned char*)B, (const unsigned*)sa, (const unsigned*)sb, (float*)C, layout);
  else if (opsel == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, s, (const unsigned char*)A, (const unsigned char*)B, (const unsigned*)sa, (const unsigned*)sb, (float*)C, layout);
  else if (opsel == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, s, (const unsigned char*)A, (const unsigned char*)B, (const unsigned*)sa, (const unsigned*)sb, (float*)C, layout);
  else hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, s, (const unsigned char*)A, (const unsigned char*)B, (const unsigned*)sa, (const unsigned*)sb, (float*)C, layout);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
