// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs, and does v_cvt_pk_f16_f32 produce them?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, float tiny) {
  f16x8 a, b;
  const f32x2 t = {tiny, tiny * 3.0f};
  const f16x2 c = __builtin_convertvector(t, f16x2);
  for (int i = 0; i < 8; ++i) { a[i] = c[0]; b[i] = (_Float16)1.0f; }
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)c[0]; out[2] = (float)c[1]; }
}
int main() {
  float* d; hipMalloc(&d, 64);
  for (float tiny : {1e-3f, 3e-5f, 1e-6f, 1e-7f, 3e-8f}) {
    k<<<1, 64>>>(d, tiny);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("tiny %.3e: cvt -> %.6e (x3 -> %.6e); mfma sum over k=16 of tiny*1 = %.6e (expected %.6e)\n", tiny, h[1], h[2], h[0], 16.0 * h[1]);
  }
  return 0;
}
