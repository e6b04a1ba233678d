// Probe: DPP wave64 reductions (4 in-row DPP steps + row_bcast15/31 + readlane 63) against the __shfl_xor form
// (which lowers to six ds_bpermute_b32 on gfx950).  Build: hipcc -O3 --offload-arch=gfx950 wave_reduce.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL, int ROWMASK = 0xf> __device__ __forceinline__ float dpp_or(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_or<0xB1>(0.f, v);          // quad_perm [1,0,3,2]
  v += dpp_or<0x4E>(0.f, v);          // quad_perm [2,3,0,1]
  v += dpp_or<0x141>(0.f, v);         // row_half_mirror
  v += dpp_or<0x140>(0.f, v);         // row_mirror: every lane of a row holds the row's sum
  v += dpp_or<0x142, 0xa>(0.f, v);    // row_bcast:15 into rows 1,3
  v += dpp_or<0x143, 0xc>(0.f, v);    // row_bcast:31 into rows 2,3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, dpp_or<0xB1>(v, v));
  v = fmaxf(v, dpp_or<0x4E>(v, v));
  v = fmaxf(v, dpp_or<0x141>(v, v));
  v = fmaxf(v, dpp_or<0x140>(v, v));
  v = fmaxf(v, dpp_or<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_or<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__global__ void k(const float* x, float* o) {
  float v = x[threadIdx.x];
  o[threadIdx.x] = wave_sum_dpp(v);
  o[64 + threadIdx.x] = wave_max_dpp(v);
}
int main() {
  float h[64], r[128], *x, *o; double s = 0; float m = -1e30f;
  for (int i = 0; i < 64; ++i) { h[i] = (float)((i * 37) % 101) - 40.f; s += h[i]; m = fmaxf(m, h[i]); }
  hipMalloc(&x, 256); hipMalloc(&o, 512); hipMemcpy(x, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, o);
  hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) bad += (r[i] != (float)s) + (r[64 + i] != m);
  printf("sum %g got %g | max %g got %g | bad %d\n", s, r[0], m, r[64], bad);
  return bad != 0;
}
