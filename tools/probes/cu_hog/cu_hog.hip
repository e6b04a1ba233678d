// A stand-in for RCCL's channel kernels on a one-GPU box: `nblocks` workgroups of 256 threads that hold their CUs for `usec`
// microseconds (spinning on the 100-MHz real-time counter, like a channel spins on its flags) on the stream given.  Every wave
// exits when the time is up: bounded by construction.  Not part of liblc2is_hip.so; built by tools/dp_cu_contention.py.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o cu_hog.so cu_hog.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void cu_hog_kernel(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = __builtin_readcyclecounter() * 0ull + wall_clock64();
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) {
    acc += 1;
    __builtin_amdgcn_s_sleep(32);
  }
  if (acc == 0xffffffffu) *sink = acc;   // (never: keeps the loop)
}

extern "C" int cu_hog_launch(int nblocks, int usec, unsigned* sink, void* stream) {
  if (nblocks <= 0 || usec <= 0 || usec > 200000) return -1;   // at most 0.2 s
  hipLaunchKernelGGL(cu_hog_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull, sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
