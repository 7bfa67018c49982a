// tools/ubench_copy.hip -- achievable HBM bandwidth of a plain copy on this device, the
// ceiling the 16 B/element roofline is held against ("also report a measured
// copy-kernel peak").  GPU counterpart of the reference's tests/bench-stream-cmg.cpp memcpy case.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o tools/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)

template <class T> __global__ __launch_bounds__(256) void copy_kernel(T *dst, const T *src, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) dst[i] = src[i];
}

template <class T> static int run(const char *name, size_t bytes) {
  T *a, *b;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
  CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
  const size_t n = bytes / sizeof(T);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 12; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(copy_kernel<T>, dim3(256 * 8), dim3(256), 0, 0, b, a, n);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2 && ms < best) best = ms;
  }
  printf("%-28s %6zu MiB  %8.1f us  %7.2f TB/s (read+write)\n", name, bytes >> 20, best * 1e3,
         2.0 * bytes / (best * 1e-3) / 1e12);
  CHECK(hipFree(a)); CHECK(hipFree(b));
  return 0;
}

int main() {
  for (size_t mib : {128, 1024, 4096}) {
    run<uint64_t>("copy, 8 B per lane", mib << 20);
    run<ulonglong2>("copy, 16 B per lane", mib << 20);
  }
  return 0;
}
