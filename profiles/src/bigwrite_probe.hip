// What happens to device-memory writes when ONE allocation grows beyond ~128 GB?  (C4 gradient: the forward sweep that stores 156 GB
// of stages runs 1.7 x slower per unit than the one that stores 104 GB, same access pattern per step; reads of the same buffer are fine.)
// Pattern of the sweep kernels: G resident workgroups, workgroup g writes CH bytes at address (s * NB + g) * CH for s = 0 .. steps-1.
//   hipcc --offload-arch=gfx950 -O3 bigwrite_probe.hip -o bigwrite_probe;  ./bigwrite_probe <GB> [mode]   mode 0 one hipMalloc, 1 VMM-mapped 2 MB granules
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_pattern(double2* __restrict__ buf, size_t ch16, size_t nb, int steps, int read) {
  const size_t g = blockIdx.x;
  double2 acc = {0.0, 0.0};
  for (int s = 0; s < steps; s++) {
    double2* p = buf + ((size_t)s * nb + g) * ch16;
    for (size_t i = threadIdx.x; i < ch16; i += blockDim.x) {
      if (read) { double2 v = p[i]; acc.x += v.x; acc.y += v.y; }
      else p[i] = double2{(double)s, (double)i};
    }
  }
  if (read && acc.x == 12345.678) buf[0] = acc;
}
__global__ void k_stream(double2* __restrict__ buf, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) buf[i] = double2{1.0, 2.0};
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 64.0;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  const size_t CH = 57600, ch16 = CH / 16, nb = 3600;
  const int steps = (int)(gb * 1e9 / (double)(nb * CH));
  const size_t bytes = (size_t)steps * nb * CH;
  double2* buf = nullptr;
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  if (mode == 0) {
    CK(hipMalloc(&buf, bytes));
  } else {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t piece = ((size_t)(mode == 1 ? 1 : 16) << 30) / gran * gran;  // 1 GiB or 16 GiB physical pieces
    const size_t total = (bytes + piece - 1) / piece * piece;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, gran, nullptr, 0));
    for (size_t off = 0; off < total; off += piece) {
      hipMemGenericAllocationHandle_t hnd;
      CK(hipMemCreate(&hnd, piece, &prop, 0));
      CK(hipMemMap((char*)va + off, piece, 0, hnd, 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    buf = (double2*)va;
    printf("VMM: granularity %zu, pieces of %zu MiB\n", gran, piece >> 20);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* what, auto launch, double nbytes) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, 0);
      launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-44s rep %d: %8.1f ms  %7.1f GB/s\n", what, rep, ms, nbytes / ms / 1e6);
    }
    return 0;
  };
  printf("allocation %.1f GB (free %.1f of %.1f GB before), steps %d, mode %d\n", bytes / 1e9, fr / 1e9, tot / 1e9, steps, mode);
  timeit("stream fill, 4096 x 256 threads", [&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, buf, bytes / 16); }, (double)bytes);
  timeit("sweep pattern WRITE, 3600 wg x 256 thr", [&] { hipLaunchKernelGGL(k_pattern, dim3(nb), dim3(256), 0, 0, buf, ch16, nb, steps, 0); }, (double)bytes);
  timeit("sweep pattern READ,  3600 wg x 256 thr", [&] { hipLaunchKernelGGL(k_pattern, dim3(nb), dim3(256), 0, 0, buf, ch16, nb, steps, 1); }, (double)bytes);
  // the sweep's occupancy: one workgroup per CU at a time
  timeit("sweep pattern WRITE, 256 wg (one round)", [&] { hipLaunchKernelGGL(k_pattern, dim3(256), dim3(256), 0, 0, buf, ch16, nb, steps, 0); }, (double)steps * 256 * CH);
  return 0;
}
