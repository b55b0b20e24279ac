// qd_inst.hip — one translation unit per (number of oscillators, Schroedinger/Lindblad): compiled
// with -DQD_Q=<1..5> -DQD_L=<0|1>.  Instantiates the persistent sweep kernels for every supported
// elements-per-thread count and exports plain launch functions for the dispatcher in qd_kernels.hip.
#include "qd_device.h"

#ifndef QD_Q
#error "compile with -DQD_Q=<1..5> -DQD_L=<0|1>"
#endif

namespace qd {

#define QD_CAT3(a, b, c) a##b##_##c
#define QD_NAME(base, q, l) QD_CAT3(base, q, l)

template <typename K>
static hipError_t set_lds(K kern, size_t bytes) {
  if (bytes > 48 * 1024)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return hipSuccess;
}

#define QD_EPT_SWITCH(KERNEL, ...)                                                   \
  switch (cfg.ept) {                                                                 \
    case 1: { auto kf = KERNEL<QD_Q, (QD_L != 0), 1>; __VA_ARGS__; } break;          \
    case 2: { auto kf = KERNEL<QD_Q, (QD_L != 0), 2>; __VA_ARGS__; } break;          \
    case 4: { auto kf = KERNEL<QD_Q, (QD_L != 0), 4>; __VA_ARGS__; } break;          \
    case 8: { auto kf = KERNEL<QD_Q, (QD_L != 0), 8>; __VA_ARGS__; } break;          \
    default: return hipErrorInvalidValue;                                            \
  }

hipError_t QD_NAME(inst_forward_, QD_Q, QD_L)(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  QD_EPT_SWITCH(k_forward, {
    hipError_t e = set_lds(kf, cfg.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3(a.nb), dim3(cfg.block), cfg.lds, st, a);
  })
  return hipGetLastError();
}

hipError_t QD_NAME(inst_adjoint_, QD_Q, QD_L)(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  QD_EPT_SWITCH(k_adjoint, {
    hipError_t e = set_lds(kf, cfg.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3(a.nb), dim3(cfg.block), cfg.lds, st, a);
  })
  return hipGetLastError();
}

hipError_t QD_NAME(inst_apply_, QD_Q, QD_L)(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb,
                                            const LaunchCfg& cfg, hipStream_t st) {
  QD_EPT_SWITCH(k_apply, {
    hipError_t e = set_lds(kf, cfg.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3(nb), dim3(cfg.block), cfg.lds, st, S, ctlrow, transpose, x, y);
  })
  return hipGetLastError();
}

}  // namespace qd
