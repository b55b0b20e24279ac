// qd_inst.hip — one translation unit per (number of oscillators, Schroedinger/Lindblad, general/qubit
// stencil): compiled with -DQD_Q=<1..5> -DQD_L=<0|1> -DQD_B=<0|1>.  Instantiates the persistent sweep
// kernels for the kernel variants that make sense for that case and exports plain launch functions
// for the dispatcher in qd_kernels.hip.
#include "qd_device.h"
#include "qd_big.h"

#if !defined(QD_Q) || !defined(QD_L) || !defined(QD_B) || !defined(QD_PART)
#error "compile with -DQD_Q=<1..8> -DQD_L=<0|1> -DQD_B=<0 general|1 qubit|2 dense> -DQD_PART=<0 forward + apply|1 adjoint|2 forward, GMRES kernels|3 adjoint, GMRES kernels>"
#endif
// (four objects per case: forward / adjoint x Neumann / GMRES kernels compile side by side - the five-oscillator Lindblad
// case alone took 13 minutes as one translation unit)
constexpr bool kGmPart = (QD_PART >= 2);

namespace qd {

#define QD_CAT4(a, q, l, b) a##q##_##l##_##b
#define QD_NAME(base, q, l, b) QD_CAT4(base, q, l, b)

constexpr bool kQubit = (QD_B == 1);
constexpr bool kDense = (QD_B == 2);  // user-supplied dense Hamiltonians (DenseStencil)
constexpr bool kLind = (QD_L != 0);
// all-qubit systems have a fixed dimension, so only the matching variants are built
constexpr int kQubitDim = kLind ? (1 << (2 * QD_Q)) : (1 << QD_Q);

template <int VAR>
constexpr bool variant_built() {
  // Built = what pick_config() can select.  Measured on MI355X and therefore NOT built: V6/V7 (two initial
  // conditions interleaved per workgroup: higher per-wave throughput, but the batches of this problem
  // class never exceed the number of SIMDs, so one initial condition per wave wins, C2 38.6M vs 26.7M
  // units/s); V3 and V5 (1024-thread blocks: 128 VGPRs are not enough, 8-12x slower than V2 on C5); V8 and
  // V10 (column layout with 4 / 6 columns per wave: 4.0M vs 4.85M units/s of V9 on C4).
  // ... and only where the variant's size class is reachable with this many oscillators (every oscillator has at
  // least two levels: dim >= 2^Q, or 4^Q for Lindblad) - the five-oscillator Lindblad unit alone took 6 minutes
  constexpr long kMinDim = kLind ? (1L << (2 * QD_Q)) : (1L << QD_Q);
  constexpr bool fits0 = kMinDim <= 64, fits1 = kMinDim <= 256, fits2 = kMinDim <= 1024;
  // six to eight oscillators, Lindblad (beyond the reference's matrix-free templates, mastereq.cpp:2977-3239): the eight-elements-per-thread
  // LDS kernel for 2^6 (dim 4096, the only such system that fits a CU) and the global-memory kernels
  if (kLind && QD_Q >= 6 && !kDense) return (VAR == 4 && QD_Q == 6) || VAR == 16;
  if (kDense && QD_Q >= 6) return VAR == 16 || (!kLind && ((VAR == 11 && fits0) || (VAR == 12 && fits1) || (VAR == 13 && fits2)));
  if (kDense) return (VAR == 11 && fits0) || (VAR == 12 && fits1) || (VAR == 13 && fits2) || (kLind && VAR == 15 && QD_Q <= 4) || (kLind && VAR == 17 && QD_Q <= 5) || VAR == 16;
  if (!kQubit) return (VAR == 0 && fits0) || (VAR == 1 && fits1) || (VAR == 2 && fits2) || VAR == 4 || (kLind && (VAR == 9 || VAR == 14)) || VAR == 16;
  if (kQubitDim <= 64) return VAR == 0;
  if (kQubitDim <= 256) return VAR == 1;
  return VAR == 2;
}

template <typename K>
static hipError_t set_lds(K kern, size_t bytes) {
  if (bytes > 48 * 1024)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return hipSuccess;
}

// Sweeps of the global-memory kernels (qd_big.h).  One workgroup per initial condition: a plain launch.  Teams: the grid is
// padded to whole rounds of the 8 XCDs (the members of a team share an XCD), the team barriers spin, so the launch is cooperative -
// the runtime refuses a grid that cannot be resident instead of letting it hang.
[[maybe_unused]] static hipError_t launch_big(const void* kern, const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  SweepArgs b = a;
  b.S.team = cfg.team > 1 ? cfg.team : 1;
  b.S.team_spread = (cfg.spread ? 1 : 0) | (cfg.blocked == 1 ? 2 : cfg.blocked == 2 ? 4 : 0);
  void* args[] = {&b};
  if (b.S.team == 1) return hipLaunchKernel(kern, dim3(a.nb), dim3(cfg.block), args, cfg.lds, st);
  hipError_t e = hipMemsetAsync(b.S.tbar, 0, sizeof(unsigned long long) * BIG_BAR_STRIDE * (size_t)a.nb, st);
  if (e != hipSuccess) return e;
  const int teams = cfg.spread ? a.nb : (a.nb + 7) / 8 * 8;
  return hipLaunchCooperativeKernel(kern, dim3(teams * b.S.team), dim3(cfg.block), args, (unsigned)cfg.lds, st);
}

#if QD_PART == 0 || QD_PART == 2
template <int VAR>
static hipError_t go_forward(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  if constexpr (VAR == 16 && variant_built<VAR>()) {
    return launch_big(reinterpret_cast<const void*>(k_forward_big<QD_Q, kLind, kDense, kGmPart>), a, cfg, st);
  } else if constexpr (VAR != 16 && variant_built<VAR>()) {
    auto kf = k_forward<QD_Q, kLind, VAR, kQubit, kGmPart>;
    if constexpr ((VAR == 0 || VAR == 1) && !kGmPart) {
      if (plain_sweep(a, cfg, 0)) kf = k_forward<QD_Q, kLind, VAR, kQubit, kGmPart, true>;
    }
    hipError_t e = set_lds(kf, cfg.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3((a.nb + Variant<VAR>::ICPB - 1) / Variant<VAR>::ICPB), dim3(cfg.block), cfg.lds, st, a);
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}
#endif
#if QD_PART == 0
template <int VAR>
static hipError_t go_apply(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb,
                           const LaunchCfg& cfg, hipStream_t st) {
  if constexpr (VAR == 16 && variant_built<VAR>()) {
    DevSys S1 = S;  // one operator application: no exchange between workgroups after the load, no team needed
    S1.team = 1;
    hipLaunchKernelGGL((k_apply_big<QD_Q, kLind, kDense>), dim3(nb), dim3(cfg.block), cfg.lds, st, S1, ctlrow, transpose, x, y, nb);
    return hipGetLastError();
  } else if constexpr (variant_built<VAR>()) {
    auto kf = k_apply<QD_Q, kLind, VAR, kQubit>;
    hipError_t e = set_lds(kf, cfg.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3((nb + Variant<VAR>::ICPB - 1) / Variant<VAR>::ICPB), dim3(cfg.block), cfg.lds, st, S, ctlrow, transpose, x, y, nb);
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}

#endif
#if QD_PART == 1 || QD_PART == 3
template <int VAR>
static hipError_t go_adjoint(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  if constexpr (VAR == 16 && variant_built<VAR>()) {
    if constexpr (!kGmPart) {
      if (a.stepper_ee) return launch_big(reinterpret_cast<const void*>(k_adjoint_big<QD_Q, kLind, kDense, false, true>), a, cfg, st);
    }
    return launch_big(reinterpret_cast<const void*>(k_adjoint_big<QD_Q, kLind, kDense, kGmPart, false>), a, cfg, st);
  } else if constexpr (VAR != 16 && variant_built<VAR>()) {
    auto kf = k_adjoint<QD_Q, kLind, VAR, kQubit, kGmPart>;
    if constexpr ((VAR == 0 || VAR == 1) && !kGmPart) {
      if (plain_sweep(a, cfg, 1)) kf = k_adjoint<QD_Q, kLind, VAR, kQubit, kGmPart, true>;
    }
    hipError_t e = set_lds(kf, cfg.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3((a.nb + Variant<VAR>::ICPB - 1) / Variant<VAR>::ICPB), dim3(cfg.block), cfg.lds, st, a);
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}
#endif

#define QD_VAR_SWITCH(FN, ...)               \
  switch (cfg.var) {                         \
    case 0: return FN<0>(__VA_ARGS__);       \
    case 1: return FN<1>(__VA_ARGS__);       \
    case 2: return FN<2>(__VA_ARGS__);       \
    case 3: return FN<3>(__VA_ARGS__);       \
    case 4: return FN<4>(__VA_ARGS__);       \
    case 5: return FN<5>(__VA_ARGS__);       \
    case 6: return FN<6>(__VA_ARGS__);       \
    case 7: return FN<7>(__VA_ARGS__);       \
    case 8: return FN<8>(__VA_ARGS__);       \
    case 9: return FN<9>(__VA_ARGS__);       \
    case 10: return FN<10>(__VA_ARGS__);     \
    case 11: return FN<11>(__VA_ARGS__);     \
    case 12: return FN<12>(__VA_ARGS__);     \
    case 13: return FN<13>(__VA_ARGS__);     \
    case 14: return FN<14>(__VA_ARGS__);     \
    case 15: return FN<15>(__VA_ARGS__);     \
    case 17: return FN<17>(__VA_ARGS__);     \
    case 16: return FN<16>(__VA_ARGS__);     \
    default: return hipErrorInvalidValue;    \
  }

#if QD_PART == 0
hipError_t QD_NAME(inst_forwardgm_, QD_Q, QD_L, QD_B)(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st);
hipError_t QD_NAME(inst_forward_, QD_Q, QD_L, QD_B)(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  if (cfg.gmres) return QD_NAME(inst_forwardgm_, QD_Q, QD_L, QD_B)(a, cfg, st);  // GMRES kernels: another object
  QD_VAR_SWITCH(go_forward, a, cfg, st)
}
hipError_t QD_NAME(inst_apply_, QD_Q, QD_L, QD_B)(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y,
                                                  int nb, const LaunchCfg& cfg, hipStream_t st) {
  QD_VAR_SWITCH(go_apply, S, ctlrow, transpose, x, y, nb, cfg, st)
}
#elif QD_PART == 1
hipError_t QD_NAME(inst_adjointgm_, QD_Q, QD_L, QD_B)(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st);
hipError_t QD_NAME(inst_adjoint_, QD_Q, QD_L, QD_B)(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  if (cfg.gmres && !(cfg.var == 16 && a.stepper_ee)) return QD_NAME(inst_adjointgm_, QD_Q, QD_L, QD_B)(a, cfg, st);  // (explicit Euler has no linear solve)
  QD_VAR_SWITCH(go_adjoint, a, cfg, st)
}
#elif QD_PART == 2
hipError_t QD_NAME(inst_forwardgm_, QD_Q, QD_L, QD_B)(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  QD_VAR_SWITCH(go_forward, a, cfg, st)
}
#else
hipError_t QD_NAME(inst_adjointgm_, QD_Q, QD_L, QD_B)(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  QD_VAR_SWITCH(go_adjoint, a, cfg, st)
}
#endif
#if QD_B == 0 && QD_PART == 0
hipError_t QD_NAME(inst_bigtable_, QD_Q, QD_L, QD_B)(const DevSys& S, double* ecoef, unsigned* edig, hipStream_t st) {
  hipLaunchKernelGGL((k_big_table<QD_Q, kLind>), dim3((S.dim + 255) / 256), dim3(256), 0, st, S, reinterpret_cast<double2*>(ecoef),
                     reinterpret_cast<uint2*>(edig));
  return hipGetLastError();
}
#endif

}  // namespace qd
