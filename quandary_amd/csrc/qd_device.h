// qd_device.h — device templates of the persistent sweep kernels (included by qd_inst.hip and
// qd_kernels.hip).  gfx950 / CDNA4 only.
//
// One workgroup owns one initial condition for the WHOLE time loop.  Each thread owns EPT elements of
// the vectorised state in registers; the vector that is being stencil-read lives in LDS as
// interleaved complex numbers (one ds_read_b128 per neighbour).  Two stencil implementations share
// the sweep skeleton:
//   * general: runtime level counts, branch-free — validity of a neighbour is folded into
//     per-oscillator coefficient tables in LDS (zero where the reference's `if` fails) and the
//     neighbour index is clamped into the vector;
//   * qubit (all n_k == 2): digits are bits of the storage index, neighbours are `it ^ bit`, every
//     ladder coefficient is +-1, no tables at all.
// Kernel variants (VAR) fix elements/thread, the maximal block size (= register budget through
// __launch_bounds__), LDS double buffering and whether the block is a single wave (no barriers).
//
// Reference semantics restated here (paths relative to the reference repository):
//   stencil            include/mastereq.hpp:316-912, src/mastereq.cpp:1464-1709
//   gradient coeffs    include/mastereq.hpp:553-604, src/mastereq.cpp:970-1276
//   IMR fwd / bwd      src/timestepper.cpp:584-694, Neumann :697-727
//   time loops         src/timestepper.cpp:96-253, penalties :256-480
//   objective / seeds  src/optimtarget.cpp:343-447, :712-897
#pragma once
#include <hip/hip_runtime.h>

#include "qd_internal.h"

// Measurement builds only (profiles/ablate.sh): QD_ABLATE removes phases of the small-system solver iteration so that their cost can
// be read off timing differences - bit 0: no stopping test (four iterations per solve, no norm reduction), bit 1: no neighbour reads
// (the element itself stands in for its neighbours), bit 2: no stencil arithmetic (a sum of the neighbours instead).  The results of
// such builds are meaningless; the product is built with QD_ABLATE = 0.
#ifndef QD_ABLATE
#define QD_ABLATE 0
#endif

namespace qd {

// ---------------------------------------------------------------------------------------------
// kernel variants
// ---------------------------------------------------------------------------------------------
// EPT    elements (slots) per thread          MAXB   largest block (= register budget via __launch_bounds__)
// DBUF   two LDS copies of the exchange vector (one barrier per solver iteration instead of two)
// ICPB   initial conditions interleaved in one workgroup
// COL    column-per-wave layout (ColStencil)
// LEAN   throughput regime: no register-carried prefetches, the vectors that are idle during a linear
//        solve are parked in L2/HBM explicitly (SweepArgs::stash) instead of being spilled by the compiler
// DENSE  user-supplied dense Hamiltonians (DenseStencil) instead of the matrix-free stencil
// PACKED several columns per wave in the column layout (N <= 32)
template <int EPT_, int MAXB_, bool DBUF_, bool ONEWAVE_, int ICPB_ = 1, bool COL_ = false, bool LEAN_ = false, bool DENSE_ = false,
          bool PACKED_ = false, bool MFMA_ = false>
struct VariantDef {
  static constexpr int EPT = EPT_, MAXB = MAXB_, ICPB = ICPB_, FENCE = 1;
  static constexpr bool DBUF = DBUF_, ONEWAVE = ONEWAVE_, BLDS = false, COL = COL_, LEAN = LEAN_, DENSE = DENSE_, PACKED = PACKED_;
  static constexpr bool MFMA = MFMA_;  // dense operator on the matrix cores (v_mfma_f64_16x16x4_f64)
};
template <int VAR> struct Variant;
template <> struct Variant<0> : VariantDef<1, 64, false, true> {};     // dim <= 64: one wave, no barriers
template <> struct Variant<1> : VariantDef<1, 256, true, false> {};    // dim <= 256
template <> struct Variant<2> : VariantDef<4, 256, true, false> {};    // dim <= 1024, many initial conditions
template <> struct Variant<3> : VariantDef<4, 1024, false, false, 1, false, true> {};
template <> struct Variant<4> : VariantDef<8, 512, false, false, 1, false, true> {};  // dim <= 4096 (Schroedinger, or N > 64)
template <> struct Variant<5> : VariantDef<1, 1024, true, false> {};   // dim <= 1024, few initial conditions
// V6: two initial conditions interleaved in ONE wave (dim <= 64): two independent dependency chains per
// lane hide the LDS / fp64 latencies that bound the single-wave kernels
template <> struct Variant<6> : VariantDef<2, 64, false, true, 2> {};
// V7: the same for dim <= 256 (four waves, two initial conditions per workgroup, one barrier serves both)
template <> struct Variant<7> : VariantDef<2, 256, true, false, 2> {};
// V8/V9/V10: column-per-wave layout for large density matrices (N <= 64, dim up to 4096): lane = row of
// rho, every wave owns EPT whole columns -> all ket-side quantities are wave-uniform, all bra-side
// quantities are loop invariants of the thread (ColStencil below)
template <> struct Variant<8> : VariantDef<4, 1024, true, false, 1, true, true> {};
template <> struct Variant<9> : VariantDef<8, 512, true, false, 1, true, true> {};
template <> struct Variant<10> : VariantDef<6, 640, true, false, 1, true, true> {};
// V11-V13: the linear-map variants V0-V2 with the dense user-Hamiltonian operator (hamiltonian_file_Hsys / _Hc)
template <> struct Variant<11> : VariantDef<1, 64, false, true, 1, false, false, true> {};
template <> struct Variant<12> : VariantDef<1, 256, true, false, 1, false, false, true> {};
template <> struct Variant<13> : VariantDef<4, 256, true, false, 1, false, false, true> {};
// V14: packed column layout for 17 <= N <= 32 (256 < dim <= 1024, non-qubit Lindblad): floor(64/N) columns per wave slot
template <> struct Variant<14> : VariantDef<4, 256, true, false, 1, true, true, false, true> {};
// V15: dense operator of a 16 x 16 density matrix (dim 256) as complex 16x16x16 products on the fp64 matrix cores:
// one wave per initial condition, the state lives in the MFMA accumulator layout
template <> struct Variant<15> : VariantDef<4, 64, false, true, 1, false, false, true, false, true> {};
// V16: states beyond one CU's LDS (dim > 4096): work vectors in global memory, exchanged through L2 (qd_big.h)
template <> struct Variant<16> : VariantDef<1, 1024, false, false> {};
// V17: dense operator of a 32 x 32 density matrix (dim 1024) on the fp64 matrix cores: four waves, one 16 x 16 tile of rho each
// (PACKED marks the two-tiles-per-dimension form of the matrix-core stencil)
template <> struct Variant<17> : VariantDef<4, 256, true, false, 1, false, false, true, true, true> {};
constexpr int NVARIANTS = 18;
// BLDS: the right-hand side of the linear solve is parked in a second LDS vector instead of registers
// (large elements-per-thread variants would otherwise spill)

constexpr int NRED = 16;  // max values reduced at once (2*QD_MAX_OSC)
constexpr int GMRES_MR = 10;    // restart length of the GMRES whose Krylov basis lives in LDS
constexpr int GMRES_MR_G = 30;  // restart length with the basis in global memory (= PETSc's default KSPGMRES restart)
constexpr int gmres_nsc(int mr) { return (mr + 2) + 2 * mr + (mr + 2) + mr * mr + mr; }  // scalars of the Hessenberg problem
constexpr int GMRES_NSC = gmres_nsc(GMRES_MR);

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int l2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const int h2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(h2, l2);
}

// fp32 sum over the 64 lanes (DPP adds, result wave-uniform).  Used ONLY for the stopping test of the
// linear solver, where the squared update norm is compared with a threshold.
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}

// Sum over the 64 lanes of a wave; the result is wave-uniform (same bits in every lane).
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x128>(v);  // row_ror:8  -> every lane holds a total of its row of 16
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}

// Sums over the 64 lanes of a wave of NV per-lane values of which each total is needed in ONE lane only (the 2Q gradient coefficients of
// an adjoint step are stored, not consumed): a reduce-scatter.  gfx950's row swaps make the two upper levels a butterfly without selects -
// v_permlane32_swap(a, b) leaves a' = [a.lo, b.lo], b' = [a.hi, b.hi], so a' + b' is the total over both halves of a in the lower 32 lanes and
// of b in the upper 32; v_permlane16_swap does the same between the even and the odd rows of 16 - and halve the number of values each
// (3 instructions per remaining value and level); the four levels inside a row are DPP adds on the K = ceil(ceil(NV / 2) / 2) values
// that are left.  NV = 8: 42 vector instructions where eight wave_sum()s are ~250.  Afterwards every lane of row r = lane >> 4 holds in
// out[m] the total of v[wave_scatter_index<NV>(r, m)] (-1: padding).
typedef unsigned int qd_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double swap_add32(double a, double b) {
  const qd_u2 l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const qd_u2 h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
}
__device__ __forceinline__ double swap_add16(double a, double b) {
  const qd_u2 l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const qd_u2 h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
}
template <int NV>
__device__ __forceinline__ constexpr int wave_scatter_index(int r, int m) {
  constexpr int H = (NV + 1) / 2, K = (H + 1) / 2;
  const int j = (r & 1) * K + m, g = (r >> 1) * H + j;
  return (j < H && g < NV) ? g : -1;
}
template <int NV>
__device__ __forceinline__ void wave_reduce_scatter(const double (&v)[NV], double (&out)[((NV + 1) / 2 + 1) / 2]) {
  constexpr int H = (NV + 1) / 2, K = (H + 1) / 2;
  double h[H];
#pragma unroll
  for (int i = 0; i < H; i++) h[i] = swap_add32(v[i], H + i < NV ? v[H + i < NV ? H + i : 0] : 0.0);
#pragma unroll
  for (int m = 0; m < K; m++) {
    double o = swap_add16(h[m], K + m < H ? h[K + m < H ? K + m : 0] : 0.0);
    o += dpp_mov<0xB1>(o);   // quad_perm [1,0,3,2]
    o += dpp_mov<0x4E>(o);   // quad_perm [2,3,0,1]
    o += dpp_mov<0x124>(o);  // row_ror:4
    o += dpp_mov<0x128>(o);  // row_ror:8
    out[m] = o;
  }
}

// 1 / sqrt(x), x > 0, to fp64 round-off without fp64 sqrt or division (~40 dependent instructions each): v_rsq_f64 estimate and two
// Newton steps.  The Hessenberg scalars of the in-kernel GMRES are computed redundantly by every lane on uniform values: their
// dependent latency is paid in full by the latency-bound small systems.
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

// value of the next (UP = true) / previous lane of the wave; lanes without a source get 0
template <bool UP>
__device__ __forceinline__ double lane_shift(double v) {
  constexpr int CTRL = UP ? 0x130 /* wave_shl:1 */ : 0x138 /* wave_shr:1 */;
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <bool UP>
__device__ __forceinline__ double2 lane_shift(double2 v) {
  return make_double2(lane_shift<UP>(v.x), lane_shift<UP>(v.y));
}

// Workgroup-wide synchronisation of the LDS exchange vector.  A single-wave workgroup needs no s_barrier
// (its LDS operations complete in order), but the COMPILER still needs the fence: lanes communicate through
// LDS, and without it a load of element it ^ m may legally be hoisted above this lane's store of element
// `it` (different addresses from one thread's point of view) - observed as loop-invariant neighbour values
// in the Neumann loop of the 2x2 Schroedinger system.
template <bool ONEWAVE>
__device__ __forceinline__ void team_sync() {
  if (!ONEWAVE) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Block-wide sum of NV values; every thread returns the same bits, so decisions taken on the result
// are uniform.  Multi-wave blocks: contains ONE __syncthreads(); `red` has two slots used
// alternately by the caller so that no second barrier is needed between consecutive reductions.
template <int NV, bool ONEWAVE>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red) {
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = wave_sum(v[i]);
  if (ONEWAVE) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; i++) red[i * nw + wave] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double s = 0.0;
    for (int w = 0; w < nw; w++) s += red[i * nw + w];
    v[i] = s;
  }
}

// Block-wide fp32 sum for the solver's stopping test (same barrier structure as block_sum).
template <bool ONEWAVE>
__device__ __forceinline__ float block_sum_f32(float v, double* red) {
  v = wave_sum_f32(v);
  if (ONEWAVE) return v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  float* rf = reinterpret_cast<float*>(red);
  if (lane == 0) rf[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < nw; w++) s += rf[w];
  return s;
}

template <int NV, bool ONEWAVE>
__device__ __forceinline__ void block_sum_f32v(float (&v)[NV], double* red) {
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = wave_sum_f32(v[i]);
  if (ONEWAVE) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  float* rf = reinterpret_cast<float*>(red);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; i++) rf[i * nw + wave] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; i++) {
    float s = 0.f;
    for (int w = 0; w < nw; w++) s += rf[i * nw + w];
    v[i] = s;
  }
}

// Block-wide sums of NV values that are only STORED: *dst(i) = total of v[i] (dst(i) may return null: nothing stored), in two halves.
// block_sum_post: the wave level (wave_reduce_scatter); a one-wave block stores at once, a multi-wave block leaves one partial sum per
// wave and value in `red` (one of the caller's two alternating slots).  block_sum_collect, AFTER a barrier of the caller (the one that
// publishes the next exchange vector: the reduction has no barrier of its own): NV threads add the partial sums in wave order and store.
template <int NV, bool ONEWAVE, typename F>
__device__ __forceinline__ void block_sum_post(const double (&v)[NV], double* red, F&& dst) {
  constexpr int K = ((NV + 1) / 2 + 1) / 2;
  double o[K];
  wave_reduce_scatter<NV>(v, o);
  const int lane = threadIdx.x & 63, r = lane >> 4;
  if ((lane & 15) != 0) return;
  const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int m = 0; m < K; m++) {
    const int g = wave_scatter_index<NV>(r, m);
    if (g < 0) continue;
    if (ONEWAVE) {
      if (double* d = dst(g)) *d = o[m];
    } else {
      red[g * nw + wave] = o[m];
    }
  }
}
template <int NV, bool ONEWAVE, typename F>
__device__ __forceinline__ void block_sum_collect(const double* red, F&& dst) {
  if (ONEWAVE || (int)threadIdx.x >= NV) return;
  const int nw = (blockDim.x + 63) >> 6;
  double t = 0.0;
  for (int w = 0; w < nw; w++) t += red[threadIdx.x * nw + w];
  if (double* d = dst((int)threadIdx.x)) *d = t;
}

// ---------------------------------------------------------------------------------------------
// step controls (wave-uniform, streamed from the table by scalar loads, prefetched one step ahead)
// ---------------------------------------------------------------------------------------------
template <int Q>
struct StepC {
  double h, p[Q], q[Q];
  double cs[Q * (Q - 1) / 2 + 1], sn[Q * (Q - 1) / 2 + 1];
  const double2* g;  // user-Hamiltonian path: G(t) = -i H(t) of this sub-step, N x N row-major (null otherwise)
};

template <int Q>
__device__ __forceinline__ void load_step(const double* __restrict__ row, StepC<Q>& c, bool with_pairs) {
  constexpr int NP = Q * (Q - 1) / 2;
  c.h = row[0];
#pragma unroll
  for (int k = 0; k < Q; k++) {
    c.p[k] = row[2 + k];
    c.q[k] = row[2 + Q + k];
  }
  if (with_pairs) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      c.cs[k] = row[2 + 2 * Q + k];
      c.sn[k] = row[2 + 2 * Q + NP + k];
    }
  }
}
// The same through the SCALAR memory path (constant address space: s_load into scalar registers; the tables were written by kernels that
// completed before the sweep was launched) - for the many-wave kernels (qd_col.hip, qd_q32.hip, qd_big.h): no vector registers, no
// v_readfirstlane, and no coupling with the vector-memory counter, which retires in order - a vector load with immediate use also waits
// for every store issued before it (the stage stores of the previous step in a gradient evaluation).  NOT for the one-wave kernels of
// this file: scalar loads share lgkmcnt with LDS and return out of order, so every LDS wait of the solver loop behind a prefetched row
// becomes lgkmcnt(0) and exposes the scalar load (one-lease A/B: 2^4 Schroedinger gradient 2.40 -> 3.52 ms, 2^4 Lindblad forward 2.68 ->
// 3.41; 2^5 gradient 36.7 -> 36.2, 2^20-element state 19.9 -> 19.2).
typedef const __attribute__((address_space(4))) double* qd_kptr;
__device__ __forceinline__ double kload(const double* p) { return *(qd_kptr)p; }
template <int Q>
__device__ __forceinline__ void load_step_k(const double* row, StepC<Q>& c, bool with_pairs) {
  constexpr int NP = Q * (Q - 1) / 2;
  qd_kptr r = (qd_kptr)row;
  c.h = r[0];
#pragma unroll
  for (int k = 0; k < Q; k++) {
    c.p[k] = r[2 + k];
    c.q[k] = r[2 + Q + k];
  }
  if (with_pairs) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      c.cs[k] = r[2 + 2 * Q + k];
      c.sn[k] = r[2 + 2 * Q + NP + k];
    }
  }
}

// Moves a wave-uniform double from vector to scalar registers.
__device__ __forceinline__ double to_scalar(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int Q>
__device__ __forceinline__ void scalarize(StepC<Q>& c, bool with_pairs) {
  constexpr int NP = Q * (Q - 1) / 2;
  c.h = to_scalar(c.h);
#pragma unroll
  for (int k = 0; k < Q; k++) {
    c.p[k] = to_scalar(c.p[k]);
    c.q[k] = to_scalar(c.q[k]);
  }
  if (with_pairs) {
#pragma unroll
    for (int k = 0; k < NP; k++) {
      c.cs[k] = to_scalar(c.cs[k]);
      c.sn[k] = to_scalar(c.sn[k]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LDS layout
// ---------------------------------------------------------------------------------------------
struct Lds {
  double2* buf0;    // state exchange vector(s): buffer b starts at buf0 + b * bstride
  int bstride;      // dim with double buffering, 0 without (both "buffers" coincide)
  double* tup;      // general stencil: tup[ofs_k + a] = (a < n_k-1) ? sqrt(a+1) : 0
  double* tdn;      //                  tdn[ofs_k + a] = sqrt(a)
  double* red;      // reduction scratch, two slots of NRED * nwaves
  double2* bvec;    // BLDS variants: right-hand side of the linear solve
  double2* gmat;    // dense operator: G(t) = -i H(t) of the current sub-step, N x N row-major
  double2* coltab;  // column stencil: coltab[c * Q + k] = (sqrt(i'_k + 1) or 0 at the top level, sqrt(i'_k)) of column c
  double2* kry;     // GMRES: Krylov basis, (GMRES_MR + 1) vectors of dim
  double* ksc;      // GMRES: wave-uniform scalars (Hessenberg column, rotations, rhs, R, solution)
};
__host__ __device__ inline int table_len(const DevSys& S) {
  int t = 0;
  for (int k = 0; k < S.Q; k++) t += S.n[k];
  return (t + 1) & ~1;
}
// krylov: 0 = Neumann only, 1 = GMRES with the Krylov basis in LDS, 2 = GMRES with the basis in global
// memory (only the small Hessenberg problem lives in LDS)
__device__ __forceinline__ Lds carve(unsigned char* smem, const DevSys& S, bool dbuf, bool blds, int krylov = 0, int icpb = 1,
                                     bool col = false, bool dense = false) {
  Lds l;
  l.buf0 = reinterpret_cast<double2*>(smem);
  l.bstride = dbuf ? S.dim * icpb : 0;
  const int tl = table_len(S);
  const int nvec = ((dbuf ? 2 : 1) + (blds ? 1 : 0)) * icpb;
  l.bvec = l.buf0 + (dbuf ? 2 : 1) * (size_t)S.dim * icpb;
  l.tup = reinterpret_cast<double*>(l.buf0 + nvec * (size_t)S.dim);
  l.tdn = l.tup + tl;
  l.red = l.tdn + tl;
  double* p = l.red + 2 * NRED * ((blockDim.x + 63) >> 6);
  l.kry = nullptr;
  l.ksc = nullptr;
  l.coltab = nullptr;
  if (col) {
    l.coltab = reinterpret_cast<double2*>(p);
    p += 2 * (size_t)S.N * S.Q;
  }
  if (krylov == 1) {
    l.kry = reinterpret_cast<double2*>(p);
    p += 2 * (size_t)(GMRES_MR + 1) * S.dim;
  }
  if (krylov) {
    l.ksc = p;
    p += krylov == 1 ? GMRES_NSC : gmres_nsc(GMRES_MR_G);
  }
  l.gmat = nullptr;
  if (dense) l.gmat = reinterpret_cast<double2*>(p + (reinterpret_cast<size_t>(p) & 8 ? 1 : 0));  // 16-byte aligned
  return l;
}
static inline size_t lds_bytes(const DevSys& S, int block, bool dbuf, bool blds, int krylov = 0, int icpb = 1, bool col = false,
                               bool dense = false) {
  return (dense ? sizeof(double2) * (size_t)S.N * S.N + 16 : 0) + (col ? sizeof(double2) * (size_t)S.N * S.Q : 0) + sizeof(double2) * (size_t)S.dim * icpb * ((dbuf ? 2 : 1) + (blds ? 1 : 0)) + sizeof(double) * 2 * (size_t)table_len(S) +
         sizeof(double) * 2 * NRED * (size_t)((block + 63) / 64) +
         (krylov == 1 ? sizeof(double2) * (size_t)(GMRES_MR + 1) * S.dim + sizeof(double) * GMRES_NSC : 0) +
         (krylov == 2 ? sizeof(double) * gmres_nsc(GMRES_MR_G) : 0);
}

// Keeps index arithmetic INSIDE the time loop: without it the compiler hoists every neighbour index,
// digit and coefficient of every owned element out of the loop and spills hundreds of registers.
// Scheduling fence between the elements of a thread (throughput variants): without it the machine
// scheduler hoists the LDS reads of ALL owned elements above the arithmetic of the first one (maximal
// latency hiding for a single wave) and the live ranges no longer fit the register budget.  Several
// waves per SIMD hide the latency instead.
template <int EPT>
__device__ __forceinline__ void slot_fence() {
  if (EPT > 1) {
    asm volatile("" ::: "memory");  // orders the memory operations already at the IR / DAG level
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same with a value pinned at the fence.  A fence orders memory operations only: the arithmetic of a slot may still sink below the
// LDS reads of the following slots (and all their neighbours are live at once); a pinned result completes the slot's arithmetic first.
#ifndef QD_NO_PIN
template <int EPT>
__device__ __forceinline__ void slot_fence_pin(double2& v) {
  if (EPT > 1) {
    asm volatile("" : "+v"(v.x), "+v"(v.y)::"memory");
    __builtin_amdgcn_sched_barrier(0);
  }
}
#else
template <int EPT>
__device__ __forceinline__ void slot_fence_pin(double2&) { slot_fence<EPT>(); }
#endif

template <int EPE> __device__ __forceinline__ int at_use(int v);
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}
// index of an owned element at a point of use: the throughput variants re-derive everything that
// depends on it (targets, weights, addresses) instead of keeping it in registers across the time loop
template <int EPE>
__device__ __forceinline__ int at_use(int v) {
  return EPE > 1 ? opaque(v) : v;
}

// the digits of one index are packed into 32 bits: up to 256 / 64 / 32 / 16 levels for <= 4 / 5 / 6 / 7-8 oscillators
constexpr int packed_digit_bits(int q) { return q <= 4 ? 8 : q == 5 ? 6 : q == 6 ? 5 : 4; }

// ---------------------------------------------------------------------------------------------
// general stencil (runtime level counts)
// ---------------------------------------------------------------------------------------------
template <int Q, bool LIND, int EPT, int EPE = EPT>
struct GenStencil {
  static constexpr bool WHOLE = false;  // apply() works one slot at a time
  static constexpr bool NEEDS_SLOTS = false;  // apply() reads every neighbour from LDS
  static constexpr int DB = packed_digit_bits(Q);  // bits per packed digit
  int it[EPT];         // storage index (clamped to dim-1 for slots beyond the vector)
  bool valid[EPT];
  unsigned dbra[EPT];  // bra digits i_k
  unsigned dket[EPT];  // ket digits i_k' (Lindblad)
  double dw[EPT];      // Delta = h(I) - h(I')   (mastereq.hpp:316-403)
  double dd[EPT];      // d = L2 + L1diag         (mastereq.hpp:339-353, :416-433)
  int ofs[Q];          // start of oscillator k in the coefficient tables
  // latency regime (one element per thread): ladder coefficients of THE element kept in registers instead
  // of four table reads per oscillator and operator application
  // ... also for the four-elements-per-thread Schroedinger kernel (V2 is only used for Schroedinger systems with
  // 256 < dim <= 1024 since the packed column layout took over its Lindblad cases): 4 x 2Q coefficients
  static constexpr bool HOIST = (EPE == 1) || (!LIND && EPE == 4);
  static constexpr int HS = HOIST ? EPE : 1;
  double hsu[HS][HOIST ? Q : 1], hsd[HS][HOIST ? Q : 1], hsup[HS][HOIST && LIND ? Q : 1], hsdp[HS][HOIST && LIND ? Q : 1];

  __device__ __forceinline__ static int dig(unsigned d, int k) { return (int)((d >> (DB * k)) & ((1u << DB) - 1u)); }

  __device__ __forceinline__ void init(const DevSys& S, const Lds& L) {
    int o = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      ofs[k] = o;
      o += S.n[k];
    }
    // coefficient tables: validity of a neighbour is a zero coefficient
    for (int k = 0; k < Q; k++)
      for (int a = threadIdx.x; a < S.n[k]; a += blockDim.x) {
        L.tup[ofs[k] + a] = (a < S.n[k] - 1) ? sqrt((double)(a + 1)) : 0.0;
        L.tdn[ofs[k] + a] = sqrt((double)a);
      }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int raw = (int)threadIdx.x + (j % EPE) * (int)blockDim.x;  // slot j = (initial condition j / EPE, element j % EPE)
      valid[j] = raw < S.dim;
      it[j] = valid[j] ? raw : S.dim - 1;
      const int I = LIND ? it[j] % S.N : it[j];
      const int Ip = LIND ? it[j] / S.N : 0;
      int ia[Q], ipa[Q];
      dbra[j] = 0;
      dket[j] = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        ia[k] = (I / S.post[k]) % S.n[k];
        ipa[k] = LIND ? (Ip / S.post[k]) % S.n[k] : 0;
        dbra[j] |= (unsigned)ia[k] << (DB * k);
        dket[j] |= (unsigned)ipa[k] << (DB * k);
      }
      double hd = 0.0, hdp = 0.0, d = 0.0;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        hd += S.detune[k] * ia[k] - S.xi[k] / 2.0 * ia[k] * (ia[k] - 1);
        if (LIND) {
          hdp += S.detune[k] * ipa[k] - S.xi[k] / 2.0 * ipa[k] * (ipa[k] - 1);
          d += S.g2[k] * (ia[k] * ipa[k] - 0.5 * (ia[k] * ia[k] + ipa[k] * ipa[k])) - S.g1[k] / 2.0 * (ia[k] + ipa[k]);
        }
#pragma unroll
        for (int l = k + 1; l < Q; l++) {
          hd -= S.xikl[pair] * ia[k] * ia[l];
          if (LIND) hdp -= S.xikl[pair] * ipa[k] * ipa[l];
          pair++;
        }
      }
      dw[j] = hd - hdp;
      dd[j] = d;
      if (HOIST && j < EPE) {
#pragma unroll
        for (int k = 0; k < Q; k++) {
          hsu[j % HS][k] = (ia[k] < S.n[k] - 1) ? sqrt((double)(ia[k] + 1)) : 0.0;
          hsd[j % HS][k] = sqrt((double)ia[k]);
          if (LIND) {
            hsup[j % HS][k] = (ipa[k] < S.n[k] - 1) ? sqrt((double)(ipa[k] + 1)) : 0.0;
            hsdp[j % HS][k] = sqrt((double)ipa[k]);
          }
        }
      }
    }
  }

  __device__ __forceinline__ void prep(const DevSys&, const Lds&, const StepC<Q>&) {}

  // Ladder-operator neighbour sums of oscillator k (control(), mastereq.hpp:818-912):
  //   U1 = sqrt(i+1) x(it+post), U2 = sqrt(i'+1) x(it+N post), D1 = sqrt(i) x(it-post), D2 = sqrt(i') x(it-N post)
  //   A = U1 + U2 - D1 - D2,  B = U1 - U2 + D1 - D2
  // so that the control part of y = M x is  y_re += q A_re + p B_im,  y_im += q A_im - p B_re, and
  // dRHSdp_getcoeffs (mastereq.hpp:553-604) is  res_p = (B_im, -B_re), res_q = (A_re, A_im).
  __device__ __forceinline__ void ladder(const DevSys& S, const Lds& L, const double2* __restrict__ sx, int k, int j, double2& A,
                                         double2& B) const {
    const int a = dig(opaque(dbra[j]), k), st = S.post[k], i0 = opaque(it[j]), top = S.dim - 1;
    const double su = L.tup[ofs[k] + a], sd = L.tdn[ofs[k] + a];
    const double2 xu = sx[min(i0 + st, top)], xd = sx[max(i0 - st, 0)];
    double er = su * xu.x, ei = su * xu.y;    // U1
    double fr = -sd * xd.x, fi = -sd * xd.y;  // -D1
    if (LIND) {
      const int ap = dig(opaque(dket[j]), k), stp = S.N * st;
      const double sup = L.tup[ofs[k] + ap], sdp = L.tdn[ofs[k] + ap];
      const double2 xup = sx[min(i0 + stp, top)], xdp = sx[max(i0 - stp, 0)];
      er = fma(-sdp, xdp.x, er);  // U1 - D2
      ei = fma(-sdp, xdp.y, ei);
      fr = fma(sup, xup.x, fr);  // U2 - D1
      fi = fma(sup, xup.y, fi);
    }
    A.x = er + fr;
    A.y = ei + fi;
    B.x = er - fr;
    B.y = ei - fi;
  }

  // y = M x (TRANS = false) or M^T x at element j.  The Hamiltonian part of the real operator is
  // antisymmetric (compare control/control_T, Jkl_coupling/Jkl_coupling_T and the drift signs at
  // mastereq.cpp:1541-1542 vs :1665-1666), the dissipator diagonal is symmetric and the T1
  // off-diagonal term moves to the mirrored neighbour (L1decay / L1decay_T, mastereq.hpp:758-797).
  // All LDS reads of the element are issued before any arithmetic (one latency, not one per oscillator);
  // the reference's thresholds |J| > 1e-10 and |gamma_1| > 1e-12 are applied once on the host
  // (coefficients below them arrive here as exact zeros), so there is no branch per term.
  template <bool TRANS>
  __device__ __forceinline__ double2 apply(const DevSys& S, const Lds& L, const double2* __restrict__ sx, const StepC<Q>& c, int j,
                                           const double2 xs) const {
    const int i0 = HOIST ? it[j] : opaque(it[j]), top = S.dim - 1;
    const unsigned db = HOIST ? dbra[j] : opaque(dbra[j]), dk = HOIST ? dket[j] : opaque(dket[j]);
    double hr = dw[j] * xs.y, hi = -dw[j] * xs.x;
    double l1r = 0.0, l1i = 0.0;  // T1 off-diagonal contribution
    if (HOIST) {
      // all LDS reads of the element first, then the arithmetic, ladder coefficients from registers
      double2 xu[Q], xd[Q], xup[Q], xdp[Q], xl[Q];
      double su[Q], sd[Q], sup[Q], sdp[Q];
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int st = S.post[k];
        su[k] = hsu[j % HS][k];
        sd[k] = hsd[j % HS][k];
        xu[k] = sx[min(i0 + st, top)];
        xd[k] = sx[max(i0 - st, 0)];
        if (LIND) {
          const int stp = S.N * st;
          sup[k] = hsup[j % HS][k];
          sdp[k] = hsdp[j % HS][k];
          xup[k] = sx[min(i0 + stp, top)];
          xdp[k] = sx[max(i0 - stp, 0)];
          xl[k] = sx[TRANS ? max(i0 - st - stp, 0) : min(i0 + st + stp, top)];
        }
      }
#pragma unroll
      for (int k = 0; k < Q; k++) {
        // A = U1 + U2 - D1 - D2,  B = U1 - U2 + D1 - D2  (see ladder())
        double er = su[k] * xu[k].x, ei = su[k] * xu[k].y;
        double fr = -sd[k] * xd[k].x, fi = -sd[k] * xd[k].y;
        if (LIND) {
          er = fma(-sdp[k], xdp[k].x, er);
          ei = fma(-sdp[k], xdp[k].y, ei);
          fr = fma(sup[k], xup[k].x, fr);
          fi = fma(sup[k], xup[k].y, fi);
          const double l1 = S.g1off[k] * (TRANS ? sd[k] * sdp[k] : su[k] * sup[k]);
          l1r = fma(l1, xl[k].x, l1r);
          l1i = fma(l1, xl[k].y, l1i);
        }
        hr = fma(c.q[k], er + fr, fma(c.p[k], ei - fi, hr));
        hi = fma(c.q[k], ei + fi, fma(-c.p[k], er - fr, hi));
      }
    } else {
      // throughput regime (several elements per thread, several waves per SIMD hide the LDS latency):
      // stream oscillator by oscillator.  The wave-uniform branches are deliberate: they bound the
      // scheduling regions, which keeps the live register set small (without them the compiler
      // batches the reads of all owned elements and spills).
#pragma unroll
      for (int k = 0; k < Q; k++) {
        double2 A, B;
        ladder(S, L, sx, k, j, A, B);
        hr = fma(c.q[k], A.x, fma(c.p[k], B.y, hr));
        hi = fma(c.q[k], A.y, fma(-c.p[k], B.x, hi));
      }
      if (LIND) {
#pragma unroll
        for (int k = 0; k < Q; k++) {
          const double g1 = S.g1off[k];
          if (g1 == 0.0) continue;
          const int a = dig(db, k), ap = dig(dk, k), st = S.post[k] * (S.N + 1);
          const double l1 = g1 * (TRANS ? L.tdn[ofs[k] + a] * L.tdn[ofs[k] + ap] : L.tup[ofs[k] + a] * L.tup[ofs[k] + ap]);
          const double2 xn = sx[TRANS ? max(i0 - st, 0) : min(i0 + st, top)];
          l1r = fma(l1, xn.x, l1r);
          l1i = fma(l1, xn.y, l1i);
        }
      }
    }
    // dipole-dipole coupling (Jkl_coupling, mastereq.hpp:632-675):
    //   T1 = sqrt(i_k (i_l+1)) x(it-post_k+post_l), T2 = sqrt(i_l (i_k+1)) x(it+post_k-post_l), T3/T4 ket analogues
    //   y += J [ sin (T1 - T2 + T3 - T4) - i cos (T1 + T2 - T3 - T4) ]
    if (S.hasJ) {
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
#pragma unroll
        for (int l = k + 1; l < Q; l++, pair++) {
          const double Jkl = S.J[pair];
          if (Jkl == 0.0) continue;
          const int a = dig(db, k), b = dig(db, l);
          const int sk = S.post[k], sl = S.post[l];
          const double s1 = L.tdn[ofs[k] + a] * L.tup[ofs[l] + b], s2 = L.tdn[ofs[l] + b] * L.tup[ofs[k] + a];
          const double2 x1 = sx[min(max(i0 - sk + sl, 0), top)], x2 = sx[min(max(i0 + sk - sl, 0), top)];
          double ar = s1 * x1.x - s2 * x2.x, ai = s1 * x1.y - s2 * x2.y;  // T1 - T2
          double br = s1 * x1.x + s2 * x2.x, bi = s1 * x1.y + s2 * x2.y;  // T1 + T2
          if (LIND) {
            const int ap = dig(dk, k), bp = dig(dk, l);
            const int skp = S.N * sk, slp = S.N * sl;
            const double s3 = L.tdn[ofs[k] + ap] * L.tup[ofs[l] + bp], s4 = L.tdn[ofs[l] + bp] * L.tup[ofs[k] + ap];
            const double2 x3 = sx[min(max(i0 - skp + slp, 0), top)], x4 = sx[min(max(i0 + skp - slp, 0), top)];
            ar += s3 * x3.x - s4 * x4.x;
            ai += s3 * x3.y - s4 * x4.y;
            br -= s3 * x3.x + s4 * x4.x;
            bi -= s3 * x3.y + s4 * x4.y;
          }
          const double co = c.cs[pair], si = c.sn[pair];
          hr += Jkl * (si * ar + co * bi);
          hi += Jkl * (si * ai - co * br);
        }
      }
    }
    double yr = TRANS ? -hr : hr, yi = TRANS ? -hi : hi;
    if (LIND) {
      yr = fma(dd[j], xs.x, yr) + l1r;
      yi = fma(dd[j], xs.y, yi) + l1i;
    }
    return make_double2(yr, yi);
  }

  // The same operator for vectors in GLOBAL memory (qd_big.h): the neighbours of the element are requested in batches of eight to ten
  // before any arithmetic on them - the ladder (and T1) neighbours of two (Lindblad) or all oscillators, the dipole-dipole neighbours of
  // two (Lindblad) or four pairs; scheduling barriers keep the batches apart (all of them at once does not fit 128 registers) - because a round trip to L2 costs 5-10 x an LDS access and the four waves per SIMD of those kernels cannot hide one per oscillator
  // and per coupling pair (apply() above: ten dependent round trips per element on the reference's nlevels_32_32_32_32 case).  No
  // branches on zero coefficients: a zero coefficient multiplies a clamped, valid read.
  template <bool TRANS>
  __device__ __forceinline__ double2 apply_batched(const DevSys& S, const Lds& L, const double2* __restrict__ sx, const StepC<Q>& c, int j,
                                                   const double2 xs) const {
    const int i0 = opaque(it[j]), top = S.dim - 1;
    const unsigned db = opaque(dbra[j]), dk = opaque(dket[j]);
    double hr = dw[j] * xs.y, hi = -dw[j] * xs.x;
    double l1r = 0.0, l1i = 0.0;
    constexpr int KC = LIND ? 2 : Q;  // oscillators per batch: at most ten reads in flight
#pragma unroll
    for (int k0 = 0; k0 < Q; k0 += KC) {
      double2 xu[KC], xd[KC], xup[KC], xdp[KC], xl[KC];
#pragma unroll
      for (int u = 0; u < KC; u++) {
        if (k0 + u >= Q) break;
        const int st = S.post[k0 + u];
        xu[u] = sx[min(i0 + st, top)];
        xd[u] = sx[max(i0 - st, 0)];
        if (LIND) {
          const int stp = S.N * st;
          xup[u] = sx[min(i0 + stp, top)];
          xdp[u] = sx[max(i0 - stp, 0)];
          xl[u] = sx[TRANS ? max(i0 - st - stp, 0) : min(i0 + st + stp, top)];
        }
      }
#pragma unroll
      for (int u = 0; u < KC; u++) {
        if (k0 + u >= Q) break;
        const int k = k0 + u, a = dig(db, k);
        const double su = L.tup[ofs[k] + a], sd = L.tdn[ofs[k] + a];
        double er = su * xu[u].x, ei = su * xu[u].y;
        double fr = -sd * xd[u].x, fi = -sd * xd[u].y;
        if (LIND) {
          const int ap = dig(dk, k);
          const double sup = L.tup[ofs[k] + ap], sdp = L.tdn[ofs[k] + ap];
          er = fma(-sdp, xdp[u].x, er);
          ei = fma(-sdp, xdp[u].y, ei);
          fr = fma(sup, xup[u].x, fr);
          fi = fma(sup, xup[u].y, fi);
          const double l1 = S.g1off[k] * (TRANS ? sd * sdp : su * sup);
          l1r = fma(l1, xl[u].x, l1r);
          l1i = fma(l1, xl[u].y, l1i);
        }
        hr = fma(c.q[k], er + fr, fma(c.p[k], ei - fi, hr));
        hi = fma(c.q[k], ei + fi, fma(-c.p[k], er - fr, hi));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (S.hasJ) {
      constexpr int NP = Q * (Q - 1) / 2 > 0 ? Q * (Q - 1) / 2 : 1, CH = LIND ? 2 : 4;  // pairs per batch: eight reads
      int pk[NP], pl[NP];
      {
        int pair = 0;
#pragma unroll
        for (int k = 0; k < Q; k++)
#pragma unroll
          for (int l = k + 1; l < Q; l++, pair++) {
            pk[pair] = k;
            pl[pair] = l;
          }
      }
#pragma unroll
      for (int p0 = 0; p0 < Q * (Q - 1) / 2; p0 += CH) {
        double2 x1[CH], x2[CH], x3[CH], x4[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
          if (p0 + u >= Q * (Q - 1) / 2) break;
          const int sk = S.post[pk[p0 + u]], sl = S.post[pl[p0 + u]];
          x1[u] = sx[min(max(i0 - sk + sl, 0), top)];
          x2[u] = sx[min(max(i0 + sk - sl, 0), top)];
          if (LIND) {
            const int skp = S.N * sk, slp = S.N * sl;
            x3[u] = sx[min(max(i0 - skp + slp, 0), top)];
            x4[u] = sx[min(max(i0 + skp - slp, 0), top)];
          }
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
          if (p0 + u >= Q * (Q - 1) / 2) break;
          const int pair = p0 + u, k = pk[pair], l = pl[pair];
          const double Jkl = S.J[pair];
          const int a = dig(db, k), b = dig(db, l);
          const double s1 = L.tdn[ofs[k] + a] * L.tup[ofs[l] + b], s2 = L.tdn[ofs[l] + b] * L.tup[ofs[k] + a];
          double ar = s1 * x1[u].x - s2 * x2[u].x, ai = s1 * x1[u].y - s2 * x2[u].y;  // T1 - T2
          double br = s1 * x1[u].x + s2 * x2[u].x, bi = s1 * x1[u].y + s2 * x2[u].y;  // T1 + T2
          if (LIND) {
            const int ap = dig(dk, k), bp = dig(dk, l);
            const double s3 = L.tdn[ofs[k] + ap] * L.tup[ofs[l] + bp], s4 = L.tdn[ofs[l] + bp] * L.tup[ofs[k] + ap];
            ar += s3 * x3[u].x - s4 * x4[u].x;
            ai += s3 * x3[u].y - s4 * x4[u].y;
            br -= s3 * x3[u].x + s4 * x4[u].x;
            bi -= s3 * x3[u].y + s4 * x4[u].y;
          }
          const double co = c.cs[pair], si = c.sn[pair];
          hr += Jkl * (si * ar + co * bi);
          hi += Jkl * (si * ai - co * br);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    double yr = TRANS ? -hr : hr, yi = TRANS ? -hi : hi;
    if (LIND) {
      yr = fma(dd[j], xs.x, yr) + l1r;
      yi = fma(dd[j], xs.y, yi) + l1i;
    }
    return make_double2(yr, yi);
  }

  // isGuardLevel (util.cpp:259-278) for a diagonal element
  __device__ __forceinline__ bool is_guard(const DevSys& S, int j) const {
    if (!valid[j]) return false;
    if (LIND && dbra[j] != dket[j]) return false;
    bool g = false;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const int a = dig(dbra[j], k);
      g = g || (a == S.n[k] - 1 && a >= S.ness[k]);
    }
    return g;
  }
};

// ---------------------------------------------------------------------------------------------
// qubit stencil (all n_k == 2): the digits are bits of `it`
//   bra digit of oscillator k = bit (Q-1-k), ket digit = bit (2Q-1-k); neighbours are it ^ bit.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double flip_if(double v, unsigned cond) {  // cond ? -v : v  (one v_xor_b32)
  return __hiloint2double(__double2hiint(v) ^ (int)(cond << 31), __double2loint(v));
}

template <int Q, bool LIND, int EPT, int EPE = EPT>
struct QubitStencil {
  static constexpr bool WHOLE = false;
  static constexpr bool NEEDS_SLOTS = false;
  int it[EPT];
  bool valid[EPT];  // only the single-wave variant can have idle lanes (dim < 64)
  double dw[EPT], dd[EPT];
  // latency regime (EPT == 1): loop invariants kept in registers instead of being re-derived from the
  // index bits in every operator application
  static constexpr bool HOIST = (EPE == 1);  // every slot of the thread is the SAME element (of different initial conditions)
  double l1f[HOIST ? Q : 1], l1t[HOIST ? Q : 1];  // T1 off-diagonal coefficient, forward / transposed
  double qb[HOIST ? Q : 1], qk[HOIST ? Q : 1];    // controls q_k with the bra / ket digit sign of this element

  // once per (sub-)step: fold the digit signs into the controls
  __device__ __forceinline__ void prep(const DevSys&, const Lds&, const StepC<Q>& c) {
    if (HOIST) {
#pragma unroll
      for (int k = 0; k < Q; k++) {
        qb[k] = flip_if(c.q[k], (it[0] >> (Q - 1 - k)) & 1);
        qk[k] = LIND ? flip_if(c.q[k], (it[0] >> (2 * Q - 1 - k)) & 1) : 0.0;
      }
    }
  }

  __device__ __forceinline__ void init(const DevSys& S, const Lds&) {
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int raw = (int)threadIdx.x + (j % EPE) * (int)blockDim.x;  // slot j = (initial condition j / EPE, element j % EPE)
      valid[j] = raw < S.dim;
      it[j] = raw & (S.dim - 1);  // dim is a power of two
      double hd = 0.0, hdp = 0.0, d = 0.0;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int a = (it[j] >> (Q - 1 - k)) & 1, ap = LIND ? (it[j] >> (2 * Q - 1 - k)) & 1 : 0;
        hd += S.detune[k] * a;  // the self-Kerr term a(a-1) vanishes for two levels
        if (LIND) {
          hdp += S.detune[k] * ap;
          d += S.g2[k] * (a * ap - 0.5 * (a + ap)) - S.g1[k] / 2.0 * (a + ap);
        }
#pragma unroll
        for (int l = k + 1; l < Q; l++) {
          const int b = (it[j] >> (Q - 1 - l)) & 1, bp = LIND ? (it[j] >> (2 * Q - 1 - l)) & 1 : 0;
          hd -= S.xikl[pair] * a * b;
          if (LIND) hdp -= S.xikl[pair] * ap * bp;
          pair++;
        }
      }
      dw[j] = hd - hdp;
      dd[j] = d;
    }
    if (HOIST && LIND) {
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int bits = (1 << (Q - 1 - k)) | (1 << (2 * Q - 1 - k));
        l1f[k] = ((it[0] & bits) == 0) ? S.g1off[k] : 0.0;
        l1t[k] = ((it[0] & bits) == bits) ? S.g1off[k] : 0.0;
      }
    }
  }

  // digit 0: only the "up" neighbour exists (U, coefficient sqrt(1) = 1); digit 1: only "down" (D).
  //   A = (U1 - D1) + (U2 - D2) = s_b x_b + s_k x_k,  B = (U1 + D1) - (U2 + D2) = x_b - x_k
  __device__ __forceinline__ void ladder(const DevSys&, const Lds&, const double2* __restrict__ sx, int k, int j, double2& A,
                                         double2& B) const {
    const int i0 = HOIST ? it[j] : opaque(it[j]);  // one element per thread: neighbour addresses may live in registers
    const unsigned a = (i0 >> (Q - 1 - k)) & 1;
    const double2 xb = sx[i0 ^ (1 << (Q - 1 - k))];
    A.x = flip_if(xb.x, a);
    A.y = flip_if(xb.y, a);
    B = xb;
    if (LIND) {
      const unsigned ap = (i0 >> (2 * Q - 1 - k)) & 1;
      const double2 xk = sx[i0 ^ (1 << (2 * Q - 1 - k))];
      A.x += flip_if(xk.x, ap);
      A.y += flip_if(xk.y, ap);
      B.x -= xk.x;
      B.y -= xk.y;
    }
  }

  // The neighbour reads and the arithmetic are separate entry points: the single-wave solver issues the reads
  // of the NEXT iteration before it reduces the update norm of the current one (Team::neumann).
  static constexpr bool SPLIT_FETCH = true;
  struct Nbrs {
    double2 xb[Q], xk[Q], xl[Q];  // (the ket / T1 sets are unused and eliminated for Schroedinger)
  };
  __device__ __forceinline__ void fetch(const double2* __restrict__ sx, int j, Nbrs& n) const {
    const int i0 = HOIST ? it[j] : opaque(it[j]);  // one element per thread: neighbour addresses may live in registers
#pragma unroll
    for (int k = 0; k < Q; k++) {
      n.xb[k] = sx[i0 ^ (1 << (Q - 1 - k))];
      if (LIND) {
        n.xk[k] = sx[i0 ^ (1 << (2 * Q - 1 - k))];
        n.xl[k] = sx[i0 ^ ((1 << (Q - 1 - k)) | (1 << (2 * Q - 1 - k)))];
      }
    }
  }

  template <bool TRANS>
  __device__ __forceinline__ double2 apply(const DevSys& S, const Lds& L, const double2* __restrict__ sx, const StepC<Q>& c, int j,
                                           const double2 xs) const {
    Nbrs n;
    fetch(sx, j, n);  // all neighbour reads first (one LDS latency per application)
    return apply_nb<TRANS>(S, sx, c, j, xs, n);
  }

  template <bool TRANS>
  __device__ __forceinline__ double2 apply_nb(const DevSys& S, const double2* __restrict__ sx, const StepC<Q>& c, int j, const double2 xs,
                                              const Nbrs& n) const {
    const int i0 = HOIST ? it[j] : opaque(it[j]);
    const double2* xb = n.xb;
    const double2* xk = n.xk;
    const double2* xl = n.xl;
    double2 own[Q];
    if (QD_ABLATE & 2) {  // no neighbour reads: the element itself stands in
#pragma unroll
      for (int k = 0; k < Q; k++) own[k] = xs;
      xb = xk = xl = own;
    }
    if (QD_ABLATE & 4) {  // no stencil arithmetic: a sum of the neighbours keeps the reads alive
      double2 acc = xs;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        acc.x += xb[k].x + (LIND ? xk[k].x + xl[k].x : 0.0);
        acc.y += xb[k].y + (LIND ? xk[k].y + xl[k].y : 0.0);
      }
      return make_double2(1e-3 * acc.x, 1e-3 * acc.y);
    }
    // control part with the digit signs folded into q:  q A.x + p B.y = (+-q) xb.x + (+-q) xk.x + p (xb.y - xk.y);
    // two accumulator pairs shorten the dependent fp64 chain (the small-system kernels are latency bound)
    double hr = dw[j] * xs.y, hi = -dw[j] * xs.x, gr = 0.0, gi = 0.0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const unsigned a = (i0 >> (Q - 1 - k)) & 1;
      const double sqb = HOIST ? qb[k] : flip_if(c.q[k], a);
      if (!LIND) {
        // Schroedinger: FMAs only, accumulated into two pairs (even / odd oscillators)
        double& ar = (k & 1) ? gr : hr;
        double& ai = (k & 1) ? gi : hi;
        ar = fma(sqb, xb[k].x, ar);
        ai = fma(sqb, xb[k].y, ai);
        ar = fma(c.p[k], xb[k].y, ar);
        ai = fma(-c.p[k], xb[k].x, ai);
      } else {
        // Lindblad (latency-bound single-wave kernels): independent product trees per oscillator, summed at
        // the end - the FMA-only chain is two instructions shorter but serial, measured 6 % slower on C2
        const unsigned ap = (i0 >> (2 * Q - 1 - k)) & 1;
        const double sqk = HOIST ? qk[k] : flip_if(c.q[k], ap);
        double tr = fma(sqk, xk[k].x, sqb * xb[k].x), ti = fma(sqk, xk[k].y, sqb * xb[k].y);
        tr = fma(c.p[k], xb[k].y - xk[k].y, tr);
        ti = fma(-c.p[k], xb[k].x - xk[k].x, ti);
        if (k == 1) { gr = tr; gi = ti; }  // (no "+= 0.0": fp64 adds are not folded)
        else if (k & 1) { gr += tr; gi += ti; }
        else { hr += tr; hi += ti; }
      }
    }
    if (Q > 1) {
      hr += gr;
      hi += gi;
    }
    if (S.hasJ) {
      // Jkl coupling for two-level systems: T1 exists iff (i_k, i_l) = (1, 0), T2 iff (0, 1), both read
      // x(it ^ (bit_k | bit_l)) with coefficient 1; T3/T4 likewise on the ket bits.
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
#pragma unroll
        for (int l = k + 1; l < Q; l++, pair++) {
          const double Jkl = S.J[pair];
          const unsigned a = (i0 >> (Q - 1 - k)) & 1, b = (i0 >> (Q - 1 - l)) & 1;
          const double2 xj = sx[i0 ^ ((1 << (Q - 1 - k)) | (1 << (Q - 1 - l)))];
          const double mb = (a != b) ? 1.0 : 0.0;
          // a=1,b=0: T1 (A +, B +); a=0,b=1: T2 (A -, B +)
          double ar = mb * flip_if(xj.x, b), ai = mb * flip_if(xj.y, b);
          double br = mb * xj.x, bi = mb * xj.y;
          if (LIND) {
            const unsigned ap = (i0 >> (2 * Q - 1 - k)) & 1, bp = (i0 >> (2 * Q - 1 - l)) & 1;
            const double2 xq = sx[i0 ^ ((1 << (2 * Q - 1 - k)) | (1 << (2 * Q - 1 - l)))];
            const double mk = (ap != bp) ? 1.0 : 0.0;
            // ap=1,bp=0: T3 (A +, B -); ap=0,bp=1: T4 (A -, B -)
            ar += mk * flip_if(xq.x, bp);
            ai += mk * flip_if(xq.y, bp);
            br -= mk * xq.x;
            bi -= mk * xq.y;
          }
          const double co = c.cs[pair], si = c.sn[pair];
          hr += Jkl * (si * ar + co * bi);
          hi += Jkl * (si * ai - co * br);
        }
      }
    }
    double yr = TRANS ? -hr : hr, yi = TRANS ? -hi : hi;
    if (LIND) {
      yr = fma(dd[j], xs.x, yr);
      yi = fma(dd[j], xs.y, yi);
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int bits = (1 << (Q - 1 - k)) | (1 << (2 * Q - 1 - k));
        // forward: both digits 0 -> neighbour with both set; transpose: both 1 -> neighbour with both cleared
        const bool v = TRANS ? ((i0 & bits) == bits) : ((i0 & bits) == 0);
        const double l1 = HOIST ? (TRANS ? l1t[k] : l1f[k]) : (v ? S.g1off[k] : 0.0);
        yr = fma(l1, xl[k].x, yr);
        yi = fma(l1, xl[k].y, yi);
      }
    }
    return make_double2(yr, yi);
  }

  __device__ __forceinline__ bool is_guard(const DevSys&, int) const { return false; }  // nessential == nlevels == 2 ... see host check
};


// ---------------------------------------------------------------------------------------------
// column stencil (Lindblad, runtime level counts, N <= 64): lane = row I of rho, wave w owns the EPT columns
// I' = w EPT .. w EPT + EPT - 1  (vectorised index it = I' N + I, util.cpp:150).  Every ket-side
// quantity (digits of I', ladder coefficients, neighbour columns) is wave-uniform and lives on the
// scalar unit or in one broadcast LDS read; every bra-side quantity is an invariant of the thread.
// Nothing per slot has to be kept in vector registers except the two diagonal coefficients.
// ---------------------------------------------------------------------------------------------
// PACKED (N <= 32): floor(64 / N) columns share one wave, lane = (column within the wave slot) N + row.  The ket
// side is then uniform per lane group instead of per wave (plain vector arithmetic instead of scalars),
// and only the bra neighbours of the stride-1 oscillator still come from registers (adjacent lanes).
template <int Q, int EPT, bool PACKED = false>
struct ColStencil {
  static constexpr bool WHOLE = false;
  static constexpr bool NEEDS_SLOTS = true;  // apply() takes the thread's other elements of the vector being read
  static constexpr int DB = packed_digit_bits(Q);
  int it[EPT];
  bool valid[EPT];
  double dw[EPT], dd[EPT];
  int N, row, col0;
  int cpw, lc;  // PACKED: columns per wave slot, this lane's column within the slot
  unsigned dbra;
  double su[Q], sd[Q];    // sqrt(i_k + 1) (0 at the top level), sqrt(i_k) of this thread's row
  double g1u[Q], g1d[Q];  // gamma_1 su / gamma_1 sd (T1 off-diagonal, forward / transposed)
  int rup[Q], rdn[Q];     // row of the bra "up" / "down" neighbour, clamped into the column
  int ofs[Q];

  __device__ __forceinline__ static int dig(unsigned d, int k) { return (int)((d >> (DB * k)) & ((1u << DB) - 1u)); }
  // column of slot j (clamped): wave-uniform unless PACKED
  __device__ __forceinline__ int colof(int j) const { return PACKED ? min((col0 + j) * cpw + lc, N - 1) : min(col0 + j, N - 1); }

  __device__ __forceinline__ void init(const DevSys& S, const Lds& L) {
    N = S.N;
    const int lane = threadIdx.x & 63;
    col0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * EPT;  // wave w owns the column slots w EPT .. w EPT + EPT - 1
    cpw = PACKED ? 64 / N : 1;
    lc = PACKED ? lane / N : 0;
    const bool rowok = PACKED ? lc < cpw : lane < N;
    row = rowok ? (PACKED ? lane - lc * N : lane) : N - 1;
    if (!rowok) lc = cpw - 1;
    int o = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      ofs[k] = o;
      o += S.n[k];
    }
    for (int k = 0; k < Q; k++)
      for (int a = threadIdx.x; a < S.n[k]; a += blockDim.x) {
        L.tup[ofs[k] + a] = (a < S.n[k] - 1) ? sqrt((double)(a + 1)) : 0.0;
        L.tdn[ofs[k] + a] = sqrt((double)a);
      }
    for (int e = threadIdx.x; e < N * Q; e += blockDim.x) {
      const int cc = e / Q, k = e % Q;
      const int ap = (cc / S.post[k]) % S.n[k];
      L.coltab[e] = make_double2((ap < S.n[k] - 1) ? sqrt((double)(ap + 1)) : 0.0, sqrt((double)ap));
    }
    int ia[Q];
    dbra = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      ia[k] = (row / S.post[k]) % S.n[k];
      dbra |= (unsigned)ia[k] << (DB * k);
      su[k] = (ia[k] < S.n[k] - 1) ? sqrt((double)(ia[k] + 1)) : 0.0;
      sd[k] = sqrt((double)ia[k]);
      g1u[k] = S.g1off[k] * su[k];
      g1d[k] = S.g1off[k] * sd[k];
      rup[k] = min(row + S.post[k], N - 1);
      rdn[k] = max(row - S.post[k], 0);
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int craw = PACKED ? (col0 + j) * cpw + lc : col0 + j, cc = min(craw, N - 1);
      valid[j] = rowok && craw < N;
      it[j] = cc * N + row;
      int ipa[Q];
#pragma unroll
      for (int k = 0; k < Q; k++) ipa[k] = (cc / S.post[k]) % S.n[k];
      double hd = 0.0, hdp = 0.0, d = 0.0;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        hd += S.detune[k] * ia[k] - S.xi[k] / 2.0 * ia[k] * (ia[k] - 1);
        hdp += S.detune[k] * ipa[k] - S.xi[k] / 2.0 * ipa[k] * (ipa[k] - 1);
        d += S.g2[k] * (ia[k] * ipa[k] - 0.5 * (ia[k] * ia[k] + ipa[k] * ipa[k])) - S.g1[k] / 2.0 * (ia[k] + ipa[k]);
#pragma unroll
        for (int l = k + 1; l < Q; l++) {
          hd -= S.xikl[pair] * ia[k] * ia[l];
          hdp -= S.xikl[pair] * ipa[k] * ipa[l];
          pair++;
        }
      }
      dw[j] = hd - hdp;
      dd[j] = d;
    }
  }

  __device__ __forceinline__ void prep(const DevSys&, const Lds&, const StepC<Q>&) {}

  // see GenStencil::ladder
  __device__ __forceinline__ void ladder(const DevSys& S, const Lds& L, const double2* __restrict__ sx, int k, int j, double2& A,
                                         double2& B) const {
    const int cc = colof(j), cN = cc * N, st = S.post[k];
    const int cu = min(cc + st, N - 1) * N, cd = max(cc - st, 0) * N;
    const double2 ct = L.coltab[cc * Q + k];
    const int r0 = opaque(row), ru = opaque(rup[k]), rd = opaque(rdn[k]);  // addresses are re-derived, not hoisted
    const double2 xu = sx[cN + ru], xd = sx[cN + rd], xup = sx[cu + r0], xdp = sx[cd + r0];
    const double er = fma(-ct.y, xdp.x, su[k] * xu.x), ei = fma(-ct.y, xdp.y, su[k] * xu.y);    // U1 - D2
    const double fr = fma(ct.x, xup.x, -sd[k] * xd.x), fi = fma(ct.x, xup.y, -sd[k] * xd.y);    // U2 - D1
    A.x = er + fr;
    A.y = ei + fi;
    B.x = er - fr;
    B.y = ei - fi;
  }

  // HASJ = false: the caller has checked S.hasJ == 0; the slot loop is then free of branches and two
  // slots can be in flight (Variant::FENCE)
  // The last oscillator has stride 1 (post[Q-1] == 1): its bra neighbours are the adjacent LANES of the
  // same slot and its ket neighbours the adjacent SLOTS of the same thread, so they come from registers
  // (`xall` = the thread's elements of the vector being read: own element xall[j], slots j-1 / j+1) instead
  // of LDS; only the first / last column of a wave's block reads the neighbouring wave's column.  A
  // neighbour that does not exist always has a zero coefficient, the value fetched for it is arbitrary
  // but finite (idle lanes and slots compute on clamped indices).
  template <bool TRANS, bool HASJ = true>
  __device__ __forceinline__ double2 apply(const DevSys& S, const Lds& L, const double2* __restrict__ sx, const StepC<Q>& c, int j,
                                           const double2 (&xall)[EPT]) const {
    const double2 xs = xall[j], xprev = xall[j > 0 ? j - 1 : 0], xnext = xall[j + 1 < EPT ? j + 1 : j];
    const int cc = colof(j), cN = cc * N;
    double hr = dw[j] * xs.y, hi = -dw[j] * xs.x;
    double l1r = 0.0, l1i = 0.0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const int st = S.post[k];
      const int cu = min(cc + st, N - 1) * N, cd = max(cc - st, 0) * N;
      const double2 ct = L.coltab[cc * Q + k];
      const int r0 = opaque(row), ru = opaque(rup[k]), rd = opaque(rdn[k]);  // addresses are re-derived, not hoisted
      const bool last = (k == Q - 1);
      const bool reg_up = !PACKED && last && j < EPT - 1, reg_dn = !PACKED && last && j > 0;  // compile-time after unrolling
      const double2 xu = last ? lane_shift<true>(xs) : sx[cN + ru];
      const double2 xd = last ? lane_shift<false>(xs) : sx[cN + rd];
      const double2 xup = reg_up ? xnext : sx[cu + r0];
      const double2 xdp = reg_dn ? xprev : sx[cd + r0];
      const double er = fma(-ct.y, xdp.x, su[k] * xu.x), ei = fma(-ct.y, xdp.y, su[k] * xu.y);
      const double fr = fma(ct.x, xup.x, -sd[k] * xd.x), fi = fma(ct.x, xup.y, -sd[k] * xd.y);
      hr = fma(c.q[k], er + fr, fma(c.p[k], ei - fi, hr));
      hi = fma(c.q[k], ei + fi, fma(-c.p[k], er - fr, hi));
      {  // T1 off-diagonal term; without decay the coefficient is an exact zero
        double2 xl;
        if (TRANS) xl = reg_dn ? lane_shift<false>(xprev) : sx[cd + rd];
        else xl = reg_up ? lane_shift<true>(xnext) : sx[cu + ru];
        const double l1 = TRANS ? g1d[k] * ct.y : g1u[k] * ct.x;
        l1r = fma(l1, xl.x, l1r);
        l1i = fma(l1, xl.y, l1i);
      }
    }
    if (HASJ && S.hasJ) {  // dipole-dipole coupling: the generic formulation of GenStencil::apply
      const int i0 = opaque(it[j]), top = S.dim - 1;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
#pragma unroll
        for (int l = k + 1; l < Q; l++, pair++) {
          const double Jkl = S.J[pair];
          if (Jkl == 0.0) continue;
          const unsigned db = opaque(dbra);
          const int a = dig(db, k), b = dig(db, l);
          const int sk = S.post[k], sl = S.post[l];
          const double s1 = L.tdn[ofs[k] + a] * L.tup[ofs[l] + b], s2 = L.tdn[ofs[l] + b] * L.tup[ofs[k] + a];
          const double2 x1 = sx[min(max(i0 - sk + sl, 0), top)], x2 = sx[min(max(i0 + sk - sl, 0), top)];
          double ar = s1 * x1.x - s2 * x2.x, ai = s1 * x1.y - s2 * x2.y;
          double br = s1 * x1.x + s2 * x2.x, bi = s1 * x1.y + s2 * x2.y;
          const int ap = (cc / sk) % S.n[k], bp = (cc / sl) % S.n[l];
          const int skp = N * sk, slp = N * sl;
          const double s3 = L.tdn[ofs[k] + ap] * L.tup[ofs[l] + bp], s4 = L.tdn[ofs[l] + bp] * L.tup[ofs[k] + ap];
          const double2 x3 = sx[min(max(i0 - skp + slp, 0), top)], x4 = sx[min(max(i0 + skp - slp, 0), top)];
          ar += s3 * x3.x - s4 * x4.x;
          ai += s3 * x3.y - s4 * x4.y;
          br -= s3 * x3.x + s4 * x4.x;
          bi -= s3 * x3.y + s4 * x4.y;
          const double co = c.cs[pair], si = c.sn[pair];
          hr += Jkl * (si * ar + co * bi);
          hi += Jkl * (si * ai - co * br);
        }
      }
    }
    const double yr = fma(dd[j], xs.x, TRANS ? -hr : hr) + l1r, yi = fma(dd[j], xs.y, TRANS ? -hi : hi) + l1i;
    return make_double2(yr, yi);
  }

  // isGuardLevel (util.cpp:259-278) for a diagonal element
  __device__ __forceinline__ bool is_guard(const DevSys& S, int j) const {
    if (!valid[j] || colof(j) != row) return false;
    bool g = false;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const int a = dig(dbra, k);
      g = g || (a == S.n[k] - 1 && a >= S.ness[k]);
    }
    return g;
  }
};


// ---------------------------------------------------------------------------------------------
// qubit stencil, four elements per thread (Lindblad, dim = 4^Q, block = dim / 4): element of slot j is
// it = tid | j << (2Q-2).  Bits 0 .. 2Q-3 of `it` belong to the thread, the two top bits (ket digits of
// oscillators 1 and 0) are the slot number, i.e. compile-time constants after unrolling.  Hence
//   * every digit sign and T1 validity is either a thread invariant or a constant (nothing per element),
//   * the LDS offset of a neighbour is a thread invariant plus an immediate slot offset,
//   * the ket neighbours of oscillators 0 and 1 are the thread's OWN other slots (registers, no LDS).
// ---------------------------------------------------------------------------------------------
template <int Q>
struct QubitSlotStencil {
  static_assert(Q >= 2, "needs two oscillators for the slot bits");
  static constexpr bool WHOLE = false;
  static constexpr bool NEEDS_SLOTS = true;
  static constexpr int EPT = 4;
  static constexpr int TB = 2 * Q - 2;                 // number of thread bits
  static constexpr unsigned SLOT_BYTES = 16u << TB;    // LDS distance of consecutive slots
  int it[EPT];
  bool valid[EPT];
  double dw[EPT], dd[EPT];
  unsigned ab[Q], ak[Q], al[Q];  // byte offsets (slot 0) of the bra / ket (k >= 2) / T1 (k >= 2) neighbour of oscillator k
  double qb[Q], qk[Q];           // q_k with the sign of the bra / (k >= 2) ket digit of this thread [per step]
  double l1f[Q], l1t[Q];         // thread part of the T1 off-diagonal coefficient, forward / transposed

  __device__ __forceinline__ static constexpr int brabit(int k) { return Q - 1 - k; }
  __device__ __forceinline__ static constexpr int ketbit(int k) { return 2 * Q - 1 - k; }
  // slot bit of oscillator k < 2: bit (1 - k) of the slot number
  __device__ __forceinline__ static constexpr int slotbit(int j, int k) { return (j >> (1 - k)) & 1; }

  __device__ __forceinline__ void init(const DevSys& S, const Lds&) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      it[j] = (int)(tid | ((unsigned)j << TB));
      valid[j] = true;
      double hd = 0.0, hdp = 0.0, d = 0.0;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int a = (it[j] >> brabit(k)) & 1, ap = (it[j] >> ketbit(k)) & 1;
        hd += S.detune[k] * a;
        hdp += S.detune[k] * ap;
        d += S.g2[k] * (a * ap - 0.5 * (a + ap)) - S.g1[k] / 2.0 * (a + ap);
#pragma unroll
        for (int l = k + 1; l < Q; l++) {
          const int b = (it[j] >> brabit(l)) & 1, bp = (it[j] >> ketbit(l)) & 1;
          hd -= S.xikl[pair] * a * b;
          hdp -= S.xikl[pair] * ap * bp;
          pair++;
        }
      }
      dw[j] = hd - hdp;
      dd[j] = d;
    }
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const unsigned bb = 1u << brabit(k), kb = 1u << ketbit(k);
      ab[k] = (tid ^ bb) << 4;
      ak[k] = k >= 2 ? (tid ^ kb) << 4 : 0u;
      al[k] = k >= 2 ? (tid ^ bb ^ kb) << 4 : ab[k];
      const bool bra0 = (tid & bb) == 0;
      if (k >= 2) {
        const bool ket0 = (tid & kb) == 0;
        l1f[k] = (bra0 && ket0) ? S.g1off[k] : 0.0;
        l1t[k] = (!bra0 && !ket0) ? S.g1off[k] : 0.0;
      } else {  // the ket digit is a slot bit: only the bra condition is a thread property
        l1f[k] = bra0 ? S.g1off[k] : 0.0;
        l1t[k] = !bra0 ? S.g1off[k] : 0.0;
      }
      qb[k] = qk[k] = 0.0;
    }
  }

  __device__ __forceinline__ void prep(const DevSys&, const Lds&, const StepC<Q>& c) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      qb[k] = flip_if(c.q[k], (tid >> brabit(k)) & 1);
      if (k >= 2) qk[k] = flip_if(c.q[k], (tid >> ketbit(k)) & 1);
    }
  }

  __device__ __forceinline__ static double2 at(const double2* __restrict__ sx, unsigned byteoff, int slot) {
    return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(sx) + byteoff + (unsigned)slot * SLOT_BYTES);
  }

  // gradient contraction (once per step): plain LDS formulation, see QubitStencil::ladder
  __device__ __forceinline__ void ladder(const DevSys&, const Lds&, const double2* __restrict__ sx, int k, int j, double2& A,
                                         double2& B) const {
    const int i0 = opaque(it[j]);
    const unsigned a = (i0 >> brabit(k)) & 1, ap = (i0 >> ketbit(k)) & 1;
    const double2 xb = sx[i0 ^ (1 << brabit(k))], xk = sx[i0 ^ (1 << ketbit(k))];
    A.x = flip_if(xb.x, a) + flip_if(xk.x, ap);
    A.y = flip_if(xb.y, a) + flip_if(xk.y, ap);
    B.x = xb.x - xk.x;
    B.y = xb.y - xk.y;
  }

  template <bool TRANS, bool HASJ = true>
  __device__ __forceinline__ double2 apply(const DevSys& S, const Lds&, const double2* __restrict__ sx, const StepC<Q>& c, int j,
                                           const double2 (&xall)[EPT]) const {
    const double2 xs = xall[j];
    double hr = dw[j] * xs.y, hi = -dw[j] * xs.x, gr = 0.0, gi = 0.0;
    double l1r = 0.0, l1i = 0.0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const double2 xb = at(sx, ab[k], j);
      double2 xk;
      double sqk;
      if (k < 2) {  // ket neighbour = own slot with the slot bit flipped; sign of the slot bit is a constant
        xk = xall[j ^ (1 << (1 - k))];
        sqk = slotbit(j, k) ? -c.q[k] : c.q[k];
      } else {
        xk = at(sx, ak[k], j);
        sqk = qk[k];
      }
      // q (s_b x_b + s_k x_k) + p (x_b.y - x_k.y) etc., accumulated by FMAs only; two accumulator pairs
      // (even / odd oscillators) keep four independent dependency chains
      double& ar = (k & 1) ? gr : hr;
      double& ai = (k & 1) ? gi : hi;
      ar = fma(qb[k], xb.x, ar);
      ai = fma(qb[k], xb.y, ai);
      ar = fma(sqk, xk.x, ar);
      ai = fma(sqk, xk.y, ai);
      ar = fma(c.p[k], xb.y, ar);
      ai = fma(-c.p[k], xb.x, ai);
      ar = fma(-c.p[k], xk.y, ar);
      ai = fma(c.p[k], xk.x, ai);
      // T1 off-diagonal: forward needs both digits 0 (neighbour has both set), transposed both 1
      const bool slot_ok = k >= 2 || (TRANS ? slotbit(j, k) == 1 : slotbit(j, k) == 0);
      if (slot_ok) {
        const double2 xl = at(sx, al[k], k < 2 ? (j ^ (1 << (1 - k))) : j);
        const double l1 = TRANS ? l1t[k] : l1f[k];
        l1r = fma(l1, xl.x, l1r);
        l1i = fma(l1, xl.y, l1i);
      }
    }
    hr += gr;
    hi += gi;
    if (HASJ && S.hasJ) {  // see QubitStencil::apply
      const int i0 = opaque(it[j]);
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
#pragma unroll
        for (int l = k + 1; l < Q; l++, pair++) {
          const double Jkl = S.J[pair];
          const unsigned a = (i0 >> brabit(k)) & 1, b = (i0 >> brabit(l)) & 1;
          const double2 xj = sx[i0 ^ ((1 << brabit(k)) | (1 << brabit(l)))];
          const double mb = (a != b) ? 1.0 : 0.0;
          double ar = mb * flip_if(xj.x, b), ai = mb * flip_if(xj.y, b);
          double br = mb * xj.x, bi = mb * xj.y;
          const unsigned ap = (i0 >> ketbit(k)) & 1, bp = (i0 >> ketbit(l)) & 1;
          const double2 xq = sx[i0 ^ ((1 << ketbit(k)) | (1 << ketbit(l)))];
          const double mk = (ap != bp) ? 1.0 : 0.0;
          ar += mk * flip_if(xq.x, bp);
          ai += mk * flip_if(xq.y, bp);
          br -= mk * xq.x;
          bi -= mk * xq.y;
          const double co = c.cs[pair], si = c.sn[pair];
          hr += Jkl * (si * ar + co * bi);
          hi += Jkl * (si * ai - co * br);
        }
      }
    }
    const double yr = fma(dd[j], xs.x, TRANS ? -hr : hr) + l1r, yi = fma(dd[j], xs.y, TRANS ? -hi : hi) + l1i;
    return make_double2(yr, yi);
  }

  __device__ __forceinline__ bool is_guard(const DevSys&, int) const { return false; }
};


// ---------------------------------------------------------------------------------------------
// dense operator (user-supplied Hamiltonians; the reference serves them with its sparse-matrix solver:
// src/hamiltonianfilereader.cpp, applyRHS_sparsemat src/mastereq.cpp:743-967).  In Hilbert space
//   G(t) = -i H(t),  H(t) = Hsys + sum_k p_k(t) Re(Hc_k) + i q_k(t) Im(Hc_k)
//   Schroedinger: y = G psi          Lindblad: y = G rho - rho G + (T1/T2 dissipators of the standard model)
// and the transposed real operator is the same expression with G^H.  G(t) of every sub-step is tabulated
// once per parameter update (k_gmat, shared by all initial conditions, read through L2); the
// dissipators reuse the general stencil's digit tables.
// ---------------------------------------------------------------------------------------------
template <int Q, bool LIND, int EPT, int EPE = EPT>
struct DenseStencil : GenStencil<Q, LIND, EPT, EPE> {
  typedef GenStencil<Q, LIND, EPT, EPE> Base;
  using Base::it;
  using Base::dbra;
  using Base::dket;
  using Base::dd;
  using Base::ofs;
  using Base::dig;

  // once per (sub-)step: stage G(t) in LDS (every element reads 2N entries of it per operator application)
  // (S.dense == 2: the matrix fits, N <= 64; otherwise it is read from the table through L2)
  __device__ __forceinline__ void prep(const DevSys& S, const Lds& L, const StepC<Q>& c) {
    if (!L.gmat) return;
    if (blockDim.x > 64) __syncthreads();  // nobody still reads the previous step's matrix
    const int nn = S.N * S.N;
    for (int e = threadIdx.x; e < nn; e += blockDim.x) L.gmat[e] = c.g[e];
    if (blockDim.x > 64) __syncthreads();
  }

  __device__ __forceinline__ static double2 cmul_acc(double2 acc, double2 a, double2 b) {  // acc + a b
    acc.x = fma(a.x, b.x, fma(-a.y, b.y, acc.x));
    acc.y = fma(a.x, b.y, fma(a.y, b.x, acc.y));
    return acc;
  }

  // sum_m Gt(I,m) x(m,I') - x(I,m) Gt(m,I')  with Gt = G or G^H
  template <bool TRANS, typename GET>
  __device__ __forceinline__ double2 commutator(const DevSys& S, const double2* __restrict__ sx, int I, int Ip, GET g) const {
    const int N = S.N;
    double2 acc = make_double2(0.0, 0.0);
    for (int m = 0; m < N; m++) {
      double2 gl = TRANS ? g(m, I) : g(I, m);
      if (TRANS) gl.y = -gl.y;
      acc = cmul_acc(acc, gl, LIND ? sx[Ip * N + m] : sx[m]);
      if (LIND) {
        double2 gr = TRANS ? g(Ip, m) : g(m, Ip);
        if (TRANS) gr.y = -gr.y;
        gr.x = -gr.x;
        gr.y = -gr.y;
        acc = cmul_acc(acc, sx[m * N + I], gr);
      }
    }
    return acc;
  }

  // gradient contraction: A = [Im(Hc_k), z], B = [Re(Hc_k), z]  (the control part of M z is q A - i p B)
  __device__ __forceinline__ void ladder(const DevSys& S, const Lds&, const double2* __restrict__ sx, int k, int j, double2& A,
                                         double2& B) const {
    const int N = S.N, i0 = opaque(it[j]);
    const int I = LIND ? i0 % N : i0, Ip = LIND ? i0 / N : 0;
    const double* hr = S.hcr + (size_t)k * N * N;
    const double* hi = S.hci + (size_t)k * N * N;
    A = commutator<false>(S, sx, I, Ip, [&](int r, int c) { return make_double2(hi[r * N + c], 0.0); });
    B = commutator<false>(S, sx, I, Ip, [&](int r, int c) { return make_double2(hr[r * N + c], 0.0); });
  }

  template <bool TRANS>
  __device__ __forceinline__ double2 apply(const DevSys& S, const Lds& L, const double2* __restrict__ sx, const StepC<Q>& c, int j,
                                           const double2 xs) const {
    const int N = S.N, i0 = opaque(it[j]), top = S.dim - 1;
    const int I = LIND ? i0 % N : i0, Ip = LIND ? i0 / N : 0;
    const double2* __restrict__ G = L.gmat ? L.gmat : c.g;
    double2 y = commutator<TRANS>(S, sx, I, Ip, [&](int r, int cc) { return G[r * N + cc]; });
    if (LIND) {
      y.x = fma(dd[j], xs.x, y.x);
      y.y = fma(dd[j], xs.y, y.y);
      const unsigned db = opaque(dbra[j]), dk = opaque(dket[j]);
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const double g1 = S.g1off[k];
        if (g1 == 0.0) continue;
        const int a = dig(db, k), ap = dig(dk, k), st = S.post[k] * (N + 1);
        const double l1 = g1 * (TRANS ? L.tdn[ofs[k] + a] * L.tdn[ofs[k] + ap] : L.tup[ofs[k] + a] * L.tup[ofs[k] + ap]);
        const double2 xn = sx[TRANS ? max(i0 - st, 0) : min(i0 + st, top)];
        y.x = fma(l1, xn.x, y.x);
        y.y = fma(l1, xn.y, y.y);
      }
    }
    return y;
  }
};


// ---------------------------------------------------------------------------------------------
// dense operator on the matrix cores, N = 16 Lindblad (dim 256): one wave owns rho in the accumulator
// layout of v_mfma_f64_16x16x4_f64 (slot r of lane l = row (l >> 4) + 4 r, column l & 15), which is at
// the same time the B-operand layout of the four K-slabs.  Per application
//   Y = G rho - rho G   = 2 complex 16x16x16 products = 32 MFMA instructions
// with G (A operand of the first, B operand of the second product) held in registers for the whole
// sub-step and rho as A operand read transposed from the published LDS vector (4 reads per lane);
// the transposed real operator swaps the two register sets and conjugates them (G -> G^H).
// Dissipators and the gradient contraction are inherited from DenseStencil.
// ---------------------------------------------------------------------------------------------
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

template <int Q>
struct DenseMfmaStencil : DenseStencil<Q, true, 4, 4> {
  typedef DenseStencil<Q, true, 4, 4> Base;
  static constexpr bool WHOLE = true;  // apply_whole() computes all four slots at once
  static constexpr int N = 16, EPT = 4;
  using Base::dbra;
  using Base::dd;
  using Base::dig;
  using Base::dket;
  using Base::it;
  using Base::ofs;
  using Base::valid;
  double2 gA[4];  // G[l & 15][4 s + (l >> 4)]
  double2 gB[4];  // G[4 s + (l >> 4)][l & 15]
  double l1f[EPT][Q], l1t[EPT][Q];  // T1 off-diagonal coefficients of the four elements (one wave per SIMD: registers are plentiful)

  __device__ __forceinline__ void init(const DevSys& S, const Lds& L) {
    Base::init(S, L);  // tables, then re-map the slots onto the accumulator layout
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int I = (lane >> 4) + 4 * j, Ip = lane & 15;
      it[j] = Ip * N + I;
      valid[j] = true;
      int ia[Q], ipa[Q];
      dbra[j] = 0;
      dket[j] = 0;
      double d = 0.0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        ia[k] = (I / S.post[k]) % S.n[k];
        ipa[k] = (Ip / S.post[k]) % S.n[k];
        dbra[j] |= (unsigned)ia[k] << (Base::DB * k);
        dket[j] |= (unsigned)ipa[k] << (Base::DB * k);
        d += S.g2[k] * (ia[k] * ipa[k] - 0.5 * (ia[k] * ia[k] + ipa[k] * ipa[k])) - S.g1[k] / 2.0 * (ia[k] + ipa[k]);
        const bool up = ia[k] < S.n[k] - 1 && ipa[k] < S.n[k] - 1;
        l1f[j][k] = up ? S.g1off[k] * sqrt((double)(ia[k] + 1)) * sqrt((double)(ipa[k] + 1)) : 0.0;
        l1t[j][k] = S.g1off[k] * sqrt((double)ia[k]) * sqrt((double)ipa[k]);
      }
      dd[j] = d;
    }
  }

  __device__ __forceinline__ void prep(const DevSys&, const Lds&, const StepC<Q>& c) {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      gA[s] = c.g[lo * N + 4 * s + hi];
      gB[s] = c.g[(4 * s + hi) * N + lo];
    }
  }

  template <bool TRANS>
  __device__ __forceinline__ void apply_whole(const DevSys& S, const Lds& L, const double2* __restrict__ sx, const StepC<Q>&,
                                              const double2 (&x)[EPT], double2 (&y)[EPT]) const {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
    mfma_d4 ar = {0.0, 0.0, 0.0, 0.0}, ai = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; s++) {
      // first product: Gt rho, Gt = G (A operand gA) or G^H (A operand conj(gB))
      const double gr = TRANS ? gB[s].x : gA[s].x, gi = TRANS ? -gB[s].y : gA[s].y;
      ar = __builtin_amdgcn_mfma_f64_16x16x4f64(gr, x[s].x, ar, 0, 0, 0);
      ar = __builtin_amdgcn_mfma_f64_16x16x4f64(-gi, x[s].y, ar, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f64_16x16x4f64(gr, x[s].y, ai, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f64_16x16x4f64(gi, x[s].x, ai, 0, 0, 0);
      // second product: - rho Gt, rho as A operand (row lo, column 4 s + hi) from the published vector
      const double2 pa = sx[(4 * s + hi) * N + lo];
      const double hr = TRANS ? gA[s].x : gB[s].x, hi2 = TRANS ? -gA[s].y : gB[s].y;
      ar = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa.x, hr, ar, 0, 0, 0);
      ar = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, hi2, ar, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa.x, hi2, ai, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa.y, hr, ai, 0, 0, 0);
    }
    const int top = S.dim - 1;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      double yr = fma(dd[j], x[j].x, ar[j]), yi = fma(dd[j], x[j].y, ai[j]);
#pragma unroll
      for (int k = 0; k < Q; k++) {
        if (S.g1off[k] == 0.0) continue;  // wave-uniform
        const int st = S.post[k] * (N + 1);
        const double l1 = TRANS ? l1t[j][k] : l1f[j][k];
        const double2 xn = sx[TRANS ? max(it[j] - st, 0) : min(it[j] + st, top)];
        yr = fma(l1, xn.x, yr);
        yi = fma(l1, xn.y, yi);
      }
      y[j] = make_double2(yr, yi);
    }
  }

  // Gradient contraction on the matrix cores: A = [Im(Hc_k), z], B = [Re(Hc_k), z] for the thread's four elements (DenseStencil::
  // ladder element by element: 2 N products per element and oscillator).  z's operands of the eight K-slabs are read from the
  // published vector once and serve every oscillator; Hc_k is real: 8 instead of 16 products per slab and commutator pair.
  struct ZOps {
    double2 pb[4], pa[4];
  };
  __device__ __forceinline__ void ladder_fetch(const double2* __restrict__ sx, ZOps& z) const {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      z.pb[s] = sx[lo * N + 4 * s + hi];
      z.pa[s] = sx[(4 * s + hi) * N + lo];
    }
  }
  __device__ __forceinline__ void ladder_whole(const DevSys& S, const ZOps& z, int k, double2 (&Av)[EPT], double2 (&Bv)[EPT]) const {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
    const double* __restrict__ hr = S.hcr + (size_t)k * N * N;
    const double* __restrict__ hm = S.hci + (size_t)k * N * N;
    mfma_d4 Ar = {0.0, 0.0, 0.0, 0.0}, Ai = Ar, Br = Ar, Bi = Ar;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int il = lo * N + 4 * s + hi, ir = (4 * s + hi) * N + lo;
      const double hmL = hm[il], hrL = hr[il], hmR = hm[ir], hrR = hr[ir];
      Ar = __builtin_amdgcn_mfma_f64_16x16x4f64(hmL, z.pb[s].x, Ar, 0, 0, 0);
      Ai = __builtin_amdgcn_mfma_f64_16x16x4f64(hmL, z.pb[s].y, Ai, 0, 0, 0);
      Br = __builtin_amdgcn_mfma_f64_16x16x4f64(hrL, z.pb[s].x, Br, 0, 0, 0);
      Bi = __builtin_amdgcn_mfma_f64_16x16x4f64(hrL, z.pb[s].y, Bi, 0, 0, 0);
      Ar = __builtin_amdgcn_mfma_f64_16x16x4f64(-z.pa[s].x, hmR, Ar, 0, 0, 0);
      Ai = __builtin_amdgcn_mfma_f64_16x16x4f64(-z.pa[s].y, hmR, Ai, 0, 0, 0);
      Br = __builtin_amdgcn_mfma_f64_16x16x4f64(-z.pa[s].x, hrR, Br, 0, 0, 0);
      Bi = __builtin_amdgcn_mfma_f64_16x16x4f64(-z.pa[s].y, hrR, Bi, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      Av[j] = make_double2(Ar[j], Ai[j]);
      Bv[j] = make_double2(Br[j], Bi[j]);
    }
  }
};

// The same for 17 <= N <= 32 (dim up to 1024; N < 32 zero-padded): four waves, wave w owns the 16 x 16 tile (row tile w & 1, column tile w >> 1) of rho in the
// accumulator layout; eight K-slabs per product, G's row block / column block of the tile in registers for the sub-step (both
// orientations: the transposed real operator needs G^H), rho's operands from the published LDS vector (16 reads per lane and
// application against 128 per element of the vector formulation): 64 MFMA instructions per wave and application.
template <int Q>
struct DenseMfma32Stencil : DenseStencil<Q, true, 4, 4> {
  typedef DenseStencil<Q, true, 4, 4> Base;
  static constexpr bool WHOLE = true;
  static constexpr int EPT = 4, NS = 8;  // 32 x 32 tiles; a smaller rho (N >= 17) is zero-padded: operands outside read as 0
  int N;
  using Base::dbra;
  using Base::dd;
  using Base::dig;
  using Base::dket;
  using Base::it;
  using Base::ofs;
  using Base::valid;
  int r0, c0;              // first row / column of this wave's tile
  double2 gA[NS], gB[NS];    // G[r0 + lo][4 s + hi], G[4 s + hi][c0 + lo]
  double2 gAh[NS], gBh[NS];  // G^H at the same places: conj(G[4 s + hi][r0 + lo]), conj(G[c0 + lo][4 s + hi])
  double l1f[EPT][Q], l1t[EPT][Q];

  __device__ __forceinline__ void init(const DevSys& S, const Lds& L) {
    Base::init(S, L);
    N = S.N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    r0 = 16 * (w & 1);
    c0 = 16 * (w >> 1);
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int Iraw = r0 + (lane >> 4) + 4 * j, Ipraw = c0 + (lane & 15);
      const int I = min(Iraw, N - 1), Ip = min(Ipraw, N - 1);  // (padding slots compute on a clamped element and are never stored)
      it[j] = Ip * N + I;
      valid[j] = Iraw < N && Ipraw < N;
      int ia[Q], ipa[Q];
      dbra[j] = 0;
      dket[j] = 0;
      double d = 0.0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        ia[k] = (I / S.post[k]) % S.n[k];
        ipa[k] = (Ip / S.post[k]) % S.n[k];
        dbra[j] |= (unsigned)ia[k] << (Base::DB * k);
        dket[j] |= (unsigned)ipa[k] << (Base::DB * k);
        d += S.g2[k] * (ia[k] * ipa[k] - 0.5 * (ia[k] * ia[k] + ipa[k] * ipa[k])) - S.g1[k] / 2.0 * (ia[k] + ipa[k]);
        const bool up = ia[k] < S.n[k] - 1 && ipa[k] < S.n[k] - 1;
        l1f[j][k] = up ? S.g1off[k] * sqrt((double)(ia[k] + 1)) * sqrt((double)(ipa[k] + 1)) : 0.0;
        l1t[j][k] = S.g1off[k] * sqrt((double)ia[k]) * sqrt((double)ipa[k]);
      }
      dd[j] = d;
    }
  }

  __device__ __forceinline__ void prep(const DevSys&, const Lds&, const StepC<Q>& c) {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
    const double2 zero = make_double2(0.0, 0.0);
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int kk = 4 * s + hi, kc = min(kk, N - 1), rr = min(r0 + lo, N - 1), cc = min(c0 + lo, N - 1);
      const bool inr = kk < N && r0 + lo < N, inc = kk < N && c0 + lo < N;
      gA[s] = inr ? c.g[rr * N + kc] : zero;
      gB[s] = inc ? c.g[kc * N + cc] : zero;
      const double2 a = c.g[kc * N + rr], b = c.g[cc * N + kc];
      gAh[s] = inr ? make_double2(a.x, -a.y) : zero;
      gBh[s] = inc ? make_double2(b.x, -b.y) : zero;
    }
  }

  template <bool TRANS>
  __device__ __forceinline__ void apply_whole(const DevSys& S, const Lds& L, const double2* __restrict__ sx, const StepC<Q>&,
                                              const double2 (&x)[EPT], double2 (&y)[EPT]) const {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
    mfma_d4 ar = {0.0, 0.0, 0.0, 0.0}, ai = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NS; s++) {
      // first product: Gt rho; A operand Gt[r0 + lo][4 s + hi] from registers, B operand rho[4 s + hi][c0 + lo] from LDS
      if (4 * s >= N) break;  // (uniform: slabs beyond a zero-padded rho contribute nothing)
      const int kk = 4 * s + hi, kc = min(kk, N - 1);
      const double2 ga = TRANS ? gAh[s] : gA[s];
      double2 pb = sx[min(c0 + lo, N - 1) * N + kc];
      if (!(kk < N && c0 + lo < N)) pb = make_double2(0.0, 0.0);
      ar = __builtin_amdgcn_mfma_f64_16x16x4f64(ga.x, pb.x, ar, 0, 0, 0);
      ar = __builtin_amdgcn_mfma_f64_16x16x4f64(-ga.y, pb.y, ar, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f64_16x16x4f64(ga.x, pb.y, ai, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f64_16x16x4f64(ga.y, pb.x, ai, 0, 0, 0);
      // second product: - rho Gt; A operand rho[r0 + lo][4 s + hi] from LDS, B operand Gt[4 s + hi][c0 + lo] from registers
      double2 pa = sx[kc * N + min(r0 + lo, N - 1)];
      if (!(kk < N && r0 + lo < N)) pa = make_double2(0.0, 0.0);
      const double2 gb = TRANS ? gBh[s] : gB[s];
      ar = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa.x, gb.x, ar, 0, 0, 0);
      ar = __builtin_amdgcn_mfma_f64_16x16x4f64(pa.y, gb.y, ar, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa.x, gb.y, ai, 0, 0, 0);
      ai = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa.y, gb.x, ai, 0, 0, 0);
    }
    const int top = S.dim - 1;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      double yr = fma(dd[j], x[j].x, ar[j]), yi = fma(dd[j], x[j].y, ai[j]);
#pragma unroll
      for (int k = 0; k < Q; k++) {
        if (S.g1off[k] == 0.0) continue;  // wave-uniform
        const int st = S.post[k] * (N + 1);
        const double l1 = TRANS ? l1t[j][k] : l1f[j][k];
        const double2 xn = sx[TRANS ? max(it[j] - st, 0) : min(it[j] + st, top)];
        yr = fma(l1, xn.x, yr);
        yi = fma(l1, xn.y, yi);
      }
      y[j] = make_double2(yr, yi);
    }
  }

  // Gradient contraction on the matrix cores: A = [Im(Hc_k), z], B = [Re(Hc_k), z] for the thread's four elements (DenseStencil::
  // ladder element by element: 2 N products per element and oscillator).  z's operands of the eight K-slabs are read from the
  // published vector once and serve every oscillator; Hc_k is real: 8 instead of 16 products per slab and commutator pair.
  struct ZOps {
    double2 pb[NS], pa[NS];
  };
  __device__ __forceinline__ void ladder_fetch(const double2* __restrict__ sx, ZOps& z) const {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int kk = 4 * s + hi, kc = min(kk, N - 1);
      z.pb[s] = sx[min(c0 + lo, N - 1) * N + kc];
      z.pa[s] = sx[kc * N + min(r0 + lo, N - 1)];
      if (!(kk < N && c0 + lo < N)) z.pb[s] = make_double2(0.0, 0.0);
      if (!(kk < N && r0 + lo < N)) z.pa[s] = make_double2(0.0, 0.0);
    }
  }
  __device__ __forceinline__ void ladder_whole(const DevSys& S, const ZOps& z, int k, double2 (&Av)[EPT], double2 (&Bv)[EPT]) const {
    const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
    const double* __restrict__ hr = S.hcr + (size_t)k * N * N;
    const double* __restrict__ hm = S.hci + (size_t)k * N * N;
    mfma_d4 Ar = {0.0, 0.0, 0.0, 0.0}, Ai = Ar, Br = Ar, Bi = Ar;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      if (4 * s >= N) break;
      const int kk = 4 * s + hi, kc = min(kk, N - 1);
      const int il = min(r0 + lo, N - 1) * N + kc, ir = kc * N + min(c0 + lo, N - 1);
      const bool inl = kk < N && r0 + lo < N, inr = kk < N && c0 + lo < N;
      const double hmL = inl ? hm[il] : 0.0, hrL = inl ? hr[il] : 0.0, hmR = inr ? hm[ir] : 0.0, hrR = inr ? hr[ir] : 0.0;
      Ar = __builtin_amdgcn_mfma_f64_16x16x4f64(hmL, z.pb[s].x, Ar, 0, 0, 0);
      Ai = __builtin_amdgcn_mfma_f64_16x16x4f64(hmL, z.pb[s].y, Ai, 0, 0, 0);
      Br = __builtin_amdgcn_mfma_f64_16x16x4f64(hrL, z.pb[s].x, Br, 0, 0, 0);
      Bi = __builtin_amdgcn_mfma_f64_16x16x4f64(hrL, z.pb[s].y, Bi, 0, 0, 0);
      Ar = __builtin_amdgcn_mfma_f64_16x16x4f64(-z.pa[s].x, hmR, Ar, 0, 0, 0);
      Ai = __builtin_amdgcn_mfma_f64_16x16x4f64(-z.pa[s].y, hmR, Ai, 0, 0, 0);
      Br = __builtin_amdgcn_mfma_f64_16x16x4f64(-z.pa[s].x, hrR, Br, 0, 0, 0);
      Bi = __builtin_amdgcn_mfma_f64_16x16x4f64(-z.pa[s].y, hrR, Bi, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      Av[j] = make_double2(Ar[j], Ai[j]);
      Bv[j] = make_double2(Br[j], Bi[j]);
    }
  }
};

template <int Q, bool LIND, int EPT, int EPE, bool QUBIT, bool COL = false, bool DENSE = false, bool PACKED = false, bool MFMA = false>
struct StencilSel { typedef GenStencil<Q, LIND, EPT, EPE> type; };
template <int Q, bool LIND, int EPT, int EPE>
struct StencilSel<Q, LIND, EPT, EPE, true, false, false, false, false> { typedef QubitStencil<Q, LIND, EPT, EPE> type; };
template <int Q, int EPT, int EPE, bool PACKED>
struct StencilSel<Q, true, EPT, EPE, false, true, false, PACKED, false> { typedef ColStencil<Q, EPT, PACKED> type; };
template <int Q>
struct StencilSel<Q, true, 4, 4, true, false, false, false, false> { typedef QubitSlotStencil<Q> type; };
template <int Q, bool LIND, int EPT, int EPE>
struct StencilSel<Q, LIND, EPT, EPE, false, false, true, false, false> { typedef DenseStencil<Q, LIND, EPT, EPE> type; };
template <int Q>
struct StencilSel<Q, true, 4, 4, false, false, true, false, true> { typedef DenseMfmaStencil<Q> type; };
template <int Q>
struct StencilSel<Q, true, 4, 4, false, false, true, true, true> { typedef DenseMfma32Stencil<Q> type; };

template <typename ST, typename = void> struct has_split_fetch { static constexpr bool value = false; };
template <typename ST> struct has_split_fetch<ST, decltype((void)ST::SPLIT_FETCH)> { static constexpr bool value = ST::SPLIT_FETCH; };

template <typename ST> __device__ __forceinline__ bool slot_valid(const ST& st, int j);
template <int Q, bool LIND, int EPT, int EPE>
__device__ __forceinline__ bool slot_valid(const GenStencil<Q, LIND, EPT, EPE>& st, int j) { return st.valid[j]; }
template <int Q, bool LIND, int EPT, int EPE>
__device__ __forceinline__ bool slot_valid(const QubitStencil<Q, LIND, EPT, EPE>& st, int j) { return st.valid[j]; }
template <int Q, int EPT, bool PACKED>
__device__ __forceinline__ bool slot_valid(const ColStencil<Q, EPT, PACKED>& st, int j) { return st.valid[j]; }
template <int Q>
__device__ __forceinline__ bool slot_valid(const QubitSlotStencil<Q>& st, int j) { return st.valid[j]; }
template <int Q, bool LIND, int EPT, int EPE>
__device__ __forceinline__ bool slot_valid(const DenseStencil<Q, LIND, EPT, EPE>& st, int j) { return st.valid[j]; }
template <int Q>
__device__ __forceinline__ bool slot_valid(const DenseMfmaStencil<Q>& st, int j) { return st.valid[j]; }
template <int Q>
__device__ __forceinline__ bool slot_valid(const DenseMfma32Stencil<Q>& st, int j) { return st.valid[j]; }

// ---------------------------------------------------------------------------------------------
// objective pieces evaluated on register-resident states (OptimTarget::evalJ / evalJ_diff)
// ---------------------------------------------------------------------------------------------
// Thread-local part of (J_re, J_im) for one element (optimtarget.cpp:712-799).
template <bool LIND>
__device__ __forceinline__ void evalJ_part(const DevSys& S, const DevTarget& tg, int b, int it, const double2 x, double& jre,
                                           double& jim) {
  const int dim = S.dim;
  switch (tg.objective_type) {
    case QD_OBJ_JFROBENIUS: {
      double tr, ti;
      if (tg.target_type != QD_TARGET_PURE) {
        tr = tg.tstates[(size_t)b * 2 * dim + it];
        ti = tg.tstates[(size_t)b * 2 * dim + dim + it];
      } else {
        tr = it == tg.idm ? 1.0 : 0.0;
        ti = 0.0;
      }
      const double dr = tr - x.x, di = ti - x.y;
      jre += 0.5 * (dr * dr + di * di);
      break;
    }
    case QD_OBJ_JTRACE: {
      const double pur = tg.purity[b];
      if (tg.target_type == QD_TARGET_PURE) {
        if (it == tg.idm) {
          jre += x.x / pur;
          jim += x.y;
        }
      } else {
        const double tr = tg.tstates[(size_t)b * 2 * dim + it], ti = tg.tstates[(size_t)b * 2 * dim + dim + it];
        jre += (tr * x.x + ti * x.y) / pur;
        if (!LIND) jim += -ti * x.x + tr * x.y;
      }
      break;
    }
    case QD_OBJ_JMEASURE: {
      if (LIND) {
        const int I = it % S.N, Ip = it / S.N;
        if (I == Ip) jre += fabs((double)(I - tg.purestate_id)) * x.x;
      } else {
        jre += fabs((double)(it - tg.purestate_id)) * (x.x * x.x + x.y * x.y);
      }
      break;
    }
  }
}

// HilbertSchmidtOverlap(state, false) thread-local part (optimtarget.cpp:343-408)
template <bool LIND>
__device__ __forceinline__ void fidelity_part(const DevSys& S, const DevTarget& tg, int b, int it, const double2 x, double& fre,
                                              double& fim) {
  const int dim = S.dim;
  if (tg.target_type == QD_TARGET_PURE) {
    if (it == tg.idm) {
      fre += x.x;
      fim += x.y;
    }
  } else {
    const double tr = tg.tstates[(size_t)b * 2 * dim + it], ti = tg.tstates[(size_t)b * 2 * dim + dim + it];
    fre += tr * x.x + ti * x.y;
    if (!LIND) fim += -ti * x.x + tr * x.y;
  }
}

// xbar += dJ/dx * (rbar, ibar) at one element (optimtarget.cpp:802-862, :410-447)
template <bool LIND>
__device__ __forceinline__ void evalJ_diff_elem(const DevSys& S, const DevTarget& tg, int b, int it, const double2 x, double2& xb,
                                                double rbar, double ibar) {
  const int dim = S.dim;
  switch (tg.objective_type) {
    case QD_OBJ_JFROBENIUS:
      if (tg.target_type != QD_TARGET_PURE) {
        const double tr = tg.tstates[(size_t)b * 2 * dim + it], ti = tg.tstates[(size_t)b * 2 * dim + dim + it];
        xb.x += rbar * (x.x - tr);
        xb.y += rbar * (x.y - ti);
      } else {
        xb.x += rbar * x.x;
        xb.y += rbar * x.y;
        if (it == tg.idm) xb.x -= rbar;
      }
      break;
    case QD_OBJ_JTRACE: {
      const double sc = 1.0 / tg.purity[b];
      if (tg.target_type == QD_TARGET_PURE) {
        if (it == tg.idm) {
          xb.x += rbar * sc;
          xb.y += ibar;
        }
      } else {
        const double tr = tg.tstates[(size_t)b * 2 * dim + it], ti = tg.tstates[(size_t)b * 2 * dim + dim + it];
        if (LIND) {
          xb.x += rbar * sc * tr;
          xb.y += rbar * sc * ti;
        } else {
          xb.x += tr * rbar * sc - ti * ibar;
          xb.y += ti * rbar * sc + tr * ibar;
        }
      }
      break;
    }
    case QD_OBJ_JMEASURE:
      if (LIND) {
        const int I = it % S.N, Ip = it / S.N;
        if (I == Ip) xb.x += fabs((double)(I - tg.purestate_id)) * rbar;
      } else {
        const double lam = fabs((double)(it - tg.purestate_id));
        xb.x += 2.0 * rbar * lam * x.x;
        xb.y += 2.0 * rbar * lam * x.y;
      }
      break;
  }
}

// finalizeJ / finalizeJ_diff (optimtarget.cpp:864-897)
template <bool LIND>
__device__ __forceinline__ double finalizeJ(const DevTarget& tg, double re, double im) {
  if (tg.objective_type == QD_OBJ_JTRACE) return LIND ? 1.0 - re : 1.0 - (re * re + im * im);
  return re;
}
template <bool LIND>
__device__ __forceinline__ void finalizeJ_diff(const DevTarget& tg, double re, double im, double& rb, double& ib) {
  if (tg.objective_type == QD_OBJ_JTRACE) {
    if (LIND) { rb = -1.0; ib = 0.0; } else { rb = -2.0 * re; ib = -2.0 * im; }
  } else {
    rb = 1.0;
    ib = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// the per-workgroup machinery shared by the forward, adjoint and apply kernels
// ---------------------------------------------------------------------------------------------
// GM: the GMRES code paths are compiled in (separate kernel instantiations, so that the Neumann kernels do
// not pay for them in registers and code size)
template <int Q, bool LIND, int VAR, bool QUBIT, bool GM = false>
struct Team {
  typedef Variant<VAR> V;
  static constexpr int EPT = V::EPT;    // slots per thread
  static constexpr int ICPB = V::ICPB;  // initial conditions per workgroup (interleaved in the same threads)
  static constexpr int EPE = EPT / ICPB;  // elements per thread of ONE initial condition
  typedef typename StencilSel<Q, LIND, EPT, EPE, QUBIT, V::COL, V::DENSE, V::PACKED, V::MFMA>::type ST;
  ST st;
  Lds L;
  int cur;      // which LDS buffer holds the vector that may be stencil-read
  int redslot;  // alternating reduction scratch slot
  int dim;
  int ic0;      // first initial condition of this workgroup
  int nb;       // batch size

  __device__ __forceinline__ void init(const DevSys& S, unsigned char* smem, int nbatch, int krylov = 0) {
    L = carve(smem, S, V::DBUF, V::BLDS, krylov, ICPB, V::COL, V::DENSE && S.dense == 2);
    st.init(S, L);
    cur = 0;
    redslot = 0;
    dim = S.dim;
    nb = nbatch;
    ic0 = blockIdx.x * ICPB;
  }
  // slot j belongs to initial condition ic(j); a workgroup past the end of the batch re-reads the last one
  __device__ __forceinline__ int icslot(int j) const { return j / EPE; }
  __device__ __forceinline__ bool icvalid(int s) const { return ICPB == 1 || ic0 + s < nb; }  // ICPB == 1: grid == batch
  __device__ __forceinline__ int ic(int j) const { return ICPB == 1 ? ic0 : min(ic0 + icslot(j), nb - 1); }
  __device__ __forceinline__ bool ok(int j) const { return slot_valid(st, j) && icvalid(icslot(j)); }
  __device__ __forceinline__ double2* bufp(int b) const { return L.buf0 + b * L.bstride; }
  __device__ __forceinline__ const double2* vec() const { return bufp(cur); }
  __device__ __forceinline__ const double2* vecj(int j) const { return bufp(cur) + icslot(j) * dim; }
  __device__ __forceinline__ int lidx(int j) const { return icslot(j) * dim + st.it[j]; }  // LDS index of slot j

  template <int NV>
  __device__ __forceinline__ void sum(double (&v)[NV]) {
    double* red = L.red + redslot * NRED * ((blockDim.x + 63) >> 6);
    block_sum<NV, V::ONEWAVE>(v, red);
    redslot ^= 1;
  }

  // Team-wide sums of NV values that are only STORED (block_sum_post / block_sum_collect): post, then a barrier of the caller, then collect
  double* pend;
  template <int NV, typename F>
  __device__ __forceinline__ void sum_store_post(const double (&v)[NV], F&& dst) {
    pend = L.red + redslot * NRED * ((blockDim.x + 63) >> 6);
    redslot ^= 1;
    block_sum_post<NV, V::ONEWAVE>(v, pend, dst);
  }
  template <int NV, typename F>
  __device__ __forceinline__ void sum_store_collect(F&& dst) const {
    block_sum_collect<NV, V::ONEWAVE>(pend, dst);
  }

  __device__ __forceinline__ float sum_f32(float v) {
    double* red = L.red + redslot * NRED * ((blockDim.x + 63) >> 6);
    redslot ^= 1;
    return block_sum_f32<V::ONEWAVE>(v, red);
  }

  template <int NV>
  __device__ __forceinline__ void sum_f32v(float (&v)[NV]) {
    double* red = L.red + redslot * NRED * ((blockDim.x + 63) >> 6);
    redslot ^= 1;
    block_sum_f32v<NV, V::ONEWAVE>(v, red);
  }

  // Make `x` the stencil-readable vector.  Single buffer: a barrier before the overwrite (every
  // thread finished reading the old content) and one after; double buffer: only the one after.
  __device__ __forceinline__ void publish(const double2 (&x)[EPT]) {
    if (!V::DBUF) team_sync<V::ONEWAVE>();
    const int nxt = V::DBUF ? cur ^ 1 : cur;
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (ok(j)) bufp(nxt)[lidx(j)] = x[j];
    cur = nxt;
    team_sync<V::ONEWAVE>();
  }

  // xall: the thread's elements of the vector being read (used by the stencils that take neighbours from registers)
  template <bool TRANS, bool HASJ>
  __device__ __forceinline__ double2 apply_slot(const DevSys& S, const double2* __restrict__ sx, const StepC<Q>& c, int j,
                                                const double2 (&xall)[EPT]) const {
    if constexpr (ST::NEEDS_SLOTS) return st.template apply<TRANS, HASJ>(S, L, sx, c, j, xall);
    else return st.template apply<TRANS>(S, L, sx, c, j, xall[j]);
  }

  template <bool TRANS, bool HASJ>
  __device__ __forceinline__ void apply_sweep(const DevSys& S, const StepC<Q>& c, const double2 (&x)[EPT], double2 (&y)[EPT]) const {
    if constexpr (ST::WHOLE) {
      st.template apply_whole<TRANS>(S, L, vec(), c, x, y);
      return;
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      y[j] = apply_slot<TRANS, HASJ>(S, vecj(j), c, j, x);
      if ((j % V::FENCE) == V::FENCE - 1) slot_fence_pin<EPE>(y[j]);
    }
  }
  template <bool TRANS>
  __device__ __forceinline__ void apply_all(const DevSys& S, const StepC<Q>& c, const double2 (&x)[EPT], double2 (&y)[EPT]) const {
    if (ST::NEEDS_SLOTS && !S.hasJ) apply_sweep<TRANS, false>(S, c, x, y);
    else apply_sweep<TRANS, true>(S, c, x, y);
  }

  // one Neumann update of every owned element: y <- b + alpha M^{(T)} y, squared update norm into dloc
  template <bool TRANS, bool HASJ>
  __device__ __forceinline__ void neumann_sweep(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2* __restrict__ src,
                                                const double2 (&b)[EPT], double2 (&y)[EPT], double (&dloc)[ICPB]) {
    double2 yold[EPT];  // the iterate being read (Jacobi update): y is overwritten slot by slot
#pragma unroll
    for (int j = 0; j < EPT; j++) yold[j] = y[j];
    double2 tall[ST::WHOLE ? EPT : 1];
    if constexpr (ST::WHOLE) st.template apply_whole<TRANS>(A.S, L, src, c, yold, tall);
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      double2 t;
      if constexpr (ST::WHOLE) t = tall[j];
      else t = apply_slot<TRANS, HASJ>(A.S, src + icslot(j) * dim, c, j, yold);
      const double2 bj = V::BLDS ? L.bvec[lidx(j)] : b[j];
      double2 w;
      w.x = fma(alpha, t.x, bj.x);
      w.y = fma(alpha, t.y, bj.y);
      const double dx = yold[j].x - w.x, dy = yold[j].y - w.y;
      dloc[icslot(j)] += ok(j) ? dx * dx + dy * dy : 0.0;
      y[j] = w;  // registers only; LDS still holds the old iterate for the other threads
      if (V::DBUF && ok(j)) bufp(cur)[lidx(j)] = w;
      if ((j % V::FENCE) == V::FENCE - 1) slot_fence_pin<EPE>(y[j]);
    }
  }

  // Solve (I - alpha M^{(T)}) y = b by the reference's Neumann iteration (timestepper.cpp:697-727).
  // On exit y is in registers AND is the published vector.  Returns the number of RHS applications.
  template <bool TRANS>
  __device__ __forceinline__ int neumann(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2 (&b)[EPT],
                                         double2 (&y)[EPT]) {
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      y[j] = b[j];
      if (V::BLDS && ok(j)) L.bvec[lidx(j)] = b[j];  // read back by the owning thread only: no barrier needed
    }
    publish(y);
    // Stopping test of the reference (timestepper.cpp:713-720) on squared norms: errnorm < abstol  <=>
    // d < abstol^2 and errnorm/errnorm0 < reltol  <=>  d < reltol^2 d0 (no fp64 sqrt / divide per
    // iteration).  The squared update norm is accumulated in fp64 per thread and reduced over the
    // workgroup in fp32: it is only compared with a threshold, and the fp32 reduction is a third of
    // the dependent latency of the fp64 one in the latency-bound small-system kernels.  Scaling by
    // 1/abstol^2 keeps the fp32 value away from the subnormal range.
    const double inv_abs2 = 1.0 / (A.abstol * A.abstol);
    const float rel2 = (float)(A.reltol * A.reltol);
    float d0 = 1.f, dprev = 1.f;
    int iter;
    if constexpr (V::ONEWAVE && !V::DBUF && EPT == 1 && ICPB == 1 && has_split_fetch<ST>::value) {
      // Single wave, one element per lane (the latency-bound small systems): software-pipelined iteration.  The
      // neighbour reads of iteration m+1 are issued right after y_{m+1} has been written to LDS and are in flight
      // while the update norm of iteration m is reduced and tested (readlane -> scalar compare -> branch).
      typename ST::Nbrs nb;
      st.fetch(vec(), 0, nb);
      for (iter = 0; iter < A.maxiter; iter++) {
        const double2 t = st.template apply_nb<TRANS>(A.S, vec(), c, 0, y[0], nb);
        double2 w;
        w.x = fma(alpha, t.x, b[0].x);
        w.y = fma(alpha, t.y, b[0].y);
        const double dx = y[0].x - w.x, dy = y[0].y - w.y;
        const double dl = ok(0) ? dx * dx + dy * dy : 0.0;
        y[0] = w;
        if (ok(0)) bufp(cur)[lidx(0)] = w;
        team_sync<true>();
        st.fetch(vec(), 0, nb);
        if (QD_ABLATE & 1) {  // no stopping test: four iterations per solve
          if (iter == 3) { iter++; break; }
          continue;
        }
        const float d = sum_f32((float)fmin(dl * inv_abs2, 1e30));
        // (one exit branch per iteration, first-iteration values by selects [r5]: qd_q32.hip / qd_col.hip measured 1 - 7 %)
        d0 = iter == 0 ? d : d0;
        {
          const float dp = iter == 0 ? d : dprev;
          const bool stop = (d < 1.f && standin_ok(A.standin_tau2, d, dp, 1.f)) | (d < rel2 * d0);
          dprev = d;
          if (stop) { iter++; break; }
        }
      }
      return iter;
    }
    for (iter = 0; iter < A.maxiter; iter++) {
      double dloc[ICPB];
#pragma unroll
      for (int q = 0; q < ICPB; q++) dloc[q] = 0.0;
      const double2* src = vec();
      if (V::DBUF) cur ^= 1;  // the new iterate goes to the other buffer: ONE barrier (inside the reduction)
      if (ST::NEEDS_SLOTS && !A.S.hasJ) neumann_sweep<TRANS, false>(A, c, alpha, src, b, y, dloc);
      else neumann_sweep<TRANS, true>(A, c, alpha, src, b, y, dloc);
      // clamp: adjoint solves of badly scaled problems have update norms whose square overflows fp32; a
      // clamped value is still far above both thresholds (the reference's reltol is 1e-20).  With several
      // initial conditions per workgroup all of them iterate until the slowest has converged (the others
      // only get more accurate).
      float d = 0.f;
      if (QD_ABLATE & 1) {  // no stopping test: four iterations per solve; the barrier that publishes the iterate stays
        team_sync<V::ONEWAVE>();
        if (!V::DBUF) {
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (ok(j)) bufp(cur)[lidx(j)] = y[j];
          team_sync<V::ONEWAVE>();
        }
        if (iter == 3) { iter++; break; }
        continue;
      }
      if (ICPB == 1) {
        d = sum_f32((float)fmin(dloc[0] * inv_abs2, 1e30));  // contains the barrier (multi-wave)
      } else {
        float dq[ICPB];
#pragma unroll
        for (int q = 0; q < ICPB; q++) dq[q] = (float)fmin(dloc[q] * inv_abs2, 1e30);
        sum_f32v<ICPB>(dq);
#pragma unroll
        for (int q = 0; q < ICPB; q++) d = fmaxf(d, dq[q]);
      }
      if (!V::DBUF) {
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (ok(j)) bufp(cur)[lidx(j)] = y[j];
        team_sync<V::ONEWAVE>();
      }
      // (one exit branch per iteration, first-iteration values by selects [r5]: qd_q32.hip / qd_col.hip measured 1 - 7 %)
      d0 = iter == 0 ? d : d0;
      {
        const float dp = iter == 0 ? d : dprev;
        const bool stop = (d < 1.f && standin_ok(A.standin_tau2, d, dp, 1.f)) | (d < rel2 * d0);
        dprev = d;
        if (stop) { iter++; break; }
      }
    }
    return iter;
  }

  // GMRES for (I - alpha M^{(T)}) y = b: stands in for KSPSolve / KSPSolveTranspose with KSPGMRES +
  // PCNONE (src/timestepper.cpp:541-550, call sites :602,:652,:674): zero initial guess, classical
  // Gram-Schmidt (PETSc's default orthogonalisation; the CPU oracle restates Saad-Schultz with the modified
  // variant - same method in exact arithmetic, the tests pin the application counts against it), Givens rotations, stop when the recurrence
  // residual <= max(rtol ||b||, abstol) or after maxiter iterations; restarted every GMRES_MR
  // iterations.  One element per thread: the Krylov basis lives in LDS (the stencil reads v_j in
  // place), the small Hessenberg problem is solved redundantly by every thread on wave-uniform
  // scalars kept in LDS.  Returns the number of RHS applications.
  template <bool TRANS>
  __device__ __forceinline__ int gmres(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2 (&b)[EPT], double2 (&y)[EPT]) {
    static_assert(EPT == 1, "in-kernel GMRES is built for the one-element-per-thread variants");
    const int dim = A.S.dim, e = st.it[0];
    const bool on = ok(0);
    double2* Vb = L.kry;
    double* hc = L.ksc;               // [MR+2] current Hessenberg column
    double* cs = hc + (GMRES_MR + 2);  // [MR]
    double* sn = cs + GMRES_MR;        // [MR]
    double* g = sn + GMRES_MR;         // [MR+2]
    double* R = g + (GMRES_MR + 2);    // [MR][MR] row-major upper triangle
    double* yk = R + GMRES_MR * GMRES_MR;
    double2 yy = make_double2(0.0, 0.0), r = b[0];
    int its = 0, napp = 0;
    double ttol = 0.0;
    for (int cycle = 0;; cycle++) {
      double t[1] = {on ? r.x * r.x + r.y * r.y : 0.0};
      sum<1>(t);
      const double ibeta = t[0] > 0.0 ? rsqrt_nr(t[0]) : 0.0;
      const double beta = t[0] * ibeta;
      if (cycle == 0) ttol = fmax(A.reltol * beta, A.abstol);
      if (beta <= ttol || its >= A.maxiter) break;
      double2 v = make_double2(r.x * ibeta, r.y * ibeta);
      if (on) Vb[e] = v;
      double gcur = beta;  // last entry of the rotated right-hand side
      team_sync<V::ONEWAVE>();
      int j = 0;
      bool conv = false;
      while (j < GMRES_MR) {
        const double2 t2 = st.template apply<TRANS>(A.S, L, Vb + (size_t)j * dim, c, 0, v);
        napp++;
        const double2 w0 = make_double2(v.x - alpha * t2.x, v.y - alpha * t2.y);
        double2 w = w0;
        // classical Gram-Schmidt: all projections against the un-updated w0, four per reduction; the basis
        // vectors and the reduced coefficients stay in registers for the subtraction
        for (int k0 = 0; k0 <= j; k0 += 4) {
          double h4[4];
          double2 vk4[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int k = k0 + q;
            vk4[q] = make_double2(0.0, 0.0);
            if (k <= j) vk4[q] = Vb[(size_t)k * dim + e];
            h4[q] = (on && k <= j) ? w0.x * vk4[q].x + w0.y * vk4[q].y : 0.0;
          }
          // as many values as there are projections in this block (the first iterations have 1, 2, 3)
          switch (min(4, j + 1 - k0)) {
            case 1: sum<1>(reinterpret_cast<double(&)[1]>(h4)); break;
            case 2: sum<2>(reinterpret_cast<double(&)[2]>(h4)); break;
            case 3: sum<3>(reinterpret_cast<double(&)[3]>(h4)); break;
            default: sum<4>(h4); break;
          }
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (k0 + q <= j) {
              hc[k0 + q] = h4[q];  // for the Givens stage
              w.x -= h4[q] * vk4[q].x;
              w.y -= h4[q] * vk4[q].y;
            }
        }
        double nn[1] = {on ? w.x * w.x + w.y * w.y : 0.0};
        sum<1>(nn);
        const double ihn = nn[0] > 0.0 ? rsqrt_nr(nn[0]) : 0.0;
        const double hn = nn[0] * ihn;
        hc[j + 1] = hn;
        // Givens rotations on the new column, update of the rotated right-hand side.  Every thread does
        // this redundantly on wave-uniform values; LDS locations are only ever written with values that
        // do not depend on an earlier write of the same phase (no read-modify-write), so waves of one
        // workgroup cannot observe each other's partial updates.
        double cur_h = hc[0];
        for (int k = 0; k < j; k++) {
          const double a1 = hc[k + 1], ck = cs[k], sk = sn[k];
          R[k * GMRES_MR + j] = ck * cur_h + sk * a1;
          cur_h = -sk * cur_h + ck * a1;
        }
        const double a = cur_h, bb = hn;
        const double s2 = a * a + bb * bb;
        const double irr = s2 > 0.0 ? rsqrt_nr(s2) : 0.0;
        const double cj = s2 > 0.0 ? a * irr : 1.0, sj = bb * irr;
        cs[j] = cj;
        sn[j] = sj;
        R[j * GMRES_MR + j] = irr;  // the diagonal is only ever divided by: keep its reciprocal
        g[j] = cj * gcur;
        gcur = -sj * gcur;
        its++;
        j++;
        if (fabs(gcur) <= ttol || hn == 0.0) { conv = true; break; }
        if (its >= A.maxiter || j >= GMRES_MR) break;
        // the next basis vector is only formed and stored when another iteration follows
        v = make_double2(w.x * ihn, w.y * ihn);
        if (on) Vb[(size_t)j * dim + e] = v;
        team_sync<V::ONEWAVE>();  // v_{j} is readable by every thread
      }
      // back substitution R yk = g, y += V yk
      for (int rw = j - 1; rw >= 0; rw--) {
        double sacc = g[rw];
        for (int cc = rw + 1; cc < j; cc++) sacc -= R[rw * GMRES_MR + cc] * yk[cc];
        yk[rw] = sacc * R[rw * GMRES_MR + rw];
      }
      for (int cc = 0; cc < j; cc++) {
        const double2 vk = Vb[(size_t)cc * dim + e];
        const double f = yk[cc];
        yy.x += f * vk.x;
        yy.y += f * vk.y;
      }
      if (conv || its >= A.maxiter) break;
      // restart: r = b - (I - alpha M) y
      team_sync<V::ONEWAVE>();
      if (on) Vb[e] = yy;
      team_sync<V::ONEWAVE>();
      const double2 t2 = st.template apply<TRANS>(A.S, L, Vb, c, 0, yy);
      napp++;
      r = make_double2(b[0].x - (yy.x - alpha * t2.x), b[0].y - (yy.y - alpha * t2.y));
      team_sync<V::ONEWAVE>();
    }
    y[0] = yy;
    return napp;
  }

  // GMRES for any number of elements per thread and any dimension: the Krylov basis lives in global memory (A.kry,
  // [nb][GMRES_MR_G + 2][dim] interleaved complex; every thread only ever touches its own elements of the basis
  // vectors), the vector the stencil reads is published in LDS like a Neumann iterate.  On exit y is in registers (NOT
  // published).
  //
  // A.gmres_poly = p > 1: right preconditioning with the Neumann polynomial P = sum_{i<p} (alpha M)^i, i.e. GMRES on
  // (I - alpha M) P = I - (alpha M)^p, y = P u.  The residual of the preconditioned system IS the true residual
  // b - (I - alpha M) y, so the stopping rule max(rtol ||b||, abstol) of the reference is unchanged; what changes is the
  // path: p applications per Krylov vector, but only ~ (iterations of plain GMRES) / p Krylov vectors.  With the basis in
  // HBM the cost of plain GMRES is its m^2 + 3m passes over 2 dim doubles per step (3x20 Lindblad: m = 11, 9.7 MB per
  // step and initial condition); p = 6 needs m = 2.  The preconditioned vectors z_j = P v_j (Horner: z <- v_j + alpha M z, p - 1
  // applications; w = z - alpha M z is the p-th) are kept next to the basis, so the solution y = sum_j yk_j z_j costs no
  // further application (restart 14 then: 15 + 14 vectors + the parked total fit the 32 slots of the plain basis).  The host only asks for p > 1 where the Neumann series provably
  // contracts (a Gershgorin bound of ||alpha M||_inf <= 0.7 from the system constants and the current control
  // parameters, qd_handle::gmres_poly_degree); otherwise, and with p = 1, this is KSPGMRES + PCNONE iteration for iteration.
  template <bool TRANS>
  __device__ __forceinline__ int gmres_g(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2 (&b)[EPT], double2 (&y)[EPT]) {
    static_assert(ICPB == 1, "one initial condition per workgroup");
    double2* __restrict__ Vg = reinterpret_cast<double2*>(A.kry) + (size_t)ic0 * (GMRES_MR_G + 2) * dim;
    double* hc = L.ksc;
    double* cs = hc + (GMRES_MR_G + 2);
    double* sn = cs + GMRES_MR_G;
    double* g = sn + GMRES_MR_G;
    double* R = g + (GMRES_MR_G + 2);
    double* yk = R + GMRES_MR_G * GMRES_MR_G;
    const int poly = A.gmres_poly > 1 ? A.gmres_poly : 1;
    const int MRE = poly > 1 ? (GMRES_MR_G - 2) / 2 : GMRES_MR_G;         // restart length
    double2* __restrict__ Zg = Vg + (size_t)(MRE + 1) * dim;               // z_j = P v_j (poly > 1)
    const double2* __restrict__ Sg = poly > 1 ? Zg : Vg;                   // the vectors the solution is a combination of
    // The right-hand side is parked in the one basis slot neither mode uses (v_k, z_k stop at GMRES_MR_G - 1) and v_0 is stored like every
    // other basis vector [r5]: b then is not live through the operator applications - one 8-element array (32 registers) less in a kernel
    // that spills; it is read back for the residual of a restart only.
    double2* __restrict__ Bg = Vg + (size_t)GMRES_MR_G * dim;
    double2 yy[EPT], r[EPT], v[EPT], w[EPT];
    int napp = 0;
    // t2 <- M z, reading z from the published vector
    auto mapply = [&](const double2(&z)[EPT], double2(&t2)[EPT]) {
      if (ST::NEEDS_SLOTS && !A.S.hasJ) apply_sweep<TRANS, false>(A.S, c, z, t2);
      else apply_sweep<TRANS, true>(A.S, c, z, t2);
      napp++;
    };
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      yy[j] = make_double2(0.0, 0.0);
      r[j] = b[j];
      if (ok(j)) Bg[at_use<EPE>(st.it[j])] = b[j];
    }
    int its = 0;
    bool have_total = false;  // a restart has parked the accumulated solution in basis slot GMRES_MR_G + 1
    double ttol = 0.0;
    for (int cycle = 0;; cycle++) {
      double t1[1] = {0.0};
#pragma unroll
      for (int j = 0; j < EPT; j++) t1[0] += ok(j) ? r[j].x * r[j].x + r[j].y * r[j].y : 0.0;
      sum<1>(t1);
      const double ibeta = t1[0] > 0.0 ? rsqrt_nr(t1[0]) : 0.0;
      const double beta = t1[0] * ibeta;
      if (cycle == 0) ttol = fmax(A.reltol * beta, A.abstol);
      if (beta <= ttol || its >= A.maxiter) {
#pragma unroll
        for (int j = 0; j < EPT; j++) yy[j] = make_double2(0.0, 0.0);
        break;
      }
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        v[j] = make_double2(r[j].x * ibeta, r[j].y * ibeta);
        if (ok(j)) Vg[at_use<EPE>(st.it[j])] = v[j];
      }
      publish(v);
      double gcur = beta;
      int jj = 0;
      bool conv = false;
      while (jj < MRE) {
        // z = P v_jj by Horner's rule (the published vector is v_jj on entry), w = (I - alpha M) z = (I - (alpha M)^p) v_jj
#pragma unroll
        for (int j = 0; j < EPT; j++) w[j] = v[j];
        for (int i = 1; i < poly; i++) {
          if (i > 1) publish(w);
          double2 t2[EPT];
          mapply(w, t2);
#pragma unroll
          for (int j = 0; j < EPT; j++) w[j] = make_double2(fma(alpha, t2[j].x, v[j].x), fma(alpha, t2[j].y, v[j].y));
        }
        if (poly > 1) {
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (ok(j)) Zg[(size_t)jj * dim + at_use<EPE>(st.it[j])] = w[j];
          publish(w);
        }
        {
          double2 t2[EPT];
          mapply(w, t2);
#pragma unroll
          for (int j = 0; j < EPT; j++) w[j] = make_double2(fma(-alpha, t2[j].x, w[j].x), fma(-alpha, t2[j].y, w[j].y));
        }
        // classical Gram-Schmidt: all projections against the un-updated w, four per block reduction.  (The squared norm of
        // the orthogonalised vector is NOT taken from ||w||^2 - sum h_k^2: classical Gram-Schmidt loses orthogonality as
        // the residual falls towards 1e-10 and that identity then misjudges h_{j+1,j} - measured: the recurrence residual
        // stops tracking the true one and every solve runs to maxiter.)
        // Positions of a block: 0 = v_jj (still in registers), 1 = v_0, then
        // v_1 .. v_{jj-1} read back from the basis in one branch-free run of loads.
        auto v0elem = [&](int j) { return Vg[at_use<EPE>(st.it[j])]; };
        for (int p0 = 0; p0 <= jj; p0 += 4) {
          double h4[4] = {0.0, 0.0, 0.0, 0.0};
          const int np = min(4, jj + 1 - p0);
          int q0 = 0;
          if (p0 == 0) {
#pragma unroll
            for (int j = 0; j < EPT; j++)
              if (ok(j)) h4[0] += w[j].x * v[j].x + w[j].y * v[j].y;
            q0 = 1;
            if (jj >= 1) {
#pragma unroll
              for (int j = 0; j < EPT; j++)
                if (ok(j)) {
                  const double2 vk = v0elem(j);
                  h4[1] += w[j].x * vk.x + w[j].y * vk.y;
                }
              q0 = 2;
            }
          }
          for (int q = q0; q < np; q++) {
#pragma unroll
            for (int j = 0; j < EPT; j++)
              if (ok(j)) {
                const double2 vk = Vg[(size_t)(p0 + q - 1) * dim + at_use<EPE>(st.it[j])];
                h4[q] += w[j].x * vk.x + w[j].y * vk.y;
              }
          }
          switch (np) {  // as many values as there are projections in this block (the first iterations have 1, 2, 3)
            case 1: sum<1>(reinterpret_cast<double(&)[1]>(h4)); break;
            case 2: sum<2>(reinterpret_cast<double(&)[2]>(h4)); break;
            case 3: sum<3>(reinterpret_cast<double(&)[3]>(h4)); break;
            default: sum<4>(h4); break;
          }
          for (int q = 0; q < np; q++) hc[p0 + q == 0 ? jj : p0 + q - 1] = h4[q];
        }
        {
          const double h = hc[jj];
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            w[j].x -= h * v[j].x;
            w[j].y -= h * v[j].y;
          }
        }
        if (jj >= 1) {
          const double h = hc[0];
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            const double2 vk = v0elem(j);
            w[j].x -= h * vk.x;
            w[j].y -= h * vk.y;
          }
        }
        for (int k = 1; k < jj; k++) {
          const double h = hc[k];
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            const double2 vk = Vg[(size_t)k * dim + at_use<EPE>(st.it[j])];
            w[j].x -= h * vk.x;
            w[j].y -= h * vk.y;
          }
        }
        double nn[1] = {0.0};
#pragma unroll
        for (int j = 0; j < EPT; j++) nn[0] += ok(j) ? w[j].x * w[j].x + w[j].y * w[j].y : 0.0;
        sum<1>(nn);
        const double ihn = nn[0] > 0.0 ? rsqrt_nr(nn[0]) : 0.0;
        const double hn = nn[0] * ihn;
        hc[jj + 1] = hn;
        // Givens rotations: redundantly by every thread on wave-uniform values, idempotent LDS writes only
        double cur_h = hc[0];
        for (int k = 0; k < jj; k++) {
          const double a1 = hc[k + 1], ck = cs[k], sk = sn[k];
          R[k * GMRES_MR_G + jj] = ck * cur_h + sk * a1;
          cur_h = -sk * cur_h + ck * a1;
        }
        const double a0 = cur_h, bb = hn;
        const double s2 = a0 * a0 + bb * bb;
        const double irr = s2 > 0.0 ? rsqrt_nr(s2) : 0.0;
        const double cj = s2 > 0.0 ? a0 * irr : 1.0, sj = bb * irr;
        cs[jj] = cj;
        sn[jj] = sj;
        R[jj * GMRES_MR_G + jj] = irr;  // the diagonal is only ever divided by: keep its reciprocal
        g[jj] = cj * gcur;
        gcur = -sj * gcur;
        its++;
        jj++;
        if (fabs(gcur) <= ttol || hn == 0.0) { conv = true; break; }
        if (its >= A.maxiter || jj >= MRE) break;
        // the next basis vector is only formed, stored and published when another iteration follows
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          v[j] = make_double2(w[j].x * ihn, w[j].y * ihn);
          if (ok(j)) Vg[(size_t)jj * dim + at_use<EPE>(st.it[j])] = v[j];
        }
        publish(v);  // v_{jj} becomes the stencil-readable vector; its barriers also order the scalar writes
      }
      for (int rw = jj - 1; rw >= 0; rw--) {
        double sacc = g[rw];
        for (int cc = rw + 1; cc < jj; cc++) sacc -= R[rw * GMRES_MR_G + cc] * yk[cc];
        yk[rw] = sacc * R[rw * GMRES_MR_G + rw];
      }
#pragma unroll
      for (int j = 0; j < EPT; j++) yy[j] = make_double2(0.0, 0.0);
      if (jj >= 1) {
        const double f = yk[0];
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const double2 vk = Sg[at_use<EPE>(st.it[j])];
          yy[j].x += f * vk.x;
          yy[j].y += f * vk.y;
        }
      }
      for (int cc = 1; cc < jj; cc++) {
        const double f = yk[cc];
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const double2 vk = Sg[(size_t)cc * dim + at_use<EPE>(st.it[j])];
          yy[j].x += f * vk.x;
          yy[j].y += f * vk.y;
        }
      }
      if (conv || its >= A.maxiter) break;
      // restart: park the accumulated solution, r = b - (I - alpha M) y_total
      double2* Yt = Vg + (size_t)(GMRES_MR_G + 1) * dim;
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        if (have_total) {
          const double2 o = Yt[at_use<EPE>(st.it[j])];
          yy[j].x += o.x;
          yy[j].y += o.y;
        }
        if (ok(j)) Yt[at_use<EPE>(st.it[j])] = yy[j];
      }
      have_total = true;
      publish(yy);
      double2 t3[EPT];
      if (ST::NEEDS_SLOTS && !A.S.hasJ) apply_sweep<TRANS, false>(A.S, c, yy, t3);
      else apply_sweep<TRANS, true>(A.S, c, yy, t3);
      napp++;
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const double2 bj = Bg[at_use<EPE>(st.it[j])];
        r[j] = make_double2(bj.x - (yy[j].x - alpha * t3[j].x), bj.y - (yy[j].y - alpha * t3[j].y));
      }
      team_sync<V::ONEWAVE>();  // every thread has read the scalars of this cycle before the next one overwrites them
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      y[j] = yy[j];
      if (have_total) {
        const double2 o = Vg[(size_t)(GMRES_MR_G + 1) * dim + at_use<EPE>(st.it[j])];
        y[j].x += o.x;
        y[j].y += o.y;
      }
    }
    return napp;
  }

  template <bool TRANS>
  __device__ __forceinline__ int solve(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2 (&b)[EPT], double2 (&y)[EPT]) {
    if constexpr (GM && EPT == 1) {
      if (A.use_gmres == 1) return gmres<TRANS>(A, c, alpha, b, y);
    }
    if constexpr (GM && ICPB == 1) {
      if (A.use_gmres == 2) return gmres_g<TRANS>(A, c, alpha, b, y);
    }
    return neumann<TRANS>(A, c, alpha, b, y);
  }
};

// ---------------------------------------------------------------------------------------------
// forward sweep: TimeStepper::solveODE for every initial condition of the batch
// ---------------------------------------------------------------------------------------------
// PLAIN: the sweep has no in-loop penalty (leakage, weighted J), no dpdm penalty and an implicit-midpoint stepper - known at launch
// (plain_sweep(), qd_internal.h).  The small-system kernels (one wave per initial condition) are bound by the NUMBER of instructions a
// single wave issues per step; with those branches, their registers (dpdm history, guard flags, penalty sums) and the scalar registers
// they pin compiled out, the 2^4 Schroedinger adjoint step drops from ~1900 static instructions (81 spilt scalar registers, reloaded by
// ~300 v_readlane per step) to the solve, the gradient contraction and one transposed application.
// All outstanding vector-memory operations have returned.  Placed in front of a time-step loop: the loop carries registers that were
// filled by global loads before it (initial state, first control row) and by arithmetic inside it; the compiler's wait-count pass merges
// the two and keeps `s_waitcnt vmcnt(0..1)` in front of the first use of the carried state in EVERY pass - directly behind the prefetch
// of the next control row, whose L2/HBM latency (~0.25 us of a ~1 us step of the one-wave kernels) was thereby exposed on every step.
// An explicit wait before the loop is seen by that pass and removes the in-loop one.
// The stored primal stages (SweepArgs::ztraj) are private to a forward / adjoint pair of kernels and never read by the host: interleaved
// (re, im) pairs, one 16-byte streaming access per element (the trajectory proper keeps the [u; v] blocks qd_get_state reads).
typedef double qd_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stage_store(double* ztraj, size_t state, int dim, int e, double re, double im) {
  const qd_d2v t = {re, im};
  __builtin_nontemporal_store(t, reinterpret_cast<qd_d2v*>(ztraj) + state * dim + e);
}
__device__ __forceinline__ double2 stage_load(const double* ztraj, size_t state, int dim, int e) {
  const qd_d2v t = __builtin_nontemporal_load(reinterpret_cast<const qd_d2v*>(ztraj) + state * dim + e);
  return make_double2(t.x, t.y);
}

__device__ __forceinline__ void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0), expcnt / lgkmcnt untouched

template <int Q, bool LIND, int VAR, bool QUBIT, bool GM, bool PLAIN = false>
__global__ void __launch_bounds__(Variant<VAR>::MAXB) k_forward(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef Team<Q, LIND, VAR, QUBIT, GM> TM;
  constexpr int EPT = TM::EPT, ICPB = TM::ICPB;
  const DevSys& S = A.S;
  TM tm;
  tm.init(S, smem, A.nb, A.use_gmres);
  const int dim = S.dim;
  double2 x[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    const double* x0 = A.x0 + (size_t)tm.ic(j) * 2 * dim;
    x[j] = make_double2(x0[tm.st.it[j]], x0[dim + tm.st.it[j]]);
  }
  team_sync<TM::V::ONEWAVE>();  // coefficient tables written by init()
  tm.publish(x);
  // penalty bookkeeping
  const bool ee = PLAIN ? false : A.stepper_ee != 0;
  const bool pen_on = PLAIN ? false : A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  // Schroedinger Jtrace is the only objective whose finalizeJ is nonlinear in the per-state sums:
  // it needs a block reduction per step; everything else accumulates thread-locally.
  const bool wj_reduce = wj_on && !LIND && A.tg.objective_type == QD_OBJ_JTRACE;
  const bool dpdm_on = PLAIN ? false : (A.gamma_dpdm > 1e-13 && !LIND);
  const bool jpairs = S.npairs > 0;
  bool guard[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++) guard[j] = A.leak_on && tm.ok(j) && tm.st.is_guard(S, j);
  double pen_local[ICPB], dpdm_local[ICPB], pen_uniform[ICPB];
#pragma unroll
  for (int q = 0; q < ICPB; q++) pen_local[q] = dpdm_local[q] = pen_uniform[q] = 0.0;
  double2 xm1[EPT], xm2[EPT];  // dpdm history (x_n, x_{n-1})
#pragma unroll
  for (int j = 0; j < EPT; j++) xm1[j] = xm2[j] = x[j];
  unsigned long long napply = 0;
  double* traj = A.traj;
  const double dtinv4 = 1.0 / (A.dt * A.dt * A.dt * A.dt);
  // latency-bound variants prefetch the next table row; the throughput variants (several waves per
  // SIMD) load the row where it is used and keep it in scalar registers
  constexpr bool PREFETCH = !TM::V::LEAN;
  constexpr bool XSTASH = TM::V::LEAN;  // explicit staging of the state through L2/HBM instead of compiler spills
  StepC<Q> c, cn;
  if (PREFETCH) load_step<Q>(A.ctl, cn, jpairs);

  vm_drain();
  for (int s = 0; s < A.nsub; s++) {
    if (PREFETCH) {
      c = cn;
      if (s + 1 < A.nsub) load_step<Q>(A.ctl + (size_t)(s + 1) * A.cs, cn, jpairs);  // prefetch the next row
    } else {
      load_step<Q>(A.ctl + (size_t)s * A.cs, c, jpairs);
      scalarize<Q>(c, jpairs);
    }
    c.g = S.dense ? reinterpret_cast<const double2*>(S.gtab) + (size_t)s * S.N * S.N : nullptr;
    tm.st.prep(S, tm.L, c);
    if (traj) {
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (tm.ok(j)) {
          double* dst = traj + ((size_t)s * A.nb + tm.ic(j)) * 2 * dim;  // written once, read by the adjoint sweep much later:
          __builtin_nontemporal_store(x[j].x, dst + tm.st.it[j]);        // streaming stores, the caches keep the solver's vectors
          __builtin_nontemporal_store(x[j].y, dst + dim + tm.st.it[j]);
        }
    }
    // rhs = M x   (ImplMidpoint::evolveFWD, timestepper.cpp:594; ExplEuler :502)
    double2 rhs[EPT];
    tm.template apply_all<false>(S, c, x, rhs);
    napply++;
    if (XSTASH && !ee) {  // x is not needed during the linear solve: park it in the output buffer
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (tm.ok(j)) {
          double* xs = A.xT + (size_t)tm.ic(j) * 2 * dim;
          const int e = at_use<TM::EPE>(tm.st.it[j]);
          xs[e] = x[j].x;
          xs[dim + e] = x[j].y;
        }
    }
    if (ee) {
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        x[j].x = fma(c.h, rhs[j].x, x[j].x);
        x[j].y = fma(c.h, rhs[j].y, x[j].y);
      }
    } else {
      double2 k[EPT];
      napply += tm.template solve<false>(A, c, 0.5 * c.h, rhs, k);
      if (XSTASH) {
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const double* xs = A.xT + (size_t)tm.ic(j) * 2 * dim;
          const int e = at_use<TM::EPE>(tm.st.it[j]);
          x[j] = make_double2(xs[e], xs[dim + e]);
        }
      }
      if (A.ztraj) {  // the primal stage z = x + h/2 k: read back by the adjoint sweep instead of repeating this solve
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (tm.ok(j)) stage_store(A.ztraj, (size_t)s * A.nb + tm.ic(j), dim, tm.st.it[j], fma(0.5 * c.h, k[j].x, x[j].x), fma(0.5 * c.h, k[j].y, x[j].y));
      }
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        x[j].x = fma(c.h, k[j].x, x[j].x);
        x[j].y = fma(c.h, k[j].y, x[j].y);
      }
    }
    tm.publish(x);

    // in-loop penalties, evaluated at the end of a FULL time step (timestepper.cpp:141-154)
    if ((pen_on || dpdm_on) && (s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages - 1;  // step index n: state is x_{n+1}
      const double tstop = (n + 1) * A.dt;
      if (pen_on) {
        double weight = 0.0;
        if (wj_on) {
          if (A.wjw) {  // tabulated per time step (qd_handle::ensure_wj_weights): no exp() next to the state registers
            weight = to_scalar(A.wjw[n]);
          } else {
            const double a = (tstop - A.Tfinal) / A.penalty_param;
            weight = 1.0 / A.penalty_param * exp(-(a * a));
          }
        }
        if (wj_reduce) {
          double v[2 * ICPB];
#pragma unroll
          for (int q = 0; q < 2 * ICPB; q++) v[q] = 0.0;
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.ok(j)) evalJ_part<LIND>(S, A.tg, tm.ic(j), at_use<TM::EPE>(tm.st.it[j]), x[j], v[2 * tm.icslot(j)], v[2 * tm.icslot(j) + 1]);
          tm.template sum<2 * ICPB>(v);
#pragma unroll
          for (int q = 0; q < ICPB; q++) pen_uniform[q] += weight * finalizeJ<LIND>(A.tg, v[2 * q], v[2 * q + 1]) * A.dt;
        } else if (wj_on) {
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.ok(j)) {
              double jr = 0.0, ji = 0.0;
              evalJ_part<LIND>(S, A.tg, tm.ic(j), at_use<TM::EPE>(tm.st.it[j]), x[j], jr, ji);
              // finalizeJ is affine here: J = jr (Jfrobenius, Jmeasure) or 1 - jr (Lindblad Jtrace)
              pen_local[tm.icslot(j)] += (A.tg.objective_type == QD_OBJ_JTRACE ? -1.0 : 1.0) * weight * A.dt * jr;
            }
          if (A.tg.objective_type == QD_OBJ_JTRACE) {
#pragma unroll
            for (int q = 0; q < ICPB; q++) pen_uniform[q] += weight * A.dt;
          }
        }
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (guard[j]) pen_local[tm.icslot(j)] += (x[j].x * x[j].x + x[j].y * x[j].y) / A.ntime;
      }
      if (dpdm_on) {
        if (n > 0) {
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.ok(j)) {
              const double t1 = x[j].x * x[j].x - 2.0 * xm1[j].x * xm1[j].x + xm2[j].x * xm2[j].x;
              const double t2 = x[j].y * x[j].y - 2.0 * xm1[j].y * xm1[j].y + xm2[j].y * xm2[j].y;
              dpdm_local[tm.icslot(j)] += dtinv4 * (t1 + t2) * (t1 + t2);
            }
        }
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          xm2[j] = xm1[j];
          xm1[j] = x[j];
        }
      }
    }
  }
  // final state (+ last trajectory slot)
#pragma unroll
  for (int j = 0; j < EPT; j++)
    if (tm.ok(j)) {
      double* xT = A.xT + (size_t)tm.ic(j) * 2 * dim;
      xT[tm.st.it[j]] = x[j].x;
      xT[dim + tm.st.it[j]] = x[j].y;
      if (traj) {
        double* dst = traj + ((size_t)A.nsub * A.nb + tm.ic(j)) * 2 * dim;
        dst[tm.st.it[j]] = x[j].x;
        dst[dim + tm.st.it[j]] = x[j].y;
      }
    }
  double v[2 * ICPB];
#pragma unroll
  for (int q = 0; q < ICPB; q++) {
    v[2 * q] = pen_local[q];
    v[2 * q + 1] = dpdm_local[q];
  }
  tm.template sum<2 * ICPB>(v);
  if (threadIdx.x == 0) {
    int nvalid = 0;
#pragma unroll
    for (int q = 0; q < ICPB; q++)
      if (tm.icvalid(q)) {
        A.pen_out[tm.ic0 + q] = v[2 * q] + pen_uniform[q];
        A.dpdm_out[tm.ic0 + q] = v[2 * q + 1] / A.ntime;
        nvalid++;
      }
    atomicAdd(A.napply, napply * nvalid);
  }
}

// ---------------------------------------------------------------------------------------------
// adjoint sweep: TimeStepper::solveAdjointODE + ImplMidpoint::evolveBWD + compute_dRHS_dParams
// (primal states come from the stored trajectory for Lindblad AND Schroedinger: 288 GB of HBM make
// the reference's backward recomputation of the Schroedinger primal unnecessary - except for explicit
// Euler, where the recomputed chain differs from the forward states and defines the reference's gradient)
// ---------------------------------------------------------------------------------------------
template <int Q, bool LIND, int VAR, bool QUBIT, bool GM, bool PLAIN = false>
__global__ void __launch_bounds__(Variant<VAR>::MAXB) k_adjoint(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef Team<Q, LIND, VAR, QUBIT, GM> TM;
  constexpr int EPT = TM::EPT, ICPB = TM::ICPB;
  const DevSys& S = A.S;
  TM tm;
  tm.init(S, smem, A.nb, A.use_gmres);
  team_sync<TM::V::ONEWAVE>();
  const int dim = S.dim;
  double2 xb[EPT], xn[EPT];  // adjoint state, primal state x_n (end of the step being reversed)
  const double* traj = A.traj;
  auto load_state = [&](int s, double2(&dst)[EPT]) {
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const double* src = traj + ((size_t)s * A.nb + tm.ic(j)) * 2 * dim;
      dst[j] = make_double2(__builtin_nontemporal_load(src + tm.st.it[j]), __builtin_nontemporal_load(src + dim + tm.st.it[j]));
    }
  };
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    const double* xbT = A.xbarT + (size_t)tm.ic(j) * 2 * dim;
    xb[j] = make_double2(xbT[tm.st.it[j]], xbT[dim + tm.st.it[j]]);
  }
  const bool jpairs = S.npairs > 0;
  const bool ee = PLAIN ? false : A.stepper_ee != 0;
  if (ee && !LIND) {
    // Schroedinger runs of the reference do not store the forward states: the adjoint loop re-computes the primal
    // backwards with the FORWARD stepper and a negative step, xprimal <- xprimal + (tstart - tstop) M(tstop) xprimal
    // (src/timestepper.cpp:229-231 with ExplEuler::evolveFWD :496-507), and so do the dpdm states (:207-211, :236-243).
    // That is exact for the symmetric IMR family but O(dt) away from the forward states for explicit Euler, and the
    // reference's EE gradient is defined on this backward chain.  Reproduce it: overwrite the stored trajectory with
    // the chain x_N, B_N x_N, B_{N-1} B_N x_N, ... (every thread only touches its own elements), then run the
    // ordinary adjoint loop on it.
    double* trajw = const_cast<double*>(traj);
    double2 xp[EPT];
    load_state(A.nsub, xp);
    for (int s = A.nsub - 1; s >= 0; s--) {
      StepC<Q> c1;
      load_step<Q>(A.ctl + (size_t)(s + 1) * A.cs, c1, jpairs);  // M(tstop of step s) = row s + 1
      if (TM::V::LEAN) scalarize<Q>(c1, jpairs);
      c1.g = S.dense ? reinterpret_cast<const double2*>(S.gtab) + (size_t)(s + 1) * S.N * S.N : nullptr;
      tm.st.prep(S, tm.L, c1);
      const double hneg = -A.ctl[(size_t)s * A.cs];
      tm.publish(xp);
      double2 t[EPT];
      tm.template apply_all<false>(S, c1, xp, t);
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        xp[j].x = fma(hneg, t[j].x, xp[j].x);
        xp[j].y = fma(hneg, t[j].y, xp[j].y);
        if (tm.ok(j)) {
          double* dst = trajw + ((size_t)s * A.nb + tm.ic(j)) * 2 * dim;
          const int e = at_use<TM::EPE>(tm.st.it[j]);
          dst[e] = xp[j].x;
          dst[dim + e] = xp[j].y;
        }
      }
    }
    __threadfence_block();
    team_sync<TM::V::ONEWAVE>();
  }
  // Latency-bound variants (few elements per thread) carry x_{n+1} in registers and prefetch x_{n-1} one
  // step ahead; the throughput variants re-read them (L2 / HBM) to keep the register footprint small.
  constexpr bool CARRY = !TM::V::LEAN && !PLAIN;  // (PLAIN: nothing in the sweep reads the primal states)
  if (CARRY) load_state(A.nsub, xn);
  double jbar_pen[ICPB], jbar_dpdm[ICPB];
#pragma unroll
  for (int q = 0; q < ICPB; q++) {
    const int bq = min(tm.ic0 + q, A.nb - 1);
    jbar_pen[q] = A.jbar[bq * 3 + 0];
    jbar_dpdm[q] = A.jbar[bq * 3 + 1];
  }
  const bool pen_on = PLAIN ? false : A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  const bool wj_reduce = wj_on && !LIND && A.tg.objective_type == QD_OBJ_JTRACE;
  const bool dpdm_on = PLAIN ? false : (A.gamma_dpdm > 1e-13 && !LIND);
  bool guard[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++) guard[j] = A.leak_on && tm.ok(j) && tm.st.is_guard(S, j);
  const double dtinv4 = 1.0 / (A.dt * A.dt * A.dt * A.dt);
  const int ntime = A.ntime;
  double2 x[EPT], xnext[CARRY ? EPT : 1];
  if (CARRY) load_state(A.nsub - 1, reinterpret_cast<double2(&)[EPT]>(xnext));  // primal at the start of the last sub-step
  // Latency-bound variants: the primal stage z and the control row of sub-step s - 1 are fetched while sub-step s is being reversed.
  // Loaded where they are used, each costs a full global / scalar memory round trip on the critical path of EVERY step of a sweep
  // that is nothing but one dependent chain per initial condition.
  constexpr bool ZAHEAD = !TM::V::LEAN;
  double2 znext[ZAHEAD ? EPT : 1];
  StepC<Q> cn;
  auto load_stage = [&](int ss, double2(&dst)[ZAHEAD ? EPT : 1]) {
#pragma unroll
    for (int j = 0; j < (ZAHEAD ? EPT : 0); j++) dst[j] = stage_load(A.ztraj, (size_t)ss * A.nb + tm.ic(j), dim, tm.st.it[j]);
  };
  if (ZAHEAD && !ee && A.nsub > 0) {
    load_stage(A.nsub - 1, znext);
    load_step<Q>(A.ctl + (size_t)(A.nsub - 1) * A.cs, cn, jpairs);
  }

  vm_drain();
  for (int s = A.nsub - 1; s >= 0; s--) {
    if (CARRY) {
#pragma unroll
      for (int j = 0; j < EPT; j++) x[j] = xnext[CARRY ? j : 0];
      if (s > 0) load_state(s - 1, reinterpret_cast<double2(&)[EPT]>(xnext));
    } else if (ee) {
      load_state(s, x);
    }
    // ---- penalty adjoints at the end of a full step, using the primal x_n (timestepper.cpp:220-227)
    if ((pen_on || dpdm_on) && (s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages;
      const double tstop = n * A.dt;
      if (!CARRY) load_state(s + 1, xn);
      if (dpdm_on) {  // penaltyDpDm_diff (timestepper.cpp:372-442); all five states come from HBM
        double2 m2[EPT], m1[EPT], p1[EPT], p2[EPT];
        if (n > 1) load_state((n - 2) * A.nstages, m2);
        if (n > 0) load_state((n - 1) * A.nstages, m1);
        if (n < ntime) load_state((n + 1) * A.nstages, p1);
        if (n < ntime - 1) load_state((n + 2) * A.nstages, p2);
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const double Jb = jbar_dpdm[tm.icslot(j)] / ntime;
          const double xr = xn[j].x, xi = xn[j].y;
          double acc = 0.0;
          if (n > 1) {
            const double t1 = m2[j].x * m2[j].x - 2.0 * m1[j].x * m1[j].x + xr * xr;
            const double t2 = m2[j].y * m2[j].y - 2.0 * m1[j].y * m1[j].y + xi * xi;
            acc += 2.0 * (t1 + t2);
          }
          if (n > 0 && n < ntime) {
            const double t1 = m1[j].x * m1[j].x - 2.0 * xr * xr + p1[j].x * p1[j].x;
            const double t2 = m1[j].y * m1[j].y - 2.0 * xi * xi + p1[j].y * p1[j].y;
            acc += -4.0 * (t1 + t2);
          }
          if (n < ntime - 1) {
            const double t1 = xr * xr - 2.0 * p1[j].x * p1[j].x + p2[j].x * p2[j].x;
            const double t2 = xi * xi - 2.0 * p1[j].y * p1[j].y + p2[j].y * p2[j].y;
            acc += 2.0 * (t1 + t2);
          }
          xb[j].x += acc * 2.0 * xr * dtinv4 * Jb;
          xb[j].y += acc * 2.0 * xi * dtinv4 * Jb;
        }
      }
      if (pen_on) {  // penaltyIntegral_diff (timestepper.cpp:300-339)
        if (wj_on) {
          double weight;
          if (A.wjw) {
            weight = to_scalar(A.wjw[n - 1]);
          } else {
            const double a = (tstop - A.Tfinal) / A.penalty_param;
            weight = 1.0 / A.penalty_param * exp(-(a * a));
          }
          double rb[ICPB], ib[ICPB];
          if (wj_reduce) {
            double v[2 * ICPB];
#pragma unroll
            for (int q = 0; q < 2 * ICPB; q++) v[q] = 0.0;
#pragma unroll
            for (int j = 0; j < EPT; j++)
              if (tm.ok(j)) evalJ_part<LIND>(S, A.tg, tm.ic(j), at_use<TM::EPE>(tm.st.it[j]), xn[j], v[2 * tm.icslot(j)], v[2 * tm.icslot(j) + 1]);
            tm.template sum<2 * ICPB>(v);
#pragma unroll
            for (int q = 0; q < ICPB; q++) finalizeJ_diff<LIND>(A.tg, v[2 * q], v[2 * q + 1], rb[q], ib[q]);
          } else {
#pragma unroll
            for (int q = 0; q < ICPB; q++) finalizeJ_diff<LIND>(A.tg, 0.0, 0.0, rb[q], ib[q]);
          }
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.ok(j)) {
              const int q = tm.icslot(j);
              evalJ_diff_elem<LIND>(S, A.tg, tm.ic(j), at_use<TM::EPE>(tm.st.it[j]), xn[j], xb[j], weight * rb[q] * jbar_pen[q] * A.dt,
                                    weight * ib[q] * jbar_pen[q] * A.dt);
            }
        }
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (guard[j]) {
            xb[j].x += 2.0 * xn[j].x * jbar_pen[tm.icslot(j)] / ntime;
            xb[j].y += 2.0 * xn[j].y * jbar_pen[tm.icslot(j)] / ntime;
          }
      }
    }
    StepC<Q> c;
    double2 znow[ZAHEAD ? EPT : 1];
    if (ZAHEAD && !ee) {
      c = cn;
#pragma unroll
      for (int j = 0; j < (ZAHEAD ? EPT : 0); j++) znow[j] = znext[j];
      if (s > 0) {
        load_stage(s - 1, znext);
        load_step<Q>(A.ctl + (size_t)(s - 1) * A.cs, cn, jpairs);
      }
    } else {
      load_step<Q>(A.ctl + (size_t)s * A.cs, c, jpairs);
    }
    if (TM::V::LEAN) scalarize<Q>(c, jpairs);
    c.g = S.dense ? reinterpret_cast<const double2*>(S.gtab) + (size_t)s * S.N * S.N : nullptr;
    tm.st.prep(S, tm.L, c);
    double cf[2 * Q * ICPB];
#pragma unroll
    for (int i = 0; i < 2 * Q * ICPB; i++) cf[i] = 0.0;
    // (the coefficient sums are completed behind the barrier that publishes the next vector: store_coeffs, tm.publish, collect_coeffs)
    auto coeff_dst = [&](int g) -> double* {
      const int q = ICPB == 1 ? 0 : g / (2 * Q), i = ICPB == 1 ? g : g % (2 * Q);
      return tm.icvalid(q) ? A.coeff + ((size_t)(tm.ic0 + q) * A.nsub + s) * 2 * Q + i : nullptr;
    };
    auto store_coeffs = [&]() { tm.template sum_store_post<2 * Q * ICPB>(cf, coeff_dst); };
    auto collect_coeffs = [&]() { tm.template sum_store_collect<2 * Q * ICPB>(coeff_dst); };
    if (ee) {
      // ExplEuler::evolveBWD (timestepper.cpp:506-520): gradient with dt * x_adj against x_{n-1}, then
      // x_adj += dt M(tstop)^T x_adj.  The table row of sub-step s holds M(tstart); M(tstop) is row s+1
      // (the last row is followed by one extra row for t = T).
      tm.publish(x);
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (tm.ok(j)) {
#pragma unroll
          for (int k = 0; k < Q; k++) {
            double2 Av, Bv;
            tm.st.ladder(S, tm.L, tm.vecj(j), k, j, Av, Bv);
            cf[tm.icslot(j) * 2 * Q + 2 * k] += c.h * (Bv.y * xb[j].x - Bv.x * xb[j].y);
            cf[tm.icslot(j) * 2 * Q + 2 * k + 1] += c.h * (Av.x * xb[j].x + Av.y * xb[j].y);
          }
        }
      store_coeffs();
      StepC<Q> c1;
      load_step<Q>(A.ctl + (size_t)(s + 1) * A.cs, c1, jpairs);
      c1.g = S.dense ? reinterpret_cast<const double2*>(S.gtab) + (size_t)(s + 1) * S.N * S.N : nullptr;
      tm.st.prep(S, tm.L, c1);
      tm.publish(xb);
      collect_coeffs();
      double2 t[EPT];
      tm.template apply_all<true>(S, c1, xb, t);
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        xb[j].x = fma(c.h, t[j].x, xb[j].x);
        xb[j].y = fma(c.h, t[j].y, xb[j].y);
      }
    } else {
      // ImplMidpoint::evolveBWD (timestepper.cpp:631-694).  The reference repeats the forward solve of the sub-step to get the
      // primal stage z = x + h/2 k (:640-652); here the forward sweep has stored z next to the trajectory (SweepArgs::ztraj,
      // the same solve on the same data: identical values), so only the adjoint solve remains and neither x nor z is alive
      // while it runs.
      double2 kb[EPT];  // adjoint stage: (I - h/2 M)^T kbar = xbar ; kbar *= h
      tm.template solve<true>(A, c, 0.5 * c.h, xb, kb);
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        kb[j].x *= c.h;
        kb[j].y *= c.h;
      }
      double2 z[EPT];
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        if (ZAHEAD) {
          z[j] = znow[ZAHEAD ? j : 0];
        } else {
          z[j] = stage_load(A.ztraj, (size_t)s * A.nb + tm.ic(j), dim, tm.st.it[j]);
        }
      }
      tm.publish(z);
      // gradient coefficients: x^T dM/dp_k z and x^T dM/dq_k z with x := kbar
      if constexpr (TM::ST::WHOLE) {  // matrix-core stencils: the commutators of all the thread's elements at once per oscillator
        typename TM::ST::ZOps zo;
        tm.st.ladder_fetch(tm.vec(), zo);
#pragma unroll
        for (int k = 0; k < Q; k++) {
          double2 Av[EPT], Bv[EPT];
          tm.st.ladder_whole(S, zo, k, Av, Bv);
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.ok(j)) {
              cf[tm.icslot(j) * 2 * Q + 2 * k] += Bv[j].y * kb[j].x - Bv[j].x * kb[j].y;
              cf[tm.icslot(j) * 2 * Q + 2 * k + 1] += Av[j].x * kb[j].x + Av[j].y * kb[j].y;
            }
        }
      } else {
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (tm.ok(j)) {
#pragma unroll
            for (int k = 0; k < Q; k++) {
              double2 Av, Bv;
              tm.st.ladder(S, tm.L, tm.vecj(j), k, j, Av, Bv);
              cf[tm.icslot(j) * 2 * Q + 2 * k] += Bv.y * kb[j].x - Bv.x * kb[j].y;
              cf[tm.icslot(j) * 2 * Q + 2 * k + 1] += Av.x * kb[j].x + Av.y * kb[j].y;
            }
          }
      }
      store_coeffs();
      // xbar += M^T kbar
      tm.publish(kb);
      collect_coeffs();
      double2 t[EPT];
      tm.template apply_all<true>(S, c, kb, t);
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        xb[j].x += t[j].x;
        xb[j].y += t[j].y;
      }
    }
    if (CARRY) {
#pragma unroll
      for (int j = 0; j < EPT; j++) xn[j] = x[j];
    }
  }
  if (A.xbar0) {
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (tm.ok(j)) {
        double* d0 = A.xbar0 + (size_t)tm.ic(j) * 2 * dim;
        d0[tm.st.it[j]] = xb[j].x;
        d0[dim + tm.st.it[j]] = xb[j].y;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// single operator application (test hook = MatMult / MatMultTranspose on the shell)
// ---------------------------------------------------------------------------------------------
template <int Q, bool LIND, int VAR, bool QUBIT>
__global__ void __launch_bounds__(Variant<VAR>::MAXB) k_apply(const DevSys S, const double* __restrict__ ctlrow, int transpose,
                                                               const double* __restrict__ xin, double* __restrict__ yout, int nb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef Team<Q, LIND, VAR, QUBIT> TM;
  constexpr int EPT = TM::EPT;
  TM tm;
  tm.init(S, smem, nb);
  const int dim = S.dim;
  double2 x[EPT], y[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    const double* x0 = xin + (size_t)tm.ic(j) * 2 * dim;
    x[j] = make_double2(x0[tm.st.it[j]], x0[dim + tm.st.it[j]]);
  }
  team_sync<TM::V::ONEWAVE>();
  tm.publish(x);
  StepC<Q> c;
  load_step<Q>(ctlrow, c, S.npairs > 0);
  c.g = S.dense ? reinterpret_cast<const double2*>(S.gtab) : nullptr;  // one-row table for the test hook
  tm.st.prep(S, tm.L, c);
  if (transpose) tm.template apply_all<true>(S, c, x, y);
  else tm.template apply_all<false>(S, c, x, y);
#pragma unroll
  for (int j = 0; j < EPT; j++)
    if (tm.ok(j)) {
      double* yo = yout + (size_t)tm.ic(j) * 2 * dim;
      yo[tm.st.it[j]] = y[j].x;
      yo[dim + tm.st.it[j]] = y[j].y;
    }
}

}  // namespace qd
