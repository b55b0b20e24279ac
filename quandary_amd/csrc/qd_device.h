// qd_device.h — device templates of the sweep kernels (included by qd_inst.hip and qd_kernels.hip).
// See qd_kernels.hip for the design notes and the reference citations.
#pragma once
#include <hip/hip_runtime.h>

#include "qd_internal.h"

namespace qd {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int digit(uint64_t d, int k) { return (int)((d >> (8 * k)) & 0xffull); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum of NV values; every thread returns the same bits (fixed summation order), so
// convergence decisions taken on the result are uniform.  Contains ONE __syncthreads(); the caller
// guarantees another barrier before the next call re-writes `red`.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    v[i] = wave_sum(v[i]);
    if (lane == 0) red[i * nw + wave] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double s = 0.0;
    for (int w = 0; w < nw; w++) s += red[i * nw + w];
    v[i] = s;
  }
}

// Per-thread description of one owned element of the vectorised state.
struct Elem {
  int it;         // storage index, or -1 when the slot is beyond dim
  uint64_t dig;   // bra digits i_k, 8 bits each (oscillator 0 slowest in the Hilbert index)
  uint64_t digp;  // ket digits i_k' (Lindblad), 0 for Schroedinger
  double dw;      // Delta = h(I) - h(I')    (mastereq.hpp:316-403)
  double dd;      // d = L2 + L1diag          (mastereq.hpp:339-353, :416-433)
};

template <int Q, bool LIND>
__device__ __forceinline__ void elem_init(const DevSys& S, int it, Elem& e) {
  e.it = it;
  e.dig = 0;
  e.digp = 0;
  e.dw = 0.0;
  e.dd = 0.0;
  if (it < 0) return;
  const int I = LIND ? it % S.N : it;
  const int Ip = LIND ? it / S.N : 0;
  int ia[Q], ipa[Q];
#pragma unroll
  for (int k = 0; k < Q; k++) {
    ia[k] = (I / S.post[k]) % S.n[k];
    ipa[k] = LIND ? (Ip / S.post[k]) % S.n[k] : 0;
    e.dig |= (uint64_t)ia[k] << (8 * k);
    e.digp |= (uint64_t)ipa[k] << (8 * k);
  }
  double hd = 0.0, hdp = 0.0, dd = 0.0;
  int pair = 0;
#pragma unroll
  for (int k = 0; k < Q; k++) {
    hd += S.detune[k] * ia[k] - S.xi[k] / 2.0 * ia[k] * (ia[k] - 1);
    if (LIND) {
      hdp += S.detune[k] * ipa[k] - S.xi[k] / 2.0 * ipa[k] * (ipa[k] - 1);
      dd += S.g2[k] * (ia[k] * ipa[k] - 0.5 * (ia[k] * ia[k] + ipa[k] * ipa[k])) - S.g1[k] / 2.0 * (ia[k] + ipa[k]);
    }
#pragma unroll
    for (int l = k + 1; l < Q; l++) {
      hd -= S.xikl[pair] * ia[k] * ia[l];
      if (LIND) hdp -= S.xikl[pair] * ipa[k] * ipa[l];
      pair++;
    }
  }
  e.dw = hd - hdp;
  e.dd = dd;
}

// Controls of one sub-step, wave-uniform (scalar loads from the table row).
template <int Q>
struct StepC {
  double h, p[Q], q[Q];
  double cs[Q * (Q - 1) / 2 + 1], sn[Q * (Q - 1) / 2 + 1];
};

template <int Q>
__device__ __forceinline__ void load_step(const double* __restrict__ row, StepC<Q>& c) {
  constexpr int NP = Q * (Q - 1) / 2;
  c.h = row[0];
#pragma unroll
  for (int k = 0; k < Q; k++) {
    c.p[k] = row[2 + k];
    c.q[k] = row[2 + Q + k];
  }
#pragma unroll
  for (int k = 0; k < NP; k++) {
    c.cs[k] = row[2 + 2 * Q + k];
    c.sn[k] = row[2 + 2 * Q + NP + k];
  }
}

// Ladder-operator neighbour sums of oscillator k at one element (control(), mastereq.hpp:818-912):
//   U1 = sqrt(i+1) x(it+post), U2 = sqrt(i'+1) x(it+N post), D1 = sqrt(i) x(it-post), D2 = sqrt(i') x(it-N post)
//   A = U1 + U2 - D1 - D2,  B = U1 - U2 + D1 - D2
// so that the control part of y = M x is  y_re += q A_re + p B_im,  y_im += q A_im - p B_re, and
// dRHSdp_getcoeffs (mastereq.hpp:553-604) is  res_p = (B_im, -B_re), res_q = (A_re, A_im).
// Invalid neighbours read the element itself with a zero coefficient (no divergence).
template <bool LIND>
__device__ __forceinline__ void ladder_AB(const DevSys& S, int k, const Elem& e, const double2* __restrict__ sx,
                                          const double* __restrict__ ssq, double2& A, double2& B) {
  const int a = digit(e.dig, k), n = S.n[k], st = S.post[k], it = e.it;
  const bool up = a < n - 1, dn = a > 0;
  const double su = up ? ssq[a + 1] : 0.0, sd = dn ? ssq[a] : 0.0;
  const double2 xu = sx[up ? it + st : it], xd = sx[dn ? it - st : it];
  double er = su * xu.x, ei = su * xu.y;  // U1
  double fr = -sd * xd.x, fi = -sd * xd.y;  // -D1
  if (LIND) {
    const int ap = digit(e.digp, k), stp = S.N * st;
    const bool upp = ap < n - 1, dnp = ap > 0;
    const double sup = upp ? ssq[ap + 1] : 0.0, sdp = dnp ? ssq[ap] : 0.0;
    const double2 xup = sx[upp ? it + stp : it], xdp = sx[dnp ? it - stp : it];
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

// y = M x (TRANS=false) or M^T x (TRANS=true) at one element.  The Hamiltonian part of the real
// 2dim x 2dim operator is antisymmetric (M_H^T = -M_H: compare control/control_T, Jkl_coupling/
// Jkl_coupling_T and the drift signs at mastereq.cpp:1541-1542 vs :1665-1666), the dissipator
// diagonal is symmetric, and the T1 off-diagonal term moves to the mirrored neighbour
// (L1decay / L1decay_T, mastereq.hpp:758-797).
template <int Q, bool LIND, bool TRANS>
__device__ __forceinline__ double2 apply_elem(const DevSys& S, const Elem& e, const double2* __restrict__ sx,
                                              const double* __restrict__ ssq, const StepC<Q>& c, const double2 xs) {
  // Hamiltonian part, accumulated for the forward operator
  double hr = e.dw * xs.y, hi = -e.dw * xs.x;
#pragma unroll
  for (int k = 0; k < Q; k++) {
    double2 A, B;
    ladder_AB<LIND>(S, k, e, sx, ssq, A, B);
    hr = fma(c.q[k], A.x, fma(c.p[k], B.y, hr));
    hi = fma(c.q[k], A.y, fma(-c.p[k], B.x, hi));
  }
  // dipole-dipole coupling (Jkl_coupling, mastereq.hpp:632-675):
  //   T1 = sqrt(i_k (i_l+1)) x(it-post_k+post_l), T2 = sqrt(i_l (i_k+1)) x(it+post_k-post_l), T3/T4 the ket analogues
  //   y += J [ sin (T1 - T2 + T3 - T4) - i cos (T1 + T2 - T3 - T4) ]
  {
    int pair = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
#pragma unroll
      for (int l = k + 1; l < Q; l++, pair++) {
        const double Jkl = S.J[pair];
        if (!(fabs(Jkl) > 1e-10)) continue;
        const int a = digit(e.dig, k), b = digit(e.dig, l), it = e.it;
        const int sk = S.post[k], sl = S.post[l];
        const bool v1 = a > 0 && b < S.n[l] - 1, v2 = a < S.n[k] - 1 && b > 0;
        const double s1 = v1 ? ssq[a] * ssq[b + 1] : 0.0, s2 = v2 ? ssq[b] * ssq[a + 1] : 0.0;
        const double2 x1 = sx[v1 ? it - sk + sl : it], x2 = sx[v2 ? it + sk - sl : it];
        double ar = s1 * x1.x - s2 * x2.x, ai = s1 * x1.y - s2 * x2.y;  // T1 - T2
        double br = s1 * x1.x + s2 * x2.x, bi = s1 * x1.y + s2 * x2.y;  // T1 + T2
        if (LIND) {
          const int ap = digit(e.digp, k), bp = digit(e.digp, l);
          const int skp = S.N * sk, slp = S.N * sl;
          const bool v3 = ap > 0 && bp < S.n[l] - 1, v4 = ap < S.n[k] - 1 && bp > 0;
          const double s3 = v3 ? ssq[ap] * ssq[bp + 1] : 0.0, s4 = v4 ? ssq[bp] * ssq[ap + 1] : 0.0;
          const double2 x3 = sx[v3 ? it - skp + slp : it], x4 = sx[v4 ? it + skp - slp : it];
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
    yr = fma(e.dd, xs.x, yr);
    yi = fma(e.dd, xs.y, yi);
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const double g1 = S.g1[k];
      if (!(fabs(g1) > 1e-12)) continue;
      const int a = digit(e.dig, k), ap = digit(e.digp, k), n = S.n[k], st = S.post[k] * (S.N + 1);
      if (!TRANS) {
        const bool v = a < n - 1 && ap < n - 1;
        const double l1 = v ? g1 * ssq[a + 1] * ssq[ap + 1] : 0.0;
        const double2 xn = sx[v ? e.it + st : e.it];
        yr = fma(l1, xn.x, yr);
        yi = fma(l1, xn.y, yi);
      } else {
        const bool v = a > 0 && ap > 0;
        const double l1 = v ? g1 * ssq[a] * ssq[ap] : 0.0;
        const double2 xn = sx[v ? e.it - st : e.it];
        yr = fma(l1, xn.x, yr);
        yi = fma(l1, xn.y, yi);
      }
    }
  }
  return make_double2(yr, yi);
}

// LDS carve-up shared by all sweep kernels
struct Lds {
  double2* sx;
  double* ssq;
  double* red;
};
__device__ __forceinline__ Lds carve(unsigned char* smem, int dim, int maxn) {
  Lds l;
  l.sx = reinterpret_cast<double2*>(smem);
  l.ssq = reinterpret_cast<double*>(l.sx + dim);
  l.red = l.ssq + ((maxn + 2 + 1) & ~1);
  return l;
}
static size_t lds_bytes(int dim, int maxn, int block, int nred) {
  return sizeof(double2) * (size_t)dim + sizeof(double) * (size_t)((maxn + 2 + 1) & ~1) + sizeof(double) * (size_t)nred * ((block + 63) / 64);
}

constexpr int NRED = 16;  // max values reduced at once (2*QD_MAX_OSC)

template <int EPT>
constexpr int launch_bound() {
  return EPT <= 4 ? 1024 : (EPT <= 8 ? 512 : 256);
}

template <int Q, bool LIND, int EPT>
__device__ __forceinline__ void init_elems(const DevSys& S, Elem (&e)[EPT], double* ssq) {
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    const int it = (int)threadIdx.x + j * (int)blockDim.x;
    elem_init<Q, LIND>(S, it < S.dim ? it : -1, e[j]);
  }
  for (int i = threadIdx.x; i < S.maxn + 2; i += blockDim.x) ssq[i] = sqrt((double)i);
}

// Solve (I - alpha M^{(T)}) y = b by the reference's Neumann iteration (timestepper.cpp:697-727).
// On entry sx may hold anything that all threads have finished reading; on exit y holds the solution
// in registers AND sx holds y (after a barrier).  Returns the number of RHS applications.
template <int Q, bool LIND, bool TRANS, int EPT>
__device__ __forceinline__ int neumann(const SweepArgs& A, const Elem (&e)[EPT], const Lds& L, const StepC<Q>& c, double alpha,
                                       const double2 (&b)[EPT], double2 (&y)[EPT]) {
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    y[j] = b[j];
    if (e[j].it >= 0) L.sx[e[j].it] = y[j];
  }
  __syncthreads();
  double err0 = 1.0;
  int iter;
  for (iter = 0; iter < A.maxiter; iter++) {
    double2 yn[EPT];
    double d[1] = {0.0};
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      yn[j] = y[j];
      if (e[j].it >= 0) {
        const double2 t = apply_elem<Q, LIND, TRANS>(A.S, e[j], L.sx, L.ssq, c, y[j]);
        yn[j].x = fma(alpha, t.x, b[j].x);
        yn[j].y = fma(alpha, t.y, b[j].y);
        const double dx = y[j].x - yn[j].x, dy = y[j].y - yn[j].y;
        d[0] += dx * dx + dy * dy;
      }
    }
    block_sum<1>(d, L.red);  // barrier: every read of sx above has completed
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      y[j] = yn[j];
      if (e[j].it >= 0) L.sx[e[j].it] = y[j];
    }
    __syncthreads();
    const double errnorm = sqrt(d[0]);
    if (iter == 0) err0 = errnorm;
    if (errnorm < A.abstol) { iter++; break; }
    if (errnorm / err0 < A.reltol) { iter++; break; }
  }
  return iter;
}

// ---------------------------------------------------------------------------------------------
// objective pieces evaluated on register-resident states (OptimTarget::evalJ / evalJ_diff)
// ---------------------------------------------------------------------------------------------
// Thread-local part of (J_re, J_im) for the elements this thread owns (optimtarget.cpp:712-799).
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
        if (LIND) {
          jre += (tr * x.x + ti * x.y) / pur;
        } else {
          jre += (tr * x.x + ti * x.y) / pur;
          jim += -ti * x.x + tr * x.y;
        }
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

// isGuardLevel (util.cpp:259-278) for the diagonal element this thread owns
template <int Q, bool LIND>
__device__ __forceinline__ bool is_guard(const DevSys& S, const Elem& e) {
  if (e.it < 0) return false;
  if (LIND && e.dig != e.digp) return false;
  bool g = false;
#pragma unroll
  for (int k = 0; k < Q; k++) {
    const int a = digit(e.dig, k);
    g = g || (a == S.n[k] - 1 && a >= S.ness[k]);
  }
  return g;
}

// ---------------------------------------------------------------------------------------------
// forward sweep: TimeStepper::solveODE for every initial condition of the batch
// ---------------------------------------------------------------------------------------------
template <int Q, bool LIND, int EPT>
__global__ void __launch_bounds__(launch_bound<EPT>()) k_forward(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const DevSys& S = A.S;
  const Lds L = carve(smem, S.dim, S.maxn);
  const int b = blockIdx.x, dim = S.dim, T = blockDim.x;
  Elem e[EPT];
  init_elems<Q, LIND, EPT>(S, e, L.ssq);
  double2 x[EPT];
  const double* x0 = A.x0 + (size_t)b * 2 * dim;
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    x[j] = make_double2(0.0, 0.0);
    if (e[j].it >= 0) {
      x[j] = make_double2(x0[e[j].it], x0[dim + e[j].it]);
      L.sx[e[j].it] = x[j];
    }
  }
  __syncthreads();
  // penalty bookkeeping
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  // Schroedinger Jtrace is the only objective whose finalizeJ is nonlinear in the per-state sums:
  // it needs a block reduction per step; everything else accumulates thread-locally.
  const bool wj_reduce = wj_on && !LIND && A.tg.objective_type == QD_OBJ_JTRACE;
  const bool dpdm_on = A.gamma_dpdm > 1e-13 && !LIND;
  bool guard[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++) guard[j] = A.leak_on && is_guard<Q, LIND>(S, e[j]);
  double pen_local = 0.0, dpdm_local = 0.0, pen_uniform = 0.0;
  double2 xm1[EPT], xm2[EPT];  // dpdm history (x_n, x_{n-1})
#pragma unroll
  for (int j = 0; j < EPT; j++) xm1[j] = xm2[j] = x[j];
  unsigned long long napply = 0;
  double* traj = A.traj;
  const double dtinv4 = 1.0 / (A.dt * A.dt * A.dt * A.dt);

  for (int s = 0; s < A.nsub; s++) {
    if (traj) {
      double* dst = traj + ((size_t)s * A.nb + b) * 2 * dim;
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) {
          dst[e[j].it] = x[j].x;
          dst[dim + e[j].it] = x[j].y;
        }
    }
    StepC<Q> c;
    load_step<Q>(A.ctl + (size_t)s * A.cs, c);
    // rhs = M x   (ImplMidpoint::evolveFWD, timestepper.cpp:594; ExplEuler :502)
    double2 rhs[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      rhs[j] = make_double2(0.0, 0.0);
      if (e[j].it >= 0) rhs[j] = apply_elem<Q, LIND, false>(S, e[j], L.sx, L.ssq, c, x[j]);
    }
    napply++;
    __syncthreads();  // all reads of x in sx done before the solver overwrites it
    if (A.stepper_ee) {
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        x[j].x = fma(c.h, rhs[j].x, x[j].x);
        x[j].y = fma(c.h, rhs[j].y, x[j].y);
      }
    } else {
      double2 k[EPT];
      napply += neumann<Q, LIND, false, EPT>(A, e, L, c, 0.5 * c.h, rhs, k);
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        x[j].x = fma(c.h, k[j].x, x[j].x);
        x[j].y = fma(c.h, k[j].y, x[j].y);
      }
    }
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (e[j].it >= 0) L.sx[e[j].it] = x[j];
    __syncthreads();

    // in-loop penalties, evaluated at the end of a FULL time step (timestepper.cpp:141-154)
    if ((s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages - 1;  // step index n: state is x_{n+1}
      const double tstop = (n + 1) * A.dt;
      if (pen_on) {
        double weight = 0.0;
        if (wj_on) {
          const double a = (tstop - A.Tfinal) / A.penalty_param;
          weight = 1.0 / A.penalty_param * exp(-(a * a));
        }
        if (wj_reduce) {
          double v[2] = {0.0, 0.0};
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (e[j].it >= 0) evalJ_part<LIND>(S, A.tg, b, e[j].it, x[j], v[0], v[1]);
          block_sum<2>(v, L.red);
          pen_uniform += weight * finalizeJ<LIND>(A.tg, v[0], v[1]) * A.dt;
          __syncthreads();
        } else if (wj_on) {
          double jr = 0.0, ji = 0.0;
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (e[j].it >= 0) evalJ_part<LIND>(S, A.tg, b, e[j].it, x[j], jr, ji);
          // finalizeJ is affine here: J = jr (Jfrobenius, Jmeasure) or 1 - jr (Lindblad Jtrace)
          if (A.tg.objective_type == QD_OBJ_JTRACE) {
            pen_local -= weight * A.dt * jr;
            pen_uniform += weight * A.dt;
          } else {
            pen_local += weight * A.dt * jr;
          }
        }
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (guard[j]) pen_local += (x[j].x * x[j].x + x[j].y * x[j].y) / A.ntime;
      }
      if (dpdm_on) {
        if (n > 0) {
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (e[j].it >= 0) {
              const double t1 = x[j].x * x[j].x - 2.0 * xm1[j].x * xm1[j].x + xm2[j].x * xm2[j].x;
              const double t2 = x[j].y * x[j].y - 2.0 * xm1[j].y * xm1[j].y + xm2[j].y * xm2[j].y;
              dpdm_local += dtinv4 * (t1 + t2) * (t1 + t2);
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
  double* xT = A.xT + (size_t)b * 2 * dim;
  double* dst = traj ? traj + ((size_t)A.nsub * A.nb + b) * 2 * dim : nullptr;
#pragma unroll
  for (int j = 0; j < EPT; j++)
    if (e[j].it >= 0) {
      xT[e[j].it] = x[j].x;
      xT[dim + e[j].it] = x[j].y;
      if (dst) {
        dst[e[j].it] = x[j].x;
        dst[dim + e[j].it] = x[j].y;
      }
    }
  double v[2] = {pen_local, dpdm_local};
  block_sum<2>(v, L.red);
  if (threadIdx.x == 0) {
    A.pen_out[b] = v[0] + pen_uniform;
    A.dpdm_out[b] = v[1] / A.ntime;
    atomicAdd(A.napply, napply);
  }
  (void)T;
}

// ---------------------------------------------------------------------------------------------
// adjoint sweep: TimeStepper::solveAdjointODE + ImplMidpoint::evolveBWD + compute_dRHS_dParams
// (primal states come from the stored trajectory for Lindblad AND Schroedinger: 288 GB of HBM make
// the reference's backward recomputation of the Schroedinger primal unnecessary)
// ---------------------------------------------------------------------------------------------
template <int Q, bool LIND, int EPT>
__global__ void __launch_bounds__(launch_bound<EPT>()) k_adjoint(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const DevSys& S = A.S;
  const Lds L = carve(smem, S.dim, S.maxn);
  const int b = blockIdx.x, dim = S.dim;
  Elem e[EPT];
  init_elems<Q, LIND, EPT>(S, e, L.ssq);
  __syncthreads();
  double2 xb[EPT], xn[EPT];  // adjoint state, primal state x_n (end of the step being reversed)
  const double* xbT = A.xbarT + (size_t)b * 2 * dim;
  const double* traj = A.traj;
  auto load_state = [&](int s, double2(&dst)[EPT]) {
    const double* src = traj + ((size_t)s * A.nb + b) * 2 * dim;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      dst[j] = make_double2(0.0, 0.0);
      if (e[j].it >= 0) dst[j] = make_double2(src[e[j].it], src[dim + e[j].it]);
    }
  };
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    xb[j] = make_double2(0.0, 0.0);
    if (e[j].it >= 0) xb[j] = make_double2(xbT[e[j].it], xbT[dim + e[j].it]);
  }
  load_state(A.nsub, xn);
  const double jbar_pen = A.jbar[b * 3 + 0], jbar_dpdm = A.jbar[b * 3 + 1];
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  const bool wj_reduce = wj_on && !LIND && A.tg.objective_type == QD_OBJ_JTRACE;
  const bool dpdm_on = A.gamma_dpdm > 1e-13 && !LIND;
  bool guard[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++) guard[j] = A.leak_on && is_guard<Q, LIND>(S, e[j]);
  const double dtinv4 = 1.0 / (A.dt * A.dt * A.dt * A.dt);
  const int ntime = A.ntime;

  for (int s = A.nsub - 1; s >= 0; s--) {
    // ---- penalty adjoints at the end of a full step, using the primal x_n (timestepper.cpp:220-227)
    if ((s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages;
      const double tstop = n * A.dt;
      if (dpdm_on) {  // penaltyDpDm_diff (timestepper.cpp:372-442); all five states come from HBM
        const double Jb = jbar_dpdm / ntime;
        double2 m2[EPT], m1[EPT], p1[EPT], p2[EPT];
        if (n > 1) load_state((n - 2) * A.nstages, m2);
        if (n > 0) load_state((n - 1) * A.nstages, m1);
        if (n < ntime) load_state((n + 1) * A.nstages, p1);
        if (n < ntime - 1) load_state((n + 2) * A.nstages, p2);
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          if (e[j].it < 0) continue;
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
          const double a = (tstop - A.Tfinal) / A.penalty_param;
          const double weight = 1.0 / A.penalty_param * exp(-(a * a));
          double rb = 1.0, ib = 0.0;
          if (wj_reduce) {
            double v[2] = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < EPT; j++)
              if (e[j].it >= 0) evalJ_part<LIND>(S, A.tg, b, e[j].it, xn[j], v[0], v[1]);
            block_sum<2>(v, L.red);
            finalizeJ_diff<LIND>(A.tg, v[0], v[1], rb, ib);
            __syncthreads();
          } else {
            finalizeJ_diff<LIND>(A.tg, 0.0, 0.0, rb, ib);
          }
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (e[j].it >= 0)
              evalJ_diff_elem<LIND>(S, A.tg, b, e[j].it, xn[j], xb[j], weight * rb * jbar_pen * A.dt, weight * ib * jbar_pen * A.dt);
        }
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (guard[j]) {
            xb[j].x += 2.0 * xn[j].x * jbar_pen / ntime;
            xb[j].y += 2.0 * xn[j].y * jbar_pen / ntime;
          }
      }
    }
    // ---- primal state at the start of the sub-step
    double2 x[EPT];
    load_state(s, x);
    StepC<Q> c;
    load_step<Q>(A.ctl + (size_t)s * A.cs, c);
    double* co = A.coeff + ((size_t)b * A.nsub + s) * 2 * Q;
    if (A.stepper_ee) {
      // ExplEuler::evolveBWD (timestepper.cpp:506-520): gradient with dt * x_adj against x_{n-1}, then
      // x_adj += dt M(tstop)^T x_adj.  The table row of sub-step s holds M(tstart); M(tstop) is row s+1
      // (the last row is followed by one extra row for t = T).
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) L.sx[e[j].it] = x[j];
      __syncthreads();
      double cf[NRED];
#pragma unroll
      for (int i = 0; i < NRED; i++) cf[i] = 0.0;
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) {
#pragma unroll
          for (int k = 0; k < Q; k++) {
            double2 Av, Bv;
            ladder_AB<LIND>(S, k, e[j], L.sx, L.ssq, Av, Bv);
            cf[2 * k] += c.h * (Bv.y * xb[j].x - Bv.x * xb[j].y);
            cf[2 * k + 1] += c.h * (Av.x * xb[j].x + Av.y * xb[j].y);
          }
        }
      block_sum<NRED>(cf, L.red);
      if (threadIdx.x < 2 * Q) co[threadIdx.x] = cf[threadIdx.x];
      StepC<Q> c1;
      load_step<Q>(A.ctl + (size_t)(s + 1) * A.cs, c1);
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) L.sx[e[j].it] = xb[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) {
          const double2 t = apply_elem<Q, LIND, true>(S, e[j], L.sx, L.ssq, c1, xb[j]);
          xb[j].x = fma(c.h, t.x, xb[j].x);
          xb[j].y = fma(c.h, t.y, xb[j].y);
        }
      __syncthreads();
    } else {
      // ImplMidpoint::evolveBWD (timestepper.cpp:631-694)
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) L.sx[e[j].it] = x[j];
      __syncthreads();
      double2 rhs[EPT];
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        rhs[j] = make_double2(0.0, 0.0);
        if (e[j].it >= 0) rhs[j] = apply_elem<Q, LIND, false>(S, e[j], L.sx, L.ssq, c, x[j]);
      }
      __syncthreads();
      double2 kb[EPT];  // adjoint stage: (I - h/2 M)^T kbar = xbar ; kbar *= h
      neumann<Q, LIND, true, EPT>(A, e, L, c, 0.5 * c.h, xb, kb);
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        kb[j].x *= c.h;
        kb[j].y *= c.h;
      }
      {
        double2 k[EPT];  // primal stage: (I - h/2 M) k = rhs ; z = x + h/2 k
        neumann<Q, LIND, false, EPT>(A, e, L, c, 0.5 * c.h, rhs, k);
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          k[j].x = fma(0.5 * c.h, k[j].x, x[j].x);
          k[j].y = fma(0.5 * c.h, k[j].y, x[j].y);
          if (e[j].it >= 0) L.sx[e[j].it] = k[j];
        }
        __syncthreads();
      }
      // gradient coefficients: x^T dM/dp_k z and x^T dM/dq_k z with x := kbar
      double cf[NRED];
#pragma unroll
      for (int i = 0; i < NRED; i++) cf[i] = 0.0;
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) {
#pragma unroll
          for (int k = 0; k < Q; k++) {
            double2 Av, Bv;
            ladder_AB<LIND>(S, k, e[j], L.sx, L.ssq, Av, Bv);
            cf[2 * k] += Bv.y * kb[j].x - Bv.x * kb[j].y;
            cf[2 * k + 1] += Av.x * kb[j].x + Av.y * kb[j].y;
          }
        }
      block_sum<NRED>(cf, L.red);  // barrier: reads of z done
      if (threadIdx.x < 2 * Q) co[threadIdx.x] = cf[threadIdx.x];
      // xbar += M^T kbar
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) L.sx[e[j].it] = kb[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (e[j].it >= 0) {
          const double2 t = apply_elem<Q, LIND, true>(S, e[j], L.sx, L.ssq, c, kb[j]);
          xb[j].x += t.x;
          xb[j].y += t.y;
        }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) xn[j] = x[j];
  }
  if (A.xbar0) {
    double* d0 = A.xbar0 + (size_t)b * 2 * dim;
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (e[j].it >= 0) {
        d0[e[j].it] = xb[j].x;
        d0[dim + e[j].it] = xb[j].y;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// single operator application (test hook = MatMult / MatMultTranspose on the shell)
// ---------------------------------------------------------------------------------------------
template <int Q, bool LIND, int EPT>
__global__ void __launch_bounds__(launch_bound<EPT>()) k_apply(const DevSys S, const double* __restrict__ ctlrow, int transpose,
                                                                const double* __restrict__ xin, double* __restrict__ yout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Lds L = carve(smem, S.dim, S.maxn);
  const int b = blockIdx.x, dim = S.dim;
  Elem e[EPT];
  init_elems<Q, LIND, EPT>(S, e, L.ssq);
  double2 x[EPT];
  const double* x0 = xin + (size_t)b * 2 * dim;
#pragma unroll
  for (int j = 0; j < EPT; j++)
    if (e[j].it >= 0) {
      x[j] = make_double2(x0[e[j].it], x0[dim + e[j].it]);
      L.sx[e[j].it] = x[j];
    }
  __syncthreads();
  StepC<Q> c;
  load_step<Q>(ctlrow, c);
  double* y = yout + (size_t)b * 2 * dim;
#pragma unroll
  for (int j = 0; j < EPT; j++)
    if (e[j].it >= 0) {
      const double2 t = transpose ? apply_elem<Q, LIND, true>(S, e[j], L.sx, L.ssq, c, x[j])
                                  : apply_elem<Q, LIND, false>(S, e[j], L.sx, L.ssq, c, x[j]);
      y[e[j].it] = t.x;
      y[dim + e[j].it] = t.y;
    }
}


}  // namespace qd
