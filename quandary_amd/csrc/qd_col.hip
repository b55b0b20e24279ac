// qd_col.hip — lean column-per-wave sweeps for Lindblad systems whose density matrix has 33..64 rows (BASELINE config 4:
// 3 x 20 levels, N = 60, dim 3600, 3600 initial conditions).  gfx950 / CDNA4 only.
//
// Same element -> thread idea as ColStencil (qd_device.h): lane = row I of rho, a wave owns whole columns I', so every
// ket-side quantity is wave-uniform and every bra-side quantity a thread invariant.  What is different is the budget: the
// round-1 kernel ran 8 waves x 8 columns with 256 VGPRs (+ scratch) = two waves per SIMD that both wait at the same
// barrier.  Here one initial condition is spread over 16 waves x 4 columns and the hot loop holds only the iterate, the
// right-hand side, the new iterate and the thread's invariants (<= 128 VGPRs): four waves per SIMD.
//   * LDS: two exchange vectors with a column stride of 64 elements (the slot offset of a neighbour read is an immediate;
//     the in-column neighbours row +- post_k cost no address arithmetic at all), one ds_read_b128 per neighbour;
//   * ket-side neighbours (column +- post_k): own registers where the column belongs to the same wave's block and
//     post_k = 1, otherwise one v_add of a scalar column base;
//   * the state itself is parked in the output buffer while the linear solve runs (as the LEAN variants of qd_device.h).
// Reference semantics: stencil include/mastereq.hpp:316-912 (control / L1decay / L2), src/mastereq.cpp:1464-1709;
// IMR forward src/timestepper.cpp:584-630, Neumann :697-727; time loop :96-181, penalties :256-298.
#include <hip/hip_runtime.h>

#include "qd_device.h"

namespace qd {

constexpr int COL_NT = 1024, COL_NW = 16, COL_CPW = 4, COL_STRIDE = 64;  // threads, waves, columns per wave, LDS column stride (elements)
constexpr unsigned COL_CB = COL_STRIDE * sizeof(double2);                // bytes per LDS column

template <int Q>
struct ColLean {
  int N, row, col0;      // row of this lane (clamped), first column of this wave
  bool rowok;
  double dw[COL_CPW], dd[COL_CPW];  // Delta, d of (row, column slot)
  double su[Q], sd[Q];               // sqrt(i_k + 1) (0 at the top level), sqrt(i_k) of this thread's row
  double g1u[Q], g1d[Q];             // gamma_1 su, gamma_1 sd (T1 off-diagonal, forward / transposed)
  unsigned arow, aup[Q], adn[Q];     // byte offsets inside a column: own row, row + post_k, row - post_k (clamped)
  double p[Q], q[Q];                 // controls of the sub-step (wave-uniform)
  const double2* coltab;             // LDS: coltab[c * Q + k] = (sqrt(i'_k + 1) or 0 at the top level, sqrt(i'_k)) of column c

  __device__ __forceinline__ int colof(int j) const { return min(col0 + j, N - 1); }
  __device__ __forceinline__ bool valid(int j) const { return rowok && col0 + j < N; }

  __device__ __forceinline__ void init(const DevSys& S, double2* ctab) {
    N = S.N;
    const int lane = threadIdx.x & 63;
    col0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * COL_CPW;
    rowok = lane < N;
    row = rowok ? lane : N - 1;
    coltab = ctab;
    for (int e = threadIdx.x; e < N * Q; e += blockDim.x) {
      const int cc = e / Q, k = e % Q;
      const int ap = (cc / S.post[k]) % S.n[k];
      ctab[e] = make_double2((ap < S.n[k] - 1) ? sqrt((double)(ap + 1)) : 0.0, sqrt((double)ap));
    }
    int ia[Q];
    arow = (unsigned)row * sizeof(double2);
#pragma unroll
    for (int k = 0; k < Q; k++) {
      ia[k] = (row / S.post[k]) % S.n[k];
      su[k] = (ia[k] < S.n[k] - 1) ? sqrt((double)(ia[k] + 1)) : 0.0;
      sd[k] = sqrt((double)ia[k]);
      g1u[k] = S.g1off[k] * su[k];
      g1d[k] = S.g1off[k] * sd[k];
      aup[k] = (unsigned)min(row + S.post[k], N - 1) * sizeof(double2);
      adn[k] = (unsigned)max(row - S.post[k], 0) * sizeof(double2);
      p[k] = q[k] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < COL_CPW; j++) {
      const int cc = colof(j);
      double hd = 0.0, hdp = 0.0, d = 0.0;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int ipk = (cc / S.post[k]) % S.n[k];
        hd += S.detune[k] * ia[k] - S.xi[k] / 2.0 * ia[k] * (ia[k] - 1);
        hdp += S.detune[k] * ipk - S.xi[k] / 2.0 * ipk * (ipk - 1);
        d += S.g2[k] * (ia[k] * ipk - 0.5 * (ia[k] * ia[k] + ipk * ipk)) - S.g1[k] / 2.0 * (ia[k] + ipk);
#pragma unroll
        for (int l = k + 1; l < Q; l++) {
          const int ipl = (cc / S.post[l]) % S.n[l];
          hd -= S.xikl[pair] * ia[k] * ia[l];
          hdp -= S.xikl[pair] * ipk * ipl;
          pair++;
        }
      }
      dw[j] = hd - hdp;
      dd[j] = d;
    }
  }

  __device__ __forceinline__ void prep(const StepC<Q>& c) {
#pragma unroll
    for (int k = 0; k < Q; k++) {
      p[k] = to_scalar(c.p[k]);
      q[k] = to_scalar(c.q[k]);
    }
  }

  __device__ __forceinline__ static double2 lds(const double2* __restrict__ sx, unsigned byteoff) {
    return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(sx) + byteoff);
  }

  // isGuardLevel (util.cpp:259-278) for the diagonal element of slot j
  __device__ __forceinline__ bool is_guard(const DevSys& S, int j) const {
    if (!valid(j) || col0 + j != row) return false;
    bool g = false;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const int a = (row / S.post[k]) % S.n[k];
      g = g || (a == S.n[k] - 1 && a >= S.ness[k]);
    }
    return g;
  }

  // y = M x (TRANS = false) or M^T x for slot j; xall = the thread's elements of the vector in `sx` (GenStencil::apply for
  // the derivation: A = U1 + U2 - D1 - D2, B = U1 - U2 + D1 - D2, control part q A - i p B, T1 term on the (anti)diagonal neighbour)
  template <bool TRANS>
  __device__ __forceinline__ double2 apply(const DevSys& S, const double2* __restrict__ sx, int j, const double2 xs) const {
    const int cc = colof(j);
    const unsigned cbase = (unsigned)cc * COL_CB;  // own column (idle slots beyond N recompute column N-1 and are never stored)
    double hr = dw[j] * xs.y, hi = -dw[j] * xs.x;
    double l1r = 0.0, l1i = 0.0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const int st = S.post[k];
      const double2 ct = coltab[cc * Q + k];
      const unsigned cu = (unsigned)min(cc + st, N - 1) * COL_CB, cd = (unsigned)max(cc - st, 0) * COL_CB;  // scalar unit
      const double2 xu = lds(sx, cbase + aup[k]), xd = lds(sx, cbase + adn[k]);
      const double2 xup = lds(sx, cu + arow), xdp = lds(sx, cd + arow);
      const double er = fma(-ct.y, xdp.x, su[k] * xu.x), ei = fma(-ct.y, xdp.y, su[k] * xu.y);  // U1 - D2
      const double fr = fma(ct.x, xup.x, -sd[k] * xd.x), fi = fma(ct.x, xup.y, -sd[k] * xd.y);  // U2 - D1
      hr = fma(q[k], er + fr, fma(p[k], ei - fi, hr));
      hi = fma(q[k], ei + fi, fma(-p[k], er - fr, hi));
      const double2 xl = TRANS ? lds(sx, cd + adn[k]) : lds(sx, cu + aup[k]);
      const double l1 = TRANS ? g1d[k] * ct.y : g1u[k] * ct.x;
      l1r = fma(l1, xl.x, l1r);
      l1i = fma(l1, xl.y, l1i);
      if (k + 1 < Q) slot_fence<2>();  // one oscillator's five neighbour reads in flight at a time (four waves per SIMD hide the latency)
    }
    return make_double2(fma(dd[j], xs.x, TRANS ? -hr : hr) + l1r, fma(dd[j], xs.y, TRANS ? -hi : hi) + l1i);
  }
};

template <int Q>
struct TeamCol {
  ColLean<Q> st;
  double2* buf;  // two exchange vectors of N columns x COL_STRIDE elements
  double* red;
  int cur, redslot, vecsz;

  static size_t lds_bytes(const DevSys& S) {
    return 2 * sizeof(double2) * (size_t)S.N * COL_STRIDE + sizeof(double2) * (size_t)S.N * Q + 2 * sizeof(double) * NRED * COL_NW;
  }
  __device__ __forceinline__ void init(const DevSys& S, unsigned char* smem) {
    vecsz = S.N * COL_STRIDE;
    buf = reinterpret_cast<double2*>(smem);
    double2* ctab = buf + 2 * (size_t)vecsz;
    red = reinterpret_cast<double*>(ctab + S.N * Q);
    cur = 0;
    redslot = 0;
    st.init(S, ctab);
    // rows N .. 63 of every column are never written by a sweep but may be read through clamped offsets? No: offsets are
    // clamped to N-1.  Nothing to initialise beyond the tables.
    __syncthreads();
  }
  __device__ __forceinline__ int lidx(int j) const { return st.colof(j) * COL_STRIDE + st.row; }
  __device__ __forceinline__ const double2* vec() const { return buf + cur * vecsz; }
  __device__ __forceinline__ void publish(const double2 (&x)[COL_CPW]) {
    double2* dst = buf + (cur ^ 1) * vecsz;
#pragma unroll
    for (int j = 0; j < COL_CPW; j++)
      if (st.valid(j)) dst[lidx(j)] = x[j];
    cur ^= 1;
    __syncthreads();
  }
  template <bool TRANS>
  __device__ __forceinline__ void apply_all(const DevSys& S, const double2 (&x)[COL_CPW], double2 (&y)[COL_CPW]) const {
    const double2* sx = vec();
#pragma unroll
    for (int j = 0; j < COL_CPW; j++) {
      y[j] = st.template apply<TRANS>(S, sx, j, x[j]);
      slot_fence<COL_CPW>();
    }
  }
  template <int NV>
  __device__ __forceinline__ void sum(double (&v)[NV]) {
    block_sum<NV, false>(v, red + redslot * NRED * COL_NW);
    redslot ^= 1;
  }
  __device__ __forceinline__ float sum_f32(float v) {
    double* r = red + redslot * NRED * COL_NW;
    redslot ^= 1;
    return block_sum_f32<false>(v, r);
  }

  // Neumann iteration (timestepper.cpp:697-727).  The iterate lives in LDS only (own element and every neighbour are read
  // from the exchange vector, the new value goes straight to the other one): registers hold the right-hand side and the
  // thread's invariants, nothing per iterate.  On exit y in registers.  Returns the RHS applications.
  template <bool TRANS>
  __device__ __forceinline__ int neumann(const SweepArgs& A, double alpha, const double2 (&b)[COL_CPW], double2 (&y)[COL_CPW]) {
    publish(b);
    const double inv_abs2 = 1.0 / (A.abstol * A.abstol);
    const float rel2 = (float)(A.reltol * A.reltol);
    float d0 = 1.f;
    int iter;
    for (iter = 0; iter < A.maxiter; iter++) {
      const double2* src = vec();
      double2* dst = buf + (cur ^ 1) * vecsz;
      double dl = 0.0;
#pragma unroll
      for (int j = 0; j < COL_CPW; j++) {
        const double2 yo = src[lidx(j)];
        const double2 t = st.template apply<TRANS>(A.S, src, j, yo);
        double2 w;
        w.x = fma(alpha, t.x, b[j].x);
        w.y = fma(alpha, t.y, b[j].y);
        const double dx = yo.x - w.x, dy = yo.y - w.y;
        dl += st.valid(j) ? dx * dx + dy * dy : 0.0;
        if (st.valid(j)) dst[lidx(j)] = w;
        slot_fence<COL_CPW>();
      }
      const float d = sum_f32((float)fmin(dl * inv_abs2, 1e30));  // the barrier that makes dst readable
      cur ^= 1;
      if (iter == 0) d0 = d;
      if (d < 1.f) { iter++; break; }
      if (d < rel2 * d0) { iter++; break; }
    }
    const double2* fin = vec();
#pragma unroll
    for (int j = 0; j < COL_CPW; j++) y[j] = fin[lidx(j)];
    return iter;
  }
};

template <int Q>
__global__ void __launch_bounds__(COL_NT) k_forward_col(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef TeamCol<Q> TM;
  constexpr int EPT = COL_CPW;
  const DevSys& S = A.S;
  TM tm;
  tm.init(S, smem);
  const int dim = S.dim, N = S.N, ic = blockIdx.x;
  auto gidx = [&](int j) { return tm.st.colof(j) * N + tm.st.row; };  // index in the reference layout (vec(rho), column-major)
  double2 x[EPT];
  {
    const double* x0 = A.x0 + (size_t)ic * 2 * dim;
#pragma unroll
    for (int j = 0; j < EPT; j++) x[j] = make_double2(x0[gidx(j)], x0[dim + gidx(j)]);
  }
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  bool guard[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++) guard[j] = A.leak_on && tm.st.is_guard(S, j);
  double pen_local = 0.0, pen_uniform = 0.0;
  unsigned long long napply = 0;
  double* xpark = A.xT + (size_t)ic * 2 * dim;  // the state waits here while the linear solve runs

  for (int s = 0; s < A.nsub; s++) {
    StepC<Q> c;
    load_step<Q>(A.ctl + (size_t)s * A.cs, c, false);
    tm.st.prep(c);
    const double h = to_scalar(c.h);
    if (A.traj) {
      double* dst = A.traj + ((size_t)s * A.nb + ic) * 2 * dim;
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (tm.st.valid(j)) {
          dst[gidx(j)] = x[j].x;
          dst[dim + gidx(j)] = x[j].y;
        }
    }
    tm.publish(x);
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (tm.st.valid(j)) {
        const int e = opaque(gidx(j));
        xpark[e] = x[j].x;
        xpark[dim + e] = x[j].y;
      }
    double2 rhs[EPT], k[EPT];
    tm.template apply_all<false>(S, x, rhs);  // rhs = M x (ImplMidpoint::evolveFWD, timestepper.cpp:594)
    napply += 1 + tm.template neumann<false>(A, 0.5 * h, rhs, k);
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int e = opaque(gidx(j));
      x[j].x = fma(h, k[j].x, xpark[e]);
      x[j].y = fma(h, k[j].y, xpark[dim + e]);
    }
    // in-loop penalties at the end of a FULL time step (timestepper.cpp:141-154, :256-298)
    if (pen_on && (s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages - 1;
      const double tstop = (n + 1) * A.dt;
      if (wj_on) {
        const double a = (tstop - A.Tfinal) / A.penalty_param;
        const double weight = 1.0 / A.penalty_param * exp(-(a * a));
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (tm.st.valid(j)) {
            double jr = 0.0, ji = 0.0;
            evalJ_part<true>(S, A.tg, ic, opaque(gidx(j)), x[j], jr, ji);
            pen_local += (A.tg.objective_type == QD_OBJ_JTRACE ? -1.0 : 1.0) * weight * A.dt * jr;
          }
        if (A.tg.objective_type == QD_OBJ_JTRACE) pen_uniform += weight * A.dt;
      }
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (guard[j]) pen_local += (x[j].x * x[j].x + x[j].y * x[j].y) / A.ntime;
    }
  }
  {
    double* xT = A.xT + (size_t)ic * 2 * dim;
    double* dst = A.traj ? A.traj + ((size_t)A.nsub * A.nb + ic) * 2 * dim : nullptr;
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (tm.st.valid(j)) {
        xT[gidx(j)] = x[j].x;
        xT[dim + gidx(j)] = x[j].y;
        if (dst) {
          dst[gidx(j)] = x[j].x;
          dst[dim + gidx(j)] = x[j].y;
        }
      }
  }
  double v[1] = {pen_local};
  tm.template sum<1>(v);
  if (threadIdx.x == 0) {
    A.pen_out[ic] = v[0] + pen_uniform;
    A.dpdm_out[ic] = 0.0;
    atomicAdd(A.napply, napply);
  }
}

template <int Q>
__global__ void __launch_bounds__(COL_NT) k_apply_col(const DevSys S, const double* __restrict__ ctlrow, int transpose,
                                                      const double* __restrict__ xin, double* __restrict__ yout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  TeamCol<Q> tm;
  tm.init(S, smem);
  const int dim = S.dim, N = S.N, ic = blockIdx.x;
  double2 x[COL_CPW], y[COL_CPW];
  const double* x0 = xin + (size_t)ic * 2 * dim;
#pragma unroll
  for (int j = 0; j < COL_CPW; j++) {
    const int e = tm.st.colof(j) * N + tm.st.row;
    x[j] = make_double2(x0[e], x0[dim + e]);
  }
  StepC<Q> c;
  load_step<Q>(ctlrow, c, false);
  tm.st.prep(c);
  tm.publish(x);
  if (transpose) tm.template apply_all<true>(S, x, y);
  else tm.template apply_all<false>(S, x, y);
  double* yo = yout + (size_t)ic * 2 * dim;
#pragma unroll
  for (int j = 0; j < COL_CPW; j++)
    if (tm.st.valid(j)) {
      const int e = tm.st.colof(j) * N + tm.st.row;
      yo[e] = y[j].x;
      yo[dim + e] = y[j].y;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
bool col_lean_available(const DevSys& S) {
  if (!S.lindblad || S.dense || S.hasJ || S.Q < 1 || S.Q > 5) return false;
  if (S.N < 49 || S.N > 64) return false;  // 16 waves x 4 columns; below ~3/4 of the lanes the linear map wins
  if (S.post[S.Q - 1] != 1) return false;
  return !getenv("QD_NO_COL_LEAN");
}

template <int Q>
static hipError_t go_fwd_col(const SweepArgs& a, hipStream_t st) {
  const size_t lds = TeamCol<Q>::lds_bytes(a.S);
  auto kf = k_forward_col<Q>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3(a.nb), dim3(COL_NT), lds, st, a);
  return hipGetLastError();
}
template <int Q>
static hipError_t go_app_col(const DevSys& S, const double* ctlrow, int tr, const double* x, double* y, int nb, hipStream_t st) {
  const size_t lds = TeamCol<Q>::lds_bytes(S);
  auto kf = k_apply_col<Q>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3(nb), dim3(COL_NT), lds, st, S, ctlrow, tr, x, y);
  return hipGetLastError();
}

hipError_t launch_forward_col(const SweepArgs& a, hipStream_t st) {
  switch (a.S.Q) {
    case 1: return go_fwd_col<1>(a, st);
    case 2: return go_fwd_col<2>(a, st);
    case 3: return go_fwd_col<3>(a, st);
    case 4: return go_fwd_col<4>(a, st);
    case 5: return go_fwd_col<5>(a, st);
  }
  return hipErrorInvalidValue;
}
hipError_t launch_apply_col(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb, hipStream_t st) {
  switch (S.Q) {
    case 1: return go_app_col<1>(S, ctlrow, transpose, x, y, nb, st);
    case 2: return go_app_col<2>(S, ctlrow, transpose, x, y, nb, st);
    case 3: return go_app_col<3>(S, ctlrow, transpose, x, y, nb, st);
    case 4: return go_app_col<4>(S, ctlrow, transpose, x, y, nb, st);
    case 5: return go_app_col<5>(S, ctlrow, transpose, x, y, nb, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace qd
