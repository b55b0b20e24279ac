// qd_col.hip — lean column kernels: Lindblad sweeps of density matrices with 33 <= N <= 64 rows and runtime level counts
// (BASELINE config 4: 3 x 20 levels, dim 3600, 3600 initial conditions).  gfx950 / CDNA4 only.
//
// Layout (as ColStencil of qd_device.h: lane = row I of rho, a wave owns EPT consecutive columns I', vectorised index
// it = I' N + I, util.cpp:150), rewritten for register and issue economy:
//   * the exchange vector lives in LDS with a PADDED column stride of 64 rows (1 KiB per column), double buffered: the address of
//     every neighbour is a thread invariant plus a compile-time immediate (slot j of the wave = + j KiB) - no per-slot address
//     registers, nothing for the compiler to hoist and spill;
//   * a neighbour that does not exist has a zero coefficient, and its address is folded onto the element itself ONCE (thread
//     invariants for the bra side, wave-uniform scalar offsets for the ket side), so no clamping happens in the hot loop;
//   * idle lanes (row >= N) and idle slots (column >= N) carry zeros through the same arithmetic: no divergence, no predication
//     except on global memory;
//   * the bra neighbours of the stride-1 oscillator are the adjacent lanes (DPP), its ket neighbours the adjacent slots
//     (registers); the other oscillators read LDS: ~5 ds_read_b128 per element and application for two oscillators;
//   * the state x is parked in its output buffer while a linear solve runs; the solver holds b and the iterate, nothing else.
//
// Reference semantics (paths relative to the reference repository): stencil include/mastereq.hpp:316-912 as instantiated by
// src/mastereq.cpp:1464-1709 (two oscillators) / :1713-2018 (three); IMR forward / adjoint src/timestepper.cpp:584-694,
// Neumann :697-727, time loops :96-253, penalties :256-339, gradient coefficients include/mastereq.hpp:553-604.
#include <hip/hip_runtime.h>

#include "qd_device.h"

namespace qd {

constexpr unsigned COLB = 1024;  // bytes per padded column: 64 rows x 16 B
// Krylov solver of these kernels (ColTeam::kry_*): restart length.  Basis V[0 .. MR], preconditioned basis Z[0 .. MR - 1], the parked
// right-hand side, the parked total of a restart and the parked state: 2 MR + 4 = 32 vectors = the GMRES_MR_G + 2 slots per workgroup of
// SweepArgs::kry (krylov_doubles)
constexpr int KRY_MR = 14;
constexpr int KRY_NSC = gmres_nsc(KRY_MR);
constexpr size_t KRY_VEC = 64 * 64;  // double2 per scratch vector of a workgroup: the padded column layout (64 rows x at most 64 columns)
static_assert(2 * KRY_MR + 4 <= GMRES_MR_G + 2, "slots of the global-memory Krylov buffer");

// Largest workgroup of the EPT-columns-per-wave kernels = their register budget: 16 waves x 128, 12 x 168 (five columns per wave
// cover N <= 60), 11 x 168, 8 x 256 VGPRs
constexpr int col_max_threads(int ept) { return ept == 5 ? 768 : 64 * ((64 + ept - 1) / ept); }

__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// SPLIT: the kernels of the diagonal-split solver keep (1 - alpha D)^-1 per element instead of the diagonal itself; the diagonal is then
// re-derived where the full operator is applied (once or twice per step) from a row part held by the thread and a column part in LDS
// USLOT: every column of a wave has the same level indices i'_k of the oscillators k < L (post[k] a multiple of EPT, no idle columns):
// the byte offsets to the ket neighbour columns are then one pair per oscillator instead of one per slot, and the compiler forms each
// neighbour address once per application instead of once per slot (12 of 300 vector instructions of a solver iteration on 3 x 20).
template <int Q, int EPT, bool SPLIT = false, bool USLOT = false>
struct ColLean {
  static constexpr int L = Q - 1;  // the stride-1 oscillator (post[Q-1] == 1)
  // thread invariants (functions of the row)
  double su[Q], sd[Q];    // sqrt(i_k + 1) (0 at the top level), sqrt(i_k)
  double g1u[Q], g1d[Q];  // gamma_1 su / gamma_1 sd: thread part of the T1 off-diagonal coefficient, forward / transposed
  double dw[SPLIT ? 1 : EPT], dd[SPLIT ? 1 : EPT];  // Delta = h(I) - h(I'), d = L2 + L1diag of the element in slot j (mastereq.hpp:316-433)
  double hrow, drow, g2ia[Q];  // SPLIT: h(I), the row part of d, gamma_2 i_k
  unsigned ctb;                // SPLIT: LDS byte address of the wave's first entry of the column table (h(I'), column part of d, i'_k)
  unsigned tb;              // LDS byte address of (row, first column of the wave) in the buffer being READ
  unsigned aru[Q], ard[Q];  // the same with the row moved up / down by post[k] where that bra neighbour exists (else tb)
  int dlt;                  // byte distance from the buffer being read to the other one (+- bufbytes)
  // wave-uniform (functions of the wave's columns; scalar registers)
  double cx[EPT][Q], cy[EPT][Q];  // sqrt(i'_k + 1) (0 at the top level), sqrt(i'_k) of the column of slot j
  int ocu[USLOT ? 1 : EPT][Q], ocd[USLOT ? 1 : EPT][Q];  // byte offset to the ket neighbour column up / down (0 where there is none)
  static __device__ __forceinline__ constexpr int us(int j) { return USLOT ? 0 : j; }
  // (USLOT: the level indices of the oscillators k < L are the same in every column of the wave, hence their square roots too - one
  //  scalar pair per oscillator instead of one per slot: 16 scalar registers less on 3 x 20, where the adjoint sweep spills ~160)
  __device__ __forceinline__ double cxv(int j, int k) const { return cx[(USLOT && k != L) ? 0 : j][k]; }
  // (USLOT, stride-1 oscillator: the columns of a wave are consecutive levels i', i' + 1, ... of it, so sqrt(i'_j) = sqrt(i'_{j-1} + 1):
  //  the down coefficient of slot j is the up coefficient of slot j - 1 - four more scalar pairs less: 594.9 -> 592.5 ms, gradient
  //  1209 -> 1202 ms in one lease)
  __device__ __forceinline__ double cyv(int j, int k) const {
    if (USLOT && k == L && j > 0) return cx[j - 1][k];
    return cy[(USLOT && k != L) ? 0 : j][k];
  }
  int N, row, col0;
  bool rowok;
  unsigned char* smem;

  __device__ __forceinline__ double2 ld(unsigned a) const { return *reinterpret_cast<const double2*>(smem + a); }
  __device__ __forceinline__ void st(unsigned a, const double2 v) const { *reinterpret_cast<double2*>(smem + a) = v; }
  __device__ __forceinline__ int colof(int j) const { return col0 + j; }
  __device__ __forceinline__ bool colok(int j) const { return col0 + j < N; }
  __device__ __forceinline__ bool ok(int j) const { return rowok && colok(j); }
  __device__ __forceinline__ int elem(int j) const { return (col0 + j) * N + row; }  // vectorised index (valid slots only)
  // the same, re-derived at the point of use: hoisted out of the time loop the per-slot indices are spilt and every global access
  // of a step starts with a scratch reload
  __device__ __forceinline__ int elem_now(int j) const { return (col0 + j) * N + opaque(row); }
  static __host__ __device__ int ncols(int N) { return (N + EPT - 1) / EPT * EPT; }
  static __host__ __device__ unsigned bufbytes(int N) { return (unsigned)ncols(N) * COLB; }
  static __host__ __device__ unsigned tab_off(int N) { return 2 * bufbytes(N) + 2 * (unsigned)sizeof(double) * NRED * (unsigned)(ncols(N) / EPT) + 128; }
  static size_t lds_bytes(int N) { return (size_t)tab_off(N) + 48 * (size_t)ncols(N); }

  __device__ __forceinline__ void init(const DevSys& S, unsigned char* sm) {
    smem = sm;
    N = S.N;
    const int lane = threadIdx.x & 63;
    const int w = uniform_i((int)(threadIdx.x >> 6));
    col0 = w * EPT;
    row = lane;
    rowok = lane < N;
    // zero the exchange buffers once: padding rows and idle columns are read (with zero coefficients) and must stay finite
    {
      const unsigned total = 2 * bufbytes(N);
      for (unsigned a = threadIdx.x * 16u; a < total; a += blockDim.x * 16u) st(a, make_double2(0.0, 0.0));
    }
    tb = (unsigned)col0 * COLB + (unsigned)lane * 16u;
    dlt = (int)bufbytes(N);
    int ia[Q];
    double hd = 0.0;  // h(I)
#pragma unroll
    for (int k = 0; k < Q; k++) {
      ia[k] = rowok ? (row / S.post[k]) % S.n[k] : 0;
      su[k] = (rowok && ia[k] < S.n[k] - 1) ? sqrt((double)(ia[k] + 1)) : 0.0;
      sd[k] = rowok ? sqrt((double)ia[k]) : 0.0;
      g1u[k] = S.g1off[k] * su[k];
      g1d[k] = S.g1off[k] * sd[k];
      aru[k] = tb + (su[k] != 0.0 ? (unsigned)S.post[k] * 16u : 0u);
      ard[k] = tb - (sd[k] != 0.0 ? (unsigned)S.post[k] * 16u : 0u);
    }
    {
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        hd += S.detune[k] * ia[k] - S.xi[k] / 2.0 * ia[k] * (ia[k] - 1);
#pragma unroll
        for (int l = k + 1; l < Q; l++) hd -= S.xikl[pair++] * ia[k] * ia[l];
      }
    }
    hrow = rowok ? hd : 0.0;
    drow = 0.0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      drow -= 0.5 * S.g2[k] * ia[k] * ia[k] + 0.5 * S.g1[k] * ia[k];
      g2ia[k] = rowok ? S.g2[k] * ia[k] : 0.0;
    }
    if (!rowok) drow = 0.0;
    ctb = tab_off(N) + (unsigned)col0 * 48u;
    if (SPLIT) {  // column table: (h(I'), column part of d), (i'_0, i'_1), (i'_2, 0); zeros for idle columns
      for (int cc = threadIdx.x; cc < ncols(N); cc += blockDim.x) {
        double hc = 0.0, dc = 0.0, ip[3] = {0.0, 0.0, 0.0};
        if (cc < N) {
          int ipa[Q], pair = 0;
#pragma unroll
          for (int k = 0; k < Q; k++) ipa[k] = (cc / S.post[k]) % S.n[k];
#pragma unroll
          for (int k = 0; k < Q; k++) {
            hc += S.detune[k] * ipa[k] - S.xi[k] / 2.0 * ipa[k] * (ipa[k] - 1);
            dc -= 0.5 * S.g2[k] * ipa[k] * ipa[k] + 0.5 * S.g1[k] * ipa[k];
            ip[k] = (double)ipa[k];
#pragma unroll
            for (int l = k + 1; l < Q; l++) hc -= S.xikl[pair++] * ipa[k] * ipa[l];
          }
        }
        double2* t = reinterpret_cast<double2*>(smem + tab_off(N) + (unsigned)cc * 48u);
        t[0] = make_double2(hc, dc);
        t[1] = make_double2(ip[0], ip[1]);
        t[2] = make_double2(ip[2], 0.0);
      }
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int cc = col0 + j;
      const bool cok = cc < N;
      int ipa[Q];
      double hdp = 0.0, d = 0.0;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) ipa[k] = cok ? (cc / S.post[k]) % S.n[k] : 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        hdp += S.detune[k] * ipa[k] - S.xi[k] / 2.0 * ipa[k] * (ipa[k] - 1);
        d += S.g2[k] * (ia[k] * ipa[k] - 0.5 * (ia[k] * ia[k] + ipa[k] * ipa[k])) - S.g1[k] / 2.0 * (ia[k] + ipa[k]);
#pragma unroll
        for (int l = k + 1; l < Q; l++) hdp -= S.xikl[pair++] * ipa[k] * ipa[l];
        const bool up = cok && ipa[k] < S.n[k] - 1, dn = cok && ipa[k] > 0;
        cx[j][k] = to_scalar(up ? sqrt((double)(ipa[k] + 1)) : 0.0);
        cy[j][k] = to_scalar(dn ? sqrt((double)ipa[k]) : 0.0);
        if (!USLOT || j == 0 || k == L) {
          // (USLOT: the offsets of the stride-1 oscillator are only used at the two edge slots, whose neighbours are not in registers:
          //  the up offset of the last slot, the down offset of the first)
          if (!USLOT || k != L || j == 0) ocd[us(j)][k] = uniform_i(dn ? -S.post[k] * (int)COLB : 0);
          if (!USLOT || k != L || j == EPT - 1 || EPT == 1) ocu[us(j)][k] = uniform_i(up ? S.post[k] * (int)COLB : 0);
        }
      }
      const bool live = rowok && cok;
      if (!SPLIT) {
        dw[j] = live ? hd - hdp : 0.0;
        dd[j] = live ? d : 0.0;
      }
    }
  }

  // diagonal of M at the element in slot j: (Delta, d)
  __device__ __forceinline__ void diag(int j, double& dwj, double& ddj) const {
    if (!SPLIT) {
      dwj = dw[j];
      ddj = dd[j];
    } else {
      const double2 t0 = ld(ctb + (unsigned)j * 48u), t1 = ld(ctb + (unsigned)j * 48u + 16u);
      dwj = hrow - t0.x;
      double d = drow + t0.y;
      d = fma(g2ia[0], t1.x, d);
      if (Q > 1) d = fma(g2ia[Q > 1 ? 1 : 0], t1.y, d);
      if (Q > 2) d = fma(g2ia[Q > 2 ? 2 : 0], ld(ctb + (unsigned)j * 48u + 32u).x, d);
      ddj = rowok ? d : 0.0;
      if (!rowok) dwj = 0.0;
    }
  }

  // the other buffer becomes the one being read
  __device__ __forceinline__ void flip() {
    tb += (unsigned)dlt;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      aru[k] += (unsigned)dlt;
      ard[k] += (unsigned)dlt;
    }
    dlt = -dlt;
  }

  // the four ladder neighbours of oscillator k of the element in slot j: bra up / down (xu, xd), ket up / down (xup, xdp);
  // own / prev / next = the thread's elements of the vector being read in slots j, j - 1, j + 1
  __device__ __forceinline__ void nbrs(int k, int j, const double2 own, const double2 prev, const double2 next, double2& xu, double2& xd,
                                       double2& xup, double2& xdp) const {
    if (k == L) {
      xu = lane_shift<true>(own);
      xd = lane_shift<false>(own);
      xup = j < EPT - 1 ? next : ld(tb + (unsigned)ocu[us(j)][k] + (unsigned)j * COLB);
      xdp = j > 0 ? prev : ld(tb + (unsigned)ocd[us(j)][k] + (unsigned)j * COLB);
    } else {
      xu = ld(aru[k] + (unsigned)j * COLB);
      xd = ld(ard[k] + (unsigned)j * COLB);
      xup = ld(tb + (unsigned)ocu[us(j)][k] + (unsigned)j * COLB);
      xdp = ld(tb + (unsigned)ocd[us(j)][k] + (unsigned)j * COLB);
    }
  }

  // y = M x (TRANS = false) or M^T x at slot j (ColStencil::apply of qd_device.h; HASJ = false)
  // NODIAG: only the off-diagonal part C = M - diag(M) (the diagonal-split solver applies the diagonal in closed form)
  template <bool TRANS, bool NODIAG = false>
  __device__ __forceinline__ double2 apply(const StepC<Q>& c, int j, const double2 own, const double2 prev, const double2 next) const {
    double ar = 0.0, ai = 0.0;
    if (!NODIAG) {
      double dwj, ddj;
      diag(j, dwj, ddj);
      if (TRANS) dwj = -dwj;
      ar = fma(dwj, own.y, ddj * own.x);
      ai = fma(-dwj, own.x, ddj * own.y);
    }
#pragma unroll
    for (int k = 0; k < Q; k++) {
      double2 xu, xd, xup, xdp;
      nbrs(k, j, own, prev, next, xu, xd, xup, xdp);
      const double er = fma(-cyv(j, k), xdp.x, su[k] * xu.x), ei = fma(-cyv(j, k), xdp.y, su[k] * xu.y);  // U1 - D2
      const double fr = fma(cxv(j, k), xup.x, -sd[k] * xd.x), fi = fma(cxv(j, k), xup.y, -sd[k] * xd.y);  // U2 - D1
      const double pk = TRANS ? -c.p[k] : c.p[k], qk = TRANS ? -c.q[k] : c.q[k];
      ar = fma(qk, er + fr, fma(pk, ei - fi, ar));
      ai = fma(qk, ei + fi, fma(-pk, er - fr, ai));
      // T1 off-diagonal term: forward couples to (row + s, column + s), transposed to (row - s, column - s)
      double2 xl;
      if (k == L) {
        if (TRANS) xl = j > 0 ? lane_shift<false>(prev) : ld(ard[k] + (unsigned)ocd[us(j)][k] + (unsigned)j * COLB);
        else xl = j < EPT - 1 ? lane_shift<true>(next) : ld(aru[k] + (unsigned)ocu[us(j)][k] + (unsigned)j * COLB);
      } else {
        xl = TRANS ? ld(ard[k] + (unsigned)ocd[us(j)][k] + (unsigned)j * COLB) : ld(aru[k] + (unsigned)ocu[us(j)][k] + (unsigned)j * COLB);
      }
      const double l1 = TRANS ? g1d[k] * cyv(j, k) : g1u[k] * cxv(j, k);
      ar = fma(l1, xl.x, ar);
      ai = fma(l1, xl.y, ai);
    }
    return make_double2(ar, ai);
  }

  // gradient contraction (ColStencil::ladder): A = e + f, B = e - f with e = U1 - D2, f = U2 - D1 of the published vector
  __device__ __forceinline__ void ladder(int k, int j, const double2 own, const double2 prev, const double2 next, double2& A, double2& B) const {
    double2 xu, xd, xup, xdp;
    nbrs(k, j, own, prev, next, xu, xd, xup, xdp);
    const double er = fma(-cyv(j, k), xdp.x, su[k] * xu.x), ei = fma(-cyv(j, k), xdp.y, su[k] * xu.y);
    const double fr = fma(cxv(j, k), xup.x, -sd[k] * xd.x), fi = fma(cxv(j, k), xup.y, -sd[k] * xd.y);
    A.x = er + fr;
    A.y = ei + fi;
    B.x = er - fr;
    B.y = ei - fi;
  }

  // isGuardLevel (util.cpp:259-278) of the row's level combination; the leakage term sums the DIAGONAL elements of those rows
  __device__ __forceinline__ bool row_is_guard(const DevSys& S) const {
    bool g = false;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const int a = rowok ? (row / S.post[k]) % S.n[k] : 0;
      g = g || (a == S.n[k] - 1 && a >= S.ness[k]);
    }
    return g && rowok;
  }
};

typedef double col_d2 __attribute__((ext_vector_type(2)));

// per-workgroup machinery: buffers, reductions, the Neumann solver
// SKIP: the solver skips stopping tests (stage / neumann below) - an instantiation of its own (a request with a relative tolerance
// that can bind, rel2 >= 1e-30, keeps the test-every-pass kernels; with both forms in one kernel the second code path cost 11 %)
template <int Q, int EPT, bool SPLIT = false, bool USLOT = false, bool SKIP = false>
struct ColTeam {
  typedef ColLean<Q, EPT, SPLIT, USLOT> ST;
  ST st;
  double* red;
  float4* fred;  // two slots of 16 partial sums of the solver's fp32 norm reduction
  int redslot, nw;
  // diagonal-split solver: P = (1 - alpha D)^-1 of the thread's elements, D = diag(M) = d - i Delta (transposed: d + i Delta),
  // for the step size palpha (recomputed when the step size changes: composite steppers)
  double pr[SPLIT ? EPT : 1], pi[SPLIT ? EPT : 1], palpha;
  int lastn, lastna;  // passes of the previous forward sub-step (stage) / iterations of the previous linear solve (neumann)

  __device__ __forceinline__ void init(const DevSys& S, unsigned char* smem) {
    st.init(S, smem);
    red = reinterpret_cast<double*>(smem + 2 * ST::bufbytes(S.N));
    redslot = 0;
    nw = (int)(blockDim.x >> 6);
    palpha = 0.0;
    lastn = lastna = 0;
#pragma unroll
    for (int j = 0; j < (SPLIT ? EPT : 1); j++) {
      pr[j] = 1.0;
      pi[j] = 0.0;
    }
    fred = reinterpret_cast<float4*>(red + 2 * NRED * nw);
    if (threadIdx.x < 32) reinterpret_cast<float*>(fred)[threadIdx.x] = 0.f;  // (16 partial sums are read whatever the number of waves)
    __syncthreads();  // zero fill complete
  }

  // x becomes the vector being read
  __device__ __forceinline__ void publish(const double2 (&x)[EPT]) {
    const unsigned wa = st.tb + (unsigned)st.dlt;
#pragma unroll
    for (int j = 0; j < EPT; j++) st.st(wa + (unsigned)j * COLB, x[j]);
    st.flip();
    __syncthreads();
  }

  template <int NV>
  __device__ __forceinline__ void sum(double (&v)[NV]) {
    block_sum<NV, false>(v, red + redslot * NRED * nw);
    redslot ^= 1;
  }
  // Workgroup sum of NV <= 4 values for the Krylov solver [r6]: the wave level is a reduce-scatter (row r of 16 lanes ends up with the
  // total of value r: ~30 vector instructions for three values where three wave_sum()s are ~70), and behind the barrier ONE LDS read per
  // lane - lane 16 i + w fetches the partial sum of value i of wave w - and four DPP adds inside the rows replace the nw x NV broadcast
  // reads and dependent adds of block_sum (36 + 36 for three values on twelve waves).  Every thread returns the same bits.
  template <int NV>
  __device__ __forceinline__ void sum_rows(double (&v)[NV]) {
    static_assert(NV <= 4, "one value per row of 16 lanes");
    double* r = red + redslot * NRED * nw;  // (NRED nw >= 64 doubles from four waves on)
    redslot ^= 1;
    double o[1];
    wave_reduce_scatter<NV>(v, o);
    const int lane = (int)(threadIdx.x & 63);
    if ((lane & 15) == 0) {
      const int g = wave_scatter_index<NV>(lane >> 4, 0);
      if (g >= 0) r[g * 16 + (int)(threadIdx.x >> 6)] = o[0];
    }
    __syncthreads();
    double t = ((lane >> 4) < NV && (lane & 15) < nw) ? r[lane] : 0.0;
    t += dpp_mov<0xB1>(t);
    t += dpp_mov<0x4E>(t);
    t += dpp_mov<0x124>(t);
    t += dpp_mov<0x128>(t);
    const int lo = __double2loint(t), hi = __double2hiint(t);
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * i), __builtin_amdgcn_readlane(lo, 16 * i));
  }
  // Workgroup sum in two halves, for values only a few threads need (the 2Q gradient coefficients of a step, written by threads 0 .. 2Q - 1):
  // every wave leaves its partial sums in LDS (wave_reduce_scatter: the total of value g ends up in one row of 16 lanes, 21 vector
  // instructions for the four values of a two-oscillator system where four wave_sum()s are ~120); AFTER a later barrier of the caller (the
  // one that publishes the next vector) thread i adds the partial sums of value i in wave order.  Saves the reduction's own barrier and the
  // nw x NV broadcast reads of every thread.
  double* pend;
  template <int NV>
  __device__ __forceinline__ void sum_post(const double (&v)[NV]) {
    pend = red + redslot * NRED * nw;
    redslot ^= 1;
    constexpr int K = ((NV + 1) / 2 + 1) / 2;
    double o[K];
    wave_reduce_scatter<NV>(v, o);
    const int lane = (int)(threadIdx.x & 63);
    if ((lane & 15) == 0) {
      const int wave = (int)(threadIdx.x >> 6);
#pragma unroll
      for (int m = 0; m < K; m++) {
        const int g = wave_scatter_index<NV>(lane >> 4, m);
        if (g >= 0) pend[g * nw + wave] = o[m];
      }
    }
  }
  // (call after a __syncthreads() that follows sum_post; thread i < NV returns the sum of value i)
  __device__ __forceinline__ double sum_collect(int i) const {
    double t = 0.0;
    for (int w = 0; w < nw; w++) t += pend[i * nw + w];
    return t;
  }
  // workgroup sum of the solver's squared update norm (fp32; only compared with a threshold).  One barrier - the one that makes the
  // new iterate readable - and ONE round of LDS latency: the <= 16 partial sums are fetched by four broadcast reads and added as a tree
  // (a loop over the waves would chain 15 dependent LDS round trips in front of every stopping test).
  __device__ __forceinline__ float sum_f32(float v) {
    float4* rf = fred + redslot * 4;
    redslot ^= 1;
    v = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) reinterpret_cast<float*>(rf)[threadIdx.x >> 6] = v;
    __syncthreads();
    const float4 a = rf[0], b = rf[1], c = rf[2], d = rf[3];
    return (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
  }

  template <bool TRANS>
  __device__ __forceinline__ void apply_all(const StepC<Q>& c, const double2 (&x)[EPT], double2 (&y)[EPT]) const {
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      y[j] = st.template apply<TRANS>(c, j, x[j], x[j > 0 ? j - 1 : 0], x[j + 1 < EPT ? j + 1 : j]);
      slot_fence<EPT>();
    }
  }

  template <bool TRANS>
  __device__ __forceinline__ void set_alpha(double alpha) {
    if (alpha == palpha) return;  // (uniform)
    palpha = alpha;
#pragma unroll
    for (int j = 0; j < (SPLIT ? EPT : 0); j++) {
      double dwj, ddj;
      st.diag(j, dwj, ddj);
      const double re = fma(-alpha, ddj, 1.0), im = (TRANS ? -alpha : alpha) * dwj;  // 1 - alpha D
      const double inv = 1.0 / fma(re, re, im * im);
      pr[j] = re * inv;
      pi[j] = -im * inv;
    }
  }

  // Solve (I - alpha M^{(T)}) y = b.  Returns the number of RHS applications; y in registers.
  // SPLIT = false: the reference's Neumann iteration y <- b + alpha M y (timestepper.cpp:697-727), started at y = b.
  // SPLIT = true: the same fixed point and the same stopping rule on the update norm, with the diagonal of M taken to the left-hand
  // side: y <- (1 - alpha D)^-1 (b + alpha (M - D) y), started at (1 - alpha D)^-1 b.  D carries the level energies (self- and
  // cross-Kerr shifts, detuning) and the diagonal decay: where those dominate the control Hamiltonian the contraction factor drops
  // from ||alpha M|| to ~||alpha (M - D)|| (3 x 20 workload: 12.4 -> 8.x iterations per solve) at the same cost per iteration.
  // The squared update norm is reduced in fp32 exactly as Team::neumann of qd_device.h.
  template <bool TRANS>
  __device__ __forceinline__ int neumann(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2 (&b)[EPT], double2 (&y)[EPT]) {
    // (set_alpha<TRANS>(alpha) has been called at the top of the step: no control flow between an operator application and its use)
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      if (SPLIT) y[j] = make_double2(fma(pr[SPLIT ? j : 0], b[j].x, -pi[SPLIT ? j : 0] * b[j].y), fma(pr[SPLIT ? j : 0], b[j].y, pi[SPLIT ? j : 0] * b[j].x));
      else y[j] = b[j];
    }
    const double inv_abs2 = A.inv_abs2;
    float rel2 = A.rel2, thr = 1.f;
    if (A.stop_residual) {
      // in place of GMRES (qd_handle::gmres_as_split): stop when kappa^2 ||y_{m+1} - y_m||^2 <= max(rtol^2 ||b||^2, abstol^2)
      double nb2[1] = {0.0};
#pragma unroll
      for (int j = 0; j < EPT; j++) nb2[0] = fma(b[j].x, b[j].x, fma(b[j].y, b[j].y, nb2[0]));
      sum<1>(nb2);
      thr = (float)fmin(fmax(A.reltol * A.reltol * nb2[0] * inv_abs2, 1.0) / A.kappa2, 1e30);
      rel2 = 0.f;
    }
    publish(y);
    float d0 = 1.f, dprev = 1.f;
    const int skip = SKIP ? lastna - (A.standin_tau2 != 0.f ? 3 : 2) : 0;  // (one more tested pass where the error estimate needs a predecessor)
    int iter;
    for (iter = 0; iter < A.maxiter; iter++) {
      const unsigned wa = st.tb + (unsigned)st.dlt;
      double dl = 0.0;
      double2 prev = y[0];
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const double2 own = y[j];
        const double2 t = st.template apply<TRANS, SPLIT>(c, j, own, prev, y[j + 1 < EPT ? j + 1 : j]);
        double2 w;
        w.x = fma(alpha, t.x, b[j].x);
        w.y = fma(alpha, t.y, b[j].y);
        if (SPLIT) w = make_double2(fma(pr[SPLIT ? j : 0], w.x, -pi[SPLIT ? j : 0] * w.y), fma(pr[SPLIT ? j : 0], w.y, pi[SPLIT ? j : 0] * w.x));
        const double dx = own.x - w.x, dy = own.y - w.y;
        dl = fma(dx, dx, fma(dy, dy, dl));
        prev = own;
        y[j] = w;
        st.st(wa + (unsigned)j * COLB, w);
        slot_fence<EPT>();
      }
      float d = 1e30f, dp;
      if constexpr (SKIP) {  // (see stage())
        const bool test = iter >= skip;
        if (test) d = sum_f32((float)fmin(dl * inv_abs2, 1e30));  // contains the barrier that makes the new iterate readable
        else __syncthreads();
        st.flip();
        if (!test) continue;
        d0 = (iter == 0 || iter == skip) ? d : d0;
        dp = (iter == 0 || iter == skip) ? d : dprev;
      } else {
        d = sum_f32((float)fmin(dl * inv_abs2, 1e30));  // contains the barrier that makes the new iterate readable
        st.flip();
        // (one exit branch per pass, first-iteration values by selects [r5]: 637.6 -> 632.3 ms on the 3600 x 2500 sweep, same counts)
        d0 = iter == 0 ? d : d0;
        dp = iter == 0 ? d : dprev;
      }
      const bool stop = (d < thr && standin_ok(A.standin_tau2, d, dp, thr)) | (d < rel2 * d0);
      dprev = d;
      if (stop) { iter++; break; }
    }
    if (SKIP) lastna = iter;
    return iter;
  }

  // Forward sub-step in STAGE form.  The reference solves (I - alpha M) k = M x and sets x += h k (ImplMidpoint::evolveFWD,
  // timestepper.cpp:594-629, alpha = h / 2).  The iterates of its Neumann solver, y_0 = b = M x, y_{m+1} = b + alpha M y_m (SPLIT:
  // y_0 = P b, y_{m+1} = P (b + alpha C y_m), P = (1 - alpha D)^-1, C = M - D), map one to one onto iterates of the stage
  // z = x + alpha k:   z_m = x + alpha y_m   satisfies   z_{-1} = x,  z_m = x + alpha M z_{m-1}   (SPLIT: z_m = P (x + alpha C z_{m-1})),
  // (SPLIT: P x - x = alpha P D x, hence z_0 = P (x + alpha C x) = x + alpha P (D + C) x = x + alpha y_0.)  So the application that
  // forms b IS the first pass of the same loop, x itself is the right-hand side - it stays in registers and is never parked - and no
  // separate operator application, no b, no y_0 = P b is needed.  The update norms agree up to the factor alpha:
  // ||y_m - y_{m-1}|| = ||z_m - z_{m-1}|| / alpha, tested from the second pass on against the same thresholds.  On exit z = x + alpha k
  // (the primal stage the adjoint sweep reads) and x_{n+1} = 2 z - x.  Returns the RHS applications (passes).
  // In place of GMRES (A.stop_residual): threshold max(rtol^2 ||b||^2, abstol^2) / kappa^2 with ||b||^2 >= ||y_0||^2 (|1 - alpha D| >= 1:
  // the diagonal of M has a non-positive real part) taken from the first pass - never looser than the rule it stands for.
  __device__ __forceinline__ int stage(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2 (&x)[EPT], double2 (&z)[EPT]) {
    const double sc = A.inv_abs2 / (alpha * alpha);
    float rel2 = A.rel2, thr = 1.f, d0 = 1.f, dprev = 1.f;
#pragma unroll
    for (int j = 0; j < EPT; j++) z[j] = x[j];
    const int skip = SKIP ? lastn - (A.standin_tau2 != 0.f ? 4 : 3) : 0;
    int iter;
    for (iter = -1; iter < A.maxiter; iter++) {
      const unsigned wa = st.tb + (unsigned)st.dlt;
      double dl = 0.0;
      double2 prev = z[0];
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const double2 own = z[j];
        const double2 t = st.template apply<false, SPLIT>(c, j, own, prev, z[j + 1 < EPT ? j + 1 : j]);
        double2 w;
        w.x = fma(alpha, t.x, x[j].x);
        w.y = fma(alpha, t.y, x[j].y);
        if (SPLIT) w = make_double2(fma(pr[SPLIT ? j : 0], w.x, -pi[SPLIT ? j : 0] * w.y), fma(pr[SPLIT ? j : 0], w.y, pi[SPLIT ? j : 0] * w.x));
        const double dx = own.x - w.x, dy = own.y - w.y;
        dl = fma(dx, dx, fma(dy, dy, dl));
        prev = own;
        z[j] = w;
        st.st(wa + (unsigned)j * COLB, w);
        slot_fence<EPT>();
      }
      // [r5] SKIP: the reduction of the update norm costs ~25 vector instructions and an LDS round trip behind the barrier.  Consecutive
      // sub-steps converge after the same number of passes (the controls move slowly): under the reference's plain update-norm rule the
      // passes up to two before the count of the previous sub-step only synchronise.  A solve that would have stopped earlier runs on to
      // the first tested pass: more passes than the reference, never fewer (3600 x 2500 sweep: 8.238 -> 8.240 passes per step, 635 -> 611 ms).
      // Under the rule that stands in for GMRES the error estimate of a tested pass needs the norm of its predecessor: one more pass is
      // tested there (without it the first tested pass compares with itself and passes are lost: 8.98 -> 9.60, 685 -> 697 ms; with it
      // 8.975 -> 8.979 passes, 675 -> 655 ms), and its first pass is reduced for ||y_0||.
      bool test = true;
      float d = 1e30f;
      if constexpr (SKIP) {
        test = iter < 0 ? A.stop_residual != 0 : iter >= skip;
        if (test) d = sum_f32((float)fmin(dl * sc, 1e30));  // contains the barrier that makes the new iterate readable
        else __syncthreads();
      } else {
        d = sum_f32((float)fmin(dl * sc, 1e30));  // contains the barrier that makes the new iterate readable
      }
      st.flip();
      if (iter < 0) {  // first pass: d = ||y_0||^2 / abstol^2
        if (A.stop_residual) {
          thr = (float)fmin(fmax(A.reltol * A.reltol * (double)d, 1.0) / A.kappa2, 1e30);  // (d is capped at 1e30: conservative)
          rel2 = 0.f;
        }
        continue;
      }
      if (SKIP && !test) continue;
      // (one exit branch per pass, first-iteration values by selects [r5]: 637.6 -> 632.3 ms on the 3600 x 2500 sweep, same counts)
      const bool first = iter == 0 || (SKIP && iter == skip);
      d0 = first ? d : d0;
      const float dp = first ? d : dprev;
      const bool stop = (d < thr && standin_ok(A.standin_tau2, d, dp, thr)) | (d < rel2 * d0);
      dprev = d;
      if (stop) { iter++; break; }
    }
    if (SKIP) lastn = iter + 1;
    return iter + 1;
  }

  // ---------------------------------------------------------------------------------------------
  // Krylov solver of the lean column kernels [r6]: linearsolver_type = gmres (KSPGMRES, src/timestepper.cpp:541-550, call sites :602,
  // :652, :674) where the stationary iteration does not stand in for it (option gmres_split = 0, or its gate has failed).
  //
  // GMRES on (I - alpha M) y = b, right-preconditioned with the polynomial of the diagonal-split iteration:
  //   I - alpha M = (I - alpha D) - alpha C,  P = (I - alpha D)^-1,  R_p = sum_{i<p} (P alpha C)^i P,  (I - alpha M) R_p = I - (alpha C P)^p.
  // The residual of the preconditioned system IS b - (I - alpha M) y, so the reference's stopping rule - residual <= max(rtol ||b||,
  // abstol) - is unchanged.  z = R_p v is Horner's rule z <- P (v + alpha C z) from z = P v: the SAME pass as the stationary iteration
  // (one NODIAG application, the iterate and the right-hand side in registers, nothing else live), without its reductions.
  //
  // Hot path (one Krylov vector suffices: the host tunes p for that, qd_handle::forward_finish): p passes, then ONE full application
  // w = (I - alpha M) z fused with the three dot products <b,b>, <r,b>, <r,r> of r = b - w in one workgroup reduction.  With v_0 = b / beta:
  //   h_00 = <w, v_0> = 1 - a,  a = <r,b> / <b,b>;   h_10^2 = ||w - h_00 v_0||^2 = <r,r> / <b,b> - a^2   (from r, not from ||w||^2 - h_00^2,
  //   which cancels to nothing at residuals of 1e-10);   y = h_00 / (h_00^2 + h_10^2) z,   residual = beta h_10 / sqrt(h_00^2 + h_10^2).
  // No basis vector is written, b never leaves its registers (adjoint) / is parked once per step (forward, stage form below).
  // Slow path (residual above the tolerance after one vector): the solve starts over in kry_generic - classical Gram-Schmidt, Givens
  // rotations, restart KRY_MR, every vector (basis V, preconditioned basis Z = R_p V, the parked right-hand side) in this WORKGROUP's 32
  // slots of SweepArgs::kry, each thread touching its own elements only.  Rare by construction of p; its cost is its own.
  // ---------------------------------------------------------------------------------------------
  double2* wg;  // this workgroup's scratch vectors in global memory, at this thread's element of slot 0: vector s, slot j = wg[s KRY_VEC + 64 j]
  // (the vectors keep the PADDED column layout of the exchange buffers - 64 rows per column: idle lanes and idle slots park their zeros
  //  like everybody else, no guard, no exec mask inside a slot loop, and one base address with immediate offsets per vector)
  double* ksc;  // Hessenberg scalars of kry_generic (LDS, behind the column table)
  // Forward hot path: b = M x is formed by the first pass and used by the last one for <b,b> and <r,b> only - quantities that enter
  // the solution as 1 - <r,b>/<b,b> with <r,b>/<b,b> ~ 1e-10: an fp32 copy serves.  Five columns per wave (N <= 60) leave room for it in
  // LDS (8 B per element in the padded layout, 30 KB); eight columns per wave (N = 61 .. 64) park b in slot SB, fetched two slots ahead.
  static constexpr bool BLDS = EPT == 5;
  unsigned b32;  // LDS byte address of this thread's element of slot 0 of the fp32 copy
  static __host__ __device__ size_t kry_lds_extra(int N) { return sizeof(double) * KRY_NSC + (BLDS ? 512u * (size_t)ST::ncols(N) : 0u); }
  static constexpr int SV = 0, SZ = KRY_MR + 1, SB = 2 * KRY_MR + 1, SY = 2 * KRY_MR + 2, SX = 2 * KRY_MR + 3;

  __device__ __forceinline__ void init_kry(const SweepArgs& A) {
    wg = reinterpret_cast<double2*>(A.kry) + (size_t)blockIdx.x * (GMRES_MR_G + 2) * KRY_VEC + (size_t)st.col0 * 64 + (threadIdx.x & 63);
    ksc = reinterpret_cast<double*>(st.smem + ST::tab_off(A.S.N) + 48u * (unsigned)ST::ncols(A.S.N));
    b32 = ST::tab_off(A.S.N) + 48u * (unsigned)ST::ncols(A.S.N) + (unsigned)sizeof(double) * KRY_NSC + (unsigned)st.col0 * 512u + (threadIdx.x & 63) * 8u;
  }
  __device__ __forceinline__ double2* vec(int s) const { return wg + (size_t)s * KRY_VEC; }
  __device__ __forceinline__ void vstore(int s, const double2 (&v)[EPT]) const {
    double2* p = vec(s);
#pragma unroll
    for (int j = 0; j < EPT; j++) p[64 * j] = v[j];
  }
  __device__ __forceinline__ void vload(int s, double2 (&v)[EPT]) const {
    const double2* p = vec(s);
#pragma unroll
    for (int j = 0; j < EPT; j++) v[j] = p[64 * j];
  }
  __device__ __forceinline__ double2 pmul(int j, const double2 w) const {
    return make_double2(fma(pr[SPLIT ? j : 0], w.x, -pi[SPLIT ? j : 0] * w.y), fma(pr[SPLIT ? j : 0], w.y, pi[SPLIT ? j : 0] * w.x));
  }
  // one pass of Horner's rule: y <- P (rhs + alpha C y); the published vector is y on entry and on exit
  template <bool TRANS>
  __device__ __forceinline__ void kry_pass(const StepC<Q>& c, double alpha, const double2 (&rhs)[EPT], double2 (&y)[EPT]) {
    const unsigned wa = st.tb + (unsigned)st.dlt;
    double2 prev = y[0];
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const double2 own = y[j];
      const double2 t = st.template apply<TRANS, true>(c, j, own, prev, y[j + 1 < EPT ? j + 1 : j]);
      const double2 w = pmul(j, make_double2(fma(alpha, t.x, rhs[j].x), fma(alpha, t.y, rhs[j].y)));
      prev = own;
      y[j] = w;
      st.st(wa + (unsigned)j * COLB, w);
      slot_fence<EPT>();
    }
    __syncthreads();
    st.flip();
  }
  // h_00 / (h_00^2 + h_10^2) of the one-vector solve, or a negative value where its residual is above the tolerance
  __device__ __forceinline__ double kry_one_vector(const SweepArgs& A, const double (&d)[3]) const {
    const double bb = d[0], ab = d[1], ss = d[2];
    const double ttol2 = fmax(A.reltol * A.reltol * bb, A.abstol * A.abstol);
    if (bb <= ttol2) return 0.0;  // ||b|| <= tolerance: KSP returns the zero initial guess
    const double ibb = 1.0 / bb, a = ab * ibb, h00 = 1.0 - a, h10sq = fmax(fma(-a, a, ss * ibb), 0.0), den = fma(h00, h00, h10sq);
    const bool conv = bb * h10sq <= A.kry_tau2 * ttol2 * den || A.maxiter <= 1;  // (kry_tau: SweepArgs)
    return conv ? h00 / den : -1.0;
  }

  // the generic path: on entry the right-hand side is in v AND in slot SB; on exit y = solution, v = right-hand side
  template <bool TRANS>
  __device__ __forceinline__ int kry_generic(const SweepArgs& A, const StepC<Q>& c, double alpha, double2 (&v)[EPT], double2 (&y)[EPT]) {
    const int poly = A.gmres_poly > 1 ? A.gmres_poly : 1;
    const int mre = A.kry_restart >= 1 && A.kry_restart < KRY_MR ? A.kry_restart : KRY_MR;  // restart length (option krylov_restart)
    double* hc = ksc;                  // [MR + 2] current Hessenberg column
    double* cs = hc + (KRY_MR + 2);    // [MR]
    double* sn = cs + KRY_MR;          // [MR]
    double* g = sn + KRY_MR;           // [MR + 2]
    double* R = g + (KRY_MR + 2);      // [MR][MR] row-major upper triangle (reciprocal diagonal)
    double* yk = R + KRY_MR * KRY_MR;  // [MR]
    int napp = 0, its = 0;
    bool have_total = false;
    double ttol = 0.0;
    for (int cycle = 0;; cycle++) {
      double t1[1] = {0.0};
#pragma unroll
      for (int j = 0; j < EPT; j++) t1[0] = fma(v[j].x, v[j].x, fma(v[j].y, v[j].y, t1[0]));
      sum_rows<1>(t1);
      const double ibeta = t1[0] > 0.0 ? rsqrt_nr(t1[0]) : 0.0, beta = t1[0] * ibeta;
      // (the acceptance factor of the one-vector path - SweepArgs::kry_tau2 - here as well: a preconditioned vector takes the residual down
      //  by the contraction of p passes at once, but the LAST one of a solve lands anywhere below the tolerance; held to kry_tau x the
      //  tolerance the generic path is as accurate as the reference's GMRES typically is, profiles/r6_kry_seed_sweep.txt)
      if (cycle == 0) ttol = sqrt(A.kry_tau2) * fmax(A.reltol * beta, A.abstol);
#pragma unroll
      for (int j = 0; j < EPT; j++) y[j] = make_double2(0.0, 0.0);
      if (beta <= ttol || its >= A.maxiter) break;
#pragma unroll
      for (int j = 0; j < EPT; j++) v[j] = make_double2(v[j].x * ibeta, v[j].y * ibeta);
      vstore(SV, v);
      double gcur = beta;
      int jj = 0;
      bool conv = false;
      while (jj < mre) {
        // z = R_p v_jj, parked in Z_jj; w = (I - alpha M) z takes its registers
#pragma unroll
        for (int j = 0; j < EPT; j++) y[j] = pmul(j, v[j]);
        publish(y);
        for (int m = 1; m < poly; m++) kry_pass<TRANS>(c, alpha, v, y);
        {
          double2* zp = vec(SZ + jj);
          double2 prev = y[0];
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            const double2 own = y[j];
            const double2 t = st.template apply<TRANS, false>(c, j, own, prev, y[j + 1 < EPT ? j + 1 : j]);
            zp[64 * j] = own;
            prev = own;
            y[j] = make_double2(fma(-alpha, t.x, own.x), fma(-alpha, t.y, own.y));
            slot_fence<EPT>();
          }
        }
        napp += poly;
        // classical Gram-Schmidt: every projection against the un-updated w, four per reduction; v_jj is in registers, v_k (k < jj) is read back
        for (int p0 = 0; p0 <= jj; p0 += 4) {
          double h4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int k = p0 + q;
            if (k == jj) {
#pragma unroll
              for (int j = 0; j < EPT; j++) h4[q] = fma(y[j].x, v[j].x, fma(y[j].y, v[j].y, h4[q]));
            } else if (k < jj) {
              const double2* vp = vec(SV + k);
#pragma unroll
              for (int j = 0; j < EPT; j++) {
                const double2 vk = vp[64 * j];
                h4[q] = fma(y[j].x, vk.x, fma(y[j].y, vk.y, h4[q]));
              }
            }
          }
          sum_rows<4>(h4);
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (p0 + q <= jj) hc[p0 + q] = h4[q];
        }
        {
          const double h = hc[jj];
#pragma unroll
          for (int j = 0; j < EPT; j++) y[j] = make_double2(fma(-h, v[j].x, y[j].x), fma(-h, v[j].y, y[j].y));
        }
        for (int k = 0; k < jj; k++) {
          const double h = hc[k];
          const double2* vp = vec(SV + k);
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            const double2 vk = vp[64 * j];
            y[j] = make_double2(fma(-h, vk.x, y[j].x), fma(-h, vk.y, y[j].y));
          }
        }
        double nn[1] = {0.0};
#pragma unroll
        for (int j = 0; j < EPT; j++) nn[0] = fma(y[j].x, y[j].x, fma(y[j].y, y[j].y, nn[0]));
        sum_rows<1>(nn);
        const double ihn = nn[0] > 0.0 ? rsqrt_nr(nn[0]) : 0.0, hn = nn[0] * ihn;
        // Givens rotations: redundantly by every thread on workgroup-uniform values, idempotent LDS writes only (Team::gmres_g of qd_device.h)
        double cur_h = hc[0];
        for (int k = 0; k < jj; k++) {
          const double a1 = hc[k + 1], ck = cs[k], sk = sn[k];
          R[k * KRY_MR + jj] = ck * cur_h + sk * a1;
          cur_h = -sk * cur_h + ck * a1;
        }
        const double s2 = cur_h * cur_h + hn * hn;
        const double irr = s2 > 0.0 ? rsqrt_nr(s2) : 0.0;
        const double cj = s2 > 0.0 ? cur_h * irr : 1.0, sj = hn * irr;
        cs[jj] = cj;
        sn[jj] = sj;
        R[jj * KRY_MR + jj] = irr;
        g[jj] = cj * gcur;
        gcur = -sj * gcur;
        its++;
        jj++;
        if (fabs(gcur) <= ttol || hn == 0.0) { conv = true; break; }
        if (its >= A.maxiter || jj >= mre) break;
#pragma unroll
        for (int j = 0; j < EPT; j++) v[j] = make_double2(y[j].x * ihn, y[j].y * ihn);
        vstore(SV + jj, v);
        __syncthreads();  // (the scalars of this column have been read by every thread before the next one overwrites hc)
      }
      for (int rw = jj - 1; rw >= 0; rw--) {
        double sacc = g[rw];
        for (int cc = rw + 1; cc < jj; cc++) sacc -= R[rw * KRY_MR + cc] * yk[cc];
        yk[rw] = sacc * R[rw * KRY_MR + rw];
      }
#pragma unroll
      for (int j = 0; j < EPT; j++) y[j] = make_double2(0.0, 0.0);
      for (int cc = 0; cc < jj; cc++) {
        const double f = yk[cc];
        const double2* zp = vec(SZ + cc);
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const double2 zk = zp[64 * j];
          y[j] = make_double2(fma(f, zk.x, y[j].x), fma(f, zk.y, y[j].y));
        }
      }
      if (conv || its >= A.maxiter) break;
      // restart: park the accumulated solution, r = b - (I - alpha M) y_total
      if (have_total) {
        const double2* tp = vec(SY);
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const double2 o = tp[64 * j];
          y[j].x += o.x;
          y[j].y += o.y;
        }
      }
      vstore(SY, y);
      have_total = true;
      publish(y);
      {
        const double2* bp = vec(SB);
        double2 prev = y[0];
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const double2 own = y[j];
          const double2 t = st.template apply<TRANS, false>(c, j, own, prev, y[j + 1 < EPT ? j + 1 : j]);
          const double2 bj = bp[64 * j];
          v[j] = make_double2(bj.x - fma(-alpha, t.x, own.x), bj.y - fma(-alpha, t.y, own.y));
          prev = own;
          slot_fence<EPT>();
        }
      }
      napp++;
      __syncthreads();  // every thread has read the scalars of this cycle before the next one overwrites them
    }
    if (have_total) {
      const double2* tp = vec(SY);
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const double2 o = tp[64 * j];
        y[j].x += o.x;
        y[j].y += o.y;
      }
    }
    vload(SB, v);
    return napp;
  }

  // (I - alpha M^{(T)}) y = b for the adjoint sweep: b stays in registers.  Returns the RHS applications.
  template <bool TRANS>
  __device__ __forceinline__ int kry_solve(const SweepArgs& A, const StepC<Q>& c, double alpha, double2 (&b)[EPT], double2 (&y)[EPT]) {
    const int poly = A.gmres_poly > 1 ? A.gmres_poly : 1;
#pragma unroll
    for (int j = 0; j < EPT; j++) y[j] = pmul(j, b[j]);
    publish(y);
    for (int m = 1; m < poly; m++) kry_pass<TRANS>(c, alpha, b, y);
    double d[3] = {0.0, 0.0, 0.0};
    {
      double2 prev = y[0];
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const double2 own = y[j];
        const double2 t = st.template apply<TRANS, false>(c, j, own, prev, y[j + 1 < EPT ? j + 1 : j]);
        const double rx = b[j].x - fma(-alpha, t.x, own.x), ry = b[j].y - fma(-alpha, t.y, own.y);  // r = b - (I - alpha M) z
        d[0] = fma(b[j].x, b[j].x, fma(b[j].y, b[j].y, d[0]));
        d[1] = fma(rx, b[j].x, fma(ry, b[j].y, d[1]));
        d[2] = fma(rx, rx, fma(ry, ry, d[2]));
        prev = own;
        slot_fence<EPT>();
      }
    }
    sum_rows<3>(d);
    const double fac = kry_one_vector(A, d);
    if (__builtin_expect(fac >= 0.0, 1)) {
#pragma unroll
      for (int j = 0; j < EPT; j++) y[j] = make_double2(fac * y[j].x, fac * y[j].y);
      return poly;
    }
#ifdef QD_KRY_NOCOLD
    return poly;
#else
    vstore(SB, b);
    return poly + kry_generic<TRANS>(A, c, alpha, b, y);
#endif
  }

  // Forward sub-step in stage form (see stage()): the passes run on z = x + alpha y with x as the right-hand side, so the application that
  // forms b = M x IS the first pass (b = C x + D x is parked in slot SB from there: 16 B per element and step through L2) and the iterate
  // of the k-system is y = (z - x) / alpha.  Residual of the k-system at y: r = b - (I - alpha M) y = M z - y.  On exit z = x + alpha k.
  __device__ __forceinline__ int kry_stage(const SweepArgs& A, const StepC<Q>& c, double alpha, double2 (&x)[EPT], double2 (&z)[EPT]) {
    const int poly = A.gmres_poly > 1 ? A.gmres_poly : 1;
    double2* bp = vec(SB);
    {  // first pass: z_0 = P (x + alpha C x), b = C x + D x
      const unsigned wa = st.tb + (unsigned)st.dlt;
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const double2 own = x[j];
        const double2 t = st.template apply<false, true>(c, j, own, x[j > 0 ? j - 1 : 0], x[j + 1 < EPT ? j + 1 : j]);
        double dwj, ddj;
        st.diag(j, dwj, ddj);
        const double2 bj = make_double2(fma(dwj, own.y, fma(ddj, own.x, t.x)), fma(-dwj, own.x, fma(ddj, own.y, t.y)));
        if constexpr (BLDS) *reinterpret_cast<float2*>(st.smem + b32 + 512u * j) = make_float2((float)bj.x, (float)bj.y);
        else bp[64 * j] = bj;
        const double2 w = pmul(j, make_double2(fma(alpha, t.x, own.x), fma(alpha, t.y, own.y)));
        z[j] = w;
        st.st(wa + (unsigned)j * COLB, w);
        slot_fence<EPT>();
      }
      __syncthreads();
      st.flip();
    }
    for (int m = 1; m < poly; m++) kry_pass<false>(c, alpha, x, z);
    double d[3] = {0.0, 0.0, 0.0};
    {
      const double ia = 1.0 / alpha;
      double2 prev = z[0];
      double2 bq[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
      if constexpr (!BLDS) {
        bq[0] = bp[0];
        bq[1] = bp[EPT > 1 ? 64 : 0];
      }
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const double2 own = z[j];
        double2 bj;
        if constexpr (BLDS) {
          const float2 f = *reinterpret_cast<const float2*>(st.smem + b32 + 512u * j);
          bj = make_double2((double)f.x, (double)f.y);
        } else {
          bj = bq[j & 1];
          if (j + 2 < EPT) bq[j & 1] = bp[64 * (j + 2 < EPT ? j + 2 : 0)];
        }
        const double2 t = st.template apply<false, false>(c, j, own, prev, z[j + 1 < EPT ? j + 1 : j]);
        const double rx = fma(-ia, own.x - x[j].x, t.x), ry = fma(-ia, own.y - x[j].y, t.y);  // r = M z - (z - x) / alpha
        d[0] = fma(bj.x, bj.x, fma(bj.y, bj.y, d[0]));
        d[1] = fma(rx, bj.x, fma(ry, bj.y, d[1]));
        d[2] = fma(rx, rx, fma(ry, ry, d[2]));
        prev = own;
        slot_fence<EPT>();
      }
    }
    sum_rows<3>(d);
    const double fac = kry_one_vector(A, d);
    if (__builtin_expect(fac >= 0.0, 1)) {
#pragma unroll
      for (int j = 0; j < EPT; j++) z[j] = make_double2(fma(fac, z[j].x - x[j].x, x[j].x), fma(fac, z[j].y - x[j].y, x[j].y));
      return poly + 1;
    }
#ifdef QD_KRY_NOCOLD
    return poly + 1;
#endif
    // the solve starts over on the k-system: b = M x again (in fp64), x parked, b in its registers
    publish(x);
    apply_all<false>(c, x, z);
    vstore(SX, x);
    vstore(SB, z);
#pragma unroll
    for (int j = 0; j < EPT; j++) x[j] = z[j];
    const int n = 1 + kry_generic<false>(A, c, alpha, x, z);
    vload(SX, x);
#pragma unroll
    for (int j = 0; j < EPT; j++) z[j] = make_double2(fma(alpha, z[j].x, x[j].x), fma(alpha, z[j].y, x[j].y));
    return poly + 1 + n;
  }
};

// ---------------------------------------------------------------------------------------------
// Time-sliced scheduling of the sweeps.  One workgroup owns one CU (two exchange buffers of N KiB), so a batch of nb initial conditions
// runs in nb / #CUs rounds of whole sweeps and the last round is as long as any other however few workgroups it holds: 3600 initial
// conditions on 256 CUs are 14.06 rounds - 6 % of the sweep with 240 CUs idle; the 450 of an eight-GPU shard 1.76 rounds - 12 %.  With
// A.sched set the sweep is cut into A.nslice slices of whole time steps and a resident grid draws (slice, initial condition) tasks from
// a counter, slice-major: the tail shrinks to one SLICE.  Slice k of an initial condition waits for slice k - 1 (a flag per initial
// condition, released at agent scope after the state has been written back; the predecessor was drawn earlier, hence is running or
// done: no deadlock whatever the dispatch order) and picks the state up from the carry buffer.  A wait that exceeds A.sched_ticks (4 s x
// the processes sharing the device x the slice length in thousands of steps, qd_handle::arm_slices) raises the error word instead of
// hanging the device.
//   sched[0] task counter | sched[1] error word | sched[2 + ic] slices of ic completed
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int sched_next(unsigned* sched, unsigned* slot) {
  __syncthreads();  // (the previous task's last reads of *slot)
  if (threadIdx.x == 0) *slot = atomicAdd(sched, 1u);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane((int)*slot);
}
// wait until `want` slices of initial condition ic are complete; false after the time limit
// The word of an initial condition: slices completed in its low byte (at most 64 slices), above it a value the finished slice hands to
// its successor (*carry, through the LDS word `slot`): the solver's pass count of the last sub-step, so that a sliced sweep skips the
// same stopping tests as an unsliced one and the two stay bit-identical.
__device__ __forceinline__ bool sched_wait(unsigned* sched, int ic, unsigned want, unsigned long long limit, unsigned* slot, int* carry) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    unsigned v;
    while (((v = __hip_atomic_load(sched + 2 + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffu) < want) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > limit) {
        atomicExch(sched + 1, 1u);
        break;
      }
    }
    *slot = v >> 8;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  *carry = __builtin_amdgcn_readfirstlane((int)*slot);
  return __hip_atomic_load(sched + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}
// the state of ic has been written: publish the completion of its slice
__device__ __forceinline__ void sched_done(unsigned* sched, int ic, unsigned done, int carry) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have reached L2
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(sched + 2 + ic, done | ((unsigned)carry << 8), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// first sub-step of slice sl (whole time steps)
__device__ __forceinline__ int slice_start(const SweepArgs& A, int sl) {
  return (int)((long long)A.ntime * sl / A.nslice) * A.nstages;
}

// ---------------------------------------------------------------------------------------------
// forward sweep (TimeStepper::solveODE for every initial condition of the batch)
// ---------------------------------------------------------------------------------------------
template <int Q, int EPT, bool SPLIT, bool USLOT = false, bool SKIP = false, bool KRY = false>
__global__ void __launch_bounds__(col_max_threads(EPT)) k_forward_col(const SweepArgs A) {
  static_assert(!KRY || (SPLIT && !SKIP), "the Krylov solver runs on the diagonal-split form");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef ColTeam<Q, EPT, SPLIT, USLOT, SKIP> TM;
  const DevSys& S = A.S;
  TM tm;
  tm.init(S, smem);
  if constexpr (KRY) tm.init_kry(A);
  __shared__ unsigned task_slot, carry_slot;
  const int dim = S.dim, ntask = A.nb * A.nslice;
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  const bool leak = pen_on && A.leak_on && tm.st.row_is_guard(S);
  for (int task = A.sched ? sched_next(A.sched, &task_slot) : (int)blockIdx.x; task < ntask; task = A.sched ? sched_next(A.sched, &task_slot) : ntask) {
  const int ic = task % A.nb, sl = task / A.nb;
  const int s_lo = slice_start(A, sl), s_hi = slice_start(A, sl + 1);
  tm.lastn = 0;  // (the pass count of the previous sub-step: none at t = 0, otherwise what the previous slice hands over)
  if (sl > 0 && !sched_wait(A.sched, ic, (unsigned)sl, A.sched_ticks, &carry_slot, &tm.lastn)) return;
  double2 x[EPT];
  {
    // slice 0 starts from the initial condition, every other one from where its predecessor left the state (the carry = xT)
    const double* x0 = (sl > 0 ? A.xT : A.x0) + (size_t)ic * 2 * dim;
#pragma unroll
    for (int j = 0; j < EPT; j++) x[j] = tm.st.ok(j) ? make_double2(x0[tm.st.elem(j)], x0[dim + tm.st.elem(j)]) : make_double2(0.0, 0.0);
  }
  double pen_local = 0.0, pen_uniform = 0.0;
  unsigned long long napply = 0;
  // global accesses of the thread's elements: one divergent region per call (rows), uniform branches inside (columns)
  auto store_state = [&](double* dst, const double2(&v)[EPT], bool nt) {
    if (tm.st.rowok) {
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (tm.st.colok(j)) {
          const int e = tm.st.elem_now(j);
          if (nt) {
            __builtin_nontemporal_store(v[j].x, dst + e);
            __builtin_nontemporal_store(v[j].y, dst + dim + e);
          } else {
            dst[e] = v[j].x;
            dst[dim + e] = v[j].y;
          }
        }
    }
  };

  for (int s = s_lo; s < s_hi; s++) {
    StepC<Q> c;
    load_step_k<Q>(A.ctl + (size_t)s * A.cs, c, false);
    if (SPLIT) tm.template set_alpha<false>(0.5 * c.h);
    if (A.traj) store_state(A.traj + ((size_t)s * A.nb + ic) * 2 * dim, x, true);
    // the sub-step in stage form (ColTeam::stage): x is the right-hand side of the solve and stays in registers
    tm.publish(x);
    double2 z[EPT];
    if constexpr (KRY) napply += tm.kry_stage(A, c, 0.5 * c.h, x, z);
    else napply += tm.stage(A, c, 0.5 * c.h, x, z);
    if (A.ztraj && tm.st.rowok) {  // the primal stage, read back by the adjoint sweep instead of repeating this solve: private to
      // this kernel pair, kept interleaved (one 16-byte streaming access per element; qd_handle tags the layout: ztraj_fmt)
      col_d2* dst = reinterpret_cast<col_d2*>(A.ztraj) + ((size_t)s * A.nb + ic) * dim;
#pragma unroll
      for (int j = 0; j < EPT; j++)
        if (tm.st.colok(j)) {
          const col_d2 t = {z[j].x, z[j].y};
          __builtin_nontemporal_store(t, dst + tm.st.elem_now(j));
        }
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {  // x_{n+1} = x + h k = 2 z - x
      x[j].x = fma(2.0, z[j].x, -x[j].x);
      x[j].y = fma(2.0, z[j].y, -x[j].y);
    }
    // in-loop penalties at the end of a FULL time step (timestepper.cpp:141-154, :256-298)
    if (pen_on && (s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages - 1;
      const double tstop = (n + 1) * A.dt;
      if (wj_on) {
        // (tabulated per time step: an exp() evaluated here, next to x and k, was spilt by the compiler and reloaded through seven
        // serialised scratch round trips - 3 us per workgroup and step, 12 % of the 3 x 20 forward sweep)
        double weight;
        if (A.wjw) {
          weight = kload(A.wjw + n);
        } else {
          const double a = (tstop - A.Tfinal) / A.penalty_param;
          weight = 1.0 / A.penalty_param * exp(-(a * a));
        }
        // finalizeJ is affine for Lindblad: J = jr (Jfrobenius, Jmeasure) or 1 - jr (Jtrace).  Jmeasure only sees the diagonal of rho,
        // which the column layout has at hand (the generic routine divides the vectorised index by N per element and step: ~12 % of
        // the 3 x 20 forward sweep)
        if (A.tg.objective_type == QD_OBJ_JMEASURE) {
          const double wrow = weight * A.dt * fabs((double)(tm.st.row - A.tg.purestate_id));
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.st.rowok && tm.st.colof(j) == tm.st.row) pen_local = fma(wrow, x[j].x, pen_local);
        } else if (tm.st.rowok) {
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.st.colok(j)) {
              double jr = 0.0, ji = 0.0;
              evalJ_part<true>(S, A.tg, ic, tm.st.elem_now(j), x[j], jr, ji);
              pen_local += (A.tg.objective_type == QD_OBJ_JTRACE ? -1.0 : 1.0) * weight * A.dt * jr;
            }
        }
        if (A.tg.objective_type == QD_OBJ_JTRACE) pen_uniform += weight * A.dt;
      }
      if (leak) {
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (tm.st.colof(j) == tm.st.row) pen_local += (x[j].x * x[j].x + x[j].y * x[j].y) / A.ntime;
      }
    }
  }
  store_state(A.xT + (size_t)ic * 2 * dim, x, false);
  if (A.traj && sl == A.nslice - 1) store_state(A.traj + ((size_t)A.nsub * A.nb + ic) * 2 * dim, x, false);
  double v[1] = {pen_local};
  tm.template sum<1>(v);
  if (threadIdx.x == 0) {
    A.pen_out[ic] = (sl > 0 ? A.pen_out[ic] : 0.0) + v[0] + pen_uniform;  // (slices of one initial condition run one after the other)
    A.dpdm_out[ic] = 0.0;  // the dpdm penalty is Schroedinger only (timestepper.cpp:143-146)
    atomicAdd(A.napply, napply);
  }
  if (A.sched) sched_done(A.sched, ic, (unsigned)(sl + 1), tm.lastn);
  }
}

// ---------------------------------------------------------------------------------------------
// adjoint sweep (TimeStepper::solveAdjointODE + ImplMidpoint::evolveBWD + compute_dRHS_dParams)
// ---------------------------------------------------------------------------------------------
template <int Q, int EPT, bool SPLIT, bool USLOT = false, bool SKIP = false, bool KRY = false>
__global__ void __launch_bounds__(col_max_threads(EPT)) k_adjoint_col(const SweepArgs A) {
  static_assert(!KRY || (SPLIT && !SKIP), "the Krylov solver runs on the diagonal-split form");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef ColTeam<Q, EPT, SPLIT, USLOT, SKIP> TM;
  const DevSys& S = A.S;
  TM tm;
  tm.init(S, smem);
  if constexpr (KRY) tm.init_kry(A);
  __shared__ unsigned task_slot, carry_slot;
  const int dim = S.dim, ntask = A.nb * A.nslice;
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  const bool leak = pen_on && A.leak_on && tm.st.row_is_guard(S);
  for (int task = A.sched ? sched_next(A.sched, &task_slot) : (int)blockIdx.x; task < ntask; task = A.sched ? sched_next(A.sched, &task_slot) : ntask) {
  // backwards in time: task slice sl covers the time slice nslice - 1 - sl
  const int ic = task % A.nb, sl = task / A.nb;
  const int s_lo = slice_start(A, A.nslice - 1 - sl), s_hi = slice_start(A, A.nslice - sl);
  tm.lastna = 0;
  if (sl > 0 && !sched_wait(A.sched, ic, (unsigned)sl, A.sched_ticks, &carry_slot, &tm.lastna)) return;
  double2 xb[EPT];
  {
    const double* xbT = (sl > 0 ? A.stash : A.xbarT) + (size_t)ic * 2 * dim;  // (the carry of the adjoint state: SweepArgs::stash)
#pragma unroll
    for (int j = 0; j < EPT; j++) xb[j] = tm.st.ok(j) ? make_double2(xbT[tm.st.elem(j)], xbT[dim + tm.st.elem(j)]) : make_double2(0.0, 0.0);
  }
  const double jbar_pen = A.jbar[ic * 3 + 0];
  auto load_state = [&](const double* base, int s, double2(&dst)[EPT]) {
    const double* src = base + ((size_t)s * A.nb + ic) * 2 * dim;
#pragma unroll
    for (int j = 0; j < EPT; j++)
      dst[j] = tm.st.ok(j) ? make_double2(__builtin_nontemporal_load(src + tm.st.elem_now(j)), __builtin_nontemporal_load(src + dim + tm.st.elem_now(j)))
                           : make_double2(0.0, 0.0);
  };

  for (int s = s_hi - 1; s >= s_lo; s--) {
    // penalty adjoints at the end of a full step, with the primal x_n (timestepper.cpp:220-227, :300-339)
    if (pen_on && (s + 1) % A.nstages == 0 && (wj_on || leak)) {
      const int n = (s + 1) / A.nstages;
      const double tstop = n * A.dt;
      // (the weighted Jmeasure's adjoint is a constant per row: without guard levels this sweep never reads the states, and the
      // forward sweep of a gradient evaluation has not stored them - qd_handle::adjoint_reads_states)
      const bool need_xn = leak || (wj_on && A.tg.objective_type != QD_OBJ_JMEASURE);
      double2 xn[EPT];
      if (need_xn) {
        load_state(A.traj, s + 1, xn);
      } else {
#pragma unroll
        for (int j = 0; j < EPT; j++) xn[j] = make_double2(0.0, 0.0);
      }
      if (wj_on) {
        double weight;
        if (A.wjw) {
          weight = kload(A.wjw + (n - 1));
        } else {
          const double a = (tstop - A.Tfinal) / A.penalty_param;
          weight = 1.0 / A.penalty_param * exp(-(a * a));
        }
        double rb, ib;
        finalizeJ_diff<true>(A.tg, 0.0, 0.0, rb, ib);
        if (A.tg.objective_type == QD_OBJ_JMEASURE) {
          const double wrow = weight * rb * jbar_pen * A.dt * fabs((double)(tm.st.row - A.tg.purestate_id));
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.st.rowok && tm.st.colof(j) == tm.st.row) xb[j].x += wrow;
        } else {
#pragma unroll
          for (int j = 0; j < EPT; j++)
            if (tm.st.ok(j)) evalJ_diff_elem<true>(S, A.tg, ic, tm.st.elem_now(j), xn[j], xb[j], weight * rb * jbar_pen * A.dt, weight * ib * jbar_pen * A.dt);
        }
      }
      if (leak) {
#pragma unroll
        for (int j = 0; j < EPT; j++)
          if (tm.st.colof(j) == tm.st.row) {
            xb[j].x += 2.0 * xn[j].x * jbar_pen / A.ntime;
            xb[j].y += 2.0 * xn[j].y * jbar_pen / A.ntime;
          }
      }
    }
    StepC<Q> c;
    load_step_k<Q>(A.ctl + (size_t)s * A.cs, c, false);
    // ImplMidpoint::evolveBWD (timestepper.cpp:631-694); the primal stage z of the sub-step was stored by the forward sweep
    if (SPLIT) tm.template set_alpha<true>(0.5 * c.h);
    double2 kb[EPT];  // adjoint stage: (I - h/2 M)^T kbar = xbar ; kbar *= h
    if constexpr (KRY) tm.template kry_solve<true>(A, c, 0.5 * c.h, xb, kb);
    else tm.template neumann<true>(A, c, 0.5 * c.h, xb, kb);
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      kb[j].x *= c.h;
      kb[j].y *= c.h;
    }
    double cf[2 * Q];
#pragma unroll
    for (int i = 0; i < 2 * Q; i++) cf[i] = 0.0;
    {
      double2 z[EPT];
      {
        const col_d2* src = reinterpret_cast<const col_d2*>(A.ztraj) + ((size_t)s * A.nb + ic) * dim;
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          if (tm.st.ok(j)) {
            const col_d2 t = __builtin_nontemporal_load(src + tm.st.elem_now(j));
            z[j] = make_double2(t.x, t.y);
          } else {
            z[j] = make_double2(0.0, 0.0);
          }
        }
      }
      tm.publish(z);
      // gradient coefficients x^T dM/dp_k z and x^T dM/dq_k z with x := kbar (mastereq.hpp:553-604)
#pragma unroll
      for (int j = 0; j < EPT; j++) {
#pragma unroll
        for (int k = 0; k < Q; k++) {
          double2 Av, Bv;
          tm.st.ladder(k, j, z[j], z[j > 0 ? j - 1 : 0], z[j + 1 < EPT ? j + 1 : j], Av, Bv);
          cf[2 * k] += Bv.y * kb[j].x - Bv.x * kb[j].y;
          cf[2 * k + 1] += Av.x * kb[j].x + Av.y * kb[j].y;
        }
        slot_fence<EPT>();
      }
    }
    tm.template sum_post<2 * Q>(cf);
    // xbar += M^T kbar
    tm.publish(kb);  // (its barrier also completes the coefficient sums)
    if (threadIdx.x < 2 * Q) A.coeff[((size_t)ic * A.nsub + s) * 2 * Q + threadIdx.x] = tm.sum_collect((int)threadIdx.x);
    double2 t[EPT];
    tm.template apply_all<true>(c, kb, t);
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      xb[j].x += t[j].x;
      xb[j].y += t[j].y;
    }
  }
  double* d0 = sl == A.nslice - 1 ? A.xbar0 : A.stash;
  if (d0) {
    d0 += (size_t)ic * 2 * dim;
#pragma unroll
    for (int j = 0; j < EPT; j++)
      if (tm.st.ok(j)) {
        d0[tm.st.elem(j)] = xb[j].x;
        d0[dim + tm.st.elem(j)] = xb[j].y;
      }
  }
  if (A.sched) sched_done(A.sched, ic, (unsigned)(sl + 1), tm.lastna);
  }
}

// single operator application (test hook = MatMult / MatMultTranspose on the shell)
template <int Q, int EPT, bool SPLIT>
__global__ void __launch_bounds__(col_max_threads(EPT)) k_apply_col(const DevSys S, const double* __restrict__ ctlrow, int transpose, const double* __restrict__ xin,
                                                          double* __restrict__ yout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef ColTeam<Q, EPT, SPLIT> TM;
  TM tm;
  tm.init(S, smem);
  const int ic = blockIdx.x, dim = S.dim;
  double2 x[EPT], y[EPT];
  const double* x0 = xin + (size_t)ic * 2 * dim;
#pragma unroll
  for (int j = 0; j < EPT; j++) x[j] = tm.st.ok(j) ? make_double2(x0[tm.st.elem(j)], x0[dim + tm.st.elem(j)]) : make_double2(0.0, 0.0);
  StepC<Q> c;
  load_step_k<Q>(ctlrow, c, false);
  scalarize<Q>(c, false);
  tm.publish(x);
  if (transpose) tm.template apply_all<true>(c, x, y);
  else tm.template apply_all<false>(c, x, y);
  double* yo = yout + (size_t)ic * 2 * dim;
#pragma unroll
  for (int j = 0; j < EPT; j++)
    if (tm.st.ok(j)) {
      yo[tm.st.elem(j)] = y[j].x;
      yo[dim + tm.st.elem(j)] = y[j].y;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int col_cu_count() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  return ncu;
}

// Slices of a sweep of nb initial conditions over ntime steps (1 = one workgroup per initial condition, no scheduler): the smallest power
// of two that brings the idle tail - (ceil(r) - r) / ceil(r) for r = nb k / #CUs rounds - below 1 %, keeping at least 32 steps per slice.
int col_slices(int nb, int ntime, const TuneOpts& o) {
  if (o.col_slices == 1) return 1;
  // (at most 255 slices: the scheduler word of an initial condition counts completed slices in its low byte, sched_wait / sched_done)
  if (o.col_slices > 1) return std::min(std::min(o.col_slices, 255), std::max(ntime, 1));
  const int ncu = col_cu_count();
  if (nb <= ncu) return 1;
  int best = 1;
  double best_waste = 1.0;
  for (int k = 1; k <= 64 && ntime / k >= 32; k *= 2) {
    const double r = (double)nb * k / ncu;
    const double waste = (ceil(r) - r) / ceil(r);
    if (waste < best_waste - 1e-12) {
      best_waste = waste;
      best = k;
    }
    if (waste < 0.01) break;
  }
  return best_waste < 0.01 || best > 1 ? best : 1;
}

// SweepArgs::kry of the Krylov kernels: GMRES_MR_G + 2 padded scratch vectors per RESIDENT workgroup (ColTeam::init_kry), in doubles
// (a sweep without time slices starts one workgroup per initial condition, a sliced one a resident grid: col_grid)
size_t col_krylov_doubles(int nb, int nslice) { return (size_t)(nslice > 1 ? std::min(nb * nslice, 2 * col_cu_count()) : nb) * (GMRES_MR_G + 2) * 2 * KRY_VEC; }

template <typename K>
static hipError_t set_lds_col(K kern, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Lindblad, matrix-free, no dipole-dipole coupling, runtime level counts that are not all 2, two or three oscillators, a density
// matrix of 33..64 rows (one lane per row), the last oscillator with stride 1 (always: post[Q-1] == 1)
bool collean_available(const DevSys& S, const TuneOpts& o) {
  if (o.no_collean) return false;
  if (!S.lindblad || S.dense || S.hasJ || (S.Q != 2 && S.Q != 3) || S.N < 33 || S.N > 64) return false;
  bool qubit = true;
  for (int k = 0; k < S.Q; k++) qubit = qubit && S.n[k] == 2;
  return !qubit && S.post[S.Q - 1] == 1;
}

// the columns of a wave share the level indices of every oscillator but the stride-1 one (ColLean's USLOT)
template <int EPT>
static bool col_uslot(const DevSys& S) {
  if (S.N % EPT != 0) return false;
  for (int k = 0; k < S.Q - 1; k++)
    if (S.post[k] % EPT != 0) return false;
  return true;
}

// grid of a sweep: one workgroup per initial condition, or - time-sliced scheduling - as many workgroups as are resident at once
template <typename K>
static int col_grid(K kern, const SweepArgs& a, int threads, size_t lds) {
  if (!a.sched) return a.nb;
  int per_cu = 1, dev = 0;
  hipDeviceProp_t prop;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  int ncu = col_cu_count();
  (void)dev;
  (void)prop;
  if (a.use_gmres) per_cu = std::min(per_cu, 2);  // (col_krylov_doubles)
  return std::min(a.nb * a.nslice, per_cu * ncu);
}

template <int Q, int EPT, bool SPLIT>
static hipError_t go_fwd_col_s(const SweepArgs& a, hipStream_t st) {
  typedef ColLean<Q, EPT> ST;
  const size_t lds = ST::lds_bytes(a.S.N);
  // (SKIP: stopping tests skipped, see ColTeam::stage)
  // (implicit midpoint only: the predictor compares the pass count of a sub-step with its predecessor's, and the sub-steps of a composite
  //  step differ in size - IMR4 / IMR8 test every pass; nothing of this lives in the kernels)
  const bool skip = a.rel2 < 1e-30f && !a.col_noskip && a.nstages == 1;
  auto kf = col_uslot<EPT>(a.S) ? (skip ? k_forward_col<Q, EPT, SPLIT, true, true> : k_forward_col<Q, EPT, SPLIT, true, false>)
                                : (skip ? k_forward_col<Q, EPT, SPLIT, false, true> : k_forward_col<Q, EPT, SPLIT, false, false>);
  hipError_t e = set_lds_col(kf, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3(col_grid(kf, a, 64 * (ST::ncols(a.S.N) / EPT), lds)), dim3(64 * (ST::ncols(a.S.N) / EPT)), lds, st, a);
  return hipGetLastError();
}
// the Krylov kernels (SweepArgs::use_gmres): five or eight columns per wave only
template <int Q, int EPT>
static hipError_t go_fwd_col_k(const SweepArgs& a, hipStream_t st) {
  if constexpr (EPT == 5 || EPT == 8) {
    typedef ColLean<Q, EPT> ST;
    const size_t lds = ST::lds_bytes(a.S.N) + ColTeam<Q, EPT, true>::kry_lds_extra(a.S.N);
    auto kf = col_uslot<EPT>(a.S) ? k_forward_col<Q, EPT, true, true, false, true> : k_forward_col<Q, EPT, true, false, false, true>;
    hipError_t e = set_lds_col(kf, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3(col_grid(kf, a, 64 * (ST::ncols(a.S.N) / EPT), lds)), dim3(64 * (ST::ncols(a.S.N) / EPT)), lds, st, a);
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}
template <int Q, int EPT>
static hipError_t go_adj_col_k(const SweepArgs& a, hipStream_t st) {
  if constexpr (EPT == 5 || EPT == 8) {
    typedef ColLean<Q, EPT> ST;
    const size_t lds = ST::lds_bytes(a.S.N) + ColTeam<Q, EPT, true>::kry_lds_extra(a.S.N);
    auto kf = col_uslot<EPT>(a.S) ? k_adjoint_col<Q, EPT, true, true, false, true> : k_adjoint_col<Q, EPT, true, false, false, true>;
    hipError_t e = set_lds_col(kf, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3(col_grid(kf, a, 64 * (ST::ncols(a.S.N) / EPT), lds)), dim3(64 * (ST::ncols(a.S.N) / EPT)), lds, st, a);
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
}
template <int Q, int EPT>
static hipError_t go_fwd_col(const SweepArgs& a, hipStream_t st) {
  if (a.use_gmres) return go_fwd_col_k<Q, EPT>(a, st);
  return a.neumann_split ? go_fwd_col_s<Q, EPT, true>(a, st) : go_fwd_col_s<Q, EPT, false>(a, st);
}
template <int Q, int EPT, bool SPLIT>
static hipError_t go_adj_col_s(const SweepArgs& a, hipStream_t st) {
  typedef ColLean<Q, EPT> ST;
  const size_t lds = ST::lds_bytes(a.S.N);
  // (implicit midpoint only: the predictor compares the pass count of a sub-step with its predecessor's, and the sub-steps of a composite
  //  step differ in size - IMR4 / IMR8 test every pass; nothing of this lives in the kernels)
  const bool skip = a.rel2 < 1e-30f && !a.col_noskip && a.nstages == 1;
  auto kf = col_uslot<EPT>(a.S) ? (skip ? k_adjoint_col<Q, EPT, SPLIT, true, true> : k_adjoint_col<Q, EPT, SPLIT, true, false>)
                                : (skip ? k_adjoint_col<Q, EPT, SPLIT, false, true> : k_adjoint_col<Q, EPT, SPLIT, false, false>);
  hipError_t e = set_lds_col(kf, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3(col_grid(kf, a, 64 * (ST::ncols(a.S.N) / EPT), lds)), dim3(64 * (ST::ncols(a.S.N) / EPT)), lds, st, a);
  return hipGetLastError();
}
template <int Q, int EPT>
static hipError_t go_adj_col(const SweepArgs& a, hipStream_t st) {
  if (a.use_gmres) return go_adj_col_k<Q, EPT>(a, st);
  return a.neumann_split ? go_adj_col_s<Q, EPT, true>(a, st) : go_adj_col_s<Q, EPT, false>(a, st);
}
template <int Q, int EPT>
static hipError_t go_app_col(const DevSys& S, const double* ctlrow, int tr, const double* x, double* y, int nb, bool split, hipStream_t st) {
  typedef ColLean<Q, EPT> ST;
  const size_t lds = ST::lds_bytes(S.N);
  // (the option neumann_split = 1 selects the kernel family that re-derives the diagonal from its compact form: test coverage)
  auto kf = split ? k_apply_col<Q, EPT, true> : k_apply_col<Q, EPT, false>;
  hipError_t e = set_lds_col(kf, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3(nb), dim3(64 * (ST::ncols(S.N) / EPT)), lds, st, S, ctlrow, tr, x, y);
  return hipGetLastError();
}

// Columns per wave.  Measured on the 3 x 20 workload (3600 initial conditions x 100 steps, forward sweep, one lease): 4 columns
// (15 waves, 128 VGPRs, 63 spills) 58.9 ms, 5 columns (12 waves, 168 VGPRs) 48.4 ms, 6 columns (10 waves, 59 spills) 62.9 ms,
// 8 columns (8 waves, 234 VGPRs, no spills) 53.4 ms; the general column kernel of qd_device.h 73.1 ms.  Five columns per wave cover
// N <= 60, eight the rest.  The option col_ept overrides (measurements).
static int col_ept(int N, const TuneOpts& o) {
  const int f = o.col_ept;
  if (f == 4 || f == 6 || f == 8 || (f == 5 && N <= 60)) return f;
  return N <= 60 ? 5 : 8;
}
#define QD_COL_DISPATCH(FN, ...)                                  \
  do {                                                            \
    const int e = col_ept(Nn, o);                                      \
    if (Qn == 2) {                                                \
      if (e == 4) return FN<2, 4>(__VA_ARGS__);                   \
      if (e == 5 && Nn <= 60) return FN<2, 5>(__VA_ARGS__);       \
      if (e == 6) return FN<2, 6>(__VA_ARGS__);                   \
      return FN<2, 8>(__VA_ARGS__);                               \
    }                                                             \
    if (Qn == 3) {                                                \
      if (e == 4) return FN<3, 4>(__VA_ARGS__);                   \
      if (e == 5 && Nn <= 60) return FN<3, 5>(__VA_ARGS__);       \
      if (e == 6) return FN<3, 6>(__VA_ARGS__);                   \
      return FN<3, 8>(__VA_ARGS__);                               \
    }                                                             \
    return hipErrorInvalidValue;                                  \
  } while (0)

// (the Krylov kernels are built with five and eight columns per wave: the automatic choices)
static TuneOpts col_opts(const SweepArgs& a, const TuneOpts& o) {
  TuneOpts t = o;
  if (a.use_gmres && t.col_ept != 5 && t.col_ept != 8) t.col_ept = 0;
  return t;
}
hipError_t launch_forward_col(const SweepArgs& a, const TuneOpts& o0, hipStream_t st) {
  const int Qn = a.S.Q, Nn = a.S.N;
  const TuneOpts o = col_opts(a, o0);
  QD_COL_DISPATCH(go_fwd_col, a, st);
}
hipError_t launch_adjoint_col(const SweepArgs& a, const TuneOpts& o0, hipStream_t st) {
  const int Qn = a.S.Q, Nn = a.S.N;
  const TuneOpts o = col_opts(a, o0);
  QD_COL_DISPATCH(go_adj_col, a, st);
}
hipError_t launch_apply_col(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb, const TuneOpts& o, hipStream_t st) {
  const int Qn = S.Q, Nn = S.N;
  QD_COL_DISPATCH(go_app_col, S, ctlrow, transpose, x, y, nb, o.neumann_split == 1, st);
}

}  // namespace qd
