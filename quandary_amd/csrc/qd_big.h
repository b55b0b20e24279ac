// qd_big.h — sweeps for states that do not fit one CU's LDS (dim > 4096): the reference's matrix-free templates go up to
// <20,20> (Lindblad dim 160 000), <4,4,4,4> (65 536) and <3,3,3,3,3> (59 049) (src/mastereq.cpp:3046-3047, :3150-3151, :3202).
// A team of G workgroups (1024 threads each; G = 1 when the batch alone fills the chip) owns one initial condition for the whole
// time loop; the vectors of the step (state, right-hand side, solver iterates, adjoint state, ...) live in a per-state work area
// in global memory and are exchanged through L2 (G = 1: workgroup-scope visibility through the fences of __syncthreads()).  Each
// thread loops over its elements; the per-element invariants (digits, Delta, d) come from a table built once per
// system (k_big_table) instead of registers.  The stencil itself is GenStencil::apply / ::ladder of qd_device.h with the
// element's invariants loaded into the (one-slot) stencil object - the same code that the LDS kernels run.
//
// Teams (G > 1, few initial conditions of a large system - e.g. one pure state of <20,20>): the elements are dealt out over
// G x 1024 threads, every workgroup barrier of the sweep becomes a team barrier (a monotone counter in global memory,
// agent-scope release / acquire: L2 write-back and invalidate, so the exchange is correct across XCDs) and every reduction
// goes through per-workgroup partial sums that all members add up in the same order (bit-identical scalars in all members:
// the solver's control flow stays uniform over the team).  The members of a team are consecutive blocks, i.e. dealt over all
// eight XCDs (measured 1.3-2x faster than a team kept on ONE XCD - member j of team t = block 8 (G (t / 8) + j) + t % 8, selected
// with S.team_spread = 0 - which shares one L2 but also one XCD's share of the fabric: profiles/r2_big_probe.jsonl).  The kernel
// is launched cooperatively (co-residency is checked by the runtime: a team that cannot be resident is an error, never a hang).
#pragma once
#include <type_traits>

#include "qd_device.h"

namespace qd {

constexpr int BIG_NV = 10;      // work vectors per initial condition
constexpr int BIG_BLOCK = 1024;

// per-element invariants: coef = (Delta, d), dig = (packed bra digits, packed ket digits)
template <int Q, bool LIND>
__global__ void k_big_table(const DevSys S, double2* __restrict__ coef, uint2* __restrict__ dig) {
  constexpr int DB = packed_digit_bits(Q);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S.dim) return;
  const int I = LIND ? e % S.N : e, Ip = LIND ? e / S.N : 0;
  unsigned dbra = 0, dket = 0;
  int ia[Q], ipa[Q];
#pragma unroll
  for (int k = 0; k < Q; k++) {
    ia[k] = (I / S.post[k]) % S.n[k];
    ipa[k] = LIND ? (Ip / S.post[k]) % S.n[k] : 0;
    dbra |= (unsigned)ia[k] << (DB * k);
    dket |= (unsigned)ipa[k] << (DB * k);
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
  coef[e] = make_double2(hd - hdp, d);
  dig[e] = make_uint2(dbra, dket);
}

template <int Q, bool LIND, bool DENSE = false, bool ADJ = false>
struct BigTeam {
  // one slot, table-driven (non-hoisted) formulation; DENSE: user-supplied Hamiltonians, G(t) read from the table in global memory
  typedef typename std::conditional<DENSE, DenseStencil<Q, LIND, 1, 2>, GenStencil<Q, LIND, 1, 2>>::type ST;
  // neighbour reads in batches (GenStencil::apply_batched) wherever the batch has the registers (the Neumann / GMRES kernels are separate
  // instantiations for that reason): everywhere but the Schroedinger adjoint sweep, measured on one lease each - 20 x 20 Lindblad forward
  // 8.9 -> 7.8 ms, gradient 25.0 -> 23.2; 32^4 Schroedinger with six coupling pairs forward 8.2 -> 7.3 ms, but gradient 20.9 -> 23.5
  // with a batched adjoint sweep (207 spilt registers)
  static constexpr bool kBatch = LIND || !ADJ;
  ST st;
  Lds L;
  int dim, redslot;
  const double2* coef;
  const uint2* dig;
  // team of G workgroups on one initial condition: element loops run e = gtid, gtid + gnt, ... < gend.  Blocked (default): member m
  // owns the contiguous elements [m chunk, (m + 1) chunk) - its own elements and most stencil neighbours (strides 1, n_Q, n_Q n_{Q-1},
  // ...) stay in the L2 of its XCD.  Strided (big_blocked = 0): every member walks the whole vector.
  int G, member, ic, gtid, gnt, gend, gslot;
  unsigned long long* bar;
  unsigned long long bar_target, sub_target;  // (thread 0 only)
  double2* scratch;  // one more work vector of the state (polynomial preconditioner of gmres)
  double* gred;

  static size_t lds_bytes(const DevSys& S) {
    return sizeof(double) * 2 * (size_t)table_len(S) + sizeof(double) * 2 * NRED * (BIG_BLOCK / 64) + sizeof(double) * gmres_nsc(GMRES_MR_G);
  }

  // false: this workgroup belongs to no team (padding of the XCD-aware grid) and must leave the kernel
  __device__ __forceinline__ bool init(const DevSys& S, unsigned char* smem, int nb) {
    dim = S.dim;
    redslot = 0;
    G = S.team > 1 ? S.team : 1;
    if (G == 1) {
      ic = blockIdx.x;
      member = 0;
    } else if (S.team_spread & 1) {
      ic = blockIdx.x / G;
      member = blockIdx.x % G;
    } else {
      const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
      ic = (r / G) * 8 + xcd;
      member = r % G;
    }
    if (ic >= nb) return false;
    if (G > 1 && (S.team_spread & 6)) {
      const int chunk = ((dim + G - 1) / G + 63) & ~63;
      // (4): the members of one XCD (equal member % 8 under round-robin dispatch) own neighbouring blocks
      const int blk = ((S.team_spread & 4) && G >= 8 && (S.team_spread & 1)) ? (member & 7) * (G >> 3) + (member >> 3) : member;
      gtid = blk * chunk + threadIdx.x;
      gnt = blockDim.x;
      gend = min(dim, (blk + 1) * chunk);
    } else {
      gtid = member * blockDim.x + threadIdx.x;
      gnt = G * blockDim.x;
      gend = dim;
    }
    gslot = 0;
    bar = S.tbar + (size_t)ic * BIG_BAR_STRIDE;
    bar_target = 0;
    sub_target = 0;
    gred = S.tred + (size_t)ic * 2 * BIG_TEAM_MAX * BIG_RED_NV;
    scratch = reinterpret_cast<double2*>(S.work) + ((size_t)ic * BIG_NV + 9) * dim;  // slot 9: unused by both sweeps
    coef = reinterpret_cast<const double2*>(S.ecoef);
    dig = reinterpret_cast<const uint2*>(S.edig);
    const int tl = table_len(S);
    L.buf0 = nullptr;
    L.bstride = 0;
    L.tup = reinterpret_cast<double*>(smem);
    L.tdn = L.tup + tl;
    L.red = L.tdn + tl;
    L.bvec = nullptr; L.gmat = nullptr; L.coltab = nullptr; L.kry = nullptr;
    L.ksc = L.red + 2 * NRED * (BIG_BLOCK / 64);
    int o = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      st.ofs[k] = o;
      o += S.n[k];
    }
    for (int k = 0; k < Q; k++)
      for (int a = threadIdx.x; a < S.n[k]; a += blockDim.x) {
        L.tup[st.ofs[k] + a] = (a < S.n[k] - 1) ? sqrt((double)(a + 1)) : 0.0;
        L.tdn[st.ofs[k] + a] = sqrt((double)a);
      }
    st.valid[0] = true;
    __syncthreads();
    return true;
  }
  // Barrier over the team.  G = 1: the workgroup barrier (workgroup-scope fences).  G > 1: every wave waits until L2 has
  // acknowledged its stores, then ONE thread per workgroup releases at agent scope (L2 write-back: the other XCDs can see the
  // data), arrives at the team's counter, waits for the G arrivals of this round and acquires (L1 / L2 invalidate, which holds for
  // the whole CU / XCD); the workgroup barrier hands the result to the other waves.  Fences issued by all 16 waves instead cost
  // 4-8x more (profiles/r2_barrier_probe.jsonl: 2.5 us against 10 us per barrier at G = 32).  From 64 members on the arrivals go
  // through eight first-level counters (members with equal blockIdx % 8, i.e. one XCD under round-robin dispatch), the last
  // arrival of a group reports to the team's counter: 5.9 us instead of 8.2 us at G = 256.
  __device__ __forceinline__ void tsync() {
    if (G == 1) {
      __syncthreads();
      return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (G >= 64) {
        bar_target += 8ull;
        sub_target += (unsigned long long)(G / 8);
        unsigned long long* sub = bar + 16 * (1 + (member & 7));
        const unsigned long long prev = __hip_atomic_fetch_add(sub, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == sub_target) __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        bar_target += (unsigned long long)G;
        __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bar_target) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  // make element e the stencil's slot
  __device__ __forceinline__ void at(int e) {
    const double2 cf = coef[e];
    const uint2 dg = dig[e];
    st.it[0] = e;
    st.dw[0] = cf.x;
    st.dd[0] = cf.y;
    st.dbra[0] = dg.x;
    st.dket[0] = dg.y;
  }
  template <bool TRANS>
  __device__ __forceinline__ double2 apply(const DevSys& S, const StepC<Q>& c, const double2* __restrict__ src, int e) {
    at(e);
    if constexpr (DENSE || !kBatch) return st.template apply<TRANS>(S, L, src, c, 0, src[e]);
    else return st.template apply_batched<TRANS>(S, L, src, c, 0, src[e]);
  }
  template <int NV>
  __device__ __forceinline__ void sum(double (&v)[NV]) {
    static_assert(NV <= BIG_RED_NV, "team reduction buffer too small");
    if (G == 1) {
      block_sum<NV, false>(v, L.red + redslot * NRED * (BIG_BLOCK / 64));
      redslot ^= 1;
      return;
    }
    // team: partial sums of the members in global memory ([value][member], two buffers: see tsync's ordering), added up by the
    // first wave of every member in the same order (lanes over members, then the wave's butterfly), handed to the other waves
    // through LDS.  LDS slot 0 serves the workgroup's own reduction, slot 1 the team's result.
    block_sum<NV, false>(v, L.red);
    double* slot = gred + (size_t)gslot * BIG_TEAM_MAX * BIG_RED_NV;
    gslot ^= 1;
    if (threadIdx.x < NV) {
      double mine = 0.0;
#pragma unroll
      for (int i = 0; i < NV; i++)
        if ((int)threadIdx.x == i) mine = v[i];
      slot[threadIdx.x * BIG_TEAM_MAX + member] = mine;
    }
    tsync();
    double* res = L.red + NRED * (BIG_BLOCK / 64);
    if (threadIdx.x < 64) {
#pragma unroll
      for (int i = 0; i < NV; i++) {
        double t = 0.0;
        for (int m = threadIdx.x; m < G; m += 64) t += slot[i * BIG_TEAM_MAX + m];
        t = wave_sum(t);
        if (threadIdx.x == 0) res[i] = t;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = res[i];
  }
  __device__ __forceinline__ float sum_f32(float v) {
    if (G > 1) {
      double t[1] = {(double)v};
      sum<1>(t);
      return (float)t[0];
    }
    double* red = L.red + redslot * NRED * (BIG_BLOCK / 64);
    redslot ^= 1;
    return block_sum_f32<false>(v, red);
  }

  // the off-diagonal part of the operator at element e: the stencil takes the element's own value as a separate argument and uses it
  // for the diagonal terms only
  template <bool TRANS>
  __device__ __forceinline__ double2 apply_off(const DevSys& S, const StepC<Q>& c, const double2* __restrict__ src, int e) {
    at(e);
    if constexpr (DENSE || !kBatch) return st.template apply<TRANS>(S, L, src, c, 0, make_double2(0.0, 0.0));
    else return st.template apply_batched<TRANS>(S, L, src, c, 0, make_double2(0.0, 0.0));
  }

  // Neumann iteration (timestepper.cpp:697-727): (I - alpha M^{(T)}) y = b, b in Bv; iterates alternate between Ya and Yb.
  // Returns the vector that holds the solution; *iters = RHS applications.
  // A.neumann_split (not for user-supplied dense Hamiltonians, whose diagonal sits inside G(t)): the diagonal-split form of qd_col.hip,
  // y <- (1 - alpha D)^-1 (b + alpha (M - D) y) from y_0 = (1 - alpha D)^-1 b, D = d -+ i Delta from the per-element table - same fixed
  // point, same stopping rule on the update norm.  A.stop_residual (a GMRES request served by this iteration, qd_handle::gmres_as_split):
  // stop on the residual of the iterate, ||b - (I - alpha M) y_m|| = ||(1 - alpha D)(y_{m+1} - y_m)|| <= max(rtol ||b||, abstol), which
  // is KSPGMRES's rule (src/timestepper.cpp:541-550); the factor is applied element by element, so the norm is exact.
  // y0_ready: the caller has produced y_0 in Yb (split_y0 in the loop that wrote b) and the team sum of |b|^2 in nb2s.
  __device__ __forceinline__ bool split_on(const SweepArgs& A) const { return !DENSE && A.neumann_split; }
  // (1 - alpha D)^-1 v at the element of the last at() / apply()
  template <bool TRANS>
  __device__ __forceinline__ double2 split_y0(double alpha, const double2 v) const {
    const double re = fma(-alpha, st.dd[0], 1.0), im = (TRANS ? -alpha : alpha) * st.dw[0];
    const double inv = 1.0 / fma(re, re, im * im);
    const double pr = re * inv, pi = -im * inv;
    return make_double2(fma(pr, v.x, -pi * v.y), fma(pr, v.y, pi * v.x));
  }
  template <bool TRANS>
  __device__ __forceinline__ double2* neumann(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2* __restrict__ Bv, double2* Ya,
                                              double2* Yb, int* iters, bool y0_ready = false, double nb2s = 0.0) {
    // y_0 = b is read in place (the caller has put a team barrier behind the last write to Bv); the iterates alternate Ya / Yb
    const double inv_abs2 = 1.0 / (A.abstol * A.abstol);
    float rel2 = (float)(A.reltol * A.reltol), thr = 1.f;
    float d0 = 1.f, dprev = 1.f;
    int iter, wb = 0;
    double2* const bufs[2] = {Ya, Yb};
    const double2* cur = Bv;
    const bool split = !DENSE && A.neumann_split;
    const bool resid = split && A.stop_residual;
    const double sal = TRANS ? -alpha : alpha;
    if (split) {
      double nb2[1] = {nb2s};
      if (!y0_ready) {
        for (int e = gtid; e < gend; e += gnt) {
          const double2 cf = coef[e], bj = Bv[e];
          const double re = fma(-alpha, cf.y, 1.0), im = sal * cf.x;  // 1 - alpha D
          const double inv = 1.0 / fma(re, re, im * im);
          const double pr = re * inv, pi = -im * inv;
          Yb[e] = make_double2(fma(pr, bj.x, -pi * bj.y), fma(pr, bj.y, pi * bj.x));
          nb2[0] = fma(bj.x, bj.x, fma(bj.y, bj.y, nb2[0]));
        }
        if (resid) sum<1>(nb2);  // (contains the team barrier)
        else tsync();
      }
      if (resid) {
        thr = (float)fmin(fmax(A.reltol * A.reltol * nb2[0] * inv_abs2, 1.0), 1e30);
        rel2 = 0.f;
      }
      cur = Yb;
    }
    for (iter = 0; iter < A.maxiter; iter++) {
      double2* nxt = bufs[wb];
      double dl = 0.0;
      if (split) {
        for (int e = gtid; e < gend; e += gnt) {
          const double2 t = apply_off<TRANS>(A.S, c, cur, e);
          const double2 bj = Bv[e], yo = cur[e];
          const double re = fma(-alpha, st.dd[0], 1.0), im = sal * st.dw[0];
          const double n2 = fma(re, re, im * im), inv = 1.0 / n2;
          const double pr = re * inv, pi = -im * inv;
          const double ux = fma(alpha, t.x, bj.x), uy = fma(alpha, t.y, bj.y);
          double2 w;
          w.x = fma(pr, ux, -pi * uy);
          w.y = fma(pr, uy, pi * ux);
          const double dx = yo.x - w.x, dy = yo.y - w.y;
          const double d2 = dx * dx + dy * dy;
          dl += resid ? n2 * d2 : d2;
          nxt[e] = w;
        }
      } else {
        for (int e = gtid; e < gend; e += gnt) {
          const double2 t = apply<TRANS>(A.S, c, cur, e);
          const double2 bj = Bv[e], yo = cur[e];
          double2 w;
          w.x = fma(alpha, t.x, bj.x);
          w.y = fma(alpha, t.y, bj.y);
          const double dx = yo.x - w.x, dy = yo.y - w.y;
          dl += dx * dx + dy * dy;
          nxt[e] = w;
        }
      }
      const float d = sum_f32((float)fmin(dl * inv_abs2, 1e30));  // contains the (team) barrier and its fences
      cur = nxt;
      wb ^= 1;
      // (one exit branch per iteration, first-iteration values by selects [r5]: qd_q32.hip / qd_col.hip measured 1 - 7 %)
      d0 = iter == 0 ? d : d0;
      {
        const float dp = iter == 0 ? d : dprev;
        const bool stop = (d < thr && standin_ok(A.standin_tau2, d, dp, thr)) | (d < rel2 * d0);
        dprev = d;
        if (stop) { iter++; break; }
      }
    }
    *iters = iter;
    return const_cast<double2*>(cur);
  }

  // GMRES (KSPGMRES + PCNONE of the reference, as Team::gmres_g with p = 1): zero initial guess, classical Gram-Schmidt,
  // Givens rotations, restart 30, stop at max(rtol ||b||, abstol).  Every vector lives in global memory; a thread only ever
  // touches its own elements (e = tid + n blockDim) except in the operator application, which is fenced by a barrier.
  // Krylov basis: A.kry, [nb][GMRES_MR_G + 2][dim]; Wv: scratch vector; the solution is written to Ysol.
  template <bool TRANS>
  __device__ __forceinline__ double2* gmres(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2* __restrict__ Bv, double2* Ysol,
                                            double2* Wv, int* iters) {
    constexpr int MR = GMRES_MR_G;
    const int nt = gnt, tid = gtid, eend = gend;
    double2* Vg = reinterpret_cast<double2*>(A.kry) + (size_t)ic * (MR + 2) * dim;
    double* hc = L.ksc;
    double* cs = hc + (MR + 2);
    double* sn = cs + MR;
    double* g = sn + MR;
    double* Rm = g + (MR + 2);
    double* yk = Rm + MR * MR;
    // A.gmres_poly = p > 1: right preconditioning with the Neumann polynomial (Team::gmres_g of qd_device.h: GMRES on
    // (I - alpha M) P = I - (alpha M)^p, the preconditioned vectors z_j = P v_j stored next to the basis, restart 14).  Every team
    // barrier and reduction saved counts double here: one Krylov vector per solve once the host has tuned p (forward_finish).
    const int poly = A.gmres_poly > 1 ? A.gmres_poly : 1;
    const int MRE = poly > 1 ? (MR - 2) / 2 : MR;
    double2* Zg = Vg + (size_t)(MRE + 1) * dim;
    const double2* Sg = poly > 1 ? Zg : Vg;
    for (int e = tid; e < eend; e += nt) Ysol[e] = make_double2(0.0, 0.0);
    int its = 0, napp = 0;
    double ttol = 0.0;
    for (int cycle = 0;; cycle++) {
      // residual: b on the first cycle, b - (I - alpha M) y afterwards (left in Wv by the restart code below)
      const double2* r = cycle == 0 ? Bv : Wv;
      double t1[1] = {0.0};
      for (int e = tid; e < eend; e += nt) {
        const double2 v = r[e];
        t1[0] += v.x * v.x + v.y * v.y;
      }
      sum<1>(t1);
      const double ibeta = t1[0] > 0.0 ? rsqrt_nr(t1[0]) : 0.0;
      const double beta = t1[0] * ibeta;
      if (cycle == 0) ttol = fmax(A.reltol * beta, A.abstol);
      if (beta <= ttol || its >= A.maxiter) break;
      for (int e = tid; e < eend; e += nt) {
        const double2 v = r[e];
        Vg[e] = make_double2(v.x * ibeta, v.y * ibeta);
      }
      tsync();
      double gcur = beta;
      int jj = 0;
      bool conv = false;
      while (jj < MRE) {
        const double2* vj = Vg + (size_t)jj * dim;
        // z = P v_jj by Horner's rule (z <- v + alpha M z, p - 1 times), alternating between the scratch vector and z's final place
        const double2* z = vj;
        for (int i = 1; i < poly; i++) {
          double2* dst = ((poly - 1 - i) & 1) ? scratch : Zg + (size_t)jj * dim;
          for (int e = tid; e < eend; e += nt) {
            const double2 t = apply<TRANS>(A.S, c, z, e);
            const double2 v = vj[e];
            dst[e] = make_double2(fma(alpha, t.x, v.x), fma(alpha, t.y, v.y));
          }
          napp++;
          tsync();
          z = dst;
        }
        for (int e = tid; e < eend; e += nt) {  // w = (I - alpha M) z
          const double2 t = apply<TRANS>(A.S, c, z, e);
          const double2 v = z[e];
          Wv[e] = make_double2(v.x - alpha * t.x, v.y - alpha * t.y);
        }
        napp++;
        for (int k0 = 0; k0 <= jj; k0 += 4) {  // classical Gram-Schmidt, four projections per pass and reduction
          double h4[4] = {0.0, 0.0, 0.0, 0.0};
          const int nk = min(4, jj + 1 - k0);
          for (int e = tid; e < eend; e += nt) {
            const double2 w = Wv[e];
            for (int q = 0; q < nk; q++) {
              const double2 vk = Vg[(size_t)(k0 + q) * dim + e];
              h4[q] += w.x * vk.x + w.y * vk.y;
            }
          }
          sum<4>(h4);
          for (int q = 0; q < nk; q++) hc[k0 + q] = h4[q];
        }
        double nn[1] = {0.0};
        for (int e = tid; e < eend; e += nt) {
          double2 w = Wv[e];
          for (int k = 0; k <= jj; k++) {
            const double h = hc[k];
            const double2 vk = Vg[(size_t)k * dim + e];
            w.x -= h * vk.x;
            w.y -= h * vk.y;
          }
          Wv[e] = w;
          nn[0] += w.x * w.x + w.y * w.y;
        }
        sum<1>(nn);
        const double ihn = nn[0] > 0.0 ? rsqrt_nr(nn[0]) : 0.0;
        const double hn = nn[0] * ihn;
        hc[jj + 1] = hn;
        double cur_h = hc[0];  // Givens rotations: redundantly by every thread on uniform values, idempotent LDS writes only
        for (int k = 0; k < jj; k++) {
          const double a1 = hc[k + 1], ck = cs[k], sk = sn[k];
          Rm[k * MR + jj] = ck * cur_h + sk * a1;
          cur_h = -sk * cur_h + ck * a1;
        }
        const double a0 = cur_h, bb = hn;
        const double s2 = a0 * a0 + bb * bb;
        const double irr = s2 > 0.0 ? rsqrt_nr(s2) : 0.0;
        const double cj = s2 > 0.0 ? a0 * irr : 1.0, sj = bb * irr;
        cs[jj] = cj;
        sn[jj] = sj;
        Rm[jj * MR + jj] = irr;  // the diagonal is only ever divided by: keep its reciprocal
        g[jj] = cj * gcur;
        gcur = -sj * gcur;
        its++;
        jj++;
        if (fabs(gcur) <= ttol || hn == 0.0) { conv = true; break; }
        if (its >= A.maxiter || jj >= MRE) break;
        // the next basis vector is only formed and stored when another iteration follows
        for (int e = tid; e < eend; e += nt) {
          const double2 w = Wv[e];
          Vg[(size_t)jj * dim + e] = make_double2(w.x * ihn, w.y * ihn);
        }
        tsync();  // v_{jj} complete (next application reads neighbours), scalars ordered
      }
      for (int rw = jj - 1; rw >= 0; rw--) {
        double sacc = g[rw];
        for (int cc = rw + 1; cc < jj; cc++) sacc -= Rm[rw * MR + cc] * yk[cc];
        yk[rw] = sacc * Rm[rw * MR + rw];
      }
      for (int e = tid; e < eend; e += nt) {
        double2 y = Ysol[e];
        for (int cc = 0; cc < jj; cc++) {
          const double f = yk[cc];
          const double2 vk = Sg[(size_t)cc * dim + e];
          y.x += f * vk.x;
          y.y += f * vk.y;
        }
        Ysol[e] = y;
      }
      tsync();
      if (conv || its >= A.maxiter) break;
      for (int e = tid; e < eend; e += nt) {  // restart: r = b - (I - alpha M) y
        const double2 t = apply<TRANS>(A.S, c, Ysol, e);
        const double2 y = Ysol[e], b = Bv[e];
        Wv[e] = make_double2(b.x - (y.x - alpha * t.x), b.y - (y.y - alpha * t.y));
      }
      napp++;
      tsync();
    }
    *iters = napp;
    return Ysol;
  }

  // GM: the Krylov solver - a kernel of its own (the stationary iterations keep their registers for the neighbour reads)
  template <bool TRANS, bool GM>
  __device__ __forceinline__ double2* solve(const SweepArgs& A, const StepC<Q>& c, double alpha, const double2* __restrict__ Bv, double2* Ya,
                                            double2* Yb, int* iters, bool y0_ready = false, double nb2s = 0.0) {
    if constexpr (GM) return gmres<TRANS>(A, c, alpha, Bv, Ya, Yb, iters);
    else return neumann<TRANS>(A, c, alpha, Bv, Ya, Yb, iters, y0_ready, nb2s);
  }
};

template <int Q, bool LIND, bool DENSE = false, bool GM = false>
__global__ void __launch_bounds__(BIG_BLOCK) k_forward_big(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef BigTeam<Q, LIND, DENSE> TM;
  const DevSys& S = A.S;
  TM tm;
  if (!tm.init(S, smem, A.nb)) return;
  const int dim = S.dim, ic = tm.ic, nt = tm.gnt, tid = tm.gtid, eend = tm.gend;
  double2* W = reinterpret_cast<double2*>(S.work) + (size_t)ic * BIG_NV * dim;
  double2 *X = W, *B = W + dim, *Ya = W + 2 * (size_t)dim, *Yb = W + 3 * (size_t)dim, *XM1 = W + 4 * (size_t)dim, *XM2 = W + 5 * (size_t)dim;
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  const bool wj_reduce = wj_on && !LIND && A.tg.objective_type == QD_OBJ_JTRACE;
  const bool dpdm_on = A.gamma_dpdm > 1e-13 && !LIND;
  const bool jpairs = S.npairs > 0;
  {
    const double* x0 = A.x0 + (size_t)ic * 2 * dim;
    for (int e = tid; e < eend; e += nt) {
      const double2 v = make_double2(x0[e], x0[dim + e]);
      X[e] = v;
      if (dpdm_on) XM1[e] = XM2[e] = v;
    }
  }
  tm.tsync();
  double pen_local = 0.0, dpdm_local = 0.0, pen_uniform = 0.0;
  unsigned long long napply = 0;
  const double dtinv4 = 1.0 / (A.dt * A.dt * A.dt * A.dt);
  vm_drain();
  for (int s = 0; s < A.nsub; s++) {
    StepC<Q> c;
    load_step_k<Q>(A.ctl + (size_t)s * A.cs, c, jpairs);
    scalarize<Q>(c, jpairs);
    c.g = DENSE ? reinterpret_cast<const double2*>(S.gtab) + (size_t)s * S.N * S.N : nullptr;
    if (A.traj) {
      double* dst = A.traj + ((size_t)s * A.nb + ic) * 2 * dim;
      for (int e = tid; e < eend; e += nt) {
        const double2 v = X[e];
        dst[e] = v.x;
        dst[dim + e] = v.y;
      }
    }
    // rhs = M x; diagonal-split iteration: its first iterate (1 - h/2 D)^-1 rhs (and |rhs|^2 for the residual rule) from the same loop
    const bool y0 = !GM && !A.stepper_ee && tm.split_on(A);
    double nb2[1] = {0.0};
    if (y0) {
      for (int e = tid; e < eend; e += nt) {
        const double2 b = tm.template apply<false>(S, c, X, e);
        B[e] = b;
        Yb[e] = tm.template split_y0<false>(0.5 * c.h, b);
        nb2[0] = fma(b.x, b.x, fma(b.y, b.y, nb2[0]));
      }
    } else {
      for (int e = tid; e < eend; e += nt) B[e] = tm.template apply<false>(S, c, X, e);
    }
    napply++;
    if (y0 && A.stop_residual) tm.template sum<1>(nb2);  // (contains the team barrier)
    else tm.tsync();
    if (A.stepper_ee) {
      for (int e = tid; e < eend; e += nt) {
        const double2 r = B[e];
        double2 v = X[e];
        v.x = fma(c.h, r.x, v.x);
        v.y = fma(c.h, r.y, v.y);
        X[e] = v;
      }
    } else {
      int its;
      const double2* K = tm.template solve<false, GM>(A, c, 0.5 * c.h, B, Ya, Yb, &its, y0, nb2[0]);
      napply += its;
      double* zdst = A.ztraj ? A.ztraj + ((size_t)s * A.nb + ic) * 2 * dim : nullptr;
      for (int e = tid; e < eend; e += nt) {
        const double2 k = K[e];
        double2 v = X[e];
        if (zdst)  // the primal stage z = x + h/2 k: read back by the adjoint sweep instead of repeating this solve (interleaved pairs)
          reinterpret_cast<double2*>(zdst)[e] = make_double2(fma(0.5 * c.h, k.x, v.x), fma(0.5 * c.h, k.y, v.y));
        v.x = fma(c.h, k.x, v.x);
        v.y = fma(c.h, k.y, v.y);
        X[e] = v;
      }
    }
    tm.tsync();
    // in-loop penalties at the end of a FULL time step (timestepper.cpp:141-154)
    if ((pen_on || dpdm_on) && (s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages - 1;
      const double tstop = (n + 1) * A.dt;
      if (pen_on) {
        double weight = 0.0;
        if (wj_on) {
          const double a = (tstop - A.Tfinal) / A.penalty_param;
          weight = 1.0 / A.penalty_param * exp(-(a * a));
        }
        if (wj_reduce) {
          double v[2] = {0.0, 0.0};
          for (int e = tid; e < eend; e += nt) evalJ_part<LIND>(S, A.tg, ic, e, X[e], v[0], v[1]);
          tm.template sum<2>(v);
          pen_uniform += weight * finalizeJ<LIND>(A.tg, v[0], v[1]) * A.dt;
        } else if (wj_on) {
          for (int e = tid; e < eend; e += nt) {
            double jr = 0.0, ji = 0.0;
            evalJ_part<LIND>(S, A.tg, ic, e, X[e], jr, ji);
            pen_local += (A.tg.objective_type == QD_OBJ_JTRACE ? -1.0 : 1.0) * weight * A.dt * jr;
          }
          if (A.tg.objective_type == QD_OBJ_JTRACE) pen_uniform += weight * A.dt;
        }
        if (A.leak_on) {
          for (int e = tid; e < eend; e += nt) {
            tm.at(e);
            if (tm.st.is_guard(S, 0)) {
              const double2 v = X[e];
              pen_local += (v.x * v.x + v.y * v.y) / A.ntime;
            }
          }
        }
      }
      if (dpdm_on) {
        for (int e = tid; e < eend; e += nt) {
          const double2 v = X[e], m1 = XM1[e], m2 = XM2[e];
          if (n > 0) {
            const double t1 = v.x * v.x - 2.0 * m1.x * m1.x + m2.x * m2.x;
            const double t2 = v.y * v.y - 2.0 * m1.y * m1.y + m2.y * m2.y;
            dpdm_local += dtinv4 * (t1 + t2) * (t1 + t2);
          }
          XM2[e] = m1;
          XM1[e] = v;
        }
      }
    }
  }
  {
    double* xT = A.xT + (size_t)ic * 2 * dim;
    double* dst = A.traj ? A.traj + ((size_t)A.nsub * A.nb + ic) * 2 * dim : nullptr;
    for (int e = tid; e < eend; e += nt) {
      const double2 v = X[e];
      xT[e] = v.x;
      xT[dim + e] = v.y;
      if (dst) {
        dst[e] = v.x;
        dst[dim + e] = v.y;
      }
    }
  }
  double v[2] = {pen_local, dpdm_local};
  tm.template sum<2>(v);
  if (tid == 0) {
    A.pen_out[ic] = v[0] + pen_uniform;
    A.dpdm_out[ic] = v[1] / A.ntime;
    atomicAdd(A.napply, napply);
  }
}

// EE: explicit Euler (its backward chain and its step are a kernel of their own: the implicit-midpoint sweep keeps its registers)
template <int Q, bool LIND, bool DENSE = false, bool GM = false, bool EE = false>
__global__ void __launch_bounds__(BIG_BLOCK) k_adjoint_big(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef BigTeam<Q, LIND, DENSE, true> TM;
  const DevSys& S = A.S;
  TM tm;
  if (!tm.init(S, smem, A.nb)) return;
  const int dim = S.dim, ic = tm.ic, nt = tm.gnt, tid = tm.gtid, eend = tm.gend;
  double2* W = reinterpret_cast<double2*>(S.work) + (size_t)ic * BIG_NV * dim;
  double2 *Ya = W + 2 * (size_t)dim, *Yb = W + 3 * (size_t)dim, *Z = W + 6 * (size_t)dim, *KB = W + 7 * (size_t)dim, *XB = W + 8 * (size_t)dim;
  const double* traj = A.traj;
  auto state = [&](int s, int e) {
    const double* src = traj + ((size_t)s * A.nb + ic) * 2 * dim;
    return make_double2(src[e], src[dim + e]);
  };
  {
    const double* xbT = A.xbarT + (size_t)ic * 2 * dim;
    for (int e = tid; e < eend; e += nt) XB[e] = make_double2(xbT[e], xbT[dim + e]);
  }
  const double jbar_pen = A.jbar[ic * 3 + 0], jbar_dpdm = A.jbar[ic * 3 + 1];
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  const bool wj_reduce = wj_on && !LIND && A.tg.objective_type == QD_OBJ_JTRACE;
  const bool dpdm_on = A.gamma_dpdm > 1e-13 && !LIND;
  const bool jpairs = S.npairs > 0;
  const double dtinv4 = 1.0 / (A.dt * A.dt * A.dt * A.dt);
  const int ntime = A.ntime;
  tm.tsync();
  if constexpr (EE && !LIND) {
    // Explicit Euler in Schroedinger mode [r3: also beyond dim 4096]: the reference re-computes the primal backwards with the FORWARD
    // stepper and a negative step (src/timestepper.cpp:229-231 with ExplEuler::evolveFWD :496-507), and its gradient is defined on
    // that chain - reproduce it by overwriting the stored trajectory (k_adjoint of qd_device.h does the same in registers).
    double* trajw = const_cast<double*>(traj);
    for (int e = tid; e < eend; e += nt) Ya[e] = state(A.nsub, e);
    tm.tsync();
    for (int s = A.nsub - 1; s >= 0; s--) {
      StepC<Q> c1;
      load_step_k<Q>(A.ctl + (size_t)(s + 1) * A.cs, c1, jpairs);  // M(tstop of step s) = row s + 1
      scalarize<Q>(c1, jpairs);
      c1.g = DENSE ? reinterpret_cast<const double2*>(S.gtab) + (size_t)(s + 1) * S.N * S.N : nullptr;
      const double hneg = -to_scalar(A.ctl[(size_t)s * A.cs]);
      for (int e = tid; e < eend; e += nt) KB[e] = tm.template apply<false>(S, c1, Ya, e);
      tm.tsync();
      double* dst = trajw + ((size_t)s * A.nb + ic) * 2 * dim;
      for (int e = tid; e < eend; e += nt) {
        const double2 t = KB[e];
        double2 v = Ya[e];
        v.x = fma(hneg, t.x, v.x);
        v.y = fma(hneg, t.y, v.y);
        Ya[e] = v;
        dst[e] = v.x;
        dst[dim + e] = v.y;
      }
      tm.tsync();
    }
  }
  vm_drain();
  for (int s = A.nsub - 1; s >= 0; s--) {
    // ---- penalty adjoints at the end of a full step, with the primal x_n (timestepper.cpp:220-227)
    if ((pen_on || dpdm_on) && (s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages;
      const double tstop = n * A.dt;
      double rb = 0.0, ib = 0.0, weight = 0.0;
      if (wj_on) {
        const double a = (tstop - A.Tfinal) / A.penalty_param;
        weight = 1.0 / A.penalty_param * exp(-(a * a));
        if (wj_reduce) {
          double v[2] = {0.0, 0.0};
          for (int e = tid; e < eend; e += nt) evalJ_part<LIND>(S, A.tg, ic, e, state(s + 1, e), v[0], v[1]);
          tm.template sum<2>(v);
          finalizeJ_diff<LIND>(A.tg, v[0], v[1], rb, ib);
        } else {
          finalizeJ_diff<LIND>(A.tg, 0.0, 0.0, rb, ib);
        }
      }
      for (int e = tid; e < eend; e += nt) {
        const double2 xn = state(s + 1, e);
        double2 xb = XB[e];
        if (dpdm_on) {  // penaltyDpDm_diff (timestepper.cpp:372-442)
          const double Jb = jbar_dpdm / ntime, xr = xn.x, xi = xn.y;
          double acc = 0.0;
          double2 m1 = make_double2(0, 0), m2 = m1, p1 = m1, p2 = m1;
          if (n > 1) m2 = state((n - 2) * A.nstages, e);
          if (n > 0) m1 = state((n - 1) * A.nstages, e);
          if (n < ntime) p1 = state((n + 1) * A.nstages, e);
          if (n < ntime - 1) p2 = state((n + 2) * A.nstages, e);
          if (n > 1) acc += 2.0 * ((m2.x * m2.x - 2.0 * m1.x * m1.x + xr * xr) + (m2.y * m2.y - 2.0 * m1.y * m1.y + xi * xi));
          if (n > 0 && n < ntime) acc += -4.0 * ((m1.x * m1.x - 2.0 * xr * xr + p1.x * p1.x) + (m1.y * m1.y - 2.0 * xi * xi + p1.y * p1.y));
          if (n < ntime - 1) acc += 2.0 * ((xr * xr - 2.0 * p1.x * p1.x + p2.x * p2.x) + (xi * xi - 2.0 * p1.y * p1.y + p2.y * p2.y));
          xb.x += acc * 2.0 * xr * dtinv4 * Jb;
          xb.y += acc * 2.0 * xi * dtinv4 * Jb;
        }
        if (wj_on) evalJ_diff_elem<LIND>(S, A.tg, ic, e, xn, xb, weight * rb * jbar_pen * A.dt, weight * ib * jbar_pen * A.dt);
        if (pen_on && A.leak_on) {
          tm.at(e);
          if (tm.st.is_guard(S, 0)) {
            xb.x += 2.0 * xn.x * jbar_pen / ntime;
            xb.y += 2.0 * xn.y * jbar_pen / ntime;
          }
        }
        XB[e] = xb;
      }
      tm.tsync();  // the adjoint solve reads xbar in place
    }
    StepC<Q> c;
    load_step_k<Q>(A.ctl + (size_t)s * A.cs, c, jpairs);
    scalarize<Q>(c, jpairs);
    c.g = DENSE ? reinterpret_cast<const double2*>(S.gtab) + (size_t)s * S.N * S.N : nullptr;
    double cf[2 * Q];
#pragma unroll
    for (int i = 0; i < 2 * Q; i++) cf[i] = 0.0;
    if constexpr (EE) {
      // ExplEuler::evolveBWD (timestepper.cpp:506-520): gradient with dt x_adj against x_{n-1}, then x_adj += dt M(tstop)^T x_adj;
      // M(tstop) is table row s + 1 (the last row is followed by one extra row for t = T)
      for (int e = tid; e < eend; e += nt) Z[e] = state(s, e);
      tm.tsync();
      for (int e = tid; e < eend; e += nt) {
        tm.at(e);
        const double2 xb = XB[e];
#pragma unroll
        for (int k = 0; k < Q; k++) {
          double2 Av, Bv;
          tm.st.ladder(S, tm.L, Z, k, 0, Av, Bv);
          cf[2 * k] += c.h * (Bv.y * xb.x - Bv.x * xb.y);
          cf[2 * k + 1] += c.h * (Av.x * xb.x + Av.y * xb.y);
        }
      }
      tm.template sum<2 * Q>(cf);
      {
        double* co = A.coeff + ((size_t)ic * A.nsub + s) * 2 * Q;
#pragma unroll
        for (int i = 0; i < 2 * Q; i++)
          if (tid == i) co[i] = cf[i];
      }
      StepC<Q> c1;
      load_step_k<Q>(A.ctl + (size_t)(s + 1) * A.cs, c1, jpairs);
      scalarize<Q>(c1, jpairs);
      c1.g = DENSE ? reinterpret_cast<const double2*>(S.gtab) + (size_t)(s + 1) * S.N * S.N : nullptr;
      for (int e = tid; e < eend; e += nt) KB[e] = tm.template apply<true>(S, c1, XB, e);
      tm.tsync();
      for (int e = tid; e < eend; e += nt) {
        const double2 t = KB[e];
        double2 xb = XB[e];
        xb.x = fma(c.h, t.x, xb.x);
        xb.y = fma(c.h, t.y, xb.y);
        XB[e] = xb;
      }
      tm.tsync();
      continue;
    }
    // ImplMidpoint::evolveBWD (timestepper.cpp:631-694); the primal stage z = x + h/2 k (:640-652) was stored by the forward sweep
    {
      const double* zsrc = A.ztraj + ((size_t)s * A.nb + ic) * 2 * dim;
      for (int e = tid; e < eend; e += nt) Z[e] = reinterpret_cast<const double2*>(zsrc)[e];
    }
    int its;
    {
      const double2* K = tm.template solve<true, GM>(A, c, 0.5 * c.h, XB, Ya, Yb, &its);
      for (int e = tid; e < eend; e += nt) {
        const double2 k = K[e];
        KB[e] = make_double2(c.h * k.x, c.h * k.y);
      }
    }
    tm.tsync();
    for (int e = tid; e < eend; e += nt) {  // gradient coefficients (mastereq.hpp:553-604) and xbar += M^T kbar
      tm.at(e);
      const double2 kb = KB[e];
#pragma unroll
      for (int k = 0; k < Q; k++) {
        double2 Av, Bv;
        tm.st.ladder(S, tm.L, Z, k, 0, Av, Bv);
        cf[2 * k] += Bv.y * kb.x - Bv.x * kb.y;
        cf[2 * k + 1] += Av.x * kb.x + Av.y * kb.y;
      }
      double2 t;
      if constexpr (DENSE || !TM::kBatch) t = tm.st.template apply<true>(S, tm.L, KB, c, 0, kb);
      else t = tm.st.template apply_batched<true>(S, tm.L, KB, c, 0, kb);
      double2 xb = XB[e];
      xb.x += t.x;
      xb.y += t.y;
      XB[e] = xb;
    }
    tm.template sum<2 * Q>(cf);
    {
      double* co = A.coeff + ((size_t)ic * A.nsub + s) * 2 * Q;
#pragma unroll
      for (int i = 0; i < 2 * Q; i++)
        if (tid == i) co[i] = cf[i];
    }
    tm.tsync();
  }
  if (A.xbar0) {
    double* d0 = A.xbar0 + (size_t)ic * 2 * dim;
    for (int e = tid; e < eend; e += nt) {
      const double2 v = XB[e];
      d0[e] = v.x;
      d0[dim + e] = v.y;
    }
  }
}

template <int Q, bool LIND, bool DENSE = false>
__global__ void __launch_bounds__(BIG_BLOCK) k_apply_big(const DevSys S, const double* __restrict__ ctlrow, int transpose,
                                                         const double* __restrict__ xin, double* __restrict__ yout, int nb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BigTeam<Q, LIND, DENSE> tm;
  if (!tm.init(S, smem, nb)) return;
  const int dim = S.dim, ic = tm.ic, nt = tm.gnt, tid = tm.gtid, eend = tm.gend;
  double2* X = reinterpret_cast<double2*>(S.work) + (size_t)ic * BIG_NV * dim;
  const double* x0 = xin + (size_t)ic * 2 * dim;
  for (int e = tid; e < eend; e += nt) X[e] = make_double2(x0[e], x0[dim + e]);
  tm.tsync();
  StepC<Q> c;
  load_step_k<Q>(ctlrow, c, S.npairs > 0);
  scalarize<Q>(c, S.npairs > 0);
  c.g = DENSE ? reinterpret_cast<const double2*>(S.gtab) : nullptr;  // one-row table of the test hook
  double* yo = yout + (size_t)ic * 2 * dim;
  for (int e = tid; e < eend; e += nt) {
    const double2 y = transpose ? tm.template apply<true>(S, c, X, e) : tm.template apply<false>(S, c, X, e);
    yo[e] = y.x;
    yo[dim + e] = y.y;
  }
}

}  // namespace qd
