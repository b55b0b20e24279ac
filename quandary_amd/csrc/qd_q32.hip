// qd_q32.hip — fp32-mixed sweeps for all-qubit Lindblad systems (BASELINE config 5: 2^5 Lindblad, superoperator
// dimension 1024, 1024 initial conditions; also the 4-qubit open system).  gfx950 / CDNA4 only.
//
// What is fp32 and what is fp64 (QD_PRECISION_F32MIXED, include/quandary_amd.h):
//   fp32  the exchange vector in LDS (float2 per element: one ds_read_b64 per neighbour, half the LDS bytes and half
//         the LDS footprint of the fp64 kernels), the stencil arithmetic y = M x / M^T x, the linear-solver iterates
//         and right-hand sides, the stored trajectory (8 dim bytes per state instead of 16 dim);
//   fp64  the state x_n and the adjoint state xbar_n themselves (register accumulators: x += h k is an fp64 add of an
//         fp32 increment, so rounding does not accumulate over the time loop), objective / penalty sums, gradient
//         coefficients and everything downstream of them (k_reduce_coeff, k_grad, k_objective, k_seed).
//   The squared update norm of the Neumann iteration is only compared with a threshold and is reduced in fp32 (as in
//   the fp64 kernels).  fp32 iterates cannot reach the reference's abstol = 1e-10 (timestepper.cpp:536), so the
//   iteration additionally stops once the update is at the fp32 resolution of the right-hand side:
//   ||y_{m+1} - y_m|| <= max(abstol, 2^-22 ||b||).
//
// Element -> thread map (the slot layout of QubitSlotStencil, qd_device.h): it = tid | j << TB, EPT = 2^SB slots per
// thread, the slot number is the ket digits of oscillators 0 .. SB-1.  Digit signs, T1 validity and LDS offsets are
// thread invariants or compile-time constants; the ket neighbours of the slot oscillators are the thread's own slots.
//
// Reference semantics (paths relative to the reference repository): stencil include/mastereq.hpp:316-912 as
// instantiated for two-level systems (src/mastereq.cpp:2412-2893), IMR forward / adjoint src/timestepper.cpp:584-694,
// Neumann :697-727, time loops :96-253, weighted-J penalty :256-339, gradient coefficients include/mastereq.hpp:553-604.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "qd_device.h"

namespace qd {

constexpr float F32_SOLVER_TOL = 2.384185791015625e-07f;  // 2^-22
// waves per SIMD the coupled 2^5 fp32-mixed kernels are budgeted for (0 = one 512-thread workgroup per CU like their fp64 form; 4 = two:
// 128 registers, 77 spilt - c5j forward 25.5 -> 32.1 ms, gradient 57.5 -> 69.5 in one lease [r6]: measurement build only)
#ifndef QD_F32HJ_W
#define QD_F32HJ_W 0
#endif
// Measurement builds (profiles/q32_ab.sh): scheduling fence after every QD_Q32_FENCE-th slot of a thread (1 in the product)
#ifndef QD_Q32_FENCE
#define QD_Q32_FENCE 1
#endif
template <int EPT>
__device__ __forceinline__ void q32_fence(int j) {
  if ((j % QD_Q32_FENCE) == QD_Q32_FENCE - 1) slot_fence<EPT>();
}

// R = float: the fp32-mixed sweeps.  R = double: the same lean kernel structure in full fp64 (every value below is a
// double, the "accumulators" and the exchange vector coincide) - the throughput kernel of the 2^5 Lindblad system in
// QD_PRECISION_F64: 4 x fewer registers than the general slot kernel of qd_device.h, hence 2 workgroups per CU.
template <typename R> struct Vec2;
template <> struct Vec2<float> { typedef float2 type; };
template <> struct Vec2<double> { typedef double2 type; };
__device__ __forceinline__ float rfma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double rfma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ double uniform(double v) { return to_scalar(v); }

template <int Q, int SB, typename R = float, bool HJ = false>
struct Q32 {
  typedef typename Vec2<R>::type f2;
  static constexpr int EPT = 1 << SB, TB = 2 * Q - SB, NT = 1 << TB, NW = NT / 64, DIM = 1 << (2 * Q);
  static constexpr bool ONEWAVE = NT == 64;
  static constexpr int BPC = sizeof(R) == 4 ? 4 : 2;  // workgroups per CU the register budget is set for (256-thread groups)
  // (the 512-thread fp64 variant of the 2^5 system runs one group per CU: two waves per SIMD, the whole register file)
  static constexpr int MINW = (Q == 5 && SB == 1) ? 2 : (BPC * NT / 256 > 0 ? BPC * NT / 256 : 1);  // 4 workgroups of 256 threads per CU (128 VGPRs); one-wave groups: 1 wave/SIMD budget
  static constexpr unsigned EB = sizeof(f2), ESH = sizeof(R) == 4 ? 3 : 4;  // bytes per element, log2
  static constexpr unsigned SLOT_BYTES = EB << TB;  // LDS distance of consecutive slots
  static_assert(NT >= 64 && NT <= 1024, "block size");

  __device__ static constexpr int brabit(int k) { return Q - 1 - k; }
  __device__ static constexpr int ketbit(int k) { return 2 * Q - 1 - k; }
  __device__ static constexpr int slotbit(int j, int k) { return (j >> (SB - 1 - k)) & 1; }  // ket digit of oscillator k < SB in slot j
  __device__ static constexpr int slotflip(int j, int k) { return j ^ (1 << (SB - 1 - k)); }

  // dipole-dipole coupling (mastereq.hpp:632-741 for two levels; one element per thread only: SB == 0): byte offsets of the element with
  // the bra / ket digits of both oscillators of pair (k, l) flipped; coefficients of the current sub-step (prep): J_kl cos / sin
  // (eta_kl t) where the two digits differ (else 0), the sine with the sign of the l digit (QubitStencil::apply of qd_device.h:
  // h += J (sin A + cos (B.y, -B.x)), A = +-x_bra +-x_ket, B = x_bra - x_ket)
  static constexpr int NP = Q * (Q - 1) / 2;
  // ([r6] R = float: the same terms on the unpacked fp32 stencil - the fp32-mixed sweeps of the coupled systems)
  static constexpr bool JOK = HJ && SB == 0;
  static_assert(!HJ || SB <= 1, "coupled systems: one or two elements per thread");
  unsigned ajb[JOK ? NP : 1], ajk[JOK ? NP : 1];
  R pjs[JOK ? NP : 1], pjc[JOK ? NP : 1], qjs[JOK ? NP : 1], qjc[JOK ? NP : 1];
  // The 2^5 system (two elements per thread, SB == 1: the ket digit of oscillator 0 is the slot) has no room for four coefficients per
  // pair beside its stencil: there the pair coefficients stay wave-uniform (J cos, J sin), the digit test moves into the ADDRESS - a
  // neighbour whose pair of digits is equal is read from an element that holds zero (Team32 keeps one behind the exchange vectors at the
  // same distance from either of them, for either slot) - and the sign of the sine terms, a property of the digit of oscillator l alone,
  // is applied once per l to the sum over k < l.
  static constexpr bool JS1 = HJ && SB == 1;
  static constexpr bool JANY = JOK || JS1;
  static constexpr unsigned ZOFF = 2u * (unsigned)DIM * (unsigned)sizeof(f2);  // the zero element, relative to the vector being read
  unsigned jab[JS1 ? NP : 1], jak[JS1 ? NP : 1][JS1 ? EPT : 1];
  unsigned alm[JS1 ? Q : 1][2];          // T1 neighbour, forward / transposed, or the zero element where the digit condition fails
  R g1u[JS1 ? Q : 1];                    // gamma_1 coefficient of the T1 neighbour (wave-uniform)
  R sgb[JS1 ? Q : 1], sgk[JS1 ? Q : 1];  // -1 where the bra / ket digit of oscillator l is 1, else +1
  R jc[JS1 ? NP : 1], js[JS1 ? NP : 1];  // J cos / J sin (eta_kl t) of the current sub-step (wave-uniform)
  R Jc[JANY ? NP : 1];                   // J_kl (wave-uniform)
  __device__ static constexpr int pairof(int k, int l) { return k * Q - k * (k + 1) / 2 + (l - k - 1); }
  R dw[EPT], dd[EPT];            // Delta = h(I) - h(I'), d = L2 + L1diag (mastereq.hpp:316-433)
  unsigned ab[Q], ak[Q], al[Q];  // byte offsets (slot 0) of the bra / ket (k >= SB) / T1 neighbour of oscillator k
  R l1f[Q], l1t[Q];              // thread part of the T1 off-diagonal coefficient, forward / transposed
  R qb[Q], qk[Q];                // q_k with the sign of the bra / ket (k >= SB) digit of this thread [per step]
  R p[Q], q[Q];                  // controls of the current sub-step (wave-uniform)

  __device__ __forceinline__ void init(const DevSys& S, bool = false) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int it = (int)(tid | ((unsigned)j << TB));
      double hd = 0.0, hdp = 0.0, d = 0.0;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int a = (it >> brabit(k)) & 1, ap = (it >> ketbit(k)) & 1;
        hd += S.detune[k] * a;  // the self-Kerr term a(a-1) vanishes for two levels
        hdp += S.detune[k] * ap;
        d += S.g2[k] * (a * ap - 0.5 * (a + ap)) - S.g1[k] / 2.0 * (a + ap);
#pragma unroll
        for (int l = k + 1; l < Q; l++) {
          const int b = (it >> brabit(l)) & 1, bp = (it >> ketbit(l)) & 1;
          hd -= S.xikl[pair] * a * b;
          hdp -= S.xikl[pair] * ap * bp;
          pair++;
        }
      }
      dw[j] = (R)(hd - hdp);  // differences formed in fp64, rounded once
      dd[j] = (R)d;
    }
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const unsigned bb = 1u << brabit(k), kb = 1u << ketbit(k);
      ab[k] = (tid ^ bb) << ESH;
      ak[k] = k >= SB ? (tid ^ kb) << ESH : 0u;
      al[k] = k >= SB ? (tid ^ bb ^ kb) << ESH : ab[k];
      const bool bra0 = (tid & bb) == 0;
      if (k >= SB) {
        const bool ket0 = (tid & kb) == 0;
        l1f[k] = (bra0 && ket0) ? (R)S.g1off[k] : (R)0;
        l1t[k] = (!bra0 && !ket0) ? (R)S.g1off[k] : (R)0;
      } else {  // the ket digit is a slot bit: only the bra condition is a thread property
        l1f[k] = bra0 ? (R)S.g1off[k] : (R)0;
        l1t[k] = !bra0 ? (R)S.g1off[k] : (R)0;
      }
      qb[k] = qk[k] = p[k] = q[k] = (R)0;
    }
    if constexpr (JOK) {
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++)
#pragma unroll
        for (int l = k + 1; l < Q; l++, pair++) {
          const unsigned bm = (1u << brabit(k)) | (1u << brabit(l)), km = (1u << ketbit(k)) | (1u << ketbit(l));
          ajb[pair] = (tid ^ bm) << ESH;
          ajk[pair] = (tid ^ km) << ESH;
          Jc[pair] = uniform((R)S.J[pair]);
          pjs[pair] = pjc[pair] = qjs[pair] = qjc[pair] = (R)0;
        }
    }
    if constexpr (JS1) {
#pragma unroll
      for (int l = 0; l < Q; l++) {
        const unsigned bb = 1u << brabit(l), kb = 1u << ketbit(l);
        const bool bra0 = (tid & bb) == 0, ket0 = l < SB || (tid & kb) == 0, ket1 = l < SB || (tid & kb) != 0;
        sgb[l] = bra0 ? (R)1 : (R)-1;
        sgk[l] = (l >= SB && (tid & kb)) ? (R)-1 : (R)1;
        // (k < SB: the ket digit is the slot - the slot condition is a compile-time one in load(), the neighbour sits in the other slot)
        alm[l][0] = (bra0 && ket0) ? al[l] : ZOFF;
        alm[l][1] = (!bra0 && ket1) ? al[l] : ZOFF;
        g1u[l] = uniform((R)S.g1off[l]);
      }
#pragma unroll
      for (int k = 0; k < Q; k++)
#pragma unroll
        for (int l = k + 1; l < Q; l++) {
          const int pr = pairof(k, l);
          const unsigned bm = (1u << brabit(k)) | (1u << brabit(l));
          const unsigned a = (tid >> brabit(k)) & 1, b = (tid >> brabit(l)) & 1, bp = (tid >> ketbit(l)) & 1;
          jab[pr] = a != b ? (tid ^ bm) << ESH : ZOFF;
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            if (k >= SB) {
              const unsigned ap = (tid >> ketbit(k)) & 1, km = (1u << ketbit(k)) | (1u << ketbit(l));
              jak[pr][j] = ap != bp ? (tid ^ km) << ESH : ZOFF;
            } else {  // the ket digit of oscillator k is the slot: the neighbour sits in the other slot (read with that slot's offset)
              jak[pr][j] = (unsigned)slotbit(j, k) != bp ? (tid ^ (1u << ketbit(l))) << ESH : ZOFF;
            }
          }
          Jc[pr] = uniform((R)S.J[pr]);
          jc[pr] = js[pr] = (R)0;
        }
    }
  }

  // once per sub-step: controls as wave-uniform floats, digit signs folded into q
  __device__ __forceinline__ void prep(const StepC<Q>& c) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      p[k] = uniform((R)c.p[k]);
      q[k] = uniform((R)c.q[k]);
      if constexpr (!JS1) {
        qb[k] = ((tid >> brabit(k)) & 1) ? -q[k] : q[k];
        if (k >= SB) qk[k] = ((tid >> ketbit(k)) & 1) ? -q[k] : q[k];
      }
    }
    if constexpr (JOK) {
      int pr = 0;
#pragma unroll
      for (int k = 0; k < Q; k++)
#pragma unroll
        for (int l = k + 1; l < Q; l++, pr++) {
          const R jc = Jc[pr] * uniform((R)c.cs[pr]), js = Jc[pr] * uniform((R)c.sn[pr]);
          const unsigned a = (tid >> brabit(k)) & 1, b = (tid >> brabit(l)) & 1, ap = (tid >> ketbit(k)) & 1, bp = (tid >> ketbit(l)) & 1;
          pjc[pr] = a != b ? jc : (R)0;
          pjs[pr] = a != b ? (b ? -js : js) : (R)0;
          qjc[pr] = ap != bp ? jc : (R)0;
          qjs[pr] = ap != bp ? (bp ? -js : js) : (R)0;
        }
    }
    if constexpr (JS1) {
#pragma unroll
      for (int pr = 0; pr < NP; pr++) {
        jc[pr] = uniform(Jc[pr] * (R)c.cs[pr]);
        js[pr] = uniform(Jc[pr] * (R)c.sn[pr]);
      }
    }
  }

  __device__ __forceinline__ static f2 at(const f2* __restrict__ sx, unsigned byteoff, int slot) {
    return *reinterpret_cast<const f2*>(reinterpret_cast<const char*>(sx) + byteoff + (unsigned)slot * SLOT_BYTES);
  }

  // The LDS neighbours of slot j, fetched into one object apart from the arithmetic on them [r5]: with the reads written inside the loop
  // over the oscillators the compiler interleaved them with the fp64 chains; all of them first is 12 % on the 2^4 sweeps, whose four
  // waves sit alone on their SIMDs (forward 2.26 -> 1.99 ms, gradient 5.44 -> 4.70; 2^5 unchanged).  Issuing the next pass's first slot
  // together with the partial sums of the stopping test (one LDS round trip less in the dependent chain) was measured on top of it
  // and lost: 2.10 ms.
  struct Nb {
    f2 xb[Q], xk[Q], xl[Q];
    f2 xjb[JOK ? NP : 1], xjk[JOK ? NP : 1];
  };
  template <bool TRANS>
  __device__ __forceinline__ void load(const f2* __restrict__ sx, int j, Nb& n) const {
#pragma unroll
    for (int k = 0; k < Q; k++) {
      n.xb[k] = at(sx, ab[k], j);
      if (k >= SB) n.xk[k] = at(sx, ak[k], j);
      const bool slot_ok = k >= SB || (TRANS ? slotbit(j, k) == 1 : slotbit(j, k) == 0);
      if (slot_ok) n.xl[k] = at(sx, JS1 ? alm[JS1 ? k : 0][TRANS ? 1 : 0] : al[k], k < SB ? slotflip(j, k) : j);
    }
    if constexpr (JOK) {
#pragma unroll
      for (int pr = 0; pr < NP; pr++) {
        n.xjb[pr] = at(sx, ajb[pr], j);
        n.xjk[pr] = at(sx, ajk[pr], j);
      }
    }
  }
  // y = M x (TRANS = false) or M^T x at slot j; see QubitSlotStencil::apply for the derivation
  template <bool TRANS>
  __device__ __forceinline__ f2 apply(const f2* __restrict__ sx, int j, const f2 (&xall)[EPT]) const {
    Nb n;
    load<TRANS>(sx, j, n);
    return apply_nb<TRANS>(j, xall, n, sx);
  }
  template <bool TRANS>
  __device__ __forceinline__ f2 apply_nb(int j, const f2 (&xall)[EPT], const Nb& n, const f2* __restrict__ sx) const {
    const f2 xs = xall[j];
    R hr = dw[j] * xs.y, hi = -dw[j] * xs.x, gr = 0, gi = 0;
    R l1r = 0, l1i = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const f2 xb = n.xb[k];
      f2 xk;
      R sqk;
      if (k < SB) {  // ket neighbour = own slot with the slot bit flipped; the sign of the slot bit is a constant
        xk = xall[slotflip(j, k)];
        sqk = slotbit(j, k) ? -q[k] : q[k];
      } else {
        xk = n.xk[k];
        sqk = qk[k];
      }
      R& ar = (k & 1) ? gr : hr;
      R& ai = (k & 1) ? gi : hi;
      if constexpr (JS1) {  // digit signs from the +-1 pairs the coupling terms keep anyway: q_k (s_b x_b + s_k x_k)
        const R sk = k < SB ? (slotbit(j, k) ? (R)-1 : (R)1) : sgk[k];
        const R gx = rfma(sgb[k], xb.x, sk * xk.x), gy = rfma(sgb[k], xb.y, sk * xk.y);
        ar = rfma(q[k], gx, ar);
        ai = rfma(q[k], gy, ai);
      } else {
        ar = rfma(qb[k], xb.x, ar);
        ai = rfma(qb[k], xb.y, ai);
        ar = rfma(sqk, xk.x, ar);
        ai = rfma(sqk, xk.y, ai);
      }
      ar = rfma(p[k], xb.y, ar);
      ai = rfma(-p[k], xb.x, ai);
      ar = rfma(-p[k], xk.y, ar);
      ai = rfma(p[k], xk.x, ai);
      // T1 off-diagonal: forward needs both digits 0 (the neighbour has both set), transposed both 1
      const bool slot_ok = k >= SB || (TRANS ? slotbit(j, k) == 1 : slotbit(j, k) == 0);
      if (slot_ok) {
        const f2 xl = n.xl[k];
        const R l1 = JS1 ? g1u[JS1 ? k : 0] : TRANS ? l1t[k] : l1f[k];
        l1r = rfma(l1, xl.x, l1r);
        l1i = rfma(l1, xl.y, l1i);
      }
    }
    hr += gr;
    hi += gi;
    if constexpr (JOK) {
#pragma unroll
      for (int pr = 0; pr < NP; pr++) {
        const f2 xj = n.xjb[pr], xq = n.xjk[pr];
        hr = rfma(pjs[pr], xj.x, rfma(pjc[pr], xj.y, rfma(qjs[pr], xq.x, rfma(-qjc[pr], xq.y, hr))));
        hi = rfma(pjs[pr], xj.y, rfma(-pjc[pr], xj.x, rfma(qjs[pr], xq.y, rfma(qjc[pr], xq.x, hi))));
      }
    }
    if constexpr (JS1) {
      // Neighbours read here, one oscillator's pairs (k < l) at a time, one group ahead of the arithmetic: 20 more values in flight do not
      // fit.  A fence alone orders the LDS reads but lets the arithmetic sink below the next group's reads (all neighbours live at once:
      // 20 registers per pair, 132 spilt, slower than the general kernel); pinning the sums at the fence completes a group before the
      // group after the next is read.
      f2 cj[Q - 1], cq[Q - 1];
      cj[0] = at(sx, jab[pairof(0, 1)], j);
      cq[0] = at(sx, jak[pairof(0, 1)][j], 0 < SB ? slotflip(j, 0) : j);
      asm volatile("" : "+v"(hr), "+v"(hi), "+v"(l1r), "+v"(l1i)::"memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int l = 1; l < Q; l++) {
        f2 nj[Q - 1], nq[Q - 1];
        if (l + 1 < Q) {
#pragma unroll
          for (int k = 0; k < l + 1; k++) {
            const int pn = pairof(k, l + 1 < Q ? l + 1 : l);
            nj[k] = at(sx, jab[pn], j);
            nq[k] = at(sx, jak[pn][j], k < SB ? slotflip(j, k) : j);
          }
        }
        R sbx = 0, sby = 0, skx = 0, sky = 0;
#pragma unroll
        for (int k = 0; k < l; k++) {
          const int pr = pairof(k, l);
          const f2 xj = cj[k], xq = cq[k];
          sbx = rfma(js[pr], xj.x, sbx);
          sby = rfma(js[pr], xj.y, sby);
          skx = rfma(js[pr], xq.x, skx);
          sky = rfma(js[pr], xq.y, sky);
          hr = rfma(jc[pr], xj.y, rfma(-jc[pr], xq.y, hr));
          hi = rfma(-jc[pr], xj.x, rfma(jc[pr], xq.x, hi));
        }
        hr = rfma(sgb[l], sbx, rfma(sgk[l], skx, hr));
        hi = rfma(sgb[l], sby, rfma(sgk[l], sky, hi));
        asm volatile("" : "+v"(hr), "+v"(hi)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (l + 1 < Q) {
#pragma unroll
          for (int k = 0; k < l + 1; k++) {
            cj[k] = nj[k];
            cq[k] = nq[k];
          }
        }
      }
    }
    f2 y;
    y.x = rfma(dd[j], xs.x, TRANS ? -hr : hr) + l1r;
    y.y = rfma(dd[j], xs.y, TRANS ? -hi : hi) + l1i;
    return y;
  }

  // Gradient contraction of all the thread's slots (QubitStencil::ladder: A = s_b x_b + s_k x_k, B = x_b - x_k with the digit signs
  // s = -1 for digit 1; include/mastereq.hpp:553-604 for two levels), summed over the slots BEFORE the signs are applied:
  //   cf[2k]     += sum_j  B_j.y w_j.x - B_j.x w_j.y
  //   cf[2k + 1] += s_b sum_j (x_b,j . w_j) + s_k sum_j (x_k,j . w_j)
  // The bra digit is a thread invariant for every oscillator, the ket digit for k >= SB and a compile-time constant of the slot below;
  // neighbours come from the thread-invariant byte offsets of apply() (one address per oscillator and side, the slot is an immediate
  // offset), the ket neighbours of the slot oscillators from the thread's own registers.  8 fused operations per (slot, oscillator) and
  // no select - the per-pair form recomputed two addresses, two digit tests and four sign selects for every pair (32 instructions per
  // pair, 640 per step of the 2^5 system).  Products in R, sums in fp64 (fp32-mixed: QD_PRECISION_F32MIXED, include/quandary_amd.h).
  __device__ __forceinline__ void ladder_all(const f2* __restrict__ sx, const f2 (&z)[EPT], const f2 (&w)[EPT], double (&cf)[2 * Q]) const {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      double c0 = 0.0, tb = 0.0, tk = 0.0;
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const f2 xb = at(sx, ab[k], j);
        const f2 xk = k < SB ? z[slotflip(j, k)] : at(sx, ak[k], j);
        const bool kneg = k < SB && slotbit(j, k);  // (compile-time)
        if constexpr (sizeof(R) == 8) {
          c0 = fma(xb.y, w[j].x, c0);
          c0 = fma(-xb.x, w[j].y, c0);
          c0 = fma(-xk.y, w[j].x, c0);
          c0 = fma(xk.x, w[j].y, c0);
          tb = fma(xb.x, w[j].x, tb);
          tb = fma(xb.y, w[j].y, tb);
          tk = fma(kneg ? -xk.x : xk.x, w[j].x, tk);
          tk = fma(kneg ? -xk.y : xk.y, w[j].y, tk);
        } else {
          c0 += (double)rfma(xb.y - xk.y, w[j].x, -(xb.x - xk.x) * w[j].y);
          tb += (double)rfma(xb.x, w[j].x, xb.y * w[j].y);
          const R t = rfma(xk.x, w[j].x, xk.y * w[j].y);
          tk += (double)(kneg ? -t : t);
        }
      }
      const bool a = (tid >> brabit(k)) & 1, ap = k >= SB && ((tid >> ketbit(k)) & 1);
      cf[2 * k] += c0;
      cf[2 * k + 1] += (a ? -tb : tb) + (ap ? -tk : tk);
    }
  }
};

#ifndef QD_F32_UNPACKED
// fp32-mixed on packed arithmetic [r5].  An element is a (re, im) pair in one 64-bit register pair and every stencil term is ONE
// v_pk_fma_f32 on it, with broadcast halves of coefficient pairs (op_sel / op_sel_hi) and whole-pair negation (neg_lo + neg_hi) as
// operand modifiers.  Coefficients are kept two to a register pair - (q_bra, q_ket) lists, T1 coefficients of two oscillators, (Delta, d)
// of a slot - and the wave-uniform (p_k, q_k) pairs stay in scalar registers, so the pairs cost no more registers than the scalar layout
// (round 2's compiler-vectorised attempt spilt 102 registers on half-empty pairs).  Products are those of the scalar code; the sums are
// reassociated (see apply).  What it buys, measured (profiles/r5_rate_probe.json, profiles/r5_q32_ab.txt): on this chip v_fma_f32 issues
// in 2.56 cycles per wave and SIMD at the reported clock (122.9 TFLOP/s, NOT the fp64 rate: v_fma_f64 4.27 cycles, 73.7), v_pk_fma_f32
// in 4.41 (142.6 TFLOP/s) - packing halves the instruction count of the solver pass (314 -> 195 vector instructions, 128 of them packed)
// and shortens its vector-pipe time by 9 %; the 2^5 forward sweep is unchanged within the lease noise (8.88 / 9.16 against 8.94 / 9.00 ms),
// the gradient evaluation 1.5 % faster (21.3 / 21.6 against 21.7 / 21.9 ms).  -DQD_F32_UNPACKED builds the scalar form for that A/B.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 pkv(const float2 a) { return (pk2){a.x, a.y}; }
__device__ __forceinline__ float2 pkf(const pk2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ pk2 pk_fma(const pk2 a, const pk2 b, const pk2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ pk2 bc(const float a) { return (pk2){a, a}; }
__device__ __forceinline__ pk2 swp(const pk2 a) { return (pk2){a.y, a.x}; }

template <int Q, int SB>
struct Q32<Q, SB, float, false> {
  typedef float R;
  typedef float2 f2;
  static constexpr int EPT = 1 << SB, TB = 2 * Q - SB, NT = 1 << TB, NW = NT / 64, DIM = 1 << (2 * Q);
  static constexpr bool ONEWAVE = NT == 64;
  static constexpr int BPC = 4;
  static constexpr bool JS1 = false;  // (no coupled systems in fp32-mixed)
  static constexpr int MINW = (Q == 5 && SB == 1) ? 2 : (BPC * NT / 256 > 0 ? BPC * NT / 256 : 1);
  static constexpr unsigned EB = sizeof(f2), ESH = 3;
  static constexpr unsigned SLOT_BYTES = EB << TB;
  static_assert(NT >= 64 && NT <= 1024, "block size");

  __device__ static constexpr int brabit(int k) { return Q - 1 - k; }
  __device__ static constexpr int ketbit(int k) { return 2 * Q - 1 - k; }
  __device__ static constexpr int slotbit(int j, int k) { return (j >> (SB - 1 - k)) & 1; }
  __device__ static constexpr int slotflip(int j, int k) { return j ^ (1 << (SB - 1 - k)); }

  // Per-thread coefficients, two to a register pair (component c of a list lives in pair c / 2, half c % 2 - compile-time indices):
  //   tq: q_k with the sign of the bra digit (c = k), then with the sign of the ket digit for k >= SB (c = Q + k - SB)   [per step]
  //   tl: thread part of the T1 off-diagonal coefficient of oscillator k, forward OR transposed (init's argument)
  static constexpr int NTQ = 2 * Q - SB;
  pk2 cd[EPT];                   // (Delta, d) of slot j
  unsigned ab[Q], ak[Q], al[Q];  // byte offsets (slot 0) of the bra / ket (k >= SB) / T1 neighbour of oscillator k
  pk2 tl[(Q + 1) / 2];
  pk2 tq[(NTQ + 1) / 2];
  pk2 pq[Q];                     // (p_k, q_k) of the current sub-step (wave-uniform: scalar registers)
  // A broadcast / swapped / negated operand is free only while instruction selection sees the pair it is built from in the SAME basic
  // block (op_sel, neg_lo, neg_hi); loop-invariant code motion would otherwise materialise every (c, c) and (p, -p) pair in registers
  // outside the solver loop - twice the coefficient registers.  here() pins a pair to its point of use (no instruction).
  __device__ __forceinline__ static pk2 here(pk2 v) {
    asm volatile("" : "+v"(v));
    return v;
  }
  // (a v_readfirstlane that survives: the builtin is folded away on a value the compiler already knows to be uniform, and the fp32
  // conversion of a uniform double still lives in a vector register)
  __device__ __forceinline__ static float to_sgpr(float v) {
    asm volatile("" : "+v"(v));  // (opaque: the compiler no longer knows the value to be uniform and keeps the instruction)
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
  }
  __device__ __forceinline__ static pk2 here_s(pk2 v) {
    asm volatile("" : "+s"(v));
    return v;
  }

  __device__ __forceinline__ void init(const DevSys& S, bool trans = false) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int it = (int)(tid | ((unsigned)j << TB));
      double hd = 0.0, hdp = 0.0, d = 0.0;
      int pair = 0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int a = (it >> brabit(k)) & 1, ap = (it >> ketbit(k)) & 1;
        hd += S.detune[k] * a;
        hdp += S.detune[k] * ap;
        d += S.g2[k] * (a * ap - 0.5 * (a + ap)) - S.g1[k] / 2.0 * (a + ap);
#pragma unroll
        for (int l = k + 1; l < Q; l++) {
          const int b = (it >> brabit(l)) & 1, bp = (it >> ketbit(l)) & 1;
          hd -= S.xikl[pair] * a * b;
          hdp -= S.xikl[pair] * ap * bp;
          pair++;
        }
      }
      cd[j] = (pk2){(float)(hd - hdp), (float)d};
    }
#pragma unroll
    for (int i = 0; i < (Q + 1) / 2; i++) tl[i] = (pk2){0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (NTQ + 1) / 2; i++) tq[i] = (pk2){0.f, 0.f};
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const unsigned bb = 1u << brabit(k), kb = 1u << ketbit(k);
      ab[k] = (tid ^ bb) << ESH;
      ak[k] = k >= SB ? (tid ^ kb) << ESH : 0u;
      al[k] = k >= SB ? (tid ^ bb ^ kb) << ESH : ab[k];
      // forward: both digits 0 (the neighbour has both set); transposed: both 1.  The ket digit of a slot oscillator is a slot bit.
      const bool bra = ((tid & bb) != 0) == trans;
      const bool ket = k < SB || (((tid & kb) != 0) == trans);
      tl[k / 2][k % 2] = (bra && ket) ? (float)S.g1off[k] : 0.f;
      pq[k] = (pk2){0.f, 0.f};
    }
  }

  __device__ __forceinline__ void prep(const StepC<Q>& c) {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const float pu = to_sgpr((float)c.p[k]), qu = to_sgpr((float)c.q[k]);
      pq[k] = (pk2){pu, qu};
      tq[k / 2][k % 2] = ((tid >> brabit(k)) & 1) ? -qu : qu;
      if (k >= SB) tq[(Q + k - SB) / 2][(Q + k - SB) % 2] = ((tid >> ketbit(k)) & 1) ? -qu : qu;
    }
  }

  __device__ __forceinline__ static pk2 at(const f2* __restrict__ sx, unsigned byteoff, int slot) {
    return *reinterpret_cast<const pk2*>(reinterpret_cast<const char*>(sx) + byteoff + (unsigned)slot * SLOT_BYTES);
  }

  // y = M x (TRANS = false) or M^T x at slot j: the scalar template above, two components per instruction.  TRANS must be the
  // argument init() was called with (the T1 coefficients).
  template <bool TRANS>
  __device__ __forceinline__ f2 apply(const f2* __restrict__ sx, int j, const f2 (&xall)[EPT]) {
    const pk2 xs = pkv(xall[j]);
    // (the pinned value replaces the member: no copy is kept for the next slot)
    const pk2 cdj = cd[j] = here(cd[j]);
#pragma unroll
    for (int i = 0; i < (NTQ + 1) / 2; i++) tq[i] = here(tq[i]);
#pragma unroll
    for (int i = 0; i < (Q + 1) / 2; i++) tl[i] = here(tl[i]);
    // Every p-term and the Hamiltonian diagonal have the form s J(v), J(v) = (v.y, -v.x): they are accumulated as U = sum s v with plain
    // broadcast coefficients and rotated once at the end, A = V + J(U) - a per-half negation is the one operand form instruction
    // selection does not fold (it rebuilds the pair with v_xor + v_mov), whole-vector negation and broadcasts it does.
    // (all LDS neighbours of the slot first, then the arithmetic: see Nb of the scalar template; here 2^4 forward 1.92 -> 1.84 ms,
    //  gradient 4.84 -> 4.57, 2^5 forward 8.6 -> 8.4)
    pk2 nxb[Q], nxk[Q], nxl[Q];
#pragma unroll
    for (int k = 0; k < Q; k++) {
      nxb[k] = at(sx, ab[k], j);
      if (k >= SB) nxk[k] = at(sx, ak[k], j);
      const bool slot_ok = k >= SB || (TRANS ? slotbit(j, k) == 1 : slotbit(j, k) == 0);
      if (slot_ok) nxl[k] = at(sx, al[k], k < SB ? slotflip(j, k) : j);
    }
    pk2 U = bc(cdj.x) * xs, V = {0.f, 0.f}, l1 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < Q; k++) {
      const pk2 pqk = pq[k] = here_s(pq[k]);
      const pk2 xb = nxb[k];
      U = pk_fma(bc(pqk.x), xb, U);
      V = pk_fma(bc(tq[k / 2][k % 2]), xb, V);
      pk2 xk;
      if (k < SB) {  // ket neighbour = own slot with the slot bit flipped; the sign of the slot bit is a constant
        xk = pkv(xall[slotflip(j, k)]);
        V = pk_fma(bc(pqk.y), slotbit(j, k) ? -xk : xk, V);
      } else {
        xk = nxk[k];
        V = pk_fma(bc(tq[(Q + k - SB) / 2][(Q + k - SB) % 2]), xk, V);
      }
      U = pk_fma(bc(pqk.x), -xk, U);
      // T1 off-diagonal: forward needs both digits 0 (the neighbour has both set), transposed both 1
      const bool slot_ok = k >= SB || (TRANS ? slotbit(j, k) == 1 : slotbit(j, k) == 0);
      if (slot_ok) {
        l1 = pk_fma(bc(tl[k / 2][k % 2]), nxl[k], l1);
      }
    }
    const pk2 A = {V.x + U.y, V.y - U.x};
    return pkf(pk_fma(bc(cdj.y), xs, TRANS ? -A : A) + l1);
  }

  // gradient contraction (see the scalar template): products and the sums over a thread's slots in packed fp32, everything after in fp64
  //   c0 = sum_j (x_b - x_k)_j.y w_j.x - (x_b - x_k)_j.x w_j.y,  tb = sum_j x_b,j . w_j,  tk = sum_j (+-) x_k,j . w_j
  __device__ __forceinline__ void ladder_all(const f2* __restrict__ sx, const f2 (&z)[EPT], const f2 (&w)[EPT], double (&cf)[2 * Q]) const {
    const unsigned tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < Q; k++) {
      pk2 c0 = {0.f, 0.f}, tb = {0.f, 0.f}, tk = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const pk2 wj = pkv(w[j]);
        const pk2 xb = at(sx, ab[k], j);
        const pk2 xk = k < SB ? pkv(z[slotflip(j, k)]) : at(sx, ak[k], j);
        const bool kneg = k < SB && slotbit(j, k);  // (compile-time)
        c0 = pk_fma(swp(xb - xk), (pk2){wj.x, -wj.y}, c0);
        tb = pk_fma(xb, wj, tb);
        tk = pk_fma(kneg ? -xk : xk, wj, tk);
      }
      const bool a = (tid >> brabit(k)) & 1, ap = k >= SB && ((tid >> ketbit(k)) & 1);
      const double tbd = (double)tb.x + (double)tb.y, tkd = (double)tk.x + (double)tk.y;
      cf[2 * k] += (double)c0.x + (double)c0.y;
      cf[2 * k + 1] += (a ? -tbd : tbd) + (ap ? -tkd : tkd);
    }
  }
};
#endif  // QD_F32_UNPACKED

// per-workgroup state of the sweeps: LDS exchange buffers, reduction scratch, the Neumann solver (GM: + in-kernel GMRES)
template <int Q, int SB, typename R, bool GM = false, bool HJ = false>
struct Team32 {
  typedef Q32<Q, SB, R, HJ> ST;
  typedef typename ST::f2 f2;
  static constexpr int EPT = ST::EPT, NW = ST::NW, DIM = ST::DIM;
  static constexpr bool ONEWAVE = ST::ONEWAVE;
  static constexpr bool F32 = sizeof(R) == 4;
  ST st;
  f2* buf;       // two exchange vectors of DIM elements
  double* red;   // two reduction slots of NRED * NW doubles
  double2* acc;  // the fp64 state accumulators, parked here while a linear solve runs (EPT > 1; only ever touched by the owning thread)
  double* ksc;   // GMRES: wave-uniform scalars of the Hessenberg problem
  int cur, redslot;
  static constexpr bool PARK = EPT > 1;

  __device__ __forceinline__ void init(const DevSys& S, unsigned char* smem, bool trans = false) {
    buf = reinterpret_cast<f2*>(smem);
    red = reinterpret_cast<double*>(smem + 2 * sizeof(f2) * DIM + ZPAD);
    acc = reinterpret_cast<double2*>(smem + acc_off());
    ksc = reinterpret_cast<double*>(smem + acc_off() + (PARK ? sizeof(double2) * DIM : 0));
    cur = 0;
    redslot = 0;
    if constexpr (ST::JS1) {  // the zero elements (see Q32): at 2 D and 3 D for slot 0, half a vector further for slot 1 (D = bytes of a vector)
      if (threadIdx.x < 4) *reinterpret_cast<f2*>(smem + (4 + threadIdx.x) * (sizeof(f2) * DIM / 2)) = f2{0, 0};
    }
    st.init(S, trans);  // (the packed fp32 stencil keeps the T1 coefficients of one direction only)
  }
  // coupled 2^5 kernels: zero elements at 2 D, 2.5 D, 3 D, 3.5 D behind the two vectors; the reduction scratch sits between the first two,
  // the parked accumulators behind the last
  static constexpr size_t ZPAD = ST::JS1 ? 16 : 0;
  static constexpr size_t acc_off() {
    return ST::JS1 ? 7 * (sizeof(f2) * DIM / 2) + 16 : 2 * sizeof(f2) * DIM + 2 * sizeof(double) * NRED * NW;
  }
  static_assert(!ST::JS1 || (ZPAD + 2 * sizeof(double) * NRED * NW <= sizeof(f2) * DIM / 2 && EPT == 2), "layout of the coupled 2^5 kernels");
  static size_t lds_bytes() {
    return acc_off() + (PARK ? sizeof(double2) * DIM : 0) + (GM ? sizeof(double) * gmres_nsc(GMRES_MR_G) : 0);
  }
  __device__ __forceinline__ int elem(int j) const { return (int)(threadIdx.x | ((unsigned)j << ST::TB)); }
  __device__ __forceinline__ const f2* vec() const { return buf + cur * DIM; }
  __device__ __forceinline__ void park(const double2 (&v)[EPT]) const {
    if (PARK) {
#pragma unroll
      for (int j = 0; j < EPT; j++) acc[elem(j)] = v[j];
    }
  }
  __device__ __forceinline__ void unpark(double2 (&v)[EPT]) const {
    if (PARK) {
#pragma unroll
      for (int j = 0; j < EPT; j++) v[j] = acc[elem(j)];
    }
  }

  __device__ __forceinline__ void publish(const f2 (&x)[EPT]) {
    f2* dst = buf + (cur ^ 1) * DIM;
#pragma unroll
    for (int j = 0; j < EPT; j++) dst[elem(j)] = x[j];
    cur ^= 1;
    team_sync<ONEWAVE>();
  }

  template <bool TRANS>
  __device__ __forceinline__ void apply_all(const f2 (&x)[EPT], f2 (&y)[EPT]) {
    const f2* sx = vec();
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      y[j] = st.template apply<TRANS>(sx, j, x);
      q32_fence<EPT>(j);  // keeps the LDS reads of one slot from being hoisted above the arithmetic of the previous one
    }
  }

  template <int NV>
  __device__ __forceinline__ void sum(double (&v)[NV]) {
    block_sum<NV, ONEWAVE>(v, red + redslot * NRED * NW);
    redslot ^= 1;
  }

  // sums that are only stored (block_sum_post / block_sum_collect of qd_device.h): post, a barrier of the caller, collect
  double* pend;
  template <int NV, typename F>
  __device__ __forceinline__ void sum_store_post(const double (&v)[NV], F&& dst) {
    pend = red + redslot * NRED * NW;
    redslot ^= 1;
    block_sum_post<NV, ONEWAVE>(v, pend, dst);
  }
  template <int NV, typename F>
  __device__ __forceinline__ void sum_store_collect(F&& dst) const {
    block_sum_collect<NV, ONEWAVE>(pend, dst);
  }

  // block-wide sum of two floats; contains the one barrier of a solver iteration (multi-wave blocks)
  __device__ __forceinline__ void sum2_f32(float& a, float& b) {
    a = wave_sum_f32(a);
    b = wave_sum_f32(b);
    if (ONEWAVE) {
      team_sync<true>();
      return;
    }
    float* rf = reinterpret_cast<float*>(red + redslot * NRED * NW);
    redslot ^= 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
      rf[wave] = a;
      rf[NW + wave] = b;
    }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      sa += rf[w];
      sb += rf[NW + w];
    }
    a = sa;
    b = sb;
  }

  // the same for one value (every iteration but the first: ||b||^2 is only needed once)
  __device__ __forceinline__ void sum1_f32(float& a) {
    a = wave_sum_f32(a);
    if (ONEWAVE) {
      team_sync<true>();
      return;
    }
    float* rf = reinterpret_cast<float*>(red + redslot * NRED * NW);
    redslot ^= 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) rf[wave] = a;
    __syncthreads();
    float sa = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) sa += rf[w];
    a = sa;
  }

  // Solve (I - alpha M^{(T)}) y = b by the reference's Neumann iteration (timestepper.cpp:697-727).
  // Returns the number of RHS applications; on exit y is in registers.  The squared update norm is only compared
  // with a threshold: it is accumulated per thread in R and reduced over the workgroup in fp32, scaled by 1/abstol^2
  // (fp64 sweeps; keeps the fp32 value away from the subnormal range) exactly as in qd_device.h.
  template <bool TRANS>
  __device__ __forceinline__ int neumann(const SweepArgs& A, R alpha, const f2 (&b)[EPT], f2 (&y)[EPT]) {
    R nb2 = 0;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      y[j] = b[j];
      if (F32) nb2 = rfma(b[j].x, b[j].x, rfma(b[j].y, b[j].y, nb2));
    }
    publish(y);
    const R scale = F32 ? (R)1 : (R)(1.0 / (A.abstol * A.abstol));
    const float abs2 = F32 ? (float)(A.abstol * A.abstol) : 1.f;
    const float rel2 = (float)(A.reltol * A.reltol);
    float d0 = 1.f, tol2 = abs2, dprev = 1.f;
    int iter;
    for (iter = 0; iter < A.maxiter; iter++) {
      const f2* src = vec();
      f2* dst = buf + (cur ^ 1) * DIM;
      f2 w[EPT];
      R dl = 0;
#ifndef QD_F32_UNPACKED
      if constexpr (F32) {  // update and squared difference on (re, im) pairs
        pk2 dl2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const pk2 t = pkv(st.template apply<TRANS>(src, j, y));
          const pk2 wj = pk_fma(bc(alpha), t, pkv(b[j]));
          const pk2 dj = pkv(y[j]) - wj;
          dl2 = pk_fma(dj, dj, dl2);
          w[j] = pkf(wj);
          dst[elem(j)] = w[j];
          q32_fence<EPT>(j);
        }
        dl = dl2.x + dl2.y;
      } else
#endif
      {
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const f2 t = st.template apply<TRANS>(src, j, y);
          w[j].x = rfma(alpha, t.x, b[j].x);
          w[j].y = rfma(alpha, t.y, b[j].y);
          const R dx = y[j].x - w[j].x, dy = y[j].y - w[j].y;
          dl = rfma(dx, dx, rfma(dy, dy, dl));
          dst[elem(j)] = w[j];
          q32_fence<EPT>(j);
        }
      }
#pragma unroll
      for (int j = 0; j < EPT; j++) y[j] = w[j];
      float d = F32 ? (float)dl : (float)fmin((double)(dl * scale), 1e30), n2 = (float)nb2;
#ifdef QD_Q32_NOSTOP  // measurement build: three iterations per solve, no reduction (results meaningless)
      team_sync<ONEWAVE>();
      cur ^= 1;
      if (iter == 2) { iter++; break; }
      continue;
#endif
      // (the barrier that makes dst readable.  fp64 sweeps have no use for ||b||^2: one value - C5 14.85 -> 14.53 ms, 2^4 2.45 -> 2.35,
      //  profiles/r5_q32_ab2.txt; fp32-mixed needs it in the first iteration only, but a branch on the iteration number costs more than
      //  the second value: 2^4 2.05 -> 2.45 ms)
      if constexpr (F32) sum2_f32(d, n2);
      else sum1_f32(d);
      cur ^= 1;
      // One exit branch per iteration, the first iteration's values by selects [r5]: two exits and a block under `iter == 0` cost the
      // 2^4 forward sweep 7 % (fp32-mixed 2.05 -> 1.91 ms, fp64 2.35 -> 2.26; profiles/r5_q32_ab3.txt).  Same decisions, same counts.
      // (fp32-mixed: the iterates cannot get below their fp32 floor - no error-estimate rule there; n2 is the same in every iteration)
      const bool first = iter == 0;
      d0 = first ? d : d0;
      if (F32) tol2 = fmaxf(abs2, F32_SOLVER_TOL * F32_SOLVER_TOL * n2);
      const bool stop = (F32 ? d <= tol2 : (d < 1.f && standin_ok(A.standin_tau2, d, first ? d : dprev, 1.f))) | (d < rel2 * d0);
      dprev = d;
      if (stop) { iter++; break; }
    }
    return iter;
  }

  // GMRES for (I - alpha M^{(T)}) y = b, as Team::gmres_g of qd_device.h (KSPGMRES + PCNONE of the reference: zero initial
  // guess, classical Gram-Schmidt, Givens rotations, restart 30, stop at max(rtol ||b||, abstol)); the Krylov basis lives in
  // global memory (L2 / Infinity-Cache resident for the 2^5 system: ~5 vectors x 16 KB per initial condition), every thread
  // only touches its own elements of it.  All projections of an iteration go through one block reduction.  Returns the
  // number of RHS applications; y in registers.
  template <bool TRANS>
  __device__ __forceinline__ int gmres(const SweepArgs& A, R alpha, const f2 (&b)[EPT], f2 (&y)[EPT]) {
    constexpr int MR = GMRES_MR_G;
    f2* __restrict__ Vg = reinterpret_cast<f2*>(A.kry) + (size_t)blockIdx.x * (MR + 2) * DIM;
    double* hc = ksc;
    double* cs = hc + (MR + 2);
    double* sn = cs + MR;
    double* g = sn + MR;
    double* Rm = g + (MR + 2);
    double* yk = Rm + MR * MR;
    f2 yy[EPT], r[EPT], v[EPT], w[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      yy[j].x = yy[j].y = 0;
      r[j] = b[j];
    }
    int its = 0, napp = 0;
    double ttol = 0.0;
    for (int cycle = 0;; cycle++) {
      double t1[1] = {0.0};
#pragma unroll
      for (int j = 0; j < EPT; j++) t1[0] += (double)r[j].x * r[j].x + (double)r[j].y * r[j].y;
      sum<1>(t1);
      const double ibeta_d = t1[0] > 0.0 ? rsqrt_nr(t1[0]) : 0.0;
      const double beta = t1[0] * ibeta_d;
      // fp32 vectors: the true residual stalls near 2^-22 ||b|| while the recurrence residual keeps falling - the same floor as
      // the fp32 Neumann iteration's stopping rule
      if (cycle == 0) ttol = fmax(fmax(A.reltol * beta, A.abstol), std::is_same<R, float>::value ? (double)F32_SOLVER_TOL * beta : 0.0);
      if (beta <= ttol || its >= A.maxiter) break;
      const R ibeta = (R)ibeta_d;
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        v[j].x = r[j].x * ibeta;
        v[j].y = r[j].y * ibeta;
        if (cycle > 0) Vg[opaque(elem(j))] = v[j];  // (first cycle: v_0 = b / beta is recomputed from registers)
      }
      publish(v);
      double gcur = beta;
      int jj = 0;
      bool conv = false;
      while (jj < MR) {
        f2 t2[EPT];
        apply_all<TRANS>(v, t2);
        napp++;
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          w[j].x = v[j].x - alpha * t2[j].x;
          w[j].y = v[j].y - alpha * t2[j].y;
        }
        // projections, up to 8 per block reduction (the norm of the orthogonalised vector is reduced separately: the identity
        // ||w - sum h_k v_k||^2 = ||w||^2 - sum h_k^2 breaks down with the orthogonality of classical Gram-Schmidt)
        // Positions of a block: 0 = v_jj (still in registers), 1 = v_0 (first cycle: b / beta, recomputed from registers), then
        // v_1 .. v_{jj-1} read back from the basis in one branch-free run of loads.
        auto v0elem = [&](int j) {
          f2 vk;
          if (cycle == 0) {
            vk.x = b[j].x * ibeta;
            vk.y = b[j].y * ibeta;
          } else {
            vk = Vg[opaque(elem(j))];
          }
          return vk;
        };
        for (int p0 = 0; p0 <= jj; p0 += 8) {
          double h8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          const int np = min(8, jj + 1 - p0);
          int q0 = 0;
          if (p0 == 0) {
#pragma unroll
            for (int j = 0; j < EPT; j++) h8[0] += (double)w[j].x * v[j].x + (double)w[j].y * v[j].y;
            q0 = 1;
            if (jj >= 1) {
#pragma unroll
              for (int j = 0; j < EPT; j++) {
                const f2 vk = v0elem(j);
                h8[1] += (double)w[j].x * vk.x + (double)w[j].y * vk.y;
              }
              q0 = 2;
            }
          }
          for (int q = q0; q < np; q++) {
#pragma unroll
            for (int j = 0; j < EPT; j++) {
              const f2 vk = Vg[(size_t)(p0 + q - 1) * DIM + opaque(elem(j))];
              h8[q] += (double)w[j].x * vk.x + (double)w[j].y * vk.y;
            }
          }
          if (np <= 2) sum<2>(reinterpret_cast<double(&)[2]>(h8));
          else if (np <= 4) sum<4>(reinterpret_cast<double(&)[4]>(h8));
          else sum<8>(h8);
          for (int q = 0; q < np; q++) hc[p0 + q == 0 ? jj : p0 + q - 1] = h8[q];
        }
        {
          const R h = (R)hc[jj];
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            w[j].x -= h * v[j].x;
            w[j].y -= h * v[j].y;
          }
        }
        if (jj >= 1) {
          const R h = (R)hc[0];
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            const f2 vk = v0elem(j);
            w[j].x -= h * vk.x;
            w[j].y -= h * vk.y;
          }
        }
        for (int k = 1; k < jj; k++) {
          const R h = (R)hc[k];
#pragma unroll
          for (int j = 0; j < EPT; j++) {
            const f2 vk = Vg[(size_t)k * DIM + opaque(elem(j))];
            w[j].x -= h * vk.x;
            w[j].y -= h * vk.y;
          }
        }
        double nn[1] = {0.0};
#pragma unroll
        for (int j = 0; j < EPT; j++) nn[0] += (double)w[j].x * w[j].x + (double)w[j].y * w[j].y;
        sum<1>(nn);
        const double ihn_d = nn[0] > 0.0 ? rsqrt_nr(nn[0]) : 0.0;
        const double hn = nn[0] * ihn_d;
        hc[jj + 1] = hn;
        // Givens rotations: redundantly by every thread on wave-uniform values, idempotent LDS writes only
        double cur_h = hc[0];
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
        if (its >= A.maxiter || jj >= MR) break;
        // the next basis vector is only formed, stored and published when another iteration follows
        const R ihn = (R)ihn_d;
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          v[j].x = w[j].x * ihn;
          v[j].y = w[j].y * ihn;
          Vg[(size_t)jj * DIM + opaque(elem(j))] = v[j];
        }
        publish(v);  // v_{jj} becomes the stencil-readable vector; its barrier also orders the scalar writes
      }
      for (int rw = jj - 1; rw >= 0; rw--) {
        double sacc = g[rw];
        for (int cc = rw + 1; cc < jj; cc++) sacc -= Rm[rw * MR + cc] * yk[cc];
        yk[rw] = sacc * Rm[rw * MR + rw];
      }
      if (jj >= 1) {
        const R f = (R)yk[0];
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          f2 vk;
          if (cycle == 0) {
            vk.x = b[j].x * ibeta;
            vk.y = b[j].y * ibeta;
          } else {
            vk = Vg[opaque(elem(j))];
          }
          yy[j].x += f * vk.x;
          yy[j].y += f * vk.y;
        }
      }
      for (int cc = 1; cc < jj; cc++) {
        const R f = (R)yk[cc];
#pragma unroll
        for (int j = 0; j < EPT; j++) {
          const f2 vk = Vg[(size_t)cc * DIM + opaque(elem(j))];
          yy[j].x += f * vk.x;
          yy[j].y += f * vk.y;
        }
      }
      if (conv || its >= A.maxiter) break;
      publish(yy);  // restart: r = b - (I - alpha M) y
      f2 t3[EPT];
      apply_all<TRANS>(yy, t3);
      napp++;
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        r[j].x = b[j].x - (yy[j].x - alpha * t3[j].x);
        r[j].y = b[j].y - (yy[j].y - alpha * t3[j].y);
      }
      team_sync<ONEWAVE>();  // every thread has read the scalars of this cycle before the next one overwrites them
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) y[j] = yy[j];
    return napp;
  }

  // [r6] GMRES right-preconditioned with the Neumann polynomial P = sum_{i<p} (alpha M)^i, one Krylov vector (fp64 sweeps, A.gmres_poly =
  // p > 1; as ColTeam::kry_solve of qd_col.hip): z = P b by Horner's rule - the Neumann pass y <- b + alpha M y, p - 1 times, WITHOUT its
  // reduction - then one application w = (I - alpha M) z fused with <b,b>, <r,b>, <r,r> of r = b - w in ONE workgroup reduction:
  // h_00 = 1 - a, a = <r,b> / <b,b>, h_10^2 = <r,r> / <b,b> - a^2, y = h_00 / (h_00^2 + h_10^2) z, residual = ||b|| h_10 / sqrt(h_00^2 + h_10^2)
  // against max(rtol ||b||, abstol) (KSPGMRES, src/timestepper.cpp:541-550).  The true residual of the accepted solution is tested - the
  // reference's rule - with one reduction per solve where the stationary iteration needs one per pass.  Returns a negative value where one
  // vector does not reach the tolerance (the caller then runs the plain GMRES from scratch).
  template <bool TRANS>
  __device__ __forceinline__ int kry1(const SweepArgs& A, R alpha, const f2 (&b)[EPT], f2 (&y)[EPT]) {
    const int poly = A.gmres_poly;
#pragma unroll
    for (int j = 0; j < EPT; j++) y[j] = b[j];
    publish(y);
    for (int m = 1; m < poly; m++) {
      const f2* src = vec();
      f2* dst = buf + (cur ^ 1) * DIM;
      f2 w[EPT];
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const f2 t = st.template apply<TRANS>(src, j, y);
        w[j].x = rfma(alpha, t.x, b[j].x);
        w[j].y = rfma(alpha, t.y, b[j].y);
        dst[elem(j)] = w[j];
        q32_fence<EPT>(j);
      }
#pragma unroll
      for (int j = 0; j < EPT; j++) y[j] = w[j];
      team_sync<ONEWAVE>();
      cur ^= 1;
    }
    double d[3] = {0.0, 0.0, 0.0};
    {
      const f2* src = vec();
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const f2 t = st.template apply<TRANS>(src, j, y);
        const double rx = (double)b[j].x - (double)rfma(-alpha, t.x, y[j].x), ry = (double)b[j].y - (double)rfma(-alpha, t.y, y[j].y);
        d[0] = fma((double)b[j].x, (double)b[j].x, fma((double)b[j].y, (double)b[j].y, d[0]));
        d[1] = fma(rx, (double)b[j].x, fma(ry, (double)b[j].y, d[1]));
        d[2] = fma(rx, rx, fma(ry, ry, d[2]));
        q32_fence<EPT>(j);
      }
    }
    sum<3>(d);
    const double bb = d[0], ttol2 = fmax(A.reltol * A.reltol * bb, A.abstol * A.abstol);
    double fac;
    if (bb <= ttol2) {
      fac = 0.0;  // ||b|| <= tolerance: KSP returns the zero initial guess
    } else {
      const double ibb = 1.0 / bb, a = d[1] * ibb, h00 = 1.0 - a, h10sq = fmax(fma(-a, a, d[2] * ibb), 0.0), den = fma(h00, h00, h10sq);
      if (!(bb * h10sq <= A.kry_tau2 * ttol2 * den || A.maxiter <= 1) || !(h00 > 0.0)) return -1;  // (kry_tau: SweepArgs)
      fac = h00 / den;
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      y[j].x = (R)fac * y[j].x;
      y[j].y = (R)fac * y[j].y;
    }
    return poly;
  }

  template <bool TRANS>
  __device__ __forceinline__ int solve(const SweepArgs& A, R alpha, const f2 (&b)[EPT], f2 (&y)[EPT]) {
    if constexpr (GM) {
      if constexpr (!F32) {
        if (A.gmres_poly > 1) {
          const int n = kry1<TRANS>(A, alpha, b, y);
          if (n >= 0) return n;
          return A.gmres_poly + gmres<TRANS>(A, alpha, b, y);
        }
      }
      return gmres<TRANS>(A, alpha, b, y);
    } else {
      return neumann<TRANS>(A, alpha, b, y);
    }
  }
};

template <typename R> __device__ __forceinline__ typename Vec2<R>::type to_r2(const double2 v);
template <> __device__ __forceinline__ float2 to_r2<float>(const double2 v) { return make_float2((float)v.x, (float)v.y); }
template <> __device__ __forceinline__ double2 to_r2<double>(const double2 v) { return v; }

// Stored trajectory.  fp32-mixed: [nsub+1][nb][dim] interleaved float2; fp64: the layout of every other fp64 kernel and of
// qd_get_state, [nsub+1][nb][2 dim] blocked [u ; v].
template <typename R> struct Traj;
// (trajectory and stored stages are written once and read by the adjoint sweep much later: streaming stores / loads)
typedef float traj_f2 __attribute__((ext_vector_type(2)));
template <> struct Traj<float> {
  __device__ __forceinline__ static void store(double* base, size_t state, int dim, int e, float2 v) {
    traj_f2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<traj_f2*>(base) + state * dim + e);
  }
  __device__ __forceinline__ static float2 load(const double* base, size_t state, int dim, int e) {
    const traj_f2 t = __builtin_nontemporal_load(reinterpret_cast<const traj_f2*>(base) + state * dim + e);
    return make_float2(t.x, t.y);
  }
};
template <> struct Traj<double> {
  __device__ __forceinline__ static void store(double* base, size_t state, int dim, int e, double2 v) {
    __builtin_nontemporal_store(v.x, base + state * 2 * dim + e);
    __builtin_nontemporal_store(v.y, base + state * 2 * dim + dim + e);
  }
  __device__ __forceinline__ static double2 load(const double* base, size_t state, int dim, int e) {
    return make_double2(__builtin_nontemporal_load(base + state * 2 * dim + e), __builtin_nontemporal_load(base + state * 2 * dim + dim + e));
  }
};

// The stored primal stages are private to a forward / adjoint pair of THESE kernels (never read by the host): one 16-byte access per
// element in fp64 too (the trajectory proper keeps the [u; v] blocks qd_get_state reads).  qd_handle records the layout a forward sweep
// stored (ztraj_fmt) and refuses an adjoint sweep of another kernel family on it.
template <typename R> struct ZTraj : Traj<R> {};
typedef double traj_d2 __attribute__((ext_vector_type(2)));
template <> struct ZTraj<double> {
  __device__ __forceinline__ static void store(double* base, size_t state, int dim, int e, double2 v) {
    traj_d2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<traj_d2*>(base) + state * dim + e);
  }
  __device__ __forceinline__ static double2 load(const double* base, size_t state, int dim, int e) {
    const traj_d2 t = __builtin_nontemporal_load(reinterpret_cast<const traj_d2*>(base) + state * dim + e);
    return make_double2(t.x, t.y);
  }
};

// ---------------------------------------------------------------------------------------------
// forward sweep (TimeStepper::solveODE for every initial condition of the batch)
// ---------------------------------------------------------------------------------------------
template <int Q, int SB, typename R, bool GM = false, bool HJ = false>
__global__ void __launch_bounds__((Q32<Q, SB, R>::NT), (QD_F32HJ_W > 0 && HJ && !GM && sizeof(R) == 4 && Q == 5 ? QD_F32HJ_W : Q32<Q, SB, R>::MINW >> ((GM && sizeof(R) == 4) || HJ ? 1 : 0))) k_forward_q32(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef Team32<Q, SB, R, GM, HJ> TM;
  typedef typename TM::f2 f2;
  constexpr int EPT = TM::EPT, DIM = TM::DIM;
  const DevSys& S = A.S;
  TM tm;
  tm.init(S, smem);
  const int ic = blockIdx.x;
  double2 x[EPT];  // the state itself: fp64 accumulators
  {
    const double* x0 = A.x0 + (size_t)ic * 2 * DIM;
#pragma unroll
    for (int j = 0; j < EPT; j++) x[j] = make_double2(x0[tm.elem(j)], x0[DIM + tm.elem(j)]);
  }
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;
  double pen_local = 0.0, pen_uniform = 0.0;
  unsigned long long napply = 0;

  vm_drain();
  for (int s = 0; s < A.nsub; s++) {
    StepC<Q> c;
    load_step_k<Q>(A.ctl + (size_t)s * A.cs, c, HJ);
    tm.st.prep(c);
    const R hf = uniform((R)c.h);
    f2 xs[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) xs[j] = to_r2<R>(x[j]);
    if (A.traj) {
#pragma unroll
      for (int j = 0; j < EPT; j++) Traj<R>::store(A.traj, (size_t)s * A.nb + ic, DIM, tm.elem(j), xs[j]);
    }
    tm.publish(xs);
    tm.park(x);  // the fp64 accumulators are dead weight during the solve
    f2 rhs[EPT], k[EPT];
    tm.template apply_all<false>(xs, rhs);  // rhs = M x (ImplMidpoint::evolveFWD, timestepper.cpp:594)
    napply += 1 + tm.template solve<false>(A, (R)0.5 * hf, rhs, k);
    tm.unpark(x);
    const double h = to_scalar(c.h);
    if (A.ztraj) {  // the primal stage z = x + h/2 k in R, as the adjoint sweep would recompute it: stored instead
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const f2 xr = to_r2<R>(x[j]);
        f2 z;
        z.x = rfma((R)0.5 * hf, k[j].x, xr.x);
        z.y = rfma((R)0.5 * hf, k[j].y, xr.y);
        ZTraj<R>::store(A.ztraj, (size_t)s * A.nb + ic, DIM, tm.elem(j), z);
      }
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {  // x += h k: fp64 accumulation of the stage
      x[j].x = fma(h, (double)k[j].x, x[j].x);
      x[j].y = fma(h, (double)k[j].y, x[j].y);
    }
    // weighted-J penalty at the end of a FULL time step (timestepper.cpp:141-154, :256-298); all levels are essential
    if (wj_on && (s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages - 1;
      const double tstop = (n + 1) * A.dt;
      const double a = (tstop - A.Tfinal) / A.penalty_param;
      const double weight = A.wjw ? to_scalar(A.wjw[n]) : 1.0 / A.penalty_param * exp(-(a * a));  // (tabulated per time step)
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        double jr = 0.0, ji = 0.0;
        evalJ_part<true>(S, A.tg, ic, opaque(tm.elem(j)), x[j], jr, ji);
        pen_local += (A.tg.objective_type == QD_OBJ_JTRACE ? -1.0 : 1.0) * weight * A.dt * jr;
      }
      if (A.tg.objective_type == QD_OBJ_JTRACE) pen_uniform += weight * A.dt;
    }
  }
  {
    double* xT = A.xT + (size_t)ic * 2 * DIM;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      xT[tm.elem(j)] = x[j].x;
      xT[DIM + tm.elem(j)] = x[j].y;
    }
    if (A.traj) {
#pragma unroll
      for (int j = 0; j < EPT; j++) Traj<R>::store(A.traj, (size_t)A.nsub * A.nb + ic, DIM, tm.elem(j), to_r2<R>(x[j]));
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

// ---------------------------------------------------------------------------------------------
// adjoint sweep (TimeStepper::solveAdjointODE + ImplMidpoint::evolveBWD + compute_dRHS_dParams)
// ---------------------------------------------------------------------------------------------
template <int Q, int SB, typename R, bool GM = false, bool HJ = false>
__global__ void __launch_bounds__((Q32<Q, SB, R>::NT), (QD_F32HJ_W > 0 && HJ && !GM && sizeof(R) == 4 && Q == 5 ? QD_F32HJ_W : Q32<Q, SB, R>::MINW >> ((GM && sizeof(R) == 4) || HJ ? 1 : 0))) k_adjoint_q32(const SweepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef Team32<Q, SB, R, GM, HJ> TM;
  typedef typename TM::f2 f2;
  constexpr int EPT = TM::EPT, DIM = TM::DIM;
  const DevSys& S = A.S;
  TM tm;
  tm.init(S, smem, true);
  const int ic = blockIdx.x;
  double2 xb[EPT];  // the adjoint state: fp64 accumulators
  {
    const double* xbT = A.xbarT + (size_t)ic * 2 * DIM;
#pragma unroll
    for (int j = 0; j < EPT; j++) xb[j] = make_double2(xbT[tm.elem(j)], xbT[DIM + tm.elem(j)]);
  }
  const double jbar_pen = A.jbar[ic * 3 + 0];
  const bool pen_on = A.gamma_penalty > 1e-13;
  const bool wj_on = pen_on && A.penalty_param > 1e-13;

  vm_drain();
  for (int s = A.nsub - 1; s >= 0; s--) {
    // penaltyIntegral_diff at the end of a full step, with the primal x_n (timestepper.cpp:220-227, :300-339)
    if (wj_on && (s + 1) % A.nstages == 0) {
      const int n = (s + 1) / A.nstages;
      const double tstop = n * A.dt;
      const double a = (tstop - A.Tfinal) / A.penalty_param;
      const double weight = A.wjw ? to_scalar(A.wjw[n - 1]) : 1.0 / A.penalty_param * exp(-(a * a));
      double rb, ib;
      finalizeJ_diff<true>(A.tg, 0.0, 0.0, rb, ib);
#pragma unroll
      for (int j = 0; j < EPT; j++) {
        const f2 v = Traj<R>::load(A.traj, (size_t)(s + 1) * A.nb + ic, DIM, tm.elem(j));
        evalJ_diff_elem<true>(S, A.tg, ic, opaque(tm.elem(j)), make_double2(v.x, v.y), xb[j], weight * rb * jbar_pen * A.dt, weight * ib * jbar_pen * A.dt);
      }
    }
    StepC<Q> c;
    load_step_k<Q>(A.ctl + (size_t)s * A.cs, c, HJ);
    tm.st.prep(c);
    const R hf = uniform((R)c.h);
    // ImplMidpoint::evolveBWD (timestepper.cpp:631-694).  The primal stage z = x + h/2 k of the sub-step (:640-652) was stored by
    // the forward sweep (SweepArgs::ztraj): only the adjoint solve remains.
    tm.park(xb);  // the fp64 adjoint accumulators are dead weight until xbar += M^T kbar
    // adjoint stage (I - h/2 M)^T kbar = xbar ; kbar *= h
    f2 kb[EPT], bb[EPT];
    if (TM::PARK) {
#pragma unroll
      for (int j = 0; j < EPT; j++) bb[j] = to_r2<R>(tm.acc[tm.elem(j)]);
    } else {
#pragma unroll
      for (int j = 0; j < EPT; j++) bb[j] = to_r2<R>(xb[j]);
    }
    tm.template solve<true>(A, (R)0.5 * hf, bb, kb);
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      kb[j].x *= hf;
      kb[j].y *= hf;
    }
    // gradient coefficients x^T dM/dp_k z, x^T dM/dq_k z with x := kbar (mastereq.hpp:553-604): products in R, fp64 sums
    f2 z[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) z[j] = ZTraj<R>::load(A.ztraj, (size_t)s * A.nb + ic, DIM, tm.elem(j));
    tm.publish(z);
    double cf[2 * Q];
#pragma unroll
    for (int i = 0; i < 2 * Q; i++) cf[i] = 0.0;
    tm.st.ladder_all(tm.vec(), z, kb, cf);
    auto coeff_dst = [&](int g) -> double* { return A.coeff + ((size_t)ic * A.nsub + s) * 2 * Q + g; };
    tm.template sum_store_post<2 * Q>(cf, coeff_dst);
    // xbar += M^T kbar
    tm.publish(kb);  // (its barrier also completes the coefficient sums)
    tm.template sum_store_collect<2 * Q>(coeff_dst);
    f2 t[EPT];
    tm.template apply_all<true>(kb, t);
    tm.unpark(xb);
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      xb[j].x += (double)t[j].x;
      xb[j].y += (double)t[j].y;
    }
  }
  if (A.xbar0) {
    double* d0 = A.xbar0 + (size_t)ic * 2 * DIM;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      d0[tm.elem(j)] = xb[j].x;
      d0[DIM + tm.elem(j)] = xb[j].y;
    }
  }
}

// single operator application (test hook: qd_apply_rhs with QD_PRECISION_F32MIXED; timing loop of the MFMA measurement)
template <int Q, int SB, typename R, bool HJ = false>
__global__ void __launch_bounds__((Q32<Q, SB, R>::NT), (Q32<Q, SB, R>::MINW >> (HJ ? 1 : 0))) k_apply_q32(const DevSys S, const double* __restrict__ ctlrow, int transpose,
                                                                                         const double* __restrict__ xin, double* __restrict__ yout, int nrep) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef Team32<Q, SB, R, false, HJ> TM;
  typedef typename TM::f2 f2;
  constexpr int EPT = TM::EPT, DIM = TM::DIM;
  TM tm;
  tm.init(S, smem, transpose != 0);
  const int ic = blockIdx.x;
  f2 x[EPT], y[EPT];
  const double* x0 = xin + (size_t)ic * 2 * DIM;
#pragma unroll
  for (int j = 0; j < EPT; j++) x[j] = to_r2<R>(make_double2(x0[tm.elem(j)], x0[DIM + tm.elem(j)]));
  StepC<Q> c;
  load_step_k<Q>(ctlrow, c, HJ);
  tm.st.prep(c);
  tm.publish(x);
  if (transpose) tm.template apply_all<true>(x, y);
  else tm.template apply_all<false>(x, y);
  // nrep > 1: timing loop of the stencil-vs-MFMA measurement (y <- M (1e-3 y), keeps the values bounded)
  for (int r = 1; r < nrep; r++) {
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      x[j].x = (R)1e-3 * y[j].x;
      x[j].y = (R)1e-3 * y[j].y;
    }
    tm.publish(x);
    if (transpose) tm.template apply_all<true>(x, y);
    else tm.template apply_all<false>(x, y);
  }
  double* yo = yout + (size_t)ic * 2 * DIM;
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    yo[tm.elem(j)] = (double)y[j].x;
    yo[DIM + tm.elem(j)] = (double)y[j].y;
  }
}

// ---------------------------------------------------------------------------------------------
// The MFMA question: the same operator as a batched dense Kronecker-factor product
//   Y = G rho - rho G + (dissipators),   G = -i H(t)  (N x N complex, N = 32)
// on the fp32 matrix cores, v_mfma_f32_32x32x2_f32, one wave per initial condition and 32 x 32 tile.  Per application
// 2 complex 32x32x32 products = 8 real ones = 128 MFMA instructions of 64 cycles each, against ~220 two-cycle VALU
// instructions per wave for the stencil: the measurement (qd_bench_apply_f32) is recorded in profiles/HISTORY.md (section 4).
// Layouts (cdna_hip_programming.md section 3): A operand lane l holds A[i = l & 31][k = l >> 5], B operand B[k = l >> 5][j = l & 31],
// C/D: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// ---------------------------------------------------------------------------------------------
typedef float mfma_f16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(64) k_apply_mfma32(const DevSys S, const double* __restrict__ ctlrow, const double* __restrict__ xin,
                                                     double* __restrict__ yout, int nrep) {
  constexpr int N = 32, Q = 5, NP = 33;           // row stride 33 floats: column reads (lane = row) are conflict-free
  __shared__ float gre[N * NP], gim[N * NP];      // G(t), row-major
  __shared__ float rre[N * NP], rim[N * NP];      // rho, row-major [row I][col I']
  const int lane = threadIdx.x, ic = blockIdx.x;
  const double* x0 = xin + (size_t)ic * 2 * N * N;
  // H(t) = diag(h(I)) + sum_k p_k (a_k + a_k^dag) + i q_k (a_k - a_k^dag)  ->  G = -i H:
  //   G[I][I] = -i h(I);  G[I][I ^ b_k] = -i p_k + s q_k  with s = +1 if digit_k(I) = 1 (the a_k entry) else -1
  for (int e = lane; e < N * NP; e += 64) {
    gre[e] = 0.f;
    gim[e] = 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (lane < N) {
    const int I = lane;
    double hd = 0.0;
    int pair = 0;
    for (int k = 0; k < Q; k++) {
      const int a = (I >> (Q - 1 - k)) & 1;
      hd += S.detune[k] * a;
      for (int l = k + 1; l < Q; l++) hd -= S.xikl[pair++] * a * ((I >> (Q - 1 - l)) & 1);
      const double p = ctlrow[2 + k], q = ctlrow[2 + Q + k];
      const int J = I ^ (1 << (Q - 1 - k));
      gre[I * NP + J] = (float)(a ? -q : q);  // -i (p + i q) = q - i p above the diagonal (digit 0), -i (p - i q) = -q - i p below
      gim[I * NP + J] = (float)(-p);
    }
    gim[I * NP + I] = (float)(-hd);
  }
  for (int e = lane; e < N * N; e += 64) {  // rho[I][I'] = x[I + N I']
    const int I = e / N, Ip = e % N;
    rre[I * NP + Ip] = (float)x0[I + N * Ip];
    rim[I * NP + Ip] = (float)x0[N * N + I + N * Ip];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int i32 = lane & 31, kh = lane >> 5;
  mfma_f16 yr, yi;
  for (int rep = 0; rep < nrep; rep++) {
#pragma unroll
    for (int r = 0; r < 16; r++) yr[r] = yi[r] = 0.f;
    for (int k0 = 0; k0 < N; k0 += 2) {
      const int kk = k0 + kh;
      // first product G rho: A = G[i][kk], B = rho[kk][j]
      const float ar = gre[i32 * NP + kk], ai = gim[i32 * NP + kk];
      const float br = rre[kk * NP + i32], bi = rim[kk * NP + i32];
      yr = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, yr, 0, 0, 0);
      yr = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, yr, 0, 0, 0);
      yi = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, yi, 0, 0, 0);
      yi = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, yi, 0, 0, 0);
      // second product - rho G: A = rho[i][kk], B = G[kk][j]
      const float cr = rre[i32 * NP + kk], ci = rim[i32 * NP + kk];
      const float dr = gre[kk * NP + i32], di = gim[kk * NP + i32];
      yr = __builtin_amdgcn_mfma_f32_32x32x2f32(-cr, dr, yr, 0, 0, 0);
      yr = __builtin_amdgcn_mfma_f32_32x32x2f32(ci, di, yr, 0, 0, 0);
      yi = __builtin_amdgcn_mfma_f32_32x32x2f32(-cr, di, yi, 0, 0, 0);
      yi = __builtin_amdgcn_mfma_f32_32x32x2f32(-ci, dr, yi, 0, 0, 0);
    }
    // dissipators (diagonal d x, T1 off-diagonal) on the accumulator layout, then write back as the next rho
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float nr[16], ni[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int I = (r & 3) + 8 * (r >> 2) + 4 * kh, Ip = i32;
      double d = 0.0;
      float l1r = 0.f, l1i = 0.f;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        const int a = (I >> (Q - 1 - k)) & 1, ap = (Ip >> (Q - 1 - k)) & 1;
        d += S.g2[k] * (a * ap - 0.5 * (a + ap)) - S.g1[k] / 2.0 * (a + ap);
        if (a == 0 && ap == 0) {
          const int I2 = I | (1 << (Q - 1 - k)), Ip2 = Ip | (1 << (Q - 1 - k));
          l1r = fmaf((float)S.g1off[k], rre[I2 * NP + Ip2], l1r);
          l1i = fmaf((float)S.g1off[k], rim[I2 * NP + Ip2], l1i);
        }
      }
      nr[r] = fmaf((float)d, rre[I * NP + Ip], yr[r]) + l1r;
      ni[r] = fmaf((float)d, rim[I * NP + Ip], yi[r]) + l1i;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (rep + 1 < nrep) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int I = (r & 3) + 8 * (r >> 2) + 4 * kh;
        rre[I * NP + i32] = 1e-3f * nr[r];
        rim[I * NP + i32] = 1e-3f * ni[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else {
      double* yo = yout + (size_t)ic * 2 * N * N;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int I = (r & 3) + 8 * (r >> 2) + 4 * kh;
        yo[I + N * i32] = (double)nr[r];
        yo[N * N + I + N * i32] = (double)ni[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <typename K>
static hipError_t set_lds32(K kern, size_t bytes) {
  if (bytes > 48 * 1024)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return hipSuccess;
}

// slot bits per thread for a Q-qubit Lindblad system: 2^5 -> 4 elements per thread (256 threads); 2^4 -> one element per
// thread (256 threads; the option f32_sb = 2 selects the one-wave, four-elements-per-thread layout for measurements)
static int q32_slot_bits(int Q, const TuneOpts& o) {
  if (Q == 5) return 2;
  return o.f32_sb == 2 ? 2 : 0;
}

template <int Q, int SB, typename R, bool GM = false, bool HJ = false>
static hipError_t go_fwd(const SweepArgs& a, hipStream_t st) {
  constexpr int nt = Q32<Q, SB, R>::NT;
  const size_t lds = Team32<Q, SB, R, GM, HJ>::lds_bytes();
  auto kf = k_forward_q32<Q, SB, R, GM, HJ>;
  hipError_t e = set_lds32(kf, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3(a.nb), dim3(nt), lds, st, a);
  return hipGetLastError();
}
template <int Q, int SB, typename R, bool GM = false, bool HJ = false>
static hipError_t go_adj(const SweepArgs& a, hipStream_t st) {
  constexpr int nt = Q32<Q, SB, R>::NT;
  const size_t lds = Team32<Q, SB, R, GM, HJ>::lds_bytes();
  auto kf = k_adjoint_q32<Q, SB, R, GM, HJ>;
  hipError_t e = set_lds32(kf, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3(a.nb), dim3(nt), lds, st, a);
  return hipGetLastError();
}
template <int Q, int SB, typename R, bool HJ = false>
static hipError_t go_app(const DevSys& S, const double* ctlrow, int tr, const double* x, double* y, int nb, int nrep, hipStream_t st) {
  constexpr int nt = Q32<Q, SB, R>::NT;
  const size_t lds = Team32<Q, SB, R, false, HJ>::lds_bytes();
  auto kf = k_apply_q32<Q, SB, R, HJ>;
  hipError_t e = set_lds32(kf, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3(nb), dim3(nt), lds, st, S, ctlrow, tr, x, y, nrep);
  return hipGetLastError();
}

// 2^5 system, batches of at most one state per CU: 512 threads x 2 elements (see lean64_sb below; the same option)
static bool small_batch_sb1(const SweepArgs& a, const TuneOpts& o) {
  if (o.lean64_sb == 1 || o.lean64_sb == 2) return o.lean64_sb == 1;
  return a.nb <= 256;
}

hipError_t launch_forward_f32(const SweepArgs& a, const TuneOpts& o, hipStream_t st) {
  const int sb = q32_slot_bits(a.S.Q, o);
  if (a.S.hasJ) {  // [r6] dipole-dipole coupling: the coupled stencils in fp32 (stationary iterations only)
    if (a.use_gmres) return hipErrorInvalidValue;
    if (a.S.Q == 5) return go_fwd<5, 1, float, false, true>(a, st);
    if (a.S.Q == 4) return go_fwd<4, 0, float, false, true>(a, st);
    return hipErrorInvalidValue;
  }
  if (a.use_gmres) {  // Krylov basis in global memory as float2, Hessenberg problem in fp64
    if (a.S.Q == 5) return go_fwd<5, 2, float, true>(a, st);
    if (a.S.Q == 4) return go_fwd<4, 0, float, true>(a, st);
    if (a.S.Q == 3) return go_fwd<3, 0, float, true>(a, st);
    return hipErrorInvalidValue;
  }
  if (a.S.Q == 5) return small_batch_sb1(a, o) ? go_fwd<5, 1, float>(a, st) : go_fwd<5, 2, float>(a, st);
  if (a.S.Q == 4) return sb == 2 ? go_fwd<4, 2, float>(a, st) : go_fwd<4, 0, float>(a, st);
  if (a.S.Q == 3) return go_fwd<3, 0, float>(a, st);  // [r3] 2x2x2 (BASELINE config 2): one wave per initial condition
  return hipErrorInvalidValue;
}
hipError_t launch_adjoint_f32(const SweepArgs& a, const TuneOpts& o, hipStream_t st) {
  const int sb = q32_slot_bits(a.S.Q, o);
  if (a.S.hasJ) {
    if (a.use_gmres) return hipErrorInvalidValue;
    if (a.S.Q == 5) return go_adj<5, 1, float, false, true>(a, st);
    if (a.S.Q == 4) return go_adj<4, 0, float, false, true>(a, st);
    return hipErrorInvalidValue;
  }
  if (a.use_gmres) {
    if (a.S.Q == 5) return go_adj<5, 2, float, true>(a, st);
    if (a.S.Q == 4) return go_adj<4, 0, float, true>(a, st);
    if (a.S.Q == 3) return go_adj<3, 0, float, true>(a, st);
    return hipErrorInvalidValue;
  }
  if (a.S.Q == 5) return small_batch_sb1(a, o) ? go_adj<5, 1, float>(a, st) : go_adj<5, 2, float>(a, st);
  if (a.S.Q == 4) return sb == 2 ? go_adj<4, 2, float>(a, st) : go_adj<4, 0, float>(a, st);
  if (a.S.Q == 3) return go_adj<3, 0, float>(a, st);
  return hipErrorInvalidValue;
}
hipError_t launch_apply_f32(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb, int nrep, int mfma,
                            const TuneOpts& o, hipStream_t st) {
  if (mfma) {
    if (S.Q != 5 || transpose) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_apply_mfma32, dim3(nb), dim3(64), 0, st, S, ctlrow, x, y, nrep);
    return hipGetLastError();
  }
  const int sb = q32_slot_bits(S.Q, o);
  if (S.hasJ) {
    if (S.Q == 5) return go_app<5, 1, float, true>(S, ctlrow, transpose, x, y, nb, nrep, st);
    if (S.Q == 4) return go_app<4, 0, float, true>(S, ctlrow, transpose, x, y, nb, nrep, st);
    return hipErrorInvalidValue;
  }
  if (S.Q == 5) return go_app<5, 2, float>(S, ctlrow, transpose, x, y, nb, nrep, st);
  if (S.Q == 4) return sb == 2 ? go_app<4, 2, float>(S, ctlrow, transpose, x, y, nb, nrep, st) : go_app<4, 0, float>(S, ctlrow, transpose, x, y, nb, nrep, st);
  if (S.Q == 3) return go_app<3, 0, float>(S, ctlrow, transpose, x, y, nb, nrep, st);
  return hipErrorInvalidValue;
}

// fp64 instantiation of the lean slot kernel: the Neumann sweeps of the 2^5 Lindblad system in QD_PRECISION_F64 and [r4] of the 2^4
// one (the 4-qubit open system: forward sweep 2.70 -> 2.46 ms against the general kernel, gradient evaluation equal; [r5] two waves of
// two elements instead of four waves of one: 2.55 against 2.25 ms, not kept)
bool lean64_available(const DevSys& S, const TuneOpts& o) {
  // (dipole-dipole coupling [r5]: instantiations of their own - 2^4 one element per thread, 2^5 two; stationary iterations only)
  if (!S.lindblad || S.dense || (S.Q != 5 && S.Q != 4)) return false;
  for (int k = 0; k < S.Q; k++)
    if (S.n[k] != 2 || S.ness[k] != 2) return false;
  return !o.no_lean64;
}
// Small batches (at most one state per CU: the shards of a multi-GPU run, BASELINE config 5 on 8 GPUs) run two elements per thread on
// 512 threads - two waves per SIMD hide part of the latency one 256-thread group per CU leaves exposed: 128 states x 1000 steps,
// gradient, one lease: 16.1 -> 15.0 ms (forward 5.5 -> 5.3).  Larger batches keep two 256-thread groups per CU.
static int lean64_sb(const SweepArgs& a, const TuneOpts& o) {
  if (a.use_gmres) return 2;
  if (o.lean64_sb == 1 || o.lean64_sb == 2) return o.lean64_sb;
  return a.nb <= 256 ? 1 : 2;
}
hipError_t launch_forward_lean64(const SweepArgs& a, const TuneOpts& o, hipStream_t st) {
  // ([r6] the Krylov solver of these kernels - Team32::kry1 in front of the plain GMRES - also on the coupled stencils)
  if (a.S.Q == 4 && a.S.hasJ) return a.use_gmres ? go_fwd<4, 0, double, true, true>(a, st) : go_fwd<4, 0, double, false, true>(a, st);
  if (a.S.Q == 5 && a.S.hasJ) return a.use_gmres ? go_fwd<5, 1, double, true, true>(a, st) : go_fwd<5, 1, double, false, true>(a, st);
  if (a.S.Q == 4) return a.use_gmres ? go_fwd<4, 0, double, true>(a, st) : go_fwd<4, 0, double>(a, st);  // 2^4: one element per thread, four waves
  if (lean64_sb(a, o) == 1) return go_fwd<5, 1, double>(a, st);
  return a.use_gmres ? go_fwd<5, 2, double, true>(a, st) : go_fwd<5, 2, double>(a, st);
}
hipError_t launch_adjoint_lean64(const SweepArgs& a, const TuneOpts& o, hipStream_t st) {
  if (a.S.Q == 4 && a.S.hasJ) return a.use_gmres ? go_adj<4, 0, double, true, true>(a, st) : go_adj<4, 0, double, false, true>(a, st);
  if (a.S.Q == 5 && a.S.hasJ) return a.use_gmres ? go_adj<5, 1, double, true, true>(a, st) : go_adj<5, 1, double, false, true>(a, st);
  if (a.S.Q == 4) return a.use_gmres ? go_adj<4, 0, double, true>(a, st) : go_adj<4, 0, double>(a, st);
  if (lean64_sb(a, o) == 1) return go_adj<5, 1, double>(a, st);
  return a.use_gmres ? go_adj<5, 2, double, true>(a, st) : go_adj<5, 2, double>(a, st);
}
hipError_t launch_apply_lean64(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb, hipStream_t st) {
  if (S.Q == 4 && S.hasJ) return go_app<4, 0, double, true>(S, ctlrow, transpose, x, y, nb, 1, st);
  if (S.Q == 5 && S.hasJ) return go_app<5, 1, double, true>(S, ctlrow, transpose, x, y, nb, 1, st);
  if (S.Q == 4) return go_app<4, 0, double>(S, ctlrow, transpose, x, y, nb, 1, st);
  return go_app<5, 2, double>(S, ctlrow, transpose, x, y, nb, 1, st);
}

}  // namespace qd
