// qd_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4) for the Quandary hot path.
//
// Design: the batch of initial conditions is the data-parallel axis.  One
// workgroup owns one initial condition for the WHOLE time loop (persistent over time): the state
// lives in registers (each thread owns EPT elements) and in LDS as interleaved complex numbers
// (one ds_read_b128 per stencil neighbour), the linear-solver iterations run inside the kernel,
// and HBM sees only the stored trajectory (Lindblad adjoint) and the final state.  Controls are
// streamed from a precomputed per-sub-step table through scalar loads.
//
// Reference semantics restated here (paths relative to the reference repository):
//   stencil            include/mastereq.hpp:316-912, src/mastereq.cpp:1464-1709 (generic in Q, runtime levels)
//   gradient coeffs    include/mastereq.hpp:553-604, src/mastereq.cpp:970-1276
//   IMR fwd / bwd      src/timestepper.cpp:584-694, Neumann :697-727
//   time loops         src/timestepper.cpp:96-253, penalties :256-480
//   controls           src/oscillator.cpp:281-381, src/controlbasis.cpp:48-96,230-254
//   objective / seeds  src/optimtarget.cpp:343-447, :712-897
#include <hip/hip_runtime.h>

#include "qd_device.h"
#include "qd_big.h"

namespace qd {

// ---------------------------------------------------------------------------------------------
// controls: Oscillator::evalControl for every table row (src/oscillator.cpp:281-337)
// ---------------------------------------------------------------------------------------------
__device__ inline double bspline2(const DevSeg& g, int id, double t) {  // controlbasis.cpp:81-96
  const double tc = g.tstart + g.dtknot * ((id + 1) - 1.5);
  const double tau = (t - tc) / g.width;
  if (tau < -1. / 2. || tau >= 1. / 2.) return 0.0;
  if (tau < -1. / 6.) return 9. / 8. + 9. / 2. * tau + 9. / 2. * tau * tau;
  if (tau < 1. / 6.) return 3. / 4. - 9. * tau * tau;
  return 9. / 8. - 9. / 2. * tau + 9. / 2. * tau * tau;
}

// getRampFactor / getRampFactor_diff, src/util.cpp:92-147 (piecewise-linear ramp up / plateau / ramp down; the branch order of
// the reference decides the value where the pieces meet)
__device__ inline double ramp_factor(double t, double t0, double t1, double tramp) {
  double r = 0.0;
  if (t <= t0 + tramp) r = 1.0 / tramp * t - t0 / tramp;
  else if (t0 + tramp <= t && t <= t1 - tramp) r = 1.0;
  else if (t >= t1 - tramp && t <= t1) r = -1.0 / tramp * t + t1 / tramp;
  if (t1 < t0 + 2 * tramp) r = 0.0;
  return r;
}
__device__ inline double ramp_factor_diff(double t, double t0, double t1, double tramp) {
  double d = 0.0;
  if (t <= t0 + tramp) d = 0.0;
  else if (t0 + tramp <= t && t <= t1 - tramp) d = 0.0;
  else if (t >= t1 - tramp && t <= t1) d = 1.0 / tramp;
  if (t1 < t0 + 2 * tramp) d = 0.0;
  return d;
}

__device__ inline void eval_control_dev(const DevCtlDesc& d, const double* __restrict__ params, int k, double t, double& p, double& q) {
  const DevOsc o = d.oscs[k];
  p = 0.0;
  q = 0.0;
  if (o.nparams > 0) {
    const double* coeff = params + o.offset;
    for (int bs = 0; bs < o.nseg; bs++) {
      const DevSeg g = d.segs[o.seg_begin + bs];
      if (g.tstart <= t && g.tstop >= t) {
        double sp = 0.0, sq = 0.0;
        for (int f = 0; f < o.ncar; f++) {
          double b1 = 0.0, b2 = 0.0;
          const double* cf = coeff + g.skip + f * g.npc;
          const double om = d.carriers[o.car_begin + f];
          if (g.type == QD_CTRL_STEP) {  // Step::evaluate, controlbasis.cpp:195-206 (index skip + 2*carrier, one carrier)
            const double tend = g.tstart + coeff[g.skip + 2 * f] * (g.tstop - g.tstart);
            const double ramp = g.a3 > 1e-13 ? ramp_factor(t, g.tstart, tend, g.a3) : 1.0;
            b1 = ramp * g.a1;
            b2 = ramp * g.a2;
          } else if (g.type == QD_CTRL_BSPLINEAMP) {  // BSpline2ndAmplitude::evaluate :127-141, oscillator.cpp:308-312
            const int lc = (int)floor((t - g.tstart) / g.dtknot);
            const int l0 = max(0, lc - 1), l1 = min(g.nsplines - 1, lc + 3);
            for (int l = l0; l <= l1; l++) {
              if (d.enforce_bc && (l <= 1 || l >= g.nsplines - 2)) continue;
              b1 += cf[l] * bspline2(g, l, t);
            }
            const double ph = g.a1 * cf[g.nsplines];
            sp += cos(om * t + ph) * b1;
            sq += sin(om * t + ph) * b1;
            continue;
          } else if (g.type == QD_CTRL_BSPLINE) {
            // only the splines whose support can contain t (the reference loops over all of them, the others
            // contribute exact zeros: controlbasis.cpp:48-66, :81-96); one spline of margin on each side
            const int lc = (int)floor((t - g.tstart) / g.dtknot);
            const int l0 = max(0, lc - 1), l1 = min(g.nsplines - 1, lc + 3);
            for (int l = l0; l <= l1; l++) {
              if (d.enforce_bc && (l <= 1 || l >= g.nsplines - 2)) continue;
              const double B = bspline2(g, l, t);
              b1 += cf[l] * B;
              b2 += cf[l + g.nsplines] * B;
            }
          } else {
            const int id = (int)ceil((t - g.tstart) / g.dtknot - 0.5);
            if (id >= 0 && id < g.nsplines) {
              b1 = cf[id];
              b2 = cf[id + g.nsplines];
            }
          }
          const double co = cos(om * t), si = sin(om * t);
          sp += co * b1 - si * b2;
          sq += si * b1 + co * b2;
        }
        p = sp;
        q = sq;
        break;
      }
    }
  }
  for (int i = 0; i < o.npulse; i++) {
    const double* pu = d.pulses + (size_t)(o.pulse_begin + i) * 3;
    if (pu[0] <= t && t <= pu[1]) {
      p = pu[2] / sqrt(2.0);
      q = p;
    }
  }
}

// Two row sets in one launch (step table + energy-penalty table of one parameter update); `zero_me`
// (optional) is the RHS-application counter of the sweep that follows.
__global__ void k_controls(const DevCtlDesc d, const double* __restrict__ params, const double* __restrict__ times,
                           const double* __restrict__ hs, int nrows, double* __restrict__ table, const double* __restrict__ times2,
                           const double* __restrict__ hs2, int nrows2, double* __restrict__ table2, int cs,
                           unsigned long long* __restrict__ zero_me) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == 0 && zero_me) *zero_me = 0ull;
  int row = idx / d.Q;
  const int k = idx % d.Q;
  if (row >= nrows + nrows2) return;
  if (row >= nrows) {  // second set
    row -= nrows;
    times = times2;
    hs = hs2;
    table = table2;
  }
  const double t = times[row];
  double p, q;
  eval_control_dev(d, params, k, t, p, q);
  double* r = table + (size_t)row * cs;
  r[2 + k] = p;
  r[2 + d.Q + k] = q;
  if (k == 0) {
    r[0] = hs[row];
    r[1] = t;
    for (int i = 0; i < d.npairs; i++) {  // MasterEq::assemble_RHS, mastereq.cpp:671-675
      r[2 + 2 * d.Q + i] = cos(d.eta[i] * t);
      r[2 + 2 * d.Q + d.npairs + i] = sin(d.eta[i] * t);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// final-time objective and adjoint seed (one block per initial condition, runtime Q)
// ---------------------------------------------------------------------------------------------
// grid (nb, P): P > 1 (one large state, launch_objective) - every block writes its share to part[b][p][4], k_objective_sum adds the
// shares in a fixed order
template <bool LIND>
__global__ void k_objective(const DevSys S, const DevTarget tg, const double* __restrict__ x, double* __restrict__ out4, double* __restrict__ part) {
  __shared__ double red[4 * 16];
  const int b = blockIdx.x, dim = S.dim, P = gridDim.y;
  const double* xs = x + (size_t)b * 2 * dim;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int it = blockIdx.y * blockDim.x + threadIdx.x; it < dim; it += blockDim.x * P) {
    const double2 xv = make_double2(xs[it], xs[dim + it]);
    evalJ_part<LIND>(S, tg, b, it, xv, v[0], v[1]);
    fidelity_part<LIND>(S, tg, b, it, xv, v[2], v[3]);
  }
  block_sum<4, false>(v, red);
  if (threadIdx.x < 4) (P > 1 ? part + ((size_t)b * P + blockIdx.y) * 4 : out4 + (size_t)b * 4)[threadIdx.x] = v[threadIdx.x];
}

__global__ void k_objective_sum(const double* __restrict__ part, int P, double* __restrict__ out4) {
  const int b = blockIdx.x;
  if (threadIdx.x < 4) {
    double s = 0.0;
    for (int p = 0; p < P; p++) s += part[((size_t)b * P + p) * 4 + threadIdx.x];
    out4[(size_t)b * 4 + threadIdx.x] = s;
  }
}

template <bool LIND>
__global__ void k_seed(const DevSys S, const DevTarget tg, const double* __restrict__ x, const double* __restrict__ rbib,
                       double* __restrict__ xbar) {
  const int b = blockIdx.x, dim = S.dim;
  const double* xs = x + (size_t)b * 2 * dim;
  double* xo = xbar + (size_t)b * 2 * dim;
  const double rbar = rbib[2 * b], ibar = rbib[2 * b + 1];
  for (int it = blockIdx.y * blockDim.x + threadIdx.x; it < dim; it += blockDim.x * gridDim.y) {
    double2 xb = make_double2(0.0, 0.0);
    evalJ_diff_elem<LIND>(S, tg, b, it, make_double2(xs[it], xs[dim + it]), xb, rbar, ibar);
    xo[it] = xb.x;
    xo[dim + it] = xb.y;
  }
}

// coeff[nb][ncol] -> sum over the batch in a fixed order (deterministic gradient)
__global__ void k_reduce_coeff(const double* __restrict__ coeff, int nb, int ncol, double* __restrict__ sum, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncol) return;
  double s = accumulate ? sum[i] : 0.0;
  for (int b = 0; b < nb; b++) s += coeff[(size_t)b * ncol + i];
  sum[i] = s;
}

// grad[d] = sum_s B_d(t_s) * {Blt1bar | Blt2bar}(s) + energy-penalty terms
// (Oscillator::evalControl_diff oscillator.cpp:339-381, BSpline2nd::derivative controlbasis.cpp:68-79,
//  BSpline0::derivative :245-254, energyPenaltyIntegral_diff timestepper.cpp:458-480)
__global__ void __launch_bounds__(64) k_grad(const DevCtlDesc d, const double* __restrict__ params, const double* __restrict__ table, int cs, int nsub, int ee,
                                             const double* __restrict__ coeffsum, const double* __restrict__ etable, int nstep,
                                             double ebar, double* __restrict__ grad, int ndesign) {
  // one wave per design parameter; lanes stride over the sub-steps, fixed-order wave reduction
  const int idx = blockIdx.x, lane = threadIdx.x;
  if (idx >= ndesign) return;
  // locate (oscillator, segment, carrier, spline, part)
  int k = 0;
  while (k < d.Q - 1 && idx >= d.oscs[k].offset + d.oscs[k].nparams) k++;
  const DevOsc o = d.oscs[k];
  const int loc = idx - o.offset;
  int gs = 0;
  for (int bs = 0; bs < o.nseg; bs++) {
    const DevSeg g = d.segs[o.seg_begin + bs];
    if (loc >= g.skip && loc < g.skip + g.npc * o.ncar) gs = bs;
  }
  const DevSeg g = d.segs[o.seg_begin + gs];
  const int r = loc - g.skip;
  const int f = r / g.npc, l = (r % g.npc) % g.nsplines, part = (r % g.npc) / g.nsplines;
  if ((g.type == QD_CTRL_BSPLINE && d.enforce_bc && (l <= 1 || l >= g.nsplines - 2)) || g.type == QD_CTRL_BSPLINEAMP) {
    if (lane == 0) grad[idx] = 0.0;  // (spline_amplitude: no gradient in the reference, rejected before this launch)
    return;
  }
  const bool step = g.type == QD_CTRL_STEP;
  // Step::derivative, controlbasis.cpp:208-216: d ramp / d t_stepend * (tstop - tstart), t_stepend = tstart + alpha (tstop - tstart)
  const double tend = step ? g.tstart + params[o.offset + g.skip + 2 * f] * (g.tstop - g.tstart) : 0.0;
  const double om = d.carriers[o.car_begin + f];
  auto active = [&](double t) {  // this segment is the FIRST one containing t (oscillator.cpp:344-346 + break)
    if (!(g.tstart <= t && g.tstop >= t)) return false;
    for (int bs = 0; bs < gs; bs++) {
      const DevSeg g2 = d.segs[o.seg_begin + bs];
      if (g2.tstart <= t && g2.tstop >= t) return false;
    }
    return true;
  };
  auto basis = [&](double t) {
    if (step) return ramp_factor_diff(t, g.tstart, tend, g.a3) * (g.tstop - g.tstart);
    if (g.type == QD_CTRL_BSPLINE) return bspline2(g, l, t);
    const int id = (int)ceil((t - g.tstart) / g.dtknot - 0.5);
    return id == l ? 1.0 : 0.0;
  };
  double acc = 0.0;
  for (int s = lane; s < nsub; s += 64) {
    const double* row = table + (size_t)s * cs;
    const double t = ee ? row[1] + row[0] : row[1];
    if (!active(t)) continue;
    const double B = basis(t);
    if (B == 0.0) continue;
    const double pbar = coeffsum[(size_t)s * 2 * d.Q + 2 * k], qbar = coeffsum[(size_t)s * 2 * d.Q + 2 * k + 1];
    const double co = cos(om * t), si = sin(om * t);
    const double b1bar = si * qbar + co * pbar, b2bar = co * qbar - si * pbar;
    acc += B * (step ? g.a1 * b1bar + g.a2 * b2bar : part == 0 ? b1bar : b2bar);
  }
  if (ebar != 0.0) {
    for (int n = lane; n < nstep; n += 64) {
      const double* row = etable + (size_t)n * cs;
      const double t = row[1];
      if (!active(t)) continue;
      const double B = basis(t);
      if (B == 0.0) continue;
      const double pbar = ebar / nstep * 2.0 * row[2 + k], qbar = ebar / nstep * 2.0 * row[2 + d.Q + k];
      const double co = cos(om * t), si = sin(om * t);
      const double b1bar = si * qbar + co * pbar, b2bar = co * qbar - si * pbar;
      acc += B * (step ? g.a1 * b1bar + g.a2 * b2bar : part == 0 ? b1bar : b2bar);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) grad[idx] = acc;
}

// ---------------------------------------------------------------------------------------------
// observables of the stored trajectory (Oscillator::expectedEnergy / population src/oscillator.cpp:430-566,
// MasterEq::expectedEnergy / population src/mastereq.cpp:2897-2974): one workgroup per (output step, initial condition).
// P(I) = rho_II (Lindblad) or |psi_I|^2; population_k[l] = sum over I with digit_k(I) = l (one thread per (k, l), ascending I:
// the summation order of the reference's loop), expected_k = sum_I digit_k(I) P(I); composite: sum_I I P(I) and P itself.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_observables(const DevSys S, const double* __restrict__ traj, int f32, int nb, int nstages, int stride,
                                                     int nlev_total, double* __restrict__ expected, double* __restrict__ population,
                                                     double* __restrict__ expcomp, double* __restrict__ popcomp) {
  __shared__ double red[2 * NRED * 4];
  const int o = blockIdx.y, b = blockIdx.x, N = S.N, dim = S.dim;
  const size_t state = (size_t)o * stride * nstages * nb + b;
  auto prob = [&](int I) -> double {
    const int e = S.lindblad ? I + I * N : I;
    if (f32) {
      const float2 v = reinterpret_cast<const float2*>(traj)[state * dim + e];
      return S.lindblad ? (double)v.x : (double)v.x * v.x + (double)v.y * v.y;
    }
    const double u = traj[state * 2 * dim + e];
    if (S.lindblad) return u;
    const double v = traj[state * 2 * dim + dim + e];
    return u * u + v * v;
  };
  const size_t ob = (size_t)o * nb + b;
  if (popcomp)
    for (int I = threadIdx.x; I < N; I += blockDim.x) popcomp[ob * N + I] = prob(I);
  if (population || expected) {
    for (int t = threadIdx.x; t < nlev_total; t += blockDim.x) {  // thread -> (oscillator k, level l)
      int k = 0, l = t;
      while (l >= S.n[k]) { l -= S.n[k]; k++; }
      const int post = S.post[k], nk = S.n[k];
      double s = 0.0;
      for (int hi = 0; hi < N / (nk * post); hi++)       // I = hi nk post + l post + lo, ascending
        for (int lo = 0; lo < post; lo++) s += prob((hi * nk + l) * post + lo);
      if (population) population[ob * nlev_total + t] = s;
    }
  }
  if (expected || expcomp) {
    double v[NRED];
#pragma unroll
    for (int i = 0; i < NRED; i++) v[i] = 0.0;
    for (int I = threadIdx.x; I < N; I += blockDim.x) {
      const double p = prob(I);
      for (int k = 0; k < S.Q; k++) v[k] += ((I / S.post[k]) % S.n[k]) * p;
      v[QD_MAX_OSC] += I * p;
    }
    block_sum<NRED, false>(v, red);
    if (threadIdx.x == 0) {
      if (expected)
        for (int k = 0; k < S.Q; k++) expected[ob * S.Q + k] = v[k];
      if (expcomp) expcomp[ob] = v[QD_MAX_OSC];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// objective level on the device (multi-GPU path: nothing returns to the host between the sweeps and the collectives)
// ---------------------------------------------------------------------------------------------
// The seven partial sums the reference all-reduces over comm_init (src/optimproblem.cpp:258-298) from the per-state
// results of the forward sweep `res` = [pen nb | dpdm nb | (J_re, J_im, fid_re, fid_im) nb]; the energy penalty
// (src/timestepper.cpp:444-455) from the table of controls at the step ends.  One workgroup, fixed summation order.
__global__ void __launch_bounds__(256) k_partial_sums(const double* __restrict__ res, int nb, const double* __restrict__ w, double inv_ninit,
                                                      qd_penalty pen, const double* __restrict__ etable, int cs, int Q, int nstep,
                                                      double* __restrict__ sums) {
  __shared__ double red[8][4];
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const double* pn = res;
  const double* dp = res + nb;
  const double* o4 = res + 2 * (size_t)nb;
  for (int i = threadIdx.x; i < nb; i += 256) {
    const double wi = w[i];
    if (pen.gamma_penalty > 1e-13) v[QD_SUM_PENALTY] += wi * pen.gamma_penalty * pn[i];
    if (pen.gamma_penalty_dpdm > 1e-13) v[QD_SUM_DPDM] += wi * pen.gamma_penalty_dpdm * dp[i];
    v[7] += wi;  // the energy integral is state independent: weight sum x gamma x E
    v[QD_SUM_COST_RE] += wi * o4[4 * i];
    v[QD_SUM_COST_IM] += wi * o4[4 * i + 1];
    v[QD_SUM_FID_RE] += inv_ninit * o4[4 * i + 2];
    v[QD_SUM_FID_IM] += inv_ninit * o4[4 * i + 3];
  }
  double e = 0.0;
  if (pen.gamma_penalty_energy > 1e-13)
    for (int n = threadIdx.x; n < nstep; n += 256) {
      const double* row = etable + (size_t)n * cs;
      double a = 0.0;
      for (int k = 0; k < Q; k++) a += row[2 + k] * row[2 + k] + row[2 + Q + k] * row[2 + Q + k];
      e += a / nstep;
    }
  v[QD_SUM_ENERGY] = e;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    v[i] = wave_sum(v[i]);
    if (lane == 0) red[i][wave] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int i = threadIdx.x;
    red[i][0] = (red[i][0] + red[i][1]) + (red[i][2] + red[i][3]);
  }
  __syncthreads();
  if (threadIdx.x < QD_NSUMS) {
    const int i = threadIdx.x;
    sums[i] = i == QD_SUM_ENERGY ? pen.gamma_penalty_energy * red[7][0] * red[QD_SUM_ENERGY][0] : red[i][0];
  }
}

// adjoint seed weights beta_i * finalizeJ_diff(GLOBAL cost) (src/optimproblem.cpp:433-436, :508-511; finalizeJ_diff
// src/optimtarget.cpp:880-897): only Schroedinger + Jtrace depends on the reduced sums
__global__ void k_seed_weights(const double* __restrict__ sums, const double* __restrict__ w, int nb, int objective_type, int lindblad,
                               double* __restrict__ rbib) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  double rb = 1.0, ib = 0.0;
  if (objective_type == QD_OBJ_JTRACE) {
    if (lindblad) rb = -1.0;
    else { rb = -2.0 * sums[QD_SUM_COST_RE]; ib = -2.0 * sums[QD_SUM_COST_IM]; }
  }
  rbib[2 * i] = w[i] * rb;
  rbib[2 * i + 1] = w[i] * ib;
}

// ---------------------------------------------------------------------------------------------
// dense user-Hamiltonian path: G(t_row) = -i Hsys + sum_k q_k Im(Hc_k) - i p_k Re(Hc_k) for every row of
// the control table (Re = Ad + sum q_k Ac_k, Im = Bd + sum p_k Bc_k with Ac = Im(Hc), Bc = -Re(Hc):
// src/mastereq.cpp:760-795, src/hamiltonianfilereader.cpp:77-84,170-176)
// ---------------------------------------------------------------------------------------------
__global__ void k_gmat(const DevSys S, const double* __restrict__ g0, const double* __restrict__ table, int cs, int nrows,
                       double* __restrict__ gtab) {
  const int nn = S.N * S.N;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)nrows * nn) return;
  const int row = (int)(e / nn), el = (int)(e % nn);
  const double* r = table + (size_t)row * cs;
  double re = g0[2 * el], im = g0[2 * el + 1];
  for (int k = 0; k < S.Q; k++) {
    re = fma(r[2 + S.Q + k], S.hci[(size_t)k * nn + el], re);
    im = fma(-r[2 + k], S.hcr[(size_t)k * nn + el], im);
  }
  gtab[2 * e] = re;
  gtab[2 * e + 1] = im;
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
// Variant choice.  One workgroup per initial condition.  Few initial conditions (latency regime):
// spread one state over as many lanes as it has elements.  Many initial conditions (throughput
// regime): more elements per thread so that several workgroups share a CU.
constexpr int QD_COL_DEFAULT = 9;

// host-side view of the kernel variants, generated from the Variant<> traits of qd_device.h
struct VarInfo {
  int ept, icpb, maxb;
  bool dbuf, col;
};
template <int V>
constexpr VarInfo var_info() {
  return {Variant<V>::EPT, Variant<V>::ICPB, Variant<V>::MAXB, Variant<V>::DBUF, Variant<V>::COL};
}
static const VarInfo kVar[NVARIANTS] = {var_info<0>(), var_info<1>(), var_info<2>(),  var_info<3>(),  var_info<4>(),
                                        var_info<5>(), var_info<6>(), var_info<7>(),  var_info<8>(),  var_info<9>(),
                                        var_info<10>(), var_info<11>(), var_info<12>(), var_info<13>(), var_info<14>(),
                                        var_info<15>(), var_info<16>(), var_info<17>()};
int variant_max_block(int var) { return (var >= 0 && var < NVARIANTS) ? kVar[var].maxb : 0; }

void big_team(const DevSys& S, int nb, const TuneOpts& o, int& team, int& spread);
LaunchCfg pick_config(const DevSys& S, int nb, const TuneOpts& o, bool want_gmres, bool adjoint) {
  LaunchCfg c{};
  const int dim = S.dim;
  bool qubit = true;
  for (int k = 0; k < S.Q; k++) qubit = qubit && S.n[k] == 2 && S.ness[k] == 2;
  if (S.dense) qubit = false;
  if (S.lindblad && S.Q > 5) qubit = false;  // (the all-qubit Lindblad stencils are instantiated for 1..5 oscillators: general stencil)
  c.qubit = S.dense ? 2 : qubit ? 1 : 0;
  c.noplain = o.no_plain;
  const bool gm = want_gmres && !o.force_neumann;
  // column layout (V8/V9): one wave per column of rho, N <= 64 lanes used
  // V14 packs floor(64 / N) columns into one wave slot
  auto colblock = [&](int v) {
    const int cpw = (v == 14 && S.N <= 64) ? 64 / S.N : 1, slots = (S.N + cpw - 1) / cpw;
    return 64 * ((slots + kVar[v].ept - 1) / kVar[v].ept);
  };
  auto fits = [&](int v) {
    if (v == 14 && S.N > 32) return false;
    if (kVar[v].col) return S.lindblad && !qubit && S.N <= 64 && colblock(v) <= kVar[v].maxb && lds_bytes(S, colblock(v), true, false, 2, 1, true) <= 160 * 1024;
    return (dim + (kVar[v].ept / kVar[v].icpb) - 1) / (kVar[v].ept / kVar[v].icpb) <= kVar[v].maxb;
  };
  auto built = [&](int v) {  // mirrors variant_built() in qd_inst.hip
    if (S.dense) return (v >= 11 && v <= 13) || (v == 15 && S.lindblad && S.N == 16) || (v == 17 && S.lindblad && S.N > 16 && S.N <= 32);
    if (S.lindblad && S.Q > 5) return v == 4 && S.Q == 6;
    if (!qubit) return v <= 2 || v == 4 || (S.lindblad && (v == 9 || v == 14));
    return dim <= 64 ? v == 0 : dim <= 256 ? v == 1 : v == 2;
  };
  int var;
  if (dim <= 64) var = 0;
  else if (dim <= 256) var = 1;
  else if (dim <= 1024) var = fits(14) ? 14 : 2;  // packed column layout for non-qubit Lindblad (3x3x3: 18.7M vs 12.4M units/s)
  // column layout when most of its 64 lanes (= rows) are used; measured: N = 36 V4 8.6M vs V9 7.2M units/s,
  // N = 49 4.3M vs 6.7M, N = 64 3.3M vs 5.2M
  // ... the lean column kernels (qd_col.hip) already from N = 33 (measured below), the general column kernel from N = 44
  else if (S.lindblad && S.Q > 5) var = 4;
  else var = (fits(QD_COL_DEFAULT) && (S.N >= 44 || (!gm && S.N >= o.col_min_n && collean_available(S, o)))) ? QD_COL_DEFAULT : 4;
  if (S.dense) var = dim <= 64 ? 11 : dim <= 256 ? 12 : 13;  // qd_set_hamiltonian limits dim to 1024
  // matrix cores for the dense operator and (adjoint sweep) for the gradient contraction's 2Q commutators per step; the option no_mfma
  // keeps the vector kernels (measurements)
  if (S.dense && S.lindblad && S.N == 16 && !o.no_mfma) var = 15;
  // (zero-padded 32 x 32 tiles pay the full 32^3 products: worth it from N = 22 on, 96.8 ms x (N / 32)^3 against 23.7 ms)
  if (S.dense && S.lindblad && S.N >= 22 && S.N <= 32 && !o.no_mfma) var = 17;
  if (o.var >= 0) {  // tuning override
    const int v = o.var;
    if (v >= 0 && v < NVARIANTS && built(v) && fits(v)) var = v;
  }
  // beyond one CU's LDS: work vectors in global memory (qd_big.h); general stencil whatever the level structure.
  // the option var = 16 forces it onto small systems (parity tests of this path against everything the LDS kernels are tested on)
  if ((S.dense ? dim > 1024 : dim > 4096) || o.var == 16) {
    c.var = 16;
    c.qubit = S.dense ? 2 : 0;
    c.block = BIG_BLOCK;
    c.gmres = gm ? 2 : 0;
    c.lds = BigTeam<1, false>::lds_bytes(S);
    big_team(S, nb, o, c.team, c.spread);
    c.blocked = o.big_blocked;
    return c;
  }
  const int epe = kVar[var].ept / kVar[var].icpb;
  c.var = var;
  c.block = kVar[var].col ? colblock(var) : ((dim + epe - 1) / epe + 63) / 64 * 64;
  if (var == 17) c.block = 256;  // four waves = the four 16 x 16 tiles, whatever N
  c.gmres = 0;
  const bool dn = S.dense == 2;  // G(t) staged in LDS
  c.lds = lds_bytes(S, c.block, kVar[var].dbuf, false, 0, kVar[var].icpb, kVar[var].col, dn);
  if (gm && kVar[var].icpb == 1) {
    const size_t in_lds = lds_bytes(S, c.block, kVar[var].dbuf, false, 1, kVar[var].icpb, kVar[var].col, dn);
    if (kVar[var].ept == 1 && in_lds <= 160 * 1024) {  // Krylov basis in LDS
      c.gmres = 1;
      c.lds = in_lds;
    } else {  // Krylov basis in global memory
      c.gmres = 2;
      c.lds = lds_bytes(S, c.block, kVar[var].dbuf, false, 2, kVar[var].icpb, kVar[var].col, dn);
    }
  }
  return c;
}

// Workgroups per initial condition of the global-memory sweeps.  A team barrier costs 2 us (8 members) to 3 us (64) and there
// are about two per operator application, one workgroup alone needs 2-17 ns per element and application: the team grows while it
// leaves each thread at least half an element, up to 64 members, and while all teams stay resident (one 1024-thread workgroup per
// CU is what the register budget of these kernels allows for sure; the cooperative launch checks it).  Members dealt over all XCDs
// by default.  the options big_team / big_spread override (tests, measurements: profiles/big_probe.py).
void big_team(const DevSys& S, int nb, const TuneOpts& o, int& team, int& spread) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 256;
  }
  spread = o.big_spread >= 0 ? o.big_spread != 0 : 1;
  const int gmax = spread ? BIG_TEAM_MAX : 32;
  const int slots = spread ? nb : (nb + 7) / 8 * 8;
  int g = 1;
  while (g * 2 <= 64 && g * 2 <= gmax && (long)slots * (g * 2) <= ncu && (size_t)2 * S.dim >= (size_t)BIG_BLOCK * (g * 2)) g *= 2;
  // very large states (the reference's nlevels_32_32_32_32: dim 2^20): beyond 64 members while every thread keeps at least four
  // elements - measured on that case (GMRES, 50 steps): 64 members 118 ms, 128: 75 ms, 256: 65 ms (20 x 20, dim 160 000: 11.3 / 11.3 / 12.1)
  // ... and 128 members from one element per thread on: 20 x 20 Lindblad with batched neighbour reads, 100 steps: 64 members 9.4 ms forward /
  // 27.4 ms gradient, 128: 9.1 / 25.0, 256: 11.6 / 29.7
  while (g >= 64 && g * 2 <= gmax && (long)slots * (g * 2) <= ncu && (size_t)S.dim >= (size_t)(g == 64 ? 1 : 4) * BIG_BLOCK * (g * 2)) g *= 2;
  if (S.dim <= 4096) g = 1;  // (the global-memory kernels forced onto a small system)
  if (o.big_team > 0) {
    const int v = o.big_team;
    if (v >= 1 && v <= gmax && (v & (v - 1)) == 0 && (long)slots * v <= ncu) g = v;
  }
  team = g;
}

size_t big_work_doubles(const DevSys& S, int nb) { return (size_t)nb * BIG_NV * 2 * (size_t)S.dim; }

size_t krylov_doubles(const DevSys& S, int nb) { return (size_t)nb * (GMRES_MR_G + 2) * 2 * (size_t)S.dim; }

// ---------------------------------------------------------------------------------------------
// options
// ---------------------------------------------------------------------------------------------
static const char* const kOptKeys[] = {"var", "force_neumann", "no_mfma", "big_team", "big_spread", "big_blocked", "f32_sb", "lean64_sb", "no_lean64", "no_collean",
                                       "col_ept", "col_slices", "no_plain", "col_min_n", "gmres_poly", "gmres_split", "neumann_split", "traj_budget_mb", "standin_tau", "sched_wait_s", "col_skip", "no_col_krylov", "krylov_tau", "krylov_restart"};
int TuneOpts::set(const char* key, const char* value) {
  if (!key || !value) return -1;
  const std::string k(key), v(value);
  char* end = nullptr;
  if (k == "traj_budget_mb") {
    const double d = strtod(value, &end);
    if (end == value || d < 0.0) return -1;
    traj_budget_mb = d;
    return 0;
  }
  if (k == "sched_wait_s") {
    const double d = v == "auto" ? 0.0 : strtod(value, &end);
    if ((v != "auto" && end == value) || d < 0.0) return -1;
    sched_wait_s = d;
    return 0;
  }
  if (k == "krylov_tau") {
    const double d = v == "auto" ? 0.1 : strtod(value, &end);
    if ((v != "auto" && end == value) || !(d > 0.0) || d > 1.0) return -1;
    krylov_tau = d;
    return 0;
  }
  if (k == "standin_tau") {
    const double d = v == "auto" ? 1e-3 : strtod(value, &end);
    if ((v != "auto" && end == value) || d < 0.0 || d > 1.0) return -1;
    standin_tau = d;
    return 0;
  }
  long iv;
  if (v == "auto") iv = (k == "var" || k == "big_spread" || k == "f32_sb" || k == "neumann_split" || k == "gmres_split") ? -1 : 0;
  else {
    iv = strtol(value, &end, 10);
    if (end == value) return -1;
  }
  if (k == "var") var = (int)iv;
  else if (k == "force_neumann") force_neumann = iv != 0;
  else if (k == "no_mfma") no_mfma = iv != 0;
  else if (k == "big_team") big_team = (int)iv;
  else if (k == "big_spread") big_spread = (int)iv;
  else if (k == "big_blocked") big_blocked = v == "auto" ? 2 : (int)(iv < 0 ? 0 : iv > 2 ? 2 : iv);
  else if (k == "f32_sb") f32_sb = (int)iv;
  else if (k == "lean64_sb") lean64_sb = (iv == 1 || iv == 2) ? (int)iv : 0;
  else if (k == "no_lean64") no_lean64 = iv != 0;
  else if (k == "col_skip") col_skip = v == "auto" ? 1 : iv != 0;
  else if (k == "no_collean") no_collean = iv != 0;
  else if (k == "no_col_krylov") no_col_krylov = iv != 0;
  else if (k == "krylov_restart") krylov_restart = (v == "auto" || iv < 1 || iv > 14) ? 14 : (int)iv;
  else if (k == "col_ept") col_ept = (int)iv;
  else if (k == "no_plain") no_plain = (int)(iv & 3);
  else if (k == "col_slices") col_slices = iv > 0 ? (int)iv : 0;
  else if (k == "col_min_n") col_min_n = iv > 0 ? (int)iv : 33;
  else if (k == "gmres_poly") gmres_poly = iv > 0 ? (int)iv : 0;
  else if (k == "neumann_split") neumann_split = iv < 0 ? -1 : iv != 0;
  else if (k == "gmres_split") gmres_split = iv < 0 ? -1 : iv != 0;
  else return -1;
  return 0;
}
void TuneOpts::load_env() {
  for (const char* key : kOptKeys) {
    std::string name = "QD_";
    for (const char* c = key; *c; c++) name += (char)toupper((unsigned char)*c);
    if (const char* ev = getenv(name.c_str())) (void)set(key, ev);
  }
}

// ---------------------------------------------------------------------------------------------
// dispatch to the per-(Q, Lindblad, qubit) translation units (qd_inst.hip)
// ---------------------------------------------------------------------------------------------
#define QD_DECL(q, l, b)                                                                                               \
  hipError_t inst_forward_##q##_##l##_##b(const SweepArgs&, const LaunchCfg&, hipStream_t);                             \
  hipError_t inst_adjoint_##q##_##l##_##b(const SweepArgs&, const LaunchCfg&, hipStream_t);                             \
  hipError_t inst_apply_##q##_##l##_##b(const DevSys&, const double*, int, const double*, double*, int, const LaunchCfg&, hipStream_t);
#define QD_DECL_Q(l, b) QD_DECL(1, l, b) QD_DECL(2, l, b) QD_DECL(3, l, b) QD_DECL(4, l, b) QD_DECL(5, l, b)
QD_DECL_Q(0, 0) QD_DECL_Q(1, 0) QD_DECL_Q(0, 1) QD_DECL_Q(1, 1)
// Schroedinger only: 6..8 oscillators (beyond the reference's matrix-free templates, which stop at 5)
QD_DECL(6, 0, 0) QD_DECL(7, 0, 0) QD_DECL(8, 0, 0) QD_DECL(6, 0, 1) QD_DECL(7, 0, 1) QD_DECL(8, 0, 1)
// ... and Lindblad on the general stencil [r5]
QD_DECL(6, 1, 0) QD_DECL(7, 1, 0) QD_DECL(8, 1, 0)
// dense user-Hamiltonian operator
QD_DECL_Q(0, 2) QD_DECL_Q(1, 2)
QD_DECL(6, 0, 2) QD_DECL(7, 0, 2) QD_DECL(8, 0, 2) QD_DECL(6, 1, 2) QD_DECL(7, 1, 2) QD_DECL(8, 1, 2)

#define QD_DECLT(q, l) hipError_t inst_bigtable_##q##_##l##_0(const DevSys&, double*, unsigned*, hipStream_t);
QD_DECLT(1, 0) QD_DECLT(2, 0) QD_DECLT(3, 0) QD_DECLT(4, 0) QD_DECLT(5, 0) QD_DECLT(6, 0) QD_DECLT(7, 0) QD_DECLT(8, 0)
QD_DECLT(1, 1) QD_DECLT(2, 1) QD_DECLT(3, 1) QD_DECLT(4, 1) QD_DECLT(5, 1) QD_DECLT(6, 1) QD_DECLT(7, 1) QD_DECLT(8, 1)
typedef hipError_t (*table_fn)(const DevSys&, double*, unsigned*, hipStream_t);
static const table_fn big_tab[2][8] = {{inst_bigtable_1_0_0, inst_bigtable_2_0_0, inst_bigtable_3_0_0, inst_bigtable_4_0_0, inst_bigtable_5_0_0,
                                        inst_bigtable_6_0_0, inst_bigtable_7_0_0, inst_bigtable_8_0_0},
                                       {inst_bigtable_1_1_0, inst_bigtable_2_1_0, inst_bigtable_3_1_0, inst_bigtable_4_1_0, inst_bigtable_5_1_0,
                                        inst_bigtable_6_1_0, inst_bigtable_7_1_0, inst_bigtable_8_1_0}};
hipError_t launch_big_table(const DevSys& S, double* ecoef, unsigned* edig, hipStream_t st) {
  if (S.Q < 1 || S.Q > 8 || !big_tab[S.lindblad ? 1 : 0][S.Q - 1]) return hipErrorInvalidValue;
  return big_tab[S.lindblad ? 1 : 0][S.Q - 1](S, ecoef, edig, st);
}

typedef hipError_t (*sweep_fn)(const SweepArgs&, const LaunchCfg&, hipStream_t);
typedef hipError_t (*apply_fn)(const DevSys&, const double*, int, const double*, double*, int, const LaunchCfg&, hipStream_t);
#define QD_ROW(base, l, b) {base##1_##l##_##b, base##2_##l##_##b, base##3_##l##_##b, base##4_##l##_##b, base##5_##l##_##b, nullptr, nullptr, nullptr}
#define QD_ROW8(base, l, b) {base##1_##l##_##b, base##2_##l##_##b, base##3_##l##_##b, base##4_##l##_##b, base##5_##l##_##b, base##6_##l##_##b, base##7_##l##_##b, base##8_##l##_##b}
// index [qubit][lindblad][Q-1]
static const sweep_fn fwd_tab[3][2][8] = {{QD_ROW8(inst_forward_, 0, 0), QD_ROW8(inst_forward_, 1, 0)},
                                          {QD_ROW8(inst_forward_, 0, 1), QD_ROW(inst_forward_, 1, 1)},
                                          {QD_ROW8(inst_forward_, 0, 2), QD_ROW8(inst_forward_, 1, 2)}};
static const sweep_fn adj_tab[3][2][8] = {{QD_ROW8(inst_adjoint_, 0, 0), QD_ROW8(inst_adjoint_, 1, 0)},
                                          {QD_ROW8(inst_adjoint_, 0, 1), QD_ROW(inst_adjoint_, 1, 1)},
                                          {QD_ROW8(inst_adjoint_, 0, 2), QD_ROW8(inst_adjoint_, 1, 2)}};
static const apply_fn app_tab[3][2][8] = {{QD_ROW8(inst_apply_, 0, 0), QD_ROW8(inst_apply_, 1, 0)},
                                          {QD_ROW8(inst_apply_, 0, 1), QD_ROW(inst_apply_, 1, 1)},
                                          {QD_ROW8(inst_apply_, 0, 2), QD_ROW8(inst_apply_, 1, 2)}};

hipError_t launch_forward(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  if (a.S.Q < 1 || a.S.Q > 8 || !fwd_tab[cfg.qubit][a.S.lindblad ? 1 : 0][a.S.Q - 1]) return hipErrorInvalidValue;
  return fwd_tab[cfg.qubit][a.S.lindblad ? 1 : 0][a.S.Q - 1](a, cfg, st);
}
hipError_t launch_adjoint(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st) {
  if (a.S.Q < 1 || a.S.Q > 8 || !adj_tab[cfg.qubit][a.S.lindblad ? 1 : 0][a.S.Q - 1]) return hipErrorInvalidValue;
  return adj_tab[cfg.qubit][a.S.lindblad ? 1 : 0][a.S.Q - 1](a, cfg, st);
}
hipError_t launch_apply(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb,
                        const LaunchCfg& cfg, hipStream_t st) {
  if (S.Q < 1 || S.Q > 8 || !app_tab[cfg.qubit][S.lindblad ? 1 : 0][S.Q - 1]) return hipErrorInvalidValue;
  return app_tab[cfg.qubit][S.lindblad ? 1 : 0][S.Q - 1](S, ctlrow, transpose, x, y, nb, cfg, st);
}

hipError_t launch_controls(const DevCtlDesc& d, const double* params, const double* times, const double* hs, int nrows,
                           double* table, int cs, hipStream_t st) {
  return launch_controls2(d, params, times, hs, nrows, table, nullptr, nullptr, 0, nullptr, cs, nullptr, st);
}

hipError_t launch_controls2(const DevCtlDesc& d, const double* params, const double* times, const double* hs, int nrows,
                            double* table, const double* times2, const double* hs2, int nrows2, double* table2, int cs,
                            unsigned long long* zero_me, hipStream_t st) {
  const int total = (nrows + nrows2) * d.Q;
  if (total == 0 && !zero_me) return hipSuccess;
  hipLaunchKernelGGL(k_controls, dim3((total > 0 ? total + 127 : 128) / 128), dim3(128), 0, st, d, params, times, hs, nrows, table,
                     times2, hs2, nrows2, table2, cs, zero_me);
  return hipGetLastError();
}

hipError_t launch_gmat(const DevSys& S, const double* g0, const double* table, int cs, int nrows, double* gtab, hipStream_t st) {
  const size_t total = (size_t)nrows * S.N * S.N;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(k_gmat, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S, g0, table, cs, nrows, gtab);
  return hipGetLastError();
}

// blocks per initial condition of the objective / seed kernels: one, except for a few large states (the team reduction buffer of
// qd_big.h, idle between the sweeps, takes the shares: [nb][P][4] <= [nb][2][BIG_TEAM_MAX][BIG_RED_NV])
static int objective_blocks(const DevSys& S, int nb) {
  if (S.dim < (1 << 16) || !S.tred) return 1;
  int p = 1;
  while (p < 128 && (long)nb * p * 2 <= 1024 && (long)S.dim >= (long)4096 * p) p *= 2;
  return p;
}

hipError_t launch_objective(const DevSys& S, const DevTarget& tg, const double* x, int nb, double* out4, hipStream_t st) {
  const int P = objective_blocks(S, nb);
  if (S.lindblad) hipLaunchKernelGGL(k_objective<true>, dim3(nb, P), dim3(256), 0, st, S, tg, x, out4, S.tred);
  else hipLaunchKernelGGL(k_objective<false>, dim3(nb, P), dim3(256), 0, st, S, tg, x, out4, S.tred);
  if (P > 1) hipLaunchKernelGGL(k_objective_sum, dim3(nb), dim3(64), 0, st, S.tred, P, out4);
  return hipGetLastError();
}

hipError_t launch_seed(const DevSys& S, const DevTarget& tg, const double* x, const double* rbar_ibar, int nb, double* xbar,
                       hipStream_t st) {
  const int P = objective_blocks(S, nb);
  if (S.lindblad) hipLaunchKernelGGL(k_seed<true>, dim3(nb, P), dim3(256), 0, st, S, tg, x, rbar_ibar, xbar);
  else hipLaunchKernelGGL(k_seed<false>, dim3(nb, P), dim3(256), 0, st, S, tg, x, rbar_ibar, xbar);
  return hipGetLastError();
}

hipError_t launch_observables(const DevSys& S, const double* traj, int f32, int nb, int nstages, int stride, int nout, int nlev_total,
                              double* expected, double* population, double* expcomp, double* popcomp, hipStream_t st) {
  hipLaunchKernelGGL(k_observables, dim3(nb, nout), dim3(256), 0, st, S, traj, f32, nb, nstages, stride, nlev_total, expected, population,
                     expcomp, popcomp);
  return hipGetLastError();
}

hipError_t launch_partial_sums(const double* res, int nb, const double* w, double inv_ninit, const qd_penalty& pen, const double* etable,
                               int cs, int Q, int nstep, double* sums, hipStream_t st) {
  hipLaunchKernelGGL(k_partial_sums, dim3(1), dim3(256), 0, st, res, nb, w, inv_ninit, pen, etable, cs, Q, nstep, sums);
  return hipGetLastError();
}

hipError_t launch_seed_weights(const double* sums, const double* w, int nb, int objective_type, int lindblad, double* rbib, hipStream_t st) {
  hipLaunchKernelGGL(k_seed_weights, dim3((nb + 127) / 128), dim3(128), 0, st, sums, w, nb, objective_type, lindblad, rbib);
  return hipGetLastError();
}

hipError_t launch_reduce_coeff(const double* coeff, int nb, int ncol, double* sum, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_coeff, dim3((ncol + 255) / 256), dim3(256), 0, st, coeff, nb, ncol, sum, accumulate);
  return hipGetLastError();
}

hipError_t launch_grad(const DevCtlDesc& d, const double* params, const double* table, int cs, int nsub, const double* coeffsum, const double* etable,
                       int nstep, double ebar, double* grad, int ndesign, hipStream_t st) {
  if (ndesign == 0) return hipSuccess;
  const int ee = nsub < 0;  // negative nsub flags the explicit-Euler gradient time (t_stop)
  const int ns = ee ? -nsub : nsub;
  hipLaunchKernelGGL(k_grad, dim3(ndesign), dim3(64), 0, st, d, params, table, cs, ns, ee, coeffsum, etable, nstep, ebar, grad,
                     ndesign);
  return hipGetLastError();
}

}  // namespace qd

// ---------------------------------------------------------------------------------------------
// fp64 vector peak (measurement hook): 8 independent FMA chains per lane, registers only
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fma_peak(double* out, int iters, double a, double b) {
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = (double)(threadIdx.x + i) * 1e-3;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = fma(v[i], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += v[i];
  if (s == 12345.678) out[0] = s;  // never true; keeps the chains alive
}

// the same for fp32 on packed pairs (v_pk_fma_f32: the form the fp32 vector peak is quoted for; profiles/r5_rate_probe.json)
typedef float qd_pk2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_fma_peak_f32(double* out, int iters, float a, float b) {
  qd_pk2 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = qd_pk2{(float)(threadIdx.x + i) * 1e-3f, (float)(threadIdx.x + 8 + i) * 1e-3f};
  const qd_pk2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = __builtin_elementwise_fma(v[i], av, bv);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += v[i].x + v[i].y;
  if (s == 12345.678f) out[0] = s;  // never true; keeps the chains alive
}

static int measure_peak(int device_ordinal, double* tflops, bool f32, const char* who);
extern "C" int qd_measure_fp64_peak(int device_ordinal, double* tflops) { return measure_peak(device_ordinal, tflops, false, "qd_measure_fp64_peak"); }
extern "C" int qd_measure_fp32_peak(int device_ordinal, double* tflops) { return measure_peak(device_ordinal, tflops, true, "qd_measure_fp32_peak"); }
static int measure_peak(int device_ordinal, double* tflops, bool f32, const char* who) {
  if (!tflops) {
    qd::set_error(std::string(who) + ": null output");
    return QD_ERR_INVALID;
  }
  if (hipSetDevice(device_ordinal) != hipSuccess) {
    qd::set_error(std::string(who) + ": no such device");
    return QD_ERR_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return QD_ERR_DEVICE;
  double* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(double)) != hipSuccess) return QD_ERR_NOMEM;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int blocks = prop.multiProcessorCount * 16, threads = 256, iters = 1 << 15;
  auto go = [&] {
    if (f32) hipLaunchKernelGGL(k_fma_peak_f32, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999999f, 1e-9f);
    else hipLaunchKernelGGL(k_fma_peak, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999999, 1e-9);
  };
  go();  // warm-up (clock ramp)
  hipError_t err = hipSuccess;
  float ms = 0.f;
  for (int rep = 0; rep < 3; rep++) {  // best of three
    (void)hipEventRecord(e0, 0);
    go();
    (void)hipEventRecord(e1, 0);
    err = hipEventSynchronize(e1);
    float m = 0.f;
    (void)hipEventElapsedTime(&m, e0, e1);
    if (err != hipSuccess) break;
    if (rep == 0 || m < ms) ms = m;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  if (err != hipSuccess || ms <= 0.f) {
    qd::set_error(std::string(who) + ": kernel failed");
    return QD_ERR_DEVICE;
  }
  *tflops = (f32 ? 2.0 : 1.0) * 2.0 * 8.0 * (double)iters * (double)blocks * (double)threads / (ms * 1e-3) / 1e12;
  return QD_OK;
}
