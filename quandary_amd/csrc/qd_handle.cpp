// Operator / stepper level of the C ABI (include/quandary_amd.h): qd_create ... qd_adjoint.
// Host code only orchestrates: every state-sized operation runs in the HIP kernels of
// qd_kernels.hip.  There is no CPU fallback: without a HIP device qd_create fails.
#include <algorithm>

#include "qd_handle.h"

#include <cmath>
#include <cstdio>

namespace qd {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace qd

using namespace qd;

static int fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}

extern "C" const char* qd_last_error(void) { return qd::g_err.c_str(); }
extern "C" const char* qd_version(void) { return "quandary_amd 0.1.0 (gfx950)"; }

extern "C" int qd_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(QD_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

// CompositionalImplMidpoint coefficients (src/timestepper.cpp:735-757)
static int stage_gammas(int stepper, double* gam) {
  if (stepper == QD_STEPPER_IMR8) {
    static const double g8[15] = {0.74167036435061295344822780,  -0.40910082580003159399730010, 0.19075471029623837995387626,
                                  -0.57386247111608226665638773, 0.29906418130365592384446354,  0.33462491824529818378495798,
                                  0.31529309239676659663205666,  -0.79688793935291635401978884, 0.31529309239676659663205666,
                                  0.33462491824529818378495798,  0.29906418130365592384446354,  -0.57386247111608226665638773,
                                  0.19075471029623837995387626,  -0.40910082580003159399730010, 0.74167036435061295344822780};
    for (int i = 0; i < 15; i++) gam[i] = g8[i];
    return 15;
  }
  if (stepper == QD_STEPPER_IMR4) {
    gam[0] = 1. / (2. - pow(2., 1. / 3.));
    gam[1] = -pow(2., 1. / 3.) * gam[0];
    gam[2] = 1. / (2. - pow(2., 1. / 3.));
    return 3;
  }
  gam[0] = 1.0;
  return 1;
}

template <typename T>
static int upload(T** dst, const std::vector<T>& v) {
  *dst = nullptr;
  size_t n = v.size() > 0 ? v.size() : 1;
  QD_HIP(hipMalloc(reinterpret_cast<void**>(dst), sizeof(T) * n));
  if (!v.empty()) QD_HIP(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
  return QD_OK;
}

extern "C" void qd_destroy(qd_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  struct Quiet { ~Quiet() { (void)hipGetLastError(); } } quiet;  // teardown never leaves a sticky error behind
  for (DBuf* b : {&h->d_params, &h->d_tbar, &h->d_tred, &h->d_sched_t, &h->d_sched_h, &h->d_etimes, &h->d_ezero, &h->d_table, &h->d_etable, &h->d_onerow,
                  &h->d_onetime, &h->d_tstates, &h->d_purity, &h->d_x0, &h->d_xT, &h->d_traj, &h->d_ztraj, &h->d_res,
                  &h->d_xbar, &h->d_jbar, &h->d_coeff, &h->d_coeffsum, &h->d_grad, &h->d_y, &h->d_sched, &h->d_stash, &h->d_kry, &h->d_ecoef, &h->d_edig, &h->d_work, &h->d_g0, &h->d_hcr, &h->d_hci, &h->d_gtab, &h->d_gone})
    b->release();
  if (h->d_segs) (void)hipFree(h->d_segs);
  if (h->d_oscs) (void)hipFree(h->d_oscs);
  if (h->d_carriers) (void)hipFree(h->d_carriers);
  if (h->d_pulses) (void)hipFree(h->d_pulses);
  h->h_etable.release();
  h->h_sched.release();
  h->h_res.release();
  h->h_params.release();
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->ev3) (void)hipEventDestroy(h->ev3);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int qd_create(const qd_system* sys, const qd_controls* ctl, const qd_time* tg, const qd_solver* sol, int device_ordinal,
                         qd_handle** out) {
  if (!sys || !ctl || !tg || !sol || !out) return fail(QD_ERR_INVALID, "qd_create: null argument");
  *out = nullptr;
  if (sys->nosc < 1 || sys->nosc > QD_MAX_OSC) return fail(QD_ERR_INVALID, "qd_create: nosc out of range");
  if (tg->ntime < 1 || !(tg->dt > 0.0)) return fail(QD_ERR_INVALID, "qd_create: ntime and dt must be positive");
  if (sol->stepper < QD_STEPPER_IMR || sol->stepper > QD_STEPPER_EE) return fail(QD_ERR_INVALID, "qd_create: unknown timestepper");
  if (sol->linsolve != QD_LINSOLVE_GMRES && sol->linsolve != QD_LINSOLVE_NEUMANN)
    return fail(QD_ERR_INVALID, "qd_create: unknown linear solver");
  int ndev = qd_device_count();
  if (ndev <= 0) return fail(QD_ERR_DEVICE, "qd_create: no HIP device visible (this library has no CPU path)");
  if (device_ordinal < 0 || device_ordinal >= ndev) return fail(QD_ERR_INVALID, "qd_create: device ordinal out of range");

  qd_handle* h = new qd_handle();
  h->device = device_ordinal;
  h->opts.load_env();  // QD_<KEY>: overrides for tests and measurements, read once
  h->tg = *tg;
  h->sol = *sol;
  // ---- system constants: src/mastereq.cpp:14-60, src/oscillator.cpp:15-23, src/main.cpp:299-307
  DevSys& S = h->S;
  std::memset(&S, 0, sizeof S);
  S.Q = sys->nosc;
  S.lindblad = sys->lindblad_type != QD_LINDBLAD_NONE;
  const bool addT1 = sys->lindblad_type == QD_LINDBLAD_DECAY || sys->lindblad_type == QD_LINDBLAD_BOTH;
  const bool addT2 = sys->lindblad_type == QD_LINDBLAD_DEPHASE || sys->lindblad_type == QD_LINDBLAD_BOTH;
  long long N = 1;
  h->dim_ess = 1;
  for (int k = 0; k < S.Q; k++) {
    S.n[k] = sys->nlevels[k];
    S.ness[k] = sys->nessential[k];
    if (S.n[k] < 2 || S.n[k] > 255) {  // (the kernel variants are instantiated for dim >= 2^Q / 4^Q)
      delete h;
      return fail(QD_ERR_INVALID, "qd_create: nlevels must be in 2..255");
    }
    if (S.ness[k] < 1 || S.ness[k] > S.n[k]) S.ness[k] = S.n[k];
    N *= S.n[k];
    h->dim_ess *= S.ness[k];
    if (S.n[k] > S.maxn) S.maxn = S.n[k];
  }
  long long dim = S.lindblad ? N * N : N;
  if (dim > QD_MAX_DIM) {
    delete h;
    return fail(QD_ERR_UNSUPPORTED, "qd_create: state dimension above QD_MAX_DIM (2^22)");
  }
  {  // packed digits (qd_device.h: packed_digit_bits)
    const int maxlev = S.Q <= 4 ? 256 : S.Q == 5 ? 64 : S.Q == 6 ? 32 : 16;
    if (S.maxn > maxlev) {
      delete h;
      return fail(QD_ERR_UNSUPPORTED, "qd_create: too many levels per oscillator for this number of oscillators (packed digits: 256/64/32/16 levels for <=4/5/6/7-8 oscillators)");
    }
  }
  S.N = (int)N;
  S.dim = (int)dim;
  h->poly_cur = h->poly_start();
  for (int k = 0; k < S.Q; k++) {
    S.post[k] = 1;
    for (int j = k + 1; j < S.Q; j++) S.post[k] *= S.n[j];
    S.detune[k] = 2.0 * M_PI * (sys->transfreq[k] - sys->rotfreq[k]);
    S.xi[k] = 2.0 * M_PI * sys->selfkerr[k];
    S.g1[k] = (sys->decay_time[k] > 1e-14 && addT1) ? 1.0 / sys->decay_time[k] : 0.0;
    S.g1off[k] = fabs(S.g1[k]) > 1e-12 ? S.g1[k] : 0.0;  // threshold of L1decay (mastereq.hpp:759)
    S.g2[k] = (sys->dephase_time[k] > 1e-14 && addT2) ? 1.0 / sys->dephase_time[k] : 0.0;
  }
  S.npairs = S.Q * (S.Q - 1) / 2;
  DevCtlDesc& D = h->dctl;
  std::memset(&D, 0, sizeof D);
  int idx = 0;
  for (int k = 0; k < S.Q; k++)
    for (int l = k + 1; l < S.Q; l++) {
      S.xikl[idx] = 2.0 * M_PI * sys->crosskerr[idx];
      S.J[idx] = 2.0 * M_PI * sys->Jkl[idx];
      if (!(fabs(S.J[idx]) > 1e-10)) S.J[idx] = 0.0;  // threshold of Jkl_coupling (mastereq.hpp:633)
      else S.hasJ = 1;
      D.eta[idx] = 2.0 * M_PI * (sys->rotfreq[k] - sys->rotfreq[l]);
      idx++;
    }
  // ---- controls: src/oscillator.cpp:45-132, src/controlbasis.cpp:20-32,219-225
  int carpos = 0, off = 0;
  for (int k = 0; k < S.Q; k++) {
    DevOsc o{};
    o.seg_begin = (int)h->segs.size();
    o.car_begin = (int)h->carriers.size();
    o.ncar = ctl->ncarrier ? ctl->ncarrier[k] : 0;
    for (int f = 0; f < o.ncar; f++) h->carriers.push_back(2.0 * M_PI * ctl->carrier_freq[carpos + f]);
    carpos += o.ncar;
    int skip = 0;
    for (int g = 0; g < ctl->nseg_total; g++) {
      if (ctl->seg_osc[g] != k) continue;
      DevSeg sg{};
      sg.type = ctl->seg_type[g];
      sg.nsplines = ctl->seg_nsplines[g];
      sg.tstart = ctl->seg_tstart[g];
      sg.tstop = ctl->seg_tstop[g];
      sg.skip = skip;
      if (sg.type == QD_CTRL_BSPLINE) {
        if (sg.nsplines < 3) { delete h; return fail(QD_ERR_INVALID, "qd_create: spline segment needs >= 3 splines"); }
        sg.dtknot = (sg.tstop - sg.tstart) / (double)(sg.nsplines - 2);
        sg.width = 3.0 * sg.dtknot;
      } else if (sg.type == QD_CTRL_BSPLINE0) {
        if (sg.nsplines < 2) { delete h; return fail(QD_ERR_INVALID, "qd_create: spline0 segment needs >= 2 splines"); }
        sg.dtknot = (sg.tstop - sg.tstart) / (sg.nsplines - 1.0);
        sg.width = sg.dtknot;
      } else if (sg.type == QD_CTRL_STEP) {  // Step ctor, controlbasis.cpp:186-191: one parameter, index skip + 2*carrier (:197)
        if (!ctl->seg_param) { delete h; return fail(QD_ERR_INVALID, "qd_create: step segment without seg_param"); }
        if (o.ncar != 1) {
          delete h;
          return fail(QD_ERR_UNSUPPORTED, "qd_create: step segment with more than one carrier wave (the reference indexes past the segment's parameter)");
        }
        sg.nsplines = 1;
        sg.npc = 1;
        sg.a1 = ctl->seg_param[3 * g];
        sg.a2 = ctl->seg_param[3 * g + 1];
        sg.a3 = ctl->seg_param[3 * g + 2];
      } else if (sg.type == QD_CTRL_BSPLINEAMP) {  // BSpline2ndAmplitude ctor, controlbasis.cpp:99-112
        if (!ctl->seg_param) { delete h; return fail(QD_ERR_INVALID, "qd_create: spline_amplitude segment without seg_param"); }
        if (sg.nsplines < 3) { delete h; return fail(QD_ERR_INVALID, "qd_create: spline_amplitude segment needs >= 3 splines"); }
        sg.dtknot = (sg.tstop - sg.tstart) / (double)(sg.nsplines - 2);
        sg.width = 3.0 * sg.dtknot;
        sg.npc = sg.nsplines + 1;
        sg.a1 = ctl->seg_param[3 * g];
        h->has_ampbasis = true;
      } else {
        delete h;
        return fail(QD_ERR_INVALID, "qd_create: unknown control segment type");
      }
      if (sg.type == QD_CTRL_BSPLINE || sg.type == QD_CTRL_BSPLINE0) sg.npc = 2 * sg.nsplines;
      skip += sg.npc * o.ncar;
      h->segs.push_back(sg);
      o.nseg++;
    }
    o.nparams = skip;
    o.offset = off;
    off += skip;
    o.pulse_begin = (int)h->pulses.size() / 3;
    for (int i = 0; i < ctl->npipulse; i++)
      if (ctl->pipulse_osc[i] == k) {
        h->pulses.push_back(ctl->pipulse_tstart[i]);
        h->pulses.push_back(ctl->pipulse_tstop[i]);
        h->pulses.push_back(ctl->pipulse_amp[i]);
        o.npulse++;
        h->has_pipulse = true;
      }
    h->oscs.push_back(o);
  }
  h->ndesign = off;
  h->params.assign(off, 0.0);

  // ---- device resources
  int rc = QD_OK;
  auto dev_setup = [&]() -> int {
    QD_HIP(qd::use_device(h->device));
    QD_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    QD_HIP(hipEventCreate(&h->ev0));
    QD_HIP(hipEventCreate(&h->ev1));
    QD_HIP(hipEventCreate(&h->ev2));
    QD_HIP(hipEventCreate(&h->ev3));
    int r;
    if ((r = upload(&h->d_segs, h->segs))) return r;
    if ((r = upload(&h->d_oscs, h->oscs))) return r;
    if ((r = upload(&h->d_carriers, h->carriers))) return r;
    if ((r = upload(&h->d_pulses, h->pulses))) return r;
    return QD_OK;
  };
  rc = dev_setup();
  if (rc) {
    qd_destroy(h);
    return rc;
  }
  D.Q = S.Q;
  D.enforce_bc = ctl->enforce_bc;
  D.npairs = S.npairs;
  D.segs = h->d_segs;
  D.oscs = h->d_oscs;
  D.carriers = h->d_carriers;
  D.pulses = h->d_pulses;
  D.Tfinal = tg->ntime * tg->dt;

  // ---- step schedule (times exactly as the reference forms them: timestepper.cpp:128-129, :587-590, :786-798)
  double gam[15];
  h->nstages = (sol->stepper == QD_STEPPER_EE) ? 1 : stage_gammas(sol->stepper, gam);
  h->nsub = tg->ntime * h->nstages;
  h->cs = ctl_stride(S.Q, S.npairs);
  for (int n = 0; n < tg->ntime; n++) {
    const double tstart = n * tg->dt, tstop = (n + 1) * tg->dt;
    if (sol->stepper == QD_STEPPER_EE) {
      h->sched_t.push_back(tstart);
      h->sched_h.push_back(tstop - tstart);
    } else if (sol->stepper == QD_STEPPER_IMR) {
      h->sched_t.push_back((tstart + tstop) / 2.0);
      h->sched_h.push_back(tstop - tstart);
    } else {
      const double dt = tstop - tstart;
      double tcurr = tstart;
      for (int s = 0; s < h->nstages; s++) {
        const double dts = gam[s] * dt;
        const double t1 = tcurr + dts;
        h->sched_t.push_back((tcurr + t1) / 2.0);
        h->sched_h.push_back(t1 - tcurr);
        tcurr = tcurr + dts;
      }
    }
    h->etimes.push_back(tstop);
  }
  if (sol->stepper == QD_STEPPER_EE) {  // extra row M(T) for the adjoint of the last step
    h->sched_t.push_back(tg->ntime * tg->dt);
    h->sched_h.push_back(0.0);
  }
  auto sched_setup = [&]() -> int {
    int r;
    const size_t nrows = h->sched_t.size();
    if ((r = h->d_sched_t.ensure(nrows))) return r;
    if ((r = h->d_sched_h.ensure(nrows))) return r;
    if ((r = h->d_etimes.ensure(h->etimes.size()))) return r;
    if ((r = h->d_ezero.ensure(h->etimes.size()))) return r;
    if ((r = h->d_table.ensure(nrows * h->cs))) return r;
    if ((r = h->d_etable.ensure(h->etimes.size() * h->cs))) return r;
    if ((r = h->d_onerow.ensure(h->cs))) return r;
    if ((r = h->d_onetime.ensure(2))) return r;
    if ((r = h->d_params.ensure(h->ndesign))) return r;
    QD_HIP(hipMemcpy(h->d_sched_t.p, h->sched_t.data(), sizeof(double) * nrows, hipMemcpyHostToDevice));
    QD_HIP(hipMemcpy(h->d_sched_h.p, h->sched_h.data(), sizeof(double) * nrows, hipMemcpyHostToDevice));
    QD_HIP(hipMemcpy(h->d_etimes.p, h->etimes.data(), sizeof(double) * h->etimes.size(), hipMemcpyHostToDevice));
    QD_HIP(hipMemset(h->d_ezero.p, 0, sizeof(double) * h->etimes.size()));
    QD_HIP(hipMemset(h->d_params.p, 0, sizeof(double) * (h->ndesign > 0 ? h->ndesign : 1)));
    return QD_OK;
  };
  rc = sched_setup();
  if (rc) {
    qd_destroy(h);
    return rc;
  }
  if (h->h_etable.ensure(h->etimes.size() * h->cs) != QD_OK || h->h_params.ensure((size_t)h->ndesign) != QD_OK) {
    qd_destroy(h);
    return QD_ERR_NOMEM;
  }
  std::memset(h->h_etable.p, 0, sizeof(double) * h->etimes.size() * h->cs);
  *out = h;
  return QD_OK;
}

extern "C" int qd_dim(const qd_handle* h) { return h ? h->S.dim : QD_ERR_INVALID; }
extern "C" int qd_dim_rho(const qd_handle* h) { return h ? h->S.N : QD_ERR_INVALID; }
extern "C" int qd_dim_ess(const qd_handle* h) { return h ? h->dim_ess : QD_ERR_INVALID; }
extern "C" int qd_ndesign(const qd_handle* h) { return h ? h->ndesign : QD_ERR_INVALID; }

extern "C" int qd_set_hamiltonian(qd_handle* h, const double* hsys_re, const double* hsys_im, const double* hc_re, const double* hc_im) {
  if (!h || !hsys_re || !hsys_im) return fail(QD_ERR_INVALID, "qd_set_hamiltonian: null system Hamiltonian");
  if ((hc_re == nullptr) != (hc_im == nullptr)) return fail(QD_ERR_INVALID, "qd_set_hamiltonian: give both parts of the control Hamiltonians or neither");
  // (dim <= 1024: the LDS kernels V11-V13 / V15; beyond: the global-memory sweeps of qd_big.h with the dense operator)
  QD_HIP(qd::use_device(h->device));
  const size_t nn = (size_t)h->S.N * h->S.N;
  if ((double)h->sched_t.size() * (double)nn * 16.0 > 16e9)
    return fail(QD_ERR_UNSUPPORTED, "qd_set_hamiltonian: the table of G(t) would exceed 16 GB (time steps x N^2)");
  // G0 = -i Hsys = Im(Hsys) - i Re(Hsys): Ad = Im(Hsys), Bd = -Re(Hsys) (src/hamiltonianfilereader.cpp:77-84)
  std::vector<double> g0(2 * nn), cr((size_t)h->S.Q * nn, 0.0), ci((size_t)h->S.Q * nn, 0.0);
  for (size_t e = 0; e < nn; e++) {
    g0[2 * e] = hsys_im[e];
    g0[2 * e + 1] = -hsys_re[e];
  }
  if (hc_re) {
    std::copy(hc_re, hc_re + cr.size(), cr.begin());
    std::copy(hc_im, hc_im + ci.size(), ci.begin());
  }
  // row sums for the Gershgorin bounds of the solver gates (row_bounds): H is Hermitian, so the row sums bound the column sums too
  {
    const int N = h->S.N;
    auto rowsum_max = [&](const double* a, const double* b) {
      double m = 0.0;
      for (int i = 0; i < N; i++) {
        double s = 0.0;
        for (int j = 0; j < N; j++) s += hypot(a[(size_t)i * N + j], b[(size_t)i * N + j]);
        m = std::max(m, s);
      }
      for (int j = 0; j < N; j++) {  // (and the column sums, should a caller hand over a non-Hermitian matrix)
        double s = 0.0;
        for (int i = 0; i < N; i++) s += hypot(a[(size_t)i * N + j], b[(size_t)i * N + j]);
        m = std::max(m, s);
      }
      return m;
    };
    std::vector<double> zero(nn, 0.0);
    h->dense_hsys_norm = rowsum_max(hsys_re, hsys_im);
    h->dense_hc_norm.assign(h->S.Q, 0.0);
    if (hc_re)
      for (int k = 0; k < h->S.Q; k++)
        h->dense_hc_norm[k] = rowsum_max(hc_re + (size_t)k * nn, zero.data()) + rowsum_max(hc_im + (size_t)k * nn, zero.data());
  }
  int r;
  if ((r = h->d_g0.ensure(g0.size())) || (r = h->d_hcr.ensure(cr.size())) || (r = h->d_hci.ensure(ci.size()))) return r;
  QD_HIP(hipMemcpy(h->d_g0.p, g0.data(), sizeof(double) * g0.size(), hipMemcpyHostToDevice));
  QD_HIP(hipMemcpy(h->d_hcr.p, cr.data(), sizeof(double) * cr.size(), hipMemcpyHostToDevice));
  QD_HIP(hipMemcpy(h->d_hci.p, ci.data(), sizeof(double) * ci.size(), hipMemcpyHostToDevice));
  h->S.dense = nn * 16 <= 64 * 1024 ? 2 : 1;  // 2: G(t) of the current sub-step is staged in LDS (N <= 64)
  h->S.hcr = h->d_hcr.p;
  h->S.hci = h->d_hci.p;
  h->S.gtab = nullptr;
  h->sub_latch = -1;
  h->S.hasJ = 0;  // the file model replaces the standard one including the dipole-dipole terms (src/mastereq.cpp:273-284)
  h->params_dirty = true;
  h->traj_valid = false;
  return QD_OK;
}

extern "C" int qd_set_option(qd_handle* h, const char* key, const char* value) {
  if (!h || !key || !value) return fail(QD_ERR_INVALID, "qd_set_option: null argument");
  if (h->opts.set(key, value) != 0) return fail(QD_ERR_INVALID, std::string("qd_set_option: unknown key or bad value: ") + key + " = " + value);
  h->sub_latch = -1;
  if (std::string(key) == "gmres_poly") {  // (any setting of the degree starts the tuner over: `auto` re-tunes at the current parameters)
    h->poly_frozen = false;
    h->poly_cur = h->poly_start();
    h->poly_lo = 1;
    h->poly_hi = 0;
    h->poly_steps = 0;
    h->poly_slow = 0;
  }
  h->traj_valid = false;  // (a stored trajectory may have another layout under the new options)
  return QD_OK;
}

extern "C" int qd_get_precision(const qd_handle* h) { return h ? h->precision : QD_ERR_INVALID; }

extern "C" int qd_set_precision(qd_handle* h, int precision) {
  if (!h) return fail(QD_ERR_INVALID, "qd_set_precision: null handle");
  if (precision != QD_PRECISION_F64 && precision != QD_PRECISION_F32MIXED) return fail(QD_ERR_INVALID, "qd_set_precision: unknown precision");
  if (precision == QD_PRECISION_F32MIXED) {
    const DevSys& S = h->S;
    bool qubits = S.lindblad && !S.dense && (S.Q == 3 || S.Q == 4 || S.Q == 5);
    for (int k = 0; k < S.Q; k++) qubits = qubits && S.n[k] == 2 && S.ness[k] == 2;
    if (!qubits || (S.hasJ && S.Q == 3))
      return fail(QD_ERR_UNSUPPORTED, "qd_set_precision: the fp32-mixed sweeps are built for all-qubit Lindblad systems with 3, 4 or 5 oscillators (dipole-dipole coupling: 4 or 5)");
    if (h->sol.stepper == QD_STEPPER_EE)
      return fail(QD_ERR_UNSUPPORTED, "qd_set_precision: the fp32-mixed sweeps need a stepper of the IMR family");
  }
  h->precision = precision;
  h->sub_latch = -1;
  h->traj_valid = false;
  return QD_OK;
}

extern "C" int qd_set_params(qd_handle* h, const double* alpha, int ndesign) {
  if (!h || (!alpha && ndesign > 0)) return fail(QD_ERR_INVALID, "qd_set_params: null argument");
  if (ndesign != h->ndesign) return fail(QD_ERR_INVALID, "qd_set_params: ndesign mismatch");
  h->params_set = true;
  QD_HIP(qd::use_device(h->device));
  if (ndesign > 0) {
    std::memcpy(h->params.data(), alpha, sizeof(double) * ndesign);
    std::memcpy(h->h_params.p, alpha, sizeof(double) * ndesign);
    QD_HIP(hipMemcpyAsync(h->d_params.p, h->h_params.p, sizeof(double) * ndesign, hipMemcpyHostToDevice, h->stream));
  }
  h->params_dirty = true;
  h->traj_valid = false;
  return QD_OK;
}

// Evaluate the control tables for the current parameters (one tiny kernel per parameter update
// instead of Q*ncarrier*nsplines basis evaluations per step and initial condition).
int qd_handle::refresh_tables() {
  if (!params_dirty) return QD_OK;
  // step table + energy-penalty table in one launch; it also resets the RHS-application counter of the sweep
  QD_HIP(launch_controls2(dctl, d_params.p, d_sched_t.p, d_sched_h.p, (int)sched_t.size(), d_table.p, d_etimes.p, d_ezero.p,
                          (int)etimes.size(), d_etable.p, cs, d_napply, stream));
  napply_zeroed = d_napply != nullptr;
  if (S.dense) {  // G(t) = -i H(t) for every table row, shared by all initial conditions
    const size_t nn = (size_t)S.N * S.N;
    int r;
    if ((r = d_gtab.ensure(sched_t.size() * nn * 2))) return r;
    S.gtab = d_gtab.p;
    QD_HIP(launch_gmat(S, d_g0.p, d_table.p, cs, (int)sched_t.size(), d_gtab.p, stream));
  }
  // asynchronous: complete at the stream synchronisation that ends the sweep (forward_dev)
  QD_HIP(hipMemcpyAsync(h_etable.p, d_etable.p, sizeof(double) * etimes.size() * cs, hipMemcpyDeviceToHost, stream));
  params_dirty = false;
  return QD_OK;
}

// energyPenaltyIntegral summed over the time loop (src/timestepper.cpp:154, :444-455)
double qd_handle::energy_penalty_host() const {
  double e = 0.0;
  for (size_t n = 0; n < etimes.size(); n++) {
    double pen = 0.0;
    const double* row = h_etable.p + n * cs;
    for (int k = 0; k < S.Q; k++) pen += (row[2 + k] * row[2 + k] + row[2 + S.Q + k] * row[2 + S.Q + k]) / tg.ntime;
    e += pen;
  }
  return e;
}

extern "C" int qd_eval_controls(qd_handle* h, const double* times, int nt, double* pq) {
  if (!h || !times || !pq || nt < 0) return fail(QD_ERR_INVALID, "qd_eval_controls: bad argument");
  QD_HIP(qd::use_device(h->device));
  for (int i = 0; i < nt; i++)
    if (times[i] > h->dctl.Tfinal) return fail(QD_ERR_INVALID, "qd_eval_controls: t > Tfinal (src/oscillator.cpp:284-287)");
  DBuf dt, dz, dtab;
  int r;
  if ((r = dt.ensure(nt)) || (r = dz.ensure(nt)) || (r = dtab.ensure((size_t)nt * h->cs))) return r;
  QD_HIP(hipMemcpyAsync(dt.p, times, sizeof(double) * nt, hipMemcpyHostToDevice, h->stream));
  QD_HIP(hipMemsetAsync(dz.p, 0, sizeof(double) * nt, h->stream));
  QD_HIP(launch_controls(h->dctl, h->d_params.p, dt.p, dz.p, nt, dtab.p, h->cs, h->stream));
  std::vector<double> tab((size_t)nt * h->cs);
  QD_HIP(hipMemcpyAsync(tab.data(), dtab.p, sizeof(double) * tab.size(), hipMemcpyDeviceToHost, h->stream));
  QD_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < nt; i++)
    for (int k = 0; k < h->S.Q; k++) {
      pq[(i * h->S.Q + k) * 2] = tab[(size_t)i * h->cs + 2 + k];
      pq[(i * h->S.Q + k) * 2 + 1] = tab[(size_t)i * h->cs + 2 + h->S.Q + k];
    }
  dt.release();
  dz.release();
  dtab.release();
  return QD_OK;
}

// element table (once per handle) and work vectors (per batch size) of the large-state variant
int qd_handle::ensure_big(int nb) {
  int r;
  if (!d_ecoef.p) {
    if ((r = d_ecoef.ensure((size_t)2 * S.dim)) || (r = d_edig.ensure((size_t)S.dim))) return r;
    QD_HIP(launch_big_table(S, d_ecoef.p, reinterpret_cast<unsigned*>(d_edig.p), stream));
    S.ecoef = d_ecoef.p;
    S.edig = reinterpret_cast<const unsigned*>(d_edig.p);
  }
  if ((r = d_work.ensure(big_work_doubles(S, nb)))) return r;
  S.work = d_work.p;
  // teams of workgroups: barrier counters and partial-sum buffers (sized for the largest team; zeroed at every launch)
  if ((r = d_tbar.ensure((size_t)nb * BIG_BAR_STRIDE)) || (r = d_tred.ensure((size_t)nb * 2 * BIG_TEAM_MAX * BIG_RED_NV))) return r;
  S.tbar = reinterpret_cast<unsigned long long*>(d_tbar.p);
  S.tred = d_tred.p;
  return QD_OK;
}

static int check_cfg(const LaunchCfg& cfg) {
  if (variant_max_block(cfg.var) <= 0 || cfg.block > variant_max_block(cfg.var))
    return fail(QD_ERR_UNSUPPORTED, "state dimension too large for the single-workgroup kernels");
  if (cfg.lds > 160 * 1024) return fail(QD_ERR_UNSUPPORTED, "state does not fit the 160 KiB LDS of one CU");
  return QD_OK;
}

extern "C" int qd_apply_rhs(qd_handle* h, double t, int transpose, const double* x, double* y, int nb) {
  if (!h || !x || !y || nb < 1) return fail(QD_ERR_INVALID, "qd_apply_rhs: bad argument");
  if (t > h->dctl.Tfinal) return fail(QD_ERR_INVALID, "qd_apply_rhs: t > Tfinal (src/oscillator.cpp:284-287)");
  QD_HIP(qd::use_device(h->device));
  const size_t n = (size_t)nb * 2 * h->S.dim;
  int r;
  if ((r = h->d_x0.ensure(n)) || (r = h->d_y.ensure(n))) return r;
  const double tt[2] = {t, 0.0};
  QD_HIP(hipMemcpyAsync(h->d_onetime.p, tt, sizeof tt, hipMemcpyHostToDevice, h->stream));
  QD_HIP(launch_controls(h->dctl, h->d_params.p, h->d_onetime.p, h->d_onetime.p + 1, 1, h->d_onerow.p, h->cs, h->stream));
  QD_HIP(hipMemcpyAsync(h->d_x0.p, x, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  LaunchCfg cfg = pick_config(h->S, nb, h->opts);
  if ((r = check_cfg(cfg))) return r;
  if (cfg.var == 16 && (r = h->ensure_big(nb))) return r;
  qd::DevSys Sone = h->S;
  if (h->S.dense) {
    if ((r = h->d_gone.ensure((size_t)2 * h->S.N * h->S.N))) return r;
    QD_HIP(launch_gmat(h->S, h->d_g0.p, h->d_onerow.p, h->cs, 1, h->d_gone.p, h->stream));
    Sone.gtab = h->d_gone.p;
  }
  if (h->precision == QD_PRECISION_F32MIXED) QD_HIP(launch_apply_f32(Sone, h->d_onerow.p, transpose, h->d_x0.p, h->d_y.p, nb, 1, 0, h->opts, h->stream));
  else if (cfg.var != 16 && lean64_available(h->S, h->opts)) QD_HIP(launch_apply_lean64(Sone, h->d_onerow.p, transpose, h->d_x0.p, h->d_y.p, nb, h->stream));
  else if (cfg.var == 9 && collean_available(h->S, h->opts)) QD_HIP(launch_apply_col(Sone, h->d_onerow.p, transpose, h->d_x0.p, h->d_y.p, nb, h->opts, h->stream));
  else QD_HIP(launch_apply(Sone, h->d_onerow.p, transpose, h->d_x0.p, h->d_y.p, nb, cfg, h->stream));
  QD_HIP(hipMemcpyAsync(y, h->d_y.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  QD_HIP(hipStreamSynchronize(h->stream));
  return QD_OK;
}

extern "C" int qd_set_target(qd_handle* h, const qd_target* tgt, int nb) {
  if (!h || !tgt || nb < 1) return fail(QD_ERR_INVALID, "qd_set_target: bad argument");
  if (tgt->target_type != QD_TARGET_PURE && !tgt->target_states) return fail(QD_ERR_INVALID, "qd_set_target: target states required");
  if (tgt->objective_type == QD_OBJ_JMEASURE && tgt->target_type != QD_TARGET_PURE)
    return fail(QD_ERR_INVALID, "qd_set_target: Jmeasure needs a pure target (src/optimtarget.cpp:758-761)");
  QD_HIP(qd::use_device(h->device));
  int r;
  h->dtg.target_type = tgt->target_type;
  h->dtg.objective_type = tgt->objective_type;
  h->dtg.purestate_id = tgt->purestate_id;
  h->dtg.idm = h->S.lindblad ? tgt->purestate_id * (h->S.N + 1) : tgt->purestate_id;
  h->dtg.tstates = nullptr;
  if (tgt->target_type != QD_TARGET_PURE) {
    const size_t n = (size_t)nb * 2 * h->S.dim;
    if ((r = h->d_tstates.ensure(n))) return r;
    QD_HIP(hipMemcpy(h->d_tstates.p, tgt->target_states, sizeof(double) * n, hipMemcpyHostToDevice));
    h->dtg.tstates = h->d_tstates.p;
  }
  std::vector<double> pur(nb, 1.0);
  if (tgt->purity)
    for (int i = 0; i < nb; i++) pur[i] = tgt->purity[i];
  if ((r = h->d_purity.ensure(nb))) return r;
  QD_HIP(hipMemcpy(h->d_purity.p, pur.data(), sizeof(double) * nb, hipMemcpyHostToDevice));
  h->dtg.purity = h->d_purity.p;
  h->target_set = true;
  h->target_nb = nb;
  return QD_OK;
}

extern "C" int qd_set_penalty(qd_handle* h, const qd_penalty* pen) {
  if (!h || !pen) return fail(QD_ERR_INVALID, "qd_set_penalty: null argument");
  h->pen = *pen;
  return QD_OK;
}

int qd_handle::traj_doubles(int nb, size_t* n) const {
  *n = (size_t)(nsub + 1) * (size_t)nb * 2 * (size_t)S.dim;
  if (precision == QD_PRECISION_F32MIXED) *n /= 2;  // float2 per element
  return QD_OK;
}

size_t qd_handle::ztraj_doubles(int nb) const {
  if (sol.stepper == QD_STEPPER_EE) return 0;
  size_t n = (size_t)nsub * (size_t)nb * 2 * (size_t)S.dim;
  if (precision == QD_PRECISION_F32MIXED) n /= 2;
  return n;
}

// max over time of |p_k(t)|, |q_k(t)| from the CURRENT control parameters: sum over carriers of max |alpha^1| + max |alpha^2| (the
// quadratic B-splines are a partition of unity, src/controlbasis.cpp:81-96); pi-pulses by their amplitude
double qd_handle::control_amplitude_bound(int k) const {
  double amp = 0.0;
  const DevOsc& o = oscs[k];
  for (int b = 0; b < o.nseg; b++) {
    const DevSeg& g = segs[o.seg_begin + b];
    double a = 0.0;
    for (int f = 0; f < o.ncar; f++) {
      double m1 = 0.0, m2 = 0.0;
      if (g.type == QD_CTRL_STEP) {
        m1 = fabs(g.a1);
        m2 = fabs(g.a2);
      } else if (g.type == QD_CTRL_BSPLINEAMP) {
        for (int l = 0; l < g.nsplines; l++) m1 = std::max(m1, fabs(params[o.offset + g.skip + f * g.npc + l]));
        m2 = m1;
      } else {
        for (int l = 0; l < g.nsplines; l++) {
          m1 = std::max(m1, fabs(params[o.offset + g.skip + f * 2 * g.nsplines + l]));
          m2 = std::max(m2, fabs(params[o.offset + g.skip + f * 2 * g.nsplines + g.nsplines + l]));
        }
      }
      a += m1 + m2;
    }
    amp = std::max(amp, a);
  }
  for (int i = 0; i < o.npulse; i++) amp = std::max(amp, fabs(pulses[(size_t)(o.pulse_begin + i) * 3 + 2]));
  return amp;
}

// Gershgorin bounds of one row of M(t) over all sub-steps, from the system constants and the CURRENT control parameters:
// diag = |Delta| + |d| (level energies, diagonal decay), off = the T1 off-diagonal entry, the control ladder entries (|p_k(t)|,
// |q_k(t)| <= sum over carriers of max |alpha^1| + max |alpha^2|: the quadratic B-splines are a partition of unity,
// src/controlbasis.cpp:81-96; pi-pulses by their amplitude) and the dipole-dipole couplings.
void qd_handle::row_bounds(double* diag, double* off) const {
  if (S.dense) {
    // user Hamiltonians (qd_set_hamiltonian): the standard-model constants do not describe the operator.  M = -i(I x H - H^T x I) +
    // the standard decay terms, H(t) = Hsys + sum_k p_k Re-part + q_k Im-part of Hc_k (src/mastereq.cpp:743-830): a row of M is
    // bounded by (2 x) the row sums of the uploaded matrices.
    const double two = S.lindblad ? 2.0 : 1.0;
    double dg = two * dense_hsys_norm, of = 0.0;
    for (int k = 0; k < S.Q; k++) {
      const double nm = S.n[k] - 1.0;
      if (S.lindblad) {
        dg += fabs(S.g2[k]) * nm * nm / 2.0 + fabs(S.g1[k]) * nm;
        of += fabs(S.g1off[k]) * nm;
      }
      of += two * control_amplitude_bound(k) * (k < (int)dense_hc_norm.size() ? dense_hc_norm[k] : 0.0);
    }
    *diag = dg;
    *off = of;
    return;
  }
  // max |h(I)| over the level combinations: a constant of the system (a million combinations for the reference's nlevels_32_32_32_32
  // case - 15 ms of host time per sweep when it was recomputed there)
  double hmax = hmax_cache;
  if (hmax < 0.0) {
    hmax = 0.0;
    std::vector<int> dg(S.Q, 0);
    for (long long I = 0; I < S.N; I++) {
      long long r = I;
      for (int k = 0; k < S.Q; k++) { dg[k] = (int)(r / S.post[k]); r %= S.post[k]; }
      double hd = 0.0;
      int pair = 0;
      for (int k = 0; k < S.Q; k++) {
        hd += S.detune[k] * dg[k] - S.xi[k] / 2.0 * dg[k] * (dg[k] - 1);
        for (int l = k + 1; l < S.Q; l++) hd -= S.xikl[pair++] * dg[k] * dg[l];
      }
      hmax = std::max(hmax, fabs(hd));
    }
    hmax_cache = hmax;
  }
  double dg = S.lindblad ? 2.0 * hmax : hmax, of = 0.0;
  for (int k = 0; k < S.Q; k++) {
    const double nm = S.n[k] - 1.0;
    if (S.lindblad) {
      dg += fabs(S.g2[k]) * nm * nm / 2.0 + fabs(S.g1[k]) * nm;
      of += fabs(S.g1off[k]) * nm;
    }
    // controls: each of the (2 or 4) ladder neighbours carries |q| + |p| times sqrt(level)
    const double amp = control_amplitude_bound(k);
    of += 2.0 * amp * (S.lindblad ? 2.0 : 1.0) * (sqrt(nm) + sqrt(std::max(nm - 1.0, 0.0)));
  }
  int pair = 0;
  for (int k = 0; k < S.Q; k++)
    for (int l = k + 1; l < S.Q; l++, pair++)
      of += fabs(S.J[pair]) * (S.lindblad ? 4.0 : 2.0) * sqrt((S.n[k] - 1.0) * (S.n[l] - 1.0)) * 2.0;
  *diag = dg;
  *off = of;
}

// Degree of the Neumann-polynomial right preconditioner of the global-memory GMRES (Team::gmres_g): poly_cur (tuned from sweep to
// sweep in forward_finish, starting at 6) where the series provably contracts, else 1 (plain KSPGMRES + PCNONE).  Criterion: Gershgorin bound of ||alpha M(t)||_inf <= 0.7 for every
// sub-step, from the system constants and the CURRENT control parameters (|p_k(t)|, |q_k(t)| <= sum over carriers of
// max |alpha^1| + max |alpha^2|: the quadratic B-splines are a partition of unity, src/controlbasis.cpp:81-96; pi-pulses
// by their amplitude).  The option gmres_poly overrides the degree (1 = never precondition).
int qd_handle::gmres_poly_degree() const {
  const int want = opts.gmres_poly > 0 ? opts.gmres_poly : poly_cur;  // tuned in forward_finish
  // only where the Krylov basis traffic is the cost (dim > 1024: the column / eight-elements-per-thread kernels); below
  // that plain GMRES keeps the oracle's iteration path, and with it results that agree far below the solver tolerance
  // [r6] ... except on the lean slot kernels (2^4 / 2^5 Lindblad, fp64): their Krylov solver takes the whole solve in ONE preconditioned
  // vector and one reduction (Team32::kry1, qd_q32.hip) - faster than the stationary iteration it is asked instead of
  const bool slot_kry = precision == QD_PRECISION_F64 && lean64_available(S, opts);
  if (want <= 1 || S.dense || (S.dim <= 1024 && !slot_kry)) return 1;
  double dg, of;
  row_bounds(&dg, &of);
  double amax = 0.0;
  for (double hh : sched_h) amax = std::max(amax, fabs(hh) / 2.0);
  // the lean column kernels precondition with the polynomial of the diagonal-split iteration [r6]: only the off-diagonal row sum has
  // to contract (ColTeam::kry_*, qd_col.hip)
  if (precision == QD_PRECISION_F64 && collean_available(S, opts) && S.N >= 44) return amax * of <= 0.7 ? want : 1;
  return amax * (dg + of) <= 0.7 ? want : 1;
}

// the lean column kernels: the stationary iterations, and [r6] the Krylov solver wherever the polynomial preconditioner is on (without
// it - KSPGMRES + PCNONE iteration for iteration - a gmres request stays on the general column kernel)
bool qd_handle::use_col(const qd::LaunchCfg& cfg) const {
  if (!(precision == QD_PRECISION_F64 && cfg.var == 9 && collean_available(S, opts) && sol.stepper != QD_STEPPER_EE)) return false;
  return !cfg.gmres || (cfg.gmres == 2 && !opts.no_col_krylov && gmres_poly_degree() > 1);
}

// linearsolver_type = gmres on the systems of the lean column kernels.  A Krylov basis of 57.6 KB vectors per initial condition has no
// room on the chip next to the exchange buffers (the column kernel's GMRES streams it through L2 / HBM: 8.6 x the algorithmic bytes,
// 3.8 x the time of the Neumann sweep on the 3 x 20 workload), while the diagonal-split stationary iteration contracts by
// ~alpha ||M - D|| per application and needs no vector besides the iterate.  Where that contraction is provably fast - Gershgorin
// bound alpha x (off-diagonal row sum) <= 0.3 for every sub-step - the request is served by that iteration under GMRES's own stopping
// rule: residual <= max(rtol ||b||, abstol) (KSPGMRES defaults as set in src/timestepper.cpp:541-550), checked through the bound
// ||b - (I - alpha M) y_m|| = ||(1 - alpha D)(y_{m+1} - y_m)|| <= kappa ||y_{m+1} - y_m||.  Same linear system, same tolerance, hence
// results that agree with GMRES at solver-tolerance level; the option gmres_split = 0 keeps the Krylov kernels.
bool qd_handle::gmres_as_split(const qd::LaunchCfg& cfg, double* kappa2) const {
  // ... and on the states beyond LDS (qd_big.h), where the Krylov basis streams through HBM (the reference's nlevels_32_32_32_32 case:
  // 12 vectors of 16 MB per initial condition) and the level energies of high levels dominate the row
  const bool col_ok = cfg.var == 9 && collean_available(S, opts), big_ok = cfg.var == 16 && !S.dense;
  if (precision != QD_PRECISION_F64 || !(col_ok || big_ok) || !cfg.gmres || sol.stepper == QD_STEPPER_EE || opts.gmres_split == 0)
    return false;
  double dg, of;
  row_bounds(&dg, &of);
  double amax = 0.0;
  for (double hh : sched_h) amax = std::max(amax, fabs(hh) / 2.0);
  if (kappa2) *kappa2 = (1.0 + amax * dg) * (1.0 + amax * dg);
  if (opts.gmres_split == 1) return true;
  return latched_substitution(1, amax * of);
}

// The gates depend on the CURRENT control parameters, and an optimiser's line search compares objectives far below the solver
// tolerance: a solver that changes between two evaluations shows up as a jump.  The decision is therefore latched per handle: taken at
// the first sweep (bound <= 0.3), kept while the stationary iteration still provably reaches the tolerance within its iteration cap,
// and given up for good - Krylov kernels from then on - the first time it does not.  The cap is three times linearsolver_maxiter
// (the reference's default 10 -> 30 applications), so "provably" is bound^cap <= 1e-13: 0.37 at the default, 0.6 from maxiter 20 on
// (never above 0.6); a small maxiter lowers the entry gate the same way.  At most one switch in the life of a handle;
// qd_set_option / qd_set_hamiltonian / qd_set_precision start over.
bool qd_handle::latched_substitution(int kind, double bound) const {
  const double cap = 3.0 * std::max(1, sol.maxiter);
  const double reach = std::pow(1e-13, 1.0 / cap);  // contraction per application that gets from ||b|| ~ 1 to 1e-13 within the cap
  const double enter = std::min(0.3, reach), keep = std::min(0.6, reach);
  if (!params_set) return bound <= enter;  // (a query before the first qd_set_params sees no controls: answer, but decide nothing)
  if (sub_latch == -1) {
    if (bound <= enter) sub_latch = kind;
    else if (kind == 2) sub_latch = 0;  // (the split gate is asked first and leaves the decision to the Neumann gate)
    return sub_latch == kind;
  }
  if (sub_latch != kind) return false;
  if (bound <= keep) return true;
  sub_latch = 0;
  return false;
}

// The same for every other kernel family: where the reference's own Neumann iteration provably contracts fast - Gershgorin bound
// alpha x (whole row sum) <= 0.3 for every sub-step - a gmres request is served by it.  Its update IS the residual of the previous
// iterate (r_m = b - (I - alpha M) y_m = y_{m+1} - y_m), so its stopping rule "update norm < abstol" is GMRES's "residual <= abstol"
// (the rtol ||b|| branch of KSP's rule can only stop GMRES earlier, i.e. less accurately); at such contraction both need the same ~4
// applications per step, and the stationary iteration has no orthogonalisation, no Hessenberg problem and one fp32 reduction per
// iteration instead of two fp64 ones (4-qubit system: 2.7 ms against 7.6 ms per 1000 steps, 2^5: 15 against 34).
bool qd_handle::gmres_as_neumann(const qd::LaunchCfg& cfg) const {
  if (!cfg.gmres || sol.stepper == QD_STEPPER_EE || opts.gmres_split != -1) return false;  // (gmres_split = 1 only forces the column path)
  double dg, of;
  row_bounds(&dg, &of);
  double amax = 0.0;
  for (double hh : sched_h) amax = std::max(amax, fabs(hh) / 2.0);
  return latched_substitution(2, amax * (dg + of));
}

// Diagonal-split Neumann iteration (qd_col.hip): same fixed point and stopping rule, the diagonal of M on the left-hand side.  It
// costs nothing per iteration, so "where it pays" is wherever the diagonal (level energies, decay) is a visible share of the row
// bound: alpha (diag + off) is the contraction bound of the plain iteration, alpha off that of the split one.
int qd_handle::neumann_split_on() const {
  if (opts.neumann_split >= 0) return opts.neumann_split;
  if (S.dense) return 0;
  double dg, of;
  row_bounds(&dg, &of);
  return dg >= 0.25 * of ? 1 : 0;
}

// Does the adjoint sweep that follows a forward sweep of nb states read the stored states x_n (SweepArgs::traj), or only the primal
// stages z?  States: explicit Euler (no stages), the dpdm penalty (second differences of x), the leakage penalty and the weighted-J
// penalty (their adjoints are functions of x_n; on the lean column kernels the weighted Jmeasure's adjoint is a constant per row) -
// and every kernel family whose adjoint kernel has not been written to do without (general / global-memory kernels).
bool qd_handle::adjoint_reads_states(int nb, const qd::DevTarget* tgp) const {
  if (sol.stepper == QD_STEPPER_EE || pen.gamma_penalty_dpdm > 1e-13) return true;
  // (the same selection as adjoint_launch makes: the adjoint flag passed through)
  LaunchCfg cfg = pick_config(S, nb, opts, sol.linsolve == QD_LINSOLVE_GMRES, /*adjoint=*/true);
  if (gmres_as_split(cfg, nullptr)) cfg.gmres = 0;
  else if (gmres_as_neumann(cfg)) cfg = pick_config(S, nb, opts, false, true);
  const bool pen_on = pen.gamma_penalty > 1e-13;
  const bool wj = pen_on && tgp && pen.penalty_param > 1e-13;
  bool leak = false;
  for (int k = 0; k < S.Q; k++)
    if (pen_on && S.ness[k] < S.n[k]) leak = true;
  const bool lean64 = cfg.var != 16 && lean64_available(S, opts);
  if (precision == QD_PRECISION_F32MIXED || lean64) return wj || leak;
  if (use_col(cfg)) return (wj && tgp->objective_type != QD_OBJ_JMEASURE) || leak;
  return true;
}

// weights of the weighted-J penalty, tabulated per time step (constants of the handle: time grid, Tfinal, optim_penalty_param)
int qd_handle::ensure_wj_weights() {
  if (!(pen.gamma_penalty > 1e-13 && pen.penalty_param > 1e-13)) return QD_OK;
  if (wjw_param == pen.penalty_param && d_wjw.p) return QD_OK;
  std::vector<double> w(tg.ntime);
  for (int n = 0; n < tg.ntime; n++) {
    const double a = ((n + 1) * tg.dt - dctl.Tfinal) / pen.penalty_param;
    w[n] = 1.0 / pen.penalty_param * exp(-(a * a));
  }
  int r;
  if ((r = d_wjw.ensure(w.size()))) return r;
  QD_HIP(hipMemcpyAsync(d_wjw.p, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice, stream));
  QD_HIP(hipStreamSynchronize(stream));  // (w is a local)
  wjw_param = pen.penalty_param;
  return QD_OK;
}

static void fill_sweep(const qd_handle* h, SweepArgs& a, int nb, const DevTarget* tg) {
  std::memset(&a, 0, sizeof a);
  a.S = h->S;
  if (tg) a.tg = *tg;
  a.ctl = h->d_table.p;
  a.cs = h->cs;
  a.nsub = h->nsub;
  a.nstages = h->nstages;
  a.ntime = h->tg.ntime;
  a.nb = nb;
  a.dt = h->tg.dt;
  a.Tfinal = h->dctl.Tfinal;
  a.stepper_ee = h->sol.stepper == QD_STEPPER_EE;
  a.linsolve = h->sol.linsolve;
  a.maxiter = h->sol.maxiter;
  a.abstol = h->sol.abstol;
  a.reltol = h->sol.reltol;
  a.inv_abs2 = 1.0 / (a.abstol * a.abstol);
  a.rel2 = (float)(a.reltol * a.reltol);
  a.gmres_poly = h->gmres_poly_degree();
  a.col_noskip = !h->opts.col_skip;
  a.kry_tau2 = h->opts.krylov_tau * h->opts.krylov_tau;
  a.kry_restart = h->opts.krylov_restart;
  a.nslice = 1;
  a.neumann_split = h->neumann_split_on();
  // penalties that need target data are only active when a target has been set
  a.gamma_penalty = h->pen.gamma_penalty;
  a.penalty_param = tg ? h->pen.penalty_param : 0.0;
  a.wjw = (tg && h->wjw_param == h->pen.penalty_param) ? h->d_wjw.p : nullptr;
  a.gamma_dpdm = h->pen.gamma_penalty_dpdm;
  a.leak_on = 0;
  for (int k = 0; k < h->S.Q; k++)
    if (h->S.ness[k] < h->S.n[k]) a.leak_on = 1;  // addLeakagePrevent, src/timestepper.cpp:28-32
}

// Time-sliced scheduling of a lean column sweep (qd_col.hip): scheduler words zeroed on the stream, carry buffer of the adjoint state.
// which = 0 forward, 1 adjoint (their scheduler words are separate: both sweeps may be queued before the first has run).
int qd_handle::arm_slices(qd::SweepArgs& a, int nb, int which) {
  a.nslice = col_slices(nb, tg.ntime, opts);
  a.sched = nullptr;
  a.sched_ticks = 0;
  if (a.nslice <= 1) {
    a.nslice = 1;
    return QD_OK;
  }
  int r;
  const size_t words = (size_t)nb + 2, dbl = (words + 1) / 2;
  if ((r = d_sched.ensure(2 * dbl)) || (r = h_sched.ensure(1))) return r;
  unsigned* p = reinterpret_cast<unsigned*>(d_sched.p + (which ? dbl : 0));
  QD_HIP(hipMemsetAsync(p, 0, sizeof(unsigned) * words, stream));
  a.sched = p;
  {  // how long a slice may wait for its predecessor: a healthy one takes a slice length of its own plus whatever the device spends
     // on the kernels of other processes that share it (several ranks on one GPU: QD_SHARE_GPUS, bench.py --dist-backend host)
    double s = opts.sched_wait_s;
    if (!(s > 0.0)) {
      const char* e = getenv("QD_DEVICE_SHARERS");
      const int sharers = e && atoi(e) > 1 ? atoi(e) : 1;
      const double steps = (double)tg.ntime / a.nslice;
      s = 4.0 * sharers * (steps > 1000.0 ? steps / 1000.0 : 1.0);
    }
    a.sched_ticks = (unsigned long long)(s * 1e8);
  }
  if (which) {
    if ((r = d_stash.ensure((size_t)nb * 2 * S.dim))) return r;
    a.stash = d_stash.p;
  }
  return QD_OK;
}

int qd_handle::forward_dev(const double* dx0, int nb, bool store, const DevTarget* tgp, double* energy) {
  int r;
  if ((r = forward_launch(dx0, nb, store, tgp))) return r;
  return forward_finish(energy);
}

// enqueue the whole forward sweep (tables, kernel, objective pieces, asynchronous result download) on the handle's stream
int qd_handle::forward_launch(const double* dx0, int nb, bool store, const DevTarget* tgp) {
  QD_HIP(qd::use_device(device));
  int r;
  if (pen.gamma_penalty > 1e-13 && pen.penalty_param > 1e-13 && !tgp)
    return fail(QD_ERR_STATE, "qd_forward: the weighted-J penalty (optim_penalty_param > 0) needs qd_set_target first");
  const size_t n = (size_t)nb * 2 * S.dim;
  if ((r = d_xT.ensure(n)) || (r = d_res.ensure((size_t)6 * nb + 1))) return r;
  d_pen = d_res.p;
  d_dpdm = d_res.p + nb;
  d_out4 = d_res.p + 2 * (size_t)nb;
  d_napply = reinterpret_cast<unsigned long long*>(d_res.p + 6 * (size_t)nb);
  napply_zeroed = false;
  if ((r = refresh_tables())) return r;
  traj_valid = false;
  const bool full = store && stores_full(nb, tgp);
  if (store) {
    size_t nt;
    traj_doubles(nb, &nt);
    if (full && (r = d_traj.ensure(nt))) return r;
    if (ztraj_doubles(nb) && (r = d_ztraj.ensure(ztraj_doubles(nb)))) return r;
  }
  {
    LaunchCfg c0 = pick_config(S, nb, opts, sol.linsolve == QD_LINSOLVE_GMRES);
    if (c0.var == 16 && (r = ensure_big(nb))) return r;
  }
  if ((r = ensure_wj_weights())) return r;
  SweepArgs a;
  fill_sweep(this, a, nb, tgp);
  last_poly = a.gmres_poly;
  fwd_poly = a.gmres_poly;
  a.x0 = dx0;
  a.xT = d_xT.p;
  a.traj = full ? d_traj.p : nullptr;
  a.ztraj = store && ztraj_doubles(nb) ? d_ztraj.p : nullptr;
  a.pen_out = d_pen;
  a.dpdm_out = d_dpdm;
  a.napply = d_napply;
  LaunchCfg cfg = pick_config(S, nb, opts, sol.linsolve == QD_LINSOLVE_GMRES);
  last_var = cfg.var;
  last_team = cfg.var == 16 && cfg.team > 1 && precision != QD_PRECISION_F32MIXED ? cfg.team : 1;
  a.use_gmres = cfg.gmres;
  last_solver = sol.stepper == QD_STEPPER_EE ? QD_SOLVER_NONE : cfg.gmres ? QD_SOLVER_KRYLOV : QD_SOLVER_NEUMANN;
  if (gmres_as_split(cfg, &a.kappa2)) {  // GMRES request served by the diagonal-split iteration of the lean column kernels
    last_solver = QD_SOLVER_GMRES_AS_SPLIT;
    a.standin_tau2 = (float)(opts.standin_tau * opts.standin_tau);
    cfg.gmres = 0;
    a.use_gmres = 0;
    a.neumann_split = 1;
    a.stop_residual = 1;
    a.maxiter *= 3;  // (linearsolver_maxiter caps KRYLOV iterations: at the gate's contraction bound 0.3 thirty stationary iterations
                     // reach what ten GMRES iterations reach in the worst case; the stopping rule ends the loop long before)
    last_poly = 1;
    cfg.lds = pick_config(S, nb, opts, false).lds;
  } else if (gmres_as_neumann(cfg)) {
    last_solver = QD_SOLVER_GMRES_AS_NEUMANN;
    a.standin_tau2 = (float)(opts.standin_tau * opts.standin_tau);
    cfg = pick_config(S, nb, opts, false);
    a.use_gmres = 0;
    a.gmres_poly = 1;
    a.maxiter *= 3;
    last_poly = 1;
    last_team = cfg.var == 16 && cfg.team > 1 && precision != QD_PRECISION_F32MIXED ? cfg.team : 1;
  }
  // (the fp32-mixed GMRES and the fp64 one of the lean slot kernels always keep their basis in global memory)
  if (cfg.gmres == 2 || (cfg.gmres && (precision == QD_PRECISION_F32MIXED || (cfg.var != 16 && lean64_available(S, opts) && sol.stepper != QD_STEPPER_EE)))) {
    if ((r = d_kry.ensure(use_col(cfg) ? col_krylov_doubles(nb, col_slices(nb, tg.ntime, opts)) : krylov_doubles(S, nb)))) return r;
    a.kry = d_kry.p;
  }
  if ((r = check_cfg(cfg))) return r;
  if (!napply_zeroed) QD_HIP(hipMemsetAsync(d_napply, 0, sizeof(unsigned long long), stream));
  QD_HIP(hipEventRecord(ev0, stream));
  // (2^4 / 2^5 Lindblad: the lean slot kernels, coupled or not; their Krylov solver keeps its basis in global memory [r6])
  const bool lean64 = cfg.var != 16 && lean64_available(S, opts) && sol.stepper != QD_STEPPER_EE;
  if (a.ztraj) ztraj_fmt = precision == QD_PRECISION_F32MIXED ? 1 : lean64 ? 2 : use_col(cfg) ? 3 : 0;
  if (precision == QD_PRECISION_F32MIXED && S.hasJ && a.use_gmres)
    return fail(QD_ERR_UNSUPPORTED, "fp32-mixed sweeps of a system with dipole-dipole coupling: the Krylov kernels are not built (option gmres_split = 0); linearsolver_type = gmres is served by the stationary iteration where it contracts");
  if (precision == QD_PRECISION_F32MIXED) QD_HIP(launch_forward_f32(a, opts, stream));
  else if (lean64) QD_HIP(launch_forward_lean64(a, opts, stream));
  else if (use_col(cfg)) {
    if ((r = arm_slices(a, nb, 0))) return r;
    QD_HIP(launch_forward_col(a, opts, stream));
  } else QD_HIP(launch_forward(a, cfg, stream));
  QD_HIP(hipEventRecord(ev1, stream));
  if (a.sched) QD_HIP(hipMemcpyAsync(h_sched.p, a.sched + 1, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
  sliced_fwd = a.sched != nullptr;
  if (tgp) QD_HIP(launch_objective(S, *tgp, d_xT.p, nb, d_out4, stream));
  // every result of the sweep in one pinned buffer, one synchronisation
  if ((r = h_res.ensure((size_t)6 * nb + 1))) return r;
  QD_HIP(hipMemcpyAsync(h_res.p, d_res.p, sizeof(double) * (6 * (size_t)nb + 1), hipMemcpyDeviceToHost, stream));
  last_nb = nb;
  pending_store = store;
  pending_full = full;
  return QD_OK;
}

// wait for the sweep enqueued by forward_launch (and whatever the caller enqueued behind it) and collect its results
int qd_handle::forward_finish(double* energy) {
  const int nb = last_nb;
  const bool store = pending_store;
  QD_HIP(hipStreamSynchronize(stream));
  if (sliced_fwd && reinterpret_cast<const unsigned*>(h_sched.p)[0] != 0u)
    return fail(QD_ERR_DEVICE, "forward sweep: a time slice waited longer than its limit for its predecessor (time-sliced scheduling: options col_slices, sched_wait_s)");
  unsigned long long nap = 0;
  std::memcpy(&nap, h_res.p + 6 * (size_t)nb, sizeof nap);
  float ms = 0.f;
  QD_HIP(hipEventElapsedTime(&ms, ev0, ev1));
  last_fwd_ms = accumulate_fwd_ms ? last_fwd_ms + ms : ms;
  last_mean_applies = (double)nap / ((double)nb * (double)tg.ntime);  // per TIME step (all stages of a composite step)
  // Degree p of the polynomial preconditioner (Team::gmres_g): a solve costs 1 + k p applications plus k rounds of
  // orthogonalisation, reductions and basis traffic (k = Krylov vectors; on the 3x20 workload one such round costs as much as five
  // applications), so the best p is the smallest one for which (almost) every solve needs a single Krylov vector.  k is known after
  // every forward sweep: bracket p between the largest degree seen with k > 1 and the smallest seen with k = 1, bisect, stay.
  // Once the bracket has closed (or after eight tuning sweeps) the degree is FROZEN for the life of the handle: from then on two
  // evaluations at the same parameters take the same GMRES path and are bit-identical (a line search compares objectives far below
  // the solver tolerance).  The option gmres_poly fixes the degree from the first sweep on.
  // A frozen degree is tuned at the parameters of the first sweeps; as an optimisation drives the amplitudes up it can become too low
  // (k > 1 Krylov vectors per solve, each costing a round of orthogonalisation and basis traffic).  Three consecutive sweeps with
  // k > 1.5 reopen the tuning - upwards only, so that the evaluations in between stay comparable; qd_set_option(gmres_poly, auto)
  // starts it over.
  if (sol.linsolve == QD_LINSOLVE_GMRES && last_poly > 1 && sol.stepper != QD_STEPPER_EE && opts.gmres_poly == 0 && poly_frozen) {
    const double k = ((double)nap / ((double)nb * (double)nsub) - 1.0) / last_poly;
    poly_slow = k > 1.5 ? poly_slow + 1 : 0;
    if (poly_slow >= 3) {
      poly_frozen = false;
      poly_slow = 0;
      poly_steps = 4;  // (at most four more tuning sweeps)
      poly_lo = last_poly;
      poly_hi = 0;
    }
  }
  if (sol.linsolve == QD_LINSOLVE_GMRES && last_poly > 1 && sol.stepper != QD_STEPPER_EE && opts.gmres_poly == 0 && !poly_frozen) {
    const double per_solve = (double)nap / ((double)nb * (double)nsub);
    const double k = (per_solve - 1.0) / last_poly;
    if (k > 1.02) {
      poly_lo = std::max(poly_lo, last_poly);
      if (poly_hi && poly_hi <= poly_lo) poly_hi = 0;  // (the controls have moved: the old bracket no longer holds)
      poly_cur = poly_hi ? (poly_lo + poly_hi + 1) / 2 : std::min(32, (int)std::ceil(last_poly * k));
      if (poly_cur <= poly_lo) poly_cur = std::min(32, poly_lo + 1);
    } else {
      poly_hi = poly_hi ? std::min(poly_hi, last_poly) : last_poly;
      if (poly_lo >= poly_hi) poly_lo = 1;
      poly_cur = poly_hi - poly_lo > 1 ? (poly_lo + poly_hi) / 2 : poly_hi;
    }
    poly_steps++;
    if ((poly_hi && poly_hi - poly_lo <= 1) || poly_steps >= 8 || poly_cur >= 32) {
      if (poly_hi) poly_cur = poly_hi;
      poly_frozen = true;
    }
  }
  last_nb = nb;
  traj_valid = store;
  traj_full = store && pending_full;
  pending_store = false;
  pending_full = false;
  if (energy) *energy = energy_penalty_host();
  return QD_OK;
}

extern "C" int qd_forward(qd_handle* h, const double* x0, int nb, int store_trajectory, qd_forward_out* out) {
  if (!h || !x0 || nb < 1) return fail(QD_ERR_INVALID, "qd_forward: bad argument");
  if (h->target_set && h->target_nb != nb) return fail(QD_ERR_INVALID, "qd_forward: batch size differs from qd_set_target");
  QD_HIP(qd::use_device(h->device));
  const size_t n = (size_t)nb * 2 * h->S.dim;
  int r;
  if ((r = h->d_x0.ensure(n))) return r;
  QD_HIP(hipMemcpyAsync(h->d_x0.p, x0, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  double energy = 0.0;
  if ((r = h->forward_dev(h->d_x0.p, nb, store_trajectory != 0, h->target_set ? &h->dtg : nullptr, &energy))) return r;
  if (out) {
    if (out->final_states) QD_HIP(hipMemcpy(out->final_states, h->d_xT.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (out->penalty_integral) std::memcpy(out->penalty_integral, h->res_pen(), sizeof(double) * nb);
    if (out->penalty_dpdm) std::memcpy(out->penalty_dpdm, h->res_dpdm(), sizeof(double) * nb);
    if (out->energy_penalty) *out->energy_penalty = energy;
    if (h->target_set && (out->J_re || out->J_im || out->fid_re || out->fid_im)) {
      const double* o4 = h->res_out4();
      for (int b = 0; b < nb; b++) {
        if (out->J_re) out->J_re[b] = o4[4 * b];
        if (out->J_im) out->J_im[b] = o4[4 * b + 1];
        if (out->fid_re) out->fid_re[b] = o4[4 * b + 2];
        if (out->fid_im) out->fid_im[b] = o4[4 * b + 3];
      }
    }
  }
  return QD_OK;
}

extern "C" int qd_get_state(qd_handle* h, int timestep, double* x) {
  if (!h || !x) return fail(QD_ERR_INVALID, "qd_get_state: null argument");
  if (!h->traj_valid || !h->traj_full) return fail(QD_ERR_STATE, "qd_get_state: no stored trajectory (call qd_forward with store_trajectory=1)");
  if (timestep < 0 || timestep > h->tg.ntime) return fail(QD_ERR_INVALID, "qd_get_state: time step out of range");
  QD_HIP(qd::use_device(h->device));
  const size_t n = (size_t)h->last_nb * 2 * h->S.dim;
  if (h->precision == QD_PRECISION_F32MIXED) {  // stored as interleaved float2 per element
    std::vector<float> tmp(n);
    QD_HIP(hipMemcpy(tmp.data(), reinterpret_cast<const float*>(h->d_traj.p) + (size_t)timestep * h->nstages * n, sizeof(float) * n, hipMemcpyDeviceToHost));
    const size_t dim = h->S.dim;
    for (int b = 0; b < h->last_nb; b++)
      for (size_t e = 0; e < dim; e++) {
        x[(size_t)b * 2 * dim + e] = tmp[((size_t)b * dim + e) * 2];
        x[(size_t)b * 2 * dim + dim + e] = tmp[((size_t)b * dim + e) * 2 + 1];
      }
    return QD_OK;
  }
  QD_HIP(hipMemcpy(x, h->d_traj.p + (size_t)timestep * h->nstages * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  return QD_OK;
}

extern "C" int qd_get_observables(qd_handle* h, int stride, double* expected, double* population, double* expected_composite,
                                  double* population_composite) {
  if (!h || stride < 1) return fail(QD_ERR_INVALID, "qd_get_observables: bad argument");
  if (!h->traj_valid || !h->traj_full) return fail(QD_ERR_STATE, "qd_get_observables: no stored trajectory (forward sweep with store_trajectory=1 first)");
  QD_HIP(qd::use_device(h->device));
  const DevSys& S = h->S;
  const int nb = h->last_nb, nout = h->tg.ntime / stride + 1;
  int nlev = 0;
  for (int k = 0; k < S.Q; k++) nlev += S.n[k];
  const size_t per = (size_t)nout * nb;
  const size_t ne = expected ? per * S.Q : 0, np = population ? per * nlev : 0, nc = expected_composite ? per : 0,
               npc = population_composite ? per * S.N : 0;
  DBuf d;
  int r;
  if ((r = d.ensure(ne + np + nc + npc + 1))) return r;
  double *de = d.p, *dp = de + ne, *dc = dp + np, *dpc = dc + nc;
  hipError_t e = launch_observables(S, h->d_traj.p, h->precision == QD_PRECISION_F32MIXED, nb, h->nstages, stride, nout, nlev,
                                    expected ? de : nullptr, population ? dp : nullptr, expected_composite ? dc : nullptr,
                                    population_composite ? dpc : nullptr, h->stream);
  if (e == hipSuccess && ne) e = hipMemcpyAsync(expected, de, sizeof(double) * ne, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && np) e = hipMemcpyAsync(population, dp, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && nc) e = hipMemcpyAsync(expected_composite, dc, sizeof(double) * nc, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && npc) e = hipMemcpyAsync(population_composite, dpc, sizeof(double) * npc, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  d.release();
  if (e != hipSuccess) {
    set_error(std::string("qd_get_observables: ") + hipGetErrorString(e));
    return QD_ERR_DEVICE;
  }
  return QD_OK;
}

int qd_handle::adjoint_dev(const double* dxbarT, const double* djbar, int nb, const DevTarget* tgp, bool accumulate) {
  int r;
  if ((r = adjoint_launch(dxbarT, djbar, nb, tgp, accumulate))) return r;
  return adjoint_finish(accumulate);
}

int qd_handle::adjoint_launch(const double* dxbarT, const double* djbar, int nb, const DevTarget* tgp, bool accumulate) {
  QD_HIP(qd::use_device(device));
  if (!(traj_valid || pending_store) || last_nb != nb) return fail(QD_ERR_STATE, "qd_adjoint: needs a forward sweep of the same batch with store_trajectory=1");
  if (has_pipulse) return fail(QD_ERR_UNSUPPORTED, "qd_adjoint: derivative of pi-pulses is not implemented in the reference (src/oscillator.cpp:373-378)");
  if (has_ampbasis) return fail(QD_ERR_UNSUPPORTED, "qd_adjoint: the spline_amplitude parameterisation has no gradient in the reference (src/oscillator.cpp:350-356)");
  int r;
  const size_t ncol = (size_t)nsub * 2 * S.Q;
  if ((r = d_coeff.ensure((size_t)nb * ncol)) || (r = d_coeffsum.ensure(ncol))) return r;
  {
    LaunchCfg c0 = pick_config(S, nb, opts, sol.linsolve == QD_LINSOLVE_GMRES, true);
    if (c0.var == 16 && (r = ensure_big(nb))) return r;
  }
  if ((r = ensure_wj_weights())) return r;
  SweepArgs a;
  fill_sweep(this, a, nb, tgp);
  // (the adjoint sweep of an evaluation runs on the degree its forward sweep ran on: the tuner moves between the two, forward_finish)
  if (opts.gmres_poly == 0 && fwd_poly > 1 && a.gmres_poly > 1) a.gmres_poly = fwd_poly;
  last_poly = a.gmres_poly;
  const bool have_states = pending_store ? pending_full : traj_full;
  if (!have_states && adjoint_reads_states(nb, tgp))
    return fail(QD_ERR_STATE, "qd_adjoint: the forward sweep stored the primal stages only, this adjoint sweep needs the states");
  a.traj = have_states ? d_traj.p : nullptr;
  a.ztraj = ztraj_doubles(nb) ? d_ztraj.p : nullptr;
  a.xbarT = dxbarT;
  a.jbar = djbar;
  a.coeff = d_coeff.p;
  LaunchCfg cfg = pick_config(S, nb, opts, sol.linsolve == QD_LINSOLVE_GMRES, /*adjoint=*/true);
  last_team = cfg.var == 16 && cfg.team > 1 && precision != QD_PRECISION_F32MIXED ? cfg.team : 1;
  a.use_gmres = cfg.gmres;
  last_solver = sol.stepper == QD_STEPPER_EE ? QD_SOLVER_NONE : cfg.gmres ? QD_SOLVER_KRYLOV : QD_SOLVER_NEUMANN;
  if (gmres_as_split(cfg, &a.kappa2)) {
    last_solver = QD_SOLVER_GMRES_AS_SPLIT;
    a.standin_tau2 = (float)(opts.standin_tau * opts.standin_tau);
    cfg.gmres = 0;
    a.use_gmres = 0;
    a.neumann_split = 1;
    a.stop_residual = 1;
    a.maxiter *= 3;
    last_poly = 1;
    cfg.lds = pick_config(S, nb, opts, false, true).lds;
  } else if (gmres_as_neumann(cfg)) {
    last_solver = QD_SOLVER_GMRES_AS_NEUMANN;
    a.standin_tau2 = (float)(opts.standin_tau * opts.standin_tau);
    cfg = pick_config(S, nb, opts, false, true);
    a.use_gmres = 0;
    a.gmres_poly = 1;
    a.maxiter *= 3;
    last_poly = 1;
    last_team = cfg.var == 16 && cfg.team > 1 && precision != QD_PRECISION_F32MIXED ? cfg.team : 1;
  }
  // (the fp32-mixed GMRES and the fp64 one of the lean slot kernels always keep their basis in global memory)
  if (cfg.gmres == 2 || (cfg.gmres && (precision == QD_PRECISION_F32MIXED || (cfg.var != 16 && lean64_available(S, opts) && sol.stepper != QD_STEPPER_EE)))) {
    if ((r = d_kry.ensure(use_col(cfg) ? col_krylov_doubles(nb, col_slices(nb, tg.ntime, opts)) : krylov_doubles(S, nb)))) return r;
    a.kry = d_kry.p;
  }
  if ((r = check_cfg(cfg))) return r;
  QD_HIP(hipEventRecord(ev2, stream));
  // (2^4 / 2^5 Lindblad: the lean slot kernels, coupled or not; their Krylov solver keeps its basis in global memory [r6])
  const bool lean64 = cfg.var != 16 && lean64_available(S, opts) && sol.stepper != QD_STEPPER_EE;
  if (a.ztraj && ztraj_fmt != (precision == QD_PRECISION_F32MIXED ? 1 : lean64 ? 2 : use_col(cfg) ? 3 : 0))
    return fail(QD_ERR_STATE, "qd_adjoint: the primal stages were stored by another kernel family (options or precision changed since the forward sweep): repeat the forward sweep");
  if (precision == QD_PRECISION_F32MIXED && S.hasJ && a.use_gmres)
    return fail(QD_ERR_UNSUPPORTED, "fp32-mixed sweeps of a system with dipole-dipole coupling: the Krylov kernels are not built (option gmres_split = 0)");
  if (precision == QD_PRECISION_F32MIXED) QD_HIP(launch_adjoint_f32(a, opts, stream));
  else if (lean64) QD_HIP(launch_adjoint_lean64(a, opts, stream));
  else if (use_col(cfg)) {
    if ((r = arm_slices(a, nb, 1))) return r;
    QD_HIP(launch_adjoint_col(a, opts, stream));
  } else QD_HIP(launch_adjoint(a, cfg, stream));
  QD_HIP(hipEventRecord(ev3, stream));
  if (a.sched) QD_HIP(hipMemcpyAsync(reinterpret_cast<unsigned*>(h_sched.p) + 1, a.sched + 1, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
  sliced_adj = a.sched != nullptr;
  QD_HIP(launch_reduce_coeff(d_coeff.p, nb, (int)ncol, d_coeffsum.p, accumulate ? 1 : 0, stream));
  return QD_OK;
}

int qd_handle::adjoint_finish(bool accumulate) {
  QD_HIP(hipStreamSynchronize(stream));
  if (sliced_adj && reinterpret_cast<const unsigned*>(h_sched.p)[1] != 0u)
    return fail(QD_ERR_DEVICE, "adjoint sweep: a time slice waited longer than its limit for its predecessor (time-sliced scheduling: options col_slices, sched_wait_s)");
  float ms = 0.f;
  QD_HIP(hipEventElapsedTime(&ms, ev2, ev3));
  last_adj_ms = accumulate ? last_adj_ms + ms : ms;
  // explicit Euler + Schroedinger: the kernel has replaced the stored forward states by the reference's backward
  // recomputation chain (k_adjoint), so the buffer no longer is the forward trajectory
  if (sol.stepper == QD_STEPPER_EE && !S.lindblad) traj_valid = false;
  return QD_OK;
}

int qd_handle::gradient_launch(double ebar, double* dgrad) {
  QD_HIP(qd::use_device(device));
  if (ndesign == 0) return QD_OK;
  const int nsub_flag = sol.stepper == QD_STEPPER_EE ? -nsub : nsub;
  QD_HIP(launch_grad(dctl, d_params.p, d_table.p, cs, nsub_flag, d_coeffsum.p, d_etable.p, tg.ntime, ebar, dgrad, ndesign, stream));
  return QD_OK;
}

int qd_handle::gradient_from_coeffs(double ebar, double* grad) {
  QD_HIP(qd::use_device(device));
  int r;
  if (ndesign == 0) return QD_OK;
  if ((r = d_grad.ensure(ndesign))) return r;
  if ((r = gradient_launch(ebar, d_grad.p))) return r;
  QD_HIP(hipMemcpyAsync(grad, d_grad.p, sizeof(double) * ndesign, hipMemcpyDeviceToHost, stream));
  QD_HIP(hipStreamSynchronize(stream));
  return QD_OK;
}

extern "C" int qd_adjoint(qd_handle* h, const double* xbarT, const double* jbar, int nb, double* grad) {
  if (!h || !xbarT || !jbar || !grad || nb < 1) return fail(QD_ERR_INVALID, "qd_adjoint: bad argument");
  QD_HIP(qd::use_device(h->device));
  const size_t n = (size_t)nb * 2 * h->S.dim;
  int r;
  if ((r = h->d_xbar.ensure(n)) || (r = h->d_jbar.ensure((size_t)3 * nb))) return r;
  QD_HIP(hipMemcpyAsync(h->d_xbar.p, xbarT, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  QD_HIP(hipMemcpyAsync(h->d_jbar.p, jbar, sizeof(double) * 3 * nb, hipMemcpyHostToDevice, h->stream));
  if ((r = h->adjoint_dev(h->d_xbar.p, h->d_jbar.p, nb, h->target_set ? &h->dtg : nullptr, false))) return r;
  double ebar = 0.0;
  if (h->pen.gamma_penalty_energy > 1e-13)
    for (int b = 0; b < nb; b++) ebar += jbar[3 * b + 2];
  return h->gradient_from_coeffs(ebar, grad);
}

extern "C" double qd_last_mean_applies(const qd_handle* h) { return h ? h->last_mean_applies : 0.0; }
extern "C" double qd_last_forward_ms(const qd_handle* h) { return h ? h->last_fwd_ms : 0.0; }
extern "C" double qd_last_adjoint_ms(const qd_handle* h) { return h ? h->last_adj_ms : 0.0; }
extern "C" int qd_last_team(const qd_handle* h) { return h ? h->last_team : QD_ERR_INVALID; }
extern "C" int qd_last_solver(const qd_handle* h) { return h ? h->last_solver : QD_ERR_INVALID; }

// Measurement hook ("settle the MFMA question"): nrep chained applications of the forward operator
// to nb copies... of a batch of states with the fp32 stencil kernel (mfma = 0) or with the dense Kronecker-factor
// product on the fp32 matrix cores (mfma = 1; 2^5 Lindblad only).  y receives the result of the chain, *ms the device time.
extern "C" int qd_bench_apply_f32(qd_handle* h, double t, const double* x, double* y, int nb, int nrep, int mfma, double* ms) {
  if (!h || !x || !y || nb < 1 || nrep < 1 || !ms) return fail(QD_ERR_INVALID, "qd_bench_apply_f32: bad argument");
  const DevSys& S = h->S;
  bool qubits = S.lindblad && !S.dense && !S.hasJ && (S.Q == 4 || S.Q == 5);
  for (int k = 0; k < S.Q; k++) qubits = qubits && S.n[k] == 2;
  if (!qubits || (mfma && S.Q != 5)) return fail(QD_ERR_UNSUPPORTED, "qd_bench_apply_f32: all-qubit Lindblad systems with 4 or 5 oscillators (MFMA: 5)");
  QD_HIP(qd::use_device(h->device));
  const size_t n = (size_t)nb * 2 * S.dim;
  int r;
  if ((r = h->d_x0.ensure(n)) || (r = h->d_y.ensure(n))) return r;
  const double tt[2] = {t, 0.0};
  QD_HIP(hipMemcpyAsync(h->d_onetime.p, tt, sizeof tt, hipMemcpyHostToDevice, h->stream));
  QD_HIP(launch_controls(h->dctl, h->d_params.p, h->d_onetime.p, h->d_onetime.p + 1, 1, h->d_onerow.p, h->cs, h->stream));
  QD_HIP(hipMemcpyAsync(h->d_x0.p, x, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  QD_HIP(launch_apply_f32(S, h->d_onerow.p, 0, h->d_x0.p, h->d_y.p, nb, nrep, mfma, h->opts, h->stream));  // warm-up
  QD_HIP(hipEventRecord(h->ev0, h->stream));
  QD_HIP(launch_apply_f32(S, h->d_onerow.p, 0, h->d_x0.p, h->d_y.p, nb, nrep, mfma, h->opts, h->stream));
  QD_HIP(hipEventRecord(h->ev1, h->stream));
  QD_HIP(hipMemcpyAsync(y, h->d_y.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  QD_HIP(hipStreamSynchronize(h->stream));
  float m = 0.f;
  QD_HIP(hipEventElapsedTime(&m, h->ev0, h->ev1));
  *ms = m;
  return QD_OK;
}
