// Objective level of the C ABI: OptimProblem::evalF / evalGradF over the local shard of initial
// conditions (src/optimproblem.cpp:224-538), with the host-side pieces of OptimTarget
// (initial-condition families src/optimtarget.cpp:450-698, target states :701-708, finalizeJ :864-897)
// and Gate (rotation + lifting + V rho V^dagger, src/gate.cpp:88-283).  Set-up runs once on the
// host; initial/target states then stay resident in HBM and every sweep runs in the HIP kernels.
#include <algorithm>
#include <cmath>
#include <complex>

#include "qd_handle.h"

using namespace qd;
typedef std::complex<double> cplx;

static int fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}

struct qd_optim {
  qd_handle* h = nullptr;
  int rank = 0, nranks = 1, ninit = 0, nlocal = 0, first = 0;
  int initcond_type = 0, target_type = 0, objective_type = 0, purestate_id = -1;
  std::vector<int> init_ids;
  std::vector<double> rho0_fixed, target_fixed;  // PURE/FROMFILE/ENSEMBLE initial state; FROMFILE target
  std::vector<cplx> V;                           // rotated + lifted gate V_f (N x N row-major)
  std::vector<double> weights;                   // beta_i, normalised, global
  std::vector<double> alpha0;
  double gamma_tik = 0.0, gamma_var = 0.0, ebar = 0.0;
  qd_penalty pen{};
  // device-resident batch
  DBuf d_x0, d_tgt, d_pur, d_rbib, d_jbar, d_xbar;
  DBuf d_w, d_red;   // multi-GPU path: beta_i of the local shard; [7 sums | ndesign gradient] reduced in place by RCCL
  HBuf h_red;
  hipEvent_t evr[4] = {nullptr, nullptr, nullptr, nullptr};  // brackets of the two collectives
  DevTarget tg{};
  std::vector<int> init_id;  // output-file ids of the local initial conditions
  // state between forward_local and adjoint_local
  std::vector<double> last_alpha;
  bool stored = false, forward_done = false;
  int last_chunks = 1;  // chunks of the last gradient evaluation (1 = the shard's trajectory fitted)
  int dist_fits = -1;  // qd_optim_evalGradF_dist: -1 undecided, 1 fused device path, 0 host-staged fallback (decided collectively)
};

// ---- index helpers (src/util.cpp:150-278) ------------------------------------------------------
static int map_ess_to_full(const DevSys& S, int i) {
  int id = 0, index = i;
  for (int k = 0; k < S.Q - 1; k++) {
    int postdim = 1, postdim_ess = 1;
    for (int j = k + 1; j < S.Q; j++) {
      postdim *= S.n[j];
      postdim_ess *= S.ness[j];
    }
    id += (index / postdim_ess) * postdim;
    index %= postdim_ess;
  }
  return id + index;
}
static int map_full_to_ess(const DevSys& S, int i) {
  int id = 0, index = i;
  for (int k = 0; k < S.Q; k++) {
    int postdim = 1, postdim_ess = 1;
    for (int j = k + 1; j < S.Q; j++) {
      postdim *= S.n[j];
      postdim_ess *= S.ness[j];
    }
    const int iblock = index / postdim;
    index %= postdim;
    if (iblock >= S.ness[k]) return -1;
    id += iblock * postdim_ess;
  }
  return id;
}
static inline int vec_id(int row, int col, int N) { return row + col * N; }

// ---- gate (src/gate.cpp:88-249) ------------------------------------------------------------------
static void build_gate(const qd_handle* h, const qd_objective* ob, std::vector<cplx>& V) {
  const DevSys& S = h->S;
  const int de = h->dim_ess, N = S.N;
  std::vector<cplx> ve((size_t)de * de);
  for (int row = 0; row < de; row++) {
    int r = row;
    double freq = 0.0;
    for (int k = 0; k < S.Q; k++) {
      int dim_post = 1;
      for (int j = k + 1; j < S.Q; j++) dim_post *= S.ness[j];
      freq += (r / dim_post) * 2.0 * M_PI * ob->gate_rot_freq[k];
      r %= dim_post;
    }
    const double ra = cos(freq * h->dctl.Tfinal), rb = sin(freq * h->dctl.Tfinal);
    for (int c = 0; c < de; c++) {
      const double a = ob->gate_re[row * de + c], b = ob->gate_im ? ob->gate_im[row * de + c] : 0.0;
      ve[(size_t)row * de + c] = cplx(ra * a - rb * b, ra * b + rb * a);
    }
  }
  V.assign((size_t)N * N, cplx(0.0, 0.0));
  for (int rf = 0; rf < N; rf++) {
    const int re = map_full_to_ess(S, rf);
    if (re < 0) {  // identity on guard levels
      V[(size_t)rf * N + rf] = 1.0;
      continue;
    }
    for (int ce = 0; ce < de; ce++) V[(size_t)rf * N + map_ess_to_full(S, ce)] = ve[(size_t)re * de + ce];
  }
}

// Gate::applyGate (src/gate.cpp:260-283): V psi (Schroedinger) or vec(V rho V^dagger) (Lindblad)
static void apply_gate(const DevSys& S, const std::vector<cplx>& V, const double* x, double* out) {
  const int N = S.N, dim = S.dim;
  if (!S.lindblad) {
    for (int r = 0; r < N; r++) {
      cplx acc = 0.0;
      for (int c = 0; c < N; c++) acc += V[(size_t)r * N + c] * cplx(x[c], x[c + dim]);
      out[r] = acc.real();
      out[r + dim] = acc.imag();
    }
    return;
  }
  std::vector<cplx> T((size_t)N * N, cplx(0.0, 0.0));
  for (int r = 0; r < N; r++)
    for (int k = 0; k < N; k++) {
      const cplx v = V[(size_t)r * N + k];
      if (v == cplx(0.0, 0.0)) continue;
      for (int c = 0; c < N; c++) T[r + (size_t)c * N] += v * cplx(x[k + c * N], x[k + c * N + dim]);
    }
  std::vector<cplx> R((size_t)N * N, cplx(0.0, 0.0));
  for (int c = 0; c < N; c++)
    for (int k = 0; k < N; k++) {
      const cplx v = std::conj(V[(size_t)c * N + k]);
      if (v == cplx(0.0, 0.0)) continue;
      for (int r = 0; r < N; r++) R[r + (size_t)c * N] += T[r + (size_t)k * N] * v;
    }
  for (int i = 0; i < dim; i++) {
    out[i] = R[i].real();
    out[i + dim] = R[i].imag();
  }
}

// ---- initial conditions (src/optimtarget.cpp:450-698); returns the output-file id -----------------
static int prepare_initial_state(const qd_optim* o, int iinit, double* rho0) {
  const DevSys& S = o->h->S;
  const int dim = S.dim, N = S.N, de = o->h->dim_ess, ninit = o->ninit;
  std::fill(rho0, rho0 + 2 * dim, 0.0);
  int id = 0;
  switch (o->initcond_type) {
    case QD_INIT_PURE:
    case QD_INIT_FROMFILE:
    case QD_INIT_ENSEMBLE:
      std::copy(o->rho0_fixed.begin(), o->rho0_fixed.end(), rho0);
      break;
    case QD_INIT_PERFORMANCE:  // incl. the index quirk of the Lindblad branch (:473-477)
      for (int i = 0; i < N; i++) {
        if (!S.lindblad) rho0[i] = rho0[i + dim] = 1. / sqrt(2. * N);
        else rho0[i] = 1. / N;
      }
      break;
    case QD_INIT_THREESTATES:
      if (iinit == 0) {
        id = 1;
        for (int i = 0; i < N; i++) rho0[vec_id(i, i, N)] = 2. * (N - i) / ((double)N * (N + 1));
      } else if (iinit == 1) {
        id = 2;
        for (int i = 0; i < N * N; i++) rho0[i] = 1. / N;
      } else {
        id = 3;
        for (int i = 0; i < N; i++) rho0[vec_id(i, i, N)] = 1. / N;
      }
      break;
    case QD_INIT_NPLUSONE:
      if (iinit < N) rho0[vec_id(iinit, iinit, N)] = 1.0;
      else
        for (int i = 0; i < N * N; i++) rho0[i] = 1.0 / N;
      id = iinit;
      break;
    case QD_INIT_DIAGONAL: {
      int dim_post = 1;
      for (int k = o->init_ids.back() + 1; k < S.Q; k++) dim_post *= S.ness[k];
      int diag = iinit * dim_post;
      if (de < N) diag = map_ess_to_full(S, diag);
      rho0[S.lindblad ? vec_id(diag, diag, N) : diag] = 1.0;
      id = S.lindblad ? iinit * ninit + iinit : iinit;
      break;
    }
    case QD_INIT_BASIS: {
      int dim_post = 1;
      for (int k = o->init_ids.back() + 1; k < S.Q; k++) dim_post *= S.ness[k];
      const int sqn = (int)sqrt((double)ninit);
      int k = iinit % sqn, j = iinit / sqn;
      id = j * sqn + k;
      k *= dim_post;
      j *= dim_post;
      if (de < N) {
        k = map_ess_to_full(S, k);
        j = map_ess_to_full(S, j);
      }
      if (k == j) rho0[vec_id(k, k, N)] = 1.0;
      else if (k < j) {
        rho0[vec_id(k, k, N)] = rho0[vec_id(j, j, N)] = 0.5;
        rho0[vec_id(k, j, N)] = rho0[vec_id(j, k, N)] = 0.5;
      } else {
        rho0[vec_id(k, k, N)] = rho0[vec_id(j, j, N)] = 0.5;
        rho0[vec_id(k, j, N) + dim] = -0.5;
        rho0[vec_id(j, k, N) + dim] = 0.5;
      }
      break;
    }
  }
  return id;
}

static void fill_from_file(const qd_handle* h, const double* data, std::vector<double>& v) {
  const DevSys& S = h->S;
  const int dim = S.dim, N = S.N, de = h->dim_ess;
  v.assign((size_t)2 * dim, 0.0);
  if (S.lindblad) {
    for (int i = 0; i < de * de; i++) {
      int k = i % de, j = i / de;
      if (de * de < dim) {
        k = map_ess_to_full(S, k);
        j = map_ess_to_full(S, j);
      }
      const int el = vec_id(k, j, N);
      v[el] = data[i];
      v[el + dim] = data[i + de * de];
    }
  } else {
    for (int i = 0; i < de; i++) {
      const int k = de < dim ? map_ess_to_full(S, i) : i;
      v[k] = data[i];
      v[k + dim] = data[i + de];
    }
  }
}

extern "C" void qd_optim_destroy(qd_optim* o) {
  if (!o) return;
  (void)hipSetDevice(o->h->device);
  struct Quiet { ~Quiet() { (void)hipGetLastError(); } } quiet;  // teardown never leaves a sticky error behind
  for (DBuf* b : {&o->d_x0, &o->d_tgt, &o->d_pur, &o->d_rbib, &o->d_jbar, &o->d_xbar, &o->d_w, &o->d_red}) b->release();
  o->h_red.release();
  for (hipEvent_t e : o->evr)
    if (e) (void)hipEventDestroy(e);
  delete o;
}

extern "C" int qd_optim_create(qd_handle* h, const qd_objective* ob, int rank, int nranks, qd_optim** out) {
  if (!h || !ob || !out) return fail(QD_ERR_INVALID, "qd_optim_create: null argument");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(QD_ERR_INVALID, "qd_optim_create: bad rank / nranks");
  const DevSys& S = h->S;
  const int dim = S.dim, N = S.N;
  qd_optim* o = new qd_optim();
  o->h = h;
  o->rank = rank;
  o->nranks = nranks;
  o->initcond_type = ob->initcond_type;
  o->target_type = ob->target_type;
  o->objective_type = ob->objective_type;
  auto bail = [&](int code, const char* msg) {
    delete o;
    return fail(code, msg);
  };
  if (!S.lindblad) {  // src/optimtarget.cpp:55-65
    if (ob->initcond_type == QD_INIT_ENSEMBLE || ob->initcond_type == QD_INIT_THREESTATES || ob->initcond_type == QD_INIT_NPLUSONE)
      return bail(QD_ERR_INVALID, "qd_optim_create: this initial condition needs the Lindblad solver");
    if (ob->initcond_type == QD_INIT_BASIS) o->initcond_type = QD_INIT_DIAGONAL;
  }
  if (ob->n_init_ids < 0 || ob->n_init_ids > QD_MAX_OSC) return bail(QD_ERR_INVALID, "qd_optim_create: bad n_init_ids");
  for (int i = 0; i < ob->n_init_ids; i++) o->init_ids.push_back(ob->init_ids[i]);
  if (ob->initcond_type == QD_INIT_DIAGONAL || ob->initcond_type == QD_INIT_BASIS || ob->initcond_type == QD_INIT_ENSEMBLE) {
    // oscillator ids: in range and consecutive, as the reference asserts (src/main.cpp:100-104, src/optimtarget.cpp:138-141)
    for (size_t i = 0; i < o->init_ids.size(); i++) {
      if (o->init_ids[i] < 0 || o->init_ids[i] >= S.Q) return bail(QD_ERR_INVALID, "qd_optim_create: initial-condition oscillator id out of range");
      if (i > 0 && o->init_ids[i] != o->init_ids[i - 1] + 1) return bail(QD_ERR_INVALID, "qd_optim_create: initial-condition oscillator ids must be consecutive");
    }
  }
  // number of initial conditions (src/main.cpp:89-128)
  switch (ob->initcond_type) {
    case QD_INIT_FROMFILE: case QD_INIT_PURE: case QD_INIT_PERFORMANCE: case QD_INIT_ENSEMBLE: o->ninit = 1; break;
    case QD_INIT_THREESTATES: o->ninit = 3; break;
    case QD_INIT_NPLUSONE: o->ninit = N + 1; break;
    case QD_INIT_DIAGONAL: case QD_INIT_BASIS: {
      if (o->init_ids.empty()) return bail(QD_ERR_INVALID, "qd_optim_create: diagonal/basis need oscillator ids");
      int ni = 1;
      for (int v : o->init_ids) ni *= S.ness[v];
      if (ob->initcond_type == QD_INIT_BASIS && S.lindblad) ni *= ni;
      o->ninit = ni;
      break;
    }
    default: return bail(QD_ERR_INVALID, "qd_optim_create: unknown initial condition type");
  }
  if (o->ninit % nranks != 0)
    return bail(QD_ERR_INVALID, "qd_optim_create: number of ranks must divide the number of initial conditions (src/main.cpp:150-153)");
  o->nlocal = o->ninit / nranks;
  o->first = rank * o->nlocal;
  // fixed initial states (src/optimtarget.cpp:74-196)
  o->rho0_fixed.assign((size_t)2 * dim, 0.0);
  if (o->initcond_type == QD_INIT_PURE) {
    if ((int)o->init_ids.size() != S.Q) return bail(QD_ERR_INVALID, "qd_optim_create: pure initial state needs one level per oscillator");
    int diag = 0;
    for (int k = 0; k < S.Q; k++) {
      if (o->init_ids[k] > S.n[k] - 1 || o->init_ids[k] < 0) return bail(QD_ERR_INVALID, "qd_optim_create: pure initial state exceeds nlevels");
      diag += o->init_ids[k] * S.post[k];
    }
    o->rho0_fixed[S.lindblad ? vec_id(diag, diag, N) : diag] = 1.0;
  } else if (o->initcond_type == QD_INIT_FROMFILE) {
    if (!ob->init_data) return bail(QD_ERR_INVALID, "qd_optim_create: initial condition from file needs data");
    fill_from_file(h, ob->init_data, o->rho0_fixed);
  } else if (o->initcond_type == QD_INIT_ENSEMBLE) {
    if (o->init_ids.empty()) return bail(QD_ERR_INVALID, "qd_optim_create: ensemble needs oscillator ids");
    int dimpost = 1, dimsub = 1;
    for (int i = 0; i < S.Q; i++) {
      if (o->init_ids.front() <= i && i <= o->init_ids.back()) dimsub *= S.ness[i];
      else dimpost *= S.ness[i];
    }
    for (int i = 0; i < dimsub; i++)
      for (int j = i; j < dimsub; j++) {
        int ifull = i * dimpost, jfull = j * dimpost;
        if (h->dim_ess < N) {
          ifull = map_ess_to_full(S, ifull);
          jfull = map_ess_to_full(S, jfull);
        }
        if (i == j) o->rho0_fixed[vec_id(ifull, jfull, N)] = 1. / dimsub;
        else {
          const double v = 0.5 / (dimsub * dimsub);
          o->rho0_fixed[vec_id(ifull, jfull, N)] = v;
          o->rho0_fixed[vec_id(ifull, jfull, N) + dim] = v;
          o->rho0_fixed[vec_id(jfull, ifull, N)] = v;
          o->rho0_fixed[vec_id(jfull, ifull, N) + dim] = -v;
        }
      }
  }
  // target (src/optimtarget.cpp:199-306)
  if (ob->target_type == QD_TARGET_GATE) {
    if (!ob->gate_re) return bail(QD_ERR_INVALID, "qd_optim_create: gate target needs the gate matrix");
    build_gate(h, ob, o->V);
  } else if (ob->target_type == QD_TARGET_PURE) {
    o->purestate_id = 0;
    for (int k = 0; k < S.Q; k++) {
      if (ob->target_pure_levels[k] >= S.n[k] || ob->target_pure_levels[k] < 0)
        return bail(QD_ERR_INVALID, "qd_optim_create: pure target exceeds nlevels");
      o->purestate_id += ob->target_pure_levels[k] * S.post[k];
    }
  } else if (ob->target_type == QD_TARGET_FROMFILE) {
    if (!ob->target_data) return bail(QD_ERR_INVALID, "qd_optim_create: target from file needs data");
    fill_from_file(h, ob->target_data, o->target_fixed);
  } else {
    return bail(QD_ERR_INVALID, "qd_optim_create: unknown target type");
  }
  if (ob->objective_type == QD_OBJ_JMEASURE && ob->target_type != QD_TARGET_PURE)
    return bail(QD_ERR_INVALID, "qd_optim_create: Jmeasure needs a pure target (src/optimtarget.cpp:758-761)");
  // weights (src/optimproblem.cpp:72-91)
  o->weights.resize(o->ninit);
  double sum = 0.0;
  for (int i = 0; i < o->ninit; i++) {
    o->weights[i] = (ob->nweights > 0 && ob->weights) ? ob->weights[i < ob->nweights ? i : ob->nweights - 1] : 1.0;
    sum += o->weights[i];
  }
  for (int i = 0; i < o->ninit; i++) o->weights[i] /= sum;
  o->gamma_tik = ob->gamma_tik;
  o->gamma_var = ob->gamma_penalty_variation;
  o->pen = ob->penalty;
  if (o->pen.gamma_penalty_dpdm > 1e-13 && S.lindblad) o->pen.gamma_penalty_dpdm = 0.0;  // src/optimproblem.cpp:119-124
  if (ob->tik0) {
    if (!ob->alpha0 && h->ndesign > 0) return bail(QD_ERR_INVALID, "qd_optim_create: optim_regul_tik0 needs alpha0");
    o->alpha0.assign(ob->alpha0, ob->alpha0 + h->ndesign);
  }
  // build the local batch on the host once, keep it resident in HBM
  const size_t n2 = (size_t)2 * dim;
  std::vector<double> x0((size_t)o->nlocal * n2), tgt, pur(o->nlocal);
  const bool need_tgt = ob->target_type != QD_TARGET_PURE;
  if (need_tgt) tgt.resize((size_t)o->nlocal * n2);
  o->init_id.resize(o->nlocal);
  for (int i = 0; i < o->nlocal; i++) {
    double* xi = x0.data() + (size_t)i * n2;
    o->init_id[i] = prepare_initial_state(o, o->first + i, xi);
    double nn = 0.0;  // purity = ||rho0||_2^2 (src/optimtarget.cpp:705-707)
    for (size_t k = 0; k < n2; k++) nn += xi[k] * xi[k];
    const double nrm = sqrt(nn);
    pur[i] = nrm * nrm;
    if (ob->target_type == QD_TARGET_GATE) apply_gate(S, o->V, xi, tgt.data() + (size_t)i * n2);
    else if (ob->target_type == QD_TARGET_FROMFILE) std::copy(o->target_fixed.begin(), o->target_fixed.end(), tgt.data() + (size_t)i * n2);
  }
  int rc = QD_OK;
  auto dev = [&]() -> int {
    QD_HIP(qd::use_device(h->device));
    int r;
    if ((r = o->d_x0.ensure(x0.size())) || (r = o->d_pur.ensure(o->nlocal))) return r;
    QD_HIP(hipMemcpy(o->d_x0.p, x0.data(), sizeof(double) * x0.size(), hipMemcpyHostToDevice));
    QD_HIP(hipMemcpy(o->d_pur.p, pur.data(), sizeof(double) * o->nlocal, hipMemcpyHostToDevice));
    if (need_tgt) {
      if ((r = o->d_tgt.ensure(tgt.size()))) return r;
      QD_HIP(hipMemcpy(o->d_tgt.p, tgt.data(), sizeof(double) * tgt.size(), hipMemcpyHostToDevice));
    }
    if ((r = o->d_rbib.ensure((size_t)2 * o->nlocal)) || (r = o->d_jbar.ensure((size_t)3 * o->nlocal)) ||
        (r = o->d_xbar.ensure(x0.size())))
      return r;
    // multi-GPU path: local weights, reduction buffer [7 sums | gradient], event brackets of the collectives
    if ((r = o->d_w.ensure(o->nlocal)) || (r = o->d_red.ensure((size_t)QD_NSUMS + 1 + h->ndesign)) ||
        (r = o->h_red.ensure((size_t)QD_NSUMS + 1 + h->ndesign)))
      return r;
    QD_HIP(hipMemcpy(o->d_w.p, o->weights.data() + o->first, sizeof(double) * o->nlocal, hipMemcpyHostToDevice));
    {  // Jbar_penalty, Jbar_penalty_dpdm, Jbar_energy_penalty of solveAdjointODE per local initial condition (src/optimproblem.cpp:514-519)
      std::vector<double> jbar((size_t)3 * o->nlocal);
      o->ebar = 0.0;
      for (int i = 0; i < o->nlocal; i++) {
        const double w = o->weights[o->first + i];
        jbar[3 * i] = w * o->pen.gamma_penalty;
        jbar[3 * i + 1] = w * o->pen.gamma_penalty_dpdm;
        jbar[3 * i + 2] = w * o->pen.gamma_penalty_energy;
        if (o->pen.gamma_penalty_energy > 1e-13) o->ebar += jbar[3 * i + 2];
      }
      QD_HIP(hipMemcpy(o->d_jbar.p, jbar.data(), sizeof(double) * jbar.size(), hipMemcpyHostToDevice));
    }
    for (hipEvent_t& e : o->evr) QD_HIP(hipEventCreate(&e));
    return QD_OK;
  };
  rc = dev();
  if (rc) {
    qd_optim_destroy(o);
    return rc;
  }
  o->tg.target_type = ob->target_type;
  o->tg.objective_type = ob->objective_type;
  o->tg.purestate_id = o->purestate_id;
  o->tg.idm = S.lindblad ? o->purestate_id * (N + 1) : o->purestate_id;
  o->tg.tstates = need_tgt ? o->d_tgt.p : nullptr;
  o->tg.purity = o->d_pur.p;
  *out = o;
  return QD_OK;
}

extern "C" int qd_optim_ninit(const qd_optim* o) { return o ? o->ninit : QD_ERR_INVALID; }
extern "C" int qd_optim_ninit_local(const qd_optim* o) { return o ? o->nlocal : QD_ERR_INVALID; }

extern "C" int qd_optim_initial_state(qd_optim* o, int i, double* x0, int* initid) {
  if (!o || !x0 || i < 0 || i >= o->nlocal) return fail(QD_ERR_INVALID, "qd_optim_initial_state: bad argument");
  QD_HIP(qd::use_device(o->h->device));
  const size_t n2 = (size_t)2 * o->h->S.dim;
  QD_HIP(hipMemcpy(x0, o->d_x0.p + (size_t)i * n2, sizeof(double) * n2, hipMemcpyDeviceToHost));
  if (initid) *initid = o->init_id[i];
  return QD_OK;
}

extern "C" int qd_optim_target_state(qd_optim* o, int i, double* xt) {
  if (!o || !xt || i < 0 || i >= o->nlocal) return fail(QD_ERR_INVALID, "qd_optim_target_state: bad argument");
  QD_HIP(qd::use_device(o->h->device));
  const size_t n2 = (size_t)2 * o->h->S.dim;
  if (o->tg.tstates) {
    QD_HIP(hipMemcpy(xt, o->d_tgt.p + (size_t)i * n2, sizeof(double) * n2, hipMemcpyDeviceToHost));
  } else {
    std::fill(xt, xt + n2, 0.0);
    xt[o->tg.idm] = 1.0;
  }
  return QD_OK;
}

// finalizeJ / finalizeJ_diff (src/optimtarget.cpp:864-897)
static double finalize_J(const qd_optim* o, double re, double im) {
  if (o->objective_type == QD_OBJ_JTRACE) return o->h->S.lindblad ? 1.0 - re : 1.0 - (re * re + im * im);
  return re;
}
static void finalize_J_diff(const qd_optim* o, double re, double im, double* rb, double* ib) {
  if (o->objective_type == QD_OBJ_JTRACE) {
    if (o->h->S.lindblad) { *rb = -1.0; *ib = 0.0; } else { *rb = -2. * re; *ib = -2. * im; }
  } else {
    *rb = 1.0;
    *ib = 0.0;
  }
}

// BSpline0::computeVariation(_diff) (src/controlbasis.cpp:257-312); every other basis returns 0
static double control_variation(const qd_handle* h, const double* alpha, double* G, double var_bar) {
  double var = 0.0;
  const double fact = 2.0 * var_bar;
  for (int k = 0; k < h->S.Q; k++) {
    const DevOsc& o = h->oscs[k];
    if (o.nparams == 0) continue;
    const double* pr = alpha + o.offset;
    double* gr = G ? G + o.offset : nullptr;
    for (int b = 0; b < o.nseg; b++) {
      const DevSeg& g = h->segs[o.seg_begin + b];
      if (g.type != QD_CTRL_BSPLINE0) continue;
      const int ns = g.nsplines;
      for (int f = 0; f < o.ncar; f++) {
        for (int part = 0; part < 2; part++) {
          const int base = g.skip + (2 * f + part) * ns;
          for (int lc = 1; lc < ns; lc++) {
            const double d = pr[base + lc] - pr[base + lc - 1];
            var += d * d;
          }
          if (gr) {
            gr[base] += fact * (pr[base] - pr[base + 1]);
            for (int lc = 1; lc < ns - 1; lc++) gr[base + lc] += fact * (2 * pr[base + lc] - pr[base + lc - 1] - pr[base + lc + 1]);
            gr[base + ns - 1] += fact * (pr[base + ns - 1] - pr[base + ns - 2]);
          }
        }
        if (h->dctl.enforce_bc) {
          const int b0 = g.skip + 2 * f * ns;
          var += pr[b0] * pr[b0] + pr[b0 + ns - 1] * pr[b0 + ns - 1] + pr[b0 + ns] * pr[b0 + ns] + pr[b0 + 2 * ns - 1] * pr[b0 + 2 * ns - 1];
          if (gr) {
            gr[b0] += fact * pr[b0];
            gr[b0 + ns - 1] += fact * pr[b0 + ns - 1];
            gr[b0 + ns] += fact * pr[b0 + ns];
            gr[b0 + 2 * ns - 1] += fact * pr[b0 + 2 * ns - 1];
          }
        }
      }
    }
  }
  return var;
}

static bool trajectory_fits(qd_handle* h, int nb, const DevTarget* tg) {
  size_t need;
  h->traj_doubles(nb, &need);
  if (!h->stores_full(nb, tg)) need = 0;      // (a gradient evaluation whose adjoint sweep reads the stages only)
  const size_t needz = h->ztraj_doubles(nb);  // the stored primal stages travel with the trajectory
  need += needz;
  if (h->opts.traj_budget_mb > 0.0)  // test hook (option traj_budget_mb): pretend HBM is this small
    return (double)need * sizeof(double) <= h->opts.traj_budget_mb * 1048576.0;
  if (need - needz <= h->d_traj.cap && needz <= h->d_ztraj.cap) return true;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  const size_t avail = free_b + (h->d_traj.cap + h->d_ztraj.cap) * sizeof(double);
  return (double)need * sizeof(double) < 0.85 * (double)avail;
}

// forward sweeps inside this scope are followed by their adjoint sweep and by nothing else that reads the trajectory
struct StagesScope {
  qd_handle* h;
  bool saved;
  explicit StagesScope(qd_handle* hh) : h(hh), saved(hh->stages_only) { h->stages_only = true; }
  ~StagesScope() { h->stages_only = saved; }
};

struct PenaltyScope {
  qd_handle* h;
  qd_penalty saved;
  PenaltyScope(qd_handle* hh, const qd_penalty& p) : h(hh), saved(hh->pen) { h->pen = p; }
  ~PenaltyScope() { h->pen = saved; }
};

static DevTarget shifted_target(const qd_optim* o, int offset) {
  DevTarget t = o->tg;
  if (t.tstates) t.tstates += (size_t)offset * 2 * o->h->S.dim;
  t.purity += offset;
  return t;
}

// the seven sums of the last forward sweep, which propagated the local initial conditions [off, off + nc): src/optimproblem.cpp:258-279
static void add_partial_sums(const qd_optim* o, int off, int nc, double energy, double* partial) {
  const qd_handle* h = o->h;
  const double *pen = h->res_pen(), *dpdm = h->res_dpdm(), *o4 = h->res_out4();  // pinned, downloaded with the sweep
  for (int i = 0; i < nc; i++) {
    const double w = o->weights[o->first + off + i];
    partial[QD_SUM_PENALTY] += w * o->pen.gamma_penalty * (o->pen.gamma_penalty > 1e-13 ? pen[i] : 0.0);
    partial[QD_SUM_DPDM] += w * o->pen.gamma_penalty_dpdm * (o->pen.gamma_penalty_dpdm > 1e-13 ? dpdm[i] : 0.0);
    partial[QD_SUM_ENERGY] += w * o->pen.gamma_penalty_energy * (o->pen.gamma_penalty_energy > 1e-13 ? energy : 0.0);
    partial[QD_SUM_COST_RE] += w * o4[4 * i];
    partial[QD_SUM_COST_IM] += w * o4[4 * i + 1];
    partial[QD_SUM_FID_RE] += 1. / o->ninit * o4[4 * i + 2];
    partial[QD_SUM_FID_IM] += 1. / o->ninit * o4[4 * i + 3];
  }
}

// How a shard whose trajectory does not fit is cut: the fewest chunks that fit, of equal size (halving until it fits left 3600 initial
// conditions in four chunks of 900 where three of 1200 fit), rounded up to whole rounds of workgroups over the CUs where that still fits.
static int plan_chunk(qd_handle* h, int nl, const DevTarget* tg) {
  if (trajectory_fits(h, nl, tg)) return nl;
  int lo = 0, hi = nl;  // largest batch that fits: lo fits (0 = none), hi does not
  while (hi - lo > 1) {
    const int mid = lo + (hi - lo) / 2;
    if (trajectory_fits(h, mid, tg)) lo = mid;
    else hi = mid;
  }
  if (lo < 1) return 0;
  const int nchunks = (nl + lo - 1) / lo;
  int chunk = (nl + nchunks - 1) / nchunks;
  const int round = 256;  // CUs of the device: one workgroup per initial condition and CU on the large systems that get here
  const int up = (chunk + round - 1) / round * round;
  if (up <= lo) chunk = up;
  return chunk;
}

extern "C" int qd_optim_forward_local(qd_optim* o, const double* alpha, int store_trajectory, double* partial) {
  if (!o || !partial || (!alpha && o->h->ndesign > 0)) return fail(QD_ERR_INVALID, "qd_optim_forward_local: null argument");
  qd_handle* h = o->h;
  QD_HIP(qd::use_device(h->device));
  int r;
  if ((r = qd_set_params(h, alpha, h->ndesign))) return r;
  o->last_alpha.assign(alpha, alpha + h->ndesign);
  const int nl = o->nlocal;
  PenaltyScope ps0(h, o->pen);
  bool store = store_trajectory != 0 && trajectory_fits(h, nl, &o->tg);
  double energy = 0.0;
  {
    PenaltyScope ps(h, o->pen);  // the objective's penalty block, for this call only (operator-level calls keep theirs)
    if ((r = h->forward_dev(o->d_x0.p, nl, store, &o->tg, &energy))) return r;
  }
  o->stored = store;
  o->forward_done = true;
  for (int i = 0; i < QD_NSUMS; i++) partial[i] = 0.0;
  add_partial_sums(o, 0, nl, energy, partial);
  return QD_OK;
}

extern "C" int qd_optim_finalize(qd_optim* o, const double* alpha, const double* s, qd_objective_value* val) {
  if (!o || !s || !val || (!alpha && o->h->ndesign > 0)) return fail(QD_ERR_INVALID, "qd_optim_finalize: null argument");
  const qd_handle* h = o->h;
  // src/optimproblem.cpp:300-329
  val->fidelity = h->S.lindblad ? s[QD_SUM_FID_RE] : s[QD_SUM_FID_RE] * s[QD_SUM_FID_RE] + s[QD_SUM_FID_IM] * s[QD_SUM_FID_IM];
  val->cost = finalize_J(o, s[QD_SUM_COST_RE], s[QD_SUM_COST_IM]);
  double xn2 = 0.0;
  for (int i = 0; i < h->ndesign; i++) {
    const double d = alpha[i] - (o->alpha0.empty() ? 0.0 : o->alpha0[i]);
    xn2 += d * d;
  }
  const double xnorm = sqrt(xn2);
  val->regul = o->gamma_tik / 2. * xnorm * xnorm;
  val->penalty = s[QD_SUM_PENALTY];
  val->penalty_dpdm = s[QD_SUM_DPDM];
  val->penalty_energy = s[QD_SUM_ENERGY];
  val->penalty_variation = 0.5 * o->gamma_var * control_variation(h, alpha, nullptr, 0.0);
  val->objective = val->cost + val->regul + val->penalty + val->penalty_dpdm + val->penalty_energy + val->penalty_variation;
  return QD_OK;
}

extern "C" int qd_optim_adjoint_local(qd_optim* o, const double* alpha, const double* sums, double* grad) {
  if (!o || !sums || !grad || (!alpha && o->h->ndesign > 0)) return fail(QD_ERR_INVALID, "qd_optim_adjoint_local: null argument");
  qd_handle* h = o->h;
  QD_HIP(qd::use_device(h->device));
  if (!o->forward_done || (int)o->last_alpha.size() != h->ndesign || !std::equal(o->last_alpha.begin(), o->last_alpha.end(), alpha))
    return fail(QD_ERR_STATE, "qd_optim_adjoint_local: call qd_optim_forward_local with the same parameters first");
  const int nl = o->nlocal, nd = h->ndesign;
  const size_t n2 = (size_t)2 * h->S.dim;
  int r;
  PenaltyScope ps(h, o->pen);
  // adjoint seeds from the GLOBAL cost (src/optimproblem.cpp:433-436, :508-511)
  double rb, ib;
  finalize_J_diff(o, sums[QD_SUM_COST_RE], sums[QD_SUM_COST_IM], &rb, &ib);
  std::vector<double> rbib((size_t)2 * nl), jbar((size_t)3 * nl);
  double ebar = 0.0;
  for (int i = 0; i < nl; i++) {
    const double w = o->weights[o->first + i];
    rbib[2 * i] = w * rb;
    rbib[2 * i + 1] = w * ib;
    jbar[3 * i] = w * o->pen.gamma_penalty;
    jbar[3 * i + 1] = w * o->pen.gamma_penalty_dpdm;
    jbar[3 * i + 2] = w * o->pen.gamma_penalty_energy;
    if (o->pen.gamma_penalty_energy > 1e-13) ebar += jbar[3 * i + 2];
  }
  QD_HIP(hipMemcpyAsync(o->d_rbib.p, rbib.data(), sizeof(double) * rbib.size(), hipMemcpyHostToDevice, h->stream));
  QD_HIP(hipMemcpyAsync(o->d_jbar.p, jbar.data(), sizeof(double) * jbar.size(), hipMemcpyHostToDevice, h->stream));
  if (o->stored) {
    QD_HIP(launch_seed(h->S, o->tg, h->d_xT.p, o->d_rbib.p, nl, o->d_xbar.p, h->stream));
    if ((r = h->adjoint_dev(o->d_xbar.p, o->d_jbar.p, nl, &o->tg, false))) return r;
  } else {
    // The trajectory of the whole shard does not fit in HBM: redo the forward sweep chunk by chunk
    // with storage and reverse each chunk at once (the seeds only need the global sums).
    StagesScope ss(h);
    const int chunk = plan_chunk(h, nl, &o->tg);
    if (chunk < 1) return fail(QD_ERR_NOMEM, "qd_optim_adjoint_local: one trajectory does not fit in device memory");
    bool first = true;
    o->last_chunks = (nl + chunk - 1) / chunk;
    for (int off = 0; off < nl; off += chunk) {
      const int nc = std::min(chunk, nl - off);
      DevTarget t = shifted_target(o, off);
      h->accumulate_fwd_ms = true;  // last_fwd_ms = the first sweep + every chunk's re-propagation
      r = h->forward_dev(o->d_x0.p + (size_t)off * n2, nc, true, &t, nullptr);
      h->accumulate_fwd_ms = false;
      if (r) return r;
      QD_HIP(launch_seed(h->S, t, h->d_xT.p, o->d_rbib.p + (size_t)2 * off, nc, o->d_xbar.p, h->stream));
      if ((r = h->adjoint_dev(o->d_xbar.p, o->d_jbar.p + (size_t)3 * off, nc, &t, !first))) return r;
      first = false;
    }
  }
  if ((r = h->gradient_from_coeffs(ebar, grad))) return r;
  if (o->rank == 0) {  // regularisation terms on ONE rank only (src/optimproblem.cpp:356-372)
    for (int i = 0; i < nd; i++) grad[i] += o->gamma_tik * (alpha[i] - (o->alpha0.empty() ? 0.0 : o->alpha0[i]));
    control_variation(h, alpha, grad, 0.5 * o->gamma_var);
  }
  return QD_OK;
}

extern "C" int qd_optim_evalF(qd_optim* o, const double* alpha, qd_objective_value* val) {
  if (!o || !val) return fail(QD_ERR_INVALID, "qd_optim_evalF: null argument");
  if (o->nranks != 1) return fail(QD_ERR_STATE, "qd_optim_evalF: single-rank wrapper; use forward_local + all-reduce + finalize");
  double sums[QD_NSUMS];
  int r;
  if ((r = qd_optim_forward_local(o, alpha, 0, sums))) return r;
  return qd_optim_finalize(o, alpha, sums, val);
}

// Gradient of a shard whose trajectory does not fit in HBM, in ONE pass over the initial conditions: chunk by chunk a storing forward sweep,
// its partial sums, its seeds and its adjoint sweep.  Possible wherever the adjoint seeds do not depend on the reduced cost - every
// objective but Schroedinger + Jtrace (finalizeJ_diff is constant, src/optimtarget.cpp:889-895; the reference propagates every initial
// condition forward and backward in turn anyway, src/optimproblem.cpp:384-452).  The two-pass form (qd_optim_forward_local without storage,
// then the chunked re-propagation of qd_optim_adjoint_local) costs one more forward sweep of the whole shard: 0.77 s of 2.5 s on the 3x20
// workload at its full time grid.  sums[7] = this shard's partial sums, grad[ndesign] = this shard's gradient WITHOUT regularisation terms.
static int gradient_one_pass(qd_optim* o, const double* alpha, double* sums, double* grad) {
  qd_handle* h = o->h;
  const int nl = o->nlocal;
  const size_t n2 = (size_t)2 * h->S.dim;
  int r;
  if ((r = qd_set_params(h, alpha, h->ndesign))) return r;
  o->last_alpha.assign(alpha, alpha + h->ndesign);
  PenaltyScope ps(h, o->pen);
  StagesScope ss(h);
  int chunk = plan_chunk(h, nl, &o->tg);
  if (chunk < 1) return fail(QD_ERR_NOMEM, "qd_optim_evalGradF: one trajectory does not fit in device memory");
  double rb, ib;
  finalize_J_diff(o, 0.0, 0.0, &rb, &ib);  // (constant for the objectives that get here)
  std::vector<double> rbib((size_t)2 * nl), jbar((size_t)3 * nl);
  double ebar = 0.0;
  for (int i = 0; i < nl; i++) {
    const double w = o->weights[o->first + i];
    rbib[2 * i] = w * rb;
    rbib[2 * i + 1] = w * ib;
    jbar[3 * i] = w * o->pen.gamma_penalty;
    jbar[3 * i + 1] = w * o->pen.gamma_penalty_dpdm;
    jbar[3 * i + 2] = w * o->pen.gamma_penalty_energy;
    if (o->pen.gamma_penalty_energy > 1e-13) ebar += jbar[3 * i + 2];
  }
  QD_HIP(hipMemcpyAsync(o->d_rbib.p, rbib.data(), sizeof(double) * rbib.size(), hipMemcpyHostToDevice, h->stream));
  QD_HIP(hipMemcpyAsync(o->d_jbar.p, jbar.data(), sizeof(double) * jbar.size(), hipMemcpyHostToDevice, h->stream));
  QD_HIP(hipStreamSynchronize(h->stream));  // (rbib / jbar are locals)
  for (int i = 0; i < QD_NSUMS; i++) sums[i] = 0.0;
  double fwd_ms = 0.0, applies = 0.0;
  bool first = true;
  for (int off = 0; off < nl; off += chunk) {
    int nc = std::min(chunk, nl - off);
    DevTarget t = shifted_target(o, off);
    double energy = 0.0;
    // plan_chunk looked at the FREE memory; another process on the same device (ranks sharing a GPU) may have taken it since: an
    // allocation that fails for the first chunk halves the chunk instead of failing the evaluation
    while ((r = h->forward_dev(o->d_x0.p + (size_t)off * n2, nc, true, &t, &energy)) == QD_ERR_NOMEM && off == 0 && chunk > 1) {
      (void)hipGetLastError();
      chunk = (chunk + 1) / 2;
      nc = std::min(chunk, nl);
    }
    if (r) return r;
    add_partial_sums(o, off, nc, energy, sums);
    fwd_ms += h->last_fwd_ms;
    applies += h->last_mean_applies * nc;
    QD_HIP(launch_seed(h->S, t, h->d_xT.p, o->d_rbib.p + (size_t)2 * off, nc, o->d_xbar.p, h->stream));
    if ((r = h->adjoint_dev(o->d_xbar.p, o->d_jbar.p + (size_t)3 * off, nc, &t, !first))) return r;  // (accumulates last_adj_ms too)
    first = false;
  }
  h->last_fwd_ms = fwd_ms;  // the whole evaluation, as the one-sweep path reports it
  h->last_mean_applies = applies / nl;
  o->stored = false;
  o->forward_done = false;  // (no sweep of the WHOLE shard is pending: qd_optim_adjoint_local needs its own forward_local)
  o->last_chunks = (nl + chunk - 1) / chunk;
  return h->gradient_from_coeffs(ebar, grad);
}

static bool seeds_need_global_cost(const qd_optim* o) { return !o->h->S.lindblad && o->objective_type == QD_OBJ_JTRACE; }

extern "C" int qd_optim_last_chunks(const qd_optim* o) { return o ? o->last_chunks : QD_ERR_INVALID; }

extern "C" int qd_optim_evalGradF(qd_optim* o, const double* alpha, qd_objective_value* val, double* grad) {
  if (!o || !val || !grad) return fail(QD_ERR_INVALID, "qd_optim_evalGradF: null argument");
  if (o->nranks != 1) return fail(QD_ERR_STATE, "qd_optim_evalGradF: single-rank wrapper; use the *_local entry points");
  double sums[QD_NSUMS];
  int r;
  StagesScope ss(o->h);
  o->last_chunks = 1;
  if (!seeds_need_global_cost(o)) {
    bool fits;
    // (the parameters first: whether the adjoint sweep reads states or stages only - and with it the size of the stored trajectory - follows
    //  the solver path, whose gates look at the current control amplitudes)
    if ((r = qd_set_params(o->h, alpha, o->h->ndesign))) return r;
    {
      PenaltyScope ps(o->h, o->pen);
      fits = trajectory_fits(o->h, o->nlocal, &o->tg);
    }
    if (!fits) {
      QD_HIP(qd::use_device(o->h->device));
      if ((r = gradient_one_pass(o, alpha, sums, grad))) return r;
      if ((r = qd_optim_finalize(o, alpha, sums, val))) return r;
      const int nd = o->h->ndesign;
      for (int i = 0; i < nd; i++) grad[i] += o->gamma_tik * (alpha[i] - (o->alpha0.empty() ? 0.0 : o->alpha0[i]));
      control_variation(o->h, alpha, grad, 0.5 * o->gamma_var);
      return QD_OK;
    }
  }
  if ((r = qd_optim_forward_local(o, alpha, 1, sums))) return r;
  if ((r = qd_optim_finalize(o, alpha, sums, val))) return r;
  return qd_optim_adjoint_local(o, alpha, sums, grad);
}

// Both sweeps of this rank's shard in one call, NO collective: for a host that reduces by itself (the reference's MPI_Allreduce pair,
// src/optimproblem.cpp:454-460 and :527) and wants the library's one-pass handling of a shard whose stored stages exceed HBM, which the
// two-call form (forward_local, all-reduce, adjoint_local) cannot offer - it has to propagate the shard twice.  Possible wherever the
// adjoint seeds do not depend on the reduced cost (finalizeJ_diff constant, src/optimtarget.cpp:889-895): everything except Schroedinger +
// Jtrace.  partial[QD_NSUMS] and grad_local[ndesign] are this rank's sums; no regularisation term is added (the caller adds it once, after
// its reduction: gamma_tik (alpha - alpha0) and the control-variation term, src/optimproblem.cpp:356-372).
extern "C" int qd_optim_gradient_local(qd_optim* o, const double* alpha, double* partial, double* grad_local) {
  if (!o || !partial || !grad_local || (!alpha && o->h->ndesign > 0)) return fail(QD_ERR_INVALID, "qd_optim_gradient_local: null argument");
  if (seeds_need_global_cost(o))
    return fail(QD_ERR_STATE, "qd_optim_gradient_local: the adjoint seeds of Schroedinger + Jtrace need the REDUCED cost (src/optimproblem.cpp:495-511); "
                              "use qd_optim_forward_local, reduce, qd_optim_adjoint_local");
  int r;
  StagesScope ss(o->h);
  o->last_chunks = 1;
  if ((r = qd_set_params(o->h, alpha, o->h->ndesign))) return r;
  bool fits;
  {
    PenaltyScope ps(o->h, o->pen);
    fits = trajectory_fits(o->h, o->nlocal, &o->tg);
  }
  if (!fits) {
    QD_HIP(qd::use_device(o->h->device));
    return gradient_one_pass(o, alpha, partial, grad_local);
  }
  if ((r = qd_optim_forward_local(o, alpha, 1, partial))) return r;
  const int rank_keep = o->rank;
  o->rank = 1;  // (no regularisation here)
  r = qd_optim_adjoint_local(o, alpha, partial, grad_local);
  o->rank = rank_keep;
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// multi-GPU evalF / evalGradF: one process per GPU, this rank's shard, RCCL reductions on the handle's stream.
// Everything between the forward sweep and the last collective stays in HBM: the partial sums are formed by
// k_partial_sums, reduced in place by ncclAllReduce, turned into seed weights by k_seed_weights; the gradient is
// reduced in the buffer k_grad wrote it to.  The reference's two collectives (src/optimproblem.cpp:454-460 before the
// adjoint seeds, :527 after the sweep) are needed as two only for Schroedinger + Jtrace (:495-511); everywhere else
// the seed does not depend on the reduced cost (finalizeJ_diff is constant, src/optimtarget.cpp:889-895) and sums and
// gradient travel in ONE all-reduce of 7 + ndesign doubles.  One host synchronisation per evaluation.
// ---------------------------------------------------------------------------------------------------------------
static int dist_finish(qd_optim* o, const double* alpha, bool grad_mode, qd_objective_value* val, double* grad, double* allreduce_ms) {
  qd_handle* h = o->h;
  const int nd = h->ndesign;
  QD_HIP(hipMemcpyAsync(o->h_red.p, o->d_red.p, sizeof(double) * (QD_NSUMS + (grad_mode ? nd : 0)), hipMemcpyDeviceToHost, h->stream));
  int r;
  if ((r = h->forward_finish(nullptr))) return r;  // the one synchronisation
  if (grad_mode && (r = h->adjoint_finish(false))) return r;
  if (allreduce_ms) {
    float a = 0.f, b = 0.f;
    QD_HIP(hipEventElapsedTime(&a, o->evr[0], o->evr[1]));
    if (grad_mode) QD_HIP(hipEventElapsedTime(&b, o->evr[2], o->evr[3]));
    allreduce_ms[0] = a;
    allreduce_ms[1] = b;
  }
  if ((r = qd_optim_finalize(o, alpha, o->h_red.p, val))) return r;
  if (grad_mode) {
    for (int i = 0; i < nd; i++) grad[i] = o->h_red.p[QD_NSUMS + i];
    // Tikhonov / variation terms: the reference adds them on rank 0 before the reduction (src/optimproblem.cpp:356-372);
    // adding them on every rank after it gives every rank the same complete gradient
    for (int i = 0; i < nd; i++) grad[i] += o->gamma_tik * (alpha[i] - (o->alpha0.empty() ? 0.0 : o->alpha0[i]));
    control_variation(h, alpha, grad, 0.5 * o->gamma_var);
  }
  return QD_OK;
}

static int dist_forward(qd_optim* o, qd_comm* c, const double* alpha, bool store) {
  qd_handle* h = o->h;
  int r;
  if ((r = qd_set_params(h, alpha, h->ndesign))) return r;
  o->last_alpha.assign(alpha, alpha + h->ndesign);
  if ((r = h->forward_launch(o->d_x0.p, o->nlocal, store, &o->tg))) return r;
  QD_HIP(launch_partial_sums(h->d_res.p, o->nlocal, o->d_w.p, 1.0 / o->ninit, o->pen, h->d_etable.p, h->cs, h->S.Q, h->tg.ntime,
                             o->d_red.p, h->stream));
  o->stored = store;
  o->forward_done = true;
  (void)c;
  return QD_OK;
}

extern "C" int qd_optim_evalF_dist(qd_optim* o, qd_comm* c, const double* alpha, qd_objective_value* val, double* allreduce_ms) {
  if (!o || !c || !val || (!alpha && o->h->ndesign > 0)) return fail(QD_ERR_INVALID, "qd_optim_evalF_dist: null argument");
  if (c->nranks != o->nranks || c->rank != o->rank) return fail(QD_ERR_INVALID, "qd_optim_evalF_dist: communicator and objective disagree on rank / nranks");
  qd_handle* h = o->h;
  QD_HIP(qd::use_device(h->device));
  PenaltyScope ps(h, o->pen);
  int r;
  if ((r = dist_forward(o, c, alpha, false))) return r;
  QD_HIP(hipEventRecord(o->evr[0], h->stream));
  if ((r = qd_comm_allreduce_dev(c, o->d_red.p, QD_NSUMS, 0, h->stream))) return r;
  QD_HIP(hipEventRecord(o->evr[1], h->stream));
  return dist_finish(o, alpha, false, val, nullptr, allreduce_ms);
}

extern "C" int qd_optim_evalGradF_dist(qd_optim* o, qd_comm* c, const double* alpha, qd_objective_value* val, double* grad, double* allreduce_ms) {
  if (!o || !c || !val || !grad || (!alpha && o->h->ndesign > 0)) return fail(QD_ERR_INVALID, "qd_optim_evalGradF_dist: null argument");
  if (c->nranks != o->nranks || c->rank != o->rank) return fail(QD_ERR_INVALID, "qd_optim_evalGradF_dist: communicator and objective disagree on rank / nranks");
  qd_handle* h = o->h;
  QD_HIP(qd::use_device(h->device));
  PenaltyScope ps(h, o->pen);
  StagesScope ss(h);
  const int nl = o->nlocal, nd = h->ndesign;
  int r;
  // Fused device path or host-staged fallback: the two issue DIFFERENT collectives, and trajectory_fits() looks at this rank's own free
  // memory - so the choice is made collectively (any rank that does not fit sends every rank down the fallback), once per objective.
  if (o->dist_fits < 0) {
    if ((r = qd_set_params(h, alpha, h->ndesign))) return r;  // (the solver gates behind trajectory_fits look at the current controls)
    double nofit = trajectory_fits(h, nl, &o->tg) ? 0.0 : 1.0;
    if (c->nranks > 1 && (r = qd_comm_allreduce(c, &nofit, 1, 1))) return r;
    o->dist_fits = nofit > 0.5 ? 0 : 1;
  }
  if (!o->dist_fits) {
    // the shard's trajectory exceeds HBM: host-staged path (chunked re-propagation), collectives through the same communicator
    double sums[QD_NSUMS];
    if (!seeds_need_global_cost(o)) {
      // one pass over the shard (the seeds are constants), then both reductions
      if ((r = gradient_one_pass(o, alpha, sums, grad))) return r;
      if ((r = qd_comm_allreduce(c, sums, QD_NSUMS, 0))) return r;
      if ((r = qd_optim_finalize(o, alpha, sums, val))) return r;
      if ((r = qd_comm_allreduce(c, grad, nd, 0))) return r;
      for (int i = 0; i < nd; i++) grad[i] += o->gamma_tik * (alpha[i] - (o->alpha0.empty() ? 0.0 : o->alpha0[i]));
      control_variation(h, alpha, grad, 0.5 * o->gamma_var);
      if (allreduce_ms) allreduce_ms[0] = allreduce_ms[1] = 0.0;
      return QD_OK;
    }
    if ((r = qd_optim_forward_local(o, alpha, 0, sums))) return r;
    if ((r = qd_comm_allreduce(c, sums, QD_NSUMS, 0))) return r;
    if ((r = qd_optim_finalize(o, alpha, sums, val))) return r;
    const int rank_keep = o->rank;
    o->rank = 1;  // regularisation is added once, after the reduction
    r = qd_optim_adjoint_local(o, alpha, sums, grad);
    o->rank = rank_keep;
    if (r) return r;
    if ((r = qd_comm_allreduce(c, grad, nd, 0))) return r;
    for (int i = 0; i < nd; i++) grad[i] += o->gamma_tik * (alpha[i] - (o->alpha0.empty() ? 0.0 : o->alpha0[i]));
    control_variation(h, alpha, grad, 0.5 * o->gamma_var);
    if (allreduce_ms) allreduce_ms[0] = allreduce_ms[1] = 0.0;
    return QD_OK;
  }
  if ((r = dist_forward(o, c, alpha, true))) return r;
  const bool two = seeds_need_global_cost(o);  // seeds need the GLOBAL cost
  QD_HIP(hipEventRecord(o->evr[0], h->stream));
  if (two && (r = qd_comm_allreduce_dev(c, o->d_red.p, QD_NSUMS, 0, h->stream))) return r;
  QD_HIP(hipEventRecord(o->evr[1], h->stream));
  QD_HIP(launch_seed_weights(o->d_red.p, o->d_w.p, nl, o->objective_type, h->S.lindblad, o->d_rbib.p, h->stream));
  // (d_jbar = beta_i x penalty coefficients: constants of the objective, resident since qd_optim_create)
  QD_HIP(launch_seed(h->S, o->tg, h->d_xT.p, o->d_rbib.p, nl, o->d_xbar.p, h->stream));
  if ((r = h->adjoint_launch(o->d_xbar.p, o->d_jbar.p, nl, &o->tg, false))) return r;
  if ((r = h->gradient_launch(o->ebar, o->d_red.p + QD_NSUMS))) return r;
  QD_HIP(hipEventRecord(o->evr[2], h->stream));
  if (two) r = qd_comm_allreduce_dev(c, o->d_red.p + QD_NSUMS, nd, 0, h->stream);
  else r = qd_comm_allreduce_dev(c, o->d_red.p, (size_t)QD_NSUMS + nd, 0, h->stream);
  if (r) return r;
  QD_HIP(hipEventRecord(o->evr[3], h->stream));
  return dist_finish(o, alpha, true, val, grad, allreduce_ms);
}
