// Host-side state behind the opaque qd_handle (one per GPU, single-threaded like the reference's
// TimeStepper/MasterEq objects).  Internal; the public boundary is include/quandary_amd.h.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "qd_internal.h"

#define QD_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      qd::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                                \
      return (_e == hipErrorOutOfMemory) ? QD_ERR_NOMEM : QD_ERR_DEVICE;                               \
    }                                                                                                  \
  } while (0)

namespace qd {

// Entry points select the handle's device and drop whatever error an earlier, unrelated HIP call of this thread left behind
// (the launch wrappers report hipGetLastError(), which is sticky per thread).
inline hipError_t use_device(int device) {
  const hipError_t e = hipSetDevice(device);
  (void)hipGetLastError();
  return e;
}

hipError_t launch_observables(const DevSys& S, const double* traj, int f32, int nb, int nstages, int stride, int nout, int nlev_total,
                              double* expected, double* population, double* expcomp, double* popcomp, hipStream_t st);

// growable device buffer of doubles
struct DBuf {
  double* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return QD_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    QD_HIP(hipMalloc(reinterpret_cast<void**>(&p), sizeof(double) * (n > 0 ? n : 1)));
    cap = n;
    return QD_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// growable PINNED host buffer: the target of the asynchronous result downloads (one stream
// synchronisation per sweep instead of one blocking copy per result array)
struct HBuf {
  double* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return QD_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    QD_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), sizeof(double) * (n > 0 ? n : 1), hipHostMallocDefault));
    cap = n;
    return QD_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace qd

struct qd_handle {
  int device = 0;
  int precision = QD_PRECISION_F64;  // qd_set_precision
  qd::TuneOpts opts;                 // qd_set_option (+ environment overrides read at qd_create)
  bool use_col(const qd::LaunchCfg& cfg) const;  // the sweep runs on the lean column kernels (qd_col.hip)
  int neumann_split_on() const;      // diagonal-split Neumann iteration for the current parameters
  // linearsolver_type = gmres served by the diagonal-split iteration of the lean column kernels under GMRES's stopping rule;
  // *kappa2 = (1 + max alpha |D|)^2, the factor between the squared update norm and the bound of the squared residual
  bool gmres_as_split(const qd::LaunchCfg& cfg, double* kappa2) const;
  bool gmres_as_neumann(const qd::LaunchCfg& cfg) const;  // ... by the plain Neumann iteration of any other kernel family
  // the decision of the two gates, latched per handle: -1 undecided, 0 Krylov kernels, 1 diagonal-split iteration, 2 Neumann iteration
  mutable int sub_latch = -1;
  bool params_set = false;  // qd_set_params has been called (the gates look at the control amplitudes)
  bool latched_substitution(int kind, double bound) const;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;  // forward / adjoint kernel brackets
  qd::DevSys S{};
  qd_time tg{};
  qd_solver sol{};
  qd_penalty pen{};
  // controls (host description + device mirror)
  std::vector<qd::DevSeg> segs;
  std::vector<qd::DevOsc> oscs;
  std::vector<double> carriers, pulses;
  qd::DevCtlDesc dctl{};
  qd::DevSeg* d_segs = nullptr;
  qd::DevOsc* d_oscs = nullptr;
  double *d_carriers = nullptr, *d_pulses = nullptr;
  int ndesign = 0, dim_ess = 1;
  int last_team = 1;  // workgroups per initial condition of the last sweep (qd_big.h)
  int last_solver = 0;  // QD_SOLVER_* of the last sweep (qd_last_solver)
  bool has_pipulse = false;
  bool has_ampbasis = false;  // a spline_amplitude segment: forward only (src/oscillator.cpp:350-356)
  std::vector<double> params;
  qd::DBuf d_params;
  qd::DBuf d_tbar, d_tred;  // team barrier counters / partial sums of the global-memory sweeps (qd_big.h)
  bool params_dirty = true;
  bool napply_zeroed = false;  // the control kernel of this parameter update has already reset d_napply
  // step schedule
  int nstages = 1, nsub = 0, cs = 0;
  std::vector<double> sched_t, sched_h, etimes;  // host copies
  qd::DBuf d_sched_t, d_sched_h, d_etimes, d_ezero, d_table, d_etable, d_onerow, d_onetime;
  qd::HBuf h_etable;  // energy-penalty table rows, downloaded with the sweep results
  qd::HBuf h_res;     // per-state results of the last forward sweep: [pen nb | dpdm nb | out4 4nb | napply]
  qd::HBuf h_params;  // staging copy of the control parameters
  // target for in-loop / final objective terms
  bool target_set = false;
  qd::DevTarget dtg{};
  qd::DBuf d_tstates, d_purity;
  int target_nb = 0;
  // sweep buffers
  qd::DBuf d_ztraj;  // stored primal stages (SweepArgs::ztraj)
  qd::DBuf d_wjw;    // Gaussian weights of the weighted-J penalty per time step (SweepArgs::wjw)
  double wjw_param = -1.0;
  int ensure_wj_weights();
  qd::DBuf d_x0, d_xT, d_traj, d_res, d_xbar, d_jbar, d_coeff, d_coeffsum, d_grad, d_y, d_stash, d_kry;
  qd::DBuf d_ecoef, d_edig, d_work;  // large states (qd_big.h): element table, work vectors
  qd::DBuf d_sched;                  // scheduler words of the time-sliced lean column sweeps (forward | adjoint)
  qd::HBuf h_sched;                  // their error words, downloaded with the sweep
  bool sliced_fwd = false, sliced_adj = false;
  int arm_slices(qd::SweepArgs& a, int nb, int which);
  int ensure_big(int nb);            // no-op unless the launch configuration is the large-state variant
  qd::DBuf d_g0, d_hcr, d_hci, d_gtab, d_gone;  // dense user-Hamiltonian path (qd_set_hamiltonian)
  // infinity norms of the uploaded Hamiltonians (row_bounds: the standard-model constants say nothing about a user Hamiltonian)
  double dense_hsys_norm = 0.0;
  std::vector<double> dense_hc_norm;  // per oscillator: ||Re Hc_k||_inf + ||Im Hc_k||_inf
  // d_res = [pen nb | dpdm nb | out4 4nb | napply]: one contiguous block, one download per sweep
  double *d_pen = nullptr, *d_dpdm = nullptr, *d_out4 = nullptr;
  unsigned long long* d_napply = nullptr;
  int last_nb = 0;
  bool traj_valid = false;
  int ztraj_fmt = 0;  // layout of the stored primal stages: 1 = fp32 pairs (fp32-mixed), 0 / 2 / 3 = fp64 pairs written by the general / 2^5 / lean column kernels (element order of the family)
  double last_mean_applies = 0.0, last_fwd_ms = 0.0, last_adj_ms = 0.0;
  bool accumulate_fwd_ms = false;  // chunked re-propagation (qd_optim_adjoint_local): add the chunks' forward times up

  // ---- internal device-pointer API used by the objective level (qd_optim.cpp) -------------------
  int refresh_tables();
  mutable double hmax_cache = -1.0;  // max |h(I)| over the level combinations (system constant, computed on first use)
  double control_amplitude_bound(int k) const;       // max_t |p_k(t)|, |q_k(t)| for the current parameters
  void row_bounds(double* diag, double* off) const;  // Gershgorin bounds of a row of M over all sub-steps (current parameters)
  int gmres_poly_degree() const;  // > 1 where the Neumann series provably contracts for the current parameters, else 1
  // degree of the polynomial preconditioner, tuned from sweep to sweep (forward_finish): smallest degree with one Krylov vector per solve
  int poly_cur = 6, poly_lo = 1, poly_hi = 0, last_poly = 1, last_var = 0, poly_steps = 0;
  int fwd_poly = 0;  // degree the last forward sweep ran on (the adjoint sweep of the same evaluation keeps it)
  int poly_start() const { return S.dim <= 1024 ? 3 : 6; }  // first degree the tuner tries (small systems contract fast)
  int poly_slow = 0;         // consecutive sweeps of a frozen degree with more than 1.5 Krylov vectors per solve
  bool poly_frozen = false;  // the bracket has closed: the degree no longer changes (reproducible evaluations)
  int traj_doubles(int nb, size_t* n) const;
  size_t ztraj_doubles(int nb) const;  // 0 for explicit Euler
  // forward sweep on device-resident states; results stay on the device (d_pen, d_dpdm, d_xT, d_out4)
  int forward_dev(const double* dx0, int nb, bool store, const qd::DevTarget* tg, double* energy);
  // the same in two halves: enqueue only / synchronise and collect (lets the caller queue the reductions, the adjoint
  // sweep and the collectives behind the forward sweep without a host round trip)
  int forward_launch(const double* dx0, int nb, bool store, const qd::DevTarget* tg);
  int forward_finish(double* energy);
  int adjoint_launch(const double* dxbarT, const double* djbar, int nb, const qd::DevTarget* tg, bool accumulate);
  int adjoint_finish(bool accumulate);
  int gradient_launch(double ebar, double* dgrad);  // k_grad into a device buffer [ndesign]
  bool pending_store = false;
  // Gradient evaluations of the objective level (qd_optim.cpp) set stages_only: the forward sweep then stores the primal stages z only
  // (SweepArgs::ztraj) wherever the adjoint sweep of the same kernel family reads nothing else - half the store traffic and half the
  // trajectory memory.  traj_full: d_traj holds the states x_n of the last stored sweep (qd_get_state, qd_get_observables, penalties
  // with state-dependent adjoints, explicit Euler).
  bool stages_only = false, traj_full = false, pending_full = false;
  bool adjoint_reads_states(int nb, const qd::DevTarget* tg) const;
  bool stores_full(int nb, const qd::DevTarget* tg) const { return !stages_only || adjoint_reads_states(nb, tg); }
  // adjoint sweep; dxbarT/djbar device pointers; coefficient sums accumulate into d_coeffsum
  int adjoint_dev(const double* dxbarT, const double* djbar, int nb, const qd::DevTarget* tg, bool accumulate);
  // gradient from d_coeffsum (+ energy term ebar); writes host grad[ndesign]
  int gradient_from_coeffs(double ebar, double* grad);
  double energy_penalty_host() const;
  const double* res_pen() const { return h_res.p; }
  const double* res_dpdm() const { return h_res.p + last_nb; }
  const double* res_out4() const { return h_res.p + 2 * (size_t)last_nb; }
};

// Communicator (qd_comm.cpp): one per process.  Backend RCCL (one process per GPU, ncclAllReduce over xGMI on the handle's stream) or
// HOST (ranks on one node reducing through a POSIX shared-memory segment: the same call sites, the same buffers, the same fixed summation
// order on every rank - for ranks that share a GPU, which RCCL refuses, and for nodes without RCCL).
namespace qd { struct HostRing; }
struct qd_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
  int backend = 0;                // 0 RCCL, 1 host shared memory
  qd::HostRing* host = nullptr;   // backend 1
  hipStream_t stream = nullptr;  // for the host-staged convenience calls; the sweeps reduce on the handle's stream
  qd::DBuf dbuf;
  qd::HBuf hbuf;                 // backend 1: pinned staging buffer of the in-stream reductions
};
// in-place all-reduce of a device buffer on stream `st` (op 0 = sum, 1 = max); asynchronous
int qd_comm_allreduce_dev(qd_comm* c, double* dbuf, size_t n, int op, hipStream_t st);
