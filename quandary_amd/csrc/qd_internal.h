// Internal declarations shared by the kernel translation unit (qd_kernels.hip) and the
// host side of the C ABI (qd_handle.cpp, qd_optim.cpp).  Not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "quandary_amd.h"

namespace qd {

// ---------------------------------------------------------------------------------------------
// Device-side description of the operator: the MatShellCtx parameter block of the reference
// (include/mastereq.hpp:20-42) with every constant already in rad/ns.  Passed BY VALUE as a
// kernel argument so that all per-oscillator constants are wave-uniform scalar loads.
// ---------------------------------------------------------------------------------------------
struct DevSys {
  int Q, lindblad, N, dim, npairs, maxn;
  int hasJ, pad0;  // any |J_kl| > 1e-10 (smaller couplings and |gamma_1| <= 1e-12 are zeroed on the host)
  int n[QD_MAX_OSC], ness[QD_MAX_OSC], post[QD_MAX_OSC];
  double detune[QD_MAX_OSC], xi[QD_MAX_OSC], g1[QD_MAX_OSC], g2[QD_MAX_OSC];
  double g1off[QD_MAX_OSC];  // gamma_1 of the off-diagonal decay term: 0 unless |gamma_1| > 1e-12 (mastereq.hpp:759)
  double xikl[QD_MAX_PAIRS], J[QD_MAX_PAIRS];
  // user-supplied dense Hamiltonians (qd_set_hamiltonian); the standard Hamiltonian model is then unused
  int dense, pad1;     // 0: matrix-free stencil; 1: dense operator, G(t) read from the table; 2: ... staged in LDS per sub-step
  const double* hcr;   // [Q][N*N] Re(Hc_k), row-major
  const double* hci;   // [Q][N*N] Im(Hc_k)
  const double* gtab;  // [rows][N*N] interleaved complex: G(t_row) = -i H(t_row), one row per control-table row
  // states beyond one CU's LDS (dim > 4096, qd_big.h): per-element invariants and the per-state work vectors in global memory
  const double* ecoef;   // [dim] (Delta, d)
  const unsigned* edig;  // [dim] (packed bra digits, packed ket digits)
  double* work;          // [nb][BIG_NV][dim] interleaved complex
  // teams of workgroups on one large state (qd_big.h): workgroups per initial condition, members dealt over all XCDs (1) or kept
  // on one (0), barrier counters [nb][BIG_BAR_STRIDE], partial sums [nb][2][BIG_TEAM_MAX][BIG_RED_NV]
  int team, team_spread;
  unsigned long long* tbar;
  double* tred;
};

// Control parameterisation on the device (src/oscillator.cpp:45-132, src/controlbasis.cpp:20-32,219-225)
struct DevSeg {
  int type, nsplines, skip, npc;  // npc = parameters per carrier wave
  double tstart, tstop, dtknot, width;
  double a1, a2, a3;              // step: amp1, amp2, tramp; spline_amplitude: scaling
};
struct DevOsc {
  int seg_begin, nseg, car_begin, ncar, offset, nparams, pulse_begin, npulse;
};
struct DevCtlDesc {
  int Q, enforce_bc, npairs, pad;
  const DevSeg* segs;
  const DevOsc* oscs;
  const double* carriers;  // rad/ns
  const double* pulses;    // [npulse_total][3] tstart, tstop, amp
  double eta[QD_MAX_PAIRS];
  double Tfinal;
};

// One row of the step-control table per sub-step: [h, t_eval, p_0..p_{Q-1}, q_0..q_{Q-1},
// cos_0.., sin_0..].  t_eval is the midpoint (IMR family) or tstart (EE).
inline int ctl_stride(int Q, int npairs) { return 2 + 2 * Q + 2 * npairs; }

struct DevTarget {
  int target_type, objective_type, purestate_id, idm;  // idm = vectorised index of the pure target
  const double* tstates;                               // [nb][2*dim] or nullptr
  const double* purity;                                // [nb]
};

struct SweepArgs {
  DevSys S;
  DevTarget tg;
  const double* ctl;  // step-control table [nsub][cs]
  int cs, nsub, nstages, ntime, nb;
  double dt, Tfinal;
  int stepper_ee, linsolve, maxiter;
  int gmres_poly;  // degree of the Neumann-polynomial right preconditioner of the global-memory GMRES (1 = none)
  int neumann_split;  // lean column kernels (qd_col.hip): diagonal of M on the left-hand side of the Neumann iteration
  // ... run in place of GMRES under GMRES's stopping rule: stop when kappa ||y_{m+1} - y_m|| <= max(rtol ||b||, abstol), where
  // kappa >= |1 - alpha D_ii| bounds the true residual ||b - (I - alpha M) y_m|| = ||(1 - alpha D)(y_{m+1} - y_m)|| from above
  int stop_residual;
  double kappa2;
  // A stationary iteration that SERVES A GMRES REQUEST (qd_handle::gmres_as_split / gmres_as_neumann) stops on the update norm like
  // the reference's Neumann solver, i.e. with an error of ~rho x abstol, rho = its contraction.  KSPGMRES typically ends far below its
  // tolerance (the last Krylov vector takes the residual down by orders of magnitude): on the reference's xgate_sparsemat case its
  // gradient sits 8.6e-9 (of the gradient norm) from the exact one, the plain update-norm rule - the reference's own Neumann solver - 1.2e-7;
  // with tau = 1e-3 (the default) the stand-in lands at 3e-9, one application more per step (profiles/xgate_probe.py).  With standin_tau2 > 0 the stand-ins additionally require the
  // error ESTIMATE below tau x tolerance: rho^2 d <= tau^2 thr with rho^2 = d / d_previous (squared update norms) - no extra iteration
  // where the contraction is strong, one more where it is not.  0 for neumann requests (the reference's rule, iteration for iteration).
  float standin_tau2;
  // [r6] The one-vector path of the Krylov solvers (ColTeam::kry_*, Team32::kry1) accepts its solution at residual <= kry_tau x max(rtol
  // ||b||, abstol): the degree of its polynomial is tuned to JUST reach the acceptance threshold, while KSPGMRES typically ends well below
  // its tolerance (its last Krylov vector takes the residual down by the contraction of a whole vector) - with kry_tau = 1 a 30-step
  // gradient of norm 1e-4 sat 4e-11 from the exact one where the oracle's GMRES sits at 5e-12 (profiles/r6_kry_seed_sweep.txt); 0.1 (the
  // default, option krylov_tau) costs at most one application more per solve.  The generic paths behind it keep the reference's rule.
  double kry_tau2;
  int kry_restart;  // restart length of the generic path of the lean column kernels' Krylov solver (<= KRY_MR = 14; option krylov_restart)
  // Gaussian weights of the weighted-J penalty, one per time step: wjw[n] = exp(-((n + 1) dt - T)^2 / param^2) / param
  // (timestepper.cpp:262-270, :304-315); tabulated once per handle so that the sweep kernels of qd_col.hip need no exp() per step
  const double* wjw;
  int use_gmres;  // 0: Neumann; 1: in-kernel GMRES, Krylov basis in LDS (one element per thread, small dim); 2: basis in global memory (kry)
  double abstol, reltol;
  double inv_abs2;  // 1 / abstol^2 and reltol^2 as the solvers use them: kernel arguments are re-read from the scalar cache, values
  float rel2;       // derived inside the kernel were hoisted out of the time loop and spilt to scratch (one round trip per solve)
  // penalties (src/timestepper.cpp:256-480)
  double gamma_penalty, penalty_param, gamma_dpdm;
  int leak_on;
  // forward
  const double* x0;  // [nb][2*dim]
  double* xT;        // [nb][2*dim]
  double* traj;      // [(nsub+1)][nb][2*dim] or nullptr
  // primal stages z_s = x_s + h/2 k_s of the implicit-midpoint family, [nsub][nb][2*dim], stored with the trajectory: the adjoint
  // sweep reads them instead of repeating the forward solve of every sub-step (ImplMidpoint::evolveBWD, timestepper.cpp:640-652)
  double* ztraj;
  double* pen_out;   // [nb]
  double* dpdm_out;  // [nb]
  unsigned long long* napply;
  // adjoint
  const double* xbarT;  // [nb][2*dim]
  const double* jbar;   // [nb][3]
  double* coeff;        // [nb][nsub][2Q]  (x^T dM/dp_k z, x^T dM/dq_k z)
  double* xbar0;        // [nb][2*dim] adjoint at t=0 (diagnostic) or nullptr
  double* kry;          // [nb][GMRES_MR_G+1][2*dim] Krylov basis of the GMRES variant whose basis does not fit in LDS
  // time-sliced scheduling of the lean column kernels (qd_col.hip): nslice slices of whole time steps, tasks drawn from sched[0]
  // (sched = nullptr: one workgroup per initial condition); the adjoint state is carried from slice to slice in `stash`
  int nslice;
  unsigned* sched;
  int col_noskip;  // lean column kernels: test the stopping rule in every pass (option col_skip = 0)
  unsigned long long sched_ticks;  // wall_clock64 ticks (100 MHz) a slice may wait for its predecessor before the error word is raised
  double* stash;        // [2][nb][2*dim] staging area of the several-elements-per-thread variants (adjoint state / midpoint state
                        // parked in L2/HBM while a linear solve runs, instead of compiler-chosen scratch spills)
};

// teams of workgroups in the global-memory sweeps (qd_big.h)
constexpr int BIG_TEAM_MAX = 256;   // workgroups per initial condition
constexpr int BIG_RED_NV = 16;      // doubles per member in the team reduction buffer
constexpr int BIG_BAR_STRIDE = 16 * 9;  // per team: the team's counter and eight first-level counters, 128 B apart

// Tuning and test options of a handle (qd_set_option, include/quandary_amd.h).  Every key can also be given as the environment
// variable QD_<KEY IN CAPITALS>; the environment is read ONCE, when the handle is created, as an override for tests and measurements.
struct TuneOpts {
  int var = -1;            // "var": force a kernel variant of qd_device.h (-1 = automatic choice)
  int force_neumann = 0;   // "force_neumann": ignore linearsolver_type = gmres
  int no_mfma = 0;         // "no_mfma": dense operator on the vector kernels
  int big_team = 0;        // "big_team": workgroups per initial condition of the global-memory sweeps (0 = automatic)
  int big_spread = -1;     // "big_spread": team members dealt over all XCDs (1), kept on one (0), default (-1)
  int big_blocked = 2;     // "big_blocked": a team member owns a contiguous block of the state (1; 2: the members of an XCD own neighbouring blocks) or every 'team'th row of 1024 elements (0)
  int f32_sb = -1;         // "f32_sb": slot bits of the fp32-mixed 2^4 kernel
  int lean64_sb = 0;       // "lean64_sb": elements per thread of the fp64 2^5 kernels as a power of two (0 = automatic: 2 elements on 512 threads for batches of
                           // at most one state per CU, else 4 on 256; 1 / 2 force)
  int no_lean64 = 0;       // "no_lean64": 2^5 / 2^4 Lindblad on the general slot kernels
  int no_collean = 0;      // "no_collean": 3 x 20-class systems on the general column kernel
  int no_col_krylov = 0;   // "no_col_krylov": the Krylov solver of 3 x 20-class systems on the general column kernel (A/B against the lean one [r6])
  int col_ept = 0;         // "col_ept": columns per wave of the lean column kernels (0 = automatic)
  double standin_tau = 1e-3;  // "standin_tau": error-estimate factor of the stationary iterations that serve gmres requests (0 = plain update-norm rule)
  int krylov_restart = 14;    // "krylov_restart": restart length of the generic path of the lean column kernels' Krylov solver (1 .. 14; KSPGMRESSetRestart)
  double krylov_tau = 0.1;    // "krylov_tau": the one-vector path of the Krylov solvers accepts at residual <= krylov_tau x the reference's tolerance (1 = at the tolerance itself)
  int no_plain = 0;        // "no_plain": 1 / 2 / 3 = forward / adjoint / both sweeps of the small systems on the general instantiation (A/B)
  int col_skip = 1;        // "col_skip": the lean column solver skips stopping tests up to two / three passes before the previous sub-step's count (0 = test every pass)
  int col_slices = 0;      // "col_slices": time slices of the lean column sweeps (0 = automatic, 1 = none, k = force k)
  int col_min_n = 33;      // "col_min_n": smallest density-matrix dimension N the lean column kernels take over from the eight-elements-per-thread kernel
  int gmres_poly = 0;      // "gmres_poly": degree of the polynomial preconditioner (0 = tuned, 1 = none)
  int gmres_split = -1;    // "gmres_split": linearsolver_type = gmres served by the diagonal-split iteration under GMRES's stopping rule where that
                           // iteration contracts fast (-1 = there, 0 = never: always the Krylov kernels, 1 = wherever it is built)
  int neumann_split = -1;  // "neumann_split": diagonal-split Neumann iteration (-1 = where it pays, 0 = never, 1 = wherever it is built)
  double traj_budget_mb = 0.0;  // "traj_budget_mb": pretend the trajectory budget is this small (chunked re-propagation)
  double sched_wait_s = 0.0;    // "sched_wait_s": seconds a time slice may wait for its predecessor (0 = automatic: 4 s x the processes that share the
                                // device (QD_DEVICE_SHARERS, set by the launchers that put several ranks on one GPU) x the slice length in units of 1000 steps)
  int set(const char* key, const char* value);  // 0 = ok, -1 = unknown key / bad value
  void load_env();
};

struct LaunchCfg {
  int var;    // kernel variant (elements per thread, register budget, LDS double buffering; qd_device.h)
  int qubit;  // stencil family: 0 general (runtime level counts), 1 all-qubit bit tricks, 2 dense user Hamiltonians
  int block;  // threads per block (one block per initial condition)
  int gmres;  // 0 Neumann, 1 GMRES with the Krylov basis in LDS, 2 GMRES with the basis in global memory
  int team;   // workgroups per initial condition (qd_big.h; 1 everywhere else)
  int spread; // team members dealt over all XCDs instead of one
  int blocked; // team members own contiguous blocks of the state
  int noplain; // A/B switch (option no_plain): bit 0 keeps the forward sweep, bit 1 the adjoint sweep off the PLAIN instantiation
  size_t lds;
};

// stopping rule of a stand-in for GMRES (SweepArgs::standin_tau2): d, dprev = squared update norms of this and the previous iteration
// (dprev = d in the first one: no contraction estimate yet), thr = the threshold d has already passed
__device__ __forceinline__ bool standin_ok(float tau2, float d, float dprev, float thr) { return tau2 == 0.f || d * d <= tau2 * thr * dprev; }

// the sweep runs on the PLAIN instantiation of the small-system kernels (qd_device.h): variants 0 / 1, Neumann kernels, an
// implicit-midpoint stepper, no in-loop penalty, no dpdm penalty
inline bool plain_sweep(const SweepArgs& a, const LaunchCfg& cfg, int adjoint) {
  return !(cfg.noplain & (adjoint ? 2 : 1)) && (cfg.var == 0 || cfg.var == 1) && !cfg.gmres && !a.stepper_ee && !(a.gamma_penalty > 1e-13) && !(a.gamma_dpdm > 1e-13 && !a.S.lindblad);
}

// kernel launch wrappers implemented in qd_kernels.hip; all return hipError_t
hipError_t launch_controls(const DevCtlDesc& d, const double* params, const double* times, const double* hs, int nrows,
                           double* table, int cs, hipStream_t st);
hipError_t launch_controls2(const DevCtlDesc& d, const double* params, const double* times, const double* hs, int nrows,
                            double* table, const double* times2, const double* hs2, int nrows2, double* table2, int cs,
                            unsigned long long* zero_me, hipStream_t st);
hipError_t launch_apply(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb,
                        const LaunchCfg& cfg, hipStream_t st);
hipError_t launch_forward(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st);
hipError_t launch_adjoint(const SweepArgs& a, const LaunchCfg& cfg, hipStream_t st);
hipError_t launch_gmat(const DevSys& S, const double* g0, const double* table, int cs, int nrows, double* gtab, hipStream_t st);
hipError_t launch_objective(const DevSys& S, const DevTarget& tg, const double* x, int nb, double* out4, hipStream_t st);
hipError_t launch_seed(const DevSys& S, const DevTarget& tg, const double* x, const double* rbar_ibar, int nb, double* xbar,
                       hipStream_t st);
hipError_t launch_partial_sums(const double* res, int nb, const double* w, double inv_ninit, const qd_penalty& pen, const double* etable,
                               int cs, int Q, int nstep, double* sums, hipStream_t st);
hipError_t launch_seed_weights(const double* sums, const double* w, int nb, int objective_type, int lindblad, double* rbib, hipStream_t st);
hipError_t launch_reduce_coeff(const double* coeff, int nb, int ncol, double* sum, int accumulate, hipStream_t st);
hipError_t launch_grad(const DevCtlDesc& d, const double* params, const double* table, int cs, int nsub, const double* coeffsum,
                       const double* etable, int nstep, double ebar, double* grad, int ndesign, hipStream_t st);
// fp32-mixed sweeps of all-qubit Lindblad systems (qd_q32.hip); the trajectory is [nsub+1][nb][dim] float2
hipError_t launch_forward_f32(const SweepArgs& a, const TuneOpts& o, hipStream_t st);
hipError_t launch_adjoint_f32(const SweepArgs& a, const TuneOpts& o, hipStream_t st);
hipError_t launch_apply_f32(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb, int nrep, int mfma,
                            const TuneOpts& o, hipStream_t st);
// the same lean slot kernel instantiated in fp64: Neumann sweeps of the 2^5 Lindblad system (QD_PRECISION_F64)
bool lean64_available(const DevSys& S, const TuneOpts& o);
hipError_t launch_forward_lean64(const SweepArgs& a, const TuneOpts& o, hipStream_t st);
hipError_t launch_adjoint_lean64(const SweepArgs& a, const TuneOpts& o, hipStream_t st);
hipError_t launch_apply_lean64(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb, hipStream_t st);
// lean column kernels (qd_col.hip): Lindblad Neumann sweeps of density matrices with 33..64 rows and runtime level counts
bool collean_available(const DevSys& S, const TuneOpts& o);
int col_slices(int nb, int ntime, const TuneOpts& o);  // time slices of a lean column sweep (1 = none)
size_t col_krylov_doubles(int nb, int nslice);  // size of SweepArgs::kry for the Krylov solver of the lean column kernels
hipError_t launch_forward_col(const SweepArgs& a, const TuneOpts& o, hipStream_t st);
hipError_t launch_adjoint_col(const SweepArgs& a, const TuneOpts& o, hipStream_t st);
hipError_t launch_apply_col(const DevSys& S, const double* ctlrow, int transpose, const double* x, double* y, int nb, const TuneOpts& o, hipStream_t st);
LaunchCfg pick_config(const DevSys& S, int nb, const TuneOpts& o, bool want_gmres = false, bool adjoint = false);
size_t krylov_doubles(const DevSys& S, int nb);
size_t big_work_doubles(const DevSys& S, int nb);
hipError_t launch_big_table(const DevSys& S, double* ecoef, unsigned* edig, hipStream_t st);
int variant_max_block(int var);  // 0 for an unknown variant  // size of SweepArgs::kry for LaunchCfg::gmres == 2

void set_error(const std::string& msg);

}  // namespace qd
