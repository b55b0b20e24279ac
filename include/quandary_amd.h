/*
 * quandary_amd.h — C ABI of the MI355X-native forward/adjoint propagator.
 *
 * This is the drop-in boundary for ONE hot path of LLNL/quandary: propagate a
 * batch of initial conditions through the Lindblad/Schroedinger time stepper,
 * sweep the discrete adjoint backwards and accumulate the control-parameter
 * gradient.  All entry points are extern "C", take plain pointers and sizes,
 * return 0 on success and a negative QD_ERR_* code on failure (the message is
 * available from qd_last_error()).  Nothing here calls exit().
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to the reference repository):
 *
 *   qd_create / qd_destroy      MasterEq ctor + MatShellCtx  (src/mastereq.cpp:14-155,
 *                               include/mastereq.hpp:20-42), Oscillator ctor
 *                               (src/oscillator.cpp:10-214), ImplMidpoint ctor
 *                               (src/timestepper.cpp:522-556)
 *   qd_set_params               MasterEq::setControlAmplitudes (src/mastereq.cpp:693-707)
 *   qd_eval_controls            Oscillator::evalControl (src/oscillator.cpp:281-337)
 *   qd_apply_rhs                MatMult / MatMultTranspose on the MATSHELL, i.e.
 *                               applyRHS_matfree<...> and ..._transpose<...>
 *                               (src/mastereq.cpp:1278-2893, dispatch :2976-3239),
 *                               preceded by MasterEq::assemble_RHS(t) (:657-678)
 *   qd_forward                  TimeStepper::solveODE over the local batch
 *                               (src/timestepper.cpp:96-181) incl. the in-loop
 *                               penalties (:256-298, :342-369, :444-455)
 *   qd_adjoint                  TimeStepper::solveAdjointODE over the local batch
 *                               (src/timestepper.cpp:184-253), ImplMidpoint::evolveBWD
 *                               (:631-694), compute_dRHS_dParams_matfree
 *                               (src/mastereq.cpp:970-1276)
 *   qd_optim_*                  OptimProblem::evalF / evalGradF (src/optimproblem.cpp:224-538)
 *                               with OptimTarget (src/optimtarget.cpp:325-897) and
 *                               Gate::assembleGate/applyGate (src/gate.cpp:88-283)
 *
 * State layout (same as the reference, docs/mkdocs/user_guide.md:305-306): one
 * state is 2*dim doubles, blocked [u ; v] (all real parts, then all imaginary
 * parts); dim = N (Schroedinger) or N^2 (Lindblad, column-major vectorisation
 * vec index = row + col*N, src/util.cpp:150-152); oscillator 0 is the slowest
 * index inside a Hilbert-space index.  A batch is nb states back to back.
 *
 *   qd_set_hamiltonian          HamiltonianFileReader + initSparseMatSolver + applyRHS_sparsemat
 *                               (src/hamiltonianfilereader.cpp, src/mastereq.cpp:192-967) for
 *                               user-supplied Hamiltonians
 *
 * Units follow the reference config file: frequencies in GHz (multiplied by
 * 2*pi inside, src/mastereq.cpp:29-37, src/oscillator.cpp:15-21), times in ns.
 *
 * Supported sizes (QD_ERR_UNSUPPORTED beyond): state dimension dim <= QD_MAX_DIM = 2^22 (one
 * workgroup owns one initial condition; up to 4096 the state lives in the CU's LDS, above it the
 * vectors of a step live in global memory and are exchanged through L2; IMR family);
 * 1..8 oscillators, Schroedinger and Lindblad (the reference's matrix-free templates stop at five, its sparse-matrix path does
 * not: six to eight Lindblad oscillators run on the general stencil - 2^6, dim 4096, in LDS, anything larger in global memory);
 * at most 256 / 64 / 32 / 16 levels per oscillator for <= 4 / 5 / 6 / 7-8 oscillators; user-supplied
 * Hamiltonians: 1..8 oscillators, table of G(t) <= 16 GB.  Control segments: "spline", "spline0", "step" and
 * "spline_amplitude" (the last one forward only, as in the reference: src/oscillator.cpp:350-356).  There is no CPU
 * fallback.
 */
#ifndef QUANDARY_AMD_H
#define QUANDARY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QD_MAX_OSC 8
#define QD_MAX_PAIRS (QD_MAX_OSC * (QD_MAX_OSC - 1) / 2)
#define QD_MAX_DIM (1 << 22)

/* error codes */
#define QD_OK 0
#define QD_ERR_INVALID (-1)   /* bad argument / inconsistent description           */
#define QD_ERR_UNSUPPORTED (-2) /* valid in the reference but not built here       */
#define QD_ERR_DEVICE (-3)    /* HIP runtime error (no device, launch failure ...) */
#define QD_ERR_NOMEM (-4)     /* host or device allocation failed                  */
#define QD_ERR_STATE (-5)     /* call order violated (e.g. adjoint before forward) */

/* LindbladType, include/defs.hpp */
enum { QD_LINDBLAD_NONE = 0, QD_LINDBLAD_DECAY = 1, QD_LINDBLAD_DEPHASE = 2, QD_LINDBLAD_BOTH = 3 };
/* ControlType, include/defs.hpp.  BSPLINEAMP carries no gradient in the reference (src/oscillator.cpp:350-356):
 * the gradient entry points return QD_ERR_UNSUPPORTED for it. */
enum { QD_CTRL_NONE = 0, QD_CTRL_BSPLINE = 1, QD_CTRL_BSPLINE0 = 2, QD_CTRL_STEP = 3, QD_CTRL_BSPLINEAMP = 4 };
/* timestepper = IMR | IMR4 | IMR8 | EE, src/main.cpp:357-366 */
enum { QD_STEPPER_IMR = 0, QD_STEPPER_IMR4 = 1, QD_STEPPER_IMR8 = 2, QD_STEPPER_EE = 3 };
/* linearsolver_type = gmres | neumann, src/main.cpp:342-350 */
enum { QD_LINSOLVE_GMRES = 0, QD_LINSOLVE_NEUMANN = 1 };
/* InitialConditionType, include/defs.hpp / src/optimtarget.cpp:42-53 */
enum {
  QD_INIT_FROMFILE = 0, QD_INIT_PURE = 1, QD_INIT_ENSEMBLE = 2, QD_INIT_DIAGONAL = 3,
  QD_INIT_BASIS = 4, QD_INIT_THREESTATES = 5, QD_INIT_NPLUSONE = 6, QD_INIT_PERFORMANCE = 7
};
/* TargetType / ObjectiveType, include/defs.hpp */
enum { QD_TARGET_GATE = 0, QD_TARGET_PURE = 1, QD_TARGET_FROMFILE = 2 };
enum { QD_OBJ_JFROBENIUS = 0, QD_OBJ_JTRACE = 1, QD_OBJ_JMEASURE = 2 };

/* Physical system: the MatShellCtx parameter block (include/mastereq.hpp:20-42)
 * in config-file units (src/main.cpp:191-316).  Pair order is 01,02,...,0(Q-1),12,...
 * (src/mastereq.cpp:2432-2441). */
typedef struct qd_system {
  int32_t nosc;
  int32_t lindblad_type;               /* QD_LINDBLAD_*                      */
  int32_t nlevels[QD_MAX_OSC];
  int32_t nessential[QD_MAX_OSC];
  double transfreq[QD_MAX_OSC];        /* GHz                                */
  double rotfreq[QD_MAX_OSC];          /* GHz                                */
  double selfkerr[QD_MAX_OSC];         /* GHz                                */
  double crosskerr[QD_MAX_PAIRS];      /* GHz                                */
  double Jkl[QD_MAX_PAIRS];            /* GHz                                */
  double decay_time[QD_MAX_OSC];       /* T1 in ns, <=1e-14 disables         */
  double dephase_time[QD_MAX_OSC];     /* T2 in ns, <=1e-14 disables         */
} qd_system;

/* Control parameterisation (src/oscillator.cpp:45-132, src/controlbasis.cpp).
 * Segments are listed oscillator by oscillator; design vector layout per
 * oscillator is [segment][carrier][2*nsplines] for spline / spline0 (src/controlbasis.cpp:58-59),
 * [segment][carrier][nsplines amplitudes, phase] for spline_amplitude (:127-140) and one parameter
 * (the relative width of the step) for step (:195-206; one carrier: the reference's index
 * skip + 2*carrier runs past the segment's single parameter for more). */
typedef struct qd_controls {
  int32_t enforce_bc;                  /* control_enforceBC                  */
  int32_t nseg_total;
  const int32_t* seg_osc;              /* [nseg_total] owning oscillator     */
  const int32_t* seg_type;             /* [nseg_total] QD_CTRL_*             */
  const int32_t* seg_nsplines;         /* [nseg_total]                       */
  const double* seg_tstart;            /* [nseg_total] ns                    */
  const double* seg_tstop;             /* [nseg_total] ns                    */
  const int32_t* ncarrier;             /* [nosc]                             */
  const double* carrier_freq;          /* concatenated, GHz                  */
  int32_t npipulse;                    /* apply_pipulse entries, already expanded
                                          to every oscillator (src/main.cpp:249-277) */
  const int32_t* pipulse_osc;          /* [npipulse]                         */
  const double* pipulse_tstart;        /* [npipulse]                         */
  const double* pipulse_tstop;         /* [npipulse]                         */
  const double* pipulse_amp;           /* [npipulse] (0 for non-target osc)  */
  const double* seg_param;             /* [nseg_total][3] or NULL.  step: amp1, amp2 (rad/ns, as given in the
                                          config), tramp (ns); spline_amplitude: scaling, -, -         */
} qd_controls;

typedef struct qd_time {
  int32_t ntime;
  double dt;
} qd_time;

typedef struct qd_solver {
  int32_t stepper;                     /* QD_STEPPER_*                       */
  int32_t linsolve;                    /* QD_LINSOLVE_*                      */
  int32_t maxiter;                     /* linearsolver_maxiter               */
  double abstol;                       /* reference: 1e-10 (timestepper.cpp:536) */
  double reltol;                       /* reference: 1e-20 (timestepper.cpp:535) */
  /* Stopping tests.  Neumann (timestepper.cpp:716-717): the update norm ||y_{m+1} - y_m|| against abstol, and against reltol times the
   * first update.  The sweep kernels accumulate the squared update per thread in fp64, scale it by 1 / abstol^2 and reduce it over the
   * workgroup in FP32 (one DPP / LDS round instead of two fp64 ones): the value is only compared with 1, so the fp32 reduction moves the
   * stopping point by at most one part in 1e7 of the tolerance - the iteration counts equal the oracle's on every tested shape
   * (tests assert |applications per step - oracle| < 0.25).  GMRES (Krylov kernels): Hessenberg problem, projections and the
   * recurrence residual in fp64, stop at max(reltol ||b||, abstol) as KSPGMRES (timestepper.cpp:541-550). */
} qd_solver;

/* What the forward sweep needs to evaluate objective-dependent terms inside
 * the time loop (weighted-J penalty, src/timestepper.cpp:262-270) and at the
 * end (OptimTarget::evalJ, src/optimtarget.cpp:712-799). */
typedef struct qd_target {
  int32_t target_type;                 /* QD_TARGET_*                        */
  int32_t objective_type;              /* QD_OBJ_*                           */
  int32_t purestate_id;                /* Hilbert-space index m for PURE     */
  const double* target_states;         /* [nb][2*dim] for GATE/FROMFILE, else NULL */
  const double* purity;                /* [nb] Tr(rho0^2) (optimtarget.cpp:705-707) */
} qd_target;

typedef struct qd_penalty {
  double gamma_penalty;                /* optim_penalty                      */
  double penalty_param;                /* optim_penalty_param                */
  double gamma_penalty_dpdm;           /* optim_penalty_dpdm (Schroedinger)  */
  double gamma_penalty_energy;         /* optim_penalty_energy               */
} qd_penalty;

/* Per-initial-condition results of one forward sweep. */
typedef struct qd_forward_out {
  double* final_states;                /* [nb][2*dim] or NULL                */
  double* penalty_integral;            /* [nb] TimeStepper::penalty_integral */
  double* penalty_dpdm;                /* [nb] TimeStepper::penalty_dpdm     */
  double* energy_penalty;              /* [1]  TimeStepper::energy_penalty_integral */
  double* J_re;                        /* [nb] OptimTarget::evalJ            */
  double* J_im;                        /* [nb]                               */
  double* fid_re;                      /* [nb] HilbertSchmidtOverlap(.,false) */
  double* fid_im;                      /* [nb]                               */
} qd_forward_out;

typedef struct qd_handle qd_handle;

const char* qd_last_error(void);
const char* qd_version(void);
/* number of HIP devices visible, or a negative QD_ERR_DEVICE */
int qd_device_count(void);

int qd_create(const qd_system* sys, const qd_controls* ctl, const qd_time* tg, const qd_solver* sol,
              int device_ordinal, qd_handle** out);
void qd_destroy(qd_handle* h);

/* sizes derived from the description */
int qd_dim(const qd_handle* h);        /* dim = N or N^2                       */
int qd_dim_rho(const qd_handle* h);    /* N                                    */
int qd_dim_ess(const qd_handle* h);    /* prod nessential                      */
int qd_ndesign(const qd_handle* h);    /* number of control parameters         */

/* User-supplied Hamiltonians, the reference's `hamiltonian_file_Hsys` / `hamiltonian_file_Hc` model
 * (src/hamiltonianfilereader.cpp, applied by applyRHS_sparsemat src/mastereq.cpp:743-967): dense complex
 * N x N matrices (N = qd_dim_rho), row-major, in rad/ns.  hc_re / hc_im hold nosc matrices back to back
 * (NULL = no control Hamiltonians, the reference's "none").  They REPLACE the standard Hamiltonian model of
 * the qd_system description - detuning, Kerr terms, dipole coupling, ladder-operator controls:
 *     H(t) = Hsys + sum_k p_k(t) Re(Hc_k) + i q_k(t) Im(Hc_k),
 * the T1/T2 dissipators of qd_system stay.  Call before the first sweep. */
int qd_set_hamiltonian(qd_handle* h, const double* hsys_re, const double* hsys_im, const double* hc_re, const double* hc_im);

int qd_set_params(qd_handle* h, const double* alpha, int ndesign);
/* p_k(t), q_k(t) for every oscillator at nt times; pq is [nt][nosc][2] */
int qd_eval_controls(qd_handle* h, const double* times, int nt, double* pq);
/* y = M(t) x (transpose=0) or M(t)^T x (transpose=1) for nb states */
int qd_apply_rhs(qd_handle* h, double t, int transpose, const double* x, double* y, int nb);

/* Trajectory access after qd_forward(..., store_trajectory=1): copies state n
 * (0..ntime) of every local initial condition, [nb][2*dim]. */
int qd_get_state(qd_handle* h, int timestep, double* x);

/* Observables of the stored trajectory, reduced on the device (Oscillator::expectedEnergy / population
 * src/oscillator.cpp:430-566, MasterEq::expectedEnergy / population src/mastereq.cpp:2897-2974; what
 * Output::writeTrajectoryDataFiles prints, src/output.cpp:203-273) for the time steps 0, stride, 2 stride, ... <= ntime
 * (nout = ntime / stride + 1 of them) of every local initial condition.  Any output pointer may be NULL.
 *   expected             [nout][nb][nosc]              expected energy level of oscillator k
 *   population           [nout][nb][sum_k nlevels[k]]  level populations, oscillator k's block after those of 0..k-1
 *   expected_composite   [nout][nb]                    sum_I I P(I)
 *   population_composite [nout][nb][N]                 P(I) = rho_II or |psi_I|^2 */
int qd_get_observables(qd_handle* h, int stride, double* expected, double* population, double* expected_composite,
                       double* population_composite);

int qd_set_target(qd_handle* h, const qd_target* tgt, int nb);
int qd_set_penalty(qd_handle* h, const qd_penalty* pen);

int qd_forward(qd_handle* h, const double* x0, int nb, int store_trajectory, qd_forward_out* out);
/* xbarT: [nb][2*dim] terminal adjoint seed; jbar: [nb][3] = beta_i*{gamma_penalty,
 * gamma_dpdm, gamma_energy} as passed to solveAdjointODE (timestepper.cpp:184);
 * grad: [ndesign] summed over the local batch (overwritten). */
int qd_adjoint(qd_handle* h, const double* xbarT, const double* jbar, int nb, double* grad);

/* Mean number of RHS applications per time step and initial condition in the
 * last forward sweep (1 + linear-solver iterations). */
double qd_last_mean_applies(const qd_handle* h);
/* Milliseconds spent in device kernels of the last forward / adjoint sweep
 * (hipEvent bracket on the handle's stream). */
double qd_last_forward_ms(const qd_handle* h);
double qd_last_adjoint_ms(const qd_handle* h);
/* Workgroups per initial condition in the last sweep: 1, or the team size when a large state (dim > 4096) with few
 * initial conditions was spread over several CUs. */
int qd_last_team(const qd_handle* h);
/* Which iteration solved the linear systems of the last sweep: QD_SOLVER_NEUMANN (requested), QD_SOLVER_KRYLOV (the in-kernel GMRES,
 * KSPGMRES iteration for iteration, src/timestepper.cpp:541-550), QD_SOLVER_GMRES_AS_SPLIT / QD_SOLVER_GMRES_AS_NEUMANN (a gmres
 * request served by a stationary iteration under GMRES's stopping rule: option gmres_split), QD_SOLVER_NONE (explicit Euler). */
#define QD_SOLVER_NONE 0
#define QD_SOLVER_NEUMANN 1
#define QD_SOLVER_KRYLOV 2
#define QD_SOLVER_GMRES_AS_SPLIT 3
#define QD_SOLVER_GMRES_AS_NEUMANN 4
int qd_last_solver(const qd_handle* h);
/* Measurement hook for the secondary (fp64 vector) roofline: runs a register-only
 * v_fma_f64 micro-benchmark on the device and returns the sustained TFLOP/s in
 * *tflops (the fp64 peak is to be measured, not quoted). */
int qd_measure_fp64_peak(int device_ordinal, double* tflops);
/* the same with v_pk_fma_f32 (fp32-mixed sweeps: the packed form is what the 157.3 TFLOP/s fp32 vector peak is quoted for) */
int qd_measure_fp32_peak(int device_ordinal, double* tflops);

/* ---------------------------------------------------------------------------
 * Objective level: OptimProblem::evalF / evalGradF over the local shard of
 * initial conditions (src/optimproblem.cpp:224-538).
 * ------------------------------------------------------------------------- */
typedef struct qd_objective {
  int32_t initcond_type;               /* QD_INIT_*                          */
  int32_t n_init_ids;                  /* entries after the keyword          */
  int32_t init_ids[QD_MAX_OSC];        /* oscillator ids (or levels for PURE) */
  const double* init_data;             /* FROMFILE: file content, 2*dim_ess(^2) */
  int32_t target_type;                 /* QD_TARGET_*                        */
  int32_t target_pure_levels[QD_MAX_OSC]; /* PURE: level per oscillator      */
  const double* gate_re;               /* GATE: V (dim_ess x dim_ess, row-major), lab frame */
  const double* gate_im;
  double gate_rot_freq[QD_MAX_OSC];    /* GHz                                */
  const double* target_data;           /* FROMFILE: file content             */
  int32_t objective_type;              /* QD_OBJ_*                           */
  int32_t nweights;                    /* optim_weights entries              */
  const double* weights;
  double gamma_tik;                    /* optim_regul                        */
  int32_t tik0;                        /* optim_regul_tik0                   */
  const double* alpha0;                /* [ndesign] initial guess for tik0   */
  qd_penalty penalty;
  double gamma_penalty_variation;      /* optim_penalty_variation            */
} qd_objective;

/* The seven partial sums the reference all-reduces over comm_init
 * (src/optimproblem.cpp:292-298, :454-460), in this order. */
enum { QD_SUM_PENALTY = 0, QD_SUM_DPDM = 1, QD_SUM_ENERGY = 2, QD_SUM_COST_RE = 3,
       QD_SUM_COST_IM = 4, QD_SUM_FID_RE = 5, QD_SUM_FID_IM = 6, QD_NSUMS = 7 };

typedef struct qd_objective_value {
  double objective;                    /* getObjective                       */
  double cost;                         /* getCostT                           */
  double regul;                        /* getRegul                           */
  double penalty;                      /* getPenalty                         */
  double penalty_dpdm;                 /* getPenaltyDpDm                     */
  double penalty_energy;               /* getPenaltyEnergy                   */
  double penalty_variation;            /* getPenaltyVariation                */
  double fidelity;                     /* getFidelity                        */
} qd_objective_value;

typedef struct qd_optim qd_optim;

/* ninit_global initial conditions are sharded contiguously over nranks
 * (iinit_global = rank*ninit_local + iinit, src/optimproblem.cpp:248). */
int qd_optim_create(qd_handle* h, const qd_objective* obj, int rank, int nranks, qd_optim** out);
void qd_optim_destroy(qd_optim* o);
int qd_optim_ninit(const qd_optim* o);        /* global                        */
int qd_optim_ninit_local(const qd_optim* o);
/* initial state / id of local initial condition i: x0 is [2*dim] */
int qd_optim_initial_state(qd_optim* o, int iinit_local, double* x0, int* initid);
int qd_optim_target_state(qd_optim* o, int iinit_local, double* xtarget);

/* Forward sweep of the local shard; partial[QD_NSUMS] are this rank's sums. */
int qd_optim_forward_local(qd_optim* o, const double* alpha, int store_trajectory, double* partial);
/* Objective from the globally reduced sums. */
int qd_optim_finalize(qd_optim* o, const double* alpha, const double* global_sums, qd_objective_value* val);
/* Adjoint sweep of the local shard seeded from the GLOBAL sums
 * (src/optimproblem.cpp:495-519); grad_local[ndesign] still has to be summed
 * over ranks (:527).  Rank 0 adds the Tikhonov / variation terms (:356-372). */
int qd_optim_adjoint_local(qd_optim* o, const double* alpha, const double* global_sums, double* grad_local);

/* Both sweeps of the local shard in ONE call, no collective, for a host that reduces by itself (the reference's two MPI_Allreduce,
 * src/optimproblem.cpp:454-460 and :527): partial[QD_NSUMS] and grad_local[ndesign] are this rank's sums, WITHOUT the regularisation
 * terms (:356-372; the caller adds them once after its reduction, then qd_optim_finalize on the reduced sums).  A shard whose stored
 * stages exceed HBM is propagated and reversed in chunks in one pass - the two-call form above has to propagate it twice.  Available
 * wherever the adjoint seeds do not depend on the reduced cost (src/optimtarget.cpp:889-895): QD_ERR_STATE for Schroedinger + Jtrace. */
int qd_optim_gradient_local(qd_optim* o, const double* alpha, double* partial, double* grad_local);

/* Single-rank convenience wrappers (nranks must be 1). */
int qd_optim_evalF(qd_optim* o, const double* alpha, qd_objective_value* val);
int qd_optim_evalGradF(qd_optim* o, const double* alpha, qd_objective_value* val, double* grad);
/* Chunks of the last gradient evaluation: 1 when the shard's stored trajectory fitted in HBM; otherwise the shard was propagated and
 * reversed in this many batches of initial conditions (the storage problem of src/timestepper.cpp:38-48 at 288 GB) - in one pass
 * (forward + adjoint per chunk) wherever the adjoint seeds do not depend on the reduced cost, i.e. everywhere but Schroedinger + Jtrace. */
int qd_optim_last_chunks(const qd_optim* o);

/* ---------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, initial conditions sharded over the ranks.
 * Replaces the reference's comm_init communicator (src/main.cpp:133-177) and its
 * MPI_Allreduce calls (src/optimproblem.cpp:292-298, :454-460, :527) by RCCL
 * (ncclAllReduce over xGMI) on the handle's HIP stream; partial sums and the
 * gradient never leave HBM between the sweeps and the collectives.
 * ------------------------------------------------------------------------- */
#define QD_COMM_ID_BYTES 128           /* = NCCL_UNIQUE_ID_BYTES                */
typedef struct qd_comm qd_comm;
/* rank 0: generate the id (ncclGetUniqueId) and hand its bytes to every rank by any means */
int qd_comm_unique_id(unsigned char id[QD_COMM_ID_BYTES]);
int qd_comm_create(const unsigned char id[QD_COMM_ID_BYTES], int rank, int nranks, int device_ordinal, qd_comm** out);
/* MPI-free bootstrap through a path every rank can see.  RCCL backend: rank 0 writes the id file, every other rank echoes the token it read
 * together with a nonce of its own (path.ack<rank>) and rank 0 answers each echo with path.go<rank> carrying that nonce; a rank enters
 * ncclCommInitRank only on a go file that returns its own nonce - leftover files of a crashed run are never acted upon;
 * QD_JOB_ID (when the launcher sets one) additionally separates jobs.  Host backend (below): the path only names the shared-memory segment.
 * Backend: environment QD_COMM_BACKEND = rccl | host | auto (default auto: host when the ranks OF THIS NODE - QD_LOCAL_SIZE, set by the
 * launchers; nranks without it - exceed the visible GPUs, i.e. when ranks share a device, which RCCL refuses; refused when such a launch
 * spans nodes). */
int qd_comm_create_from_file(const char* path, int rank, int nranks, int device_ordinal, double timeout_s, qd_comm** out);
/* HOST backend: the ranks of ONE node reduce through a POSIX shared-memory segment named after `name` (the same string on every rank of a
 * job, different for concurrent jobs).  Same call sites and buffers as the RCCL backend (qd_optim_evalF_dist / evalGradF_dist,
 * qd_comm_allreduce), summation in rank order on every rank (bit-identical results on all ranks); the reduction buffer makes one round
 * trip HBM -> pinned host -> HBM per collective.  For ranks that share a GPU (mpirun -np 4 on a one-GPU box) and nodes without RCCL.
 * Replaces the same MPI_Allreduce calls (src/optimproblem.cpp:292-298, :454-460, :527).  At most 64 ranks. */
int qd_comm_create_host(const char* name, int rank, int nranks, int device_ordinal, double timeout_s, qd_comm** out);
/* 0 = RCCL, 1 = host shared memory */
int qd_comm_backend(const qd_comm* c);
void qd_comm_destroy(qd_comm* c);
int qd_comm_size(const qd_comm* c);
int qd_comm_rank(const qd_comm* c);
/* host convenience: in-place all-reduce of n doubles, op 0 = sum, 1 = max (staged through HBM, blocking) */
int qd_comm_allreduce(qd_comm* c, double* buf, int n, int op);
int qd_comm_barrier(qd_comm* c);
/* evalF / evalGradF over ALL ranks (every rank calls with its own shard `o` created with the same rank / nranks as
 * `c`); every rank receives the same value and the complete gradient.  allreduce_ms (NULL or [2]): device time of the
 * objective-sum and the gradient collective. */
int qd_optim_evalF_dist(qd_optim* o, qd_comm* c, const double* alpha, qd_objective_value* val, double* allreduce_ms);
int qd_optim_evalGradF_dist(qd_optim* o, qd_comm* c, const double* alpha, qd_objective_value* val, double* grad, double* allreduce_ms);

/* ---------------------------------------------------------------------------
 * Arithmetic of the sweeps.  QD_PRECISION_F64 (default): everything IEEE double like the reference.
 * QD_PRECISION_F32MIXED (BASELINE config 5): the exchange vector in LDS, the stencil arithmetic, the linear-solver
 * iterates and the stored trajectory are fp32; the state / adjoint-state accumulators, every norm, objective sum and
 * gradient coefficient are fp64.  Built for all-qubit Lindblad systems with 4 or 5 oscillators, both linear solvers
 * (GMRES: Krylov basis as float2, Hessenberg problem in fp64), IMR family; QD_ERR_UNSUPPORTED elsewhere.  Call before
 * the first sweep.
 * ------------------------------------------------------------------------- */
enum { QD_PRECISION_F64 = 0, QD_PRECISION_F32MIXED = 1 };
int qd_set_precision(qd_handle* h, int precision);
/* Tuning and test options of a handle, as "key", "value" strings (the counterpart of PETSc's options database the reference is
 * steered with, e.g. -ksp_* for the linear solver of src/timestepper.cpp:541-550).  Integer values unless noted; "auto" restores
 * the default.  Keys: neumann_split (diagonal of M on the left-hand side of the Neumann iteration: same fixed point and stopping
 * rule, fewer iterations on systems whose level energies dominate; auto = where it pays, 0 = the reference's iteration everywhere),
 * gmres_split (linearsolver_type = gmres served by a stationary iteration under GMRES's stopping rule wherever that iteration provably
 * contracts fast - the reference's Neumann iteration, or the diagonal-split one on 3x20-class systems and on states beyond LDS: auto = there, 0 = always the
 * Krylov kernels), gmres_poly (degree of the polynomial preconditioner, 0 = tuned then frozen, 1 = none), krylov_tau (double: the one-vector
 * path of the lean kernels' Krylov solvers accepts at residual <= krylov_tau x the reference's tolerance, default 0.1), krylov_restart (restart length of those solvers' generic path, 1 .. 14), force_neumann, var (kernel variant), no_mfma,
 * no_lean64, lean64_sb, no_collean, no_col_krylov, col_ept, col_min_n, big_team, big_spread, big_blocked, f32_sb, traj_budget_mb (double).  Every key is also read from the
 * environment variable QD_<KEY> once, at qd_create (tests, measurements).  Unknown keys: QD_ERR_INVALID. */
int qd_set_option(qd_handle* h, const char* key, const char* value);
int qd_get_precision(const qd_handle* h);
/* Measurement hook: nrep chained forward applications y <- M(t) (1e-3 y) on nb states, starting from x, in fp32 by the
 * stencil kernel (mfma = 0) or as the dense Kronecker-factor product G rho - rho G on the fp32 matrix cores (mfma = 1,
 * five qubits only); *ms = device time of the launch.  profiles/HISTORY.md (section 4) records the comparison. */
int qd_bench_apply_f32(qd_handle* h, double t, const double* x, double* y, int nb, int nrep, int mfma, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* QUANDARY_AMD_H */
