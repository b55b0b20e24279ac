// quandary — config-file driver on top of the C ABI (include/quandary_amd.h).
//
// Host-side mirror of the reference's executable for the hot path: same command line
// (`quandary <config.cfg> [--quiet]`, src/util.cpp:27-79), same config format and defaults
// (src/config.cpp:19-74, src/main.cpp:56-366), same output files and formats (src/output.cpp:80-273,
// src/main.cpp:482-487).  Everything state-sized runs on the GPU behind qd_* calls; this file only
// parses, describes the problem, and writes files.  `runtype = optimization` uses a small projected
// L-BFGS here (the reference delegates to PETSc TAO, a third-party optimiser that is out of scope):
// it serves f and grad f from the same entry points and writes the same optim_history.dat columns.
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <set>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "quandary_amd.h"

typedef std::vector<std::string> strvec;

static void die(const std::string& msg) {
  fprintf(stderr, "\nERROR: %s\n", msg.c_str());
  exit(1);
}
#define QDCHK(call)                                                       \
  do {                                                                    \
    if ((call) != QD_OK) die(std::string(#call) + ": " + qd_last_error()); \
  } while (0)

// ---- Config (src/config.cpp:19-74): all blanks/tabs stripped, '#' and '/' start comments --------
struct Config : std::map<std::string, std::string> {
  void read(const std::string& fn) {
    std::ifstream f(fn);
    if (!f.is_open()) die("Cannot open config file '" + fn + "'");
    std::string line;
    while (std::getline(f, line)) {
      std::string s;
      for (char ch : line)
        if (ch != ' ' && ch != '\t' && ch != '\r') s.push_back(ch);
      if (s.empty() || s[0] == '#' || s[0] == '/') continue;
      const size_t pos = s.find('=');
      (*this)[s.substr(0, pos)] = pos == std::string::npos ? s : s.substr(pos + 1);
    }
  }
  bool has(const std::string& k) const { return count(k) > 0; }
  // Every parameter that is asked for is recorded once, with the value that was used (the file's text, or the default), in the order of
  // the first query: config_log.dat of the reference (export_param, include/config.hpp:141-148; written by src/main.cpp:382-393) - a
  // complete config file of the run that a later run can be started from.
  mutable std::vector<std::pair<std::string, std::string>> asked;
  mutable std::set<std::string> seen;
  template <typename T>
  void note(const std::string& k, const T& d) const {
    if (!seen.insert(k).second) return;
    std::ostringstream v;
    v.precision(17);
    if (has(k)) v << at(k);
    else v << d;
    asked.emplace_back(k, v.str());
  }
  std::string log_text() const {
    std::string t;
    for (auto& kv : asked) t += kv.first + " = " + kv.second + "\n";
    return t;
  }
  std::string str(const std::string& k, const std::string& d) const { note(k, d); return has(k) ? at(k) : d; }
  double dbl(const std::string& k, double d) const { note(k, d); return has(k) ? atof(at(k).c_str()) : d; }
  int integer(const std::string& k, int d) const { note(k, d); return has(k) ? atoi(at(k).c_str()) : d; }
  bool boolean(const std::string& k, bool d) const {
    note(k, d ? "true" : "false");
    if (!has(k)) return d;
    const std::string& v = at(k);
    return v == "yes" || v == "true" || v == "True" || v == "TRUE" || v == "YES" || v == "1";
  }
  strvec vstr(const std::string& k, const std::string& d) const {
    note(k, d);
    strvec out;
    std::stringstream ss(has(k) ? at(k) : d);
    std::string t;
    while (std::getline(ss, t, ',')) out.push_back(t);
    return out;
  }
  std::vector<double> vdbl(const std::string& k, double d) const {
    std::vector<double> out;
    note(k, d);
    if (!has(k)) return {d};
    for (auto& t : vstr(k, "")) out.push_back(atof(t.c_str()));
    return out;
  }
  std::vector<int> vint(const std::string& k, int d) const {
    std::vector<int> out;
    if (!has(k)) return {d};
    for (auto& t : vstr(k, "")) out.push_back(atoi(t.c_str()));
    return out;
  }
};

template <typename T>
static void copy_last(std::vector<T>& v, size_t n) {  // copyLast, src/util.hpp
  while (v.size() < n) v.push_back(v.back());
}

static std::vector<double> read_vector(const std::string& fn, size_t n) {  // src/util.cpp:566
  std::ifstream f(fn);
  if (!f.is_open()) die("Cannot read file '" + fn + "'");
  std::vector<double> v;
  double x;
  while (v.size() < n && (f >> x)) v.push_back(x);
  if (v.size() < n) die("File '" + fn + "' holds fewer values than expected");
  return v;
}

// ---- gates in essential dimensions, lab frame (src/gate.cpp:286-541) ------------------------------
static void gate_matrix(const strvec& tgt, int de, int Q, const std::string& dir, std::vector<double>& re, std::vector<double>& im) {
  re.assign((size_t)de * de, 0.0);
  im.assign((size_t)de * de, 0.0);
  auto R = [&](int r, int c) -> double& { return re[(size_t)r * de + c]; };
  auto I = [&](int r, int c) -> double& { return im[(size_t)r * de + c]; };
  const std::string& g = tgt[1];
  auto need = [&](int d) { if (de != d) die("gate '" + g + "' needs an essential dimension of " + std::to_string(d)); };
  if (g == "none") {  // dummy gate (src/gate.cpp:3-7): applyGate leaves the zero target untouched
  } else if (g == "xgate") { need(2); R(0, 1) = R(1, 0) = 1.0; }
  else if (g == "ygate") { need(2); I(0, 1) = -1.0; I(1, 0) = 1.0; }
  else if (g == "zgate") { need(2); I(0, 0) = 1.0; I(1, 1) = -1.0; }  // as the reference fills it (src/gate.cpp:331-332)
  else if (g == "hadamard") { need(2); const double v = 1. / sqrt(2.); R(0, 0) = R(0, 1) = R(1, 0) = v; R(1, 1) = -v; }
  else if (g == "cnot") { need(4); R(0, 0) = R(1, 1) = R(2, 3) = R(3, 2) = 1.0; }
  else if (g == "swap") { need(4); R(0, 0) = R(1, 2) = R(2, 1) = R(3, 3) = 1.0; }
  else if (g == "swap0q") {
    for (int i = 0; i < (1 << (Q - 2)); i++) R(2 * i, 2 * i) = 1.0;
    for (int i = (1 << (Q - 2)); i < (1 << (Q - 1)); i++) R(2 * i + 1, 2 * i + 1) = 1.0;
    for (int i = 0; i < (1 << (Q - 2)); i++) R(2 * i + 1, 2 * i + (1 << (Q - 1))) = R(2 * i + (1 << (Q - 1)), 2 * i + 1) = 1.0;
  } else if (g == "cqnot") {
    for (int k = 0; k < de - 2; k++) R(k, k) = 1.0;
    R(de - 2, de - 1) = R(de - 1, de - 2) = 1.0;
  } else if (g == "qft") {
    const double sq = sqrt((double)de);
    for (int j = 0; j < de; j++)
      for (int k = 0; k < de; k++) {
        R(j, k) = cos(2.0 * M_PI * j * k / de) / sq;
        I(j, k) = sin(2.0 * M_PI * j * k / de) / sq;
      }
  } else if (g == "file") {
    if (tgt.size() < 3) die("optim_target = gate, file needs a file name");
    auto v = read_vector(dir + tgt[2], (size_t)2 * de * de);
    for (int i = 0; i < de * de; i++) {  // column-major, real block then imaginary block
      R(i % de, i / de) = v[i];
      I(i % de, i / de) = v[i + (size_t)de * de];
    }
  } else die("Could not find target gate '" + g + "'");
}

// ---- output (src/output.cpp) ------------------------------------------------------------------------
struct Output {
  std::string datadir;
  int output_frequency = 1, optim_monitor_freq = 10;
  FILE* optimfile = nullptr;
  std::vector<strvec> outputstr;
  bool full = false, ecomp = false, pcomp = false;
  std::vector<bool> wexp, wpop;
  bool root = true;  // rank 0 writes the shared files (history, gradient, parameters, controls); every rank its own trajectories
  void init(const Config& cfg, int Q, bool is_root) {
    root = is_root;
    datadir = cfg.str("datadir", "./data_out");
    mkdir(datadir.c_str(), 0777);
    optim_monitor_freq = cfg.integer("optim_monitor_frequency", 10);
    output_frequency = cfg.integer("output_frequency", 1);
    optimfile = fopen(root ? (datadir + "/optim_history.dat").c_str() : "/dev/null", "w");
    if (!optimfile) die("cannot write to " + datadir);
    fprintf(optimfile,
            "#\"iter\"    \"Objective\"           \"||Pr(grad)||\"           \"LS step\"           \"F_avg\"           \"Terminal cost\"   "
            "      \"Tikhonov-regul\"        \"Penalty-term\"          \"State variation\"        \"Energy-term\"           \"Control "
            "variation\"\n");
    wexp.assign(Q, false);
    wpop.assign(Q, false);
    for (int i = 0; i < Q; i++) {
      outputstr.push_back(cfg.vstr("output" + std::to_string(i), "none"));
      for (auto& s : outputstr[i]) {
        if (s == "expectedEnergy") wexp[i] = true;
        if (s == "expectedEnergyComposite") ecomp = true;
        if (s == "population") wpop[i] = true;
        if (s == "populationComposite") pcomp = true;
        if (s == "fullstate") full = true;
      }
    }
  }
  void optim_row(int it, const qd_objective_value& v, double gnorm, double step) {
    fprintf(optimfile, "%05d  %1.14e  %1.14e  %.8f  %1.14e  %1.14e  %1.14e  %1.14e  %1.14e  %1.14e  %1.14e\n", it, v.objective, gnorm, step,
            v.fidelity, v.cost, v.regul, v.penalty, v.penalty_dpdm, v.penalty_energy, v.penalty_variation);
    fflush(optimfile);
  }
  void write_vec(const char* name, const std::vector<double>& x) {
    if (!root) return;
    FILE* f = fopen((datadir + "/" + name).c_str(), "w");
    for (double v : x) fprintf(f, "%1.14e\n", v);
    fclose(f);
  }
};

struct Problem {
  Config cfg;
  std::string cfgdir;
  int Q = 0, ntime = 0, N = 1, dim = 1, dim_ess = 1, ninit = 1;
  double dt = 0.0;
  bool lindblad = false;
  std::vector<int> nlevels, ness;
  qd_system sys{};
  qd_controls ctl{};
  qd_time tg{};
  qd_solver sol{};
  qd_objective obj{};
  // storage behind the pointers
  std::vector<int32_t> seg_osc, seg_type, seg_ns, ncar, pi_osc;
  std::vector<double> seg_t0, seg_t1, seg_par, cars, pi_t0, pi_t1, pi_amp, gate_re, gate_im, weights, init_data, target_data;
  std::vector<double> params0, bounds, transfreq;
};

static void build(Problem& P) {
  const Config& cfg = P.cfg;
  P.nlevels = cfg.vint("nlevels", 0);
  const int Q = P.Q = (int)P.nlevels.size();
  if (Q < 1 || Q > QD_MAX_OSC) die("nlevels: between 1 and 8 oscillators");
  P.ntime = cfg.integer("ntime", 1000);
  P.dt = cfg.dbl("dt", 0.01);
  const double total_time = P.ntime * P.dt;
  P.ness = P.nlevels;
  auto rn = cfg.vint("nessential", -1);
  if (rn[0] > -1)
    for (int i = 0; i < Q; i++) P.ness[i] = std::min(i < (int)rn.size() ? rn[i] : rn.back(), P.nlevels[i]);
  const std::string lind = cfg.str("collapse_type", "none");
  int lt = lind == "none" ? 0 : lind == "decay" ? 1 : lind == "dephase" ? 2 : lind == "both" ? 3 : -1;
  if (lt < 0) die("Unknown lindblad type: " + lind);
  P.lindblad = lt != 0;
  auto trans = cfg.vdbl("transfreq", 1e20), rot = cfg.vdbl("rotfreq", 1e20), selfk = cfg.vdbl("selfkerr", 0.0);
  auto t1 = cfg.vdbl("decay_time", 0.0), t2 = cfg.vdbl("dephase_time", 0.0);
  copy_last(trans, Q); copy_last(rot, Q); copy_last(selfk, Q); copy_last(t1, Q); copy_last(t2, Q);
  const int np = Q * (Q - 1) / 2;
  auto ck = cfg.vdbl("crosskerr", 0.0), jkl = cfg.vdbl("Jkl", 0.0);
  copy_last(ck, std::max(np, 1)); copy_last(jkl, std::max(np, 1));
  qd_system& s = P.sys;
  s.nosc = Q;
  s.lindblad_type = lt;
  for (int i = 0; i < Q; i++) {
    s.nlevels[i] = P.nlevels[i]; s.nessential[i] = P.ness[i];
    s.transfreq[i] = trans[i]; s.rotfreq[i] = rot[i]; s.selfkerr[i] = selfk[i];
    s.decay_time[i] = t1[i]; s.dephase_time[i] = t2[i];
    P.N *= P.nlevels[i];
    P.dim_ess *= P.ness[i];
  }
  P.transfreq = trans;
  for (int i = 0; i < np; i++) { s.crosskerr[i] = ck[i]; s.Jkl[i] = jkl[i]; }
  P.dim = P.lindblad ? P.N * P.N : P.N;
  P.tg.ntime = P.ntime;
  P.tg.dt = P.dt;
  const std::string ls = cfg.str("linearsolver_type", "gmres"), ts = cfg.str("timestepper", "IMR");
  P.sol.linsolve = ls == "gmres" ? QD_LINSOLVE_GMRES : ls == "neumann" ? QD_LINSOLVE_NEUMANN : -1;
  if (P.sol.linsolve < 0) die("Unknown linear solver type: " + ls);
  P.sol.stepper = ts == "IMR" ? QD_STEPPER_IMR : ts == "IMR4" ? QD_STEPPER_IMR4 : ts == "IMR8" ? QD_STEPPER_IMR8 : ts == "EE" ? QD_STEPPER_EE : -1;
  if (P.sol.stepper < 0) die("Unknow timestepping type: " + ts);
  P.sol.maxiter = cfg.integer("linearsolver_maxiter", 10);
  P.sol.abstol = 1e-10;
  P.sol.reltol = 1e-20;

  // ---- controls and their initialisation (src/main.cpp:222-277, src/oscillator.cpp:45-205)
  const bool bc = cfg.boolean("control_enforceBC", true);
  int seed = cfg.integer("rand_seed", -1);
  if (seed < 0) seed = (int)std::random_device{}();
  char tbuf[64];
  snprintf(tbuf, sizeof tbuf, "%f", total_time);
  std::string default_seg = std::string("spline, 10, 0.0, ") + tbuf, default_init = "constant, 0.0";
  for (int i = 0; i < Q; i++) {
    auto carrier = cfg.vdbl("carrier_frequency" + std::to_string(i), 0.0);
    strvec segs = cfg.vstr("control_segments" + std::to_string(i), default_seg);
    strvec inits = cfg.vstr("control_initialization" + std::to_string(i), default_init);
    strvec bnd = cfg.vstr("control_bounds" + std::to_string(i), "10000.0");
    P.ncar.push_back((int)carrier.size());
    for (double c : carrier) P.cars.push_back(c);
    struct Seg { int type, ns; double t0, t1; int skip; double par[3]; };
    auto npc_of = [](const Seg& g) { return g.type == QD_CTRL_STEP ? 1 : g.type == QD_CTRL_BSPLINEAMP ? g.ns + 1 : 2 * g.ns; };
    std::vector<Seg> my;
    size_t idx = 0;
    int skip = 0;
    auto window = [&](double& t0, double& t1s) {
      t0 = 0.0; t1s = total_time;
      if (segs.size() >= idx + 2) { t0 = atof(segs[idx].c_str()); t1s = atof(segs[idx + 1].c_str()); idx += 2; }
    };
    while (idx < segs.size()) {
      if (segs[idx] == "spline" || segs[idx] == "spline0") {
        const int type = segs[idx] == "spline" ? QD_CTRL_BSPLINE : QD_CTRL_BSPLINE0;
        idx++;
        if (idx >= segs.size()) die("Wrong setting for control segments: Number of splines not found.");
        const int ns = atoi(segs[idx++].c_str());
        double t0, t1s;
        window(t0, t1s);
        my.push_back({type, ns, t0, t1s, skip, {0.0, 0.0, 0.0}});
      } else if (segs[idx] == "step") {  // step, amp1, amp2, tramp [, tstart, tstop]: src/oscillator.cpp:50-70
        idx++;
        if (segs.size() <= idx + 2) die("Wrong setting for control segments: Step Amplitudes or tramp not found.");
        const double a1 = atof(segs[idx].c_str()), a2 = atof(segs[idx + 1].c_str()), tramp = atof(segs[idx + 2].c_str());
        idx += 3;
        double t0, t1s;
        window(t0, t1s);
        my.push_back({QD_CTRL_STEP, 1, t0, t1s, skip, {a1, a2, tramp}});
      } else if (segs[idx] == "spline_amplitude") {  // spline_amplitude, nsplines, scaling [, tstart, tstop]: :109-127
        idx++;
        if (idx + 1 >= segs.size()) die("Wrong setting for control segments: Number of splines not found.");
        const int ns = atoi(segs[idx].c_str());
        const double scaling = atof(segs[idx + 1].c_str());
        idx += 2;
        double t0, t1s;
        window(t0, t1s);
        my.push_back({QD_CTRL_BSPLINEAMP, ns, t0, t1s, skip, {scaling, 0.0, 0.0}});
      } else {
        idx++;
        continue;
      }
      skip += npc_of(my.back()) * (int)carrier.size();
    }
    std::vector<double> p;
    std::mt19937 rng;  // passed BY VALUE to every oscillator in the reference: each restarts the stream
    rng.seed(seed);
    size_t idini = 0;
    for (auto& sg : my) {  // src/oscillator.cpp:134-196
      if (inits.size() < idini + 2) { inits.push_back("constant"); inits.push_back(sg.type == QD_CTRL_STEP ? "1.0" : "0.0"); }
      double initval = atof(inits[idini + 1].c_str()) * 2.0 * M_PI;
      const int npc = npc_of(sg);
      const double phase = inits.size() > idini + 2 ? atof(inits[idini + 2].c_str()) : 0.0;  // spline_amplitude only (:159-162)
      if (inits[idini] == "constant") {
        if (sg.type == QD_CTRL_STEP) initval = std::min(1.0, std::max(0.0, initval));
        for (size_t f = 0; f < carrier.size(); f++) {
          p.insert(p.end(), npc, initval);
          if (sg.type == QD_CTRL_BSPLINEAMP) p.back() = phase;
        }
      } else if (inits[idini] == "random") {
        std::uniform_real_distribution<double> unit(0.0, 1.0);
        for (size_t f = 0; f < carrier.size(); f++) {
          for (int k = 0; k < npc; k++) {
            double val = initval * unit(rng);
            val = sg.type == QD_CTRL_STEP ? std::min(1.0, std::max(0.0, val)) : 2 * val - initval;
            p.push_back(val);
          }
          if (sg.type == QD_CTRL_BSPLINEAMP) p.back() = phase;
        }
      } else p.insert(p.end(), (size_t)npc * carrier.size(), 0.0);
      idini += 2;
    }
    if (!p.empty() && bc)
      for (auto& sg : my)
        for (size_t f = 0; f < carrier.size(); f++) {
          if (sg.type == QD_CTRL_BSPLINE) {
            for (int l = 0; l < sg.ns; l++)
              if (l <= 1 || l >= sg.ns - 2) p[sg.skip + f * sg.ns * 2 + l] = p[sg.skip + f * sg.ns * 2 + l + sg.ns] = 0.0;
          } else if (sg.type == QD_CTRL_BSPLINEAMP) {  // src/controlbasis.cpp:118-125
            for (int l = 0; l < sg.ns; l++)
              if (l <= 1 || l >= sg.ns - 2) p[sg.skip + f * (sg.ns + 1) + l] = 0.0;
          } else if (sg.type == QD_CTRL_BSPLINE0) {
            p[sg.skip + 2 * f * sg.ns] = p[sg.skip + 2 * f * sg.ns + sg.ns - 1] = 0.0;
            p[sg.skip + (2 * f + 1) * sg.ns] = p[sg.skip + (2 * f + 1) * sg.ns + sg.ns - 1] = 0.0;
          }
        }
    P.params0.insert(P.params0.end(), p.begin(), p.end());
    for (size_t iseg = 0; iseg < my.size(); iseg++) {  // bounds, src/optimproblem.cpp:137-163
      double bv = atof((iseg < bnd.size() ? bnd[iseg] : bnd.back()).c_str());
      bv = bv / (sqrt(2.0) * carrier.size()) * 2.0 * M_PI;
      const size_t nsp = (size_t)npc_of(my[iseg]) * carrier.size(), b0 = P.bounds.size();
      P.bounds.insert(P.bounds.end(), nsp, bv);
      if (my[0].type == QD_CTRL_BSPLINEAMP)  // no bound on the phase (the first segment decides, :152-159)
        for (size_t f = 0; f < carrier.size(); f++) {
          const size_t j = f * (my[0].ns + 1) + my[0].ns;
          if (j < nsp) P.bounds[b0 + j] = 1e10;
        }
      for (double v : my[iseg].par) P.seg_par.push_back(v);
      P.seg_osc.push_back(i); P.seg_type.push_back(my[iseg].type); P.seg_ns.push_back(my[iseg].ns);
      P.seg_t0.push_back(my[iseg].t0); P.seg_t1.push_back(my[iseg].t1);
    }
    default_seg.clear();
    default_init.clear();
    for (auto& t : segs) default_seg += t + ", ";
    for (auto& t : inits) default_init += t + ", ";
  }
  strvec init0 = cfg.vstr("control_initialization0", "constant, 0.0");
  if (!init0.empty() && init0[0] == "file") P.params0 = read_vector(P.cfgdir + init0[1], P.params0.size());
  strvec pp = cfg.vstr("apply_pipulse", "none");
  if (pp[0] != "none") {
    if (pp.size() % 4) die("Wrong pi-pulse configuration. Number of elements must be multiple of 4!");
    for (size_t k = 0; k < pp.size(); k += 4)
      for (int i = 0; i < Q; i++) {
        P.pi_osc.push_back(i);
        P.pi_t0.push_back(atof(pp[k + 1].c_str()));
        P.pi_t1.push_back(atof(pp[k + 2].c_str()));
        P.pi_amp.push_back(i == atoi(pp[k].c_str()) ? atof(pp[k + 3].c_str()) : 0.0);
      }
  }
  qd_controls& c = P.ctl;
  c.enforce_bc = bc;
  c.nseg_total = (int)P.seg_osc.size();
  c.seg_osc = P.seg_osc.data(); c.seg_type = P.seg_type.data(); c.seg_nsplines = P.seg_ns.data();
  c.seg_tstart = P.seg_t0.data(); c.seg_tstop = P.seg_t1.data();
  c.seg_param = P.seg_par.data();
  c.ncarrier = P.ncar.data(); c.carrier_freq = P.cars.data();
  c.npipulse = (int)P.pi_osc.size();
  c.pipulse_osc = P.pi_osc.data(); c.pipulse_tstart = P.pi_t0.data(); c.pipulse_tstop = P.pi_t1.data(); c.pipulse_amp = P.pi_amp.data();

  // ---- objective (src/main.cpp:89-128, src/optimproblem.cpp:61-131, src/optimtarget.cpp:22-316)
  qd_objective& o = P.obj;
  strvec ic = cfg.vstr("initialcondition", "none");
  static const std::map<std::string, int> ictypes = {{"file", QD_INIT_FROMFILE}, {"pure", QD_INIT_PURE}, {"ensemble", QD_INIT_ENSEMBLE},
      {"diagonal", QD_INIT_DIAGONAL}, {"basis", QD_INIT_BASIS}, {"3states", QD_INIT_THREESTATES}, {"Nplus1", QD_INIT_NPLUSONE},
      {"performance", QD_INIT_PERFORMANCE}};
  if (!ictypes.count(ic[0])) die("Wrong setting for initial condition.");
  o.initcond_type = ictypes.at(ic[0]);
  std::vector<int> ids;
  const size_t nel = P.lindblad ? (size_t)2 * P.dim_ess * P.dim_ess : (size_t)2 * P.dim_ess;
  if (ic[0] == "file") {
    if (ic.size() < 2) die("initialcondition = file needs a file name");
    P.init_data = read_vector(P.cfgdir + ic[1], nel);
    o.init_data = P.init_data.data();
  } else {
    for (size_t i = 1; i < ic.size(); i++) ids.push_back(atoi(ic[i].c_str()));
    if (ids.empty())
      for (int i = 0; i < Q; i++) ids.push_back(i);
  }
  o.n_init_ids = (int)std::min<size_t>(ids.size(), QD_MAX_OSC);
  for (int i = 0; i < o.n_init_ids; i++) o.init_ids[i] = ids[i];
  strvec tgt = cfg.vstr("optim_target", "pure");
  if (tgt[0] == "gate") {
    o.target_type = QD_TARGET_GATE;
    if (tgt.size() < 2) die("You want to optimize for a gate, but didn't specify which one.");
    gate_matrix(tgt, P.dim_ess, Q, P.cfgdir, P.gate_re, P.gate_im);
    o.gate_re = P.gate_re.data();
    o.gate_im = P.gate_im.data();
    auto grot = cfg.vdbl("gate_rot_freq", 1e20);
    copy_last(grot, Q);
    for (int i = 0; i < Q; i++) o.gate_rot_freq[i] = grot[0] < 1e20 ? grot[i] : 0.0;
  } else if (tgt[0] == "pure") {
    o.target_type = QD_TARGET_PURE;
    std::vector<int> lv;
    for (size_t i = 1; i < tgt.size(); i++) lv.push_back(atoi(tgt[i].c_str()));
    if (lv.empty()) lv.assign(Q, 0);
    copy_last(lv, Q);
    for (int i = 0; i < Q; i++) o.target_pure_levels[i] = lv[i];
  } else if (tgt[0] == "file") {
    o.target_type = QD_TARGET_FROMFILE;
    if (tgt.size() < 2) die("optim_target = file needs a file name");
    P.target_data = read_vector(P.cfgdir + tgt[1], nel);
    o.target_data = P.target_data.data();
  } else die("Unknown optimization target: " + tgt[0]);
  const std::string ob = cfg.str("optim_objective", "Jfrobenius");
  o.objective_type = ob == "Jfrobenius" ? QD_OBJ_JFROBENIUS : ob == "Jtrace" ? QD_OBJ_JTRACE : ob == "Jmeasure" ? QD_OBJ_JMEASURE : -1;
  if (o.objective_type < 0) die("Unknown objective function: " + ob);
  P.weights = cfg.vdbl("optim_weights", 1.0);
  o.nweights = (int)P.weights.size();
  o.weights = P.weights.data();
  o.gamma_tik = cfg.dbl("optim_regul", 1e-4);
  o.tik0 = cfg.has("optim_regul_tik0") ? cfg.boolean("optim_regul_tik0", false) : cfg.boolean("optim_regul_interpolate", false);
  o.alpha0 = o.tik0 ? P.params0.data() : nullptr;
  o.penalty.gamma_penalty = cfg.dbl("optim_penalty", 0.0);
  o.penalty.penalty_param = cfg.dbl("optim_penalty_param", 0.5);
  o.penalty.gamma_penalty_dpdm = cfg.dbl("optim_penalty_dpdm", 0.0);
  o.penalty.gamma_penalty_energy = cfg.dbl("optim_penalty_energy", 0.0);
  o.gamma_penalty_variation = cfg.dbl("optim_penalty_variation", 0.01);
}

static void write_trajectories(const Problem& P, Output& out, qd_handle* h, qd_optim* o) {
  const int nl = qd_optim_ninit_local(o);
  const size_t n2 = (size_t)2 * P.dim;
  std::vector<int> ids(nl);
  std::vector<double> tmp(n2);
  for (int i = 0; i < nl; i++) QDCHK(qd_optim_initial_state(o, i, tmp.data(), &ids[i]));
  struct Files { std::vector<FILE*> e, p; FILE *ec = nullptr, *pc = nullptr, *u = nullptr, *v = nullptr; };
  std::vector<Files> F(nl);
  char fn[512];
  for (int i = 0; i < nl; i++) {  // Output::openTrajectoryDataFiles (src/output.cpp:159-201)
    F[i].e.assign(P.Q, nullptr);
    F[i].p.assign(P.Q, nullptr);
    for (int k = 0; k < P.Q; k++) {
      if (out.wexp[k]) {
        snprintf(fn, sizeof fn, "%s/expected%d.iinit%04d.dat", out.datadir.c_str(), k, ids[i]);
        F[i].e[k] = fopen(fn, "w");
        fprintf(F[i].e[k], "#\"time\"      \"expected energy level\"\n");
      }
      if (out.wpop[k]) {
        snprintf(fn, sizeof fn, "%s/population%d.iinit%04d.dat", out.datadir.c_str(), k, ids[i]);
        F[i].p[k] = fopen(fn, "w");
        fprintf(F[i].p[k], "#\"time\"      \"diagonal of the density matrix\"\n");
      }
    }
    if (out.ecomp) {
      snprintf(fn, sizeof fn, "%s/expected_composite.iinit%04d.dat", out.datadir.c_str(), ids[i]);
      F[i].ec = fopen(fn, "w");
      fprintf(F[i].ec, "#\"time\"      \"expected energy level\"\n");
    }
    if (out.pcomp) {
      snprintf(fn, sizeof fn, "%s/population_composite.iinit%04d.dat", out.datadir.c_str(), ids[i]);
      F[i].pc = fopen(fn, "w");
      fprintf(F[i].pc, "#\"time\"      \"population\"\n");
    }
    if (out.full) {
      snprintf(fn, sizeof fn, "%s/rho_Re.iinit%04d.dat", out.datadir.c_str(), ids[i]);
      F[i].u = fopen(fn, "w");
      snprintf(fn, sizeof fn, "%s/rho_Im.iinit%04d.dat", out.datadir.c_str(), ids[i]);
      F[i].v = fopen(fn, "w");
    }
  }
  // observables are reduced on the device from the stored trajectory (one launch, one download); full states only travel
  // to the host when `fullstate` output is requested
  bool any_e = false, any_p = false;
  for (int k = 0; k < P.Q; k++) { any_e = any_e || out.wexp[k]; any_p = any_p || out.wpop[k]; }
  int nlev = 0;
  std::vector<int> lev0(P.Q, 0);
  for (int k = 0; k < P.Q; k++) { lev0[k] = nlev; nlev += P.nlevels[k]; }
  const int nout = P.ntime / out.output_frequency + 1;
  std::vector<double> oe(any_e ? (size_t)nout * nl * P.Q : 0), op(any_p ? (size_t)nout * nl * nlev : 0), oec(out.ecomp ? (size_t)nout * nl : 0),
      opc(out.pcomp ? (size_t)nout * nl * P.N : 0);
  if (any_e || any_p || out.ecomp || out.pcomp)
    QDCHK(qd_get_observables(h, out.output_frequency, any_e ? oe.data() : nullptr, any_p ? op.data() : nullptr, out.ecomp ? oec.data() : nullptr,
                             out.pcomp ? opc.data() : nullptr));
  std::vector<double> states(out.full ? (size_t)nl * n2 : 0);
  for (int n = 0; n <= P.ntime; n++) {  // Output::writeTrajectoryDataFiles (src/output.cpp:203-273)
    if (n % out.output_frequency) continue;
    if (out.full) QDCHK(qd_get_state(h, n, states.data()));
    const double time = n * P.dt;
    const size_t io = (size_t)(n / out.output_frequency) * nl;
    for (int i = 0; i < nl; i++) {
      const double* x = out.full ? states.data() + (size_t)i * n2 : nullptr;
      for (int k = 0; k < P.Q; k++) {
        if (F[i].e[k]) fprintf(F[i].e[k], "%.8f %1.14e\n", time, oe[(io + i) * P.Q + k]);
        if (F[i].p[k]) {
          fprintf(F[i].p[k], "%.8f ", time);
          for (int l = 0; l < P.nlevels[k]; l++) fprintf(F[i].p[k], " %1.14e", op[(io + i) * nlev + lev0[k] + l]);
          fprintf(F[i].p[k], "\n");
        }
      }
      if (F[i].ec) fprintf(F[i].ec, "%.8f %1.14e\n", time, oec[io + i]);
      if (F[i].pc) {
        fprintf(F[i].pc, "%.8f  ", time);
        for (int r = 0; r < P.N; r++) fprintf(F[i].pc, "%1.14e  ", opc[(io + i) * P.N + r]);
        fprintf(F[i].pc, "\n");
      }
      if (F[i].u) {
        fprintf(F[i].u, "%.8f  ", time);
        fprintf(F[i].v, "%.8f  ", time);
        for (int r = 0; r < P.dim; r++) {
          fprintf(F[i].u, "%1.10e  ", x[r]);
          fprintf(F[i].v, "%1.10e  ", x[r + P.dim]);
        }
        fprintf(F[i].u, "\n");
        fprintf(F[i].v, "\n");
      }
    }
  }
  for (auto& f : F) {
    for (FILE* p : f.e) if (p) fclose(p);
    for (FILE* p : f.p) if (p) fclose(p);
    for (FILE* p : {f.ec, f.pc, f.u, f.v}) if (p) fclose(p);
  }
}

// Output::writeControls (src/output.cpp:111-156)
static void write_controls(const Problem& P, Output& out, qd_handle* h, const std::vector<double>& x) {
  out.write_vec("params.dat", x);
  QDCHK(qd_set_params(h, x.data(), (int)x.size()));
  std::vector<double> times;
  for (int i = 0; i <= P.ntime; i += out.output_frequency) times.push_back(i * P.dt);
  std::vector<double> pq(times.size() * P.Q * 2);
  QDCHK(qd_eval_controls(h, times.data(), (int)times.size(), pq.data()));
  for (int k = 0; k < P.Q; k++) {
    FILE* f = fopen((out.datadir + "/control" + std::to_string(k) + ".dat").c_str(), "w");
    fprintf(f, "#\"time\"         \"p(t) (rotating)\"          \"q(t) (rotating)\"         \"f(t) (labframe)\"\n");
    const double w = 2.0 * M_PI * P.transfreq[k];
    for (size_t i = 0; i < times.size(); i++) {
      const double p = pq[(i * P.Q + k) * 2], q = pq[(i * P.Q + k) * 2 + 1], t = times[i];
      const double lab = 2.0 * (p * cos(w * t) - q * sin(w * t));  // Oscillator::evalControl_Labframe (src/oscillator.cpp:383-428)
      fprintf(f, "% 1.8f   % 1.14e   % 1.14e   % 1.14e \n", t, p / (2.0 * M_PI), q / (2.0 * M_PI), lab / (2.0 * M_PI));
    }
    fclose(f);
  }
}

static double norm2(const std::vector<double>& v) {
  double s = 0.0;
  for (double x : v) s += x * x;
  return sqrt(s);
}

// One rank's view of evalF / evalGradF: single GPU, or this rank's shard behind an RCCL communicator
// (QD_NRANKS > 1: one process per GPU, src/main.cpp:133-177 without MPI).
struct Evaluator {
  qd_optim* o = nullptr;
  qd_comm* c = nullptr;
  int rank = 0, nranks = 1;
  void F(const double* x, qd_objective_value* v) const {
    if (c) QDCHK(qd_optim_evalF_dist(o, c, x, v, nullptr));
    else QDCHK(qd_optim_evalF(o, x, v));
  }
  void G(const double* x, qd_objective_value* v, double* g) const {
    if (c) QDCHK(qd_optim_evalGradF_dist(o, c, x, v, g, nullptr));
    else QDCHK(qd_optim_evalGradF(o, x, v, g));
  }
};

// Bounded quasi-Newton standing in for TAO BQNLS (src/optimproblem.cpp:178-189): L-BFGS on the free variables (active
// set = variables at a bound whose gradient points outwards), strong-Wolfe line search along the projected path
// (bracketing + cubic zoom, c1 = 1e-4, c2 = 0.9 as More-Thuente's defaults in TAO); monitor, stopping rules and
// the final trajectory-writing evaluation as TaoMonitor (src/optimproblem.cpp:586-660).  Iterates differ from
// TAO's by construction (third-party internals); acceptance = reaching the same thresholds in a comparable number
// of iterations (tests/test_driver_regression.py).
static void optimize(const Problem& P, Output& out, qd_handle* h, const Evaluator& ev, std::vector<double>& x, bool quiet) {
  const Config& cfg = P.cfg;
  const int n = (int)x.size(), maxiter = cfg.integer("optim_maxiter", 200), mem = 10;
  const double gatol = cfg.dbl("optim_atol", 1e-8), grtol = cfg.dbl("optim_rtol", 1e-4), fatol = cfg.dbl("optim_ftol", 1e-8),
               inftol = cfg.dbl("optim_inftol", 1e-5);
  const bool root = ev.rank == 0;
  auto project = [&](std::vector<double>& v) { for (int i = 0; i < n; i++) v[i] = std::min(P.bounds[i], std::max(-P.bounds[i], v[i])); };
  auto pgnorm = [&](const std::vector<double>& xx, const std::vector<double>& g) {
    double s = 0.0;
    for (int i = 0; i < n; i++) {
      const double step = std::min(P.bounds[i], std::max(-P.bounds[i], xx[i] - g[i])) - xx[i];
      s += step * step;
    }
    return sqrt(s);
  };
  auto dot = [&](const std::vector<double>& a, const std::vector<double>& b) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; };
  project(x);
  std::vector<double> g(n), xn(n), gn(n), d(n), gm(n);
  std::vector<std::vector<double>> S, Y;
  qd_objective_value v{}, vn{};
  ev.G(x.data(), &v, g.data());
  const double g0 = pgnorm(x, g);
  double gnorm = g0, step = 1.0;
  int nfev = 1;
  bool retry = false;  // the line search of this iteration is repeated from steepest descent: no second history row / controls file
  for (int it = 0;; it++) {
    const char* why = nullptr;
    if (1.0 - v.fidelity <= inftol) why = "Optimization converged with small infidelity.";
    else if (v.cost <= fatol) why = "Optimization converged with small final time cost.";
    else if (it >= maxiter) why = "Optimization stopped at maximum number of iterations.";
    else if (gnorm < gatol) why = "Optimization converged with small gradient norm.";
    else if (gnorm / g0 < grtol) why = "Optimization converged with small relative gradient norm.";
    // history row every optim_monitor_frequency iterations and on the last one (src/optimproblem.cpp:634-645)
    if (root && !retry && (it % out.optim_monitor_freq == 0 || why)) {
      out.optim_row(it, v, gnorm, step);
      if (!quiet) printf("%d  %1.14e + %1.14e + %1.14e + %1.14e + %1.14e + %1.14e  Fidelity = %1.14e  ||Grad|| = %1.14e\n", it, v.cost, v.regul,
                         v.penalty, v.penalty_dpdm, v.penalty_energy, v.penalty_variation, v.fidelity, gnorm);
    }
    if (why) {
      if (root && !quiet) printf("%s (%d function/gradient evaluations)\n", why, nfev);
      break;
    }
    if (root && !retry && it % out.optim_monitor_freq == 0) write_controls(P, out, h, x);
    retry = false;
    // active set and masked gradient
    for (int i = 0; i < n; i++) {
      const bool act = (x[i] <= -P.bounds[i] && g[i] > 0.0) || (x[i] >= P.bounds[i] && g[i] < 0.0);
      gm[i] = act ? 0.0 : g[i];
    }
    // two-loop recursion on the masked gradient
    d = gm;
    std::vector<double> al(S.size());
    for (int k = (int)S.size() - 1; k >= 0; k--) {
      const double sy = dot(S[k], Y[k]);
      al[k] = dot(S[k], d) / sy;
      for (int i = 0; i < n; i++) d[i] -= al[k] * Y[k][i];
    }
    if (!S.empty()) {
      const double sc = dot(S.back(), Y.back()) / dot(Y.back(), Y.back());
      for (int i = 0; i < n; i++) d[i] *= sc;
    }
    for (size_t k = 0; k < S.size(); k++) {
      const double be = dot(Y[k], d) / dot(S[k], Y[k]);
      for (int i = 0; i < n; i++) d[i] += S[k][i] * (al[k] - be);
    }
    for (int i = 0; i < n; i++)
      if (gm[i] == 0.0 && g[i] != 0.0) d[i] = 0.0;  // stay on the active bounds
    double dphi0 = -dot(g, d);
    if (!(dphi0 < 0.0)) {  // not a descent direction: restart from steepest descent
      S.clear();
      Y.clear();
      d = gm;
      dphi0 = -dot(g, d);
      if (!(dphi0 < 0.0)) { if (root && !quiet) printf("Projected gradient vanished.\n"); break; }
    }
    // strong-Wolfe line search on phi(a) = f(P(x - a d))
    const double c1 = 1e-4, c2 = 0.9, phi0 = v.objective;
    auto eval_at = [&](double a, double& phi, double& dphi) {
      for (int i = 0; i < n; i++) xn[i] = x[i] - a * d[i];
      project(xn);
      ev.G(xn.data(), &vn, gn.data());
      nfev++;
      phi = vn.objective;
      dphi = 0.0;
      for (int i = 0; i < n; i++) {
        const double raw = x[i] - a * d[i];
        if (raw > -P.bounds[i] && raw < P.bounds[i]) dphi -= gn[i] * d[i];  // clamped components do not move
      }
    };
    double a = S.empty() ? std::min(1.0, 1.0 / norm2(d)) : 1.0;
    double a_lo = 0.0, phi_lo = phi0, dphi_lo = dphi0, a_hi = 0.0, phi_hi = 0.0, dphi_hi = 0.0;
    double phi = 0.0, dphi = 0.0, a_prev = 0.0, phi_prev = phi0, dphi_prev = dphi0;
    bool ok = false, bracket = false;
    for (int ls = 0; ls < 12 && !ok; ls++) {
      eval_at(a, phi, dphi);
      if (!(phi <= phi0 + c1 * a * dphi0) || (ls > 0 && phi >= phi_prev)) {
        a_lo = a_prev; phi_lo = phi_prev; dphi_lo = dphi_prev; a_hi = a; phi_hi = phi; dphi_hi = dphi; bracket = true;
        break;
      }
      if (fabs(dphi) <= -c2 * dphi0) { ok = true; break; }
      if (dphi >= 0.0) {
        a_lo = a; phi_lo = phi; dphi_lo = dphi; a_hi = a_prev; phi_hi = phi_prev; dphi_hi = dphi_prev; bracket = true;
        break;
      }
      a_prev = a; phi_prev = phi; dphi_prev = dphi;
      a *= 2.0;
    }
    if (!ok && bracket) {
      for (int z = 0; z < 15 && !ok; z++) {
        // minimiser of the cubic through (a_lo, phi_lo, dphi_lo), (a_hi, phi_hi, dphi_hi); bisection when it leaves the interval
        const double dd1 = dphi_lo + dphi_hi - 3.0 * (phi_lo - phi_hi) / (a_lo - a_hi);
        const double rad = dd1 * dd1 - dphi_lo * dphi_hi;
        double at = 0.5 * (a_lo + a_hi);
        if (rad >= 0.0) {
          const double dd2 = (a_hi > a_lo ? 1.0 : -1.0) * sqrt(rad);
          const double cand = a_hi - (a_hi - a_lo) * (dphi_hi + dd2 - dd1) / (dphi_hi - dphi_lo + 2.0 * dd2);
          const double lo = std::min(a_lo, a_hi), hi = std::max(a_lo, a_hi);
          if (cand > lo + 0.05 * (hi - lo) && cand < hi - 0.05 * (hi - lo)) at = cand;
        }
        a = at;
        eval_at(a, phi, dphi);
        if (!(phi <= phi0 + c1 * a * dphi0) || phi >= phi_lo) {
          a_hi = a; phi_hi = phi; dphi_hi = dphi;
        } else {
          if (fabs(dphi) <= -c2 * dphi0) { ok = true; break; }
          if (dphi * (a_hi - a_lo) >= 0.0) { a_hi = a_lo; phi_hi = phi_lo; dphi_hi = dphi_lo; }
          a_lo = a; phi_lo = phi; dphi_lo = dphi;
        }
      }
      if (!ok && a_lo > 0.0) {  // best point with sufficient decrease
        a = a_lo;
        eval_at(a, phi, dphi);
        ok = phi <= phi0 + c1 * a * dphi0;
      }
    }
    if (!ok) {
      if (S.empty()) { if (root && !quiet) printf("Line search failed.\n"); break; }
      S.clear();
      Y.clear();
      it--;  // retry this iteration from steepest descent
      retry = true;
      continue;
    }
    step = a;
    std::vector<double> sk(n), yk(n);
    for (int i = 0; i < n; i++) { sk[i] = xn[i] - x[i]; yk[i] = gn[i] - g[i]; }
    const double sy = dot(sk, yk);
    if (sy > 1e-12 * norm2(sk) * norm2(yk)) {
      S.push_back(sk);
      Y.push_back(yk);
      if ((int)S.size() > mem) { S.erase(S.begin()); Y.erase(Y.begin()); }
    }
    x = xn;
    g = gn;
    v = vn;
    gnorm = pgnorm(x, g);
  }
  // last iteration: controls + one more forward evaluation that writes the trajectory files (src/optimproblem.cpp:647-656)
  if (root) write_controls(P, out, h, x);
  double sums[QD_NSUMS];
  QDCHK(qd_optim_forward_local(ev.o, x.data(), 1, sums));
  write_trajectories(P, out, h, ev.o);
}

// total number of initial conditions, from the config alone (src/main.cpp:89-128) - needed before any device object exists, to decide how
// many of the launched ranks can work
static int count_initial_conditions(const Problem& P) {
  strvec ic = P.cfg.vstr("initialcondition", "none");
  if (ic.empty()) return 1;
  if (ic[0] == "3states") return 3;
  if (ic[0] == "Nplus1") return P.N + 1;
  if (ic[0] == "diagonal" || ic[0] == "basis") {
    if (ic.size() < 2)
      for (int j = 0; j < P.Q; j++) ic.push_back(std::to_string(j));
    long long n = 1;
    for (size_t i = 1; i < ic.size(); i++) {
      const int k = atoi(ic[i].c_str());
      if (k >= 0 && k < P.Q) n *= P.sys.nessential[k];
    }
    if (ic[0] == "basis" && P.lindblad) n *= n;
    return (int)n;
  }
  return 1;  // file, pure, performance, ensemble
}

// Rank and size of this process as the launcher describes them.  No MPI is linked: the launchers export what is needed -
// QD_RANK / QD_NRANKS (any launcher, takes precedence), Open MPI (OMPI_COMM_WORLD_*), MPICH / Hydra and Intel MPI (PMI_RANK, PMI_SIZE,
// MPI_LOCALRANKID), PMIx (PMIX_RANK), MVAPICH (MV2_COMM_WORLD_*), Slurm's srun (SLURM_PROCID, SLURM_NTASKS, SLURM_LOCALID).
struct LaunchEnv {
  int rank = 0, size = 1, local_rank = -1, local_size = -1;
};
static LaunchEnv launch_env() {
  auto geti = [](const char* k, int* v) {
    const char* e = getenv(k);
    if (!e || !*e) return false;
    *v = atoi(e);
    return true;
  };
  LaunchEnv le;
  static const char* const pairs[][2] = {{"QD_RANK", "QD_NRANKS"}, {"OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE"}, {"PMI_RANK", "PMI_SIZE"},
                                         {"MV2_COMM_WORLD_RANK", "MV2_COMM_WORLD_SIZE"}, {"PMIX_RANK", "SLURM_NTASKS"}, {"SLURM_PROCID", "SLURM_NTASKS"}};
  for (const auto& p : pairs) {
    int r, n;
    if (geti(p[0], &r) && geti(p[1], &n)) {
      le.rank = r;
      le.size = n;
      break;
    }
  }
  static const char* const lr[] = {"QD_LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "MV2_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"};
  for (const char* k : lr)
    if (geti(k, &le.local_rank)) break;
  static const char* const ls[] = {"QD_LOCAL_NRANKS", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "MV2_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE"};
  for (const char* k : ls)
    if (geti(k, &le.local_size)) break;
  return le;
}

int main(int argc, char** argv) {
  if (argc < 2 || std::string(argv[1]) == "--help") {
    printf("\nQUANDARY (MI355X path) - Optimal control for quantum systems\n\nUSAGE:\n  quandary <config_file> [--quiet]\n  quandary --version\n\n");
    return 0;
  }
  if (std::string(argv[1]) == "--version") {
    printf("%s\n", qd_version());
    return 0;
  }
  bool quiet = false;
  for (int i = 2; i < argc; i++)
    if (std::string(argv[i]) == "--quiet") quiet = true;  // --petsc-options is accepted and ignored (no PETSc here)
  Problem P;
  const std::string cfgfile = argv[1];
  const size_t slash = cfgfile.find_last_of('/');
  P.cfgdir = "";  // file names inside the config are relative to the working directory, as in the reference
  (void)slash;
  P.cfg.read(cfgfile);
  build(P);
  // multi-GPU: one process per GPU.  The reference's front end starts `mpirun -np <ncores> quandary config.cfg --quiet` with ncores a
  // divisor of the number of initial conditions (quandary.py:506-519, :1431-1450); this executable links no MPI and takes rank and size
  // from the launcher's environment (launch_env).  The ranks shard the initial conditions over the GPUs of the node:
  //   default                    the first `nactive` ranks work, one per GPU over RCCL (src/main.cpp:133-177 with np_init = nactive):
  //                              nactive = the largest divisor of ninit that is <= min(ranks, GPUs of the node) - the reference takes
  //                              np_init = min(ninit, size) and gives the rest to PETSc (src/main.cpp:140-153); here the surplus ranks have
  //                              nothing to do and exit 0 at once, so that a CPU-sized core count from quandary.py (`mpirun -np 8` for 4 initial
  //                              conditions, or 8 ranks on a one-GPU box) neither fails the run nor puts several processes on one GPU
  //   QD_SHARE_GPUS = 1          every rank works, rank r on GPU r mod ndev, reductions through the shared-memory backend
  //   several nodes              one rank per GPU is required (local ranks <= local GPUs); RCCL; ninit must divide by the rank count
  Evaluator ev;
  const LaunchEnv le = launch_env();
  ev.rank = le.rank;
  ev.nranks = le.size;
  if (ev.nranks < 1 || ev.rank < 0 || ev.rank >= ev.nranks) die("rank / size from the launcher's environment out of range");
  int device = P.cfg.integer("device", 0);
  if (ev.nranks > 1) {
    const int ndev = qd_device_count();
    if (ndev < 1) die("no HIP device visible");
    const bool share = getenv("QD_SHARE_GPUS") && atoi(getenv("QD_SHARE_GPUS")) != 0;
    const bool multinode = le.local_size > 0 && le.local_size < ev.nranks;
    const int ninit_global = count_initial_conditions(P);
    if (multinode && le.local_size > ndev) die("more ranks per node than GPUs on a multi-node launch: start one rank per GPU");
    if (!multinode && !share && (ev.nranks > ndev || ninit_global % ev.nranks != 0)) {
      int nactive = 1;
      for (int d = std::min(std::min(ndev, ev.nranks), ninit_global); d >= 1; d--)
        if (ninit_global % d == 0) {
          nactive = d;
          break;
        }
      if (ev.rank >= nactive) return 0;  // surplus rank: nothing to do, nothing to write
      if (ev.rank == 0 && !quiet)
        printf("%d ranks were started for %d initial conditions on %d GPU(s): %d rank(s) work, the others exit.\n", ev.nranks, ninit_global, ndev, nactive);
      ev.nranks = nactive;
    }
    if (ninit_global % ev.nranks != 0) die("the number of initial conditions must be a multiple of the number of working ranks (src/main.cpp:150-153)");
    device = getenv("QD_DEVICE") ? atoi(getenv("QD_DEVICE")) : (le.local_rank >= 0 ? le.local_rank : ev.rank) % ndev;
    // what the library needs to know about the launch (qd_comm_create_from_file: backend by ranks per node; qd_col.hip: scheduler time
    // limit by processes per device)
    const int local = multinode ? le.local_size : ev.nranks;
    setenv("QD_LOCAL_SIZE", std::to_string(local).c_str(), 1);
    if (local > ndev) setenv("QD_DEVICE_SHARERS", std::to_string((local + ndev - 1) / ndev).c_str(), 0);
  }
  if (ev.rank != 0) quiet = true;
  Output out;
  out.init(P.cfg, P.Q, ev.rank == 0);
  qd_handle* h = nullptr;
  QDCHK(qd_create(&P.sys, &P.ctl, &P.tg, &P.sol, device, &h));
  {  // extension of this build: precision = f64 (default, like the reference) | f32mixed
    const std::string prec = P.cfg.str("precision", "f64");
    if (prec == "f32mixed") QDCHK(qd_set_precision(h, QD_PRECISION_F32MIXED));
    else if (prec != "f64") die("precision must be f64 or f32mixed");
  }
  if (qd_ndesign(h) != (int)P.params0.size()) die("internal: parameter count mismatch");
  {  // user-supplied Hamiltonians (src/main.cpp:309-316, src/hamiltonianfilereader.cpp)
    const std::string fsys = P.cfg.str("hamiltonian_file_Hsys", "none"), fc = P.cfg.str("hamiltonian_file_Hc", "none");
    if (fsys != "none" || fc != "none") {
      const int N = qd_dim_rho(h);
      const size_t nn = (size_t)N * N;
      std::vector<double> sre(nn, 0.0), sim(nn, 0.0), cre((size_t)P.Q * nn, 0.0), cim((size_t)P.Q * nn, 0.0);
      auto each_line = [&](const std::string& fn, int nfields, const std::function<void(const std::vector<double>&)>& fnc) {
        std::ifstream f(P.cfgdir + fn);
        if (!f.is_open()) die("Could not open '" + fn + "'");
        std::string line;
        while (std::getline(f, line)) {
          if (line.empty() || line[0] == '#') continue;
          std::istringstream iss(line);
          std::vector<double> v;
          double x;
          while ((int)v.size() < nfields && (iss >> x)) v.push_back(x);
          if ((int)v.size() == nfields) fnc(v);
        }
      };
      if (fsys != "none")
        each_line(fsys, 4, [&](const std::vector<double>& v) {  // row col real imag
          const size_t e = (size_t)v[0] * N + (size_t)v[1];
          if (v[0] < 0 || v[0] >= N || v[1] < 0 || v[1] >= N) die("hamiltonian_file_Hsys: index out of range");
          sre[e] = v[2];
          sim[e] = v[3];
        });
      if (fc != "none")
        each_line(fc, 5, [&](const std::vector<double>& v) {  // oscillator row col real imag
          if (v[0] < 0 || v[0] >= P.Q || v[1] < 0 || v[1] >= N || v[2] < 0 || v[2] >= N) die("hamiltonian_file_Hc: index out of range");
          const size_t e = (size_t)v[0] * nn + (size_t)v[1] * N + (size_t)v[2];
          cre[e] += v[3];
          cim[e] += v[4];
        });
      QDCHK(qd_set_hamiltonian(h, sre.data(), sim.data(), fc != "none" ? cre.data() : nullptr, fc != "none" ? cim.data() : nullptr));
      if (!quiet) printf("# Hamiltonian model read from files.\n");
    }
  }
  qd_optim* o = nullptr;
  QDCHK(qd_optim_create(h, &P.obj, ev.rank, ev.nranks, &o));
  ev.o = o;
  const std::string idfile = getenv("QD_COMM_FILE") ? getenv("QD_COMM_FILE") : out.datadir + "/.qd_comm_id";
  if (ev.nranks > 1 || getenv("QD_FORCE_COMM")) {  // (QD_FORCE_COMM: one-rank communicator, exercises the RCCL path on a one-GPU box)
    QDCHK(qd_comm_create_from_file(idfile.c_str(), ev.rank, ev.nranks, device, 120.0, &ev.c));
    if (!quiet) {
      if (qd_comm_backend(ev.c) == 1)
        printf("Shared-memory communicator over %d ranks (GPU %d shared: rank r on GPU r mod ndev); %d initial conditions per rank.\n", ev.nranks, device,
               qd_optim_ninit_local(o));
      else printf("RCCL communicator over %d ranks (one GPU each); %d initial conditions per rank.\n", ev.nranks, qd_optim_ninit_local(o));
    }
  }
  if (!quiet) {
    printf("Number of initial conditions: %d\n", qd_optim_ninit(o));
    printf("State dimension (complex): %d\nTime: [0:%g], N=%d, dt=%g\nNumber of control parameters: %d\n", qd_dim(h), P.ntime * P.dt, P.ntime, P.dt,
           qd_ndesign(h));
  }
  const std::string runtype = P.cfg.str("runtype", "simulation");
  std::vector<double> x = P.params0, grad(x.size(), 0.0);
  // config_log.dat (src/main.cpp:382-393): every parameter the run asked for with the value it used, a config file in its own right.
  // Written before the run like the reference's and once more at the end (the optimiser reads its parameters when it starts).
  auto write_config_log = [&](bool announce) {
    if (ev.rank != 0) return;
    const std::string fn = out.datadir + "/config_log.dat";
    std::ofstream lf(fn);
    if (!lf.is_open()) {
      fprintf(stderr, "Unable to open %s\n", fn.c_str());
      return;
    }
    lf << P.cfg.log_text();
    if (announce && !quiet) printf("File written: %s\n", fn.c_str());
  };
  write_config_log(true);
  const auto t0 = std::chrono::steady_clock::now();
  qd_objective_value v{};
  double gnorm = 0.0;
  const bool root = ev.rank == 0;
  if (runtype == "simulation") {
    if (root) write_controls(P, out, h, x);
    double sums[QD_NSUMS];
    QDCHK(qd_optim_forward_local(o, x.data(), 1, sums));
    if (ev.c) QDCHK(qd_comm_allreduce(ev.c, sums, QD_NSUMS, 0));
    QDCHK(qd_optim_finalize(o, x.data(), sums, &v));
    write_trajectories(P, out, h, o);
    if (!quiet) printf("\nTotal objective = %1.14e, \n", v.objective);
  } else if (runtype == "gradient") {
    if (root) write_controls(P, out, h, x);
    if (ev.c) {  // trajectory files of this rank's initial conditions, then the distributed gradient
      double sums[QD_NSUMS];
      QDCHK(qd_optim_forward_local(o, x.data(), 1, sums));
      write_trajectories(P, out, h, o);
      ev.G(x.data(), &v, grad.data());
    } else {
      double sums[QD_NSUMS];
      QDCHK(qd_optim_forward_local(o, x.data(), 1, sums));
      QDCHK(qd_optim_finalize(o, x.data(), sums, &v));
      write_trajectories(P, out, h, o);
      QDCHK(qd_optim_adjoint_local(o, x.data(), sums, grad.data()));
    }
    gnorm = norm2(grad);
    if (!quiet) printf("\nGradient norm: %1.14e\n", gnorm);
    out.write_vec("grad.dat", grad);
  } else if (runtype == "optimization") {
    optimize(P, out, h, ev, x, quiet);
  } else if (runtype == "evalcontrols") {
    if (root) write_controls(P, out, h, x);
  } else {
    printf("\n\n WARNING: Unknown runtype: %s.\n\n", runtype.c_str());
  }
  if (runtype != "optimization") out.optim_row(0, v, gnorm, 0.0);
  const double used = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!quiet) printf("\n Used Time:        %.2f seconds\n Processors used:  %d (MI355X)\n\n", used, ev.nranks);
  if (root) {
    FILE* tf = fopen((out.datadir + "/timing.dat").c_str(), "w");  // src/main.cpp:482-487
    fprintf(tf, "%d  %1.8e\n", ev.nranks, used);
    fclose(tf);
  }
  fclose(out.optimfile);
  write_config_log(false);
  if (ev.c) {
    (void)qd_comm_barrier(ev.c);
    qd_comm_destroy(ev.c);
    if (root) remove(idfile.c_str());  // a stale id must not be picked up by the next run
  }
  qd_optim_destroy(o);
  qd_destroy(h);
  if (!quiet) printf("Output directory: %s\n", out.datadir.c_str());
  return 0;
}
