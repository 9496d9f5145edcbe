/*
 * csim.h -- C-ABI of the MI355X batched MNA solve engine (libcsim.so).
 *
 * extern "C", plain pointers and sizes, no C++/torch types, no exceptions
 * across the boundary: every call returns 0 (CSIM_OK) or a negative error and
 * csim_last_error() describes it.  Per-instance trouble (non-convergence,
 * non-finite solve, tiny pivot) never fails a call: it is reported in the
 * per-instance status words (CSIM_ST_* in csim_ir.h), so one bad instance
 * cannot kill a batch -- this replaces the reference's stderr WARNINGs and
 * its std::runtime_error (src/tanalisis.cpp:360-376, src/dcanalysis.cpp:135-158).
 *
 * There is NO CPU backend behind this ABI.  csim_engine_create() fails with
 * CSIM_ERR_NO_DEVICE when no HIP device is usable; nothing falls back to host
 * arithmetic.  (The CPU restatement lives in oracle/ and is test-only.)
 *
 * Reference interface each group replaces (ZyuRao/CircuitSimulator):
 *   csim_netlist_*        parseNetlist()                 include/parser.hpp:67-75
 *                         Circuit::assignEquationIndices src/circuit.cpp:42-61  (src/main.cpp:29,34)
 *   csim_dc_batch*        computeDcOperatingPoint()      include/tanalisis.hpp:9
 *                         dcSolve/dcSolveLU              include/dcanalysis.hpp:8-14, src/dcanalysis.cpp:242-262
 *   csim_tran_batch*      runTransientAnalysisBackwardEuler  include/tanalisis.hpp:15-17, src/tanalisis.cpp:83-424
 *   csim_lu_solve_batch   Solver::solveLinearSystemLU / luDecompose  include/solver.hpp:30-131
 *   (stamping)            Element::stamp virtuals        include/element.hpp:28-31, src/element.cpp:9-307
 *                         -- no entry point of their own: the stamps run inside the DC/TRAN kernels.
 *
 * Data layout.  Every per-instance table is "slot-major": element [i][b] at
 * i*B + b, b = instance.  That is the coalesced layout for the
 * lane-per-instance kernels and costs the wave-per-instance kernels one
 * strided read per launch.  The *_dev entry points take DEVICE pointers in
 * that layout and enqueue on the given HIP stream; the host-pointer entry
 * points take the instance-major tables proposed in SURVEY.md 8(b) ([B][P],
 * [B][N]) and do the copies and transposes.
 *
 * Arithmetic.  The general kernels, the "faithful" generated kernels (transient and DC) and the kernels for
 * linear circuits perform the reference's floating-point operations in the reference's order (the device's
 * sin() may differ from glibc's in the last bit).  The FAST generated transient kernels (family "scheduled":
 * one, four and sixteen lanes per instance) deviate deliberately, inside the 1e-9 bar, NR counts equal: FMA
 * contraction; one refined reciprocal per pivot instead of a division per multiplier; four-/sixteen-lane kernels:
 * matrix assembled as (step-constant part) + (MOSFET part), back substitution in descending column order, the
 * recorded pivot accepted where a LATER row exceeds it by less than 8 ulp (the reference would swap: the two
 * pivots then agree to 15 digits; include/solver.hpp:48-56), the update norm summed across lanes; convergence
 * decided on the squared norm.  A convergence decision within 2e-8 (relative) of its threshold is re-done by
 * the faithful kernel and rolled back if it falls the other way; steps that do not contract at the damping
 * rate, and factorisations no recorded pivot sequence fits, are handed to the faithful / general kernels.
 * What remains: a step whose update norm lands within ~1e-9 of the tolerance is decided by the last bits of the
 * whole trajectory; measured on dbmixer.sp, a fast family takes one NR pass more or less than the faithful
 * family about once per 1e10 step decisions.  csim_engine_set_kernel(eng, 3) selects the
 * faithful family outright.
 *
 * Streams.  With the default option hybrid_sync = 1 a *_dev call that runs
 * generated ("scheduled") kernels WAITS on its stream once per stage of the
 * hand-over ladder to read two flag words -- usually once per call -- and
 * returns as soon as no instance is left unfinished.  With hybrid_sync = 0
 * such a call only enqueues (a fixed sequence of launches, each returning at
 * once when it finds nothing to do) and never waits: use that to overlap
 * streams (copies or collectives of one launch's results under the next launch).
 * Graph capture is NOT supported (step_first is a kernel argument, and a replayed
 * capture did not reproduce a direct call reliably: tools/dev/graph_capture_probe.py).
 * Calls on engines without a generated
 * kernel never wait.  An engine owns ONE set of hand-over buffers: calls on
 * the same engine must be ordered with respect to each other (same stream, or
 * event-ordered); use one engine per concurrent stream.
 */
#ifndef CSIM_H
#define CSIM_H

#include <stdint.h>
#include "csim_ir.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CSIM_OK                0
#define CSIM_ERR_ARG          -1   /* null pointer, bad size                       */
#define CSIM_ERR_IO           -2   /* netlist file cannot be opened                */
#define CSIM_ERR_NO_DEVICE    -3   /* no usable HIP device: there is no CPU path   */
#define CSIM_ERR_HIP          -4   /* a HIP runtime call failed                    */
#define CSIM_ERR_UNSUPPORTED  -5   /* circuit outside what the kernels cover       */
#define CSIM_ERR_EMPTY        -6   /* circuit has no unknowns                      */
#define CSIM_ERR_CONFIG       -7   /* invalid .TRAN numbers (tstep/tstop <= 0)     */

typedef struct csim_netlist csim_netlist;
typedef struct csim_engine  csim_engine;

const char* csim_last_error(void);
const char* csim_version(void);

/* ---------------------------------------------------------------- netlist
 * Parse + index a netlist (host only, runs once, microseconds).             */
int  csim_netlist_parse_file(const char* path, csim_netlist** out);
/* same, from memory (what a rank receives from the broadcast of the netlist) */
int  csim_netlist_parse_text(const char* text, int64_t len, csim_netlist** out);
void csim_netlist_free(csim_netlist* nl);

/* flattened circuit; owned by the netlist, valid until csim_netlist_free */
const csim_ir* csim_netlist_ir(const csim_netlist* nl);
/* "Circuit summary" numbers of src/main.cpp:36-41 */
int  csim_netlist_counts(const csim_netlist* nl, int32_t* n_nodes, int32_t* n_elems,
                         int32_t* n_unknowns, int32_t* n_node_eq, int32_t* n_branch_eq);
/* nominal parameter vector, P doubles */
int  csim_netlist_nominal_params(const csim_netlist* nl, double* out);
/* node name (node equations) or element name (branch equations) of equation eq */
const char* csim_netlist_eq_name(const csim_netlist* nl, int32_t eq);
/* equation index of a node name, -1 for ground, -2 if unknown */
int  csim_netlist_node_eq(const csim_netlist* nl, const char* node_name);
/* .TRAN card (src/parser.cpp:497-524) */
int  csim_netlist_tran(const csim_netlist* nl, int32_t* enabled, double* tstep, double* tstop, double* tstart);
/* node-voltage probes named by .PLOTNV / .PRINT cards, as equation indices */
int  csim_netlist_num_probes(const csim_netlist* nl);
int  csim_netlist_probe_eq(const csim_netlist* nl, int32_t i);
/* .DC cards (src/parser.cpp:476-495): source element index and sweep numbers */
int  csim_netlist_num_dc_sweeps(const csim_netlist* nl);
int  csim_netlist_dc_sweep(const csim_netlist* nl, int32_t i, int32_t* src_elem,
                           double* start, double* stop, double* step);
/* .DC sweep as a batch axis (the reference parses the card and never executes it,
 * src/parser.cpp:476-495; src/main.cpp never reads sim.dcSweeps): sweep point j is the
 * nominal circuit with the swept source's dcValue replaced by start + j*step.
 * points = floor((stop-start)/step + 1e-9) + 1, 0 for step == 0, a step of the wrong sign
 * or a card whose source is not a V/I element.                                        */
int64_t csim_netlist_dc_sweep_points(const csim_netlist* nl, int32_t i);
/* parameter table [P][n_points] (slot-major) + the swept values [n_points]              */
int  csim_netlist_dc_sweep_params(const csim_netlist* nl, int32_t i, int64_t n_points,
                                  double* params, double* values);
/* header line of the reference's transient CSV (src/tanalisis.cpp:191-206):
 * "time,V(<node>)...,I(<elem>)..."; returns the length needed (excl. NUL) */
int  csim_netlist_csv_header(const csim_netlist* nl, char* buf, int32_t cap);
/* Monte-Carlo recipe per parameter slot: 0 fixed, 1 scaled by (1+sigma z),
 * 2 MOS K rebuilt from a MU draw: K = (MU(1+sigma z))*COX*(W/L)              */
int  csim_netlist_mc_kinds(const csim_netlist* nl, int32_t* kinds);

/* ----------------------------------------------------------------- engine */
/* One engine per (circuit, device).  Uploads the circuit plan; owns only its
 * handle and device scratch.  CSIM_ERR_NO_DEVICE if `device` is not a usable
 * HIP device.                                                                */
int  csim_engine_create(const csim_netlist* nl, int32_t device, csim_engine** out);
void csim_engine_destroy(csim_engine* eng);
/* which transient kernel the engine will use: "general" (wave-per-instance,
 * dense LDS LU with dynamic pivoting), "scheduled" (circuit-specialised code with
 * a verified pivot schedule) or "faithful" (the same with the reference's arithmetic) */
const char* csim_engine_tran_kernel(const csim_engine* eng);
/* description of the loaded generated library ("" if none): circuit, pivot schedules, LDS doubles
 * per lane, and the floating-point operations ONE solve on the first schedule executes
 * ("ops_per_solve: fma=.. mul=.. addsub=.. recip=.. cmp=..")                                   */
const char* csim_engine_sched_info(const csim_engine* eng);
/* lanes per instance the scheduled transient kernel would use for a batch of B instances (1, 4 or 16;
 * 0 = the general kernel runs: one 64-lane wavefront per instance).  A linear circuit's library has one
 * transient kernel: 16 (tape and iterate in registers) or 1 (larger circuits), whatever B is.        */
int  csim_engine_lanes_for_batch(const csim_engine* eng, int32_t B);
/* force a kernel family: 0 = auto, 1 = general only, 2 = scheduled required, 3 = the generated kernel
 * with the reference's arithmetic ("faithful": true divisions, no FMA contraction, steps at the NR cap
 * kept and flagged; bit-faithful on recorded pivot sequences, run-time pivoting kernel for the rest)    */
int  csim_engine_set_kernel(csim_engine* eng, int32_t which);

/* Run-time options of one engine, as text.  The environment variable named with each key is read
 * ONCE, in csim_engine_create, as the key's default; nothing on a hot path reads the environment.
 *   hybrid_rounds (CSIM_HYBRID_ROUNDS, 4)   hand-back rounds between the scheduled and the general
 *                                           kernel per transient call
 *   hybrid_steps  (CSIM_HYBRID_STEPS, 64)   most steps the general kernel keeps an instance per round
 *   lanes_per_instance (CSIM_LANES_PER_INSTANCE, 0)  scheduled transient kernel: 1 = lane per instance,
 *                                           4 = four lanes per instance (circuits of up to 32 unknowns),
 *                                           16 = sixteen lanes per instance (up to 96 unknowns), 0 = chosen by batch size
 *                                           (<= 4096: 16; up to 16 384: 4; beyond: 1)
 *   sched_variant (CSIM_SCHED_VARIANT, 0)   tuning kernels of a generated library (2 rich, 10+k sweep)
 *   auto_jit (CSIM_AUTO_JIT, off)           csim_tran_batch specialises a new circuit on first use
 *   jit_dir (CSIM_JIT_DIR; default $XDG_CACHE_HOME/csim_jit or /tmp/csim_jit.<uid>)  JIT cache: created
 *                                           0700; must be a real directory of the calling user, not
 *                                           writable by group/others; only regular files of the calling
 *                                           user are ever loaded from it
 *   hipcc (CSIM_HIPCC, /opt/rocm/bin/hipcc), jit_timeout (CSIM_JIT_TIMEOUT, 600 s)
 *   jit_dc_alts (CSIM_JIT_DC_ALTS, 4), jit_dc_force (CSIM_JIT_DC_FORCE, off)  DC schedules kept by the JIT
 *   jit_gen_opts                            generator options of this engine's JIT, "key=value,key=value"
 *                                           (near_band, near_band_dc, stage_ahead, group4, place_search, ...:
 *                                           engine/codegen.hpp);
 *                                           they are part of the hash a cached library is checked against
 *   near_test_rollback (0)                  test aid: every verified near-threshold decision is treated as a
 *                                           mismatch, so the roll-back path runs (results must not change)
 *   hybrid_sync (CSIM_HYBRID_SYNC, 1)       see "Streams" above
 *   dc_fast (CSIM_DC_FAST, 0)               DC operating points start on the fast generated kernel (FMA
 *                                           contraction, reciprocal pivots; controller decisions within
 *                                           its rounding noise are replayed) instead of the faithful one
 * Unknown key or bad value: CSIM_ERR_ARG.                                                        */
int  csim_engine_set_option(csim_engine* eng, const char* key, const char* value);
/* Counters of one engine since its creation (-1: unknown key).  Maintained in synchronous mode
 * (hybrid_sync = 1) only:
 *   "near_verified"      near-threshold convergence decisions of the fast transient kernels that the
 *                        faithful kernel re-did (src/tanalisis.cpp:369; DESIGN.md "near-threshold guard")
 *   "near_rolled_back"   of those, the ones whose pass count differed: the instance was rolled back      */
int64_t csim_engine_stat(const csim_engine* eng, const char* key);

/* Monte-Carlo parameter table on the device: instance b_first+i of the global
 * batch -> column i.  Instance 0 is the nominal circuit.  Counter-based:
 * any shard regenerates any instance from (seed, instance, slot).            */
int  csim_mc_params_dev(csim_engine* eng, uint64_t seed, double sigma, int64_t b_first,
                        int32_t B, double* d_params /*[P][B]*/, void* stream);
/* host mirror of the same generator (bit-identical; for fixtures and tests)  */
int  csim_mc_params_host(const csim_netlist* nl, uint64_t seed, double sigma, int64_t b_first,
                         int32_t B, double* params /*[P][B]*/);

/* DC operating point of B instances.                                         */
int  csim_dc_batch_dev(csim_engine* eng, const double* d_params /*[P][B]*/, int32_t B,
                       double* d_x /*[N][B]*/, int32_t* d_iters /*[B]*/,
                       uint32_t* d_status /*[B]*/, void* stream);

/* n_steps backward-Euler time steps for B instances, steps
 * step_first+1 .. step_first+n_steps of the run (t = step*tstep).
 *   d_x      in: state at step_first (the DC solution for step_first == 0);
 *            out: state after the last step.  Histories (capacitor voltages,
 *            inductor currents, MOS junction voltages) are functions of the
 *            previous state, so x is the whole per-instance state.
 *   d_wave   optional [n_rows_total][n_probe][B]: row r holds step r*out_stride
 *            (row 0 = t=0 state, written when step_first == 0).
 *   d_iters  [B], accumulated (+=): NR iterations.   d_status [B], OR-ed.
 *   d_step_iters optional [n_steps][B] NR iterations of each step of this call */
int  csim_tran_batch_dev(csim_engine* eng, const double* d_params /*[P][B]*/, int32_t B,
                         double tstep, int64_t step_first, int64_t n_steps,
                         const int32_t* probe_eq /*host*/, int32_t n_probe, int32_t out_stride,
                         double* d_wave, double* d_x /*[N][B]*/, int64_t* d_iters,
                         uint32_t* d_status, int32_t* d_step_iters, void* stream);

/* Host-pointer forms.  params [B][P] instance-major (NULL =
 * nominal for every instance), x_out/x_final [B][N].                          */
int  csim_dc_batch(csim_engine* eng, const double* params, int32_t B,
                   double* x_out, int32_t* nr_iters, uint32_t* status);
/* Runs DC then the whole transient.  wave_out optional
 * [B][n_rows][n_probe], n_rows = floor(nSteps/out_stride)+1 minus rows with
 * t < tstart (suppressed like dumpRow, src/tanalisis.cpp:208-209).  Waveforms
 * are streamed out in chunks while the next chunk is computed; device memory
 * does not grow with the length of the run.  CSIM_AUTO_JIT=1 in the
 * environment specialises a circuit without a prebuilt kernel on first use.   */
int  csim_tran_batch(csim_engine* eng, const double* params, int32_t B,
                     double tstep, double tstop, double tstart,
                     const int32_t* probe_eq, int32_t n_probe, int32_t out_stride,
                     double* wave_out, double* x_final, int64_t* nr_iters, uint32_t* status);
/* The transient of ONE instance of a batch description, written as the reference's CSV
 * (src/tanalisis.cpp:189-231: header "time,V(<node>)...,I(<source or inductor>)...", values "%.9e", one row per
 * time step, rows with t < tstart suppressed) -- what plot_tran.py reads.  For the Monte-Carlo instances one
 * wants to look at; the whole batch's waveforms go through csim_tran_batch's wave_out.
 *   params    [B][P] instance-major host table or NULL (nominal); `instance` picks its row
 *   probe_eq  columns (equation indices) in file order; n_probe == 0: the netlist's .PLOTNV / .PRINT probes
 *             when it names any (src/parser.cpp:630-723), else every unknown -- the reference's own file
 * Runs the DC operating point and the transient of that instance alone (a batch of one on the engine's
 * kernels) and formats on the host.  CSIM_ERR_IO if the file cannot be written.                       */
int  csim_tran_write_csv(csim_engine* eng, const double* params, int32_t B, int32_t instance,
                         double tstep, double tstop, double tstart,
                         const int32_t* probe_eq, int32_t n_probe, const char* path);
/* number of rows csim_tran_batch writes per instance for these numbers       */
int64_t csim_tran_num_rows(double tstep, double tstop, double tstart, int32_t out_stride);
int64_t csim_tran_num_steps(double tstep, double tstop);

/* Batched dense solve A x = b with the engine's pivoted LU
 * (Solver::solveLinearSystemLU semantics: first-maximum partial pivoting,
 * tiny pivot -> zero vector).  A [B][n][n] row-major, b/x [B][n], host
 * pointers.  flags [B] optional (CSIM_ST_LU_*).  device: HIP device index.
 * n <= 63 runs LDS-resident, 64 <= n <= 1024 in place in global memory.       */
int  csim_lu_solve_batch(int32_t device, int32_t n, int32_t B, const double* A,
                         const double* b, double* x, uint32_t* flags);

/* ---- Gauss-Seidel variant of the reference (never reached from its main(), kept as public API) ----
 * Batched Solver::solveLinearSystemGaussSeidel (include/solver.hpp:139-204): sweeps in row order with the
 * newest values, a diagonal below 1e-12 replaced by +-1e-12, stop when ||x - x_prev|| < tol or after
 * max_iters sweeps; whatever the sweeps left is returned (it may be non-finite).  A [B][n][n] row-major,
 * b / x0 / x [B][n] host pointers; x0 NULL = start from zero (the two-argument overload); sweeps [B]
 * optional = sweeps performed.  One lane per system, same operation order as the reference.            */
int  csim_gs_solve_batch(int32_t device, int32_t n, int32_t B, const double* A, const double* b,
                         const double* x0, int32_t max_iters, double tol, double* x, int32_t* sweeps);
/* dcSolveGaussSeidel (src/dcanalysis.cpp:71-92,166-237,254-258) for B instances: linear circuits one
 * Gauss-Seidel solve (2000 sweeps, 1e-10), circuits with MOSFETs the source ramp with 60 (last step 120)
 * Newton passes per step, inner solve warm-started from x, ConvController update.  A pass whose inner
 * solve turns non-finite raises gmin x10 and is dropped (CSIM_ST_DC_NONFINITE), exactly as upstream; on
 * circuits with voltage sources (zero diagonal entries) that is every pass and x stays 0.  A LINEAR
 * circuit whose sweeps diverge returns what the reference's dense loops leave (a pattern of +-inf / NaN,
 * reproduced component by component) and no flag -- upstream checks nothing there (:89-91).  N <= 63.   */
int  csim_dc_gs_batch_dev(csim_engine* eng, const double* d_params /*[P][B]*/, int32_t B,
                          double* d_x /*[N][B]*/, int32_t* d_iters, uint32_t* d_status, void* stream);
int  csim_dc_gs_batch(csim_engine* eng, const double* params /*[B][P] or NULL*/, int32_t B,
                      double* x_out /*[B][N]*/, int32_t* nr_iters, uint32_t* status);

/* Runtime specialisation for a netlist without a prebuilt libcsim_sched_<topology>.so:
 * records the pivot schedule of instance 0 of d_params with the general kernel
 * (plan_steps transient steps), generates the lane-per-instance kernel, compiles it
 * (up to 4 distinct sequences seen while planning become alternatives), compiles it
 * with hipcc (--offload-arch=gfx950; a child process started from an argument vector,
 * no shell, killed after jit_timeout seconds) into the private JIT cache directory
 * (csim_engine_set_option) and loads it.  A cached library is reused only if it is the
 * caller's own regular file and reports the hash of exactly this (topology, constants,
 * schedules, generator revision).  After CSIM_OK, csim_engine_tran_kernel() reports
 * "scheduled".                                                                      */
int  csim_engine_jit_scheduled(csim_engine* eng, const double* d_params /*[P][B]*/, int32_t B,
                               double tstep, int64_t plan_steps);
/* The build half of the above for schedules the caller already has (from the planner
 * entry points below, or from a schedule file): pivot_pos [n_alts][N] transient
 * sequences, most frequent first (1..16); dc_pivot_pos [n_dc_alts][N] sequences of the DC
 * operating point (0..8; 0 = no DC kernel).  Schedules decide speed only: every
 * factorisation re-verifies the sequence it uses.                                   */
int  csim_engine_jit_with_schedules(csim_engine* eng, const int32_t* pivot_pos, int32_t n_alts,
                                    const int32_t* dc_pivot_pos, int32_t n_dc_alts);

/* Batched Solver::luDecompose (include/solver.hpp:30-80): LU [B][n][n] holds U on
 * and above the diagonal and the multipliers below it, perm [B][n] the row
 * permutation (b_perm[i] = b[perm[i]]).  flags[b] = CSIM_ST_LU_TINY_PIVOT where
 * the reference returns false (LU/perm of that system are then unspecified).  */
int  csim_lu_decompose_batch(int32_t device, int32_t n, int32_t B, const double* A,
                             double* LU, int32_t* perm, uint32_t* flags);

/* Planner: run DC + n_steps transient steps of ONE instance (column `instance` of
 * d_params) with the general kernel and report the partial-pivot row position chosen
 * for every column in the first transient factorisation (pivot_pos[N], host), the
 * number of factorisations seen and how many used a different sequence.  This is what
 * a schedule file of csrc/schedules/ is recorded from.                              */
int  csim_record_pivot_schedule(csim_engine* eng, const double* d_params /*[P][B]*/, int32_t B,
                                int32_t instance, double tstep, int64_t n_steps,
                                int32_t* pivot_pos, int64_t* n_factorizations, int64_t* n_differ);
/* Same run, every DISTINCT sequence seen (at most 8, most frequent first): pivot_pos
 * [max_alts][N], counts [max_alts] = factorisations that used each, *n_alts = how many were
 * found, *n_other = factorisations that failed or used a sequence beyond the eighth.  A
 * switching circuit alternates between a few sequences; a schedule file may list several. */
int  csim_record_pivot_schedules(csim_engine* eng, const double* d_params /*[P][B]*/, int32_t B,
                                 int32_t instance, double tstep, int64_t n_steps, int32_t max_alts,
                                 int32_t* pivot_pos, int64_t* counts, int32_t* n_alts, int64_t* n_other);
/* The same planner on the DC operating point of instance `instance` (source ramp + adaptive
 * gmin, reference src/dcanalysis.cpp:95-163): its distinct pivot sequences, most frequent first
 * ("dc" lines of a schedule file).  Circuits of up to 63 unknowns.                           */
int  csim_record_dc_pivot_schedules(csim_engine* eng, const double* d_params /*[P][B]*/, int32_t B,
                                    int32_t instance, int32_t max_alts, int32_t* pivot_pos,
                                    int64_t* counts, int32_t* n_alts, int64_t* n_other);

#ifdef __cplusplus
}
#endif
#endif /* CSIM_H */
