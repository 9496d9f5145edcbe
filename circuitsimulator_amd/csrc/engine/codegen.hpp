// codegen.hpp -- generator of circuit-specialised ("scheduled") transient kernels.
//
// The general kernels give one 64-lane wavefront to ONE instance and spend it
// on a dense LDS matrix that is ~90 % zeros.  For a fixed topology the zero
// pattern, the fill pattern and -- through the whole transient, for every
// Monte-Carlo instance sampled so far -- the partial-pivot row-swap sequence
// are the same.  The generator therefore
// EXECUTES the reference's algorithm symbolically once per circuit:
//
//   stamp (src/tanalisis.cpp:269-356) -> luDecompose with the recorded pivot
//   rows (include/solver.hpp:46-77) -> substitution (:100-128) -> damped
//   update (src/tanalisis.cpp:365-371)
//
// over the abstract domain {structural zero, exact constant, run-time value}
// and emits straight-line HIP code for the run-time values only.  One LANE
// owns one instance; the whole working set lives in registers; there is no
// LDS matrix and no cross-lane traffic.  The emitted code
//   * performs the same elimination steps in the same order as the reference
//     (a skipped operation is one whose multiplier or pivot-row entry is a
//     structural zero: a - 0*b == a; a folded one multiplies by exactly +-1),
//   * VERIFIES at every column that the scheduled pivot row is the row the
//     reference would pick (first row attaining the column maximum, and
//     >= 1e-15); an instance that fails a check is not advanced: it is handed
//     to the general kernel for that launch and flagged CSIM_ST_SCHED_FALLBACK.
// Deliberate floating-point differences from the reference (bounded by the
// 1e-9 parity bar, NR counts must still match): FMA contraction, and
// multipliers formed with one reciprocal per pivot instead of one division
// per multiplier.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "csim_ir.h"
#include "plan.hpp"

namespace csim {

// pivotPos[k] = current row position (>= k) holding the pivot of column k
struct PivotSchedule {
    std::vector<int> pivotPos;
    static PivotSchedule identity(int N);
    // "k:p,k:p,..." (only the swapped columns), e.g. "0:21,8:22"
    static bool parse(const std::string& text, int N, PivotSchedule& out);
    std::string str() const;
};

// Several schedules of one circuit (a switching circuit alternates between a few pivot
// sequences).  The generated kernel carries one solve body per alternative and tries them in
// order, per Newton iteration, for the lanes whose pivot checks failed so far.
struct ScheduleSet {
    std::vector<PivotSchedule> alts;       // transient; most frequent first
    std::vector<PivotSchedule> dcAlts;     // DC operating point (lines prefixed "dc"); may be empty
    // one schedule per line (or ';'-separated); '#' starts a comment; "-" = no swaps;
    // "dc <schedule>" = a schedule of the DC Newton solve
    static bool parse(const std::string& text, int N, ScheduleSet& out);
    std::string str() const;
};

// Everything besides (circuit, schedules) that changes the emitted code.  There are NO environment
// switches in the generator: options come from the caller (csim_codegen's command line; the engine's
// JIT always uses the defaults) and are part of the hash that names a cached library.
struct GeneratorOptions {
    int barrierEvery = 3;        // scheduling barrier every n-th elimination column (0 = none)
    std::vector<int> sweep;      // tuning aid: extra kernels csim_tran_sched_kernel_sweep<k> (see codegen.cpp)
    int stageAhead = 3;          // sixteen-lane kernel: a staging row is read this many columns before the column
                                 // that first needs it (-1: all rows read and added before the elimination)
    int pipelineMos = 1;         // sixteen-lane kernel: 1 = the MOSFET pass and the staging reads of iteration i+1 run at
                                 // the end of iteration i, under its convergence bookkeeping; 0 = at the head of i+1
    int groupWavesPerEu = 0;     // sixteen-lane kernel: amdgpu_waves_per_eu(n, n) on the kernel (0 = leave it to the compiler)
    // Near-threshold guard of the FAST kernels (FMA contraction, reciprocal pivots).  A branch-deciding comparison
    // whose two sides agree to within this relative band could fall the other way in the reference's arithmetic:
    //  transient `err < tol` (src/tanalisis.cpp:369): the kernel goes on speculatively, keeps the state at the
    //    start of that time step, and the engine has the bit-faithful generated kernel redo the step afterwards --
    //    equal pass count: the speculation stands, else the instance is rolled back to that step;
    //  DC `err < tol`, `err > prevErr * slow`, `err < prevErr * fast` (src/dcanalysis.cpp:150,285-296): the
    //    instance is replayed by the bit-faithful DC kernel.
    // Measured noise of err at the deciding pass, FMA against non-FMA build of the oracle on dbmixer.sp: transient
    // median 9e-11, max 1.2e-9 (relative); DC up to 6e-7 (tol 1e-9 magnifies the solve's rounding).  0 = no guard.
    double nearBand = 2e-8;
    double nearBandDc = 1e-5;
    int ldsPad = 0;              // sixteen-lane kernels: 1 = per-instance LDS stride padded to 16 (mod 32) doubles, 0 = as it comes
                                 // (measured, profiles/r03_group16_b4096_*: SQ_LDS_BANK_CONFLICT 30.9 % of the LDS-active cycles
                                 // either way -- the conflicts come from the per-lane gathers / scatters of the MOSFET pass, not
                                 // from the [row][16] reads -- and 0.8 % slower padded)
    int dummyOneCell = 1;        // sixteen-lane kernel: lanes without a MOSFET (and stamps to ground) scatter into ONE dummy cell per instance
                                 // instead of one each: writes to one address do not conflict, sixteen scattered dummies did with the real
                                 // destinations (SQ_LDS_BANK_CONFLICT 3.85e8 -> 2.97e8 per launch, 30.8 % -> 25.6 % of the LDS-active
                                 // cycles; +0.4 % on the bench, same box: gpurun_out/r03v)
    int group4 = 1;              // 1 = the library also carries the four-lanes-per-instance form of the group kernel (csim_tran_group4_kernel),
                                 // for circuits of up to 32 unknowns: the batches between the sixteen-lane and the lane-per-instance kernel
    int placeSearch = 1;         // group kernels: rows are placed by a local search on the solves' planned instruction count instead of
                                 // position-cyclically; bit 0: the four-lane kernel, bit 1: the sixteen-lane kernel
    int linChainBarrier = 0;     // linear sixteen-lane kernel: scheduling barrier after each critical link of the forward chain
    int linFactorBlock = 64;     // linear sixteen-lane library: lanes per workgroup of the factor kernel (0 = 16 / 32 / 64 by batch size;
                                 // measured on the N = 257 ladder at B = 8192, same box: 64 -> 3.09 ms per launch, by batch size (16) -> 3.16 ms)
    int linSrcLds = 1;           // linear sixteen-lane kernel: 1 = the sources' parameters are copied to LDS once per launch
    int nearForm = 0;            // sixteen-lane kernel, how a pass records a near tie: 0 = running minimum of |err - tol| (two
                                 // VALU instructions, one loop-carried double), 1 = two more compares into a loop-carried lane mask
    bool set(const std::string& keyval);      // "barrier_every=3", "sweep=0,16,32", "stage_ahead=3", "pipeline_mos=0", "group_waves=2", "near_band=2e-8", "near_band_dc=1e-5"
};

// bumped whenever the emitted code or the launcher ABI of a generated library changes
constexpr int kGeneratorRevision = 41;

// identifies (topology, constants, schedule); names the generated library
uint64_t scheduleHash(const csim_ir& ir, const PivotSchedule& sch);
uint64_t scheduleHash(const csim_ir& ir, const ScheduleSet& set, const GeneratorOptions& gopt = GeneratorOptions());

struct CodegenStats {
    int nMul = 0, nFma = 0, nAddSub = 0, nRecip = 0, nCmp = 0, nDynU = 0, nLower = 0;
};

// codegen_linear.cpp: kernel for circuits without nonlinear devices (factor once per launch, substitute
// once per step); "" when the circuit has MOSFETs or its iterate does not fit LDS.  workDoubles = doubles
// per instance of the factor store the launcher needs, lanesPerWave = instances per workgroup.
std::string emitLinearKernel(const csim_ir& ir, const AssemblyPlan& ap, const PivotSchedule& sc,
                             int* workDoubles, int* lanesPerWave);

// codegen_linear.cpp: the same for SIXTEEN lanes per instance with iterate, x_raw and the factor tape in registers
// (kernels csim_lin16_factor_kernel + csim_tran_linear16_kernel; needs groupPreludeSource() in front); "" when the
// circuit has MOSFETs or its tape does not fit the register file.  workDoubles: doubles per instance of the tape.
std::string emitLinearGroupKernel(const csim_ir& ir, const AssemblyPlan& ap, const PivotSchedule& sc, int* workDoubles,
                                  const GeneratorOptions& gopt = GeneratorOptions());

// codegen_linear.cpp: DC operating point of a linear circuit (one direct solve, lane per instance, the reference's
// arithmetic) on the recorded DC pivot schedule; workDoubles: doubles per instance of its tape
std::string emitLinearDcKernel(const csim_ir& ir, const AssemblyPlan& ap, const PivotSchedule& dcSchedule, int* workDoubles);

// complete .hip translation unit: kernel + extern "C" launcher + metadata
std::string generateTranKernelSource(const csim_ir& ir, const AssemblyPlan& ap, const ScheduleSet& set,
                                     const std::string& label, CodegenStats* stats,
                                     const GeneratorOptions& gopt = GeneratorOptions());

} // namespace csim
