// group_plan.hpp -- plan of the SIXTEEN- (and FOUR-) LANES-PER-INSTANCE scheduled kernels.
//
// The lane-per-instance kernel (codegen.cpp) needs 64 instances per wavefront; BASELINE's
// headline batch (4096 instances) then fills 64 of the chip's 1024 SIMDs.  This plan splits
// ONE instance's solve over a DPP row of 16 lanes (4 instances per wave, 1024 waves at
// B = 4096) for the loop of the reference that is parallel over rows: for a fixed column k
// the updates of the rows below the pivot are independent (include/solver.hpp:70-76).
//
// Layout ("position-cyclic").  For a recorded pivot schedule the row that ends at pivot
// position pos is known before the kernel runs.  Lane g of a group, slot s, holds the row of
// final position 16*s + g: register class a[s][j] = entry (16*s+g, j) of the permuted system,
// j = N the right-hand side.  Consequences:
//   * the pivot row of column k is always lane k%16, slot k/16; its entries right of the
//     diagonal are broadcast with DPP row_newbcast (v_mov_b64_dpp), unless the generator
//     knows them to be exact constants (the +-1 incidence rows of sources and inductors);
//   * rows of slots < k/16 are finished U rows, rows of slots > k/16 are all still active,
//     and in slot k/16 exactly the lanes g > k%16 are active: one per-lane 0/1 factor
//     (launch constant) on the multiplier keeps finished rows untouched;
//   * a structural zero below the pivot is an exact 0.0 in its register, its multiplier is
//     0 and the update a - 0*u leaves the entry as it is -- what the reference's dense loops
//     do with it (solver.hpp:71-75);
//   * back substitution runs column-wise: x_j is formed in lane j%16, broadcast, and
//     subtracted from the right-hand sides of the rows above; rows at or below j hold dead
//     right-hand sides by then, so they need no protection.
// Four lanes per instance (lanes = 4: a DPP quad, 16 instances per wavefront, up to 8 rows per lane) is the same plan
// with G = 4; there the rows are not placed position-cyclically but by optimizeGroupPlacement() below (which rows share
// a slot decides how many update instructions a column costs), so pivot rows sit at arbitrary lanes and the finished
// rows of a slot are removed by explicit lane masks -- the form every ALTERNATIVE schedule has always had, since
// alternatives are planned over the first schedule's placement.
// Arithmetic differences from the reference (inside the 1e-9 bar, like the lane-per-instance
// kernel): FMA contraction, one Newton-refined reciprocal per pivot, the per-iteration matrix
// is formed as (terms that do not change within a time step) + (MOSFET terms) instead of one
// sum in stamping order, and back substitution subtracts in descending column order.
// Every factorisation still verifies the recorded pivot choice against the reference's rule
// (first row attaining the column maximum, >= 1e-15); a failed check sends the instance to
// the general kernel exactly as in the lane-per-instance kernel.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "codegen.hpp"
#include "csim_ir.h"
#include "plan.hpp"

namespace csim {

constexpr int kGroupLanes = 16;

struct GroupPlan {
    int N = 0, S = 0;                        // unknowns, slots per lane = ceil(N / G)
    int G = kGroupLanes;                     // lanes per instance: 16 (one DPP row) or 4 (one quad; 16 instances per wavefront)
    int ldsDoubles = 0;                      // LDS image of one instance in the emitted kernel (set by emitGroupKernel)
    std::vector<int> finalPos;               // original row -> pivot position
    std::vector<int> rowAtPos;               // pivot position -> original row

    struct Check { int slot; bool strict; unsigned laneMask; };     // |pivot| > (strict) or >= |a[slot][k]| on these lanes
    struct UEntry { int j; bool isConst; double c; };                // pivot-row entry right of the diagonal (j == N: rhs)
    struct Column {
        int pivLane = 0, pivSlot = 0;        // where the pivot row of this column sits (first schedule: k%16, k/16)
        // per slot of lSlots: the lanes whose rows are still unpivoted after this column (the rows an update
        // may touch); keepAll = no finished row and no pivot row in the slot (no factor needed); suffix >= 0:
        // the mask is "lane > suffix" (the launch-constant factors mk<suffix>)
        struct SlotMask { unsigned lanes = 0; bool keepAll = false; int suffix = -1; };
        std::vector<SlotMask> lMask;         // parallel to lSlots
        std::vector<SlotMask> checkMask;     // per slot s (size S): the same notion for the candidates' magnitudes
        bool zeroPivot = false;              // scheduled pivot is a structural zero: always a violation
        bool contradiction = false;          // schedule contradicts exact constants: always a violation
        bool pivotConst = false;             // pivot is an exact constant (rinv is a literal, no eps test)
        double pivotValue = 0.0;
        std::vector<Check> checks;
        std::vector<int> lSlots;             // slots holding rows with a non-zero below the pivot
        std::vector<UEntry> u;
    };
    std::vector<Column> cols;
    // elimination instructions (per wave, one solve) by what their operands depend on: [1] terms constant over the
    // launch, [2] per-step terms (right-hand side), [3] the iterate (MOSFET terms); [0] exact constants
    std::array<int, 4> opsByLevel{{0, 0, 0, 0}};
    std::vector<std::vector<int>> backSlots; // [j]: slots with rows pivoted before column j whose U(i,j) may be non-zero
    std::vector<std::vector<uint8_t>> classLive;   // [s][j], j <= N: the register class is ever non-zero

    // ---- assembly
    // (a) terms that do not change within a Newton loop, gathered per cell in the reference's
    //     stamping order; cell = class * 16 + lane; con = (term << 1) | negate
    struct ClassRef { int s, j; };
    std::vector<ClassRef> gClasses;          // matrix classes with launch-constant terms
    std::vector<int32_t> gCellPtr, gCellCon;
    std::vector<int32_t> iCellPtr, iCellCon; // right-hand side: cell = s * 16 + lane (per-step terms)
    // (b) MOSFET channel terms, evaluated by lane m of the group for MOSFET m and scattered into
    //     staging rows of 16 cells; the owning lanes add their cells in stamping order
    std::vector<int> mosElem;                // element index of MOSFET m
    struct StageRow { int s, j; };
    std::vector<StageRow> stageRows;         // in the order in which they are added
    // destination cell (stageRow * 16 + lane) of MOSFET m's stamp kind, or -1 (ground / absent):
    // kinds = {DD+gd, DG+gg, DS+gs, ID-cst, SD-gd, SG-gg, SS-gs, IS+cst} (element.cpp:290-304)
    std::vector<std::array<int, 8>> mosDest;

    // operation counts of one solve (wave-level instructions, 4 instances each)
    int nBcast = 0, nFma = 0, nMul = 0, nCmp = 0, nRecip = 0;
    // critical path of one solve in cycles under a simple latency model (8 cycles per dependent VALU
    // result, 40 for a Newton-refined reciprocal): after the elimination, and after back substitution
    int depthElimination = 0, depthSolve = 0;
};

// false: the circuit does not fit this kernel (sixteen lanes: N > 96; four lanes: N > 32).  placement: where the rows sit.  The first
// schedule of a circuit places them itself (row of final pivot position p at lane p%16, slot p/16); further
// alternatives are planned over THAT placement (placement = the first plan), so that they share the
// step-constant matrix part and the MOSFET staging: their pivot rows then sit at arbitrary lanes and the
// finished rows of a slot are no longer a lane prefix (explicit lane masks instead of the mk factors).
bool buildGroupPlan(const csim_ir& ir, const AssemblyPlan& ap, const PivotSchedule& sch, GroupPlan& out,
                    const GroupPlan* placement = nullptr, int lanes = kGroupLanes);

// Row placement by local search (exchanges of two rows' cells) on the planned instruction count of the solves, the
// first schedule's weighing most.  placement receives N, G, finalPos, rowAtPos: pass it to buildGroupPlan.
bool optimizeGroupPlacement(const csim_ir& ir, const AssemblyPlan& ap, const std::vector<PivotSchedule>& schedules, int lanes,
                            GroupPlan& placement, double* costBefore = nullptr, double* costAfter = nullptr);

// Host interpreter of the plan, lane by lane, for ONE system: T[nTerms] are the term values of
// plan.hpp.  Writes x[N]; *violated = a pivot check failed; *planError = a lane the kernel's all-lane
// candidate test would see is neither a candidate, masked, nor an exact zero.  Used by the self test
// (csim_codegen --selftest-group) to validate masks, placement, staging and substitution order
// against a plain pivoted elimination, without a GPU.
void interpretGroupPlan(const GroupPlan& gp, const AssemblyPlan& ap, const csim_ir& ir, const double* T, double eps,
                        double* x, bool* violated, bool* planError = nullptr);

// device code shared by the sixteen-lanes-per-instance kernels (DPP broadcast / row sum, source waveforms, the
// MOSFET linearisation): emitted once per generated library, before emitGroupKernel's / emitLinearGroupKernel's text
std::string groupPreludeSource(const csim_ir& ir);

// the __global__ kernel "csim_tran_group_kernel" + its tables; "" if the circuit does not fit.  One solve
// body per schedule, tried in order per Newton pass for the groups whose checks failed so far.
// lanes = 4: the same kernel over DPP quads, "csim_tran_group4_kernel" (16 instances per wavefront; tables in
// namespace csim_q4).
std::string emitGroupKernel(const csim_ir& ir, const AssemblyPlan& ap, const std::vector<PivotSchedule>& schedules,
                            const GeneratorOptions& gopt, GroupPlan* planOut, int lanes = kGroupLanes);

} // namespace csim
