// plan.hpp -- assembly plan: the reference's stamping sequence, recorded
// symbolically once per circuit on the host, so that the device can build
// G and I by GATHERING.
//
// The reference builds the system by scattering: every element's stamp() does
// "G(r,c) += v" in netlist order (src/dcanalysis.cpp:126-130,
// src/tanalisis.cpp:269-356).  On a wavefront that is a scatter with
// collisions.  Instead each structural non-zero of G (and each row of I) owns
// the ordered list of terms the reference would have accumulated into it:
//     G(r,c) = ((0 + s1*T[t1]) + s2*T[t2]) + ...      (same order, same bits)
// where T is a small per-instance table of element values ("terms") that
// device code evaluates element-parallel: 1/R, C/dt, L/dt, source values,
// MOS gd/gg/gs/cst, history currents, gmin, 1.0.
#pragma once

#include <cstdint>
#include <vector>

#include "csim_ir.h"

namespace csim {

// term slots of one element, relative to termBase[e]
enum TermOffset {
    T_R_G = 0,                                       // R  : 1/R
    T_C_GC = 0, T_C_IH = 1,                          // C  : C/dt, -Gc*vPrev
    T_L_REQ = 0, T_L_VH = 1, T_L_ONE = 2,            // L  : L/dt, -Req*iPrev, (L>0 ? 1 : 0)
    T_SRC_VAL = 0,                                   // V,I: source value
    T_M_GD = 0, T_M_GG = 1, T_M_GS = 2, T_M_CST = 3, // MOS: channel linearisation
    T_M_GCH = 4, T_M_GCF = 5,                        //      (0.5*Cj0)/dt, Cj0/dt
    T_M_IHGS = 6, T_M_IHGD = 7, T_M_IHSB = 8, T_M_IHDB = 9
};

inline int termsOfKind(int kind)
{
    switch (kind) {
        case CSIM_R: return 1;
        case CSIM_C: return 2;
        case CSIM_L: return 3;
        case CSIM_V: case CSIM_I: return 1;
        case CSIM_NMOS: case CSIM_PMOS: return 10;
        default: return 0;
    }
}

// one mode's gather lists (CSR over structural non-zeros)
struct GatherPlan {
    std::vector<int32_t> gPtr, gPos, gCon;   // gPos = r*LD + c; gCon = (term<<1)|negate
    std::vector<int32_t> iPtr, iRow, iCon;
    int nnzG() const { return static_cast<int>(gPos.size()); }
    int nnzI() const { return static_cast<int>(iRow.size()); }
};

struct AssemblyPlan {
    int N = 0, LD = 0, nTerms = 0, termOne = 0, termGmin = 0;
    std::vector<int32_t> termBase;           // per element
    GatherPlan dc, tran;
    // dense structural pattern (row-major N*N, 1 = may be non-zero), per mode
    std::vector<uint8_t> patDc, patTran;
};

// leading dimension of the LDS matrix: holds N columns + the RHS column and is
// ODD so that a column walk (stride LD doubles) hits 32 distinct 8-byte banks
inline int ldFor(int N) { const int need = N + 1; return (need & 1) ? need : need + 1; }

AssemblyPlan buildAssemblyPlan(const csim_ir& ir);

// Slow-step rule of the hybrid stepping.  A damped Newton update x += alpha (x_raw - x) of a well-behaved
// iteration contracts the error by (1 - alpha) per pass, so an initial error as large as 1e3 (volts) is
// below tol after ceil(log(tol * 1e-3) / log(1 - alpha)) passes: 35 for the transient (tol 1e-6, alpha
// 0.45).  A time step that needs more is not contracting at the damping rate; such steps were measured
// to be chaotic (the 42..50-pass steps of the inverter-chain test circuit move by +-1..4 passes under
// the 1e-16 of an FMA contraction), so the generated kernels do not keep them: the step is redone by the
// bit-faithful general kernel, which also does not hand an instance back after such a step.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int slowStepIters(double tol, double alpha, int cap)
{
    if (!(alpha > 0.0 && alpha < 1.0) || !(tol > 0.0)) return cap;
    // computed with integer steps so that host (generator) and device agree exactly
    int n = 0;
    double e = 1.0e3;
    while (e >= tol && n < cap) { e *= (1.0 - alpha); ++n; }
    return n < cap ? n : cap;
}

} // namespace csim
