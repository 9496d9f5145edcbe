/*
 * csim_ir.h -- flattened circuit description ("circuit IR"), plain C POD.
 *
 * This is the only circuit representation that crosses the C-ABI, is uploaded
 * to the GPU and is broadcast between ranks.  It is produced from the C++
 * Circuit object (circuitsimulator_amd/csrc/api/circuit.hpp) by
 * csim::flatten() and consumed by
 *   - the HIP engine (circuitsimulator_amd/csrc/engine/),
 *   - the CPU oracle  (oracle/mna_oracle.c, test infrastructure only).
 *
 * What it replaces in the reference (ZyuRao/CircuitSimulator): the pointer
 * graph  Circuit::elements (vector<shared_ptr<Element>>, include/circuit.hpp:35)
 * + Node::eqIndex (include/circuit.hpp:14) + VoltageSource/Inductor
 * branchEqIndex (include/element.hpp:73,114) that every stamp() walks with
 * virtual dispatch and dynamic_pointer_cast (src/tanalisis.cpp:269-353).
 *
 * Layout rules
 *   - elements are listed in netlist order (== Circuit::elements order); every
 *     accumulation order of the reference is derived from that order.
 *   - equation index -1 means ground (Node::eqIndex == -1).
 *   - node equations are 0 .. n_node_eq-1 (creation order, ground skipped),
 *     branch equations follow, V sources and inductors interleaved in element
 *     order (src/circuit.cpp:42-61).
 *   - every per-instance value lives in the parameter vector P (doubles);
 *     an element's values start at param_slot[e]:
 *        R   : [R]
 *        C   : [C]
 *        L   : [L]
 *        V, I: NONE/SIN [dc, v0, va, freq, td, phi]      (SourceSpec, include/sim.hpp:146)
 *              PULSE    [dc, v1, v2, td, tr, tf, ton, per] (PulseSpec, include/sim.hpp:46-54)
 *              PWL      [dc, t_0..t_{n-1}, v_0..v_{n-1}], n = wave_n[e]  (PwlSpec, :64-67)
 *        MOS : [Vth, K, lambda, Cj0]            (MosfetBase, include/element.hpp:134)
 *   - batched parameter tables are slot-major:  params[p * B + b].
 */
#ifndef CSIM_IR_H
#define CSIM_IR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum csim_elem_kind {
    CSIM_R     = 0,
    CSIM_C     = 1,
    CSIM_L     = 2,
    CSIM_V     = 3,
    CSIM_I     = 4,
    CSIM_NMOS  = 5,
    CSIM_PMOS  = 6
};

/* transient waveform attached to a V/I source (include/sim.hpp:25-30, evaluated by
 * TranWaveform::eval :75-143).  The reference's parser only produces NONE and SIN
 * (src/parser.cpp:346-351); PULSE and PWL are reachable through its C++ API and are accepted
 * by this repository's netlist dialect as a superset. */
enum csim_wave_kind {
    CSIM_WAVE_NONE  = 0,
    CSIM_WAVE_PULSE = 1,
    CSIM_WAVE_SIN   = 2,
    CSIM_WAVE_PWL   = 3
};

#define CSIM_PARAMS_R     1
#define CSIM_PARAMS_C     1
#define CSIM_PARAMS_L     1
#define CSIM_PARAMS_SRC   6       /* NONE / SIN sources */
#define CSIM_PARAMS_PULSE 8
#define CSIM_PARAMS_MOS   4

/* literal constants of the hot path.  Kept in the IR
 * so that oracle and engine read the same numbers. */
typedef struct csim_consts {
    double lu_eps;            /* 1e-15   include/solver.hpp:31,122            */
    int32_t dc_ramp_steps;    /* 10      src/dcanalysis.cpp:104               */
    int32_t dc_max_iters;     /* 50      src/dcanalysis.cpp:105               */
    double dc_tol;            /* 1e-9    src/dcanalysis.cpp:106               */
    double dc_alpha;          /* 0.35    src/dcanalysis.cpp:274               */
    double dc_alpha_min;      /* 0.1     src/dcanalysis.cpp:264               */
    double dc_alpha_max;      /* 0.5                                          */
    double gmin_high;         /* 1e-6    src/dcanalysis.cpp:264               */
    double gmin_low;          /* 3.35e-7 src/dcanalysis.cpp:265               */
    double gmin_abs_max;      /* 1e-4                                         */
    double fast_ratio;        /* 0.7                                          */
    double slow_ratio;        /* 1.05                                         */
    double gmin_nonfinite_mul;/* 10      src/dcanalysis.cpp:136               */
    double gmin_nonfinite_cap;/* 1e-2                                         */
    int32_t tran_max_iters;   /* 50      src/tanalisis.cpp:241                */
    int32_t pad0;
    double tran_tol;          /* 1e-6    src/tanalisis.cpp:242                */
    double tran_gmin;         /* 1e-6    src/tanalisis.cpp:243                */
    double tran_alpha;        /* 0.45    src/tanalisis.cpp:244                */
    double mos_off_gds;       /* 1e-12   src/element.cpp:246                  */
    double pi;                /* 3.14159265358979323846  include/sim.hpp:8    */
} csim_consts;

typedef struct csim_ir {
    int32_t n_unknowns;       /* N  = n_node_eq + n_branch_eq                 */
    int32_t n_node_eq;        /* rows that receive gmin (src/tanalisis.cpp:30)*/
    int32_t n_branch_eq;
    int32_t n_elems;
    int32_t n_params;         /* P                                            */
    int32_t has_nonlinear;    /* any MOSFET -> Newton DC (src/dcanalysis.cpp:26)*/

    /* per element, length n_elems */
    const int32_t* kind;      /* csim_elem_kind                               */
    const int32_t* eq;        /* [n_elems][4]: R/C/L/V/I: {eq1|p, eq2|m,-1,-1};
                                 MOS: {eqD, eqG, eqS, eqB}                    */
    const int32_t* branch_eq; /* V, L: branch equation; others -1             */
    const int32_t* param_slot;/* first slot in P                              */
    const int32_t* wave;      /* V, I: csim_wave_kind; others 0               */
    const int32_t* wave_n;    /* PWL sources: number of (t, v) points; else 0 */

    csim_consts k;
} csim_ir;

/* fill k with the reference's literals */
static inline void csim_consts_default(csim_consts* k)
{
    k->lu_eps = 1e-15;
    k->dc_ramp_steps = 10;
    k->dc_max_iters = 50;
    k->dc_tol = 1e-9;
    k->dc_alpha = 0.35;
    k->dc_alpha_min = 0.1;
    k->dc_alpha_max = 0.5;
    k->gmin_high = 1e-6;
    k->gmin_low = 3.35e-7;
    k->gmin_abs_max = 1e-4;
    k->fast_ratio = 0.7;
    k->slow_ratio = 1.05;
    k->gmin_nonfinite_mul = 10.0;
    k->gmin_nonfinite_cap = 1e-2;
    k->tran_max_iters = 50;
    k->pad0 = 0;
    k->tran_tol = 1e-6;
    k->tran_gmin = 1e-6;
    k->tran_alpha = 0.45;
    k->mos_off_gds = 1e-12;
    k->pi = 3.14159265358979323846;
}

/* per-instance status word (replaces stderr warnings / runtime_error of the
 * reference: src/tanalisis.cpp:360-376, src/dcanalysis.cpp:135-158,
 * include/solver.hpp:58-61,94-97). */
#define CSIM_ST_TRAN_NONFINITE   0x0001u  /* LU gave NaN/Inf in TRAN: reference throws; instance stopped */
#define CSIM_ST_TRAN_NONCONV     0x0002u  /* >=1 time step hit the NR cap (WARNING in the reference)     */
#define CSIM_ST_LU_TINY_PIVOT    0x0004u  /* >=1 factorization failed -> zero solution vector            */
#define CSIM_ST_DC_NONCONV       0x0008u  /* >=1 DC ramp step hit the NR cap                             */
#define CSIM_ST_DC_NONFINITE     0x0010u  /* >=1 DC solve was non-finite (gmin bumped, iteration retried)*/
#define CSIM_ST_SCHED_FALLBACK   0x0020u  /* pre-recorded pivot schedule violated -> instance re-run dense */
#define CSIM_ST_LU_ZERO_DIAG     0x0040u  /* back-substitution met |diag|<eps -> x(i)=0 (solver.hpp:122)  */

#define CSIM_ST_SCHED_FAITHFUL   0x0100u  /* >=1 time step ran on the generated kernel with the reference's arithmetic (informational) */
#define CSIM_ST_SCHED_FALLBACK_DC 0x0080u /* same, for the DC operating point (informational: results are the general kernel's) */

#ifdef __cplusplus
}
#endif
#endif /* CSIM_IR_H */
