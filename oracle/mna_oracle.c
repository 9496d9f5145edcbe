/*
 * mna_oracle.c -- CPU oracle (plain C restatement of the reference hot path).
 * TEST INFRASTRUCTURE ONLY -- see mna_oracle.h for who may call this and for
 * the pinning status ("parity unpinned" by the reference's own tests; pinned
 * against SURVEY.md-recorded reference outputs).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  No
 * -march flags: the recorded reference numbers come from an x86-64 baseline
 * build without FMA, and every expression below keeps the reference's
 * association so the results agree bit for bit.
 *
 * Reference = ZyuRao/CircuitSimulator, paths relative to its root.
 */
#include "mna_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define P_(slot) (params[(int64_t)(slot) * pstride])

/* ------------------------------------------------------------------ *
 * pivot-sequence recorder (diagnostics for SURVEY.md Appendix F)
 * ------------------------------------------------------------------ */
#define PLOG_MAX_SEQ   64
#define PLOG_MAX_LEN   260
static int g_plog_on = 0;
static int g_plog_n = 0;
static int g_plog_len[PLOG_MAX_SEQ];
static int g_plog_seq[PLOG_MAX_SEQ][2 * PLOG_MAX_LEN];

void oracle_pivot_log_reset(int enable) { g_plog_on = enable; g_plog_n = 0; }
int  oracle_pivot_log_distinct(void) { return g_plog_n; }
int  oracle_pivot_log_get(int which, int* seq, int cap)
{
    if (which < 0 || which >= g_plog_n) return -1;
    int n = g_plog_len[which];
    for (int i = 0; i < 2 * n && i < cap; ++i) seq[i] = g_plog_seq[which][i];
    if (2 * n < cap) seq[2 * n] = -1;
    return n;
}
static void plog_record(const int* sw, int n)
{
    if (n > PLOG_MAX_LEN) n = PLOG_MAX_LEN;
    for (int s = 0; s < g_plog_n; ++s) {
        if (g_plog_len[s] == n && memcmp(g_plog_seq[s], sw, sizeof(int) * 2 * (size_t)n) == 0) return;
    }
    if (g_plog_n >= PLOG_MAX_SEQ) return;
    g_plog_len[g_plog_n] = n;
    memcpy(g_plog_seq[g_plog_n], sw, sizeof(int) * 2 * (size_t)n);
    ++g_plog_n;
}

/* ------------------------------------------------------------------ *
 * Solver::luDecompose            include/solver.hpp:30-80
 * Doolittle LU with partial (row) pivoting on a copy of A.
 * ------------------------------------------------------------------ */
int oracle_lu_decompose(int n, const double* A, double* LU, int* perm)
{
    const double eps = 1e-15;                               /* :31 */
    int sw[2 * PLOG_MAX_LEN];
    int nsw = 0;
    if (n <= 0) return 0;                                   /* :34 */

    memcpy(LU, A, sizeof(double) * (size_t)n * (size_t)n);  /* :40 */
    for (int i = 0; i < n; ++i) perm[i] = i;                /* :42-44 */

    for (int k = 0; k < n; ++k) {
        /* :48-56  first row attaining the column maximum (strict >) */
        int pivot = k;
        double maxAbs = fabs(LU[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            double val = fabs(LU[i * n + k]);
            if (val > maxAbs) { maxAbs = val; pivot = i; }
        }
        if (maxAbs < eps) return 0;                         /* :58-61 */

        if (pivot != k) {                                   /* :64-67 whole-row swap */
            for (int j = 0; j < n; ++j) {
                double t = LU[k * n + j];
                LU[k * n + j] = LU[pivot * n + j];
                LU[pivot * n + j] = t;
            }
            int tp = perm[k]; perm[k] = perm[pivot]; perm[pivot] = tp;
            if (g_plog_on && nsw < PLOG_MAX_LEN) { sw[2 * nsw] = k; sw[2 * nsw + 1] = pivot; ++nsw; }
        }

        for (int i = k + 1; i < n; ++i) {                   /* :70-76 */
            double factor = LU[i * n + k] / LU[k * n + k];
            LU[i * n + k] = factor;
            for (int j = k + 1; j < n; ++j) {
                LU[i * n + j] -= factor * LU[k * n + j];
            }
        }
    }
    if (g_plog_on) plog_record(sw, nsw);
    return 1;
}

/* ------------------------------------------------------------------ *
 * Solver::solveLinearSystemLU    include/solver.hpp:83-131
 * ------------------------------------------------------------------ */
unsigned oracle_solve_lu(int n, const double* A, const double* b, double* x)
{
    unsigned flags = 0;
    for (int i = 0; i < n; ++i) x[i] = 0.0;                 /* :85 */
    if (n <= 0) return 0;

    double* LU = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
    int* perm = (int*)malloc(sizeof(int) * (size_t)n);
    double* y = (double*)malloc(sizeof(double) * (size_t)n);

    if (!oracle_lu_decompose(n, A, LU, perm)) {             /* :94-97 zero vector */
        flags |= CSIM_ST_LU_TINY_PIVOT;
        free(LU); free(perm); free(y);
        return flags;
    }

    for (int i = 0; i < n; ++i) {                           /* :100-113 */
        double sum = b[perm[i]];
        for (int j = 0; j < i; ++j) sum -= LU[i * n + j] * y[j];
        y[i] = sum;
    }
    for (int i = n - 1; i >= 0; --i) {                      /* :116-128 */
        double sum = y[i];
        for (int j = i + 1; j < n; ++j) sum -= LU[i * n + j] * x[j];
        double diag = LU[i * n + i];
        if (fabs(diag) < 1e-15) { x[i] = 0.0; flags |= CSIM_ST_LU_ZERO_DIAG; }
        else x[i] = sum / diag;
    }
    free(LU); free(perm); free(y);
    return flags;
}

/* ------------------------------------------------------------------ *
 * sources                          include/sim.hpp:75-143, 146-163
 * ------------------------------------------------------------------ */
static double clamp01_(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }   /* utils.hpp:80-84 */

static double src_eval_dc(const csim_ir* ir, int e, const double* params, int64_t pstride, double scale)
{
    const int s = ir->param_slot[e];
    double base = P_(s + 0);                                /* dcValue */
    if (ir->wave[e] == CSIM_WAVE_SIN) base += P_(s + 1);    /* + v0   (:154-156) */
    return base * scale;
}

static double src_eval_tran(const csim_ir* ir, int e, const double* params, int64_t pstride, double t)
{
    const int s = ir->param_slot[e];
    const double dc = P_(s + 0);
    double w = 0.0;                                         /* WaveformType::NONE -> 0 (:77-78) */
    if (ir->wave[e] == CSIM_WAVE_SIN) {
        const double v0 = P_(s + 1), va = P_(s + 2), freq = P_(s + 3), td = P_(s + 4), phi = P_(s + 5);
        if (t < td) {
            w = v0;                                         /* :118 */
        } else {
            double tau = t - td;
            double om = 2.0 * ir->k.pi * freq;              /* :120 */
            w = v0 + va * sin(om * tau + phi);              /* :121 */
        }
    } else if (ir->wave[e] == CSIM_WAVE_PULSE) {            /* :80-115 */
        const double v1 = P_(s + 1), v2 = P_(s + 2), td = P_(s + 3), tr = P_(s + 4), tf = P_(s + 5);
        const double ton = P_(s + 6), per = P_(s + 7);
        if (per <= 0.0) {                                   /* single shot */
            double tau = t - td;
            if (tau <= 0.0) w = v1;
            else if (tau < tr) { double k = clamp01_(tau / tr); w = v1 + k * (v2 - v1); }
            else if (tau < tr + ton) w = v2;
            else { double tfall = tau - (tr + ton); double k = clamp01_(tfall / tf); w = v2 + k * (v1 - v2); }
        } else if (t < td) {
            w = v1;
        } else {                                            /* periodic */
            double tau = fmod(t - td, per);
            if (tau < 0.0) tau += per;
            if (tau < tr) { double k = clamp01_(tau / tr); w = v1 + (v2 - v1) * k; }
            else if (tau < tr + ton) w = v2;
            else if (tau < tr + ton + tf) { double tfall = tau - (tr + ton); double k = clamp01_(tfall / tf); w = v2 + (v1 - v2) * k; }
            else w = v1;
        }
    } else if (ir->wave[e] == CSIM_WAVE_PWL) {              /* :124-138 */
        const int n = ir->wave_n[e];
        #define PT_(i) P_(s + 1 + (i))
        #define PV_(i) P_(s + 1 + n + (i))
        if (n <= 0) w = 0.0;
        else if (t <= PT_(0)) w = PV_(0);
        else if (t >= PT_(n - 1)) w = PV_(n - 1);
        else {
            w = PV_(n - 1);
            for (int i = 0; i + 1 < n; ++i)
                if (t > PT_(i) && t <= PT_(i + 1)) {
                    double k = (t - PT_(i)) / (PT_(i + 1) - PT_(i));
                    w = PV_(i) + (PV_(i + 1) - PV_(i)) * k;
                    break;
                }
        }
        #undef PT_
        #undef PV_
    }
    return dc + w;                                          /* :161 */
}

/* ------------------------------------------------------------------ *
 * device stamps                    src/element.cpp
 * ------------------------------------------------------------------ */
#define G_(r, c) G[(r) * N + (c)]

/* Resistor::stamp  element.cpp:9-32 */
static void stamp_resistor(int N, int eq1, int eq2, double R, double* G)
{
    if (R == 0.0) return;                                   /* :20-23 warn + skip */
    double g = 1.0 / R;
    if (eq1 >= 0) G_(eq1, eq1) += g;
    if (eq2 >= 0) G_(eq2, eq2) += g;
    if (eq1 >= 0 && eq2 >= 0) {
        G_(eq1, eq2) -= g;
        G_(eq2, eq1) -= g;
    }
}

/* CurrentSource::stamp  element.cpp:34-66 */
static void stamp_isource(int eqP, int eqM, double Ival, double* I)
{
    if (eqP >= 0) I[eqP] -= Ival;
    if (eqM >= 0) I[eqM] += Ival;
}

/* VoltageSource::stamp  element.cpp:83-123 */
static void stamp_vsource(int N, int eqP, int eqM, int k, double Vval, double* G, double* I)
{
    if (k < 0 || k >= N) return;                            /* :94-97 */
    if (eqP >= 0) G_(eqP, k) += 1.0;
    if (eqM >= 0) G_(eqM, k) -= 1.0;
    if (eqP >= 0) G_(k, eqP) += 1.0;
    if (eqM >= 0) G_(k, eqM) -= 1.0;
    I[k] += Vval;
}

/* Inductor::stamp (DC: a 0 V source)  element.cpp:156-178 */
static void stamp_inductor_dc(int N, int eqP, int eqM, int k, double* G)
{
    if (k < 0 || k >= N) return;
    if (eqP >= 0) G_(eqP, k) += 1.0;
    if (eqM >= 0) G_(eqM, k) -= 1.0;
    if (eqP >= 0) G_(k, eqP) += 1.0;
    if (eqM >= 0) G_(k, eqM) -= 1.0;
}

/* MosfetBase::stamp  element.cpp:181-307  (Level-1, bulk ignored, no D/S swap) */
static void stamp_mosfet(int N, int isP, int eqD, int eqG, int eqS,
                         double Vth, double K, double lambda, double off_gds,
                         const double* x, double* G, double* I)
{
    double Vd = (eqD >= 0 && eqD < N) ? x[eqD] : 0.0;       /* :196-203 */
    double Vg = (eqG >= 0 && eqG < N) ? x[eqG] : 0.0;
    double Vs = (eqS >= 0 && eqS < N) ? x[eqS] : 0.0;

    double p = isP ? -1.0 : 1.0;                            /* :207 */
    double Vgs_eff = p * (Vg - Vs);                         /* :210-211 */
    double Vds_eff = p * (Vd - Vs);

    double Ids0 = 0.0, gds0 = 0.0, gm0 = 0.0;
    int on = 0;
    if (Vgs_eff > Vth && Vds_eff >= 0) {                    /* :223 */
        on = 1;
        double Vov = Vgs_eff - Vth;
        if (Vds_eff < Vov) {                                /* triode :232-236 */
            Ids0 = K * (Vov * Vds_eff - 0.5 * Vds_eff * Vds_eff);
            gds0 = K * (Vov - Vds_eff);
            gm0  = K * Vds_eff;
        } else {                                            /* saturation :239-241 */
            Ids0 = 0.5 * K * Vov * Vov;
            gds0 = 0.0;
            gm0  = K * Vov;
        }
    }
    if (!on) {                                              /* :245-252 */
        Ids0 = 0.0;
        gm0 = 0.0;
        gds0 = off_gds;
    }

    double factor = 1.0 + lambda * Vds_eff;                 /* :255-256 */
    if (factor < 0.0) factor = 0.0;
    double Ids_eff = Ids0 * factor;                         /* :257 */
    double dId_dVds_eff = gds0 * factor + Ids0 * lambda;    /* :260 */
    double dId_dVgs_eff = gm0 * factor;                     /* :263 */

    double Ids = p * Ids_eff;                               /* :266 */
    double gd = dId_dVds_eff;                               /* :269-271 */
    double gg = dId_dVgs_eff;
    double gs = -(dId_dVds_eff + dId_dVgs_eff);
    double cst = Ids - gd * Vd - gg * Vg - gs * Vs;         /* :274 */

    if (eqD >= 0) {                                         /* :290-295 */
        G_(eqD, eqD) += gd;
        if (eqG >= 0) G_(eqD, eqG) += gg;
        if (eqS >= 0) G_(eqD, eqS) += gs;
        I[eqD] -= cst;
    }
    if (eqS >= 0) {                                         /* :299-304 */
        if (eqD >= 0) G_(eqS, eqD) += -gd;
        if (eqG >= 0) G_(eqS, eqG) += -gg;
        G_(eqS, eqS) += -gs;
        I[eqS] += cst;
    }
}

/* stampCapBE  src/tanalisis.cpp:59-80 */
static void stamp_cap_be(int N, int eq1, int eq2, double C, double dt, double vPrev, double* G, double* I)
{
    if (C <= 0.0 || dt <= 0.0) return;                      /* :65 */
    double Gc = C / dt;
    if (eq1 >= 0) G_(eq1, eq1) += Gc;
    if (eq2 >= 0) G_(eq2, eq2) += Gc;
    if (eq1 >= 0 && eq2 >= 0) {
        G_(eq1, eq2) -= Gc;
        G_(eq2, eq1) -= Gc;
    }
    double I_hist = -Gc * vPrev;                            /* :77 */
    if (eq1 >= 0) I[eq1] -= I_hist;
    if (eq2 >= 0) I[eq2] += I_hist;
}

/* stampGlobalGmin  src/tanalisis.cpp:30-41, src/dcanalysis.cpp:36-43.
 * Node equations are exactly rows 0..n_node_eq-1 (src/circuit.cpp:46-52). */
static void stamp_gmin(const csim_ir* ir, double gmin, double* G)
{
    const int N = ir->n_unknowns;
    for (int eq = 0; eq < ir->n_node_eq; ++eq) G_(eq, eq) += gmin;
}

static double volt(const double* x, int N, int eq)            /* getNodeVoltage tanalisis.cpp:20-27 */
{
    return (eq >= 0 && eq < N) ? x[eq] : 0.0;
}

/* ------------------------------------------------------------------ *
 * one DC system: every element's stamp() in netlist order, then gmin
 * src/dcanalysis.cpp:120-130 (Newton) / :55-63 (linear, no gmin)
 * ------------------------------------------------------------------ */
void oracle_stamp_dc(const csim_ir* ir, const double* params, int64_t pstride,
                     const double* x, double scale, double gmin, double* G, double* I)
{
    const int N = ir->n_unknowns;
    memset(G, 0, sizeof(double) * (size_t)N * (size_t)N);
    memset(I, 0, sizeof(double) * (size_t)N);
    for (int e = 0; e < ir->n_elems; ++e) {
        const int* q = ir->eq + 4 * e;
        const int s = ir->param_slot[e];
        switch (ir->kind[e]) {
            case CSIM_R: stamp_resistor(N, q[0], q[1], P_(s), G); break;
            case CSIM_C: break;                               /* open at DC, element.hpp:103-108 */
            case CSIM_L: stamp_inductor_dc(N, q[0], q[1], ir->branch_eq[e], G); break;
            case CSIM_V: stamp_vsource(N, q[0], q[1], ir->branch_eq[e],
                                       src_eval_dc(ir, e, params, pstride, scale), G, I); break;
            case CSIM_I: stamp_isource(q[0], q[1], src_eval_dc(ir, e, params, pstride, scale), I); break;
            case CSIM_NMOS: case CSIM_PMOS:
                stamp_mosfet(N, ir->kind[e] == CSIM_PMOS, q[0], q[1], q[2],
                             P_(s + 0), P_(s + 1), P_(s + 2), ir->k.mos_off_gds, x, G, I);
                break;
            default: break;
        }
    }
    if (gmin >= 0.0) stamp_gmin(ir, gmin, G);
}

/* ------------------------------------------------------------------ *
 * one transient system            src/tanalisis.cpp:259-356
 * ------------------------------------------------------------------ */
void oracle_stamp_tran(const csim_ir* ir, const double* params, int64_t pstride,
                       const double* x, const double* xprev, double tnow, double dt,
                       double* G, double* I)
{
    const int N = ir->n_unknowns;
    memset(G, 0, sizeof(double) * (size_t)N * (size_t)N);
    memset(I, 0, sizeof(double) * (size_t)N);

    /* 1) everything that is not C / L / MOS, in element order (:269-274) */
    for (int e = 0; e < ir->n_elems; ++e) {
        const int* q = ir->eq + 4 * e;
        const int s = ir->param_slot[e];
        switch (ir->kind[e]) {
            case CSIM_R: stamp_resistor(N, q[0], q[1], P_(s), G); break;
            case CSIM_V: stamp_vsource(N, q[0], q[1], ir->branch_eq[e],
                                       src_eval_tran(ir, e, params, pstride, tnow), G, I); break;
            case CSIM_I: stamp_isource(q[0], q[1], src_eval_tran(ir, e, params, pstride, tnow), I); break;
            default: break;
        }
    }
    /* 2) MOS channel linearisation (:277-279) */
    for (int e = 0; e < ir->n_elems; ++e) {
        if (ir->kind[e] != CSIM_NMOS && ir->kind[e] != CSIM_PMOS) continue;
        const int* q = ir->eq + 4 * e;
        const int s = ir->param_slot[e];
        stamp_mosfet(N, ir->kind[e] == CSIM_PMOS, q[0], q[1], q[2],
                     P_(s + 0), P_(s + 1), P_(s + 2), ir->k.mos_off_gds, x, G, I);
    }
    /* 3) explicit capacitors, BE companion (:282-291) */
    for (int e = 0; e < ir->n_elems; ++e) {
        if (ir->kind[e] != CSIM_C) continue;
        const int* q = ir->eq + 4 * e;
        double vPrev = volt(xprev, N, q[0]) - volt(xprev, N, q[1]);   /* :381-388 */
        stamp_cap_be(N, q[0], q[1], P_(ir->param_slot[e]), dt, vPrev, G, I);
    }
    /* 4) inductors, Thevenin BE companion (:294-319) */
    for (int e = 0; e < ir->n_elems; ++e) {
        if (ir->kind[e] != CSIM_L) continue;
        const int* q = ir->eq + 4 * e;
        double Lval = P_(ir->param_slot[e]);
        if (Lval <= 0.0) continue;                          /* :296 */
        int k = ir->branch_eq[e];
        if (k < 0 || k >= N) continue;                      /* :304 */
        double R_eq = Lval / dt;
        double iPrev = xprev[k];                            /* :390-397 */
        double V_hist = -R_eq * iPrev;
        int eqP = q[0], eqM = q[1];
        if (eqP >= 0) G_(eqP, k) += 1.0;
        if (eqM >= 0) G_(eqM, k) -= 1.0;
        if (eqP >= 0) G_(k, eqP) += 1.0;
        if (eqM >= 0) G_(k, eqM) -= 1.0;
        G_(k, k) += -R_eq;
        I[k] += V_hist;
    }
    /* 5) MOS parasitic capacitors (:322-353) */
    for (int e = 0; e < ir->n_elems; ++e) {
        if (ir->kind[e] != CSIM_NMOS && ir->kind[e] != CSIM_PMOS) continue;
        const int* q = ir->eq + 4 * e;
        int eqD = q[0], eqG = q[1], eqS = q[2], eqB = q[3];
        double Cj0 = P_(ir->param_slot[e] + 3);
        double Cgs = 0.5 * Cj0, Cgd = 0.5 * Cj0, CsJ = Cj0, CdJ = Cj0;   /* :337-341 */
        double vD = volt(xprev, N, eqD), vG = volt(xprev, N, eqG);
        double vS = volt(xprev, N, eqS), vB = volt(xprev, N, eqB);
        stamp_cap_be(N, eqG, eqS, Cgs, dt, vG - vS, G, I);  /* :346 */
        stamp_cap_be(N, eqG, eqD, Cgd, dt, vG - vD, G, I);  /* :348 */
        stamp_cap_be(N, eqS, eqB, CsJ, dt, vS - vB, G, I);  /* :350 */
        stamp_cap_be(N, eqD, eqB, CdJ, dt, vD - vB, G, I);  /* :352 */
    }
    /* 6) gmin to ground on every node row (:356) */
    stamp_gmin(ir, ir->k.tran_gmin, G);
}

static int all_finite(const double* v, int n)
{
    for (int i = 0; i < n; ++i) if (!isfinite(v[i])) return 0;
    return 1;
}

/* xNew = x + alpha*(xRaw - x); err = ||xNew - x||_2 (sum in index order) */
static double damped_update(int n, double alpha, const double* x, const double* xRaw, double* xNew)
{
    double ss = 0.0;
    for (int i = 0; i < n; ++i) xNew[i] = x[i] + alpha * (xRaw[i] - x[i]);
    for (int i = 0; i < n; ++i) { double d = xNew[i] - x[i]; ss += d * d; }
    return sqrt(ss);
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ConvController::baseGmin  include/dcanalysis.hpp:45-48 */
static double base_gmin(const csim_consts* k, double s)
{
    s = clampd(s, 0.0, 1.0);
    return k->gmin_high * (1.0 - s) + k->gmin_low * s;
}

/* ------------------------------------------------------------------ *
 * DC operating point   src/dcanalysis.cpp:242-262, :46-68, :95-163, :268-307
 * ------------------------------------------------------------------ */
int oracle_dc(const csim_ir* ir, const double* params, int64_t pstride,
              double* x, int32_t* iters, uint32_t* status)
{
    const int N = ir->n_unknowns;
    const csim_consts* k = &ir->k;
    int32_t it_total = 0;
    uint32_t st = 0;
    if (iters) *iters = 0;
    if (status) *status = 0;
    if (N <= 0) return -1;

    double* G = (double*)malloc(sizeof(double) * (size_t)N * (size_t)N);
    double* I = (double*)malloc(sizeof(double) * (size_t)N);
    double* xRaw = (double*)malloc(sizeof(double) * (size_t)N);
    double* xNew = (double*)malloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; ++i) x[i] = 0.0;

    if (!ir->has_nonlinear) {
        /* dcSolveDirectLU :46-68 -- one solve, x = 0, scale = 1, NO gmin */
        oracle_stamp_dc(ir, params, pstride, x, 1.0, -1.0, G, I);
        st |= oracle_solve_lu(N, G, I, xRaw);
        for (int i = 0; i < N; ++i) x[i] = xRaw[i];
        it_total = 1;
    } else {
        /* dcSolveNewtonLU :95-163 */
        for (int step = 1; step <= k->dc_ramp_steps; ++step) {
            double scale = (double)step / k->dc_ramp_steps;              /* :113 */
            double gmin = base_gmin(k, scale);                           /* :116 */
            double prevErr = INFINITY;                                   /* :117 */
            for (int iter = 0; iter < k->dc_max_iters; ++iter) {
                oracle_stamp_dc(ir, params, pstride, x, scale, gmin, G, I);   /* :120-130 */
                st |= oracle_solve_lu(N, G, I, xRaw);                    /* :134 */
                ++it_total;
                if (!all_finite(xRaw, N)) {                              /* :135-138 */
                    double g10 = gmin * k->gmin_nonfinite_mul;
                    gmin = g10 < k->gmin_nonfinite_cap ? g10 : k->gmin_nonfinite_cap;
                    st |= CSIM_ST_DC_NONFINITE;
                    continue;
                }
                /* ConvController::update :268-307 */
                double alpha = clampd(k->dc_alpha, k->dc_alpha_min, k->dc_alpha_max);   /* :274 */
                double err = damped_update(N, alpha, x, xRaw, xNew);     /* :275-276 */
                double gminBase = base_gmin(k, scale);
                double gminNext = gminBase;
                if (iter == 0 || !isfinite(prevErr)) {                   /* :280-282 */
                    gminNext = gminBase;
                } else if (err > prevErr * k->slow_ratio) {              /* :285-288 */
                    double g2 = gmin * 2.0;
                    gminNext = g2 < k->gmin_abs_max ? g2 : k->gmin_abs_max;
                } else if (err < prevErr * k->fast_ratio) {              /* :289-293 */
                    gminNext = 0.5 * gmin + 0.5 * gminBase;
                } else {                                                 /* :296 */
                    gminNext = 0.7 * gmin + 0.3 * gminBase;
                }
                for (int i = 0; i < N; ++i) x[i] = xNew[i];              /* :145 */
                gmin = gminNext;                                         /* :147 */
                prevErr = err;                                           /* :148 */
                if (err < k->dc_tol) break;                              /* :150, :304 */
                if (iter == k->dc_max_iters - 1) st |= CSIM_ST_DC_NONCONV;   /* :153-158 */
            }
        }
    }
    free(G); free(I); free(xRaw); free(xNew);
    if (iters) *iters = it_total;
    if (status) *status = st;
    return 0;
}

/* ------------------------------------------------------------------ *
 * Solver::solveLinearSystemGaussSeidel    include/solver.hpp:139-204
 * x0 == NULL: start from the zero vector (the two-argument overload, :197-204).
 * Returns the number of sweeps performed.
 * ------------------------------------------------------------------ */
int oracle_solve_gs(int n, const double* A, const double* b, const double* x0, int max_iters, double tol,
                    double* x)
{
    if (n <= 0) return 0;                                    /* :146 */
    for (int i = 0; i < n; ++i) x[i] = x0 ? x0[i] : 0.0;     /* :145, :154-157 */
    double* xOld = (double*)malloc(sizeof(double) * (size_t)n);
    const double diagEps = 1e-12;                            /* :160 */
    int sweeps = 0;
    for (int iter = 0; iter < max_iters; ++iter) {           /* :162 */
        ++sweeps;
        for (int i = 0; i < n; ++i) xOld[i] = x[i];          /* :163 */
        for (int i = 0; i < n; ++i) {
            double diag = A[i * n + i];                      /* :166 */
            if (fabs(diag) < diagEps) {                      /* :169-173: keep the sign, default positive */
                double sign = (diag >= 0.0 ? 1.0 : -1.0);
                diag = sign * diagEps;
            }
            double sum = b[i];                               /* :175 */
            for (int j = 0; j < i; ++j) sum -= A[i * n + j] * x[j];          /* :178-180 newest values */
            for (int j = i + 1; j < n; ++j) sum -= A[i * n + j] * xOld[j];   /* :181-183 previous sweep */
            x[i] = sum / diag;                               /* :185 */
        }
        double ss = 0.0;                                     /* :188 (x - xOld).norm(), summed in index order */
        for (int i = 0; i < n; ++i) { double d = x[i] - xOld[i]; ss += d * d; }
        if (sqrt(ss) < tol) break;                           /* :189-192 */
    }
    free(xOld);
    return sweeps;
}

/* ------------------------------------------------------------------ *
 * dcSolveGaussSeidel   src/dcanalysis.cpp:254-258 (dispatch), :71-92 (linear: one
 * Gauss-Seidel solve of the system stamped at x = 0, scale 1, no gmin, 2000 sweeps,
 * tol 1e-10), :166-237 (Newton ramp: 10 steps, 60 passes per step and 120 in the last,
 * inner solve warm-started from x, ConvController::update).
 * iters = passes through the Newton loop body (1 for the linear path).
 * ------------------------------------------------------------------ */
int oracle_dc_gs(const csim_ir* ir, const double* params, int64_t pstride,
                 double* x, int32_t* iters, uint32_t* status)
{
    const int N = ir->n_unknowns;
    const csim_consts* k = &ir->k;
    int32_t it_total = 0;
    uint32_t st = 0;
    if (iters) *iters = 0;
    if (status) *status = 0;
    if (N <= 0) return -1;
    const int gsSweeps = 2000;                               /* :88, :198 */
    const double gsTol = 1e-10;
    const int maxNewton = 60;                                /* :176 */
    const double tol = 1e-9;                                 /* :177 */

    double* G = (double*)malloc(sizeof(double) * (size_t)N * (size_t)N);
    double* I = (double*)malloc(sizeof(double) * (size_t)N);
    double* xRaw = (double*)malloc(sizeof(double) * (size_t)N);
    double* xNew = (double*)malloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; ++i) x[i] = 0.0;

    if (!ir->has_nonlinear) {
        oracle_stamp_dc(ir, params, pstride, x, 1.0, -1.0, G, I);            /* :81-86 */
        oracle_solve_gs(N, G, I, NULL, gsSweeps, gsTol, xRaw);               /* :89 */
        for (int i = 0; i < N; ++i) x[i] = xRaw[i];
        it_total = 1;
    } else {
        for (int step = 1; step <= k->dc_ramp_steps; ++step) {               /* :183 */
            double scale = (double)step / k->dc_ramp_steps;
            double gmin = base_gmin(k, scale);                               /* :186 */
            double prevErr = INFINITY;
            int maxIterThisStep = maxNewton;
            if (step == k->dc_ramp_steps) maxIterThisStep = maxNewton * 2;   /* :189-191 */
            for (int iter = 0; iter < maxIterThisStep; ++iter) {
                oracle_stamp_dc(ir, params, pstride, x, scale, gmin, G, I);  /* :193-203 */
                oracle_solve_gs(N, G, I, x, gsSweeps, gsTol, xRaw);          /* :206-207 warm start */
                ++it_total;
                if (!all_finite(xRaw, N)) {                                  /* :209-215 */
                    double g10 = gmin * 10.0;
                    gmin = g10 < 1e-2 ? g10 : 1e-2;
                    st |= CSIM_ST_DC_NONFINITE;
                    continue;
                }
                double alpha = clampd(k->dc_alpha, k->dc_alpha_min, k->dc_alpha_max);   /* update() ignores alphaCurrent, :274 */
                double err = damped_update(N, alpha, x, xRaw, xNew);
                double gminBase = base_gmin(k, scale);
                double gminNext = gminBase;
                if (iter == 0 || !isfinite(prevErr)) gminNext = gminBase;
                else if (err > prevErr * k->slow_ratio) { double g2 = gmin * 2.0; gminNext = g2 < k->gmin_abs_max ? g2 : k->gmin_abs_max; }
                else if (err < prevErr * k->fast_ratio) gminNext = 0.5 * gmin + 0.5 * gminBase;
                else gminNext = 0.7 * gmin + 0.3 * gminBase;
                for (int i = 0; i < N; ++i) x[i] = xNew[i];                  /* :220 */
                gmin = gminNext;
                prevErr = err;
                if (err < tol) break;                                        /* :225-227 */
                if (iter == maxNewton - 1) st |= CSIM_ST_DC_NONCONV;         /* :228-233: warns at pass 60 also in the last step */
            }
        }
    }
    free(G); free(I); free(xRaw); free(xNew);
    if (iters) *iters = it_total;
    if (status) *status = st;
    return 0;
}

int64_t oracle_tran_num_steps(double tstep, double tstop)
{
    if (!(tstep > 0.0) || !(tstop > 0.0)) return -1;        /* rejected at tanalisis.cpp:94-97 */
    return (int64_t)(int)floor(tstop / tstep + 1e-12);      /* tanalisis.cpp:238 */
}

/* ------------------------------------------------------------------ *
 * runTransientAnalysisBackwardEuler    src/tanalisis.cpp:83-424
 * ------------------------------------------------------------------ */
int64_t oracle_tran(const csim_ir* ir, const double* params, int64_t pstride,
                    double tstep, double tstop, double tstart, const double* x0,
                    double* rows, int64_t max_rows, int64_t* n_rows,
                    double* x_final, int64_t* iters, int32_t* iters_per_step,
                    uint32_t* status)
{
    const int N = ir->n_unknowns;
    const csim_consts* k = &ir->k;
    int64_t it_total = 0, nr = 0;
    uint32_t st = 0;
    if (n_rows) *n_rows = 0;
    if (iters) *iters = 0;
    if (status) *status = 0;
    if (tstep <= 0.0 || tstop <= 0.0) return -2;            /* :94-97 */
    if (N <= 0) return -1;                                  /* :103-107 */

    const double dt = tstep;
    double* G = (double*)malloc(sizeof(double) * (size_t)N * (size_t)N);
    double* I = (double*)malloc(sizeof(double) * (size_t)N);
    double* x = (double*)malloc(sizeof(double) * (size_t)N);
    double* xprev = (double*)malloc(sizeof(double) * (size_t)N);
    double* xRaw = (double*)malloc(sizeof(double) * (size_t)N);
    double* xNew = (double*)malloc(sizeof(double) * (size_t)N);

    if (x0) {
        memcpy(x, x0, sizeof(double) * (size_t)N);
    } else {
        uint32_t dst = 0;
        oracle_dc(ir, params, pstride, x, NULL, &dst);      /* :112 */
        st |= dst;
    }
    memcpy(xprev, x, sizeof(double) * (size_t)N);           /* histories from xdc :139-180 */

#define DUMP_ROW(t_, v_) do {                                            \
        if (!((t_) < tstart)) {                                          \
            if (rows && nr < max_rows) {                                 \
                rows[nr * (1 + N)] = (t_);                               \
                memcpy(rows + nr * (1 + N) + 1, (v_), sizeof(double) * (size_t)N); \
            }                                                            \
            ++nr;                                                        \
        } } while (0)

    DUMP_ROW(0.0, x);                                       /* :250 */

    const int64_t nSteps = oracle_tran_num_steps(tstep, tstop);
    int aborted = 0;
    for (int64_t step = 0; step < nSteps && !aborted; ++step) {
        double tNow = (double)(int)(step + 1) * dt;         /* :256 (int * double) */
        int32_t it_step = 0;
        for (int iter = 0; iter < k->tran_max_iters; ++iter) {
            oracle_stamp_tran(ir, params, pstride, x, xprev, tNow, dt, G, I);   /* :259-356 */
            st |= oracle_solve_lu(N, G, I, xRaw);           /* :359 */
            ++it_step;
            if (!all_finite(xRaw, N)) {                     /* :360-362 throws */
                st |= CSIM_ST_TRAN_NONFINITE;
                aborted = 1;
                break;
            }
            double err = damped_update(N, k->tran_alpha, x, xRaw, xNew);   /* :365-366 */
            memcpy(x, xNew, sizeof(double) * (size_t)N);    /* :367 */
            if (err < k->tran_tol) break;                   /* :369-371 */
            if (iter == k->tran_max_iters - 1) st |= CSIM_ST_TRAN_NONCONV;   /* :372-376 */
        }
        it_total += it_step;
        if (iters_per_step) iters_per_step[step] = it_step;
        if (aborted) break;
        memcpy(xprev, x, sizeof(double) * (size_t)N);       /* :381-417 histories */
        DUMP_ROW(tNow, x);                                  /* :419 */
    }
#undef DUMP_ROW

    if (x_final) memcpy(x_final, x, sizeof(double) * (size_t)N);
    if (iters) *iters = it_total;
    if (status) *status = st;
    if (n_rows) *n_rows = nr;
    free(G); free(I); free(x); free(xprev); free(xRaw); free(xNew);
    return nSteps;
}

/* CSV body in the reference's format: std::scientific, setprecision(9)
 * (tanalisis.cpp:189), comma separated, '\n' terminated (:211-230). */
int oracle_write_csv_rows(const char* path, const char* header, const double* rows,
                          int64_t n_rows, int n_cols)
{
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    if (header) fprintf(f, "%s\n", header);
    for (int64_t r = 0; r < n_rows; ++r) {
        for (int c = 0; c < n_cols; ++c) {
            if (c) fputc(',', f);
            fprintf(f, "%.9e", rows[r * n_cols + c]);
        }
        fputc('\n', f);
    }
    fclose(f);
    return 0;
}

/* ------------------------------------------------------------------ *
 * Host-threaded driver for the CPU baseline of bench.py ("OpenMP over instances on all
 * cores"): n_inst instances of a slot-major table run oracle_tran() concurrently on n_threads POSIX threads,
 * instances handed out one at a time from a shared counter.  Not part of the restatement proper.
 * ------------------------------------------------------------------ */
typedef struct {
    const csim_ir* ir; const double* params; int64_t pstride; int b_first, n_inst;
    double tstep, tstop; int64_t* iters_out; int next; pthread_mutex_t lock;
} batch_job;

static void* batch_worker(void* arg)
{
    batch_job* j = (batch_job*)arg;
    for (;;) {
        pthread_mutex_lock(&j->lock);
        const int i = j->next < j->n_inst ? j->next++ : -1;
        pthread_mutex_unlock(&j->lock);
        if (i < 0) return NULL;
        int64_t it = 0;
        uint32_t st = 0;
        oracle_tran(j->ir, j->params + (j->b_first + i), j->pstride, j->tstep, j->tstop, 0.0, NULL, NULL, 0, NULL, NULL,
                    &it, NULL, &st);
        j->iters_out[i] = it;
    }
}

int oracle_tran_batch_mt(const csim_ir* ir, const double* params, int64_t pstride, int b_first, int n_inst,
                         double tstep, double tstop, int n_threads, int64_t* iters_out)
{
    if (!ir || !params || !iters_out || n_inst < 0 || n_threads < 1) return -1;
    batch_job j = {ir, params, pstride, b_first, n_inst, tstep, tstop, iters_out, 0, PTHREAD_MUTEX_INITIALIZER};
    if (n_threads > 1024) n_threads = 1024;
    pthread_t th[1024];
    int started = 0;
    for (int t = 0; t < n_threads; ++t)
        if (pthread_create(&th[started], NULL, batch_worker, &j) == 0) ++started;
    if (started == 0) batch_worker(&j);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    return started;
}
