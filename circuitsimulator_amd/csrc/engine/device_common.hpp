// device_common.hpp -- device-side building blocks of the wave-per-instance
// ("general") kernels: element terms, gather assembly, and the
// wavefront-cooperative pivoted LU on an LDS-resident augmented matrix.
//
// One wavefront (64 lanes, one workgroup) owns one circuit instance.  The
// dense system [G | I] lives in LDS, row-major with an ODD leading dimension
// LD >= N+1 (8-byte words): a row walk is unit-stride and a column walk has
// stride LD, both conflict-free over 32 banks of 8 bytes.
//
// Reference arithmetic restated here (ZyuRao/CircuitSimulator):
//   mos_eval()        MosfetBase::stamp          src/element.cpp:181-307
//   source_value_*()  SourceSpec::evalDC/evalTran, TranWaveform::eval
//                                                include/sim.hpp:117-122,146-163
//   lu_solve_wave()   Solver::luDecompose + solveLinearSystemLU
//                                                include/solver.hpp:30-131
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csim_ir.h"
#include "plan.hpp"

namespace csim {

// device view of the circuit + one mode's gather lists (device pointers)
struct GenPlan {
    int N, LD, nNodeEq, nElem, P, nTerms, termOne, termGmin;
    int nnzG, nnzI, hasNonlinear, nConG, nConI, pad;      // nCon*: lengths of gCon / iCon
    const int32_t *kind, *eq, *branch, *slot, *wave, *waveN, *termBase;
    const int32_t *gPtr, *gPos, *gCon;
    const int32_t *iPtr, *iRow, *iCon;
    csim_consts k;
};

// LDS carve-up (in doubles) for one instance
struct LdsLayout {
    int G, T, P, xs, xp, sc, piv, total;
};
__host__ __device__ inline LdsLayout ldsLayout(int N, int LD, int nTerms, int P)
{
    LdsLayout l;
    l.G = 0;
    l.T = l.G + N * LD;
    l.P = l.T + nTerms;
    l.xs = l.P + P;
    l.xp = l.xs + N;
    l.sc = l.G;                 // norm scratch aliases the matrix (dead after the solve)
    l.piv = l.xp + N;           // N int32: pivot sequence of the current factorisation
    l.total = l.piv + (N + 1) / 2;
    return l;
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// The general kernels are the FAITHFUL path (planner, DC, fallback): keep every
// multiply and add separately rounded, like the reference's x86-64 build,
// instead of letting hipcc contract them into FMAs.  Measured on buffer.sp:
// worst deviation from the oracle 6.7e-10 -> 4.9e-11 (of a 1e-9 A floor).
#pragma clang fp contract(off)

// ---- the plan's index arrays, staged in LDS.  Every Newton iteration walks them (element records in the
// MOSFET pass, gather lists in the assembly) in short dependent chains: pointer, list entry, term.  Only the
// packed kernels (kernels_packed.hip) stage them -- they run one wave per SIMD whatever their LDS need, and gain
// 5-8 % (buffer.sp).  The wave-per-instance kernels hide those loads behind their other waves and LOSE
// throughput when the copies cost occupancy (dbmixer, B = 4096: 6.7e7 -> 4.0e7 NR-iter*inst/s); alone on a SIMD
// (B = 64) they run at 42 us per iteration either way, so they read the arrays from global memory.
constexpr size_t kStagedLdsLimit = 64 * 1024;      // a launch whose LDS need with the staged arrays exceeds this reads them from global memory
__host__ __device__ inline int planLdsInts(const GenPlan& pl)
{
    return 10 * pl.nElem + (pl.nnzG + 1) + pl.nnzG + pl.nConG + (pl.nnzI + 1) + pl.nnzI + pl.nConI;
}
__device__ __forceinline__ GenPlan plan_in_lds(const GenPlan& pl, int32_t* dst, int lane, int stride)
{
    GenPlan o = pl;
    int32_t* w = dst;
    auto put = [&](const int32_t* src, int n) {
        for (int i = lane; i < n; i += stride) w[i] = src[i];
        const int32_t* at = w;
        w += n;
        return at;
    };
    o.kind = put(pl.kind, pl.nElem);
    o.eq = put(pl.eq, 4 * pl.nElem);
    o.branch = put(pl.branch, pl.nElem);
    o.slot = put(pl.slot, pl.nElem);
    o.wave = put(pl.wave, pl.nElem);
    o.waveN = put(pl.waveN, pl.nElem);
    o.termBase = put(pl.termBase, pl.nElem);
    o.gPtr = put(pl.gPtr, pl.nnzG + 1);
    o.gPos = put(pl.gPos, pl.nnzG);
    o.gCon = put(pl.gCon, pl.nConG);
    o.iPtr = put(pl.iPtr, pl.nnzI + 1);
    o.iRow = put(pl.iRow, pl.nnzI);
    o.iCon = put(pl.iCon, pl.nConI);
    return o;                                 // the caller synchronises before the first use
}

// one wavefront per workgroup: LDS traffic of the wave is ordered by issue, so
// this is a wait on the LDS queue plus a scheduling fence (the s_barrier of a
// single-wave workgroup is elided by the compiler)
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// broadcast a double from a wave-uniform lane index through SGPRs
__device__ __forceinline__ double read_lane(double v, int srcLane)
{
    const int l = __builtin_amdgcn_readfirstlane(srcLane);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

__device__ __forceinline__ double volt_of(const double* x, int eq) { return eq >= 0 ? x[eq] : 0.0; }

// SourceSpec::evalDC (sim.hpp:152-158)
__device__ __forceinline__ double source_value_dc(const double* Pv, int slot, int wave, double scale)
{
    double base = Pv[slot + 0];
    if (wave == CSIM_WAVE_SIN) base += Pv[slot + 1];
    return base * scale;
}

__device__ __forceinline__ double clamp01_dev(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// SourceSpec::evalTran (sim.hpp:160-162) with TranWaveform::eval (:75-143): SIN, PULSE, PWL.
// PS = stride between consecutive parameter slots (1 for an LDS copy, B for the global table).
template <typename PGet>
__device__ __forceinline__ double source_wave_value(PGet P, int wave, int waveN, double t, double pi)
{
    double w = 0.0;
    if (wave == CSIM_WAVE_SIN) {
        const double v0 = P(1), va = P(2), freq = P(3);
        const double td = P(4), phi = P(5);
        if (t < td) w = v0;
        else {
            const double tau = t - td;
            const double om = 2.0 * pi * freq;
            w = v0 + va * sin(om * tau + phi);
        }
    } else if (wave == CSIM_WAVE_PULSE) {
        const double v1 = P(1), v2 = P(2), td = P(3), tr = P(4), tf = P(5), ton = P(6), per = P(7);
        if (per <= 0.0) {
            const double tau = t - td;
            if (tau <= 0.0) w = v1;
            else if (tau < tr) { const double k = clamp01_dev(tau / tr); w = v1 + k * (v2 - v1); }
            else if (tau < tr + ton) w = v2;
            else { const double tfall = tau - (tr + ton); const double k = clamp01_dev(tfall / tf); w = v2 + k * (v1 - v2); }
        } else if (t < td) {
            w = v1;
        } else {
            double tau = fmod(t - td, per);
            if (tau < 0.0) tau += per;
            if (tau < tr) { const double k = clamp01_dev(tau / tr); w = v1 + (v2 - v1) * k; }
            else if (tau < tr + ton) w = v2;
            else if (tau < tr + ton + tf) { const double tfall = tau - (tr + ton); const double k = clamp01_dev(tfall / tf); w = v2 + (v1 - v2) * k; }
            else w = v1;
        }
    } else if (wave == CSIM_WAVE_PWL) {
        const int n = waveN;
        if (n <= 0) w = 0.0;
        else if (t <= P(1)) w = P(1 + n);
        else if (t >= P(n)) w = P(2 * n);
        else {
            w = P(2 * n);
            for (int i = 0; i + 1 < n; ++i) {
                const double ta = P(1 + i), tb = P(2 + i);
                if (t > ta && t <= tb) {
                    const double va = P(1 + n + i), vb = P(2 + n + i);
                    const double k = (t - ta) / (tb - ta);
                    w = va + (vb - va) * k;
                    break;
                }
            }
        }
    }
    return w;
}

__device__ __forceinline__ double source_value_tran(const double* Pv, int slot, int wave, int waveN, double t, double pi)
{
    const double dc = Pv[slot + 0];
    return dc + source_wave_value([&](int i) { return Pv[slot + i]; }, wave, waveN, t, pi);
}

// Level-1 MOSFET linearisation at (Vd, Vg, Vs): MosfetBase::stamp, element.cpp:207-274
struct MosLin { double gd, gg, gs, cst; };
__device__ __forceinline__ MosLin mos_eval(bool isP, double Vth, double K, double lambda, double offGds,
                                           double Vd, double Vg, double Vs)
{
    const double p = isP ? -1.0 : 1.0;
    const double Vgs = p * (Vg - Vs);
    const double Vds = p * (Vd - Vs);
    double Ids0 = 0.0, gds0 = offGds, gm0 = 0.0;        // off state (:245-252)
    if (Vgs > Vth && Vds >= 0.0) {                      // (:223)
        const double Vov = Vgs - Vth;
        if (Vds < Vov) {                                // triode (:232-236)
            Ids0 = K * (Vov * Vds - 0.5 * Vds * Vds);
            gds0 = K * (Vov - Vds);
            gm0  = K * Vds;
        } else {                                        // saturation (:239-241)
            Ids0 = 0.5 * K * Vov * Vov;
            gds0 = 0.0;
            gm0  = K * Vov;
        }
    }
    double factor = 1.0 + lambda * Vds;                 // (:255-256)
    if (factor < 0.0) factor = 0.0;
    const double Ids = p * (Ids0 * factor);             // (:257,266)
    MosLin r;
    r.gd = gds0 * factor + Ids0 * lambda;               // (:260,269)
    r.gg = gm0 * factor;                                // (:263,270)
    r.gs = -(r.gd + r.gg);                              // (:271)
    r.cst = Ids - r.gd * Vd - r.gg * Vg - r.gs * Vs;    // (:274)
    return r;
}

// ---- terms that stay constant over a launch
template <bool TRAN>
__device__ __forceinline__ void terms_const(const GenPlan& pl, const double* Pv, double* T, double dt, int lane, int stride = 64)
{
    for (int e = lane; e < pl.nElem; e += stride) {
        const int kind = pl.kind[e], s = pl.slot[e], tb = pl.termBase[e];
        if (kind == CSIM_R) {
            const double R = Pv[s];
            T[tb + T_R_G] = (R == 0.0) ? 0.0 : 1.0 / R;            // element.cpp:20-24
        } else if (TRAN && kind == CSIM_C) {
            const double C = Pv[s];
            T[tb + T_C_GC] = (C > 0.0 && dt > 0.0) ? C / dt : 0.0; // tanalisis.cpp:65-67
            T[tb + T_C_IH] = 0.0;
        } else if (TRAN && kind == CSIM_L) {
            const double L = Pv[s];
            const bool on = L > 0.0;                               // tanalisis.cpp:296
            T[tb + T_L_REQ] = on ? L / dt : 0.0;
            T[tb + T_L_VH] = 0.0;
            T[tb + T_L_ONE] = on ? 1.0 : 0.0;
        } else if (TRAN && (kind == CSIM_NMOS || kind == CSIM_PMOS)) {
            const double Cj0 = Pv[s + 3];
            const double Ch = 0.5 * Cj0;                           // tanalisis.cpp:337-341
            T[tb + T_M_GCH] = (Ch > 0.0 && dt > 0.0) ? Ch / dt : 0.0;
            T[tb + T_M_GCF] = (Cj0 > 0.0 && dt > 0.0) ? Cj0 / dt : 0.0;
        }
    }
    if (lane == 0) T[pl.termOne] = 1.0;
}

// ---- terms that change once per time step: sources and history currents
__device__ __forceinline__ void terms_step_tran(const GenPlan& pl, const double* Pv, double* T,
                                                const double* xp, double t, int lane, int stride = 64)
{
    for (int e = lane; e < pl.nElem; e += stride) {
        const int kind = pl.kind[e], s = pl.slot[e], tb = pl.termBase[e];
        const int32_t* q = pl.eq + 4 * e;
        if (kind == CSIM_V || kind == CSIM_I) {
            T[tb + T_SRC_VAL] = source_value_tran(Pv, s, pl.wave[e], pl.waveN[e], t, pl.k.pi);
        } else if (kind == CSIM_C) {
            const double vPrev = volt_of(xp, q[0]) - volt_of(xp, q[1]);
            T[tb + T_C_IH] = -T[tb + T_C_GC] * vPrev;              // tanalisis.cpp:77
        } else if (kind == CSIM_L) {
            const int k = pl.branch[e];
            const double iPrev = (k >= 0 && k < pl.N) ? xp[k] : 0.0;
            T[tb + T_L_VH] = -T[tb + T_L_REQ] * iPrev;             // tanalisis.cpp:308
        } else if (kind == CSIM_NMOS || kind == CSIM_PMOS) {
            const double vD = volt_of(xp, q[0]), vG = volt_of(xp, q[1]);
            const double vS = volt_of(xp, q[2]), vB = volt_of(xp, q[3]);
            const double gh = T[tb + T_M_GCH], gf = T[tb + T_M_GCF];
            T[tb + T_M_IHGS] = -gh * (vG - vS);
            T[tb + T_M_IHGD] = -gh * (vG - vD);
            T[tb + T_M_IHSB] = -gf * (vS - vB);
            T[tb + T_M_IHDB] = -gf * (vD - vB);
        }
    }
}

__device__ __forceinline__ void terms_step_dc(const GenPlan& pl, const double* Pv, double* T, double scale, int lane, int stride = 64)
{
    for (int e = lane; e < pl.nElem; e += stride) {
        const int kind = pl.kind[e];
        if (kind == CSIM_V || kind == CSIM_I)
            T[pl.termBase[e] + T_SRC_VAL] = source_value_dc(Pv, pl.slot[e], pl.wave[e], scale);
    }
}

// ---- terms that change every Newton iteration: MOS channel at the iterate x
__device__ __forceinline__ void terms_iter_mos(const GenPlan& pl, const double* Pv, double* T, const double* x, int lane, int stride = 64)
{
    for (int e = lane; e < pl.nElem; e += stride) {
        const int kind = pl.kind[e];
        if (kind != CSIM_NMOS && kind != CSIM_PMOS) continue;
        const int s = pl.slot[e], tb = pl.termBase[e];
        const int32_t* q = pl.eq + 4 * e;
        const MosLin m = mos_eval(kind == CSIM_PMOS, Pv[s + 0], Pv[s + 1], Pv[s + 2], pl.k.mos_off_gds,
                                  volt_of(x, q[0]), volt_of(x, q[1]), volt_of(x, q[2]));
        T[tb + T_M_GD] = m.gd;
        T[tb + T_M_GG] = m.gg;
        T[tb + T_M_GS] = m.gs;
        T[tb + T_M_CST] = m.cst;
    }
}

// ---- build [G | I] in LDS: clear, then every structural non-zero sums its
// terms in the reference's accumulation order
__device__ __forceinline__ void assemble(const GenPlan& pl, const double* T, double* Gm, int lane, int stride = 64)
{
    const int total = pl.N * pl.LD;
    for (int i = lane; i < total; i += stride) Gm[i] = 0.0;
    wave_sync();
    for (int n = lane; n < pl.nnzG; n += stride) {
        double acc = 0.0;
        for (int c = pl.gPtr[n]; c < pl.gPtr[n + 1]; ++c) {
            const int con = pl.gCon[c];
            const double v = T[con >> 1];
            acc = (con & 1) ? acc - v : acc + v;
        }
        Gm[pl.gPos[n]] = acc;
    }
    for (int n = lane; n < pl.nnzI; n += stride) {
        double acc = 0.0;
        for (int c = pl.iPtr[n]; c < pl.iPtr[n + 1]; ++c) {
            const int con = pl.iCon[c];
            const double v = T[con >> 1];
            acc = (con & 1) ? acc - v : acc + v;
        }
        Gm[pl.iRow[n] * pl.LD + pl.N] = acc;
    }
    wave_sync();
}

// is the pivot sequence in curPiv (LDS, N ints) one of the nAlts known ones ([nAlts][N], global)?
__device__ __forceinline__ bool sequence_is_known(const int32_t* curPiv, const int32_t* alts, int nAlts, int N, int lane)
{
    wave_sync();
    bool known = false;
    for (int a = 0; a < nAlts && !known; ++a) {
        bool same = true;
        for (int k = lane; k < N; k += 64) same = same && (curPiv[k] == alts[a * N + k]);
        known = __ballot(!same) == 0ull;
    }
    return known;
}

// ---- planner log (global memory, one per recorded instance):
//   [0] number of distinct sequences stored (<= PIVLOG_MAX)   [1] factorisations seen
//   [2] factorisations that failed or did not fit the table
//   then PIVLOG_MAX records of (N positions + 1 count), then N ints of the sequence in progress
constexpr int PIVLOG_MAX = 8;
__host__ __device__ inline int pivlog_ints(int N) { return 3 + PIVLOG_MAX * (N + 1) + N; }
__device__ __forceinline__ int32_t* pivlog_cur(int32_t* log, int N) { return log + 3 + PIVLOG_MAX * (N + 1); }
__device__ __forceinline__ void pivlog_commit(int32_t* log, int N, bool failed, int lane)
{
    __threadfence_block();
    wave_sync();
    if (lane == 0) {
        log[1] += 1;
        if (failed) log[2] += 1;
        else {
            const int32_t* cur = pivlog_cur(log, N);
            int hit = -1;
            for (int s = 0; s < log[0] && hit < 0; ++s) {
                const int32_t* rec = log + 3 + s * (N + 1);
                bool same = true;
                for (int k = 0; k < N && same; ++k) same = rec[k] == cur[k];
                if (same) hit = s;
            }
            if (hit < 0 && log[0] < PIVLOG_MAX) {
                hit = log[0];
                int32_t* rec = log + 3 + hit * (N + 1);
                for (int k = 0; k < N; ++k) rec[k] = cur[k];
                rec[N] = 0;
                log[0] += 1;
            }
            if (hit >= 0) log[3 + hit * (N + 1) + N] += 1; else log[2] += 1;
        }
    }
    __threadfence_block();
    wave_sync();
}

// ---- wavefront-cooperative LU with partial pivoting + substitution on the
// augmented LDS matrix (N <= 63: row i and column j are owned by lane i / j,
// the RHS is column N).
//
// Same pivot rule as Solver::luDecompose: the FIRST row attaining the column
// maximum (solver.hpp:48-56, strict '>'), failure if that maximum is < eps
// (:58-61) -> zero solution vector (:94-97).  Forward substitution is fused
// into the elimination by carrying the RHS column (identical operation order
// per row: multipliers are applied in ascending k).  Rows whose multiplier is
// exactly zero are skipped: a - 0*b == a, so this is bit-identical while the
// matrix is ~10 % dense.  Back substitution keeps the reference's row-wise,
// ascending-j summation order.
//
// Returns the solution component of lane i (< N) and ORs CSIM_ST_LU_* flags.
//
// pivLog (optional, planner): see PIVLOG_* below -- the distinct pivot sequences seen, with
// how many factorisations used each.
__device__ __forceinline__ double lu_solve_wave(double* Gm, int N, int LD, double eps, int lane, unsigned& flags,
                                                int32_t* pivLog = nullptr, int32_t* curPiv = nullptr)
{
    int32_t* const logCur = pivLog ? pivlog_cur(pivLog, N) : nullptr;
    double diag = 1.0;          // lane k keeps U(k,k)
    bool failed = false;

    for (int k = 0; k < N; ++k) {
        // column k below and including the diagonal: lane i holds a(i,k)
        double colv = (lane < N) ? Gm[lane * LD + k] : 0.0;
        const double av = fabs(colv);
        const double akk = read_lane(av, k);
        int piv = k;
        double maxAbs = akk;
        if (akk == akk) {       // a NaN diagonal keeps pivot = k in the reference
            // solver.hpp:50-56 restricted to the rows that can win: a zero (or NaN)
            // entry never satisfies "val > maxAbs", so only non-zero candidates
            // below the diagonal are visited, in ascending row order
            unsigned long long cand = __ballot(lane > k && lane < N && av > 0.0);
            while (cand) {
                const int i = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const double v = read_lane(av, i);
                if (v > maxAbs) { maxAbs = v; piv = i; }
            }
        }
        if (maxAbs < eps) { failed = true; break; }
        if (logCur && lane == 0) logCur[k] = piv;
        if (curPiv && lane == 0) curPiv[k] = piv;      // LDS: this factorisation's sequence (hybrid stepping)

        if (piv != k) {         // swap rows k and piv (columns >= k and the RHS)
            if (lane >= k && lane <= N) {
                const double a = Gm[k * LD + lane], b = Gm[piv * LD + lane];
                Gm[k * LD + lane] = b;
                Gm[piv * LD + lane] = a;
            }
            const double ck = read_lane(colv, k), cp = read_lane(colv, piv);
            if (lane == k) colv = cp;
            if (lane == piv) colv = ck;
            wave_sync();
        }

        const double pivv = read_lane(colv, k);
        if (lane == k) diag = pivv;
        // lane j holds the pivot row entry a(k,j), j in (k, N]
        const double rowv = (lane > k && lane <= N) ? Gm[k * LD + lane] : 0.0;

        const bool active = lane > k && lane < N && colv != 0.0;
        const double fmine = active ? colv / pivv : 0.0;            // solver.hpp:71, row = lane
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const double f = read_lane(fmine, i);
            if (lane > k && lane <= N) Gm[i * LD + lane] -= f * rowv;   // :74 (+ RHS)
        }
        wave_sync();
    }

    if (pivLog) pivlog_commit(pivLog, N, failed, lane);
    if (failed) {
        flags |= CSIM_ST_LU_TINY_PIVOT;
        return 0.0;
    }

    // back substitution, solver.hpp:116-128, in the reference's order: row i
    // (descending) subtracts U(i,j)*x(j) for j ascending.  Lane j owns x(j) and
    // forms its product; the ordered sum walks the non-zero products only
    // (sum - 0 == sum), so the sparse rows cost ~3 steps each.
    const double y = (lane < N) ? Gm[lane * LD + N] : 0.0;
    double xv = 0.0;
    for (int i = N - 1; i >= 0; --i) {
        const double u = (lane > i && lane < N) ? Gm[i * LD + lane] : 0.0;
        const double prod = u * xv;                                  // :119
        unsigned long long todo = __ballot(lane > i && lane < N && prod != 0.0);
        double sum = read_lane(y, i);                                // :117
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            sum -= read_lane(prod, j);
        }
        const double d = read_lane(diag, i);                         // :121
        double xi;
        if (fabs(d) < eps) { xi = 0.0; flags |= CSIM_ST_LU_ZERO_DIAG; }   // :122-124
        else xi = sum / d;                                           // :126
        if (lane == i) xv = xi;
    }
    return xv;
}

// Solver::luDecompose proper (solver.hpp:30-80): in-place P*A = L*U with the
// multipliers stored below the diagonal and whole-row swaps, for callers that
// want the factors.  perm value of row position `lane` is returned in permv.
__device__ __forceinline__ bool lu_factor_wave(double* Gm, int N, int LD, double eps, int lane, int& permv)
{
    permv = lane;
    for (int k = 0; k < N; ++k) {
        double colv = (lane < N) ? Gm[lane * LD + k] : 0.0;
        const double av = fabs(colv);
        const double akk = read_lane(av, k);
        int piv = k;
        double maxAbs = akk;
        if (akk == akk) {
            unsigned long long cand = __ballot(lane > k && lane < N && av > 0.0);
            while (cand) {
                const int i = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const double v = read_lane(av, i);
                if (v > maxAbs) { maxAbs = v; piv = i; }
            }
        }
        if (maxAbs < eps) return false;                              // :58-61
        if (piv != k) {                                              // :64-67 whole rows + perm
            if (lane < N) {
                const double a = Gm[k * LD + lane], b = Gm[piv * LD + lane];
                Gm[k * LD + lane] = b;
                Gm[piv * LD + lane] = a;
            }
            const double ck = read_lane(colv, k), cp = read_lane(colv, piv);
            const int pk = __builtin_amdgcn_readlane(permv, __builtin_amdgcn_readfirstlane(k));
            const int pp = __builtin_amdgcn_readlane(permv, __builtin_amdgcn_readfirstlane(piv));
            if (lane == k) { colv = cp; permv = pp; }
            if (lane == piv) { colv = ck; permv = pk; }
            wave_sync();
        }
        const double pivv = read_lane(colv, k);
        const double rowv = (lane > k && lane < N) ? Gm[k * LD + lane] : 0.0;
        const bool below = lane > k && lane < N;
        const double fmine = below ? colv / pivv : 0.0;              // :71
        if (below) Gm[lane * LD + k] = fmine;                        // :72
        unsigned long long todo = __ballot(below && colv != 0.0);
        while (todo) {
            const int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const double f = read_lane(fmine, i);
            if (lane > k && lane < N) Gm[i * LD + lane] -= f * rowv; // :74
        }
        wave_sync();
    }
    return true;
}

// ||v||_2 with the squares summed in index order 0..N-1 (the order of the
// oracle's norm); sc is an N-double LDS scratch
__device__ __forceinline__ double norm_in_order(double d, double* sc, int N, int lane)
{
    if (lane < N) sc[lane] = d * d;
    wave_sync();
    double ss = 0.0;
    for (int i = 0; i < N; ++i) ss += sc[i];
    wave_sync();
    return sqrt(ss);
}

#endif // device

} // namespace csim
