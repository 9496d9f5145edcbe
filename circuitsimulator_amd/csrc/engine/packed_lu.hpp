// packed_lu.hpp -- device code shared by kernels_packed.hip and kernels_packed_lu.hip: one DPP row of 16 lanes
// per system, the augmented matrix in registers (see kernels_packed.hip for the design).  Include inside
// namespace csim, after device_common.hpp, in a HIP translation unit.
#pragma once

namespace {

constexpr int G16 = 16;          // lanes per instance
constexpr int IPW = 4;           // instances per wavefront

__device__ __forceinline__ unsigned grp_mask(bool pred, int q)
{
    return (unsigned)((__ballot(pred) >> (q * G16)) & 0xFFFFull);
}

__device__ __forceinline__ double clampd_p(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ double base_gmin_p(const csim_consts& k, double s)
{
    s = clampd_p(s, 0.0, 1.0);
    return k.gmin_high * (1.0 - s) + k.gmin_low * s;
}

// ---- cross-lane plumbing inside one DPP row of 16 lanes (= one instance)
template <int L> __device__ __forceinline__ double row_bcast(double v)          // lane L of the row -> every lane of it
{
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xF, 0xF, true);      // v_mov_b64_dpp row_newbcast:L
}
__device__ __forceinline__ double row_shr_keep(double v, int n)                 // lane i <- lane i - n of the row; the first n lanes keep v
{
    const long long b = __double_as_longlong(v);
    long long r;
    switch (n) {
        case 1: r = __builtin_amdgcn_update_dpp(b, b, 0x111, 0xF, 0xF, false); break;
        case 2: r = __builtin_amdgcn_update_dpp(b, b, 0x112, 0xF, 0xF, false); break;
        case 4: r = __builtin_amdgcn_update_dpp(b, b, 0x114, 0xF, 0xF, false); break;
        default: r = __builtin_amdgcn_update_dpp(b, b, 0x118, 0xF, 0xF, false); break;
    }
    return __longlong_as_double(r);
}
// maximum over the 16 lanes of a row, in every lane; a NaN operand is ignored (v_max_f64)
__device__ __forceinline__ double row_max16(double v)
{
    v = fmax(v, row_shr_keep(v, 1));
    v = fmax(v, row_shr_keep(v, 2));
    v = fmax(v, row_shr_keep(v, 4));
    v = fmax(v, row_shr_keep(v, 8));
    return row_bcast<15>(v);                                                    // lane 15 holds the maximum of all 16
}

// ---- LDS carve-up.  Per instance (doubles): element terms, parameters, iterate, previous state, gathered
// non-zeros (+ one 0.0 for the structural zeros), right-hand side, norm scratch, pivot sequence (ints).
// Per workgroup (ints, behind the four instances): rowMap[N * LD], then the staged plan arrays.
struct PackedLayout {
    int T, P, xs, xp, Gs, Rs, sc, piv, total;
};
__host__ __device__ inline PackedLayout packedLayout(const GenPlan& pl)
{
    const int Npad = (pl.N + 1) & ~1;
    PackedLayout l;
    l.T = 0;
    l.P = l.T + pl.nTerms;
    l.xs = l.P + pl.P;
    l.xp = l.xs + Npad;
    l.Gs = l.xp + Npad;
    l.Rs = l.Gs + pl.nnzG + 1;
    l.sc = l.Rs + Npad;
    l.piv = l.sc + Npad;
    l.total = l.piv + Npad / 2 + 1;
    return l;
}
inline size_t packedLdsBytes(const GenPlan& pl)
{
    return sizeof(double) * (size_t)packedLayout(pl).total * IPW +
           sizeof(int32_t) * ((size_t)pl.N * pl.LD + (size_t)planLdsInts(pl));
}

// every structural non-zero sums its terms in the reference's accumulation order (device_common.hpp
// assemble(), without the dense matrix)
__device__ __forceinline__ void gather_nonzeros(const GenPlan& pl, const double* T, double* Gs, double* Rs, int g)
{
    // (two non-zeros per trip, to overlap the dependent LDS reads of two sums, was measured slower: 1.25e8 -> 1.20e8)
    for (int n = g; n < pl.nnzG; n += G16) {
        double acc = 0.0;
        for (int c = pl.gPtr[n]; c < pl.gPtr[n + 1]; ++c) {
            const int con = pl.gCon[c];
            const double v = T[con >> 1];
            acc = (con & 1) ? acc - v : acc + v;
        }
        Gs[n] = acc;
    }
    for (int n = g; n < pl.nnzI; n += G16) {
        double acc = 0.0;
        for (int c = pl.iPtr[n]; c < pl.iPtr[n + 1]; ++c) {
            const int con = pl.iCon[c];
            const double v = T[con >> 1];
            acc = (con & 1) ? acc - v : acc + v;
        }
        Rs[pl.iRow[n]] = acc;
    }
}

// Solver::luDecompose + solveLinearSystemLU (solver.hpp:30-131) for one group of 16 lanes.  NP = the padded
// size (N for N <= 16; N or N + 1, even, for 17 <= N <= 32), S = slots (rows per lane).  The column and row
// indices are template parameters (a recursion instead of loops the compiler would have to unroll whole).

// column K of the elimination (:46-77)
template <int NP, int S, int K>
__device__ __forceinline__ void lu_column(double (&a)[S][NP + 1], int N, double eps, int g, int q, bool& failed, int32_t* curPiv)
{
    if (S > 1 && K == NP - 1 && K >= N) return;                           // the padding column: nothing to do
    constexpr int sk = K / 16, lk = K % 16;
    double av[S];
#pragma unroll
    for (int s = 0; s < S; ++s) av[s] = fabs(a[s][K]);
    const double akk = row_bcast<lk>(av[sk]);
    int piv = K;
    double maxAbs = akk;
    if (K + 1 < NP) {
        // Candidates below the diagonal.  In most columns no row of any of the wave's four instances exceeds its
        // diagonal (the pivot stays): one comparison and a ballot find that out and skip the reduction (a NaN
        // diagonal or a NaN candidate compares false, as in the reference's "val > maxAbs", :53).
        bool bigger = false;
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s >= sk) bigger = bigger || (16 * s + g > K && 16 * s + g < N && av[s] > akk);
        if (__any(bigger)) {
            double m = -1.0;                                              // v_max_f64 drops a NaN operand
#pragma unroll
            for (int s = 0; s < S; ++s)
                if (s >= sk) m = fmax(m, (16 * s + g > K && 16 * s + g < N) ? av[s] : -1.0);
            m = row_max16(m);
            if (akk == akk && m > akk) {                                  // a NaN diagonal keeps pivot = K
                maxAbs = m;
                piv = -1;                                                 // FIRST row attaining it: lowest slot, lowest lane
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (s < sk) continue;
                    const unsigned cand = grp_mask(16 * s + g > K && 16 * s + g < N && av[s] == m, q);
                    if (piv < 0 && cand != 0u) piv = 16 * s + __ffs((int)cand) - 1;
                }
            }
        }
    }
    if (!failed && maxAbs < eps) failed = true;                           // :58-61
    const bool live = !failed;
    if (curPiv && g == 0 && live) curPiv[K] = piv;
    if (K + 1 >= NP) return;
    const bool sw = live && piv != K;                                     // :64-67 (columns >= K and the RHS matter)
    const int ps = piv >> 4, pl = piv & 15;
    const bool swSame = sw && ps == sk;
    if (__any(swSame)) {
        const int src = q * 16 + (swSame ? (g == lk ? pl : (g == pl ? lk : g)) : g);
#pragma unroll
        for (int j = K; j <= NP; ++j) a[sk][j] = __shfl(a[sk][j], src);
    }
    if (sk + 1 < S) {                                                     // the pivot row may sit in the other slot
        const bool swCross = sw && ps != sk;
        if (__any(swCross)) {
            const int srcA = q * 16 + ((swCross && g == lk) ? pl : g);    // lane lk fetches row piv (its lane is run-time)
#pragma unroll
            for (int j = K; j <= NP; ++j) {
                const double tA = __shfl(a[S - 1][j], srcA);
                const double tB = row_bcast<lk>(a[sk][j]);                // row K sits in a lane known here: one DPP move
                a[sk][j] = (swCross && g == lk) ? tA : a[sk][j];
                a[S - 1][j] = (swCross && g == pl) ? tB : a[S - 1][j];
            }
        }
    }
    const double pivv = row_bcast<lk>(a[sk][K]);
    double u[NP + 1];
#pragma unroll
    for (int j = K + 1; j <= NP; ++j) u[j] = row_bcast<lk>(a[sk][j]);    // the pivot row, to every lane
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s < sk) continue;
        if (live && 16 * s + g > K && 16 * s + g < N) {                   // :70-76, rows below the pivot
            const double f = a[s][K] / pivv;                              // :71
#pragma unroll
            for (int j = K + 1; j <= NP; ++j) {
                if (S > 1 && j == NP - 1 && j >= N) continue;             // padding column
                a[s][j] = a[s][j] - f * u[j];                             // :74 (+ RHS = forward substitution)
            }
        }
    }
}
template <int NP, int S, int K>
__device__ __forceinline__ void lu_columns(double (&a)[S][NP + 1], int N, double eps, int g, int q, bool& failed, int32_t* curPiv)
{
    if constexpr (K < NP) {
        lu_column<NP, S, K>(a, N, eps, g, q, failed, curPiv);
        lu_columns<NP, S, K + 1>(a, N, eps, g, q, failed, curPiv);
    }
}

// row I of the back substitution (:116-128): subtracts U(I,j) x(j) for j ascending.  Every lane runs the sum on
// its own row of slot I / 16; lane I % 16's is row I's, and its x(I) is broadcast for the rows above.
template <int NP, int S, int I>
__device__ __forceinline__ void lu_back_rows(const double (&a)[S][NP + 1], double (&x)[NP], int N, double eps, int g, bool on,
                                             unsigned& flags, double (&xout)[S])
{
    if constexpr (I >= 0) {
        if (S > 1 && I == NP - 1 && I >= N) {
            x[I] = 0.0;
        } else {
            constexpr int si = I / 16, li = I % 16;
            double sum = a[si][NP];
#pragma unroll
            for (int j = I + 1; j < NP; ++j) {
                if (S > 1 && j == NP - 1 && j >= N) continue;
                sum -= a[si][j] * x[j];                                   // :119
            }
            const double d = row_bcast<li>(a[si][I]);                     // :121 U(I,I)
            const bool tiny = fabs(d) < eps;
            const double xi = tiny ? 0.0 : sum / d;                       // :122-126 (lane li's is x(I))
            if (on && tiny) flags |= CSIM_ST_LU_ZERO_DIAG;
            x[I] = row_bcast<li>(xi);
            if (g == li) xout[si] = xi;
        }
        lu_back_rows<NP, S, I - 1>(a, x, N, eps, g, on, flags, xout);
    }
}

// xout[s] returns the solution component of row 16 s + g.  `on` = this group's solve counts (flags are only
// raised for such groups).
template <int NP, int S>
__device__ __forceinline__ void lu_solve_rows(const double* Gs, const double* Rs, const int32_t* rowMap, int zeroAt, int N, int LD,
                                              double eps, int g, int q, bool on, unsigned& flags, int32_t* curPiv, double (&xout)[S])
{
    static_assert(S == (NP + 15) / 16, "slots");
    double a[S][NP + 1];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int r = 16 * s + g;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int at = (r < N && j < N) ? rowMap[r * LD + j] : zeroAt;
            const double v = Gs[at];
            a[s][j] = (r >= N && r == j) ? 1.0 : v;                       // padding row: identity
        }
        a[s][NP] = (r < N) ? Rs[r] : 0.0;
        xout[s] = 0.0;
    }
    bool failed = false;
    lu_columns<NP, S, 0>(a, N, eps, g, q, failed, curPiv);
    if (failed) {                                                         // :94-97: zero vector
        if (on) flags |= CSIM_ST_LU_TINY_PIVOT;
        return;
    }
    double x[NP];
    lu_back_rows<NP, S, NP - 1>(a, x, N, eps, g, on, flags, xout);
}

// one body per size, so that the size is a constant inside (the switch is uniform)
#define CSIM_LU_CASE(NPV, SV) case NPV: lu_solve_rows<NPV, SV>(Gs, Rs, rowMap, zeroAt, N, LD, eps, g, q, on, flags, curPiv, x); break;
__device__ __forceinline__ void lu_solve_dispatch(const double* Gs, const double* Rs, const int32_t* rowMap, int zeroAt, int N, int LD,
                                                  double eps, int g, int q, bool on, unsigned& flags, int32_t* curPiv, double (&x)[1])
{
    switch (N) {
        CSIM_LU_CASE(1, 1) CSIM_LU_CASE(2, 1) CSIM_LU_CASE(3, 1) CSIM_LU_CASE(4, 1) CSIM_LU_CASE(5, 1)
        CSIM_LU_CASE(6, 1) CSIM_LU_CASE(7, 1) CSIM_LU_CASE(8, 1) CSIM_LU_CASE(9, 1) CSIM_LU_CASE(10, 1)
        CSIM_LU_CASE(11, 1) CSIM_LU_CASE(12, 1) CSIM_LU_CASE(13, 1) CSIM_LU_CASE(14, 1) CSIM_LU_CASE(15, 1)
        default: lu_solve_rows<16, 1>(Gs, Rs, rowMap, zeroAt, N, LD, eps, g, q, on, flags, curPiv, x); break;
    }
}
__device__ __forceinline__ void lu_solve_dispatch(const double* Gs, const double* Rs, const int32_t* rowMap, int zeroAt, int N, int LD,
                                                  double eps, int g, int q, bool on, unsigned& flags, int32_t* curPiv, double (&x)[2])
{
    switch ((N + 1) & ~1) {
        CSIM_LU_CASE(18, 2) CSIM_LU_CASE(20, 2) CSIM_LU_CASE(22, 2) CSIM_LU_CASE(24, 2)
        CSIM_LU_CASE(26, 2) CSIM_LU_CASE(28, 2) CSIM_LU_CASE(30, 2)
        default: lu_solve_rows<32, 2>(Gs, Rs, rowMap, zeroAt, N, LD, eps, g, q, on, flags, curPiv, x); break;
    }
}
#undef CSIM_LU_CASE

} // namespace
