// plan.cpp -- symbolic replay of the reference's stamping order.
#include "plan.hpp"

#include <map>
#include <utility>

namespace csim {
namespace {

// ordered (position -> contributions) recorder
struct Recorder {
    int N, LD;
    std::vector<std::pair<int, int32_t>> g, i;     // (pos|row, con) in stamping order

    void G(int r, int c, int term, bool neg)
    {
        if (r < 0 || c < 0) return;                 // ground row/column: "if (eq >= 0)" guards
        g.emplace_back(r * LD + c, (term << 1) | (neg ? 1 : 0));
    }
    void I(int r, int term, bool neg)
    {
        if (r < 0) return;
        i.emplace_back(r, (term << 1) | (neg ? 1 : 0));
    }
    // a two-terminal conductance stamp: Resistor::stamp (element.cpp:26-31),
    // stampCapBE (tanalisis.cpp:69-74)
    void conductance(int a, int b, int term)
    {
        G(a, a, term, false);
        G(b, b, term, false);
        if (a >= 0 && b >= 0) { G(a, b, term, true); G(b, a, term, true); }
    }
    // VoltageSource::stamp / Inductor::stamp incidence (element.cpp:115-120, 173-177;
    // tanalisis.cpp:311-316)
    void incidence(int p, int m, int k, int oneTerm)
    {
        G(p, k, oneTerm, false);
        G(m, k, oneTerm, true);
        G(k, p, oneTerm, false);
        G(k, m, oneTerm, true);
    }
    // stampCapBE history current (tanalisis.cpp:77-79): I(a) -= Ih; I(b) += Ih
    void capHistory(int a, int b, int ihTerm)
    {
        I(a, ihTerm, true);
        I(b, ihTerm, false);
    }
};

void group(const std::vector<std::pair<int, int32_t>>& seq, std::vector<int32_t>& ptr,
           std::vector<int32_t>& key, std::vector<int32_t>& con)
{
    std::map<int, std::vector<int32_t>> by;        // sorted by position, order kept inside
    for (const auto& pc : seq) by[pc.first].push_back(pc.second);
    ptr.assign(1, 0);
    key.clear();
    con.clear();
    for (const auto& kv : by) {
        key.push_back(kv.first);
        con.insert(con.end(), kv.second.begin(), kv.second.end());
        ptr.push_back(static_cast<int32_t>(con.size()));
    }
}

void mosChannel(Recorder& r, const int* q, int base)
{
    const int D = q[0], Gt = q[1], S = q[2];
    // MosfetBase::stamp rows D and S (element.cpp:290-304)
    if (D >= 0) {
        r.G(D, D, base + T_M_GD, false);
        r.G(D, Gt, base + T_M_GG, false);
        r.G(D, S, base + T_M_GS, false);
        r.I(D, base + T_M_CST, true);
    }
    if (S >= 0) {
        r.G(S, D, base + T_M_GD, true);
        r.G(S, Gt, base + T_M_GG, true);
        r.G(S, S, base + T_M_GS, true);
        r.I(S, base + T_M_CST, false);
    }
}

} // namespace

AssemblyPlan buildAssemblyPlan(const csim_ir& ir)
{
    AssemblyPlan pl;
    pl.N = ir.n_unknowns;
    pl.LD = ldFor(pl.N);
    pl.termBase.resize(static_cast<std::size_t>(ir.n_elems));
    int nt = 0;
    for (int e = 0; e < ir.n_elems; ++e) {
        pl.termBase[static_cast<std::size_t>(e)] = nt;
        nt += termsOfKind(ir.kind[e]);
    }
    pl.termOne = nt++;
    pl.termGmin = nt++;
    pl.nTerms = nt;
    const int N = pl.N;
    auto validBranch = [N](int k) { return k >= 0 && k < N; };

    // ---- DC: every element in netlist order, then gmin (dcanalysis.cpp:126-130)
    {
        Recorder r{pl.N, pl.LD, {}, {}};
        for (int e = 0; e < ir.n_elems; ++e) {
            const int* q = ir.eq + 4 * e;
            const int base = pl.termBase[static_cast<std::size_t>(e)];
            const int k = ir.branch_eq[e];
            switch (ir.kind[e]) {
                case CSIM_R: r.conductance(q[0], q[1], base + T_R_G); break;
                case CSIM_C: break;                                  // open circuit
                case CSIM_L: if (validBranch(k)) r.incidence(q[0], q[1], k, pl.termOne); break;
                case CSIM_V:
                    if (validBranch(k)) { r.incidence(q[0], q[1], k, pl.termOne); r.I(k, base + T_SRC_VAL, false); }
                    break;
                case CSIM_I: r.I(q[0], base + T_SRC_VAL, true); r.I(q[1], base + T_SRC_VAL, false); break;
                case CSIM_NMOS: case CSIM_PMOS: mosChannel(r, q, base); break;
                default: break;
            }
        }
        for (int eq = 0; eq < ir.n_node_eq; ++eq) r.G(eq, eq, pl.termGmin, false);
        group(r.g, pl.dc.gPtr, pl.dc.gPos, pl.dc.gCon);
        group(r.i, pl.dc.iPtr, pl.dc.iRow, pl.dc.iCon);
    }

    // ---- TRAN: the six phases of tanalisis.cpp:269-356
    {
        Recorder r{pl.N, pl.LD, {}, {}};
        for (int e = 0; e < ir.n_elems; ++e) {                       // 1) R, V, I
            const int* q = ir.eq + 4 * e;
            const int base = pl.termBase[static_cast<std::size_t>(e)];
            const int k = ir.branch_eq[e];
            switch (ir.kind[e]) {
                case CSIM_R: r.conductance(q[0], q[1], base + T_R_G); break;
                case CSIM_V:
                    if (validBranch(k)) { r.incidence(q[0], q[1], k, pl.termOne); r.I(k, base + T_SRC_VAL, false); }
                    break;
                case CSIM_I: r.I(q[0], base + T_SRC_VAL, true); r.I(q[1], base + T_SRC_VAL, false); break;
                default: break;
            }
        }
        for (int e = 0; e < ir.n_elems; ++e)                         // 2) MOS channel
            if (ir.kind[e] == CSIM_NMOS || ir.kind[e] == CSIM_PMOS)
                mosChannel(r, ir.eq + 4 * e, pl.termBase[static_cast<std::size_t>(e)]);
        for (int e = 0; e < ir.n_elems; ++e) {                       // 3) explicit capacitors
            if (ir.kind[e] != CSIM_C) continue;
            const int* q = ir.eq + 4 * e;
            const int base = pl.termBase[static_cast<std::size_t>(e)];
            r.conductance(q[0], q[1], base + T_C_GC);
            r.capHistory(q[0], q[1], base + T_C_IH);
        }
        for (int e = 0; e < ir.n_elems; ++e) {                       // 4) inductors
            if (ir.kind[e] != CSIM_L) continue;
            const int* q = ir.eq + 4 * e;
            const int base = pl.termBase[static_cast<std::size_t>(e)];
            const int k = ir.branch_eq[e];
            if (!validBranch(k)) continue;
            r.incidence(q[0], q[1], k, base + T_L_ONE);
            r.G(k, k, base + T_L_REQ, true);
            r.I(k, base + T_L_VH, false);
        }
        for (int e = 0; e < ir.n_elems; ++e) {                       // 5) MOS parasitics
            if (ir.kind[e] != CSIM_NMOS && ir.kind[e] != CSIM_PMOS) continue;
            const int* q = ir.eq + 4 * e;
            const int base = pl.termBase[static_cast<std::size_t>(e)];
            const int D = q[0], Gt = q[1], S = q[2], Bk = q[3];
            r.conductance(Gt, S, base + T_M_GCH);  r.capHistory(Gt, S, base + T_M_IHGS);
            r.conductance(Gt, D, base + T_M_GCH);  r.capHistory(Gt, D, base + T_M_IHGD);
            r.conductance(S, Bk, base + T_M_GCF);  r.capHistory(S, Bk, base + T_M_IHSB);
            r.conductance(D, Bk, base + T_M_GCF);  r.capHistory(D, Bk, base + T_M_IHDB);
        }
        for (int eq = 0; eq < ir.n_node_eq; ++eq) r.G(eq, eq, pl.termGmin, false);   // 6) gmin
        group(r.g, pl.tran.gPtr, pl.tran.gPos, pl.tran.gCon);
        group(r.i, pl.tran.iPtr, pl.tran.iRow, pl.tran.iCon);
    }

    auto pattern = [&](const GatherPlan& g, std::vector<uint8_t>& pat) {
        pat.assign(static_cast<std::size_t>(pl.N) * static_cast<std::size_t>(pl.N), 0);
        for (int32_t pos : g.gPos) {
            const int rr = pos / pl.LD, cc = pos % pl.LD;
            pat[static_cast<std::size_t>(rr) * static_cast<std::size_t>(pl.N) + static_cast<std::size_t>(cc)] = 1;
        }
    };
    pattern(pl.dc, pl.patDc);
    pattern(pl.tran, pl.patTran);
    return pl;
}

} // namespace csim
