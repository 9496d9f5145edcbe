// group_plan.cpp -- see group_plan.hpp: plan builder and host interpreter.
#include "group_plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>

namespace csim {

namespace {

struct Kind {
    enum { ZERO, CONST, DYN } k = ZERO;
    double c = 0.0;
    int lvl = 0;          // what a DYN value depends on: 1 = launch-constant terms, 2 = per-step terms, 3 = the iterate
    bool zero() const { return k == ZERO; }
};

// kind of stamp a MOS channel term is, from (term offset, negate): see GroupPlan::mosDest
int mosStampKind(int termOffset, bool neg)
{
    switch (termOffset) {
        case T_M_GD: return neg ? 4 : 0;
        case T_M_GG: return neg ? 5 : 1;
        case T_M_GS: return neg ? 6 : 2;
        case T_M_CST: return neg ? 3 : 7;
        default: return -1;
    }
}

} // namespace

bool buildGroupPlan(const csim_ir& ir, const AssemblyPlan& ap, const PivotSchedule& sch, GroupPlan& gp,
                    const GroupPlan* placement, int lanes)
{
    const int N = ir.n_unknowns;
    gp = GroupPlan();
    const int G = lanes;                                     // lanes per instance: 16 (a DPP row) or 4 (a quad)
    if (G != 16 && G != 4) return false;
    const unsigned laneAll = (1u << G) - 1u;
    gp.G = G;
    // sixteen lanes: up to six rows per lane (N <= 96).  From five rows on the kernel spills (N = 65: 216 registers, N = 95
    // more) and takes the small LDS image of the four-lane kernel; it still beats the lane-per-instance kernel threefold at
    // B = 4096 (N = 65: 2.8e8 against 7.6e7, N = 95: 4.5e7 against ~1.4e7) and the general kernel by 13 ... 50.
    if (N <= 0 || N > (G == 16 ? 6 : 8) * G) return false;
    if (static_cast<int>(sch.pivotPos.size()) != N) return false;
    gp.N = N;
    gp.S = (N + G - 1) / G;
    const int S = gp.S, LD = ap.LD;
    const GatherPlan& g = ap.tran;

    // ---- which terms belong to a MOSFET's channel linearisation (they change every iteration)
    std::vector<int> mosOfTerm(static_cast<std::size_t>(ap.nTerms), -1), offOfTerm(static_cast<std::size_t>(ap.nTerms), -1);
    for (int e = 0; e < ir.n_elems; ++e) {
        if (ir.kind[e] != CSIM_NMOS && ir.kind[e] != CSIM_PMOS) continue;
        const int m = static_cast<int>(gp.mosElem.size());
        gp.mosElem.push_back(e);
        for (int o = T_M_GD; o <= T_M_CST; ++o) {
            mosOfTerm[static_cast<std::size_t>(ap.termBase[static_cast<std::size_t>(e)] + o)] = m;
            offOfTerm[static_cast<std::size_t>(ap.termBase[static_cast<std::size_t>(e)] + o)] = o;
        }
    }
    // exact constants: the global ONE and the inductor incidence (precondition L > 0, checked per instance)
    std::vector<char> termIsOne(static_cast<std::size_t>(ap.nTerms), 0);
    termIsOne[static_cast<std::size_t>(ap.termOne)] = 1;
    for (int e = 0; e < ir.n_elems; ++e)
        if (ir.kind[e] == CSIM_L) termIsOne[static_cast<std::size_t>(ap.termBase[static_cast<std::size_t>(e)] + T_L_ONE)] = 1;

    // what each term depends on (see Kind::lvl)
    std::vector<int> termLevel(static_cast<std::size_t>(ap.nTerms), 1);
    for (int e = 0; e < ir.n_elems; ++e) {
        const int tb = ap.termBase[static_cast<std::size_t>(e)];
        auto set = [&](int off, int lvl) { termLevel[static_cast<std::size_t>(tb + off)] = lvl; };
        switch (ir.kind[e]) {
            case CSIM_V: case CSIM_I: set(T_SRC_VAL, 2); break;
            case CSIM_C: set(T_C_IH, 2); break;
            case CSIM_L: set(T_L_VH, 2); break;
            case CSIM_NMOS: case CSIM_PMOS:
                for (int o = T_M_GD; o <= T_M_CST; ++o) set(o, 3);
                for (int o = T_M_IHGS; o <= T_M_IHDB; ++o) set(o, 2);
                break;
            default: break;
        }
    }

    // ---- abstract matrix [row][col], col N = rhs
    std::vector<std::vector<Kind>> M(static_cast<std::size_t>(N), std::vector<Kind>(static_cast<std::size_t>(N + 1)));
    auto kindOf = [&](const int32_t* con, int n) {
        Kind r;
        if (n == 0) return r;
        bool allConst = true;
        double acc = 0.0;
        int lvl = 0;
        for (int c = 0; c < n; ++c) {
            const int t = con[c] >> 1;
            if (!termIsOne[static_cast<std::size_t>(t)]) { allConst = false; lvl = std::max(lvl, termLevel[static_cast<std::size_t>(t)]); }
            acc = acc + ((con[c] & 1) ? -1.0 : 1.0);
        }
        if (!allConst) { r.k = Kind::DYN; r.lvl = lvl; return r; }
        if (acc != 0.0) { r.k = Kind::CONST; r.c = acc; }
        return r;
    };
    for (int n = 0; n < g.nnzG(); ++n) {
        const int pos = g.gPos[static_cast<std::size_t>(n)];
        M[static_cast<std::size_t>(pos / LD)][static_cast<std::size_t>(pos % LD)] =
            kindOf(&g.gCon[static_cast<std::size_t>(g.gPtr[static_cast<std::size_t>(n)])],
                   g.gPtr[static_cast<std::size_t>(n + 1)] - g.gPtr[static_cast<std::size_t>(n)]);
    }
    for (int n = 0; n < g.nnzI(); ++n)
        M[static_cast<std::size_t>(g.iRow[static_cast<std::size_t>(n)])][static_cast<std::size_t>(N)] =
            kindOf(&g.iCon[static_cast<std::size_t>(g.iPtr[static_cast<std::size_t>(n)])],
                   g.iPtr[static_cast<std::size_t>(n + 1)] - g.iPtr[static_cast<std::size_t>(n)]);

    // ---- placement: replay the swaps
    std::vector<int> cur(static_cast<std::size_t>(N));
    for (int i = 0; i < N; ++i) cur[static_cast<std::size_t>(i)] = i;
    gp.finalPos.assign(static_cast<std::size_t>(N), -1);
    gp.rowAtPos.assign(static_cast<std::size_t>(N), -1);
    std::vector<int> pivotStep(static_cast<std::size_t>(N), -1);         // original row -> column it is the pivot of
    {
        std::vector<int> c2 = cur;
        for (int k = 0; k < N; ++k) {
            const int p = sch.pivotPos[static_cast<std::size_t>(k)];
            if (p < k || p >= N) return false;
            std::swap(c2[static_cast<std::size_t>(k)], c2[static_cast<std::size_t>(p)]);
            pivotStep[static_cast<std::size_t>(c2[static_cast<std::size_t>(k)])] = k;
        }
    }
    if (placement) {
        if (placement->N != N) return false;
        gp.finalPos = placement->finalPos;
        gp.rowAtPos = placement->rowAtPos;
    } else {
        for (int r = 0; r < N; ++r) { gp.finalPos[static_cast<std::size_t>(r)] = pivotStep[static_cast<std::size_t>(r)]; gp.rowAtPos[static_cast<std::size_t>(pivotStep[static_cast<std::size_t>(r)])] = r; }
    }
    auto slotOf = [&](int row) { return gp.finalPos[static_cast<std::size_t>(row)] / G; };
    auto laneOf = [&](int row) { return gp.finalPos[static_cast<std::size_t>(row)] % G; };

    gp.classLive.assign(static_cast<std::size_t>(S), std::vector<uint8_t>(static_cast<std::size_t>(N + 1), 0));
    for (int s = 0; s < S; ++s) gp.classLive[static_cast<std::size_t>(s)][static_cast<std::size_t>(N)] = 1;   // right-hand sides always exist
    auto markLive = [&]() {
        for (int r = 0; r < N; ++r)
            for (int c = 0; c <= N; ++c)
                if (!M[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)].zero())
                    gp.classLive[static_cast<std::size_t>(slotOf(r))][static_cast<std::size_t>(c)] = 1;
    };
    markLive();

    // ---- symbolic elimination with the recorded pivots (solver.hpp:46-77)
    gp.cols.resize(static_cast<std::size_t>(N));
    for (int k = 0; k < N; ++k) {
        GroupPlan::Column& col = gp.cols[static_cast<std::size_t>(k)];
        const int p = sch.pivotPos[static_cast<std::size_t>(k)];
        const int P = cur[static_cast<std::size_t>(p)];
        const Kind pv = M[static_cast<std::size_t>(P)][static_cast<std::size_t>(k)];
        if (pv.zero()) col.zeroPivot = true;
        col.pivotConst = pv.k == Kind::CONST;
        col.pivotValue = pv.c;
        col.pivLane = laneOf(P);
        col.pivSlot = slotOf(P);
        // rows still unpivoted after this column, per slot (an update or a candidate test may touch only those)
        auto maskOf = [&](int s) {
            GroupPlan::Column::SlotMask m;
            bool anyGone = false;
            for (int lane = 0; lane < G; ++lane) {
                const int pos = s * G + lane;
                if (pos >= N) { m.lanes |= 1u << lane; continue; }            // no row there: its registers are exact zeros
                const int R2 = gp.rowAtPos[static_cast<std::size_t>(pos)];
                if (pivotStep[static_cast<std::size_t>(R2)] > k) m.lanes |= 1u << lane;
                else anyGone = true;
            }
            m.keepAll = !anyGone;
            for (int t = 0; t < G && !m.keepAll; ++t)
                if (m.lanes == ((laneAll << (t + 1)) & laneAll)) m.suffix = t;
            return m;
        };
        col.checkMask.resize(static_cast<std::size_t>(S));
        for (int s = 0; s < S; ++s) col.checkMask[static_cast<std::size_t>(s)] = maskOf(s);
        // the reference picks the FIRST row attaining the column maximum (solver.hpp:48-56): rows before the
        // scheduled one must be strictly smaller, rows after it not larger; tiny pivot fails (:58-61)
        std::map<std::pair<int, bool>, unsigned> groups;
        if (!col.zeroPivot) {
            if (col.pivotConst && std::fabs(pv.c) < ir.k.lu_eps) col.contradiction = true;
            for (int i = k; i < N; ++i) {
                if (i == p) continue;
                const int R = cur[static_cast<std::size_t>(i)];
                const Kind a = M[static_cast<std::size_t>(R)][static_cast<std::size_t>(k)];
                if (a.zero()) continue;
                const bool strict = i < p;
                if (a.k == Kind::CONST && pv.k == Kind::CONST) {
                    const bool ok = strict ? std::fabs(pv.c) > std::fabs(a.c) : std::fabs(pv.c) >= std::fabs(a.c);
                    if (!ok) col.contradiction = true;
                    continue;
                }
                groups[{slotOf(R), strict}] |= 1u << laneOf(R);
            }
        }
        for (const auto& kv : groups) col.checks.push_back({kv.first.first, kv.first.second, kv.second});
        gp.nCmp += static_cast<int>(col.checks.size()) + (col.pivotConst ? 0 : 1);
        if (!col.pivotConst && !col.zeroPivot) ++gp.nRecip;
        std::swap(cur[static_cast<std::size_t>(k)], cur[static_cast<std::size_t>(p)]);
        // pivot-row entries right of the diagonal
        for (int j = k + 1; j <= N; ++j) {
            const Kind u = M[static_cast<std::size_t>(P)][static_cast<std::size_t>(j)];
            if (u.zero()) continue;
            col.u.push_back({j, u.k == Kind::CONST, u.c});
            if (u.k != Kind::CONST) ++gp.nBcast;
        }
        ++gp.nBcast;                                     // the pivot itself
        for (const GroupPlan::UEntry& u : col.u)
            if (!u.isConst) gp.opsByLevel[static_cast<std::size_t>(M[static_cast<std::size_t>(P)][static_cast<std::size_t>(u.j)].lvl)] += 1;   // its broadcast
        // instructions of this column by what they depend on (slot granularity: one instruction serves the 16
        // rows of a slot, so it is as variable as its most variable row)
        auto clsLevel = [&](int s2, int j2) {
            int l = 0;
            for (int R2 = 0; R2 < N; ++R2)
                if (slotOf(R2) == s2 && !M[static_cast<std::size_t>(R2)][static_cast<std::size_t>(j2)].zero())
                    l = std::max(l, M[static_cast<std::size_t>(R2)][static_cast<std::size_t>(j2)].k == Kind::CONST ? 0 : M[static_cast<std::size_t>(R2)][static_cast<std::size_t>(j2)].lvl);
            return l;
        };
        const int pvL = pv.k == Kind::CONST ? 0 : pv.lvl;
        if (!col.pivotConst && !col.zeroPivot) gp.opsByLevel[static_cast<std::size_t>(pvL)] += 1 + 5 + 2;     // broadcast, reciprocal, eps test
        for (const GroupPlan::Check& c : col.checks) gp.opsByLevel[static_cast<std::size_t>(std::max(pvL, clsLevel(c.slot, k)))] += 2;
        std::vector<int> fLevel(static_cast<std::size_t>(S), 0);
        for (int s2 = 0; s2 < S; ++s2) fLevel[static_cast<std::size_t>(s2)] = std::max(pvL, clsLevel(s2, k));
        std::vector<char> slotHasL(static_cast<std::size_t>(S), 0);
        for (int i = k + 1; i < N; ++i) {
            const int R = cur[static_cast<std::size_t>(i)];
            Kind& a = M[static_cast<std::size_t>(R)][static_cast<std::size_t>(k)];
            if (a.zero()) continue;
            slotHasL[static_cast<std::size_t>(slotOf(R))] = 1;
            Kind f;                                                  // multiplier a / pivot (solver.hpp:71)
            if (a.k == Kind::CONST && pv.k == Kind::CONST) { f.k = Kind::CONST; f.c = a.c * (1.0 / pv.c); }
            else f.k = Kind::DYN;
            f.lvl = std::max(a.k == Kind::CONST ? 0 : a.lvl, pvL);
            for (const GroupPlan::UEntry& u : col.u) {
                Kind& t = M[static_cast<std::size_t>(R)][static_cast<std::size_t>(u.j)];
                const Kind& uk = M[static_cast<std::size_t>(P)][static_cast<std::size_t>(u.j)];
                const int newLvl = std::max(std::max(f.lvl, uk.k == Kind::CONST ? 0 : uk.lvl), t.k == Kind::DYN ? t.lvl : 0);
                if (f.k == Kind::CONST && u.isConst && t.k != Kind::DYN) {
                    const double v = (t.k == Kind::CONST ? t.c : 0.0) - f.c * u.c;
                    t.k = v == 0.0 ? Kind::ZERO : Kind::CONST;
                    t.c = v;
                    // a value that cancels exactly in the generator still occupies its register class
                    gp.classLive[static_cast<std::size_t>(slotOf(R))][static_cast<std::size_t>(u.j)] = 1;
                } else {
                    t.k = Kind::DYN;
                    t.lvl = newLvl;
                }
            }
            a = Kind();                                              // below the diagonal: never read again
        }
        for (int s = 0; s < S; ++s)
            if (slotHasL[static_cast<std::size_t>(s)]) {
                gp.opsByLevel[static_cast<std::size_t>(fLevel[static_cast<std::size_t>(s)])] += 1;                 // multiplier
                for (const GroupPlan::UEntry& u : col.u) {
                    const Kind& uk = M[static_cast<std::size_t>(P)][static_cast<std::size_t>(u.j)];
                    const int ul = uk.k == Kind::CONST ? 0 : uk.lvl;
                    // after the update loop above clsLevel(s, u.j) already includes this column's contribution
                    gp.opsByLevel[static_cast<std::size_t>(std::max(std::max(fLevel[static_cast<std::size_t>(s)], ul), clsLevel(s, u.j)))] += 1;
                }
                col.lSlots.push_back(s);
                col.lMask.push_back(col.checkMask[static_cast<std::size_t>(s)]);
                gp.nMul += col.lMask.back().keepAll ? 1 : 2;
                gp.nFma += static_cast<int>(col.u.size());
            }
        markLive();
    }

    // ---- back substitution (solver.hpp:116-128), column-wise
    gp.backSlots.assign(static_cast<std::size_t>(N), {});
    for (int j = 0; j < N; ++j) {
        std::vector<char> has(static_cast<std::size_t>(S), 0);
        for (int R = 0; R < N; ++R)
            if (pivotStep[static_cast<std::size_t>(R)] < j && !M[static_cast<std::size_t>(R)][static_cast<std::size_t>(j)].zero())
                has[static_cast<std::size_t>(slotOf(R))] = 1;
        for (int s = 0; s < S; ++s)
            if (has[static_cast<std::size_t>(s)]) { gp.backSlots[static_cast<std::size_t>(j)].push_back(s); ++gp.nFma; }
        gp.nMul += 1;
        gp.nBcast += 1;
    }

    // ---- assembly tables
    // matrix cells: terms that are constant over a launch, in stamping order; MOS terms go to staging rows
    std::map<std::pair<int, int>, int> classIndex;                      // (s, j) -> index in gClasses
    std::vector<std::vector<int32_t>> cellCon;                           // [class*16 + lane]
    std::map<std::tuple<int, int, int>, int> stageIndex;                 // (s, j, round) -> staging row
    gp.mosDest.assign(gp.mosElem.size(), {{-1, -1, -1, -1, -1, -1, -1, -1}});
    auto addCell = [&](int row, int colj, const int32_t* con, int n, bool rhs) {
        const int s = slotOf(row), lane = laneOf(row);
        int round = 0;
        for (int c = 0; c < n; ++c) {
            const int t = con[c] >> 1;
            const bool neg = (con[c] & 1) != 0;
            const int m = mosOfTerm[static_cast<std::size_t>(t)];
            if (m >= 0) {
                const auto key = std::make_tuple(s, colj, round++);
                auto it = stageIndex.find(key);
                if (it == stageIndex.end()) {
                    it = stageIndex.emplace(key, static_cast<int>(gp.stageRows.size())).first;
                    gp.stageRows.push_back({s, colj});
                }
                const int kind = mosStampKind(offOfTerm[static_cast<std::size_t>(t)], neg);
                gp.mosDest[static_cast<std::size_t>(m)][static_cast<std::size_t>(kind)] = it->second * G + lane;
                continue;
            }
            if (rhs) continue;                                           // handled below (per-step gather)
            auto ci = classIndex.find({s, colj});
            if (ci == classIndex.end()) {
                ci = classIndex.emplace(std::make_pair(s, colj), static_cast<int>(gp.gClasses.size())).first;
                gp.gClasses.push_back({s, colj});
                cellCon.resize(gp.gClasses.size() * G);
            }
            cellCon[static_cast<std::size_t>(ci->second * G + lane)].push_back(con[c]);
        }
    };
    for (int n = 0; n < g.nnzG(); ++n) {
        const int pos = g.gPos[static_cast<std::size_t>(n)];
        addCell(pos / LD, pos % LD, &g.gCon[static_cast<std::size_t>(g.gPtr[static_cast<std::size_t>(n)])],
                g.gPtr[static_cast<std::size_t>(n + 1)] - g.gPtr[static_cast<std::size_t>(n)], false);
    }
    std::vector<std::vector<int32_t>> rhsCon(static_cast<std::size_t>(S * G));
    for (int n = 0; n < g.nnzI(); ++n) {
        const int row = g.iRow[static_cast<std::size_t>(n)];
        const int32_t* con = &g.iCon[static_cast<std::size_t>(g.iPtr[static_cast<std::size_t>(n)])];
        const int cnt = g.iPtr[static_cast<std::size_t>(n + 1)] - g.iPtr[static_cast<std::size_t>(n)];
        addCell(row, N, con, cnt, true);
        for (int c = 0; c < cnt; ++c)
            if (mosOfTerm[static_cast<std::size_t>(con[c] >> 1)] < 0)
                rhsCon[static_cast<std::size_t>(slotOf(row) * G + laneOf(row))].push_back(con[c]);
    }
    // staging rows are added in the order (class, round): sort them so, and remap
    {
        std::vector<int> order(gp.stageRows.size());
        for (std::size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
        std::vector<std::tuple<int, int, int>> keys(gp.stageRows.size());
        for (const auto& kv : stageIndex) keys[static_cast<std::size_t>(kv.second)] = kv.first;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return keys[static_cast<std::size_t>(a)] < keys[static_cast<std::size_t>(b)]; });
        std::vector<int> newOf(order.size());
        std::vector<GroupPlan::StageRow> rows(order.size());
        for (std::size_t i = 0; i < order.size(); ++i) { newOf[static_cast<std::size_t>(order[i])] = static_cast<int>(i); rows[i] = gp.stageRows[static_cast<std::size_t>(order[i])]; }
        gp.stageRows = rows;
        for (auto& d : gp.mosDest)
            for (int& cell : d)
                if (cell >= 0) cell = newOf[static_cast<std::size_t>(cell / G)] * G + cell % G;
    }
    // ---- critical path of one solve (dependent-issue latency model: every VALU result is usable lat
    // cycles after issue; the Newton-refined reciprocal is a chain of 1 + 4 operations)
    {
        const int lat = 8, latRcp = 5 * 8;
        std::vector<std::vector<int>> t(static_cast<std::size_t>(S), std::vector<int>(static_cast<std::size_t>(N + 1), 0));
        int worst = 0;
        std::vector<int> tr(static_cast<std::size_t>(N), 0);
        for (int k = 0; k < N; ++k) {
            const GroupPlan::Column& col = gp.cols[static_cast<std::size_t>(k)];
            const int sk = col.pivSlot;
            int r = 0;
            if (!col.pivotConst) r = t[static_cast<std::size_t>(sk)][static_cast<std::size_t>(k)] + lat + latRcp;
            tr[static_cast<std::size_t>(k)] = r;
            std::vector<int> tf(static_cast<std::size_t>(S), 0);
            for (std::size_t li = 0; li < col.lSlots.size(); ++li) {
                const int s = col.lSlots[li];
                const int rr = (!col.lMask[li].keepAll && !col.pivotConst) ? r + lat : r;      // r * mask
                tf[static_cast<std::size_t>(s)] = std::max(t[static_cast<std::size_t>(s)][static_cast<std::size_t>(k)], rr) + lat;
            }
            for (const GroupPlan::UEntry& u : col.u) {
                const int tu = u.isConst ? 0 : t[static_cast<std::size_t>(sk)][static_cast<std::size_t>(u.j)] + lat;
                for (int s : col.lSlots) {
                    int& d = t[static_cast<std::size_t>(s)][static_cast<std::size_t>(u.j)];
                    d = std::max(d, std::max(tf[static_cast<std::size_t>(s)], tu)) + lat;
                }
            }
        }
        gp.depthElimination = 0;
        for (int s = 0; s < S; ++s) for (int j = 0; j <= N; ++j) gp.depthElimination = std::max(gp.depthElimination, t[static_cast<std::size_t>(s)][static_cast<std::size_t>(j)]);
        for (int j = N - 1; j >= 0; --j) {
            const int sj = gp.cols[static_cast<std::size_t>(j)].pivSlot;
            const int tx = std::max(t[static_cast<std::size_t>(sj)][static_cast<std::size_t>(N)], tr[static_cast<std::size_t>(j)]) + lat + lat;
            worst = std::max(worst, tx);
            for (int s : gp.backSlots[static_cast<std::size_t>(j)]) {
                int& d = t[static_cast<std::size_t>(s)][static_cast<std::size_t>(N)];
                d = std::max(d, std::max(tx, t[static_cast<std::size_t>(s)][static_cast<std::size_t>(j)])) + lat;
            }
        }
        gp.depthSolve = worst;
    }

    gp.gCellPtr.assign(1, 0);
    for (const auto& c : cellCon) {
        gp.gCellCon.insert(gp.gCellCon.end(), c.begin(), c.end());
        gp.gCellPtr.push_back(static_cast<int32_t>(gp.gCellCon.size()));
    }
    gp.iCellPtr.assign(1, 0);
    for (const auto& c : rhsCon) {
        gp.iCellCon.insert(gp.iCellCon.end(), c.begin(), c.end());
        gp.iCellPtr.push_back(static_cast<int32_t>(gp.iCellCon.size()));
    }
    return true;
}

// ---------------------------------------------------------------- placement search
// wave-level instructions of one solve as the emitter will write it (broadcasts of a quad cost two moves)
static double placementCost(const GroupPlan& gp)
{
    double c = (gp.G == 4 ? 2.0 : 1.0) * gp.nBcast + gp.nFma + gp.nMul + gp.nCmp + 2.0 * static_cast<double>(gp.stageRows.size());
    int live = 0;
    for (const auto& row : gp.classLive) for (uint8_t v : row) live += v ? 1 : 0;
    c += 0.5 * live;                                         // every class is assembled once per solve (and costs registers)
    for (const GroupPlan::Column& col : gp.cols)
        for (const GroupPlan::Column::SlotMask& m : col.lMask)
            if (!m.keepAll && m.suffix < 0) c += gp.G == 4 ? 1.0 : 4.0;   // an explicit lane mask instead of a launch-constant prefix factor
                                                                          // (a quad has 14 such masks, they stay in registers; a row of 16 does not)
    return c;
}

bool optimizeGroupPlacement(const csim_ir& ir, const AssemblyPlan& ap, const std::vector<PivotSchedule>& schedules, int lanes,
                            GroupPlan& placement, double* costBefore, double* costAfter)
{
    if (schedules.empty()) return false;
    GroupPlan first;
    if (!buildGroupPlan(ir, ap, schedules[0], first, nullptr, lanes)) return false;
    const int N = first.N;
    placement = GroupPlan();
    placement.N = N;
    placement.G = lanes;
    placement.finalPos = first.finalPos;
    placement.rowAtPos = first.rowAtPos;
    // (the staging rows live in LDS, and the four-lane kernel's image is sized to the last double: never more of them
    // than the position-cyclic placement needs)
    const std::size_t maxStage = first.stageRows.size();
    auto total = [&](const GroupPlan& pl, double* out) {
        double c = 0.0;
        for (std::size_t a = 0; a < schedules.size(); ++a) {
            GroupPlan gp;
            if (!buildGroupPlan(ir, ap, schedules[a], gp, &pl, lanes)) return false;
            if (gp.stageRows.size() > maxStage) return false;
            c += (a == 0 ? 1.0 : 0.25) * placementCost(gp);  // the first schedule is the one most solves take
        }
        *out = c;
        return true;
    };
    double best = 0.0;
    if (!total(placement, &best)) return false;
    if (costBefore) *costBefore = best;
    // first-improvement local search over exchanges of two rows' cells; deterministic
    for (int sweep = 0; sweep < 12; ++sweep) {
        bool improved = false;
        for (int c1 = 0; c1 < N; ++c1)
            for (int c2 = c1 + 1; c2 < N; ++c2) {
                const int r1 = placement.rowAtPos[static_cast<std::size_t>(c1)], r2 = placement.rowAtPos[static_cast<std::size_t>(c2)];
                std::swap(placement.rowAtPos[static_cast<std::size_t>(c1)], placement.rowAtPos[static_cast<std::size_t>(c2)]);
                placement.finalPos[static_cast<std::size_t>(r1)] = c2;
                placement.finalPos[static_cast<std::size_t>(r2)] = c1;
                double c = 0.0;
                if (total(placement, &c) && c < best - 1e-9) { best = c; improved = true; continue; }
                std::swap(placement.rowAtPos[static_cast<std::size_t>(c1)], placement.rowAtPos[static_cast<std::size_t>(c2)]);
                placement.finalPos[static_cast<std::size_t>(r1)] = c1;
                placement.finalPos[static_cast<std::size_t>(r2)] = c2;
            }
        if (!improved) break;
    }
    if (costAfter) *costAfter = best;
    return true;
}

// ---------------------------------------------------------------- interpreter
void interpretGroupPlan(const GroupPlan& gp, const AssemblyPlan& ap, const csim_ir& ir, const double* T, double eps,
                        double* x, bool* violated, bool* planError)
{
    bool wrongPlan = false;
    (void)ir;
    const int N = gp.N, S = gp.S, G = gp.G;
    // registers a[s][j][lane]
    std::vector<double> a(static_cast<std::size_t>(S) * (N + 1) * G, 0.0);
    auto A = [&](int s, int j, int lane) -> double& { return a[(static_cast<std::size_t>(s) * (N + 1) + j) * G + lane]; };
    // (a) launch/step-constant parts
    for (std::size_t c = 0; c < gp.gClasses.size(); ++c)
        for (int lane = 0; lane < G; ++lane) {
            double acc = 0.0;
            for (int t = gp.gCellPtr[c * G + lane]; t < gp.gCellPtr[c * G + lane + 1]; ++t) {
                const int con = gp.gCellCon[static_cast<std::size_t>(t)];
                acc = (con & 1) ? acc - T[con >> 1] : acc + T[con >> 1];
            }
            A(gp.gClasses[c].s, gp.gClasses[c].j, lane) = acc;
        }
    for (int s = 0; s < S; ++s)
        for (int lane = 0; lane < G; ++lane) {
            double acc = 0.0;
            for (int t = gp.iCellPtr[static_cast<std::size_t>(s * G + lane)]; t < gp.iCellPtr[static_cast<std::size_t>(s * G + lane + 1)]; ++t) {
                const int con = gp.iCellCon[static_cast<std::size_t>(t)];
                acc = (con & 1) ? acc - T[con >> 1] : acc + T[con >> 1];
            }
            A(s, N, lane) = acc;
        }
    // (b) MOS staging: lane m scatters, owners add row by row
    std::vector<double> stage(gp.stageRows.size() * G, 0.0);
    for (std::size_t m = 0; m < gp.mosElem.size(); ++m) {
        const int tb = ap.termBase[static_cast<std::size_t>(gp.mosElem[m])];
        const double gd = T[tb + T_M_GD], gg = T[tb + T_M_GG], gs = T[tb + T_M_GS], cst = T[tb + T_M_CST];
        const double val[8] = {gd, gg, gs, -cst, -gd, -gg, -gs, cst};
        for (int kd = 0; kd < 8; ++kd)
            if (gp.mosDest[m][static_cast<std::size_t>(kd)] >= 0) stage[static_cast<std::size_t>(gp.mosDest[m][static_cast<std::size_t>(kd)])] = val[kd];
    }
    for (std::size_t r = 0; r < gp.stageRows.size(); ++r)
        for (int lane = 0; lane < G; ++lane) A(gp.stageRows[r].s, gp.stageRows[r].j, lane) += stage[r * G + lane];

    bool bad = false;
    for (int k = 0; k < N; ++k) {
        const GroupPlan::Column& col = gp.cols[static_cast<std::size_t>(k)];
        const int sk = col.pivSlot, lk = col.pivLane;
        if (col.zeroPivot || col.contradiction) { bad = true; continue; }
        const double pb = A(sk, k, lk);                                   // broadcast
        if (!col.pivotConst && !(std::fabs(pb) >= eps)) bad = true;
        for (const GroupPlan::Check& c : col.checks)
            for (int lane = 0; lane < G; ++lane)
                if (c.laneMask & (1u << lane)) {
                    const double v = std::fabs(A(c.slot, k, lane));
                    if (!(c.strict ? std::fabs(pb) > v : std::fabs(pb) >= v)) bad = true;
                }
        // what the kernel tests on EVERY lane of the slots that hold candidates must not fire on the others:
        // finished rows are masked, rows without an entry hold an exact zero
        for (int s = 0; s < S; ++s) {
            bool any = false;
            for (const GroupPlan::Check& c : col.checks) any = any || c.slot == s;
            if (!any) continue;
            for (int lane = 0; lane < G; ++lane) {
                bool isCand = false;
                for (const GroupPlan::Check& c : col.checks) isCand = isCand || (c.slot == s && (c.laneMask & (1u << lane)));
                const bool kept = col.checkMask[static_cast<std::size_t>(s)].keepAll || (col.checkMask[static_cast<std::size_t>(s)].lanes & (1u << lane));
                if (!isCand && kept && !(s == sk && lane == lk) && A(s, k, lane) != 0.0) wrongPlan = true;   // would be tested
            }
        }
        const double r = 1.0 / pb;
        for (std::size_t li = 0; li < col.lSlots.size(); ++li) {
            const int s = col.lSlots[li];
            for (int lane = 0; lane < G; ++lane) {
                const bool keep = col.lMask[li].keepAll || (col.lMask[li].lanes & (1u << lane));
                const double f = A(s, k, lane) * (r * (keep ? 1.0 : 0.0));
                for (const GroupPlan::UEntry& u : col.u) {
                    const double ub = u.isConst ? u.c : A(sk, u.j, lk);
                    A(s, u.j, lane) = A(s, u.j, lane) - f * ub;
                }
            }
        }
    }
    for (int j = N - 1; j >= 0; --j) {
        const GroupPlan::Column& col = gp.cols[static_cast<std::size_t>(j)];
        const int sj = col.pivSlot, lj = col.pivLane;
        const double piv = A(sj, j, lj);
        const double xj = (col.zeroPivot || col.contradiction) ? 0.0 : A(sj, N, lj) * (1.0 / piv);
        x[j] = xj;
        for (int s : gp.backSlots[static_cast<std::size_t>(j)])
            for (int lane = 0; lane < G; ++lane) A(s, N, lane) = A(s, N, lane) - A(s, j, lane) * xj;
    }
    if (violated) *violated = bad;
    if (planError) *planError = wrongPlan;
}

} // namespace csim
