// codegen.cpp -- see codegen.hpp.  Symbolic execution of stamp + pivoted LU +
// substitution over {zero, exact constant, run-time value}; emits HIP source.
#include "codegen.hpp"
#include "group_plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <sstream>

namespace csim {

// codegen_linear.cpp
void emitTranSourceValue(std::ostream& src, const std::string& i2, const csim_ir& ir, int e,
                         const std::function<std::string(int)>& P, const std::string& target);

// ------------------------------------------------------------ schedule utils

PivotSchedule PivotSchedule::identity(int N)
{
    PivotSchedule s;
    s.pivotPos.resize(static_cast<std::size_t>(N));
    for (int k = 0; k < N; ++k) s.pivotPos[static_cast<std::size_t>(k)] = k;
    return s;
}

bool PivotSchedule::parse(const std::string& text, int N, PivotSchedule& out)
{
    out = identity(N);
    std::size_t i = 0;
    while (i < text.size()) {
        while (i < text.size() && (text[i] == ',' || text[i] == ' ' || text[i] == '\n' || text[i] == '\t')) ++i;
        if (i >= text.size()) break;
        int k = 0, p = 0, used = 0;
        if (std::sscanf(text.c_str() + i, "%d:%d%n", &k, &p, &used) != 2) return false;
        if (k < 0 || k >= N || p < k || p >= N) return false;
        out.pivotPos[static_cast<std::size_t>(k)] = p;
        i += static_cast<std::size_t>(used);
    }
    return true;
}

std::string PivotSchedule::str() const
{
    std::ostringstream o;
    bool first = true;
    for (std::size_t k = 0; k < pivotPos.size(); ++k) {
        if (pivotPos[k] == static_cast<int>(k)) continue;
        o << (first ? "" : ",") << k << ":" << pivotPos[k];
        first = false;
    }
    return o.str();
}

bool ScheduleSet::parse(const std::string& text, int N, ScheduleSet& out)
{
    out.alts.clear();
    out.dcAlts.clear();
    std::size_t i = 0;
    while (i <= text.size()) {
        std::size_t e = text.find('\n', i);
        if (e == std::string::npos) e = text.size();
        std::string line = text.substr(i, e - i);
        i = e + 1;
        const std::size_t hashAt = line.find('#');                  // comment to end of line
        if (hashAt != std::string::npos) line = line.substr(0, hashAt);
        std::size_t j = 0;
        while (j <= line.size()) {                                   // ';' separates schedules on one line
            std::size_t f = line.find(';', j);
            if (f == std::string::npos) f = line.size();
            std::string body = line.substr(j, f - j);
            j = f + 1;
            bool blank = true, dash = false;
            for (char ch : body) { blank = blank && (ch == ' ' || ch == '\t' || ch == '\r' || ch == ','); dash = dash || ch == '-'; }
            if (blank) continue;
            bool isDc = false;                                        // "dc <schedule>"
            {
                std::size_t w = 0;
                while (w < body.size() && (body[w] == ' ' || body[w] == '\t')) ++w;
                if (w + 2 <= body.size() && (body[w] == 'd' || body[w] == 'D') && (body[w + 1] == 'c' || body[w + 1] == 'C') &&
                    (w + 2 == body.size() || body[w + 2] == ' ' || body[w + 2] == '\t')) {
                    isDc = true;
                    body = body.substr(w + 2);
                }
            }
            if (dash) for (char& ch : body) if (ch == '-') ch = ' ';  // "-" = no swaps
            PivotSchedule one;
            if (!PivotSchedule::parse(body, N, one)) return false;
            std::vector<PivotSchedule>& dst = isDc ? out.dcAlts : out.alts;
            bool dup = false;
            for (const PivotSchedule& a : dst) dup = dup || a.pivotPos == one.pivotPos;
            if (!dup) dst.push_back(one);
        }
    }
    return !out.alts.empty();
}

std::string ScheduleSet::str() const
{
    std::string o;
    for (std::size_t a = 0; a < alts.size(); ++a) o += (a ? " ; " : "") + (alts[a].str().empty() ? std::string("-") : alts[a].str());
    for (std::size_t a = 0; a < dcAlts.size(); ++a) o += " ; dc " + (dcAlts[a].str().empty() ? std::string("-") : dcAlts[a].str());
    return o;
}

bool GeneratorOptions::set(const std::string& keyval)
{
    const std::size_t eq = keyval.find('=');
    if (eq == std::string::npos) return false;
    const std::string key = keyval.substr(0, eq), val = keyval.substr(eq + 1);
    if (key == "barrier_every") { barrierEvery = std::max(0, std::atoi(val.c_str())); return true; }
    if (key == "stage_ahead") { stageAhead = std::max(-1, std::atoi(val.c_str())); return true; }
    if (key == "pipeline_mos") { pipelineMos = std::atoi(val.c_str()) != 0; return true; }
    if (key == "group_waves") { groupWavesPerEu = std::max(0, std::atoi(val.c_str())); return true; }
    if (key == "near_band") { nearBand = std::max(0.0, std::atof(val.c_str())); return true; }
    if (key == "near_band_dc") { nearBandDc = std::max(0.0, std::atof(val.c_str())); return true; }
    if (key == "near_form") { nearForm = std::atoi(val.c_str()) != 0 ? 1 : 0; return true; }
    if (key == "lds_pad") { ldsPad = std::atoi(val.c_str()) != 0 ? 1 : 0; return true; }
    if (key == "lin_factor_block") { linFactorBlock = std::atoi(val.c_str()); return linFactorBlock == 0 || linFactorBlock == 16 || linFactorBlock == 32 || linFactorBlock == 64; }
    if (key == "lin_src_lds") { linSrcLds = std::atoi(val.c_str()) != 0 ? 1 : 0; return true; }
    if (key == "dummy_one_cell") { dummyOneCell = std::atoi(val.c_str()) != 0 ? 1 : 0; return true; }
    if (key == "place_search") { placeSearch = std::atoi(val.c_str()) & 3; return true; }
    if (key == "group4") { group4 = std::atoi(val.c_str()) != 0 ? 1 : 0; return true; }
    if (key == "lin_chain_barrier") { linChainBarrier = std::atoi(val.c_str()) != 0 ? 1 : 0; return true; }
    if (key == "sweep") {
        sweep.clear();
        std::size_t i = 0;
        while (i < val.size()) { sweep.push_back(std::atoi(val.c_str() + i)); i = val.find(',', i); if (i == std::string::npos) break; ++i; }
        return true;
    }
    return false;
}

uint64_t scheduleHash(const csim_ir& ir, const ScheduleSet& set, const GeneratorOptions& gopt)
{
    uint64_t h = scheduleHash(ir, set.alts.empty() ? PivotSchedule::identity(ir.n_unknowns) : set.alts[0]);
    // everything that changes the emitted code is part of the identity of a generated library
    auto mix = [&h](uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); };
    mix(static_cast<uint64_t>(gopt.barrierEvery) + 1);
    mix(static_cast<uint64_t>(gopt.stageAhead + 2) * 0x100000001b3ull);
    mix(static_cast<uint64_t>(gopt.pipelineMos + 7) * 0x9E3779B1ull);
    mix(static_cast<uint64_t>(gopt.groupWavesPerEu + 11) * 0x85EBCA6Bull);
    for (int v : gopt.sweep) mix(static_cast<uint64_t>(static_cast<int64_t>(v)) ^ 0x5bd1e995ull);
    { uint64_t bits; std::memcpy(&bits, &gopt.nearBand, sizeof bits); mix(bits ^ 0xC2B2AE3D27D4EB4Full); }
    { uint64_t bits; std::memcpy(&bits, &gopt.nearBandDc, sizeof bits); mix(bits ^ 0x165667B19E3779F9ull); }
    mix(static_cast<uint64_t>(gopt.nearForm + 3) * 0x27D4EB2F165667C5ull);
    mix(static_cast<uint64_t>(gopt.ldsPad + 5) * 0x9E3779B185EBCA87ull);
    mix(static_cast<uint64_t>(gopt.linFactorBlock + 13) * 0xC2B2AE3D27D4EB4Full);
    mix(static_cast<uint64_t>(gopt.linSrcLds + 17) * 0x165667B19E3779F9ull);
    mix(static_cast<uint64_t>(gopt.dummyOneCell + 19) * 0x85EBCA77C2B2AE63ull);
    mix(static_cast<uint64_t>(gopt.linChainBarrier + 23) * 0x27D4EB2F165667C5ull);
    mix(static_cast<uint64_t>(gopt.group4 + 29) * 0x9FB21C651E98DF25ull);
    mix(static_cast<uint64_t>(gopt.placeSearch + 31) * 0xD6E8FEB86659FD93ull);
    for (std::size_t a = 1; a < set.alts.size(); ++a) {
        h ^= 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        for (int p : set.alts[a].pivotPos) { h ^= static_cast<uint64_t>(p + 1); h *= 1099511628211ull; }
    }
    for (std::size_t a = 0; a < set.dcAlts.size(); ++a) {
        h ^= 0xD1B54A32D192ED03ull + (h << 6) + (h >> 2);
        for (int p : set.dcAlts[a].pivotPos) { h ^= static_cast<uint64_t>(p + 1); h *= 1099511628211ull; }
    }
    return h;
}

uint64_t scheduleHash(const csim_ir& ir, const PivotSchedule& sch)
{
    uint64_t h = 1469598103934665603ull;                 // FNV-1a
    auto mixBytes = [&h](const void* p, std::size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (std::size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    auto mixInt = [&](int32_t v) { mixBytes(&v, sizeof v); };
    mixInt(kGeneratorRevision);
    mixInt(ir.n_unknowns); mixInt(ir.n_node_eq); mixInt(ir.n_branch_eq);
    mixInt(ir.n_elems); mixInt(ir.n_params); mixInt(ir.has_nonlinear);
    for (int e = 0; e < ir.n_elems; ++e) {
        mixInt(ir.kind[e]);
        for (int t = 0; t < 4; ++t) mixInt(ir.eq[4 * e + t]);
        mixInt(ir.branch_eq[e]); mixInt(ir.param_slot[e]); mixInt(ir.wave[e]);
    }
    mixBytes(&ir.k, sizeof ir.k);
    for (int p : sch.pivotPos) mixInt(p);
    return h;
}

// ------------------------------------------------------------ abstract values

namespace {

struct AV {
    enum Kind { ZERO, CONST, DYN } kind = ZERO;
    double c = 0.0;          // CONST
    std::string v;           // DYN: variable name
    bool neg = false;        // DYN: value is -(v)
    static AV zero() { return AV(); }
    static AV konst(double x) { AV a; if (x == 0.0) return a; a.kind = CONST; a.c = x; return a; }
    static AV dyn(const std::string& name, bool n = false) { AV a; a.kind = DYN; a.v = name; a.neg = n; return a; }
    bool isZero() const { return kind == ZERO; }
};

std::string lit(double x)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%a", x);
    return std::string("(") + buf + ")";
}

struct Gen {
    const csim_ir& ir;
    const AssemblyPlan& ap;
    std::ostringstream out;
    // deferred entries: ordered term lists not yet turned into code (lazy assembly keeps
    // an entry out of the register file until the elimination first touches it)
    std::vector<std::vector<std::vector<AV>>> pending;
    int tmp = 0;
    CodegenStats st;
    std::string ind = "            ";

    Gen(const csim_ir& i, const AssemblyPlan& a) : ir(i), ap(a) {}

    std::string fresh() { return "v" + std::to_string(tmp++); }
    std::string ref(const AV& a) const
    {
        if (a.kind == AV::CONST) return lit(a.c);
        if (a.kind == AV::DYN) return a.neg ? "(-" + a.v + ")" : a.v;
        return "0.0";
    }
    AV emit(const std::string& expr)
    {
        const std::string n = fresh();
        out << ind << "const double " << n << " = " << expr << ";\n";
        return AV::dyn(n);
    }
    AV negate(AV a)
    {
        if (a.kind == AV::CONST) a.c = -a.c;
        else if (a.kind == AV::DYN) a.neg = !a.neg;
        return a;
    }
    // a * b
    AV mul(const AV& a, const AV& b)
    {
        if (a.isZero() || b.isZero()) return AV::zero();
        if (a.kind == AV::CONST && b.kind == AV::CONST) return AV::konst(a.c * b.c);
        if (a.kind == AV::CONST || b.kind == AV::CONST) {
            const AV& k = a.kind == AV::CONST ? a : b;
            const AV& d = a.kind == AV::CONST ? b : a;
            if (k.c == 1.0) return d;
            if (k.c == -1.0) return negate(d);
            ++st.nMul;
            return emit(lit(k.c) + " * " + ref(d));
        }
        ++st.nMul;
        AV r = emit(a.v + " * " + b.v);
        r.neg = a.neg != b.neg;
        return r;
    }
    // a / b (faithful kernels: solver.hpp:71 and :126 divide)
    AV div(const AV& a, const AV& b)
    {
        if (a.isZero()) return AV::zero();
        if (a.kind == AV::CONST && b.kind == AV::CONST) return AV::konst(a.c / b.c);
        if (b.kind == AV::CONST) {
            if (b.c == 1.0) return a;
            if (b.c == -1.0) return negate(a);
            return emit(ref(a) + " / " + lit(b.c));
        }
        AV r = emit((a.kind == AV::CONST ? lit(a.c) : a.v) + " / " + b.v);
        r.neg = (a.kind == AV::DYN && a.neg) != b.neg;
        return r;
    }
    // a - f*u
    AV fnma(const AV& a, const AV& f, const AV& u)
    {
        if (f.isZero() || u.isZero()) return a;
        if (a.isZero()) return negate(mul(f, u));
        // fold exact +-1 factors into an add/sub
        const bool f1 = f.kind == AV::CONST && std::fabs(f.c) == 1.0;
        const bool u1 = u.kind == AV::CONST && std::fabs(u.c) == 1.0;
        if (f.kind == AV::CONST && u.kind == AV::CONST) {
            const double p = f.c * u.c;
            if (a.kind == AV::CONST) return AV::konst(a.c - p);
            ++st.nAddSub;
            return emit(ref(a) + " - " + lit(p));
        }
        if (f1 || u1) {
            AV w = f1 ? u : f;
            const double s = f1 ? f.c : u.c;
            if (s < 0) w = negate(w);
            ++st.nAddSub;
            return emit(ref(a) + " - " + ref(w));
        }
        ++st.nFma;
        return emit(ref(a) + " - " + ref(f) + " * " + ref(u));
    }
    // ordered sum of signed terms (the reference's accumulation order)
    AV orderedSum(const std::vector<AV>& terms)
    {
        bool allConst = true;
        for (const AV& t : terms) allConst = allConst && t.kind != AV::DYN;
        if (allConst) {
            double acc = 0.0;
            for (const AV& t : terms) acc = acc + (t.kind == AV::CONST ? t.c : 0.0);
            return AV::konst(acc);
        }
        std::vector<AV> nz;
        for (const AV& t : terms) if (!t.isZero()) nz.push_back(t);
        if (nz.size() == 1) return nz[0];
        std::string e;
        for (std::size_t i = 0; i < nz.size(); ++i) {
            const AV& t = nz[i];
            if (i == 0) { e = ref(t); continue; }
            if (t.kind == AV::DYN) e = "(" + e + (t.neg ? " - " : " + ") + t.v + ")";
            else e = "(" + e + " + " + lit(t.c) + ")";
            ++st.nAddSub;
        }
        return emit(e);
    }
};

// the iterate x lives in LDS, one private column per lane: X(i) = lds[i*64 + lane]
std::string xname(int eq) { return eq >= 0 ? "X(" + std::to_string(eq) + ")" : std::string("0.0"); }
// register copy of x(eq) made at the top of an iteration
std::string xloc(int eq) { return eq >= 0 ? "xl" + std::to_string(eq) : std::string("0.0"); }
std::string pname(int slot) { return "p" + std::to_string(slot); }
std::string tname(int t) { return "t" + std::to_string(t); }

// difference of two node voltages as an expression
std::string vdiff(int a, int b)
{
    if (a < 0 && b < 0) return "0.0";
    if (b < 0) return xname(a);
    if (a < 0) return "(-" + xname(b) + ")";
    return "(" + xname(a) + " - " + xname(b) + ")";
}

} // namespace

// ------------------------------------------------------------------ generator

namespace {

struct VariantOptions {
    const char* kernelName;
    bool rich;          // launch constants and loop parameters in LDS
    bool stepInLds;     // per-step terms (sources, history currents) in LDS (else registers)
    int parkBudget;     // how many finished U-row values may be parked in LDS (-1 = all)
    int ckUnroll = 4;   // unroll factor of the per-step checkpoint loop (swept 1/2/4/8/31: 4 best, 31 pins 2N VGPRs)
    bool dcMode = false;// emit the DC operating-point kernel (source ramp + ConvController) instead of the transient
    int wavesPerEu = 0; // > 0: amdgpu_waves_per_eu(n, n) -- caps registers at 512 / n per lane (tuning aid)
    // the reference's arithmetic: no FMA contraction, one true division per multiplier and per solution entry
    // instead of a reciprocal, and no slow-step rule (a step that ends at the NR cap is kept and flagged as
    // upstream): on a recorded pivot sequence this kernel is bit-faithful, like the general kernels
    bool faithful = false;
};

// emits ONE __global__ kernel; returns the number of LDS doubles per lane it uses
int emitKernel(std::ostringstream& src, const csim_ir& ir, const AssemblyPlan& ap, const ScheduleSet& set,
               const VariantOptions& opt, const GeneratorOptions& gopt, CodegenStats* statsOut)
{
    const int N = ir.n_unknowns;
    const int LD = ap.LD;
    const csim_consts& K = ir.k;
    Gen g(ir, ap);
    // near-threshold guard (codegen.hpp GeneratorOptions::nearBand): the fast kernels only
    const bool guard = !opt.faithful && (opt.dcMode ? gopt.nearBandDc : gopt.nearBand) > 0.0;

    // term -> abstract value.  Exact constants: the global ONE term and the
    // inductor incidence "one" (precondition L > 0 is checked per instance).
    std::vector<AV> termAV(static_cast<std::size_t>(ap.nTerms));
    for (int t = 0; t < ap.nTerms; ++t) termAV[static_cast<std::size_t>(t)] = AV::dyn(tname(t));
    termAV[static_cast<std::size_t>(ap.termOne)] = AV::konst(1.0);
    for (int e = 0; e < ir.n_elems; ++e)
        if (ir.kind[e] == CSIM_L)
            termAV[static_cast<std::size_t>(ap.termBase[static_cast<std::size_t>(e)] + T_L_ONE)] = AV::konst(1.0);

    // terms that do not change inside the Newton loop (launch- or step-constant)
    std::vector<char> invariant(static_cast<std::size_t>(ap.nTerms), 1);
    for (int e = 0; e < ir.n_elems; ++e)
        if (ir.kind[e] == CSIM_NMOS || ir.kind[e] == CSIM_PMOS)
            for (int o = T_M_GD; o <= T_M_CST; ++o)
                invariant[static_cast<std::size_t>(ap.termBase[static_cast<std::size_t>(e)] + o)] = 0;

    // ---- which terms are per-step (sources, history currents): they live in LDS too
    std::vector<int> stepSlot(static_cast<std::size_t>(ap.nTerms), -1);
    int nStep = 0;
    for (int e = 0; e < ir.n_elems; ++e) {
        const int tb = ap.termBase[static_cast<std::size_t>(e)];
        switch (ir.kind[e]) {
            case CSIM_V: case CSIM_I: stepSlot[static_cast<std::size_t>(tb + T_SRC_VAL)] = nStep++; break;
            case CSIM_C: if (!opt.dcMode) stepSlot[static_cast<std::size_t>(tb + T_C_IH)] = nStep++; break;
            case CSIM_L: if (!opt.dcMode) stepSlot[static_cast<std::size_t>(tb + T_L_VH)] = nStep++; break;
            case CSIM_NMOS: case CSIM_PMOS:
                if (opt.dcMode) break;
                for (int o = T_M_IHGS; o <= T_M_IHDB; ++o) stepSlot[static_cast<std::size_t>(tb + o)] = nStep++;
                break;
            default: break;
        }
    }
    if (!opt.stepInLds) nStep = 0;     // per-step terms stay in registers: plain variables st<j>
    auto sname = [&](int t) {
        const std::string j = std::to_string(stepSlot[static_cast<std::size_t>(t)]);
        return opt.stepInLds ? "S(" + j + ")" : "st" + j;
    };
    for (int t = 0; t < ap.nTerms; ++t)
        if (stepSlot[static_cast<std::size_t>(t)] >= 0) termAV[static_cast<std::size_t>(t)] = AV::dyn(sname(t));

    // LDS layout: lds[slot*64 + lane]; slots 0..N-1 = x, N.. = per-step terms, then (rich
    // variant) launch constants, parameters used inside the loops and parked U rows.  Every
    // lane only ever touches its own column, so no barrier or fence is needed.
    int ldsNext = N + nStep;       // next free LDS slot (doubles per lane)
    auto qslot = [&]() { return "Q(" + std::to_string(ldsNext++) + ")"; };
    // parameters / launch terms referenced inside the time loop: registers (lean) or LDS (rich)
    std::vector<std::string> pRef(static_cast<std::size_t>(ir.n_params));
    for (int p = 0; p < ir.n_params; ++p) pRef[static_cast<std::size_t>(p)] = pname(p);
    std::vector<std::string> tRef(static_cast<std::size_t>(ap.nTerms));
    for (int t = 0; t < ap.nTerms; ++t) tRef[static_cast<std::size_t>(t)] = tname(t);
    std::ostringstream ldsInit;    // stores that fill the rich variant's LDS copies
    auto toLdsParam = [&](int p) {
        if (!opt.rich || pRef[static_cast<std::size_t>(p)][0] == 'Q') return;
        const std::string q = qslot();
        ldsInit << "    " << q << " = " << pname(p) << ";\n";
        pRef[static_cast<std::size_t>(p)] = q;
    };
    auto toLdsTerm = [&](int t) {
        if (!opt.rich) return;
        const std::string q = qslot();
        ldsInit << "    " << q << " = " << tname(t) << ";\n";
        tRef[static_cast<std::size_t>(t)] = q;
        termAV[static_cast<std::size_t>(t)] = AV::dyn(q);
    };
    for (int e = 0; e < ir.n_elems; ++e) {
        const int sl = ir.param_slot[e], tb = ap.termBase[static_cast<std::size_t>(e)];
        switch (ir.kind[e]) {
            case CSIM_R: toLdsTerm(tb + T_R_G); break;
            case CSIM_C: if (!opt.dcMode) toLdsTerm(tb + T_C_GC); break;
            case CSIM_L: if (!opt.dcMode) toLdsTerm(tb + T_L_REQ); break;
            case CSIM_V: case CSIM_I:
                for (int o = 0, cnt = (e + 1 < ir.n_elems ? ir.param_slot[e + 1] : ir.n_params) - sl; o < cnt; ++o) {
                    toLdsParam(sl + o);
                    // lean: source parameters are needed once per time step only -- re-read them
                    // from the table (L2-resident) instead of pinning 6 doubles per source in
                    // registers for the whole kernel.  The index carries an offset `vo` that is always 0
                    // but opaque to the compiler, which keeps LICM from hoisting the loads; a volatile
                    // access did the same but compiled to flat_load (both wait counters): measured
                    // 6.59e8 -> 6.88e8 at B = 4096 and 8.97e9 -> 9.86e9 at B = 65 536
                    if (!opt.rich)
                        pRef[static_cast<std::size_t>(sl + o)] = "params[" + std::to_string(sl + o) + "LL * SB + bb + vo]";
                }
                break;
            case CSIM_NMOS: case CSIM_PMOS:
                if (!opt.dcMode) { toLdsTerm(tb + T_M_GCH); toLdsTerm(tb + T_M_GCF); }
                for (int o = 0; o < 3; ++o) toLdsParam(sl + o);
                break;
            default: break;
        }
    }

    if (opt.dcMode) {
        if (opt.faithful) src << "#pragma clang fp contract(off)\n";
        src << "extern \"C\" __global__ void __launch_bounds__(64)\n"
            << opt.kernelName << "(const double* __restrict__ params, int B, double* __restrict__ xout,\n"
            << "                       int* __restrict__ iters, unsigned* __restrict__ status,\n"
            << "                       unsigned char* __restrict__ fallback, int* __restrict__ violFlag,\n"
            << "                       const unsigned char* __restrict__ only)\n{\n"
            << "    __shared__ double lds[@LDS_DOUBLES@ * 64];\n"
            << "    const int lane = threadIdx.x;\n"
            << "    const int b = blockIdx.x * 64 + threadIdx.x;\n"
            << "    const bool inb = b < B && (!only || only[b < B ? b : 0] != 0);   // only: the instances this launch replays\n"
            << "    if (!__any(inb)) return;\n"
            << "    const long long bb = inb ? b : B - 1;      // out-of-range lanes shadow the last instance, never store\n"
            << "    const long long SB = B;\n"
            << "    const bool splitFlag = B < 0;              // never true; opaque to the compiler\n";
    } else {
        if (opt.faithful) src << "#pragma clang fp contract(off)\n";
        src << "extern \"C\" __global__ void __launch_bounds__(64)"
            << (opt.wavesPerEu > 0 ? " __attribute__((amdgpu_waves_per_eu(" + std::to_string(opt.wavesPerEu) + ", " + std::to_string(opt.wavesPerEu) + ")))" : std::string()) << "\n"
            << opt.kernelName << "(const double* __restrict__ params, int B, double dt, long long stepFirst,\n"
            << "                       long long nSteps, const int* __restrict__ probeEq, int nProbe, int outStride,\n"
            << "                       double* __restrict__ wave, double* __restrict__ xio, long long* __restrict__ iters,\n"
            << "                       unsigned* __restrict__ status, int* __restrict__ stepIters,\n"
            << "                       unsigned char* __restrict__ fallback, int* __restrict__ done,\n"
            << "                       int* __restrict__ violFlag, double* __restrict__ nearX, int* __restrict__ nearStep,\n"
            << "                       int* __restrict__ nearIt, long long* __restrict__ nearItAfter)\n{\n"
            << "    __shared__ double lds[@LDS_DOUBLES@ * 64];\n"
            << "    const int lane = threadIdx.x;\n"
            << "    const int b = blockIdx.x * 64 + threadIdx.x;\n"
            << "    const bool inb = b < B;\n"
            << "    const long long bb = inb ? b : B - 1;      // out-of-range lanes shadow the last instance, never store\n"
            << "    const long long SB = B;\n"
            << "    const bool splitFlag = outStride < 0;      // never true (the engine rejects it); opaque to the compiler\n"
            << "    // hand-back launches: nothing to do for this wave unless one of its instances is unfinished\n"
            << "    if (!__any(inb && done[bb] < nSteps)) return;\n";

    }

    // ---- parameters
    for (int p = 0; p < ir.n_params; ++p) {
        if (pRef[static_cast<std::size_t>(p)][0] == '(') continue;        // re-read at its use
        src << "    const double " << pname(p) << " = params[" << p << "LL * SB + bb];\n";
    }

    // ---- launch-constant terms (device_common.hpp terms_const<true>)
    src << "    bool viol = false;      // pivot schedule (or a precondition of it) violated -> general kernel\n";
    for (int e = 0; e < ir.n_elems; ++e) {
        const int s = ir.param_slot[e], tb = ap.termBase[static_cast<std::size_t>(e)];
        switch (ir.kind[e]) {
            case CSIM_R:
                src << "    const double " << tname(tb + T_R_G) << " = (" << pname(s) << " == 0.0) ? 0.0 : 1.0 / " << pname(s) << ";\n";
                break;
            case CSIM_C:
                if (opt.dcMode) break;                      // open circuit at DC
                src << "    const double " << tname(tb + T_C_GC) << " = (" << pname(s) << " > 0.0 && dt > 0.0) ? " << pname(s) << " / dt : 0.0;\n";
                break;
            case CSIM_L:
                if (opt.dcMode) break;                      // 0 V source at DC: constant incidence only
                src << "    const double " << tname(tb + T_L_REQ) << " = " << pname(s) << " / dt;\n"
                    << "    viol = viol || !(" << pname(s) << " > 0.0);   // incidence +-1 folded as constants\n";
                break;
            case CSIM_NMOS: case CSIM_PMOS:
                if (opt.dcMode) break;                      // no junction capacitors at DC
                src << "    const double ch" << e << " = 0.5 * " << pname(s + 3) << ";\n"
                    << "    const double " << tname(tb + T_M_GCH) << " = (ch" << e << " > 0.0 && dt > 0.0) ? ch" << e << " / dt : 0.0;\n"
                    << "    const double " << tname(tb + T_M_GCF) << " = (" << pname(s + 3) << " > 0.0 && dt > 0.0) ? " << pname(s + 3) << " / dt : 0.0;\n";
                break;
            default: break;
        }
    }
    if (!opt.dcMode) src << "    const double " << tname(ap.termGmin) << " = " << lit(K.tran_gmin) << ";\n";
    else termAV[static_cast<std::size_t>(ap.termGmin)] = AV::dyn("gminv");      // ConvController's gmin, per iteration
    src << ldsInit.str();

    // ---- state
    if (opt.dcMode) {
        // dcSolveNewtonLU (reference src/dcanalysis.cpp:95-163): x = 0, sources ramped in dc_ramp_steps
        // steps, damped Newton with the ConvController's adaptive gmin in each
        src << "#pragma unroll 1\n    for (int i = 0; i < " << N << "; ++i) X(i) = 0.0;\n"
            << "    unsigned st = 0u;\n    int itTotal = 0;\n"
            << "    for (int step = 1; step <= " << K.dc_ramp_steps << "; ++step) {\n"
            << "        if (!__any(inb && !viol)) break;\n"
            << "        const double scale = (double)step / " << K.dc_ramp_steps << ";\n"
            << "        // baseGmin(scale) (dcanalysis.hpp:45-48), every product and sum rounded separately\n"
            << "        const double gb = csim_add_rn(csim_mul_rn(" << lit(K.gmin_high) << ", 1.0 - scale), csim_mul_rn(" << lit(K.gmin_low) << ", scale));\n"
            << "        double gminv = gb;\n"
            << "        double prevErr = INFINITY;\n"
            << "        const long long vo = splitFlag ? (long long)step : 0LL;\n";
    } else {
        src << "    {\n        const double* xin = xio + bb;\n#pragma unroll 1\n"
            << "        for (int i = 0; i < " << N << "; ++i, xin += SB) X(i) = *xin;\n    }\n";
        src << "    unsigned st = inb ? status[bb] : 0u;\n"
            << "    bool dead = !inb || (st & ST_TRAN_NONFINITE) != 0u;   // the reference would have thrown: stay stopped\n"
            << "    long long itTotal = 0;\n"
            << "    // steps of this launch already completed for this instance (hybrid stepping: after a schedule\n"
            << "    // violation the general kernel advances the instance a few steps and hands it back)\n"
            << "    long long sdone = (inb && !dead) ? (long long)done[bb] : nSteps;\n"
            << (opt.faithful
                ? "    // hand-over reason 2 (engine.cpp): a near-threshold convergence decision of a fast kernel is redone here with\n"
                  "    // the reference's arithmetic -- this lane runs ONE time step, then the fast kernel has the instance back\n"
                  "    const long long lend = (inb && fallback[bb] == 2 && sdone < nSteps) ? sdone + 1 : nSteps;\n"
                : (guard ? "    int nearS = 0;          // step (of this launch) of the first near-threshold convergence decision; 0 = none\n" : ""))
            << "    if (stepFirst == 0 && sdone == 0 && wave && inb) {\n"
            << "        for (int q = 0; q < nProbe; ++q) wave[((long long)q) * SB + b] = X(probeEq[q]);\n"
            << "    }\n\n"
            << "    // the step counter stays wave-uniform (scalar registers): start at the least advanced lane's next\n"
            << "    // step; a lane takes part in step s when s is ITS next step (lanes handed back by the general\n"
            << "    // kernel may be ahead of or behind their wave-mates)\n"
            << "    int smin = (int)(sdone < nSteps ? sdone + 1 : nSteps + 1);\n"
            << "    for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(smin, m); smin = o < smin ? o : smin; }\n"
            << "    smin = __builtin_amdgcn_readfirstlane(smin);\n"
            << "    for (long long s = smin; s <= nSteps; ++s) {\n"
            << "        if (!__any(!dead && !viol && sdone < " << (opt.faithful ? "lend" : "nSteps") << ")) break;\n"
            << "        const bool live = !dead && !viol && sdone + 1 == s" << (opt.faithful ? " && sdone < lend" : "") << ";\n"
            << "        const long long gstep = stepFirst + s;\n"
            << "        const double tNow = (double)(int)gstep * dt;\n"
            << "        const long long vo = splitFlag ? s : 0LL;   // always 0, but not to the compiler: keeps per-step re-reads in the loop\n"
            << "        if (live) {     // checkpoint: state at the start of this step.  A ROLLED loop: unrolled,\n"
            << "                        // hipcc hoists the N store addresses out of the time loop and pins 2N VGPRs\n"
            << "            double* ck = xio + b;\n"
            << "#pragma unroll " << opt.ckUnroll << "\n"
            << "            for (int i = 0; i < " << N << "; ++i, ck += SB) *ck = X(i);\n"
            << "        }\n";

    }

    // ---- per-step terms (device_common.hpp terms_step_tran), stored to LDS or kept in registers
    const std::string i2 = "        ";
    if (!opt.stepInLds)
        for (int t = 0; t < ap.nTerms; ++t)
            if (stepSlot[static_cast<std::size_t>(t)] >= 0) src << i2 << "double " << sname(t) << ";\n";
    for (int e = 0; e < ir.n_elems; ++e) {
        const int s = ir.param_slot[e], tb = ap.termBase[static_cast<std::size_t>(e)];
        const int32_t* q = ir.eq + 4 * e;
        switch (ir.kind[e]) {
            case CSIM_V: case CSIM_I:
                if (opt.dcMode) {
                    // SourceSpec::evalDC (reference include/sim.hpp:152-158): (dc + (SIN ? v0 : 0)) * scale
                    if (ir.wave[e] == CSIM_WAVE_SIN)
                        src << i2 << sname(tb) << " = (" << pRef[static_cast<std::size_t>(s)] << " + " << pRef[static_cast<std::size_t>(s + 1)] << ") * scale;\n";
                    else
                        src << i2 << sname(tb) << " = " << pRef[static_cast<std::size_t>(s)] << " * scale;\n";
                } else {
                    emitTranSourceValue(src, i2, ir, e, [&](int o) { return pRef[static_cast<std::size_t>(s + o)]; }, sname(tb));
                }
                break;
            case CSIM_C:
                if (opt.dcMode) break;
                src << i2 << sname(tb + T_C_IH) << " = -" << tRef[static_cast<std::size_t>(tb + T_C_GC)] << " * " << vdiff(q[0], q[1]) << ";\n";
                break;
            case CSIM_L:
                if (opt.dcMode) break;
                src << i2 << sname(tb + T_L_VH) << " = -" << tRef[static_cast<std::size_t>(tb + T_L_REQ)] << " * X(" << ir.branch_eq[e] << ");\n";
                break;
            case CSIM_NMOS: case CSIM_PMOS:
                if (opt.dcMode) break;
                src << i2 << sname(tb + T_M_IHGS) << " = -" << tRef[static_cast<std::size_t>(tb + T_M_GCH)] << " * " << vdiff(q[1], q[2]) << ";\n"
                    << i2 << sname(tb + T_M_IHGD) << " = -" << tRef[static_cast<std::size_t>(tb + T_M_GCH)] << " * " << vdiff(q[1], q[0]) << ";\n"
                    << i2 << sname(tb + T_M_IHSB) << " = -" << tRef[static_cast<std::size_t>(tb + T_M_GCF)] << " * " << vdiff(q[2], q[3]) << ";\n"
                    << i2 << sname(tb + T_M_IHDB) << " = -" << tRef[static_cast<std::size_t>(tb + T_M_GCF)] << " * " << vdiff(q[0], q[3]) << ";\n";
                break;
            default: break;
        }
    }

    if (opt.dcMode)
        src << i2 << "bool active = inb && !viol;\n"
            << i2 << "for (int iter = 0; iter < " << K.dc_max_iters << "; ++iter) {\n"
            << i2 << "    if (!__any(active)) break;\n";
    else
        src << i2 << "bool active = live;\n"
            << i2 << "int it = 0;\n"
            << (guard ? i2 + "double nearMin = 1.0;   // smallest |ss - tol^2| of this step's passes\n" : std::string())
            << i2 << "for (int iter = 0; iter < " << K.tran_max_iters << "; ++iter) {\n"
            << i2 << "    if (!__any(active)) break;\n";
    // ---- per-iteration terms: MOS channel (device_common.hpp mos_eval)
    std::vector<char> xLoaded(static_cast<std::size_t>(N), 0);
    for (int e = 0; e < ir.n_elems; ++e) {
        if (ir.kind[e] != CSIM_NMOS && ir.kind[e] != CSIM_PMOS) continue;
        const int s = ir.param_slot[e], tb = ap.termBase[static_cast<std::size_t>(e)];
        const int32_t* q = ir.eq + 4 * e;
        const bool isP = ir.kind[e] == CSIM_PMOS;
        const std::string m = "m" + std::to_string(e) + "_";
        for (int tq = 0; tq < 3; ++tq) {
            if (q[tq] >= 0 && !xLoaded[static_cast<std::size_t>(q[tq])]) {
                g.out << g.ind << "const double " << xloc(q[tq]) << " = " << xname(q[tq]) << ";\n";
                xLoaded[static_cast<std::size_t>(q[tq])] = 1;
            }
        }
        const std::string Vd = xloc(q[0]), Vg = xloc(q[1]), Vs = xloc(q[2]);
        const std::string pVth = "m" + std::to_string(e) + "_vth", pK = "m" + std::to_string(e) + "_k",
                          pLam = "m" + std::to_string(e) + "_lam";
        g.out << g.ind << "const double " << pVth << " = " << pRef[static_cast<std::size_t>(s)] << ", " << pK << " = "
              << pRef[static_cast<std::size_t>(s + 1)] << ", " << pLam << " = " << pRef[static_cast<std::size_t>(s + 2)] << ";\n";
        g.out << g.ind << "// MOS element " << e << (isP ? " (PMOS)" : " (NMOS)") << "\n";
        g.out << g.ind << "const double " << m << "vgs = " << (isP ? "-" : "") << "(" << Vg << " - " << Vs << ");\n"
              << g.ind << "const double " << m << "vds = " << (isP ? "-" : "") << "(" << Vd << " - " << Vs << ");\n"
              << g.ind << "const double " << m << "vov = " << m << "vgs - " << pVth << ";\n"
              << g.ind << "const bool " << m << "on = (" << m << "vgs > " << pVth << ") && (" << m << "vds >= 0.0);\n"
              << g.ind << "const bool " << m << "tri = " << m << "vds < " << m << "vov;\n"
              << g.ind << "const double " << m << "id0 = " << m << "on ? (" << m << "tri ? " << pK << " * (" << m << "vov * " << m
              << "vds - 0.5 * " << m << "vds * " << m << "vds) : 0.5 * " << pK << " * " << m << "vov * " << m << "vov) : 0.0;\n"
              << g.ind << "const double " << m << "gds0 = " << m << "on ? (" << m << "tri ? " << pK << " * (" << m << "vov - " << m
              << "vds) : 0.0) : " << lit(K.mos_off_gds) << ";\n"
              << g.ind << "const double " << m << "gm0 = " << m << "on ? (" << m << "tri ? " << pK << " * " << m << "vds : "
              << pK << " * " << m << "vov) : 0.0;\n"
              << g.ind << "const double " << m << "fac = fmax(1.0 + " << pLam << " * " << m << "vds, 0.0);\n"
              << g.ind << "const double " << tname(tb + T_M_GD) << " = " << m << "gds0 * " << m << "fac + " << m << "id0 * " << pLam << ";\n"
              << g.ind << "const double " << tname(tb + T_M_GG) << " = " << m << "gm0 * " << m << "fac;\n"
              << g.ind << "const double " << tname(tb + T_M_GS) << " = -(" << tname(tb + T_M_GD) << " + " << tname(tb + T_M_GG) << ");\n"
              << g.ind << "const double " << tname(tb + T_M_CST) << " = " << (isP ? "-" : "") << "(" << m << "id0 * " << m << "fac) - "
              << tname(tb + T_M_GD) << " * " << Vd << " - " << tname(tb + T_M_GG) << " * " << Vg << " - " << tname(tb + T_M_GS) << " * " << Vs << ";\n";
    }

    // ---- one solve body per recorded pivot schedule: assembly (lazy) + elimination with that
    // schedule's pivots + back substitution.  Returns the abstract solution; its pivot checks
    // accumulate into the bool named pvName.
    const int ldsBase = ldsNext;
    int ldsMax = ldsNext;
    auto emitSolve = [&](const PivotSchedule& sc, const std::string& pvName) -> std::vector<AV> {
        ldsNext = ldsBase;                 // parked slots are reused by every alternative
        // ---- assemble [G | I] symbolically (gather lists of plan.cpp, reference order).
        // Entries are only RECORDED here; the code of an entry is emitted when the
        // elimination first reads it (shortens live ranges: ~137 entries would
        // otherwise all be live at once).
        std::vector<std::vector<AV>> M(static_cast<std::size_t>(N), std::vector<AV>(static_cast<std::size_t>(N + 1)));
        g.out << g.ind << "// assembly (lazy) + elimination\n";
        g.pending.assign(static_cast<std::size_t>(N), std::vector<std::vector<AV>>(static_cast<std::size_t>(N + 1)));
        const GatherPlan& gp = opt.dcMode ? ap.dc : ap.tran;
        for (int n = 0; n < gp.nnzG(); ++n) {
            std::vector<AV> terms;
            for (int c = gp.gPtr[static_cast<std::size_t>(n)]; c < gp.gPtr[static_cast<std::size_t>(n + 1)]; ++c) {
                const int con = gp.gCon[static_cast<std::size_t>(c)];
                AV t = termAV[static_cast<std::size_t>(con >> 1)];
                terms.push_back((con & 1) ? g.negate(t) : t);
            }
            const int pos = gp.gPos[static_cast<std::size_t>(n)];
            g.pending[static_cast<std::size_t>(pos / LD)][static_cast<std::size_t>(pos % LD)] = terms;
            // structural marker so that zero tests see the entry before it is materialised
            M[static_cast<std::size_t>(pos / LD)][static_cast<std::size_t>(pos % LD)] = AV::dyn("?");
        }
        for (int n = 0; n < gp.nnzI(); ++n) {
            std::vector<AV> terms;
            for (int c = gp.iPtr[static_cast<std::size_t>(n)]; c < gp.iPtr[static_cast<std::size_t>(n + 1)]; ++c) {
                const int con = gp.iCon[static_cast<std::size_t>(c)];
                AV t = termAV[static_cast<std::size_t>(con >> 1)];
                terms.push_back((con & 1) ? g.negate(t) : t);
            }
            const int r = gp.iRow[static_cast<std::size_t>(n)];
            g.pending[static_cast<std::size_t>(r)][static_cast<std::size_t>(N)] = terms;
            M[static_cast<std::size_t>(r)][static_cast<std::size_t>(N)] = AV::dyn("?");
        }
        // resolve every all-constant entry now (they cost no code and decide the zero pattern)
        for (int r = 0; r < N; ++r)
            for (int c = 0; c <= N; ++c) {
                auto& pend = g.pending[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)];
                if (pend.empty()) continue;
                bool allConst = true;
                for (const AV& t : pend) allConst = allConst && t.kind != AV::DYN;
                if (allConst) { M[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)] = g.orderedSum(pend); pend.clear(); }
            }
        // materialise on first use
        auto at = [&](int r, int c) -> AV& {
            auto& pend = g.pending[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)];
            AV& slot = M[static_cast<std::size_t>(r)][static_cast<std::size_t>(c)];
            if (!pend.empty()) { slot = g.orderedSum(pend); pend.clear(); }
            return slot;
        };

        // ---- elimination with the scheduled pivots (solver.hpp:46-77), RHS carried along
        std::vector<AV> rinv(static_cast<std::size_t>(N));      // 1 / U(k,k)
        int parked = 0;
        for (int k = 0; k < N; ++k) {
            const int p = sc.pivotPos[static_cast<std::size_t>(k)];
            const AV ap_ = at(p, k);
            g.out << g.ind << "// column " << k << ": pivot row position " << p << "\n";
            // A scheduling barrier every few columns.  hipcc schedules each basic block for ILP and inflates
            // the live set of this 3000-instruction body; with barriers the allocator ends at 86 spilled
            // registers, without at ~200 (measured 6.5e8 vs 5.1e8 NR-iter*inst/s at B = 4096).  Spacing swept
            // 1/2/3/4/5/6/8/10 with the branch-free checks: 8.41, 8.46, 8.54, 8.53, 8.47, 8.49, 8.48, 8.43e8
            // at B = 4096; a real block boundary (scalar branch on an opaque flag) measured equal within noise.
            if (gopt.barrierEvery > 0 && (k % gopt.barrierEvery) == 0) g.out << g.ind << "__builtin_amdgcn_sched_barrier(0);\n";
            // the reference picks the FIRST row attaining the column maximum (solver.hpp:48-56)
            // and fails below 1e-15 (:58-61)
            if (ap_.isZero()) {
                g.out << g.ind << pvName << " = true;   // scheduled pivot is a structural zero\n";
            } else {
                const std::string absP = ap_.kind == AV::CONST ? lit(std::fabs(ap_.c)) : "fabs(" + ap_.v + ")";
                std::vector<std::string> conds;        // all must hold
                bool contradiction = false;
                if (ap_.kind == AV::DYN) { conds.push_back("(" + absP + " >= " + lit(K.lu_eps) + ")"); ++g.st.nCmp; }
                else if (std::fabs(ap_.c) < K.lu_eps) contradiction = true;
                for (int i = k; i < N; ++i) {
                    if (i == p) continue;
                    const AV& ai = at(i, k);
                    if (ai.isZero()) continue;
                    const bool before = i < p;
                    if (ai.kind == AV::CONST && ap_.kind == AV::CONST) {
                        const bool ok = before ? (std::fabs(ap_.c) > std::fabs(ai.c)) : (std::fabs(ap_.c) >= std::fabs(ai.c));
                        if (!ok) contradiction = true;   // schedule contradicts constant entries
                        continue;
                    }
                    const std::string absI = ai.kind == AV::CONST ? lit(std::fabs(ai.c)) : "fabs(" + ai.v + ")";
                    conds.push_back("(" + absP + (before ? " > " : " >= ") + absI + ")");
                    ++g.st.nCmp;
                }
                if (contradiction) g.out << g.ind << pvName << " = true;\n";
                else if (ap_.kind == AV::DYN) {
                    // one running maximum per side (rows before / after the scheduled one), then at most two
                    // tests per column, accumulated WITHOUT branches ("|=": 8.2e8 vs 6.9e8 at B = 4096 for
                    // short-circuit "||" chains) with 1e-15 folded into the later-rows maximum (+1.9 %).
                    // fmax ignores a NaN operand exactly like the reference's "> maxVal" scan.
                    std::string mb, ma;
                    for (int i = k; i < N; ++i) {
                        if (i == p) continue;
                        const AV& ai = at(i, k);
                        if (ai.isZero()) continue;
                        const std::string absI = ai.kind == AV::CONST ? lit(std::fabs(ai.c)) : "fabs(" + ai.v + ")";
                        std::string& m = (i < p) ? mb : ma;
                        m = m.empty() ? absI : "fmax(" + m + ", " + absI + ")";
                    }
                    std::string e = ma.empty() ? "(" + absP + " >= " + lit(K.lu_eps) + ")"
                                               : "(" + absP + " >= fmax(" + ma + ", " + lit(K.lu_eps) + "))";
                    if (!mb.empty()) e += " & (" + absP + " > " + mb + ")";
                    g.out << g.ind << pvName << " |= !(" << e << ");\n";
                } else {
                    // constant pivot (a +-1 incidence entry) against run-time candidates: short-circuit form.
                    // Its block boundaries are load-bearing for hipcc's register allocation of this body
                    // (branch-free "|=" here: 46 -> 243 spilled registers, 8.5e8 -> 4.8e8 at B = 4096).
                    for (const std::string& c : conds) g.out << g.ind << pvName << " = " << pvName << " || !" << c << ";\n";
                }
            }
            if (p != k) {
                std::swap(M[static_cast<std::size_t>(p)], M[static_cast<std::size_t>(k)]);
                std::swap(g.pending[static_cast<std::size_t>(p)], g.pending[static_cast<std::size_t>(k)]);
            }
            const AV piv = at(k, k);
            AV r;                       // what the row is scaled with: 1 / pivot, or (faithful) the pivot itself as divisor
            if (opt.faithful) r = piv;
            else if (piv.kind == AV::CONST) r = AV::konst(1.0 / piv.c);
            else if (piv.kind == AV::DYN) { r = g.emit("rcp_nr(" + g.ref(piv) + ")"); ++g.st.nRecip; }
            rinv[static_cast<std::size_t>(k)] = r;
            for (int i = k + 1; i < N; ++i) {
                const AV aik = at(i, k);
                if (aik.isZero()) continue;
                ++g.st.nLower;
                const AV f = opt.faithful ? g.div(aik, r) : g.mul(aik, r);     // multiplier (solver.hpp:71)
                for (int j = k + 1; j <= N; ++j) {
                    if (M[static_cast<std::size_t>(k)][static_cast<std::size_t>(j)].isZero()) continue;
                    const AV u = at(k, j);
                    if (u.isZero()) continue;
                    const AV a = at(i, j);
                    M[static_cast<std::size_t>(i)][static_cast<std::size_t>(j)] = g.fnma(a, f, u);   // :74
                }
                M[static_cast<std::size_t>(i)][static_cast<std::size_t>(k)] = AV::zero();
            }
            // row k is final: it is next read in the back substitution.  The rich variant parks
            // its run-time entries (and the pivot reciprocal) in LDS instead of leaving it to the
            // register allocator to spill them to scratch.
            if (opt.parkBudget != 0) {
                for (int j = k + 1; j <= N; ++j) {
                    if (M[static_cast<std::size_t>(k)][static_cast<std::size_t>(j)].isZero()) continue;
                    const AV v = at(k, j);
                    if (v.kind != AV::DYN) continue;
                    if (opt.parkBudget > 0 && parked >= opt.parkBudget) break;
                    ++parked;
                    const std::string q = qslot();
                    g.out << g.ind << q << " = " << g.ref(v) << ";\n";
                    M[static_cast<std::size_t>(k)][static_cast<std::size_t>(j)] = AV::dyn(q);
                }
                if (r.kind == AV::DYN && (opt.parkBudget < 0 || parked < opt.parkBudget)) {
                    ++parked;
                    const std::string q = qslot();
                    g.out << g.ind << q << " = " << g.ref(r) << ";\n";
                    rinv[static_cast<std::size_t>(k)] = AV::dyn(q);
                }
            }
        }

        // ---- back substitution (solver.hpp:116-128): row i descending, j ascending
        g.out << g.ind << "// back substitution\n";
        std::vector<AV> xr(static_cast<std::size_t>(N));
        for (int i = N - 1; i >= 0; --i) {
            AV sum = at(i, N);
            for (int j = i + 1; j < N; ++j) {
                if (M[static_cast<std::size_t>(i)][static_cast<std::size_t>(j)].isZero()) continue;
                const AV u = at(i, j);
                if (u.isZero()) continue;
                if (u.kind == AV::DYN) ++g.st.nDynU;
                sum = g.fnma(sum, u, xr[static_cast<std::size_t>(j)]);
            }
            xr[static_cast<std::size_t>(i)] = opt.faithful ? g.div(sum, rinv[static_cast<std::size_t>(i)])      // :126
                                                           : g.mul(sum, rinv[static_cast<std::size_t>(i)]);
        }


        if (ldsNext > ldsMax) ldsMax = ldsNext;
        return xr;
    };

    // ---- the alternatives are tried in order; lanes whose checks failed take the next one
    CodegenStats firstAlt;
    g.out << g.ind << "bool pv = false;     // every schedule tried so far failed its pivot checks\n";
    for (int i = 0; i < N; ++i) g.out << g.ind << "double xr" << i << ";\n";
    const std::vector<PivotSchedule>& alternatives = opt.dcMode ? set.dcAlts : set.alts;
    for (std::size_t alt = 0; alt < alternatives.size(); ++alt) {
        const std::string pvName = "pvA" + std::to_string(alt);
        if (alt == 0) g.out << g.ind << "{\n";
        else g.out << g.ind << "if (__any(active && pv)) {   // alternative schedule " << alt << "\n";
        g.out << g.ind << "bool " << pvName << " = false;\n";
        const std::vector<AV> sol = emitSolve(alternatives[alt], pvName);
        if (alt == 0) firstAlt = g.st;             // operation counts of ONE solve on the most frequent schedule
        for (int i = 0; i < N; ++i) {
            if (alt == 0) g.out << g.ind << "xr" << i << " = " << g.ref(sol[static_cast<std::size_t>(i)]) << ";\n";
            else g.out << g.ind << "xr" << i << " = pv ? " << g.ref(sol[static_cast<std::size_t>(i)]) << " : xr" << i << ";\n";
        }
        if (alt == 0) g.out << g.ind << "pv = " << pvName << ";\n";
        else g.out << g.ind << "pv = pv && " << pvName << ";\n";
        g.out << g.ind << "}\n";
    }
    ldsNext = ldsMax;

    if (opt.dcMode) {
        // ---- ConvController::update (reference src/dcanalysis.cpp:264-307) and the loop tail (:135-158)
        std::ostringstream& o = g.out;
        const double alpha = std::min(std::max(K.dc_alpha, K.dc_alpha_min), K.dc_alpha_max);     // :274
        o << g.ind << "double ss = 0.0;\n";
        for (int i = 0; i < N; ++i) {
            o << g.ind << "const double xo" << i << " = X(" << i << ");\n"
              << g.ind << "const double xn" << i << " = xo" << i << " + " << lit(alpha) << " * (xr" << i << " - xo" << i << ");\n"
              << g.ind << "{ const double d = xn" << i << " - xo" << i << "; ss += d * d; }\n";
        }
        // a non-finite solve (the reference bumps gmin and retries, :135-138) or a failed pivot check
        // sends the instance to the general kernel, which replays its operating point exactly
        o << g.ind << "const double err = sqrt(ss);\n"
          << g.ind << "if (active) {\n"
          << g.ind << "    if (pv || !(ss < 1.0e300)) { viol = true; active = false; }\n"
          << g.ind << "    else {\n"
          << g.ind << "        ++itTotal;\n";
        if (guard)
            // a branch of the controller decided within the rounding noise of this kernel's contracted arithmetic
            // (:150, :285-296): the bit-faithful DC kernel replays the instance
            o << g.ind << "        {\n"
              << g.ind << "            const bool first = (iter == 0 || !isfinite(prevErr));\n"
              << g.ind << "            const double band = " << lit(gopt.nearBandDc) << " * err;\n"
              << g.ind << "            if (fabs(err - " << lit(K.dc_tol) << ") <= " << lit(gopt.nearBandDc * K.dc_tol) << " ||\n"
              << g.ind << "                (!first && (fabs(err - csim_mul_rn(prevErr, " << lit(K.slow_ratio) << ")) <= band ||\n"
              << g.ind << "                            fabs(err - csim_mul_rn(prevErr, " << lit(K.fast_ratio) << ")) <= band))) { viol = true; active = false; }\n"
              << g.ind << "        }\n";
        for (int i = 0; i < N; ++i) o << g.ind << "        X(" << i << ") = xn" << i << ";\n";
        o << g.ind << "        double gnext;\n"
          << g.ind << "        if (iter == 0 || !isfinite(prevErr)) gnext = gb;                                   // :280-282\n"
          << g.ind << "        else if (err > csim_mul_rn(prevErr, " << lit(K.slow_ratio) << ")) gnext = fmin(csim_mul_rn(gminv, 2.0), " << lit(K.gmin_abs_max) << ");   // :285-288\n"
          << g.ind << "        else if (err < csim_mul_rn(prevErr, " << lit(K.fast_ratio) << ")) gnext = csim_add_rn(csim_mul_rn(0.5, gminv), csim_mul_rn(0.5, gb));   // :289-293\n"
          << g.ind << "        else gnext = csim_add_rn(csim_mul_rn(0.7, gminv), csim_mul_rn(0.3, gb));               // :296\n"
          << g.ind << "        gminv = gnext;\n"
          << g.ind << "        prevErr = err;\n"
          << g.ind << "        if (err < " << lit(K.dc_tol) << ") active = false;                                 // :150\n"
          // :153-158.  A ramp step that ends at the cap is common and harmless while the ramp goes on (the
          // damped iteration is a slow contraction: dbmixer ends 9 of its 10 ramp steps there).  When the
          // FINAL ramp step ends at the cap the returned operating point is wherever the trajectory
          // stopped, not a fixed point, and only bit-faithful arithmetic reproduces the reference's: the
          // instance is replayed by the general kernel.
          // (the faithful kernel keeps such a step and flags it, as upstream)
          << g.ind << "        else if (iter == " << (K.dc_max_iters - 1) << ") {\n"
          << (opt.faithful ? std::string()
                           : g.ind + "            if (step == " + std::to_string(K.dc_ramp_steps) + ") { viol = true; active = false; }\n" + g.ind + "            else\n")
          << g.ind << "            st |= ST_DC_NONCONV;\n"
          << g.ind << "        }\n"
          << g.ind << "    }\n"
          << g.ind << "}\n";
        src << g.out.str();
        src << i2 << "}\n"      // NR loop
            << "    }\n\n"     // ramp loop
            << "    if (inb) {\n"
            << "        if (viol) { fallback[b] = 1; *violFlag = 1; }\n"
            << "        else {\n"
            << "            double* xo = xout + b;\n#pragma unroll 1\n"
            << "            for (int i = 0; i < " << N << "; ++i, xo += SB) *xo = X(i);\n"
            << "            iters[b] = itTotal;\n"
            << "            status[b] = st;\n"
            << "        }\n"
            << "    }\n"
            << "}\n" << (opt.faithful ? "#pragma clang fp contract(fast)\n" : "") << "\n";
    } else {
        // ---- damped update, norm in index order, convergence (tanalisis.cpp:360-376)
        std::ostringstream& o = g.out;
        o << g.ind << "double ss = 0.0;\n";
        for (int i = 0; i < N; ++i) {
            o << g.ind << "const double xo" << i << " = X(" << i << ");\n"
              << g.ind << "const double xn" << i << " = xo" << i << " + " << lit(K.tran_alpha) << " * (xr" << i << " - xo" << i << ");\n"
              << g.ind << "{ const double d = xn" << i << " - xo" << i << "; ss += d * d; }\n";
        }
        // a non-finite solve (tanalisis.cpp:360-362) makes ss non-finite; so does an overflow of
        // finite but absurd values -- both are left to the general kernel to classify exactly
        // (guard on: the pass decides on the squared norm, see codegen_group.cpp -- no square root in the loop)
        o << (guard ? std::string() : g.ind + "const double err = sqrt(ss);\n")
          << g.ind << "if (active) {\n"
          << g.ind << "    if (pv || !(ss < 1.0e300)) { viol = true; active = false; }\n"
          << g.ind << "    else {\n"
          << g.ind << "        ++it;\n";
        for (int i = 0; i < N; ++i) o << g.ind << "        X(" << i << ") = xn" << i << ";\n";
        // Slow steps leave this kernel (plan.hpp slowStepIters).  The reference keeps a step that reaches
        // the NR cap with a WARNING (tanalisis.cpp:372-376); steps that need anywhere near that many passes
        // are chaotic -- the 1e-16 of FMA contraction in this kernel moves them by +-1..4 passes and 4e-7
        // in the state -- so such a step is treated like a failed pivot check: the lane stops at the step's
        // checkpoint and the bit-faithful general kernel redoes it (and sets CSIM_ST_TRAN_NONCONV if due).
        const int slowIters = slowStepIters(K.tran_tol, K.tran_alpha, K.tran_max_iters);
        o << g.ind << "        if (" << (guard ? "ss < " + lit(K.tran_tol * K.tran_tol) : "err < " + lit(K.tran_tol)) << ") active = false;\n";
        if (opt.faithful) o << g.ind << "        else if (iter == " << (K.tran_max_iters - 1) << ") st |= ST_TRAN_NONCONV;   // tanalisis.cpp:372-376\n";
        else o << g.ind << "        else if (iter >= " << (slowIters - 1) << ") { viol = true; active = false; }\n";
        if (guard)
            // near-threshold guard: how close did `err < tol` (tanalisis.cpp:369) come to a tie in this step?  Decided
            // once per step, below.
            o << g.ind << "        nearMin = fmin(nearMin, fabs(ss - " << lit(K.tran_tol * K.tran_tol) << "));\n";
        o
          << g.ind << "    }\n"
          << g.ind << "}\n";

        src << g.out.str();
        src << i2 << "}\n"      // NR loop
            << (guard
                // `err < tol` decided within the rounding noise of this kernel's arithmetic: go on speculatively; the step's
                // checkpoint is kept and the engine has the faithful kernel verify the pass count.  One checkpoint per
                // launch: a second such step stops the lane at the start of that step.
                ? i2 + "const bool nearEvent = live && !viol && nearMin <= " + lit(2.0 * gopt.nearBand * K.tran_tol * K.tran_tol) + ";\n"
                  + i2 + "viol = viol || (nearEvent && nearS != 0);\n"
                : std::string())
            << i2 << "if (live && !viol) {\n"
            << (guard
                ? i2 + "    if (nearEvent) {     // xio still holds the state at the start of this step: keep it for the verification\n"
                  + i2 + "        nearS = (int)s;\n"
                  + i2 + "        const double* ck = xio + b;\n" + i2 + "        double* nk = nearX + b;\n#pragma unroll 1\n"
                  + i2 + "        for (int i = 0; i < " + std::to_string(N) + "; ++i, ck += SB, nk += SB) *nk = *ck;\n"
                  + i2 + "        nearStep[b] = (int)s; nearIt[b] = it; nearItAfter[b] = itTotal;\n"
                  + i2 + "    }\n"
                : std::string())
            << i2 << "    itTotal += it;\n"
            << i2 << "    if (stepIters) stepIters[(s - 1) * SB + b] = it;\n"
            << i2 << "    if (wave && !dead && (gstep % outStride) == 0) {\n"
            << i2 << "        const long long row = gstep / outStride;\n"
            << i2 << "        for (int q = 0; q < nProbe; ++q) wave[(row * nProbe + q) * SB + b] = X(probeEq[q]);\n"
            << i2 << "    }\n"
            << (opt.faithful ? i2 + "    st |= ST_SCHED_FAITHFUL;\n" : std::string())
            << i2 << "    sdone = dead ? nSteps : s;\n"
            << i2 << "}\n"
            << "    }\n\n"
            << "    if (inb) {\n"
            << "        if (viol) fallback[b] = 1;   // xio holds the checkpoint of the step that failed\n"
            << "        else {\n";
        src << "            double* xo = xio + b;\n#pragma unroll 1\n"
            << "            for (int i = 0; i < " << N << "; ++i, xo += SB) *xo = X(i);\n"
            << "        }\n"
            << "        iters[b] += itTotal;\n"
            << "        status[b] |= st;\n"
            << "        done[b] = (int)sdone;\n"
            << "        if (sdone < nSteps) violFlag[0] = 1;      // unfinished: the engine goes on with this instance\n"
            << (opt.faithful ? "        if (!viol && fallback[b] == 2) fallback[b] = 0;   // the one step asked for is done\n" : "")
            << (guard ? "        if (nearS != 0 && sdone >= nearS) { nearItAfter[b] = itTotal - nearItAfter[b]; violFlag[1] = 1; }   // to be verified\n" : "")
            << "    }\n"
            << "}\n" << (opt.faithful ? "#pragma clang fp contract(fast)\n" : "") << "\n";

    }

    if (statsOut) *statsOut = firstAlt;
    return ldsNext;
}

} // namespace

std::string generateTranKernelSource(const csim_ir& ir, const AssemblyPlan& ap, const ScheduleSet& set,
                                     const std::string& label, CodegenStats* statsOut, const GeneratorOptions& gopt)
{
    const int N = ir.n_unknowns;
    const uint64_t hash = scheduleHash(ir, set, gopt);
    if (set.alts.empty()) return std::string();
    std::ostringstream src;
    src << "// GENERATED by circuitsimulator_amd/csrc/engine/codegen.cpp -- do not edit.\n"
        << "// circuit: " << label << "   N=" << N << "  elements=" << ir.n_elems << "  P=" << ir.n_params << "\n"
        << "// pivot schedules (column:row position), tried in this order: " << set.str() << "\n"
        << "// One lane = one circuit instance; see codegen.hpp for what is and is not\n"
        << "// identical to the reference arithmetic.  Two variants of the same arithmetic:\n"
        << "//   csim_tran_sched_kernel       x and as many finished U-row values as fit 40 KB of LDS per\n"
        << "//                                wave (4 waves per CU, one per SIMD)\n"
        << "//   csim_tran_sched_kernel_rich  every finished U-row value parked in LDS (tuning aid: measured\n"
        << "//                                slower -- ds traffic costs more than the spills it removes)\n"
        << "#include <hip/hip_runtime.h>\n#include <stdint.h>\n\n"
        << "#define ST_TRAN_NONFINITE 0x0001u\n#define ST_TRAN_NONCONV 0x0002u\n#define ST_DC_NONCONV 0x0008u\n#define ST_SCHED_FAITHFUL 0x0100u\n\n"
        << "#define Q(k) lds[(k) * 64 + lane]\n#define X(i) Q(i)\n#define S(j) Q(" << N << " + (j))\n\n"
        << "// a product / a sum that is rounded by itself wherever it is used (the DC controller's gmin arithmetic,\n"
        << "// src/dcanalysis.cpp:45-48,285-296).  NOT hip's __dmul_rn / __dadd_rn: those are inline functions of a header\n"
        << "// compiled with contraction allowed, and inlined here their operations fuse -- base gmin of ramp step 2 came out\n"
        << "// one ulp off, visible on a node whose diagonal is gmin alone (tools/fuzz_generated.py seed 10266)\n"
        << "#pragma clang fp contract(off)\n"
        << "__device__ __forceinline__ double csim_mul_rn(double a, double b) { return a * b; }\n"
        << "__device__ __forceinline__ double csim_add_rn(double a, double b) { return a + b; }\n"
        << "#pragma clang fp contract(fast)\n"
        << "// refined reciprocal for the pivots: v_rcp_f64 + one cubic step (three dependent FMAs)\n"
        << "__device__ __forceinline__ double clamp01_cg(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }\n"
        << "__device__ __forceinline__ double rcp_nr(double a)\n{\n"
        << "    // v_rcp_f64 (2^-24.4 relative) + one cubic step: the correctly rounded reciprocal, as two quadratic\n"
        << "    // steps give it, in three dependent FMAs instead of four (tools/dev/ubench/rcp_acc.hip)\n"
        << "    const double r = __builtin_amdgcn_rcp(a);\n"
        << "    const double e = fma(-a, r, 1.0);\n"
        << "    return fma(fma(e, e, e), r, r);\n}\n\n";

    // linear circuits: factor once per launch, substitute once per step (codegen_linear.cpp).  It replaces
    // the per-iteration kernels below for such circuits (they would re-factor in every Newton pass, and for
    // N = 257 cost minutes of compile time)
    // Linear circuits first try the sixteen-lanes-per-instance form (tape, iterate and x_raw in registers); circuits
    // whose tape does not fit the register file get the lane-per-instance form (iterate in LDS, tape streamed).
    int linWork = 0, linLanes = 0;
    const std::string lin16Src = emitLinearGroupKernel(ir, ap, set.alts[0], &linWork, gopt);
    const bool haveLinear16 = !lin16Src.empty();
    const std::string linSrc = haveLinear16 ? std::string() : emitLinearKernel(ir, ap, set.alts[0], &linWork, &linLanes);
    const bool haveLinear = haveLinear16 || !linSrc.empty();
    // the sixteen-lanes-per-instance transient kernel of Newton circuits (group_plan.hpp): one solve body per schedule
    // over the first one's row placement; a sequence none of them covers is a violation and goes through the ladder
    GroupPlan groupPlan;
    const std::string groupSrc = haveLinear ? std::string() : emitGroupKernel(ir, ap, set.alts, gopt, &groupPlan);
    const bool haveGroup = !groupSrc.empty();
    // ... and its four-lanes-per-instance form for the batches between the two (generator option group4)
    GroupPlan quadPlan;
    const std::string quadSrc = (haveGroup && gopt.group4) ? emitGroupKernel(ir, ap, set.alts, gopt, &quadPlan, 4) : std::string();
    const bool haveQuad = !quadSrc.empty();
    if (haveLinear16 || haveGroup) src << groupPreludeSource(ir);
    src << lin16Src << linSrc;
    // DC operating point of a linear circuit on its recorded DC pivot sequence (one direct solve)
    int linDcWork = 0;
    const std::string linDcSrc = (haveLinear && !set.dcAlts.empty()) ? emitLinearDcKernel(ir, ap, set.dcAlts[0], &linDcWork) : std::string();
    const bool haveLinearDc = !linDcSrc.empty();
    src << linDcSrc;
    if (linDcWork > linWork) linWork = linDcWork;               // one work area serves both

    const int leanBudget = 80 - N;
    // a variant is emitted only if its LDS image fits one CU (163 840 B)
    auto emitVariant = [&](const VariantOptions& opt, CodegenStats* st) {
        std::ostringstream k;
        const int ldsDoubles = emitKernel(k, ir, ap, set, opt, gopt, st);
        if (ldsDoubles * 512 > 160 * 1024) return -1;
        std::string text = k.str();
        const std::string token = "@LDS_DOUBLES@";
        const std::size_t at = text.find(token);
        text.replace(at, token.size(), std::to_string(ldsDoubles));
        src << text;
        return ldsDoubles;
    };
    // lean: 40 KB of LDS per wave (4 waves per CU): x, then finished U rows up to the budget;
    // for N > 80 the iterate alone exceeds that and the kernel runs fewer waves per CU
    CodegenStats leanStats;
    const int ldsLean = haveLinear ? 0 : emitVariant({"csim_tran_sched_kernel", false, false, leanBudget > 0 ? leanBudget : 0}, &leanStats);
    if (statsOut) *statsOut = leanStats;
    if (ldsLean < 0) return std::string();          // the iterate does not fit LDS: no scheduled kernel
    // rich: same residency, every finished U row parked (more LDS per wave: fewer waves per CU)
    const int ldsRich = haveLinear ? -1 : emitVariant({"csim_tran_sched_kernel_rich", false, false, -1}, nullptr);
    const bool haveRich = ldsRich >= 0;
    // faithful: the reference's arithmetic on the recorded pivot sequences (what slow steps are redone with)
    int ldsFaith = -1;
    if (!haveLinear) {
        VariantOptions fo{"csim_tran_faithful_kernel", false, false, leanBudget > 0 ? leanBudget : 0};
        fo.faithful = true;
        ldsFaith = emitVariant(fo, nullptr);
    }
    const bool haveFaithful = ldsFaith >= 0;
    const std::vector<int>& sweep = gopt.sweep;      // tuning aid (csim_codegen --sweep): extra kernels
    {
        for (std::size_t k = 0; k < sweep.size(); ++k) {
            static std::vector<std::string> names;
            names.push_back("csim_tran_sched_kernel_sweep" + std::to_string(k));
            // value >= 1000: checkpoint-unroll sweep (value - 1000) at the lean park budget
            // value >= 2000: two waves per SIMD (<= 256 registers per lane) at park budget (value - 2000)
            // 4000 + b: launch constants AND per-step terms in LDS, park budget b;  3000 + b: launch constants in LDS
            if (sweep[k] >= 4000) emitVariant({names.back().c_str(), true, true, sweep[k] - 4000}, nullptr);
            else if (sweep[k] >= 3000) emitVariant({names.back().c_str(), true, false, sweep[k] - 3000}, nullptr);
            else if (sweep[k] >= 2000) { VariantOptions o{names.back().c_str(), false, false, sweep[k] - 2000}; o.wavesPerEu = 2; emitVariant(o, nullptr); }
            else if (sweep[k] >= 1000) emitVariant({names.back().c_str(), false, false, leanBudget > 0 ? leanBudget : 0, sweep[k] - 1000}, nullptr);
            else emitVariant({names.back().c_str(), false, false, sweep[k]}, nullptr);
        }
    }

    // DC operating point with its own recorded schedules (Newton circuits only: a linear circuit's
    // DC is one solve, left to the general kernel)
    // Two kernels: csim_dc_faithful_kernel performs the reference's operations (no contraction, true divisions),
    // bit for bit the general kernel's operating points on the recorded sequences -- the engine's default;
    // csim_dc_sched_kernel is the fast one (contraction, reciprocal pivots, near-threshold guard).
    int ldsDc = -1;
    if (!set.dcAlts.empty() && ir.has_nonlinear) {
        VariantOptions dcOpt{"csim_dc_sched_kernel", false, false, leanBudget > 0 ? leanBudget : 0};
        dcOpt.dcMode = true;
        ldsDc = emitVariant(dcOpt, nullptr);
        if (ldsDc >= 0) {
            VariantOptions dcFaith{"csim_dc_faithful_kernel", false, false, leanBudget > 0 ? leanBudget : 0};
            dcFaith.dcMode = true;
            dcFaith.faithful = true;
            if (emitVariant(dcFaith, nullptr) < 0) return std::string();     // cannot happen: same LDS image
        }
    }
    const bool haveDc = ldsDc >= 0;

    src << groupSrc << quadSrc;

    char hbuf[32];
    std::snprintf(hbuf, sizeof hbuf, "0x%016llxull", static_cast<unsigned long long>(hash));
    char tbuf[32];
    std::snprintf(tbuf, sizeof tbuf, "0x%016llxull",
                  static_cast<unsigned long long>(scheduleHash(ir, PivotSchedule::identity(N))));
    src << "extern \"C\" unsigned long long csim_sched_hash(void) { return " << hbuf << "; }\n"
        << "extern \"C\" unsigned long long csim_sched_topology(void) { return " << tbuf << "; }\n"
        << "extern \"C\" const char* csim_sched_info(void) { return \"" << label << " N=" << N << " schedule="
        << set.str() << " lds_doubles_per_lane=" << ldsLean << "/" << ldsRich
        << " ops_per_solve: fma=" << leanStats.nFma << " mul=" << leanStats.nMul << " addsub=" << leanStats.nAddSub
        << " recip=" << leanStats.nRecip << " cmp=" << leanStats.nCmp;
    if (haveGroup)
        src << " group16_wave_ops_per_solve: bcast=" << groupPlan.nBcast << " fma=" << groupPlan.nFma << " mul=" << groupPlan.nMul
            << " cmp=" << groupPlan.nCmp << " recip=" << groupPlan.nRecip;
    if (haveQuad)
        src << " group4_wave_ops_per_solve: bcast=" << quadPlan.nBcast << " fma=" << quadPlan.nFma << " mul=" << quadPlan.nMul
            << " cmp=" << quadPlan.nCmp << " recip=" << quadPlan.nRecip;
    src << "\"; }\n"
        << "// doubles per instance of the work area csim_sched_launch needs (0: none): the linear-circuit kernel parks\n"
        << "// its factors there\n"
        << "extern \"C\" int csim_sched_work_doubles(void) { return " << (haveLinear ? linWork : 0) << "; }\n"
        << "// 1 when this library carries csim_tran_faithful_kernel (csim_sched_launch variant 3)\n"
        << "extern \"C\" int csim_sched_has_faithful(void) { return " << (haveFaithful ? 1 : 0) << "; }\n"
        << "// 16 when this library also carries csim_tran_group_kernel (sixteen lanes per instance, one solve body per schedule)\n"
        << "extern \"C\" int csim_sched_group_lanes(void) { return " << (haveGroup ? 16 : 0) << "; }\n"
        << "// 4 when this library also carries csim_tran_group4_kernel (four lanes per instance; csim_sched_launch variant 4)\n"
        << "extern \"C\" int csim_sched_group4_lanes(void) { return " << (haveQuad ? 4 : 0) << "; }\n"
        << "// instances of that kernel one CU holds at a time: 16 per workgroup, one wave per SIMD, 160 KB of LDS\n"
        << "extern \"C\" int csim_sched_group4_per_cu(void) { return " << (haveQuad ? 16 * std::min(4, (160 * 1024) / (quadPlan.ldsDoubles * 16 * 8)) : 0) << "; }\n"
        << "// lanes per instance of the linear-circuit kernel: 16 (registers), 1 (LDS + streamed tape), 0 (not a linear circuit)\n"
        << "extern \"C\" int csim_sched_linear_lanes(void) { return " << (haveLinear16 ? 16 : (haveLinear ? 1 : 0)) << "; }\n";
    src << ""
        << "// the recorded alternatives, [n_alts][N] pivot row positions (the engine hands them to the\n"
        << "// general kernel so that it can tell when an instance is back on a known sequence)\n"
        << "extern \"C\" const int* csim_sched_alts(int* nAlts, int* n)\n{\n    static const int table[] = {";
    for (std::size_t a = 0; a < set.alts.size(); ++a)
        for (int k = 0; k < N; ++k) src << (a + k ? ", " : "") << set.alts[a].pivotPos[static_cast<std::size_t>(k)];
    src << "};\n    *nAlts = " << set.alts.size() << ";\n    *n = " << N << ";\n    return table;\n}\n"
        << "// DC operating point: recorded alternatives and launcher (n_alts == 0: no DC kernel in this library)\n"
        << "extern \"C\" const int* csim_sched_dc_alts(int* nAlts, int* n)\n{\n    static const int table[] = {";
    if (haveDc || haveLinearDc) {
        for (std::size_t a = 0; a < (haveDc ? set.dcAlts.size() : 1); ++a)
            for (int k = 0; k < N; ++k) src << (a + k ? ", " : "") << set.dcAlts[a].pivotPos[static_cast<std::size_t>(k)];
    } else {
        src << "0";
    }
    src << "};\n    *nAlts = " << (haveDc ? set.dcAlts.size() : (haveLinearDc ? 1 : 0)) << ";\n    *n = " << N << ";\n    return table;\n}\n"
        << "// variant 0 = the faithful kernel, 2 = the fast one; only (or null) = mask of the instances to run;\n"
        << "// an instance the kernel cannot finish (pivot check, non-finite solve, guarded decision) gets fallback[b] = 1\n"
        << "extern \"C\" int csim_sched_dc_launch(const double* params, int B, double* xout, int* iters, unsigned* status,\n"
        << "                                    unsigned char* fallback, int* violFlag, const unsigned char* only,\n"
        << "                                    double* work, void* stream, int variant)\n{\n";
    if (haveLinearDc)
        src << "    if (B <= 0) return 0;\n    (void)variant;\n"
            << "    if (!work) return (int)hipErrorInvalidValue;     // the linear-circuit kernels need their work area\n"
            << "    hipLaunchKernelGGL(csim_dc_linear_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream,\n"
            << "                       params, B, xout, iters, status, fallback, violFlag, only, work);\n"
            << "    return (int)hipGetLastError();\n}\n";
    else if (haveDc)
        src << "    if (B <= 0) return 0;\n    (void)work;\n"
            << "    if (variant == 2)\n"
            << "        hipLaunchKernelGGL(csim_dc_sched_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream,\n"
            << "                           params, B, xout, iters, status, fallback, violFlag, only);\n"
            << "    else\n"
            << "        hipLaunchKernelGGL(csim_dc_faithful_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)stream,\n"
            << "                           params, B, xout, iters, status, fallback, violFlag, only);\n"
            << "    return (int)hipGetLastError();\n}\n";
    else
        src << "    (void)params; (void)B; (void)xout; (void)iters; (void)status; (void)fallback; (void)violFlag; (void)only; (void)work; (void)stream; (void)variant;\n    return -1;\n}\n";
    src << "// variant: 0/1 = lean (measured fastest at every batch size: park-budget sweep in DESIGN.md),\n"
        << "//          2 = rich (tuning aid), 10+k = sweep kernels when generated with CSIM_CG_SWEEP\n"
        << "// auxiliaries of the hand-over protocol (device pointers; engine_internal.hpp holds the same struct):\n"
        << "//   fallback[B]  why an instance left a kernel unfinished (1: a check failed / slow step; 2, set by the engine: a\n"
        << "//                near-threshold decision is to be redone -- the faithful kernel then runs ONE step of it)\n"
        << "//   done[B]      steps of this launch completed;  flags[2]: [0] some instance is unfinished, [1] some\n"
        << "//                near-threshold decision awaits verification;  work: factor store of the linear-circuit kernel\n"
        << "//   nearX[N][B], nearStep[B], nearIt[B], nearItAfter[B]: state at the start of an instance's first near-threshold\n"
        << "//                step of the launch, that step (1-based), its pass count, passes counted from that step on\n"
        << "struct csim_sched_aux { unsigned char* fallback; int* done; int* flags; double* work; double* nearX; int* nearStep; int* nearIt; long long* nearItAfter; };\n"
        << "#define CSIM_AUX_ARGS aux->fallback, aux->done, aux->flags, aux->nearX, aux->nearStep, aux->nearIt, aux->nearItAfter\n"
        << "extern \"C\" int csim_sched_launch(const double* params, int B, double dt, long long stepFirst, long long nSteps,\n"
        << "                                 const int* probeEq, int nProbe, int outStride, double* wave, double* xio,\n"
        << "                                 long long* iters, unsigned* status, int* stepIters, const csim_sched_aux* aux,\n"
        << "                                 void* stream, int variant)\n{\n"
        << "    if (B <= 0) return 0;\n"
        << "    double* const work = aux->work;\n"
        << "    const unsigned waves = (unsigned)((B + 63) / 64);\n"
        << "    const bool rich = " << (haveRich ? "(variant == 2)" : "false") << ";\n"
        ;
    src << "    (void)work;\n";
    if (haveLinear16)
        src << "    if (work) {   // linear circuit: factor once per launch (lane per instance), then the time steps (16 lanes per instance)\n"
            << "        // (lanes per workgroup of the factorisation: generator option lin_factor_block)\n"
            << "        const unsigned fl = " << (gopt.linFactorBlock ? std::to_string(gopt.linFactorBlock) + "u" : std::string("B <= 16384 ? 16u : (B <= 32768 ? 32u : 64u)")) << ";\n"
            << "        hipLaunchKernelGGL(csim_lin16_factor_kernel, dim3(((unsigned)B + fl - 1) / fl), dim3(fl), 0, (hipStream_t)stream,\n"
            << "                           params, B, dt, nSteps, outStride, aux->done, aux->fallback, work);\n"
            << "        hipLaunchKernelGGL(csim_tran_linear16_kernel, dim3((unsigned)((B + 3) / 4)), dim3(64), 0, (hipStream_t)stream,\n"
            << "                           params, B, dt, stepFirst, nSteps, probeEq, nProbe, outStride, wave, xio, iters, status,\n"
            << "                           stepIters, aux->fallback, aux->done, aux->flags, work);\n"
            << "        return (int)hipGetLastError();\n    }\n";
    else if (haveLinear)
        src << "    if (work) {   // linear circuit: factor once per launch, substitute once per step\n"
            << "        hipLaunchKernelGGL(csim_tran_linear_kernel, dim3((unsigned)((B + " << linLanes - 1 << ") / " << linLanes << ")), dim3(" << linLanes << "), 0, (hipStream_t)stream,\n"
            << "                           params, B, dt, stepFirst, nSteps, probeEq, nProbe, outStride, wave, xio, iters, status,\n"
            << "                           stepIters, aux->fallback, aux->done, aux->flags, work);\n"
            << "        return (int)hipGetLastError();\n    }\n";
    if (haveFaithful)
        src << "    if (variant == 3) {\n"
            << "        hipLaunchKernelGGL(csim_tran_faithful_kernel, dim3(waves), dim3(64), 0, (hipStream_t)stream,\n"
            << "                           params, B, dt, stepFirst, nSteps, probeEq, nProbe, outStride, wave, xio, iters, status,\n"
            << "                           stepIters, CSIM_AUX_ARGS);\n"
            << "        return (int)hipGetLastError();\n    }\n";
    if (haveGroup)
        src << "    if (variant == 16) {\n"
            << "        hipLaunchKernelGGL(csim_tran_group_kernel, dim3((unsigned)((B + 3) / 4)), dim3(64), 0, (hipStream_t)stream,\n"
            << "                           params, B, dt, stepFirst, nSteps, probeEq, nProbe, outStride, wave, xio, iters, status,\n"
            << "                           stepIters, CSIM_AUX_ARGS);\n"
            << "        return (int)hipGetLastError();\n    }\n";
    if (haveQuad)
        src << "    if (variant == 4) {\n"
            << "        hipLaunchKernelGGL(csim_q4::csim_tran_group4_kernel, dim3((unsigned)((B + 15) / 16)), dim3(64), 0, (hipStream_t)stream,\n"
            << "                           params, B, dt, stepFirst, nSteps, probeEq, nProbe, outStride, wave, xio, iters, status,\n"
            << "                           stepIters, CSIM_AUX_ARGS);\n"
            << "        return (int)hipGetLastError();\n    }\n";
    for (std::size_t k = 0; k < sweep.size(); ++k)
        src << "    if (variant == " << (10 + k) << ") { hipLaunchKernelGGL(csim_tran_sched_kernel_sweep" << k
            << ", dim3(waves), dim3(64), 0, (hipStream_t)stream, params, B, dt, stepFirst, nSteps, probeEq, nProbe, outStride, wave, xio, iters, status, stepIters, CSIM_AUX_ARGS); return (int)hipGetLastError(); }\n";
    if (haveRich)
        src << "    if (rich) {\n"
            << "        hipLaunchKernelGGL(csim_tran_sched_kernel_rich, dim3(waves), dim3(64), 0, (hipStream_t)stream,\n"
            << "                           params, B, dt, stepFirst, nSteps, probeEq, nProbe, outStride, wave, xio, iters, status,\n"
            << "                           stepIters, CSIM_AUX_ARGS);\n"
            << "        return (int)hipGetLastError();\n    }\n";
    src << "    (void)rich; (void)waves;\n";
    if (haveLinear)
        src << "    return (int)hipErrorInvalidValue;   // the linear-circuit kernel needs its work area\n}\n";
    else
        src << "    hipLaunchKernelGGL(csim_tran_sched_kernel, dim3(waves), dim3(64), 0, (hipStream_t)stream,\n"
            << "                       params, B, dt, stepFirst, nSteps, probeEq, nProbe, outStride, wave, xio, iters, status,\n"
            << "                       stepIters, CSIM_AUX_ARGS);\n"
            << "    return (int)hipGetLastError();\n}\n";
    return src.str();
}

} // namespace csim
