// netlist_capi.cpp -- C-ABI over the C++ front-end: parse, index, flatten.
// Host only.  Replaces the parseNetlist() + assignEquationIndices() preamble
// every caller of the reference performs (src/main.cpp:29,34).
#include "netlist_internal.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "mc_draw.h"

namespace csim {

thread_local std::string g_lastError;

void setError(const std::string& msg) { g_lastError = msg; }

} // namespace csim

namespace {

int finishNetlist(csim_netlist* nl)
{
    nl->sim.ensureDefaultOp();
    nl->ckt.assignEquationIndices();
    nl->cir = csim::flatten(nl->ckt);
    nl->cir.view();

    // node-voltage probes named by .PLOTNV / .PRINT (first mention wins)
    for (const PrintCommand& pc : nl->sim.printCommands) {
        for (const ProbeSpec& p : pc.probes) {
            if (p.kind != ProbeKind::NodeVoltage || p.node1.empty()) continue;
            const auto hit = nl->ckt.nodeNameToId.find(p.node1);
            if (hit == nl->ckt.nodeNameToId.end()) continue;
            const int eq = nl->ckt.nodes[static_cast<std::size_t>(hit->second)].eqIndex;
            if (eq < 0) continue;
            bool seen = false;
            for (int q : nl->probeEq) seen = seen || (q == eq);
            if (!seen) nl->probeEq.push_back(eq);
        }
    }

    // reference CSV header: time, V(node) in node order, I(elem) for V sources
    // and inductors in element order (src/tanalisis.cpp:191-206) == equation order
    std::ostringstream h;
    h << "time";
    const csim_ir* ir = nl->cir.view();
    for (int eq = 0; eq < ir->n_unknowns; ++eq)
        h << (eq < ir->n_node_eq ? ",V(" : ",I(") << nl->cir.eqNames[static_cast<std::size_t>(eq)] << ")";
    nl->csvHeader = h.str();
    return CSIM_OK;
}

} // namespace

extern "C" {

const char* csim_last_error(void) { return csim::g_lastError.c_str(); }
const char* csim_version(void) { return "circuitsimulator_amd 0.1 (gfx950)"; }

int csim_netlist_parse_file(const char* path, csim_netlist** out)
{
    if (!path || !out) { csim::setError("csim_netlist_parse_file: null argument"); return CSIM_ERR_ARG; }
    *out = nullptr;
    auto* nl = new csim_netlist();
    NetlistParser parser(nl->ckt, nl->sim);
    if (!parser.parseFile(path)) {
        delete nl;
        csim::setError(std::string("cannot open netlist file ") + path);
        return CSIM_ERR_IO;
    }
    finishNetlist(nl);
    *out = nl;
    return CSIM_OK;
}

int csim_netlist_parse_text(const char* text, int64_t len, csim_netlist** out)
{
    if (!text || len < 0 || !out) { csim::setError("csim_netlist_parse_text: bad argument"); return CSIM_ERR_ARG; }
    *out = nullptr;
    auto* nl = new csim_netlist();
    std::istringstream in(std::string(text, static_cast<std::size_t>(len)));
    NetlistParser parser(nl->ckt, nl->sim);
    parser.parseStream(in, "<memory>");
    finishNetlist(nl);
    *out = nl;
    return CSIM_OK;
}

void csim_netlist_free(csim_netlist* nl) { delete nl; }

const csim_ir* csim_netlist_ir(const csim_netlist* nl) { return nl ? nl->cir.view() : nullptr; }

int csim_netlist_counts(const csim_netlist* nl, int32_t* n_nodes, int32_t* n_elems,
                        int32_t* n_unknowns, int32_t* n_node_eq, int32_t* n_branch_eq)
{
    if (!nl) return CSIM_ERR_ARG;
    if (n_nodes)     *n_nodes = static_cast<int32_t>(nl->ckt.nodes.size());
    if (n_elems)     *n_elems = static_cast<int32_t>(nl->ckt.elements.size());
    if (n_unknowns)  *n_unknowns = nl->ckt.numUnknowns();
    if (n_node_eq)   *n_node_eq = nl->ckt.numNodeEquations();
    if (n_branch_eq) *n_branch_eq = nl->ckt.numVoltageBranches();
    return CSIM_OK;
}

int csim_netlist_nominal_params(const csim_netlist* nl, double* out)
{
    if (!nl || !out) return CSIM_ERR_ARG;
    std::memcpy(out, nl->cir.nominal.data(), sizeof(double) * nl->cir.nominal.size());
    return CSIM_OK;
}

const char* csim_netlist_eq_name(const csim_netlist* nl, int32_t eq)
{
    if (!nl || eq < 0 || eq >= static_cast<int32_t>(nl->cir.eqNames.size())) return nullptr;
    return nl->cir.eqNames[static_cast<std::size_t>(eq)].c_str();
}

int csim_netlist_node_eq(const csim_netlist* nl, const char* node_name)
{
    if (!nl || !node_name) return -2;
    const auto hit = nl->ckt.nodeNameToId.find(node_name);
    if (hit == nl->ckt.nodeNameToId.end()) return -2;
    return nl->ckt.nodes[static_cast<std::size_t>(hit->second)].eqIndex;
}

int csim_netlist_tran(const csim_netlist* nl, int32_t* enabled, double* tstep, double* tstop, double* tstart)
{
    if (!nl) return CSIM_ERR_ARG;
    if (enabled) *enabled = nl->sim.tran.enabled ? 1 : 0;
    if (tstep)   *tstep = nl->sim.tran.tstep;
    if (tstop)   *tstop = nl->sim.tran.tstop;
    if (tstart)  *tstart = nl->sim.tran.tstart;
    return CSIM_OK;
}

int csim_netlist_num_probes(const csim_netlist* nl) { return nl ? static_cast<int>(nl->probeEq.size()) : 0; }

int csim_netlist_probe_eq(const csim_netlist* nl, int32_t i)
{
    if (!nl || i < 0 || i >= static_cast<int32_t>(nl->probeEq.size())) return -2;
    return nl->probeEq[static_cast<std::size_t>(i)];
}

int csim_netlist_num_dc_sweeps(const csim_netlist* nl) { return nl ? static_cast<int>(nl->sim.dcSweeps.size()) : 0; }

int csim_netlist_dc_sweep(const csim_netlist* nl, int32_t i, int32_t* src_elem,
                          double* start, double* stop, double* step)
{
    if (!nl || i < 0 || i >= static_cast<int32_t>(nl->sim.dcSweeps.size())) return CSIM_ERR_ARG;
    const DCSweepConfig& dc = nl->sim.dcSweeps[static_cast<std::size_t>(i)];
    int found = -1;
    for (std::size_t e = 0; e < nl->ckt.elements.size(); ++e)
        if (toLower(nl->ckt.elements[e]->getName()) == toLower(dc.sourceName)) { found = static_cast<int>(e); break; }
    if (src_elem) *src_elem = found;
    if (start) *start = dc.start;
    if (stop)  *stop = dc.stop;
    if (step)  *step = dc.step;
    return CSIM_OK;
}

static int sweepSource(const csim_netlist* nl, int32_t i, int* elem, double* a, double* b, double* st)
{
    int32_t e = -1;
    if (csim_netlist_dc_sweep(nl, i, &e, a, b, st) != CSIM_OK) return CSIM_ERR_ARG;
    if (e < 0) return CSIM_ERR_ARG;
    const int kind = nl->cir.kind[static_cast<std::size_t>(e)];
    if (kind != CSIM_V && kind != CSIM_I) return CSIM_ERR_ARG;
    *elem = e;
    return CSIM_OK;
}

int64_t csim_netlist_dc_sweep_points(const csim_netlist* nl, int32_t i)
{
    int e = -1;
    double a = 0, b = 0, st = 0;
    if (!nl || sweepSource(nl, i, &e, &a, &b, &st) != CSIM_OK) return 0;
    if (st == 0.0 || (b - a) / st < 0.0) return 0;
    return static_cast<int64_t>(std::floor((b - a) / st + 1e-9)) + 1;
}

int csim_netlist_dc_sweep_params(const csim_netlist* nl, int32_t i, int64_t n_points, double* params, double* values)
{
    int e = -1;
    double a = 0, b = 0, st = 0;
    if (!nl || !params || n_points < 0 || sweepSource(nl, i, &e, &a, &b, &st) != CSIM_OK) {
        csim::setError("csim_netlist_dc_sweep_params: bad sweep");
        return CSIM_ERR_ARG;
    }
    const csim::CircuitIR& c = nl->cir;
    const int P = static_cast<int>(c.nominal.size());
    const int slot = c.paramSlot[static_cast<std::size_t>(e)];      // SourceSpec::dcValue
    for (int p = 0; p < P; ++p)
        for (int64_t j = 0; j < n_points; ++j)
            params[static_cast<int64_t>(p) * n_points + j] = c.nominal[static_cast<std::size_t>(p)];
    for (int64_t j = 0; j < n_points; ++j) {
        const double v = a + static_cast<double>(j) * st;
        params[static_cast<int64_t>(slot) * n_points + j] = v;
        if (values) values[j] = v;
    }
    return CSIM_OK;
}

int csim_netlist_csv_header(const csim_netlist* nl, char* buf, int32_t cap)
{
    if (!nl) return CSIM_ERR_ARG;
    const int need = static_cast<int>(nl->csvHeader.size());
    if (buf && cap > 0) {
        const int n = need < cap - 1 ? need : cap - 1;
        std::memcpy(buf, nl->csvHeader.data(), static_cast<std::size_t>(n));
        buf[n] = '\0';
    }
    return need;
}

int csim_netlist_mc_kinds(const csim_netlist* nl, int32_t* kinds)
{
    if (!nl || !kinds) return CSIM_ERR_ARG;
    std::memcpy(kinds, nl->cir.mcKind.data(), sizeof(int32_t) * nl->cir.mcKind.size());
    return CSIM_OK;
}

int csim_mc_params_host(const csim_netlist* nl, uint64_t seed, double sigma, int64_t b_first,
                        int32_t B, double* params)
{
    if (!nl || !params || B < 0 || b_first < 0) { csim::setError("csim_mc_params_host: bad argument"); return CSIM_ERR_ARG; }
    const csim::CircuitIR& c = nl->cir;
    const int P = static_cast<int>(c.nominal.size());
    for (int p = 0; p < P; ++p) {
        for (int b = 0; b < B; ++b) {
            const uint64_t inst = static_cast<uint64_t>(b_first + b);
            const int kind = c.mcKind[static_cast<std::size_t>(p)];
            const double z = kind ? csim_mc::draw_z(seed, inst, static_cast<uint64_t>(p)) : 0.0;
            params[static_cast<int64_t>(p) * B + b] = csim_mc::perturb(
                kind, c.nominal[static_cast<std::size_t>(p)], c.mcMu[static_cast<std::size_t>(p)],
                c.mcCox[static_cast<std::size_t>(p)], c.mcW[static_cast<std::size_t>(p)],
                c.mcL[static_cast<std::size_t>(p)], sigma, z);
        }
    }
    return CSIM_OK;
}

} // extern "C"
