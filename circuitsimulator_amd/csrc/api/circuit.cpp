// circuit.cpp -- Circuit bookkeeping and flattening to the engine's IR.
// Behaviour to match: reference src/circuit.cpp (getOrCreateNode :5-14,
// counts :16-40, assignEquationIndices :42-61, add* :63-168).
#include "circuit.hpp"

#include <cmath>
#include <iostream>

int Circuit::getOrCreateNode(const std::string& name)
{
    const auto hit = nodeNameToId.find(name);
    if (hit != nodeNameToId.end()) return hit->second;
    const int id = static_cast<int>(nodes.size());
    nodes.emplace_back(id, name);
    nodeNameToId.emplace(name, id);
    return id;
}

int Circuit::numNodeEquations() const
{
    int n = 0;
    for (const Node& nd : nodes) n += isGroundName(nd.name) ? 0 : 1;
    return n;
}

// only voltage sources and inductors carry a branch-current unknown
int Circuit::numVoltageBranches() const
{
    int n = 0;
    for (const auto& e : elements) {
        if (dynamic_cast<const VoltageSource*>(e.get()) || dynamic_cast<const Inductor*>(e.get())) ++n;
    }
    return n;
}

int Circuit::numUnknowns() const { return numNodeEquations() + numVoltageBranches(); }

void Circuit::assignEquationIndices()
{
    int next = 0;
    for (Node& nd : nodes) nd.eqIndex = isGroundName(nd.name) ? -1 : next++;
    for (auto& e : elements) {
        if (auto* vs = dynamic_cast<VoltageSource*>(e.get()))      vs->setBranchEqIndex(next++);
        else if (auto* ind = dynamic_cast<Inductor*>(e.get()))     ind->setBranchEqIndex(next++);
    }
}

void Circuit::link(std::shared_ptr<Element> e, const std::vector<int>& touched)
{
    const int idx = static_cast<int>(elements.size());
    elements.push_back(std::move(e));
    for (int nid : touched) nodes[static_cast<std::size_t>(nid)].attachedElements.push_back(idx);
}

void Circuit::addResistor(const std::string& name, const std::string& n1, const std::string& n2, double value)
{
    const int a = getOrCreateNode(n1), b = getOrCreateNode(n2);
    link(std::make_shared<Resistor>(name, a, b, value), {a, b});
}

void Circuit::addCapacitor(const std::string& name, const std::string& n1, const std::string& n2, double value)
{
    const int a = getOrCreateNode(n1), b = getOrCreateNode(n2);
    link(std::make_shared<CapacitorElement>(name, a, b, value), {a, b});
}

void Circuit::addInductor(const std::string& name, const std::string& n1, const std::string& n2, double value)
{
    const int a = getOrCreateNode(n1), b = getOrCreateNode(n2);
    link(std::make_shared<Inductor>(name, a, b, value), {a, b});
}

void Circuit::addCurrentSource(const std::string& name, const std::string& np, const std::string& nm,
                               const SourceSpec& spec)
{
    const int p = getOrCreateNode(np), m = getOrCreateNode(nm);
    link(std::make_shared<CurrentSource>(name, p, m, spec), {p, m});
}

void Circuit::addVoltageSource(const std::string& name, const std::string& np, const std::string& nm,
                               const SourceSpec& spec)
{
    const int p = getOrCreateNode(np), m = getOrCreateNode(nm);
    link(std::make_shared<VoltageSource>(name, p, m, spec), {p, m});
}

void Circuit::addMosfet(const std::string& name, const std::string& nd, const std::string& ng,
                        const std::string& ns, const std::string& modelId, double W, double L)
{
    const MosModel* model = findMosModel(modelId);
    if (!model) {
        std::cerr << "Unknown MOS model: " << modelId << "\n";
        return;
    }
    // node creation order D, G, S, then the bulk node "0" (always ground,
    // created even if nothing else references it: reference circuit.cpp:142)
    const int d = getOrCreateNode(nd), g = getOrCreateNode(ng), s = getOrCreateNode(ns);
    const int b = getOrCreateNode("0");

    const double K = model->MU * model->COX * (W / L);   // evaluation order of circuit.cpp:144
    const double vth = std::abs(model->VT);

    std::shared_ptr<MosfetBase> m;
    if (model->isP) m = std::make_shared<PMosElement>(name, d, g, s, b, vth, K, model->LAMBDA, model->CJO);
    else            m = std::make_shared<NMosElement>(name, d, g, s, b, vth, K, model->LAMBDA, model->CJO);
    m->setModelGeometry(model->MU, model->COX, W, L);
    link(m, {d, g, s, b});
}

void Circuit::addMosModel(const MosModel& m) { mosModels[m.name] = m; }

const MosModel* Circuit::findMosModel(const std::string& id) const
{
    const auto hit = mosModels.find(id);
    return hit == mosModels.end() ? nullptr : &hit->second;
}

void Circuit::printConnectivity() const
{
    std::cout << "========== nodes and attached elements ==========\n";
    for (const Node& nd : nodes) {
        std::cout << "Node " << nd.name << " (id=" << nd.id << ", eqIndex=" << nd.eqIndex << "): ";
        for (int ei : nd.attachedElements) std::cout << elements[static_cast<std::size_t>(ei)]->getName() << " ";
        std::cout << "\n";
    }
}

// ---------------------------------------------------------------- describe()

namespace {
int eqOf(const Circuit& ckt, int nodeId) { return ckt.nodes[static_cast<std::size_t>(nodeId)].eqIndex; }

void fillSource(csim::IrRecord& r, const SourceSpec& s)
{
    r.params.assign(1, s.dcValue);
    if (s.tran.type == WaveformType::PULSE) {
        const PulseSpec& p = s.tran.pulse;
        r.wave = CSIM_WAVE_PULSE;
        for (double v : {p.v1, p.v2, p.td, p.tr, p.tf, p.ton, p.per}) r.params.push_back(v);
    } else if (s.tran.type == WaveformType::PWL) {
        const PwlSpec& w = s.tran.pwl;
        const std::size_t n = w.t.size() < w.v.size() ? w.t.size() : w.v.size();
        r.wave = CSIM_WAVE_PWL;
        r.waveN = static_cast<int>(n);
        for (std::size_t i = 0; i < n; ++i) r.params.push_back(w.t[i]);
        for (std::size_t i = 0; i < n; ++i) r.params.push_back(w.v[i]);
    } else {
        r.wave = (s.tran.type == WaveformType::SIN) ? CSIM_WAVE_SIN : CSIM_WAVE_NONE;
        for (double v : {s.tran.sine.v0, s.tran.sine.va, s.tran.sine.freq, s.tran.sine.td, s.tran.sine.phi}) r.params.push_back(v);
    }
    r.nParams = static_cast<int>(r.params.size());
}
} // namespace

csim::IrRecord Resistor::describe(const Circuit& ckt) const
{
    csim::IrRecord r;
    r.kind = CSIM_R;
    r.eq[0] = eqOf(ckt, nodeIds[0]);
    r.eq[1] = eqOf(ckt, nodeIds[1]);
    r.nParams = CSIM_PARAMS_R;
    r.params = {R};
    return r;
}

csim::IrRecord CapacitorElement::describe(const Circuit& ckt) const
{
    csim::IrRecord r;
    r.kind = CSIM_C;
    r.eq[0] = eqOf(ckt, nodeIds[0]);
    r.eq[1] = eqOf(ckt, nodeIds[1]);
    r.nParams = CSIM_PARAMS_C;
    r.params = {C};
    return r;
}

csim::IrRecord Inductor::describe(const Circuit& ckt) const
{
    csim::IrRecord r;
    r.kind = CSIM_L;
    r.eq[0] = eqOf(ckt, nodeIds[0]);
    r.eq[1] = eqOf(ckt, nodeIds[1]);
    r.branchEq = branchEqIndex;
    r.nParams = CSIM_PARAMS_L;
    r.params = {L};
    return r;
}

csim::IrRecord VoltageSource::describe(const Circuit& ckt) const
{
    csim::IrRecord r;
    r.kind = CSIM_V;
    r.eq[0] = eqOf(ckt, nodeIds[0]);
    r.eq[1] = eqOf(ckt, nodeIds[1]);
    r.branchEq = branchEqIndex;
    fillSource(r, spec);
    return r;
}

csim::IrRecord CurrentSource::describe(const Circuit& ckt) const
{
    csim::IrRecord r;
    r.kind = CSIM_I;
    r.eq[0] = eqOf(ckt, nodeIds[0]);
    r.eq[1] = eqOf(ckt, nodeIds[1]);
    fillSource(r, spec);
    return r;
}

csim::IrRecord MosfetBase::describe(const Circuit& ckt) const
{
    csim::IrRecord r;
    r.kind = isP ? CSIM_PMOS : CSIM_NMOS;
    for (int t = 0; t < 4; ++t) r.eq[t] = eqOf(ckt, nodeIds[static_cast<std::size_t>(t)]);
    r.nParams = CSIM_PARAMS_MOS;
    r.params = {Vth, K, lambda, Cj0};
    r.mosMu = mu_; r.mosCox = cox_; r.mosW = w_; r.mosL = l_;
    return r;
}

// ------------------------------------------------------------------ flatten

namespace csim {

CircuitIR flatten(const Circuit& ckt)
{
    CircuitIR out;
    const int nNode = ckt.numNodeEquations();
    const int nBranch = ckt.numVoltageBranches();
    const int nElem = static_cast<int>(ckt.elements.size());

    out.eqNames.assign(static_cast<std::size_t>(nNode + nBranch), std::string());
    for (const Node& nd : ckt.nodes)
        if (nd.eqIndex >= 0) out.eqNames[static_cast<std::size_t>(nd.eqIndex)] = nd.name;

    bool nonlinear = false;
    for (const auto& e : ckt.elements) {
        const IrRecord r = e->describe(ckt);
        out.kind.push_back(r.kind);
        for (int t = 0; t < 4; ++t) out.eq.push_back(r.eq[t]);
        out.branchEq.push_back(r.branchEq);
        out.wave.push_back(r.wave);
        out.waveN.push_back(r.waveN);
        out.paramSlot.push_back(static_cast<int32_t>(out.nominal.size()));
        if (r.branchEq >= 0 && r.branchEq < nNode + nBranch)
            out.eqNames[static_cast<std::size_t>(r.branchEq)] = e->getName();
        const bool mos = (r.kind == CSIM_NMOS || r.kind == CSIM_PMOS);
        nonlinear = nonlinear || mos;
        for (int p = 0; p < r.nParams; ++p) {
            out.nominal.push_back(r.params[static_cast<std::size_t>(p)]);
            int mc = 0;
            if (r.kind == CSIM_R || r.kind == CSIM_C || r.kind == CSIM_L) mc = 1;
            else if (mos && p == 0) mc = 1;    // VT
            else if (mos && p == 1) mc = 2;    // K, rebuilt from a MU draw
            out.mcKind.push_back(mc);
            out.mcMu.push_back(mc == 2 ? r.mosMu : 0.0);
            out.mcCox.push_back(mc == 2 ? r.mosCox : 0.0);
            out.mcW.push_back(mc == 2 ? r.mosW : 0.0);
            out.mcL.push_back(mc == 2 ? r.mosL : 0.0);
        }
    }

    out.ir.n_unknowns = nNode + nBranch;
    out.ir.n_node_eq = nNode;
    out.ir.n_branch_eq = nBranch;
    out.ir.n_elems = nElem;
    out.ir.n_params = static_cast<int32_t>(out.nominal.size());
    out.ir.has_nonlinear = nonlinear ? 1 : 0;
    csim_consts_default(&out.ir.k);
    out.view();
    return out;
}

} // namespace csim
