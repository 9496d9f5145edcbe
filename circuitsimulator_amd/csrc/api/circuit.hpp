// circuit.hpp -- circuit model: nodes, devices, MOS models, equation indexing,
// and the flattening step Circuit -> csim_ir that feeds the engine.
//
// Public surface mirrors the reference's include/circuit.hpp (Node :11-19,
// MosModel :22-31, Circuit :33-66).  Equation numbering must be bit-exact with
// the reference (src/circuit.cpp:42-61): node equations in node-creation
// order skipping ground, then V sources and inductors interleaved in element
// order.
#pragma once

#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "element.hpp"
#include "utils.hpp"

struct Node {
    int id;
    std::string name;
    int eqIndex;                          // MNA row/column, -1 for ground
    std::vector<int> attachedElements;    // indices into Circuit::elements

    Node(int i, const std::string& n) : id(i), name(n), eqIndex(-1) {}
};

struct MosModel {
    std::string name;        // model id as written on the .MODEL card
    bool isP = false;        // sign of VT on the card
    double VT     = 0.7;     // stored positive
    double MU     = 1e-3;
    double COX    = 1e-3;
    double LAMBDA = 0.0;
    double CJO    = 0.0;
};

class Circuit {
public:
    std::vector<Node> nodes;
    std::vector<std::shared_ptr<Element>> elements;
    std::unordered_map<std::string, int> nodeNameToId;
    std::unordered_map<std::string, MosModel> mosModels;

    int getOrCreateNode(const std::string& name);

    int numNodeEquations() const;
    int numVoltageBranches() const;
    int numUnknowns() const;
    void assignEquationIndices();

    void addResistor(const std::string& name, const std::string& n1, const std::string& n2, double value);
    void addCapacitor(const std::string& name, const std::string& n1, const std::string& n2, double value);
    void addInductor(const std::string& name, const std::string& n1, const std::string& n2, double value);
    void addCurrentSource(const std::string& name, const std::string& np, const std::string& nm,
                          const SourceSpec& spec);
    void addVoltageSource(const std::string& name, const std::string& np, const std::string& nm,
                          const SourceSpec& spec);
    void addMosfet(const std::string& name, const std::string& nd, const std::string& ng,
                   const std::string& ns, const std::string& modelId, double W, double L);

    void addMosModel(const MosModel& m);
    const MosModel* findMosModel(const std::string& id) const;

    void printConnectivity() const;

private:
    // register a freshly built element with the nodes it touches
    void link(std::shared_ptr<Element> e, const std::vector<int>& touched);
};

namespace csim {

// owning storage behind a csim_ir view
struct CircuitIR {
    std::vector<int32_t> kind, eq, branchEq, paramSlot, wave, waveN;
    std::vector<double> nominal;          // P nominal parameter values
    // Monte-Carlo recipe per parameter slot (see engine/mc.hip):
    //   mcKind[p]: 0 fixed, 1 scale by (1+sigma z), 2 MOS K rebuilt from MU draw
    //   mcAux[p] : for kind 2 the product COX*(W/L) pieces: see mcMu/mcCox/mcWL
    std::vector<int32_t> mcKind;
    std::vector<double> mcMu, mcCox, mcW, mcL;
    // names for output headers: node name per node equation, element name per
    // branch equation
    std::vector<std::string> eqNames;
    mutable csim_ir ir{};

    // (re)binds the array pointers, so the view survives moves and copies
    const csim_ir* view() const
    {
        ir.kind = kind.data();
        ir.eq = eq.data();
        ir.branch_eq = branchEq.data();
        ir.param_slot = paramSlot.data();
        ir.wave = wave.data();
        ir.wave_n = waveN.data();
        return &ir;
    }
};

// Circuit -> IR.  assignEquationIndices() must have run (as in the reference,
// src/main.cpp:34, this is the caller's duty).
CircuitIR flatten(const Circuit& ckt);

} // namespace csim
