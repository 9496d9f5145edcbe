// element.hpp -- device classes of the circuit model.
//
// Class shapes follow the reference's include/element.hpp (Element :13-39,
// Resistor :41-52, CurrentSource :55-69, VoltageSource :71-89,
// CapacitorElement :93-109, Inductor :112-129, MosfetBase/NMosElement/
// PMosElement :132-170) so code that builds or inspects a Circuit keeps
// compiling.  The difference is what an element does for the solver: the
// reference's virtual stamp() adds numbers into a host matrix once per Newton
// iteration; here an element only DESCRIBES itself (describe() -> one IR
// record) and the stamping arithmetic runs inside the HIP kernels
// (engine/device_common.hpp), once per instance per iteration.
#pragma once

#include <string>
#include <vector>

#include "linalg.hpp"
#include "sim.hpp"
#include "csim_ir.h"

class Circuit;

namespace csim {
// one flattened element: what csim_ir stores per device
struct IrRecord {
    int kind = CSIM_R;              // csim_elem_kind
    int eq[4] = {-1, -1, -1, -1};   // terminal equation indices (-1 = ground)
    int branchEq = -1;
    int wave = CSIM_WAVE_NONE;
    int waveN = 0;                  // PWL: number of points
    int nParams = 0;
    std::vector<double> params;     // nParams values
    // how a Monte-Carlo draw perturbs this element (engine/mc.hip):
    //   scaleMask bit i: params[i] *= (1 + sigma z)
    //   MOS: Vth scaled directly; K rebuilt as (MU(1+sigma z))*COX*(W/L)
    double mosMu = 0.0, mosCox = 0.0, mosW = 0.0, mosL = 0.0;
};
}

class Element {
protected:
    std::string name;
    std::vector<int> nodeIds;   // indices into Circuit::nodes

public:
    Element(const std::string& n, const std::vector<int>& nodes) : name(n), nodeIds(nodes) {}
    virtual ~Element() {}

    const std::string& getName() const { return name; }
    const std::vector<int>& getNodeIds() const { return nodeIds; }

    // flatten this device for the engine: what the solve path uses instead of stamp()
    virtual csim::IrRecord describe(const Circuit& ckt) const = 0;

    // The reference's stamping seam (include/element.hpp:28-31), kept for source compatibility:
    // adds this device into a caller-owned HOST system at iterate x.  Host arithmetic for callers of
    // the reference's headers only -- no analysis of this library calls it (api/stamp_host.cpp).
    virtual void stamp(Eigen::MatrixXd& G, Eigen::VectorXd& I, const Circuit& ckt,
                       const Eigen::VectorXd& x, const AnalysisContext& ctx) const;
    // AC stamping is a stub upstream as well (include/element.hpp:33-38: the default does nothing,
    // no analysis calls it); kept so that overriding code compiles.
    virtual void stampAC(Eigen::MatrixXcd& /*Y*/, Eigen::VectorXcd& /*J*/, const Circuit& /*ckt*/,
                         double /*omega*/) const {}

protected:
    // sources hand their SourceSpec to the host stamp
    virtual const SourceSpec* sourceSpec() const { return nullptr; }
};

class Resistor : public Element {
    double R;
public:
    Resistor(const std::string& n, int n1, int n2, double r) : Element(n, {n1, n2}), R(r) {}
    double getR() const { return R; }
    csim::IrRecord describe(const Circuit& ckt) const override;
};

// current flows from nodeIds[0] to nodeIds[1] through the source
class CurrentSource : public Element {
    SourceSpec spec;
public:
    CurrentSource(const std::string& n, int np, int nm, const SourceSpec& s)
        : Element(n, {np, nm}), spec(s) {}
    const SourceSpec& getSpec() const { return spec; }
    const SourceSpec& setSpec() const { return spec; }   // (sic) name kept from the reference
    csim::IrRecord describe(const Circuit& ckt) const override;
protected:
    const SourceSpec* sourceSpec() const override { return &spec; }
};

class VoltageSource : public Element {
    SourceSpec spec;
    int branchEqIndex;
public:
    VoltageSource(const std::string& n, int np, int nm, const SourceSpec& s)
        : Element(n, {np, nm}), spec(s), branchEqIndex(-1) {}
    void setBranchEqIndex(int idx) { branchEqIndex = idx; }
    int  getBranchEqIndex() const { return branchEqIndex; }
    const SourceSpec& getSpec() const { return spec; }
    csim::IrRecord describe(const Circuit& ckt) const override;
protected:
    const SourceSpec* sourceSpec() const override { return &spec; }
};

// open circuit at DC, backward-Euler companion in transient
class CapacitorElement : public Element {
    double C;
public:
    CapacitorElement(const std::string& n, int n1, int n2, double c) : Element(n, {n1, n2}), C(c) {}
    double getC() const { return C; }
    csim::IrRecord describe(const Circuit& ckt) const override;
};

// 0 V source at DC (own branch current), Thevenin BE companion in transient
class Inductor : public Element {
    double L;
    int branchEqIndex;
public:
    Inductor(const std::string& n, int n1, int n2, double l)
        : Element(n, {n1, n2}), L(l), branchEqIndex(-1) {}
    void setBranchEqIndex(int idx) { branchEqIndex = idx; }
    int  getBranchEqIndex() const { return branchEqIndex; }
    double getL() const { return L; }
    csim::IrRecord describe(const Circuit& ckt) const override;
};

// Level-1 MOSFET, terminals D G S B (bulk is always node "0")
class MosfetBase : public Element {
protected:
    bool isP;
    double Vth;      // |VT|
    double K;        // MU*COX*(W/L)
    double lambda;
    double Cj0;
    // model/geometry values K was built from (kept for Monte-Carlo draws)
    double mu_ = 0.0, cox_ = 0.0, w_ = 0.0, l_ = 0.0;

public:
    MosfetBase(const std::string& n, int nd, int ng, int ns, int nb, bool isPchannel,
               double Vth_, double K_, double lambda_, double Cj0_)
        : Element(n, {nd, ng, ns, nb}), isP(isPchannel), Vth(Vth_), K(K_),
          lambda(lambda_), Cj0(Cj0_) {}

    bool   isPChannel() const { return isP; }
    double getVth() const { return Vth; }
    double getK() const { return K; }
    double getLambda() const { return lambda; }
    double getCj0() const { return Cj0; }
    void setModelGeometry(double mu, double cox, double w, double l) { mu_ = mu; cox_ = cox; w_ = w; l_ = l; }
    csim::IrRecord describe(const Circuit& ckt) const override;
};

class NMosElement : public MosfetBase {
public:
    NMosElement(const std::string& n, int nd, int ng, int ns, int nb,
                double Vth_, double K_, double lambda_, double Cj0_)
        : MosfetBase(n, nd, ng, ns, nb, false, Vth_, K_, lambda_, Cj0_) {}
};

class PMosElement : public MosfetBase {
public:
    PMosElement(const std::string& n, int nd, int ng, int ns, int nb,
                double Vth_, double K_, double lambda_, double Cj0_)
        : MosfetBase(n, nd, ng, ns, nb, true, Vth_, K_, lambda_, Cj0_) {}
};
