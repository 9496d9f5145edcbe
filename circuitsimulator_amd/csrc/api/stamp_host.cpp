// stamp_host.cpp -- Element::stamp on the host, FOR SOURCE COMPATIBILITY ONLY.
//
// The reference's seam `virtual void Element::stamp(G, I, ckt, x, ctx) const`
// (include/element.hpp:28-31) is what third-party code written against its headers calls to
// add one device into a host matrix.  Nothing in this library's solve path does: the engine
// flattens every device with describe() and stamps inside the HIP kernels
// (engine/device_common.hpp, the generated kernels).  This file exists so that such callers keep
// compiling and get the reference's numbers; it is host arithmetic by design, it is not a fallback
// of any analysis, and tests/test_capi_symbols.py::test_product_does_not_touch_the_oracle together
// with the engine's CSIM_ERR_NO_DEVICE keep it that way.
//
// Formulas restated from the reference: Resistor::stamp src/element.cpp:9-32, CurrentSource::stamp
// :34-66, VoltageSource::stamp :83-123, Inductor::stamp (DC form) :156-178, MosfetBase::stamp
// :181-307; CapacitorElement::stamp is a no-op there (include/element.hpp:103-108).
#include <cmath>
#include <iostream>

#include "circuit.hpp"
#include "element.hpp"

namespace {

// additive entry with the reference's "if (eq >= 0)" ground guards
struct Adder {
    Eigen::MatrixXd& G;
    Eigen::VectorXd& I;
    void g(int r, int c, double v) const { if (r >= 0 && c >= 0) G(r, c) += v; }
    void i(int r, double v) const { if (r >= 0) I(r) += v; }
};

double volt(const Eigen::VectorXd& x, int eq) { return (eq >= 0 && eq < x.size()) ? x(eq) : 0.0; }

} // namespace

void Element::stamp(Eigen::MatrixXd& G, Eigen::VectorXd& I, const Circuit& ckt, const Eigen::VectorXd& x,
                    const AnalysisContext& ctx) const
{
    const csim::IrRecord r = describe(ckt);
    const Adder add{G, I};
    const int a = r.eq[0], b = r.eq[1];
    switch (r.kind) {
        case CSIM_R: {
            const double R = r.params[0];
            if (R == 0.0) { std::cerr << "Warning: resistor " << name << " has zero resistance.\n"; return; }
            const double g = 1.0 / R;
            add.g(a, a, g); add.g(b, b, g); add.g(a, b, -g); add.g(b, a, -g);
            return;
        }
        case CSIM_C:
            return;                                             // open circuit outside the transient companion
        case CSIM_I:
        case CSIM_V: {
            const SourceSpec* s = sourceSpec();
            if (!s) return;
            double val = 0.0;
            if (ctx.type == AnalysisType::OP || ctx.type == AnalysisType::DC) val = s->evalDC(ctx.sourceScale);
            else if (ctx.type == AnalysisType::TRAN) val = s->evalTran(ctx.time);
            else if (ctx.type == AnalysisType::AC || ctx.type == AnalysisType::NONE) return;
            // HB falls out of the reference's switch with the value 0 and is stamped
            if (r.kind == CSIM_I) { add.i(a, -val); add.i(b, val); return; }
            const int k = r.branchEq;
            if (k < 0 || k >= G.rows()) { std::cerr << "Internal error: invalid branchEqIndex for " << name << "\n"; return; }
            add.g(a, k, 1.0); add.g(b, k, -1.0); add.g(k, a, 1.0); add.g(k, b, -1.0);
            I(k) += val;
            return;
        }
        case CSIM_L: {                                          // a 0 V source at DC
            const int k = r.branchEq;
            if (k < 0 || k >= G.rows()) { std::cerr << "Internal error: invalid branchEqIndex for inductor " << name << "\n"; return; }
            add.g(a, k, 1.0); add.g(b, k, -1.0); add.g(k, a, 1.0); add.g(k, b, -1.0);
            return;
        }
        case CSIM_NMOS:
        case CSIM_PMOS: {
            const int D = r.eq[0], Gt = r.eq[1], S = r.eq[2];
            const double Vth = r.params[0], K = r.params[1], lambda = r.params[2];
            const double p = r.kind == CSIM_PMOS ? -1.0 : 1.0;
            const double Vd = volt(x, D), Vg = volt(x, Gt), Vs = volt(x, S);
            const double Vgs = p * (Vg - Vs), Vds = p * (Vd - Vs);
            double Ids0 = 0.0, gds0 = 1e-12, gm0 = 0.0;         // off: a 1e-12 S channel
            if (Vgs > Vth && Vds >= 0.0) {
                const double Vov = Vgs - Vth;
                if (Vds < Vov) { Ids0 = K * (Vov * Vds - 0.5 * Vds * Vds); gds0 = K * (Vov - Vds); gm0 = K * Vds; }
                else           { Ids0 = 0.5 * K * Vov * Vov;               gds0 = 0.0;             gm0 = K * Vov; }
            }
            double factor = 1.0 + lambda * Vds;
            if (factor < 0.0) factor = 0.0;
            const double Ids = p * (Ids0 * factor);
            const double gd = gds0 * factor + Ids0 * lambda;
            const double gg = gm0 * factor;
            const double gs = -(gd + gg);
            const double cst = Ids - gd * Vd - gg * Vg - gs * Vs;
            if (D >= 0) { add.g(D, D, gd); add.g(D, Gt, gg); add.g(D, S, gs); I(D) -= cst; }
            if (S >= 0) { add.g(S, D, -gd); add.g(S, Gt, -gg); add.g(S, S, -gs); I(S) += cst; }
            return;
        }
        default:
            return;
    }
}
