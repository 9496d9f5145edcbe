// dcanalysis.hpp -- DC operating point.
//
// Same free functions as the reference's include/dcanalysis.hpp:8-14; the
// work happens in the HIP kernel k_dc_general (engine/kernels_general.hip),
// which restates dcSolveDirectLU / dcSolveNewtonLU / ConvController::update
// (src/dcanalysis.cpp:46-68, 95-163, 268-307).  Non-convergence is not an
// exception upstream either: a WARNING is printed to stderr and the last
// iterate is returned.
#pragma once

#include "linalg.hpp"

class Circuit;

// DC operating point (direct LU for linear circuits, source ramp + damped
// Newton for circuits with MOSFETs).  Needs a HIP device.
Eigen::VectorXd dcSolve(const Circuit& ckt);
Eigen::VectorXd dcSolveLU(const Circuit& ckt);
// the same with the Gauss-Seidel inner solver (reference include/dcanalysis.hpp:14,
// src/dcanalysis.cpp:71-92,166-237): HIP kernel k_dc_gs.  Upstream nothing calls it; on circuits with
// voltage sources every inner solve diverges and the zero vector comes back, here as there.
Eigen::VectorXd dcSolveGaussSeidel(const Circuit& ckt);

struct ConvStatus {
    Eigen::VectorXd xNext;
    double alphaNext;
    double gminNext;
    double error;
    bool converged;
};

// The reference's convergence controller (src/dcanalysis.cpp:264-307).  The DC kernels run this rule
// per instance on the device; the host class is kept, update() included, for callers of the
// reference's header -- plain host arithmetic on the two vectors it is given, used by no analysis here.
class ConvController {
    double alphaMin, alphaMax;
    double gminHighBase, gminLowBase, gminAbsMax;
    double fastConvRatio, slowConvRatio;
public:
    ConvController();
    ConvStatus update(const Eigen::VectorXd& x, const Eigen::VectorXd& xRaw, double prevErr, int iter,
                      double alphaCurrent, double gminCurrent, double rampScale, double tol) const;
    double baseGmin(double rampScale) const;
    double alphaLow() const { return alphaMin; }
    double alphaHigh() const { return alphaMax; }
    double gminCeiling() const { return gminAbsMax; }
    double fastRatio() const { return fastConvRatio; }
    double slowRatio() const { return slowConvRatio; }
    double initialAlphaLU() const { return 0.5; }
    double initialAlphaGS() const { return 0.7; }
};
