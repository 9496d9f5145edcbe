// tanalisis.hpp -- transient analysis (backward Euler + damped Newton).
// Same two entry points as the reference's include/tanalisis.hpp:9-17.
#pragma once

#include <string>

#include "circuit.hpp"
#include "linalg.hpp"
#include "sim.hpp"

// DC operating point used as the t = 0 state (wraps dcSolve)
Eigen::VectorXd computeDcOperatingPoint(const Circuit& ckt);

// Runs .TRAN tstep tstop [tstart] on the GPU and writes the reference's CSV:
// "time,V(<node>)...,I(<V source | inductor>)...", std::scientific with 9
// decimals, rows with t < tstart suppressed (src/tanalisis.cpp:189-231).
// Bad .TRAN numbers or an unopenable file: message on stderr and return;
// a non-finite solve: throws std::runtime_error, like the reference (:360-362).
void runTransientAnalysisBackwardEuler(const Circuit& ckt, const SimulationConfig& sim,
                                       const std::string& outFile);
