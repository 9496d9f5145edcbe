// netlist_internal.hpp -- what a csim_netlist handle holds (library-private).
#pragma once

#include <string>
#include <vector>

#include "csim.h"
#include "../api/circuit.hpp"
#include "../api/parser.hpp"
#include "../api/sim.hpp"

struct csim_netlist {
    Circuit ckt;
    SimulationConfig sim;
    csim::CircuitIR cir;
    std::vector<int> probeEq;      // .PLOTNV / .PRINT node-voltage probes
    std::string csvHeader;
};

namespace csim {
void setError(const std::string& msg);
}
