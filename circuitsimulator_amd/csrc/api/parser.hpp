// parser.hpp -- SPICE-style netlist reader.
//
// Same entry points as the reference's include/parser.hpp (NetlistParser
// {parseFile, parseStream} :9-64, parseNetlist() :67-75) and the same dialect
// (src/parser.cpp; summarised in SURVEY.md Appendix B), so tests/*.sp of the
// reference drop in unchanged.  Runs once per netlist; not accelerated.
#pragma once

#include <iosfwd>
#include <string>
#include <vector>

#include "circuit.hpp"
#include "sim.hpp"

class NetlistParser {
public:
    NetlistParser(Circuit& circuit, SimulationConfig& simConfig);

    // false only if the file cannot be opened; syntax errors are reported on
    // stderr, the offending statement is skipped and parsing continues
    bool parseFile(const std::string& filename);
    bool parseStream(std::istream& in, const std::string& originName = "<stream>");

private:
    struct Statement {
        int lineNo = 0;                     // first physical line
        std::string raw;                    // joined logical line, comments stripped
        std::vector<std::string> tokens;    // whitespace separated
    };

    Circuit& ckt;
    SimulationConfig& sim;
    std::string sourceName;
    std::vector<Statement> stmts;

    void lex(std::istream& in);
    void parseStatements();

    void deviceStatement(const Statement& st);
    void dotCard(const Statement& st);

    void twoTerminal(const Statement& st, char kind);
    void voltageSource(const Statement& st);
    void currentSource(const Statement& st);
    void mosfet(const Statement& st);

    void modelCard(const Statement& st);
    void tranCard(const Statement& st);
    void dcCard(const Statement& st);
    void acCard(const Statement& st);
    void hbCard(const Statement& st);
    void printCard(const Statement& st);
    void plotNvCard(const Statement& st);
    void plotNcCard(const Statement& st);

    static ProbeSpec probeFromToken(const std::string& token);
};

inline bool parseNetlist(const std::string& filename, Circuit& ckt, SimulationConfig& sim)
{
    NetlistParser parser(ckt, sim);
    const bool ok = parser.parseFile(filename);
    sim.ensureDefaultOp();
    return ok;
}
