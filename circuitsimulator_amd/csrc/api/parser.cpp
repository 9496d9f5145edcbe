// parser.cpp -- netlist dialect of the reference (src/parser.cpp), restated:
//   physical lines, trailing CR dropped; '$' starts an inline comment; a line
//   whose first non-blank is '*' or ';' is a comment; a leading '+' continues
//   the previous logical line; tokens split on blanks.
//   Devices by first letter (any case): R C L V I M.   Cards: .MODEL (read in
//   a first pass, so it may follow its users) .TRAN .OP .DC .AC .HB .PRINT
//   .PLOTNV .PLOTNC; anything else is reported and skipped.
#include "parser.hpp"

#include <cctype>
#include <exception>
#include <fstream>
#include <iostream>
#include <sstream>

namespace {

std::string dropInlineComment(const std::string& s)
{
    const std::size_t at = s.find('$');
    return at == std::string::npos ? s : s.substr(0, at);
}

std::string clean(const std::string& s) { return rtrim(ltrim(dropInlineComment(s))); }

char upperHead(const std::string& tok)
{
    return static_cast<char>(std::toupper(static_cast<unsigned char>(tok[0])));
}

// position of the first '(' and of the last ')' in s, -1 when absent
void parenSpan(const std::string& s, int& open, int& close)
{
    open = close = -1;
    for (int i = 0; i < static_cast<int>(s.size()); ++i) {
        if (s[static_cast<std::size_t>(i)] == '(' && open < 0) open = i;
        if (s[static_cast<std::size_t>(i)] == ')') close = i;
    }
}

AnalysisType analysisFromToken(const std::string& tok)
{
    const std::string t = toLower(tok);
    if (t == "op")   return AnalysisType::OP;
    if (t == "dc")   return AnalysisType::DC;
    if (t == "ac")   return AnalysisType::AC;
    if (t == "tran") return AnalysisType::TRAN;
    if (t == "hb")   return AnalysisType::HB;
    return AnalysisType::NONE;
}

} // namespace

NetlistParser::NetlistParser(Circuit& circuit, SimulationConfig& simConfig) : ckt(circuit), sim(simConfig) {}

bool NetlistParser::parseFile(const std::string& filename)
{
    std::ifstream fin(filename);
    if (!fin) {
        std::cerr << "cannot open netlist file " << filename << "\n";
        return false;
    }
    return parseStream(fin, filename);
}

bool NetlistParser::parseStream(std::istream& in, const std::string& originName)
{
    sourceName = originName;
    lex(in);
    parseStatements();
    return true;
}

void NetlistParser::lex(std::istream& in)
{
    stmts.clear();
    std::string pending;       // logical line being assembled
    int pendingLine = 0;

    auto flush = [&]() {
        const std::string text = clean(pending);
        pending.clear();
        if (text.empty()) return;
        Statement st;
        st.lineNo = pendingLine;
        st.raw = text;
        std::istringstream split(text);
        for (std::string tok; split >> tok;) st.tokens.push_back(tok);
        if (!st.tokens.empty()) stmts.push_back(std::move(st));
    };

    std::string line;
    int lineNo = 0;
    while (std::getline(in, line)) {
        ++lineNo;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        const std::string s = clean(line);
        if (s.empty()) continue;
        if (s[0] == '*' || s[0] == ';') continue;

        if (s[0] == '+') {
            const std::string rest = ltrim(s.substr(1));
            if (pending.empty()) {          // stray continuation opens a statement
                pendingLine = lineNo;
                pending = rest;
            } else {
                pending += " ";
                pending += rest;
            }
        } else {
            if (!pending.empty()) flush();
            pendingLine = lineNo;
            pending = s;
        }
    }
    if (!pending.empty()) flush();
}

void NetlistParser::parseStatements()
{
    // pass 1: models, so devices may precede their .MODEL card
    for (const Statement& st : stmts)
        if (toLower(st.tokens[0]) == ".model") modelCard(st);

    // pass 2: everything else, in netlist order (this order defines node and
    // element numbering)
    for (const Statement& st : stmts) {
        const std::string& head = st.tokens[0];
        if (head[0] == '.') {
            if (toLower(head) != ".model") dotCard(st);
        } else {
            // a leading title line is not special-cased: like the reference
            // it is handed to the device dispatcher, which reports it
            deviceStatement(st);
        }
    }
    sim.ensureDefaultOp();
}

void NetlistParser::deviceStatement(const Statement& st)
{
    switch (upperHead(st.tokens[0])) {
        case 'R': case 'C': case 'L': twoTerminal(st, upperHead(st.tokens[0])); break;
        case 'V': voltageSource(st); break;
        case 'I': currentSource(st); break;
        case 'M': mosfet(st); break;
        default:
            std::cerr << "Line " << st.lineNo << ": unsupported element or syntax: " << st.raw << "\n";
    }
}

// Rname n1 n2 value | Cname ... | Lname ...
void NetlistParser::twoTerminal(const Statement& st, char kind)
{
    const auto& t = st.tokens;
    const char* what = kind == 'R' ? "resistor" : (kind == 'C' ? "capacitor" : "inductor");
    if (t.size() < 4) {
        std::cerr << "Line " << st.lineNo << ": invalid " << what << ": " << st.raw << "\n";
        return;
    }
    double value = 0.0;
    try {
        value = parseSpiceNumber(t[3]);
    } catch (const std::exception& e) {
        std::cerr << "Line " << st.lineNo << ": cannot parse " << kind << " value: " << e.what()
                  << " in '" << st.raw << "'\n";
        return;
    }
    if (kind == 'R')      ckt.addResistor(t[0], t[1], t[2], value);
    else if (kind == 'C') ckt.addCapacitor(t[0], t[1], t[2], value);
    else                  ckt.addInductor(t[0], t[1], t[2], value);
}

// PULSE / PWL waveforms.  The reference's netlist dialect has no syntax for them (its parser
// rejects the statement, src/parser.cpp:330-345) although its solver evaluates them
// (include/sim.hpp:80-138) for circuits built through the C++ API.  Accepted here as a superset:
//   PULSE v1 v2 [td [tr [tf [ton [per]]]]]      PWL t0 v0 t1 v1 ...
// with optional SPICE-style parentheses and commas.  Returns true when a waveform keyword was
// consumed (valid or not).
namespace {
bool startsWithKeyword(const std::string& token, const char* kw)
{
    const std::string low = toLower(token);
    const std::string k(kw);
    return low.compare(0, k.size(), k) == 0 && (low.size() == k.size() || low[k.size()] == '(');
}

bool parsePulseOrPwl(const std::vector<std::string>& t, std::size_t next, int lineNo, const std::string& raw,
                     SourceSpec& spec)
{
    if (next >= t.size()) return false;
    const bool pulse = startsWithKeyword(t[next], "pulse");
    const bool pwl = startsWithKeyword(t[next], "pwl");
    if (!pulse && !pwl) return false;

    std::string rest;
    for (std::size_t i = next; i < t.size(); ++i) rest += t[i] + " ";
    rest = rest.substr(pulse ? 5 : 3);
    for (char& c : rest) if (c == '(' || c == ')' || c == ',') c = ' ';
    std::vector<double> vals;
    try {
        std::istringstream is(rest);
        std::string tok;
        while (is >> tok) vals.push_back(parseSpiceNumber(tok));
    } catch (const std::exception& e) {
        std::cerr << "Line " << lineNo << ": cannot parse " << (pulse ? "PULSE" : "PWL") << " parameters: "
                  << e.what() << " in '" << raw << "'\n";
        return true;
    }
    if (pulse) {
        if (vals.size() < 2 || vals.size() > 7) {
            std::cerr << "Line " << lineNo << ": PULSE needs 2 to 7 parameters (v1 v2 td tr tf ton per): " << raw << "\n";
            return true;
        }
        vals.resize(7, 0.0);
        PulseSpec p;
        p.v1 = vals[0]; p.v2 = vals[1]; p.td = vals[2]; p.tr = vals[3]; p.tf = vals[4]; p.ton = vals[5]; p.per = vals[6];
        spec.tran.type = WaveformType::PULSE;
        spec.tran.pulse = p;
    } else {
        if (vals.size() < 2 || vals.size() % 2 != 0) {
            std::cerr << "Line " << lineNo << ": PWL needs (time value) pairs: " << raw << "\n";
            return true;
        }
        PwlSpec w;
        for (std::size_t i = 0; i < vals.size(); i += 2) { w.t.push_back(vals[i]); w.v.push_back(vals[i + 1]); }
        spec.tran.type = WaveformType::PWL;
        spec.tran.pwl = w;
    }
    return true;
}
} // namespace

// Vname np nm <value> [SIN ...] | Vname np nm DC <value> [SIN ...] |
// Vname np nm SIN v0 va freq [td [phi]]
void NetlistParser::voltageSource(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 4) {
        std::cerr << "Line " << st.lineNo << ": invalid voltage source: " << st.raw << "\n";
        return;
    }

    SourceSpec spec;
    std::size_t next = 3;      // token where a waveform keyword may start
    try {
        if (t.size() >= 5 && toLower(t[3]) == "dc") {
            spec.dcValue = parseSpiceNumber(t[4]);
            next = 5;
        } else if (toLower(t[3]) == "sin" || startsWithKeyword(t[3], "pulse") || startsWithKeyword(t[3], "pwl")) {
            spec.dcValue = 0.0;
            next = 3;
        } else {
            spec.dcValue = parseSpiceNumber(t[3]);
            next = 4;
        }
    } catch (const std::exception& e) {
        std::cerr << "Line " << st.lineNo << ": cannot parse V DC value: " << e.what()
                  << " in '" << st.raw << "'\n";
        return;
    }

    if (parsePulseOrPwl(t, next, st.lineNo, st.raw, spec)) {
        // superset of the reference dialect, see above
    } else if (next < t.size() && toLower(t[next]) == "sin") {
        // SIN v0 va freq [td [phi]] -- td in seconds, phi in radians
        if (t.size() < next + 4) {
            std::cerr << "Line " << st.lineNo << ": SIN needs at least 3 parameters (v0 va freq): "
                      << st.raw << "\n";
        } else {
            try {
                SinSpec s;
                s.v0   = parseSpiceNumber(t[next + 1]);
                s.va   = parseSpiceNumber(t[next + 2]);
                s.freq = parseSpiceNumber(t[next + 3]);
                if (t.size() > next + 4) s.td  = parseSpiceNumber(t[next + 4]);
                if (t.size() > next + 5) s.phi = parseSpiceNumber(t[next + 5]);
                spec.tran.type = WaveformType::SIN;
                spec.tran.sine = s;
            } catch (const std::exception& e) {
                std::cerr << "Line " << st.lineNo << ": cannot parse SIN parameters: " << e.what()
                          << " in '" << st.raw << "'\n";
            }
        }
    }
    ckt.addVoltageSource(t[0], t[1], t[2], spec);
}

// Iname np nm [DC] value        (+ optional PULSE / PWL waveform: superset, see above)
void NetlistParser::currentSource(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 4) {
        std::cerr << "Line " << st.lineNo << ": invalid current source: " << st.raw << "\n";
        return;
    }
    SourceSpec spec;
    try {
        const bool dcForm = t.size() >= 5 && toLower(t[3]) == "dc";
        if (startsWithKeyword(t[3], "pulse") || startsWithKeyword(t[3], "pwl")) {
            parsePulseOrPwl(t, 3, st.lineNo, st.raw, spec);
        } else {
            spec.dcValue = parseSpiceNumber(dcForm ? t[4] : t[3]);
            parsePulseOrPwl(t, dcForm ? 5 : 4, st.lineNo, st.raw, spec);
        }
    } catch (const std::exception& e) {
        std::cerr << "Line " << st.lineNo << ": cannot parse I value: " << e.what()
                  << " in '" << st.raw << "'\n";
        return;
    }
    ckt.addCurrentSource(t[0], t[1], t[2], spec);
}

// Mname nd ng ns model W L            (7 tokens)
// Mname nd ng ns <p|n> W L modelId    (8 tokens; the type letter is ignored)
void NetlistParser::mosfet(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() != 7 && t.size() != 8) {
        std::cerr << "Line " << st.lineNo << ": invalid MOSFET: " << st.raw << "\n";
        return;
    }
    const std::string modelId = (t.size() == 7) ? t[4] : t.back();
    double W = 0.0, L = 0.0;
    try {
        W = parseSpiceNumber(t[5]);
        L = parseSpiceNumber(t[6]);
    } catch (const std::exception& e) {
        std::cerr << "Line " << st.lineNo << ": cannot parse MOS W/L: " << e.what()
                  << " in '" << st.raw << "'\n";
        return;
    }
    ckt.addMosfet(t[0], t[1], t[2], t[3], modelId, W, L);
}

void NetlistParser::dotCard(const Statement& st)
{
    const std::string head = toLower(st.tokens[0]);
    if      (head == ".op")     sim.doOp = true;
    else if (head == ".dc")     dcCard(st);
    else if (head == ".tran")   tranCard(st);
    else if (head == ".ac")     acCard(st);
    else if (head == ".print")  printCard(st);
    else if (head == ".model")  modelCard(st);
    else if (head == ".hb")     hbCard(st);
    else if (head == ".plotnv") plotNvCard(st);
    else if (head == ".plotnc") plotNcCard(st);
    else std::cerr << "Line " << st.lineNo << ": unsupported control card: " << st.raw << "\n";
}

// .DC src start stop step   (recorded; executed by the batch API as a sweep axis)
void NetlistParser::dcCard(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 5) {
        std::cerr << "Line " << st.lineNo << ": invalid .DC syntax: " << st.raw << "\n";
        return;
    }
    DCSweepConfig dc;
    dc.sourceName = t[1];
    try {
        dc.start = parseSpiceNumber(t[2]);
        dc.stop  = parseSpiceNumber(t[3]);
        dc.step  = parseSpiceNumber(t[4]);
    } catch (const std::exception& e) {
        std::cerr << "Line " << st.lineNo << ": cannot parse .DC numbers: " << e.what()
                  << " in '" << st.raw << "'\n";
        return;
    }
    sim.dcSweeps.push_back(dc);
}

// .TRAN tstep tstop [tstart]
void NetlistParser::tranCard(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 3) {
        std::cerr << "Line " << st.lineNo << ": invalid .TRAN syntax: " << st.raw << "\n";
        return;
    }
    TranConfig cfg;
    try {
        cfg.tstep  = parseSpiceNumber(t[1]);
        cfg.tstop  = parseSpiceNumber(t[2]);
        cfg.tstart = t.size() >= 4 ? parseSpiceNumber(t[3]) : 0.0;
    } catch (const std::exception& e) {
        std::cerr << "Line " << st.lineNo << ": cannot parse .TRAN numbers: " << e.what()
                  << " in '" << st.raw << "'\n";
        return;
    }
    cfg.enabled = true;
    sim.tran = cfg;
}

// .AC {LIN|DEC|OCT} npoints fstart fstop
void NetlistParser::acCard(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 5) {
        std::cerr << "Line " << st.lineNo << ": invalid .AC syntax: " << st.raw << "\n";
        return;
    }
    AcConfig cfg;
    const std::string sweep = toLower(t[1]);
    cfg.sweepType = sweep == "lin" ? AcSweepType::LIN : (sweep == "oct" ? AcSweepType::OCT : AcSweepType::DEC);
    try {
        cfg.nPoints = std::stoi(t[2]);
        cfg.fstart  = parseSpiceNumber(t[3]);
        cfg.fstop   = parseSpiceNumber(t[4]);
    } catch (const std::exception& e) {
        std::cerr << "Line " << st.lineNo << ": cannot parse .AC arguments: " << e.what()
                  << " in '" << st.raw << "'\n";
        return;
    }
    cfg.enabled = true;
    sim.ac = cfg;
}

// .HB f0 nharm
void NetlistParser::hbCard(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 3) {
        std::cerr << "Line " << st.lineNo << ": invalid .hb syntax: " << st.raw << "\n";
        return;
    }
    HbConfig cfg;
    try {
        cfg.f0 = parseSpiceNumber(t[1]);
        cfg.nHarm = std::stoi(t[2]);
    } catch (const std::exception& e) {
        std::cerr << "Line " << st.lineNo << ": cannot parse .hb arguments: " << e.what()
                  << " in '" << st.raw << "'\n";
        return;
    }
    cfg.enabled = true;
    sim.hb = cfg;
}

// V(n) | V(n1,n2) | I(elem)
ProbeSpec NetlistParser::probeFromToken(const std::string& token)
{
    ProbeSpec p;
    p.expr = token;
    if (token.empty()) return p;

    int open = -1, close = -1;
    parenSpan(token, open, close);
    const bool hasArg = open >= 0 && close > open + 1;
    const std::string inside = hasArg
        ? token.substr(static_cast<std::size_t>(open + 1), static_cast<std::size_t>(close - open - 1))
        : std::string();

    const char head = upperHead(token);
    if (head == 'V') {
        p.kind = ProbeKind::NodeVoltage;
        if (hasArg) {
            const std::size_t comma = inside.find(',');
            if (comma == std::string::npos) {
                p.node1 = rtrim(ltrim(inside));
            } else {
                p.node1 = rtrim(ltrim(inside.substr(0, comma)));
                p.node2 = rtrim(ltrim(inside.substr(comma + 1)));
                p.kind = ProbeKind::DiffVoltage;
            }
        }
    } else if (head == 'I') {
        p.kind = ProbeKind::BranchCurrent;
        if (hasArg) p.eleName = rtrim(ltrim(inside));
    }
    return p;
}

// .PRINT <analysis> probe...
void NetlistParser::printCard(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 3) {
        std::cerr << "Line " << st.lineNo << ": invalid .PRINT: " << st.raw << "\n";
        return;
    }
    PrintCommand pc;
    pc.analysis = analysisFromToken(t[1]);
    if (pc.analysis == AnalysisType::NONE) {
        std::cerr << "Line " << st.lineNo << ": unknown analysis type in .PRINT: " << t[1]
                  << " in '" << st.raw << "'\n";
        return;
    }
    for (std::size_t i = 2; i < t.size(); ++i) pc.probes.push_back(probeFromToken(t[i]));
    sim.printCommands.push_back(std::move(pc));
}

// .PLOTNV node...   (node voltages, not tied to an analysis)
void NetlistParser::plotNvCard(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 2) {
        std::cerr << "Line " << st.lineNo << ": invalid .PLOTNV: " << st.raw << "\n";
        return;
    }
    PrintCommand pc;
    pc.analysis = AnalysisType::NONE;
    for (std::size_t i = 1; i < t.size(); ++i)
        if (!t[i].empty()) pc.probes.push_back(probeFromToken("V(" + t[i] + ")"));
    if (!pc.probes.empty()) sim.printCommands.push_back(std::move(pc));
}

// .PLOTNC elem | elem(port) ...   (branch currents)
void NetlistParser::plotNcCard(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 2) {
        std::cerr << "Line " << st.lineNo << ": invalid .PLOTNC: " << st.raw << "\n";
        return;
    }
    PrintCommand pc;
    pc.analysis = AnalysisType::NONE;
    for (std::size_t i = 1; i < t.size(); ++i) {
        const std::string& tok = t[i];
        if (tok.empty()) continue;
        ProbeSpec p;
        p.kind = ProbeKind::BranchCurrent;
        p.expr = tok;
        int open = -1, close = -1;
        parenSpan(tok, open, close);
        if (open < 0) {
            p.eleName = tok;
        } else {
            p.eleName = rtrim(ltrim(tok.substr(0, static_cast<std::size_t>(open))));
            p.elePort = rtrim(ltrim(tok.substr(static_cast<std::size_t>(open + 1),
                                               static_cast<std::size_t>(close - open - 1))));
        }
        pc.probes.push_back(std::move(p));
    }
    if (!pc.probes.empty()) sim.printCommands.push_back(std::move(pc));
}

// .MODEL id {VT|MU|COX|LAMBDA|CJ0|CJO value}...   sign of VT selects PMOS
void NetlistParser::modelCard(const Statement& st)
{
    const auto& t = st.tokens;
    if (t.size() < 4) {
        std::cerr << "Line " << st.lineNo << ": invalid .MODEL: " << st.raw << "\n";
        return;
    }
    MosModel m;
    m.name = t[1];
    for (std::size_t i = 2; i + 1 < t.size(); i += 2) {
        const std::string key = toLower(t[i]);
        double val = 0.0;
        try {
            val = parseSpiceNumber(t[i + 1]);
        } catch (const std::exception& e) {
            std::cerr << "Line " << st.lineNo << ": cannot parse .MODEL param " << t[i] << " = "
                      << t[i + 1] << " : " << e.what() << "\n";
            return;
        }
        if      (key == "vt")     m.VT = val;
        else if (key == "mu")     m.MU = val;
        else if (key == "cox")    m.COX = val;
        else if (key == "lambda") m.LAMBDA = val;
        else if (key == "cj0" || key == "cjo") m.CJO = val;
    }
    m.isP = m.VT < 0.0;
    if (m.isP) m.VT = -m.VT;
    ckt.addMosModel(m);
}
