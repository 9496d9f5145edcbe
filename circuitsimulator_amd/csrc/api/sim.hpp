// sim.hpp -- analysis context, source specifications and analysis cards.
//
// Public names mirror the reference's include/sim.hpp so that callers written
// against it keep compiling: AnalysisType / WaveformType / ProbeKind enums
// (:10-36), AnalysisContext (:38-44), PulseSpec/SinSpec/PwlSpec (:46-67),
// TranWaveform (:69-144), SourceSpec (:146-163), DCSweepConfig, TranConfig,
// AcConfig, HbConfig, ProbeSpec, PrintCommand, SimulationConfig (:165-226).
//
// The numeric evaluation of sources on the hot path happens on the GPU
// (engine/device_common.hpp: source_value()); the host evaluators below exist
// for API compatibility and for printing, and state the same formulas.
#pragma once

#include <cmath>
#include <cstddef>
#include <string>
#include <vector>

#include "utils.hpp"

enum class AnalysisType { NONE, OP, DC, AC, TRAN, HB };
enum class AcSweepType  { LIN, DEC, OCT };
enum class WaveformType { NONE, PULSE, SIN, PWL };
enum class ProbeKind    { NodeVoltage, DiffVoltage, BranchCurrent };

// what every stamp is told about the analysis being run
struct AnalysisContext {
    AnalysisType type = AnalysisType::OP;
    double sourceScale = 1.0;   // DC source ramp factor
    double time        = 0.0;   // TRAN
    double omega       = 0.0;   // AC (unused)
};

struct PulseSpec {
    double v1 = 0.0, v2 = 0.0;          // initial / pulsed level
    double td = 0.0, tr = 0.0, tf = 0.0;
    double ton = 0.0;
    double per = 0.0;                   // 0: single shot
};

struct SinSpec {
    double v0 = 0.0;     // offset
    double va = 0.0;     // amplitude
    double freq = 0.0;   // Hz
    double td = 0.0;     // delay in SECONDS (4th SIN argument)
    double phi = 0.0;    // phase in RADIANS (5th SIN argument)
};

struct PwlSpec {
    std::vector<double> t;
    std::vector<double> v;
};

struct TranWaveform {
    static constexpr double kPi = 3.14159265358979323846;

    WaveformType type = WaveformType::NONE;
    PulseSpec pulse;
    SinSpec   sine;
    PwlSpec   pwl;

    // value added on top of SourceSpec::dcValue at time t
    double eval(double t) const
    {
        if (type == WaveformType::SIN) {
            if (t < sine.td) return sine.v0;
            const double tau = t - sine.td;
            const double w = 2.0 * kPi * sine.freq;
            return sine.v0 + sine.va * std::sin(w * tau + sine.phi);
        }
        if (type == WaveformType::PULSE) return evalPulse(t);
        if (type == WaveformType::PWL)   return evalPwl(t);
        return 0.0;
    }

private:
    double evalPulse(double t) const
    {
        const PulseSpec& p = pulse;
        if (p.per <= 0.0) {
            const double tau = t - p.td;
            if (tau <= 0.0) return p.v1;
            if (tau < p.tr) return p.v1 + clamp01(tau / p.tr) * (p.v2 - p.v1);
            if (tau < p.tr + p.ton) return p.v2;
            return p.v2 + clamp01((tau - (p.tr + p.ton)) / p.tf) * (p.v1 - p.v2);
        }
        if (t < p.td) return p.v1;
        double tau = std::fmod(t - p.td, p.per);
        if (tau < 0.0) tau += p.per;
        if (tau < p.tr) return p.v1 + (p.v2 - p.v1) * clamp01(tau / p.tr);
        if (tau < p.tr + p.ton) return p.v2;
        if (tau < p.tr + p.ton + p.tf)
            return p.v2 + (p.v1 - p.v2) * clamp01((tau - (p.tr + p.ton)) / p.tf);
        return p.v1;
    }

    double evalPwl(double t) const
    {
        if (pwl.t.empty()) return 0.0;
        if (t <= pwl.t.front()) return pwl.v.front();
        if (t >= pwl.t.back())  return pwl.v.back();
        for (std::size_t i = 0; i + 1 < pwl.t.size(); ++i) {
            if (t > pwl.t[i] && t <= pwl.t[i + 1]) {
                const double k = (t - pwl.t[i]) / (pwl.t[i + 1] - pwl.t[i]);
                return pwl.v[i] + (pwl.v[i + 1] - pwl.v[i]) * k;
            }
        }
        return pwl.v.back();
    }
};

struct SourceSpec {
    double dcValue    = 0.0;
    double acMag      = 0.0;
    double acPhaseDeg = 0.0;
    TranWaveform tran;

    // operating point value: SIN sources contribute their offset v0, the
    // phase is ignored at t = 0 (reference quirk)
    double evalDC(double scale) const
    {
        double base = dcValue;
        if (tran.type == WaveformType::SIN) base += tran.sine.v0;
        return base * scale;
    }

    double evalTran(double t) const { return dcValue + tran.eval(t); }
};

struct DCSweepConfig {
    std::string sourceName;
    double start = 0.0, stop = 0.0, step = 0.0;
};

struct TranConfig {
    bool enabled = false;
    double tstep = 0.0, tstop = 0.0, tstart = 0.0;
};

struct AcConfig {
    bool enabled = false;
    AcSweepType sweepType = AcSweepType::DEC;
    int nPoints = 0;
    double fstart = 0.0, fstop = 0.0;
};

struct HbConfig {
    bool enabled = false;
    double f0 = 0.0;
    int nHarm = 0;
};

struct ProbeSpec {
    ProbeKind kind = ProbeKind::NodeVoltage;
    std::string expr;
    std::string node1, node2;       // V(n1) / V(n1,n2)
    std::string eleName, elePort;   // I(elem) / elem(port)
};

struct PrintCommand {
    AnalysisType analysis = AnalysisType::NONE;
    std::vector<ProbeSpec> probes;
};

class SimulationConfig {
public:
    bool doOp = false;
    std::vector<DCSweepConfig> dcSweeps;
    TranConfig tran;
    AcConfig ac;
    HbConfig hb;
    std::vector<PrintCommand> printCommands;

    bool hasAnyAnalysis() const
    {
        return doOp || !dcSweeps.empty() || tran.enabled || ac.enabled || hb.enabled;
    }
    void ensureDefaultOp() { doOp = !hasAnyAnalysis(); }
};

inline bool matchAnalysis(const PrintCommand& pc, AnalysisType cur)
{
    return pc.analysis == AnalysisType::NONE || pc.analysis == cur;
}
