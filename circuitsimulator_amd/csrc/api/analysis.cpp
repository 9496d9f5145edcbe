// analysis.cpp -- C++ shims over the C-ABI: the reference's scalar entry points
// are the batch-of-one case of the GPU engine.
#include "analysis.hpp"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "dcanalysis.hpp"
#include "solver.hpp"
#include "tanalisis.hpp"
#include "../engine/netlist_internal.hpp"

namespace csim {

namespace {
[[noreturn]] void fail(const char* what)
{
    throw std::runtime_error(std::string(what) + ": " + csim_last_error());
}
} // namespace

BatchEngine::BatchEngine(const Circuit& ckt, int device)
{
    ir_ = flatten(ckt);
    ir_.view();
    // wrap the already-built circuit in a netlist handle for the C-ABI
    nl_ = new csim_netlist();
    nl_->ckt = ckt;
    nl_->cir = ir_;
    nl_->cir.view();
    if (csim_engine_create(nl_, device, &eng_) != CSIM_OK) {
        delete nl_;
        nl_ = nullptr;
        fail("csim_engine_create");
    }
}

BatchEngine::~BatchEngine()
{
    if (eng_) csim_engine_destroy(eng_);
    delete nl_;
}

std::vector<double> BatchEngine::monteCarloParams(uint64_t seed, double sigma, int64_t bFirst, int B) const
{
    const int P = numParams();
    std::vector<double> slotMajor(static_cast<std::size_t>(P) * static_cast<std::size_t>(B));
    if (csim_mc_params_host(nl_, seed, sigma, bFirst, B, slotMajor.data()) != CSIM_OK) fail("csim_mc_params_host");
    std::vector<double> out(slotMajor.size());
    for (int p = 0; p < P; ++p)
        for (int b = 0; b < B; ++b)
            out[static_cast<std::size_t>(b) * P + p] = slotMajor[static_cast<std::size_t>(p) * B + b];
    return out;
}

BatchDcResult BatchEngine::dc(const std::vector<double>& params, int B)
{
    BatchDcResult r;
    const int N = numUnknowns();
    r.x.assign(static_cast<std::size_t>(B) * N, 0.0);
    r.iters.assign(static_cast<std::size_t>(B), 0);
    r.status.assign(static_cast<std::size_t>(B), 0);
    if (csim_dc_batch(eng_, params.empty() ? nullptr : params.data(), B, r.x.data(), r.iters.data(),
                      r.status.data()) != CSIM_OK)
        fail("csim_dc_batch");
    return r;
}

BatchTranResult BatchEngine::tran(const std::vector<double>& params, int B, double tstep, double tstop,
                                  double tstart, const std::vector<int32_t>& probeEq, int outStride)
{
    BatchTranResult r;
    const int N = numUnknowns();
    const int np = static_cast<int>(probeEq.size());
    r.rows = np ? csim_tran_num_rows(tstep, tstop, tstart, outStride) : 0;
    if (np && r.rows < 0) { csim::setError("invalid .TRAN numbers"); fail("csim_tran_num_rows"); }
    r.wave.assign(static_cast<std::size_t>(B) * static_cast<std::size_t>(r.rows) * np, 0.0);
    r.xFinal.assign(static_cast<std::size_t>(B) * N, 0.0);
    r.iters.assign(static_cast<std::size_t>(B), 0);
    r.status.assign(static_cast<std::size_t>(B), 0);
    if (csim_tran_batch(eng_, params.empty() ? nullptr : params.data(), B, tstep, tstop, tstart,
                        np ? probeEq.data() : nullptr, np, outStride, np ? r.wave.data() : nullptr,
                        r.xFinal.data(), r.iters.data(), r.status.data()) != CSIM_OK)
        fail("csim_tran_batch");
    return r;
}

void BatchEngine::writeCsv(const std::vector<double>& params, int B, int instance, const SimulationConfig& sim,
                           const std::string& path, const std::vector<int32_t>& probeEq)
{
    std::vector<int32_t> cols = probeEq;
    if (cols.empty()) {
        // node-voltage probes of .PLOTNV / .PRINT cards, in card order (src/parser.cpp:630-723)
        for (const PrintCommand& pc : sim.printCommands)
            for (const ProbeSpec& ps : pc.probes) {
                if (ps.kind != ProbeKind::NodeVoltage) continue;
                const int eq = csim_netlist_node_eq(nl_, ps.node1.c_str());
                if (eq >= 0 && std::find(cols.begin(), cols.end(), eq) == cols.end()) cols.push_back(eq);
            }
    }
    if (cols.empty())
        for (int i = 0; i < numUnknowns(); ++i) cols.push_back(i);
    if (csim_tran_write_csv(eng_, params.empty() ? nullptr : params.data(), B, instance, sim.tran.tstep, sim.tran.tstop,
                            sim.tran.tstart, cols.data(), static_cast<int32_t>(cols.size()), path.c_str()) != CSIM_OK)
        fail("csim_tran_write_csv");
}

} // namespace csim

// ------------------------------------------------------------ dcanalysis.hpp

ConvController::ConvController()
    : alphaMin(0.1), alphaMax(0.5), gminHighBase(1e-6), gminLowBase(3.35e-7), gminAbsMax(1e-4),
      fastConvRatio(0.7), slowConvRatio(1.05) {}

double ConvController::baseGmin(double rampScale) const
{
    const double s = rampScale < 0.0 ? 0.0 : (rampScale > 1.0 ? 1.0 : rampScale);
    return gminHighBase * (1.0 - s) + gminLowBase * s;
}

Eigen::VectorXd dcSolveLU(const Circuit& ckt)
{
    const int N = ckt.numUnknowns();
    if (N == 0) {                                           // dcanalysis.cpp:50-53, 99-102
        std::cerr << "DC solve (LU): no unknowns.\n";
        return Eigen::VectorXd::Zero(0);
    }
    csim::BatchEngine eng(ckt, 0);
    const csim::BatchDcResult r = eng.dc({}, 1);
    if (r.status[0] & CSIM_ST_DC_NONCONV)
        std::cerr << "WARNING: Newton (LU) did not converge within the iteration cap at one or more ramp steps\n";
    if (r.status[0] & CSIM_ST_LU_TINY_PIVOT) std::cerr << "LU solve: decomposition failed.\n";
    Eigen::VectorXd x(N);
    for (int i = 0; i < N; ++i) x(i) = r.x[static_cast<std::size_t>(i)];
    return x;
}

Eigen::VectorXd dcSolve(const Circuit& ckt) { return dcSolveLU(ckt); }

Eigen::VectorXd dcSolveGaussSeidel(const Circuit& ckt)
{
    const int N = ckt.numUnknowns();
    if (N == 0) {                                           // dcanalysis.cpp:75-78, 170-173
        std::cerr << "DC solve (GS): no unknowns.\n";
        return Eigen::VectorXd::Zero(0);
    }
    csim::BatchEngine eng(ckt, 0);
    std::vector<double> x(static_cast<std::size_t>(N), 0.0);
    int32_t iters = 0;
    uint32_t status = 0;
    if (csim_dc_gs_batch(eng.handle(), nullptr, 1, x.data(), &iters, &status) != CSIM_OK)
        throw std::runtime_error(std::string("csim_dc_gs_batch: ") + csim_last_error());
    if (status & CSIM_ST_DC_NONFINITE) std::cerr << "WARNING: GS produced non-finite x at one or more Newton passes (gmin was raised)\n";
    if (status & CSIM_ST_DC_NONCONV) std::cerr << "WARNING: Newton (GS) did not converge within the iteration cap at one or more ramp steps\n";
    Eigen::VectorXd out(N);
    for (int i = 0; i < N; ++i) out(i) = x[static_cast<std::size_t>(i)];
    return out;
}

// host arithmetic, for callers of the reference's header only (dcanalysis.hpp): src/dcanalysis.cpp:268-307
ConvStatus ConvController::update(const Eigen::VectorXd& x, const Eigen::VectorXd& xRaw, double prevErr, int iter,
                                  double /*alphaCurrent*/, double gminCurrent, double rampScale, double tol) const
{
    ConvStatus st;
    double alpha = 0.35 < alphaMin ? alphaMin : (0.35 > alphaMax ? alphaMax : 0.35);       // :274
    const long n = x.size();
    Eigen::VectorXd xNew(n);
    double ss = 0.0;
    for (long i = 0; i < n; ++i) xNew(i) = x(i) + alpha * (xRaw(i) - x(i));
    for (long i = 0; i < n; ++i) { const double d = xNew(i) - x(i); ss += d * d; }
    const double err = std::sqrt(ss);
    const double gminBase = baseGmin(rampScale);
    double gminNext = gminBase;
    if (!(iter == 0 || !std::isfinite(prevErr))) {
        if (err > prevErr * slowConvRatio) { alpha = std::fmax(alpha * 0.7, alphaMin); gminNext = std::fmin(gminCurrent * 2.0, gminAbsMax); }
        else if (err < prevErr * fastConvRatio) { alpha = std::fmin(alpha * 1.1, alphaMax); gminNext = 0.5 * gminCurrent + 0.5 * gminBase; }
        else gminNext = 0.7 * gminCurrent + 0.3 * gminBase;
    }
    st.xNext = xNew;
    st.alphaNext = alpha;
    st.gminNext = gminNext;
    st.error = err;
    st.converged = err < tol;
    return st;
}

// ------------------------------------------------------------- tanalisis.hpp

Eigen::VectorXd computeDcOperatingPoint(const Circuit& ckt) { return dcSolve(ckt); }

void runTransientAnalysisBackwardEuler(const Circuit& ckt, const SimulationConfig& sim, const std::string& outFile)
{
    const TranConfig& cfg = sim.tran;
    if (!cfg.enabled) { std::cerr << "Transient analysis is not enabled (.TRAN missing).\n"; return; }
    if (cfg.tstep <= 0.0 || cfg.tstop <= 0.0) { std::cerr << "Invalid .TRAN card: tstep and tstop must be > 0.\n"; return; }
    const int N = ckt.numUnknowns();
    if (N <= 0) { std::cerr << "Transient: circuit has no unknowns.\n"; return; }

    csim::BatchEngine eng(ckt, 0);
    std::ofstream ofs(outFile);
    if (!ofs) { std::cerr << "Cannot open transient output file '" << outFile << "'.\n"; return; }

    std::vector<int32_t> probes(static_cast<std::size_t>(N));
    for (int i = 0; i < N; ++i) probes[static_cast<std::size_t>(i)] = i;
    const int64_t nSteps = csim_tran_num_steps(cfg.tstep, cfg.tstop);
    std::cout << "[TRAN] tstep=" << std::scientific << cfg.tstep << ", tstop=" << cfg.tstop
              << ", tstart=" << cfg.tstart << "\n[TRAN] total steps = " << nSteps << "\n";

    const csim::BatchTranResult r = eng.tran({}, 1, cfg.tstep, cfg.tstop, cfg.tstart, probes, 1);
    if (r.status[0] & CSIM_ST_TRAN_NONFINITE) throw std::runtime_error("Transient: LU produced NaN/Inf.");
    if (r.status[0] & CSIM_ST_TRAN_NONCONV)
        std::cerr << "WARNING: transient Newton did not converge within the iteration cap at one or more steps\n";

    // CSV in the reference's format (tanalisis.cpp:189-231)
    const csim_ir* ir = eng.ir().view();
    ofs << "time";
    for (int eq = 0; eq < N; ++eq)
        ofs << (eq < ir->n_node_eq ? ",V(" : ",I(") << eng.ir().eqNames[static_cast<std::size_t>(eq)] << ")";
    ofs << "\n";
    const int64_t allRows = nSteps + 1;
    const int64_t first = allRows - r.rows;                 // rows with t < tstart were dropped
    char buf[32];
    for (int64_t k = 0; k < r.rows; ++k) {
        const int64_t row = first + k;
        const double t = row == 0 ? 0.0 : static_cast<double>(static_cast<int>(row)) * cfg.tstep;
        std::snprintf(buf, sizeof buf, "%.9e", t);
        ofs << buf;
        for (int i = 0; i < N; ++i) {
            std::snprintf(buf, sizeof buf, "%.9e", r.wave[static_cast<std::size_t>(k) * N + i]);
            ofs << "," << buf;
        }
        ofs << "\n";
    }
    std::cout << "Transient analysis (Backward Euler) finished. Results written to '" << outFile << "'.\n";
}

// ---------------------------------------------------------------- solver.hpp

namespace Solver {

bool luDecompose(const MatrixXd& A, MatrixXd& LU, std::vector<int>& perm)
{
    const int n = static_cast<int>(A.rows());
    if (n == 0) return false;
    if (A.cols() != n) { std::cerr << "LU: matrix is not square.\n"; return false; }
    std::vector<double> a(static_cast<std::size_t>(n) * n), lu(a.size());
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) a[static_cast<std::size_t>(i) * n + j] = A(i, j);
    std::vector<int32_t> p(static_cast<std::size_t>(n));
    uint32_t flags = 0;
    if (csim_lu_decompose_batch(0, n, 1, a.data(), lu.data(), p.data(), &flags) != CSIM_OK)
        throw std::runtime_error(std::string("csim_lu_decompose_batch: ") + csim_last_error());
    if (flags & CSIM_ST_LU_TINY_PIVOT) { std::cerr << "LU: zero (or tiny) pivot.\n"; return false; }
    LU = MatrixXd(n, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) LU(i, j) = lu[static_cast<std::size_t>(i) * n + j];
    perm.assign(p.begin(), p.end());
    return true;
}

VectorXd solveLinearSystemLU(const MatrixXd& A, const VectorXd& b)
{
    const int n = static_cast<int>(A.rows());
    VectorXd x = VectorXd::Zero(n);
    if (n == 0) return x;
    if (A.cols() != n || b.size() != n) { std::cerr << "LU solve: dimension mismatch.\n"; return x; }
    std::vector<double> a(static_cast<std::size_t>(n) * n), rhs(static_cast<std::size_t>(n)), sol(rhs.size());
    for (int i = 0; i < n; ++i) {
        rhs[static_cast<std::size_t>(i)] = b(i);
        for (int j = 0; j < n; ++j) a[static_cast<std::size_t>(i) * n + j] = A(i, j);
    }
    uint32_t flags = 0;
    if (csim_lu_solve_batch(0, n, 1, a.data(), rhs.data(), sol.data(), &flags) != CSIM_OK)
        throw std::runtime_error(std::string("csim_lu_solve_batch: ") + csim_last_error());
    if (flags & CSIM_ST_LU_TINY_PIVOT) std::cerr << "LU solve: decomposition failed.\n";
    for (int i = 0; i < n; ++i) x(i) = sol[static_cast<std::size_t>(i)];
    return x;
}

VectorXd solveLinearSystemGaussSeidel(const MatrixXd& A, const VectorXd& b, const VectorXd& x0, int maxIters, double tol)
{
    const int n = static_cast<int>(A.rows());
    if (n == 0) return x0;                                                   // solver.hpp:146
    if (A.cols() != n || b.size() != n) { std::cerr << "Gauss-Seidel: dimension mismatch.\n"; return VectorXd::Zero(n); }
    std::vector<double> a(static_cast<std::size_t>(n) * n), rhs(static_cast<std::size_t>(n)), start(rhs.size(), 0.0), sol(rhs.size());
    for (int i = 0; i < n; ++i) {
        rhs[static_cast<std::size_t>(i)] = b(i);
        if (x0.size() == n) start[static_cast<std::size_t>(i)] = x0(i);      // a wrong-sized x0 starts from zero (:154-157)
        for (int j = 0; j < n; ++j) a[static_cast<std::size_t>(i) * n + j] = A(i, j);
    }
    if (csim_gs_solve_batch(0, n, 1, a.data(), rhs.data(), start.data(), maxIters, tol, sol.data(), nullptr) != CSIM_OK)
        throw std::runtime_error(std::string("csim_gs_solve_batch: ") + csim_last_error());
    VectorXd x(n);
    for (int i = 0; i < n; ++i) x(i) = sol[static_cast<std::size_t>(i)];
    return x;
}

VectorXd solveLinearSystemGaussSeidel(const MatrixXd& A, const VectorXd& b, int maxIters, double tol)
{
    return solveLinearSystemGaussSeidel(A, b, VectorXd::Zero(b.size()), maxIters, tol);       // solver.hpp:197-204
}

} // namespace Solver
