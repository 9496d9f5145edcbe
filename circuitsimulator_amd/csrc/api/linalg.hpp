// linalg.hpp -- the dense vector/matrix value types used in the public C++
// signatures (computeDcOperatingPoint, Solver::solveLinearSystemLU, ...).
//
// The reference spells these Eigen::VectorXd / Eigen::MatrixXd but uses Eigen
// purely as a container; all arithmetic of the hot path
// runs in the HIP engine here, so the API only needs owning arrays with
// element access.  When a real Eigen is on the include path it is used
// unchanged; otherwise the two names resolve to the small types below, so a
// caller that says Eigen::VectorXd keeps compiling either way.
#pragma once

#if defined(__has_include)
#  if __has_include(<Eigen/Dense>) && !defined(CSIM_NO_EIGEN)
#    define CSIM_HAVE_EIGEN 1
#  endif
#endif

#ifdef CSIM_HAVE_EIGEN
#include <Eigen/Dense>
namespace csim {
using VectorXd = Eigen::VectorXd;
using MatrixXd = Eigen::MatrixXd;
using VectorXcd = Eigen::VectorXcd;
using MatrixXcd = Eigen::MatrixXcd;
}
#else

#include <cmath>
#include <complex>
#include <cstddef>
#include <vector>

namespace csim {

class VectorXd {
    std::vector<double> v_;
public:
    VectorXd() = default;
    explicit VectorXd(long n) : v_(static_cast<std::size_t>(n < 0 ? 0 : n), 0.0) {}
    static VectorXd Zero(long n) { return VectorXd(n); }

    long size() const { return static_cast<long>(v_.size()); }
    double&       operator()(long i)       { return v_[static_cast<std::size_t>(i)]; }
    const double& operator()(long i) const { return v_[static_cast<std::size_t>(i)]; }
    double&       operator[](long i)       { return v_[static_cast<std::size_t>(i)]; }
    const double& operator[](long i) const { return v_[static_cast<std::size_t>(i)]; }
    double*       data()       { return v_.data(); }
    const double* data() const { return v_.data(); }

    void setZero() { for (double& x : v_) x = 0.0; }
    void setZero(long n) { v_.assign(static_cast<std::size_t>(n < 0 ? 0 : n), 0.0); }
    void resize(long n) { v_.resize(static_cast<std::size_t>(n < 0 ? 0 : n)); }

    bool allFinite() const
    {
        for (double x : v_) if (!std::isfinite(x)) return false;
        return true;
    }
};

// row-major; storage order is not observable through this interface
class MatrixXd {
    long r_ = 0, c_ = 0;
    std::vector<double> a_;
public:
    MatrixXd() = default;
    MatrixXd(long r, long c) : r_(r), c_(c), a_(static_cast<std::size_t>(r * c), 0.0) {}
    static MatrixXd Zero(long r, long c) { return MatrixXd(r, c); }

    long rows() const { return r_; }
    long cols() const { return c_; }
    double&       operator()(long i, long j)       { return a_[static_cast<std::size_t>(i * c_ + j)]; }
    const double& operator()(long i, long j) const { return a_[static_cast<std::size_t>(i * c_ + j)]; }
    double*       data()       { return a_.data(); }
    const double* data() const { return a_.data(); }
    void setZero() { for (double& x : a_) x = 0.0; }
};

// complex containers: they appear in Element::stampAC's signature only (AC is a stub upstream)
class VectorXcd {
    std::vector<std::complex<double>> v_;
public:
    VectorXcd() = default;
    explicit VectorXcd(long n) : v_(static_cast<std::size_t>(n < 0 ? 0 : n)) {}
    static VectorXcd Zero(long n) { return VectorXcd(n); }
    long size() const { return static_cast<long>(v_.size()); }
    std::complex<double>&       operator()(long i)       { return v_[static_cast<std::size_t>(i)]; }
    const std::complex<double>& operator()(long i) const { return v_[static_cast<std::size_t>(i)]; }
};
class MatrixXcd {
    long r_ = 0, c_ = 0;
    std::vector<std::complex<double>> a_;
public:
    MatrixXcd() = default;
    MatrixXcd(long r, long c) : r_(r), c_(c), a_(static_cast<std::size_t>(r * c)) {}
    static MatrixXcd Zero(long r, long c) { return MatrixXcd(r, c); }
    long rows() const { return r_; }
    long cols() const { return c_; }
    std::complex<double>&       operator()(long i, long j)       { return a_[static_cast<std::size_t>(i * c_ + j)]; }
    const std::complex<double>& operator()(long i, long j) const { return a_[static_cast<std::size_t>(i * c_ + j)]; }
};

} // namespace csim

namespace Eigen {
using VectorXd = ::csim::VectorXd;
using MatrixXd = ::csim::MatrixXd;
using VectorXcd = ::csim::VectorXcd;
using MatrixXcd = ::csim::MatrixXcd;
}
#endif
