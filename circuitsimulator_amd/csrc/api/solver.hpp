// solver.hpp -- dense linear solves G x = I.
//
// Same names as the reference's include/solver.hpp (namespace Solver,
// luDecompose :30-80, solveLinearSystemLU :83-131, enum LinearSolver :18-21).
// Both run on the GPU (k_lu_solve / k_lu_factor, engine/kernels_general.hip):
// Doolittle LU with partial pivoting, FIRST row attaining the column maximum,
// failure below 1e-15.  solveLinearSystemGaussSeidel (:139-204; unreachable from the
// reference's main(), kept as public API) runs on the GPU as well (k_gs_solve).
#pragma once

#include <vector>

#include "linalg.hpp"

namespace Solver {

using Eigen::MatrixXd;
using Eigen::VectorXd;

enum class LinearSolver { DirectLU, GaussSeidel };

// P*A = L*U, L (unit diagonal) and U packed in LU, perm[i] = original row at
// position i.  false on an empty / non-square matrix or a tiny pivot.
bool luDecompose(const MatrixXd& A, MatrixXd& LU, std::vector<int>& perm);

// solves A x = b; the zero vector if the decomposition fails
VectorXd solveLinearSystemLU(const MatrixXd& A, const VectorXd& b);

// Gauss-Seidel sweeps from x0 (warm start); a diagonal below 1e-12 is replaced by +-1e-12; returns
// whatever the last sweep left (no convergence guarantee, possibly non-finite).  A wrong-sized x0
// starts from zero, mismatched A/b give the zero vector (include/solver.hpp:139-204).
VectorXd solveLinearSystemGaussSeidel(const MatrixXd& A, const VectorXd& b, const VectorXd& x0,
                                      int maxIters = 1000, double tol = 1e-10);
VectorXd solveLinearSystemGaussSeidel(const MatrixXd& A, const VectorXd& b, int maxIters = 1000, double tol = 1e-10);

} // namespace Solver
