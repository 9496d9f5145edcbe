// solver.hpp -- dense linear solves G x = I.
//
// Same names as the reference's include/solver.hpp (namespace Solver,
// luDecompose :30-80, solveLinearSystemLU :83-131, enum LinearSolver :18-21).
// Both run on the GPU (k_lu_solve / k_lu_factor, engine/kernels_general.hip):
// Doolittle LU with partial pivoting, FIRST row attaining the column maximum,
// failure below 1e-15.  The Gauss-Seidel variant of the reference is dead code
// upstream (unreachable from dcSolve) and is not provided.
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

} // namespace Solver
