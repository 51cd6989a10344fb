// ORACLE / TEST INFRASTRUCTURE ONLY: just enough of the ceres names for include/Initializer/initial_sfm.h to PARSE when larvio.cpp is
// compiled in place (the structure-from-motion itself is not compiled: it needs Ceres and OpenCV proper)
#pragma once
namespace ceres {
struct CostFunction { virtual ~CostFunction() {} };
template <typename F, int... N> struct AutoDiffCostFunction : CostFunction { explicit AutoDiffCostFunction(F* f) : f_(f) {} ~AutoDiffCostFunction() { delete f_; } F* f_; };
template <typename T> inline void QuaternionRotatePoint(const T*, const T*, T*) {}
}
