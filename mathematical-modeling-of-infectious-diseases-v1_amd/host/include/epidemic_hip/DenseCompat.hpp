// DenseCompat.hpp -- the reference's interfaces speak Eigen::VectorXd / Eigen::MatrixXd
// (include/sir_age_structured/interfaces/IObjectiveFunction.hpp:24).  Inside the reference tree the
// real Eigen is used; where Eigen is not installed (this build image) a minimal storage-only stand-in
// with the same spelling keeps the adapter sources identical.  Only storage and element access are
// needed on this path: the reference uses no Eigen numerics inside the RHS or the likelihood.
#pragma once
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#else
#include <cstddef>
#include <vector>
namespace Eigen {
using Index = std::ptrdiff_t;
class VectorXd {
public:
    VectorXd() = default;
    explicit VectorXd(Index n) : v_(static_cast<size_t>(n), 0.0) {}
    static VectorXd Zero(Index n) { return VectorXd(n); }
    static VectorXd Constant(Index n, double c) { VectorXd r(n); for (auto& x : r.v_) x = c; return r; }
    Index size() const { return static_cast<Index>(v_.size()); }
    void resize(Index n) { v_.resize(static_cast<size_t>(n)); }
    double& operator()(Index i) { return v_[static_cast<size_t>(i)]; }
    double operator()(Index i) const { return v_[static_cast<size_t>(i)]; }
    double& operator[](Index i) { return v_[static_cast<size_t>(i)]; }
    double operator[](Index i) const { return v_[static_cast<size_t>(i)]; }
    double* data() { return v_.data(); }
    const double* data() const { return v_.data(); }
    double sum() const { double s = 0; for (double x : v_) s += x; return s; }
private:
    std::vector<double> v_;
};
// column-major like Eigen's default
class MatrixXd {
public:
    MatrixXd() = default;
    MatrixXd(Index r, Index c) : r_(r), c_(c), v_(static_cast<size_t>(r * c), 0.0) {}
    static MatrixXd Zero(Index r, Index c) { return MatrixXd(r, c); }
    static MatrixXd Identity(Index r, Index c) { MatrixXd m(r, c); for (Index i = 0; i < (r < c ? r : c); ++i) m(i, i) = 1.0; return m; }
    Index rows() const { return r_; }
    Index cols() const { return c_; }
    void resize(Index r, Index c) { r_ = r; c_ = c; v_.assign(static_cast<size_t>(r * c), 0.0); }
    double& operator()(Index i, Index j) { return v_[static_cast<size_t>(j * r_ + i)]; }
    double operator()(Index i, Index j) const { return v_[static_cast<size_t>(j * r_ + i)]; }
    double* data() { return v_.data(); }
    const double* data() const { return v_.data(); }
private:
    Index r_ = 0, c_ = 0;
    std::vector<double> v_;
};
}  // namespace Eigen
#endif
