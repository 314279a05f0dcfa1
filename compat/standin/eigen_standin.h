// eigen_standin.h -- TEST DOUBLE, not Eigen.  The smallest column-major dynamic matrix that lets
// compat/kalmanfilter.h keep the reference's signatures (odometry/kalmanfilter.h:31) in a tree
// without Eigen.  Where <Eigen/Dense> exists, compat/kalmanfilter.h uses the real thing instead.
#pragma once
#include <cstddef>
#include <vector>

namespace Eigen {
class MatrixXd {
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(std::ptrdiff_t rows, std::ptrdiff_t cols) : r_(rows), c_(cols), d_((size_t)rows * cols, 0.0) {}
    double &operator()(std::ptrdiff_t i, std::ptrdiff_t j) { return d_[(size_t)j * r_ + i]; }
    double operator()(std::ptrdiff_t i, std::ptrdiff_t j) const { return d_[(size_t)j * r_ + i]; }
    std::ptrdiff_t rows() const { return r_; }
    std::ptrdiff_t cols() const { return c_; }
    std::ptrdiff_t size() const { return r_ * c_; }
    const double *data() const { return d_.data(); }  // column-major, like Eigen
    double *data() { return d_.data(); }

private:
    std::ptrdiff_t r_, c_;
    std::vector<double> d_;
};
}  // namespace Eigen
