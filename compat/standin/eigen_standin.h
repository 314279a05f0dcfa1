// eigen_standin.h -- TEST DOUBLE, not Eigen.  The smallest column-major dynamic matrix that lets
// compat/kalmanfilter.h keep the reference's signatures (odometry/kalmanfilter.h:31) in a tree
// without Eigen.  Where <Eigen/Dense> exists, compat/kalmanfilter.h uses the real thing instead.
// The comma initialiser, the product and transpose() are what slam.cpp:152-167 writes when it builds a measurement and its R
// (z_chunk << ..., R_chunk = G * R * G.transpose()); they exist so that the compiler's front end can check that file against compat/.
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

    class CommaInit {  // `m << a, b, c, d;` fills row by row, as Eigen's does
    public:
        CommaInit(MatrixXd &m, double first) : m_(m), k_(0) { put(first); }
        CommaInit &operator,(double v) { put(v); return *this; }

    private:
        void put(double v) {
            if (k_ < m_.size()) m_(k_ / m_.cols(), k_ % m_.cols()) = v;
            k_++;
        }
        MatrixXd &m_;
        std::ptrdiff_t k_;
    };
    CommaInit operator<<(double first) { return CommaInit(*this, first); }
    MatrixXd transpose() const {
        MatrixXd t(c_, r_);
        for (std::ptrdiff_t i = 0; i < r_; i++)
            for (std::ptrdiff_t j = 0; j < c_; j++) t(j, i) = (*this)(i, j);
        return t;
    }
    MatrixXd operator*(const MatrixXd &o) const {  // (inner dimensions are the caller's business, as with Eigen in a release build)
        MatrixXd p(r_, o.c_);
        for (std::ptrdiff_t i = 0; i < r_; i++)
            for (std::ptrdiff_t j = 0; j < o.c_; j++) {
                double s = 0.0;
                for (std::ptrdiff_t k = 0; k < c_; k++) s += (*this)(i, k) * o(k, j);
                p(i, j) = s;
            }
        return p;
    }

private:
    std::ptrdiff_t r_, c_;
    std::vector<double> d_;
};
}  // namespace Eigen
