// mat.hpp -- the `Mat` of the reference (env/env.hpp:14: Eigen::Matrix<float, Dynamic, Dynamic, RowMajor>).
//
// With Eigen installed the typedef is exactly the reference's, so environments written against the reference
// headers (env_mock.hpp, hexapod_env.hpp) compile against host/env/env.hpp unchanged.  This image has no Eigen, so a
// small row-major float matrix stands in; the host layer only uses the subset both types share:
//   Mat(r, c), Mat::Zero(r, c), Mat::Ones(r, c), rows(), cols(), data(), operator()(i, j), scalar * Mat
// and raw row-major loops over data().
#pragma once

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && !defined(PPO_FORCE_MAT_SHIM)
#include <Eigen/Dense>
#define PPO_HAVE_EIGEN 1
typedef Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> Mat;
#endif
#endif

#ifndef PPO_HAVE_EIGEN
#include <cassert>
#include <cstddef>
#include <vector>

class Mat {
public:
    Mat() : r_(0), c_(0) {}
    Mat(long rows, long cols) : r_(rows), c_(cols), v_((size_t)rows * cols) {}
    static Mat Zero(long rows, long cols) { Mat m(rows, cols); return m; }
    static Mat Ones(long rows, long cols) { return Constant(rows, cols, 1.0f); }
    static Mat Constant(long rows, long cols, float x) { Mat m(rows, cols); for (auto& e : m.v_) e = x; return m; }
    long rows() const { return r_; }
    long cols() const { return c_; }
    long size() const { return r_ * c_; }
    float* data() { return v_.data(); }
    const float* data() const { return v_.data(); }
    float& operator()(long i, long j) { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return v_[(size_t)i * c_ + j]; }
    float operator()(long i, long j) const { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return v_[(size_t)i * c_ + j]; }
    float squaredNorm() const { double s = 0; for (float e : v_) s += (double)e * e; return (float)s; }
    float sum() const { double s = 0; for (float e : v_) s += e; return (float)s; }
    Mat operator-(const Mat& o) const { assert(r_ == o.r_ && c_ == o.c_); Mat m(r_, c_); for (size_t i = 0; i < v_.size(); ++i) m.v_[i] = v_[i] - o.v_[i]; return m; }
    Mat operator+(const Mat& o) const { assert(r_ == o.r_ && c_ == o.c_); Mat m(r_, c_); for (size_t i = 0; i < v_.size(); ++i) m.v_[i] = v_[i] + o.v_[i]; return m; }
    Mat col(long j) const { Mat m(r_, 1); for (long i = 0; i < r_; ++i) m.v_[i] = v_[(size_t)i * c_ + j]; return m; }
    Mat row(long i) const { Mat m(1, c_); for (long j = 0; j < c_; ++j) m.v_[j] = v_[(size_t)i * c_ + j]; return m; }
private:
    long r_, c_;
    std::vector<float> v_;
};
inline Mat operator*(float s, const Mat& a) { Mat m(a.rows(), a.cols()); for (long i = 0; i < a.size(); ++i) m.data()[i] = s * a.data()[i]; return m; }
inline Mat operator*(double s, const Mat& a) { return (float)s * a; }
#endif

#include <cstring>
// row helpers that work on either Mat type
inline void mat_set_row(Mat& dst, long i, const float* src) { std::memcpy(dst.data() + (size_t)i * dst.cols(), src, sizeof(float) * (size_t)dst.cols()); }
inline const float* mat_row_ptr(const Mat& m, long i) { return m.data() + (size_t)i * m.cols(); }
