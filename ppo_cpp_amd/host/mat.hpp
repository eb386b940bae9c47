// mat.hpp -- the `Mat` of the reference (env/env.hpp:14: Eigen::Matrix<float, Dynamic, Dynamic, RowMajor>).
//
// With Eigen installed the typedef is exactly the reference's, so environments written against the reference headers
// (env_mock.hpp, hexapod_env.hpp) compile against host/env/env.hpp unchanged.  This image has no Eigen, so a small eager
// row-major float matrix stands in under the SAME name -- Eigen::Matrix<float, Dynamic, Dynamic, RowMajor>, reachable
// through host/shim/Eigen/Dense -- with the part of the Eigen surface that the reference's env-side headers and its
// VecEnv test use (env/env_mock.hpp:44-58, env/vec_env.hpp:43-47,102,118,247-252,
// env/env_normalize.hpp:71-104, common/running_statistics.hpp:38-53,90-101, test/vecenv_test.cpp:16-46):
//   Mat(r,c), Zero/Ones/Constant, rows/cols/size/data, (i,j), row(i)/col(i)/block(i,j,r,c) as l- and r-values,
//   + - (matrix and scalar *), matrix product, unary -, transpose(), transposeInPlace(), squaredNorm(), sum(), mean(),
//   cwiseProduct/cwiseMax/cwiseMin/cwiseSqrt/cwiseInverse/cwiseAbs, colwise().mean()/sum(), rowwise() - rowvector,
//   asDiagonal() as the right factor of a product, Eigen::Map<Mat>, operator<<.
// No expression templates: every operation returns a Mat (`auto x = 2.0 * Mat::Ones(r, c)` therefore holds a value).
#pragma once

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(PPO_FORCE_MAT_SHIM) && !defined(PPO_EIGEN_SHIM_DENSE)
#include <Eigen/Dense>
#define PPO_HAVE_EIGEN 1
typedef Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> Mat;
#endif
#endif

#ifndef PPO_HAVE_EIGEN
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <new>
#include <ostream>
#include <vector>

// element-access bounds checks: on wherever assert() is, unless the build defines PPO_MAT_NO_BOUNDS (the product libraries do: the host loop walks
// 4096-row columns element by element every env step; shape checks elsewhere stay asserts)
#if defined(PPO_MAT_NO_BOUNDS)
#define PPO_MAT_BOUNDS(x) ((void)0)
#else
#define PPO_MAT_BOUNDS(x) assert(x)
#endif

namespace Eigen {

// Storage of the stand-in matrix.  The host loop hands whole batches around BY VALUE (std::vector<Mat> Env::step(const Mat&) is the
// reference's interface): at 4096 environments every env step allocates and frees several buffers of 16 - 300 KB, which glibc serves with
// mmap / munmap and fresh page faults each time (measured: 70 us of a 145 us VecEnv::step).  Buffers of 16 KB .. 64 MB therefore go through
// a small per-thread cache: a freed block is kept (at most 8, at most 128 MB per thread) and handed to the next request of exactly its size.
namespace mat_detail {
struct BlockCache {                    // plain data (no destructor): stays addressable for matrices that die after the thread's guard below
    static constexpr int kSlots = 8;
    static constexpr size_t kMin = 16u << 10, kMax = 64u << 20, kCap = 128u << 20;
    void* p[kSlots];
    size_t n[kSlots];
    size_t held;
    bool dead;                         // the thread is past its guard (static / thread-local destruction): no more caching
    void* take(size_t bytes) {
        if (dead) return nullptr;
        for (int i = 0; i < kSlots; ++i) if (p[i] && n[i] == bytes) { void* q = p[i]; p[i] = nullptr; held -= bytes; return q; }
        return nullptr;
    }
    bool give(void* q, size_t bytes) {
        if (dead || held + bytes > kCap) return false;
        for (int i = 0; i < kSlots; ++i) if (!p[i]) { p[i] = q; n[i] = bytes; held += bytes; return true; }
        return false;
    }
    struct Guard {                     // frees what the thread still holds when it ends
        BlockCache* c;
        ~Guard() { for (int i = 0; i < kSlots; ++i) if (c->p[i]) { ::operator delete(c->p[i]); c->p[i] = nullptr; } c->held = 0; c->dead = true; }
    };
    static BlockCache& mine() {
        static thread_local BlockCache c = {};
        static thread_local Guard g{&c};
        (void)g;
        return c;
    }
};
template <class T>
struct RecyclingAllocator {
    typedef T value_type;
    RecyclingAllocator() = default;
    template <class U> RecyclingAllocator(const RecyclingAllocator<U>&) {}
    T* allocate(size_t count) {
        const size_t bytes = count * sizeof(T);
        if (bytes >= BlockCache::kMin && bytes <= BlockCache::kMax) if (void* q = BlockCache::mine().take(bytes)) return static_cast<T*>(q);
        return static_cast<T*>(::operator new(bytes));
    }
    void deallocate(T* q, size_t count) {
        const size_t bytes = count * sizeof(T);
        if (bytes >= BlockCache::kMin && bytes <= BlockCache::kMax && BlockCache::mine().give(q, bytes)) return;
        ::operator delete(q);
    }
    template <class U> bool operator==(const RecyclingAllocator<U>&) const { return true; }
    template <class U> bool operator!=(const RecyclingAllocator<U>&) const { return false; }
};
}  // namespace mat_detail

const int Dynamic = -1;
enum StorageOptions { ColMajor = 0, RowMajor = 1 };

template <class Scalar, int R, int C, int Opt> class Matrix;
typedef Matrix<float, Dynamic, Dynamic, RowMajor> MatF;

// view of a rectangle of a matrix (row / col / block); assignable, converts to a matrix
class BlockRef;
class ConstBlockRef;
struct DiagonalWrapper { const MatF* v; };

template <>
class Matrix<float, Dynamic, Dynamic, RowMajor> {
public:
    typedef float Scalar;
    Matrix() : r_(0), c_(0) {}
    Matrix(long rows, long cols) : r_(rows), c_(cols), v_((size_t)rows * cols) {}
    inline Matrix(const ConstBlockRef& b);
    inline Matrix(const BlockRef& b);
    static Matrix Zero(long rows, long cols) { return Matrix(rows, cols); }
    static Matrix Ones(long rows, long cols) { return Constant(rows, cols, 1.0f); }
    static Matrix Constant(long rows, long cols, float x) { Matrix m(rows, cols); for (auto& e : m.v_) e = x; return m; }
    long rows() const { return r_; }
    long cols() const { return c_; }
    long size() const { return r_ * c_; }
    float* data() { return v_.data(); }
    const float* data() const { return v_.data(); }
    void resize(long rows, long cols) { r_ = rows; c_ = cols; v_.assign((size_t)rows * cols, 0.f); }
    void setZero() { for (auto& e : v_) e = 0.f; }
    float& operator()(long i, long j) { PPO_MAT_BOUNDS(i >= 0 && i < r_ && j >= 0 && j < c_); return v_[(size_t)i * c_ + j]; }
    float operator()(long i, long j) const { PPO_MAT_BOUNDS(i >= 0 && i < r_ && j >= 0 && j < c_); return v_[(size_t)i * c_ + j]; }
    // reductions (double accumulation, rounded once)
    float squaredNorm() const { double s = 0; for (float e : v_) s += (double)e * e; return (float)s; }
    float norm() const { return std::sqrt(squaredNorm()); }
    float sum() const { double s = 0; for (float e : v_) s += e; return (float)s; }
    float mean() const { return sum() / (float)size(); }
    float maxCoeff() const { float m = v_.at(0); for (float e : v_) m = e > m ? e : m; return m; }
    float minCoeff() const { float m = v_.at(0); for (float e : v_) m = e < m ? e : m; return m; }
    // elementwise
    template <class F> Matrix unary(F f) const { Matrix m(r_, c_); for (size_t i = 0; i < v_.size(); ++i) m.v_[i] = f(v_[i]); return m; }
    template <class F> Matrix binary(const Matrix& o, F f) const {
        assert(r_ == o.r_ && c_ == o.c_);
        Matrix m(r_, c_);
        for (size_t i = 0; i < v_.size(); ++i) m.v_[i] = f(v_[i], o.v_[i]);
        return m;
    }
    Matrix operator-(const Matrix& o) const { return binary(o, [](float a, float b) { return a - b; }); }
    Matrix operator+(const Matrix& o) const { return binary(o, [](float a, float b) { return a + b; }); }
    Matrix operator-() const { return unary([](float a) { return -a; }); }
    Matrix& operator+=(const Matrix& o) { *this = *this + o; return *this; }
    Matrix& operator-=(const Matrix& o) { *this = *this - o; return *this; }
    Matrix operator*(float s) const { return unary([s](float a) { return a * s; }); }
    Matrix operator/(float s) const { return unary([s](float a) { return a / s; }); }
    Matrix& operator*=(float s) { for (auto& e : v_) e *= s; return *this; }
    Matrix& operator/=(float s) { for (auto& e : v_) e /= s; return *this; }
    Matrix operator*(const Matrix& o) const {                 // matrix product
        assert(c_ == o.r_);
        Matrix m(r_, o.c_);
        for (long i = 0; i < r_; ++i)
            for (long k = 0; k < c_; ++k) { const float a = (*this)(i, k); for (long j = 0; j < o.c_; ++j) m(i, j) += a * o(k, j); }
        return m;
    }
    Matrix operator*(const DiagonalWrapper& d) const {        // A * diag(v): column j scaled by v_j (env_normalize.hpp:79,101)
        assert(d.v->size() == c_);
        Matrix m(r_, c_);
        for (long i = 0; i < r_; ++i) for (long j = 0; j < c_; ++j) m(i, j) = (*this)(i, j) * d.v->data()[j];
        return m;
    }
    Matrix& operator*=(const DiagonalWrapper& d) { *this = *this * d; return *this; }
    Matrix& operator*=(const Matrix& o) { *this = *this * o; return *this; }
    Matrix cwiseProduct(const Matrix& o) const { return binary(o, [](float a, float b) { return a * b; }); }
    Matrix cwiseQuotient(const Matrix& o) const { return binary(o, [](float a, float b) { return a / b; }); }
    Matrix cwiseMax(const Matrix& o) const { return binary(o, [](float a, float b) { return a > b ? a : b; }); }
    Matrix cwiseMin(const Matrix& o) const { return binary(o, [](float a, float b) { return a < b ? a : b; }); }
    Matrix cwiseMax(float s) const { return unary([s](float a) { return a > s ? a : s; }); }
    Matrix cwiseMin(float s) const { return unary([s](float a) { return a < s ? a : s; }); }
    Matrix cwiseSqrt() const { return unary([](float a) { return std::sqrt(a); }); }
    Matrix cwiseInverse() const { return unary([](float a) { return 1.0f / a; }); }
    Matrix cwiseAbs() const { return unary([](float a) { return std::fabs(a); }); }
    Matrix transpose() const { Matrix m(c_, r_); for (long i = 0; i < r_; ++i) for (long j = 0; j < c_; ++j) m(j, i) = (*this)(i, j); return m; }
    void transposeInPlace() { *this = transpose(); }
    DiagonalWrapper asDiagonal() const { return DiagonalWrapper{this}; }
    // views
    inline BlockRef block(long i, long j, long r, long c);
    inline ConstBlockRef block(long i, long j, long r, long c) const;
    inline BlockRef row(long i);
    inline ConstBlockRef row(long i) const;
    inline BlockRef col(long j);
    inline ConstBlockRef col(long j) const;
    // partial reductions
    struct Colwise {
        const Matrix* m;
        Matrix sum() const { Matrix o(1, m->c_); for (long i = 0; i < m->r_; ++i) for (long j = 0; j < m->c_; ++j) o(0, j) += (*m)(i, j); return o; }
        Matrix mean() const { return sum() / (float)m->r_; }
    };
    struct Rowwise {
        const Matrix* m;
        Matrix sum() const { Matrix o(m->r_, 1); for (long i = 0; i < m->r_; ++i) for (long j = 0; j < m->c_; ++j) o(i, 0) += (*m)(i, j); return o; }
        Matrix mean() const { return sum() / (float)m->c_; }
        Matrix operator-(const Matrix& rowvec) const {         // (obs.rowwise() - mean.row(0)), env_normalize.hpp:100
            assert(rowvec.size() == m->c_);
            Matrix o(m->r_, m->c_);
            for (long i = 0; i < m->r_; ++i) for (long j = 0; j < m->c_; ++j) o(i, j) = (*m)(i, j) - rowvec.data()[j];
            return o;
        }
        Matrix operator+(const Matrix& rowvec) const { return *this - (-rowvec); }
    };
    Colwise colwise() const { return Colwise{this}; }
    Rowwise rowwise() const { return Rowwise{this}; }

private:
    long r_, c_;
    std::vector<float, mat_detail::RecyclingAllocator<float>> v_;
};

class ConstBlockRef {
public:
    ConstBlockRef(const MatF* m, long i, long j, long r, long c) : m_(m), i_(i), j_(j), r_(r), c_(c) { assert(i >= 0 && j >= 0 && i + r <= m->rows() && j + c <= m->cols()); }
    long rows() const { return r_; }
    long cols() const { return c_; }
    float operator()(long a, long b) const { return (*m_)(i_ + a, j_ + b); }
    MatF eval() const { MatF o(r_, c_); for (long a = 0; a < r_; ++a) for (long b = 0; b < c_; ++b) o(a, b) = (*this)(a, b); return o; }
    MatF operator-(const MatF& o) const { return eval() - o; }
    MatF operator+(const MatF& o) const { return eval() + o; }
    MatF operator*(float s) const { return eval() * s; }
    MatF transpose() const { return eval().transpose(); }
    float squaredNorm() const { return eval().squaredNorm(); }
    float sum() const { return eval().sum(); }
    float mean() const { return eval().mean(); }
    DiagonalWrapper asDiagonal() const { keep_ = eval(); return DiagonalWrapper{&keep_}; }     // valid while this view lives (one full expression)
    MatF cwiseSqrt() const { return eval().cwiseSqrt(); }
    MatF cwiseInverse() const { return eval().cwiseInverse(); }
private:
    const MatF* m_; long i_, j_, r_, c_;
    mutable MatF keep_;
};

class BlockRef {
public:
    BlockRef(MatF* m, long i, long j, long r, long c) : m_(m), i_(i), j_(j), r_(r), c_(c) { assert(i >= 0 && j >= 0 && i + r <= m->rows() && j + c <= m->cols()); }
    long rows() const { return r_; }
    long cols() const { return c_; }
    float& operator()(long a, long b) { return (*m_)(i_ + a, j_ + b); }
    float operator()(long a, long b) const { return (*m_)(i_ + a, j_ + b); }
    MatF eval() const { return ConstBlockRef(m_, i_, j_, r_, c_).eval(); }
    BlockRef& operator=(const MatF& o) { assert(o.rows() == r_ && o.cols() == c_); for (long a = 0; a < r_; ++a) for (long b = 0; b < c_; ++b) (*this)(a, b) = o(a, b); return *this; }
    BlockRef& operator=(const BlockRef& o) { return *this = o.eval(); }
    BlockRef& operator=(const ConstBlockRef& o) { return *this = o.eval(); }
    BlockRef& operator+=(const MatF& o) { return *this = eval() + o; }
    BlockRef& operator-=(const MatF& o) { return *this = eval() - o; }
    BlockRef& operator*=(float s) { return *this = eval() * s; }
    MatF operator-(const MatF& o) const { return eval() - o; }
    MatF operator+(const MatF& o) const { return eval() + o; }
    MatF operator*(float s) const { return eval() * s; }
    MatF transpose() const { return eval().transpose(); }
    float squaredNorm() const { return eval().squaredNorm(); }
    float sum() const { return eval().sum(); }
    float mean() const { return eval().mean(); }
    DiagonalWrapper asDiagonal() const { keep_ = eval(); return DiagonalWrapper{&keep_}; }
    MatF cwiseSqrt() const { return eval().cwiseSqrt(); }
    MatF cwiseInverse() const { return eval().cwiseInverse(); }
private:
    MatF* m_; long i_, j_, r_, c_;
    mutable MatF keep_;
};

inline MatF::Matrix(const ConstBlockRef& b) { *this = b.eval(); }
inline MatF::Matrix(const BlockRef& b) { *this = b.eval(); }
inline BlockRef MatF::block(long i, long j, long r, long c) { return BlockRef(this, i, j, r, c); }
inline ConstBlockRef MatF::block(long i, long j, long r, long c) const { return ConstBlockRef(this, i, j, r, c); }
inline BlockRef MatF::row(long i) { return BlockRef(this, i, 0, 1, c_); }
inline ConstBlockRef MatF::row(long i) const { return ConstBlockRef(this, i, 0, 1, c_); }
inline BlockRef MatF::col(long j) { return BlockRef(this, 0, j, r_, 1); }
inline ConstBlockRef MatF::col(long j) const { return ConstBlockRef(this, 0, j, r_, 1); }

inline MatF operator*(float s, const MatF& a) { return a * s; }
inline MatF operator*(double s, const MatF& a) { return a * (float)s; }
inline MatF operator*(int s, const MatF& a) { return a * (float)s; }
inline MatF operator-(const MatF& a, const ConstBlockRef& b) { return a - b.eval(); }
inline MatF operator-(const MatF& a, const BlockRef& b) { return a - b.eval(); }
inline MatF operator+(const MatF& a, const ConstBlockRef& b) { return a + b.eval(); }
inline MatF operator+(const MatF& a, const BlockRef& b) { return a + b.eval(); }
inline std::ostream& operator<<(std::ostream& o, const MatF& m) {
    for (long i = 0; i < m.rows(); ++i) { for (long j = 0; j < m.cols(); ++j) o << (j ? " " : "") << m(i, j); if (i + 1 < m.rows()) o << "\n"; }
    return o;
}

// Eigen::Map<Mat>(ptr, rows, cols): a view of caller-owned row-major storage (running_statistics.hpp:77-81)
template <class M> class Map;
template <>
class Map<MatF> {
public:
    Map(float* p, long rows, long cols) : p_(p), r_(rows), c_(cols) {}
    long rows() const { return r_; }
    long cols() const { return c_; }
    float& operator()(long i, long j) { return p_[(size_t)i * c_ + j]; }
    float operator()(long i, long j) const { return p_[(size_t)i * c_ + j]; }
    operator MatF() const { MatF m(r_, c_); for (long i = 0; i < r_ * c_; ++i) m.data()[i] = p_[i]; return m; }
    Map& operator=(const MatF& m) { assert(m.rows() == r_ && m.cols() == c_); for (long i = 0; i < r_ * c_; ++i) p_[i] = m.data()[i]; return *this; }
private:
    float* p_; long r_, c_;
};
template <>
class Map<const MatF> {
public:
    Map(const float* p, long rows, long cols) : p_(p), r_(rows), c_(cols) {}
    long rows() const { return r_; }
    long cols() const { return c_; }
    float operator()(long i, long j) const { return p_[(size_t)i * c_ + j]; }
    operator MatF() const { MatF m(r_, c_); for (long i = 0; i < r_ * c_; ++i) m.data()[i] = p_[i]; return m; }
private:
    const float* p_; long r_, c_;
};

}  // namespace Eigen

typedef Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> Mat;
#endif

#include <cstring>
// row helpers that work on either Mat type
inline void mat_set_row(Mat& dst, long i, const float* src) { std::memcpy(dst.data() + (size_t)i * dst.cols(), src, sizeof(float) * (size_t)dst.cols()); }
inline const float* mat_row_ptr(const Mat& m, long i) { return m.data() + (size_t)i * m.cols(); }
