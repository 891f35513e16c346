// lvba_eigen_standin.h -- TEST INFRASTRUCTURE ONLY.
//
// A small, eagerly evaluated stand-in for the subset of the Eigen 3 API that the reference's BALM headers
// (include/BALM/tools.hpp, include/BALM/bavoxel.hpp), include/utils.hpp and src/lvba_system.cpp / src/dataset_io.cpp use, so
// that those files can be compiled UNMODIFIED, from where they lie under /root/reference, into oracle/_ref/libbalm_ref.so and
// oracle/_ref/liblvba_system_ref.so (oracle/Makefile, target `ref`).  Eigen itself is not
// installed in this image and cannot be fetched.  Everything the reference computes with its own statements -- cluster
// transforms, the Hessian / gradient assembly of acc_evaluate2, the LM control flow of damping_iter, voxel keys, the
// octree recursion, the down-sampling rules -- therefore runs as the reference wrote it.  What this file supplies in
// Eigen's place, and what the pinning consequently does NOT cover bit for bit:
//   * dense fixed/dynamic matrices with plain loops (no expression templates: a product chain is evaluated left to right
//     with temporaries, as Eigen does for these sizes, but without FMA contraction guarantees either way);
//   * SelfAdjointEigenSolver<Matrix3d>: cyclic Jacobi, eigenvalues ascending (Eigen: tridiagonalisation + implicit QL);
//     eigenvector signs may differ -- the reference only uses sign-invariant products u u^T and the plane normal;
//   * SimplicialLDLT: unpivoted dense LDL^T of the lower triangle, no fill-reducing permutation (Eigen: AMD ordering);
//   * colPivHouseholderQr().solve: normal equations (only esti_plane uses it; not on the tested path);
//   * SelfAdjointEigenSolver<Matrix4d> (TriangulateTrackDLT): the same cyclic Jacobi;
//   * Quaterniond: construction from a rotation matrix by Eigen's published branch rule, toRotationMatrix, product.
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <memory>
#include <type_traits>
#include <vector>

namespace Eigen {

const int Dynamic = -1;

template <class T>
struct aligned_allocator : public std::allocator<T> {
    aligned_allocator() = default;
    template <class U>
    aligned_allocator(const aligned_allocator<U> &) {}
    template <class U>
    struct rebind { typedef aligned_allocator<U> other; };
};

template <class S, int R, int C>
class Matrix;

namespace detail {
template <class S, int R, int C, bool Fixed = (R >= 0 && C >= 0)>
struct Store {
    std::array<S, (size_t)(R * C)> d{};
    void alloc(int, int) {}
    int rows() const { return R; }
    int cols() const { return C; }
};
template <class S, int R, int C>
struct Store<S, R, C, false> {
    std::vector<S> d;
    int r = (R >= 0 ? R : 0), c = (C >= 0 ? C : 0);
    void alloc(int rr, int cc) { r = rr; c = cc; d.assign((size_t)rr * cc, S(0)); }
    int rows() const { return r; }
    int cols() const { return c; }
};
} // namespace detail

template <class M>
struct CommaInit {
    M &m;
    int idx;
    CommaInit(M &mm, double v) : m(mm), idx(0) { put(v); }
    void put(double v)
    {
        const int c = m.cols();
        m(idx / c, idx % c) = (typename M::Scalar)v; // row-major fill order, as Eigen's comma initialiser
        ++idx;
    }
    CommaInit &operator,(double v) { put(v); return *this; }
};

template <class M, int BR, int BC>
struct Block;
template <class M>
struct DiagRef;
template <class S>
struct LstsqSolver;

template <class S, int R, int C>
class Matrix {
  public:
    typedef S Scalar;
    enum { RowsAtCompileTimeStandin = R, ColsAtCompileTimeStandin = C };
    detail::Store<S, R, C> st;

    Matrix() {}
    explicit Matrix(int n)
    {
        if (R < 0 && C == 1) st.alloc(n, 1);
        else if (R == 1 && C < 0) st.alloc(1, n);
        else if (R < 0 && C < 0) st.alloc(n, n);
    }
    Matrix(int r, int c) { init2(r, c, std::integral_constant<bool, (R * C == 2 && R >= 0 && C >= 0)>()); }
    Matrix(S x, S y, S z)
    {
        static_assert(R * C == 3, "3-vector constructor");
        st.d[0] = x; st.d[1] = y; st.d[2] = z;
    }
    Matrix(S x, S y, S z, S w)
    {
        static_assert(R * C == 4, "4-vector constructor");
        st.d[0] = x; st.d[1] = y; st.d[2] = z; st.d[3] = w;
    }
    template <int R2, int C2>
    Matrix(const Matrix<S, R2, C2> &o) { assign(o); }
    template <int R2, int C2>
    Matrix &operator=(const Matrix<S, R2, C2> &o) { assign(o); return *this; }
    Matrix(const Matrix &) = default;
    Matrix &operator=(const Matrix &) = default;

    int rows() const { return st.rows(); }
    int cols() const { return st.cols(); }
    int size() const { return rows() * cols(); }
    void resize(int r, int c) { st.alloc(r, c); }
    void resize(int n)
    {
        if (C == 1) st.alloc(n, 1); else st.alloc(1, n);
    }
    S &operator()(int i, int j) { return st.d[(size_t)j * rows() + i]; }
    const S &operator()(int i, int j) const { return st.d[(size_t)j * rows() + i]; }
    S &operator()(int i) { return st.d[i]; }
    const S &operator()(int i) const { return st.d[i]; }
    S &operator[](int i) { return st.d[i]; }
    const S &operator[](int i) const { return st.d[i]; }
    S &x() { return st.d[0]; }
    S &y() { return st.d[1]; }
    S &z() { return st.d[2]; }
    const S &x() const { return st.d[0]; }
    const S &y() const { return st.d[1]; }
    const S &z() const { return st.d[2]; }
    S &w() { return st.d[3]; }
    const S &w() const { return st.d[3]; }
    S *data() { return st.d.data(); }
    const S *data() const { return st.d.data(); }

    Matrix &setZero() { for (auto &v : st.d) v = S(0); return *this; }
    Matrix &setOnes() { for (auto &v : st.d) v = S(1); return *this; }
    Matrix &setIdentity()
    {
        setZero();
        for (int i = 0; i < std::min(rows(), cols()); ++i) (*this)(i, i) = S(1);
        return *this;
    }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Zero() { Matrix m; m.setZero(); return m; }
    static Matrix Identity(int r, int c) { Matrix m; m.resize(r, c); m.setIdentity(); return m; }
    static Matrix Zero(int r, int c) { Matrix m; m.resize(r, c); return m; }
    static Matrix Zero(int n) { Matrix m(n); return m; }

    Matrix<S, C, R> transpose() const
    {
        Matrix<S, C, R> t;
        t.st.alloc(cols(), rows());
        for (int j = 0; j < cols(); ++j)
            for (int i = 0; i < rows(); ++i) t(j, i) = (*this)(i, j);
        return t;
    }
    S squaredNorm() const { S s = 0; for (auto v : st.d) s += v * v; return s; }
    S norm() const { return std::sqrt(squaredNorm()); }
    Matrix normalized() const { return *this / norm(); }
    void normalize() { *this /= norm(); }
    S trace() const { S s = 0; for (int i = 0; i < std::min(rows(), cols()); ++i) s += (*this)(i, i); return s; }
    S sum() const { S s = 0; for (auto v : st.d) s += v; return s; }
    Matrix cwiseMin(const Matrix &o) const { Matrix m(*this); for (int i = 0; i < size(); ++i) m.st.d[i] = std::min(st.d[i], o.st.d[i]); return m; }
    Matrix cwiseMax(const Matrix &o) const { Matrix m(*this); for (int i = 0; i < size(); ++i) m.st.d[i] = std::max(st.d[i], o.st.d[i]); return m; }
    bool isZero(S prec = S(1e-12)) const { for (auto v : st.d) if (!(std::abs(v) <= prec)) return false; return true; }
    bool allFinite() const { for (auto v : st.d) if (!std::isfinite(v)) return false; return true; }
    static Matrix UnitX() { Matrix m; m.st.d[0] = S(1); return m; }
    static Matrix UnitY() { Matrix m; m.st.d[1] = S(1); return m; }
    static Matrix UnitZ() { Matrix m; m.st.d[2] = S(1); return m; }
    template <int R2, int C2>
    S dot(const Matrix<S, R2, C2> &o) const
    {
        assert(size() == o.size());
        S s = 0;
        for (int i = 0; i < size(); ++i) s += st.d[i] * o.st.d[i];
        return s;
    }
    Matrix cross(const Matrix &o) const
    {
        static_assert(R * C == 3, "cross of 3-vectors");
        return Matrix(st.d[1] * o.st.d[2] - st.d[2] * o.st.d[1], st.d[2] * o.st.d[0] - st.d[0] * o.st.d[2],
                      st.d[0] * o.st.d[1] - st.d[1] * o.st.d[0]);
    }
    Matrix<S, R, 1> col(int j) const
    {
        Matrix<S, R, 1> v;
        v.st.alloc(rows(), 1);
        for (int i = 0; i < rows(); ++i) v(i) = (*this)(i, j);
        return v;
    }
    Matrix<S, 1, C> row(int i) const
    {
        Matrix<S, 1, C> v;
        v.st.alloc(1, cols());
        for (int j = 0; j < cols(); ++j) v(j) = (*this)(i, j);
        return v;
    }
    Matrix<S, Dynamic, 1> diagonal() const
    {
        Matrix<S, Dynamic, 1> v(std::min(rows(), cols()));
        for (int i = 0; i < v.size(); ++i) v(i) = (*this)(i, i);
        return v;
    }
    DiagRef<Matrix> diagonal() { return DiagRef<Matrix>{*this}; }
    template <int BR, int BC>
    Block<Matrix, BR, BC> block(int i, int j) { return Block<Matrix, BR, BC>{*this, i, j}; }
    template <int BR, int BC>
    Matrix<S, BR, BC> block(int i0, int j0) const
    {
        Matrix<S, BR, BC> b;
        for (int j = 0; j < BC; ++j)
            for (int i = 0; i < BR; ++i) b(i, j) = (*this)(i0 + i, j0 + j);
        return b;
    }
    template <int N>
    Matrix<S, N, 1> head() const
    {
        Matrix<S, N, 1> b;
        for (int i = 0; i < N; ++i) b(i) = st.d[i];
        return b;
    }
    template <class T>
    Matrix<T, R, C> cast() const
    {
        Matrix<T, R, C> m;
        m.st.alloc(rows(), cols());
        for (int i = 0; i < size(); ++i) m.st.d[i] = (T)st.d[i];
        return m;
    }
    CommaInit<Matrix> operator<<(double v) { return CommaInit<Matrix>(*this, v); }

    Matrix &operator+=(const Matrix &o) { assert(size() == o.size()); for (int i = 0; i < size(); ++i) st.d[i] += o.st.d[i]; return *this; }
    Matrix &operator-=(const Matrix &o) { assert(size() == o.size()); for (int i = 0; i < size(); ++i) st.d[i] -= o.st.d[i]; return *this; }
    Matrix &operator*=(S s) { for (auto &v : st.d) v *= s; return *this; }
    Matrix &operator/=(S s) { for (auto &v : st.d) v /= s; return *this; }
    Matrix operator-() const { Matrix m(*this); for (auto &v : m.st.d) v = -v; return m; }
    Matrix operator+(const Matrix &o) const { Matrix m(*this); m += o; return m; }
    Matrix operator-(const Matrix &o) const { Matrix m(*this); m -= o; return m; }
    Matrix operator*(S s) const { Matrix m(*this); m *= s; return m; }
    Matrix operator/(S s) const { Matrix m(*this); m /= s; return m; }

    LstsqSolver<S> colPivHouseholderQr() const;

  private:
    void init2(int r, int c, std::true_type) { st.d[0] = (S)r; st.d[1] = (S)c; }
    void init2(int r, int c, std::false_type) { st.alloc(r, c); }
    template <int R2, int C2>
    void assign(const Matrix<S, R2, C2> &o)
    {
        static_assert((R < 0 || R2 < 0 || R == R2) && (C < 0 || C2 < 0 || C == C2), "matrix dimensions differ");
        st.alloc(o.rows(), o.cols());
        assert(rows() == o.rows() && cols() == o.cols());
        for (int i = 0; i < o.size(); ++i) st.d[i] = o.st.d[i];
    }
};

template <class S, int R, int C>
Matrix<S, R, C> operator*(double s, const Matrix<S, R, C> &m) { return m * (S)s; }

template <class S, int R, int K, int K2, int C>
Matrix<S, R, C> operator*(const Matrix<S, R, K> &a, const Matrix<S, K2, C> &b)
{
    static_assert(K < 0 || K2 < 0 || K == K2, "inner dimensions differ");
    assert(a.cols() == b.rows());
    Matrix<S, R, C> m;
    m.st.alloc(a.rows(), b.cols());
    for (int j = 0; j < b.cols(); ++j)
        for (int i = 0; i < a.rows(); ++i) {
            S s = 0;
            for (int k = 0; k < a.cols(); ++k) s += a(i, k) * b(k, j);
            m(i, j) = s;
        }
    return m;
}

// lvalue view of a fixed-size block
template <class M, int BR, int BC>
struct Block {
    typedef typename M::Scalar S;
    typedef Matrix<S, BR, BC> Value;
    M &m;
    int i0, j0;
    Value value() const
    {
        Value b;
        for (int j = 0; j < BC; ++j)
            for (int i = 0; i < BR; ++i) b(i, j) = m(i0 + i, j0 + j);
        return b;
    }
    operator Value() const { return value(); }
    Block &operator=(const Value &v)
    {
        for (int j = 0; j < BC; ++j)
            for (int i = 0; i < BR; ++i) m(i0 + i, j0 + j) = v(i, j);
        return *this;
    }
    Block &operator=(const Block &o) { return *this = o.value(); }
    template <class M2>
    Block &operator=(const Block<M2, BR, BC> &o) { return *this = o.value(); }
    Block &operator+=(const Value &v) { return *this = value() + v; }
    Block &operator-=(const Value &v) { return *this = value() - v; }
    S norm() const { return value().norm(); }
    Matrix<S, BC, BR> transpose() const { return value().transpose(); }
    Value operator-() const { return -value(); }
};
template <class M, int BR, int BC>
Matrix<typename M::Scalar, BR, BC> operator+(const Block<M, BR, BC> &a, const Matrix<typename M::Scalar, BR, BC> &b) { return a.value() + b; }
template <class M, int BR, int BC>
Matrix<typename M::Scalar, BR, BC> operator+(const Matrix<typename M::Scalar, BR, BC> &a, const Block<M, BR, BC> &b) { return a + b.value(); }
template <class M, int BR, int BC>
Matrix<typename M::Scalar, BR, BC> operator-(const Block<M, BR, BC> &a, const Matrix<typename M::Scalar, BR, BC> &b) { return a.value() - b; }
template <class M, int BR, int BC>
Matrix<typename M::Scalar, BR, BC> operator-(const Matrix<typename M::Scalar, BR, BC> &a, const Block<M, BR, BC> &b) { return a - b.value(); }

template <class M>
struct DiagRef {
    typedef typename M::Scalar S;
    M &m;
    Matrix<S, Dynamic, 1> value() const { return static_cast<const M &>(m).diagonal(); }
    operator Matrix<S, Dynamic, 1>() const { return value(); }
    DiagRef &operator=(const Matrix<S, Dynamic, 1> &v)
    {
        for (int i = 0; i < v.size(); ++i) m(i, i) = v(i);
        return *this;
    }
    DiagRef &operator=(const DiagRef &o) { return *this = o.value(); }
    template <class M2>
    DiagRef &operator=(const DiagRef<M2> &o) { return *this = o.value(); }
};

template <class S>
struct LstsqSolver {
    Matrix<S, Dynamic, Dynamic> A;
    template <int R2, int C2>
    Matrix<S, Dynamic, 1> solve(const Matrix<S, R2, C2> &b) const
    {
        const int n = A.cols();
        Matrix<S, Dynamic, Dynamic> N = A.transpose() * A;
        Matrix<S, Dynamic, 1> r = A.transpose() * Matrix<S, Dynamic, 1>(b);
        for (int k = 0; k < n; ++k) { // Gaussian elimination with partial pivoting
            int p = k;
            for (int i = k + 1; i < n; ++i) if (std::fabs(N(i, k)) > std::fabs(N(p, k))) p = i;
            for (int j = 0; j < n; ++j) std::swap(N(k, j), N(p, j));
            std::swap(r(k), r(p));
            for (int i = k + 1; i < n; ++i) {
                const S f = N(i, k) / N(k, k);
                for (int j = k; j < n; ++j) N(i, j) -= f * N(k, j);
                r(i) -= f * r(k);
            }
        }
        for (int k = n - 1; k >= 0; --k) {
            for (int j = k + 1; j < n; ++j) r(k) -= N(k, j) * r(j);
            r(k) /= N(k, k);
        }
        return r;
    }
};
template <class S, int R, int C>
LstsqSolver<S> Matrix<S, R, C>::colPivHouseholderQr() const { return LstsqSolver<S>{Matrix<S, Dynamic, Dynamic>(*this)}; }

typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, Dynamic, 1> VectorXd;

// Symmetric eigen-decomposition, eigenvalues ascending, eigenvectors in the columns (cyclic Jacobi; sizes here are 3).
template <class M>
class SelfAdjointEigenSolver {
  public:
    typedef typename M::Scalar S;
    SelfAdjointEigenSolver() {}
    explicit SelfAdjointEigenSolver(const M &A) { compute(A); }
    SelfAdjointEigenSolver &compute(const M &A0)
    {
        const int n = A0.rows();
        M A(A0);
        for (int j = 0; j < n; ++j) // Eigen reads the lower triangle only
            for (int i = 0; i < j; ++i) A(i, j) = A(j, i);
        vec_.resize(n, n);
        vec_.setIdentity();
        for (int sweep = 0; sweep < 64; ++sweep) {
            S off = 0, dia = 0;
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) (i == j ? dia : off) += A(i, j) * A(i, j);
            if (off <= S(1e-34) * dia || off == 0) break;
            for (int p = 0; p < n - 1; ++p)
                for (int q = p + 1; q < n; ++q) {
                    if (A(p, q) == 0) continue;
                    const S theta = (A(q, q) - A(p, p)) / (2 * A(p, q));
                    const S t = (theta >= 0 ? S(1) : S(-1)) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                    const S c = 1 / std::sqrt(t * t + 1), s = t * c;
                    for (int k = 0; k < n; ++k) {
                        const S akp = A(k, p), akq = A(k, q);
                        A(k, p) = c * akp - s * akq;
                        A(k, q) = s * akp + c * akq;
                    }
                    for (int k = 0; k < n; ++k) {
                        const S apk = A(p, k), aqk = A(q, k);
                        A(p, k) = c * apk - s * aqk;
                        A(q, k) = s * apk + c * aqk;
                    }
                    for (int k = 0; k < n; ++k) {
                        const S vkp = vec_(k, p), vkq = vec_(k, q);
                        vec_(k, p) = c * vkp - s * vkq;
                        vec_(k, q) = s * vkp + c * vkq;
                    }
                }
        }
        val_.resize(n, 1);
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return A(a, a) < A(b, b); });
        M V(vec_);
        for (int j = 0; j < n; ++j) {
            val_(j) = A(order[j], order[j]);
            for (int k = 0; k < n; ++k) vec_(k, j) = V(k, order[j]);
        }
        return *this;
    }
    const Matrix<S, M::RowsAtCompileTimeStandin, 1> &eigenvalues() const { return val_; }
    const M &eigenvectors() const { return vec_; }
    int info() const { return 0; } // Eigen::Success: the Jacobi sweeps below always terminate

  private:
    Matrix<S, M::RowsAtCompileTimeStandin, 1> val_;
    M vec_;
};

// Axis-angle rotation: from a rotation matrix (jr_inv in tools.hpp) or from (angle, axis) (EulerToRot in utils.hpp);
// products compose rotations.
template <class S>
class AngleAxis {
  public:
    typedef Matrix<S, 3, 3> Mat3;
    typedef Matrix<S, 3, 1> Vec3;
    explicit AngleAxis(const Mat3 &Rm) : R_(Rm)
    {
        const S c = std::min(S(1), std::max(S(-1), S(0.5) * (Rm.trace() - S(1))));
        angle_ = std::acos(c);
        Vec3 k(Rm(2, 1) - Rm(1, 2), Rm(0, 2) - Rm(2, 0), Rm(1, 0) - Rm(0, 1));
        const S n = k.norm();
        axis_ = n > 0 ? Vec3(k / n) : Vec3(S(1), S(0), S(0));
    }
    AngleAxis(S angle, const Vec3 &axis) : axis_(axis), angle_(angle)
    {
        Mat3 K;
        K << S(0), -axis[2], axis[1], axis[2], S(0), -axis[0], -axis[1], axis[0], S(0);
        R_ = Mat3::Identity() + std::sin(angle) * K + (S(1) - std::cos(angle)) * (K * K);
    }
    const Vec3 &axis() const { return axis_; }
    S angle() const { return angle_; }
    Mat3 toRotationMatrix() const { return R_; }
    AngleAxis operator*(const AngleAxis &o) const { return AngleAxis(Mat3(R_ * o.R_)); }

  private:
    Mat3 R_;
    Vec3 axis_;
    S angle_;
};
typedef AngleAxis<double> AngleAxisd;
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };

// Eigen::Quaternion<S>: (w, x, y, z) constructor, construction from a rotation matrix by the branch on the trace / the
// largest diagonal entry (Eigen's published quaternion-from-matrix rule), toRotationMatrix.
template <class S>
class Quaternion {
  public:
    typedef Matrix<S, 3, 3> Mat3;
    Quaternion() : w_(1), x_(0), y_(0), z_(0) {}
    Quaternion(S w, S x, S y, S z) : w_(w), x_(x), y_(y), z_(z) {}
    explicit Quaternion(const Mat3 &m)
    {
        S t = m.trace();
        if (t > S(0)) {
            t = std::sqrt(t + S(1));
            w_ = S(0.5) * t;
            t = S(0.5) / t;
            x_ = (m(2, 1) - m(1, 2)) * t;
            y_ = (m(0, 2) - m(2, 0)) * t;
            z_ = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            S v[3];
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + S(1));
            v[i] = S(0.5) * t;
            t = S(0.5) / t;
            w_ = (m(k, j) - m(j, k)) * t;
            v[j] = (m(j, i) + m(i, j)) * t;
            v[k] = (m(k, i) + m(i, k)) * t;
            x_ = v[0]; y_ = v[1]; z_ = v[2];
        }
    }
    S w() const { return w_; }
    S x() const { return x_; }
    S y() const { return y_; }
    S z() const { return z_; }
    S norm() const { return std::sqrt(w_ * w_ + x_ * x_ + y_ * y_ + z_ * z_); }
    void normalize() { const S n = norm(); w_ /= n; x_ /= n; y_ /= n; z_ /= n; }
    Quaternion normalized() const { Quaternion q(*this); q.normalize(); return q; }
    Quaternion conjugate() const { return Quaternion(w_, -x_, -y_, -z_); }
    Quaternion operator*(const Quaternion &b) const
    {
        return Quaternion(w_ * b.w_ - x_ * b.x_ - y_ * b.y_ - z_ * b.z_, w_ * b.x_ + x_ * b.w_ + y_ * b.z_ - z_ * b.y_,
                          w_ * b.y_ + y_ * b.w_ + z_ * b.x_ - x_ * b.z_, w_ * b.z_ + z_ * b.w_ + x_ * b.y_ - y_ * b.x_);
    }
    Mat3 toRotationMatrix() const
    {
        const S tx = S(2) * x_, ty = S(2) * y_, tz = S(2) * z_;
        const S twx = tx * w_, twy = ty * w_, twz = tz * w_, txx = tx * x_, txy = ty * x_, txz = tz * x_, tyy = ty * y_, tyz = tz * y_,
                tzz = tz * z_;
        Mat3 r;
        r(0, 0) = S(1) - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
        r(1, 0) = txy + twz; r(1, 1) = S(1) - (txx + tzz); r(1, 2) = tyz - twx;
        r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = S(1) - (txx + tyy);
        return r;
    }
    Matrix<S, 3, 1> operator*(const Matrix<S, 3, 1> &v) const { return toRotationMatrix() * v; }

  private:
    S w_, x_, y_, z_;
};
typedef Quaternion<double> Quaterniond;

template <class S>
class Triplet {
  public:
    Triplet() : r_(0), c_(0), v_(0) {}
    Triplet(int r, int c, S v) : r_(r), c_(c), v_(v) {}
    int row() const { return r_; }
    int col() const { return c_; }
    S value() const { return v_; }

  private:
    int r_, c_;
    S v_;
};

// "Sparse" matrix kept dense: the reference fills it from the non-zeros of a dense matrix anyway (bavoxel.hpp:696-703)
template <class S>
class SparseMatrix {
  public:
    typedef S Scalar;
    SparseMatrix() {}
    SparseMatrix(int r, int c) { dense.resize(r, c); }
    template <class It>
    void setFromTriplets(It b, It e)
    {
        dense.setZero();
        for (; b != e; ++b) dense(b->row(), b->col()) += b->value(); // duplicates are summed, as in Eigen
    }
    void makeCompressed() {}
    int rows() const { return dense.rows(); }
    int cols() const { return dense.cols(); }
    Matrix<S, Dynamic, Dynamic> dense;
};


// Unpivoted LDL^T of the lower triangle (Eigen's default UpLo = Lower), D may be indefinite -- like SimplicialLDLT.
template <class SpMat>
class SimplicialLDLT {
  public:
    typedef typename SpMat::Scalar S;
    SimplicialLDLT() {}
    explicit SimplicialLDLT(const SpMat &A) { compute(A); }
    SimplicialLDLT &compute(const SpMat &A)
    {
        const int n = A.rows();
        L = A.dense;
        info_ = Success;
        for (int j = 0; j < n; ++j) {
            S d = L(j, j);
            for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k) * L(k, k);
            L(j, j) = d;
            if (d == 0) { info_ = NumericalIssue; return *this; }
            for (int i = j + 1; i < n; ++i) {
                S s = L(i, j);
                for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k) * L(k, k);
                L(i, j) = s / d;
            }
        }
        return *this;
    }
    template <int R2, int C2>
    Matrix<S, Dynamic, 1> solve(const Matrix<S, R2, C2> &b) const
    {
        const int n = L.rows();
        Matrix<S, Dynamic, 1> x(b);
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < i; ++k) x(i) -= L(i, k) * x(k);
        for (int i = 0; i < n; ++i) x(i) /= L(i, i);
        for (int i = n - 1; i >= 0; --i)
            for (int k = i + 1; k < n; ++k) x(i) -= L(k, i) * x(k);
        return x;
    }
    ComputationInfo info() const { return info_; }

  private:
    Matrix<S, Dynamic, Dynamic> L;
    ComputationInfo info_ = Success;
};

} // namespace Eigen
