// stand-in for <ceres/ceres.h> (Ceres Solver 2.1.0 is neither under /root/reference nor installed): a forward-mode dual
// number ("Jet") so that the reference's templated cost functors can be differentiated exactly as Ceres' AutoDiffCostFunction
// would, plus the class names their Create() functions mention.  No solver.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cmath>
namespace ceres {
template <class T, int N>
struct Jet {
    T a;
    T v[N];
    Jet() : a(T(0)) { for (int i = 0; i < N; ++i) v[i] = T(0); }
    Jet(const T &s) : a(s) { for (int i = 0; i < N; ++i) v[i] = T(0); } // NOLINT: implicit, like ceres::Jet
    Jet(const T &s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = T(0); v[k] = T(1); }
};
#define LVBA_JET_BIN(op, A_EXPR, V_EXPR)                                                                     \
    template <class T, int N> Jet<T, N> operator op(const Jet<T, N> &f, const Jet<T, N> &g)                  \
    { Jet<T, N> h; h.a = A_EXPR; for (int i = 0; i < N; ++i) h.v[i] = V_EXPR; return h; }
LVBA_JET_BIN(+, f.a + g.a, f.v[i] + g.v[i])
LVBA_JET_BIN(-, f.a - g.a, f.v[i] - g.v[i])
LVBA_JET_BIN(*, f.a * g.a, f.a * g.v[i] + f.v[i] * g.a)
LVBA_JET_BIN(/, f.a / g.a, (f.v[i] - (f.a / g.a) * g.v[i]) / g.a)
#undef LVBA_JET_BIN
template <class T, int N> Jet<T, N> operator-(const Jet<T, N> &f) { Jet<T, N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <class T, int N> Jet<T, N> operator+(const Jet<T, N> &f, T s) { Jet<T, N> h(f); h.a += s; return h; }
template <class T, int N> Jet<T, N> operator+(T s, const Jet<T, N> &f) { return f + s; }
template <class T, int N> Jet<T, N> operator-(const Jet<T, N> &f, T s) { Jet<T, N> h(f); h.a -= s; return h; }
template <class T, int N> Jet<T, N> operator-(T s, const Jet<T, N> &f) { return -f + s; }
template <class T, int N> Jet<T, N> operator*(const Jet<T, N> &f, T s) { Jet<T, N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h; }
template <class T, int N> Jet<T, N> operator*(T s, const Jet<T, N> &f) { return f * s; }
template <class T, int N> Jet<T, N> operator/(const Jet<T, N> &f, T s) { return f * (T(1) / s); }
template <class T, int N> Jet<T, N> &operator+=(Jet<T, N> &f, const Jet<T, N> &g) { f = f + g; return f; }
template <class T, int N> Jet<T, N> &operator-=(Jet<T, N> &f, const Jet<T, N> &g) { f = f - g; return f; }
template <class T, int N> bool operator<=(const Jet<T, N> &f, const Jet<T, N> &g) { return f.a <= g.a; }
template <class T, int N> bool operator<(const Jet<T, N> &f, const Jet<T, N> &g) { return f.a < g.a; }
template <class T, int N> bool operator>(const Jet<T, N> &f, const Jet<T, N> &g) { return f.a > g.a; }
template <class T, int N> bool operator>=(const Jet<T, N> &f, const Jet<T, N> &g) { return f.a >= g.a; }
template <class T, int N> Jet<T, N> sqrt(const Jet<T, N> &f)
{
    Jet<T, N> h;
    h.a = std::sqrt(f.a);
    const T d = T(1) / (T(2) * h.a);
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * d;
    return h;
}
inline double sqrt(double x) { return std::sqrt(x); }

class CostFunction { public: virtual ~CostFunction() {} };
template <class Functor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public CostFunction {
  public:
    explicit AutoDiffCostFunction(Functor *f) : functor_(f) {}
    ~AutoDiffCostFunction() override { delete functor_; }
    Functor *functor_;
};
} // namespace ceres
