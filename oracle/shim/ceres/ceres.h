// stand-in for <ceres/ceres.h> (Ceres Solver 2.1.0 is neither under /root/reference nor installed): a forward-mode dual
// number ("Jet") so that the reference's templated cost functors can be differentiated exactly as Ceres' AutoDiffCostFunction
// would, plus a Problem that RECORDS what the reference adds to it (parameter blocks, manifolds, residual blocks, losses) and hands
// it to a hook instead of solving.  No solver.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cmath>
#include <string>
#include <utility>
#include <vector>
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

// ---- cost functions: Evaluate() differentiates the functor with Jets, as AutoDiffCostFunction does ----------------------
class CostFunction {
  public:
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    virtual int num_residuals() const = 0;
    virtual std::vector<int> parameter_block_sizes() const = 0;
};
template <class Functor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public CostFunction {
  public:
    explicit AutoDiffCostFunction(Functor *f) : functor_(f) {}
    ~AutoDiffCostFunction() override { delete functor_; }
    int num_residuals() const override { return kNumResiduals; }
    std::vector<int> parameter_block_sizes() const override { return std::vector<int>{Ns...}; }
    // jacobians[b] (may be null, as may jacobians itself): row-major kNumResiduals x Ns[b]
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override
    {
        return eval(parameters, residuals, jacobians, std::make_index_sequence<sizeof...(Ns)>());
    }
    Functor *functor_;

  private:
    template <size_t... I>
    bool eval(double const *const *parameters, double *residuals, double **jacobians, std::index_sequence<I...>) const
    {
        constexpr int kB = sizeof...(Ns), kTotal = (Ns + ... + 0);
        const int sizes[kB] = {Ns...};
        int offs[kB];
        for (int b = 0, o = 0; b < kB; ++b) { offs[b] = o; o += sizes[b]; }
        if (!(*functor_)(parameters[I]..., residuals)) return false;
        if (!jacobians) return true;
        typedef Jet<double, kTotal> J;
        std::vector<J> x((size_t)kTotal);
        for (int b = 0; b < kB; ++b)
            for (int i = 0; i < sizes[b]; ++i) x[offs[b] + i] = J(parameters[b][i], offs[b] + i);
        J r[kNumResiduals];
        if (!(*functor_)((x.data() + offs[I])..., r)) return false;
        for (int b = 0; b < kB; ++b) {
            if (!jacobians[b]) continue;
            for (int a = 0; a < kNumResiduals; ++a)
                for (int i = 0; i < sizes[b]; ++i) jacobians[b][a * sizes[b] + i] = r[a].v[offs[b] + i];
        }
        return true;
    }
};

// ---- the problem description: RECORDED, not solved --------------------------------------------------------------------
// ceres::Solve hands the recorded problem to the hook the test glue installs (lvba_solve_hook()); without a hook it reports
// NO_CONVERGENCE and leaves the parameters untouched.  There is no Ceres solver here: the reference's solve stays unpinned.
class Manifold { public: virtual ~Manifold() {} virtual int AmbientSize() const = 0; virtual int TangentSize() const = 0; };
class EigenQuaternionManifold : public Manifold { public: int AmbientSize() const override { return 4; } int TangentSize() const override { return 3; } };
class LossFunction { public: virtual ~LossFunction() {} virtual double scale() const = 0; };
class HuberLoss : public LossFunction { public: explicit HuberLoss(double a) : a_(a) {} double scale() const override { return a_; } double a_; };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };
class Problem {
  public:
    struct ParameterBlock { double *values; int size; Manifold *manifold; bool constant; };
    struct ResidualBlock { CostFunction *cost; LossFunction *loss; std::vector<double *> parameters; };
    ~Problem()
    {
        for (auto &r : residual_blocks) delete r.cost;
        for (auto &p : parameter_blocks) delete p.manifold;
    }
    void AddParameterBlock(double *values, int size, Manifold *m = nullptr)
    {
        for (auto &p : parameter_blocks)
            if (p.values == values) { if (m) { delete p.manifold; p.manifold = m; } return; }
        parameter_blocks.push_back(ParameterBlock{values, size, m, false});
    }
    void SetParameterBlockConstant(double *values)
    {
        for (auto &p : parameter_blocks) if (p.values == values) p.constant = true;
    }
    template <class... P> void AddResidualBlock(CostFunction *cost, LossFunction *loss, P *...params)
    {
        ResidualBlock rb{cost, loss, std::vector<double *>{params...}};
        const std::vector<int> sz = cost->parameter_block_sizes();
        for (size_t i = 0; i < rb.parameters.size(); ++i) AddParameterBlock(rb.parameters[i], sz[i]);
        residual_blocks.push_back(rb);
    }
    std::vector<ParameterBlock> parameter_blocks;
    std::vector<ResidualBlock> residual_blocks;
};
struct Solver {
    struct Options {
        int max_num_iterations = 50;
        LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
        int num_threads = 1;
        bool minimizer_progress_to_stdout = false;
        double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    };
    struct Summary {
        TerminationType termination_type = NO_CONVERGENCE;
        std::string BriefReport() const { return "ceres stand-in: problem recorded (no Ceres solver in this build)"; }
        std::string FullReport() const { return BriefReport(); }
    };
};
typedef void (*LvbaSolveHook)(const Solver::Options &, Problem *, Solver::Summary *);
inline LvbaSolveHook &lvba_solve_hook() { static LvbaSolveHook h = nullptr; return h; }
inline void Solve(const Solver::Options &o, Problem *p, Solver::Summary *s)
{
    if (lvba_solve_hook()) lvba_solve_hook()(o, p, s);
}
} // namespace ceres
