// lvref_ceres.hpp - TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  The handful of Ceres names /root/reference/src/initial_sfm.cpp uses
// (Problem, AutoDiffCostFunction, QuaternionParameterization, Solver, QuaternionRotatePoint), so that the file can be compiled where it
// lies (oracle/Makefile, target `ref` -> oracle/_ref/liblvref_dyninit.so).  Ceres is not installed.  Behind the names: derivatives by
// central differences of the functor (no dual numbers), a dense Levenberg-Marquardt on the tangent space of the free blocks, run to
// convergence.  Real Ceres stops at its own tolerances (function tolerance 1e-6, and the reference's 0.2 s time limit): the minimum is
// the same, the digits it is reached to are not - comparisons through this header state a tolerance accordingly.
#pragma once
#include <vector>
#include <map>
#include <cmath>
#include <string>
#include <utility>
#include <algorithm>
namespace ceres {
struct CostFunction {
    virtual ~CostFunction() {}
    virtual int num_residuals() const = 0;
    virtual const std::vector<int>& block_sizes() const = 0;
    virtual void residuals(double const* const* params, double* r) const = 0;
};
template <typename F, int NR, int... N> class AutoDiffCostFunction : public CostFunction {
    F* f_; std::vector<int> sizes_{N...};
    template <size_t... I> void call(double const* const* p, double* r, std::index_sequence<I...>) const { (*f_)(p[I]..., r); }
public:
    explicit AutoDiffCostFunction(F* f) : f_(f) {}
    ~AutoDiffCostFunction() { delete f_; }
    int num_residuals() const override { return NR; }
    const std::vector<int>& block_sizes() const override { return sizes_; }
    void residuals(double const* const* params, double* r) const override { call(params, r, std::make_index_sequence<sizeof...(N)>()); }
};
struct LossFunction {};
struct LocalParameterization { virtual ~LocalParameterization() {} virtual int tangent() const = 0; virtual void plus(const double* x, const double* d, double* out) const = 0; };
struct QuaternionParameterization : LocalParameterization {
    int tangent() const override { return 3; }
    void plus(const double* x, const double* d, double* o) const override
    {   // Ceres: x_plus_delta = [cos|d|, sin|d| d/|d|] (x) x, quaternions as (w, x, y, z)
        const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        double q[4] = {1, 0, 0, 0};
        if (n > 0) { const double s = std::sin(n) / n; q[0] = std::cos(n); q[1] = s * d[0]; q[2] = s * d[1]; q[3] = s * d[2]; }
        o[0] = q[0] * x[0] - q[1] * x[1] - q[2] * x[2] - q[3] * x[3];
        o[1] = q[0] * x[1] + q[1] * x[0] + q[2] * x[3] - q[3] * x[2];
        o[2] = q[0] * x[2] - q[1] * x[3] + q[2] * x[0] + q[3] * x[1];
        o[3] = q[0] * x[3] + q[1] * x[2] - q[2] * x[1] + q[3] * x[0];
    }
};
template <typename T> inline void QuaternionRotatePoint(const T q[4], const T pt[3], T result[3])
{   // ceres/rotation.h: normalise, then the unit-quaternion rotation, (w, x, y, z)
    const T scale = T(1) / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const T w = q[0] * scale, x = q[1] * scale, y = q[2] * scale, z = q[3] * scale;
    const T t2 = w * x, t3 = w * y, t4 = w * z, t5 = -x * x, t6 = x * y, t7 = x * z, t8 = -y * y, t9 = y * z, t1 = -z * z;
    result[0] = T(2) * ((t8 + t1) * pt[0] + (t6 - t4) * pt[1] + (t3 + t7) * pt[2]) + pt[0];
    result[1] = T(2) * ((t4 + t6) * pt[0] + (t5 + t1) * pt[1] + (t9 - t2) * pt[2]) + pt[1];
    result[2] = T(2) * ((t7 - t3) * pt[0] + (t2 + t9) * pt[1] + (t5 + t8) * pt[2]) + pt[2];
}
enum LinearSolverType { DENSE_QR, DENSE_SCHUR, SPARSE_SCHUR, DENSE_NORMAL_CHOLESKY };
enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE };
class Problem {
public:
    struct Block { double* p; int size; LocalParameterization* lp; bool constant; };
    struct Res { CostFunction* f; std::vector<double*> p; };
    std::vector<Block> blocks; std::map<double*, int> index; std::vector<Res> res; std::vector<LocalParameterization*> owned;
    ~Problem() { for (auto& r : res) delete r.f; std::sort(owned.begin(), owned.end()); owned.erase(std::unique(owned.begin(), owned.end()), owned.end()); for (auto* l : owned) delete l; }
    void AddParameterBlock(double* p, int size, LocalParameterization* lp = nullptr) { if (index.count(p)) return; index[p] = (int)blocks.size(); blocks.push_back(Block{p, size, lp, false}); if (lp) owned.push_back(lp); }
    void SetParameterBlockConstant(double* p) { blocks[(size_t)index.at(p)].constant = true; }
    template <typename... P> void AddResidualBlock(CostFunction* f, LossFunction*, P*... ps)
    {
        Res r; r.f = f; r.p = {ps...};
        for (size_t k = 0; k < r.p.size(); ++k) AddParameterBlock(r.p[k], f->block_sizes()[k]);
        res.push_back(r);
    }
};
struct Solver {
    struct Options { LinearSolverType linear_solver_type = DENSE_QR; double max_solver_time_in_seconds = 1e9; bool minimizer_progress_to_stdout = false; int max_num_iterations = 50; };
    struct Summary { TerminationType termination_type = NO_CONVERGENCE; double final_cost = 0, initial_cost = 0; int iterations = 0; std::string BriefReport() const { return "lvref_ceres"; } };
};
inline void Solve(const Solver::Options& opt, Problem* pb, Solver::Summary* sum)
{
    // tangent layout of the free blocks
    std::vector<int> off(pb->blocks.size(), -1); int nt = 0;
    for (size_t b = 0; b < pb->blocks.size(); ++b) if (!pb->blocks[b].constant) { off[b] = nt; nt += pb->blocks[b].lp ? pb->blocks[b].lp->tangent() : pb->blocks[b].size; }
    int nr = 0; for (auto& r : pb->res) nr += r.f->num_residuals();
    auto cost_of = [&](std::vector<double>& rv) { rv.assign((size_t)nr, 0.0); int k = 0; double c = 0; for (auto& r : pb->res) { r.f->residuals(r.p.data(), &rv[(size_t)k]); k += r.f->num_residuals(); } for (double v : rv) c += v * v; return 0.5 * c; };
    auto apply = [&](const std::vector<double>& d, std::vector<std::vector<double>>& saved) {
        saved.clear();
        for (size_t b = 0; b < pb->blocks.size(); ++b) {
            Problem::Block& B = pb->blocks[b]; saved.push_back(std::vector<double>(B.p, B.p + B.size));
            if (B.constant) continue;
            if (B.lp) { std::vector<double> o((size_t)B.size); B.lp->plus(B.p, &d[(size_t)off[b]], o.data()); for (int i = 0; i < B.size; ++i) B.p[i] = o[(size_t)i]; }
            else for (int i = 0; i < B.size; ++i) B.p[i] += d[(size_t)(off[b] + i)];
        }
    };
    auto restore = [&](const std::vector<std::vector<double>>& saved) { for (size_t b = 0; b < pb->blocks.size(); ++b) for (int i = 0; i < pb->blocks[b].size; ++i) pb->blocks[b].p[i] = saved[b][(size_t)i]; };
    std::vector<double> r0, r1; double cost = cost_of(r0); sum->initial_cost = cost;
    double lambda = 1e-4; const double h = 1e-6;
    bool ceres_stop = false;      // a successful step changed the cost by less than Ceres' default function_tolerance (1e-6) times the cost: where Ceres itself reports CONVERGENCE
    if (nt == 0 || nr == 0) { sum->final_cost = cost; sum->termination_type = CONVERGENCE; return; }
    for (int it = 0; it < opt.max_num_iterations; ++it) {
        // J^T J and J^T r, one residual block at a time (central differences in the tangent of each of its free parameter blocks)
        std::vector<double> H((size_t)nt * nt, 0.0), g((size_t)nt, 0.0);
        int row = 0;
        for (auto& r : pb->res) {
            const int m = r.f->num_residuals();
            std::vector<std::vector<double>> Jb(r.p.size()); std::vector<int> col(r.p.size(), -1), tsz(r.p.size(), 0);
            for (size_t k = 0; k < r.p.size(); ++k) {
                Problem::Block& B = pb->blocks[(size_t)pb->index.at(r.p[k])];
                if (B.constant) continue;
                col[k] = off[(size_t)pb->index.at(r.p[k])]; tsz[k] = B.lp ? B.lp->tangent() : B.size;
                Jb[k].assign((size_t)m * tsz[k], 0.0);
                std::vector<double> keep(B.p, B.p + B.size), rp((size_t)m), rm((size_t)m), d((size_t)tsz[k]), o((size_t)B.size);
                for (int j = 0; j < tsz[k]; ++j) {
                    for (int sgn = 0; sgn < 2; ++sgn) {
                        std::fill(d.begin(), d.end(), 0.0); d[(size_t)j] = sgn ? -h : h;
                        if (B.lp) { B.lp->plus(keep.data(), d.data(), o.data()); for (int i = 0; i < B.size; ++i) B.p[i] = o[(size_t)i]; } else { for (int i = 0; i < B.size; ++i) B.p[i] = keep[(size_t)i] + d[(size_t)i]; }
                        r.f->residuals(r.p.data(), sgn ? rm.data() : rp.data());
                    }
                    for (int i = 0; i < m; ++i) Jb[k][(size_t)i * tsz[k] + j] = (rp[(size_t)i] - rm[(size_t)i]) / (2 * h);
                }
                for (int i = 0; i < B.size; ++i) B.p[i] = keep[(size_t)i];
            }
            for (size_t a = 0; a < r.p.size(); ++a) {
                if (col[a] < 0) continue;
                for (int i = 0; i < m; ++i) for (int ja = 0; ja < tsz[a]; ++ja) {
                    const double ja_v = Jb[a][(size_t)i * tsz[a] + ja];
                    g[(size_t)(col[a] + ja)] += ja_v * r0[(size_t)(row + i)];
                    for (size_t b = 0; b < r.p.size(); ++b) { if (col[b] < 0) continue; for (int jb = 0; jb < tsz[b]; ++jb) H[(size_t)(col[a] + ja) * nt + col[b] + jb] += ja_v * Jb[b][(size_t)i * tsz[b] + jb]; }
                }
            }
            row += m;
        }
        bool stepped = false; double gmax = 0; for (double v : g) gmax = std::max(gmax, std::fabs(v));
        if (gmax < 1e-14) { sum->termination_type = CONVERGENCE; break; }
        for (int tries = 0; tries < 12 && !stepped; ++tries) {
            // (H + lambda diag H) d = -g by Cholesky
            std::vector<double> A = H, d((size_t)nt);
            for (int i = 0; i < nt; ++i) A[(size_t)i * nt + i] += lambda * std::max(H[(size_t)i * nt + i], 1e-12);
            bool ok = true;
            for (int j = 0; j < nt && ok; ++j) {
                double s = A[(size_t)j * nt + j]; for (int k = 0; k < j; ++k) s -= A[(size_t)j * nt + k] * A[(size_t)j * nt + k];
                if (s <= 0) { ok = false; break; }
                const double l = std::sqrt(s); A[(size_t)j * nt + j] = l;
                for (int i = j + 1; i < nt; ++i) { double t = A[(size_t)i * nt + j]; for (int k = 0; k < j; ++k) t -= A[(size_t)i * nt + k] * A[(size_t)j * nt + k]; A[(size_t)i * nt + j] = t / l; }
            }
            if (!ok) { lambda *= 10; continue; }
            for (int i = 0; i < nt; ++i) { double t = -g[(size_t)i]; for (int k = 0; k < i; ++k) t -= A[(size_t)i * nt + k] * d[(size_t)k]; d[(size_t)i] = t / A[(size_t)i * nt + i]; }
            for (int i = nt - 1; i >= 0; --i) { double t = d[(size_t)i]; for (int k = i + 1; k < nt; ++k) t -= A[(size_t)k * nt + i] * d[(size_t)k]; d[(size_t)i] = t / A[(size_t)i * nt + i]; }
            std::vector<std::vector<double>> saved; apply(d, saved);
            const double c1 = cost_of(r1);
            double dn = 0; for (double v : d) dn = std::max(dn, std::fabs(v));
            if (c1 <= cost) { const bool tiny = dn < 1e-13 || cost - c1 <= 1e-18 * std::max(cost, 1e-30); if (cost - c1 <= 1e-6 * cost) ceres_stop = true; cost = c1; r0 = r1; lambda = std::max(lambda * 0.3, 1e-12); stepped = true; if (tiny) { it = opt.max_num_iterations; sum->termination_type = CONVERGENCE; } }
            else { restore(saved); lambda *= 10; }
        }
        sum->iterations = it + 1;
        if (!stepped) { sum->termination_type = CONVERGENCE; break; }
    }
    // (out of iterations on a cost that still falls in its leading digits: NO_CONVERGENCE stands, and the reference's second test -
    // final_cost - decides; falling only below Ceres' function tolerance: Ceres would have stopped there and said CONVERGENCE)
    if (ceres_stop) sum->termination_type = CONVERGENCE;
    sum->final_cost = cost;
}
}  // namespace ceres
