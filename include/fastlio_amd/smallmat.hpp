// smallmat.hpp -- a minimal fixed-size dense matrix layer for the host side of the IEKF.
// The reference uses Eigen (absent from this build environment); only what
// esekf::update_iterated_dyn_share_modified / predict need is provided: fixed-size storage,
// products, transposes, blocks and a partial-pivot LU inverse (Eigen's inverse() for n > 4).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace fastlio_amd {

template <int R, int C>
struct Mat {
    double a[R * C];
    static constexpr int Rows = R, Cols = C;
    Mat() { std::memset(a, 0, sizeof(a)); }
    static Mat Zero() { return Mat(); }
    static Mat Identity() {
        Mat m;
        for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
        return m;
    }
    double& operator()(int i, int j) { return a[i * C + j]; }
    const double& operator()(int i, int j) const { return a[i * C + j]; }
    double& operator[](int i) { return a[i]; }  // vectors
    const double& operator[](int i) const { return a[i]; }
    Mat<C, R> transpose() const {
        Mat<C, R> t;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j) t(j, i) = (*this)(i, j);
        return t;
    }
    Mat operator+(const Mat& o) const {
        Mat r;
        for (int i = 0; i < R * C; ++i) r.a[i] = a[i] + o.a[i];
        return r;
    }
    Mat operator-(const Mat& o) const {
        Mat r;
        for (int i = 0; i < R * C; ++i) r.a[i] = a[i] - o.a[i];
        return r;
    }
    Mat operator-() const {
        Mat r;
        for (int i = 0; i < R * C; ++i) r.a[i] = -a[i];
        return r;
    }
    Mat operator*(double s) const {
        Mat r;
        for (int i = 0; i < R * C; ++i) r.a[i] = a[i] * s;
        return r;
    }
    Mat operator/(double s) const {
        Mat r;
        for (int i = 0; i < R * C; ++i) r.a[i] = a[i] / s;
        return r;
    }
    Mat& operator+=(const Mat& o) {
        for (int i = 0; i < R * C; ++i) a[i] += o.a[i];
        return *this;
    }
    template <int K>
    Mat<R, K> operator*(const Mat<C, K>& o) const {
        Mat<R, K> r;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < K; ++j) {
                double s = 0;
                for (int k = 0; k < C; ++k) s += (*this)(i, k) * o(k, j);
                r(i, j) = s;
            }
        return r;
    }
    template <int BR, int BC>
    Mat<BR, BC> block(int r0, int c0) const {
        Mat<BR, BC> b;
        for (int i = 0; i < BR; ++i)
            for (int j = 0; j < BC; ++j) b(i, j) = (*this)(r0 + i, c0 + j);
        return b;
    }
    template <int BR, int BC>
    void set_block(int r0, int c0, const Mat<BR, BC>& b) {
        for (int i = 0; i < BR; ++i)
            for (int j = 0; j < BC; ++j) (*this)(r0 + i, c0 + j) = b(i, j);
    }
    double norm() const {
        double s = 0;
        for (int i = 0; i < R * C; ++i) s += a[i] * a[i];
        return std::sqrt(s);
    }
    double squaredNorm() const {
        double s = 0;
        for (int i = 0; i < R * C; ++i) s += a[i] * a[i];
        return s;
    }
};
template <int N>
using Vec = Mat<N, 1>;
typedef Vec<3> V3;
typedef Mat<3, 3> M3;

inline M3 hat(const V3& v) {  // mtkmath.hpp:176-183
    M3 r;
    r(0, 1) = -v[2]; r(0, 2) = v[1];
    r(1, 0) = v[2]; r(1, 2) = -v[0];
    r(2, 0) = -v[1]; r(2, 1) = v[0];
    return r;
}

// Dense n x n inverse, partial-pivot LU (row-major).  Stand-in for Eigen's inverse()
// (esekfom.hpp:1738,1782,1802).  The identity is carried through the elimination and the two triangular
// solves row-wise, so every inner loop runs over contiguous memory (the 23x23 case is called twice per IEKF
// pass and sits on the critical path between two GPU launches).  Returns false if a zero pivot was met.
inline bool inverse_lu(const double* A, int n, double* Ainv) {
    constexpr int kStack = 32;
    double lu_s[kStack * kStack], x_s[kStack * kStack];
    std::vector<double> lu_h, x_h;
    double* LU = lu_s;
    double* X = x_s;  // starts as the row-permuted identity, ends as the inverse
    if (n > kStack) {
        lu_h.resize((size_t)n * n);
        x_h.resize((size_t)n * n);
        LU = lu_h.data();
        X = x_h.data();
    }
    std::memcpy(LU, A, sizeof(double) * (size_t)n * n);
    std::memset(X, 0, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) X[i * n + i] = 1.0;
    bool ok = true;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(LU[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            const double v = std::fabs(LU[i * n + k]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) ok = false;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(LU[k * n + j], LU[p * n + j]);
            for (int j = 0; j < n; ++j) std::swap(X[k * n + j], X[p * n + j]);
        }
        const double piv = LU[k * n + k];
        const double* rk = LU + k * n;
        const double* xk = X + k * n;
        for (int i = k + 1; i < n; ++i) {
            const double l = LU[i * n + k] / piv;
            LU[i * n + k] = l;
            double* ri = LU + i * n;
            for (int j = k + 1; j < n; ++j) ri[j] -= l * rk[j];
            double* xi = X + i * n;  // forward substitution L y = P I, applied on the fly
            for (int j = 0; j < n; ++j) xi[j] -= l * xk[j];
        }
    }
    for (int i = n - 1; i >= 0; --i) {  // back substitution U x = y, all right-hand sides at once
        double* xi = X + i * n;
        for (int r = i + 1; r < n; ++r) {
            const double u = LU[i * n + r];
            const double* xr = X + r * n;
            for (int j = 0; j < n; ++j) xi[j] -= u * xr[j];
        }
        const double inv = 1.0 / LU[i * n + i];
        for (int j = 0; j < n; ++j) xi[j] *= inv;
    }
    std::memcpy(Ainv, X, sizeof(double) * (size_t)n * n);
    return ok;
}
// The same elimination with compile-time size and rows padded to a multiple of 8 doubles: every inner loop has a
// constant trip count over aligned, contiguous memory and vectorises fully (AVX2: 6 vectors per 23-wide row).  Each
// element sees exactly the operations of inverse_lu() in the same order (the padding columns stay zero and are never
// read back), so the two agree bit for bit; tests/test_host_iekf.py pins the filter built on it to the oracle's LU.
template <int N>
inline bool inverse_lu_fixed(const double* A, double* Ainv) {
    constexpr int S = (N + 7) & ~7;
    alignas(64) double LU[N * S];
    alignas(64) double X[N * S];
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) { LU[i * S + j] = A[i * N + j]; X[i * S + j] = 0.0; }
        for (int j = N; j < S; ++j) { LU[i * S + j] = 0.0; X[i * S + j] = 0.0; }
        X[i * S + i] = 1.0;
    }
    bool ok = true;
    for (int k = 0; k < N; ++k) {
        int p = k;
        double best = std::fabs(LU[k * S + k]);
        for (int i = k + 1; i < N; ++i) {
            const double v = std::fabs(LU[i * S + k]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) ok = false;
        if (p != k) {
            for (int j = 0; j < S; ++j) { const double t = LU[k * S + j]; LU[k * S + j] = LU[p * S + j]; LU[p * S + j] = t; }
            for (int j = 0; j < S; ++j) { const double t = X[k * S + j]; X[k * S + j] = X[p * S + j]; X[p * S + j] = t; }
        }
        const double piv = LU[k * S + k];
        const double* rk = LU + k * S;
        const double* xk = X + k * S;
        for (int i = k + 1; i < N; ++i) {
            const double l = LU[i * S + k] / piv;
            double* ri = LU + i * S;
            // columns <= k of row i are final (the multipliers); updating all S columns keeps the loop branch-free and
            // vector-wide -- the entries left of the diagonal are overwritten with l below / never read again
            for (int j = 0; j < S; ++j) ri[j] -= l * rk[j];
            ri[k] = l;
            double* xi = X + i * S;
            for (int j = 0; j < S; ++j) xi[j] -= l * xk[j];
        }
    }
    for (int i = N - 1; i >= 0; --i) {
        double* xi = X + i * S;
        for (int r = i + 1; r < N; ++r) {
            const double u = LU[i * S + r];
            const double* xr = X + r * S;
            for (int j = 0; j < S; ++j) xi[j] -= u * xr[j];
        }
        const double inv = 1.0 / LU[i * S + i];
        for (int j = 0; j < S; ++j) xi[j] *= inv;
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) Ainv[i * N + j] = X[i * S + j];
    return ok;
}

// The first NC columns of A^-1 only (the information-form update reads just P_inv.block<23,12>(0,0),
// esekfom.hpp:1803-1806): same elimination, the identity carried with NC (padded) columns instead of N.  Column j of an
// inverse depends on no other column, so these are bit for bit the first NC columns of inverse_lu_fixed<N>().
template <int N, int NC>
inline bool inverse_lu_fixed_cols(const double* A, double* Acols /* N x NC, row-major */) {
    constexpr int S = (N + 7) & ~7;
    constexpr int SC = (NC + 7) & ~7;
    alignas(64) double LU[N * S];
    alignas(64) double X[N * SC];
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) LU[i * S + j] = A[i * N + j];
        for (int j = N; j < S; ++j) LU[i * S + j] = 0.0;
        for (int j = 0; j < SC; ++j) X[i * SC + j] = 0.0;
        if (i < NC) X[i * SC + i] = 1.0;
    }
    bool ok = true;
    for (int k = 0; k < N; ++k) {
        int p = k;
        double best = std::fabs(LU[k * S + k]);
        for (int i = k + 1; i < N; ++i) {
            const double v = std::fabs(LU[i * S + k]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) ok = false;
        if (p != k) {
            for (int j = 0; j < S; ++j) { const double t = LU[k * S + j]; LU[k * S + j] = LU[p * S + j]; LU[p * S + j] = t; }
            for (int j = 0; j < SC; ++j) { const double t = X[k * SC + j]; X[k * SC + j] = X[p * SC + j]; X[p * SC + j] = t; }
        }
        const double piv = LU[k * S + k];
        const double* rk = LU + k * S;
        const double* xk = X + k * SC;
        for (int i = k + 1; i < N; ++i) {
            const double l = LU[i * S + k] / piv;
            double* ri = LU + i * S;
            for (int j = 0; j < S; ++j) ri[j] -= l * rk[j];
            ri[k] = l;
            double* xi = X + i * SC;
            for (int j = 0; j < SC; ++j) xi[j] -= l * xk[j];
        }
    }
    for (int i = N - 1; i >= 0; --i) {
        double* xi = X + i * SC;
        for (int r = i + 1; r < N; ++r) {
            const double u = LU[i * S + r];
            const double* xr = X + r * SC;
            for (int j = 0; j < SC; ++j) xi[j] -= u * xr[j];
        }
        const double inv = 1.0 / LU[i * S + i];
        for (int j = 0; j < SC; ++j) xi[j] *= inv;
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < NC; ++j) Acols[i * NC + j] = X[i * SC + j];
    return ok;
}
template <int N, int NC>
inline Mat<N, NC> inverse_cols(const Mat<N, N>& A) {
    Mat<N, NC> r;
    inverse_lu_fixed_cols<N, NC>(A.a, r.a);
    return r;
}

// Solves A X = B (A: N x N, B and X: N x NC, all row-major) by the same partial-pivot elimination, the right-hand sides carried
// along instead of an identity: constant trip counts over aligned rows padded to a multiple of 8 doubles.
template <int N, int NC>
inline bool solve_lu_fixed(const double* A, const double* B, double* Xout) {
    constexpr int S = (N + 7) & ~7;
    constexpr int SC = (NC + 7) & ~7;
    alignas(64) double LU[N * S];
    alignas(64) double X[N * SC];
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) LU[i * S + j] = A[i * N + j];
        for (int j = N; j < S; ++j) LU[i * S + j] = 0.0;
        for (int j = 0; j < NC; ++j) X[i * SC + j] = B[i * NC + j];
        for (int j = NC; j < SC; ++j) X[i * SC + j] = 0.0;
    }
    bool ok = true;
    for (int k = 0; k < N; ++k) {
        int p = k;
        double best = std::fabs(LU[k * S + k]);
        for (int i = k + 1; i < N; ++i) {
            const double v = std::fabs(LU[i * S + k]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) ok = false;
        if (p != k) {
            for (int j = 0; j < S; ++j) { const double t = LU[k * S + j]; LU[k * S + j] = LU[p * S + j]; LU[p * S + j] = t; }
            for (int j = 0; j < SC; ++j) { const double t = X[k * SC + j]; X[k * SC + j] = X[p * SC + j]; X[p * SC + j] = t; }
        }
        const double piv = LU[k * S + k];
        const double* rk = LU + k * S;
        const double* xk = X + k * SC;
        for (int i = k + 1; i < N; ++i) {
            const double l = LU[i * S + k] / piv;
            double* ri = LU + i * S;
            for (int j = 0; j < S; ++j) ri[j] -= l * rk[j];
            ri[k] = l;
            double* xi = X + i * SC;
            for (int j = 0; j < SC; ++j) xi[j] -= l * xk[j];
        }
    }
    for (int i = N - 1; i >= 0; --i) {
        double* xi = X + i * SC;
        for (int r = i + 1; r < N; ++r) {
            const double u = LU[i * S + r];
            const double* xr = X + r * SC;
            for (int j = 0; j < SC; ++j) xi[j] -= u * xr[j];
        }
        const double inv = 1.0 / LU[i * S + i];
        for (int j = 0; j < SC; ++j) xi[j] *= inv;
    }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < NC; ++j) Xout[i * NC + j] = X[i * SC + j];
    return ok;
}

template <int N>
inline Mat<N, N> inverse(const Mat<N, N>& A) {
    Mat<N, N> r;
    if (N >= 8) inverse_lu_fixed<N>(A.a, r.a);
    else inverse_lu(A.a, N, r.a);
    return r;
}

}  // namespace fastlio_amd
