// The generalised time-reversible nucleotide model as the reference sets it up (TransitionMatrix.tcc:26-60 createGTR,
// :160-232 createTransitionMatrix): rate matrix from six exchange rates and four frequencies, symmetrised with the square
// roots of the stationary distribution, eigen-decomposed (Householder tridiagonalisation + implicit QL, the classic
// tred2 / tqli pair, TransitionMatrix.tcc:368-520), and turned into the tables the likelihood kernels take
// (vft_set_transition_matrix): stat, 1/stat, eigenvalues, codeFreq (rows = codes in eigen-space, plus the gap row = their
// sum), eigeninv and its transpose.  Everything in double, narrowed by the caller.  Plain C++11.
#ifndef VFT_GTR_MODEL_H
#define VFT_GTR_MODEL_H

#include <cmath>
#include <stdexcept>

namespace veryfasttree {

    template<int N>
    struct TransitionTables {
        double stat[N], statinv[N], eigenval[N];
        double codeFreq[N + 1][N];   /* row N = NOCODE */
        double eigeninv[N][N], eigeninvT[N][N];
    };
    typedef TransitionTables<4> TransitionTables4;

    namespace gtr_detail {
        inline double hypot2(double a, double b) {   /* sqrt(a^2 + b^2) without overflow, as the reference's pythag */
            const double absa = std::fabs(a), absb = std::fabs(b);
            if (absa > absb) return absa * std::sqrt(1 + (absb / absa) * (absb / absa));
            return absb == 0 ? 0 : absb * std::sqrt(1 + (absa / absb) * (absa / absb));
        }

        /* Householder reduction of the symmetric n x n matrix held column-major in a (element (i, j) at a[j * n + i]) to
           tridiagonal form; on return a holds the orthogonal transformation, d the diagonal, e the sub-diagonal */
        inline void tridiagonalise(double *a, int n, double *d, double *e) {
            auto A = [&](int i, int j) -> double & { return a[j * n + i]; };
            for (int i = n - 1; i > 0; i--) {
                const int l = i - 1;
                double h = 0, scale = 0;
                if (l > 0) {
                    for (int k = 0; k <= l; k++) scale += std::fabs(A(i, k));
                    if (scale == 0) {
                        e[i] = A(i, l);
                    } else {
                        for (int k = 0; k <= l; k++) {
                            A(i, k) /= scale;
                            h += A(i, k) * A(i, k);
                        }
                        double f = A(i, l);
                        double g = -std::sqrt(h);
                        if (f < 0) g = -g;
                        e[i] = scale * g;
                        h -= f * g;
                        A(i, l) = f - g;
                        f = 0;
                        for (int j = 0; j <= l; j++) {
                            A(j, i) = A(i, j) / h;
                            g = 0;
                            for (int k = 0; k <= j; k++) g += A(j, k) * A(i, k);
                            for (int k = j + 1; k <= l; k++) g += A(k, j) * A(i, k);
                            e[j] = g / h;
                            f += e[j] * A(i, j);
                        }
                        const double hh = f / (h + h);
                        for (int j = 0; j <= l; j++) {
                            f = A(i, j);
                            g = e[j] - hh * f;
                            e[j] = g;
                            for (int k = 0; k <= j; k++) A(j, k) -= f * e[k] + g * A(i, k);
                        }
                    }
                } else {
                    e[i] = A(i, l);
                }
                d[i] = h;
            }
            d[0] = 0;
            e[0] = 0;
            for (int i = 0; i < n; i++) {
                const int l = i - 1;
                if (d[i] != 0) {
                    for (int j = 0; j <= l; j++) {
                        double g = 0;
                        for (int k = 0; k <= l; k++) g += A(i, k) * A(k, j);
                        for (int k = 0; k <= l; k++) A(k, j) -= g * A(k, i);
                    }
                }
                d[i] = A(i, i);
                A(i, i) = 1;
                for (int j = 0; j <= l; j++) A(i, j) = A(j, i) = 0;
            }
        }

        /* implicit QL on the tridiagonal matrix (d, e); z (column-major, the output of tridiagonalise) is rotated into
           the eigenvectors, d becomes the eigenvalues */
        inline void implicitQL(double *d, double *e, int n, double *z) {
            auto Z = [&](int i, int j) -> double & { return z[j * n + i]; };
            for (int i = 1; i < n; i++) e[i - 1] = e[i];
            e[n - 1] = 0;
            for (int l = 0; l < n; l++) {
                int iter = 0;
                for (;;) {
                    int m = l;
                    for (; m < n - 1; m++) {
                        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                        if (std::fabs(e[m]) + dd == dd) break;
                    }
                    if (m == l) break;
                    if (++iter > 30) throw std::runtime_error("GTR eigen-decomposition did not converge");
                    double g = (d[l + 1] - d[l]) / (2 * e[l]);
                    double r = hypot2(g, 1.);
                    g = d[m] - d[l] + e[l] / (g + (g < 0 ? -r : r));
                    double s = 1, c = 1, p = 0;
                    bool restart = false;
                    for (int i = m - 1; i >= l; i--) {
                        double f = s * e[i];
                        const double b = c * e[i];
                        r = hypot2(f, g);
                        e[i + 1] = r;
                        if (r == 0) {
                            d[i + 1] -= p;
                            e[m] = 0;
                            restart = true;
                            break;
                        }
                        s = f / r;
                        c = g / r;
                        g = d[i + 1] - p;
                        r = (d[i] - g) * s + 2 * c * b;
                        p = s * r;
                        d[i + 1] = g + p;
                        g = c * r - b;
                        for (int k = 0; k < n; k++) {
                            f = Z(k, i + 1);
                            Z(k, i + 1) = s * Z(k, i) + c * f;
                            Z(k, i) = c * Z(k, i) - s * f;
                        }
                    }
                    if (restart) continue;
                    d[l] -= p;
                    e[l] = g;
                    e[m] = 0;
                }
            }
        }
    }

    /* createTransitionMatrix (TransitionMatrix.tcc:158-232) for any alphabet: matrix[i][j] off-diagonal rates (the
       diagonal is ignored and set so that columns sum to 0), stat the stationary distribution.  REAL = numeric_t: the
       reference stores its tables in numeric_t and sums the gap row from the stored values in numeric_t
       (TransitionMatrix.tcc:218-226), so the tables come back already narrowed (held in double) */
    template<typename REAL, int N>
    inline void createTransitionTables(const double matrix[N][N], const double stat[N], TransitionTables<N> &t) {
        const int n = N;
        double sqrtstat[N];
        for (int i = 0; i < n; i++) {
            t.stat[i] = stat[i];
            t.statinv[i] = 1.0 / stat[i];
            sqrtstat[i] = std::sqrt(stat[i]);
        }
        double sym[N * N];
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) sym[n * i + j] = matrix[i][j];
        for (int j = 0; j < n; j++) {   /* diagonals so that the columns sum to 0 */
            double sum = 0;
            sym[n * j + j] = 0;
            for (int i = 0; i < n; i++) sum += sym[n * i + j];
            sym[n * j + j] = -sum;
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) sym[n * i + j] *= sqrtstat[j] / sqrtstat[i];
        double w[N * N], eval[N], e[N];
        for (int i = 0; i < n * n; i++) w[i] = sym[i];
        gtr_detail::tridiagonalise(w, n, eval, e);
        gtr_detail::implicitQL(eval, e, n, w);
        for (int i = 0; i < n; i++) {
            t.stat[i] = (double) (REAL) t.stat[i];
            t.statinv[i] = (double) (REAL) t.statinv[i];
            t.eigenval[i] = (double) (REAL) eval[i];
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                t.eigeninv[i][j] = (double) (REAL) (w[n * i + j] / sqrtstat[j]);
                t.eigeninvT[j][i] = t.eigeninv[i][j];
            }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) t.codeFreq[i][j] = (double) (REAL) (w[j * n + i] * sqrtstat[i]);
        for (int j = 0; j < n; j++) {
            REAL acc = 0;
            for (int i = 0; i < n; i++) acc += (REAL) t.codeFreq[i][j];
            t.codeFreq[n][j] = (double) acc;
        }
    }

    /* createGTR (TransitionMatrix.tcc:26-60).  rates: ac ag at cg ct gt; freq: A C G T */
    template<typename REAL>
    inline void createGTR(const double rates[6], const double freq[4], TransitionTables4 &t) {
        const int n = 4;
        double matrix[4][4];
        int im = 0;
        for (int i = 0; i < n; i++) {
            matrix[i][i] = 0;
            for (int j = i + 1; j < n; j++) {
                const double rate = rates[im++];
                if (!(rate > 0)) throw std::invalid_argument("GTR rates must be positive");
                matrix[i][j] = rate * freq[i];   /* so that the stationary distribution stays freq */
                matrix[j][i] = rate * freq[j];
            }
        }
        double total = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) total += freq[i] * matrix[i][j];
        const double inv = 1.0 / total;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) matrix[i][j] *= inv;
        createTransitionTables<REAL, 4>(matrix, freq, t);
    }

}

#endif
