// TEST INFRASTRUCTURE (authoring container): the ten trait methods of HipOperations<P> against the reference's
// BasicOperations<P> (operations/BasicOperations.h:20-39, semantics BasicOperations.tcc:6-216) on random vectors of the
// sizes the call sites use (4, 20, and nPos*nCodeSize for vector_add), including the in-place aliasing of
// NJ.tcc:857, 2035, 2389 and all four fastexp levels.  Bit-exact comparison.  Prints "ok <checks>" or the mismatches.
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>

#include "Utils.h"
#include "operations/BasicOperations.h"
#include "HipOperations.h"

static uint64_t rngState = 0x1234567ull;

static double unit() {
    uint64_t z = (rngState += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double) (z >> 11) * (1.0 / 9007199254740992.0);
}

static long checks = 0, bad = 0;

template<typename P>
static void same(const char *what, const P *a, const P *b, int64_t n) {
    checks++;
    if (memcmp(a, b, (size_t) n * sizeof(P)) != 0) {
        bad++;
        for (int64_t i = 0; i < n; i++)
            if (memcmp(a + i, b + i, sizeof(P)) != 0) {
                printf("MISMATCH %s<%s>[%ld]: %.17g vs %.17g\n", what, sizeof(P) == 4 ? "float" : "double", (long) i, (double) a[i], (double) b[i]);
                break;
            }
    }
}

template<typename P>
static void run() {
    veryfasttree::BasicOperations<P> ref;
    veryfasttree::HipOperations<P> hip;
    const int64_t sizes[] = {4, 20, 24, 400, 4000};
    for (int rep = 0; rep < 200; rep++)
        for (int64_t n: sizes) {
            std::vector<P> a(n), b(n), c(n), o1(n), o2(n);
            for (int64_t i = 0; i < n; i++) {
                a[i] = (P) (unit() * 2 - 0.5);
                b[i] = (P) (unit() * 3 - 1.0);
                c[i] = (P) (unit() - 0.3);
            }
            ref.vector_multiply(a.data(), b.data(), n, o1.data());
            hip.vector_multiply(a.data(), b.data(), n, o2.data());
            same("vector_multiply", o1.data(), o2.data(), n);
            P r1 = ref.vector_multiply_sum(a.data(), b.data(), n), r2 = hip.vector_multiply_sum(a.data(), b.data(), n);
            same("vector_multiply_sum", &r1, &r2, 1);
            r1 = ref.vector_multiply3_sum(a.data(), b.data(), c.data(), n);
            r2 = hip.vector_multiply3_sum(a.data(), b.data(), c.data(), n);
            same("vector_multiply3_sum", &r1, &r2, 1);
            r1 = ref.vector_dot_product_rot(a.data(), b.data(), c.data(), n);
            r2 = hip.vector_dot_product_rot(a.data(), b.data(), c.data(), n);
            same("vector_dot_product_rot", &r1, &r2, 1);
            r1 = ref.vector_sum(a.data(), n);
            r2 = hip.vector_sum(a.data(), n);
            same("vector_sum", &r1, &r2, 1);
            o1 = a;
            o2 = a;
            ref.vector_add(o1.data(), b.data(), n);
            hip.vector_add(o2.data(), b.data(), n);
            same("vector_add", o1.data(), o2.data(), n);
            ref.vector_multiply_by(a.data(), b[0], n, o1.data());
            hip.vector_multiply_by(a.data(), b[0], n, o2.data());
            same("vector_multiply_by", o1.data(), o2.data(), n);
            o1 = a;
            o2 = a;   /* in place: normalizeFreq, NJ.tcc:857; posteriorProfile, NJ.tcc:2389 */
            ref.vector_multiply_by(o1.data(), c[1], n, o1.data());
            hip.vector_multiply_by(o2.data(), c[1], n, o2.data());
            same("vector_multiply_by(in place)", o1.data(), o2.data(), n);
            o1 = a;
            o2 = a;
            ref.vector_add_mult(o1.data(), b.data(), c[2], n);
            hip.vector_add_mult(o2.data(), b.data(), c[2], n);
            same("vector_add_mult", o1.data(), o2.data(), n);
            for (int lvl = 0; lvl < 4; lvl++) {   /* in place, as expEigenRates uses it (NJ.tcc:2035) */
                for (int64_t i = 0; i < n; i++) o1[i] = o2[i] = (P) (-unit() * 30.0 * (i % 3 == 0 ? 0.01 : 1.0));
                ref.fastexp(o1.data(), n, lvl);
                hip.fastexp(o2.data(), n, lvl);
                same(lvl == 0 ? "fastexp0" : lvl == 1 ? "fastexp1" : lvl == 2 ? "fastexp2" : "fastexp3", o1.data(), o2.data(), n);
            }
        }
    for (int rep = 0; rep < 1000; rep++) {
        P mat[4][4], mat8[4][8], v[4], o1[4], o2[4];
        for (int i = 0; i < 4; i++) {
            v[i] = (P) unit();
            for (int j = 0; j < 4; j++) mat[i][j] = mat8[i][j] = (P) (unit() * 2 - 1);
        }
        ref.template matrix_by_vector4<4>(mat, v, o1);
        hip.template matrix_by_vector4<4>(mat, v, o2);
        same("matrix_by_vector4<4>", o1, o2, 4);
        ref.template matrix_by_vector4<8>(mat8, v, o1);
        hip.template matrix_by_vector4<8>(mat8, v, o2);
        same("matrix_by_vector4<8>", o1, o2, 4);
    }
}

int main() {
    run<float>();
    run<double>();
    if (bad) {
        printf("FAILED %ld of %ld checks\n", bad, checks);
        return 1;
    }
    printf("ok %ld\n", checks);
    return 0;
}
