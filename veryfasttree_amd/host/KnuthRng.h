// Knuth's "ran_array" (TAOCP vol. 2, 3rd ed., section 3.6; rng.c of 2002): x[n] = (x[n-100] - x[n-37]) mod 2^30.
// The reference draws its bootstrap columns from this generator as 2^-30 * ran_arr_next() (src/Knuth.cpp:95-111) and
// never seeds it, so its stream always starts from the routine's default seed 314159.  Written from the published
// algorithm; tests/test_abi_cpu.py pins it (through vft_knuth_stream) against 5000 values of the reference's stream.
#ifndef VFT_HOST_KNUTHRNG_H
#define VFT_HOST_KNUTHRNG_H

namespace veryfasttree {
    class KnuthRng {
    public:
        explicit KnuthRng(long seed = 314159L) { start(seed); }

        /* uniform in [0, 1): the reference's knuth_rand() */
        double rand() { return 9.31322574615479e-10 * (double) next(); }

        long next() {
            if (pos >= 0 && buf[pos] >= 0) return buf[pos++];
            array(buf, QUALITY);
            buf[KK] = -1;   /* only the first KK values of a refill are handed out */
            pos = 1;
            return buf[0];
        }

    private:
        static const int KK = 100, LL = 37, QUALITY = 1009, TT = 70;
        static const long MM = 1L << 30;
        long x[KK], buf[QUALITY];
        int pos;

        static long diff(long a, long b) { return (a - b) & (MM - 1); }

        void array(long *aa, int n) {
            int i, j;
            for (j = 0; j < KK; j++) aa[j] = x[j];
            for (; j < n; j++) aa[j] = diff(aa[j - KK], aa[j - LL]);
            for (i = 0; i < LL; i++, j++) x[i] = diff(aa[j - KK], aa[j - LL]);
            for (; i < KK; i++, j++) x[i] = diff(aa[j - KK], x[i - LL]);
        }

        void start(long seed) {
            long t[KK + KK - 1];
            long ss = (seed + 2) & (MM - 2);
            for (int j = 0; j < KK; j++) {
                t[j] = ss;
                ss <<= 1;
                if (ss >= MM) ss -= MM - 2;
            }
            t[1]++;
            int left = TT - 1;
            for (ss = seed & (MM - 1); left;) {
                for (int j = KK - 1; j > 0; j--) {
                    t[j + j] = t[j];
                    t[j + j - 1] = 0;
                }
                for (int j = KK + KK - 2; j >= KK; j--) {
                    t[j - (KK - LL)] = diff(t[j - (KK - LL)], t[j]);
                    t[j - KK] = diff(t[j - KK], t[j]);
                }
                if (ss & 1) {
                    for (int j = KK; j > 0; j--) t[j] = t[j - 1];
                    t[0] = t[KK];
                    t[LL] = diff(t[LL], t[KK]);
                }
                if (ss) ss >>= 1;
                else left--;
            }
            for (int j = 0; j < LL; j++) x[j + KK - LL] = t[j];
            for (int j = LL; j < KK; j++) x[j - LL] = t[j];
            for (int j = 0; j < 10; j++) array(t, KK + KK - 1);
            pos = -1;
        }
    };
}
#endif
