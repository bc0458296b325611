/*
 * TEST INFRASTRUCTURE — body of the CPU restatement, included once per precision by vft_oracle.c with
 *   REAL  = float | double          (the reference's numeric_t)
 *   FN(x) = x##_f32 | x##_f64
 *
 * Every function restates one reference member of NeighbourJoining<Precision,Operations>
 * (/root/reference/src/NeighbourJoining.tcc, "NJ.tcc" below) on DENSE profiles:
 *     w[nPos]  weights, c[nPos] codes (NOCODE = 127), f[nPos*nCodes] frequency vector of column i,
 *     meaningful only where the reference would hold a vector (c == NOCODE && w > 0, NJ.tcc:2040-2042).
 * The reference's arithmetic types are followed literally (which products are formed in numeric_t and
 * which in double) because join order depends on the low bits; comments name the line being mirrored.
 *
 * Backend-dependent reductions are pinned to the backends the reference's dispatcher really selects
 * (VeryFastTree.cpp:46-66): SSE128 for float, AVX256 for double.  Both reduce a dot product with four
 * strided lane accumulators and finish with (s0+s1)+(s2+s3) (SSE128Operations.tcc:14-20,70-80;
 * AVX256Operations.tcc:14-18,58-70) — FN(red4_*) below.
 */

/* ---- backend reductions (n % 4 == 0: nCodes is 4 or 20) */
static REAL FN(red4_mul)(const REAL *a, const REAL *b, int n) { /* vector_multiply_sum */
    REAL s[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i += 4)
        for (int l = 0; l < 4; l++) {
            REAL p = a[i + l] * b[i + l];
            s[l] = p + s[l];
        }
    REAL lo = s[0] + s[1], hi = s[2] + s[3];
    return lo + hi;
}

static REAL FN(red4_mul3)(const REAL *a, const REAL *b, const REAL *c, int n) { /* vector_multiply3_sum */
    REAL s[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i += 4)
        for (int l = 0; l < 4; l++) {
            REAL p = a[i + l] * b[i + l];
            p = p * c[i + l];
            s[l] = p + s[l];
        }
    REAL lo = s[0] + s[1], hi = s[2] + s[3];
    return lo + hi;
}

static REAL FN(red4_sum)(const REAL *a, int n) { /* vector_sum */
    REAL s[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i += 4)
        for (int l = 0; l < 4; l++) s[l] = a[i + l] + s[l];
    REAL lo = s[0] + s[1], hi = s[2] + s[3];
    return lo + hi;
}

/* ---- a2: seqDist, NJ.tcc:1601-1624.  distances == NULL -> %-different */
void FN(vfto_seqdist)(const unsigned char *c1, const unsigned char *c2, int64_t nPos, int nCodes,
                      const REAL *distances, REAL *dist, REAL *weight) {
    double top = 0;
    int64_t nUse = 0;
    if (!distances) {
        int nDiff = 0;
        for (int64_t i = 0; i < nPos; i++)
            if (c1[i] != VFTO_NOCODE && c2[i] != VFTO_NOCODE) {
                nUse++;
                if (c1[i] != c2[i]) nDiff++;
            }
        top = (double) nDiff;
    } else {
        for (int64_t i = 0; i < nPos; i++)
            if (c1[i] != VFTO_NOCODE && c2[i] != VFTO_NOCODE) {
                nUse++;
                top += distances[(int) c1[i] * nCodes + c2[i]];
            }
    }
    *weight = (REAL) (double) nUse;
    *dist = (REAL) (nUse > 0 ? top / (double) nUse : 1.0);
}

/* ---- a3: profileDistPiece, NJ.tcc:900-941.  f == NULL means "no vector at this column". */
static double FN(piece)(int code1, int code2, const REAL *f1, const REAL *f2, const REAL *codeDist2, int nCodes,
                        const FN(vfto_dmat) *dm) {
    if (dm) {
        if (code1 != VFTO_NOCODE && code2 != VFTO_NOCODE) return dm->distances[code1 * nCodes + code2];
        if (codeDist2 && code1 != VFTO_NOCODE) return codeDist2[code1];
        if (!f1) {
            if (code1 == VFTO_NOCODE) return 10.0;
            f1 = dm->codefreq + code1 * nCodes;
        }
        if (!f2) {
            if (code2 == VFTO_NOCODE) return 10.0;
            f2 = dm->codefreq + code2 * nCodes;
        }
        return FN(red4_mul3)(f1, f2, dm->eigenval, nCodes);
    }
    if (code1 != VFTO_NOCODE) {
        if (code2 != VFTO_NOCODE) return code1 == code2 ? 0.0 : 1.0;
        if (!f2) return 10.0;
        return 1.0 - f2[code1];
    }
    if (code2 != VFTO_NOCODE) {
        if (!f1) return 10.0;
        return 1.0 - f1[code2];
    }
    if (!f1 || !f2) return 10.0;
    double piece = 1.0;
    for (int k = 0; k < nCodes; k++) {
        REAL p = f1[k] * f2[k]; /* numeric_t product, NJ.tcc:935 */
        piece -= p;
    }
    return piece;
}

#define HASVEC(w, c, i) ((w)[i] > 0 && (c)[i] == VFTO_NOCODE)

/* ---- a3: profileDist, NJ.tcc:1167-1190.  cd2 = codeDist of profile 2 (out-profile with matrix) or NULL */
void FN(vfto_profiledist)(const REAL *w1, const unsigned char *c1, const REAL *f1, const REAL *w2,
                          const unsigned char *c2, const REAL *f2, const REAL *cd2, int64_t nPos, int nCodes,
                          const FN(vfto_dmat) *dm, REAL *dist, REAL *weight) {
    double top = 0, denom = 0;
    for (int64_t i = 0; i < nPos; i++) {
        if (w1[i] > 0 && w2[i] > 0) {
            REAL ww = w1[i] * w2[i]; /* numeric_t product widened afterwards, NJ.tcc:1176 */
            double wgt = ww;
            denom += wgt;
            double piece = FN(piece)(c1[i], c2[i], HASVEC(w1, c1, i) ? f1 + i * nCodes : NULL,
                                     HASVEC(w2, c2, i) ? f2 + i * nCodes : NULL, cd2 ? cd2 + i * nCodes : NULL,
                                     nCodes, dm);
            top += wgt * piece;
        }
    }
    *weight = (REAL) (denom > 0 ? denom : 0.01);
    *dist = (REAL) (denom > 0 ? top / denom : 1);
}

/* ---- addToFreq, NJ.tcc:821-833 (weight arrives as double, is narrowed by vector_add_mult's signature) */
static void FN(add_to_freq)(REAL *fOut, double weight, int codeIn, const REAL *fIn, int nCodes,
                            const FN(vfto_dmat) *dm) {
    if (fIn) {
        REAL wr = (REAL) weight;
        for (int k = 0; k < nCodes; k++) {
            REAL p = fIn[k] * wr;
            fOut[k] = fOut[k] + p;
        }
    } else if (dm) {
        REAL wr = (REAL) weight;
        const REAL *cf = dm->codefreq + codeIn * nCodes;
        for (int k = 0; k < nCodes; k++) {
            REAL p = cf[k] * wr;
            fOut[k] = fOut[k] + p;
        }
    } else {
        fOut[codeIn] = (REAL) ((double) fOut[codeIn] + weight);
    }
}

/* ---- normalizeFreq, NJ.tcc:843-871 */
static void FN(normalize_freq)(REAL *freq, int nCodes, const FN(vfto_dmat) *dm, double fPostTotalTolerance) {
    double total = 0;
    if (dm) {
        total = FN(red4_mul)(freq, dm->eigentot, nCodes);
    } else {
        for (int k = 0; k < nCodes; k++) total += freq[k];
    }
    if (total > fPostTotalTolerance) {
        REAL inv = (REAL) (1.0 / total);
        for (int k = 0; k < nCodes; k++) freq[k] = freq[k] * inv;
    } else if (!dm) {
        for (int k = 0; k < nCodes; k++) freq[k] = (REAL) (1.0 / nCodes);
    } else {
        for (int k = 0; k < nCodes; k++) freq[k] = dm->codefreq[k];
    }
}

/* ---- setCodeDist, NJ.tcc:873-898 (profile is an out-profile: every column NOCODE with a vector) */
static void FN(set_code_dist)(const unsigned char *c, const REAL *w, const REAL *f, REAL *cd, int64_t nPos,
                              int nCodes, const FN(vfto_dmat) *dm) {
    for (int64_t i = 0; i < nPos; i++)
        for (int k = 0; k < nCodes; k++)
            cd[i * nCodes + k] = (REAL) FN(piece)(c[i], k, HASVEC(w, c, i) ? f + i * nCodes : NULL, NULL, NULL,
                                                  nCodes, dm);
}

/* ---- a8: averageProfile, NJ.tcc:2067-2135.  bionjWeight < 0 -> 0.5 */
void FN(vfto_average_profile)(REAL *wo, unsigned char *co, REAL *fo, const REAL *w1, const unsigned char *c1,
                              const REAL *f1, const REAL *w2, const unsigned char *c2, const REAL *f2,
                              int64_t nPos, int nCodes, double bionjWeight, const FN(vfto_dmat) *dm,
                              double fPostTotalTolerance) {
    if (bionjWeight < 0) bionjWeight = 0.5;
    for (int64_t i = 0; i < nPos; i++) {
        wo[i] = (REAL) (bionjWeight * w1[i] + (1 - bionjWeight) * w2[i]);
        co[i] = VFTO_NOCODE;
        REAL *f = fo + i * nCodes;
        for (int k = 0; k < nCodes; k++) f[k] = 0;
        if (wo[i] > 0) {
            if (w1[i] > 0 && c1[i] != VFTO_NOCODE && (w2[i] <= 0 || c1[i] == c2[i])) {
                co[i] = c1[i];
            } else if (w1[i] <= 0 && w2[i] > 0 && c2[i] != VFTO_NOCODE) {
                co[i] = c2[i];
            }
            if (co[i] == VFTO_NOCODE) {
                if (w1[i] > 0)
                    FN(add_to_freq)(f, w1[i] * bionjWeight, c1[i], HASVEC(w1, c1, i) ? f1 + i * nCodes : NULL,
                                    nCodes, dm);
                if (w2[i] > 0)
                    FN(add_to_freq)(f, w2[i] * (1.0 - bionjWeight), c2[i],
                                    HASVEC(w2, c2, i) ? f2 + i * nCodes : NULL, nCodes, dm);
                FN(normalize_freq)(f, nCodes, dm, fPostTotalTolerance);
            }
        }
    }
}

/* ---- a9: outProfile, NJ.tcc:729-815 at one thread: profiles are accumulated in list order.
   W/C/F are n packed dense profiles; cdo may be NULL when there is no distance matrix. */
void FN(vfto_out_profile)(REAL *wo, unsigned char *co, REAL *fo, REAL *cdo, const REAL *W, const unsigned char *C,
                          const REAL *F, int64_t n, int64_t nPos, int nCodes, const FN(vfto_dmat) *dm,
                          double fPostTotalTolerance) {
    double inweight = 1.0 / (double) n;
    for (int64_t i = 0; i < nPos; i++) {
        wo[i] = 0;
        for (int64_t p = 0; p < n; p++) wo[i] = (REAL) ((double) wo[i] + W[p * nPos + i] * inweight);
        if (wo[i] <= 0) wo[i] = (REAL) 1e-20;
        co[i] = VFTO_NOCODE;
        for (int k = 0; k < nCodes; k++) fo[i * nCodes + k] = 0;
    }
    for (int64_t p = 0; p < n; p++) {
        const REAL *w = W + p * nPos;
        const unsigned char *c = C + p * nPos;
        const REAL *f = F + p * nPos * nCodes;
        for (int64_t i = 0; i < nPos; i++)
            if (w[i] > 0)
                FN(add_to_freq)(fo + i * nCodes, w[i], c[i], HASVEC(w, c, i) ? f + i * nCodes : NULL, nCodes, dm);
    }
    for (int64_t i = 0; i < nPos; i++) FN(normalize_freq)(fo + i * nCodes, nCodes, dm, fPostTotalTolerance);
    if (dm && cdo) FN(set_code_dist)(co, wo, fo, cdo, nPos, nCodes, dm);
}

/* ---- a9: updateOutProfile, NJ.tcc:943-1010 (in place on the out-profile) */
void FN(vfto_update_out_profile)(REAL *wo, REAL *fo, REAL *cdo, const unsigned char *co, const REAL *w1,
                                 const unsigned char *c1, const REAL *f1, const REAL *w2,
                                 const unsigned char *c2, const REAL *f2, const REAL *wn,
                                 const unsigned char *cn, const REAL *fn, int64_t nActiveOld, int64_t nPos,
                                 int nCodes, const FN(vfto_dmat) *dm, double fPostTotalTolerance) {
    for (int64_t i = 0; i < nPos; i++) {
        REAL *f = fo + i * nCodes;
        REAL om = wo[i] * (REAL) nActiveOld; /* numeric_t * int64 is a numeric_t product, NJ.tcc:963 */
        double originalMult = om;
        double newMult = originalMult + wn[i] - w1[i] - w2[i];
        wo[i] = (REAL) (newMult / (nActiveOld - 1));
        if (wo[i] <= 0) wo[i] = (REAL) 1e-20;
        for (int k = 0; k < nCodes; k++) f[k] = (REAL) ((double) f[k] * originalMult);
        if (w1[i] > 0) {
            REAL neg = -w1[i];
            FN(add_to_freq)(f, neg, c1[i], HASVEC(w1, c1, i) ? f1 + i * nCodes : NULL, nCodes, dm);
        }
        if (w2[i] > 0) {
            REAL neg = -w2[i];
            FN(add_to_freq)(f, neg, c2[i], HASVEC(w2, c2, i) ? f2 + i * nCodes : NULL, nCodes, dm);
        }
        if (wn[i] > 0) FN(add_to_freq)(f, wn[i], cn[i], HASVEC(wn, cn, i) ? fn + i * nCodes : NULL, nCodes, dm);
        FN(normalize_freq)(f, nCodes, dm, fPostTotalTolerance);
    }
    if (dm && cdo) FN(set_code_dist)(co, wo, fo, cdo, nPos, nCodes, dm);
}

/* ---- a4: out-distance formula, NJ.tcc:1046-1053.  (dist, weight) = profileDist(node, outprofile).
   The products are numeric_t products in the reference (numeric_t * int64_t), only the division is double. */
REAL FN(vfto_out_distance)(REAL dist, REAL weight, int64_t nActive, REAL selfweight, REAL selfdist, REAL diameter,
                           double totdiam) {
    REAL t1 = dist * weight;
    REAL t2 = t1 * (REAL) nActive;
    REAL t3 = selfweight * selfdist;
    REAL t4 = t2 - t3;
    REAL topr = (REAL) (nActive - 1) * t4;
    REAL b1 = weight * (REAL) nActive;
    REAL botr = b1 - selfweight;
    double top = topr, bottom = botr;
    double pdistOutWithoutA = top / bottom;
    REAL dn = diameter * (REAL) (nActive - 1);
    double r = bottom > 0.01 ? pdistOutWithoutA - dn - (totdiam - diameter) : 3.0;
    return (REAL) r;
}

/* ---- a4: criterion with stale out-distance rescale, NJ.tcc:1099-1107 */
REAL FN(vfto_criterion)(REAL dist, REAL outI, int64_t nOutI, REAL outJ, int64_t nOutJ, int64_t nActive) {
    double oi = outI, oj = outJ;
    if (nOutI != nActive) oi *= (nActive - 1) / (double) (nOutI - 1);
    if (nOutJ != nActive) oj *= (nActive - 1) / (double) (nOutJ - 1);
    return (REAL) ((double) dist - (oi + oj) / (double) (nActive - 2));
}

/* ---- a5: setBestHit -> setDistCriterion -> setCriterion (-> setOutDistance), NJ.tcc:3571-3646, 1085-1124.
   State arrays are indexed by node id; W/C/F hold every node's dense profile (leaves: w in {0,1}, no vectors).
   outDist / nOutActive are updated in place exactly when the reference refreshes them lazily. */
static void FN(refresh_out)(int64_t v, int64_t nActive, const FN(vfto_state) *st, REAL *outDist,
                            int64_t *nOutActive) {
    if (nOutActive[v] == nActive) return; /* NJ.tcc:1013-1015 */
    REAL d, w;
    FN(vfto_profiledist)(st->W + v * st->nPos, st->C + v * st->nPos, st->F + v * st->nPos * st->nCodes, st->out_w,
                         st->out_c, st->out_f, st->out_cd, st->nPos, st->nCodes, st->dm, &d, &w);
    outDist[v] = FN(vfto_out_distance)(d, w, nActive, st->selfweight[v], st->selfdist[v], st->diameter[v],
                                       st->totdiam);
    nOutActive[v] = nActive;
}

void FN(vfto_set_dist_criterion)(const FN(vfto_state) *st, int64_t i, int64_t j, int64_t nActive,
                                 int64_t nDiffAllow, REAL *outDist, int64_t *nOutActive, REAL *dist, REAL *weight,
                                 REAL *crit) {
    REAL d, w;
    if (i < st->nSeqs && j < st->nSeqs) {
        FN(vfto_seqdist)(st->C + i * st->nPos, st->C + j * st->nPos, st->nPos, st->nCodes,
                         st->dm ? st->dm->distances : NULL, &d, &w);
    } else {
        FN(vfto_profiledist)(st->W + i * st->nPos, st->C + i * st->nPos, st->F + i * st->nPos * st->nCodes,
                             st->W + j * st->nPos, st->C + j * st->nPos, st->F + j * st->nPos * st->nCodes, NULL,
                             st->nPos, st->nCodes, st->dm, &d, &w);
        REAL dd = st->diameter[i] + st->diameter[j];
        d = d - dd; /* NJ.tcc:1120, numeric_t arithmetic */
    }
    *dist = d;
    *weight = w;
    if (nOutActive[i] - nActive > nDiffAllow) FN(refresh_out)(i, nActive, st, outDist, nOutActive);
    if (nOutActive[j] - nActive > nDiffAllow) FN(refresh_out)(j, nActive, st, outDist, nOutActive);
    *crit = FN(vfto_criterion)(d, outDist[i], nOutActive[i], outDist[j], nOutActive[j], nActive);
}

void FN(vfto_set_best_hit)(const FN(vfto_state) *st, int64_t node, int64_t nActive, int64_t nDiffAllow,
                           REAL *outDist, int64_t *nOutActive, int64_t *hit_i, int64_t *hit_j, REAL *hit_weight,
                           REAL *hit_dist, REAL *hit_crit, int64_t *best_j) {
    REAL bestc = (REAL) 1e20;
    *best_j = -1;
    for (int64_t j = 0; j < st->maxnode; j++) {
        hit_i[j] = node;
        hit_j[j] = j;
        if (st->parent[j] >= 0) {
            hit_i[j] = -1;
            hit_weight[j] = 0;
            hit_crit[j] = hit_dist[j] = (REAL) 1e20;
            continue;
        }
        FN(vfto_set_dist_criterion)(st, node, j, nActive, nDiffAllow, outDist, nOutActive, &hit_dist[j],
                                    &hit_weight[j], &hit_crit[j]);
        if (hit_crit[j] < bestc && node != j) {
            bestc = hit_crit[j];
            *best_j = j;
        }
    }
}

/* ---- a12: P(t) tables.  pSame/pDiff NJ.tcc:2005-2018; expEigenRates NJ.tcc:2020-2038 (NDEBUG branch, fastexp 0) */
void FN(vfto_psame_pdiff)(double length, const REAL *rates, int nRate, double *pSame, double *pDiff) {
    for (int r = 0; r < nRate; r++) {
        pSame[r] = 0.25 + 0.75 * exp((-4.0 / 3.0) * fabs(length * rates[r]));
        pDiff[r] = (1.0 - pSame[r]) / 3.0;
    }
}

void FN(vfto_exp_eigen_rates)(double length, const REAL *rates, int nRate, const REAL *eigenval, int nCodes,
                              double MLMinRelBranchLength, REAL *out) {
    for (int r = 0; r < nRate; r++) {
        double relLen = length * rates[r];
        if (relLen < MLMinRelBranchLength) relLen = MLMinRelBranchLength;
        REAL rl = (REAL) relLen; /* vector_multiply_by takes numeric_t */
        for (int j = 0; j < nCodes; j++) {
            REAL x = eigenval[j] * rl;
            out[r * nCodes + j] = (REAL) exp((double) x);
        }
    }
}

/* ---- a10: pairLogLk, NJ.tcc:1192-1447.  tm == NULL -> Jukes-Cantor.  site_lk may be NULL (multiplied in place). */
double FN(vfto_pair_loglk)(const REAL *w1, const unsigned char *c1, const REAL *f1, const REAL *w2,
                           const unsigned char *c2, const REAL *f2, int64_t nPos, int nCodes, double length,
                           const REAL *rates, int nRate, const int64_t *ratecat, const FN(vfto_tmat) *tm,
                           double MLMinRelBranchLength, double *site_lk) {
    double lk = 1.0, loglk = 0.0;
    if (!tm) {
        double pSame[VFTO_MAXRATES], pDiff[VFTO_MAXRATES];
        FN(vfto_psame_pdiff)(length, rates, nRate, pSame, pDiff);
        for (int64_t i = 0; i < nPos; i++) {
            int r = (int) ratecat[i];
            double wA = w1[i], wB = w2[i];
            int codeA = c1[i], codeB = c2[i];
            const REAL *fA = HASVEC(w1, c1, i) ? f1 + i * 4 : NULL;
            const REAL *fB = HASVEC(w2, c2, i) ? f2 + i * 4 : NULL;
            double lkAB = 0;
            if (!fA && !fB) {
                if (codeA == VFTO_NOCODE) lkAB = 0.25;
                else if (codeB == VFTO_NOCODE) lkAB = 0.25;
                else if (codeA == codeB) lkAB = pSame[r] * wA * wB + 0.25 * (1 - wA * wB);
                else lkAB = pDiff[r] * wA * wB + 0.25 * (1 - wA * wB);
            } else if (!fA) {
                if (codeA == VFTO_NOCODE) lkAB = 0.25;
                else lkAB = wA * (pDiff[r] + fB[codeA] * (pSame[r] - pDiff[r])) + (1.0 - wA) * 0.25;
            } else if (!fB) {
                if (codeB == VFTO_NOCODE) lkAB = 0.25;
                else lkAB = wB * (pDiff[r] + fA[codeB] * (pSame[r] - pDiff[r])) + (1.0 - wB) * 0.25;
            } else {
                for (int j = 0; j < 4; j++) {
                    REAL om = 1 - fA[j]; /* int - numeric_t is numeric_t, NJ.tcc:1253 */
                    lkAB += fB[j] * (fA[j] * pSame[r] + om * pDiff[r]);
                }
            }
            lk *= lkAB;
            while (lk < VFTO_LK_UNDERFLOW) {
                lk *= VFTO_LK_UNDERFLOW_INV;
                loglk -= VFTO_LOG_LK_UNDERFLOW;
            }
            if (site_lk) site_lk[i] *= lkAB;
        }
    } else {
        REAL expeig[VFTO_MAXRATES * VFTO_MAXCODES];
        FN(vfto_exp_eigen_rates)(length, rates, nRate, tm->eigenval, nCodes, MLMinRelBranchLength, expeig);
        const REAL *fGap = tm->codefreq + nCodes * nCodes; /* row nCodes = NOCODE */
        REAL fAmix[VFTO_MAXCODES], fBmix[VFTO_MAXCODES];
        for (int64_t i = 0; i < nPos; i++) {
            const REAL *ee = expeig + ratecat[i] * nCodes;
            double wA = w1[i], wB = w2[i];
            if (wA == 0 && wB == 0 && c1[i] == VFTO_NOCODE && c2[i] == VFTO_NOCODE) continue;
            const REAL *fA = HASVEC(w1, c1, i) ? f1 + i * nCodes : NULL;
            const REAL *fB = HASVEC(w2, c2, i) ? f2 + i * nCodes : NULL;
            if (!fA) fA = tm->codefreq + (c1[i] == VFTO_NOCODE ? nCodes : c1[i]) * nCodes;
            if (wA > 0.0 && wA < 1.0) {
                for (int j = 0; j < nCodes; j++) fAmix[j] = (REAL) (wA * fA[j] + (1.0 - wA) * fGap[j]);
                fA = fAmix;
            }
            if (!fB) fB = tm->codefreq + (c2[i] == VFTO_NOCODE ? nCodes : c2[i]) * nCodes;
            if (wB > 0.0 && wB < 1.0) {
                for (int j = 0; j < nCodes; j++) fBmix[j] = (REAL) (wB * fB[j] + (1.0 - wB) * fGap[j]);
                fB = fBmix;
            }
            double lkAB = 0;
            if (nCodes == 4) {
                for (int j = 0; j < 4; j++) {
                    REAL p = ee[j] * fA[j]; /* numeric_t triple product, NJ.tcc:1305 */
                    p = p * fB[j];
                    lkAB += p;
                }
            } else {
                lkAB = FN(red4_mul3)(ee, fA, fB, nCodes); /* NJ.tcc:1359 */
            }
            if (site_lk) site_lk[i] *= lkAB;
            lk *= lkAB;
            while (lk < VFTO_LK_UNDERFLOW) {
                lk *= VFTO_LK_UNDERFLOW_INV;
                loglk -= VFTO_LOG_LK_UNDERFLOW;
            }
            while (lk > VFTO_LK_UNDERFLOW_INV) {
                lk *= VFTO_LK_UNDERFLOW;
                loglk += VFTO_LOG_LK_UNDERFLOW;
            }
        }
    }
    loglk += log(lk);
    return loglk;
}

/* ---- a11: posteriorProfile, NJ.tcc:2137-2447 (exact ML: options.exactML is the default) */
void FN(vfto_posterior_profile)(REAL *wo, unsigned char *co, REAL *fo, const REAL *w1a, const unsigned char *c1,
                                const REAL *f1a, const REAL *w2a, const unsigned char *c2, const REAL *f2a,
                                int64_t nPos, int nCodes, double len1, double len2, const REAL *rates, int nRate,
                                const int64_t *ratecat, const FN(vfto_tmat) *tm, double MLMinBranchLength,
                                double MLMinRelBranchLength) {
    if (len1 < MLMinBranchLength) len1 = MLMinBranchLength;
    if (len2 < MLMinBranchLength) len2 = MLMinBranchLength;
    for (int64_t i = 0; i < nPos; i++) {
        co[i] = VFTO_NOCODE;
        wo[i] = 1.0;
        for (int k = 0; k < nCodes; k++) fo[i * nCodes + k] = 0;
    }
    if (!tm) {
        double PS1[VFTO_MAXRATES], PD1[VFTO_MAXRATES], PS2[VFTO_MAXRATES], PD2[VFTO_MAXRATES];
        FN(vfto_psame_pdiff)(len1, rates, nRate, PS1, PD1);
        FN(vfto_psame_pdiff)(len2, rates, nRate, PS2, PD2);
        REAL mix1[4], mix2[4];
        for (int64_t i = 0; i < nPos; i++) {
            int r = (int) ratecat[i];
            double w1 = w1a[i], w2 = w2a[i];
            int code1 = c1[i], code2 = c2[i];
            const REAL *f1 = HASVEC(w1a, c1, i) ? f1a + i * 4 : NULL;
            const REAL *f2 = HASVEC(w2a, c2, i) ? f2a + i * 4 : NULL;
            if (!f1 && !f2) {
                if (code1 == VFTO_NOCODE && code2 == VFTO_NOCODE) {
                    co[i] = VFTO_NOCODE;
                    wo[i] = 0.0;
                    continue;
                } else if (code1 == VFTO_NOCODE) {
                    co[i] = (unsigned char) code2;
                    wo[i] = (REAL) (w2 * (PS2[r] - PD2[r]));
                    continue;
                } else if (code2 == VFTO_NOCODE) {
                    co[i] = (unsigned char) code1;
                    wo[i] = (REAL) (w1 * (PS1[r] - PD1[r]));
                    continue;
                } else if (code1 == code2) {
                    co[i] = (unsigned char) code1;
                    double f12code = (w1 * PS1[r] + (1 - w1) * 0.25) * (w2 * PS2[r] + (1 - w2) * 0.25);
                    double f12other = (w1 * PD1[r] + (1 - w1) * 0.25) * (w2 * PD2[r] + (1 - w2) * 0.25);
                    double pcode = f12code / (f12code + 3 * f12other);
                    wo[i] = (REAL) ((pcode - 0.25) * 4.0 / 3.0);
                    if (wo[i] < 1e-6) wo[i] = (REAL) 1e-6;
                    continue;
                }
            }
            if (!f1) {
                for (int j = 0; j < 4; j++) mix1[j] = (REAL) ((1 - w1) * 0.25);
                if (code1 != VFTO_NOCODE) mix1[code1] = (REAL) ((double) mix1[code1] + w1);
                f1 = mix1;
            }
            if (!f2) {
                for (int j = 0; j < 4; j++) mix2[j] = (REAL) ((1 - w2) * 0.25);
                if (code2 != VFTO_NOCODE) mix2[code2] = (REAL) ((double) mix2[code2] + w2);
                f2 = mix2;
            }
            co[i] = VFTO_NOCODE;
            wo[i] = 1.0;
            REAL *f = fo + i * 4;
            double lkAB = 0;
            for (int j = 0; j < 4; j++) {
                f[j] = (REAL) ((f1[j] * PS1[r] + (1.0 - f1[j]) * PD1[r]) * (f2[j] * PS2[r] + (1.0 - f2[j]) * PD2[r]));
                lkAB += f[j];
            }
            double inv = 1.0 / lkAB;
            for (int j = 0; j < 4; j++) f[j] = (REAL) ((double) f[j] * inv);
        }
        return;
    }
    REAL ee1[VFTO_MAXRATES * VFTO_MAXCODES], ee2[VFTO_MAXRATES * VFTO_MAXCODES];
    FN(vfto_exp_eigen_rates)(len1, rates, nRate, tm->eigenval, nCodes, MLMinRelBranchLength, ee1);
    FN(vfto_exp_eigen_rates)(len2, rates, nRate, tm->eigenval, nCodes, MLMinRelBranchLength, ee2);
    const REAL *fGap = tm->codefreq + nCodes * nCodes;
    REAL f1mix[VFTO_MAXCODES], f2mix[VFTO_MAXCODES], fM1[VFTO_MAXCODES], fM2[VFTO_MAXCODES], fPost[VFTO_MAXCODES];
    for (int64_t i = 0; i < nPos; i++) {
        if (c1[i] == VFTO_NOCODE && c2[i] == VFTO_NOCODE && w1a[i] == 0 && w2a[i] == 0) {
            wo[i] = 0;
            continue;
        }
        const REAL *e1 = ee1 + ratecat[i] * nCodes, *e2 = ee2 + ratecat[i] * nCodes;
        const REAL *f1 = HASVEC(w1a, c1, i) ? f1a + i * nCodes : NULL;
        const REAL *f2 = HASVEC(w2a, c2, i) ? f2a + i * nCodes : NULL;
        REAL *fOut = fo + i * nCodes;
        if (!f1) {
            f1 = tm->codefreq + (c1[i] == VFTO_NOCODE ? nCodes : c1[i]) * nCodes;
            double w = w1a[i];
            if (w > 0.0 && w < 1.0) {
                for (int j = 0; j < nCodes; j++) f1mix[j] = (REAL) (w * f1[j] + (1.0 - w) * fGap[j]);
                f1 = f1mix;
            }
        }
        if (!f2) {
            f2 = tm->codefreq + (c2[i] == VFTO_NOCODE ? nCodes : c2[i]) * nCodes;
            double w = w2a[i];
            if (w > 0.0 && w < 1.0) {
                for (int j = 0; j < nCodes; j++) f2mix[j] = (REAL) (w * f2[j] + (1.0 - w) * fGap[j]);
                f2 = f2mix;
            }
        }
        for (int j = 0; j < nCodes; j++) {
            fM1[j] = f1[j] * e1[j];
            fM2[j] = f2[j] * e2[j];
        }
        if (nCodes == 4) {
            for (int j = 0; j < 4; j++) {
                double out1 = 0, out2 = 0;
                for (int k = 0; k < 4; k++) {
                    REAL p1 = fM1[k] * tm->codefreq[j * 4 + k];
                    REAL p2 = fM2[k] * tm->codefreq[j * 4 + k];
                    out1 += p1;
                    out2 += p2;
                }
                fPost[j] = (REAL) (out1 * out2 * tm->statinv[j]);
            }
            double tot = 0;
            for (int j = 0; j < 4; j++) tot += fPost[j];
            double inv = 1.0 / tot;
            for (int j = 0; j < 4; j++) fPost[j] = (REAL) ((double) fPost[j] * inv);
            /* matrix_by_vector4(eigeninvT, fPost, fOut): o = sum_j fPost[j] * eigeninvT[j][:], SSE/AVX order */
            for (int cidx = 0; cidx < 4; cidx++) {
                REAL o = 0;
                for (int j = 0; j < 4; j++) {
                    REAL p = fPost[j] * tm->eigeninvT[j * 4 + cidx];
                    o = o + p;
                }
                fOut[cidx] = o;
            }
        } else {
            for (int j = 0; j < nCodes; j++) {
                /* vector_dot_product_rot(fM1, fM2, codeFreq[j]) * statinv[j], NJ.tcc:2381 */
                REAL d1 = FN(red4_mul)(fM1, tm->codefreq + j * nCodes, nCodes);
                REAL d2 = FN(red4_mul)(fM2, tm->codefreq + j * nCodes, nCodes);
                REAL value = d1 * d2;
                value = value * tm->statinv[j];
                fPost[j] = value >= 0 ? value : 0;
            }
            double tot = FN(red4_sum)(fPost, nCodes);
            double inv = 1.0 / tot;
            REAL invr = (REAL) inv; /* vector_multiply_by takes numeric_t */
            for (int j = 0; j < nCodes; j++) fPost[j] = fPost[j] * invr;
            for (int j = 0; j < nCodes; j++) fOut[j] = FN(red4_mul)(fPost, tm->eigeninv + j * nCodes, nCodes);
        }
    }
}

#undef HASVEC
