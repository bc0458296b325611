/* TEST INFRASTRUCTURE — the reference's hit sort, restated as a total order.
   psort (Utils.h:126-146) = boost parallel_stable_sort with CompareHitsByCriterion (NJ.tcc:7301-7306), a
   comparator that answers "true" on ties.  With one sort thread the observable result is: ascending
   criterion, ties in DESCENDING original position (pinned by the *.sorted_j fixtures). */
typedef struct {
    REAL key;
    int64_t idx;
} FN(sortrec);

static int FN(sortcmp)(const void *a, const void *b) {
    const FN(sortrec) *x = (const FN(sortrec) *) a, *y = (const FN(sortrec) *) b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return x->idx > y->idx ? -1 : (x->idx < y->idx ? 1 : 0);
}

void FN(vfto_sort_hits)(const REAL *crit, int64_t n, int64_t *order) {
    FN(sortrec) *r = (FN(sortrec) *) malloc((size_t) n * sizeof(FN(sortrec)));
    for (int64_t i = 0; i < n; i++) {
        r[i].key = crit[i];
        r[i].idx = i;
    }
    qsort(r, (size_t) n, sizeof(FN(sortrec)), FN(sortcmp));
    for (int64_t i = 0; i < n; i++) order[i] = r[i].idx;
    free(r);
}
