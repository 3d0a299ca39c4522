/* pq_oracle_hot.h -- the two hot loops of the oracle, compiled twice by pq_oracle.c:
 * once for the baseline ISA (fmaf -> libm, correctly rounded) and once with AVX2+FMA
 * (fmaf -> vfmadd).  Both produce identical bits; selection is by cpuid at run time.
 * TEST INFRASTRUCTURE ONLY (see pq_oracle.c). */
/* one row; P is [d,d] row-major (rx[c] = sum_k x[k] P[k][c]); scratch ab[d] */
HOT_ATTR
static void HOT_NAME(rotate_row)(const float *x, const float *P, int64_t d, float *out, float *ab)
{
    for (int64_t kb = 0; kb < d; kb += PQO_KC) {
        int64_t ke = kb + PQO_KC < d ? kb + PQO_KC : d;
        for (int64_t c = 0; c < d; ++c) ab[c] = 0.0f;
        for (int64_t k = kb; k < ke; ++k) {
            const float xv = x[k];
            const float *Pk = P + k * d;
            for (int64_t c = 0; c < d; ++c) ab[c] = __builtin_fmaf(xv, Pk[c], ab[c]);
        }
        if (kb == 0)
            for (int64_t c = 0; c < d; ++c) out[c] = ab[c];
        else
            for (int64_t c = 0; c < d; ++c) out[c] = out[c] + ab[c];
    }
}

/* one (already rotated, contiguous) row -> M codes (as int64 so any index width fits) */
HOT_ATTR
static void HOT_NAME(encode_row)(const enc_tables *t, const float *row, float *acc, int64_t *codes)
{
    const int64_t K = t->K, dsub = t->dsub;
    for (int64_t m = 0; m < t->M; ++m) {
        const float *xs = row + m * dsub;
        const float xx = pqo_dot_unrolled(xs, xs, dsub);
        const float *cc = t->cc + m * K;
        /* (2) one fmaf chain per centroid, k ascending, restart every KC */
        for (int64_t kb = 0; kb < dsub; kb += PQO_KC) {
            int64_t ke = kb + PQO_KC < dsub ? kb + PQO_KC : dsub;
            float *ab = (kb == 0) ? acc : acc + K;
            for (int64_t j = 0; j < K; ++j) ab[j] = 0.0f;
            for (int64_t k = kb; k < ke; ++k) {
                const float xv = xs[k];
                const float *ctk = t->ct + (m * dsub + k) * K;
                for (int64_t j = 0; j < K; ++j) ab[j] = __builtin_fmaf(xv, ctk[j], ab[j]);
            }
            if (kb != 0)
                for (int64_t j = 0; j < K; ++j) acc[j] = acc[j] + ab[j];
        }
        /* (3) combine */
        int any_nan = 0;
        for (int64_t j = 0; j < K; ++j) {
            float tt = xx + cc[j];
            float u = acc[j] + acc[j];
            float dj = tt - u;
            acc[j] = dj;
            any_nan |= (dj != dj);
        }
        /* (4) first minimum */
        int64_t best = 0;
        if (any_nan) {
            best = pqo_first_min(acc, K);
        } else {
            float bv = acc[0];
            for (int64_t j = 1; j < K; ++j)
                if (acc[j] < bv) { bv = acc[j]; best = j; }
        }
        codes[m] = best;
    }
}

