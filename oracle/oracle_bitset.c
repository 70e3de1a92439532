/* CPU oracle, bit-packed form: greedy maximum-coverage over column-major bitsets.
 *
 * TEST INFRASTRUCTURE ONLY.  Not linked into, loaded by, or called from anything under
 * utmos_amd/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * It restates /root/reference/utmos/select.py on the layout the GPU uses, so that GPU
 * results can be checked at sizes where the row-major numpy restatement
 * (oracle/utmos_oracle.py, pinned on the reference's golden TSVs and on traces of its own code; this file is checked against the same traces) is too slow.  The two
 * oracles are checked against each other in tests/test_oracle_bitset.py.
 *
 *   score   select.py:33-41   a variant is skipped when a *used* (state 0) sample carries it;
 *                             otherwise its value is added to every carrier's score and 1 to its count.
 *                             Here: covered = OR of used columns; count[s] = popcount(col_s & ~covered);
 *                             with AF the values are added one by one in ascending variant order into
 *                             a double, which is the rounding sequence of `scores += row` (adding 0.0
 *                             for non-carriers changes nothing).
 *   mask    select.py:43      non-selectable samples score 0 (before weights)
 *   weights select.py:45-47   scores *= weights
 *   argmax  select.py:48      first maximum (numpy: a NaN is a maximum)
 *   stop    select.py:51-52   best score == 0 -> no row;  select.py:110-112  all variants captured -> stop after the row
 *   update  select.py:100     winner becomes used (and therefore covers from the next iteration on)
 *
 * A variant whose AF value is exactly 0.0 is an all-zero row of the reference's float matrix:
 * it is never counted, never scored and never "covered" -- af_mode masks it out.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_AF_NONE = 0, ORC_AF_F32 = 1, ORC_AF_F64 = 2 };

static inline int popc64(uint64_t x) { return __builtin_popcountll(x); }

/* Per-sample count and score of one iteration.  cols[s*stride + w]; live = ~covered & afmask. */
static void score_columns(const uint64_t *cols, uint64_t stride, uint64_t n_words, uint32_t n_samp,
                          const uint64_t *live, const uint8_t *state, int af_mode, const void *af,
                          int64_t *count, double *score)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (uint32_t s = 0; s < n_samp; ++s) {
        count[s] = 0;
        score[s] = 0.0;
        if (state[s] != 1)
            continue; /* its score is forced to 0 anyway and its count is never read */
        const uint64_t *c = cols + (uint64_t)s * stride;
        int64_t n = 0;
        double acc = 0.0;
        for (uint64_t w = 0; w < n_words; ++w) {
            uint64_t x = c[w] & live[w];
            if (!x)
                continue;
            n += popc64(x);
            if (af_mode == ORC_AF_F32) {
                const float *a = (const float *)af + w * 64;
                while (x) { acc += (double)a[__builtin_ctzll(x)]; x &= x - 1; }
            } else if (af_mode == ORC_AF_F64) {
                const double *a = (const double *)af + w * 64;
                while (x) { acc += a[__builtin_ctzll(x)]; x &= x - 1; }
            }
        }
        count[s] = n;
        score[s] = af_mode == ORC_AF_NONE ? (double)n : acc;
    }
}

/* numpy argmax over doubles: first maximum; a NaN beats everything and the first NaN wins. */
static int64_t argmax_first(const double *v, uint32_t n)
{
    int64_t best = 0;
    if (isnan(v[0]))
        return 0;
    for (uint32_t i = 1; i < n; ++i) {
        if (isnan(v[i]))
            return i;
        if (v[i] > v[best])
            best = i;
    }
    return best;
}

static uint64_t *build_live(const uint64_t *cols, uint64_t stride, uint64_t n_var, uint32_t n_samp,
                            const uint8_t *state, int af_mode, const void *af)
{
    uint64_t n_words = (n_var + 63) / 64;
    uint64_t *live = (uint64_t *)malloc(n_words * 8 + 8);
    for (uint64_t w = 0; w < n_words; ++w) {
        uint64_t m = ~0ull;
        if (w == n_words - 1 && (n_var & 63))
            m = (1ull << (n_var & 63)) - 1;
        if (af_mode != ORC_AF_NONE) {
            for (int b = 0; b < 64; ++b) {
                uint64_t v = w * 64 + b;
                if (v >= n_var) break;
                double a = af_mode == ORC_AF_F32 ? (double)((const float *)af)[v] : ((const double *)af)[v];
                if (a == 0.0) m &= ~(1ull << b);
            }
        }
        live[w] = m;
    }
    for (uint32_t s = 0; s < n_samp; ++s)
        if (state[s] == 0) {
            const uint64_t *c = cols + (uint64_t)s * stride;
            for (uint64_t w = 0; w < n_words; ++w) live[w] &= ~c[w];
        }
    return live;
}

static int64_t finish(const uint8_t *state, const double *weights, uint32_t n_samp, double *score)
{
    for (uint32_t s = 0; s < n_samp; ++s) {
        if (state[s] != 1) score[s] = 0.0;
        if (weights) score[s] *= weights[s];
    }
    int64_t best = argmax_first(score, n_samp);
    return score[best] == 0 ? -1 : best;
}

/* One scoring pass with no selection: fills count[S] and final (masked, weighted) score[S].
 * Returns the argmax index, or -1 when the best score is 0. */
int64_t orc_score(const uint64_t *cols, uint64_t stride, uint64_t n_var, uint32_t n_samp,
                  const uint8_t *state, const double *weights, int af_mode, const void *af,
                  int64_t *count, double *score)
{
    uint64_t n_words = (n_var + 63) / 64;
    uint64_t *live = build_live(cols, stride, n_var, n_samp, state, af_mode, af);
    score_columns(cols, stride, n_words, n_samp, live, state, af_mode, af, count, score);
    free(live);
    return finish(state, weights, n_samp, score);
}

/* Full greedy loop.  state is updated in place.  Returns the number of rows produced.
 * idx_out/new_out need k_max entries; score_out (k_max) may be NULL. */
int64_t orc_greedy(const uint64_t *cols, uint64_t stride, uint64_t n_var, uint32_t n_samp,
                   uint8_t *state, const double *weights, int af_mode, const void *af,
                   int64_t k_max, int64_t *idx_out, int64_t *new_out, double *score_out)
{
    uint64_t n_words = (n_var + 63) / 64;
    int64_t *count = (int64_t *)malloc((size_t)n_samp * 8);
    double *score = (double *)malloc((size_t)n_samp * 8);
    /* covered is kept incrementally: the used set only ever grows by the winner */
    uint64_t *live = build_live(cols, stride, n_var, n_samp, state, af_mode, af);
    int64_t done = 0, captured = 0;
    for (int64_t k = 0; k < k_max; ++k) {
        score_columns(cols, stride, n_words, n_samp, live, state, af_mode, af, count, score);
        int64_t best = finish(state, weights, n_samp, score);
        if (best < 0)
            break;
        idx_out[done] = best;
        new_out[done] = count[best];
        if (score_out) score_out[done] = score[best];
        ++done;
        captured += count[best];
        state[best] = 0;
        const uint64_t *c = cols + (uint64_t)best * stride;
        for (uint64_t w = 0; w < n_words; ++w) live[w] &= ~c[w];
        if (captured >= (int64_t)n_var)
            break;
    }
    free(count);
    free(score);
    free(live);
    return done;
}
