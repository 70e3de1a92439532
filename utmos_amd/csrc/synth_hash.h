// Counter-based synthetic presence-matrix generator, shared by the HIP kernel and the host
// version so that both produce bit-identical matrices (integer arithmetic only).
//
// Shape of the data (DESIGN.md "Synthetic input"): per variant v a carrier probability p_v drawn
// log-uniformly over [2^-L, 1), L = ceil(log2 S) octaves -- the 1/c site-frequency spectrum
// SURVEY.md §8d asks for -- every cell (s, v) an independent Bernoulli(p_v), plus one forced carrier
// per variant so that every row is informative (select.py:276-279 would drop it otherwise).
// About 5 % of the variants end up private to one sample, which keeps a "select all" run going
// for ~S iterations instead of saturating early.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define UTM_HD __host__ __device__ __forceinline__
#else
#define UTM_HD static inline
#endif

UTM_HD uint64_t utm_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

UTM_HD uint32_t utm_octaves(uint32_t n_samp_total)
{
    uint32_t l = 1;
    while ((1u << l) < n_samp_total && l < 31) ++l;
    return l;
}

// per-variant draw
UTM_HD uint64_t utm_var_key(uint64_t seed, uint64_t v_global)
{
    return utm_mix64(seed + 0x9E3779B97F4A7C15ull * (v_global + 1));
}

// 32-bit Bernoulli threshold: P(cell set) = thr / 2^32
UTM_HD uint32_t utm_var_threshold(uint64_t key, uint32_t octaves)
{
    uint32_t hi = (uint32_t)(key >> 32);
    uint32_t e = (uint32_t)(((uint64_t)(hi & 0xFFFFu) * octaves) >> 16); // 0 .. octaves-1
    uint32_t m = 0x80000000u | (hi >> 1);                               // [2^31, 2^32)
    return m >> e;
}

UTM_HD uint32_t utm_var_forced(uint64_t key, uint32_t n_samp_total)
{
    return (uint32_t)(((uint64_t)(uint32_t)key * n_samp_total) >> 32);
}

UTM_HD uint64_t utm_sample_key(uint32_t s_global)
{
    return 0xD6E8FEB86659FD93ull * ((uint64_t)s_global + 1);
}

UTM_HD int utm_cell(uint64_t key, uint32_t thr, uint32_t forced, uint64_t skey, uint32_t s_global)
{
    uint32_t r = (uint32_t)(utm_mix64(key ^ skey) >> 32);
    return (r < thr) | (s_global == forced);
}

// Synthetic allele frequency: expected carrier count / (2 S), as the float32 the hdf5 path would hold.
UTM_HD float utm_var_af(uint32_t thr, uint32_t n_samp_total)
{
    uint64_t c = ((uint64_t)thr * n_samp_total) >> 32;
    if (c < 1) c = 1;
    return (float)((double)c / (double)(2ull * n_samp_total));
}
