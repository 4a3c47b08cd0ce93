/*
 * fic_synth.h -- the synthetic grey benchmark inputs of SURVEY.md section 8(d), integer-only so that every language
 * generates the same bytes (the Python twin is fractal-image-compression_amd/synth.py; tests/test_host_cpu.py
 * compares the two).  Header-only, C and C++.
 *
 *   U : pix(x,y) = splitmix64(seed + y*W + x) >> 56            iid uniform bytes -- the distribution throughput is
 *                                                              quoted on (no flat blocks, rem == 0 with prob. 1/n)
 *   S : t = (x>>5) + (y>>5);  S(x,y) = ((t&3) << 6) + ((((x>>5) ^ (y>>5)) & 1) ? U(x,y) >> 6 : 0)
 *       32x32 tiles of levels {0,64,128,192}, alternate tiles carrying 2-bit noise: flat domain blocks (variance 0),
 *       flat ranges (rem == 0 -> index 0, FractalCompression.java:677), exact error ties and NaN fits.
 *   Seeds: cfg2 0xF1C0002, cfg3 0xF1C0003, cfg4 0xF1C0004, cfg5 0xF1C0005 + 3*image + channel.
 */
#ifndef FIC_SYNTH_H
#define FIC_SYNTH_H

#include <stdint.h>

#define FIC_SEED_CFG2 0xF1C0002ull
#define FIC_SEED_CFG3 0xF1C0003ull
#define FIC_SEED_CFG4 0xF1C0004ull
#define FIC_SEED_CFG5 0xF1C0005ull

/* output function of SplitMix64 applied to state z */
static inline uint64_t fic_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint8_t fic_synth_u_pixel(int x, int y, int w, uint64_t seed)
{
    return (uint8_t)(fic_splitmix64(seed + (uint64_t)y * (uint64_t)w + (uint64_t)x) >> 56);
}

static inline uint8_t fic_synth_s_pixel(int x, int y, int w, uint64_t seed)
{
    const int t = (x >> 5) + (y >> 5);
    const int noisy = ((x >> 5) ^ (y >> 5)) & 1;
    return (uint8_t)(((t & 3) << 6) + (noisy ? (fic_synth_u_pixel(x, y, w, seed) >> 6) : 0));
}

/* kind 'U' or 'S'; out: h rows of w bytes, scanline order */
static inline void fic_synth_image(char kind, int w, int h, uint64_t seed, uint8_t* out)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            out[(uint64_t)y * (uint64_t)w + (uint64_t)x] =
                (kind == 'S' || kind == 's') ? fic_synth_s_pixel(x, y, w, seed) : fic_synth_u_pixel(x, y, w, seed);
}

#endif /* FIC_SYNTH_H */
