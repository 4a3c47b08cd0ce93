// fic_device.h -- shared host/device declarations for the gfx950 fractal-encode path.
//
// Geometry and notation follow SURVEY.md section 8:  image W x H, block side B,
// n = B*B, scaled image Ws x Hs (2:1 box average), range grid Rw x Rh (N_r),
// domain-pool grid Dw x Dh (N_d), window side wK (wK == Dw == Dh: full search).
#pragma once
#include <stdint.h>

struct FicGeom {
    int W, H, B, n, lgn;     // lgn = log2(n)
    int Ws, Hs, abstand;     // abstand = B/4, pool stride in scaled pixels (FC:1019)
    int Rw, Rh, Nr;
    int Dw, Dh, Nd;
    int wK, n_iso, planes;
    int DW;                  // dwords per block = n/4
    int NR;                  // range blocks held per lane by the fast sweep (tile = 64*NR ranges)
    int tiles;               // ceil(Nr / (64*NR))
    int Nr_pad;              // tiles * 64 * NR
    int Nd_pad;              // Nd + FIC_POOL_PAD (zero tail so the prefetch may over-read)
    int full;                // 1 when wK == Dw == Dh (window origin is (0,0) for every range)
    int q_shape;             // MFMA shape of the 1-isometry k_sweep_q at B = 8 / 16: 0 = by pool size, 1 = 16x16x32, 2 = 32x32x16 (option "q_shape")
};

#define FIC_POOL_PAD 8

// Per-domain-block statistics streamed beside the pixels (8 bytes).
//   sum = sum of the n pixels  (Domainblock.mittelWert = sum / n,  DB:92-98)
//   s32 = (float) sqrt((double) variance)   -- ONLY used by the conservative prune test
struct FicDomStat {
    uint32_t sum;
    float s32;
};

// Per-range-block statistics.
//   rM  = getMittelwert(range)            (FC:67-73)
//   rem = sum(range) - n*rM = sum_i (r_i - rM)  == varianzRange after the loop at FC:665-672
struct FicRngStat {
    int32_t rM;
    int32_t rem;
};

// Candidate key: (orderable f32 error) << 32 | candidate index.  Unsigned min over
// keys == Java's strict '<' scan in ascending candidate order (FC:619-632).
#define FIC_KEY_NONE 0xFFFFFFFFFFFFFFFFull

// Per-plane decoder state (decodeGreyScale FC:381-418), device resident.
struct FicDecodeState {
    unsigned long long ssd[50];   // exact sum (range - value)^2 of each iteration
    float avg;                    // FractalCompression.avgError carried INTO the next iteration (FC:407,416)
    float avg_out;                // avgError after the last executed iteration (what CTL:180 displays)
    int iters;                    // iterations executed
    int done;                     // loop ended (avgError < 1, FC:414, or counter == 49)
    int bad_index;                // a row pointed outside the pool (Java: ArrayIndexOutOfBounds at FC:394)
    int seq_sums;                 // iterations whose sum had to be re-accumulated in Java's order (not exact in float)
};

// Joint-RGB path (encodeRGB FC:171-219).  Per domain block:
//   msum  = mittelWertR + mittelWertG + mittelWertB            (DB:33-39)
//   vD    = sum_i greyD_i = varianzDomain after the loop        (FC:778-791; Domainblock.variance stays 0)
//   varsq = varianceR + varianceG + mittelWertB  (sic, f32)     (FC:776)
struct FicRgbDomStat {
    int32_t msum, vD;
    float varsq;
    int32_t mR, mG, mB;
    int32_t pad0, pad1;
};
// Per range block: channel means (FC:771-773) and vR = sum_i greyR_i = varianzRange (FC:790).
struct FicRgbRngStat {
    int32_t mR, mG, mB, vR;
};
