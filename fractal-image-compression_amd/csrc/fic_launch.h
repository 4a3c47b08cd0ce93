// fic_launch.h -- internal: device buffer bundles + the kernel launchers of fic_prep / fic_sweep / fic_mfma /
// fic_decode / fic_rgb .hip, as called by fic_capi.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fic_device.h"

// Device-resident working set of one context (all planes of the batch).
//   pool_pix : u8  [planes][Nd_pad][n]    expanded domain pool, block pixels row-major (FC:1036), 64-B aligned
//   pool_st  :     [planes][Nd_pad]       {sum, (float)sqrt(var)}  -- streamed beside the pixels
//   pool_var : u32 [planes][Nd_pad]       Domainblock.variance (exact integer)
//   pool_s64 : f64 [planes][Nd_pad]       Math.sqrt((double) variance)
//   rng_pix  : u32 [planes][tiles][NR][n_iso][DW][64]   lane-transposed range blocks + isometry copies
//   rng_st   :     [planes][Nr_pad]       {rM, rem}
//   key      : u64 [planes][Nr_pad]       (orderable error << 32) | candidate
struct FicBuffers {
    uint8_t* gray;
    uint8_t* scaled;
    uint8_t* pool_pix;
    FicDomStat* pool_st;
    uint32_t* pool_var;
    double* pool_s64;
    uint32_t* rng_pix;
    FicRngStat* rng_st;
    unsigned long long* key;
};

// Per-range results, [planes][Nr] each (qrows: [planes][Nr][3]).
struct FicOutputs {
    int32_t* idx_local;   // window-local index  = imageInfo[j][0]   (FC:156, 629)
    int32_t* idx_global;  // pool index          = calculateIndices  (FC:853-893)
    int32_t* iso;         // winning isometry (0 when n_iso == 1)
    float* a;             // unquantised contrast   imageInfo[j][1]
    float* b;             // unquantised brightness imageInfo[j][2]
    float* err;           // winning error (diagnostic)
    int32_t* qrows;       // writeData rows: (int)idx, (int)(a*100), (int)b   (FC:242-244)
    int32_t* records;     // [planes][Nr][6] {idx_local, a bits, b bits, iso, (int)(a*100), (int)b}: the unit of the multi-GPU gather
};

#ifdef __HIPCC__
#include "fic_devfn.h"
// The tail of getBestDomainblock (FC:634-642: a = cov / varD, clamp, b = rM - a dM, never fused) + writeData's quantiser (FC:242-244)
// + the packed 24-byte record, for range j whose winner is (window-local wloc, pool block gi, isometry k); acc = sum copy_k[pos] d[pos].
// Shared by k_finalize and by the sweep's fused tail (fic_q.hip).
__device__ __forceinline__ void finalize_store(const FicOutputs& out, const FicGeom& g, int plane, int j, unsigned long long kk, int wloc,
                                               int gi, int k, uint32_t acc, FicDomStat ds, FicRngStat rs, uint32_t varu)
{
    int dM = (int)(ds.sum >> g.lgn);
    int cov = (int)acc - rs.rM * (int)ds.sum - dM * rs.rem;
    float var = (float)varu;
    float a = __fdiv_rn((float)cov, var);              // FC:634  (0/0 -> NaN when a flat block wins)
    if (a < -1.0f) a = -1.0f;                          // FC:636-639 (NaN passes through)
    else if (a > 1.0f) a = 1.0f;
    float b = __fsub_rn((float)rs.rM, __fmul_rn(a, (float)dM));   // FC:641, never fused
    size_t o = (size_t)plane * g.Nr + j;
    const int qa = java_f2i(__fmul_rn(a, 100.0f)), qb = java_f2i(b);
    out.qrows[3 * o + 0] = wloc;                       // (int) imageInfo[row][0]
    out.qrows[3 * o + 1] = qa;
    out.qrows[3 * o + 2] = qb;
    if (a != a) a = __uint_as_float(0x7FC00000u);      // canonical NaN (Java has one NaN value)
    if (b != b) b = __uint_as_float(0x7FC00000u);
    out.idx_local[o] = wloc;
    out.idx_global[o] = gi;
    out.iso[o] = k;
    out.a[o] = a;
    out.b[o] = b;
    out.err[o] = f32_from_orderable((uint32_t)(kk >> 32));
    // the same row once more as one 24-byte record: what a rank contributes to the codebook gather
    int32_t* rec = out.records + 6 * o;
    rec[0] = wloc;
    rec[1] = (int32_t)__float_as_uint(a);
    rec[2] = (int32_t)__float_as_uint(b);
    rec[3] = k;
    rec[4] = qa;
    rec[5] = qb;
}
#endif

int fic_launch_argb_to_gray(const int32_t* argb, uint8_t* gray, size_t npix, hipStream_t s);
int fic_launch_scale(const uint8_t* gray, uint8_t* scaled, const FicGeom& g, hipStream_t s);
int fic_launch_pool(const uint8_t* scaled, uint8_t* pool_pix, FicDomStat* st, uint32_t* var, double* s64,
                    const FicGeom& g, hipStream_t s);
int fic_launch_range(const uint8_t* gray, uint32_t* rng_pix, FicRngStat* rst, const FicGeom& g, hipStream_t s,
                     int with_copies = 1);      // 0: statistics only (the sweep brings its own range store)
int fic_launch_sweep_generic(const FicBuffers& b, const FicGeom& g, int r_begin, int r_count, hipStream_t s);
int fic_fast_variant(int B, int n_iso, int* NR, int* NC);
int fic_launch_sweep_fast(const FicBuffers& b, const FicGeom& g, int tile0, int ntiles, int chunk_len, int nchunks,
                          hipStream_t s);
int fic_launch_finalize(const FicBuffers& b, const FicOutputs& out, const FicGeom& g, int r_begin, int r_count,
                        hipStream_t s, int from_gray = 0);   // 1: recompute the winner's covariance from the image (no rng_pix)
int fic_launch_collage(const FicBuffers& b, const FicOutputs& out, int32_t* collage, const FicGeom& g, hipStream_t s);
int fic_launch_sqrt_probe(double* out, uint32_t first, uint32_t count, hipStream_t s);

// joint-RGB encode (FC:171-219): per-range results, [N_r] each (qrows: [N_r][5], FC:250-254)
struct FicRgbOutputs {
    int32_t* idx_local;
    int32_t* idx_global;
    float* a;
    float* bR;
    float* bG;
    float* bB;
    int32_t* qrows;
};
struct FicRgbBuffers {
    int32_t* argb;           // input [H][W]
    int32_t* scaled;         // [H/2][W/2] packed ARGB
    uint16_t* pool_sum;      // [N_d][n]  R+G+B per domain pixel
    float* pool_cf;          // [N_d][n]  greyD_i = pool_sum - msum as exact f32 (full search at B = 4 / 8 only, else NULL)
    FicRgbDomStat* pool_st;  // [N_d]
    int16_t* rng_t;          // [N_r][n]  greyR_i
    FicRgbRngStat* rng_st;   // [N_r]
    unsigned long long* key; // [N_r]
};
// buffers of the matrix-core full search of one image (k_sweep_q<NK, 3>); all NULL = the VALU sweeps
struct FicRgbQ {
    void *poolQ = nullptr, *dflat = nullptr, *rngQ = nullptr, *qst = nullptr, *rngE = nullptr, *theta_g = nullptr, *amax = nullptr;
    int ndtiles = 0, ndtiles_alloc = 0, nct_alloc = 0, tiles_per_chunk = 0, nchunks = 0;
};
int fic_launch_rgb_encode(const FicRgbBuffers& b, const FicRgbOutputs& out, int32_t* collage, const FicGeom& g,
                          hipStream_t s, const FicRgbQ* q = nullptr);
int fic_launch_rgbq(const uint16_t* pool_sum, const FicRgbDomStat* pool_st, const int16_t* rng_t, const FicRgbRngStat* rng_st,
                    unsigned long long* key, void* poolQ, void* dflat, void* rngQ, void* qst, void* rngE, void* theta_g, void* amax,
                    const FicGeom& g, int ndtiles, int ndtiles_alloc, int nct_alloc, int tiles_per_chunk, int nchunks, hipStream_t s);

// opt-in matrix-core sweeps ("sweep" = 3)
int fic_mfma8_group(int B);          // range blocks per workgroup of the 8-isometry kernel
int fic_launch_mfma_prep_pool(const uint8_t* pool_pix, void* poolB, const FicGeom& g, int ndtiles_alloc, hipStream_t s);
int fic_launch_mfma_prep_range(const uint32_t* rng_pix, const FicRngStat* rng_st, void* rngA, int* rconst,
                               const FicGeom& g, int ngroups, hipStream_t s);
int fic_launch_sweep_mfma(const FicBuffers& b, const void* poolB, const void* rngA, const int* rconst, const FicGeom& g,
                          int ngroups, int group0, int ngroups_launch, int ndtiles, int ndtiles_alloc, int tiles_per_chunk,
                          int nchunks, hipStream_t s);
int fic_mfma1_ct(int B);
int fic_launch_mfma1_prep(const FicBuffers& b, void* poolA, void* pool_sw, void* rngB, void* rconst, const FicGeom& g,
                          int ndtiles_alloc, int nctiles_alloc, hipStream_t s);
int fic_launch_sweep_mfma1(const FicBuffers& b, const void* poolA, const void* pool_sw, const void* rngB,
                           const void* rconst, const FicGeom& g, int ct_begin, int ct_end, int ndtiles, int ndtiles_alloc,
                           int nctiles_alloc, int tiles_per_chunk, int nchunks, hipStream_t s);
// bf16-operand matrix-core sweeps (fic_bf16.hip): "sweep" = 3 at B = 4 / 8
int fic_bf16_steps(int B);           // MFMA steps of K = 16 per block
int fic_bf16_group8(int B);          // range blocks per workgroup, 8-isometry kernel
int fic_bf16_ct1(int B);             // column tiles (x32 ranges) per workgroup, 1-isometry kernel
int fic_launch_bf16_prep(const FicBuffers& b, void* poolF, void* pool_w, void* rngF, const FicGeom& g, int ndtiles_alloc,
                         int nrtiles_alloc, hipStream_t s);
int fic_launch_sweep_bf16(const FicBuffers& b, const void* poolF, const void* pool_w, const void* rngF, const FicGeom& g,
                          int nrtiles_alloc, int group0, int ngroups_launch, int ndtiles, int ndtiles_alloc, int tiles_per_chunk, int nchunks,
                          hipStream_t s);
int fic_launch_sweep_bf16_1(const FicBuffers& b, const void* poolF, const void* pool_w, const void* rngF, const FicGeom& g,
                            int ct_begin, int ct_end, int ndtiles, int ndtiles_alloc, int nctiles_alloc, int tiles_per_chunk,
                            int nchunks, hipStream_t s);
// VALU sweep with algebraic isometries (fic_d4.hip): n_iso = 8, B = 8 / 16
int fic_d4_words(int B);             // dwords (dot2 pairs) per block in the group-Fourier form, 0 if not built for B
int fic_launch_d4_prep(const FicBuffers& b, uint32_t* rng_d4, uint32_t* pool_d4, const FicGeom& g, hipStream_t s);
int fic_launch_sweep_d4(const FicBuffers& b, const uint32_t* rng_d4, const uint32_t* pool_d4, const FicGeom& g, int tile0,
                        int ntiles, int chunk_len, int nchunks, hipStream_t s);
// default full-search sweep (fic_q.hip, "sweep" = 6): normalised f16 prune GEMM on the matrix cores + exact evaluation of survivors
int fic_q_ct(int B);                 // column tiles (x32 columns) per workgroup
int fic_q_ctw_host(int B);           // column tiles (x32 columns) one wave keeps in registers
int fic_q_prep_fused(const FicGeom& g, int ndtiles_alloc, int ngrp);   // 1: fic_launch_q_prep makes the scaled image itself (no k_scale before it)
int fic_q_shape16(const FicGeom& g);
int fic_q_multi_kind(int nchunks, int tiles_per_chunk);   // 0: k_sweep_q<.., false>, 1: <.., true>, 2: k_sweep_qs / k_sweep_q16s (short chunks) // 1: k_sweep_q16 (v_mfma_f32_16x16x32_f16) runs this geometry's 1-isometry sweep
int fic_q_cols_per_range(int B, int n_iso);   // sweep columns per range block: 1 (1 isometry), 8 (B = 4), 4 (isometry pairs, B = 8 / 16)
int fic_q_unroll(int B, int n_iso);   // unroll factor of the sweep loop: chunks are whole multiples, the store has that many spare tiles twice
int fic_q_resident(int B);          // workgroups of k_sweep_q a CU holds at once (chunk policy)
int fic_launch_q_prep(const FicBuffers& b, void* poolQ, void* dflat, void* rngQ, void* rngC, void* rngE, void* theta_g,
                      const FicGeom& g, int ndtiles_alloc, int nct_alloc, int grp0, int ngrp, hipStream_t s);
int fic_launch_sweep_q(const FicBuffers& b, const void* poolQ, const void* dflat, const void* rngQ, const void* rngC, const void* rngE,
                       void* theta_g, const FicGeom& g, int ct_begin, int ct_end, int ndtiles, int ndtiles_alloc,
                       int nct_alloc, int tiles_per_chunk, int nchunks, hipStream_t s, unsigned long long* stats = nullptr, int dbg_noflag = 0,
                       const FicOutputs* fin_out = nullptr, unsigned int* fin_count = nullptr, int r_begin = 0, int r_count = 0);
int fic_launch_decode_iteration_rgb(int32_t* scaled, int32_t* image, const int32_t* qrows5, FicDecodeState* state,
                                    uint32_t* sqbuf, int counter, const FicGeom& g, hipStream_t s);

// decoder (FC:356-421)
// maps: fic_float_sum_map_words(count) u32 of scratch; out2[0] = the sum, out2[1] = segments that took the sequential-order path
int fic_launch_float_sum_probe(float carry, const uint32_t* vals, int count, uint32_t* maps, float* out2, hipStream_t s);
size_t fic_float_sum_map_words(size_t count);
// u32 words of the decoder's `sqbuf` scratch for `planes` images of wh pixels (squares + segment maps of the float sum)
size_t fic_decode_sq_words(size_t planes, size_t wh);
// sqbuf: u32 [planes][W*H] per-pixel squared changes of the iteration in Java's visiting order (for the sequential f32 sum)
int fic_launch_decode_step(FicDecodeState* state, uint32_t* sqbuf, int counter, int wh, int planes, hipStream_t s);
int fic_launch_decode_iteration(uint8_t* scaled, uint8_t* image, const int32_t* qrows, const int32_t* iso,
                                FicDecodeState* state, uint32_t* sqbuf, int counter, const FicGeom& g, hipStream_t s);
