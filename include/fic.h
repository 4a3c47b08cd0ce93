/*
 * fic.h -- C ABI of libfic_hip.so: the MI355X (gfx950) drop-in for the grey encode hot path
 * of bvk_ss19.FractalCompression (LariWa/Fractal-Image-Compression).
 *
 * The reference has no FFI seam of its own; the boundary is the body of
 * FractalCompression.encodeGrayScale between pool build and writeData
 * (src/bvk_ss19/FractalCompression.java:119-159), reached through
 * FractalCompression.encode(RasterImage, DataOutputStream) (FractalCompression.java:54).
 * Each entry point below names the reference code it replaces.  Plain pointers and sizes
 * only; INTEGRATION.md shows the JNI stub that binds them under the reference's own
 * Java entry point.
 *
 * Conventions
 *   - every int-returning function returns FIC_OK (0) or a negative FIC_E_* code;
 *     fic_last_error() then holds a human-readable message for the calling thread.
 *     The reference throws unchecked exceptions on bad geometry; the JNI
 *     shim turns a negative code into a thrown java.lang.Exception.
 *   - the library never retains caller pointers past return.
 *   - B = FractalCompression.blockgroesse (FractalCompression.java:14), wK =
 *     FractalCompression.widthKernel (:15).  Supported: B in {4, 8, 16} (the GUI's slider
 *     values, RLEAppController.java:131), 1 <= wK <= min(Dw, Dh); wK == Dw == Dh is full search.
 *   - n_iso = 1 is the reference algorithm (bit-identical); n_iso = 8 is this build's
 *     extension (the 8 isometries of the square; order documented in DESIGN.md).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry fails
 *     with FIC_E_NO_DEVICE.
 */
#ifndef FIC_H
#define FIC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIC_API __attribute__((visibility("default")))

#define FIC_OK 0
#define FIC_E_GEOMETRY (-1)    /* W,H,B combination the reference cannot encode (AIOOBE / div-by-zero there) */
#define FIC_E_WINDOW (-2)      /* wK outside 1..min(Dw,Dh)  (negative index in FractalCompression.java:145) */
#define FIC_E_ARGUMENT (-3)    /* null pointer, bad n_iso, bad plane/range span ... */
#define FIC_E_NO_DEVICE (-4)   /* no HIP device / device index out of range */
#define FIC_E_HIP (-5)         /* a HIP runtime call failed (message has the hipError string) */
#define FIC_E_NOT_GREY (-6)    /* argb input has r!=g or g!=b somewhere (isGreyScale, FractalCompression.java:32-45) */
#define FIC_E_STATE (-7)       /* call order: no input set / nothing encoded yet */
#define FIC_E_CAPACITY (-8)    /* output buffer too small */

typedef struct fic_ctx fic_ctx;

/* ---- library / device ------------------------------------------------------------------ */
FIC_API const char* fic_version(void);
FIC_API const char* fic_last_error(void);
FIC_API int fic_last_error_code(void);   /* code of the last failure on this thread (for NULL returns) */
FIC_API int fic_device_count(void);

/* Block-grid geometry exactly as encodeGrayScale derives it (FractalCompression.java:111-116)
 * and createCodebuch sizes the pool (:1019-1022).  Also the validator: FIC_E_GEOMETRY for
 * sizes on which the reference throws.  Any out pointer may be NULL. */
FIC_API int fic_geometry(int w, int h, int B, int* Rw, int* Rh, int* Dw, int* Dh);

/* isGreyScale (FractalCompression.java:32-45) on a host ARGB buffer: 1 grey, 0 colour. */
FIC_API int fic_is_greyscale_argb(const int32_t* argb, int w, int h);

/* ---- one-shot host-buffer encode (what the JNI shim binds) ------------------------------ */
/* Replaces the search of encodeGrayScale (FractalCompression.java:119-159): pool build
 * (createCodebuch :1015-1050), per-range window selection (:128-150), getBestDomainblock
 * (:613-644) with getErrorVarianceCovariance (:655-687).
 *   argb      RasterImage.argb, w*h ints, only (argb>>16)&0xff is read (:596, :977)
 *   idx_local [N_r] imageInfo[j][0] -- window-local candidate index
 *   a, b      [N_r] imageInfo[j][1..2] -- unquantised float32, bit-exact (needed by
 *             getBestGeneratedCollage :287); a NaN is returned as 0x7FC00000
 *   iso       [N_r] winning isometry id, all 0 for n_iso = 1; may be NULL
 *   qrows     [N_r][3] the ints writeData emits per range (:242-244); may be NULL
 * device: HIP device ordinal. */
FIC_API int fic_encode_gray_argb(const int32_t* argb, int w, int h, int B, int wK, int n_iso, int device,
                                 int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows);
/* Same with the R channel already extracted (one byte per pixel, scanline order). */
FIC_API int fic_encode_gray_u8(const uint8_t* gray, int w, int h, int B, int wK, int n_iso, int device,
                               int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows);

/* The same call sharded over the first n_gpus HIP devices of the node, still ONE synchronous call on one host thread --
 * what FractalCompression.encode (FractalCompression.java:54-59) is to its caller (RLEAppController.java:172-188).  The
 * range loop it replaces (:125-159) carries no state between iterations: device g sweeps a contiguous, tile-aligned span
 * of range blocks against its own replica of the pool (built from the replicated image), and the 24-byte codebook
 * records are gathered to device 0 with one grouped RCCL send/recv over xGMI (librccl is loaded on first use; n_gpus = 1
 * is exactly fic_encode_gray_argb on device 0 and needs no RCCL).  Results do not depend on n_gpus.
 * Environment: FIC_GATHER=copy replaces the RCCL gather by peer-to-peer copies; FIC_FAKE_DEVICES=k (test knob) lets
 * n_gpus exceed the visible devices up to k, logical devices sharing the real ones round-robin (gather by device copies:
 * RCCL does not allow two ranks on one device). */
FIC_API int fic_encode_gray_argb_multi(const int32_t* argb, int w, int h, int B, int wK, int n_iso, int n_gpus,
                                       int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows);
FIC_API int fic_encode_gray_u8_multi(const uint8_t* gray, int w, int h, int B, int wK, int n_iso, int n_gpus,
                                     int32_t* idx_local, float* a, float* b, int32_t* iso, int32_t* qrows);

/* The one-shot entries (grey and RGB) keep the device working sets of the last few geometries (the GUI
 * re-encodes the same image on every slider move, RLEAppController.java:125-145); this frees them, and the RCCL communicators of the multi-device entries. */
FIC_API void fic_release_cache(void);

/* writeData, grey branch (FractalCompression.java:230-246): big-endian int32 header
 * {0, w, h, B, wK} then qrows.  Returns bytes written (20 + 12*n_ranges) or a negative code. */
FIC_API int64_t fic_write_run_gray(const int32_t* qrows, int n_ranges, int w, int h, int B, int wK, uint8_t* out,
                                   int64_t capacity);

/* ---- handle API: device-resident, batched planes, range shards --------------------------- */
/* A context owns the working set for `planes` grey images of one geometry on one device
 * (config 5: 192 planes of 1024x1024; multi-GPU: one context per rank). */
FIC_API fic_ctx* fic_ctx_create(int device, int w, int h, int B, int wK, int n_iso, int planes);
FIC_API void fic_ctx_destroy(fic_ctx* ctx);

/* Input: host bytes [planes][h][w] (copied), host ARGB (copied + R extracted on device), or a
 * device pointer to bytes [planes][h][w] that stays owned by the caller and must remain valid
 * until fic_ctx_sync (no copy; this is the path bench.py times: inputs resident in HBM). */
FIC_API int fic_ctx_set_gray_host(fic_ctx* ctx, const uint8_t* gray);
FIC_API int fic_ctx_set_argb_host(fic_ctx* ctx, const int32_t* argb);
FIC_API int fic_ctx_set_gray_device(fic_ctx* ctx, const void* dev_gray);

/* Asynchronous encode of range blocks [range_begin, range_begin+range_count) of every plane on
 * `hip_stream` (a hipStream_t, NULL = default stream): pool build + range prep + sweep +
 * finalise.  range_count < 0 means "to the end".  Multi-GPU sharding calls this
 * with a different span per rank; results for other ranges are left untouched. */
FIC_API int fic_ctx_encode(fic_ctx* ctx, int range_begin, int range_count, void* hip_stream);
FIC_API int fic_ctx_sync(fic_ctx* ctx);

/* Results.  Host copies are [planes][N_r] ([planes][N_r][3] for qrows); any pointer may be NULL. */
FIC_API int fic_ctx_get_results_host(fic_ctx* ctx, int32_t* idx_local, float* a, float* b, int32_t* iso,
                                     int32_t* qrows, int32_t* idx_global, float* err);
/* Device pointers of the same arrays (for an RCCL gather by the host layer). */
FIC_API int fic_ctx_result_device_ptrs(fic_ctx* ctx, void** idx_local, void** a, void** b, void** iso, void** qrows,
                                       void** idx_global, void** err);

/* The same rows packed as the unit of the multi-GPU codebook gather: device pointer to int32
 * [planes][N_r][6] = {idx_local, a bits, b bits, iso, (int)(a*100), (int)b} (imageInfo[j] of
 * FractalCompression.java:156 + the ints writeData emits, :242-244), written by every fic_ctx_encode for the range
 * span it covered.  Owned by the context. */
FIC_API int fic_ctx_records_device_ptr(fic_ctx* ctx, void** records);

/* getBestGeneratedCollage (FractalCompression.java:269-300): grey ARGB [planes][h][w] of the
 * one-step collage from the unquantised a,b of the last encode (all ranges must be encoded). */
FIC_API int fic_ctx_collage_host(fic_ctx* ctx, int32_t* argb_out);

/* ---- joint-RGB encode ------------------------------------------------------------------------- */
/* encodeRGB (FractalCompression.java:171-219), what FractalCompression.encode dispatches to for
 * colour input (:55-58): scaleImageRGB (:901-962), createCodebuchRGB (:1058-1093) + Domainblock RGB
 * ctor (Domainblock.java:30-41), getBestDomainblockRGB (:697-735), getErrorVarianceCovarianceRGB
 * (:760-808).  One contrast `a` for the three channels, three brightness values.
 *   idx_local, a, bR, bG, bB  [N_r] = imageInfoRGB[j][0..4], unquantised float32 (NaN as 0x7FC00000)
 *   qrows5        [N_r][5] the ints writeData emits (:250-254); may be NULL
 *   collage_argb  getBestGeneratedCollageRGB (:308-347), w*h ARGB ints; may be NULL */
FIC_API int fic_encode_rgb_argb(const int32_t* argb, int w, int h, int B, int wK, int device, int32_t* idx_local,
                                float* a, float* bR, float* bG, float* bB, int32_t* qrows5, int32_t* collage_argb);
/* Handle API of the same path: the device-resident working set of `planes` colour images of one geometry (a batch of
 * config-5 style images, or one image encoded again and again); same kernels, same bits as fic_encode_rgb_argb per image.
 * Input: host ARGB [planes][h][w] (copied) or a device pointer that stays owned by the caller until fic_rgb_ctx_sync.
 * fic_rgb_ctx_encode is asynchronous on hip_stream; with_collage != 0 also builds getBestGeneratedCollageRGB (:308-347).
 * Results: [planes][N_r] arrays ([planes][N_r][5] qrows5, [planes][h][w] collage); any pointer may be NULL.
 * fic_rgb_ctx_decode_host runs decodeRGB (:430-508) from the context's quantised rows: argb_out [planes][h][w],
 * avg_error_out / iterations_out [planes] (may be NULL). */
typedef struct fic_rgb_ctx fic_rgb_ctx;
FIC_API fic_rgb_ctx* fic_rgb_ctx_create(int device, int w, int h, int B, int wK, int planes);
FIC_API void fic_rgb_ctx_destroy(fic_rgb_ctx* ctx);
FIC_API int fic_rgb_ctx_set_argb_host(fic_rgb_ctx* ctx, const int32_t* argb);
FIC_API int fic_rgb_ctx_set_argb_device(fic_rgb_ctx* ctx, const void* dev_argb);
FIC_API int fic_rgb_ctx_encode(fic_rgb_ctx* ctx, int with_collage, void* hip_stream);
FIC_API int fic_rgb_ctx_sync(fic_rgb_ctx* ctx);
/* "sweep": 0 (default) = automatic, 1 = the VALU sweeps, 2 = the matrix-core full search (k_sweep_q<NK, 3>: prune on the
 * MFMA output, flagged pairs with the reference's sequential f32 sums; full search only).  All choices give the same
 * bits.  The environment variable FIC_RGB_SWEEP=1|2 overrides.  fic_rgb_ctx_last_sweep: what the last encode ran (1 / 2).
 * "chunks": pool chunks of the matrix-core sweep (0 = automatic; results do not depend on it). */
FIC_API int fic_rgb_ctx_set_option(fic_rgb_ctx* ctx, const char* name, int value);
FIC_API int fic_rgb_ctx_last_sweep(fic_rgb_ctx* ctx);
FIC_API int fic_rgb_ctx_get_results_host(fic_rgb_ctx* ctx, int32_t* idx_local, float* a, float* bR, float* bG, float* bB,
                                         int32_t* qrows5, int32_t* collage_argb);
FIC_API int fic_rgb_ctx_decode_host(fic_rgb_ctx* ctx, int32_t* argb_out, float* avg_error_out, int* iterations_out);

/* writeData, RGB branch (FractalCompression.java:230-238, 248-257): header {1,w,h,B,wK} + 5 ints per row. */
FIC_API int64_t fic_write_run_rgb(const int32_t* qrows5, int n_ranges, int w, int h, int B, int wK, uint8_t* out,
                                  int64_t capacity);

/* ---- decoder --------------------------------------------------------------------------------- */
/* FractalCompression.decode on a complete grey .run stream (FractalCompression.java:547-553 ->
 * decodeGreyScale :356-421): header, rows, calculateIndices (:853-893), then up to 50 iterations
 * of {rebuild pool from the current image, repaint every range block, accumulate the squared
 * change}, stopping when the mean change drops below 1 (:413-415).
 *   gray_out        R channel of the decoded image, w*h bytes (the reference returns grey ARGB)
 *   avg_error_io    in: FractalCompression.avgError before the call (the static is never reset,
 *                   :20,:407); out: its value after the call = the GUI's "MSE" label
 *                   (RLEAppController.java:180).  May be NULL (treated as 0).
 *   iterations      iterations executed; may be NULL.
 * avgError is Java's float accumulation (:407), one add per pixel in range-block order: taken from the exact integer sum
 * when every partial sum is an exact float, re-accumulated sequentially in that order otherwise (large or
 * non-converging decodes), so the returned value and the "< 1" decision (:413-415) match for any size. */
FIC_API int fic_decode_gray_run(const uint8_t* run, int64_t len, int device, uint8_t* gray_out, int64_t capacity,
                                int* w, int* h, float* avg_error_io, int* iterations);
/* decodeRGB (FractalCompression.java:430-508) on a complete colour .run stream (isRGB != 0):
 * rows {idx, a*1e6, bR*1e5, bG*1e5, bB} (:446-450), scaleImageRGB pool, per-channel repaint, the
 * three squared channel changes summed per pixel (:493).  argb_out: w*h ARGB ints. */
FIC_API int fic_decode_rgb_run(const uint8_t* run, int64_t len, int device, int32_t* argb_out,
                               int64_t capacity_pixels, int* w, int* h, float* avg_error_io, int* iterations);
/* The same loop driven from the context's last encode: quantised rows (and isometry ids, so
 * n_iso = 8 codebooks decode too) stay on the device.  gray_out [planes][h][w]; avg_error_out and
 * iterations_out [planes], may be NULL.  Every range block must have been encoded. */
FIC_API int fic_ctx_decode_host(fic_ctx* ctx, uint8_t* gray_out, float* avg_error_out, int* iterations_out);

/* Tuning / instrumentation knobs:
 *   "sweep"       0 auto: windowed search -> generic kernel; full search -> the VALU sweep (k_sweep_d4, the
 *                     group-Fourier form, for 8 isometries at B = 8; k_sweep_fast otherwise)
 *                 1 generic, 2 k_sweep_fast (VALU, v_dot4), 5 k_sweep_d4 (VALU, v_dot2c; n_iso = 8, B = 8 only),
 *                 3 = opt-in matrix-core sweep (B = 4/8/16, n_iso = 1 or 8, full search; same results): bf16
 *                     operands (centred pixels are exact bf16) at B = 4/8 and at B = 16 with 8 isometries, i8
 *                     operands at B = 16 with 1 isometry -- whichever is faster
 *                 4 = matrix-core sweep with i8 operands at every block size (the round's first kernels; kept)
 *                 6 = the DEFAULT full-search sweep k_sweep_q (fic_q.hip): f16 matrix-core prune GEMM on a normalised domain
 *                     operand + exact evaluation of the surviving pairs (B = 4/8/16, n_iso = 1 or 8)
 *                 (with "sweep" = 0 the environment variable FIC_SWEEP=3 selects the matrix-core sweep process-wide
 *                 for full-search launches of >= 5e7 (B = 4/8) / 5e8 (B = 16) (range, domain) pairs; smaller
 *                 launches and windowed search keep the VALU sweep, which is faster there)
 *   "q_shape"     matrix instruction of the 1-isometry k_sweep_q at B = 8 / 16: 0 by pool size (default: v_mfma_f32_16x16x32_f16
 *                 from 10^5 K-steps per range column, else 32x32x16), 1 = 16x16x32 (k_sweep_q16), 2 = 32x32x16; same codebooks
 *   "chunks"      domain-pool chunks per range tile for the fast kernel (0 = auto)
 *   "time_sweep"  1: bracket every sweep launch with hipEvents on its stream */
FIC_API int fic_ctx_set_option(fic_ctx* ctx, const char* name, int value);
/* Sum of sweep-kernel durations (ms) and launch count since the last reset ("time_sweep" = 1).
 * Synchronises the events.  reset != 0 clears the accumulators afterwards. */
FIC_API int fic_ctx_sweep_time(fic_ctx* ctx, double* total_ms, int* launches, int reset);
/* Counters of the default sweep k_sweep_q since the last reset (option "sweep_stats" = 1 first): out[0] tile epilogues
 * (32 range copies x 32 domain blocks each), out[1] tiles that had flagged pairs, out[2] pairs evaluated exactly,
 * out[3] waves; of a sample of the waves: out[4] shader-clock cycles and out[5] 100 MHz ticks they were alive (summed),
 * out[6] their number (out[4] / out[5] / 10 = the clock in GHz the chip held under this kernel); out[7] reserved.
 * Synchronises the context's stream. */
FIC_API int fic_ctx_sweep_stats(fic_ctx* ctx, uint64_t* out8, int reset);
/* Geometry actually in use: out[0..9] = Rw, Rh, N_r, Dw, Dh, N_d, NR, tiles, chunks, sweep kind. */
FIC_API int fic_ctx_info(fic_ctx* ctx, int* out10);
/* Name of the sweep kernel the context's last encode launched (as rocprofv3 prints it, without the argument list), e.g.
 * "k_sweep_q<4, 2, false>" (one pool chunk), "k_sweep_q16<4, true>" (several) or "k_sweep_qs<4, 2>" / "k_sweep_q16s<4>" (several short
 * ones: theta is shared between the chunks in the fast path) -- so that a caller can match its timing with a profile.  A small launch
 * of the default sweep (one image up to about 512x512 at B = 8) runs as two kernels instead of five -- k_prep_q8 (scale + pool +
 * range prep) and the sweep, whose last workgroup per range-column group also does k_finalize's work -- and the name then
 * carries the suffix " after k_prep_q8, finalising". */
FIC_API int fic_ctx_last_kernel(fic_ctx* ctx, char* out, int capacity);
/* Range blocks one wave of sweep `kind` keeps in registers, i.e. how many range blocks share one read of a pool block
 * (the reuse factor between SURVEY 8(d)'s byte model and the physical traffic): k_sweep_q 32 / 128 (B = 8: 8 / 1
 * isometries), 32 / 256 (B = 4), 16 / 64 (B = 16); the VALU sweeps 64 x NR.  0: not defined for that kind. */
FIC_API int fic_sweep_ranges_per_pool_read(int kind, int B, int n_iso);

/* Test hook: out[i] = sqrt((double)(first + i)) computed on the device exactly as the pool
 * kernel does for Domainblock.variance (FractalCompression.java:677,680 Math.sqrt). */
FIC_API int fic_debug_sqrt_f64(int device, uint32_t first, uint32_t count, double* out);
/* Test hook: out[0] = the decoder's reproduction of Java's loop `avg = carry; for (i) avg += (float) vals[i];`
 * (FractalCompression.java:407) on the device. */
FIC_API int fic_debug_float_sum(int device, float carry, const uint32_t* vals, int count, float* out);
/* Segments (65 536 values) of this thread's last fic_debug_float_sum whose sum left its binade (or had a fractional carry-in)
 * and therefore took the sequential-order path instead of the precomputed segment map. */
FIC_API int fic_debug_float_sum_fallbacks(void);
/* Test hook: fic_decode_gray_run that also reports in seq_sums how many iterations took the sequential float sum. */
FIC_API int fic_debug_decode_gray_run(const uint8_t* run, int64_t len, int device, uint8_t* gray_out, int64_t capacity,
                                      float* avg_error_io, int* iterations, int* seq_sums);
/* Test hook: loads RCCL, creates one communicator per device 0..|n|-1 and (|n| >= 2) runs the codebook gather's grouped
 * send/recv pattern on dummy records, checking what arrives on device 0.  n < 0: the gather's ERROR path on |n| devices
 * (a send to a peer that does not exist must fail, leave no group open and no communicator cached; the same pattern
 * must then succeed on fresh communicators). */
FIC_API int fic_debug_rccl_selftest(int n);
/* Calls of fic_encode_gray_*_multi in this process that finished with peer copies after an RCCL failure (FIC_GATHER
 * unset / "rccl": fall back; "rccl-only": return the error; "copy": peer copies only). */
FIC_API int fic_debug_gather_fallbacks(void);
/* Test hook: copies the pool of the last encode to the host: pix u8 [planes][N_d][n],
 * sum u32 [planes][N_d], var u32 [planes][N_d], scaled u8 [planes][h/2][w/2]. NULLs allowed. */
FIC_API int fic_ctx_debug_pool_host(fic_ctx* ctx, uint8_t* pix, uint32_t* sum, uint32_t* var, uint8_t* scaled);

#ifdef __cplusplus
}
#endif
#endif /* FIC_H */
