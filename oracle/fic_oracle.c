/*
 * fic_oracle.c -- CPU restatement of the bvk_ss19 fractal codec arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it -- as the checker (or the timed CPU baseline), never as the
 * thing shipped.  Nothing under fractal-image-compression_amd/ links, imports
 * or calls it.
 *
 * It restates, line by line and in the same operation order, these parts of
 * the reference (paths relative to /root/reference/src/bvk_ss19/, FC =
 * FractalCompression.java, DB = Domainblock.java):
 *
 *   scaleImage                 FC:970-1007      -> fo_scale_image
 *   createCodebuch             FC:1015-1050     -> fo_create_codebuch
 *   Domainblock (grey)         DB:23-29,92-115  -> fo_block_mean / fo_block_var
 *   getMittelwert              FC:67-73         -> fo_block_mean
 *   generateKernel             FC:84-100        -> fo_generate_kernel
 *   getDomainBlockIndex        FC:516-545       -> fo_domain_block_index
 *   getRangeblock              FC:588-602       -> fo_get_rangeblock
 *   getErrorVarianceCovariance FC:655-687       -> fo_error_var_cov
 *   getBestDomainblock         FC:613-644       -> fo_best_domainblock
 *   encodeGrayScale            FC:109-162       -> fo_encode_gray
 *   writeData (grey + RGB)     FC:230-261       -> fo_write_run_gray / _rgb
 *   calculateIndices           FC:853-893       -> fo_calculate_indices
 *   decodeGreyScale            FC:356-421       -> fo_decode_gray
 *   getBestGeneratedCollage    FC:269-300       -> fo_collage_gray
 *   RGB encode (K1 pin only)   FC:171-219,697-735,760-808,901-962,1058-1093,
 *                              DB:30-41         -> fo_encode_rgb
 *
 * Java semantics kept: 32-bit int with truncating '/', float = IEEE binary32
 * with one rounding per operation (compile with -ffp-contract=off, no
 * fast-math), double only where Java promotes (FC:680, Math.sqrt), and the
 * Java (int) cast of a float (NaN -> 0, saturating) as fo_java_f2i.
 *
 * n_iso = 8 is NOT in the reference (SURVEY.md top box, item 1): it is this
 * build's extension and this file is its only definition ("parity unpinned"
 * by the reference).  n_iso = 1 is the reference path, bit for bit.
 *
 * Pinned by (tests/test_oracle_golden.py): K2 = the five Animation.gif
 * "MSE" labels on LenaGrey.png (grey encode -> quantise -> decode end to
 * end), K1 = unknown.run <-> LenaColored.jpg (1024 RGB rows; pins the index /
 * window / pool-order / cast logic the grey path shares).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FO_API __attribute__((visibility("default")))

/* ---- Java (int) cast of a float: JLS 5.1.3 -------------------------------- */
FO_API int32_t fo_java_f2i(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}

static inline int red(int32_t argb) { return (argb >> 16) & 0xff; }
static inline int green(int32_t argb) { return (argb >> 8) & 0xff; }
static inline int blue(int32_t argb) { return argb & 0xff; }
static inline int32_t grey_argb(int v)
{
    return (int32_t)(0xff000000u | ((uint32_t)v << 16) | ((uint32_t)v << 8) | (uint32_t)v);
}

/* ---- geometry -------------------------------------------------------------- */
/* FC:111-116, FC:1019-1022.  Returns 0 when the reference would run without an
 * exception for this geometry, else a negative code. */
FO_API int fo_geometry(int w, int h, int B, int* Rw, int* Rh, int* Dw, int* Dh)
{
    if (B < 4 || (B % 4) != 0) return -1;       /* abstand = B/4 must be >= 1 (FC:1019) */
    if (w <= 0 || h <= 0 || (w % 2) || (h % 2)) return -2; /* scaleImage overruns on odd sizes */
    if ((w % B) || (h % B)) return -3;
    int rw = w / B, rh = h / B;
    int dw = rw * 2 - 3, dh = rh * 2 - 3;
    if (dw < 1 || dh < 1) return -4;
    if (Rw) *Rw = rw;
    if (Rh) *Rh = rh;
    if (Dw) *Dw = dw;
    if (Dh) *Dh = dh;
    return 0;
}

/* ---- scaleImage FC:970-1007 ------------------------------------------------ */
/* out has (w/2)*(h/2) entries, grey ARGB.  Note FC:993 compares x+1 against
 * image.HEIGHT (not width): kept. */
FO_API void fo_scale_image(const int32_t* argb, int w, int h, int32_t* out)
{
    int i = 0;
    for (int y = 0; y < h; y += 2) {
        for (int x = 0; x < w; x += 2) {
            int m = red(argb[x + y * w]);
            if (x + 1 >= w) {
                m += 128;
            } else {
                m += red(argb[x + 1 + y * w]);
                if (y + 1 >= h)
                    m += 128;
                else
                    m += red(argb[x + (y + 1) * w]);
            }
            if (y + 1 >= h)
                m += 128;
            else {
                if (x + 1 >= h)
                    m += 128;
                else
                    m += red(argb[x + 1 + (y + 1) * w]);
            }
            m = m / 4;
            out[i] = grey_argb(m);
            i++;
        }
    }
}

/* ---- Domainblock grey: setMittelwert DB:92-98 / getMittelwert FC:67-73 ------ */
static int fo_block_mean(const int32_t* v, int n)
{
    int sum = 0;
    for (int i = 0; i < n; i++) sum += v[i];
    return sum / n;
}
/* setVarianz DB:106-115: float accumulation of (d - mean)^2 */
static float fo_block_var(int mean, const int32_t* v, int n)
{
    float var = 0;
    for (int i = 0; i < n; i++) {
        float g = (float)(v[i] - mean);
        var += g * g;
    }
    return var;
}

/* ---- createCodebuch FC:1015-1050 ------------------------------------------- */
typedef struct {
    int count;       /* number of domain blocks */
    int n;           /* B*B */
    int32_t* pix;    /* [count][n] grey values, index rx + ry*B  (FC:1036) */
    int32_t* mean;   /* Domainblock.mittelWert */
    float* var;      /* Domainblock.variance  */
} fo_codebook;

static void fo_codebook_free(fo_codebook* cb)
{
    free(cb->pix);
    free(cb->mean);
    free(cb->var);
    memset(cb, 0, sizeof(*cb));
}

static int fo_create_codebuch(const int32_t* argb, int w, int h, int B, fo_codebook* cb)
{
    int sw = w / 2, sh = h / 2;
    int32_t* scaled = (int32_t*)malloc(sizeof(int32_t) * (size_t)sw * sh);
    if (!scaled) return -10;
    fo_scale_image(argb, w, h, scaled);
    int abstand = B / 4;
    int n = B * B;
    int count = (sw / abstand - 3) * (sh / abstand - 3);
    cb->count = count;
    cb->n = n;
    cb->pix = (int32_t*)malloc(sizeof(int32_t) * (size_t)count * n);
    cb->mean = (int32_t*)malloc(sizeof(int32_t) * (size_t)count);
    cb->var = (float*)malloc(sizeof(float) * (size_t)count);
    if (!cb->pix || !cb->mean || !cb->var) { free(scaled); return -10; }
    int i = 0;
    for (int y = 0; y < sh; y += abstand) {
        for (int x = 0; x < sw; x += abstand) {
            if (y + B <= sh && x + B <= sw) {
                if (i >= count) { free(scaled); return -11; } /* Java: AIOOBE */
                int32_t* blk = cb->pix + (size_t)i * n;
                for (int ry = 0; ry < B; ry++)
                    for (int rx = 0; rx < B; rx++)
                        blk[rx + ry * B] = red(scaled[x + rx + (y + ry) * sw]);
                cb->mean[i] = fo_block_mean(blk, n);
                cb->var[i] = fo_block_var(cb->mean[i], blk, n);
                i++;
            }
        }
    }
    free(scaled);
    if (i != count) return -11;
    return 0;
}

/* Exposed for tests of the pool kernel: fills pix (u8 widened to int32), mean, var. */
FO_API int fo_pool(const int32_t* argb, int w, int h, int B, int32_t* pix, int32_t* mean, float* var)
{
    int rc = fo_geometry(w, h, B, 0, 0, 0, 0);
    if (rc) return rc;
    fo_codebook cb;
    rc = fo_create_codebuch(argb, w, h, B, &cb);
    if (rc) { fo_codebook_free(&cb); return rc; }
    if (pix) memcpy(pix, cb.pix, sizeof(int32_t) * (size_t)cb.count * cb.n);
    if (mean) memcpy(mean, cb.mean, sizeof(int32_t) * (size_t)cb.count);
    if (var) memcpy(var, cb.var, sizeof(float) * (size_t)cb.count);
    fo_codebook_free(&cb);
    return 0;
}

/* ---- generateKernel FC:84-100 ---------------------------------------------- */
FO_API void fo_generate_kernel(int Dw, int Dh, int index, int wK, int* dy_out, int* dx_out)
{
    int dy = (int)(index / Dw) - wK / 2;
    int dx = index % Dw - wK / 2;
    if (dx < 0) dx = 0;
    if (dy < 0) dy = 0;
    if (dx + wK >= Dw) dx = Dw - wK;
    if (dy + wK >= Dh) dy = Dh - wK;
    *dy_out = dy;
    *dx_out = dx;
}

/* ---- getDomainBlockIndex FC:516-545 ---------------------------------------- */
FO_API int fo_domain_block_index(int x, int y, int Rw, int Rh, int Dw, int B)
{
    int xr = x / B;
    int yr = y / B;
    int i = 0;
    if (yr == 0) yr = 1;
    if (xr == 0) xr = 1;
    if (yr == Rh - 1) yr = yr - 1;
    if (xr == Rw - 1) xr = xr - 1;
    if (xr > 1) {
        if (yr == 0)
            i = xr;
        else
            i = (xr * 2) - 2 + (yr + yr - 1) * Dw;
    } else if (xr == 1) {
        if (yr == 0)
            i = xr;
        else
            i = xr + (yr + yr - 1) * Dw;
    }
    return i;
}

/* ---- getRangeblock FC:588-602 ---------------------------------------------- */
static void fo_get_rangeblock(const int32_t* argb, int w, int h, int x, int y, int B, int32_t* out)
{
    int i = 0;
    memset(out, 0, sizeof(int32_t) * (size_t)B * B);
    for (int ry = 0; ry < B && y + ry < h; ry++)
        for (int rx = 0; rx < B && x + rx < w; rx++) {
            out[i] = red(argb[(x + rx) + (y + ry) * w]);
            i++;
        }
}

/* ---- getErrorVarianceCovariance FC:655-687 ---------------------------------- */
/* result = { error, kovarianz, varianzSquare, rangeMittelwert, domainM } */
static void fo_error_var_cov(const int32_t* range, int rangeM, const int32_t* domain, int domMean,
                             float domVar, int n, float result[5])
{
    float domainM = (float)domMean;
    float kovarianz = 0;
    float varianzRange = 0;
    float varianzSquare = domVar;
    for (int i = 0; i < n; i++) {
        float greyR = (float)(range[i] - rangeM);
        float greyD = (float)domain[i] - domainM;
        kovarianz += greyR * greyD;
        varianzRange += greyR;
    }
    float r = 0;
    float error = 0;
    if (varianzRange == 0 || sqrt((double)varianzSquare) == 0)
        r = 0;
    else
        r = (float)((double)kovarianz / ((double)varianzRange * sqrt((double)varianzSquare)));
    r = r * r;
    error = (varianzRange * varianzRange) * (1 - r);
    result[0] = error;
    result[1] = kovarianz;
    result[2] = varianzSquare;
    result[3] = (float)rangeM;
    result[4] = domainM;
}

/* ---- the 8 isometries of the square (extension; k = 0 is the identity) ------
 * out[y][x] = d[sy][sx]:
 *   0 identity (x,y)            1 rot 90 cw  (sx=y,     sy=B-1-x)
 *   2 rot 180 (B-1-x,B-1-y)     3 rot 270 cw (sx=B-1-y, sy=x)
 *   4 mirror left-right (B-1-x,y)   5 mirror top-bottom (x,B-1-y)
 *   6 transpose (sx=y,sy=x)     7 anti-transpose (sx=B-1-y, sy=B-1-x)      */
FO_API int fo_iso_source(int k, int B, int x, int y)
{
    int sx, sy, m = B - 1;
    switch (k) {
    default:
    case 0: sx = x;     sy = y;     break;
    case 1: sx = y;     sy = m - x; break;
    case 2: sx = m - x; sy = m - y; break;
    case 3: sx = m - y; sy = x;     break;
    case 4: sx = m - x; sy = y;     break;
    case 5: sx = x;     sy = m - y; break;
    case 6: sx = y;     sy = x;     break;
    case 7: sx = m - y; sy = m - x; break;
    }
    return sx + sy * B;
}

static void fo_apply_iso(int k, int B, const int32_t* d, int32_t* out)
{
    for (int y = 0; y < B; y++)
        for (int x = 0; x < B; x++) out[x + y * B] = d[fo_iso_source(k, B, x, y)];
}

/* ---- getBestDomainblock FC:613-644 (+ isometry-minor extension) ------------- */
/* result = { i_local, a, b };  *iso_out = winning isometry (0 when n_iso==1).
 * Candidate order: window index i ascending, then k ascending; strict '<'
 * (FC:627) so the first candidate in that order wins a tie. */
static void fo_best_domainblock(const fo_codebook* cb, const int* indices, int ncand,
                                const int32_t* range, int rangeM, int B, int n_iso, float result[3],
                                int* iso_out, float* err_out)
{
    int n = B * B;
    float smallestError = 10000000;
    float bestBlock[6] = {0, 0, 0, 0, 0, 0};
    int bestIso = 0;
    int32_t* tmp = (n_iso > 1) ? (int32_t*)malloc(sizeof(int32_t) * (size_t)n) : 0;
    for (int i = 0; i < ncand; i++) {
        int g = indices[i];
        const int32_t* dom = cb->pix + (size_t)g * n;
        for (int k = 0; k < n_iso; k++) {
            const int32_t* d = dom;
            if (k > 0) { fo_apply_iso(k, B, dom, tmp); d = tmp; }
            float ab[5];
            fo_error_var_cov(range, rangeM, d, cb->mean[g], cb->var[g], n, ab);
            float error = ab[0];
            if (error < smallestError) {
                smallestError = error;
                bestBlock[0] = (float)i;
                bestBlock[1] = ab[0];
                bestBlock[2] = ab[1];
                bestBlock[3] = ab[2];
                bestBlock[4] = ab[3];
                bestBlock[5] = ab[4];
                bestIso = k;
            }
        }
    }
    free(tmp);
    float a = bestBlock[2] / bestBlock[3];
    if (a < -1)
        a = -1;
    else if (a > 1)
        a = 1;
    float b = bestBlock[4] - a * bestBlock[5];
    result[0] = bestBlock[0];
    result[1] = a;
    result[2] = b;
    if (iso_out) *iso_out = bestIso;
    if (err_out) *err_out = smallestError;
}

/* ---- encodeGrayScale FC:109-162 (search part; writeData is separate) -------- */
/* info: float[N_r][3] = {i_local, a, b} (FC:124,156); iso: int[N_r] or NULL;
 * err: float[N_r] or NULL (winning error, for diagnostics).
 * Only ranges j in [r0, r1) are computed (others left untouched): the loop is
 * independent per range, so this is the same arithmetic. */
FO_API int fo_encode_gray(const int32_t* argb, int w, int h, int B, int wK, int n_iso, int r0, int r1,
                          float* info, int32_t* iso, float* err)
{
    int Rw, Rh, Dw, Dh;
    int rc = fo_geometry(w, h, B, &Rw, &Rh, &Dw, &Dh);
    if (rc) return rc;
    if (wK < 1 || wK > Dw || wK > Dh) return -5; /* Java: negative index -> AIOOBE */
    if (n_iso != 1 && n_iso != 8) return -6;
    int n = B * B;
    fo_codebook cb;
    rc = fo_create_codebuch(argb, w, h, B, &cb);
    if (rc) { fo_codebook_free(&cb); return rc; }
    int* indices = (int*)malloc(sizeof(int) * (size_t)wK * wK);
    int32_t* rangeBlock = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int j = 0;
    for (int y = 0; y < h; y += B) {
        for (int x = 0; x < w; x += B) {
            if (j >= r0 && j < r1) {
                int i = fo_domain_block_index(x, y, Rw, Rh, Dw, B);
                int dy, dx;
                fo_generate_kernel(Dw, Dh, i, wK, &dy, &dx);
                int nn = 0;
                for (int ky = 0; ky < wK; ky++)
                    for (int kx = 0; kx < wK; kx++) {
                        indices[nn] = dx + kx + (dy + ky) * Dw;
                        nn++;
                    }
                fo_get_rangeblock(argb, w, h, x, y, B, rangeBlock);
                int rangeM = fo_block_mean(rangeBlock, n);
                int k = 0;
                float e = 0;
                fo_best_domainblock(&cb, indices, wK * wK, rangeBlock, rangeM, B, n_iso, info + 3 * (size_t)j,
                                    &k, &e);
                if (iso) iso[j] = k;
                if (err) err[j] = e;
            }
            j++;
        }
    }
    free(indices);
    free(rangeBlock);
    fo_codebook_free(&cb);
    return 0;
}

/* ---- writeData, grey branch FC:230-246,259 ---------------------------------- */
/* out must hold 20 + 12*N_r bytes.  Big-endian (DataOutputStream.writeInt). */
static uint8_t* put_be32(uint8_t* p, int32_t v)
{
    uint32_t u = (uint32_t)v;
    p[0] = (uint8_t)(u >> 24);
    p[1] = (uint8_t)(u >> 16);
    p[2] = (uint8_t)(u >> 8);
    p[3] = (uint8_t)u;
    return p + 4;
}
static int32_t get_be32(const uint8_t* p)
{
    return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
}

FO_API int64_t fo_write_run_gray(const float* info, int n_ranges, int w, int h, int B, int wK, uint8_t* out)
{
    uint8_t* p = out;
    p = put_be32(p, 0);
    p = put_be32(p, w);
    p = put_be32(p, h);
    p = put_be32(p, B);
    p = put_be32(p, wK);
    for (int row = 0; row < n_ranges; row++) {
        p = put_be32(p, fo_java_f2i(info[3 * row + 0]));
        p = put_be32(p, fo_java_f2i(info[3 * row + 1] * 100));
        p = put_be32(p, fo_java_f2i(info[3 * row + 2]));
    }
    return (int64_t)(p - out);
}

/* ---- calculateIndices FC:853-893 (mutates data[i*stride+0] local -> global) -- */
FO_API void fo_calculate_indices(float* data, int stride, int w, int h, int B, int wK)
{
    int Rw = w / B, Rh = h / B;
    int Dw = Rw * 2 - 3, Dh = Rh * 2 - 3;
    int i = 0;
    for (int y = 0; y < h; y += B) {
        for (int x = 0; x < w; x += B) {
            int di = fo_domain_block_index(x, y, Rw, Rh, Dw, B);
            int dy = (int)(di / Dw) - wK / 2;
            int dx = di % Dw - wK / 2;
            if (dx < 0) dx = 0;
            if (dy < 0) dy = 0;
            if (dx + wK >= Dw) dx = Dw - wK;
            if (dy + wK >= Dh) dy = Dh - wK;
            float v = data[(size_t)i * stride];
            int yd = fo_java_f2i(v / (float)wK);
            int xd = fo_java_f2i(fmodf(v, (float)wK)); /* Java float % == C fmodf */
            int result = xd + dx + (yd + dy) * Dw;
            data[(size_t)i * stride] = (float)result;
            i++;
        }
    }
}

/* ---- decodeGreyScale FC:356-421 --------------------------------------------- */
/* run = the full .run stream including the leading isRGB int (FC:548).
 * out_argb: w*h ints.  avg_error_io: in = FractalCompression.avgError before
 * the call (a static, never reset: FC:20,407), out = its value after.
 * iters_out: number of iterations executed. */
FO_API int fo_decode_gray(const uint8_t* run, int64_t len, int32_t* out_argb, int out_capacity,
                          float* avg_error_io, int* iters_out, int* w_out, int* h_out)
{
    if (len < 20) return -20;
    if (get_be32(run) != 0) return -21;
    int width = get_be32(run + 4), height = get_be32(run + 8);
    int B = get_be32(run + 12), wK = get_be32(run + 16);
    int Rw, Rh, Dw, Dh;
    int rc = fo_geometry(width, height, B, &Rw, &Rh, &Dw, &Dh);
    if (rc) return rc;
    if (w_out) *w_out = width;
    if (h_out) *h_out = height;
    if ((int64_t)width * height > out_capacity) return -22;
    int nr = Rw * Rh;
    if (len < 20 + 12 * (int64_t)nr) return -23;
    for (int i = 0; i < width * height; i++) out_argb[i] = grey_argb(128); /* generateGrayImage FC:1142 */
    float* imgData = (float*)malloc(sizeof(float) * 3 * (size_t)nr);
    const uint8_t* p = run + 20;
    for (int rows = 0; rows < nr; rows++) {
        imgData[3 * rows + 0] = (float)get_be32(p);
        imgData[3 * rows + 1] = (float)get_be32(p + 4) / 100.0f;
        imgData[3 * rows + 2] = (float)get_be32(p + 8);
        p += 12;
    }
    fo_calculate_indices(imgData, 3, width, height, B, wK);
    float avgError = avg_error_io ? *avg_error_io : 0.0f;
    int iters = 0;
    int n = B * B;
    for (int counter = 0; counter < 50; counter++) {
        fo_codebook cb;
        rc = fo_create_codebuch(out_argb, width, height, B, &cb);
        if (rc) { fo_codebook_free(&cb); free(imgData); return rc; }
        int i = 0;
        for (int y = 0; y < height; y += B) {
            for (int x = 0; x < width; x += B) {
                int g = fo_java_f2i(imgData[3 * i + 0]);
                if (g < 0 || g >= cb.count) { fo_codebook_free(&cb); free(imgData); return -24; }
                for (int ry = 0; ry < B && y + ry < height; ry++) {
                    for (int rx = 0; rx < B && x + rx < width; rx++) {
                        int range = red(out_argb[x + rx + (y + ry) * width]);
                        int domain = cb.pix[(size_t)g * n + rx + ry * B];
                        int value = fo_java_f2i(imgData[3 * i + 1] * (float)domain + imgData[3 * i + 2]);
                        if (value < 0)
                            value = 0;
                        else if (value > 255)
                            value = 255;
                        out_argb[x + rx + (y + ry) * width] = grey_argb(value);
                        avgError += (float)((range - value) * (range - value));
                    }
                }
                i++;
            }
        }
        fo_codebook_free(&cb);
        iters = counter + 1;
        avgError = avgError / (float)(width * height);
        if (avgError < 1) break;
        if (counter != 49) avgError = 0;
    }
    free(imgData);
    if (avg_error_io) *avg_error_io = avgError;
    if (iters_out) *iters_out = iters;
    return 0;
}

/* ---- decoder for the n_iso = 8 extension --------------------------------------------------
 * Same loop as fo_decode_gray, driven from quantised rows {i_local,(int)(a*100),(int)b} plus an
 * isometry id per range: the domain pixel is read through iso_k (out[y][x] = d[sy][sx]).
 * With iso == NULL (or all zero) it must equal fo_decode_gray bit for bit (tested). */
FO_API int fo_decode_rows(const int32_t* qrows, const int32_t* iso, int width, int height, int B, int wK,
                          int32_t* out_argb, float* avg_error_io, int* iters_out)
{
    int Rw, Rh, Dw, Dh;
    int rc = fo_geometry(width, height, B, &Rw, &Rh, &Dw, &Dh);
    if (rc) return rc;
    int nr = Rw * Rh, n = B * B;
    for (int i = 0; i < width * height; i++) out_argb[i] = grey_argb(128);
    float* imgData = (float*)malloc(sizeof(float) * 3 * (size_t)nr);
    for (int rows = 0; rows < nr; rows++) {
        imgData[3 * rows + 0] = (float)qrows[3 * rows + 0];
        imgData[3 * rows + 1] = (float)qrows[3 * rows + 1] / 100.0f;
        imgData[3 * rows + 2] = (float)qrows[3 * rows + 2];
    }
    fo_calculate_indices(imgData, 3, width, height, B, wK);
    float avgError = avg_error_io ? *avg_error_io : 0.0f;
    int iters = 0;
    for (int counter = 0; counter < 50; counter++) {
        fo_codebook cb;
        rc = fo_create_codebuch(out_argb, width, height, B, &cb);
        if (rc) { fo_codebook_free(&cb); free(imgData); return rc; }
        int i = 0;
        for (int y = 0; y < height; y += B)
            for (int x = 0; x < width; x += B) {
                int g = fo_java_f2i(imgData[3 * i + 0]);
                if (g < 0 || g >= cb.count) { fo_codebook_free(&cb); free(imgData); return -24; }
                int k = iso ? iso[i] : 0;
                for (int ry = 0; ry < B; ry++)
                    for (int rx = 0; rx < B; rx++) {
                        int range = red(out_argb[x + rx + (y + ry) * width]);
                        int domain = cb.pix[(size_t)g * n + fo_iso_source(k, B, rx, ry)];
                        int value = fo_java_f2i(imgData[3 * i + 1] * (float)domain + imgData[3 * i + 2]);
                        if (value < 0) value = 0;
                        else if (value > 255) value = 255;
                        out_argb[x + rx + (y + ry) * width] = grey_argb(value);
                        avgError += (float)((range - value) * (range - value));
                    }
                i++;
            }
        fo_codebook_free(&cb);
        iters = counter + 1;
        avgError = avgError / (float)(width * height);
        if (avgError < 1) break;
        if (counter != 49) avgError = 0;
    }
    free(imgData);
    if (avg_error_io) *avg_error_io = avgError;
    if (iters_out) *iters_out = iters;
    return 0;
}

/* ---- getBestGeneratedCollage FC:269-300 -------------------------------------- */
/* info is float[N_r][3] as produced by the encoder (local idx, unquantised a,b).
 * Like the reference it MUTATES info[.][0] to the global index (FC:273). */
FO_API int fo_collage_gray(const int32_t* argb, int w, int h, int B, int wK, float* info, int32_t* out_argb)
{
    int rc = fo_geometry(w, h, B, 0, 0, 0, 0);
    if (rc) return rc;
    fo_calculate_indices(info, 3, w, h, B, wK);
    fo_codebook cb;
    rc = fo_create_codebuch(argb, w, h, B, &cb);
    if (rc) { fo_codebook_free(&cb); return rc; }
    int n = B * B, i = 0;
    for (int k = 0; k < w * h; k++) out_argb[k] = 0; /* new RasterImage: zeroed ints */
    for (int y = 0; y < h; y += B) {
        for (int x = 0; x < w; x += B) {
            int g = fo_java_f2i(info[3 * i + 0]);
            for (int ry = 0; ry < B && y + ry < h; ry++)
                for (int rx = 0; rx < B && x + rx < w; rx++) {
                    int domain = cb.pix[(size_t)g * n + rx + ry * B];
                    int value = fo_java_f2i(info[3 * i + 1] * (float)domain + info[3 * i + 2]);
                    if (value < 0)
                        value = 0;
                    else if (value > 255)
                        value = 255;
                    out_argb[x + rx + (y + ry) * w] = grey_argb(value);
                }
            i++;
        }
    }
    fo_codebook_free(&cb);
    return 0;
}

/* ---- isGreyScale FC:32-45 ---------------------------------------------------- */
FO_API int fo_is_greyscale(const int32_t* argb, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int r = red(argb[x + y * w]), g = green(argb[x + y * w]), b = blue(argb[x + y * w]);
            if (r != g || g != b || b != r) return 0;
        }
    return 1;
}

/* ============================================================================
 * RGB joint encode -- restated ONLY to pin the oracle against K1 (unknown.run).
 * Not on the GPU hot path this round (SURVEY 8f "next #1").
 * ==========================================================================*/

/* scaleImageRGB FC:901-962.  Note FC:940-948: the 4th tap re-reads (x, y+1). */
static void fo_scale_image_rgb(const int32_t* argb, int w, int h, int32_t* out)
{
    int i = 0;
    for (int y = 0; y < h; y += 2) {
        for (int x = 0; x < w; x += 2) {
            int mR = red(argb[x + y * w]), mG = green(argb[x + y * w]), mB = blue(argb[x + y * w]);
            if (x + 1 >= w) {
                mR += 128; mG += 128; mB += 128;
            } else {
                mR += red(argb[x + 1 + y * w]);
                mG += green(argb[x + 1 + y * w]);
                mB += blue(argb[x + 1 + y * w]);
                if (y + 1 >= h) {
                    mR += 128; mG += 128; mB += 128;
                } else {
                    mR += red(argb[x + (y + 1) * w]);
                    mG += green(argb[x + (y + 1) * w]);
                    mB += blue(argb[x + (y + 1) * w]);
                }
            }
            if (y + 1 >= h) {
                mR += 128; mG += 128; mB += 128;
            } else {
                if (x + 1 >= h) {
                    mR += 128; mG += 128; mB += 128;
                } else {
                    mR += red(argb[x + (y + 1) * w]);
                    mG += green(argb[x + (y + 1) * w]);
                    mB += blue(argb[x + (y + 1) * w]);
                }
            }
            mR /= 4; mG /= 4; mB /= 4;
            out[i] = (int32_t)(0xff000000u | ((uint32_t)mR << 16) | ((uint32_t)mG << 8) | (uint32_t)mB);
            i++;
        }
    }
}

typedef struct {
    int count, n;
    int32_t* argb;          /* [count][n] packed */
    int32_t *mR, *mG, *mB;  /* mittelWertR/G/B   DB:33-39 */
    float *vR, *vG;         /* varianceR/G       DB:34,37 */
} fo_codebook_rgb;

static void fo_codebook_rgb_free(fo_codebook_rgb* cb)
{
    free(cb->argb); free(cb->mR); free(cb->mG); free(cb->mB); free(cb->vR); free(cb->vG);
    memset(cb, 0, sizeof(*cb));
}

static int chan(int32_t v, int c) { return c == 0 ? red(v) : c == 1 ? green(v) : blue(v); }

/* createCodebuchRGB FC:1058-1093 + Domainblock RGB ctor DB:30-41 */
static int fo_create_codebuch_rgb(const int32_t* argb, int w, int h, int B, fo_codebook_rgb* cb)
{
    int sw = w / 2, sh = h / 2, n = B * B, abstand = B / 4;
    int32_t* scaled = (int32_t*)malloc(sizeof(int32_t) * (size_t)sw * sh);
    fo_scale_image_rgb(argb, w, h, scaled);
    int count = (sw / abstand - 3) * (sh / abstand - 3);
    cb->count = count; cb->n = n;
    cb->argb = (int32_t*)malloc(sizeof(int32_t) * (size_t)count * n);
    cb->mR = (int32_t*)malloc(sizeof(int32_t) * (size_t)count);
    cb->mG = (int32_t*)malloc(sizeof(int32_t) * (size_t)count);
    cb->mB = (int32_t*)malloc(sizeof(int32_t) * (size_t)count);
    cb->vR = (float*)malloc(sizeof(float) * (size_t)count);
    cb->vG = (float*)malloc(sizeof(float) * (size_t)count);
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int i = 0;
    for (int y = 0; y < sh; y += abstand)
        for (int x = 0; x < sw; x += abstand)
            if (y + B <= sh && x + B <= sw) {
                if (i >= count) { free(tmp); free(scaled); return -11; }
                int32_t* blk = cb->argb + (size_t)i * n;
                for (int ry = 0; ry < B; ry++)
                    for (int rx = 0; rx < B; rx++) {
                        int32_t v = scaled[x + rx + (y + ry) * sw];
                        blk[rx + ry * B] =
                            (int32_t)(0xff000000u | ((uint32_t)red(v) << 16) | ((uint32_t)green(v) << 8) | (uint32_t)blue(v));
                    }
                for (int c = 0; c < 3; c++) {
                    for (int k = 0; k < n; k++) tmp[k] = chan(blk[k], c);
                    int m = fo_block_mean(tmp, n);
                    if (c == 0) { cb->mR[i] = m; cb->vR[i] = fo_block_var(m, tmp, n); }
                    if (c == 1) { cb->mG[i] = m; cb->vG[i] = fo_block_var(m, tmp, n); }
                    if (c == 2) { cb->mB[i] = m; }
                }
                i++;
            }
    free(tmp);
    free(scaled);
    return i == count ? 0 : -11;
}

/* getErrorVarianceCovarianceRGB FC:760-808; result[9] */
static void fo_error_var_cov_rgb(const int32_t* domain, const int32_t* range, const fo_codebook_rgb* cb, int g,
                                 int n, float result[9])
{
    float domainR = (float)cb->mR[g], domainG = (float)cb->mG[g], domainB = (float)cb->mB[g];
    int sR = 0, sG = 0, sB = 0;
    for (int i = 0; i < n; i++) { sR += red(range[i]); sG += green(range[i]); sB += blue(range[i]); }
    int rangeRM = sR / n, rangeGM = sG / n, rangeBM = sB / n;
    float kovarianz = 0;
    /* FC:776: varianceR + varianceG + mittelWertB (sic) */
    float varianzSquare = (cb->vR[g] + cb->vG[g] + (float)cb->mB[g]);
    float varianzRange = 0;
    /* FC:778: Domainblock.variance is never set on the RGB branch (DB:30-41) -> 0 */
    float varianzDomain = (float)sqrt((double)0.0f);
    for (int i = 0; i < n; i++) {
        float greyD = ((float)red(domain[i]) - domainR) + ((float)green(domain[i]) - domainG) +
                      ((float)blue(domain[i]) - domainB);
        float greyR = (float)((red(range[i]) - rangeRM) + (green(range[i]) - rangeGM) + (blue(range[i]) - rangeBM));
        kovarianz += greyR * greyD;
        varianzRange += greyR;
        varianzDomain += greyD;
    }
    float r = 0, error = 0;
    if (varianzRange == 0 || varianzDomain == 0)
        r = 0;
    else
        r = kovarianz / (varianzRange * varianzDomain);
    r = r * r;
    error = (varianzRange * varianzRange) * (1 - r);
    result[0] = error; result[1] = kovarianz; result[2] = varianzSquare;
    result[3] = (float)rangeRM; result[4] = domainR;
    result[5] = (float)rangeGM; result[6] = domainG;
    result[7] = (float)rangeBM; result[8] = domainB;
}

/* encodeRGB FC:171-219 + getBestDomainblockRGB FC:697-735; info float[N_r][5] */
FO_API int fo_encode_rgb(const int32_t* argb, int w, int h, int B, int wK, float* info)
{
    int Rw, Rh, Dw, Dh;
    int rc = fo_geometry(w, h, B, &Rw, &Rh, &Dw, &Dh);
    if (rc) return rc;
    if (wK < 1 || wK > Dw || wK > Dh) return -5;
    int n = B * B;
    fo_codebook_rgb cb;
    rc = fo_create_codebuch_rgb(argb, w, h, B, &cb);
    if (rc) { fo_codebook_rgb_free(&cb); return rc; }
    int32_t* rangeblock = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int j = 0;
    for (int y = 0; y < h; y += B) {
        for (int x = 0; x < w; x += B) {
            int i0 = fo_domain_block_index(x, y, Rw, Rh, Dw, B);
            int dy, dx;
            fo_generate_kernel(Dw, Dh, i0, wK, &dy, &dx);
            memset(rangeblock, 0, sizeof(int32_t) * (size_t)n);
            int q = 0;
            for (int ry = 0; ry < B && y + ry < h; ry++)
                for (int rx = 0; rx < B && x + rx < w; rx++) rangeblock[q++] = argb[(x + rx) + (y + ry) * w];
            float smallestError = 10000000;
            float best[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            int c = 0;
            for (int ky = 0; ky < wK; ky++)
                for (int kx = 0; kx < wK; kx++) {
                    int g = dx + kx + (dy + ky) * Dw;
                    float ab[9];
                    fo_error_var_cov_rgb(cb.argb + (size_t)g * n, rangeblock, &cb, g, n, ab);
                    if (ab[0] < smallestError) {
                        smallestError = ab[0];
                        best[0] = (float)c;
                        for (int t = 0; t < 9; t++) best[1 + t] = ab[t];
                    }
                    c++;
                }
            float a = best[2] / best[3];
            if (a > 1) a = 1;
            if (a < -1) a = -1;
            float bR = best[4] - a * best[5];
            float bG = best[6] - a * best[7];
            float bB = best[8] - a * best[9];
            float* o = info + 5 * (size_t)j;
            o[0] = best[0]; o[1] = a; o[2] = bR; o[3] = bG; o[4] = bB;
            j++;
        }
    }
    free(rangeblock);
    fo_codebook_rgb_free(&cb);
    return 0;
}

/* getBestGeneratedCollageRGB FC:308-347; info float[N_r][5], its column 0 is mutated to the global index */
FO_API int fo_collage_rgb(const int32_t* argb, int w, int h, int B, int wK, float* info, int32_t* out_argb)
{
    int rc = fo_geometry(w, h, B, 0, 0, 0, 0);
    if (rc) return rc;
    fo_calculate_indices(info, 5, w, h, B, wK);
    fo_codebook_rgb cb;
    rc = fo_create_codebuch_rgb(argb, w, h, B, &cb);
    if (rc) { fo_codebook_rgb_free(&cb); return rc; }
    int n = B * B, i = 0;
    for (int y = 0; y < h; y += B)
        for (int x = 0; x < w; x += B) {
            int g = fo_java_f2i(info[5 * i + 0]);
            for (int ry = 0; ry < B && y + ry < h; ry++)
                for (int rx = 0; rx < B && x + rx < w; rx++) {
                    int32_t d = cb.argb[(size_t)g * n + rx + ry * B];
                    int vR = fo_java_f2i(info[5 * i + 1] * (float)red(d) + info[5 * i + 2]);
                    int vG = fo_java_f2i(info[5 * i + 1] * (float)green(d) + info[5 * i + 3]);
                    int vB = fo_java_f2i(info[5 * i + 1] * (float)blue(d) + info[5 * i + 4]);
                    vR = vR < 0 ? 0 : (vR > 255 ? 255 : vR);
                    vG = vG < 0 ? 0 : (vG > 255 ? 255 : vG);
                    vB = vB < 0 ? 0 : (vB > 255 ? 255 : vB);
                    out_argb[x + rx + (y + ry) * w] =
                        (int32_t)(0xff000000u | ((uint32_t)vR << 16) | ((uint32_t)vG << 8) | (uint32_t)vB);
                }
            i++;
        }
    fo_codebook_rgb_free(&cb);
    return 0;
}

/* decodeRGB FC:430-508.  run = full stream including the leading isRGB int.  out_argb: w*h ints. */
FO_API int fo_decode_rgb(const uint8_t* run, int64_t len, int32_t* out_argb, int out_capacity, float* avg_error_io,
                         int* iters_out, int* w_out, int* h_out)
{
    if (len < 20) return -20;
    if (get_be32(run) == 0) return -21;
    int width = get_be32(run + 4), height = get_be32(run + 8);
    int B = get_be32(run + 12), wK = get_be32(run + 16);
    int Rw, Rh, Dw, Dh;
    int rc = fo_geometry(width, height, B, &Rw, &Rh, &Dw, &Dh);
    if (rc) return rc;
    if (w_out) *w_out = width;
    if (h_out) *h_out = height;
    if ((int64_t)width * height > out_capacity) return -22;
    int nr = Rw * Rh, n = B * B;
    if (len < 20 + 20 * (int64_t)nr) return -23;
    for (int i = 0; i < width * height; i++) out_argb[i] = grey_argb(128);
    float* imgData = (float*)malloc(sizeof(float) * 5 * (size_t)nr);
    const uint8_t* p = run + 20;
    for (int rows = 0; rows < nr; rows++) {
        imgData[5 * rows + 0] = (float)get_be32(p);
        imgData[5 * rows + 1] = (float)get_be32(p + 4) / 1000000.0f;
        imgData[5 * rows + 2] = (float)get_be32(p + 8) / 100000.0f;
        imgData[5 * rows + 3] = (float)get_be32(p + 12) / 100000.0f;
        imgData[5 * rows + 4] = (float)get_be32(p + 16);
        p += 20;
    }
    fo_calculate_indices(imgData, 5, width, height, B, wK);
    float avgError = avg_error_io ? *avg_error_io : 0.0f;
    int iters = 0;
    for (int counter = 0; counter < 50; counter++) {
        fo_codebook_rgb cb;
        rc = fo_create_codebuch_rgb(out_argb, width, height, B, &cb);
        if (rc) { fo_codebook_rgb_free(&cb); free(imgData); return rc; }
        int i = 0;
        for (int y = 0; y < height; y += B)
            for (int x = 0; x < width; x += B) {
                int g = fo_java_f2i(imgData[5 * i + 0]);
                if (g < 0 || g >= cb.count) { fo_codebook_rgb_free(&cb); free(imgData); return -24; }
                for (int ry = 0; ry < B && y + ry < height; ry++)
                    for (int rx = 0; rx < B && x + rx < width; rx++) {
                        int32_t cur = out_argb[x + rx + (y + ry) * width];
                        int rangeR = red(cur), rangeG = green(cur), rangeB = blue(cur);
                        int32_t d = cb.argb[(size_t)g * n + rx + ry * B];
                        int vR = fo_java_f2i(imgData[5 * i + 1] * (float)red(d) + imgData[5 * i + 2]);
                        int vG = fo_java_f2i(imgData[5 * i + 1] * (float)green(d) + imgData[5 * i + 3]);
                        int vB = fo_java_f2i(imgData[5 * i + 1] * (float)blue(d) + imgData[5 * i + 4]);
                        vR = vR < 0 ? 0 : (vR > 255 ? 255 : vR);
                        vG = vG < 0 ? 0 : (vG > 255 ? 255 : vG);
                        vB = vB < 0 ? 0 : (vB > 255 ? 255 : vB);
                        out_argb[x + rx + (y + ry) * width] =
                            (int32_t)(0xff000000u | ((uint32_t)vR << 16) | ((uint32_t)vG << 8) | (uint32_t)vB);
                        avgError += (float)((rangeR - vR) * (rangeR - vR) + (rangeG - vG) * (rangeG - vG) +
                                            (rangeB - vB) * (rangeB - vB));
                    }
                i++;
            }
        fo_codebook_rgb_free(&cb);
        iters = counter + 1;
        avgError = avgError / (float)(width * height);
        if (avgError < 1) break;
        if (counter != 49) avgError = 0;
    }
    free(imgData);
    if (avg_error_io) *avg_error_io = avgError;
    if (iters_out) *iters_out = iters;
    return 0;
}

/* writeData, RGB branch FC:248-257 */
FO_API int64_t fo_write_run_rgb(const float* info, int n_ranges, int w, int h, int B, int wK, uint8_t* out)
{
    uint8_t* p = out;
    p = put_be32(p, 1);
    p = put_be32(p, w);
    p = put_be32(p, h);
    p = put_be32(p, B);
    p = put_be32(p, wK);
    for (int row = 0; row < n_ranges; row++) {
        const float* r = info + 5 * (size_t)row;
        p = put_be32(p, fo_java_f2i(r[0]));
        p = put_be32(p, fo_java_f2i(r[1] * 1000000));
        p = put_be32(p, fo_java_f2i(r[2] * 100000));
        p = put_be32(p, fo_java_f2i(r[3] * 100000));
        p = put_be32(p, fo_java_f2i(r[4]));
    }
    return (int64_t)(p - out);
}
