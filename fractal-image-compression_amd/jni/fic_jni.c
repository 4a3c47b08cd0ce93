/*
 * fic_jni.c -- JNI shim between bvk_ss19.FicNative and the C ABI of include/fic.h.
 * Build where a JDK exists (none in this image, so this file is NOT built by build(); tests/test_host_cpu.py only
 * parses it against a stub jni.h):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       fic_jni.c -L.. -lfic_hip -Wl,-rpath,'$ORIGIN' -o libfic_jni.so
 *
 * Ownership: Java owns every array.  Inputs are COPIED out with Get<Type>ArrayRegion into malloc'ed
 * staging before the library is called and results are copied back with Set<Type>ArrayRegion afterwards: no JNI critical
 * region is ever held across a library call (an encode allocates device memory, copies, launches kernels and blocks on a
 * stream -- seconds at full search -- and a critical region must neither block nor run long).  The library keeps no
 * pointer past return.  Every failure, allocation failures included, surfaces as a Java exception, never as a JVM crash
 * or a silently empty result.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include "fic.h"

static void throw_new(JNIEnv* env, const char* cls, const char* msg)
{
    jclass c = (*env)->FindClass(env, cls);
    if (c) (*env)->ThrowNew(env, c, msg);
}

static void throw_fic(JNIEnv* env, int code)
{
    throw_new(env,
              (code == FIC_E_GEOMETRY || code == FIC_E_WINDOW || code == FIC_E_ARGUMENT) ? "java/lang/IllegalArgumentException"
                                                                                         : "java/lang/RuntimeException",
              fic_last_error());
}

static void throw_oom(JNIEnv* env) { throw_new(env, "java/lang/OutOfMemoryError", "fic: host staging buffer"); }

static void throw_short(JNIEnv* env) { throw_new(env, "java/lang/ArrayIndexOutOfBoundsException", "fic: array shorter than the geometry needs"); }

/* big-endian int of a .run stream (DataOutputStream.writeInt, FractalCompression.java:234-245) */
static int32_t be32(const uint8_t* p)
{
    return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]);
}

JNIEXPORT jint JNICALL Java_bvk_1ss19_FicNative_deviceCount(JNIEnv* env, jclass k)
{
    (void)env; (void)k;
    return fic_device_count();
}

/* shared body of encodeGray / encodeGrayMulti: n_gpus < 0 selects the single-device entry on `device` */
static void encode_gray_common(JNIEnv* env, jintArray argb, jint w, jint h, jint B, jint wK, jint n_iso, jint device,
                               jint n_gpus, jfloatArray out3N, jintArray quant3N, jintArray isoN)
{
    int Rw = 0, Rh = 0;
    int rc = fic_geometry(w, h, B, &Rw, &Rh, NULL, NULL);
    if (rc) { throw_fic(env, rc); return; }
    /* sizes in 64 bits first: 3 * nr and w * h overflow a jsize for very large geometries before any comparison */
    const int64_t nr64 = (int64_t)Rw * Rh, npix64 = (int64_t)w * h;
    if (3 * nr64 > 2147483647LL || npix64 > 2147483647LL) {
        throw_new(env, "java/lang/IllegalArgumentException", "fic: image too large for Java arrays (3 * N_r or w * h above 2^31 - 1)");
        return;
    }
    const jsize nr = (jsize)nr64;
    const jsize npix = (jsize)npix64;
    if ((*env)->GetArrayLength(env, argb) < npix || (*env)->GetArrayLength(env, out3N) < 3 * nr ||
        (quant3N && (*env)->GetArrayLength(env, quant3N) < 3 * nr) || (isoN && (*env)->GetArrayLength(env, isoN) < nr)) {
        throw_short(env);
        return;
    }
    int32_t* px = (int32_t*)malloc(sizeof(int32_t) * (size_t)npix);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nr);
    int32_t* iso = (int32_t*)malloc(sizeof(int32_t) * (size_t)nr);
    float* a = (float*)malloc(sizeof(float) * (size_t)nr);
    float* b = (float*)malloc(sizeof(float) * (size_t)nr);
    int32_t* q = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)nr);
    float* o = (float*)malloc(sizeof(float) * 3 * (size_t)nr);
    if (!px || !idx || !iso || !a || !b || !q || !o) { throw_oom(env); goto done; }
    (*env)->GetIntArrayRegion(env, argb, 0, npix, (jint*)px);          /* copy in; nothing pinned */
    if ((*env)->ExceptionCheck(env)) goto done;
    rc = n_gpus < 0 ? fic_encode_gray_argb(px, w, h, B, wK, n_iso, device, idx, a, b, iso, q)
                    : fic_encode_gray_argb_multi(px, w, h, B, wK, n_iso, n_gpus, idx, a, b, iso, q);
    if (rc) { throw_fic(env, rc); goto done; }
    for (jsize j = 0; j < nr; j++) { o[3 * j] = (float)idx[j]; o[3 * j + 1] = a[j]; o[3 * j + 2] = b[j]; }
    (*env)->SetFloatArrayRegion(env, out3N, 0, 3 * nr, (const jfloat*)o);
    if (quant3N) (*env)->SetIntArrayRegion(env, quant3N, 0, 3 * nr, (const jint*)q);
    if (isoN) (*env)->SetIntArrayRegion(env, isoN, 0, nr, (const jint*)iso);
done:
    free(px); free(idx); free(iso); free(a); free(b); free(q); free(o);
}

JNIEXPORT void JNICALL Java_bvk_1ss19_FicNative_encodeGray(JNIEnv* env, jclass k, jintArray argb, jint w, jint h,
                                                           jint B, jint wK, jint device, jfloatArray out3N,
                                                           jintArray quant3N)
{
    (void)k;
    encode_gray_common(env, argb, w, h, B, wK, 1, device, -1, out3N, quant3N, NULL);
}

JNIEXPORT void JNICALL Java_bvk_1ss19_FicNative_encodeGrayMulti(JNIEnv* env, jclass k, jintArray argb, jint w, jint h,
                                                                jint B, jint wK, jint nIso, jint nGpus,
                                                                jfloatArray out3N, jintArray quant3N, jintArray isoN)
{
    (void)k;
    if (nGpus < 1) { throw_new(env, "java/lang/IllegalArgumentException", "fic: nGpus < 1"); return; }
    encode_gray_common(env, argb, w, h, B, wK, nIso, 0, nGpus, out3N, quant3N, isoN);
}

JNIEXPORT void JNICALL Java_bvk_1ss19_FicNative_encodeRgb(JNIEnv* env, jclass k, jintArray argb, jint w, jint h, jint B,
                                                          jint wK, jint device, jfloatArray out5N)
{
    (void)k;
    int Rw = 0, Rh = 0;
    int rc = fic_geometry(w, h, B, &Rw, &Rh, NULL, NULL);
    if (rc) { throw_fic(env, rc); return; }
    const int64_t nr64 = (int64_t)Rw * Rh, npix64 = (int64_t)w * h;
    if (5 * nr64 > 2147483647LL || npix64 > 2147483647LL) {
        throw_new(env, "java/lang/IllegalArgumentException", "fic: image too large for Java arrays (5 * N_r or w * h above 2^31 - 1)");
        return;
    }
    const jsize nr = (jsize)nr64;
    const jsize npix = (jsize)npix64;
    if ((*env)->GetArrayLength(env, argb) < npix || (*env)->GetArrayLength(env, out5N) < 5 * nr) {
        throw_short(env);
        return;
    }
    int32_t* px = (int32_t*)malloc(sizeof(int32_t) * (size_t)npix);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nr);
    float* v = (float*)malloc(sizeof(float) * 4 * (size_t)nr);      /* a, bR, bG, bB */
    float* o = (float*)malloc(sizeof(float) * 5 * (size_t)nr);
    if (!px || !idx || !v || !o) { throw_oom(env); goto done; }
    (*env)->GetIntArrayRegion(env, argb, 0, npix, (jint*)px);
    if ((*env)->ExceptionCheck(env)) goto done;
    rc = fic_encode_rgb_argb(px, w, h, B, wK, device, idx, v, v + nr, v + 2 * nr, v + 3 * nr, NULL, NULL);
    if (rc) { throw_fic(env, rc); goto done; }
    for (jsize j = 0; j < nr; j++) {
        o[5 * j] = (float)idx[j];
        o[5 * j + 1] = v[j]; o[5 * j + 2] = v[nr + j]; o[5 * j + 3] = v[2 * nr + j]; o[5 * j + 4] = v[3 * nr + j];
    }
    (*env)->SetFloatArrayRegion(env, out5N, 0, 5 * nr, (const jfloat*)o);
done:
    free(px); free(idx); free(v); free(o);
}

JNIEXPORT jintArray JNICALL Java_bvk_1ss19_FicNative_decode(JNIEnv* env, jclass k, jbyteArray run, jint device,
                                                             jfloatArray avgError)
{
    (void)k;
    const jsize len = (*env)->GetArrayLength(env, run);
    if (len < 20 || (*env)->GetArrayLength(env, avgError) < 1) {
        throw_new(env, "java/lang/IllegalArgumentException", "fic: stream shorter than its 20-byte header, or avgError not float[1]");
        return NULL;
    }
    jintArray result = NULL;
    int32_t* img = NULL;
    uint8_t* g = NULL;
    uint8_t* buf = (uint8_t*)malloc((size_t)len);
    if (!buf) { throw_oom(env); return NULL; }
    (*env)->GetByteArrayRegion(env, run, 0, len, (jbyte*)buf);
    if ((*env)->ExceptionCheck(env)) goto done;
    {
        const int isRGB = be32(buf) != 0;
        const int w = be32(buf + 4), h = be32(buf + 8);
        int Rw = 0, Rh = 0;
        int rc = fic_geometry(w, h, be32(buf + 12), &Rw, &Rh, NULL, NULL);
        if (rc) { throw_fic(env, rc); goto done; }
        /* the header is untrusted: before any allocation sized by it, the stream must really hold its 3 (grey) or 5 (RGB)
         * ints per range block -- a 20-byte stream must not be able to ask for gigabytes */
        const int64_t need = 20 + (int64_t)(isRGB ? 20 : 12) * (int64_t)Rw * (int64_t)Rh;
        const int64_t npix64 = (int64_t)w * (int64_t)h;
        if ((int64_t)len < need || npix64 + 2 > 2147483647LL) {
            throw_new(env, "java/lang/IllegalArgumentException",
                      "fic: stream shorter than its header's geometry implies (EOFException in the reference), or image too large for int[]");
            goto done;
        }
        const size_t npix = (size_t)npix64;
        jfloat avg = 0;
        (*env)->GetFloatArrayRegion(env, avgError, 0, 1, &avg);
        img = (int32_t*)malloc(sizeof(int32_t) * (npix + 2));
        if (!img) { throw_oom(env); goto done; }
        img[0] = w; img[1] = h;
        if (isRGB) {
            rc = fic_decode_rgb_run(buf, len, device, img + 2, (int64_t)npix, NULL, NULL, &avg, NULL);
        } else {
            g = (uint8_t*)malloc(npix);
            if (!g) { throw_oom(env); goto done; }
            rc = fic_decode_gray_run(buf, len, device, g, (int64_t)npix, NULL, NULL, &avg, NULL);
            if (rc == FIC_OK)
                for (size_t i = 0; i < npix; i++)
                    img[2 + i] = (int32_t)(0xff000000u | ((uint32_t)g[i] << 16) | ((uint32_t)g[i] << 8) | g[i]);
        }
        if (rc) { throw_fic(env, rc); goto done; }
        (*env)->SetFloatArrayRegion(env, avgError, 0, 1, &avg);
        result = (*env)->NewIntArray(env, (jsize)(npix + 2));      /* NULL: OutOfMemoryError already pending */
        if (result) (*env)->SetIntArrayRegion(env, result, 0, (jsize)(npix + 2), (const jint*)img);
    }
done:
    free(g); free(img); free(buf);
    return result;
}
