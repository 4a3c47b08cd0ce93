/*
 * fic_jni.c -- JNI shim between bvk_ss19.FicNative and the C ABI of include/fic.h.
 * Build where a JDK exists (none in this image, so this file is NOT compiled by build()):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       fic_jni.c -L.. -lfic_hip -Wl,-rpath,'$ORIGIN' -o libfic_jni.so
 *
 * Ownership: Java owns every array; they are pinned with
 * Get/ReleasePrimitiveArrayCritical only around the copy-in / copy-out, the library keeps no
 * pointer past return.  Failures surface as Java exceptions, never as a JVM crash.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include "fic.h"

static void throw_fic(JNIEnv* env, int code)
{
    const char* cls = (code == FIC_E_GEOMETRY || code == FIC_E_WINDOW || code == FIC_E_ARGUMENT)
                          ? "java/lang/IllegalArgumentException"
                          : "java/lang/RuntimeException";
    jclass c = (*env)->FindClass(env, cls);
    if (c) (*env)->ThrowNew(env, c, fic_last_error());
}

JNIEXPORT jint JNICALL Java_bvk_1ss19_FicNative_deviceCount(JNIEnv* env, jclass k)
{
    (void)env; (void)k;
    return fic_device_count();
}

JNIEXPORT void JNICALL Java_bvk_1ss19_FicNative_encodeGray(JNIEnv* env, jclass k, jintArray argb, jint w, jint h,
                                                           jint B, jint wK, jint device, jfloatArray out3N,
                                                           jintArray quant3N)
{
    (void)k;
    int Rw = 0, Rh = 0;
    int rc = fic_geometry(w, h, B, &Rw, &Rh, NULL, NULL);
    if (rc) { throw_fic(env, rc); return; }
    const jsize nr = (jsize)Rw * Rh;
    if ((*env)->GetArrayLength(env, argb) < (jsize)w * h || (*env)->GetArrayLength(env, out3N) < 3 * nr ||
        (quant3N && (*env)->GetArrayLength(env, quant3N) < 3 * nr)) {
        jclass c = (*env)->FindClass(env, "java/lang/ArrayIndexOutOfBoundsException");
        if (c) (*env)->ThrowNew(env, c, "fic: array shorter than the geometry needs");
        return;
    }
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nr);
    float* a = (float*)malloc(sizeof(float) * (size_t)nr);
    float* b = (float*)malloc(sizeof(float) * (size_t)nr);
    int32_t* q = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)nr);
    if (!idx || !a || !b || !q) {
        jclass c = (*env)->FindClass(env, "java/lang/OutOfMemoryError");
        if (c) (*env)->ThrowNew(env, c, "fic: host staging");
        goto done;
    }
    {
        jint* px = (jint*)(*env)->GetPrimitiveArrayCritical(env, argb, NULL);
        if (!px) goto done;   /* OutOfMemoryError already pending */
        rc = fic_encode_gray_argb((const int32_t*)px, w, h, B, wK, 1, device, idx, a, b, NULL, q);
        (*env)->ReleasePrimitiveArrayCritical(env, argb, px, JNI_ABORT);
    }
    if (rc) { throw_fic(env, rc); goto done; }
    {
        jfloat* o = (jfloat*)(*env)->GetPrimitiveArrayCritical(env, out3N, NULL);
        if (!o) goto done;
        for (jsize j = 0; j < nr; j++) { o[3 * j] = (jfloat)idx[j]; o[3 * j + 1] = a[j]; o[3 * j + 2] = b[j]; }
        (*env)->ReleasePrimitiveArrayCritical(env, out3N, o, 0);
    }
    if (quant3N) (*env)->SetIntArrayRegion(env, quant3N, 0, 3 * nr, (const jint*)q);
done:
    free(idx); free(a); free(b); free(q);
}


JNIEXPORT void JNICALL Java_bvk_1ss19_FicNative_encodeRgb(JNIEnv* env, jclass k, jintArray argb, jint w, jint h, jint B,
                                                          jint wK, jint device, jfloatArray out5N)
{
    (void)k;
    int Rw = 0, Rh = 0;
    int rc = fic_geometry(w, h, B, &Rw, &Rh, NULL, NULL);
    if (rc) { throw_fic(env, rc); return; }
    const jsize nr = (jsize)Rw * Rh;
    if ((*env)->GetArrayLength(env, argb) < (jsize)w * h || (*env)->GetArrayLength(env, out5N) < 5 * nr) {
        jclass c = (*env)->FindClass(env, "java/lang/ArrayIndexOutOfBoundsException");
        if (c) (*env)->ThrowNew(env, c, "fic: array shorter than the geometry needs");
        return;
    }
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)nr);
    float* v = (float*)malloc(sizeof(float) * 4 * (size_t)nr);      /* a, bR, bG, bB */
    if (!idx || !v) { free(idx); free(v); return; }
    jint* px = (jint*)(*env)->GetPrimitiveArrayCritical(env, argb, NULL);
    if (px) {
        rc = fic_encode_rgb_argb((const int32_t*)px, w, h, B, wK, device, idx, v, v + nr, v + 2 * nr, v + 3 * nr, NULL, NULL);
        (*env)->ReleasePrimitiveArrayCritical(env, argb, px, JNI_ABORT);
        if (rc) {
            throw_fic(env, rc);
        } else {
            jfloat* o = (jfloat*)(*env)->GetPrimitiveArrayCritical(env, out5N, NULL);
            if (o) {
                for (jsize j = 0; j < nr; j++) {
                    o[5 * j] = (jfloat)idx[j];
                    o[5 * j + 1] = v[j]; o[5 * j + 2] = v[nr + j]; o[5 * j + 3] = v[2 * nr + j]; o[5 * j + 4] = v[3 * nr + j];
                }
                (*env)->ReleasePrimitiveArrayCritical(env, out5N, o, 0);
            }
        }
    }
    free(idx); free(v);
}

JNIEXPORT jintArray JNICALL Java_bvk_1ss19_FicNative_decode(JNIEnv* env, jclass k, jbyteArray run, jint device,
                                                             jfloatArray avgError)
{
    (void)k;
    const jsize len = (*env)->GetArrayLength(env, run);
    if (len < 20 || (*env)->GetArrayLength(env, avgError) < 1) { throw_fic(env, FIC_E_ARGUMENT); return NULL; }
    uint8_t* buf = (uint8_t*)malloc((size_t)len);
    if (!buf) return NULL;
    (*env)->GetByteArrayRegion(env, run, 0, len, (jbyte*)buf);
    const int isRGB = (buf[0] | buf[1] | buf[2] | buf[3]) != 0;
    const int w = (int)((buf[4] << 24) | (buf[5] << 16) | (buf[6] << 8) | buf[7]);
    const int h = (int)((buf[8] << 24) | (buf[9] << 16) | (buf[10] << 8) | buf[11]);
    jintArray result = NULL;
    int rc = fic_geometry(w, h, (int)((buf[12] << 24) | (buf[13] << 16) | (buf[14] << 8) | buf[15]), NULL, NULL, NULL, NULL);
    if (rc) { throw_fic(env, rc); free(buf); return NULL; }
    const size_t npix = (size_t)w * h;
    int32_t* img = (int32_t*)malloc(sizeof(int32_t) * (npix + 2));
    jfloat avg = 0;
    (*env)->GetFloatArrayRegion(env, avgError, 0, 1, &avg);
    if (img) {
        img[0] = w; img[1] = h;
        if (isRGB) {
            rc = fic_decode_rgb_run(buf, len, device, img + 2, (int64_t)npix, NULL, NULL, &avg, NULL);
        } else {
            uint8_t* g = (uint8_t*)malloc(npix);
            rc = g ? fic_decode_gray_run(buf, len, device, g, (int64_t)npix, NULL, NULL, &avg, NULL) : FIC_E_HIP;
            if (rc == FIC_OK)
                for (size_t i = 0; i < npix; i++)
                    img[2 + i] = (int32_t)(0xff000000u | ((uint32_t)g[i] << 16) | ((uint32_t)g[i] << 8) | g[i]);
            free(g);
        }
        if (rc) {
            throw_fic(env, rc);
        } else {
            (*env)->SetFloatArrayRegion(env, avgError, 0, 1, &avg);
            result = (*env)->NewIntArray(env, (jsize)(npix + 2));
            if (result) (*env)->SetIntArrayRegion(env, result, 0, (jsize)(npix + 2), (const jint*)img);
        }
    }
    free(img); free(buf);
    return result;
}
