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
