// FicNative.java -- the only Java a maintainer adds to the reference tree (same package):
// the native binding of libfic_hip.so.  INTEGRATION.md shows the 10-line change in
// FractalCompression.encodeGrayScale that calls it.  Not compiled in this image (no JDK).
package bvk_ss19;

public final class FicNative {
    static {
        System.loadLibrary("fic_jni");   // libfic_jni.so (jni/fic_jni.c), which links libfic_hip.so
    }

    private FicNative() {}

    /** HIP devices visible to the process; 0 means "keep the pure-Java path". */
    public static native int deviceCount();

    /**
     * Replaces FractalCompression.java:119-159 (pool build + per-range search).
     *
     * @param argb      RasterImage.argb (only the R byte is read, as FractalCompression.java:596,977)
     * @param out3N     float[N_r*3] receiving {i_local, a, b} per range block = imageInfo rows
     *                  (FractalCompression.java:124,156), a and b unquantised
     * @param quant3N   int[N_r*3] receiving the ints writeData emits (FractalCompression.java:242-244), may be null
     * @throws IllegalArgumentException / RuntimeException with the library's message on any
     *         negative return code (the reference throws unchecked exceptions in the same cases)
     */
    public static native void encodeGray(int[] argb, int width, int height, int blockgroesse, int widthKernel,
                                         int device, float[] out3N, int[] quant3N);
}
