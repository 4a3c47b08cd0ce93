// FicNative.java -- the only Java a maintainer adds to the reference tree (same package):
// the native binding of libfic_hip.so.  INTEGRATION.md shows the small changes in
// FractalCompression that call it.  Not compiled in this image (no JDK).
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

    /**
     * The same search sharded over the first {@code nGpus} HIP devices of the node: still one synchronous call on the
     * calling thread (FractalCompression.encode, FractalCompression.java:54-59, is one call on the JavaFX thread); range
     * blocks are independent (FractalCompression.java:125-159), the codebook rows are gathered inside the library with
     * RCCL.  The result does not depend on nGpus.
     *
     * @param nIso   1 = the reference algorithm; 8 = this build's isometry extension (rows then need isoN)
     * @param isoN   int[N_r] receiving the winning isometry id per range block (all 0 for nIso = 1), may be null
     */
    public static native void encodeGrayMulti(int[] argb, int width, int height, int blockgroesse, int widthKernel,
                                              int nIso, int nGpus, float[] out3N, int[] quant3N, int[] isoN);

    /**
     * Replaces the search of encodeRGB, FractalCompression.java:181-215.
     *
     * @param out5N  float[N_r*5] receiving imageInfoRGB rows {i_local, a, bR, bG, bB} (FractalCompression.java:185,212)
     */
    public static native void encodeRgb(int[] argb, int width, int height, int blockgroesse, int widthKernel,
                                        int device, float[] out5N);

    /**
     * decodeGreyScale / decodeRGB (FractalCompression.java:356-421 / 430-508) on a complete .run stream,
     * INCLUDING the leading isRGB int that FractalCompression.decode consumed (FractalCompression.java:548).
     *
     * @param run       the stream bytes
     * @param avgError  float[1]: in = FractalCompression.avgError before the call (the static is never reset),
     *                  out = its value afterwards
     * @return int[2 + w*h]: {width, height, argb...} of the decoded image
     */
    public static native int[] decode(byte[] run, int device, float[] avgError);
}
