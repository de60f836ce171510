package thi.ng.raymarchcl;

import java.nio.ByteBuffer;

/**
 * Static native methods of the JNI shim (bindings/jni/raymarch_jni.c) over libraymarch_hip.so.
 * Unverified in the build image (no JDK); every buffer must be a DIRECT java.nio buffer in
 * native byte order -- what thi.ng.simplecl passes to JOCL in the reference (core.clj:137-145).
 * Non-zero library return codes arrive as RuntimeException with the library's message.
 */
public final class Native {
    static { System.loadLibrary("raymarch_jni"); }
    private Native() {}

    public static native long create(int device);
    public static native long createMulti(ByteBuffer deviceIds, int nDevices);
    public static native int deviceCount();
    public static native void destroy(long handle);
    public static native int setVolume(long handle, ByteBuffer voxels, int rx, int ry, int rz);
    public static native int makeGyroidVolume(long handle, int rx, int ry, int rz, ByteBuffer voxelsOut);
    public static native int renderImage(long handle, ByteBuffer mc, ByteBuffer opts, ByteBuffer pixels, int n);
    public static native int tonemapImage(long handle, ByteBuffer pixels, ByteBuffer opts, ByteBuffer argb, int n);
    public static native int renderFrame(long handle, ByteBuffer optsArray, ByteBuffer mcArray, int iter, int n,
                                         ByteBuffer pixelsOut, ByteBuffer argbOut);
    public static native float lastFrameMillis(long handle);
    public static native int makeScatterTable(long seed, ByteBuffer out);
}
