package thi.ng.raymarchcl;

import java.nio.ByteBuffer;

/**
 * Static native methods of the JNI shim (bindings/jni/raymarch_jni.c) over libraymarch_hip.so.
 * Unverified in the build image (no JDK).  EVERY buffer is a DIRECT java.nio.ByteBuffer in native
 * byte order (sizes are checked in bytes; use asFloatBuffer()/asIntBuffer() views on the Java side)
 * -- what thi.ng.simplecl passes to JOCL in the reference (core.clj:137-145).  A null, non-direct
 * or too small required buffer raises IllegalArgumentException before any native call; outputs
 * documented as optional may be null.  Non-zero library return codes arrive as RuntimeException
 * with the library's message.
 */
public final class Native {
    static { System.loadLibrary("raymarch_jni"); }
    private Native() {}

    public static native long create(int device);
    public static native long createMulti(ByteBuffer deviceIds, int nDevices);
    public static native int deviceCount();
    public static native void destroy(long handle);
    public static native int setVolume(long handle, ByteBuffer voxels, int rx, int ry, int rz);
    /** The volume of the NEXT animation frame: copied, its derived tables built on the library's own stream (no wait);
     *  commitStagedVolume makes it the resident one for every later call. */
    public static native int stageVolume(long handle, ByteBuffer voxels, int rx, int ry, int rz, int isoVal);
    public static native int commitStagedVolume(long handle);
    public static native int makeGyroidVolume(long handle, int rx, int ry, int rz, ByteBuffer voxelsOut);
    public static native int renderImage(long handle, ByteBuffer mc, ByteBuffer opts, ByteBuffer pixels, int n);
    public static native int tonemapImage(long handle, ByteBuffer pixels, ByteBuffer opts, ByteBuffer argb, int n);
    public static native int renderFrame(long handle, ByteBuffer optsArray, ByteBuffer mcArray, int iter, int n,
                                         ByteBuffer pixelsOut, ByteBuffer argbOut);
    public static native float lastFrameMillis(long handle);
    public static native int makeScatterTable(long seed, ByteBuffer out);
    /** Whose results the kernels reproduce bit for bit (rm_set_contract): 2 = the reference kernel as ROCm's OpenCL compiler
     *  builds it for this GPU with no options (the library default), 1 = the same built with -ffp-contract=off and
     *  correctly rounded divide/sqrt, 0 = an OpenCL CPU device on x86-64. */
    public static native int setContract(long handle, int contract);
    /** Header of a .vox file (io.clj:9-33): out3 receives rx, ry, rz (3 ints). */
    public static native int voxInfo(String path, ByteBuffer out3);
    /** The volume bytes of a .vox file into voxelsOut (capacity = rx*ry*rz from voxInfo). */
    public static native int voxLoad(String path, ByteBuffer voxelsOut, long capacity);
    /** Page-lock a long-lived direct buffer handed to renderFrame every frame (DMA at PCIe speed). */
    public static native int pin(long handle, ByteBuffer buf, long bytes);
    public static native int unpin(long handle, ByteBuffer buf);
}
