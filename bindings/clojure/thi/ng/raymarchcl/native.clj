;; native.clj -- what a maintainer of thi-ng/raymarchcl adds to run the render
;; path on libraymarch_hip.so instead of OpenCL.  UNVERIFIED HERE (no JVM in the
;; build image).  It keeps the signatures of the reference's host functions
;; (core.clj): `init-renderer` still returns a state map, `execute` still
;; returns the IntBuffer of packed ARGB pixels that test-render copies into a
;; BufferedImage (core.clj:171-179).  structgen's `sg/encode` keeps producing
;; the 544-byte TRenderOpts records (core.clj:99-106) -- the native library
;; consumes exactly that layout.
(ns thi.ng.raymarchcl.native
  (:import [java.nio ByteBuffer ByteOrder]))

(gen-class
 :name thi.ng.raymarchcl.Native
 :methods [^:static [create [int] long]
           ^:static [destroy [long] void]
           ^:static [setVolume [long java.nio.ByteBuffer int int int] int]
           ^:static [renderImage [long java.nio.ByteBuffer java.nio.ByteBuffer java.nio.ByteBuffer int] int]
           ^:static [tonemapImage [long java.nio.ByteBuffer java.nio.ByteBuffer java.nio.ByteBuffer int] int]
           ^:static [renderFrame [long java.nio.ByteBuffer java.nio.ByteBuffer int int
                                  java.nio.ByteBuffer java.nio.ByteBuffer] int]])
;; (in practice: a 10-line Java class with `static native` methods and
;;  System.loadLibrary("raymarch_jni"); gen-class cannot declare natives)

(defn direct [n] (.order (ByteBuffer/allocateDirect n) (ByteOrder/nativeOrder)))

(defn init-renderer
  "As core/init-renderer (core.clj:119-148), minus the OpenCL state."
  [{:keys [width height iter opts-bytes mc-floats voxels vres]}]
  (let [h (thi.ng.raymarchcl.Native/create 0)]
    (thi.ng.raymarchcl.Native/setVolume h voxels (vres 0) (vres 1) (vres 2))
    {:handle h :num (* width height) :iter iter
     :opts opts-bytes   ; iter x 544 B, from (sg/encode t-opts (render-options ...))
     :mc mc-floats      ; iter x 0x4000 x 4 floats, from gen/generate-scatter-offsets
     :q-buf (direct (* 4 width height))}))

(defn execute
  "As (ops/execute-pipeline (:pipeline state) ...) (core.clj:171): returns the ARGB IntBuffer."
  [{:keys [handle num iter opts mc q-buf]}]
  (thi.ng.raymarchcl.Native/renderFrame handle opts mc iter num nil q-buf)
  (.asIntBuffer q-buf))
