;; native.clj -- the reference's host entry points on libraymarch_hip.so instead of OpenCL.
;;
;; What a maintainer of thi-ng/raymarchcl adds next to src/thi/ng/raymarchcl/core.clj.  The
;; parameter layer is REUSED from core.clj (render-options, compute-eyepos and the struct
;; registry it fills from renderer.cl's typedef, core.clj:24-74,150-152); what thi.ng.simplecl
;; did -- context, buffers, the compiled pipeline of core.clj:76-148 -- becomes calls of
;; thi.ng.raymarchcl.Native (bindings/java/.../Native.java over bindings/jni/raymarch_jni.c).
;; init-renderer, make-render-option-buffer, update-render-option-buffer, test-render and
;; test-anim keep the reference's names and parameter lists (core.clj:99-213).
;;
;; No OpenCL is touched at run time: volumes are read by the library (Native/voxInfo, Native/voxLoad)
;; -- the reference's vio/load-volume wraps the bytes into a simplecl buffer and needs a bound OpenCL
;; state (io.clj:28-33), the very thing this namespace replaces.
;;
;; UNVERIFIED HERE: the build image has no JVM.  tests/test_jni_shim.py checks statically that
;; the parameter lists match the reference's and that only declared natives are called; the JNI
;; shim itself is compiled and run from C on the GPU.
(ns thi.ng.raymarchcl.native
  (:require
   [thi.ng.raymarchcl.core :as core]
   [thi.ng.raymarchcl.generators :as gen]
   [thi.ng.structgen.core :as sg]
   [thi.ng.math.core :as m]
   [piksel.core :as pix])
  (:import
   [java.nio ByteBuffer ByteOrder FloatBuffer IntBuffer]
   [thi.ng.raymarchcl Native]))

(def ^:const record-bytes 544)          ; sizeof(TRenderOpts), renderer.cl:35-78
(def ^:const table-floats (* 0x4000 4)) ; one scatter table, generators.clj:8-16

;; unchanged host functions of the reference
(def render-options core/render-options)   ; [{:keys [width height vres t iter eyepos mat fov dof targetpos gamma groundY voxelSize] :as opts}]
(def compute-eyepos core/compute-eyepos)   ; [theta dist y]

(defn direct
  "n bytes of direct memory in native byte order (what JOCL's buffers are in the reference)."
  ^ByteBuffer [n]
  (.order (ByteBuffer/allocateDirect n) (ByteOrder/nativeOrder)))

(defn- fill-records!
  "Encodes one TRenderOpts per pass into `buf`, pass i at time i*dt."
  [^ByteBuffer buf opts dt]
  (let [layout (sg/lookup :TRenderOpts)
        passes (quot (.capacity buf) record-bytes)]
    (.clear buf)
    (dotimes [i passes]
      (.put buf ^ByteBuffer (.rewind ^ByteBuffer (sg/encode layout (render-options (assoc opts :t (* i dt)))))))
    (.rewind buf)
    buf))

(defn make-render-option-buffer
  "As core.clj:99-106 (pass i at t = 0.333 i), but ONE direct buffer of n records back to back
  -- the library's opts_array -- instead of n OpenCL buffers."
  [n opts]
  (fill-records! (direct (* n record-bytes)) opts 0.333))

(defn update-render-option-buffer
  "As core.clj:108-117: re-encodes in place; note the reference's other time step, 0.3333."
  [buffers opts]
  (fill-records! buffers opts 0.3333))

(defn- scatter-tables
  ^ByteBuffer [passes]
  (let [buf (direct (* passes table-floats 4))
        fb  (.asFloatBuffer buf)]
    (dotimes [_ passes]
      (.put ^FloatBuffer fb (float-array (gen/generate-scatter-offsets 0x4000))))
    buf))

(defn load-volume
  "A .vox file (io.clj:9-33: \"VOXEL\", three big-endian ints, element size, bytes) -> {:res [rx ry rz]
  :voxels direct-buffer}, read by the library -- no OpenCL buffer is involved."
  [path]
  (let [hdr (direct 12)]
    (Native/voxInfo path hdr)
    (let [res (mapv #(.getInt hdr (* 4 %)) (range 3))
          n   (long (reduce * res))
          buf (direct n)]
      (Native/voxLoad path buf n)
      {:res res :voxels buf})))

(defn- open-device
  "One GPU, or -- with :devices n in the renderer args -- the first n GPUs of the node sharing
  every frame (image tiles; tile accumulators gathered on GPU 0 inside the library)."
  [n]
  (if (> n 1)
    (let [ids (direct (* 4 n))]
      (dotimes [i n] (.putInt ids (* 4 i) i))
      (Native/createMulti ids n))
    (Native/create 0)))

(defn init-renderer
  [{:keys [width height vres iter vname] :as args}]
  (let [{:keys [res voxels]} (load-volume (or vname "gyroid-sliced-512-s0.01.vox"))
        [rx ry rz] res
        handle     (open-device (get args :devices 1))
        pixels     (* width height)]
    (when-let [c (get args :contract)] (Native/setContract handle (case c :gfx950-default 2 (:gfx950 :gfx950-strict) 1 :cpu 0)))
    (Native/setVolume handle voxels rx ry rz)
    (let [mc (scatter-tables iter)
          q  (direct (* 4 pixels))]
      ;; allocated once, handed to the library every frame: page-locked for DMA
      (Native/pin handle mc (.capacity mc))
      (Native/pin handle q (.capacity q))
      {:handle       handle
       :num          pixels
       :iter         iter
       :opts-buffers (make-render-option-buffer iter args)
       :mc-buffers   mc
       :q-buf        q})))

(defn execute-pipeline
  "The reference's (ops/execute-pipeline (:pipeline state) ...), core.clj:76-97 + :171, as one
  native call: zeroed accumulator, `iter` RenderImage passes in order, TonemapImage with the
  first record; returns the packed ARGB pixels."
  ^IntBuffer [{:keys [handle opts-buffers mc-buffers iter num q-buf]}]
  (Native/renderFrame handle opts-buffers mc-buffers iter num nil q-buf)
  (.asIntBuffer ^ByteBuffer (.rewind ^ByteBuffer q-buf)))

(defn release [state] (Native/destroy (:handle state)))

(defn render-volume-sequence
  "A new volume per frame (the reference's heat-map animation, meshvoxel.clj:85-89 feeding core.clj:181-213):
  `volumes` = a seq of direct ByteBuffers of the resident volume's resolution `[rx ry rz]`; `f` is called with the
  frame number and the ARGB IntBuffer of each frame (e.g. to write a PNG).  While `f` works on frame k the library
  builds the tables of volume k+1 on its own stream (stageVolume), so no frame waits for them."
  [state [rx ry rz] volumes f]
  (let [handle (:handle state)
        iso    32]
    (when-let [v0 (first volumes)]
      (Native/stageVolume handle v0 rx ry rz iso)
      (Native/commitStagedVolume handle)
      (loop [k 0, more (rest volumes)]
        (let [argb (execute-pipeline state)]
          (when-let [v (first more)] (Native/stageVolume handle v rx ry rz iso))
          (f k argb)
          (when (first more)
            (Native/commitStagedVolume handle)
            (recur (inc k) (rest more))))))))

(defn- save-frame!
  [^IntBuffer argb img path]
  (let [dst (pix/get-pixels img)]
    (.get argb dst)
    (pix/set-pixels img dst)
    (pix/save-png img path)))

(defn test-render
  [& {:keys [width height iter vres mat vname out-path theta dist]
      :or {width 640 height 360 iter 1 vres 256 mat :metal out-path "foo.png"
           theta 135 dist 2.25}
      :as opts}]
  (let [camera {:eyepos (compute-eyepos theta dist 0.35) :targetpos [0 -0.4 0]}
        state  (init-renderer (merge {:width width :height height :vres vres :iter iter :mat mat :vname vname}
                                     camera opts))]
    (try
      (save-frame! (time (execute-pipeline state)) (pix/make-image width height) out-path)
      (finally (release state)))))

(def ^:private orbit
  "The camera path of the reference's animation (core.clj:192-198): 35 frames around the volume."
  {:frames 35 :theta [0 350] :radius [2.25 2.25] :eye-y [0.44 0.45] :target-y [-0.15 -0.15] :fov [115 115]})

(defn- orbit-args
  [frame]
  (let [t  (m/map-interval frame 0 (:frames orbit) 0.0 1.0)
        at (fn [k] (let [[a b] (orbit k)] (m/map-interval t 0 1 a b)))]
    {:fov (at :fov) :targetpos [0 (at :target-y) 0] :eyepos (compute-eyepos (at :theta) (at :radius) (at :eye-y))}))

(defn test-anim
  [width height iter res mat & vname]
  (let [args  {:width width :height height :vres [res res res] :iter iter :mat mat :vname (first vname)}
        img   (pix/make-image width height)
        state (init-renderer args)]
    (try
      (time
       (doseq [frame (range (:frames orbit))]
         (prn "rendering frame #" frame)
         (update-render-option-buffer (:opts-buffers state) (merge args (orbit-args frame)))
         (save-frame! (time (execute-pipeline state)) img (format "export/frame-%04d.png" frame))))
      (finally (release state)))))
