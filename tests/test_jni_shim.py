"""The JNI shim (bindings/jni/raymarch_jni.c) compiled against a hand-written declaration guard
for the JNI names it uses (bindings/jni/test/jni.h -- the image has no JDK) and driven from C
through a stand-in JNIEnv function table (bindings/jni/test/harness.c).

not gpu: it compiles warning-free and exports exactly the methods Native.java declares.
gpu:     every Java_* entry point runs on the device; frames equal the direct C-ABI result
         and the oracle; library errors arrive as Java exceptions."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI = os.path.join(ROOT, "bindings", "jni")
JAVA = os.path.join(ROOT, "bindings", "java", "thi", "ng", "raymarchcl", "Native.java")
CLJ = os.path.join(ROOT, "bindings", "clojure", "thi", "ng", "raymarchcl", "native.clj")
CFLAGS = ["-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-fPIC", "-I" + os.path.join(JNI, "test"),
          "-I" + os.path.join(ROOT, "include")]


def _java_natives():
    src = open(JAVA).read()
    return sorted(re.findall(r"public static native \w+ (\w+)\(", src))


def test_shim_compiles_and_exports_what_native_java_declares(tmp_path):
    obj = tmp_path / "raymarch_jni.o"
    subprocess.check_call(["gcc", *CFLAGS, "-c", os.path.join(JNI, "raymarch_jni.c"), "-o", str(obj)])
    syms = subprocess.check_output(["nm", "--defined-only", str(obj)], text=True)
    exported = sorted(m.group(1) for m in re.finditer(r" T Java_thi_ng_raymarchcl_Native_(\w+)", syms))
    assert exported == _java_natives() and len(exported) >= 16
    # the harness compiles against the same declarations (used by the gpu test)
    subprocess.check_call(["gcc", *CFLAGS, "-c", os.path.join(JNI, "test", "harness.c"), "-o", str(tmp_path / "h.o")])
    # every C-ABI function the shim calls is declared in the product header (-Werror above
    # rejects implicit declarations) and exported by the library
    from raymarchcl_amd import _native

    und = subprocess.check_output(["nm", "--undefined-only", str(obj)], text=True)
    for name in re.findall(r" U (rm_\w+)", und):
        assert name in _native.EXPORTS, name


def test_clojure_namespace_keeps_the_reference_entry_points():
    """Static check (no JVM here): the namespace defines the functions of core.clj:28-213 a user
    calls, with the reference's parameter lists, and only calls natives Native.java declares."""
    src = open(CLJ).read()
    # the parameter layer is the reference's own (reused, not restated)
    assert re.search(r"\(def render-options core/render-options\)", src)
    assert re.search(r"\(def compute-eyepos core/compute-eyepos\)", src)
    for sig in (r"\(defn make-render-option-buffer\s+(\"[^\"]*\"\s+)?\[n opts\]",
                r"\(defn update-render-option-buffer\s+(\"[^\"]*\"\s+)?\[buffers opts\]",
                r"\(defn init-renderer\s+\[\{:keys \[width height vres iter vname\] :as args\}\]",
                r"\(defn test-render\s+\[& \{:keys \[width height iter vres mat vname out-path theta dist\]\s+"
                r":or \{width 640 height 360 iter 1 vres 256 mat :metal out-path \"foo.png\"\s+theta 135 dist 2.25\}",
                r"\(defn test-anim\s+\[width height iter res mat & vname\]"):
        assert re.search(sig, src), sig
    called = set(re.findall(r"Native/(\w+)", src))
    assert called and called <= set(_java_natives())
    # nothing of the OpenCL glue is reachable from this namespace: no simplecl require, no call into the
    # reference's io namespace (its load-volume builds a simplecl buffer and needs a bound OpenCL state,
    # io.clj:28-33); volumes come through the library's own .vox reader
    code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith(";"))
    assert "simplecl" not in code and "raymarchcl.io" not in code and "vio/" not in code
    assert {"voxInfo", "voxLoad", "setVolume", "renderFrame"} <= called
    assert src.count("(") == src.count(")") and src.count("[") == src.count("]") and src.count("{") == src.count("}")


@pytest.mark.gpu
def test_every_entry_point_through_a_stand_in_jnienv(tmp_path, native, oracle_mod):
    """A JNI caller that never mentions a contract gets the library default, RM_CONTRACT_GFX950_DEFAULT:
    its frames equal the reference kernel built for this chip (live build or its recording)."""
    import gfx950_pin
    import scenes

    native.build()
    exe = tmp_path / "harness"
    libdir = os.path.dirname(native.LIB_PATH)
    # link against torch's HIP runtime if the library was built to find it there; plain rpath otherwise
    subprocess.check_call(["gcc", *CFLAGS, os.path.join(JNI, "raymarch_jni.c"), os.path.join(JNI, "test", "harness.c"),
                           "-L" + libdir, "-l:" + os.path.basename(native.LIB_PATH), "-Wl,-rpath," + libdir,
                           "-Wl,--allow-shlib-undefined", "-o", str(exe)])
    sc = scenes.build("metal_3spp")
    n, it = sc["n"], sc["iter"]
    with open(tmp_path / "scene.bin", "wb") as f:
        f.write(np.array([*sc["vres"], it, n], np.int32).tobytes())
        f.write(sc["vox"].tobytes())
        f.write(sc["opts"])
        f.write(np.ascontiguousarray(sc["mc"], np.float32).tobytes())
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe), str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    raw = np.fromfile(tmp_path / "out.bin", dtype=np.uint32)
    px, argb, px1, argb1, checks = np.split(raw, [4 * n, 5 * n, 9 * n, 10 * n])
    want, want_argb = gfx950_pin.Checker(oracle_mod, "default").frame("metal_3spp", sc["vox"], sc["opts"], sc["mc"], n)
    assert np.array_equal(px, want.view(np.uint32)) and np.array_equal(argb, want_argb)
    assert np.array_equal(px1, want.view(np.uint32)) and np.array_equal(argb1, want_argb)
    assert checks.tolist() == [1] * 10, (checks.tolist(), r.stderr[-1500:])
