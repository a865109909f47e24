"""Build libadmm_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

Usage: python -m admm_amd.build [--force]
Each .hip translation unit is compiled to an object under admm_amd/csrc/_obj (only when stale)
and linked against rocBLAS / rocSOLVER / RCCL from /opt/rocm into admm_amd/lib/libadmm_hip.so.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libadmm_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-Wno-unused-result", f"-I{os.path.join(ROCM, 'include')}"]
CXXFLAGS += os.environ.get("ADMM_HIP_EXTRA_CXXFLAGS", "").split()      # dev builds only (e.g. -DADMM_HIP_PROBE)
LDFLAGS = ["-shared", "-fPIC", f"--offload-arch={ARCH}", f"-L{os.path.join(ROCM, 'lib')}",
           "-lrccl", "-ldl", f"-Wl,-rpath,{os.path.join(ROCM, 'lib')}"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


PUBLIC_HEADER = os.path.join(HERE, "..", "include", "admm_hip.h")


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    if os.path.exists(PUBLIC_HEADER):
        hs.append(PUBLIC_HEADER)
    return hs


def _object_key(src, flags):
    """What an object file was compiled from: the translation unit, EVERY header (any of them may be included) and the flags, by
    content.  Modification times are not evidence (rsync -t, archive extraction: an object newer than a changed source), ADVICE r4."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for p in [os.path.join(CSRC, src)] + _headers():
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _object_is_current(obj, key):
    try:
        with open(obj + ".key") as fh:
            return os.path.exists(obj) and fh.read().strip() == key
    except OSError:
        return False


def _compile(src, force):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    key = _object_key(src, CXXFLAGS)
    if not force and _object_is_current(obj, key):
        return obj, False
    cmd = [HIPCC] + CXXFLAGS + ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(obj + ".key", "w") as fh:
        fh.write(key + "\n")
    return obj, True


def source_hash():
    """sha256 over every .hip / .h under csrc/ and include/admm_hip.h (names + contents, sorted)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    if os.path.exists(PUBLIC_HEADER):                      # a package shipped without include/: the hash then covers csrc/ only
        files.append(PUBLIC_HEADER)
    for p in files:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _hash_file(lib):
    return lib + ".srchash"


def _lib_is_current(lib):
    """The linked library was built from exactly these sources: its side file holds the hash of the sources it was linked from
    (round 3 compared modification times, which a checkout / rsync / archive extraction does not preserve: a stale prebuilt
    library could have been used against newer sources and a newer ctypes struct).  Nothing to do then even where the object
    files did not travel (the GPU box receives the .so files and their side files, not csrc/_obj*)."""
    if not os.path.exists(lib) or not os.path.exists(_hash_file(lib)):
        return False
    with open(_hash_file(lib)) as fh:
        return fh.read().strip() == source_hash()


def _write_hash(lib):
    with open(_hash_file(lib), "w") as fh:
        fh.write(source_hash() + "\n")


# ---- host-sanitizer variant (tests/test_sanitizers.py, tests/test_gpu_sanitizers.py): the HOST half of every translation unit
# under AddressSanitizer + UndefinedBehaviorSanitizer (the device code is compiled as usual: -fno-gpu-sanitize), linked against
# the shared sanitizer runtime so that a python process can LD_PRELOAD it.  Test infrastructure: never loaded by default.
OBJ_SAN = os.path.join(CSRC, "_obj_san")
LIB_SAN = os.path.join(LIBDIR, "libadmm_hip_san.so")
SANFLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan"]


def sanitizer_runtime():
    """Path of clang's shared ASan runtime (to LD_PRELOAD), or None."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROCM, "lib", "llvm", "lib", "clang", "*", "lib", "linux", "libclang_rt.asan-x86_64.so")))
    return hits[-1] if hits else None


def _compile_san(src, force):
    obj = os.path.join(OBJ_SAN, src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    flags = [f for f in CXXFLAGS if f != "-O3"] + SANFLAGS
    key = _object_key(src, flags)
    if not force and _object_is_current(obj, key):
        return obj, False
    r = subprocess.run([HIPCC] + flags + ["-c", srcp, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc (sanitizers) failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(obj + ".key", "w") as fh:
        fh.write(key + "\n")
    return obj, True


def build_sanitized(force=False, verbose=True):
    if not force and _lib_is_current(LIB_SAN):
        if verbose:
            print(f"[admm_amd.build] up to date (source hash matches): {LIB_SAN}")
        return LIB_SAN
    os.makedirs(OBJ_SAN, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile_san(s, force), srcs))
    objs = [o for o, _ in results]
    if any(c for _, c in results) or not os.path.exists(LIB_SAN):
        r = subprocess.run([HIPCC] + objs + LDFLAGS + ["-fsanitize=address,undefined", "-shared-libsan", "-o", LIB_SAN], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link (sanitizers) failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[admm_amd.build] linked {LIB_SAN} ({len(objs)} objects, host ASan + UBSan)")
    elif verbose:
        print(f"[admm_amd.build] up to date: {LIB_SAN}")
    _write_hash(LIB_SAN)
    return LIB_SAN


def build(force=False, verbose=True):
    if not force and _lib_is_current(LIB):
        if verbose:
            print(f"[admm_amd.build] up to date (source hash matches): {LIB}")
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC] + objs + LDFLAGS + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[admm_amd.build] linked {LIB} ({len(objs)} objects)")
    elif verbose:
        print(f"[admm_amd.build] up to date: {LIB}")
    _write_hash(LIB)             # every object above was verified against the CONTENT of its sources (or just compiled from them)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
