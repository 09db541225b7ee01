"""Build libmucon_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m mucon_amd.build [--force]

Translation units (SOURCES): mucon_hip.hip (encoder / head, FMA contraction allowed), viterbi.hip
(-ffp-contract=off: every add is a single IEEE operation, as the bit-exact decode requires),
shead.hip (the s-head's bidirectional LSTM) and metrics.hip (evaluation counters).
The .so lands next to this file (git-ignored; gpurun ships it to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmucon_hip.so")
PYHOST_LIB = os.path.join(HERE, "libmucon_pyhost.so")    # host-side helper of the Python binding (csrc/pyhost.c: CPython API, no HIP)
ARCH = "gfx950"
# mucon_hip.hip: -fno-slp-vectorize -- the SLP pass packs the split's adjacent fp32 subtractions into v_pk_add_f32, which costs issue cycles
# beside MFMAs (MI355X_MICROARCH.md: "an anti-lever beside MFMAs"); same box, alternated: 0.7571 -> 0.7545 ms per step, the 16x16x32 weight-gradient
# kernel 236 -> 222 us (profiles/r05_mfma_shape.txt)
# ... -amdgpu-kernarg-preload-count=16 (r6): gfx950 hands the first 16 dwords of LEADING scalar / pointer kernel arguments to a wave in SGPRs; the kernels of the
# latency-bound chain (cs_kernel, ...) take their operand pointers that way and issue their first loads without a scalar-memory round trip (by-value structs are never preloaded)
SOURCES = [("mucon_hip.hip", ["-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=16"]), ("viterbi.hip", ["-ffp-contract=off"]), ("viterbi_beam.hip", ["-ffp-contract=off"]), ("shead.hip", ["-mllvm", "-amdgpu-kernarg-preload-count=16"]),
           ("metrics.hip", ["-ffp-contract=off"])]


last_compiled = []   # the translation units the last build() call compiled (tests/test_build_staleness.py)


def _obj(src):
    return os.path.join(HERE, "build", src.replace(".hip", ".o"))


def _deps_of(src):
    """Files the object of `src` was compiled from (hipcc -MMD wrote them next to it); None: unknown, rebuild."""
    d = _obj(src) + ".d"
    if not os.path.exists(d) or not os.path.exists(_obj(src)):
        return None
    toks = open(d).read().replace("\\\n", " ").split()
    return [t for t in toks[1:] if os.path.exists(t)] or None


def _flags_tag(extra):
    return " ".join(extra + os.environ.get("MUCON_HIPCC_FLAGS", "").split())


def _tu_stale(src, extra):
    deps = _deps_of(src)
    tag = _obj(src) + ".flags"
    if deps is None or not os.path.exists(tag) or open(tag).read() != _flags_tag(extra):
        return True
    t = os.path.getmtime(_obj(src))
    return any(os.path.getmtime(f) > t for f in deps + [os.path.join(CSRC, src)])


def build_pyhost(force: bool = False, verbose: bool = False) -> str:
    """gcc for csrc/pyhost.c (the C loop over the Python lists ops.viterbi_decode_batch is handed; loaded with ctypes.PyDLL)."""
    import sysconfig

    src = os.path.join(CSRC, "pyhost.c")
    hdr = os.path.join(CSRC, "..", "..", "include", "mucon_hip.h")
    if not force and os.path.exists(PYHOST_LIB) and os.path.getmtime(PYHOST_LIB) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return PYHOST_LIB
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-Wall", "-I", sysconfig.get_paths()["include"], src, "-o", PYHOST_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return PYHOST_LIB


def stale_units():
    """The translation units that must be recompiled: those whose object, dependency file (hipcc -MMD: every header the unit
    actually included, so no hand-kept list can fall behind an #include) or flags record is missing, or one of whose
    dependencies is newer than the object."""
    return [src for src, extra in SOURCES if _tu_stale(src, extra)]


def _stale():
    if not os.path.exists(LIB) or stale_units():
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(_obj(src)) > t for src, _ in SOURCES)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the translation units whose sources (or compile flags) changed since their object was built, then link.
    force: every translation unit."""
    build_pyhost(force, verbose)
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    objs = []
    procs = []
    global last_compiled
    last_compiled = []
    for src, extra in SOURCES:
        obj = _obj(src)
        objs.append(obj)
        if not force and not _tu_stale(src, extra):
            continue
        last_compiled.append(src)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-MMD", "-MF", obj + ".d",
               "-c", os.path.join(CSRC, src), "-o", obj] + extra + os.environ.get("MUCON_HIPCC_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd), obj, extra))
    for cmd, p, obj, extra in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        with open(obj + ".flags", "w") as f:
            f.write(_flags_tag(extra))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


STAMP_LIB = os.path.join(HERE, "libmucon_hip_stamp.so")
STAMP_FLAGS = ["-DCLK_STAMP=1", "-DCS_STAMP=1", "-DFS_STAMP=1"]


def build_stamp(force: bool = False, verbose: bool = False) -> str:
    """libmucon_hip_stamp.so: the same library with the in-kernel s_memtime stamps of the four kernel families compiled in (mucon_hip.hip under
    STAMP_FLAGS; the other translation units are the shipped objects).  Never loaded by the product path: bench.py's `kernel_cycles` child
    (tools/kernel_cycles.py, MUCON_LIB_VARIANT=stamp) reads box-independent cycle counts from it in ONE extra step outside every timed region."""
    build(force, verbose)
    src, extra = SOURCES[0]
    obj = os.path.join(HERE, "build", "mucon_hip_stamp.o")
    deps = _deps_of(src) or []
    newest = max([os.path.getmtime(f) for f in deps + [os.path.join(CSRC, src)]] or [0])
    if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest or not os.path.exists(STAMP_LIB) or os.path.getmtime(STAMP_LIB) < os.path.getmtime(LIB):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
               "-c", os.path.join(CSRC, src), "-o", obj] + extra + STAMP_FLAGS
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs = [obj] + [_obj(s_) for s_, _ in SOURCES[1:]]
        subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", STAMP_LIB] + objs)
    return STAMP_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--stamp" in sys.argv:
        print(build_stamp(force="--force" in sys.argv, verbose=True))
