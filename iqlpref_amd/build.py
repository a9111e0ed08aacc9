"""Build libiqlhip.so (gfx950) in-tree with hipcc.  Run: python -m iqlpref_amd.build

The library is stamped with a hash of the sources it was built from
(``iqlhip_build_tag()``); ``_lib.load()`` recomputes the hash of the sources lying
beside the library and refuses a mismatching one, so a stale ``.so`` can neither be
measured nor tested by accident."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(HERE, "..", "include", "iqlhip.h")
SOURCES = ["api.hip", "iql_step.hip", "iql_deep.hip", "buffer.hip", "mlp_f32.hip", "cvar.hip", "pt.hip", "prep.hip"]
LIB = os.path.join(HERE, "libiqlhip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
         # kernarg preload: the first 16 kernarg dwords (the descriptor pointers) arrive in SGPRs
         # with the wave instead of through a dependent s_load at the top of every kernel
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def source_tag(extra=()):
    """sha256 (16 hex digits) over every file of csrc/, the public header and the flags."""
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                   if f.endswith((".hip", ".h", ".hpp")))
    for path in files + [HEADER]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + list(extra)).encode())
    return h.hexdigest()[:16]


def built_tag(lib=LIB):
    """Tag embedded in an existing library (read from the file, nothing is loaded), or None."""
    if not os.path.exists(lib):
        return None
    with open(lib, "rb") as f:
        blob = f.read()
    marker = b"IQLHIP_BUILD_TAG="
    i = blob.find(marker)
    if i < 0:
        return None
    return blob[i + len(marker):i + len(marker) + 16].decode("ascii", "replace")


def needs_build():
    return built_tag() != source_tag()


def build(force=False, verbose=True, stamps=False, variant=None, defines=()):
    """stamps=True builds the diagnostic libiqlhip_stamps.so (tools/stamps.py), never the product.
    variant="x" + defines=["IQL_FOO=1"]: an experimental build libiqlhip_x.so for A/B runs
    (tools/ab.sh, loaded through IQLHIP_LIB), never the product either."""
    if not stamps and not variant and not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = LIB.replace(".so", "_stamps.so") if stamps else LIB
    extra = ["-DIQL_STAMPS"] if stamps else []
    if variant:
        out = LIB.replace(".so", f"_{variant}{'_stamps' if stamps else ''}.so")
        extra += [f"-D{d}" for d in defines]
    tag = source_tag(extra)
    cmd = [hipcc] + FLAGS + extra + [f'-DIQLHIP_BUILD_TAG="{tag}"', "-o", out] + \
          [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    # python -m iqlpref_amd.build [--force] [--stamps] [--variant NAME -DX=1 ...]
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    build(force="--force" in sys.argv, stamps="--stamps" in sys.argv, variant=_variant,
          defines=[a[2:] for a in sys.argv if a.startswith("-D")])
