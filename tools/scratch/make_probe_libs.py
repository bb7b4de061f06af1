#!/usr/bin/env python
"""Build the timing-only probe libraries behind profiles/r02_wgrad_probes.txt and DESIGN.md section 5 ("timing-only probes").

    python tools/scratch/make_probe_libs.py          ->  build/libudaseg_wprobe.so, build/libudaseg_p{1,2,3,4}.so

The probes are patched COPIES of csrc/conv_wgrad.hip / csrc/conv_igemm.hip (results are wrong on purpose); nothing in the
package changes.  Run a probe library through the whole step with  UDASEG_LIB=$PWD/build/<lib> python bench.py ...
(weight gradient: UDASEG_WGRAD_PROBE=<bits>, see tools/scratch/wprobe_run.sh)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "uda_aerial_semantic_segmentation_research_amd", "csrc")
OUT = os.path.join(ROOT, "build")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics"]


def sub(s, old, new):
    assert old in s, old[:60]
    return s.replace(old, new, 1)


def wgrad_probe():
    s = open(os.path.join(CSRC, "conv_wgrad.hip")).read()
    s = sub(s, "  int ci_full, c_off, J_ld, up;\n};", "  int ci_full, c_off, J_ld, up;\n  int probe;   // PROBE BUILD ONLY\n};")
    s = sub(s, "    if (more) load_tile(kt + 1);\n#pragma unroll\n    for (int k2 = 0; k2 < WBK / 2; ++k2) {",
            "    if (more && !(a.probe & 1)) load_tile(kt + 1);\n    if (!(a.probe & 16))\n#pragma unroll\n    for (int k2 = 0; k2 < WBK / 2; ++k2) {")
    s = sub(s, "    if (more) store_tile(cur ^ 1);\n    __syncthreads();\n    cur ^= 1;\n  }\n\n  // D reg v of lane (lr, lh)",
            "    if (more && !(a.probe & 2)) store_tile(cur ^ 1);\n    if (!(a.probe & 4)) __syncthreads();\n    cur ^= 1;\n  }\n"
            "  if (a.probe & 8) return;\n\n  // D reg v of lane (lr, lh)")
    s = sub(s, "  dim3 grid((unsigned)tiles, (unsigned)splits), block(256);",
            '  { const char* e = getenv("UDASEG_WGRAD_PROBE"); a.probe = e ? atoi(e) : 0; }\n  dim3 grid((unsigned)tiles, (unsigned)splits), block(256);')
    return s


def igemm_probe():
    s = open(os.path.join(CSRC, "conv_igemm.hip")).read()
    s = sub(s, "    const int m = m0 + lrow + 32 * p;", "    const int m = (IGEMM_PROBE == 1 ? 0 : m0) + lrow + 32 * p;")
    s = sub(s, "    load_tile(kt + 2, ra0, rb0);\n    mfma_tile(0);\n    store_tile(1, ra1, rb1);\n    __syncthreads();\n"
               "    load_tile(kt + 3, ra1, rb1);\n    mfma_tile(1);\n    store_tile(0, ra0, rb0);\n    __syncthreads();\n  }",
            "    if (IGEMM_PROBE < 2) load_tile(kt + 2, ra0, rb0);\n    mfma_tile(0);\n    if (IGEMM_PROBE < 3) store_tile(1, ra1, rb1);\n"
            "    if (IGEMM_PROBE < 4) __syncthreads();\n    if (IGEMM_PROBE < 2) load_tile(kt + 3, ra1, rb1);\n    mfma_tile(1);\n"
            "    if (IGEMM_PROBE < 3) store_tile(0, ra0, rb0);\n    if (IGEMM_PROBE < 4) __syncthreads();\n  }")
    return s


def build(src_text, name, replaced_obj, defines=()):
    os.makedirs(OUT, exist_ok=True)
    tmp = os.path.join(CSRC, f"_probe_{name}.hip")          # next to the sources: relative includes
    obj = os.path.join(OUT, f"{name}.o")
    try:
        open(tmp, "w").write(src_text)
        subprocess.check_call([HIPCC, *FLAGS, *defines, "-c", tmp, "-o", obj])
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    objs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and "asan" not in f and f != replaced_obj]
    so = os.path.join(OUT, f"libudaseg_{name}.so")
    subprocess.check_call([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, obj, "-o", so])
    print(so)


def main():
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
    build(wgrad_probe(), "wprobe", "conv_wgrad.o")
    ig = igemm_probe()
    for p in (1, 2, 3, 4):     # 1: every block gathers the same rows; 2: no loads; 3: + no LDS stores; 4: + no barriers
        build(ig, f"p{p}", "conv_igemm.o", (f"-DIGEMM_PROBE={p}",))


if __name__ == "__main__":
    sys.exit(main())
