"""Build libspeechclip_hip.so (gfx950) in-tree with hipcc.  No torch headers, no JIT cache."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.abspath(os.path.join(HERE, "..", "include"))
LIB = os.path.join(CSRC, "libspeechclip_hip.so")
DIAG_LIB = os.path.join(CSRC, "libspeechclip_hip_diag.so")    # same sources + -DSC_DIAG_BUILD: diagnostic kernels, LayerNorm-folded GEMMs
DIAG_SOURCES = ["sc_error.cpp", "gemm256_bf16.hip", "attention.hip"]    # the files whose contents depend on SC_DIAG_BUILD
# CHECKER build (tests/test_gpu_model.py's paired activation test only): -DSC_GELU_EXACT puts the 3e-7-accurate A&S erf-GELU at every
# bf16-output site instead of the five-term fit (csrc/sc_common.h), nothing else differs; the product path never loads it
GELU_EXACT_LIB = os.path.join(CSRC, "libspeechclip_hip_gelu_exact.so")
GELU_EXACT_SOURCES = ["backward.hip", "frontend.hip", "gemm_bf16.hip", "gemm256_bf16.hip", "posconv.hip", "rowops.hip"]   # every user of gelu_bf / gelu_bf2
SOURCES = ["sc_error.cpp", "hubert_layer.cpp", "gemm_bf16.hip", "gemm256_bf16.hip", "attention.hip", "attention_bwd.hip", "rowops.hip", "frontend.hip", "posconv.hip", "posconv_bwd.hip", "clspool.hip",
           "loss_optim.hip", "headtail.hip", "rowtail.hip", "backward.hip", "softmax.hip", "cif.hip", "vq.hip", "prompt.hip", "attn_short.hip"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, diag: bool = True) -> str:
    """-> path of the product library.  ``diag``: also build libspeechclip_hip_diag.so (tools/, bench.py's in-kernel clock probe and
    the opt-in SC_FUSED_LN path load it through _lib.diag_lib(); the product path never does) and the exact-GELU checker build."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))] + [os.path.join(INCLUDE, "speechclip_hip.h")]
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has a unified register file); otherwise every VALU touch
    # of an accumulator (softmax, epilogues) pays v_accvgpr_read/write moves
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-x", "hip", "-I", INCLUDE, "-I", CSRC]
    objs, diag_objs, exact_objs, procs = [], [], [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        stem = os.path.splitext(src)[0]
        obj = os.path.join(CSRC, stem + ".o")
        objs.append(obj)
        variants = [(obj, [])]
        if diag and src in DIAG_SOURCES:
            dobj = os.path.join(CSRC, stem + "_diag.o")
            diag_objs.append(dobj)
            variants.append((dobj, ["-DSC_DIAG_BUILD=1"]))
        elif diag:
            diag_objs.append(obj)
        if diag and src in GELU_EXACT_SOURCES:
            eobj = os.path.join(CSRC, stem + "_gelu_exact.o")
            exact_objs.append(eobj)
            variants.append((eobj, ["-DSC_GELU_EXACT=1"]))
        elif diag:
            exact_objs.append(obj)
        for o, flags in variants:
            if force or _stale(o, [sp] + headers):
                cmd = base + flags + ["-c", sp, "-o", o]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    for lib, parts in ((LIB, objs),) + (((DIAG_LIB, diag_objs), (GELU_EXACT_LIB, exact_objs)) if diag else ()):
        if force or _stale(lib, parts):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + parts
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
