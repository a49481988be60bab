"""Build libspeechclip_hip.so (gfx950) in-tree with hipcc.  No torch headers, no JIT cache."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.abspath(os.path.join(HERE, "..", "include"))
LIB = os.path.join(CSRC, "libspeechclip_hip.so")
SOURCES = ["sc_error.cpp", "hubert_layer.cpp", "gemm_bf16.hip", "gemm256_bf16.hip", "attention.hip", "attention_bwd.hip", "rowops.hip", "frontend.hip", "posconv.hip", "posconv_bwd.hip", "clspool.hip",
           "loss_optim.hip", "headtail.hip", "rowtail.hip", "backward.hip", "softmax.hip", "cif.hip", "vq.hip"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))] + [os.path.join(INCLUDE, "speechclip_hip.h")]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has a unified register file); otherwise every VALU touch
            # of an accumulator (softmax, epilogues) pays v_accvgpr_read/write moves
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-x", "hip",
                   "-I", INCLUDE, "-I", CSRC,
                   "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
