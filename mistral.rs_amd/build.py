"""Build the gfx950 shared libraries in-tree with hipcc (no cmake, no torch extension).

    python mistral.rs_amd/build.py [-j N] [--force]

Outputs (git-ignored, shipped to the GPU box by gpurun):
    mistral.rs_amd/lib/libmistralrsquant.so            <- replaces libmistralrsquant.a
    mistral.rs_amd/lib/libmistralrspagedattention.so   <- replaces libmistralrspagedattention.a
    mistral.rs_amd/lib/libmistralrscuda.so             <- replaces libmistralrscuda.a (hot-path subset)
    mistral.rs_amd/lib/libmrs_hip_ext.so               <- MI355X-native extras (fused decode path, host runtime)
(the three reference static libs: mistralrs-quant/build.rs:220-249, mistralrs-paged-attn/build.rs:146-215,
 mistralrs-core/build.rs:59-70)
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "lib")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=off",
            "-I" + CSRC, "-I" + os.path.join(os.path.dirname(HERE), "include")]

MMVQ_TYPES = {"q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8,
              "q2_k": 10, "q3_k": 11, "q4_k": 12, "q5_k": 13, "q6_k": 14}


def _tu(src, obj=None, defines=()):
    return (src, obj or os.path.splitext(src)[0] + ".o", tuple(defines))


def libraries():
    quant = [_tu("mmvq_quantize.hip")]
    quant += [_tu("mmvq_inst.hip", f"mmvq_{tag}.o", (f"-DMRS_TAG={tag}", f"-DMRS_TYPE={tid}",
                                                      "-DMRS_MOE_TAG=" + (tag.replace("_k", "k") if tag.endswith("_k") else tag)))
              for tag, tid in MMVQ_TYPES.items()]
    # Q8_1 as a weight format: MoE launchers only (gguf/ffi.rs:268,424,601,800)
    quant.append(_tu("mmvq_inst.hip", "mmvq_q8_1.o", ("-DMRS_TAG=q8_1", "-DMRS_TYPE=9", "-DMRS_MOE_TAG=q8_1", "-DMRS_MOE_ONLY")))
    libs = {"libmistralrsquant.so": quant}
    pa = [_tu("kv_cache_ops.hip")]
    for tag, t, ct, abi in (("f16", "mrs::f16_t", "mrs::f16_t", True), ("bf16", "mrs::bf16_t", "mrs::bf16_t", True),
                            ("f32", "float", "float", True), ("f32_bf16", "float", "mrs::bf16_t", False)):
        d = [f"-DMRS_PA_TAG={tag}", f"-DMRS_PA_T={t}", f"-DMRS_PA_CT={ct}"] + (["-DMRS_PA_EXPORT_ABI"] if abi else ["-DMRS_PA_DECODE_Q8_1"])
        pa.append(_tu("paged_attention.hip", f"paged_attention_{tag}.o", d))
    for tag, t in (("f16", "mrs::f16_t"), ("bf16", "mrs::bf16_t"), ("f32", "float")):  # fp8 (E4M3) cache read as query dtype `tag`
        pa.append(_tu("paged_attention.hip", f"paged_attention_{tag}_fp8.o",
                      [f"-DMRS_PA_TAG={tag}", f"-DMRS_PA_T={t}", "-DMRS_PA_CT=mrs::fp8_t", "-DMRS_PA_FP8"]))
    libs["libmistralrspagedattention.so"] = pa
    # optional translation units are picked up as soon as the file exists
    optional = {
        "libmistralrsquant.so": ["quant_ops.hip", "mmq.hip", "moe.hip", "gemv.hip", "hqq.hip"],
        "libmistralrscuda.so": ["core_ops.hip", "sampling.hip"],
        "libmrs_hip_ext.so": ["ext_decode.hip", "ext_dec.hip", "ext_dec_mm.hip", "ext_gemm.hip", "ext_gemm_qi.hip", "ext_attn_prefill.hip", "ext_comm.hip", "ext_p2p.hip", "ext_hqq_gemv.hip", "ext_isq.hip", "ext_prefetch.hip", "ext_gemm_lt.hip",
                              "host/runtime.cpp", "host/kv_cache_manager.cpp"],
    }
    # experiment knob (default off): MRS_DECODE_MIN_WAVES=4 caps the decode GEMV kernels at 128 VGPRs (csrc/ext_decode.hip); use with --force
    decode_defs = {"ext_decode.hip": (f"-DMRS_DECODE_MIN_WAVES={os.environ['MRS_DECODE_MIN_WAVES']}",)} if os.environ.get("MRS_DECODE_MIN_WAVES") else {}
    if os.path.exists(os.path.join(CSRC, "ext_dec_gemv.hip")):  # the GEMV phase kernel of the decode engine, one translation unit per activation-column count
        libs.setdefault("libmrs_hip_ext.so", []).extend(_tu("ext_dec_gemv.hip", f"ext_dec_gemv_nc{nc}.o", (f"-DMRS_DEC_NC={nc}",)) for nc in range(1, 9))
    for lib, srcs in optional.items():
        for s in srcs:
            if os.path.exists(os.path.join(CSRC, s)):
                libs.setdefault(lib, []).append(_tu(s, os.path.basename(os.path.splitext(s)[0]) + ".o", decode_defs.get(s, ())))
    return libs


def _headers_mtime():
    """(newest csrc header, newest public header under include/).  Only the translation units that include a public header are rebuilt when one of those changes
    (a declaration added to include/mrs_hip_ext.h used to recompile all ~50 units: 12 minutes)."""
    m = 0.0
    for root, _, files in os.walk(CSRC):
        if root.startswith(OBJ):
            continue
        for f in files:
            if f.endswith((".cuh", ".h", ".hpp")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    pub = 0.0
    inc = os.path.join(os.path.dirname(HERE), "include")
    if os.path.isdir(inc):
        for f in os.listdir(inc):
            pub = max(pub, os.path.getmtime(os.path.join(inc, f)))
    return m, pub


def _includes_public_header(path):
    try:
        txt = open(path, errors="ignore").read()
    except OSError:
        return True
    return any(h in txt for h in ("mrs_hip_ext.h", "mistralrs_quant.h", "mistralrs_paged_attn.h", "mistralrs_core.h"))


def _compile(src, obj, defines, force, hdr_m):
    s, o = os.path.join(CSRC, src), os.path.join(OBJ, obj)
    dep_m = max(hdr_m[0], hdr_m[1] if _includes_public_header(s) else 0.0)
    if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), dep_m):
        return o, False
    cmd = [HIPCC, *CXXFLAGS, *defines, "-c", s, "-o", o]
    if src.endswith(".cpp"):
        cmd = [HIPCC, "-x", "hip", *CXXFLAGS, *defines, "-c", s, "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src} {defines}:\n{r.stderr[-4000:]}")
    return o, True


def build(jobs: int | None = None, force: bool = False, verbose: bool = True) -> dict:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB, exist_ok=True)
    libs = libraries()
    hdr_m = _headers_mtime()
    jobs = jobs or min(16, os.cpu_count() or 4)
    out = {}
    with cf.ThreadPoolExecutor(jobs) as ex:
        futs = {name: [ex.submit(_compile, *tu, force, hdr_m) for tu in tus] for name, tus in libs.items()}
        for name, fl in sorted(futs.items(), key=lambda kv: kv[0] == "libmrs_hip_ext.so"):  # ext links last
            res = [f.result() for f in fl]
            objs = [o for o, _ in res]
            target = os.path.join(LIB, name)
            if force or any(ch for _, ch in res) or not os.path.exists(target):
                cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", target, *objs]
                if name == "libmrs_hip_ext.so":  # the runner calls the drop-in symbols of the other three
                    cmd += ["-L" + LIB, "-lmistralrsquant", "-lmistralrspagedattention", "-lmistralrscuda",
                            "-Wl,-rpath,$ORIGIN", "-ldl"]
                    if any(o.endswith("ext_gemm_lt.o") for o in objs):  # plain bf16 library GEMMs of the bf16-shadow prompt path (hipBLASLt ships with ROCm)
                        cmd += ["-L/opt/rocm/lib", "-lhipblaslt", "-Wl,-rpath,/opt/rocm/lib"]
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError(f"link failed for {name}:\n{r.stderr[-4000:]}")
                if verbose:
                    print(f"[build] linked {name} ({len(objs)} objects)", file=sys.stderr)
            out[name] = target
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=None)
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    for k, v in build(a.j, a.force).items():
        print(k, "->", v)
