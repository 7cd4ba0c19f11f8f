"""Builds libpn2_hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python open3d-pointnet2-semantic3d_amd/build.py [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it
travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpn2_hip.so")
SOURCES = ["pn2_abi.hip", "pn2_sampling.hip", "pn2_grouping.hip", "pn2_interpolate.hip",
           "pn2_linear.hip", "pn2_sa_fused.hip", "pn2_sa_fused_bf16.hip", "pn2_label_interp.hip", "pn2_fps_bucket.hip",
           "pn2_bn.hip", "pn2_scene.hip", "pn2_train.hip", "pn2_mlp_wide.hip", "pn2_pool.hip", "pn2_hoist.hip",
           "pn2_coarse_geometry.hip", "pn2_bwd_fused.hip"]
# per-file additions.  pn2_sa_fused.hip: MFMA results in VGPRs where they fit (the fused chains feed every accumulator back into
# the next layer's MFMA as an A / B operand, which must be an arch VGPR: with AGPR accumulators each of those 128 values per
# tile costs a v_accvgpr_read, and non-MFMA instructions cost their full issue time on a SIMD whose matrix pipe is busy)
FILE_FLAGS = {"pn2_sa_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [
        os.path.join(HERE, "..", "include", "pn2_abi.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """extra_flags/out: build an experimental variant next to the default library (tools/ A/B runs)."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = _hipcc()
    target = out or LIB
    tag = "" if out is None else "." + os.path.basename(out).replace(".so", "")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", tag + ".o"))
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out.decode(errors="replace")))
        elif verbose and out:
            print(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("libpn2_hip.so build failed")
    tmp = target + ".tmp%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    subprocess.check_call(cmd)
    os.replace(tmp, target)  # atomic: a concurrent importer sees the old or the new library, never a partial one
    if tag:  # an experimental variant: its objects are of no further use (and would travel to the GPU box with the tree)
        for obj in objs:
            try:
                os.remove(obj)
            except OSError:
                pass
    return target


if __name__ == "__main__":
    # --tuning: compile the kernel-selection knobs of the A/B scripts (tools/) as process-global variables behind
    # pn2_debug_set (csrc/pn2_common.h PN2_TUNABLE).  Experiments only: the shipped library has no mutable global state.
    tuning = "--tuning" in sys.argv
    print(build(force="--force" in sys.argv or tuning, verbose="--verbose" in sys.argv,
                extra_flags=["-DPN2_TUNING_HOOKS"] if tuning else ()))
