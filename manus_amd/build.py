"""Build libmanus_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m manus_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["raster_fwd.hip", "raster_bwd.hip", "lbs_sh.hip", "knn.hip", "image_loss.hip", "optim.hip", "contact.hip", "mesh.hip", "exchange.hip"]
HEADERS = ["mgr_common.h", "instance_math.h", "il_list.h", os.path.join("..", "..", "include", "manus_hip.h")]
# MGR_VARIANT=name builds an instrumented copy (libmanus_hip_<name>.so, objects under build_<name>/) next to the product
# library; MANUS_HIP_VARIANT=name makes _lib load it (tools/instr only)
VARIANT = os.environ.get("MGR_VARIANT", "")
LIB = os.path.join(HERE, "libmanus_hip%s.so" % ("_" + VARIANT if VARIANT else ""))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-disable-unclustered-high-rp-reschedule: the scheduler's extra stage for regions of high register pressure is
# skipped -- measured over a sweep of the AMDGPU scheduling options on the bench step (round 4): k_blend_bwd 0.418 -> 0.411 ms,
# k_inst_fwd 0.173 -> 0.169, the others unchanged, 716 -> 720 iters/s; max-ilp / max-memory-clause strategies, the AMDGPU
# pressure trackers and no post-RA scheduling all cost k_blend_bwd 0.01 - 0.05 ms.  Scheduling only: results are bit for bit the same.
FLAGS = (os.environ.get("MGR_EXTRA_FLAGS", "").split()) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
         "-Wno-unused-function", "-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule=1"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    objdir = os.path.join(HERE, "build" + ("_" + VARIANT if VARIANT else ""))
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    stamp = os.path.join(objdir, "flags.txt")    # objects built with other flags (MGR_EXTRA_FLAGS) are stale too
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(FLAGS):
        force = True
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    with open(stamp, "w") as f:
        f.write(" ".join(FLAGS))
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
