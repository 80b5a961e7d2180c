"""Build libjolt_hip.so (hand-written gfx950 HIP kernels + C ABI + host mirror) in-tree with hipcc.

    python -m jolt_amd.build            # incremental
    python -m jolt_amd.build --force

The shared object is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libjolt_hip.so")
# host_mirror.hip, shm_exchange.hip, read_raf_address.hip and stage_ops.hip hold HOST code only (the C++ mirror of the reference's drivers above the ABI, the shared-memory round
# exchange, the 128 read-RAF address rounds); they keep the .hip suffix and go through hipcc like the rest because they share field.hip.h (JOLT_HD functions: one
# definition of the field arithmetic for both sides) -- hipcc emits no device code for them
SOURCES = ["capi.hip", "host_mirror.hip", "batch.hip", "views.hip", "msm.hip", "msm_fixed.hip", "hyperkzg.hip", "comm.hip", "onehot.hip", "dory.hip", "pcs.hip", "rw_matrix.hip", "r1cs.hip", "small_r1cs.hip", "read_raf.hip", "key_index.hip", "shm_exchange.hip", "read_raf_address.hip", "stage_ops.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("JOLT_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DJOLT_BUCKET_WAVES=2 for A/B builds


def _headers():
    out = [os.path.join(HERE, "..", "include", "jolt_hip.h")]
    for f in os.listdir(CSRC):
        if f.endswith((".hip.h", ".hpp", ".h")):
            out.append(os.path.join(CSRC, f))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
