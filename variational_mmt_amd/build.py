"""Builds libvmmt.so (hand-written HIP kernels, gfx950 only) in-tree with hipcc.
`python -m variational_mmt_amd.build` or `__graft_entry__.build()`; hipcc cross-compiles without a GPU."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libvmmt.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result"]
# gemm.hip is compiled three times, one operand layout per object (see the note at the end of that file)
GEMM_PARTS = 3
SOURCES = ["lstm.hip", "lstm_seq.hip", "qnet.hip", "attention.hip", "generator.hip", "generator_fused.hip", "elementwise.hip", "optim.hip", "runtime.hip", "conditional.hip", "table.hip", "beam.hip"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "glds_gemm.hpp"), os.path.join(ROOT, "include", "vmmt.h")]
    jobs = []
    for part in range(GEMM_PARTS):           # the long poles first
        src = os.path.join(CSRC, "gemm.hip")
        obj = os.path.join(objdir, "gemm_p%d.o" % part)
        if force or _stale(obj, [src] + common):
            jobs.append((src, obj, ["-DVMMT_GEMM_PART=%d" % part]))
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + common):
            jobs.append((src, obj, []))
    stale_whole = os.path.join(objdir, "gemm.o")          # object of the former single-unit build
    if os.path.exists(stale_whole):
        os.remove(stale_whole)

    def cc(job):
        src, obj, extra = job
        cmd = [HIPCC] + FLAGS + extra + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(7, max(1, len(jobs)))) as ex:
        for s in ex.map(cc, jobs):
            if verbose:
                print("[vmmt build] compiled", os.path.basename(s), flush=True)
    objs = [os.path.join(objdir, "gemm_p%d.o" % part) for part in range(GEMM_PARTS)] + \
           [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("[vmmt build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
