"""Builds libvmmt.so (hand-written HIP kernels, gfx950 only) in-tree with hipcc.
`python -m variational_mmt_amd.build` or `__graft_entry__.build()`; hipcc cross-compiles without a GPU."""
import concurrent.futures
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libvmmt.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage"]
# every kernel's register / scratch figures as the compiler reports them (the remarks above), one JSON next to the library: several
# kernels of this library only work TOGETHER while their registers add up to a SIMD's 512 -- tests/test_kernel_resources.py holds the sums
RESOURCES = os.path.join(HERE, "kernel_resources.json")
# gemm.hip is compiled three times, one operand layout per object (see the note at the end of that file)
GEMM_PARTS = 3
SOURCES = ["lstm.hip", "lstm_seq.hip", "qnet.hip", "attention.hip", "generator.hip", "generator_fused.hip", "elementwise.hip", "optim.hip", "runtime.hip", "conditional.hip", "table.hip", "beam.hip"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _resources(remarks):
    """{mangled kernel name: {vgprs, agprs, sgprs, scratch, vgpr_spill, occupancy}} from -Rpass-analysis=kernel-resource-usage"""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill",
            "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds_static"}
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*?): (-?\d+)", line)
        if m and cur is not None and m.group(1) in keys:
            cur[keys[m.group(1)]] = int(m.group(2))
    return out


def build(force=False, verbose=True):
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "glds_gemm.hpp"), os.path.join(ROOT, "include", "vmmt.h")]
    jobs = []
    for part in range(GEMM_PARTS):           # the long poles first
        src = os.path.join(CSRC, "gemm.hip")
        obj = os.path.join(objdir, "gemm_p%d.o" % part)
        if force or _stale(obj, [src] + common) or not os.path.exists(obj + ".resources.json"):
            jobs.append((src, obj, ["-DVMMT_GEMM_PART=%d" % part]))
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + common) or not os.path.exists(obj + ".resources.json"):
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
        with open(obj + ".resources.json", "w") as f:
            json.dump(_resources(r.stderr), f)
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
    if force or jobs or not os.path.exists(RESOURCES):
        allk = {}
        for o in objs:
            with open(o + ".resources.json") as f:
                allk.update(json.load(f))
        with open(RESOURCES, "w") as f:
            json.dump(allk, f, indent=0, sort_keys=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
