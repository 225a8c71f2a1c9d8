"""Shader cycles of the VQ search kernel's phases (library built with -DVQ_PROF: crank_amd/libcrank_hip_vqprof.so), mean over the
workgroups of a 32 000-frame call: python tools/vq_phase_cycles.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CRANK_AMD_LIB"] = os.environ.get("VQ_PROF_LIB") or os.path.join(ROOT, "crank_amd", "libcrank_hip_vqprof.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from crank_amd import _lib, ops  # noqa: E402


def report(tag):
    L = _lib.lib()
    L.crk_debug_vq_prof.argtypes = [ctypes.c_void_p]
    b = np.zeros(1024, dtype=np.uint64)
    L.crk_debug_vq_prof(b.ctypes.data)
    b = b.reshape(256, 4).astype(float)
    print("%s cycles since kernel start: planes staged %.0f, candidates merged %.0f, decided %.0f, end %.0f"
          % ((tag,) + tuple(b[:250, [0, 3, 1, 2]].mean(0))))


def main():
    x = torch.randn(64, 500, 64, device="cuda")
    w = torch.randn(512, 64, device="cuda") * 0.7
    for _ in range(3):
        ops.vq_apply(x, w)
    torch.cuda.synchronize()
    report("plain call (e, qx)            ")
    img = torch.empty(ops.vq_image_bytes(512, 64), device="cuda", dtype=torch.uint8)
    ops.vq_image_build([w], [img])
    for _ in range(3):
        ops.vq_apply(x, w, image=img)
    torch.cuda.synchronize()
    report("plain call, prepared image    ")
    a = torch.randn(64, 500, 64, device="cuda")
    m = torch.ones(64, 500, dtype=torch.bool, device="cuda")
    for _ in range(3):
        ops.vq_commit_apply(x, w, m, add=a)
    torch.cuda.synchronize()
    report("fused call (+ add, xsum, commit)")
    for _ in range(3):
        ops.vq_commit_apply(x, w, m, add=a, image=img)
    torch.cuda.synchronize()
    report("fused call, prepared image      ")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for im, k in ((None, 0), (img, 2)):
        ev[k].record()
        for _ in range(50):
            ops.vq_commit_apply(x, w, m, add=a, image=im)
        ev[k + 1].record()
    torch.cuda.synchronize()
    print("fused call, 50 back to back: %.1f us per call derived per call, %.1f us with the prepared image"
          % (ev[0].elapsed_time(ev[1]) * 20, ev[2].elapsed_time(ev[3]) * 20))


if __name__ == "__main__":
    main()
