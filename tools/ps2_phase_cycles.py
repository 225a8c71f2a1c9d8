"""Phase cycles of pstack2_kernel (-DPS2_PROF) and of the plain convs' weight gradient (pstack_wgrad_body, -DPW_PROF) for the
speaker classifier C (8 layers, k 5) and the speaker-adversarial net (3 layers, k 3) at the benchmark shape:
python tools/ps2_phase_cycles.py build (here: cross-compiles crank_amd/libcrank_hip_ps2prof.so), then on the GPU box
python tools/ps2_phase_cycles.py"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LIB = os.path.join(REPO, "crank_amd", "libcrank_hip_ps2prof.so")


def build():
    """the instrumented library: both plain-chain sources with their phase timers"""
    subprocess.run(["bash", os.path.join(REPO, "tools", "build_variant.sh"), "ps2prof", "pstack2_kernels.hip", "-DPS2_PROF",
                    "pstack_kernels.hip", "-DPW_PROF"], check=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        sys.exit(0)
    os.environ["CRANK_AMD_LIB"] = LIB
    import numpy as np
    import torch
    from crank_amd import _lib, ops
    from crank_amd.bin.train import get_model
    from crank_amd.utils import load_yaml

    ops.set_precision("bf16")
    L = _lib.lib()
    L.crk_debug_ps2_prof.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    conf = load_yaml(None, batch_size=64, batch_len=500)
    m = get_model(conf, 14, "cuda")
    x = torch.randn(64, 500, 80, device="cuda", requires_grad=True)
    e = torch.randn(64, 500, 128, device="cuda", requires_grad=True)
    names = ["pro: operand -> LDS + barrier", "fragment wait + MFMAs", "next fragments + epilogue", "barrier", "pro: requests", "TOTAL", "pro: table, guards", "pro: biases"]

    def report(tag):
        torch.cuda.synchronize()
        buf = np.zeros(512 * 4 * 8, dtype=np.uint64)
        res = np.zeros(1024 * 2, dtype=np.uint64)
        assert L.crk_debug_ps2_prof(buf.ctypes.data, res.ctypes.data) == 0
        v = buf.reshape(512, 4, 8).astype(np.float64)
        v = v[v[:, 0, 5] > 0]
        r = res.reshape(1024, 2)
        r = r[r[:, 1] > 0]
        span = (r[:, 1].max() - r[:, 0].min()) / 100.0
        life = (r[:, 1] - r[:, 0]).astype(np.float64) / 100.0
        print(f"{tag}: {len(r)} workgroups, kernel span {span:.1f} us, workgroup life mean {life.mean():.1f} us (min {life.min():.1f} max {life.max():.1f}), "
              f"last start at {(r[:, 0].max() - r[:, 0].min()) / 100.0:.1f} us")
        mean = v.mean(axis=(0, 1))
        print("   cycles per wave: " + "  ".join(f"{n} {mean[i]:8.0f}" for i, n in enumerate(names)))

    def report_pw(tag):
        if not hasattr(L, "crk_debug_pw_prof"):
            return
        torch.cuda.synchronize()
        buf = np.zeros(512 * 4 * 8, dtype=np.uint64)
        L.crk_debug_pw_prof.argtypes = [ctypes.c_void_p]
        assert L.crk_debug_pw_prof(buf.ctypes.data) == 0
        v = buf.reshape(512, 4, 8).astype(np.float64)
        v = v[v[:, 0, 5] > 0]
        pn = ["set-up", "barrier + tiles -> LDS + barrier", "next requests", "fragments + MFMAs", "partial sums out", "TOTAL"]
        mean = v.mean(axis=(0, 1))
        print(f"{tag} weight gradient: {len(v)} workgroups recorded; cycles per wave: " + "  ".join(f"{n} {mean[i]:8.0f}" for i, n in enumerate(pn)))

    for _ in range(2):
        y = m["C"](x.transpose(1, 2))
    report("C forward (8 layers k5, 80 -> 64 x6 -> 14)")
    y.sum().backward()
    report("C data gradient")
    report_pw("C")
    for _ in range(2):
        z = m["SPKRADV"]([e[..., :64], e[..., 64:]])
    report("SPKRADV forward (3 layers k3, 128 -> 64 -> 64 -> 14)")
    z.sum().backward()
    report("SPKRADV data gradient")
    report_pw("SPKRADV")
