"""Phase cycles of stack2_bwd_kernel (instrumented build, -DS2B_PROF) for the generator's stacks at the benchmark shape:
python tools/s2b_phase_cycles.py [build]"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LIB = os.path.join(REPO, "crank_amd", "libcrank_hip_s2bprof.so")


def build():
    csrc = os.path.join(REPO, "crank_amd", "csrc")
    srcs = ["conv_kernels", "stack_kernels", "stack2_kernels", "stack2b_kernels", "pstack_kernels", "net", "vq_kernels", "loss_kernels",
            "mlfb_kernels", "dataset_kernels", "mcd_kernels"]
    objs = []
    for s in srcs:
        o = os.path.join(csrc, s + (".prof.o" if s == "stack2b_kernels" else ".o"))
        if s == "stack2b_kernels":
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DS2B_PROF",
                            "-c", os.path.join(csrc, s + ".hip"), "-o", o], check=True)
        objs.append(o)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB], check=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        sys.exit(0)
    os.environ["CRANK_AMD_LIB"] = LIB
    import numpy as np
    import torch
    from crank_amd import _lib, ops
    from crank_amd.net.module.flat import FlatModel
    from crank_amd.net.module.pwg import KIND_GENERATOR, HipStack

    ops.set_precision("bf16")
    L = _lib.lib()
    L.crk_debug_s2b_prof.argtypes = [ctypes.c_void_p]
    names = ["prologue", "P1 1x1+gate", "wait A", "taps(rest)", "dX epi", "wait B", "first conv", "TOTAL", "step0", "steps1-8", "steps9-16", "-"]
    for tag, cin, cout, k, layers, stacks, aux in (("enc0", 80, 64, 5, 8, 4, 0), ("dec0", 128, 80, 5, 8, 4, 34), ("enc1", 64, 64, 3, 6, 3, 0)):
        class M(FlatModel):
            def __init__(self):
                super().__init__()
                self.stack = HipStack(KIND_GENERATOR, cin, cout, k, layers, stacks=stacks, aux_channels=aux, bias=True)
                self._alloc(self.stack.entries("", 0), self.stack.n_params, "cuda")
                self.stack.bind(self, 0)
                self.stack.init_parameters()
        m = M()
        x = torch.randn(64, 500, cin, device="cuda", requires_grad=True)
        a = torch.randn(64, 500, aux, device="cuda", requires_grad=True) if aux else None
        for _ in range(2):
            y = m.stack(x, c=a)
            y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        buf = np.zeros(256 * 4 * 12, dtype=np.uint64)
        assert L.crk_debug_s2b_prof(buf.ctypes.data) == 0
        v = buf.reshape(256, 4, 12).astype(np.float64)
        print(f"{tag} ({layers} blocks, k{k}, aux {aux}): cycles per wave (mean over 256 workgroups)")
        for w in range(4):
            print(f"  wave {w} (mt {w & 1}, fh {w >> 1}): " + "  ".join(f"{n} {v[:, w, i].mean():7.0f}" for i, n in enumerate(names)))
        mean = v.mean(axis=(0, 1))
        print("  all : " + "  ".join(f"{n} {mean[i]:7.0f}" for i, n in enumerate(names)),
              "| per block: " + " ".join(f"{names[i]} {mean[i] / layers:.0f}" for i in (1, 2, 8, 9, 10, 3, 4, 5)))
