"""Phase cycles of stack_bwd_kernel (instrumented build, -DSKB_PROF) for the generator's stacks at the benchmark shape:
python tools/skb_phase_cycles.py [build]"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LIB = os.path.join(REPO, "crank_amd", "libcrank_hip_skbprof.so")


def build():
    csrc = os.path.join(REPO, "crank_amd", "csrc")
    srcs = ["conv_kernels", "stack_kernels", "stack2_kernels", "pstack_kernels", "net", "vq_kernels", "loss_kernels", "mlfb_kernels",
            "dataset_kernels", "mcd_kernels"]
    objs = []
    for s in srcs:
        o = os.path.join(csrc, s + (".prof.o" if s == "stack_kernels" else ".o"))
        if s == "stack_kernels":
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DSKB_PROF",
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
    L.crk_debug_skb_prof.argtypes = [ctypes.c_void_p]
    names = ["prologue", "barrier", "1x1 mfma", "gate bwd", "taps", "dX epi", "commit", "TOTAL", "1x1 issue loads", "1x1 convert"]
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
            y.sum().backward()
        torch.cuda.synchronize()
        buf = np.zeros(256 * 8 * 12, dtype=np.uint64)
        assert L.crk_debug_skb_prof(buf.ctypes.data) == 0
        v = buf.reshape(256, 8, 12).astype(np.float64)
        v = v[v[:, 0, 7] > 0]
        mean = v.mean(axis=(0, 1))
        print(f"{tag} ({layers} blocks, k{k}, aux {aux}): {len(v)} workgroups; cycles per wave: " +
              "  ".join(f"{n} {mean[i]:7.0f}" for i, n in enumerate(names)) +
              f" | per block: barrier {mean[1]/layers:.0f} 1x1 loads {mean[8]/layers:.0f} convert {mean[9]/layers:.0f} mfma {mean[2]/layers:.0f} "
              f"gate {mean[3]/layers:.0f} taps {mean[4]/layers:.0f} dX {mean[5]/layers:.0f} commit {mean[6]/layers:.0f}")
