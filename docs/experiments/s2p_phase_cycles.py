"""Phase cycles of stack2p_fwd_kernel (instrumented build, -DS2P_PROF) at the benchmark shape: per-wave shader cycles in the
O parts, the T chains, the gates and the waits at the stage barriers, averaged over 256 workgroups, frame half 0 / 1 apart.
    python tools/s2p_phase_cycles.py build   (container: hipcc)      python tools/s2p_phase_cycles.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LIB = os.path.join(REPO, "crank_amd", "libcrank_hip_s2pprof.so")


def build(extra=()):
    csrc = os.path.join(REPO, "crank_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = [w[:-4] for w in mk.split("SRCS :=")[1].split("\n")[0].split()]
    objs = []
    for s in srcs:
        o = os.path.join(csrc, s + (".s2pprof.o" if s == "stack2p_kernels" else ".o"))
        if s == "stack2p_kernels":
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DS2P_PROF"] + list(extra) +
                           ["-c", os.path.join(csrc, s + ".hip"), "-o", o], check=True)
        objs.append(o)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB], check=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(sys.argv[2:])
        sys.exit(0)
    os.environ["CRANK_AMD_LIB"] = LIB
    import numpy as np
    import torch
    from crank_amd import _lib, ops
    from crank_amd.net.module.flat import FlatModel
    from crank_amd.net.module.pwg import KIND_GENERATOR, HipStack

    ops.set_precision("bf16")
    L = _lib.lib()
    L.crk_debug_s2p_prof.argtypes = [ctypes.c_void_p]
    names = ["prologue", "O parts", "T chains", "gates", "stage barriers", "head", "TOTAL"]
    for tag, cin, cout, k, layers, stacks, aux in (("enc0", 80, 64, 5, 8, 4, 0), ("dec0", 128, 80, 5, 8, 4, 34), ("enc1", 64, 64, 3, 6, 3, 0)):
        class M(FlatModel):
            def __init__(self):
                super().__init__()
                self.stack = HipStack(KIND_GENERATOR, cin, cout, k, layers, stacks=stacks, aux_channels=aux, bias=True)
                self._alloc(self.stack.entries("", 0), self.stack.n_params, "cuda")
                self.stack.bind(self, 0)
                self.stack.init_parameters()
        m = M()
        x = torch.randn(64, 500, cin, device="cuda")
        a = torch.randn(64, 500, aux, device="cuda") if aux else None
        for grad in (False, True):
            with torch.set_grad_enabled(grad):
                xi = x.clone().requires_grad_(grad)
                for _ in range(3):
                    m.stack(xi, c=a)
            torch.cuda.synchronize()
            buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
            assert L.crk_debug_s2p_prof(buf.ctypes.data) == 0
            buf = buf.reshape(256, 8, 8).astype(np.float64)
            nb = layers
            for half, sl in (("frame half 0 (O, T, gate)", slice(0, 4)), ("frame half 1 (T, gate, O)", slice(4, 8))):
                v = buf[:, sl, :7].mean((0, 1))
                print(f"{tag} {'saving planes' if grad else 'no-grad'} {half}: " + "  ".join(f"{n} {c:8.0f}" for n, c in zip(names, v)) +
                      f" | per block: O {v[1] / nb:.0f} T {v[2] / nb:.0f} gate {v[3] / nb:.0f} barriers {v[4] / nb:.0f}")
