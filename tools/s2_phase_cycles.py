"""Phase cycles of stack2_fwd_kernel (instrumented build, -DS2_PROF) at the benchmark shape.
Builds crank_amd/libcrank_hip_prof.so next to the product library when missing (hipcc), runs one stack forward per
configuration and prints the per-wave shader cycles of each phase, averaged over the first 256 workgroups.
    python tools/s2_phase_cycles.py [build]"""
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LIB = os.path.join(REPO, "crank_amd", "libcrank_hip_prof.so")


def build():
    csrc = os.path.join(REPO, "crank_amd", "csrc")
    srcs = ["conv_kernels", "stack_kernels", "stack2_kernels", "stack2b_kernels", "pstack_kernels", "pstack2_kernels", "net", "vq_kernels", "loss_kernels", "mlfb_kernels",
            "dataset_kernels", "mcd_kernels"]
    objs = []
    for s in srcs:
        o = os.path.join(csrc, s + (".prof.o" if s == "stack2_kernels" else ".o"))
        if s == "stack2_kernels" or not os.path.exists(o):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] +
                           (["-DS2_PROF"] if s == "stack2_kernels" else []) + ["-c", os.path.join(csrc, s + ".hip"), "-o", o], check=True)
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
    L.crk_debug_s2_prof.argtypes = [ctypes.c_void_p]
    names = ["taps", "gate", "wait A", "1x1+upd", "operand", "wait B", "prologue barrier", "TOTAL", "pro: first conv / state", "pro: tables", "pro: cond tile", "pro: operand put"]
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
                for _ in range(2):
                    m.stack(xi, c=a)
            torch.cuda.synchronize()
            res = np.zeros(1024 * 4, dtype=np.uint64)
            L.crk_debug_s2_res.argtypes = [ctypes.c_void_p]
            assert L.crk_debug_s2_res(res.ctypes.data) == 0
            res = res.reshape(1024, 4)
            live = res[res[:, 1] > 0]
            t0 = live[:, 0].min()
            ev = {}
            for a_, b_, hw, xcc in live:
                key = (int(xcc) & 0xf, (int(hw) >> 13) & 7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 0xf)
                ev.setdefault(key, []).extend([(int(a_), 1), (int(b_), -1)])
            mx = 0
            for k, lst in ev.items():
                c = 0
                for _, d in sorted(lst):
                    c += d
                    mx = max(mx, c)
            dur = (live[:, 1] - live[:, 0]).astype(np.float64) / 100.0
            print(f"   residency: {len(live)} workgroups on {len(ev)} CUs, max co-resident per CU {mx}, workgroup life {dur.mean():.1f} us (min {dur.min():.1f} max {dur.max():.1f}), "
                  f"kernel span {(live[:, 1].max() - t0) / 100.0:.1f} us, last start at {(live[:, 0].max() - t0) / 100.0:.1f} us")
            buf = np.zeros(256 * 8 * 16, dtype=np.uint64)
            assert L.crk_debug_s2_prof(buf.ctypes.data) == 0
            v = buf.reshape(256, 8, 16)[:, :, :12].astype(np.float64)
            print(f"{tag} {'saving' if grad else 'no-grad'}: cycles per wave (mean over 256 workgroups); waves 0-3 = frame half 0, tiles 0,1 residual / 2,3 skip")
            for w in range(8):
                print(f"  wave {w}: " + "  ".join(f"{n} {v[:, w, i].mean():8.0f}" for i, n in enumerate(names)))
            mean = v.mean(axis=(0, 1))
            print("  all   : " + "  ".join(f"{n} {mean[i]:8.0f}" for i, n in enumerate(names)), f"| per block: taps {mean[0]/layers:.0f} gate {mean[1]/layers:.0f} waitA {mean[2]/layers:.0f} 1x1 {mean[3]/layers:.0f} operand {mean[4]/layers:.0f} waitB {mean[5]/layers:.0f}")
