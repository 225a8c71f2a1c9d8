import os, sys, ctypes
sys.path.insert(0,'/root/repo')
os.environ["CRANK_AMD_LIB"]="/root/repo/crank_amd/libcrank_hip_vqprof.so"
import numpy as np, torch
from crank_amd import ops, _lib
x=torch.randn(64,500,64,device="cuda"); w=torch.randn(512,64,device="cuda")*0.7
for _ in range(3): ops.vq_apply(x,w)
torch.cuda.synchronize()
L=_lib.lib(); L.crk_debug_vq_prof.argtypes=[ctypes.c_void_p]
b=np.zeros(1024,dtype=np.uint64); L.crk_debug_vq_prof(b.ctypes.data); b=b.reshape(256,4).astype(float)
print("cycles: after staging %.0f, after search %.0f, end %.0f"%tuple(b[:250,:3].mean(0)))
