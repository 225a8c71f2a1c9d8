"""Profile driver: the speaker classifier C and the speaker-adversarial net forward + backward at the benchmark shape
(used under rocprofv3 by tools/ps_ablate.sh): python tools/prof_c.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crank_amd import ops  # noqa: E402
from crank_amd.bin.train import get_model  # noqa: E402
from crank_amd.utils import load_yaml  # noqa: E402

ops.set_precision("bf16")
conf = load_yaml(None, batch_size=64, batch_len=500)
m = get_model(conf, 14, "cuda")
x = torch.randn(64, 500, 80, device="cuda", requires_grad=True)
e = torch.randn(64, 500, 128, device="cuda", requires_grad=True)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    y = m["C"](x.transpose(1, 2))
    y.sum().backward()
    z = m["SPKRADV"]([e[..., :64], e[..., 64:]])
    z.sum().backward()
torch.cuda.synchronize()
print("done")
