import sys
sys.path.insert(0, ".")
from tests.debug_nets import run
for (ci, co) in [(80, 14), (64, 64), (64, 14), (80, 64)]:
    for L in [2, 4, 5, 6, 7, 8]:
        run(2, dict(in_channels=ci, out_channels=co, kernel_size=5, layers=L, conv_channels=64), 3, 150)
