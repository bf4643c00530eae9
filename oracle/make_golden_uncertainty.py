"""Golden for the optional UncertaintyNet (networks.py:91-103), produced by the reference itself (build container
only).  TEST INFRASTRUCTURE ONLY.  Output: tests/golden/uncertainty.npz (weights, input, output)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import _load  # noqa: E402


def main():
    net = _load("networks")
    g = torch.Generator().manual_seed(31)
    u = net.UncertaintyNet(32, 32).eval()
    w1, w2 = torch.randn(32, 33, generator=g), torch.randn(1, 32, generator=g)
    with torch.no_grad():
        u.linear1.weight.copy_(w1)
        u.linear2.weight.copy_(w2)
        u.gain.fill_(0.7)
        x = torch.randn(6, 32, generator=g)
        y = u(x)
    np.savez(os.path.join(ROOT, "tests", "golden", "uncertainty.npz"), w1=w1.numpy(), w2=w2.numpy(), gain=np.float32(0.7),
             x=x.numpy(), y=y.numpy())
    print("uncertainty golden:", y.flatten().tolist())


if __name__ == "__main__":
    main()
