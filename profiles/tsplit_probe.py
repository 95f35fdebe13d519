#!/usr/bin/env python3
"""The two gathered-operand GEMMs at Reddit's hop-1 shape (77k rows, F = 602 + 3, 256 outputs) in a loop, for
rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grapes_amd import ops
rng = np.random.default_rng(0)
N, F, ni, fo, n = 232965, 602, 3, 256, 77015
X = torch.randn(N, F, device="cuda"); Xp, _ = ops.pad_features(X)
ids = torch.from_numpy(np.sort(rng.permutation(N)[:n]).astype(np.int32)).cuda()
code = torch.zeros(N, dtype=torch.int32, device="cuda")
W = torch.randn(fo, F + ni, device="cuda") * 0.04
img = ops.weight_split_image(W)
kp = (F + ni + 3) // 4 * 4
Wp = torch.zeros(fo, kp, device="cuda"); Wp[:, :F + ni] = W
dh = torch.randn(n, fo, device="cuda")
dW = torch.zeros(fo, kp, device="cuda")
for r in range(30):
    ops.linear_fwd_gathered(Xp, F, ids, Wp, code, 0, ni, w_image=img)
    ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dW, code, 0, ni, split=True)
torch.cuda.synchronize(); print("ok")
