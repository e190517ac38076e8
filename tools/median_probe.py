#!/usr/bin/env python3
"""Diagnostic (GPFQ_DIAG="-DGPFQ_MEDIAN_STAMPS"): where the two passes of the one-GPU median spend their time -- s_memrealtime stamps (10 ns) of
workgroup 0 (zeroing + reads + merge) and of the last workgroup (start -> ticket, pick)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_neural_networks_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 4096
W = torch.from_numpy((np.random.default_rng(0).standard_normal(n) / 64).astype(np.float32)).cuda()
lib = hip.load()
nbytes = lib.gpfq_median_abs_workspace_bytes_for(n)
for rep in range(3):
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    assert lib.gpfq_median_abs(W.data_ptr(), n, out.data_ptr(), ws.data_ptr(), nbytes, None) == 0
    b.record(); torch.cuda.synchronize()
    st = ws[64 + 16:64 + 16 + 48].view(torch.int64).cpu().numpy() / 100.0
    print(f"median {out.item():.6g} in {a.elapsed_time(b) * 1e3:.1f} us; pass 0: workgroup 0 {st[0]:.1f} us, last workgroup to its ticket {st[1]:.1f}, pick {st[2]:.1f}; "
          f"pass 1: {st[3]:.1f} / {st[4]:.1f} / {st[5]:.1f}")
