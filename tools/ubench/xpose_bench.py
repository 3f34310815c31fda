#!/usr/bin/env python3
"""Transpose kernels (brbd_transpose_to_*) on derivative-sized matrices: GB/s of read + write."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pinocchio_amd import capi

L = capi.lib()
for dof, B in ((324, 65536), (324, 262144), (19, 262144), (1024, 65536)):
    bm = torch.randn((dof, B), dtype=torch.float64, device="cuda")
    im = torch.empty((B, dof), dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    res = {"dof": dof, "B": B}
    for name, fn in (("to_im", lambda: L.brbd_transpose_to_instance_major(0, dof, B, bm.data_ptr(), B, im.data_ptr(), dof, st)),
                     ("to_bm", lambda: L.brbd_transpose_to_batch_major(0, dof, B, im.data_ptr(), dof, bm.data_ptr(), B, st))):
        for _ in range(3): capi.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / 10
        res[name + "_us"] = t * 1e6
        res[name + "_GBps"] = 2 * dof * B * 8 / t / 1e9
    assert torch.equal(im.t(), bm)
    print(json.dumps(res))
