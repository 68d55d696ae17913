"""Times the Phred scoring kernel alone (HIP events on the library's stream) on synthetic reads generated in HBM.
usage: python tools/bench_phred_kernel.py [n_reads] [window_size] [quality profile 0|1]   (FLX_LIB_PATH selects an experimental build)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from filtlong_amd import api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
ws = int(sys.argv[2]) if len(sys.argv) > 2 else 250
profile = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = api.Context(0)
dev = torch.device("cuda", 0)
lengths = synth.lengths(n)
offsets = np.zeros(n, dtype=np.uint64)
pb = C.c_uint64()
ctx.L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
order = api.length_order(lengths)
d_plane = torch.empty(pb.value, dtype=torch.uint8, device=dev)
d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
d_len = torch.from_numpy(lengths).to(dev)
d_ord = torch.from_numpy(order.view(np.int32)).to(dev)
d_ids = torch.arange(0, n, dtype=torch.int64, device=dev)
d_mean = torch.empty(n, dtype=torch.float64, device=dev)
d_win = torch.empty(n, dtype=torch.float64, device=dev)
d_pass = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx.synth_qual_dev(synth.SEED, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ids.data_ptr(), n, profile=profile)
params = api.make_params(window_size=ws)
for rep in range(6):
    if rep == 1:
        ctx.timing_enable(True)
        ctx.timing_reset()
    ctx.score_reads_dev(d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n, params,
                        d_mean.data_ptr(), d_win.data_ptr(), d_pass.data_ptr())
ms, k = ctx.timing_get("flx_score_phred")
bases = int(lengths.astype(np.int64).sum())
chk = float(d_mean.sum().item()), float(d_win.sum().item())
print("profile %d reads %d bases %d ws %d: kernel %.3f ms avg over %d launches = %.1f Gbases/s   checksum %.6f %.6f" % (
    profile, n, bases, ws, ms / k, k, bases / (ms / k) / 1e6, chk[0], chk[1]))
ctx.close()
