"""One rank of tests/test_gpu_comm2.py: takes its contiguous block of the reads2 scalars in `data.npz`, joins the
communicator (id through a file, like the command line does) and runs the library's multi-rank global stage.
usage: _comm2_worker.py RANK WORLD WORKDIR CASE"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from filtlong_amd import api  # noqa: E402

rank, world, work, case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
d = np.load(os.path.join(work, case + ".npz"))
kw = json.loads(str(d["kw"]))
bounds = d["bounds"]
lo, hi = int(bounds[rank]), int(bounds[rank + 1])
ctx = api.Context(0)  # every rank on the one GPU of the box: only the loopback communicator allows that
idf = os.path.join(work, case + ".id")
if rank == 0:
    uid = ctx.comm_unique_id()
    open(idf + ".tmp", "wb").write(uid)
    os.rename(idf + ".tmp", idf)
else:
    for _ in range(3000):
        if os.path.exists(idf):
            break
        time.sleep(0.01)
    uid = open(idf, "rb").read()
ctx.comm_init(uid, rank, world)
total_bases = int(ctx.comm_sum_u64([int(d["length"][lo:hi].astype(np.int64).sum())])[0])
out = ctx.rank_and_cut_comm(d["mean"][lo:hi], d["window"][lo:hi], d["length"][lo:hi], d["passed"][lo:hi],
                            total_bases=total_bases, want_scores=True, **kw)
rep = out["report"]
np.savez(os.path.join(work, "%s.out%d.npz" % (case, rank)), passed=out["passed"], final_score=out["final_score"],
         report=np.array([rep.target_bases, rep.kept_bases, rep.outcome, rep.exact_fallback], dtype=np.int64),
         stats=np.array([rep.mean_quality, rep.stdev_quality, rep.min_z, rep.max_z]), total_bases=total_bases)
ctx.comm_destroy()
ctx.close()
