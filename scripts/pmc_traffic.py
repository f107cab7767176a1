#!/usr/bin/env python3
"""HBM traffic per kernel class from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) over
scripts/pmc_target.py.  Counters are KiB; on gfx950 FETCH_SIZE reports half of the streamed read bytes
(MI355X_MICROARCH.md, HBM section; calibrated here on vec_axpy / vec_dot whose byte counts are known), so
reads are doubled; WRITE_SIZE is exact.

usage: pmc_traffic.py FETCH.csv WRITE.csv nx ny out.json"""
import csv, hashlib, json, os, sys
from collections import defaultdict

fetch_csv, write_csv, nx, ny, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from navier_stokes_solver_amd import problem as P

CLASSES = [  # (label, substring(s) of the kernel name, launches per unit, op id or None)
    ("vec_axpy (calibration)", ["vec_axpy"], 1, None),
    ("vec_dot (calibration)", ["vec_dot"], 1, None),
    ("spmv_blk_kernel<2,2> on F", ["spmv_blk_kernel<2, 2>"], 1, 0),
    ("spmv_stream_kernel on S", ["spmv_stream_kernel<1, 0>", "spmv_stream_kernel<3, 0>"], 1, 5),
    ("tri_blk_kernel lower, one ILU(F) apply", ["tri_blk_kernel<1,"], None, 20),
    ("tri_blk_kernel upper, one ILU(F) apply", ["tri_blk_kernel<0,"], None, 20),
    ("tri_blk_sf_kernel lower, one ILU(F) apply", ["tri_blk_sf_kernel<1,"], 1, 20),
    ("tri_blk_sf_kernel upper, one ILU(F) apply", ["tri_blk_sf_kernel<0,"], 1, 20),
    ("tri_stream_sf_kernel lower, one ILU(S) apply", ["tri_stream_sf_kernel<1,"], 1, 21),
    ("tri_stream_sf_kernel upper, one ILU(S) apply", ["tri_stream_sf_kernel<0,"], 1, 21),
]


def per_dispatch(path):
    d = defaultdict(list)
    for r in csv.DictReader(open(path)):
        for label, subs, _, _ in CLASSES:
            if any(s in r["Kernel_Name"] for s in subs):
                d[label].append(float(r["Counter_Value"]))
    return d


F, W = per_dispatch(fetch_csv), per_dispatch(write_csv)
i = P.mesh_info(nx, ny)
n_u, n_p = i["n_u_global"], i["n_p_global"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KSRC = [os.path.join(ROOT, "navier_stokes_solver_amd", "csrc", f) for f in ("nsk_kernels.hip", "nsk_tri.cpp")]
# what the counters were taken from: bench.py quotes the traffic only while these sources are unchanged
sha = hashlib.sha256(b"".join(open(f, "rb").read() for f in KSRC)).hexdigest()
out_d = {"note": __doc__.split("usage")[0].strip(), "mesh": [nx, ny], "kernel_sources": [os.path.relpath(f, ROOT) for f in KSRC],
         "kernel_sources_sha256": sha, "kernels": {}, "by_op": {}}
for label, subs, lpu, op in CLASSES:
    if not F.get(label):
        continue
    f = sum(F[label]) / len(F[label])
    w = sum(W[label]) / len(W[label]) if W.get(label) else 0.0
    n_launch = len(F[label])
    if lpu is None:   # one apply = all level launches of that half: infer from the number of applies (4 per op)
        lpu = max(1, round(n_launch / 4))
    traffic = (2.0 * f + w) * 1024.0 * lpu
    out_d["kernels"][label] = {"FETCH_SIZE_KiB_raw_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
                               "launches_per_unit": lpu, "launches_seen": n_launch,
                               "traffic_bytes_corrected": traffic}
    if op is not None:
        out_d["by_op"].setdefault(str(op), {"traffic_bytes_corrected": 0.0})["traffic_bytes_corrected"] += traffic
json.dump(out_d, open(out, "w"), indent=1)
print(json.dumps(out_d["by_op"], indent=1))
