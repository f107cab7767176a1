#!/usr/bin/env python3
"""bench.py — DoF·iters/s of the FGMRES + aSIMPLE velocity-pressure solve on MI355X.

Metric (BASELINE.json): DoF·iters/sec (FGMRES+aSIMPLE, Re=100) at 1/2/4/8 GPUs; achieved HBM GB/s.

Workload
  N = 1   BASELINE configs[2]: stationary 1200x400, Re=100 (nu = 1/90, Newton system), FGMRES + aSIMPLE on one MI355X.
  N >= 2  BASELINE configs[3] (north_star): stationary 4800x1600, Re=200 (nu = 1/190), FGMRES + aSIMPLE, row-partitioned
          in x-strips over the N GPUs (STRONG scaling: halo exchange + all-reduce over RCCL) — whenever its working set
          (about 3.9 KB per DoF: blocks, factors, node-block copies, Krylov bases) fits the N GPUs, i.e. from N = 4;
          otherwise (N = 2) the weak-scaling mesh 1200 N x 400 at Re=100.  `--scaling weak|strong` and `--mesh` override.

A step is ONE outer FGMRES iteration: aSIMPLE apply (inner FGMRES on F with ILU(0), B, inner CG on S with ILU(0), B^T,
D^-1) + jacobian SpMV + modified Gram-Schmidt + least squares/check.  The timed region is `solve_system`'s solver.solve()
limited to exactly K iterations from the initial guess (tolerance 0: nothing converges in K steps — the inner iteration
counts, hence the time per step, grow along the Krylov space, so K is part of the metric label), inputs resident in HBM;
the preconditioner set-up (diag, SpGEMM, 2x ILU(0)) is reported separately.

One process per GPU: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
HBM_BYTES = 288e9
BYTES_PER_DOF = 3.9e3          # measured working set of FGMRES + aSIMPLE at 1200x400 (41 GB / 10.48 M DoFs)
N_DOFS_NORTH_STAR = 167_545_276  # 4800x1600
NAMES_S = ["GMRES", "FGMRES", "Bicgstab"]
NAMES_P = ["blockDiagonal", "blockTriangular", "aSIMPLE"]


_T0 = time.time()
_PHASE = ["start"]


def say(msg):
    """Progress line on stderr (rank 0 of a multi-rank run prints too little otherwise: a silent run looks hung)."""
    _PHASE[0] = msg
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def start_pulse(period=60.0):
    """One line per minute while a long host-side phase (hand-off generation, symbolic analysis) runs."""
    def run():
        while True:
            time.sleep(period)
            if int(os.environ.get("RANK", "0")) == 0:
                print(f"[bench {time.time() - _T0:7.1f}s] ... still in: {_PHASE[0]}", file=sys.stderr, flush=True)
    threading.Thread(target=run, daemon=True).start()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=58)     # two full restart cycles of 29 counted iterations
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mesh", type=str, default="", help="X,Y: per-GPU mesh (weak) or global mesh (strong); default per --scaling")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto")
    ap.add_argument("--lx", type=float, default=2.2,
                    help="channel length of the generated mesh (default: the reference's 2.2).  --mesh 600,1600 --lx 0.275 is "
                         "ONE RANK'S SHARE of BASELINE configs[3] (4800x1600 on 8 ranks) as a channel of its own: same cells, "
                         "same lattice height, same sizes per rank; 600x1600 on the whole channel would stretch the cells 8x")
    ap.add_argument("--reynolds", type=float, default=0.0, help="default: 100 (N = 1, weak) / 200 (strong: configs[3])")
    ap.add_argument("--solver", type=int, default=1)
    ap.add_argument("--preconditioner", type=int, default=2)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--ordering", type=int, default=1, help="0 natural, 1 multicolour (triangular solves)")
    ap.add_argument("--subdomains", type=int, default=1)
    ap.add_argument("--sync-free", type=int, default=2, help="0 per-colour launches, 1 single-launch S/Mp solves, 2 also F")
    ap.add_argument("--cg-single-reduction", type=int, default=-1,
                    help="inner CG with one fused all-reduce per iteration (NSK_OPT_CG_SINGLE_REDUCTION); default: on for N > 1")
    ap.add_argument("--inner-gs", type=int, default=-1,
                    help="Gram-Schmidt of the inner FGMRES: 0 modified, 1 fused classical (default on one GPU), 2 fused classical "
                         "with one reduction per iteration (default for N > 1)")
    ap.add_argument("--line-groups", type=int, default=2,
                    help="NSK_OPT_TRI_LINE_GROUPS: 1 colour pairs of velocity nodes / triples of pressure DoFs along the lattice "
                         "lines in the triangular factors' orderings (fewer colours), 0 colour single DoFs, 2 (default) by size")
    ap.add_argument("--schur-sign", type=int, default=1,
                    help="NSK_OPT_SCHUR_SIGN of the --converge solve ONLY: -1 negates aSIMPLE's S (labelled deviation from the "
                         "reference, off by default; the timed region always runs the reference's +1)")
    ap.add_argument("--blas1-pairs", type=int, default=-1, help="NSK_OPT_BLAS1_PAIRS: -1 by variant (default), 0, 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-mesh", type=str, default="",
                    help="mesh of the CPU baseline's first sample; default: the bench mesh itself (1200,400 at N = 1)")
    ap.add_argument("--cpu-steps", type=int, default=1,
                    help="outer iterations of the CPU baseline's first sample (the oracle needs ~60-90 s for ONE at 1200x400 on "
                         "16 cores; three took 277 s, profiles/r04_bench_line_K20_cpu_sample_K3.json)")
    ap.add_argument("--cpu-mesh2", type=str, default="300,100", help="second, smaller CPU sample ('' = none)")
    ap.add_argument("--cpu-steps2", type=int, default=12)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--converge", type=float, default=0.0, help="if > 0: also one full solve to this tolerance (see --converge-*)")
    ap.add_argument("--converge-mesh", type=str, default="300,100")
    ap.add_argument("--converge-preconditioner", type=int, default=0, help="default: BASELINE configs[1], FGMRES + blockDiagonal")
    ap.add_argument("--converge-budget", type=float, default=600.0, help="seconds after which the solve is cancelled")
    return ap.parse_args()


def choose_workload(world, scaling="auto", mesh="", reynolds=0.0):
    """(scaling, nx, ny, Reynolds number) of a run on `world` GPUs.  N = 1: BASELINE configs[2] (1200x400, Re 100).
    N >= 2: the north star's workload — strong scaling on 4800x1600, Re 200 — whenever a rank's share fits its HBM
    (3.9 KB per DoF measured; from N = 4), otherwise weak scaling with 1200x400 per GPU.  `mesh` is the per-GPU mesh
    (weak) or the global one (strong); explicit arguments win."""
    if scaling == "auto":
        scaling = "weak"
        if world > 1 and not mesh and BYTES_PER_DOF * N_DOFS_NORTH_STAR / world <= 0.8 * HBM_BYTES:
            scaling = "strong"
    if mesh:
        mx, my = (int(v) for v in mesh.split(","))
    else:
        mx, my = (4800, 1600) if scaling == "strong" else (1200, 400)
    nx = mx * world if scaling == "weak" else mx
    if reynolds <= 0.0:
        reynolds = 200.0 if (scaling == "strong" and not mesh) else 100.0
    return scaling, nx, my, reynolds


def make_solver(S, PT, P, dist, args, nx, ny, nu, inv_dt, world, rank, local_rank, lx=2.2):
    """Generate this rank's hand-off, create the handle, upload.  Returns (ls, pr, n_global, t_gen, t_upload)."""
    t0 = time.time()
    say(f"generating the hand-off of {nx}x{ny} (rank {rank} of {world}) on the host")
    pr = P.generate(nx, ny, nu=nu, mode=1, state=1, inv_dt=inv_dt, U=0.1 if args.variant == 0 else 0.3,
                    nranks=world, rank=rank, lx=lx)
    t_gen = time.time() - t0
    say(f"hand-off ready ({t_gen:.1f} s): n_u {pr.n_u}, n_p {pr.n_p}, nnz(F) {pr.F.nnz}; uploading blocks")
    n_global = int(pr.info["n_u_global"] + pr.info["n_p_global"])
    uid, plan = None, None
    if world > 1:
        box = [S.get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
        gu, gp = [None] * world, [None] * world
        dist.all_gather_object(gu, pr.ghost_u)
        dist.all_gather_object(gp, pr.ghost_p)
        plan = {S.SPACE_U: PT.build_halo_plan(rank, pr.u_ranges, gu),
                S.SPACE_P: PT.build_halo_plan(rank, pr.p_ranges, gp)}
    ls = S.LinearSolver(rank, world, local_rank, uid)
    ls.set_option(S.OPT_TRI_ORDERING, args.ordering)
    ls.set_option(S.OPT_SUBDOMAINS, args.subdomains)
    ls.set_option(S.OPT_TRI_SYNC_FREE, args.sync_free)
    ls.set_option(S.OPT_TRI_LINE_GROUPS, args.line_groups)
    ls.set_option(S.OPT_CG_SINGLE_REDUCTION, int(args.cg_single_reduction if args.cg_single_reduction >= 0 else world > 1))
    ls.set_option(S.OPT_INNER_FUSED_GS, int(args.inner_gs if args.inner_gs >= 0 else (2 if world > 1 else 1)))
    ls.set_option(S.OPT_BLAS1_PAIRS, args.blas1_pairs)
    t0 = time.time()
    ls.set_problem(pr, plan)
    say(f"blocks on the device ({time.time() - t0:.1f} s)")
    return ls, pr, n_global, t_gen, time.time() - t0


def timed_steps(ls, pr, solver, prec, variant, steps, warmup, barrier, sync):
    """W untimed + exactly K timed outer iterations from the same initial state, fresh preconditioner object each."""
    ls.setup_preconditioner(prec, variant, 0.5)
    ls.upload_system(pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    ls.solve_resident(solver, 0.0, max(1, warmup))   # (falls back to per-colour launches by itself if a hand-off gives up)
    ls.setup_preconditioner(prec, variant, 0.5)
    ls.upload_system(pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    ls.reset_stats()
    barrier()
    sync()
    t0 = time.perf_counter()
    its, res, rc = ls.solve_resident(solver, 0.0, steps)
    sync()
    barrier()
    return its, res, rc, time.perf_counter() - t0


def cpu_baseline(args, nu, mesh, steps):
    """Oracle (CPU restatement of the reference path, kind 'port') on a bounded sample of the same workload: same
    solver / preconditioner / Reynolds number on a smaller mesh, a few outer iterations, run the way the reference runs
    on a node: one emulated MPI rank per host core (x-strip shards, block-Jacobi ILU(0) = Ifpack overlap 0), OpenMP
    threads standing in for the ranks."""
    import numpy as np
    from navier_stokes_solver_amd import problem as P
    from oracle import oracle as O
    nx, ny = (int(v) for v in mesh.split(","))
    say(f"CPU baseline: oracle on {nx}x{ny}, {steps} outer iterations")
    from navier_stokes_solver_amd._threads import cpu_budget
    cores = max(1, min(args.cpu_threads or min(cpu_budget(), 16), nx // 4))   # a one-GPU box's CPU share is 16 cores
    pr = P.generate(nx, ny, nu=nu, mode=1, state=1)
    ranges = P.generate(nx, ny, nu=nu, mode=1, state=1, nranks=cores, rank=0) if cores > 1 else None
    kw = dict(u_shard_off=ranges.u_ranges, p_shard_off=ranges.p_ranges) if ranges is not None else {}
    O.set_threads(cores)
    try:
        op = O.OracleProblem.from_local(pr, **kw)
        b = np.concatenate([pr.rhs_u, pr.rhs_p])
        x0 = np.concatenate([pr.x0_u, pr.x0_p])
        t0 = time.time()
        _, info = op.solve(b, x0, solver=args.solver, prec=args.preconditioner, variant=args.variant, tol=0.0,
                           max_iter=steps)
        wall = time.time() - t0
    finally:
        O.set_threads(1)
    its = max(1, info["iters"])
    return {
        "value": pr.n * its / info["solve_seconds"], "unit": "DoF*iters/s", "cores": cores, "kind": "port",
        "sample": f"oracle (C restatement, {cores} OpenMP threads = {cores} emulated MPI ranks: {cores} x-strip shards, each "
                  f"with its own ILU(0) in NATURAL order = Ifpack overlap 0, how the reference runs on a CPU node; the GPU "
                  f"pair below is ONE shard in multicolour order, so the inner iteration counts per step differ — "
                  f"dof_inner_iters_per_s counts the work actually done), "
                  f"stationary {nx}x{ny} Re={args.reynolds:g}, first {its} outer iterations; "
                  f"solve {info['solve_seconds']:.1f}s + setup {info['setup_seconds']:.1f}s (wall {wall:.1f}s)",
        "incl_setup_value": pr.n * its / (info["solve_seconds"] + info["setup_seconds"]),
        "mesh": f"{nx}x{ny}", "K": its,
        "inner_F_its_per_step": info["inner_u_its"] / max(1, info["prec_applies"]),
        "inner_S_its_per_step": info["inner_p_its"] / max(1, info["prec_applies"]),
        # inner-iteration throughput: (velocity DoFs x F iterations + pressure DoFs x S iterations) per second of solve
        "dof_inner_iters_per_s": (pr.n_u * info["inner_u_its"] + pr.n_p * info["inner_p_its"]) / info["solve_seconds"],
        "ordering": "natural, per shard", "shards": cores,
    }


def converged_solve(S, PT, P, dist, args, world, rank, local_rank, sync):
    """One full solve to a tolerance with a stated time budget (default: BASELINE configs[1], 300x100 FGMRES +
    blockDiagonal).  A heartbeat prints outer iterations and the current residual; past the budget the solve is
    cancelled (nsk_cancel) and reported as such — a run can no longer end without a record."""
    import numpy as np
    nx, ny = (int(v) for v in args.converge_mesh.split(","))
    nu = P.reynolds_to_nu(100.0, stationary=True)
    ls, pr, n_global, _, _ = make_solver(S, PT, P, dist, args, nx, ny, nu, 0.0, world, rank, local_rank)
    prec = args.converge_preconditioner
    if args.schur_sign != 1:
        ls.set_option(S.OPT_SCHUR_SIGN, args.schur_sign)
    ls.setup_preconditioner(prec, 0, 0.5)
    ls.upload_system(pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    sync()
    done = threading.Event()
    cancelled = [False]
    t_start = time.time()

    def heartbeat():   # the C call blocks (GIL released): report progress, enforce the budget
        while not done.wait(20.0):
            st = ls.stats()
            el = time.time() - t_start
            print(f"[bench] converging {nx}x{ny} {NAMES_P[prec]}: {el:.0f} s, outer iteration {st['cur_outer_iters']}, "
                  f"residual {st['cur_residual']:.3e} (target {args.converge:g})", file=sys.stderr, flush=True)
            if el > args.converge_budget and not cancelled[0]:
                cancelled[0] = True
                ls.cancel()

    hb = threading.Thread(target=heartbeat, daemon=True)
    hb.start()
    t0 = time.perf_counter()
    its, res, rc = ls.solve_resident(args.solver, args.converge, 20000)
    sync()
    dt = time.perf_counter() - t0
    done.set()
    out = {"workload": f"stationary {nx}x{ny} Re=100 (nu=1/90) Newton system, {NAMES_S[args.solver]} + {NAMES_P[prec]}"
                       + ("" if args.schur_sign == 1 else " with NSK_OPT_SCHUR_SIGN = -1 (S negated: a LABELLED DEVIATION from the reference)"),
           "schur_sign": args.schur_sign,
           "dofs": n_global, "tol": args.converge, "iters": its, "final_res": res, "status": rc, "seconds": dt,
           "cancelled_after_budget_s": args.converge_budget if cancelled[0] else None,
           "dof_iters_per_s": n_global * its / dt}
    if world == 1 and rc == 0 and n_global <= 20_000_000:
        xu, xp = ls.download_solution()
        J = pr.jacobian_scipy()
        out["true_residual"] = float(np.linalg.norm(np.concatenate([pr.rhs_u, pr.rhs_p]) - J @ np.concatenate([xu, xp])))
    st = ls.stats()
    out["inner_F_its_per_step"] = st["inner_u_its"] / max(1, st["prec_applies"])
    out["inner_P_its_per_step"] = st["inner_p_its"] / max(1, st["prec_applies"])
    ls.close()
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if "OMP_NUM_THREADS" not in os.environ or (os.environ["OMP_NUM_THREADS"] == "1" and
                                               (world > 1 or "TORCHELASTIC_RUN_ID" in os.environ)):
        # The host-side hand-off generation, the one-off symbolic analysis and the CPU baseline are OpenMP loops:
        # size their teams after the cgroup CPU quota (the boxes show 256 CPUs for a 16-core share), split over
        # the ranks of this node (torch.distributed.run pins OMP_NUM_THREADS=1).  Set before any OpenMP runtime loads.
        from navier_stokes_solver_amd._threads import cpu_budget
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(32, cpu_budget() // world)))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solve path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        # rendezvous / barriers / object exchange on gloo; the data path uses RCCL inside libnsk_hip.so
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from navier_stokes_solver_amd import partition as PT
    from navier_stokes_solver_amd import problem as P
    from navier_stokes_solver_amd import solver as S

    # ---- workload (see the module docstring)
    scaling, nx, ny, args.reynolds = choose_workload(world, args.scaling, args.mesh, args.reynolds)
    nu = P.reynolds_to_nu(args.reynolds, stationary=(args.variant == 0))
    inv_dt = 0.0 if args.variant == 0 else 100.0

    def barrier():
        if world > 1:
            dist.barrier()

    sync = torch.cuda.synchronize
    start_pulse()
    ls, pr, n_global, t_gen, t_upload = make_solver(S, PT, P, dist, args, nx, ny, nu, inv_dt, world, rank, local_rank,
                                                    lx=args.lx)
    t0 = time.time()
    say("first preconditioner set-up (symbolic analysis on the host + numeric phase on the device)")
    ls.setup_preconditioner(args.preconditioner, args.variant, 0.5)
    t_setup_first = time.time() - t0     # includes the one-off symbolic analysis
    t0 = time.time()
    say(f"first set-up done ({t_setup_first:.1f} s); numeric set-up again")
    ls.setup_preconditioner(args.preconditioner, args.variant, 0.5)
    t_setup = time.time() - t0           # numeric refactorisation only (what every Newton step pays)
    say(f"numeric set-up {t_setup:.2f} s; warm-up of {max(1, args.warmup)} outer iterations")

    aS = args.preconditioner == 2
    ops = (20, 0, 21) + ((5,) if aS else (3,))
    # warm-up + timed region; HIP events around every launch of the sampled ops inside the timed solve
    ls.setup_preconditioner(args.preconditioner, args.variant, 0.5)
    ls.upload_system(pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    ls.solve_resident(args.solver, 0.0, max(1, args.warmup))   # (falls back to per-colour launches by itself if needed)
    ls.setup_preconditioner(args.preconditioner, args.variant, 0.5)
    ls.upload_system(pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    ls.reset_stats()
    for op in ops:
        ls.profile_begin(op, 1024)
    say(f"timed region: {args.steps} outer iterations")
    barrier()
    sync()
    t0 = time.perf_counter()
    its, res, rc = ls.solve_resident(args.solver, 0.0, args.steps)
    sync()
    barrier()
    dt = time.perf_counter() - t0
    say(f"timed region done: {dt:.2f} s")
    prof = {op: ls.profile_read(op) for op in ops}
    ls.profile_end()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st = ls.stats()
    free_b, total_b = torch.cuda.mem_get_info()       # everything of this solve is still resident
    dev_bytes = int(total_b - free_b)
    assert its == args.steps, (its, args.steps)
    n_u_local, n_p_local, nnz_F_local = pr.n_u, pr.n_p, pr.F.nnz

    # ---- like-for-like pairs of the CPU baseline: the same K iterations on the CPU samples' meshes, on the GPU
    def gpu_pair(lsx, prx, nx_, ny_, K):
        i2, r2, _, dt2 = timed_steps(lsx, prx, args.solver, args.preconditioner, args.variant, K, 2, lambda: None, sync)
        st2 = lsx.stats()
        return {"value": prx.n * i2 / dt2, "unit": "DoF*iters/s", "mesh": f"{nx_}x{ny_}", "K": i2,
                "ms_per_step": 1e3 * dt2 / max(1, i2),
                "inner_F_its_per_step": st2["inner_u_its"] / max(1, st2["prec_applies"]),
                "inner_S_its_per_step": st2["inner_p_its"] / max(1, st2["prec_applies"]),
                "dof_inner_iters_per_s": (prx.n_u * st2["inner_u_its"] + prx.n_p * st2["inner_p_its"]) / dt2,
                "ordering": ["natural", "multicolour"][args.ordering], "shards": args.subdomains}

    cpu_samples = []     # (mesh string, K, GPU pair)
    if world == 1 and not args.no_cpu_baseline:
        m1 = args.cpu_mesh or f"{nx},{ny}"
        for mesh_s, K in ((m1, args.cpu_steps), (args.cpu_mesh2, args.cpu_steps2)):
            if not mesh_s or K <= 0 or any(mesh_s == c[0] for c in cpu_samples):
                continue
            cx, cy = (int(v) for v in mesh_s.split(","))
            if (cx, cy) == (nx, ny) and args.lx == 2.2:
                say(f"GPU pair of the CPU sample on the bench mesh: the first {K} outer iterations")
                cpu_samples.append((mesh_s, K, gpu_pair(ls, pr, cx, cy, K)))
    ls.close()
    del pr
    if world == 1 and not args.no_cpu_baseline:
        for mesh_s, K in ((m1, args.cpu_steps), (args.cpu_mesh2, args.cpu_steps2)):
            if not mesh_s or K <= 0 or any(mesh_s == c[0] for c in cpu_samples):
                continue
            cx, cy = (int(v) for v in mesh_s.split(","))
            ls2, pr2, n2, _, _ = make_solver(S, PT, P, dist, args, cx, cy, nu, inv_dt, 1, 0, local_rank)
            cpu_samples.append((mesh_s, K, gpu_pair(ls2, pr2, cx, cy, K)))
            ls2.close()
            del pr2

    conv = None
    if args.converge > 0:
        conv = converged_solve(S, PT, P, dist, args, world, rank, local_rank, sync)

    if rank == 0:
        value = n_global * args.steps / dt
        lu, lp = st["n_colors_u"] or st["n_levels_u"], st["n_colors_p"] or st["n_levels_p"]
        names = {0: "spmv_blk_kernel<2,2>: SpMV with F (inner FGMRES), 2x2 node blocks",
                 3: "spmv_stream_kernel<3,0>: SpMV with Mp (inner CG), pairs of entries per lane", 5: "spmv_stream_kernel<3,0>: SpMV with S (inner CG), pairs of entries per lane",
                 20: (f"tri_blk_kernel: ILU(0)/SGS apply on F ({lu}+{lu} node-colour level launches "
                      "of one apply)" if args.sync_free != 2 else
                      "tri_blk_sf_kernel: ILU(0)/SGS apply on F (one launch per half, in-kernel hand-off)"),
                 21: ("tri_ring_kernel: natural-order ILU(0)/SGS apply on the pressure mass matrix (gather + one one-workgroup launch "
                      "per half through an LDS ring; what bounds it is the chain of dependent levels on ONE CU, not HBM)"
                      if st.get("ring_applies", 0) > 0 else
                      "tri_stream_sf_kernel: ILU(0)/SGS apply on the pressure block (one launch per half, in-kernel hand-off)"
                      if args.sync_free >= 1 else
                      f"tri_stream_kernel: ILU(0)/SGS apply on the pressure block ({lp}+{lp} level "
                      "launches of one apply)")}
        klass = {}
        for op, (ms, cnt, by, ncalls, byf) in prof.items():
            ach = (by / 1e9) / (ms / 1e3) if ms > 0 else 0.0
            achf = (byf / 1e9) / (ms / 1e3) if ms > 0 else 0.0
            klass[op] = dict(kernel=names[op], avg_ms=ms, launches_sampled=cnt, calls=ncalls,
                             bytes_csr_algorithmic=by, bytes_format=byf,
                             achieved_algorithmic=ach, achieved_format=achf,
                             # fraction of the HBM peak from the bytes the storage format really streams: cannot exceed 1
                             frac=achf / HBM_PEAK_GBS, frac_algorithmic=ach / HBM_PEAK_GBS,
                             time_share=ncalls * ms / 1e3 / dt)   # share of the timed solve spent in this class
        dom = max(klass, key=lambda o: klass[o]["time_share"])
        D = klass[dom]
        # HBM traffic of the dominant class from the committed PMC passes (rocprofv3 --pmc cannot run inside
        # this process); only quoted when it was measured on this very workload
        traffic, traffic_src, traffic_stale = None, None, None
        import hashlib
        ksrc = [os.path.join(ROOT, "navier_stokes_solver_amd", "csrc", f) for f in ("nsk_kernels.hip", "nsk_tri.cpp")]
        sha_now = hashlib.sha256(b"".join(open(f, "rb").read() for f in ksrc)).hexdigest()
        for pmc_name in ("r04_pmc_traffic_1200x400.json", "r03_pmc_traffic_1200x400.json", "r02_pmc_traffic_1200x400.json"):
            pmc = os.path.join(ROOT, "profiles", pmc_name)
            if (nx, ny, world, args.lx) == (1200, 400, 1, 2.2) and os.path.exists(pmc):
                rec = json.load(open(pmc))
                kk = rec.get("by_op", {})
                if str(dom) in kk:
                    traffic = kk[str(dom)]["traffic_bytes_corrected"]
                    traffic_src = f"profiles/{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
                    # the counters belong to the kernel sources they were taken from (sha recorded by scripts/pmc_traffic.py)
                    traffic_stale = rec.get("kernel_sources_sha256") != sha_now
                    break
        label = f"{NAMES_S[args.solver]}+{NAMES_P[args.preconditioner]}, Re={args.reynolds:g}"
        out = {
            "metric": f"DoF*iters/s ({label}; first K={args.steps} outer iterations)",
            "value": value, "unit": "DoF*iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{'stationary' if args.variant == 0 else 'unsteady (dt=0.01)'} {nx}x{ny} Q3/Q2"
                            f"{'' if args.lx == 2.2 else f' on the leading {args.lx:g} of the 2.2 x 0.41 channel'}, "
                            f"Re={args.reynolds:g} (nu=1/{1 / nu:g}) Newton system, "
                            f"solver {NAMES_S[args.solver]} + {NAMES_P[args.preconditioner]}",
                "K": args.steps, "restart": 30, "tolerance": 0.0,
                "dofs": n_global, "n_u_local": n_u_local, "n_p_local": n_p_local, "nnz_F_local": nnz_F_local,
                "nnz_S_local": st["nnz_s"], "partition": f"x-strips x{world}",
                "tri_ordering": ["natural", "multicolor"][args.ordering], "line_groups": bool(args.line_groups),
                "colors_u": st["n_colors_u"],
                "colors_p": st["n_colors_p"], "subdomains_per_gpu": args.subdomains,
                "inner_F_its_per_step": st["inner_u_its"] / max(1, st["prec_applies"]),
                "inner_S_its_per_step": st["inner_p_its"] / max(1, st["prec_applies"]),
                "residual_after_K": res,
                "inner_cg": "single-reduction (Chronopoulos-Gear)" if (args.cg_single_reduction if args.cg_single_reduction >= 0
                                                                      else world > 1) else "deal.II recurrence",
                "inner_gram_schmidt": ["modified", "fused classical", "fused classical, one reduction per iteration"][
                    int(args.inner_gs if args.inner_gs >= 0 else (2 if world > 1 else 1))],
                "blas1_reductions": "16-byte loads (pairs)" if (args.blas1_pairs if args.blas1_pairs >= 0 else int(args.variant == 0))
                                    else "8-byte loads",
            },
            "roofline": {
                "bound": "hbm", "kernel": D["kernel"], "time_share": D["time_share"],
                "achieved": D["achieved_algorithmic"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": D["achieved_algorithmic"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                "bytes_per_launch": D["bytes_csr_algorithmic"], "bytes_format": D["bytes_format"],
                "frac_format": D["frac"], "avg_ms": D["avg_ms"], "launches_sampled": D["launches_sampled"],
            },
            "kernel_classes": [klass[o] for o in sorted(klass)],
            "phases": {
                "generate_s": t_gen, "upload_s": t_upload, "setup_first_s": t_setup_first, "setup_numeric_s": t_setup,
                "solve_s": dt, "spmv_GB": st["spmv_bytes"] / 1e9, "tri_GB": st["tri_bytes"] / 1e9,
                "blas1_GB": st["blas1_bytes"] / 1e9,
                "algorithmic_GBps_whole_solve": (st["spmv_bytes"] + st["tri_bytes"] + st["blas1_bytes"]) / 1e9 / dt,
                "host_syncs": st["host_syncs"], "reductions": st["reductions"],
                "host_syncs_per_step": st["host_syncs"] / args.steps, "reductions_per_step": st["reductions"] / args.steps,
                "sync_free_fallbacks": st["sync_free_fallbacks"],
                "spmvs_overlapped_with_halo_per_step": st["overlapped_spmvs"] / args.steps,
                "dof_iters_per_s_incl_setup": n_global * args.steps / (dt + t_setup),
                "device_bytes_in_use": dev_bytes, "device_KB_per_dof": dev_bytes / 1e3 / (n_u_local + n_p_local),
            },
        }
        if conv:
            out["converged_solve"] = conv
        if world == 1 and not args.no_cpu_baseline:
            # first sample: the bench mesh itself (a few outer iterations: the oracle needs tens of seconds for one at
            # 1200x400); second sample: a mesh the oracle does a dozen iterations on
            cbs = []
            for mesh_s, K, pair in cpu_samples:
                cb = cpu_baseline(args, nu, mesh_s, K)
                cb["gpu_same_mesh"] = pair    # the like-for-like pair: same mesh, same K
                cb["gpu_over_cpu_same_mesh"] = pair["value"] / cb["value"]
                cb["gpu_over_cpu_inner_work"] = pair["dof_inner_iters_per_s"] / cb["dof_inner_iters_per_s"]
                cbs.append(cb)
            if cbs:
                out["cpu_baseline"] = cbs[0]
                if len(cbs) > 1:
                    out["cpu_baseline"]["second_sample"] = cbs[1]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
