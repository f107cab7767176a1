"""Command-line drivers with the reference's flag surface.

    python -m navier_stokes_solver_amd.cli StationaryNSSolver -m 60,20 -r 20 -s 1 -p 0
    python -m navier_stokes_solver_amd.cli NSSolver -T 5,0.01 -m 600,200 -r 100

Flags, defaults, help text and the configuration echo follow `lab_new/src/testStationary.cpp:7-123`
and `lab_new/src/test.cpp:8-146` (getopt string "M:m:r:s:t:p:h" / "T:M:m:r:s:t:p:h", so `-M` swallows
the next token exactly as in the reference).  What runs is the hot path only: for every continuation
level the reference would visit (`NSSolverStationary.cpp:662-665`, `NSSolver.cpp:684`):
`StationaryNSSolver` runs the reference's whole `solve_newton()` (continuation, Stokes phase, Newton iterations
with backtracking) with assembly, linear solves and vector updates resident on the GPU (`newton.py`);
`NSSolver` runs the reference's time loop (`NSSolver::solve()`, one `solve_newton()` per step with the mass term
and the `solution_old` term in the device assembly).  `-M FILE`: P2/P1 on a gmsh triangle mesh — hand-off producer `simplex.py`, device assembly on
the general cells, the same GPU solves.
"""
from __future__ import annotations

import getopt
import os
import sys
import time

SOLVERS = {0: "GMRES", 1: "FGMRES", 2: "Bicgstab"}
PRECS = {0: "blockDiagonal", 1: "blockTriangular", 2: "aSIMPLE"}


def print_help(unsteady: bool):
    out = "Usage: ./NSSolver [options]\n\nOptions:\n"
    if unsteady:
        out += "  -T, --timespan-step T,dt  Set time span and time step (two floating point values separated by a comma)\n"
    out += ("  -M, --read-mesh-from-file  Read mesh from file instead or generate it inside the program\n"
            "  -m, --mesh-size X,Y       Set mesh size (two integers separated by a comma)\n"
            "  -r, --reynolds N         Set Reynolds number (floating point value)\n"
            "  -s, --solver N            Select solver (valid values: 0: GMRES, 1: FGMRES, 2: Bicgstab)\n"
            "  -t, --tolerance D         Set tolerance (floating point value)\n"
            "  -p, --preconditioner N    Select preconditioner (valid values: 0: blockDiagonal, 1: blockTriangular, 2: aSIMPLE)\n"
            "  -h, --help                Display this help message\n")
    sys.stdout.write(out)


def parse(argv, unsteady: bool):
    cfg = dict(read_mesh=False, Re=100.0, mx=100, my=100, solver=1, tol=1e-6, prec=0, T=1.0, dt=0.01)
    short = ("T:" if unsteady else "") + "M:m:r:s:t:p:h"
    longs = (["timespan-step="] if unsteady else []) + ["read-mesh-from-file", "mesh-size=", "reynolds=", "solver=",
                                                          "tolerance=", "preconditioner=", "help"]
    try:
        opts, _ = getopt.getopt(argv, short, longs)
    except getopt.GetoptError:
        print_help(unsteady)
        return None, 1
    for o, a in opts:
        if o in ("-T", "--timespan-step"):
            if "," not in a:
                sys.stderr.write("Error: timespan-step requires two values separated by comma\n")
                return None, 1
            cfg["T"], cfg["dt"] = (float(v) for v in a.split(",", 1))
        elif o in ("-M", "--read-mesh-from-file"):
            cfg["read_mesh"] = a if (o == "-M" and a) else True    # -M swallows the next token: taken as the mesh file
        elif o in ("-m", "--mesh-size"):
            if "," not in a:
                sys.stderr.write("Error: mesh-size requires two values separated by comma\n")
                return None, 1
            cfg["mx"], cfg["my"] = (int(v) for v in a.split(",", 1))
        elif o in ("-r", "--reynolds"):
            cfg["Re"] = float(a)
        elif o in ("-s", "--solver"):
            cfg["solver"] = int(a)
        elif o in ("-t", "--tolerance"):
            cfg["tol"] = float(a)
        elif o in ("-p", "--preconditioner"):
            cfg["prec"] = int(a)
        elif o in ("-h", "--help"):
            print_help(unsteady)
            return None, 0
    if cfg["tol"] <= 0 or (unsteady and (cfg["dt"] <= 0 or cfg["T"] <= 0)):
        sys.stderr.write("Error: time_step, time_span, and tolerance must be positive\n" if unsteady
                         else "Error: tolerance must be positive\n")
        return None, 1
    return cfg, 0


def echo(cfg, unsteady: bool):
    p = sys.stdout.write
    p("--------- CONFIGURATION PARAMETERS --------- \n")
    if unsteady:
        p(f"Time span: {cfg['T']:g}\nTime step: {cfg['dt']:g}\n")
    p(f"Mesh size: {cfg['mx']}x{cfg['my']}\nReynolds number: {cfg['Re']:g}\n")
    p("Solver type: " + (SOLVERS.get(cfg["solver"], "") + "\n" if cfg["solver"] in SOLVERS else ""))
    p(f"Tolerance: {cfg['tol']:g}\n")
    p("Preconditioner: " + (PRECS.get(cfg["prec"], "") + "\n" if cfg["prec"] in PRECS else ""))
    p("-----------------------------------------------\n")


def _levels(first, step, target):
    out, re = [], first
    while re <= target:
        out.append(re)
        re += step
    return out or [first]


def _report(backend, nx, ny, nu, inlet_u, name, counter, n_digits):
    """What the reference's main() / time loop does with the solution (testStationary.cpp:133-136,
    NSSolver.cpp:830-833): VTU record, lift and drag forces over boundary id 10, their coefficients."""
    import os

    from . import postprocess as PP
    u, p = backend.solution()
    print("===============================================")
    PP.write_vtu(os.environ.get("NSK_OUTPUT_DIR", "./"), name, counter, nx, ny, u, p, n_digits=n_digits)
    print("Output written to output-stokes")          # the reference prints this name in both drivers
    print("===============================================")
    print("===============================================\nComputing lift and drag forces")
    drag, lift = PP.lift_drag(nx, ny, u, p, nu)
    cd, cl = PP.coefficients(drag, lift, inlet_u)
    print(f"===============================================\nLift coefficient: {cl:g}")
    print(f"===============================================\nDrag coefficient: {cd:g}")


def run_gmsh(cfg, unsteady: bool) -> int:
    """`-M FILE`: P2/P1 on a gmsh triangle mesh (NSSolverStationary.cpp:144-206).  The reference reads a hard-coded
    path (testStationary.cpp:127) and its getopt string lets -M swallow the next token: here that token names the file."""
    import numpy as np

    from . import gmsh as G
    from . import newton as N
    from . import simplex as SX
    from . import solver as S
    if cfg["prec"] not in PRECS:
        raise ValueError("Invalid preconditioner type. Use 0: blockDiagonal, 1: blockTriangular, 2: aSIMPLE.")
    # no file after -M: the reference's hard-coded path (testStationary.cpp:127), or $NSK_MESH_FILE
    path = cfg["read_mesh"] if isinstance(cfg["read_mesh"], str) else os.environ.get(
        "NSK_MESH_FILE", "/home/users/gdaneri/navier_stokes_solver/lab_new/mesh/new_mesh.msh")
    if not os.path.isfile(path):
        raise SystemExit(f"Error: mesh file '{path}' not found (give it after -M, or set NSK_MESH_FILE)")
    print("Initializing the mesh")
    print(f"Mesh file name = {path}")
    space = SX.build_space(G.read_msh(path))
    print(f"  Number of elements = {len(space.cell_u)}")
    print("Initializing the finite element space\n  Velocity degree:           = 2\n  Pressure degree:           = 1\n"
          "  DoFs per cell              = 15\n  Quadrature points per cell = 7\n  Quadrature points per face = 3")
    print("-----------------------------------------------\nInitializing the DoF handler\n  Number of DoFs: ")
    print(f"    velocity = {space.n_u}\n    pressure = {space.n_p}\n    total    = {space.n_u + space.n_p}")
    print("-----------------------------------------------")
    from . import postprocess as PP
    nranks = int(os.environ.get("NSK_RANKS", "1"))    # `mpirun -n N`: N rank threads (in-process transport), see below
    ls = None
    if nranks <= 1 or unsteady:
        ls = S.LinearSolver()
        ls.set_option(S.OPT_TRI_ORDERING, S.ORDER_MULTICOLOR)
        _env_options(ls, S)

    def report(name, nu, inlet_u):
        u, p = backend.solution()
        out_dir = os.environ.get("NSK_OUTPUT_DIR", "./")
        if nranks > 1 and not unsteady:     # every rank writes its own piece, rank 0 the record naming them (.cpp:793-796)
            stem = name[:-4]
            pieces = [f"{stem}.{r}.vtu" for r in range(nranks)]
            for r in range(nranks):
                SX.write_vtu(os.path.join(out_dir, pieces[r]), space, u, p, cells=np.nonzero(backend.layout.cell_rank == r)[0])
            SX.write_pvtu(os.path.join(out_dir, stem + ".pvtu"), pieces)
        else:
            SX.write_vtu(os.path.join(out_dir, name), space, u, p)
        # (the reference prints this fixed name in both drivers, whatever file DataOut wrote)
        print("===============================================\nOutput written to output-stokes")
        print("===============================================\n===============================================\nComputing lift and drag forces")
        if nranks > 1 and not unsteady:     # each rank its own share of the obstacle, summed (Utilities::MPI::sum, .cpp:895-896)
            drag, lift, _ = backend.lift_drag(nu)
        else:
            drag, lift = SX.lift_drag(space, u, p, nu)
        cd, cl = PP.coefficients(drag, lift, inlet_u)
        print(f"===============================================\nLift coefficient: {cl:g}")
        print(f"===============================================\nDrag coefficient: {cd:g}")

    if unsteady:           # NSSolver: the time loop over the same cells (mass and solution_old terms in nsk_assemble)
        first = SX.assemble(space, 1.0, mode=0, inlet_bc=1, U=0.3)
        first.simplex = SX.device_handoff(space, first)
        backend = N.DeviceBackend(ls, first, cfg["solver"], cfg["prec"], cfg["tol"], max_iter=100000, inv_dt=1.0 / cfg["dt"])
        t0 = time.time()
        try:
            nu_last = 1.0 / max(_levels(1.0, 10.0, cfg["Re"]))
            N.time_loop(backend, cfg["T"], cfg["dt"], cfg["Re"],
                        after_step=lambda step: report(f"output_{step:03d}.vtu", nu_last, 0.3))
        finally:
            print(f"[nsk] {backend.assemblies} assemblies (device, P2/P1), {backend.total_linear_iterations} outer iterations of "
                  f"solve_system() on the GPU, {time.time() - t0:.3f} s in the time loop")
            ls.close()
        return 0
    host_assembly = bool(os.environ.get("NSK_HOST_ASSEMBLY")) or nranks > 1
    if nranks > 1:
        # Several ranks (the reference: mpirun -n N, METIS partition, NSSolverStationary.cpp:160-166): coordinate bisection of
        # the cells, one handle per rank.  Here the ranks are threads of this process joined by the in-process transport,
        # one GPU each when the process sees several, else sharing the one there is; assembly on the host.
        import torch
        ndev = max(1, torch.cuda.device_count())
        devices = [r % ndev for r in range(nranks)]
        uid = S.local_group_id(nranks, on_stream=(ndev == 1))
        backend = N.MultiRankSimplexBackend(space, nranks, cfg["solver"], cfg["prec"], cfg["tol"], uid, devices=devices)
        print(f"  Number of ranks            = {nranks} (cells per rank: "
              f"{' '.join(str(int(c)) for c in np.bincount(backend.layout.cell_rank, minlength=nranks))})")
    elif host_assembly:      # the hand-off producer on the host for every assembly (the yardstick of the device assembly)
        backend = N.SimplexBackend(ls, space, cfg["solver"], cfg["prec"], cfg["tol"])
    else:                  # first hand-off (pattern, constant blocks) from the host, then nsk_assemble on the P2/P1 cells
        first = SX.assemble(space, 0.1, mode=0, inlet_bc=1, U=0.1)
        first.simplex = SX.device_handoff(space, first)
        backend = N.DeviceBackend(ls, first, cfg["solver"], cfg["prec"], cfg["tol"])
    t0 = time.time()
    try:
        N.solve_newton(backend, cfg["Re"])
        report("output-stokes_0.vtu", 1.0 / max(_levels(10.0, 20.0, cfg["Re"])), 1.0)
    finally:
        dt = time.time() - t0
        its = backend.total_linear_iterations
        print(f"[nsk] {backend.assemblies} assemblies ({'host' if host_assembly else 'device'}, P2/P1), {its} outer iterations "
              f"of solve_system() on the GPU, {dt:.3f} s in solve_newton")
        if ls is not None:
            ls.close()
        else:
            backend.close()
    return 0


def _env_options(ls, S):
    """Opt-in switches of the drivers.  NSK_SCHUR_SIGN=-1: aSIMPLE with the Schur approximation negated — a labelled
    deviation from the reference (include/nsk.h); unset: the reference's S = B~ D^-1 B~^T."""
    v = os.environ.get("NSK_SCHUR_SIGN")
    if v is not None:
        ls.set_option(S.OPT_SCHUR_SIGN, float(v))
        if float(v) < 0 and ls.rank == 0:
            print("[nsk] NSK_SCHUR_SIGN=-1: aSIMPLE's Schur approximation negated (deviation from the reference)")


class _ThreadGroup:
    """Control plane of `NSK_RANKS=N`: the ranks are threads of this process."""

    def __init__(self, n):
        import threading
        self.n, self.meet, self.slots = n, threading.Barrier(n), [None] * n

    def allgather(self, r, obj):
        self.slots[r] = obj
        self.meet.wait()
        out = list(self.slots)
        self.meet.wait()
        return out

    def abort(self):
        self.meet.abort()


class _ProcessGroup:
    """Control plane of one process per rank (torch.distributed.run): gloo; the data path is RCCL inside the library."""

    def __init__(self, dist, n):
        self.dist, self.n = dist, n

    def allgather(self, r, obj):
        out = [None] * self.n
        self.dist.all_gather_object(out, obj)
        return out

    def abort(self):
        pass


def _rank_main(cfg, unsteady, r, nranks, grp, make_handle, say):
    """One rank of `mpirun -n N StationaryNSSolver / NSSolver -m X,Y`: its x-strip of the mesh, its handle, device-resident
    state and the reference's driver loop; `grp` is the control plane (ghost lists, VTU pieces, lift / drag sums)."""
    import numpy as np

    from . import newton as N
    from . import partition as PT
    from . import postprocess as PP
    from . import problem as P
    from . import solver as S
    nx, ny = cfg["mx"], cfg["my"]
    U = 0.3 if unsteady else 0.1
    nu0 = 1.0 if unsteady else 0.1
    first = P.generate(nx, ny, nu=nu0, mode=0, state=0, inlet_bc=1, U=U, nranks=nranks, rank=r)
    ghosts = grp.allgather(r, (first.ghost_u, first.ghost_p))
    ur, prg = first.u_ranges, first.p_ranges
    plan = {S.SPACE_U: PT.build_halo_plan(r, ur, [g[0] for g in ghosts]),
            S.SPACE_P: PT.build_halo_plan(r, prg, [g[1] for g in ghosts])}
    n_u, n_p = int(ur[-1]), int(prg[-1])
    ls = make_handle()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, S.ORDER_MULTICOLOR)
        _env_options(ls, S)
        backend = N.DeviceBackend(ls, first, cfg["solver"], cfg["prec"], cfg["tol"],
                                  max_iter=100000 if unsteady else 20000, inv_dt=1.0 / cfg["dt"] if unsteady else 0.0,
                                  plan=plan)

        def report(nu, inlet_u, name, counter, n_digits):
            # the ranks' owned pieces side by side are the global vectors (x-strips own contiguous ranges)
            pieces = grp.allgather(r, backend.solution())
            u, p = np.concatenate([q[0] for q in pieces]), np.concatenate([q[1] for q in pieces])
            say("===============================================")
            PP.write_vtu(os.environ.get("NSK_OUTPUT_DIR", "./"), name, counter, nx, ny, u, p, n_digits=n_digits, rank=r,
                         nranks=nranks)
            say("Output written to output-stokes")
            say("===============================================")
            say("===============================================\nComputing lift and drag forces")
            forces = grp.allgather(r, PP.lift_drag(nx, ny, u, p, nu, rank=r, nranks=nranks))   # (Utilities::MPI::sum)
            drag, lift = sum(f[0] for f in forces), sum(f[1] for f in forces)
            cd, cl = PP.coefficients(drag, lift, inlet_u)
            say(f"===============================================\nLift coefficient: {cl:g}")
            say(f"===============================================\nDrag coefficient: {cd:g}")

        t0 = time.time()
        if unsteady:
            nu_last = 1.0 / max(_levels(1.0, 10.0, cfg["Re"]))
            N.time_loop(backend, cfg["T"], cfg["dt"], cfg["Re"], log=say,
                        after_step=lambda step: report(nu_last, 0.3, "output", step, 3))
        else:
            N.solve_newton(backend, cfg["Re"], log=say)
            report(1.0 / max(_levels(10.0, 20.0, cfg["Re"])), 1.0, "output-stokes", 0, None)
        dt = time.time() - t0
        its = backend.total_linear_iterations
        say(f"[nsk] {nranks} ranks, {backend.assemblies} assemblies, {its} outer iterations of solve_system(), {dt:.3f} s in "
            f"{'the time loop' if unsteady else 'solve_newton'} -> {(n_u + n_p) * its / max(dt, 1e-12):.4g} DoF*iters/s")
    finally:
        ls.close()


def _say_ranks(nx, nranks):
    from . import postprocess as PP
    cols = PP.cell_columns(nx, nranks)
    print(f"  Number of ranks            = {nranks} (cell columns per rank: "
          f"{' '.join(str(b - a) for a, b in zip(cols[:-1], cols[1:]))})")


def run_ranks(cfg, unsteady: bool, nranks: int) -> int:
    """`mpirun -n N StationaryNSSolver / NSSolver -m X,Y` (NSSolverStationary.cpp:226-242: DoFs distributed by mesh
    partition) with the ranks as THREADS of this process joined by the in-process transport (one GPU each when the
    process sees several, else sharing the one there is): what `NSK_RANKS=N` selects — the way a one-GPU box runs the
    multi-rank drivers.  `run_processes` is the same driver with one process per rank over RCCL.  Every rank writes its
    VTU piece, rank 0 the .pvtu record; lift / drag are summed over the ranks' shares (Utilities::MPI::sum, .cpp:895-896)."""
    import threading

    import torch

    from . import solver as S
    if nranks > cfg["mx"]:
        raise ValueError("more ranks than cell columns")
    ndev = max(1, torch.cuda.device_count())
    uid = S.local_group_id(nranks, on_stream=(ndev == 1))
    grp = _ThreadGroup(nranks)
    errs = []

    def rank_main(r):
        try:
            _rank_main(cfg, unsteady, r, nranks, grp, lambda: S.LinearSolver(r, nranks, r % ndev, uid),
                       print if r == 0 else (lambda *_a, **_k: None))
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            # a rank that fails on its own must not leave the others waiting: the Python rendezvous AND the library's
            # collectives (the peers may be inside nsk_solve's all-reduce) are taken down; the peers get error -25
            grp.abort()
            S.abort_local_group(uid)

    _say_ranks(cfg["mx"], nranks)
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(nranks)]
    [t.start() for t in th]
    while any(t.is_alive() for t in th) and not errs:
        [t.join(0.2) for t in th]
    if errs:
        # the aborts above end every wait of the peers; should one still be stuck (in a HIP call, say) it is left behind as
        # a daemon thread after a deadline rather than hanging the process
        deadline = time.time() + float(os.environ.get("NSK_RANK_JOIN_TIMEOUT", "60"))
        [t.join(max(0.0, deadline - time.time())) for t in th]
        stuck = [r for r, t in enumerate(th) if t.is_alive()]
        raise RuntimeError(f"rank failures: {sorted(errs)}" + (f"; ranks still blocked after the deadline: {stuck}" if stuck else ""))
    return 0


def run_processes(cfg, unsteady: bool) -> int:
    """The same drivers with ONE PROCESS PER RANK, as the reference runs under mpirun:
        python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m navier_stokes_solver_amd.cli StationaryNSSolver -m X,Y ...
    Control plane (ghost lists, VTU pieces, lift / drag sums, the RCCL id): torch.distributed over gloo, initialised
    before anything touches the GPU; data path: RCCL inside the library, one GPU per process (LOCAL_RANK)."""
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > cfg["mx"]:
        raise ValueError("more ranks than cell columns")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch

        from . import solver as S
        if local_rank >= torch.cuda.device_count():
            raise RuntimeError(f"rank {rank}: no GPU {local_rank} on this node (one process per GPU)")
        torch.cuda.set_device(local_rank)
        box = [S.get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        if rank == 0:
            _say_ranks(cfg["mx"], world)
        _rank_main(cfg, unsteady, rank, world, _ProcessGroup(dist, world),
                   lambda: S.LinearSolver(rank, world, local_rank, box[0]),
                   print if rank == 0 else (lambda *_a, **_k: None))
    finally:
        dist.destroy_process_group()
    return 0


def run(cfg, unsteady: bool) -> int:
    from . import problem as P
    from . import solver as S
    if cfg["read_mesh"]:
        return run_gmsh(cfg, unsteady)
    if cfg["prec"] not in PRECS:
        raise ValueError("Invalid preconditioner type. Use 0: blockDiagonal, 1: blockTriangular, 2: aSIMPLE.")
    nx, ny = cfg["mx"], cfg["my"]
    info = P.mesh_info(nx, ny)
    print(f"  Number of elements = {info['n_cells']}")
    print("Initializing the finite element space\n  Velocity degree:           = 3\n  Pressure degree:           = 2\n"
          "  DoFs per cell              = 41\n  Quadrature points per cell = 16\n  Quadrature points per face = 4")
    print("-----------------------------------------------\nInitializing the DoF handler\n  Number of DoFs: ")
    print(f"    velocity = {info['n_u_global']}\n    pressure = {info['n_p_global']}\n"
          f"    total    = {info['n_u_global'] + info['n_p_global']}")
    print("-----------------------------------------------")
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and "TORCHELASTIC_RUN_ID" in os.environ:
        return run_processes(cfg, unsteady)      # launched by torch.distributed.run: one process per rank
    if int(os.environ.get("NSK_RANKS", "1")) > 1:
        return run_ranks(cfg, unsteady, int(os.environ["NSK_RANKS"]))
    if not unsteady:
        # the reference's solve_newton() over device-resident state: assembly, linear solves and updates on the GPU
        from . import newton as N
        ls = S.LinearSolver()
        ls.set_option(S.OPT_TRI_ORDERING, S.ORDER_MULTICOLOR)
        _env_options(ls, S)
        first_system = P.generate(nx, ny, nu=0.1, mode=0, state=0, inlet_bc=1, U=0.1)   # first level: nu = 1/10
        backend = N.DeviceBackend(ls, first_system, cfg["solver"], cfg["prec"], cfg["tol"])
        t0 = time.time()
        try:
            N.solve_newton(backend, cfg["Re"])
            _report(backend, nx, ny, 1.0 / max(l for l in _levels(10.0, 20.0, cfg["Re"])), 1.0, "output-stokes", 0, None)
        finally:
            dt = time.time() - t0
            n = info["n_u_global"] + info["n_p_global"]
            its = backend.total_linear_iterations
            print(f"[nsk] {backend.assemblies} assemblies, {its} outer iterations of solve_system(), {dt:.3f} s in "
                  f"solve_newton -> {n * its / max(dt, 1e-12):.4g} DoF*iters/s")
            ls.close()
        return 0
    # NSSolver: the reference's time loop (NSSolver::solve(), NSSolver.cpp:799-837) with one solve_newton() per step
    from . import newton as N
    ls = S.LinearSolver()
    ls.set_option(S.OPT_TRI_ORDERING, S.ORDER_MULTICOLOR)
    _env_options(ls, S)
    first_system = P.generate(nx, ny, nu=1.0, mode=0, state=0, inlet_bc=1, U=0.3)        # first level: nu = 1/1
    backend = N.DeviceBackend(ls, first_system, cfg["solver"], cfg["prec"], cfg["tol"], max_iter=100000,
                              inv_dt=1.0 / cfg["dt"])
    t0 = time.time()
    try:
        nu_last = 1.0 / max(_levels(1.0, 10.0, cfg["Re"]))
        N.time_loop(backend, cfg["T"], cfg["dt"], cfg["Re"],
                    after_step=lambda step: _report(backend, nx, ny, nu_last, 0.3, "output", step, 3))
    finally:
        dt = time.time() - t0
        n = info["n_u_global"] + info["n_p_global"]
        its = backend.total_linear_iterations
        print(f"[nsk] {backend.assemblies} assemblies, {its} outer iterations of solve_system(), {dt:.3f} s in the time "
              f"loop -> {n * its / max(dt, 1e-12):.4g} DoF*iters/s")
        ls.close()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("StationaryNSSolver", "NSSolver"):
        sys.stderr.write("usage: python -m navier_stokes_solver_amd.cli {StationaryNSSolver|NSSolver} [options]\n")
        return 2
    unsteady = argv[0] == "NSSolver"
    cfg, rc = parse(argv[1:], unsteady)
    if cfg is None:
        return rc
    if "TORCHELASTIC_RUN_ID" in os.environ and int(os.environ.get("RANK", "0")) != 0:
        import contextlib
        with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):   # pcout: rank 0 speaks (errors go to stderr)
            echo(cfg, unsteady)
            return run(cfg, unsteady)
    echo(cfg, unsteady)
    return run(cfg, unsteady)


if __name__ == "__main__":
    sys.exit(main())
