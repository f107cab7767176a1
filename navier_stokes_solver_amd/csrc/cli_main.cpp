// cli_main.cpp — StationaryNSSolver / NSSolver command-line drivers over the two C ABIs
// (include/nsk_problem.h: synthetic hand-off, include/nsk.h: GPU solve path).
//
// Flag surface, defaults, help text and the configuration echo follow the reference's drivers
// (lab_new/src/testStationary.cpp:7-123, lab_new/src/test.cpp:8-146; README.md:56-67): same getopt
// string ("M:m:r:s:t:p:h", plus "T:" for the unsteady driver — so -M swallows the next token exactly
// as there), same integer codes for -s / -p.
//   StationaryNSSolver runs the reference's whole solve_newton() (NSSolverStationary.cpp:649-758: continuation
//   ladder, Stokes phase, <= 15 Newton iterations with backtracking) with assembly (nsk_assemble), linear solves
//   (nsk_solve_resident) and vector updates (nsk_state_*) resident on the GPU — SURVEY 8f rows 1 and 3.
//   NSSolver runs the reference's time loop (NSSolver::solve(), NSSolver.cpp:799-837) with one solve_newton()
//   (NSSolver.cpp:674-754) per step; the device assembly carries the mass term and the solution_old term.
//   VTU output and lift/drag are consumers and not part of it.
//
// Built twice from this file: -DNSK_UNSTEADY=0 -> StationaryNSSolver, -DNSK_UNSTEADY=1 -> NSSolver.
#include <getopt.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "../../include/nsk.h"
#include "../../include/nsk_problem.h"

#ifndef NSK_UNSTEADY
#define NSK_UNSTEADY 0
#endif

static void print_help() {
  std::cout << "Usage: ./NSSolver [options]\n\nOptions:\n";
  if (NSK_UNSTEADY)
    std::cout << "  -T, --timespan-step T,dt  Set time span and time step (two floating point values separated by a comma)\n";
  std::cout << "  -M, --read-mesh-from-file  Read mesh from file instead or generate it inside the program\n"
            << "  -m, --mesh-size X,Y       Set mesh size (two integers separated by a comma)\n"
            << "  -r, --reynolds N         Set Reynolds number (floating point value)\n"
            << "  -s, --solver N            Select solver (valid values: 0: GMRES, 1: FGMRES, 2: Bicgstab)\n"
            << "  -t, --tolerance D         Set tolerance (floating point value)\n"
            << "  -p, --preconditioner N    Select preconditioner (valid values: 0: blockDiagonal, 1: blockTriangular, 2: aSIMPLE)\n"
            << "  -h, --help                Display this help message\n";
}

static void check(nsk_handle h, int rc, const char *what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + nsk_last_error(h));
}


// solve_newton() of both reference drivers over device-resident state.
struct InletVelocity {  // NSSolverStationary.hpp:59-111
  double u = 0.1;
  const double U_m = 1.0;
  double reynolds(double nu) const { return (2.0 * u / 3.0) * 0.1 / nu; }  // get_reynolds(), .cpp:760-763, 899-903
  bool incrementVelocity(double re) {
    if (u == U_m) return true;
    u += 0.15;
    if (re == 0.0) u = 0.01;
    if (u > U_m) u = U_m;
    return false;
  }
};

struct NewtonDriver {
  nsk_handle h;
  int solver_type, preconditioner;
  double tolerance, p_out;
  double nu_mp;          // pressure_mass currently holds 1/nu_mp * M
  double inv_dt = 0.0;   // 1/delta_t for the unsteady driver
  bool stokes_signs = true;
  long total_its = 0;
  int assemblies = 0;

  double assemble(bool first, bool stokes, double nu) {  // assemble_system(global_first_iter, computing_stokes)
    if (nu != nu_mp) { check(h, nsk_scale_values(h, NSK_BLK_MP, nu_mp / nu), "nsk_scale_values"); nu_mp = nu; }
    if (stokes_signs != stokes) { check(h, nsk_scale_values(h, NSK_BLK_B, -1.0), "nsk_scale_values"); stokes_signs = stokes; }
    double nrm = 0.0;
    // the Stokes-like first assembly has no mass term either (NSSolver.cpp:381-404)
    check(h, nsk_assemble(h, stokes ? 1 : 0, nu, stokes ? 0.0 : inv_dt, p_out, first ? 1 : 0, &nrm), "nsk_assemble");
    ++assemblies;
    return nrm;
  }
  int solve_system() {
    check(h, nsk_setup_preconditioner(h, preconditioner, NSK_UNSTEADY ? NSK_VARIANT_UNSTEADY : NSK_VARIANT_STATIONARY, 0.5),
          "nsk_setup_preconditioner");
    int iters = 0;
    double res = 0.0;
    const int rc = nsk_solve_resident(h, solver_type, tolerance, NSK_UNSTEADY ? 100000 : 20000, &iters, &res);
    check(h, rc, "nsk_solve_resident");
    if (rc > 0)
      throw std::runtime_error("Iterative method reported convergence failure in step " + std::to_string(iters) +
                               ". The residual in the last step was " + std::to_string(res) + ".");
    total_its += iters;
    return iters;
  }
  void run(double target_Re) {
    const unsigned n_max_iters = 15;
    const double residual_tolerance = 1e-9;
    bool global_first_iter = true, computing_stokes = true, inlet_reached = false;
    InletVelocity inlet_velocity;
    for (double current_Re = 10.0; current_Re <= target_Re; current_Re += 20.0) {
      std::cout << "===============================================" << std::endl;
      const double nu = 1.0 / current_Re;
      inlet_reached = false;
      std::cout << "Solving for nu = " << nu << ", Re = " << inlet_velocity.reynolds(nu) << std::endl;
      while (!inlet_reached) {
        std::cout << "Solving for inlet velocity: " << inlet_velocity.u << std::endl;
        if (global_first_iter) std::cout << "Solving Stokes adding BCs" << std::endl;
        else if (computing_stokes) std::cout << "Solving Stokes without adding BCs" << std::endl;
        else std::cout << "Solving NS" << std::endl;
        unsigned n_iter = 0;
        double residual_norm = residual_tolerance + 1, prev_residual = 0.0;
        while (n_iter < n_max_iters && residual_norm > residual_tolerance) {
          if (global_first_iter) { global_first_iter = false; residual_norm = assemble(true, true, nu); }
          else residual_norm = assemble(false, computing_stokes, nu);
          prev_residual = n_iter == 0 ? residual_norm + 1 : prev_residual;
          std::printf("Newton iteration %u/%u - ||r|| = %.6e", n_iter, n_max_iters, residual_norm);
          std::fflush(stdout);
          if (residual_norm > residual_tolerance) {
            const int GMRES_iter = solve_system();
            std::cout << "   " << GMRES_iter << " solver iterations" << std::endl;
            if (GMRES_iter == 0) break;
            check(h, nsk_state_save(h), "nsk_state_save");          // evaluation_point = solution
            for (double alpha = 1; alpha > 1e-12; alpha *= 0.1) {
              check(h, nsk_state_update(h, alpha), "nsk_state_update");
              residual_norm = assemble(false, computing_stokes, nu);
              std::cout << "  Evaluating alpha=" << alpha << ", ||r||=" << residual_norm << std::endl;
              if (residual_norm < prev_residual) break;
            }
            prev_residual = residual_norm;
          } else {
            std::cout << " < tolerance" << std::endl;
            break;
          }
          ++n_iter;
        }
        inlet_reached = inlet_velocity.incrementVelocity(inlet_velocity.reynolds(nu));
        if (inlet_reached) computing_stokes = false;
      }
    }
    std::cout << "===============================================" << std::endl;
  }

  // NSSolver::solve_newton() (NSSolver.cpp:674-754), once per time step
  void run_unsteady(double target_Re, bool apply_first) {
    const unsigned n_max_iters = 10;
    const double residual_tolerance = 1e-9;
    bool first_iter = true;
    std::cout << "===============================================\nTarget Re = " << target_Re << std::endl;
    for (double current_Re = 1.0; current_Re <= target_Re; current_Re += 10.0) {
      std::cout << "===============================================" << std::endl;
      const double nu = 1.0 / current_Re;
      std::cout << "Solving for Re = " << 0.02 / nu << std::endl;   // get_reynolds(), NSSolver.cpp:756-759
      unsigned n_iter = 0;
      double residual_norm = residual_tolerance + 1, prev_residual = 0.0;
      while (n_iter < n_max_iters && residual_norm > residual_tolerance) {
        if (first_iter) { first_iter = false; residual_norm = assemble(apply_first, true, nu); }
        else residual_norm = assemble(false, false, nu);
        prev_residual = n_iter == 0 ? residual_norm + 1 : prev_residual;
        std::printf("Newton iteration %u/%u - ||r|| = %.6e", n_iter, n_max_iters, residual_norm);
        std::fflush(stdout);
        if (residual_norm > residual_tolerance) {
          const int GMRES_iter = solve_system();
          std::cout << "   " << GMRES_iter << " iterations" << std::endl;
          if (GMRES_iter == 0) break;
          check(h, nsk_state_save(h), "nsk_state_save");
          for (double alpha = 1; alpha > 1e-12; alpha *= 0.1) {
            check(h, nsk_state_update(h, alpha), "nsk_state_update");
            residual_norm = assemble(false, false, nu);
            std::cout << "  Evaluating alpha=" << alpha << ", ||r||=" << residual_norm << std::endl;
            if (residual_norm <= prev_residual) break;               // NSSolver.cpp:738
          }
          prev_residual = residual_norm;
        } else {
          std::cout << " < tolerance" << std::endl;
          break;
        }
        ++n_iter;
      }
    }
    std::cout << "===============================================" << std::endl;
  }

  // NSSolver::solve() (NSSolver.cpp:799-837) without output / lift-drag
  void time_loop(double T, double delta_t, double target_Re) {
    double time = 0.0;
    unsigned time_step = 0;
    bool apply_first = true;
    while (time < T - 0.5 * delta_t) {
      time += delta_t;
      ++time_step;
      check(h, nsk_state_save_old(h), "nsk_state_save_old");          // solution_old = solution
      std::printf("n = %3u, t = %5.6f\n", time_step, time);
      run_unsteady(target_Re, apply_first);
      apply_first = false;
      std::cout << std::endl;
    }
  }
};

int main(int argc, char *argv[]) {
  bool read_mesh_from_file = false;
  double Re = 100.0, tolerance = 1e-6, time_span = 1.0, time_step = 0.01;
  int mesh_size_x = 100, mesh_size_y = 100, solver_type = 1, preconditioner = 0;

  static struct option long_options[] = {{"timespan-step", required_argument, 0, 'T'},
                                         {"read-mesh-from-file", no_argument, 0, 'M'},
                                         {"mesh-size", required_argument, 0, 'm'},
                                         {"reynolds", required_argument, 0, 'r'},
                                         {"solver", required_argument, 0, 's'},
                                         {"tolerance", required_argument, 0, 't'},
                                         {"preconditioner", required_argument, 0, 'p'},
                                         {"help", no_argument, 0, 'h'},
                                         {0, 0, 0, 0}};
  const char *shortopts = NSK_UNSTEADY ? "T:M:m:r:s:t:p:h" : "M:m:r:s:t:p:h";
  int opt;
  while ((opt = getopt_long(argc, argv, shortopts, NSK_UNSTEADY ? long_options : long_options + 1, nullptr)) != -1) {
    switch (opt) {
      case 'T': {
        char *comma = strchr(optarg, ',');
        if (!comma) { std::cerr << "Error: timespan-step requires two values separated by comma\n"; return 1; }
        *comma = '\0';
        time_span = std::atof(optarg);
        time_step = std::atof(comma + 1);
        break;
      }
      case 'M': read_mesh_from_file = true; break;
      case 'm': {
        char *comma = strchr(optarg, ',');
        if (!comma) { std::cerr << "Error: mesh-size requires two values separated by comma\n"; return 1; }
        *comma = '\0';
        mesh_size_x = std::atoi(optarg);
        mesh_size_y = std::atoi(comma + 1);
        break;
      }
      case 'r': Re = std::atof(optarg); break;
      case 's': solver_type = std::atoi(optarg); break;
      case 't': tolerance = std::atof(optarg); break;
      case 'p': preconditioner = std::atoi(optarg); break;
      case 'h': print_help(); return 0;
      default: print_help(); return 1;
    }
  }
  if (tolerance <= 0 || (NSK_UNSTEADY && (time_step <= 0 || time_span <= 0))) {
    std::cerr << (NSK_UNSTEADY ? "Error: time_step, time_span, and tolerance must be positive\n"
                               : "Error: tolerance must be positive\n");
    return 1;
  }

  std::cout << "--------- CONFIGURATION PARAMETERS --------- \n";
  if (NSK_UNSTEADY) std::cout << "Time span: " << time_span << "\nTime step: " << time_step << "\n";
  std::cout << "Mesh size: " << mesh_size_x << "x" << mesh_size_y << "\nReynolds number: " << Re << "\nSolver type: ";
  if (solver_type == 0) std::cout << "GMRES\n";
  else if (solver_type == 1) std::cout << "FGMRES\n";
  else if (solver_type == 2) std::cout << "Bicgstab\n";
  std::cout << "Tolerance: " << tolerance << "\nPreconditioner: ";
  if (preconditioner == 0) std::cout << "blockDiagonal\n";
  else if (preconditioner == 1) std::cout << "blockTriangular\n";
  else if (preconditioner == 2) std::cout << "aSIMPLE\n";
  std::cout << "-----------------------------------------------\n";

  if (read_mesh_from_file) {
    std::cerr << "-M (gmsh P2/P1 mesh from file): host assembly lives in the Python driver — python -m navier_stokes_solver_amd.cli StationaryNSSolver -M FILE ...\n";
    return 1;
  }
  nsk_handle h = nullptr;
  nsp_mesh *mesh = nullptr;
  try {
    if (preconditioner < 0 || preconditioner > 2)
      throw std::invalid_argument("Invalid preconditioner type. Use 0: blockDiagonal, 1: blockTriangular, 2: aSIMPLE.");
    mesh = nsp_mesh_create(mesh_size_x, mesh_size_y, 1, 0);
    if (!mesh) throw std::invalid_argument("mesh size rejected");
    nsp_info info;
    nsp_mesh_info(mesh, &info);
    std::cout << "  Number of elements = " << info.n_cells << "\n"
              << "Initializing the finite element space\n  Velocity degree:           = 3\n"
              << "  Pressure degree:           = 2\n  DoFs per cell              = 41\n"
              << "  Quadrature points per cell = 16\n  Quadrature points per face = 4\n"
              << "-----------------------------------------------\nInitializing the DoF handler\n  Number of DoFs: \n"
              << "    velocity = " << info.n_u_global << "\n    pressure = " << info.n_p_global << "\n    total    = "
              << info.n_u_global + info.n_p_global << "\n-----------------------------------------------\n"
              << "===============================================\nTarget Re = " << Re << std::endl;

    h = nsk_create(0, 1, 0, nullptr);
    if (!h) throw std::runtime_error("nsk_create failed: no usable GPU (there is no CPU fallback)");
    {
      // first hand-off: pattern, the state-independent blocks at the first level's viscosity (Stokes signs), the
      // cell connectivity and the inlet values; everything after that happens on the device
      const double nu0 = NSK_UNSTEADY ? 1.0 : 0.1;   // first continuation level: Re = 1 / Re = 10
      nsp_params prm;
      std::memset(&prm, 0, sizeof(prm));
      prm.mode = 0; prm.state = 0; prm.inlet_bc = 1; prm.nu = nu0; prm.U = NSK_UNSTEADY ? 0.3 : 0.1; prm.p_out = 1.0;
      if (nsp_assemble(mesh, &prm) != 0) throw std::runtime_error("nsp_assemble failed");
      const int n_u = (int)nsp_block_rows(mesh, NSP_BLK_F), n_p = (int)nsp_block_rows(mesh, NSP_BLK_B);
      check(h, nsk_set_partition(h, NSK_SPACE_U, 0, n_u, 0, nullptr), "nsk_set_partition");
      check(h, nsk_set_partition(h, NSK_SPACE_P, 0, n_p, 0, nullptr), "nsk_set_partition");
      {   // ordering hint for the triangular factors (DoFTools::map_dofs_to_support_points on the reference's side)
        std::vector<double> xy(2 * (size_t)std::max(n_u, n_p));
        nsp_support_points(mesh, 0, xy.data());
        check(h, nsk_set_support_points(h, NSK_SPACE_U, xy.data()), "nsk_set_support_points");
        nsp_support_points(mesh, 1, xy.data());
        check(h, nsk_set_support_points(h, NSK_SPACE_P, xy.data()), "nsk_set_support_points");
      }
      // NSK_TRI_ORDERING=0: the caller's (lattice) order in the triangular factors — what one MPI rank of the reference
      // factorises, O(nx + ny) dependent levels; default 1: the library's multicolour ordering
      if (const char *e = std::getenv("NSK_TRI_ORDERING")) check(h, nsk_set_option(h, NSK_OPT_TRI_ORDERING, std::atof(e)), "nsk_set_option");
      if (const char *e = std::getenv("NSK_MASS_ORDERING")) check(h, nsk_set_option(h, NSK_OPT_MASS_ORDERING, std::atof(e)), "nsk_set_option");
      // NSK_SCHUR_SIGN=-1: aSIMPLE with the Schur approximation negated — a LABELLED DEVIATION from the reference (nsk.h,
      // DESIGN.md 5e.2); unset / +1: the reference's S = B~ D^-1 B~^T
      if (const char *e = std::getenv("NSK_SCHUR_SIGN")) {
        check(h, nsk_set_option(h, NSK_OPT_SCHUR_SIGN, std::atof(e)), "nsk_set_option");
        if (std::atof(e) < 0) std::printf("[nsk] NSK_SCHUR_SIGN=-1: aSIMPLE's Schur approximation negated (deviation from the reference)\n");
      }
      const int blks[4] = {NSP_BLK_F, NSP_BLK_BT, NSP_BLK_B, NSP_BLK_MP};
      for (int b : blks)
        check(h, nsk_set_block_csr(h, b, (int)nsp_block_rows(mesh, b), (int)nsp_block_cols(mesh, b),
                                   nsp_block_rowptr(mesh, b), nsp_block_col(mesh, b), nsp_block_val(mesh, b)),
              "nsk_set_block_csr");
      double tables[944];
      nsp_cell_tables(mesh, tables);
      check(h, nsk_assembly_set_cells(h, nsp_n_cells_local(mesh), nsp_cell_u_nodes(mesh), nsp_cell_p_dofs(mesh),
                                      nsp_cell_flags(mesh), tables, nsp_cell_of_dof0(mesh)), "nsk_assembly_set_cells");
      check(h, nsk_assembly_set_dirichlet(h, nsp_dirichlet_u(mesh), nsp_x0_u(mesh)), "nsk_assembly_set_dirichlet");
      std::vector<double> zu((size_t)n_u, 0.0), zp((size_t)n_p, 0.0);
      check(h, nsk_state_set(h, zu.data(), zp.data()), "nsk_state_set");   // solution = 0
      NewtonDriver drv{h, solver_type, preconditioner, tolerance, 1.0, nu0, NSK_UNSTEADY ? 1.0 / time_step : 0.0};
      const auto t0 = std::chrono::steady_clock::now();
      if (NSK_UNSTEADY) drv.time_loop(time_span, time_step, Re);
      else drv.run(Re);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const double n = (double)(info.n_u_global + info.n_p_global);
      std::printf("[nsk] %d assemblies, %ld outer iterations of solve_system(), %.3f s in %s -> %.4g DoF*iters/s\n",
                  drv.assemblies, drv.total_its, dt, NSK_UNSTEADY ? "the time loop" : "solve_newton",
                  n * drv.total_its / (dt > 0 ? dt : 1e-12));
    }
  } catch (const std::exception &e) {
    std::cerr << e.what() << std::endl;
    if (h) nsk_destroy(h);
    if (mesh) nsp_mesh_destroy(mesh);
    return 2;
  }
  nsk_destroy(h);
  nsp_mesh_destroy(mesh);
  return 0;
}
