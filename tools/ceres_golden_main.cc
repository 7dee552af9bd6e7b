// Producer of tests/golden/ceres/*_ceres.{bin,json}: the REAL reference (TheiaSfM + Ceres) run on a
// reconstruction this repository exported.  It cannot be built in the development image (no Eigen /
// Ceres / glog / gflags there); build it on any machine with TheiaSfM installed:
//
//   g++ -O2 -std=c++14 ceres_golden_main.cc -o ceres_golden $(pkg-config --cflags eigen3) \
//       -I<theia>/include -I<theia>/include/theia/libraries/... -ltheia -lceres -lglog -lgflags ...
//   ./ceres_golden <name>_input.bin <name>_ceres.bin <name>_ceres.json <solver> <inner 0|1> <max_iter> <point_dof_note>
//
// It calls exactly the entry point this repository replaces:
//   theia::BundleAdjustReconstruction(options, &reconstruction)   (bundle_adjustment.cc:66-80)
// with BundleAdjustmentOptions left at their defaults (bundle_adjustment.h:78-122) except the fields
// named on the command line, and records what Ceres reports.  See tools/make_ceres_golden.md.
#include <theia/theia.h>

#include <fstream>
#include <iomanip>
#include <string>

int main(int argc, char** argv) {
  if (argc < 7) {
    std::cerr << "usage: " << argv[0] << " in.bin out.bin out.json SOLVER inner(0|1) max_iterations\n"
              << "  SOLVER: DENSE_SCHUR | SPARSE_SCHUR | ITERATIVE_SCHUR\n";
    return 2;
  }
  theia::Reconstruction reconstruction;
  if (!theia::ReadReconstruction(argv[1], &reconstruction)) return 1;
  theia::BundleAdjustmentOptions options;  // reference defaults
  const std::string solver = argv[4];
  options.linear_solver_type = solver == "DENSE_SCHUR"    ? ceres::DENSE_SCHUR
                               : solver == "SPARSE_SCHUR" ? ceres::SPARSE_SCHUR
                                                          : ceres::ITERATIVE_SCHUR;
  options.preconditioner_type = ceres::SCHUR_JACOBI;
  options.use_inner_iterations = std::stoi(argv[5]) != 0;
  options.max_num_iterations = std::stoi(argv[6]);
  options.verbose = true;  // Ceres' FullReport goes to the log: keep it next to the golden
  const theia::BundleAdjustmentSummary summary = theia::BundleAdjustReconstruction(options, &reconstruction);
  if (!theia::WriteReconstruction(reconstruction, argv[2])) return 1;
  std::ofstream js(argv[3]);
  js << std::setprecision(17) << "{\"success\": " << (summary.success ? 1 : 0)
     << ", \"initial_cost\": " << summary.initial_cost << ", \"final_cost\": " << summary.final_cost
     << ", \"setup_time_in_seconds\": " << summary.setup_time_in_seconds
     << ", \"solve_time_in_seconds\": " << summary.solve_time_in_seconds << ", \"linear_solver\": \"" << solver
     << "\", \"use_inner_iterations\": " << (options.use_inner_iterations ? 1 : 0)
     << ", \"max_num_iterations\": " << options.max_num_iterations << ", \"ceres_version\": \"" << CERES_VERSION_STRING
     << "\", \"num_threads\": " << options.num_threads << "}\n";
  return summary.success ? 0 : 1;
}
