// examples/test_main.cpp -- the reference's TestMain.cpp comparison harness (test_all, :52-326) on the drop-in headers:
// one NormalAOPoseAdapter<double> per trial, the six RANSAC solvers in the reference's order, then nl_shinji_kneip_ls
// without ('opt') and with ('dw') the simulated weights.  Defaults = Parameters.yml (total 100, outlier 0.1, noise 15 px /
// 0.05 m / 2 deg, 300 iterations, thre_2d 8, thre_3d 0.2, normal_thre 0.1, Gaussian); override with key=value arguments
// (total= outlier= noise_2d= noise_3d= noise_normal= iteration= thre_2d= thre_3d= normal_thre= test_n= noise_model= seed=).
// The OpenCV EPnP / iterative competitor column and the MATLAB boxplot emitter are out of scope; a table of medians is
// printed instead, plus 'gn' (3D-3D + 2D-3D Gauss-Newton) and 'gnf' (all three modalities fused, Huber) columns (GaussNewton.hpp, new).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include "AbsoluteOrientation.hpp"
#include "AbsoluteOrientationNormal.hpp"
#include "GaussNewton.hpp"
#include "Simulator.hpp"

typedef rpe::MatrixX<double> MatrixXd;
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
  std::map<std::string, std::string> kv = {{"total", "100"}, {"outlier", "0.1"}, {"noise_2d", "15"}, {"noise_3d", "0.05"}, {"noise_normal", "2"},
                                           {"iteration", "300"}, {"thre_2d", "8"}, {"thre_3d", "0.2"}, {"normal_thre", "0.1"}, {"test_n", "20"},
                                           {"noise_model", "Gaussian"}, {"seed", "11"}};
  for (int i = 1; i < argc; i++) {
    const char* eq = std::strchr(argv[i], '=');
    if (eq) kv[std::string(argv[i], eq - argv[i])] = std::string(eq + 1);
  }
  const int total = std::stoi(kv["total"]), iteration = std::stoi(kv["iteration"]), test_n = std::stoi(kv["test_n"]);
  const double orr = std::stod(kv["outlier"]), n2d = std::stod(kv["noise_2d"]), n3d = std::stod(kv["noise_3d"]);
  const double nnl = std::stod(kv["noise_normal"]) / 180. * M_PI, thre_2d = std::stod(kv["thre_2d"]), thre_3d = std::stod(kv["thre_3d"]);
  const double thre_nl = std::stod(kv["normal_thre"]), confidence = 0.99999, min_depth = 0.4, f = 585.;
  double max_depth = 8.;
  const std::string noise_model = kv["noise_model"];
  rpe::sim_seed(std::stoull(kv["seed"]));
  rpe::seed(std::stoull(kv["seed"]));
  try {
    const rpe::Point3<double> t = generate_random_translation_uniform<double>(5.0);
    const rpe::SO3<double> R = generate_random_rotation<double>(M_PI / 2, false);
    const char* names[10] = {"k", "s", "sk", "nk", "ns", "nsk", "opt", "dw", "gn", "gnf"};
    std::vector<double> te[10], re[10];
    for (int jj = 0; jj < test_n; jj++) {
      MatrixXd Q, P, U, M, N, all_weights(total, 3);
      if (noise_model == "Kinect") {
        max_depth = 3.;
        simulate_kinect_2d_3d_nl_correspondences<double>(R, t, total, n2d, orr, orr, nnl, orr, min_depth, max_depth, f, &Q, &M, &P, &N, &U, &all_weights);
      } else {
        simulate_2d_3d_nl_correspondences<double>(R, t, total, n2d, orr, n3d, orr, nnl, orr, min_depth, max_depth, f, noise_model != "Uniform",
                                                  &Q, &M, &P, &N, &U, &all_weights);
      }
      NormalAOPoseAdapter<double> adapter(U, P, N, Q, M);
      adapter.setFocal(f, f);
      auto record = [&](int k) { const rpe::Point3<double> e = calc_percentage_err<double>(R, t, &adapter); te[k].push_back(e[0]); re[k].push_back(e[1]); };
      int updated_iter = iteration;
      kneip_ransac<double>(adapter, thre_2d, updated_iter, confidence); record(0);
      updated_iter = iteration;
      shinji_ransac<double>(adapter, thre_3d, updated_iter, confidence); record(1);
      updated_iter = iteration;
      shinji_kneip_ransac<double>(adapter, thre_3d, thre_2d, updated_iter, confidence); record(2);
      updated_iter = iteration;
      nl_kneip_ransac<double>(adapter, thre_2d, thre_nl, updated_iter, confidence); record(3);
      updated_iter = iteration;
      nl_shinji_ransac<double>(adapter, thre_3d, thre_nl, updated_iter, confidence); record(4);
      updated_iter = iteration;
      nl_shinji_kneip_ransac<double>(adapter, thre_3d, thre_2d, thre_nl, updated_iter, confidence); record(5);
      const rpe::SO3<double> R_nsk = adapter.getRcw();
      const rpe::Point3<double> t_nsk = adapter.gettw();
      nl_shinji_kneip_ls<double>(adapter); record(6);
      adapter.setWeights(all_weights);
      nl_shinji_kneip_ls<double>(adapter); record(7);
      adapter.setRcw(R_nsk); adapter.sett(t_nsk);
      gn_refine_joint<double>(adapter); record(8);
      adapter.setRcw(R_nsk); adapter.sett(t_nsk);
      rpe::JointOptions jo; jo.robust = RPE_ROBUST_HUBER; jo.k_33 = 0.1; jo.k_23 = 0.02; jo.k_nn = 0.06;
      gn_refine_full<double>(adapter, jo); record(9);   // all three modalities, one fused pass per iteration
    }
    std::printf("%%total = %d  outlier = %g  noise_2d = %g  noise_3d = %g  noise_normal(rad) = %g  iteration = %d  test_n = %d  model = %s\n", total,
                orr, n2d, n3d, nnl, iteration, test_n, noise_model.c_str());
    std::printf("%%method  median t_e%%  median r_e%%\n");
    for (int k = 0; k < 10; k++) std::printf("%-6s %10.4f %10.4f\n", names[k], median(te[k]), median(re[k]));
    // smoke criterion: the joint solvers and their refinements land within a few percent on the default scene
    const bool ok = median(te[5]) < 5 && median(re[5]) < 2 && median(te[6]) < 5 && median(re[6]) < 2 && median(te[8]) < 5 && median(re[8]) < 2;
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "test_main: %s\n", e.what());
    return 2;
  }
}
