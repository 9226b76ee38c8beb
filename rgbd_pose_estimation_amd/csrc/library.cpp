// Part 1 of include/rgbd_pose_hip.h: the reference's own FFI (Library.cpp:15-82), served by the HIP backend.
#include "../../include/rgbd_pose_hip.h"
#include <cstdio>
#include <cstdlib>
#include <iostream>

namespace {
bool quiet() { const char* q = getenv("RPE_QUIET"); return q && q[0] == '1'; }
[[noreturn]] void die(const char* where) {
  // the reference aborts on failure too (SOPHUS_ENSURE -> abort, sophus/common.hpp:114-133); it has no error channel
  std::fprintf(stderr, "librgbdpose_hip: %s failed: %s\n", where, rpe_last_error());
  std::abort();
}
struct Ctx {
  rpe_context* c = nullptr;
  Ctx() { if (rpe_create(&c, 0, nullptr) != RPE_OK) die("rpe_create"); }
  ~Ctx() { rpe_destroy(c); }
};
}  // namespace

extern "C" {

void ao(float* x_w_, float* x_c_, int n_, float* R_cw_, float* t_) {
  if (!quiet()) std::cout << "ao()" << std::endl;
  Ctx ctx;
  double m[17], R[9], t[3];
  if (rpe_set_problem(ctx.c, n_, RPE_F32) || rpe_upload(ctx.c, RPE_XW, x_w_) || rpe_upload(ctx.c, RPE_XC, x_c_)) die("ao: upload");
  if (rpe_p2p_moments(ctx.c, 0, m)) die("ao: rpe_p2p_moments");   // shinji_ls2: ALL columns, no validity test (AbsoluteOrientation.hpp:331-336)
  if (rpe_pose_from_moments(m, R, t)) die("ao: rpe_pose_from_moments");
  for (int i = 0; i < 9; i++) R_cw_[i] = (float)R[i];  // row-major, as Library.cpp:35-39
  for (int i = 0; i < 3; i++) t_[i] = (float)t[i];
}

void py2c(float* array, int N) {
  for (int i = 0; i < N; i++) std::cout << array[i] << std::endl;
}

}  // extern "C"
