// Per-pixel projective data association, shared by the stand-alone association kernel (rpe_frontend.hip) and by the
// fused ICP kernel (rpe_icp.hip) so that both pair pixels IDENTICALLY: fp32, fixed operation order, no FMA
// contraction (the contract flag is per instruction and survives inlining).
#pragma once
#include "rpe_kernels.h"

namespace rpe {

struct AssocParams {
  Camera mcam;       // intrinsics of the model view
  PoseF M;           // world -> model camera
  float dist_sq;     // squared distance gate [m^2]
  float cos_thr;     // normal gate
  int use_normals;
};

// Xw = R^T (Xc - t), rows of R^T = columns of R
__device__ __forceinline__ void to_world(const PoseF& T, float x, float y, float z, float& ox, float& oy, float& oz) {
#pragma clang fp contract(off)
  const float dx = x - T.t[0], dy = y - T.t[1], dz = z - T.t[2];
  ox = T.R[0] * dx + T.R[3] * dy + T.R[6] * dz;
  oy = T.R[1] * dx + T.R[4] * dy + T.R[7] * dz;
  oz = T.R[2] * dx + T.R[5] * dy + T.R[8] * dz;
}
__device__ __forceinline__ void rot_to_world(const PoseF& T, float x, float y, float z, float& ox, float& oy, float& oz) {
#pragma clang fp contract(off)
  ox = T.R[0] * x + T.R[3] * y + T.R[6] * z;
  oy = T.R[1] * x + T.R[4] * y + T.R[7] * z;
  oz = T.R[2] * x + T.R[5] * y + T.R[8] * z;
}

// Frame vertex (x, y, z) with normal (nx, ny, nz) under the pose guess T (Xc = R Xw + t): move it to the world, project it
// into the model view (nearest pixel), fetch the model vertex m and normal g there, apply the gates.  m = g = 0 when unpaired.
__device__ __forceinline__ bool associate_pixel(const PoseF& T, const AssocParams& P, const float* __restrict__ mv,
    const float* __restrict__ mn,
                                                float x, float y, float z, float nx, float ny, float nz, float& mx, float& my, float& mz,
                                                float& gx, float& gy, float& gz) {
#pragma clang fp contract(off)
  bool ok = !(x != x || y != y || z != z);
  float wx, wy, wz;
  to_world(T, x, y, z, wx, wy, wz);
  const float px = P.M.R[0] * wx + P.M.R[1] * wy + P.M.R[2] * wz + P.M.t[0];
  const float py = P.M.R[3] * wx + P.M.R[4] * wy + P.M.R[5] * wz + P.M.t[1];
  const float pz = P.M.R[6] * wx + P.M.R[7] * wy + P.M.R[8] * wz + P.M.t[2];
  ok = ok && pz > 0.0f;
  const float uf = floorf(P.mcam.fx * (px / pz) + P.mcam.cx + 0.5f), vf = floorf(P.mcam.fy * (py / pz) + P.mcam.cy + 0.5f);
  ok = ok && uf >= 0.0f && uf <= (float)(P.mcam.width - 1) && vf >= 0.0f && vf <= (float)(P.mcam.height - 1);
  mx = my = mz = gx = gy = gz = 0.f;
  if (ok) {
    const int64_t j = (int64_t)(int)vf * P.mcam.width + (int)uf;
    const float ax = mv[3 * j], ay = mv[3 * j + 1], az = mv[3 * j + 2];
    const float bx = mn[3 * j], by = mn[3 * j + 1], bz = mn[3 * j + 2];
    const float ex = ax - wx, ey = ay - wy, ez = az - wz;
    ok = (ex * ex + ey * ey + ez * ez) <= P.dist_sq;  // false for a NaN model vertex
    if (P.use_normals) {
      float qx, qy, qz;
      rot_to_world(T, nx, ny, nz, qx, qy, qz);
      ok = ok && (qx * bx + qy * by + qz * bz) >= P.cos_thr;  // false if either normal is NaN
    }
    mx = ok ? ax : 0.f; my = ok ? ay : 0.f; mz = ok ? az : 0.f;
    gx = ok ? bx : 0.f; gy = ok ? by : 0.f; gz = ok ? bz : 0.f;
  }
  return ok;
}

}  // namespace rpe
