// gfx950 kernels of the FRONT END: the step before the hot path.  They turn a 640x480
// depth frame into the correspondence arrays the solvers consume -- points_c / normal_c / bearingVectors of the current
// frame, points_g / normal_g of the model -- so that those arrays are BORN in HBM (no 3 x N host matrices, no upload).
//
//   F1  frame_maps_kernel       depth image -> vertex map (pinhole back-projection, the camera of Simulator.hpp:150-162:
//                               u - cx = f X / Z), unit bearing vectors (the conversion of Simulator.hpp:215-222) and a
//                               normal map (central differences of the vertex map, oriented towards the camera)
//   F2  to_world_kernel         frame maps -> world maps under a pose (Xw = R^T (Xc - t)): the model of the next frame
//   F3  associate_kernel        projective data association of the frame against the model under a pose guess, with a
//                               distance and a normal-angle gate; writes XW XC BV NW NC aligned per pixel
//
// The reference has no such stage (its arrays come from Simulator.hpp or from the caller), so there is no reference text
// to follow: parity is against the numpy statement of the same arithmetic that the tests hold, BIT-EXACT -- every expression below is
// evaluated in fp32 in the written order with FMA contraction off, and the numpy statement performs the same IEEE operations.
// All three kernels are image-parallel and HBM/L2 streaming: one thread owns 4 consecutive pixels so that every
// xyz-interleaved map is read and written as three 16-byte accesses per lane, like the solver kernels read them.
#include "rpe_assoc.h"

namespace rpe {

#pragma clang fp contract(off)

namespace {

constexpr int kFeBlock = 256;

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fc00000); }

template <class D> __device__ __forceinline__ float depth_at(const D* __restrict__ d, int idx,
    float scale) { return (float)d[idx] * scale; }

struct Vtx { float x, y, z; bool ok; };

template <class D>
__device__ __forceinline__ Vtx vertex_at(const D* __restrict__ depth, const Camera& cam, int u, int v, float scale, float dmin,
    float dmax) {
  Vtx r;
  const float z = depth_at(depth, v * cam.width + u, scale);
  r.ok = z > dmin && z < dmax;  // false for NaN
  const float xn = ((float)u - cam.cx) / cam.fx, yn = ((float)v - cam.cy) / cam.fy;
  r.x = xn * z; r.y = yn * z; r.z = z;
  return r;
}

__device__ __forceinline__ void store4(float* __restrict__ out, int64_t g, int64_t n, const float (&v)[12]) {
  if ((g + 1) * 4 <= n) {
    float4* q = reinterpret_cast<float4*>(out) + 3 * g;
    q[0] = make_float4(v[0], v[1], v[2], v[3]);
    q[1] = make_float4(v[4], v[5], v[6], v[7]);
    q[2] = make_float4(v[8], v[9], v[10], v[11]);
  } else {
    for (int i = 0; i < 12; i++) { const int64_t idx = g * 12 + i; if (idx < 3 * n) out[idx] = v[i]; }
  }
}
__device__ __forceinline__ void load4(const float* __restrict__ in, int64_t g, int64_t n, float (&v)[12]) {
  if ((g + 1) * 4 <= n) {
    const float4* q = reinterpret_cast<const float4*>(in) + 3 * g;
    const float4 a = q[0], b = q[1], c = q[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
  } else {
    for (int i = 0; i < 12; i++) { const int64_t idx = g * 12 + i; v[i] = idx < 3 * n ? in[idx] : qnan(); }
  }
}

// ---------------------------------------------------------------------------------------------- F1
template <class D>
__global__ __launch_bounds__(kFeBlock) void frame_maps_kernel(const D* __restrict__ depth, Camera cam, float scale, float dmin, float dmax,
                             float max_jump, float* __restrict__ vmap, float* __restrict__ nmap, float* __restrict__ bmap) {
  const int64_t n = (int64_t)cam.width * cam.height;
  const int64_t g = (int64_t)blockIdx.x * kFeBlock + threadIdx.x;
  if (g * 4 >= n) return;
  float V[12], N[12], B[12];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int64_t i = g * 4 + k;
    float vx = qnan(), vy = qnan(), vz = qnan(), nx = qnan(), ny = qnan(), nz = qnan(), bx = qnan(), by = qnan(), bz = qnan();
    if (i < n) {
      const int u = (int)(i % cam.width), v = (int)(i / cam.width);
      const float xn = ((float)u - cam.cx) / cam.fx, yn = ((float)v - cam.cy) / cam.fy;
      const float s = sqrtf(xn * xn + yn * yn + 1.0f);
      bx = xn / s; by = yn / s; bz = 1.0f / s;
      const Vtx c = vertex_at(depth, cam, u, v, scale, dmin, dmax);
      if (c.ok) {
        vx = c.x; vy = c.y; vz = c.z;
        if (u > 0 && u < cam.width - 1 && v > 0 && v < cam.height - 1) {
          const Vtx l = vertex_at(depth, cam, u - 1, v, scale, dmin, dmax), r = vertex_at(depth, cam, u + 1, v, scale, dmin, dmax);
          const Vtx t = vertex_at(depth, cam, u, v - 1, scale, dmin, dmax), b = vertex_at(depth, cam, u, v + 1, scale, dmin, dmax);
          const bool smooth = l.ok && r.ok && t.ok && b.ok && fabsf(l.z - c.z) <= max_jump && fabsf(r.z - c.z) <= max_jump &&
                              fabsf(t.z - c.z) <= max_jump && fabsf(b.z - c.z) <= max_jump;
          if (smooth) {
            const float ax = r.x - l.x, ay = r.y - l.y, az = r.z - l.z;   // d/du
            const float ex = b.x - t.x, ey = b.y - t.y, ez = b.z - t.z;   // d/dv
            float cx = ey * az - ez * ay, cy = ez * ax - ex * az, cz = ex * ay - ey * ax;  // (d/dv) x (d/du): towards the camera
            const float len = sqrtf(cx * cx + cy * cy + cz * cz);
            if (len > 0.0f) {
              cx = cx / len; cy = cy / len; cz = cz / len;
              const float facing = cx * c.x + cy * c.y + cz * c.z;
              if (facing > 0.0f) { cx = -cx; cy = -cy; cz = -cz; }
              nx = cx; ny = cy; nz = cz;
            }
          }
        }
      }
    }
    V[3 * k] = vx; V[3 * k + 1] = vy; V[3 * k + 2] = vz;
    N[3 * k] = nx; N[3 * k + 1] = ny; N[3 * k + 2] = nz;
    B[3 * k] = bx; B[3 * k + 1] = by; B[3 * k + 2] = bz;
  }
  store4(vmap, g, n, V);
  store4(nmap, g, n, N);
  store4(bmap, g, n, B);
}

// ---------------------------------------------------------------------------------------------- F2
__global__ __launch_bounds__(kFeBlock) void to_world_kernel(const float* __restrict__ vmap, const float* __restrict__ nmap, int64_t n,
                             PoseF T, float* __restrict__ vw, float* __restrict__ nw) {
  const int64_t g = (int64_t)blockIdx.x * kFeBlock + threadIdx.x;
  if (g * 4 >= n) return;
  float V[12], N[12], OV[12], ON[12];
  load4(vmap, g, n, V);
  load4(nmap, g, n, N);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    to_world(T, V[3 * k], V[3 * k + 1], V[3 * k + 2], OV[3 * k], OV[3 * k + 1], OV[3 * k + 2]);
    rot_to_world(T, N[3 * k], N[3 * k + 1], N[3 * k + 2], ON[3 * k], ON[3 * k + 1], ON[3 * k + 2]);
  }
  store4(vw, g, n, OV);
  store4(nw, g, n, ON);
}

// ---------------------------------------------------------------------------------------------- F3
// T: pose guess of the frame (Xc = R Xw + t).  M: pose of the model view (world -> model camera), mcam its intrinsics.
// use_normals = 0: the normal gate is skipped and pairs do not need normals (NW / NC are still written when present).
// pose_dev != null: T is read from HBM (12 doubles, the device-resident Gauss-Newton pose) instead of the argument;
// done != null: the launch returns at once when *done is set (GnState::done of that loop).
__global__ __launch_bounds__(kFeBlock) void associate_kernel(const float* __restrict__ vmap, const float* __restrict__ nmap,
                                                             const float* __restrict__ bmap, int64_t n, const float* __restrict__ mv,
                                                             const float* __restrict__ mn, PoseF T, AssocParams P, const double* __restrict__ pose_dev,
                                                             const int* __restrict__ done, float* __restrict__ xw, float* __restrict__ xc, float* __restrict__ bv,
                                                             float* __restrict__ nw, float* __restrict__ nc, int* __restrict__ count) {
  const int64_t g = (int64_t)blockIdx.x * kFeBlock + threadIdx.x;
  int matched = 0;
  if (done != nullptr && *done) return;  // device-resident loop already converged: keep the arrays of the last iteration
  if (pose_dev != nullptr) {
#pragma unroll
    for (int k = 0; k < 9; k++) T.R[k] = (float)pose_dev[k];
#pragma unroll
    for (int k = 0; k < 3; k++) T.t[k] = (float)pose_dev[9 + k];
  }
  if (g * 4 < n) {
    float V[12], N[12], B[12], OW[12], OC[12], OB[12], ONW[12], ONC[12];
    load4(vmap, g, n, V);
    load4(nmap, g, n, N);
    load4(bmap, g, n, B);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float x = V[3 * k], y = V[3 * k + 1], z = V[3 * k + 2];
      float mx, my, mz, gx, gy, gz;
      const bool ok = associate_pixel(T, P, mv, mn, x, y, z, N[3 * k], N[3 * k + 1], N[3 * k + 2], mx, my, mz, gx, gy, gz);
      matched += ok ? 1 : 0;
      const float nan = qnan();
      OW[3 * k] = mx; OW[3 * k + 1] = my; OW[3 * k + 2] = mz;
      ONW[3 * k] = gx; ONW[3 * k + 1] = gy; ONW[3 * k + 2] = gz;
      OC[3 * k] = ok ? x : nan; OC[3 * k + 1] = ok ? y : nan; OC[3 * k + 2] = ok ? z : nan;
      ONC[3 * k] = ok ? N[3 * k] : nan; ONC[3 * k + 1] = ok ? N[3 * k + 1] : nan; ONC[3 * k + 2] = ok ? N[3 * k + 2] : nan;
      OB[3 * k] = ok ? B[3 * k] : nan; OB[3 * k + 1] = ok ? B[3 * k + 1] : nan; OB[3 * k + 2] = ok ? B[3 * k + 2] : nan;
    }
    store4(xw, g, n, OW);
    store4(xc, g, n, OC);
    store4(bv, g, n, OB);
    store4(nw, g, n, ONW);
    store4(nc, g, n, ONC);
  }
  if (count != nullptr) {  // integer total: one atomic per wave64
    for (int off = 32; off > 0; off >>= 1) matched += __shfl_down(matched, off, 64);
    if ((threadIdx.x & 63) == 0 && matched) atomicAdd(count, matched);
  }
}

int fe_grid(int64_t n) { return (int)((n + 4 * kFeBlock - 1) / (4 * kFeBlock)); }

}  // namespace

hipError_t launch_frame_maps(const void* d_depth, int depth_type, const Camera& cam, float scale, float dmin, float dmax, float max_jump,
                             float* vmap, float* nmap, float* bmap, hipStream_t s) {
  const int64_t n = (int64_t)cam.width * cam.height;
  if (n == 0) return hipSuccess;
  if (depth_type == 0)
    hipLaunchKernelGGL(frame_maps_kernel<unsigned short>, dim3(fe_grid(n)), dim3(kFeBlock), 0, s, (const unsigned short*)d_depth, cam,
        scale, dmin,
                       dmax, max_jump, vmap, nmap, bmap);
  else
    hipLaunchKernelGGL(frame_maps_kernel<float>, dim3(fe_grid(n)), dim3(kFeBlock), 0, s, (const float*)d_depth, cam, scale, dmin, dmax,
        max_jump,
                       vmap, nmap, bmap);
  return hipGetLastError();
}

hipError_t launch_to_world(const float* vmap, const float* nmap, int64_t n, const PoseF& T, float* vw, float* nw, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(to_world_kernel, dim3(fe_grid(n)), dim3(kFeBlock), 0, s, vmap, nmap, n, T, vw, nw);
  return hipGetLastError();
}

hipError_t launch_associate(const float* vmap, const float* nmap, const float* bmap, int64_t n, const float* mv, const float* mn,
                            const Camera& mcam, const PoseF& T, const PoseF& M, float dist_sq, float cos_thr, int use_normals,
                            const double* pose_dev, const int* done, float* xw, float* xc, float* bv, float* nw, float* nc, int* d_count,
                            hipStream_t s) {
  if (n == 0) return hipSuccess;
  AssocParams P;
  P.mcam = mcam; P.M = M; P.dist_sq = dist_sq; P.cos_thr = cos_thr; P.use_normals = use_normals;
  hipLaunchKernelGGL(associate_kernel, dim3(fe_grid(n)), dim3(kFeBlock), 0, s, vmap, nmap, bmap, n, mv, mn, T, P, pose_dev, done, xw,
      xc, bv,
                     nw, nc, d_count);
  return hipGetLastError();
}

void preload_frontend() {
  hipFuncAttributes a;
  if (hipFuncGetAttributes(&a, (const void*)to_world_kernel) != hipSuccess) (void)hipGetLastError();
}

}  // namespace rpe
