// BENCH / TEST PLUMBING -- not part of the hot path or of the C-ABI: a fused ray caster for the synthetic Velodyne scans
// of contour-context_amd/synth.py (boxes, cylinders, optional terrain relief; 64 x 1875 rays; misses emitted
// as far points), so that bench.py and the GPU tests can synthesise tens of thousands of 120 000-point scans directly in
// HBM in seconds.  synth.cast_scan (torch) does the same arithmetic with ~60 launches and ~250 MB intermediates per op
// (1.9 ms per scan in the sparse world, 11 ms in the dense one); this kernel walks a scan's culled object list from LDS
// (~30 us per scan).  Same geometry and conventions; the range noise / intensity come from a counter-based hash instead of
// torch's generator, so the two are statistically, not bitwise, alike (tests/test_synth_hip.py compares them without noise).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC synth_cast.hip -o libcc_synth.so
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SC_MAX_RANGE 80.0f
#define SC_SENSOR_H 1.73f
#define SC_TILE 256

struct sc_params {
  const float *dirs;      // [n_rays][3] sensor-frame unit directions
  const float *poses;     // [n_scans][4]: x, y, yaw, ground height under the sensor
  const float *boxes;     // concatenated per scan: x0 y0 z0 x1 y1 z1
  const int *box_off;     // [n_scans + 1]
  const float *cyls;      // concatenated per scan: x y r h base porous_id (id as float bits)
  const int *cyl_off;     // [n_scans + 1]
  const long long *seeds; // [n_scans]
  float *out;             // [n_scans][n_rays][4]
  int n_rays, n_scans, relief;   // relief: 0 flat ground, 1 terrain relief (march), 2 flat + rough ground (kitti world)
  float noise_sigma;
  float wave[6][5];       // terrain relief / roughness: amplitude, fx, phase x, fy, phase y
  const float *vols;      // porous cylinders, concatenated per scan: x y r z0 z1 extinction id (id as float bits) pad
  const int *vol_off;     // [n_scans + 1] or NULL
};

__device__ __forceinline__ float sc_ground(const sc_params &P, float x, float y) {
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < 6; i++) z += P.wave[i][0] * sinf(P.wave[i][1] * x + P.wave[i][2]) * sinf(P.wave[i][3] * y + P.wave[i][4]);
  return z;
}

__device__ __forceinline__ unsigned sc_hash(unsigned a, unsigned b, unsigned c) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return h;
}

// grid = (ceil(n_rays / 256), n_scans), block = 256
__global__ void __launch_bounds__(256) sc_cast(sc_params P) {
  __shared__ float obj[SC_TILE][6];
  const int scan = blockIdx.y;
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = ray < P.n_rays;
  const float px = P.poses[scan * 4 + 0], py = P.poses[scan * 4 + 1], yaw = P.poses[scan * 4 + 2], gz = P.poses[scan * 4 + 3];
  const float c = cosf(yaw), s = sinf(yaw);
  float dsx = 0.f, dsy = 0.f, dsz = -1.f;
  if (live) {
    dsx = P.dirs[ray * 3 + 0];
    dsy = P.dirs[ray * 3 + 1];
    dsz = P.dirs[ray * 3 + 2];
  }
  const float dx = c * dsx - s * dsy, dy = s * dsx + c * dsy, dz = dsz;
  const float ox = px, oy = py, oz = gz + SC_SENSOR_H;
  float tb = __builtin_inff();
  // ---- ground
  if (P.relief != 1) {
    if (dz < -1e-6f) {
      tb = -SC_SENSOR_H / dz;
      if (P.relief == 2) {  // rough ground: the height under the flat-plane hit shifts the hit along the ray (synth.cast_scan)
        const float tq = fminf(tb, 2.f * SC_MAX_RANGE);
        tb = -(SC_SENSOR_H - sc_ground(P, ox + tq * dx, oy + tq * dy)) / dz;
      }
    }
  } else if (live) {
    // 0.5 m march, first sample below the terrain, linear interpolation, two secant refinements (synth.cast_scan)
    float fprev = SC_SENSOR_H;
    for (int k = 1; k <= 160; k++) {
      const float t = 0.5f * (float)k;
      const float f = oz + dz * t - sc_ground(P, ox + dx * t, oy + dy * t);
      if (f <= 0.f) {
        float ts = (t - 0.5f) + 0.5f * fprev / fmaxf(fprev - f, 1e-6f);
        for (int it = 0; it < 2; it++) {
          const float fa = oz + ts * dz - sc_ground(P, ox + ts * dx, oy + ts * dy);
          const float e = 0.05f;
          const float fb = oz + (ts + e) * dz - sc_ground(P, ox + (ts + e) * dx, oy + (ts + e) * dy);
          float df = (fb - fa) / e;
          if (fabsf(df) < 1e-4f) df = -1e-4f;
          ts = ts - fa / df;
        }
        tb = fmaxf(ts, 0.f);
        break;
      }
      fprev = f;
    }
  }
  // ---- boxes (slab test)
  const float ix = 1.f / (fabsf(dx) < 1e-9f ? 1e-9f : dx), iy = 1.f / (fabsf(dy) < 1e-9f ? 1e-9f : dy), iz = 1.f / (fabsf(dz) < 1e-9f ? 1e-9f : dz);
  for (int b0 = P.box_off[scan]; b0 < P.box_off[scan + 1]; b0 += SC_TILE) {
    const int nb = min(SC_TILE, P.box_off[scan + 1] - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 6; i += blockDim.x) obj[i / 6][i % 6] = P.boxes[(size_t)b0 * 6 + i];
    __syncthreads();
    for (int i = 0; i < nb; i++) {
      const float t0x = (obj[i][0] - ox) * ix, t1x = (obj[i][3] - ox) * ix;
      const float t0y = (obj[i][1] - oy) * iy, t1y = (obj[i][4] - oy) * iy;
      const float t0z = (obj[i][2] - oz) * iz, t1z = (obj[i][5] - oz) * iz;
      const float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
      const float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
      if (tmax >= tmin && tmax > 0.f) tb = fminf(tb, tmin > 0.f ? tmin : tmax);
    }
  }
  // ---- cylinders
  const float a = dx * dx + dy * dy;
  for (int c0 = P.cyl_off[scan]; c0 < P.cyl_off[scan + 1]; c0 += SC_TILE) {
    const int nc = min(SC_TILE, P.cyl_off[scan + 1] - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < nc * 6; i += blockDim.x) obj[i / 6][i % 6] = P.cyls[(size_t)c0 * 6 + i];
    __syncthreads();
    for (int i = 0; i < nc; i++) {
      const float qx = ox - obj[i][0], qy = oy - obj[i][1], r = obj[i][2], h = obj[i][3], base = obj[i][4];
      const float b = 2.f * (qx * dx + qy * dy);
      const float cc = qx * qx + qy * qy - r * r;
      const float disc = b * b - 4.f * a * cc;
      const float tc = (-b - sqrtf(fmaxf(disc, 0.f))) / (2.f * a + 1e-12f);
      const float z = oz + tc * dz;
      bool ok = disc > 0.f && tc > 0.f;
      if (P.relief != 1) {
        ok = ok && z >= 0.f && z <= h;
      } else {
        ok = ok && z >= base - 1.5f && z <= base + h;
        if (ok && r > 0.45f) {  // foliage is porous: stopped with probability 0.3, decided by a hash of (ray, object)
          const long long cid = (long long)__float_as_int(obj[i][5]);
          const long long hsh = (((long long)ray * 2654435761ll + cid * 40503ll + 12345ll) >> 7) & 1023ll;
          ok = hsh < 307;
        }
      }
      if (ok) tb = fminf(tb, tc);
    }
  }
  const unsigned sd = (unsigned)(P.seeds[scan] & 0xFFFFFFFFll) ^ (unsigned)(P.seeds[scan] >> 32);
  // ---- porous cylinders (crowns, bushes): the ray ends inside with probability 1 - exp(-extinction * chord), at a uniform
  //      depth along the chord; both draws from one hash of (ray, object, scan) -- synth._vol_hash
  if (P.vol_off) {
    const float izv = 1.f / (fabsf(dz) < 1e-9f ? 1e-9f : dz);
    for (int v0 = P.vol_off[scan]; v0 < P.vol_off[scan + 1]; v0 += SC_TILE * 6 / 8) {
      const int nv = min(SC_TILE * 6 / 8, P.vol_off[scan + 1] - v0);
      float(*vo)[8] = (float(*)[8]) & obj[0][0];
      __syncthreads();
      for (int i = threadIdx.x; i < nv * 8; i += blockDim.x) vo[i / 8][i % 8] = P.vols[(size_t)v0 * 8 + i];
      __syncthreads();
      for (int i = 0; i < nv; i++) {
        const float qx = ox - vo[i][0], qy = oy - vo[i][1], r = vo[i][2];
        const float b = 2.f * (qx * dx + qy * dy);
        const float cc = qx * qx + qy * qy - r * r;
        const float disc = b * b - 4.f * a * cc;
        if (!(disc > 0.f)) continue;
        const float sq = sqrtf(disc), ia = 1.f / (2.f * a + 1e-12f);
        const float t_in = (-b - sq) * ia, t_out = (-b + sq) * ia;
        const float tz0 = (vo[i][3] - oz) * izv, tz1 = (vo[i][4] - oz) * izv;
        const float lo = fmaxf(fmaxf(t_in, fminf(tz0, tz1)), 0.f), hi = fminf(t_out, fmaxf(tz0, tz1));
        const float ln = hi - lo;
        if (!(ln > 0.f) || lo >= tb) continue;
        const unsigned vid = (unsigned)__float_as_int(vo[i][6]);
        unsigned h = ((unsigned)ray * 0x9E3779B1u) ^ ((vid + 0x7F4A7C15u) * 0x85EBCA77u) ^ ((sd + 0x165667B1u) * 0xC2B2AE3Du);
        h ^= h >> 15;
        h *= 0x2C1B3C6Du;
        h ^= h >> 12;
        h *= 0x297A2D39u;
        h ^= h >> 15;
        const float u1 = (float)(h & 0xFFFFu) * (1.f / 65536.f), u2 = (float)(h >> 16) * (1.f / 65536.f);
        if (u1 < 1.f - expf(-vo[i][5] * ln)) tb = fminf(tb, lo + u2 * ln);
      }
    }
  }
  if (!live) return;
  const bool hit = tb < SC_MAX_RANGE;
  const unsigned h1 = sc_hash(sd, (unsigned)ray, 1u), h2 = sc_hash(sd, (unsigned)ray, 2u), h3 = sc_hash(sd, (unsigned)ray, 3u);
  const float u1 = ((float)(h1 >> 8) + 0.5f) * (1.f / 16777216.f), u2 = ((float)(h2 >> 8) + 0.5f) * (1.f / 16777216.f);
  const float nrm = sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
  const float t = hit ? tb + nrm * P.noise_sigma : 0.f;
  float4 o;
  o.x = hit ? dsx * t : 1000.f;
  o.y = hit ? dsy * t : 1000.f;
  o.z = hit ? dsz * t : 0.f;
  o.w = ((float)(h3 >> 8) + 0.5f) * (1.f / 16777216.f);
  ((float4 *)P.out)[(size_t)scan * P.n_rays + ray] = o;
}

extern "C" int sc_cast_scans(const float *d_dirs, int n_rays, int n_scans, const float *d_poses, const float *d_boxes, const int *d_box_off,
                             const float *d_cyls, const int *d_cyl_off, const long long *d_seeds, int relief, const float *h_wave /*[30] or NULL*/,
                             float noise_sigma, float *d_out, void *stream, const float *d_vols, const int *d_vol_off) {
  sc_params P;
  P.vols = d_vols;
  P.vol_off = d_vol_off;
  P.dirs = d_dirs;
  P.poses = d_poses;
  P.boxes = d_boxes;
  P.box_off = d_box_off;
  P.cyls = d_cyls;
  P.cyl_off = d_cyl_off;
  P.seeds = d_seeds;
  P.out = d_out;
  P.n_rays = n_rays;
  P.n_scans = n_scans;
  P.relief = relief;
  P.noise_sigma = noise_sigma;
  for (int i = 0; i < 30; i++) (&P.wave[0][0])[i] = (relief && h_wave) ? h_wave[i] : 0.f;  // relief 1: terrain, 2: roughness
  if (n_scans <= 0) return 0;
  hipLaunchKernelGGL(sc_cast, dim3((n_rays + 255) / 256, n_scans), dim3(256), 0, (hipStream_t)stream, P);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
