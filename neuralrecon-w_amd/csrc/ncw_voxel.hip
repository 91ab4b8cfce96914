// Ray / sparse-voxel near-far (config 3, voxel-guided sampling): native replacement of kaolin's
// `spc_render.unbatched_raytrace(..., return_depth=True, with_exit=False)` as used by
// tools/prepare_data/generate_voxel.py:311-439 (get_near_far) and rendering/renderer.py:380-456.
//
// Occupancy is a bit-packed dense grid of side G = 2^level over the normalised cube [-1,1]^3 (x index
// slowest, like kaolin's points[:,0]) plus a brick mask (8^3 voxels per brick) for empty-space
// skipping.  One thread per ray: 3-D DDA; per ray we need the ENTRY depth of the first and of the last
// occupied voxel (with_exit=False: "far" is the entry of the last voxel, generate_voxel.py:370-372).
// kaolin's source is not part of the reference tree (unpinned fork): this follows the documented
// contract and is validated geometrically against a brute-force slab test (oracle.ray_voxel_near_far).
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

__global__ void voxel_build_kernel(const float* __restrict__ pts, int64_t n, int level, uint32_t* __restrict__ occ,
                                   uint32_t* __restrict__ brick) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int G = 1 << level;
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = (pts[i * 3 + a] + 1.0f) * (0.5f * (float)G);
        if (!(u >= 0.f) || u >= (float)G) return;  // outside the cube (or NaN)
        c[a] = (int)u;
    }
    const int64_t v = ((int64_t)c[0] * G + c[1]) * G + c[2];
    atomicOr(&occ[v >> 5], 1u << (v & 31));
    const int Gb = G >> 3 > 0 ? G >> 3 : 1;
    const int64_t b = ((int64_t)(c[0] >> 3) * Gb + (c[1] >> 3)) * Gb + (c[2] >> 3);
    atomicOr(&brick[b >> 5], 1u << (b & 31));
}

// The 3-D DDA both ray kernels share: walks the level-`level` voxels a ray crosses inside the cube, in depth order, and calls
// hit(t_entry, t_exit, linear voxel index) for every OCCUPIED one.  u = origin in grid coordinates, du = direction per unit depth.
template <class F>
__device__ __forceinline__ void dda_walk(const float (&u)[3], const float (&du)[3], int G, const uint32_t* __restrict__ occ,
                                         const uint32_t* __restrict__ brick, F&& hit) {
    const int Gb = G >> 3 > 0 ? G >> 3 : 1;
    // cube entry / exit
    float t0 = -3.0e38f, t1 = 3.0e38f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ta = (0.f - u[a]) / du[a], tb = ((float)G - u[a]) / du[a];
        t0 = fmaxf(t0, fminf(ta, tb));
        t1 = fminf(t1, fmaxf(ta, tb));
    }
    if (!(t1 >= fmaxf(t0, 0.f))) return;
    float t_entry = fmaxf(t0, 0.f);
    // The exit depth of the current voxel along axis a is computed FROM THE VOXEL INDEX at every step, (boundary - u) / du, not by
    // accumulating tmax += 1 / |du|: at level 10 a ray crosses up to 3072 voxels and the accumulated rounding (up to ~0.1 voxel at the
    // far side of the cube) let the walk visit voxels the ray does not touch (tests/test_gpu_voxel.py, level 10).
    int idx[3], step[3];
    float tmax[3], inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float pos = u[a] + du[a] * t_entry;
        int i = (int)floorf(pos);
        // a ray entering through a face sits exactly on the boundary: step into the cube
        if (du[a] > 0.f) i = min(max(i, 0), G - 1);
        else i = min(max((int)ceilf(pos) - 1, 0), G - 1);
        idx[a] = i;
        step[a] = du[a] > 0.f ? 1 : -1;
        inv[a] = 1.0f / du[a];
        tmax[a] = ((float)(i + (du[a] > 0.f ? 1 : 0)) - u[a]) * inv[a];
    }
    int cur_brick = -1;
    bool brick_on = false;
    for (int it = 0; it < 3 * G + 3; ++it) {
        int ax = 0;
        if (tmax[1] < tmax[ax]) ax = 1;
        if (tmax[2] < tmax[ax]) ax = 2;
        const int b = ((idx[0] >> 3) * Gb + (idx[1] >> 3)) * Gb + (idx[2] >> 3);
        if (b != cur_brick) {
            cur_brick = b;
            brick_on = (brick[b >> 5] >> (b & 31)) & 1u;
        }
        if (brick_on) {
            const int64_t v = ((int64_t)idx[0] * G + idx[1]) * G + idx[2];
            if ((occ[v >> 5] >> (v & 31)) & 1u) hit(t_entry, tmax[ax], (int)v);
        }
        // advance to the next voxel along the ray
        t_entry = tmax[ax];
        if (ax == 0) { idx[0] += step[0]; tmax[0] = ((float)(idx[0] + (step[0] > 0 ? 1 : 0)) - u[0]) * inv[0]; }
        else if (ax == 1) { idx[1] += step[1]; tmax[1] = ((float)(idx[1] + (step[1] > 0 ? 1 : 0)) - u[1]) * inv[1]; }
        else { idx[2] += step[2]; tmax[2] = ((float)(idx[2] + (step[2] > 0 ? 1 : 0)) - u[2]) * inv[2]; }
        if (idx[0] < 0 || idx[0] >= G || idx[1] < 0 || idx[1] >= G || idx[2] < 0 || idx[2] >= G) break;
    }
}

__global__ void ray_voxel_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int R,
                                 float ox, float oy, float oz, float scale, int level,
                                 const uint32_t* __restrict__ occ, const uint32_t* __restrict__ brick,
                                 float* __restrict__ near_out, float* __restrict__ far_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int G = 1 << level;
    const float half = 0.5f * (float)G;
    const float org[3] = {ox, oy, oz};
    float u[3], du[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = rays_d[r * 3 + a] + 1e-7f;            // generate_voxel.py:332
        const float o = (rays_o[r * 3 + a] + 1e-7f - org[a]) / scale;  // :333, :345
        u[a] = (o + 1.0f) * half;   // grid coordinates
        du[a] = d * half;           // per unit of depth (depth is along the un-normalised direction)
    }
    float near = 0.f, far = 0.f;
    bool found = false;
    dda_walk(u, du, G, occ, brick, [&](float t_in, float, int) {
        if (!found) { near = t_in; found = true; }
        far = t_in;
    });
    const bool valid = found && near > 1e-4f;  // generate_voxel.py:397
    near_out[r] = valid ? near * scale : 0.f;  // :436-439
    far_out[r] = valid ? far * scale : 0.f;
}

// kaolin.render.spc.unbatched_raytrace's contract (generate_voxel.py:358-368 is its one call site): EVERY (ray, occupied voxel)
// intersection -- a "nugget" -- ordered by ray, then by depth.  Two passes of the same walk: offsets == NULL counts the
// nuggets of each ray, otherwise ray r writes its nuggets from offsets[r] on.  Origins are already normalised to the cube
// [-1, 1]^3 and nothing is added to them (the caller's get_near_far does that, :332-345).
__global__ void ray_voxel_trace_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int R, int level,
                                       const uint32_t* __restrict__ occ, const uint32_t* __restrict__ brick,
                                       const int32_t* __restrict__ offsets, int32_t* __restrict__ counts,
                                       int32_t* __restrict__ nug_ray, int32_t* __restrict__ nug_voxel,
                                       float* __restrict__ nug_depth) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int G = 1 << level;
    const float half = 0.5f * (float)G;
    float u[3], du[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        u[a] = (rays_o[r * 3 + a] + 1.0f) * half;
        du[a] = rays_d[r * 3 + a] * half;
    }
    if (!offsets) {
        int n = 0;
        dda_walk(u, du, G, occ, brick, [&](float, float, int) { ++n; });
        counts[r] = n;
    } else {
        int64_t at = offsets[r];
        dda_walk(u, du, G, occ, brick, [&](float t_in, float t_out, int v) {
            nug_ray[at] = r;
            nug_voxel[at] = v;
            nug_depth[2 * at] = t_in;
            nug_depth[2 * at + 1] = t_out;
            ++at;
        });
    }
}

extern "C" int ncw_voxel_build(const float* pts_normalised, int64_t n, int level, uint32_t* occ, uint32_t* brick,
                               void* stream) {
    if (n <= 0) return 0;
    if (level < 3 || level > 10) return NCW_E_BADARG;
    hipLaunchKernelGGL(voxel_build_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pts_normalised, n, level, occ, brick);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_ray_voxel_near_far(const float* rays_o_sfm, const float* rays_d, int R, const float* scene_origin_host,
                                      float scale, int level, const uint32_t* occ, const uint32_t* brick, float* near_sfm,
                                      float* far_sfm, void* stream) {
    if (R <= 0) return 0;
    if (level < 3 || level > 10 || !scene_origin_host) return NCW_E_BADARG;
    hipLaunchKernelGGL(ray_voxel_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, rays_o_sfm, rays_d, R,
                       scene_origin_host[0], scene_origin_host[1], scene_origin_host[2], scale, level, occ, brick, near_sfm,
                       far_sfm);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_ray_voxel_trace(const float* rays_o_norm, const float* rays_d, int R, int level, const uint32_t* occ,
                                   const uint32_t* brick, const int32_t* offsets, int32_t* counts, int32_t* nug_ray,
                                   int32_t* nug_voxel, float* nug_depth, void* stream) {
    if (R <= 0) return 0;
    if (level < 3 || level > 10 || !rays_o_norm || !rays_d || !occ || !brick) return NCW_E_BADARG;
    if (offsets ? (!nug_ray || !nug_voxel || !nug_depth) : !counts) return NCW_E_BADARG;
    hipLaunchKernelGGL(ray_voxel_trace_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, rays_o_norm, rays_d, R,
                       level, occ, brick, offsets, counts, nug_ray, nug_voxel, nug_depth);
    NCW_CHECK_LAUNCH();
    return 0;
}
