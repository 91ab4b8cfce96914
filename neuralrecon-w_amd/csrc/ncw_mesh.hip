// Iso-surface extraction on the GPU: replaces the rank-0 CPU call
// `skimage.measure.marching_cubes(sdf, level=0, mask=mask_dense)` of utils/visualization.py:114.
// skimage (Lewiner's tables) is not part of the reference tree and not installable here, so its triangulation
// cannot be pinned; this is MARCHING TETRAHEDRA on the same grid: every cube is split into the six tetrahedra
// around its main diagonal (the face diagonals of neighbouring cubes coincide, so the mesh is watertight), each
// tetrahedron contributes 0, 1 or 2 triangles, vertices are the linear zero crossings on grid edges -- the same
// vertex set rule as marching cubes (one vertex per sign-changing edge), a different (finer) triangulation.
//
// Two passes over the (Dx-1)(Dy-1)(Dz-1) cubes, one thread per cube: count triangles, then (after an exclusive
// prefix sum done by the caller) emit them.  A vertex is identified by the ordered pair of grid points of its
// edge (key = lo * n_points + hi) and is interpolated from lo to hi, so every cube sharing the edge produces
// bit-identical coordinates; the caller welds by key.
#include "../../include/neuconw_hip.h"
#include "ncw_common.h"

namespace {

// cube corners: bit0 = +x, bit1 = +y, bit2 = +z ; the six tetrahedra around the diagonal corner 0 -> corner 7
__constant__ int MT_TETS[6][4] = {{0, 1, 3, 7}, {0, 3, 2, 7}, {0, 2, 6, 7}, {0, 6, 4, 7}, {0, 4, 5, 7}, {0, 5, 1, 7}};

struct Cube {
    float v[8];
    int64_t id[8];
    int x, y, z;
};

__device__ __forceinline__ bool load_cube(const float* __restrict__ sdf, const uint8_t* __restrict__ mask, int Dx, int Dy,
                                          int Dz, int64_t c, Cube& q) {
    const int cz = Dz - 1, cy = Dy - 1;
    q.z = (int)(c % cz);
    q.y = (int)((c / cz) % cy);
    q.x = (int)(c / ((int64_t)cz * cy));
    if (mask != nullptr && !mask[((int64_t)(q.x + 1) * Dy + (q.y + 1)) * Dz + (q.z + 1)]) return false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t id = ((int64_t)(q.x + (k & 1)) * Dy + (q.y + ((k >> 1) & 1))) * Dz + (q.z + ((k >> 2) & 1));
        q.id[k] = id;
        q.v[k] = sdf[id];
    }
    return true;
}

__device__ __forceinline__ int tet_count(const Cube& q, int t, float level) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) k += q.v[MT_TETS[t][i]] < level ? 1 : 0;
    return (k == 1 || k == 3) ? 1 : (k == 2 ? 2 : 0);
}

struct Vtx {
    float p[3];
    int64_t key;
};

// zero crossing on the grid edge between cube corners a and b (ordered by grid point id: lo -> hi)
__device__ __forceinline__ Vtx edge_vertex(const Cube& q, int a, int b, float level, int64_t npts) {
    if (q.id[a] > q.id[b]) { const int t = a; a = b; b = t; }
    const float va = q.v[a], vb = q.v[b];
    const float t = (level - va) / (vb - va);
    Vtx o;
    const float pa[3] = {(float)(q.x + (a & 1)), (float)(q.y + ((a >> 1) & 1)), (float)(q.z + ((a >> 2) & 1))};
    const float pb[3] = {(float)(q.x + (b & 1)), (float)(q.y + ((b >> 1) & 1)), (float)(q.z + ((b >> 2) & 1))};
#pragma unroll
    for (int i = 0; i < 3; ++i) o.p[i] = pa[i] + t * (pb[i] - pa[i]);
    o.key = q.id[a] * npts + q.id[b];
    return o;
}

__device__ __forceinline__ void put_tri(float* __restrict__ pos, int64_t* __restrict__ key, int64_t t, Vtx a, Vtx b, Vtx c,
                                        const float (&g)[3]) {
    // orient so that the normal points towards increasing values (g: inside-centroid -> outside-centroid)
    const float u[3] = {b.p[0] - a.p[0], b.p[1] - a.p[1], b.p[2] - a.p[2]};
    const float w[3] = {c.p[0] - a.p[0], c.p[1] - a.p[1], c.p[2] - a.p[2]};
    const float n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
    if (n[0] * g[0] + n[1] * g[1] + n[2] * g[2] < 0.f) { const Vtx s = b; b = c; c = s; }
    const Vtx vs[3] = {a, b, c};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        key[t * 3 + i] = vs[i].key;
#pragma unroll
        for (int d = 0; d < 3; ++d) pos[(t * 3 + i) * 3 + d] = vs[i].p[d];
    }
}

__global__ __launch_bounds__(256) void mt_count_kernel(const float* __restrict__ sdf, const uint8_t* __restrict__ mask, int Dx,
                                                       int Dy, int Dz, float level, int64_t ncubes,
                                                       int32_t* __restrict__ counts) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncubes) return;
    Cube q;
    int n = 0;
    if (load_cube(sdf, mask, Dx, Dy, Dz, c, q)) {
#pragma unroll
        for (int t = 0; t < 6; ++t) n += tet_count(q, t, level);
    }
    counts[c] = n;
}

__global__ __launch_bounds__(256) void mt_emit_kernel(const float* __restrict__ sdf, const uint8_t* __restrict__ mask, int Dx,
                                                      int Dy, int Dz, float level, int64_t ncubes,
                                                      const int64_t* __restrict__ offsets, float* __restrict__ pos,
                                                      int64_t* __restrict__ key) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncubes) return;
    Cube q;
    if (!load_cube(sdf, mask, Dx, Dy, Dz, c, q)) return;
    int64_t out = offsets[c];
    const int64_t npts = (int64_t)Dx * Dy * Dz;
    for (int t = 0; t < 6; ++t) {
        int in[4], ni = 0, ou[4], no = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cn = MT_TETS[t][i];
            if (q.v[cn] < level) in[ni++] = cn; else ou[no++] = cn;
        }
        if (ni == 0 || ni == 4) continue;
        float g[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < ni; ++i) { g[0] -= (float)(in[i] & 1) / ni; g[1] -= (float)((in[i] >> 1) & 1) / ni; g[2] -= (float)((in[i] >> 2) & 1) / ni; }
        for (int i = 0; i < no; ++i) { g[0] += (float)(ou[i] & 1) / no; g[1] += (float)((ou[i] >> 1) & 1) / no; g[2] += (float)((ou[i] >> 2) & 1) / no; }
        if (ni == 1) {
            put_tri(pos, key, out++, edge_vertex(q, in[0], ou[0], level, npts), edge_vertex(q, in[0], ou[1], level, npts),
                    edge_vertex(q, in[0], ou[2], level, npts), g);
        } else if (ni == 3) {
            put_tri(pos, key, out++, edge_vertex(q, ou[0], in[0], level, npts), edge_vertex(q, ou[0], in[1], level, npts),
                    edge_vertex(q, ou[0], in[2], level, npts), g);
        } else {  // 2 inside, 2 outside: the quad (a c, a d, b d, b c)
            const Vtx ac = edge_vertex(q, in[0], ou[0], level, npts), ad = edge_vertex(q, in[0], ou[1], level, npts);
            const Vtx bd = edge_vertex(q, in[1], ou[1], level, npts), bc = edge_vertex(q, in[1], ou[0], level, npts);
            put_tri(pos, key, out++, ac, ad, bd, g);
            put_tri(pos, key, out++, ac, bd, bc, g);
        }
    }
}

}  // namespace

extern "C" int ncw_mt_count(const float* sdf, const uint8_t* mask, int Dx, int Dy, int Dz, float level, int32_t* counts,
                            void* stream) {
    if (!sdf || !counts || Dx < 2 || Dy < 2 || Dz < 2) return NCW_E_BADARG;
    const int64_t ncubes = (int64_t)(Dx - 1) * (Dy - 1) * (Dz - 1);
    hipLaunchKernelGGL(mt_count_kernel, dim3((unsigned)((ncubes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sdf, mask, Dx,
                       Dy, Dz, level, ncubes, counts);
    NCW_CHECK_LAUNCH();
    return 0;
}

extern "C" int ncw_mt_emit(const float* sdf, const uint8_t* mask, int Dx, int Dy, int Dz, float level, const int64_t* offsets,
                           float* tri_pos, int64_t* tri_key, void* stream) {
    if (!sdf || !offsets || !tri_pos || !tri_key || Dx < 2 || Dy < 2 || Dz < 2) return NCW_E_BADARG;
    const int64_t ncubes = (int64_t)(Dx - 1) * (Dy - 1) * (Dz - 1);
    hipLaunchKernelGGL(mt_emit_kernel, dim3((unsigned)((ncubes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sdf, mask, Dx,
                       Dy, Dz, level, ncubes, offsets, tri_pos, tri_key);
    NCW_CHECK_LAUNCH();
    return 0;
}
