// Layout-only weight preparation: dense re-layout of the coarse hash-grid levels.
//
// nerfstudio's torch HashEncoding hashes EVERY level, so a coarse level with (res+2)^3 <= T
// vertices is scattered over T table slots (one vertex per cache line).  tn_hashgrid_prepare copies those
// levels into dense[x][y][z] = table[hash(x,y,z)] (z fastest): same values, contiguous neighbours.
#include "tn_device.h"

namespace {

__global__ void dense_fill_kernel(const float2 *__restrict__ table, unsigned mask, int res, float4 *__restrict__ dense) {
    const long long n = (long long)res * res * res;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned z = (unsigned)(i % res);
        const unsigned y = (unsigned)((i / res) % res);
        const unsigned x = (unsigned)(i / ((long long)res * res));
        // an entry and its z-neighbour as one aligned 16-byte piece (every entry is stored twice: no unaligned pair loads)
        const float2 a = table[(x ^ (y * TN_P1) ^ (z * TN_P2)) & mask], b = table[(x ^ (y * TN_P1) ^ ((z + 1) * TN_P2)) & mask];
        dense[i] = make_float4(a.x, a.y, b.x, b.y);
    }
}

// number of leading levels whose dense form fits `max_bytes` in total; fills offsets (float2 units)
int plan_dense(const tn_hashgrid *g, long long max_bytes, long long *offsets, int *res, long long *total_elems) {
    long long off = 0;
    int nd = 0;
    for (int l = 0; l < g->num_levels; ++l) {
        const long long side = (long long)g->scalings[l] + 2;
        const long long elems = side * side * side;
        if ((off + elems) * 16 > max_bytes || side > 1024) break;
        offsets[l] = off;
        res[l] = (int)side;
        off += elems;
        nd = l + 1;
    }
    *total_elems = off;
    return nd;
}

}  // namespace

extern "C" {

size_t tn_hashgrid_prepare_bytes(const tn_hashgrid *grid, int64_t max_bytes) {
    if (!grid || tn_check_grid(*grid) != TN_OK) return 0;
    long long offsets[TN_MAX_LEVELS];
    int res[TN_MAX_LEVELS];
    long long total = 0;
    plan_dense(grid, max_bytes, offsets, res, &total);
    return (size_t)total * 16;
}

int tn_hashgrid_prepare(const tn_hashgrid *grid_in, tn_hashgrid *grid_out, void *dense_dev, size_t dense_bytes,
                        void *stream) {
    if (!grid_in || !grid_out) return TN_ERR_NULL;
    const int e = tn_check_grid(*grid_in);
    if (e) return e;
    long long offsets[TN_MAX_LEVELS] = {0};
    int res[TN_MAX_LEVELS] = {0};
    long long total = 0;
    const int nd = plan_dense(grid_in, (long long)dense_bytes, offsets, res, &total);
    tn_hashgrid out = *grid_in;
    out.dense = nullptr;
    out.num_dense_levels = 0;
    if (nd > 0) {
        if (!dense_dev) return TN_ERR_NULL;
        const unsigned mask = (1u << grid_in->log2_hashmap_size) - 1u;
        for (int l = 0; l < nd; ++l) {
            const float2 *table = reinterpret_cast<const float2 *>(grid_in->table) + ((size_t)l << grid_in->log2_hashmap_size);
            float4 *dst = reinterpret_cast<float4 *>(dense_dev) + offsets[l];
            const long long n = (long long)res[l] * res[l] * res[l];
            const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
            hipLaunchKernelGGL(dense_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, mask, res[l], dst);
            out.dense_offset[l] = offsets[l];
            out.dense_res[l] = res[l];
        }
        if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
        out.dense = reinterpret_cast<const float *>(dense_dev);
        out.num_dense_levels = nd;
    }
    *grid_out = out;
    return TN_OK;
}

}  // extern "C"
