// Training-step building blocks: taped forward, losses, backward.
//
// A training batch is small next to a rendered frame (4096 rays x (256 + 96 + S) samples), so the step is laid out
// stage by stage on [N, width] fp32 matrices in HBM instead of as one fused kernel: every stage is a short kernel
// with an exact adjoint, the tape (per-sample activations) is a few hundred MB that stays in L2/MALL, and the heavy
// part — the hash-table scatter-add — is atomics-bound whatever surrounds it.
//
//   tn_hash_encode_fwd / _bwd     NS HashEncoding.pytorch_fwd and its adjoint w.r.t. the table (fp32 atomics)
//   tn_linear_fwd / _bwd          torch.nn.Linear + ReLU/Sigmoid; bwd gives dx, dW (+=), db (+=)
//   tn_density_act_fwd / _bwd     average_init_density * trunc_exp(raw) * selector
//   tn_weights_bwd                adjoint of RaySamples.get_weights (reverse wave scan per ray)
//   tn_composite_bwd              adjoint of the "last_sample" compositing (RGB, thermal)
//   tn_color_input_fwd / _bwd     [SH | geo | appearance] assembly of mlp_head's input; embedding gradient
//   tn_distortion_loss            NS lossfun_distortion in O(n) per ray (sorted mid-points -> prefix sums)
//   tn_interlevel_loss            NS lossfun_outer: envelope look-ups + gradient by difference arrays
#include "tn_field_eval.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

using namespace tn;

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ void atomic_add_f32(float *addr, float v) { unsafeAtomicAdd(addr, v); }

inline int grid_for(long long work, int block, int cap) {
    long long b = (work + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// ------------------------------------------------------------------------------------------------------
// hash encoding
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
hash_encode_fwd_kernel(Grid g, tn_space space, const float *__restrict__ positions, long long n,
                       float *__restrict__ enc, float *__restrict__ selector) {
    const Space sp = make_space(space);
    const int L = g.num_levels;
    const long long total = n * L;
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < total; t += (long long)gridDim.x * kBlock) {
        const long long i = t / L;
        const int l = (int)(t - i * L);
        float px, py, pz;
        const float sel = normalize_position(sp, positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2], px, py, pz);
        const float2 f = encode_level<false, false>(g, l, px, py, pz);
        reinterpret_cast<float2 *>(enc)[t] = f;
        if (l == 0 && selector) selector[i] = sel;
    }
}

// ---- segmented wave scan on the vector unit ----------------------------------------------------------------------------
// The run merging of the scatter kernels is an inclusive segmented scan over the 64 lanes (heads flagged).  As __shfl_up
// steps it is 6 x (values + 1) ds_bpermute — the LDS pipe's work, next to the pass's own LDS adds (the owner pass without its
// scan: 463 -> 370 us per 786 k-sample call).  The same combination tree on DPP: row_shr 1 / 2 / 4 / 8 inside the rows of 16,
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 (the GCN wave64 scan), with the flag riding along: a lane
// that has no source, or is masked off, receives (flag 0, value 0) and stays as it is.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, false);
}
template <int CTRL, int ROWS, int NV>
__device__ __forceinline__ void seg_scan_step(int &f, float (&v)[NV]) {
    const int fu = dpp_i<CTRL, ROWS>(f);
    float u[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) u[e] = dpp_f<CTRL, ROWS>(v[e]);
    if (f == 0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] += u[e];
        f = fu;
    }
}
// f: 1 on the first lane of a run (lane 0 included), 0 elsewhere; v: per-lane values -> per-lane sums over the run up to the lane.
// gfx9 DPP controls (row_shr, row_bcast:15 / :31; wave_shr / wave_shl in the neighbour moves): this library targets gfx950
// only (no other target is built), and like tn_device.h's wave scans it needs every lane of the wave active at the call.
template <int NV>
__device__ __forceinline__ void seg_scan_wave(int f, float (&v)[NV]) {
    seg_scan_step<0x111, 0xf>(f, v);  // row_shr:1
    seg_scan_step<0x112, 0xf>(f, v);  // row_shr:2
    seg_scan_step<0x114, 0xf>(f, v);  // row_shr:4
    seg_scan_step<0x118, 0xf>(f, v);  // row_shr:8
    seg_scan_step<0x142, 0xa>(f, v);  // row_bcast:15 -> rows 1, 3
    seg_scan_step<0x143, 0xc>(f, v);  // row_bcast:31 -> rows 2, 3
}

// Scatter-add of d_enc into the table.  Device-scope float atomics execute memory-side on this part (one fabric
// transaction each, ~10 G/s measured whatever the address pattern), so the kernel's job is to issue fewer of them:
// a wave takes 64 CONSECUTIVE samples (neighbours along a ray) at ONE level; samples in the same grid cell form
// contiguous runs of lanes, a segmented wave scan adds each run's 8 corner contributions, and only the last lane of a
// run issues the atomics (x3.9 / x1.8 / x1.3 fewer on the 256- / 96- / 48-sample levels of the reference config).
// Coarse levels, "spread": at a level with a few thousand vertices the samples of a trained scene crowd onto the same entries
// (level 0 of the field: 4 913 vertices for 196 k samples), and same-address atomics retire one at a time in the memory-side
// atomic unit (~10 ns each): launched alone, level 0 costs 200 us against 49 for level 15 (tools/scatter_levels.py).  Such
// levels accumulate into `copies` private DENSE vertex grids instead (copy = block % copies; side^3 vertices of two floats,
// x fastest, so the x-neighbour pairing below still holds), which spread_reduce_kernel sums and hashes into d_table.
struct Spread {
    float *buf;            // [copies][per_copy][2]
    int levels, copies;    // levels [0, levels) are spread
    unsigned per_copy;     // vertices of all spread levels
    unsigned off[TN_MAX_LEVELS];
    int side[TN_MAX_LEVELS];
};

__global__ void __launch_bounds__(kBlock)
hash_encode_bwd_kernel(Grid g, tn_space space, const float *__restrict__ positions, const float *__restrict__ d_enc,
                       long long n, float *__restrict__ d_table, int level_begin, int level_end, Spread spr) {
    const Space sp = make_space(space);
    const int L = g.num_levels, LS = level_end - level_begin;  // levels [level_begin, level_end) of the L in d_enc
    const int lane = threadIdx.x & 63;
    const long long chunks = (n + 63) >> 6;
    const long long waves = chunks * LS;
    const long long wstride = (long long)gridDim.x * (kBlock / 64);
    for (long long wv = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); wv < waves; wv += wstride) {
        const long long chunk = wv / LS;
        const int l = level_begin + (int)(wv - chunk * LS);
        const long long i = chunk * 64 + lane;
        const bool live = i < n;
        const long long ic = live ? i : n - 1;
        float2 ge = reinterpret_cast<const float2 *>(d_enc)[ic * L + l];
        if (!live) ge = make_float2(0.0f, 0.0f);
        float px, py, pz;
        normalize_position(sp, positions[ic * 3], positions[ic * 3 + 1], positions[ic * 3 + 2], px, py, pz);
        // same corner / offset arithmetic as encode_level<false>
        const float s = g.scal[l];
        const float sx = mul_rn(px, s), sy = mul_rn(py, s), sz = mul_rn(pz, s);
        const float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
        const float ox = sub_rn(sx, fxf), oy = sub_rn(sy, fyf), oz = sub_rn(sz, fzf);
        const float qx = sub_rn(1.0f, ox), qy = sub_rn(1.0f, oy), qz = sub_rn(1.0f, oz);
        const int cxi = (int)ceilf(sx), cyi = (int)ceilf(sy), czi = (int)ceilf(sz);
        const int fxi = (int)fxf, fyi = (int)fyf, fzi = (int)fzf;
        // enc = ((f0 ox + f3 qx) oy + (f1 ox + f2 qx) qy) oz + ((f4 ox + f7 qx) oy + (f5 ox + f6 qx) qy) qz
        const float wgt[8] = {ox * oy * oz, ox * qy * oz, qx * qy * oz, qx * oy * oz,
                              ox * oy * qz, ox * qy * qz, qx * qy * qz, qx * oy * qz};
        float v[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            v[2 * c] = wgt[c] * ge.x;
            v[2 * c + 1] = wgt[c] * ge.y;
        }
        // runs of lanes with identical floor AND ceil corners (=> identical 8 table entries)
        // (shuffles first, all lanes active: a short-circuit && would run them under divergence)
        // (all lanes active: the DPP moves read every lane's registers)
        auto prev = [](int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false); };  // wave_shr:1
        const int ufx = prev(fxi), ufy = prev(fyi), ufz = prev(fzi), ucx = prev(cxi), ucy = prev(cyi), ucz = prev(czi);
        const bool same = (lane > 0) & (ufx == fxi) & (ufy == fyi) & (ufz == fzi) & (ucx == cxi) & (ucy == cyi) & (ucz == czi);
        const int head = same ? 0 : 1;
        const int next_head = __builtin_amdgcn_update_dpp(1, head, 0x130, 0xf, 0xf, false);  // wave_shl:1 (lane 63 keeps 1)
        const bool tail = next_head != 0;
        seg_scan_wave<16>(head, v);  // run sums on the vector unit (102 ds_bpermute as shuffle steps)
        const unsigned cx = (unsigned)cxi, cy = (unsigned)cyi, cz = (unsigned)czi;
        const unsigned fx = (unsigned)fxi, fy = (unsigned)fyi, fz = (unsigned)fzi;
        const unsigned hcy = cy * TN_P1, hfy = fy * TN_P1, hcz = cz * TN_P2, hfz = fz * TN_P2;
        const bool spread = l < spr.levels;  // wave-uniform
        float *tb = spread ? spr.buf + ((size_t)(blockIdx.x % spr.copies) * spr.per_copy + spr.off[l]) * 2
                           : d_table + ((size_t)l * g.tsize) * 2;
        const unsigned m = g.mask;
        unsigned idx[8] = {(cx ^ hcy ^ hcz) & m, (cx ^ hfy ^ hcz) & m, (fx ^ hfy ^ hcz) & m, (fx ^ hcy ^ hcz) & m,
                           (cx ^ hcy ^ hfz) & m, (cx ^ hfy ^ hfz) & m, (fx ^ hfy ^ hfz) & m, (fx ^ hcy ^ hfz) & m};
        if (spread) {
            const unsigned sd = (unsigned)spr.side[l];
            const unsigned ycz = sd * (cy + sd * cz), yfz = sd * (fy + sd * cz), ycf = sd * (cy + sd * fz), yff = sd * (fy + sd * fz);
            idx[0] = cx + ycz; idx[1] = cx + yfz; idx[2] = fx + yfz; idx[3] = fx + ycz;
            idx[4] = cx + ycf; idx[5] = cx + yff; idx[6] = fx + yff; idx[7] = fx + ycf;
        }
        // The memory side retires ~21 G atomic transactions/s on this part, one per 64-byte line an instruction touches,
        // whatever the line carries (tools/micro/atomics.hip: 4, 8, 16 ... 64 contiguous bytes of adds cost the same).  So an
        // instruction should put as much of a line as possible on its lanes: the hash's first prime is 1, hence the two
        // x-neighbours (ceil-x and floor-x corner at the same y, z) sit in the same aligned group of 8 entries = the same line
        // 7 times out of 8.  Lane L adds feature (L & 1) of x-corner ((L >> 1) & 1) of sample (L >> 2) of each quarter-wave,
        // one (y, z) combination per instruction: 16 lines per instruction instead of 32 eight-byte segments on 32 lines.
        const int feat = lane & 1, xc = (lane >> 1) & 1, tl = tail ? 1 : 0;
        constexpr int kPairs[4][2] = {{0, 3}, {1, 2}, {4, 7}, {5, 6}};  // (ceil-x, floor-x) corners sharing (y, z)
#pragma unroll
        for (int quarter = 0; quarter < 4; ++quarter) {
            const int src = quarter * 16 + (lane >> 2);
            const int st = __shfl(tl, src, 64);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int ca = kPairs[p][0], cb = kPairs[p][1];
                const float a0 = __shfl(v[2 * ca], src, 64), a1 = __shfl(v[2 * ca + 1], src, 64);
                const float b0 = __shfl(v[2 * cb], src, 64), b1 = __shfl(v[2 * cb + 1], src, 64);
                const unsigned ia = (unsigned)__shfl((int)idx[ca], src, 64), ib = (unsigned)__shfl((int)idx[cb], src, 64);
                const float val = xc ? (feat ? b1 : b0) : (feat ? a1 : a0);
                const unsigned id = xc ? ib : ia;
                if (st && val != 0.0f) atomic_add_f32(tb + (size_t)id * 2 + feat, val);
            }
        }
    }
}

// d_table[l][hash(x, y, z)] += sum over the copies of the spread grid of level l (one thread per vertex)
__global__ void __launch_bounds__(kBlock) spread_reduce_kernel(Grid g, Spread spr, float *__restrict__ d_table) {
    for (unsigned v = blockIdx.x * kBlock + threadIdx.x; v < spr.per_copy; v += gridDim.x * kBlock) {
        int l = 0;
        while (l + 1 < spr.levels && v >= spr.off[l + 1]) ++l;
        const float2 *src = reinterpret_cast<const float2 *>(spr.buf) + v;
        float sx = 0.0f, sy = 0.0f;
        for (int c = 0; c < spr.copies; ++c) {
            const float2 t = src[(size_t)c * spr.per_copy];
            sx += t.x;
            sy += t.y;
        }
        if (sx == 0.0f && sy == 0.0f) continue;
        const unsigned r = v - spr.off[l], sd = (unsigned)spr.side[l];
        const unsigned x = r % sd, y = (r / sd) % sd, z = r / (sd * sd);
        float *dst = d_table + ((size_t)l * g.tsize + ((x ^ (y * TN_P1) ^ (z * TN_P2)) & g.mask)) * 2;
        if (sx != 0.0f) atomic_add_f32(dst, sx);
        if (sy != 0.0f) atomic_add_f32(dst + 1, sy);
    }
}

// ------------------------------------------------------------------------------------------------------
// The same scatter WITHOUT global atomics: bucket, then accumulate in LDS.
// The L2 atomic path retires ~21 G line-atomics/s whatever is done on the issuing side (about one per clock and XCD:
// tools/micro/atomics*.hip), while the same lines are READ five times faster and streamed far faster still — so the
// contributions are first written out as records, bucketed by the OWNER of the table entry they belong to, and every owner
// then sums its bucket in LDS:
//   * a level's table is cut into slices of 2^14 entries (128 KB of float2 = one CU's LDS); bin = (level, slice);
//   * a record = the two x-neighbour corners of one (y, z) combination of one sample: (local index pair, 4 floats) — both
//     corners are in the same slice because x only touches the low 12 bits of the hashed index;
//   * emit pass: every bin has a fixed region of `capacity` records (1.25 x its share of an even spread + 1024: the hash
//     spreads the fine levels' entries evenly over the slices); a block reserves a contiguous run per bin with ONE global
//     atomic per bin and block, ranks come from LDS atomics; a record that no longer fits its bin's region (forced coarse
//     levels, degenerate inputs) is added to d_table with global atomics right there — always correct, no counting pass;
//   * owner pass: one 1024-thread block per bin adds its records into the LDS slice and adds the slice to d_table with plain
//     coalesced loads / stores.
// Traffic: 20 B per record written and read once (252 MB per 196 k-sample call) + the table once, all streaming.
// ------------------------------------------------------------------------------------------------------
#ifndef TN_SORT_SLICE_LOG2
#define TN_SORT_SLICE_LOG2 14
#endif
#ifndef TN_OWNER_BLOCK
#define TN_OWNER_BLOCK 1024
#endif
constexpr int kSortSliceLog2 = TN_SORT_SLICE_LOG2;  // entries per owner slice (x 8 B = 128 KB of LDS)
#ifndef TN_SORT_SAMPLES
#define TN_SORT_SAMPLES 256
#endif
constexpr int kSortSamples = TN_SORT_SAMPLES;  // samples per block of the emit pass (22 B of LDS staging per record, 4 records per sample)
constexpr int kSortMaxOwners = 1024;       // log2_hashmap_size <= 24
constexpr int kOwnerBlock = TN_OWNER_BLOCK;
constexpr int kSortMinBins = 128;
#ifndef TN_SORT_MIN_SCALING
#define TN_SORT_MIN_SCALING 200.0f
#endif
constexpr float kSortMinScaling = TN_SORT_MIN_SCALING;  // 200: from level 8 of the reference grid (scalings 154 | 212.8 | 294)

struct SortRec {
    unsigned owner[4];   // slice of the (y, z) combination
    unsigned pair[4];    // local index of the ceil-x corner | local index of the floor-x corner << 16
    float4 val[4];       // (w_ceil ge.x, w_ceil ge.y, w_floor ge.x, w_floor ge.y)
};

// the four records of one (sample, level): the corner / offset arithmetic of hash_encode_bwd_kernel
template <bool VALUES>
__device__ __forceinline__ void sort_records(const Grid &g, int l, int slice_log2, float px, float py, float pz, float2 ge,
                                             SortRec &r) {
    const float s = g.scal[l];
    const float sx = mul_rn(px, s), sy = mul_rn(py, s), sz = mul_rn(pz, s);
    const float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
    const unsigned cx = (unsigned)(int)ceilf(sx), cy = (unsigned)(int)ceilf(sy), cz = (unsigned)(int)ceilf(sz);
    const unsigned fx = (unsigned)(int)fxf, fy = (unsigned)(int)fyf, fz = (unsigned)(int)fzf;
    const unsigned hcy = cy * TN_P1, hfy = fy * TN_P1, hcz = cz * TN_P2, hfz = fz * TN_P2;
    const unsigned m = g.mask, lm = (1u << slice_log2) - 1u;
    // (y, z) combinations in the order of kPairs = {0,3} {1,2} {4,7} {5,6}: (cy,cz) (fy,cz) (cy,fz) (fy,fz)
    const unsigned hyz[4] = {hcy ^ hcz, hfy ^ hcz, hcy ^ hfz, hfy ^ hfz};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned ia = (cx ^ hyz[p]) & m, ib = (fx ^ hyz[p]) & m;
        r.owner[p] = ia >> slice_log2;
        r.pair[p] = (ia & lm) | ((ib & lm) << 16);
    }
    if (VALUES) {
        const float ox = sub_rn(sx, fxf), oy = sub_rn(sy, fyf), oz = sub_rn(sz, fzf);
        const float qx = sub_rn(1.0f, ox), qy = sub_rn(1.0f, oy), qz = sub_rn(1.0f, oz);
        // enc = ((f0 ox + f3 qx) oy + (f1 ox + f2 qx) qy) oz + ((f4 ox + f7 qx) oy + (f5 ox + f6 qx) qy) qz
        // weights as the atomic kernel forms them: (ox * oy) * oz ... — keep its association
        const float wc[4] = {ox * oy * oz, ox * qy * oz, ox * oy * qz, ox * qy * qz};
        const float wf[4] = {qx * oy * oz, qx * qy * oz, qx * oy * qz, qx * qy * qz};
#pragma unroll
        for (int p = 0; p < 4; ++p) r.val[p] = make_float4(wc[p] * ge.x, wc[p] * ge.y, wf[p] * ge.x, wf[p] * ge.y);
    }
}

struct SortArgs {
    Grid g;
    tn_space space;
    const float *positions, *d_enc;
    long long n;
    int owners, slice_log2, level_begin, levels;  // levels [level_begin, level_begin + levels)
    unsigned *cursors;   // [bins] records reserved so far (may run past capacity)
    unsigned capacity;   // records per bin region
    unsigned *rec_pair;  // [bins][capacity]
    float4 *rec_val;
    float *d_table;      // for the records that overflow a region
};

// blockIdx.x = chunk * levels + level: kSortSamples consecutive samples at one level -> ranks, one reservation per bin, records.
// The records leave the block as CONTIGUOUS RUNS: a store instruction whose 64 lanes go to 64 different lines costs the CU
// ~4 cycles per line in the texture path (the first form wrote every record straight from the lane that made it — 32 such
// instructions per thread, 329 us per 786 k-sample call for 440 MB: bound by store instructions, not by bytes), so the block
// first gathers its records per bin in LDS (slot = bin's offset in the block + rank) and then copies LDS record r to
// base[bin] + (r - offset[bin]): consecutive threads write consecutive 16-byte records of one bin's run.
__global__ void __launch_bounds__(kBlock) sort_emit_kernel(SortArgs a) {
    __shared__ unsigned wave_tot[kBlock / 64];
    __shared__ unsigned total_s;
    extern __shared__ __attribute__((aligned(16))) unsigned char stage_raw[];
    constexpr int REC = kSortSamples * 4;       // records per block at most
    float4 *sval = reinterpret_cast<float4 *>(stage_raw);                                   // [REC]
    unsigned *spair = reinterpret_cast<unsigned *>(stage_raw + (size_t)REC * 16);           // [REC]
    unsigned short *sown = reinterpret_cast<unsigned short *>(stage_raw + (size_t)REC * 20);  // [REC]
    const int PER = (a.owners + kBlock - 1) / kBlock, own_pad = PER * kBlock;  // bins per thread in the scan
    unsigned *hist = reinterpret_cast<unsigned *>(stage_raw + (size_t)REC * 22);  // [own_pad] records per bin in this block
    unsigned *base = hist + own_pad;   // [own_pad] the run's first slot in the bin's global region
    unsigned *off = base + own_pad;    // [own_pad] the run's first slot in the block's LDS staging area
    const Space sp = make_space(a.space);
    const int L = a.g.num_levels;
    // workgroups go round the 8 XCDs (block b -> XCD b % 8), each with its own L2.  The blocks of one chunk — one per level — read
    // the same 128-byte rows of d_enc (8 bytes each): they are given the same XCD and consecutive turns on it, so that the row
    // comes from HBM / the Infinity Cache once instead of once per level.
    const long long turn = blockIdx.x >> 3;
    const long long chunk = (turn / a.levels) * 8 + (blockIdx.x & 7);
    const int lb = (int)(turn % a.levels), l = a.level_begin + lb;  // lb: level index inside the bins
    if (chunk * kSortSamples >= a.n) return;
    for (int o = threadIdx.x; o < own_pad; o += kBlock) hist[o] = 0u;
    __syncthreads();
    constexpr int SPT = kSortSamples / kBlock;
    SortRec rec[SPT];
    unsigned slot[SPT][4];
    bool has[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const long long i = chunk * kSortSamples + k * kBlock + threadIdx.x;
        float2 ge = make_float2(0.0f, 0.0f);
        if (i < a.n) ge = reinterpret_cast<const float2 *>(a.d_enc)[i * L + l];
        has[k] = ge.x != 0.0f || ge.y != 0.0f;  // a sample without gradient writes no record
        if (has[k]) {
            float px, py, pz;
            normalize_position(sp, a.positions[i * 3], a.positions[i * 3 + 1], a.positions[i * 3 + 2], px, py, pz);
            sort_records<true>(a.g, l, a.slice_log2, px, py, pz, ge, rec[k]);
#pragma unroll
            for (int p = 0; p < 4; ++p) slot[k][p] = atomicAdd(&hist[rec[k].owner[p]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of hist over the bins (PER per thread) -> off[]; the global reservation of every non-empty bin beside it
    {
        unsigned sum = 0u;
        for (int j = 0; j < PER; ++j) sum += hist[threadIdx.x * PER + j];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned before = 0u, all = 0u;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            if (w < wave) before += wave_tot[w];
            all += wave_tot[w];
        }
        unsigned run = before + incl - sum;
        for (int j = 0; j < PER; ++j) {
            const int o = threadIdx.x * PER + j;
            const unsigned h = hist[o];
            off[o] = run;
            run += h;
            base[o] = (h && o < a.owners) ? atomicAdd(&a.cursors[(size_t)lb * a.owners + o], h) : 0u;
        }
        if (threadIdx.x == 0) total_s = all;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        if (!has[k]) continue;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned o = rec[k].owner[p], at = off[o] + slot[k][p];
            sval[at] = rec[k].val[p];
            spair[at] = rec[k].pair[p];
            sown[at] = (unsigned short)o;
        }
    }
    __syncthreads();
    const unsigned total = total_s;
    float *tb = a.d_table + ((size_t)l * a.g.tsize) * 2;
    for (unsigned r = threadIdx.x; r < total; r += kBlock) {
        const unsigned o = sown[r], at = base[o] + (r - off[o]);
        const unsigned pk = spair[r];
        const float4 v = sval[r];
        if (at < a.capacity) {
            const size_t w = ((size_t)lb * a.owners + o) * a.capacity + at;
            a.rec_pair[w] = pk;
            a.rec_val[w] = v;
        } else {  // the bin's region is full: straight into the table
            float *pa = tb + (((size_t)o << a.slice_log2) + (pk & 0xffffu)) * 2;
            float *pb = tb + (((size_t)o << a.slice_log2) + (pk >> 16)) * 2;
            if (v.x != 0.0f) atomic_add_f32(pa, v.x);
            if (v.y != 0.0f) atomic_add_f32(pa + 1, v.y);
            if (v.z != 0.0f) atomic_add_f32(pb, v.z);
            if (v.w != 0.0f) atomic_add_f32(pb + 1, v.w);
        }
    }
}

// fp32 add into LDS through a compare-and-swap loop: ds_add_f32 retires 0.33 lanes per clock and CU on this part whatever the
// addresses, ds_cmpst_rtn_b32 / ds_add_u32 ~5-6 (tools/micro/lds_atomics.hip); a slice sees ~3 adds per float, so a loop
// almost never repeats.
__device__ __forceinline__ void lds_add_f32(float *p, float v) {
    unsigned *u = reinterpret_cast<unsigned *>(p);
    unsigned old = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (true) {
        const unsigned got = atomicCAS(u, old, __float_as_uint(__uint_as_float(old) + v));
        if (got == old) break;
        old = got;
    }
}

// one block per bin: the bucket's records into the LDS slice, the slice into d_table (+=)
__global__ void __launch_bounds__(kOwnerBlock, 1)
sort_owner_kernel(const unsigned *__restrict__ cursors, unsigned capacity, const unsigned *__restrict__ rec_pair,
                  const float4 *__restrict__ rec_val, int owners, int slice_log2, unsigned tsize, int level_begin,
                  float *__restrict__ d_table) {
    extern __shared__ __attribute__((aligned(16))) float slice[];  // [2 << slice_log2]
    const unsigned bin = blockIdx.x;
    const unsigned filled = min(cursors[bin], capacity);
    if (filled == 0u) return;  // nothing touched this slice
    const size_t region = (size_t)bin * capacity;
    rec_pair += region;
    rec_val += region;
    const unsigned r0 = 0u, r1 = filled;
    const int nfl = 2 << slice_log2;
    for (int e = threadIdx.x * 4; e < nfl; e += kOwnerBlock * 4) *reinterpret_cast<float4 *>(slice + e) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    // Consecutive records of a bin come from consecutive samples of a ray, which share their cell at the coarse levels: a wave
    // takes 64 CONSECUTIVE records, sums runs of equal index pairs with a segmented scan and only a run's last lane touches
    // LDS — same-address ds_add_f32 retire one after the other (~12 cycles each measured), and level 0 has ~300 addresses
    // per slice for ~100 k adds.
    const int lane = threadIdx.x & 63;
    for (unsigned rb = r0 + (threadIdx.x & ~63u); rb < r1; rb += kOwnerBlock) {
        const unsigned r = rb + lane;
        const bool live = r < r1;
        const unsigned pk = live ? rec_pair[r] : 0xffffffffu - lane;  // dead lanes: distinct keys nobody shares
        float4 v = live ? rec_val[r] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk, 0x138, 0xf, 0xf, false);  // wave_shr:1 (lane 0: unused)
        const int f = (lane > 0 && up == pk) ? 0 : 1;  // head of a run
        const int next_head = __builtin_amdgcn_update_dpp(1, f, 0x130, 0xf, 0xf, false);  // wave_shl:1 (lane 63 keeps 1)
        const bool tail = next_head != 0;
        {
            float sv[4] = {v.x, v.y, v.z, v.w};
            seg_scan_wave<4>(f, sv);
            v = make_float4(sv[0], sv[1], sv[2], sv[3]);
        }
        if (live && tail) {
            float *pa = slice + 2 * (pk & 0xffffu), *pb = slice + 2 * (pk >> 16);
            if (v.x != 0.0f) lds_add_f32(pa, v.x);
            if (v.y != 0.0f) lds_add_f32(pa + 1, v.y);
            if (v.z != 0.0f) lds_add_f32(pb, v.z);
            if (v.w != 0.0f) lds_add_f32(pb + 1, v.w);
        }
    }
    __syncthreads();
    const unsigned lb = bin / owners, o = bin - lb * owners, level = level_begin + lb;
    float *dst = d_table + ((size_t)level * tsize + ((size_t)o << slice_log2)) * 2;
    for (int e = threadIdx.x * 4; e < nfl; e += kOwnerBlock * 4) {
        const float4 add = *reinterpret_cast<const float4 *>(slice + e);
        if (add.x != 0.0f || add.y != 0.0f || add.z != 0.0f || add.w != 0.0f) {
            float4 cur = *reinterpret_cast<float4 *>(dst + e);
            cur.x += add.x; cur.y += add.y; cur.z += add.z; cur.w += add.w;
            *reinterpret_cast<float4 *>(dst + e) = cur;
        }
    }
}

struct SortLayout {
    int owners, slice_log2, bins;
    unsigned capacity;
    size_t slots, off_pair, off_val, bytes;
};
inline bool sort_layout(const tn_hashgrid &h, long long n, int level_begin, SortLayout &w) {
    if (level_begin < 0 || level_begin >= h.num_levels) return false;
    const int levels = h.num_levels - level_begin;
    w.slice_log2 = h.log2_hashmap_size < kSortSliceLog2 ? h.log2_hashmap_size : kSortSliceLog2;
    w.owners = 1 << (h.log2_hashmap_size - w.slice_log2);
    w.bins = w.owners * levels;
    // both x-neighbours of a record must fall into one slice: x (<= finest scaling + 1) may only touch bits below the slice
    float top = 0.0f;
    for (int l = 0; l < h.num_levels; ++l) top = h.scalings[l] > top ? h.scalings[l] : top;
    if (w.owners > 1 && top + 2.0f >= (float)(1 << w.slice_log2)) return false;
    if (w.owners > kSortMaxOwners || w.slice_log2 > 16) return false;
    // a bin's region: its share of an even spread of the level's 4 n records, + 25 % + 1024
    const long long share = (n * 4 + w.owners - 1) / w.owners;
    const long long cap = ((share + share / 4 + 1024) + 63) & ~63LL;
    const long long slots = cap * w.bins;
    if (cap >= (1LL << 31) || slots >= (1LL << 40)) return false;
    w.capacity = (unsigned)cap;
    w.slots = (size_t)slots;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    w.off_pair = up((size_t)w.bins * 4);
    w.off_val = w.off_pair + up(w.slots * 4);
    w.bytes = w.off_val + w.slots * 16;
    return true;
}

// Gradient of the encoding w.r.t. the WORLD position (camera-pose optimisation): d enc / d offset per level from the
// same 8 corners (table reads this time), times the level scale; then back through `p * selector`, the
// (x + 2) / 4 shift and the L-inf contraction (or the AABB normalisation).
// One lane per (sample, level): LP = 16 or 8 adjacent lanes hold the levels of one sample, so every lane has ONE memory round
// trip (a lane looping over the levels pays one per level: hipcc does not overlap the levels' gathers) and the d_enc read is
// contiguous; the three sums over the levels are an xor-shuffle tree over the LP lanes, lane 0 of a group finishes the sample.
template <int LP>
__global__ void __launch_bounds__(kBlock)
hash_encode_bwd_input_kernel(Grid g, tn_space space, const float *__restrict__ positions, const float *__restrict__ d_enc,
                             long long n, float *__restrict__ d_pos) {
    const Space sp = make_space(space);
    const int L = g.num_levels;
    constexpr int SPW = 64 / LP;  // samples per wave
    const int lane = threadIdx.x & 63;
    const int l = lane % LP, sub = lane / LP;
    const long long wstride = (long long)gridDim.x * (kBlock / 64) * SPW;
    for (long long base = ((long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * SPW; base < n; base += wstride) {
        const long long i = base + sub;
        const bool live = i < n;
        const long long ic = live ? i : n - 1;
        const float x = positions[ic * 3], y = positions[ic * 3 + 1], z = positions[ic * 3 + 2];
        float px, py, pz;
        const float sel = normalize_position(sp, x, y, z, px, py, pz);
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;  // d loss / d p (p = normalised, selector applied)
        if (live && l < L) encode_level_grad(g, l, px, py, pz, reinterpret_cast<const float2 *>(d_enc)[ic * L + l], gx, gy, gz);
#pragma unroll
        for (int o = LP / 2; o > 0; o >>= 1) {
            gx += __shfl_xor(gx, o, 64);
            gy += __shfl_xor(gy, o, 64);
            gz += __shfl_xor(gz, o, 64);
        }
        if (!live || l != 0) continue;
        float rx, ry, rz;
        position_grad_finish(sp, x, y, z, sel, gx, gy, gz, rx, ry, rz);
        d_pos[i * 3] = rx; d_pos[i * 3 + 1] = ry; d_pos[i * 3 + 2] = rz;
    }
}

// Frustums.get_positions backward: pos = o + d (s + e) / 2  ->  d_o += sum_i g_i,  d_d += sum_i g_i (s_i + e_i) / 2.
__global__ void __launch_bounds__(kBlock)
frustum_positions_bwd_kernel(const float *__restrict__ d_pos, const float *__restrict__ starts,
                             const float *__restrict__ ends, long long R, int n, float *__restrict__ d_o,
                             float *__restrict__ d_d) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    float a[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = lane; i < n; i += 64) {
        const long long k = ray * n + i;
        const float t = (starts[k] + ends[k]) / 2.0f;
        const float gx = d_pos[k * 3], gy = d_pos[k * 3 + 1], gz = d_pos[k * 3 + 2];
        a[0] += gx; a[1] += gy; a[2] += gz;
        a[3] += gx * t; a[4] += gy * t; a[5] += gz * t;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) a[q] = wave_sum(a[q]);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            d_o[ray * 3 + q] += a[q];
            d_d[ray * 3 + q] += a[3 + q];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Linear layers.  Backward: a block owns tiles of 64 rows (samples); 256 threads = 64 rows x 4 column groups.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == TN_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == TN_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}
__device__ __forceinline__ float act_bwd(float y, int act) {  // derivative expressed through the OUTPUT y
    if (act == TN_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (act == TN_ACT_SIGMOID) return y * (1.0f - y);
    return 1.0f;
}

constexpr int TILE = 64;
constexpr int kLinBwdBlocks = 768;  // persistent blocks of tn_linear_bwd (3 per CU fit the 49 KB of LDS each)
constexpr int LDP = 65;  // padded row length of the row tiles in LDS (odd: conflict-free column walks)

// NI = inputs per thread (IN <= 4 * NI)
template <int NI>
__global__ void __launch_bounds__(kBlock)
linear_bwd_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ y, const float *__restrict__ dy, int ldy,
                  const float *__restrict__ W, int IN, int OUT, int act, long long n, float *__restrict__ dx, int lddx,
                  int accumulate_dx, float *__restrict__ dW, float *__restrict__ db, float *__restrict__ partials) {
    constexpr int INP = 4 * NI;
    __shared__ __attribute__((aligned(16))) float Ws[64 * INP];  // [o][i], zero padded
    __shared__ float gs[TILE * LDP];                               // g = dy * act'(y)   [row][o]
    __shared__ float xs[TILE * LDP];                               // [row][i]
    for (int e = threadIdx.x; e < OUT * INP; e += kBlock) {
        const int o = e / INP, i = e - o * INP;
        Ws[e] = i < IN ? W[o * IN + i] : 0.0f;
    }
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float accw[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) accw[k] = 0.0f;
    float accb = 0.0f;
    const bool want_w = dW != nullptr || db != nullptr;
    const long long tiles = (n + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        __syncthreads();
        for (int e = threadIdx.x; e < TILE * OUT; e += kBlock) {
            const int r = e / OUT, o = e - r * OUT;
            float g = 0.0f;
            if (base + r < n) {
                const size_t a = (size_t)(base + r) * ldy + o;
                g = dy[a] * act_bwd(act == TN_ACT_NONE ? 0.0f : y[a], act);
            }
            gs[r * LDP + o] = g;
        }
        if (want_w) {
            for (int e = threadIdx.x; e < TILE * INP; e += kBlock) {
                const int r = e / INP, i = e - r * INP;
                xs[r * LDP + i] = (base + r < n && i < IN) ? x[(base + r) * ldx + i] : 0.0f;
            }
        }
        __syncthreads();
        if (dx) {  // dx[row = lane][i = grp*NI + k] = sum_o g[row][o] W[o][i]
            float acc[NI];
#pragma unroll
            for (int k = 0; k < NI; ++k) acc[k] = 0.0f;
            for (int o = 0; o < OUT; ++o) {
                const float gv = gs[lane * LDP + o];
                const float *wr = Ws + o * INP + grp * NI;
#pragma unroll
                for (int k = 0; k < NI; ++k) acc[k] = fmaf(wr[k], gv, acc[k]);
            }
            if (base + lane < n) {
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int i = grp * NI + k;
                    if (i < IN) {
                        float *p = dx + (size_t)(base + lane) * lddx + i;
                        *p = accumulate_dx ? *p + acc[k] : acc[k];
                    }
                }
            }
        }
        if (want_w && lane < OUT) {  // dW[o = lane][i = grp*NI + k] += sum_rows g[row][o] x[row][i]
            for (int r = 0; r < TILE; ++r) {
                const float gv = gs[r * LDP + lane];
                const float *xr = xs + r * LDP + grp * NI;
#pragma unroll
                for (int k = 0; k < NI; ++k) accw[k] = fmaf(xr[k], gv, accw[k]);
                if (grp == 0) accb += gv;
            }
        }
    }
    if (want_w && lane < OUT) {
        if (partials) {  // [block][i (INP rows) | bias row][64]: plain coalesced stores, summed by linear_bwd_reduce_kernel
            float *pb = partials + (size_t)blockIdx.x * (INP + 1) * 64;
#pragma unroll
            for (int k = 0; k < NI; ++k) pb[(grp * NI + k) * 64 + lane] = accw[k];
            if (grp == 0) pb[INP * 64 + lane] = accb;
        } else {
            if (dW) {
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const int i = grp * NI + k;
                    if (i < IN) atomic_add_f32(dW + lane * IN + i, accw[k]);
                }
            }
            if (db && grp == 0) atomic_add_f32(db + lane, accb);
        }
    }
}

// The same backward on the matrix pipe (v_mfma_f32_32x32x2_f32; fp32 MFMA runs at the packed-VALU rate, but its operands
// come from two conflict-free ds_read_b32 per 2048 MACs instead of one broadcast LDS read per 4 FMAs, which is what binds the
// VALU form above).  Per 64-row tile, 4 waves = 4 output tiles of 32x32:
//   dx^T[i][n] = sum_o W[o][i] g[n][o]     A = W^T (rows i, k = o)      B = g^T (k = o, cols n)     K = OUT
//   dW  [o][i] = sum_n g[n][o] x[n][i]     A = g^T (rows o, k = n)      B = x   (k = n, cols i)     K = 64 rows
// dx^T is transposed back through LDS for coalesced stores; dW accumulates in registers across the block's tiles.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ void __launch_bounds__(kBlock)
linear_bwd_mfma_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ y, const float *__restrict__ dy, int ldy,
                       const float *__restrict__ W, int ldw, int IN, int OUT, int act, long long n, float *__restrict__ dx,
                       int lddx, int accumulate_dx, int want_w, float *__restrict__ partials, int vec, int vec_dx) {
    // (W: the [OUT][IN] block of a weight matrix whose rows are ldw floats apart — the whole matrix, ldw = IN, for layers up to 64
    // wide; one 64 x 64 block of a wider layer otherwise, see tn_linear_bwd)
    __shared__ float Ws[64 * LDP];  // [o][i], zero padded to 64 x 64
    __shared__ float gs[TILE * LDP];  // [row][o]
    __shared__ float xs[TILE * LDP];  // [row][i]; reused as dx staging [row][i]
    for (int e = threadIdx.x; e < 64 * 64; e += kBlock) {
        const int o = e >> 6, i = e & 63;
        Ws[o * LDP + i] = (o < OUT && i < IN) ? W[o * ldw + i] : 0.0f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    const int n_it = (IN + 31) >> 5, n_ot = (OUT + 31) >> 5;
    const int ksteps_o = (OUT + 1) >> 1;
    // dx tile of this wave: (it, nt); dW tile: (ot, it2)
    const bool has_dx = dx != nullptr && wave < n_it * 2;
    const int it = wave % n_it, nt = wave / n_it;
    const bool has_dw = want_w && wave < n_ot * n_it;
    const int ot = wave % n_ot, it2 = wave / n_ot;
    f32x16 accw;
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[r] = 0.0f;
    float accb = 0.0f;
    const long long tiles = (n + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        __syncthreads();
        if (vec) {  // 16-byte loads: rows are float4-aligned and the widths are multiples of 4
            float4 gq[4], xq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = threadIdx.x + q * kBlock, r = e >> 4, c = (e & 15) * 4;
                gq[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                xq[q] = gq[q];
                if (base + r < n) {
                    if (c < OUT) {
                        const size_t a = (size_t)(base + r) * ldy + c;
                        const float4 d4 = *reinterpret_cast<const float4 *>(dy + a);
                        float4 y4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if (act != TN_ACT_NONE) y4 = *reinterpret_cast<const float4 *>(y + a);
                        gq[q] = make_float4(d4.x * act_bwd(y4.x, act), d4.y * act_bwd(y4.y, act), d4.z * act_bwd(y4.z, act),
                                            d4.w * act_bwd(y4.w, act));
                    }
                    if (want_w && c < IN) xq[q] = *reinterpret_cast<const float4 *>(x + (size_t)(base + r) * ldx + c);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = threadIdx.x + q * kBlock, r = e >> 4, c = (e & 15) * 4;
                float *gp = gs + r * LDP + c, *xp = xs + r * LDP + c;
                gp[0] = gq[q].x; gp[1] = gq[q].y; gp[2] = gq[q].z; gp[3] = gq[q].w;
                xp[0] = xq[q].x; xp[1] = xq[q].y; xp[2] = xq[q].z; xp[3] = xq[q].w;
            }
        } else {
            for (int e = threadIdx.x; e < TILE * 64; e += kBlock) {
                const int r = e >> 6, c = e & 63;
                float g = 0.0f, xv = 0.0f;
                if (base + r < n) {
                    if (c < OUT) {
                        const size_t a = (size_t)(base + r) * ldy + c;
                        g = dy[a] * act_bwd(act == TN_ACT_NONE ? 0.0f : y[a], act);
                    }
                    if (want_w && c < IN) xv = x[(base + r) * ldx + c];
                }
                gs[r * LDP + c] = g;
                xs[r * LDP + c] = xv;
            }
        }
        __syncthreads();
        if (has_dw) {
#pragma unroll 8
            for (int s2 = 0; s2 < 32; ++s2) {
                const float a = gs[(2 * s2 + h) * LDP + ot * 32 + l31];
                const float b = xs[(2 * s2 + h) * LDP + it2 * 32 + l31];
                accw = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accw, 0, 0, 0);
            }
        }
        if (want_w && threadIdx.x < 64) {  // bias: column sums of g
            float sb = 0.0f;
            for (int r = 0; r < TILE; ++r) sb += gs[r * LDP + threadIdx.x];
            accb += sb;
        }
        f32x16 accd;
        if (has_dx) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accd[r] = 0.0f;
            for (int s2 = 0; s2 < ksteps_o; ++s2) {
                const float a = Ws[(2 * s2 + h) * LDP + it * 32 + l31];
                const float b = gs[(nt * 32 + l31) * LDP + 2 * s2 + h];
                accd = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accd, 0, 0, 0);
            }
        }
        if (dx) {
            __syncthreads();  // everyone is done reading xs
            if (has_dx) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xs[(nt * 32 + l31) * LDP + it * 32 + crow(r, h)] = accd[r];
            }
            __syncthreads();
            if (vec_dx) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = threadIdx.x + q * kBlock, r = e >> 4, c = (e & 15) * 4;
                    if (base + r < n && c < IN) {
                        const float *sp = xs + r * LDP + c;
                        *reinterpret_cast<float4 *>(dx + (size_t)(base + r) * lddx + c) = make_float4(sp[0], sp[1], sp[2], sp[3]);
                    }
                }
            } else {
                for (int e = threadIdx.x; e < TILE * 64; e += kBlock) {
                    const int r = e >> 6, i = e & 63;
                    if (base + r < n && i < IN) {
                        float *p = dx + (size_t)(base + r) * lddx + i;
                        const float v = xs[r * LDP + i];
                        *p = accumulate_dx ? *p + v : v;
                    }
                }
            }
        }
    }
    if (want_w) {  // slab [i (64 rows) | bias row][o (64)]
        float *pb = partials + (size_t)blockIdx.x * 65 * 64;
        __syncthreads();
        for (int e = threadIdx.x; e < 65 * 64; e += kBlock) pb[e] = 0.0f;
        __syncthreads();
        if (has_dw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[(it2 * 32 + l31) * 64 + ot * 32 + crow(r, h)] = accw[r];
        }
        if (threadIdx.x < 64) pb[64 * 64 + threadIdx.x] = accb;
    }
}


// ------------------------------------------------------------------------------------------------------
// linear_chain_bwd_kernel — the backward of up to three CONSECUTIVE Linear(+activation) layers of an MLP in one launch
// (mlp_head: 3 layers, mlp_thermal + head: 3, mlp_base: 2).  Layer by layer the single-layer kernel above streams x, y, dy in
// and dx out of HBM: four [N, 64] matrices per layer.  In a chain the input x_j of layer j IS the activated output of the
// layer below, so  g_{j+1} = dx_j . act'(x_j)  is formed in LDS from the tile that was loaded for dW_j anyway: per layer ONE
// [64-row, width] tile is read, and only the chain's last dx is written.  Per 64-row tile and layer the 4 waves own the 4
// 32x32 tiles of dW (g^T x, K = the 64 rows) and of dx^T (W^T g^T, K = OUT), as in linear_bwd_mfma_kernel; dW of every layer
// accumulates in registers across the block's tiles and leaves as one slab per layer and block (linear_bwd_reduce_kernel).
// ------------------------------------------------------------------------------------------------------
constexpr int kChainMax = 3;
constexpr int kChainBlocks = 512;  // persistent; LDS = the layers' weights (OUT rows of 64) + 2 row tiles: <= 80 KB -> 2 blocks per CU

struct ChainLayerDev {
    const float *W;   // [OUT][IN]
    const float *x;   // input rows of the layer (= activated output of the layer below), ldx floats apart
    int IN, OUT, ldx, act_x, vec_x;
};
struct ChainArgs {
    ChainLayerDev L[kChainMax];  // L[0] = the layer nearest the loss
    int nl;
    const float *dy;     // [n, OUT_0] rows lddy apart
    const float *y_top;  // activated output of layer 0 (its activation's derivative), or nullptr when act_top == none
    int lddy, act_top, vec_top;
    long long n;
    float *dx;           // [n, IN_last]
    int lddx, accumulate_dx, vec_dx;
    float *partials;     // [nl][blocks][65][64]
};

// a [64-row, <= 64 column] tile of a row-major matrix in 16 registers per thread (thread t: row (t + 256 q) >> 4, columns
// 4 ((t + 256 q) & 15) .. +3, q = 0..3); vec: 16-byte loads, else element loads; out-of-range entries are zero
struct TileRegs {
    float4 v[4];
};
__device__ __forceinline__ void tile_fetch(TileRegs &t, const float *__restrict__ p, int ld, int cols, long long base, long long n,
                                           int vec) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + q * kBlock, r = e >> 4, c = (e & 15) * 4;
        float4 x4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (base + r < n && c < cols) {
            const float *src = p + (size_t)(base + r) * ld + c;
            if (vec) {
                x4 = *reinterpret_cast<const float4 *>(src);
            } else {
                x4.x = src[0];
                if (c + 1 < cols) x4.y = src[1];
                if (c + 2 < cols) x4.z = src[2];
                if (c + 3 < cols) x4.w = src[3];
            }
        }
        t.v[q] = x4;
    }
}
__device__ __forceinline__ void tile_to_lds(const TileRegs &t, float *dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + q * kBlock, r = e >> 4, c = (e & 15) * 4;
        float *d = dst + r * LDP + c;
        d[0] = t.v[q].x; d[1] = t.v[q].y; d[2] = t.v[q].z; d[3] = t.v[q].w;
    }
}

template <int NL>
__global__ void __launch_bounds__(kBlock, 2) linear_chain_bwd_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // W_j as [o][i]: even(OUT_j) rows of 64 (read along rows only: no padding needed), zero padded
    float *Ws = smem;
    int woff[NL + 1];
    woff[0] = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) woff[j + 1] = woff[j] + ((a.L[j].OUT + 1) & ~1) * 64;
    float *gs = Ws + woff[NL];              // [TILE * LDP]    g rows [row][o]
    float *xs = gs + TILE * LDP;            // [TILE * LDP]    x rows [row][i]; staging of the final dx
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int IN = a.L[j].IN, OUT = a.L[j].OUT;
        for (int e = threadIdx.x; e < ((OUT + 1) & ~1) * 64; e += kBlock) {
            const int o = e >> 6, i = e & 63;
            Ws[woff[j] + e] = (o < OUT && i < IN) ? a.L[j].W[o * IN + i] : 0.0f;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    f32x16 accw[NL];
    float accb[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[j][r] = 0.0f;
        accb[j] = 0.0f;
    }
    const long long tiles = (a.n + TILE - 1) / TILE;
    // Every tile of the chain (dy, y_top, x_0 .. x_{NL-1}) is fetched into registers ONE TILE AHEAD: a register set is
    // re-issued for the next tile right after it was copied to LDS, so the loads fly during the MFMAs of the current tile and
    // the block never waits for HBM in steady state (one block per CU: there is no other block to hide the latency).
    TileRegs rdy, ry, rx[NL];
    const int OUT0 = a.L[0].OUT;
    auto fetch_top = [&](long long tile) {
        const long long base = tile * TILE;
        tile_fetch(rdy, a.dy, a.lddy, OUT0, base, a.n, a.vec_top);
        if (a.act_top != TN_ACT_NONE) tile_fetch(ry, a.y_top, a.lddy, OUT0, base, a.n, a.vec_top);
    };
    if ((long long)blockIdx.x < tiles) {
        fetch_top(blockIdx.x);
#pragma unroll
        for (int j = 0; j < NL; ++j) tile_fetch(rx[j], a.L[j].x, a.L[j].ldx, a.L[j].IN, (long long)blockIdx.x * TILE, a.n, a.L[j].vec_x);
    }
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        const long long next = tile + gridDim.x;
        __syncthreads();  // the previous tile's readers of gs / xs are done
        {   // g of the top layer: dy . act'(y)
            TileRegs g0 = rdy;
            if (a.act_top != TN_ACT_NONE) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    g0.v[q].x *= act_bwd(ry.v[q].x, a.act_top);
                    g0.v[q].y *= act_bwd(ry.v[q].y, a.act_top);
                    g0.v[q].z *= act_bwd(ry.v[q].z, a.act_top);
                    g0.v[q].w *= act_bwd(ry.v[q].w, a.act_top);
                }
            }
            tile_to_lds(g0, gs);
            if (next < tiles) fetch_top(next);
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const ChainLayerDev &L = a.L[j];
            const int IN = L.IN, OUT = L.OUT;
            const int n_it = (IN + 31) >> 5, n_ot = (OUT + 31) >> 5, ksteps_o = (OUT + 1) >> 1;
            const float *Wj = Ws + woff[j];
            tile_to_lds(rx[j], xs);
            if (next < tiles) tile_fetch(rx[j], L.x, L.ldx, IN, next * TILE, a.n, L.vec_x);
            __syncthreads();  // gs (g_j) and xs (x_j) complete
            const int ot = wave % n_ot, it2 = wave / n_ot;
            if (wave < n_ot * n_it) {  // dW_j tile (ot, it2)
#pragma unroll 8
                for (int s2 = 0; s2 < 32; ++s2) {
                    const float av = gs[(2 * s2 + h) * LDP + ot * 32 + l31];
                    const float bv = xs[(2 * s2 + h) * LDP + it2 * 32 + l31];
                    accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw[j], 0, 0, 0);
                }
            }
            {   // bias_j: column sums of g_j, a quarter of the rows per wave (summed over the waves when the slab is written)
                float sb = 0.0f;
#pragma unroll
                for (int r = 0; r < TILE / 4; ++r) sb += gs[(wave * (TILE / 4) + r) * LDP + lane];
                accb[j] += sb;
            }
            const bool last = j == NL - 1;
            const bool has_dx = (!last || a.dx != nullptr) && wave < n_it * 2;
            const int it = wave % n_it, nt = wave / n_it;
            f32x16 accd;
            if (has_dx) {  // dx_j^T tile (it, nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) accd[r] = 0.0f;
#pragma unroll 8
                for (int s2 = 0; s2 < ksteps_o; ++s2) {
                    const float av = Wj[(2 * s2 + h) * 64 + it * 32 + l31];
                    const float bv = gs[(nt * 32 + l31) * LDP + 2 * s2 + h];
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accd, 0, 0, 0);
                }
            }
            __syncthreads();  // every MFMA operand read of gs / xs is done
            if (!last) {
                // g_{j+1}[row][i] = dx_j[row][i] . act'(x_j[row][i]) -> gs
                if (has_dx) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = nt * 32 + l31, col = it * 32 + crow(r, h);
                        gs[row * LDP + col] = accd[r] * act_bwd(xs[row * LDP + col], L.act_x);
                    }
                }
                if (n_it == 1) {  // the upper 32 columns are not covered by a dx tile: clear them (the next layer's K range)
                    for (int e = threadIdx.x; e < TILE * 32; e += kBlock) gs[(e >> 5) * LDP + 32 + (e & 31)] = 0.0f;
                }
                __syncthreads();  // the next layer's x tile overwrites xs: all act' reads above are done
            } else if (a.dx) {
                if (has_dx) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) xs[(nt * 32 + l31) * LDP + it * 32 + crow(r, h)] = accd[r];
                }
                __syncthreads();
                if (a.vec_dx) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int e = threadIdx.x + q * kBlock, r = e >> 4, c = (e & 15) * 4;
                        if (base + r < a.n && c < IN) {
                            const float *sp = xs + r * LDP + c;
                            *reinterpret_cast<float4 *>(a.dx + (size_t)(base + r) * a.lddx + c) = make_float4(sp[0], sp[1], sp[2], sp[3]);
                        }
                    }
                } else {
                    for (int e = threadIdx.x; e < TILE * 64; e += kBlock) {
                        const int r = e >> 6, i = e & 63;
                        if (base + r < a.n && i < IN) {
                            float *p = a.dx + (size_t)(base + r) * a.lddx + i;
                            const float v = xs[r * LDP + i];
                            *p = a.accumulate_dx ? *p + v : v;
                        }
                    }
                }
            }
        }
    }
    // one slab [i (64 rows) | bias row][o (64)] per layer and block
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int n_it = (a.L[j].IN + 31) >> 5, n_ot = (a.L[j].OUT + 31) >> 5;
        float *pb = a.partials + ((size_t)j * gridDim.x + blockIdx.x) * 65 * 64;
        __syncthreads();
        for (int e = threadIdx.x; e < 65 * 64; e += kBlock) pb[e] = 0.0f;
        __syncthreads();
        const int ot = wave % n_ot, it2 = wave / n_ot;
        if (wave < n_ot * n_it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[(it2 * 32 + l31) * 64 + ot * 32 + crow(r, h)] = accw[j][r];
        }
        __syncthreads();  // the four waves' bias partials meet in LDS (the row tiles are no longer needed)
        gs[wave * 64 + lane] = accb[j];
        __syncthreads();
        if (wave == 0) pb[64 * 64 + lane] = (gs[lane] + gs[64 + lane]) + (gs[128 + lane] + gs[192 + lane]);
    }
}

// Layers wider than 64 (config.hidden_dim / hidden_dim_color / hidden_dim_transient up to kLinWide, stage-by-stage fields only):
// a block owns tiles of 64 rows, the x tile in LDS [64][IN | 1]; lane = row, a wave walks the outputs four at a time (its weight
// rows are wave-uniform: scalar loads).  Same order of additions as the narrow kernel: bias first, inputs ascending.
constexpr int kLinWide = 256;
__global__ void __launch_bounds__(kBlock)
linear_fwd_wide_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ W, const float *__restrict__ b, int IN,
                       int OUT, int act, long long n, float *__restrict__ y, int ldy, int vec_out) {
    extern __shared__ __attribute__((aligned(16))) float xs_wide[];
    const int LDK = IN | 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long tiles = (n + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long base = tile * TILE;
        __syncthreads();
        for (int e = threadIdx.x; e < TILE * IN; e += kBlock) {
            const int r = e / IN, c = e - r * IN;
            xs_wide[r * LDK + c] = base + r < n ? x[(size_t)(base + r) * ldx + c] : 0.0f;
        }
        __syncthreads();
        const float *xr = xs_wide + lane * LDK;
        const bool live = base + lane < n;
        float *yp = y + (size_t)(base + lane) * ldy;
        for (int o0 = 4 * wave; o0 < OUT; o0 += 4 * (kBlock / 64)) {
            const int oa = o0, ob = min(o0 + 1, OUT - 1), oc = min(o0 + 2, OUT - 1), od = min(o0 + 3, OUT - 1);
            const float *wa = W + (size_t)oa * IN, *wb = W + (size_t)ob * IN, *wc = W + (size_t)oc * IN, *wd = W + (size_t)od * IN;
            float a0 = b ? b[oa] : 0.0f, a1 = b ? b[ob] : 0.0f, a2 = b ? b[oc] : 0.0f, a3 = b ? b[od] : 0.0f;
#pragma unroll 8
            for (int i = 0; i < IN; ++i) {
                const float xv = xr[i];
                a0 = fmaf(xv, wa[i], a0);
                a1 = fmaf(xv, wb[i], a1);
                a2 = fmaf(xv, wc[i], a2);
                a3 = fmaf(xv, wd[i], a3);
            }
            if (!live) continue;
            if (vec_out) {
                *reinterpret_cast<float4 *>(yp + o0) = make_float4(act_fwd(a0, act), act_fwd(a1, act), act_fwd(a2, act), act_fwd(a3, act));
            } else {
                yp[o0] = act_fwd(a0, act);
                if (o0 + 1 < OUT) yp[o0 + 1] = act_fwd(a1, act);
                if (o0 + 2 < OUT) yp[o0 + 2] = act_fwd(a2, act);
                if (o0 + 3 < OUT) yp[o0 + 3] = act_fwd(a3, act);
            }
        }
    }
}

// dW[o][i] += sum_b partials[b][i][o];  db[o] += sum_b partials[b][INP][o].  grid (INP + 1 rows, kRedSplit slices of
// the block range); 256 threads = 64 outputs x 4 interleaved block streams; one atomic per (entry, slice).  dW rows ldw apart.
constexpr int kRedSplit = 8;
__global__ void __launch_bounds__(kBlock)
linear_bwd_reduce_kernel(const float *__restrict__ partials, int blocks, int INP, int IN, int OUT, float *__restrict__ dW,
                         int ldw, float *__restrict__ db) {
    __shared__ float red[kBlock];
    const int i = blockIdx.x, o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int per = (blocks + kRedSplit - 1) / kRedSplit;
    const int b0 = blockIdx.y * per, b1 = min(blocks, b0 + per);
    const float *p = partials + (size_t)i * 64 + o;
    const size_t st = (size_t)(INP + 1) * 64;
    float s0 = 0.0f, s1 = 0.0f;
    int b = b0 + q;
    for (; b + 4 < b1; b += 8) {
        s0 += p[(size_t)b * st];
        s1 += p[(size_t)(b + 4) * st];
    }
    if (b < b1) s0 += p[(size_t)b * st];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (q == 0 && o < OUT) {
        const float tot = (red[o] + red[64 + o]) + (red[128 + o] + red[192 + o]);
        if (i == INP) {
            if (db) atomic_add_f32(db + o, tot);
        } else if (i < IN && dW) {
            atomic_add_f32(dW + o * ldw + i, tot);
        }
    }
}

// the same for all layers of a chain in one launch (blockIdx.z = layer; slabs are [65][64] per block)
struct RedChainArgs {
    const float *partials[kChainMax];
    float *dW[kChainMax], *db[kChainMax];
    int IN[kChainMax], OUT[kChainMax];
};
__global__ void __launch_bounds__(kBlock) linear_bwd_reduce_chain_kernel(RedChainArgs a, int blocks) {
    __shared__ float red[kBlock];
    const int j = blockIdx.z;
    const float *partials = a.partials[j];
    float *dW = a.dW[j], *db = a.db[j];
    if (!dW && !db) return;
    const int IN = a.IN[j], OUT = a.OUT[j];
    const int i = blockIdx.x, o = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (i != 64 && i >= IN) return;  // rows of padded inputs
    const int per = (blocks + kRedSplit - 1) / kRedSplit;
    const int b0 = blockIdx.y * per, b1 = min(blocks, b0 + per);
    const float *p = partials + (size_t)i * 64 + o;
    const size_t st = (size_t)65 * 64;
    float s0 = 0.0f, s1 = 0.0f;
    int b = b0 + q;
    for (; b + 4 < b1; b += 8) {
        s0 += p[(size_t)b * st];
        s1 += p[(size_t)(b + 4) * st];
    }
    if (b < b1) s0 += p[(size_t)b * st];
    red[threadIdx.x] = s0 + s1;
    __syncthreads();
    if (q == 0 && o < OUT) {
        const float tot = (red[o] + red[64 + o]) + (red[128 + o] + red[192 + o]);
        if (i == 64) {
            if (db) atomic_add_f32(db + o, tot);
        } else if (dW) {
            atomic_add_f32(dW + o * IN + i, tot);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Linear forward, lane = row.  A wave-uniform operand costs nothing when it comes through the SCALAR cache (s_load +
// v_pk_fma with SGPR sources), while the same value broadcast from LDS costs LDS->VGPR bandwidth for all 64 lanes (what
// bounds the tiled kernel above): the row's inputs sit in VGPRs, W[o][:] is read as scalars, no LDS, no barriers:
// 96 -> 50 us per 64x64 layer over 196 k rows.  (The same idea for dx — 64x64 fully unrolled, instruction-cache bound —
// and for dW — x rows streamed through the scalar cache — measured 2-3x SLOWER than the tiled backward kernel: kept out.)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int INP>
__global__ void __launch_bounds__(kBlock)
linear_fwd_rows_kernel(const float *__restrict__ x, int ldx, const float *__restrict__ W, const float *__restrict__ b, int IN,
                       int OUT, int act, long long n, float *__restrict__ y, int ldy, int vec_in, int vec_out) {
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long base = (long long)blockIdx.x * kBlock; base < n; base += stride) {
        const long long row = base + threadIdx.x;
        const bool live = row < n;
        const float *xp = x + (live ? row : n - 1) * ldx;
        float xr[INP];
        if (vec_in) {
#pragma unroll
            for (int i = 0; i < INP; i += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(xp + i);
                xr[i] = v.x; xr[i + 1] = v.y; xr[i + 2] = v.z; xr[i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < INP; ++i) xr[i] = i < IN ? xp[i] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < INP; ++i)
            if (i >= IN) xr[i] = 0.0f;
        float *yp = y + row * ldy;
        for (int o0 = 0; o0 < OUT; o0 += 4) {
            const int oa = o0, ob = min(o0 + 1, OUT - 1), oc = min(o0 + 2, OUT - 1), od = min(o0 + 3, OUT - 1);
            const float *wa = W + oa * IN, *wb = W + ob * IN, *wc = W + oc * IN, *wd = W + od * IN;
            float a0 = b ? b[oa] : 0.0f, a1 = b ? b[ob] : 0.0f, a2 = b ? b[oc] : 0.0f, a3 = b ? b[od] : 0.0f;
#pragma unroll
            for (int c = 0; c < INP; c += 4) {
                if (c + 3 < IN) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        a0 = fmaf(xr[c + k], wa[c + k], a0);
                        a1 = fmaf(xr[c + k], wb[c + k], a1);
                        a2 = fmaf(xr[c + k], wc[c + k], a2);
                        a3 = fmaf(xr[c + k], wd[c + k], a3);
                    }
                } else if (c < IN) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = min(c + k, IN - 1);  // xr is zero beyond IN: the clamped weight is multiplied by 0
                        a0 = fmaf(xr[c + k], wa[i], a0);
                        a1 = fmaf(xr[c + k], wb[i], a1);
                        a2 = fmaf(xr[c + k], wc[i], a2);
                        a3 = fmaf(xr[c + k], wd[i], a3);
                    }
                }
            }
            if (live) {
                if (vec_out) {
                    *reinterpret_cast<float4 *>(yp + o0) = make_float4(act_fwd(a0, act), act_fwd(a1, act), act_fwd(a2, act), act_fwd(a3, act));
                } else {
                    yp[o0] = act_fwd(a0, act);
                    if (o0 + 1 < OUT) yp[o0 + 1] = act_fwd(a1, act);
                    if (o0 + 2 < OUT) yp[o0 + 2] = act_fwd(a2, act);
                    if (o0 + 3 < OUT) yp[o0 + 3] = act_fwd(a3, act);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// density activation
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
density_act_fwd_kernel(const float *__restrict__ raw, int ld, const float *__restrict__ sel, float avg, long long n,
                       float *__restrict__ density) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
        density[i] = mul_rn(mul_rn(avg, expf(raw[i * ld])), sel[i]);
}
__global__ void __launch_bounds__(kBlock)
density_act_bwd_kernel(const float *__restrict__ raw, int ld, const float *__restrict__ sel, float avg, float clamp_min,
                       const float *__restrict__ dd, long long n, float *__restrict__ d_raw, int ldd, int clear_cols) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const float g = dd[i] * sel[i] * avg * expf(fminf(fmaxf(raw[i * ld], clamp_min), 15.0f));
        float *row = d_raw + i * ldd;
        if (clear_cols > 1 && (ldd & 3) == 0 && (clear_cols & 3) == 0 && (reinterpret_cast<uintptr_t>(d_raw) & 15) == 0) {
            // the row's other columns are the += targets of later stages: whole 16-byte pieces (a 16-float row = one line)
            *reinterpret_cast<float4 *>(row) = make_float4(g, 0.0f, 0.0f, 0.0f);
            for (int c = 4; c < clear_cols; c += 4) *reinterpret_cast<float4 *>(row + c) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            row[0] = g;
            for (int c = 1; c < clear_cols; ++c) row[c] = 0.0f;
        }
    }
}

// NS scale_gradients_by_distance_squared (use_gradient_scaling, REF thermal_nerf_model.py:228-231): backward-only —
// every field output's gradient is multiplied by clamp(((start + end) / 2)^2, 0, 1) of its sample.
__global__ void __launch_bounds__(kBlock)
gradient_scale_kernel(const float *__restrict__ starts, const float *__restrict__ ends, long long n, float *__restrict__ d_density,
                      float *__restrict__ d_rgb, float *__restrict__ d_thermal) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const float dist = (starts[i] + ends[i]) / 2.0f;
        const float s = fminf(fmaxf(dist * dist, 0.0f), 1.0f);
        if (d_density) d_density[i] *= s;
        if (d_rgb) {
            d_rgb[i * 3 + 0] *= s;
            d_rgb[i * 3 + 1] *= s;
            d_rgb[i * 3 + 2] *= s;
        }
        if (d_thermal) d_thermal[i] *= s;
    }
}

// ------------------------------------------------------------------------------------------------------
// NS CameraOptimizer (mode "SO3xR3").apply_to_raybundle, the first statement of the training forward
// [REF thermal_nerf_model.py:218-219]:  M_c = exp_map_SO3xR3(pose_adjustment[c]);  o' = o + M_c[:, 3];  d' = M_c[:, :3] d.
//   theta = sqrt(max(|w|^2, 1e-4)),  a = sin(theta)/theta,  b = (1 - cos(theta))/theta^2,  R = I + a K(w) + b K(w)^2
// applied to a vector without forming R:  R d = d + a (w x d) + b (w x (w x d)).  As ~60 tiny torch launches (slice assignments,
// two bmm's over [R,3,3], their autograd twins) this cost more device time than the whole proposal pass of a step.
// ------------------------------------------------------------------------------------------------------
struct Pose {
    float t[3], w[3], a, b, theta;
    bool clamped;
};
__device__ __forceinline__ Pose pose_load(const float *__restrict__ pose, long long c) {
    Pose p;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p.t[k] = pose[c * 6 + k];
        p.w[k] = pose[c * 6 + 3 + k];
    }
    const float n2 = p.w[0] * p.w[0] + p.w[1] * p.w[1] + p.w[2] * p.w[2];
    p.clamped = !(n2 >= 1e-4f);  // torch.clamp passes the gradient where the input is >= min
    p.theta = sqrtf(fmaxf(n2, 1e-4f));
    const float inv = 1.0f / p.theta;
    p.a = inv * sinf(p.theta);
    p.b = inv * inv * (1.0f - cosf(p.theta));
    return p;
}
__device__ __forceinline__ void cross3(const float *u, const float *v, float *o) {
    o[0] = u[1] * v[2] - u[2] * v[1];
    o[1] = u[2] * v[0] - u[0] * v[2];
    o[2] = u[0] * v[1] - u[1] * v[0];
}

__global__ void __launch_bounds__(kBlock)
camera_opt_fwd_kernel(const float *__restrict__ pose, const long long *__restrict__ cam, int num_cameras,
                      const float *__restrict__ origins, const float *__restrict__ dirs, long long n, float *__restrict__ out_o,
                      float *__restrict__ out_d) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const long long c = cam[i];
        if (c < 0 || c >= num_cameras) {
            // a stale / eval-split index: torch's index_select would raise; no table row is read and the ray is poisoned so that the
            // step's loss turns NaN instead of training silently on another memory location
#pragma unroll
            for (int k = 0; k < 3; ++k) out_o[i * 3 + k] = out_d[i * 3 + k] = __int_as_float(0x7fc00000);
            continue;
        }
        const Pose p = pose_load(pose, c);
        const float d[3] = {dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]};
        float u[3], v[3];
        cross3(p.w, d, u);
        cross3(p.w, u, v);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            out_o[i * 3 + k] = origins[i * 3 + k] + p.t[k];
            out_d[i * 3 + k] = d[k] + p.a * u[k] + p.b * v[k];
        }
    }
}

// backward: per ray the 6 components of d loss / d pose_adjustment[c]; lanes of a wave that share a camera are summed first
// (a loop over the distinct cameras of the wave), so one atomic per (wave, camera, component) reaches the [C, 6] gradient.
//   d/dt = g_o;   d/dw = a (d x g) + b (g (w.d) + d (g.w) - 2 (g.d) w) + (a'(theta) g.(w x d) + b'(theta) g.(w x (w x d))) w / theta
// with the last term only where |w|^2 >= 1e-4 (inside the clamp theta is a constant).  Optional d_dirs_in = R^T g.
__global__ void __launch_bounds__(kBlock)
camera_opt_bwd_kernel(const float *__restrict__ pose, const long long *__restrict__ cam, int num_cameras,
                      const float *__restrict__ dirs, const float *__restrict__ g_o, const float *__restrict__ g_d, long long n,
                      float *__restrict__ d_pose, float *__restrict__ d_dirs_in) {
    const long long stride = (long long)gridDim.x * kBlock;
    const long long rounds = (n + stride - 1) / stride;
    for (long long rd = 0; rd < rounds; ++rd) {
        const long long i = rd * stride + (long long)blockIdx.x * kBlock + threadIdx.x;
        bool live = i < n;
        long long c = -1;
        float q[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (live) {
            c = cam[i];
            if (c < 0 || c >= num_cameras) {  // out of the table (the forward poisoned this ray): no read, no atomic
                live = false;
                c = -1;
            }
        }
        if (live) {
            const Pose p = pose_load(pose, c);
            const float d[3] = {dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]};
            float g[3] = {0.0f, 0.0f, 0.0f};
            if (g_d) {
                g[0] = g_d[i * 3]; g[1] = g_d[i * 3 + 1]; g[2] = g_d[i * 3 + 2];
            }
            if (g_o) {
                q[0] = g_o[i * 3]; q[1] = g_o[i * 3 + 1]; q[2] = g_o[i * 3 + 2];
            }
            float u[3], v[3], dxg[3];
            cross3(p.w, d, u);
            cross3(p.w, u, v);
            cross3(d, g, dxg);
            const float wd = p.w[0] * d[0] + p.w[1] * d[1] + p.w[2] * d[2];
            const float gw = g[0] * p.w[0] + g[1] * p.w[1] + g[2] * p.w[2];
            const float gd = g[0] * d[0] + g[1] * d[1] + g[2] * d[2];
            float radial = 0.0f;
            if (!p.clamped) {
                const float th = p.theta, gu = g[0] * u[0] + g[1] * u[1] + g[2] * u[2], gv = g[0] * v[0] + g[1] * v[1] + g[2] * v[2];
                float da, db;
                if (th < 0.1f) {  // series: the closed forms cancel catastrophically in fp32 for small angles
                    const float t2 = th * th;
                    da = th * (-1.0f / 3.0f + t2 * (1.0f / 30.0f - t2 / 840.0f));
                    db = th * (-1.0f / 12.0f + t2 * (1.0f / 180.0f - t2 / 6720.0f));
                } else {
                    const float sn = sinf(th), cs = cosf(th);
                    da = (th * cs - sn) / (th * th);
                    db = (th * sn - 2.0f * (1.0f - cs)) / (th * th * th);
                }
                radial = (da * gu + db * gv) / th;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                q[3 + k] = p.a * dxg[k] + p.b * (g[k] * wd + d[k] * gw - 2.0f * gd * p.w[k]) + radial * p.w[k];
            if (d_dirs_in) {  // R^T g = g - a (w x g) + b (w x (w x g))
                float wg[3], wwg[3];
                cross3(p.w, g, wg);
                cross3(p.w, wg, wwg);
#pragma unroll
                for (int k = 0; k < 3; ++k) d_dirs_in[i * 3 + k] = g[k] - p.a * wg[k] + p.b * wwg[k];
            }
        }
        // distinct cameras of this wave, one at a time (a training batch draws its rays from all images: up to 64 rounds)
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(live);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const long long cc = __shfl(c, leader, 64);
            const bool mine = live && c == cc;
            float r[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) r[k] = wave_sum(mine ? q[k] : 0.0f);
            if (lane == leader) {
#pragma unroll
                for (int k = 0; k < 6; ++k) atomic_add_f32(d_pose + cc * 6 + k, r[k]);
            }
            todo &= ~__ballot(mine);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// get_weights backward: one wave per ray, chunks of 64 samples walked back to front.
//   a_i = delta_i sigma_i,  T_i = exp(-sum_{j<i} a_j),  w_i = (1 - e^{-a_i}) T_i
//   dL/da_k = g_k e^{-a_k} T_k - sum_{i>k} g_i w_i
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_incl_scan_rev(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_down(v, o, 64);
        if (lane + o < 64) v += t;
    }
    return v;
}

constexpr int kMaxChunks = 16;  // n <= 1024

__global__ void __launch_bounds__(kBlock)
weights_bwd_kernel(const float *__restrict__ deltas, const float *__restrict__ dens, const float *__restrict__ gw,
                   long long R, int n, float *__restrict__ gd) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float *dl = deltas + ray * n, *dn = dens + ray * n, *g = gw + ray * n;
    float *out = gd + ray * n;
    const int chunks = (n + 63) / 64;
    float csum[kMaxChunks];
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        csum[c] = 0.0f;
        if (c < chunks) {
            const int i = c * 64 + lane;
            csum[c] = wave_sum(i < n ? mul_rn(dl[i], dn[i]) : 0.0f);
        }
    }
    float suffix = 0.0f;  // sum of g_i w_i over the chunks behind this one
#pragma unroll
    for (int c = kMaxChunks - 1; c >= 0; --c) {
        if (c >= chunks) continue;
        float before = 0.0f;
#pragma unroll
        for (int q = 0; q < kMaxChunks; ++q)
            if (q < c) before += csum[q];
        const int i = c * 64 + lane;
        const bool live = i < n;
        const float a = live ? mul_rn(dl[i], dn[i]) : 0.0f;
        const float incl = wave_incl_scan(a, lane);
        const float excl = wave_excl_from_incl(incl, lane) + before;
        const float T = expf(-excl), ea = expf(-a);
        const float w = (1.0f - ea) * T;
        const float gi = live ? g[i] : 0.0f;
        const float gwv = live ? gi * w : 0.0f;
        const float rincl = wave_incl_scan_rev(gwv, lane);
        const float behind = rincl - gwv + suffix;  // sum_{i > k}
        if (live) out[i] = dl[i] * (gi * ea * T - behind);
        suffix += lane_value<0>(rincl);
    }
}

// ------------------------------------------------------------------------------------------------------
// The final level's per-ray renderers of a training step and their adjoints, one launch each way (round 3: they were
// tn_weights_fwd + 2 x tn_composite_fwd, and g_w clone / += + 2 x tn_composite_bwd + tn_weights_bwd + tn_gradient_scale_bwd).
// One wave per ray.  forward: w = get_weights(deltas, density) [REF thermal_nerf_model.py:233], rgb / thermal = sum w v + v_last
// (1 - sum w) [REF :237, :271-273; thermal_renderer.py:55-79 with "last_sample"], accumulation = sum w.
// backward: d_w[i] = d_w_ext[i] + d_acc + sum_c d_rgb[c] (rgb_i[c] - rgb_last[c]) + d_th (th_i - th_last);
//           d_rgb_s[i] = d_rgb (w_i + [i = last] (1 - acc)), likewise thermal;  d_density = adjoint of get_weights applied to d_w;
//           optional scale_gradients_by_distance_squared [REF :228-231] on the three per-sample gradients.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
ray_render_fwd_kernel(const float *__restrict__ deltas, const float *__restrict__ dens, const float *__restrict__ rgb_s,
                      const float *__restrict__ th_s, long long R, int n, float *__restrict__ weights, float *__restrict__ rgb,
                      float *__restrict__ thermal, float *__restrict__ acc) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float *dl = deltas + ray * n, *dn = dens + ray * n;
    float carry = 0.0f, sw = 0.0f, s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, st = 0.0f;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool live = i < n;
        const float a = live ? mul_rn(dl[i], dn[i]) : 0.0f;
        const float incl = wave_incl_scan(a, lane);
        const float excl = wave_excl_from_incl(incl, lane) + carry;
        carry += lane_value<63>(incl);
        const float w = live ? nan_to_num(mul_rn(sub_rn(1.0f, expf(-a)), expf(-excl))) : 0.0f;
        if (live) {
            const long long t = ray * n + i;
            weights[t] = w;
            sw += w;
            s0 += mul_rn(w, rgb_s[t * 3]);
            s1 += mul_rn(w, rgb_s[t * 3 + 1]);
            s2 += mul_rn(w, rgb_s[t * 3 + 2]);
            st += mul_rn(w, th_s[t]);
        }
    }
    sw = wave_sum(sw); s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); st = wave_sum(st);
    if (lane == 0) {
        const long long last = ray * n + n - 1;
        const float bg = sub_rn(1.0f, sw);
        rgb[ray * 3] = add_rn(s0, mul_rn(rgb_s[last * 3], bg));
        rgb[ray * 3 + 1] = add_rn(s1, mul_rn(rgb_s[last * 3 + 1], bg));
        rgb[ray * 3 + 2] = add_rn(s2, mul_rn(rgb_s[last * 3 + 2], bg));
        thermal[ray] = add_rn(st, mul_rn(th_s[last], bg));
        acc[ray] = sw;
    }
}

// ray_render_fwd_kernel + the level's depth renderers in the same pass over the ray (round 6): the weights are in registers when the
// median / expected depths need them, so tn_depth_fwd's three launches (bounds reset, depth_kernel, clip) shrink to the clip.
// Same operations in the same order as depth_kernel on the stored weights (bit-equal outputs: tests/test_gpu_training.py compares
// the step calls with the per-call path).  DepthRenderer("expected") clips to the call's [min, max] sample mid-point: every block
// leaves its rays' bounds in block_bounds[blockIdx] (no atomics, nothing to reset) and ray_depth_clip_kernel reduces them.
__global__ void __launch_bounds__(kBlock)
ray_render_depth_fwd_kernel(const float *__restrict__ deltas, const float *__restrict__ dens, const float *__restrict__ rgb_s,
                            const float *__restrict__ th_s, const float *__restrict__ starts, const float *__restrict__ ends, long long R,
                            int n, float *__restrict__ weights, float *__restrict__ rgb, float *__restrict__ thermal,
                            float *__restrict__ acc, float *__restrict__ median, float *__restrict__ expected,
                            float2 *__restrict__ block_bounds) {
    __shared__ float red[2][kBlock / 64];
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    float smin = INFINITY, smax = -INFINITY;
    if (ray < R) {
        const float *dl = deltas + ray * n, *dn = dens + ray * n, *st = starts + ray * n, *en = ends + ray * n;
        float carry = 0.0f, sw = 0.0f, s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, sth = 0.0f;
        float wcarry = 0.0f, wsteps = 0.0f;
        int med_idx = n;  // first index with cumsum(w) >= 0.5 (searchsorted side="left")
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            const bool live = i < n;
            const float a = live ? mul_rn(dl[i], dn[i]) : 0.0f;
            const float incl = wave_incl_scan(a, lane);
            const float excl = wave_excl_from_incl(incl, lane) + carry;
            carry += lane_value<63>(incl);
            const float w = live ? nan_to_num(mul_rn(sub_rn(1.0f, expf(-a)), expf(-excl))) : 0.0f;
            const float step = live ? add_rn(st[i], en[i]) / 2.0f : 0.0f;
            const float wincl = wave_incl_scan(w, lane) + wcarry;
            const unsigned long long hit = __ballot(live && (wincl >= 0.5f));
            if (hit && med_idx == n) med_idx = base + __ffsll((long long)hit) - 1;
            wcarry = lane_value<63>(wincl);
            wsteps += mul_rn(w, step);
            if (live) {
                const long long t = ray * n + i;
                weights[t] = w;
                sw += w;
                s0 += mul_rn(w, rgb_s[t * 3]);
                s1 += mul_rn(w, rgb_s[t * 3 + 1]);
                s2 += mul_rn(w, rgb_s[t * 3 + 2]);
                sth += mul_rn(w, th_s[t]);
                smin = fminf(smin, step);
                smax = fmaxf(smax, step);
            }
        }
        sw = wave_sum(sw); s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); sth = wave_sum(sth);
        wsteps = wave_sum(wsteps);
        if (lane == 0) {
            const long long last = ray * n + n - 1;
            const float bg = sub_rn(1.0f, sw);
            rgb[ray * 3] = add_rn(s0, mul_rn(rgb_s[last * 3], bg));
            rgb[ray * 3 + 1] = add_rn(s1, mul_rn(rgb_s[last * 3 + 1], bg));
            rgb[ray * 3 + 2] = add_rn(s2, mul_rn(rgb_s[last * 3 + 2], bg));
            thermal[ray] = add_rn(sth, mul_rn(th_s[last], bg));
            acc[ray] = sw;
            const int idx = min(med_idx, n - 1);
            median[ray] = add_rn(st[idx], en[idx]) / 2.0f;
            expected[ray] = wsteps / add_rn(sw, 1e-10f);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        smin = fminf(smin, __shfl_xor(smin, o, 64));
        smax = fmaxf(smax, __shfl_xor(smax, o, 64));
    }
    if (lane == 0) {
        red[0][threadIdx.x >> 6] = smin;
        red[1][threadIdx.x >> 6] = smax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int wv = 1; wv < kBlock / 64; ++wv) {
            smin = fminf(smin, red[0][wv]);
            smax = fmaxf(smax, red[1][wv]);
        }
        block_bounds[blockIdx.x] = make_float2(smin, smax);
    }
}

// expected depth clipped to the call's [min, max] mid-point: every block reduces the (few thousand) per-block bounds itself
__global__ void __launch_bounds__(kBlock)
ray_depth_clip_kernel(float *__restrict__ expected, long long R, const float2 *__restrict__ block_bounds, int nblocks) {
    __shared__ float red[2][kBlock / 64];
    float lo = INFINITY, hi = -INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        const float2 v = block_bounds[b];
        lo = fminf(lo, v.x);
        hi = fmaxf(hi, v.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = lo;
        red[1][threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    lo = red[0][0];
    hi = red[1][0];
#pragma unroll
    for (int wv = 1; wv < kBlock / 64; ++wv) {
        lo = fminf(lo, red[0][wv]);
        hi = fmaxf(hi, red[1][wv]);
    }
    const long long r = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (r < R && lo <= hi) expected[r] = fminf(fmaxf(expected[r], lo), hi);
}

__global__ void __launch_bounds__(kBlock)
ray_render_bwd_kernel(const float *__restrict__ deltas, const float *__restrict__ dens, const float *__restrict__ rgb_s,
                      const float *__restrict__ th_s, const float *__restrict__ acc, const float *__restrict__ g_rgb,
                      const float *__restrict__ g_th, const float *__restrict__ g_acc, const float *__restrict__ g_w_ext,
                      const float *__restrict__ starts, const float *__restrict__ ends, long long R, int n,
                      float *__restrict__ d_rgb_s, float *__restrict__ d_th_s, float *__restrict__ d_dens) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float *dl = deltas + ray * n, *dn = dens + ray * n;
    const long long last = ray * n + n - 1;
    const float gr0 = g_rgb ? g_rgb[ray * 3] : 0.0f, gr1 = g_rgb ? g_rgb[ray * 3 + 1] : 0.0f, gr2 = g_rgb ? g_rgb[ray * 3 + 2] : 0.0f;
    const float gt = g_th ? g_th[ray] : 0.0f, ga = g_acc ? g_acc[ray] : 0.0f;
    const float l0 = g_rgb ? rgb_s[last * 3] : 0.0f, l1 = g_rgb ? rgb_s[last * 3 + 1] : 0.0f, l2 = g_rgb ? rgb_s[last * 3 + 2] : 0.0f;
    const float lt = g_th ? th_s[last] : 0.0f;
    const float bg = 1.0f - acc[ray];
    const int chunks = (n + 63) / 64;
    float csum[kMaxChunks];
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        csum[c] = 0.0f;
        if (c < chunks) {
            const int i = c * 64 + lane;
            csum[c] = wave_sum(i < n ? mul_rn(dl[i], dn[i]) : 0.0f);
        }
    }
    float suffix = 0.0f;  // sum of d_w_i w_i over the chunks behind this one
#pragma unroll
    for (int c = kMaxChunks - 1; c >= 0; --c) {
        if (c >= chunks) continue;
        float before = 0.0f;
#pragma unroll
        for (int q = 0; q < kMaxChunks; ++q)
            if (q < c) before += csum[q];
        const int i = c * 64 + lane;
        const bool live = i < n;
        const long long t = ray * n + (live ? i : n - 1);
        const float a = live ? mul_rn(dl[i], dn[i]) : 0.0f;
        const float incl = wave_incl_scan(a, lane);
        const float excl = wave_excl_from_incl(incl, lane) + before;
        const float T = expf(-excl), ea = expf(-a);
        const float w = (1.0f - ea) * T;
        float scale = 1.0f;
        if (starts) {
            const float dist = (starts[t] + ends[t]) / 2.0f;
            scale = fminf(fmaxf(dist * dist, 0.0f), 1.0f);
        }
        float dwi = (g_w_ext ? g_w_ext[t] : 0.0f) + ga;
        const float wv = w + (i == n - 1 ? bg : 0.0f);
        if (g_rgb) {
            const float v0 = rgb_s[t * 3], v1 = rgb_s[t * 3 + 1], v2 = rgb_s[t * 3 + 2];
            dwi += gr0 * (v0 - l0) + gr1 * (v1 - l1) + gr2 * (v2 - l2);
            if (live) {
                d_rgb_s[t * 3] = gr0 * wv * scale;
                d_rgb_s[t * 3 + 1] = gr1 * wv * scale;
                d_rgb_s[t * 3 + 2] = gr2 * wv * scale;
            }
        }
        if (g_th) {
            dwi += gt * (th_s[t] - lt);
            if (live) d_th_s[t] = gt * wv * scale;
        }
        const float gi = live ? dwi : 0.0f;
        const float gwv = live ? gi * w : 0.0f;
        const float rincl = wave_incl_scan_rev(gwv, lane);
        const float behind = rincl - gwv + suffix;  // sum_{k > i}
        if (live) d_dens[t] = dl[i] * (gi * ea * T - behind) * scale;
        suffix += lane_value<0>(rincl);
    }
}

// ------------------------------------------------------------------------------------------------------
// compositing backward (training mode):  out_c = sum_i w_i v_ic + v_last,c (1 - acc)
// ------------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(kBlock)
composite_bwd_kernel(const float *__restrict__ v, const float *__restrict__ w, const float *__restrict__ acc,
                     const float *__restrict__ go, long long R, int n, float *__restrict__ gv, float *__restrict__ gw) {
    const long long total = R * n;
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < total; t += (long long)gridDim.x * kBlock) {
        const long long r = t / n;
        const int i = (int)(t - r * n);
        const float wi = w[t];
        const float bg = 1.0f - acc[r];
        float dwi = 0.0f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float g = go[r * C + c];
            gv[t * C + c] = g * (i == n - 1 ? wi + bg : wi);
            dwi += g * (v[t * C + c] - v[(r * n + n - 1) * C + c]);
        }
        gw[t] += dwi;
    }
}

// ------------------------------------------------------------------------------------------------------
// mlp_head input
// ------------------------------------------------------------------------------------------------------
struct CinArgs {
    const float *appearance;
    int num_images, app_dim, geo_dim, use_avg, sh_shifted, training;
};

// one thread per (sample, 4-float column group): coalesced 16-byte stores of the [N,64] rows
__global__ void __launch_bounds__(kBlock)
color_input_fwd_kernel(CinArgs a, const float *__restrict__ dirs, const float *__restrict__ geo, int ld_geo,
                       const int *__restrict__ cams, long long R, int n, float *__restrict__ cin) {
    const long long total = R * n * 16;
    const int g0 = 16, a0 = 16 + a.geo_dim, end = a0 + a.app_dim;
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < total; t += (long long)gridDim.x * kBlock) {
        const long long smp = t >> 4;
        const int c0 = (int)(t & 15) * 4;
        const long long r = smp / n;
        float v[4];
        if (c0 < 16) {
            float dx = dirs[r * 3], dy = dirs[r * 3 + 1], dz = dirs[r * 3 + 2];
            if (a.sh_shifted) {
                dx = add_rn(dx, 1.0f) / 2.0f; dy = add_rn(dy, 1.0f) / 2.0f; dz = add_rn(dz, 1.0f) / 2.0f;
            }
            float c[16];
            sh16(dx, dy, dz, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x = c[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) x = (c0 + q == k) ? c[k] : x;
                v[q] = x;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = c0 + q;
                float x = 0.0f;
                if (k >= g0 && k < a0) {
                    x = geo[smp * ld_geo + (k - g0)];
                } else if (k >= a0 && k < end) {
                    const int j = k - a0;
                    if (a.training) {
                        x = a.appearance[(size_t)cams[r] * a.app_dim + j];
                    } else if (a.use_avg) {
                        float sum = 0.0f;
                        for (int im = 0; im < a.num_images; ++im) sum += a.appearance[(size_t)im * a.app_dim + j];
                        x = sum / (float)a.num_images;
                    }
                }
                v[q] = x;
            }
        }
        *reinterpret_cast<float4 *>(cin + smp * 64 + c0) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// geo part: one thread per (sample, geo component)
__global__ void __launch_bounds__(kBlock)
color_input_bwd_geo_kernel(int geo_dim, const float *__restrict__ d_cin, long long N, float *__restrict__ d_geo, int ldg) {
    const long long total = N * geo_dim;
    for (long long t = (long long)blockIdx.x * kBlock + threadIdx.x; t < total; t += (long long)gridDim.x * kBlock) {
        const long long i = t / geo_dim;
        const int k = (int)(t - i * geo_dim);
        d_geo[i * ldg + k] += d_cin[i * 64 + 16 + k];
    }
}

// appearance part: one wave per ray sums its samples first, then app_dim atomics per ray
__global__ void __launch_bounds__(kBlock)
color_input_bwd_app_kernel(int geo_dim, int app_dim, const float *__restrict__ d_cin, const int *__restrict__ cams,
                           long long R, int n, float *__restrict__ d_app) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    const int k = lane & 31, half = lane >> 5;
    float s = 0.0f;
    if (k < app_dim)
        for (int i = half; i < n; i += 2) s += d_cin[(ray * n + i) * 64 + 16 + geo_dim + k];
    s += __shfl_xor(s, 32, 64);
    if (half == 0 && k < app_dim) atomic_add_f32(d_app + (size_t)cams[ray] * app_dim + k, s);
}

// direction part: d_dir[r] += J_SH(dir)^T sum_i d_cin[r, i, 0:16]  (x 1/2 when the SH input is (d + 1) / 2)
__global__ void __launch_bounds__(kBlock)
color_input_bwd_dir_kernel(int sh_shifted, const float *__restrict__ d_cin, const float *__restrict__ dirs, long long R,
                           int n, float *__restrict__ d_dirs) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    const int k = lane & 15, q = lane >> 4;  // component, sample stream
    float s = 0.0f;
    for (int i = q; i < n; i += 4) s += d_cin[(ray * n + i) * 64 + k];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    float g[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) g[c] = __shfl(s, c, 64);
    if (lane != 0) return;
    float x = dirs[ray * 3], y = dirs[ray * 3 + 1], z = dirs[ray * 3 + 2];
    if (sh_shifted) {
        x = add_rn(x, 1.0f) / 2.0f; y = add_rn(y, 1.0f) / 2.0f; z = add_rn(z, 1.0f) / 2.0f;
    }
    const float xx = x * x, yy = y * y, zz = z * z;
    const float a1 = 0.4886025119029199f, a4 = 1.0925484305920792f, a6 = 0.9461746957575601f, a8 = 0.5462742152960396f;
    const float a9 = 0.5900435899266435f, a10 = 2.890611442640554f, a11 = 0.4570457994644658f, a12 = 0.3731763325901154f;
    const float a14 = 1.445305721320277f;
    float dx = g[3] * a1 + g[4] * a4 * y + g[7] * a4 * z + g[8] * 2.0f * a8 * x + g[9] * 6.0f * a9 * x * y + g[10] * a10 * y * z +
               g[13] * a11 * (5.0f * zz - 1.0f) + g[14] * 2.0f * a14 * x * z + g[15] * a9 * (3.0f * xx - 3.0f * yy);
    float dy = g[1] * a1 + g[4] * a4 * x + g[5] * a4 * z - g[8] * 2.0f * a8 * y + g[9] * a9 * (3.0f * xx - 3.0f * yy) +
               g[10] * a10 * x * z + g[11] * a11 * (5.0f * zz - 1.0f) - g[14] * 2.0f * a14 * y * z - g[15] * 6.0f * a9 * x * y;
    float dz = g[2] * a1 + g[5] * a4 * y + g[6] * 2.0f * a6 * z + g[7] * a4 * x + g[10] * a10 * x * y + g[11] * 10.0f * a11 * y * z +
               g[12] * a12 * (15.0f * zz - 3.0f) + g[13] * 10.0f * a11 * x * z + g[14] * a14 * (xx - yy);
    const float f = sh_shifted ? 0.5f : 1.0f;
    d_dirs[ray * 3] += f * dx;
    d_dirs[ray * 3 + 1] += f * dy;
    d_dirs[ray * 3 + 2] += f * dz;
}

// ------------------------------------------------------------------------------------------------------
// distortion loss: one wave per ray.  With sorted mid-points u:
//   S_i = sum_j w_j |u_i - u_j| = u_i (W_<i - W_>i) - (WU_<i - WU_>i)
//   loss = sum_i w_i S_i + sum_i w_i^2 d_i / 3,   dloss/dw_i = 2 S_i + 2 w_i d_i / 3
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
distortion_kernel(const float *__restrict__ bins, const float *__restrict__ weights, long long R, int n, float scale,
                  float mult, bool pair, float *__restrict__ loss_sum, float *__restrict__ gw) {
    // pair: loss_sum[0] is the metric (scale * sum), loss_sum[1] the loss term (mult * that) and gw the gradient of the TERM —
    // what get_metrics_dict / get_loss_dict make of it [REF thermal_nerf_model.py:301-304] without elementwise launches between
    const float gscale = scale * mult;
    // a wave walks several rays and a block adds ONE value to loss_sum: atomics on a single address retire one at a time in
    // L2 (~12 ns each), so one per ray made this kernel 50 us of waiting for 4096 rays
    __shared__ float red[kBlock / 64];
    const int lane = threadIdx.x & 63;
    const int chunks = (n + 63) / 64;
    float loss_acc = 0.0f;
    for (long long ray = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); ray < R; ray += (long long)gridDim.x * (kBlock / 64)) {
    const float *t = bins + ray * (n + 1), *w = weights + ray * n;
    float W = 0.0f, WU = 0.0f;
    for (int c = 0; c < chunks; ++c) {
        const int i = c * 64 + lane;
        const float wi = i < n ? w[i] : 0.0f;
        const float ui = i < n ? (t[i + 1] + t[i]) / 2.0f : 0.0f;
        W += wave_sum(wi);
        WU += wave_sum(wi * ui);
    }
    float cW = 0.0f, cWU = 0.0f, loss = 0.0f;
    for (int c = 0; c < chunks; ++c) {
        const int i = c * 64 + lane;
        const bool live = i < n;
        const float wi = live ? w[i] : 0.0f;
        const float t0 = live ? t[i] : 0.0f, t1 = live ? t[i + 1] : 0.0f;
        const float ui = (t1 + t0) / 2.0f, di = t1 - t0;
        const float iw = wave_incl_scan(wi, lane), iwu = wave_incl_scan(wi * ui, lane);
        const float Wlt = cW + iw - wi, WUlt = cWU + iwu - wi * ui;
        const float Wgt = W - Wlt - wi, WUgt = WU - WUlt - wi * ui;
        const float Si = ui * (Wlt - Wgt) - (WUlt - WUgt);
        if (live) {
            gw[ray * n + i] = gscale * (2.0f * Si + 2.0f * wi * di / 3.0f);
            loss += wi * Si + wi * wi * di / 3.0f;
        }
        cW += lane_value<63>(iw);
        cWU += lane_value<63>(iwu);
    }
    loss_acc += loss;
    }
    loss_acc = wave_sum(loss_acc);
    if (lane == 0) red[threadIdx.x >> 6] = loss_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float block = (red[0] + red[1]) + (red[2] + red[3]);
        atomic_add_f32(loss_sum, scale * block);
        if (pair) atomic_add_f32(loss_sum + 1, gscale * block);
    }
}

// ------------------------------------------------------------------------------------------------------
// interlevel loss (one proposal level): one wave per ray, LDS per wave: cp[p+1], cum[p+1], d_hi[p], d_lo[p+1]
// ------------------------------------------------------------------------------------------------------
constexpr int kMaxP = 1024;

__device__ __forceinline__ int upper_bound(const float *a, int len, float x) {  // first index with a[idx] > x
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] > x) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// all proposal levels of the loss in one launch: blockIdx.y = level (the levels are independent and each is latency-bound)
constexpr int kMaxInterlevels = 4;
struct InterlevelArgs {
    const float *c, *w;
    const float *cp[kMaxInterlevels], *wp[kMaxInterlevels];
    float *g_wp[kMaxInterlevels];
    int p[kMaxInterlevels];
    long long R;
    int n;
    float scale;
    float *loss_sum;
};

__global__ void __launch_bounds__(kBlock) interlevel_kernel(InterlevelArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[kBlock / 64];  // one atomic on loss_sum per block (see distortion_kernel)
    const float *__restrict__ c = a.c, *__restrict__ w = a.w;
    const float *__restrict__ cp = a.cp[blockIdx.y], *__restrict__ wp = a.wp[blockIdx.y];
    float *__restrict__ g_wp = a.g_wp[blockIdx.y], *__restrict__ loss_sum = a.loss_sum;
    const long long R = a.R;
    const int n = a.n, p = a.p[blockIdx.y];
    const float scale = a.scale;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss_acc = 0.0f;
    for (long long ray = (long long)blockIdx.x * (kBlock / 64) + wave; ray < R; ray += (long long)gridDim.x * (kBlock / 64)) {
    const int stride = 4 * (p + 2);
    float *edges = lds + wave * stride;  // [p+1]
    float *cum = edges + (p + 2);        // [p+1] exclusive cumsum of wp (cum[p] = total)
    float *dhi = cum + (p + 2);          // [p]   + coef at hi
    float *dlo = dhi + (p + 2);          // [p+1] + coef at lo
    const float *cpr = cp + ray * (p + 1), *wpr = wp + ray * p;
    float carry = 0.0f;
    for (int b = 0; b <= p; b += 64) {
        const int k = b + lane;
        if (k <= p) edges[k] = cpr[k];
        const float v = k < p ? wpr[k] : 0.0f;
        const float incl = wave_incl_scan(v, lane);
        if (k <= p) cum[k] = carry + incl - v;
        if (k <= p) { dlo[k] = 0.0f; if (k < p) dhi[k] = 0.0f; }
        carry += lane_value<63>(incl);
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const float *cr = c + ray * (n + 1), *wr = w + ray * n;
    float loss = 0.0f;
    for (int i = lane; i < n; i += 64) {
        int lo = upper_bound(edges, p, cr[i]) - 1;          // searchsorted(starts = cp[:-1], right) - 1
        lo = min(max(lo, 0), p - 1);
        int hi = upper_bound(edges + 1, p, cr[i + 1]);       // searchsorted(ends = cp[1:], right)
        hi = min(max(hi, 0), p - 1);
        const float w_outer = cum[hi + 1] - cum[lo];
        const float wi = wr[i];
        const float d = fmaxf(wi - w_outer, 0.0f);
        const float den = wi + 1.0e-7f;
        loss += d * d / den;
        const float coef = -2.0f * d / den;  // d loss / d w_outer
        if (coef != 0.0f) {
            atomicAdd(&dhi[hi], coef);  // + coef on every k <= hi
            atomicAdd(&dlo[lo], coef);  // - coef on every k <  lo
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // grad_k = sum_{h >= k} dhi[h] - sum_{l > k} dlo[l]; walk the chunks back to front
    float s_hi = 0.0f, s_lo = 0.0f;
    for (int b = ((p - 1) / 64) * 64; b >= 0; b -= 64) {
        const int k = b + lane;
        const float vh = k < p ? dhi[k] : 0.0f;
        const float vl = k < p ? dlo[k + 1] : 0.0f;  // shifted by one: sum_{j >= k} dlo[j + 1] = sum_{l > k} dlo[l]
        const float rh = wave_incl_scan_rev(vh, lane);
        const float rl = wave_incl_scan_rev(vl, lane);
        if (k < p) g_wp[ray * p + k] = scale * ((s_hi + rh) - (s_lo + rl));
        s_hi += lane_value<0>(rh);
        s_lo += lane_value<0>(rl);
    }
    loss_acc += loss;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();  // the next ray of this wave reuses the LDS arrays
    }
    loss_acc = wave_sum(loss_acc);
    if (lane == 0) red[wave] = loss_acc;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add_f32(loss_sum, scale * ((red[0] + red[1]) + (red[2] + red[3])));
}


// ------------------------------------------------------------------------------------------------------
// the two image losses of get_loss_dict [REF thermal_nerf_model.py:294-295, 319-323] and the PSNR metric of get_metrics_dict in
// ONE launch (as torch ops: two MSE kernels, two mean reductions, six elementwise launches for the PSNR and two more kernels
// in the backward — a tenth of a 2 ms step in launches alone).  One block: a training batch is a few thousand pixels.
// out[0] = mean((gt_rgb - rgb)^2), out[1] = mean((thermal - gt_thermal)^2), out[2] = 10 log10(1 / out[0]);
// d_rgb = 2 (rgb - gt_rgb) / (3 R), d_thermal = 2 (thermal - gt_thermal) / R  (the gradients of the two means).
// ------------------------------------------------------------------------------------------------------
constexpr int kLossBlock = 1024;
__global__ void __launch_bounds__(kLossBlock)
image_losses_kernel(const float *__restrict__ rgb, const float *__restrict__ gt_rgb, const float *__restrict__ th,
                    const float *__restrict__ gt_th, long long R, float *__restrict__ out, float *__restrict__ d_rgb,
                    float *__restrict__ d_th) {
    __shared__ float red[2][kLossBlock / 64];
    const long long n3 = R * 3;
    const float k3 = 2.0f / (float)n3, k1 = 2.0f / (float)R;
    float s3 = 0.0f, s1 = 0.0f;
    for (long long i = threadIdx.x; i < n3; i += kLossBlock) {
        const float e = rgb[i] - gt_rgb[i];
        s3 = fmaf(e, e, s3);
        d_rgb[i] = k3 * e;
    }
    if (th) {
        for (long long i = threadIdx.x; i < R; i += kLossBlock) {
            const float e = th[i] - gt_th[i];
            s1 = fmaf(e, e, s1);
            d_th[i] = k1 * e;
        }
    }
    s3 = wave_sum(s3);
    s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = s3;
        red[1][wave] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, b = 0.0f;
        for (int w = 0; w < kLossBlock / 64; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        const float mse = a / (float)n3;
        out[0] = mse;
        out[1] = b / (float)R;
        out[2] = 10.0f * log10f(1.0f / mse);
    }
}

// the step's total loss: NS Trainer.train_iteration reduces the loss dictionary with torch.add, left to right — one launch here
struct ScalarList {
    const float *p[TN_SUM_MAX_TERMS];
    int count;
};
__global__ void sum_scalars_kernel(ScalarList L, float *__restrict__ out) {
    float s = L.p[0][0];
    for (int k = 1; k < L.count; ++k) s = add_rn(s, L.p[k][0]);
    out[0] = s;
}

}  // namespace

extern "C" {

int tn_hash_encode_fwd(const tn_hashgrid *grid, const tn_space *space, const float *positions, int64_t n, float *enc,
                       float *selector, void *stream) {
    if (!grid || !space) return TN_ERR_NULL;
    TN_TRY(tn_check_grid(*grid));
    if (n == 0) return TN_OK;
    if (!positions || !enc) return TN_ERR_NULL;
    if (n < 0) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(hash_encode_fwd_kernel, dim3(grid_for(n * grid->num_levels, kBlock, 1 << 16)), dim3(kBlock), 0,
                       (hipStream_t)stream, tn_make_grid(*grid), *space, positions, (long long)n, enc, selector);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// the spread layout of a grid: the leading levels whose dense vertex grid is small (<= kSpreadMaxSide per axis)
constexpr int kSpreadCopies = 16, kSpreadMaxSide = 48;
static Spread spread_layout(const tn_hashgrid &h, int level_begin, int level_end) {
    Spread s{};
    s.copies = kSpreadCopies;
    if (level_begin != 0) return s;  // only when the call starts at the coarsest level
    unsigned off = 0;
    for (int l = 0; l < level_end && l < TN_MAX_LEVELS; ++l) {
        const int side = (int)h.scalings[l] + 2;  // coordinates 0 .. ceil(scaling)
        if (side > kSpreadMaxSide) break;
        s.off[l] = off;
        s.side[l] = side;
        off += (unsigned)side * side * side;
        s.levels = l + 1;
    }
    s.per_copy = off;
    return s;
}

static int hash_encode_bwd_levels(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                                  int64_t n, float *d_table, int level_begin, int level_end, void *stream, void *spread_ws = nullptr,
                                  size_t spread_bytes = 0) {
    if (!grid || !space) return TN_ERR_NULL;
    TN_TRY(tn_check_grid(*grid));
    if (level_begin < 0 || level_end > grid->num_levels || level_begin > level_end) return TN_ERR_SHAPE;
    if (n == 0 || level_begin == level_end) return TN_OK;
    if (!positions || !d_enc || !d_table) return TN_ERR_NULL;
    if (n < 0) return TN_ERR_SHAPE;
    Spread spr{};
    spr.copies = 1;
    if (spread_ws) {
        spr = spread_layout(*grid, level_begin, level_end);
        const size_t need = (size_t)spr.copies * spr.per_copy * 2 * sizeof(float);
        if (spr.levels > 0 && spread_bytes < need) return TN_ERR_WORKSPACE;
        if (spr.levels > 0) {
            spr.buf = reinterpret_cast<float *>(spread_ws);
            if (hipMemsetAsync(spread_ws, 0, need, (hipStream_t)stream) != hipSuccess) return TN_ERR_LAUNCH;
        }
    }
    const Grid g = tn_make_grid(*grid);
    hipLaunchKernelGGL(hash_encode_bwd_kernel, dim3(grid_for(((n + 63) / 64) * (level_end - level_begin), kBlock / 64, 1 << 16)),
                       dim3(kBlock), 0, (hipStream_t)stream, g, *space, positions, d_enc, (long long)n, d_table, level_begin,
                       level_end, spr);
    TN_LAUNCH_CHECK();
    if (spr.levels > 0) {
        hipLaunchKernelGGL(spread_reduce_kernel, dim3(grid_for(spr.per_copy, kBlock, 1024)), dim3(kBlock), 0, (hipStream_t)stream, g,
                           spr, d_table);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}

int tn_hash_encode_bwd(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                       int64_t n, float *d_table, void *stream) {
    if (!grid) return TN_ERR_NULL;
    return hash_encode_bwd_levels(grid, space, positions, d_enc, n, d_table, 0, grid->num_levels, stream);
}

int tn_hash_encode_bwd_levels(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                              int64_t n, float *d_table, int32_t level_begin, int32_t level_end, void *stream) {
    return hash_encode_bwd_levels(grid, space, positions, d_enc, n, d_table, level_begin, level_end, stream);
}

size_t tn_hash_encode_bwd_spread_workspace_bytes(const tn_hashgrid *grid) {
    if (!grid || tn_check_grid(*grid) != TN_OK) return 0;
    const Spread s = spread_layout(*grid, 0, grid->num_levels);
    return (size_t)s.copies * s.per_copy * 2 * sizeof(float);
}

int tn_hash_encode_bwd_spread(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                              int64_t n, float *d_table, int32_t level_begin, int32_t level_end, void *workspace,
                              size_t workspace_bytes, void *stream) {
    return hash_encode_bwd_levels(grid, space, positions, d_enc, n, d_table, level_begin, level_end, stream, workspace,
                                  workspace_bytes);
}

size_t tn_hash_encode_bwd_sorted_workspace_bytes(const tn_hashgrid *grid, int64_t n, int32_t level_begin) {
    SortLayout w;
    if (!grid || n <= 0 || tn_check_grid(*grid) != TN_OK || !sort_layout(*grid, n, level_begin, w)) return 0;
    return w.bytes;
}

// Where the bucketed form pays: the first level from which it should take over (the atomic kernel keeps the levels below), or
// -1 = nowhere.  Measured on a real step's inputs (tools/scatter_bench.py --real --levels): at the coarse levels the samples
// of a ray — and of its neighbours — crowd into few entries and the owner pass's compare-and-swap adds retry (40-90 us per
// level against 20-30 for the atomics); from a scaling of ~256 up the bucketed form costs 16-20 us per level against 29-33.
// With the run-writing emit pass (round 3c) the step is shortest with the split one level lower, at a scaling of 200 = level 8
// of the reference grid: 2.55 against 2.63 ms per step at S=192 (levels 7 / 6 / 5: 2.68 / 2.64 / 2.70, and 3.1 - 3.7 ms on a
// single-view batch, whose samples crowd), within noise at S = 48 ... 96; 8 levels x 32 slices = one bin per CU.  And one block
// per bin owns a CU's LDS: fewer than 128 bins (the proposal grids: 5 levels x 8 slices) leave the chip idle behind a few long
// buckets.
int tn_hash_encode_bwd_sorted_first_level(const tn_hashgrid *grid, int64_t n) {
    if (!grid || n <= 0 || tn_check_grid(*grid) != TN_OK) return -1;
    int first = 0;
    while (first < grid->num_levels && grid->scalings[first] < kSortMinScaling) ++first;
    SortLayout w;
    if (first >= grid->num_levels || !sort_layout(*grid, n, first, w)) return -1;
    return w.bins >= kSortMinBins ? first : -1;
}

// phase 1 = the emit pass (clears the cursors, writes the records), 2 = the owner pass (sums them into d_table), 3 = both
static int hash_encode_bwd_sorted_phases(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                                         int64_t n, float *d_table, int32_t level_begin, void *workspace, size_t workspace_bytes,
                                         int phases, void *stream) {
    if (!grid || !space) return TN_ERR_NULL;
    TN_TRY(tn_check_grid(*grid));
    if (n == 0) return TN_OK;
    if (!positions || !d_enc || !d_table || !workspace) return TN_ERR_NULL;
    if (n < 0) return TN_ERR_SHAPE;
    SortLayout w;
    if (!sort_layout(*grid, n, level_begin, w)) return TN_ERR_UNSUPPORTED;
    if (workspace_bytes < w.bytes) return TN_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return TN_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    char *ws = reinterpret_cast<char *>(workspace);
    SortArgs a;
    a.g = tn_make_grid(*grid);
    a.space = *space;
    a.positions = positions; a.d_enc = d_enc; a.n = n;
    a.owners = w.owners; a.slice_log2 = w.slice_log2;
    a.level_begin = level_begin; a.levels = grid->num_levels - level_begin;
    a.cursors = reinterpret_cast<unsigned *>(ws);
    a.capacity = w.capacity;
    a.rec_pair = reinterpret_cast<unsigned *>(ws + w.off_pair);
    a.rec_val = reinterpret_cast<float4 *>(ws + w.off_val);
    a.d_table = d_table;
    if (phases & 1) {
        if (hipMemsetAsync(ws, 0, (size_t)w.bins * 4, s) != hipSuccess) return TN_ERR_LAUNCH;
        const long long chunks = (n + kSortSamples - 1) / kSortSamples;
        const long long blocks = ((chunks + 7) / 8) * 8 * a.levels;
        if (blocks > 0x7fffffffLL) return TN_ERR_SHAPE;
        const size_t emit_smem = (size_t)kSortSamples * 4 * 22 + (size_t)3 * 4 * kBlock * ((w.owners + kBlock - 1) / kBlock);
        if (emit_smem > 64 * 1024 && !tn_ensure_dynamic_lds<sort_emit_kernel>(emit_smem)) return TN_ERR_LAUNCH;
        hipLaunchKernelGGL(sort_emit_kernel, dim3((unsigned)blocks), dim3(kBlock), emit_smem, s, a);
        TN_LAUNCH_CHECK();
    }
    if (phases & 2) {
        const size_t smem = (size_t)(2 << w.slice_log2) * sizeof(float);
        if (smem > 64 * 1024 && !tn_ensure_dynamic_lds<sort_owner_kernel>(smem)) return TN_ERR_LAUNCH;
        hipLaunchKernelGGL(sort_owner_kernel, dim3((unsigned)w.bins), dim3(kOwnerBlock), smem, s, a.cursors, w.capacity, a.rec_pair,
                           a.rec_val, w.owners, w.slice_log2, 1u << grid->log2_hashmap_size, level_begin, d_table);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}

int tn_hash_encode_bwd_sorted(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                              int64_t n, float *d_table, int32_t level_begin, void *workspace, size_t workspace_bytes,
                              void *stream) {
    return hash_encode_bwd_sorted_phases(grid, space, positions, d_enc, n, d_table, level_begin, workspace, workspace_bytes, 3, stream);
}

int tn_hash_encode_bwd_input(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                             int64_t n, float *d_positions, void *stream) {
    if (!grid || !space) return TN_ERR_NULL;
    TN_TRY(tn_check_grid(*grid));
    if (n == 0) return TN_OK;
    if (!positions || !d_enc || !d_positions) return TN_ERR_NULL;
    if (n < 0) return TN_ERR_SHAPE;
    if (grid->num_levels > 8)
        hipLaunchKernelGGL(hash_encode_bwd_input_kernel<16>, dim3(grid_for(n * 16, kBlock, 1 << 16)), dim3(kBlock), 0, (hipStream_t)stream,
                       tn_make_grid(*grid), *space, positions, d_enc, (long long)n, d_positions);
    else
        hipLaunchKernelGGL(hash_encode_bwd_input_kernel<8>, dim3(grid_for(n * 8, kBlock, 1 << 16)), dim3(kBlock), 0, (hipStream_t)stream,
                       tn_make_grid(*grid), *space, positions, d_enc, (long long)n, d_positions);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_frustum_positions_bwd(const float *d_positions, const float *starts, const float *ends, int64_t num_rays, int32_t n,
                             float *d_origins, float *d_directions, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!d_positions || !starts || !ends || !d_origins || !d_directions) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(frustum_positions_bwd_kernel, dim3((unsigned)((num_rays + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream,
                       d_positions, starts, ends, (long long)num_rays, n, d_origins, d_directions);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_linear_fwd(const float *x, int32_t ldx, const tn_linear *lin, int32_t act, int64_t n, float *y, int32_t ldy,
                  void *stream) {
    if (!lin || !lin->weight) return TN_ERR_NULL;
    const int IN = lin->in_dim, OUT = lin->out_dim;
    if (IN < 1 || IN > kLinWide || OUT < 1 || OUT > kLinWide || ldx < IN || ldy < OUT || n < 0) return TN_ERR_SHAPE;
    if (act < TN_ACT_NONE || act > TN_ACT_SIGMOID) return TN_ERR_UNSUPPORTED;
    if (n == 0) return TN_OK;
    if (!x || !y) return TN_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    if (IN > 64 || OUT > 64) {
        const size_t smem = (size_t)TILE * (IN | 1) * sizeof(float);
        if (!tn_ensure_dynamic_lds<linear_fwd_wide_kernel>(smem)) return TN_ERR_LAUNCH;
        const int vec = (ldy % 4 == 0) && (OUT % 4 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
        hipLaunchKernelGGL(linear_fwd_wide_kernel, dim3(grid_for((n + TILE - 1) / TILE, 1, 1024)), dim3(kBlock), smem, st, x, ldx,
                           lin->weight, lin->bias, IN, OUT, act, (long long)n, y, ldy, vec);
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    const int INP = IN <= 16 ? 16 : IN <= 32 ? 32 : 64;
    const int vec_in = (ldx % 4 == 0) && (ldx >= INP) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    const int vec_out = (ldy % 4 == 0) && (OUT % 4 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
    const dim3 g(grid_for(n, kBlock, 1 << 16)), b(kBlock);
    if (INP == 16)
        hipLaunchKernelGGL(linear_fwd_rows_kernel<16>, g, b, 0, st, x, ldx, lin->weight, lin->bias, IN, OUT, act, (long long)n, y,
                           ldy, vec_in, vec_out);
    else if (INP == 32)
        hipLaunchKernelGGL(linear_fwd_rows_kernel<32>, g, b, 0, st, x, ldx, lin->weight, lin->bias, IN, OUT, act, (long long)n, y,
                           ldy, vec_in, vec_out);
    else
        hipLaunchKernelGGL(linear_fwd_rows_kernel<64>, g, b, 0, st, x, ldx, lin->weight, lin->bias, IN, OUT, act, (long long)n, y,
                           ldy, vec_in, vec_out);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

size_t tn_linear_bwd_workspace_bytes(void) { return (size_t)kLinBwdBlocks * 65 * 64 * sizeof(float); }

int tn_linear_bwd(const float *x, int32_t ldx, const float *y, const float *dy, int32_t ldy, const tn_linear *lin,
                  int32_t act, int64_t n, float *dx, int32_t lddx, int32_t accumulate_dx, float *d_weight,
                  float *d_bias, void *workspace, size_t workspace_bytes, void *stream) {
    if (!lin || !lin->weight) return TN_ERR_NULL;
    const int IN = lin->in_dim, OUT = lin->out_dim;
    if (IN < 1 || IN > kLinWide || OUT < 1 || OUT > kLinWide || ldx < IN || ldy < OUT || n < 0) return TN_ERR_SHAPE;
    if (dx && lddx < IN) return TN_ERR_SHAPE;
    if (act < TN_ACT_NONE || act > TN_ACT_SIGMOID) return TN_ERR_UNSUPPORTED;
    if (n == 0) return TN_OK;
    if (!dy || (act != TN_ACT_NONE && !y) || ((d_weight || d_bias) && !x)) return TN_ERR_NULL;
    // persistent blocks; their partial weight gradients go to the workspace and are summed by a second kernel
    // (without a workspace: one memory-side atomic per block and weight entry)
    const int blocks = grid_for((n + TILE - 1) / TILE, 1, kLinBwdBlocks);
    const dim3 g(blocks), b(kBlock);
    hipStream_t st = (hipStream_t)stream;
    const int INP = IN <= 16 ? 16 : IN <= 32 ? 32 : 64;
    const bool want_w = d_weight || d_bias;
    float *partials = nullptr;
    if (want_w && workspace && workspace_bytes >= (size_t)blocks * 65 * 64 * sizeof(float))
        partials = reinterpret_cast<float *>(workspace);
    const bool wide = IN > 64 || OUT > 64;
    if (wide && want_w && !partials) return TN_ERR_WORKSPACE;
    if (!want_w || partials) {  // matrix-pipe form (needs the workspace for its partial weight gradients)
        auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
        // a layer wider than 64 goes through the same kernel one 64 x 64 block of its weight matrix at a time (launches on one
        // stream: the partial-sum slabs are reused): block (ob, kb) takes the columns 64 ob .. of g = dy . act'(y) to
        // dx[:, 64 kb ..] (+= behind the first ob), to dW[64 ob .., 64 kb ..] and — once per ob — to db[64 ob ..]
        for (int ob = 0; ob < OUT; ob += 64) {
            for (int kb = 0; kb < IN; kb += 64) {
                const int OUTb = OUT - ob < 64 ? OUT - ob : 64, INb = IN - kb < 64 ? IN - kb : 64;
                const float *xb = x ? x + kb : nullptr, *yb = y ? y + ob : nullptr, *dyb = dy + ob;
                float *dxb = dx ? dx + kb : nullptr;
                const int acc = accumulate_dx || ob > 0;
                const int vec = (ldy % 4 == 0) && (OUTb % 4 == 0) && al16(dyb) && (act == TN_ACT_NONE || al16(yb)) &&
                                (!want_w || ((ldx % 4 == 0) && (INb % 4 == 0) && al16(xb)));
                const int vec_dx = dxb && !acc && (lddx % 4 == 0) && (INb % 4 == 0) && al16(dxb);
                hipLaunchKernelGGL(linear_bwd_mfma_kernel, g, b, 0, st, xb, ldx, yb, dyb, ldy, lin->weight + (size_t)ob * IN + kb, IN,
                                   INb, OUTb, act, (long long)n, dxb, lddx, acc, want_w ? 1 : 0, partials, vec, vec_dx);
                TN_LAUNCH_CHECK();
                if (want_w) {
                    hipLaunchKernelGGL(linear_bwd_reduce_kernel, dim3(65, kRedSplit), dim3(kBlock), 0, st, partials, blocks, 64, INb,
                                       OUTb, d_weight ? d_weight + (size_t)ob * IN + kb : nullptr, IN,
                                       (d_bias && kb == 0) ? d_bias + ob : nullptr);
                    TN_LAUNCH_CHECK();
                }
            }
        }
        return TN_OK;
    }
    if (IN <= 16)
        hipLaunchKernelGGL(linear_bwd_kernel<4>, g, b, 0, st, x, ldx, y, dy, ldy, lin->weight, IN, OUT, act, (long long)n, dx,
                           lddx, accumulate_dx, d_weight, d_bias, partials);
    else if (IN <= 32)
        hipLaunchKernelGGL(linear_bwd_kernel<8>, g, b, 0, st, x, ldx, y, dy, ldy, lin->weight, IN, OUT, act, (long long)n, dx,
                           lddx, accumulate_dx, d_weight, d_bias, partials);
    else
        hipLaunchKernelGGL(linear_bwd_kernel<16>, g, b, 0, st, x, ldx, y, dy, ldy, lin->weight, IN, OUT, act, (long long)n, dx,
                           lddx, accumulate_dx, d_weight, d_bias, partials);
    TN_LAUNCH_CHECK();
    if (partials) {
        hipLaunchKernelGGL(linear_bwd_reduce_kernel, dim3(INP + 1, kRedSplit), dim3(kBlock), 0, st, partials, blocks, INP, IN, OUT,
                           d_weight, IN, d_bias);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}

size_t tn_linear_chain_bwd_workspace_bytes(void) { return (size_t)kChainMax * kChainBlocks * 65 * 64 * sizeof(float); }

int tn_linear_chain_bwd(const tn_chain_layer *layers, int32_t num_layers, const float *y_top, int32_t act_top, const float *dy,
                        int32_t lddy, int64_t n, float *dx, int32_t lddx, int32_t accumulate_dx, void *workspace,
                        size_t workspace_bytes, void *stream) {
    if (!layers || !dy) return TN_ERR_NULL;
    if (num_layers < 1 || num_layers > kChainMax || n < 0) return TN_ERR_SHAPE;
    if (act_top < TN_ACT_NONE || act_top > TN_ACT_SIGMOID) return TN_ERR_UNSUPPORTED;
    if (act_top != TN_ACT_NONE && !y_top) return TN_ERR_NULL;
    if (n == 0) return TN_OK;
    if (!workspace || workspace_bytes < tn_linear_chain_bwd_workspace_bytes()) return TN_ERR_WORKSPACE;
    auto al16 = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    ChainArgs a;
    a.nl = num_layers;
    for (int j = 0; j < num_layers; ++j) {
        const tn_chain_layer &l = layers[j];
        if (!l.lin.weight || !l.x) return TN_ERR_NULL;
        const int IN = l.lin.in_dim, OUT = l.lin.out_dim;
        if (IN < 1 || IN > 64 || OUT < 1 || OUT > 64 || l.ldx < IN) return TN_ERR_SHAPE;
        if (j > 0 && layers[j - 1].lin.in_dim != OUT) return TN_ERR_SHAPE;  // x_{j-1} is this layer's output
        if (l.act_x < TN_ACT_NONE || l.act_x > TN_ACT_SIGMOID) return TN_ERR_UNSUPPORTED;
        a.L[j].W = l.lin.weight; a.L[j].x = l.x; a.L[j].IN = IN; a.L[j].OUT = OUT; a.L[j].ldx = l.ldx; a.L[j].act_x = l.act_x;
        a.L[j].vec_x = (l.ldx % 4 == 0) && (IN % 4 == 0) && al16(l.x);
    }
    const int OUT0 = layers[0].lin.out_dim, INL = layers[num_layers - 1].lin.in_dim;
    if (lddy < OUT0 || (dx && lddx < INL)) return TN_ERR_SHAPE;
    a.dy = dy; a.y_top = y_top; a.lddy = lddy; a.act_top = act_top;
    a.vec_top = (lddy % 4 == 0) && (OUT0 % 4 == 0) && al16(dy) && (act_top == TN_ACT_NONE || al16(y_top));
    a.n = n; a.dx = dx; a.lddx = lddx; a.accumulate_dx = accumulate_dx;
    a.vec_dx = dx && !accumulate_dx && (lddx % 4 == 0) && (INL % 4 == 0) && al16(dx);
    a.partials = reinterpret_cast<float *>(workspace);
    const int blocks = grid_for((n + TILE - 1) / TILE, 1, kChainBlocks);
    hipStream_t st = (hipStream_t)stream;
    size_t wrows = 0;
    for (int j = 0; j < num_layers; ++j) wrows += (size_t)((layers[j].lin.out_dim + 1) & ~1);
    const size_t smem = (wrows * 64 + 2 * TILE * LDP) * sizeof(float);
    if (num_layers == 1) {
        if (!tn_ensure_dynamic_lds<linear_chain_bwd_kernel<1>>(smem)) return TN_ERR_LAUNCH;
        hipLaunchKernelGGL(linear_chain_bwd_kernel<1>, dim3(blocks), dim3(kBlock), smem, st, a);
    } else if (num_layers == 2) {
        if (!tn_ensure_dynamic_lds<linear_chain_bwd_kernel<2>>(smem)) return TN_ERR_LAUNCH;
        hipLaunchKernelGGL(linear_chain_bwd_kernel<2>, dim3(blocks), dim3(kBlock), smem, st, a);
    } else {
        if (!tn_ensure_dynamic_lds<linear_chain_bwd_kernel<3>>(smem)) return TN_ERR_LAUNCH;
        hipLaunchKernelGGL(linear_chain_bwd_kernel<3>, dim3(blocks), dim3(kBlock), smem, st, a);
    }
    TN_LAUNCH_CHECK();
    RedChainArgs ra;
    bool any = false;
    for (int j = 0; j < kChainMax; ++j) {
        const bool on = j < num_layers;
        ra.partials[j] = a.partials + (size_t)j * blocks * 65 * 64;
        ra.dW[j] = on ? layers[j].d_weight : nullptr;
        ra.db[j] = on ? layers[j].d_bias : nullptr;
        ra.IN[j] = on ? layers[j].lin.in_dim : 0;
        ra.OUT[j] = on ? layers[j].lin.out_dim : 0;
        any = any || ra.dW[j] || ra.db[j];
    }
    if (any) {
        hipLaunchKernelGGL(linear_bwd_reduce_chain_kernel, dim3(65, kRedSplit, num_layers), dim3(kBlock), 0, st, ra, blocks);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}

int tn_density_act_fwd(const float *raw, int32_t ld_raw, const float *selector, float average_init_density, int64_t n,
                       float *density, void *stream) {
    if (n == 0) return TN_OK;
    if (!raw || !selector || !density) return TN_ERR_NULL;
    if (n < 0 || ld_raw < 1) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(density_act_fwd_kernel, dim3(grid_for(n, kBlock, 1 << 16)), dim3(kBlock), 0, (hipStream_t)stream, raw,
                       ld_raw, selector, average_init_density, (long long)n, density);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_density_act_bwd(const float *raw, int32_t ld_raw, const float *selector, float average_init_density,
                       float trunc_exp_min, const float *d_density, int64_t n, float *d_raw, int32_t ld_d_raw,
                       int32_t clear_cols, void *stream) {
    if (n == 0) return TN_OK;
    if (!raw || !selector || !d_density || !d_raw) return TN_ERR_NULL;
    if (n < 0 || ld_raw < 1 || ld_d_raw < 1 || clear_cols < 0 || clear_cols > ld_d_raw) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(density_act_bwd_kernel, dim3(grid_for(n, kBlock, 1 << 16)), dim3(kBlock), 0, (hipStream_t)stream, raw,
                       ld_raw, selector, average_init_density, trunc_exp_min, d_density, (long long)n, d_raw, ld_d_raw, clear_cols);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_gradient_scale_bwd(const float *starts, const float *ends, int64_t n, float *d_density, float *d_rgb,
                          float *d_thermal, void *stream) {
    if (n == 0) return TN_OK;
    if (!starts || !ends) return TN_ERR_NULL;
    if (n < 0) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(gradient_scale_kernel, dim3(grid_for(n, kBlock, 1 << 16)), dim3(kBlock), 0, (hipStream_t)stream, starts,
                       ends, (long long)n, d_density, d_rgb, d_thermal);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_camera_opt_fwd(const float *pose_adjustment, const int64_t *camera_indices, const float *origins,
                      const float *directions, int64_t num_rays, int32_t num_cameras, float *out_origins,
                      float *out_directions, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!pose_adjustment || !camera_indices || !origins || !directions || !out_origins || !out_directions) return TN_ERR_NULL;
    if (num_rays < 0 || num_cameras < 1) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(camera_opt_fwd_kernel, dim3(grid_for(num_rays, kBlock, 1 << 16)), dim3(kBlock), 0, (hipStream_t)stream,
                       pose_adjustment, reinterpret_cast<const long long *>(camera_indices), num_cameras, origins, directions,
                       (long long)num_rays, out_origins, out_directions);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_camera_opt_bwd(const float *pose_adjustment, const int64_t *camera_indices, const float *directions,
                      const float *d_out_origins, const float *d_out_directions, int64_t num_rays, int32_t num_cameras,
                      float *d_pose_adjustment, float *d_directions, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!pose_adjustment || !camera_indices || !directions || !d_pose_adjustment) return TN_ERR_NULL;
    if (num_rays < 0 || num_cameras < 1) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(camera_opt_bwd_kernel, dim3(grid_for(num_rays, kBlock, 1 << 16)), dim3(kBlock), 0, (hipStream_t)stream,
                       pose_adjustment, reinterpret_cast<const long long *>(camera_indices), num_cameras, directions, d_out_origins,
                       d_out_directions, (long long)num_rays, d_pose_adjustment, d_directions);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_weights_bwd(const float *deltas, const float *densities, const float *d_weights, int64_t num_rays, int32_t n,
                   float *d_densities, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!deltas || !densities || !d_weights || !d_densities) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1 || n > 64 * kMaxChunks) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(weights_bwd_kernel, dim3((unsigned)((num_rays + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, deltas,
                       densities, d_weights, (long long)num_rays, n, d_densities);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_composite_bwd(const float *values, const float *weights, const float *accumulation, const float *d_out,
                     int64_t num_rays, int32_t n, int32_t channels, float *d_values, float *d_weights, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!values || !weights || !accumulation || !d_out || !d_values || !d_weights) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    const dim3 g(grid_for(num_rays * n, kBlock, 1 << 16)), b(kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (channels == 3)
        hipLaunchKernelGGL(composite_bwd_kernel<3>, g, b, 0, st, values, weights, accumulation, d_out, (long long)num_rays, n,
                           d_values, d_weights);
    else if (channels == 1)
        hipLaunchKernelGGL(composite_bwd_kernel<1>, g, b, 0, st, values, weights, accumulation, d_out, (long long)num_rays, n,
                           d_values, d_weights);
    else
        return TN_ERR_UNSUPPORTED;
    TN_LAUNCH_CHECK();
    return TN_OK;
}

static int cin_args(const tn_thermal_field *f, int training, CinArgs *a) {
    if (!f) return TN_ERR_NULL;
    if (f->geo_feat_dim < 1 || f->app_dim < 0 || 16 + f->geo_feat_dim + f->app_dim > 64) return TN_ERR_SHAPE;
    if (f->app_dim > 32) return TN_ERR_UNSUPPORTED;
    if (f->app_dim > 0 && (training || f->use_average_appearance) && !f->appearance) return TN_ERR_NULL;
    a->appearance = f->appearance; a->num_images = f->num_images; a->app_dim = f->app_dim; a->geo_dim = f->geo_feat_dim;
    a->use_avg = f->use_average_appearance; a->sh_shifted = f->sh_shifted; a->training = training;
    return TN_OK;
}

int tn_color_input_fwd(const tn_thermal_field *field, const float *directions, const float *geo, int32_t ld_geo,
                       const int32_t *camera_indices, int32_t training, int64_t num_rays, int32_t n, float *cin,
                       void *stream) {
    CinArgs a;
    TN_TRY(cin_args(field, training, &a));
    if (num_rays == 0) return TN_OK;
    if (!directions || !geo || !cin || (training && !camera_indices)) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1 || ld_geo < a.geo_dim) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(color_input_fwd_kernel, dim3(grid_for(num_rays * n * 16, kBlock, 1 << 16)), dim3(kBlock), 0,
                       (hipStream_t)stream, a, directions, geo, ld_geo, camera_indices, (long long)num_rays, n, cin);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_color_input_bwd(const tn_thermal_field *field, const float *d_cin, const int32_t *camera_indices,
                       int32_t training, int64_t num_rays, int32_t n, float *d_geo, int32_t ld_d_geo,
                       float *d_appearance, const float *directions, float *d_directions, void *stream) {
    CinArgs a;
    TN_TRY(cin_args(field, training, &a));
    if (num_rays == 0) return TN_OK;
    if (!d_cin) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    if (d_geo) {
        if (ld_d_geo < a.geo_dim) return TN_ERR_SHAPE;
        hipLaunchKernelGGL(color_input_bwd_geo_kernel, dim3(grid_for(num_rays * n * a.geo_dim, kBlock, 1 << 16)), dim3(kBlock),
                           0, (hipStream_t)stream, a.geo_dim, d_cin, (long long)(num_rays * n), d_geo, ld_d_geo);
        TN_LAUNCH_CHECK();
    }
    if (d_appearance && training && a.app_dim > 0) {
        if (!camera_indices) return TN_ERR_NULL;
        hipLaunchKernelGGL(color_input_bwd_app_kernel, dim3((unsigned)((num_rays + 3) / 4)), dim3(kBlock), 0,
                           (hipStream_t)stream, a.geo_dim, a.app_dim, d_cin, camera_indices, (long long)num_rays, n,
                           d_appearance);
        TN_LAUNCH_CHECK();
    }
    if (d_directions) {
        if (!directions) return TN_ERR_NULL;
        hipLaunchKernelGGL(color_input_bwd_dir_kernel, dim3((unsigned)((num_rays + 3) / 4)), dim3(kBlock), 0,
                           (hipStream_t)stream, a.sh_shifted, d_cin, directions, (long long)num_rays, n, d_directions);
        TN_LAUNCH_CHECK();
    }
    return TN_OK;
}

int tn_distortion_loss(const float *spacing_bins, const float *weights, int64_t num_rays, int32_t n, float scale,
                       float *loss_sum, float *d_weights, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!spacing_bins || !weights || !loss_sum || !d_weights) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(distortion_kernel, dim3(grid_for(num_rays, kBlock / 64, 512)), dim3(kBlock), 0, (hipStream_t)stream,
                       spacing_bins, weights, (long long)num_rays, n, scale, 1.0f, false, loss_sum, d_weights);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_distortion_loss_term(const float *spacing_bins, const float *weights, int64_t num_rays, int32_t n, float scale,
                            float mult, float *loss_pair, float *d_weights, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!spacing_bins || !weights || !loss_pair || !d_weights) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(distortion_kernel, dim3(grid_for(num_rays, kBlock / 64, 512)), dim3(kBlock), 0, (hipStream_t)stream,
                       spacing_bins, weights, (long long)num_rays, n, scale, mult, true, loss_pair, d_weights);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_interlevel_loss_levels(const float *c, const float *w, int64_t num_rays, int32_t n, int32_t num_levels,
                              const float *const *cp, const float *const *wp, const int32_t *p, float scale, float *loss_sum,
                              float *const *d_wp, void *stream) {
    if (num_rays == 0 || num_levels == 0) return TN_OK;
    if (!c || !w || !cp || !wp || !p || !loss_sum || !d_wp) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1 || num_levels < 0 || num_levels > kMaxInterlevels) return TN_ERR_SHAPE;
    InterlevelArgs a{};
    a.c = c; a.w = w; a.R = (long long)num_rays; a.n = n; a.scale = scale; a.loss_sum = loss_sum;
    int pmax = 0;
    for (int l = 0; l < num_levels; ++l) {
        if (!cp[l] || !wp[l] || !d_wp[l]) return TN_ERR_NULL;
        if (p[l] < 1 || p[l] > kMaxP) return TN_ERR_SHAPE;
        a.cp[l] = cp[l]; a.wp[l] = wp[l]; a.g_wp[l] = d_wp[l]; a.p[l] = p[l];
        pmax = p[l] > pmax ? p[l] : pmax;
    }
    const size_t lds = (size_t)(kBlock / 64) * 4 * (pmax + 2) * sizeof(float);
    if (lds > 48 * 1024 && !tn_ensure_dynamic_lds<interlevel_kernel>(lds)) return TN_ERR_LAUNCH;
    hipLaunchKernelGGL(interlevel_kernel, dim3(grid_for(num_rays, kBlock / 64, 512), num_levels), dim3(kBlock), lds,
                       (hipStream_t)stream, a);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_interlevel_loss(const float *c, const float *w, const float *cp, const float *wp, int64_t num_rays, int32_t n,
                       int32_t p, float scale, float *loss_sum, float *d_wp, void *stream) {
    return tn_interlevel_loss_levels(c, w, num_rays, n, 1, &cp, &wp, &p, scale, loss_sum, &d_wp, stream);
}

int tn_sum_scalars(const float *const *terms, int32_t count, float *out, void *stream) {
    if (!terms || !out) return TN_ERR_NULL;
    if (count < 1 || count > TN_SUM_MAX_TERMS) return TN_ERR_SHAPE;
    ScalarList L;
    for (int k = 0; k < count; ++k) {
        if (!terms[k]) return TN_ERR_NULL;
        L.p[k] = terms[k];
    }
    L.count = count;
    hipLaunchKernelGGL(sum_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, L, out);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_image_losses(const float *rgb, const float *gt_rgb, const float *thermal, const float *gt_thermal, int64_t num_rays,
                    float *out, float *d_rgb, float *d_thermal, void *stream) {
    if (!rgb || !gt_rgb || !out || !d_rgb) return TN_ERR_NULL;
    if (thermal && (!gt_thermal || !d_thermal)) return TN_ERR_NULL;
    if (num_rays < 1) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(image_losses_kernel, dim3(1), dim3(kLossBlock), 0, (hipStream_t)stream, rgb, gt_rgb, thermal, gt_thermal,
                       (long long)num_rays, out, d_rgb, d_thermal);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_ray_render_fwd(const float *deltas, const float *densities, const float *rgb_samples, const float *thermal_samples,
                      int64_t num_rays, int32_t n, float *weights, float *rgb, float *thermal, float *accumulation, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!deltas || !densities || !rgb_samples || !thermal_samples || !weights || !rgb || !thermal || !accumulation) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(ray_render_fwd_kernel, dim3((unsigned)((num_rays + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, deltas,
                       densities, rgb_samples, thermal_samples, (long long)num_rays, n, weights, rgb, thermal, accumulation);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_ray_render_depth_fwd(const float *deltas, const float *densities, const float *rgb_samples, const float *thermal_samples,
                            const float *starts, const float *ends, int64_t num_rays, int32_t n, float *weights, float *rgb,
                            float *thermal, float *accumulation, float *median, float *expected, float *bounds_scratch, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!deltas || !densities || !rgb_samples || !thermal_samples || !starts || !ends || !weights || !rgb || !thermal || !accumulation ||
        !median || !expected || !bounds_scratch)
        return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    const unsigned blocks = (unsigned)((num_rays + 3) / 4);
    float2 *bounds = reinterpret_cast<float2 *>(bounds_scratch);
    hipLaunchKernelGGL(ray_render_depth_fwd_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, deltas, densities, rgb_samples,
                       thermal_samples, starts, ends, (long long)num_rays, n, weights, rgb, thermal, accumulation, median, expected, bounds);
    hipLaunchKernelGGL(ray_depth_clip_kernel, dim3((unsigned)((num_rays + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       expected, (long long)num_rays, bounds, (int)blocks);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_ray_render_bwd(const float *deltas, const float *densities, const float *rgb_samples, const float *thermal_samples,
                      const float *accumulation, const float *d_rgb, const float *d_thermal, const float *d_accumulation,
                      const float *d_weights, const float *starts, const float *ends, int64_t num_rays, int32_t n,
                      float *d_rgb_samples, float *d_thermal_samples, float *d_densities, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!deltas || !densities || !accumulation || !d_densities) return TN_ERR_NULL;
    if (d_rgb && (!rgb_samples || !d_rgb_samples)) return TN_ERR_NULL;
    if (d_thermal && (!thermal_samples || !d_thermal_samples)) return TN_ERR_NULL;
    if ((starts == nullptr) != (ends == nullptr)) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1 || n > 64 * kMaxChunks) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(ray_render_bwd_kernel, dim3((unsigned)((num_rays + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, deltas,
                       densities, rgb_samples, thermal_samples, accumulation, d_rgb, d_thermal, d_accumulation, d_weights, starts,
                       ends, (long long)num_rays, n, d_rgb_samples, d_thermal_samples, d_densities);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"


