// mcubes.hip -- marching-cubes iso-surface extraction (reference: extractIsoSurfaceKernel & helpers,
// src/cuda/marchingcube.cu:5-164; tables src/cuda/marchingcube_table.h; MarchingcubeData src/cuda/MarchingcubeData.h).
//
// The reference appends triangles with one global atomicAdd, so its output ORDER is nondeterministic (and its
// check-then-add can overshoot the buffer, marchingcube.cu:29-32).  Here extraction is deterministic and reads voxels only where
// the surface can be:
//   dilate : the bricks with a negative voxel in their 3x3x3 brick neighbourhood (brick flags only) -> a bit per brick and a list;
//   codes  : a 2-bit class per voxel (unobserved / not negative / negative) of the listed bricks, one coalesced pass over them;
//   sift   : a cell whose 27 voxels are all of one class -- or hold an unobserved one -- cannot produce a triangle: one lane
//            decides that for the eight cells of a brick x-row from 27 cached 16-bit loads -> a bit per cell, a bit per 256-cell block;
//   list   : the 256-cell blocks (z, y, x order) that hold a surviving cell, unordered;
//   count  : the surviving cells of the listed blocks, queued into dense waves, evaluated as the reference evaluates a cell
//            (eight trilinear corner lookups) -> per-block triangle count, and a record per cell that holds triangles;
//   scan   : exclusive prefix over ALL blocks in order (three parallel steps);
//   emit   : one lane per record: the cell again, its triangles land at block prefix + offset in the block: a fixed index.
// The canonical order is (z, y, x, k) -- identical for 1 GPU and for concatenated z-slabs.
// The triangle table is packed into 256 x 64-bit words (16 nibbles per case) instead of the reference's 16 KiB int table; the
// 12-bit edge mask is derived from it.
#include "kf_internal.h"

__constant__ unsigned long long c_tri_words[256] = {
#include "mc_tables.inc"
};

struct McArgs {
  KfVolume vol;
  int z0, z1;                    // cell layers processed (owned slab)
  int has_color;
  float thr;
  unsigned* block_counts;        // [n_blocks + 1]
  unsigned n_blocks;
  kf_triangle* tris; unsigned max_tris;
  KfCounters* cnt;
  int count_work;                // measurement passes: count the blocks that pass the neighbourhood test (kf_stage_timers bit 16)
  unsigned* nbr_bits;            // one bit per stored brick: some brick of its 3x3x3 neighbourhood holds a negative voxel
  unsigned* list;                // blocks that may hold surface, in no particular order (their output position comes from the scan)
  unsigned* n_list;              // length of `list` (device)
  unsigned* partials;            // per 4096-block chunk: sum of its block counts, then the exclusive prefix of those sums
  unsigned short* codes;         // [stored brick slot][brick row (z&7)<<3 | (y&7)]: eight voxels x 2 bits (k_mc_codes)
  unsigned char* surv;           // [stored brick slot][brick row]: bit i = cell 8 bx + i of that row survives the sieve (k_mc_sift)
  unsigned* block_bits;          // one bit per 256-cell block: some cell of it survives (set by k_mc_sift, listed by k_mc_list)
  unsigned* d1_list;             // stored brick slots with a negative voxel in their 3x3x3 brick neighbourhood, unordered (k_mc_dilate)
  unsigned* n_d1;
  uint2* recs;                   // cells with triangles: {block, lane | triangles << 8 | offset inside the block << 12} (k_mc_count)
  unsigned* n_recs;              // records appended (may exceed recs_cap: then *recs_overflow is set and k_mc_emit walks the blocks instead)
  unsigned* recs_overflow;
  unsigned recs_cap;
};
#define MC_CHUNK 4096u             // block counts scanned by one workgroup (16 per lane)

__device__ __forceinline__ float sel8(const float d[8], int k) {
  float r = d[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) r = (k == i) ? d[i] : r;
  return r;
}
__device__ __forceinline__ uchar4 sel8c(const uchar4 d[8], int k) {
  uchar4 r = d[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) r = (k == i) ? d[i] : r;
  return r;
}

// corner k (reference evaluation order 000,100,010,001,110,011,101,111 as x,y,z bits)
__device__ __forceinline__ int corner_bits(int k) {
  const int cb[8] = {0, 1, 2, 4, 3, 6, 5, 7};     // bit0 = x, bit1 = y, bit2 = z
  return cb[k];
}

struct CellEval { float d[8]; uchar4 c[8]; unsigned ci; unsigned long long word; int ntri; float3 wp; };

// does any brick in the +-2 voxel neighbourhood of (x,y,z) carry `mask`?
__device__ __forceinline__ bool neighbourhood_has(const KfVolume& v, int x, int y, int z, unsigned mask) {
  const int R = v.res;
  const int bx0 = max(x - 2, 0) >> 3, bx1 = min(x + 2, R - 1) >> 3;
  const int by0 = max(y - 2, 0) >> 3, by1 = min(y + 2, R - 1) >> 3;
  const int bz0 = max(max(z - 2, 0) >> 3, v.bz0), bz1 = min(min(z + 2, R - 1) >> 3, v.bz1 - 1);
  for (int bz = bz0; bz <= bz1; ++bz)
    for (int by = by0; by <= by1; ++by)
      for (int bx = bx0; bx <= bx1; ++bx)
        if (v.flags[kf_brick_slot(v, bx, by, bz)] & mask) return true;
  return false;
}

// extractIsoSurfaceAtPosition marchingcube.cu:41-113 up to the table lookup; returns the triangle count of the cell
// (the caller has already dropped the cells mc_pretest rules out)
__device__ __forceinline__ int eval_cell(const McArgs& a, int x, int y, int z, CellEval& e) {
  const KfVolume& v = a.vol;
  e.ntri = 0;
  const float cell = v.cell;
  e.wp = kf3(((float)x + 0.5f) * cell, ((float)y + 0.5f) * cell, ((float)z + 0.5f) * cell);     // tsdfVolume.h:38-49
  const float P = cell * 0.5f, M = cell * (-0.5f);
  // The eight corner lookups (tsdfVolume.h:98-122 + :151-172 each).  A lookup's voxel index and weight are computed per AXIS
  // from that axis' coordinate alone, and the corners' coordinates take only two values per axis (centre -/+ half a cell): six
  // axis evaluations instead of twenty-four, same arithmetic, same bits.  The voxel gathers then go out two corners at a time;
  // the reference's early returns are pure, so testing the corners in its order afterwards gives the same outcome.
  // (Staging the block's 258 x 3 x 3 voxel neighbourhood in LDS was tried: 0.96 -> 1.53 ms at 512^3.)
  const KfRecip rS = kf_recip(v.size), rcell = kf_recip(cell);
  struct Axis { bool ok; int g; float w; };
  const float rf = (float)v.res;
  const int R = v.res;
  auto axis = [&](float pos) {
    Axis r; r.w = 0.f;
    int g = kf_f2i(kf_div(pos * rf, rS));                                   // tsdfVolume.h:50-56
    r.ok = !(g <= 0 || g >= R - 1);                                         // :153-155
    g = (pos < ((float)g + 0.5f) * cell) ? (g - 1) : g;                     // :160-162
    r.g = g;
    r.w = kf_div(pos - ((float)g + 0.5f) * cell, rcell);                    // :164-166
    return r;
  };
  const Axis ax[2] = {axis(e.wp.x + M), axis(e.wp.x + P)}, ay[2] = {axis(e.wp.y + M), axis(e.wp.y + P)}, az[2] = {axis(e.wp.z + M), axis(e.wp.z + P)};
  auto corner = [&](int b) {
    const Axis &X = ax[b & 1], &Y = ay[(b >> 1) & 1], &Z = az[(b >> 2) & 1];
    KfInterp it; it.g = make_int3(X.g, Y.g, Z.g); it.a = X.w; it.b = Y.w; it.c = Z.w;
    it.ok = X.ok && Y.ok && Z.ok && kf_z_stored(v, Z.g) && kf_z_stored(v, Z.g + 1);
    return it;
  };
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const int b0 = corner_bits(k), b1 = corner_bits(k + 1);
    const KfInterp i0 = corner(b0), i1 = corner(b1);
    float2 q0[8], q1[8];
    kf_interp_load(v, i0, q0); kf_interp_load(v, i1, q1);          // (per-axis address shares kept in registers were tried: 0.89 -> 1.21 ms)
    e.d[k] = 0.f; e.d[k + 1] = 0.f;
    if (!kf_interp_finish(i0, q0, e.d[k])) return 0;
    e.c[k] = make_uchar4(0, 0, 0, 0);
    if (a.has_color) kf_interpolate_color(v, kf_add(e.wp, kf3((b0 & 1) ? P : M, (b0 & 2) ? P : M, (b0 & 4) ? P : M)), e.c[k]);
    if (!kf_interp_finish(i1, q1, e.d[k + 1])) return 0;
    e.c[k + 1] = make_uchar4(0, 0, 0, 0);
    if (a.has_color) kf_interpolate_color(v, kf_add(e.wp, kf3((b1 & 1) ? P : M, (b1 & 2) ? P : M, (b1 & 4) ? P : M)), e.c[k + 1]);
  }
  // :77-85  cube index bit order 010,110,100,000,011,111,101,001  (k: 0=000 1=100 2=010 3=001 4=110 5=011 6=101 7=111)
  unsigned ci = 0;
  if (e.d[2] < 0.f) ci += 1;
  if (e.d[4] < 0.f) ci += 2;
  if (e.d[1] < 0.f) ci += 4;
  if (e.d[0] < 0.f) ci += 8;
  if (e.d[5] < 0.f) ci += 16;
  if (e.d[7] < 0.f) ci += 32;
  if (e.d[6] < 0.f) ci += 64;
  if (e.d[3] < 0.f) ci += 128;
#pragma unroll
  for (int k = 0; k < 8; ++k) if (fabsf(e.d[k]) > a.thr) return 0;                               // :101-108
  const unsigned long long w = c_tri_words[ci];
  unsigned emask = 0; int n = 0;
#pragma unroll
  for (int i = 0; i < 15; ++i) { unsigned ed = (unsigned)((w >> (4 * i)) & 0xF); if (ed != 0xF) { emask |= 1u << ed; ++n; } }
  if (emask == 0 || emask == 255) return 0;                                                      // :110
  e.ci = ci; e.word = w; e.ntri = n / 3;
  return e.ntri;
}

// vertexInterp marchingcube.cu:5-26 for edge `ed` of the evaluated cell (edge -> corner pairs :116-127)
__device__ __forceinline__ kf_vertex edge_vertex(const McArgs& a, const CellEval& e, unsigned ed) {
  const int ea[12] = {2, 4, 1, 0, 5, 7, 6, 3, 2, 4, 1, 0};
  const int eb[12] = {4, 1, 0, 2, 7, 6, 3, 5, 5, 7, 6, 3};
  int ka = 0, kb = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) { ka = (ed == (unsigned)i) ? ea[i] : ka; kb = (ed == (unsigned)i) ? eb[i] : kb; }
  const float cell = a.vol.cell;
  const float P = cell * 0.5f, M = cell * (-0.5f);
  const int ba = corner_bits(ka), bb = corner_bits(kb);
  const float3 p1 = kf_add(e.wp, kf3((ba & 1) ? P : M, (ba & 2) ? P : M, (ba & 4) ? P : M));
  const float3 p2 = kf_add(e.wp, kf3((bb & 1) ? P : M, (bb & 2) ? P : M, (bb & 4) ? P : M));
  const float d1 = sel8(e.d, ka), d2 = sel8(e.d, kb);
  const uchar4 c1 = sel8c(e.c, ka), c2 = sel8c(e.c, kb);
  const float inv255 = (float)(1.0 / (double)255.f);          // `*(1.0/ 255.f)` : double quotient narrowed
  kf_vertex r;
  const float iso = 0.0f;
  const bool pick1 = fabsf(iso - d1) < 0.00001f;
  const bool pick2 = !pick1 && fabsf(iso - d2) < 0.00001f;
  const bool pick1b = !pick1 && !pick2 && fabsf(d1 - d2) < 0.00001f;
  if (pick1 || pick1b) {
    r.pos[0] = p1.x; r.pos[1] = p1.y; r.pos[2] = p1.z;
    r.color[0] = (float)c1.x * inv255; r.color[1] = (float)c1.y * inv255; r.color[2] = (float)c1.z * inv255;
  } else if (pick2) {
    r.pos[0] = p2.x; r.pos[1] = p2.y; r.pos[2] = p2.z;
    r.color[0] = (float)c2.x * inv255; r.color[1] = (float)c2.y * inv255; r.color[2] = (float)c2.z * inv255;
  } else {
    const float mu = (iso - d1) / (d2 - d1);
    r.pos[0] = p1.x + mu * (p2.x - p1.x); r.pos[1] = p1.y + mu * (p2.y - p1.y); r.pos[2] = p1.z + mu * (p2.z - p1.z);
    r.color[0] = ((float)c1.x + mu * (float)((int)c2.x - (int)c1.x)) / 255.f;
    r.color[1] = ((float)c1.y + mu * (float)((int)c2.y - (int)c1.y)) / 255.f;
    r.color[2] = ((float)c1.z + mu * (float)((int)c2.z - (int)c1.z)) / 255.f;
  }
  return r;
}

// ---- the voxel classes and the cell sieve --------------------------------------------------------------------------------------
// class of a voxel: 0 unobserved (weight == 0), 1 observed and not negative, 2 observed and tsdf in [-1e18, -1e-18], 3 any other
// observed negative.  Bricks outside the dilated has-negative set (k_mc_dilate) keep class 0 throughout without ever being read
// or written (the table is cleared when it is allocated and whenever the volume is reset or uploaded; between those the set only
// grows): a cell that touches such a brick has no negative voxel among its 27 (the bricks a cell touches are mutual neighbours),
// so "unobserved" and "no negative anywhere" lead to the same verdict below.
#define MC_NEG_LO (-1.0e18f)
#define MC_NEG_HI (-1.0e-18f)
__global__ void __launch_bounds__(256) k_mc_codes(McArgs a) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned n_d1 = *a.n_d1;
  for (unsigned i = blockIdx.x * 4u + (threadIdx.x >> 6); i < n_d1; i += gridDim.x * 4u) {                 // a wave per brick, a lane per x-row
    const unsigned slot = a.d1_list[i];
    unsigned code = 0;
    const float4* p = reinterpret_cast<const float4*>(a.vol.tw + (size_t)slot * KF_BRICK_VOX + lane * 8u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 q = p[k];                                                                                // two voxels: (tsdf, weight) x 2
      const unsigned c0 = q.y == 0.f ? 0u : !(q.x < 0.f) ? 1u : (q.x <= MC_NEG_HI && q.x >= MC_NEG_LO) ? 2u : 3u;
      const unsigned c1 = q.w == 0.f ? 0u : !(q.z < 0.f) ? 1u : (q.z <= MC_NEG_HI && q.z >= MC_NEG_LO) ? 2u : 3u;
      code |= (c0 | (c1 << 2)) << (4 * k);
    }
    a.codes[(size_t)slot * 64u + lane] = (unsigned short)code;
  }
}
// Which cells can produce a triangle at all?  A cell's eight corner lookups read exactly the voxels x-1..x+1, y-1..y+1, z-1..z+1
// (corner = cell centre -/+ half a cell; tsdfVolume.h:151-172 picks the voxel pair around it), each with weights in [0, 1] of
// which the larger per axis is >= 0.5, summed as eight products (:98-122).  Hence, exactly as the full evaluation would find:
//   an unobserved voxel among the 27        -> some corner lookup fails                      -> no triangle (marchingcube.cu:60-75)
//   all 27 observed and none negative       -> every corner sum >= 0: cube index 0           -> no triangle
//   all 27 in [-1e18, -1e-18]               -> every corner sum < 0 (one product <= -1e-18/8, no overflow, no NaN): index 255
// Every other cell survives and is evaluated in full; so is every cell on the volume's or the slab's rim (its 27 voxels are not
// all there) if a brick near it holds a negative voxel.  One lane sieves the eight cells of a brick x-row with bit-parallel
// operations on the 2-bit classes of the 10 x 3 x 3 voxels around them: 27 cached 16-bit loads for eight cells.
// Out: surv[slot][row] = one byte, bit i = cell (8 bx + i, y, z) survives; block_bits: the 256-cell blocks with a surviving cell.
__global__ void __launch_bounds__(256) k_mc_sift(McArgs a) {
  const KfVolume& v = a.vol;
  const unsigned lane = threadIdx.x & 63u;
  const int R = v.res, nb = v.nb;
  const unsigned n_d1 = *a.n_d1;
  for (unsigned i = blockIdx.x * 4u + (threadIdx.x >> 6); i < n_d1; i += gridDim.x * 4u) {
    const unsigned slot = a.d1_list[i];
    unsigned mask8 = 0;
    {
      const int bx = (int)(slot % (unsigned)nb), by = (int)((slot / (unsigned)nb) % (unsigned)nb), bz = (int)(slot / ((unsigned)nb * nb)) + v.bz0;
      const int y = by * 8 + (int)(lane & 7u), z = bz * 8 + (int)(lane >> 3);
      const bool rim_row = y < 1 || y > R - 2 || z < 1 || z > R - 2 || !kf_z_stored(v, z - 1) || !kf_z_stored(v, z + 1);
      if (!rim_row) {
        unsigned all_nz = 0x5555u, all_and = 0xFFFFu, all_or = 0u;
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz) {
          const unsigned zz = (unsigned)(z + dz);
#pragma unroll
          for (int dy = -1; dy <= 1; ++dy) {
            const unsigned yy = (unsigned)(y + dy);
            const size_t rowbase = (((size_t)((zz >> 3) - (unsigned)v.bz0) * nb + (yy >> 3)) * nb) * 64u + (((zz & 7u) << 3) | (yy & 7u));
            const unsigned mid = a.codes[rowbase + (size_t)bx * 64u];
            const unsigned left = bx > 0 ? a.codes[rowbase + (size_t)(bx - 1) * 64u] : 0u;
            const unsigned right = bx < nb - 1 ? a.codes[rowbase + (size_t)(bx + 1) * 64u] : 0u;
            const unsigned w = (left >> 14) | (mid << 2) | ((right & 3u) << 18);        // classes of voxels 8 bx - 1 .. 8 bx + 8
            const unsigned nz = (w | (w >> 1)) & 0x55555u;                               // bit 2i: voxel i of the window is observed
            all_nz &= nz & (nz >> 2) & (nz >> 4);                                        // bit 2i: so are the three voxels of cell i
            all_and &= w & (w >> 2) & (w >> 4);                                          // bits 2i, 2i+1: AND / OR of the three classes
            all_or |= w | (w >> 2) | (w >> 4);
          }
        }
        const unsigned eq = ~(all_and ^ all_or);
        const unsigned same = eq & (eq >> 1) & 0x5555u;                                  // all 27 classes equal ...
        const unsigned one_or_two = (all_and ^ (all_and >> 1)) & 0x5555u;                // ... and that class is 1 or 2
        unsigned s = all_nz & ~(same & one_or_two) & 0x5555u;
        s = (s | (s >> 1)) & 0x3333u; s = (s | (s >> 2)) & 0x0F0Fu; s = (s | (s >> 4)) & 0x00FFu;          // even bits -> a byte
        mask8 = s;
        if (bx == 0) mask8 = (mask8 & ~1u) | (neighbourhood_has(v, 0, y, z, KF_FLAG_HASNEG) ? 1u : 0u);
        if (bx == nb - 1) mask8 = (mask8 & ~0x80u) | (neighbourhood_has(v, R - 1, y, z, KF_FLAG_HASNEG) ? 0x80u : 0u);
      } else {
        for (int i = 0; i < 8; ++i) if (neighbourhood_has(v, bx * 8 + i, y, z, KF_FLAG_HASNEG)) mask8 |= 1u << i;
      }
    }
    a.surv[(size_t)slot * 64u + lane] = (unsigned char)mask8;
    // the 256-cell block these eight cells lie in (R is a multiple of 8) -> its bit.  The eight rows of one z share a word (two at
    // 2048^3): the first lane of each run of equal words ORs the run's bits together and issues the one atomic.
    unsigned word = 0xFFFFFFFFu, bit = 0u;
    {
      const int bx = (int)(slot % (unsigned)nb), by = (int)((slot / (unsigned)nb) % (unsigned)nb), bz = (int)(slot / ((unsigned)nb * nb)) + v.bz0;
      const int y = by * 8 + (int)(lane & 7u), z = bz * 8 + (int)(lane >> 3);
      if (z >= a.z0 && z < a.z1) {
        const unsigned blk = (unsigned)((((size_t)(z - a.z0) * R + y) * R + (size_t)bx * 8u) >> 8);
        word = blk >> 5; bit = mask8 ? 1u << (blk & 31u) : 0u;
      }
    }
    const unsigned prev_word = (unsigned)__shfl_up((int)word, 1, 8);
    unsigned bits = bit;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      const unsigned wj = (unsigned)__shfl_down((int)word, j, 8), bj = (unsigned)__shfl_down((int)bit, j, 8);
      if ((lane & 7u) + (unsigned)j < 8u && wj == word) bits |= bj;
    }
    if (bits && ((lane & 7u) == 0u || prev_word != word) && (a.block_bits[word] & bits) != bits) atomicOr(&a.block_bits[word], bits);
  }
}
__device__ __forceinline__ bool mc_survives(const McArgs& a, int x, int y, int z) {
  const size_t slot = kf_brick_slot(a.vol, x >> 3, y >> 3, z >> 3);
  return (a.surv[slot * 64u + (size_t)(((z & 7) << 3) | (y & 7))] >> (x & 7)) & 1u;
}

// ---- which blocks need visiting ------------------------------------------------------------------------------------------------
// A cell can produce triangles only if a voxel within +-2 cells of it is negative, i.e. only if one of the 3x3x3 bricks around its
// own brick carries KF_FLAG_HASNEG.  k_mc_dilate writes that as one bit per stored brick (from the packed has-negative bits, 27
// cached word loads per brick) and lists those bricks; the sieve runs over that list and sets one bit per 256-cell block -- a run
// of cells in (z, y, x) order: one x-row segment at 512^3, several short rows in a small volume -- that holds a surviving cell;
// k_mc_list turns the bits into a list.  The list's order does not matter: a block's triangles land where the prefix sum of the
// block counts says.
__global__ void __launch_bounds__(256) k_mc_dilate(KfVolume v, unsigned* __restrict__ nbr_bits, unsigned n_slots, unsigned* __restrict__ d1_list, unsigned* n_d1, KfCounters* work) {
  const unsigned slot = blockIdx.x * 256u + threadIdx.x;
  bool any = false;
  if (slot < n_slots) {
    const int nb = v.nb;
    const int bx = (int)(slot % (unsigned)nb), by = (int)((slot / (unsigned)nb) % (unsigned)nb), bz = (int)(slot / ((unsigned)nb * nb)) + v.bz0;
    for (int dz = -1; dz <= 1; ++dz) {
      const int z = bz + dz;
      if (z < v.bz0 || z >= v.bz1) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        const int y = by + dy;
        if (y < 0 || y >= nb) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int x = bx + dx;
          if (x < 0 || x >= nb) continue;
          const size_t s2 = kf_brick_slot(v, x, y, z);
          any = any || ((v.negbits[s2 >> 5] >> (s2 & 31u)) & 1u);
        }
      }
    }
  }
  const unsigned long long m = __ballot(any);                     // a wave = 64 consecutive slots = two whole words
  unsigned at = 0;
  if ((threadIdx.x & 63) == 0 && slot < n_slots) {
    nbr_bits[slot >> 5] = (unsigned)m; nbr_bits[(slot >> 5) + 1] = (unsigned)(m >> 32);
    if (m) at = atomicAdd(n_d1, (unsigned)__popcll(m));
    if (work && m) atomicAdd(&work->mc_blocks[(blockIdx.x & 63u) * 16u], (unsigned long long)__popcll(m));       // bricks the extraction reads
  }
  at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
  if (any) d1_list[at + (unsigned)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = slot;
}

// the same dilation 32 bricks at a time, for volumes whose brick rows are whole words of the packed has-negative bits (res a multiple
// of 256): a lane ORs the 3 x 3 neighbouring rows' words, each widened by one bit to either side
__global__ void __launch_bounds__(256) k_mc_dilate_words(KfVolume v, unsigned* __restrict__ nbr_bits, unsigned n_words, unsigned* __restrict__ d1_list, unsigned* n_d1,
                                                         KfCounters* work) {
  const unsigned w = blockIdx.x * 256u + threadIdx.x;
  unsigned bits = 0;
  if (w < n_words) {
    const unsigned nb = (unsigned)v.nb, wpr = nb >> 5, nbz = (unsigned)(v.bz1 - v.bz0);
    const unsigned xw = w % wpr, row = w / wpr;
    const int by = (int)(row % nb), bzl = (int)(row / nb);
    for (int dz = -1; dz <= 1; ++dz) {
      const int zz = bzl + dz;
      if (zz < 0 || zz >= (int)nbz) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = by + dy;
        if (yy < 0 || yy >= (int)nb) continue;
        const size_t base = ((size_t)zz * nb + (size_t)yy) * wpr;
        const unsigned c = v.negbits[base + xw];
        const unsigned l = xw > 0 ? v.negbits[base + xw - 1] : 0u, r = xw + 1 < wpr ? v.negbits[base + xw + 1] : 0u;
        bits |= c | (c << 1) | (c >> 1) | (l >> 31) | (r << 31);
      }
    }
    nbr_bits[w] = bits;
  }
  const unsigned mine = (unsigned)__popc(bits);
  unsigned pre = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(pre, off, 64); if ((threadIdx.x & 63) >= (unsigned)off) pre += t; }
  const unsigned wave_total = (unsigned)__shfl(pre, 63, 64);
  unsigned base = 0;
  if ((threadIdx.x & 63) == 63 && wave_total) {
    base = atomicAdd(n_d1, wave_total);
    if (work) atomicAdd(&work->mc_blocks[(blockIdx.x & 63u) * 16u], (unsigned long long)wave_total);
  }
  base = (unsigned)__shfl(base, 63, 64) + pre - mine;
  for (unsigned m = bits; m; m &= m - 1u) d1_list[base++] = w * 32u + (unsigned)__builtin_ctz(m);
}

__global__ void __launch_bounds__(256) k_mc_list(McArgs a) {
  const unsigned n_words = (a.n_blocks + 31u) / 32u;
  const unsigned w = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
  const unsigned bits = w < n_words ? a.block_bits[w] : 0u;
  const unsigned mine = (unsigned)__popc(bits);
  unsigned pre = mine;                                              // wave prefix of the popcounts, one atomic per wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(pre, off, 64); if ((threadIdx.x & 63) >= (unsigned)off) pre += t; }
  const unsigned wave_total = (unsigned)__shfl(pre, 63, 64);
  unsigned base = 0;
  if ((threadIdx.x & 63) == 63 && wave_total) base = atomicAdd(a.n_list, wave_total);
  base = (unsigned)__shfl(base, 63, 64) + pre - mine;
  for (unsigned m = bits; m; m &= m - 1u) a.list[base++] = w * 32u + (unsigned)__builtin_ctz(m);
}

// cell (x, y, z) of lane `tid` in the 256-cell block `blk`: the block's first cell is decoded once (wave-uniform), the lane's offset
// is added with carries -- three 64-bit divisions per LANE here cost as much as the cell's whole evaluation
__device__ __forceinline__ bool mc_cell_of(const McArgs& a, unsigned blk, unsigned tid, size_t n_cells, int& x, int& y, int& z) {
  const unsigned R = (unsigned)a.vol.res;
  const size_t f = (size_t)blk * 256;
  if (f + tid >= n_cells) return false;
  const unsigned row = (unsigned)(f / R);                           // uniform
  unsigned cx = (unsigned)(f - (size_t)row * R) + tid, cy = row % R, cz = row / R;
  while (cx >= R) { cx -= R; if (++cy == R) { cy = 0; ++cz; } }      // at most 256 / R turns (none when R is a multiple of 256)
  x = (int)cx; y = (int)cy; z = a.z0 + (int)cz;
  return true;
}

// workgroup-wide exclusive prefix of one value per lane (256 lanes); s_wave: 4 words; the trailing barrier frees s_wave again
__device__ __forceinline__ unsigned mc_block_excl_scan(unsigned local, unsigned* s_wave, unsigned& total) {
  unsigned inc = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(inc, off, 64); if ((threadIdx.x & 63) >= (unsigned)off) inc += t; }
  if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = inc;
  __syncthreads();
  unsigned wave_off = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += s_wave[w];
  total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  __syncthreads();
  return wave_off + inc - local;
}

// count pass: persistent workgroups take MC_BATCH listed blocks at a time.  The surviving cells of the whole batch -- a few per
// block where the surface crosses the x-rows, all 256 where it runs along them -- are queued in LDS in (block, x) order, one lane
// per sieve BYTE (eight cells) doing the queueing, and evaluated by dense waves; a prefix over the queue gives every cell with
// triangles its offset inside its block, and the cell is appended -- in no particular order -- to a global record list {block,
// lane, triangles, offset}: the emit pass then needs no workgroup structure at all.  Out: block_counts[blk] = triangles of the block.
#ifndef MC_BATCH
#define MC_BATCH 64
#endif
#ifndef MC_COUNT_ATTR
#define MC_COUNT_ATTR
#endif
__global__ void __launch_bounds__(256) MC_COUNT_ATTR k_mc_count(McArgs a) {
  __shared__ unsigned s_cnt[MC_BATCH], s_qstart[MC_BATCH], s_bstart[MC_BATCH], s_wave[4];
  __shared__ unsigned short s_q[MC_BATCH * 256];
  const int R = a.vol.res;
  const size_t n_cells = (size_t)(a.z1 - a.z0) * R * R;
  const unsigned n_list = *a.n_list;
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned base = blockIdx.x * MC_BATCH; base < n_list; base += gridDim.x * MC_BATCH) {
    const unsigned nbat = n_list - base < MC_BATCH ? n_list - base : MC_BATCH;
    // the batch's sieve bytes: pass p, lane (b_local, k) -> byte k (cells 8k .. 8k+7) of block p * 8 + b_local; all loads first
    unsigned sv[MC_BATCH / 8];
#pragma unroll
    for (unsigned p = 0; p < MC_BATCH / 8; ++p) {
      const unsigned b = p * 8u + (threadIdx.x >> 5), k = threadIdx.x & 31u;
      sv[p] = 0;
      int cx, cy, cz;
      if (b < nbat && mc_cell_of(a, a.list[base + b], 8u * k, n_cells, cx, cy, cz))
        sv[p] = a.surv[(size_t)kf_brick_slot(a.vol, cx >> 3, cy >> 3, cz >> 3) * 64u + (size_t)(((cz & 7) << 3) | (cy & 7))];
    }
    if (threadIdx.x < MC_BATCH) s_cnt[threadIdx.x] = 0;
    unsigned nq = 0;
#pragma unroll
    for (unsigned p = 0; p < MC_BATCH / 8; ++p) {                                // queue positions: blocks in batch order, cells in x order
      unsigned total;
      unsigned at = nq + mc_block_excl_scan((unsigned)__popc(sv[p]), s_wave, total);
      if ((threadIdx.x & 31u) == 0) s_qstart[p * 8u + (threadIdx.x >> 5)] = at;
      const unsigned q0 = ((p * 8u + (threadIdx.x >> 5)) << 8) | (8u * (threadIdx.x & 31u));
      for (unsigned m = sv[p]; m; m &= m - 1u) s_q[at++] = (unsigned short)(q0 + (unsigned)__builtin_ctz(m));
      nq += total;
    }
    __syncthreads();
    unsigned carry = 0;
    for (unsigned c0 = 0; c0 < nq; c0 += 256u) {
      const unsigned i = c0 + threadIdx.x;
      unsigned b = 0, t = 0, blk = 0; int n = 0;
      if (i < nq) {
        const unsigned q = s_q[i];
        b = q >> 8; t = q & 255u; blk = a.list[base + b];
        int cx, cy, cz;
        mc_cell_of(a, blk, t, n_cells, cx, cy, cz);
        CellEval e;
        n = eval_cell(a, cx, cy, cz, e);
      }
      unsigned total;
      const unsigned excl = carry + mc_block_excl_scan((unsigned)n, s_wave, total);
      if (i < nq && i == s_qstart[b]) s_bstart[b] = excl;                        // the prefix at the block's first queued cell
      __syncthreads();
      const unsigned long long m = __ballot(n > 0);
      unsigned at = 0;
      if (lane == 0 && m) at = atomicAdd(a.n_recs, (unsigned)__popcll(m));
      at = (unsigned)__builtin_amdgcn_readfirstlane((int)at) + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
      if (n > 0) {
        atomicAdd(&s_cnt[b], (unsigned)n);
        if (at < a.recs_cap) a.recs[at] = make_uint2(blk, t | ((unsigned)n << 8) | ((excl - s_bstart[b]) << 12));
        else *a.recs_overflow = 1u;
      }
      carry += total;
    }
    __syncthreads();
    if (threadIdx.x < nbat) a.block_counts[a.list[base + threadIdx.x]] = s_cnt[threadIdx.x];
    __syncthreads();
  }
}

// ---- exclusive prefix sum of the block counts, in place, in three parallel steps ----------------------------------------------
// (1) every workgroup sums its chunk of MC_CHUNK counts; (2) one workgroup turns the chunk sums into their exclusive prefix (a
// few thousand values even at 2048^3); (3) every workgroup rescans its chunk on top of its offset.  total -> counts[n].
__global__ void __launch_bounds__(256) k_mc_scan_reduce(const unsigned* __restrict__ counts, unsigned n, unsigned* __restrict__ partials) {
  __shared__ unsigned s_wave[4];
  const unsigned i0 = blockIdx.x * MC_CHUNK + threadIdx.x * 16u;
  unsigned local = 0;
  if (i0 + 16u <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(counts + i0);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint4 q = p[k]; local += q.x + q.y + q.z + q.w; }
  } else for (unsigned k = 0; k < 16u; ++k) if (i0 + k < n) local += counts[i0 + k];
  unsigned total;
  mc_block_excl_scan(local, s_wave, total);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256) k_mc_scan_partials(unsigned* partials, unsigned n_chunks, unsigned* counts, unsigned n, KfCounters* cnt) {
  __shared__ unsigned s_wave[4]; __shared__ unsigned s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (unsigned base = 0; base < n_chunks; base += 1024u) {
    const unsigned i0 = base + threadIdx.x * 4u;
    unsigned v[4]; unsigned local = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (i0 + k < n_chunks) ? partials[i0 + k] : 0u; local += v[k]; }
    unsigned total;
    unsigned excl = s_carry + mc_block_excl_scan(local, s_wave, total);
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (i0 + k < n_chunks) partials[i0 + k] = excl; excl += v[k]; }
    __syncthreads();
    if (threadIdx.x == 0) s_carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) { counts[n] = s_carry; cnt->scan_total = s_carry; }
}
__global__ void __launch_bounds__(256) k_mc_scan_apply(unsigned* __restrict__ counts, unsigned n, const unsigned* __restrict__ partials) {
  __shared__ unsigned s_wave[4];
  const unsigned i0 = blockIdx.x * MC_CHUNK + threadIdx.x * 16u;
  unsigned v[16]; unsigned local = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { v[k] = (i0 + k < n) ? counts[i0 + k] : 0u; local += v[k]; }
  unsigned total;
  unsigned excl = partials[blockIdx.x] + mc_block_excl_scan(local, s_wave, total);
#pragma unroll
  for (int k = 0; k < 16; ++k) { if (i0 + k < n) counts[i0 + k] = excl; excl += v[k]; }
}

// triangles of an evaluated cell -> fixed positions start, start + 1, ... (marchingcube.cu:28-38, :129-136)
__device__ __forceinline__ void mc_write_triangles(const McArgs& a, const CellEval& e, int n, unsigned start) {
  for (int t = 0; t < n; ++t) {
    const unsigned dst = start + (unsigned)t;
    if (dst >= a.max_tris) break;                                                  // marchingcube.cu:29-31
    kf_triangle tri;
    tri.v0 = edge_vertex(a, e, (unsigned)((e.word >> (12 * t)) & 0xF));
    tri.v1 = edge_vertex(a, e, (unsigned)((e.word >> (12 * t + 4)) & 0xF));
    tri.v2 = edge_vertex(a, e, (unsigned)((e.word >> (12 * t + 8)) & 0xF));
    a.tris[dst] = tri;
  }
}
// emit pass: one lane per recorded cell, in whatever order the records were appended: the cell is evaluated again and its
// triangles land at  (what the buffer already held) + (exclusive prefix of the block counts) + (the cell's offset in its block)
// -- the reference never clears its counter (MarchingcubeData.h:56,99), so an extraction appends.
__global__ void __launch_bounds__(256) k_mc_emit_recs(McArgs a) {
  if (*a.recs_overflow) return;
  const unsigned n_recs = *a.n_recs;
  const int R = a.vol.res;
  const size_t n_cells = (size_t)(a.z1 - a.z0) * R * R;
  const unsigned held = a.cnt->n_triangles;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n_recs; i += gridDim.x * 256u) {
    const uint2 r = a.recs[i];
    int cx, cy, cz;
    mc_cell_of(a, r.x, r.y & 255u, n_cells, cx, cy, cz);
    CellEval e;
    const int n = eval_cell(a, cx, cy, cz, e);
    mc_write_triangles(a, e, n, held + a.block_counts[r.x] + (r.y >> 12));
  }
}
// the same from the block counts alone, for an extraction whose cells with triangles outnumber the record list (more cells than
// the triangle buffer holds triangles): the listed blocks that counted triangles, one at a time; ORDER-PRESERVING compaction of the
// surviving cells, evaluation by the first waves, intra-workgroup prefix of the triangle counts.
__global__ void __launch_bounds__(256) k_mc_emit(McArgs a) {
  __shared__ unsigned s_wave[4], s_wc[4];
  __shared__ unsigned char s_q[256];
  if (!*a.recs_overflow) return;
  const int R = a.vol.res;
  const size_t n_cells = (size_t)(a.z1 - a.z0) * R * R;
  const unsigned n_list = *a.n_list;
  for (unsigned li = blockIdx.x; li < n_list; li += gridDim.x) {
    const unsigned blk = a.list[li];
    const unsigned my_base = a.block_counts[blk], my_count = a.block_counts[blk + 1] - my_base;
    if (my_count == 0) continue;                                                  // uniform
    int cx, cy, cz;
    const bool pass = mc_cell_of(a, blk, threadIdx.x, n_cells, cx, cy, cz) && mc_survives(a, cx, cy, cz);
    const unsigned long long m = __ballot(pass);
    if ((threadIdx.x & 63) == 0) s_wc[threadIdx.x >> 6] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned at = (unsigned)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
    for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) at += s_wc[w];
    const unsigned n_pass = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    if (pass) s_q[at] = (unsigned char)threadIdx.x;
    __syncthreads();
    CellEval e; e.ntri = 0;
    int n = 0;
    if (threadIdx.x < n_pass) {
      mc_cell_of(a, blk, s_q[threadIdx.x], n_cells, cx, cy, cz);
      n = eval_cell(a, cx, cy, cz, e);
    }
    unsigned total;
    const unsigned off0 = mc_block_excl_scan((unsigned)n, s_wave, total);
    mc_write_triangles(a, e, n, a.cnt->n_triangles + my_base + off0);
  }
}

__global__ void k_mc_finish(KfCounters* cnt, unsigned max_tris) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned long long t = (unsigned long long)cnt->n_triangles + cnt->scan_total;
    cnt->n_triangles = (unsigned)(t > max_tris ? max_tris : t);
  }
}

extern "C" int kf_marching_cubes(kf_ctx* c, int has_color, float thr) {
  if (!c) return KF_ERR_ARG;
  if (!c->triangles || c->max_triangles == 0) return KF_ERR_STATE;
  if (has_color && !c->vol.color) return KF_ERR_STATE;
  McArgs a;
  a.vol = c->vol; a.z0 = c->vol.own_z0; a.z1 = c->vol.own_z1; a.has_color = has_color; a.thr = thr;
  const size_t n_cells = (size_t)(a.z1 - a.z0) * c->vol.res * c->vol.res;
  a.n_blocks = (unsigned)((n_cells + 255) / 256);
  if (a.n_blocks > c->mc_blocks_cap) return KF_ERR_STATE;
  const unsigned n_chunks = (a.n_blocks + MC_CHUNK - 1) / MC_CHUNK;
  if (!c->mc_list) {                                       // extraction scratch: allocated by the first extraction, not by every context
    KF_CHECK(hipSetDevice(c->cfg.device));                 // (per 4-KiB brick: 128 B of voxel classes + 64 B of sieve bits + 8 B of row bits)
    // all or nothing: the pointers are committed to the context only once every allocation has succeeded (at 2048^3 the scratch is
    // ~3.3 GB next to 68.7 GB of voxels -- a failure must not leave a half-allocated set behind that the next call would trust)
    const size_t sizes[8] = {(c->mc_blocks_cap + 4) * sizeof(unsigned),                                  // list: [0] length, [1] records, [2] overflow, [3] bricks; then the block ids
                             (c->n_stored_bricks / 32 + 4) * sizeof(unsigned),                           // neighbourhood bits
                             ((c->mc_blocks_cap + MC_CHUNK - 1) / MC_CHUNK + 1) * sizeof(unsigned),       // scan partials
                             c->n_stored_bricks * 64 * sizeof(unsigned short),                            // voxel classes
                             c->n_stored_bricks * 64,                                                    // sieve survivors
                             c->n_stored_bricks * sizeof(unsigned),                                      // brick list
                             (c->mc_blocks_cap / 32 + 2) * sizeof(unsigned),                             // block bits
                             (size_t)c->max_triangles * sizeof(uint2)};                                  // records: a recorded cell holds >= 1 triangle
    void* got[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 8; ++i) {
      const hipError_t e = hipMalloc(&got[i], sizes[i]);
      if (e != hipSuccess) { for (int j = 0; j < i; ++j) hipFree(got[j]); (void)hipGetLastError(); return (int)e; }
    }
    c->mc_list = (unsigned*)got[0]; c->mc_nbr_bits = (unsigned*)got[1]; c->mc_partials = (unsigned*)got[2]; c->mc_codes = (unsigned short*)got[3];
    c->mc_surv = (unsigned char*)got[4]; c->mc_d1_list = (unsigned*)got[5]; c->mc_block_bits = (unsigned*)got[6]; c->mc_recs = (uint2*)got[7];
    c->mc_zero_serial = c->vol_flags_serial - 1;
  }
  a.block_counts = c->mc_block_counts; a.tris = c->triangles; a.max_tris = c->max_triangles; a.cnt = c->counters;
  a.count_work = c->count_work;
  a.nbr_bits = c->mc_nbr_bits; a.n_list = c->mc_list; a.n_recs = c->mc_list + 1; a.recs_overflow = c->mc_list + 2; a.list = c->mc_list + 4;
  a.partials = c->mc_partials; a.codes = c->mc_codes; a.surv = c->mc_surv; a.block_bits = c->mc_block_bits;
  a.recs = c->mc_recs; a.recs_cap = c->max_triangles; a.d1_list = c->mc_d1_list; a.n_d1 = c->mc_list + 3;
  kf_evt_begin(c, KF_STAGE_MCUBES);
  if (c->mc_zero_serial != c->vol_flags_serial) {          // bricks outside the has-negative neighbourhood set must read as class 0 / no survivor
    KF_CHECK(hipMemsetAsync(c->mc_codes, 0, c->n_stored_bricks * 64 * sizeof(unsigned short), c->stream));
    KF_CHECK(hipMemsetAsync(c->mc_surv, 0, c->n_stored_bricks * 64, c->stream));
    c->mc_zero_serial = c->vol_flags_serial;
  }
  KF_CHECK(hipMemsetAsync(c->mc_block_counts, 0, ((size_t)a.n_blocks + 1) * sizeof(unsigned), c->stream));
  KF_CHECK(hipMemsetAsync(c->mc_list, 0, 4 * sizeof(unsigned), c->stream));
  KF_CHECK(hipMemsetAsync(c->mc_block_bits, 0, ((size_t)a.n_blocks / 32 + 1) * sizeof(unsigned), c->stream));
  const unsigned n_slots = (unsigned)c->n_stored_bricks;
  if (c->vol.nb % 32 == 0)
    hipLaunchKernelGGL(k_mc_dilate_words, dim3((n_slots / 32 + 255) / 256), dim3(256), 0, c->stream, c->vol, c->mc_nbr_bits, n_slots / 32, a.d1_list, a.n_d1, c->count_work ? c->counters : nullptr);
  else
    hipLaunchKernelGGL(k_mc_dilate, dim3((n_slots + 255) / 256), dim3(256), 0, c->stream, c->vol, c->mc_nbr_bits, n_slots, a.d1_list, a.n_d1, c->count_work ? c->counters : nullptr);
  const unsigned walk = (unsigned)c->num_cus * 8u;         // persistent workgroups walking the brick list / the block list / the records
  hipLaunchKernelGGL(k_mc_codes, dim3(walk), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_mc_sift, dim3(walk), dim3(256), 0, c->stream, a);
  const unsigned list_wgs = ((a.n_blocks + 31u) / 32u + 255u) / 256u;
  const unsigned lgx = list_wgs < 65535u ? list_wgs : 65535u, lgy = (list_wgs + lgx - 1) / lgx;
  hipLaunchKernelGGL(k_mc_list, dim3(lgx, lgy), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_mc_count, dim3(walk), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_mc_scan_reduce, dim3(n_chunks), dim3(256), 0, c->stream, a.block_counts, a.n_blocks, a.partials);
  hipLaunchKernelGGL(k_mc_scan_partials, dim3(1), dim3(256), 0, c->stream, a.partials, n_chunks, a.block_counts, a.n_blocks, c->counters);
  hipLaunchKernelGGL(k_mc_scan_apply, dim3(n_chunks), dim3(256), 0, c->stream, a.block_counts, a.n_blocks, a.partials);
  hipLaunchKernelGGL(k_mc_emit_recs, dim3(walk), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_mc_emit, dim3(walk), dim3(256), 0, c->stream, a);              // returns at once unless the record list overflowed
  hipLaunchKernelGGL(k_mc_finish, dim3(1), dim3(64), 0, c->stream, c->counters, c->max_triangles);
  kf_evt_end(c, KF_STAGE_MCUBES);
  return (int)hipGetLastError();
}

extern "C" int kf_clear_triangles(kf_ctx* c) {
  if (!c) return KF_ERR_ARG;
  KF_CHECK(hipMemsetAsync(&c->counters->n_triangles, 0, sizeof(unsigned), c->stream));
  return 0;
}

extern "C" int kf_triangle_count(kf_ctx* c, uint32_t* count) {
  if (!c || !count) return KF_ERR_ARG;
  KF_CHECK(hipMemcpyAsync(c->host_pinned, &c->counters->n_triangles, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  *count = *(unsigned*)c->host_pinned;
  return 0;
}

extern "C" int kf_read_triangles(kf_ctx* c, kf_triangle* dst, uint32_t first, uint32_t count) {
  if (!c || !dst) return KF_ERR_ARG;
  if ((uint64_t)first + count > c->max_triangles) return KF_ERR_ARG;
  if (count == 0) return 0;
  KF_CHECK(hipMemcpyAsync(dst, c->triangles + first, (size_t)count * sizeof(kf_triangle), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}
