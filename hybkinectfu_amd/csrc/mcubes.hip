// mcubes.hip -- marching-cubes iso-surface extraction (reference: extractIsoSurfaceKernel & helpers,
// src/cuda/marchingcube.cu:5-164; tables src/cuda/marchingcube_table.h; MarchingcubeData src/cuda/MarchingcubeData.h).
//
// The reference appends triangles with one global atomicAdd, so its output ORDER is nondeterministic (and its
// check-then-add can overshoot the buffer, marchingcube.cu:29-32).  Here extraction is deterministic:
//   mark   : which 256-cell blocks (z, y, x order) can hold surface at all -> an unordered list (brick flags only, no voxel read);
//   count  : one lane per cell of the listed blocks -> per-block triangle count;
//   scan   : exclusive prefix over ALL blocks in order (three parallel steps);
//   emit   : the same cell logic again, intra-workgroup prefix, triangle k of cell (x,y,z) lands at a fixed index.
// The canonical order is (z, y, x, k) -- identical for 1 GPU and for concatenated z-slabs.
// Space skipping: a cell can only produce triangles if a corner SDF is negative, which needs a negative voxel in the
// cell's +-2 neighbourhood (trilinear weights are convex), so workgroups and cells whose neighbourhood bricks carry no
// KF_FLAG_HASNEG are dropped before any voxel is read.  The triangle table is packed into 256 x 64-bit words (16
// nibbles per case) instead of the reference's 16 KiB int table; the 12-bit edge mask is derived from it.
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
__device__ __forceinline__ int eval_cell(const McArgs& a, int x, int y, int z, CellEval& e) {
  const KfVolume& v = a.vol;
  e.ntri = 0;
  if (!neighbourhood_has(v, x, y, z, KF_FLAG_HASNEG)) return 0;
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

// ---- which blocks need visiting ------------------------------------------------------------------------------------------------
// A cell can produce triangles only if a voxel within +-2 cells of it is negative, i.e. only if one of the 3x3x3 bricks around its
// own brick carries KF_FLAG_HASNEG.  k_mc_dilate writes that as one bit per stored brick (from the packed has-negative bits, 27
// cached word loads per brick); k_mc_mark tests each 256-cell block -- a run of cells in (z, y, x) order: one x-row segment at
// 512^3, several short rows in a small volume -- against the bits of the bricks it crosses and appends the survivors to a list.
// The list's order does not matter: a block's triangles land where the prefix sum of the block counts says.
__global__ void __launch_bounds__(256) k_mc_dilate(KfVolume v, unsigned* __restrict__ nbr_bits, unsigned n_slots) {
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
  if ((threadIdx.x & 63) == 0 && slot < n_slots) { nbr_bits[slot >> 5] = (unsigned)m; nbr_bits[(slot >> 5) + 1] = (unsigned)(m >> 32); }
}

__global__ void __launch_bounds__(256) k_mc_mark(McArgs a) {
  const KfVolume& v = a.vol;
  const unsigned R = (unsigned)v.res;
  const unsigned blk = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
  bool live = false;
  if (blk < a.n_blocks) {
    const size_t n_cells = (size_t)(a.z1 - a.z0) * R * R;
    size_t c = (size_t)blk * 256, end = c + 256 < n_cells ? c + 256 : n_cells;
    while (c < end && !live) {                                     // one iteration per x-row the block crosses
      const unsigned x = (unsigned)(c % R), y = (unsigned)((c / R) % R), z = (unsigned)a.z0 + (unsigned)(c / ((size_t)R * R));
      const unsigned run = (unsigned)((end - c) < (size_t)(R - x) ? (end - c) : (size_t)(R - x));
      const size_t s0 = kf_brick_slot(v, (int)(x >> 3), (int)(y >> 3), (int)(z >> 3));
      for (unsigned b = 0; b <= ((x + run - 1) >> 3) - (x >> 3); ++b) { const size_t s2 = s0 + b; live = live || ((a.nbr_bits[s2 >> 5] >> (s2 & 31u)) & 1u); }
      c += run;
    }
  }
  const unsigned long long m = __ballot(live);
  unsigned base = 0;
  if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(a.n_list, (unsigned)__popcll(m));
  base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
  if (live) a.list[base + (unsigned)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = blk;
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

// count pass: one workgroup per listed block (grid-stride), one lane per cell
__global__ void __launch_bounds__(256) k_mc_count(McArgs a) {
  __shared__ unsigned s_sum[4];
  const int R = a.vol.res;
  const size_t n_cells = (size_t)(a.z1 - a.z0) * R * R;
  const unsigned n_list = *a.n_list;
  for (unsigned li = blockIdx.x; li < n_list; li += gridDim.x) {
    const unsigned blk = a.list[li];
    const size_t i = (size_t)blk * 256 + threadIdx.x;
    int n = 0;
    int cx, cy, cz;
    if (mc_cell_of(a, blk, threadIdx.x, n_cells, cx, cy, cz)) { CellEval e; n = eval_cell(a, cx, cy, cz, e); }
    const float s = kf_wave_sum((float)n);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = (unsigned)s;
    __syncthreads();
    if (threadIdx.x == 0) a.block_counts[blk] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    __syncthreads();
  }
}

// ---- exclusive prefix sum of the block counts, in place, in three parallel steps ----------------------------------------------
// (1) every workgroup sums its chunk of MC_CHUNK counts; (2) one workgroup turns the chunk sums into their exclusive prefix (a
// few thousand values even at 2048^3); (3) every workgroup rescans its chunk on top of its offset.  total -> counts[n].
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

// emit pass: the same cell logic again for the listed blocks that counted triangles, intra-workgroup prefix, fixed output index
__global__ void __launch_bounds__(256) k_mc_emit(McArgs a) {
  __shared__ unsigned s_wave[4];
  const int R = a.vol.res;
  const size_t n_cells = (size_t)(a.z1 - a.z0) * R * R;
  const unsigned n_list = *a.n_list;
  for (unsigned li = blockIdx.x; li < n_list; li += gridDim.x) {
    const unsigned blk = a.list[li];
    const unsigned my_base = a.block_counts[blk], my_count = a.block_counts[blk + 1] - my_base;
    if (my_count == 0) continue;                                                  // uniform
    const size_t i = (size_t)blk * 256 + threadIdx.x;
    CellEval e; e.ntri = 0;
    int n = 0;
    int cx, cy, cz;
    if (mc_cell_of(a, blk, threadIdx.x, n_cells, cx, cy, cz)) n = eval_cell(a, cx, cy, cz, e);
    unsigned inc = (unsigned)n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { unsigned t = __shfl_up(inc, off, 64); if ((threadIdx.x & 63) >= off) inc += t; }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    unsigned off0 = inc - (unsigned)n;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off0 += s_wave[w];
    __syncthreads();
    // append after what the buffer already holds (the reference never clears its counter: MarchingcubeData.h:56,99)
    const unsigned start = a.cnt->n_triangles + my_base + off0;
    for (int t = 0; t < n; ++t) {
      const unsigned dst = start + (unsigned)t;
      if (dst >= a.max_tris) break;                                                // marchingcube.cu:29-31
      kf_triangle tri;
      tri.v0 = edge_vertex(a, e, (unsigned)((e.word >> (12 * t)) & 0xF));
      tri.v1 = edge_vertex(a, e, (unsigned)((e.word >> (12 * t + 4)) & 0xF));
      tri.v2 = edge_vertex(a, e, (unsigned)((e.word >> (12 * t + 8)) & 0xF));
      a.tris[dst] = tri;
    }
  }
}

__global__ void k_mc_finish(KfCounters* cnt, unsigned max_tris, const unsigned* n_list, int count_work) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned long long t = (unsigned long long)cnt->n_triangles + cnt->scan_total;
    cnt->n_triangles = (unsigned)(t > max_tris ? max_tris : t);
    if (count_work) cnt->mc_blocks[0] += *n_list;
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
    KF_CHECK(hipSetDevice(c->cfg.device));
    KF_CHECK(hipMalloc((void**)&c->mc_list, (c->mc_blocks_cap + 1) * sizeof(unsigned)));                 // [0] = length, then the block ids
    KF_CHECK(hipMalloc((void**)&c->mc_nbr_bits, (c->n_stored_bricks / 32 + 4) * sizeof(unsigned)));
    KF_CHECK(hipMalloc((void**)&c->mc_partials, ((c->mc_blocks_cap + MC_CHUNK - 1) / MC_CHUNK + 1) * sizeof(unsigned)));
  }
  a.block_counts = c->mc_block_counts; a.tris = c->triangles; a.max_tris = c->max_triangles; a.cnt = c->counters;
  a.count_work = c->count_work;
  a.nbr_bits = c->mc_nbr_bits; a.n_list = c->mc_list; a.list = c->mc_list + 1; a.partials = c->mc_partials;
  kf_evt_begin(c, KF_STAGE_MCUBES);
  KF_CHECK(hipMemsetAsync(c->mc_block_counts, 0, ((size_t)a.n_blocks + 1) * sizeof(unsigned), c->stream));
  KF_CHECK(hipMemsetAsync(c->mc_list, 0, sizeof(unsigned), c->stream));
  const unsigned n_slots = (unsigned)c->n_stored_bricks;
  hipLaunchKernelGGL(k_mc_dilate, dim3((n_slots + 255) / 256), dim3(256), 0, c->stream, c->vol, c->mc_nbr_bits, n_slots);
  const unsigned mark_wgs = (a.n_blocks + 255) / 256;
  const unsigned mgx = mark_wgs < 65535u ? mark_wgs : 65535u, mgy = (mark_wgs + mgx - 1) / mgx;
  hipLaunchKernelGGL(k_mc_mark, dim3(mgx, mgy), dim3(256), 0, c->stream, a);
  const unsigned walk = (unsigned)c->num_cus * 8u;         // persistent workgroups walking the list
  hipLaunchKernelGGL(k_mc_count, dim3(walk), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_mc_scan_reduce, dim3(n_chunks), dim3(256), 0, c->stream, a.block_counts, a.n_blocks, a.partials);
  hipLaunchKernelGGL(k_mc_scan_partials, dim3(1), dim3(256), 0, c->stream, a.partials, n_chunks, a.block_counts, a.n_blocks, c->counters);
  hipLaunchKernelGGL(k_mc_scan_apply, dim3(n_chunks), dim3(256), 0, c->stream, a.block_counts, a.n_blocks, a.partials);
  hipLaunchKernelGGL(k_mc_emit, dim3(walk), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_mc_finish, dim3(1), dim3(64), 0, c->stream, c->counters, c->max_triangles, c->mc_list, c->count_work);
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
