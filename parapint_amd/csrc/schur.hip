// Schur update S = -sum_i A_i K_i^-1 A_i^T (mpi_explicit_schur_complement.py:312-333) from the coupling rows of the
// factor panels, inertia counts, status publication, and the dispatch of the factorisation of S.
#include "common.hpp"

namespace {

// counters[0..2] += (pos, neg, zero) over all block pivots (codes of padded instances are 0);
// a code is pos | neg << 4 | zero << 8 in 16 bits, 8 codes per 16-byte load
__global__ __launch_bounds__(256) void k_count_codes(const unsigned short* __restrict__ codes, size_t total8,
                                                     int* counters, int* __restrict__ growth, int* __restrict__ growth_seen,
                                                     int batch) {
  __shared__ int red[3][256];
  int pos = 0, neg = 0, zero = 0;
  if (blockIdx.x == 0) {     // instances whose factor showed element growth beyond 1 / u_rt: counted, kept for
    int gr = 0;              // pp_find_growth and cleared for the next factorisation
    for (int i = threadIdx.x; i < batch; i += 256) {
      const int f = growth[i];
      gr += f != 0;
      growth_seen[i] = f;
      if (f) growth[i] = 0;
    }
    if (gr) atomicAdd(&counters[3], gr);
  }
  const uint4* c4 = reinterpret_cast<const uint4*>(codes);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
    const uint4 v = c4[i];
    const unsigned int wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned int x = wds[q];
      pos += (int)((x & 15u) + ((x >> 16) & 15u));
      neg += (int)(((x >> 4) & 15u) + ((x >> 20) & 15u));
      zero += (int)(((x >> 8) & 15u) + ((x >> 24) & 15u));
    }
  }
  red[0][threadIdx.x] = pos; red[1][threadIdx.x] = neg; red[2][threadIdx.x] = zero;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 3 && red[threadIdx.x][0] != 0) atomicAdd(&counters[threadIdx.x], red[threadIdx.x][0]);
}

__global__ void k_publish_status(const double* __restrict__ tail, const int* __restrict__ bk, long long* out,
                                 long long seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) publish_status(tail, bk, out, seq);
}

// ------------------------------------------------------------------------------------------
// Inertia codes and growth flags of the group, counted by `ncb` 64-thread workgroups (this is workgroup `cb` of them) into
// the slotted counters; runs in front of the Schur tile workgroups of the same launch.
__device__ __forceinline__ void count_codes_block(const GroupDev& g, unsigned cb, unsigned ncb, size_t total8, int* counters, int lane) {
  int cnt[4] = {0, 0, 0, 0};     // pos, neg, zero, growth
  if (cb == 0) {
    int* growth_seen = g.growth + g.bpad;
    for (int i = lane; i < g.batch; i += 64) {
      const int f = g.growth[i];
      cnt[3] += f != 0;
      growth_seen[i] = f;
      if (f) g.growth[i] = 0;
    }
  }
  const uint4* c4 = reinterpret_cast<const uint4*>(g.codes);
  const size_t stride = (size_t)ncb * 64;
  for (size_t i = (size_t)cb * 64 + lane; i < total8; i += 4 * stride) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (i + u * stride < total8) ? c4[i + u * stride] : make_uint4(0, 0, 0, 0);   // in flight together
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned int wds[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned int x = wds[q];
        cnt[0] += (int)((x & 15u) + ((x >> 16) & 15u));
        cnt[1] += (int)(((x >> 4) & 15u) + ((x >> 20) & 15u));
        cnt[2] += (int)(((x >> 8) & 15u) + ((x >> 24) & 15u));
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    for (int off = 32; off > 0; off >>= 1) cnt[q] += __shfl_xor(cnt[q], off);
    if (lane == 0 && cnt[q] != 0) atomicAdd(&counters[4 * (cb % PP_CSLOTS) + q], cnt[q]);   // (slots: same-address atomics serialise, ~12 ns each)
  }
}

// Schur tile: half of an 8x8 tile (8 rows x 4 columns, blockIdx.z selects the column half) in
// registers over all panels holding rows of both tile ranges, then summed over the 64 instances of
// the wave through LDS.  Two waves per tile halve the register footprint (4 waves/SIMD).
// The workgroups in front of the ntile_all * nchunk tile workgroups (z = 0 only) count the inertia codes and collect the
// growth flags (the work of k_count_codes) beside the tiles instead of in a launch of their own in front of them.
__global__ __launch_bounds__(64) void k_schur_tiles(GroupDev g, int ntile_all, size_t total8, int* counters) {
  __shared__ double red[32][65];
  const int lane = threadIdx.x;
  const unsigned ncb = gridDim.x - (unsigned)(ntile_all * g.nchunk);    // counting workgroups come first in the grid
  if (blockIdx.x < ncb) {
    if (blockIdx.z != 0 || blockIdx.y != 0) return;
    count_codes_block(g, blockIdx.x, ncb, total8, counters, lane);
    return;
  }
  const unsigned wg = blockIdx.x - ncb;
  const int chunk = pp_chunk64_perm(wg % (unsigned)g.nchunk, (unsigned)g.nchunk);
  const int b = chunk * 64 + lane;
  const size_t bpad = (size_t)g.bpad;
  const int tile = (int)(wg / (unsigned)g.nchunk), half = blockIdx.z;
  double acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  // column-step records {w, position of (row 0, column t) of the panel, -, -, slotA[8], slotB[8]}: one per panel
  // column holding rows of both tile ranges; four steps (48 loads) are in flight together
  // (mapped groups: gridDim.y workgroups share a tile's records -- every instance keeps a clique of its own, so a tile is
  // 64 lanes x one wave whatever the batch is, and the 512 time blocks of C4 gave 1456 waves of 280 us; the slices leave
  // partial cliques in slots of their own, k_scatter_schur adds them)
  const int rs0 = g.stile_ptr[tile], rs1 = g.stile_ptr[tile + 1];
  const int r0 = rs0 + (int)((long long)(rs1 - rs0) * blockIdx.y / gridDim.y);
  const int r1 = rs0 + (int)((long long)(rs1 - rs0) * (blockIdx.y + 1) / gridDim.y);
  constexpr int SG = 4;
  for (int r = r0; r < r1; r += SG) {
    double la[SG][8], ub[SG][4];
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      const bool live = r + s < r1;
      const int* rec = g.stile_rec + 20 * (size_t)max(min(r + s, r1 - 1), rs0);
      const int w = rec[0];
      const double* Up = g.U + (size_t)rec[1] * bpad + b;
      const double* Lp = g.L + (size_t)rec[1] * bpad + b;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int sa = rec[4 + i];
        const double v = Lp[(size_t)(max(sa, 0) * w) * bpad];
        la[s][i] = (sa >= 0 && live) ? v : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sb = rec[12 + 4 * half + j];
        const double v = Up[(size_t)(max(sb, 0) * w) * bpad];
        ub[s][j] = (sb >= 0) ? v : 0.0;
      }
    }
#pragma unroll
    for (int s = 0; s < SG; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] -= la[s][i] * ub[s][j];
  }
  if (g.cmapT) {
    // mapped group (time blocks: every instance has coupling rows of its own): no sum over the lanes, the clique of
    // every instance is kept and scattered by k_scatter_schur
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        g.Sloc[(((size_t)blockIdx.y * ntile_all + tile) * 64 + i * 8 + 4 * half + j) * bpad + b] = acc[i][j];
    return;
  }
  const double mask = (b < g.batch) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[i * 4 + j][lane] = acc[i][j] * mask;
  __syncthreads();
  if (lane < 32) {
    double s = 0.0;
    for (int l = 0; l < 64; ++l) s += red[lane][l];
    // entry (i, 4*half + j) of the tile -> slot i*8 + 4*half + j of the 64-entry tile record
    const int i = lane >> 2, j = lane & 3;
    g.Spart[((size_t)chunk * ntile_all + tile) * 64 + i * 8 + 4 * half + j] = s;
  }
}

// S[ci][cj] += sum over chunks of the tile partials (both triangles of the dense S)
// overwrite: the tiles of this group cover all of S and it is the first group: S = instead of S += (no memset in front)
__global__ __launch_bounds__(64) void k_schur_reduce(GroupDev g, int ntiles, double* __restrict__ S,
                                                     int* __restrict__ counters, int overwrite) {
  const int lane = threadIdx.x, tile = blockIdx.x;
  // last group of the handle: the inertia counters (complete: the counting workgroups ran in the launch before this
  // one) go to the tail of the S buffer, so that they travel with the all-reduce, and are cleared for the next
  // factorisation (saves the one-thread k_write_tail launch and a memset)
  if (counters && tile == 0) {
    int4 c = reinterpret_cast<int4*>(counters)[lane];       // PP_CSLOTS == 64: one slot per lane
    reinterpret_cast<int4*>(counters)[lane] = make_int4(0, 0, 0, 0);
    for (int off = 32; off > 0; off >>= 1) {
      c.x += __shfl_xor(c.x, off); c.y += __shfl_xor(c.y, off); c.z += __shfl_xor(c.z, off); c.w += __shfl_xor(c.w, off);
    }
    if (lane == 0) {
      double* tail = S + (size_t)g.nc * g.nc;
      tail[0] = (double)c.z;
      tail[1] = (double)c.x;
      tail[2] = (double)c.y;
      tail[3] = 0.0;
      tail[4] = (double)c.w;
      tail[5] = tail[6] = tail[7] = 0.0;
    }
  }
  double s = 0.0;
  for (int c = 0; c < g.nchunk; ++c) s += g.Spart[((size_t)c * ntiles + tile) * 64 + lane];
  const int ci = g.stile_a[tile] * 8 + (lane >> 3), cj = g.stile_b[tile] * 8 + (lane & 7);
  if (ci < g.nc && cj < g.nc && ci >= cj) {
    const size_t lo = (size_t)ci + (size_t)cj * g.nc, up = (size_t)cj + (size_t)ci * g.nc;
    if (overwrite) { S[lo] = s; if (ci != cj) S[up] = s; }
    else { S[lo] += s; if (ci != cj) S[up] += s; }
  }
}

// MFMA form of the Schur update for unmapped groups (round 2).  S = - sum over panel columns and INSTANCES of
// (coupling part of the L column)(coupling part of the U column)^T: the instance index is a genuine GEMM K dimension
// -- every operand row is a contiguous [instance] vector -- so a 16 x 16 tile of S over one panel column and the 64
// instances of a chunk is 16 v_mfma_f64_16x16x4 with K = 4 instances each.  Lane (li, lk) feeds row li of the tile's row
// (A, from L) and column (B, from U) ranges with the instances 16 lk .. 16 lk + 15 of the chunk (any assignment of
// instances to K slots is fine as long as A and B agree): 128 contiguous bytes per lane and operand, eight 16-byte
// loads, and a wave reads each 512-byte row of the chunk exactly once.  Against the register-tile kernel (k_schur_tiles:
// 12 loads of 8 bytes per lane for 32 multiply-adds) that is a third of the operand traffic per multiply-add and a
// tenth of the instructions.  Records: per (tile, panel column) the 16 + 16 row positions, -1 where the panel has no row.
// nsplit waves share the records of a tile; partial tiles go to Spart [chunk][tile][split][256] and are added in fixed order
// by k_schur_reduce_mfma (deterministic, no atomics).
__global__ __launch_bounds__(64) void k_schur_mfma(GroupDev g, int nwork_items, size_t total8, int* counters) {
  const int lane = threadIdx.x;
  const unsigned nwork = (unsigned)(nwork_items * g.nchunk);
  const unsigned ncb = gridDim.x - nwork;                                // counting workgroups come first in the grid
  if (blockIdx.x < ncb) { count_codes_block(g, blockIdx.x, ncb, total8, counters, lane); return; }
  const unsigned wg = blockIdx.x - ncb;
  const int chunk = pp_chunk64_perm(wg % (unsigned)g.nchunk, (unsigned)g.nchunk);
  const int item = (int)(wg / (unsigned)g.nchunk);       // a slice of at most PP_MT_SLICE records of one tile
  const int li = lane & 15, lk = lane >> 4;
  const size_t bpad = (size_t)g.bpad;
  const int ra = g.mt_item[2 * item], rb = g.mt_item[2 * item + 1];
  const size_t lane_off = (size_t)chunk * 64 + (size_t)lk * 16;      // first of this lane's 16 instances
  // instances beyond the batch (ragged last chunk) hold undefined factor values: their K slots are zeroed
  const int nvalid = min(16, max(0, g.batch - (int)lane_off));
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int r = ra; r < rb; r += 2) {
    // two records at a time: all their operands are requested before the first multiply
    int oa[2], ob[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bool live = r + s < rb;
      oa[s] = live ? g.mt_rec[(size_t)(r + s) * 32 + li] : -1;
      ob[s] = live ? g.mt_rec[(size_t)(r + s) * 32 + 16 + li] : -1;
    }
    double2 a[2][8], b[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[s][q] = make_double2(0.0, 0.0); b[s][q] = make_double2(0.0, 0.0); }
      if (oa[s] >= 0) {         // (rows the panel does not have are not requested)
        const double2* pa = reinterpret_cast<const double2*>(g.L + (size_t)oa[s] * bpad + lane_off);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[s][q] = pa[q];
      }
      if (ob[s] >= 0) {
        const double2* pb = reinterpret_cast<const double2*>(g.U + (size_t)ob[s] * bpad + lane_off);
#pragma unroll
        for (int q = 0; q < 8; ++q) b[s][q] = pb[q];
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double a0 = (2 * q < nvalid) ? a[s][q].x : 0.0, a1 = (2 * q + 1 < nvalid) ? a[s][q].y : 0.0;
        const double b0 = (2 * q < nvalid) ? b[s][q].x : 0.0, b1 = (2 * q + 1 < nvalid) ? b[s][q].y : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
      }
    }
  }
  // D[row = lk + 4 r][col = li] (see k_ldl_regs): slot (row * 16 + col) of the 256-entry partial tile
  double* out = g.Spart + ((size_t)chunk * nwork_items + item) * 256;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(lk + 4 * r) * 16 + li] = -acc[r];
}

// The same with 32 x 32 super-tiles (2 x 2 tiles per wave) for large coupling dimensions (n_c >= PP_MT_WIDE_NC: the
// 1000 x 1000 S of C5 has 2016 tiles).  A 16 x 16 tile reads 16 + 16 operand rows for 16 matrix instructions per panel
// column and chunk -- 2 flop per byte, HBM/L2-bound at ~10 TFLOP/s (C5: 23 % MFMA-busy); a super-tile reads 32 + 32 rows
// for 64.  Records: 64 row positions (A rows of the two row tiles, B rows of the two column tiles); work items
// {first record, end, super-tile, -}; partial tiles [chunk][item][2 a + b][256].  With the 91 tiles of n_c = 200 the
// super-tiles were slower (too few waves), hence the threshold.
__global__ __launch_bounds__(64) void k_schur_mfma_wide(GroupDev g, int nwork_items, size_t total8, int* counters) {
  const int lane = threadIdx.x;
  const unsigned nwork = (unsigned)(nwork_items * g.nchunk);
  const unsigned ncb = gridDim.x - nwork;                                // counting workgroups come first in the grid
  if (blockIdx.x < ncb) { count_codes_block(g, blockIdx.x, ncb, total8, counters, lane); return; }
  const unsigned wg = blockIdx.x - ncb;
  const int chunk = pp_chunk64_perm(wg % (unsigned)g.nchunk, (unsigned)g.nchunk);
  const int item = (int)(wg / (unsigned)g.nchunk);
  const int li = lane & 15, lk = lane >> 4;
  const size_t bpad = (size_t)g.bpad;
  const int ra = g.mt_item[4 * item], rb = g.mt_item[4 * item + 1], super = g.mt_item[4 * item + 2];
  const bool upper_used = g.mt_a[4 * super + 1] >= 0;                 // (false on the diagonal: tile (2 s, 2 s + 1) lies above it)
  const size_t lane_off = (size_t)chunk * 64 + (size_t)lk * 16;
  const int nvalid = min(16, max(0, g.batch - (int)lane_off));
  double4_t acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
  for (int r = ra; r < rb; ++r) {
    int oa[2], ob[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      oa[x] = g.mt_rec[(size_t)r * 64 + 16 * x + li];
      ob[x] = g.mt_rec[(size_t)r * 64 + 32 + 16 * x + li];
    }
    double2 a[2][8], b[2][8];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { a[x][q] = make_double2(0.0, 0.0); b[x][q] = make_double2(0.0, 0.0); }
      if (oa[x] >= 0) {
        const double2* pa = reinterpret_cast<const double2*>(g.L + (size_t)oa[x] * bpad + lane_off);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[x][q] = pa[q];
      }
      if (ob[x] >= 0) {
        const double2* pb = reinterpret_cast<const double2*>(g.U + (size_t)ob[x] * bpad + lane_off);
#pragma unroll
        for (int q = 0; q < 8; ++q) b[x][q] = pb[q];
      }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (!(2 * q < nvalid)) { a[x][q].x = 0.0; b[x][q].x = 0.0; }
        if (!(2 * q + 1 < nvalid)) { a[x][q].y = 0.0; b[x][q].y = 0.0; }
      }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        if (x == 0 && y == 1 && !upper_used) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x][q].x, b[y][q].x, acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x][q].y, b[y][q].y, acc[x][y], 0, 0, 0);
        }
      }
  }
  double* out = g.Spart + ((size_t)chunk * nwork_items + item) * 1024;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(2 * x + y) * 256 + (lk + 4 * r) * 16 + li] = -acc[x][y][r];
}

// S[ci][cj] (+)= sum over the work items of the tile and the chunks of the partial 16 x 16 tiles (both triangles of the dense
// S).  Sixteen partial sums per entry (one per residue of the chunk index; fixed order inside: deterministic) meet in LDS
// and are added as a fixed tree.  Tail as in k_schur_reduce.
__global__ __launch_bounds__(1024) void k_schur_reduce_mfma(GroupDev g, int nwork_items, double* __restrict__ S,
                                                            int* __restrict__ counters, int overwrite, int wide) {
  __shared__ double part[4][256];
  const int tid = threadIdx.x, tile = blockIdx.x, e = tid & 255, sub = tid >> 8;
  if (counters && tile == 0 && tid < 64) {
    const int lane = tid;
    int4 c = reinterpret_cast<int4*>(counters)[lane];       // PP_CSLOTS == 64: one slot per lane
    reinterpret_cast<int4*>(counters)[lane] = make_int4(0, 0, 0, 0);
    for (int off = 32; off > 0; off >>= 1) {
      c.x += __shfl_xor(c.x, off); c.y += __shfl_xor(c.y, off); c.z += __shfl_xor(c.z, off); c.w += __shfl_xor(c.w, off);
    }
    if (lane == 0) {
      double* tail = S + (size_t)g.nc * g.nc;
      tail[0] = (double)c.z;
      tail[1] = (double)c.x;
      tail[2] = (double)c.y;
      tail[3] = 0.0;
      tail[4] = (double)c.w;
      tail[5] = tail[6] = tail[7] = 0.0;
    }
  }
  if (g.mt_a[tile] < 0) return;          // (wide form: the unused quarter of a diagonal super-tile; uniform per workgroup, before any barrier)
  // wide form: tile = 4 * super-tile + quarter, the items belong to the super-tile, partial tiles [chunk][item][quarter][256]
  const int i0 = g.mt_wptr[wide ? tile >> 2 : tile], i1 = g.mt_wptr[(wide ? tile >> 2 : tile) + 1];
  const size_t istride = wide ? 1024 : 256, qoff = wide ? (size_t)(tile & 3) * 256 : 0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int c = sub; c < g.nchunk; c += 4) {                  // (the four quarter-sums: chunks 0,4,8,.. / 1,5,9,.. / ...)
    const double* base = g.Spart + ((size_t)c * nwork_items) * istride + qoff + e;
    int it = i0;
    for (; it + 3 < i1; it += 4) {
      s0 += base[(size_t)it * istride]; s1 += base[(size_t)(it + 1) * istride]; s2 += base[(size_t)(it + 2) * istride]; s3 += base[(size_t)(it + 3) * istride];
    }
    for (; it < i1; ++it) s0 += base[(size_t)it * istride];
  }
  part[sub][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub != 0) return;
  const double s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
  const int ci = g.mt_a[tile] * 16 + (e >> 4), cj = g.mt_b[tile] * 16 + (e & 15);
  if (ci < g.nc && cj < g.nc && ci >= cj) {
    const size_t lo = (size_t)ci + (size_t)cj * g.nc, up = (size_t)cj + (size_t)ci * g.nc;
    if (overwrite) { S[lo] = s; if (ci != cj) S[up] = s; }
    else { S[lo] += s; if (ci != cj) S[up] += s; }
  }
}

// Mapped groups: S[gi][gj] += clique entry (ci, cj) of instance b, gi = cmap[b][ci].  Target: the dense n_c x n_c S
// (both triangles) or the block-tridiagonal storage D[G][gs][gs] | E[G-1][gs][gs] (E_t = rows of block t+1 x columns of
// block t, column-major inside a block).  Entries of different instances may coincide in general: atomic adds.
struct SchurTarget { double* S; int nc, btd, gs, G; int* err; };

__device__ __forceinline__ void schur_add(const SchurTarget& T, int gi, int gj, double v) {
  if (!T.btd) {
    atomicAdd(&T.S[(size_t)gi + (size_t)gj * T.nc], v);
    if (gi != gj) atomicAdd(&T.S[(size_t)gj + (size_t)gi * T.nc], v);
    return;
  }
  const int bi = gi / T.gs, bj = gj / T.gs, ri = gi % T.gs, rj = gj % T.gs;
  const size_t g2 = (size_t)T.gs * T.gs;
  if (bi == bj) {
    atomicAdd(&T.S[(size_t)bi * g2 + ri + (size_t)rj * T.gs], v);
    if (gi != gj) atomicAdd(&T.S[(size_t)bi * g2 + rj + (size_t)ri * T.gs], v);
  } else if (bi == bj + 1) {
    atomicAdd(&T.S[(size_t)T.G * g2 + (size_t)bj * g2 + ri + (size_t)rj * T.gs], v);
  } else if (bj == bi + 1) {
    atomicAdd(&T.S[(size_t)T.G * g2 + (size_t)bi * g2 + rj + (size_t)ri * T.gs], v);
  } else {
    T.err[0] = 1;       // a clique that spans non-adjacent blocks: the structure given to pp_set_coupling_structure is wrong
  }
}

// (blockIdx.y: one of eight slices of the tile's 64 entries -- a lane that walks all 64 alone is a chain of 64 load ->
// atomic round trips: 180 us for the 512 time blocks of C4, MEASURED)
__global__ __launch_bounds__(64) void k_scatter_schur(GroupDev g, int ntiles, SchurTarget T) {
  const int lane = threadIdx.x;
  const int tile = PP_TASK_OF_WG(g.nchunk), b = PP_CHUNK_OF_WG(g.nchunk) * 64 + lane;
  if (b >= g.batch) return;
  const size_t bpad = (size_t)g.bpad;
  const int ta = g.stile_a[tile], tb = g.stile_b[tile];
  for (int e = 8 * (int)blockIdx.y; e < 8 * (int)blockIdx.y + 8; ++e) {
    const int ci = ta * 8 + (e >> 3), cj = tb * 8 + (e & 7);
    if (ci >= g.nc || cj >= g.nc || ci < cj) continue;
    double v = 0.0;
#pragma unroll
    for (int sl = 0; sl < PP_SCHUR_SLICES; ++sl) v += g.Sloc[(((size_t)sl * ntiles + tile) * 64 + e) * bpad + b];
    if (v == 0.0) continue;
    schur_add(T, g.cmapT[(size_t)ci * bpad + b], g.cmapT[(size_t)cj * bpad + b], v);
  }
}

// The same for a group of a few instances (the single first / last time block of a dynamic problem): lane = entry of the
// tile, loop over the instances -- with lane = instance one lane walks the 64 entries of a tile alone (41-90 us for ONE block,
// twice per step on the handle's stream between the factorisation and the reduction of S).
__global__ __launch_bounds__(64) void k_scatter_schur_few(GroupDev g, int ntiles, SchurTarget T) {
  const int e = threadIdx.x, tile = blockIdx.x;
  const size_t bpad = (size_t)g.bpad;
  const int ta = g.stile_a[tile], tb = g.stile_b[tile];
  const int ci = ta * 8 + (e >> 3), cj = tb * 8 + (e & 7);
  if (ci >= g.nc || cj >= g.nc || ci < cj) return;
  for (int b = 0; b < g.batch; ++b) {
    double v = 0.0;
#pragma unroll
    for (int sl = 0; sl < PP_SCHUR_SLICES; ++sl) v += g.Sloc[(((size_t)sl * ntiles + tile) * 64 + e) * bpad + b];
    if (v == 0.0) continue;
    schur_add(T, g.cmapT[(size_t)ci * bpad + b], g.cmapT[(size_t)cj * bpad + b], v);
  }
}

__global__ void k_write_tail(int* counters, double* tail) {
  if (threadIdx.x == 0) {
    int c[4] = {0, 0, 0, 0};
    for (int sl = 0; sl < PP_CSLOTS; ++sl)
      for (int q = 0; q < 4; ++q) { c[q] += counters[4 * sl + q]; counters[4 * sl + q] = 0; }   // (ready for the next factorisation)
    tail[0] = (double)c[2];  // numerically zero pivots
    tail[1] = (double)c[0];
    tail[2] = (double)c[1];
    tail[3] = 0.0;
    tail[4] = (double)c[3];
    tail[5] = tail[6] = tail[7] = 0.0;
  }
}


}  // namespace

extern "C" {

int pp_numeric_schur(pp_handle h) { return pp_numeric_schur_ex(h, 0); }

int pp_numeric_schur_ex(pp_handle h, int side_stream) {
  if (!h || !h->symbolic_done || !h->blocks_factored) return fail(h, 3, "pp_numeric_schur before pp_numeric_factor_blocks");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = join_dense(h)) return rc;          // (S is written below: a dense phase still reading it must be over)
  hipStream_t st = h->stream;
  // side_stream: the Schur update (and the dense phase behind it) on the dense stream, forked behind the factor levels, so
  // that a forward sweep enqueued on the handle's stream right after this call runs beside both
  // MEASURED AND NOT ADOPTED (round 4, C3): 0.825-0.845 ms per step against 0.798 ms with only the dense phase beside the
  // forward sweep -- the Schur update (350 MB of coupling rows) and the sweep compete for the same HBM bandwidth.  Opt-in:
  static const bool want_side = pp::env_switch("PP_SCHUR_SIDE") != nullptr;
  h->schur_on_side = side_stream && want_side && h->dense_overlap && h->dense_stream && !h->profile && !h->btd && h->groups.size() <= 2;
  if (h->schur_on_side) {
    PP_HIP(hipEventRecord(h->ev_dense_fork, h->stream));
    PP_HIP(hipStreamWaitEvent(h->dense_stream, h->ev_dense_fork, 0));
    st = h->dense_stream;
  }
  const int nc = h->nc;
  // S starts from zero -- unless the first group is a plain (unmapped) one whose tiles cover all of S: its reduction
  // then stores instead of adding.  The counters are cleared by whoever writes the tail (zero at allocation).
  bool first_covers = false;
  if (!h->groups.empty() && !h->btd) {
    const Group* g0 = h->groups.front();
    const int nt8 = (nc + 7) / 8;
    first_covers = g0->ntiles > 0 && !g0->dev.cmapT && g0->ntiles == nt8 * (nt8 + 1) / 2;
  }
  if (!first_covers) PP_HIP(hipMemsetAsync(h->S, 0, (schur_doubles(h) + PP_TAIL) * sizeof(double), st));
  bool tail_written = false;
  for (Group* g : h->groups) {
    const pp::Plan& P = g->plan;
    GroupDev& d = g->dev;
    {
      PhaseScope ps(h, 2, g->ntiles > 0 ? 2 : 1);
      const size_t total8 = (size_t)P.npiv * d.bpad / 8;   // bpad is a multiple of 64
      const unsigned ncb = (unsigned)std::min<size_t>(2048, (total8 + 255) / 256);   // counting workgroups in front of the tiles
      if (g->ntiles > 0 && d.cmapT) {
        // mapped group: per-instance cliques, scattered into the dense or the block-tridiagonal S
        hipLaunchKernelGGL(k_schur_tiles, dim3((unsigned)g->ntiles * d.nchunk + ncb, PP_SCHUR_SLICES, 2), dim3(64), 0, st, d, g->ntiles,
                           total8, h->counters);
        const SchurTarget T{h->S, nc, h->btd, h->gs, h->G, h->scatter_err};
        if (d.batch <= 8) hipLaunchKernelGGL(k_scatter_schur_few, dim3((unsigned)g->ntiles), dim3(64), 0, st, d, g->ntiles, T);
        else hipLaunchKernelGGL(k_scatter_schur, dim3((unsigned)g->ntiles * d.nchunk, 8), dim3(64), 0, st, d, g->ntiles, T);
      } else if (g->ntiles > 0 && h->schur_mfma && g->nmt > 0) {
        if (g->mt_wide)
          hipLaunchKernelGGL(k_schur_mfma_wide, dim3((unsigned)g->nmt_items * d.nchunk + ncb), dim3(64), 0, st, d, g->nmt_items, total8,
                             h->counters);
        else
          hipLaunchKernelGGL(k_schur_mfma, dim3((unsigned)g->nmt_items * d.nchunk + ncb), dim3(64), 0, st, d, g->nmt_items, total8,
                             h->counters);
        const bool last = (g == h->groups.back());
        hipLaunchKernelGGL(k_schur_reduce_mfma, dim3(g->nmt), dim3(1024), 0, st, d, g->nmt_items, h->S,
                           last ? h->counters : (int*)nullptr, (first_covers && g == h->groups.front()) ? 1 : 0, g->mt_wide ? 1 : 0);
        tail_written = last;
      } else if (g->ntiles > 0) {
        hipLaunchKernelGGL(k_schur_tiles, dim3((unsigned)g->ntiles * d.nchunk + ncb, 1, 2), dim3(64), 0, st, d, g->ntiles, total8,
                           h->counters);
        const bool last = (g == h->groups.back());
        hipLaunchKernelGGL(k_schur_reduce, dim3(g->ntiles), dim3(64), 0, st, d, g->ntiles, h->S,
                           last ? h->counters : (int*)nullptr, (first_covers && g == h->groups.front()) ? 1 : 0);
        tail_written = last;
      } else {
        hipLaunchKernelGGL(k_count_codes, dim3((unsigned)std::min<size_t>(512, (total8 + 255) / 256)), dim3(256), 0, st,
                           d.codes, total8, h->counters, d.growth, d.growth + d.bpad, d.batch);
      }
    }
  }
  if (!tail_written) hipLaunchKernelGGL(k_write_tail, dim3(1), dim3(64), 0, st, h->counters, h->S + schur_doubles(h));
  PP_HIP(hipGetLastError());
  if (h->schur_on_side) {
    PP_HIP(hipEventRecord(h->ev_dense_done, st));
    h->dense_pending = true;              // (whoever touches S on the handle's stream joins)
  }
  h->numeric_done = true;
  h->schur_done = false;
  return 0;
}

int pp_numeric_local(pp_handle h) {
  if (int rc = pp_numeric_factor_blocks(h)) return rc;
  return pp_numeric_schur(h);
}

int pp_fail_local(pp_handle h, int status) {
  if (!h || !h->symbolic_done) return fail(h, 3, "pp_fail_local before symbolic factorization");
  if (status < 1 || status > 3) return fail(h, 3, "pp_fail_local: status must be 1 (not_enough_memory), 2 (singular) or 3 (error)");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = join_dense(h)) return rc;
  const size_t nn = schur_doubles(h);
  PP_HIP(hipMemsetAsync(h->S, 0, (nn + PP_TAIL) * sizeof(double), h->stream));
  h->fail_code = status == 1 ? 1.0 : status == 2 ? 1e3 : 1e6;
  PP_HIP(hipMemcpyAsync(h->S + nn + 3, &h->fail_code, sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->numeric_done = true;       // the Schur buffer is defined (zero contribution): the collective and the dense phase may run
  h->schur_done = false;
  return 0;
}

double* pp_schur_buffer(pp_handle h) { return (h && h->symbolic_done) ? h->S : nullptr; }

int pp_bind_schur_buffer(pp_handle h, double* dev_ptr) {
  if (!h || !h->symbolic_done) return fail(h, 3, "pp_bind_schur_buffer before symbolic factorization");
  h->S = dev_ptr ? dev_ptr : h->S_own;
  return 0;
}

static int factor_schur_impl(pp_handle h, const double* Q_host, long long corner_nnz);

int pp_factor_schur(pp_handle h, const double* Q_host) {
  if (h) h->corner_nnz = (h->btd && Q_host) ? -1 : 0;        // (a flat Q of a block-tridiagonal S is not kept in pair form)
  return factor_schur_impl(h, Q_host, 0);
}

int pp_factor_schur_corner(pp_handle h, int64_t nnz, const int64_t* pos, const double* val) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_factor_schur_corner before pp_numeric_local");
  if (!h->btd) return fail(h, 3, "pp_factor_schur_corner: S is dense (use pp_factor_schur)");
  if (nnz < 0 || (nnz > 0 && (!pos || !val))) return fail(h, 3, "pp_factor_schur_corner: bad arguments");
  PP_HIP(hipSetDevice(h->device));
  const long long nn = (long long)schur_doubles(h);
  for (int64_t k = 0; k < nnz; ++k)
    if (pos[k] < 0 || pos[k] >= nn) return fail(h, 3, "pp_factor_schur_corner: position outside the Schur buffer");
  if (nnz > 0) {
    // the pairs travel on an upload stream of their own: a copy on the handle's stream would wait behind the block
    // factorisation that is still running there, and the host with it
    if (!h->ev_corner_up) {
      PP_HIP(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
      PP_HIP(hipEventCreateWithFlags(&h->ev_corner_up, hipEventDisableTiming));
      PP_HIP(hipEventCreateWithFlags(&h->ev_corner_done, hipEventDisableTiming));
    }
    hipStream_t up = h->up_stream;
    if ((size_t)nnz > h->corner_cap) {
      PP_HIP(hipStreamSynchronize(h->stream));
      if (h->corner_pos) (void)hipFree(h->corner_pos);
      if (h->corner_val) (void)hipFree(h->corner_val);
      h->corner_pos = nullptr; h->corner_val = nullptr; h->corner_cap = 0;
      int rc = 0;
      if ((rc = dev_alloc<long long>(h, nullptr, &h->corner_pos, (size_t)nnz))) return rc;
      if ((rc = dev_alloc<double>(h, nullptr, &h->corner_val, (size_t)nnz))) return rc;
      h->corner_cap = (size_t)nnz;
      h->corner_used = false;
    }
    if (h->corner_used) PP_HIP(hipStreamWaitEvent(up, h->ev_corner_done, 0));    // the previous scatter has read them
    PP_HIP(hipMemcpyAsync(h->corner_pos, pos, (size_t)nnz * sizeof(long long), hipMemcpyHostToDevice, up));
    PP_HIP(hipMemcpyAsync(h->corner_val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, up));
    PP_HIP(hipEventRecord(h->ev_corner_up, up));
    PP_HIP(hipStreamSynchronize(up));            // the caller's arrays are free again when this returns
    PP_HIP(hipStreamWaitEvent(h->stream, h->ev_corner_up, 0));
  }
  h->corner_nnz = (long long)nnz;
  return factor_schur_impl(h, nullptr, (long long)nnz);
}

static int factor_schur_impl(pp_handle h, const double* Q_host, long long corner_nnz) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_factor_schur before pp_numeric_local");
  PP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const int nc = h->nc;
  const size_t nn = schur_doubles(h);
  if (nc > 0 && h->btd) return ppi_btd_factor_schur(h, Q_host, corner_nnz);      // bcr.hip
  if (nc > 0) {
    if (int rc = ppi_dense_factor_schur(h, Q_host)) return rc;                     // dense.hip
  } else {
    PP_HIP(hipMemsetAsync(h->bkinfo, 0, 4 * sizeof(int), st));
    hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(64), 0, st, h->S + nn, h->bkinfo, h->status_dev, ++h->status_seq);
  }
  PP_HIP(hipGetLastError());
  h->schur_done = true;
  return 0;
}

int pp_get_status(pp_handle h, int64_t out[4]) {
  if (!h || !h->schur_done) return fail(h, 3, "pp_get_status before pp_factor_schur");
  PP_HIP(hipSetDevice(h->device));
  // poll the mailbox; after a bounded spin fall back to a stream synchronisation (which also surfaces
  // an asynchronous device error instead of spinning on it)
  const long long want = h->status_seq;
  bool seen = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (long spin = 0;; ++spin) {
    if (__atomic_load_n((const long long*)(h->status_host + 4), __ATOMIC_ACQUIRE) == want) { seen = true; break; }
    if ((spin & 1023) == 1023 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.25) break;
  }
  if (!seen) {
    if (int rc = join_dense(h)) return rc;
    PP_HIP(hipStreamSynchronize(h->stream));
    if (__atomic_load_n((const long long*)(h->status_host + 4), __ATOMIC_ACQUIRE) != want)
      return fail(h, 3, "pp_get_status: status mailbox was not written");
  }
  for (int i = 0; i < 4; ++i) out[i] = (int64_t)h->status_host[i];
  // element growth beyond 1 / u_runtime, if the caller asked for the guard (pp_set_pivot_tolerance), is reported like a
  // breakdown: the host class refreshes the static pivot order from the offending instance and, if that does not help,
  // the inertia-correction loop regularises (MA27 would have re-pivoted).  Every rank sees the same all-reduced count.
  if (h->growth_fatal && out[0] == 0 && h->status_host[5] > 0) out[0] = 2;
  return 0;
}

int pp_get_schur(pp_handle h, double* S_host) {
  if (!h || !h->numeric_done) return fail(h, 3, "pp_get_schur before pp_numeric_local");
  PP_HIP(hipSetDevice(h->device));
  if (int rc = join_dense(h)) return rc;
  PP_HIP(hipMemcpyAsync(S_host, h->S, schur_doubles(h) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PP_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int pp_numeric_local_shifted(pp_handle h, double delta_w, double delta_c) {
  if (!h) return 3;
  h->shift_w = delta_w;
  h->shift_c = delta_c;
  const int rc = pp_numeric_local(h);
  h->shift_w = 0.0;
  h->shift_c = 0.0;
  return rc;
}

}  // extern "C"
