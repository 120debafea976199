// Part of kernels.hpp (included there, in order): triangle_counting::G, the matrix squares.
#pragma once

namespace sc {

// ------------------------------------------------------------------------------------
// triangle_counting::G (triangle-counting/src/lib.rs:22-166): g(X,Y,Z) = f(X,Y) f(Y,Z) f(X,Z).

// P[(z << k) | x] = sum_y f[(y << k) | x] * f[(z << k) | y]: the square of the adjacency MLE's
// matrix.  sum_{y} f1(x,y) f2(y,z) is multilinear in x and in z, so the k x-rounds of the
// sumcheck are a product-of-two-tables sumcheck on (P, f3) - one n^3 pass here instead of an
// n^3 pass per round (the reference's to_univariate walks all 2^(3k) evaluations, :138-165).
// Consecutive lanes own consecutive x: the column read is coalesced, the row read a broadcast.
template <class F>
__global__ void __launch_bounds__(kBlock)
matsq_kernel(F f, const u64* __restrict__ T, int k, u64* __restrict__ P, size_t z_begin, size_t z_rows) {
  const size_t n = (size_t)1 << k, first = z_begin * n, total = (z_begin + z_rows) * n;   // rows z_begin .. of P
  for (size_t o = first + (size_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (size_t)gridDim.x * kBlock) {
    const size_t z = o >> k, x = o & (n - 1);
    typename F::Acc acc;
    f.acc_zero(acc);
    for (size_t y = 0; y < n; ++y) f.acc_mac(acc, T[(y << k) | x], T[(z << k) | y]);
    P[o] = f.acc_get(acc);
  }
}

// The same square, LDS-tiled, for n >= 64: a block of 256 threads owns a 64 x 64 tile of P and walks y in
// steps of 32; per step it stages A[y][x0..x0+64) and, transposed, Bt[y][z0..z0+64) = f[(z << k) | y] in LDS
// (16 KiB each), and every thread accumulates a 4 x 4 patch: 4 ds_read_b128 per 16 lazy multiply-adds
// instead of 2 global loads per multiply-add, so the kernel runs at the VALU rate of the products
// (15 instructions each) rather than at the L1 rate of the naive form.
template <class F>
__global__ void __launch_bounds__(kBlock)
matsq_tiled_kernel(F f, const u64* __restrict__ T, int k, u64* __restrict__ P, size_t z_begin, size_t z_rows,
                   const unsigned* __restrict__ only_if /* null, or: run only if this word is non-zero */) {
  constexpr int TS = 64, KT = 32;
  if (only_if && *only_if == 0) return;   // the table was 0/1: matsq_mfma_kernel has done the work
  __shared__ ull2 lds_a[KT * TS / 2];   // A[yy][xx], 16 KiB
  __shared__ u64 lds_b[KT * TS];        // Bt[yy][zz], 16 KiB
  const size_t n = (size_t)1 << k;
  const int tiles = (int)(n / TS), tiles_z = (int)(z_rows / TS);   // rows z_begin .. z_begin + z_rows of P (a rank's share)
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  for (int tile = blockIdx.x; tile < tiles * tiles_z; tile += gridDim.x) {
    const size_t x0 = (size_t)(tile % tiles) * TS, z0 = z_begin + (size_t)(tile / tiles) * TS;
    typename F::Acc acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) f.acc_zero(acc[j][i]);
    for (size_t y0 = 0; y0 < n; y0 += KT) {
      __syncthreads();   // the previous step's reads are done
      // A tile: 32 rows of 64 entries; thread t loads 16-byte pieces (row t/32 + 8i, piece t%32)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (threadIdx.x >> 5) + 8 * i, pc = threadIdx.x & 31;
        lds_a[row * (TS / 2) + pc] = reinterpret_cast<const ull2*>(T + ((y0 + row) << k) + x0)[pc];
      }
      // B tile: rows z0 + zz hold 32 consecutive y; thread t loads piece t%16 of row t/16 + 16i and
      // stores it transposed
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int zz = (threadIdx.x >> 4) + 16 * i, pc = threadIdx.x & 15;
        const ull2 v = reinterpret_cast<const ull2*>(T + ((z0 + zz) << k) + y0)[pc];
        lds_b[(2 * pc) * TS + zz] = v.x;
        lds_b[(2 * pc + 1) * TS + zz] = v.y;
      }
      __syncthreads();
      for (int yy = 0; yy < KT; ++yy) {
        const ull2 a01 = lds_a[yy * (TS / 2) + 2 * tx], a23 = lds_a[yy * (TS / 2) + 2 * tx + 1];
        const ull2 b01 = reinterpret_cast<const ull2*>(lds_b + yy * TS)[2 * ty],
                   b23 = reinterpret_cast<const ull2*>(lds_b + yy * TS)[2 * ty + 1];
        const u64 a[4] = {a01.x, a01.y, a23.x, a23.y}, b[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) f.acc_mac(acc[j][i], a[i], b[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ull2 o0 = {f.acc_get(acc[j][0]), f.acc_get(acc[j][1])}, o1 = {f.acc_get(acc[j][2]), f.acc_get(acc[j][3])};
      ull2* dst = reinterpret_cast<ull2*>(P + ((z0 + 4 * ty + j) << k) + x0 + 4 * tx);
      dst[0] = o0;
      dst[1] = o1;
    }
  }
}

// The square of a 0/1 matrix on the matrix cores.  G::new_adj_matrix (triangle-counting/src/lib.rs:32-51) builds the
// three tables from a Vec<bool>: every entry is 0 or 1, so P[z][x] = sum_y T[z][y] T[y][x] is a COUNT (<= n <= 2^15)
// and an int8 x int8 -> int32 MFMA computes it exactly - this is a matrix product by nature, not a reshaped stream.
//  1. matsq_bytes_kernel: the table as bytes, row-major (T8[z][y]) and transposed (T8t[x][y] = T[y][x]) so that both
//     MFMA operands are 16 contiguous bytes per lane; any entry that is neither 0 nor 1 raises `flag`.
//  2. matsq_mfma_kernel (if the flag stayed down): one wave per 32 x 32 tile of P, v_mfma_i32_32x32x32_i8 over y in
//     steps of 32, operands straight from the (L2-resident) byte tables; count -> Montgomery word (count * R^2 * R^-1).
//  3. matsq_tiled_kernel (if the flag went up; launched behind the other two either way, no host round trip): the
//     generic field-valued square.
// The hardware pairs element e of lane (r, h)'s A fragment with element e of lane (r', h)'s B fragment; both are loaded
// with the same y = y0 + 16 h + e, so whatever k order the instruction uses inside a step the sum is over the same y.
// C/D layout (cdna_hip_programming.md section 3): col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));
template <class F>
__global__ void __launch_bounds__(kBlock)
matsq_bytes_kernel(F f, const u64* __restrict__ T, int k, unsigned char* __restrict__ T8, unsigned char* __restrict__ T8t,
                   unsigned* __restrict__ flag) {
  constexpr int TS = 64;
  __shared__ unsigned char tile[TS][TS + 16];   // rows 16-byte aligned (80 bytes)
  const size_t n = (size_t)1 << k;
  const int tps = (int)(n / TS);
  const u64 one = f.one();
  int bad = 0;
  for (int tid = blockIdx.x; tid < tps * tps; tid += gridDim.x) {
    const size_t r0 = (size_t)(tid / tps) * TS, c0 = (size_t)(tid % tps) * TS;
    __syncthreads();   // the previous tile has been written out
#pragma unroll
    for (int i = 0; i < TS * TS / kBlock; ++i) {
      const int row = (threadIdx.x >> 6) + 4 * i, col = threadIdx.x & 63;
      const u64 v = T[((r0 + row) << k) | (c0 + col)];
      bad |= (v != 0 && v != one) ? 1 : 0;
      tile[row][col] = (v == one) ? 1 : 0;
    }
    __syncthreads();
    const int rr = threadIdx.x >> 2, q = threadIdx.x & 3;   // 64 rows x 4 chunks of 16 bytes
    *reinterpret_cast<uint4*>(T8 + (r0 + rr) * n + c0 + 16 * q) = *reinterpret_cast<const uint4*>(&tile[rr][16 * q]);
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = (unsigned)tile[16 * q + 4 * j][rr] | ((unsigned)tile[16 * q + 4 * j + 1][rr] << 8) | ((unsigned)tile[16 * q + 4 * j + 2][rr] << 16) |
             ((unsigned)tile[16 * q + 4 * j + 3][rr] << 24);
    *reinterpret_cast<uint4*>(T8t + (c0 + rr) * n + r0 + 16 * q) = uint4{w[0], w[1], w[2], w[3]};
  }
  if (bad) atomicOr(flag, 1u);
}
template <class F>
__global__ void __launch_bounds__(kBlock)
matsq_mfma_kernel(F f, const unsigned char* __restrict__ T8, const unsigned char* __restrict__ T8t, int k, u64* __restrict__ P,
                  size_t z_begin, size_t z_rows, const unsigned* __restrict__ flag) {
  if (*flag != 0) return;   // not a 0/1 table: the generic kernel behind this launch does the work
  const size_t n = (size_t)1 << k;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, r = lane & 31, h = lane >> 5;
  const size_t tiles_x = n / 32, n_tiles = (z_rows / 32) * tiles_x;
  const u64 r2 = f.r_squared();
  for (size_t tid = (size_t)blockIdx.x * (kBlock / kWave) + wave; tid < n_tiles; tid += (size_t)gridDim.x * (kBlock / kWave)) {
    const size_t z0 = z_begin + (tid / tiles_x) * 32, x0 = (tid % tiles_x) * 32;
    const unsigned char* ap = T8 + (z0 + r) * n + 16 * h;    // row z0 + r of T:  T[z][y0 + 16 h + e]
    const unsigned char* bp = T8t + (x0 + r) * n + 16 * h;   // column x0 + r of T: T[y0 + 16 h + e][x]
    mfma_v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t y0 = 0;
    for (; y0 + 128 <= n; y0 += 128) {   // four steps of loads in flight
      mfma_v4i a[4], b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = *reinterpret_cast<const mfma_v4i*>(ap + y0 + 32 * s);
        b[s] = *reinterpret_cast<const mfma_v4i*>(bp + y0 + 32 * s);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], acc, 0, 0, 0);
    }
    for (; y0 < n; y0 += 32)
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const mfma_v4i*>(ap + y0), *reinterpret_cast<const mfma_v4i*>(bp + y0), acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      P[((z0 + row) << k) | (x0 + r)] = f.mul((u64)(unsigned)acc[reg], r2);   // count -> Montgomery word
    }
  }
}

// Round sums H(0), H(1), H(inf) of G in ANY state (xv, yv, zv variables left), by walking every
// remaining (x, y, z) like the reference does: the generic SumCheckPolynomial::to_univariate.
// Two of the three copies hold the current variable (pairs p, q), the third a constant c.
template <class F>
__global__ void __launch_bounds__(kBlock)
tri_sums_kernel(F f, const u64* __restrict__ f1, const u64* __restrict__ f2, const u64* __restrict__ f3, int xv, int yv,
                int zv, PassOut out) {
  __shared__ u64 lds[(kBlock / kWave) * 3];
  __shared__ int lds_flag;
  const size_t total = (size_t)1 << (xv + yv + zv - 1);
  typename F::Acc acc[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) f.acc_zero(acc[s]);
  for (size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock) {
    u64 p0, p1, q0, q1, c;
    if (xv > 0) {
      const size_t xh = t & (((size_t)1 << (xv - 1)) - 1), y = (t >> (xv - 1)) & (((size_t)1 << yv) - 1),
                   z = t >> (xv - 1 + yv);
      const size_t i1 = (y << xv) | (2 * xh), i3 = (z << xv) | (2 * xh);
      p0 = f1[i1]; p1 = f1[i1 + 1]; q0 = f3[i3]; q1 = f3[i3 + 1]; c = f2[(z << yv) | y];
    } else if (yv > 0) {
      const size_t yh = t & (((size_t)1 << (yv - 1)) - 1), z = t >> (yv - 1);
      const size_t i2 = (z << yv) | (2 * yh);
      p0 = f1[2 * yh]; p1 = f1[2 * yh + 1]; q0 = f2[i2]; q1 = f2[i2 + 1]; c = f3[z];
    } else {
      p0 = f2[2 * t]; p1 = f2[2 * t + 1]; q0 = f3[2 * t]; q1 = f3[2 * t + 1]; c = f1[0];
    }
    f.acc_mac(acc[0], f.mul(p0, q0), c);
    f.acc_mac(acc[1], f.mul(p1, q1), c);
    f.acc_mac(acc[2], f.mul(f.sub(p1, p0), f.sub(q1, q0)), c);
  }
  u64 res[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) res[s] = f.acc_get(acc[s]);
  block_reduce<F, 3>(f, res, lds);
  finish_pass<F, 3>(f, out, res[0], &lds_flag);
}

// G::to_evaluations (:138-165): out[((x << yv) | y) << zv | z] = f1[(y<<xv)|x] f2[(z<<yv)|y] f3[(z<<xv)|x]
template <class F>
__global__ void __launch_bounds__(kBlock)
tri_to_evaluations_kernel(F f, const u64* __restrict__ f1, const u64* __restrict__ f2, const u64* __restrict__ f3, int xv,
                          int yv, int zv, u64* __restrict__ out) {
  const size_t total = (size_t)1 << (xv + yv + zv);
  for (size_t o = (size_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (size_t)gridDim.x * kBlock) {
    const size_t z = o & (((size_t)1 << zv) - 1), y = (o >> zv) & (((size_t)1 << yv) - 1), x = o >> (zv + yv);
    out[o] = f.mul(f.mul(f1[(y << xv) | x], f2[(z << yv) | y]), f3[(z << xv) | x]);
  }
}

}  // namespace sc
