// Part of kernels.hpp (included there, in order): the FOUR-round first pass of a large proof on the int8 matrix cores
// (gram_pass_kernel: the Gram matrix of the tables' bytes AND, since round 5, its reduction to the 81 cells - one launch).
#pragma once

namespace sc {

// ------------------------------------------------------------------------------------
// Four rounds from ONE read of the caller's tables.
//
// The round polynomials of rounds 1..K of the product sumcheck (sum-check-protocol/src/lib.rs:105-112 applied K times)
// are functions of the Gram matrix
//     M[x][y] = sum_rows a[2^K row + x] * b[2^K row + y],        x, y in {0,1}^K (the K lowest index bits),
// between the 2^K "slices" of the two tables: cell (d_1..d_K), d_j in {0, 1, inf}, of the {0,1,inf}^K grid that
// wgrid_pass_kernel accumulates is sum_rows prod_j e_{d_j}(a) * prod_j e_{d_j}(b) with e_0 = the entry at x_j = 0,
// e_1 = at x_j = 1, e_inf = their difference - a signed sum of entries of M.  M is A^T B for the tables seen as matrices of
// 2^(n-K) rows by 2^K entries: a matrix product by nature (as the square of the adjacency matrix is for the triangle
// prover), and with the entries' eight BYTES as separate columns it is an int8 GEMM whose operands are the tables' bytes
// exactly as they lie in HBM:
//     G[8x+i][8y+j] = sum_rows byte_i(a[.. + x]) * byte_j(b[.. + y])         (exact integers)
//     sum_rows a b  = sum_{i,j} 2^(8(i+j)) G[8x+i][8y+j]                      (an integer of <= 64+64+n bits)
// and a Montgomery product a (*) b = a b 2^-64 mod p summed over the rows is that integer times 2^-64 mod p.  The loop
// never sees p (the modulus enters in the epilogue, once per block and partial), and the 27-cell first pass's ~560 VALU
// instructions per 8 entries per table pair (86 % VALU-busy at 0.80 of the HBM peak) become ~45 per 1024 entries.  K = 4: 128 x 128 byte columns,
// 128 MACs per byte read; measured (tools/gram/gram4_bench.hip) the pass then runs at the rate the DMA skeleton alone
// reaches (6.2 TB/s), at K = 5 (256 MACs per byte) the chip lowers its clock under the matrix cores' load
// (1.57 GHz instead of 2.1) and the pass is 11 % slower than at K = 4 - and the fold pass behind a four-round pass
// writes a sixteenth of the tables instead of an eighth, the pass behind that reads a sixteenth.
//
// Signed bytes: v_mfma_i32_32x32x32_i8 multiplies SIGNED bytes; u = (u ^ 0x80) + 128 as a signed byte s plus 128, so
//     sum u u' = sum s s' + 128 (sum s + sum s') + 16384 rows,   sum s = sum u - 128 rows
// with the unsigned column sums sum u taken by v_sad_u8 on the operands the wave holds anyway.
// int32 accumulators: |s s'| <= 2^14, so one accumulator takes 2^16 rows; a launch cuts the rows into `n_partials`
// interleaved slices of at most that many (step t belongs to partial t % n_partials) and a block walks the partials
// blockIdx, blockIdx + gridDim, ...; behind each partial the block reduces its accumulators to the 256 Gram entries mod p
// (gram_reduce_partial) and adds them to the entries of its earlier partials.
//
// Data path: global_load_lds_dwordx4 (1 KiB per wave instruction, no staging registers) into a ring of kGramStages
// stages of 8 KiB per table; ds_read_b64_tr_b8 hands a lane the eight ROWS of one byte column (the contraction index
// of the MFMA is the row, which is the slow index in memory - the transposed LDS read of gfx950 exists for exactly
// this); the XOR swizzle of the 16-byte chunks of a row is applied to the SOURCE address of the DMA (its destination
// is lane-linear) and makes every transposed read conflict-free (eight rows x two adjacent chunks = all 64 banks).
// One workgroup barrier per stage; the operands of stage s + 1 are read while the matrix cores work on stage s.
// 8 waves = 2 (row half of G) x 4 (column quarter); 132 VGPRs at K = 4.
constexpr int kGramStages = 4;
constexpr int kGramTabBytes = 8192;              // bytes per table and stage
constexpr int kGramStageBytes = 2 * kGramTabBytes;
constexpr int kGramThreads = 512;
constexpr int kGramMaxRows = 1 << 16;            // rows one int32 accumulator takes
template <int K1>
struct GramGeo {
  static constexpr int RB = 8 << K1;             // bytes per row (2^K1 entries)
  static constexpr int ROWS = kGramTabBytes / RB;   // rows per stage: 64 (K1 = 4) or 32
  static constexpr int KSUB = ROWS / 32;         // MFMA k-steps per stage
  static constexpr int MB = RB / 64, NBK = RB / 128;   // 32-column blocks per wave along the rows / columns of G
  static constexpr int CPR = RB / 16;            // 16-byte chunks per row
  static constexpr int kEntriesPerStep = kGramTabBytes / 8;
  static constexpr int kMaxSteps = kGramMaxRows / ROWS;   // steps per partial at most
};
// chunk position inside a row of the LDS image
template <int K1>
__device__ __forceinline__ int gram_swz(int row, int chunk) {
  if constexpr (K1 == 5) return (((chunk >> 1) ^ (row & 7)) << 1) | (chunk & 1);
  else return (((chunk >> 1) ^ ((row >> 1) & 3)) << 1) | (chunk & 1);
}
typedef int gram_v2i __attribute__((ext_vector_type(2)));
typedef int gram_v4i __attribute__((ext_vector_type(4)));
typedef int gram_v16i __attribute__((ext_vector_type(16)));

#define SC_TR8(R, AD, OFF) "ds_read_b64_tr_b8 %" #R ", %" #AD " offset:%" #OFF "\n\t"
// the twelve transposed reads of a stage (eight of a, four of b), stage offset as an immediate; no wait
#define SC_GRAM_READ12(R, OFF)                                                                                                     \
  asm volatile(SC_TR8(0, 12, 24) SC_TR8(1, 13, 24) SC_TR8(2, 14, 24) SC_TR8(3, 15, 24) SC_TR8(4, 16, 24) SC_TR8(5, 17, 24)          \
               SC_TR8(6, 18, 24) SC_TR8(7, 19, 24) SC_TR8(8, 20, 24) SC_TR8(9, 21, 24) SC_TR8(10, 22, 24)                           \
               "ds_read_b64_tr_b8 %11, %23 offset:%24"                                                                             \
               : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]), "=&v"(R[5]), "=&v"(R[6]), "=&v"(R[7]), "=&v"(R[8]),   \
                 "=&v"(R[9]), "=&v"(R[10]), "=&v"(R[11])                                                                           \
               : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(aa[4]), "v"(aa[5]), "v"(aa[6]), "v"(aa[7]), "v"(ba[0]), "v"(ba[1]),   \
                 "v"(ba[2]), "v"(ba[3]), "n"(OFF)                                                                                   \
               : "memory")
// the registers are valid behind this wait (the operands tie their uses to it)
#define SC_GRAM_WAIT12(R)                                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                               \
               : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]), "+v"(R[8]), "+v"(R[9]),   \
                 "+v"(R[10]), "+v"(R[11])                                                                                           \
               :                                                                                                                    \
               : "memory")

// Words per block of the hand-off rows (>= 3^4 cells)
constexpr int kGramRowWords = 128;

// A, B: the tables as bytes.  Step t (8 KiB of each table) belongs to partial t % n_partials; steps_per_partial is a
// multiple of kGramStages and at most GramGeo::kMaxSteps.
// NT: the DMA loads carry the nontemporal hint (tables far larger than the 256 MiB Infinity Cache are read once: with
// allocating loads the same loop runs at 6.1 TB/s instead of 7.0, tools/gram/gram4_bench.hip -DGRAM_AUX=2)
//
// Round 5: ONE launch.  Until then the blocks stored their int32 accumulators (66 KiB per partial, 17 MB per launch) and a
// second kernel of 256 blocks added them: 12-14 us + a kernel boundary on every proof and every shard.  Now each block
// reduces what it holds (gram_reduce_partial: accumulators -> the 256 Gram entries mod p, ~2 us on the block's own LDS),
// turns the entries into the 3^K1 cells (linear: gram_entries_to_cells), stores 81 words, draws a ticket, and the block
// that draws the last one adds the rows and hands the cells on exactly as wgrid_pass_kernel<F, K1> does (publish_cells:
// the wide mailbox, the in-kernel exchange with the peers, or split limbs for a collective).  rows: [gridDim][kGramRowWords]
// words; *ticket rests at zero.
template <class F, int K1, bool NT>
__global__ void __launch_bounds__(kGramThreads)
gram_pass_kernel(F f, const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, unsigned n_partials, unsigned steps_per_partial,
                 u64* __restrict__ rows, unsigned* __restrict__ ticket, WgOut out) {
  typedef GramGeo<K1> G;
  constexpr int NS = kGramStages;
  extern __shared__ __attribute__((aligned(16))) unsigned char gram_lds[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glob_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mh = wave >> 2, nq = wave & 3;
  const int h = lane >> 5, g2 = (lane >> 4) & 1, ll = lane & 15, q = ll >> 1, p = ll & 1;
  // DMA: 16 instructions of 1 KiB per stage, two per wave (i < 8: table a).  Lane L of an instruction lands at
  // row r0 + L / CPR, chunk position L % CPR, and fetches the chunk that belongs there.
  size_t src_off[2];
  unsigned dst_off[2];
  const unsigned char* tab_of[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = 2 * wave + u, tab = i >> 3, piece = i & 7;
    const int row = piece * (1024 / G::RB) + lane / G::CPR, pos = lane % G::CPR;
    src_off[u] = (size_t)row * G::RB + 16 * gram_swz<K1>(row, pos);
    dst_off[u] = (unsigned)(tab * kGramTabBytes + piece * 1024);
    tab_of[u] = tab ? B : A;
  }
  auto issue = [&](size_t step, unsigned stage_off) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((glob_ptr_t)(tab_of[u] + step * (size_t)kGramTabBytes + src_off[u]),
                                       (lds_ptr_t)(gram_lds + stage_off + dst_off[u]), 16, 0, NT ? 2 : 0);
  };
  // transposed reads: lane 2q + p of a 16-lane group supplies row q, bytes 8p .. 8p + 7 of the group's 16 columns and
  // receives the eight rows of column (lane & 15) - operand register pair t covers rows 16 h + 8 t + (0..7) of the k-step
  auto tr_addr = [&](int tab, int m0, int ks, int t) -> unsigned {
    const int row = 32 * ks + 16 * h + 8 * t + q, chunk = (m0 >> 4) + g2;
    return (unsigned)(size_t)(lds_ptr_t)gram_lds + (unsigned)(tab * kGramTabBytes + row * G::RB + 16 * gram_swz<K1>(row, chunk) + 8 * p);
  };
  static_assert(G::MB * G::KSUB * 2 == 8 && G::NBK * G::KSUB * 2 == 4, "eight + four transposed reads per stage");
  unsigned aa[8], ba[4];   // index: ((block * KSUB) + ks) * 2 + t
#pragma unroll
  for (int a = 0; a < G::MB; ++a)
#pragma unroll
    for (int ks = 0; ks < G::KSUB; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) aa[(a * G::KSUB + ks) * 2 + t] = tr_addr(0, (G::RB / 2) * mh + 32 * a, ks, t);
#pragma unroll
  for (int b = 0; b < G::NBK; ++b)
#pragma unroll
    for (int ks = 0; ks < G::KSUB; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) ba[(b * G::KSUB + ks) * 2 + t] = tr_addr(1, (G::RB / 4) * nq + 32 * b, ks, t);

  gram_v16i acc[G::MB][G::NBK];
  unsigned su_a[G::MB], su_b[G::NBK];
  auto compute = [&](gram_v2i (&r)[12]) {
    gram_v4i fa[G::MB][G::KSUB], fb[G::NBK][G::KSUB];
#pragma unroll
    for (int a = 0; a < G::MB; ++a)
#pragma unroll
      for (int ks = 0; ks < G::KSUB; ++ks) {
        const int i = (a * G::KSUB + ks) * 2;
        fa[a][ks] = gram_v4i{r[i].x, r[i].y, r[i + 1].x, r[i + 1].y};
      }
#pragma unroll
    for (int b = 0; b < G::NBK; ++b)
#pragma unroll
      for (int ks = 0; ks < G::KSUB; ++ks) {
        const int i = 8 + (b * G::KSUB + ks) * 2;
        fb[b][ks] = gram_v4i{r[i].x, r[i].y, r[i + 1].x, r[i + 1].y};
      }
    // unsigned column sums (each column is owned by one wave: the a columns by the waves of column quarter 0, ...)
    if (nq == 0) {
#pragma unroll
      for (int a = 0; a < G::MB; ++a)
#pragma unroll
        for (int ks = 0; ks < G::KSUB; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) su_a[a] = __builtin_amdgcn_sad_u8((unsigned)fa[a][ks][e], 0u, su_a[a]);
    }
    if (mh == 0) {
#pragma unroll
      for (int b = 0; b < G::NBK; ++b)
#pragma unroll
        for (int ks = 0; ks < G::KSUB; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) su_b[b] = __builtin_amdgcn_sad_u8((unsigned)fb[b][ks][e], 0u, su_b[b]);
    }
#pragma unroll
    for (int ks = 0; ks < G::KSUB; ++ks) {
#pragma unroll
      for (int b = 0; b < G::NBK; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) fb[b][ks][e] ^= 0x80808080;
#pragma unroll
      for (int a = 0; a < G::MB; ++a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) fa[a][ks][e] ^= 0x80808080;
#pragma unroll
        for (int b = 0; b < G::NBK; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a][ks], fb[b][ks], acc[a][b], 0, 0, 0);
      }
    }
  };
  const size_t stride = n_partials, my_steps = steps_per_partial;
  // top of step s: step s + 1 becomes readable (this wave's part has landed; barrier: everyone's, and everyone holds
  // the operands of step s in registers), the stage of step s is refilled with step s + NS
#define SC_GRAM_TOP(S, STAGE)                                                                         \
  do {                                                                                                \
    if ((S) + NS - 1 < my_steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 2)) : "memory");  \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    __builtin_amdgcn_s_barrier();                                                                     \
    if ((S) + NS < my_steps) issue(first + ((S) + NS) * stride, (STAGE) * kGramStageBytes);           \
  } while (0)
  u64 m_acc = 0;   // thread e < 256: Gram entry e = 16 x + y of this block's partials, mod p
  for (unsigned part = blockIdx.x; part < n_partials; part += gridDim.x) {
    const size_t first = part;
#pragma unroll
    for (int a = 0; a < G::MB; ++a)
#pragma unroll
      for (int b = 0; b < G::NBK; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0;
#pragma unroll
    for (int a = 0; a < G::MB; ++a) su_a[a] = 0;
#pragma unroll
    for (int b = 0; b < G::NBK; ++b) su_b[b] = 0;
    gram_v2i r0[12], r1[12];
    __builtin_amdgcn_s_barrier();   // (a second partial: everyone is done with the stages of the previous one)
#pragma unroll
    for (int s = 0; s < NS; ++s) issue(first + (size_t)s * stride, s * kGramStageBytes);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    SC_GRAM_READ12(r0, 0);
    SC_GRAM_WAIT12(r0);
    for (size_t s = 0; s < my_steps; s += NS) {
      const bool more = s + NS < my_steps;
      SC_GRAM_TOP(s, 0);     SC_GRAM_READ12(r1, 1 * kGramStageBytes); compute(r0); SC_GRAM_WAIT12(r1);
      SC_GRAM_TOP(s + 1, 1); SC_GRAM_READ12(r0, 2 * kGramStageBytes); compute(r1); SC_GRAM_WAIT12(r0);
      SC_GRAM_TOP(s + 2, 2); SC_GRAM_READ12(r1, 3 * kGramStageBytes); compute(r0); SC_GRAM_WAIT12(r1);
      SC_GRAM_TOP(s + 3, 3);
      if (more) SC_GRAM_READ12(r0, 0);
      compute(r1);
      if (more) SC_GRAM_WAIT12(r0);
    }
    // ---- this partial's accumulators -> the 256 Gram entries mod p, added to the block's running entries ----
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (cdna_hip_programming.md
    // section 3): register e of block a holds G[m][n], m = 64 mh + 32 a + 8 (e >> 2) + (e & 3) + 4 h, n = 32 nq + (lane & 31),
    // i.e. Gram entry (x, y) = (m / 8, n / 8), limb product (i, j) = (m % 8, n % 8) = ((e & 3) + 4 h, lane & 7).
    //   T_ij = sum_rows u_i u'_j = G + 128 (Su_i + Su'_j) - 16384 rows   (>= 0, < 2^32: a partial has <= 2^16 rows)
    //   entry = sum_ij 2^(8(i+j)) T_ij  (< 2^153)  ->  ONE reduction (wide_get, which also supplies the 2^-64 of the Montgomery products)
    static_assert(K1 == 4, "the epilogue's thread mapping is written for 128 x 128 byte columns");
    long long* const ca = reinterpret_cast<long long*>(gram_lds + 32768);        // [RB]: 128 Su_i - 16384 rows
    long long* const cb = ca + G::RB;                                            // [RB]: 128 Su'_j
    u64* const vbuf = reinterpret_cast<u64*>(gram_lds);                          // [256 entries][16 = (h, j)], rotated by the entry
    __syncthreads();   // every wave has its last operands in registers: the stages are free
    const long long rows_part = (long long)my_steps * G::ROWS;
    if (nq == 0) {
#pragma unroll
      for (int a = 0; a < G::MB; ++a) {
        const unsigned t = su_a[a] + (unsigned)__shfl_xor((int)su_a[a], 32, 64);   // (the two k-halves of a column live in lanes l, l + 32)
        if (h == 0) ca[(G::RB / 2) * mh + 32 * a + (lane & 31)] = 128ll * (long long)t - 16384ll * rows_part;
      }
    }
    if (mh == 0) {
#pragma unroll
      for (int b = 0; b < G::NBK; ++b) {
        const unsigned t = su_b[b] + (unsigned)__shfl_xor((int)su_b[b], 32, 64);
        if (h == 0) cb[(G::RB / 4) * nq + 32 * b + (lane & 31)] = 128ll * (long long)t;
      }
    }
    __syncthreads();
    {
      const int n = (G::RB / 4) * nq + (lane & 31), y = n >> 3, jj = n & 7;
      const long long cbv = cb[n];
#pragma unroll
      for (int a = 0; a < G::MB; ++a)
#pragma unroll
        for (int xq = 0; xq < 4; ++xq) {
          const int m0 = (G::RB / 2) * mh + 32 * a + 8 * xq + 4 * h;
          u64 v = 0;
#pragma unroll
          for (int il = 0; il < 4; ++il) v += (u64)((long long)acc[a][0][4 * xq + il] + ca[m0 + il] + cbv) << (8 * il);   // < 2^57
          const int e = (m0 >> 3) * 16 + y;
          vbuf[e * 16 + ((8 * h + jj + e) & 15)] = v;
        }
    }
    __syncthreads();
    if (tid < 256) {
      // sum_k v_k 2^(32 h + 8 j), k = 8 h + j: three words
      u64 w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const u64 v = vbuf[tid * 16 + ((k + tid) & 15)];
        const int sh = 32 * (k >> 3) + 8 * (k & 7);
        if (sh < 64) {
          const u64 lo = v << sh, hi = sh ? v >> (64 - sh) : 0;
          u64 t;
          const bool c = __builtin_add_overflow(w0, lo, &t);
          w0 = t;
          const bool c2 = __builtin_add_overflow(w1, hi, &t);
          const bool c3 = __builtin_add_overflow(t, (u64)(c ? 1 : 0), &t);
          w1 = t;
          w2 += (c2 ? 1 : 0) + (c3 ? 1 : 0);
        } else {
          const int s2 = sh - 64;
          const u64 lo = v << s2, hi = s2 ? v >> (64 - s2) : 0;
          u64 t;
          const bool c = __builtin_add_overflow(w1, lo, &t);
          w1 = t;
          w2 += hi + (c ? 1 : 0);
        }
      }
      m_acc = f.add(m_acc, f.wide_get(w0, w1, (u32)w2));   // (w2 < 2^25)
    }
  }
#undef SC_GRAM_TOP
  // ---- the block's entries -> its cells -> the hand-off ----
  // M -> cells, one variable at a time: the pair (bit j of x, bit j of y) becomes the digit d_j in {0, 1, inf}:
  // d = 0 / 1 pins both bits; inf is (a1 - a0)(b1 - b0) = M11 - M10 - M01 + M00 in that variable.  Index of the working
  // array after j variables: ((x >> j) * (X >> j) + (y >> j)) * 3^j + digits, digits = sum_{i<j} d_i 3^(j-1-i) - after K1
  // variables the cell index itself (variable 0 on the slowest axis).  (Unrolled: every division is by a constant.)
  // Linear in M, so it commutes with the sum over the blocks: done per block, the rows that cross the chip are 81 words.
  constexpr int X = 1 << K1;
  constexpr int kPow3[6] = {1, 3, 9, 27, 81, 243};
  constexpr int cells = kPow3[K1];
  u64* const Mb = reinterpret_cast<u64*>(gram_lds + 36864);   // (behind ca / cb; a block with no partial left the LDS untouched)
  u64* const Wb = Mb + X * X;
  __shared__ int last_flag;
  __syncthreads();
  if (tid < X * X) Mb[tid] = m_acc;
  __syncthreads();
  u64* src = Mb;
  u64* dst = Wb;
#pragma unroll
  for (int j = 0; j < K1; ++j) {
    const int side = X >> j, half = side / 2, pow3 = kPow3[j], n_out = half * half * pow3 * 3;
    for (int o = tid; o < n_out; o += kGramThreads) {
      const int d = o % 3, dig = (o / 3) % pow3, yx = o / (3 * pow3), yr = yx % half, xr = yx / half;
      auto at = [&](int bx, int by) { return src[((2 * xr + bx) * side + (2 * yr + by)) * pow3 + dig]; };
      u64 v;
      if (d == 0) v = at(0, 0);
      else if (d == 1) v = at(1, 1);
      else v = f.sub(f.add(at(1, 1), at(0, 0)), f.add(at(1, 0), at(0, 1)));
      dst[(xr * half + yr) * (pow3 * 3) + dig * 3 + d] = v;
    }
    __syncthreads();
    u64* t = src; src = dst; dst = t;
  }
  u64 total = tid < cells ? src[tid] : 0;
  if (gridDim.x > 1) {
    // hand-off as in finish_pass (Guideline 16, R1): write-through stores, every storing wave drains, ticket; the last block acquires
    if (tid < cells) __hip_atomic_store(rows + (size_t)blockIdx.x * kGramRowWords + tid, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (t == gridDim.x - 1) ? 1 : 0;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    // thread = (slice of the blocks, cell): a wave's load reads one contiguous run of a row.  EVERY load of a thread is in flight
    // at once (64 at 256 blocks): the rows were written by other XCDs and come from memory, ~1.5 us a round trip - with batches
    // of sixteen the last block spent four of them here (the fused pass measured 7 us over the loop alone, r05 first run)
    constexpr int kSlices = kGramThreads / kGramRowWords;   // 4
    constexpr int kU = 64;
    u64* const fin = Mb;                                    // [kSlices][kGramRowWords]
    {
      const int slice = tid / kGramRowWords, c = tid % kGramRowWords;
      u64 part = 0;
      if (c < cells) {
        const int n_blocks = gridDim.x;
        for (int b0 = slice; b0 < n_blocks; b0 += kSlices * kU) {
          u64 x[kU];
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            const int b = b0 + u * kSlices;
            x[u] = (b < n_blocks) ? __hip_atomic_load(rows + (size_t)b * kGramRowWords + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
          }
#pragma unroll
          for (int w = kU / 2; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) x[u] = f.add(x[u], x[u + w]);
          part = f.add(part, x[0]);
        }
      }
      fin[slice * kGramRowWords + c] = part;
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every block has drawn: back to rest
    __syncthreads();
    total = 0;
    if (tid < cells) {
#pragma unroll
      for (int q = 0; q < kSlices; ++q) total = f.add(total, fin[q * kGramRowWords + tid]);
    }
  }
  publish_cells<cells>(out, total);   // (unsharded: the wide mailbox; sharded: as a grid pass's cells)
}
#undef SC_GRAM_READ12
#undef SC_GRAM_WAIT12
#undef SC_TR8

}  // namespace sc
