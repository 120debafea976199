// What a device-side Fiat-Shamir transcript would cost ("optional later step"): the challenge
// of round j is hash_to_field over the transcript so far (fiat-shamir/src/lib.rs:75-98) - SHA-256 expand_message_xmd, a strictly
// serial chain of compressions (the transcript's new bytes + padding, then b_0 -> b_1: >= 3 compressions per round with the
// running midstate kept).  A round's challenge feeds the next round's polynomial, so nothing of a proof's 28 rounds overlaps.
// This measures one SHA-256 compression on ONE lane of a wave that runs alone - the shape such a kernel would have - against the
// host's (not part of the product).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__host__ __device__ inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__host__ __device__ inline void compress(uint32_t st[8], const uint32_t blk[16]) {
  const uint32_t K[64] = {0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
    0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
    0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
    0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
  uint32_t w[64];
  for (int i = 0; i < 16; ++i) w[i] = blk[i];
  for (int i = 16; i < 64; ++i) {
    const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 64; ++i) {
    const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = h + S1 + ch + K[i] + w[i];
    const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
__global__ void chain(uint32_t* out, int n, unsigned long long* ticks) {
  if (threadIdx.x != 0) return;
  uint32_t st[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19}, blk[16];
  for (int i = 0; i < 16; ++i) blk[i] = i * 0x01010101u;
  const unsigned long long t0 = wall_clock64();
  for (int k = 0; k < n; ++k) {
    compress(st, blk);
    blk[0] = st[0];   // the chain: each compression's input depends on the one before
  }
  *ticks = wall_clock64() - t0;   // 100 MHz
  for (int i = 0; i < 8; ++i) out[i] = st[i];
}
int main() {
  uint32_t* d; unsigned long long* dt;
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&dt, 8));
  const int n = 2000;
  hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, n, dt);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, n, dt);
  CK(hipDeviceSynchronize());
  uint32_t h[8]; unsigned long long ticks;
  CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ticks, dt, 8, hipMemcpyDeviceToHost));
  uint32_t st[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19}, blk[16];
  for (int i = 0; i < 16; ++i) blk[i] = i * 0x01010101u;
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < n; ++k) { compress(st, blk); blk[0] = st[0]; }
  const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("device, one lane of a lone wave: %.2f us per SHA-256 compression (%d chained); host (portable C, no SHA-NI): %.3f us; digests %s\n",
         ticks / 100.0 / n, n, host_us / n, h[0] == st[0] && h[7] == st[7] ? "agree" : "DIFFER");
  printf("a round's challenge needs >= 3 compressions (new transcript bytes + padding, b_0, b_1): >= %.1f us per round on the device, x 5 rounds per pass = %.0f us per pass,\n"
         "against ~7 us for the host round trip it would replace\n", 3 * ticks / 100.0 / n, 15 * ticks / 100.0 / n);
  return 0;
}
