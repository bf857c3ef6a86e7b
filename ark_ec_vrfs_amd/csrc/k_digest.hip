// k_digest.hip -- batch digest kernels (digest.cuh): one leaf hash per item, then levels of 16-ary nodes (DIGEST_FAN).
#include "digest.cuh"
#include "kernels.h"

namespace vrf {

// Minimum waves per SIMD the compiler must leave room for.  1 = no register cap: the leaf kernel then takes 310 registers (one
// wave per SIMD) and is the fastest of the three builds measured at 2^20 items (the part of a batched Pedersen verification
// outside its stage events: 0.94 ms uncapped, 1.18 ms capped at 256 registers, 4.08 ms at 128 -- the SHA-512 message schedule
// spills; profiles/r04/digest_ab.log, tools/gpu_digest_ab.py).
#ifndef DIGEST_MINW
#define DIGEST_MINW 1
#endif

__global__ void __launch_bounds__(BLOCK, DIGEST_MINW) k_digest_leaves(DigestSrc src, size_t n, uint64_t index0, uint8_t* leaves) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  digest_leaf(leaves + i * 32, src, i, index0 + i);
}

__global__ void __launch_bounds__(64, DIGEST_MINW) k_digest_nodes(const uint8_t* children, size_t n_children, uint8_t* nodes) {
  size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  size_t first = t * DIGEST_FAN;
  if (first >= n_children) return;
  size_t count = n_children - first < (size_t)DIGEST_FAN ? n_children - first : (size_t)DIGEST_FAN;
  digest_node(nodes + t * 32, children + first * 32, (uint32_t)count);
}

static size_t level_nodes(size_t n) { return (n + DIGEST_FAN - 1) / DIGEST_FAN; }

size_t digest_ws_bytes(size_t n) {
  size_t total = n, m = n;
  do { m = level_nodes(m); total += m; } while (m > 1);
  return total * 32 + 64;
}

void launch_batch_digest(const DigestSrc& src, size_t n, uint64_t index0, uint8_t* ws, uint8_t* root, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_digest_leaves, grid_for(n), dim3(BLOCK), 0, st, src, n, index0, ws);
  const uint8_t* level = ws;
  uint8_t* next = ws + n * 32;
  size_t m = n;
  do {
    const size_t nodes = level_nodes(m);
    uint8_t* out = nodes == 1 ? root : next;
    hipLaunchKernelGGL(k_digest_nodes, dim3((unsigned)((nodes + 63) / 64)), dim3(64), 0, st, level, m, out);
    level = next;
    next += nodes * 32;
    m = nodes;
  } while (m > 1);
}

}  // namespace vrf
