// tai_find.cuh -- the work-queue counter search of try-and-increment hash-to-curve (`Input::new` on the TAI suites,
// /root/reference src/lib.rs:15-16): shared by the provers (k_prove.hip), `vrfhip_hash_to_curve_batch` (k_misc.hip) and
// verification from alpha (k_verify.hip).  Each translation unit instantiates the kernel for its own suites.
#pragma once
#include "kernels.h"
#include <algorithm>

VRF_NS_BEGIN

// lanes of a wave that must hold a candidate before the wave runs the Jacobi symbol
#ifndef TAI_READY_MIN
#define TAI_READY_MIN 48
#endif

// stage 0 for try-and-increment suites: find every item's first counter whose candidate decodes.  A lane that
// loops until ITS item succeeds makes the wave pay the maximum over 64 geometric trials (about 7 attempts for
// an expected 2).  Here the lanes of a persistent wave draw items from a global queue: a lane whose attempt
// succeeded records the counter and takes the next item at once, so a wave performs about two attempts per
// item and the only idle lanes are those of the last few iterations of the whole grid.
// An attempt has a cheap half (hash the counter, y < q, denominator non-zero) and an expensive one (the Jacobi symbol,
// four times the hash).  Where the cheap half rejects often -- Baby-JubJub: q = 0.378 * 2^255, five hashes per point --
// a wave that ran both halves every trip would run the symbol for the one lane in three that needs it; so a lane that
// passed the cheap half WAITS (ready), the others keep hashing, and the wave runs the symbol once three quarters of its
// lanes are ready (or nobody is left to hash).  Suites whose candidates rarely fail the cheap half (JubJub 0.91,
// Ed25519 1.0) take the symbol every trip, as before.
template <class S>
__global__ void __launch_bounds__(64, 2) k_tai_find(size_t n, BytesView msg, uint8_t* ctr_out, SqrtTables T,
                                                 unsigned long long* queue) {
  constexpr size_t NONE = ~size_t(0);
  const int lane = threadIdx.x;
  size_t item = NONE;
  uint32_t next = 0;                                       // the item's next counter to hash
  bool ready = false, has2 = false;                        // candidate A awaits its Jacobi symbol; B is the one after it
  uint32_t ca = 0, cb = 0;
  FeN wa = fe_zero(), wb = fe_zero();
  bool drained = false;                                    // the queue has no items left (wave-uniform)
  while (true) {
    const bool need = item == NONE && !drained;
    const unsigned long long mask = __ballot(need);
    if (mask) {
      const uint32_t cnt = (uint32_t)__popcll(mask);
      const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(queue, (unsigned long long)cnt);
      base = __shfl(base, 0, 64);
      if (need && base + rank < n) { item = (size_t)(base + rank); next = 0; ready = false; has2 = false; }
      if (base + cnt >= n) drained = true;
    }
    if (!__any(item != NONE)) break;                       // every lane idle and nothing left to draw
    // The cheap half.  A lane that already holds a candidate does not sit the trip out: it hashes on and keeps the NEXT
    // counter that passes (B), so that a failed symbol of A is followed by B's at once (the order of the counters is kept:
    // the first one whose candidate decodes still wins).
    if (item != NONE && !has2 && next < 256) {
      const uint8_t* m; uint32_t len;
      bytes_get(msg, item, m, len);
      FeN w;
      if (tai_attempt_candidate<S>(w, m, len, next, T)) {
        if (!ready) { wa = w; ca = next; ready = true; }
        else { wb = w; cb = next; has2 = true; }
      }
      ++next;
    }
    if (item != NONE && !ready && next >= 256) {           // out of counters with nothing to test:
      ctr_out[item] = 255;                                 // hash_to_curve_tai reports the failure
      item = NONE;
    }
    const unsigned long long rmask = __ballot(ready), hmask = __ballot(item != NONE && !ready);
    if (__popcll(rmask) >= TAI_READY_MIN || (rmask && !hmask)) {      // the expensive half, for the lanes that hold a candidate
      if (ready) {
        if (fe_is_square_or_zero(wa, T) || ca == 255) {
          ctr_out[item] = (uint8_t)ca;
          item = NONE; ready = false; has2 = false;
        } else if (has2) {
          wa = wb; ca = cb; has2 = false;                  // B moves up; the lane stays ready
        } else {
          ready = false;
        }
      }
    }
  }
}

// the launch: the queue counter is reset on the stream first; persistent waves, as many as are RESIDENT at once (the kernel
// holds 256 registers: two waves per SIMD) -- later waves of a larger grid would find the queue empty.  Measured: 44 of 64
// lanes active per vector instruction on JubJub, 28 on bandersnatch_sw (profiles/r04/lane_utilisation_bench.log): a wave's
// last trips run with ever fewer lanes, each lane's last item ending on its own attempt.
template <class S>
inline void launch_tai_find_t(size_t n, BytesView msg, uint8_t* ctr_out, const SqrtTables& T, unsigned long long* queue,
                              hipStream_t st) {
  (void)hipMemsetAsync(queue, 0, sizeof(unsigned long long), st);
  static const size_t resident = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return (size_t)cus * 4 * 2;
  }();
  const size_t waves = std::min<size_t>((n + 63) / 64, resident);
  hipLaunchKernelGGL(k_tai_find<S>, dim3((unsigned)waves), dim3(64), 0, st, n, msg, ctr_out, T, queue);
}

VRF_NS_END
