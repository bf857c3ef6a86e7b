// kernels.h -- launch interface between the C ABI (api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include "vrf_core.cuh"

namespace vrf {
// BLS12-381 pairing kernels (k_pairing*.hip): independent of the VRF base field, compiled once
// g2_stride == 0: every item uses the same two G2 points; prep (pairing_prep_bytes() of device memory, nullable)
// then receives their Miller-loop lines, computed once
size_t pairing_prep_bytes();
// layout: 0 = chosen by batch size; otherwise a test asks for one (vrfhip_debug_set, VRFHIP_DEBUG_PAIRING_LAYOUT):
// 1 one item per lane, 2 per DPP quad, 3 per 16-lane row, 4 per wave, 5 per 8 lanes; | 0x100: no prepared lines
enum : int { PAIRING_AUTO = 0, PAIRING_LANE = 1, PAIRING_QUAD = 2, PAIRING_ROW = 3, PAIRING_TRI = 4, PAIRING_OCT = 5,
             PAIRING_NOPREP = 0x100 };
void launch_pairing_check2(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status,
                           hipStream_t st, uint32_t* prep = nullptr, int layout = PAIRING_AUTO);
void launch_pairing_quad_selftest(size_t n, const uint8_t* in, uint8_t* status, hipStream_t st);
void launch_pairing_oct_selftest(size_t n, const uint8_t* in, uint8_t* status, hipStream_t st);
}  // namespace vrf

VRF_NS_BEGIN

constexpr int AUX_WORDS = 32;
constexpr int WS_TABS = 6;

// Workspace, VerifyArgs, ProveArgs, PedersenVerifyArgs: vrf_types.h
constexpr size_t WS_BYTES_PER_ITEM =
    (WS_TABS * WIN_TABLE_WORDS + PROVE_PTS_WORDS + AUX_WORDS) * sizeof(uint32_t) + 1;

// the suites compiled into this field's objects (suite ids: vrf_types.h)
#if VRF_FIELD == 0
#define VRF_DISPATCH_SUITE(suite, CALL)                 \
  do {                                                  \
    if ((suite) == SUITE_JJ) { using S = SuiteJJ; CALL; } \
    else { using S = SuiteBS; CALL; }                   \
  } while (0)
#elif VRF_FIELD == 1
#define VRF_DISPATCH_SUITE(suite, CALL) do { using S = SuiteED; CALL; } while (0)
#else
#define VRF_DISPATCH_SUITE(suite, CALL) do { using S = SuiteBJ; CALL; } while (0)
#endif

// proofs per lane for the inversion-sharing stages: 8 for big batches, fewer when that would leave the
// chip short of waves (the stages are latency-bound below ~2 waves per SIMD)
inline int lanes_k(size_t n, int kmax) {
  int k = kmax;
  while (k > 1 && n / k < (size_t)131072) k >>= 1;
  return k;
}

#include "launchers.inc"

// 32-byte item <-> registers
VRF_HD void load32(uint32_t w[8], const uint8_t* base, size_t i) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(base + i * 32);
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = p[k];
}
VRF_HD void store32(uint8_t* base, size_t i, const uint32_t w[8]) {
  uint32_t* p = reinterpret_cast<uint32_t*>(base + i * 32);
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = w[k];
}

constexpr int BLOCK = 128;
// Verification batches up to this size run their two Straus halves (U and V) in one launch: below two waves
// per SIMD a launch lasts as long as one lane's 128-doubling chain, so the halves overlap instead of queueing
// (measured: 2^14 2.73 -> 1.98 ms, 2^16 4.35 -> 3.25 ms, neutral at 2^17, slightly worse beyond).
constexpr size_t STRAUS_FUSE_MAX_ITEMS = (size_t)1 << 17;
inline dim3 grid_for(size_t threads) { return dim3((unsigned)((threads + BLOCK - 1) / BLOCK)); }

// The kernels that run K proofs per lane (or one heavy item per lane) need 256 VGPRs plus 18..98 AGPRs when
// compiled freely: one wave per SIMD and ~72 % VALU utilisation.  Capped at 256 registers (MINW = 2) they
// spill a little and run two waves per SIMD, which pays once the grid has two waves per SIMD to offer; below
// that (2^16-item batches) the uncapped build is faster.  VRF_LAUNCH_MINW picks the instantiation.
inline bool two_waves_pay(size_t lanes) { return lanes >= size_t(131072); }
// Static LDS of one kernel instantiation.  The compiler moves indexed per-thread arrays into LDS (8 KiB per
// workgroup in k_prove_prepare_multi, 10 KiB in k_prove_mul) and that comes on top of the dynamic bytes
// spread_lds_bytes() reserves: unaccounted, four 38 KiB workgroups no longer fit a CU and a 1024-workgroup grid
// runs in two rounds (measured: prepare 7.8 -> 11.7 ms at 2^20).  The launch macro subtracts it.
template <auto Kernel>
inline size_t static_lds_of() {
  static const size_t bytes = [] {
    hipFuncAttributes fa{};
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(Kernel)) == hipSuccess ? (size_t)fa.sharedSizeBytes
                                                                                          : (size_t)0;
  }();
  return bytes;
}
inline size_t dynamic_lds(size_t want, size_t static_bytes) { return want > static_bytes ? want - static_bytes : 0; }
#define VRF_LAUNCH_MINW(KERNEL, S_, lanes, grid, lds, st, args)                                   \
  do {                                                                                            \
    if (two_waves_pay(lanes))                                                                     \
      hipLaunchKernelGGL((KERNEL<S_, 2>), grid, dim3(BLOCK),                                      \
                         dynamic_lds(lds, static_lds_of<(KERNEL<S_, 2>)>()), st, args);           \
    else                                                                                          \
      hipLaunchKernelGGL((KERNEL<S_, 1>), grid, dim3(BLOCK),                                      \
                         dynamic_lds(lds, static_lds_of<(KERNEL<S_, 1>)>()), st, args);           \
  } while (0)

// Dynamic LDS bytes that make a SMALL grid spread over the whole chip.  The kernels that share inversions
// across K proofs per lane launch n/K lanes: at 2^20 proofs that is 1024 workgroups for 256 CUs, and since
// such a kernel needs ~140 VGPRs (three waves per SIMD fit) the dispatcher packs six workgroups per CU and
// leaves a third of the CUs idle (measured: SQ_WAVE_CYCLES of k_verify_decode at 0.56 of the Straus
// kernels', 75 % VALU utilisation).  Reserving 152 KiB / ceil(workgroups / CUs) of LDS per workgroup caps
// the workgroups a CU accepts at the even share; big grids get 0 (no effect).
inline size_t spread_lds_bytes(size_t workgroups) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    else
      cus = 256;
  }
  const size_t share = (workgroups + cus - 1) / cus;          // even number of workgroups per CU
  if (share == 0 || share > 8) return 0;                      // large grid: occupancy is not the problem
  size_t bytes = (size_t(152) * 1024) / share;                // `share` workgroups fit in the 160 KiB, share + 1 do not
  bytes -= bytes % 1024;
  return bytes > 64 * 1024 ? 64 * 1024 : bytes;               // default dynamic-LDS limit without an attribute
}

VRF_NS_END
