// kernels.h -- launch interface between the C ABI (api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include "vrf_core.cuh"

namespace vrf {

using BytesView = BytesViewLite;
VRF_HD void bytes_get(const BytesView& v, size_t i, const uint8_t*& p, uint32_t& n) { bytes_lite_get(v, i, p, n); }

constexpr int AUX_WORDS = 32;
constexpr int WS_TABS = 6;

// per-context device workspace (capacity `cap` items)
struct Workspace {
  uint32_t* tabs;    // [cap][WS_TABS][WIN_TABLE_WORDS]  window tables (verify: 6 GLV tables per proof)
  uint32_t* pts;     // [cap][PROVE_PTS_WORDS]      projective intermediates (verify uses 2*UV_WORDS)
  uint32_t* aux;     // [cap][32]   prove: enc(H) | k | blinding b | kb ; Pedersen verify: challenge c
  uint8_t* flags;    // [cap]                       validity of decoded inputs
};
constexpr size_t WS_BYTES_PER_ITEM =
    (WS_TABS * WIN_TABLE_WORDS + PROVE_PTS_WORDS + AUX_WORDS) * sizeof(uint32_t) + 1;

// suite ids follow vrfhip_suite (include/vrfhip.h)
constexpr int SUITE_BS = 1, SUITE_JJ = 2;
#define VRF_DISPATCH_SUITE(suite, CALL)                 \
  do {                                                  \
    if ((suite) == SUITE_JJ) { using S = SuiteJJ; CALL; } \
    else { using S = SuiteBS; CALL; }                   \
  } while (0)

// proofs per lane for the inversion-sharing stages: 8 for big batches, fewer when that would leave the
// chip short of waves (the stages are latency-bound below ~2 waves per SIMD)
inline int lanes_k(size_t n, int kmax) {
  int k = kmax;
  while (k > 1 && n / k < (size_t)131072) k >>= 1;
  return k;
}

struct VerifyArgs {
  int suite;
  int k_lane;
  size_t n;
  const uint8_t *pk, *h, *gamma, *c, *s;   // affine_in != 0: pk, h, gamma are 64-byte x || y
  int affine_in;
  uint32_t check_mask;                     // CHK_* bits: which decoded points get the subgroup test
  BytesView ad;
  uint8_t* status;
  // keyed verification (key_index != nullptr): pk is the key set's encodings [n_keys][32]; the U half uses the
  // key's context-resident comb instead of per-proof tables
  const uint32_t* key_index;    // [n] index of each proof's key
  const uint32_t* key_combs;    // [n_keys][32][255][PTA_WORDS]
  const uint8_t* key_valid;     // [n_keys] 1 = decodes to a point of the prime-order subgroup
  size_t n_keys;
  Workspace ws;
  DevTables T;
};

struct ProveArgs {
  int suite;
  int k_lane;
  size_t n;
  const uint8_t* sk;
  BytesView msg;
  const uint8_t* h_given;     // nullable
  BytesView ad;
  uint8_t *gamma, *c, *s, *pk_out, *h_out, *status;
  // Pedersen (pedersen != 0): c is unused; pk_out receives pk_com; extra outputs below
  int pedersen;
  uint32_t check_mask;        // CHK_INPUT: subgroup test of a given H
  int out_affine;             // != 0: gamma, pk_out, r_out, ok_out are n x 64 B (x || y, canonical little-endian)
  uint8_t *r_out, *ok_out, *sb_out, *blinding_out;
  unsigned long long* tai_queue;   // 8-byte device counter for k_tai_find (try-and-increment suites)
  Workspace ws;
  DevTables T;
};

struct PedersenVerifyArgs {
  int suite;
  size_t n;
  const uint8_t *h, *gamma, *pk_com, *r, *ok, *s, *sb;
  uint32_t check_mask;        // CHK_INPUT | CHK_OUTPUT | CHK_PROOF
  BytesView ad;
  uint8_t* status;
  Workspace ws;
  DevTables T;
};

// launchers (each defined next to its kernels)
// generator tables: GCOMB_WORDS words each; prefix: 2 * GC_ROWS * GC_SEGS * GC_SEG * 9 words of build scratch.
// gb_xy (device, 128 B): the descriptor's generator and blinding base, x || y little-endian; mont (device, 36 words)
// receives their Montgomery coordinates, flags (device, 2 B) 1 = valid point of the prime-order subgroup.
void launch_init_tables(int suite, uint32_t* g_win, uint32_t* g_comb, uint32_t* b_comb, uint32_t* prefix,
                        const uint8_t* gb_xy, uint32_t* mont, uint8_t* flags, SqrtTables T, hipStream_t st);
// ev: optional 5 events recorded on `st` before stage 1 and after stages 1, 2a, 2b, 3 (profiling)
void launch_ietf_verify(const VerifyArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
void launch_ietf_prove(const ProveArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
void launch_pedersen_verify(const PedersenVerifyArgs& a, hipStream_t st, hipEvent_t* ev = nullptr);
void launch_hash_to_curve(int suite, size_t n, BytesView msg, uint8_t* points, DevTables T, hipStream_t st);
void launch_output_hash(int suite, size_t n, const uint8_t* gamma, uint8_t* hash, DevTables T, hipStream_t st);
void launch_secret_from_seed(int suite, size_t n, const uint8_t* seeds, uint32_t seed_len, uint8_t* sk,
                             uint8_t* pk, DevTables T, hipStream_t st);
// key sets: decode + validate the keys (xy: [n][18] Montgomery words), then build one comb per key.
// prefix: [n_keys * 32][255][9] words of scratch for the build.
void launch_keyset_build(int suite, size_t n_keys, const uint8_t* pks, uint32_t* xy, uint8_t* valid, uint32_t* combs,
                         uint32_t* prefix, DevTables T, hipStream_t st);
void launch_point_validate(int suite, size_t n, const uint8_t* pts, uint8_t* xy, uint8_t* status,
                           uint32_t* tabs, DevTables T, hipStream_t st);
// MSM: ws must hold msm_workspace_bytes(n, groups) bytes; groups = msm_groups(n, n, #CUs)  (msm.cuh)
void launch_msm(int suite, size_t n, const uint8_t* xy, const uint8_t* scalars, uint8_t* out_enc,
                uint8_t* out_xy, uint8_t* status, void* ws, int groups, hipStream_t st);
// g2_stride == 0: every item uses the same two G2 points; prep (pairing_prep_bytes() of device memory, nullable)
// then receives their Miller-loop lines, computed once
size_t pairing_prep_bytes();
void launch_pairing_check2(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status,
                           hipStream_t st, uint32_t* prep = nullptr);
void launch_pairing_quad_selftest(size_t n, const uint8_t* in, uint8_t* status, hipStream_t st);
void launch_fq_mul(size_t n, const uint8_t* a, const uint8_t* b, uint8_t* r, hipStream_t st);
// test primitives: group law, variable-base scalar multiplication (tabs: 2 * WIN_TABLE_WORDS words per item), hashes
void launch_test_point_add(int suite, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* status,
                           DevTables T, hipStream_t st);
void launch_test_scalar_mul(int suite, size_t n, const uint8_t* k, const uint8_t* p, uint8_t* out, uint8_t* status,
                            uint32_t* tabs, DevTables T, hipStream_t st);
void launch_test_hash(int suite, size_t n, BytesView msg, uint8_t* out, int which, DevTables T, hipStream_t st);

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

}  // namespace vrf
