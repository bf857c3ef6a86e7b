// api.hip -- the C ABI of libvrfhip (include/vrfhip.h): context, HBM workspace, launches.
// Host code only orchestrates; every field / curve / hash operation runs in the HIP kernels.
#include "../../include/vrfhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "digest.cuh"
#include "msm.cuh"
#include "msm_g1.h"
#include "p256.h"
#include "bsw.h"
#include "constants_bsw.gen.h"

namespace vrf {
// k_pairing_oct.hip: per-item G2 points as two kernels (lines -> HBM -> Miller loop + final exponentiation)
size_t pairing_oct_lines_bytes(size_t items);
void launch_pairing_check2_oct_split(size_t n, const uint8_t* g1, const uint8_t* g2, size_t g2_stride, uint8_t* status, void* ws,
                                     size_t chunk, hipStream_t st);
constexpr size_t PAIRING_OCT_SPLIT_MIN_ITEMS = 1;         // every size: 4.6-4.7 ms for 8 .. 4096 items against 10.9 per quad
constexpr size_t PAIRING_OCT_SPLIT_CHUNK = size_t(1) << 15;   // items per pass: 45.7 KB of lines each, 1.5 GB of workspace
// k_pairing_row.hip: the selftest operands through the row-distributed tower (bls12_row.cuh); ORs 64 / 128 into status[i]
void launch_pairing_row_selftest(size_t n, const uint8_t* in, uint8_t* status, hipStream_t st);
// The kernel objects of the other base fields (field.h): this file is compiled for field 0 (kernels.h declared its
// launchers in vrf::f_bls381fr) and reaches the 2^255 - 19 and BN254 Fr builds through the same declarations.
inline namespace f_bls381fr {
#include "te_sw_map.inc"
}
namespace f_25519 {
#include "launchers.inc"
#include "te_sw_map.inc"
}
namespace f_bn254fr {
#include "launchers.inc"
#include "te_sw_map.inc"
}
}  // namespace vrf
using namespace vrf;
// CALL in the kernel objects of the context's base field
#define FIELD_CALL(ctx_, CALL)                          \
  do {                                                  \
    switch ((ctx_)->field) {                            \
      case 1: vrf::f_25519::CALL; break;                \
      case 2: vrf::f_bn254fr::CALL; break;              \
      default: vrf::f_bls381fr::CALL; break;            \
    }                                                   \
  } while (0)
// Proofs per lane in the verifiers' inversion-sharing stages (decode, finish; the kernels take any K <= VERIFY_K).  With
// the exponentiation a^(q-2) an inversion cost as much as a third of a proof's decode and K = 8 paid; since fe_inv is the
// divsteps loop (~18 k instructions) the extra waves of a small K are worth more than the shared inversion: measured at
// 2^20, IETF verify 40.3 (K = 8) / 39.7 ms (K = 2), the batched Pedersen decode on JubJub 22.1 / 20.7 ms.
constexpr int VERIFY_K_POLICY = 2;

namespace {

thread_local std::string g_last_error;

int32_t fail(int32_t code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess)                                                                 \
      return fail(e__ == hipErrorOutOfMemory ? VRFHIP_ERR_OOM : VRFHIP_ERR_HIP,            \
                  std::string(#expr) + ": " + hipGetErrorString(e__));                     \
  } while (0)

constexpr size_t DEFAULT_CHUNK = size_t(1) << 20;
const char BSW_UNSUPPORTED_MSG[] =
    "not available for the bandersnatch_sw suite (secret keys, hash-to-curve, output hash, point validation, the IETF scheme incl. "
    "verification from alpha, the x || y forms, key sets, the Pedersen scheme per proof and batched, MSM, and the pairing / G1 "
    "entry points are; the twisted-Edwards test primitives are not)";

}  // namespace

struct vrfhip_ctx {
  int device = 0;
  vrfhip_suite suite = VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2;   // selects the compiled arithmetic (= desc.curve)
  int field = 0;                     // base field of the curve: which build of the kernels runs (field.h)
  vrfhip_suite_desc desc{};
  hipStream_t stream = nullptr;      // used by the host-pointer entry points
  std::recursive_mutex mu;
  // shared tables
  uint32_t* d_sqrt_p = nullptr;
  uint8_t* d_sqrt_lut = nullptr;
  uint32_t* d_g_win = nullptr;
  uint32_t* d_g_comb = nullptr;
  uint32_t* d_b_comb = nullptr;
  DevTables T{};
  // workspace
  void* d_ws = nullptr;
  size_t ws_cap = 0;                 // items
  size_t chunk_limit = DEFAULT_CHUNK; // largest number of items processed per launch group
  size_t ws_bytes = 0;
  Workspace ws{};
  // MSM workspace (grown on demand) and device facts
  void* d_msm_ws = nullptr;
  size_t msm_ws_bytes = 0;
  unsigned long long* d_queue = nullptr;   // work-queue counter of k_tai_find
  uint32_t* d_pair_prep = nullptr;         // Miller-loop lines of shared G2 points (pairing check, SRS case)
  int cus = 256;
  uint32_t flags = 0;                      // VRFHIP_FLAG_PREVALIDATED_* (vrfhip_ctx_set_flags)
  uint8_t* d_h2c_ctr = nullptr;            // try-and-increment counters of vrfhip_hash_to_curve_batch: one byte per item of a chunk
  size_t h2c_cap = 0;
  bool have_blinding = true;               // false: the descriptor's blinding base is all-zero -- a suite without the Pedersen scheme
  bool has_pedersen() const { return sw ? d_p256_comb_b != nullptr : have_blinding; }
  // test / tuning knobs (vrfhip_debug_set; nothing in the product reads the environment)
  int dbg_pairing_layout = 0;              // kernels.h PAIRING_*: 0 = by batch size
  int pipe_first_log2 = 17, pipe_chunk_log2 = 18;   // host-pointer pipeline: first chunk, later chunks (items, log2)
  int dbg_prove_k = 0;                     // > 0: proofs per lane in the provers' prepare / finish stages (default: lanes_k)
  int dbg_p256_msm_groups = 0;             // > 0: point groups per window of the secp256r1 MSM (default: p256::msm_groups)
  uint32_t check_mask() const { return ~flags & (uint32_t)VRFHIP_FLAG_PREVALIDATED_ALL; }
  // secp256r1 (`suites::secp256r1`): short-Weierstrass law, Sec1 wire format (33-byte points, big-endian scalars), SHA-256.
  // Its kernels (k_p256.hip) have their own tables and workspace; the entry points below branch on `sw` where the
  // Edwards suites go through FIELD_CALL, and the byte widths of the arrays come from the two functions that follow.
  bool sw = false;
  uint32_t* d_p256_comb = nullptr;
  uint32_t* d_p256_comb_b = nullptr;       // comb of the Pedersen blinding base (nullptr: the descriptor's base is all-zero)
  p256::Ws p256_ws{};
  // `suites::bandersnatch_sw`: a Bandersnatch (twisted-Edwards) context whose codec is arkworks' short-Weierstrass one --
  // 33-byte points on the wire; tables, workspace and the arithmetic stages are the Edwards suite's (bsw.h, k_bsw.hip)
  bool bsw = false;
  size_t pt_bytes() const { return (sw || bsw) ? 33 : 32; }        // one compressed point on the wire
  size_t hash_bytes() const { return sw ? 32 : 64; }      // `Output::hash`: the suite hasher's output
  size_t prove_point_bytes() const { return (flags & VRFHIP_FLAG_PROVE_POINTS_AFFINE) ? 64 : pt_bytes(); }
  bool coords_mont256() const { return (flags & VRFHIP_FLAG_COORDS_MONT256) != 0; }
  // optional per-stage timing (hipEvents on the launch stream), see vrfhip_ctx_profile
  bool prof = false;
  std::vector<hipEvent_t> prof_ev;   // 5 per launch group
  // staging for the host-pointer entry points
  void* d_stage = nullptr;
  size_t stage_bytes = 0;
  // pipelined host path (HostPipe below): a second stream for the copies and a ring of two pinned staging slots
  hipStream_t copy_stream = nullptr;
  uint8_t* h_pin = nullptr;                // 2 slots of pin_slot_bytes (hipHostMalloc)
  size_t pin_slot_bytes = 0;
  hipEvent_t ev_copied[2] = {nullptr, nullptr};
};

// `Public` keys with context-resident fixed-base tables (keyed verification)
struct vrfhip_keyset {
  vrfhip_ctx* ctx = nullptr;
  size_t n_keys = 0;
  uint8_t* d_enc = nullptr;       // [n_keys][32] encodings (hashed by the challenge)
  uint8_t* d_valid = nullptr;     // [n_keys]
  uint32_t* d_combs = nullptr;    // [n_keys][32][255][27]   (secp256r1: [n_keys][rows][128][28])
  uint32_t* d_aff = nullptr;      // secp256r1 only: [n_keys][18] Montgomery affine coordinates
  int rows = 0;                   // secp256r1 only: comb rows per key = challenge_len + 1
  size_t bytes = 0;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int32_t alloc_workspace(vrfhip_ctx* ctx, size_t cap) {
  if (ctx->d_ws) {
    HIP_TRY(hipFree(ctx->d_ws));
    ctx->d_ws = nullptr;
    ctx->ws_cap = 0;
    ctx->ws_bytes = 0;
  }
  if (ctx->sw) {
    const size_t total = p256::ws_bytes(cap);
    HIP_TRY(hipMalloc(&ctx->d_ws, total));
    ctx->p256_ws = p256::ws_carve(ctx->d_ws, cap);
    ctx->ws_cap = cap;
    ctx->ws_bytes = total;
    return VRFHIP_SUCCESS;
  }
  size_t tabs_b = cap * WS_TABS * WIN_TABLE_WORDS * sizeof(uint32_t);
  size_t pts_b = cap * PROVE_PTS_WORDS * sizeof(uint32_t);
  size_t aux_b = cap * AUX_WORDS * sizeof(uint32_t);
  size_t flags_b = (cap + 255) & ~size_t(255);
  size_t total = tabs_b + pts_b + aux_b + flags_b;
  HIP_TRY(hipMalloc(&ctx->d_ws, total));
  uint8_t* p = static_cast<uint8_t*>(ctx->d_ws);
  ctx->ws.tabs = reinterpret_cast<uint32_t*>(p); p += tabs_b;
  ctx->ws.pts = reinterpret_cast<uint32_t*>(p); p += pts_b;
  ctx->ws.aux = reinterpret_cast<uint32_t*>(p); p += aux_b;
  ctx->ws.flags = p;
  ctx->ws_cap = cap;
  ctx->ws_bytes = total;
  return VRFHIP_SUCCESS;
}

// workspace for a batch of `items`: at most chunk_limit items; larger batches are processed in
// chunks of ws_cap
int32_t ensure_workspace(vrfhip_ctx* ctx, size_t items) {
  size_t want = std::min(items, ctx->chunk_limit);
  if (want <= ctx->ws_cap) return VRFHIP_SUCCESS;
  return alloc_workspace(ctx, want);
}

int32_t ensure_stage(vrfhip_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->stage_bytes) return VRFHIP_SUCCESS;
  if (ctx->d_stage) {
    HIP_TRY(hipMemset(ctx->d_stage, 0, ctx->stage_bytes));     // may have held secret keys (prove entry points)
    HIP_TRY(hipFree(ctx->d_stage));
    ctx->d_stage = nullptr;
    ctx->stage_bytes = 0;
  }
  HIP_TRY(hipMalloc(&ctx->d_stage, bytes));
  ctx->stage_bytes = bytes;
  return VRFHIP_SUCCESS;
}

int32_t ensure_msm_workspace(vrfhip_ctx* ctx, size_t need) {
  if (need <= ctx->msm_ws_bytes) return VRFHIP_SUCCESS;
  if (ctx->d_msm_ws) HIP_TRY(hipFree(ctx->d_msm_ws));
  ctx->d_msm_ws = nullptr;
  ctx->msm_ws_bytes = 0;
  HIP_TRY(hipMalloc(&ctx->d_msm_ws, need));
  ctx->msm_ws_bytes = need;
  return VRFHIP_SUCCESS;
}

// bump allocator over the staging buffer
struct Stage {
  uint8_t* base;
  size_t off = 0;
  explicit Stage(void* b) : base(static_cast<uint8_t*>(b)) {}
  static size_t pad(size_t n) { return (n + 255) & ~size_t(255); }
  uint8_t* take(size_t n) {
    uint8_t* p = base + off;
    off += pad(n);
    return p;
  }
};

// five fresh events for one launch group when profiling is on, else nullptr
hipEvent_t* prof_events(vrfhip_ctx* ctx) {
  if (!ctx->prof) return nullptr;
  size_t base = ctx->prof_ev.size();
  ctx->prof_ev.resize(base + 5);
  for (int i = 0; i < 5; ++i)
    if (hipEventCreate(&ctx->prof_ev[base + i]) != hipSuccess) {
      ctx->prof_ev.resize(base);
      return nullptr;
    }
  return ctx->prof_ev.data() + base;
}

// ---- pipelined host -> device path of the host-pointer entry points -------------------------------------------
// A batch handed over in host memory is cut into chunks of PIPE_CHUNK items; chunk k + 1 travels while chunk k is
// verified: its slices of the caller's arrays are gathered into one of two pinned staging slots by the calling thread
// (skipped for arrays that already are pinned: hipHostMalloc / hipHostRegister / vrfhip_host_alloc memory goes out by DMA
// as it lies), sent on the context's copy stream, and the compute stream waits for the chunk's event only.  Before, the
// whole batch went through five pageable hipMemcpyAsync calls in front of the first kernel (45.1 ms against 40.7 ms per
// 2^20 verifications); now only the first chunk's copy is exposed.
constexpr size_t PIPE_CHUNK = size_t(1) << 18;    // 2^18 items: 10.3 ms of verification, 40 MiB of wire data
struct PipeArr {
  const uint8_t* h;      // host array
  uint8_t* d;            // device array (same item order)
  size_t w;              // bytes per item
};
bool is_pinned_host(const void* p) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}
int32_t pipe_prepare(vrfhip_ctx* ctx, size_t slot_bytes) {
  if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  for (hipEvent_t& e : ctx->ev_copied)
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (slot_bytes > ctx->pin_slot_bytes) {
    if (ctx->h_pin) {
      HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
      std::memset(ctx->h_pin, 0, 2 * ctx->pin_slot_bytes);
      HIP_TRY(hipHostFree(ctx->h_pin));
      ctx->h_pin = nullptr;
      ctx->pin_slot_bytes = 0;
    }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), 2 * slot_bytes, hipHostMallocDefault));
    ctx->pin_slot_bytes = slot_bytes;
  }
  return VRFHIP_SUCCESS;
}
// items [base, base + m) of every array -> device, through pinned slot `slot`; the compute stream is made to wait for them
int32_t pipe_send(vrfhip_ctx* ctx, const PipeArr* arrs, const bool* pinned, int na, size_t base, size_t m, int slot) {
  HIP_TRY(hipEventSynchronize(ctx->ev_copied[slot]));           // the slot's previous chunk has left host memory
  uint8_t* stage = ctx->h_pin + (size_t)slot * ctx->pin_slot_bytes;
  size_t off = 0;
  for (int a = 0; a < na; ++a) {
    const size_t bytes = m * arrs[a].w;
    const uint8_t* src = arrs[a].h + base * arrs[a].w;
    if (!pinned[a]) {
      std::memcpy(stage + off, src, bytes);
      src = stage + off;
      off += Stage::pad(bytes);
    }
    HIP_TRY(hipMemcpyAsync(arrs[a].d + base * arrs[a].w, src, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  }
  HIP_TRY(hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream));
  HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_copied[slot], 0));
  return VRFHIP_SUCCESS;
}

BytesView make_view(const uint8_t* blob, const uint32_t* off, uint32_t len, bool shared) {
  BytesView v;
  v.blob = blob;
  v.off = off;
  v.len = len;
  v.stride = shared ? 0u : len;
  return v;
}

// size in bytes of a host-side variable-length blob
size_t blob_bytes(size_t n, const uint32_t* off, uint32_t len, bool shared) {
  if (off) return off[n];
  return shared ? len : n * (size_t)len;
}

}  // namespace

extern "C" {

int32_t vrfhip_abi_version(void) { return 144; }

const char* vrfhip_last_error(void) { return g_last_error.c_str(); }

static const char NO_BLINDING_MSG[] =
    "the suite descriptor carries no Pedersen blinding base (PedersenSuite::BLINDING_BASE): only Bandersnatch's is pinned by an "
    "upstream vector; supply the suite's constant in vrfhip_suite_desc.blinding_base";

// The nothing-up-my-sleeve points rounds 1-3 shipped as default blinding bases (tools/gen_constants.py: try-and-increment on
// "vrfhip-<suite>-blinding-base").  They are NOT upstream's constants: proofs made with them do not verify under ark-vrf.
// Tests and bench legs that want the Pedersen scheme on a suite whose constant is unpinned ask for them by name.
int32_t vrfhip_test_blinding_base(vrfhip_suite suite, uint8_t out_xy[64]) {
  if (!out_xy) return fail(VRFHIP_ERR_BAD_ARG, "out is NULL");
  uint8_t g[64];
  bool have = false;
  if (suite == VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2) have = vrf::f_bls381fr::field_default_points(SUITE_BS, g, out_xy);
  else if (suite == VRFHIP_SUITE_JUBJUB_SHA512_TAI) have = vrf::f_bls381fr::field_default_points(SUITE_JJ, g, out_xy);
  else if (suite == VRFHIP_SUITE_ED25519_SHA512_TAI) have = vrf::f_25519::field_default_points(SUITE_ED, g, out_xy);
  else if (suite == VRFHIP_SUITE_BABY_JUBJUB_SHA512_TAI) have = vrf::f_bn254fr::field_default_points(SUITE_BJ, g, out_xy);
  else if (suite == VRFHIP_SUITE_SECP256R1_SHA256_TAI) { p256::default_blinding_base(out_xy); have = true; }
  else if (suite == VRFHIP_SUITE_BANDERSNATCH_SW_SHA512_TAI) { std::memcpy(out_xy, vrfk_bsw::BB_XY, 64); have = true; }
  if (!have) return fail(VRFHIP_ERR_UNSUPPORTED, "unsupported suite");
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_suite_desc_default(vrfhip_suite suite, vrfhip_suite_desc* out) {
  if (!out) return fail(VRFHIP_ERR_BAD_ARG, "out is NULL");
  std::memset(out, 0, sizeof *out);
  out->struct_size = (uint32_t)sizeof *out;
  out->challenge_len = 32;
  auto put = [](uint8_t* dst, uint32_t& len, const char* s) {
    len = (uint32_t)std::strlen(s);
    std::memcpy(dst, s, len);
  };
  bool have = false;
  if (suite == VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2) {
    out->curve = VRFHIP_CURVE_BANDERSNATCH;
    put(out->suite_id, out->suite_id_len, "Bandersnatch_SHA-512_ELL2");
    put(out->h2c_dst, out->h2c_dst_len, "ECVRF_Bandersnatch_XMD:SHA-512_ELL2_RO_Bandersnatch_SHA-512_ELL2");
    have = vrf::f_bls381fr::field_default_points(SUITE_BS, out->generator, out->blinding_base);
  } else if (suite == VRFHIP_SUITE_JUBJUB_SHA512_TAI) {
    out->curve = VRFHIP_CURVE_JUBJUB;
    put(out->suite_id, out->suite_id_len, "JubJub_SHA-512_TAI");
    have = vrf::f_bls381fr::field_default_points(SUITE_JJ, out->generator, out->blinding_base);
    std::memset(out->blinding_base, 0, 64);          // upstream's constant is not pinned here: no Pedersen scheme by default
  } else if (suite == VRFHIP_SUITE_ED25519_SHA512_TAI) {
    out->curve = VRFHIP_CURVE_ED25519;
    out->challenge_len = 16;                         // upstream: `CHALLENGE_LEN = 16` (RFC 9381 cLen of the edwards25519 suites)
    put(out->suite_id, out->suite_id_len, "Ed25519_SHA-512_TAI");
    have = vrf::f_25519::field_default_points(SUITE_ED, out->generator, out->blinding_base);
    std::memset(out->blinding_base, 0, 64);
  } else if (suite == VRFHIP_SUITE_BABY_JUBJUB_SHA512_TAI) {
    out->curve = VRFHIP_CURVE_BABY_JUBJUB;
    put(out->suite_id, out->suite_id_len, "BabyJubJub_SHA-512_TAI");
    have = vrf::f_bn254fr::field_default_points(SUITE_BJ, out->generator, out->blinding_base);
    std::memset(out->blinding_base, 0, 64);
  }
  else if (suite == VRFHIP_SUITE_SECP256R1_SHA256_TAI) {
    out->curve = VRFHIP_CURVE_SECP256R1;
    out->challenge_len = 16;                         // RFC 9381 cLen of ECVRF-P256-SHA256-TAI
    out->suite_id[0] = 0x01;                         // RFC 9381 suite_string
    out->suite_id_len = 1;
    p256::default_generator(out->generator);
    // blinding_base stays all-zero: upstream's constant is not known here (vrfhip_test_blinding_base has a placeholder)
    have = true;
  }
  else if (suite == VRFHIP_SUITE_BANDERSNATCH_SW_SHA512_TAI) {
    out->curve = VRFHIP_CURVE_BANDERSNATCH_SW;
    put(out->suite_id, out->suite_id_len, "Bandersnatch_SW_SHA-512_TAI");
    std::memcpy(out->generator, vrfk_bsw::G_XY, 64);          // te_sw_map image of the twisted-Edwards suite's generator
    // blinding_base stays all-zero: upstream's constant is not pinned here (vrfhip_test_blinding_base: the image of the
    // twisted-Edwards suite's)
    have = true;
  }
  if (!have) return fail(VRFHIP_ERR_UNSUPPORTED, "unsupported suite");
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_ctx_create(vrfhip_suite suite, int32_t device, vrfhip_ctx** out) {
  if (!out) return fail(VRFHIP_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  vrfhip_suite_desc d;
  int32_t rc = vrfhip_suite_desc_default(suite, &d);
  if (rc) return rc;
  return vrfhip_ctx_create_desc(&d, device, out);
}

int32_t vrfhip_ctx_get_desc(const vrfhip_ctx* ctx, vrfhip_suite_desc* out) {
  if (!ctx || !out) return fail(VRFHIP_ERR_BAD_ARG, "NULL argument");
  *out = ctx->desc;
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_ctx_create_desc(const vrfhip_suite_desc* desc, int32_t device, vrfhip_ctx** out) {
  if (!out) return fail(VRFHIP_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (!desc) return fail(VRFHIP_ERR_BAD_ARG, "desc is NULL");
  if (desc->struct_size != sizeof(vrfhip_suite_desc)) return fail(VRFHIP_ERR_BAD_ARG, "desc.struct_size mismatch");
  if (desc->curve < VRFHIP_CURVE_BANDERSNATCH || desc->curve > VRFHIP_CURVE_BANDERSNATCH_SW)
    return fail(VRFHIP_ERR_UNSUPPORTED, "unsupported curve");
  if (desc->curve == VRFHIP_CURVE_SECP256R1 && desc->flags) return fail(VRFHIP_ERR_UNSUPPORTED, "secp256r1 takes no suite flags");
  const bool bsw = desc->curve == VRFHIP_CURVE_BANDERSNATCH_SW;
  if (bsw && desc->flags) return fail(VRFHIP_ERR_UNSUPPORTED, "bandersnatch_sw takes no suite flags");
  if (desc->challenge_len == 0 || desc->challenge_len > 32) return fail(VRFHIP_ERR_UNSUPPORTED, "challenge_len must be 1..32");
  if (desc->flags & ~(uint32_t)VRFHIP_SUITE_FLAG_ALL) return fail(VRFHIP_ERR_UNSUPPORTED, "unknown suite flag bits");
  if (desc->suite_id_len == 0 || desc->suite_id_len > sizeof desc->suite_id)
    return fail(VRFHIP_ERR_BAD_ARG, "suite_id_len out of range");
  const bool ell2 = desc->curve == VRFHIP_CURVE_BANDERSNATCH;
  if (ell2 && (desc->h2c_dst_len == 0 || desc->h2c_dst_len > sizeof desc->h2c_dst))
    return fail(VRFHIP_ERR_BAD_ARG, "h2c_dst_len out of range");
  // one built-in suite per curve: its id names the compiled arithmetic (SUITE_* in vrf_types.h follow vrfhip_suite)
  // (the short-Weierstrass presentation of Bandersnatch runs the twisted-Edwards suite's arithmetic: bsw_core.cuh)
  const vrfhip_suite suite = bsw ? VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2 : (vrfhip_suite)desc->curve;
  static_assert((int)VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2 == SUITE_BS && (int)VRFHIP_SUITE_JUBJUB_SHA512_TAI == SUITE_JJ &&
                (int)VRFHIP_SUITE_ED25519_SHA512_TAI == SUITE_ED && (int)VRFHIP_SUITE_BABY_JUBJUB_SHA512_TAI == SUITE_BJ &&
                (int)VRFHIP_CURVE_ED25519 == SUITE_ED && (int)VRFHIP_CURVE_BABY_JUBJUB == SUITE_BJ, "suite / curve ids");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(VRFHIP_ERR_NO_DEVICE, "no HIP device visible: libvrfhip has no CPU path");
  if (device < 0 || device >= count) return fail(VRFHIP_ERR_BAD_ARG, "device index out of range");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(VRFHIP_ERR_HIP, "hipSetDevice failed");
  vrfhip_ctx* ctx = new vrfhip_ctx();
  ctx->device = device;
  ctx->suite = suite;
  ctx->field = suite_field((int)suite);
  ctx->bsw = bsw;
  ctx->desc = *desc;
  auto cleanup = [&](int32_t rc) {
    vrfhip_ctx_destroy(ctx);
    return rc;
  };
#define HIP_TRY_C(expr)                                                                   \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess)                                                                \
      return cleanup(fail(VRFHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__))); \
  } while (0)
  HIP_TRY_C(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      ctx->cus = cus;
  }
  if (desc->curve == VRFHIP_CURVE_SECP256R1) {
    // the short-Weierstrass suite: its one table is the generator's comb; its strings travel in ctx->T.sq.str like the others'
    ctx->sw = true;
    SuiteStr hs{};
    hs.suite_id_len = desc->suite_id_len;
    hs.challenge_len = desc->challenge_len;
    for (size_t i = 0; i < desc->suite_id_len; ++i) hs.suite_id_w[i >> 3] |= (uint64_t)desc->suite_id[i] << (56 - 8 * (i & 7));
    ctx->T.sq.str = hs;
    uint8_t* d_gen = nullptr;
    HIP_TRY_C(hipMalloc(&ctx->d_p256_comb, p256::comb_bytes()));
    HIP_TRY_C(hipMalloc(&ctx->d_queue, 256));
    HIP_TRY_C(hipMalloc(&d_gen, 256));
    uint8_t gen_ok = 0;
    hipError_t e0 = hipMemcpyAsync(d_gen, desc->generator, 64, hipMemcpyHostToDevice, ctx->stream);   // ordered with the kernel below
    p256::launch_init_comb(ctx->d_p256_comb, d_gen, d_gen + 128, ctx->stream);
    hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ctx->stream);
    hipError_t e3 = hipMemcpy(&gen_ok, d_gen + 128, 1, hipMemcpyDeviceToHost);
    (void)hipFree(d_gen);
    HIP_TRY_C(e0); HIP_TRY_C(e1); HIP_TRY_C(e2); HIP_TRY_C(e3);
    if (!gen_ok) return cleanup(fail(VRFHIP_ERR_BAD_ARG, "desc.generator is not a point of the curve"));
    bool have_b = false;
    for (size_t i = 0; i < 64; ++i) have_b = have_b || desc->blinding_base[i] != 0;
    if (have_b) {                               // `PedersenSuite::BLINDING_BASE`; all-zero = a suite without the Pedersen scheme
      uint8_t* d_b = nullptr;
      HIP_TRY_C(hipMalloc(&ctx->d_p256_comb_b, p256::comb_bytes()));
      HIP_TRY_C(hipMalloc(&d_b, 256));
      uint8_t b_ok = 0;
      hipError_t f0 = hipMemcpyAsync(d_b, desc->blinding_base, 64, hipMemcpyHostToDevice, ctx->stream);
      p256::launch_init_comb(ctx->d_p256_comb_b, d_b, d_b + 128, ctx->stream);
      hipError_t f1 = hipGetLastError(), f2 = hipStreamSynchronize(ctx->stream);
      hipError_t f3 = hipMemcpy(&b_ok, d_b + 128, 1, hipMemcpyDeviceToHost);
      (void)hipFree(d_b);
      HIP_TRY_C(f0); HIP_TRY_C(f1); HIP_TRY_C(f2); HIP_TRY_C(f3);
      if (!b_ok) return cleanup(fail(VRFHIP_ERR_BAD_ARG, "desc.blinding_base is not a point of the curve"));
    }
    *out = ctx;
    return VRFHIP_SUCCESS;
  }
  size_t sqrt_p_bytes = 0, lut_bytes = 0;
  const uint32_t* h_sqrt_p = nullptr;
  const uint8_t* h_sqrt_lut = nullptr;
  switch (ctx->field) {
    case 1: h_sqrt_p = vrf::f_25519::field_sqrt_p(&sqrt_p_bytes); h_sqrt_lut = vrf::f_25519::field_sqrt_lut(&lut_bytes); break;
    case 2: h_sqrt_p = vrf::f_bn254fr::field_sqrt_p(&sqrt_p_bytes); h_sqrt_lut = vrf::f_bn254fr::field_sqrt_lut(&lut_bytes); break;
    default: h_sqrt_p = vrf::f_bls381fr::field_sqrt_p(&sqrt_p_bytes); h_sqrt_lut = vrf::f_bls381fr::field_sqrt_lut(&lut_bytes); break;
  }
  const size_t comb_bytes = GCOMB_WORDS * sizeof(uint32_t);            // 56.6 MB per generator (16-bit signed windows)
  const size_t prefix_bytes = (size_t)2 * GC_ROWS * GC_SEGS * GC_SEG * NL * sizeof(uint32_t);
  uint32_t* d_prefix = nullptr;
  uint8_t* d_init = nullptr;          // 128 B points | 36 words Montgomery | 2 B flags
  HIP_TRY_C(hipMalloc(&ctx->d_sqrt_p, sqrt_p_bytes));
  HIP_TRY_C(hipMalloc(&ctx->d_sqrt_lut, lut_bytes));
  HIP_TRY_C(hipMalloc(&ctx->d_g_win, 2 * WIN_TABLE_WORDS * sizeof(uint32_t)));
  HIP_TRY_C(hipMalloc(&ctx->d_g_comb, comb_bytes));
  HIP_TRY_C(hipMalloc(&ctx->d_b_comb, comb_bytes));
  HIP_TRY_C(hipMalloc(&ctx->d_queue, 256));
  HIP_TRY_C(hipMalloc(&ctx->d_pair_prep, pairing_prep_bytes()));
  HIP_TRY_C(hipMemset(ctx->d_pair_prep, 0, pairing_prep_bytes()));
  HIP_TRY_C(hipMemcpyAsync(ctx->d_sqrt_p, h_sqrt_p, sqrt_p_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY_C(hipMemcpyAsync(ctx->d_sqrt_lut, h_sqrt_lut, lut_bytes, hipMemcpyHostToDevice, ctx->stream));
  // the suite's byte strings, packed big-endian into the 64-bit words SHA-512 absorbs; they travel in the kernel
  // arguments (SuiteStr, fe.cuh)
  SuiteStr hs{};
  auto pack = [](uint64_t* w, const uint8_t* b, size_t n) {
    for (size_t i = 0; i < n; ++i) w[i >> 3] |= (uint64_t)b[i] << (56 - 8 * (i & 7));
  };
  hs.suite_id_len = desc->suite_id_len;
  hs.challenge_len = desc->challenge_len;
  hs.flags = desc->flags;
  pack(hs.suite_id_w, desc->suite_id, desc->suite_id_len);
  if (ell2) {
    uint8_t dstp[129];
    std::memcpy(dstp, desc->h2c_dst, desc->h2c_dst_len);
    dstp[desc->h2c_dst_len] = (uint8_t)desc->h2c_dst_len;
    hs.dst_prime_len = desc->h2c_dst_len + 1;
    pack(hs.dst_prime_w, dstp, hs.dst_prime_len);
  }
  uint8_t gb[128];
  std::memcpy(gb, desc->generator, 64);
  std::memcpy(gb + 64, desc->blinding_base, 64);
  {
    bool have_b = false;
    for (size_t i = 0; i < 64; ++i) have_b = have_b || desc->blinding_base[i] != 0;
    ctx->have_blinding = have_b;
    if (!have_b) std::memcpy(gb + 64, desc->generator, 64);   // the table kernel wants a valid point; the scheme stays off
  }
  HIP_TRY_C(hipMalloc(&d_init, 512));
  HIP_TRY_C(hipMalloc(&d_prefix, prefix_bytes));
  auto free_tmp = [&]() { (void)hipFree(d_prefix); (void)hipFree(d_init); };
  ctx->T.sq.P = ctx->d_sqrt_p;
  ctx->T.sq.lut = ctx->d_sqrt_lut;
  ctx->T.sq.str = hs;
  {
    // on the context's stream: ordered in front of the table kernels by construction (gb outlives the synchronisation below)
    hipError_t e1 = hipMemcpyAsync(d_init, gb, sizeof gb, hipMemcpyHostToDevice, ctx->stream);
    if (e1 != hipSuccess) { free_tmp(); HIP_TRY_C(e1); }
  }
  uint32_t* d_mont = reinterpret_cast<uint32_t*>(d_init + 128);
  uint8_t* d_flags = d_init + 128 + 4 * NL * sizeof(uint32_t);
  if (bsw) {
    // the descriptor holds the short-Weierstrass coordinates (`Suite::generator()` of the SW suite): tables are built from
    // their te_sw_map images.  A point without one (infinity, y = 0) is no generator.
    uint8_t map_st[2] = {1, 1};
    vrf::f_bls381fr::launch_te_sw_map(SUITE_BS, 2, 1, 0, d_init, d_init + 288, d_init + 420, ctx->stream);
    hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ctx->stream);
    hipError_t e3 = hipMemcpy(map_st, d_init + 420, 2, hipMemcpyDeviceToHost);
    // on the context's stream, in front of the table kernels: a device-to-device hipMemcpy on the null stream need not have
    // finished when it returns, and the (non-blocking) stream would not wait for it
    hipError_t e4 = hipMemcpyAsync(d_init, d_init + 288, 128, hipMemcpyDeviceToDevice, ctx->stream);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
      free_tmp();
      HIP_TRY_C(e1); HIP_TRY_C(e2); HIP_TRY_C(e3); HIP_TRY_C(e4);
    }
    if (map_st[0] || map_st[1]) {
      free_tmp();
      return cleanup(fail(VRFHIP_ERR_BAD_ARG, map_st[0] ? "desc.generator has no twisted-Edwards image" : "desc.blinding_base has no twisted-Edwards image"));
    }
  }
  FIELD_CALL(ctx, launch_init_tables((int)suite, ctx->d_g_win, ctx->d_g_comb, ctx->d_b_comb, d_prefix, d_init, d_mont, d_flags, ctx->T.sq,
                     ctx->stream));
  uint8_t base_ok[2] = {0, 0};
  {
    hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(ctx->stream);
    hipError_t e3 = hipMemcpy(base_ok, d_flags, 2, hipMemcpyDeviceToHost);
    free_tmp();
    HIP_TRY_C(e1);
    HIP_TRY_C(e2);
    HIP_TRY_C(e3);
  }
#undef HIP_TRY_C
  if (!base_ok[0]) return cleanup(fail(VRFHIP_ERR_BAD_ARG, "desc.generator is not a non-identity point of the prime-order subgroup"));
  if (!base_ok[1]) return cleanup(fail(VRFHIP_ERR_BAD_ARG, "desc.blinding_base is not a non-identity point of the prime-order subgroup"));
  ctx->T.g_win = ctx->d_g_win;
  ctx->T.g_comb = ctx->d_g_comb;
  ctx->T.b_comb = ctx->d_b_comb;
  *out = ctx;
  return VRFHIP_SUCCESS;
}

void vrfhip_ctx_destroy(vrfhip_ctx* ctx) {
  if (!ctx) return;
  {
    DeviceGuard guard(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);              // its secret part (nonces, blindings) is wiped after every prove
    if (ctx->d_stage) {
      (void)hipMemset(ctx->d_stage, 0, ctx->stage_bytes);      // staged secret keys: `Secret` zeroizes on drop
      (void)hipFree(ctx->d_stage);
    }
    if (ctx->d_msm_ws) (void)hipFree(ctx->d_msm_ws);
    if (ctx->d_sqrt_p) (void)hipFree(ctx->d_sqrt_p);
    if (ctx->d_sqrt_lut) (void)hipFree(ctx->d_sqrt_lut);
    if (ctx->d_g_win) (void)hipFree(ctx->d_g_win);
    if (ctx->d_g_comb) (void)hipFree(ctx->d_g_comb);
    if (ctx->d_b_comb) (void)hipFree(ctx->d_b_comb);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_pair_prep) (void)hipFree(ctx->d_pair_prep);
    if (ctx->d_p256_comb) (void)hipFree(ctx->d_p256_comb);
    if (ctx->d_p256_comb_b) (void)hipFree(ctx->d_p256_comb_b);
    if (ctx->d_h2c_ctr) (void)hipFree(ctx->d_h2c_ctr);
    if (ctx->h_pin) {
      std::memset(ctx->h_pin, 0, 2 * ctx->pin_slot_bytes);     // may have staged caller data
      (void)hipHostFree(ctx->h_pin);
    }
    for (hipEvent_t e : ctx->ev_copied) if (e) (void)hipEventDestroy(e);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  }
  delete ctx;
}

int32_t vrfhip_ctx_reserve(vrfhip_ctx* ctx, size_t max_items) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (max_items == 0) return fail(VRFHIP_ERR_BAD_ARG, "max_items is 0");
  ctx->chunk_limit = max_items;
  if (max_items != ctx->ws_cap) return alloc_workspace(ctx, max_items);
  return VRFHIP_SUCCESS;
}

size_t vrfhip_ctx_workspace_bytes(const vrfhip_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

int32_t vrfhip_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(VRFHIP_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (bytes == 0) return fail(VRFHIP_ERR_BAD_ARG, "bytes is 0");
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable));
  return VRFHIP_SUCCESS;
}
void vrfhip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int32_t vrfhip_ctx_set_flags(vrfhip_ctx* ctx, uint32_t flags) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (flags & ~(uint32_t)(VRFHIP_FLAG_PREVALIDATED_ALL | VRFHIP_FLAG_PROVE_POINTS_AFFINE | VRFHIP_FLAG_COORDS_MONT256 | VRFHIP_FLAG_CT_TABLES))
    return fail(VRFHIP_ERR_BAD_ARG, "unknown flag bits");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  ctx->flags = flags;
  return VRFHIP_SUCCESS;
}
uint32_t vrfhip_ctx_get_flags(const vrfhip_ctx* ctx) { return ctx ? ctx->flags : 0; }
size_t vrfhip_ctx_point_bytes(const vrfhip_ctx* ctx) { return ctx ? ctx->pt_bytes() : 0; }
size_t vrfhip_ctx_hash_bytes(const vrfhip_ctx* ctx) { return ctx ? ctx->hash_bytes() : 0; }

int32_t vrfhip_debug_proofs_per_lane(size_t n) { return lanes_k(n, VERIFY_K_POLICY); }

int32_t vrfhip_debug_set(vrfhip_ctx* ctx, int32_t key, int32_t value) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  switch (key) {
    case VRFHIP_DEBUG_PAIRING_LAYOUT:
      if ((value & 0xff) > 6 || (value & ~0x1ff)) return fail(VRFHIP_ERR_BAD_ARG, "unknown pairing layout");
      ctx->dbg_pairing_layout = value;
      return VRFHIP_SUCCESS;
    case VRFHIP_DEBUG_PIPE_FIRST_LOG2:
    case VRFHIP_DEBUG_PIPE_CHUNK_LOG2:
      if (value < 12 || value > 18) return fail(VRFHIP_ERR_BAD_ARG, "pipeline chunk log2 must lie in 12..18");
      (key == VRFHIP_DEBUG_PIPE_FIRST_LOG2 ? ctx->pipe_first_log2 : ctx->pipe_chunk_log2) = value;
      return VRFHIP_SUCCESS;
    case VRFHIP_DEBUG_PROVE_K:
      if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(VRFHIP_ERR_BAD_ARG, "proofs per lane: 0 (default), 1, 2, 4 or 8");
      ctx->dbg_prove_k = value;
      return VRFHIP_SUCCESS;
    case VRFHIP_DEBUG_P256_MSM_GROUPS:
      if (value < 0 || value > 4096) return fail(VRFHIP_ERR_BAD_ARG, "groups out of range");
      ctx->dbg_p256_msm_groups = value;
      return VRFHIP_SUCCESS;
    default:
      return fail(VRFHIP_ERR_BAD_ARG, "unknown debug key");
  }
}

int32_t vrfhip_ctx_profile(vrfhip_ctx* ctx, int32_t enable) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  ctx->prof = enable != 0;
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_ctx_profile_read(vrfhip_ctx* ctx, double stage_ms[4], uint64_t* launches) {
  if (!ctx || !stage_ms || !launches) return fail(VRFHIP_ERR_BAD_ARG, "NULL argument");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  stage_ms[0] = stage_ms[1] = stage_ms[2] = stage_ms[3] = 0.0;
  *launches = ctx->prof_ev.size() / 5;
  for (size_t g = 0; g + 4 < ctx->prof_ev.size(); g += 5) {
    HIP_TRY(hipEventSynchronize(ctx->prof_ev[g + 4]));
    for (int k = 0; k < 4; ++k) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_ev[g + k], ctx->prof_ev[g + k + 1]));
      stage_ms[k] += ms;
    }
  }
  for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
  ctx->prof_ev.clear();
  return VRFHIP_SUCCESS;
}

// ------------------------------------------------------------------------- IETF verify
}  // extern "C"

namespace {
int32_t verify_dev_impl(vrfhip_ctx* ctx, size_t n, bool affine, const uint8_t* d_pk, const uint8_t* d_input,
                        const uint8_t* d_output, const uint8_t* d_c, const uint8_t* d_s, const uint8_t* d_ad,
                        const uint32_t* d_ad_off, uint32_t ad_len, uint8_t* d_status, void* stream,
                        const vrfhip_keyset* ks = nullptr, const uint32_t* d_key_index = nullptr) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ks && (ks->ctx != ctx || !d_key_index)) return fail(VRFHIP_ERR_BAD_ARG, "key set of another context, or NULL key index");
  if (n == 0) return VRFHIP_SUCCESS;
  if (ks) d_pk = ks->d_enc;
  if (!d_pk || !d_input || !d_output || !d_c || !d_s || !d_status)
    return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || d_ad_off) && !d_ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_workspace(ctx, n);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ctx->sw) {
    for (size_t base = 0; base < n; base += ctx->ws_cap) {
      const size_t m = std::min(ctx->ws_cap, n - base);
      p256::VerifyArgs a{};
      a.n = m;
      const size_t pw = affine ? 64 : 33;
      a.pk = d_pk + base * pw; a.h = d_input + base * pw; a.gamma = d_output + base * pw;
      a.affine_in = affine ? (ctx->coords_mont256() ? 2 : 1) : 0;
      a.c = d_c + base * 32; a.s = d_s + base * 32;
      a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
      a.status = d_status + base;
      a.ws = ctx->p256_ws;
      a.comb = ctx->d_p256_comb;
      a.str = ctx->T.sq.str;
      if (ks) {
        a.pk = nullptr;
        a.key_index = d_key_index + base; a.n_keys = ks->n_keys; a.key_enc = ks->d_enc; a.key_aff = ks->d_aff;
        a.key_valid = ks->d_valid; a.key_combs = ks->d_combs; a.key_rows = ks->rows;
      }
      p256::launch_verify(a, st, prof_events(ctx));
    }
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  const size_t pw = affine ? 64 : ctx->pt_bytes();
  for (size_t base = 0; base < n; base += ctx->ws_cap) {
    size_t m = std::min(ctx->ws_cap, n - base);
    VerifyArgs a;
    a.suite = (int)ctx->suite;
    a.k_lane = lanes_k(m, VERIFY_K_POLICY);
    a.n = m;
    a.pk = ks ? d_pk : d_pk + base * pw; a.h = d_input + base * pw; a.gamma = d_output + base * pw;
    a.affine_in = affine ? (ctx->coords_mont256() ? 2 : 1) : 0;
    a.h_in_tabs = 0;
    a.check_mask = ctx->check_mask();
    a.key_index = ks ? d_key_index + base : nullptr;
    a.key_combs = ks ? ks->d_combs : nullptr;
    a.key_valid = ks ? ks->d_valid : nullptr;
    a.n_keys = ks ? ks->n_keys : 0;
    a.c = d_c + base * 32; a.s = d_s + base * 32;
    a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
    a.status = d_status + base;
    a.ws = ctx->ws;
    a.T = ctx->T;
    if (ctx->bsw) launch_bsw_ietf_verify(a, st, prof_events(ctx));
    else FIELD_CALL(ctx, launch_ietf_verify(a, st, prof_events(ctx)));
  }
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t verify_host_impl(vrfhip_ctx* ctx, size_t n, bool affine, const uint8_t* pk, const uint8_t* input,
                         const uint8_t* output, const uint8_t* c, const uint8_t* s, const uint8_t* ad,
                         const uint32_t* ad_off, uint32_t ad_len, uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!pk || !input || !output || !c || !s || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || ad_off) && !ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  size_t adb = blob_bytes(n, ad_off, ad_len, true);
  const size_t pw = affine ? 64 : ctx->pt_bytes();
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  size_t need = 3 * Stage::pad(n * pw) + 2 * Stage::pad(n * 32) + Stage::pad(adb + 1) +
                Stage::pad((n + 1) * 4) + Stage::pad(n);
  int32_t rc = ensure_stage(ctx, need);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_pk = sg.take(n * pw);
  uint8_t* d_h = sg.take(n * pw);
  uint8_t* d_g = sg.take(n * pw);
  uint8_t* d_c = sg.take(n * 32);
  uint8_t* d_s = sg.take(n * 32);
  uint8_t* d_ad = sg.take(adb + 1);
  uint32_t* d_off = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  uint8_t* d_st = sg.take(n);
  if (adb) HIP_TRY(hipMemcpyAsync(d_ad, ad, adb, hipMemcpyHostToDevice, ctx->stream));
  if (ad_off) HIP_TRY(hipMemcpyAsync(d_off, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  const PipeArr arrs[5] = {{pk, d_pk, pw}, {input, d_h, pw}, {output, d_g, pw}, {c, d_c, 32}, {s, d_s, 32}};
  if (n <= PIPE_CHUNK) {                         // one chunk: nothing to overlap with
    for (const PipeArr& a : arrs) HIP_TRY(hipMemcpyAsync(a.d, a.h, n * a.w, hipMemcpyHostToDevice, ctx->stream));
    rc = verify_dev_impl(ctx, n, affine, d_pk, d_h, d_g, d_c, d_s, d_ad, ad_off ? d_off : nullptr, ad_len, d_st, ctx->stream);
    if (rc) return rc;
  } else {
    bool pinned[5];
    size_t slot_bytes = 0;
    for (int a = 0; a < 5; ++a) {
      pinned[a] = is_pinned_host(arrs[a].h);
      if (!pinned[a]) slot_bytes += Stage::pad(PIPE_CHUNK * arrs[a].w);
    }
    rc = pipe_prepare(ctx, std::max<size_t>(slot_bytes, 256));
    if (rc) return rc;
    int slot = 0;
    // the first chunk's copy is the only one nothing hides: it is a smaller chunk (tuning: vrfhip_debug_set keys 2, 3)
    const size_t first_chunk = size_t(1) << ctx->pipe_first_log2, chunk = size_t(1) << ctx->pipe_chunk_log2;
    for (size_t base = 0, m = 0; base < n; base += m, slot ^= 1) {
      m = std::min(base == 0 ? first_chunk : chunk, n - base);
      rc = pipe_send(ctx, arrs, pinned, 5, base, m, slot);
      if (rc) return rc;
      rc = verify_dev_impl(ctx, m, affine, d_pk + base * pw, d_h + base * pw, d_g + base * pw, d_c + base * 32, d_s + base * 32,
                           d_ad, ad_off ? d_off + base : nullptr, ad_len, d_st + base, ctx->stream);
      if (rc) return rc;
    }
  }
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}
}  // namespace

extern "C" {

int32_t vrfhip_ietf_verify_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_pk,
                                     const uint8_t* d_input, const uint8_t* d_output,
                                     const uint8_t* d_c, const uint8_t* d_s, const uint8_t* d_ad,
                                     const uint32_t* d_ad_off, uint32_t ad_len, uint8_t* d_status,
                                     void* stream) {
  return verify_dev_impl(ctx, n, false, d_pk, d_input, d_output, d_c, d_s, d_ad, d_ad_off, ad_len, d_status,
                         stream);
}
// `Input::new(alpha)` + `ietf::Verifier::verify` in one launch group: H never leaves the device and is never compressed,
// decompressed or subgroup-tested (k_verify.hip k_verify_input_from_alpha).  Per chunk of the workspace: hash-to-curve writes
// enc(H) into the (otherwise idle) aux region and H's tables into slot 1, the decode stage handles pk and Gamma.
int32_t vrfhip_ietf_verify_batch_alpha_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_pk, const uint8_t* d_msg,
                                           const uint32_t* d_msg_off, uint32_t msg_len, const uint8_t* d_output,
                                           const uint8_t* d_c, const uint8_t* d_s, const uint8_t* d_ad,
                                           const uint32_t* d_ad_off, uint32_t ad_len, uint8_t* d_status, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_pk || !d_output || !d_c || !d_s || !d_status || (!d_msg && (msg_len || d_msg_off)))
    return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || d_ad_off) && !d_ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_workspace(ctx, n);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ctx->sw) {
    // secp256r1: the two stages as they are (cofactor 1: there is no subgroup test to skip), H's Sec1 string parked in the
    // projective-results region, which nothing writes before the decode stage has read it
    for (size_t base = 0; base < n; base += ctx->ws_cap) {
      const size_t m = std::min(ctx->ws_cap, n - base);
      uint8_t* d_henc = reinterpret_cast<uint8_t*>(ctx->p256_ws.pts);
      BytesView mv = d_msg_off ? make_view(d_msg, d_msg_off + base, msg_len, false)
                               : make_view(d_msg ? d_msg + base * (size_t)msg_len : d_msg, nullptr, msg_len, false);
      p256::launch_hash_to_curve(m, mv, d_henc, ctx->T.sq.str, st, ctx->p256_ws.flags, ctx->d_queue);
      p256::VerifyArgs a{};
      a.n = m;
      a.pk = d_pk + base * 33; a.h = d_henc; a.gamma = d_output + base * 33;
      a.affine_in = 0;
      a.c = d_c + base * 32; a.s = d_s + base * 32;
      a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
      a.status = d_status + base;
      a.ws = ctx->p256_ws;
      a.comb = ctx->d_p256_comb;
      a.str = ctx->T.sq.str;
      p256::launch_verify(a, st, prof_events(ctx));
    }
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  if (ctx->bsw) {
    // the two stages as they are: H's 33-byte strings parked in the (otherwise idle) MSM workspace -- every region of the
    // verify workspace is written by the decode stage; H is a cofactor multiple by construction, so its subgroup test is skipped
    {
      const int32_t rc2 = ensure_msm_workspace(ctx, std::min(ctx->ws_cap, n) * 33 + 256);
      if (rc2) return rc2;
    }
    for (size_t base = 0; base < n; base += ctx->ws_cap) {
      const size_t m = std::min(ctx->ws_cap, n - base);
      uint8_t* d_henc = static_cast<uint8_t*>(ctx->d_msm_ws);
      BytesView mv = d_msg_off ? make_view(d_msg, d_msg_off + base, msg_len, false)
                               : make_view(d_msg ? d_msg + base * (size_t)msg_len : d_msg, nullptr, msg_len, false);
      launch_bsw_hash_to_curve(m, mv, d_henc, ctx->T, st, ctx->ws.flags, ctx->d_queue);
      VerifyArgs a{};
      a.suite = (int)ctx->suite;
      a.k_lane = 1;
      a.n = m;
      a.pk = d_pk + base * 33; a.h = d_henc; a.gamma = d_output + base * 33;
      a.check_mask = ctx->check_mask() & ~2u;
      a.c = d_c + base * 32; a.s = d_s + base * 32;
      a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
      a.status = d_status + base;
      a.ws = ctx->ws;
      a.T = ctx->T;
      launch_bsw_ietf_verify(a, st, prof_events(ctx));
    }
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  for (size_t base = 0; base < n; base += ctx->ws_cap) {
    const size_t m = std::min(ctx->ws_cap, n - base);
    uint8_t* d_henc = reinterpret_cast<uint8_t*>(ctx->ws.aux);              // [m][32]: AUX_WORDS * 4 >= 32 bytes per item
    BytesView mv = d_msg_off ? make_view(d_msg, d_msg_off + base, msg_len, false)
                             : make_view(d_msg ? d_msg + base * (size_t)msg_len : d_msg, nullptr, msg_len, false);
    FIELD_CALL(ctx, launch_verify_input_from_alpha((int)ctx->suite, m, mv, d_henc, ctx->ws.tabs, ctx->T, st, ctx->ws.flags,
                                                   ctx->d_queue));
    VerifyArgs a;
    a.suite = (int)ctx->suite;
    a.k_lane = lanes_k(m, VERIFY_K_POLICY);
    a.n = m;
    a.pk = d_pk + base * 32; a.h = d_henc; a.gamma = d_output + base * 32;
    a.affine_in = 0;
    a.h_in_tabs = 1;
    a.check_mask = ctx->check_mask() & ~2u;             // CHK_INPUT: H is a cofactor multiple by construction
    a.key_index = nullptr; a.key_combs = nullptr; a.key_valid = nullptr; a.n_keys = 0;
    a.c = d_c + base * 32; a.s = d_s + base * 32;
    a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
    a.status = d_status + base;
    a.ws = ctx->ws;
    a.T = ctx->T;
    FIELD_CALL(ctx, launch_ietf_verify(a, st, prof_events(ctx)));
  }
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_ietf_verify_batch_alpha(vrfhip_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* msg,
                                       const uint32_t* msg_off, uint32_t msg_len, const uint8_t* output, const uint8_t* c,
                                       const uint8_t* s, const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                       uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!pk || !output || !c || !s || !status || (!msg && (msg_len || msg_off))) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || ad_off) && !ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  const size_t msgb = blob_bytes(n, msg_off, msg_len, false), adb = blob_bytes(n, ad_off, ad_len, true);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  const size_t pw = ctx->pt_bytes();
  int32_t rc = ensure_stage(ctx, 2 * Stage::pad(n * pw) + 2 * Stage::pad(n * 32) + Stage::pad(msgb + 1) + Stage::pad(adb + 1) +
                                     2 * Stage::pad((n + 1) * 4) + Stage::pad(n));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t *d_pk = sg.take(n * pw), *d_g = sg.take(n * pw), *d_c = sg.take(n * 32), *d_s = sg.take(n * 32);
  uint8_t *d_msg = sg.take(msgb + 1), *d_ad = sg.take(adb + 1);
  uint32_t* d_moff = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  uint32_t* d_aoff = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  uint8_t* d_st = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_pk, pk, n * pw, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_g, output, n * pw, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_c, c, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_s, s, n * 32, hipMemcpyHostToDevice, ctx->stream));
  if (msgb) HIP_TRY(hipMemcpyAsync(d_msg, msg, msgb, hipMemcpyHostToDevice, ctx->stream));
  if (adb) HIP_TRY(hipMemcpyAsync(d_ad, ad, adb, hipMemcpyHostToDevice, ctx->stream));
  if (msg_off) HIP_TRY(hipMemcpyAsync(d_moff, msg_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (ad_off) HIP_TRY(hipMemcpyAsync(d_aoff, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = vrfhip_ietf_verify_batch_alpha_dev(ctx, n, d_pk, d_msg, msg_off ? d_moff : nullptr, msg_len, d_g, d_c, d_s, d_ad,
                                          ad_off ? d_aoff : nullptr, ad_len, d_st, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_ietf_verify_batch_affine_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_pk_xy,
                                            const uint8_t* d_input_xy, const uint8_t* d_output_xy,
                                            const uint8_t* d_c, const uint8_t* d_s, const uint8_t* d_ad,
                                            const uint32_t* d_ad_off, uint32_t ad_len,
                                            uint8_t* d_status, void* stream) {
  return verify_dev_impl(ctx, n, true, d_pk_xy, d_input_xy, d_output_xy, d_c, d_s, d_ad, d_ad_off, ad_len,
                         d_status, stream);
}
int32_t vrfhip_ietf_verify_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* input,
                                 const uint8_t* output, const uint8_t* c, const uint8_t* s,
                                 const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                 uint8_t* status) {
  return verify_host_impl(ctx, n, false, pk, input, output, c, s, ad, ad_off, ad_len, status);
}
int32_t vrfhip_ietf_verify_batch_affine(vrfhip_ctx* ctx, size_t n, const uint8_t* pk_xy,
                                        const uint8_t* input_xy, const uint8_t* output_xy,
                                        const uint8_t* c, const uint8_t* s, const uint8_t* ad,
                                        const uint32_t* ad_off, uint32_t ad_len, uint8_t* status) {
  return verify_host_impl(ctx, n, true, pk_xy, input_xy, output_xy, c, s, ad, ad_off, ad_len, status);
}

// ------------------------------------------------------------------------- key sets, keyed verification
int32_t vrfhip_keyset_create(vrfhip_ctx* ctx, size_t n_keys, const uint8_t* pks, uint8_t* status,
                             vrfhip_keyset** out) {
  if (!out) return fail(VRFHIP_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n_keys == 0 || !pks) return fail(VRFHIP_ERR_BAD_ARG, "no keys");
  if (n_keys > (size_t(1) << 24)) return fail(VRFHIP_ERR_BAD_ARG, "too many keys");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  vrfhip_keyset* ks = new vrfhip_keyset();
  ks->ctx = ctx;
  ks->n_keys = n_keys;
  if (ctx->sw) {
    // secp256r1: 33-byte Sec1 keys; per key the comb rows a challenge can reach (challenge_len + 1 rows of 128 entries)
    ks->rows = (int)ctx->desc.challenge_len + 1;
    const size_t cb = n_keys * p256::key_comb_bytes(ks->rows);
    auto bail_sw = [&](hipError_t e, const char* what) {
      vrfhip_keyset_destroy(ks);
      return fail(e == hipErrorOutOfMemory ? VRFHIP_ERR_OOM : VRFHIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc(&ks->d_enc, n_keys * 33 + 3)) != hipSuccess) return bail_sw(e, "hipMalloc(keys)");
    if ((e = hipMalloc(&ks->d_valid, (n_keys + 255) & ~size_t(255))) != hipSuccess) return bail_sw(e, "hipMalloc(valid)");
    if ((e = hipMalloc(&ks->d_aff, n_keys * 18 * sizeof(uint32_t))) != hipSuccess) return bail_sw(e, "hipMalloc(affine)");
    if ((e = hipMalloc(&ks->d_combs, cb)) != hipSuccess) return bail_sw(e, "hipMalloc(combs)");
    ks->bytes = n_keys * (33 + 1 + 72) + cb;
    if ((e = hipMemcpyAsync(ks->d_enc, pks, n_keys * 33, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return bail_sw(e, "copy keys");
    p256::launch_keyset_build(n_keys, ks->d_enc, ks->d_aff, ks->d_valid, ks->d_combs, ks->rows, ctx->stream);
    if ((e = hipGetLastError()) != hipSuccess) return bail_sw(e, "launch");
    std::vector<uint8_t> valid(n_keys);
    if ((e = hipMemcpyAsync(valid.data(), ks->d_valid, n_keys, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return bail_sw(e, "copy back");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail_sw(e, "synchronize");
    if (status)
      for (size_t i = 0; i < n_keys; ++i) status[i] = valid[i] ? VRFHIP_ST_OK : VRFHIP_ST_INVALID_DATA;
    *out = ks;
    return VRFHIP_SUCCESS;
  }
  const size_t comb_bytes = n_keys * COMB_WORDS * sizeof(uint32_t);
  const size_t prefix_bytes = n_keys * COMB_ROWS * (size_t)COMB_COLS * NL * sizeof(uint32_t);
  uint32_t *d_xy = nullptr, *d_prefix = nullptr;
  auto bail = [&](int32_t rc) {
    if (d_xy) (void)hipFree(d_xy);
    if (d_prefix) (void)hipFree(d_prefix);
    vrfhip_keyset_destroy(ks);
    return rc;
  };
#define HIP_TRY_K(expr)                                                                            \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return bail(fail(e__ == hipErrorOutOfMemory ? VRFHIP_ERR_OOM : VRFHIP_ERR_HIP,               \
                       std::string(#expr) + ": " + hipGetErrorString(e__)));                       \
  } while (0)
  const size_t kw = ctx->pt_bytes();                 // 32; 33 for bandersnatch_sw
  HIP_TRY_K(hipMalloc(&ks->d_enc, n_keys * kw + 3));
  HIP_TRY_K(hipMalloc(&ks->d_valid, (n_keys + 255) & ~size_t(255)));
  HIP_TRY_K(hipMalloc(&ks->d_combs, comb_bytes));
  HIP_TRY_K(hipMalloc(&d_xy, n_keys * 2 * NL * sizeof(uint32_t)));
  HIP_TRY_K(hipMalloc(&d_prefix, prefix_bytes));
  ks->bytes = n_keys * kw + n_keys + comb_bytes;
  HIP_TRY_K(hipMemcpyAsync(ks->d_enc, pks, n_keys * kw, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->bsw) launch_bsw_keyset_build(n_keys, ks->d_enc, d_xy, ks->d_valid, ks->d_combs, d_prefix, ctx->T, ctx->stream);
  else FIELD_CALL(ctx, launch_keyset_build((int)ctx->suite, n_keys, ks->d_enc, d_xy, ks->d_valid, ks->d_combs, d_prefix, ctx->T, ctx->stream));
  HIP_TRY_K(hipGetLastError());
  if (status) {
    std::vector<uint8_t> valid(n_keys);
    HIP_TRY_K(hipMemcpyAsync(valid.data(), ks->d_valid, n_keys, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY_K(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n_keys; ++i) status[i] = valid[i] ? VRFHIP_ST_OK : VRFHIP_ST_INVALID_DATA;
  } else {
    HIP_TRY_K(hipStreamSynchronize(ctx->stream));
  }
#undef HIP_TRY_K
  (void)hipFree(d_xy);
  (void)hipFree(d_prefix);
  *out = ks;
  return VRFHIP_SUCCESS;
}

void vrfhip_keyset_destroy(vrfhip_keyset* ks) {
  if (!ks) return;
  if (ks->ctx) {
    DeviceGuard guard(ks->ctx->device);
    if (ks->d_enc) (void)hipFree(ks->d_enc);
    if (ks->d_valid) (void)hipFree(ks->d_valid);
    if (ks->d_combs) (void)hipFree(ks->d_combs);
    if (ks->d_aff) (void)hipFree(ks->d_aff);
  }
  delete ks;
}

size_t vrfhip_keyset_bytes(const vrfhip_keyset* ks) { return ks ? ks->bytes : 0; }

int32_t vrfhip_ietf_verify_batch_keyed_dev(vrfhip_ctx* ctx, const vrfhip_keyset* keys, size_t n,
                                           const uint32_t* d_key_index, const uint8_t* d_input,
                                           const uint8_t* d_output, const uint8_t* d_c, const uint8_t* d_s,
                                           const uint8_t* d_ad, const uint32_t* d_ad_off, uint32_t ad_len,
                                           uint8_t* d_status, void* stream) {
  if (!keys) return fail(VRFHIP_ERR_BAD_ARG, "key set is NULL");
  return verify_dev_impl(ctx, n, false, nullptr, d_input, d_output, d_c, d_s, d_ad, d_ad_off, ad_len, d_status, stream,
                         keys, d_key_index);
}

int32_t vrfhip_ietf_verify_batch_keyed(vrfhip_ctx* ctx, const vrfhip_keyset* keys, size_t n,
                                       const uint32_t* key_index, const uint8_t* input, const uint8_t* output,
                                       const uint8_t* c, const uint8_t* s, const uint8_t* ad,
                                       const uint32_t* ad_off, uint32_t ad_len, uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (!keys) return fail(VRFHIP_ERR_BAD_ARG, "key set is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!key_index || !input || !output || !c || !s || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || ad_off) && !ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  size_t adb = blob_bytes(n, ad_off, ad_len, true);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  const size_t pw = ctx->pt_bytes();           // input / output: 32-byte points, 33-byte Sec1 strings on secp256r1
  size_t need = 2 * Stage::pad(n * pw) + 2 * Stage::pad(n * 32) + Stage::pad(n * 4) + Stage::pad(adb + 1) + Stage::pad((n + 1) * 4) + Stage::pad(n);
  int32_t rc = ensure_stage(ctx, need);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  const uint8_t* src[4] = {input, output, c, s};
  uint8_t* d[4];
  for (int i = 0; i < 4; ++i) {
    const size_t w = i < 2 ? pw : 32;
    d[i] = sg.take(n * w);
    HIP_TRY(hipMemcpyAsync(d[i], src[i], n * w, hipMemcpyHostToDevice, ctx->stream));
  }
  uint32_t* d_idx = reinterpret_cast<uint32_t*>(sg.take(n * 4));
  uint8_t* d_ad = sg.take(adb + 1);
  uint32_t* d_off = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  uint8_t* d_st = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_idx, key_index, n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (adb) HIP_TRY(hipMemcpyAsync(d_ad, ad, adb, hipMemcpyHostToDevice, ctx->stream));
  if (ad_off) HIP_TRY(hipMemcpyAsync(d_off, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = vrfhip_ietf_verify_batch_keyed_dev(ctx, keys, n, d_idx, d[0], d[1], d[2], d[3], d_ad, ad_off ? d_off : nullptr,
                                          ad_len, d_st, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

// ------------------------------------------------------------------------- IETF prove
}  // extern "C"

namespace {
struct ProveOut {
  uint8_t *output, *c, *s, *pk, *input, *status;
  uint8_t *r, *ok, *sb, *blinding;     // Pedersen only
};
int32_t prove_dev_impl(vrfhip_ctx* ctx, size_t n, bool pedersen, const uint8_t* d_sk,
                       const uint8_t* d_msg, const uint32_t* d_msg_off, uint32_t msg_len,
                       const uint8_t* d_input, const uint8_t* d_ad, const uint32_t* d_ad_off,
                       uint32_t ad_len, const ProveOut& o, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_sk || !o.output || !o.s) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if (pedersen ? (!o.pk || !o.r || !o.ok || !o.sb) : !o.c) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if (!d_input && !d_msg && (msg_len || d_msg_off)) return fail(VRFHIP_ERR_BAD_ARG, "msg is NULL");
  if ((ad_len || d_ad_off) && !d_ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  if (pedersen && !ctx->has_pedersen()) return fail(VRFHIP_ERR_UNSUPPORTED, NO_BLINDING_MSG);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_workspace(ctx, n);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto at = [](uint8_t* p, size_t base, size_t w) -> uint8_t* { return p ? p + base * w : nullptr; };
  if (ctx->sw) {
    for (size_t base = 0; base < n; base += ctx->ws_cap) {
      const size_t m = std::min(ctx->ws_cap, n - base);
      p256::ProveArgs a;
      a.n = m;
      a.sk = d_sk + base * 32;
      if (d_msg_off) a.msg = make_view(d_msg, d_msg_off + base, 0, false);
      else a.msg = make_view(d_msg ? d_msg + base * (size_t)msg_len : nullptr, nullptr, msg_len, false);
      a.h_given = d_input ? d_input + base * 33 : nullptr;
      a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
      const size_t ptw = ctx->prove_point_bytes();       // 33: Sec1, 64: x || y (VRFHIP_FLAG_PROVE_POINTS_AFFINE)
      a.out_affine = ptw == 64 ? (ctx->coords_mont256() ? 2 : 1) : 0;
      a.gamma = at(o.output, base, ptw); a.c = at(o.c, base, 32); a.s = at(o.s, base, 32);
      a.pk_out = at(o.pk, base, ptw); a.h_out = at(o.input, base, 33); a.status = at(o.status, base, 1);
      a.pedersen = pedersen ? 1 : 0;
      a.ct_tables = (ctx->flags & VRFHIP_FLAG_CT_TABLES) ? 1 : 0;
      a.r_out = at(o.r, base, ptw); a.ok_out = at(o.ok, base, ptw); a.sb_out = at(o.sb, base, 32);
      a.blinding_out = at(o.blinding, base, 32);
      a.comb_b = ctx->d_p256_comb_b;
      a.tai_queue = ctx->d_queue;
      a.ws = ctx->p256_ws;
      a.comb = ctx->d_p256_comb;
      a.str = ctx->T.sq.str;
      p256::launch_prove(a, st, prof_events(ctx));
      // sk and the nonce k of these items lay in the scalar region: nothing secret outlives the call in device memory
      HIP_TRY(hipMemsetAsync(ctx->p256_ws.sc, 0, ctx->p256_ws.cap * p256::WS_SC_WORDS * sizeof(uint32_t), st));
    }
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  const size_t ipw = ctx->pt_bytes();                    // input points and enc(H) out: always compressed
  for (size_t base = 0; base < n; base += ctx->ws_cap) {
    size_t m = std::min(ctx->ws_cap, n - base);
    ProveArgs a;
    a.suite = (int)ctx->suite;
    a.k_lane = ctx->dbg_prove_k > 0 ? ctx->dbg_prove_k : lanes_k(m, PROVE_K);
    a.n = m;
    a.sk = d_sk + base * 32;
    if (d_msg_off) a.msg = make_view(d_msg, d_msg_off + base, 0, false);
    else a.msg = make_view(d_msg ? d_msg + base * (size_t)msg_len : nullptr, nullptr, msg_len, false);
    a.h_given = d_input ? d_input + base * ipw : nullptr;
    a.tai_queue = ctx->d_queue;
    a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
    const size_t ptw = ctx->prove_point_bytes();        // 32: compressed, 64: x || y (VRFHIP_FLAG_PROVE_POINTS_AFFINE)
    a.out_affine = ptw == 64 ? 1 : 0;
    a.gamma = at(o.output, base, ptw); a.c = at(o.c, base, 32); a.s = at(o.s, base, 32);
    a.pk_out = at(o.pk, base, ptw);
    a.h_out = at(o.input, base, ipw);
    a.status = at(o.status, base, 1);
    a.pedersen = pedersen ? 1 : 0;
    a.check_mask = ctx->check_mask() | ((ctx->flags & VRFHIP_FLAG_CT_TABLES) ? (uint32_t)CHK_CT_TABLES : 0u);
    a.r_out = at(o.r, base, ptw); a.ok_out = at(o.ok, base, ptw); a.sb_out = at(o.sb, base, 32);
    a.blinding_out = at(o.blinding, base, 32);
    a.ws = ctx->ws;
    a.T = ctx->T;
    if (ctx->bsw) launch_bsw_prove(a, st, prof_events(ctx));
    else FIELD_CALL(ctx, launch_ietf_prove(a, st, prof_events(ctx)));
    if (a.out_affine && ctx->coords_mont256()) {       // x || y outputs in arkworks' in-memory form
      FIELD_CALL(ctx, launch_xy_to_mont256(m, a.gamma, st));
      FIELD_CALL(ctx, launch_xy_to_mont256(m, a.pk_out, st));
      if (pedersen) { FIELD_CALL(ctx, launch_xy_to_mont256(m, a.r_out, st)); FIELD_CALL(ctx, launch_xy_to_mont256(m, a.ok_out, st)); }
    }
    // the aux region held the nonces k, kb and the blinding factor b of these items: wipe it (the reference's
    // `Secret` zeroizes on drop; nothing secret may outlive the call in device memory)
    HIP_TRY(hipMemsetAsync(ctx->ws.aux, 0, m * AUX_WORDS * sizeof(uint32_t), st));
  }
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

// host-pointer form shared by the IETF and Pedersen provers
int32_t prove_host_impl(vrfhip_ctx* ctx, size_t n, bool pedersen, const uint8_t* sk, const uint8_t* msg,
                        const uint32_t* msg_off, uint32_t msg_len, const uint8_t* input,
                        const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len, const ProveOut& h) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!sk || !h.output || !h.s) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if (pedersen ? (!h.pk || !h.r || !h.ok || !h.sb) : !h.c) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if (!input && !msg && (msg_len || msg_off)) return fail(VRFHIP_ERR_BAD_ARG, "msg is NULL");
  if ((ad_len || ad_off) && !ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  size_t adb = blob_bytes(n, ad_off, ad_len, true);
  size_t msgb = input ? 0 : blob_bytes(n, msg_off, msg_len, false);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  const size_t ptw = ctx->prove_point_bytes(), ipw = ctx->pt_bytes();     // outputs (maybe x || y) and input points
  size_t need = 5 * Stage::pad(n * 32) + 2 * Stage::pad(n * ipw) + 4 * Stage::pad(n * ptw) + Stage::pad(msgb + 1) + Stage::pad(adb + 1) +
                2 * Stage::pad((n + 1) * 4) + Stage::pad(n);
  int32_t rc = ensure_stage(ctx, need);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_sk = sg.take(n * 32);
  uint8_t* d_in = sg.take(n * ipw);
  ProveOut d{};
  d.output = sg.take(n * ptw); d.c = sg.take(n * 32); d.s = sg.take(n * 32);
  d.pk = sg.take(n * ptw); d.input = sg.take(n * ipw);
  d.r = sg.take(n * ptw); d.ok = sg.take(n * ptw); d.sb = sg.take(n * 32); d.blinding = sg.take(n * 32);
  uint8_t* d_msg = sg.take(msgb + 1);
  uint8_t* d_ad = sg.take(adb + 1);
  uint32_t* d_moff = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  uint32_t* d_aoff = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  d.status = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_sk, sk, n * 32, hipMemcpyHostToDevice, ctx->stream));
  if (input) HIP_TRY(hipMemcpyAsync(d_in, input, n * ipw, hipMemcpyHostToDevice, ctx->stream));
  if (msgb) HIP_TRY(hipMemcpyAsync(d_msg, msg, msgb, hipMemcpyHostToDevice, ctx->stream));
  if (adb) HIP_TRY(hipMemcpyAsync(d_ad, ad, adb, hipMemcpyHostToDevice, ctx->stream));
  if (msg_off && !input)
    HIP_TRY(hipMemcpyAsync(d_moff, msg_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (ad_off) HIP_TRY(hipMemcpyAsync(d_aoff, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = prove_dev_impl(ctx, n, pedersen, d_sk, d_msg, (msg_off && !input) ? d_moff : nullptr, msg_len,
                      input ? d_in : nullptr, d_ad, ad_off ? d_aoff : nullptr, ad_len, d, ctx->stream);
  if (rc) return rc;
  auto back = [&](uint8_t* dst, const uint8_t* src, size_t bytes) -> hipError_t {
    return dst ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
  };
  HIP_TRY(back(h.output, d.output, n * ptw));
  HIP_TRY(back(h.s, d.s, n * 32));
  HIP_TRY(back(h.pk, d.pk, n * ptw));
  HIP_TRY(back(h.input, d.input, n * ipw));
  HIP_TRY(back(h.status, d.status, n));
  if (pedersen) {
    HIP_TRY(back(h.r, d.r, n * ptw));
    HIP_TRY(back(h.ok, d.ok, n * ptw));
    HIP_TRY(back(h.sb, d.sb, n * 32));
    HIP_TRY(back(h.blinding, d.blinding, n * 32));
  } else {
    HIP_TRY(back(h.c, d.c, n * 32));
  }
  HIP_TRY(hipMemsetAsync(d_sk, 0, n * 32, ctx->stream));               // staged secret keys
  if (pedersen) HIP_TRY(hipMemsetAsync(d.blinding, 0, n * 32, ctx->stream));   // and blinding factors
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}
}  // namespace

extern "C" {

int32_t vrfhip_ietf_prove_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_sk,
                                    const uint8_t* d_msg, const uint32_t* d_msg_off,
                                    uint32_t msg_len, const uint8_t* d_input, const uint8_t* d_ad,
                                    const uint32_t* d_ad_off, uint32_t ad_len, uint8_t* d_output,
                                    uint8_t* d_c, uint8_t* d_s, uint8_t* d_pk_out,
                                    uint8_t* d_input_out, uint8_t* d_status, void* stream) {
  ProveOut o{};
  o.output = d_output; o.c = d_c; o.s = d_s; o.pk = d_pk_out; o.input = d_input_out; o.status = d_status;
  return prove_dev_impl(ctx, n, false, d_sk, d_msg, d_msg_off, msg_len, d_input, d_ad, d_ad_off, ad_len, o,
                        stream);
}

int32_t vrfhip_pedersen_prove_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_sk,
                                        const uint8_t* d_msg, const uint32_t* d_msg_off,
                                        uint32_t msg_len, const uint8_t* d_input,
                                        const uint8_t* d_ad, const uint32_t* d_ad_off,
                                        uint32_t ad_len, uint8_t* d_output, uint8_t* d_pk_com,
                                        uint8_t* d_r, uint8_t* d_ok, uint8_t* d_s, uint8_t* d_sb,
                                        uint8_t* d_blinding_out, uint8_t* d_input_out,
                                        uint8_t* d_status, void* stream) {
  ProveOut o{};
  o.output = d_output; o.s = d_s; o.pk = d_pk_com; o.input = d_input_out; o.status = d_status;
  o.r = d_r; o.ok = d_ok; o.sb = d_sb; o.blinding = d_blinding_out;
  return prove_dev_impl(ctx, n, true, d_sk, d_msg, d_msg_off, msg_len, d_input, d_ad, d_ad_off, ad_len, o,
                        stream);
}

int32_t vrfhip_ietf_prove_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* sk, const uint8_t* msg,
                                const uint32_t* msg_off, uint32_t msg_len, const uint8_t* input,
                                const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                uint8_t* output, uint8_t* c, uint8_t* s, uint8_t* pk_out,
                                uint8_t* input_out, uint8_t* status) {
  ProveOut h{};
  h.output = output; h.c = c; h.s = s; h.pk = pk_out; h.input = input_out; h.status = status;
  return prove_host_impl(ctx, n, false, sk, msg, msg_off, msg_len, input, ad, ad_off, ad_len, h);
}

int32_t vrfhip_pedersen_prove_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* sk, const uint8_t* msg,
                                    const uint32_t* msg_off, uint32_t msg_len, const uint8_t* input,
                                    const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                    uint8_t* output, uint8_t* pk_com, uint8_t* r, uint8_t* ok,
                                    uint8_t* s, uint8_t* sb, uint8_t* blinding_out,
                                    uint8_t* input_out, uint8_t* status) {
  ProveOut h{};
  h.output = output; h.s = s; h.pk = pk_com; h.input = input_out; h.status = status;
  h.r = r; h.ok = ok; h.sb = sb; h.blinding = blinding_out;
  return prove_host_impl(ctx, n, true, sk, msg, msg_off, msg_len, input, ad, ad_off, ad_len, h);
}

// ------------------------------------------------------------------------- Pedersen verify
int32_t vrfhip_pedersen_verify_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_input,
                                         const uint8_t* d_output, const uint8_t* d_pk_com,
                                         const uint8_t* d_r, const uint8_t* d_ok, const uint8_t* d_s,
                                         const uint8_t* d_sb, const uint8_t* d_ad,
                                         const uint32_t* d_ad_off, uint32_t ad_len,
                                         uint8_t* d_status, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_input || !d_output || !d_pk_com || !d_r || !d_ok || !d_s || !d_sb || !d_status)
    return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || d_ad_off) && !d_ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  if (!ctx->has_pedersen()) return fail(VRFHIP_ERR_UNSUPPORTED, NO_BLINDING_MSG);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_workspace(ctx, n);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ctx->sw) {
    for (size_t base = 0; base < n; base += ctx->ws_cap) {
      const size_t m = std::min(ctx->ws_cap, n - base);
      p256::PedVerifyArgs a;
      a.n = m;
      a.h = d_input + base * 33; a.gamma = d_output + base * 33; a.pk_com = d_pk_com + base * 33;
      a.r = d_r + base * 33; a.ok = d_ok + base * 33; a.s = d_s + base * 32; a.sb = d_sb + base * 32;
      a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
      a.status = d_status + base;
      a.ws = ctx->p256_ws;
      a.comb = ctx->d_p256_comb; a.comb_b = ctx->d_p256_comb_b;
      a.str = ctx->T.sq.str;
      p256::launch_pedersen_verify(a, st, prof_events(ctx));
    }
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  for (size_t base = 0; base < n; base += ctx->ws_cap) {
    size_t m = std::min(ctx->ws_cap, n - base);
    PedersenVerifyArgs a;
    a.suite = (int)ctx->suite;
    a.n = m;
    const size_t pw = ctx->pt_bytes();
    a.h = d_input + base * pw; a.gamma = d_output + base * pw; a.pk_com = d_pk_com + base * pw;
    a.r = d_r + base * pw; a.ok = d_ok + base * pw; a.s = d_s + base * 32; a.sb = d_sb + base * 32;
    a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
    a.check_mask = ctx->check_mask();
    a.status = d_status + base;
    a.ws = ctx->ws;
    a.T = ctx->T;
    if (ctx->bsw) launch_bsw_pedersen_verify(a, st, prof_events(ctx));
    else FIELD_CALL(ctx, launch_pedersen_verify(a, st, prof_events(ctx)));
  }
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_pedersen_verify_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* input,
                                     const uint8_t* output, const uint8_t* pk_com, const uint8_t* r,
                                     const uint8_t* ok, const uint8_t* s, const uint8_t* sb,
                                     const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                     uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!input || !output || !pk_com || !r || !ok || !s || !sb || !status)
    return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || ad_off) && !ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  size_t adb = blob_bytes(n, ad_off, ad_len, true);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  const size_t pw = ctx->pt_bytes();
  size_t need = 5 * Stage::pad(n * pw) + 2 * Stage::pad(n * 32) + Stage::pad(adb + 1) + Stage::pad((n + 1) * 4) + Stage::pad(n);
  int32_t rc = ensure_stage(ctx, need);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  const uint8_t* src[7] = {input, output, pk_com, r, ok, s, sb};
  uint8_t* d[7];
  for (int i = 0; i < 7; ++i) {
    const size_t w = i < 5 ? pw : 32;
    d[i] = sg.take(n * w);
    HIP_TRY(hipMemcpyAsync(d[i], src[i], n * w, hipMemcpyHostToDevice, ctx->stream));
  }
  uint8_t* d_ad = sg.take(adb + 1);
  uint32_t* d_off = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  uint8_t* d_st = sg.take(n);
  if (adb) HIP_TRY(hipMemcpyAsync(d_ad, ad, adb, hipMemcpyHostToDevice, ctx->stream));
  if (ad_off) HIP_TRY(hipMemcpyAsync(d_off, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = vrfhip_pedersen_verify_batch_dev(ctx, n, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d_ad,
                                        ad_off ? d_off : nullptr, ad_len, d_st, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

// ------------------------------------------------------------------------- Pedersen verify, batched (RLC)
}  // extern "C"
namespace {
int32_t rlc_dev_impl(vrfhip_ctx* ctx, size_t n, bool affine, const uint8_t* d_input,
                     const uint8_t* d_output, const uint8_t* d_pk_com,
                     const uint8_t* d_r, const uint8_t* d_ok, const uint8_t* d_s,
                     const uint8_t* d_sb, const uint8_t* d_ad,
                     const uint32_t* d_ad_off, uint32_t ad_len,
                     const uint8_t seed[32], uint8_t* d_status,
                     uint8_t* d_fail_flag, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (!d_fail_flag || !seed) return fail(VRFHIP_ERR_BAD_ARG, "NULL fail flag or seed");
  if (n && (!d_input || !d_output || !d_pk_com || !d_r || !d_ok || !d_s || !d_sb || !d_status))
    return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || d_ad_off) && !d_ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  if (!ctx->has_pedersen()) return fail(VRFHIP_ERR_UNSUPPORTED, NO_BLINDING_MSG);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemsetAsync(d_fail_flag, 0, 1, st));
  if (n == 0) return VRFHIP_SUCCESS;
  if (ctx->sw) {
    // secp256r1: launch groups of at most 2^20 proofs, each ONE MSM over 5 m + 2 points; no per-proof workspace at all
    const size_t cap = std::min<size_t>(n, size_t(1) << 20);
    const size_t Ncap = 5 * cap + 2;
    auto p256_groups = [&](size_t N_, size_t long_) {
      int g = ctx->dbg_p256_msm_groups > 0 ? ctx->dbg_p256_msm_groups : p256::msm_groups(N_, long_, ctx->cus);
      const size_t min_g = (N_ + (size_t(1) << 21) - 1) >> 21;
      return (size_t)g < min_g ? (int)min_g : g;
    };
    const int groups_cap = p256_groups(Ncap, 3 * cap + 2);
    const size_t msm_b = Stage::pad(p256::msm_workspace_bytes(Ncap, groups_cap));
    int32_t rc2 = ensure_msm_workspace(ctx, msm_b + Stage::pad(digest_ws_bytes(cap)) + 256);
    if (rc2) return rc2;
    uint8_t* d_dws = static_cast<uint8_t*>(ctx->d_msm_ws) + msm_b;
    uint8_t* d_rt = d_dws + Stage::pad(digest_ws_bytes(cap));
    for (size_t base = 0; base < n; base += cap) {
      const size_t m = std::min(cap, n - base), N = 5 * m + 2;
      p256::RlcArgs a{};
      a.n = m;
      a.index0 = base;
      const size_t pw = affine ? 64 : 33;
      a.h = d_input + base * pw; a.gamma = d_output + base * pw; a.pk_com = d_pk_com + base * pw;
      a.r = d_r + base * pw; a.ok = d_ok + base * pw; a.s = d_s + base * 32; a.sb = d_sb + base * 32;
      a.affine_in = affine ? (ctx->coords_mont256() ? 2 : 1) : 0;
      a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
      a.status = d_status + base;
      a.L = p256::msm_layout(N, 3 * m + 2, p256_groups(N, 3 * m + 2), ctx->d_msm_ws);
      std::memcpy(a.seed, seed, 32);
      std::memcpy(a.gen_xy, ctx->desc.generator, 64);
      std::memcpy(a.b_xy, ctx->desc.blinding_base, 64);
      a.str = ctx->T.sq.str;
      DigestSrc ds{};
      const uint8_t* arr[7] = {a.h, a.gamma, a.pk_com, a.r, a.ok, a.s, a.sb};
      for (int j = 0; j < 7; ++j) { ds.p[j] = arr[j]; ds.w[j] = j < 5 ? (uint32_t)pw : 32u; }
      ds.n_arr = 7;
      ds.ad = a.ad;
      launch_batch_digest(ds, m, base, d_dws, d_rt, st);
      a.root = d_rt;
      p256::launch_pedersen_rlc(a, d_fail_flag, st, prof_events(ctx));
    }
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  int32_t rc = ensure_workspace(ctx, n);
  if (rc) return rc;
  size_t msm_bytes;
  {
    size_t m = std::min(ctx->ws_cap, n), N = 5 * m + 2;
    msm_bytes = Stage::pad(msm_workspace_bytes(N, msm_groups(N, 3 * m + 2, ctx->cus)));
    rc = ensure_msm_workspace(ctx, msm_bytes + Stage::pad(digest_ws_bytes(m)) + 256);   // MSM | digest tree | root
    if (rc) return rc;
  }
  uint8_t* d_digest_ws = static_cast<uint8_t*>(ctx->d_msm_ws) + msm_bytes;
  uint8_t* d_root = d_digest_ws + Stage::pad(digest_ws_bytes(std::min(ctx->ws_cap, n)));
  for (size_t base = 0; base < n; base += ctx->ws_cap) {
    size_t m = std::min(ctx->ws_cap, n - base), N = 5 * m + 2;
    RlcArgs a;
    a.suite = (int)ctx->suite;
    a.k_lane = lanes_k(m, VERIFY_K_POLICY);
    a.n = m;
    a.index0 = base;
    const size_t pw = affine ? 64 : ctx->pt_bytes();
    a.h = d_input + base * pw; a.gamma = d_output + base * pw; a.pk_com = d_pk_com + base * pw;
    a.r = d_r + base * pw; a.ok = d_ok + base * pw; a.s = d_s + base * 32; a.sb = d_sb + base * 32;
    a.affine_in = affine ? (ctx->coords_mont256() ? 2 : 1) : 0;
    a.check_mask = ctx->check_mask();
    a.ad = make_view(d_ad, d_ad_off ? d_ad_off + base : nullptr, ad_len, true);
    a.status = d_status + base;
    a.scratch = ctx->ws.tabs;
    a.scratch_stride = WS_TABS * WIN_TABLE_WORDS;
    a.L = msm_layout(N, 3 * m + 2, msm_groups(N, 3 * m + 2, ctx->cus), ctx->d_msm_ws);
    a.fixed_cols = reinterpret_cast<uint64_t*>(a.L.flags + 128);
    a.T = ctx->T;
    std::memcpy(a.seed, seed, 32);
    // the weights of this launch group depend on every input byte of the group (digest.cuh)
    DigestSrc ds{};
    const uint8_t* arr[7] = {a.h, a.gamma, a.pk_com, a.r, a.ok, a.s, a.sb};
    for (int j = 0; j < 7; ++j) { ds.p[j] = arr[j]; ds.w[j] = j < 5 ? (uint32_t)pw : 32u; }
    ds.n_arr = 7;
    ds.ad = a.ad;
    launch_batch_digest(ds, m, base, d_digest_ws, d_root, st);
    a.root = d_root;
    if (ctx->bsw) launch_bsw_pedersen_rlc(a, d_fail_flag, st, prof_events(ctx));
    else FIELD_CALL(ctx, launch_pedersen_rlc(a, d_fail_flag, st, prof_events(ctx)));
  }
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t rlc_host_impl(vrfhip_ctx* ctx, size_t n, bool affine, const uint8_t* input,
                      const uint8_t* output, const uint8_t* pk_com, const uint8_t* r,
                      const uint8_t* ok, const uint8_t* s, const uint8_t* sb,
                      const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                      const uint8_t seed[32], uint8_t* status, int32_t* batch_ok) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (!seed) return fail(VRFHIP_ERR_BAD_ARG, "seed is NULL");
  if (batch_ok) *batch_ok = 1;
  if (n == 0) return VRFHIP_SUCCESS;
  if (!input || !output || !pk_com || !r || !ok || !s || !sb || !status)
    return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if ((ad_len || ad_off) && !ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  size_t adb = blob_bytes(n, ad_off, ad_len, true);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  const size_t pw = affine ? 64 : ctx->pt_bytes();
  size_t need = 5 * Stage::pad(n * pw) + (affine ? 5 : 0) * Stage::pad(n * ctx->pt_bytes()) + 2 * Stage::pad(n * 32) +
                Stage::pad(adb + 1) + Stage::pad((n + 1) * 4) + Stage::pad(n) + 256;
  int32_t rc = ensure_stage(ctx, need);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  const uint8_t* src[7] = {input, output, pk_com, r, ok, s, sb};
  uint8_t* d[7];
  for (int i = 0; i < 7; ++i) {
    const size_t w = i < 5 ? pw : 32;
    d[i] = sg.take(n * w);
    HIP_TRY(hipMemcpyAsync(d[i], src[i], n * w, hipMemcpyHostToDevice, ctx->stream));
  }
  uint8_t* d_ad = sg.take(adb + 1);
  uint32_t* d_off = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  uint8_t* d_st = sg.take(n);
  uint8_t* d_flag = sg.take(1);
  if (adb) HIP_TRY(hipMemcpyAsync(d_ad, ad, adb, hipMemcpyHostToDevice, ctx->stream));
  if (ad_off) HIP_TRY(hipMemcpyAsync(d_off, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  rc = rlc_dev_impl(ctx, n, affine, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d_ad,
                    ad_off ? d_off : nullptr, ad_len, seed, d_st, d_flag, ctx->stream);
  if (rc) return rc;
  uint8_t flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, d_flag, 1, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (flag) {
    // some proof in the batch is wrong: the per-proof kernels say which.  Items the batch stage already
    // rejected as InvalidData keep that status (an off-curve affine point has no compressed form).
    if (batch_ok) *batch_ok = 0;
    if (affine) {
      const size_t ew = ctx->pt_bytes();
      for (int i = 0; i < 5; ++i) {
        uint8_t* enc = sg.take(n * ew);
        const int m256 = ctx->coords_mont256() ? 1 : 0;
        if (ctx->sw) p256::launch_affine_compress(n, d[i], m256, enc, ctx->stream);
        else if (ctx->bsw) launch_bsw_affine_compress(n, d[i], m256, enc, ctx->stream);
        else {
          if (m256) FIELD_CALL(ctx, launch_xy_from_mont256(n, d[i], ctx->stream));      // the staged copy, in place
          FIELD_CALL(ctx, launch_affine_compress(n, d[i], enc, ctx->T.sq.str.flags, ctx->stream));
        }
        d[i] = enc;
      }
    }
    rc = vrfhip_pedersen_verify_batch_dev(ctx, n, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d_ad,
                                          ad_off ? d_off : nullptr, ad_len, d_st, ctx->stream);
    if (rc) return rc;
    std::vector<uint8_t> item(n);
    HIP_TRY(hipMemcpyAsync(item.data(), d_st, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i)
      if (status[i] != VRFHIP_ST_INVALID_DATA) status[i] = item[i];
  }
  return VRFHIP_SUCCESS;
}
}  // namespace

extern "C" {

int32_t vrfhip_pedersen_verify_batch_rlc_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_input,
                                             const uint8_t* d_output, const uint8_t* d_pk_com,
                                             const uint8_t* d_r, const uint8_t* d_ok, const uint8_t* d_s,
                                             const uint8_t* d_sb, const uint8_t* d_ad,
                                             const uint32_t* d_ad_off, uint32_t ad_len,
                                             const uint8_t seed[32], uint8_t* d_status,
                                             uint8_t* d_fail_flag, void* stream) {
  return rlc_dev_impl(ctx, n, false, d_input, d_output, d_pk_com, d_r, d_ok, d_s, d_sb, d_ad, d_ad_off, ad_len,
                      seed, d_status, d_fail_flag, stream);
}
int32_t vrfhip_pedersen_verify_batch_rlc_affine_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_input_xy,
                                                    const uint8_t* d_output_xy, const uint8_t* d_pk_com_xy,
                                                    const uint8_t* d_r_xy, const uint8_t* d_ok_xy,
                                                    const uint8_t* d_s, const uint8_t* d_sb,
                                                    const uint8_t* d_ad, const uint32_t* d_ad_off,
                                                    uint32_t ad_len, const uint8_t seed[32],
                                                    uint8_t* d_status, uint8_t* d_fail_flag, void* stream) {
  return rlc_dev_impl(ctx, n, true, d_input_xy, d_output_xy, d_pk_com_xy, d_r_xy, d_ok_xy, d_s, d_sb, d_ad,
                      d_ad_off, ad_len, seed, d_status, d_fail_flag, stream);
}
int32_t vrfhip_pedersen_verify_batch_rlc(vrfhip_ctx* ctx, size_t n, const uint8_t* input,
                                         const uint8_t* output, const uint8_t* pk_com, const uint8_t* r,
                                         const uint8_t* ok, const uint8_t* s, const uint8_t* sb,
                                         const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                         const uint8_t seed[32], uint8_t* status, int32_t* batch_ok) {
  return rlc_host_impl(ctx, n, false, input, output, pk_com, r, ok, s, sb, ad, ad_off, ad_len, seed, status,
                       batch_ok);
}
int32_t vrfhip_pedersen_verify_batch_rlc_affine(vrfhip_ctx* ctx, size_t n, const uint8_t* input_xy,
                                                const uint8_t* output_xy, const uint8_t* pk_com_xy,
                                                const uint8_t* r_xy, const uint8_t* ok_xy, const uint8_t* s,
                                                const uint8_t* sb, const uint8_t* ad, const uint32_t* ad_off,
                                                uint32_t ad_len, const uint8_t seed[32], uint8_t* status,
                                                int32_t* batch_ok) {
  return rlc_host_impl(ctx, n, true, input_xy, output_xy, pk_com_xy, r_xy, ok_xy, s, sb, ad, ad_off, ad_len,
                       seed, status, batch_ok);
}

// ------------------------------------------------------------------------- MSM
int32_t vrfhip_msm_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_bases_xy, const uint8_t* d_scalars,
                       uint8_t* d_out_point, uint8_t* d_out_xy, uint8_t* d_status, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (!d_out_point || !d_status) return fail(VRFHIP_ERR_BAD_ARG, "NULL output");
  if (n && (!d_bases_xy || !d_scalars)) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (ctx->bsw) {
    // bandersnatch_sw: the bases are Weierstrass x || y.  They cross te_sw_map into the MSM workspace, the Edwards kernels add
    // them up, and the sum goes back: a 33-byte encoding (infinity for the empty sum) and Weierstrass x || y
    const size_t m = n ? n : 1;
    const int groups = msm_groups(m, m, ctx->cus);
    const size_t core = msm_workspace_bytes(m, groups);
    int32_t rc = ensure_msm_workspace(ctx, core + Stage::pad(m * 64) + Stage::pad(m) + 256);
    if (rc) return rc;
    uint8_t* d_map = static_cast<uint8_t*>(ctx->d_msm_ws) + core;
    uint8_t* d_mst = d_map + Stage::pad(m * 64);
    uint8_t* d_sum = d_mst + Stage::pad(m);             // 64 B Edwards x || y, then 32 B of its (unused) compressed form
    if (n == 0) {
      uint8_t inf[33] = {0};
      inf[32] = 0x40;
      HIP_TRY(hipMemcpyAsync(d_out_point, inf, 33, hipMemcpyHostToDevice, st));
      if (d_out_xy) HIP_TRY(hipMemsetAsync(d_out_xy, 0, 64, st));
      HIP_TRY(hipMemsetAsync(d_status, 0, 1, st));
      HIP_TRY(hipStreamSynchronize(st));     // the source above is a stack buffer
      return VRFHIP_SUCCESS;
    }
    const int m256 = ctx->coords_mont256() ? 1 : 0;
    vrf::f_bls381fr::launch_te_sw_map(SUITE_BS, n, 1, m256, d_bases_xy, d_map, d_mst, st);
    vrf::f_bls381fr::launch_msm_coords(SUITE_BS, n, d_map, d_scalars, d_sum + 64, d_sum, d_status, ctx->d_msm_ws, groups, m256, 0u, st);
    launch_bsw_msm_out(d_sum, d_out_point, d_out_xy, d_status, st);
    if (d_out_xy && m256) vrf::f_bls381fr::launch_xy_to_mont256(1, d_out_xy, st);
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  if (ctx->sw) {
    // secp256r1: big-endian scalars, the sum as a 33-byte Sec1 string (0x00 + zeros = the point at infinity)
    const int groups = p256::msm_groups(n ? n : 1, n ? n : 1, ctx->cus);
    int32_t rc = ensure_msm_workspace(ctx, p256::msm_workspace_bytes(n ? n : 1, groups));
    if (rc) return rc;
    p256::launch_msm(n, d_bases_xy, ctx->coords_mont256() ? 1 : 0, d_scalars, d_out_point, d_out_xy, d_status, ctx->d_msm_ws, groups, st);
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  if (n == 0) {                       // empty sum: the identity (0, 1)
    uint8_t id[64] = {0};
    id[32] = 1;
    uint8_t enc[32] = {1};
    HIP_TRY(hipMemcpyAsync(d_out_point, enc, 32, hipMemcpyHostToDevice, st));
    if (d_out_xy) HIP_TRY(hipMemcpyAsync(d_out_xy, id, 64, hipMemcpyHostToDevice, st));
    if (d_out_xy && ctx->coords_mont256()) FIELD_CALL(ctx, launch_xy_to_mont256(1, d_out_xy, st));
    HIP_TRY(hipMemsetAsync(d_status, 0, 1, st));
    HIP_TRY(hipStreamSynchronize(st));     // the sources above are stack buffers
    return VRFHIP_SUCCESS;
  }
  int groups = msm_groups(n, n, ctx->cus);
  int32_t rc = ensure_msm_workspace(ctx, msm_workspace_bytes(n, groups));
  if (rc) return rc;
  FIELD_CALL(ctx, launch_msm_coords((int)ctx->suite, n, d_bases_xy, d_scalars, d_out_point, d_out_xy, d_status, ctx->d_msm_ws, groups,
                    ctx->coords_mont256() ? 1 : 0, ctx->T.sq.str.flags, st));
  if (d_out_xy && ctx->coords_mont256()) FIELD_CALL(ctx, launch_xy_to_mont256(1, d_out_xy, st));
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_msm(vrfhip_ctx* ctx, size_t n, const uint8_t* bases_xy, const uint8_t* scalars,
                   uint8_t* out_point, uint8_t* out_xy, uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (!out_point || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL output");
  if (n && (!bases_xy || !scalars)) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, Stage::pad(n * 64 + 1) + Stage::pad(n * 32 + 1) + 3 * 256);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_xy = sg.take(n * 64 + 1);
  uint8_t* d_k = sg.take(n * 32 + 1);
  uint8_t* d_out = sg.take(64);             // 32 (Edwards) or 33 (Sec1) bytes are used
  uint8_t* d_oxy = sg.take(64);
  uint8_t* d_st = sg.take(1);
  if (n) {
    HIP_TRY(hipMemcpyAsync(d_xy, bases_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_k, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  }
  rc = vrfhip_msm_dev(ctx, n, d_xy, d_k, d_out, d_oxy, d_st, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out_point, d_out, ctx->pt_bytes(), hipMemcpyDeviceToHost, ctx->stream));
  if (out_xy) HIP_TRY(hipMemcpyAsync(out_xy, d_oxy, 64, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(status, d_st, 1, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

// ------------------------------------------------------------------------- pairing check
int32_t vrfhip_pairing_check_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_g1,
                                       const uint8_t* d_g2, int32_t g2_shared, uint8_t* d_status,
                                       void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_g1 || !d_g2 || !d_status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  {
    // Per-item G2 points: the G2 walk and the Miller
    // loop as two kernels with the lines parked in HBM between them (k_pairing_oct.hip); layout 6 = the same work as ONE
    // kernel, kept for comparison.
    const int lay = ctx->dbg_pairing_layout, mode = lay & 0xff;
    const bool unprepared = !g2_shared || (lay & PAIRING_NOPREP) != 0;
    if (unprepared && (mode == PAIRING_OCT || (mode == PAIRING_AUTO && n >= PAIRING_OCT_SPLIT_MIN_ITEMS))) {
      const size_t chunk = std::min<size_t>(n, PAIRING_OCT_SPLIT_CHUNK);
      int32_t rc = ensure_msm_workspace(ctx, pairing_oct_lines_bytes(chunk));
      if (rc) return rc;
      launch_pairing_check2_oct_split(n, d_g1, d_g2, g2_shared ? 0 : 384, d_status, ctx->d_msm_ws, chunk, static_cast<hipStream_t>(stream));
      HIP_TRY(hipGetLastError());
      return VRFHIP_SUCCESS;
    }
    launch_pairing_check2(n, d_g1, d_g2, g2_shared ? 0 : 384, d_status, static_cast<hipStream_t>(stream),
                          g2_shared ? ctx->d_pair_prep : nullptr, mode == 6 ? ((lay & ~0xff) | PAIRING_OCT) : lay);
  }
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_pairing_check_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2,
                                   int32_t g2_shared, uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!g1 || !g2 || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  size_t g2b = g2_shared ? 384 : n * 384;
  int32_t rc = ensure_stage(ctx, Stage::pad(n * 192) + Stage::pad(g2b) + Stage::pad(n));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_g1 = sg.take(n * 192);
  uint8_t* d_g2 = sg.take(g2b);
  uint8_t* d_st = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_g1, g1, n * 192, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_g2, g2, g2b, hipMemcpyHostToDevice, ctx->stream));
  rc = vrfhip_pairing_check_batch_dev(ctx, n, d_g1, d_g2, g2_shared, d_st, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

// ------------------------------------------------------------------------- G1 MSM, batched pairing check
int32_t vrfhip_g1_msm_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_bases, const uint8_t* d_scalars, uint8_t* d_out,
                          uint8_t* d_status, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (!d_out || !d_status) return fail(VRFHIP_ERR_BAD_ARG, "NULL output");
  if (n && (!d_bases || !d_scalars)) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if (n > (size_t(1) << 28)) return fail(VRFHIP_ERR_BAD_ARG, "batch too large");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int groups = g1_msm_groups(n ? n : 1, 1, G1_W_FULL, ctx->cus);
  int32_t rc = ensure_msm_workspace(ctx, g1_msm_workspace_bytes(n ? n : 1, 1, G1_W_FULL, groups));
  if (rc) return rc;
  G1MsmLayout L = g1_msm_layout(n, 1, G1_W_FULL, groups, ctx->d_msm_ws);
  launch_g1_msm(L, d_bases, d_scalars, d_status, st);
  HIP_TRY(hipMemcpyAsync(d_out, L.sums, 96, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_g1_msm(vrfhip_ctx* ctx, size_t n, const uint8_t* bases, const uint8_t* scalars, uint8_t* out,
                      uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (!out || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL output");
  if (n && (!bases || !scalars)) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, Stage::pad(n * 96 + 1) + Stage::pad(n * 32 + 1) + 2 * 256);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_b = sg.take(n * 96 + 1);
  uint8_t* d_k = sg.take(n * 32 + 1);
  uint8_t* d_o = sg.take(96);
  uint8_t* d_st = sg.take(1);
  if (n) {
    HIP_TRY(hipMemcpyAsync(d_b, bases, n * 96, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_k, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  }
  rc = vrfhip_g1_msm_dev(ctx, n, d_b, d_k, d_o, d_st, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, d_o, 96, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(status, d_st, 1, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (status[0] != VRFHIP_ST_OK) std::memset(out, 0, 96);
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_pairing_check_batch_rlc_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_g1, const uint8_t* d_g2_shared,
                                           const uint8_t seed[32], uint8_t* d_status, uint8_t* d_verdict, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (!d_verdict || !seed) return fail(VRFHIP_ERR_BAD_ARG, "NULL verdict or seed");
  if (n && (!d_g1 || !d_g2_shared || !d_status)) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  if (n > (size_t(1) << 28)) return fail(VRFHIP_ERR_BAD_ARG, "batch too large");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemsetAsync(d_verdict, 0, 1, st));
  if (n == 0) return VRFHIP_SUCCESS;
  const int groups = g1_msm_groups(n, 2, G1_W_SHORT, ctx->cus);
  const size_t msm_bytes = Stage::pad(g1_msm_workspace_bytes(n, 2, G1_W_SHORT, groups));
  int32_t rc = ensure_msm_workspace(ctx, msm_bytes + Stage::pad(digest_ws_bytes(n)) + 256);   // MSM | digest tree | root
  if (rc) return rc;
  G1MsmLayout L = g1_msm_layout(n, 2, G1_W_SHORT, groups, ctx->d_msm_ws);
  uint8_t* d_digest_ws = static_cast<uint8_t*>(ctx->d_msm_ws) + msm_bytes;
  uint8_t* d_root = d_digest_ws + Stage::pad(digest_ws_bytes(n));
  hipEvent_t* ev = prof_events(ctx);            // start | prep | buckets | final | pairing
  if (ev) (void)hipEventRecord(ev[0], st);
  DigestSrc ds{};
  ds.p[0] = d_g1; ds.w[0] = 192; ds.n_arr = 1;
  // ... and on the shared G2 pair, absorbed as the leaves' common byte string (with the pair outside the digest a
  // caller-influenced pair could be chosen after the weights, Q1 = a Q0 with a = -(sum z alpha) / (sum z beta))
  ds.ad = make_view(d_g2_shared, nullptr, 384, true);
  launch_batch_digest(ds, n, 0, d_digest_ws, d_root, st);   // the weights depend on every byte of the batch
  launch_g1_rlc(L, d_g1, seed, d_root, 0, d_status, st, ev ? ev + 1 : nullptr);
  // one pairing check for the whole batch: (sum z A, sum z B) against the shared pair (prepared lines)
  launch_pairing_check2(1, L.sums, d_g2_shared, 0, d_verdict, st, ctx->d_pair_prep, ctx->dbg_pairing_layout);
  if (ev) (void)hipEventRecord(ev[4], st);
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_pairing_check_batch_rlc(vrfhip_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2_shared,
                                       const uint8_t seed[32], uint8_t* status, int32_t* batch_ok) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (!seed) return fail(VRFHIP_ERR_BAD_ARG, "seed is NULL");
  if (batch_ok) *batch_ok = 1;
  if (n == 0) return VRFHIP_SUCCESS;
  if (!g1 || !g2_shared || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, Stage::pad(n * 192) + Stage::pad(384) + Stage::pad(n) + 256);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_g1 = sg.take(n * 192);
  uint8_t* d_g2 = sg.take(384);
  uint8_t* d_st = sg.take(n);
  uint8_t* d_v = sg.take(1);
  HIP_TRY(hipMemcpyAsync(d_g1, g1, n * 192, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_g2, g2_shared, 384, hipMemcpyHostToDevice, ctx->stream));
  rc = vrfhip_pairing_check_batch_rlc_dev(ctx, n, d_g1, d_g2, seed, d_st, d_v, ctx->stream);
  if (rc) return rc;
  uint8_t verdict = 0;
  HIP_TRY(hipMemcpyAsync(&verdict, d_v, 1, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (verdict != 0) {
    // some item is false (or the shared pair is invalid): the per-item kernel names it
    if (batch_ok) *batch_ok = 0;
    launch_pairing_check2(n, d_g1, d_g2, 0, d_st, ctx->stream, ctx->d_pair_prep, ctx->dbg_pairing_layout);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

// Test-only: quad-distributed Fp12 tower operations against the one-lane operations (k_pairing.hip)
int32_t vrfhip_test_pairing_quad_ops(vrfhip_ctx* ctx, size_t n, const uint8_t* fp12_pairs, uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!fp12_pairs || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, Stage::pad(n * 1152) + Stage::pad(n));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_in = sg.take(n * 1152);
  uint8_t* d_st = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_in, fp12_pairs, n * 1152, hipMemcpyHostToDevice, ctx->stream));
  launch_pairing_quad_selftest(n, d_in, d_st, ctx->stream);
  launch_pairing_row_selftest(n, d_in, d_st, ctx->stream);        // bits 64 / 128: the row-distributed tower
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

// Test-only: the 8-lanes-per-item Fp12 tower (bls12_oct.cuh) and its cross-lane moves against the one-lane operations
int32_t vrfhip_test_pairing_oct_ops(vrfhip_ctx* ctx, size_t n, const uint8_t* fp12_pairs, uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!fp12_pairs || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, Stage::pad(n * 1152) + Stage::pad(n));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_in = sg.take(n * 1152);
  uint8_t* d_st = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_in, fp12_pairs, n * 1152, hipMemcpyHostToDevice, ctx->stream));
  launch_pairing_oct_selftest(n, d_in, d_st, ctx->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

// ------------------------------------------------------------------------- building blocks
int32_t vrfhip_hash_to_curve_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_msg,
                                       const uint32_t* d_msg_off, uint32_t msg_len,
                                       uint8_t* d_points, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_points || (!d_msg && (msg_len || d_msg_off))) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  // The try-and-increment suites find their counters with the provers' work-queue search, which wants one byte per item: a
  // buffer of its own (n bytes, at most 2^20 per chunk) -- not the prove / verify workspace (7 KiB per item), which this
  // entry point neither needs nor should resize.
  const size_t chunk = std::min<size_t>(n, size_t(1) << 20);
  if (ctx->sw || ctx->bsw || ctx->suite != VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2) {
    if (ctx->h2c_cap < chunk) {
      if (ctx->d_h2c_ctr) { (void)hipStreamSynchronize(static_cast<hipStream_t>(stream)); (void)hipFree(ctx->d_h2c_ctr); ctx->d_h2c_ctr = nullptr; ctx->h2c_cap = 0; }
      if (hipMalloc(&ctx->d_h2c_ctr, chunk) != hipSuccess) return fail(VRFHIP_ERR_OOM, "hipMalloc(hash-to-curve counters) failed");
      ctx->h2c_cap = chunk;
    }
  }
  if (ctx->sw) {
    for (size_t base = 0; base < n; base += chunk) {
      const size_t m = std::min(chunk, n - base);
      BytesView mv = d_msg_off ? make_view(d_msg, d_msg_off + base, msg_len, false)
                               : make_view(d_msg ? d_msg + base * (size_t)msg_len : d_msg, nullptr, msg_len, false);
      p256::launch_hash_to_curve(m, mv, d_points + base * 33, ctx->T.sq.str, static_cast<hipStream_t>(stream), ctx->d_h2c_ctr,
                                 ctx->d_queue);
    }
  } else if (ctx->bsw) {
    for (size_t base = 0; base < n; base += chunk) {
      const size_t m = std::min(chunk, n - base);
      BytesView mv = d_msg_off ? make_view(d_msg, d_msg_off + base, msg_len, false)
                               : make_view(d_msg ? d_msg + base * (size_t)msg_len : d_msg, nullptr, msg_len, false);
      launch_bsw_hash_to_curve(m, mv, d_points + base * 33, ctx->T, static_cast<hipStream_t>(stream), ctx->d_h2c_ctr, ctx->d_queue);
    }
  } else if (ctx->suite == VRFHIP_SUITE_BANDERSNATCH_SHA512_ELL2) {
    FIELD_CALL(ctx, launch_hash_to_curve((int)ctx->suite, n, make_view(d_msg, d_msg_off, msg_len, false), d_points, ctx->T,
                                         static_cast<hipStream_t>(stream)));
  } else {
    for (size_t base = 0; base < n; base += chunk) {
      const size_t m = std::min(chunk, n - base);
      BytesView mv = d_msg_off ? make_view(d_msg, d_msg_off + base, msg_len, false)
                               : make_view(d_msg ? d_msg + base * (size_t)msg_len : d_msg, nullptr, msg_len, false);
      FIELD_CALL(ctx, launch_hash_to_curve((int)ctx->suite, m, mv, d_points + base * 32, ctx->T, static_cast<hipStream_t>(stream),
                                           ctx->d_h2c_ctr, ctx->d_queue));
    }
  }
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_hash_to_curve_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* msg,
                                   const uint32_t* msg_off, uint32_t msg_len, uint8_t* points) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!points || (!msg && (msg_len || msg_off))) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  size_t msgb = blob_bytes(n, msg_off, msg_len, false);
  uint8_t *d_msg, *d_pts;
  uint32_t* d_off;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  {
    int32_t rc = ensure_stage(ctx, Stage::pad(msgb + 1) + Stage::pad((n + 1) * 4) + Stage::pad(n * ctx->pt_bytes()));
    if (rc) return rc;
    Stage sg(ctx->d_stage);
    d_msg = sg.take(msgb + 1);
    d_off = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
    d_pts = sg.take(n * ctx->pt_bytes());
    if (msgb) HIP_TRY(hipMemcpyAsync(d_msg, msg, msgb, hipMemcpyHostToDevice, ctx->stream));
    if (msg_off) HIP_TRY(hipMemcpyAsync(d_off, msg_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  int32_t rc = vrfhip_hash_to_curve_batch_dev(ctx, n, d_msg, msg_off ? d_off : nullptr, msg_len, d_pts,
                                              ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(points, d_pts, n * ctx->pt_bytes(), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_output_hash_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_output,
                                     uint8_t* d_hash, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_output || !d_hash) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (ctx->sw) p256::launch_output_hash(n, d_output, d_hash, ctx->T.sq.str, static_cast<hipStream_t>(stream));
  else if (ctx->bsw) launch_bsw_output_hash(n, d_output, d_hash, ctx->T, static_cast<hipStream_t>(stream));
  else
  FIELD_CALL(ctx, launch_output_hash((int)ctx->suite, n, d_output, d_hash, ctx->T, static_cast<hipStream_t>(stream)));
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_output_hash_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* output, uint8_t* hash) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!output || !hash) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  uint8_t *d_in, *d_out;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  {
    int32_t rc = ensure_stage(ctx, Stage::pad(n * ctx->pt_bytes()) + Stage::pad(n * 64));
    if (rc) return rc;
    Stage sg(ctx->d_stage);
    d_in = sg.take(n * ctx->pt_bytes());
    d_out = sg.take(n * 64);
    HIP_TRY(hipMemcpyAsync(d_in, output, n * ctx->pt_bytes(), hipMemcpyHostToDevice, ctx->stream));
  }
  int32_t rc = vrfhip_output_hash_batch_dev(ctx, n, d_in, d_out, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(hash, d_out, n * ctx->hash_bytes(), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_secret_from_seed_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_seeds,
                                          uint32_t seed_len, uint8_t* d_sk_out, uint8_t* d_pk_out,
                                          void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_sk_out || (!d_seeds && seed_len)) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (ctx->sw) p256::launch_secret_from_seed(n, d_seeds, seed_len, d_sk_out, d_pk_out, ctx->d_p256_comb, static_cast<hipStream_t>(stream));
  else if (ctx->bsw) launch_bsw_secret_from_seed(n, d_seeds, seed_len, d_sk_out, d_pk_out, ctx->T, static_cast<hipStream_t>(stream));
  else
  FIELD_CALL(ctx, launch_secret_from_seed((int)ctx->suite, n, d_seeds, seed_len, d_sk_out, d_pk_out, ctx->T,
                          static_cast<hipStream_t>(stream)));
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_secret_from_seed_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* seeds,
                                      uint32_t seed_len, uint8_t* sk_out, uint8_t* pk_out) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!sk_out || (!seeds && seed_len)) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  uint8_t *d_seed, *d_sk, *d_pk;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  {
    int32_t rc = ensure_stage(ctx, Stage::pad(n * (size_t)seed_len + 1) + Stage::pad(n * 32) + Stage::pad(n * ctx->pt_bytes()));
    if (rc) return rc;
    Stage sg(ctx->d_stage);
    d_seed = sg.take(n * (size_t)seed_len + 1);
    d_sk = sg.take(n * 32);
    d_pk = sg.take(n * ctx->pt_bytes());
    if (seed_len)
      HIP_TRY(hipMemcpyAsync(d_seed, seeds, n * (size_t)seed_len, hipMemcpyHostToDevice, ctx->stream));
  }
  int32_t rc = vrfhip_secret_from_seed_batch_dev(ctx, n, d_seed, seed_len, d_sk, pk_out ? d_pk : nullptr,
                                                 ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(sk_out, d_sk, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  if (pk_out) HIP_TRY(hipMemcpyAsync(pk_out, d_pk, n * ctx->pt_bytes(), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemsetAsync(d_sk, 0, n * 32, ctx->stream));                // staged secrets and their seeds
  if (seed_len) HIP_TRY(hipMemsetAsync(d_seed, 0, n * (size_t)seed_len, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_point_validate_batch_dev(vrfhip_ctx* ctx, size_t n, const uint8_t* d_points,
                                        uint8_t* d_xy_out, uint8_t* d_status, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_points || !d_status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (ctx->sw) {
    p256::launch_point_validate(n, d_points, d_xy_out, ctx->coords_mont256() ? 1 : 0, d_status, static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  if (ctx->bsw) {                  // x || y out: the short-Weierstrass coordinates
    launch_bsw_point_validate(n, d_points, d_xy_out, d_status, ctx->T, static_cast<hipStream_t>(stream));
    if (d_xy_out && ctx->coords_mont256()) FIELD_CALL(ctx, launch_xy_to_mont256(n, d_xy_out, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipGetLastError());
    return VRFHIP_SUCCESS;
  }
  int32_t rc = ensure_workspace(ctx, n);
  if (rc) return rc;
  // one window table per item: the tabs region holds WS_TABS per workspace item
  size_t cap = ctx->ws_cap * WS_TABS;
  for (size_t base = 0; base < n; base += cap) {
    size_t m = std::min(cap, n - base);
    FIELD_CALL(ctx, launch_point_validate((int)ctx->suite, m, d_points + base * 32, d_xy_out ? d_xy_out + base * 64 : nullptr,
                          d_status + base, ctx->ws.tabs, ctx->T, static_cast<hipStream_t>(stream)));
    if (d_xy_out && ctx->coords_mont256()) FIELD_CALL(ctx, launch_xy_to_mont256(m, d_xy_out + base * 64, static_cast<hipStream_t>(stream)));
  }
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_point_validate_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* points,
                                    uint8_t* xy_out, uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!points || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  uint8_t *d_in, *d_xy, *d_st;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  {
    int32_t rc = ensure_stage(ctx, Stage::pad(n * ctx->pt_bytes()) + Stage::pad(n * 64) + Stage::pad(n));
    if (rc) return rc;
    Stage sg(ctx->d_stage);
    d_in = sg.take(n * ctx->pt_bytes());
    d_xy = sg.take(n * 64);
    d_st = sg.take(n);
    HIP_TRY(hipMemcpyAsync(d_in, points, n * ctx->pt_bytes(), hipMemcpyHostToDevice, ctx->stream));
  }
  int32_t rc = vrfhip_point_validate_batch_dev(ctx, n, d_in, xy_out ? d_xy : nullptr, d_st, ctx->stream);
  if (rc) return rc;
  if (xy_out) HIP_TRY(hipMemcpyAsync(xy_out, d_xy, n * 64, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_te_sw_map_batch_dev(vrfhip_ctx* ctx, size_t n, int32_t to_te, const uint8_t* d_in_xy, uint8_t* d_out_xy,
                                   uint8_t* d_status, void* stream) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "te_sw_map: the context's curve is not twisted Edwards");
  if (to_te != 0 && to_te != 1) return fail(VRFHIP_ERR_BAD_ARG, "to_te must be 0 (TE -> SW) or 1 (SW -> TE)");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!d_in_xy || !d_out_xy || !d_status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  FIELD_CALL(ctx, launch_te_sw_map((int)ctx->suite, n, to_te, ctx->coords_mont256() ? 1 : 0, d_in_xy, d_out_xy, d_status,
                                   static_cast<hipStream_t>(stream)));
  HIP_TRY(hipGetLastError());
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_te_sw_map_batch(vrfhip_ctx* ctx, size_t n, int32_t to_te, const uint8_t* in_xy, uint8_t* out_xy,
                               uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "te_sw_map: the context's curve is not twisted Edwards");
  if (to_te != 0 && to_te != 1) return fail(VRFHIP_ERR_BAD_ARG, "to_te must be 0 (TE -> SW) or 1 (SW -> TE)");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!in_xy || !out_xy || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, 2 * Stage::pad(n * 64) + Stage::pad(n));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_in = sg.take(n * 64);
  uint8_t* d_out = sg.take(n * 64);
  uint8_t* d_st = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_in, in_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
  rc = vrfhip_te_sw_map_batch_dev(ctx, n, to_te, d_in, d_out, d_st, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out_xy, d_out, n * 64, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_fq_mul_batch(vrfhip_ctx* ctx, size_t n, const uint8_t* a, const uint8_t* b,
                            uint8_t* r) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (n == 0) return VRFHIP_SUCCESS;
  if (!a || !b || !r) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, 3 * Stage::pad(n * 32));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_a = sg.take(n * 32);
  uint8_t* d_b = sg.take(n * 32);
  uint8_t* d_r = sg.take(n * 32);
  HIP_TRY(hipMemcpyAsync(d_a, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_b, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
  FIELD_CALL(ctx, launch_fq_mul(n, d_a, d_b, d_r, ctx->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(r, d_r, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}


// ------------------------------------------------------------------------- test primitives
int32_t vrfhip_test_point_add(vrfhip_ctx* ctx, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out,
                              uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (ctx->bsw) return fail(VRFHIP_ERR_UNSUPPORTED, BSW_UNSUPPORTED_MSG);
  if (n == 0) return VRFHIP_SUCCESS;
  if (!a || !b || !out || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, 3 * Stage::pad(n * 32) + Stage::pad(n));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t *d_a = sg.take(n * 32), *d_b = sg.take(n * 32), *d_o = sg.take(n * 32), *d_st = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_a, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_b, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
  FIELD_CALL(ctx, launch_test_point_add((int)ctx->suite, n, d_a, d_b, d_o, d_st, ctx->T, ctx->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_o, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_test_scalar_mul(vrfhip_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
                               uint8_t* status) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (ctx->bsw) return fail(VRFHIP_ERR_UNSUPPORTED, BSW_UNSUPPORTED_MSG);
  if (n == 0) return VRFHIP_SUCCESS;
  if (!scalars || !points || !out || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_workspace(ctx, n);
  if (rc) return rc;
  rc = ensure_stage(ctx, 3 * Stage::pad(n * 32) + Stage::pad(n));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t *d_k = sg.take(n * 32), *d_p = sg.take(n * 32), *d_o = sg.take(n * 32), *d_st = sg.take(n);
  HIP_TRY(hipMemcpyAsync(d_k, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_p, points, n * 32, hipMemcpyHostToDevice, ctx->stream));
  const size_t cap = ctx->ws_cap * (WS_TABS / 2);          // two window tables per item
  for (size_t base = 0; base < n; base += cap) {
    size_t m = std::min(cap, n - base);
    FIELD_CALL(ctx, launch_test_scalar_mul((int)ctx->suite, m, d_k + base * 32, d_p + base * 32, d_o + base * 32, d_st + base,
                           ctx->ws.tabs, ctx->T, ctx->stream));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_o, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(status, d_st, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}
}  // extern "C"

namespace {
int32_t test_hash_impl(vrfhip_ctx* ctx, size_t n, const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len,
                       uint8_t* out, int which) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (ctx->bsw) return fail(VRFHIP_ERR_UNSUPPORTED, BSW_UNSUPPORTED_MSG);
  if (n == 0) return VRFHIP_SUCCESS;
  if (!out || (!msg && (msg_len || msg_off))) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  const size_t ob = which ? 96 : 64;
  size_t msgb = blob_bytes(n, msg_off, msg_len, false);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, Stage::pad(msgb + 1) + Stage::pad((n + 1) * 4) + Stage::pad(n * ob));
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  uint8_t* d_msg = sg.take(msgb + 1);
  uint32_t* d_off = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  uint8_t* d_out = sg.take(n * ob);
  if (msgb) HIP_TRY(hipMemcpyAsync(d_msg, msg, msgb, hipMemcpyHostToDevice, ctx->stream));
  if (msg_off) HIP_TRY(hipMemcpyAsync(d_off, msg_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  FIELD_CALL(ctx, launch_test_hash((int)ctx->suite, n, make_view(d_msg, msg_off ? d_off : nullptr, msg_len, false), d_out, which, ctx->T,
                   ctx->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, d_out, n * ob, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

// ---- one call, several devices: contiguous slices of the batch, one host thread per context ----
// fn(ctx, lo, hi) handles items [lo, hi); the first failing slice decides the return value.
template <class F>
int32_t run_sharded(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, F fn) {
  if (!ctxs || n_ctx <= 0) return fail(VRFHIP_ERR_BAD_ARG, "no contexts");
  for (int32_t g = 0; g < n_ctx; ++g)
    if (!ctxs[g]) return fail(VRFHIP_ERR_BAD_ARG, "NULL context");
  std::vector<int32_t> rcs(n_ctx, VRFHIP_SUCCESS);
  std::vector<std::string> errs(n_ctx);
  std::vector<std::thread> th;
  bool spawn_failed = false;
  try {                                          // nothing may unwind across the C ABI, and a joinable std::thread must
    th.reserve((size_t)n_ctx);                   // not be destroyed: a failed creation joins what has started
    for (int32_t g = 0; g < n_ctx; ++g) {
      const size_t lo = n * (size_t)g / (size_t)n_ctx, hi = n * (size_t)(g + 1) / (size_t)n_ctx;
      if (lo == hi) continue;
      th.emplace_back([&, g, lo, hi] {
        rcs[g] = fn(ctxs[g], lo, hi);
        if (rcs[g]) errs[g] = g_last_error;      // thread-local in the worker: carry it to the caller's thread
      });
    }
  } catch (...) {
    spawn_failed = true;
  }
  for (auto& t : th) t.join();
  if (spawn_failed) return fail(VRFHIP_ERR_HIP, "could not start a worker thread per context");
  for (int32_t g = 0; g < n_ctx; ++g)
    if (rcs[g]) return fail(rcs[g], "device slice " + std::to_string(g) + ": " + errs[g]);
  return VRFHIP_SUCCESS;
}
// a slice of a variable-length blob: offsets rebased to the slice's first byte
struct BlobSlice {
  const uint8_t* blob;
  const uint32_t* off;         // nullptr: fixed layout
  std::vector<uint32_t> local;
  BlobSlice(const uint8_t* b, const uint32_t* o, uint32_t len, bool shared, size_t lo, size_t hi) {
    if (o) {
      local.resize(hi - lo + 1);
      for (size_t i = lo; i <= hi; ++i) local[i - lo] = o[i] - o[lo];
      blob = b + o[lo];
      off = local.data();
    } else {
      blob = (b && !shared) ? b + lo * (size_t)len : b;
      off = nullptr;
    }
  }
};
const uint8_t* at32(const uint8_t* p, size_t i) { return p ? p + i * 32 : nullptr; }
uint8_t* at32(uint8_t* p, size_t i) { return p ? p + i * 32 : nullptr; }
uint8_t* at_w(uint8_t* p, size_t i, size_t w) { return p ? p + i * w : nullptr; }
// width of a compressed point on the wire for a set of contexts (32; 33 for secp256r1): all must agree
size_t point_bytes_of(vrfhip_ctx* const* ctxs, int32_t n_ctx) {
  if (!ctxs || n_ctx < 1 || !ctxs[0]) return 32;              // run_sharded reports the bad argument
  const size_t w = ctxs[0]->pt_bytes();
  for (int32_t g = 1; g < n_ctx; ++g)
    if (!ctxs[g] || ctxs[g]->pt_bytes() != w) return 0;
  return w;
}
// width of the provers' point outputs for a set of contexts: all must agree on VRFHIP_FLAG_PROVE_POINTS_AFFINE
size_t prove_point_bytes_of(vrfhip_ctx* const* ctxs, int32_t n_ctx) {
  if (!ctxs || n_ctx < 1 || !ctxs[0]) return 32;              // run_sharded reports the bad argument
  const size_t w = ctxs[0]->prove_point_bytes();
  for (int32_t g = 1; g < n_ctx; ++g)
    if (!ctxs[g] || ctxs[g]->prove_point_bytes() != w) return 0;
  return w;
}
}  // namespace

extern "C" {

int32_t vrfhip_test_sha512(vrfhip_ctx* ctx, size_t n, const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len,
                           uint8_t* out) {
  return test_hash_impl(ctx, n, msg, msg_off, msg_len, out, 0);
}
int32_t vrfhip_test_xmd(vrfhip_ctx* ctx, size_t n, const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len,
                        uint8_t* out) {
  return test_hash_impl(ctx, n, msg, msg_off, msg_len, out, 1);
}

int32_t vrfhip_test_batch_digest(vrfhip_ctx* ctx, size_t n, int32_t n_arr, const uint8_t* const* arrays,
                                 const uint32_t* widths, const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                 uint64_t index0, uint8_t root[32]) {
  if (!ctx) return fail(VRFHIP_ERR_BAD_ARG, "ctx is NULL");
  if (ctx->sw) return fail(VRFHIP_ERR_UNSUPPORTED, "not available for the secp256r1 suite (the IETF and Pedersen schemes per proof, hash-to-curve, output hash, keys and point validation are)");
  if (!root || !arrays || !widths || n == 0) return fail(VRFHIP_ERR_BAD_ARG, "NULL argument or empty batch");
  if (n_arr < 1 || n_arr > DIGEST_MAX_ARRAYS) return fail(VRFHIP_ERR_BAD_ARG, "1..8 arrays");
  if ((ad_len || ad_off) && !ad) return fail(VRFHIP_ERR_BAD_ARG, "ad is NULL");
  size_t need = Stage::pad(digest_ws_bytes(n)) + 512, adb = blob_bytes(n, ad_off, ad_len, true);
  for (int32_t j = 0; j < n_arr; ++j) {
    if (!arrays[j] || widths[j] == 0 || widths[j] % 4) return fail(VRFHIP_ERR_BAD_ARG, "array NULL or width not a multiple of 4");
    need += Stage::pad(n * (size_t)widths[j]);
  }
  need += Stage::pad(adb + 1) + Stage::pad((n + 1) * 4);
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  DeviceGuard guard(ctx->device);
  int32_t rc = ensure_stage(ctx, need);
  if (rc) return rc;
  Stage sg(ctx->d_stage);
  DigestSrc ds{};
  ds.n_arr = n_arr;
  for (int32_t j = 0; j < n_arr; ++j) {
    uint8_t* d = sg.take(n * (size_t)widths[j]);
    HIP_TRY(hipMemcpyAsync(d, arrays[j], n * (size_t)widths[j], hipMemcpyHostToDevice, ctx->stream));
    ds.p[j] = d;
    ds.w[j] = widths[j];
  }
  uint8_t* d_ad = sg.take(adb + 1);
  uint32_t* d_off = reinterpret_cast<uint32_t*>(sg.take((n + 1) * 4));
  if (adb) HIP_TRY(hipMemcpyAsync(d_ad, ad, adb, hipMemcpyHostToDevice, ctx->stream));
  if (ad_off) HIP_TRY(hipMemcpyAsync(d_off, ad_off, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (ad) ds.ad = make_view(d_ad, ad_off ? d_off : nullptr, ad_len, true);
  uint8_t* d_ws = sg.take(digest_ws_bytes(n));
  uint8_t* d_root = sg.take(32);
  launch_batch_digest(ds, n, index0, d_ws, d_root, ctx->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(root, d_root, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return VRFHIP_SUCCESS;
}

int32_t vrfhip_ietf_verify_batch_multi(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, const uint8_t* pk,
                                       const uint8_t* input, const uint8_t* output, const uint8_t* c, const uint8_t* s,
                                       const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len, uint8_t* status) {
  if (n == 0) return VRFHIP_SUCCESS;
  if (!pk || !input || !output || !c || !s || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  const size_t pw = point_bytes_of(ctxs, n_ctx);
  if (!pw) return fail(VRFHIP_ERR_BAD_ARG, "contexts disagree on the suite's point encoding");
  return run_sharded(ctxs, n_ctx, n, [&](vrfhip_ctx* ctx, size_t lo, size_t hi) {
    BlobSlice a(ad, ad_off, ad_len, true, lo, hi);
    return vrfhip_ietf_verify_batch(ctx, hi - lo, pk + lo * pw, input + lo * pw, output + lo * pw, at32(c, lo), at32(s, lo),
                                    a.blob, a.off, ad_len, status + lo);
  });
}

int32_t vrfhip_ietf_prove_batch_multi(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, const uint8_t* sk,
                                      const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len, const uint8_t* input,
                                      const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len, uint8_t* output,
                                      uint8_t* c, uint8_t* s, uint8_t* pk_out, uint8_t* input_out, uint8_t* status) {
  if (n == 0) return VRFHIP_SUCCESS;
  if (!sk || !output || !c || !s) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  const size_t w = prove_point_bytes_of(ctxs, n_ctx), pw = point_bytes_of(ctxs, n_ctx);
  if (!w || !pw) return fail(VRFHIP_ERR_BAD_ARG, "contexts disagree on VRFHIP_FLAG_PROVE_POINTS_AFFINE or on the suite's point encoding");
  return run_sharded(ctxs, n_ctx, n, [&](vrfhip_ctx* ctx, size_t lo, size_t hi) {
    BlobSlice m(msg, input ? nullptr : msg_off, msg_len, false, lo, hi), a(ad, ad_off, ad_len, true, lo, hi);
    return vrfhip_ietf_prove_batch(ctx, hi - lo, at32(sk, lo), m.blob, m.off, msg_len, input ? input + lo * pw : nullptr, a.blob, a.off,
                                   ad_len, at_w(output, lo, w), at32(c, lo), at32(s, lo), at_w(pk_out, lo, w),
                                   at_w(input_out, lo, pw), status ? status + lo : nullptr);
  });
}

int32_t vrfhip_pedersen_prove_batch_multi(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, const uint8_t* sk,
                                          const uint8_t* msg, const uint32_t* msg_off, uint32_t msg_len,
                                          const uint8_t* input, const uint8_t* ad, const uint32_t* ad_off, uint32_t ad_len,
                                          uint8_t* output, uint8_t* pk_com, uint8_t* r, uint8_t* ok, uint8_t* s,
                                          uint8_t* sb, uint8_t* blinding_out, uint8_t* input_out, uint8_t* status) {
  if (n == 0) return VRFHIP_SUCCESS;
  if (!sk || !output || !pk_com || !r || !ok || !s || !sb) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  const size_t w = prove_point_bytes_of(ctxs, n_ctx), pw = point_bytes_of(ctxs, n_ctx);
  if (!w || !pw) return fail(VRFHIP_ERR_BAD_ARG, "contexts disagree on VRFHIP_FLAG_PROVE_POINTS_AFFINE or on the suite's point encoding");
  return run_sharded(ctxs, n_ctx, n, [&](vrfhip_ctx* ctx, size_t lo, size_t hi) {
    BlobSlice m(msg, input ? nullptr : msg_off, msg_len, false, lo, hi), a(ad, ad_off, ad_len, true, lo, hi);
    return vrfhip_pedersen_prove_batch(ctx, hi - lo, at32(sk, lo), m.blob, m.off, msg_len, input ? input + lo * pw : nullptr, a.blob, a.off,
                                       ad_len, at_w(output, lo, w), at_w(pk_com, lo, w), at_w(r, lo, w), at_w(ok, lo, w), at32(s, lo),
                                       at32(sb, lo), at32(blinding_out, lo), at_w(input_out, lo, pw),
                                       status ? status + lo : nullptr);
  });
}

int32_t vrfhip_pedersen_verify_batch_multi(vrfhip_ctx* const* ctxs, int32_t n_ctx, size_t n, const uint8_t* input,
                                           const uint8_t* output, const uint8_t* pk_com, const uint8_t* r,
                                           const uint8_t* ok, const uint8_t* s, const uint8_t* sb, const uint8_t* ad,
                                           const uint32_t* ad_off, uint32_t ad_len, const uint8_t* rlc_seed,
                                           uint8_t* status) {
  if (n == 0) return VRFHIP_SUCCESS;
  if (!input || !output || !pk_com || !r || !ok || !s || !sb || !status) return fail(VRFHIP_ERR_BAD_ARG, "NULL array");
  const size_t pw = point_bytes_of(ctxs, n_ctx);
  if (!pw) return fail(VRFHIP_ERR_BAD_ARG, "contexts disagree on the suite's point encoding");
  return run_sharded(ctxs, n_ctx, n, [&](vrfhip_ctx* ctx, size_t lo, size_t hi) {
    BlobSlice a(ad, ad_off, ad_len, true, lo, hi);
    if (rlc_seed)      // one multi-scalar multiplication per device slice; per-proof fallback inside the slice
      return vrfhip_pedersen_verify_batch_rlc(ctx, hi - lo, input + lo * pw, output + lo * pw, pk_com + lo * pw, r + lo * pw,
                                              ok + lo * pw, at32(s, lo), at32(sb, lo), a.blob, a.off, ad_len, rlc_seed,
                                              status + lo, nullptr);
    return vrfhip_pedersen_verify_batch(ctx, hi - lo, input + lo * pw, output + lo * pw, pk_com + lo * pw, r + lo * pw,
                                        ok + lo * pw, at32(s, lo), at32(sb, lo), a.blob, a.off, ad_len, status + lo);
  });
}

}  // extern "C"
