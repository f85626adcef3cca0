// gpsx_ctx.hpp -- internals shared by the host-side translation units of libgpsx.so (gpsx_api.hip, gpsx_capture.hip,
// gpsx_group.hip): the context record, the capture ring record, error plumbing, the scratch arena.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "gpsx_kernels.hpp"

// IF ingest ring (include/gpsx.h "capture ring"): pinned host slots + HBM mirror.  Blocks are 2 KB: their copies go on the
// context's own stream (measured: a separate copy stream costs two more events per block than it can ever win back),
// so readers enqueued later need no synchronisation object at all; one event per slot tells the producer when a slot's
// pinned bytes have left.
struct gpsx_capture {
  gpsx_ctx *ctx = nullptr;
  int n_slots = 0;
  size_t block_bytes = 0;
  uint8_t *h_ring = nullptr;           // [n_slots][block_bytes], hipHostMalloc
  uint8_t *d_ring = nullptr;           // [n_slots][block_bytes] + 2
  uint8_t *d_window = nullptr;         // [n_slots][block_bytes] + 2: windows that wrap are gathered here
  std::vector<hipEvent_t> sent;        // per slot: its host bytes have been read by the copy engine
  std::vector<uint8_t> mirrored;       // per slot: HBM mirror matches the host slot (cleared when handed out for writing)
  int write_slot = 0;
  int ready_slot = -1;
  uint32_t packet_cnt = 0;
};

struct gpsx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // the pipelined tracking step of very many channels: correlator and copy-out streams, events between the stages
  hipStream_t aux_stream = nullptr, out_stream = nullptr;
  hipEvent_t aux_event = nullptr;
  std::vector<hipEvent_t> chunk_events;
  uint32_t *h_bad_prn = nullptr;      // page-locked flag the tracking kernels raise on a PRN outside 1..210
  uint32_t *d_bad_prn = nullptr;      // its device address
  std::string err;
  const char *last_kernel = "";      // dominant kernel of the last acquisition / device-loop launch (gpsx_last_kernel)
  hipDeviceProp_t prop;

  // tables for every PRN, slot == prn (slot 0 is the empty code): K1 output
  uint8_t *d_chips_all = nullptr;    // [211][1024]
  uint32_t *d_bits_all = nullptr;    // [211][32]
  uint32_t *d_cw_all = nullptr;      // [211][256] (group of 1)
  uint32_t *d_cw8_all = nullptr;     // [211][128] (group of 1)
  uint32_t *d_trk_rep = nullptr;     // [211][kTrackRepStride]: replica bit streams for the tracking correlators
  int if_format = GPSX_IF_1BIT;
  int loop_schedule = GPSX_SCHED_EVERY_MS;   // gpsx_loop_set_schedule
  int loop_word_sync = GPSX_WORDSYNC_DEVICE; // gpsx_loop_set_word_sync
  uint8_t *d_weighted_prns = nullptr;        // gpsx_acq_grid_weighted's PRN list
  int weighted_prns_cap = 0;
  int loop_draws = GPSX_DRAWS_XORSHIFT;      // gpsx_loop_set_draws
  // GPSX_DRAWS_LIBC: per-channel candidate table, event list, channel list of the second pass (grow-only), event counter (mapped)
  gpsx::gpsx_loop_reseed_t *d_loop_reseeds = nullptr;
  gpsx::gpsx_loop_event_t *d_loop_events = nullptr;
  int *d_loop_chmap = nullptr;
  int loop_draws_capacity = 0;
  uint32_t *d_loop_n_events = nullptr;          // the kernel's event counter: DEVICE memory (copied back with the event list)
  gpsx::gpsx_loop_reseed_t *d_loop_cand = nullptr;    // the candidates of one replay pass, uploaded in one piece
  int if_hz = GPSX_IF_HZ;             // gpsx_config_t.if_hz
  int algo = gpsx::kAlgoMx;                // $GPSX_ACQ_ALGO = mx (default: the matrix-core kernel, at every launch size) | poly |
                                           // dot8 | sad, for A/B measurements and the parity tests of the alternative kernels
  uint32_t *d_acc = nullptr;         // poly: packed-key and sum planes merged across workgroups
  size_t acc_entries = 0;
  int seg_force = 0;                 // $GPSX_ACQ_SEG = 4 | 8 | 16: the polyphase kernel at that many offsets per workgroup (tests, A/B);
                                     // implies $GPSX_ACQ_ALGO=poly unless another algorithm was named
  int track_wave_from = 1;           // $GPSX_TRACK_WAVE_FROM: channels from which k_track_epl_wave serves the step (default: always;
                                     // a large value selects the workgroup-per-channel kernel: tests, A/B)
  bool in_chunk_callback = false;    // set around the on_chunk calls of gpsx_track_epl_batch_chunked: entry points refuse re-entry
  int split_force = 0;               // $GPSX_ACQ_SPLIT: workgroups per cluster of the split form (2, 4, 8; 0 = by launch size)
  bool no_split = false;             // $GPSX_ACQ_NO_SPLIT: small single-block fine grids stay one workgroup per cluster (tests, A/B)
  int ms_mode = 0;                   // $GPSX_ACQ_MS_MODE = walk | blocks: force one multi-block form (tests, A/B); 0 = by size
  uint32_t *d_energy = nullptr;      // poly, n_ms > 1: running per-hypothesis sums between blocks (grow-only)
  size_t energy_bytes = 0;
  // grouped tables for the PRN list of the last grid call
  std::vector<uint8_t> grid_prns;
  int grid_slots = 0;
  uint8_t *d_grid_prns = nullptr;
  uint8_t *d_grid_chips = nullptr;
  uint32_t *d_grid_bits = nullptr;
  uint32_t *d_grid_cw = nullptr;
  uint32_t *d_grid_cw8 = nullptr;
  uint32_t *d_grid_mx_a = nullptr;   // matrix-core kernel: A fragments per 32-slot cluster
  uint32_t *d_grid_mx_t = nullptr;   //                     transposed chip words

  // the per-millisecond tracking step as a captured graph (one launch instead of five runtime calls), for the last
  // (channel count, sample format) shapes it was called with; pinned staging on both sides
  struct TrackGraph {
    int n_ch = 0, if_format = -1;
    hipGraphExec_t exec = nullptr;
    uint8_t *h_in = nullptr, *h_out = nullptr, *d_buf = nullptr;
    size_t blk_off = 0, st_off = 0, in_bytes = 0, out_bytes = 0;
  };
  std::vector<TrackGraph> trk_graphs;   // a few shapes (per-channel calls and batch calls alternate), most recent first
  bool trk_graph_unusable = false;

  std::vector<gpsx_capture *> captures;   // IF ingest rings opened on this context

  // grow-only scratch arena for the host-pointer entry points
  char *d_arena = nullptr;
  size_t arena_bytes = 0;
  size_t arena_used = 0;
};

namespace gpsx_host {


inline int fail(gpsx_ctx *ctx, int code, const std::string &msg)
{
  if (ctx)
    ctx->err = msg;
  return code;
}

#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail((ctx), GPSX_EIO, std::string(#call) + ": " + hipGetErrorString(e_));          \
  } while (0)

#define LAUNCHCHK(ctx, what)                                                                     \
  do {                                                                                           \
    hipError_t e_ = hipGetLastError();                                                           \
    if (e_ != hipSuccess)                                                                        \
      return fail((ctx), GPSX_EIO, std::string(what) + " launch: " + hipGetErrorString(e_));     \
  } while (0)

inline int arena_reset(gpsx_ctx *ctx, size_t need)
{
  ctx->arena_used = 0;
  if (need <= ctx->arena_bytes)
    return GPSX_OK;
  if (ctx->d_arena) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(ctx->d_arena));
    ctx->d_arena = nullptr;
    ctx->arena_bytes = 0;
  }
  const size_t want = std::max(need, (size_t)1 << 20);
  if (hipMalloc((void **)&ctx->d_arena, want) != hipSuccess)
    return fail(ctx, GPSX_ENOMEM, "hipMalloc(arena) failed");
  ctx->arena_bytes = want;
  return GPSX_OK;
}

template <typename T>
inline T *arena_take(gpsx_ctx *ctx, size_t count)
{
  const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
  T *p = reinterpret_cast<T *>(ctx->d_arena + ctx->arena_used);
  ctx->arena_used += bytes;
  return p;
}

inline size_t arena_size(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

inline int use_device(gpsx_ctx *ctx)
{
  if (!ctx)
    return GPSX_EINVAL;
  if (ctx->in_chunk_callback) {
    // gpsx_track_epl_batch_chunked is still using the context's arena and side streams while its callback runs: an entry
    // point that resets the arena or grows a buffer would pull them from under the pieces in flight
    return fail(ctx, GPSX_EINVAL, "called from inside a gpsx_track_epl_batch_chunked callback (use another context there)");
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return GPSX_OK;
}

// A host pointer that lies in a committed, unmodified part of one of the context's capture rings has an HBM mirror:
// returns the mirror, or nullptr -> the caller copies (gpsx_capture.hip).
const uint8_t *capture_mirror(gpsx_ctx *ctx, const uint8_t *host, size_t bytes);

}  // namespace gpsx_host
