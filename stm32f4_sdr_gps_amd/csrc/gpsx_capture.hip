// gpsx_capture.hip -- IF ingest: capture rings (include/gpsx.h), host code.
#include <cstdio>
#include <cstring>
#include <new>

#include "gpsx_ctx.hpp"

using namespace gpsx_host;

// A host pointer that lies in a committed, unmodified part of one of the context's capture rings has an HBM mirror:
// returns the mirror (and makes the compute stream wait for the ring's copies), or nullptr -> the caller copies.
const uint8_t *gpsx_host::capture_mirror(gpsx_ctx *ctx, const uint8_t *host, size_t bytes)
{
  for (gpsx_capture *cap : ctx->captures) {
    const size_t ring_bytes = (size_t)cap->n_slots * cap->block_bytes;
    if (host < cap->h_ring || host >= cap->h_ring + ring_bytes || bytes == 0 || bytes > ring_bytes)
      continue;
    const size_t off = (size_t)(host - cap->h_ring);
    if (off % cap->block_bytes || bytes % cap->block_bytes || off + bytes > ring_bytes)
      return nullptr;
    for (size_t slot = off / cap->block_bytes; slot < (off + bytes) / cap->block_bytes; slot++)
      if (!cap->mirrored[slot])
        return nullptr;
    return cap->d_ring + off;   // the copies were enqueued on this very stream: ordered before any reader
  }
  return nullptr;
}

/* ---- IF ingest: capture ring ---------------------------------------------------------------------------------- */

int gpsx_capture_create(gpsx_ctx *ctx, int n_slots, gpsx_capture **out)
{
  if (int rc = use_device(ctx)) return rc;
  if (!out || n_slots < 1 || n_slots > 4096)
    return fail(ctx, GPSX_EINVAL, "capture ring: 1 <= n_slots <= 4096");
  gpsx_capture *cap = new (std::nothrow) gpsx_capture;
  if (!cap)
    return fail(ctx, GPSX_ENOMEM, "out of host memory");
  cap->ctx = ctx;
  cap->n_slots = n_slots;
  cap->block_bytes = ctx->if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : GPSX_BYTES_PER_MS;
  cap->mirrored.assign(n_slots, 0);
  cap->sent.assign(n_slots, nullptr);
  const size_t ring_bytes = (size_t)n_slots * cap->block_bytes;
  bool ok = hipHostMalloc((void **)&cap->h_ring, ring_bytes, hipHostMallocDefault) == hipSuccess &&
            hipMalloc((void **)&cap->d_ring, ring_bytes + 2) == hipSuccess &&
            hipMalloc((void **)&cap->d_window, ring_bytes + 2) == hipSuccess &&
            hipMemsetAsync(cap->d_ring, 0, ring_bytes + 2, ctx->stream) == hipSuccess &&
            hipMemsetAsync(cap->d_window, 0, ring_bytes + 2, ctx->stream) == hipSuccess;
  for (int i = 0; ok && i < n_slots; i++)
    ok = hipEventCreateWithFlags(&cap->sent[i], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
  ctx->captures.push_back(cap);
  if (!ok) {
    const hipError_t e = hipGetLastError();
    gpsx_capture_destroy(cap);
    return fail(ctx, GPSX_ENOMEM, std::string("capture ring allocation: ") + hipGetErrorString(e));
  }
  std::memset(cap->h_ring, 0, ring_bytes);
  *out = cap;
  return GPSX_OK;
}

void gpsx_capture_destroy(gpsx_capture *cap)
{
  if (!cap)
    return;
  gpsx_ctx *ctx = cap->ctx;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream)
    (void)hipStreamSynchronize(ctx->stream);   // pending copies, and launches that read the mirror
  ctx->captures.erase(std::remove(ctx->captures.begin(), ctx->captures.end(), cap), ctx->captures.end());
  for (hipEvent_t e : cap->sent)
    if (e) (void)hipEventDestroy(e);
  if (cap->d_ring) (void)hipFree(cap->d_ring);
  if (cap->d_window) (void)hipFree(cap->d_window);
  if (cap->h_ring) (void)hipHostFree(cap->h_ring);
  delete cap;
}

uint8_t *gpsx_capture_write_slot(gpsx_capture *cap)
{
  if (!cap)
    return nullptr;
  if (cap->mirrored[cap->write_slot]) {
    // the slot's previous block may not have left the pinned buffer yet: the producer must not overwrite it
    (void)hipSetDevice(cap->ctx->device);
    (void)hipEventSynchronize(cap->sent[cap->write_slot]);
    cap->mirrored[cap->write_slot] = 0;
  }
  return cap->h_ring + (size_t)cap->write_slot * cap->block_bytes;
}

int gpsx_capture_commit(gpsx_capture *cap)
{
  if (!cap)
    return GPSX_EINVAL;
  gpsx_ctx *ctx = cap->ctx;
  if (int rc = use_device(ctx)) return rc;
  const int slot = cap->write_slot;
  // stream order does the rest: earlier launches that read this device slot finish first, later ones see the new block
  HIPCHK(ctx, hipMemcpyAsync(cap->d_ring + (size_t)slot * cap->block_bytes, cap->h_ring + (size_t)slot * cap->block_bytes,
                             cap->block_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipEventRecord(cap->sent[slot], ctx->stream));
  cap->mirrored[slot] = 1;
  cap->ready_slot = slot;
  cap->write_slot = (slot + 1) % cap->n_slots;
  cap->packet_cnt++;
  return GPSX_OK;
}

int gpsx_capture_push(gpsx_capture *cap, const uint8_t *block)
{
  if (!cap || !block)
    return GPSX_EINVAL;
  std::memcpy(gpsx_capture_write_slot(cap), block, cap->block_bytes);
  return gpsx_capture_commit(cap);
}

const uint8_t *gpsx_capture_ready_buf(const gpsx_capture *cap)
{
  return cap && cap->ready_slot >= 0 ? cap->h_ring + (size_t)cap->ready_slot * cap->block_bytes : nullptr;
}

int gpsx_capture_window_dev(gpsx_capture *cap, int n_blocks, const void **d_blocks)
{
  if (!cap || !d_blocks)
    return GPSX_EINVAL;
  gpsx_ctx *ctx = cap->ctx;
  if (int rc = use_device(ctx)) return rc;
  if (n_blocks < 1 || n_blocks > cap->n_slots || (uint32_t)n_blocks > cap->packet_cnt)
    return fail(ctx, GPSX_EINVAL, "capture window: more blocks than the ring holds / has received");
  const int first = (cap->ready_slot - (n_blocks - 1) + cap->n_slots) % cap->n_slots;   // oldest block of the window
  if (first + n_blocks <= cap->n_slots) {
    *d_blocks = cap->d_ring + (size_t)first * cap->block_bytes;
    return GPSX_OK;
  }
  // the window wraps: gather its two pieces (device to device, on the stream) into the window buffer
  const int head = cap->n_slots - first;
  HIPCHK(ctx, hipMemcpyAsync(cap->d_window, cap->d_ring + (size_t)first * cap->block_bytes, (size_t)head * cap->block_bytes,
                             hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(cap->d_window + (size_t)head * cap->block_bytes, cap->d_ring,
                             (size_t)(n_blocks - head) * cap->block_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  *d_blocks = cap->d_window;
  return GPSX_OK;
}

uint32_t gpsx_capture_packet_cnt(const gpsx_capture *cap) { return cap ? cap->packet_cnt : 0; }
size_t gpsx_capture_block_bytes(const gpsx_capture *cap) { return cap ? cap->block_bytes : 0; }

long gpsx_capture_replay_file(gpsx_capture *cap, const char *path, long first_block, long max_blocks,
                              gpsx_capture_block_fn on_block, void *user)
{
  if (!cap || !path || first_block < 0)
    return GPSX_EINVAL;
  gpsx_ctx *ctx = cap->ctx;
  std::FILE *f = std::fopen(path, "rb");
  if (!f)
    return fail(ctx, GPSX_EIO, std::string("cannot open ") + path);
  long done = 0;
  int rc = GPSX_OK;
  if (fseeko(f, (off_t)first_block * (off_t)cap->block_bytes, SEEK_SET) != 0)
    rc = fail(ctx, GPSX_EIO, "seek past the end of the IF file");
  while (rc == GPSX_OK && (max_blocks < 0 || done < max_blocks)) {
    uint8_t *slot = gpsx_capture_write_slot(cap);
    if (std::fread(slot, 1, cap->block_bytes, f) != cap->block_bytes)
      break;   // end of the recording (a trailing partial block is dropped, as a 1 ms DMA transfer would never complete)
    rc = gpsx_capture_commit(cap);
    if (rc != GPSX_OK)
      break;
    done++;
    if (on_block && on_block(user, cap, first_block + done - 1) != 0)
      break;
  }
  std::fclose(f);
  return rc == GPSX_OK ? done : rc;
}

