// gpsx_api.hip -- the C ABI of libgpsx.so (include/gpsx.h): context, device buffers, argument checking, launches.
// Host code only; every computation named in the header happens in the kernels of k_*.hip.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <sched.h>
#include <new>
#include <string>
#include <vector>

#include "gpsx_ctx.hpp"

using namespace gpsx;
using namespace gpsx_host;

namespace {
void track_graph_release(gpsx_ctx *ctx);
}

namespace {

int ensure_grid_tables(gpsx_ctx *ctx, const uint8_t *prns, int n_prn)
{
  if ((int)ctx->grid_prns.size() == n_prn && std::equal(prns, prns + n_prn, ctx->grid_prns.begin()))
    return GPSX_OK;
  const int slots = (n_prn + kAcqGroup - 1) / kAcqGroup * kAcqGroup;
  if (slots > ctx->grid_slots) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // free + null first: a failed hipMalloc below must not leave dangling members for the next call / gpsx_destroy
    void **members[] = {(void **)&ctx->d_grid_prns, (void **)&ctx->d_grid_chips, (void **)&ctx->d_grid_bits,
                        (void **)&ctx->d_grid_cw, (void **)&ctx->d_grid_cw8, (void **)&ctx->d_grid_mx_a,
                        (void **)&ctx->d_grid_mx_t};
    for (void **m : members) {
      if (*m)
        (void)hipFree(*m);
      *m = nullptr;
    }
    ctx->grid_slots = 0;
    ctx->grid_prns.clear();
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_grid_prns, slots));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_grid_chips, (size_t)slots * 1024));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_grid_bits, (size_t)slots * 32 * 4));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_grid_cw, (size_t)slots * kCodeWords * 4));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_grid_cw8, (size_t)slots * (kCodeWords / 2) * 4));
    const size_t sets = (size_t)(slots + 31) / 32;
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_grid_mx_a, sets * 4096 * 4));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_grid_mx_t, sets * 1032 * 4));
    ctx->grid_slots = slots;
  }
  std::vector<uint8_t> padded(slots, 0);
  std::copy(prns, prns + n_prn, padded.begin());
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_grid_prns, padded.data(), slots, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // `padded` goes out of scope
  launch_build_codes(ctx->stream, ctx->d_grid_prns, slots, kAcqGroup, ctx->d_grid_chips, ctx->d_grid_bits,
                     ctx->d_grid_cw, ctx->d_grid_cw8);
  LAUNCHCHK(ctx, "k_build_codes");
  launch_build_mx_tables(ctx->stream, ctx->d_grid_bits, slots, ctx->d_grid_mx_a, ctx->d_grid_mx_t);
  LAUNCHCHK(ctx, "k_build_mx_tables");
  ctx->grid_prns.assign(prns, prns + n_prn);
  return GPSX_OK;
}

int check_grid(gpsx_ctx *ctx, const gpsx_acq_grid_t *g, int n_blocks)
{
  if (!g || !g->prns)
    return fail(ctx, GPSX_EINVAL, "null descriptor");
  if (g->n_search < 1 || g->n_ms < 1 || g->n_ms > kMaxMs || g->n_prn < 1 || g->n_prn > 255 || g->n_dopp < 1)
    return fail(ctx, GPSX_EINVAL, "n_search/n_ms/n_prn/n_dopp out of range (n_ms <= 128)");
  if (g->phase_mode != GPSX_PHASES_BYTE && g->phase_mode != GPSX_PHASES_FINE)
    return fail(ctx, GPSX_EINVAL, "phase_mode must be 2046 or 16368");
  if (g->win_start < 0 || g->win_stop > GPSX_PHASES_BYTE || g->win_start > g->win_stop)
    return fail(ctx, GPSX_EINVAL, "window must satisfy 0 <= start <= stop <= 2046");
  if (g->shard_count < 0 || (g->shard_count > 0 && (g->shard_index < 0 || g->shard_index >= g->shard_count)))
    return fail(ctx, GPSX_EINVAL, "bad shard_index/shard_count");
  if (g->search_stride_blocks < 0)
    return fail(ctx, GPSX_EINVAL, "negative search stride");
  const long last = (long)(g->n_search - 1) * g->search_stride_blocks + g->n_ms;
  if (last > n_blocks)
    return fail(ctx, GPSX_EINVAL, "descriptor reads past the supplied IF blocks");
  for (int i = 0; i < g->n_prn; i++)
    if (g->prns[i] < 1 || g->prns[i] > GPSX_MAX_PRN)
      return fail(ctx, GPSX_EINVAL, "prn must be 1..210");
  if (g->n_dopp > 1 && g->dopp_step_hz <= 0)
    return fail(ctx, GPSX_EINVAL, "dopp_step_hz must be positive when n_dopp > 1");
  // the kernels index units, keys and peaks with 32-bit integers
  const long long n_keys = (long long)g->n_search * g->n_prn * g->n_dopp;
  if (n_keys * 8 > 0x7FFFFFFFLL)
    return fail(ctx, GPSX_EINVAL, "n_search * n_prn * n_dopp too large for one call (split the searches)");
  const long f_lo = (long)ctx->if_hz + g->dopp_min_hz;
  const long f_hi = f_lo + (long)(g->n_dopp - 1) * g->dopp_step_hz;
  if (f_lo <= 0 || f_hi <= 0 || f_lo >= 16368000 || f_hi >= 16368000)
    return fail(ctx, GPSX_EINVAL, "carrier frequency outside (0, fs)");
  return GPSX_OK;
}

}  // namespace

extern "C" {

int gpsx_version(void) { return GPSX_VERSION; }

int gpsx_abi_check(int header_version, size_t sizeof_loop_state, size_t sizeof_acq_grid, size_t sizeof_peak)
{
  // same minor series and the same record sizes: what a host strides device arrays and descriptor tables by
  if (header_version / 10 != GPSX_VERSION / 10)
    return GPSX_EINVAL;
  if (sizeof_loop_state != sizeof(gpsx_loop_state_t) || sizeof_acq_grid != sizeof(gpsx_acq_grid_t) || sizeof_peak != sizeof(gpsx_peak_t))
    return GPSX_EINVAL;
  return GPSX_OK;
}

const char *gpsx_strerror(int code)
{
  switch (code) {
    case GPSX_OK: return "ok";
    case GPSX_EIO: return "HIP runtime failure";
    case GPSX_ENOMEM: return "out of device memory";
    case GPSX_ENODEV: return "no usable gfx950 device";
    case GPSX_EINVAL: return "invalid argument";
    default: return "unknown gpsx error";
  }
}

const char *gpsx_last_error(const gpsx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

const char *gpsx_last_kernel(const gpsx_ctx *ctx) { return ctx ? ctx->last_kernel : ""; }

int gpsx_create(gpsx_ctx **out, int device, void *stream)
{
  if (!out)
    return GPSX_EINVAL;
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
    (void)hipGetLastError();
    return GPSX_ENODEV;
  }
  gpsx_ctx *ctx = new (std::nothrow) gpsx_ctx();
  if (!ctx)
    return GPSX_ENOMEM;
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) {
    delete ctx;
    return GPSX_ENODEV;
  }
  if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
    std::fprintf(stderr, "libgpsx: device %d is %s; this library carries gfx950 (MI355X) code only\n", device,
                 ctx->prop.gcnArchName);
    delete ctx;
    return GPSX_ENODEV;
  }
#ifdef GPSX_LAB
  // lib/libgpsx_lab.so only (csrc/Makefile; the product library is built without GPSX_LAB and reads none of these): forced
  // kernel forms for the parity tests of the alternative kernels and for A/B measurements
  if (const char *sg = std::getenv("GPSX_ACQ_SEG")) {
    const int v = std::atoi(sg);
    ctx->seg_force = (v == 4 || v == 8 || v == 16) ? v : 0;
  }
  ctx->no_split = std::getenv("GPSX_ACQ_NO_SPLIT") != nullptr;
  if (const char *sp = std::getenv("GPSX_ACQ_SPLIT")) {
    const int v = std::atoi(sp);
    ctx->split_force = v == 8 ? 8 : v == 4 ? 4 : 2;
  }
  if (const char *w = std::getenv("GPSX_TRACK_WAVE_FROM"))
    ctx->track_wave_from = std::atoi(w) > 0 ? std::atoi(w) : 1;
  if (const char *m = std::getenv("GPSX_ACQ_MS_MODE"))
    ctx->ms_mode = std::strcmp(m, "walk") == 0 ? 1 : (std::strcmp(m, "blocks") == 0 ? 2 : 0);
  if (const char *a = std::getenv("GPSX_ACQ_ALGO")) {
    ctx->algo = std::strcmp(a, "sad") == 0 ? kAlgoSad
                : std::strcmp(a, "dot8") == 0 ? kAlgoDot8
                : std::strcmp(a, "poly") == 0 ? kAlgoPoly
                                              : kAlgoMx;
  }
  if (ctx->seg_force && ctx->algo == kAlgoMx)
    ctx->algo = kAlgoPoly;   // $GPSX_ACQ_SEG names a form of the polyphase kernel: it selects that kernel too
#endif
  if (stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return GPSX_EIO;
    }
    ctx->own_stream = true;
  }
  // K1 for every PRN once: slot == prn
  const int slots = GPSX_MAX_PRN + 1;
  uint8_t *d_prns = nullptr;
  std::vector<uint8_t> prns(slots);
  for (int i = 0; i < slots; i++)
    prns[i] = (uint8_t)i;
  bool ok = hipMalloc((void **)&d_prns, slots) == hipSuccess &&
            hipMalloc((void **)&ctx->d_chips_all, (size_t)slots * 1024) == hipSuccess &&
            hipMalloc((void **)&ctx->d_bits_all, (size_t)slots * 32 * 4) == hipSuccess &&
            hipMalloc((void **)&ctx->d_cw_all, (size_t)slots * kCodeWords * 4) == hipSuccess &&
            hipMalloc((void **)&ctx->d_cw8_all, (size_t)slots * (kCodeWords / 2) * 4) == hipSuccess &&
            hipMalloc((void **)&ctx->d_trk_rep, (size_t)slots * kTrackRepStride * 4) == hipSuccess &&
            hipMemcpyAsync(d_prns, prns.data(), slots, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
            hipHostMalloc((void **)&ctx->h_bad_prn, 2 * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void **)&ctx->d_bad_prn, ctx->h_bad_prn, 0) == hipSuccess;
  if (ok) {
    ctx->h_bad_prn[0] = ctx->h_bad_prn[1] = 0;
    launch_build_codes(ctx->stream, d_prns, slots, 1, ctx->d_chips_all, ctx->d_bits_all, ctx->d_cw_all, ctx->d_cw8_all);
    launch_build_track_rep(ctx->stream, ctx->d_bits_all, slots, ctx->d_trk_rep);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
  }
  if (d_prns)
    (void)hipFree(d_prns);
  if (!ok) {
    std::fprintf(stderr, "libgpsx: device initialisation failed: %s\n", hipGetErrorString(hipGetLastError()));
    gpsx_destroy(ctx);
    return GPSX_EIO;
  }
  *out = ctx;
  return GPSX_OK;
}

void gpsx_destroy(gpsx_ctx *ctx)
{
  if (!ctx)
    return;
  (void)hipSetDevice(ctx->device);
  while (!ctx->captures.empty())
    gpsx_capture_destroy(ctx->captures.back());
  track_graph_release(ctx);
  if (ctx->stream)
    (void)hipStreamSynchronize(ctx->stream);
  void *bufs[] = {ctx->d_chips_all, ctx->d_bits_all, ctx->d_cw_all, ctx->d_cw8_all, ctx->d_trk_rep, ctx->d_grid_prns, ctx->d_grid_chips,
                  ctx->d_grid_bits, ctx->d_grid_cw, ctx->d_grid_cw8, ctx->d_grid_mx_a, ctx->d_grid_mx_t, ctx->d_arena, ctx->d_acc, ctx->d_energy};
  for (void *p : bufs)
    if (p)
      (void)hipFree(p);
  if (ctx->aux_stream) {
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->out_stream) {
      (void)hipStreamSynchronize(ctx->out_stream);
      (void)hipStreamDestroy(ctx->out_stream);
    }
    (void)hipEventDestroy(ctx->aux_event);
    for (hipEvent_t e : ctx->chunk_events)
      if (e)
        (void)hipEventDestroy(e);
  }
  if (ctx->h_bad_prn)
    (void)hipHostFree(ctx->h_bad_prn);
  if (ctx->d_loop_n_events)
    (void)hipFree(ctx->d_loop_n_events);
  if (ctx->d_loop_cand)
    (void)hipFree(ctx->d_loop_cand);
  if (ctx->d_weighted_prns) (void)hipFree(ctx->d_weighted_prns);
  if (ctx->d_loop_reseeds) (void)hipFree(ctx->d_loop_reseeds);
  if (ctx->d_loop_events) (void)hipFree(ctx->d_loop_events);
  if (ctx->d_loop_chmap) (void)hipFree(ctx->d_loop_chmap);
  if (ctx->own_stream && ctx->stream)
    (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int gpsx_synchronize(gpsx_ctx *ctx)
{
  if (int rc = use_device(ctx)) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->h_bad_prn && ctx->h_bad_prn[1]) {   // a gpsx_track_epl_batch_dev since the last look (its own flag: the
    ctx->h_bad_prn[1] = 0;                     // host-pointer step calls in between neither see nor clear it)
    return fail(ctx, GPSX_EINVAL, "a tracking batch held a prn outside 1..210 (correlated against the empty code)");
  }
  return GPSX_OK;
}

int gpsx_device_info(const gpsx_ctx *ctx, char *name, size_t name_len, int *compute_units, int *clock_khz)
{
  if (!ctx)
    return GPSX_EINVAL;
  if (name && name_len) {
    std::snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  }
  if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
  if (clock_khz) *clock_khz = ctx->prop.clockRate;
  return GPSX_OK;
}

int gpsx_set_if_format(gpsx_ctx *ctx, int if_format)
{
  if (!ctx)
    return GPSX_EINVAL;
  if (if_format != GPSX_IF_1BIT && if_format != GPSX_IF_2BIT_SM)
    return fail(ctx, GPSX_EINVAL, "unknown IF sample format");
  ctx->if_format = if_format;
  return GPSX_OK;
}

void gpsx_config_default(gpsx_config_t *cfg)
{
  if (cfg) {
    cfg->sample_rate_hz = 16368000u;
    cfg->if_hz = GPSX_IF_HZ;
  }
}

int gpsx_set_config(gpsx_ctx *ctx, const gpsx_config_t *cfg)
{
  if (!ctx || !cfg)
    return GPSX_EINVAL;
  if (cfg->sample_rate_hz != 16368000u)
    return fail(ctx, GPSX_EINVAL, "sample_rate_hz: this build carries the 16.368 MHz geometry only (2046 bytes per ms)");
  if (cfg->if_hz <= 0 || cfg->if_hz >= 16368000 / 2)
    return fail(ctx, GPSX_EINVAL, "if_hz out of range (0, sample_rate_hz / 2)");
  if (cfg->if_hz != ctx->if_hz) {
    if (use_device(ctx) == GPSX_OK) {
      (void)hipStreamSynchronize(ctx->stream);
      track_graph_release(ctx);   // the captured tracking steps carry the old value as a kernel argument
    }
    ctx->if_hz = cfg->if_hz;
  }
  return GPSX_OK;
}

int gpsx_get_config(const gpsx_ctx *ctx, gpsx_config_t *cfg)
{
  if (!ctx || !cfg)
    return GPSX_EINVAL;
  cfg->sample_rate_hz = 16368000u;
  cfg->if_hz = ctx->if_hz;
  return GPSX_OK;
}

int gpsx_if_unpack2(gpsx_ctx *ctx, const uint8_t *if_2bit, int n_blocks, uint8_t *sign_plane, uint8_t *magnitude_plane)
{
  if (int rc = use_device(ctx)) return rc;
  if (!if_2bit || n_blocks < 1 || (!sign_plane && !magnitude_plane))
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  const size_t in_bytes = (size_t)n_blocks * GPSX_BYTES_PER_MS_2BIT, out_bytes = (size_t)n_blocks * GPSX_BYTES_PER_MS;
  if (int rc = arena_reset(ctx, arena_size(in_bytes) + 2 * arena_size(out_bytes))) return rc;
  uint8_t *d_in = arena_take<uint8_t>(ctx, in_bytes);
  uint8_t *d_sign = arena_take<uint8_t>(ctx, out_bytes);
  uint8_t *d_mag = arena_take<uint8_t>(ctx, out_bytes);
  HIPCHK(ctx, hipMemcpyAsync(d_in, if_2bit, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  launch_unpack2(ctx->stream, d_in, n_blocks, sign_plane ? d_sign : nullptr, magnitude_plane ? d_mag : nullptr);
  LAUNCHCHK(ctx, "k_unpack2");
  if (sign_plane)
    HIPCHK(ctx, hipMemcpyAsync(sign_plane, d_sign, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  if (magnitude_plane)
    HIPCHK(ctx, hipMemcpyAsync(magnitude_plane, d_mag, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

int gpsx_malloc(gpsx_ctx *ctx, void **dptr, size_t bytes)
{
  if (int rc = use_device(ctx)) return rc;
  if (!dptr)
    return fail(ctx, GPSX_EINVAL, "null dptr");
  if (hipMalloc(dptr, bytes ? bytes : 1) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, GPSX_ENOMEM, "hipMalloc failed");
  }
  return GPSX_OK;
}

int gpsx_host_alloc(gpsx_ctx *ctx, void **hptr, size_t bytes)
{
  if (int rc = use_device(ctx)) return rc;
  if (!hptr)
    return fail(ctx, GPSX_EINVAL, "null hptr");
  if (hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, GPSX_ENOMEM, "hipHostMalloc failed");
  }
  return GPSX_OK;
}

int gpsx_bind_thread_to_device(gpsx_ctx *ctx)
{
  if (int rc = use_device(ctx)) return rc;
  char bdf[32] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, ctx->device) != hipSuccess) {
    (void)hipGetLastError();
    return GPSX_ENODEV;
  }
  for (char *c = bdf; *c; c++)
    *c = (char)std::tolower((unsigned char)*c);
  char path[128];
  std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  std::FILE *f = std::fopen(path, "r");
  if (!f)
    return GPSX_ENODEV;
  char list[1024] = {0};
  const bool got = std::fgets(list, (int)sizeof list, f) != nullptr;
  std::fclose(f);
  if (!got)
    return GPSX_ENODEV;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n_set = 0;
  for (const char *c = list; *c;) {        // "0-63,128-191"
    char *end = nullptr;
    const long lo = std::strtol(c, &end, 10);
    if (end == c)
      break;
    long hi = lo;
    if (*end == '-')
      hi = std::strtol(end + 1, &end, 10);
    for (long k = lo; k <= hi && k < CPU_SETSIZE; k++, n_set++)
      CPU_SET((int)k, &set);
    c = (*end == ',') ? end + 1 : end;
    if (*end != ',')
      break;
  }
  if (n_set == 0 || sched_setaffinity(0, sizeof set, &set) != 0)
    return GPSX_ENODEV;
  return GPSX_OK;
}

int gpsx_host_free(gpsx_ctx *ctx, void *hptr)
{
  if (int rc = use_device(ctx)) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipHostFree(hptr));
  return GPSX_OK;
}

int gpsx_free(gpsx_ctx *ctx, void *dptr)
{
  if (int rc = use_device(ctx)) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipFree(dptr));
  return GPSX_OK;
}

int gpsx_memcpy_h2d(gpsx_ctx *ctx, void *dst, const void *src, size_t bytes)
{
  if (int rc = use_device(ctx)) return rc;
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

int gpsx_memcpy_d2h(gpsx_ctx *ctx, void *dst, const void *src, size_t bytes)
{
  if (int rc = use_device(ctx)) return rc;
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

int gpsx_event_create(gpsx_ctx *ctx, void **event)
{
  if (int rc = use_device(ctx)) return rc;
  if (!event)
    return fail(ctx, GPSX_EINVAL, "null event pointer");
  hipEvent_t e;
  HIPCHK(ctx, hipEventCreate(&e));
  *event = e;
  return GPSX_OK;
}

int gpsx_event_record(gpsx_ctx *ctx, void *event)
{
  if (int rc = use_device(ctx)) return rc;
  HIPCHK(ctx, hipEventRecord(reinterpret_cast<hipEvent_t>(event), ctx->stream));
  return GPSX_OK;
}

int gpsx_event_elapsed_ms(gpsx_ctx *ctx, void *start, void *stop, float *ms)
{
  if (int rc = use_device(ctx)) return rc;
  HIPCHK(ctx, hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
  HIPCHK(ctx, hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
  return GPSX_OK;
}

int gpsx_event_destroy(gpsx_ctx *ctx, void *event)
{
  if (int rc = use_device(ctx)) return rc;
  HIPCHK(ctx, hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
  return GPSX_OK;
}

/* ---- K1 ----------------------------------------------------------------------------------------------------- */

int gpsx_ca_codes(gpsx_ctx *ctx, const uint8_t *prns, int n_prn, uint8_t *chips_out)
{
  if (int rc = use_device(ctx)) return rc;
  if (!prns || !chips_out || n_prn < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  for (int i = 0; i < n_prn; i++) {
    if (prns[i] < 1 || prns[i] > GPSX_MAX_PRN)
      return fail(ctx, GPSX_EINVAL, "prn must be 1..210");
    HIPCHK(ctx, hipMemcpyAsync(chips_out + (size_t)i * 1023, ctx->d_chips_all + (size_t)prns[i] * 1024, 1023,
                               hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

/* ---- acquisition ---------------------------------------------------------------------------------------------- */

int gpsx_acq_bits(int phase_mode) { return phase_mode == GPSX_PHASES_FINE ? 8 : 1; }

size_t gpsx_acq_keys_count(const gpsx_acq_grid_t *g)
{
  return g ? (size_t)g->n_search * g->n_prn * g->n_dopp : 0;
}

size_t gpsx_acq_peaks_count(const gpsx_acq_grid_t *g)
{
  return g ? gpsx_acq_keys_count(g) * gpsx_acq_bits(g->phase_mode) : 0;
}

namespace {

int ensure_acc(gpsx_ctx *ctx, size_t n_peaks)
{
  if (n_peaks <= ctx->acc_entries)
    return GPSX_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->d_acc)
    (void)hipFree(ctx->d_acc);
  ctx->d_acc = nullptr;
  ctx->acc_entries = 0;
  HIPCHK(ctx, hipMalloc((void **)&ctx->d_acc, 2 * n_peaks * sizeof(uint32_t)));
  // all-zero from here on: the kernels accumulate into it with atomics, k_acq_finalize* zeroes every entry it converts
  HIPCHK(ctx, hipMemsetAsync(ctx->d_acc, 0, 2 * n_peaks * sizeof(uint32_t), ctx->stream));
  ctx->acc_entries = n_peaks;
  return GPSX_OK;
}

}  // namespace

int gpsx_acq_grid_dev(gpsx_ctx *ctx, const gpsx_acq_grid_t *g, const void *d_if_blocks, int n_blocks,
                      gpsx_peak_t *d_peaks, int64_t *d_keys, gpsx_peak_t *d_per_ms, uint32_t *d_energy, uint16_t *d_cnt)
{
  if (int rc = use_device(ctx)) return rc;
  if (int rc = check_grid(ctx, g, n_blocks)) return rc;
  if (!d_if_blocks || !d_peaks)
    return fail(ctx, GPSX_EINVAL, "null device pointer");
  if (int rc = ensure_grid_tables(ctx, g->prns, g->n_prn)) return rc;

  const int shard_count = g->shard_count > 0 ? g->shard_count : 1;
  const int shard_index = g->shard_count > 0 ? g->shard_index : 0;
  const int n_bits = gpsx_acq_bits(g->phase_mode);
  const int n_groups = (g->n_prn + kAcqGroup - 1) / kAcqGroup;
  const long n_units = (long)g->n_search * n_groups * g->n_dopp;
  const long unit_lo = n_units * shard_index / shard_count, unit_hi = n_units * (shard_index + 1) / shard_count;
  const long local_units = unit_hi - unit_lo;
  if (local_units * n_bits * kSuperGroups > 0x7FFFFFFFL)
    return fail(ctx, GPSX_EINVAL, "grid too large for one launch");

  if (shard_count > 1) {
    // entries owned by other shards must read as zero
    HIPCHK(ctx, hipMemsetAsync(d_peaks, 0, gpsx_acq_peaks_count(g) * sizeof(gpsx_peak_t), ctx->stream));
  }
  AcqParams prm{};
  prm.split_segs = ctx->split_force;   // 0: launch_acq_mx picks by launch size
  prm.n_ms = g->n_ms;
  prm.search_stride_blocks = g->search_stride_blocks;
  prm.n_prn = g->n_prn;
  prm.n_groups = n_groups;
  prm.n_dopp = g->n_dopp;
  prm.dopp_min_hz = g->dopp_min_hz;
  prm.dopp_step_hz = g->dopp_step_hz;
  prm.n_bits = n_bits;
  prm.unit_lo = (int32_t)unit_lo;
  prm.unit_hi = (int32_t)unit_hi;
  prm.win_start = g->win_start;
  prm.win_stop = g->win_stop;
  prm.if_format = ctx->if_format;
  prm.if_hz = ctx->if_hz;
#if defined(GPSX_MX_ABLATIONS) && !defined(GPSX_LAB)
#error "GPSX_MX_ABLATIONS builds wrong-result timing variants: with -DGPSX_LAB only (tools/build_variant.sh), never into lib/libgpsx.so"
#endif
#ifdef GPSX_MX_ABLATIONS   // timing ablations of k_acq_mx (results are then wrong): tools/build_variant.sh -DGPSX_MX_ABLATIONS only
  {
    static const char *ex = std::getenv("GPSX_MX_EXPERIMENT");
    prm.experiment = ex ? std::atoi(ex) : 0;
  }
#endif
  prm.jobs = nullptr;
  prm.peaks = d_peaks;
  prm.per_ms = d_per_ms;
  prm.energy = d_energy;
  prm.cnt = d_cnt;
  const bool inspect = d_per_ms || d_energy || d_cnt;
  const bool fine = (ctx->algo == kAlgoPoly || ctx->algo == kAlgoMx) && n_bits == 8 && !inspect;
  // The matrix-core kernel (one 512-thread workgroup per (search, Doppler, 32 PRNs), one per CU) is the faster one at
  // every launch size measured, a single capture included (0.155 ms against 0.166 ms, profiles/r02_launch_size_sweep.json).
  // (and serves the byte-phase grid as sample offsets 0 and 8 of the fine one: ten of its seventeen passes, two epilogues)
  bool mx = ctx->algo == kAlgoMx && !inspect && (n_bits == 8 || g->n_ms == 1);
  if (mx) {
    const long clusters = acq_mx_clusters(prm);
    // n_ms > 1, few searches: a workgroup per (cluster, block) instead of a workgroup walking its cluster's blocks -- a lone
    // ten-block search is 210 workgroups (one round of the chip) instead of 21 doing ten blocks each
    bool mx_blocks = false;
    if (g->n_ms > 1) {
      mx_blocks = ctx->ms_mode ? ctx->ms_mode == 2
                               : clusters < ctx->prop.multiProcessorCount &&
                                     acq_poly_vals_bytes(g->n_search, g->n_ms, g->n_prn, g->n_dopp) <= ((size_t)8 << 30);
      const size_t need = mx_blocks ? acq_poly_vals_bytes(g->n_search, g->n_ms, g->n_prn, g->n_dopp) : acq_mx_energy_bytes(clusters);
      if (need > ctx->energy_bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_energy)
          (void)hipFree(ctx->d_energy);
        ctx->d_energy = nullptr;
        ctx->energy_bytes = 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 2 &&
            hipMalloc((void **)&ctx->d_energy, need) == hipSuccess)
          ctx->energy_bytes = need;
        else {
          (void)hipGetLastError();
          mx = false;
        }
      }
    }
    if (mx) {
      uint32_t *d_planes = nullptr;
      // the split form's result planes: launches of at most half a round of the chip, and launches of whole rounds plus a
      // last one that fills at most half of it (launch_acq_mx hands that tail to the split form)
      const long cus = ctx->prop.multiProcessorCount, tail = clusters % cus;
      if (g->n_ms == 1 && n_bits == 8 && shard_count == 1 && !ctx->no_split &&
          (2 * clusters <= cus || (clusters > cus && tail > 0 && 2 * tail <= cus))) {
        if (ensure_acc(ctx, gpsx_acq_peaks_count(g)) == GPSX_OK)
          d_planes = ctx->d_acc;
        else
          (void)hipGetLastError();   // (no memory for the planes: the unsplit form runs; the error must not stick to its launch)
      }
      bool keys_done = false;
      prm.keys = shard_count == 1 ? d_keys : nullptr;   // (a shard's foreign units must read as zero: k_acq_keys sees to that)
      ctx->last_kernel = launch_acq_mx(ctx->stream, prm, static_cast<const uint8_t *>(d_if_blocks), ctx->d_grid_mx_a,
                                       ctx->d_grid_mx_t, d_peaks, ctx->d_energy, mx_blocks, gpsx_acq_peaks_count(g), d_planes,
                                       ctx->prop.multiProcessorCount, &keys_done);
      if (d_planes && hipPeekAtLastError() != hipSuccess)
        ctx->acc_entries = 0;   // a launch between the split and finalize kernels failed: the planes may hold partial sums --
                                // the next use allocates and zeroes them afresh (ensure_acc)
      LAUNCHCHK(ctx, "k_acq_mx");
      if (d_keys && !keys_done) {
        launch_acq_keys(ctx->stream, d_peaks, d_keys, g->n_search, g->n_prn, n_groups, g->n_dopp, n_bits, (int)unit_lo,
                        (int)unit_hi);
        LAUNCHCHK(ctx, "k_acq_keys");
      }
      return GPSX_OK;
    }
  }
  bool poly = fine;
  bool block_parallel = false;
  if (poly && g->n_ms > 1) {
    // Scratch in HBM between blocks.  Many searches: each workgroup walks the blocks of its unit and keeps 64 KB of
    // running sums per (PRN, Doppler) pair of this shard (2.7 GB for 64 simultaneous cold-start searches).  Fewer
    // searches (fewer workgroups than six rounds of the chip's 768 slots): a workgroup per (unit, block) instead, all
    // blocks' magnitudes as u16 (32 KB per pair and block), summed and searched by a second small kernel -- a single
    // 10-block cold-start search then takes 0.84 ms instead of 2.8, and the form stays ahead up to ~40 searches.
    // When the scratch cannot be had, the register-resident dot8 kernel does the job.
    block_parallel = ctx->ms_mode ? ctx->ms_mode == 2
                                  : local_units * kSuperGroups < 6 * 768 &&
                                        acq_poly_vals_bytes(g->n_search, g->n_ms, g->n_prn, g->n_dopp) <= ((size_t)8 << 30);
    const size_t need = block_parallel ? acq_poly_vals_bytes(g->n_search, g->n_ms, g->n_prn, g->n_dopp)
                                       : acq_poly_energy_bytes(local_units);
    if (need > ctx->energy_bytes) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_energy)
        (void)hipFree(ctx->d_energy);
      ctx->d_energy = nullptr;
      ctx->energy_bytes = 0;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 2 &&
          hipMalloc((void **)&ctx->d_energy, need) == hipSuccess)
        ctx->energy_bytes = need;
      else {
        (void)hipGetLastError();
        poly = false;
      }
    }
  }
  if (poly) {
    const size_t n_peaks = gpsx_acq_peaks_count(g);
    if (int rc = ensure_acc(ctx, n_peaks))   // (zeroed when allocated, kept zero by k_acq_finalize)
      return rc;
    ctx->last_kernel = launch_acq_poly(ctx->stream, local_units, prm, static_cast<const uint8_t *>(d_if_blocks), ctx->d_grid_cw8,
                    ctx->d_grid_bits, ctx->d_acc, ctx->d_acc + n_peaks, n_peaks, d_peaks, shard_count > 1,
                    ctx->d_energy, block_parallel, ctx->seg_force);
    if (hipPeekAtLastError() != hipSuccess)
      ctx->acc_entries = 0;     // (as above: stale partial sums must not reach the next call)
    LAUNCHCHK(ctx, "k_acq_poly");
  } else {
    const int algo = ctx->algo == kAlgoSad ? kAlgoSad : kAlgoDot8;
    launch_acq(ctx->stream, kAcqGroup, algo, local_units, prm, static_cast<const uint8_t *>(d_if_blocks),
               algo == kAlgoDot8 ? ctx->d_grid_cw8 : ctx->d_grid_cw, ctx->d_grid_bits);
    LAUNCHCHK(ctx, "k_acq");
    ctx->last_kernel = algo == kAlgoSad ? (g->n_ms > 1 ? "k_acq<8,true,sad>" : "k_acq<8,false,sad>")
                                        : (g->n_ms > 1 ? "k_acq<8,true,dot8>" : "k_acq<8,false,dot8>");
  }
  if (d_keys) {
    launch_acq_keys(ctx->stream, d_peaks, d_keys, g->n_search, g->n_prn, n_groups, g->n_dopp, n_bits, (int)unit_lo,
                    (int)unit_hi);
    LAUNCHCHK(ctx, "k_acq_keys");
  }
  return GPSX_OK;
}

int gpsx_acq_grid_async(gpsx_ctx *ctx, const gpsx_acq_grid_t *g, const uint8_t *if_blocks, int n_blocks, gpsx_peak_t *peaks,
                        int64_t *keys)
{
  if (int rc = use_device(ctx)) return rc;
  if (int rc = check_grid(ctx, g, n_blocks)) return rc;
  if (!if_blocks || !peaks)
    return fail(ctx, GPSX_EINVAL, "null host pointer");
  const size_t n_peaks = gpsx_acq_peaks_count(g), n_keys = gpsx_acq_keys_count(g);
  const size_t if_bytes = (size_t)n_blocks * (ctx->if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : GPSX_BYTES_PER_MS);
  if (int rc = arena_reset(ctx, arena_size(if_bytes + 2) + arena_size(n_peaks * sizeof(gpsx_peak_t)) +
                                    arena_size(n_keys * sizeof(int64_t))))
    return rc;
  const uint8_t *d_if = capture_mirror(ctx, if_blocks, if_bytes);
  uint8_t *d_if_copy = arena_take<uint8_t>(ctx, if_bytes + 2);
  gpsx_peak_t *d_peaks = arena_take<gpsx_peak_t>(ctx, n_peaks);
  int64_t *d_keys = keys ? arena_take<int64_t>(ctx, n_keys) : nullptr;
  if (!d_if) {
    HIPCHK(ctx, hipMemcpyAsync(d_if_copy, if_blocks, if_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_if = d_if_copy;
  }
  if (int rc = gpsx_acq_grid_dev(ctx, g, d_if, n_blocks, d_peaks, d_keys, nullptr, nullptr, nullptr)) return rc;
  HIPCHK(ctx, hipMemcpyAsync(peaks, d_peaks, n_peaks * sizeof(gpsx_peak_t), hipMemcpyDeviceToHost, ctx->stream));
  if (keys)
    HIPCHK(ctx, hipMemcpyAsync(keys, d_keys, n_keys * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  return GPSX_OK;   // stream order protects the arena: the next call's copies and launches queue behind these
}

int gpsx_acq_grid(gpsx_ctx *ctx, const gpsx_acq_grid_t *g, const uint8_t *if_blocks, int n_blocks, gpsx_peak_t *peaks,
                  int64_t *keys)
{
  if (int rc = gpsx_acq_grid_async(ctx, g, if_blocks, n_blocks, peaks, keys)) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

int gpsx_acq_jobs(gpsx_ctx *ctx, const gpsx_acq_job_t *jobs, int n_jobs, const uint8_t *if_blocks, int n_blocks,
                  gpsx_peak_t *peaks, uint32_t *energy_opt)
{
  if (int rc = use_device(ctx)) return rc;
  if (!jobs || !if_blocks || !peaks || n_jobs < 1 || n_blocks < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  std::vector<AcqJobRec> recs(n_jobs);
  const int n_ms = jobs[0].n_ms;
  for (int i = 0; i < n_jobs; i++) {
    const gpsx_acq_job_t &j = jobs[i];
    if (j.n_ms != n_ms)
      return fail(ctx, GPSX_EINVAL, "all jobs of one call must share n_ms");
    if (j.n_ms < 1 || j.n_ms > kMaxMs || j.block < 0 || j.block + j.n_ms > n_blocks)
      return fail(ctx, GPSX_EINVAL, "job block range outside the supplied IF blocks");
    if (j.prn < 1 || j.prn > GPSX_MAX_PRN)
      return fail(ctx, GPSX_EINVAL, "prn must be 1..210");
    if (j.offset_bits < 0 || j.offset_bits > 7)
      return fail(ctx, GPSX_EINVAL, "offset_bits must be 0..7");
    if (j.win_start < 0 || j.win_stop > GPSX_PHASES_BYTE || j.win_start > j.win_stop)
      return fail(ctx, GPSX_EINVAL, "window must satisfy 0 <= start <= stop <= 2046");
    if (!(j.freq_hz > 0.0f && j.freq_hz < 16368000.0f))
      return fail(ctx, GPSX_EINVAL, "carrier frequency outside (0, fs)");
    recs[i] = AcqJobRec{j.block, j.prn, j.freq_hz, j.offset_bits, j.win_start, j.win_stop, i};
  }
  const size_t if_bytes = (size_t)n_blocks * (ctx->if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : GPSX_BYTES_PER_MS);
  const size_t e_count = energy_opt ? (size_t)n_jobs * GPSX_PHASES_BYTE : 0;
  if (int rc = arena_reset(ctx, arena_size(if_bytes + 2) + arena_size(n_jobs * sizeof(AcqJobRec)) +
                                    arena_size(n_jobs * sizeof(gpsx_peak_t)) + arena_size(e_count * 4)))
    return rc;
  const uint8_t *d_if = capture_mirror(ctx, if_blocks, if_bytes);
  uint8_t *d_if_copy = arena_take<uint8_t>(ctx, if_bytes + 2);
  AcqJobRec *d_jobs = arena_take<AcqJobRec>(ctx, n_jobs);
  gpsx_peak_t *d_peaks = arena_take<gpsx_peak_t>(ctx, n_jobs);
  uint32_t *d_energy = energy_opt ? arena_take<uint32_t>(ctx, e_count) : nullptr;
  if (!d_if) {
    HIPCHK(ctx, hipMemcpyAsync(d_if_copy, if_blocks, if_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_if = d_if_copy;
  }
  HIPCHK(ctx, hipMemcpyAsync(d_jobs, recs.data(), n_jobs * sizeof(AcqJobRec), hipMemcpyHostToDevice, ctx->stream));
  if (d_energy)
    HIPCHK(ctx, hipMemsetAsync(d_energy, 0, e_count * 4, ctx->stream));
  AcqParams prm{};
  prm.n_ms = n_ms;
  prm.n_bits = 1;
  prm.if_format = ctx->if_format;
  prm.if_hz = ctx->if_hz;
  prm.jobs = d_jobs;
  prm.peaks = d_peaks;
  prm.energy = d_energy;
  const int job_algo = ctx->algo == kAlgoSad ? kAlgoSad : kAlgoDot8;
  launch_acq(ctx->stream, 1, job_algo, n_jobs, prm, d_if, job_algo == kAlgoDot8 ? ctx->d_cw8_all : ctx->d_cw_all,
             ctx->d_bits_all);
  LAUNCHCHK(ctx, "k_acq(jobs)");
  HIPCHK(ctx, hipMemcpyAsync(peaks, d_peaks, n_jobs * sizeof(gpsx_peak_t), hipMemcpyDeviceToHost, ctx->stream));
  if (energy_opt)
    HIPCHK(ctx, hipMemcpyAsync(energy_opt, d_energy, e_count * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // also keeps `recs` alive until the copy has been consumed
  return GPSX_OK;
}

/* ---- tracking ------------------------------------------------------------------------------------------------- */

int gpsx_track_epl_batch_dev(gpsx_ctx *ctx, const void *d_if_block, gpsx_trk_state_t *d_st, int n_ch, int16_t *d_iq_out)
{
  if (int rc = use_device(ctx)) return rc;
  if (!d_if_block || !d_st || !d_iq_out || n_ch < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  launch_track_epl(ctx->stream, static_cast<const uint8_t *>(d_if_block), ctx->if_format, ctx->if_hz, d_st, n_ch,
                   ctx->d_chips_all, ctx->d_bits_all, ctx->d_trk_rep, d_iq_out, ctx->d_bad_prn + 1, ctx->track_wave_from);
  LAUNCHCHK(ctx, "k_track_epl");
  return GPSX_OK;
}

namespace {

constexpr int kTrackGraphMaxCh = 4096;    // measured (page-locked caller buffers): graph + staging copies 70 / 116 / 238 us for
                                          // 16384 / 32768 / 65536 channels, plain copies and a launch 81 / 111 / 170 us -- the graph
                                          // only pays where the step is launch-bound, and every new shape costs ~10 ms to
                                          // instantiate: a miss of ten deadlines in a closed loop whose channel count drifts

constexpr size_t kTrackGraphShapes = 12;  // every capacity there is (4, 8, .. 4096): a shape is instantiated once per context

void track_graph_free(gpsx_ctx::TrackGraph &t)
{
  if (t.exec) (void)hipGraphExecDestroy(t.exec);
  if (t.h_in) (void)hipHostFree(t.h_in);
  if (t.h_out) (void)hipHostFree(t.h_out);
  if (t.d_buf) (void)hipFree(t.d_buf);
  t = gpsx_ctx::TrackGraph{};
}

void track_graph_release(gpsx_ctx *ctx)
{
  for (gpsx_ctx::TrackGraph &t : ctx->trk_graphs)
    track_graph_free(t);
  ctx->trk_graphs.clear();
}

// Channel capacity a graph is captured for: the next power of two (4 at least), so that a receiver whose channels come and
// go between pre-tracking and tracking (gps_tracking_process_batch) keeps hitting the same few graphs -- eleven shapes in
// all, each instantiated once -- instead of re-instantiating one per count.  The padding channels carry kTrackPadPrn (the
// empty code) and are computed and copied like the others.
int track_graph_capacity(int n_ch)
{
  int p2 = 4;
  while (p2 < n_ch)
    p2 <<= 1;
  return p2;
}

// H2D(block + states) -> k_track_epl -> D2H(states + accumulators), captured once per (channel capacity, format) shape;
// returns the shape's graph (moved to the front of the small cache) or nullptr -> plain path
gpsx_ctx::TrackGraph *track_graph_prepare(gpsx_ctx *ctx, int n_ch_asked, size_t blk_bytes)
{
  if (ctx->trk_graph_unusable)
    return nullptr;
  const int n_ch = track_graph_capacity(n_ch_asked);
  std::vector<gpsx_ctx::TrackGraph> &cache = ctx->trk_graphs;
  for (size_t i = 0; i < cache.size(); i++)
    if (cache[i].n_ch == n_ch && cache[i].if_format == ctx->if_format) {
      if (i)
        std::rotate(cache.begin(), cache.begin() + i, cache.begin() + i + 1);
      return &cache[0];
    }
  (void)hipStreamSynchronize(ctx->stream);
  if (cache.size() >= kTrackGraphShapes) {
    track_graph_free(cache.back());
    cache.pop_back();
  }
  gpsx_ctx::TrackGraph t;
  t.blk_off = 0;
  t.st_off = (blk_bytes + 2 + 255) & ~(size_t)255;
  t.in_bytes = t.st_off + (size_t)n_ch * sizeof(gpsx_trk_state_t);
  t.out_bytes = (size_t)n_ch * (sizeof(gpsx_trk_state_t) + 12);
  bool ok = hipHostMalloc((void **)&t.h_in, t.in_bytes, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void **)&t.h_out, t.out_bytes, hipHostMallocDefault) == hipSuccess &&
            hipMalloc((void **)&t.d_buf, t.in_bytes + (size_t)n_ch * 12) == hipSuccess;
  hipGraph_t graph = nullptr;
  if (ok) {
    std::memset(t.h_in, 0, t.in_bytes);
    gpsx_trk_state_t *d_st = reinterpret_cast<gpsx_trk_state_t *>(t.d_buf + t.st_off);
    int16_t *d_iq = reinterpret_cast<int16_t *>(t.d_buf + t.in_bytes);   // right behind the states: one copy back
    ok = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      ok = hipMemcpyAsync(t.d_buf, t.h_in, t.in_bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
      launch_track_epl(ctx->stream, t.d_buf + t.blk_off, ctx->if_format, ctx->if_hz, d_st, n_ch, ctx->d_chips_all, ctx->d_bits_all,
                       ctx->d_trk_rep, d_iq, ctx->d_bad_prn, ctx->track_wave_from);
      ok = ok && hipGetLastError() == hipSuccess &&
           hipMemcpyAsync(t.h_out, d_st, t.out_bytes, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      ok = (hipStreamEndCapture(ctx->stream, &graph) == hipSuccess) && ok && graph;
    }
    ok = ok && hipGraphInstantiate(&t.exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph)
      (void)hipGraphDestroy(graph);
  }
  if (!ok) {
    (void)hipGetLastError();
    track_graph_free(t);
    ctx->trk_graph_unusable = true;   // this runtime / stream cannot capture: keep to the plain path
    return nullptr;
  }
  t.n_ch = n_ch;
  t.if_format = ctx->if_format;
  cache.insert(cache.begin(), t);
  return &cache[0];
}

}  // namespace

// what the kernels of the step just waited for said about its PRNs
static int track_prn_verdict(gpsx_ctx *ctx)
{
  if (*ctx->h_bad_prn == 0)
    return GPSX_OK;
  *ctx->h_bad_prn = 0;
  return fail(ctx, GPSX_EINVAL, "prn must be 1..210 (the channel was correlated against the empty code)");
}

// The side streams and events of the chunked tracking step, created on first use.  Failure-atomic: everything is made into
// locals and committed to the context only when all of it exists, so a half-built set is never seen by a later call.
static int track_pipeline_init(gpsx_ctx *ctx)
{
  constexpr int kMaxChunks = 16;
  if (ctx->aux_stream)
    return GPSX_OK;
  hipStream_t aux = nullptr, out = nullptr;
  hipEvent_t aux_event = nullptr;
  std::vector<hipEvent_t> events(3 * kMaxChunks, nullptr);   // per chunk: states arrived, correlators done, results out
  bool ok = hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&out, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&aux_event, hipEventDisableTiming) == hipSuccess;
  for (size_t i = 0; ok && i < events.size(); i++)
    ok = hipEventCreateWithFlags(&events[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    const hipError_t e = hipGetLastError();
    for (hipEvent_t ev : events)
      if (ev) (void)hipEventDestroy(ev);
    if (aux_event) (void)hipEventDestroy(aux_event);
    if (out) (void)hipStreamDestroy(out);
    if (aux) (void)hipStreamDestroy(aux);
    return fail(ctx, GPSX_EIO, std::string("tracking pipeline streams/events: ") + hipGetErrorString(e));
  }
  ctx->aux_stream = aux;
  ctx->out_stream = out;
  ctx->aux_event = aux_event;
  ctx->chunk_events = std::move(events);
  return GPSX_OK;
}

// The three-stage pipeline of a tracking step: chunk c's states go in on the context's stream, its correlators run on a
// second, its states and accumulators come out on a third, chained by events.  on_chunk (may be null): called on this thread
// as soon as a chunk's results are in the caller's arrays -- the GPU is then busy with the next ones.
static int track_pipeline_run(gpsx_ctx *ctx, const uint8_t *d_if, gpsx_trk_state_t *st, gpsx_trk_state_t *d_st, int n_ch,
                              int16_t *iq_out, int16_t *d_iq, int n_chunks, gpsx_track_chunk_fn on_chunk, void *user)
{
  if (int rc = track_pipeline_init(ctx)) return rc;
  const int per = ((n_ch + n_chunks - 1) / n_chunks + 3) & ~3;
  // (a failure inside the loops must not return while copies into the caller's arrays are still in flight on the side streams)
#define CHUNKCHK(call)                                                                           \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (void)hipStreamSynchronize(ctx->aux_stream);                                               \
      (void)hipStreamSynchronize(ctx->out_stream);                                               \
      return fail(ctx, GPSX_EIO, std::string(#call) + ": " + hipGetErrorString(e_));            \
    }                                                                                            \
  } while (0)
  int c = 0;
  for (int first = 0; first < n_ch; c++, first += per) {
    const int n = std::min(per, n_ch - first);
    hipEvent_t arrived = ctx->chunk_events[3 * c], done = ctx->chunk_events[3 * c + 1], out = ctx->chunk_events[3 * c + 2];
    CHUNKCHK(hipMemcpyAsync(d_st + first, st + first, n * sizeof(gpsx_trk_state_t), hipMemcpyHostToDevice, ctx->stream));
    CHUNKCHK(hipEventRecord(arrived, ctx->stream));      // (also: the block and everything before it on the stream)
    CHUNKCHK(hipStreamWaitEvent(ctx->aux_stream, arrived, 0));
    launch_track_epl(ctx->aux_stream, d_if, ctx->if_format, ctx->if_hz, d_st + first, n, ctx->d_chips_all, ctx->d_bits_all,
                     ctx->d_trk_rep, d_iq + (size_t)first * 6, ctx->d_bad_prn, ctx->track_wave_from);
    CHUNKCHK(hipGetLastError());
    CHUNKCHK(hipEventRecord(done, ctx->aux_stream));
    CHUNKCHK(hipStreamWaitEvent(ctx->out_stream, done, 0));
    CHUNKCHK(hipMemcpyAsync(st + first, d_st + first, n * sizeof(gpsx_trk_state_t), hipMemcpyDeviceToHost, ctx->out_stream));
    CHUNKCHK(hipMemcpyAsync(iq_out + (size_t)first * 6, d_iq + (size_t)first * 6, (size_t)n * 12, hipMemcpyDeviceToHost,
                            ctx->out_stream));
    if (on_chunk)
      CHUNKCHK(hipEventRecord(out, ctx->out_stream));
  }
  if (on_chunk) {
    c = 0;
    for (int first = 0; first < n_ch; c++, first += per) {
      // polled, not hipEventSynchronize: the pieces are tens of microseconds apart and the caller's loops are waiting
      // (closed loop at 131072 channels: 712 us per step polled, 774 us blocking)
      hipError_t q;
      while ((q = hipEventQuery(ctx->chunk_events[3 * c + 2])) == hipErrorNotReady)
        __builtin_ia32_pause();
      CHUNKCHK(q);
      ctx->in_chunk_callback = true;    // (the callback may not call into THIS context: use_device refuses)
      on_chunk(user, first, std::min(per, n_ch - first));
      ctx->in_chunk_callback = false;
    }
  }
#undef CHUNKCHK
  HIPCHK(ctx, hipStreamSynchronize(ctx->out_stream));
  // the context's stream stays the timeline of the context: what is enqueued on it next sees this step complete
  HIPCHK(ctx, hipEventRecord(ctx->aux_event, ctx->out_stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_event, 0));
  return track_prn_verdict(ctx);
}

int gpsx_track_epl_batch_chunked(gpsx_ctx *ctx, const uint8_t *if_block, gpsx_trk_state_t *st, int n_ch, int16_t *iq_out,
                                 int n_chunks, gpsx_track_chunk_fn on_chunk, void *user)
{
  if (int rc = use_device(ctx)) return rc;
  if (!if_block || !st || !iq_out || n_ch < 1 || !on_chunk)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  if (n_chunks < 1 || n_chunks > 16)
    return fail(ctx, GPSX_EINVAL, "n_chunks must be 1..16");
  ctx->h_bad_prn[0] = 0;
  const size_t blk_bytes = ctx->if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : GPSX_BYTES_PER_MS;
  if (int rc = arena_reset(ctx, arena_size(blk_bytes + 2) + arena_size(n_ch * sizeof(gpsx_trk_state_t)) +
                                    arena_size((size_t)n_ch * 12)))
    return rc;
  const uint8_t *d_if = capture_mirror(ctx, if_block, blk_bytes);
  uint8_t *d_if_copy = arena_take<uint8_t>(ctx, blk_bytes + 2);
  gpsx_trk_state_t *d_st = arena_take<gpsx_trk_state_t>(ctx, n_ch);
  int16_t *d_iq = arena_take<int16_t>(ctx, (size_t)n_ch * 6);
  if (!d_if) {
    HIPCHK(ctx, hipMemcpyAsync(d_if_copy, if_block, blk_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_if = d_if_copy;
  }
  return track_pipeline_run(ctx, d_if, st, d_st, n_ch, iq_out, d_iq, n_chunks, on_chunk, user);
}

int gpsx_track_epl_batch(gpsx_ctx *ctx, const uint8_t *if_block, gpsx_trk_state_t *st, int n_ch, int16_t *iq_out)
{
  if (int rc = use_device(ctx)) return rc;
  if (!if_block || !st || !iq_out || n_ch < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  // (the PRNs are checked by the kernel: a host loop over 400 000 states costs a sixth of the millisecond; flag 0 is this
  //  entry point's own -- every call waits for its kernels and reads it before it returns, so nothing can be pending in it;
  //  a gpsx_track_epl_batch_dev still in flight reports through flag 1 and gpsx_synchronize)
  ctx->h_bad_prn[0] = 0;
  const size_t blk_bytes = ctx->if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : GPSX_BYTES_PER_MS;
  // The real-time shape (a few channels to a few thousand, every millisecond): the whole step is ONE graph launch
  // between two small host copies into / out of pinned staging, instead of two copies in, a launch, two copies out.
  gpsx_ctx::TrackGraph *tg = n_ch <= kTrackGraphMaxCh ? track_graph_prepare(ctx, n_ch, blk_bytes) : nullptr;
  if (tg) {
    gpsx_ctx::TrackGraph &t = *tg;   // t.n_ch = the capacity the graph was captured for (>= n_ch)
    std::memcpy(t.h_in + t.blk_off, if_block, blk_bytes);
    std::memcpy(t.h_in + t.st_off, st, (size_t)n_ch * sizeof(gpsx_trk_state_t));
    gpsx_trk_state_t *pad = reinterpret_cast<gpsx_trk_state_t *>(t.h_in + t.st_off) + n_ch;
    for (int i = n_ch; i < t.n_ch; i++, pad++)   // padding channels: the empty code
      *pad = gpsx_trk_state_t{kTrackPadPrn, 0.0f, 0.0f, 0u};
    HIPCHK(ctx, hipGraphLaunch(t.exec, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(st, t.h_out, (size_t)n_ch * sizeof(gpsx_trk_state_t));
    std::memcpy(iq_out, t.h_out + (size_t)t.n_ch * sizeof(gpsx_trk_state_t), (size_t)n_ch * 12);
    return track_prn_verdict(ctx);
  }
  if (int rc = arena_reset(ctx, arena_size(blk_bytes + 2) + arena_size(n_ch * sizeof(gpsx_trk_state_t)) +
                                    arena_size((size_t)n_ch * 12)))
    return rc;
  const uint8_t *d_if = capture_mirror(ctx, if_block, blk_bytes);
  uint8_t *d_if_copy = arena_take<uint8_t>(ctx, blk_bytes + 2);
  gpsx_trk_state_t *d_st = arena_take<gpsx_trk_state_t>(ctx, n_ch);
  int16_t *d_iq = arena_take<int16_t>(ctx, (size_t)n_ch * 6);
  if (!d_if) {
    HIPCHK(ctx, hipMemcpyAsync(d_if_copy, if_block, blk_bytes, hipMemcpyHostToDevice, ctx->stream));
    d_if = d_if_copy;
  }
  // Very many channels: the step is the kernel plus PCIe (16 B of state in, 28 B of state + accumulators out per channel),
  // and the link is full duplex: the channels go through in chunks on a three-stage pipeline -- copy-in on the context's
  // stream, correlators on a second, copy-out on a third, chained by events -- so that chunk c + 1 arrives and chunk c - 1
  // leaves while chunk c is correlated, each copy at the link's full rate (page-locked caller buffers assumed:
  // gpsx_host_alloc; with pageable ones the copies serialise in the runtime and nothing is lost).
  constexpr int kChunkFrom = 131072, kMaxChunks = 16;
  // (1 / 2 / 3 / 4 / 6 / 8 chunks: 905 / 747 / 699 / 675 / 673 / 701 us for 458752 channels, 1260 / 1021 / 950 / 915 / 896 /
  //  905 us for 655360, 312 / 279 / 277 / 274 / 299 / 330 us for 131072; $GPSX_TRACK_CHUNKS overrides)
  static const int kChunksEnv = [] {
    const char *c = std::getenv("GPSX_TRACK_CHUNKS");
    const int v = c ? std::atoi(c) : 0;
    return v >= 1 && v <= kMaxChunks ? v : 0;
  }();
  const int kChunks = kChunksEnv ? kChunksEnv : (n_ch >= 393216 ? 6 : 4);
  if (n_ch >= kChunkFrom)
    return track_pipeline_run(ctx, d_if, st, d_st, n_ch, iq_out, d_iq, kChunks, nullptr, nullptr);
  HIPCHK(ctx, hipMemcpyAsync(d_st, st, n_ch * sizeof(gpsx_trk_state_t), hipMemcpyHostToDevice, ctx->stream));
  launch_track_epl(ctx->stream, d_if, ctx->if_format, ctx->if_hz, d_st, n_ch, ctx->d_chips_all, ctx->d_bits_all, ctx->d_trk_rep, d_iq,
                   ctx->d_bad_prn, ctx->track_wave_from);
  LAUNCHCHK(ctx, "k_track_epl");
  HIPCHK(ctx, hipMemcpyAsync(st, d_st, n_ch * sizeof(gpsx_trk_state_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(iq_out, d_iq, (size_t)n_ch * 12, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return track_prn_verdict(ctx);
}

/* ---- extension: weighted two-bit acquisition grid ------------------------------------------------------------------------ */

namespace {
int check_weighted(gpsx_ctx *ctx, const gpsx_acq_weighted_t *g, int n_blocks)
{
  if (!g || !g->prns)
    return fail(ctx, GPSX_EINVAL, "null descriptor");
  if (g->n_search < 1 || g->n_prn < 1 || g->n_dopp < 1 || g->search_stride_blocks < 0 || g->n_prn > 255)
    return fail(ctx, GPSX_EINVAL, "bad grid shape");
  if (g->weights != GPSX_WEIGHTS_SIGN_ONLY && g->weights != GPSX_WEIGHTS_SIGN_MAGNITUDE)
    return fail(ctx, GPSX_EINVAL, "unknown weights");
  if ((long)(g->n_search - 1) * g->search_stride_blocks + 1 > n_blocks)
    return fail(ctx, GPSX_EINVAL, "not enough blocks for the searches");
  for (int i = 0; i < g->n_prn; i++)
    if (g->prns[i] < 1 || g->prns[i] > GPSX_MAX_PRN)
      return fail(ctx, GPSX_EINVAL, "PRN outside 1..210");
  return GPSX_OK;
}
}  // namespace

int gpsx_acq_grid_weighted_dev(gpsx_ctx *ctx, const gpsx_acq_weighted_t *g, const void *d_if_blocks_2bit, int n_blocks,
                               gpsx_peak_t *d_peaks)
{
  if (int rc = use_device(ctx)) return rc;
  if (int rc = check_weighted(ctx, g, n_blocks)) return rc;
  if (!d_if_blocks_2bit || !d_peaks)
    return fail(ctx, GPSX_EINVAL, "null device pointer");
  if (ctx->algo == kAlgoMx) {
    // the matrix-core form (GPSX_ACQ_PATH_MATRIX, the default): chips from the sign-only grid's tables
    if (int rc = ensure_grid_tables(ctx, g->prns, g->n_prn)) return rc;
    launch_acq_mxw(ctx->stream, static_cast<const uint8_t *>(d_if_blocks_2bit), g->n_search, g->search_stride_blocks, g->n_prn,
                   ctx->d_grid_mx_a, ctx->if_hz, g->dopp_min_hz, g->dopp_step_hz, g->n_dopp,
                   g->weights == GPSX_WEIGHTS_SIGN_MAGNITUDE, d_peaks);
    LAUNCHCHK(ctx, "k_acq_mxw");
    ctx->last_kernel = "k_acq_mxw";
    return GPSX_OK;
  }
  // the vector-ALU form (GPSX_ACQ_PATH_VECTOR).  The PRN list behind the kernel: a few bytes through the arena's tail would
  // collide with a host-pointer caller's buffers -- its own small allocation, grow-only
  if (ctx->weighted_prns_cap < g->n_prn) {
    if (ctx->d_weighted_prns) (void)hipFree(ctx->d_weighted_prns);
    ctx->d_weighted_prns = nullptr;
    ctx->weighted_prns_cap = 0;
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_weighted_prns, 256));
    ctx->weighted_prns_cap = 255;
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_weighted_prns, g->prns, (size_t)g->n_prn, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // (the caller's PRN array is the caller's again)
  if (launch_acq_weighted(ctx->stream, static_cast<const uint8_t *>(d_if_blocks_2bit), g->n_search, g->search_stride_blocks, g->n_prn,
                          ctx->d_chips_all, ctx->d_weighted_prns, ctx->if_hz, g->dopp_min_hz, g->dopp_step_hz, g->n_dopp,
                          g->weights == GPSX_WEIGHTS_SIGN_MAGNITUDE, d_peaks))
    return fail(ctx, GPSX_EIO, "k_acq_weighted: the kernel's LDS size was refused");
  LAUNCHCHK(ctx, "k_acq_weighted");
  ctx->last_kernel = "k_acq_weighted";
  return GPSX_OK;
}

int gpsx_acq_grid_weighted(gpsx_ctx *ctx, const gpsx_acq_weighted_t *g, const uint8_t *if_blocks_2bit, int n_blocks, gpsx_peak_t *peaks)
{
  if (int rc = use_device(ctx)) return rc;
  if (int rc = check_weighted(ctx, g, n_blocks)) return rc;
  if (!if_blocks_2bit || !peaks)
    return fail(ctx, GPSX_EINVAL, "null host pointer");
  const size_t if_bytes = (size_t)n_blocks * GPSX_BYTES_PER_MS_2BIT, n_peaks = (size_t)g->n_search * g->n_prn * g->n_dopp;
  if (int rc = arena_reset(ctx, arena_size(if_bytes + 2) + arena_size(n_peaks * sizeof(gpsx_peak_t))))
    return rc;
  uint8_t *d_if = arena_take<uint8_t>(ctx, if_bytes + 2);
  gpsx_peak_t *d_peaks = arena_take<gpsx_peak_t>(ctx, n_peaks);
  HIPCHK(ctx, hipMemcpyAsync(d_if, if_blocks_2bit, if_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (int rc = gpsx_acq_grid_weighted_dev(ctx, g, d_if, n_blocks, d_peaks)) return rc;
  HIPCHK(ctx, hipMemcpyAsync(peaks, d_peaks, n_peaks * sizeof(gpsx_peak_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

/* ---- the tracking loops on the device ---------------------------------------------------------------------------------- */

int gpsx_loop_set_schedule(gpsx_ctx *ctx, int schedule)
{
  if (!ctx)
    return GPSX_EINVAL;
  if (schedule != GPSX_SCHED_EVERY_MS && schedule != GPSX_SCHED_MUX17)
    return fail(ctx, GPSX_EINVAL, "unknown serving schedule");
  ctx->loop_schedule = schedule;
  return GPSX_OK;
}

int gpsx_is_lab_build(void)
{
#ifdef GPSX_LAB
  return 1;
#else
  return 0;
#endif
}

int gpsx_set_acq_path(gpsx_ctx *ctx, int path)
{
  if (!ctx)
    return GPSX_EINVAL;
  if (path != GPSX_ACQ_PATH_MATRIX && path != GPSX_ACQ_PATH_VECTOR)
    return fail(ctx, GPSX_EINVAL, "unknown acquisition path");
  ctx->algo = path == GPSX_ACQ_PATH_VECTOR ? kAlgoPoly : kAlgoMx;
  return GPSX_OK;
}

int gpsx_loop_set_draws(gpsx_ctx *ctx, int draws)
{
  if (!ctx)
    return GPSX_EINVAL;
  if (draws != GPSX_DRAWS_XORSHIFT && draws != GPSX_DRAWS_LIBC)
    return fail(ctx, GPSX_EINVAL, "unknown source of false-lock draws");
  ctx->loop_draws = draws;
  return GPSX_OK;
}

namespace {

// One launch of the device loops.  GPSX_DRAWS_XORSHIFT: enqueue and return.  GPSX_DRAWS_LIBC (the reference's own
// draws, PM/GPS/tracking.c:309-326): first pass with an empty candidate table -- a channel whose false-lock detector fires
// reports (channel, millisecond, its carrier) and stands still, its state in HBM as it came into the registers (the launch's
// input, or under the multiplex the start of the slot it stops in -- ms_from); the host draws for the reported
// jumps in the order a single-threaded loop over the milliseconds and channels makes them (libc's rand(), the `do ... while`
// of the reference with its int16 arithmetic); second pass over the reported channels only, one per wave, each from its ms_from
// on, with the candidates in the table.  A channel jumps at most once in 81 four-millisecond groups, hence the
// launch-length limit; the passes are repeated should a replayed channel report again.  Waits for its kernels.
int run_track_loop(gpsx_ctx *ctx, const uint8_t *d_if, size_t blk_bytes, int n_blocks, gpsx_loop_state_t *d_state, int n_ch,
                   uint32_t first_tick, uint8_t *d_flags, gpsx_loop_trace_t *d_trace, uint32_t *d_bad_prn)
{
  const int word_sync = ctx->loop_word_sync == GPSX_WORDSYNC_DEVICE;
  if (ctx->loop_draws != GPSX_DRAWS_LIBC) {
    launch_track_loop(ctx->stream, d_if, (uint32_t)blk_bytes, n_blocks, ctx->if_format, ctx->if_hz, d_state, n_ch, first_tick,
                      ctx->loop_schedule, word_sync, ctx->d_bits_all, ctx->d_trk_rep, d_flags, d_trace, d_bad_prn, nullptr, 0, nullptr,
                      nullptr, nullptr);
    LAUNCHCHK(ctx, "k_track_loop");
    return GPSX_OK;
  }
  if (n_blocks > 320)
    return fail(ctx, GPSX_EINVAL, "GPSX_DRAWS_LIBC: at most 320 ms per launch (a channel can jump once per 324 ms)");
  if (ctx->loop_draws_capacity < n_ch) {
    if (ctx->d_loop_reseeds) (void)hipFree(ctx->d_loop_reseeds);
    if (ctx->d_loop_events) (void)hipFree(ctx->d_loop_events);
    if (ctx->d_loop_chmap) (void)hipFree(ctx->d_loop_chmap);
    if (ctx->d_loop_cand) (void)hipFree(ctx->d_loop_cand);
    ctx->d_loop_reseeds = nullptr; ctx->d_loop_events = nullptr; ctx->d_loop_chmap = nullptr; ctx->d_loop_cand = nullptr;
    ctx->loop_draws_capacity = 0;
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_loop_reseeds, (size_t)n_ch * sizeof(gpsx_loop_reseed_t)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_loop_events, (size_t)n_ch * sizeof(gpsx_loop_event_t)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_loop_chmap, (size_t)n_ch * sizeof(int)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_loop_cand, (size_t)n_ch * sizeof(gpsx_loop_reseed_t)));
    ctx->loop_draws_capacity = n_ch;
  }
  // the event counter lives in DEVICE memory (the kernel bumps it with an ordinary device atomic: nothing here depends on PCIe
  // atomics or on fine-grained coherence of host-mapped pages) and comes back with one small copy per pass
  if (!ctx->d_loop_n_events)
    HIPCHK(ctx, hipMalloc((void **)&ctx->d_loop_n_events, sizeof(uint32_t)));
  HIPCHK(ctx, hipMemsetAsync(ctx->d_loop_reseeds, 0xFF, (size_t)n_ch * sizeof(gpsx_loop_reseed_t), ctx->stream));   // ms = -1: none
  HIPCHK(ctx, hipMemsetAsync(ctx->d_loop_n_events, 0, sizeof(uint32_t), ctx->stream));
  launch_track_loop(ctx->stream, d_if, (uint32_t)blk_bytes, n_blocks, ctx->if_format, ctx->if_hz, d_state, n_ch, first_tick,
                    ctx->loop_schedule, word_sync, ctx->d_bits_all, ctx->d_trk_rep, d_flags, d_trace, d_bad_prn, nullptr, 0,
                    ctx->d_loop_reseeds, ctx->d_loop_events, ctx->d_loop_n_events);
  LAUNCHCHK(ctx, "k_track_loop");
  uint32_t n = 0;
  HIPCHK(ctx, hipMemcpyAsync(&n, ctx->d_loop_n_events, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<gpsx_loop_event_t> ev;
  std::vector<gpsx_loop_reseed_t> cand;
  std::vector<int> chans;
  std::vector<int> jumped;      // channels that already hold a candidate in this call, sorted
  for (int pass = 0; n; pass++) {
    // After GPSX_EIO the states, flags and traces of this call are UNDEFINED (a replay pass may have run on part of the
    // channels): the caller restores d_state from its own copy or drops the channels (include/gpsx.h).
    if (pass >= 4 || n > (uint32_t)n_ch)
      return fail(ctx, GPSX_EIO, "GPSX_DRAWS_LIBC: the false-lock replay does not settle (states of this call are undefined)");
    ev.resize(n);
    HIPCHK(ctx, hipMemcpyAsync(ev.data(), ctx->d_loop_events, n * sizeof(gpsx_loop_event_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::sort(ev.begin(), ev.end(), [](const gpsx_loop_event_t &a, const gpsx_loop_event_t &b) {
      return a.ms != b.ms ? a.ms < b.ms : a.channel < b.channel;
    });
    cand.resize(n);
    chans.resize(n);
    for (uint32_t i = 0; i < n; i++) {
      // ONE reseed slot per channel and launch: a channel's false-lock counter needs 81 four-millisecond groups to fill again,
      // and a launch is at most 320 ms (checked above) -- a second report of one channel inside a call breaks that invariant
      // and would overwrite the first candidate, so it is refused rather than replayed wrongly
      const auto at = std::lower_bound(jumped.begin(), jumped.end(), (int)ev[i].channel);
      if (at != jumped.end() && *at == (int)ev[i].channel)
        return fail(ctx, GPSX_EIO, "GPSX_DRAWS_LIBC: a channel reported a second false-lock jump inside one launch (states of this call are undefined)");
      jumped.insert(at, (int)ev[i].channel);
      int16_t delta, candidate;
      do {   // tracking.c:313-324, its types
        const uint16_t r = (uint16_t)(std::rand() % 500);   // ACQ_SEARCH_STEP_HZ
        candidate = (int16_t)(ev[i].found_freq_hz - r + 250);
        delta = (int16_t)((int16_t)ev[i].if_freq_i16 - candidate);
      } while (std::abs((int)delta) < 200);
      cand[i] = gpsx_loop_reseed_t{ev[i].ms, candidate, ev[i].ms_from};
      chans[i] = ev[i].channel;
    }
    // the pass's candidates and channel list in two uploads, scattered to the channels' slots on the device
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_loop_cand, cand.data(), n * sizeof(gpsx_loop_reseed_t), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_loop_chmap, chans.data(), n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    launch_loop_scatter_reseeds(ctx->stream, ctx->d_loop_reseeds, ctx->d_loop_chmap, ctx->d_loop_cand, (int)n);
    HIPCHK(ctx, hipMemsetAsync(ctx->d_loop_n_events, 0, sizeof(uint32_t), ctx->stream));
    launch_track_loop(ctx->stream, d_if, (uint32_t)blk_bytes, n_blocks, ctx->if_format, ctx->if_hz, d_state, n_ch, first_tick,
                      ctx->loop_schedule, word_sync, ctx->d_bits_all, ctx->d_trk_rep, d_flags, d_trace, d_bad_prn, ctx->d_loop_chmap,
                      (int)n, ctx->d_loop_reseeds, ctx->d_loop_events, ctx->d_loop_n_events);
    LAUNCHCHK(ctx, "k_track_loop");
    HIPCHK(ctx, hipMemcpyAsync(&n, ctx->d_loop_n_events, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // (cand / chans stay alive until here: the uploads above have completed)
  }
  return GPSX_OK;
}

}  // namespace

int gpsx_loop_set_word_sync(gpsx_ctx *ctx, int owner)
{
  if (!ctx)
    return GPSX_EINVAL;
  if (owner != GPSX_WORDSYNC_DEVICE && owner != GPSX_WORDSYNC_HOST)
    return fail(ctx, GPSX_EINVAL, "unknown word-sync owner");
  ctx->loop_word_sync = owner;
  return GPSX_OK;
}

int gpsx_track_loop_dev(gpsx_ctx *ctx, const void *d_if_blocks, int n_blocks, gpsx_loop_state_t *d_state, int n_ch,
                        uint32_t first_tick_ms, uint8_t *d_flags, gpsx_loop_trace_t *d_trace_opt)
{
  if (int rc = use_device(ctx)) return rc;
  if (!d_if_blocks || !d_state || !d_flags || n_ch < 1 || n_blocks < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  const size_t blk_bytes = ctx->if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : GPSX_BYTES_PER_MS;
  if (int rc = run_track_loop(ctx, static_cast<const uint8_t *>(d_if_blocks), blk_bytes, n_blocks, d_state, n_ch, first_tick_ms,
                              d_flags, d_trace_opt, ctx->d_bad_prn + 1))
    return rc;
  ctx->last_kernel = "k_track_loop";
  return GPSX_OK;
}

int gpsx_track_loop(gpsx_ctx *ctx, const uint8_t *if_blocks, int n_blocks, gpsx_loop_state_t *d_state, int n_ch,
                    uint32_t first_tick_ms, uint8_t *flags, gpsx_loop_trace_t *trace_opt)
{
  if (int rc = use_device(ctx)) return rc;
  if (!if_blocks || !d_state || !flags || n_ch < 1 || n_blocks < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  const size_t blk_bytes = ctx->if_format == GPSX_IF_2BIT_SM ? GPSX_BYTES_PER_MS_2BIT : GPSX_BYTES_PER_MS;
  const size_t n_rec = (size_t)n_blocks * n_ch;
  if (int rc = arena_reset(ctx, arena_size(blk_bytes * n_blocks + 2) + arena_size(n_rec) +
                                    (trace_opt ? arena_size(n_rec * sizeof(gpsx_loop_trace_t)) : 0)))
    return rc;
  const uint8_t *d_if = capture_mirror(ctx, if_blocks, blk_bytes * n_blocks);
  uint8_t *d_if_copy = arena_take<uint8_t>(ctx, blk_bytes * n_blocks + 2);
  uint8_t *d_flags = arena_take<uint8_t>(ctx, n_rec);
  gpsx_loop_trace_t *d_trace = trace_opt ? arena_take<gpsx_loop_trace_t>(ctx, n_rec) : nullptr;
  if (!d_if) {
    HIPCHK(ctx, hipMemcpyAsync(d_if_copy, if_blocks, blk_bytes * n_blocks, hipMemcpyHostToDevice, ctx->stream));
    d_if = d_if_copy;
  }
  ctx->h_bad_prn[0] = 0;   // (flag 0: this entry point waits for its kernel, as gpsx_track_epl_batch does)
  if (int rc = run_track_loop(ctx, d_if, blk_bytes, n_blocks, d_state, n_ch, first_tick_ms, d_flags, d_trace, ctx->d_bad_prn))
    return rc;
  ctx->last_kernel = "k_track_loop";
  HIPCHK(ctx, hipMemcpyAsync(flags, d_flags, n_rec, hipMemcpyDeviceToHost, ctx->stream));
  if (trace_opt)
    HIPCHK(ctx, hipMemcpyAsync(trace_opt, d_trace, n_rec * sizeof(gpsx_loop_trace_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return track_prn_verdict(ctx);
}

int gpsx_loop_set_polarity(gpsx_ctx *ctx, gpsx_loop_state_t *d_state, const int *channels, const uint8_t *values, int n)
{
  if (int rc = use_device(ctx)) return rc;
  if (!d_state || !channels || !values || n < 0)
    return fail(ctx, GPSX_EINVAL, "null argument");
  if (n == 0)
    return GPSX_OK;
  for (int i = 0; i < n; i++)
    if (channels[i] < 0)
      return fail(ctx, GPSX_EINVAL, "negative channel index");
  // one small kernel for all of them (half of a million channels can find their polarity within the same second)
  if (int rc = arena_reset(ctx, arena_size((size_t)n * sizeof(int)) + arena_size((size_t)n)))
    return rc;
  int *d_idx = arena_take<int>(ctx, (size_t)n);
  uint8_t *d_val = arena_take<uint8_t>(ctx, (size_t)n);
  HIPCHK(ctx, hipMemcpyAsync(d_idx, channels, (size_t)n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_val, values, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  launch_loop_set_polarity(ctx->stream, d_state, d_idx, d_val, n);
  LAUNCHCHK(ctx, "k_loop_set_polarity");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // (the caller's arrays are free again, and so is the arena)
  return GPSX_OK;
}

int gpsx_loop_reset_code_filter(gpsx_ctx *ctx, gpsx_loop_state_t *d_state, int n_ch)
{
  if (int rc = use_device(ctx)) return rc;
  if (!d_state || n_ch < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  static_assert(offsetof(gpsx_loop_state_t, code_phase_fine_filt) == offsetof(gpsx_loop_state_t, code_filt_cnt) + 2,
                "the two window fields are adjacent");
  // six bytes per state: a strided fill (rows of sizeof(gpsx_loop_state_t), 6 B wide)
  HIPCHK(ctx, hipMemset2DAsync(reinterpret_cast<uint8_t *>(d_state) + offsetof(gpsx_loop_state_t, code_filt_cnt), sizeof(gpsx_loop_state_t),
                               0, 6, (size_t)n_ch, ctx->stream));
  return GPSX_OK;
}

int gpsx_rewind(gpsx_ctx *ctx, gpsx_trk_state_t *st, int n_ch, const uint8_t *steps)
{
  if (int rc = use_device(ctx)) return rc;
  if (!st || !steps || n_ch < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  if (int rc = arena_reset(ctx, arena_size(n_ch * sizeof(gpsx_trk_state_t)) + arena_size(n_ch)))
    return rc;
  gpsx_trk_state_t *d_st = arena_take<gpsx_trk_state_t>(ctx, n_ch);
  uint8_t *d_steps = arena_take<uint8_t>(ctx, n_ch);
  HIPCHK(ctx, hipMemcpyAsync(d_st, st, n_ch * sizeof(gpsx_trk_state_t), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_steps, steps, n_ch, hipMemcpyHostToDevice, ctx->stream));
  launch_rewind(ctx->stream, ctx->if_hz, d_st, n_ch, d_steps);
  LAUNCHCHK(ctx, "k_rewind");
  HIPCHK(ctx, hipMemcpyAsync(st, d_st, n_ch * sizeof(gpsx_trk_state_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

/* ---- per-call primitives -------------------------------------------------------------------------------------- */

int gpsx_wipeoff(gpsx_ctx *ctx, const uint8_t *signal, float freq_hz, uint32_t *accum, uint8_t *data_i, uint8_t *data_q)
{
  if (int rc = use_device(ctx)) return rc;
  if (!signal || !accum || !data_i || !data_q)
    return fail(ctx, GPSX_EINVAL, "null argument");
  if (int rc = arena_reset(ctx, 3 * arena_size(2048) + arena_size(4))) return rc;
  uint8_t *d_sig = arena_take<uint8_t>(ctx, 2048);
  uint8_t *d_i = arena_take<uint8_t>(ctx, 2048);
  uint8_t *d_q = arena_take<uint8_t>(ctx, 2048);
  uint32_t *d_acc = arena_take<uint32_t>(ctx, 1);
  HIPCHK(ctx, hipMemcpyAsync(d_sig, signal, GPSX_BYTES_PER_MS, hipMemcpyHostToDevice, ctx->stream));
  launch_wipeoff(ctx->stream, d_sig, freq_hz, *accum, d_i, d_q, d_acc);
  LAUNCHCHK(ctx, "k_wipeoff");
  // bytes 2044, 2045 of the caller's buffers are left alone, as the reference leaves them
  HIPCHK(ctx, hipMemcpyAsync(data_i, d_i, 2044, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(data_q, d_q, 2044, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(accum, d_acc, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

int gpsx_replica(gpsx_ctx *ctx, const uint8_t *chips, unsigned offset_bits, uint16_t *out)
{
  if (int rc = use_device(ctx)) return rc;
  if (!chips || !out)
    return fail(ctx, GPSX_EINVAL, "null argument");
  if (int rc = arena_reset(ctx, arena_size(1024) + arena_size(2048))) return rc;
  uint8_t *d_chips = arena_take<uint8_t>(ctx, 1024);
  uint16_t *d_out = arena_take<uint16_t>(ctx, 1024);
  HIPCHK(ctx, hipMemcpyAsync(d_chips, chips, 1023, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_out + 1023, out + 1023, 2, hipMemcpyHostToDevice, ctx->stream));  // pad word is OR-ed
  launch_replica(ctx->stream, d_chips, offset_bits, d_out);
  LAUNCHCHK(ctx, "k_replica");
  HIPCHK(ctx, hipMemcpyAsync(out, d_out, 2048, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

int gpsx_corr_offsets(gpsx_ctx *ctx, const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q,
                      const uint16_t *offsets, int n, uint16_t *cnt_i, uint16_t *cnt_q, int16_t *corr8)
{
  if (int rc = use_device(ctx)) return rc;
  if (!replica || !data_i || !data_q || !offsets || n < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  for (int i = 0; i < n; i++)
    if (offsets[i] > GPSX_PHASES_BYTE)
      return fail(ctx, GPSX_EINVAL, "offset must be 0..2046");
  if (int rc = arena_reset(ctx, 3 * arena_size(2048) + 4 * arena_size((size_t)n * 2))) return rc;
  uint8_t *d_rep = arena_take<uint8_t>(ctx, 2048);
  uint8_t *d_i = arena_take<uint8_t>(ctx, 2048);
  uint8_t *d_q = arena_take<uint8_t>(ctx, 2048);
  uint16_t *d_off = arena_take<uint16_t>(ctx, n);
  uint16_t *d_ci = arena_take<uint16_t>(ctx, n);
  uint16_t *d_cq = arena_take<uint16_t>(ctx, n);
  int16_t *d_c8 = arena_take<int16_t>(ctx, n);
  HIPCHK(ctx, hipMemcpyAsync(d_rep, replica, GPSX_BYTES_PER_MS, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_i, data_i, GPSX_BYTES_PER_MS, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_q, data_q, GPSX_BYTES_PER_MS, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_off, offsets, (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream));
  launch_corr_offsets(ctx->stream, d_rep, d_i, d_q, d_off, 0, n, d_ci, d_cq, d_c8);
  LAUNCHCHK(ctx, "k_corr_offsets");
  if (cnt_i) HIPCHK(ctx, hipMemcpyAsync(cnt_i, d_ci, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
  if (cnt_q) HIPCHK(ctx, hipMemcpyAsync(cnt_q, d_cq, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
  if (corr8) HIPCHK(ctx, hipMemcpyAsync(corr8, d_c8, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

int gpsx_mag8(gpsx_ctx *ctx, const uint16_t *cnt_i, const uint16_t *cnt_q, int n, int16_t *out)
{
  if (int rc = use_device(ctx)) return rc;
  if (!cnt_i || !cnt_q || !out || n < 1)
    return fail(ctx, GPSX_EINVAL, "null/empty argument");
  if (int rc = arena_reset(ctx, 3 * arena_size((size_t)n * 2) + arena_size(4))) return rc;
  uint16_t *d_i = arena_take<uint16_t>(ctx, n);
  uint16_t *d_q = arena_take<uint16_t>(ctx, n);
  int16_t *d_o = arena_take<int16_t>(ctx, n);
  uint32_t *d_bad = arena_take<uint32_t>(ctx, 1);
  uint32_t bad = 0;
  HIPCHK(ctx, hipMemcpyAsync(d_i, cnt_i, (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_q, cnt_q, (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
  launch_mag8(ctx->stream, d_i, d_q, n, d_o, d_bad);
  LAUNCHCHK(ctx, "k_mag8");
  HIPCHK(ctx, hipMemcpyAsync(out, d_o, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (bad)  // the grid kernel's trimmed magnitude must be the same function
    return fail(ctx, GPSX_EIO, "internal: mag8_fast disagrees with mag8 on " + std::to_string(bad) + " inputs");
  return GPSX_OK;
}

int gpsx_corr_search(gpsx_ctx *ctx, const uint16_t *replica, const uint16_t *data_i, const uint16_t *data_q,
                     unsigned start_shift, unsigned stop_shift, gpsx_peak_t *peak)
{
  if (int rc = use_device(ctx)) return rc;
  if (!replica || !data_i || !data_q || !peak)
    return fail(ctx, GPSX_EINVAL, "null argument");
  if (stop_shift > (unsigned)GPSX_PHASES_BYTE + 1u)
    return fail(ctx, GPSX_EINVAL, "stop_shift must be <= 2047");
  const int n = stop_shift > start_shift ? (int)(stop_shift - start_shift) : 0;
  if (int rc = arena_reset(ctx, 3 * arena_size(2048) + arena_size((size_t)(n + 1) * 2) + arena_size(sizeof(gpsx_peak_t))))
    return rc;
  uint8_t *d_rep = arena_take<uint8_t>(ctx, 2048);
  uint8_t *d_i = arena_take<uint8_t>(ctx, 2048);
  uint8_t *d_q = arena_take<uint8_t>(ctx, 2048);
  int16_t *d_c8 = arena_take<int16_t>(ctx, n + 1);
  gpsx_peak_t *d_peak = arena_take<gpsx_peak_t>(ctx, 1);
  HIPCHK(ctx, hipMemcpyAsync(d_rep, replica, GPSX_BYTES_PER_MS, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_i, data_i, GPSX_BYTES_PER_MS, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_q, data_q, GPSX_BYTES_PER_MS, hipMemcpyHostToDevice, ctx->stream));
  launch_corr_offsets(ctx->stream, d_rep, d_i, d_q, nullptr, (int)start_shift, n, nullptr, nullptr, d_c8);
  LAUNCHCHK(ctx, "k_corr_offsets");
  launch_search_reduce(ctx->stream, d_c8, n, (int)start_shift, d_peak);
  LAUNCHCHK(ctx, "k_search_reduce");
  HIPCHK(ctx, hipMemcpyAsync(peak, d_peak, sizeof(gpsx_peak_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GPSX_OK;
}

}  // extern "C"
