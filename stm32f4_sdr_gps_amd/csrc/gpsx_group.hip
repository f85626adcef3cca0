// gpsx_group.hip -- the sharded sweep inside one process: contexts joined by RCCL communicators (include/gpsx.h), host code.
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>   // declarations only: the library is dlopen'ed
#define GPSX_HAVE_RCCL_HEADERS 1
#else                    // a ROCm install without the RCCL development headers still builds the single-GPU library
#define GPSX_HAVE_RCCL_HEADERS 0
#endif

#include <new>

#include "gpsx_ctx.hpp"

using namespace gpsx_host;

/* ---- multi-GPU group: RCCL inside one process ------------------------------------------------------------------- */
#if !GPSX_HAVE_RCCL_HEADERS
// built without <rccl/rccl.h>: the group entry points exist and say so; everything single-GPU is unaffected
struct gpsx_group {};
int gpsx_group_create(gpsx_ctx *const *ctxs, int n, gpsx_group **)
{
  return ctxs && n > 0 && ctxs[0] ? fail(ctxs[0], GPSX_ENODEV, "group: this libgpsx was built without the RCCL headers") : GPSX_EINVAL;
}
void gpsx_group_destroy(gpsx_group *) {}
int gpsx_acq_grid_sharded(gpsx_group *, const gpsx_acq_grid_t *, const void *const *, int, gpsx_peak_t *const *, int64_t *const *)
{
  return GPSX_ENODEV;
}
#else

namespace {

// the handful of RCCL entry points used, resolved at run time so that libgpsx.so itself does not depend on librccl
// (a process that drives the ranks through torch.distributed already carries its own copy).  Types, prototypes and the
// enumerators (ncclInt64, ncclMax) come from the image's <rccl/rccl.h> at build time -- nothing is restated here.
struct Rccl {
  typedef ncclComm_t comm_t;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  static constexpr ncclDataType_t kInt64 = ncclInt64;
  static constexpr ncclRedOp_t kMax = ncclMax;
  bool ok = false;
};

Rccl &rccl()
{
  static Rccl r;
  static bool tried = false;
  if (tried)
    return r;
  tried = true;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h)
    h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h)
    return r;
  r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  r.ok = r.CommInitAll && r.CommDestroy && r.AllReduce && r.GroupStart && r.GroupEnd && r.GetErrorString;
  return r;
}

}  // namespace

struct gpsx_group {
  std::vector<gpsx_ctx *> ctxs;
  std::vector<Rccl::comm_t> comms;
};

int gpsx_group_create(gpsx_ctx *const *ctxs, int n, gpsx_group **out)
{
  if (!ctxs || !out || n < 1 || n > 64)
    return GPSX_EINVAL;
  gpsx_ctx *c0 = ctxs[0];
  std::vector<int> devs(n);
  for (int i = 0; i < n; i++) {
    if (!ctxs[i])
      return GPSX_EINVAL;
    devs[i] = ctxs[i]->device;
    for (int j = 0; j < i; j++)
      if (devs[j] == devs[i])
        return fail(c0, GPSX_EINVAL, "group: two contexts on the same device (RCCL wants one rank per GPU)");
  }
  Rccl &r = rccl();
  if (!r.ok)
    return fail(c0, GPSX_EIO, "group: librccl.so could not be loaded");
  gpsx_group *grp = new (std::nothrow) gpsx_group;
  if (!grp)
    return fail(c0, GPSX_ENOMEM, "out of host memory");
  grp->ctxs.assign(ctxs, ctxs + n);
  grp->comms.assign(n, nullptr);
  const ncclResult_t rc = r.CommInitAll(grp->comms.data(), n, devs.data());
  if (rc != ncclSuccess) {
    delete grp;
    return fail(c0, GPSX_EIO, std::string("ncclCommInitAll: ") + r.GetErrorString(rc));
  }
  *out = grp;
  return GPSX_OK;
}

void gpsx_group_destroy(gpsx_group *grp)
{
  if (!grp)
    return;
  for (size_t i = 0; i < grp->ctxs.size(); i++) {
    (void)hipSetDevice(grp->ctxs[i]->device);
    (void)hipStreamSynchronize(grp->ctxs[i]->stream);
    if (grp->comms[i])
      (void)rccl().CommDestroy(grp->comms[i]);
  }
  delete grp;
}

int gpsx_acq_grid_sharded(gpsx_group *grp, const gpsx_acq_grid_t *g, const void *const *d_if_blocks, int n_blocks,
                          gpsx_peak_t *const *d_peaks, int64_t *const *d_keys)
{
  if (!grp || !g || !d_if_blocks || !d_peaks || !d_keys)
    return GPSX_EINVAL;
  const int n = (int)grp->ctxs.size();
  gpsx_acq_grid_t gi = *g;
  gi.shard_count = n;
  for (int i = 0; i < n; i++) {
    gi.shard_index = i;
    if (!d_keys[i])
      return fail(grp->ctxs[i], GPSX_EINVAL, "sharded sweep: the key table is what gets merged, it cannot be NULL");
    if (int rc = gpsx_acq_grid_dev(grp->ctxs[i], &gi, d_if_blocks[i], n_blocks, d_peaks[i], d_keys[i], nullptr, nullptr,
                                   nullptr))
      return rc;
  }
  // the one exchange step of the path: max over ranks of (energy << 14 | 16383 - phase), entries of foreign units are 0
  Rccl &r = rccl();
  const size_t n_keys = gpsx_acq_keys_count(g);
  ncclResult_t rc = r.GroupStart();
  for (int i = 0; i < n && rc == ncclSuccess; i++) {
    (void)hipSetDevice(grp->ctxs[i]->device);
    rc = r.AllReduce(d_keys[i], d_keys[i], n_keys, Rccl::kInt64, Rccl::kMax, grp->comms[i], grp->ctxs[i]->stream);
  }
  const ncclResult_t rc_end = r.GroupEnd();
  if (rc != ncclSuccess || rc_end != ncclSuccess)
    return fail(grp->ctxs[0], GPSX_EIO, std::string("ncclAllReduce: ") + r.GetErrorString(rc != ncclSuccess ? rc : rc_end));
  return GPSX_OK;
}
#endif   // GPSX_HAVE_RCCL_HEADERS
