// ctx.hip -- context lifetime, uploads/downloads, volume import/export.
// Replaces CudaDeviceDataMan (src/cuda/CudaDeviceDataMan.h:24-67), CudaMap1D/2D (src/cuda/DataMap.h) and the
// blocking clone(CPU)/copyDataFrom calls the reference's host classes make on the singleton.
#include "kf_internal.h"
#include <string.h>
#include <stdlib.h>
#include <new>

#include <stdio.h>
#include <errno.h>
#include <signal.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>

// contexts alive per device: the persistent ICP loop needs its workgroups co-resident, which only one context per GPU guarantees
static int g_live_ctx[64];
int kf_live_contexts(int device) { return (device >= 0 && device < 64) ? __atomic_load_n(&g_live_ctx[device], __ATOMIC_RELAXED) : 2; }

// ---- the same question across PROCESSES ------------------------------------------------------------------------------------------
// Two processes that each run a persistent ICP loop on one GPU can leave both loops partly resident, each waiting for workgroups the
// other one keeps off the chip (since round 4 such a loop is finished by one workgroup instead of losing the frame, but that costs tens of
// milliseconds).  Processes of this library therefore register per physical device (keyed by PCI bus id, so HIP_VISIBLE_DEVICES renumbering
// does not matter): each holds an exclusive advisory lock (flock) on ONE slot file /dev/shm/hybkf_<uid>_<bus>.<slot> for as long as it has
// a context on the device.  Ownership is the kernel's business: a process that dies releases its lock, whatever its pid was and whichever
// pid namespace it lived in (round 3 probed raw pids with kill(pid, 0): a live peer in another namespace looked dead, a recycled pid
// looked alive).  "Is the device shared?" = can a lock on somebody else's slot NOT be taken.  Probing costs a few system calls, so it is
// done once per 256 frames and whenever this process's own contexts change; everything is behind one mutex (contexts may be created and
// tracked from several threads).  No /dev/shm: the answer is "not shared" and the loop's solo finish remains the safety net.
#include <sys/file.h>
#include <mutex>
#define KF_SHM_SLOTS 16
static std::mutex g_reg_mutex;
// (slot fd + 1 per device, so that the zero-initialised array means "none" and fd 0 -- a daemon with stdin closed -- is a valid slot)
static int g_slot_fd1[64];
static int g_slot_idx[64];
static int g_shared_cached[64];
static unsigned g_shared_calls[64];
static pid_t g_reg_pid;                       // the process the slots were taken in: a fork()ed child inherits the open file descriptions -- and with them
                                              // the PARENT's locks, which are not its own to hold: it drops them on first use and registers for itself
static bool slot_path(int device, int slot, char* out, size_t n) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, device) != hipSuccess) return false;
  for (char* q = bus; *q; ++q) if (*q == ':' || *q == '.') *q = '_';
  snprintf(out, n, "/dev/shm/hybkf_%u_%s.%d", (unsigned)getuid(), bus, slot);
  return true;
}
static void shm_after_fork_check() {                         // (under g_reg_mutex)
  const pid_t me = getpid();
  if (g_reg_pid == me) return;
  if (g_reg_pid != 0)                                        // a child of the registering process: the inherited descriptors share the parent's lock -- only close them
    for (int d = 0; d < 64; ++d) if (g_slot_fd1[d] > 0) { close(g_slot_fd1[d] - 1); g_slot_fd1[d] = 0; g_shared_cached[d] = 0; }
  g_reg_pid = me;
}
static void shm_register(int device) {                       // (under g_reg_mutex)
  shm_after_fork_check();
  if (device < 0 || device >= 64 || g_slot_fd1[device] > 0) return;
  char path[160];
  for (int s = 0; s < KF_SHM_SLOTS; ++s) {
    if (!slot_path(device, s, path, sizeof(path))) return;
    const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0600);
    if (fd < 0) return;
    if (flock(fd, LOCK_EX | LOCK_NB) == 0) { g_slot_fd1[device] = fd + 1; g_slot_idx[device] = s; g_shared_calls[device] = 0; return; }
    close(fd);
  }
}
static void shm_unregister(int device) {                     // (under g_reg_mutex)
  shm_after_fork_check();
  if (device < 0 || device >= 64 || g_slot_fd1[device] <= 0) return;
  flock(g_slot_fd1[device] - 1, LOCK_UN); close(g_slot_fd1[device] - 1);
  g_slot_fd1[device] = 0; g_shared_cached[device] = 0;
}
// does another live process of this library hold a context on `device`?
int kf_device_shared(int device) {
  if (device < 0 || device >= 64) return 0;
  std::lock_guard<std::mutex> lock(g_reg_mutex);
  if (g_reg_pid != 0 && g_reg_pid != getpid()) {             // first question after a fork(): this process holds no slot of its own yet
    shm_after_fork_check();
    if (__atomic_load_n(&g_live_ctx[device], __ATOMIC_RELAXED) > 0) shm_register(device);
  }
  if (g_slot_fd1[device] <= 0) return 0;
  if ((g_shared_calls[device]++ & 255u) != 0u) return g_shared_cached[device];
  int others = 0;
  char path[160];
  for (int s = 0; s < KF_SHM_SLOTS && !others; ++s) {
    if (s == g_slot_idx[device] || !slot_path(device, s, path, sizeof(path))) continue;
    const int fd = open(path, O_RDWR | O_CLOEXEC);
    if (fd < 0) continue;                                    // never created: nobody there
    if (flock(fd, LOCK_EX | LOCK_NB) != 0) others = 1; else flock(fd, LOCK_UN);
    close(fd);
  }
  g_shared_cached[device] = others;
  return others;
}

extern "C" const char* kf_version(void) { return "hybkf-gfx950 0.1"; }

extern "C" const char* kf_error_string(int s) {
  switch (s) {
    case 0: return "ok";
    case KF_ERR_ARG: return "invalid argument";
    case KF_ERR_STATE: return "invalid state";
    case KF_ERR_ALLOC: return "allocation failed";
    default: return hipGetErrorString((hipError_t)s);
  }
}

template <typename T>
static int dev_alloc(T** p, size_t n) {
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) { *p = nullptr; return (int)e; }
  return 0;
}

extern "C" int kf_destroy(kf_ctx* c) {
  if (!c) return KF_ERR_ARG;
  hipSetDevice(c->cfg.device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->own_stream) hipStreamSynchronize(c->own_stream);
  void* ptrs[] = {c->up_dev[0], c->up_dev[1], c->up_dev[2], c->raw_depth, c->trunced_depth, c->filtered_depth, c->raw_rgb, c->raycast_rgb, c->icp_partials, c->icp_loop_slots,
                  c->track, c->counters, c->grid_barrier, c->scratch_mats, c->vol.tw, c->vol.color, c->vol.flags, c->vol.macrobits, c->vol.negbits, c->vol.pend, c->layer_work, c->active_bricks,
                  c->tile_max_depth, c->triangles, c->mc_block_counts, c->mc_list, c->mc_nbr_bits, c->mc_partials, c->mc_codes, c->mc_surv, c->mc_block_bits, c->mc_recs, c->mc_d1_list};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->up_stream) { hipStreamSynchronize(c->up_stream); hipStreamDestroy(c->up_stream); }
  for (int i = 0; i < KF_UP_SLOTS; ++i) {
    if (c->up_host[i]) hipHostFree(c->up_host[i]);
    if (c->up_copied[i]) hipEventDestroy(c->up_copied[i]);
    if (c->up_consumed[i]) hipEventDestroy(c->up_consumed[i]);
  }
  void* alts[] = {c->alt_raw, c->alt_trunced, c->alt_filtered, c->alt_v0, c->alt_n0, c->rgb_staging, c->alt_v12[0], c->alt_v12[1], c->alt_n12[0], c->alt_n12[1]};
  for (void* p : alts) if (p) hipFree(p);
  if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
  if (c->ev_track) hipEventDestroy(c->ev_track);
  if (c->ev_preprocessed) hipEventDestroy(c->ev_preprocessed);
  if (c->ev_prefetched) hipEventDestroy(c->ev_prefetched);
  for (int l = 0; l < KF_MAX_LEVELS; ++l) {
    if (c->new_v[l]) hipFree(c->new_v[l]);
    if (c->new_n[l]) hipFree(c->new_n[l]);
    if (c->model_v[l]) hipFree(c->model_v[l]);
    if (c->model_n[l]) hipFree(c->model_n[l]);
  }
  for (int s = 0; s < 8; ++s) for (int k = 0; k < 2; ++k) for (int i = 0; i < 64; ++i) if (c->ev[s][k][i]) hipEventDestroy(c->ev[s][k][i]);
  if (c->host_pinned) hipHostFree(c->host_pinned);
  if (c->own_stream) c->stream = c->own_stream;
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->registered && c->cfg.device >= 0 && c->cfg.device < 64) {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    if (--g_live_ctx[c->cfg.device] == 0) shm_unregister(c->cfg.device);     // the process's last context on the device
  }
  delete c;
  return 0;
}

extern "C" int kf_create(const kf_config* cfg, kf_ctx** out) {
  if (!cfg || !out) return KF_ERR_ARG;
  *out = nullptr;
  const uint32_t R = cfg->volume.resolution;
  if (R == 0 || (R % KF_BRICK) != 0 || R > 8192u) return KF_ERR_ARG;      // (bricks per axis <= 1024: kf_brick_slot's 24-bit multiplies)
  if (cfg->pyramid_levels < 1 || cfg->pyramid_levels > KF_MAX_LEVELS) return KF_ERR_ARG;
  if (cfg->depth_camera.cols == 0 || cfg->depth_camera.rows == 0) return KF_ERR_ARG;
  uint32_t z0 = cfg->slab_z_begin, z1 = cfg->slab_z_end ? cfg->slab_z_end : R;
  if (z0 >= z1 || z1 > R || (z0 % KF_BRICK) || (z1 % KF_BRICK)) return KF_ERR_ARG;
  KF_CHECK(hipSetDevice(cfg->device));
  kf_ctx* c = new (std::nothrow) kf_ctx();
  if (!c) return KF_ERR_ALLOC;
  memset((void*)c, 0, sizeof(*c));
  c->cfg = *cfg;
  c->cfg.slab_z_end = z1;
  c->cols = cfg->depth_camera.cols; c->rows = cfg->depth_camera.rows; c->levels = cfg->pyramid_levels;
  { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess) cu = 0; c->num_cus = cu; }
  int st = 0;
#define TRY(x) do { st = (x); if (st) { kf_destroy(c); return st; } } while (0)
  TRY((int)hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  const size_t npx = (size_t)c->cols * c->rows;
  c->pending_slot = -1; c->n_staged = 0; c->defer_override = -1;
  TRY(dev_alloc(&c->raw_depth, npx)); TRY(dev_alloc(&c->trunced_depth, npx)); TRY(dev_alloc(&c->filtered_depth, npx));
  const size_t nrgb = (size_t)cfg->rgb_camera.cols * cfg->rgb_camera.rows;
  if (cfg->has_color) { TRY(dev_alloc(&c->raw_rgb, nrgb ? nrgb : npx)); TRY(dev_alloc(&c->raycast_rgb, npx)); }
  int lc = c->cols, lr = c->rows;
  for (int l = 0; l < c->levels; ++l) {               // CudaDeviceDataMan.h:36-47
    c->lvl_cols[l] = lc; c->lvl_rows[l] = lr;
    size_t n = (size_t)lc * lr;
    TRY(dev_alloc(&c->new_v[l], n)); TRY(dev_alloc(&c->new_n[l], n));
    TRY(dev_alloc(&c->model_v[l], n)); TRY(dev_alloc(&c->model_n[l], n));
    TRY((int)hipMemsetAsync(c->model_v[l], 0, n * sizeof(float4), c->stream));
    TRY((int)hipMemsetAsync(c->model_n[l], 0, n * sizeof(float4), c->stream));
    TRY((int)hipMemsetAsync(c->new_v[l], 0, n * sizeof(float4), c->stream));
    TRY((int)hipMemsetAsync(c->new_n[l], 0, n * sizeof(float4), c->stream));
    lc >>= 1; lr >>= 1;
  }
  TRY(dev_alloc(&c->icp_partials, (size_t)2 * KF_ICP_MAX_WG * 32));      // double-buffered by Gauss-Newton step parity
  TRY(dev_alloc(&c->icp_loop_slots, (size_t)KF_ICP_LOOP_STEPS * KF_ICP_LOOP_MAX_WG * 32));
  TRY((int)hipMemsetAsync(c->icp_loop_slots, 0, (size_t)KF_ICP_LOOP_STEPS * KF_ICP_LOOP_MAX_WG * 32 * sizeof(unsigned long long), c->stream));
  TRY(dev_alloc(&c->track, 1)); TRY(dev_alloc(&c->counters, 1)); TRY(dev_alloc(&c->grid_barrier, 1));
  TRY((int)hipMemsetAsync(c->grid_barrier, 0, sizeof(KfGridBarrier), c->stream)); TRY(dev_alloc(&c->scratch_mats, 8 * 16));
  TRY((int)hipMemsetAsync(c->track, 0, sizeof(KfTrackState), c->stream));
  TRY((int)hipMemsetAsync(c->counters, 0, sizeof(KfCounters), c->stream));
  // volume: tsdfvolume::init (tsdfVolume.h:29-37), bricked, slab + halo
  KfVolume& v = c->vol;
  v.res = (int)R; v.nb = (int)(R / KF_BRICK);
  uint32_t halo_b = (cfg->slab_halo + KF_BRICK - 1) / KF_BRICK;
  int b0 = (int)(z0 / KF_BRICK) - (int)halo_b, b1 = (int)(z1 / KF_BRICK) + (int)halo_b;
  v.bz0 = b0 < 0 ? 0 : b0; v.bz1 = b1 > v.nb ? v.nb : b1;
  v.own_z0 = (int)z0; v.own_z1 = (int)z1;
  v.size = cfg->volume.size_m; v.max_weight = cfg->volume.max_weight;
  v.cell = v.size / (float)v.res;
  c->n_stored_bricks = (size_t)(v.bz1 - v.bz0) * v.nb * v.nb;
  c->n_stored_vox = c->n_stored_bricks * KF_BRICK_VOX;
  TRY(dev_alloc(&v.tw, c->n_stored_vox));
  if (cfg->has_color) TRY(dev_alloc(&v.color, c->n_stored_vox));
  TRY(dev_alloc(&v.flags, c->n_stored_bricks + 4));      // updated with 32-bit atomics: keep the last word whole
  v.nm = (v.res + KF_MACRO - 1) / KF_MACRO;
  v.ns = (v.nm + (1 << KF_SUPER_SHIFT) - 1) >> KF_SUPER_SHIFT;
  v.macro_words = kf_bit_words((size_t)v.nm * v.nm * v.nm); v.super_words = kf_bit_words((size_t)v.ns * v.ns * v.ns);
  v.nq = (v.nb + 1) >> 1; v.meso_words = kf_bit_words((size_t)v.nq * v.nq * v.nq);
  TRY(dev_alloc(&v.macrobits, kf_skip_table_words(v)));
  TRY(dev_alloc(&v.negbits, kf_negbit_words(c->n_stored_bricks)));
  TRY(dev_alloc(&v.pend, c->n_stored_bricks));
  TRY(dev_alloc(&c->active_bricks, c->n_stored_bricks + 8));       // + 16 aligned spare bytes behind the queue (integrate.hip: queue_pad; +8 words keeps them aligned for any count)
  {                                                          // tile maxima over 8- and 16-pixel tiles, see integrate.hip
    size_t n = 0;
    for (int l = 0; l < 2; ++l) n += (size_t)kf_div_up(c->cols, 8 << l) * kf_div_up(c->rows, 8 << l);
    TRY(dev_alloc(&c->tile_max_depth, 2 * n));             // [maxima | minima]: the second half holds the tile MINIMA (same layout)
    TRY((int)hipMemsetAsync(c->tile_max_depth, 0, n * sizeof(float), c->stream));
    TRY((int)hipMemsetD32Async((hipDeviceptr_t)(c->tile_max_depth + n), 0x7F800000, n, c->stream));     // cleared minima: +inf
    c->n_tile_floats = (int)n; c->tiles_clear = 1;
  }
  c->max_triangles = cfg->max_triangles;
  if (c->max_triangles) TRY(dev_alloc(&c->triangles, (size_t)c->max_triangles));
  c->mc_blocks_cap = (c->n_stored_vox + 255) / 256;
  TRY(dev_alloc(&c->mc_block_counts, c->mc_blocks_cap + 1));
  TRY((int)hipHostMalloc(&c->host_pinned, 4096, hipHostMallocDefault));
  memset(c->host_pinned, 0, 4096);
  TRY(kf_reset_volume(c));
  TRY((int)hipStreamSynchronize(c->stream));
#undef TRY
  if (cfg->device >= 0 && cfg->device < 64) { std::lock_guard<std::mutex> lock(g_reg_mutex); ++g_live_ctx[cfg->device]; c->registered = 1; shm_register(cfg->device); }
  *out = c;
  return 0;
}

extern "C" int kf_reset_volume(kf_ctx* c) {
  if (!c) return KF_ERR_ARG;
  KF_CHECK(hipSetDevice(c->cfg.device));
  KF_CHECK(hipMemsetAsync(c->vol.tw, 0, c->n_stored_vox * sizeof(float2), c->stream));
  if (c->vol.color) KF_CHECK(hipMemsetAsync(c->vol.color, 0, c->n_stored_vox * sizeof(uchar4), c->stream));
  KF_CHECK(hipMemsetAsync(c->vol.flags, 0, c->n_stored_bricks, c->stream));
  KF_CHECK(hipMemsetAsync(c->vol.macrobits, 0, kf_skip_table_words(c->vol) * sizeof(unsigned), c->stream));
  KF_CHECK(hipMemsetAsync(c->vol.negbits, 0, kf_negbit_words(c->n_stored_bricks) * sizeof(unsigned), c->stream));
  KF_CHECK(hipMemsetAsync(c->vol.pend, 0, c->n_stored_bricks * sizeof(unsigned long long), c->stream));
  KF_CHECK(hipMemsetAsync(c->counters, 0, sizeof(KfCounters), c->stream));
  ++c->vol_flags_serial; c->pend_live = 0;
  c->wgt0_base = 0; c->wgt0_valid = 1;                       // nothing observed; the shards were zeroed with the counters
  return 0;
}

extern "C" int kf_synchronize(kf_ctx* c) {
  if (!c) return KF_ERR_ARG;
  KF_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" void* kf_stream(kf_ctx* c) { return c ? (void*)c->stream : nullptr; }
// Adopt a caller-owned HIP stream (e.g. the stream torch.distributed/RCCL work is ordered on) for everything the context
// enqueues from now on; the context's own stream is kept for destruction.  Pass NULL to return to the private stream.
extern "C" int kf_set_stream(kf_ctx* c, void* hip_stream) {
  if (!c) return KF_ERR_ARG;
  KF_CHECK(hipStreamSynchronize(c->stream));
  if (!c->own_stream) c->own_stream = c->stream;
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return 0;
}

extern "C" int kf_stored_z_range(kf_ctx* c, uint32_t* z_begin, uint32_t* z_end) {
  if (!c || !z_begin || !z_end) return KF_ERR_ARG;
  *z_begin = c->vol.bz0 * KF_BRICK; *z_end = c->vol.bz1 * KF_BRICK;
  return 0;
}

// ---- frame upload: HybKinectfu::copyFrameToGPU (src/HybKinectfu.cpp:63-96) ----------------------------------------
// `float v = mat.at<unsigned short>(row,col)*0.001;` is an int * double product narrowed to float.
__global__ void __launch_bounds__(256) k_depth_mm_to_m(const uint16_t* __restrict__ mm, float* __restrict__ out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)((double)mm[i] * 0.001);
}

// The u16 -> f32 conversion is deferred: kf_preprocess fuses it into its first kernel.  Anything else that reads raw_depth
// (kf_trunc_depth, kf_download_map) materialises it first.  The caller keeps dev_mm alive until then (bench: resident frames).
extern "C" int kf_set_depth_mm_device(kf_ctx* c, const uint16_t* dev_mm, uint32_t cols, uint32_t rows) {
  if (!c || !dev_mm || (int)cols != c->cols || (int)rows != c->rows) return KF_ERR_ARG;
  c->pending_mm = dev_mm; c->pending_slot = -1;
  return 0;
}
// the kernel that reads pending_mm has just been enqueued on the main stream: its upload buffer may be refilled after it
int kf_pending_depth_consumed(kf_ctx* c) {
  if (c->pending_slot >= 0) { KF_CHECK(hipEventRecord(c->up_consumed[c->pending_slot], c->stream)); c->pending_slot = -1; }
  return 0;
}
int kf_materialize_raw_depth(kf_ctx* c) {
  if (!c->pending_mm) return 0;
  int n = c->cols * c->rows;
  hipLaunchKernelGGL(k_depth_mm_to_m, dim3(kf_div_up(n, 256)), dim3(256), 0, c->stream, c->pending_mm, c->raw_depth, n);
  c->pending_mm = nullptr;
  int st = (int)hipGetLastError();
  return st ? st : kf_pending_depth_consumed(c);
}

// An upload slot is given up without its frame ever having been taken and preprocessed (frames staged ahead that a kf_upload_depth_mm drops; a slot that comes
// round again while a kf_prefetch_frame request still names it): up_consumed[slot] -- which the next DMA into the slot waits for -- was last recorded when the
// slot's PREVIOUS frame was consumed, but a reader of THIS frame (the prefetch's filter riding in a tracking or raycast launch) may still be queued on the
// context's stream.  Record the event where the stream stands now (everything enqueued so far precedes the refill) and void what was derived from the frame.
static int up_slot_abandon(kf_ctx* c, int slot) {
  if (slot < 0 || slot >= KF_UP_SLOTS || !c->up_used[slot]) return 0;
  KF_CHECK(hipEventRecord(c->up_consumed[slot], c->stream));
  const uint16_t* dev = c->up_dev[slot];
  if (c->fp_src == dev) { c->fp_pending = 0; c->fp_filtered = 0; c->fp_done = 0; }
  if (c->prefetch_src == dev) c->prefetch_valid = 0;
  c->up_unwaited[slot] = 0;
  return 0;
}
// pinned staging -> DMA on the copy stream -> device slot `*slot_out`.  Three slots in rotation: the current frame and up to two staged ahead.
// wait_now: the context's stream waits for the copy at once (the frame is used next); otherwise whoever first reads the slot on that stream
// asks for the wait (kf_upload_wait_for) -- a frame staged two ahead has crossed PCIe long before anything reads it.
static int upload_into_next_slot(kf_ctx* c, const uint16_t* host_mm, uint32_t cols, uint32_t rows, bool wait_now, int* slot_out) {
  if (!c || !host_mm || (int)cols != c->cols || (int)rows != c->rows) return KF_ERR_ARG;
  KF_CHECK(hipSetDevice(c->cfg.device));
  const size_t bytes = (size_t)cols * rows * sizeof(uint16_t);
  if (!c->up_stream) {
    KF_CHECK(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    for (int i = 0; i < KF_UP_SLOTS; ++i) {
      KF_CHECK(hipHostMalloc((void**)&c->up_host[i], bytes, hipHostMallocDefault));
      KF_CHECK(hipMalloc((void**)&c->up_dev[i], bytes));
      KF_CHECK(hipEventCreateWithFlags(&c->up_copied[i], hipEventDisableTiming));
      KF_CHECK(hipEventCreateWithFlags(&c->up_consumed[i], hipEventDisableTiming));
    }
  }
  const int p = c->up_next; c->up_next = (p + 1) % KF_UP_SLOTS;
  if (c->pending_slot == p) c->pending_slot = -1;            // the frame uploaded three calls ago was never consumed: it is being replaced
  if (c->up_used[p] && (c->up_unwaited[p] || (c->fp_pending && c->fp_src == c->up_dev[p]) || (c->prefetch_valid && c->prefetch_src == c->up_dev[p]))) {
    const int st = up_slot_abandon(c, p);                    // still referenced by a staged copy / a prefetch nobody picked up: its readers first
    if (st) return st;
  }
  if (c->up_used[p]) {
    KF_CHECK(hipEventSynchronize(c->up_copied[p]));           // the pinned buffer's previous DMA (three uploads ago: long finished)
    KF_CHECK(hipStreamWaitEvent(c->up_stream, c->up_consumed[p], 0));   // the device buffer's last reader
  }
  memcpy(c->up_host[p], host_mm, bytes);
  KF_CHECK(hipMemcpyAsync(c->up_dev[p], c->up_host[p], bytes, hipMemcpyHostToDevice, c->up_stream));
  KF_CHECK(hipEventRecord(c->up_copied[p], c->up_stream));
  c->up_unwaited[p] = 1;
  if (wait_now) { KF_CHECK(hipStreamWaitEvent(c->stream, c->up_copied[p], 0)); c->up_unwaited[p] = 0; }
  if (!c->up_used[p]) KF_CHECK(hipEventRecord(c->up_consumed[p], c->stream));   // give the event a defined state before its first wait
  c->up_used[p] = 1;
  *slot_out = p;
  return 0;
}
// a kernel that reads `dev_mm` is about to be enqueued on the context's stream: if that is an upload slot whose copy the stream has not waited for, it does now
int kf_upload_wait_for(kf_ctx* c, const uint16_t* dev_mm) {
  for (int i = 0; i < KF_UP_SLOTS; ++i)
    if (c->up_unwaited[i] && dev_mm == c->up_dev[i]) {
      // a frame staged two ahead has usually crossed PCIe by now: then there is nothing for the stream to wait for, and no barrier packet goes into it
      const hipError_t q = hipEventQuery(c->up_copied[i]);
      if (q == hipErrorNotReady) KF_CHECK(hipStreamWaitEvent(c->stream, c->up_copied[i], 0));
      else if (q != hipSuccess) return (int)q;
      c->up_unwaited[i] = 0;
    }
  return 0;
}
extern "C" int kf_upload_depth_mm(kf_ctx* c, const uint16_t* host_mm, uint32_t cols, uint32_t rows) {
  int p = -1;
  const int st = upload_into_next_slot(c, host_mm, cols, rows, true, &p);
  if (st) return st;
  for (int i = 0; i < c->n_staged; ++i) { const int ds = up_slot_abandon(c, c->staged[i]); if (ds) return ds; }   // frames staged behind the one this call replaces are dropped with it
  c->n_staged = 0;
  c->pending_mm = c->up_dev[p]; c->pending_slot = p;
  return 0;
}
// Frames AHEAD of the current one (HybKinectfu::copyFrameToGPU src/HybKinectfu.cpp:63-96, run early): the upload goes into a free slot and
// does not replace the current frame; up to two frames may be staged, kf_take_next_depth makes the oldest of them the current one (what
// kf_upload_depth_mm would have done).  *dev_mm (may be null) is where the frame lies on the device -- what kf_prefetch_frame wants, so that
// the frame's front end rides in its predecessor's launches.  Staging TWO ahead (upload frame k+2 while frame k is processed and frame k+1's
// front end rides along) takes the copy off the critical path altogether: a host-fed stream then runs at the resident stream's rate.
extern "C" int kf_upload_depth_mm_next(kf_ctx* c, const uint16_t* host_mm, uint32_t cols, uint32_t rows, const uint16_t** dev_mm) {
  if (!c) return KF_ERR_ARG;
  if (c->n_staged >= 2) return KF_ERR_STATE;
  int p = -1;
  const int st = upload_into_next_slot(c, host_mm, cols, rows, false, &p);
  if (st) return st;
  c->staged[c->n_staged++] = p;
  if (dev_mm) *dev_mm = c->up_dev[p];
  return 0;
}
extern "C" int kf_take_next_depth(kf_ctx* c) {
  if (!c) return KF_ERR_ARG;
  if (c->n_staged < 1) return KF_ERR_STATE;
  const int p = c->staged[0];
  c->staged[0] = c->staged[1]; c->n_staged--;
  c->pending_mm = c->up_dev[p]; c->pending_slot = p;
  return kf_upload_wait_for(c, c->pending_mm);
}

__global__ void __launch_bounds__(256) k_rgb3_to_rgb4(const unsigned char* __restrict__ in, uchar4* __restrict__ out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = make_uchar4(in[3 * i], in[3 * i + 1], in[3 * i + 2], 0);
}
__global__ void __launch_bounds__(256) k_rgb4_to_rgb3(const uchar4* __restrict__ in, unsigned char* __restrict__ out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { uchar4 p = in[i]; out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z; }
}

extern "C" int kf_upload_rgb(kf_ctx* c, const uint8_t* host_bgr, uint32_t cols, uint32_t rows) {
  if (!c || !host_bgr || !c->raw_rgb || cols != c->cfg.rgb_camera.cols || rows != c->cfg.rgb_camera.rows) return KF_ERR_ARG;
  const size_t n = (size_t)cols * rows;
  if (!c->rgb_staging) KF_CHECK(hipMalloc((void**)&c->rgb_staging, n * 3));        // 3-byte pixels as they come from the host
  // the copy out of (pageable) host memory has returned to the caller by the time this function does: no synchronisation needed
  KF_CHECK(hipMemcpyAsync(c->rgb_staging, host_bgr, n * 3, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_rgb3_to_rgb4, dim3(kf_div_up((int)n, 256)), dim3(256), 0, c->stream, c->rgb_staging, c->raw_rgb, (int)n);
  return (int)hipGetLastError();
}

// ---- map download / upload (CudaMap2D::clone(CPU), DataMap.h) -----------------------------------------------------
static int map_ptr(kf_ctx* c, int id, uint32_t level, void** p, size_t* bytes, bool* is_rgb) {
  *is_rgb = false;
  size_t npx = (size_t)c->cols * c->rows;
  switch (id) {
    case KF_MAP_RAW_DEPTH: *p = c->raw_depth; *bytes = npx * 4; return 0;
    case KF_MAP_TRUNCED_DEPTH: *p = c->trunced_depth; *bytes = npx * 4; return 0;
    case KF_MAP_FILTERED_DEPTH: *p = c->filtered_depth; *bytes = npx * 4; return 0;
    case KF_MAP_RAW_RGB: *p = c->raw_rgb; *bytes = (size_t)c->cfg.rgb_camera.cols * c->cfg.rgb_camera.rows * 3; *is_rgb = true; return c->raw_rgb ? 0 : KF_ERR_STATE;
    case KF_MAP_RAYCAST_RGB: *p = c->raycast_rgb; *bytes = npx * 3; *is_rgb = true; return c->raycast_rgb ? 0 : KF_ERR_STATE;
    default: break;
  }
  if (level >= (uint32_t)c->levels) return KF_ERR_ARG;
  size_t n = (size_t)c->lvl_cols[level] * c->lvl_rows[level] * sizeof(float4);
  switch (id) {
    case KF_MAP_NEW_VERTICES: *p = c->new_v[level]; break;
    case KF_MAP_NEW_NORMALS: *p = c->new_n[level]; break;
    case KF_MAP_MODEL_VERTICES: *p = c->model_v[level]; break;
    case KF_MAP_MODEL_NORMALS: *p = c->model_n[level]; break;
    default: return KF_ERR_ARG;
  }
  *bytes = n;
  return 0;
}

extern "C" int kf_download_map(kf_ctx* c, int id, uint32_t level, void* dst, size_t dst_bytes) {
  if (!c || !dst) return KF_ERR_ARG;
  if (id == KF_MAP_RAW_DEPTH) { int ms = kf_materialize_raw_depth(c); if (ms) return ms; }
  void* p; size_t bytes; bool rgb;
  int st = map_ptr(c, id, level, &p, &bytes, &rgb);
  if (st) return st;
  if (dst_bytes < bytes) return KF_ERR_ARG;
  if (rgb) {
    int n = (int)(bytes / 3);
    unsigned char* tmp = nullptr;
    KF_CHECK(hipMalloc((void**)&tmp, bytes));
    hipLaunchKernelGGL(k_rgb4_to_rgb3, dim3(kf_div_up(n, 256)), dim3(256), 0, c->stream, (const uchar4*)p, tmp, n);
    hipError_t e = hipMemcpyAsync(dst, tmp, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(tmp);
    return (int)e;
  }
  KF_CHECK(hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int kf_upload_map(kf_ctx* c, int id, uint32_t level, const void* src, size_t src_bytes) {
  if (!c || !src) return KF_ERR_ARG;
  void* p; size_t bytes; bool rgb;
  int st = map_ptr(c, id, level, &p, &bytes, &rgb);
  if (st) return st;
  if (src_bytes != bytes) return KF_ERR_ARG;
  if (id == KF_MAP_TRUNCED_DEPTH) c->trunc_serial++;             // (only once the arguments are known to be good) the integrate tile maxima no longer describe this map
  if (rgb) {
    if (id != KF_MAP_RAW_RGB) return KF_ERR_ARG;
    return kf_upload_rgb(c, (const uint8_t*)src, c->cfg.rgb_camera.cols, c->cfg.rgb_camera.rows);
  }
  if (id == KF_MAP_NEW_VERTICES || id == KF_MAP_NEW_NORMALS) c->new_pyr_ok = 0;          // the tracker rebuilds the pyramids from level 0, as it always did
  if (id == KF_MAP_MODEL_VERTICES || id == KF_MAP_MODEL_NORMALS) c->model_pyr_ok = 0;
  KF_CHECK(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- volume import / export in the reference's linear order (tsdfVolume.h:57-60) -----------------------------------
__global__ void __launch_bounds__(256) k_volume_export(KfVolume v, int z_begin, int z_end, float* __restrict__ tsdf,
                                                       float* __restrict__ weight, unsigned char* __restrict__ color) {
  size_t n = (size_t)(z_end - z_begin) * v.res * v.res;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    int x = (int)(i % v.res), y = (int)((i / v.res) % v.res), z = z_begin + (int)(i / ((size_t)v.res * v.res));
    size_t idx = kf_vox_index(v, x, y, z);
    float2 q = v.tw[idx];
    // a quarter brick in a deferred state: the pending whole-quarter free-space steps belong to the weight (kf_internal.h, KF_PEND_SAT)
    const unsigned pq = reinterpret_cast<const unsigned short*>(v.pend)[(idx >> 9) * 4u + ((idx >> 7) & 3u)];
    tsdf[i] = q.x; weight[i] = kf_pend_weight(q.y, pq, v.max_weight);
    if (color && v.color) { uchar4 cc = v.color[idx]; color[3 * i] = cc.x; color[3 * i + 1] = cc.y; color[3 * i + 2] = cc.z; }
  }
}
__global__ void __launch_bounds__(256) k_volume_import(KfVolume v, int z_begin, int z_end, const float* __restrict__ tsdf,
                                                       const float* __restrict__ weight, const unsigned char* __restrict__ color) {
  size_t n = (size_t)(z_end - z_begin) * v.res * v.res;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    int x = (int)(i % v.res), y = (int)((i / v.res) % v.res), z = z_begin + (int)(i / ((size_t)v.res * v.res));
    size_t idx = kf_vox_index(v, x, y, z);
    v.tw[idx] = make_float2(tsdf[i], weight[i]);
    if (color && v.color) v.color[idx] = make_uchar4(color[3 * i], color[3 * i + 1], color[3 * i + 2], 0);
  }
}
// rebuild the per-brick flags from the voxel data (after an import)
__global__ void __launch_bounds__(256) k_rebuild_flags(KfVolume v, size_t n_bricks) {
  __shared__ unsigned s_flag;
  for (size_t b = blockIdx.x; b < n_bricks; b += gridDim.x) {
    if (threadIdx.x == 0) s_flag = 0;
    __syncthreads();
    unsigned f = 0;
    for (int k = threadIdx.x; k < KF_BRICK_VOX; k += 256) {
      float2 q = v.tw[b * KF_BRICK_VOX + k];
      if (q.y > 0.f) f |= KF_FLAG_OBSERVED;
      if (q.x < 0.f) f |= KF_FLAG_HASNEG;
    }
    if (f) atomicOr(&s_flag, f);
    __syncthreads();
    if (threadIdx.x == 0) {
      v.flags[b] = (uint8_t)s_flag;
      if (s_flag & KF_FLAG_HASNEG) atomicOr(&v.negbits[b >> 5], 1u << (b & 31)); else atomicAnd(&v.negbits[b >> 5], ~(1u << (b & 31)));
      if (s_flag & KF_FLAG_HASNEG) kf_mark_macro(v, (int)(b % v.nb), (int)((b / v.nb) % v.nb), (int)(b / ((size_t)v.nb * v.nb)) + v.bz0);
    }
    __syncthreads();
  }
}

static int volume_xfer(kf_ctx* c, uint32_t z0, uint32_t z1, float* tsdf, float* weight, uint8_t* color, bool to_host) {
  if (!c || !tsdf || !weight) return KF_ERR_ARG;
  if (z0 >= z1 || (int)z0 < c->vol.bz0 * KF_BRICK || (int)z1 > c->vol.bz1 * KF_BRICK) return KF_ERR_ARG;
  size_t n = (size_t)(z1 - z0) * c->vol.res * c->vol.res;
  float *dt = nullptr, *dw = nullptr; unsigned char* dc = nullptr;
  hipError_t e = hipMalloc((void**)&dt, n * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&dw, n * 4);
  if (e == hipSuccess && color && c->vol.color) e = hipMalloc((void**)&dc, n * 3);
  int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  if (e == hipSuccess) {
    if (to_host) {
      hipLaunchKernelGGL(k_volume_export, dim3(grid), dim3(256), 0, c->stream, c->vol, (int)z0, (int)z1, dt, dw, dc);
      e = hipMemcpyAsync(tsdf, dt, n * 4, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(weight, dw, n * 4, hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess && dc) e = hipMemcpyAsync(color, dc, n * 3, hipMemcpyDeviceToHost, c->stream);
    } else {
      e = hipMemcpyAsync(dt, tsdf, n * 4, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(dw, weight, n * 4, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess && dc) e = hipMemcpyAsync(dc, color, n * 3, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_volume_import, dim3(grid), dim3(256), 0, c->stream, c->vol, (int)z0, (int)z1, dt, dw, dc);
        int fg = (int)(c->n_stored_bricks > 4096 ? 4096 : c->n_stored_bricks);
        hipMemsetAsync(c->vol.macrobits, 0, kf_skip_table_words(c->vol) * sizeof(unsigned), c->stream);   // rebuilt with the flags
        hipLaunchKernelGGL(k_rebuild_flags, dim3(fg), dim3(256), 0, c->stream, c->vol, c->n_stored_bricks);
      }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  }
  if (dt) hipFree(dt);
  if (dw) hipFree(dw);
  if (dc) hipFree(dc);
  return (int)e;
}

extern "C" int kf_download_volume(kf_ctx* c, uint32_t z0, uint32_t z1, float* tsdf, float* weight, uint8_t* color) {
  return volume_xfer(c, z0, z1, tsdf, weight, color, true);
}
extern "C" int kf_upload_volume(kf_ctx* c, uint32_t z0, uint32_t z1, const float* tsdf, const float* weight, const uint8_t* color) {
  if (!c || !tsdf || !weight) return KF_ERR_ARG;
  if (z0 >= z1 || (int)z0 < c->vol.bz0 * KF_BRICK || (int)z1 > c->vol.bz1 * KF_BRICK) return KF_ERR_ARG;       // volume_xfer's own checks, before any bookkeeping moves
  ++c->vol_flags_serial;                                   // the upload rebuilds the brick flags: some may be cleared
  c->wgt0_valid = 0;                                       // ... and writes weights behind the running count's back
  // pending weight steps are applied first (the upload may cover part of a quarter brick), then every deferred-weight word is dropped
  { const int fs = kf_flush_pending(c); if (fs) return fs; }
  KF_CHECK(hipMemsetAsync(c->vol.pend, 0, c->n_stored_bricks * sizeof(unsigned long long), c->stream)); c->pend_live = 0;
  return volume_xfer(c, z0, z1, (float*)tsdf, (float*)weight, (uint8_t*)color, false);
}

// ---- the same import / export with CALLER-OWNED DEVICE buffers, asynchronous on the context's stream --------------------------------
// What z-slab ranks move between GPUs when slab boundaries change (pipeline.SlabPipeline.rebalance: torch tensors sent with
// torch.distributed send / recv -- RCCL point-to-point over xGMI): planes in the reference's linear order, pending weight steps applied.
extern "C" int kf_download_volume_device(kf_ctx* c, uint32_t z0, uint32_t z1, float* dev_tsdf, float* dev_weight, uint8_t* dev_color) {
  if (!c || !dev_tsdf || !dev_weight) return KF_ERR_ARG;
  if (z0 >= z1 || (int)z0 < c->vol.bz0 * KF_BRICK || (int)z1 > c->vol.bz1 * KF_BRICK) return KF_ERR_ARG;
  const size_t n = (size_t)(z1 - z0) * c->vol.res * c->vol.res;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipLaunchKernelGGL(k_volume_export, dim3(grid), dim3(256), 0, c->stream, c->vol, (int)z0, (int)z1, dev_tsdf, dev_weight, (unsigned char*)dev_color);
  return (int)hipGetLastError();
}
extern "C" int kf_upload_volume_device(kf_ctx* c, uint32_t z0, uint32_t z1, const float* dev_tsdf, const float* dev_weight, const uint8_t* dev_color) {
  if (!c || !dev_tsdf || !dev_weight) return KF_ERR_ARG;
  if (z0 >= z1 || (int)z0 < c->vol.bz0 * KF_BRICK || (int)z1 > c->vol.bz1 * KF_BRICK) return KF_ERR_ARG;
  ++c->vol_flags_serial; c->wgt0_valid = 0;
  { const int fs = kf_flush_pending(c); if (fs) return fs; }
  KF_CHECK(hipMemsetAsync(c->vol.pend, 0, c->n_stored_bricks * sizeof(unsigned long long), c->stream)); c->pend_live = 0;
  const size_t n = (size_t)(z1 - z0) * c->vol.res * c->vol.res;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipLaunchKernelGGL(k_volume_import, dim3(grid), dim3(256), 0, c->stream, c->vol, (int)z0, (int)z1, dev_tsdf, dev_weight, (const unsigned char*)dev_color);
  const int fg = (int)(c->n_stored_bricks > 4096 ? 4096 : c->n_stored_bricks);
  KF_CHECK(hipMemsetAsync(c->vol.macrobits, 0, kf_skip_table_words(c->vol) * sizeof(unsigned), c->stream));   // rebuilt with the flags
  hipLaunchKernelGGL(k_rebuild_flags, dim3(fg), dim3(256), 0, c->stream, c->vol, c->n_stored_bricks);
  return (int)hipGetLastError();
}

// ---- a z-slab context changes the layers it owns (dynamic re-balancing of the slab boundaries) -----------------------------------------------
// New owned range [z_begin, z_end) + halo -> new stored brick layers.  The layers stored before AND after keep their voxels (a brick layer is
// one contiguous run of bricks: device-to-device copies); layers that are new read as never observed until the caller fills them
// (kf_upload_volume_device with what their previous owner sent); layers no longer stored are dropped.  Everything else of the context --
// device-resident pose and tracker state, frame maps, counters, streams -- stays.  Blocking (allocations).
extern "C" int kf_resize_slab(kf_ctx* c, uint32_t z_begin, uint32_t z_end, uint32_t halo) {
  if (!c) return KF_ERR_ARG;
  KfVolume& v = c->vol;
  const uint32_t R = (uint32_t)v.res;
  if (z_begin >= z_end || z_end > R || (z_begin % KF_BRICK) || (z_end % KF_BRICK)) return KF_ERR_ARG;
  KF_CHECK(hipSetDevice(c->cfg.device));
  { const int ds = kf_tail_cull_discard(c); if (ds) return ds; }   // (a cull that ran for the old slab)
  c->wgt0_valid = 0;                                               // another owned range: the running count of observed voxels is re-based by the next kf_get_volume_stats
  const int halo_b = (int)((halo + KF_BRICK - 1) / KF_BRICK);
  int nb0 = (int)(z_begin / KF_BRICK) - halo_b, nb1 = (int)(z_end / KF_BRICK) + halo_b;
  nb0 = nb0 < 0 ? 0 : nb0; nb1 = nb1 > v.nb ? v.nb : nb1;
  { const int fs = kf_flush_pending(c); if (fs) return fs; }                        // the stored weights become authoritative: no word moves
  const size_t per_layer = (size_t)v.nb * v.nb;                                      // bricks of one brick layer
  const size_t n_new = (size_t)(nb1 - nb0) * per_layer;
  float2* tw = nullptr; uchar4* color = nullptr; uint8_t* flags = nullptr; unsigned* negbits = nullptr; unsigned long long* pend = nullptr;
  unsigned* queue = nullptr; unsigned* mc_counts = nullptr;
  int st = 0;
  if (!st) st = dev_alloc(&tw, n_new * KF_BRICK_VOX);
  if (!st && v.color) st = dev_alloc(&color, n_new * KF_BRICK_VOX);
  if (!st) st = dev_alloc(&flags, n_new + 4);
  if (!st) st = dev_alloc(&negbits, kf_negbit_words(n_new));
  if (!st) st = dev_alloc(&pend, n_new);
  if (!st) st = dev_alloc(&queue, n_new + 8);
  if (!st) st = dev_alloc(&mc_counts, (n_new * KF_BRICK_VOX + 255) / 256 + 1);
  hipError_t e = hipSuccess;
  if (!st) {
    e = hipMemsetAsync(tw, 0, n_new * KF_BRICK_VOX * sizeof(float2), c->stream);
    if (e == hipSuccess && color) e = hipMemsetAsync(color, 0, n_new * KF_BRICK_VOX * sizeof(uchar4), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(flags, 0, n_new + 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(negbits, 0, kf_negbit_words(n_new) * sizeof(unsigned), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(pend, 0, n_new * sizeof(unsigned long long), c->stream);
    const int lo = nb0 > v.bz0 ? nb0 : v.bz0, hi = nb1 < v.bz1 ? nb1 : v.bz1;       // brick layers stored before and after
    if (e == hipSuccess && hi > lo) {
      const size_t src = (size_t)(lo - v.bz0) * per_layer * KF_BRICK_VOX, dst = (size_t)(lo - nb0) * per_layer * KF_BRICK_VOX, cnt = (size_t)(hi - lo) * per_layer * KF_BRICK_VOX;
      e = hipMemcpyAsync(tw + dst, v.tw + src, cnt * sizeof(float2), hipMemcpyDeviceToDevice, c->stream);
      if (e == hipSuccess && color) e = hipMemcpyAsync(color + dst, v.color + src, cnt * sizeof(uchar4), hipMemcpyDeviceToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    st = (int)e;
  }
  if (st) {
    void* fresh[] = {tw, color, flags, negbits, pend, queue, mc_counts};
    for (void* p : fresh) if (p) hipFree(p);
    return st;
  }
  void* old[] = {v.tw, v.color, v.flags, v.negbits, v.pend, c->active_bricks, c->mc_block_counts,
                 c->mc_list, c->mc_nbr_bits, c->mc_partials, c->mc_codes, c->mc_surv, c->mc_block_bits, c->mc_recs, c->mc_d1_list};
  for (void* p : old) if (p) hipFree(p);
  v.tw = tw; v.color = color; v.flags = flags; v.negbits = negbits; v.pend = pend; c->active_bricks = queue; c->mc_block_counts = mc_counts;
  c->mc_list = nullptr; c->mc_nbr_bits = nullptr; c->mc_partials = nullptr; c->mc_codes = nullptr; c->mc_surv = nullptr; c->mc_block_bits = nullptr; c->mc_recs = nullptr; c->mc_d1_list = nullptr;
  v.bz0 = nb0; v.bz1 = nb1; v.own_z0 = (int)z_begin; v.own_z1 = (int)z_end;
  c->cfg.slab_z_begin = z_begin; c->cfg.slab_z_end = z_end; c->cfg.slab_halo = halo;
  c->n_stored_bricks = n_new; c->n_stored_vox = n_new * KF_BRICK_VOX; c->mc_blocks_cap = (c->n_stored_vox + 255) / 256;
  ++c->vol_flags_serial; c->pend_live = 0;
  // brick flags, has-negative bits and the macro / super cell tables of what is stored now
  KF_CHECK(hipMemsetAsync(v.macrobits, 0, kf_skip_table_words(v) * sizeof(unsigned), c->stream));
  hipLaunchKernelGGL(k_rebuild_flags, dim3((unsigned)(n_new > 4096 ? 4096 : n_new)), dim3(256), 0, c->stream, v, n_new);
  KF_CHECK(hipStreamSynchronize(c->stream));
  return (int)hipGetLastError();
}

// ---- work per brick layer (what the slab boundaries are balanced on) --------------------------------------------------------------------------
// kf_count_layer_work: the next `frames` kf_integrate_volume calls add, per brick layer of the WHOLE volume, the voxels of queued bricks that passed
// the update predicate (bricks the cull retires as whole free space cost nothing and are not counted).  kf_read_layer_work: blocking; out has
// resolution / 8 entries (zeros for layers this context does not store); reset != 0 clears the counts afterwards.
extern "C" int kf_count_layer_work(kf_ctx* c, int frames) {
  if (!c || frames < 0) return KF_ERR_ARG;
  if (!c->layer_work) {
    KF_CHECK(hipMalloc((void**)&c->layer_work, (size_t)c->vol.nb * sizeof(unsigned long long)));
    KF_CHECK(hipMemsetAsync(c->layer_work, 0, (size_t)c->vol.nb * sizeof(unsigned long long), c->stream));
  }
  c->layer_work_frames = frames;
  return 0;
}
extern "C" int kf_read_layer_work(kf_ctx* c, uint64_t* out, int reset) {
  if (!c || !out) return KF_ERR_ARG;
  if (!c->layer_work) { memset(out, 0, (size_t)c->vol.nb * sizeof(uint64_t)); return 0; }
  KF_CHECK(hipMemcpyAsync(out, c->layer_work, (size_t)c->vol.nb * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  if (reset) KF_CHECK(hipMemsetAsync(c->layer_work, 0, (size_t)c->vol.nb * sizeof(unsigned long long), c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- statistics ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count_weight(KfVolume v, KfCounters* cnt) {
  // owned layers only; bricks are 8 layers thick and the owned range is brick aligned
  size_t b0 = (size_t)(v.own_z0 / KF_BRICK - v.bz0) * v.nb * v.nb, b1 = (size_t)(v.own_z1 / KF_BRICK - v.bz0) * v.nb * v.nb;
  size_t n0 = b0 * KF_BRICK_VOX, n1 = b1 * KF_BRICK_VOX;
  unsigned local = 0;
  for (size_t i = n0 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += (size_t)gridDim.x * 256) local += v.tw[i].y > 0.f ? 1u : 0u;
  float s = kf_wave_sum((float)local);      // local <= n/(grid*256) stays far below 2^24: exact in fp32
  if ((threadIdx.x & 63) == 0 && s > 0.f) atomicAdd(&cnt->weight_gt0, (unsigned long long)s);
}

// the sweep: every owned voxel with weight > 0, counted from the volume itself (blocking).  kf_get_volume_stats uses it only to (re)base its running count --
// after an upload or a slab resize; tests call it as the cross-check of that running count (kf_count_observed_voxels)
static int sweep_observed(kf_ctx* c, unsigned long long* out) {
  KF_CHECK(hipMemsetAsync(&c->counters->weight_gt0, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_count_weight, dim3(2048), dim3(256), 0, c->stream, c->vol, c->counters);
  KF_CHECK(hipMemcpyAsync(c->host_pinned, &c->counters->weight_gt0, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  memcpy(out, c->host_pinned, sizeof(unsigned long long));
  return 0;
}
extern "C" int kf_count_observed_voxels(kf_ctx* c, uint64_t* out) {
  if (!c || !out) return KF_ERR_ARG;
  unsigned long long n = 0;
  const int st = sweep_observed(c, &n);
  *out = n;
  return st;
}
// something other than a fusion kernel has written weights (upload, slab resize): the running count no longer describes the volume
void kf_observed_count_invalidate(kf_ctx* c) { c->wgt0_valid = 0; }

// everything kf_get_volume_stats reports EXCEPT the observed-voxel count (weight_gt0 = 0): the fusion passes' update counters, read back -- no sweep, and not
// a "question" for the count's bookkeeping (a measurement harness that brackets its regions with this call does not switch the COUNT kernels on)
static int read_fusion_counters(kf_ctx* c, kf_volume_stats* out, unsigned long long* fresh_out) {
  KfCounters h;
  KF_CHECK(hipMemcpyAsync(&h, c->counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  KF_CHECK(hipStreamSynchronize(c->stream));
  unsigned long long last = 0, total = 0, fresh = 0;
  for (int i = 0; i < 64; ++i) { last += h.upd_shard[c->last_parity][i * 16]; total += h.upd_shard[c->last_parity ^ 1][i * 16] + h.upd_total_shard[i]; fresh += h.wgt0_shard[i * 16]; }
  out->updated_last = last; out->weight_gt0 = 0;
  out->bricks_active = h.n_active[c->last_parity]; out->bricks_total = c->n_stored_bricks;
  out->updated_total = total + last; out->frames_fused = h.frames_fused; out->frames_lost = h.frames_lost;
  if (fresh_out) *fresh_out = fresh;
  return 0;
}
extern "C" int kf_get_fusion_counters(kf_ctx* c, kf_volume_stats* out) {
  if (!c || !out) return KF_ERR_ARG;
  return read_fusion_counters(c, out, nullptr);
}

extern "C" int kf_get_volume_stats(kf_ctx* c, kf_volume_stats* out) {
  if (!c || !out) return KF_ERR_ARG;
  // weight_gt0 (the reference prints it after every integrate: integrateVolume.cu:91-94).  A host that asks now and then gets it from a sweep of the volume
  // (0.25 ms at 512^3, 1.5 ms at 1024^3, 12 ms at 2048^3); one that asks again within 8 fused frames switches the fusion launches to their COUNT instantiations,
  // which add the voxels they observe for the first time to KfCounters::wgt0_shard (+3.6 us per frame at 512^3): from then on the count is a read-back.
  // KF_OBSERVED_COUNT=0: always sweep; 1: track from the first question on.
  static int mode_env = -2;
  if (mode_env == -2) { const char* e = getenv("KF_OBSERVED_COUNT"); mode_env = e ? atoi(e) : -1; }
  const bool frequent = c->wgt0_asked_before && c->wgt0_frames_unasked <= 8;
  c->wgt0_asked_before = 1; c->wgt0_frames_unasked = 0;
  if (mode_env == 0) c->wgt0_tracking = 0;
  else if (mode_env == 1 || frequent) c->wgt0_tracking = 1;
  if (!(c->wgt0_tracking && c->wgt0_valid)) {
    unsigned long long n = 0;
    const int st = sweep_observed(c, &n);
    if (st) return st;
    KF_CHECK(hipMemsetAsync(c->counters->wgt0_shard, 0, sizeof(c->counters->wgt0_shard), c->stream));      // (stream-ordered in front of the next fusion pass)
    c->wgt0_base = n; c->wgt0_valid = 1;
  }
  unsigned long long fresh = 0;
  const int st = read_fusion_counters(c, out, &fresh);
  if (st) return st;
  out->weight_gt0 = c->wgt0_base + fresh;
  return 0;
}

// ---- per-stage device timers ----------------------------------------------------------------------------------------
static void evt_fold(kf_ctx* c, int s) {
  if (c->ev_n[s] == 0) return;
  (void)hipEventSynchronize(c->ev[s][1][c->ev_n[s] - 1]);
  for (int i = 0; i < c->ev_n[s]; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[s][0][i], c->ev[s][1][i]) == hipSuccess) { c->ev_ms[s] += ms; c->ev_count[s] += 1; }
  }
  c->ev_n[s] = 0;
}
void kf_evt_begin(kf_ctx* c, int s) {
  if (!(c->timers_enabled & (1 << s))) return;
  c->ev_open[s] = (c->ev_seen[s]++ % c->timers_period) == 0;          // sampling: every timers_period-th interval is timed
  if (!c->ev_open[s]) return;
  if (c->ev_n[s] == 64) evt_fold(c, s);
  if (!c->ev[s][0][c->ev_n[s]]) { (void)hipEventCreate(&c->ev[s][0][c->ev_n[s]]); (void)hipEventCreate(&c->ev[s][1][c->ev_n[s]]); }
  (void)hipEventRecord(c->ev[s][0][c->ev_n[s]], c->stream);
}
// A kernel timed by its OWN dispatch: the event pair is handed to hipExtLaunchKernelGGL, which stamps it with the dispatch's start and end
// (what rocprofv3 reports as the kernel's duration) instead of two separate event records around the launch -- those add the records' own
// latency, ~3-5 us, which is a quarter of a 19 us kernel.  Returns false when this interval is not sampled (launch plainly); on true the
// caller launches with (*e0, *e1) and then calls kf_evt_attached_done.
bool kf_evt_attach(kf_ctx* c, int s, hipEvent_t* e0, hipEvent_t* e1) {
  if (!(c->timers_enabled & (1 << s))) return false;
  if ((c->ev_seen[s]++ % c->timers_period) != 0) return false;
  if (c->ev_n[s] == 64) evt_fold(c, s);
  if (!c->ev[s][0][c->ev_n[s]]) { (void)hipEventCreate(&c->ev[s][0][c->ev_n[s]]); (void)hipEventCreate(&c->ev[s][1][c->ev_n[s]]); }
  *e0 = c->ev[s][0][c->ev_n[s]]; *e1 = c->ev[s][1][c->ev_n[s]];
  return true;
}
void kf_evt_attached_done(kf_ctx* c, int s) { c->ev_n[s] += 1; }
void kf_evt_end(kf_ctx* c, int s) {
  if (!(c->timers_enabled & (1 << s)) || !c->ev_open[s]) return;
  c->ev_open[s] = 0;
  (void)hipEventRecord(c->ev[s][1][c->ev_n[s]], c->stream);
  c->ev_n[s] += 1;
}

// enable: bits 0-7 = mask of KF_STAGE_* to time (0 = off), bits 8-15 = sampling period N (0 or 1: every interval; N: every
// N-th interval of a stage, so the event records perturb a benchmark N times less).  Resets the accumulators.
extern "C" int kf_stage_timers(kf_ctx* c, int enable) {
  if (!c) return KF_ERR_ARG;
  for (int s = 0; s < 8; ++s) { evt_fold(c, s); c->ev_ms[s] = 0.0; c->ev_count[s] = 0; c->ev_seen[s] = 0; c->ev_open[s] = 0; }
  c->timers_enabled = enable & 0xFF;
  c->timers_period = ((enable >> 8) & 0xFF) > 1 ? (unsigned)((enable >> 8) & 0xFF) : 1u;
  c->count_work = (enable >> 16) & 1;
  if (c->count_work) KF_CHECK(hipMemsetAsync(c->counters->rc_steps, 0, 3 * 64 * 16 * sizeof(unsigned long long), c->stream));
  return 0;
}
// out[0] = raycast samples of the reference's march, out[1] = rays whose crossing was evaluated, out[2] = 256-cell marching-cubes
// blocks visited, out[3] = triangles in the buffer -- accumulated since kf_stage_timers(... | 1 << 16).  Blocking.
extern "C" int kf_read_work_counters(kf_ctx* c, uint64_t out[4]) {
  if (!c || !out) return KF_ERR_ARG;
  KfCounters* h = (KfCounters*)malloc(sizeof(KfCounters));
  if (!h) return KF_ERR_ALLOC;
  hipError_t e = hipMemcpyAsync(h, c->counters, sizeof(KfCounters), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) {
    out[0] = out[1] = out[2] = 0;
    for (int i = 0; i < 64; ++i) { out[0] += h->rc_steps[i * 16]; out[1] += h->rc_hits[i * 16]; out[2] += h->mc_blocks[i * 16]; }
    out[3] = h->n_triangles;
  }
  free(h);
  return (int)e;
}
// out_ms[s] = accumulated milliseconds of stage s; counts[s] (may be null) = number of timed intervals.  Blocking.
extern "C" int kf_read_stage_ms(kf_ctx* c, float out_ms[8], uint32_t* counts) {
  if (!c || !out_ms) return KF_ERR_ARG;
  for (int s = 0; s < 8; ++s) { evt_fold(c, s); out_ms[s] = (float)c->ev_ms[s]; if (counts) counts[s] = c->ev_count[s]; }
  return 0;
}
