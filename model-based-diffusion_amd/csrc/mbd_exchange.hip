// mbd_exchange.hip — the in-library exchange (include/mbd_hip.h): the one collective of a sharded diffusion step as peer
// stores into hipIpc-mapped fine-grained windows plus epoch flags.
#include "mbd_internal.h"

// ---- in-library exchange (include/mbd_hip.h) -------------------------------------------------------------------------
// Window of a rank: [2 parities][rows][N] floats, then [2][world] flag words.  Step e (1, 2, ...) uses parity e & 1: a
// peer can only be one step ahead of its slowest peer (it needs everybody's flags of step e to finish step e), and its
// push of step e + 1 follows its own reads of step e - 1's values in stream order, so two parities never collide.
namespace {
struct XchgPeers {
  float* win[MBD_EXCHANGE_MAX_RANKS];
};
__global__ __launch_bounds__(256) void exchange_push_kernel(XchgPeers P, int world, int rank, int rows, int shard, int N,
                                                            unsigned epoch, const float* __restrict__ local) {
  const int dst = blockIdx.x, par = (int)(epoch & 1u);
  float* __restrict__ w = P.win[dst] + (size_t)par * rows * N;
  for (int e = threadIdx.x; e < rows * shard; e += blockDim.x) {
    const int r = e / shard, j = e - r * shard;
    __hip_atomic_store(w + (size_t)r * N + (size_t)rank * shard + j, local[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* flags = reinterpret_cast<unsigned*>(P.win[dst] + (size_t)2 * rows * N);
    __hip_atomic_store(flags + par * world + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// waits for the world's flags of `epoch` (bounded: ~2 s of the 100 MHz wall clock, then *err = 1), then copies the window's
// [rows][N] values — system-scope loads, whatever the window's caching — into an ordinary device buffer
__global__ __launch_bounds__(256) void exchange_wait_kernel(const float* win, int world, int rows, int N, unsigned epoch,
                                                            float* __restrict__ out, int* err) {
  const int par = (int)(epoch & 1u);
  const unsigned* flags = reinterpret_cast<const unsigned*>(win + (size_t)2 * rows * N);
  // (a wait that ran into its limit is sticky: later steps do not spin their two seconds again — the caller finds out
  // from mbd_exchange_status, and a run with a dead peer ends in seconds, not in steps x 2 s)
  if ((int)threadIdx.x < world && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flags + par * world + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
      __builtin_amdgcn_s_sleep(8);
      if (wall_clock64() - t0 > 200000000ull) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  const float* __restrict__ w = win + (size_t)par * rows * N;
  for (int e = threadIdx.x; e < rows * N; e += blockDim.x)
    out[e] = __hip_atomic_load(w + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

struct mbd_exchange {
  int device = 0, rank = 0, world = 1, rows = 1, shard = 0, N = 0;
  unsigned epoch = 0;
  float* d_win = nullptr;               // own window
  float* peer[MBD_EXCHANGE_MAX_RANKS];  // every rank's window as seen from here (own: d_win)
  bool opened[MBD_EXCHANGE_MAX_RANKS];
  bool connected = false;
  bool fine_grained = false;            // the window is fine-grained device memory (hipDeviceMallocFinegrained)
  float* d_all = nullptr;               // [rows][N]: what mbd_exchange_all_gather hands out
  int* d_err = nullptr;
  size_t win_bytes = 0;
  mbd_exchange() { for (int r = 0; r < MBD_EXCHANGE_MAX_RANKS; ++r) { peer[r] = nullptr; opened[r] = false; } }
  mbd_exchange(const mbd_exchange&) = delete;
  mbd_exchange& operator=(const mbd_exchange&) = delete;
  ~mbd_exchange() {
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < MBD_EXCHANGE_MAX_RANKS; ++r)
      if (opened[r]) (void)hipIpcCloseMemHandle(peer[r]);
    (void)hipFree(d_win); (void)hipFree(d_all); (void)hipFree(d_err);
  }
};

extern "C" int mbd_exchange_create(int device, int rank, int world, int rows, int shard, mbd_exchange** out) {
  if (!out) return fail(MBD_ERR_INVALID, "out is NULL");
  if (device_count_quiet() < 1) return fail(MBD_ERR_NO_DEVICE, "no HIP device: this library has no CPU fallback");
  if (world < 1 || world > MBD_EXCHANGE_MAX_RANKS || rank < 0 || rank >= world || rows < 1 || rows > 4 || shard < 1)
    return fail(MBD_ERR_INVALID, "exchange: rank %d of %d, %d rows x %d", rank, world, rows, shard);
  static_assert(sizeof(hipIpcMemHandle_t) <= MBD_IPC_HANDLE_BYTES, "IPC handle size");
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<mbd_exchange> guard(new mbd_exchange());
  mbd_exchange* x = guard.get();
  x->device = device; x->rank = rank; x->world = world; x->rows = rows; x->shard = shard; x->N = world * shard;
  x->win_bytes = sizeof(float) * (size_t)2 * rows * x->N + sizeof(unsigned) * (size_t)2 * world;
  // fine-grained device memory: peers' stores (xGMI) and this device's loads meet at system scope without a cached
  // copy in between (the kernels use system-scope accesses; a whole allocation: hipIpcGetMemHandle wants its base
  // either way).  Without a fine-grained pool the window would be ordinary (coarse-grained) memory: a peer's stores
  // can then sit behind a stale line of the owner's L2 whatever the scope of the owner's loads — a flag seen while the
  // rewards beside it are old.  That is refused (the caller keeps the process group's all-gather) unless the lever
  // MBD_EXCHANGE_COARSE_OK=1 asks for it (single-device dry runs on a runtime without the pool).
  x->fine_grained = true;
  if (env_flag("MBD_EXCHANGE_NO_FINEGRAINED") ||
      hipExtMallocWithFlags((void**)&x->d_win, x->win_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    x->d_win = nullptr;
    x->fine_grained = false;
    if (!env_flag("MBD_EXCHANGE_COARSE_OK"))
      return fail(MBD_ERR_UNSUPPORTED, "exchange: no fine-grained device memory on device %d (hipExtMallocWithFlags): "
                  "use the process group's all-gather", device);
    HIP_TRY(hipMalloc(&x->d_win, x->win_bytes));
  }
  HIP_TRY(hipMemset(x->d_win, 0, x->win_bytes));
  HIP_TRY(hipMalloc(&x->d_all, sizeof(float) * (size_t)rows * x->N));
  HIP_TRY(hipMalloc(&x->d_err, sizeof(int)));
  HIP_TRY(hipMemset(x->d_err, 0, sizeof(int)));
  HIP_TRY(hipDeviceSynchronize());
  x->peer[rank] = x->d_win;
  if (world == 1) x->connected = true;
  *out = guard.release();
  return MBD_OK;
}

extern "C" int mbd_exchange_destroy(mbd_exchange* x) {
  delete x;
  return MBD_OK;
}

extern "C" int mbd_exchange_local_handle(mbd_exchange* x, void* handle_out) {
  if (!x || !handle_out) return fail(MBD_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(x->device));
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, x->d_win));
  std::memset(handle_out, 0, MBD_IPC_HANDLE_BYTES);
  std::memcpy(handle_out, &h, sizeof(h));
  return MBD_OK;
}

extern "C" int mbd_exchange_connect(mbd_exchange* x, const void* handles) {
  if (!x || !handles) return fail(MBD_ERR_INVALID, "NULL argument");
  if (x->connected) return fail(MBD_ERR_STATE, "exchange already connected");
  HIP_TRY(hipSetDevice(x->device));
  for (int r = 0; r < x->world; ++r) {
    if (r == x->rank) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + (size_t)r * MBD_IPC_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    x->peer[r] = static_cast<float*>(p);
    x->opened[r] = true;
  }
  x->connected = true;
  return MBD_OK;
}

extern "C" int mbd_exchange_all_gather(mbd_exchange* x, const float* d_local, const float** d_all_out, void* stream_) {
  if (!x || !d_local || !d_all_out) return fail(MBD_ERR_INVALID, "NULL argument");
  if (!x->connected) return fail(MBD_ERR_STATE, "exchange: mbd_exchange_connect first");
  HIP_TRY(hipSetDevice(x->device));
  hipStream_t s = (hipStream_t)stream_;
  x->epoch += 1;
  XchgPeers P;
  for (int r = 0; r < MBD_EXCHANGE_MAX_RANKS; ++r) P.win[r] = x->peer[r];
  hipLaunchKernelGGL(exchange_push_kernel, dim3(x->world), dim3(256), 0, s, P, x->world, x->rank, x->rows, x->shard, x->N,
                     x->epoch, d_local);
  hipLaunchKernelGGL(exchange_wait_kernel, dim3(1), dim3(256), 0, s, (const float*)x->d_win, x->world, x->rows, x->N,
                     x->epoch, x->d_all, x->d_err);
  HIP_TRY(hipGetLastError());
  *d_all_out = x->d_all;
  return MBD_OK;
}

extern "C" int mbd_exchange_fine_grained(const mbd_exchange* x, int* out) {
  if (!x || !out) return fail(MBD_ERR_INVALID, "NULL argument");
  *out = x->fine_grained ? 1 : 0;
  return MBD_OK;
}

extern "C" int mbd_exchange_status(mbd_exchange* x) {
  if (!x) return fail(MBD_ERR_INVALID, "exchange is NULL");
  HIP_TRY(hipSetDevice(x->device));
  HIP_TRY(hipDeviceSynchronize());
  int err = 0;
  HIP_TRY(hipMemcpy(&err, x->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (err) return fail(MBD_ERR_STATE, "exchange: a wait ran into its time limit (rank %d of %d, step %u): a peer never arrived",
                       x->rank, x->world, x->epoch);
  return MBD_OK;
}
