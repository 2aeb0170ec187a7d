// gp_runtime.hip -- HIP runtime glue of the C-ABI: error reporting, streams, memory,
// TempBufferManager and StreamTempBufferRoundRobin.
//
// Replaces (reference, src/gtsam_points/cuda/): check_error.cu, cuda_stream.cu, cuda_memory.cu,
// cuda_buffer.cu, cuda_device_names.cu, cuda_device_sync.cu, stream_roundrobin.cu:10-34,
// stream_temp_buffer_roundrobin.cu:11-80.
#include <cstdio>
#include <cstring>

#include "gp_host.hpp"

namespace gp {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

int hip_fail(hipError_t err, const char* expr, const char* file, int line) {
  char buf[1024];
  snprintf(buf, sizeof(buf), "%s : %s (%s) at %s:%d", hipGetErrorName(err), hipGetErrorString(err), expr, file, line);
  g_last_error = buf;
  return GP_ERROR_HIP;
}

}  // namespace gp

struct gp_stream_pool {
  size_t init_buffer_size = 0;
  std::atomic_int cursor{0};  // StreamRoundRobin::cursor is a std::atomic_int (stream_roundrobin.hpp:28)
  std::vector<hipStream_t> streams;
  std::vector<gp_temp_buffer*> buffers;
};

extern "C" {

const char* gp_last_error(void) { return gp::g_last_error.c_str(); }

const char* gp_version(void) { return "gtsam_points_hip 0.1 (gfx950)"; }

int gp_device_count(int* count) {
  if (!count) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_device_count: null");
  GP_HIP(hipGetDeviceCount(count));
  return GP_OK;
}

int gp_set_device(int device) {
  GP_HIP(hipSetDevice(device));
  return GP_OK;
}

int gp_get_device(int* device) {
  GP_HIP(hipGetDevice(device));
  return GP_OK;
}

int gp_device_name(int device, char* name, size_t name_len) {
  hipDeviceProp_t prop;
  GP_HIP(hipGetDeviceProperties(&prop, device));
  snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  return GP_OK;
}

int gp_device_synchronize(void) {
  GP_HIP(hipDeviceSynchronize());
  return GP_OK;
}

int gp_trim_device_cache(void) {
  gp::BlockCache::get().trim();
  gp::release_side_streams();  // gp_knn.hip: the calling thread's candidate side streams, events and probe words on the current device
  return GP_OK;
}

int gp_stream_create(gp_stream_t* stream) {
  hipStream_t s;
  GP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return GP_OK;
}

int gp_stream_destroy(gp_stream_t stream) {
  // blocks parked in a thread's BlockCache are tagged with the stream they were released on; a later stream may be given the same
  // handle value, so the work of this one is finished before the handle can be recycled
  GP_HIP(hipStreamSynchronize((hipStream_t)stream));
  GP_HIP(hipStreamDestroy((hipStream_t)stream));
  return GP_OK;
}

int gp_stream_synchronize(gp_stream_t stream) {
  GP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return GP_OK;
}

int gp_malloc(void** ptr, size_t bytes) {
  GP_HIP(hipMalloc(ptr, bytes ? bytes : 16));
  return GP_OK;
}

int gp_free(void* ptr) {
  if (ptr) (void)gp_source_mirror_invalidate(ptr);  // the address may be handed out again: no later factor may join a packed mirror built from what lay here
  if (ptr) GP_HIP(hipFree(ptr));
  return GP_OK;
}

int gp_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, gp_stream_t stream) {
  GP_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return GP_OK;
}

int gp_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, gp_stream_t stream) {
  GP_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return GP_OK;
}

int gp_memset(void* dst_dev, int value, size_t bytes, gp_stream_t stream) {
  GP_HIP(hipMemsetAsync(dst_dev, value, bytes, (hipStream_t)stream));
  return GP_OK;
}

int gp_host_malloc(void** ptr, size_t bytes) {
  GP_HIP(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
  return GP_OK;
}

int gp_host_free(void* ptr) {
  if (ptr) GP_HIP(hipHostFree(ptr));
  return GP_OK;
}

// ---- TempBufferManager ------------------------------------------------------------------------

int gp_temp_buffer_create(size_t init_buffer_size, gp_temp_buffer_t** out) {
  if (!out) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_temp_buffer_create: null out");
  auto* tb = new gp_temp_buffer;
  if (init_buffer_size) {
    gp_temp_buffer::Buffer b;
    hipError_t e = hipMalloc((void**)&b.buffer, init_buffer_size);
    if (e != hipSuccess) {
      delete tb;
      return gp::hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    }
    b.size = init_buffer_size;
    tb->buffers.push_back(b);
  }
  *out = tb;
  return GP_OK;
}

// get_buffer(): return the newest buffer if large enough, otherwise allocate 1.2x the request and keep the
// older ones alive (in-flight kernels may still use them), stream_temp_buffer_roundrobin.cu:27-33
int gp_temp_buffer_get(gp_temp_buffer_t* tb, size_t size, void** dev_ptr) {
  if (!tb || !dev_ptr) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_temp_buffer_get: null");
  if (tb->buffers.empty() || tb->buffers.back().size < size) {
    gp_temp_buffer::Buffer b;
    const size_t n = (size_t)((double)size * 1.2) + 256;
    GP_HIP(hipMalloc((void**)&b.buffer, n));
    b.size = n;
    tb->buffers.push_back(b);
  }
  *dev_ptr = tb->buffers.back().buffer;
  return GP_OK;
}

int gp_temp_buffer_clear(gp_temp_buffer_t* tb) {
  if (!tb) return GP_OK;
  while (tb->buffers.size() > 1) {
    (void)hipFree(tb->buffers.front().buffer);
    tb->buffers.erase(tb->buffers.begin());
  }
  return GP_OK;
}

int gp_temp_buffer_clear_all(gp_temp_buffer_t* tb) {
  if (!tb) return GP_OK;
  for (auto& b : tb->buffers) (void)hipFree(b.buffer);
  tb->buffers.clear();
  return GP_OK;
}

int gp_temp_buffer_destroy(gp_temp_buffer_t* tb) {
  if (!tb) return GP_OK;
  gp_temp_buffer_clear_all(tb);
  delete tb;
  return GP_OK;
}

// ---- StreamTempBufferRoundRobin ---------------------------------------------------------------

int gp_stream_pool_create(int num_streams, size_t init_buffer_size, gp_stream_pool_t** out) {
  if (!out || num_streams <= 0) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_stream_pool_create: bad arguments");
  auto* pool = new gp_stream_pool;
  pool->init_buffer_size = init_buffer_size;
  for (int i = 0; i < num_streams; i++) {
    hipStream_t s;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
      gp_stream_pool_destroy(pool);
      return gp::hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
    }
    pool->streams.push_back(s);
    gp_temp_buffer* tb = nullptr;
    int rc = gp_temp_buffer_create(init_buffer_size, &tb);
    if (rc != GP_OK) {
      gp_stream_pool_destroy(pool);
      return rc;
    }
    pool->buffers.push_back(tb);
  }
  *out = pool;
  return GP_OK;
}

int gp_stream_pool_get(gp_stream_pool_t* pool, gp_stream_t* stream, gp_temp_buffer_t** buffer) {
  if (!pool || pool->streams.empty()) return gp::fail(GP_ERROR_INVALID_ARGUMENT, "gp_stream_pool_get: empty pool");
  const int i = (pool->cursor++) % (int)pool->streams.size();
  if (stream) *stream = pool->streams[i];
  if (buffer) *buffer = pool->buffers[i];
  return GP_OK;
}

int gp_stream_pool_sync_all(gp_stream_pool_t* pool) {
  if (!pool) return GP_OK;
  for (auto s : pool->streams) GP_HIP(hipStreamSynchronize(s));
  return GP_OK;
}

int gp_stream_pool_clear(gp_stream_pool_t* pool) {
  if (!pool) return GP_OK;
  for (auto* tb : pool->buffers) gp_temp_buffer_clear(tb);
  return GP_OK;
}

int gp_stream_pool_clear_all(gp_stream_pool_t* pool) {
  if (!pool) return GP_OK;
  for (auto* tb : pool->buffers) gp_temp_buffer_clear_all(tb);
  return GP_OK;
}

int gp_stream_pool_destroy(gp_stream_pool_t* pool) {
  if (!pool) return GP_OK;
  for (auto s : pool->streams) (void)hipStreamDestroy(s);
  for (auto* tb : pool->buffers) gp_temp_buffer_destroy(tb);
  delete pool;
  return GP_OK;
}

void gp_linearized6_to_f32(const gp_linearized6* in, gp_linearized6_f32* out) {
  out->num_inliers = (int)in->num_inliers;
  out->error = (float)in->error;
  out->pad_[0] = out->pad_[1] = 0.0f;
  for (int i = 0; i < 36; i++) {
    out->H_target[i] = (float)in->H_target[i];
    out->H_source[i] = (float)in->H_source[i];
    out->H_target_source[i] = (float)in->H_target_source[i];
  }
  for (int i = 0; i < 6; i++) {
    out->b_target[i] = (float)in->b_target[i];
    out->b_source[i] = (float)in->b_source[i];
  }
}

}  // extern "C"
