// stream_temp_buffer_roundrobin.hpp -- TempBufferManager / StreamTempBufferRoundRobin
// (cuda/stream_temp_buffer_roundrobin.hpp:19-65, cuda/stream_roundrobin.hpp) over the C-ABI.
#pragma once
#include <memory>
#include <utility>

#include "check_error.hpp"

namespace gtsam_points {

class TempBufferManager {
public:
  using Ptr = std::shared_ptr<TempBufferManager>;
  TempBufferManager(size_t init_buffer_size = 0) : owned(true) { check_error << gp_temp_buffer_create(init_buffer_size, &h); }
  TempBufferManager(gp_temp_buffer_t* borrowed, bool) : h(borrowed), owned(false) {}
  ~TempBufferManager() {
    if (owned) check_error << gp_temp_buffer_destroy(h);
  }
  TempBufferManager(const TempBufferManager&) = delete;
  TempBufferManager& operator=(const TempBufferManager&) = delete;

  char* get_buffer(size_t buffer_size) {
    void* p = nullptr;
    check_error << gp_temp_buffer_get(h, buffer_size, &p);
    return static_cast<char*>(p);
  }
  void clear() { check_error << gp_temp_buffer_clear(h); }
  void clear_all() { check_error << gp_temp_buffer_clear_all(h); }
  gp_temp_buffer_t* handle() const { return h; }

private:
  gp_temp_buffer_t* h = nullptr;
  bool owned;
};

class StreamTempBufferRoundRobin {
public:
  StreamTempBufferRoundRobin(int num_streams = 4, size_t init_buffer_size = 512 * 1024) { check_error << gp_stream_pool_create(num_streams, init_buffer_size, &h); }
  ~StreamTempBufferRoundRobin() { check_error << gp_stream_pool_destroy(h); }
  StreamTempBufferRoundRobin(const StreamTempBufferRoundRobin&) = delete;
  StreamTempBufferRoundRobin& operator=(const StreamTempBufferRoundRobin&) = delete;

  std::pair<CUstream_st*, TempBufferManager::Ptr> get_stream_buffer() {
    gp_stream_t s = nullptr;
    gp_temp_buffer_t* b = nullptr;
    check_error << gp_stream_pool_get(h, &s, &b);
    return {reinterpret_cast<CUstream_st*>(s), std::make_shared<TempBufferManager>(b, false)};
  }
  void sync_all() { check_error << gp_stream_pool_sync_all(h); }
  void clear() { check_error << gp_stream_pool_clear(h); }
  void clear_all() { check_error << gp_stream_pool_clear_all(h); }

private:
  gp_stream_pool_t* h = nullptr;
};

}  // namespace gtsam_points
