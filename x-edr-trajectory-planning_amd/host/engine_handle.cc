#include "engine_handle.h"

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace tpamd {
namespace {
std::mutex g_mutex;
std::mutex g_create_mutex;
tpamd_engine *g_engine = nullptr;
}  // namespace

tpamd_engine *shared_engine() {
  std::lock_guard<std::mutex> lock(g_create_mutex);
  if (g_engine) return g_engine;
  int device = 0;
  if (const char *env = std::getenv("TPAMD_DEVICE")) device = std::atoi(env);
  const int rc = tpamd_engine_create(device, &g_engine);
  if (rc != 0) {
    std::fprintf(stderr, "[tpamd host] cannot create the GPU engine on device %d: %s\n", device,
                 tpamd_error_string(rc));
    g_engine = nullptr;
  }
  return g_engine;
}

void engine_lock() { g_mutex.lock(); }
void engine_unlock() { g_mutex.unlock(); }

}  // namespace tpamd
