// Device memory of prism_gpu_execute_noalloc, the entry point for callers that cannot call
// hipMalloc (code under HIP graph capture, serving loops, harnesses that forbid it): a fixed
// array in the code object instead of blocks from the scratch pool.  Every GPU the library
// runs on holds its own copy, at its own address, until the process ends.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include "engine.hpp"

namespace prism {

namespace {

// descriptors | counters | accumulators (engine.hpp: kNoAllocBytes = 9 MiB)
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];

struct PerDevice {
  void* ptr = nullptr;
  std::mutex call;
};
std::mutex g_mu;
std::map<int, PerDevice> g_devices;

}  // namespace

// In host code `scratch` is a stand-in, not a device address: the address of the array on
// the current device comes from hipGetSymbolAddress, once per device.
void noalloc_init(int device) {
  std::lock_guard<std::mutex> g(g_mu);
  PerDevice& d = g_devices[device];
  if (d.ptr) return;
  PRISM_HIP(hipSetDevice(device));
  void* p = nullptr;
  PRISM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
  if (!p) fail(PRISM_GPU_EHIP, "hipGetSymbolAddress returned no address for the library's device array");
  d.ptr = p;
}

void* noalloc_scratch(int device) {
  std::lock_guard<std::mutex> g(g_mu);
  auto it = g_devices.find(device);
  if (it == g_devices.end() || !it->second.ptr) fail(PRISM_GPU_EINTERNAL, "the library's device array was not set up for this device");
  return it->second.ptr;
}

std::mutex& noalloc_mutex(int device) {
  std::lock_guard<std::mutex> g(g_mu);
  return g_devices[device].call;
}

}  // namespace prism
