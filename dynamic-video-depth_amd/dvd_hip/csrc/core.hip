// Library-wide pieces of libdvd_hip.so: error channel, ABI version, device query.
#include "dvd_common.h"

namespace dvd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ void zero_words_kernel(unsigned* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// Small device-side clears (operand-scale scalars, headers) are KERNELS, not hipMemsetAsync: inside a captured HIP graph a
// memset node was seen to run out of order with the kernels around it once the graph's pool re-used the block (round 3:
// a convolution read a weight-scale header that the memset of a LATER packing zeroed; eager execution was unaffected).
int zero_words(void* p, int n_words, hipStream_t stream) {
  hipLaunchKernelGGL(zero_words_kernel, dim3((n_words + 63) / 64), dim3(64), 0, stream, static_cast<unsigned*>(p), n_words);
  DVD_LAUNCH_OK();
  return DVD_OK;
}

// Algorithmic-work accounting of the matrix kernels (round 5; bench.py's roofline_mfma): every launch of a convolution /
// weight-gradient kernel adds its 2 * MACs to the counter of the kernel CLASS it dispatched to, on the host, at launch (or
// graph-capture) time.  Process wide, relaxed: a measurement aid, not part of any result.
static double g_flops[DVD_FLOP_CLASSES] = {0};
void flops_add(int cls, double flops) {
  if (cls >= 0 && cls < DVD_FLOP_CLASSES) g_flops[cls] += flops;
}

static double g_bytes[DVD_BYTES_CLASSES] = {0};
void bytes_add(int cls, double bytes) {
  if (cls >= 0 && cls < DVD_BYTES_CLASSES) g_bytes[cls] += bytes;
}

}  // namespace dvd

extern "C" {

int dvd_byte_counters(double* out, int n, int reset) {
  DVD_REQUIRE(out && n > 0, "byte_counters: null output");
  for (int i = 0; i < n; ++i) out[i] = i < DVD_BYTES_CLASSES ? dvd::g_bytes[i] : 0.0;
  if (reset)
    for (int i = 0; i < DVD_BYTES_CLASSES; ++i) dvd::g_bytes[i] = 0.0;
  return DVD_OK;
}

int dvd_abi_version(void) { return DVD_ABI_VERSION; }

int dvd_flop_counters(double* out, int n, int reset) {
  DVD_REQUIRE(out && n > 0, "flop_counters: null output");
  for (int i = 0; i < n; ++i) out[i] = i < DVD_FLOP_CLASSES ? dvd::g_flops[i] : 0.0;
  if (reset)
    for (int i = 0; i < DVD_FLOP_CLASSES; ++i) dvd::g_flops[i] = 0.0;
  return DVD_OK;
}

const char* dvd_last_error(void) { return dvd::g_err; }

int dvd_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

}  // extern "C"
