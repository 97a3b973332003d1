// errors.hip -- per-thread last-error string and ABI version of libmgar_hip.so.
#include "common.hpp"

namespace mgar {
static thread_local const char *g_last_error = "";
void set_error(const char *msg) { g_last_error = msg; }
}  // namespace mgar

extern "C" __attribute__((visibility("default"))) int mgar_abi_version(void) { return 1; }
extern "C" __attribute__((visibility("default"))) const char *mgar_last_error(void) { return mgar::g_last_error; }
