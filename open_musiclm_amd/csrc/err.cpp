// thread-local last-error string for the C-ABI (include/omlm.h: omlm_last_error)
#include <string.h>
static thread_local char g_err[512] = "";
extern "C" void omlm_set_error(const char* msg) { strncpy(g_err, msg, sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* omlm_last_error(void) { return g_err; }
extern "C" int omlm_version(void) { return 100; }
