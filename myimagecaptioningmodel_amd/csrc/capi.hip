// Error reporting + version of libcapmi.so.
#include <stdarg.h>
#include <stdlib.h>
#include "common.h"

static thread_local char g_err[512] = "";

extern "C" void capmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* capmi_last_error(void) { return g_err; }
extern "C" int capmi_version(void) { return CAPMI_ABI_VERSION; }

// ------------------------------------------------------------------ deterministic mode (verification)
// CAPMI_DETERMINISTIC=1 (or capmi_set_deterministic(1)): every f32 atomic accumulation of the library -- weight-gradient
// split-K on sub-tile outputs, the embedding scatter, bias column sums, the attention's d fc_10 -- is replaced by a
// fixed-order reduction, so that two runs of the same launch sequence are bit-identical whatever the lanes' timing.
static int g_deterministic = -1;
extern "C" int capmi_deterministic(void) {
    if (g_deterministic < 0) {
        const char* e = getenv("CAPMI_DETERMINISTIC");
        g_deterministic = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return g_deterministic;
}
extern "C" int capmi_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return 0;
}

// ------------------------------------------------------------------ general epilogue (verification)
// CAPMI_NT_GENERAL=1 (or capmi_set_general_epilogue(1)): every NT launch runs the GENERAL epilogue instantiation (class 0) instead
// of its class (convolution forward / data gradient, decoder fc, inference, f32 logits: igemm.hip nt_epilogue EPI).  The classes
// are the same arithmetic with the paths a launch cannot take compiled out; this switch lets a test hold them to the general form
// bit for bit.
static int g_general_epilogue = -1;
extern "C" int capmi_general_epilogue(void) {
    if (g_general_epilogue < 0) {
        const char* e = getenv("CAPMI_NT_GENERAL");
        g_general_epilogue = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return g_general_epilogue;
}
extern "C" int capmi_set_general_epilogue(int on) {
    g_general_epilogue = on ? 1 : 0;
    return 0;
}

// ------------------------------------------------------------------ lane synchronisation
// Device-scope events for ordering two HIP streams of the SAME device (plan lanes).  Created without
// timing and without the system-scope release a default hipEventRecord performs (an L2 write-back that
// shows up as ~6 us of idle queue per record); the data handed from lane to lane never leaves the device.
extern "C" int capmi_event_create(void** event) {
    CAPMI_CHECK(event, "capmi_event_create: null pointer");
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence);
    CAPMI_CHECK(e == hipSuccess, "capmi_event_create: %s", hipGetErrorString(e));
    *event = (void*)ev;
    return 0;
}
// Timing events for per-launch measurements (bench.py's roofline pass): device-scope like the lane events -- a default
// hipEventRecord releases at system scope, i.e. the interval would include the write-back of the kernel's dirty lines and
// ~6 us of idle queue per record.
extern "C" int capmi_event_create_timed(void** event) {
    CAPMI_CHECK(event, "capmi_event_create_timed: null pointer");
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableSystemFence);
    CAPMI_CHECK(e == hipSuccess, "capmi_event_create_timed: %s", hipGetErrorString(e));
    *event = (void*)ev;
    return 0;
}
extern "C" int capmi_event_elapsed_ms(void* start, void* stop, float* ms) {
    CAPMI_CHECK(start && stop && ms, "capmi_event_elapsed_ms: null pointer");
    hipError_t e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    CAPMI_CHECK(e == hipSuccess, "capmi_event_elapsed_ms: %s", hipGetErrorString(e));
    return 0;
}
extern "C" int capmi_event_destroy(void* event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
    return 0;
}
extern "C" int capmi_event_record(void* event, void* stream) {
    hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    CAPMI_CHECK(e == hipSuccess, "capmi_event_record: %s", hipGetErrorString(e));
    return 0;
}
extern "C" int capmi_stream_wait_event(void* stream, void* event) {
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
    CAPMI_CHECK(e == hipSuccess, "capmi_stream_wait_event: %s", hipGetErrorString(e));
    return 0;
}

// A stream of the current device for the side lane.  priority < 0: the lowest priority the device offers
// (the side lane should fill the gaps of the main lane, not take compute units from it), 0: default.
extern "C" int capmi_stream_create(void** stream, int priority) {
    CAPMI_CHECK(stream, "capmi_stream_create: null pointer");
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    CAPMI_CHECK(e == hipSuccess, "capmi_stream_create: %s", hipGetErrorString(e));
    hipStream_t s;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority < 0 ? least : (priority > 0 ? greatest : 0));
    CAPMI_CHECK(e == hipSuccess, "capmi_stream_create: %s", hipGetErrorString(e));
    *stream = (void*)s;
    return 0;
}
