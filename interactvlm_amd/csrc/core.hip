// Library identity + error strings for libivlm_hip.so.
#include "ivlm_common.h"

static thread_local char g_last_hip_error[512] = "";
// kernel-attached timing events for the launches of the NEXT calls on this thread (ivlm_profile_launches): the start event
// goes to the first instrumented launch, the stop event to every one (it ends up holding the completion of the last)
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
static thread_local int g_prof_launches = 0;

bool ivlm_profile_events(hipEvent_t* start, hipEvent_t* stop) {
    if (!g_prof_stop) return false;
    *start = g_prof_start;  // (null after the first launch)
    *stop = g_prof_stop;
    g_prof_start = nullptr;
    ++g_prof_launches;
    return true;
}

extern "C" {

void ivlm_set_last_hip_error(int code, const char* where) {
    snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s (hipError %d) at %s",
             hipGetErrorString((hipError_t)code), code, where ? where : "?");
}

const char* ivlm_last_hip_error(void) { return g_last_hip_error; }

int ivlm_profile_launches(void* start_event, void* stop_event) {
    const int n = g_prof_launches;
    g_prof_start = static_cast<hipEvent_t>(start_event);
    g_prof_stop = static_cast<hipEvent_t>(stop_event);
    g_prof_launches = 0;
    return n;
}

// 2: fp32 residual streams / fp32 activation flags (round 2); 3: ivlm_sam_block.rel_cat padded to a multiple of 64 rows and read by
// the window attention's table mode (round 3); 4: rel_cat also read by the 64 x 64 grid's table mode (>= 254 rows), fp16-operand
// entry points (*_f16), larger streaming-lift workspaces (one slab per block), ivlm_attention_f16_qsplit (round 4)
int ivlm_abi_version(void) { return 5; }

const char* ivlm_build_arch(void) { return "gfx950"; }

const char* ivlm_error_string(int code) {
    switch (code) {
        case IVLM_OK: return "ok";
        case IVLM_ERR_INVALID_ARG: return "invalid argument";
        case IVLM_ERR_WORKSPACE: return "workspace too small";
        case IVLM_ERR_LAUNCH: return "kernel launch / HIP runtime error";
        case IVLM_ERR_UNSUPPORTED: return "unsupported dtype or configuration";
        default: return "unknown error";
    }
}

}  // extern "C"
